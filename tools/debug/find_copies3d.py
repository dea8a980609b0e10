"""Where do the small device copies (__amd_rocclr_copyBuffer rows of the rocprofv3 summary) of a 3-D step come from?
Two eager steps under torch.profiler with Python stacks; every CPU-side op that owns a Memcpy / Memset activity is
printed with the innermost arco_amd frame.  GRAPHS=1: the trainer's default schedule (captured passes) instead."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from torch.profiler import profile, ProfilerActivity
from arco_amd import train_arco_3d as T

extra = ["--act_dtype", "f16"] if os.environ.get("ACT_DTYPE") == "f16" else []
args = T.build_parser().parse_args(["--batch_size", "2", "--queue_size", "4096", "--synthetic", "1", "--graphs", os.environ.get("GRAPHS", "0"),
                                    "--num_classes", "2"] + extra)
st = T.ArcoStep3D(args, "cuda:0")
l, ll = T.synthetic_volume_batch(2, args.patch_size, 2, 1, "cuda:0")
u, _ = T.synthetic_volume_batch(2, args.patch_size, 2, 2, "cuda:0")
for _ in range(4):
    st.step(l, ll, u)
torch.cuda.synchronize()
N = 2
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(N):
        st.step(l, ll, u)
    torch.cuda.synchronize()
cnt = collections.Counter()
dev = collections.Counter()
for e in prof.events():
    ks = [k for k in getattr(e, "kernels", [])]
    for k in ks:
        nm = k.name
        if "emcpy" in nm or "emset" in nm or "copyBuffer" in nm or "fillBuffer" in nm:
            site = "?"
            for fr in (e.stack or []):
                if "arco_amd/" in fr:
                    site = fr.split("arco_amd/")[-1]
                    break
            cnt[(nm[:40], e.name[:40], site[:90])] += 1
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU and ("emcpy" in e.name or "emset" in e.name or "rocclr" in e.name):
        dev[e.name[:60]] += 1
for (k, op, site), n in sorted(cnt.items(), key=lambda kv: -kv[1]):
    print(f"{n / N:6.1f}/step  {k:40s} {op:40s} {site}")
print("device-side activities:", {k: v / N for k, v in dev.items()})
ka = prof.key_averages()
for r in sorted(ka, key=lambda r: -r.count)[:0]:
    pass

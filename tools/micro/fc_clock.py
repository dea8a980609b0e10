"""In-kernel clock of conv3d_fc_kernel under its own load (MI355X_MICROARCH.md, DVFS notes item 6): a DIAGNOSTIC build of csrc/conv3d_fl.hip
(-DARCO_FC_CLOCK: consumer wave 0 of every workgroup stores s_memtime / s_memrealtime deltas around its whole tile loop into a buffer of its
own - no output depends on them) is launched back to back for >= 2 s on random data; clock = delta(s_memtime) / delta(s_memrealtime) x 100 MHz,
median over workgroups.  Also prints MFMAs per wave and the cycles per MFMA that follow.
  python tools/micro/fc_clock.py build      (here: hipcc cross-compiles build/libarco_hip_fcclock.so)
  ARCO_LIB=build/libarco_hip_fcclock.so python tools/micro/fc_clock.py      (on the GPU box)"""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "build":
    cs = os.path.join(ROOT, "arco_amd", "csrc")
    os.makedirs(os.path.join(ROOT, "build"), exist_ok=True)
    subprocess.check_call(["make", "-C", cs, "-j6"], stdout=subprocess.DEVNULL)
    obj = os.path.join(ROOT, "build", "conv3d_fl_clock.o")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-DARCO_FC_CLOCK", "-c",
                           os.path.join(cs, "conv3d_fl.hip"), "-o", obj])
    objs = [os.path.join(cs, f"{n}.o") for n in ("loss_front", "igemm", "conv_sp", "gemm_sp", "conv_h", "elementwise", "det_scatter", "glue", "sampler_host", "augment")]
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", obj] + objs + ["-lpthread", "-o", os.path.join(ROOT, "build", "libarco_hip_fcclock.so")])
    print("built build/libarco_hip_fcclock.so")
    sys.exit(0)
import ctypes, time
import torch
from arco_amd import ops, _lib as L
ops.CONV_MMA = 3
lib = ctypes.CDLL(L.LIB_PATH)
lib.arco_fc_clock_buffer.argtypes = [ctypes.c_void_p]
for (nv, c, sp) in ((4, 32, (56, 56, 40)), (4, 64, (28, 28, 20))):
    d3, h, w = sp
    x = torch.randn(nv, d3, h, w, c, device="cuda").permute(0, 4, 1, 2, 3)
    wt = torch.randn(c, c, 3, 3, 3, device="cuda") * 0.05
    wp = ops.pack_weight(wt, 27, 0)
    xr, ld = ops.rows_view(x)
    stamps = torch.zeros(2 * 512 * 8, dtype=torch.int64, device="cuda")
    lib.arco_fc_clock_buffer(ctypes.c_void_p(stamps.data_ptr()))
    f = lambda: ops.conv_raw(xr, ld, c, wp, c, nv, h, w, 27, stats=True, d3=d3)
    cfg = L.query("arco_conv_config_mma", 27, nv * d3, h, w, c, c, ld, 3)
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 2.5:
        for _ in range(50): f()
        torch.cuda.synchronize(); n += 50
    dt = (time.perf_counter() - t0) / n * 1e6
    pr = stamps.view(2, 512, 8)[1].cpu()
    s = stamps.view(2, 512, 8)[0].cpu()
    s = s[s[:, 1] > 0]
    clk = (s[:, 0].double() / s[:, 1].double() * 100.0).median().item()          # MHz
    cyc = s[:, 0].double().median().item()
    mf = s[:, 2].double().median().item()
    pr = pr[pr[:, 5] > 0].double(); nint = pr[:, 5].median().item()
    print("   loader waves per chunk interval (cycles, median): barrier wait %.0f, weight DMA issue + wait for the activations %.0f, split + LDS stores %.0f, activation load issue %.0f, wait for the weight burst %.0f" % tuple((pr[:, i].median().item() / nint) for i in range(5)))
    print(f"cfg {cfg} {c}->{c} @{sp} x{nv}: {dt:7.1f} us per launch (host clock, back to back); in-kernel clock {clk:7.1f} MHz (median of {len(s)} workgroups); "
          f"tile loop {cyc:9.0f} cycles, {mf:7.0f} MFMAs per wave -> {cyc / max(mf, 1):5.2f} cycles per MFMA incl. rendezvous and epilogues; "
          f"of the loop: waiting at the chunk rendezvous {s[:, 4].double().median().item() / cyc:5.3f} ({s[:, 4].double().median().item() / max(1.0, s[:, 6].double().median().item()):6.0f} cycles per chunk), "
          f"epilogues {s[:, 5].double().median().item() / cyc:5.3f} ({s[:, 5].double().median().item() / max(1.0, s[:, 3].double().median().item()):6.0f} cycles per tile) "
          f"-> {(cyc - s[:, 4].double().median().item() - s[:, 5].double().median().item()) / max(mf, 1):5.2f} cycles per MFMA in the steps themselves; "
          f"epilogue split per tile: drain + positions {(s[:, 7] >> 32).double().median().item() / max(1.0, s[:, 3].double().median().item()):6.0f}, bias + stores {(s[:, 7] & 0xffffffff).double().median().item() / max(1.0, s[:, 3].double().median().item()):6.0f} cycles, the rest statistics")

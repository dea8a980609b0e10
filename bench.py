"""Benchmark of the ARCO 2-D hot-path training step on MI355X.

  python bench.py --gpus N --steps K --warmup W
N > 1 without a torchrun environment: this process spawns `python -m torch.distributed.run --nproc-per-node N ...
bench.py --gpus N ...` (one rank per GPU over RCCL) BEFORE touching the GPU and relays rank 0's JSON line; under
torchrun (WORLD_SIZE set) it is one rank.  A world size different from --gpus is an error (exit code 2).
Prints ONE JSON line (rank 0).

Workload = BASELINE.json configs[1]: ACDC-shaped 2-D 256x256, --batch_size 8 per stream (16 images/step/GPU), C=4,
D=496, stratified sampler (smc) + 4096-key/class queue, nq=256, nn=512, temp 0.5; synthetic data and random-init weights
resident in HBM before timing; every other flag at the trainer's (= the reference's) default, in particular --k2 1.0:
a step = SURVEY §8a rows N1-N5, T1, L1-L6, O1 plus the §8f row-1 terms the reference trains with by default (CE + Dice,
unsupervised CE, the TPS equivariance term with its extra student pass) and the cutmix mixing of --apply_aug.
Weak scaling: per-GPU work is fixed; value = N*K 16-image steps / wall time (max over ranks).

The timed region runs the product configuration with NO instrumentation.  Afterwards, outside the timed region:
a sustained run (>= --sustain_s seconds), a k2 = 0 run (the north-star path alone), an eager pass with HIP events
around the conv / weight-gradient launches for the roofline objects (graphs off: the kernels must be visible to the
events), sub-records for BASELINE.json configs[2..4] (each in a child process) and the CPU baseline.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

# Host logic needs no CPU parallelism; torch's default (one OpenMP thread per hardware thread, spinning
# after every small CPU op) oversubscribes shared hosts and stalls the launch thread by 20-80 ms at
# random (measured: 40 ms/step steady with 4 threads vs 40-200 ms with 128).  Must precede `import torch`.
if os.environ.get("ARCO_CPU_BASELINE_CHILD") != "1":
    os.environ.setdefault("OMP_NUM_THREADS", "4")
    os.environ.setdefault("MKL_NUM_THREADS", "4")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3        # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, 256 CUs @ 2.4 GHz
PEAK_F16_MFMA_TFLOPS = 2500.0       # dense f16/bf16 MFMA (MI355X_MICROARCH.md); --conv_mma f16 launches are priced against it
PEAK_SPLIT_TFLOPS = 2500.0 / 6.0     # split-bf16 mode: 6 bf16 MFMAs per fp32-accurate product -> 416.7 TFLOP/s of ALGORITHMIC fp32 work
PEAK_BY_MMA = {0: PEAK_F32_MFMA_TFLOPS, 1: PEAK_F16_MFMA_TFLOPS, 2: PEAK_F16_MFMA_TFLOPS, 3: PEAK_SPLIT_TFLOPS, 4: PEAK_F16_MFMA_TFLOPS}
PEAK_HBM_TBS = 8.0
# HBM bytes per launch from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate passes):
# profiles/r01_pmc_conv_traffic.md (3x3 backbone kernel) and profiles/r01_pmc_gemm_traffic.md; key (taps, M, N, K)
PMC_TRAFFIC_BYTES = {(9, 65536, 64, 64): 3.526e7, (9, 65536, 64, 128): 5.322e7,
                     (9, 32768, 64, 64): 1.840e7, (9, 32768, 64, 128): 2.797e7,
                     (1, 1048576, 496, 496): 5.595e9, (1, 262144, 480, 480): 1.240e9}


# (taps, M, N, K) -> HBM bytes per launch of the split-bf16 kernels (profiles/r02_pmc_conv_traffic.md)
# conv3x3_sp_kernel launches, profiles/r02_pmc_conv_sp_traffic.md (keys: taps, M, N, K)
PMC_TRAFFIC_SPLIT = {(9, 65536, 64, 64): 3.647e7, (9, 65536, 64, 128): 5.502e7, (9, 262144, 32, 32): 6.904e7, (9, 262144, 32, 64): 1.0304e8,
                     (9, 1048576, 32, 16): 2.0646e8, (9, 16384, 128, 128): 2.479e7, (9, 32768, 64, 64): 1.948e7,
                     # conv3x3_rw_kernel<8,1> launches, profiles/r02_pmc_conv_rw_traffic.md
                     (9, 1048576, 16, 16): 1.3612e8, (9, 1048576, 16, 32): 2.0341e8, (9, 1048576, 4, 16): 8.500e7, (9, 1048576, 16, 4): 8.579e7,
                     (9, 524288, 16, 16): 6.848e7}
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E ~ 8 TB/s


def conv_algorithmic_bytes(taps, m, n, k, mma):
    """Bytes a conv / GEMM launch must move: the input read once, the output written once (fp32; f16 storage, mma 4: 2 B), the
    packed weights (split-bf16: 6 B per weight, f16 storage: 2 B, else 4 B)."""
    if mma == 4:
        return 2.0 * m * (k + n) + 2.0 * taps * n * k
    return 4.0 * m * (k + n) + (6.0 if mma == 3 else 4.0) * taps * n * k


def _host_cpu():
    """(physical cores, logical CPUs, model string) of this host."""
    model, pairs = "unknown", set()
    try:
        phys = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    pairs.add((phys, core))
                phys = core = None
    except OSError:
        pass
    logical = os.cpu_count() or 1
    physical = len(pairs)
    if not physical:
        try:
            import psutil
            physical = psutil.cpu_count(logical=False) or logical
        except Exception:
            physical = logical
    try:      # a container may be pinned to fewer CPUs than the host has
        logical = min(logical, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    return min(physical, logical), logical, model


def cpu_loss_only(threads, b=2, patch=(256, 256), n_cls=4, D=496, qsize=4096, reps=2):
    """compute_contra_memobank_loss forward + backward ALONE on the host (oracle, torch-CPU fp32), at the
    sample's batch (--batch_size 2: 4 images of 256 x 256, D = 496, 4096-key queues already full, smc, 256 queries x 512
    negatives) - the CPU twin of `contrastive_loss_ms_per_step`.  Returns mean milliseconds per call."""
    import numpy as np
    import arco_oracle as orc
    import fixture_inputs as fx
    rs = np.random.RandomState(7)
    bank = [[torch.from_numpy(rs.standard_normal((qsize, D)).astype(np.float32))] for _ in range(n_cls)]
    ptr = [torch.tensor([qsize]) for _ in range(n_cls)]
    qs = [qsize] * n_cls
    total = 0.0
    for r in range(reps + 1):
        inp = fx.loss_inputs(500 + r, b=b, n_cls=n_cls, feat=D, spatial=patch)
        rep = inp["rep"].clone().requires_grad_(True)
        t0 = time.perf_counter()
        _, loss = orc.compute_contra_memobank_loss(rep, inp["label_l"], inp["label_u"], inp["prob_l"], inp["prob_u"], inp["low_mask"],
                                                   inp["high_mask"], bank, ptr, qs, inp["rep_teacher"], delta_n=0.97, func='smc',
                                                   num_queries=256, num_negatives=512)
        loss.backward()
        if r:                     # the first call warms the allocator / thread pool
            total += time.perf_counter() - t0
    return total / reps * 1e3


def cpu_baseline_child():
    """Runs in a child process (one thread per PHYSICAL core, no GPU): chained oracle steps at --batch_size 2 with the default
    loss terms, then the contrastive loss alone."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cpu_step
    physical, logical, model = _host_cpu()
    # the thread count at which the oracle step is FASTEST on this pool's hosts (2 x 64-core EPYC 9575F), measured in round 4:
    # 16 / 32 / 64 / 128 threads -> 4.6 / 4.3 / 6.6 / 10.7 s per step, loss alone 806 / 769 / 931 / 1461 ms (torch's intra-op
    # pool loses to cross-socket traffic beyond one CCD group); `cores` reports the threads used, physical_cores the host's.
    # ARCO_CPU_THREADS overrides.
    torch.set_num_threads(int(os.environ.get("ARCO_CPU_THREADS", max(1, min(physical, 32)))))
    cpu_step.timed_sample(b=1, patch=(64, 64), k2=1.0, bt=True)          # warm the thread pool / allocator
    secs, threads = cpu_step.timed_sample(b=2, steps=3, k2=1.0, bt=True)
    loss_ms = cpu_loss_only(threads)
    # ... and the same step on ALL physical cores (SURVEY 8d asks for them): one step, reported beside the headline
    secs_all = None
    if physical > threads and "ARCO_CPU_THREADS" not in os.environ:
        torch.set_num_threads(physical)
        secs_all, _ = cpu_step.timed_sample(b=2, steps=1, k2=1.0, bt=True)
    print(json.dumps({"secs": secs, "threads": threads, "steps": 3, "physical_cores": physical, "logical_cpus": logical,
                      "cpu_model": model, "loss_only_ms": loss_ms, "secs_all_cores": secs_all}))


def cpu_baseline():
    """CPU oracle (port) timed on this box's host cores: three chained full steps at --batch_size 2 (4 images), and
    compute_contra_memobank_loss forward + backward alone at the same batch."""
    env = dict(os.environ, ARCO_CPU_BASELINE_CHILD="1")
    env.pop("OMP_NUM_THREADS", None); env.pop("MKL_NUM_THREADS", None)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu_baseline_child"], env=env,
                         capture_output=True, text=True, timeout=1200)
    r = json.loads(out.stdout.strip().splitlines()[-1])
    secs, threads = r["secs"], r["threads"]
    # a 16-image step is 4x the 4-image sample (per-image work is constant)
    return {"value": round(1.0 / (4.0 * secs), 5), "unit": "steps/s (16-image steps)", "cores": threads, "kind": "port",
            "physical_cores": r.get("physical_cores"), "logical_cpus": r.get("logical_cpus"), "cpu_model": r.get("cpu_model"),
            "loss_only_ms": round(r.get("loss_only_ms", float("nan")), 1),
            "all_physical_cores": (None if not r.get("secs_all_cores") else
                                   {"cores": r.get("physical_cores"), "value": round(1.0 / (4.0 * r["secs_all_cores"]), 5),
                                    "seconds_per_4_image_step": round(r["secs_all_cores"], 2),
                                    "note": "one oracle step with one torch thread per physical core (slower than the 32-thread figure on "
                                            "these two-socket hosts: torch's intra-op pool loses to cross-socket traffic); `value` above "
                                            "is the FASTEST thread count, so the GPU / CPU ratio is not flattered"}),
            "loss_only_note": "oracle compute_contra_memobank_loss fwd + bwd alone at --batch_size 2 (4 images, D = 496, full 4096-key "
                              "queues, 256 x 512 samples per class); the GPU twin is contrastive_loss_ms_per_step at 16 images",
            "sample": f"{r.get('steps', 1)} chained full oracle steps at --batch_size 2 (4 images, 256x256, C=4, D=496, cutmix, "
                      f"k2 = 1 equivariance term and batch_transform on, like the GPU headline), {secs:.1f} s per step on {threads} threads "
                      f"({r.get('physical_cores')} physical cores on the host); scaled x4 to the 16-image step"}


def spawn_ranks(a, argv):
    """--gpus N > 1 outside torchrun: launch N ranks as children (nothing here has touched the GPU), relay rank 0's line."""
    n_dev = torch.cuda.device_count()          # counting devices does not initialise the GPU runtime
    if n_dev < a.gpus:
        print(f"bench.py: --gpus {a.gpus} but only {n_dev} GPU(s) visible", file=sys.stderr)
        return 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    p = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
    if p.returncode != 0 or line is None:
        sys.stdout.write(p.stdout)
        print(f"bench.py: the {a.gpus}-rank launch failed (rc={p.returncode})", file=sys.stderr)
        return p.returncode or 1
    if json.loads(line).get("n_gpus") != a.gpus:
        print(f"bench.py: world size {json.loads(line).get('n_gpus')} != --gpus {a.gpus}", file=sys.stderr)
        return 2
    print(line)
    return 0


MMA_NAMES = {0: "fp32 MFMA 16x16x4", 1: "f16-operand MFMA 16x16x16, fp32 accumulate", 2: "bf16-operand MFMA 16x16x16, fp32 accumulate",
             3: "split-bf16 (3 x bf16 = exact fp32 operands) MFMA 16x16x32, fp32 accumulate",
             4: "f16 storage, f16 MFMA 16x16x32, fp32 accumulate"}


def _kernel_name(key):
    if isinstance(key, tuple):
        return f"wgrad<taps={key[1]},co={key[2]},ci={key[3]}> (weight gradient, fp32 MFMA 16x16x4)"
    mma, key = key // 100000000, key % 100000000
    if key >= 50000000:          # ops.conv_raw: the PRO instantiation of the pipelined 3x3 kernels
        return _kernel_name(mma * 100000000 + key - 50000000).replace("> (", ",PRO> (consumer-side BatchNorm + LeakyReLU + dropout of the producing layer applied in the loader; ", 1)
    if mma == 4 and key // 1000000 == 27:
        return "conv3d_image_kernel<3> (one-channel fp32 volume -> 16 channels, f16 output; the 27 taps are the reduction dimension)"
    if mma == 4 and 9400000 <= key < 9500000:
        return (f"hconv_rw_kernel<TH={(key - 9400000) // 1000},N={key % 1000}> (f16 activation storage; persistent workgroups, all 27 taps' weights "
                "resident in LDS, ring of three input planes walked along the depth axis)")
    if mma == 4 and 9260000 <= key < 9270000:
        return (f"hconv_fc_kernel<A_T={key % 10000 // 1000},C_T={key % 1000 // 16}> (f16 activation storage, v_mfma_f32_16x16x32_f16; 3x3x3 as a 3x3 over 3 K "
                "virtual channels on flat tiles, 4 MFMA + 4 LDS-DMA loader waves, one rendezvous per 16-channel chunk)")
    if mma == 4 and not (9700000 <= key < 9900000):
        flat = 9500000 <= key < 9700000
        base = key - (9500000 if flat else (key // 1000000) * 1000000)
        return (f"hconv_kernel<{key // 1000000},{base // 1000},{base % 1000},...{',FLAT' if flat else ''}> (f16 activation storage, "
                "v_mfma_f32_16x16x32_f16, fp32 accumulate; 3x3x3 depth taps looped)")
    if 9700000 <= key < 9900000:
        return f"conv3x3_image_kernel<Cin={key % 100000 // 1000}> (few-channel input: the taps are the reduction dimension; {MMA_NAMES[mma]})"
    if 9500000 <= key < 9700000:        # flat-position tiles of the 3x3x3 kernels (narrow planes)
        return f"igemm_kernel<9,{(key - 9500000) // 1000},{key % 1000},...,FLAT,MMA={mma}> ({MMA_NAMES[mma]} implicit GEMM, 3x3x3 depth taps looped)"
    if 9270000 <= key < 9300000:        # conv3d_fl.hip: the pipelined flat-tile 3x3x3 kernels
        form = {7: ("conv3d_fl_kernel", "one rendezvous per tap-pair step, LDS-DMA weight ring"),
                8: ("conv3d_dw_kernel", "depth-walking columns, three accumulator sets"),
                9: ("conv3d_fc_kernel", "one rendezvous per 16-channel chunk, double-buffered chunk weights")}[key // 10000 % 10]
        a_t = key % 10000 // 1000
        ncw = 8 if a_t > 5 else 4          # (conv3d_fc_kernel with eight MFMA waves: ids A_T + 5)
        return (f"{form[0]}<A_T={a_t - 5 if a_t > 5 else a_t},C_T={key % 1000 // 16}{',NCW=8' if ncw == 8 else ''}> ({MMA_NAMES[mma]}; 3x3x3 as a 3x3 over 3 K virtual channels on flat "
                f"tiles, persistent workgroups of {ncw} MFMA + 4 loader waves, {form[1]})")
    if 9450000 <= key < 9500000:
        return f"conv3d_rw16_kernel ({MMA_NAMES[mma]}; 3x3x3 16 -> 16: persistent workgroups, 27 taps' weights resident in LDS, ring of three input planes along the depth axis)"
    if 9350000 <= key < 9400000:
        th = 4 * ((key - 9350000) // 1000)            # the id names the tile height; the instantiation depends on ARCO_CONV_RW8 (csrc/conv_sp.hip, default 2)
        rw8 = int(os.environ.get("ARCO_CONV_RW8", "2"))
        inst = f"A_T={th // 8},C_T={key % 1000 // 16},NCW=8" if rw8 == 2 else f"A_T={th // 4},C_T={key % 1000 // 16}"
        return (f"conv3x3_rw_kernel<{inst}> ({MMA_NAMES[mma]}; persistent workgroups, {th} x 16-pixel tiles, "
                f"{'8' if rw8 == 2 else '4 (8 at K <= 16)' if rw8 == 1 else '4'} MFMA + 4 loader waves, weights resident in LDS)")
    if 9300000 <= key < 9350000:
        return f"conv3x3_sp_kernel<A_T={(key - 9300000) // 1000},C_T={key % 1000 // 16}> ({MMA_NAMES[mma]}; persistent workgroups, 4 MFMA + 4 loader waves, LDS-DMA weight ring)"
    if key >= 9900000:
        return f"conv3x3_halo_kernel<{(key - 9900000) // 1000},{key % 1000},..> ({MMA_NAMES[mma]} implicit GEMM)"
    return f"igemm_kernel<{key // 1000000},{key // 1000 % 1000},{key % 1000},...,MMA={mma}> ({MMA_NAMES[mma]} implicit GEMM)"


def roofline_from_profile(prof, n_steps, step_ms):
    """Roofline objects from ops.PROFILE of an eager pass over n_steps steps (every conv / weight-gradient launch
    counted, every PROFILE_EVERY-th bracketed by HIP events on the launch stream).
    dominant kernel = the 3x3 backbone instantiation with the largest total time (rocprofv3's top MFMA row for the same
    command); achieved = its algorithmic FLOP per launch / its average launch duration.  `step_ms` is the step time of
    the UN-instrumented product run: shares and the whole-step figure are quoted against it."""
    if not prof:
        return None, None
    prof = {c: v for c, v in prof.items() if c != "__work__"}
    avg = {c: sum(s_.elapsed_time(e_) for s_, e_, _, _ in v["timed"]) / max(1, len(v["timed"])) for c, v in prof.items()}
    tot = {c: avg[c] * v["n"] for c, v in prof.items()}
    is3 = lambda c: (not isinstance(c, tuple)) and c % 50000000 // 1000000 == 9
    cand = {c: t for c, t in tot.items() if is3(c)} or tot
    # the dominant 3x3 kernel is chosen by FAMILY (an instantiation and its PRO twin - consumer-side activation in the loader - are one
    # template; ops.conv_raw records them apart): the family with the largest total time, reported through its plain member, so that
    # the `roofline` object names the same kernel from round to round
    fam = lambda c: c if isinstance(c, tuple) else (c - 50000000 if c % 100000000 >= 50000000 else c)
    fam_tot = {}
    for c, t in cand.items():
        fam_tot[fam(c)] = fam_tot.get(fam(c), 0.0) + t
    best = max(fam_tot, key=fam_tot.get)
    cfg = best if best in cand else max((c for c in cand if fam(c) == best), key=cand.get)
    rec = prof[cfg]
    launches = rec["timed"]
    avg_ms = avg[cfg]
    avg_flop = sum(f for _, _, f, _ in launches) / len(launches)
    ach = avg_flop / (avg_ms * 1e-3) / 1e12
    shapes = {}
    for s_, e_, f, shp in launches:
        d_ = shapes.setdefault(shp, [0, 0.0, f]); d_[0] += 1; d_[1] += s_.elapsed_time(e_)
    (taps, m, n, k), (cnt, ms_sum, f) = max(shapes.items(), key=lambda kv: kv[1][1])
    fam_ms = sum(tot.values()); fam_flop = sum(v["flop"] for v in prof.values())
    wg_ms = sum(t for c, t in tot.items() if isinstance(c, tuple)); wg_flop = sum(v["flop"] for c, v in prof.items() if isinstance(c, tuple))
    k_mma = 0 if isinstance(cfg, tuple) else cfg // 100000000
    peak = PEAK_BY_MMA[k_mma]
    # which roofline bounds the kernel: its arithmetic intensity (algorithmic FLOP per algorithmic byte, over the timed
    # launches) against the machine balance peak FLOP/s / peak HBM B/s
    avg_bytes = sum(conv_algorithmic_bytes(*shp, k_mma) for _, _, _, shp in launches) / len(launches)
    intensity, balance = avg_flop / avg_bytes, peak * 1e12 / (HBM_PEAK_GBS * 1e9)
    if intensity < balance:
        gbs = avg_bytes / (avg_ms * 1e-3) / 1e9
        head = {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                "mfma_tflops": round(ach, 2), "mfma_frac": round(ach / peak, 4)}
    else:
        head = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4)}
    head.update({"flop_per_algorithmic_byte": round(intensity, 1), "machine_balance_flop_per_byte": round(balance, 1),
                 "avg_algorithmic_bytes_per_launch": avg_bytes})
    # HBM bytes per launch from the PMC passes (profiles/*_traffic.md), averaged over the SAME timed launches as `achieved`
    # (a launch class without a PMC row counts with the mean measured ratio of the classes that have one)
    table = PMC_TRAFFIC_BYTES if k_mma == 0 else PMC_TRAFFIC_SPLIT
    known = [(table[shp], conv_algorithmic_bytes(*shp, k_mma)) for _, _, _, shp in launches if shp in table]
    traffic = ratio = by_shape = None
    if known:
        mean_ratio = sum(t for t, _ in known) / sum(a_ for _, a_ in known)
        traffic = sum(table.get(shp, conv_algorithmic_bytes(*shp, k_mma) * mean_ratio) for _, _, _, shp in launches) / len(launches)
        ratio = round(traffic / avg_bytes, 4)
        by_shape = {f"{shp[0]}x{shp[1]}x{shp[2]}x{shp[3]}": round(table[shp] / conv_algorithmic_bytes(*shp, k_mma), 4)
                    for shp in sorted({l_[3] for l_ in launches}) if shp in table}
    roof = {**head, "traffic": traffic, "traffic_over_algorithmic": ratio, "traffic_over_algorithmic_by_shape": by_shape,
            "traffic_source": ("table: HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE, gfx950 units) measured ONCE per launch shape in separate "
                               "rocprofv3 --pmc passes (tools/pmc_conv*.py; summaries profiles/r0N_pmc_*_traffic.md, newest profiles/r06_pmc_traffic.md, re-measured at the last commit for the eight-MFMA-wave instantiation) "
                               "and averaged here over the launches timed in THIS run - bench.py itself collects no counters (a --pmc pass "
                               "cannot run inside this process)"),
            "peak_note": {0: "fp32 MFMA peak", 3: "dense bf16 MFMA peak / 6 (six bf16 MFMAs per fp32-accurate product); the native fp32 MFMA peak is 157.3"}.get(k_mma, "dense f16/bf16 MFMA peak"),
            "kernel": _kernel_name(cfg), "avg_launch_ms": round(avg_ms, 4), "launches_per_step": round(rec["n"] / n_steps, 1),
            "launches_timed": len(launches), "avg_flop_per_launch": avg_flop,
            "share_of_step": round(tot[cfg] / n_steps / step_ms, 4),
            "largest_shape": {"taps": taps, "M": m, "N": n, "K": k, "launches_timed": cnt,
                              "tflops": round(f / (ms_sum / cnt * 1e-3) / 1e12, 2)},
            "method": "eager pass outside the timed region, HIP events on the launch stream around every 5th launch"}
    # the largest-time 3x3 instantiation under the OTHER roofline rides along (e.g. the matrix-pipe-bound 64-channel kernel
    # when the dominant one is HBM-bound), so that lines stay comparable across rounds
    def _bound_of(c):
        l_ = prof[c]["timed"]
        if not l_ or isinstance(c, tuple):
            return None
        mm = c // 100000000
        fl = sum(f for _, _, f, _ in l_) / len(l_)
        by = sum(conv_algorithmic_bytes(*shp, mm) for _, _, _, shp in l_) / len(l_)
        return ("hbm" if fl / by < PEAK_BY_MMA[mm] * 1e12 / (HBM_PEAK_GBS * 1e9) else "mfma"), fl, by, mm
    others = {c: t for c, t in cand.items() if c != cfg and _bound_of(c) and _bound_of(c)[0] != roof["bound"]}
    if others:
        c2 = max(others, key=others.get)
        b2, fl2, by2, mm2 = _bound_of(c2)
        if b2 == "mfma":
            a2 = fl2 / (avg[c2] * 1e-3) / 1e12
            second = {"bound": "mfma", "achieved": round(a2, 2), "peak": round(PEAK_BY_MMA[mm2], 1), "unit": "TFLOP/s", "frac": round(a2 / PEAK_BY_MMA[mm2], 4)}
        else:
            a2 = by2 / (avg[c2] * 1e-3) / 1e9
            second = {"bound": "hbm", "achieved": round(a2, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(a2 / HBM_PEAK_GBS, 4)}
        second.update({"kernel": _kernel_name(c2).split(" (")[0], "avg_launch_ms": round(avg[c2], 4),
                       "launches_per_step": round(prof[c2]["n"] / n_steps, 1), "share_of_step": round(tot[c2] / n_steps / step_ms, 4)})
        roof["largest_kernel_under_the_other_roofline"] = second
    if roof["bound"] == "hbm":
        roof["peak_note"] = ("HBM3E peak ~ 8 TB/s; achieved = algorithmic bytes (input + output + packed weights) / average launch time. "
                             "mfma_tflops / mfma_frac price the same launches against " + roof["peak_note"])
    top = sorted(tot.items(), key=lambda kv: -kv[1])[:8]
    whole = {"mfma_flop_per_step": fam_flop / n_steps, "ms_per_step": round(step_ms, 3),
             "tflops_over_whole_step": round(fam_flop / n_steps / (step_ms * 1e-3) / 1e12, 2),
             "frac_of_fp32_mfma_peak": round(fam_flop / n_steps / (step_ms * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
             "mfma_kernels": {"ms_per_step": round(fam_ms / n_steps, 3), "tflops": round(fam_flop / (fam_ms * 1e-3) / 1e12, 2)},
             "weight_gradient_kernels": {"ms_per_step": round(wg_ms / n_steps, 3),
                                         "tflops": round(wg_flop / max(wg_ms * 1e-3, 1e-9) / 1e12, 2)},
             "top_kernels": [{"kernel": _kernel_name(c).split(" (")[0], "ms_per_step": round(t / n_steps, 3),
                              "tflops": round(prof[c]["flop"] / (t * 1e-3) / 1e12, 1)} for c, t in top],
             "note": "FLOP = MFMA work actually launched (forward + data gradient + weight gradient of every conv / GEMM, "
                     "incl. the passes that are graph-replayed in the product run); SURVEY 8d's 7.4 TFLOP/step counts the "
                     "dense reference dataflow (the row-sparse head and lazy teacher remove ~3.3 TFLOP of it, DESIGN.md 4)"}
    return roof, whole


def eager_profile(stepper, run_steps, n_steps):
    """n_steps steps with every graph off and HIP-event instrumentation on; returns ops.PROFILE."""
    from arco_amd import graphs, ops
    prev = graphs.set_enabled(stepper, False)
    run_steps(2)
    torch.cuda.synchronize()
    ops.PROFILE, ops.PROFILE_EVERY = {}, 5
    ops.WORK = {"bytes": 0.0, "flop": 0.0, "launches": 0}
    run_steps(n_steps)
    torch.cuda.synchronize()
    prof, ops.PROFILE = ops.PROFILE, None
    work, ops.WORK = ops.WORK, None
    graphs.set_enabled(stepper, prev)
    prof["__work__"] = {k: v / n_steps for k, v in work.items()}
    return prof


def step_roofline(work, step_ms, split=True):
    """The WHOLE step against both rooflines: the algorithmic HBM bytes (every operand of a launch counted once) and
    the FLOP of every launch of the convolution / weight-gradient / BatchNorm / pooling / resize families of one step (ops.WORK, counted
    in the eager pass - the same launches the graphs replay), divided by the step time of the un-instrumented run."""
    if not work:
        return None
    peak_f = PEAK_SPLIT_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
    gbs = work["bytes"] / (step_ms * 1e-3) / 1e9
    tf = work["flop"] / (step_ms * 1e-3) / 1e12
    t_hbm, t_mfma = work["bytes"] / (HBM_PEAK_GBS * 1e9) * 1e3, work["flop"] / (peak_f * 1e12) * 1e3
    return {"algorithmic_bytes_per_step": work["bytes"], "flop_per_step": work["flop"], "launches_counted": round(work["launches"], 1),
            "ms_per_step": round(step_ms, 3),
            "hbm": {"achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4)},
            "mfma": {"achieved": round(tf, 2), "peak": round(peak_f, 1), "unit": "TFLOP/s", "frac": round(tf / peak_f, 4)},
            "lower_bound_ms": {"hbm": round(t_hbm, 3), "mfma": round(t_mfma, 3), "max": round(max(t_hbm, t_mfma), 3)},
            "frac_of_lower_bound": round(max(t_hbm, t_mfma) / step_ms, 4),
            "note": "bytes: inputs + outputs + weights of every counted launch once (what a perfectly cached kernel must move); the BatchNorm "
                    "backward counts its two passes (5 tensor crossings); loss front end, glue, augmentation and optimiser launches "
                    "(< 10 % of the kernel time) are not counted.  The step is a dependent chain of ~800 launches on two streams: "
                    "frac_of_lower_bound is the whole-step roofline fraction"}


def timed(run_steps, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(n)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


# ---------------------------------------------------------------------------------------------------------------
# sub-records: BASELINE.json configs[2..4] on ONE GPU, each in a child process (`bench.py --sub NAME`)
# ---------------------------------------------------------------------------------------------------------------
SUBS = {   # every entry: kind, batch_size, patch, classes, mma (+ in_chns for 2-D)
    "config3d_la_vnet": dict(kind="3d", batch_size=2, patch=[112, 112, 80], classes=2, mma="f32x3",
                             workload="LA 3D V-Net 112x112x80, --batch_size 2 (4 volumes/step), C=2, D=16, asmc "
                                      "(BASELINE.json configs[2])"),
    "cityscapes_19c_512x1024": dict(kind="2d", batch_size=1, patch=[512, 1024], classes=19, in_chns=3, mma="f32x3",
                                    workload="Cityscapes-shaped 19-class 3x512x1024, 2 images/GPU (--batch_size 1), D=496, "
                                             "4096-key/class queue (the per-GPU shard of BASELINE.json configs[3])"),
    "dropin_dense_dataflow": dict(kind="2d", batch_size=8, patch=[256, 256], classes=4, in_chns=1, mma="f32x3",
                                  extra=["--dense_head", "1", "--dense_teacher", "1", "--graphs", "0", "--func", "smc"],
                                  workload="the headline workload (BASELINE.json configs[1]) in the DENSE reference dataflow - what "
                                           "the reference's own train_arco_2d.py drives when it binds this package through dropin/ "
                                           "(dense 496-channel FeatureExtractor / q_representation maps, dense teacher, no HIP graphs)"),
    "lits_160x160x96_f16": dict(kind="3d", batch_size=1, patch=[160, 160, 96], classes=2, mma="f32x3", act="f16",
                                workload="LiTS-shaped 3D 2-class 160x160x96, 1+1 volumes/GPU, --act_dtype f16: f16 activation "
                                         "storage + f16 matrix cores in the V-Net, fp32 heads / losses / statistics / optimizer "
                                         "(the per-GPU shard of BASELINE.json configs[4])"),
}


def run_sub(name, steps):
    import random
    import numpy as np
    from arco_amd import ops
    cfg = SUBS[name]
    random.seed(1337); np.random.seed(1337); torch.manual_seed(1337)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    b = cfg["batch_size"]
    if cfg["kind"] == "3d":
        from arco_amd import train_arco_3d as T3
        args = T3.build_parser().parse_args(["--batch_size", str(b), "--queue_size", "4096", "--synthetic", "1",
                                             "--num_classes", str(cfg["classes"]), "--conv_mma", cfg["mma"],
                                             "--act_dtype", cfg.get("act", "f32")])
        args.patch_size = cfg["patch"]
        st = T3.ArcoStep3D(args, dev)
        l, ll = T3.synthetic_volume_batch(b, args.patch_size, cfg["classes"], 1, dev)
        u, _ = T3.synthetic_volume_batch(b, args.patch_size, cfg["classes"], 2, dev)
    else:
        from arco_amd import train_arco_2d as T
        args = T.build_parser().parse_args(["--batch_size", str(b), "--queue_size", "4096", "--synthetic", "1",
                                            "--num_classes", str(cfg["classes"]), "--in_chns", str(cfg["in_chns"]), "--conv_mma", cfg["mma"]]
                                           + list(cfg.get("extra", [])))
        args.patch_size = cfg["patch"]
        st = T.ArcoStep2D(args, dev)
        l, ll = T.synthetic_batch(b, args.patch_size, cfg["classes"], 1, dev, in_chns=cfg["in_chns"])
        u, _ = T.synthetic_batch(b, args.patch_size, cfg["classes"], 2, dev, in_chns=cfg["in_chns"])

    def run_steps(n):
        for _ in range(n):
            st.step(l, ll, u, 0, 100)
    run_steps(5)
    t_pre = time.perf_counter()          # past the power-management transient (see main()): >= 2.5 s of load before timing
    while time.perf_counter() - t_pre < 2.5:
        run_steps(4)
        torch.cuda.synchronize()
    sb0 = dict(ops.sparse_bwd_stats)
    ms = timed(run_steps, steps)
    sb = {k: round((ops.sparse_bwd_stats[k] - sb0[k]) / max(1, steps), 2) for k in sb0}       # 1x1-conv backward calls per step by route
    mem = torch.cuda.max_memory_allocated() / 1e9
    if cfg["kind"] == "2d":
        T.TEACHER_SIDE = 0            # single-stream eager pass for the per-kernel timing (see main())
    else:
        T3.PASS_SIDE = 0
    prof = eager_profile(st, run_steps, 2)
    roof, whole = roofline_from_profile(prof, 2, ms)
    terms = {k: round(float(v), 5) for k, v in st.last_terms.items()}
    print(json.dumps({"sub": name, "workload": cfg["workload"], "ms_per_step": round(ms, 3), "steps_per_s": round(1e3 / ms, 3),
                      "steps": steps, "dtype": ("f16 activation storage + f16 MFMA (fp32 accumulate) in the V-Net; fp32 elsewhere" if cfg.get("act") == "f16" else
                                                {"f32": "f32", "f32x3": "f32 (split-bf16 matrix-core mode, fp32-accurate)"}.get(cfg["mma"], "f32 storage, f16/bf16 MFMA operands")),
                      "flags": "trainer defaults" + (" + " + " ".join(cfg["extra"]) if cfg.get("extra") else "") + ("" if cfg["kind"] == "2d" else " (--eqv_pass 1)"),
                      "peak_mem_gb": round(mem, 2), "loss_terms": terms,
                      "wide_1x1_backward_calls_per_step": {"on_nonzero_rows": sb["sparse"], "dense": sb["dense"]},
                      "roofline": roof, "whole_step": whole}))


def sub_record(name, steps):
    try:
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--sub", name, "--sub_steps", str(steps)],
                             capture_output=True, text=True, timeout=600)
        for ln in reversed(out.stdout.splitlines()):
            if ln.startswith("{") and '"sub"' in ln:
                return json.loads(ln)
        return {"sub": name, "error": (out.stderr or out.stdout)[-400:]}
    except Exception as e:                                       # a sub-record must never take the headline down
        return {"sub": name, "error": repr(e)}


def forced_dist_record(steps=20):
    """The headline step with ARCO_FORCE_DIST=1 (a one-rank `nccl` group: every exchange of arco_amd/dist.py issued through RCCL
    and ProcessGroupNCCL's stream inside the two-stream, graph-replayed step): the per-step cost of the collective plumbing at
    N = 1.  No scaling claim - a one-GPU box cannot make one."""
    try:
        env = dict(os.environ, ARCO_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "--steps", str(steps), "--warmup", "5", "--no_subs", "--no_cpu_baseline",
                              "--k2_0_steps", "0", "--sustain_s", "0"], env=env, capture_output=True, text=True, timeout=600)
        for ln in reversed(out.stdout.splitlines()):
            if ln.startswith("{") and '"metric"' in ln:
                r = json.loads(ln)
                return {"sub": "forced_dist_world1", "ms_per_step": r["ms_per_step"], "steps_per_s": r["value"], "steps": steps,
                        "note": "ARCO_FORCE_DIST=1: one-rank nccl group, all of dist.py's collectives issued every step (two gradient "
                                "buckets, counter all-gather, tail-key broadcast, prototype all-reduce, percentile sums)"}
        return {"sub": "forced_dist_world1", "error": (out.stderr or out.stdout)[-400:]}
    except Exception as e:
        return {"sub": "forced_dist_world1", "error": repr(e)}


def dropin_user_record(steps=6):
    """A reference-style USER of the drop-in boundary at the headline size, in a child process: tests/dropin_user.py (this
    repository's own form of the reference trainer's sequence of calls; `dropin/` first on sys.path binds arco_amd through the
    reference's own import statements) with everything such a trainer builds itself in plain torch - nn.Conv2d q_representation,
    torch.optim.SGD, CPU-side entropy percentiles and bank bookkeeping, no graphs / plans / row-sparse head.  What the boundary's
    user gets per step, as opposed to ArcoStep2D (the headline) and to ArcoStep2D in the dense dataflow (`dropin_dense_dataflow`)."""
    try:
        env = {k: v for k, v in os.environ.items() if k not in ("PYTHONPATH", "RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "LOCAL_RANK")}
        script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "dropin_user.py")
        out = subprocess.run([sys.executable, script, "time", str(steps)], env=env, capture_output=True, text=True, timeout=600)
        for ln in reversed(out.stdout.splitlines()):
            if ln.startswith("DROPIN_USER_TIME "):
                r = json.loads(ln[len("DROPIN_USER_TIME "):])
                return {"sub": "dropin_user_step", "ms_per_step": r["ms_per_step"], "steps_per_s": round(1e3 / r["ms_per_step"], 3), "steps": steps,
                        "peak_mem_gb": r["peak_mem_gb"], "loss_terms": {k: round(v, 5) for k, v in r["last"].items() if k.startswith(("loss", "reco", "unsup"))},
                        "workload": "the headline workload (BASELINE.json configs[1]) driven by a reference-style user of dropin/: dense "
                                    "496-channel maps, torch nn.Conv2d q_representation and torch.optim.SGD, three student passes, "
                                    "CPU-side percentiles, eager"}
        return {"sub": "dropin_user_step", "error": (out.stderr or out.stdout)[-400:]}
    except Exception as e:
        return {"sub": "dropin_user_step", "error": repr(e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch_size", type=int, default=8)
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--no_subs", action="store_true", help="skip the configs[2..4] sub-records")
    ap.add_argument("--dense_head", type=int, default=0)
    ap.add_argument("--graphs", type=int, default=1)
    ap.add_argument("--graph_train", type=int, default=-1, help="-1: the trainer's default")
    ap.add_argument("--batched_passes", type=int, default=1)
    ap.add_argument("--k2", type=float, default=1.0, help="weight of the equivariance term (reference default 1.0)")
    ap.add_argument("--k2_0_steps", type=int, default=10, help="extra steps timed with k2 = 0, the north-star path alone (0: skip)")
    ap.add_argument("--sustain_s", type=float, default=3.0, help="length of the extra sustained run (0: skip)")
    ap.add_argument("--dense_teacher", type=int, default=0)
    ap.add_argument("--batch_transform", type=int, default=1, help="the reference's batch_transform (PIL round trip, jitter, blur, AdvMorph); 0: off")
    ap.add_argument("--conv_mma", type=str, default="f32x3", help="matrix-core mode: f32x3 (default, split-bf16, fp32-accurate) or f32 (native fp32 MFMA)")
    ap.add_argument("--head_levels", type=int, default=-1, help="row-sparse head depth of the trainer (default: the trainer's)")
    ap.add_argument("--settle_s", type=float, default=2.5, help="untimed steps for this many seconds before the warmup (clock settling)")
    ap.add_argument("--cpu_baseline_child", action="store_true")
    ap.add_argument("--sub", type=str, default="")
    ap.add_argument("--sub_steps", type=int, default=6)
    a = ap.parse_args()
    if a.cpu_baseline_child:
        return cpu_baseline_child()
    if a.sub:
        return run_sub(a.sub, a.sub_steps)
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a, sys.argv[1:]))

    from arco_amd import dist as adist
    from arco_amd import ops
    from arco_amd import train_arco_2d as T
    rank, world = adist.init()
    if world != a.gpus:
        print(f"bench.py: launched with world size {world} but --gpus {a.gpus}", file=sys.stderr)
        sys.exit(2)
    dev = torch.device("cuda", adist.local_rank())
    torch.cuda.set_device(dev)
    import random
    import numpy as np
    random.seed(1337); np.random.seed(1337); torch.manual_seed(1337)

    flags = ["--batch_size", str(a.batch_size), "--queue_size", "4096", "--func", "smc", "--synthetic", "1",
             "--dense_head", str(a.dense_head), "--graphs", str(a.graphs), "--batched_passes", str(a.batched_passes),
             "--k2", str(a.k2), "--dense_teacher", str(a.dense_teacher), "--batch_transform", str(a.batch_transform),
             "--conv_mma", a.conv_mma]
    if a.graph_train >= 0:
        flags += ["--graph_train", str(a.graph_train)]
    if a.head_levels > 0:
        flags += ["--head_levels", str(a.head_levels)]
    args = T.build_parser().parse_args(flags)
    stepper = T.ArcoStep2D(args, dev)
    if world > 1:             # per-rank generator streams (cutmix boxes, sampler indices, TPS warps), after the weight broadcast
        adist.seed_data_pipeline(1337)
    b = a.batch_size
    batches = []
    for i in range(4):        # a few resident synthetic batches, cycled
        l_img, l_lab = T.synthetic_batch(b, args.patch_size, args.num_classes, 100 + 2 * i * world + rank, dev)
        u_img, _ = T.synthetic_batch(b, args.patch_size, args.num_classes, 101 + 2 * i * world + rank, dev)
        batches.append((l_img, l_lab, u_img))
    cursor = [0]

    def run(n):
        for _ in range(n):
            l_img, l_lab, u_img = batches[cursor[0] % len(batches)]
            cursor[0] += 1
            stepper.step(l_img, l_lab, u_img, 0, 100)

    # one-off costs - HIP-graph capture on a pass's third call, first use of the 50 %-probability augmentation branches
    # (AdvMorph, blur), pinned staging buffers, allocator growth - must lie before the timed region whatever --warmup is:
    # extra untimed steps only (measured: isolated 60-90 ms steps as late as step 11, tools/step_times.py)
    run(max(0, 14 - a.warmup))
    # ... and so must the power-management transient: on these boxes the step runs at 16.4 ms for the first ~0.7 s of load,
    # 18-19 ms for the next ~0.7 s, and settles at the sustained rate from ~1.6 s on (tools/step_times.py, N=500:
    # gpurun_out/steptimes_r2h.log).  Untimed steps until the device has been busy for a_settle seconds, so the
    # K timed steps measure the settled clocks whatever K and W are.
    t_pre = time.perf_counter()
    while a.settle_s > 0 and time.perf_counter() - t_pre < a.settle_s:
        run(10)
        torch.cuda.synchronize()
    run(a.warmup)
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(a.steps)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    # contrastive-loss ms/step (the metric's second half): HIP events around the three loss segments, in a separate pass
    # after the timed region.  The student passes run eagerly in this pass: an event recorded right behind a HIP-graph
    # replay is timestamped unreliably on this stack (the kernel trace of the same step shows the segment's kernels back to
    # back: profiles/README.md), so the graph-replayed passes next to the segments are launched as plain kernels here.
    from arco_amd import graphs as _graphs
    # (the instrumented passes below run on ONE stream: HIP events on the launch stream cannot price a segment or a kernel while
    #  another pass runs beside it on the side stream - T.TEACHER_SIDE is the step's pass-concurrency switch)
    side_mode, T.TEACHER_SIDE = T.TEACHER_SIDE, 0
    prev_flags = _graphs.set_enabled(stepper, {"s_train_tps": False, "s_train_lu": False})
    stepper.profile_loss = True
    stepper.loss_events = []
    run(8)
    torch.cuda.synchronize()
    evs = stepper.loss_events[2:]
    loss_seg = [round(sum(e[i][0].elapsed_time(e[i][1]) for e in evs) / max(1, len(evs)), 3) for i in range(3)]
    if evs and len(evs[0]) > 3 and evs[0][3] is not None:       # row lists + prototypes: queued ahead of the host sync since round 2
        loss_seg[1] = round(loss_seg[1] + sum(e[3][0].elapsed_time(e[3][1]) for e in evs) / len(evs), 3)
    loss_ms = sum(loss_seg)
    stepper.profile_loss = False
    stepper.loss_events = []
    _graphs.set_enabled(stepper, prev_flags)
    T.TEACHER_SIDE = side_mode
    ranks_seen = 1
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        one = torch.ones(1, dtype=torch.float64, device=dev)       # every rank contributes 1 over RCCL: the record proves the collective saw N ranks
        torch.distributed.all_reduce(one, op=torch.distributed.ReduceOp.SUM)
        ranks_seen = int(round(float(one.item())))
    step_ms = dt / a.steps * 1e3

    sustained = k2_0 = roof = whole = step_roof = None
    if world == 1:
        if a.sustain_s > 0:              # clocks settle lower under sustained load: a >= 3 s figure beside the K-step one
            n_s = max(a.steps, int(a.sustain_s * 1e3 / step_ms) + 1)
            ms_s = timed(run, n_s)
            sustained = {"steps": n_s, "ms_per_step": round(ms_s, 3), "steps_per_s": round(1e3 / ms_s, 3),
                         "seconds": round(n_s * ms_s / 1e3, 2)}
        if a.k2_0_steps > 0 and a.k2 != 0:
            stepper.args.k2 = 0.0
            run(4)
            ms0 = timed(run, a.k2_0_steps)
            stepper.args.k2 = a.k2
            k2_0 = {"ms_per_step": round(ms0, 3), "steps_per_s": round(1e3 / ms0, 3), "steps": a.k2_0_steps,
                    "note": "--k2 0: the north-star path alone (contrastive + supervised + unsupervised terms), no "
                            "equivariance pass; round 1's headline configuration"}
    # the roofline objects at every N: each rank runs the same instrumented eager steps (they contain the step's collectives, so
    # all ranks take part); rank 0's launches are the ones reported
    T.TEACHER_SIDE = 0            # single-stream eager pass: per-kernel HIP-event timing (see above)
    prof = eager_profile(stepper, run, 3)
    T.TEACHER_SIDE = side_mode
    roof, whole = roofline_from_profile(prof, 3, step_ms)
    step_roof = step_roofline(prof.get("__work__"), step_ms, split=a.conv_mma == "f32x3")

    if rank == 0:
        out = {
            "metric": "train steps/sec, ACDC 2D 256x256 bs=16", "value": round(world * a.steps / dt, 4),
            "unit": "steps/s (16-image steps, all GPUs)", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(step_ms, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            # the short scalars first (a truncated copy of this line keeps them): contrastive-loss ms/step is the metric's other half
            "contrastive_loss_ms_per_step": round(loss_ms, 3),
            "sustained_ms_per_step": (sustained or {}).get("ms_per_step"),
            "sustained_steps_per_s": (sustained or {}).get("steps_per_s"),
            "whole_step_tflops": (whole or {}).get("tflops_over_whole_step"),
            "k2_0_ms_per_step": (k2_0 or {}).get("ms_per_step"),
            "config": {"workload": f"ACDC 2D 256x256 bs=16 per GPU on {world}xMI355X, stratified sampler + 4096-key/class queue "
                                   "(BASELINE.json configs[1]), reference-default flags",
                       "batch_size_per_stream": b, "images_per_step_per_gpu": 2 * b,
                       "classes": 4, "rep_dim": 496, "num_queries": 256, "num_negatives": 512, "func": "smc", "apply_aug": args.apply_aug,
                       "loss_terms": f"k1*contrastive + k3*unsupervised + CE + Dice + k2*equivariance (k2 = {a.k2:g})",
                       "graph_train": int(bool(getattr(args, "graph_train", 0))), "parallelism": f"dp{world}",
                       "pass_concurrency": f"ARCO_TEACHER_SIDE={T.TEACHER_SIDE} (teacher / statistics / warped student passes on a second stream), ARCO_SIDE_SYNC={T.SIDE_SYNC} (round 4's host-side wait in front of backward; 0 = off since the round-5 kernel fix)",
                       "contrastive_loss_ms_per_step": round(loss_ms, 3),
                       "sustained_ms_per_step": (sustained or {}).get("ms_per_step"),
                       "whole_step_tflops": (whole or {}).get("tflops_over_whole_step")},
            # compute_contra_memobank_loss on the GPU clock: masks | lists, prototypes, keys, banks | anchors, row-sparse head,
            # InfoNCE INCLUDING its analytic gradient w.r.t. the anchors (the loss's backward is computed in the forward)
            "contrastive_loss_segments_ms": {"masks_counts": loss_seg[0], "lists_prototypes_keys_banks": loss_seg[1],
                                             "anchors_head_infonce": loss_seg[2]},
            "roofline": roof,
            "step_roofline": step_roof,
            "ranks_seen": ranks_seen,
        }
        if whole is not None:
            out["whole_step"] = whole
        if sustained is not None:
            out["sustained"] = sustained
        if k2_0 is not None:
            out["north_star_path_only_k2_0"] = k2_0
        if world == 1 and not a.no_subs:
            del stepper
            batches.clear()
            torch.cuda.empty_cache()
            out["configs"] = [sub_record(n, a.sub_steps) for n in SUBS] + [forced_dist_record(), dropin_user_record()]
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()

"""Repeat tests/test_configs_at_size_gpu.py::test_cfg2_graph_replay_equals_eager_at_full_size N times in one process and report
every failing trial with the per-parameter deviations (ARCO_TEST_DIAG lines)."""
import os, sys, io, contextlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
os.environ["ARCO_TEST_DIAG"] = "1"
import torch
import test_configs_at_size_gpu as TC
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bad = 0
for t in range(n):
    buf = io.StringIO()
    t0 = time.time()
    try:
        with contextlib.redirect_stdout(buf):
            TC.test_cfg2_graph_replay_equals_eager_at_full_size()
        print(f"trial {t}: ok ({time.time() - t0:.1f} s)", flush=True)
    except AssertionError as e:
        bad += 1
        print(f"trial {t}: FAILED\n{buf.getvalue()}\n{str(e)[:600]}", flush=True)
    torch.cuda.empty_cache()
print(f"{bad} of {n} trials failed")

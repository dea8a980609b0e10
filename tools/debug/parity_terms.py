"""Run one variant of tests/test_step_parity_gpu.py::test_two_steps_vs_cpu_oracle with every comparison printed instead of asserted.
python tools/debug/parity_terms.py <variant index>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
real = np.testing.assert_allclose
def loud(a, b, rtol=1e-7, atol=0, err_msg="", **kw):
    a_, b_ = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    d = np.abs(a_ - b_)
    rel = float((d / np.maximum(np.abs(b_), 1e-30)).max()) if d.size else 0.0
    bad = bool((d > atol + rtol * np.abs(b_)).any())
    if a_.size <= 4 or bad:
        print(f"{'FAIL' if bad else 'ok  '} {err_msg or '-':28s} n={a_.size} max abs {float(d.max()) if d.size else 0:.3e} max rel {rel:.3e} (rtol {rtol}, atol {atol})"
              + (f"  got {a_.ravel()[:3]} want {b_.ravel()[:3]}" if a_.size <= 4 else ""), flush=True)
np.testing.assert_allclose = loud
import test_step_parity_gpu as T
import pytest
idx = int(sys.argv[1]) if len(sys.argv) > 1 else 6
variants = [m.args[1] for m in T.test_two_steps_vs_cpu_oracle.pytestmark if m.name == "parametrize"][0]
T.test_two_steps_vs_cpu_oracle.__wrapped__(variants[idx]) if hasattr(T.test_two_steps_vs_cpu_oracle, "__wrapped__") else T.test_two_steps_vs_cpu_oracle(variants[idx])

"""Debug aid: how close do the whole-step parity tests come to their tolerances?  Wraps numpy's assert_allclose and records,
per call, max |actual - desired| / (atol + rtol |desired|)  (1.0 = at the bound)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
real = np.testing.assert_allclose
worst = {}
def spy(actual, desired, rtol=1e-7, atol=0, err_msg="", **kw):
    a, d = np.asarray(actual, dtype=np.float64), np.asarray(desired, dtype=np.float64)
    if a.shape == d.shape and a.size:
        r = float(np.max(np.abs(a - d) / (atol + rtol * np.abs(d) + 1e-300)))
        key = (cur[0], str(err_msg).split(" ")[-1] if err_msg else f"shape{tuple(a.shape)}")
        worst[key] = max(worst.get(key, 0.0), r)
    return real(actual, desired, rtol=rtol, atol=atol, err_msg=err_msg, **kw)
np.testing.assert_allclose = spy
cur = [""]
import test_configs_at_size_gpu as TC, test_step_parity_gpu as TS, test_step3d_parity_gpu as T3
cur[0] = "cfg4_oracle"; TC.test_cfg4_cityscapes_shape_vs_cpu_oracle()
for v in T3.VARIANTS:
    cur[0] = "step3d_" + v["tag"]; T3.test_three_steps_3d_vs_cpu_oracle(v)
for k, r in sorted(worst.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{r:8.3f} of the bound  {k[0]:28s} {k[1]}")

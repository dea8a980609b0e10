"""Differential trace of one variant of test_two_steps_vs_cpu_oracle: checksums of every module output of the student / teacher
U-Nets and heads, and of the augmentation outputs, in call order -> a text file; run twice (e.g. with two ARCO_LIB builds) and diff.
python tools/debug/parity_trace.py <variant> <out file>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np, torch
np.testing.assert_allclose = lambda *a, **k: None
import test_step_parity_gpu as T
from arco_amd import train_arco_2d as T2, augment
out = open(sys.argv[2], "w")
seq = [0]
def cs(tag, t):
    if isinstance(t, (tuple, list)):
        for i, x in enumerate(t):
            cs(f"{tag}[{i}]", x)
        return
    if not isinstance(t, torch.Tensor) or not t.is_floating_point():
        if isinstance(t, torch.Tensor):
            out.write(f"{seq[0]:05d} {tag} shape {tuple(t.shape)} isum {int(t.long().sum())}\n"); seq[0] += 1
        return
    d = t.detach().double()
    out.write(f"{seq[0]:05d} {tag} shape {tuple(t.shape)} sum {float(d.sum()):.10e} abs {float(d.abs().sum()):.10e} sq {float((d * d).sum()):.10e}\n"); seq[0] += 1
real_init = T2.ArcoStep2D.__init__
def init(self, *a, **k):
    real_init(self, *a, **k)
    for nm, net in (("student", self.model), ("teacher", self.ema_model), ("qfe", self.q_feature_extractor), ("kfe", self.k_feature_extractor)):
        for mn, m in net.named_modules():
            m.register_forward_hook(lambda mod, inp, o, tag=f"{nm}.{mn}": cs(tag, o))
T2.ArcoStep2D.__init__ = init
for fn in ("batch_transform", "generate_unsup_data", "jitter_blur"):
    if hasattr(augment, fn):
        real = getattr(augment, fn)
        def wrap(*a, _real=real, _fn=fn, **k):
            r = _real(*a, **k); cs("augment." + _fn, r); return r
        setattr(augment, fn, wrap)
variants = [m.args[1] for m in T.test_two_steps_vs_cpu_oracle.pytestmark if m.name == "parametrize"][0]
try:
    T.test_two_steps_vs_cpu_oracle(variants[int(sys.argv[1])])
except AssertionError as e:
    print("assert", e)
out.close()
print("trace lines", seq[0])

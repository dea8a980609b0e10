"""Drop-in for the reference's code/utils/ramps.py (the trainers call `ramps.sigmoid_rampup`: train_arco_2d.py:17,123,
train_arco_3d.py:120, pretrain_2D.py:124, pretrain_3D.py:110).  Host scalars; same names, arguments and return types
(utils/ramps.py:18-55), pinned to the reference in tests/golden/g16_boundary.npz."""
import math


def sigmoid_rampup(current, rampup_length):
    """exp(-5 (1 - t)^2) with t = clip(current, 0, L) / L; 1.0 when L == 0 (utils/ramps.py:18-26)."""
    if rampup_length == 0:
        return 1.0
    t = min(max(float(current), 0.0), float(rampup_length)) / rampup_length
    return float(math.exp(-5.0 * (1.0 - t) * (1.0 - t)))


def linear_rampup(current, rampup_length):
    """utils/ramps.py:29-35."""
    assert current >= 0 and rampup_length >= 0
    return 1.0 if current >= rampup_length else current / rampup_length


def cosine_rampdown(current, rampdown_length):
    """utils/ramps.py:38-41."""
    assert 0 <= current <= rampdown_length
    return float(0.5 * (math.cos(math.pi * current / rampdown_length) + 1))


def exp_rampup(rampup_length):
    """utils/ramps.py:44-55: the sigmoid ramp as a closure over the ramp length, 1.0 from rampup_length on."""
    def wrapper(epoch):
        return sigmoid_rampup(epoch, rampup_length) if epoch < rampup_length else 1.0
    return wrapper

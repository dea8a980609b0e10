"""Volume variant of RandTPS - drop-in for the reference's code/tps/rand_tps_3d.py `RandTPS` (:82-166): the SAME
random 2-D similarity + thin-plate-spline warp, applied to every slice x[..., z] of a [B,C,X,Y,Z] volume (:155-165).
`depth` is part of the reference's signature and unused there as well."""
from .rand_tps import RandTPS as _RandTPS2D


class RandTPS(_RandTPS2D):
    def __init__(self, width, height, depth, batch_size=16, sigma=0.01, border_padding=False, random_mirror=True,
                 random_scale=(0.7, 1.1), mode='affine', device=None):
        super().__init__(width, height, batch_size=batch_size, sigma=sigma, border_padding=border_padding,
                         random_mirror=random_mirror, random_scale=random_scale, mode=mode, device=device)
        self.depth = depth

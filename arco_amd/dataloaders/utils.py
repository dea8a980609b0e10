"""`from dataloaders import utils` (pretrain_2D.py:23, pretrain_3D.py:23): the stage-1 trainers import the module and read
nothing from it, so the import has to succeed and that is all.  Of the reference file's helpers (colour tables of other
datasets, plotting, report writers: out of scope) only the schedule helper is restated."""


def lr_poly(base_lr, iter_, max_iter=100, power=0.9):
    """dataloaders/utils.py:141-142."""
    return base_lr * ((1 - float(iter_) / max_iter) ** power)

"""Drop-in for the reference's code/loss_helper_3d.py - the 4-D (B,C,H,W) = 2-D image
version of the contrastive loss that train_arco_2d.py imports (train_arco_2d.py:24).
Same public names and signatures (loss_helper_3d.py:12,35,83,120,187,271)."""
from ._contrast import compute_contra_memobank_loss, dequeue_and_enqueue
from .samplers import (as_monte_carlo_sample, grid_as_monte_carlo_sample, grid_monte_carlo_sample,
                       monte_carlo_sample)

__all__ = ["compute_contra_memobank_loss", "dequeue_and_enqueue", "grid_monte_carlo_sample",
           "grid_as_monte_carlo_sample", "monte_carlo_sample", "as_monte_carlo_sample"]

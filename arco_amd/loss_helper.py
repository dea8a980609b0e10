"""Drop-in for the reference's code/loss_helper.py - the 5-D (B,C,H,W,D) = 3-D volume
version of the contrastive loss that train_arco_3d.py imports (train_arco_3d.py:22).
Same public names and signatures (loss_helper.py:142,165,213,250,317,442); the device
implementation is rank-generic and shared with loss_helper_3d."""
from ._contrast import compute_contra_memobank_loss, dequeue_and_enqueue
from .samplers import (as_monte_carlo_sample, grid_as_monte_carlo_sample, grid_monte_carlo_sample,
                       monte_carlo_sample)

__all__ = ["compute_contra_memobank_loss", "dequeue_and_enqueue", "grid_monte_carlo_sample",
           "grid_as_monte_carlo_sample", "monte_carlo_sample", "as_monte_carlo_sample"]

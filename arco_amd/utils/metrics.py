"""Segmentation metrics behind the evaluation entry points - the names of the reference's
code/utils/metrics.py (`cal_dice` :5-17, `calculate_metric_percase` :20-26).

The reference delegates to medpy.metric.binary (medpy==0.4.0, environment.yml:301), which is not installed here
and is itself host-side scipy code.  `binary` below restates its published algorithms on scipy.ndimage:
  dc   = 2|A & B| / (|A| + |B|)                (0.0 when both are empty)
  jc   = |A & B| / |A | B|                     (0.0 when both are empty)
  surface distances(A -> B): border(X) = X ^ binary_erosion(X, cross structuring element); distances of A's border
         voxels to the nearest border voxel of B through distance_transform_edt(~border(B), voxelspacing)
  hd95 = 95th percentile of the distances A->B and B->A pooled;  asd = mean of the distances A->B
This is post-processing of a finished label map on the host, exactly where the reference runs it; the overlap
counts of the batched 2-D path come from the GPU (test_2D.overlap_counts)."""
import numpy as np
from scipy.ndimage import binary_erosion, distance_transform_edt, generate_binary_structure


class binary:
    """medpy.metric.binary subset used by the reference (dc, jc, hd95, asd)."""

    @staticmethod
    def dc(result, reference):
        a, b = np.atleast_1d(np.asarray(result).astype(bool)), np.atleast_1d(np.asarray(reference).astype(bool))
        inter = np.count_nonzero(a & b)
        sa, sb = np.count_nonzero(a), np.count_nonzero(b)
        return 2.0 * inter / float(sa + sb) if sa + sb > 0 else 0.0

    @staticmethod
    def jc(result, reference):
        a, b = np.atleast_1d(np.asarray(result).astype(bool)), np.atleast_1d(np.asarray(reference).astype(bool))
        union = np.count_nonzero(a | b)
        return float(np.count_nonzero(a & b)) / float(union) if union > 0 else 0.0

    @staticmethod
    def _surface_distances(result, reference, voxelspacing=None, connectivity=1):
        a, b = np.atleast_1d(np.asarray(result).astype(bool)), np.atleast_1d(np.asarray(reference).astype(bool))
        if not a.any():
            raise RuntimeError('The first supplied array does not contain any binary object.')
        if not b.any():
            raise RuntimeError('The second supplied array does not contain any binary object.')
        footprint = generate_binary_structure(a.ndim, connectivity)
        a_border = a ^ binary_erosion(a, structure=footprint, iterations=1)
        b_border = b ^ binary_erosion(b, structure=footprint, iterations=1)
        dt = distance_transform_edt(~b_border, sampling=voxelspacing)
        return dt[a_border]

    @staticmethod
    def hd95(result, reference, voxelspacing=None, connectivity=1):
        d1 = binary._surface_distances(result, reference, voxelspacing, connectivity)
        d2 = binary._surface_distances(reference, result, voxelspacing, connectivity)
        return float(np.percentile(np.hstack((d1, d2)), 95))

    @staticmethod
    def asd(result, reference, voxelspacing=None, connectivity=1):
        return float(binary._surface_distances(result, reference, voxelspacing, connectivity).mean())


def cal_dice(prediction, label, num=2):
    """Dice of every class 1..num-1 (utils/metrics.py:5-17)."""
    total_dice = np.zeros(num - 1)
    for i in range(1, num):
        p, l = (prediction == i).astype(np.float64), (label == i).astype(np.float64)
        total_dice[i - 1] += 2 * np.sum(p * l) / (np.sum(p) + np.sum(l))
    return total_dice


def calculate_metric_percase(pred, gt):
    """(dice, jaccard, hd95, asd) of two binary masks (utils/metrics.py:20-26, test_util.py:214-220)."""
    return binary.dc(pred, gt), binary.jc(pred, gt), binary.hd95(pred, gt), binary.asd(pred, gt)

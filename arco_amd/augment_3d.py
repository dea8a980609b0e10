"""Drop-in for the names train_arco_3d.py takes from the reference's code/augment_3d.py through `from augment_3d import *`
(train_arco_3d.py:21,263-277,270): `batch_transform`, `generate_unsup_data_3d` (+ the mask helpers).

`batch_transform` (augment_3d.py:209-225) applies `transform` (:133-160) to every volume - whose colour-jitter / blur body is
commented out in the reference, i.e. the identity with or without augmentation - and stacks the results: the inputs come back
unchanged (data, label, logits), on the GPU, with no generator draw."""
from .augment import generate_class_mask, generate_cutout_mask_3d, generate_unsup_data_3d  # noqa: F401


def transform(image, label, logits=None, crop_size=(256, 256), scale_size=(0.8, 1.0), augmentation=True):
    """augment_3d.py:133-160: the identity (every augmentation of the reference body is commented out)."""
    return (image, label, logits) if logits is not None else (image, label)


def batch_transform(data, label, logits, scale_size, apply_augmentation):
    """augment_3d.py:209-225."""
    return data, label, logits

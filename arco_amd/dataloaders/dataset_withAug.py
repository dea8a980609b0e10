"""The two batch-level augmentations of stage-1 pre-training (code/dataloaders/dataset_withAug.py:323-370), applied by
pretrain_2D.py's `transform_student` to a collated batch {'image': [B,1,H,W], 'label': ...}:

  RandomColorJitter(color=(b, c, s, h), p): with probability p (numpy gate) every image goes through torchvision's
      ColorJitter in TENSOR mode - one-channel float images, so saturation and hue leave the image unchanged and
      brightness / contrast are clamped blends (functional_tensor.adjust_brightness / adjust_contrast).  The images are
      modified IN PLACE, as in the reference (the student batch, which shares the tensor, sees the jitter too).
  RandomNoise(p): with probability p every image is blurred as an 8-bit PIL image - ToPILImage (x*255 truncated to a
      byte), ImageFilter.GaussianBlur(radius = random.uniform(0.15, 1.15)), /255 - and a NEW tensor is returned.

Generator consumption follows the reference: numpy for the gates, the torch CPU generator for ColorJitter.get_params
(randperm(4), then one uniform per factor), python's `random` for the blur radius.  The blur runs on the GPU with
Pillow's integer arithmetic (csrc/augment.hip, pinned against Pillow in tests/test_augment_gpu.py); torchvision is not
installed in the build image, so the tensor-mode jitter is restated from torchvision 0.13's published source
(parity-unpinned, like the PIL-mode glue of arco_amd.augment)."""
import random

import numpy as np
import torch


def _jitter_params(color):
    """ColorJitter.get_params (torchvision/transforms/transforms.py): order, then brightness / contrast / saturation / hue
    factors, all drawn from the torch CPU generator."""
    fn_idx = torch.randperm(4)
    rng = [(max(0.0, 1.0 - color[0]), 1.0 + color[0]), (max(0.0, 1.0 - color[1]), 1.0 + color[1]),
           (max(0.0, 1.0 - color[2]), 1.0 + color[2]), (-color[3], color[3])]
    fac = [float(torch.empty(1).uniform_(lo, hi)) for lo, hi in rng]
    return [int(i) for i in fn_idx], fac


def jitter_gray_(img, order, fac):
    """ColorJitter.forward on a one-channel float image [1, H, W], in place."""
    for fn in order:
        if fn == 0:                                    # adjust_brightness: blend with zeros, clamp to [0, 1]
            img.mul_(fac[0]).clamp_(0.0, 1.0)
        elif fn == 1:                                  # adjust_contrast: blend with the mean of the (gray) image
            mean = img.mean()
            img.mul_(fac[1]).add_((1.0 - fac[1]) * mean).clamp_(0.0, 1.0)
        # fn == 2 / 3: adjust_saturation / adjust_hue return a one-channel image unchanged
    return img


class RandomColorJitter(object):
    def __init__(self, color=(0.4, 0.4, 0.4, 0.1), p=0.1):
        self.color, self.p = color, p

    def __call__(self, sample):
        if np.random.uniform(low=0, high=1, size=1) > self.p:
            return sample
        image, label = sample['image'], sample['label']
        for j in range(image.shape[0]):
            order, fac = _jitter_params(self.color)
            jitter_gray_(image[j], order, fac)
        return {'image': image, 'label': label}


class RandomNoise(object):
    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, sample):
        if np.random.uniform(low=0, high=1, size=1) > self.p:
            return sample
        from .. import augment
        image, label = sample['image'], sample['label']
        sigma = random.uniform(0.15, 1.15)
        params = [dict(order=None, factors=None, sigma=sigma)] * int(image.shape[0])
        blurred = augment.jitter_blur(image[:, 0:1].float(), params)          # [B, 1, H, W] -> the reference's [B, H, W]
        return {'image': blurred[:, 0], 'label': label}

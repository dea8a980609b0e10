"""2-D slice ingest with the reference's names: the per-sample transforms of
code/dataloaders/dataset.py (`random_rot_flip` :147-155, `random_rotate` :157-161, `random_crop` :163-178,
`RandomGenerator` :180-200) and the two-stream batch sampler (`TwoStreamBatchSampler` :456-482, `iterate_once`,
`iterate_eternally`, `grouper` :486-505).  Host-side numpy / scipy on single slices, as in the reference; every
random choice is drawn from the same generator (numpy's global RandomState, python's `random`) in the same order."""
import itertools
import random

import numpy as np
import torch
from scipy import ndimage
from scipy.ndimage import zoom
from torch.utils.data import Dataset
from torch.utils.data.sampler import Sampler

from ._io import read_case, read_list


def random_rot_flip(image, label):
    """k quarter turns (k ~ randint(0, 4)) then a flip along axis ~ randint(0, 2), image and label alike."""
    k = np.random.randint(0, 4)
    axis = np.random.randint(0, 2)
    return tuple(np.flip(np.rot90(a, k), axis=axis).copy() for a in (image, label))


def random_rotate(image, label):
    """Nearest-neighbour rotation by an integer angle in [-20, 20) without reshaping."""
    angle = np.random.randint(-20, 20)
    return tuple(ndimage.rotate(a, angle, order=0, reshape=False) for a in (image, label))


def random_crop(image, label, output_size=(256, 256)):
    """Despite the name a CENTRE crop to 256 x 256 (dataset.py:163-178), zero-padding by (missing // 2 + 3) per side
    first when a side is not larger than the target."""
    oh, ow = output_size
    if label.shape[0] <= oh or label.shape[1] <= ow:
        pw, ph = max((oh - label.shape[0]) // 2 + 3, 0), max((ow - label.shape[1]) // 2 + 3, 0)
        image, label = (np.pad(a, [(pw, pw), (ph, ph)], mode='constant', constant_values=0) for a in (image, label))
    w, h = image.shape
    w1, h1 = int(round((w - oh) / 2.)), int(round((h - ow) / 2.))
    return image[w1:w1 + oh, h1:h1 + ow], label[w1:w1 + oh, h1:h1 + ow]


class RandomGenerator(object):
    """zoom(order=0) of the slice to `output_size`, then ONE of rot-flip / rotate / centre-crop chosen by a cascade of
    `random.random() > 0.5` draws (dataset.py:180-200); returns image float32 [1, H, W], label uint8 [H, W]."""

    def __init__(self, output_size):
        self.output_size = output_size

    def __call__(self, sample):
        image, label = sample['image'], sample['label']
        x, y = image.shape
        factors = (self.output_size[0] / x, self.output_size[1] / y)
        image, label = zoom(image, factors, order=0), zoom(label, factors, order=0)
        for op in (random_rot_flip, random_rotate, random_crop):
            if random.random() > 0.5:
                image, label = op(image, label)
                break
        return {'image': torch.from_numpy(image.astype(np.float32)).unsqueeze(0),
                'label': torch.from_numpy(label.astype(np.uint8))}


def iterate_once(iterable):
    return np.random.permutation(iterable)


def iterate_eternally(indices):
    def shuffles():
        while True:
            yield np.random.permutation(indices)
    return itertools.chain.from_iterable(shuffles())


def grouper(iterable, n):
    """grouper('ABCDEFG', 3) -> ABC DEF (incomplete tail dropped)."""
    return zip(*([iter(iterable)] * n))


class TwoStreamBatchSampler(Sampler):
    """Batches of (batch_size - secondary_batch_size) primary indices followed by secondary_batch_size secondary
    indices; one epoch = one pass over a permutation of the primary indices, the secondary stream reshuffles forever."""

    def __init__(self, primary_indices, secondary_indices, batch_size, secondary_batch_size):
        self.primary_indices, self.secondary_indices = primary_indices, secondary_indices
        self.secondary_batch_size = secondary_batch_size
        self.primary_batch_size = batch_size - secondary_batch_size
        assert len(self.primary_indices) >= self.primary_batch_size > 0
        assert len(self.secondary_indices) >= self.secondary_batch_size > 0

    def __iter__(self):
        primary = grouper(iterate_once(self.primary_indices), self.primary_batch_size)
        secondary = grouper(iterate_eternally(self.secondary_indices), self.secondary_batch_size)
        return (p + s for p, s in zip(primary, secondary))

    def __len__(self):
        return len(self.primary_indices) // self.primary_batch_size


class BaseDataSets(Dataset):
    """code/dataloaders/dataset.py:97-144 (train_arco_2d.py:20, pretrain_2D.py:24): the un-indexed ACDC / MM slice
    dataset - every slice of `train_slices.list` (`train_slices.txt` for MM) under `<base_dir>/data/slices/`, validation
    volumes of `val.list` (`test_vol.txt`) under `<base_dir>/data/`; `num` keeps the first `num` training slices."""

    def __init__(self, base_dir=None, split='train', num=None, transform=None):
        self._base_dir, self.split, self.transform = base_dir, split, transform
        self.sample_list = []
        acdc = 'ACDC' in base_dir
        if split == 'train' and (acdc or 'MM' in base_dir):
            self.sample_list = read_list(base_dir + ('/train_slices.list' if acdc else '/train_slices.txt'), strip='' if acdc else '.h5')
        elif split == 'val' and (acdc or 'MM' in base_dir):
            self.sample_list = read_list(base_dir + ('/val.list' if acdc else '/test_vol.txt'))
        if num is not None and split == "train":
            self.sample_list = self.sample_list[:num]
        print("total {} samples".format(len(self.sample_list)))

    def __len__(self):
        return len(self.sample_list)

    def __getitem__(self, idx):
        case = self.sample_list[idx]
        image, label = read_case(self._base_dir + ("/data/slices/" if self.split == "train" else "/data/") + case)
        sample = {'image': image, 'label': label}
        if self.split == "train" and self.transform is not None:
            sample = self.transform(sample)
        sample["idx"] = idx
        return sample

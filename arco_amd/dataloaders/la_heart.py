"""3-D volume ingest with the reference's names (code/dataloaders/la_heart.py): `LAHeartWithIndex`
(:46-82), `RandomCrop` (:113-146), `RandomRotFlip` (:149-166), `ToTensor` (:193-203).  Same numpy draws, same order."""
import numpy as np
import torch
from torch.utils.data import Dataset

from ._io import read_case, read_list


class LAHeartWithIndex(Dataset):
    """Left-atrium volumes `<base_dir>/<case>/mri_norm2.{h5,npz}` listed in `<base_dir>/../{train,test}.list`; the
    first `index` cases are the labeled set (label_type=1), the rest the unlabeled set (label_type=0)."""

    def __init__(self, base_dir=None, split='train', num=None, transform=None, index=4, label_type=1):
        self._base_dir, self.transform = base_dir, transform
        cases = read_list(f"{base_dir}/../{'train' if split == 'train' else 'test'}.list")
        cases = cases[:index] if label_type == 1 else cases[index:]
        self.image_list = cases[:num] if num is not None else cases
        print("total {} samples".format(len(self.image_list)))

    def __len__(self):
        return len(self.image_list)

    def __getitem__(self, idx):
        image, label = read_case(f"{self._base_dir}/{self.image_list[idx]}/mri_norm2")
        sample = {'image': image, 'label': label}
        if self.transform:
            sample = self.transform(sample)
        sample['idx'] = idx
        return sample


class RandomCrop(object):
    """Random `output_size` crop of a volume; axes not larger than the target are first zero-padded by
    (missing // 2 + 3) per side (all three, as soon as one is too small)."""

    def __init__(self, output_size):
        self.output_size = output_size

    def __call__(self, sample):
        image, label = sample['image'], sample['label']
        o = self.output_size
        if any(label.shape[a] <= o[a] for a in range(3)):
            pads = [(max((o[a] - label.shape[a]) // 2 + 3, 0),) * 2 for a in range(3)]
            image, label = (np.pad(v, pads, mode='constant', constant_values=0) for v in (image, label))
        w, h, d = image.shape
        w1 = np.random.randint(0, w - o[0])
        h1 = np.random.randint(0, h - o[1])
        d1 = np.random.randint(0, d - o[2])
        box = (slice(w1, w1 + o[0]), slice(h1, h1 + o[1]), slice(d1, d1 + o[2]))
        return {'image': image[box], 'label': label[box]}


class RandomRotFlip(object):
    """k quarter turns in the first two axes, then a flip along axis ~ randint(0, 2)."""

    def __call__(self, sample):
        k = np.random.randint(0, 4)
        axis = np.random.randint(0, 2)
        return {key: np.flip(np.rot90(sample[key], k), axis=axis).copy() for key in ('image', 'label')}


class ToTensor(object):
    """image -> float32 [1, X, Y, Z], label -> int64."""

    def __call__(self, sample):
        image = sample['image']
        out = {'image': torch.from_numpy(image.reshape(1, *image.shape).astype(np.float32)),
               'label': torch.from_numpy(np.ascontiguousarray(sample['label'])).long()}
        if 'onehot_label' in sample:
            out['onehot_label'] = torch.from_numpy(sample['onehot_label']).long()
        return out

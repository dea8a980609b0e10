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


class LAHeart(LAHeartWithIndex):
    """code/dataloaders/la_heart.py:14-43 (pretrain_3D.py:28): every case of the list, no labeled / unlabeled split."""

    def __init__(self, base_dir=None, split='train', num=None, transform=None):
        self._base_dir, self.transform = base_dir, transform
        cases = read_list(f"{base_dir}/../{'train' if split == 'train' else 'test'}.list")
        self.image_list = cases[:num] if num is not None else cases
        print("total {} samples".format(len(self.image_list)))


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


class RandomColorJitter(object):
    """code/dataloaders/la_heart.py:254-273: with probability p every X-Y slice image[j, :, :, :, t] of every volume goes
    through torchvision's tensor-mode ColorJitter (one channel: brightness / contrast blends), in place - one set of
    generator draws per slice."""

    def __init__(self, color=(0.04, 0.04, 0.04, 0.01), p=0.1):
        self.color, self.p = color, p

    def __call__(self, sample):
        import numpy as np
        from .dataset_withAug import _jitter_params, jitter_gray_
        if np.random.uniform(low=0, high=1, size=1) > self.p:
            return sample
        image, label = sample['image'], sample['label']
        for j in range(image.shape[0]):
            for t in range(image.shape[-1]):
                order, fac = _jitter_params(self.color)
                jitter_gray_(image[j, :, :, :, t], order, fac)
        return {'image': image, 'label': label}


class RandomNoise(object):
    """code/dataloaders/la_heart.py:276-294: with probability p every X-Y slice is blurred as an 8-bit PIL image, in
    place, and - as in the reference, which drops ToTensor's /255 here - stored back as the BYTE values 0..255."""

    def __init__(self, p=0.5):
        self.p = p

    def __call__(self, sample):
        import random
        import numpy as np
        import torch
        from .. import augment
        if np.random.uniform(low=0, high=1, size=1) > self.p:
            return sample
        image, label = sample['image'], sample['label']
        sigma = random.uniform(0.15, 1.15)
        for i in range(image.shape[0]):
            sl = image[i, 0].permute(2, 0, 1).unsqueeze(1).float().contiguous()          # [Z, 1, X, Y]
            out = augment.jitter_blur(sl, [dict(order=None, factors=None, sigma=sigma)] * int(sl.shape[0]))
            image[i, 0] = torch.round(out[:, 0] * 255.0).permute(1, 2, 0).to(image.dtype)
        return {'image': image, 'label': label}

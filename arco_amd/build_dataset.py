"""Slice datasets of the 2-D trainer - `BaseDataSetsWithIndex` of code/build_dataset.py:18-69:
`<base_dir>/train_slices.list` (ACDC) or `train_slices.txt` (MM) names the training slices under
`<base_dir>/data/slices/<case>.{h5,npz}`; the first `index` entries form the labeled set (label_type=1), the rest the
unlabeled set; `val.list` names whole volumes under `<base_dir>/data/`."""
import os

import numpy as np
from torch.utils.data.dataset import Dataset

from .dataloaders._io import read_case, read_list


class BaseDataSetsWithIndex(Dataset):
    def __init__(self, base_dir=None, split='train', num=None, transform=None, index=16, label_type=0):
        self._base_dir, self.index, self.split, self.transform = base_dir, index, split, transform
        self.sample_list = []
        if split == 'train' and ('ACDC' in base_dir or 'MM' in base_dir):
            acdc = 'ACDC' in base_dir
            names = read_list(base_dir + ('/train_slices.list' if acdc else '/train_slices.txt'), strip='' if acdc else '.h5')
            self.sample_list = names[:index] if label_type == 1 else names[index:]
        elif split == 'val':
            self.sample_list = read_list(base_dir + '/val.list')
        if num is not None and split == 'train':
            self.sample_list = self.sample_list[:num - index]
        print("total {} samples".format(len(self.sample_list)))

    def __len__(self):
        return len(self.sample_list)

    def __getitem__(self, idx):
        case = self.sample_list[idx]
        sub = "/data/slices/" if self.split == "train" else "/data/"
        image, label = read_case(self._base_dir + sub + case)
        sample = {'image': image, 'label': label}
        if self.split == "train" and self.transform is not None:
            sample = self.transform(sample)
        sample["idx"] = idx
        return sample


class Synapse_datasetWithIndex(Dataset):
    """npz slice datasets of the Synapse / LiTS / JHU experiments (code/build_dataset.py:159-200): `<list_dir>/<split>.txt`
    (`<split>_40.txt` for LiTS, `<split>_vol.txt` for test / val volumes) names the cases, training slices are
    `<base_dir>/<name>.npz` with 'image' / 'label'; the first `index` entries are the labeled set (label_type=1)."""

    def __init__(self, base_dir, list_dir, split, transform=None, index=221, label_type=1):
        self.transform, self.split, self.data_dir, self.index, self.label_type = transform, split, base_dir, index, label_type
        if 'Lits' in list_dir:
            name = split + '_40.txt'
        elif split in ("test", "val"):
            name = split + '_vol.txt'
        else:
            name = split + '.txt'
        names = read_list(os.path.join(list_dir, name))
        self.sample_list = names[:index] if label_type == 1 else names[index:]

    def __len__(self):
        return len(self.sample_list)

    def __getitem__(self, idx):
        name = self.sample_list[idx]
        if self.split == "train":
            with np.load(os.path.join(self.data_dir, name + '.npz')) as data:
                image, label = data['image'], data['label']
        else:
            image, label = read_case(self.data_dir + "/{}.npy".format(name))        # <name>.npy.h5 volumes
        sample = {'image': image, 'label': label}
        if self.transform:
            sample = self.transform(sample)
        sample['case_name'] = name
        return sample


class Synapse_dataset(Dataset):
    """code/build_dataset.py:127-157 (imported by train_arco_2d.py:21): the un-indexed npz slice dataset - `<split>_40.txt`
    training slices `<base_dir>/<name>.npz`, `<split>_vol_40.txt` validation volumes `<base_dir>/<name>.npy.{h5,npz}`."""

    def __init__(self, base_dir, list_dir, split, transform=None):
        self.transform, self.split, self.data_dir = transform, split, base_dir
        self.sample_list = read_list(os.path.join(list_dir, split + ('_vol_40.txt' if split in ('test', 'val') else '_40.txt')))

    def __len__(self):
        return len(self.sample_list)

    def __getitem__(self, idx):
        name = self.sample_list[idx]
        if self.split == "train":
            with np.load(os.path.join(self.data_dir, name + '.npz')) as data:
                image, label = data['image'], data['label']
        else:
            image, label = read_case(self.data_dir + "/{}.npy".format(name))
        sample = {'image': image, 'label': label}
        if self.transform:
            sample = self.transform(sample)
        sample['case_name'] = name
        return sample

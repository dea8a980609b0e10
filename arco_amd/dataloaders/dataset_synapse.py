"""`Synapse_dataset` of the reference's code/dataloaders/dataset_synapse.py:68-104 (pretrain_2D.py:27): the npz slice dataset
without the labeled / unlabeled index - `<list_dir>/<split>.txt` (`<split>_40.txt` for LiTS, `<split>_vol.txt` for test / val
volumes), training slices `<base_dir>/<name>.npz`, volumes `<base_dir>_40/<name>.{h5,npz}`; `num` keeps the first `num`."""
import os

import numpy as np
from torch.utils.data import Dataset

from ._io import read_case, read_list


class Synapse_dataset(Dataset):
    def __init__(self, base_dir, list_dir, split, num=None, transform=None):
        self.transform, self.split, self.data_dir = transform, split, base_dir
        if 'Lits' in list_dir:
            name = split + '_40.txt'
        elif split in ("test", "val"):
            name = split + '_vol.txt'
        else:
            name = split + '.txt'
        self.sample_list = read_list(os.path.join(list_dir, name))
        if num is not None and split == "train":
            self.sample_list = self.sample_list[:num]
        print("total {} samples".format(len(self.sample_list)))

    def __len__(self):
        return len(self.sample_list)

    def __getitem__(self, idx):
        name = self.sample_list[idx]
        if self.split == "train":
            with np.load(os.path.join(self.data_dir, name + '.npz')) as data:
                image, label = data['image'], data['label']
        else:
            image, label = read_case(self.data_dir + "_40/{}".format(name))
        sample = {'image': image, 'label': label}
        if self.transform:
            sample = self.transform(sample)
        sample['case_name'] = name
        return sample

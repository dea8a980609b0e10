"""Data ingest: arco_amd's per-sample transforms, two-stream sampler and datasets vs outputs of
the reference's own classes (tests/golden/g8_ingest.npz, oracle/gen_golden.py g8) - bit-exact arrays and the same
consumption of the numpy / python generators.  Host-side code: runs without a GPU."""
import os
import random

import numpy as np
import pytest
import torch

import fixture_inputs as fx

G8 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g8_ingest.npz"))


def _seed(s):
    random.seed(s); np.random.seed(s); torch.manual_seed(s)


@pytest.mark.parametrize("seed", fx.INGEST_SEEDS)
def test_random_generator_matches_reference(seed):
    from arco_amd.dataloaders.dataset import RandomGenerator
    img, lab = fx.ingest_slice(seed)
    _seed(seed)
    r = RandomGenerator([32, 40])({'image': img, 'label': lab})
    assert r['image'].dtype == torch.float32 and r['label'].dtype == torch.uint8     # (a quarter turn swaps H and W)
    np.testing.assert_array_equal(r['image'].numpy(), G8[f"gen{seed}_image"])
    np.testing.assert_array_equal(r['label'].numpy(), G8[f"gen{seed}_label"])
    np.testing.assert_array_equal(np.array((float(np.random.uniform()), random.random())), G8[f"gen{seed}_probe"])


def test_random_generator_cases_cover_every_branch():
    branches = set()
    for seed in fx.INGEST_SEEDS:
        random.seed(seed)
        draws = [random.random() for _ in range(3)]
        branches.add(next((i for i, d in enumerate(draws) if d > 0.5), 3))
    assert branches == {0, 1, 2, 3}, branches          # rot-flip, rotate, crop, none


@pytest.mark.parametrize("seed", fx.INGEST_SEEDS)
def test_volume_transforms_match_reference(seed):
    from arco_amd.dataloaders.la_heart import RandomCrop, RandomRotFlip, ToTensor
    vol, vlab = fx.ingest_volume(seed)
    _seed(seed)
    r = ToTensor()(RandomRotFlip()(RandomCrop((16, 12, 16))({'image': vol, 'label': vlab})))
    assert r['label'].dtype == torch.int64
    np.testing.assert_array_equal(r['image'].numpy(), G8[f"vol{seed}_image"])
    np.testing.assert_array_equal(r['label'].numpy(), G8[f"vol{seed}_label"].astype(np.int64))
    np.testing.assert_array_equal(np.array((float(np.random.uniform()), random.random())), G8[f"vol{seed}_probe"])


@pytest.mark.parametrize("name", ["random_rot_flip", "random_rotate", "random_crop"])
def test_slice_ops_match_reference(name):
    from arco_amd.dataloaders import dataset
    img, lab = fx.ingest_slice(99, (40, 36))
    _seed(3)
    a, b = getattr(dataset, name)(img, lab)
    np.testing.assert_array_equal(a, G8[f"{name}_image"])
    np.testing.assert_array_equal(b, G8[f"{name}_label"])


def test_two_stream_sampler_matches_reference():
    from arco_amd.dataloaders.dataset import TwoStreamBatchSampler
    _seed(11)
    smp = TwoStreamBatchSampler(list(range(7)), list(range(7, 30)), 6, 4)
    got = np.array([list(b) for _ in range(3) for b in smp], dtype=np.int64)
    np.testing.assert_array_equal(got, G8["two_stream"])
    assert len(smp) == int(G8["two_stream_len"]) == 3
    assert all(set(b[:2]) <= set(range(7)) and set(b[2:]) <= set(range(7, 30)) for b in got)


def _fake_acdc(root, n_slices=10, shape=(40, 36)):
    os.makedirs(os.path.join(root, "data", "slices"))
    names = [f"patient{i:03d}_frame01_slice_{i}" for i in range(n_slices)]
    for i, n in enumerate(names):
        img, lab = fx.ingest_slice(50 + i, shape)
        np.savez(os.path.join(root, "data", "slices", n + ".npz"), image=img, label=lab)
    with open(os.path.join(root, "train_slices.list"), "w") as f:
        f.write("\n".join(names) + "\n")
    vol = np.stack([fx.ingest_slice(80 + i, shape)[0] for i in range(3)]); lab = np.stack([fx.ingest_slice(80 + i, shape)[1] for i in range(3)])
    np.savez(os.path.join(root, "data", "patient100_frame01.npz"), image=vol, label=lab)
    with open(os.path.join(root, "val.list"), "w") as f:
        f.write("patient100_frame01\n")
    return names


def test_base_dataset_splits_and_reads(tmp_path):
    from arco_amd.build_dataset import BaseDataSetsWithIndex
    from arco_amd.dataloaders.dataset import RandomGenerator
    root = str(tmp_path / "ACDC")
    names = _fake_acdc(root)
    lab_set = BaseDataSetsWithIndex(base_dir=root, split="train", transform=RandomGenerator([32, 32]), index=4, label_type=1)
    unl_set = BaseDataSetsWithIndex(base_dir=root, split="train", transform=RandomGenerator([32, 32]), index=4, label_type=0)
    assert lab_set.sample_list == names[:4] and unl_set.sample_list == names[4:]
    _seed(1)
    s = lab_set[2]
    assert s["idx"] == 2 and tuple(s["image"].shape) == (1, 32, 32) and tuple(s["label"].shape) == (32, 32)
    val = BaseDataSetsWithIndex(base_dir=root, split="val")
    v = val[0]
    assert v["image"].shape == (3, 40, 36) and v["label"].shape == (3, 40, 36)      # whole volume, no transform
    from arco_amd.dataloaders._io import read_case
    with pytest.raises(FileNotFoundError):
        read_case(os.path.join(root, "data", "slices", "missing_case"))


def test_la_dataset_reads(tmp_path):
    from arco_amd.dataloaders.la_heart import LAHeartWithIndex, RandomCrop, ToTensor
    from arco_amd.dataloaders import Compose
    base = tmp_path / "LA" / "2018LA_Seg_Training Set"
    cases = [f"case{i}" for i in range(5)]
    for i, c in enumerate(cases):
        os.makedirs(base / c)
        vol, lab = fx.ingest_volume(i)
        np.savez(base / c / "mri_norm2.npz", image=vol, label=lab)
    with open(tmp_path / "LA" / "train.list", "w") as f:
        f.write("\n".join(cases) + "\n")
    ds_l = LAHeartWithIndex(base_dir=str(base), split="train", transform=Compose([RandomCrop((16, 16, 12)), ToTensor()]), index=2, label_type=1)
    ds_u = LAHeartWithIndex(base_dir=str(base), split="train", transform=Compose([RandomCrop((16, 16, 12)), ToTensor()]), index=2, label_type=0)
    assert len(ds_l) == 2 and len(ds_u) == 3
    _seed(0)
    s = ds_u[1]
    assert tuple(s["image"].shape) == (1, 16, 16, 12) and s["label"].dtype == torch.int64 and s["idx"] == 1


def test_npz_slice_dataset(tmp_path):
    """Synapse / LiTS / JHU style npz datasets (build_dataset.py:159-200): list naming rules and the labeled split."""
    from arco_amd.build_dataset import Synapse_datasetWithIndex
    from arco_amd.dataloaders import Compose
    from arco_amd.dataloaders.dataset import RandomGenerator
    data, lists = tmp_path / "train_npz_40", tmp_path / "Lits"
    os.makedirs(data); os.makedirs(lists)
    names = [f"case{i:04d}_slice{i:03d}" for i in range(9)]
    for i, n in enumerate(names):
        img, lab = fx.ingest_slice(400 + i, (30, 34), n_cls=3)
        np.savez(data / (n + ".npz"), image=img, label=lab)
    (lists / "train_40.txt").write_text("\n".join(names) + "\n")
    lab_set = Synapse_datasetWithIndex(base_dir=str(data), list_dir=str(lists), split="train", index=3, label_type=1,
                                       transform=Compose([RandomGenerator([32, 32])]))
    unl_set = Synapse_datasetWithIndex(base_dir=str(data), list_dir=str(lists), split="train", index=3, label_type=0)
    assert lab_set.sample_list == names[:3] and unl_set.sample_list == names[3:] and len(unl_set) == 6
    _seed(2)
    s = lab_set[1]
    assert s["case_name"] == names[1] and s["image"].dtype == torch.float32 and s["label"].dtype == torch.uint8
    raw = unl_set[0]
    np.testing.assert_array_equal(raw["image"], fx.ingest_slice(403, (30, 34), n_cls=3)[0])
    other = tmp_path / "lists_Synapse"
    os.makedirs(other)
    (other / "train.txt").write_text("\n".join(names[:4]) + "\n")
    assert len(Synapse_datasetWithIndex(base_dir=str(data), list_dir=str(other), split="train", index=1, label_type=0)) == 3

"""The trainers' real-data path (--synthetic 0): datasets on disk (npz twins of the reference's h5
files) -> reference-named datasets / transforms / loaders -> the HIP training step (-m gpu)."""
import os
import random

import numpy as np
import pytest
import torch

import fixture_inputs as fx

pytestmark = pytest.mark.gpu


def _seed(s):
    random.seed(s); np.random.seed(s); torch.manual_seed(s)


def test_2d_trainer_on_slices_from_disk(tmp_path, monkeypatch):
    from arco_amd import train_arco_2d as T
    root = tmp_path / "ACDC"
    os.makedirs(root / "data" / "slices")
    names = [f"patient{i:03d}_slice_{i}" for i in range(30)]          # patients_to_slices('ACDC', 1) = 23 labeled slices
    for i, n in enumerate(names):
        img, lab = fx.ingest_slice(300 + i, (40, 36))
        np.savez(root / "data" / "slices" / (n + ".npz"), image=img, label=lab)
    (root / "train_slices.list").write_text("\n".join(names) + "\n")
    monkeypatch.chdir(tmp_path)
    args = T.build_parser().parse_args(["--synthetic", "0", "--root_path", str(root), "--exp", "ACDC/ingest_test", "--labeled_num", "1",
                                        "--batch_size", "2", "--queue_size", "256", "--num_queries", "32", "--num_negatives", "16",
                                        "--max_iterations", "4"])
    # the default 256 x 256 patch: RandomGenerator's crop branch is hard-wired to 256 x 256 in the reference
    _seed(1)
    l_loader, u_loader = T.build_loaders(args)
    assert len(u_loader) == (30 - 23) // 2 and len(l_loader.dataset) >= len(u_loader.dataset)
    batch = next(iter(l_loader))
    assert tuple(batch["image"].shape) == (2, 1, 256, 256) and tuple(batch["label"].shape) == (2, 256, 256)
    snap = tmp_path / "snap"
    os.makedirs(snap)
    assert T.train(args, str(snap)) == "Training Finished!"


def test_3d_trainer_on_volumes_from_disk(tmp_path, monkeypatch):
    from arco_amd import train_arco_3d as T3
    base = tmp_path / "LA" / "2018LA_Seg_Training Set"
    cases = [f"case{i}" for i in range(5)]
    for i, c in enumerate(cases):
        os.makedirs(base / c)
        vol, lab = fx.ingest_volume(i, (40, 38, 36))
        np.savez(base / c / "mri_norm2.npz", image=vol, label=lab)
    (tmp_path / "LA" / "train.list").write_text("\n".join(cases) + "\n")
    monkeypatch.chdir(tmp_path)
    args = T3.build_parser().parse_args(["--synthetic", "0", "--root_path", str(base), "--exp", "LA/ingest_test", "--labeled_num", "2",
                                         "--batch_size", "1", "--num_classes", "2", "--queue_size", "256", "--num_queries", "32",
                                         "--num_negatives", "16", "--max_iterations", "3", "--eqv_pass", "0"])
    args.patch_size = [32, 32, 32]
    _seed(2)
    snap = tmp_path / "snap3"
    os.makedirs(snap)
    assert T3.train(args, str(snap)) == "Training Finished!"

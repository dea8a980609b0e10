"""Reading one case: {'image', 'label'} arrays from `<stem>.h5` (h5py, when importable) or from `<stem>.npz`
(pure numpy, same two keys) - the reference reads h5 only; the npz twin lets the ingest path run and be tested
where h5py is not installed (it is not in this image)."""
import os

import numpy as np

try:
    import h5py as _h5py
except Exception:                                   # pragma: no cover - optional
    _h5py = None


def read_case(stem):
    h5, npz = stem + ".h5", stem + ".npz"
    if _h5py is not None and os.path.exists(h5):
        with _h5py.File(h5, "r") as f:
            return f["image"][:], f["label"][:]
    if os.path.exists(npz):
        with np.load(npz) as f:
            return f["image"], f["label"]
    if os.path.exists(h5):
        raise RuntimeError(f"arco_amd: {h5} needs h5py, which is not importable here (or provide {npz})")
    raise FileNotFoundError(stem + ".{h5,npz}")


def read_list(path, strip=""):
    with open(path, "r") as f:
        return [line.replace(strip, "").strip() for line in f if line.strip()]

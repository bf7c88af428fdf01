"""Datasets and coalition samplers (reference src/datasets.py).

Coalition bookkeeping is a bit-exact contract: every sampler consumes
``np.random.RandomState(seed)`` in the same order as the reference
(remove_data_by_shapley :631-697, _datamodel :582-628, _uniform :559-579,
_class :525-556, _loo :700, _for_aoi :710, removed_by_classes :720-742) and is
checked against index lists produced by the reference itself
(tests/golden/samplers.json).

Real CIFAR/CelebA files cannot be fetched offline (the reference passes
``download=True``, :427-456), so ``create_dataset`` serves seeded synthetic
stand-ins with the reference's names, cardinalities and label structure unless
``GAD_DATA=real`` points it at local arrays (see ``ArrayDataset.from_npz``).
"""
import os
from typing import Sequence, Tuple

import numpy as np
import torch

import src.constants as constants


# ----------------------------------------------------------------------------
# datasets
# ----------------------------------------------------------------------------
class ArrayDataset(torch.utils.data.Dataset):
    """Images held as one uint8 array [N,H,W,C] (CIFAR python-pickle layout) + int labels.
    ``__getitem__`` reproduces the reference transform chain for the CIFAR family
    (RandomHorizontalFlip -> ToTensor -> Normalize(.5,.5): datasets.py:444-451)."""

    def __init__(self, data: np.ndarray, targets: Sequence[int], train: bool = True, flip: bool = True):
        assert data.dtype == np.uint8 and data.ndim == 4
        self.data = data
        self.targets = [int(t) for t in targets]
        self.train = train
        self.flip = flip

    def __len__(self):
        return len(self.targets)

    def __getitem__(self, i):
        x = torch.from_numpy(self.data[i]).permute(2, 0, 1).float().div_(255.0)
        if self.flip and torch.rand(1).item() < 0.5:   # torchvision RandomHorizontalFlip draws torch.rand(1)
            x = x.flip(-1)
        return x.sub_(0.5).div_(0.5), self.targets[i]

    @property
    def labels(self):
        return self.targets

    def device_tensor(self, device, idx=None):
        """Whole (sub)set as one normalised NCHW fp32 tensor resident in HBM -
        10 000 CIFAR images are 123 MB, so the GPU loader keeps them on the card."""
        d = self.data if idx is None else self.data[np.asarray(idx)]
        x = torch.from_numpy(d).to(device).permute(0, 3, 1, 2).float()
        return x.div_(127.5).sub_(1.0)

    @classmethod
    def from_npz(cls, path, **kw):
        z = np.load(path)
        return cls(z["data"], z["targets"].tolist(), **kw)


def synthetic_images(n, size, channels, n_cls, per_class_seed=1234):
    """Seeded stand-in images: each class gets its own low-frequency colour field plus
    noise so that class membership changes the data distribution (and therefore the
    model-behaviour score) in a learnable way."""
    rng = np.random.RandomState(per_class_seed)
    yy, xx = np.meshgrid(np.linspace(-1, 1, size), np.linspace(-1, 1, size), indexing="ij")
    per = n // n_cls
    out = np.empty((n, size, size, channels), dtype=np.uint8)
    labels = []
    for c in range(n_cls):
        freq = rng.uniform(0.5, 3.0, size=(channels, 2))
        phase = rng.uniform(0, 2 * np.pi, size=channels)
        base = np.stack([np.sin(freq[k, 0] * np.pi * xx + freq[k, 1] * np.pi * yy + phase[k])
                         for k in range(channels)], axis=-1)
        lo, hi = c * per, (c + 1) * per if c < n_cls - 1 else n
        img = 0.6 * base[None] + 0.25 * rng.standard_normal((hi - lo, size, size, channels))
        out[lo:hi] = np.clip((img * 0.5 + 0.5) * 255.0 + 0.5, 0, 255).astype(np.uint8)
        labels += [c] * (hi - lo)
    return out, labels


_SYNTH_SPECS = {
    # name: (n_train, n_test, size, channels, n_classes)  -- cardinalities of the reference datasets
    "cifar": (50000, 10000, 32, 3, 10),
    "cifar2": (10000, 2000, 32, 3, 2),          # CIFAR2 :22-56 (automobile, horse)
    "cifar100": (10000, 2000, 32, 3, 20),       # "CIFAR-20": 20 CIFAR-100 classes x 500 (:59-118)
    "cifar100_f": (10000, 2000, 32, 3, 20),
    "mnist": (60000, 10000, 28, 1, 10),
    "toy2": (128, 32, 32, 3, 2),                # BASELINE config 1: 2 contributor groups x 64 images
}


def create_dataset(dataset_name: str, train: bool, dataset_dir: str = None) -> torch.utils.data.Dataset:
    """Same signature as the reference (:398-513)."""
    dataset_dir = dataset_dir or constants.DATASET_DIR
    real = os.path.join(dataset_dir, dataset_name, "train.npz" if train else "test.npz")
    if os.environ.get("GAD_DATA", "synthetic") == "real":
        if not os.path.exists(real):
            raise FileNotFoundError(f"GAD_DATA=real but {real} is missing")
        return ArrayDataset.from_npz(real, train=train)
    if dataset_name not in _SYNTH_SPECS:
        raise ValueError(f"dataset_name={dataset_name} should be one of {sorted(_SYNTH_SPECS)} "
                         "(celeba / imagenette need local files: GAD_DATA=real)")
    n_tr, n_te, size, ch, n_cls = _SYNTH_SPECS[dataset_name]
    scale = float(os.environ.get("GAD_SYNTH_SCALE", "1"))
    n = int((n_tr if train else n_te) * scale)
    n -= n % n_cls
    data, labels = synthetic_images(n, size, ch, n_cls, per_class_seed=1234 if train else 4321)
    return ArrayDataset(data, labels, train=train)


class TensorDataset(torch.utils.data.Dataset):
    """Wraps a stacked image tensor (reference :375-395)."""

    def __init__(self, data, transform=None, label=None):
        self.data, self.transform, self.label = data, transform, label

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        x = self.data[idx]
        if self.transform:
            x = self.transform(x)
        return x if self.label is None else (x, self.label[idx])


# ----------------------------------------------------------------------------
# coalition samplers
# ----------------------------------------------------------------------------
def _labels_of(dataset) -> list:
    """Label column without materialising images when the dataset exposes it."""
    for attr in ("targets", "labels"):
        t = getattr(dataset, attr, None)
        if t is not None and len(t) == len(dataset):
            return list(t)
    return [item[1] for item in dataset]


def _shapley_size(rng, n):
    """Draw |S| in 1..n-1 with P(|S|=s) ∝ (n-1)/(s(n-s)) - the Shapley kernel's size marginal."""
    sizes = np.arange(1, n)
    p = (n - 1) / (sizes * (n - sizes))
    p /= p.sum()
    return rng.choice(sizes, size=1, p=p)[0]


def remove_data_by_shapley(dataset, seed: int = 0, by_class: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    rng = np.random.RandomState(seed)
    if by_class:
        labels = _labels_of(dataset)
        classes = np.unique(labels)
        keep = _shapley_size(rng, len(classes))
        perm = np.arange(len(classes))
        rng.shuffle(perm)
        dropped = classes[perm[keep:]]
        removed = np.array([i for i, l in enumerate(labels) if l in dropped])
        return np.setdiff1d(np.arange(len(labels)), removed), removed
    n = len(dataset)
    keep = _shapley_size(rng, n)
    perm = np.arange(n)
    rng.shuffle(perm)
    return perm[:keep], perm[keep:]


def remove_data_by_datamodel(dataset, alpha: float = 0.5, seed: int = 0, by_class: bool = False):
    rng = np.random.RandomState(seed)
    if by_class:
        labels = _labels_of(dataset)
        classes = np.unique(labels).tolist()
        keep = int(alpha * len(classes))
        rng.shuffle(classes)
        kept = classes[:keep]
        remaining = np.array([i for i, l in enumerate(labels) if l in kept])
        return remaining, np.setdiff1d(np.arange(len(labels)), remaining)
    n = len(dataset)
    perm = np.arange(n)
    keep = int(alpha * n)
    rng.shuffle(perm)
    return perm[:keep], perm[keep:]


def remove_data_by_uniform(dataset, seed: int = 0):
    """NB no ``by_class`` parameter: the reference entry points pass one and fail with
    TypeError (main.py:268-270 vs datasets.py:559); kept as is."""
    rng = np.random.RandomState(seed)
    take = rng.normal(size=len(dataset)) > 0
    idx = np.arange(len(dataset))
    return idx[take], idx[~take]


def remove_data_by_class(dataset, excluded_class: list):
    labels = _labels_of(dataset)
    rank = {l: i for i, l in enumerate(sorted(set(labels)))}
    excluded = [rank[c] for c in excluded_class]
    removed = np.array([i for i, l in enumerate(labels) if rank[l] in excluded])
    return np.setdiff1d(np.arange(len(labels)), removed), removed


def remove_data_by_loo(dataset, loo_idx: int):
    n = len(dataset)
    return np.array([i for i in range(n) if i != loo_idx]), np.array([loo_idx])


def remove_data_for_aoi(dataset, aoi_idx: int):
    n = len(dataset)
    return np.array([aoi_idx]), np.array([i for i in range(n) if i != aoi_idx])


def removed_by_classes(dataset, seed: int = 0):
    """(remaining_classes, removed_classes) of the by-class Shapley draw (:720-742)."""
    rng = np.random.RandomState(seed)
    classes = np.unique(_labels_of(dataset))
    keep = _shapley_size(rng, len(classes))
    perm = np.arange(len(classes))
    rng.shuffle(perm)
    return classes[perm[:keep]], classes[perm[keep:]]

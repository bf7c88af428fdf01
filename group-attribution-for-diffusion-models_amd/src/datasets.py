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


# ---- torchvision-free readers of the on-disk formats the reference's torchvision classes consume ----
CIFAR20_CLASSES = [40, 41, 42, 43, 44, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 80, 81, 82, 83, 84]  # :80-105


def read_cifar_python(root: str, kind: str, train: bool):
    """The "python version" CIFAR archives as torchvision lays them out under `root`
    (cifar-10-batches-py/{data_batch_1..5,test_batch}, key `labels`; cifar-100-python/{train,test}, key
    `fine_labels`): rows are 3072 uint8 in CHW plane order -> ([N,32,32,3] uint8, labels)."""
    import pickle
    if kind == "cifar10":
        base, files, key = "cifar-10-batches-py", ([f"data_batch_{i}" for i in range(1, 6)] if train else ["test_batch"]), "labels"
    elif kind == "cifar100":
        base, files, key = "cifar-100-python", (["train"] if train else ["test"]), "fine_labels"
    else:
        raise ValueError(kind)
    data, labels = [], []
    for f in files:
        path = os.path.join(root, base, f)
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} is missing (the reference downloads it; there is no network here)")
        with open(path, "rb") as fh:
            entry = pickle.load(fh, encoding="latin1")
        data.append(np.asarray(entry["data"], dtype=np.uint8))
        labels += list(entry[key])
    data = np.vstack(data).reshape(-1, 3, 32, 32).transpose(0, 2, 3, 1)
    return np.ascontiguousarray(data), [int(t) for t in labels]


def _keep_classes(data, labels, classes_to_keep):
    """CIFAR2 (:46-56) / CIFAR100_original (:108-118): keep the listed classes, relabel by list position."""
    keep = [i for i, t in enumerate(labels) if t in classes_to_keep]
    return data[keep], [classes_to_keep.index(labels[i]) for i in keep]


def _cifar100_filter(data, labels):
    """CIFAR100_filter.filter_data (:294-309): the first 2*(c+1) samples of class c, labels unchanged."""
    cap = np.arange(1, 101) * 2
    count = np.zeros(100, dtype=int)
    keep = []
    for i, t in enumerate(labels):
        if count[t] < cap[t]:
            keep.append(i)
            count[t] += 1
    return data[keep], [labels[i] for i in keep]


def read_mnist_idx(root: str, train: bool):
    """torchvision's MNIST/raw idx files, resized 28->32 bilinear through PIL as transforms.Resize does (:471-477)."""
    import gzip
    from PIL import Image
    stem = "train" if train else "t10k"

    def load(name):
        for cand in (os.path.join(root, "MNIST", "raw", name), os.path.join(root, "MNIST", "raw", name + ".gz")):
            if os.path.exists(cand):
                with (gzip.open if cand.endswith(".gz") else open)(cand, "rb") as fh:
                    return fh.read()
        raise FileNotFoundError(os.path.join(root, "MNIST", "raw", name))

    img = np.frombuffer(load(f"{stem}-images-idx3-ubyte"), dtype=np.uint8, offset=16).reshape(-1, 28, 28)
    lab = np.frombuffer(load(f"{stem}-labels-idx1-ubyte"), dtype=np.uint8, offset=8)
    out = np.stack([np.asarray(Image.fromarray(x).resize((32, 32), Image.BILINEAR)) for x in img])[..., None]
    return out, [int(t) for t in lab]


class LatentDataset(torch.utils.data.Dataset):
    """CelebA-HQ in `--precompute_stage reuse` mode (main.py:531-546,675-680): rows of labels.csv (columns
    `filename`, `celeb`; :319-325) joined with the VQ-VAE latent dictionary `{filename: [3,64,64] float}` stored
    at {outdir}/{dataset}/precomputed_emb/vqvae_output.pt.  Items are (latent, celeb, filename) like the
    reference's CelebA.__getitem__ (:338-345) with the image already encoded."""

    def __init__(self, labels_csv: str, latent_file: str):
        import pandas as pd
        df = pd.read_csv(labels_csv)
        assert df["filename"].nunique() == len(df), "filename should be unique"
        self.filenames = df["filename"].tolist()
        self.targets = [int(c) for c in df["celeb"].tolist()]
        lat = torch.load(latent_file, map_location="cpu", weights_only=True)
        missing = [f for f in self.filenames if f not in lat]
        if missing:
            raise KeyError(f"{len(missing)} files of {labels_csv} have no latent in {latent_file}, e.g. {missing[:3]}")
        self.latents = torch.stack([lat[f].float() for f in self.filenames])

    def __len__(self):
        return len(self.filenames)

    def __getitem__(self, i):
        return self.latents[i], self.targets[i], self.filenames[i]

    @property
    def labels(self):
        return self.targets

    def device_tensor(self, device, idx=None):
        x = self.latents if idx is None else self.latents[torch.as_tensor(np.asarray(idx), dtype=torch.long)]
        return x.to(device)

    flip = False


def _real_dataset(dataset_name, train, dataset_dir):
    """GAD_DATA=real: the same directories the reference reads (:412-512), no download."""
    if dataset_name == "cifar":
        return ArrayDataset(*read_cifar_python(os.path.join(dataset_dir, "cifar"), "cifar10", train), train=train)
    if dataset_name == "cifar2":
        d, l = read_cifar_python(os.path.join(dataset_dir, "cifar2"), "cifar10", train)
        return ArrayDataset(*_keep_classes(d, l, [1, 7]), train=train)
    if dataset_name == "cifar100":
        d, l = read_cifar_python(os.path.join(dataset_dir, "cifar100"), "cifar100", train)
        return ArrayDataset(*_keep_classes(d, l, CIFAR20_CLASSES), train=train)
    if dataset_name == "cifar100_f":
        d, l = read_cifar_python(os.path.join(dataset_dir, "cifar100"), "cifar100", train)
        return ArrayDataset(*_cifar100_filter(d, l), train=train)
    if dataset_name == "mnist":
        return ArrayDataset(*read_mnist_idx(os.path.join(dataset_dir, "mnist"), train), train=train, flip=False)
    if dataset_name == "celeba":
        root = os.path.join(dataset_dir, "celeba_hq_256_50_resized")
        lat = os.environ.get("GAD_LATENTS", os.path.join(constants.OUTDIR, "celeba", "precomputed_emb", "vqvae_output.pt"))
        return LatentDataset(os.path.join(root, "labels.csv"), lat)
    raise ValueError(f"dataset_name={dataset_name}: no local reader (cifar100_new needs a pretrained resnet18, "
                     "imagenette an image decoder pipeline; both are off the attribution path)")


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
        if os.path.exists(real):
            return ArrayDataset.from_npz(real, train=train)
        return _real_dataset(dataset_name, train, dataset_dir)
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

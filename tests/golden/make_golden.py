"""Generate golden vectors by RUNNING THE REFERENCE'S OWN CODE (build container
only; /root/reference does not exist on the GPU box, tests read the committed
fixtures).  Run:  python tests/golden/make_golden.py

What is imported from /root/reference (read-only, unmodified):
  src/attributions/methods/datashapley.py   data_shapley, kernel_shap
  src/attributions/methods/databanzhaf.py   data_banzhaf
  src/attributions/methods/datamodel.py     datamodel
  src/datasets.py                           remove_data_by_{shapley,datamodel,uniform,loo}, remove_data_for_aoi
  src/ddpm_config.py                        DDPMConfig / PromptConfig / Lora* registries
`src/datasets.py` and `src/ddpm_config.py` import torchvision / lightning-free
helpers and the user-supplied, git-ignored `src/constants.py` at module top;
those names are absent from this image, so empty placeholder modules are
registered for the *unrelated* imports only - none of the functions recorded
here touches them (they are pure numpy).

Outputs (data only - inputs + expected outputs):
  tests/golden/samplers.json, shapley.npz, configs.json, datamodel.npz
"""
import json
import os
import sys
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, k):
        return _Anything()


def main():
    sys.path.insert(0, REF)
    tv = _placeholder("torchvision")
    tv.models = _placeholder("torchvision.models", resnet18=_Anything)
    tv.transforms = _placeholder("torchvision.transforms", PILToTensor=_Anything, Compose=_Anything, ToTensor=_Anything)
    tv.datasets = _placeholder("torchvision.datasets", CIFAR10=object, CIFAR100=object, MNIST=object,
                               ImageFolder=object)
    tv.datasets.folder = _placeholder("torchvision.datasets.folder", default_loader=None)
    _placeholder("src.constants", DATASET_DIR="/tmp/_ds", OUTDIR="/tmp/_out", LOGDIR="/tmp/_log",
                 MAX_NUM_SAMPLE_IMAGES_TO_SAVE=64)
    from src import datasets as rds
    from src.attributions.methods.datashapley import data_shapley, kernel_shap
    from src.attributions.methods.databanzhaf import data_banzhaf
    from src.ddpm_config import DDPMConfig
    import src.ddpm_config as rcfg

    # ---------------- samplers --------------------------------------------
    cases = []

    def fake(n, n_cls, order):
        if order == "block":
            labels = [i // (n // n_cls) for i in range(n)]
        else:
            labels = [i % n_cls for i in range(n)]
        return [(None, l) for l in labels], labels

    for (n, n_cls, order) in [(200, 20, "mod"), (10000, 20, "block"), (60, 2, "mod"), (258, 258, "mod")]:
        ds, labels = fake(n, n_cls, order)
        for seed in list(range(8)) + [42, 123, 1000]:
            for by_class in (True, False):
                rem, rmv = rds.remove_data_by_shapley(ds, seed=seed, by_class=by_class)
                cases.append(dict(fn="shapley", n=n, n_cls=n_cls, order=order, seed=seed, by_class=by_class,
                                  remaining=np.asarray(rem).tolist(), removed=np.asarray(rmv).tolist()))
            for alpha in (0.25, 0.5, 0.75):
                for by_class in (True, False):
                    rem, rmv = rds.remove_data_by_datamodel(ds, alpha=alpha, seed=seed, by_class=by_class)
                    cases.append(dict(fn="datamodel", n=n, n_cls=n_cls, order=order, seed=seed, alpha=alpha,
                                      by_class=by_class, remaining=np.asarray(rem).tolist(),
                                      removed=np.asarray(rmv).tolist()))
            rem, rmv = rds.remove_data_by_uniform(ds, seed=seed)
            cases.append(dict(fn="uniform", n=n, n_cls=n_cls, order=order, seed=seed,
                              remaining=rem.tolist(), removed=rmv.tolist()))
        rem, rmv = rds.remove_data_by_loo(ds, 3)
        cases.append(dict(fn="loo", n=n, n_cls=n_cls, order=order, idx=3, remaining=rem.tolist(),
                          removed=rmv.tolist()))
        rem, rmv = rds.remove_data_for_aoi(ds, 5)
        cases.append(dict(fn="aoi", n=n, n_cls=n_cls, order=order, idx=5, remaining=rem.tolist(),
                          removed=rmv.tolist()))
        if n_cls <= 20:
            rem, rmv = rds.remove_data_by_class(ds, excluded_class=[1])
            cases.append(dict(fn="class", n=n, n_cls=n_cls, order=order, excluded=[1],
                              remaining=rem.tolist(), removed=rmv.tolist()))
        for seed in range(4):
            rc, xc = rds.removed_by_classes(ds, seed=seed)
            cases.append(dict(fn="classes", n=n, n_cls=n_cls, order=order, seed=seed,
                              remaining=np.asarray(rc).tolist(), removed=np.asarray(xc).tolist()))
    # compact: long index lists -> (count, sha256 of the int64 little-endian bytes, first 8) to keep the fixture small
    import hashlib

    def compact(v):
        if isinstance(v, list) and len(v) > 300:
            a = np.asarray(v, dtype="<i8")
            return {"n": int(a.size), "sha256": hashlib.sha256(a.tobytes()).hexdigest(), "head": a[:8].tolist()}
        return v

    for c in cases:
        c["remaining"], c["removed"] = compact(c["remaining"]), compact(c["removed"])
    with open(os.path.join(OUT, "samplers.json"), "w") as f:
        json.dump(cases, f, separators=(",", ":"))

    # ---------------- solvers ----------------------------------------------
    rng = np.random.RandomState(0)
    blobs = {}
    for name, (n, d) in {"a": (64, 20), "b": (1000, 20), "c": (300, 258), "d": (10, 20)}.items():
        X = (rng.rand(n, d) > 0.5).astype(np.float64)
        w = rng.randn(d)
        y = X @ w + 0.05 * rng.randn(n)
        v1, v0 = float(w.sum()), 0.1
        blobs[f"{name}_X"], blobs[f"{name}_y"] = X, y
        blobs[f"{name}_v"] = np.array([v1, v0])
        blobs[f"{name}_shapley"] = data_shapley(d, X, y, v1, v0)
        blobs[f"{name}_banzhaf"] = data_banzhaf(X, y)
        if n >= d:
            blobs[f"{name}_kernelshap"] = kernel_shap(d, X, y, v1, v0)
    np.savez_compressed(os.path.join(OUT, "shapley.npz"), **blobs)

    # ---------------- config registry ---------------------------------------
    reg = {}
    for k, v in vars(DDPMConfig).items():
        if k.endswith("_config") and isinstance(v, dict):
            reg[f"DDPMConfig.{k}"] = v
    for cls_name in ("PromptConfig", "LoraTrainingConfig", "LoraUnlearningConfig", "LoraSparseUnlearningConfig",
                     "TextToImageGenerationConfig", "TextToImageModelBehaviorConfig", "DatasetStats"):
        cls = getattr(rcfg, cls_name, None)
        if cls is None:
            continue
        for k, v in vars(cls).items():
            if not k.startswith("_") and isinstance(v, (dict, list, int, float, str)):
                reg[f"{cls_name}.{k}"] = v
    with open(os.path.join(OUT, "configs.json"), "w") as f:
        json.dump(reg, f, indent=1, sort_keys=True, default=str)

    # ---------------- datamodel (bootstrap RidgeCV, global numpy RNG) -------
    from src.attributions.methods.datamodel import datamodel
    rng = np.random.RandomState(7)
    Xd = (rng.rand(120, 20) > 0.5).astype(np.float64)
    yd = Xd @ rng.randn(20) + 0.1 * rng.randn(120)
    np.random.seed(123)
    coef = datamodel(Xd, yd, 4)
    np.savez_compressed(os.path.join(OUT, "datamodel.npz"), X=Xd, y=yd, seed=np.array(123), coef=coef)
    print("wrote", len(cases), "sampler cases;", len(blobs), "solver arrays;", len(reg), "config entries; datamodel", coef.shape)


if __name__ == "__main__":
    main()

"""CPU tests: host logic vs golden vectors dumped from the reference's own code
(tests/golden/make_golden.py).  Bit-exact for index bookkeeping."""
import hashlib
import json
import os

import numpy as np
import pytest

from src import datasets as ds
from src.attributions.methods.databanzhaf import data_banzhaf
from src.attributions.methods.datashapley import data_shapley, kernel_shap


class _Fake:
    def __init__(self, labels):
        self.targets = labels

    def __len__(self):
        return len(self.targets)

    def __iter__(self):
        return iter([(None, l) for l in self.targets])

    def __getitem__(self, i):
        return None, self.targets[i]


def _fake(n, n_cls, order):
    return _Fake([i // (n // n_cls) for i in range(n)] if order == "block" else [i % n_cls for i in range(n)])


def _same(got, want):
    got = np.asarray(got)
    if isinstance(want, dict):
        a = got.astype("<i8")
        assert a.size == want["n"]
        assert a[:8].tolist() == want["head"]
        assert hashlib.sha256(a.tobytes()).hexdigest() == want["sha256"]
    else:
        assert got.tolist() == want


@pytest.fixture(scope="module")
def sampler_cases(golden_dir):
    return json.load(open(os.path.join(golden_dir, "samplers.json")))


def test_samplers_bit_exact(sampler_cases):
    seen = set()
    for c in sampler_cases:
        d = _fake(c["n"], c["n_cls"], c["order"])
        fn = c["fn"]
        if fn == "shapley":
            r, x = ds.remove_data_by_shapley(d, seed=c["seed"], by_class=c["by_class"])
        elif fn == "datamodel":
            r, x = ds.remove_data_by_datamodel(d, alpha=c["alpha"], seed=c["seed"], by_class=c["by_class"])
        elif fn == "uniform":
            r, x = ds.remove_data_by_uniform(d, seed=c["seed"])
        elif fn == "loo":
            r, x = ds.remove_data_by_loo(d, c["idx"])
        elif fn == "aoi":
            r, x = ds.remove_data_for_aoi(d, c["idx"])
        elif fn == "class":
            r, x = ds.remove_data_by_class(d, c["excluded"])
        elif fn == "classes":
            r, x = ds.removed_by_classes(d, seed=c["seed"])
        else:
            raise AssertionError(fn)
        _same(r, c["remaining"])
        _same(x, c["removed"])
        seen.add(fn)
    assert seen == {"shapley", "datamodel", "uniform", "loo", "aoi", "class", "classes"}


def test_sampler_fallback_iterates_dataset():
    # datasets without .targets are iterated like the reference does
    lab = [i % 5 for i in range(50)]
    a = ds.remove_data_by_shapley(_Fake(lab), seed=3, by_class=True)
    b = ds.remove_data_by_shapley([(None, l) for l in lab], seed=3, by_class=True)
    assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()


def test_uniform_has_no_by_class_kwarg():
    with pytest.raises(TypeError):
        ds.remove_data_by_uniform(_fake(20, 2, "mod"), seed=0, by_class=True)


def test_solvers_match_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "shapley.npz"))
    for name in "abcd":
        X, y, (v1, v0) = z[f"{name}_X"], z[f"{name}_y"], z[f"{name}_v"]
        d = X.shape[1]
        np.testing.assert_allclose(data_shapley(d, X, y, v1, v0), z[f"{name}_shapley"], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(data_banzhaf(X, y), z[f"{name}_banzhaf"], rtol=1e-9, atol=1e-12)
        if f"{name}_kernelshap" in z:
            np.testing.assert_allclose(kernel_shap(d, X, y, v1, v0), z[f"{name}_kernelshap"], rtol=1e-7, atol=1e-9)


def test_shapley_efficiency_and_exact_recovery():
    rng = np.random.RandomState(1)
    X = (rng.rand(200, 20) > 0.5).astype(float)
    w = rng.randn(20)
    coef = data_shapley(20, X, X @ w + 2.0, w.sum() + 2.0, 2.0)
    np.testing.assert_allclose(coef.ravel(), w, atol=1e-8)
    assert abs(coef.sum() - w.sum()) < 1e-8


def test_config_registry_matches_reference(golden_dir):
    import src.ddpm_config as c

    g = json.load(open(os.path.join(golden_dir, "configs.json")))
    assert len(g) == 14
    for k, v in g.items():
        cls, name = k.split(".")
        assert json.loads(json.dumps(getattr(getattr(c, cls), name))) == v, k


def test_synthetic_cifar20_shape():
    os.environ["GAD_SYNTH_SCALE"] = "0.02"
    try:
        d = ds.create_dataset("cifar100", train=True)
    finally:
        del os.environ["GAD_SYNTH_SCALE"]
    assert len(d) == 200 and len(set(d.targets)) == 20
    x, y = d[0]
    assert x.shape == (3, 32, 32) and x.min() >= -1 and x.max() <= 1 and y == 0

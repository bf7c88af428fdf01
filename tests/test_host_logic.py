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


def test_optimizer_state_round_trips_through_torch_adam():
    """ckpt_steps_*.pt `optimizer` entry (reference main.py:827-840) = torch.optim.Adam.state_dict(): the flat
    moments must import from it and export a dict torch.optim.Adam itself accepts, conv weights included
    (flat storage is [Cout,KH,KW,Cin], the state dict is logical [Cout,Cin,KH,KW])."""
    import torch
    from gad.training import _slot, adam_state_from_torch, adam_state_to_torch, flat_views

    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Conv2d(3, 5, 3), torch.nn.Flatten(), torch.nn.Linear(5 * 36, 7))
    params = list(net.parameters())
    opt = torch.optim.Adam(params, lr=1e-3)
    for _ in range(3):
        opt.zero_grad()
        net(torch.randn(2, 3, 8, 8)).square().mean().backward()
        opt.step()
    sd = opt.state_dict()
    total = sum(_slot(p.numel()) for p in params)              # the flat layout's 32-byte slots (flatten_params)
    m, v = torch.full((total,), 9.0), torch.full((total,), 9.0)
    assert adam_state_from_torch(sd, params, m, v) == 3
    for i, (a, b) in enumerate(zip(flat_views(params, m), flat_views(params, v))):
        assert torch.equal(a, sd["state"][i]["exp_avg"]) and torch.equal(b, sd["state"][i]["exp_avg_sq"])
    # conv moments really are stored tap-major / channel-minor in the flat buffer
    assert torch.equal(m[:5 * 27].view(5, 3, 3, 3), sd["state"][0]["exp_avg"].permute(0, 2, 3, 1))
    back = adam_state_to_torch(params, m, v, 3, dict(lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0))
    opt2 = torch.optim.Adam(params, lr=1e-3)
    opt2.load_state_dict(back)                                   # torch accepts the exported dict
    for i in range(len(params)):
        assert torch.equal(opt2.state[params[i]]["exp_avg"], opt.state[params[i]]["exp_avg"])
        assert float(opt2.state[params[i]]["step"]) == 3.0
    # legacy flat layout still loads
    m2, v2 = torch.zeros(total), torch.zeros(total)
    assert adam_state_from_torch({"step": 3, "exp_avg": m, "exp_avg_sq": v}, params, m2, v2) == 3 and torch.equal(m2, m)
    with pytest.raises(ValueError):
        adam_state_from_torch(sd, params[:-1], m, v)


def test_real_data_readers(tmp_path, monkeypatch):
    """torchvision-free readers of the archives the reference's dataset classes consume
    (reference src/datasets.py:22-118,294-309,412-477) on miniature files in the same on-disk format."""
    import pickle
    import torch
    rng = np.random.RandomState(0)
    root = tmp_path / "datasets"
    c100 = root / "cifar100" / "cifar-100-python"
    c100.mkdir(parents=True)
    labels = [int(t) for t in rng.randint(0, 100, size=600)]
    rows = rng.randint(0, 256, size=(600, 3072)).astype(np.uint8)
    with open(c100 / "train", "wb") as f:
        pickle.dump({"data": rows, "fine_labels": labels, "coarse_labels": [0] * 600}, f)
    c10 = root / "cifar2" / "cifar-10-batches-py"
    c10.mkdir(parents=True)
    l10 = [int(t) for t in rng.randint(0, 10, size=50)]
    for i in range(1, 6):
        with open(c10 / f"data_batch_{i}", "wb") as f:
            pickle.dump({"data": rows[10 * (i - 1):10 * i], "labels": l10[10 * (i - 1):10 * i]}, f)
    monkeypatch.setenv("GAD_DATA", "real")
    d = ds.create_dataset("cifar100", train=True, dataset_dir=str(root))
    keep = [i for i, t in enumerate(labels) if t in ds.CIFAR20_CLASSES]
    assert len(d) == len(keep) and d.targets == [ds.CIFAR20_CLASSES.index(labels[i]) for i in keep]
    # CHW planes -> HWC pixels: pixel (y,x) channel c of row r is rows[r, c*1024 + y*32 + x]
    assert d.data[3, 5, 7, 2] == rows[keep[3], 2 * 1024 + 5 * 32 + 7]
    x, y = d[0]
    assert x.shape == (3, 32, 32) and float(x.min()) >= -1 and float(x.max()) <= 1 and y == d.targets[0]
    f = ds.create_dataset("cifar100_f", train=True, dataset_dir=str(root))
    cnt = np.bincount(f.targets, minlength=100)
    full = np.bincount(labels, minlength=100)
    assert (cnt == np.minimum(full, 2 * np.arange(1, 101))).all()
    c2 = ds.create_dataset("cifar2", train=True, dataset_dir=str(root))
    assert c2.targets == [[1, 7].index(t) for t in l10 if t in (1, 7)]
    with pytest.raises(FileNotFoundError):
        ds.create_dataset("cifar", train=True, dataset_dir=str(root))
    # CelebA in precompute "reuse" mode: labels.csv joined with the latent dictionary
    import pandas as pd
    cel = root / "celeba_hq_256_50_resized"
    cel.mkdir()
    names = [f"{i:05d}.jpg" for i in range(6)]
    pd.DataFrame({"filename": names, "celeb": [3, 3, 1, 0, 1, 2]}).to_csv(cel / "labels.csv", index=False)
    lat = {n: torch.full((3, 4, 4), float(i)) for i, n in enumerate(names)}
    torch.save(lat, tmp_path / "vqvae_output.pt")
    monkeypatch.setenv("GAD_LATENTS", str(tmp_path / "vqvae_output.pt"))
    L = ds.create_dataset("celeba", train=True, dataset_dir=str(root))
    assert len(L) == 6 and L.targets == [3, 3, 1, 0, 1, 2] and L[4][2] == "00004.jpg" and float(L[4][0].mean()) == 4.0
    assert L.device_tensor("cpu", [5, 0]).flatten(1).mean(1).tolist() == [5.0, 0.0] and L.flip is False
    remaining, removed = ds.remove_data_by_shapley(L, seed=1, by_class=True)
    assert len(remaining) + len(removed) == 6


def test_datamodel_matches_reference(golden_dir):
    """Bootstrap RidgeCV datamodel (reference src/attributions/methods/datamodel.py:8-36) with the global numpy RNG
    seeded as in the fixture run of the reference function."""
    from src.attributions.methods.datamodel import datamodel
    z = np.load(os.path.join(golden_dir, "datamodel.npz"))
    np.random.seed(int(z["seed"]))
    got = datamodel(z["X"], z["y"], 4)
    assert got.shape == z["coef"].shape == (4, 20)
    assert np.allclose(got, z["coef"], rtol=1e-9, atol=1e-12)


def test_lora_file_keys_follow_unet_save_attn_procs(tmp_path):
    """diffusers-0.24 `unet.save_attn_procs` (train_text_to_image_lora.py:1367,1493; prune_lora.py:102,196)
    writes `<attn>.processor.to_q_lora.down.weight` with no `unet.` prefix; the loader takes that form and the
    pipeline-level `unet.`-prefixed one, and never loads a foreign file as a silent no-op."""
    import pytest
    import torch
    from safetensors.torch import load_file, save_file
    import gad

    cfg = dict(block_out_channels=(32, 64, 64, 64), attention_head_dim=2, cross_attention_dim=48, norm_num_groups=16)
    torch.manual_seed(0)
    net = gad.UNet2DConditionModel(**cfg)
    names = list(net.attention_modules())
    ranks = {f"{n[:-len('.processor')]}.{p}": 3 + (i % 4) for i, n in enumerate(names) for p in ("to_q", "to_k", "to_v", "to_out")}
    net.inject_lora(rank=4, ranks=ranks)                                      # ragged ranks, as after prune_lora.py
    for n, p in net.named_parameters():
        if "lora_layer" in n:
            torch.nn.init.normal_(p)
    net.save_attn_procs(str(tmp_path), weight_name="w.safetensors")
    sd = load_file(str(tmp_path / "w.safetensors"))
    assert len(sd) == 2 * 4 * len(names) and not any(k.startswith("unet.") for k in sd)
    assert "mid_block.attentions.0.transformer_blocks.0.attn1.processor.to_out_lora.up.weight" in sd
    # reference-format file -> fresh model
    other = gad.UNet2DConditionModel(**cfg)
    assert other.load_attn_procs(str(tmp_path), weight_name="w.safetensors") == 4 * len(names)
    for (ka, a), (kb, b) in zip(sorted(net.lora_state_dict().items()), sorted(other.lora_state_dict().items())):
        assert ka == kb and torch.equal(a, b)
    # pipeline-level (`unet.`-prefixed) form loads too
    save_file({"unet." + k: v for k, v in sd.items()}, str(tmp_path / "p.safetensors"))
    third = gad.UNet2DConditionModel(**cfg)
    assert third.load_attn_procs(str(tmp_path), weight_name="p.safetensors") == 4 * len(names)
    # foreign / partly foreign files are errors, not no-ops
    save_file({"text_encoder.foo.weight": torch.zeros(2, 2)}, str(tmp_path / "bad.safetensors"))
    with pytest.raises(KeyError, match="none of the"):
        gad.UNet2DConditionModel(**cfg).load_attn_procs(str(tmp_path), weight_name="bad.safetensors")
    save_file({**sd, "down_blocks.9.attn.processor.to_q_lora.down.weight": torch.zeros(2, 2)}, str(tmp_path / "bad2.safetensors"))
    with pytest.raises(KeyError, match="match no attention projection"):
        gad.UNet2DConditionModel(**cfg).load_attn_procs(str(tmp_path), weight_name="bad2.safetensors")


def test_pad_heads_is_a_zero_padding_of_the_per_head_blocks_and_slices_the_gradient_back():
    """ops.PadHeadsFn (host tensor ops only): the projections of a head-grouped-pruned attention (heads of 23,
    unconditional_generation/prune.py:337-342) are zero-padded per head to 24 so that every launch is float4-aligned."""
    import torch
    from gad import ops
    heads, d, dp, C = 3, 5, 8, 4
    w = torch.nn.Parameter(torch.arange(heads * d * C, dtype=torch.float32).view(heads * d, C))
    b = torch.nn.Parameter(torch.arange(heads * d, dtype=torch.float32))
    wo = torch.nn.Parameter(torch.arange(C * heads * d, dtype=torch.float32).view(C, heads * d))
    wp, bp, wop = ops.pad_heads(w, heads, d, dp, 0), ops.pad_heads(b, heads, d, dp, 0), ops.pad_heads(wo, heads, d, dp, 1)
    assert wp.shape == (heads * dp, C) and bp.shape == (heads * dp,) and wop.shape == (C, heads * dp)
    for h in range(heads):
        assert torch.equal(wp[h * dp:h * dp + d], w[h * d:(h + 1) * d]) and not wp[h * dp + d:(h + 1) * dp].any()
        assert torch.equal(bp[h * dp:h * dp + d], b[h * d:(h + 1) * d]) and not bp[h * dp + d:(h + 1) * dp].any()
        assert torch.equal(wop[:, h * dp:h * dp + d], wo[:, h * d:(h + 1) * d]) and not wop[:, h * dp + d:(h + 1) * dp].any()
    gw, gb, go = torch.randn_like(wp), torch.randn_like(bp), torch.randn_like(wop)
    ((wp * gw).sum() + (bp * gb).sum() + (wop * go).sum()).backward()
    assert torch.equal(w.grad, gw.view(heads, dp, C)[:, :d].reshape(heads * d, C))
    assert torch.equal(b.grad, gb.view(heads, dp)[:, :d].reshape(-1))
    assert torch.equal(wo.grad, go.view(C, heads, dp)[:, :, :d].reshape(C, heads * d))
    # x W^T on the padded weight = the unpadded product scattered into the padded layout
    x = torch.randn(7, C)
    assert torch.equal((x @ wp.t()).view(7, heads, dp)[:, :, :d].reshape(7, heads * d), x @ w.t())

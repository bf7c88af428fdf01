"""BASELINE config 1 on CPU: CIFAR-20-style DDPM, 2 contributor groups, 4 coalitions, 50 inference
timesteps through the kept entry points unconditional_generation/main.py and unlearn.py
(plumbing: directory grammar, checkpoint keys, resume, jsonl schema, aggregation).  The model classes
are the CPU oracle injected as `backend` - the shipped default backend is the HIP engine."""
import json
import os
import shutil

import numpy as np
import pytest
import torch

import oracle_backend as OB
from src.attributions.methods.datashapley import data_shapley
from unconditional_generation import main as train_main
from unconditional_generation import unlearn as unlearn_main

TINY = dict(block_out_channels=[32, 32, 64, 64], norm_num_groups=8)


@pytest.fixture()
def tiny_registry(monkeypatch):
    from src.ddpm_config import DDPMConfig
    cfg = {**DDPMConfig.cifar100_config}
    cfg["unet_config"] = dict(cfg["unet_config"], **TINY)
    cfg["n_samples"] = 4
    cfg["training_steps"] = dict(cfg["training_steps"], retrain=2)
    cfg["sample_freq"] = dict(cfg["sample_freq"], retrain=2)
    cfg["ckpt_freq"] = dict(cfg["ckpt_freq"], retrain=1)
    monkeypatch.setattr(DDPMConfig, "cifar100_config", cfg)
    return cfg


@pytest.mark.timeout(600)
def test_config1_cpu_plumbing(tmp_path, tiny_registry):
    out = str(tmp_path / "results")
    db_train, db = str(tmp_path / "train.jsonl"), str(tmp_path / "db.jsonl")
    # ---- 1. "pre-train" the full model for 2 steps on the 2-group toy set ----
    a = train_main.parse_args(["--dataset", "toy2", "--method", "retrain", "--outdir", out, "--db", db_train,
                               "--batch_size", "8", "--num_inference_steps", "50", "--device", "cpu",
                               "--keep_all_ckpts", "--log_freq", "1"])
    assert train_main.main(a, backend=OB)
    mdir = os.path.join(out, "toy2", "retrain", "models", "full")
    assert sorted(f for f in os.listdir(mdir) if f.startswith("ckpt")) == ["ckpt_steps_00000001.pt", "ckpt_steps_00000002.pt"]
    ck = torch.load(os.path.join(mdir, "ckpt_steps_00000002.pt"), weights_only=False)
    assert {"unet", "unet_ema", "optimizer", "lr_scheduler", "remaining_idx", "removed_idx", "total_steps_time"} <= set(ck)
    assert ck["unet_ema"]["optimization_step"] == 2 and len(ck["remaining_idx"]) == 128
    assert os.path.exists(os.path.join(out, "toy2", "retrain", "samples", "full", "steps_00000002.png"))
    assert np.load(os.path.join(mdir, "remaining_idx.npy")).shape == (128,)
    # resume: nothing left to do, newest checkpoint is picked up by file name
    assert train_main.main(a, backend=OB)
    # ---- 2. the "pruned" starting point of sFT (architecture + weights in one file) ----
    pdir = os.path.join(out, "toy2", "pruned", "models", "pruner=magnitude_pruning_ratio=0.3_threshold=0.05")
    os.makedirs(pdir)
    torch.save({"unet": ck["unet"], "unet_config": ck["unet_config"]}, os.path.join(pdir, "ckpt_steps_00000000.pt"))
    # ---- 3. four Shapley coalitions through unlearn.py ----
    for seed in range(4):
        u = unlearn_main.parse_args(["--dataset", "toy2", "--method", "gd", "--removal_dist", "shapley",
                                     "--removal_seed", str(seed), "--load", mdir, "--outdir", out, "--db", db,
                                     "--gd_steps", "2", "--n_samples", "8", "--batch_size", "4",
                                     "--num_inference_steps", "50", "--model_behavior", "global",
                                     "--exp_name", f"gd_shapley_seed_{seed}", "--device", "cpu"])
        assert unlearn_main.main(u, backend=OB)
    rows = [json.loads(l) for l in open(db)]
    assert len(rows) == 4
    need = {"dataset", "method", "removal_dist", "removal_seed", "exp_name", "gd_steps", "remaining_idx", "removed_idx",
            "fid_value", "total_steps_time", "total_sampling_time", "trained_steps", "device", "opt_seed"}
    for r in rows:
        assert need <= set(r)
        assert r["method"] == "gd" and r["gd_steps"] == 2 and np.isfinite(r["fid_value"])
        assert len(r["remaining_idx"]) == 64 and len(r["removed_idx"]) == 64       # one of the two groups
    # ---- 4. aggregation exactly as lds.py:203-257,423-430 does it: class masks + data_shapley ----
    labels = [i // 64 for i in range(128)]
    X = np.array([[float(c in {labels[i] for i in r["remaining_idx"]}) for c in (0, 1)] for r in rows])
    y = np.array([r["fid_value"] for r in rows])
    coef = data_shapley(2, X, y, v1=float(y.mean()) - 1.0, v0=float(y.mean()) + 1.0)
    assert coef.shape == (2, 1) and np.isfinite(coef).all()
    assert abs(coef.sum() - (-2.0)) < 1e-8                                           # efficiency: v1 - v0
    # ---- 5. the acceptance harness: the reference's own reader (lds.py::collect_data, run unchanged by
    #         tests/golden/make_lds_golden.py on a db this build wrote) extracted exactly these masks / seeds ----
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "lds_collect.json")))
    for r, g in zip(rows, gold["rows"]):
        for k in ("dataset", "method", "removal_dist", "removal_seed", "exp_name", "gd_steps", "remaining_idx"):
            assert r[k] == g[k], k                                                   # same bookkeeping, bit for bit
        assert isinstance(r["total_steps_time"], float) and isinstance(r["total_sampling_time"], float)
    assert gold["seeds"] == [r["removal_seed"] for r in rows]
    assert np.array_equal(np.array(gold["masks"]), X)                                # class masks as the reference builds them
    assert [b[0] for b in gold["behaviors"]] == [g["fid_value"] for g in gold["rows"]]
    assert np.allclose(y, [b[0] for b in gold["behaviors"]], rtol=0.05)              # same pipeline, CPU float noise only


def test_uniform_removal_reproduces_reference_typeerror(tmp_path, tiny_registry):
    a = train_main.parse_args(["--dataset", "toy2", "--method", "retrain", "--outdir", str(tmp_path), "--removal_dist",
                               "uniform", "--device", "cpu"])
    with pytest.raises(TypeError):
        train_main.main(a, backend=OB)


def test_removal_directory_grammar():
    a = train_main.parse_args(["--dataset", "cifar100", "--method", "retrain"])
    assert train_main.removal_directory(a) == "full"
    a = train_main.parse_args(["--dataset", "cifar100", "--method", "retrain", "--removal_dist", "datamodel",
                               "--datamodel_alpha", "0.25", "--removal_seed", "7"])
    assert train_main.removal_directory(a) == "datamodel/datamodel_alpha=0.25_seed=7"
    a = train_main.parse_args(["--dataset", "cifar100", "--method", "retrain", "--removal_dist", "shapley", "--removal_seed", "3"])
    assert train_main.removal_directory(a) == "shapley/shapley_seed=3"


def test_sd_entry_point_bookkeeping(tmp_path):
    """Directory grammar, removal-unit table, removal_idx.csv and the cosine schedule of the SD LoRA trainer (CPU)."""
    import pandas as pd
    from text_to_image import train_text_to_image_lora as T
    import gad
    a = T.parse_args(["--train_data_dir", str(tmp_path / "artbench"), "--removal_dist", "shapley", "--removal_unit", "artist",
                      "--removal_seed", "3"])
    assert T.removal_directory(a) == "artist_shapley/shapley_seed=3"
    a2 = T.parse_args(["--train_data_dir", "x", "--removal_dist", "datamodel", "--datamodel_alpha", "0.5", "--removal_unit",
                       "artist", "--removal_seed", "1"])
    assert T.removal_directory(a2) == "artist_datamodel_alpha=0.5/datamodel_alpha=0.5_seed=1"
    a3 = T.parse_args(["--train_data_dir", "x", "--removal_dist", "loo", "--removal_unit", "filename", "--loo_idx", "7"])
    assert T.removal_directory(a3) == "filename_loo/loo_idx=7"
    cache = T.synthetic_cache(str(tmp_path / "artbench" / "latent_cache.pt"), n=600, n_artists=258, res=64)
    units = pd.read_csv(tmp_path / "artbench" / "post_impressionism_artists.csv")
    assert len(units) == 258 and cache["latents"].shape == (600, 4, 8, 8)
    os.makedirs(tmp_path / "m")
    rem, rmv = T.coalition_rows(a, str(tmp_path / "m"), units)
    from src.datasets import remove_data_by_shapley
    r2, x2 = remove_data_by_shapley(units, 3)
    assert rem.tolist() == r2.tolist() and rmv.tolist() == x2.tolist() and len(rem) + len(rmv) == 258
    rem_again, _ = T.coalition_rows(a, str(tmp_path / "m"), units)       # second call reads removal_idx.csv
    assert sorted(rem_again.tolist()) == sorted(rem.tolist())
    f = gad.lr_lambda("cosine", 200)
    assert f(0) == 1.0 and abs(f(100) - 0.5) < 1e-12 and f(200) == 0.0 and gad.lr_lambda("constant", 10)(7) == 1.0


def test_prune_lora_ranks(tmp_path):
    """LoRA rank pruning: stops at the target, removes the smallest-magnitude ranks, leaves ragged ranks."""
    import pandas as pd
    from safetensors.torch import load_file, save_file
    from text_to_image import prune_lora as P
    g = torch.Generator().manual_seed(0)
    sd = {}
    dims = [(320, 320), (768, 320), (640, 640), (1280, 1280)]
    for m, (cin, cout) in enumerate(dims):
        scale = torch.linspace(0.1, 2.0, 16)[torch.randperm(16, generator=g)]
        sd[f"unet.blk{m}.attn.processor.to_q_lora.down.weight"] = torch.randn(16, cin, generator=g) * scale[:, None]
        sd[f"unet.blk{m}.attn.processor.to_q_lora.up.weight"] = torch.randn(cout, 16, generator=g) * scale[None, :]
    src = tmp_path / "artbench_post_impressionism" / "retrain" / "models" / "full"
    os.makedirs(src)
    save_file(sd, str(src / "pytorch_lora_weights.safetensors"))
    outdir = P.main(P.parse_args(["--lora_dir", str(src), "--pruning_ratio", "0.5"]))
    assert outdir.endswith("artbench_post_impressionism/pruned_ratio=0.5/models/full")
    out = load_file(os.path.join(outdir, "pytorch_lora_weights.safetensors"))
    total = sum(t.numel() for t in sd.values())
    kept = sum(t.numel() for t in out.values())
    # the reference charges a removed pair 2x the size of the node that was met first (down: in_features, up:
    # out_features), so for non-square projections the realised ratio only approximates the target (info.csv
    # records it as actual_pruning_ratio)
    assert abs(kept / total - 0.5) < 0.05
    ranks = [out[f"unet.blk{m}.attn.processor.to_q_lora.down.weight"].shape[0] for m in range(4)]
    assert all(out[f"unet.blk{m}.attn.processor.to_q_lora.up.weight"].shape[1] == r for m, r in enumerate(ranks))
    assert 0 < min(ranks) and max(ranks) < 16
    # kept ranks of a module are its highest-scoring ones
    d0, u0 = sd["unet.blk0.attn.processor.to_q_lora.down.weight"], sd["unet.blk0.attn.processor.to_q_lora.up.weight"]
    sc = P.rank_scores(d0, u0)
    kept_rows = out["unet.blk0.attn.processor.to_q_lora.down.weight"]
    kept_idx = [i for i in range(16) if any(torch.equal(d0[i], r) for r in kept_rows)]
    assert sorted(kept_idx) == sorted(np.argsort(sc)[16 - len(kept_idx):].tolist())
    info = pd.read_csv(os.path.join(outdir, "info.csv")).set_index("metric")["value"]
    assert int(info["pruned_lora_params"]) == kept and abs(info["actual_pruning_ratio"] - kept / total) < 1e-4


def test_structural_magnitude_pruning(tmp_path):
    """prune.py restatement: dependency spaces cover every prunable dimension exactly once, widths 128->96 /
    256->192 at ratio 0.3, the sliced state_dict loads strictly, kept channels are the high-magnitude ones."""
    from src.ddpm_config import DDPMConfig
    from unconditional_generation import prune as P
    cfg = dict(DDPMConfig.cifar100_config["unet_config"])
    assert P.pruned_width(128, 0.3, 32) == 96 and P.pruned_width(256, 0.3, 32) == 192
    torch.manual_seed(0)
    net = OB.UNet2DModel(**cfg)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    # make channel 5 of down_blocks.0.resnets.0's internal width tiny and channel 6 huge
    for name in ("down_blocks.0.resnets.0.conv1.weight", "down_blocks.0.resnets.0.time_emb_proj.weight"):
        sd[name][5] *= 1e-3
        sd[name][6] *= 30
    sd["down_blocks.0.resnets.0.conv2.weight"][:, 5] *= 1e-3
    sd["down_blocks.0.resnets.0.conv2.weight"][:, 6] *= 30
    new_cfg, new_sd = P.prune_state_dict(cfg, sd, 0.3)
    assert new_cfg["block_out_channels"] == [96, 192, 192, 192]
    small = OB.UNet2DModel(**new_cfg)
    small.load_state_dict(new_sd)                                        # strict
    assert sum(v.numel() for v in new_sd.values()) == sum(p.numel() for p in small.parameters())
    assert new_sd["time_embedding.linear_2.weight"].shape == (384, 384)
    assert new_sd["up_blocks.2.resnets.2.conv1.weight"].shape == (192, 192 + 96, 3, 3)
    spaces = {s.name: s for s in P.build_spaces(cfg, 0.3)}
    inner = spaces["down_blocks.0.resnets.0.inner"]
    keep = P.select_channels(inner, sd, inner.target)
    assert 6 in keep and 5 not in keep and len(keep) == 96
    per_group = np.bincount(keep // 4, minlength=32)                     # 3 of every 4-channel norm group survive
    assert (per_group == 3).all()
    y = small(torch.randn(1, 3, 32, 32), torch.tensor([10])).sample
    assert y.shape == (1, 3, 32, 32) and torch.isfinite(y).all()
    # end to end through the entry point + the sFT loader
    mdir = tmp_path / "toy2" / "retrain" / "models" / "full"
    os.makedirs(mdir)
    torch.save({"unet": sd, "unet_config": cfg}, mdir / "ckpt_steps_00000010.pt")
    out = P.main(P.parse_args(["--load", str(mdir), "--dataset", "toy2", "--outdir", str(tmp_path)]), backend=OB)
    ck = torch.load(os.path.join(out, "ckpt_steps_00000000.pt"), weights_only=False)
    assert ck["unet_config"]["block_out_channels"] == [96, 192, 192, 192]
    assert out.endswith("toy2/pruned/models/pruner=magnitude_pruning_ratio=0.3_threshold=0.05")


def test_taylor_and_diff_pruning_importance(tmp_path):
    """`--pruner taylor | diff-pruning` (reference prune.py:320-332, 358-378): gradients accumulated over the training
    timesteps on ONE batch with ONE noise draw; importance |sum w g| (taylor) or sum |w g| (diff-pruning) per channel,
    mean over the space's tensors; diff-pruning stops once L_t < thr * L_max.  Checked on a small U-Net against a direct
    recomputation from autograd, then end to end through the entry point on the oracle backend."""
    from src.ddpm_config import DDPMConfig
    from unconditional_generation import prune as P
    cfg = dict(DDPMConfig.cifar100_config["unet_config"], block_out_channels=[32, 32, 64, 64], norm_num_groups=8)
    torch.manual_seed(0)
    net = OB.UNet2DModel(**cfg)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    sched = OB.DDPMScheduler(num_train_timesteps=6)
    g = torch.Generator().manual_seed(1)
    clean, noise = torch.rand(4, 3, 32, 32, generator=g) * 2 - 1, torch.randn(4, 3, 32, 32, generator=g)
    msgs = []
    grads = P.taylor_gradients(net, sched, clean, noise, "taylor", 0.05, log=msgs.append)
    assert "6 timesteps" in msgs[0] and all(p.grad is None or not p.grad.any() for p in net.parameters())
    # direct recomputation: sum over t of d mse(model(add_noise(x, n, t), t), n) / d theta
    ref = OB.UNet2DModel(**cfg)
    ref.load_state_dict(sd)
    for t in range(6):
        tt = torch.full((4,), t, dtype=torch.long)
        torch.nn.functional.mse_loss(ref(sched.add_noise(clean, noise, tt), tt).sample, noise).backward()
    for n, p in ref.named_parameters():
        assert torch.allclose(grads[n], p.grad, rtol=1e-5, atol=1e-7), n
    spaces = {s.name: s for s in P.build_spaces(cfg, 0.25)}
    inner = spaces["down_blocks.0.resnets.0.inner"]
    for mode, multi in (("taylor", True), ("diff-pruning", False)):
        score = P.channel_scores(inner, sd, grads, multivariable=multi)
        want = []
        for name, dim, off in inner.scored:
            wg = (sd[name].double() * grads[name].double()).transpose(0, dim).reshape(sd[name].shape[dim], -1)[off:off + inner.width]
            want.append(wg.sum(1).abs() if multi else wg.abs().sum(1))
        want = torch.stack(want).mean(0)
        assert np.allclose(score, (want / want.mean()).numpy(), rtol=1e-10)
        keep = P.select_channels(inner, sd, inner.target, mode, None, grads)
        cpg = inner.width // 8
        for grp in range(8):                                            # within each norm group: the highest-importance channels
            idx = np.arange(grp * cpg, (grp + 1) * cpg)
            kept = [c for c in keep if c in idx]
            assert len(kept) == inner.target // 8 and min(score[kept]) >= max([score[c] for c in idx if c not in kept] + [-1])
    # diff-pruning stops once L_t < thr * L_max: force it with a threshold above 1
    msgs.clear()
    P.taylor_gradients(net, sched, clean, noise, "diff-pruning", 1.1, log=msgs.append)
    assert "1 timesteps" in msgs[0]                                    # L_0 < 1.1 * L_max already at the first step (:372-376)
    # entry point on the oracle backend
    mdir = tmp_path / "toy2" / "retrain" / "models" / "full"
    os.makedirs(mdir)
    torch.save({"unet": sd, "unet_config": cfg}, mdir / "ckpt_steps_00000010.pt")
    out = P.main(P.parse_args(["--load", str(mdir), "--dataset", "toy2", "--outdir", str(tmp_path), "--pruner", "taylor",
                               "--pruning_ratio", "0.25", "--batch_size", "2", "--device", "cpu"]), backend=OB)
    ck = torch.load(os.path.join(out, "ckpt_steps_00000000.pt"), weights_only=False)
    assert ck["unet_config"]["block_out_channels"] == [24, 24, 48, 48]
    assert out.endswith("toy2/pruned/models/pruner=taylor_pruning_ratio=0.25_threshold=0.05")
    small = OB.UNet2DModel(**ck["unet_config"])
    small.load_state_dict(ck["unet"])


def test_head_grouped_pruning_of_the_celeba_topology():
    """prune.py:337-342 (channel_groups[to_q/k/v] = heads): every head of q / k / v keeps the SAME in-head channels, the
    stream widths follow the GroupNorm rule, the sliced state_dict loads strictly into the `attention_layout` model and
    - when the dropped channels are exactly zero - the pruned attention computes the same function."""
    from src.ddpm_config import DDPMConfig
    from unconditional_generation import prune as P
    cfg = dict(DDPMConfig.celeba_config["unet_config"], block_out_channels=[64, 128, 128, 128], attention_head_dim=16,
               norm_num_groups=16, sample_size=16)
    assert P.pruned_head_dim(448, 32, 0.3) == 23 and P.pruned_head_dim(896, 32, 0.3) == 23       # the real CelebA widths
    assert P.pruned_head_dim(128, 16, 0.3) == 12
    torch.manual_seed(0)
    net = OB.UNet2DModel(**cfg)
    sd = {k: v.detach().clone() for k, v in net.state_dict().items()}
    # in-head positions 1, 5, 9, 13 of every head of one attention: zero q / k / v rows (and the to_out columns of v)
    pre = "down_blocks.1.attentions.0"
    dead = [h * 16 + j for h in range(8) for j in (1, 5, 9, 13)]
    for proj in ("to_q", "to_k", "to_v"):
        sd[f"{pre}.{proj}.weight"][dead] = 0
        sd[f"{pre}.{proj}.bias"][dead] = 0
    new_cfg, new_sd = P.prune_state_dict(cfg, sd, 0.3)
    assert new_cfg["block_out_channels"] == [48, 96, 96, 96] and new_cfg["attention_layout"] == [[4, 11], [8, 12], [8, 12], [8, 12]]
    assert new_sd[f"{pre}.to_q.weight"].shape == (96, 96) and new_sd[f"{pre}.to_out.0.weight"].shape == (96, 96)
    spaces = {s.name: s for s in P.build_spaces(cfg, 0.3)}
    for nm in (f"{pre}.qk", f"{pre}.v"):
        keep = P.select_channels(spaces[nm], sd, spaces[nm].target)
        assert len(keep) == 96 and not set(keep) & set(dead)
        per_head = keep.reshape(8, 12) - 16 * np.arange(8)[:, None]
        assert (per_head == per_head[0]).all()                                 # identical in-head positions in every head
    small = OB.UNet2DModel(**new_cfg)
    small.load_state_dict(new_sd)                                              # strict
    assert small.down_blocks[1].attentions[0].heads == 8
    y = small(torch.randn(1, 3, 16, 16), torch.tensor([10])).sample
    assert y.shape == (1, 3, 16, 16) and torch.isfinite(y).all()
    # function preservation on the attention whose dropped q/k/v channels were zero: slice the stream channels by hand
    att, att_s = net.down_blocks[1].attentions[0], small.down_blocks[1].attentions[0]
    net.load_state_dict(sd)
    skeep = P.select_channels(spaces["down_blocks.1.resnets.0.out"], sd, 96)
    qk_keep = P.select_channels(spaces[f"{pre}.qk"], sd, 96)
    v_keep = P.select_channels(spaces[f"{pre}.v"], sd, 96)
    h = torch.randn(2, 64, 128)                                                # [B, T, C] tokens after the group norm
    with torch.no_grad():
        def core(a, x, heads):
            q, k, v = a.to_q(x), a.to_k(x), a.to_v(x)
            d = q.shape[-1] // heads
            sp = lambda z: z.view(2, 64, heads, d).transpose(1, 2)
            return torch.nn.functional.scaled_dot_product_attention(sp(q), sp(k), sp(v)).transpose(1, 2).reshape(2, 64, -1)
        hs = h.clone()
        mask = torch.zeros(128, dtype=torch.bool)
        mask[torch.as_tensor(skeep)] = True
        hs[..., ~mask] = 0                                                     # the pruned stream channels are silent
        got = core(att_s, h[..., torch.as_tensor(skeep)], 8)
    # the softmax scale changes with the head dim (1/sqrt(12) instead of 1/sqrt(16)): compare against the full model
    # evaluated at the new scale
    with torch.no_grad():
        q, k, v = (getattr(att, n)(hs) for n in ("to_q", "to_k", "to_v"))
        sp = lambda z, idx: z[..., torch.as_tensor(idx)].view(2, 64, 8, 12).transpose(1, 2)
        ref = torch.nn.functional.scaled_dot_product_attention(sp(q, qk_keep), sp(k, qk_keep), sp(v, v_keep)).transpose(1, 2).reshape(2, 64, -1)
    assert torch.allclose(got, ref, atol=1e-5)


def test_artbench_metadata_to_latent_cache(tmp_path):
    """metadata.csv / {style}_artists.csv grammar of the reference (create_metadata.py:77-95,113-114) joined with
    precomputed latents into the trainer's latent cache."""
    import pandas as pd
    import torch
    from text_to_image.artbench.metadata import assemble_latent_cache, read_metadata, unit_table
    rows = []
    for style, n in (("post_impressionism", 5), ("ukiyo_e", 3)):
        for i in range(n):
            f = f"{style}/artist-{i % 2}_work-{i}.jpg"
            rows.append({"file_name": f, "caption": f"a {style} painting", "artist": f"artist-{i % 2}", "style": style,
                         "filename": f})
    pd.DataFrame(rows).to_csv(tmp_path / "metadata.csv", index=False)
    pd.DataFrame({"artist": ["artist-0", "artist-1"]}).to_csv(tmp_path / "post_impressionism_artists.csv", index=False)
    assert len(read_metadata(str(tmp_path))) == 8
    assert len(read_metadata(str(tmp_path), "style", "post_impressionism")) == 5
    assert unit_table(str(tmp_path), "post_impressionism", "artist")["artist"].tolist() == ["artist-0", "artist-1"]
    lat = {r["file_name"]: torch.full((4, 2, 2), float(k)) for k, r in enumerate(rows)}
    cache = assemble_latent_cache(str(tmp_path), lat, torch.ones(77, 8), cls="post_impressionism")
    assert cache["latents"].shape == (5, 4, 2, 2) and cache["latents"][:, 0, 0, 0].tolist() == [0., 1., 2., 3., 4.]
    assert cache["text_emb"].shape == (1, 77, 8) and cache["artist"] == ["artist-0", "artist-1"] * 2 + ["artist-0"]
    back = torch.load(tmp_path / "latent_cache.pt", weights_only=False)
    assert back["filename"] == cache["filename"]
    per_caption = assemble_latent_cache(str(tmp_path), lat, {"a ukiyo_e painting": torch.zeros(77, 8)}, cls="ukiyo_e",
                                        out=str(tmp_path / "u.pt"))
    assert per_caption["text_emb"].shape == (3, 77, 8)
    del lat[rows[0]["file_name"]]
    import pytest
    with pytest.raises(KeyError):
        assemble_latent_cache(str(tmp_path), lat, torch.ones(77, 8), cls="post_impressionism")


def test_sd_behaviour_rows_are_what_the_reference_reader_extracted():
    """tests/golden/sd_lds_collect.json = the outputs of the reference's text_to_image/shapley_lds.py::collect_data
    (:105-135, run unchanged) on rows built by compute_model_behaviors.py::assemble_row; the same rows are rebuilt here."""
    from text_to_image import compute_model_behaviors as M
    gold = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "sd_lds_collect.json")))
    rows = []
    for k, remaining in enumerate(([0, 2, 5, 7], [1, 2, 3, 4, 8, 9], [6])):
        args = M.parse_args(["--reference_lora_dir", "ref", "--lora_dir", f"lora_{k}", "--db", "db.jsonl", "--num_images", "2",
                             "--exp_name", f"retrain_artist_shapley_seed_{k}"])
        lists = {b: [0.1 * (k + 1) + 0.01 * i + 0.001 * j for i in range(2)] for j, b in enumerate(M.BEHAVIOURS)}
        times = {b: [1.0 + i for i in range(2)] for b in M.BEHAVIOURS}
        rows.append(json.loads(json.dumps(M.assemble_row(args, lists, times, remaining, [i for i in range(10) if i not in remaining]))))
    masks = np.zeros((3, 10))
    for i, r in enumerate(rows):
        masks[i, r["remaining_idx"]] = 1
    assert np.array_equal(masks, np.array(gold["masks"]))
    # (pandas' json float parser may differ from Python's in the last ulp)
    assert np.allclose([[r["aesthetic_score_avg"]] for r in rows], gold["aesthetic_score_avg"], rtol=1e-14, atol=0)
    assert np.allclose([[r[f"generated_image_{i}_simple_loss"] for i in range(2)] for r in rows], gold["simple_loss"], rtol=1e-14, atol=0)
    assert [int(r["exp_name"].split("seed_")[1]) for r in rows] == gold["subset_seed"]
    # the reference's own column grammar (:459-498): per-image values and times, quantiles, totals
    r = rows[0]
    for key in ("generated_image_1_ssim", "generated_image_1_nrmse_time", "aesthetic_score_0.5", "aesthetic_score_0.9",
                "clip_prompt_score_0.75", "clip_prompt_score_avg", "aesthetic_score_time", "removal_idx", "reference_lora_dir",
                "lora_steps", "cls", "seed", "no_duplicate"):
        assert key in r, key
    assert r["aesthetic_score_0.5"] == pytest.approx(np.quantile([0.105, 0.115], 0.5))

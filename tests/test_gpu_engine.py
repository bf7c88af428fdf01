"""GPU tests of the engine around the kernels: fused trainer vs the oracle training step, fused-batch sampler vs
per-batch pipeline calls, entry points and the sharded coalition runner on the HIP backend."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")
TINY = dict(block_out_channels=[32, 32, 64, 64], norm_num_groups=8)


def _cfg():
    from src.ddpm_config import DDPMConfig
    return dict(DDPMConfig.cifar100_config["unet_config"], **TINY), DDPMConfig.cifar100_config["scheduler_config"]


def test_fused_trainer_matches_oracle_training_steps():
    import gad
    from oracle import diffusers_ref as R
    ucfg, scfg = _cfg()
    torch.manual_seed(0)
    ref = R.UNet2DModel(**ucfg)
    net = gad.UNet2DModel(**ucfg)
    net.load_state_dict(ref.state_dict())
    net.to(dev)
    opt = torch.optim.Adam(ref.parameters(), lr=1e-4)
    ema_r = R.EMAModel(ref.parameters())
    ema_g = gad.EMAModel(net.parameters())
    for e in (ema_r, ema_g):
        e.optimization_step = 5000
    tr = gad.FusedTrainer(net, gad.DDPMScheduler(**scfg), ema_g, lr=1e-4)
    sch = R.DDPMScheduler(**scfg)
    g = torch.Generator().manual_seed(1)
    for i in range(3):
        x, n = torch.rand(8, 3, 32, 32, generator=g) * 2 - 1, torch.randn(8, 3, 32, 32, generator=g)
        t = R.antithetic_timesteps(torch.randint(0, 1000, (5,), generator=g), 1000, 8)
        lr_, gn = R.train_step(ref, opt, ema_r, sch, x, n, t)
        lg = tr.step(x.to(dev), n.to(dev), t.to(dev))
        assert abs(lg.item() - lr_.item()) < 1e-4 * max(1.0, abs(lr_.item()))
        assert abs(tr.grad_norm().item() - gn.item()) < 2e-3 * gn.item()
    for (k, a), b in zip(net.state_dict().items(), ref.state_dict().values()):
        assert torch.allclose(a.cpu(), b, atol=2e-5), k
    for a, b in zip(ema_g.shadow_params, ema_r.shadow_params):
        assert torch.allclose(a.cpu(), b, atol=2e-6)
    assert ema_g.optimization_step == ema_r.optimization_step == 5003


def test_fused_sampler_equals_per_batch_pipeline_calls():
    import gad
    from gad.coalition import FusedSampler
    ucfg, _ = _cfg()
    torch.manual_seed(0)
    net = gad.UNet2DModel(**ucfg).to(dev).eval()
    sch = gad.DDIMScheduler()
    fused = FusedSampler(net, sch, batch_size=4, fuse=3).generate(10, 5)          # batches 4,4,2 -> one launch group
    pipe = gad.DDPMPipeline(net, gad.DDIMScheduler())
    parts = []
    for counter, bs in enumerate([4, 4, 2]):
        im = pipe(batch_size=bs, generator=torch.Generator().manual_seed(counter), num_inference_steps=5,
                  output_type="numpy").images
        x = torch.from_numpy(im).permute(0, 3, 1, 2)
        parts.append(x.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8).float().div_(255))
    want = torch.cat(parts)
    got = fused.cpu()
    assert got.shape == want.shape == (10, 3, 32, 32)
    # identical per-sample arithmetic; allow one 8-bit level where a value sits on a rounding boundary
    assert (got - want).abs().max().item() <= 1 / 255 + 1e-6
    assert ((got - want).abs() > 1e-6).float().mean().item() < 1e-3


def test_generate_images_routes_through_the_fused_sampler():
    """the kept CLI's generate_images (src/diffusion_utils.py:319-357) on a gad pipeline = the engine's
    fused-batch sampler - same per-batch CPU-generator seeds, same images as batch-by-batch pipeline calls (one 8-bit
    level allowed where a value sits on a rounding boundary: the launch width changes the fp32 summation order)."""
    from types import SimpleNamespace
    import gad
    from src.diffusion_utils import _fused_sampler_for, generate_images
    ucfg, _ = _cfg()
    torch.manual_seed(0)
    net = gad.UNet2DModel(**ucfg).to(dev).eval()
    args = SimpleNamespace(batch_size=4, n_samples=10, num_inference_steps=5)
    pipe = gad.DDPMPipeline(net, gad.DDIMScheduler())
    assert _fused_sampler_for(pipe, 4, 32) is not None and _fused_sampler_for(pipe, 4, 1) is None
    fused = generate_images(args, pipe).cpu()
    per_batch = generate_images(args, pipe, fuse=1).cpu()
    assert fused.shape == per_batch.shape == (10, 3, 32, 32)
    assert (fused - per_batch).abs().max().item() <= 1 / 255 + 1e-6
    assert ((fused - per_batch).abs() > 1e-6).float().mean().item() < 1e-3
    assert (fused * 255 - (fused * 255).round()).abs().max().item() < 1e-4    # the uint8 round trip survived


def test_entry_points_and_sharded_runner_on_gpu(tmp_path, monkeypatch):
    from src.ddpm_config import DDPMConfig
    cfg = {**DDPMConfig.cifar100_config}
    cfg["unet_config"] = dict(cfg["unet_config"], **TINY)
    cfg["n_samples"] = 4
    cfg["training_steps"] = dict(cfg["training_steps"], retrain=3)
    cfg["ckpt_freq"] = dict(cfg["ckpt_freq"], retrain=3)
    cfg["sample_freq"] = dict(cfg["sample_freq"], retrain=3)
    monkeypatch.setattr(DDPMConfig, "cifar100_config", cfg)
    from unconditional_generation import main as train_main
    from unconditional_generation import unlearn as unlearn_main
    out, db = str(tmp_path / "res"), str(tmp_path / "db.jsonl")
    a = train_main.parse_args(["--dataset", "toy2", "--method", "retrain", "--outdir", out, "--batch_size", "16",
                               "--num_inference_steps", "10", "--log_freq", "1"])
    assert train_main.main(a)
    mdir = os.path.join(out, "toy2", "retrain", "models", "full")
    ck = torch.load(os.path.join(mdir, "ckpt_steps_00000003.pt"), weights_only=False)
    assert ck["unet"]["conv_in.weight"].shape == (32, 3, 3, 3) and ck["unet_ema"]["optimization_step"] == 3
    pdir = os.path.join(out, "toy2", "pruned", "models", "pruner=magnitude_pruning_ratio=0.3_threshold=0.05")
    os.makedirs(pdir)
    torch.save({"unet": ck["unet"], "unet_config": ck["unet_config"]}, os.path.join(pdir, "ckpt_steps_00000000.pt"))
    u = unlearn_main.parse_args(["--dataset", "toy2", "--method", "gd", "--removal_dist", "shapley", "--removal_seed", "1",
                                 "--load", mdir, "--outdir", out, "--db", db, "--gd_steps", "3", "--n_samples", "16",
                                 "--batch_size", "8", "--num_inference_steps", "10", "--model_behavior", "global"])
    assert unlearn_main.main(u)
    row = json.loads(open(db).readline())
    assert np.isfinite(row["fid_value"]) and len(row["remaining_idx"]) == 64 and row["trained_steps"] == 3
    assert 0.0 <= row["precision"] <= 1.0 and 0.0 <= row["recall"] <= 1.0 and row["is"] >= 1.0
    # in-process scheduler on the same toy problem
    from gad.coalition import CoalitionEngine, run_sharded
    eng = CoalitionEngine("toy2", device=dev, gd_steps=2, n_samples=16, sample_batch=8, fuse=2, num_inference_steps=5,
                          unet_overrides=TINY, feature_dims=64)
    db2 = str(tmp_path / "db2.jsonl")
    recs = run_sharded(eng, [0, 1, 2], db_path=db2)
    assert [r.removal_seed for r in recs] == [0, 1, 2] and all(np.isfinite(r.fid_value) for r in recs)
    rows = [json.loads(l) for l in open(db2)]
    assert [r["removal_seed"] for r in rows] == [0, 1, 2]
    assert run_sharded(eng, [0, 1, 2], db_path=db2) == []            # idempotent: everything is already in the db


def test_celeba_latent_mode_entry_points(tmp_path, monkeypatch):
    """BASELINE config 3 in latent space: CelebA-HQ rows = labels.csv x the VQ-VAE latent dictionary
    (`--precompute_stage reuse`, main.py:531-546), sFT by celebrity group, latent sampling with the LDM scheduler, and the
    demographic-diversity behaviours (entropy / cluster_count / cluster_proportions, unlearn.py:787-803) in the jsonl."""
    import pandas as pd
    import src.constants as constants
    from src.ddpm_config import DDPMConfig
    cfg = {**DDPMConfig.celeba_config}
    cfg["unet_config"] = dict(cfg["unet_config"], block_out_channels=[32, 64, 64, 64], attention_head_dim=8,
                              norm_num_groups=8, sample_size=16)
    cfg["n_samples"] = 4
    cfg["batch_size"] = 8
    for k in ("training_steps", "ckpt_freq", "sample_freq"):
        cfg[k] = dict(cfg[k], retrain=2)
    monkeypatch.setattr(DDPMConfig, "celeba_config", cfg)
    root = tmp_path / "datasets" / "celeba_hq_256_50_resized"
    root.mkdir(parents=True)
    names = [f"{i:05d}.jpg" for i in range(60)]
    pd.DataFrame({"filename": names, "celeb": [i % 5 for i in range(60)]}).to_csv(root / "labels.csv", index=False)
    g = torch.Generator().manual_seed(0)
    torch.save({n: torch.randn(3, 16, 16, generator=g) * 0.5 + 0.2 * (i % 5) for i, n in enumerate(names)}, tmp_path / "vqvae_output.pt")
    monkeypatch.setenv("GAD_DATA", "real")
    monkeypatch.setenv("GAD_LATENTS", str(tmp_path / "vqvae_output.pt"))
    monkeypatch.setattr(constants, "DATASET_DIR", str(tmp_path / "datasets"))
    from unconditional_generation import main as train_main
    from unconditional_generation import unlearn as unlearn_main
    out, db = str(tmp_path / "res"), str(tmp_path / "db.jsonl")
    with pytest.raises(NotImplementedError):                          # pixels would need the VQ-VAE
        train_main.main(train_main.parse_args(["--dataset", "celeba", "--method", "retrain", "--outdir", out]))
    a = train_main.parse_args(["--dataset", "celeba", "--method", "retrain", "--outdir", out, "--precompute_stage", "reuse",
                               "--num_inference_steps", "5", "--log_freq", "1"])
    assert train_main.main(a)
    mdir = os.path.join(out, "celeba", "retrain", "models", "full")
    ck = torch.load(os.path.join(mdir, "ckpt_steps_00000002.pt"), weights_only=False)
    pdir = os.path.join(out, "celeba", "pruned", "models", "pruner=magnitude_pruning_ratio=0.3_threshold=0.05")
    os.makedirs(pdir)
    torch.save({"unet": ck["unet"], "unet_config": ck["unet_config"]}, os.path.join(pdir, "ckpt_steps_00000000.pt"))
    u = unlearn_main.parse_args(["--dataset", "celeba", "--method", "gd", "--removal_dist", "shapley", "--removal_seed", "2",
                                 "--load", mdir, "--outdir", out, "--db", db, "--gd_steps", "2", "--n_samples", "12",
                                 "--batch_size", "6", "--num_inference_steps", "5", "--model_behavior", "global",
                                 "--precompute_stage", "reuse"])
    assert unlearn_main.main(u)
    row = json.loads(open(db).readline())
    kept = {i % 5 for i in row["remaining_idx"]}
    assert 0 < len(kept) < 5 and all((i % 5) in kept for i in row["remaining_idx"])      # whole celebrity groups
    assert len(row["cluster_count"]) == 20 and sum(row["cluster_count"]) == 12
    assert abs(sum(row["cluster_proportions"]) - 1.0) < 1e-9 and 0.0 <= row["entropy"] <= np.log2(20) + 1e-9
    assert "fid_value" not in row and row["trained_steps"] == 2
    # the same coalitions through the one-coalition-per-GPU scheduler (configs 3-5 in flight: gad.cycles + run_sharded)
    from gad.coalition import run_sharded
    from gad.cycles import CelebaCycle
    db2 = str(tmp_path / "db_sharded.jsonl")
    cyc = CelebaCycle(dev, ["--load", mdir, "--outdir", out, "--gd_steps", "2", "--n_samples", "12", "--batch_size", "6",
                            "--num_inference_steps", "5", "--precompute_stage", "reuse"])
    assert cyc.n_groups == 5 and len(cyc.extra_keys) == 21
    recs = run_sharded(cyc, [2, 3], db_path=db2, verbose=True)
    rows = [json.loads(l) for l in open(db2)]
    assert [r["removal_seed"] for r in rows] == [2, 3] and not os.path.exists(db2 + ".rank0")
    assert rows[0]["remaining_idx"] == row["remaining_idx"] and rows[0]["cluster_count"] == row["cluster_count"]
    assert rows[0]["entropy"] == pytest.approx(row["entropy"], abs=1e-12)      # same seeds -> the same coalition, bit for bit
    assert recs[0].extra[0] == pytest.approx(row["entropy"]) and recs[0].extra[1:] == row["cluster_count"]
    assert recs[0].remaining_classes == sorted(kept)
    v = recs[0].pack(cyc.n_groups, 21)                                         # what the final all_gather carries
    assert type(recs[0]).unpack(v, 21).extra == recs[0].extra


def test_ddpm_pipeline_with_the_ancestral_scheduler():
    """DDPMPipeline(unet, DDPMScheduler()) - the pipeline object main.py builds (:550-552); the reference only samples
    through DDIM, this keeps the object usable: images in [0,1], reproducible with a CPU generator."""
    import gad
    ucfg, scfg = _cfg()
    torch.manual_seed(0)
    net = gad.UNet2DModel(**ucfg).to(dev).eval()
    pipe = gad.DDPMPipeline(net, gad.DDPMScheduler(**scfg))
    a = pipe(batch_size=2, num_inference_steps=6, output_type="numpy", generator=torch.Generator().manual_seed(3)).images
    b = pipe(batch_size=2, num_inference_steps=6, output_type="numpy", generator=torch.Generator().manual_seed(3)).images
    assert a.shape == (2, 32, 32, 3) and np.array_equal(a, b) and 0.0 <= a.min() and a.max() <= 1.0 and np.isfinite(a).all()


def test_ldm_pipeline_decode_protocol():
    """LDMPipeline(unet, vqvae, scheduler): latents / scaling_factor -> vqvae.decode(...).sample -> [0,1] images
    (reference src/diffusion_utils.py:393-412); vqvae=None returns the post-processed latents."""
    import gad
    from types import SimpleNamespace
    ucfg, _ = _cfg()
    torch.manual_seed(0)
    net = gad.UNet2DModel(**ucfg).to(dev).eval()

    class ToyVQ(torch.nn.Module):                                  # decode protocol only: 2x nearest upsample of scaled latents
        config = SimpleNamespace(scaling_factor=0.5)

        def decode(self, z):
            return SimpleNamespace(sample=torch.nn.functional.interpolate(z, scale_factor=2.0, mode="nearest"))

    sch = gad.DDIMScheduler(clip_sample=False)
    lat = gad.LDMPipeline(net, None, sch)(batch_size=2, num_inference_steps=3, output_type="numpy",
                                           generator=torch.Generator().manual_seed(1)).images
    img = gad.LDMPipeline(net, ToyVQ(), sch)(batch_size=2, num_inference_steps=3, output_type="numpy",
                                            generator=torch.Generator().manual_seed(1)).images
    assert lat.shape == (2, 32, 32, 3) and img.shape == (2, 64, 64, 3)
    # decoded = clamp((latent / 0.5) / 2 + 0.5): recover the raw latents where the un-decoded image is not saturated
    raw = lat * 2 - 1
    want = np.clip(raw / 0.5 / 2 + 0.5, 0, 1)
    inside = (lat > 1e-6) & (lat < 1 - 1e-6)
    assert np.allclose(img[:, ::2, ::2][inside], want[inside], atol=1e-5)


def test_graph_replay_reads_the_current_winograd_weights():
    """The captured U-Net forward reads the Winograd-transformed weights at a fixed address: a pipeline whose weights
    change between two calls (EMA copy, the next coalition's checkpoint) must sample from the NEW weights on replay
    (pipelines.DDPMPipeline._run_steps refreshes the shadows before replaying)."""
    import gad
    from gad import ops
    from src.ddpm_config import DDPMConfig
    torch.manual_seed(0)
    net = gad.UNet2DModel(**dict(DDPMConfig.cifar100_config["unet_config"], block_out_channels=[64, 64, 128, 128])).to(dev).eval()
    sch = gad.DDIMScheduler(**DDPMConfig.cifar100_config["scheduler_config"])
    pipe = gad.DDPMPipeline(net, sch)
    B = 128                                                     # 32 x 32 maps of 64 channels: 128 x 256 tiles -> the Winograd route
    call = lambda p: p(batch_size=B, num_inference_steps=2, output_type="tensor", generator=torch.Generator().manual_seed(3)).images
    ops.PROFILER = prof = ops.GemmProfiler()
    try:
        with torch.no_grad():
            net.forward_nhwc(torch.zeros(B, 32, 32, 3, device=dev), torch.zeros(B, device=dev, dtype=torch.int64))
        torch.cuda.synchronize()
    finally:
        ops.PROFILER = None
    assert any(k[0].startswith("conv_fwd_wino") for k in prof.summary())
    a = call(pipe)
    with torch.no_grad():
        for p in net.parameters():
            if p.ndim == 4:
                p.mul_(0.5)
    b = call(pipe)                                              # replays the captured graph
    eager = gad.DDPMPipeline(net, sch)
    eager.use_graph = False
    want = call(eager)
    assert not torch.equal(a, b)
    assert torch.equal(b, want)


def test_batched_time_embedding_projection_matches_per_block():
    """Sampling computes the 22 `time_emb_proj` Linears of the U-Net as ONE GEMM on concatenated weights (UNet2DModel._temb_rows)
    and every ResnetBlock2D adds its column block in its first convolution's epilogue: same output as the per-block launches (the
    tile / split-K plan of the wide GEMM sums in another order: fp32 rounding), rebuilt when a weight changes, not used with grad."""
    import gad
    from gad.nn import ResnetBlock2D
    from src.ddpm_config import DDPMConfig
    torch.manual_seed(0)
    net = gad.UNet2DModel(**DDPMConfig.cifar100_config["unet_config"]).to(dev).eval()
    x = torch.randn(8, 32, 32, 3, device=dev)
    t = torch.randint(0, 1000, (8,), device=dev)
    with torch.no_grad():
        y = net.forward_nhwc(x, t)
        assert net._temb_w.shape == (sum(r.time_emb_proj.weight.shape[0] for r in net.modules() if isinstance(r, ResnetBlock2D)), 512)
        keep = net._temb_rows
        net._temb_rows = lambda temb: ()
        y0 = net.forward_nhwc(x, t)
        net._temb_rows = keep
        assert (y - y0).abs().max().item() < 2e-5 * max(1.0, y0.abs().max().item())
        r0 = next(m for m in net.modules() if isinstance(m, ResnetBlock2D))
        r0.time_emb_proj.bias.add_(torch.randn_like(r0.time_emb_proj.bias))   # torch-side write (not a constant: GroupNorm would remove it): the concatenation must follow
        y1 = net.forward_nhwc(x, t)
        net._temb_rows = lambda temb: ()
        y2 = net.forward_nhwc(x, t)
        net._temb_rows = keep
        assert (y1 - y2).abs().max().item() < 2e-5 * max(1.0, y2.abs().max().item()) and (y1 - y).abs().max().item() > 1e-4
    assert all(getattr(m, "_temb_row", None) is None for m in net.modules())          # nothing left behind for a later grad-mode call


def test_pipelined_coalitions_equal_sequential(monkeypatch):
    """CoalitionEngine.run_pipelined (two coalitions in flight: the training phase of one beside the sampling phase of the previous
    one, on two HIP streams, one host thread) gives the records `run_coalition` gives one after the other - bit for bit:
    same kernels in the same order per stream, a workspace per stream, the training RNG seeded per coalition as before."""
    monkeypatch.setenv("GAD_SYNTH_SCALE", "0.1")
    from gad.coalition import CoalitionEngine
    eng = CoalitionEngine("cifar100", device="cuda:0", gd_steps=6, n_samples=64, num_inference_steps=4, fuse=1, sample_batch=32,
                          unet_overrides=dict(block_out_channels=(32, 64, 64, 64), norm_num_groups=8))
    seq = [eng.run_coalition(s) for s in (0, 1, 2)]
    seen = []
    pip = eng.run_pipelined([0, 1, 2], on_record=seen.append, n_train=1)
    assert [r.removal_seed for r in pip] == [0, 1, 2] and seen == pip
    pip2 = eng.run_pipelined([0, 1, 2, 3, 4], n_train=2)                   # two training phases taking turns beside the sampler
    assert [r.removal_seed for r in pip2] == [0, 1, 2, 3, 4]
    seq = seq + [eng.run_coalition(s) for s in (3, 4)]
    for a, b in list(zip(seq, pip)) + list(zip(seq, pip2)):
        assert (a.removal_seed, a.n_remaining, a.remaining_classes, a.trained_steps) == (b.removal_seed, b.n_remaining, b.remaining_classes, b.trained_steps)
        assert a.fid_value == b.fid_value and a.loss_last == b.loss_last
        assert a.inception_score == b.inception_score and a.precision == b.precision and a.recall == b.recall
    # a coalition that raises is reported and dropped, the others finish
    real = eng.coalition
    eng.coalition = lambda seed: (_ for _ in ()).throw(RuntimeError("synthetic")) if seed == 1 else real(seed)
    errs = []
    got = eng.run_pipelined([0, 1, 2], on_error=lambda s, e: errs.append((s, str(e))))
    eng.coalition = real
    assert [r.removal_seed for r in got] == [0, 2] and errs == [(1, "synthetic")]
    assert got[0].fid_value == seq[0].fid_value and got[1].fid_value == seq[2].fid_value


def test_graphed_training_step_equals_eager_cifar():
    """FusedTrainer(use_graph=True) on the fp32 CIFAR U-Net (Winograd shadows, rotated weights, EMA, all parameters trainable): weights,
    Adam moments, EMA shadow and losses bit-identical to the eager launches over several steps."""
    import torch
    import gad
    from gad.coalition import antithetic_timesteps
    from src.ddpm_config import DDPMConfig
    dev = torch.device("cuda:0")
    cfg = dict(DDPMConfig.cifar100_config["unet_config"])
    cfg["block_out_channels"] = (64, 128, 128, 128)
    sch = gad.DDPMScheduler(**DDPMConfig.cifar100_config["scheduler_config"])
    g = torch.Generator(device=dev).manual_seed(5)
    xs = [torch.randn(32, 3, 32, 32, device=dev, generator=g) for _ in range(6)]
    ns = [torch.randn(32, 3, 32, 32, device=dev, generator=g) for _ in range(6)]
    ts = [antithetic_timesteps(1000, 32, dev, generator=g) for _ in range(6)]

    def run(use_graph):
        torch.manual_seed(0)
        net = gad.UNet2DModel(**cfg).to(dev)
        ema = gad.EMAModel(net.parameters())
        tr = gad.FusedTrainer(net, sch, ema, lr=1e-4, max_grad_norm=1.0, use_graph=use_graph)
        losses = [float(tr.step(xs[i], ns[i], ts[i]).item()) for i in range(6)]
        return tr, losses
    eager, le = run(False)
    graphed, lg = run(True)
    assert graphed._graph is not None and not graphed._graph_failed, "the step was not captured"
    assert le == lg
    for a, b in ((eager.flat, graphed.flat), (eager.m, graphed.m), (eager.v, graphed.v), (eager.ema_flat, graphed.ema_flat)):
        assert torch.equal(a, b)

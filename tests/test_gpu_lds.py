"""North-star acceptance proxy (VERDICT r1 #9d): the linear datamodeling score computed from coalitions
run on the HIP engine against the same coalitions run on the CPU oracle - same toy problem, same seeds, same host-drawn
randomness (batches, noise, timesteps, sampler noise), same scorer - must agree within 0.02 (2 points on lds.py's x100
scale).  Toy problem: CIFAR-20 layout at GAD_SYNTH_SCALE=0.032 (20 contributor classes x 16 images), a 4-stage U-Net,
3 sFT steps at B=32, 16 samples x 5 DDIM steps per coalition; 20 Shapley coalitions to fit, 3 test sets x 12
datamodel(alpha=0.5) subsets, full / null behaviours for the efficiency constraint; behaviour = Frechet distance under
the test backend's projection features (the same function scores both sides)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")
GD_STEPS, B, N_SAMPLES, SAMPLE_B, INF_STEPS = 3, 32, 16, 8, 5


def _coalition_images(kind, base_sd, x_all, remaining, rng_seed, ucfg, scfg):
    """sFT on `remaining`, EMA weights, DDIM samples -> [N,3,32,32] in {k/255}.  All randomness comes from CPU generators."""
    import gad
    from oracle import diffusers_ref as R
    g = torch.Generator().manual_seed(rng_seed)
    if kind == "hip":
        net = gad.UNet2DModel(**ucfg)
        net.load_state_dict(base_sd)
        net.to(dev)
        ema = gad.EMAModel(net.parameters())
        tr = gad.FusedTrainer(net, gad.DDPMScheduler(**scfg), ema, lr=2e-3)
    else:
        net = R.UNet2DModel(**ucfg)
        net.load_state_dict(base_sd)
        ema = R.EMAModel(net.parameters())
        opt = torch.optim.Adam(net.parameters(), lr=2e-3)
        sch = R.DDPMScheduler(**scfg)
    xs = x_all[torch.as_tensor(np.asarray(remaining), dtype=torch.long)]
    for _ in range(GD_STEPS if len(remaining) else 0):
        sel = torch.randperm(len(xs), generator=g)[:B]
        image = xs[sel]
        noise = torch.randn(image.shape, generator=g)
        ts = R.antithetic_timesteps(torch.randint(0, 1000, (len(sel) // 2 + 1,), generator=g), 1000, len(sel))
        if kind == "hip":
            tr.step(image.to(dev), noise.to(dev), ts.to(dev))
        else:
            R.train_step(net, opt, ema, sch, image, noise, ts)
    ema.copy_to(net.parameters())
    net.eval()
    pipe = gad.DDPMPipeline(net, gad.DDIMScheduler()) if kind == "hip" else R.DDPMPipeline(net, R.DDIMScheduler())
    out = []
    for b in range(N_SAMPLES // SAMPLE_B):
        kw = dict(batch_size=SAMPLE_B, generator=torch.Generator().manual_seed(b), num_inference_steps=INF_STEPS)
        im = pipe(output_type="numpy", **kw).images if kind == "hip" else pipe(**kw).images
        x = torch.from_numpy(np.asarray(im)).permute(0, 3, 1, 2)
        out.append(x.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8).float().div_(255))
    return torch.cat(out)


def test_lds_from_hip_rows_matches_lds_from_oracle_rows(monkeypatch):
    import oracle_backend as OB
    from gad.lds import masks_and_behaviours, shapley_lds
    from oracle import diffusers_ref as R
    from src.datasets import create_dataset, remove_data_by_datamodel, remove_data_by_shapley
    from src.ddpm_config import DDPMConfig
    monkeypatch.setenv("GAD_SYNTH_SCALE", "0.032")
    torch.set_num_threads(min(16, torch.get_num_threads()))               # the GPU box grants a 16-CPU share
    ds = create_dataset("cifar100", train=True)
    assert len(ds) == 320 and len(set(ds.targets)) == 20
    x_all = ds.device_tensor("cpu")
    group_of = {i: int(t) for i, t in enumerate(ds.targets)}
    cfg = DDPMConfig.cifar100_config
    ucfg = dict(cfg["unet_config"], block_out_channels=[32, 32, 64, 64], norm_num_groups=8)
    scfg = cfg["scheduler_config"]
    torch.manual_seed(0)
    base_sd = {k: v.clone() for k, v in R.UNet2DModel(**ucfg).state_dict().items()}

    def behaviour(kind, remaining, rng_seed):
        return OB.fid_against_dataset(_coalition_images(kind, base_sd, x_all, remaining, rng_seed, ucfg, scfg), ds, "cpu")

    def rows(kind, subsets, seed0):
        import sys
        import time
        t0 = time.time()
        out = [dict(removal_seed=k, remaining_idx=[int(i) for i in rem], fid_value=behaviour(kind, rem, seed0 + k))
               for k, rem in subsets]
        print(f"[lds toy] {kind}: {len(out)} coalitions in {time.time() - t0:.1f}s", file=sys.stderr, flush=True)
        return out
    fit = [(k, remove_data_by_shapley(ds, seed=k, by_class=True)[0]) for k in range(20)]
    tests = [(k, remove_data_by_datamodel(ds, alpha=0.5, seed=k, by_class=True)[0]) for k in range(12)]
    res = {}
    for kind in ("hip", "oracle"):
        tr_m, tr_y, _ = masks_and_behaviours(rows(kind, fit, 1000), group_of, 20)
        test_sets = []
        for s in (42, 43, 44):                                            # three "retraining seeds" of the same test subsets
            m, y, _ = masks_and_behaviours(rows(kind, tests, 100 * s), group_of, 20)
            test_sets.append((m, y))
        full = np.array([[behaviour(kind, np.arange(320), 7)]])
        null = np.array([[behaviour(kind, np.array([], dtype=int), 8)]])
        (lds, ci), attrs = shapley_lds(tr_m, tr_y, test_sets, full, null)
        res[kind] = dict(lds=lds, ci=ci, fit=tr_y[:, 0], test=np.concatenate([y[:, 0] for _, y in test_sets]), attrs=attrs[0])
    h, o = res["hip"], res["oracle"]
    print(f"LDS hip {h['lds']:.3f} ({h['ci']:.2f})  oracle {o['lds']:.3f} ({o['ci']:.2f})")
    # per-coalition behaviours: fp32 kernels vs fp32 CPU, a few DDIM steps, uint8 quantisation -> 0.5 % relative
    assert np.abs(h["fit"] - o["fit"]).max() < 5e-3 * np.abs(o["fit"]).max()
    assert np.abs(h["test"] - o["test"]).max() < 5e-3 * np.abs(o["test"]).max()
    assert np.isfinite(h["lds"]) and abs(h["lds"] - o["lds"]) <= 2.0            # +-0.02 on the correlation scale
    spread = np.abs(o["attrs"]).max()
    assert np.abs(h["attrs"] - o["attrs"]).max() < 0.05 * spread                # the attributions themselves agree

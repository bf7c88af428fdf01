"""Known-answer tests that pin the oracle's restatement of diffusers-0.24.0:
closed-form values only - no reference code can be run for these (diffusers is absent)."""
import math

import numpy as np
import torch

from oracle import diffusers_ref as R
from src.ddpm_config import DDPMConfig


def test_cifar_unet_param_count_and_names():
    net = R.UNet2DModel(**DDPMConfig.cifar100_config["unet_config"])
    assert sum(p.numel() for p in net.parameters()) == 35_746_307       # google/ddpm-cifar10-32
    sd = net.state_dict()
    for k in ("conv_in.weight", "time_embedding.linear_1.weight", "down_blocks.0.resnets.0.time_emb_proj.bias",
              "down_blocks.1.attentions.0.to_q.weight", "down_blocks.1.attentions.1.to_out.0.bias",
              "down_blocks.0.downsamplers.0.conv.weight", "mid_block.attentions.0.group_norm.weight",
              "up_blocks.2.attentions.2.to_v.weight", "up_blocks.0.upsamplers.0.conv.bias",
              "up_blocks.3.resnets.2.conv_shortcut.weight", "conv_norm_out.weight", "conv_out.bias"):
        assert k in sd, k
    assert sd["up_blocks.2.resnets.2.conv1.weight"].shape == (256, 384, 3, 3)
    assert sd["up_blocks.3.resnets.0.conv1.weight"].shape == (128, 384, 3, 3)
    assert "up_blocks.3.upsamplers.0.conv.weight" not in sd and "down_blocks.3.downsamplers.0.conv.weight" not in sd


def test_beta_alpha_tables():
    s = R.DDPMScheduler(**DDPMConfig.cifar100_config["scheduler_config"])
    assert s.betas.dtype == torch.float32 and len(s.betas) == 1000
    assert abs(s.betas[0].item() - 1e-4) < 1e-10 and abs(s.betas[-1].item() - 0.02) < 1e-8
    ac = np.cumprod(1.0 - np.linspace(1e-4, 0.02, 1000))
    np.testing.assert_allclose(s.alphas_cumprod.numpy(), ac, rtol=2e-5)
    assert abs(ac[-1] - 4.0358e-05) < 1e-8
    sl = R.DDIMScheduler(beta_start=0.0015, beta_end=0.0195, beta_schedule="scaled_linear", clip_sample=False)
    np.testing.assert_allclose(sl.betas.numpy(), np.linspace(0.0015 ** 0.5, 0.0195 ** 0.5, 1000) ** 2, rtol=1e-5)


def test_timestep_embedding_closed_form():
    t = torch.tensor([0, 1, 37, 999])
    e = R.get_timestep_embedding(t, 128, flip_sin_to_cos=False, downscale_freq_shift=1)
    assert e.shape == (4, 128)
    i = np.arange(64)
    f = np.exp(-math.log(10000) * i / (64 - 1))
    want = np.concatenate([np.sin(t.numpy()[:, None] * f), np.cos(t.numpy()[:, None] * f)], 1)
    np.testing.assert_allclose(e.numpy(), want, atol=2e-4)
    assert np.allclose(e[0, :64], 0) and np.allclose(e[0, 64:], 1)
    e2 = R.get_timestep_embedding(t, 224, flip_sin_to_cos=True, downscale_freq_shift=0)
    f2 = np.exp(-math.log(10000) * np.arange(112) / 112)
    np.testing.assert_allclose(e2[:, :112].numpy(), np.cos(t.numpy()[:, None] * f2), atol=2e-4)


def test_ddim_timesteps_and_step():
    s = R.DDIMScheduler()
    s.set_timesteps(100)
    assert s.timesteps.tolist() == list(range(990, -1, -10))
    s.set_timesteps(50)
    assert s.timesteps[:3].tolist() == [980, 960, 940] and s.timesteps[-1].item() == 0
    s.set_timesteps(100)
    x, e = torch.full((1, 1, 2, 2), 0.3), torch.full((1, 1, 2, 2), -0.2)
    for t in (990, 10, 0):
        a_t = s.alphas_cumprod[t].item()
        a_p = s.alphas_cumprod[t - 10].item() if t >= 10 else 1.0
        x0 = max(-1.0, min(1.0, (0.3 - math.sqrt(1 - a_t) * -0.2) / math.sqrt(a_t)))
        want = math.sqrt(a_p) * x0 + math.sqrt(1 - a_p) * -0.2
        assert abs(s.step(e, t, x).prev_sample[0, 0, 0, 0].item() - want) < 1e-6
    assert torch.allclose(s.step(e, 0, x).prev_sample, s.step(e, 0, x).pred_original_sample)   # a_prev = 1 at the last step


def test_ema_decay_sequence_and_update():
    p = torch.nn.Parameter(torch.ones(3))
    ema = R.EMAModel([p])
    seq = []
    for _ in range(5):
        ema.step([p])
        seq.append(ema.cur_decay_value)
    assert seq[0] == 0.0 and np.allclose(seq[1:], [2 / 11, 3 / 12, 4 / 13, 5 / 14])
    assert ema.get_decay(10 ** 9) == 0.9999
    ema2 = R.EMAModel([torch.nn.Parameter(torch.zeros(2))])
    ema2.optimization_step = 100
    q = torch.nn.Parameter(torch.ones(2))
    ema2.step([q])
    d = (1 + 100) / (10 + 100)
    assert torch.allclose(ema2.shadow_params[0], torch.full((2,), 1 - d))
    sd = ema2.state_dict()
    assert set(sd) == {"decay", "min_decay", "optimization_step", "update_after_step", "use_ema_warmup", "inv_gamma",
                       "power", "shadow_params"}


def test_antithetic_timesteps():
    t1 = torch.tensor([3, 500, 999])
    assert R.antithetic_timesteps(t1, 1000, 4).tolist() == [3, 500, 999, 996]
    assert R.antithetic_timesteps(t1, 1000, 5).tolist() == [3, 500, 999, 996, 499]


def test_pipeline_noise_is_cpu_generator_exact_and_deterministic():
    cfg = dict(DDPMConfig.cifar100_config["unet_config"], block_out_channels=[32, 32, 32, 32], norm_num_groups=8)
    torch.manual_seed(0)
    net = R.UNet2DModel(**cfg)
    pipe = R.DDPMPipeline(net, R.DDIMScheduler())
    a = pipe(batch_size=2, generator=torch.Generator().manual_seed(7), num_inference_steps=2).images
    b = pipe(batch_size=2, generator=torch.Generator().manual_seed(7), num_inference_steps=2).images
    assert a.shape == (2, 32, 32, 3) and a.min() >= 0 and a.max() <= 1 and np.array_equal(a, b)


def test_ddpm_ancestral_step_closed_form():
    """DDPMScheduler.step (off the reference's hot path; kept so DDPMPipeline(unet, DDPMScheduler()) works): with eps = the
    true noise the predicted x0 is exact, the mean follows the posterior q(x_{t-1} | x_t, x_0), the added noise has
    variance beta_tilde_t, and t = 0 adds none."""
    import torch
    import gad
    sch = gad.DDPMScheduler()
    torch.manual_seed(0)
    x0, eps = torch.rand(2, 3, 4, 4) * 1.6 - 0.8, torch.randn(2, 3, 4, 4)
    for t in (999, 500, 1, 0):
        a_t = sch.alphas_cumprod[t].double().item()
        a_p = sch.alphas_cumprod[t - 1].double().item() if t > 0 else 1.0
        beta = 1 - a_t / a_p
        xt = a_t ** 0.5 * x0.double() + (1 - a_t) ** 0.5 * eps.double()
        g = torch.Generator().manual_seed(5)
        out = sch.step(eps, t, xt.float(), generator=g).prev_sample.double()
        mean = (a_p ** 0.5 * beta / (1 - a_t)) * x0.double() + ((a_t / a_p) ** 0.5 * (1 - a_p) / (1 - a_t)) * xt
        if t == 0:
            assert torch.allclose(out, x0.double(), atol=1e-5)
        else:
            z = torch.randn(xt.shape, generator=torch.Generator().manual_seed(5)).double()
            var = (1 - a_p) / (1 - a_t) * beta
            assert torch.allclose(out, mean + var ** 0.5 * z, atol=2e-4), t
    sch.set_timesteps(10)
    assert sch.timesteps.tolist() == [900, 800, 700, 600, 500, 400, 300, 200, 100, 0]


def test_product_antithetic_timesteps_is_bit_exact_with_the_oracle():
    """gad.coalition.antithetic_timesteps itself (main.py:684-696), seeded generator, vs the oracle."""
    import torch
    from gad.coalition import antithetic_timesteps
    from oracle import diffusers_ref as R
    for B, N, seed in ((128, 1000, 0), (127, 1000, 1), (32, 50, 2), (1, 1000, 3), (2, 7, 4)):
        got = antithetic_timesteps(N, B, "cpu", generator=torch.Generator().manual_seed(seed))
        t1 = torch.randint(0, N, (B // 2 + 1,), generator=torch.Generator().manual_seed(seed)).long()
        want = R.antithetic_timesteps(t1, N, B)
        assert got.dtype == torch.int64 and torch.equal(got, want)
        assert torch.equal(got[B // 2 + 1:], (N - 1 - got[:B // 2 + 1])[:B - (B // 2 + 1)])

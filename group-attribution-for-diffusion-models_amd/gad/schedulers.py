"""DDPM / DDIM schedulers with the diffusers-0.24 constructor keys and methods the
reference calls (main.py:551,686,698; src/diffusion_utils.py:311,406).
Tables are float32 on the host exactly as diffusers builds them; the per-element work
(add_noise, the DDIM update) runs in one fused HIP kernel each."""
from __future__ import annotations

from types import SimpleNamespace

import numpy as np
import torch

from . import ops
from .nn import FrozenConfig


def _betas(beta_start, beta_end, n, schedule, trained_betas=None):
    if trained_betas is not None:
        return torch.tensor(trained_betas, dtype=torch.float32)
    if schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    raise NotImplementedError(f"beta_schedule={schedule}")


class _Base:
    def _tables(self, c):
        self.betas = _betas(c["beta_start"], c["beta_end"], c["num_train_timesteps"], c["beta_schedule"],
                            c.get("trained_betas"))
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self._ac_dev = {}
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None

    def _ac_on(self, device):
        key = (device.type, device.index)
        if key not in self._ac_dev:
            self._ac_dev[key] = self.alphas_cumprod.to(device)
        return self._ac_dev[key]

    def add_noise(self, original_samples, noise, timesteps):
        """x_t = sqrt(acp[t]) x_0 + sqrt(1-acp[t]) eps, per-sample t (fused gather + axpby kernel)."""
        x0 = original_samples.contiguous()
        return ops.add_noise_raw(x0, noise.contiguous(), timesteps.to(x0.device), self._ac_on(x0.device))

    def scale_model_input(self, sample, timestep=None):
        return sample


class DDPMScheduler(_Base):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0, sample_max_value=1.0,
                 timestep_spacing="leading", steps_offset=0, **unused):
        c = dict(locals())
        for k in ("self", "unused", "__class__"):
            c.pop(k, None)
        self.config = FrozenConfig(c)
        self._tables(self.config)
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def set_timesteps(self, num_inference_steps, device=None):
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64) + self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)

    def step(self, model_output, timestep, sample, generator=None, return_dict=True):
        """Ancestral DDPM update (diffusers DDPMScheduler.step, epsilon prediction, variance_type fixed_small / fixed_large):
        x0 = (x - sqrt(1-abar_t) eps) / sqrt(abar_t), clipped; mean = c0 x0 + c1 x; x_prev = mean + sigma z for t > 0.
        Off the reference's hot path - its samplers are DDIM (src/diffusion_utils.py:311,406) and this scheduler only
        supplies add_noise there - so it is plain torch on the tensors' device."""
        c = self.config
        if c.prediction_type != "epsilon" or c.thresholding or c.variance_type not in ("fixed_small", "fixed_large"):
            raise NotImplementedError("DDPMScheduler.step: epsilon prediction with fixed variance only")
        t = int(timestep)
        n = self.num_inference_steps or c.num_train_timesteps
        prev_t = t - c.num_train_timesteps // n
        ac = self.alphas_cumprod
        a_t = ac[t].item()
        a_p = ac[prev_t].item() if prev_t >= 0 else 1.0
        cur_alpha = a_t / a_p
        cur_beta = 1.0 - cur_alpha
        x0 = (sample - (1.0 - a_t) ** 0.5 * model_output) / a_t ** 0.5
        if c.clip_sample:
            x0 = x0.clamp(-c.clip_sample_range, c.clip_sample_range)
        mean = (a_p ** 0.5 * cur_beta / (1.0 - a_t)) * x0 + (cur_alpha ** 0.5 * (1.0 - a_p) / (1.0 - a_t)) * sample
        if t > 0:
            var = max((1.0 - a_p) / (1.0 - a_t) * cur_beta, 1e-20) if c.variance_type == "fixed_small" else cur_beta
            if generator is not None and generator.device.type == "cpu" and sample.is_cuda:
                z = torch.randn(sample.shape, generator=generator, dtype=sample.dtype).to(sample.device)
            else:
                z = torch.randn(sample.shape, generator=generator, dtype=sample.dtype, device=sample.device)
            mean = mean + var ** 0.5 * z
        return SimpleNamespace(prev_sample=mean)


def min_snr_weights(alphas_cumprod, timesteps, snr_gamma):
    """compute_snr + the epsilon-prediction weights of train_text_to_image_lora.py:1276-1290:
    SNR_t = abar_t / (1 - abar_t);  w_t = min(SNR_t, gamma) / SNR_t."""
    a = alphas_cumprod.to(timesteps.device)[timesteps].float()
    snr = a / (1.0 - a)
    return torch.minimum(snr, torch.full_like(snr, float(snr_gamma))) / snr


class DDIMScheduler(_Base):
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, **unused):
        c = dict(locals())
        for k in ("self", "unused", "__class__"):
            c.pop(k, None)
        self.config = FrozenConfig(c)
        if prediction_type != "epsilon" or thresholding or timestep_spacing != "leading" or rescale_betas_zero_snr:
            raise NotImplementedError("DDIMScheduler: only the reference's epsilon / leading configuration")
        self._tables(self.config)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    @classmethod
    def from_config(cls, config):
        return cls(**dict(config))

    def set_timesteps(self, num_inference_steps, device=None):
        if num_inference_steps > self.config.num_train_timesteps:
            raise ValueError("num_inference_steps > num_train_timesteps")
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts)          # kept on the host: the loop index is host data

    def step_coefficients(self, timestep):
        """(alpha_bar_t, alpha_bar_prev) as python floats holding the float32 table values."""
        t = int(timestep)
        prev = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t].item()
        a_p = self.alphas_cumprod[prev].item() if prev >= 0 else self.final_alpha_cumprod.item()
        return a_t, a_p

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False, generator=None,
             variance_noise=None, return_dict=True, out=None):
        if eta != 0.0 or use_clipped_model_output:
            raise NotImplementedError("DDIMScheduler.step: the reference samples with eta=0 "
                                      "(src/diffusion_utils.py:336-341)")
        a_t, a_p = self.step_coefficients(timestep)
        clip = float(self.config.clip_sample_range) if self.config.clip_sample else 0.0
        prev = ops.ddim_step_raw(sample.contiguous(), model_output.contiguous(), a_t, a_p, clip, out=out)
        return SimpleNamespace(prev_sample=prev)

"""Pure-PyTorch (CPU) restatement of the diffusers-0.24.0 objects the reference
calls on its hot path.  TEST INFRASTRUCTURE (see oracle/__init__.py).

The reference imports these from the un-vendored dependency diffusers==0.24.0
(reference requirements.txt:5); call sites:
  unconditional_generation/main.py:234,332 (model ctor), :550-552 (pipeline),
  :698 (add_noise), :707 (forward), :725 (EMA step);
  src/diffusion_utils.py:311,336-341 (DDPMPipeline + DDIMScheduler sampling);
  src/diffusers/models/attention_processor.py:1256-1341 (attention semantics,
  the one vendored file - followed line by line in ``Attention.forward``).
Every class keeps the diffusers state_dict key names so reference checkpoints
(`ckpt_steps_*.pt`, main.py:827-840) load.

Everything here is stock torch ops on CPU tensors (any float dtype: run the
module `.double()` for an fp64 reference).
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------------------
# embeddings (diffusers.models.embeddings.get_timestep_embedding)
# ----------------------------------------------------------------------------
def get_timestep_embedding(timesteps, dim, flip_sin_to_cos=False, downscale_freq_shift=1.0,
                           scale=1.0, max_period=10000):
    half = dim // 2
    exponent = -math.log(max_period) * torch.arange(0, half, dtype=torch.float32, device=timesteps.device)
    exponent = exponent / (half - downscale_freq_shift)
    emb = torch.exp(exponent)
    emb = timesteps[:, None].float() * emb[None, :]
    emb = scale * emb
    emb = torch.cat([torch.sin(emb), torch.cos(emb)], dim=-1)
    if flip_sin_to_cos:
        emb = torch.cat([emb[:, half:], emb[:, :half]], dim=-1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


class Timesteps(nn.Module):
    def __init__(self, num_channels, flip_sin_to_cos, downscale_freq_shift):
        super().__init__()
        self.num_channels = num_channels
        self.flip_sin_to_cos = flip_sin_to_cos
        self.downscale_freq_shift = downscale_freq_shift

    def forward(self, t):
        return get_timestep_embedding(t, self.num_channels, self.flip_sin_to_cos, self.downscale_freq_shift)


class TimestepEmbedding(nn.Module):
    def __init__(self, in_channels, time_embed_dim):
        super().__init__()
        self.linear_1 = nn.Linear(in_channels, time_embed_dim)
        self.act = nn.SiLU()
        self.linear_2 = nn.Linear(time_embed_dim, time_embed_dim)

    def forward(self, x):
        return self.linear_2(self.act(self.linear_1(x)))


# ----------------------------------------------------------------------------
# LoRA (diffusers.models.lora)
# ----------------------------------------------------------------------------
class LoRALinearLayer(nn.Module):
    def __init__(self, in_features, out_features, rank=4, network_alpha=None):
        super().__init__()
        self.down = nn.Linear(in_features, rank, bias=False)
        self.up = nn.Linear(rank, out_features, bias=False)
        self.network_alpha = network_alpha
        self.rank = rank
        self.in_features, self.out_features = in_features, out_features
        nn.init.normal_(self.down.weight, std=1 / rank)
        nn.init.zeros_(self.up.weight)

    def forward(self, x):
        orig = x.dtype
        y = self.up(self.down(x.to(self.down.weight.dtype)))
        if self.network_alpha is not None:
            y = y * (self.network_alpha / self.rank)
        return y.to(orig)


class LoRACompatibleLinear(nn.Linear):
    def __init__(self, *a, lora_layer=None, **k):
        super().__init__(*a, **k)
        self.lora_layer = lora_layer

    def set_lora_layer(self, lora_layer):
        self.lora_layer = lora_layer

    def forward(self, x, scale: float = 1.0):
        if self.lora_layer is None:
            return super().forward(x)
        return super().forward(x) + scale * self.lora_layer(x)


# ----------------------------------------------------------------------------
# blocks
# ----------------------------------------------------------------------------
class ResnetBlock2D(nn.Module):
    def __init__(self, in_channels, out_channels, temb_channels=512, groups=32, eps=1e-6,
                 output_scale_factor=1.0):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, padding=1)
        self.time_emb_proj = nn.Linear(temb_channels, out_channels)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.output_scale_factor = output_scale_factor

    def forward(self, x, temb):
        h = self.conv1(F.silu(self.norm1(x)))
        h = h + self.time_emb_proj(F.silu(temb))[:, :, None, None]
        h = self.conv2(F.silu(self.norm2(h)))  # dropout p=0
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return (x + h) / self.output_scale_factor


class Attention(nn.Module):
    """Self-attention block as executed by AttnProcessor2_0
    (reference src/diffusers/models/attention_processor.py:1265-1341)."""

    def __init__(self, query_dim, heads, dim_head, eps, norm_num_groups, rescale_output_factor=1.0,
                 residual_connection=True, bias=True):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.rescale_output_factor = rescale_output_factor
        self.residual_connection = residual_connection
        self.group_norm = nn.GroupNorm(norm_num_groups, query_dim, eps=eps, affine=True) \
            if norm_num_groups is not None else None
        self.to_q = LoRACompatibleLinear(query_dim, inner, bias=bias)
        self.to_k = LoRACompatibleLinear(query_dim, inner, bias=bias)
        self.to_v = LoRACompatibleLinear(query_dim, inner, bias=bias)
        self.to_out = nn.ModuleList([LoRACompatibleLinear(inner, query_dim, bias=True), nn.Dropout(0.0)])

    def forward(self, x, scale: float = 1.0):
        residual = x                                                 # :1273
        b, c, hh, ww = x.shape
        h = x.view(b, c, hh * ww).transpose(1, 2)                    # :1283-1285
        if self.group_norm is not None:
            h = self.group_norm(h.transpose(1, 2)).transpose(1, 2)   # :1297-1298
        q = self.to_q(h, scale)                                      # :1301
        k = self.to_k(h, scale)                                      # :1308
        v = self.to_v(h, scale)                                      # :1309
        d = k.shape[-1] // self.heads
        q = q.view(b, -1, self.heads, d).transpose(1, 2)             # :1314-1317
        k = k.view(b, -1, self.heads, d).transpose(1, 2)
        v = v.view(b, -1, self.heads, d).transpose(1, 2)
        # explicit softmax(q k^T / sqrt(d)) v == F.scaled_dot_product_attention (:1321-1323)
        w = torch.softmax(q @ k.transpose(-1, -2) / math.sqrt(d), dim=-1)
        h = (w @ v).transpose(1, 2).reshape(b, -1, self.heads * d)   # :1325
        h = self.to_out[0](h, scale)                                 # :1329
        h = h.transpose(-1, -2).reshape(b, c, hh, ww)                # :1334
        if self.residual_connection:
            h = h + residual                                         # :1336-1337
        return h / self.rescale_output_factor                        # :1339


class Downsample2D(nn.Module):
    def __init__(self, channels, padding):
        super().__init__()
        self.padding = padding
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=padding)

    def forward(self, x):
        if self.padding == 0:
            x = F.pad(x, (0, 1, 0, 1), mode="constant", value=0)
        return self.conv(x)


class Upsample2D(nn.Module):
    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        return self.conv(F.interpolate(x, scale_factor=2.0, mode="nearest"))


class DownBlock(nn.Module):
    """DownBlock2D / AttnDownBlock2D."""

    def __init__(self, in_c, out_c, temb_c, num_layers, eps, groups, add_downsample, downsample_padding,
                 attn=None):
        super().__init__()
        resnets = nn.ModuleList([
            ResnetBlock2D(in_c if i == 0 else out_c, out_c, temb_c, groups, eps) for i in range(num_layers)])
        # diffusers registers `attentions` before `resnets` in Attn*Block2D (parameters() order -> EMA list order)
        if attn is not None:                       # (heads, dim_head)
            self.attentions = nn.ModuleList([
                Attention(out_c, attn[0], attn[1], eps, groups) for _ in range(num_layers)])
        else:
            self.attentions = None
        self.resnets = resnets
        self.downsamplers = nn.ModuleList([Downsample2D(out_c, downsample_padding)]) if add_downsample else None

    def forward(self, h, temb):
        outs = ()
        for i, r in enumerate(self.resnets):
            h = r(h, temb)
            if self.attentions is not None:
                h = self.attentions[i](h)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs += (h,)
        return h, outs


class UNetMidBlock2D(nn.Module):
    def __init__(self, c, temb_c, eps, groups, attn, add_attention=True):
        super().__init__()
        resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_c, groups, eps),
                                 ResnetBlock2D(c, c, temb_c, groups, eps)])
        self.attentions = nn.ModuleList([
            Attention(c, attn[0], attn[1], eps, groups) if add_attention else None])
        self.resnets = resnets

    def forward(self, h, temb):
        h = self.resnets[0](h, temb)
        if self.attentions[0] is not None:
            h = self.attentions[0](h)
        return self.resnets[1](h, temb)


class UpBlock(nn.Module):
    """UpBlock2D / AttnUpBlock2D."""

    def __init__(self, in_c, prev_c, out_c, temb_c, num_layers, eps, groups, add_upsample, attn=None):
        super().__init__()
        res = []
        for i in range(num_layers):
            skip_c = in_c if i == num_layers - 1 else out_c
            r_in = prev_c if i == 0 else out_c
            res.append(ResnetBlock2D(r_in + skip_c, out_c, temb_c, groups, eps))
        if attn is not None:
            self.attentions = nn.ModuleList([
                Attention(out_c, attn[0], attn[1], eps, groups) for _ in range(num_layers)])
        else:
            self.attentions = None
        self.resnets = nn.ModuleList(res)
        self.upsamplers = nn.ModuleList([Upsample2D(out_c)]) if add_upsample else None

    def forward(self, h, skips, temb):
        for i, r in enumerate(self.resnets):
            s = skips[-1]
            skips = skips[:-1]
            h = r(torch.cat([h, s], dim=1), temb)
            if self.attentions is not None:
                h = self.attentions[i](h)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class UNet2DModel(nn.Module):
    """diffusers.UNet2DModel restated.  Accepts every key
    of the reference config dicts (src/ddpm_config.py:235-269 / :423-451)."""

    def __init__(self, sample_size=None, in_channels=3, out_channels=3, center_input_sample=False,
                 time_embedding_type="positional", freq_shift=0, flip_sin_to_cos=True,
                 down_block_types=("DownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D", "AttnDownBlock2D"),
                 up_block_types=("AttnUpBlock2D", "AttnUpBlock2D", "AttnUpBlock2D", "UpBlock2D"),
                 block_out_channels=(224, 448, 672, 896), layers_per_block=2, mid_block_scale_factor=1,
                 downsample_padding=1, downsample_type="conv", upsample_type="conv", dropout=0.0,
                 act_fn="silu", attention_head_dim=8, norm_num_groups=32, attn_norm_num_groups=None,
                 norm_eps=1e-5, resnet_time_scale_shift="default", add_attention=True,
                 class_embed_type=None, num_class_embeds=None, num_train_timesteps=None, attention_layout=None,
                 **unused):
        super().__init__()
        cfg = dict(locals())
        for k in ("self", "unused", "__class__"):
            cfg.pop(k, None)
        cfg.update(unused)
        self.config = SimpleNamespace(**cfg)
        assert time_embedding_type == "positional" and act_fn == "silu" and class_embed_type is None
        assert downsample_type == "conv" and upsample_type == "conv" and resnet_time_scale_shift == "default"
        boc = list(block_out_channels)
        temb_c = boc[0] * 4
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_proj = Timesteps(boc[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(boc[0], temb_c)
        self.down_blocks = nn.ModuleList()
        def attn_of(level, c):          # (heads, dim_head); attention_layout: head-grouped pruned models (prune.py:337-342)
            if attention_layout is not None:
                return int(attention_layout[level][0]), int(attention_layout[level][1])
            hd_ = attention_head_dim if attention_head_dim is not None else c
            return c // hd_, hd_
        out_c = boc[0]
        for i, typ in enumerate(down_block_types):
            in_c, out_c = out_c, boc[i]
            final = i == len(boc) - 1
            hd = None
            if typ == "AttnDownBlock2D":
                hd = attn_of(i, out_c)
            elif typ != "DownBlock2D":
                raise ValueError(typ)
            self.down_blocks.append(DownBlock(in_c, out_c, temb_c, layers_per_block, norm_eps, norm_num_groups,
                                              not final, downsample_padding, hd))
        self.mid_block = UNetMidBlock2D(boc[-1], temb_c, norm_eps, norm_num_groups, attn_of(len(boc) - 1, boc[-1]),
                                        add_attention)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_c = rev[0]
        for i, typ in enumerate(up_block_types):
            prev_c, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            final = i == len(boc) - 1
            hd = None
            if typ == "AttnUpBlock2D":
                hd = attn_of(len(boc) - 1 - i, out_c)
            elif typ != "UpBlock2D":
                raise ValueError(typ)
            self.up_blocks.append(UpBlock(in_c, prev_c, out_c, temb_c, layers_per_block + 1, norm_eps,
                                          norm_num_groups, not final, hd))
        g = norm_num_groups if norm_num_groups is not None else min(boc[0] // 4, 32)
        self.conv_norm_out = nn.GroupNorm(g, boc[0], eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    @property
    def device(self):
        return self.conv_in.weight.device

    def forward(self, sample, timestep):
        if self.config.center_input_sample:
            sample = 2 * sample - 1.0
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=sample.device)
        elif t.ndim == 0:
            t = t[None].to(sample.device)
        t = t * torch.ones(sample.shape[0], dtype=t.dtype, device=t.device)
        emb = self.time_embedding(self.time_proj(t).to(self.dtype))
        h = self.conv_in(sample)
        skips = (h,)
        for blk in self.down_blocks:
            h, outs = blk(h, emb)
            skips += outs
        h = self.mid_block(h, emb)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            res, skips = skips[-n:], skips[:-n]
            h = blk(h, res, emb)
        h = self.conv_out(self.conv_act(self.conv_norm_out(h)))
        return SimpleNamespace(sample=h)


# ----------------------------------------------------------------------------
# schedulers
# ----------------------------------------------------------------------------
def _make_betas(beta_start, beta_end, n, schedule, trained_betas=None):
    if trained_betas is not None:
        return torch.tensor(trained_betas, dtype=torch.float32)
    if schedule == "linear":
        return torch.linspace(beta_start, beta_end, n, dtype=torch.float32)
    if schedule == "scaled_linear":
        return torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n, dtype=torch.float32) ** 2
    raise NotImplementedError(schedule)


class DDPMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, variance_type="fixed_small", clip_sample=True, prediction_type="epsilon",
                 thresholding=False, dynamic_thresholding_ratio=0.995, clip_sample_range=1.0,
                 sample_max_value=1.0, timestep_spacing="leading", steps_offset=0, **unused):
        cfg = dict(locals())
        for k in ("self", "unused"):
            cfg.pop(k)
        self.config = SimpleNamespace(**cfg)
        self.betas = _make_betas(beta_start, beta_end, num_train_timesteps, beta_schedule, trained_betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.one = torch.tensor(1.0)
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy())

    def add_noise(self, original_samples, noise, timesteps):
        ac = self.alphas_cumprod.to(device=original_samples.device, dtype=original_samples.dtype)
        timesteps = timesteps.to(original_samples.device)
        sa = ac[timesteps] ** 0.5
        sb = (1 - ac[timesteps]) ** 0.5
        sa = sa.flatten()
        sb = sb.flatten()
        while sa.ndim < original_samples.ndim:
            sa = sa.unsqueeze(-1)
            sb = sb.unsqueeze(-1)
        return sa * original_samples + sb * noise


class DDIMScheduler:
    def __init__(self, num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, beta_schedule="linear",
                 trained_betas=None, clip_sample=True, set_alpha_to_one=True, steps_offset=0,
                 prediction_type="epsilon", thresholding=False, dynamic_thresholding_ratio=0.995,
                 clip_sample_range=1.0, sample_max_value=1.0, timestep_spacing="leading",
                 rescale_betas_zero_snr=False, **unused):
        cfg = dict(locals())
        for k in ("self", "unused"):
            cfg.pop(k)
        self.config = SimpleNamespace(**cfg)
        assert prediction_type == "epsilon" and not thresholding and timestep_spacing == "leading"
        self.betas = _make_betas(beta_start, beta_end, num_train_timesteps, beta_schedule, trained_betas)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.final_alpha_cumprod = torch.tensor(1.0) if set_alpha_to_one else self.alphas_cumprod[0]
        self.init_noise_sigma = 1.0
        self.num_inference_steps = None
        self.timesteps = torch.from_numpy(np.arange(0, num_train_timesteps)[::-1].copy().astype(np.int64))

    def set_timesteps(self, num_inference_steps, device=None):
        self.num_inference_steps = num_inference_steps
        ratio = self.config.num_train_timesteps // num_inference_steps
        ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
        ts += self.config.steps_offset
        self.timesteps = torch.from_numpy(ts).to(device)

    add_noise = DDPMScheduler.add_noise

    def step(self, model_output, timestep, sample, eta=0.0, use_clipped_model_output=False,
             generator=None, variance_noise=None):
        t = int(timestep)
        prev_t = t - self.config.num_train_timesteps // self.num_inference_steps
        a_t = self.alphas_cumprod[t]
        a_p = self.alphas_cumprod[prev_t] if prev_t >= 0 else self.final_alpha_cumprod
        b_t = 1 - a_t
        x0 = (sample - b_t ** 0.5 * model_output) / a_t ** 0.5
        eps = model_output
        if self.config.clip_sample:
            x0 = x0.clamp(-self.config.clip_sample_range, self.config.clip_sample_range)
        var = ((1 - a_p) / (1 - a_t)) * (1 - a_t / a_p)
        std = eta * var ** 0.5
        if use_clipped_model_output:
            eps = (sample - a_t ** 0.5 * x0) / b_t ** 0.5
        direction = (1 - a_p - std ** 2) ** 0.5 * eps
        prev = a_p ** 0.5 * x0 + direction
        if eta > 0:
            if variance_noise is None:
                variance_noise = torch.randn(model_output.shape, generator=generator, dtype=model_output.dtype)
            prev = prev + std * variance_noise
        return SimpleNamespace(prev_sample=prev, pred_original_sample=x0)


# ----------------------------------------------------------------------------
# pipelines
# ----------------------------------------------------------------------------
class DDPMPipeline:
    def __init__(self, unet, scheduler):
        self.unet, self.scheduler = unet, scheduler
        self.device = torch.device("cpu")

    def to(self, device):
        self.device = torch.device(device)
        self.unet.to(device)
        return self

    @torch.no_grad()
    def __call__(self, batch_size=1, generator=None, num_inference_steps=1000, output_type="numpy", eta=0.0):
        ss = self.unet.config.sample_size
        shape = (batch_size, self.unet.config.in_channels, ss, ss)
        # randn_tensor: a CPU generator draws on CPU, then the tensor is moved.
        image = torch.randn(shape, generator=generator, dtype=torch.float32).to(self.device, self.unet.dtype)
        self.scheduler.set_timesteps(num_inference_steps)
        for t in self.scheduler.timesteps:
            out = self.unet(image, t).sample
            image = self.scheduler.step(out, t, image, eta=eta, generator=generator).prev_sample
        image = (image / 2 + 0.5).clamp(0, 1)
        image = image.cpu().permute(0, 2, 3, 1).float().numpy()
        return SimpleNamespace(images=image)


DDIMPipeline = DDPMPipeline  # identical loop for eta=0


# ----------------------------------------------------------------------------
# EMA (diffusers.training_utils.EMAModel)
# ----------------------------------------------------------------------------
class EMAModel:
    def __init__(self, parameters, decay=0.9999, min_decay=0.0, update_after_step=0, use_ema_warmup=False,
                 inv_gamma=1.0, power=2 / 3, model_cls=None, model_config=None):
        self.shadow_params = [p.clone().detach() for p in parameters]
        self.temp_stored_params = None
        self.decay, self.min_decay, self.update_after_step = decay, min_decay, update_after_step
        self.use_ema_warmup, self.inv_gamma, self.power = use_ema_warmup, inv_gamma, power
        self.optimization_step = 0
        self.cur_decay_value = None

    def get_decay(self, optimization_step):
        step = max(0, optimization_step - self.update_after_step - 1)
        if step <= 0:
            return 0.0
        if self.use_ema_warmup:
            cur = 1 - (1 + step / self.inv_gamma) ** -self.power
        else:
            cur = (1 + step) / (10 + step)
        return max(min(cur, self.decay), self.min_decay)

    @torch.no_grad()
    def step(self, parameters):
        parameters = list(parameters)
        self.optimization_step += 1
        decay = self.get_decay(self.optimization_step)
        self.cur_decay_value = decay
        one_minus = 1 - decay
        for s, p in zip(self.shadow_params, parameters):
            if p.requires_grad:
                s.sub_(one_minus * (s - p))
            else:
                s.copy_(p)

    def copy_to(self, parameters):
        for s, p in zip(self.shadow_params, list(parameters)):
            p.data.copy_(s.to(p.device).data)

    def store(self, parameters):
        self.temp_stored_params = [p.detach().cpu().clone() for p in parameters]

    def restore(self, parameters):
        for c, p in zip(self.temp_stored_params, parameters):
            p.data.copy_(c.data)
        self.temp_stored_params = None

    def to(self, device=None, dtype=None):
        self.shadow_params = [p.to(device=device, dtype=dtype) if p.is_floating_point() else p.to(device=device)
                              for p in self.shadow_params]

    def state_dict(self):
        return {"decay": self.decay, "min_decay": self.min_decay, "optimization_step": self.optimization_step,
                "update_after_step": self.update_after_step, "use_ema_warmup": self.use_ema_warmup,
                "inv_gamma": self.inv_gamma, "power": self.power, "shadow_params": self.shadow_params}

    def load_state_dict(self, sd):
        for k in ("decay", "min_decay", "optimization_step", "update_after_step", "use_ema_warmup",
                  "inv_gamma", "power"):
            setattr(self, k, sd.get(k, getattr(self, k)))
        sp = sd.get("shadow_params", None)
        if sp is not None:
            self.shadow_params = [p.clone().detach() for p in sp]


# ----------------------------------------------------------------------------
# the training step body (reference unconditional_generation/main.py:681-725,
# unlearn.py:588-636) restated with explicit inputs so it is deterministic.
# ----------------------------------------------------------------------------
def antithetic_timesteps(t_half: torch.Tensor, n_train: int, batch: int) -> torch.Tensor:
    """main.py:684-696: t = cat([t1, N - t1 - 1])[:B] with t1 of length B//2+1."""
    return torch.cat([t_half, n_train - t_half - 1], dim=0)[:batch]


def train_step(model, optimizer, ema: Optional[EMAModel], scheduler: DDPMScheduler,
               image, noise, timesteps, max_norm=1.0):
    """One fwd+bwd+clip+Adam+EMA step on given (image, noise, t). Returns (loss, grad_norm)."""
    model.train()
    noisy = scheduler.add_noise(image, noise, timesteps)            # main.py:698
    optimizer.zero_grad()
    eps = model(noisy, timesteps).sample                            # :707
    loss = F.mse_loss(eps, noise)                                   # :708
    loss.backward()                                                 # :713
    gn = torch.nn.utils.clip_grad_norm_(model.parameters(), max_norm)  # :718
    optimizer.step()                                                # :719
    if ema is not None:
        ema.step(model.parameters())                                # :725
    return loss.detach(), gn

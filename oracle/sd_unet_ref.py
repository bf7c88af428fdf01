"""Pure-PyTorch (CPU) restatement of diffusers-0.24.0 `UNet2DConditionModel` as used by the Stable-Diffusion
LoRA path (reference text_to_image/train_text_to_image_lora.py:720-820,1268-1270; topology per SURVEY
Appendix A.15).  TEST INFRASTRUCTURE (see oracle/__init__.py).  Parity status: "parity unpinned" by the
reference (diffusers is un-vendored and absent; the SD weights are hub-fetched); restated from the published
architecture, pinned by the parameter count of SD-1.x (859 520 964) and state_dict key names."""
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle.diffusers_ref import (Downsample2D, LoRACompatibleLinear, ResnetBlock2D, Timesteps, TimestepEmbedding,
                                  Upsample2D)


class CrossAttention(nn.Module):
    """diffusers Attention inside BasicTransformerBlock: no group norm, no internal residual, q/k/v without bias."""

    def __init__(self, query_dim, context_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = LoRACompatibleLinear(query_dim, inner, bias=False)
        self.to_k = LoRACompatibleLinear(context_dim or query_dim, inner, bias=False)
        self.to_v = LoRACompatibleLinear(context_dim or query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([LoRACompatibleLinear(inner, query_dim, bias=True), nn.Dropout(0.0)])

    def forward(self, x, context=None, scale=1.0):
        ctx = x if context is None else context
        b, t, _ = x.shape
        q, k, v = self.to_q(x, scale), self.to_k(ctx, scale), self.to_v(ctx, scale)
        d = q.shape[-1] // self.heads

        def split(z):
            return z.view(b, -1, self.heads, d).transpose(1, 2)
        o = F.scaled_dot_product_attention(split(q), split(k), split(v))
        return self.to_out[0](o.transpose(1, 2).reshape(b, t, -1), scale)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = nn.Linear(dim, inner * 2)

    def forward(self, x):
        h, gate = self.proj(x).chunk(2, dim=-1)
        return h * F.gelu(gate)


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), nn.Linear(dim * 4, dim)])

    def forward(self, x):
        for m in self.net:
            x = m(x)
        return x


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, context_dim):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn1 = CrossAttention(dim, None, heads, dim_head)
        self.norm2 = nn.LayerNorm(dim)
        self.attn2 = CrossAttention(dim, context_dim, heads, dim_head)
        self.norm3 = nn.LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context):
        x = self.attn1(self.norm1(x)) + x
        x = self.attn2(self.norm2(x), context) + x
        return self.ff(self.norm3(x)) + x


class Transformer2DModel(nn.Module):
    def __init__(self, channels, heads, dim_head, context_dim, groups=32):
        super().__init__()
        self.norm = nn.GroupNorm(groups, channels, eps=1e-6, affine=True)
        self.proj_in = nn.Conv2d(channels, channels, 1)
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(channels, heads, dim_head, context_dim)])
        self.proj_out = nn.Conv2d(channels, channels, 1)

    def forward(self, x, context):
        b, c, h, w = x.shape
        res = x
        y = self.proj_in(self.norm(x)).permute(0, 2, 3, 1).reshape(b, h * w, c)
        for blk in self.transformer_blocks:
            y = blk(y, context)
        y = y.reshape(b, h, w, c).permute(0, 3, 1, 2).contiguous()
        return self.proj_out(y) + res


class CrossAttnDownBlock2D(nn.Module):
    def __init__(self, cin, cout, temb_c, layers, eps, groups, heads, context_dim, add_down, cross=True):
        super().__init__()
        attns = nn.ModuleList([Transformer2DModel(cout, heads, cout // heads, context_dim, groups) for _ in range(layers)]) \
            if cross else None
        resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_c, groups, eps) for i in range(layers)])
        if cross:
            self.attentions = attns
        self.resnets = resnets
        self.cross = cross
        self.downsamplers = nn.ModuleList([Downsample2D(cout, 1)]) if add_down else None

    def forward(self, h, temb, context):
        outs = ()
        for i, r in enumerate(self.resnets):
            h = r(h, temb)
            if self.cross:
                h = self.attentions[i](h, context)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs += (h,)
        return h, outs


class UNetMidBlock2DCrossAttn(nn.Module):
    def __init__(self, c, temb_c, eps, groups, heads, context_dim):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, c // heads, context_dim, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_c, groups, eps), ResnetBlock2D(c, c, temb_c, groups, eps)])

    def forward(self, h, temb, context):
        h = self.resnets[0](h, temb)
        h = self.attentions[0](h, context)
        return self.resnets[1](h, temb)


class CrossAttnUpBlock2D(nn.Module):
    def __init__(self, cin, prev_c, cout, temb_c, layers, eps, groups, heads, context_dim, add_up, cross=True):
        super().__init__()
        if cross:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cout // heads, context_dim, groups)
                                             for _ in range(layers)])
        res = []
        for i in range(layers):
            skip_c = cin if i == layers - 1 else cout
            r_in = prev_c if i == 0 else cout
            res.append(ResnetBlock2D(r_in + skip_c, cout, temb_c, groups, eps))
        self.resnets = nn.ModuleList(res)
        self.cross = cross
        self.upsamplers = nn.ModuleList([Upsample2D(cout)]) if add_up else None

    def forward(self, h, skips, temb, context):
        for i, r in enumerate(self.resnets):
            s, skips = skips[-1], skips[:-1]
            h = r(torch.cat([h, s], dim=1), temb)
            if self.cross:
                h = self.attentions[i](h, context)
        if self.upsamplers is not None:
            h = self.upsamplers[0](h)
        return h


class UNet2DConditionModel(nn.Module):
    def __init__(self, sample_size=32, in_channels=4, out_channels=4, center_input_sample=False, flip_sin_to_cos=True,
                 freq_shift=0, down_block_types=("CrossAttnDownBlock2D",) * 3 + ("DownBlock2D",),
                 up_block_types=("UpBlock2D",) + ("CrossAttnUpBlock2D",) * 3, block_out_channels=(320, 640, 1280, 1280),
                 layers_per_block=2, downsample_padding=1, mid_block_scale_factor=1, act_fn="silu", norm_num_groups=32,
                 norm_eps=1e-5, cross_attention_dim=768, attention_head_dim=8, **unused):
        super().__init__()
        cfg = dict(locals())
        for k in ("self", "unused", "__class__"):
            cfg.pop(k, None)
        cfg.update(unused)
        self.config = SimpleNamespace(**cfg)
        boc = list(block_out_channels)
        temb_c = boc[0] * 4
        heads = attention_head_dim            # SD-1.x: `attention_head_dim` is the number of heads
        self.conv_in = nn.Conv2d(in_channels, boc[0], 3, padding=1)
        self.time_proj = Timesteps(boc[0], flip_sin_to_cos, freq_shift)
        self.time_embedding = TimestepEmbedding(boc[0], temb_c)
        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i, typ in enumerate(down_block_types):
            in_c, out_c = out_c, boc[i]
            self.down_blocks.append(CrossAttnDownBlock2D(in_c, out_c, temb_c, layers_per_block, norm_eps, norm_num_groups,
                                                         heads, cross_attention_dim, i != len(boc) - 1,
                                                         cross=typ == "CrossAttnDownBlock2D"))
        self.mid_block = UNetMidBlock2DCrossAttn(boc[-1], temb_c, norm_eps, norm_num_groups, heads, cross_attention_dim)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_c = rev[0]
        for i, typ in enumerate(up_block_types):
            prev_c, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(CrossAttnUpBlock2D(in_c, prev_c, out_c, temb_c, layers_per_block + 1, norm_eps,
                                                     norm_num_groups, heads, cross_attention_dim, i != len(boc) - 1,
                                                     cross=typ == "CrossAttnUpBlock2D"))
        self.conv_norm_out = nn.GroupNorm(norm_num_groups, boc[0], eps=norm_eps)
        self.conv_act = nn.SiLU()
        self.conv_out = nn.Conv2d(boc[0], out_channels, 3, padding=1)

    @property
    def dtype(self):
        return self.conv_in.weight.dtype

    def forward(self, sample, timestep, encoder_hidden_states):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long)
        elif t.ndim == 0:
            t = t[None]
        t = t.expand(sample.shape[0])
        emb = self.time_embedding(self.time_proj(t).to(self.dtype))
        h = self.conv_in(sample)
        skips = (h,)
        for blk in self.down_blocks:
            h, outs = blk(h, emb, encoder_hidden_states)
            skips += outs
        h = self.mid_block(h, emb, encoder_hidden_states)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            res, skips = skips[-n:], skips[:-n]
            h = blk(h, res, emb, encoder_hidden_states)
        return SimpleNamespace(sample=self.conv_out(self.conv_act(self.conv_norm_out(h))))

"""`UNet2DConditionModel` (Stable-Diffusion 1.x / miniSD topology) on the HIP kernels, with
the LoRA plumbing the reference trainer uses (text_to_image/train_text_to_image_lora.py:776-853,1459):
`set_lora_layer` on to_q/to_k/to_v/to_out[0] of all 32 attention modules, per-projection (ragged) ranks as
produced by text_to_image/prune_lora.py:173-180, `save_attn_procs` / `load_attn_procs` with diffusers key names.
Activations NHWC, so the [B, HW, C] token view of a feature map is free."""
from __future__ import annotations

import os
from types import SimpleNamespace

import torch
import torch.nn as nn

from . import ops
from .nn import Conv2d, FrozenConfig, GroupNorm, Linear, LoRALinearLayer, ResnetBlock2D, TimestepEmbedding, _Sampler


class LayerNorm(nn.Module):
    def __init__(self, dim, eps=1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)

    def with_bypass(self, x):
        return ops.layer_norm_bypass(x, self.weight, self.bias, self.eps)


class CrossAttention(nn.Module):
    def __init__(self, query_dim, context_dim, heads, dim_head):
        super().__init__()
        inner = heads * dim_head
        self.heads = heads
        self.to_q = Linear(query_dim, inner, bias=False)
        self.to_k = Linear(context_dim or query_dim, inner, bias=False)
        self.to_v = Linear(context_dim or query_dim, inner, bias=False)
        self.to_out = nn.ModuleList([Linear(inner, query_dim, bias=True), nn.Dropout(0.0)])

    def forward(self, x, context=None, residual=None, scale: float = 1.0):
        ctx = x if context is None else context
        q, k, v = self.to_q(x, scale=scale), self.to_k(ctx, scale=scale), self.to_v(ctx, scale=scale)
        return self.to_out[0](ops.attention_core(q, k, v, self.heads), residual=residual, scale=scale)


class GEGLU(nn.Module):
    def __init__(self, dim, inner):
        super().__init__()
        self.proj = Linear(dim, inner * 2)

    def forward(self, x):
        return ops.geglu(self.proj(x))


class FeedForward(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.net = nn.ModuleList([GEGLU(dim, dim * 4), nn.Dropout(0.0), Linear(dim * 4, dim)])

    def forward(self, x, residual=None):
        return self.net[2](self.net[0](x), residual=residual)


class BasicTransformerBlock(nn.Module):
    def __init__(self, dim, heads, dim_head, context_dim):
        super().__init__()
        self.norm1 = LayerNorm(dim)
        self.attn1 = CrossAttention(dim, None, heads, dim_head)
        self.norm2 = LayerNorm(dim)
        self.attn2 = CrossAttention(dim, context_dim, heads, dim_head)
        self.norm3 = LayerNorm(dim)
        self.ff = FeedForward(dim)

    def forward(self, x, context, scale=1.0):
        # each sub-block's input feeds its norm AND its residual: the norm hands out the alias the residual uses, so the
        # two gradients meet inside the norm's backward kernel (ops.LayerNormBypassFn)
        h, x = self.norm1.with_bypass(x)
        x = self.attn1(h, residual=x, scale=scale)
        h, x = self.norm2.with_bypass(x)
        x = self.attn2(h, context, residual=x, scale=scale)
        h, x = self.norm3.with_bypass(x)
        return self.ff(h, residual=x)


class Transformer2DModel(nn.Module):
    def __init__(self, channels, heads, dim_head, context_dim, groups=32):
        super().__init__()
        self.norm = GroupNorm(groups, channels, 1e-6)
        self.proj_in = Conv2d(channels, channels, 1, pad=(0, 0, 0, 0))
        self.transformer_blocks = nn.ModuleList([BasicTransformerBlock(channels, heads, dim_head, context_dim)])
        self.proj_out = Conv2d(channels, channels, 1, pad=(0, 0, 0, 0))

    def forward(self, x, context, scale=1.0):
        b, h, w, c = x.shape
        y, x = self.norm.with_bypass(x)
        y = self.proj_in(y).view(b, h * w, c)
        for blk in self.transformer_blocks:
            y = blk(y, context, scale)
        return self.proj_out(y.view(b, h, w, c), residual=x)


class _DownBlock(nn.Module):
    def __init__(self, cin, cout, temb_c, layers, eps, groups, heads, context_dim, add_down, cross):
        super().__init__()
        if cross:
            self.attentions = nn.ModuleList([Transformer2DModel(cout, heads, cout // heads, context_dim, groups)
                                             for _ in range(layers)])
        self.resnets = nn.ModuleList([ResnetBlock2D(cin if i == 0 else cout, cout, temb_c, groups, eps)
                                      for i in range(layers)])
        self.cross = cross
        self.downsamplers = nn.ModuleList([_Sampler(Conv2d(cout, cout, 3, stride=2, pad=(1, 1, 1, 1)))]) if add_down else None

    def forward(self, h, temb_act, context, scale):
        outs = ()
        for i, r in enumerate(self.resnets):
            h = r(h, temb_act)
            if self.cross:
                h = self.attentions[i](h, context, scale)
            outs += (h,)
        if self.downsamplers is not None:
            h = self.downsamplers[0](h)
            outs += (h,)
        return h, outs


class _MidBlock(nn.Module):
    def __init__(self, c, temb_c, eps, groups, heads, context_dim):
        super().__init__()
        self.attentions = nn.ModuleList([Transformer2DModel(c, heads, c // heads, context_dim, groups)])
        self.resnets = nn.ModuleList([ResnetBlock2D(c, c, temb_c, groups, eps), ResnetBlock2D(c, c, temb_c, groups, eps)])

    def forward(self, h, temb_act, context, scale):
        h = self.resnets[0](h, temb_act)
        h = self.attentions[0](h, context, scale)
        return self.resnets[1](h, temb_act)


class _UpBlock(nn.Module):
    def __init__(self, cin, prev_c, cout, temb_c, layers, eps, groups, heads, context_dim, add_up, cross):
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
        self.upsamplers = nn.ModuleList([_Sampler(Conv2d(cout, cout, 3, upsample=True))]) if add_up else None

    def forward(self, h, skips, temb_act, context, scale):
        for i, r in enumerate(self.resnets):
            s, skips = skips[-1], skips[:-1]
            h = r(h, temb_act, x2=s) if r.cat_in_place_ok(h, s) else r(ops.concat(h, s), temb_act)
            if self.cross:
                h = self.attentions[i](h, context, scale)
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
        self.config = FrozenConfig(cfg)
        if act_fn != "silu" or downsample_padding != 1 or mid_block_scale_factor != 1:
            raise NotImplementedError("UNet2DConditionModel: configuration outside SD-1.x / miniSD")
        boc = list(block_out_channels)
        temb_c = boc[0] * 4
        heads = attention_head_dim                      # SD-1.x: attention_head_dim is the head COUNT
        self.conv_in = Conv2d(in_channels, boc[0], 3)
        self.time_proj = SimpleNamespace(num_channels=boc[0], flip_sin_to_cos=flip_sin_to_cos, downscale_freq_shift=freq_shift)
        self.time_embedding = TimestepEmbedding(boc[0], temb_c)
        self.down_blocks = nn.ModuleList()
        out_c = boc[0]
        for i, typ in enumerate(down_block_types):
            in_c, out_c = out_c, boc[i]
            self.down_blocks.append(_DownBlock(in_c, out_c, temb_c, layers_per_block, norm_eps, norm_num_groups, heads,
                                               cross_attention_dim, i != len(boc) - 1, typ == "CrossAttnDownBlock2D"))
        self.mid_block = _MidBlock(boc[-1], temb_c, norm_eps, norm_num_groups, heads, cross_attention_dim)
        self.up_blocks = nn.ModuleList()
        rev = list(reversed(boc))
        out_c = rev[0]
        for i, typ in enumerate(up_block_types):
            prev_c, out_c = out_c, rev[i]
            in_c = rev[min(i + 1, len(boc) - 1)]
            self.up_blocks.append(_UpBlock(in_c, prev_c, out_c, temb_c, layers_per_block + 1, norm_eps, norm_num_groups,
                                           heads, cross_attention_dim, i != len(boc) - 1, typ == "CrossAttnUpBlock2D"))
        self.conv_norm_out = GroupNorm(norm_num_groups, boc[0], norm_eps)
        self.conv_out = Conv2d(boc[0], out_channels, 3)

    @property
    def dtype(self):
        return self.conv_in.bias.dtype

    @property
    def device(self):
        return self.conv_in.bias.device

    def forward_nhwc(self, x, timestep, context, scale: float = 1.0):
        t = timestep
        if not torch.is_tensor(t):
            t = torch.tensor([t], dtype=torch.long, device=x.device)
        elif t.ndim == 0:
            t = t[None]
        t = t.to(x.device)
        if t.shape[0] != x.shape[0]:
            t = t.expand(x.shape[0])
        tp = self.time_proj
        emb = ops.timestep_embedding(t, tp.num_channels, tp.flip_sin_to_cos, tp.downscale_freq_shift)
        temb_act = ops.silu(self.time_embedding(emb))
        context = context.to(torch.float32).contiguous()
        h = self.conv_in(x)
        half = ops.half_activations()
        if half:
            # the half-precision activation path (gad/half.py; the reference's `--mixed_precision=fp16` jobs,
            # text_to_image/experiments/setup_train_commands.py:127): from here to conv_out every activation and gradient is bf16
            from . import half as H
            h = H.ToHalfFn.apply(h)
            context = H.to_half(context)
        skips = (h,)
        for blk in self.down_blocks:
            h, outs = blk(h, temb_act, context, scale)
            skips += outs
        h = self.mid_block(h, temb_act, context, scale)
        for blk in self.up_blocks:
            n = len(blk.resnets)
            res, skips = skips[-n:], skips[:-n]
            h = blk(h, res, temb_act, context, scale)
        y = self.conv_out(self.conv_norm_out(h, silu=True))
        return H.ToFloatFn.apply(y) if half else y

    def forward(self, sample, timestep, encoder_hidden_states, cross_attention_kwargs=None):
        if not sample.is_cuda:
            raise ops._capi.GadError("gad.UNet2DConditionModel runs on the MI355X only (no CPU path)")
        scale = (cross_attention_kwargs or {}).get("scale", 1.0)
        y = self.forward_nhwc(ops.nchw_to_nhwc_raw(sample.to(torch.float32).contiguous()), timestep, encoder_hidden_states, scale)
        return SimpleNamespace(sample=ops.to_nchw(y))

    # ---- LoRA plumbing --------------------------------------------------------------------------
    def attention_modules(self):
        """name -> CrossAttention for the 32 attention modules, in diffusers' `attn_processors` order/names."""
        out = {}
        for name, m in self.named_modules():
            if isinstance(m, CrossAttention):
                out[f"{name}.processor"] = m
        return out

    @property
    def attn_processors(self):
        return self.attention_modules()

    def inject_lora(self, rank=4, ranks=None):
        """train_text_to_image_lora.py:779-820: a LoRALinearLayer on to_q/to_k/to_v/to_out[0] of every attention;
        `ranks` (name -> int) overrides the rank per projection (pruned, ragged LoRA)."""
        for p in self.parameters():
            p.requires_grad_(False)                                                 # :746 base frozen
        params = []
        for name, attn in self.attention_modules().items():
            base = name[: -len(".processor")]
            for proj, lin in (("to_q", attn.to_q), ("to_k", attn.to_k), ("to_v", attn.to_v), ("to_out", attn.to_out[0])):
                r = (ranks or {}).get(f"{base}.{proj}", rank)
                layer = LoRALinearLayer(lin.weight.shape[1], lin.weight.shape[0], rank=r).to(lin.weight.device)
                lin.set_lora_layer(layer)
                params += list(layer.parameters())
        return params

    def lora_state_dict(self, prefix: str = ""):
        """Keys as diffusers-0.24 `unet.save_attn_procs` writes them (train_text_to_image_lora.py:1367,1493;
        prune_lora.py:102,196): ``<attn_processors name>.to_{q,k,v,out}_lora.{down,up}.weight`` with NO `unet.` prefix
        (that prefix belongs to pipeline-level `save_lora_weights` files; pass prefix="unet." for those)."""
        sd = {}
        for name, attn in self.attention_modules().items():
            for proj, lin in (("to_q", attn.to_q), ("to_k", attn.to_k), ("to_v", attn.to_v), ("to_out", attn.to_out[0])):
                if lin.lora_layer is not None:
                    sd[f"{prefix}{name}.{proj}_lora.down.weight"] = lin.lora_layer.down.weight.detach().cpu().contiguous()
                    sd[f"{prefix}{name}.{proj}_lora.up.weight"] = lin.lora_layer.up.weight.detach().cpu().contiguous()
        return sd

    def save_attn_procs(self, save_directory, weight_name="pytorch_lora_weights.safetensors"):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        save_file(self.lora_state_dict(), os.path.join(save_directory, weight_name))

    def load_lora_state_dict(self, sd: dict):
        """Accepts the `unet.save_attn_procs` key form and the `unet.`-prefixed pipeline form, per-projection ranks
        included (the upstream loader's single-rank assumption is the bug the reference patches with my_get_processor,
        src/utils.py:84-96).  A non-empty file none of whose keys belong to this U-Net, or one with keys left over,
        is an error - never a silent no-op that would train / score the bare base model."""
        sd = {(k[len("unet."):] if k.startswith("unet.") else k): v for k, v in sd.items()}
        used = set()
        for name, attn in self.attention_modules().items():
            for proj, lin in (("to_q", attn.to_q), ("to_k", attn.to_k), ("to_v", attn.to_v), ("to_out", attn.to_out[0])):
                kd, ku = f"{name}.{proj}_lora.down.weight", f"{name}.{proj}_lora.up.weight"
                if kd in sd or ku in sd:
                    if not (kd in sd and ku in sd):
                        raise KeyError(f"LoRA file has only one of {kd} / {ku}")
                    down, up = sd[kd], sd[ku]
                    if down.shape[1] != lin.weight.shape[1] or up.shape[0] != lin.weight.shape[0] or down.shape[0] != up.shape[1]:
                        raise ValueError(f"{name}.{proj}: LoRA shapes {tuple(down.shape)} / {tuple(up.shape)} do not fit a "
                                         f"{tuple(lin.weight.shape)} projection")
                    layer = LoRALinearLayer(down.shape[1], up.shape[0], rank=down.shape[0]).to(lin.weight.device)
                    with torch.no_grad():
                        layer.down.weight.copy_(down)
                        layer.up.weight.copy_(up)
                    lin.set_lora_layer(layer)
                    used.update((kd, ku))
        left = sorted(set(sd) - used)
        if sd and not used:
            raise KeyError(f"none of the {len(sd)} keys of the LoRA file name an attention projection of this U-Net "
                           f"(first key: {next(iter(sd))!r})")
        if left:
            raise KeyError(f"{len(left)} LoRA keys match no attention projection of this U-Net, e.g. {left[0]!r}")
        ops.WEIGHT_EPOCH[0] += 1
        return len(used) // 2

    def load_attn_procs(self, directory, weight_name="pytorch_lora_weights.safetensors"):
        from safetensors.torch import load_file
        return self.load_lora_state_dict(load_file(os.path.join(directory, weight_name)))

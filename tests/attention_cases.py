"""Seeded cases shared by tests/golden/make_attention_golden.py (which runs the reference's vendored
attention_processor.py on them) and the tests that compare the oracle and the HIP attention block with what it
produced.  Everything comes from numpy RandomState streams, so nothing but outputs has to be stored.

Shapes follow the models the reference runs: CIFAR `AttnDownBlock2D` attention (ddpm_config.py:235-269: C = 256, one
head, GroupNorm(32, eps 1e-6), residual, `_from_deprecated_attn_block`) at 16x16 and 4x4; CelebA-HQ LDM
(ddpm_config.py:425-450: heads 14 / 21 / 28 of dim 32) and its head-grouped-pruned form (prune.py:337-342: heads stay,
head dim 32 -> 23); SD-1.x transformer attentions (self d = 40 / 80 / 160, cross Tk = 77 from 768-wide context, no
bias, no norm, no internal residual) with and without ragged-rank LoRA (prune_lora.py:173-180)."""
import zlib

import numpy as np

_BLOCK = dict(bias=True, groups=32, residual=True, deprecated_block=True)
CASES = {
    "cifar_16x16": dict(_BLOCK, query_dim=256, heads=1, dim_head=256, eps=1e-6, shape=(1, 256, 16, 16)),
    "cifar_4x4": dict(_BLOCK, query_dim=256, heads=1, dim_head=256, eps=1e-6, shape=(3, 256, 4, 4)),
    "celeba_h14": dict(_BLOCK, query_dim=448, heads=14, dim_head=32, eps=1e-5, shape=(1, 448, 8, 8)),
    "celeba_h21": dict(_BLOCK, query_dim=672, heads=21, dim_head=32, eps=1e-5, shape=(1, 672, 4, 4)),
    "celeba_h28": dict(_BLOCK, query_dim=896, heads=28, dim_head=32, eps=1e-5, shape=(1, 896, 4, 4)),
    "celeba_pruned_d23": dict(_BLOCK, query_dim=320, heads=14, dim_head=23, eps=1e-5, shape=(1, 320, 8, 8)),
    "celeba_rescaled": dict(_BLOCK, query_dim=64, heads=2, dim_head=32, eps=1e-5, shape=(2, 64, 4, 4), rescale=1.5),
    "sd_self_d40": dict(query_dim=320, heads=8, dim_head=40, bias=False, residual=False, shape=(2, 32, 320)),
    "sd_self_d80": dict(query_dim=640, heads=8, dim_head=80, bias=False, residual=False, shape=(1, 32, 640)),
    "sd_self_d160": dict(query_dim=1280, heads=8, dim_head=160, bias=False, residual=False, shape=(1, 16, 1280)),
    "sd_cross_d40": dict(query_dim=320, cross_dim=768, heads=8, dim_head=40, bias=False, residual=False,
                         shape=(2, 32, 320), ctx=(2, 77, 768)),
    "sd_cross_d160": dict(query_dim=1280, cross_dim=768, heads=8, dim_head=160, bias=False, residual=False,
                          shape=(1, 16, 1280), ctx=(1, 77, 768)),
    "sd_self_lora_ragged": dict(query_dim=320, heads=8, dim_head=40, bias=False, residual=False, shape=(2, 32, 320),
                                ranks=dict(to_q=4, to_k=3, to_v=7, to_out=5), scale=1.0),
    "sd_cross_lora_ragged": dict(query_dim=320, cross_dim=768, heads=8, dim_head=40, bias=False, residual=False,
                                 shape=(2, 32, 320), ctx=(2, 77, 768), ranks=dict(to_q=8, to_k=2, to_v=5, to_out=12),
                                 scale=0.5, network_alpha=6.0),
}


def _rs(case, what):
    return np.random.RandomState(zlib.crc32(f"{case}/{what}".encode()) & 0x7FFFFFFF)


def projections(attn):
    """(name, linear) of the four projections of a reference- or product-side attention module."""
    return (("to_q", attn.to_q), ("to_k", attn.to_k), ("to_v", attn.to_v), ("to_out", attn.to_out[0]))


def weights(case):
    """state_dict (reference key names) of float32 arrays."""
    c = CASES[case]
    qd, inner, kd = c["query_dim"], c["heads"] * c["dim_head"], c.get("cross_dim") or c["query_dim"]
    w = {}

    def lin(name, cout, cin, bias):
        w[f"{name}.weight"] = (_rs(case, name + ".w").standard_normal((cout, cin)) / np.sqrt(cin)).astype(np.float32)
        if bias:
            w[f"{name}.bias"] = (0.1 * _rs(case, name + ".b").standard_normal(cout)).astype(np.float32)

    if c.get("groups"):
        w["group_norm.weight"] = (1 + 0.1 * _rs(case, "gn.w").standard_normal(qd)).astype(np.float32)
        w["group_norm.bias"] = (0.1 * _rs(case, "gn.b").standard_normal(qd)).astype(np.float32)
    lin("to_q", inner, qd, c["bias"])
    lin("to_k", inner, kd, c["bias"])
    lin("to_v", inner, kd, c["bias"])
    lin("to_out.0", qd, inner, True)
    for proj, r in c.get("ranks", {}).items():
        name = "to_out.0" if proj == "to_out" else proj
        cout, cin = w[f"{name}.weight"].shape
        w[f"{name}.lora_layer.down.weight"] = (_rs(case, proj + ".down").standard_normal((r, cin)) / r).astype(np.float32)
        w[f"{name}.lora_layer.up.weight"] = (0.05 * _rs(case, proj + ".up").standard_normal((cout, r))).astype(np.float32)
    return w


def inputs(case):
    """x (NCHW or [B, T, C]), context or None, and the cotangent g of the output - float32 arrays."""
    c = CASES[case]
    x = (_rs(case, "x").standard_normal(c["shape"]) + 0.3).astype(np.float32)
    ctx = _rs(case, "ctx").standard_normal(c["ctx"]).astype(np.float32) if "ctx" in c else None
    g = _rs(case, "g").standard_normal(c["shape"]).astype(np.float32)
    return x, ctx, g


def probe(case, key, shape):
    """The fixed random tensor a parameter gradient is projected on (float64)."""
    return _rs(case, "probe/" + key).standard_normal(shape)


def checksums(case):
    out = {k: float(np.asarray(v, np.float64).sum()) for k, v in weights(case).items()}
    x, ctx, g = inputs(case)
    out["x"], out["g"] = float(x.astype(np.float64).sum()), float(g.astype(np.float64).sum())
    if ctx is not None:
        out["ctx"] = float(ctx.astype(np.float64).sum())
    return out

"""GPU parity of the half-precision ACTIVATION path (csrc/half.hip, gad/half.py; the reference's `--mixed_precision=fp16` SD jobs,
text_to_image/experiments/setup_train_commands.py:127): every kernel against an fp64 evaluation of the SAME bf16-rounded
operands (so the only differences are fp32 accumulation order and the final bf16 rounding of the output), then the whole
SD U-Net fwd + LoRA backward in bf16 activations against the fp32 path.

Tolerances: contraction outputs stored as bf16: |err| <= 2^-8 |value| + 1e-5 sqrt(K) |a||b| (one bf16 rounding of the result + fp32
accumulation); fp32 outputs (parameter gradients): 2e-5 sqrt(K) of the scale; norms / elementwise: one bf16 rounding of the result."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")
BF = torch.bfloat16


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def hb(t):
    """bf16-rounded copy on the device and the fp64 value of exactly those numbers"""
    h = t.to(BF)
    return h.to(dev), h.double()


def close_h(got, want, extra=0.0, what=""):
    """got: bf16/fp32 device tensor; want: fp64.  One bf16 rounding of the result (2^-8 relative) + `extra` absolute."""
    g = got.detach().float().cpu().double()
    assert g.shape == want.shape, (g.shape, want.shape)
    tol = want.abs() * 2.0 ** -8 + extra + 1e-30
    bad = (g - want).abs() > tol
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.numel()} off; worst {(g - want).abs().max().item():.3e} vs tol {tol[bad].min().item():.3e}"


# ---------------------------------------------------------------------------------------------------------------
# hgemm: dense
# ---------------------------------------------------------------------------------------------------------------
DENSE = [  # M, N, K, tile_hint, splitk_hint
    (256, 320, 320, 0, 0), (300, 320, 328, 0, 0), (128, 640, 64, 2, 0), (1000, 256, 320, 0, 0), (130, 72, 40, 0, 0),
    (77 * 2, 128, 96, 0, 0), (512, 320, 1280, 0, 3), (256, 128, 2048, 1, 4), (129, 321, 72, 1, 0), (640, 960, 192, 2, 2),
    (64, 4, 2880, 0, 0), (512, 320, 320, 5, 0), (700, 640, 328, 5, 0), (256, 960, 1280, 5, 3), (1000, 320, 64, 5, 0),
    (700, 640, 328, 6, 0), (300, 320, 328, 7, 0), (256, 960, 1280, 6, 3), (1000, 256, 328, 8, 0), (130, 72, 40, 8, 2), (1000, 256, 328, 9, 0), (129, 321, 72, 9, 2),
]


@pytest.mark.parametrize("M,N,K,tile,sk", DENSE)
def test_hgemm_dense(M, N, K, tile, sk):
    from gad import half
    a, ad = hb(rnd(M, K, seed=1))
    b, bd = hb(rnd(N, K, seed=2))
    bias = rnd(N, seed=3).to(dev)
    res, resd = hb(rnd(M, N, seed=4))
    want = 0.5 * (ad @ bd.T) + bias.cpu().double() + resd
    out = torch.empty((M, N), device=dev, dtype=BF)
    half.hgemm_raw(a, b, out, M, N, K, K, K, N, alpha=0.5, bias=bias, residual=res, ldr=N, tile_hint=tile, splitk_hint=sk)
    close_h(out, want, extra=2e-5 * math.sqrt(K), what="bf16 out")
    # fp32 output, then accumulate on top of it
    o32 = torch.empty((M, N), device=dev, dtype=torch.float32)
    half.hgemm_raw(a, b, o32, M, N, K, K, K, N, out_f32=True, tile_hint=tile, splitk_hint=sk)
    w32 = ad @ bd.T
    assert (o32.cpu().double() - w32).abs().max() < 2e-5 * math.sqrt(K) * max(1.0, w32.abs().max().item())
    half.hgemm_raw(a, b, o32, M, N, K, K, K, N, out_f32=True, accumulate=True, tile_hint=tile, splitk_hint=sk)
    assert (o32.cpu().double() - 2 * w32).abs().max() < 4e-5 * math.sqrt(K) * max(1.0, w32.abs().max().item())


@pytest.mark.parametrize("M,N,K,r,tile", [(256, 320, 320, 256, 0), (200, 640, 768, 8, 0), (128, 320, 96, 24, 0), (256, 128, 320, 64, 0),
                                         (600, 320, 320, 256, 5), (300, 640, 96, 24, 5)])
def test_hgemm_k_concat(M, N, K, r, tile):
    """[x | mid] . [W | up]^T in one launch (the fused LoRA linear)"""
    from gad import half
    x, xd = hb(rnd(M, K, seed=1))
    mid, midd = hb(rnd(M, r, seed=2))
    w, wd = hb(rnd(N, K, seed=3, scale=0.1))
    up, upd = hb(rnd(N, r, seed=4, scale=0.1))
    out = torch.empty((M, N), device=dev, dtype=BF)
    half.hgemm_raw(x, w, out, M, N, K + r, K, K, N, A2=mid, B2=up, lda2=r, ldb2=r, k_split=K, tile_hint=tile)
    close_h(out, xd @ wd.T + midd @ upd.T, extra=2e-5 * math.sqrt(K + r), what="k-concat")


@pytest.mark.parametrize("T,M,N,sk", [(256, 128, 128, 0), (1000, 320, 256, 0), (77 * 2, 96, 136, 0), (4096, 320, 320, 0), (300, 6, 72, 0), (640, 264, 8, 3)])
def test_hgemm_tn(T, M, N, sk):
    """C = A^T B with both operands [tokens][channels] (the LoRA parameter gradients), fp32 output, overwrite then accumulate;
    output dims that are not multiples of 8 read zero pad columns of operands allocated 8-aligned"""
    from gad import half
    M8, N8 = (M + 7) // 8 * 8, (N + 7) // 8 * 8
    a, b = torch.zeros(T, M8), torch.zeros(T, N8)
    a[:, :M], b[:, :N] = rnd(T, M, seed=1), rnd(T, N, seed=2)
    ah, ad = hb(a)
    bh, bd = hb(b)
    want = ad[:, :M].T @ bd[:, :N]
    out = torch.full((M, N), 7.0, device=dev)
    half.wgrad_raw(ah, bh, out, accumulate=False)
    tol = 2e-5 * math.sqrt(T) * max(1.0, want.abs().max().item())
    assert (out.cpu().double() - want).abs().max() < tol
    half.wgrad_raw(ah, bh, out, accumulate=True, alpha=0.5)
    assert (out.cpu().double() - 1.5 * want).abs().max() < 2 * tol


def test_hgemm_rejects_misaligned():
    from gad import _capi, half
    a = torch.zeros((64, 40), device=dev, dtype=BF)
    b = torch.zeros((64, 40), device=dev, dtype=BF)
    out = torch.empty((64, 64), device=dev, dtype=BF)
    with pytest.raises(_capi.GadError):
        half.hgemm_raw(a[:, :36], b[:, :36], out, 64, 64, 36, 40, 40, 64)           # K % 8 != 0


# ---------------------------------------------------------------------------------------------------------------
# hgemm: convolution gathers
# ---------------------------------------------------------------------------------------------------------------
def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


CONVS = [  # B, H, W, Cin, Cout, k, stride, pad, upsample
    (2, 16, 16, 64, 320, 3, 1, 1, False), (3, 8, 8, 128, 128, 3, 1, 1, False), (2, 16, 16, 64, 64, 3, 2, 1, False),
    (2, 8, 8, 64, 320, 3, 1, 1, True), (1, 12, 20, 72, 40, 3, 1, 1, False), (2, 16, 16, 320, 4, 3, 1, 1, False),
    (2, 8, 8, 64, 96, 1, 1, 0, False),
]


@pytest.mark.parametrize("tile", [0, 5, 6, 7])
@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad,ups", CONVS)
def test_hconv_forward(B, H, W, Cin, Cout, k, stride, pad, ups, tile):
    from gad import half
    if tile >= 5 and Cout % 320:
        pytest.skip("the eight-wave form takes 320-column tiles")
    x, xd = hb(rnd(B, Cin, H, W, seed=1))
    w, wd = hb(rnd(Cout, Cin, k, k, seed=2, scale=0.05))
    bias = rnd(Cout, seed=3).to(dev)
    xin = F.interpolate(xd, scale_factor=2, mode="nearest") if ups else xd
    want = F.conv2d(xin, wd, bias.cpu().double(), stride=stride, padding=pad)
    temb = rnd(B, Cout, seed=5).to(dev)
    res, resd = hb(rnd(*want.shape, seed=4))
    want = want + temb.cpu().double()[:, :, None, None] + resd
    wh = w.permute(0, 2, 3, 1).contiguous()
    y = half.conv_fwd_raw(_nhwc(x), wh, bias, k, k, stride, (pad,) * 4, ups, rowadd=temb, residual=_nhwc(res), tile_hint=tile)
    close_h(y, _nhwc(want), extra=2e-5 * math.sqrt(k * k * Cin), what="conv fwd")


def test_hconv_two_sources():
    from gad import half
    B, H, W, C1, C2, Cout = 2, 8, 8, 64, 128, 320
    x1, x1d = hb(rnd(B, C1, H, W, seed=1))
    x2, x2d = hb(rnd(B, C2, H, W, seed=2))
    w, wd = hb(rnd(Cout, C1 + C2, 3, 3, seed=3, scale=0.05))
    want = F.conv2d(torch.cat([x1d, x2d], 1), wd, padding=1)
    y = half.conv_fwd_raw(_nhwc(x1), w.permute(0, 2, 3, 1).contiguous(), None, 3, 3, x2=_nhwc(x2))
    close_h(y, _nhwc(want), extra=2e-5 * math.sqrt(9 * (C1 + C2)), what="two-source conv")


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad,ups", CONVS[:5] + CONVS[6:])
def test_hconv_autograd_dgrad(B, H, W, Cin, Cout, k, stride, pad, ups):
    """HConv2dFn: data gradient through rotated / transposed weight copies (stride 1, stride 2, upsample-fused, 1x1)"""
    from gad import ops
    x, xd = hb(rnd(B, Cin, H, W, seed=1))
    w = (rnd(Cout, Cin, k, k, seed=2, scale=0.05)).to(BF).float()
    wp = torch.nn.Parameter(w.to(dev).contiguous(memory_format=torch.channels_last), requires_grad=False)
    xr = xd.clone().requires_grad_(True)
    xin = F.interpolate(xr, scale_factor=2, mode="nearest") if ups else xr
    want = F.conv2d(xin, w.double(), stride=stride, padding=pad)
    dy, dyd = hb(rnd(*want.shape, seed=3))
    want.backward(dyd)
    xg = _nhwc(x).requires_grad_(True)
    y = ops.conv2d(xg, wp, None, None, None, stride, (pad,) * 4, ups)
    assert y.dtype == BF
    y.backward(_nhwc(dy))
    close_h(y, _nhwc(want.detach()), extra=2e-5 * math.sqrt(k * k * Cin), what="conv fwd")
    # upsample-fused: the gradient on the 2x grid is stored as bf16 before its 2 x 2 blocks are summed: four roundings, not one
    extra = 2.0 ** -7 * xr.grad.abs().max().item() if ups else 0.0
    close_h(xg.grad, _nhwc(xr.grad), extra=2e-5 * math.sqrt(k * k * Cout) * 2 + extra, what="conv dgrad")


# ---------------------------------------------------------------------------------------------------------------
# small kernels
# ---------------------------------------------------------------------------------------------------------------
def test_cast_transpose_add_upsample():
    from gad import _capi, half
    x = rnd(777, 100, seed=1)
    assert torch.equal(half.to_half(x.to(dev)).cpu(), x.to(BF))
    assert torch.equal(half.to_float(x.to(BF).to(dev)).cpu(), x.to(BF).float())
    t = half.transpose_raw(x.to(dev))                                # fp32 in, R = 777 -> padded to 784
    assert t.shape == (100, 784) and torch.equal(t[:, :777].cpu(), x.to(BF).T) and not t[:, 777:].any()
    t2 = half.transpose_raw(x.to(BF).to(dev)[:, :96])                 # strided bf16 source
    assert torch.equal(t2[:, :777].cpu(), x.to(BF)[:, :96].T)
    a, b = x.to(BF).to(dev), rnd(777, 100, seed=2).to(BF).to(dev)
    assert torch.equal(half.add_raw(a, b).cpu(), (a.float() + b.float()).to(BF).cpu())
    dy = rnd(2, 8, 12, 64, seed=3).to(BF)
    dx = torch.empty((2, 4, 6, 64), device=dev, dtype=BF)
    _capi.check(_capi.load().gad_h_upsample2x_bwd(dy.to(dev).data_ptr(), dx.data_ptr(), 2, 4, 6, 64, half._st()), "up")
    want = dy.float().view(2, 4, 2, 6, 2, 64).sum(dim=(2, 4))
    close_h(dx, want.double(), what="upsample2x bwd")


@pytest.mark.parametrize("B,HW,C,G,silu", [(2, 64, 320, 32, True), (3, 256, 64, 32, False), (2, 100, 1280, 32, True), (1, 4096, 320, 32, True),
                                           (2, 64, 2560, 32, True), (2, 64, 960, 32, True)])
def test_h_groupnorm_fwd_bwd(B, HW, C, G, silu):
    from gad import ops
    x, xd = hb(rnd(B, HW, C, seed=1) * 1.5 + 0.3)
    gamma, beta = (rnd(C, seed=2) * 0.2 + 1), rnd(C, seed=3) * 0.1
    xr = xd.clone().requires_grad_(True)
    y = F.group_norm(xr.transpose(1, 2), G, gamma.double(), beta.double(), 1e-5).transpose(1, 2)
    if silu:
        y = F.silu(y)
    dy, dyd = hb(rnd(B, HW, C, seed=4))
    byp, bypd = hb(rnd(B, HW, C, seed=5))
    y.backward(dyd)
    g_, b_ = torch.nn.Parameter(gamma.to(dev), requires_grad=False), torch.nn.Parameter(beta.to(dev), requires_grad=False)
    xg = x.clone().requires_grad_(True)
    out, alias = ops.group_norm_bypass(xg, g_, b_, G, 1e-5, silu)
    assert out.dtype == BF
    (out.float() * dy.float()).sum().backward(retain_graph=True)                  # gradient of `out` only
    close_h(out, y.detach(), extra=2e-5, what="gn fwd")
    close_h(xg.grad, xr.grad, extra=3e-5 * max(1.0, xr.grad.abs().max().item()), what="gn bwd")
    xg.grad = None
    torch.autograd.backward([out, alias], [dy, byp])                                # both consumers: summed in the kernel's store
    close_h(xg.grad, xr.grad + bypd, extra=3e-5 * max(1.0, xr.grad.abs().max().item()), what="gn bwd + bypass")


def test_h_groupnorm_two_sources():
    from gad import half
    B, HW, C1, C2, G = 2, 64, 640, 320, 32
    x1, x1d = hb(rnd(B, HW, C1, seed=1))
    x2, x2d = hb(rnd(B, HW, C2, seed=2) * 2 + 1)
    gamma, beta = (rnd(C1 + C2, seed=3) * 0.2 + 1), rnd(C1 + C2, seed=4) * 0.1
    want = F.silu(F.group_norm(torch.cat([x1d, x2d], -1).transpose(1, 2), G, gamma.double(), beta.double(), 1e-5).transpose(1, 2))
    y, _, _ = half.group_norm_raw(x1, x2, gamma.to(dev), beta.to(dev), G, 1e-5, True)
    close_h(y, want, extra=2e-5, what="two-source gn")


@pytest.mark.parametrize("rows,C", [(7, 320), (300, 640), (77 * 3, 1280), (5, 96)])
def test_h_layernorm_fwd_bwd(rows, C):
    from gad import ops
    x, xd = hb(rnd(rows, C, seed=1) * 1.7 + 0.4)
    gamma, beta = (rnd(C, seed=2) * 0.2 + 1), rnd(C, seed=3) * 0.1
    xr = xd.clone().requires_grad_(True)
    y = F.layer_norm(xr, (C,), gamma.double(), beta.double(), 1e-5)
    dy, dyd = hb(rnd(rows, C, seed=4))
    byp, bypd = hb(rnd(rows, C, seed=5))
    y.backward(dyd)
    g_, b_ = torch.nn.Parameter(gamma.to(dev), requires_grad=False), torch.nn.Parameter(beta.to(dev), requires_grad=False)
    xg = x.clone().requires_grad_(True)
    out, alias = ops.layer_norm_bypass(xg, g_, b_, 1e-5)
    torch.autograd.backward([out, alias], [dy, byp])
    close_h(out, y.detach(), extra=2e-5, what="ln fwd")
    close_h(xg.grad, xr.grad + bypd, extra=3e-5 * max(1.0, xr.grad.abs().max().item()), what="ln bwd")


def test_h_geglu_fwd_bwd():
    from gad import ops
    h, hd = hb(rnd(50, 2 * 128, seed=1))
    hr = hd.clone().requires_grad_(True)
    a, gate = hr.chunk(2, dim=-1)
    y = a * F.gelu(gate)
    dy, dyd = hb(rnd(50, 128, seed=2))
    y.backward(dyd)
    hg = h.clone().requires_grad_(True)
    out = ops.geglu(hg)
    out.backward(dy)
    close_h(out, y.detach(), extra=1e-6, what="geglu fwd")
    close_h(hg.grad, hr.grad, extra=1e-6, what="geglu bwd")


@pytest.mark.parametrize("B,Tq,Tk,heads,d", [(2, 256, 256, 8, 40), (2, 64, 77, 4, 40), (1, 200, 200, 5, 64), (2, 64, 64, 8, 160), (2, 128, 77, 4, 16),
                                             (1, 1024, 1024, 2, 80), (8, 1024, 1024, 8, 40), (8, 1000, 1000, 8, 40), (16, 1024, 1024, 8, 40), (8, 1024, 1024, 8, 80)])   # (32 rows per wave in the backward; 64 queries per wave forward; d = 80 with 32 rows)
def test_h_attention_fwd_bwd(B, Tq, Tk, heads, d):
    """bf16 q / k / v / dO in, bf16 o / dq / dk / dv out; the products run on bf16-rounded probabilities, so the comparison with
    an fp64 evaluation of the same inputs carries the operand rounding of P and dS (2^-8 relative per product term)."""
    from gad import ops
    C = heads * d
    q, qd = hb(rnd(B, Tq, C, seed=1, scale=0.5))
    k, kd = hb(rnd(B, Tk, C, seed=2, scale=0.5))
    v, vd = hb(rnd(B, Tk, C, seed=3))
    qr, kr, vr = (t.clone().requires_grad_(True) for t in (qd, kd, vd))

    def split(t, T):
        return t.view(B, T, heads, d).transpose(1, 2)
    o = F.scaled_dot_product_attention(split(qr, Tq), split(kr, Tk), split(vr, Tk)).transpose(1, 2).reshape(B, Tq, C)
    do, dod = hb(rnd(B, Tq, C, seed=4))
    o.backward(dod)
    gq, gk, gv = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = ops.attention_core(gq, gk, gv, heads)
    assert out.dtype == BF
    out.backward(do)

    def rel(got, want):
        return ((got.float().cpu().double() - want).norm() / want.norm()).item()
    assert rel(out, o.detach()) < 6e-3
    assert rel(gq.grad, qr.grad) < 1.2e-2 and rel(gk.grad, kr.grad) < 1.2e-2 and rel(gv.grad, vr.grad) < 1.2e-2
    # element-wise: nothing wildly off (a mis-indexed tile would be O(1))
    assert (out.float().cpu().double() - o.detach()).abs().max() < 0.03 * o.detach().abs().max()


# ---------------------------------------------------------------------------------------------------------------
# LoRA linear through autograd (ragged ranks, gradients in fp32)
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("M,K,N,r", [(256, 320, 320, 256), (77 * 2, 96, 128, 6), (300, 640, 640, 17)])
def test_h_lora_linear_autograd(M, K, N, r):
    from gad import ops
    x, xd = hb(rnd(M, K, seed=1))
    w = rnd(N, K, seed=2, scale=0.05).to(BF).float()
    bias = rnd(N, seed=3)
    down = rnd(r, K, seed=4, scale=0.1).to(BF).float()
    up = rnd(N, r, seed=5, scale=0.1).to(BF).float()
    res, resd = hb(rnd(M, N, seed=6))
    s = 0.7
    xr = xd.clone().requires_grad_(True)
    dr, ur = down.double().requires_grad_(True), up.double().requires_grad_(True)
    mid = s * (xr @ dr.T)
    y = xr @ w.double().T + bias.double() + mid.to(BF).double() @ ur.T + resd        # (mid is stored as bf16 on the device)
    y_exact = xr @ w.double().T + bias.double() + mid @ ur.T + resd
    dy, dyd = hb(rnd(M, N, seed=7))
    y_exact.backward(dyd)
    wp = torch.nn.Parameter(w.to(dev), requires_grad=False)
    bp = torch.nn.Parameter(bias.to(dev), requires_grad=False)
    dp, upp = torch.nn.Parameter(down.to(dev)), torch.nn.Parameter(up.to(dev))
    xg = x.clone().requires_grad_(True)
    out = ops.lora_linear(xg, wp, bp, dp, upp, s, res)
    out.backward(dy)
    # mid is stored as bf16: where the device's fp32 value and the fp64 one straddle a rounding boundary the two differ by one
    # bf16 step of mid, times an up entry
    flip = 2.0 ** -7 * mid.detach().abs().max().item() * up.abs().max().item()
    close_h(out, y.detach(), extra=3e-5 * math.sqrt(K + r) + flip, what="lora fwd")

    def rel(got, want):
        return ((got.float().cpu().double() - want).norm() / want.norm()).item()
    assert rel(xg.grad, xr.grad) < 6e-3            # dmid is stored as bf16 on the way
    assert dp.grad.dtype == torch.float32 and upp.grad.dtype == torch.float32
    assert rel(dp.grad, dr.grad) < 6e-3 and rel(upp.grad, ur.grad) < 6e-3


@pytest.mark.parametrize("r", [256, 24])
def test_h_lora_gradients_into_flat_slots(r):
    """FusedTrainer's route: the LoRA gradients land in the flat gradient buffer's slots (first use overwrites, a second use of
    the same matrices adds) and equal what autograd receives without sinks - for the full-weight-gradient form (r = 256) and the
    direct form (r = 24)."""
    from gad import ops
    from gad.training import flatten_params
    M, K, N, s = 384, 320, 320, 0.5
    x1, x2 = (rnd(M, K, seed=i).to(BF).to(dev) for i in (1, 2))
    dy1, dy2 = (rnd(M, N, seed=i).to(BF).to(dev) for i in (3, 4))
    w = torch.nn.Parameter(rnd(N, K, seed=5, scale=0.05).to(dev), requires_grad=False)

    def fresh():
        return torch.nn.Parameter(rnd(r, K, seed=6, scale=0.1).to(dev)), torch.nn.Parameter(rnd(N, r, seed=7, scale=0.1).to(dev))
    down, up = fresh()
    for xx, dd in ((x1, dy1), (x2, dy2)):
        ops.lora_linear(xx.clone().requires_grad_(True), w, None, down, up, s, None).backward(dd)
    want_d, want_u = down.grad.clone(), up.grad.clone()                      # autograd accumulated the two uses
    down, up = fresh()
    flat, gflat = flatten_params([down, up])
    gflat.fill_(123.0)                                                        # stale values a first write must overwrite
    ops.begin_backward_step()
    try:
        for xx, dd in ((x1, dy1), (x2, dy2)):
            ops.lora_linear(xx.clone().requires_grad_(True), w, None, down, up, s, None).backward(dd)
    finally:
        ops.end_backward_step()
    assert down.grad is None and up.grad is None
    for got, want in ((down._gad_sink, want_d), (up._gad_sink, want_u)):
        assert (got - want).abs().max() <= 1e-5 * want.abs().max() + 1e-6


# ---------------------------------------------------------------------------------------------------------------
# the whole SD U-Net: bf16 activations vs the fp32 path
# ---------------------------------------------------------------------------------------------------------------
def test_sd_unet_half_activations_vs_fp32():
    import gad
    from gad import ops
    from test_gpu_sd import _pair
    _, net = _pair(lora_rank=8)
    x, ctx, t = rnd(2, 4, 16, 16, seed=1).to(dev), rnd(2, 77, 96, seed=2).to(dev), torch.tensor([5, 800]).to(dev)
    noise = rnd(2, 4, 16, 16, seed=3).to(dev)

    def run():
        for p in net.parameters():
            p.grad = None
        y = net(x, t, ctx).sample
        _, d = ops.mse_fwd_bwd_raw(y.contiguous(), noise)
        y.backward(d)
        return y.detach().clone(), {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}
    y32, g32 = run()
    try:
        gad.set_operand_precision("bf16")
        assert ops.half_activations()
        y16, g16 = run()
    finally:
        gad.set_operand_precision("no")
    assert y16.dtype == torch.float32 and set(g16) == set(g32) and len(g16) == 32 * 4 * 2
    rel = ((y16 - y32).norm() / y32.norm()).item()
    assert rel < 3e-2, rel
    num = sum(((g16[n] - g32[n]).double() ** 2).sum().item() for n in g32)
    den = sum((g32[n].double() ** 2).sum().item() for n in g32)
    assert math.sqrt(num / den) < 8e-2, math.sqrt(num / den)
    for n in g32:                                                   # every gradient points the same way
        a, b = g16[n].double().flatten(), g32[n].double().flatten()
        if b.norm() > 1e-3 * math.sqrt(den / len(g32)):
            assert torch.dot(a, b) / (a.norm() * b.norm()) > 0.97, n


def test_sd_unet_half_sampling_of_a_trainable_model_and_refusal_to_train_the_base():
    """A model whose base parameters still require grad samples fine under no_grad on the half path (nothing wants their
    gradients); asking autograd for them is an error, never a silent zero gradient."""
    import gad
    from gad import _capi
    from test_gpu_sd import SMALL
    torch.manual_seed(0)
    net = gad.UNet2DConditionModel(**SMALL).to(dev)
    assert all(p.requires_grad for p in net.parameters())
    x, ctx, t = rnd(2, 4, 16, 16, seed=1).to(dev), rnd(2, 77, 96, seed=2).to(dev), torch.tensor([5, 800]).to(dev)
    y32 = net(x, t, ctx).sample.detach()
    try:
        gad.set_operand_precision("bf16")
        with torch.no_grad():
            y16 = net(x, t, ctx).sample
        with pytest.raises(_capi.GadError):
            net(x, t, ctx)
    finally:
        gad.set_operand_precision("no")
    assert ((y16 - y32).norm() / y32.norm()).item() < 3e-2


def test_graphed_training_step_equals_eager():
    """FusedTrainer(use_graph=True): the half-path LoRA step replayed from a hipGraph leaves bit-identical weights, optimizer state and
    losses to the eager launches (same kernels, same order), step after step with changing inputs and learning rate."""
    import gad
    from gad import ops
    from test_gpu_sd import SMALL
    sch = gad.DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000)
    xs = [rnd(4, 4, 16, 16, seed=10 + i).to(dev) for i in range(6)]
    ns = [rnd(4, 4, 16, 16, seed=20 + i).to(dev) for i in range(6)]
    ts = [torch.randint(0, 1000, (4,), generator=torch.Generator().manual_seed(30 + i)).to(dev) for i in range(6)]
    cs = [rnd(4, 77, 96, seed=40 + i).to(dev) for i in range(6)]

    def run(use_graph):
        torch.manual_seed(0)
        net = gad.UNet2DConditionModel(**SMALL).to(dev)
        lora = net.inject_lora(rank=8)
        with torch.no_grad():
            for n, p in net.named_parameters():
                if n.endswith("lora_layer.up.weight"):
                    p.copy_(rnd(*p.shape, seed=hash(n) % 1000, scale=0.02).to(dev))
        tr = gad.FusedTrainer(net, sch, None, lr=3e-4, weight_decay=1e-6, adamw=True, max_grad_norm=1.0, params=lora,
                              lr_schedule=gad.lr_lambda("cosine", 50, 0), use_graph=use_graph)
        losses = [float(tr.step(xs[i], ns[i], ts[i], cs[i]).item()) for i in range(6)]
        return tr, losses
    try:
        gad.set_operand_precision("bf16")
        eager, le = run(False)
        graphed, lg = run(True)
    finally:
        gad.set_operand_precision("no")
    assert graphed._graph is not None and not graphed._graph_failed, "the step was not captured"
    assert le == lg
    assert torch.equal(eager.flat, graphed.flat) and torch.equal(eager.m, graphed.m) and torch.equal(eager.v, graphed.v)


# ---------------------------------------------------------------------------------------------------------------
# out-of-bounds canaries (as tests/test_gpu_guards.py does for the fp32-storage kernels): every output of the half path at its exact
# shape and every caller-owned scratch region at EXACTLY the size the C ABI's query returns, each between poisoned bands
# ---------------------------------------------------------------------------------------------------------------
PAD, POISON = 8192, 0x5A


class HGuards:
    def __init__(self):
        self.regions = []

    def _alloc(self, kind, nbytes, device):
        buf = torch.full((nbytes + 2 * PAD,), POISON, dtype=torch.uint8, device=device)
        self.regions.append((kind, buf, nbytes))
        return buf[PAD:PAD + nbytes]

    def scratch(self, kind, nbytes, device):
        return self._alloc(kind, nbytes, device)

    def out(self, shape, device, dtype):
        n = math.prod(shape) * torch.empty((), dtype=dtype).element_size()
        return self._alloc(f"out{tuple(shape)}", n, device).view(dtype).view(shape)

    def check(self):
        assert self.regions, "no guarded allocation was made"
        for kind, buf, n in self.regions:
            lo, hi = buf[:PAD], buf[PAD + n:]
            assert bool((lo == POISON).all()) and bool((hi == POISON).all()), \
                f"guard band of {kind} ({n} bytes) was written ({int((lo != POISON).sum())} bytes below, {int((hi != POISON).sum())} above)"


def hguarded(fn):
    from gad import ops
    g = HGuards()
    ops.SCRATCH_ALLOC, ops.OUT_ALLOC_DT = g.scratch, g.out
    try:
        r = fn()
        torch.cuda.synchronize()
    finally:
        ops.SCRATCH_ALLOC = ops.OUT_ALLOC_DT = None
    g.check()
    return r, g


@pytest.mark.parametrize("M,N,K,tile,sk", [(130, 72, 40, 0, 0), (300, 320, 328, 2, 0), (300, 320, 328, 5, 0), (257, 640, 1288, 5, 3), (129, 321, 72, 1, 2),
                                           (1000, 256, 320, 0, 0), (64, 4, 2880, 0, 0), (513, 960, 64, 2, 0)])
def test_hgemm_guard_bands(M, N, K, tile, sk):
    from gad import half
    a, b = rnd(M, K, seed=1).to(BF).to(dev), rnd(N, K, seed=2).to(BF).to(dev)
    res = rnd(M, N, seed=3).to(BF).to(dev)

    def run():
        out = half._empty((M, N), dev)
        half.hgemm_raw(a, b, out, M, N, K, K, K, N, residual=res, ldr=N, tile_hint=tile, splitk_hint=sk)
        o32 = half._empty((M, N), dev, torch.float32)
        half.hgemm_raw(a, b, o32, M, N, K, K, K, N, out_f32=True, tile_hint=tile, splitk_hint=sk)
        return out, o32
    (g_out, g_32), g = hguarded(run)
    out, o32 = run()
    assert torch.equal(g_out, out) and torch.equal(g_32, o32)
    if sk > 1:
        assert any(k == "ws" for k, _, _ in g.regions)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,stride,pad,ups", CONVS[:5] + CONVS[6:])
def test_hconv_autograd_guard_bands(B, H, W, Cin, Cout, k, stride, pad, ups):
    from gad import ops
    x = _nhwc(rnd(B, Cin, H, W, seed=1)).to(BF).to(dev)
    wp = torch.nn.Parameter(rnd(Cout, Cin, k, k, seed=2, scale=0.05).to(dev).contiguous(memory_format=torch.channels_last), requires_grad=False)
    Ho = ((2 * H if ups else H) + 2 * pad - k) // stride + 1
    dy = rnd(B, Ho, Ho if H == W else ((2 * W if ups else W) + 2 * pad - k) // stride + 1, Cout, seed=3).to(BF).to(dev)

    def run():
        xg = x.clone().requires_grad_(True)
        y = ops.conv2d(xg, wp, None, None, None, stride, (pad,) * 4, ups)
        y.backward(dy)
        return y.detach(), xg.grad
    (gy, gdx), _ = hguarded(run)
    y, dx = run()
    assert torch.equal(gy, y) and torch.equal(gdx, dx)


def test_half_block_ops_guard_bands():
    """GroupNorm (+ bypass), LayerNorm, GEGLU, the LoRA linear (both gradient forms) and ragged attention, forward + backward"""
    from gad import ops

    def P(t, grad=False):
        return torch.nn.Parameter(t.to(dev), requires_grad=grad)
    x = rnd(2, 100, 320, seed=1).to(BF).to(dev)
    dy = rnd(2, 100, 320, seed=2).to(BF).to(dev)
    gam, bet = P(rnd(320, seed=3) * 0.1 + 1), P(rnd(320, seed=4) * 0.1)
    w = P(rnd(320, 320, seed=5, scale=0.05))
    h4 = rnd(200, 2 * 320, seed=6).to(BF).to(dev)
    q, k, v = (rnd(2, 200, 160, seed=i, scale=0.5).to(BF).to(dev) for i in (7, 8, 9))
    kc, vc = (rnd(2, 77, 160, seed=i, scale=0.5).to(BF).to(dev) for i in (10, 11))

    def run():
        outs = []
        for silu in (True, False):
            xg = x.clone().requires_grad_(True)
            o, al = ops.group_norm_bypass(xg, gam, bet, 32, 1e-5, silu)
            torch.autograd.backward([o, al], [dy, dy])
            outs += [o.detach(), xg.grad]
        xg = x.clone().requires_grad_(True)
        o, al = ops.layer_norm_bypass(xg, gam, bet, 1e-5)
        torch.autograd.backward([o, al], [dy, dy])
        outs += [o.detach(), xg.grad]
        hg = h4.clone().requires_grad_(True)
        o = ops.geglu(hg)
        o.backward(dy.view(200, 320))
        outs += [o.detach(), hg.grad]
        for r in (256, 12):
            down, up = P(rnd(r, 320, seed=12, scale=0.1), True), P(rnd(320, r, seed=13, scale=0.1), True)
            xg = x.view(200, 320).clone().requires_grad_(True)
            o = ops.lora_linear(xg, w, None, down, up, 0.5, None)
            o.backward(dy.view(200, 320))
            outs += [o.detach(), xg.grad, down.grad, up.grad]
        for kk, vv in ((k, v), (kc, vc)):
            gq, gk, gv = (t.clone().requires_grad_(True) for t in (q, kk, vv))
            o = ops.attention_core(gq, gk, gv, 4)
            o.backward(q)
            outs += [o.detach(), gq.grad, gk.grad, gv.grad]
        return outs
    got, g = hguarded(run)
    want = run()
    assert len(got) == len(want) and all(torch.equal(a, b) for a, b in zip(got, want))
    assert any(k == "ws" for k, _, _ in g.regions)

"""GPU parity tests (run on the MI355X box: pytest -m gpu): every HIP kernel, called
through the C ABI, against a CPU fp64 restatement built from stock torch ops on the same
seeded inputs.  Tolerance for the fp32 MFMA contractions: |err| <= 2e-5 * K^0.5 * rms(A)*rms(B)
-ish, expressed below as rtol/atol on outputs normalised to O(1)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

dev = torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from gad import ops as o
    return o


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g, dtype=torch.float32) * scale


def close(got, want, rtol=2e-4, atol=2e-4):
    got = got.detach().cpu().double()
    want = want.detach().cpu().double()
    assert got.shape == want.shape, (got.shape, want.shape)
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    assert torch.allclose(got, want, rtol=rtol, atol=atol * max(1.0, ref)), f"max err {err:.3e} (ref max {ref:.3e})"


def nhwc(x):  # NCHW cpu -> NHWC gpu
    return x.permute(0, 2, 3, 1).contiguous().to(dev)


def cl_weight(w):  # [O,I,kh,kw] cpu -> channels_last-stored gpu parameter-like tensor
    return w.to(dev).contiguous(memory_format=torch.channels_last)


# ------------------------------------------------------------------ dense contraction ----
@pytest.mark.parametrize("M,N,K", [(64, 64, 32), (200, 132, 100), (512, 256, 1024), (31 * 4, 3, 64), (1000, 512, 128)])
@pytest.mark.parametrize("tile,splitk", [(0, 0), (1, 0), (2, 0), (2, 3), (1, 2)])
def test_gemm_kc_kc(ops, M, N, K, tile, splitk):
    from gad._capi import A_KC, B_KC
    a, b = rnd(M, K, seed=1), rnd(N, K, seed=2)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    c = torch.empty(M, N, device=dev)
    ops.gemm_raw(a.to(dev), b.to(dev), c, A_KC, B_KC, M, N, K, K, K, N, alpha=0.5, bias=bias.to(dev),
                 residual=res.to(dev), ldr=N, tile_hint=tile, splitk_hint=splitk)
    want = 0.5 * (a.double() @ b.double().T) + bias.double() + res.double()
    close(c, want, atol=2e-5 * math.sqrt(K))


@pytest.mark.parametrize("M,N,K", [(64, 64, 64), (132, 200, 96), (256, 512, 1000)])
@pytest.mark.parametrize("tile", [1, 2])
def test_gemm_other_layouts(ops, M, N, K, tile):
    from gad._capi import A_KC, A_MC, B_MC
    a, b = rnd(M, K, seed=1), rnd(K, N, seed=2)
    want = a.double() @ b.double()
    c = torch.empty(M, N, device=dev)
    ops.gemm_raw(a.to(dev), b.to(dev), c, A_KC, B_MC, M, N, K, K, N, N, tile_hint=tile)
    close(c, want, atol=2e-5 * math.sqrt(K))
    at = a.T.contiguous()  # stored [K][M]
    c2 = torch.empty(M, N, device=dev)
    ops.gemm_raw(at.to(dev), b.to(dev), c2, A_MC, B_MC, M, N, K, M, N, N, tile_hint=tile, splitk_hint=2)
    close(c2, want, atol=2e-5 * math.sqrt(K))


# ------------------------------------------------------------------------------- conv ----
CONV_CASES = [
    # B, Cin, Cout, H, k, stride, pad(t,b,l,r), upsample
    (2, 128, 128, 16, 3, 1, (1, 1, 1, 1), False),
    (2, 3, 128, 32, 3, 1, (1, 1, 1, 1), False),      # conv_in  (Cin=3 -> scalar gather path)
    (2, 128, 3, 32, 3, 1, (1, 1, 1, 1), False),      # conv_out (N=3)
    (3, 128, 128, 32, 3, 2, (0, 1, 0, 1), False),    # Downsample2D padding=0: F.pad(0,1,0,1)
    (2, 224, 224, 16, 3, 2, (1, 1, 1, 1), False),    # CelebA downsample padding=1, C=224
    (2, 256, 256, 4, 3, 1, (1, 1, 1, 1), True),      # Upsample2D fused
    (2, 384, 256, 16, 1, 1, (0, 0, 0, 0), False),    # conv_shortcut 1x1
    (1, 512, 256, 4, 3, 1, (1, 1, 1, 1), False),     # small M, long K -> split-K
    (5, 100, 60, 7, 3, 1, (1, 1, 1, 1), False),      # pruned-like odd widths / odd spatial
]


def conv_ref(x, w, b, stride, pad, upsample):
    x = x.double()
    if upsample:
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
    x = F.pad(x, (pad[2], pad[3], pad[0], pad[1]))
    return F.conv2d(x, w.double(), b.double() if b is not None else None, stride=stride)


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("tile,splitk", [(0, 0), (1, 0), (2, 2)])
def test_conv_fwd(ops, case, tile, splitk):
    B, Cin, Cout, H, k, stride, pad, ups = case
    x, w, b = rnd(B, Cin, H, H, seed=1), rnd(Cout, Cin, k, k, seed=2, scale=1 / math.sqrt(Cin * k * k)), rnd(Cout, seed=3)
    temb = rnd(B, Cout, seed=4)
    want = conv_ref(x, w, b, stride, pad, ups) + temb.double()[:, :, None, None]
    res = rnd(*want.shape, seed=5)
    want = want + res.double()
    y = ops.conv2d_fwd_raw(nhwc(x), cl_weight(w), b.to(dev), stride, pad, ups, rowadd=temb.to(dev), residual=nhwc(res),
                           tile_hint=tile, splitk_hint=splitk)
    close(y.permute(0, 3, 1, 2), want, atol=3e-5)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_bwd(ops, case):
    B, Cin, Cout, H, k, stride, pad, ups = case
    x = rnd(B, Cin, H, H, seed=1).double().requires_grad_(True)
    w = rnd(Cout, Cin, k, k, seed=2, scale=1 / math.sqrt(Cin * k * k)).double().requires_grad_(True)
    b = rnd(Cout, seed=3).double().requires_grad_(True)
    y = conv_ref(x, w, b, stride, pad, ups)
    dy = rnd(*y.shape, seed=6)
    y.backward(dy.double())
    dyg = nhwc(dy)
    wg = cl_weight(w.detach().float())
    if Cin % 4 == 0:
        dx = ops.conv2d_dgrad_raw(dyg, wg, (B, H, H, Cin), stride, pad, ups)
        close(dx.permute(0, 3, 1, 2), x.grad, atol=3e-5)
    dw = ops.conv2d_wgrad_raw(dyg, nhwc(x.detach().float()), wg, stride, pad, ups)
    close(dw, w.grad, atol=3e-5 * math.sqrt(B * y.shape[-1] * y.shape[-2] / 16))
    db = ops.colsum_raw(dyg.view(-1, Cout), 1).view(-1)
    close(db, b.grad, atol=1e-5 * math.sqrt(B * y.shape[-1] * y.shape[-2]))


@pytest.mark.parametrize("B,Cin,Cout,H,ups", [(8, 64, 96, 32, False), (4, 96, 192, 16, False), (4, 192, 288, 8, False),
                                              (2, 64, 672, 16, False), (4, 96, 192, 16, True), (16, 288, 96, 32, False),
                                              (2, 64, 96, 64, False), (3, 32, 192, 32, True), (4, 64, 160, 32, False),
                                              (2, 96, 64, 32, False), (2, 64, 32, 32, False), (2, 32, 320, 16, False)])
def test_wgrad_patch_kernel_96_channel_tiles(ops, B, Cin, Cout, H, ups):
    """Weight gradient of a 3x3 convolution whose output-channel count is a multiple of 96 but not of 128 (the pruned
    widths 96 / 192 that unlearn.py:363-367 fine-tunes, 288, CelebA's 672): the patch kernel deals the 3 channel groups x
    9 taps of a 96-channel tile 7 / 7 / 7 / 6 to its four waves instead of idling a wave in a 128-channel tile.  Against
    fp64 autograd and against the 128-channel instance (tile_hint = 1) on the same inputs."""
    x = rnd(B, Cin, H, H, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=0.05)
    He = 2 * H if ups else H
    dy = rnd(B, Cout, He, He, seed=6)
    xd = x.double().requires_grad_(True)
    wd = w.double().requires_grad_(True)
    conv_ref(xd, wd, None, 1, (1, 1, 1, 1), ups).backward(dy.double())
    ops.PROFILER = prof = ops.GemmProfiler()
    try:
        dw = ops.conv2d_wgrad_raw(nhwc(dy), nhwc(x), cl_weight(w), 1, (1, 1, 1, 1), ups)
        dw128 = ops.conv2d_wgrad_raw(nhwc(dy), nhwc(x), cl_weight(w), 1, (1, 1, 1, 1), ups, tile_hint=1)
        torch.cuda.synchronize()
    finally:
        ops.PROFILER = None
    keys = list(prof.summary())
    # 96 / 64 / 32: narrow tiles (units dealt to the waves); 224: a 128-channel-tile launch + a narrower-tile launch over
    # disjoint channel rows (160 = 128 + 32, 320 = 256 + 64)
    assert [k[0] for k in keys] == [f"conv_wgrad_patch_w{He}"] * len(keys) and sorted(k[1] for k in keys)[0] in (32, 64, 96, 128) \
        and sorted(k[1] for k in keys)[1] in (128, 224), keys
    tol = 3e-5 * math.sqrt(B * He * He / 16)
    close(dw, wd.grad, atol=tol)
    close(dw, dw128, atol=tol)


@pytest.mark.parametrize("B,Cin,Cout,H,ups", [(2, 64, 224, 64, False), (2, 160, 160, 64, False), (1, 224, 128, 32, True), (3, 32, 64, 64, False)])
def test_wgrad_patch_kernel_64_wide_maps(ops, B, Cin, Cout, H, ups):
    """The 64x64 level of the CelebA-HQ LDM U-Net (ddpm_config.py:425-450: 224 channels, 160 pruned): a K step of the patch
    weight-gradient kernel is half an image row there (columns 0-31 / 32-63 alternate, halo columns from the same row).
    Against fp64 autograd and the im2col-gather kernel (no_patch)."""
    x = rnd(B, Cin, H, H, seed=1)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=0.05)
    He = 2 * H if ups else H
    dy = rnd(B, Cout, He, He, seed=6)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    conv_ref(xd, wd, None, 1, (1, 1, 1, 1), ups).backward(dy.double())
    ops.PROFILER = prof = ops.GemmProfiler()
    try:
        dw = ops.conv2d_wgrad_raw(nhwc(dy), nhwc(x), cl_weight(w), 1, (1, 1, 1, 1), ups)
        torch.cuda.synchronize()
    finally:
        ops.PROFILER = None
    assert [k[0] for k in prof.summary()] == ["conv_wgrad_patch_w64"], list(prof.summary())
    with ops.kernel_flags(no_patch=True):
        dw_gather = ops.conv2d_wgrad_raw(nhwc(dy), nhwc(x), cl_weight(w), 1, (1, 1, 1, 1), ups)
    tol = 3e-5 * math.sqrt(B * He * He / 16)
    close(dw, wd.grad, atol=tol)
    close(dw, dw_gather, atol=tol)


@pytest.mark.parametrize("B,Cin,Cout,H", [(32, 32, 224, 64), (64, 32, 448, 32), (8, 64, 224, 32)])
def test_conv_patch_split_channel_ranges_224_448(ops, B, Cin, Cout, H):
    """CelebA-HQ LDM widths (ddpm_config.py:425-450): 224 = 128 + 96 and 448 = 2 x 128 + 2 x 96 output channels run as two
    launches over disjoint column ranges (128-wide tiles, then 96-wide tiles) instead of padding an eighth of the MFMA
    work - forward with the fused bias + time-embedding + residual epilogue, and the weight gradient's channel rows.
    Same products in the same order as the single launch: bit-identical to tile_hint = 1 (128-wide tiles everywhere)."""
    x, w, b = rnd(B, Cin, H, H, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=0.05), rnd(Cout, seed=3)
    temb, res = rnd(B, Cout, seed=4), rnd(B, Cout, H, H, seed=5)
    want = conv_ref(x, w, b, 1, (1, 1, 1, 1), False) + temb.double()[:, :, None, None] + res.double()
    ops.PROFILER = prof = ops.GemmProfiler()
    try:
        with ops.kernel_flags(no_wino=True):           # the direct LDS-patch kernels (the default here is the Winograd route)
            y = ops.conv2d_fwd_raw(nhwc(x), cl_weight(w), b.to(dev), 1, (1, 1, 1, 1), False, rowadd=temb.to(dev), residual=nhwc(res))
        torch.cuda.synchronize()
    finally:
        ops.PROFILER = None
    # the planner splits where two exact launches model faster than one padded one (round quantisation counts): the
    # CelebA-sized launches do (tile code 224 = 128-wide then 96-wide tiles), the small one runs a single width
    keys = list(prof.summary())
    assert [k[0] for k in keys] == [f"conv_fwd_patch_w{H}"] and keys[0][1] == (224 if B >= 32 else 96), keys
    y128 = ops.conv2d_fwd_raw(nhwc(x), cl_weight(w), b.to(dev), 1, (1, 1, 1, 1), False, rowadd=temb.to(dev), residual=nhwc(res), tile_hint=1)
    close(y.permute(0, 3, 1, 2), want, atol=3e-5)
    assert torch.equal(y, y128)
    dy = rnd(B, Cout, H, H, seed=6)
    xd, wd = x.double().requires_grad_(True), w.double().requires_grad_(True)
    conv_ref(xd, wd, None, 1, (1, 1, 1, 1), False).backward(dy.double())
    dw = ops.conv2d_wgrad_raw(nhwc(dy), nhwc(x), cl_weight(w), 1, (1, 1, 1, 1), False)
    close(dw, wd.grad, atol=3e-5 * math.sqrt(B * H * H / 16))


WINO_CASES = [
    # B, Cin, Cout, H, W, upsample, epilogue
    (8, 64, 128, 32, 32, False, True),         # 64 tiles x 128 channels
    (6, 96, 192, 16, 16, False, True),         # pruned widths: 128 tiles x 64 channels
    (3, 256, 256, 8, 8, True, True),           # behind the fused nearest-2x upsample
    (5, 32, 68, 34, 30, False, True),          # tiles per image / per row not powers of two, ragged channels; F(2x2) only (30 % 4)
    (2, 96, 68, 20, 12, False, True),          # the same for 4x4 tiles
    (4, 128, 320, 16, 16, False, False),       # SD width, no epilogue terms
    (1, 64, 1024, 64, 64, False, True),        # one image, many channel blocks
    (1, 32, 64, 4, 4, False, True),            # a single 4x4 tile / four 2x2 tiles
    (176, 64, 128, 32, 32, False, True),       # F(4x4): >= 1024 blocks -> the six-position product kernel (24 half-transformed panels)
    (40, 96, 192, 32, 32, True, True),         #         the same on 128 x 64 blocks, behind the upsample
]


@pytest.mark.parametrize("form", [2, 4, 9, 10, 11])
@pytest.mark.parametrize("B,Cin,Cout,H,W,ups,epi", WINO_CASES)
def test_conv_fwd_winograd(ops, B, Cin, Cout, H, W, ups, epi, form):
    """Winograd routes against fp64 - with bias, the time-embedding row and the residual in the epilogue - and against the
    direct kernels on the same inputs.  F(2x2, 3x3): wino_input_kernel + wino_gemm_kernel on gad_wino_weights' U (tile_hint 7
    forces it); F(4x4, 3x3) on gad_wino4_weights' U: the planner's form (tile_hint 8), wino4_input_kernel + wino4_fused_kernel -
    all 36 products and the whole output transform in one launch, on 32-tile blocks / two workgroups per CU (tile_hint 9) or 64-tile
    blocks / one per CU (tile_hint 11) -, or the three-launch forms wino4_input_kernel +
    products (36 batched on the generic engine, or the six-position kernel) + wino4_output_kernel (tile_hint 10).  Same tolerance
    as every fp32 contraction (F(4x4) measures ~4e-6 of the output scale, a decimal digit more than the direct kernels), not
    bit-identical, deterministic."""
    if form != 2 and ((H * (2 if ups else 1)) % 4 or (W * (2 if ups else 1)) % 4):
        pytest.skip("F(4x4) needs output maps that are multiples of 4")
    x, w, b = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=1 / math.sqrt(Cin * 9)), rnd(Cout, seed=3)
    want = conv_ref(x, w, b if epi else None, 1, (1, 1, 1, 1), ups)
    temb, res = rnd(B, Cout, seed=4), rnd(*want.shape, seed=5)
    if epi:
        want = want + temb.double()[:, :, None, None] + res.double()
    kw = dict(rowadd=temb.to(dev), residual=nhwc(res)) if epi else {}
    xg, wg, bg = nhwc(x), cl_weight(w), b.to(dev) if epi else None
    hint = {2: 7, 4: 8, 9: 9, 10: 10, 11: 11}[form]
    ops.PROFILER = prof = ops.GemmProfiler()
    try:
        y = ops.conv2d_fwd_raw(xg, wg, bg, 1, (1, 1, 1, 1), ups, tile_hint=hint, **kw)
        torch.cuda.synchronize()
    finally:
        ops.PROFILER = None
    assert [k[0] for k in prof.summary()] == ["conv_fwd_wino" if form == 2 else "conv_fwd_wino4"], list(prof.summary())
    if form in (9, 10, 11):
        want_tile = {9: 32 if Cout % 64 == 0 or (Cout + 31) // 32 * 32 == (Cout + 63) // 64 * 64 else 65, 10: 128, 11: 64}[form]
        assert [k[1] for k in prof.summary()] == [want_tile], list(prof.summary())       # tile 32 / 65 / 64 name the one-launch forms (65: 64 tiles x 32 channels)
    y = ops.conv2d_fwd_raw(xg, wg, bg, 1, (1, 1, 1, 1), ups, tile_hint=hint, **kw)
    close(y.permute(0, 3, 1, 2), want, atol=3e-5)
    with ops.kernel_flags(no_wino=True):
        y0 = ops.conv2d_fwd_raw(xg, wg, bg, 1, (1, 1, 1, 1), ups, **kw)
    assert not torch.equal(y, y0)
    close(y, y0, rtol=3e-5, atol=3e-5)
    assert torch.equal(y, ops.conv2d_fwd_raw(xg, wg, bg, 1, (1, 1, 1, 1), ups, tile_hint=hint, **kw))        # deterministic


@pytest.mark.parametrize("B,Cin,Cout,H,W,ups", [(4, 64, 128, 16, 16, False), (3, 96, 64, 8, 8, True), (2, 32, 68, 16, 12, False),
                                                 (8, 128, 128, 32, 32, False), (2, 320, 320, 8, 8, False)])
def test_conv_wgrad_winograd(ops, B, Cin, Cout, H, W, ups):
    """Weight gradient in Winograd F(4x4, 3x3) form (wino4_dy_kernel + wino4_input_kernel + 36 batched products + wino4_dw_kernel;
    tile_hint 8 forces it) against fp64 autograd and against the direct kernels, into a fresh tensor and into a caller-owned
    [Cout, Cin, 3, 3] channels_last slot (the flat gradient buffer's)."""
    x = rnd(B, Cin, H, W, seed=1).double().requires_grad_(False)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=0.05).double().requires_grad_(True)
    y = conv_ref(x, w, None, 1, (1, 1, 1, 1), ups)
    dy = rnd(*y.shape, seed=6)
    y.backward(dy.double())
    xg, dyg, wg = nhwc(x.float()), nhwc(dy), cl_weight(w.detach().float())
    ops.PROFILER = prof = ops.GemmProfiler()
    try:
        dw = ops.conv2d_wgrad_raw(dyg, xg, wg, 1, (1, 1, 1, 1), ups, tile_hint=8)
        torch.cuda.synchronize()
    finally:
        ops.PROFILER = None
    assert [k[0] for k in prof.summary()] == ["conv_wgrad_wino4"], list(prof.summary())
    dw = ops.conv2d_wgrad_raw(dyg, xg, wg, 1, (1, 1, 1, 1), ups, tile_hint=8)
    He = H * (2 if ups else 1)
    close(dw, w.grad, atol=3e-5 * math.sqrt(B * He * He / 16))
    with ops.kernel_flags(no_wino=True):
        dw0 = ops.conv2d_wgrad_raw(dyg, xg, wg, 1, (1, 1, 1, 1), ups)
    assert not torch.equal(dw, dw0)
    close(dw, dw0, rtol=1e-4, atol=1e-4)
    slot = torch.full((Cout, Cin, 3, 3), 7.0, device=dev).contiguous(memory_format=torch.channels_last)
    ops.conv2d_wgrad_raw(dyg, xg, wg, 1, (1, 1, 1, 1), ups, tile_hint=8, out=slot)
    assert torch.equal(slot, dw)


@pytest.mark.parametrize("B,Cin,Cout,H,W,ups,hint", [(4, 64, 128, 16, 16, False, 9), (3, 96, 64, 8, 8, True, 10), (8, 128, 128, 32, 32, False, 8),
                                                      (2, 32, 68, 16, 12, False, 10)])
def test_conv_wgrad_winograd_takes_the_forward_input_image(ops, B, Cin, Cout, H, W, ups, hint):
    """Training: the F(4x4) forward launch of a convolution keeps its transformed input V (`conv2d_fwd_raw(keep_v=)`, the start
    of the route's scratch in the one-launch and three-launch forms alike) and the weight gradient of the same convolution reads
    it (`GAD_GEMM_WINO_SKIP_INPUT` + `B_wino4` = V) instead of transforming x again: same kernels on the same image, bit-identical;
    a forward launch on any other route keeps nothing; the autograd function does both by itself."""
    x, w = rnd(B, Cin, H, W, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=0.05)
    xg, wg = nhwc(x), cl_weight(w)
    keep = []
    y = ops.conv2d_fwd_raw(xg, wg, None, 1, (1, 1, 1, 1), ups, tile_hint=hint, keep_v=keep)
    assert len(keep) == 1 and keep[0].numel() >= 36 * (y.shape[0] * y.shape[1] * y.shape[2] // 16) * Cin * 4
    dyg = nhwc(rnd(B, Cout, y.shape[1], y.shape[2], seed=6))
    dw = ops.conv2d_wgrad_raw(dyg, xg, wg, 1, (1, 1, 1, 1), ups, tile_hint=8)
    ops.PROFILER = prof = ops.GemmProfiler()
    try:
        dw_v = ops.conv2d_wgrad_raw(dyg, xg, wg, 1, (1, 1, 1, 1), ups, tile_hint=8, wino_v=keep[0])
        torch.cuda.synchronize()
    finally:
        ops.PROFILER = None
    assert [k[0] for k in prof.summary()] == ["conv_wgrad_wino4"]
    assert torch.equal(dw_v, dw)
    poisoned = torch.full_like(xg, float("nan"))                 # with V given the activation itself is not read
    assert torch.equal(ops.conv2d_wgrad_raw(dyg, poisoned, wg, 1, (1, 1, 1, 1), ups, tile_hint=8, wino_v=keep[0]), dw)
    none = []
    with ops.kernel_flags(no_wino=True):
        ops.conv2d_fwd_raw(xg, wg, None, 1, (1, 1, 1, 1), ups, keep_v=none)
    assert none == []
    # through autograd (planner's routes): the same gradient with and without the kept image
    grads = []
    for on in (True, False):
        ops.KEEP_WINO_V[0] = on
        try:
            wp = wg.clone().requires_grad_(True)
            ops.conv2d(xg, wp, None, None, None, 1, (1, 1, 1, 1), ups).backward(dyg)
            grads.append(wp.grad)
        finally:
            ops.KEEP_WINO_V[0] = True
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("B,C,Cout,H,W,G,bypass", [(8, 128, 128, 32, 32, 32, True), (4, 64, 128, 16, 16, 16, False), (3, 256, 256, 8, 8, 32, True),
                                                    (2, 32, 64, 6, 6, 8, False), (2, 128, 128, 64, 64, 32, True)])
def test_gn_silu_conv3x3_training_node(ops, B, C, Cout, H, W, G, bypass):
    """ResnetBlock2D's training halves as ONE autograd node (ops.GnSiluConv3x3Fn: GroupNorm writes the Winograd route's transformed
    input, kept for the weight gradient; the normalised activation never exists) against the two separate nodes: every gradient
    (input incl. the bypass alias's, norm affine, weight, bias, time-embedding row, residual) bit for bit by default (same launches;
    the weight gradient reads the forward's kept image) and with the kept-image path off, to fp32 rounding with GroupNorm writing the
    image itself (`GAD_TRAIN_GN_WINO=1`)."""
    dt = dict(device=dev)
    x0, w0 = nhwc(rnd(B, C, H, W, seed=1)), cl_weight(rnd(Cout, C, 3, 3, seed=2, scale=1 / math.sqrt(9 * C)))
    g0, b0, bias0 = (1 + 0.1 * rnd(C, seed=3)).to(dev), rnd(C, seed=4).to(dev), rnd(Cout, seed=5).to(dev)
    row0, res0 = rnd(B, Cout, seed=6).to(dev), nhwc(rnd(B, Cout, H, W, seed=7))
    dy, da = nhwc(rnd(B, Cout, H, W, seed=8)), nhwc(rnd(B, C, H, W, seed=9))

    def run(fused):
        leaves = [t.clone().requires_grad_(True) for t in (x0, g0, b0, w0, bias0, row0, res0)]
        x, g, b, w, bias, row, res = leaves
        xin = x * 1.0                                    # (a non-leaf input, as inside the U-Net)
        if fused:
            out = ops.gn_silu_conv3x3(xin, g, b, G, 1e-5, w, bias, rowadd=row, residual=res, bypass=bypass)
            y, alias = out if bypass else (out, None)
        else:
            if bypass:
                h, alias = ops.group_norm_bypass(xin, g, b, G, 1e-5, True)
            else:
                h, alias = ops.group_norm(xin, g, b, G, 1e-5, True), None
            y = ops.conv2d(h, w, bias, row, res)
        loss = (y * dy).sum() + ((alias * da).sum() if bypass else 0.0)
        return [y.detach(), *torch.autograd.grad(loss, leaves)]

    want = run(False)
    for a, b_ in zip(run(True), want):                   # default: the norm's own launch, the kept image for the weight gradient
        assert torch.equal(a, b_)
    ops.TRAIN_GN_WINO[0] = True                          # GroupNorm writes the image itself
    try:
        got = run(True)
    finally:
        ops.TRAIN_GN_WINO[0] = False
    for a, b_ in zip(got, want):
        close(a, b_)
    ops.KEEP_WINO_V[0] = False
    try:
        same = run(True)
    finally:
        ops.KEEP_WINO_V[0] = True
    for a, b_ in zip(same, want):
        assert torch.equal(a, b_)


def test_winograd_planner_takes_the_large_launches(ops):
    """The planner's modelled times against the direct plan's: maps that are multiples of 4 go to F(4x4) down to small
    launches, other even maps to F(2x2) when the launch is large, tiny launches stay direct."""
    def route(B, H, W, Cin, Cout):
        x, w = torch.zeros(B, H, W, Cin, device=dev), cl_weight(rnd(Cout, Cin, 3, 3, seed=2, scale=0.05))
        ops.PROFILER = prof = ops.GemmProfiler()
        try:
            ops.conv2d_fwd_raw(x, w, None)
            torch.cuda.synchronize()
        finally:
            ops.PROFILER = None
        return [k[0].replace("conv_fwd_", "").split("_")[0] for k in prof.summary()]
    assert route(128, 32, 32, 128, 128) == ["wino4"] and route(128, 16, 16, 256, 256) == ["wino4"] and route(1024, 8, 8, 256, 256) == ["wino4"]
    assert route(16, 64, 64, 320, 320) == ["wino4"] and route(16, 8, 8, 1280, 1280) == ["wino4"] and route(128, 8, 8, 256, 256) == ["wino4"]
    assert route(128, 34, 30, 128, 128) == ["wino"]
    assert route(1, 8, 8, 64, 64)[0] not in ("wino", "wino4") and route(2, 18, 18, 64, 64)[0] not in ("wino", "wino4")


def test_winograd_is_not_taken_where_it_does_not_apply(ops):
    """Small launches, stride 2, two-source gathers, forced tiles, bf16-operand mode and Cin % 32 != 0 stay on the direct
    kernels: bit-identical to no_wino."""
    import gad
    cases = [dict(B=2, Cin=128, Cout=128, H=16), dict(B=64, Cin=48, Cout=128, H=32), dict(B=64, Cin=64, Cout=128, H=32, stride=2),
             dict(B=64, Cin=64, Cout=128, H=32, tile=1), dict(B=64, Cin=64, Cout=128, H=32, bf16=True), dict(B=64, Cin=64, Cout=32, H=32)]
    for c in cases:
        x = nhwc(rnd(c["B"], c["Cin"], c["H"], c["H"], seed=1))
        w = cl_weight(rnd(c["Cout"], c["Cin"], 3, 3, seed=2, scale=0.05))
        run = lambda: ops.conv2d_fwd_raw(x, w, None, c.get("stride", 1), (1, 1, 1, 1), False, tile_hint=c.get("tile", 0))
        with gad.operand_precision("bf16" if c.get("bf16") else "f32"):
            y = run()
            with ops.kernel_flags(no_wino=True):
                y0 = run()
        assert torch.equal(y, y0), c


def test_winograd_shadow_of_a_flat_buffer_follows_the_weights(ops):
    """Weights in a flat parameter buffer: ONE gad_wino_weights launch transforms every eligible 3x3 weight (and, for the
    data gradients, every weight of the rotated shadow); forward and data gradient through Conv2dFn equal fp64 autograd
    before and after an optimizer-style rewrite (epoch bump) and a write through torch (version bump of one parameter)."""
    from gad.training import flatten_params
    shapes = [(128, 128, False, 32), (256, 128, True, 16), (128, 256, False, 32), (128, 3, False, 32)]     # Cout, Cin, upsample, H
    ws = [torch.nn.Parameter(rnd(co, ci, 3, 3, seed=10 + i, scale=0.05).to(dev).contiguous(memory_format=torch.channels_last))
          for i, (co, ci, _, _) in enumerate(shapes)]
    extra = torch.nn.Parameter(rnd(77, seed=20).to(dev))
    flat, _ = flatten_params([ws[0], extra] + ws[1:])
    Bn = 128

    def check_all():
        for w, (co, ci, up, H) in zip(ws[:3], shapes[:3]):
            x = rnd(Bn, H, H, ci, seed=3).to(dev).requires_grad_(True)
            ops.PROFILER = prof = ops.GemmProfiler()
            try:
                y = ops.Conv2dFn.apply(x, w, None, None, None, 1, (1, 1, 1, 1), up)
                dy = rnd(*y.shape, seed=4).to(dev)
                y.backward(dy)
                torch.cuda.synchronize()
            finally:
                ops.PROFILER = None
            summ = prof.summary()
            assert sum(v["launches"] for k, v in summ.items() if k[0].startswith("conv_fwd_wino")) == 2, list(summ)   # forward and data gradient
            xr = x.detach().permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
            xe = F.interpolate(xr, scale_factor=2.0, mode="nearest") if up else xr
            yr = F.conv2d(xe, w.detach().double().cpu(), padding=1)
            yr.backward(dy.permute(0, 3, 1, 2).double().cpu())
            y = ops.Conv2dFn.apply(x, w, None, None, None, 1, (1, 1, 1, 1), up)
            x.grad = None
            y.backward(dy)
            close(y.permute(0, 3, 1, 2), yr, atol=3e-5)
            close(x.grad, xr.grad.permute(0, 2, 3, 1), atol=3e-5)
            w.grad = None

    check_all()
    U = flat._gad_wino4[1]                                                      # 32 x 32 maps: the F(4x4) form
    assert ws[3]._gad_flat[1] not in flat._gad_wino4[2]                         # Cin = 3: not transformed
    d0, dn = flat._gad_wino4[2][ws[1]._gad_flat[1]]
    G = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6],
                      [0, 0, 1]], dtype=torch.float64)
    want = torch.einsum("ar,orsi,bs->aboi", G, ops.weight_krsc(ws[1]).detach().double().cpu(), G).reshape(36, 256, 128)
    close(U[d0:d0 + dn].view(36, 256, 128), want, rtol=1e-6, atol=1e-6)
    with torch.no_grad():
        flat.detach().mul_(1.5)
    flat._gad_epoch = getattr(flat, "_gad_epoch", 0) + 1
    check_all()
    with torch.no_grad():
        ws[2].copy_(ws[2] * 0.25)
    check_all()
    assert flat._gad_wino4[1] is U


def test_conv_autograd_function(ops):
    B, Cin, Cout, H = 2, 64, 96, 8
    x, w, b, t, r = rnd(B, Cin, H, H, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=0.05), rnd(Cout, seed=3), rnd(B, Cout, seed=4), rnd(B, Cout, H, H, seed=5)
    leaves = [v.double().requires_grad_(True) for v in (x, w, b, t, r)]
    y = F.conv2d(leaves[0], leaves[1], leaves[2], padding=1) + leaves[3][:, :, None, None] + leaves[4]
    dy = rnd(*y.shape, seed=7)
    y.backward(dy.double())
    gx, gw, gb, gt, gr = (nhwc(x).requires_grad_(True), cl_weight(w).requires_grad_(True), b.to(dev).requires_grad_(True),
                          t.to(dev).requires_grad_(True), nhwc(r).requires_grad_(True))
    out = ops.conv2d(gx, gw, gb, gt, gr)
    out.backward(nhwc(dy))
    close(out.permute(0, 3, 1, 2), y)
    close(gx.grad.permute(0, 3, 1, 2), leaves[0].grad)
    close(gw.grad, leaves[1].grad, atol=1e-4)
    close(gb.grad, leaves[2].grad, atol=1e-4)
    close(gt.grad, leaves[3].grad, atol=1e-4)
    close(gr.grad.permute(0, 3, 1, 2), leaves[4].grad)


# -------------------------------------------------------------------------- groupnorm ----
@pytest.mark.parametrize("B,C,H,G,silu", [(2, 128, 32, 32, True), (3, 256, 16, 32, True), (2, 384, 16, 32, True),
                                          (2, 512, 4, 32, True), (2, 224, 8, 32, False), (1, 896, 8, 32, True),
                                          (32, 128, 32, 32, True), (2, 1280, 8, 32, True), (1, 2560, 4, 32, True),
                                          (2, 384, 32, 32, True), (2, 96, 8, 32, True), (3, 256, 32, 32, False),
                                          (2, 320, 64, 32, True),
                                          # channels per group not a multiple of 4: the per-channel one-pass plan - pruned CIFAR
                                          # widths (3, 6, 9 per group), CelebA (7), SD (10, 20); B % 8 == 0 lets slabs that are
                                          # not whole 128-B lines go to one XCD per image
                                          (8, 96, 32, 32, True), (16, 192, 16, 32, True), (8, 288, 16, 32, True),
                                          (8, 288, 32, 32, True), (8, 224, 16, 32, False), (8, 320, 32, 32, True),
                                          (8, 640, 16, 32, True), (3, 192, 8, 32, True), (8, 96, 16, 32, True)])
@pytest.mark.parametrize("two_pass", [False, True])
def test_groupnorm_fwd_bwd(ops, B, C, H, G, silu, two_pass):
    """Forward has two plans: one pass with the (image, channel slab) held in registers, or stats + apply passes
    (odd channels per group, slabs that are not whole 128-B lines, too many pixels); both against fp64."""
    with ops.kernel_flags(gn_two_pass=two_pass):
        _groupnorm_case(ops, B, C, H, G, silu)


def _groupnorm_case(ops, B, C, H, G, silu):
    x = (rnd(B, C, H, H, seed=1) * 2 + 0.7).double().requires_grad_(True)   # non-zero mean stresses the variance
    ga, be = (rnd(C, seed=2) * 0.3 + 1).double().requires_grad_(True), (rnd(C, seed=3) * 0.2).double().requires_grad_(True)
    eps = 1e-6
    y = F.group_norm(x, G, ga, be, eps)
    if silu:
        y = F.silu(y)
    dy = rnd(B, C, H, H, seed=4)
    y.backward(dy.double())
    gx = nhwc(x.detach().float()).requires_grad_(True)
    gg, gb = ga.detach().float().to(dev).requires_grad_(True), be.detach().float().to(dev).requires_grad_(True)
    out = ops.group_norm(gx, gg, gb, G, eps, silu)
    out.backward(nhwc(dy))
    close(out.permute(0, 3, 1, 2), y, atol=2e-5)
    close(gx.grad.permute(0, 3, 1, 2), x.grad, atol=5e-5)
    close(gg.grad, ga.grad, atol=2e-5 * math.sqrt(B * H * H))
    close(gb.grad, be.grad, atol=2e-5 * math.sqrt(B * H * H))


# -------------------------------------------------------------------------- attention ----
@pytest.mark.parametrize("B,T,heads,d", [(2, 256, 1, 256), (3, 16, 1, 256), (2, 64, 7, 32), (1, 1024, 2, 32)])
def test_attention_core(ops, B, T, heads, d):
    C = heads * d
    q, k, v = (rnd(B, T, C, seed=s, scale=0.5).double().requires_grad_(True) for s in (1, 2, 3))

    def split(t):
        return t.view(B, T, heads, d).transpose(1, 2)
    w = torch.softmax(split(q) @ split(k).transpose(-1, -2) / math.sqrt(d), dim=-1)
    o = (w @ split(v)).transpose(1, 2).reshape(B, T, C)
    do = rnd(B, T, C, seed=4)
    o.backward(do.double())
    gq, gk, gv = (t.detach().float().to(dev).requires_grad_(True) for t in (q, k, v))
    out = ops.attention_core(gq, gk, gv, heads)
    out.backward(do.to(dev))
    close(out, o, atol=3e-5)
    close(gq.grad, q.grad, atol=5e-5)
    close(gk.grad, k.grad, atol=5e-5)
    close(gv.grad, v.grad, atol=5e-5)


def test_linear_and_lora(ops):
    from gad import nn as gnn
    torch.manual_seed(0)
    lin = gnn.Linear(320, 640).to(dev)
    lora = gnn.LoRALinearLayer(320, 640, rank=24).to(dev)
    with torch.no_grad():
        lora.up.weight.copy_(rnd(640, 24, seed=9, scale=0.1))
    lin.set_lora_layer(lora)
    x = rnd(6, 77, 320, seed=1)
    gx = x.to(dev).requires_grad_(True)
    y = lin(gx, scale=0.7)
    dy = rnd(6, 77, 640, seed=2)
    y.backward(dy.to(dev))
    xd = x.double().requires_grad_(True)
    W, b = lin.weight.detach().cpu().double().requires_grad_(True), lin.bias.detach().cpu().double()
    A, Bm = lora.down.weight.detach().cpu().double().requires_grad_(True), lora.up.weight.detach().cpu().double().requires_grad_(True)
    want = xd @ W.T + b + 0.7 * ((xd @ A.T) @ Bm.T)
    want.backward(dy.double())
    close(y, want)
    close(gx.grad, xd.grad)
    close(lin.weight.grad, W.grad, atol=1e-4)
    close(lora.down.weight.grad, A.grad, atol=1e-4)
    close(lora.up.weight.grad, Bm.grad, atol=1e-4)


@pytest.mark.parametrize("K,N,r,bias,res,rows", [(320, 320, 256, False, True, (4, 256)), (768, 640, 256, False, False, (3, 77)),
                                                 (1280, 1280, 64, True, True, (2, 64)), (320, 640, 4, True, False, (1, 100)),
                                                 (640, 320, 131, False, True, (2, 50)), (96, 64, 8, True, True, (2, 33))])
def test_fused_lora_linear_forward_backward(ops, K, N, r, bias, res, rows):
    """LoRACompatibleLinear with a LoRALinearLayer (train_text_to_image_lora.py:786-820): the K-concatenated launches
    (y = [x | s x A^T] [W | B]^T, dx = [dy | s dy B] [W ; A]) against fp64, at the SD shapes (q/out 320x320 r=256;
    cross-attention k/v 768 -> 640 on 77 tokens; 1280 wide) and a pruned ragged rank (131: not a multiple of 4 -> the
    two-launch route).  LoRA-only training (base frozen) and base-trainable gradients."""
    from gad import nn as gnn
    torch.manual_seed(0)
    lin = gnn.Linear(K, N, bias=bias).to(dev)
    lora = gnn.LoRALinearLayer(K, N, rank=r).to(dev)
    with torch.no_grad():
        lora.up.weight.copy_(rnd(N, r, seed=9, scale=0.1))
    lin.set_lora_layer(lora)
    assert ops.lora_fusable(K, N, r) == (r % 4 == 0)
    x = rnd(*rows, K, seed=1)
    residual = rnd(*rows, N, seed=3) if res else None
    dy = rnd(*rows, N, seed=2)
    xd = x.double().requires_grad_(True)
    W = lin.weight.detach().cpu().double().requires_grad_(True)
    A, Bm = lora.down.weight.detach().cpu().double().requires_grad_(True), lora.up.weight.detach().cpu().double().requires_grad_(True)
    want = xd @ W.T + 0.7 * ((xd @ A.T) @ Bm.T)
    if bias:
        want = want + lin.bias.detach().cpu().double()
    if res:
        want = want + residual.double()
    want.backward(dy.double())
    for freeze in (False, True):
        for p_ in lin.parameters(recurse=False):
            p_.requires_grad_(not freeze)
            p_.grad = None
        for p_ in lora.parameters():
            p_.grad = None
        gx = x.to(dev).requires_grad_(True)
        gres = residual.to(dev).requires_grad_(True) if res else None
        y = lin(gx, residual=gres, scale=0.7)
        y.backward(dy.to(dev))
        tol = 2e-5 * math.sqrt(K + r)
        close(y, want, atol=tol)
        close(gx.grad, xd.grad, atol=2e-5 * math.sqrt(N + r))
        close(lora.down.weight.grad, A.grad, atol=1e-4)
        close(lora.up.weight.grad, Bm.grad, atol=1e-4)
        if res:
            close(gres.grad, dy)
        if freeze:
            assert lin.weight.grad is None
        else:
            close(lin.weight.grad, W.grad, atol=1e-4)


def test_norm_backward_with_frozen_affine_parameters_and_wide_column_sums(ops):
    """LoRA training freezes every norm: GroupNorm / LayerNorm backward then return dx only and skip the reduction of
    the dgamma / dbeta partials (NULL pointers through the ABI); dx is bit-identical to the trainable case.  Column sums
    wider than 1024 (LayerNorm 2C = 2560, GEGLU 5120 / 10240) take the float4 path."""
    x = rnd(3, 8, 8, 320, seed=1).to(dev)
    dy = rnd(3, 8, 8, 320, seed=2).to(dev)
    outs = []
    for frozen in (False, True):
        ga = (rnd(320, seed=3) * 0.3 + 1).to(dev).requires_grad_(not frozen)
        be = (rnd(320, seed=4) * 0.2).to(dev).requires_grad_(not frozen)
        gx = x.clone().requires_grad_(True)
        ops.group_norm(gx, ga, be, 32, 1e-5, True).backward(dy)
        lx = x.view(-1, 320).clone().requires_grad_(True)
        ops.layer_norm(lx, ga, be, 1e-5).backward(dy.view(-1, 320))
        outs.append((gx.grad.clone(), lx.grad.clone(), ga.grad))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert outs[0][2] is not None and outs[1][2] is None
    for N in (2560, 5120, 1284, 1028):
        m = rnd(700, N, seed=N).to(dev)
        close(ops.colsum_raw(m, 1).view(-1), m.double().sum(0), atol=2e-4)
        close(ops.colsum_raw(m, 7), m.double().view(7, 100, N).sum(1), atol=1e-4)


# ------------------------------------------------------------------------ elementwise ----
@pytest.mark.parametrize("two_pass", [False, True])
@pytest.mark.parametrize("B,C,H,G,silu", [(4, 128, 16, 32, True), (2, 96, 32, 32, True), (3, 256, 8, 32, False), (2, 320, 16, 32, False)])
def test_groupnorm_bypass_sums_the_residual_gradient_in_its_backward_kernel(ops, B, C, H, G, silu, two_pass):
    """ResnetBlock2D / attention block input: x feeds the norm and the residual.  `group_norm_bypass` hands the residual
    branch the norm node's own alias of x, so autograd sees one consumer and the backward adds both gradients in the
    kernel's store (gad_groupnorm_args.dx_add).  Bit-identical to the plain node followed by autograd's add launch."""
    x0 = rnd(B, H, H, C, seed=1).to(dev)
    gamma = (1 + 0.1 * rnd(C, seed=2)).to(dev).requires_grad_(True)
    beta = (0.1 * rnd(C, seed=3)).to(dev).requires_grad_(True)
    wy, wr = rnd(B, H, H, C, seed=4).to(dev), rnd(B, H, H, C, seed=5).to(dev)
    res = []
    for bypass in (False, True):
        x = x0.clone().requires_grad_(True)
        gamma.grad = beta.grad = None
        with ops.kernel_flags(gn_two_pass=two_pass):
            if bypass:
                y, xr = ops.group_norm_bypass(x, gamma, beta, G, 1e-5, silu)
            else:
                y, xr = ops.group_norm(x, gamma, beta, G, 1e-5, silu), x
            ((y * wy).sum() + (xr * wr * 2.0).sum()).backward()
        res.append((y.detach(), x.grad.clone(), gamma.grad.clone(), beta.grad.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    # only the bypass output used downstream / no grad mode
    x = x0.clone().requires_grad_(True)
    y, xr = ops.group_norm_bypass(x, gamma, beta, G, 1e-5, silu)
    (xr * wr).sum().backward()
    assert torch.equal(x.grad, wr)
    with torch.no_grad(), ops.kernel_flags(gn_two_pass=two_pass):
        y2, xr2 = ops.group_norm_bypass(x0, gamma, beta, G, 1e-5, silu)
    assert xr2 is x0 and torch.equal(y2, res[0][0])


def test_layernorm_bypass_sums_the_residual_gradient_in_its_backward_kernel(ops):
    x0 = rnd(2, 77, 320, seed=1).to(dev)
    gamma, beta = (1 + 0.1 * rnd(320, seed=2)).to(dev).requires_grad_(True), (0.1 * rnd(320, seed=3)).to(dev).requires_grad_(True)
    wy, wr = rnd(2, 77, 320, seed=4).to(dev), rnd(2, 77, 320, seed=5).to(dev)
    res = []
    for bypass in (False, True):
        x = x0.clone().requires_grad_(True)
        gamma.grad = beta.grad = None
        y, xr = ops.layer_norm_bypass(x, gamma, beta) if bypass else (ops.layer_norm(x, gamma, beta), x)
        ((y * wy).sum() + (xr * wr).sum()).backward()
        res.append((y.detach(), x.grad.clone(), gamma.grad.clone(), beta.grad.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


def test_timestep_embedding(ops):
    from oracle.diffusers_ref import get_timestep_embedding
    t = torch.tensor([0, 1, 10, 500, 990, 999])
    for dim, flip, shift in [(128, False, 1), (224, True, 0)]:
        got = ops.timestep_embedding(t.to(dev), dim, flip, shift)
        want = get_timestep_embedding(t, dim, flip, shift)
        close(got, want, rtol=1e-5, atol=2e-4)   # sin/cos of arguments up to 999 rad in fp32


def test_ddim_add_noise_image(ops):
    from oracle.diffusers_ref import DDIMScheduler, DDPMScheduler
    sch = DDIMScheduler()
    sch.set_timesteps(100)
    x, e = rnd(4, 3, 32, 32, seed=1), rnd(4, 3, 32, 32, seed=2)
    for t in (990, 500, 10, 0):
        want = sch.step(e, t, x).prev_sample
        prev = t - 10
        a_t = sch.alphas_cumprod[t].item()
        a_p = sch.alphas_cumprod[prev].item() if prev >= 0 else 1.0
        got = ops.ddim_step_raw(x.to(dev), e.to(dev), a_t, a_p, 1.0)
        close(got, want, rtol=1e-5, atol=1e-5)
    dd = DDPMScheduler()
    ts = torch.tensor([0, 999, 500, 3])
    want = dd.add_noise(x, e, ts)
    got = ops.add_noise_raw(x.to(dev), e.to(dev), ts.to(dev), dd.alphas_cumprod.to(dev))
    close(got, want, rtol=1e-6, atol=1e-6)
    close(ops.to_image01_raw(x.to(dev)), (x / 2 + 0.5).clamp(0, 1), rtol=0, atol=1e-7)


def test_small_kernels(ops):
    x, y = rnd(3, 5, 7, 11, seed=1), rnd(3, 5, 7, 11, seed=2)
    loss, d = ops.mse_fwd_bwd_raw(x.to(dev), y.to(dev))
    xd = x.double().requires_grad_(True)
    l = F.mse_loss(xd, y.double())
    l.backward()
    close(loss, l.detach().view(1), rtol=1e-5, atol=1e-6)
    close(d, xd.grad, rtol=1e-5, atol=1e-7)
    gx = x.to(dev).requires_grad_(True)
    s = ops.silu(gx)
    s.backward(y.to(dev))
    xs = x.double().requires_grad_(True)
    F.silu(xs).backward(y.double())
    close(s, F.silu(x.double()), atol=1e-6)
    close(gx.grad, xs.grad, atol=1e-6)
    a, b = rnd(2, 4, 4, 128, seed=3), rnd(2, 4, 4, 64, seed=4)
    ga, gb = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    c = ops.concat(ga, gb)
    assert torch.equal(c.cpu(), torch.cat([a, b], -1))
    c.backward(c.detach() * 2)
    assert torch.equal(ga.grad.cpu(), a * 2) and torch.equal(gb.grad.cpu(), b * 2)
    z = rnd(3, 5, 6, 7, seed=5)
    assert torch.equal(ops.nchw_to_nhwc_raw(z.to(dev)).cpu(), z.permute(0, 2, 3, 1).contiguous())
    assert torch.equal(ops.nhwc_to_nchw_raw(z.to(dev)).cpu(), z.permute(0, 3, 1, 2).contiguous())
    m = rnd(6 * 50, 36, seed=6)
    close(ops.colsum_raw(m.to(dev), 6), m.view(6, 50, 36).double().sum(1), atol=1e-5)
    g = rnd(100003, seed=7)
    close(ops.sumsq_raw(g.to(dev)), (g.double() ** 2).sum().view(1), rtol=1e-5)


def test_fused_clip_adam_ema_matches_torch(ops):
    from oracle.diffusers_ref import EMAModel
    n = 10007
    p0, g_list = rnd(n, seed=1), [rnd(n, seed=10 + i, scale=0.05 * (i + 1)) for i in range(4)]
    pr = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.Adam([pr], lr=1e-4)
    ema = EMAModel([pr], decay=0.9999)
    ema.optimization_step = 5000
    p, m, v = p0.to(dev).clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    e = p0.to(dev).clone()
    for i, g in enumerate(g_list):
        pr.grad = g.double().clone()
        torch.nn.utils.clip_grad_norm_([pr], 1.0)
        opt.step()
        ema.step([pr])
        ss = ops.sumsq_raw(g.to(dev))
        ops.clip_adam_ema_raw(p, g.to(dev), m, v, e, ss, max_norm=1.0, lr=1e-4, betas=(0.9, 0.999), eps=1e-8,
                              weight_decay=0.0, adamw=False, step=i + 1, ema_decay=ema.cur_decay_value)
    close(p, pr.detach(), rtol=1e-6, atol=1e-6)
    close(e, ema.shadow_params[0], rtol=1e-6, atol=1e-6)


# ------------------------------------------------------------------------- whole U-Net ----
def _models(cfg_name="cifar100_config", shrink=None):
    from oracle import diffusers_ref as R
    from gad import nn as G
    from src.ddpm_config import DDPMConfig
    cfg = dict(getattr(DDPMConfig, cfg_name)["unet_config"])
    if shrink:
        cfg.update(shrink)
    torch.manual_seed(0)
    ref = R.UNet2DModel(**cfg)
    mine = G.UNet2DModel(**cfg)
    mine.load_state_dict(ref.state_dict())
    return ref, mine.to(dev), cfg


def test_unet_forward_backward_cifar(ops):
    ref, mine, cfg = _models()
    x, t = rnd(2, 3, 32, 32, seed=1), torch.tensor([7, 950])
    noise = rnd(2, 3, 32, 32, seed=2)
    want = ref(x, t).sample
    F.mse_loss(want, noise).backward()
    got = mine(x.to(dev), t.to(dev)).sample
    loss, d = ops.mse_fwd_bwd_raw(got.contiguous(), noise.to(dev))
    got.backward(d)
    close(got, want, atol=1e-4)
    gref = dict(ref.named_parameters())
    # to_k.bias gradients are analytically zero (softmax is shift invariant): compare every parameter's
    # error with the typical gradient norm rather than with its own (noise-level) norm
    typical = float(np.median([g.grad.norm().item() for g in gref.values()]))
    worst = 0.0
    for n, p in mine.named_parameters():
        a, b = p.grad.detach().cpu().double(), gref[n].grad.double()
        worst = max(worst, ((a - b).norm() / (b.norm() + 1e-3 * typical)).item())
    assert worst < 2e-3, worst


def test_unet_forward_celeba_like(ops):
    # CelebA topology (head_dim 32 -> multi-head, cpg = 7, symmetric downsample padding) at reduced width/size
    ref, mine, cfg = _models("celeba_config", dict(block_out_channels=[64, 128, 192, 224], sample_size=16))
    x, t = rnd(2, 3, 16, 16, seed=1), torch.tensor([3, 700])
    close(mine(x.to(dev), t.to(dev)).sample, ref(x, t).sample, atol=1e-4)


def test_ddim_sampling_matches_oracle(ops):
    from oracle import diffusers_ref as R
    from gad import pipelines as P
    ref, mine, cfg = _models()
    pr = R.DDPMPipeline(ref, R.DDIMScheduler())
    pm = P.DDPMPipeline(mine, P.DDIMScheduler()).to(dev)
    a = pr(batch_size=2, generator=torch.Generator().manual_seed(3), num_inference_steps=4).images
    b = pm(batch_size=2, generator=torch.Generator().manual_seed(3), num_inference_steps=4, output_type="numpy").images
    assert a.shape == b.shape == (2, 32, 32, 3)
    np.testing.assert_allclose(b, a, atol=2e-4)


def test_unet_forward_pruned_widths(ops):
    # widths produced by unconditional_generation/prune.py at ratio 0.3
    ref, mine, cfg = _models(shrink=dict(block_out_channels=[96, 192, 192, 192]))
    x, t = rnd(2, 3, 32, 32, seed=1), torch.tensor([3, 700])
    close(mine(x.to(dev), t.to(dev)).sample, ref(x, t).sample, atol=1e-4)


def test_unet_forward_backward_head_grouped_pruned_celeba_layout(ops):
    """A CelebA-topology model after unconditional_generation/prune.py (head-grouped q/k/v, prune.py:337-342): heads stay,
    the head dim shrinks (32 -> 23 at the real widths; 16 -> 12 here, plus an odd 16 -> 11 level), inner dim != stream
    width.  Forward and a parameter gradient against the oracle on the same pruned weights."""
    import gad
    from oracle import diffusers_ref as R
    from src.ddpm_config import DDPMConfig
    from unconditional_generation import prune as P
    cfg = dict(DDPMConfig.celeba_config["unet_config"], block_out_channels=[64, 128, 128, 128], attention_head_dim=16,
               norm_num_groups=16, sample_size=16, down_block_types=["AttnDownBlock2D"] * 4, up_block_types=["AttnUpBlock2D"] * 4)
    torch.manual_seed(0)
    big = R.UNet2DModel(**cfg)
    new_cfg, new_sd = P.prune_state_dict(cfg, {k: v.detach().clone() for k, v in big.state_dict().items()}, 0.3)
    assert new_cfg["attention_layout"] == [[4, 11], [8, 12], [8, 12], [8, 12]]
    ref, mine = R.UNet2DModel(**new_cfg), gad.UNet2DModel(**new_cfg)
    ref.load_state_dict(new_sd)
    mine.load_state_dict(new_sd)
    mine.to(dev)
    x, t = rnd(2, 3, 16, 16, seed=1), torch.tensor([3, 700])
    want = ref(x, t).sample
    got = mine(x.to(dev), t.to(dev)).sample
    close(got, want, atol=1e-4)
    dy = rnd(2, 3, 16, 16, seed=2)
    want.backward(dy)
    got.backward(dy.to(dev))
    for name in ("down_blocks.0.attentions.0.to_q.weight", "mid_block.attentions.0.to_v.weight", "up_blocks.1.attentions.2.to_out.0.weight"):
        gr, gm = dict(ref.named_parameters())[name].grad, dict(mine.named_parameters())[name].grad.cpu()
        assert ((gm - gr).norm() / gr.norm()).item() < 2e-3, name
    with torch.no_grad():                                      # sampling mode: inner != stream width -> three projections
        close(mine(x.to(dev), t.to(dev)).sample, want, atol=1e-4)


# ------------------------------------------------------------------- bf16-operand mode ----
def _bf16_round(t):
    return t.to(torch.bfloat16).to(torch.float64)


@pytest.mark.parametrize("case", [dict(B=2, Cin=64, Cout=96, H=16, k=3, stride=1, pad=(1, 1, 1, 1), ups=False),
                                  dict(B=3, Cin=128, Cout=128, H=8, k=3, stride=2, pad=(0, 1, 0, 1), ups=False),
                                  dict(B=2, Cin=32, Cout=64, H=8, k=3, stride=1, pad=(1, 1, 1, 1), ups=True),
                                  dict(B=2, Cin=96, Cout=40, H=8, k=1, stride=1, pad=(0, 0, 0, 0), ups=False),
                                  dict(B=1, Cin=36, Cout=20, H=5, k=3, stride=1, pad=(1, 1, 1, 1), ups=False),
                                  # whole-row tiles of 3x3/s1/p1 convs take the LDS-patch kernel (W = 32 and W = 16)
                                  dict(B=2, Cin=128, Cout=256, H=32, k=3, stride=1, pad=(1, 1, 1, 1), ups=False),
                                  dict(B=3, Cin=32, Cout=128, H=16, k=3, stride=1, pad=(1, 1, 1, 1), ups=False),
                                  dict(B=1, Cin=96, Cout=72, H=32, k=3, stride=1, pad=(1, 1, 1, 1), ups=False),
                                  dict(B=1, Cin=64, Cout=96, H=64, k=3, stride=1, pad=(1, 1, 1, 1), ups=False)])
@pytest.mark.parametrize("tile", [0, 1, 2])
def test_conv_fwd_bf16_operands(ops, case, tile):
    """operand_precision=1: x and w are rounded to bf16 (RNE) inside the kernel, products are exact in fp32 and
    accumulated in fp32 -> against an fp64 convolution of the bf16-rounded operands only the fp32 summation
    order differs (tight tolerance); bias / time-embedding add / residual stay fp32."""
    c = case
    x, w, b = rnd(c["B"], c["Cin"], c["H"], c["H"], seed=1), rnd(c["Cout"], c["Cin"], c["k"], c["k"], seed=2) * 0.1, rnd(c["Cout"], seed=3)
    xin = _bf16_round(x)
    if c["ups"]:
        xin = F.interpolate(xin, scale_factor=2.0, mode="nearest")
    pt, pb, pl, pr = c["pad"]
    want = F.conv2d(F.pad(xin, (pl, pr, pt, pb)), _bf16_round(w), b.double(), stride=c["stride"])
    temb = rnd(c["B"], c["Cout"], seed=4)
    res = rnd(*want.shape, seed=5)
    want = want + temb.double()[:, :, None, None] + res.double()
    with ops.operand_precision("bf16"):
        got = ops.conv2d_fwd_raw(nhwc(x), cl_weight(w), b.to(dev), c["stride"], c["pad"], c["ups"], rowadd=temb.to(dev),
                                 residual=nhwc(res), tile_hint=tile)
    K = c["Cin"] * c["k"] ** 2
    close(got.permute(0, 3, 1, 2), want, rtol=1e-5, atol=2e-6 * math.sqrt(K))
    # inference (no_grad): the weights come from a cached bf16 copy streamed by LDS-DMA - the same roundings and products
    with torch.no_grad(), ops.operand_precision("bf16"):
        wg = cl_weight(w)
        got_ng = ops.conv2d_fwd_raw(nhwc(x), wg, b.to(dev), c["stride"], c["pad"], c["ups"], rowadd=temb.to(dev),
                                    residual=nhwc(res), tile_hint=tile)
        assert torch.equal(got_ng, got)
        if c["k"] == 3 and c["Cin"] % 32 == 0:
            assert wg._gad_bf16[1].dtype == torch.bfloat16
            wg.mul_(2.0)                                       # in-place change -> the cached copy is rebuilt
            got2 = ops.conv2d_fwd_raw(nhwc(x), wg, None, c["stride"], c["pad"], c["ups"], tile_hint=tile)
            ref2 = ops.conv2d_fwd_raw(nhwc(x), cl_weight(w * 2), None, c["stride"], c["pad"], c["ups"], tile_hint=tile)
            assert torch.equal(got2, ref2)
    # and it is a different result from the fp32-operand kernel by about bf16 rounding, not more
    exact = ops.conv2d_fwd_raw(nhwc(x), cl_weight(w), b.to(dev), c["stride"], c["pad"], c["ups"], rowadd=temb.to(dev),
                               residual=nhwc(res), tile_hint=tile)
    d = (got - exact).abs().max().item()
    assert 0 < d < 2.0 ** -7 * 0.1 * 3 * math.sqrt(K)


def test_linear_and_attention_scores_bf16_operands(ops):
    x, w, b = rnd(300, 256, seed=1), rnd(192, 256, seed=2) * 0.1, rnd(192, seed=3)
    want = _bf16_round(x) @ _bf16_round(w).T + b.double()
    with ops.operand_precision("bf16"):
        got = ops.linear_fwd_raw(x.to(dev), w.to(dev), b.to(dev))
        assert ops.OPERAND_PRECISION[0] >= 1          # ("bf16" selects half-precision activations where a model has them: mode 2; fp32 tensors take the bf16-operand kernels either way)
    assert ops.OPERAND_PRECISION[0] == 0
    close(got, want, rtol=1e-5, atol=2e-6 * 16)
    # K = 27 (conv_in) has no 16-B aligned float4 path: the flag is a permission, the launch stays fp32-exact
    x3, w3 = rnd(2, 3, 8, 8, seed=5), rnd(16, 3, 3, 3, seed=6)
    with ops.operand_precision("bf16"):
        y3 = ops.conv2d_fwd_raw(nhwc(x3), cl_weight(w3), None)
    close(y3.permute(0, 3, 1, 2), F.conv2d(x3.double(), w3.double(), padding=1), atol=1e-5)


@pytest.mark.parametrize("M,N,K", [(200, 132, 100), (512, 256, 1024), (128, 64, 32)])
@pytest.mark.parametrize("tile", [1, 2])
def test_gemm_row_contiguous_layouts_bf16_operands(ops, M, N, K, tile):
    """[k][row]-stored operands in bf16 mode: their LDS image keeps the [k][row] order and the MFMA fragments are
    fetched with the transposing LDS read."""
    from gad._capi import A_KC, A_MC, B_MC
    a, b = rnd(M, K, seed=1), rnd(K, N, seed=2)
    want = _bf16_round(a) @ _bf16_round(b)
    with ops.operand_precision("bf16"):
        c = torch.empty(M, N, device=dev)
        ops.gemm_raw(a.to(dev), b.to(dev), c, A_KC, B_MC, M, N, K, K, N, N, tile_hint=tile)
        close(c, want, rtol=1e-5, atol=2e-6 * math.sqrt(K))
        c2 = torch.empty(M, N, device=dev)
        ops.gemm_raw(a.T.contiguous().to(dev), b.to(dev), c2, A_MC, B_MC, M, N, K, M, N, N, tile_hint=tile, splitk_hint=2)
        close(c2, want, rtol=1e-5, atol=2e-6 * math.sqrt(K))


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[1] % 4 == 0 and c[2] % 4 == 0])
def test_conv_bwd_bf16_operands(ops, case):
    """dgrad (transposed gather x weight read as [k=(tap,co)][ci]) and wgrad (dy^T x im2col columns) with bf16
    operands against fp64 on the bf16-rounded tensors."""
    B, Cin, Cout, H, k, stride, pad, ups = case
    x = _bf16_round(rnd(B, Cin, H, H, seed=1)).requires_grad_(True)
    w = _bf16_round(rnd(Cout, Cin, k, k, seed=2, scale=1 / math.sqrt(Cin * k * k))).requires_grad_(True)
    y = conv_ref(x, w, None, stride, pad, ups)
    dy = rnd(*y.shape, seed=6)
    y.backward(_bf16_round(dy))
    with ops.operand_precision("bf16"):
        dx = ops.conv2d_dgrad_raw(nhwc(dy), cl_weight(w.detach().float()), (B, H, H, Cin), stride, pad, ups)
        dw = ops.conv2d_wgrad_raw(nhwc(dy), nhwc(x.detach().float()), cl_weight(w.detach().float()), stride, pad, ups)
    close(dx.permute(0, 3, 1, 2), x.grad, rtol=1e-5, atol=3e-6 * math.sqrt(Cout * k * k))
    close(dw, w.grad, rtol=1e-5, atol=3e-6 * math.sqrt(B * y.shape[-1] * y.shape[-2]))


def test_unet_forward_backward_bf16_close_to_fp32(ops):
    """Whole U-Net, fwd + bwd with bf16 operands vs the fp32-operand engine: outputs within bf16 rounding noise
    (relative rms < 2e-2), parameter gradients correlated > 0.999."""
    import gad
    from src.ddpm_config import DDPMConfig
    cfg = dict(DDPMConfig.cifar100_config["unet_config"], block_out_channels=[32, 64, 64, 64], norm_num_groups=8)
    torch.manual_seed(0)
    net = gad.UNet2DModel(**cfg).to(dev)
    x, t, tgt = rnd(4, 3, 32, 32, seed=1).to(dev), torch.tensor([3, 500, 900, 41], device=dev), rnd(4, 3, 32, 32, seed=2).to(dev)
    outs, grads = {}, {}
    for prec in ("f32", "bf16"):
        net.zero_grad(set_to_none=True)
        with ops.operand_precision(prec):
            y = net(x, t).sample
            (y - tgt).square().mean().backward()
        outs[prec] = y.detach()
        grads[prec] = torch.cat([p.grad.flatten() for p in net.parameters()])
    rel = ((outs["bf16"] - outs["f32"]).norm() / outs["f32"].norm()).item()
    assert 0 < rel < 2e-2, rel
    cos = torch.nn.functional.cosine_similarity(grads["bf16"], grads["f32"], dim=0).item()
    assert cos > 0.999, cos


def test_bf16_shadow_follows_parameters_written_through_torch(ops):
    """A parameter living in the flat buffer (training.flatten_params) and rewritten THROUGH TORCH - load_state_dict,
    p.copy_(), a torch.optim step - bumps its own version counter, not the buffer's: the bf16 shadow of the buffer must
    re-cast that slice (a stale shadow would silently sample / train with the old weights).  flatten, bf16 forward,
    load new weights, forward again, compare with an unflattened model holding the new weights."""
    import gad
    from gad.training import flatten_params
    from src.ddpm_config import DDPMConfig
    cfg = dict(DDPMConfig.cifar100_config["unet_config"], block_out_channels=[32, 64, 64, 64], norm_num_groups=8)
    torch.manual_seed(0)
    net, other, plain = (gad.UNet2DModel(**cfg).to(dev) for _ in range(3))
    with torch.no_grad():
        for p in other.parameters():
            p.mul_(1.5).add_(0.01)
    plain.load_state_dict(other.state_dict())
    flatten_params(list(net.parameters()))
    x, t = rnd(2, 3, 32, 32, seed=1).to(dev), torch.tensor([3, 900], device=dev)
    with torch.no_grad(), ops.operand_precision("bf16"):
        y_old = net(x, t).sample.clone()
        net.load_state_dict(other.state_dict())                    # writes the flat storage through the parameters
        y_new = net(x, t).sample
        want = plain(x, t).sample
        assert (y_new - y_old).abs().max().item() > 1e-3           # the weights did change
        assert torch.equal(y_new, want), (y_new - want).abs().max().item()
        w0 = next(p for p in net.parameters() if p.ndim == 4 and p.shape[1] % 32 == 0)
        w0.copy_(w0 * 0.5)                                         # a single parameter, in place
        plain.load_state_dict(net.state_dict())
        assert torch.equal(net(x, t).sample, plain(x, t).sample)


# -------------------------------------------------- channel concat read in place (up blocks) ----
@pytest.mark.parametrize("C1,C2,H,k", [(128, 128, 16, 1), (256, 128, 8, 1), (64, 32, 8, 3), (256, 256, 4, 3)])
@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_conv_two_source_equals_concat(ops, C1, C2, H, k, prec):
    """conv(cat([x, x2])) with the gather reading x and x2 in place: the same products as on the materialised
    concatenation (which may run a different kernel / split, so only the fp32 summation order can differ); and equal
    to the fp64 convolution."""
    x, x2 = rnd(2, C1, H, H, seed=1), rnd(2, C2, H, H, seed=2)
    w, b = rnd(96, C1 + C2, k, k, seed=3, scale=1 / math.sqrt((C1 + C2) * k * k)), rnd(96, seed=4)
    pad = (1, 1, 1, 1) if k == 3 else (0, 0, 0, 0)
    with ops.operand_precision(prec):
        got = ops.conv2d_fwd_raw(nhwc(x), cl_weight(w), b.to(dev), 1, pad, False, x2=nhwc(x2))
        cat = ops.conv2d_fwd_raw(torch.cat([nhwc(x), nhwc(x2)], -1).contiguous(), cl_weight(w), b.to(dev), 1, pad, False)
    assert (got - cat).abs().max().item() < 2e-6 * math.sqrt((C1 + C2) * k * k)
    if prec == "f32":
        close(got.permute(0, 3, 1, 2), conv_ref(torch.cat([x, x2], 1), w, b, 1, pad, False), atol=3e-5)
    with pytest.raises(Exception):
        ops.conv2d_fwd_raw(nhwc(x)[..., :24].contiguous(), cl_weight(w[:, :24 + C2].contiguous()), None, 1, pad, False, x2=nhwc(x2))


@pytest.mark.parametrize("B,C1,C2,H", [(3, 128, 128, 32), (2, 256, 256, 16), (2, 256, 256, 8), (2, 256, 256, 4), (2, 64, 64, 8),
                                       (2, 256, 128, 16), (2, 128, 256, 8),    # slabs of 96 channels straddle the split
                                       (8, 192, 96, 16), (8, 96, 96, 32), (8, 192, 192, 8)])   # pruned widths: 9 / 6 / 12 per group
def test_groupnorm_two_source_equals_concat(ops, B, C1, C2, H):
    x, x2 = nhwc(rnd(B, C1, H, H, seed=1) + 0.5), nhwc(rnd(B, C2, H, H, seed=2) * 2)
    ga, be = (rnd(C1 + C2, seed=3) * 0.3 + 1).to(dev), rnd(C1 + C2, seed=4).to(dev)
    assert ops.group_norm_two_source_ok(x, x2, 32)
    got = ops.group_norm_cat_raw(x, x2, ga, be, 32, 1e-6, True)
    with torch.no_grad():
        want = ops.group_norm(torch.cat([x, x2], -1).contiguous(), ga, be, 32, 1e-6, True)
    assert torch.equal(got, want)


@pytest.mark.parametrize("B,C1,C2,H,G", [(1, 256, 128, 32, 32), (2, 192, 96, 32, 32), (2, 96, 96, 32, 32), (3, 64, 32, 16, 8)])
def test_groupnorm_two_source_on_the_two_pass_plan(ops, B, C1, C2, H, G):
    """Shapes without a one-pass slab plan (384 channels at 32x32; the pruned widths' 288 = 192 + 96 and 192 = 96 + 96 with 9 /
    6 channels per group) - and the forced two-pass plan - read cat([x, x2]) in place too: bit-identical to the
    materialised concatenation on the same plan."""
    x, x2 = nhwc(rnd(B, C1, H, H, seed=1) + 0.5), nhwc(rnd(B, C2, H, H, seed=2) * 2)
    ga, be = (rnd(C1 + C2, seed=3) * 0.3 + 1).to(dev), rnd(C1 + C2, seed=4).to(dev)
    assert ops.group_norm_two_source_ok(x, x2, G)
    for two_pass in (False, True):
        with torch.no_grad(), ops.kernel_flags(gn_two_pass=two_pass):
            got = ops.group_norm_cat_raw(x, x2, ga, be, G, 1e-6, True)
            want = ops.group_norm(torch.cat([x, x2], -1).contiguous(), ga, be, G, 1e-6, True)
        assert torch.equal(got, want)
    xr = torch.cat([x, x2], -1).permute(0, 3, 1, 2).double().cpu()
    ref = F.silu(F.group_norm(xr, G, ga.double().cpu(), be.double().cpu(), 1e-6)).permute(0, 2, 3, 1)
    close(got, ref, rtol=2e-5, atol=2e-5)


def test_unet_sampling_forward_without_concat_equals_grad_mode_forward(ops):
    """no_grad forward (up blocks read h and the skip in place) vs grad-mode forward (materialised torch.cat)."""
    import gad
    from src.ddpm_config import DDPMConfig
    torch.manual_seed(0)
    net = gad.UNet2DModel(**DDPMConfig.cifar100_config["unet_config"]).to(dev)
    x, t = rnd(3, 3, 32, 32, seed=1).to(dev), torch.tensor([1, 500, 999], device=dev)
    with torch.no_grad():
        a = net(x, t).sample
    b = net(x, t).sample.detach()
    # the concat-free gathers are bit-identical to the materialised torch.cat; the two modes differ in three other places, all
    # fp32 summation order only: training at one 256-wide head keeps the three-launch attention while sampling runs the fused
    # kernel (round 2); sampling lets GroupNorm write the Winograd input transform (its moments are reduced per channel slab of
    # that kernel) and computes all time-embedding projections as one GEMM (round 4)
    assert (a - b).abs().max().item() < 2e-5 * max(1.0, b.abs().max().item())
    monkey, keep = ops.attention_core, net._temb_rows
    ops.attention_core = ops.attention_core_fused            # same attention route, norm and projections in both modes -> bit-identical again
    net._temb_rows = lambda temb: ()
    try:
        b2 = net(x, t).sample.detach()
        with torch.no_grad(), ops.kernel_flags(no_gn_wino=True):
            a2 = net(x, t).sample
    finally:
        ops.attention_core, net._temb_rows = monkey, keep
    assert torch.equal(a2, b2)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_conv_patch_kernel_small_maps(ops, prec):
    """8x8 maps: a 128-pixel tile is two whole images, each with its own halo'd sub-patch in LDS."""
    B, Cin, Cout, H = 6, 64, 72, 8
    x, w, b = rnd(B, Cin, H, H, seed=1), rnd(Cout, Cin, 3, 3, seed=2, scale=1 / math.sqrt(Cin * 9)), rnd(Cout, seed=3)
    res = rnd(B, Cout, H, H, seed=5)
    with ops.operand_precision(prec):
        got = ops.conv2d_fwd_raw(nhwc(x), cl_weight(w), b.to(dev), 1, (1, 1, 1, 1), False, residual=nhwc(res), tile_hint=1, splitk_hint=1)
        gen = ops.conv2d_fwd_raw(nhwc(x), cl_weight(w), b.to(dev), 1, (1, 1, 1, 1), False, residual=nhwc(res), tile_hint=2, splitk_hint=1)
    if prec == "f32":
        close(got.permute(0, 3, 1, 2), conv_ref(x, w, b, 1, (1, 1, 1, 1), False) + res.double(), atol=3e-5)
    else:
        want = F.conv2d(_bf16_round(x), _bf16_round(w), b.double(), padding=1) + res.double()
        close(got.permute(0, 3, 1, 2), want, rtol=1e-5, atol=2e-6 * math.sqrt(Cin * 9))
    assert (got - gen).abs().max().item() < 3e-5


def test_attention_fused_qkv_inference_path_is_bit_identical(ops, monkeypatch):
    """no_grad forward of the attention block (one [3C, C] projection, q/k/v read in place) vs the grad-mode forward
    (three projections); the cached fused weight follows in-place parameter updates.  Both modes are put on the fused
    attention kernel (training at one 256-wide head would otherwise take the three-launch route)."""
    import gad
    from gad.nn import Attention
    monkeypatch.setattr(ops, "attention_core", ops.attention_core_fused)
    torch.manual_seed(0)
    for C, heads, d, H in ((256, 1, 256, 16), (224, 7, 32, 8)):
        att = Attention(C, heads, d, 1e-6, 32).to(dev)
        x = nhwc(rnd(3, C, H, H, seed=C))
        with torch.no_grad():
            a = att(x)
        b = att(x).detach()
        assert torch.equal(a, b)
        with torch.no_grad():
            att.to_k.weight.mul_(1.5)                      # in-place update (optimizer / EMA copy_to): cache must refresh
            a2 = att(x)
        assert torch.equal(a2, att(x).detach()) and not torch.equal(a2, a)
    # a raw optimizer kernel writes the parameters through device pointers (no torch version bump): the cache must still refresh
    att = Attention(256, 1, 256, 1e-6, 32).to(dev)
    flat, gflat = gad.flatten_params(list(att.parameters()))
    x = nhwc(rnd(2, 256, 8, 8, seed=9))
    with torch.no_grad():
        before = att(x)
    gflat.fill_(1.0)
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    ops.clip_adam_ema_raw(flat, gflat, m, v, None, None, max_norm=0.0, lr=1e-2, betas=(0.9, 0.999), eps=1e-8,
                          weight_decay=0.0, adamw=False, step=1, ema_decay=0.0)
    with torch.no_grad():
        after = att(x)
    assert not torch.equal(before, after) and torch.equal(after, att(x).detach())


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_float4_epilogue_is_bit_identical_to_scalar_stores(ops, prec):
    """The LDS-transposed float4 epilogue (csrc/gemm_f32.hip::store_block) against dword stores straight from the
    accumulators (kernel_flags(scalar_epilogue=True)): same arithmetic per element, so the outputs must be equal bit
    for bit - with bias, alpha, time-embedding row add and residual fused, ragged M (last tile partly outside), N that
    is a multiple of 4 but not of the tile, split-K partials, batched outputs, 64- and 128-wide tiles, the 3x3 patch
    kernels (forward with 96 / 128 / 160-channel tiles, data gradient).  N % 4 != 0 must take the scalar path either
    way and still be right."""
    from gad._capi import A_KC, B_KC
    import gad
    with gad.operand_precision(prec):
        def both(fn):
            outs = []
            for flags in ({}, {"scalar_epilogue": True}):
                with ops.kernel_flags(**flags):
                    outs.append(fn().clone())
            assert torch.equal(outs[0], outs[1])
            return outs[0]

        for (M, N, K, tile, sk) in ((1000, 100, 96, 0, 0), (777, 260, 160, 1, 0), (333, 68, 2048, 2, 4), (4096, 320, 320, 0, 0)):
            x, w, b = rnd(M, K, seed=1).to(dev), rnd(N, K, seed=2, scale=0.1).to(dev), rnd(N, seed=3).to(dev)
            r = rnd(M, N, seed=4).to(dev)
            y = torch.empty(M, N, device=dev)
            got = both(lambda: (ops.gemm_raw(x, w, y, A_KC, B_KC, M, N, K, K, K, N, alpha=0.5, bias=b, residual=r, ldr=N,
                                             tile_hint=tile, splitk_hint=sk), y)[1])
            if prec == "f32":
                close(got, 0.5 * (x.double() @ w.double().t()) + b.double() + r.double(), rtol=3e-4, atol=3e-4)
        # N % 4 != 0: scalar path
        M, N, K = 500, 77, 64
        x, w = rnd(M, K, seed=5).to(dev), rnd(N, K, seed=6, scale=0.1).to(dev)
        y = torch.empty(M, N, device=dev)
        got = both(lambda: (ops.gemm_raw(x, w, y, A_KC, B_KC, M, N, K, K, K, N), y)[1])
        if prec == "f32":
            close(got, x.double() @ w.double().t(), rtol=3e-4, atol=3e-4)
        # batched output with strides
        Bt, M, N, K = 6, 200, 72, 64
        x, w = rnd(Bt, M, K, seed=7).to(dev), rnd(Bt, N, K, seed=8, scale=0.1).to(dev)
        y = torch.empty(Bt, M, N, device=dev)
        both(lambda: (ops.gemm_raw(x, w, y, A_KC, B_KC, M, N, K, K, K, N, batch=Bt, sA=(M * K, 0), sB=(N * K, 0),
                                   sC=(M * N, 0)), y)[1])
        # 3x3 convolutions on the patch kernels, forward (+ temb + residual) and data gradient
        for (Bn, Cin, Cout, H) in ((6, 64, 96, 32), (5, 96, 128, 16), (4, 64, 160, 32), (16, 64, 128, 8), (32, 64, 64, 4)):
            x = rnd(Bn, H, H, Cin, seed=9).to(dev)
            w = (rnd(Cout, Cin, 3, 3, seed=10, scale=0.05)).to(dev).contiguous(memory_format=torch.channels_last)
            b, temb = rnd(Cout, seed=11).to(dev), rnd(Bn, Cout, seed=12).to(dev)
            res, dy = rnd(Bn, H, H, Cout, seed=13).to(dev), rnd(Bn, H, H, Cout, seed=14).to(dev)
            both(lambda: ops.conv2d_fwd_raw(x, w, b, rowadd=temb, residual=res))
            both(lambda: ops.conv2d_dgrad_raw(dy, w, (Bn, H, H, Cin)))


@pytest.mark.parametrize("B,Cin,Cout,H", [(8, 128, 3, 32), (4, 224, 3, 64), (8, 320, 4, 32), (16, 64, 2, 16), (8, 96, 1, 32)])
def test_conv3x3_few_output_channels_vector_alu_kernel(ops, B, Cin, Cout, H):
    """conv_out of the U-Nets (128 -> 3, 224 -> 3, 320 -> 4; unconditional_generation/main.py:332 / UNet2DConditionModel
    conv_out): `conv3x3_fewout_kernel` (vector ALUs, weights through the scalar cache) against an fp64 convolution and
    against the MFMA engine on the same inputs; gad_gemm_kernel_id must say which one ran."""
    x = rnd(B, H, H, Cin, seed=1).to(dev)
    w = rnd(Cout, Cin, 3, 3, seed=2, scale=0.05).to(dev).contiguous(memory_format=torch.channels_last)
    b = rnd(Cout, seed=3).to(dev)
    ops.PROFILER = prof = ops.GemmProfiler()
    try:
        y = ops.conv2d_fwd_raw(x, w, b)
        with ops.kernel_flags(no_patch=True):
            y_mfma = ops.conv2d_fwd_raw(x, w, b)
        torch.cuda.synchronize()
    finally:
        ops.PROFILER = None
    names = [k[0] for k in prof.summary()]
    assert any("fewout_valu" in n for n in names) and any(n == "conv_fwd" for n in names), names
    ref = F.conv2d(x.permute(0, 3, 1, 2).double().cpu(), w.double().cpu(), b.double().cpu(), padding=1).permute(0, 2, 3, 1)
    close(y, ref, rtol=2e-5, atol=2e-5)
    close(y, y_mfma, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
@pytest.mark.parametrize("B,Cin,Cout,H", [(4, 64, 96, 32), (3, 160, 64, 16), (16, 128, 128, 8)])
def test_frozen_weight_data_gradient_runs_as_forward_conv(ops, prec, B, Cin, Cout, H):
    """SD LoRA step: the base U-Net's convolutions are frozen (train_text_to_image_lora.py:776-784), so their data
    gradient is a forward convolution of dy with the rotated / transposed weight (ops.rotated_weight).  Against the
    data-gradient kernels on the same inputs and against fp64 autograd; a later in-place change of the weight must be seen."""
    import gad
    x = rnd(B, H, H, Cin, seed=1).to(dev).requires_grad_(True)
    w = torch.nn.Parameter(rnd(Cout, Cin, 3, 3, seed=2, scale=0.05).to(dev).contiguous(memory_format=torch.channels_last), requires_grad=False)
    dy = rnd(B, H, H, Cout, seed=3).to(dev)
    with gad.operand_precision(prec):
        assert ops.dgrad_as_forward(w, 1, (1, 1, 1, 1))
        y = ops.Conv2dFn.apply(x, w, None, None, None, 1, (1, 1, 1, 1), False)
        y.backward(dy)
        dx_direct = ops.conv2d_dgrad_raw(dy, w, x.shape)
    xr = x.detach().permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    F.conv2d(xr, w.detach().double().cpu(), padding=1).backward(dy.permute(0, 3, 1, 2).double().cpu())
    want = xr.grad.permute(0, 2, 3, 1)
    tol = 3e-5 if prec == "f32" else 2e-2
    close(x.grad, want, rtol=tol, atol=tol)
    close(x.grad, dx_direct, rtol=tol, atol=tol)
    with torch.no_grad():
        w.mul_(2.0)                                   # version bump: the cached rotated copy must be rebuilt
    x.grad = None
    with gad.operand_precision(prec):
        ops.Conv2dFn.apply(x, w, None, None, None, 1, (1, 1, 1, 1), False).backward(dy)
    close(x.grad, 2.0 * want, rtol=tol, atol=tol)


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_trainable_weights_in_a_flat_buffer_take_the_rotated_shadow(ops, prec):
    """Parameters living in a flat buffer (training.flatten_params): ONE `gad_rotate_conv3x3` launch rotates every 3x3
    weight, the data gradients run as forward convolutions on the shadow - also behind the fused nearest-2x upsample -
    and must equal the data-gradient kernels (fp64 autograd as the referee).  The shadow must follow an optimizer-style
    rewrite of the buffer (epoch bump) and a write through torch (version bump of one parameter)."""
    import gad
    from gad.training import flatten_params
    shapes = [(96, 64, False, 16), (192, 96, True, 8), (64, 288, False, 8), (32, 3, False, 8)]   # Cout, Cin, upsample, H; the last is not eligible
    ws = [torch.nn.Parameter(rnd(co, ci, 3, 3, seed=10 + i, scale=0.05).to(dev).contiguous(memory_format=torch.channels_last))
          for i, (co, ci, _, _) in enumerate(shapes)]
    extra = torch.nn.Parameter(rnd(77, seed=20).to(dev))                       # a non-conv resident between them
    flat, _ = flatten_params([ws[0], extra] + ws[1:])
    assert [ops.dgrad_as_forward(w, 1, (1, 1, 1, 1)) for w in ws] == [True, True, True, False]

    def run(w, ci, up, H, native):
        x = rnd(2, H, H, ci, seed=3).to(dev).requires_grad_(True)
        with gad.operand_precision(prec), ops.kernel_flags(native_dgrad=native):
            y = ops.Conv2dFn.apply(x, w, None, None, None, 1, (1, 1, 1, 1), up)
            dy = rnd(*y.shape, seed=4).to(dev)
            y.backward(dy)
        return x, dy, x.grad

    def check_all():
        for w, (co, ci, up, H) in zip(ws[:3], shapes[:3]):
            w.grad = None
            x, dy, dx = run(w, ci, up, H, native=False)
            _, _, dx_native = run(w, ci, up, H, native=True)
            xr = x.detach().permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
            xe = F.interpolate(xr, scale_factor=2.0, mode="nearest") if up else xr
            F.conv2d(xe, w.detach().double().cpu(), padding=1).backward(dy.permute(0, 3, 1, 2).double().cpu())
            tol = 3e-5 if prec == "f32" else 2e-2
            close(dx, xr.grad.permute(0, 2, 3, 1), rtol=tol, atol=tol)
            close(dx, dx_native, rtol=tol, atol=tol)

    check_all()
    shadow = flat._gad_rot[1]
    got = shadow[ws[1]._gad_flat[1]:][:ws[1].numel()].view(96, 3, 3, 192)      # [ci][2-r][2-s][co]
    assert torch.equal(got, ops.weight_krsc(ws[1]).detach().flip(1, 2).permute(3, 1, 2, 0))
    with torch.no_grad():
        flat.detach().mul_(1.5)                                                 # what the raw optimizer kernel does ...
    flat._gad_epoch = getattr(flat, "_gad_epoch", 0) + 1                        # ... and how it announces it
    check_all()
    with torch.no_grad():
        ws[2].copy_(ws[2] * 0.25)                                               # through torch: only ws[2]._version moves
    check_all()
    assert flat._gad_rot[1] is shadow                                           # refreshed in place, one buffer


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_lean_dense_loaders_are_bit_identical_to_the_masking_ones(ops, prec):
    """Dense GEMM operands with K % 32 == 0 take the lean loaders (rows beyond the tensor clamped instead of masked, one
    add per DMA slot and K step; csrc/gemm_f32.hip LeanKC / LeanMC / LeanKC2 / LeanMC2); kernel_flags(general_loaders=True)
    forces the masking loaders.  Same products in the same order: equal bit for bit - ragged M and N (last tiles partly
    outside), all three layout pairs, split-K, batches, every tile shape, and the K-concatenated (LoRA) forms."""
    from gad._capi import A_KC, A_MC, B_KC, B_MC
    import gad

    def both(fn):
        outs = []
        for flags in ({}, {"general_loaders": True}):
            with ops.kernel_flags(**flags), gad.operand_precision(prec):
                outs.append(fn().clone())
        assert torch.equal(outs[0], outs[1])
        return outs[0]

    def close(got, want, **kw):          # values against fp64 only in fp32 mode (bf16 mode: the equality above is the test)
        if prec == "f32":
            globals()["close"](got, want, **kw)

    for (M, N, K, tile, sk) in ((1000, 100, 96, 0, 0), (777, 260, 160, 1, 0), (333, 68, 2048, 2, 4), (20000, 320, 320, 0, 0),
                                (20000, 320, 320, 3, 0), (130, 132, 32, 0, 0)):
        x, w = rnd(M, K, seed=1).to(dev), rnd(N, K, seed=2, scale=0.1).to(dev)
        dy = rnd(M, N, seed=3).to(dev)
        y, dx, dw = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
        got = both(lambda: (ops.gemm_raw(x, w, y, A_KC, B_KC, M, N, K, K, K, N, tile_hint=tile, splitk_hint=sk), y)[1])
        close(got, x.double() @ w.double().t(), rtol=3e-4, atol=3e-4)
        got = both(lambda: (ops.gemm_raw(dy, w, dx, A_KC, B_MC, M, K, N, N, K, K, tile_hint=tile if N % 32 == 0 else 0), dx)[1])
        close(got, dy.double() @ w.double(), rtol=3e-4, atol=3e-4)
        if M % 32 == 0:
            got = both(lambda: (ops.gemm_raw(dy, x, dw, A_MC, B_MC, N, K, M, N, K, K), dw)[1])
            close(got, dy.double().t() @ x.double(), rtol=3e-4, atol=3e-4)
    # batched, with strides
    Bt, M, N, K = 6, 200, 72, 64
    x, w = rnd(Bt, M, K, seed=7).to(dev), rnd(Bt, N, K, seed=8, scale=0.1).to(dev)
    y = torch.empty(Bt, M, N, device=dev)
    got = both(lambda: (ops.gemm_raw(x, w, y, A_KC, B_KC, M, N, K, K, K, N, batch=Bt, sA=(M * K, 0), sB=(N * K, 0),
                                     sC=(M * N, 0)), y)[1])
    close(got, torch.einsum("bmk,bnk->bmn", x.double(), w.double()), rtol=3e-4, atol=3e-4)
    # K-concatenated forms (fused LoRA): forward [x | mid] . [W | U]^T and data gradient [dy | dmid] . [W ; D]
    M, N, K, r = 4100, 320, 320, 64
    x, mid = rnd(M, K, seed=1).to(dev), rnd(M, r, seed=2).to(dev)
    w, up, down = rnd(N, K, seed=3, scale=0.1).to(dev), rnd(N, r, seed=4, scale=0.1).to(dev), rnd(r, K, seed=5, scale=0.1).to(dev)
    dy, dmid = rnd(M, N, seed=6).to(dev), rnd(M, r, seed=7).to(dev)
    y, dx = torch.empty(M, N, device=dev), torch.empty(M, K, device=dev)
    got = both(lambda: (ops.gemm_raw(x, w, y, A_KC, B_KC, M, N, K + r, K, K, N, A_k2=mid, B_k2=up, k_split=K), y)[1])
    close(got, x.double() @ w.double().t() + mid.double() @ up.double().t(), rtol=3e-4, atol=3e-4)
    got = both(lambda: (ops.gemm_raw(dy, w, dx, A_KC, B_MC, M, K, N + r, N, K, K, A_k2=dmid, B_k2=down, k_split=N), dx)[1])
    close(got, dy.double() @ w.double() + dmid.double() @ down.double(), rtol=3e-4, atol=3e-4)
    # 1x1 / stride 1 / pad 0 convolutions run as dense GEMMs over the pixel rows (two sources: the K-concatenated form);
    # with the flag they stay on the im2col gather - same products, same order
    for (Bn, C1, C2, Cout, H) in ((20, 128, 0, 128, 32), (70, 256, 128, 256, 16), (9, 64, 32, 96, 8)):
        x = rnd(Bn, H, H, C1, seed=1).to(dev)
        x2 = rnd(Bn, H, H, C2, seed=2).to(dev) if C2 else None
        w = rnd(Cout, C1 + C2, 1, 1, seed=3, scale=0.1).to(dev).contiguous(memory_format=torch.channels_last)
        b = rnd(Cout, seed=4).to(dev)
        got = both(lambda: ops.conv2d_fwd_raw(x, w, b, pad=(0, 0, 0, 0), x2=x2))
        xin = x if x2 is None else torch.cat([x, x2], -1)
        want = xin.double().reshape(-1, C1 + C2) @ w.double().reshape(Cout, -1).t() + b.double()
        close(got.reshape(-1, Cout), want, rtol=3e-4, atol=3e-4)


GN_WINO_CASES = [  # B, H, W, C (C1 of it from the first source, 0 = single source), groups
    (4, 32, 32, 128, 0, 32), (2, 16, 16, 256, 0, 32), (3, 8, 8, 512, 256, 32), (2, 16, 16, 384, 256, 32), (2, 4, 4, 256, 0, 32),
    (2, 32, 32, 256, 128, 32), (1, 16, 12, 96, 0, 8), (8, 32, 32, 96, 0, 32),
]


@pytest.mark.parametrize("B,H,W,C,C1,G", GN_WINO_CASES)
def test_groupnorm_writes_the_winograd_input_transform(ops, B, H, W, C, C1, G):
    """gad_groupnorm_silu_wino4 (norm -> silu -> F(4x4) input transform in one kernel, the normalised activation only in LDS)
    against the two launches it replaces - gad_groupnorm_silu_fwd, then the route's own input stage
    (GAD_GEMM_WINO_ONLY_INPUT) - on the same inputs: V equal to fp32 rounding (bit-identical where both kernels cut the image
    into the same channel slabs: the moments' reduction order follows the slab), mean / rstd likewise; two-source inputs
    (UpBlock2D's cat read in place); a shape without a plan (3 channels per group) says so."""
    from gad import _capi
    lib = _capi.load()
    xs = rnd(B, H, W, C, seed=1).to(dev)
    x, x2 = (xs, None) if not C1 else (xs[..., :C1].contiguous(), xs[..., C1:].contiguous())
    gamma, beta = (1 + 0.1 * rnd(C, seed=2)).to(dev), (0.1 * rnd(C, seed=3)).to(dev)
    a = _capi.GroupNormArgs()
    mean, rstd = torch.empty(B, G, device=dev), torch.empty(B, G, device=dev)
    a.x, a.gamma, a.beta, a.mean, a.rstd = x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    a.B, a.HW, a.C, a.G, a.eps, a.silu = B, H * W, C, G, 1e-5, 1
    if x2 is not None:
        a.x2, a.C1 = x2.data_ptr(), C1
    ok = lib.gad_groupnorm_wino4_ok(_capi.C.byref(a), W)
    assert ok == (1 if (C // G) % 4 == 0 else 0)
    T = B * (H // 4) * (W // 4)
    V = torch.full((36 * T * C + 64,), 7.0, device=dev)
    if not ok:
        with pytest.raises(_capi.GadError, match="no slab plan|outside the plan"):
            _capi.check(lib.gad_groupnorm_silu_wino4(_capi.C.byref(a), V.data_ptr(), W, ops._stream()), "gad_groupnorm_silu_wino4")
        return
    _capi.check(lib.gad_groupnorm_silu_wino4(_capi.C.byref(a), V.data_ptr(), W, ops._stream()), "gad_groupnorm_silu_wino4")
    assert bool((V[36 * T * C:] == 7.0).all())                                     # nothing past the 36 panels
    # the two launches it replaces
    y = ops.group_norm_cat_raw(x, x2, gamma, beta, G, 1e-5, True) if x2 is not None else ops.group_norm(xs, gamma, beta, G, 1e-5, True)
    w = cl_weight(rnd(64, C, 3, 3, seed=4, scale=0.05))
    grabbed = []

    def grab(kind, n, device):
        t = torch.empty(n, dtype=torch.uint8, device=device)
        if kind == "wino":
            grabbed.append(t)
        return t
    ops.SCRATCH_ALLOC = grab
    try:
        with ops.kernel_flags():
            ops.KERNEL_FLAGS["gemm"] |= _capi.GEMM_WINO_ONLY_INPUT
            ops.conv2d_fwd_raw(y, w, None, tile_hint=10)
    finally:
        ops.SCRATCH_ALLOC = None
    Vref = grabbed[0].view(torch.float32)[:36 * T * C]
    # B^T . B multiplies by up to 25 per direction-pair entry; y itself agrees to ~1e-6
    assert (V[:36 * T * C] - Vref).abs().max().item() <= 2e-6 * Vref.abs().max().item() + 2e-5
    m2, r2 = torch.empty(B, G, device=dev), torch.empty(B, G, device=dev)
    a.y, a.mean, a.rstd = torch.empty_like(y).data_ptr(), m2.data_ptr(), r2.data_ptr()
    ws = ops.workspace(dev)
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    _capi.check(lib.gad_groupnorm_silu_fwd(_capi.C.byref(a), ops._stream()), "gad_groupnorm_silu_fwd")
    close(mean, m2, atol=1e-6, rtol=1e-6)
    close(rstd, r2, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("B,H,C,C1,Cout,epi", [(64, 32, 128, 0, 128, "rowadd"), (64, 32, 128, 0, 128, "residual"), (32, 32, 384, 256, 128, "rowadd"),
                                                (64, 16, 256, 0, 256, "residual"), (256, 8, 512, 256, 256, "rowadd"), (2, 8, 256, 0, 256, "rowadd")])
def test_fused_norm_silu_conv_matches_the_separate_launches(ops, B, H, C, C1, Cout, epi):
    """ops.gn_silu_conv3x3_raw (the sampler's ResnetBlock2D halves) = group norm, then convolution, to fp32 rounding, whatever
    route the planner gives the convolution (the last case is too small for Winograd: the plain launches run, bit-identical)."""
    xs = rnd(B, H, H, C, seed=1).to(dev)
    x, x2 = (xs, None) if not C1 else (xs[..., :C1].contiguous(), xs[..., C1:].contiguous())
    gamma, beta = (1 + 0.1 * rnd(C, seed=2)).to(dev), (0.1 * rnd(C, seed=3)).to(dev)
    w, b = cl_weight(rnd(Cout, C, 3, 3, seed=4, scale=1 / math.sqrt(9 * C))), rnd(Cout, seed=5).to(dev)
    kw = dict(rowadd=rnd(B, Cout, seed=6).to(dev)) if epi == "rowadd" else dict(residual=rnd(B, H, H, Cout, seed=7).to(dev))
    with torch.no_grad():
        got = ops.gn_silu_conv3x3_raw(x, x2, gamma, beta, 32, 1e-5, w, b, **kw)
        with ops.kernel_flags(no_gn_wino=True):
            want = ops.gn_silu_conv3x3_raw(x, x2, gamma, beta, 32, 1e-5, w, b, **kw)
    close(got, want, atol=3e-5, rtol=3e-5)
    if B == 2:
        assert torch.equal(got, want)
    ref = conv_ref(torch.nn.functional.silu(torch.nn.functional.group_norm(xs.cpu().double().permute(0, 3, 1, 2), 32, gamma.cpu().double(), beta.cpu().double(), 1e-5)),
                   w.cpu().double(), b.cpu(), 1, (1, 1, 1, 1), False)
    ref = ref + (kw["rowadd"].cpu().double()[:, :, None, None] if epi == "rowadd" else kw["residual"].cpu().double().permute(0, 3, 1, 2))
    close(got.permute(0, 3, 1, 2), ref, atol=5e-5, rtol=5e-5)

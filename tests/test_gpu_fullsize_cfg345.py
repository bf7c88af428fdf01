"""Full-size property tests for BASELINE configs 3-5: the CelebA-HQ LDM U-Net of
src/ddpm_config.py:423-461 (224/448/672/896 on 64x64x3 latents, heads of 32, T up to 1024) at its training batch 32,
and the SD-1.x UNet2DConditionModel (320/640/1280, 8 heads of 40/80/160, Tk = 77, LoRA r = 256 and ragged ranks;
src/ddpm_config.py:624-672, train_text_to_image_lora.py:776-853) at 32x32 and 64x64 latents.  The CPU oracle would
need minutes to hours at these sizes, so parity is asserted through size-independent properties:
  * sampled outputs of the layers' kernels against fp64 dot products / fp64 softmax rows at the real shapes;
  * invariance to the kernel family (LDS-patch vs im2col-gather convolutions; fused vs three-launch attention);
  * the analytic gradient against a central finite difference of the loss along a random direction
    (<grad, v> = dL/de at e = 0): one number that involves every backward kernel of the model at full width;
  * bit-reproducibility of a whole training step.
Tolerances (stated where used): fp32 contractions 2e-5*sqrt(K) on O(1) data; whole-model family invariance 2e-4
relative; directional derivative 2 % (fp32 finite differences)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


def _g(seed):
    return torch.Generator(device=dev).manual_seed(seed)


# ---------------------------------------------------------------------------------------------- kernels at size
@pytest.mark.parametrize("Cin,Cout,B,H", [(224, 224, 32, 64), (448, 224, 32, 64), (672, 448, 32, 32), (1344, 672, 32, 16),
                                         (1792, 896, 32, 8), (320, 320, 16, 64), (960, 640, 8, 32), (2560, 1280, 16, 8),
                                         (1920, 1280, 4, 16), (640, 320, 4, 64)])
def test_conv_shapes_of_celeba_and_sd_vs_fp64(Cin, Cout, B, H):
    from gad import ops
    g = _g(Cin + Cout)
    x = torch.randn(B, H, H, Cin, device=dev, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev, generator=g) * (Cin * 9) ** -0.5).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev, generator=g)
    y = ops.conv2d_fwd_raw(x, w, b)
    gi = torch.Generator().manual_seed(0)
    xp = torch.nn.functional.pad(x, (0, 0, 1, 1, 1, 1))
    wk = w.permute(0, 2, 3, 1).double()
    picks = list(zip(torch.randint(0, B, (24,), generator=gi).tolist(), torch.randint(0, H, (24,), generator=gi).tolist(),
                     torch.randint(0, H, (24,), generator=gi).tolist())) + [(0, 0, 0), (B - 1, H - 1, H - 1), (B - 1, 0, H - 1)]
    tol = 2e-5 * (9 * Cin) ** 0.5
    for n, i, j in picks:
        ref = (wk * xp[n, i:i + 3, j:j + 3, :].double()[None]).sum((1, 2, 3)) + b.double()
        assert torch.allclose(y[n, i, j].double(), ref, atol=tol), (n, i, j)
    with ops.kernel_flags(no_patch=True):
        assert (ops.conv2d_fwd_raw(x, w, b) - y).abs().max().item() < tol
    # adjointness of the three kernels at this shape: <dy, conv x> = <dgrad dy, x> = <wgrad, w>
    dy = torch.randn(B, H, H, Cout, device=dev, generator=g)
    y0 = ops.conv2d_fwd_raw(x, w, None)
    a = (dy.double() * y0.double()).sum()
    bb = (ops.conv2d_dgrad_raw(dy, w, x.shape).double() * x.double()).sum()
    c = (ops.conv2d_wgrad_raw(dy, x, w).double() * w.double()).sum()
    # each of the numel(y) terms carries the kernels' rounding (<= ~1e-5 of the output scale on the F(4x4) Winograd routes the
    # planner takes here for all three; ~4e-7 on the direct kernels): the sums agree to that noise over sqrt(numel) terms
    slack = 1e-5 * abs(a) + 1e-3 + 2e-5 * (y0.numel() ** 0.5)
    assert abs(a - bb) < slack and abs(a - c) < slack
    with ops.kernel_flags(no_wino=True):           # the three direct kernels at the old bar
        a0 = (dy.double() * ops.conv2d_fwd_raw(x, w, None).double()).sum()
        b0 = (ops.conv2d_dgrad_raw(dy, w, x.shape).double() * x.double()).sum()
        c0 = (ops.conv2d_wgrad_raw(dy, x, w).double() * w.double()).sum()
    assert abs(a0 - b0) < 1e-5 * abs(a0) + 1e-3 and abs(a0 - c0) < 1e-5 * abs(a0) + 1e-3


@pytest.mark.parametrize("B,T,Tk,heads,d", [(32, 1024, 1024, 14, 32), (32, 256, 256, 21, 32), (32, 64, 64, 28, 32),
                                           (16, 4096, 4096, 8, 40), (16, 4096, 77, 8, 40), (16, 1024, 1024, 8, 80),
                                           (16, 1024, 77, 8, 80), (16, 256, 256, 8, 160), (64, 64, 77, 8, 160)])
def test_attention_shapes_of_celeba_and_sd_vs_fp64_rows(B, T, Tk, heads, d):
    """Full-size launches; fp64 softmax(q k^T / sqrt d) v for sampled (batch, head, query) rows; gradients through the
    adjoint identity <dO, O> linearisation: dq/dk/dv vs the three-launch route where its score tensor fits."""
    from gad import ops
    C = heads * d
    g = _g(T + Tk + d)
    q = torch.randn(B, T, C, device=dev, generator=g) * 0.6
    k = torch.randn(B, Tk, C, device=dev, generator=g) * 0.6
    v = torch.randn(B, Tk, C, device=dev, generator=g)
    do = torch.randn(B, T, C, device=dev, generator=g)
    gq, gk, gv = (t.clone().requires_grad_(True) for t in (q, k, v))
    out = ops.attention_core_fused(gq, gk, gv, heads)
    out.backward(do)
    gi = torch.Generator().manual_seed(1)
    for _ in range(12):
        b_, h_, i_ = (int(torch.randint(0, n, (1,), generator=gi)) for n in (B, heads, T))
        sl = slice(h_ * d, (h_ + 1) * d)
        s = (k[b_, :, sl].double() @ q[b_, i_, sl].double()) / math.sqrt(d)
        p = torch.softmax(s, dim=0)
        want = p @ v[b_, :, sl].double()
        assert (out[b_, i_, sl].double() - want).abs().max().item() < 3e-5
        # dq row: dS = p * (dP - sum(p dP)), dq = dS K / sqrt(d)
        dp = v[b_, :, sl].double() @ do[b_, i_, sl].double()
        ds = p * (dp - (p * dp).sum())
        assert (gq.grad[b_, i_, sl].double() - (ds @ k[b_, :, sl].double()) / math.sqrt(d)).abs().max().item() < 6e-5
    if 4.0 * B * heads * T * Tk * 3 < 8e9:          # the three-launch route needs S, dP: only where they fit comfortably
        uq, uk, uv = (t.clone().requires_grad_(True) for t in (q, k, v))
        ref = ops.attention_core_unfused(uq, uk, uv, heads)
        ref.backward(do)
        assert (ref - out).abs().max().item() < 3e-5
        for a_, b_ in ((uq.grad, gq.grad), (uk.grad, gk.grad), (uv.grad, gv.grad)):
            assert (a_ - b_).abs().max().item() < 1e-4 * max(1.0, a_.abs().max().item())
    else:                                           # dk, dv: column sums against fp64 on sampled keys
        for _ in range(6):
            b_, h_, j_ = (int(torch.randint(0, n, (1,), generator=gi)) for n in (B, heads, Tk))
            sl = slice(h_ * d, (h_ + 1) * d)
            S = (q[b_, :, sl].double() @ k[b_, :, sl].double().T) / math.sqrt(d)          # [T, Tk] of one (b, h)
            P = torch.softmax(S, dim=1)
            dP = do[b_, :, sl].double() @ v[b_, :, sl].double().T
            dS = P * (dP - (P * dP).sum(1, keepdim=True))
            assert (gv.grad[b_, j_, sl].double() - P[:, j_] @ do[b_, :, sl].double()).abs().max().item() < 2e-4
            assert (gk.grad[b_, j_, sl].double() - (dS[:, j_] @ q[b_, :, sl].double()) / math.sqrt(d)).abs().max().item() < 2e-4


@pytest.mark.parametrize("B,C,H,G", [(32, 224, 64, 32), (32, 448, 64, 32), (32, 1344, 16, 32), (32, 1792, 8, 32),
                                     (16, 320, 64, 32), (16, 960, 64, 32), (16, 2560, 8, 32), (16, 1920, 16, 32)])
def test_groupnorm_shapes_of_celeba_and_sd(B, C, H, G):
    """Per-(image, group) mean ~ 0 / variance ~ 1 of the normalised output at full size (affine off), the apply against
    fp64 on sampled images, and one-pass vs two-pass plans."""
    from gad import ops
    g = _g(C + H)
    x = torch.randn(B, H, H, C, device=dev, generator=g) * 1.7 + 0.4
    ga, be = torch.randn(C, device=dev, generator=g) * 0.3 + 1, torch.randn(C, device=dev, generator=g) * 0.2
    one, zero = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    with torch.no_grad():
        y = ops.group_norm(x, one, zero, G, 1e-5, False)
        yg = y.view(B, H * H, G, C // G).permute(0, 2, 1, 3).reshape(B, G, -1)
        assert yg.mean(-1).abs().max().item() < 2e-5 and (yg.var(-1, unbiased=False) - 1).abs().max().item() < 1e-3
        z = ops.group_norm(x, ga, be, G, 1e-5, True)
        for n in (0, B - 1):
            xd = x[n].double().view(H * H, G, C // G)
            m, var = xd.mean((0, 2), keepdim=True), xd.var((0, 2), unbiased=False, keepdim=True)
            ref = ((xd - m) / (var + 1e-5).sqrt()).view(H, H, C) * ga.double() + be.double()
            ref = ref * torch.sigmoid(ref)
            assert (z[n].double() - ref).abs().max().item() < 3e-5
        with ops.kernel_flags(gn_two_pass=True):
            assert (ops.group_norm(x, ga, be, G, 1e-5, True) - z).abs().max().item() < 3e-5


# ---------------------------------------------------------------------------------------------- whole models
def _directional_check(loss_fn, params, grads, seed, eps):
    """<grad, v> against (L(theta + eps v) - L(theta - eps v)) / (2 eps) for a random unit-RMS direction v."""
    g = _g(seed)
    vs = [torch.randn(p.shape, device=dev, generator=g) for p in params]
    analytic = sum((gr.double() * v.double()).sum() for gr, v in zip(grads, vs)).item()
    with torch.no_grad():
        saved = [p.detach().clone() for p in params]
        for p, v in zip(params, vs):
            p.add_(v, alpha=eps)
        lp = loss_fn()
        for p, v, s in zip(params, vs, saved):
            p.copy_(s).add_(v, alpha=-eps)
        lm = loss_fn()
        for p, s in zip(params, saved):
            p.copy_(s)                                   # exact restore (p + e v - 2 e v + e v is not bit-exact in fp32)
    return analytic, (lp - lm) / (2 * eps)


def test_celeba_unet_full_size_properties():
    import gad
    from gad import ops
    from src.ddpm_config import DDPMConfig
    cfg = DDPMConfig.celeba_config["unet_config"]
    torch.manual_seed(0)
    with torch.device(dev):
        net = gad.UNet2DModel(**cfg)
    net.to(dev)
    assert sum(p.numel() for p in net.parameters()) > 250e6
    g = _g(3)
    x = torch.randn(32, 3, 64, 64, device=dev, generator=g)
    noise = torch.randn(32, 3, 64, 64, device=dev, generator=g)
    t = torch.randint(0, 1000, (32,), device=dev, generator=g)
    with torch.no_grad():
        y = net(x, t).sample
        assert torch.isfinite(y).all() and torch.equal(net(x, t).sample, y)              # deterministic
        with ops.kernel_flags(no_patch=True):
            y_gen = net(x, t).sample
        assert (y_gen - y).abs().max().item() < 2e-4 * max(1.0, y.abs().max().item())     # kernel family
        orig = ops.attention_core_qkv_raw

        def unfused_qkv(qkv, Bn, T, Cq, heads, scale=None):
            q, k, v = (qkv[:, i * Cq:(i + 1) * Cq].reshape(Bn, T, Cq).contiguous() for i in range(3))
            return ops.UnfusedAttentionCoreFn.apply(q, k, v, heads, scale)
        ops.attention_core_qkv_raw = unfused_qkv
        try:
            y_unf = net(x, t).sample
        finally:
            ops.attention_core_qkv_raw = orig
        assert (y_unf - y).abs().max().item() < 2e-4 * max(1.0, y.abs().max().item())     # fused vs three-launch attention

    def loss_of():
        with torch.no_grad():
            return float(ops.mse_fwd_bwd_raw(net(x, t).sample.contiguous(), noise)[0].double())
    out = net(x, t).sample
    loss, d = ops.mse_fwd_bwd_raw(out.contiguous(), noise)
    out.backward(d)
    params = [p for p in net.parameters()]
    grads = [p.grad for p in params]
    assert all(gr is not None and torch.isfinite(gr).all() for gr in grads)
    analytic, numeric = _directional_check(loss_of, params, grads, seed=11, eps=1e-4)
    assert abs(analytic - numeric) < 0.02 * abs(numeric) + 1e-6, (analytic, numeric)
    # bit-reproducible backward
    for p in params:
        p.grad = None
    out2 = net(x, t).sample
    out2.backward(ops.mse_fwd_bwd_raw(out2.contiguous(), noise)[1])
    assert all(torch.equal(a, p.grad) for a, p in zip(grads, params))


@pytest.mark.parametrize("latent,B,ragged", [(32, 8, False), (64, 2, False), (32, 4, True)])
def test_sd_unet_full_size_properties(latent, B, ragged):
    """SD-1.x widths (859.5 M parameters), LoRA r = 256 on the 128 projections (or ragged ranks as text_to_image/prune_lora.py
    leaves them): forward determinism, kernel-family and attention-route invariance, LoRA-gradient directional
    derivative, bit-reproducible training step."""
    import gad
    from gad import ops
    torch.manual_seed(0)
    with torch.device(dev):
        net = gad.UNet2DConditionModel(sample_size=latent)
    net.to(dev)
    assert sum(p.numel() for p in net.parameters()) == 859_520_964
    ranks = None
    if ragged:
        rng = torch.Generator().manual_seed(5)
        ranks = {f"{n[:-len('.processor')]}.{p}": int(torch.randint(17, 257, (1,), generator=rng))
                 for n in net.attention_modules() for p in ("to_q", "to_k", "to_v", "to_out")}
    lora = net.inject_lora(rank=256, ranks=ranks)
    if not ragged:
        assert sum(p.numel() for p in lora) == 51_019_776
    with torch.no_grad():
        for n, p in net.named_parameters():
            if n.endswith("lora_layer.up.weight"):
                p.normal_(0.0, 0.02, generator=None)
    g = _g(latent)
    x = torch.randn(B, 4, latent, latent, device=dev, generator=g) * 0.8
    noise = torch.randn(B, 4, latent, latent, device=dev, generator=g)
    ctx = torch.randn(B, 77, 768, device=dev, generator=g) * 0.5
    t = torch.randint(0, 1000, (B,), device=dev, generator=g)
    with torch.no_grad():
        y = net(x, t, ctx).sample
        assert torch.isfinite(y).all() and torch.equal(net(x, t, ctx).sample, y)
        with ops.kernel_flags(no_patch=True):
            y_gen = net(x, t, ctx).sample
        assert (y_gen - y).abs().max().item() < 2e-4 * max(1.0, y.abs().max().item())
        if latent == 32:                                 # three-launch attention: S = B*8*1024*1024*4 B fits
            orig = ops.attention_core
            ops.attention_core = ops.attention_core_unfused
            try:
                y_unf = net(x, t, ctx).sample
            finally:
                ops.attention_core = orig
            assert (y_unf - y).abs().max().item() < 2e-4 * max(1.0, y.abs().max().item())

    def loss_of():
        with torch.no_grad():
            return float(ops.mse_fwd_bwd_raw(net(x, t, ctx).sample.contiguous(), noise)[0].double())
    out = net(x, t, ctx).sample
    out.backward(ops.mse_fwd_bwd_raw(out.contiguous(), noise)[1])
    grads = [p.grad for p in lora]
    assert all(gr is not None and torch.isfinite(gr).all() for gr in grads)
    assert all(p.grad is None for n, p in net.named_parameters() if "lora_layer" not in n)      # base frozen (:746)
    analytic, numeric = _directional_check(loss_of, lora, grads, seed=13, eps=2e-4)
    assert abs(analytic - numeric) < 0.02 * abs(numeric) + 1e-6, (analytic, numeric)
    for p in lora:
        p.grad = None
    out2 = net(x, t, ctx).sample
    out2.backward(ops.mse_fwd_bwd_raw(out2.contiguous(), noise)[1])
    assert all(torch.equal(a, p.grad) for a, p in zip(grads, lora))

"""Fused attention kernels (csrc/attention.hip, gad_attention_fwd / gad_attention_bwd) against an fp64 restatement of
F.scaled_dot_product_attention (reference src/diffusers/models/attention_processor.py:1314-1325) at every head dim the
reference's models use - 256 (CIFAR, 1 head), 32 (CelebA-HQ), 40 / 80 / 160 (SD-1.x self and cross attention,
Tk = 77), 192 (pruned CIFAR) - plus ragged lengths, strided q|k|v views, a forced online-softmax rescale, and the
unfused three-launch route as an independent implementation.  Tolerances: outputs 3e-5, gradients 6e-5 (fp32 products,
fp32 exp2; O(1) data)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    return torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale


def sdpa64(q, k, v, heads):
    B, Tq, C = q.shape
    Tk, d = k.shape[1], C // heads

    def split(t, T):
        return t.view(B, T, heads, d).transpose(1, 2)
    w = torch.softmax(split(q, Tq) @ split(k, Tk).transpose(-1, -2) / math.sqrt(d), dim=-1)
    return (w @ split(v, Tk)).transpose(1, 2).reshape(B, Tq, C)


CASES = [  # B, Tq, Tk, heads, d
    (2, 256, 256, 1, 256), (3, 16, 16, 1, 256), (2, 64, 64, 7, 32), (1, 1024, 1024, 2, 32), (2, 256, 256, 8, 40),
    (2, 64, 77, 4, 40), (1, 100, 77, 2, 80), (2, 64, 64, 2, 160), (1, 70, 77, 8, 160), (2, 256, 256, 1, 192),
    (1, 33, 95, 3, 16), (1, 130, 50, 2, 64), (1, 64, 64, 1, 96), (1, 48, 48, 1, 128), (1, 40, 40, 1, 224),
]


@pytest.mark.parametrize("B,Tq,Tk,heads,d", CASES)
def test_fused_attention_forward_backward_vs_fp64(B, Tq, Tk, heads, d):
    from gad import ops
    C = heads * d
    assert ops.fused_attention_ok(d, C)
    q, k, v = rnd(B, Tq, C, seed=1, scale=0.7), rnd(B, Tk, C, seed=2, scale=0.7), rnd(B, Tk, C, seed=3)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    o = sdpa64(qd, kd, vd, heads)
    do = rnd(B, Tq, C, seed=4)
    o.backward(do.double())
    gq, gk, gv = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    out = ops.attention_core_fused(gq, gk, gv, heads)
    out.backward(do.to(dev))
    for got, want, tol in ((out, o, 3e-5), (gq.grad, qd.grad, 6e-5), (gk.grad, kd.grad, 6e-5), (gv.grad, vd.grad, 6e-5)):
        err = (got.detach().cpu().double() - want.detach()).abs().max().item()
        assert err < tol, err
    # the three-launch route (batched Q K^T, row softmax, P V) is an independent implementation of the same function
    uq, uk, uv = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    ref = ops.attention_core_unfused(uq, uk, uv, heads)
    ref.backward(do.to(dev))
    assert (ref - out).abs().max().item() < 3e-5
    assert (uq.grad - gq.grad).abs().max().item() < 6e-5 and (uk.grad - gk.grad).abs().max().item() < 6e-5


@pytest.mark.parametrize("heads,d,T", [(1, 256, 256), (8, 40, 1024), (7, 32, 64)])
def test_fused_attention_reads_q_k_v_in_place_from_one_projection(heads, d, T):
    """q | k | v as column blocks of a [B*T, 3C] projection output (row stride 3C): bit-identical to separate tensors."""
    from gad import ops
    B, C = 2, heads * d
    qkv = rnd(B * T, 3 * C, seed=7, scale=0.6).to(dev)
    fused = ops.attention_core_qkv_raw(qkv, B, T, C, heads)
    q, k, v = (qkv[:, i * C:(i + 1) * C].reshape(B, T, C).contiguous() for i in range(3))
    with torch.no_grad():
        sep = ops.attention_core(q, k, v, heads)
    assert torch.equal(fused, sep)
    want = sdpa64(q.cpu().double(), k.cpu().double(), v.cpu().double(), heads)
    assert (fused.cpu().double() - want).abs().max().item() < 3e-5


def test_online_softmax_rescale_branch_is_exercised():
    """cdna_hip_programming.md rule 26: a key deep in the sequence that dominates a query's row forces the running
    maximum to jump at a late tile (O and l rescaled by ~e^-40); early dominant keys exercise the opposite case."""
    from gad import ops
    B, T, heads, d = 1, 256, 2, 40
    C = heads * d
    q, k, v = rnd(B, T, C, seed=1, scale=0.3), rnd(B, T, C, seed=2, scale=0.3), rnd(B, T, C, seed=3)
    k[0, 200, :d] = q[0, 17, :d] * 60.0            # query 17 of head 0 meets its maximum in tile 6
    k[0, 3, d:] = q[0, 90, d:] * 60.0              # query 90 of head 1 meets it in tile 0
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    o = sdpa64(qd, kd, vd, heads)
    do = rnd(B, T, C, seed=4)
    o.backward(do.double())
    gq, gk, gv = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    out = ops.attention_core_fused(gq, gk, gv, heads)
    out.backward(do.to(dev))
    assert (out.detach().cpu().double() - o.detach()).abs().max().item() < 5e-5
    for got, want in ((gq.grad, qd.grad), (gk.grad, kd.grad), (gv.grad, vd.grad)):
        scale = max(1.0, want.abs().max().item())
        assert (got.cpu().double() - want).abs().max().item() < 1e-4 * scale


def test_fused_attention_is_bit_reproducible_and_leaves_no_score_tensor():
    """Two runs give identical bits (no atomics anywhere), and the forward allocates nothing of size B*heads*Tq*Tk."""
    from gad import ops
    B, T, heads, d = 4, 1024, 8, 40
    C = heads * d
    q, k, v, do = (rnd(B, T, C, seed=s, scale=0.5).to(dev) for s in (1, 2, 3, 4))
    outs = []
    for _ in range(2):
        gq, gk, gv = (t.clone().requires_grad_(True) for t in (q, k, v))
        torch.cuda.reset_peak_memory_stats(dev)
        base = torch.cuda.memory_allocated(dev)
        out = ops.attention_core(gq, gk, gv, heads)
        peak_fwd = torch.cuda.max_memory_allocated(dev) - base
        out.backward(do)
        outs.append((out.detach().clone(), gq.grad.clone(), gk.grad.clone(), gv.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert peak_fwd < 4 * B * heads * T * T                  # the S tensor alone would be 4*B*heads*T*T bytes = 134 MB
    assert peak_fwd <= 2 * (4 * B * T * C + 4 * B * heads * T) + (1 << 20)


@pytest.mark.parametrize("B,Tq,Tk,heads,d", [(1, 2048, 2048, 2, 40), (2, 300, 700, 3, 23), (2, 64, 77, 4, 40), (1, 1000, 1030, 2, 32),
                                           (1, 520, 260, 2, 80), (1, 200, 130, 1, 96), (2, 256, 256, 14, 32), (1, 96, 4100, 1, 24)])
def test_single_pass_backward_vs_the_two_kernel_pair_and_fp64(B, Tq, Tk, heads, d):
    """Head dims up to 96 run ONE backward kernel per key block (S and dP once; dQ partial slabs + fixed-order reduce when
    a (b, h) has several key blocks: Tk = 2048 at d = 40 is 8 slabs, Tk = 77 writes dQ directly).  Same gradients as the
    recomputing dQ + dK/dV pair (`kernel_flags(two_kernel_attn_bwd=True)`) and as fp64 autograd; bit-reproducible."""
    from gad import ops
    C = heads * d
    q, k, v = rnd(B, Tq, C, seed=1, scale=0.7), rnd(B, Tk, C, seed=2, scale=0.7), rnd(B, Tk, C, seed=3)
    do = rnd(B, Tq, C, seed=4)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    sdpa64(qd, kd, vd, heads).backward(do.double())
    grads = []
    for pair in (False, True, False):
        gq, gk, gv = (t.to(dev).requires_grad_(True) for t in (q, k, v))
        with ops.kernel_flags(two_kernel_attn_bwd=pair):
            ops.attention_core_fused(gq, gk, gv, heads).backward(do.to(dev))
        grads.append((gq.grad, gk.grad, gv.grad))
    for one, two, again, want in zip(grads[0], grads[1], grads[2], (qd.grad, kd.grad, vd.grad)):
        assert torch.equal(one, again)
        scale = want.abs().max().item()
        assert (one.cpu().double() - want).abs().max().item() < 6e-5 * max(1.0, scale)
        assert (one - two).abs().max().item() < 6e-5 * max(1.0, scale)


@pytest.mark.parametrize("B,Tq,Tk,heads,d", [(3, 256, 256, 1, 256), (2, 256, 256, 1, 192), (2, 100, 77, 2, 160), (1, 70, 130, 1, 224)])
def test_wide_head_forward_8_wave_kernel_vs_the_4_wave_kernel(B, Tq, Tk, heads, d):
    """d >= 160 runs the 8-wave forward (each wave half the head dim, partial scores exchanged through LDS); the 4-wave
    kernel (`kernel_flags(narrow_attn_fwd=True)`) computes the same function with one k-sum instead of two halves - equal
    to fp32 rounding, both equal to fp64, LSE included (the backward consumes either's)."""
    from gad import ops
    C = heads * d
    q, k, v = rnd(B, Tq, C, seed=1, scale=0.7).to(dev), rnd(B, Tk, C, seed=2, scale=0.7).to(dev), rnd(B, Tk, C, seed=3).to(dev)
    wide, lse_w = ops.attention_fwd_raw(q, k, v, B, heads, Tq, Tk, d, C, C, C)
    with ops.kernel_flags(narrow_attn_fwd=True):
        narrow, lse_n = ops.attention_fwd_raw(q, k, v, B, heads, Tq, Tk, d, C, C, C)
    want = sdpa64(q.cpu().double(), k.cpu().double(), v.cpu().double(), heads)
    assert (wide.cpu().double() - want).abs().max().item() < 3e-5 and (narrow.cpu().double() - want).abs().max().item() < 3e-5
    assert (wide - narrow).abs().max().item() < 1e-5 and (lse_w - lse_n).abs().max().item() < 1e-4
    again, _ = ops.attention_fwd_raw(q, k, v, B, heads, Tq, Tk, d, C, C, C)
    assert torch.equal(again, wide)


def test_attention_argument_contract():
    from gad import _capi, ops
    q = rnd(1, 8, 300, seed=1).to(dev)
    assert ops.fused_attention_ok(24, 24) and ops.fused_attention_ok(40, 322) and ops.fused_attention_ok(23, 322)
    assert ops.fused_attention_ok(256, 256) and not ops.fused_attention_ok(257, 257) and not ops.fused_attention_ok(0, 8)
    a = ops._attention_args(q, q, q, q, None, 1, 1, 8, 8, 300, 300, 300, 300, 2400, 2400, 2400)
    with pytest.raises(_capi.GadError, match="outside 1..256"):
        _capi.check(_capi.load().gad_attention_fwd(_capi.C.byref(a), ops._stream()), "gad_attention_fwd")
    a = ops._attention_args(q, q, q, q, None, 1, 2, 8, 8, 40, 40, 80, 80, 320, 640, 640)      # ldq < heads * d
    with pytest.raises(_capi.GadError, match="row stride"):
        _capi.check(_capi.load().gad_attention_fwd(_capi.C.byref(a), ops._stream()), "gad_attention_fwd")


# Any head dim, any alignment: the head-grouped-pruned CelebA-HQ model keeps its 14 / 21 / 28 heads and shrinks the head
# dim 32 -> 23 (reference unconditional_generation/prune.py:337-342; rows of 322 / 483 / 644 floats), pruned SD-style
# widths give 20 or 46; 7 x 9 = 63-float rows, a head dim below one MFMA step, a wide odd head and an exact instance
# behind an unaligned row stride (q | k | v blocks of a 3 x 322-wide projection) take the same dword-staged instances.
RAGGED = [  # B, Tq, Tk, heads, d
    (2, 1024, 1024, 14, 23), (2, 256, 256, 21, 23), (3, 64, 64, 28, 23), (2, 100, 77, 8, 20), (1, 130, 95, 5, 46),
    (2, 40, 40, 7, 9), (1, 33, 50, 3, 3), (1, 70, 70, 2, 100), (1, 48, 40, 1, 200), (1, 64, 64, 1, 250),
]


@pytest.mark.parametrize("B,Tq,Tk,heads,d", RAGGED)
def test_fused_attention_any_head_dim_vs_fp64(B, Tq, Tk, heads, d):
    from gad import ops
    C = heads * d
    q, k, v = rnd(B, Tq, C, seed=1, scale=0.7), rnd(B, Tk, C, seed=2, scale=0.7), rnd(B, Tk, C, seed=3)
    qd, kd, vd = (t.double().requires_grad_(True) for t in (q, k, v))
    o = sdpa64(qd, kd, vd, heads)
    do = rnd(B, Tq, C, seed=4)
    o.backward(do.double())
    gq, gk, gv = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    base = torch.cuda.memory_allocated(dev)
    torch.cuda.reset_peak_memory_stats(dev)
    # the product's router must pick the fused kernels (the one exception it makes: TRAINING at one wide head, d > 160,
    # over <= 256 x 256 scores, see ops.attention_core)
    out = ops.attention_core(gq, gk, gv, heads) if d <= 160 else ops.attention_core_fused(gq, gk, gv, heads)
    assert torch.cuda.max_memory_allocated(dev) - base < 4 * B * heads * Tq * Tk or B * heads * Tq * Tk < (1 << 16), \
        "a score tensor was allocated: the launch went through the three-launch route"
    out.backward(do.to(dev))
    for got, want, tol in ((out, o, 3e-5), (gq.grad, qd.grad, 6e-5), (gk.grad, kd.grad, 6e-5), (gv.grad, vd.grad, 6e-5)):
        err = (got.detach().cpu().double() - want.detach()).abs().max().item()
        assert err < tol, err
    # bit-reproducible, and bf16 mode falls back to the same exact-fp32 instances for these launches
    uq, uk, uv = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    with ops.operand_precision("bf16"):
        again = ops.attention_core_fused(uq, uk, uv, heads)
        again.backward(do.to(dev))
    assert torch.equal(again, out) and torch.equal(uq.grad, gq.grad) and torch.equal(uk.grad, gk.grad) and torch.equal(uv.grad, gv.grad)


def test_fused_attention_exact_instance_behind_unaligned_rows():
    """q | k | v as column blocks of one [B T, 3 x 322]-wide projection (the sampling path of a pruned model): head dim 23
    at row stride 966, block offsets 322 and 644 floats - and a d = 32 instance whose rows are 4-float aligned but whose
    base pointer is not."""
    from gad import ops
    B, T, heads, d = 2, 64, 14, 23
    C = heads * d
    qkv = rnd(B * T, 3 * C, seed=5, scale=0.7).to(dev)
    got = ops.attention_core_qkv_raw(qkv, B, T, C, heads)
    q, k, v = (qkv[:, i * C:(i + 1) * C].reshape(B, T, C).cpu().double() for i in range(3))
    assert (got.cpu().double() - sdpa64(q, k, v, heads)).abs().max().item() < 3e-5
    heads, d = 3, 32
    C = heads * d
    buf = rnd(3 * B * T * C + 1, seed=6, scale=0.7).to(dev)
    q, k, v = (buf[1 + i * B * T * C: 1 + (i + 1) * B * T * C].view(B, T, C) for i in range(3))      # 4-byte aligned only
    out, _ = ops.attention_fwd_raw(q, k, v, B, heads, T, T, d, C, C, C)
    assert (out.cpu().double() - sdpa64(q.cpu().double(), k.cpu().double(), v.cpu().double(), heads)).abs().max().item() < 3e-5


@pytest.mark.parametrize("B,Tq,Tk,heads,d", [(2, 256, 256, 1, 256), (2, 64, 64, 7, 32), (2, 256, 256, 8, 40), (2, 64, 77, 4, 40),
                                           (1, 100, 77, 2, 80), (2, 64, 64, 2, 160), (1, 33, 95, 3, 16), (1, 130, 50, 2, 64),
                                           (1, 64, 64, 1, 96), (1, 48, 48, 1, 128), (1, 256, 256, 1, 192), (1, 40, 40, 1, 224)])
def test_fused_attention_bf16_operands(B, Tq, Tk, heads, d):
    """operand_precision = 1: q (pre-scaled), k, v and the probabilities are rounded to bf16 (RNE), products exact,
    fp32 accumulation and statistics.  Against fp64 attention of the bf16-rounded q, k, v the remaining differences are
    the bf16 rounding of q * scale * log2(e) and of P (2^-9 relative each): 1.5e-2 on O(1) values; and the result stays
    within bf16 distance of the fp32 kernel."""
    from gad import ops
    C = heads * d
    q, k, v = rnd(B, Tq, C, seed=1, scale=0.7), rnd(B, Tk, C, seed=2, scale=0.7), rnd(B, Tk, C, seed=3)
    r = lambda t: t.to(torch.bfloat16).double()
    want = sdpa64(r(q), r(k), r(v), heads)
    with torch.no_grad():
        exact = ops.attention_core_fused(q.to(dev), k.to(dev), v.to(dev), heads)
        with ops.operand_precision("bf16"):
            got = ops.attention_core_fused(q.to(dev), k.to(dev), v.to(dev), heads)
    assert (got.cpu().double() - want).abs().max().item() < 1.5e-2
    d_ = (got - exact).abs().max().item()
    assert 0 < d_ < 3e-2
    # backward: gradients of the bf16-operand kernels against fp64 autograd through the bf16-rounded inputs; dS and P are
    # rounded to bf16 before the last contraction: 2 % of each gradient's scale
    qd, kd, vd = (r(t).requires_grad_(True) for t in (q, k, v))
    do = rnd(B, Tq, C, seed=4)
    sdpa64(qd, kd, vd, heads).backward(r(do))
    gq, gk, gv = (t.to(dev).requires_grad_(True) for t in (q, k, v))
    with ops.operand_precision("bf16"):
        out = ops.attention_core_fused(gq, gk, gv, heads)
        out.backward(do.to(dev))
    for got_g, want_g in ((gq.grad, qd.grad), (gk.grad, kd.grad), (gv.grad, vd.grad)):
        assert (got_g.cpu().double() - want_g).abs().max().item() < 2e-2 * max(1.0, want_g.abs().max().item())

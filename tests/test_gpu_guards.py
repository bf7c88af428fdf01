"""Out-of-bounds canaries on caller-owned memory (SURVEY §5 "sanitizers": GPU AddressSanitizer is not available on this
pool, so guard words are the defence).  Every caller-owned scratch region is allocated at EXACTLY the size the C ABI's
query returns - `gad_attention_bwd_workspace_bytes`, `gad_gemm_wino_bytes`, `gad_gemm_workspace_bytes` - and every output
at its exact shape, each between two poisoned guard bands; after the launch the bands must be untouched and the result
must equal the un-guarded launch bit for bit.

Why: round 3's development abort (gpurun_out/r3_t6.log, recorded in DESIGN.md §3) was the single-pass attention backward
storing dQ slabs through a NULL workspace when the size query and the launch router disagreed about a bf16-mode launch of
the d = 23 instance.  A disagreement that leaves a too-small (rather than null) region writes into mapped memory and no
other test would notice."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")
PAD = 8192                      # guard band, bytes (each side)
POISON = 0x5A


class Guards:
    """Allocator for ops.SCRATCH_ALLOC / ops.OUT_ALLOC: hands out 512-byte-aligned views between poisoned bands."""

    def __init__(self):
        self.regions = []       # (kind, whole uint8 buffer, payload bytes)

    def _alloc(self, kind, nbytes, device):
        buf = torch.full((nbytes + 2 * PAD,), POISON, dtype=torch.uint8, device=device)
        self.regions.append((kind, buf, nbytes))
        return buf[PAD:PAD + nbytes]

    def scratch(self, kind, nbytes, device):
        return self._alloc(kind, nbytes, device)

    def out(self, shape, device):
        n = 4 * math.prod(shape)
        return self._alloc("out", n, device).view(torch.float32).view(shape)

    def check(self):
        assert self.regions, "no guarded allocation was made: the hooks did not reach the launch"
        for kind, buf, n in self.regions:
            lo, hi = buf[:PAD], buf[PAD + n:]
            assert bool((lo == POISON).all()) and bool((hi == POISON).all()), \
                f"guard band of a {kind!r} region of {n} bytes was written ({int((lo != POISON).sum())} bytes below, {int((hi != POISON).sum())} above)"

    def kinds(self):
        return [k for k, _, _ in self.regions]


def guarded(ops, fn):
    """run fn() with every scratch / output allocation guarded; -> (result, Guards)"""
    g = Guards()
    ops.SCRATCH_ALLOC, ops.OUT_ALLOC = g.scratch, g.out
    try:
        r = fn()
        torch.cuda.synchronize()
    finally:
        ops.SCRATCH_ALLOC = ops.OUT_ALLOC = None
    g.check()
    return r, g


@pytest.fixture(scope="module")
def ops():
    from gad import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    return (torch.randn(*shape, generator=torch.Generator().manual_seed(seed)) * scale).to(dev)


# the shapes of tests/test_gpu_attention.py::RAGGED (dword-staged instances, d = 23 first: 16 key blocks x 14 heads) + SD's
# d = 40 (three 16-wide tiles: the PAIR dQ schedule) + multi-block aligned heads + cross attention (a single key block)
ATTN = [(2, 1024, 1024, 14, 23), (2, 256, 256, 21, 23), (3, 64, 64, 28, 23), (2, 100, 77, 8, 20), (1, 130, 95, 5, 46),
        (2, 40, 40, 7, 9), (1, 33, 50, 3, 3), (1, 70, 70, 2, 100), (1, 48, 40, 1, 200),
        (2, 1024, 1024, 8, 40), (2, 300, 300, 4, 40), (2, 64, 77, 4, 40), (1, 1024, 1024, 2, 32), (1, 320, 320, 2, 80), (2, 256, 256, 1, 256)]


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("B,Tq,Tk,heads,d", ATTN)
def test_attention_guard_bands(ops, B, Tq, Tk, heads, d, mode):
    """forward + backward with o, lse, dq, dk, dv, delta and the dQ-slab workspace between guard bands, in fp32 mode and in
    bf16-operand mode (where launches that cannot take aligned float4 rows - d = 23 - still run the exact-fp32 instances:
    the case of the round-3 abort); results equal the un-guarded launch bit for bit."""
    C = heads * d
    q, k, v, do = rnd(B, Tq, C, seed=1, scale=0.7), rnd(B, Tk, C, seed=2, scale=0.7), rnd(B, Tk, C, seed=3), rnd(B, Tq, C, seed=4)

    def run():
        gq, gk, gv = (t.clone().requires_grad_(True) for t in (q, k, v))
        with ops.operand_precision(mode):
            out = ops.attention_core_fused(gq, gk, gv, heads)
            out.backward(do)
        return out.detach(), gq.grad, gk.grad, gv.grad
    want = run()
    got, g = guarded(ops, run)
    for a, b in zip(got, want):
        assert torch.equal(a, b)
    assert g.kinds().count("out") >= 6          # o, lse, dq, dk, dv, delta


def test_attention_backward_without_workspace_falls_back(ops):
    """The size query and the router must agree: a launch whose workspace is missing (or too small) runs the dQ + dK/dV
    pair instead of storing through it - same gradients to fp32 rounding, never a fault."""
    from gad import _capi
    B, T, heads, d = 2, 256, 21, 23
    C = heads * d
    q, k, v, do = rnd(B, T, C, seed=1, scale=0.7), rnd(B, T, C, seed=2, scale=0.7), rnd(B, T, C, seed=3), rnd(B, T, C, seed=4)
    o, lse = ops.attention_fwd_raw(q, k, v, B, heads, T, T, d, C, C, C)
    lib = _capi.load()
    res = []
    for prec in (0, 1):
        for give in ("exact", "none", "short"):
            dq, dk, dv, delta = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v), torch.empty_like(lse)
            a = ops._attention_args(q, k, v, o, lse, B, heads, T, T, d, C, C, C, T * C, T * C, T * C)
            a.d_o, a.delta, a.dq, a.dk, a.dv = do.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
            a.ld_do = a.ld_dq = a.ld_dk = a.ld_dv = C
            a.stride_do = a.stride_dq = a.stride_dk = a.stride_dv = T * C
            a.operand_precision = prec
            need = lib.gad_attention_bwd_workspace_bytes(_capi.C.byref(a))
            assert need == 4 * B * T * C * 4          # 4 key blocks of 64: the d = 23 launch takes the single-pass kernel in either mode
            gs = Guards()
            if give != "none":
                n = need if give == "exact" else need // 2
                ws = gs.scratch("attn_ws", n, dev)
                a.ws, a.ws_bytes = ws.data_ptr(), n
            _capi.check(lib.gad_attention_bwd(_capi.C.byref(a), ops._stream()), "gad_attention_bwd")
            torch.cuda.synchronize()
            if give != "none":
                gs.check()
            res.append((dq, dk, dv))
    for dq, dk, dv in res[1:]:
        for a_, b_ in zip((dq, dk, dv), res[0]):
            assert (a_ - b_).abs().max().item() < 2e-5


WINO = [  # B, Cin, Cout, H, W, upsample: tests/test_gpu_kernels.py::WINO_CASES
    (8, 64, 128, 32, 32, False), (6, 96, 192, 16, 16, False), (3, 256, 256, 8, 8, True), (5, 32, 68, 34, 30, False),
    (2, 96, 68, 20, 12, False), (4, 128, 320, 16, 16, False), (1, 64, 1024, 64, 64, False), (1, 32, 64, 4, 4, False),
    (176, 64, 128, 32, 32, False), (40, 96, 192, 32, 32, True),
]


@pytest.mark.parametrize("hint", [7, 8, 9, 10, 11, 0])
@pytest.mark.parametrize("B,Cin,Cout,H,W,ups", WINO)
def test_winograd_scratch_guard_bands(ops, B, Cin, Cout, H, W, ups, hint):
    """Forward convolution on every Winograd route (F(2x2), the planner's F(4x4), the one-launch F(4x4), the three-launch
    F(4x4), and the planner's free choice) with V / M scratch at exactly gad_gemm_wino_bytes, the split-K workspace of the
    batched products at exactly gad_gemm_workspace_bytes, and y at its exact shape - all between guard bands."""
    He, We = H * (2 if ups else 1), W * (2 if ups else 1)
    if hint in (8, 9, 10, 11) and (He % 4 or We % 4):
        pytest.skip("F(4x4) needs output maps that are multiples of 4")
    x = rnd(B, H, W, Cin, seed=1)
    w = (torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(2)) / math.sqrt(9 * Cin)).to(dev).contiguous(memory_format=torch.channels_last)
    b, temb, res = rnd(Cout, seed=3), rnd(B, Cout, seed=4), rnd(B, He, We, Cout, seed=5)
    run = lambda: ops.conv2d_fwd_raw(x, w, b, 1, (1, 1, 1, 1), ups, rowadd=temb, residual=res, tile_hint=hint)
    want = run()
    got, g = guarded(ops, run)
    assert torch.equal(got, want)
    if hint != 0:
        assert "wino" in g.kinds(), g.kinds()


@pytest.mark.parametrize("B,Cin,Cout,H,W,ups", [(4, 64, 128, 16, 16, False), (3, 96, 64, 8, 8, True), (2, 32, 68, 16, 12, False),
                                                 (8, 128, 128, 32, 32, False), (2, 320, 320, 8, 8, False)])
def test_winograd_wgrad_scratch_guard_bands(ops, B, Cin, Cout, H, W, ups):
    """Winograd weight gradient: Wy | V | dU regions of the scratch and the split-K workspace of its 36 batched products."""
    He, We = H * (2 if ups else 1), W * (2 if ups else 1)
    x, dy = rnd(B, H, W, Cin, seed=1), rnd(B, He, We, Cout, seed=6)
    w = torch.empty(Cout, Cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    run = lambda: ops.conv2d_wgrad_raw(dy, x, w, 1, (1, 1, 1, 1), ups, tile_hint=8)
    want = run()
    got, g = guarded(ops, run)
    assert torch.equal(got, want) and "wino" in g.kinds()


@pytest.mark.parametrize("case", [
    dict(B=4, Cin=256, Cout=256, H=4),                 # patch forward, split K over channel chunks
    dict(B=2, Cin=128, Cout=128, H=16),                # small launch: generic / patch plan
    dict(B=16, Cin=128, Cout=256, H=16, stride=2),     # stride-2 gather
    dict(B=2, Cin=512, Cout=256, H=8),
    dict(B=8, Cin=3, Cout=128, H=32),                  # conv_in: scalar gather (VEC = 1)
])
def test_direct_conv_guard_bands(ops, case):
    """Direct kernels (LDS-patch forward / data gradient / weight gradient with their split-K slabs and reduce, the generic
    gather, the scalar path): outputs and exact-size workspaces between guard bands."""
    B, Cin, Cout, H = case["B"], case["Cin"], case["Cout"], case["H"]
    stride = case.get("stride", 1)
    pad = (1, 1, 1, 1) if stride == 1 else (0, 1, 0, 1)
    x = rnd(B, H, H, Cin, seed=1)
    w = (torch.randn(Cout, Cin, 3, 3, generator=torch.Generator().manual_seed(2)) / math.sqrt(9 * Cin)).to(dev).contiguous(memory_format=torch.channels_last)
    with ops.kernel_flags(no_wino=True):
        want = ops.conv2d_fwd_raw(x, w, None, stride, pad, False)
        got, _ = guarded(ops, lambda: ops.conv2d_fwd_raw(x, w, None, stride, pad, False))
        assert torch.equal(got, want)
        dy = rnd(*want.shape, seed=7)
        want = ops.conv2d_wgrad_raw(dy, x, w, stride, pad, False)
        got, _ = guarded(ops, lambda: ops.conv2d_wgrad_raw(dy, x, w, stride, pad, False))
        assert torch.equal(got, want)
        if Cin % 4 == 0:                         # (conv_in's data gradient is never needed: the image has no gradient)
            with ops.kernel_flags(no_wino=True, native_dgrad=True):
                want = ops.conv2d_dgrad_raw(dy, w, x.shape, stride, pad, False)
                got, _ = guarded(ops, lambda: ops.conv2d_dgrad_raw(dy, w, x.shape, stride, pad, False))
            assert torch.equal(got, want)


@pytest.mark.parametrize("M,N,K", [(128, 256, 1024), (64, 512, 128), (2048, 320, 320), (130, 68, 100), (16, 1280, 1280)])
def test_linear_splitk_guard_bands(ops, M, N, K):
    """Linear forward / data gradient / weight gradient (the short-M split-K plans of the time-embedding and attention
    projections: gemm_nt_t64_sk4, gemm_tn_t64_skN) with exact-size workspaces."""
    x, w, dy = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=1 / math.sqrt(K)), rnd(M, N, seed=3)
    for fn in (lambda: ops.linear_fwd_raw(x, w), lambda: ops.linear_dgrad_raw(dy, w), lambda: ops.linear_wgrad_raw(dy, x)):
        want = fn()
        got, _ = guarded(ops, fn)
        assert torch.equal(got, want)

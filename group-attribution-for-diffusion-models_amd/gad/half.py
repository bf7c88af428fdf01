"""Host side of the half-precision ACTIVATION path (csrc/half.hip): the arithmetic the reference's Stable-Diffusion jobs
run - `--mixed_precision=fp16` (text_to_image/experiments/setup_train_commands.py:127,142,165, setup_unlearn_commands.py:166):
frozen U-Net weights cast to the 16-bit type (train_text_to_image_lora.py:752-760), autocast around the forward
(:1268-1270), LoRA matrices and the optimizer in fp32 (:777).  bf16 is the MI355X-native 16-bit type.

Activations and their gradients are `torch.bfloat16` NHWC / [rows, C] tensors; `gad.ops` dispatches here whenever an
operator's input is bf16, so the model classes (gad/sd.py, gad/nn.py) are the same objects on both paths.  Weights:
  * frozen parameters get bf16 copies made ONCE (plus the transposed copy a Linear's data gradient reads, and the rotated
    copy a 3x3 convolution's data gradient reads as a forward convolution), cached on the parameter;
  * trainable LoRA matrices get a bf16 shadow (and its transpose) per optimizer step, keyed like `ops.weight_key`;
  * parameter gradients are written in fp32 straight into the flat gradient buffer's slots (`ops._sink`).
Every FLOP runs in libgad_hip.so; there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import math

import torch

from . import _capi, ops
from ._capi import AttentionArgs, GroupNormArgs, HGemmArgs, check

BF16 = torch.bfloat16


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _st():
    """the current HIP stream as a void* (one C call: this path issues ~3 000 launches per training step from one host thread)"""
    if _raw_stream is not None:
        return C.c_void_p(_raw_stream(torch.cuda.current_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _reqh(t, name):
    if not (t.is_cuda and t.dtype == BF16 and t.is_contiguous()):
        raise _capi.GadError(f"{name}: expected a contiguous bf16 device tensor, got {t.dtype} {t.device} contiguous={t.is_contiguous()}")
    return t


def _empty(shape, device, dtype=BF16):
    if ops.OUT_ALLOC_DT is not None:             # out-of-bounds canaries (tests/test_gpu_half.py): outputs between poisoned bands
        return ops.OUT_ALLOC_DT(tuple(shape), device, dtype)
    return torch.empty(tuple(shape), device=device, dtype=dtype)


# ----------------------------------------------------------------------------------
# casts / transposes / small kernels
# ----------------------------------------------------------------------------------
def to_half(x):
    """fp32 -> bf16 (RNE) copy, one launch"""
    x = x.contiguous()
    if x.dtype == BF16:
        return x
    y = _empty(x.shape, x.device)
    check(_capi.load().gad_h_cast(x.data_ptr(), y.data_ptr(), x.numel(), 0, _st()), "gad_h_cast")
    return y


def to_float(x):
    x = x.contiguous()
    if x.dtype == torch.float32:
        return x
    y = _empty(x.shape, x.device, torch.float32)
    check(_capi.load().gad_h_cast(x.data_ptr(), y.data_ptr(), x.numel(), 1, _st()), "gad_h_cast")
    return y


class ToHalfFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return to_half(x)

    @staticmethod
    def backward(ctx, g):
        return to_float(g)


class ToFloatFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return to_float(x)

    @staticmethod
    def backward(ctx, g):
        return to_half(g)


def transpose_raw(x2d):
    """[R, C] (fp32 or bf16, row stride = x2d.stride(0)) -> bf16 [C, R8] with R8 = R rounded up to a multiple of 8 (the
    contraction engine reads 16-byte chunks); the pad columns are zero.  Returns the [C, R8] tensor."""
    R, Cc = x2d.shape
    if x2d.stride(1) != 1:
        raise _capi.GadError("transpose_raw: columns must be contiguous")
    R8 = (R + 7) // 8 * 8
    y = _empty((Cc, R8), x2d.device)
    if R8 != R:
        y.zero_()
    check(_capi.load().gad_h_transpose(x2d.data_ptr(), y.data_ptr(), R, Cc, x2d.stride(0), R8, int(x2d.dtype == torch.float32), _st()),
          "gad_h_transpose")
    return y


def add_raw(a, b):
    out = _empty(a.shape, a.device)
    check(_capi.load().gad_h_add(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), _st()), "gad_h_add")
    return out


# ----------------------------------------------------------------------------------
# weight copies
# ----------------------------------------------------------------------------------
def _cached(w, attr, make):
    key = ops.weight_key(w)
    c = getattr(w, attr, None)
    if c is None or c[0] != key:
        c = (key, make())
        setattr(w, attr, c)
    return c[1]


def half_weight(w):
    """bf16 copy of a parameter in its storage order ([out][in] Linear, [Cout][KH][KW][Cin] conv), refreshed when the
    parameter changes (once for a frozen weight; once per optimizer step for a LoRA matrix in the flat buffer)."""
    if w.ndim == 4:
        return _cached(w, "_gad_h", lambda: to_half(ops.weight_krsc(w).detach()))
    return _cached(w, "_gad_h", lambda: to_half(w.detach()))


def half_weight_t(w):
    """bf16 [in][out] transpose of a Linear / 1x1-conv weight: what the data gradient dx = dy W reads as B[n = in][k = out]"""
    src = w.detach().reshape(w.shape[0], -1) if w.ndim == 2 else ops.weight_krsc(w).detach().reshape(w.shape[0], -1)
    return _cached(w, "_gad_ht", lambda: transpose_raw(src))


def half_weight_rot(w):
    """bf16 W'[ci][2-r][2-s][co] = W[co][r][s][ci]: the data gradient of a 3x3 convolution as a forward gather"""
    def make():
        had = hasattr(w, "_gad_rot")
        r = to_half(ops.weight_krsc(ops.rotated_weight(w)).detach())
        if not had and hasattr(w, "_gad_rot"):
            del w._gad_rot                      # the fp32 rotated copy was only a stepping stone
        return r
    return _cached(w, "_gad_hrot", make)


def _flat_pairs(flat):
    """(bf16 shadow, bf16 shadow of the transposes) of every 2-D matrix resident in a flat parameter buffer, refreshed by ONE launch
    per optimizer step (gad_h_shadow_pairs); the table of 64 x 64 tiles is built once per buffer."""
    key = (flat._version, ops.WEIGHT_EPOCH[0], getattr(flat, "_gad_epoch", 0))
    c = getattr(flat, "_gad_hpairs", None)
    if c is None:
        rows = [(off, p_.shape[0], p_.shape[1], a, b) for p_, off, _ in getattr(flat, "_gad_params", ()) if p_.ndim == 2
                for a in range(0, p_.shape[0], 64) for b in range(0, p_.shape[1], 64)]
        table = torch.tensor(rows, dtype=torch.int64, device=flat.device)
        c = [None, torch.zeros(flat.numel(), device=flat.device, dtype=BF16), torch.zeros(flat.numel(), device=flat.device, dtype=BF16),
             table, len(rows), {}]
        flat._gad_hpairs = c
    versions = {o_: p_._version for p_, o_, _ in getattr(flat, "_gad_params", ())}
    if c[0] != key or c[5] != versions:
        check(_capi.load().gad_h_shadow_pairs(flat.data_ptr(), c[1].data_ptr(), c[2].data_ptr(), c[3].data_ptr(), c[4], _st()), "gad_h_shadow_pairs")
        c[0], c[5] = key, versions
    return c[1], c[2]


def _half_of(w):
    """bf16 copy of a 2-D parameter: a slice of its flat buffer's shadow where it lives in one"""
    if ops._in_flat_buffer(w):
        flat, off, n = w._gad_flat
        return _flat_pairs(flat)[0][off:off + n].view(w.shape)
    return to_half(w.detach())


def _half_t_of(w):
    """bf16 transpose [cols][rows] of a 2-D parameter (slice of the flat buffer's transposed shadow where it lives in one)"""
    if ops._in_flat_buffer(w):
        flat, off, n = w._gad_flat
        return _flat_pairs(flat)[1][off:off + n].view(w.shape[1], w.shape[0])
    return transpose_raw(to_half(w.detach()))


def lora_half(down, up):
    """bf16 shadows of a LoRA pair, the rank padded with zeros to a multiple of 8 (ragged pruned ranks,
    text_to_image/prune_lora.py:173-180): (down [r8][K], up [N][r8], down^T [K][r8], up^T [r8][N]); refreshed when either
    matrix changes (once per optimizer step)."""
    key = (ops.weight_key(down), ops.weight_key(up))
    c = getattr(down, "_gad_hlora", None)
    if c is None or c[0] != key:
        r, K = down.shape
        N = up.shape[0]
        r8 = (r + 7) // 8 * 8
        if r8 == r and N % 8 == 0 and K % 8 == 0:
            c = (key, (_half_of(down), _half_of(up), _half_t_of(down), _half_t_of(up)))
            down._gad_hlora = c
            return c[1]
        if r8 == r:
            dh, uh = to_half(down.detach()), to_half(up.detach())
        else:
            dh = torch.zeros((r8, K), device=down.device, dtype=BF16)
            uh = torch.zeros((N, r8), device=down.device, dtype=BF16)
            dh[:r].copy_(down.detach())
            uh[:, :r].copy_(up.detach())
        c = (key, (dh, uh, transpose_raw(dh), transpose_raw(uh)))
        down._gad_hlora = c
    return c[1]


# ----------------------------------------------------------------------------------
# hgemm
# ----------------------------------------------------------------------------------
def hgemm_raw(A, B, out, M, N, K, lda, ldb, ldc, *, A2=None, B2=None, lda2=0, ldb2=0, k_split=None, conv=0, geom=None,
              alpha=1.0, bias=None, rowadd=None, rows_per_group=1, residual=None, ldr=0, out_f32=False, accumulate=False,
              tile_hint=0, splitk_hint=0):
    lib = _capi.load()
    a = HGemmArgs()
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), out.data_ptr()
    a.A2 = A2.data_ptr() if A2 is not None else None
    a.B2 = B2.data_ptr() if B2 is not None else None
    a.M, a.N, a.K = M, N, K
    a.lda, a.lda2, a.ldb, a.ldb2, a.ldc = lda, lda2, ldb, ldb2, ldc
    a.conv = conv
    if geom is not None:
        (a.H, a.W, a.Cin, a.Ho, a.Wo, a.KH, a.KW, a.stride, a.pad_t, a.pad_l, a.upsample) = geom
    a.k_split = (a.Cin if conv else K) if k_split is None else k_split
    a.alpha = alpha
    a.bias = bias.data_ptr() if bias is not None else None
    if rowadd is not None:
        a.rowadd, a.rows_per_group = rowadd.data_ptr(), rows_per_group
        a.ld_rowadd = rowadd.stride(0) if rowadd.ndim == 2 else rowadd.shape[-1]
    if residual is not None:
        a.residual, a.ldr = residual.data_ptr(), ldr
    a.out_f32, a.accumulate = int(out_f32), int(accumulate)
    a.tile_hint, a.splitk_hint = tile_hint, splitk_hint
    if ops.SCRATCH_ALLOC is not None or ops.PROFILER is not None:       # canary / profiled runs: the workspace at exactly the size the planner asks for
        need = lib.gad_hgemm_workspace_bytes(C.byref(a))
        if need < 0:
            raise _capi.GadError(f"gad_hgemm: {lib.gad_last_error().decode()}")
        if need:
            ws = ops._scratch("ws", need, A.device) if ops.SCRATCH_ALLOC is not None else ops.workspace(A.device)
            if ws.numel() < need:
                ws = torch.empty(need, dtype=torch.uint8, device=A.device)
            a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
        if ops.PROFILER is not None:
            ops.PROFILER.hgemm(lib, a)
            return
    else:                                                                # product path: hand the stream's 1 GiB workspace over, one C call
        ws = ops.workspace(A.device)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
        rc = lib.gad_hgemm(C.byref(a), _st())
        if rc == 0:
            return
        need = lib.gad_hgemm_workspace_bytes(C.byref(a))                 # a split-K plan larger than the shared workspace: its own allocation
        if need <= ws.numel():
            check(rc, "gad_hgemm")
        big = torch.empty(need, dtype=torch.uint8, device=A.device)
        a.ws, a.ws_bytes = big.data_ptr(), need
    check(lib.gad_hgemm(C.byref(a), _st()), "gad_hgemm")


def conv_fwd_raw(x, wh, bias, KH, KW, stride=1, pad=(1, 1, 1, 1), upsample=False, rowadd=None, residual=None, x2=None,
                 conv=1, out_hw=None, out_f32=False, tile_hint=0):
    """x [B,H,W,C1] (+ x2 [B,H,W,C2]: channel concat read in place) bf16, wh bf16 [Cout][KH][KW][Cin] -> [B,Ho,Wo,Cout]."""
    Bn, H, W, C1 = x.shape
    Cin = C1 + (x2.shape[-1] if x2 is not None else 0)
    Cout = wh.shape[0]
    if wh.numel() != Cout * KH * KW * Cin:
        raise _capi.GadError(f"half conv: weight {tuple(wh.shape)} does not fit {Cin} input channels")
    if out_hw is None:
        He, We = (2 * H, 2 * W) if upsample else (H, W)
        Ho = (He + pad[0] + pad[1] - KH) // stride + 1
        Wo = (We + pad[2] + pad[3] - KW) // stride + 1
    else:
        Ho, Wo = out_hw
    y = _empty((Bn, Ho, Wo, Cout), x.device, torch.float32 if out_f32 else BF16)
    geom = (H, W, Cin, Ho, Wo, KH, KW, stride, pad[0], pad[2], int(upsample))
    hgemm_raw(x, wh, y, Bn * Ho * Wo, Cout, KH * KW * Cin, C1, KH * KW * Cin, Cout, A2=x2,
              lda2=(x2.shape[-1] if x2 is not None else 0), k_split=C1, conv=conv, geom=geom, bias=bias, rowadd=rowadd,
              rows_per_group=Ho * Wo, residual=residual, ldr=Cout, out_f32=out_f32, tile_hint=tile_hint)
    return y


def linear_raw(x2d, wh, bias=None, residual=None, alpha=1.0, A2=None, B2=None, out=None, out_f32=False, accumulate=False):
    """y[M][N] = alpha * ([x2d | A2] . [wh | B2]^T) + bias + residual; wh [N][K] bf16"""
    M, K = x2d.shape
    N = wh.shape[0]
    k2 = A2.shape[1] if A2 is not None else 0
    y = _empty((M, N), x2d.device, torch.float32 if out_f32 else BF16) if out is None else out
    hgemm_raw(x2d, wh, y, M, N, K + k2, x2d.stride(0), wh.stride(0), y.stride(0), A2=A2, B2=B2, lda2=(A2.stride(0) if A2 is not None else 0),
              ldb2=(B2.stride(0) if B2 is not None else 0), k_split=K, alpha=alpha, bias=bias, residual=residual,
              ldr=(residual.stride(0) if residual is not None else 0), out_f32=out_f32, accumulate=accumulate)
    return y


def wgrad_raw(dy2d, x2d, out, accumulate, alpha=1.0, splitk_hint=0):
    """out[N][K] (fp32) (+)= alpha * dy2d[M][:N]^T x2d[M][:K]: the LoRA parameter gradients (N, K = out's shape; the operands may
    carry zero pad columns).  A contraction over the token axis M with both operands read in place (gad_hgemm_tn: transposing LDS
    reads), split along M."""
    lib = _capi.load()
    N, K = out.shape
    a = HGemmArgs()
    a.A, a.B, a.C = dy2d.data_ptr(), x2d.data_ptr(), out.data_ptr()
    a.M, a.N, a.K = N, K, dy2d.shape[0]
    a.lda, a.ldb, a.ldc = dy2d.stride(0), x2d.stride(0), out.stride(0)
    a.alpha, a.out_f32, a.accumulate, a.splitk_hint = alpha, 1, int(accumulate), splitk_hint
    need = lib.gad_hgemm_tn_workspace_bytes(C.byref(a))
    if need < 0:
        raise _capi.GadError(f"gad_hgemm_tn: {lib.gad_last_error().decode()}")
    if need:
        ws = ops._scratch("ws", need, dy2d.device) if ops.SCRATCH_ALLOC is not None else ops.workspace(dy2d.device)
        if ws.numel() < need:
            ws = torch.empty(need, dtype=torch.uint8, device=dy2d.device)
        a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    if ops.PROFILER is not None:
        ops.PROFILER.hgemm(lib, a, tn=True)
        return
    check(lib.gad_hgemm_tn(C.byref(a), _st()), "gad_hgemm_tn")


# ----------------------------------------------------------------------------------
# autograd functions
# ----------------------------------------------------------------------------------
def _frozen(*params):
    """The path computes data gradients and LoRA gradients only: a base parameter that autograd would want a gradient for
    (requires_grad with grad mode on - checked by the public operators below, where grad mode is still the caller's; sampling under
    no_grad never trips this) is an error, not a silent zero gradient."""
    if torch.is_grad_enabled() and any(p is not None and p.requires_grad for p in params):
        raise _capi.GadError("half-precision activation path: base parameters must be frozen (the path trains LoRA matrices only, "
                             "as the reference's mixed-precision jobs do: train_text_to_image_lora.py:746-760)")


class HConv2dFn(torch.autograd.Function):
    """bf16 convolution with the fused epilogue of ops.Conv2dFn; frozen weight (data gradient only)."""

    @staticmethod
    def forward(ctx, x, w, bias, rowadd, residual, stride, pad, upsample):
        _reqh(x, "half conv x")
        Cout, Cin, KH, KW = w.shape
        ctx.w, ctx.cfg, ctx.xshape = w, (stride, tuple(pad), upsample, residual is not None), x.shape
        if KH == 1 and KW == 1 and stride == 1 and not upsample and tuple(pad) == (0, 0, 0, 0) and rowadd is None:
            r2 = residual.view(-1, Cout) if residual is not None else None                  # a 1x1 convolution is a dense GEMM over the pixel rows
            return linear_raw(x.view(-1, Cin), half_weight(w).view(Cout, Cin), bias, r2).view(*x.shape[:-1], Cout)
        return conv_fwd_raw(x, half_weight(w), bias, KH, KW, stride, pad, upsample, rowadd=rowadd, residual=residual)

    @staticmethod
    def backward(ctx, dy):
        w = ctx.w
        stride, pad, upsample, has_res = ctx.cfg
        dy = dy.contiguous()
        Cout, Cin, KH, KW = w.shape
        Bn, H, W, _ = ctx.xshape
        dx = None
        if ctx.needs_input_grad[0]:
            if KH == 1 and KW == 1 and stride == 1 and not upsample:
                dx = linear_raw(dy.view(-1, Cout), half_weight_t(w)).view(ctx.xshape)
            elif KH == 3 and KW == 3 and stride == 1 and pad == (1, 1, 1, 1):
                wr = half_weight_rot(w)
                if Cout % 8:                      # conv_out (4 channels): the gather reads 16-byte chunks - zero-pad dy and W' to 8 channels
                    dy = torch.nn.functional.pad(dy, (0, 8 - Cout % 8))
                    wr = _cached(w, "_gad_hrot8", lambda: torch.nn.functional.pad(wr, (0, 8 - Cout % 8)).contiguous())
                dxe = conv_fwd_raw(dy, wr, None, 3, 3)                                    # forward conv of dy with the rotated weight
                if upsample:
                    dx = _empty(ctx.xshape, dy.device)
                    check(_capi.load().gad_h_upsample2x_bwd(dxe.data_ptr(), dx.data_ptr(), Bn, H, W, Cin, _st()), "gad_h_upsample2x_bwd")
                else:
                    dx = dxe
            elif KH == 3 and KW == 3 and stride == 2 and not upsample:
                # dx[y][x] = sum_{r,s} dy[(y + pad_t - r) / 2][...] W[.][r][s][.] over the taps where the division is exact
                # = the conv == 2 gather with the rotated weight and pads (2 - pad_t, 2 - pad_l)
                dx = conv_fwd_raw(dy, half_weight_rot(w), None, 3, 3, 1, (2 - pad[0], 0, 2 - pad[2], 0), conv=2, out_hw=(H, W))
            else:
                raise _capi.GadError(f"half conv backward: unsupported geometry k={KH} stride={stride}")
        dres = dy if (has_res and ctx.needs_input_grad[4]) else None
        return dx, None, None, None, dres, None, None, None


class HLinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, residual):
        _reqh(x, "half linear x")
        shp = x.shape
        r2 = residual.view(-1, w.shape[0]) if residual is not None else None
        ctx.w, ctx.shp, ctx.has_res = w, shp, residual is not None
        return linear_raw(x.view(-1, shp[-1]), half_weight(w), bias, r2).view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        w = ctx.w
        dy2 = dy.contiguous().view(-1, w.shape[0])
        dx = linear_raw(dy2, half_weight_t(w)).view(ctx.shp) if ctx.needs_input_grad[0] else None
        return dx, None, None, (dy if (ctx.has_res and ctx.needs_input_grad[3]) else None)


class HLoraLinearFn(torch.autograd.Function):
    """y = x W^T + b + s (x A^T) B^T [+ residual] with frozen W, b and trainable fp32 A (down), B (up)
    (diffusers LoRACompatibleLinear, train_text_to_image_lora.py:786-820).  Forward: mid = s x A^T, then ONE launch over the
    concatenated K axis [x | mid] . [W | B]^T.  Backward: dmid = s dy B; dx = [dy | dmid] . [W^T | A^T]^T in one launch;
    dB = dy^T mid and dA = dmid^T x in fp32 into the flat gradient buffer."""

    @staticmethod
    def forward(ctx, x, w, bias, down, up, s, residual):
        _reqh(x, "half lora linear x")
        shp = x.shape
        x2 = x.view(-1, shp[-1])
        N = w.shape[0]
        dh, uh, _, _ = lora_half(down, up)
        mid = linear_raw(x2, dh, alpha=s)
        r2 = residual.view(-1, N) if residual is not None else None
        y = linear_raw(x2, half_weight(w), bias, r2, A2=mid, B2=uh)
        ctx.save_for_backward(x2, mid)
        ctx.meta = (w, down, up, shp, s, residual is not None)
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, mid = ctx.saved_tensors
        w, down, up, shp, s, has_res = ctx.meta
        N = w.shape[0]
        dy2 = dy.contiguous().view(-1, N)
        need_x = ctx.needs_input_grad[0]
        _, _, dht, uht = lora_half(down, up)
        dmid = None
        if need_x or ctx.needs_input_grad[3]:
            dmid = linear_raw(dy2, uht, alpha=s)                                              # s dy B   (B^T as [r8][N])
        dx = None
        if need_x:
            dx = linear_raw(dy2, half_weight_t(w), A2=dmid, B2=dht).view(shp)                 # dy W + dmid A
        dd = du = None
        r, K = down.shape
        if ctx.needs_input_grad[3] and ctx.needs_input_grad[4] and N * K <= 1.5 * r * (N + K):
            # Wide ranks (the reference's r = 256 on 320-wide projections): the two LoRA gradients through the FULL-weight gradient
            # G = dy^T x (fp32, [N][K]):  dB = s G A^T,  dA = s B^T G  - the same sums reassociated.  One contraction over the
            # token axis instead of two (2 M N K against 2 M r (N + K) FLOPs), one operand transpose (dy; x's is shared by the
            # projections of an attention) instead of three; the two [N x K x r] products that follow are small fp32 GEMMs.
            G = _empty((N, K), dy2.device, torch.float32)
            wgrad_raw(dy2, x2, G, accumulate=False)
            du = _small_grad32(up, lambda o, acc: ops.gemm_raw(G, down.detach(), o, ops.A_KC, ops.B_KC, N, r, K, K, K, r, alpha=s,
                                                             residual=(o if acc else None), ldr=r, force_f32=True))
            dd = _small_grad32(down, lambda o, acc: ops.gemm_raw(up.detach(), G, o, ops.A_MC, ops.B_MC, r, K, N, r, K, K, alpha=s,
                                                               residual=(o if acc else None), ldr=K, force_f32=True))
        else:
            if ctx.needs_input_grad[3]:
                dd = _param_grad32(down, dmid, x2)                                            # dmid^T x
            if ctx.needs_input_grad[4]:
                du = _param_grad32(up, dy2, mid)                                              # dy^T mid
        dres = dy if (has_res and ctx.needs_input_grad[6]) else None
        return dx, None, None, dd, du, None, dres


def _param_grad32(param, dy2d, x2d):
    """fp32 gradient of a [N][K] parameter: straight into its flat-buffer slot when a FusedTrainer backward is running"""
    v, first = ops._sink(param)
    if v is not None:
        wgrad_raw(dy2d, x2d, v.view(param.shape), accumulate=not first)
        return None
    out = _empty(param.shape, dy2d.device, torch.float32)
    wgrad_raw(dy2d, x2d, out, accumulate=False)
    return out


def _small_grad32(param, launch):
    """launch(out, accumulate) writes an fp32 gradient of param's shape: into the flat-buffer slot (first gradient of the step
    overwrites, later ones add) or a fresh tensor handed back to autograd"""
    v, first = ops._sink(param)
    if v is not None:
        launch(v, not first)
        return None
    out = _empty(param.shape, param.device, torch.float32)
    launch(out, False)
    return out


def _gn_args(x, x2, y, gamma, beta, mean, rstd, G, eps, silu):
    a = GroupNormArgs()
    Bn = x.shape[0]
    C_ = x.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
    a.x, a.y, a.gamma, a.beta = x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
    a.B, a.HW, a.C, a.G = Bn, x.numel() // (Bn * x.shape[-1]), C_, G
    a.eps, a.silu = eps, int(silu)
    if x2 is not None:
        a.x2, a.C1 = x2.data_ptr(), x.shape[-1]
    need = _capi.load().gad_h_groupnorm_workspace_bytes(C.byref(a))
    if need < 0:
        raise _capi.GadError("half groupnorm: unsupported shape (C % 8)")
    ws = ops._scratch("ws", need, x.device) if ops.SCRATCH_ALLOC is not None else ops.workspace(x.device)
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    return a


def group_norm_raw(x, x2, gamma, beta, G, eps, silu):
    _reqh(x, "half groupnorm x")
    C_ = x.shape[-1] + (x2.shape[-1] if x2 is not None else 0)
    y = _empty((*x.shape[:-1], C_), x.device)
    mean = _empty((x.shape[0], G), x.device, torch.float32)
    rstd = _empty((x.shape[0], G), x.device, torch.float32)
    a = _gn_args(x, x2, y, gamma, beta, mean, rstd, G, eps, silu)
    check(_capi.load().gad_h_groupnorm_silu_fwd(C.byref(a), _st()), "gad_h_groupnorm_silu_fwd")
    return y, mean, rstd


class HGroupNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, silu, bypass):
        y, mean, rstd = group_norm_raw(x, None, gamma, beta, G, eps, silu)
        ctx.save_for_backward(x, mean, rstd)
        ctx.cfg = (gamma, beta, G, eps, silu)
        if bypass:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dbypass=None):
        if dy is None:
            return dbypass, None, None, None, None, None, None
        x, mean, rstd = ctx.saved_tensors
        gamma, beta, G, eps, silu = ctx.cfg
        dy = dy.contiguous()
        dx = _empty(x.shape, x.device)
        a = _gn_args(x, None, dx, gamma, beta, mean, rstd, G, eps, silu)
        a.dy = dy.data_ptr()
        if dbypass is not None:
            a.dx_add = _reqh(dbypass.contiguous(), "half groupnorm bypass gradient").data_ptr()
        check(_capi.load().gad_h_groupnorm_silu_bwd(C.byref(a), _st()), "gad_h_groupnorm_silu_bwd")
        return dx, None, None, None, None, None, None


class HLayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps, bypass):
        _reqh(x, "half layernorm x")
        C_ = x.shape[-1]
        rows = x.numel() // C_
        y = _empty(x.shape, x.device)
        mean = _empty((rows,), x.device, torch.float32)
        rstd = _empty((rows,), x.device, torch.float32)
        check(_capi.load().gad_h_layernorm_fwd(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(),
                                               rstd.data_ptr(), rows, C_, eps, _st()), "gad_h_layernorm_fwd")
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma = gamma
        if bypass:
            return y, x.view_as(x)
        return y

    @staticmethod
    def backward(ctx, dy, dbypass=None):
        if dy is None:
            return dbypass, None, None, None, None
        x, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        C_ = x.shape[-1]
        dx = _empty(x.shape, x.device)
        add = _reqh(dbypass.contiguous(), "half layernorm bypass gradient") if dbypass is not None else None
        check(_capi.load().gad_h_layernorm_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), add.data_ptr() if add is not None else None,
                                               ctx.gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), x.numel() // C_, C_, _st()),
              "gad_h_layernorm_bwd")
        return dx, None, None, None, None


class HGegluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h):
        _reqh(h, "half geglu h")
        F2 = h.shape[-1]
        out = _empty((*h.shape[:-1], F2 // 2), h.device)
        check(_capi.load().gad_h_geglu_fwd(h.data_ptr(), out.data_ptr(), h.numel() // F2, F2 // 2, _st()), "gad_h_geglu_fwd")
        ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        dout = dout.contiguous()
        F2 = h.shape[-1]
        dh = _empty(h.shape, h.device)
        check(_capi.load().gad_h_geglu_bwd(h.data_ptr(), dout.data_ptr(), dh.data_ptr(), h.numel() // F2, F2 // 2, _st()), "gad_h_geglu_bwd")
        return dh


def _attn_args(q, k, v, o, lse, Bn, heads, Tq, Tk, d, scale):
    a = AttentionArgs()
    a.q, a.k, a.v, a.o, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), (lse.data_ptr() if lse is not None else None)
    a.B, a.heads, a.Tq, a.Tk, a.d = Bn, heads, Tq, Tk, d
    Cq = heads * d
    a.ldq = a.ldk = a.ldv = a.ldo = Cq
    a.stride_q, a.stride_k, a.stride_v, a.stride_o = Tq * Cq, Tk * Cq, Tk * Cq, Tq * Cq
    a.scale = 1.0 / math.sqrt(d) if scale is None else scale
    if scale is not None:
        a.alg_d = int(round(1.0 / (scale * scale)))
    a.operand_precision = 1
    return a


class HAttentionFn(torch.autograd.Function):
    """F.scaled_dot_product_attention on bf16 [B, T, heads*d] operands (attention_processor.py:1314-1325 under autocast)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale=None):
        for t_, n_ in ((q, "q"), (k, "k"), (v, "v")):
            _reqh(t_, n_)
        Bn, Tq, Cq = q.shape
        Tk, d = k.shape[1], Cq // heads
        o = _empty((Bn, Tq, Cq), q.device)
        lse = _empty((Bn, heads, Tq), q.device, torch.float32)
        a = _attn_args(q, k, v, o, lse, Bn, heads, Tq, Tk, d, scale)
        fn = _capi.load().gad_h_attention_fwd
        if ops.PROFILER is not None:
            ops.PROFILER.attention(fn, a, "fwd", elem_bytes=2)
        else:
            check(fn(C.byref(a), _st()), "gad_h_attention_fwd")
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.heads, ctx.scale = heads, scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        heads = ctx.heads
        do = do.contiguous()
        Bn, Tq, Cq = q.shape
        Tk, d = k.shape[1], Cq // heads
        dq, dk, dv = _empty(q.shape, q.device), _empty(k.shape, k.device), _empty(v.shape, v.device)
        delta = _empty(lse.shape, lse.device, torch.float32)
        a = _attn_args(q, k, v, o, lse, Bn, heads, Tq, Tk, d, ctx.scale)
        a.d_o, a.delta, a.dq, a.dk, a.dv = do.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
        a.ld_do = a.ld_dq = a.ld_dk = a.ld_dv = Cq
        a.stride_do = a.stride_dq = Tq * Cq
        a.stride_dk = a.stride_dv = Tk * Cq
        fn = _capi.load().gad_h_attention_bwd
        if ops.PROFILER is not None:
            ops.PROFILER.attention(fn, a, "bwd", elem_bytes=2)
        else:
            check(fn(C.byref(a), _st()), "gad_h_attention_bwd")
        return dq, dk, dv, None, None


class HConcatFn(torch.autograd.Function):
    """torch.cat([h, skip], channels) for bf16 NHWC tensors (training: the consumers' backward wants one tensor)"""

    @staticmethod
    def forward(ctx, a, b):
        ctx.cs = (a.shape[-1], b.shape[-1])
        return torch.cat([a, b], dim=-1)

    @staticmethod
    def backward(ctx, d):
        c1, _ = ctx.cs
        return d[..., :c1].contiguous(), d[..., c1:].contiguous()


# ----------------------------------------------------------------------------------
# public operators (what gad.ops dispatches to for bf16 inputs)
# ----------------------------------------------------------------------------------
def conv2d(x, w, bias, rowadd, residual, stride, pad, upsample):
    _frozen(w, bias, rowadd)                     # (the time-embedding row comes from frozen layers too)
    return HConv2dFn.apply(x, w, bias, rowadd, residual, stride, tuple(pad), upsample)


def linear(x, w, bias, residual):
    _frozen(w, bias)
    return HLinearFn.apply(x, w, bias, residual)


def lora_linear(x, w, bias, down, up, s, residual):
    _frozen(w, bias)
    return HLoraLinearFn.apply(x, w, bias, down, up, float(s), residual)


def group_norm(x, gamma, beta, G, eps, silu):
    _frozen(gamma, beta)
    return HGroupNormFn.apply(x, gamma, beta, G, eps, silu, False)


def group_norm_bypass(x, gamma, beta, G, eps, silu):
    _frozen(gamma, beta)
    if not (torch.is_grad_enabled() and x.requires_grad):
        return group_norm(x, gamma, beta, G, eps, silu), x
    return HGroupNormFn.apply(x, gamma, beta, G, eps, silu, True)


def layer_norm(x, gamma, beta, eps):
    _frozen(gamma, beta)
    return HLayerNormFn.apply(x, gamma, beta, eps, False)


def layer_norm_bypass(x, gamma, beta, eps):
    _frozen(gamma, beta)
    if not (torch.is_grad_enabled() and x.requires_grad):
        return layer_norm(x, gamma, beta, eps), x
    return HLayerNormFn.apply(x, gamma, beta, eps, True)


def geglu(h):
    return HGegluFn.apply(h)


def attention_core(q, k, v, heads, scale=None):
    return HAttentionFn.apply(q, k, v, heads, scale)


def concat(a, b):
    return HConcatFn.apply(a, b)

"""Host-side operators over the C ABI: raw launch wrappers (``*_raw``) and
``torch.autograd.Function`` s that chain them.  PyTorch supplies device memory, the
stream and the autograd graph; every FLOP runs in libgad_hip.so.

Layout: activations are contiguous fp32 NHWC tensors ``[B, H, W, C]`` (or ``[M, C]``),
conv weights are parameters of logical shape ``[Cout, Cin, KH, KW]`` whose storage is
``[Cout, KH, KW, Cin]`` (torch channels_last), Linear weights ``[out, in]``.
"""
from __future__ import annotations

import ctypes as C
import math
import os
from typing import Optional

import torch

from . import _capi
from ._capi import (A_CONV, A_CONVT, A_KC, A_MC, B_CONV, B_KC, B_MC, B_WDGRAD, AdamArgs, ConvGeom, GemmArgs,
                    GroupNormArgs, check)

# ----------------------------------------------------------------------------------
# plumbing
# ----------------------------------------------------------------------------------
_WS = {}
WS_BYTES = 1 << 30


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


WS_OVERRIDE = [None]         # a caller-provided workspace that replaces the per-stream one (FusedTrainer's graph capture: the launches recorded
#                              into a hipGraph must not share scratch with other streams' work, whatever stream the capture ran on)


def workspace(device) -> torch.Tensor:
    """One caller-owned scratch buffer per device AND stream (split-K partials, reduction partials): launches on one stream are
    ordered, so they can share it; two coalitions in flight on two streams (a training phase beside a sampling phase,
    coalition.run_pipelined) must not.  Allocated once per stream so that hipGraph replays see a fixed address."""
    if WS_OVERRIDE[0] is not None:
        return WS_OVERRIDE[0]
    stream = _raw_stream(device.index if device.index is not None else torch.cuda.current_device()) if _raw_stream is not None \
        else torch.cuda.current_stream(device).cuda_stream
    key = (device.type, device.index, stream)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty(WS_BYTES, dtype=torch.uint8, device=device)
        _WS[key] = ws
    return ws


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _req(t: torch.Tensor, name: str):
    if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
        raise _capi.GadError(f"{name}: expected a contiguous fp32 device tensor, got {t.dtype} {t.device} "
                             f"contiguous={t.is_contiguous()}")
    return t


def weight_krsc(w: torch.Tensor) -> torch.Tensor:
    """View of a conv parameter as its physical [Cout, KH, KW, Cin] array (no copy)."""
    v = w.permute(0, 2, 3, 1)
    if not v.is_contiguous():
        raise _capi.GadError("conv weight storage must be [Cout,KH,KW,Cin] (channels_last)")
    return v


# ----------------------------------------------------------------------------------
# gradient sinks: when a parameter lives in a flat gradient buffer (training.flatten_params) its gradient
# kernels write straight into that slot while a FusedTrainer backward is running, and autograd is handed None for
# that parameter: no AccumulateGrad node runs, so there is no per-parameter clone / accumulate kernel and no
# memset of the flat buffer.  A second gradient for the same parameter within one step is added on top.
# Outside begin_/end_backward_step the sinks are ignored and .grad behaves as usual.
# ----------------------------------------------------------------------------------
_SINK_EPOCH = [0]        # sinks live on the parameter object: p._gad_sink (view), p._gad_sink_epoch (last write)
_SINK_ACTIVE = [False]


def begin_backward_step():
    """Call once per training step before backward(): marks every sink as not yet written and routes parameter
    gradients into the sinks until end_backward_step()."""
    _SINK_EPOCH[0] += 1
    _SINK_ACTIVE[0] = True


def end_backward_step():
    _SINK_ACTIVE[0] = False


def _sink(param):
    """(destination view or None, first_write flag)"""
    v = getattr(param, "_gad_sink", None) if (param is not None and _SINK_ACTIVE[0]) else None
    if v is None:
        return None, True
    first = getattr(param, "_gad_sink_epoch", -1) != _SINK_EPOCH[0]
    param._gad_sink_epoch = _SINK_EPOCH[0]
    return v, first


def _deliver(param, grad):
    """Route a freshly computed gradient tensor: into the parameter's sink (returning None to autograd) if it has an
    active one, else back to autograd unchanged."""
    v, first = _sink(param)
    if v is None:
        return grad
    if first:
        v.copy_(grad)
    else:
        v.add_(grad)
    return None


# ----------------------------------------------------------------------------------
# raw contraction launches
# ----------------------------------------------------------------------------------
OPERAND_PRECISION = [0]      # 0: fp32 operands (reference default); 1: bf16 operands, fp32 storage (see gad.h); 2: bf16 ACTIVATIONS
#                              (gad/half.py: models that support it keep activations and their gradients in bf16; everything
#                              that still flows as fp32 - conv_in, the time-embedding MLP - runs as mode 1)


class operand_precision:
    """``with ops.operand_precision("bf16"):`` - the analogue of the reference's autocast
    (`--mixed_precision`, text_to_image/train_text_to_image_lora.py:659-668): contractions that have a bf16
    instance (conv forward, Linear forward, Q K^T) round A and B to bf16 in flight and accumulate in fp32;
    everything else, and all storage, stays fp32."""

    def __init__(self, name):
        if name not in ("f32", "fp32", "no", "bf16", "bf16-operands", "bf16-activations"):
            raise ValueError(f"operand precision {name!r}: use 'f32', 'bf16' (= 'bf16-activations') or 'bf16-operands'")
        self.value = {"bf16": 2, "bf16-activations": 2, "bf16-operands": 1}.get(name, 0)

    def __enter__(self):
        self.prev = OPERAND_PRECISION[0]
        OPERAND_PRECISION[0] = self.value
        return self

    def __exit__(self, *exc):
        OPERAND_PRECISION[0] = self.prev
        return False


def set_operand_precision(name):
    """Process-wide form of `operand_precision` for the entry points' --mixed_precision flag.  "fp16" maps to
    bf16: the MI355X-native 16-bit type, which with fp32's exponent range needs no loss scaling (the reference's
    GradScaler, accelerate's fp16 path).  "fp16" / "bf16" = half-precision ACTIVATIONS where the model supports them
    (UNet2DConditionModel: what the reference's SD jobs run) and bf16 operands elsewhere; "bf16-operands" = bf16
    operands with fp32 storage everywhere (the round-2..4 mode, kept for A/B)."""
    OPERAND_PRECISION[0] = 0 if name in (None, "no", "f32", "fp32") else operand_precision("bf16" if name in ("fp16", "bf16") else name).value


def half_activations() -> bool:
    """Is the half-precision activation path selected? (models that support it cast at their boundary: gad/sd.py)"""
    return OPERAND_PRECISION[0] == 2


# Kernel-family switches for A/B tools and the invariance tests (never set in production): they travel in the argument
# structs (gad_gemm_args.flags / gad_groupnorm_args.flags); the library itself reads no environment variable.
KERNEL_FLAGS = {"gemm": 0, "gn": 0, "native_dgrad": False, "two_kernel_attn_bwd": False, "narrow_attn_fwd": False, "no_wino4": False,
                "no_gn_wino": False}


class kernel_flags:
    """``with ops.kernel_flags(no_patch=True, tap_major_k=True, scalar_epilogue=True, general_loaders=True, gn_two_pass=True,
    native_dgrad=True): ...``  (native_dgrad is a host-side switch: data-gradient kernels instead of forward kernels on
    rotated weights, see `dgrad_as_forward`)"""

    def __init__(self, no_patch=False, tap_major_k=False, gn_two_pass=False, scalar_epilogue=False, general_loaders=False,
                 native_dgrad=False, two_kernel_attn_bwd=False, narrow_attn_fwd=False, no_wino=False, no_wino4=False, no_gn_wino=False):
        self.native_dgrad, self.two_kernel_attn_bwd, self.narrow_attn_fwd, self.no_wino4 = native_dgrad, two_kernel_attn_bwd, narrow_attn_fwd, no_wino4
        self.no_gn_wino = no_gn_wino
        self.gemm = ((_capi.GEMM_NO_PATCH if no_patch else 0) | (_capi.GEMM_TAP_MAJOR_K if tap_major_k else 0)
                     | (_capi.GEMM_NO_WINO if no_wino else 0)
                     | (_capi.GEMM_SCALAR_EPILOGUE if scalar_epilogue else 0)
                     | (_capi.GEMM_GENERAL_LOADERS if general_loaders else 0))
        self.gn = _capi.GN_TWO_PASS if gn_two_pass else 0

    def __enter__(self):
        self.prev = dict(KERNEL_FLAGS)
        KERNEL_FLAGS["gemm"], KERNEL_FLAGS["gn"], KERNEL_FLAGS["native_dgrad"] = self.gemm, self.gn, self.native_dgrad
        KERNEL_FLAGS["two_kernel_attn_bwd"], KERNEL_FLAGS["narrow_attn_fwd"] = self.two_kernel_attn_bwd, self.narrow_attn_fwd
        KERNEL_FLAGS["no_wino4"] = self.no_wino4
        KERNEL_FLAGS["no_gn_wino"] = self.no_gn_wino
        return self

    def __exit__(self, *exc):
        KERNEL_FLAGS.update(self.prev)
        return False


# Allocation hooks for the out-of-bounds canaries (tests/test_gpu_guards.py; never set in production): SCRATCH_ALLOC(kind,
# nbytes, device) -> uint8 tensor provides every caller-owned scratch region at EXACTLY the size the C ABI's *_bytes query
# returns (kind: "ws" split-K / reduction workspace, "wino" Winograd scratch, "attn_ws" dQ slabs); OUT_ALLOC(shape, device)
# provides output tensors.  The tests hand out views between poisoned guard bands and check the bands afterwards.
SCRATCH_ALLOC = None
OUT_ALLOC = None
OUT_ALLOC_DT = None          # the same for the half path's outputs: (shape, device, dtype) -> tensor (gad/half.py::_empty)


ROUTE_STATS = None         # a dict while tools/wino_route_stats.py counts which 3x3 launches take a Winograd route and who made their V
# Training: GroupNorm writing the Winograd input image itself (as in sampling).  OFF by default: at B = 128 it is no faster than GroupNorm + the
# route's input transform (31.2 vs 31.3 ms per step: the activation round trip stays in the Infinity Cache) and it changes fp32 rounding, i.e. the
# 1000-step trajectory; with it off the training step's results are bit-identical to the separate launches'.  GAD_TRAIN_GN_WINO=1 switches it on.
TRAIN_GN_WINO = [os.environ.get("GAD_TRAIN_GN_WINO", "0") == "1"]
KEEP_WINO_V = [os.environ.get("GAD_KEEP_WINO_V", "1") != "0"]      # training: the forward's Winograd input image serves the weight gradient


def _scratch(kind, nbytes, device):
    if SCRATCH_ALLOC is not None:
        return SCRATCH_ALLOC(kind, nbytes, device)
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def _out(shape, device):
    if OUT_ALLOC is not None:
        return OUT_ALLOC(tuple(shape), device)
    return torch.empty(tuple(shape), device=device, dtype=torch.float32)


def gemm_raw(A, B, Cout, a_mode, b_mode, M, N, K, lda, ldb, ldc, *, geom: Optional[ConvGeom] = None, alpha=1.0,
             bias=None, rowadd=None, rows_per_group=1, residual=None, ldr=0, batch=1, batch_inner=1,
             sA=(0, 0), sB=(0, 0), sC=(0, 0), tile_hint=0, splitk_hint=0, A2=None, a_split=0, B_bf16=None,
             A_k2=None, B_k2=None, k_split=0, B_wino=None, B_wino4=None, wino_wgrad=False, wino_input=None, force_f32=False,
             keep_v=None, wino_v=None):
    """`keep_v` (a list): a forward convolution that takes an F(4x4) Winograd route appends its scratch - the transformed
    input V at its start - for the weight gradient of the same convolution, which takes it as `wino_v` and skips its own
    input transform (training: Conv2dFn).
    `force_f32`: exact fp32 products whatever the process-wide operand precision (small parameter-gradient products).
    `wino_input(V)`: the caller supplies the F(4x4) Winograd input transform itself (GroupNorm writing V directly,
    `gn_silu_conv3x3_raw`): if the planner puts this launch on an F(4x4) route the callback is run on the route's scratch and
    the convolution starts behind its input stage (-> True); on any other route nothing is launched (-> False)."""
    lib = _capi.load()
    ws = workspace(A.device)
    a = GemmArgs()
    a.A, a.B, a.C = A.data_ptr(), B.data_ptr(), Cout.data_ptr()
    a.a_mode, a.b_mode = a_mode, b_mode
    a.M, a.N, a.K = M, N, K
    a.lda, a.ldb, a.ldc = lda, ldb, ldc
    a.batch, a.batch_inner = batch, batch_inner
    a.strideA0, a.strideA1 = sA
    a.strideB0, a.strideB1 = sB
    a.strideC0, a.strideC1 = sC
    if geom is not None:
        a.g = geom
    a.alpha = alpha
    a.bias, a.rowadd, a.residual = _ptr(bias), _ptr(rowadd), _ptr(residual)
    a.rows_per_group = rows_per_group
    a.ld_rowadd = (rowadd.stride(0) if rowadd.ndim == 2 else rowadd.shape[-1]) if rowadd is not None else 0      # (a column block of a wider matrix: nn.UNet2DModel._temb_rows)
    a.ldr = ldr
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    a.tile_hint, a.splitk_hint = tile_hint, splitk_hint
    a.operand_precision = 1 if (OPERAND_PRECISION[0] and not force_f32) else 0
    a.flags = KERNEL_FLAGS["gemm"]
    if A2 is not None:
        a.A2, a.a_split, a.ldx2 = A2.data_ptr(), a_split, A2.shape[-1]
    if B_bf16 is not None:
        a.B_bf16 = B_bf16.data_ptr()
    if A_k2 is not None:            # K-concatenated operands (fused LoRA): k >= k_split reads A_k2 / B_k2
        a.A_k2, a.B_k2, a.k_split = A_k2.data_ptr(), B_k2.data_ptr(), k_split
        a.lda_k2, a.ldb_k2 = A_k2.shape[-1], B_k2.shape[-1]
    if B_wino is not None or B_wino4 is not None:      # Winograd routes: the planner says whether this launch takes one and how much scratch it needs
        a.B_wino, a.B_wino4 = _ptr(B_wino), _ptr(B_wino4)
        need = lib.gad_gemm_wino_bytes(C.byref(a))
        if need:
            V = _scratch("wino", need, A.device)      # stream-ordered: safe to drop after the launch
            a.wino_ws, a.wino_ws_bytes = V.data_ptr(), need
            if keep_v is not None and a.B_wino4 and lib.gad_gemm_kernel_id(C.byref(a)) == 6:
                keep_v.append(V)
        else:
            a.B_wino = a.B_wino4 = None
    if wino_wgrad:                   # 3x3 weight gradient: the F(4x4) Winograd form where the planner models it faster
        a.flags |= _capi.GEMM_WINO_WGRAD
        need = lib.gad_gemm_wino_bytes(C.byref(a))
        if need:
            V = _scratch("wino", need, A.device)
            a.wino_ws, a.wino_ws_bytes = V.data_ptr(), need
            if wino_v is not None and wino_v.numel() >= 36 * (K // 16) * geom.C * 4:      # the forward launch's V: [36][pixels / 16][Cin] fp32
                a.B_wino4 = wino_v.data_ptr()
                a.flags |= _capi.GEMM_WINO_SKIP_INPUT
        else:
            a.flags &= ~_capi.GEMM_WINO_WGRAD
    if isinstance(B, _ShapeOnly) and not (a.flags & _capi.GEMM_WINO_SKIP_INPUT):
        return False                 # the activation exists only as the kept image and this launch would not read it: the caller re-makes it
    if wino_input is not None:
        if lib.gad_gemm_kernel_id(C.byref(a)) != 6 or not a.wino_ws:
            return False
        wino_input(V)
        a.flags |= _capi.GEMM_WINO_SKIP_INPUT
    if ROUTE_STATS is not None and (a.B_wino4 or a.flags & _capi.GEMM_WINO_WGRAD):      # tools/wino_route_stats.py
        key = (lib.gad_gemm_kernel_id(C.byref(a)), M, N, K, bool(a.flags & _capi.GEMM_WINO_SKIP_INPUT), torch.is_grad_enabled())
        ROUTE_STATS[key] = ROUTE_STATS.get(key, 0) + 1
    if SCRATCH_ALLOC is not None:    # canary runs: the workspace at exactly the size the planner asks for this launch
        need = lib.gad_gemm_workspace_bytes(C.byref(a))
        ws = _scratch("ws", need, A.device) if need else None
        a.ws, a.ws_bytes = (ws.data_ptr() if need else None), need
    if PROFILER is not None:
        PROFILER.gemm(lib, a, batch)
        return True
    check(lib.gad_gemm(C.byref(a), _stream()), "gad_gemm")
    return True


class GemmProfiler:
    """Brackets every contraction launch with HIP events on the launch stream (bench.py's live
    per-kernel timing).  Keyed by the kernel instance the C side picks (modes, tile, vec)."""
    NAMES = {(A_CONV, B_KC): "conv_fwd", (A_CONVT, B_WDGRAD): "conv_dgrad", (A_MC, B_CONV): "conv_wgrad",
             (A_KC, B_KC): "gemm_nt", (A_KC, B_MC): "gemm_nn", (A_MC, B_MC): "gemm_tn"}

    def __init__(self):
        self.records = []          # (key, flops, algorithmic bytes, start_event, end_event)

    def gemm(self, lib, a, batch):
        tile, sk, vec = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib.gad_gemm_plan(C.byref(a), C.byref(tile), C.byref(sk), C.byref(vec)), "gad_gemm_plan")
        kid = lib.gad_gemm_kernel_id(C.byref(a))
        s, e, mid = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), None
        s.record()
        if kid in (5, 6) and not (a.flags & _capi.GEMM_WINO_SKIP_INPUT):
            # a Winograd forward launch as its two stages with an event between them - the same kernels in the same order as
            # the plain call (GAD_GEMM_WINO_ONLY_INPUT / _SKIP_INPUT): input transform (an HBM stream) | products + output transform
            flags = a.flags
            mid = torch.cuda.Event(enable_timing=True)
            a.flags = flags | _capi.GEMM_WINO_ONLY_INPUT
            check(lib.gad_gemm(C.byref(a), _stream()), "gad_gemm")
            mid.record()
            a.flags = flags | _capi.GEMM_WINO_SKIP_INPUT
            check(lib.gad_gemm(C.byref(a), _stream()), "gad_gemm")
            a.flags = flags
        else:
            check(lib.gad_gemm(C.byref(a), _stream()), "gad_gemm")
        e.record()
        name = self.NAMES.get((a.a_mode, a.b_mode), "gemm") + ("", "_bf16", f"_patch_w{a.g.Wo}", f"_patch_bf16_w{a.g.Wo}", f"_fewout_valu_w{a.g.Wo}", "_wino", "_wino4", "_wino4")[kid]   # (the Winograd kernels are one instance for every map width)
        if kid == 3:                         # bf16 patch kernel: fixed 128 x 128 tiles
            tile.value, sk.value = 128, 1
        key = (name, tile.value, sk.value, vec.value)
        # algorithmic bytes: every operand once (gathered tensor, not its im2col expansion) + the output
        g = a.g
        if a.a_mode in (A_CONV, A_CONVT):
            a_elems = (a.M // max(1, g.Ho * g.Wo)) * g.H * g.W * g.C
        else:
            a_elems = a.M * a.K
        b_elems = (a.K // max(1, g.Ho * g.Wo)) * g.H * g.W * g.C if a.b_mode == B_CONV else a.N * a.K
        extra = a.M * a.N if a.residual else 0
        nbytes = 4.0 * max(1, batch) * (a_elems + b_elems + a.M * a.N + extra)
        flops = 2.0 * a.M * a.N * a.K * max(1, batch)
        # the MFMA work actually issued: Winograd F(4x4) multiplies 36 positions per 4x4 tile where the direct form multiplies
        # 16 x 9 (36/144), F(2x2) 16 per 2x2 tile instead of 36; the weight gradient's F(4x4) form likewise
        executed = flops * ({5: 16.0 / 36.0, 6: 0.25, 7: 0.25}.get(kid, 1.0))
        stage = None
        if kid in (5, 6):
            npos, tpx = (16, 4) if kid == 5 else (36, 16)
            v_bytes = 4.0 * npos * (a.M // tpx) * g.C
            # input stage: x in, V out (None: another kernel - GroupNorm - wrote V) | rest: V, U in, y out (+ residual in; the product
            # panels of the three-launch forms go out and back in on top of that: not algorithmic)
            stage = (mid, 4.0 * a_elems + v_bytes if mid is not None else 0.0, v_bytes + 4.0 * (b_elems * npos / 9.0 + a.M * a.N + extra))
        self.records.append((key, flops, nbytes, s, e, executed, stage))

    def hgemm(self, lib, a, tn=False):
        """Bracket a half-precision-path contraction (gad_hgemm / gad_hgemm_tn): bf16 operands and output (fp32 output for parameter gradients)."""
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if tn:
            s.record()
            check(lib.gad_hgemm_tn(C.byref(a), _stream()), "gad_hgemm_tn")
            e.record()
            nbytes = 2.0 * a.K * (a.M + a.N) + 4.0 * a.M * a.N
            flops = 2.0 * a.M * a.N * a.K
            self.records.append((("hgemm_tn_wgrad", 128, 0, 8), flops, nbytes, s, e, flops, None))
            return
        tile, sk = C.c_int32(), C.c_int32()
        check(lib.gad_hgemm_plan(C.byref(a), C.byref(tile), C.byref(sk)), "gad_hgemm_plan")
        s.record()
        check(lib.gad_hgemm(C.byref(a), _stream()), "gad_hgemm")
        e.record()
        if a.conv:
            name = f"hconv{a.KH}x{a.KW}" + ("_s2dgrad" if a.conv == 2 else "_s2" if a.stride == 2 else "_up" if a.upsample else "")
            a_elems = (a.M // max(1, a.Ho * a.Wo)) * a.H * a.W * a.Cin
        else:
            name = "hgemm_wgrad" if a.out_f32 else "hgemm_lora" if a.A2 else "hgemm"
            a_elems = a.M * a.K
        nbytes = 2.0 * (a_elems + a.N * a.K + (a.M * a.N if a.residual else 0)) + (4.0 if a.out_f32 else 2.0) * a.M * a.N
        flops = 2.0 * a.M * a.N * a.K
        self.records.append(((name, {1: 128, 2: 320, 5: 256320, 6: 256320, 7: 320, 8: 128, 9: 128}[tile.value], sk.value, 8), flops, nbytes, s, e, flops, None))

    def attention(self, fn, a, what, elem_bytes=4):
        """Bracket a fused attention launch.  Algorithmic FLOPs: forward 4 B h Tq Tk d (Q K^T and P V), backward
        10 B h Tq Tk d (the five products of the minimal scheme; the recomputing kernels execute seven).  Bytes: q, k,
        v, o once forward; those plus dO, dq, dk, dv backward."""
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        check(fn(C.byref(a), _stream()), f"gad_attention_{what}")
        e.record()
        d_alg = getattr(a, "alg_d", a.d)               # zero-padded heads (ops.PadHeadsFn): count the TRUE head dim's FLOPs
        unit = float(a.B) * a.heads * a.Tq * a.Tk * d_alg
        qb, kb = float(elem_bytes) * a.B * a.Tq * a.heads * d_alg, float(elem_bytes) * a.B * a.Tk * a.heads * d_alg
        flops, nbytes = (4.0 * unit, 2 * qb + 2 * kb) if what == "fwd" else (10.0 * unit, 4 * qb + 4 * kb)
        self.records.append(((f"attn_{what}_d{d_alg}", a.Tq, a.Tk, elem_bytes), flops, nbytes, s, e, flops, None))

    def summary(self):
        """{key: dict(launches, ms, flops, bytes, executed[, ms_input, bytes_input, bytes_rest])} - call after a device
        synchronize.  flops = algorithmic (2 M N K of the direct form), executed = the MFMA work issued; Winograd forward
        launches also carry the duration and the stream bytes of their input-transform stage."""
        out = {}
        for key, fl, nb, s, e, ex, stage in self.records:
            d = out.setdefault(key, dict(launches=0, ms=0.0, flops=0.0, bytes=0.0, executed=0.0))
            d["launches"] += 1
            d["ms"] += s.elapsed_time(e)
            d["flops"] += fl
            d["bytes"] += nb
            d["executed"] += ex
            if stage is not None:
                d["bytes_rest"] = d.get("bytes_rest", 0.0) + stage[2]
                d.setdefault("ms_input", 0.0)
                d.setdefault("bytes_input", 0.0)
                d.setdefault("n_input", 0)
                if stage[0] is not None:                 # this launch ran its own input transform
                    d["ms_input"] += s.elapsed_time(stage[0])
                    d["bytes_input"] += stage[1]
                    d["n_input"] += 1
        return out


PROFILER = None


def _conv_out_size(h, k, stride, pad_lo, pad_hi):
    return (h + pad_lo + pad_hi - k) // stride + 1


def conv2d_fwd_raw(x, w, bias, stride=1, pad=(1, 1, 1, 1), upsample=False, rowadd=None, residual=None,
                   tile_hint=0, splitk_hint=0, x2=None, wino_input=None, keep_v=None):
    """x [B,H,W,Cin] -> y [B,Ho,Wo,Cout]; pad = (top, bottom, left, right).
    x2 [B,H,W,C2]: the conv input is cat([x, x2], channels) without materialising it (UpBlock2D's skip concat)."""
    _req(x, "conv x")
    Bn, H, W, C1 = x.shape
    Cin = C1
    if x2 is not None:
        _req(x2, "conv x2")
        if x2.shape[:3] != x.shape[:3]:
            raise _capi.GadError(f"conv x2 {tuple(x2.shape)} does not match x {tuple(x.shape)}")
        Cin = C1 + x2.shape[-1]
    wk = weight_krsc(w)
    Cout, KH, KW, wc = wk.shape
    if wc != Cin:
        raise _capi.GadError(f"conv weight expects {wc} input channels, got {Cin}")
    He, We = (2 * H, 2 * W) if upsample else (H, W)
    Ho = _conv_out_size(He, KH, stride, pad[0], pad[1])
    Wo = _conv_out_size(We, KW, stride, pad[2], pad[3])
    y = _out((Bn, Ho, Wo, Cout), x.device)
    g = ConvGeom(H, W, Cin, C1, Ho, Wo, KH, KW, stride, pad[0], pad[2], int(upsample))
    if residual is not None:
        _req(residual, "conv residual")
    # Winograd forms of the weight: F(4x4) wherever the output map is a multiple of 4 (it measured faster than F(2x2) at every
    # shape of tools/ab_winograd.py), F(2x2) for the other even maps; the library's planner still decides per launch
    wino_ok = (KH == 3 and KW == 3 and stride == 1 and x2 is None and tile_hint in (0, 7, 8, 9, 10, 11) and splitk_hint == 0
               and tuple(pad) == (1, 1, 1, 1))
    f4_maps = Ho % 4 == 0 and Wo % 4 == 0 and not KERNEL_FLAGS.get("no_wino4")
    launched = gemm_raw(x, wk, y, A_CONV, B_KC, Bn * Ho * Wo, Cout, KH * KW * Cin, 0, KH * KW * Cin, Cout, geom=g,
             bias=bias, rowadd=rowadd, rows_per_group=Ho * Wo, residual=residual, ldr=Cout,
             tile_hint=tile_hint, splitk_hint=splitk_hint, A2=x2, a_split=C1, wino_input=wino_input, keep_v=keep_v,
             B_wino=wino_weight(w, 2) if (wino_ok and (tile_hint == 7 or (tile_hint == 0 and not f4_maps))) else None,
             B_wino4=wino_weight(w, 4) if (wino_ok and tile_hint != 7 and f4_maps) else None,
             B_bf16=bf16_weight(w) if (OPERAND_PRECISION[0] >= 1 and KH == 3 and Cin % 32 == 0 and x2 is None
                                       and not torch.cuda.is_current_stream_capturing()) else None)
    return y if launched is not False else None


def gn_silu_conv3x3_raw(x, x2, gamma, beta, G, eps, w, bias, rowadd=None, residual=None, tile_hint=0):
    """conv3x3(SiLU(GroupNorm(cat([x, x2])))) + bias + rowadd + residual, forward only (the sampler's ResnetBlock2D halves:
    reference diffusers ResnetBlock2D.forward, norm1 -> silu -> conv1 (+ temb) and norm2 -> silu -> conv2 (+ shortcut)).
    Where the convolution takes an F(4x4) Winograd route and the norm has a plan for it (`gad_groupnorm_wino4_ok`), GroupNorm
    writes the route's transformed input V directly - the normalised activation never exists in HBM and the route's input
    transform launch disappears; otherwise the two ordinary launches run.  Same results to fp32 rounding either way."""
    _req(x, "groupnorm x")
    Bn, H, W, C1 = x.shape
    Cin = C1 + (x2.shape[-1] if x2 is not None else 0)
    lib = _capi.load()

    def plain():
        h = group_norm_cat_raw(x, x2, gamma, beta, G, eps, True) if x2 is not None else group_norm(x, gamma, beta, G, eps, True)
        return conv2d_fwd_raw(h, w, bias, 1, (1, 1, 1, 1), False, rowadd=rowadd, residual=residual, tile_hint=tile_hint)
    if (torch.is_grad_enabled() or tuple(w.shape[2:]) != (3, 3) or w.shape[1] != Cin or OPERAND_PRECISION[0] != 0 or KERNEL_FLAGS["gn"]
            or KERNEL_FLAGS.get("no_gn_wino") or H % 4 or W % 4):
        return plain()
    mean = torch.empty((Bn, G), device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    a = GroupNormArgs()
    a.x, a.gamma, a.beta, a.mean, a.rstd = x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr()
    a.B, a.HW, a.C, a.G, a.eps, a.silu = Bn, H * W, Cin, G, eps, 1
    if x2 is not None:
        _req(x2, "groupnorm x2")
        a.x2, a.C1 = x2.data_ptr(), C1
    if not lib.gad_groupnorm_wino4_ok(C.byref(a), W):
        return plain()

    def fill(V):
        check(lib.gad_groupnorm_silu_wino4(C.byref(a), V.data_ptr(), W, _stream()), "gad_groupnorm_silu_wino4")
    # the convolution's "input" is only a shape here: its first stage is replaced by `fill`
    y = conv2d_fwd_raw(_ShapeOnly(x, (Bn, H, W, Cin)), w, bias, 1, (1, 1, 1, 1), False, rowadd=rowadd, residual=residual,
                       tile_hint=tile_hint, wino_input=fill)
    return y if y is not None else plain()


class _ShapeOnly:
    """Stands in for the input tensor of a convolution whose input stage another kernel has taken over: shape, device and a
    valid aligned pointer (never dereferenced)."""

    def __init__(self, t, shape):
        self._t, self.shape, self.device, self.dtype = t, tuple(shape), t.device, t.dtype
        self.is_cuda = t.is_cuda

    def data_ptr(self):
        return self._t.data_ptr()

    def is_contiguous(self):
        return True


def bf16_weight(w):
    """bf16 (RNE) copy of a conv weight in its [Cout][KH][KW][Cin] storage: the bf16-mode patch convolution streams it by
    LDS-DMA instead of rounding the fp32 weights in every workgroup.  A parameter that lives in a flat buffer
    (training.flatten_params) is served from ONE bf16 shadow of the whole buffer, refreshed by a single cast when the
    weights changed (torch version counter, or the epoch the raw optimizer kernels bump) - so training pays one cast per
    step, not one per layer; any other parameter caches its own copy."""
    home = getattr(w, "_gad_flat", None)
    if home is not None and w.data_ptr() == home[0].data_ptr() + 4 * home[1]:
        flat, off, n = home
        key = (flat._version, WEIGHT_EPOCH[0], getattr(flat, "_gad_epoch", 0))
        cached = getattr(flat, "_gad_bf16", None)
        if cached is None or cached[0] != key:
            shadow = cached[1] if cached is not None else torch.empty(flat.numel(), device=flat.device, dtype=torch.bfloat16)
            shadow.copy_(flat.detach())
            # A resident parameter written through torch (load_state_dict, p.copy_(), a torch.optim step) bumps only ITS
            # version counter, not the buffer's: remember every resident's version as of this cast ...
            cached = (key, shadow, {o_: p_._version for p_, o_, _ in getattr(flat, "_gad_params", ())})
            flat._gad_bf16 = cached
        if cached[2].get(off, w._version) != w._version:          # ... and re-cast the slice of one that moved since
            cached[1][off:off + n].copy_(flat.detach()[off:off + n])
            cached[2][off] = w._version
        o, i, kh, kw = w.shape
        return cached[1][off:off + n].view(o, kh, kw, i)
    key = weight_key(w)
    cached = getattr(w, "_gad_bf16", None)
    if cached is None or cached[0] != key:
        cached = (key, weight_krsc(w).detach().to(torch.bfloat16).contiguous())
        w._gad_bf16 = cached
    return cached[1]


def _in_flat_buffer(w) -> bool:
    home = getattr(w, "_gad_flat", None)
    return home is not None and w.data_ptr() == home[0].data_ptr() + 4 * home[1]


def dgrad_as_forward(w, stride, pad) -> bool:
    """The data gradient of a 3x3 / stride-1 / pad-1 convolution is the FORWARD convolution of dy with the 180-degree-
    rotated, channel-transposed weight, and the forward patch kernels (4 x 1 waves, [n][k] weight tiles read as b128, 96 /
    128 / 160-channel tiles - no padding at the pruned widths 96 / 192 / 288; in bf16 mode the LDS-DMA bf16 weight stream)
    are faster than the data-gradient instances (128-column tiles, [k][n] weight tiles read as dwords).  Taken when the
    rotated copy is cheap: a FROZEN weight (the SD base U-Net under LoRA: made once) or one living in a flat parameter
    buffer (training.flatten_params: ONE launch per optimizer step rotates every 3x3 weight of the model,
    `gad_rotate_conv3x3`).  `kernel_flags(native_dgrad=True)` keeps the data-gradient kernels (A/B tools, tests)."""
    return (tuple(w.shape[2:]) == (3, 3) and stride == 1 and tuple(pad) == (1, 1, 1, 1) and w.shape[0] % 32 == 0
            and w.shape[1] >= 64 and (not w.requires_grad or _in_flat_buffer(w)) and not KERNEL_FLAGS.get("native_dgrad"))


def _rot_table(flat):
    """(device table, tiles) of `gad_rotate_conv3x3` for the 3x3 weights resident in `flat` - built once per buffer."""
    cached = getattr(flat, "_gad_rot_table", None)
    if cached is None:
        rows = [(off, p_.shape[0], p_.shape[1], a, b)
                for p_, off, _ in getattr(flat, "_gad_params", ()) if p_.ndim == 4 and tuple(p_.shape[2:]) == (3, 3)
                for a in range(0, p_.shape[0], 32) for b in range(0, p_.shape[1], 32)]
        cached = (torch.tensor(rows, dtype=torch.int64, device=flat.device), len(rows)) if rows else (None, 0)
        flat._gad_rot_table = cached
    return cached


def rotated_weight(w):
    """W'[ci][2-r][2-s][co] = W[co][r][s][ci] as a conv parameter of logical shape [Cin, Cout, 3, 3] (storage
    [Cin][3][3][Cout]).  A weight living in a flat parameter buffer is served from ONE rotated shadow of the buffer,
    refreshed by a single launch when the weights changed (same keys as `bf16_weight`; the returned view carries
    `_gad_flat`, so bf16 mode casts the whole shadow once, too); any other weight caches its own copy per version."""
    if _in_flat_buffer(w):
        flat, off, n = w._gad_flat
        table, tiles = _rot_table(flat)
        key = (flat._version, WEIGHT_EPOCH[0], getattr(flat, "_gad_epoch", 0))
        cached = getattr(flat, "_gad_rot", None)
        stale = cached is None or cached[0] != key
        if not stale and cached[2].get(off, w._version) != w._version:      # written through torch since the last refresh
            stale = True
        if stale and tiles:
            shadow = cached[1] if cached is not None else torch.empty_like(flat)
            shadow._gad_conv3x3 = [(o_, b, a) for o_, a, b in _conv3x3_residents(flat)]      # rotated: [Cin][3][3][Cout]
            check(_capi.load().gad_rotate_conv3x3(flat.data_ptr(), shadow.data_ptr(), table.data_ptr(), tiles, _stream()),
                  "gad_rotate_conv3x3")
            shadow._gad_epoch = getattr(shadow, "_gad_epoch", 0) + 1         # its residents changed (bf16_weight's key)
            cached = (key, shadow, {o_: p_._version for p_, o_, _ in flat._gad_params}, cached[3] if cached is not None else {})
            flat._gad_rot = cached
        view = cached[3].get(off)
        if view is None:
            co, ci = w.shape[0], w.shape[1]
            view = cached[1][off:off + n].view(ci, 3, 3, co).permute(0, 3, 1, 2)
            view._gad_flat = (cached[1], off, n)
            cached[3][off] = view
        return view
    key = weight_key(w)
    cached = getattr(w, "_gad_rot", None)
    if cached is None or cached[0] != key:
        store = weight_krsc(w).detach().flip(1, 2).permute(3, 1, 2, 0).contiguous()      # [Cin][KH][KW][Cout]
        cached = (key, store.permute(0, 3, 1, 2))
        w._gad_rot = cached
    return cached[1]


def _conv3x3_residents(flat):
    """[(offset, Cout, Cin)] of the 3x3 conv weights stored in a flat parameter buffer ([Cout][3][3][Cin] each); a rotated
    shadow (`rotated_weight`) carries the list of its source with the channel roles swapped."""
    r = getattr(flat, "_gad_conv3x3", None)
    if r is None:
        r = [(off, p_.shape[0], p_.shape[1]) for p_, off, _ in getattr(flat, "_gad_params", ())
             if p_.ndim == 4 and tuple(p_.shape[2:]) == (3, 3)]
        flat._gad_conv3x3 = r
    return r


def _wino_ok(co, ci):
    return ci % 32 == 0 and co % 4 == 0 and co >= 64


def _wino_transform(src, items, npos):
    """items [(src offset, Cout, Cin)] -> (U buffer, {src offset: (dst offset, floats)}, table, tiles); npos = 16 or 36"""
    rows, where, doff = [], {}, 0
    for off, co, ci in items:
        where[off] = (doff, npos * co * ci)
        rows += [(off, doff, co, ci, a, b) for a in range(0, co, 32) for b in range(0, ci, 32)]
        doff += npos * co * ci
    U = torch.empty(max(doff, 1), device=src.device, dtype=torch.float32)
    table = torch.tensor(rows, dtype=torch.int64, device=src.device)
    return U, where, table, len(rows)


def wino_weight(w, f=2):
    """The Winograd form U = G w G^T of a 3x3 conv weight: [16][Cout][Cin] for F(2x2, 3x3) (`gad_wino_weights`, f = 2) or
    [36][Cout][Cin] for F(4x4, 3x3) (`gad_wino4_weights`, f = 4); None for a weight the Winograd kernels do not take
    (Cin % 32, Cout % 4, < 64 output channels, bf16-operand mode, the no_wino switch).  A weight living in a flat parameter
    buffer - or in the rotated shadow of one, for the data gradients - is served from ONE transformed shadow of that buffer
    per form, refreshed by a single launch when the weights changed (the keys of `bf16_weight`); any other weight caches its
    own transform per version."""
    co, ci = w.shape[0], w.shape[1]
    if (tuple(w.shape[2:]) != (3, 3) or not _wino_ok(co, ci) or OPERAND_PRECISION[0] >= 1
            or KERNEL_FLAGS["gemm"] & (_capi.GEMM_NO_WINO | _capi.GEMM_NO_PATCH | _capi.GEMM_SCALAR_EPILOGUE | _capi.GEMM_TAP_MAJOR_K)
            or (f == 4 and KERNEL_FLAGS.get("no_wino4"))):
        return None
    lib = _capi.load()
    launch, npos, attr = (lib.gad_wino_weights, 16, "_gad_wino") if f == 2 else (lib.gad_wino4_weights, 36, "_gad_wino4")
    if _in_flat_buffer(w):
        flat, off, n = w._gad_flat
        key = (flat._version, WEIGHT_EPOCH[0], getattr(flat, "_gad_epoch", 0))
        cached = getattr(flat, attr, None)
        if cached is None:
            items = [(o_, a, b) for o_, a, b in _conv3x3_residents(flat) if _wino_ok(a, b)]
            cached = [None, *_wino_transform(flat, items, npos), {}]
            setattr(flat, attr, cached)
        _, U, where, table, tiles, versions = cached
        if off not in where:
            return None
        stale = cached[0] != key or versions.get(off, w._version) != w._version
        if stale:
            check(launch(flat.data_ptr(), U.data_ptr(), table.data_ptr(), tiles, _stream()), "gad_wino_weights")
            cached[0] = key
            cached[5] = {o_: p_._version for p_, o_, _ in getattr(flat, "_gad_params", ())}
        d0, dn = where[off]
        return U[d0:d0 + dn]
    key = weight_key(w)
    cached = getattr(w, attr, None)
    if cached is None or cached[0] != key:
        src = weight_krsc(w).detach()
        if cached is None:
            U, _, table, tiles = _wino_transform(src, [(0, co, ci)], npos)
        else:
            U, table, tiles = cached[1], cached[2], cached[3]
        check(launch(src.data_ptr(), U.data_ptr(), table.data_ptr(), tiles, _stream()), "gad_wino_weights")
        cached = (key, U, table, tiles)
        setattr(w, attr, cached)
    return cached[1]


def refresh_wino(module):
    """Bring the Winograd shadows of every 3x3 weight of `module` up to date NOW - before replaying a captured graph, whose
    launches read the shadow at a fixed address but cannot notice that the weights moved since the capture."""
    for p in module.parameters():
        if p.ndim == 4 and tuple(p.shape[2:]) == (3, 3):
            home = p._gad_flat[0] if _in_flat_buffer(p) else p
            for f, attr in ((2, "_gad_wino"), (4, "_gad_wino4")):
                if getattr(home, attr, None) is not None:            # a shadow some launch has used
                    wino_weight(p, f)


def two_source_ok(c1: int, c2: int) -> bool:
    """Can conv2d_fwd_raw(x, ..., x2=) gather this channel split? (32-channel K steps must not straddle it)"""
    return c1 % 32 == 0 and c2 % 32 == 0


def conv2d_dgrad_raw(dy, w, x_shape, stride=1, pad=(1, 1, 1, 1), upsample=False, tile_hint=0, splitk_hint=0):
    """dy [B,Ho,Wo,Cout] -> dx [B,H,W,Cin] (sums the 2x2 replicas if the conv was upsample-fused)."""
    _req(dy, "conv dy")
    Bn, H, W, Cin = x_shape
    wk = weight_krsc(w)
    Cout, KH, KW, _ = wk.shape
    _, Ho, Wo, _ = dy.shape
    He, We = (2 * H, 2 * W) if upsample else (H, W)
    dxe = _out((Bn, He, We, Cin), dy.device)
    if (KH == 1 and KW == 1 and stride == 1 and tuple(pad) == (0, 0, 0, 0) and not upsample and Cin % 4 == 0 and Cout % 4 == 0
            and not KERNEL_FLAGS["gemm"] & _capi.GEMM_GENERAL_LOADERS):
        # a 1x1 / stride-1 convolution (ResNet shortcuts, proj_in / proj_out of the SD transformer blocks) is a dense
        # GEMM over the pixel rows: dx = dy W on the lean dense loaders instead of the transposed im2col gather
        gemm_raw(dy, wk, dxe, A_KC, B_MC, Bn * H * W, Cin, Cout, Cout, Cin, Cin, tile_hint=tile_hint, splitk_hint=splitk_hint)
        return dxe
    g = ConvGeom(Ho, Wo, Cout, Cout, He, We, KH, KW, stride, pad[0], pad[2], 0)
    gemm_raw(dy, wk, dxe, A_CONVT, B_WDGRAD, Bn * He * We, Cin, KH * KW * Cout, 0, 0, Cin, geom=g,
             tile_hint=tile_hint, splitk_hint=splitk_hint)
    if not upsample:
        return dxe
    dx = torch.empty((Bn, H, W, Cin), device=dy.device, dtype=torch.float32)
    check(_capi.load().gad_upsample2x_bwd(dxe.data_ptr(), dx.data_ptr(), Bn, H, W, Cin, _stream()), "gad_upsample2x_bwd")
    return dx


def conv2d_wgrad_raw(dy, x, w_like, stride=1, pad=(1, 1, 1, 1), upsample=False, tile_hint=0, splitk_hint=0, out=None, wino_v=None):
    """dW with the parameter's logical shape [Cout,Cin,KH,KW] and channels_last storage; `out` = a
    [Cout,Cin,KH,KW] view with that storage to write into (flat gradient slot); `wino_v` = the scratch the forward launch of
    this convolution kept (`conv2d_fwd_raw(keep_v=)`): the Winograd form then reads the transformed input from it."""
    _req(dy, "conv dy")
    if not isinstance(x, _ShapeOnly):
        _req(x, "conv x")
    Bn, H, W, Cin = x.shape
    Cout, _, KH, KW = w_like.shape
    _, Ho, Wo, _ = dy.shape
    dwk = _out((Cout, KH, KW, Cin), dy.device) if out is None else weight_krsc(out)
    if (KH == 1 and KW == 1 and stride == 1 and tuple(pad) == (0, 0, 0, 0) and not upsample and Cin % 4 == 0 and Cout % 4 == 0
            and not KERNEL_FLAGS["gemm"] & _capi.GEMM_GENERAL_LOADERS):
        gemm_raw(dy, x, dwk, A_MC, B_MC, Cout, Cin, Bn * H * W, Cout, Cin, Cin, tile_hint=tile_hint, splitk_hint=splitk_hint)   # dW = dy^T x
        return dwk.permute(0, 3, 1, 2)
    g = ConvGeom(H, W, Cin, Cin, Ho, Wo, KH, KW, stride, pad[0], pad[2], int(upsample))
    wino = (KH == 3 and KW == 3 and stride == 1 and tuple(pad) == (1, 1, 1, 1) and tile_hint in (0, 8) and splitk_hint == 0
            and Ho % 4 == 0 and Wo % 4 == 0 and OPERAND_PRECISION[0] == 0 and not KERNEL_FLAGS.get("no_wino4")
            and not KERNEL_FLAGS["gemm"] & (_capi.GEMM_NO_WINO | _capi.GEMM_NO_PATCH | _capi.GEMM_SCALAR_EPILOGUE | _capi.GEMM_TAP_MAJOR_K
                                            | _capi.GEMM_GENERAL_LOADERS))
    if gemm_raw(dy, x, dwk, A_MC, B_CONV, Cout, KH * KW * Cin, Bn * Ho * Wo, Cout, 0, KH * KW * Cin, geom=g,
                tile_hint=tile_hint, splitk_hint=splitk_hint, wino_wgrad=wino, wino_v=wino_v) is False:
        return None                  # (x given as a shape and no Winograd launch on the kept image: nothing ran)
    return dwk.permute(0, 3, 1, 2)


def colsum_raw(dy2d: torch.Tensor, segments=1, out=None):
    """[S*M, N] -> [S, N] column sums per segment (`out`: contiguous destination of S*N floats)."""
    lib = _capi.load()
    ws = workspace(dy2d.device)
    rows, N = dy2d.shape
    if out is None:
        out = torch.empty((segments, N), device=dy2d.device, dtype=torch.float32)
    check(lib.gad_colsum(dy2d.data_ptr(), out.data_ptr(), segments, rows // segments, N, ws.data_ptr(), ws.numel(),
                         _stream()), "gad_colsum")
    return out


def linear_fwd_raw(x2d, w, bias=None, residual=None, alpha=1.0):
    M, K = x2d.shape
    N = w.shape[0]
    y = _out((M, N), x2d.device)
    gemm_raw(x2d, w, y, A_KC, B_KC, M, N, K, K, K, N, bias=bias, residual=residual, ldr=N, alpha=alpha)
    return y


def linear_dgrad_raw(dy2d, w):
    M, N = dy2d.shape
    K = w.shape[1]
    dx = _out((M, K), dy2d.device)
    gemm_raw(dy2d, w, dx, A_KC, B_MC, M, K, N, N, K, K)
    return dx


def linear_wgrad_raw(dy2d, x2d, out=None):
    M, N = dy2d.shape
    K = x2d.shape[1]
    dw = _out((N, K), dy2d.device) if out is None else out
    gemm_raw(dy2d, x2d, dw, A_MC, B_MC, N, K, M, N, K, K)
    return dw


# ----------------------------------------------------------------------------------
# autograd functions
# ----------------------------------------------------------------------------------
def _param_grad(param, compute):
    """compute(out) produces the gradient, writing into `out` when given.  With a sink: first gradient of the
    step goes straight into the slot, later ones are added."""
    v, first = _sink(param)
    if v is None:
        return compute(None)
    if first:
        compute(v)
    else:
        v.add_(compute(None))
    return None


class Conv2dFn(torch.autograd.Function):
    """y = conv(x) + bias + rowadd[b] (time-embedding add) + residual, all fused in the
    contraction epilogue (ResnetBlock2D / Downsample2D / Upsample2D)."""

    @staticmethod
    def forward(ctx, x, w, bias, rowadd, residual, stride, pad, upsample):
        ctx.save_for_backward(x, w)
        ctx.bias_ref = bias
        ctx.cfg = (stride, pad, upsample, bias is not None, rowadd is not None, residual is not None)
        # an F(4x4) forward launch leaves the transformed input V in its scratch: kept for the weight gradient, which would make
        # the same image from x again (KEEP_WINO_V off: the scratch is dropped after the launch as before)
        keep = [] if (KEEP_WINO_V[0] and ctx.needs_input_grad[1]) else None
        y = conv2d_fwd_raw(x, w, bias, stride, pad, upsample, rowadd, residual, keep_v=keep)
        ctx.wino_v = keep[0] if keep else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, upsample, has_b, has_r, has_res = ctx.cfg
        dy = dy.contiguous()
        Bn, Ho, Wo, Cout = dy.shape
        dx = None
        if ctx.needs_input_grad[0]:
            if dgrad_as_forward(w, stride, pad):
                dx = conv2d_fwd_raw(dy, rotated_weight(w), None)
                if upsample:                         # the conv ran on the nearest-2x grid: sum the 2 x 2 replicas
                    dxe, dx = dx, torch.empty(x.shape, device=dy.device, dtype=torch.float32)
                    check(_capi.load().gad_upsample2x_bwd(dxe.data_ptr(), dx.data_ptr(), x.shape[0], x.shape[1], x.shape[2],
                                                          x.shape[3], _stream()), "gad_upsample2x_bwd")
            else:
                dx = conv2d_dgrad_raw(dy, w, x.shape, stride, pad, upsample)
        dw = _param_grad(w, lambda o: conv2d_wgrad_raw(dy, x, w, stride, pad, upsample, out=o, wino_v=ctx.wino_v)) \
            if ctx.needs_input_grad[1] else None
        ctx.wino_v = None
        db, dr = _conv_epilogue_grads(dy, ctx.bias_ref, has_b and ctx.needs_input_grad[2], has_r and ctx.needs_input_grad[3])
        dres = dy if (has_res and ctx.needs_input_grad[4]) else None
        return dx, dw, db, dr, dres, None, None, None


def _conv_epilogue_grads(dy, bias, need_bias, need_rowadd):
    """-> (d bias, d rowadd) of y = conv(x) + bias + rowadd[b]: column sums of dy, per image for the time-embedding row."""
    Bn, Ho, Wo, Cout = dy.shape
    db = dr = None
    if need_rowadd:
        dr = colsum_raw(dy.view(Bn * Ho * Wo, Cout), segments=Bn)
        if need_bias:
            db = _param_grad(bias, lambda o: colsum_raw(dr, 1, out=o).view(Cout))
    elif need_bias:
        db = _param_grad(bias, lambda o: colsum_raw(dy.view(Bn * Ho * Wo, Cout), 1, out=o).view(Cout))
    return db, dr


class GnSiluConv3x3Fn(torch.autograd.Function):
    """Training form of a ResnetBlock2D half: (y [, alias of x]) = (conv3x3(SiLU(GroupNorm(x))) + bias + rowadd[b] + residual [, x]).
    Where the convolution takes an F(4x4) Winograd route and the norm has a plan for it, GroupNorm writes the route's
    transformed input V (`gad_groupnorm_silu_wino4`, as in sampling: `gn_silu_conv3x3_raw`) and V is ALL that is kept of the
    normalised activation: the weight gradient reads it (`GAD_GEMM_WINO_SKIP_INPUT`), the data gradient and the norm's backward
    never needed it - one pass less over the activation forward and backward, and no input-transform launch on either side.
    On any other route: the two ordinary launches, the activation kept.  Backward = Conv2dFn's then GroupNormSiluFn's kernels."""

    @staticmethod
    def forward(ctx, x, gamma, beta, w, bias, rowadd, residual, G, eps, bypass):
        _req(x, "groupnorm x")
        Bn, H, W, Cin = x.shape
        lib = _capi.load()
        mean = torch.empty((Bn, G), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        keep = [] if ctx.needs_input_grad[3] else None
        y = h = None
        if not (OPERAND_PRECISION[0] != 0 or KERNEL_FLAGS["gn"] or KERNEL_FLAGS.get("no_gn_wino") or H % 4 or W % 4 or not KEEP_WINO_V[0]
                or keep is None or not TRAIN_GN_WINO[0]):
            a = GroupNormArgs()
            a.x, a.gamma, a.beta, a.mean, a.rstd = x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mean.data_ptr(), rstd.data_ptr()
            a.B, a.HW, a.C, a.G, a.eps, a.silu = Bn, H * W, Cin, G, eps, 1
            if lib.gad_groupnorm_wino4_ok(C.byref(a), W):
                def fill(V):
                    check(lib.gad_groupnorm_silu_wino4(C.byref(a), V.data_ptr(), W, _stream()), "gad_groupnorm_silu_wino4")
                y = conv2d_fwd_raw(_ShapeOnly(x, (Bn, H, W, Cin)), w, bias, 1, (1, 1, 1, 1), False, rowadd=rowadd, residual=residual,
                                   wino_input=fill, keep_v=keep)
                if y is None:
                    keep.clear()                         # (planned on another route: nothing was launched)
        if y is None:
            h = torch.empty_like(x)
            check(lib.gad_groupnorm_silu_fwd(C.byref(_gn_args(x, h, gamma, beta, mean, rstd, G, eps, True)), _stream()), "gad_groupnorm_silu_fwd")
            y = conv2d_fwd_raw(h, w, bias, 1, (1, 1, 1, 1), False, rowadd=rowadd, residual=residual, keep_v=keep)
        ctx.save_for_backward(x, gamma, beta, mean, rstd, w, h if keep is not None else None)     # (a frozen weight: nothing reads h again)
        ctx.wino_v = keep[0] if keep else None
        ctx.bias_ref = bias
        ctx.cfg = (G, eps, bias is not None, rowadd is not None, residual is not None)
        return (y, x.view_as(x)) if bypass else y

    @staticmethod
    def backward(ctx, dy, dbypass=None):
        x, gamma, beta, mean, rstd, w, h = ctx.saved_tensors
        G, eps, has_b, has_r, has_res = ctx.cfg
        need = ctx.needs_input_grad
        if dy is None:                                   # only the alias of x was used downstream
            return (dbypass, *([None] * 9))
        dy = dy.contiguous()
        V, ctx.wino_v = ctx.wino_v, None
        dh = None
        if need[0] or need[1] or need[2]:
            dh = conv2d_fwd_raw(dy, rotated_weight(w), None) if dgrad_as_forward(w, 1, (1, 1, 1, 1)) \
                else conv2d_dgrad_raw(dy, w, x.shape, 1, (1, 1, 1, 1), False)
        dw = None
        if need[3]:
            def wgrad(o):
                r = None
                if h is None:                            # the activation exists only as V: the Winograd form on it
                    r = conv2d_wgrad_raw(dy, _ShapeOnly(x, x.shape), w, 1, (1, 1, 1, 1), False, tile_hint=8, out=o, wino_v=V)
                if r is None:
                    hh = h
                    if hh is None:                       # (the library declined: make the activation again)
                        hh = torch.empty_like(x)
                        check(_capi.load().gad_groupnorm_silu_fwd(C.byref(_gn_args(x, hh, gamma, beta, mean.clone(), rstd.clone(), G, eps, True)),
                                                                  _stream()), "gad_groupnorm_silu_fwd")
                    r = conv2d_wgrad_raw(dy, hh, w, 1, (1, 1, 1, 1), False, out=o, wino_v=V)
                return r
            dw = _param_grad(w, wgrad)
        db, dr = _conv_epilogue_grads(dy, ctx.bias_ref, has_b and need[4], has_r and need[5])
        dres = dy if (has_res and need[6]) else None
        dx = dgamma = dbeta = None
        if dh is not None:
            dx, dgamma, dbeta = _gn_backward(x, gamma, beta, mean, rstd, G, eps, True, dh, dbypass, need[1] or need[2])
        elif dbypass is not None:
            dx = dbypass
        return dx, dgamma, dbeta, dw, db, dr, dres, None, None, None


def gn_silu_conv3x3(x, gamma, beta, G, eps, w, bias, rowadd=None, residual=None, bypass=False):
    """Differentiable conv3x3(SiLU(GroupNorm(x))) + bias + rowadd + residual (GnSiluConv3x3Fn); with `bypass` also an alias of
    x for the block's residual branch (as `group_norm_bypass`).  Half-precision activations and odd shapes: the two separate ops."""
    if x.dtype == torch.bfloat16 or tuple(w.shape[2:]) != (3, 3) or w.shape[1] != x.shape[-1] or x.ndim != 4:
        if bypass:
            h, x = group_norm_bypass(x, gamma, beta, G, eps, True)
        else:
            h = group_norm(x, gamma, beta, G, eps, True)
        y = conv2d(h, w, bias, rowadd, residual)
        return (y, x) if bypass else y
    if bypass and not (torch.is_grad_enabled() and x.requires_grad):
        return GnSiluConv3x3Fn.apply(x, gamma, beta, w, bias, rowadd, residual, G, eps, False), x
    return GnSiluConv3x3Fn.apply(x, gamma, beta, w, bias, rowadd, residual, G, eps, bypass)


def _half():
    from . import half
    return half


def conv2d(x, w, bias=None, rowadd=None, residual=None, stride=1, pad=(1, 1, 1, 1), upsample=False):
    if x.dtype == torch.bfloat16:
        return _half().conv2d(x, w, bias, rowadd, residual, stride, pad, upsample)
    return Conv2dFn.apply(x, w, bias, rowadd, residual, stride, pad, upsample)


class LinearFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, residual):
        shp = x.shape
        x2 = _req(x, "linear x").view(-1, shp[-1])
        ctx.save_for_backward(x2, w)
        ctx.shp = shp
        ctx.bias_ref = bias
        ctx.flags = (bias is not None, residual is not None)
        r2 = residual.view(-1, w.shape[0]) if residual is not None else None
        return linear_fwd_raw(x2, w, bias, r2).view(*shp[:-1], w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        x2, w = ctx.saved_tensors
        dy2 = dy.contiguous().view(-1, w.shape[0])
        dx = linear_dgrad_raw(dy2, w).view(ctx.shp) if ctx.needs_input_grad[0] else None
        dw = _param_grad(w, lambda o: linear_wgrad_raw(dy2, x2, out=o)) if ctx.needs_input_grad[1] else None
        db = _param_grad(ctx.bias_ref, lambda o: colsum_raw(dy2, 1, out=o).view(-1)) \
            if (ctx.flags[0] and ctx.needs_input_grad[2]) else None
        dres = dy if (ctx.flags[1] and ctx.needs_input_grad[3]) else None
        return dx, dw, db, dres


def linear(x, w, bias=None, residual=None):
    if x.dtype == torch.bfloat16:
        return _half().linear(x, w, bias, residual)
    return LinearFn.apply(x, w, bias, residual)


def lora_fusable(in_features: int, out_features: int, rank: int) -> bool:
    """Can y = x W^T + b + s (x A^T) B^T run as down-GEMM + ONE K-concatenated GEMM (forward and data gradient)?
    The K split points (in_features forward, out_features backward) must be multiples of the 32-wide K step and the rank
    a multiple of 4 (float4 staging); pruned ragged ranks that are not take the two-launch route."""
    return in_features % 32 == 0 and out_features % 32 == 0 and rank % 4 == 0 and rank > 0


class LoraLinearFn(torch.autograd.Function):
    """diffusers LoRACompatibleLinear with a LoRALinearLayer set (text_to_image/train_text_to_image_lora.py:786-820,
    SURVEY A.11): y = x W^T + b + s (x A^T) B^T [+ residual].
    Forward:  mid = s x A^T (alpha epilogue), then ONE launch over the concatenated K axis [x | mid] . [W | B]^T.
    Backward: dmid = s dy B, then ONE launch dx = [dy | dmid] . [W ; A]; dB = dy^T mid, dA = dmid^T x (and dW, db when
    the base is trainable).  Against base GEMM + residual-accumulating side GEMM this drops one launch and one
    read + write pass over y (forward) / dx (backward) per projection - 128 projections in the SD U-Net."""

    @staticmethod
    def forward(ctx, x, w, bias, down, up, s, residual):
        shp = x.shape
        x2 = _req(x, "lora linear x").view(-1, shp[-1])
        M, K = x2.shape
        N, r = up.shape
        mid = torch.empty((M, r), device=x.device, dtype=torch.float32)
        gemm_raw(x2, down, mid, A_KC, B_KC, M, r, K, K, K, r, alpha=s)
        y = torch.empty((M, N), device=x.device, dtype=torch.float32)
        r2 = residual.view(-1, N) if residual is not None else None
        gemm_raw(x2, w, y, A_KC, B_KC, M, N, K + r, K, K, N, bias=bias, residual=r2, ldr=N, A_k2=mid, B_k2=up, k_split=K)
        ctx.save_for_backward(x2, w, down, up, mid)
        ctx.meta = (shp, s, bias, residual is not None)
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, w, down, up, mid = ctx.saved_tensors
        shp, s, bias, has_res = ctx.meta
        M, K = x2.shape
        N, r = up.shape
        dy2 = dy.contiguous().view(-1, N)
        need_x = ctx.needs_input_grad[0]
        need_lora = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        dmid = None
        if need_x or ctx.needs_input_grad[3]:
            dmid = torch.empty((M, r), device=dy2.device, dtype=torch.float32)
            gemm_raw(dy2, up, dmid, A_KC, B_MC, M, r, N, N, r, r, alpha=s)                  # s dy B
        dx = None
        if need_x:
            dx = torch.empty((M, K), device=dy2.device, dtype=torch.float32)
            gemm_raw(dy2, w, dx, A_KC, B_MC, M, K, N + r, N, K, K, A_k2=dmid, B_k2=down, k_split=N)
            dx = dx.view(shp)
        dw = _param_grad(w, lambda o: linear_wgrad_raw(dy2, x2, out=o)) if ctx.needs_input_grad[1] else None
        db = _param_grad(bias, lambda o: colsum_raw(dy2, 1, out=o).view(-1)) if (bias is not None and ctx.needs_input_grad[2]) else None
        dd = _param_grad(down, lambda o: linear_wgrad_raw(dmid, x2, out=o)) if ctx.needs_input_grad[3] else None      # dmid^T x
        du = _param_grad(up, lambda o: linear_wgrad_raw(dy2, mid, out=o)) if ctx.needs_input_grad[4] else None        # dy^T mid
        dres = dy if (has_res and ctx.needs_input_grad[6]) else None
        return dx, dw, db, dd, du, None, dres


def lora_linear(x, w, bias, down, up, s=1.0, residual=None):
    if x.dtype == torch.bfloat16:
        return _half().lora_linear(x, w, bias, down, up, s, residual)
    return LoraLinearFn.apply(x, w, bias, down, up, float(s), residual)


def _gn_args(x, y, gamma, beta, mean, rstd, G, eps, silu):
    a = GroupNormArgs()
    Bn, C_ = x.shape[0], x.shape[-1]
    HW = x.numel() // (Bn * C_)
    ws = workspace(x.device)
    a.x, a.y, a.gamma, a.beta = x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr()
    a.mean, a.rstd = mean.data_ptr(), rstd.data_ptr()
    a.B, a.HW, a.C, a.G = Bn, HW, C_, G
    a.eps, a.silu = eps, int(silu)
    a.ws, a.ws_bytes = ws.data_ptr(), ws.numel()
    a.flags = KERNEL_FLAGS["gn"]
    return a


def _gn_backward(x, gamma, beta, mean, rstd, G, eps, silu, dy, dbypass, affine_grads):
    """-> (dx, dgamma, dbeta) of y = [SiLU](GroupNorm(x)); `dbypass` (the gradient of an alias of x) is summed in the kernel's
    store; parameter gradients go to their flat-buffer slots where they have one (then returned as None)."""
    dy = dy.contiguous()
    dx = torch.empty_like(x)
    a = _gn_args(x, dx, gamma, beta, mean, rstd, G, eps, silu)
    a.dy = dy.data_ptr()
    if dbypass is not None:
        a.dx_add = _req(dbypass.contiguous(), "groupnorm bypass gradient").data_ptr()
    if not affine_grads:                                 # frozen norm (LoRA training): dx only
        check(_capi.load().gad_groupnorm_silu_bwd(C.byref(a), _stream()), "gad_groupnorm_silu_bwd")
        return dx, None, None
    (sg, fg), (sb, fb) = _sink(gamma), _sink(beta)
    direct_g, direct_b = sg is not None and fg, sb is not None and fb
    dgamma = sg if direct_g else torch.empty_like(gamma)
    dbeta = sb if direct_b else torch.empty_like(beta)
    a.dgamma, a.dbeta = dgamma.data_ptr(), dbeta.data_ptr()
    check(_capi.load().gad_groupnorm_silu_bwd(C.byref(a), _stream()), "gad_groupnorm_silu_bwd")
    if sg is not None:
        if not fg:
            sg.add_(dgamma)
        dgamma = None
    if sb is not None:
        if not fb:
            sb.add_(dbeta)
        dbeta = None
    return dx, dgamma, dbeta


class GroupNormSiluFn(torch.autograd.Function):
    """y = [SiLU](GroupNorm(x)) on NHWC; x is [B, ..., C]."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, silu):
        _req(x, "groupnorm x")
        y = torch.empty_like(x)
        mean = torch.empty((x.shape[0], G), device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        a = _gn_args(x, y, gamma, beta, mean, rstd, G, eps, silu)
        check(_capi.load().gad_groupnorm_silu_fwd(C.byref(a), _stream()), "gad_groupnorm_silu_fwd")
        ctx.save_for_backward(x, gamma, beta, mean, rstd)
        ctx.cfg = (G, eps, silu)
        return y

    @staticmethod
    def backward(ctx, dy, dbypass=None):
        x, gamma, beta, mean, rstd = ctx.saved_tensors
        G, eps, silu = ctx.cfg
        return (*_gn_backward(x, gamma, beta, mean, rstd, G, eps, silu, dy, dbypass,
                              ctx.needs_input_grad[1] or ctx.needs_input_grad[2]), None, None, None)


class GroupNormBypassFn(GroupNormSiluFn):
    """(y, x_bypass) = (GroupNorm(+SiLU)(x), x): the input of ResnetBlock2D / the attention block feeds the norm AND the
    residual branch.  Handing the second consumer this node's own alias of x makes x single-consumer for autograd, and the
    backward receives both gradients at once: dx = gn_bwd(dy) + d(bypass) is one kernel (`gad_groupnorm_args.dx_add`)
    instead of the norm's backward plus an elementwise add launch."""

    @staticmethod
    def forward(ctx, x, gamma, beta, G, eps, silu):
        y = GroupNormSiluFn.forward(ctx, x, gamma, beta, G, eps, silu)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dbypass):
        if dy is None:                                   # only the bypass was used downstream
            return dbypass, None, None, None, None, None
        return GroupNormSiluFn.backward(ctx, dy, dbypass)


def group_norm_bypass(x, gamma, beta, G, eps, silu):
    """-> (GroupNorm(+SiLU)(x), alias of x for the residual branch); see GroupNormBypassFn."""
    if x.dtype == torch.bfloat16:
        return _half().group_norm_bypass(x, gamma, beta, G, eps, silu)
    if not (torch.is_grad_enabled() and x.requires_grad):
        return group_norm(x, gamma, beta, G, eps, silu), x
    return GroupNormBypassFn.apply(x, gamma, beta, G, eps, silu)


def group_norm(x, gamma, beta, G, eps, silu):
    if x.dtype == torch.bfloat16:
        return _half().group_norm(x, gamma, beta, G, eps, silu)
    return GroupNormSiluFn.apply(x, gamma, beta, G, eps, silu)


def _gn2_args(x, x2, y, gamma, beta, mean, rstd, G, eps, silu):
    a = _gn_args(y, y, gamma, beta, mean, rstd, G, eps, silu)      # shapes of the concatenated tensor = y's
    a.x, a.x2, a.C1 = x.data_ptr(), x2.data_ptr(), x.shape[-1]
    return a


def group_norm_two_source_ok(x, x2, G) -> bool:
    """Can the forward of GroupNorm(cat([x, x2], channels)) read both sources in place?  (Both plans can - the one-pass
    slab plan and, since round 3, the two-pass plan the pruned widths' 288 = 192 + 96 channels at 32x32 take -, as long as
    a channel quad never straddles the two sources.)"""
    C_ = x.shape[-1] + x2.shape[-1]
    return C_ % G == 0 and x.shape[-1] % 4 == 0 and x2.shape[-1] % 4 == 0


def group_norm_cat_raw(x, x2, gamma, beta, G, eps, silu):
    """[SiLU](GroupNorm(cat([x, x2], -1))) without the concatenated tensor (forward only, no autograd)."""
    _req(x, "groupnorm x")
    _req(x2, "groupnorm x2")
    y = torch.empty((*x.shape[:-1], x.shape[-1] + x2.shape[-1]), device=x.device, dtype=torch.float32)
    mean = torch.empty((x.shape[0], G), device=x.device, dtype=torch.float32)
    rstd = torch.empty_like(mean)
    a = _gn2_args(x, x2, y, gamma, beta, mean, rstd, G, eps, silu)
    check(_capi.load().gad_groupnorm_silu_fwd(C.byref(a), _stream()), "gad_groupnorm_silu_fwd")
    return y


class SiluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _req(x, "silu x")
        y = torch.empty_like(x)
        check(_capi.load().gad_silu_fwd(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "gad_silu_fwd")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        check(_capi.load().gad_silu_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), x.numel(), _stream()), "gad_silu_bwd")
        return dx


def silu(x):
    return SiluFn.apply(x)


class ConcatFn(torch.autograd.Function):
    """torch.cat([h, skip], dim=channel) for NHWC tensors (UpBlock2D skip connections)."""

    @staticmethod
    def forward(ctx, a, b):
        _req(a, "concat a")
        _req(b, "concat b")
        C1, C2 = a.shape[-1], b.shape[-1]
        out = torch.empty((*a.shape[:-1], C1 + C2), device=a.device, dtype=torch.float32)
        check(_capi.load().gad_concat_channels(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel() // C1, C1, C2,
                                               _stream()), "gad_concat_channels")
        ctx.cs = (C1, C2)
        return out

    @staticmethod
    def backward(ctx, d):
        C1, C2 = ctx.cs
        d = d.contiguous()
        da = torch.empty((*d.shape[:-1], C1), device=d.device, dtype=torch.float32)
        db = torch.empty((*d.shape[:-1], C2), device=d.device, dtype=torch.float32)
        check(_capi.load().gad_split_channels(d.data_ptr(), da.data_ptr(), db.data_ptr(), d.numel() // (C1 + C2), C1, C2,
                                              _stream()), "gad_split_channels")
        return da, db


def concat(a, b):
    if a.dtype == torch.bfloat16:
        return _half().concat(a, b)
    return ConcatFn.apply(a, b)


def nchw_to_nhwc_raw(x):
    Bn, C_, H, W = x.shape
    y = torch.empty((Bn, H, W, C_), device=x.device, dtype=torch.float32)
    check(_capi.load().gad_nchw_to_nhwc(x.data_ptr(), y.data_ptr(), Bn, C_, H * W, _stream()), "gad_nchw_to_nhwc")
    return y


def nhwc_to_nchw_raw(x):
    Bn, H, W, C_ = x.shape
    y = torch.empty((Bn, C_, H, W), device=x.device, dtype=torch.float32)
    check(_capi.load().gad_nhwc_to_nchw(x.data_ptr(), y.data_ptr(), Bn, C_, H * W, _stream()), "gad_nhwc_to_nchw")
    return y


class ToNCHWFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return nhwc_to_nchw_raw(_req(x, "to_nchw x"))

    @staticmethod
    def backward(ctx, dy):
        return nchw_to_nhwc_raw(dy.contiguous())


def to_nchw(x):
    return ToNCHWFn.apply(x)


def timestep_embedding(t: torch.Tensor, dim, flip_sin_to_cos, freq_shift, max_period=10000.0):
    t = t.to(torch.int64).contiguous()
    out = torch.empty((t.shape[0], dim), device=t.device, dtype=torch.float32)
    check(_capi.load().gad_timestep_embedding(t.data_ptr(), out.data_ptr(), t.shape[0], dim, int(flip_sin_to_cos),
                                              float(freq_shift), float(max_period), _stream()), "gad_timestep_embedding")
    return out


# ---- attention core: softmax(q k^T / sqrt(d)) v with q,k,v given as [B, T, heads*d] ----
def _attn_gemm(A, B, Cm, a_mode, b_mode, M, N, K, lda, ldb, ldc, Bn, heads, sA, sB, sC, alpha=1.0):
    gemm_raw(A, B, Cm, a_mode, b_mode, M, N, K, lda, ldb, ldc, alpha=alpha, batch=Bn * heads, batch_inner=heads,
             sA=sA, sB=sB, sC=sC)


def fused_attention_ok(d: int, *lds) -> bool:
    """Does the fused (flash-style) attention take this head dim?  Any d <= 256 does; row strides need no alignment
    (an exact instance - 16, 24, 32, 40, 48, 64, 80, 96, 128, 160, 192, 224, 256 - with float4-aligned rows streams by
    LDS-DMA, anything else - the head-grouped-pruned CelebA model's d = 23 - runs the dword-staged RG instance)."""
    return bool(_capi.load().gad_attention_supported(int(d)))


def _attention_args(q, k, v, o, lse, Bn, heads, Tq, Tk, d, ldq, ldk, ldv, sq, sk, sv, scale=None):
    a = _capi.AttentionArgs()
    a.q, a.k, a.v, a.o, a.lse = q.data_ptr(), k.data_ptr(), v.data_ptr(), o.data_ptr(), _ptr(lse)
    a.B, a.heads, a.Tq, a.Tk, a.d = Bn, heads, Tq, Tk, d
    a.ldq, a.ldk, a.ldv, a.ldo = ldq, ldk, ldv, heads * d
    a.stride_q, a.stride_k, a.stride_v, a.stride_o = sq, sk, sv, Tq * heads * d
    a.scale = 1.0 / math.sqrt(d) if scale is None else scale       # (zero-padded heads keep the true head dim's scale)
    if scale is not None:
        a.alg_d = int(round(1.0 / (scale * scale)))
    a.operand_precision = 0
    return a


def attention_fwd_raw(q, k, v, Bn, heads, Tq, Tk, d, ldq, ldk, ldv, need_lse=True, scale=None):
    """Fused attention forward on [Bn, T, *] views (q, k, v may be column blocks of one projection output: their
    data_ptr is the column offset, ld* the row stride).  -> (o [Bn, Tq, heads*d], lse [Bn, heads, Tq] or None)"""
    o = _out((Bn, Tq, heads * d), q.device)
    lse = _out((Bn, heads, Tq), q.device) if need_lse else None
    a = _attention_args(q, k, v, o, lse, Bn, heads, Tq, Tk, d, ldq, ldk, ldv, Tq * ldq, Tk * ldk, Tk * ldv, scale)
    a.operand_precision = 1 if OPERAND_PRECISION[0] else 0         # bf16 modes: bf16-operand instance of the fused kernel
    a.flags = _capi.ATTN_NARROW_FWD if KERNEL_FLAGS.get("narrow_attn_fwd") else 0
    if PROFILER is not None:
        PROFILER.attention(_capi.load().gad_attention_fwd, a, "fwd")
    else:
        check(_capi.load().gad_attention_fwd(C.byref(a), _stream()), "gad_attention_fwd")
    return o, lse


class AttentionCoreFn(torch.autograd.Function):
    """F.scaled_dot_product_attention(q, k, v) of AttnProcessor2_0
    (reference src/diffusers/models/attention_processor.py:1314-1325) on [B, T, heads*d] operands: ONE fused kernel
    forward (online softmax, the scores never leave the CU; only the per-row log-sum-exp is kept) and a
    recomputing dQ + dK/dV kernel pair backward (csrc/attention.hip)."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale=None):
        for t_, n_ in ((q, "q"), (k, "k"), (v, "v")):
            _req(t_, n_)
        Bn, Tq, Cq = q.shape
        Tk = k.shape[1]
        d = Cq // heads
        o, lse = attention_fwd_raw(q, k, v, Bn, heads, Tq, Tk, d, Cq, Cq, Cq, need_lse=True, scale=scale)
        ctx.save_for_backward(q, k, v, o, lse)
        ctx.heads, ctx.prec, ctx.scale = heads, (1 if OPERAND_PRECISION[0] else 0), scale
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, o, lse = ctx.saved_tensors
        heads = ctx.heads
        do = do.contiguous()
        Bn, Tq, Cq = q.shape
        Tk = k.shape[1]
        d = Cq // heads
        dq, dk, dv = _out(q.shape, q.device), _out(k.shape, k.device), _out(v.shape, v.device)
        delta = _out(lse.shape, lse.device)
        a = _attention_args(q, k, v, o, lse, Bn, heads, Tq, Tk, d, Cq, Cq, Cq, Tq * Cq, Tk * Cq, Tk * Cq, ctx.scale)
        a.d_o, a.delta, a.dq, a.dk, a.dv = do.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
        a.ld_do = a.ld_dq = a.ld_dk = a.ld_dv = Cq
        a.stride_do = a.stride_dq = Tq * Cq
        a.stride_dk = a.stride_dv = Tk * Cq
        a.operand_precision = ctx.prec                   # the precision the forward's LSE was computed in
        a.flags = _capi.ATTN_TWO_KERNEL_BWD if KERNEL_FLAGS.get("two_kernel_attn_bwd") else 0
        need = _capi.load().gad_attention_bwd_workspace_bytes(C.byref(a))
        if need:                                         # dQ partial slabs of the single-pass kernel (one per key block)
            slabs = _scratch("attn_ws", need, q.device)
            a.ws, a.ws_bytes = slabs.data_ptr(), need
        if PROFILER is not None:
            PROFILER.attention(_capi.load().gad_attention_bwd, a, "bwd")
        else:
            check(_capi.load().gad_attention_bwd(C.byref(a), _stream()), "gad_attention_bwd")
        return dq, dk, dv, None, None


class UnfusedAttentionCoreFn(torch.autograd.Function):
    """The same function as three launches (batched Q K^T, row softmax, batched P V; P kept for backward): the route
    for head dims without a fused instance, and the independent implementation the fused kernels are tested against."""

    @staticmethod
    def forward(ctx, q, k, v, heads, scale=None):
        for t_, n_ in ((q, "q"), (k, "k"), (v, "v")):
            _req(t_, n_)
        Bn, Tq, Cq = q.shape
        Tk = k.shape[1]
        d = Cq // heads
        scale = 1.0 / math.sqrt(d) if scale is None else scale
        ctx.scale = scale
        S = torch.empty((Bn, heads, Tq, Tk), device=q.device, dtype=torch.float32)
        _attn_gemm(q, k, S, A_KC, B_KC, Tq, Tk, d, Cq, Cq, Tk, Bn, heads,
                   (Tq * Cq, d), (Tk * Cq, d), (heads * Tq * Tk, Tq * Tk))
        check(_capi.load().gad_softmax_fwd(S.data_ptr(), S.data_ptr(), Bn * heads * Tq, Tk, scale, _stream()),
              "gad_softmax_fwd")
        o = torch.empty((Bn, Tq, Cq), device=q.device, dtype=torch.float32)
        _attn_gemm(S, v, o, A_KC, B_MC, Tq, d, Tk, Tk, Cq, Cq, Bn, heads,
                   (heads * Tq * Tk, Tq * Tk), (Tk * Cq, d), (Tq * Cq, d))
        ctx.save_for_backward(q, k, v, S)
        ctx.heads = heads
        return o

    @staticmethod
    def backward(ctx, do):
        q, k, v, P = ctx.saved_tensors
        heads = ctx.heads
        do = do.contiguous()
        Bn, Tq, Cq = q.shape
        Tk = k.shape[1]
        d = Cq // heads
        scale = ctx.scale
        sP = (heads * Tq * Tk, Tq * Tk)
        sQ, sK = (Tq * Cq, d), (Tk * Cq, d)
        # dV[b,h] = P^T dO   (A = P as [k=Tq][m=Tk], B = dO as [k=Tq][n=d])
        dv = torch.empty_like(v)
        _attn_gemm(P, do, dv, A_MC, B_MC, Tk, d, Tq, Tk, Cq, Cq, Bn, heads, sP, sQ, sK)
        # dP = dO V^T
        dP = torch.empty_like(P)
        _attn_gemm(do, v, dP, A_KC, B_KC, Tq, Tk, d, Cq, Cq, Tk, Bn, heads, sQ, sK, sP)
        # dS = scale * P * (dP - rowsum(dP*P))   (in place into dP)
        check(_capi.load().gad_softmax_bwd(P.data_ptr(), dP.data_ptr(), dP.data_ptr(), Bn * heads * Tq, Tk, scale,
                                           _stream()), "gad_softmax_bwd")
        # dQ = dS K ; dK = dS^T Q
        dq = torch.empty_like(q)
        _attn_gemm(dP, k, dq, A_KC, B_MC, Tq, d, Tk, Tk, Cq, Cq, Bn, heads, sP, sK, sQ)
        dk = torch.empty_like(k)
        _attn_gemm(dP, q, dk, A_MC, B_MC, Tk, d, Tq, Tk, Cq, Cq, Bn, heads, sP, sQ, sK)
        return dq, dk, dv, None, None


class PadHeadsFn(torch.autograd.Function):
    """Zero-pad the per-head blocks of an attention projection parameter from head dim d to dpad (a multiple of 8):
    axis 0 - the rows of to_q / to_k / to_v weights [heads*d, C] and biases [heads*d]; axis 1 - the columns of the to_out
    weight [C, heads*d].  The head-grouped-pruned CelebA model has d = 23 (reference unconditional_generation/prune.py:
    337-342): 322-float rows are not float4-aligned, which puts the four projections on the scalar-gather GEMM path and
    the attention on its dword-staged instances.  With zero-padded heads (d = 24) every launch is back on LDS-DMA and the
    result is the same function: the extra q / k channels are 0 * 0, the extra v / output channels meet zero columns of
    to_out.  The gradient of the padded tensor is sliced back into the parameter (its flat-buffer sink when a
    FusedTrainer step is running)."""

    @staticmethod
    def forward(ctx, w, heads, d, dpad, axis):
        ctx.param, ctx.cfg = w, (heads, d, dpad, axis)
        src = w.detach()
        if axis == 0:
            rest = src.shape[1:]
            out = src.new_zeros((heads, dpad) + tuple(rest))
            out[:, :d] = src.reshape((heads, d) + tuple(rest))
            return out.view((heads * dpad,) + tuple(rest))
        out = src.new_zeros((src.shape[0], heads, dpad))
        out[:, :, :d] = src.reshape(src.shape[0], heads, d)
        return out.view(src.shape[0], heads * dpad)

    @staticmethod
    def backward(ctx, g):
        heads, d, dpad, axis = ctx.cfg
        w = ctx.param
        if axis == 0:
            gs = g.view((heads, dpad) + tuple(w.shape[1:]))[:, :d].reshape(w.shape)
        else:
            gs = g.view(w.shape[0], heads, dpad)[:, :, :d].reshape(w.shape)
        return _deliver(w, gs), None, None, None, None


def pad_heads(w, heads, d, dpad, axis):
    return PadHeadsFn.apply(w, heads, d, dpad, axis)


def attention_core(q, k, v, heads, scale=None):
    if q.dtype == torch.bfloat16:
        return _half().attention_core(q, k, v, heads, scale)
    d = q.shape[-1] // heads
    # Training at one wide head over a short sequence (CIFAR: d = 256 / 192, T <= 256) keeps the three-launch route: S is a
    # few MB there, and the recomputing backward (7 products, 1 wave per SIMD at d >= 192) measured 0.48 ms against
    # 0.32 ms (tools/bench_attention.py).  Everything else - every sampling forward, CelebA's and SD's heads - is fused.
    wide_short = torch.is_grad_enabled() and d > 160 and q.shape[1] * k.shape[1] <= 256 * 256
    if fused_attention_ok(d, q.shape[-1]) and not wide_short:
        return AttentionCoreFn.apply(q, k, v, heads, scale)
    return UnfusedAttentionCoreFn.apply(q, k, v, heads, scale)


def attention_core_fused(q, k, v, heads):
    return AttentionCoreFn.apply(q, k, v, heads)


def attention_core_unfused(q, k, v, heads):
    return UnfusedAttentionCoreFn.apply(q, k, v, heads)


def attention_core_qkv_raw(qkv, Bn, T, Cq, heads, scale=None):
    """Inference form on the output of ONE fused projection: qkv is [Bn*T, 3*Cq] (q | k | v along the columns); the
    fused attention kernel reads q, k, v in place through their row stride 3*Cq.  No autograd."""
    d = Cq // heads
    ld = 3 * Cq
    q, k, v = qkv[:, 0:Cq], qkv[:, Cq:2 * Cq], qkv[:, 2 * Cq:3 * Cq]        # views: data_ptr = column offset
    if fused_attention_ok(d, ld):
        return attention_fwd_raw(q, k, v, Bn, heads, T, T, d, ld, ld, ld, need_lse=False, scale=scale)[0]
    scale = 1.0 / math.sqrt(d) if scale is None else scale
    S = torch.empty((Bn, heads, T, T), device=qkv.device, dtype=torch.float32)
    _attn_gemm(q, k, S, A_KC, B_KC, T, T, d, ld, ld, T, Bn, heads, (T * ld, d), (T * ld, d), (heads * T * T, T * T))
    check(_capi.load().gad_softmax_fwd(S.data_ptr(), S.data_ptr(), Bn * heads * T, T, scale, _stream()), "gad_softmax_fwd")
    o = torch.empty((Bn, T, Cq), device=qkv.device, dtype=torch.float32)
    _attn_gemm(S, v, o, A_KC, B_MC, T, d, T, T, ld, Cq, Bn, heads, (heads * T * T, T * T), (T * ld, d), (T * Cq, d))
    return o


# ----------------------------------------------------------------------------------
# scheduler / loss / optimizer kernels (no autograd)
# ----------------------------------------------------------------------------------
def add_noise_raw(x0, eps, t, alphas_cumprod):
    _req(x0, "add_noise x0")
    _req(eps, "add_noise eps")
    t = t.to(torch.int64).contiguous()
    out = torch.empty_like(x0)
    check(_capi.load().gad_add_noise(x0.data_ptr(), eps.data_ptr(), t.data_ptr(), alphas_cumprod.data_ptr(),
                                     out.data_ptr(), x0.shape[0], x0.numel() // x0.shape[0], _stream()), "gad_add_noise")
    return out


def ddim_step_raw(x, eps, alpha_t: float, alpha_prev: float, clip: float, out=None):
    _req(x, "ddim x")
    _req(eps, "ddim eps")
    out = torch.empty_like(x) if out is None else out
    check(_capi.load().gad_ddim_step(x.data_ptr(), eps.data_ptr(), out.data_ptr(), x.numel(), alpha_t, alpha_prev, clip,
                                     _stream()), "gad_ddim_step")
    return out


def to_image01_raw(x):
    _req(x, "to_image01 x")
    y = torch.empty_like(x)
    check(_capi.load().gad_to_image01(x.data_ptr(), y.data_ptr(), x.numel(), _stream()), "gad_to_image01")
    return y


def mse_fwd_bwd_raw(pred, target, grad_scale=1.0):
    """(loss [1], dloss/dpred) for nn.MSELoss(reduction='mean') (reference main.py:602,708)."""
    _req(pred, "mse pred")
    _req(target, "mse target")
    ws = workspace(pred.device)
    loss = torch.empty(1, device=pred.device, dtype=torch.float32)
    d = torch.empty_like(pred)
    check(_capi.load().gad_mse_fwd_bwd(pred.data_ptr(), target.data_ptr(), loss.data_ptr(), d.data_ptr(), pred.numel(),
                                       grad_scale, ws.data_ptr(), ws.numel(), _stream()), "gad_mse_fwd_bwd")
    return loss, d


def sumsq_raw(g, out=None):
    ws = workspace(g.device)
    out = torch.empty(1, device=g.device, dtype=torch.float32) if out is None else out
    check(_capi.load().gad_sumsq(g.data_ptr(), out.data_ptr(), g.numel(), ws.data_ptr(), ws.numel(), _stream()), "gad_sumsq")
    return out


# Raw kernels update parameters through device pointers, which torch's per-tensor version counters do not see.  Anything
# that caches a function of the weights (the fused q|k|v projection of the sampling path) keys on this epoch as well.
WEIGHT_EPOCH = [0]            # rare events that rewrite arbitrary parameters behind torch's back (EMA copy_to / restore ...)


def weight_key(w):
    """Cache key of anything derived from parameter `w`: its address, torch's version counter, the global epoch, and -
    for a parameter living in a flat buffer (training.flatten_params) - that buffer's own epoch, which the raw optimizer
    kernel bumps every step.  Frozen parameters outside the buffer (the SD base U-Net under LoRA) therefore keep their
    derived copies (bf16 casts, rotated convolution weights) across training steps."""
    home = getattr(w, "_gad_flat", None)
    return (w.data_ptr(), w._version, WEIGHT_EPOCH[0], getattr(home[0], "_gad_epoch", 0) if home is not None else 0)


def clip_adam_ema_raw(p, g, m, v, ema, sumsq, *, max_norm, lr, betas, eps, weight_decay, adamw, step, ema_decay):
    p._gad_epoch = getattr(p, "_gad_epoch", 0) + 1        # p is the flat parameter buffer: its residents changed
    a = AdamArgs()
    a.p, a.g, a.m, a.v, a.ema = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), _ptr(ema)
    a.n = p.numel()
    a.sumsq = _ptr(sumsq)
    a.max_norm = max_norm
    a.lr, a.beta1, a.beta2, a.eps, a.weight_decay = lr, betas[0], betas[1], eps, weight_decay
    a.adamw, a.step, a.ema_decay = int(adamw), step, ema_decay
    check(_capi.load().gad_clip_adam_ema(C.byref(a), _stream()), "gad_clip_adam_ema")


def ema_update_raw(ema, p, decay: float):
    check(_capi.load().gad_ema_update(ema.data_ptr(), p.data_ptr(), p.numel(), decay, _stream()), "gad_ema_update")


# ----------------------------------------------------------------------------------
# transformer-block operators (UNet2DConditionModel)
# ----------------------------------------------------------------------------------
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        _req(x, "layernorm x")
        C_ = x.shape[-1]
        rows = x.numel() // C_
        y = torch.empty_like(x)
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty_like(mean)
        check(_capi.load().gad_layernorm_fwd(x.data_ptr(), y.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                             mean.data_ptr(), rstd.data_ptr(), rows, C_, eps, _stream()), "gad_layernorm_fwd")
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.beta_ref = beta
        return y

    @staticmethod
    def backward(ctx, dy, dbypass=None):
        x, gamma, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        C_ = x.shape[-1]
        rows = x.numel() // C_
        dx = torch.empty_like(x)
        want = ctx.needs_input_grad[1] or ctx.needs_input_grad[2]            # frozen LayerNorm (LoRA training): dx only
        dgb = torch.empty(2 * C_, device=x.device, dtype=torch.float32) if want else None
        ws = workspace(x.device)
        add = _req(dbypass.contiguous(), "layernorm bypass gradient") if dbypass is not None else None
        check(_capi.load().gad_layernorm_bwd(x.data_ptr(), dy.data_ptr(), dx.data_ptr(), _ptr(add), gamma.data_ptr(),
                                             mean.data_ptr(), rstd.data_ptr(), _ptr(dgb), rows, C_, ws.data_ptr(), ws.numel(),
                                             _stream()), "gad_layernorm_bwd")
        if not want:
            return dx, None, None, None
        return dx, _deliver(gamma, dgb[:C_]), _deliver(ctx.beta_ref, dgb[C_:]), None


class LayerNormBypassFn(LayerNormFn):
    """(LayerNorm(x), alias of x): the residual branch of a transformer sub-block takes the alias, so the norm's backward
    receives both gradients and adds them in its own store (see GroupNormBypassFn)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        return LayerNormFn.forward(ctx, x, gamma, beta, eps), x.view_as(x)

    @staticmethod
    def backward(ctx, dy, dbypass):
        if dy is None:
            return dbypass, None, None, None
        return LayerNormFn.backward(ctx, dy, dbypass)


def layer_norm(x, gamma, beta, eps=1e-5):
    if x.dtype == torch.bfloat16:
        return _half().layer_norm(x, gamma, beta, eps)
    return LayerNormFn.apply(x, gamma, beta, eps)


def layer_norm_bypass(x, gamma, beta, eps=1e-5):
    if x.dtype == torch.bfloat16:
        return _half().layer_norm_bypass(x, gamma, beta, eps)
    if not (torch.is_grad_enabled() and x.requires_grad):
        return layer_norm(x, gamma, beta, eps), x
    return LayerNormBypassFn.apply(x, gamma, beta, eps)


class GegluFn(torch.autograd.Function):
    """out = h[..., :F] * gelu(h[..., F:])  (diffusers GEGLU after its projection)."""

    @staticmethod
    def forward(ctx, h):
        _req(h, "geglu h")
        F2 = h.shape[-1]
        out = torch.empty((*h.shape[:-1], F2 // 2), device=h.device, dtype=torch.float32)
        check(_capi.load().gad_geglu_fwd(h.data_ptr(), out.data_ptr(), h.numel() // F2, F2 // 2, _stream()), "gad_geglu_fwd")
        ctx.save_for_backward(h)
        return out

    @staticmethod
    def backward(ctx, dout):
        (h,) = ctx.saved_tensors
        dout = dout.contiguous()
        F2 = h.shape[-1]
        dh = torch.empty_like(h)
        check(_capi.load().gad_geglu_bwd(h.data_ptr(), dout.data_ptr(), dh.data_ptr(), h.numel() // F2, F2 // 2, _stream()),
              "gad_geglu_bwd")
        return dh


def geglu(h):
    if h.dtype == torch.bfloat16:
        return _half().geglu(h)
    return GegluFn.apply(h)


def cfg_ddim_step_raw(x, eps_uc, guidance: float, alpha_t: float, alpha_prev: float, clip: float, out=None):
    """x [n...], eps_uc = U-Net output of the doubled batch [uncond ; cond] -> guided DDIM update of x."""
    _req(x, "cfg x")
    _req(eps_uc, "cfg eps")
    if eps_uc.numel() != 2 * x.numel():
        raise _capi.GadError("cfg_ddim_step: eps must hold the doubled batch")
    out = torch.empty_like(x) if out is None else out
    check(_capi.load().gad_cfg_ddim_step(x.data_ptr(), eps_uc.data_ptr(), out.data_ptr(), x.numel(), guidance, alpha_t,
                                         alpha_prev, clip, _stream()), "gad_cfg_ddim_step")
    return out

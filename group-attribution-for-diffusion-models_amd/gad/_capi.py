"""ctypes binding of libgad_hip.so (C ABI declared in include/gad.h).

The product path has NO fallback: if the shared library is missing or a call fails the
caller gets an exception.  Loading the library needs no GPU (used by the CPU test that
checks every declared symbol is exported)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgad_hip.so")


class GadError(RuntimeError):
    pass


class ConvGeom(C.Structure):
    _fields_ = [("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("ldx", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32),
                ("stride", C.c_int32), ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("upsample", C.c_int32)]


class GemmArgs(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
                ("a_mode", C.c_int32), ("b_mode", C.c_int32),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("ldb", C.c_int32), ("ldc", C.c_int32),
                ("batch", C.c_int32), ("batch_inner", C.c_int32),
                ("strideA0", C.c_int64), ("strideA1", C.c_int64), ("strideB0", C.c_int64),
                ("strideB1", C.c_int64), ("strideC0", C.c_int64), ("strideC1", C.c_int64),
                ("g", ConvGeom),
                ("alpha", C.c_float), ("bias", C.c_void_p), ("rowadd", C.c_void_p),
                ("rows_per_group", C.c_int32), ("ld_rowadd", C.c_int32),
                ("residual", C.c_void_p), ("ldr", C.c_int32),
                ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
                ("tile_hint", C.c_int32), ("splitk_hint", C.c_int32), ("operand_precision", C.c_int32),
                ("A2", C.c_void_p), ("a_split", C.c_int32), ("ldx2", C.c_int32), ("B_bf16", C.c_void_p),
                ("A_k2", C.c_void_p), ("B_k2", C.c_void_p), ("lda_k2", C.c_int32), ("ldb_k2", C.c_int32),
                ("k_split", C.c_int32), ("flags", C.c_int32),
                ("B_wino", C.c_void_p), ("wino_ws", C.c_void_p), ("wino_ws_bytes", C.c_int64), ("B_wino4", C.c_void_p)]


class GroupNormArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("y", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("mean", C.c_void_p), ("rstd", C.c_void_p), ("dy", C.c_void_p),
                ("dgamma", C.c_void_p), ("dbeta", C.c_void_p),
                ("B", C.c_int32), ("HW", C.c_int32), ("C", C.c_int32), ("G", C.c_int32),
                ("eps", C.c_float), ("silu", C.c_int32), ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
                ("x2", C.c_void_p), ("C1", C.c_int32), ("flags", C.c_int32), ("dx_add", C.c_void_p)]


class AttentionArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("o", C.c_void_p), ("lse", C.c_void_p),
                ("d_o", C.c_void_p), ("delta", C.c_void_p), ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p),
                ("B", C.c_int32), ("heads", C.c_int32), ("Tq", C.c_int32), ("Tk", C.c_int32), ("d", C.c_int32),
                ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32),
                ("ld_do", C.c_int32), ("ld_dq", C.c_int32), ("ld_dk", C.c_int32), ("ld_dv", C.c_int32),
                ("stride_q", C.c_int64), ("stride_k", C.c_int64), ("stride_v", C.c_int64), ("stride_o", C.c_int64),
                ("stride_do", C.c_int64), ("stride_dq", C.c_int64), ("stride_dk", C.c_int64), ("stride_dv", C.c_int64),
                ("scale", C.c_float), ("operand_precision", C.c_int32),
                ("ws", C.c_void_p), ("ws_bytes", C.c_int64), ("flags", C.c_int32)]


ATTN_TWO_KERNEL_BWD, ATTN_NARROW_FWD = 1, 2


class HGemmArgs(C.Structure):
    """gad_hgemm_args (include/gad.h): the half-precision activation path's contraction"""
    _fields_ = [("A", C.c_void_p), ("A2", C.c_void_p), ("B", C.c_void_p), ("B2", C.c_void_p), ("C", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lda", C.c_int32), ("lda2", C.c_int32), ("ldb", C.c_int32), ("ldb2", C.c_int32), ("ldc", C.c_int32),
                ("k_split", C.c_int32), ("conv", C.c_int32),
                ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32),
                ("pad_t", C.c_int32), ("pad_l", C.c_int32), ("upsample", C.c_int32),
                ("alpha", C.c_float), ("bias", C.c_void_p), ("rowadd", C.c_void_p),
                ("rows_per_group", C.c_int32), ("ld_rowadd", C.c_int32),
                ("residual", C.c_void_p), ("ldr", C.c_int32),
                ("out_f32", C.c_int32), ("accumulate", C.c_int32),
                ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
                ("tile_hint", C.c_int32), ("splitk_hint", C.c_int32)]


class AdamArgs(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("ema", C.c_void_p),
                ("n", C.c_int64), ("sumsq", C.c_void_p), ("max_norm", C.c_float),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("weight_decay", C.c_float), ("adamw", C.c_int32), ("step", C.c_int32), ("ema_decay", C.c_float)]


GEMM_NO_PATCH, GEMM_TAP_MAJOR_K, GEMM_SCALAR_EPILOGUE, GEMM_GENERAL_LOADERS, GN_TWO_PASS = 1, 2, 4, 8, 1
GEMM_NO_WINO = 16
GEMM_WINO_WGRAD = 32
GEMM_WINO_ONLY_INPUT, GEMM_WINO_SKIP_INPUT = 64, 128
A_KC, A_MC, A_CONV, A_CONVT = 0, 1, 2, 3
B_KC, B_MC, B_WDGRAD, B_CONV = 0, 1, 2, 3

_vp, _i32, _i64, _f32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes); every symbol include/gad.h declares
SIGNATURES = {
    "gad_version": (C.c_int, []),
    "gad_last_error": (C.c_char_p, []),
    "gad_gemm_workspace_bytes": (_i64, [C.POINTER(GemmArgs)]),
    "gad_gemm": (C.c_int, [C.POINTER(GemmArgs), _vp]),
    "gad_gemm_uses_bf16": (C.c_int, [C.POINTER(GemmArgs)]),
    "gad_gemm_kernel_id": (C.c_int, [C.POINTER(GemmArgs)]),
    "gad_gemm_wino_bytes": (_i64, [C.POINTER(GemmArgs)]),
    "gad_wino_weights": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "gad_wino4_weights": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "gad_groupnorm_one_pass": (C.c_int, [C.POINTER(GroupNormArgs)]),
    "gad_gemm_plan": (C.c_int, [C.POINTER(GemmArgs), C.POINTER(_i32), C.POINTER(_i32), C.POINTER(_i32)]),
    "gad_groupnorm_workspace_bytes": (_i64, [C.POINTER(GroupNormArgs)]),
    "gad_groupnorm_silu_fwd": (C.c_int, [C.POINTER(GroupNormArgs), _vp]),
    "gad_groupnorm_silu_bwd": (C.c_int, [C.POINTER(GroupNormArgs), _vp]),
    "gad_groupnorm_wino4_ok": (C.c_int, [C.POINTER(GroupNormArgs), C.c_int32]),
    "gad_groupnorm_silu_wino4": (C.c_int, [C.POINTER(GroupNormArgs), _vp, C.c_int32, _vp]),
    "gad_attention_supported": (C.c_int, [_i32]),
    "gad_attention_uses_bf16": (C.c_int, [C.POINTER(AttentionArgs), _i32]),
    "gad_attention_bwd_workspace_bytes": (_i64, [C.POINTER(AttentionArgs)]),
    "gad_attention_fwd": (C.c_int, [C.POINTER(AttentionArgs), _vp]),
    "gad_attention_bwd": (C.c_int, [C.POINTER(AttentionArgs), _vp]),
    "gad_softmax_fwd": (C.c_int, [_vp, _vp, _i64, _i32, _f32, _vp]),
    "gad_softmax_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "gad_layernorm_workspace_bytes": (_i64, [_i64, _i32]),
    "gad_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "gad_layernorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i64, _vp]),
    "gad_geglu_fwd": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "gad_geglu_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "gad_timestep_embedding": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _f32, _f32, _vp]),
    "gad_rotate_conv3x3": (C.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "gad_silu_fwd": (C.c_int, [_vp, _vp, _i64, _vp]),
    "gad_silu_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "gad_concat_channels": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "gad_split_channels": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    "gad_nchw_to_nhwc": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "gad_nhwc_to_nchw": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp]),
    "gad_upsample2x_bwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "gad_colsum": (C.c_int, [_vp, _vp, _i32, _i64, _i32, _vp, _i64, _vp]),
    "gad_add_noise": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _vp]),
    "gad_ddim_step": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _vp]),
    "gad_cfg_ddim_step": (C.c_int, [_vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _vp]),
    "gad_to_image01": (C.c_int, [_vp, _vp, _i64, _vp]),
    "gad_mse_fwd_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _vp, _i64, _vp]),
    "gad_sumsq": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp]),
    "gad_clip_adam_ema": (C.c_int, [C.POINTER(AdamArgs), _vp]),
    "gad_ema_update": (C.c_int, [_vp, _vp, _i64, _f32, _vp]),
    # half-precision activation path (csrc/half.hip, csrc/attention.hip)
    "gad_hgemm_workspace_bytes": (_i64, [C.POINTER(HGemmArgs)]),
    "gad_hgemm_plan": (C.c_int, [C.POINTER(HGemmArgs), C.POINTER(_i32), C.POINTER(_i32)]),
    "gad_hgemm": (C.c_int, [C.POINTER(HGemmArgs), _vp]),
    "gad_hgemm_tn_workspace_bytes": (_i64, [C.POINTER(HGemmArgs)]),
    "gad_hgemm_tn": (C.c_int, [C.POINTER(HGemmArgs), _vp]),
    "gad_h_transpose": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp]),
    "gad_h_cast": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "gad_h_shadow_pairs": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp]),
    "gad_h_add": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "gad_h_groupnorm_workspace_bytes": (_i64, [C.POINTER(GroupNormArgs)]),
    "gad_h_groupnorm_silu_fwd": (C.c_int, [C.POINTER(GroupNormArgs), _vp]),
    "gad_h_groupnorm_silu_bwd": (C.c_int, [C.POINTER(GroupNormArgs), _vp]),
    "gad_h_layernorm_fwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp]),
    "gad_h_layernorm_bwd": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp]),
    "gad_h_geglu_fwd": (C.c_int, [_vp, _vp, _i64, _i32, _vp]),
    "gad_h_geglu_bwd": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "gad_h_upsample2x_bwd": (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _vp]),
    "gad_h_attention_fwd": (C.c_int, [C.POINTER(AttentionArgs), _vp]),
    "gad_h_attention_bwd": (C.c_int, [C.POINTER(AttentionArgs), _vp]),
}

_lib = None


def load():
    """Load the shared library (raises GadError if it has not been built)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GadError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the product path has no CPU fallback)")
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)     # AttributeError if a declared symbol is not exported
            fn.restype, fn.argtypes = res, args
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        raise GadError(f"{what} failed (rc={rc}): {load().gad_last_error().decode()}")

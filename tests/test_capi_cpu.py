"""CPU checks of the drop-in boundary: the shared library loads without a GPU and exports
every symbol include/gad.h declares; the ctypes structs match the C layout."""
import ctypes
import os
import re

from gad import _capi


def _declared_symbols():
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "gad.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(gad_[a-z0-9_]+)\s*\(", hdr)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = _capi.load()
    names = _declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), n
        assert n in _capi.SIGNATURES, f"{n} declared in gad.h but not bound in _capi.SIGNATURES"
    assert sorted(_capi.SIGNATURES) == names
    assert lib.gad_version() >= 100


def test_struct_layout_matches_header(tmp_path):
    """Every field of every argument struct: sizeof and offsetof as gcc lays out include/gad.h against the ctypes
    mirror in gad/_capi.py (guards against a field drifting between the two)."""
    import subprocess
    structs = {"gad_conv_geom": _capi.ConvGeom, "gad_gemm_args": _capi.GemmArgs, "gad_groupnorm_args": _capi.GroupNormArgs,
               "gad_adam_args": _capi.AdamArgs, "gad_attention_args": _capi.AttentionArgs, "gad_hgemm_args": _capi.HGemmArgs}
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "gad.h"', 'int main(void) {']
    for cname, cls in structs.items():
        src.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            src.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    src += ['  return 0;', '}']
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    inc = os.path.join(os.path.dirname(__file__), "..", "include")
    subprocess.run(["gcc", "-I", inc, str(c), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in structs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"


def test_host_side_argument_validation_without_gpu():
    lib = _capi.load()
    a = _capi.GemmArgs()
    assert lib.gad_gemm(ctypes.byref(a), None) != 0          # null pointers are rejected on the host
    assert b"null" in lib.gad_last_error()
    a.A = a.B = a.C = 16
    a.M, a.N, a.K = 8, 8, 8
    a.lda = a.ldb = 8
    a.ldc = 4                                                  # ldc < N would write out of bounds
    assert lib.gad_gemm(ctypes.byref(a), None) != 0
    assert b"ldc" in lib.gad_last_error()


def test_product_path_has_no_cpu_fallback():
    import pytest
    import torch
    import gad
    net = gad.UNet2DModel(block_out_channels=(32, 32), down_block_types=("DownBlock2D", "DownBlock2D"),
                          up_block_types=("UpBlock2D", "UpBlock2D"), layers_per_block=1, attention_head_dim=None,
                          sample_size=8)
    with pytest.raises(_capi.GadError):
        net(torch.zeros(1, 3, 8, 8), torch.tensor([1]))


def test_error_state_is_per_thread():
    """the library is re-entrant; the last-error string is thread-local, so concurrent host threads
    (one per sampling stream) cannot read each other's failures."""
    import threading
    lib = _capi.load()
    seen = {}

    def worker(name, make_error):
        if make_error:
            a = _capi.GemmArgs()
            assert lib.gad_gemm(ctypes.byref(a), None) != 0
        barrier.wait()
        seen[name] = bytes(lib.gad_last_error())

    barrier = threading.Barrier(2)
    ts = [threading.Thread(target=worker, args=("bad", True)), threading.Thread(target=worker, args=("good", False))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert b"null" in seen["bad"] and seen["good"] == b""


def _conv_args(a_mode, b_mode, M, N, K, H, W, C, Ho=None, Wo=None, k=3, stride=1, pad=1, ups=0, prec=0):
    a = _capi.GemmArgs()
    a.a_mode, a.b_mode, a.M, a.N, a.K = a_mode, b_mode, M, N, K
    a.lda, a.ldb, a.ldc = (M if a_mode == _capi.A_MC else K), K, N
    a.g = _capi.ConvGeom(H, W, C, C, Ho or H, Wo or W, k, k, stride, pad, pad, ups)
    a.operand_precision = prec
    return a


def test_kernel_family_selection_is_a_pure_host_decision():
    """gad_gemm_kernel_id: 0 generic fp32, 1 generic bf16, 2 LDS-patch fp32, 3 LDS-patch bf16 - decided from shapes
    alone, so it can be checked without a GPU (the GPU tests check that each family computes the same numbers)."""
    lib = _capi.load()
    kid = lambda a: lib.gad_gemm_kernel_id(ctypes.byref(a))     # noqa: E731
    big = 512 * 32 * 32
    fwd = _conv_args(_capi.A_CONV, _capi.B_KC, big, 128, 9 * 128, 32, 32, 128)
    assert kid(fwd) == 2                                                     # 3x3 s1 p1, whole-row tiles, 128-tile plan
    fwd.operand_precision = 1
    assert kid(fwd) == 3
    assert kid(_conv_args(_capi.A_CONV, _capi.B_KC, big // 4, 128, 9 * 128, 32, 32, 128, Ho=16, Wo=16, stride=2, pad=0)) == 0
    assert kid(_conv_args(_capi.A_CONV, _capi.B_KC, big, 128, 128, 32, 32, 128, k=1, pad=0)) == 0        # 1x1
    assert kid(_conv_args(_capi.A_CONV, _capi.B_KC, big, 128, 9 * 100, 32, 32, 100)) == 0               # C % 32 != 0
    assert kid(_conv_args(_capi.A_CONV, _capi.B_KC, 2 * 32 * 32, 128, 9 * 128, 32, 32, 128)) == 0       # tiny M: split-K plan
    small = _conv_args(_capi.A_CONV, _capi.B_KC, 1024 * 16, 256, 9 * 256, 4, 4, 256)                     # 4x4 maps, B = 1024
    assert kid(small) == 2 and lib.gad_gemm_workspace_bytes(ctypes.byref(small)) == 2 * 1024 * 16 * 256 * 4   # 2 chunk splits
    up = _conv_args(_capi.A_CONV, _capi.B_KC, big, 256, 9 * 256, 16, 16, 256, Ho=32, Wo=32, ups=1)
    assert kid(up) == 2                                                      # nearest-2x fused into the patch fetch
    dg = _conv_args(_capi.A_CONVT, _capi.B_WDGRAD, big, 128, 9 * 128, 32, 32, 128)
    assert kid(dg) == 2
    wg = _conv_args(_capi.A_MC, _capi.B_CONV, 128, 9 * 128, 128 * 32 * 32, 32, 32, 128)
    assert kid(wg) == 2 and lib.gad_gemm_workspace_bytes(ctypes.byref(wg)) == 128 * 128 * 1152 * 4     # 128 pixel splits
    lin = _capi.GemmArgs()
    lin.a_mode, lin.b_mode, lin.M, lin.N, lin.K, lin.lda, lin.ldb, lin.ldc = _capi.A_KC, _capi.B_KC, 4096, 256, 256, 256, 256, 256
    assert kid(lin) == 0
    lin.operand_precision = 1
    assert kid(lin) == 1 and lib.gad_gemm_uses_bf16(ctypes.byref(lin)) == 1
    lin.K = lin.lda = lin.ldb = 27                                           # no aligned float4 path: stays fp32
    assert kid(lin) == 0 and lib.gad_gemm_uses_bf16(ctypes.byref(lin)) == 0


def test_winograd_weight_gradient_plan_is_the_same_with_a_kept_input_image():
    """A 3x3 weight gradient in Winograd F(4x4) form (GAD_GEMM_WINO_WGRAD) given the forward launch's transformed input
    (GAD_GEMM_WINO_SKIP_INPUT + B_wino4 = V) is the same plan - kernel id 7, the same scratch request - as without it: only the
    input-transform launch is dropped (host decisions, no GPU); a misaligned image is refused before anything is launched."""
    lib = _capi.load()
    wg = _conv_args(_capi.A_MC, _capi.B_CONV, 128, 9 * 128, 128 * 32 * 32, 32, 32, 128)
    wg.A = wg.B = wg.C = 4096
    wg.lda, wg.ldc, wg.alpha = 128, 9 * 128, 1.0
    wg.flags = _capi.GEMM_WINO_WGRAD
    assert lib.gad_gemm_kernel_id(ctypes.byref(wg)) == 7
    need = lib.gad_gemm_wino_bytes(ctypes.byref(wg))
    T = 128 * 32 * 32 // 16
    assert need == 36 * 4 * (T * 128 + T * 128 + 128 * 128)                  # transformed dy, transformed x, the 36 product panels
    wg.flags |= _capi.GEMM_WINO_SKIP_INPUT
    wg.B_wino4 = 8192
    assert lib.gad_gemm_kernel_id(ctypes.byref(wg)) == 7 and lib.gad_gemm_wino_bytes(ctypes.byref(wg)) == need
    wg.wino_ws, wg.wino_ws_bytes, wg.B_wino4 = 1 << 20, need, 8196          # image not 16-byte aligned
    assert lib.gad_gemm(ctypes.byref(wg), None) != 0 and b"kept Winograd input image" in lib.gad_last_error()


def test_half_path_host_side_checks_and_planner_without_gpu():
    """gad_hgemm's argument validation and its planner are host code: misuse is rejected before anything is launched, and the
    tile / split-K choices for the SD step's shapes are the ones DESIGN.md §4.4 states (no GPU needed)."""
    lib = _capi.load()
    a = _capi.HGemmArgs()
    assert lib.gad_hgemm(ctypes.byref(a), None) != 0 and b"null" in lib.gad_last_error()
    a.A = a.B = a.C = 4096
    a.M, a.N, a.K, a.lda, a.ldb, a.ldc, a.k_split = 64, 64, 36, 40, 40, 64, 36
    assert lib.gad_hgemm(ctypes.byref(a), None) != 0 and b"multiples of 8" in lib.gad_last_error()      # K % 8
    a.K = a.k_split = 40
    a.A = 4098
    assert lib.gad_hgemm(ctypes.byref(a), None) != 0 and b"aligned" in lib.gad_last_error()
    assert lib.gad_hgemm_workspace_bytes(ctypes.byref(a)) < 0

    def plan(M, N, K, conv=False, out_f32=False):
        g = _capi.HGemmArgs()
        g.A = g.B = g.C = 4096
        g.M, g.N, g.K, g.ldb, g.ldc, g.out_f32 = M, N, K, K, N, int(out_f32)
        if conv:
            cin = K // 9
            hw = int(round((M // 16) ** 0.5))
            g.conv, g.KH, g.KW, g.Cin, g.H, g.W, g.Ho, g.Wo, g.stride, g.pad_t, g.pad_l = 1, 3, 3, cin, hw, hw, hw, hw, 1, 1, 1
            g.lda, g.k_split = cin, cin
        else:
            g.lda, g.k_split = K, K
        tile, sk = ctypes.c_int32(), ctypes.c_int32()
        assert lib.gad_hgemm_plan(ctypes.byref(g), ctypes.byref(tile), ctypes.byref(sk)) == 0, lib.gad_last_error()
        need = lib.gad_hgemm_workspace_bytes(ctypes.byref(g))
        assert need == (sk.value * M * N * 4 if sk.value > 1 else 0)
        return tile.value, sk.value
    assert plan(65536, 320, 2880, conv=True) == (6, 1)             # 64x64 maps: one round of 256 x 320 tiles on eight waves
    assert plan(16384, 640, 5760, conv=True) == (6, 2)             # 32x32 maps: the same form, two K slices
    assert plan(4096, 1280, 11520, conv=True) == (6, 4)
    t, sk = plan(1024, 1280, 11520, conv=True)                     # 8x8 maps: 128 x 320 tiles, split
    assert t == 7 and sk >= 8
    assert plan(65536, 2560, 320) == (7, 1)                        # GEGLU projection: full rounds of 128 x 320
    assert plan(65536, 256, 320) == (8, 1)                         # LoRA rank products: 128 x 128, four workgroups per CU
    assert plan(4096, 1280, 1280) == (9, 1)                        # 16x16 level Linear: more 128 x 128 tiles than 128 x 320 ones
    assert plan(65536, 4, 2880, conv=True)[0] in (1, 8, 9)         # conv_out

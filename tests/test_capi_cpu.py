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


def test_struct_layout_matches_header():
    # sizes computed from the C declaration (LP64): guards against a field drifting between gad.h and _capi.py
    assert ctypes.sizeof(_capi.ConvGeom) == 12 * 4
    assert ctypes.sizeof(_capi.GemmArgs) == 3 * 8 + 10 * 4 + 6 * 8 + 48 + 4 + 4 + 8 + 8 + 4 + 4 + 8 + 4 + 4 + 8 + 8 + 4 + 4 + 4 + 4 + 8 + 4 + 4   # ... hints, operand_precision, pad, A2, a_split, ldx2
    assert ctypes.sizeof(_capi.GroupNormArgs) == 9 * 8 + 4 * 4 + 4 + 4 + 8 + 8 + 8 + 4 + 4   # ... x2, C1, tail pad
    assert ctypes.sizeof(_capi.AdamArgs) == 5 * 8 + 8 + 8 + 4 + 5 * 4 + 4 + 4 + 4 + 4


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

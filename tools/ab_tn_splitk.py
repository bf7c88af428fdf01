"""Weight-gradient (tn) GEMMs of the CIFAR training step (dW[N][K] = dy[tokens][N]^T x[tokens][K]): the planner's choice against forced
tiles x split-K factors.  usage (GPU box): python tools/ab_tn_splitk.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd"))
sys.path.insert(0, ROOT)
import ctypes as C  # noqa: E402

import torch  # noqa: E402
from gad import _capi, ops  # noqa: E402
from gad._capi import A_MC, B_MC  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


SHAPES = [(32768, 256, 256), (131072, 128, 256), (131072, 128, 384), (32768, 256, 384), (32768, 256, 512), (8192, 256, 512), (8192, 256, 256),
          (131072, 128, 128), (128, 256, 512), (128, 128, 512)]
for T, N, K in SHAPES:
    dy, x, dw = torch.randn(T, N, device=dev), torch.randn(T, K, device=dev), torch.empty(N, K, device=dev)
    fl = 2.0 * T * N * K
    line = f"tokens {T:6d} N {N:4d} K {K:4d}: auto {timeit(lambda: ops.gemm_raw(dy, x, dw, A_MC, B_MC, N, K, T, N, K, K)):6.1f} us |"
    for tile in (1, 2):
        r = []
        for sk in (8, 16, 32, 64, 128):
            if sk * 32 * 4 > T:
                continue
            r.append(f"sk{sk} {timeit(lambda: ops.gemm_raw(dy, x, dw, A_MC, B_MC, N, K, T, N, K, K, tile_hint=tile, splitk_hint=sk)):6.1f}")
        line += f" tile {('128', '64')[tile - 1]}: " + " ".join(r) + " |"
    print(line + f"  ({fl / 1e9:.1f} GFLOP)", flush=True)

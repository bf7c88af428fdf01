"""Is the half path's contraction engine bound by its memory system?  The same launches with every DMA source on the zero block."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "group-attribution-for-diffusion-models_amd"))
import torch
from gad import half
from ab_hgemm import timeit, dev, BF
for Bn, H, Cin, Cout in [(16, 64, 320, 320), (16, 64, 960, 320), (16, 32, 1280, 640)]:
    x = (torch.randn(Bn, H, H, Cin, device=dev) * 0.5).to(BF)
    w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02).to(BF)
    M, K = Bn * H * H, 9 * Cin
    geom = (H, H, Cin, H, H, 3, 3, 1, 1, 1, 0)
    y = torch.empty(Bn, H, H, Cout, device=dev, dtype=BF)
    for tile in (7, 107, 6, 106):
        t = timeit(lambda: half.hgemm_raw(x, w, y, M, Cout, K, Cin, K, Cout, conv=1, geom=geom, k_split=Cin, tile_hint=tile, splitk_hint=1))
        print(f"conv {H}x{H} {Cin}->{Cout} tile_hint {tile}: {t:.1f} us {2.0 * M * Cout * K / t / 1e6:.0f} TF/s", flush=True)
for M, N, K in [(65536, 2560, 320), (65536, 320, 1280)]:
    a = (torch.randn(M, K, device=dev) * 0.5).to(BF)
    b = (torch.randn(N, K, device=dev) * 0.05).to(BF)
    out = torch.empty(M, N, device=dev, dtype=BF)
    for tile in (7, 107, 6, 106):
        t = timeit(lambda: half.hgemm_raw(a, b, out, M, N, K, K, K, N, tile_hint=tile, splitk_hint=1))
        print(f"dense {M}x{N}x{K} tile_hint {tile}: {t:.1f} us {2.0 * M * N * K / t / 1e6:.0f} TF/s", flush=True)

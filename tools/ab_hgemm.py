"""A/B of the half path's contraction engine over the SD LoRA step's shapes (B = 16 at 512^2): tile / ring / split-K forms.
Run on the GPU box: python tools/ab_hgemm.py > gpurun_out/ab_hgemm.txt"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "group-attribution-for-diffusion-models_amd"))
import torch
from gad import half

dev = torch.device("cuda:0")
BF = torch.bfloat16


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3          # us


DENSE = [  # M, N, K
    (16384, 640, 1280), (4096, 1280, 640), (16384, 256, 1280), (4096, 256, 1280), (4096, 1280, 256),
    (65536, 320, 320), (65536, 2560, 320), (65536, 320, 1280), (16384, 640, 640), (16384, 5120, 640), (16384, 640, 2560),
    (4096, 1280, 1280), (4096, 10240, 1280), (4096, 1280, 5120), (65536, 256, 320), (16384, 256, 640), (65536, 320, 576), (1232, 320, 768),
]
CONV = [  # B, H, Cin, Cout
    (16, 64, 320, 320), (16, 64, 640, 320), (16, 64, 960, 320), (16, 32, 640, 640), (16, 32, 1280, 640), (16, 32, 1920, 640),
    (16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 8, 1280, 1280), (16, 8, 2560, 1280),
]
FORMS = [(1, 1), (9, 1), (8, 1), (7, 1), (0, 0)]          # (tile_hint, splitk_hint); (0, 0) = planner

if __name__ == "__main__":
    print("dense: us (TF/s) per form  [tile,sk]:", FORMS)
    for M, N, K in DENSE:
        a = (torch.randn(M, K, device=dev) * 0.5).to(BF)
        b = (torch.randn(N, K, device=dev) * 0.05).to(BF)
        out = torch.empty(M, N, device=dev, dtype=BF)
        row = []
        for tile, sk in FORMS:
            if tile in (2, 5, 6, 7) and N % 320:
                row.append("      -      ")
                continue
            t = timeit(lambda: half.hgemm_raw(a, b, out, M, N, K, K, K, N, tile_hint=tile, splitk_hint=sk))
            row.append("%7.1f (%4.0f)" % (t, 2.0 * M * N * K / t / 1e6))
        print("M %6d N %5d K %5d | " % (M, N, K) + " | ".join(row), flush=True)
    print("token-axis contraction (hgemm_tn): us per split-K hint [0 = planner, 16, 32, 64, 128]")
    for T, M, N in [(65536, 320, 320), (65536, 320, 256), (16384, 640, 640), (16384, 640, 256), (4096, 1280, 256), (1232, 320, 768)]:
        a = (torch.randn(T, M, device=dev) * 0.5).to(BF)
        b = (torch.randn(T, N, device=dev) * 0.5).to(BF)
        out = torch.empty(M, N, device=dev)
        row = []
        for sk in (0, 16, 32, 64, 128):
            t = timeit(lambda: half.wgrad_raw(a, b, out, False, splitk_hint=sk))
            row.append("%7.1f (%4.0f)" % (t, 2.0 * T * M * N / t / 1e6))
        print("T %6d M %5d N %5d | " % (T, M, N) + " | ".join(row), flush=True)
    if "dense" in sys.argv[1:]:
        sys.exit(0)
    CONV[:] = CONV[-4:]
    print("conv3x3: us (TF/s)")
    for Bn, H, Cin, Cout in CONV:
        x = (torch.randn(Bn, H, H, Cin, device=dev) * 0.5).to(BF)
        w = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.02).to(BF)
        M, K = Bn * H * H, 9 * Cin
        geom = (H, H, Cin, H, H, 3, 3, 1, 1, 1, 0)
        y = torch.empty(Bn, H, H, Cout, device=dev, dtype=BF)
        row = []
        for tile, sk in [(6, 1), (6, 2), (6, 4), (2, 10), (7, 10), (2, 16), (7, 16), (0, 0)]:
            t = timeit(lambda: half.hgemm_raw(x, w, y, M, Cout, K, Cin, K, Cout, conv=1, geom=geom, k_split=Cin, tile_hint=tile, splitk_hint=sk))
            row.append("%7.1f (%4.0f)" % (t, 2.0 * M * Cout * K / t / 1e6))
        print("B %2d %2dx%2d Cin %4d Cout %4d | " % (Bn, H, H, Cin, Cout) + " | ".join(row), flush=True)

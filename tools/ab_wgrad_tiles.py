"""Weight-gradient patch kernel: output-channel tile plans at the CelebA / pruned-CelebA shapes - one launch on 128 / 96 / 64-
channel tiles (tile_hint 1 / 4 / 5), forced two-launch splits (tile_hint 1000 + m1: rows [0, m1) on 128-channel tiles, the
rest planned) and the planner's own choice (0).  Calibrates wgrad_tile_cost in csrc/gemm_f32.hip.
usage: python tools/ab_wgrad_tiles.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
from gad import ops
dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


SHAPES = [(32, 224, 224, 64), (32, 448, 448, 32), (32, 672, 672, 16), (32, 160, 160, 64), (32, 320, 320, 32), (32, 480, 480, 16),
          (128, 192, 192, 16), (128, 96, 96, 32), (128, 128, 64, 32)]
for B, Cin, Cout, H in SHAPES:
    x = torch.randn(B, H, H, Cin, device=dev)
    dy = torch.randn(B, H, H, Cout, device=dev)
    w = torch.empty(Cout, Cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    fl = 2.0 * B * H * H * Cout * Cin * 9
    hints = [0, 1, 4, 5] + [1000 + m1 for m1 in range(128, Cout, 128)]
    line = f"B={B} {Cin}->{Cout} @{H}x{H}: "
    for h in hints:
        try:
            ms = timeit(lambda: ops.conv2d_wgrad_raw(dy, x, w, tile_hint=h))
            line += f"{'auto' if h == 0 else h}: {ms*1e3:7.1f}us {fl/ms/1e9:5.1f}TF | "
        except Exception as e:
            line += f"{h}: err | "
    print(line, flush=True)

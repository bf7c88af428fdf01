"""Weight gradient of the models' 3x3 convolutions: Winograd F(4x4,3x3) route (wino4_dy_kernel + wino4_input_kernel + 36 batched
products + wino4_dw_kernel) against the direct LDS-patch kernels (HIP events, median).  TF/s are ALGORITHMIC for both.
usage: python tools/ab_winograd_wgrad.py [launches]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
from gad import ops
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 7


def ev(fn):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(N):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]


SHAPES = [(128, 32, 128, 128), (128, 32, 256, 128), (128, 16, 256, 256), (128, 16, 512, 256), (128, 8, 256, 256), (128, 4, 256, 256),
          (128, 32, 96, 96), (128, 16, 192, 192), (128, 16, 288, 192),
          (32, 64, 224, 224), (32, 32, 448, 448), (32, 16, 672, 672), (32, 8, 896, 896), (32, 64, 160, 160), (32, 32, 320, 320)]
for B, H, Cin, Cout in SHAPES:
    x = torch.randn(B, H, H, Cin, device=dev)
    dy = torch.randn(B, H, H, Cout, device=dev)
    w = torch.zeros(Cout, Cin, 3, 3, device=dev).contiguous(memory_format=torch.channels_last)
    ops.PROFILER = prof = ops.GemmProfiler()
    ops.conv2d_wgrad_raw(dy, x, w)
    torch.cuda.synchronize()
    ops.PROFILER = None
    took = "winograd" if "wino" in list(prof.summary())[0][0] else "direct  "
    ta = ev(lambda: ops.conv2d_wgrad_raw(dy, x, w))
    t8 = ev(lambda: ops.conv2d_wgrad_raw(dy, x, w, tile_hint=8))
    with ops.kernel_flags(no_wino=True):
        t0 = ev(lambda: ops.conv2d_wgrad_raw(dy, x, w))
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    print(f"B{B:4d} {H:2d}x{H:<2d} {Cin:4d}->{Cout:<4d}: planner {took} {ta:7.3f} ms | forced winograd {t8:7.3f} ms ({fl / t8 / 1e9:5.0f} TF/s) | "
          f"direct kernels {t0:7.3f} ms ({fl / t0 / 1e9:5.0f} TF/s) | x{t0 / t8:.2f}", flush=True)

"""Winograd routes - F(2x2,3x3) (wino_input_kernel + wino_gemm_kernel), F(4x4,3x3) in its three-launch and one-launch forms
(wino4_input_kernel + wino4_gemm / batched products + wino4_output_kernel; wino4_input_kernel + wino4_fused[2]_kernel) - against the direct LDS-patch kernels on the 3x3 convolutions
of the reference's models (HIP events, median).  TF/s are ALGORITHMIC (2 x 9 Cin Cout per output pixel) for both.
usage: python tools/ab_winograd.py [launches]"""
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


SHAPES = [  # B, H, Cin, Cout          CIFAR sampler (B = 1024) and trainer (B = 128), pruned widths, SD-1.x at 512^2 / 256^2, CelebA-HQ LDM
    (1024, 32, 128, 128), (1024, 32, 256, 128), (1024, 16, 256, 256), (1024, 16, 512, 256), (1024, 8, 256, 256), (1024, 4, 256, 256),
    (128, 32, 128, 128), (128, 16, 256, 256), (128, 8, 256, 256),
    (1024, 32, 96, 96), (1024, 16, 192, 192), (1024, 8, 288, 192),
    (16, 64, 320, 320), (16, 32, 640, 640), (16, 16, 1280, 1280), (16, 8, 1280, 1280), (64, 32, 320, 320), (64, 16, 640, 640),
    (32, 64, 224, 224), (32, 32, 448, 448), (32, 16, 672, 672), (32, 8, 896, 896),
]
for B, H, Cin, Cout in SHAPES:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev)
    ops.PROFILER = prof = ops.GemmProfiler()
    ops.conv2d_fwd_raw(x, w, b)
    torch.cuda.synchronize()
    ops.PROFILER = None
    took = "F(4x4)  " if any("wino4" in k[0] for k in prof.summary()) else ("F(2x2)  " if any("wino" in k[0] for k in prof.summary()) else "direct  ")
    t1 = ev(lambda: ops.conv2d_fwd_raw(x, w, b))
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    with ops.kernel_flags(no_wino=True):
        t0 = ev(lambda: ops.conv2d_fwd_raw(x, w, b))
    t2 = ev(lambda: ops.conv2d_fwd_raw(x, w, b, tile_hint=7))
    t4 = ev(lambda: ops.conv2d_fwd_raw(x, w, b, tile_hint=10)) if H % 4 == 0 else float("nan")      # three launches (products -> HBM -> output transform)
    t9 = ev(lambda: ops.conv2d_fwd_raw(x, w, b, tile_hint=9)) if H % 4 == 0 else float("nan")       # input transform + wino4_fused_kernel
    t11 = ev(lambda: ops.conv2d_fwd_raw(x, w, b, tile_hint=11)) if H % 4 == 0 else float("nan")     # the same on 64-tile blocks, one workgroup per CU
    ex = fl / 4.0 / t9 / 1e9 if t9 == t9 else float("nan")
    forced = (f" | forced F(2x2) {t2:7.3f} ms (x{t0 / t2:.2f}) F(4x4) three-launch {t4:7.3f} ms (x{t0 / t4:.2f}) one-launch/64 {t11:7.3f} ms one-launch/32 {t9:7.3f} ms (x{t0 / t9:.2f}, "
              f"{ex:5.1f} TF/s executed incl. the input transform = {ex / 157.3:.2f})")
    print(f"B{B:5d} {H:2d}x{H:<2d} {Cin:4d}->{Cout:<4d}: planner {took} {t1:7.3f} ms ({fl / t1 / 1e9:6.1f} TF/s) | direct kernels {t0:7.3f} ms "
          f"({fl / t0 / 1e9:6.1f} TF/s) | x{t0 / t1:.2f}" + forced, flush=True)

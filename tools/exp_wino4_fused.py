"""Timing experiments on wino4_fused2_kernel (results of modes 12-15 are WRONG by construction: they switch parts of the kernel off).
9 = full kernel; 12 = no epilogue stores; 13 = no DMA in the K loop; 14 = DMA re-reads stage-step 0 (always L2-hot); 15 = 12 + 13."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
from gad import ops
dev = torch.device("cuda:0")
def ev(fn, N=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(N):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]
for B, H, Cin, Cout in [(1024, 32, 128, 128), (1024, 32, 256, 128), (1024, 16, 256, 256), (1024, 16, 512, 256)]:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev)
    res = torch.randn(B, H, H, Cout, device=dev)
    T = B * H * H // 16
    out = [f"B{B} {H}x{H} {Cin}->{Cout}: ideal MFMA {36 * T * Cin * Cout * 2 / 157.3e12 * 1e3:.3f} ms |"]
    for hint in (9, 12, 13, 14, 15, 11, 10):
        t = ev(lambda: ops.conv2d_fwd_raw(x, w, b, tile_hint=hint, residual=res))
        out.append(f"h{hint} {t:.3f}")
    print(" ".join(out), flush=True)

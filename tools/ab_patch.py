"""A/B: fp32 LDS-patch conv kernel vs the generic im2col-gather kernel on the eligible shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
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
for B in (1024, 128):
    for Cin, Cout, H in ((128, 128, 32), (256, 128, 32), (384, 128, 32), (256, 256, 16), (512, 256, 16), (384, 256, 16), (128, 256, 16),
                         (96, 96, 32), (192, 96, 32), (288, 96, 32), (192, 192, 16), (384, 192, 16), (320, 320, 32), (640, 320, 32)):
        x = torch.randn(B, H, H, Cin, device=dev)
        w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Cout, device=dev)
        fl = 2.0 * B * H * H * Cout * Cin * 9
        res = []
        outs = []
        for label, flags in (("patch", {}), ("generic", {"no_patch": True})):
            with ops.kernel_flags(**flags):
                outs.append(ops.conv2d_fwd_raw(x, w, b))
                ms = timeit(lambda: ops.conv2d_fwd_raw(x, w, b))
            res.append(f"{label} {fl/ms/1e9:6.1f} TF/s ({ms*1e3:6.0f} us)")
        res.append(f"max|patch-generic| {(outs[0]-outs[1]).abs().max().item():.1e}")
        print(f"B={B} {Cin}->{Cout}@{H}: " + " | ".join(res), flush=True)

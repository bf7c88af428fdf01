import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch
from gad import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
for B in (1024, 128):
    for C, H in ((128, 32), (256, 32), (384, 32), (256, 16), (384, 16), (512, 16), (256, 8), (512, 8), (256, 4), (96, 32), (192, 32), (288, 32), (192, 16), (288, 16)):
        x = torch.randn(B, H, H, C, device=dev); g = torch.randn(C, device=dev); b = torch.randn(C, device=dev)
        res = []
        ref = None
        for label, two in (("two-pass", True), ("one-pass slab", False)):
            with torch.no_grad(), ops.kernel_flags(gn_two_pass=two):
                y = ops.group_norm(x, g, b, 32, 1e-6, True)
                if ref is None: ref = y
                err = (y - ref).abs().max().item()
                t = timeit(lambda: ops.group_norm(x, g, b, 32, 1e-6, True))
            res.append(f"{label} {t*1e6:7.1f}us {8*x.numel()/t/1e9:5.0f}GB/s err {err:.1e}")
        print(f"B={B} C={C} {H}x{H}: " + " | ".join(res), flush=True)

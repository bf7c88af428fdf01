"""A/B: GroupNorm(+SiLU) backward, one-pass slab kernel (default) vs the two-pass stats + apply kernels
(kernel_flags(gn_two_pass=True)); GB/s counts the algorithmic 12 B/elem (read x, dy; write dx).
usage: python tools/gn_bwd_ab.py"""
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
for B, shapes in ((128, ((128, 32), (256, 32), (384, 32), (256, 16), (384, 16), (512, 16), (256, 8), (512, 8), (256, 4))),
                  (64, ((320, 32), (640, 16), (1280, 8), (960, 32), (1920, 16), (2560, 8)))):
    for C, H in shapes:
        x = torch.randn(B, H, H, C, device=dev, requires_grad=True)
        g = torch.randn(C, device=dev, requires_grad=True); b = torch.randn(C, device=dev, requires_grad=True)
        dy = torch.randn(B, H, H, C, device=dev)
        res, grads = [], []
        for label, two in (("two-pass", True), ("one-pass", False)):
            with ops.kernel_flags(gn_two_pass=two):
                y = ops.group_norm(x, g, b, 32, 1e-6, True)
                gx, gg, gb = torch.autograd.grad(y, (x, g, b), dy, retain_graph=True)
                grads.append((gx, gg, gb))
                t = timeit(lambda: torch.autograd.grad(y, (x, g, b), dy, retain_graph=True))
            res.append(f"{label} {t*1e6:7.1f}us {12*x.numel()/t/1e9:5.0f}GB/s")
        errs = [((a - c).abs().max() / c.abs().max()).item() for a, c in zip(grads[1], grads[0])]
        print(f"B={B} C={C} {H}x{H}: " + " | ".join(res) + f" | rel diff dx {errs[0]:.1e} dgamma {errs[1]:.1e} dbeta {errs[2]:.1e}", flush=True)

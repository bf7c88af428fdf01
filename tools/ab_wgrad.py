"""A/B: LDS-patch wgrad kernel vs the generic im2col-columns kernel (B=128 training shapes)."""
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
B = 128
for Cin, Cout, H in ((128, 128, 32), (256, 256, 16), (512, 256, 16), (256, 256, 8), (512, 256, 8), (96, 96, 32), (192, 192, 16), (384, 192, 16)):
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    dy = torch.randn(B, H, H, Cout, device=dev)
    fl = 2.0 * B * H * H * Cout * Cin * 9
    res = []
    outs = []
    for no_patch in (False, True):
        with ops.kernel_flags(no_patch=no_patch):
            outs.append(ops.conv2d_wgrad_raw(dy, x, w))
            ms = timeit(lambda: ops.conv2d_wgrad_raw(dy, x, w))
        res.append(f"{'generic' if no_patch else 'patch'} {fl/ms/1e9:6.1f}")
    err = (outs[0] - outs[-1]).abs().max().item()
    print(f"wgrad B={B} {Cin}->{Cout}@{H}: " + " | ".join(res) + f" | max diff {err:.2e}", flush=True)

"""A/B timing of conv-forward shapes (run twice with different env to compare)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
from gad import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for B in (32, 128):
    for (Cin, Cout, H) in [(128, 128, 32), (256, 256, 16), (384, 128, 32), (512, 256, 16), (256, 256, 8)]:
        x = torch.randn(B, H, H, Cin, device=dev)
        w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Cout, device=dev)
        fl = 2.0 * B * H * H * Cout * Cin * 9
        r = []
        for tile in (1, 2):
            ms = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, (1, 1, 1, 1), False, tile_hint=tile))
            r.append(f"t{tile}: {fl/ms/1e9:6.1f} TF ({ms*1e3:5.0f} us)")
        dy = torch.randn(B, H, H, Cout, device=dev)
        ms_d = timeit(lambda: ops.conv2d_dgrad_raw(dy, w, x.shape, 1, (1, 1, 1, 1), False))
        ms_w = timeit(lambda: ops.conv2d_wgrad_raw(dy, x, w, 1, (1, 1, 1, 1), False))
        print(f"PRIO={os.environ.get('GAD_GEMM_PRIO','0')} B={B:3d} {Cin}->{Cout}@{H}: " + "  ".join(r) + f"  dgrad {fl/ms_d/1e9:6.1f}  wgrad {fl/ms_w/1e9:6.1f}", flush=True)

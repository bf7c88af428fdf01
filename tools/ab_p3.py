import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
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
    return s.elapsed_time(e) / iters
for B in (128, 512):
    for (Cin, Cout, H) in [(128,128,32),(256,256,16),(384,128,32),(512,256,16),(256,256,8)]:
        x = torch.randn(B, H, H, Cin, device=dev)
        w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Cout, device=dev)
        dy = torch.randn(B, H, H, Cout, device=dev)
        fl = 2.0 * B * H * H * Cout * Cin * 9
        out = []
        ys = {}
        for mode in ("1", "0", "1", "0"):
            os.environ["GAD_NO_P3"] = mode
            ms = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, (1,1,1,1), False, tile_hint=1))
            md = timeit(lambda: ops.conv2d_dgrad_raw(dy, w, x.shape, 1, (1,1,1,1), False, tile_hint=1))
            mw = timeit(lambda: ops.conv2d_wgrad_raw(dy, x, w, 1, (1,1,1,1), False, tile_hint=1, splitk_hint=32))
            ys[mode] = ops.conv2d_fwd_raw(x, w, b, 1, (1,1,1,1), False, tile_hint=1)
            out.append(f"{'p2' if mode=='1' else 'p3'}: f{fl/ms/1e9:6.1f} d{fl/md/1e9:6.1f} w{fl/mw/1e9:6.1f}")
        err = (ys["0"] - ys["1"]).abs().max().item()
        print(f"B={B} {Cin}->{Cout}@{H}: " + " | ".join(out) + f" | maxdiff {err:.2e}", flush=True)

"""Tile / split-K sweep over the CIFAR U-Net conv shapes (fwd, dgrad, wgrad) to tune make_plan()."""
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
shapes = [(128,128,32),(128,256,16),(256,256,16),(256,256,8),(256,256,4),(512,256,4),(512,256,8),(512,256,16),(384,256,16),(384,128,32),(256,128,32)]
which = sys.argv[1:] or ["fwd", "dgrad", "wgrad"]
for B in (32, 128):
    for (Cin, Cout, H) in shapes:
        x = torch.randn(B, H, H, Cin, device=dev)
        w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
        b = torch.randn(Cout, device=dev)
        dy = torch.randn(B, H, H, Cout, device=dev)
        fl = 2.0 * B * H * H * Cout * Cin * 9
        for kind in which:
            if kind != "fwd" and B == 32: continue
            res = {}
            for tile in (1, 2):
                for sk in (1, 2, 4, 8, 16, 32, 64):
                    try:
                        if kind == "fwd": fn = lambda: ops.conv2d_fwd_raw(x, w, b, 1, (1,1,1,1), False, tile_hint=tile, splitk_hint=sk)
                        elif kind == "dgrad": fn = lambda: ops.conv2d_dgrad_raw(dy, w, x.shape, 1, (1,1,1,1), False, tile_hint=tile, splitk_hint=sk)
                        else: fn = lambda: ops.conv2d_wgrad_raw(dy, x, w, 1, (1,1,1,1), False, tile_hint=tile, splitk_hint=sk)
                        res[(tile, sk)] = fl / timeit(fn) / 1e9
                    except Exception as ex:
                        pass
            if kind == "fwd": auto = fl / timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, (1,1,1,1), False)) / 1e9
            elif kind == "dgrad": auto = fl / timeit(lambda: ops.conv2d_dgrad_raw(dy, w, x.shape, 1, (1,1,1,1), False)) / 1e9
            else: auto = fl / timeit(lambda: ops.conv2d_wgrad_raw(dy, x, w, 1, (1,1,1,1), False)) / 1e9
            best = sorted(res.items(), key=lambda kv: -kv[1])[:4]
            M = B*H*H
            print(f"{kind:5s} B={B:3d} {Cin}->{Cout}@{H:2d} M={M:6d} auto {auto:6.1f} | " + "  ".join(f"t{128 if k[0]==1 else 64}/sk{k[1]}:{v:6.1f}" for k, v in best), flush=True)

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
B=128
for (Cin, Cout, H) in [(128,128,32),(256,256,16),(384,128,32)]:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev)
    fl = 2.0 * B * H * H * Cout * Cin * 9
    ms = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, (1,1,1,1), False, tile_hint=1))
    print(f"DBG={os.environ.get('GAD_DBG','0')} {Cin}->{Cout}@{H}: {fl/ms/1e9:6.1f} TF ({ms*1e3:.0f} us)", flush=True)

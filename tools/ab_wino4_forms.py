"""The one-launch F(4x4) forms against each other and the three-launch form, with and without a residual, median of 7, ms, incl. the
input transform.  hint 9: 32 tiles x 64 ch, 4 waves, two workgroups per CU (shipped); 12: 64 tiles x 64 ch, 8 waves of 16 x 32, one
workgroup per CU (one U tile for eight waves); 11: 64 tiles, 4 waves of 32 x 32; 10: three launches."""
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
for B, H, Cin, Cout in [(1024, 32, 128, 128), (1024, 32, 256, 128), (1024, 32, 384, 128), (1024, 16, 256, 256), (1024, 16, 512, 256), (1024, 8, 256, 256), (128, 32, 128, 128), (128, 16, 256, 256)]:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev)
    res = torch.randn(B, H, H, Cout, device=dev)
    ref = ops.conv2d_fwd_raw(x, w, b, tile_hint=10, residual=res)
    out = [f"B{B} {H}x{H} {Cin}->{Cout}:"]
    for hint in (9, 11, 10):
        y = ops.conv2d_fwd_raw(x, w, b, tile_hint=hint, residual=res)
        err = (y - ref).abs().max().item()
        t0 = ev(lambda: ops.conv2d_fwd_raw(x, w, b, tile_hint=hint))
        t1 = ev(lambda: ops.conv2d_fwd_raw(x, w, b, tile_hint=hint, residual=res))
        out.append(f"h{hint} {t0:.3f} / res {t1:.3f} (err {err:.1e})")
    print(" | ".join(out), flush=True)

import os, sys
sys.path.insert(0, "/root/repo/group-attribution-for-diffusion-models_amd"); sys.path.insert(0, "/root/repo")
import torch
from gad import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
def ev(fn, n=7):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]
# correctness of the 8-wave variant
x = torch.randn(128, 32, 32, 64, device=dev); w = (torch.randn(128, 64, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
b = torch.randn(128, device=dev); ra = torch.randn(128, 128, device=dev); res = torch.randn(128, 32, 32, 128, device=dev)
y7 = ops.conv2d_fwd_raw(x, w, b, rowadd=ra, residual=res, tile_hint=7)
y8 = ops.conv2d_fwd_raw(x, w, b, rowadd=ra, residual=res, tile_hint=8)
y9 = ops.conv2d_fwd_raw(x, w, b, rowadd=ra, residual=res, tile_hint=9)
print("3-stage equal:", torch.equal(y9, y8))
with ops.kernel_flags(no_wino=True):
    y0 = ops.conv2d_fwd_raw(x, w, b, rowadd=ra, residual=res)
print("8-wave vs 4-wave equal:", torch.equal(y7, y8), "max diff vs direct", (y7 - y0).abs().max().item(), (y8 - y0).abs().max().item(), flush=True)
for (B, H, Cin, Cout) in [(1024, 32, 256, 256), (1024, 32, 128, 128), (1024, 16, 256, 256), (1024, 8, 256, 256), (128, 32, 256, 256), (128, 16, 256, 256), (128, 32, 128, 128),
                          (1024, 32, 512, 256), (1024, 16, 384, 256), (16, 64, 640, 640), (32, 64, 448, 448)]:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev)
    t7 = ev(lambda: ops.conv2d_fwd_raw(x, w, b, tile_hint=7))
    t8 = ev(lambda: ops.conv2d_fwd_raw(x, w, b, tile_hint=8))
    ta = ev(lambda: ops.conv2d_fwd_raw(x, w, b))
    t9 = ev(lambda: ops.conv2d_fwd_raw(x, w, b, tile_hint=9))
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    print(f"B{B} {H}x{H} {Cin}->{Cout}: 8-wave 128x128 {t7:.3f} ms ({fl/t7/1e9:.1f}) | 4-wave {t8:.3f} ms ({fl/t8/1e9:.1f}) | 4-wave 3-stage {t9:.3f} ms ({fl/t9/1e9:.1f}) | auto {ta:.3f}", flush=True)

import os, sys, time
sys.path.insert(0, "/root/repo/group-attribution-for-diffusion-models_amd"); sys.path.insert(0, "/root/repo")
import torch
from gad import ops, _capi
dev = torch.device("cuda:0")
torch.manual_seed(0)
def conv_ref(x, w, b, up):
    xn = x.permute(0, 3, 1, 2).double()
    if up: xn = torch.nn.functional.interpolate(xn, scale_factor=2, mode="nearest")
    y = torch.nn.functional.conv2d(xn, w.double(), b.double() if b is not None else None, padding=1)
    return y.permute(0, 2, 3, 1)
for (B, H, Cin, Cout, up) in [(64, 32, 64, 128, False), (128, 16, 96, 192, False), (64, 8, 256, 256, True), (128, 16, 128, 320, False), (256, 4, 32, 64, True), (33, 32, 32, 68, False), (2, 64, 64, 1024, False)]:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev)
    He = 2 * H if up else H
    rowadd = torch.randn(B, Cout, device=dev)
    res = torch.randn(B, He, He, Cout, device=dev)
    lib = _capi.load()
    y = ops.conv2d_fwd_raw(x, w, b, upsample=up, rowadd=rowadd, residual=res)
    with ops.kernel_flags(no_wino=True):
        y0 = ops.conv2d_fwd_raw(x, w, b, upsample=up, rowadd=rowadd, residual=res)
    ref = conv_ref(x, w, b, up) + rowadd.double()[:, None, None, :] + res.double()
    e1 = ((y.double() - ref).abs().max() / ref.abs().max()).item()
    e0 = ((y0.double() - ref).abs().max() / ref.abs().max()).item()
    r1 = ((y.double() - ref).norm() / ref.norm()).item(); r0 = ((y0.double() - ref).norm() / ref.norm()).item()
    print(f"B{B} {H}x{H} {Cin}->{Cout} up={up}: wino max {e1:.2e} rel {r1:.2e} | direct max {e0:.2e} rel {r0:.2e} same={torch.equal(y, y0)}", flush=True)
# timing
def ev(fn, n=5):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]
for (B, H, Cin, Cout) in [][:0] + [(128, 8, 256, 256), (128, 16, 128, 128), (1024, 4, 256, 256), (16, 16, 1280, 1280), (16, 32, 1280, 640), (16,64,640,320), (64, 32, 320, 320), (64, 16, 640, 640), (64, 8, 1280, 1280), (32, 32, 448, 448), (32, 16, 672, 672)]:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev)
    t1 = ev(lambda: ops.conv2d_fwd_raw(x, w, b))
    with ops.kernel_flags(no_wino=True):
        t0 = ev(lambda: ops.conv2d_fwd_raw(x, w, b))
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    print(f"B{B} {H}x{H} {Cin}->{Cout}: wino {t1:.3f} ms ({fl/t1/1e9:.1f} TF/s alg) | direct {t0:.3f} ms ({fl/t0/1e9:.1f} TF/s) speedup {t0/t1:.2f}", flush=True)

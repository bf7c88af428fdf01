"""A/B: LDS-transposed float4 epilogue (default) vs dword stores straight from the accumulators
(kernel_flags(scalar_epilogue=True)) on 3x3 convolutions (patch kernels) and dense GEMMs; bit-identity checked.
usage: python tools/ab_epilogue.py [f32|bf16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch
import gad
from gad import ops
from gad._capi import A_KC, A_MC, B_KC, B_MC
gad.set_operand_precision("bf16" if "bf16" in sys.argv[1:] else "no")
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
def ab(label, fl, fn):
    res, outs = [], []
    for name, flags in (("float4", {}), ("scalar", {"scalar_epilogue": True}), ("float4", {})):
        with ops.kernel_flags(**flags):
            outs.append(fn().clone())
            ms = timeit(fn)
        res.append(f"{name} {fl / ms / 1e9:6.1f}")
    print(f"{label}: " + " | ".join(res) + f" TF/s | bit-identical {torch.equal(outs[0], outs[1])}", flush=True)
B = 1024
for Cin, Cout, H in ((128, 128, 32), (256, 128, 32), (384, 128, 32), (256, 256, 16), (512, 256, 16), (256, 256, 8), (256, 256, 4), (96, 96, 32), (320, 320, 32)):
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev); temb = torch.randn(B, Cout, device=dev); res = torch.randn(B, H, H, Cout, device=dev)
    dy = torch.randn(B, H, H, Cout, device=dev)
    fl = 2.0 * B * H * H * Cout * Cin * 9
    ab(f"conv3x3 fwd   B={B} {Cin}->{Cout}@{H}", fl, lambda: ops.conv2d_fwd_raw(x, w, b))
    ab(f"  +temb+resid B={B} {Cin}->{Cout}@{H}", fl, lambda: ops.conv2d_fwd_raw(x, w, b, rowadd=temb, residual=res))
    ab(f"conv3x3 dgrad B={B} {Cin}->{Cout}@{H}", fl, lambda: ops.conv2d_dgrad_raw(dy, w, (B, H, H, Cin)))
for Cin, Cout, H, k, s in ((128, 128, 32, 3, 2), (256, 256, 16, 3, 2), (512, 256, 16, 1, 1), (256, 128, 32, 1, 1)):
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev)
    pad = (0, 1, 0, 1) if s == 2 else (0, 0, 0, 0)
    Ho = H // s
    fl = 2.0 * B * Ho * Ho * Cout * Cin * k * k
    ab(f"conv{k}x{k}s{s}     B={B} {Cin}->{Cout}@{H}", fl, lambda: ops.conv2d_fwd_raw(x, w, b, stride=s, pad=pad))
for M, N, K in ((262144, 768, 256), (262144, 256, 256), (32768, 256, 256), (65536, 320, 320), (16384, 640, 640), (4096, 1280, 1280), (65536, 2560, 320)):
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; y = torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev); bias = torch.randn(N, device=dev)
    fl = 2.0 * M * N * K
    ab(f"gemm nt M={M} N={N} K={K}", fl, lambda: (ops.gemm_raw(x, w, y, A_KC, B_KC, M, N, K, K, K, N, bias=bias), y)[1])
    ab(f"gemm nn M={M} N={N} K={K}", fl, lambda: (ops.gemm_raw(dy, w, dx, A_KC, B_MC, M, K, N, N, K, K), dx)[1])
    ab(f"gemm tn M={M} N={N} K={K}", fl, lambda: (ops.gemm_raw(dy, x, dw, A_MC, B_MC, N, K, M, N, K, K), dw)[1])

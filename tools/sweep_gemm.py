"""Dense GEMM tile sweep at the SD-1.x LoRA step's shapes (B=64 @ 32x32 latents): auto plan vs forced 128 / 64 tiles for
the forward (nt), data-gradient (nn) and weight-gradient (tn) forms.  usage: python tools/sweep_gemm.py [f32|bf16]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import ctypes as C
import torch
from gad import ops, _capi
from gad._capi import A_KC, A_MC, B_KC, B_MC
dev = torch.device("cuda:0")
import gad
gad.set_operand_precision("bf16" if "bf16" in sys.argv[1:] else "no")


def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


SHAPES = [(262144, 768, 256), (262144, 256, 256), (65536, 320, 320), (65536, 256, 320), (65536, 320, 256), (65536, 2560, 320), (65536, 320, 1280),
          (16384, 640, 640), (16384, 256, 640), (16384, 640, 256), (16384, 5120, 640), (16384, 640, 2560),
          (4096, 1280, 1280), (4096, 256, 1280), (4096, 1280, 256), (4096, 10240, 1280), (4096, 1280, 5120),
          (1024, 1280, 1280), (4928, 320, 768), (4928, 1280, 768), (4928, 256, 768)]
for M, N, K in SHAPES:
    x = torch.randn(M, K, device=dev); w = torch.randn(N, K, device=dev) * 0.05; y = torch.empty(M, N, device=dev)
    dy = torch.randn(M, N, device=dev); dx = torch.empty(M, K, device=dev); dw = torch.empty(N, K, device=dev)
    fl = 2.0 * M * N * K
    line = f"M={M:6d} N={N:5d} K={K:5d}: "
    for form, fn in (("nt", lambda t: ops.gemm_raw(x, w, y, A_KC, B_KC, M, N, K, K, K, N, tile_hint=t)),
                     ("nn", lambda t: ops.gemm_raw(dy, w, dx, A_KC, B_MC, M, K, N, N, K, K, tile_hint=t)),
                     ("tn", lambda t: ops.gemm_raw(dy, x, dw, A_MC, B_MC, N, K, M, N, K, K, tile_hint=t))):
        r = []
        for t in (0, 1, 2, 3):
            ms = timeit(lambda: fn(t))
            r.append(f"{fl / ms / 1e9:5.1f}")
        line += f"{form} auto/128/64/128x64 = {'/'.join(r)} TF/s | "
    print(line, flush=True)

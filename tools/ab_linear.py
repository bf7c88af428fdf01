"""Short-K dense GEMMs (attention projections, 1x1 shortcuts): tile choice A/B."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch
from gad import ops
from gad._capi import A_KC, B_KC
dev = torch.device("cuda:0")
def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
for M, N, K in ((262144, 256, 256), (131072, 256, 256), (32768, 256, 256), (262144, 256, 512), (1048576, 128, 256), (16384, 256, 256)):
    a = torch.randn(M, K, device=dev); b = torch.randn(N, K, device=dev) * 0.05; bias = torch.randn(N, device=dev)
    c = torch.empty(M, N, device=dev); res_t = torch.randn(M, N, device=dev)
    fl = 2.0 * M * N * K
    out = []
    for tile in (0, 1, 2):
        ms = timeit(lambda: ops.gemm_raw(a, b, c, A_KC, B_KC, M, N, K, K, K, N, bias=bias, residual=res_t, ldr=N, tile_hint=tile))
        out.append(f"tile{tile}: {fl/ms/1e9:6.1f} TF/s ({ms*1e3:5.0f} us)")
    gb = (M * K + M * N * 2) * 4 / 1e9
    print(f"M={M} N={N} K={K} (min HBM {gb:.2f} GB): " + " | ".join(out), flush=True)

"""Dense GEMMs of the CIFAR sampler at B = 1024 (attention projections, 1x1 shortcuts) on the 128x128 / 128x64 / 64x64 tiles and the planner's choice."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
from gad import ops
from gad._capi import A_KC, B_KC
dev = torch.device("cuda:0")
def ev(fn, N=9):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(N):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return sorted(ts)[len(ts) // 2]
for M, N, K in [(262144, 768, 256), (262144, 256, 256), (1048576, 128, 256), (1048576, 128, 384), (262144, 256, 512), (65536, 256, 512), (1024, 4992, 512)]:
    x, w, b = torch.randn(M, K, device=dev), torch.randn(N, K, device=dev) * 0.05, torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    out = [f"M{M} N{N} K{K}:"]
    for hint, name in ((0, "auto"), (1, "128x128"), (3, "128x64"), (2, "64x64")):
        t = ev(lambda: ops.gemm_raw(x, w, y, A_KC, B_KC, M, N, K, K, K, N, bias=b, tile_hint=hint))
        out.append(f"{name} {t * 1e3:7.1f} us ({2.0 * M * N * K / t / 1e9:5.1f} TF/s)")
    print(" | ".join(out), flush=True)

"""Which 3x3 convolutions of one cifar20 slice (training step B = 128 + sampler step B = 1024) run on a Winograd F(4x4) route, and for
which of them the route's own input transform still runs (SKIP_INPUT False): forward launches whose V GroupNorm did not write, weight
gradients that did not get the forward's V.  usage (GPU box): python tools/wino_route_stats.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (puts the package on the path)
bench.torch = torch  # (bench.py imports torch in main())
from gad import ops  # noqa: E402
from gad.coalition import CoalitionEngine  # noqa: E402

dev = torch.device("cuda:0")
engine = CoalitionEngine("cifar100", device=dev, gd_steps=1000, n_samples=10240, sample_batch=bench.SAMPLE_B, fuse=bench.FUSE,
                         num_inference_steps=bench.DDIM_STEPS)
run = bench.SliceRunner(engine, removal_seed=0, in_flight=1)
run.slice()
torch.cuda.synchronize()
for phase, fn in (("training step", run.train_step), ("sampler step", run.sampler_step)):
    ops.ROUTE_STATS = {}
    fn()
    torch.cuda.synchronize()
    st, ops.ROUTE_STATS = ops.ROUTE_STATS, None
    print(f"== {phase}: kernel id (0-4 direct forms, 5 F(2x2), 6 F(4x4) forward, 7 F(4x4) weight gradient), M, N, K, V supplied, launches")
    for (kid, M, N, K, skip, grad), n in sorted(st.items()):
        print(f"  id {kid}  M {M:8d} N {N:5d} K {K:8d}  V supplied {str(skip):5s}  x{n}")
    own = sum(n for (kid, *_r, skip, _g), n in st.items() if kid in (6, 7) and not skip)
    print(f"  launches that run their own input transform: {own}")

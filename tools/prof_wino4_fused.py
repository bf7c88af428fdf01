"""Fixed workload for rocprofv3 passes over the F(4x4) Winograd forms: for each shape, 4 launches of the one-launch form on
32-tile blocks (hint 9), on 64-tile blocks (hint 11) and of the three-launch form (hint 10).  With `sum <dir>` it prints, per
kernel, the launch count and the per-launch average of every counter found in the pass.
usage: rocprofv3 --kernel-trace --pmc ... -d DIR --output-format csv -- python3 tools/prof_wino4_fused.py ; python3 tools/prof_wino4_fused.py sum DIR"""
import csv, glob, os, sys
from collections import defaultdict

if len(sys.argv) > 2 and sys.argv[1] == "sum":
    per = defaultdict(lambda: defaultdict(float))
    for f in glob.glob(f"{sys.argv[2]}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            per[(r["Kernel_Name"][:60], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
    agg = defaultdict(lambda: defaultdict(list))
    for (k, _), cs in per.items():
        for c, v in cs.items():
            agg[k][c].append(v)
    for k in sorted(agg):
        if "wino4" not in k and ", 4, 1>" not in k:
            continue
        print(k)
        for c in sorted(agg[k]):
            v = agg[k][c]
            print(f"    {c:28s} n={len(v):3d} avg={sum(v) / len(v):16.1f}")
    sys.exit(0)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
from gad import ops
dev = torch.device("cuda:0")
shapes = [(1024, 32, 128, 128)] if os.environ.get("ONE") else [(1024, 32, 128, 128), (1024, 16, 256, 256)]
for B, H, Cin, Cout in shapes:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev)
    res = torch.randn(B, H, H, Cout, device=dev)
    for hint in (9, 11, 10):
        for _ in range(4):
            ops.conv2d_fwd_raw(x, w, b, tile_hint=hint, residual=res)
    torch.cuda.synchronize()
    print("shape", B, H, Cin, Cout, flush=True)

"""Per-kernel durations of the F(4x4) route at chosen shapes (rocprofv3 --kernel-trace of this script, parsed by the caller):
prints the shapes in launch order so the trace rows can be matched.  usage: rocprofv3 --kernel-trace ... -- python3 tools/trace_wino4.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
from gad import ops
dev = torch.device("cuda:0")
for B, H, Cin, Cout in [(1024, 32, 128, 128), (1024, 32, 256, 128), (1024, 16, 256, 256), (1024, 16, 512, 256), (128, 32, 128, 128), (16, 64, 320, 320), (32, 64, 224, 224)]:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    for _ in range(3):
        ops.conv2d_fwd_raw(x, w, None)
    torch.cuda.synchronize()
    print("shape", B, H, Cin, Cout, flush=True)

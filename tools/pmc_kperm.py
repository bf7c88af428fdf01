import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
from gad import ops
dev = torch.device("cuda:0")
for B, Cin, Cout, H in [(128, 256, 256, 16), (512, 128, 128, 32)]:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev)
    for tap_major in (True, False):
        with ops.kernel_flags(no_patch=True, tap_major_k=tap_major):
            for _ in range(6):
                y = ops.conv2d_fwd_raw(x, w, b, 1, (1, 1, 1, 1), False, tile_hint=1)
        torch.cuda.synchronize()

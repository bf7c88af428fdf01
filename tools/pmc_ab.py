import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
from gad import ops
dev = torch.device("cuda:0")
B, Cin, Cout, H = 128, 256, 256, 16
x = torch.randn(B, H, H, Cin, device=dev)
w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
b = torch.randn(Cout, device=dev)
for _ in range(30):
    y = ops.conv2d_fwd_raw(x, w, b, 1, (1, 1, 1, 1), False, tile_hint=1)
torch.cuda.synchronize()

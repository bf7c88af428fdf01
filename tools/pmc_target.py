"""Small fixed workload for rocprofv3 --pmc passes: the dominant conv-forward instance
(gemm_kernel<A_CONV,B_KC,128,128,4>) on two CIFAR U-Net shapes at B=128, 10 launches each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
from gad import ops
dev = torch.device("cuda:0")
B = 128
for (Cin, Cout, H) in [(128, 128, 32), (256, 256, 16)]:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    b = torch.randn(Cout, device=dev)
    for _ in range(10):
        y = ops.conv2d_fwd_raw(x, w, b, 1, (1, 1, 1, 1), False)
torch.cuda.synchronize()
print("ok")

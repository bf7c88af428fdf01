import sys
sys.path.insert(0, "/root/repo/group-attribution-for-diffusion-models_amd"); sys.path.insert(0, "/root/repo")
import torch
from gad import ops
dev = torch.device("cuda:0")
for (B, H, Cin, Cout) in [(1024, 32, 256, 256)]:
    x = torch.randn(B, H, H, Cin, device=dev)
    w = (torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    for hint in (8,):
        for _ in range(3):
            ops.conv2d_fwd_raw(x, w, None, tile_hint=hint)
    with ops.kernel_flags(no_wino=True):
        for _ in range(3):
            ops.conv2d_fwd_raw(x, w, None)
    torch.cuda.synchronize()

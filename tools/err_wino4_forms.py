"""Error of every F(4x4) form against an fp64 convolution (unit-scale outputs): max |err| and rms err."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
import torch.nn.functional as F
from gad import ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
for B, H, Cin, Cout in [(8, 32, 128, 128), (4, 16, 512, 256), (8, 32, 64, 128)]:
    x = torch.randn(B, Cin, H, H)
    w = torch.randn(Cout, Cin, 3, 3) / math.sqrt(9 * Cin)
    ref = F.conv2d(x.double(), w.double(), padding=1)
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wg = w.to(dev).contiguous(memory_format=torch.channels_last)
    out = [f"B{B} {H}x{H} {Cin}->{Cout}:"]
    for hint, flags in ((9, {}), (11, {}), (10, {}), (7, {}), (0, dict(no_wino=True))):
        with ops.kernel_flags(**flags):
            y = ops.conv2d_fwd_raw(xg, wg, None, tile_hint=hint).permute(0, 3, 1, 2).cpu().double()
        e = y - ref
        out.append(f"h{hint}{'d' if flags else ''} max {e.abs().max().item():.2e} rms {e.pow(2).mean().sqrt().item():.2e}")
    print(" | ".join(out), flush=True)

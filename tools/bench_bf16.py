"""fp32-operand vs bf16-operand contraction kernels on the sampler's conv shapes (B=512) and the whole U-Net forward."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch

import gad
from gad import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "conv"):
        B = 512
        for Cin, Cout, H, k in ((128, 128, 32, 3), (256, 128, 32, 3), (384, 128, 32, 3), (256, 256, 16, 3), (512, 256, 16, 3),
                                (256, 256, 8, 3), (384, 128, 32, 1), (256, 256, 16, 1)):
            x = torch.randn(B, H, H, Cin, device=dev)
            w = (torch.randn(Cout, Cin, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
            b = torch.randn(Cout, device=dev)
            pad = (1, 1, 1, 1) if k == 3 else (0, 0, 0, 0)
            fl = 2.0 * B * H * H * Cout * Cin * k * k
            res = []
            for prec, tile, ng in (("f32", 1, True), ("bf16", 1, False), ("bf16", 1, True), ("bf16", 2, True)):
                with ops.operand_precision(prec), torch.set_grad_enabled(not ng):
                    ms = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, 1, pad, False, tile_hint=tile))
                tag = f"{prec} t{tile}" + (" (bf16 weight copy)" if prec == "bf16" and ng and tile == 1 else "")
                res.append(f"{tag}: {fl/ms/1e9:6.0f} TF/s ({ms*1e3:6.0f} us)")
            print(f"conv B={B} {Cin}->{Cout}@{H} k{k}: " + " | ".join(res), flush=True)
    if which in ("all", "unet"):
        from src.ddpm_config import DDPMConfig
        net = gad.UNet2DModel(**DDPMConfig.cifar100_config["unet_config"]).to(dev).eval()
        for B in (512, 128):
            x = torch.randn(B, 32, 32, 3, device=dev)
            t = torch.randint(0, 1000, (B,), device=dev)
            outs = {}
            for prec in ("f32", "bf16"):
                with torch.no_grad(), ops.operand_precision(prec):
                    outs[prec] = net.forward_nhwc(x, t)
                    ms = timeit(lambda: net.forward_nhwc(x, t), iters=5)
                print(f"unet fwd B={B} {prec}: {ms:.2f} ms  {12.44e9*B/ms/1e9:.0f} TF/s", flush=True)
            d = (outs["bf16"] - outs["f32"])
            print(f"   bf16 vs f32 output: max abs diff {d.abs().max().item():.3e}, rel rms {(d.norm()/outs['f32'].norm()).item():.3e}")
    if which in ("all", "sd"):
        from gad.sd import UNet2DConditionModel
        net = UNet2DConditionModel().to(dev).eval()
        B = 16
        x = torch.randn(B, 32, 32, 4, device=dev)
        t = torch.randint(0, 1000, (B,), device=dev)
        ctx = torch.randn(B, 77, 768, device=dev)
        for prec in ("f32", "bf16"):
            with torch.no_grad(), ops.operand_precision(prec):
                ms = timeit(lambda: net.forward_nhwc(x, t, ctx), iters=3, warm=1)
            print(f"SD unet fwd B={B} 32x32 latents {prec}: {ms:.1f} ms", flush=True)


if __name__ == "__main__":
    main()

"""Per-shape micro-benchmark of the contraction engine on the CIFAR U-Net layer shapes
(run on the GPU box).  Prints TF/s per (shape, tile, splitk) and the U-Net forward time."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch

import gad
from gad import ops

dev = torch.device("cuda:0")


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters  # ms


def conv_shapes():
    # (Cin, Cout, H, k, stride, pad, ups)
    return [(128, 128, 32, 3, 1, (1, 1, 1, 1), False), (128, 256, 16, 3, 1, (1, 1, 1, 1), False),
            (256, 256, 16, 3, 1, (1, 1, 1, 1), False), (256, 256, 8, 3, 1, (1, 1, 1, 1), False),
            (256, 256, 4, 3, 1, (1, 1, 1, 1), False), (512, 256, 8, 3, 1, (1, 1, 1, 1), False),
            (512, 256, 16, 3, 1, (1, 1, 1, 1), False), (384, 256, 16, 3, 1, (1, 1, 1, 1), False),
            (384, 128, 32, 3, 1, (1, 1, 1, 1), False), (256, 128, 32, 3, 1, (1, 1, 1, 1), False),
            (256, 256, 16, 3, 1, (1, 1, 1, 1), True), (128, 128, 32, 3, 2, (0, 1, 0, 1), False),
            (384, 128, 32, 1, 1, (0, 0, 0, 0), False)]


def main():
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which in ("all", "conv"):
        for B in (32, 128):
            for (Cin, Cout, H, k, stride, pad, ups) in conv_shapes():
                x = torch.randn(B, H, H, Cin, device=dev)
                w = torch.randn(Cout, Cin, k, k, device=dev).contiguous(memory_format=torch.channels_last) * 0.05
                b = torch.randn(Cout, device=dev)
                Ho = (2 * H if ups else H) // stride
                fl = 2.0 * B * Ho * Ho * Cout * Cin * k * k
                res = []
                for tile, sk in ((1, 1), (2, 1), (2, 2), (2, 4), (0, 0)):
                    try:
                        ms = timeit(lambda: ops.conv2d_fwd_raw(x, w, b, stride, pad, ups, tile_hint=tile, splitk_hint=sk))
                        res.append(f"t{tile}s{sk}:{fl / ms / 1e9:6.1f}TF({ms*1e3:6.0f}us)")
                    except Exception as ex:  # noqa
                        res.append(f"t{tile}s{sk}:ERR")
                print(f"fwd B={B:3d} {Cin:3d}->{Cout:3d}@{H:2d} k{k} s{stride} u{int(ups)}  " + "  ".join(res), flush=True)
        # backward kernels at B=128
        for (Cin, Cout, H, k, stride, pad, ups) in conv_shapes()[:6]:
            B = 128
            x = torch.randn(B, H, H, Cin, device=dev)
            w = torch.randn(Cout, Cin, k, k, device=dev).contiguous(memory_format=torch.channels_last) * 0.05
            Ho = H // stride
            dy = torch.randn(B, Ho, Ho, Cout, device=dev)
            fl = 2.0 * B * Ho * Ho * Cout * Cin * k * k
            ms_d = timeit(lambda: ops.conv2d_dgrad_raw(dy, w, x.shape, stride, pad, ups))
            ms_w = timeit(lambda: ops.conv2d_wgrad_raw(dy, x, w, stride, pad, ups))
            print(f"bwd B={B} {Cin}->{Cout}@{H}: dgrad {fl/ms_d/1e9:6.1f}TF ({ms_d*1e3:.0f}us)  wgrad {fl/ms_w/1e9:6.1f}TF ({ms_w*1e3:.0f}us)", flush=True)
    if which in ("all", "unet"):
        from src.ddpm_config import DDPMConfig
        cfg = dict(DDPMConfig.cifar100_config["unet_config"])
        net = gad.UNet2DModel(**cfg).to(dev)
        for B in (32, 128):
            x = torch.randn(B, 32, 32, 3, device=dev)
            t = torch.randint(0, 1000, (B,), device=dev)
            with torch.no_grad():
                ms = timeit(lambda: net.forward_nhwc(x, t), iters=10)
            fl = 12.44e9 * B
            print(f"unet fwd eager B={B}: {ms:.2f} ms  {fl/ms/1e9:.1f} TF/s", flush=True)
            with torch.no_grad():
                g = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    net.forward_nhwc(x, t)
                torch.cuda.current_stream().wait_stream(s)
                with torch.cuda.graph(g):
                    y = net.forward_nhwc(x, t)
                ms = timeit(lambda: g.replay(), iters=10)
            print(f"unet fwd graph B={B}: {ms:.2f} ms  {fl/ms/1e9:.1f} TF/s", flush=True)
        # training step
        sch = gad.DDPMScheduler()
        net2 = gad.UNet2DModel(**cfg).to(dev)
        ema = gad.EMAModel(net2.parameters())
        tr = gad.FusedTrainer(net2, sch, ema)
        B = 128
        img, noise = torch.randn(B, 3, 32, 32, device=dev), torch.randn(B, 3, 32, 32, device=dev)
        ts = torch.randint(0, 1000, (B,), device=dev)
        ms = timeit(lambda: tr.step(img, noise, ts), iters=5, warm=2)
        print(f"train step eager B={B}: {ms:.2f} ms  {3*12.44e9*B/ms/1e9:.1f} TF/s  loss {tr.last_loss.item():.4f}", flush=True)


if __name__ == "__main__":
    main()

"""Sampler throughput experiments: fused batch size and two-stream overlap."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch
import gad
from src.ddpm_config import DDPMConfig
dev = torch.device("cuda:0")
cfg = dict(DDPMConfig.cifar100_config["unet_config"])
net = gad.UNet2DModel(**cfg).to(dev).eval()
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(iters): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / iters
with torch.no_grad():
    for B in (512, 1024, 2048):
        x = torch.randn(B, 32, 32, 3, device=dev); t = torch.randint(0, 1000, (B,), device=dev)
        ms = timeit(lambda: net.forward_nhwc(x, t)) * 1e3
        print(f"fwd B={B}: {ms:.2f} ms  {ms/B*1e3:.1f} us/img  {12.44e9*B/ms/1e9:.1f} TF/s", flush=True)
    # two independent B=128 groups on two streams
    B = 128
    xs = [torch.randn(B, 32, 32, 3, device=dev) for _ in range(2)]
    ts = [torch.randint(0, 1000, (B,), device=dev) for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def two():
        for s, x, t in zip(streams, xs, ts):
            with torch.cuda.stream(s):
                net.forward_nhwc(x, t)
    for s in streams: s.wait_stream(torch.cuda.current_stream())
    ms = timeit(two) * 1e3
    print(f"2 streams x B=128: {ms:.2f} ms per pair  {ms/256*1e3:.1f} us/img", flush=True)

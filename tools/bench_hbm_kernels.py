"""Achieved HBM GB/s of the bandwidth-bound kernels at the workload's own sizes (run on the GPU box).
bytes = algorithmic bytes (GN+SiLU fwd 8 B/elem, bwd 16 B/elem (+4 if SiLU recompute reads x), DDIM step
12 B/elem, add_noise 12 B/elem, clip+Adam+EMA 36 B/param (+4 B/param for the sum-of-squares pass)); time = HIP events
over `iters` back-to-back launches; peak = 8 TB/s (MI355X_MICROARCH.md).  Tensors smaller than the 256 MB Infinity
Cache stay cache-resident between launches, so their figure is a cache rate - flagged in the output."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd"))
sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch

from gad import ops

dev = torch.device("cuda:0")
PEAK = 8000.0


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
    return s.elapsed_time(e) / iters * 1e-3


def report(name, nbytes, sec, footprint):
    gbs = nbytes / sec / 1e9
    tag = "" if footprint > 256e6 else "  (footprint %.0f MB < Infinity Cache: cache-resident rate)" % (footprint / 1e6)
    print(f"{name:58s} {nbytes/1e6:9.1f} MB {sec*1e6:9.1f} us {gbs:8.0f} GB/s  {gbs/PEAK:5.2f} of 8 TB/s{tag}", flush=True)


def main():
    # GroupNorm + SiLU at the sampler width (B=512) and the training batch (B=128), the U-Net's (C, H) levels
    for B in (512, 128):
        for C, H in ((128, 32), (384, 32), (256, 16), (512, 16), (256, 8)):
            x = torch.randn(B, H, H, C, device=dev, requires_grad=True)
            g = torch.randn(C, device=dev, requires_grad=True)
            b = torch.randn(C, device=dev, requires_grad=True)
            n = x.numel()
            with torch.no_grad():
                t = timeit(lambda: ops.group_norm(x, g, b, 32, 1e-6, True))
            report(f"groupnorm+silu fwd  B={B} C={C} {H}x{H}", 8 * n, t, 8 * n)
            y = ops.group_norm(x, g, b, 32, 1e-6, True)
            dy = torch.randn_like(y)
            t = timeit(lambda: torch.autograd.grad(y, (x, g, b), dy, retain_graph=True))
            report(f"groupnorm+silu bwd  B={B} C={C} {H}x{H}", 16 * n, t, 16 * n)
            del x, y, dy
    # scheduler / loss kernels on the sampler's x_t and the training batch
    for B in (512, 128):
        x = torch.randn(B, 32, 32, 3, device=dev)
        e = torch.randn_like(x)
        t = timeit(lambda: ops.ddim_step_raw(x, e, 0.5, 0.6, 1.0, out=x), iters=200)
        report(f"ddim_step           B={B} [B,32,32,3]", 12 * x.numel(), t, 12 * x.numel())
    x0 = torch.randn(128, 3, 32, 32, device=dev)
    eps = torch.randn_like(x0)
    ts = torch.randint(0, 1000, (128,), device=dev)
    ac = torch.linspace(0.9999, 0.0001, 1000, device=dev)
    t = timeit(lambda: ops.add_noise_raw(x0, eps, ts, ac), iters=200)
    report("add_noise           B=128 [B,3,32,32]", 12 * x0.numel(), t, 12 * x0.numel())
    t = timeit(lambda: ops.mse_fwd_bwd_raw(x0, eps), iters=200)
    report("mse fwd+bwd         B=128 [B,3,32,32]", 12 * x0.numel(), t, 12 * x0.numel())
    # fused clip + Adam + EMA over the 35.75 M-parameter flat buffers (and the 51 M LoRA parameters of SD r=256)
    for n, label in ((35_746_308, "CIFAR U-Net 35.75 M"), (51_019_776, "SD LoRA r=256 51.0 M")):
        p, g, m, v, s = (torch.randn(n, device=dev) * 0.01 for _ in range(5))
        v.abs_()
        ssq = torch.zeros(1, device=dev)

        def step():
            ops.sumsq_raw(g, out=ssq)
            ops.clip_adam_ema_raw(p, g, m, v, s, ssq, max_norm=1.0, lr=1e-4, betas=(0.9, 0.999), eps=1e-8,
                                  weight_decay=0.0, adamw=False, step=10, ema_decay=0.999)
        t = timeit(step, iters=30)
        report(f"sumsq + clip+adam+ema  {label}", 40 * n, t, 36 * n)
    # transformer-block kernels at SD-1.x sizes (B=64, 32x32 latents -> 1024 tokens x 320; 256 x 640)
    for rows, C in ((64 * 1024, 320), (64 * 256, 640), (64 * 64, 1280)):
        x = torch.randn(rows, C, device=dev)
        g, b = torch.randn(C, device=dev), torch.randn(C, device=dev)
        with torch.no_grad():
            t = timeit(lambda: ops.layer_norm(x, g, b))
        report(f"layernorm fwd       rows={rows} C={C}", 8 * x.numel(), t, 8 * x.numel())
        h = torch.randn(rows, 8 * C, device=dev)
        with torch.no_grad():
            t = timeit(lambda: ops.geglu(h))
        report(f"geglu fwd           rows={rows} C={4*C}", 12 * rows * 4 * C, t, 12 * rows * 4 * C)


if __name__ == "__main__":
    main()

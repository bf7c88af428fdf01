"""Per-SHAPE timing of the contraction launches of one CIFAR-20 slice (1 sampler step at B = 1024 + 1 training step at
B = 128): which (family, tile, split, M, N, K, geometry) launches the time of the non-patch families goes to.
usage: python tools/prof_shapes.py [pruned] [iters]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import ctypes as C
import torch
import gad
from gad import ops
from gad._capi import A_CONV, A_CONVT, check
from src.ddpm_config import DDPMConfig

dev = torch.device("cuda:0")
pruned = "pruned" in sys.argv[1:]
iters = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 3
cfg = dict(DDPMConfig.cifar100_config["unet_config"])
if pruned:
    cfg["block_out_channels"] = [96, 192, 192, 192]


class ShapeProfiler(ops.GemmProfiler):
    def gemm(self, lib, a, batch):
        n0 = len(self.records)
        super().gemm(lib, a, batch)
        key, fl, nb, s, e = self.records[n0]
        g = a.g
        geo = f"{g.KH}x{g.KW}s{g.stride}{'u' if g.upsample else ''} {g.H}x{g.W}x{g.C}" if a.a_mode in (A_CONV, A_CONVT) or a.b_mode == 3 else ""
        self.records[n0] = (key + (a.M, a.N, a.K, max(1, batch), geo, "2src" if a.A2 else ""), fl, nb, s, e)


net = gad.UNet2DModel(**cfg).to(dev)
sched = gad.DDPMScheduler(num_train_timesteps=1000)
trainer = gad.FusedTrainer(net, sched, None, lr=1e-4, max_grad_norm=1.0)
xs = torch.randn(1024, 32, 32, 3, device=dev); ts = torch.randint(0, 1000, (1024,), device=dev)
img = torch.randn(128, 3, 32, 32, device=dev); tt = torch.randint(0, 1000, (128,), device=dev)


def one():
    net.eval()
    with torch.no_grad():
        net.forward_nhwc(xs, ts)
    net.train()
    trainer.step(img, torch.randn_like(img), tt)


for _ in range(2):
    one()
torch.cuda.synchronize()
ops.PROFILER = prof = ShapeProfiler()
for _ in range(iters):
    one()
torch.cuda.synchronize()
ops.PROFILER = None
rows = sorted(prof.summary().items(), key=lambda kv: -kv[1]["ms"])
tot = sum(v["ms"] for _, v in rows)
print(f"total contraction time {tot / iters:.2f} ms per slice")
for k, v in rows[:70]:
    if len(k) < 10:
        print(f"{v['ms'] / iters:7.3f} ms  n={v['launches'] // iters:3d}  {v['flops'] / v['ms'] / 1e9:6.1f} TF/s  {k}", flush=True)
        continue
    name, tile, sk, vec, M, N, K, batch, geo, src = k
    print(f"{v['ms'] / iters:7.3f} ms  n={v['launches'] // iters:3d}  {v['flops'] / v['ms'] / 1e9:6.1f} TF/s  {name:24s} t{tile} sk{sk} v{vec} "
          f"M={M} N={N} K={K} b={batch} {geo} {src}", flush=True)

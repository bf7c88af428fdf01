"""CelebA-HQ LDM U-Net (config 3: UNet2DModel 224/448/672/896 on 3x64x64 latents): training step at the reference batch
(B=32; reference 1.32 s/step) and sampler forward."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch
import gad
from src.ddpm_config import DDPMConfig
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
gad.set_operand_precision(prec)
cfg = DDPMConfig.celeba_config
net = gad.UNet2DModel(**cfg["unet_config"]).to(dev)
npar = sum(p.numel() for p in net.parameters())
print(f"CelebA LDM U-Net: {npar/1e6:.1f} M parameters, operand precision {prec}", flush=True)
def timeit(fn, n=5, w=2):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n
for B in (32, 128):
    x = torch.randn(B, 64, 64, 3, device=dev); t = torch.randint(0, 1000, (B,), device=dev)
    with torch.no_grad():
        ms = timeit(lambda: net.forward_nhwc(x, t)) * 1e3
    print(f"forward B={B}: {ms:.1f} ms = {ms/B:.2f} ms/image", flush=True)
sch = gad.DDPMScheduler(**{k: v for k, v in cfg["scheduler_config"].items() if k in ("beta_start", "beta_end", "beta_schedule", "num_train_timesteps")})
tr = gad.FusedTrainer(net, sch, gad.EMAModel(net.parameters()), lr=1e-4, adamw=True)
B = 32
x, n = torch.randn(B, 3, 64, 64, device=dev), torch.randn(B, 3, 64, 64, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
s = timeit(lambda: tr.step(x, n, t))
print(f"training step B={B} (fwd+bwd+clip+AdamW+EMA): {s*1e3:.1f} ms = {1/s:.2f} steps/s (reference: 1.32 s/step)", flush=True)

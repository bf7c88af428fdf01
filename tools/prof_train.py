"""Training steps only (CIFAR-20 U-Net, B = 128, FusedTrainer) - run under rocprofv3 --kernel-trace --stats to see where
the training half of the slice goes.  usage: python tools/prof_train.py [steps] [pruned]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch
import gad
from src.ddpm_config import DDPMConfig
dev = torch.device("cuda:0")
steps = int([a for a in sys.argv[1:] if a.isdigit()][0]) if any(a.isdigit() for a in sys.argv[1:]) else 20
cfg = dict(DDPMConfig.cifar100_config["unet_config"])
if "pruned" in sys.argv[1:]:
    cfg["block_out_channels"] = [96, 192, 192, 192]
net = gad.UNet2DModel(**cfg).to(dev).train()
ema = gad.EMAModel(net.parameters(), decay=0.9999, use_ema_warmup=False, inv_gamma=1.0, power=0.75)
trainer = gad.FusedTrainer(net, gad.DDPMScheduler(num_train_timesteps=1000), ema, lr=1e-4, max_grad_norm=1.0)
img = torch.randn(128, 3, 32, 32, device=dev); tt = torch.randint(0, 1000, (128,), device=dev)
for _ in range(3):
    trainer.step(img, torch.randn_like(img), tt)
torch.cuda.synchronize(); t0 = time.time()
for _ in range(steps):
    trainer.step(img, torch.randn_like(img), tt)
torch.cuda.synchronize()
print(f"{(time.time() - t0) / steps * 1e3:.2f} ms per training step", flush=True)

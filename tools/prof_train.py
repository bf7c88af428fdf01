import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch, gad
from src.ddpm_config import DDPMConfig
dev = torch.device("cuda:0")
cfg = dict(DDPMConfig.cifar100_config["unet_config"])
net = gad.UNet2DModel(**cfg).to(dev)
ema = gad.EMAModel(net.parameters())
tr = gad.FusedTrainer(net, gad.DDPMScheduler(), ema)
B = 128
img, noise = torch.randn(B, 3, 32, 32, device=dev), torch.randn(B, 3, 32, 32, device=dev)
ts = torch.randint(0, 1000, (B,), device=dev)
for _ in range(6):
    tr.step(img, noise, ts)
torch.cuda.synchronize()

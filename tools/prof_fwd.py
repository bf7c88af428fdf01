import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch, gad
from src.ddpm_config import DDPMConfig
dev = torch.device("cuda:0")
net = gad.UNet2DModel(**dict(DDPMConfig.cifar100_config["unet_config"])).to(dev).eval()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
x = torch.randn(B, 32, 32, 3, device=dev); t = torch.randint(0, 1000, (B,), device=dev)
with torch.no_grad():
    for _ in range(6):
        net.forward_nhwc(x, t)
torch.cuda.synchronize()

"""rocprofv3 target: U-Net forward at the sampler width in bf16-operand mode (5 iterations)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch
import gad
from gad import ops
from src.ddpm_config import DDPMConfig
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16"
net = gad.UNet2DModel(**DDPMConfig.cifar100_config["unet_config"]).to(dev).eval()
x = torch.randn(512, 32, 32, 3, device=dev); t = torch.randint(0, 1000, (512,), device=dev)
with torch.no_grad(), ops.operand_precision(prec):
    for _ in range(5):
        net.forward_nhwc(x, t)
torch.cuda.synchronize()

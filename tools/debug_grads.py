import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch, torch.nn.functional as F
import gad
from gad import ops
from oracle import diffusers_ref as R
from src.ddpm_config import DDPMConfig
dev = torch.device("cuda:0")
cfg = dict(DDPMConfig.cifar100_config["unet_config"])
torch.manual_seed(0)
ref = R.UNet2DModel(**cfg); net = gad.UNet2DModel(**cfg); net.load_state_dict(ref.state_dict()); net.to(dev)
g = torch.Generator().manual_seed(1)
x, noise = torch.randn(2, 3, 32, 32, generator=g), torch.randn(2, 3, 32, 32, generator=g)
t = torch.tensor([7, 950])
want = ref(x, t).sample
F.mse_loss(want, noise).backward()
got = net(x.to(dev), t.to(dev)).sample
loss, d = ops.mse_fwd_bwd_raw(got.contiguous(), noise.to(dev))
got.backward(d)
gref = dict(ref.named_parameters())
for n, p in net.named_parameters():
    a, b = p.grad.detach().cpu().double(), gref[n].grad.double()
    rel = ((a - b).norm() / (b.norm() + 1e-12)).item()
    if rel > 1e-3:
        print(f"{n:60s} rel {rel:.3e}  |ref| {b.norm():.3e} |got| {a.norm():.3e}")
print("done")

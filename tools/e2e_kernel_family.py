"""How fast do two kernel families (LDS-patch vs im2col-gather convolutions: same products, different fp32 summation
order) drift apart along a DDIM trajectory?  With RANDOM-INIT weights the sampler is a chaotic map (the untrained
U-Net amplifies perturbations), so this measures the sensitivity of the trajectory, not an error of either kernel: the
per-step difference starts at fp32 rounding level and grows geometrically.  For reference the same is done for two
different split-K choices of the SAME generic kernel family."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
os.environ.setdefault("GAD_OUTDIR", "/tmp/_out")
import torch, gad
from gad import ops
from src.ddpm_config import DDPMConfig
dev = torch.device("cuda:0")
torch.manual_seed(0)
net = gad.UNet2DModel(**DDPMConfig.cifar100_config["unet_config"]).to(dev).eval()
sch = gad.DDIMScheduler()
sch.set_timesteps(100)
x0 = torch.randn(256, 32, 32, 3, device=dev)
t = torch.empty(256, device=dev, dtype=torch.int64)

def run(flags):
    x = x0.clone()
    snaps = {}
    with torch.no_grad(), ops.kernel_flags(**flags):
        for i, ts in enumerate(sch.timesteps.tolist()):
            t.fill_(ts)
            eps = net.forward_nhwc(x, t)
            a_t, a_p = sch.step_coefficients(ts)
            ops.ddim_step_raw(x, eps, a_t, a_p, 1.0, out=x)
            if i + 1 in (1, 2, 5, 10, 20, 50, 100):
                snaps[i + 1] = x.clone()
    return snaps

base = run({})
for name, env in (("patch vs im2col-gather kernels", {"no_patch": True}),
                  ("im2col-gather, chunk-major vs tap-major K order", None)):
    if env is None:
        a, b = run({"no_patch": True}), run({"no_patch": True, "tap_major_k": True})
    else:
        a, b = base, run(env)
    print(name + ": max |x_a - x_b| after k steps: " + ", ".join(f"k={k}: {(a[k] - b[k]).abs().max().item():.2e}" for k in sorted(a)))

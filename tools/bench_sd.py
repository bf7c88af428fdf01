"""SD-1.x UNet2DConditionModel LoRA training-step / forward timing (config 4/5 shapes: B=64, 4x32x32 latents, ctx 77x768)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "group-attribution-for-diffusion-models_amd")); sys.path.insert(0, ROOT)
import torch
import gad
from gad import ops
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
PREC = sys.argv[2] if len(sys.argv) > 2 else "f32"
LAT = int(sys.argv[3]) if len(sys.argv) > 3 else 32      # latent edge: 32 = 256x256 images (the reference's runs), 64 = 512x512
gad.set_operand_precision(PREC)
print(f"operand precision: {PREC}", flush=True)
net = gad.UNet2DConditionModel().to(dev)
lora = net.inject_lora(rank=256)
x, ctx = torch.randn(B, 4, LAT, LAT, device=dev), torch.randn(B, 77, 768, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
noise = torch.randn_like(x)
sched = gad.DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000)
trainer = gad.FusedTrainer(net, sched, None, lr=3e-4, adamw=True, weight_decay=1e-2, max_grad_norm=1.0, params=lora)
def step():
    return trainer.step(x, noise, t, ctx)     # add_noise + fwd + mse + bwd + clip + AdamW, as the entry point runs it
def timeit(fn, n=3, w=1):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time() - t0) / n
with torch.no_grad():
    tf = timeit(lambda: net(x, t, ctx).sample)
print(f"SD unet fwd B={B} latents {LAT}x{LAT}: {tf*1e3:.1f} ms", flush=True)
prof = ops.GemmProfiler(); ops.PROFILER = prof
ts = timeit(step, n=2, w=1)
ops.PROFILER = None
torch.cuda.synchronize()
print(f"SD LoRA train step (fwd+bwd, r=256) B={B} latents {LAT}x{LAT}: {ts*1e3:.1f} ms = {1/ts:.2f} steps/s = {B/ts:.1f} images/s", flush=True)
summ = prof.summary()
tot = sum(v['ms'] for v in summ.values())
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]['ms'])[:14]:
    print(f"  {k}: {v['ms']/3:.1f} ms/step  {v['flops']/v['ms']/1e9:.1f} TF/s  launches {v['launches']//3}")
print("  contraction total ms/step", tot / 3)

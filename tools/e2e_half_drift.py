"""End-to-end statement for the half-precision activation path: the SAME SD-1.x LoRA fine-tuning run (full-size U-Net, same seeds, same batches,
same noise and timesteps) in fp32 and in bf16 activations - per-step losses and the trained LoRA matrices side by side.
Run on the GPU box: python tools/e2e_half_drift.py > gpurun_out/r04_half_path_drift.txt"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "group-attribution-for-diffusion-models_amd"))
import torch
import gad
from gad.coalition import seed_everything

dev = torch.device("cuda:0")
STEPS, B, LAT, RANK = 40, 8, 32, 16


def run(precision):
    gad.set_operand_precision(precision)
    seed_everything(0)
    with torch.device(dev):
        net = gad.UNet2DConditionModel(sample_size=LAT)
    net.to(dev)
    lora = net.inject_lora(rank=RANK)
    sched = gad.DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000)
    tr = gad.FusedTrainer(net, sched, None, lr=1e-3, weight_decay=1e-6, adamw=True, max_grad_norm=1.0, params=lora,
                          lr_schedule=gad.lr_lambda("cosine", STEPS, 0))
    g = torch.Generator(device=dev).manual_seed(1)
    lat = torch.randn(64, 4, LAT, LAT, device=dev, generator=g) * 0.8
    txt = torch.randn(64, 77, 768, device=dev, generator=g) * 0.5
    losses = []
    for i in range(STEPS):
        sel = torch.arange(i * B, (i + 1) * B, device=dev) % 64
        x0 = lat.index_select(0, sel)
        noise = torch.randn(x0.shape, device=dev, generator=g)
        ts = torch.randint(0, 1000, (B,), device=dev, generator=g).long()
        losses.append(float(tr.step(x0, noise, ts, txt.index_select(0, sel)).item()))
    gad.set_operand_precision("no")
    return losses, tr.flat.detach().clone()


l32, w32 = run("no")
l16, w16 = run("bf16")
print(f"SD-1.x U-Net (859.5 M frozen), LoRA r = {RANK}, B = {B} at {LAT}x{LAT} latents, {STEPS} AdamW steps, lr 1e-3 cosine; same seeds / batches / noise")
print("step  loss fp32   loss bf16-activations   relative difference")
for i, (a, b) in enumerate(zip(l32, l16)):
    if i < 5 or i % 5 == 4:
        print(f"{i:4d}  {a:.6f}   {b:.6f}   {abs(a - b) / a:.2e}")
rel = [abs(a - b) / a for a, b in zip(l32, l16)]
print(f"max relative loss difference over {STEPS} steps: {max(rel):.3e}; mean: {sum(rel) / len(rel):.3e}")
d = (w16 - w32).norm() / (w32.norm() + 1e-30)
cos = torch.dot(w16, w32) / (w16.norm() * w32.norm())
print(f"trained LoRA buffer ({w32.numel()} values): relative difference {d.item():.3e}, cosine {cos.item():.6f}")

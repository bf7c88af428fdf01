"""DDPMPipeline / DDIMPipeline with the call signature the reference uses
(src/diffusion_utils.py:336-341,404-412; unconditional_generation/unlearn.py:761-765).

The denoising loop keeps x_t resident in HBM as NHWC; the U-Net forward plus the fused
DDIM update of one step are captured once into a hipGraph per (model, batch) and
replayed for every timestep and every batch (launch-bound otherwise: ~250 kernel
launches per step).  The initial noise is drawn exactly like diffusers' randn_tensor: a
CPU generator draws on the host and the tensor is moved (bit-exact noise parity)."""
from __future__ import annotations

from types import SimpleNamespace

import torch

from . import ops
from .schedulers import DDIMScheduler


class DDPMPipeline:
    def __init__(self, unet, scheduler):
        self.unet, self.scheduler = unet, scheduler
        self.vqvae = None
        self.device = unet.device
        self._graphs = {}
        self.use_graph = True

    def to(self, device):
        self.unet.to(device)
        self.device = torch.device(device)
        return self

    def _decode(self, x_nhwc):
        return x_nhwc

    def _run_steps(self, x, num_inference_steps, generator=None):
        """x: NHWC device tensor, updated in place through all timesteps."""
        sch = self.scheduler
        if not isinstance(sch, DDIMScheduler):
            # any other scheduler with the diffusers step() protocol (DDPMScheduler: ancestral sampling).  Off the
            # reference's path (its samplers are DDIM); the step sees NCHW views so that its noise draws keep diffusers' order
            tdev = torch.empty(x.shape[0], device=x.device, dtype=torch.int64)
            for t in sch.timesteps.tolist():
                tdev.fill_(t)
                eps = self.unet.forward_nhwc(x, tdev)
                nxt = sch.step(eps.permute(0, 3, 1, 2), t, x.permute(0, 3, 1, 2), generator=generator).prev_sample
                x.copy_(nxt.permute(0, 2, 3, 1))
            return x
        clip = float(sch.config.clip_sample_range) if sch.config.clip_sample else 0.0
        B = x.shape[0]
        tdev = torch.empty(B, device=x.device, dtype=torch.int64)
        use_graph = self.use_graph and isinstance(sch, DDIMScheduler)
        if not use_graph:
            for t in sch.timesteps.tolist():
                tdev.fill_(t)
                eps = self.unet.forward_nhwc(x, tdev)
                a_t, a_p = sch.step_coefficients(t)
                ops.ddim_step_raw(x, eps, a_t, a_p, clip, out=x)
            return x
        # one captured graph per distinct (a_t, a_p) would defeat the purpose: capture the U-Net
        # forward once (t is a device tensor) and launch the tiny DDIM kernel eagerly after it.
        key = (B, tuple(x.shape[1:]))
        g = self._graphs.get(key)
        if g is None:
            g = SimpleNamespace(x=torch.zeros_like(x), t=torch.zeros_like(tdev), eps=None, graph=None)
            side = torch.cuda.Stream(device=x.device)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                         # warm-up outside capture (workspace alloc)
                self.unet.forward_nhwc(g.x, g.t)
            torch.cuda.current_stream().wait_stream(side)
            g.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g.graph):
                g.eps = self.unet.forward_nhwc(g.x, g.t)
            self._graphs[key] = g
        g.x.copy_(x)
        ops.refresh_wino(self.unet)            # derived weights the captured launches read at fixed addresses
        for t in sch.timesteps.tolist():
            g.t.fill_(t)
            g.graph.replay()
            a_t, a_p = sch.step_coefficients(t)
            ops.ddim_step_raw(g.x, g.eps, a_t, a_p, clip, out=g.x)
        x.copy_(g.x)
        return x

    @torch.no_grad()
    def __call__(self, batch_size=1, generator=None, num_inference_steps=1000, output_type="pil", eta=0.0,
                 return_dict=True):
        if eta != 0.0:
            raise NotImplementedError("eta != 0")
        cfg = self.unet.config
        ss = cfg.sample_size if isinstance(cfg.sample_size, (tuple, list)) else (cfg.sample_size, cfg.sample_size)
        shape = (batch_size, cfg.in_channels, *ss)
        if generator is not None and generator.device.type == "cpu":
            noise = torch.randn(shape, generator=generator, dtype=torch.float32).to(self.device)
        else:
            noise = torch.randn(shape, generator=generator, dtype=torch.float32, device=self.device)
        self.scheduler.set_timesteps(num_inference_steps)
        x = ops.nchw_to_nhwc_raw(noise.contiguous())
        x = self._run_steps(x, num_inference_steps, generator)
        x = self._decode(x)
        img = ops.to_image01_raw(x)                               # (x/2+0.5).clamp(0,1); already NHWC
        if output_type == "tensor":
            return SimpleNamespace(images=img)
        images = img.cpu().numpy()
        if output_type == "pil":
            from PIL import Image
            images = [Image.fromarray((im * 255).round().astype("uint8").squeeze()) for im in images]
        return SimpleNamespace(images=images)


class LDMPipeline(DDPMPipeline):
    """diffusers.LDMPipeline as the reference drives it for CelebA-HQ (src/diffusion_utils.py:393-412, unlearn.py:753-765):
    the same denoising loop on 3x64x64 latents, then `latents / vqvae.config.scaling_factor` through `vqvae.decode` and the
    usual (x/2+0.5).clamp(0,1).  The VQ-VAE is any module with diffusers' decode protocol (hub-fetched in the reference,
    `CompVis/ldm-celebahq-256`); with `vqvae=None` - what `--precompute_stage reuse` leaves the reference with
    (main.py:531-533) - the pipeline returns the latents through the same post-processing."""

    def __init__(self, unet, vqvae=None, scheduler=None):
        super().__init__(unet, scheduler if scheduler is not None else DDIMScheduler())
        self.vqvae = vqvae

    def _decode(self, x_nhwc):
        if self.vqvae is None:
            return x_nhwc
        lat = ops.nhwc_to_nchw_raw(x_nhwc) / float(getattr(self.vqvae.config, "scaling_factor", 1.0))
        img = self.vqvae.decode(lat).sample
        return ops.nchw_to_nhwc_raw(img.to(torch.float32).contiguous())


class DDIMPipeline(DDPMPipeline):
    """Same loop; re-creates the scheduler from the given one's config like diffusers does."""

    def __init__(self, unet, scheduler):
        super().__init__(unet, DDIMScheduler.from_config(scheduler.config))


class StableDiffusionLatentPipeline:
    """The U-Net half of StableDiffusionPipeline.__call__ as driven by the reference's behaviour script
    (text_to_image/compute_model_behaviors.py:311-326): classifier-free guidance on a doubled batch and the
    scheduler update, returning LATENTS.  The VAE decoder and the CLIP text encoder (hub-fetched, frozen,
    off the hot path) stay outside: the caller passes prompt / negative-prompt embeddings and decodes.
    Scheduler: DDIM over the SD beta schedule (the sampler class shipped with miniSD is unknown offline,
    SURVEY A.13, so it is a parameter)."""

    def __init__(self, unet, scheduler=None):
        from .schedulers import DDIMScheduler
        self.unet = unet
        self.scheduler = scheduler or DDIMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear",
                                                    clip_sample=False, set_alpha_to_one=False, steps_offset=1)
        self.device = unet.device

    @torch.no_grad()
    def __call__(self, prompt_embeds, negative_prompt_embeds, num_inference_steps=100, guidance_scale=7.5,
                 generator=None, height=None, width=None):
        cfg = self.unet.config
        B = prompt_embeds.shape[0]
        hw = (height or cfg.sample_size * 8) // 8, (width or cfg.sample_size * 8) // 8
        shape = (B, cfg.in_channels, *hw)
        if generator is not None and generator.device.type == "cpu":
            lat = torch.randn(shape, generator=generator, dtype=torch.float32).to(self.device)
        else:
            lat = torch.randn(shape, generator=generator, dtype=torch.float32, device=self.device)
        sch = self.scheduler
        sch.set_timesteps(num_inference_steps)
        x = ops.nchw_to_nhwc_raw((lat * sch.init_noise_sigma).contiguous())
        ctx = torch.cat([negative_prompt_embeds, prompt_embeds], 0).to(self.device, torch.float32).contiguous()
        t2 = torch.empty(2 * B, device=self.device, dtype=torch.int64)
        clip = float(sch.config.clip_sample_range) if sch.config.clip_sample else 0.0
        for t in sch.timesteps.tolist():
            t2.fill_(t)
            eps = self.unet.forward_nhwc(torch.cat([x, x], 0), t2, ctx)
            a_t, a_p = sch.step_coefficients(t)
            ops.cfg_ddim_step_raw(x, eps, guidance_scale, a_t, a_p, clip, out=x)
        return SimpleNamespace(latents=ops.nhwc_to_nchw_raw(x))


def sd_simple_loss(unet, scheduler, latents0, prompt_embeds, timesteps, n_noises=3, generator=None):
    """compute_model_behaviors.py:391-417: mean over `n_noises` of MSE(unet(add_noise(z0, eps, t_list), t_list), eps)
    with the batch being the scheduler's whole timestep list."""
    dev = unet.device
    T = timesteps.shape[0]
    z = latents0.to(dev).expand(T, -1, -1, -1).contiguous()
    ctx = prompt_embeds.to(dev).expand(T, -1, -1).contiguous()
    total = 0.0
    with torch.no_grad():
        for _ in range(n_noises):
            if generator is not None and generator.device.type == "cpu":      # host draw, then moved (randn_tensor semantics)
                eps = torch.randn(z.shape, generator=generator, dtype=torch.float32).to(dev)
            else:
                eps = torch.randn(z.shape, generator=generator, device=dev, dtype=torch.float32)
            noisy = scheduler.add_noise(z, eps, timesteps.to(dev))
            pred = unet(noisy, timesteps.to(dev), ctx).sample
            loss, _ = ops.mse_fwd_bwd_raw(pred.contiguous(), eps)
            total += float(loss)
    return total / n_noises

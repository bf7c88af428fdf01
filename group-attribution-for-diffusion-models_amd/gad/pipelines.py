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

    def _run_steps(self, x, num_inference_steps):
        """x: NHWC device tensor, updated in place through all timesteps."""
        sch = self.scheduler
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
        x = self._run_steps(x, num_inference_steps)
        img = ops.to_image01_raw(x)                               # (x/2+0.5).clamp(0,1); already NHWC
        if output_type == "tensor":
            return SimpleNamespace(images=img)
        images = img.cpu().numpy()
        if output_type == "pil":
            from PIL import Image
            images = [Image.fromarray((im * 255).round().astype("uint8").squeeze()) for im in images]
        return SimpleNamespace(images=images)


class DDIMPipeline(DDPMPipeline):
    """Same loop; re-creates the scheduler from the given one's config like diffusers does."""

    def __init__(self, unet, scheduler):
        super().__init__(unet, DDIMScheduler.from_config(scheduler.config))

"""Test-only backend namespace: the names the entry points take from `gad`, served by the CPU oracle
(so BASELINE config 1 - the CPU plumbing run - exercises the product's entry-point code without a GPU).
Never imported by the product."""
import numpy as np
import torch

from gad.coalition import DeviceLoader, antithetic_timesteps, seed_everything  # noqa: F401  (pure torch host code)
from gad.scoring import feature_stats, frechet_distance
from oracle.diffusers_ref import (DDIMPipeline, DDIMScheduler, DDPMPipeline, DDPMScheduler, EMAModel,  # noqa: F401
                                  UNet2DModel, train_step)


class FusedTrainer:
    """Same surface as gad.FusedTrainer on torch.optim + the oracle EMA."""

    def __init__(self, model, scheduler, ema, lr=1e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, adamw=False,
                 max_grad_norm=1.0, loss_sign=1.0):
        cls = torch.optim.AdamW if adamw else torch.optim.Adam
        self.opt = cls(model.parameters(), lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self.model, self.scheduler, self.ema, self.max_norm, self.sign = model, scheduler, ema, max_grad_norm, loss_sign
        self._gn = torch.zeros(())

    def step(self, image, noise, timesteps):
        assert self.sign == 1.0
        loss, self._gn = train_step(self.model, self.opt, self.ema, self.scheduler, image, noise, timesteps, self.max_norm)
        return loss

    def grad_norm(self):
        return self._gn

    def state_dict(self):
        return self.opt.state_dict()

    def load_state_dict(self, sd):
        self.opt.load_state_dict(sd)


def fid_against_dataset(images01, dataset, device, batch_size=512, feature_dims=32):
    rng = np.random.RandomState(0)
    proj = rng.standard_normal((3 * 32 * 32, feature_dims)) / 55.0

    def feats(x):
        return x.reshape(len(x), -1).double().numpy() @ proj
    ref = dataset.device_tensor("cpu").add_(1).div_(2)
    return frechet_distance(*feature_stats(feats(images01.cpu())), *feature_stats(feats(ref)))

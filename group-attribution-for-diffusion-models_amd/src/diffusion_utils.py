"""Pipeline helpers with the reference's names and call signatures (src/diffusion_utils.py):
load_ckpt_model :111-205, build_pipeline :208-316, generate_images :319-357, run_inference :360-416.
`backend` is the module that supplies the diffusers-named classes: the MI355X engine `gad` by default."""
import os

import numpy as np
import torch

from src.ddpm_config import DDPMConfig
from src.utils import get_max_steps

_CFG = {"cifar": "cifar_config", "cifar2": "cifar2_config", "cifar100": "cifar100_config",
        "cifar100_new": "cifar100_config", "cifar100_f": "cifar100_f_config", "celeba": "celeba_config",
        "mnist": "mnist_config", "imagenette": "imagenette_config", "toy2": "cifar100_config"}


def _backend(backend=None):
    if backend is None:
        import gad
        return gad
    return backend


def dataset_config(name):
    if name not in _CFG:
        raise ValueError(f"dataset={name} is not one of {sorted(_CFG)}")
    return {**getattr(DDPMConfig, _CFG[name])}


def pruned_ckpt_path(args):
    tag = f"pruner={args.pruner}_pruning_ratio={args.pruning_ratio}_threshold={args.thr}"
    return os.path.join(args.outdir, args.dataset, "pruned", "models", tag, f"ckpt_steps_{0:0>8}.pt")


def build_model(args, config, backend=None, pruned=False):
    """Full-width model from the registry, or - for the sparsified methods - the pruned architecture.
    The reference pickles the whole pruned nn.Module (prune.py:416-421); here the pruned checkpoint
    carries its own `unet_config` (per-stage widths) next to the state_dict."""
    be = _backend(backend)
    cfg = dict(getattr(args, "unet_overrides", None) or {}, **{})
    ucfg = dict(config["unet_config"], **cfg)
    if pruned:
        ck = torch.load(pruned_ckpt_path(args), map_location="cpu", weights_only=False)
        ucfg = dict(ck.get("unet_config", ucfg))
        model = getattr(be, ucfg["_class_name"])(**ucfg)
        model.load_state_dict(ck["unet"])
        return model
    return getattr(be, ucfg["_class_name"])(**ucfg)


def load_ckpt_model(args, model_loaddir, backend=None):
    """(model, ema_model, remaining_idx, removed_idx) from the newest checkpoint of a directory."""
    be = _backend(backend)
    config = dataset_config(args.dataset)
    steps = args.trained_steps if getattr(args, "trained_steps", None) is not None else get_max_steps(model_loaddir)
    if steps is None:
        raise ValueError(f"No trained checkpoints found at {model_loaddir}")
    path = os.path.join(model_loaddir, f"ckpt_steps_{steps:0>8}.pt")
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    model = build_model(args, config, be, pruned=args.method not in ["retrain", "gd_u"])
    try:
        remaining_idx = ckpt["remaining_idx"].numpy().tolist()
        removed_idx = ckpt["removed_idx"].numpy().tolist()
    except KeyError:
        from src.datasets import create_dataset
        remaining_idx = np.arange(len(create_dataset(dataset_name=args.dataset, train=True)))
        removed_idx = np.array([], dtype=int)
    model.load_state_dict(ckpt["unet"])
    model.eval()
    print(f"Trained U-Net loaded from {path}")
    ema_model = be.EMAModel(model.parameters(), model_cls=type(model), model_config=model.config)
    ema_model.load_state_dict(ckpt["unet_ema"])       # overrides the defaults incl. optimization_step
    print(f"\tEMA loaded from {path}")
    return model, ema_model, remaining_idx, removed_idx


def build_pipeline(args, model, backend=None):
    be = _backend(backend)
    if args.dataset == "imagenette":
        raise NotImplementedError("the text-conditioned LDM needs the VQ-VAE / BERT weights "
                                  "(fetched from the hub in the reference, diffusion_utils.py:214-235)")
    if args.dataset == "celeba":
        # LDMPipeline without its (hub-fetched) VQ-VAE: the U-Net half on 3x64x64 latents with the LDM scheduler of the
        # registry (ddpm_config.py:452-461); `--precompute_stage reuse` keeps the reference in latent space the same way
        # (main.py:531-546 sets pipeline.vqvae = None)
        if getattr(args, "precompute_stage", None) != "reuse":
            raise NotImplementedError("celeba samples in latent space: pass --precompute_stage reuse (no VQ-VAE weights here)")
        sc = {k: v for k, v in dataset_config("celeba")["scheduler_config"].items() if not k.startswith("_")}
        pipe_cls = getattr(be, "LDMPipeline", None)
        pipeline = (pipe_cls(unet=model, vqvae=None, scheduler=be.DDIMScheduler(**sc)) if pipe_cls is not None
                    else be.DDPMPipeline(unet=model, scheduler=be.DDIMScheduler(**sc))).to(args.device)
        return pipeline, None, None
    pipeline = be.DDPMPipeline(unet=model, scheduler=be.DDIMScheduler()).to(args.device)
    return pipeline, None, None


def generate_images(args, pipeline, fuse=32):
    """n_samples images in batches of args.batch_size; batch b draws its noise from a CPU generator seeded
    with b; every image goes through the uint8 round trip of the PNG writer (:344-355).  Returns a float
    tensor [n,3,H,W] of k/255 values.

    A `gad` DDPM/DDIM/latent-LDM pipeline (DDIM scheduler, no VQ-VAE decode) takes the engine's fused-batch sampler:
    `fuse` reference batches - each with its own CPU-generator seed, exactly as below - are stacked per U-Net launch
    (the U-Net has no cross-sample op), which is what the benchmarked throughput is measured with.  Any other
    pipeline (the oracle backend of the CPU tests, a VQ-VAE-decoding LDM) runs batch by batch."""
    fused = _fused_sampler_for(pipeline, args.batch_size, fuse)
    if fused is not None:
        return fused.generate(args.n_samples, args.num_inference_steps).float()
    sizes = [args.batch_size] * (args.n_samples // args.batch_size)
    if args.n_samples % args.batch_size:
        sizes.append(args.n_samples % args.batch_size)
    out = []
    with torch.no_grad():
        for counter, bs in enumerate(sizes):
            images = pipeline(batch_size=bs, num_inference_steps=args.num_inference_steps, output_type="numpy",
                              generator=torch.Generator().manual_seed(counter)).images
            x = torch.from_numpy(np.asarray(images)).permute(0, 3, 1, 2)
            q = x.mul(255).add_(0.5).clamp_(0, 255).to(torch.uint8)
            out.append(q.float().div_(255))
    return torch.cat(out, 0).float()


def _fused_sampler_for(pipeline, batch_size, fuse):
    try:
        import gad
        from gad.coalition import FusedSampler
    except Exception:
        return None
    if not isinstance(pipeline, gad.DDPMPipeline) or getattr(pipeline, "vqvae", None) is not None:
        return None
    if not isinstance(pipeline.scheduler, gad.DDIMScheduler) or not isinstance(pipeline.unet, gad.UNet2DModel):
        return None
    if pipeline.unet.device.type != "cuda" or fuse <= 1:
        return None
    return FusedSampler(pipeline.unet, pipeline.scheduler, batch_size, fuse)


def run_inference(model, ema_model, config, args, backend=None):
    """Preview samples from the EMA weights (store / copy_to / sample / restore)."""
    be = _backend(backend)
    model.eval()
    ema_model.store(model.parameters())
    ema_model.copy_to(model.parameters())
    with torch.no_grad():
        pipe = be.DDIMPipeline(unet=model, scheduler=be.DDIMScheduler(num_train_timesteps=args.num_train_steps))
        samples = pipe(batch_size=config["n_samples"], num_inference_steps=args.num_inference_steps,
                       output_type="numpy").images
    samples = torch.from_numpy(np.asarray(samples)).permute(0, 3, 1, 2)
    ema_model.restore(model.parameters())
    return samples

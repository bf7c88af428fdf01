"""Train / retrain / prune-fine-tune a DDPM on a (sub)set of contributors.

Entry point kept from the reference (unconditional_generation/main.py): same flags, output
directory grammar, checkpoint file names and keys, resume rule and jsonl logging, so scripts and
artefacts interchange.  The hot loop (reference :654-726) runs on the MI355X engine: one
``trainer.step`` = add_noise + U-Net fwd/bwd + clip + Adam + EMA in HIP kernels.
"""
import argparse
import glob
import json
import math
import os
import shutil
import sys
import time

import numpy as np
import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import src.constants as constants  # noqa: E402
from src.datasets import (create_dataset, remove_data_by_class, remove_data_by_datamodel,  # noqa: E402
                          remove_data_by_shapley, remove_data_by_uniform)
from src.diffusion_utils import build_model, dataset_config, run_inference  # noqa: E402
from src.utils import compute_param_norm, get_max_steps, save_image_grid  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Training DDPM")
    p.add_argument("--load", type=str, default=None, help="path for loading pre-trained model")
    p.add_argument("--dataset", type=str, default="mnist", choices=constants.DATASET + ["toy2"])
    p.add_argument("--log_freq", type=int, default=20)
    p.add_argument("--excluded_class", type=int, default=None)
    p.add_argument("--removal_dist", type=str, default=None, choices=["uniform", "datamodel", "shapley"])
    p.add_argument("--wandb", action="store_true", default=False)
    p.add_argument("--datamodel_alpha", type=float, default=0.5)
    p.add_argument("--removal_seed", type=int, default=0)
    p.add_argument("--method", type=str, required=True, choices=constants.METHOD)
    p.add_argument("--opt_seed", type=int, default=42)
    p.add_argument("--outdir", type=str, default=constants.OUTDIR)
    p.add_argument("--keep_all_ckpts", action="store_true", default=False)
    p.add_argument("--db", type=str, default=None)
    p.add_argument("--exp_name", type=str, default=None)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--pruning_ratio", type=float, default=0.3)
    p.add_argument("--pruner", type=str, default="magnitude",
                   choices=["taylor", "random", "magnitude", "reinit", "diff-pruning"])
    p.add_argument("--thr", type=float, default=0.05)
    p.add_argument("--mixed_precision", type=str, default="no", choices=["no", "fp16", "bf16"])
    p.add_argument("--precompute_stage", type=str, default=None, choices=[None, "save", "reuse"])
    p.add_argument("--use_8bit_optimizer", default=False, action="store_true")
    p.add_argument("--ema_inv_gamma", type=float, default=1.0)
    p.add_argument("--ema_power", type=float, default=3 / 4)
    p.add_argument("--ema_max_decay", type=float, default=0.9999)
    p.add_argument("--num_inference_steps", type=int, default=100)
    p.add_argument("--num_train_steps", type=int, default=1000)
    p.add_argument("--save_null_model", action="store_true", default=False)
    # engine extras (not in the reference)
    p.add_argument("--training_steps", type=int, default=None, help="override the registry's step count")
    p.add_argument("--batch_size", type=int, default=None, help="override the registry's batch size")
    p.add_argument("--device", type=str, default="cuda:0")
    return p.parse_args(argv)


def removal_directory(args):
    """full | excluded_{c} | {dist}/{dist}[_alpha=a]_seed={k}   (reference :236-243)."""
    d = "full"
    if args.excluded_class is not None:
        d = f"excluded_{args.excluded_class}"
    if args.removal_dist is not None:
        d = f"{args.removal_dist}/{args.removal_dist}"
        if args.removal_dist == "datamodel":
            d += f"_alpha={args.datamodel_alpha}"
        d += f"_seed={args.removal_seed}"
    return d


def split_contributors(args, dataset):
    by_class_sets = ["cifar100", "cifar100_f", "celeba", "toy2"]
    if args.excluded_class is not None:
        return remove_data_by_class(dataset, excluded_class=args.excluded_class)
    if args.removal_dist == "uniform":
        # the reference passes by_class=True to a function without that parameter (main.py:268-270) -> TypeError
        return remove_data_by_uniform(dataset, seed=args.removal_seed, by_class=True)
    if args.removal_dist == "datamodel":
        return remove_data_by_datamodel(dataset, alpha=args.datamodel_alpha, seed=args.removal_seed,
                                        by_class=args.dataset in by_class_sets)
    if args.removal_dist == "shapley":
        return remove_data_by_shapley(dataset, seed=args.removal_seed, by_class=args.dataset in by_class_sets)
    return np.arange(len(dataset)), np.array([], dtype=int)


def main(args, backend=None):
    if backend is None:
        import gad as backend
    if args.use_8bit_optimizer or args.gradient_accumulation_steps != 1:
        raise NotImplementedError("the MI355X engine runs the reference default: Adam(W) in fp32, no accumulation")
    if hasattr(backend, "set_operand_precision"):      # --mixed_precision fp16|bf16 -> bf16 operands, fp32 everything else
        backend.set_operand_precision(args.mixed_precision)
    if args.dataset == "celeba" and args.precompute_stage != "reuse":
        # :486-530 encode with the hub-fetched CompVis/ldm-celebahq-256 VQ-VAE; only the latent path is on the card
        raise NotImplementedError("celeba trains on precomputed VQ-VAE latents: pass --precompute_stage reuse with "
                                  "{outdir}/celeba/precomputed_emb/vqvae_output.pt (or GAD_LATENTS) in place")
    device = torch.device(args.device)
    config = dataset_config(args.dataset)
    removal_dir = removal_directory(args)
    model_outdir = os.path.join(args.outdir, args.dataset, args.method, "models", removal_dir)
    sample_outdir = os.path.join(args.outdir, args.dataset, args.method, "samples", removal_dir)
    os.makedirs(model_outdir, exist_ok=True)
    os.makedirs(sample_outdir, exist_ok=True)

    train_dataset = create_dataset(dataset_name=args.dataset, train=True)
    remaining_idx, removed_idx = split_contributors(args, train_dataset)
    if args.method == "ga":
        remaining_idx, removed_idx = removed_idx, remaining_idx
    np.save(os.path.join(model_outdir, "remaining_idx.npy"), remaining_idx)
    np.save(os.path.join(model_outdir, "removed_idx.npy"), removed_idx)

    backend.seed_everything(args.opt_seed)                     # seed for model optimisation (:308)
    training_steps = args.training_steps if args.training_steps is not None else config["training_steps"][args.method]
    batch_size = args.batch_size or config["batch_size"]

    def fresh_model():
        m = build_model(args, config, backend, pruned=args.method != "retrain")
        e = backend.EMAModel(m.parameters(), decay=args.ema_max_decay, use_ema_warmup=False,
                             inv_gamma=args.ema_inv_gamma, power=args.ema_power, model_cls=type(m),
                             model_config=m.config)
        return m, e

    # ---- resume from the newest checkpoint; a corrupt one wipes the directory and restarts (:334-381) ----
    total_steps_time, done = 0.0, 0
    model, ema_model = fresh_model()
    existing = get_max_steps(model_outdir)
    opt_state = None
    if existing is not None:
        path = os.path.join(model_outdir, f"ckpt_steps_{existing:0>8}.pt")
        try:
            ckpt = torch.load(path, map_location="cpu", weights_only=False)
            model.load_state_dict(ckpt["unet"])
            ema_model.load_state_dict(ckpt["unet_ema"])
            remaining_idx, removed_idx = ckpt["remaining_idx"].numpy(), ckpt["removed_idx"].numpy()
            total_steps_time, done, opt_state = ckpt["total_steps_time"], existing, ckpt.get("optimizer")
            print(f"U-Net and U-Net EMA resumed from {path}")
        except (RuntimeError, EOFError, KeyError) as err:
            print(f"Check point {path} is corrupted ({err}); restarting from scratch")
            shutil.rmtree(model_outdir)
            os.makedirs(model_outdir, exist_ok=True)
            model, ema_model = fresh_model()
    model.to(device)
    ema_model.to(device)

    scheduler = backend.DDPMScheduler(**config["scheduler_config"])
    okw = dict(config["optimizer_config"]["kwargs"])
    trainer = backend.FusedTrainer(model, scheduler, ema_model, lr=okw.get("lr", 1e-4),
                                   weight_decay=okw.get("weight_decay", 0.0),
                                   adamw=config["optimizer_config"]["class_name"] == "AdamW", max_grad_norm=1.0,
                                   loss_sign=-1.0 if args.method == "ga" else 1.0)
    if opt_state is not None:
        trainer.load_state_dict(opt_state)
    loader = backend.DeviceLoader(train_dataset, remaining_idx, batch_size, device)
    n_t = scheduler.config.num_train_timesteps

    def save_ckpt(step):
        if not args.keep_all_ckpts:
            for f in glob.glob(os.path.join(model_outdir, "ckpt_steps_*.pt")):
                os.remove(f)
        torch.save({"unet": {k: v.detach().cpu().contiguous() for k, v in model.state_dict().items()},
                    "unet_config": _cfg_dict(model.config), "unet_ema": _cpu(ema_model.state_dict()),
                    "optimizer": trainer.state_dict(), "lr_scheduler": {"last_epoch": step},
                    "remaining_idx": torch.from_numpy(np.asarray(remaining_idx)),
                    "removed_idx": torch.from_numpy(np.asarray(removed_idx)), "total_steps_time": total_steps_time},
                   os.path.join(model_outdir, f"ckpt_steps_{step:0>8}.pt"))
        print(f"Checkpoint saved at step {step}")

    if args.save_null_model:
        save_ckpt(done)

    t_mark = time.time()
    loss = None
    while done < training_steps:
        for image, _ in loader:
            noise = torch.randn_like(image)
            ts = backend.antithetic_timesteps(n_t, image.shape[0], device)
            loss = trainer.step(image, noise, ts)
            done += 1
            if done % args.log_freq == 0:
                dt = time.time() - t_mark
                total_steps_time += dt
                print(f"Step[{done}/{training_steps}], steps_time: {dt:.3f}, loss: {float(loss):.5f}, "
                      f"gradient norms: {float(trainer.grad_norm()):.5f}, parameters norms: "
                      f"{compute_param_norm(model):.5f}, lr: {okw.get('lr', 1e-4):.6f}", flush=True)
                t_mark = time.time()
            if done % config["sample_freq"][args.method] == 0 or done == training_steps:
                t_s = time.time()
                samples = run_inference(model, ema_model, config, args, backend)
                print(f"Step[{done}/{training_steps}], sampling_time: {time.time() - t_s:.3f}", flush=True)
                if args.db is not None:
                    info = dict(vars(args), param_update_steps=f"{done}", loss=f"{float(loss):.5f}",
                                lr=f"{okw.get('lr', 1e-4):.6f}", sampling_time=f"{time.time() - t_s:.3f}")
                    with open(args.db, "a+") as f:
                        f.write(json.dumps(info, default=str) + "\n")
                samples = samples[: constants.MAX_NUM_SAMPLE_IMAGES_TO_SAVE]
                save_image_grid(samples, os.path.join(sample_outdir, f"steps_{done:0>8}.png"),
                                nrow=int(math.sqrt(config["n_samples"])))
                t_mark = time.time()
            if done % config["ckpt_freq"][args.method] == 0 or done == training_steps:
                save_ckpt(done)
                t_mark = time.time()
            if done == training_steps:
                break
    return True


def _cfg_dict(cfg):
    return dict(cfg) if isinstance(cfg, dict) else dict(vars(cfg))


def _cpu(sd):
    return {k: ([t.detach().cpu() for t in v] if isinstance(v, list) else v) for k, v in sd.items()}


if __name__ == "__main__":
    if main(parse_args()):
        print("Model optimization done!")

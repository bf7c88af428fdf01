"""Sparsified fine-tuning ("unlearning") of a pre-trained DDPM on one contributor coalition, then the
model-behaviour score and one jsonl row.

Entry point kept from the reference (unconditional_generation/unlearn.py): same flags for the sFT path
(`--method gd`), same coalition semantics (the by_class quirk of :331 included), same jsonl keys
(:277,834-837,960-968).  The cycle itself is `gad.coalition.CoalitionEngine`: fused training steps,
EMA weights, 64-image preview, fused-batch DDIM sampling and the Frechet score on the MI355X.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

import src.constants as constants  # noqa: E402
from src.datasets import (create_dataset, remove_data_by_datamodel, remove_data_by_loo,  # noqa: E402
                          remove_data_by_shapley, remove_data_by_uniform, remove_data_for_aoi)
from src.diffusion_utils import build_pipeline, dataset_config, generate_images, load_ckpt_model  # noqa: E402
from src.utils import save_image_grid  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Training DDPM")
    p.add_argument("--load", type=str, default=None, help="directory of the pre-trained (pruned) model")
    p.add_argument("--dataset", type=str, default="mnist", choices=constants.DATASET + ["toy2"])
    p.add_argument("--excluded_class", type=str, default=None)
    p.add_argument("--removal_dist", type=str, default=None,
                   choices=["uniform", "datamodel", "shapley", "loo", "add_one_in"])
    p.add_argument("--datamodel_alpha", type=float, default=0.5)
    p.add_argument("--removal_seed", type=int, default=0)
    p.add_argument("--method", type=str, required=True, choices=constants.METHOD)
    p.add_argument("--iu_ratio", type=float, default=0.5)
    p.add_argument("--ga_ratio", type=float, default=1.0)
    p.add_argument("--gd_steps", type=int, default=4000)
    p.add_argument("--lora_rank", type=int, default=16)
    p.add_argument("--lora_dropout", type=float, default=0.05)
    p.add_argument("--opt_seed", type=int, default=42)
    p.add_argument("--outdir", type=str, default=constants.OUTDIR)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--db", type=str, required=True)
    p.add_argument("--reference_dir", type=str, default=None)
    p.add_argument("--batch_size", type=int, default=32)
    p.add_argument("--n_samples", type=int, default=10240)
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
    p.add_argument("--model_behavior", type=str, default=None, choices=[None, "global", "local"])
    p.add_argument("--use_ema", default=False, action="store_true")
    p.add_argument("--exp_name", type=str, default=None)
    p.add_argument("--n_noises", type=int, default=50)
    p.add_argument("--num_inference_steps", type=int, default=100)
    p.add_argument("--num_train_steps", type=int, default=1000)
    p.add_argument("--trained_steps", type=int, default=None)
    p.add_argument("--device", type=str, default="cuda:0")
    return p.parse_args(argv)


def coalition(args, dataset):
    d = args.removal_dist
    if d == "uniform":
        return remove_data_by_uniform(dataset, seed=args.removal_seed, by_class=True)   # TypeError, as :323-325
    if d == "datamodel":
        return remove_data_by_datamodel(dataset, alpha=args.datamodel_alpha, seed=args.removal_seed, by_class=True)
    if d == "shapley":
        # `if args.dataset == "cifar100" or "celeba":` is always true in the reference (:331) -> always by class
        return remove_data_by_shapley(dataset, seed=args.removal_seed, by_class=True)
    if d == "loo":
        return remove_data_by_loo(dataset, args.removal_seed)
    if d == "add_one_in":
        return remove_data_for_aoi(dataset, args.removal_seed)
    return np.arange(len(dataset)), np.array([], dtype=int)


def main(args, backend=None):
    if backend is None:
        import gad as backend
    if args.method not in ("gd", "gd_u", "ga", "ga_u"):
        raise NotImplementedError(f"method={args.method}: the engine implements the sFT family (gd/ga); "
                                  "iu / lora / esd are baseline methods outside the hot path")
    if args.model_behavior == "local":
        raise NotImplementedError("local model behaviours (SSIM / per-image losses) are outside the hot path")
    if hasattr(backend, "set_operand_precision"):      # --mixed_precision fp16|bf16 -> bf16 operands, fp32 everything else
        backend.set_operand_precision(args.mixed_precision)
    device = torch.device(args.device)
    args.device = device
    info = dict(vars(args))
    config = dataset_config(args.dataset)
    dataset = create_dataset(dataset_name=args.dataset, train=True)
    remaining_idx, removed_idx = coalition(args, dataset)
    if args.method in ("ga", "ga_u"):
        remaining_idx, removed_idx = removed_idx, remaining_idx
    backend.seed_everything(args.opt_seed)

    model, ema_model, _, _ = load_ckpt_model(args, args.load, backend)
    model.to(device)
    ema_model.to(device)
    scheduler = backend.DDPMScheduler(**config["scheduler_config"])
    okw = dict(config["optimizer_config"]["kwargs"])
    trainer = backend.FusedTrainer(model, scheduler, ema_model, lr=okw.get("lr", 1e-4),
                                   weight_decay=okw.get("weight_decay", 0.0),
                                   adamw=config["optimizer_config"]["class_name"] == "AdamW", max_grad_norm=1.0,
                                   loss_sign=-1.0 if args.method.startswith("ga") else 1.0)
    loader = backend.DeviceLoader(dataset, remaining_idx, config["batch_size"], device)
    n_t = scheduler.config.num_train_timesteps
    steps_goal = args.gd_steps if args.method.startswith("gd") else int(config["training_steps"]["ga"] // args.ga_ratio)

    t0 = time.time()
    steps = 0
    while steps < steps_goal:
        for image, _ in loader:
            noise = torch.randn_like(image)
            ts = backend.antithetic_timesteps(n_t, image.shape[0], device)
            trainer.step(image, noise, ts)
            steps += 1
            if steps == steps_goal:
                break
    if device.type == "cuda":
        torch.cuda.synchronize(device)
    total_steps_time = time.time() - t0

    ema_model.store(model.parameters())           # the EMA is used for inference (:751-753)
    ema_model.copy_to(model.parameters())
    model.eval()
    pipeline, _, _ = build_pipeline(args, model, backend)
    sample_outdir = os.path.join(args.outdir, args.dataset, args.method, "samples",
                                 f"{args.removal_dist}/{args.removal_dist}_seed={args.removal_seed}")
    preview = pipeline(batch_size=config["n_samples"], num_inference_steps=args.num_inference_steps,
                       output_type="numpy").images
    save_image_grid(torch.from_numpy(np.asarray(preview)).permute(0, 3, 1, 2),
                    os.path.join(sample_outdir, f"prutirb_ratio_{args.iu_ratio}_steps_{steps_goal:0>8}.png"),
                    nrow=int(np.sqrt(config["n_samples"])))

    t1 = time.time()
    if args.model_behavior == "global":
        print(f"Generating {args.n_samples}...")
        images = generate_images(args, pipeline)
        if args.dataset == "celeba":                               # entropy, cluster_count, cluster_proportions (:787-803)
            info.update(backend.diversity_against_dataset(images, dataset, device, num_cluster=20))
            print(f"entropy: {info['entropy']}")
        elif hasattr(backend, "global_scores_against_dataset"):      # fid_value, is, precision, recall (:807-837)
            info.update(backend.global_scores_against_dataset(images, dataset, device, args.batch_size))
        else:
            info["fid_value"] = backend.fid_against_dataset(images, dataset, device, args.batch_size)
        print("; ".join(f"{k}: {info[k]}" for k in ("fid_value", "precision", "recall", "is") if k in info))
    info.update(total_steps_time=total_steps_time, trained_steps=steps_goal,
                remaining_idx=np.asarray(remaining_idx).tolist(), removed_idx=np.asarray(removed_idx).tolist(),
                device=str(device), total_sampling_time=time.time() - t1)
    with open(args.db, "a+") as f:
        f.write(json.dumps(info, default=str) + "\n")
    print(f"Results saved to the database at {args.db}")
    return True


if __name__ == "__main__":
    if main(parse_args()):
        print("Unlearning is done!")

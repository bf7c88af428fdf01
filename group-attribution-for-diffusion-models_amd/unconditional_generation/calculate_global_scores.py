"""Global model behaviours of a trained / retrained model: load the newest checkpoint of a model directory, sample
`--n_samples` images, score them, append one jsonl row.

Entry point kept from the reference (unconditional_generation/calculate_global_scores.py:28-482, the `--generate_samples`
branch with `--sample_dir` unset): same flags, the same model directory grammar (:166-176,239-245), `remaining_idx` /
`removed_idx` from the checkpoint, `--use_ema` (:250-251), row = vars(args) + scores + total_sampling_time + index lists
(:473-481).  These rows are what lds.py reads as its test / null / full databases (lds.py:298-345).  The sampling and
the score tail are the engine's (fused-batch DDIM sampler, on-device Frechet / IS / precision-recall; the CelebA branch
writes entropy / cluster_count / cluster_proportions as calculate_global_scores_diversity.py does)."""
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
from src.datasets import create_dataset  # noqa: E402
from src.diffusion_utils import build_pipeline, generate_images, load_ckpt_model  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Calculate model behavior scores")
    p.add_argument("--sample_dir", type=str, default=None)
    p.add_argument("--reference_dir", type=str, default=None)
    p.add_argument("--outdir", type=str, default=constants.OUTDIR)
    p.add_argument("--dataset", type=str, choices=constants.DATASET + ["toy2"], default=None)
    p.add_argument("--db", type=str, required=True)
    p.add_argument("--excluded_class", type=str, default=None)
    p.add_argument("--removal_dist", type=str, default=None)
    p.add_argument("--datamodel_alpha", type=float, default=0.5)
    p.add_argument("--removal_seed", type=int, default=0)
    p.add_argument("--method", type=str, choices=constants.METHOD)
    p.add_argument("--exp_name", type=str, default=None, required=True)
    p.add_argument("--batch_size", type=int, default=512)
    p.add_argument("--device", type=str, default="cuda:0")
    p.add_argument("--seed", type=int, default=42)
    p.add_argument("--generate_samples", action="store_true", default=False)
    p.add_argument("--n_samples", type=int, default=100000)
    p.add_argument("--num_inference_steps", type=int, default=100)
    p.add_argument("--use_ema", action="store_true", default=False)
    p.add_argument("--trained_steps", type=int, default=None)
    p.add_argument("--pruning_ratio", type=float, default=0.3)
    p.add_argument("--pruner", type=str, default="magnitude", choices=["taylor", "random", "magnitude", "reinit", "diff-pruning"])
    p.add_argument("--thr", type=float, default=0.05)
    p.add_argument("--precompute_stage", type=str, default=None, choices=[None, "save", "reuse"])   # celeba latent mode
    return p.parse_args(argv)


def removal_directory(args):
    """:166-176"""
    d = "full"
    if args.excluded_class is not None:
        d = "excluded_" + ",".join(map(str, sorted(int(k) for k in args.excluded_class.split(","))))
    if args.removal_dist is not None:
        d = f"{args.removal_dist}/{args.removal_dist}"
        if args.removal_dist == "datamodel":
            d += f"_alpha={args.datamodel_alpha}"
        d += f"_seed={args.removal_seed}"
    return d


def main(args, backend=None):
    if backend is None:
        import gad as backend
    if args.sample_dir:
        raise NotImplementedError("scoring a directory of PNG samples (--sample_dir) is outside the hot path; "
                                  "the engine scores the tensors it generates")
    backend.seed_everything(args.seed)
    info = dict(vars(args))
    device = torch.device(args.device)
    dataset = create_dataset(dataset_name=args.dataset, train=True)
    model_loaddir = os.path.join(args.outdir, args.dataset, args.method, "models", removal_directory(args))
    model, ema_model, remaining_idx, removed_idx = load_ckpt_model(args, model_loaddir, backend)
    model.to(device)
    if args.use_ema:                                                    # :250-251
        ema_model.to(device)
        ema_model.copy_to(model.parameters())
    model.eval()
    pipeline, _, _ = build_pipeline(args, model, backend)
    t0 = time.time()
    images = generate_images(args, pipeline)
    if args.dataset == "celeba":
        info.update(backend.diversity_against_dataset(images, dataset, device, num_cluster=20))
    elif hasattr(backend, "global_scores_against_dataset"):
        sc = backend.global_scores_against_dataset(images, dataset, device, args.batch_size)
        info.update(sc)
        info["fid_value"] = float(sc["fid_value"])                      # lds.py reads float(record["fid_value"])
    else:
        info["fid_value"] = backend.fid_against_dataset(images, dataset, device, args.batch_size)
    info.update(total_sampling_time=time.time() - t0, sample_dir=args.sample_dir,
                remaining_idx=np.asarray(remaining_idx).tolist(), removed_idx=np.asarray(removed_idx).tolist())
    with open(args.db, "a+") as f:
        f.write(json.dumps(info, default=str) + "\n")
    print(f"Results saved to the database at {args.db}")
    return True


if __name__ == "__main__":
    main(parse_args())

"""LoRA fine-tuning of a Stable-Diffusion U-Net on a coalition of artists.

Entry point kept from the reference (text_to_image/train_text_to_image_lora.py): method routing
retrain / pruned_ft / sparse_gd / gd (:614-642), output directory grammar (:579-647), skip-if-done (:649-657),
coalition over the rows of `{cls}_{unit}s.csv` with `removal_idx.csv` (:935-1024), LoRA on to_q/to_k/to_v/to_out
of all 32 attentions (:776-820) or loaded, possibly pruned/ragged, from `lora_dir` (:821-853), AdamW + cosine
(:896-902,1122-1127), the hot loop (:1215-1311), `time.csv` (:1203-1209,1315-1319) and
`pytorch_lora_weights.safetensors` (:1459).

The frozen VAE encoder and CLIP text encoder of the reference are hub-fetched models outside the hot path: this
entry point trains from a latent cache `{train_data_dir}/latent_cache.pt`
(dict: latents [N,4,h,w] already multiplied by vae.config.scaling_factor, text_emb [N or 1,77,768], and one
list per metadata column, e.g. "artist", "filename", "style") - the analogue of the reference's
`vqvae_output.pt` cache for CelebA (unconditional_generation/main.py:496-525).  `--synthetic_cache` writes a
seeded stand-in with ArtBench's shape (5 000 images, 258 artists) when no cache exists.
"""
import argparse
import math
import os
import sys
import time

import numpy as np
import pandas as pd
import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _HERE not in sys.path:
    sys.path.insert(0, _HERE)

from src.datasets import (remove_data_by_datamodel, remove_data_by_loo, remove_data_by_shapley,  # noqa: E402
                          remove_data_by_uniform, remove_data_for_aoi)
from src.ddpm_config import LoraSparseUnlearningConfig, LoraUnlearningConfig  # noqa: E402


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="LoRA fine-tuning of a text-to-image U-Net")
    p.add_argument("--pretrained_model_name_or_path", type=str, default="lambdalabs/miniSD-diffusers")
    p.add_argument("--unet_weights", type=str, default=None, help="local state_dict of the base U-Net (optional)")
    p.add_argument("--train_data_dir", type=str, required=True)
    p.add_argument("--output_dir", type=str, default="sd-model-finetuned-lora")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--resolution", type=int, default=256)
    p.add_argument("--train_batch_size", type=int, default=16)
    p.add_argument("--num_train_epochs", type=int, default=100)
    p.add_argument("--max_train_steps", type=int, default=None)
    p.add_argument("--learning_rate", type=float, default=1e-4)
    p.add_argument("--lr_scheduler", type=str, default="constant")
    p.add_argument("--lr_warmup_steps", type=int, default=0)
    p.add_argument("--adam_beta1", type=float, default=0.9)
    p.add_argument("--adam_beta2", type=float, default=0.999)
    p.add_argument("--adam_weight_decay", type=float, default=1e-2)
    p.add_argument("--adam_epsilon", type=float, default=1e-08)
    p.add_argument("--max_grad_norm", type=float, default=1.0)
    p.add_argument("--snr_gamma", type=float, default=None, help="min-SNR loss weighting (reference :319)")
    p.add_argument("--noise_offset", type=float, default=0.0, help="offset noise scale (reference :458)")
    p.add_argument("--mixed_precision", type=str, default=None, choices=["no", "fp16", "bf16"])
    p.add_argument("--rank", type=int, default=4)
    p.add_argument("--cls_key", type=str, default=None)
    p.add_argument("--cls", type=str, default=None)
    p.add_argument("--removal_dist", type=str, default=None, choices=["uniform", "shapley", "datamodel", "loo", "aoi"])
    p.add_argument("--datamodel_alpha", type=float, default=None)
    p.add_argument("--removal_unit", type=str, default=None, choices=["artist", "filename"])
    p.add_argument("--removal_seed", type=int, default=0)
    p.add_argument("--loo_idx", type=int, default=None)
    p.add_argument("--aoi_idx", type=int, default=None)
    p.add_argument("--removal_rank_file", type=str, default=None)
    p.add_argument("--removal_rank_proportion", type=float, default=None)
    p.add_argument("--removal_bottom_proportion", type=float, default=None)
    p.add_argument("--method", type=str, default="retrain", choices=["retrain", "pruned_ft", "sparse_gd", "gd"])
    p.add_argument("--pruning_ratio", type=float, default=None)
    p.add_argument("--lora_dir", type=str, default=None)
    p.add_argument("--lora_steps", type=int, default=None)
    p.add_argument("--synthetic_cache", action="store_true", help="create a seeded stand-in latent cache if missing")
    p.add_argument("--unet_overrides", type=str, default=None, help="json dict of UNet2DConditionModel kwargs (tests)")
    p.add_argument("--device", type=str, default="cuda:0")
    return p.parse_args(argv)


def removal_directory(a):
    d = "full"
    if a.removal_dist is not None:
        if a.removal_unit is None:
            raise ValueError("--removal_unit is not specified")
        dist = a.removal_dist + (f"_alpha={a.datamodel_alpha}" if a.removal_dist == "datamodel" else "")
        d = f"{a.removal_unit}_{dist}"
        if a.removal_dist == "loo":
            d += f"/{dist}_idx={a.loo_idx}"
        elif a.removal_dist == "aoi":
            d += f"/{dist}_idx={a.aoi_idx}"
        else:
            d += f"/{dist}_seed={a.removal_seed}"
    if a.removal_rank_file is not None:
        if a.removal_rank_proportion is not None:
            d = f"counterfactual_top_{a.removal_rank_proportion}"
        elif a.removal_bottom_proportion is not None:
            d = f"counterfactual_bottom_{a.removal_bottom_proportion}"
        else:
            raise ValueError
        d += "/" + os.path.basename(a.removal_rank_file).split(".")[0]
    return d


def synthetic_cache(path, n=5000, n_artists=258, res=256, ctx_dim=768, seed=0):
    """Seeded stand-in with ArtBench post-impressionism's shape (ddpm_config.py:703: 258 groups, 5 000 images)."""
    g = torch.Generator().manual_seed(seed)
    rng = np.random.RandomState(seed)
    artists = [f"artist_{i:03d}" for i in rng.randint(0, n_artists, size=n)]
    for i in range(n_artists):                       # every artist appears at least once
        artists[i] = f"artist_{i:03d}"
    cache = {"latents": torch.randn(n, 4, res // 8, res // 8, generator=g) * 0.8,
             "text_emb": torch.randn(1, 77, ctx_dim, generator=g) * 0.5,
             "artist": artists, "filename": [f"img_{i:05d}.jpg" for i in range(n)],
             "style": ["post_impressionism"] * n}
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save(cache, path)
    units = sorted(set(artists))
    pd.DataFrame({"artist": units}).to_csv(os.path.join(os.path.dirname(path), "post_impressionism_artists.csv"), index=False)
    pd.DataFrame({"filename": cache["filename"]}).to_csv(
        os.path.join(os.path.dirname(path), "post_impressionism_filenames.csv"), index=False)
    return cache


def coalition_rows(a, model_outdir, units_df):
    """remaining / removed indices into the removal-unit table; cached in removal_idx.csv (:948-990)."""
    f = os.path.join(model_outdir, "removal_idx.csv")
    if os.path.exists(f):
        df = pd.read_csv(f)
        return df["idx"][df["remaining"]].to_numpy(), df["idx"][~df["remaining"]].to_numpy()
    if a.removal_dist == "shapley":
        rem, rmv = remove_data_by_shapley(units_df, a.removal_seed)
    elif a.removal_dist == "uniform":
        rem, rmv = remove_data_by_uniform(units_df, a.removal_seed)
    elif a.removal_dist == "datamodel":
        rem, rmv = remove_data_by_datamodel(dataset=units_df, seed=a.removal_seed, alpha=a.datamodel_alpha)
    elif a.removal_dist == "loo":
        rem, rmv = remove_data_by_loo(dataset=units_df, loo_idx=a.loo_idx)
    elif a.removal_dist == "aoi":
        rem, rmv = remove_data_for_aoi(dataset=units_df, aoi_idx=a.aoi_idx)
    else:
        rank = np.load(a.removal_rank_file)
        if a.removal_rank_proportion is not None:
            k = math.floor(len(rank) * a.removal_rank_proportion)
            rmv, rem = rank[:k], rank[k:]
        else:
            k = math.floor(len(rank) * a.removal_bottom_proportion)
            rmv, rem = rank[-k:], rank[:-k]
    pd.concat([pd.DataFrame({"idx": rem, "remaining": True}), pd.DataFrame({"idx": rmv, "remaining": False})]).to_csv(f, index=False)
    return np.asarray(rem), np.asarray(rmv)


def main(a, backend=None):
    if backend is None:
        import gad as backend
    # :659-668 autocast -> bf16 operands in the contraction kernels (fp32 storage / accumulation; no loss scaling needed)
    if hasattr(backend, "set_operand_precision"):
        backend.set_operand_precision(a.mixed_precision)
    removal_dir = removal_directory(a)
    a.dataset = "artbench" if "artbench" in a.train_data_dir else os.path.basename(os.path.normpath(a.train_data_dir))
    if a.cls is not None and a.cls_key is not None:
        a.dataset += f"_{a.cls}"
    if a.method == "pruned_ft":
        a.method = f"pruned_ft_ratio={a.pruning_ratio}_lr={a.learning_rate}"
        a.lora_dir = os.path.join(a.output_dir, a.dataset, f"pruned_ratio={a.pruning_ratio}", "models", removal_dir)
    elif a.method in ("sparse_gd", "gd"):
        cfgs = LoraSparseUnlearningConfig if a.method == "sparse_gd" else LoraUnlearningConfig
        if a.dataset != "artbench_post_impressionism":
            raise NotImplementedError(a.dataset)
        cfg = cfgs.artbench_post_impressionism_config
        a.lora_dir = a.lora_dir or cfg["lora_dir"]
        a.lora_steps = cfg.get("lora_steps")
        a.max_train_steps = a.max_train_steps or cfg["max_train_steps"]
    out_root = os.path.join(a.output_dir, a.dataset, a.method)
    model_outdir = os.path.join(out_root, "models", removal_dir)
    a.model_outdir = model_outdir
    if os.path.exists(os.path.join(model_outdir, "pytorch_lora_weights.safetensors")):
        print(f"Found trained LoRA weights at {model_outdir}. Process cancelled.")
        return False
    os.makedirs(model_outdir, exist_ok=True)
    if a.seed is not None:
        backend.seed_everything(a.seed)
    device = torch.device(a.device)

    # ---- data: latent cache + removal-unit table ----
    cache_path = os.path.join(a.train_data_dir, "latent_cache.pt")
    if not os.path.exists(cache_path):
        if not a.synthetic_cache:
            raise FileNotFoundError(f"{cache_path} not found (pass --synthetic_cache for a seeded stand-in)")
        synthetic_cache(cache_path, res=a.resolution)
    cache = torch.load(cache_path, map_location="cpu", weights_only=False)
    keep = np.arange(len(cache["latents"]))
    if a.cls is not None and a.cls_key is not None:
        keep = keep[np.array(cache[a.cls_key]) == a.cls]
    if a.removal_dist is not None or a.removal_rank_file is not None:
        unit_file = os.path.join(a.train_data_dir, f"{a.cls}_{a.removal_unit}s.csv" if a.cls else f"{a.removal_unit}s.csv")
        units_df = pd.read_csv(unit_file)
        rem, _ = coalition_rows(a, model_outdir, units_df)
        kept_units = set(units_df.iloc[rem, 0].tolist())
        col = np.array(cache[a.removal_unit])[keep]
        keep = keep[np.isin(col, list(kept_units))]
        assert set(np.array(cache[a.removal_unit])[keep]) == kept_units or len(keep) == 0

    # ---- model ----
    import json
    ucfg = json.loads(a.unet_overrides) if a.unet_overrides else {}
    unet = backend.UNet2DConditionModel(**ucfg)
    if a.unet_weights:
        unet.load_state_dict(torch.load(a.unet_weights, map_location="cpu", weights_only=False))
    unet.to(device)
    if a.method == "retrain":
        lora_params = unet.inject_lora(rank=a.rank)
    else:
        name = "pytorch_lora_weights.safetensors" if a.lora_steps is None else f"pytorch_lora_weights_{a.lora_steps}.safetensors"
        for p in unet.parameters():
            p.requires_grad_(False)
        unet.load_attn_procs(a.lora_dir, weight_name=name)
        lora_params = [p for n, p in unet.named_parameters() if "lora_layer" in n]
        for p in lora_params:
            p.requires_grad_(True)
    print(f"Number of trainable LoRA parameters: {sum(p.numel() for p in lora_params)}")
    if len(keep) == 0:                                       # all data removed: save and stop (:1026-1033)
        unet.save_attn_procs(model_outdir)
        return True

    latents = cache["latents"][keep].to(device)               # resident in HBM (5 000 x 4x32x32 fp32 = 82 MB)
    text = cache["text_emb"].to(device)
    steps_per_epoch = math.ceil(len(keep) / a.train_batch_size)
    max_steps = a.max_train_steps or a.num_train_epochs * steps_per_epoch
    sched = backend.DDPMScheduler(beta_start=0.00085, beta_end=0.012, beta_schedule="scaled_linear", num_train_timesteps=1000)
    trainer = backend.FusedTrainer(unet, sched, None, lr=a.learning_rate, betas=(a.adam_beta1, a.adam_beta2),
                                   eps=a.adam_epsilon, weight_decay=a.adam_weight_decay, adamw=True,
                                   max_grad_norm=a.max_grad_norm, params=lora_params,
                                   lr_schedule=backend.lr_lambda(a.lr_scheduler, max_steps, a.lr_warmup_steps),
                                   # GAD_TRAIN_GRAPH=1: replay the step from a hipGraph (gad.FusedTrainer; for the launch-bound
                                   # half-precision step; the injected test backend does not take the keyword)
                                   **({"use_graph": True} if os.environ.get("GAD_TRAIN_GRAPH") else {}))
    time_file = os.path.join(model_outdir, "time.csv")
    if not os.path.exists(time_file):
        with open(time_file, "w") as f:
            f.write("step,time,gpu\n" if a.max_train_steps is not None else "epoch,time,gpu\n")
    gpu_name = torch.cuda.get_device_name(device) if device.type == "cuda" else "cpu"
    step = 0
    while step < max_steps:
        t_epoch = time.time()
        perm = torch.randperm(len(keep), device=device)
        for s in range(0, len(keep), a.train_batch_size):
            t0 = time.time()
            sel = perm[s:s + a.train_batch_size]
            x0 = latents.index_select(0, sel)
            noise = torch.randn_like(x0)
            if a.noise_offset:                                                          # :1227-1232
                noise += a.noise_offset * torch.randn((x0.shape[0], x0.shape[1], 1, 1), device=device)
            ts = torch.randint(0, 1000, (x0.shape[0],), device=device).long()
            ctx = text.expand(x0.shape[0], -1, -1) if text.shape[0] == 1 else text.index_select(0, sel)
            lw = None
            if a.snr_gamma is not None:                                                  # :1276-1298
                from gad.schedulers import min_snr_weights
                lw = min_snr_weights(sched.alphas_cumprod, ts, a.snr_gamma)
            loss = trainer.step(x0, noise, ts, ctx.contiguous(), loss_weights=lw)
            step += 1
            if a.max_train_steps is not None:
                if device.type == "cuda":
                    torch.cuda.synchronize(device)
                with open(time_file, "a") as f:
                    f.write(f"{step},{time.time() - t0:.8f},{gpu_name}\n")
            if step >= max_steps:
                break
        if a.max_train_steps is None:
            with open(time_file, "a") as f:
                f.write(f"{step // steps_per_epoch},{time.time() - t_epoch:.8f},{gpu_name}\n")
    unet.save_attn_procs(model_outdir)
    print(f"LoRA weights saved to {model_outdir}; last loss {float(loss):.5f}")
    return True


if __name__ == "__main__":
    main(parse_args())

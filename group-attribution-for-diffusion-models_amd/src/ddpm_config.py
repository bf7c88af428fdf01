"""Configuration registry with the same public names and values as the
reference's src/ddpm_config.py (DDPMConfig :13-602, PromptConfig :605-619,
LoraTrainingConfig :622-642, LoraUnlearningConfig :645-653,
LoraSparseUnlearningConfig :656-672, generation/behaviour configs :675-697,
DatasetStats :700-703).  The values are *data*; they are rebuilt here from a few
composable pieces and checked entry-by-entry against a fixture dumped from the
reference registry (tests/golden/configs.json, tests/test_host_logic.py)."""
import os

import src.constants as constants

_METHODS6 = ("retrain", "prune_fine_tune", "ga", "gd", "gd_u", "esd")


def _per_method(names, values):
    return dict(zip(names, values))


def _ddpm_scheduler(full=True):
    cfg = {"_class_name": "DDPMScheduler", "_diffusers_version": "0.24.0"}
    if full:
        cfg.update(beta_end=0.02, beta_schedule="linear", beta_start=0.0001, clip_sample=True,
                   clip_sample_range=1.0, dynamic_thresholding_ratio=0.995, num_train_timesteps=1000,
                   prediction_type="epsilon", sample_max_value=1.0, steps_offset=0, thresholding=False,
                   timestep_spacing="leading", trained_betas=None, variance_type="fixed_large")
    else:
        cfg["num_train_timesteps"] = 1000
    return cfg


def _ddim_scheduler(beta_start, beta_end, schedule, **extra):
    cfg = {"_class_name": "DDIMScheduler", "_diffusers_version": "0.0.4", "beta_end": beta_end,
           "beta_schedule": schedule, "beta_start": beta_start, "clip_sample": False,
           "num_train_timesteps": 1000, "trained_betas": None}
    cfg.update(extra)
    return cfg


def _cifar_unet():
    """google/ddpm-cifar10-32 topology: 35 746 307 parameters."""
    return {
        "_class_name": "UNet2DModel", "_diffusers_version": "0.24.0", "act_fn": "silu", "add_attention": True,
        "attention_head_dim": None, "attn_norm_num_groups": None, "block_out_channels": [128, 256, 256, 256],
        "center_input_sample": False, "class_embed_type": None,
        "down_block_types": ["DownBlock2D", "AttnDownBlock2D", "DownBlock2D", "DownBlock2D"],
        "downsample_padding": 0, "downsample_type": "conv", "dropout": 0.0, "flip_sin_to_cos": False,
        "freq_shift": 1, "in_channels": 3, "layers_per_block": 2, "mid_block_scale_factor": 1,
        "norm_eps": 1e-06, "norm_num_groups": 32, "num_class_embeds": None, "num_train_timesteps": None,
        "out_channels": 3, "resnet_time_scale_shift": "default", "sample_size": 32,
        "time_embedding_type": "positional",
        "up_block_types": ["UpBlock2D", "UpBlock2D", "AttnUpBlock2D", "UpBlock2D"], "upsample_type": "conv",
    }


def _ldm_unet(cls_name, boc, down, up, head_dim, sample_size, in_ch, positional=True):
    cfg = {"_class_name": cls_name, "_diffusers_version": "0.0.4", "act_fn": "silu",
           "attention_head_dim": head_dim, "block_out_channels": boc, "center_input_sample": False,
           "down_block_types": down, "downsample_padding": 1, "flip_sin_to_cos": True, "freq_shift": 0,
           "in_channels": in_ch, "layers_per_block": 2, "mid_block_scale_factor": 1, "norm_eps": 1e-05,
           "norm_num_groups": 32, "out_channels": in_ch, "sample_size": sample_size, "up_block_types": up}
    if positional:
        cfg["time_embedding_type"] = "positional"
    return cfg


def _cifar_like(dataset, methods, steps, ckpt, sample):
    return {
        "dataset": dataset, "image_size": 32,
        "optimizer_config": {"class_name": "Adam", "kwargs": {"lr": 1e-4}},
        "lr_scheduler_config": {"name": "constant", "kwargs": {"num_warmup_steps": 0}},
        "batch_size": 128,
        "training_steps": _per_method(methods, steps), "ckpt_freq": _per_method(methods, ckpt),
        "sample_freq": _per_method(methods, sample),
        "n_samples": 64, "unet_config": _cifar_unet(), "scheduler_config": _ddpm_scheduler(),
    }


class DDPMConfig:
    """DDPM / LDM configurations, keyed as in the reference."""

    _m5 = ("retrain", "prune_fine_tune", "ga", "gd", "esd")
    cifar_config = _cifar_like("cifar", _m5, (200000, 200000, 2000, 4000, 5000),
                               (10000, 10000, 400, 400, 1000), (200000, 200000, 2000, 4000, 5000))
    _m6if = ("retrain", "prune_fine_tune", "ga", "gd", "esd", "if")
    cifar2_config = _cifar_like("cifar", _m6if, (20000, 10000, 2000, 4000, 5000, 1),
                                (10000, 10000, 400, 400, 1000, 1), (2000, 2000, 400, 400, 100, 20))
    _m7 = ("retrain", "prune_fine_tune", "ga", "gd", "gd_u", "esd", "iu")
    cifar100_config = _cifar_like("cifar100", _m7, (20000, 10000, 40, 1000, 1000, 5000, 1),
                                  (400, 5000, 400, 500, 500, 1000, 1), (2000, 2000, 400, 500, 4000, 100, 20))
    _m6iu = ("retrain", "prune_fine_tune", "ga", "gd", "esd", "iu")
    cifar100_f_config = _cifar_like("cifar100_f", _m6iu, (20000, 20000, 40, 4000, 5000, 1),
                                    (10000, 5000, 400, 500, 1000, 1), (2000, 2000, 400, 500, 100, 20))

    celeba_config = {
        "dataset": "celeba", "image_size": 256,
        "optimizer_config": {"class_name": "AdamW", "kwargs": {"lr": 1.0e-4, "weight_decay": 0.0}},
        "lr_scheduler_config": {"name": "constant", "kwargs": {"num_warmup_steps": 0}},
        "batch_size": 32,
        "training_steps": _per_method(_METHODS6, (20000, 20000, 5, 500, 500, 500)),
        "ckpt_freq": _per_method(_METHODS6, (5000, 5000, 1, 500, 500, 100)),
        "sample_freq": _per_method(_METHODS6, (200000, 200000, 1, 40000, 5000, 100)),
        "n_samples": 4,
        "unet_config": _ldm_unet("UNet2DModel", [224, 448, 672, 896],
                                 ["DownBlock2D"] + ["AttnDownBlock2D"] * 3,
                                 ["AttnUpBlock2D"] * 3 + ["UpBlock2D"], 32, 64, 3),
        "scheduler_config": _ddim_scheduler(0.0015, 0.0195, "scaled_linear"),
        "vqvae_config": {
            "_class_name": "VQModel", "_diffusers_version": "0.1.2", "act_fn": "silu",
            "block_out_channels": [128, 256, 512], "down_block_types": ["DownEncoderBlock2D"] * 3,
            "in_channels": 3, "latent_channels": 3, "layers_per_block": 2, "num_vq_embeddings": 8192,
            "out_channels": 3, "sample_size": 256, "up_block_types": ["UpDecoderBlock2D"] * 3,
        },
    }

    _m4 = ("retrain", "ga", "gd", "esd")
    mnist_config = {
        "dataset": "mnist", "image_size": 28,
        "optimizer_config": {"class_name": "Adam", "kwargs": {"lr": 1e-3, "weight_decay": 0.0}},
        "lr_scheduler_config": {"name": "constant", "kwargs": {"num_warmup_steps": 0}},
        "batch_size": 64,
        "training_steps": _per_method(_m4, (100, 5, 10, 100)),
        "ckpt_freq": _per_method(_m4, (2, 1, 1, 20)),
        "sample_freq": _per_method(_m4, (20, 1, 1, 20)),
        "n_samples": 500,
        "trained_model": "/projects/leelab/mingyulu/data_att/results/mnist/retrain/models/full/steps_00065660.pt",
        "unet_config": {
            "_class_name": "UNet2DModel", "_diffusers_version": "0.24.0",
            "block_out_channels": [128, 128, 256, 512],
            "down_block_types": ["DownBlock2D", "DownBlock2D", "AttnDownBlock2D", "DownBlock2D"],
            "in_channels": 1, "layers_per_block": 2, "out_channels": 1, "sample_size": 32,
            "up_block_types": ["UpBlock2D", "AttnUpBlock2D", "UpBlock2D", "UpBlock2D"],
        },
        "scheduler_config": _ddpm_scheduler(full=False),
    }

    imagenette_config = {
        "dataset": "imagenette", "image_size": 256,
        "optimizer_config": {"class_name": "AdamW", "kwargs": {"lr": 1e-4, "weight_decay": 1e-6}},
        "lr_scheduler_config": {"name": "constant", "kwargs": {"num_warmup_steps": 0}},
        "batch_size": 64,
        "training_steps": _per_method(_m4, (50000, 5, 10, 150)),
        "ckpt_freq": _per_method(_m4, (2500, 1, 1, 50)),
        "sample_freq": _per_method(_m4, (2500, 1, 1, 50)),
        "n_samples": 60,
        "unet_config": _ldm_unet("UNet2DConditionModel", [320, 640, 1280, 1280],
                                 ["CrossAttnDownBlock2D"] * 3 + ["DownBlock2D"],
                                 ["UpBlock2D"] + ["CrossAttnUpBlock2D"] * 3, 8, 32, 4, positional=False),
        "scheduler_config": _ddim_scheduler(0.00085, 0.012, "linear", timestep_values=None),
    }


class PromptConfig:
    """Prompts per ArtBench style."""

    artbench_config = {
        "art_nouveau": "an Art Nouveau painting", "baroque": "a Baroque painting",
        "expressionism": "an Expressionist painting", "impressionism": "an Impressionist painting",
        "post_impressionism": "a Post-Impressionist painting", "realism": "a Realist painting",
        "renaissance": "a painting from the Renaissance", "romanticism": "a Romanticist painting",
        "surrealism": "a Surrealist painting", "ukiyo_e": "a ukiyo-e print",
    }


_MINI_SD = "lambdalabs/miniSD-diffusers"
_ARTBENCH = os.path.join(constants.OUTDIR, "seed42", "artbench_post_impressionism")


class LoraTrainingConfig:
    artbench_post_impressionism_config = {
        "pretrained_model_name_or_path": _MINI_SD, "resolution": 256, "train_batch_size": 64,
        "dataloader_num_workers": 4, "center_crop": True, "random_flip": True, "num_train_epochs": 200,
        "learning_rate": 3e-4, "lr_scheduler": "cosine", "adam_weight_decay": 1e-6, "rank": 256,
        "cls_key": "style", "cls": "post_impressionism", "checkpointing_steps": 500,
        "resume_from_checkpoint": "latest", "checkpoints_total_limit": 1,
    }


class LoraUnlearningConfig:
    artbench_post_impressionism_config = {
        "lora_dir": os.path.join(_ARTBENCH, "retrain", "models", "full"), "max_train_steps": 200,
    }


class LoraSparseUnlearningConfig:
    artbench_post_impressionism_config = {
        "lora_dir": os.path.join(_ARTBENCH, "pruned_ft_ratio=0.5_lr=3e-05", "models", "full"),
        "lora_steps": 1580, "max_train_steps": 200,
    }


class TextToImageGenerationConfig:
    artbench_post_impressionism_config = {
        "pretrained_model_name_or_path": _MINI_SD, "resolution": 256, "dataset": "artbench",
        "cls": "post_impressionism",
    }


class TextToImageModelBehaviorConfig:
    artbench_post_impressionism_config = {
        "pretrained_model_name_or_path": _MINI_SD, "dataset": "artbench", "cls": "post_impressionism",
        "reference_lora_dir": os.path.join(_ARTBENCH, "retrain", "models", "full"), "no_duplicate": True,
    }


class DatasetStats:
    artbench_post_impressionism_stats = {"num_groups": 258}

"""Acceptance check of the jsonl schema: the reference's own reader, `lds.py::collect_data` (lds.py:182-266), is run
UNCHANGED on a database written by this build, and what it extracted is stored as a fixture.

Two stages in two processes (this build's `src` package and the reference's `src` package cannot share a sys.path):
  stage "write": the kept entry points main.py / unlearn.py (CPU oracle backend, toy 2-group dataset, BASELINE config 1)
                 train a base model and run 4 Shapley coalitions -> a jsonl db  (what tests/test_entrypoints_cpu.py does)
  stage "read":  import /root/reference/lds.py (placeholders only for its unrelated top-level imports: torchvision, pynvml,
                 src.constants, the vendored diffusers file), point its `create_dataset` at the same 128-item toy label
                 layout, call collect_data(db, {"dataset","removal_dist","method"}, by_class=True) and dump its outputs.
Run (build container only):  python tests/golden/make_lds_golden.py
Output: tests/golden/lds_collect.json = {"rows": [the db rows, keys the reader uses], "masks", "behaviors", "seeds"}
The Stable-Diffusion evaluator gets the same treatment: rows assembled by this build's
text_to_image/compute_model_behaviors.py::assemble_row ("write_sd") are read by the reference's
text_to_image/shapley_lds.py::collect_data ("read_sd") -> tests/golden/sd_lds_collect.json.
"""
import argparse
import json
import os
import subprocess
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
KEYS = ["dataset", "method", "removal_dist", "removal_seed", "exp_name", "gd_steps", "remaining_idx", "fid_value",
        "total_steps_time", "total_sampling_time"]


def stage_write(db):
    sys.path[:0] = [os.path.join(ROOT, "group-attribution-for-diffusion-models_amd"), ROOT, os.path.join(ROOT, "tests")]
    os.environ.setdefault("GAD_OUTDIR", os.path.join(os.path.dirname(db), "results"))
    import torch
    import oracle_backend as OB
    from src.ddpm_config import DDPMConfig
    from unconditional_generation import main as train_main
    from unconditional_generation import unlearn as unlearn_main
    cfg = {**DDPMConfig.cifar100_config}
    cfg["unet_config"] = dict(cfg["unet_config"], block_out_channels=[32, 32, 64, 64], norm_num_groups=8)
    cfg["n_samples"] = 4
    for k, v in (("training_steps", 2), ("sample_freq", 2), ("ckpt_freq", 1)):
        cfg[k] = dict(cfg[k], retrain=v)
    DDPMConfig.cifar100_config = cfg
    out = os.path.join(os.path.dirname(db), "results")
    a = train_main.parse_args(["--dataset", "toy2", "--method", "retrain", "--outdir", out, "--db", db + ".train",
                               "--batch_size", "8", "--num_inference_steps", "50", "--device", "cpu", "--log_freq", "1"])
    assert train_main.main(a, backend=OB)
    mdir = os.path.join(out, "toy2", "retrain", "models", "full")
    ck = torch.load(os.path.join(mdir, "ckpt_steps_00000002.pt"), weights_only=False)
    pdir = os.path.join(out, "toy2", "pruned", "models", "pruner=magnitude_pruning_ratio=0.3_threshold=0.05")
    os.makedirs(pdir, exist_ok=True)
    torch.save({"unet": ck["unet"], "unet_config": ck["unet_config"]}, os.path.join(pdir, "ckpt_steps_00000000.pt"))
    for seed in range(4):
        u = unlearn_main.parse_args(["--dataset", "toy2", "--method", "gd", "--removal_dist", "shapley",
                                     "--removal_seed", str(seed), "--load", mdir, "--outdir", out, "--db", db,
                                     "--gd_steps", "2", "--n_samples", "8", "--batch_size", "4",
                                     "--num_inference_steps", "50", "--model_behavior", "global",
                                     "--exp_name", f"gd_shapley_seed_{seed}", "--device", "cpu"])
        assert unlearn_main.main(u, backend=OB)


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, k):
        return _Anything()


def stage_read(db, out):
    def placeholder(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    sys.path.insert(0, REF)
    tv = placeholder("torchvision")
    tv.models = placeholder("torchvision.models", resnet18=_Anything)
    tv.transforms = placeholder("torchvision.transforms", PILToTensor=_Anything, Compose=_Anything, ToTensor=_Anything)
    tv.datasets = placeholder("torchvision.datasets", CIFAR10=object, CIFAR100=object, MNIST=object, ImageFolder=object)
    tv.datasets.folder = placeholder("torchvision.datasets.folder", default_loader=None)
    placeholder("pynvml")
    placeholder("src.constants", DATASET_DIR="/tmp/_ds", OUTDIR="/tmp/_out", LOGDIR="/tmp/_log", MAX_NUM_SAMPLE_IMAGES_TO_SAVE=64)
    placeholder("src.diffusers"), placeholder("src.diffusers.models")
    placeholder("src.diffusers.models.attention_processor", my_get_processor=None)
    import lds                                                   # the reference's evaluator, unmodified
    lds.args = argparse.Namespace(gd_steps=2)                    # collect_data reads the module-global args (lds.py:247)
    toy = [(None, i // 64) for i in range(128)]                  # 2 contributor groups x 64 items, as the toy2 dataset
    lds.create_dataset = lambda dataset_name, train: toy
    masks, behaviors, seeds = lds.collect_data(db, {"dataset": "toy2", "removal_dist": "shapley", "method": "gd"},
                                               "toy2", "fid_value", None, True)
    rows = [{k: r[k] for k in KEYS} for r in (json.loads(l) for l in open(db))]
    json.dump({"rows": rows, "masks": masks.tolist(), "behaviors": behaviors.tolist(), "seeds": seeds.tolist()},
              open(out, "w"), separators=(",", ":"))
    print("reference collect_data ->", masks.tolist(), behaviors.ravel().tolist(), seeds.tolist())


def sd_rows():
    """Three behaviour rows through this build's text_to_image/compute_model_behaviors.py::assemble_row (pure host code)."""
    sys.path[:0] = [os.path.join(ROOT, "group-attribution-for-diffusion-models_amd"), ROOT]
    from text_to_image import compute_model_behaviors as M
    rows = []
    for k, remaining in enumerate(([0, 2, 5, 7], [1, 2, 3, 4, 8, 9], [6])):
        args = M.parse_args(["--reference_lora_dir", "ref", "--lora_dir", f"lora_{k}", "--db", "db.jsonl", "--num_images", "2",
                             "--exp_name", f"retrain_artist_shapley_seed_{k}"])
        lists = {b: [0.1 * (k + 1) + 0.01 * i + 0.001 * j for i in range(2)] for j, b in enumerate(M.BEHAVIOURS)}
        times = {b: [1.0 + i for i in range(2)] for b in M.BEHAVIOURS}
        rows.append(M.assemble_row(args, lists, times, remaining, [i for i in range(10) if i not in remaining]))
    return rows


def stage_write_sd(db):
    with open(db, "w") as f:
        for r in sd_rows():
            f.write(json.dumps(r) + "\n")


def stage_read_sd(db, out):
    def placeholder(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    sys.path[:0] = [REF, os.path.join(REF, "text_to_image")]
    placeholder("pynvml")
    placeholder("src.constants", DATASET_DIR="/tmp/_ds", OUTDIR="/tmp/_out", LOGDIR="/tmp/_log", MAX_NUM_SAMPLE_IMAGES_TO_SAVE=64)
    placeholder("src.diffusers"), placeholder("src.diffusers.models")
    placeholder("src.diffusers.models.attention_processor", my_get_processor=None)
    import pandas as pd
    import shapley_lds                                            # the reference's SD evaluator, unmodified
    df = pd.read_json(db, lines=True)
    masks, avg = shapley_lds.collect_data(df, num_groups=10, model_behavior_key="aesthetic_score_avg", n_samples=None)
    local = shapley_lds.collect_data(df, num_groups=10, model_behavior_key="simple_loss", n_samples=2,
                                     collect_remaining_masks=False)
    seeds = df["exp_name"].str.split("seed_", expand=True)[1].astype(int).tolist()     # shapley_lds.py:160-162
    json.dump({"masks": masks.tolist(), "aesthetic_score_avg": avg.tolist(), "simple_loss": local.tolist(), "subset_seed": seeds},
              open(out, "w"), separators=(",", ":"))
    print("reference shapley_lds.collect_data ->", masks.tolist(), avg.ravel().tolist(), local.tolist(), seeds)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        {"write": stage_write, "read": stage_read, "write_sd": stage_write_sd, "read_sd": stage_read_sd}[sys.argv[1]](*sys.argv[2:])
    else:
        import tempfile
        tmp = tempfile.mkdtemp()
        db = os.path.join(tmp, "db.jsonl")
        subprocess.run([sys.executable, __file__, "write", db], check=True)
        subprocess.run([sys.executable, __file__, "read", db, os.path.join(HERE, "lds_collect.json")], check=True)
        sdb = os.path.join(tmp, "sd.jsonl")
        subprocess.run([sys.executable, __file__, "write_sd", sdb], check=True)
        subprocess.run([sys.executable, __file__, "read_sd", sdb, os.path.join(HERE, "sd_lds_collect.json")], check=True)

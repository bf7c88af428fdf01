"""The reference's evaluator `lds.py` run as __main__, UNCHANGED, on databases written by this build.

Build container only (the reference cannot travel).  Two processes, because this build's `src` package and the
reference's cannot share one sys.path:

  stage "write"  this build's kept entry points on the CPU oracle backend (no GPU here), dataset "cifar100" at
                 GAD_SYNTH_SCALE=0.032 (20 contributor classes x 16 synthetic images), a 4-stage toy U-Net:
                   main.py --method retrain                      -> the full model (+ its step-0 "null" checkpoint)
                   unlearn.py --method gd --removal_dist shapley -> train_db: 40 sFT coalitions       (exp_name toy_train)
                   main.py --method retrain --removal_dist datamodel + calculate_global_scores.py
                                                                 -> 3 test dbs x 8 retrained datamodel subsets (toy_test)
                   calculate_global_scores.py on the full / null checkpoints -> full_db / null_db
  stage "read"   runpy.run_path("/root/reference/lds.py", run_name="__main__") with sys.argv set as a user would; the
                 only accommodations are the ones tests/golden/make_lds_golden.py already makes (placeholder modules for
                 its unrelated top-level imports, `create_dataset` answering the same 320-item label layout) plus ONE
                 more: lds.py:298-321 hard-codes its test databases under /gscratch/aims/mingyulu/results_ming/...,
                 which this pipeline may not create (nothing outside the repository is written), so the `open` the
                 module sees maps that prefix onto the directory where stage "write" put the files - a chroot for
                 one path prefix; the module's source is untouched.
Output: tests/golden/lds_main.json = the database rows (keys the evaluator reads) + what lds.py printed
("Mean: .. (..)" per fit size).  tests/test_lds_cpu.py recomputes the same numbers from the rows with gad/lds.py."""
import io
import json
import os
import re
import subprocess
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
GSCRATCH = "/gscratch/aims/mingyulu/results_ming"
KEYS = ["dataset", "method", "removal_dist", "datamodel_alpha", "removal_seed", "exp_name", "gd_steps", "remaining_idx",
        "fid_value", "total_steps_time", "total_sampling_time"]
N_TRAIN, N_TEST, GD_STEPS = 40, 8, 3


def stage_write(work):
    sys.path[:0] = [os.path.join(ROOT, "group-attribution-for-diffusion-models_amd"), ROOT, os.path.join(ROOT, "tests")]
    os.environ["GAD_SYNTH_SCALE"] = "0.032"
    os.environ.setdefault("GAD_OUTDIR", os.path.join(work, "results"))
    import torch
    import oracle_backend as OB
    from src.ddpm_config import DDPMConfig
    from unconditional_generation import calculate_global_scores as score_main
    from unconditional_generation import main as train_main
    from unconditional_generation import unlearn as unlearn_main
    torch.set_num_threads(8)
    cfg = {**DDPMConfig.cifar100_config}
    cfg["unet_config"] = dict(cfg["unet_config"], block_out_channels=[32, 32, 64, 64], norm_num_groups=8)
    cfg["n_samples"], cfg["batch_size"] = 4, 32
    cfg["optimizer_config"] = dict(cfg["optimizer_config"], kwargs=dict(cfg["optimizer_config"]["kwargs"], lr=2e-3))
    for k, v in (("training_steps", 3), ("sample_freq", 100), ("ckpt_freq", 3)):
        cfg[k] = dict(cfg[k], retrain=v)
    DDPMConfig.cifar100_config = cfg
    samp = ["--n_samples", "16", "--batch_size", "8", "--num_inference_steps", "5", "--device", "cpu"]

    def train(out, extra):
        a = train_main.parse_args(["--dataset", "cifar100", "--method", "retrain", "--outdir", out, "--num_inference_steps", "5",
                                   "--device", "cpu", "--log_freq", "100"] + extra)
        assert train_main.main(a, backend=OB)

    def score(out, db, extra):
        a = score_main.parse_args(["--dataset", "cifar100", "--method", "retrain", "--outdir", out, "--db", db, "--use_ema",
                                   "--generate_samples"] + samp + extra)
        assert score_main.main(a, backend=OB)

    out = os.path.join(work, "results")
    train(out, ["--save_null_model", "--keep_all_ckpts"])
    score(out, os.path.join(work, "full.jsonl"), ["--exp_name", "toy_full"])
    score(out, os.path.join(work, "null.jsonl"), ["--exp_name", "toy_null", "--trained_steps", "0"])
    mdir = os.path.join(out, "cifar100", "retrain", "models", "full")
    ck = torch.load(os.path.join(mdir, "ckpt_steps_00000003.pt"), weights_only=False)
    pdir = os.path.join(out, "cifar100", "pruned", "models", "pruner=magnitude_pruning_ratio=0.3_threshold=0.05")
    os.makedirs(pdir, exist_ok=True)
    torch.save({"unet": ck["unet"], "unet_config": ck["unet_config"]}, os.path.join(pdir, "ckpt_steps_00000000.pt"))
    for seed in range(N_TRAIN):
        u = unlearn_main.parse_args(["--dataset", "cifar100", "--method", "gd", "--removal_dist", "shapley", "--removal_seed", str(seed),
                                     "--load", mdir, "--outdir", out, "--db", os.path.join(work, "train.jsonl"),
                                     "--gd_steps", str(GD_STEPS), "--model_behavior", "global", "--exp_name", "toy_train"] + samp)
        assert unlearn_main.main(u, backend=OB)
    tdir = os.path.join(work, "gscratch", "cifar100", "datamodel")
    os.makedirs(tdir, exist_ok=True)
    for db_seed in (42, 43, 44):
        o = os.path.join(work, f"results_test{db_seed}")
        for k in range(N_TEST):
            sub = ["--removal_dist", "datamodel", "--datamodel_alpha", "0.5", "--removal_seed", str(k)]
            train(o, sub + ["--opt_seed", str(db_seed)])
            score(o, os.path.join(tdir, f"retrain_global_behavior_seed{db_seed}.jsonl"), sub + ["--exp_name", "toy_test", "--seed", str(db_seed)])


class _Anything:
    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return self

    def __getattr__(self, k):
        return _Anything()


def stage_read(work, out_json):
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
    placeholder("src.constants", DATASET_DIR="/tmp/_ds", OUTDIR="/tmp/_out", LOGDIR="/tmp/_log", MAX_NUM_SAMPLE_IMAGES_TO_SAVE=64,
                DATASET=["cifar", "cifar100", "celeba"], METHOD=["retrain", "gd", "gd_u"])
    placeholder("src.diffusers"), placeholder("src.diffusers.models")
    placeholder("src.diffusers.models.attention_processor", my_get_processor=None)
    import src.datasets as ref_datasets                           # the reference's module (its samplers are used by lds.py)
    toy = [(None, i // 16) for i in range(320)]                   # cifar100 at GAD_SYNTH_SCALE=0.032: labels i // 16
    ref_datasets.create_dataset = lambda dataset_name, train: toy
    import builtins
    import runpy
    real_open = builtins.open

    def mapped_open(path, *a, **k):                               # chroot for the one hard-coded prefix (lds.py:298-321)
        if isinstance(path, str) and path.startswith(GSCRATCH):
            path = os.path.join(work, "gscratch") + path[len(GSCRATCH):]
        return real_open(path, *a, **k)

    argv = ["lds.py", "--dataset", "cifar100", "--removal_dist", "shapley", "--method", "gd", "--by_class",
            "--train_db", os.path.join(work, "train.jsonl"), "--train_exp_name", "toy_train",
            "--test_db", os.path.join(work, "unused_test.jsonl"), "--test_exp_name", "toy_test", "--datamodel_alpha", "0.5",
            "--null_db", os.path.join(work, "null.jsonl"), "--full_db", os.path.join(work, "full.jsonl"),
            "--max_train_size", str(N_TRAIN), "--num_test_subset", str(N_TEST), "--model_behavior_key", "fid_value",
            "--gd_steps", str(GD_STEPS)]
    buf = io.StringIO()
    old_argv, old_stdout = sys.argv, sys.stdout
    sys.argv, sys.stdout = argv, buf
    try:
        runpy.run_path(os.path.join(REF, "lds.py"), run_name="__main__", init_globals={"open": mapped_open})
    finally:
        sys.argv, sys.stdout = old_argv, old_stdout
    text = buf.getvalue()
    print(text)
    sizes = [int(m) for m in re.findall(r"Estimating scores with (\d+) subsets\.", text)]
    means = [(float(a), float(b)) for a, b in re.findall(r"Mean: (-?[\d.]+|nan) \((-?[\d.]+|nan)\)", text)]
    assert len(sizes) == len(means) and sizes, (sizes, means)

    def rows(path):
        return [{k: r.get(k) for k in KEYS} for r in (json.loads(l) for l in open(path))]
    tdir = os.path.join(work, "gscratch", "cifar100", "datamodel")
    json.dump({"argv": argv[1:], "fit_sizes": sizes, "lds_mean_ci": means,
               "printed": [l for l in text.splitlines() if l.startswith(("Mean", "Confidence", "Estimating"))],
               "train": rows(os.path.join(work, "train.jsonl")), "full": rows(os.path.join(work, "full.jsonl")),
               "null": rows(os.path.join(work, "null.jsonl")),
               "test": {str(s): rows(os.path.join(tdir, f"retrain_global_behavior_seed{s}.jsonl")) for s in (42, 43, 44)}},
              open(out_json, "w"), separators=(",", ":"))


# ------------------------------------------------------------------------------------------------------------------
# text_to_image/shapley_lds.py __main__ (the SD evaluator), same treatment.  Its databases are behaviour rows of
# compute_model_behaviors.py; running 400 SD coalitions on the CPU oracle is out of reach, so the rows go through this
# build's row assembler (`assemble_row`, the code that writes every SD row) with seeded synthetic behaviours over the
# 258 artists and this build's coalition samplers - the grammar the evaluator reads is the product's, the numbers are
# a seeded linear model + noise.  `sd_lds_rows()` is deterministic: tests/test_lds_cpu.py regenerates the same rows.
# ------------------------------------------------------------------------------------------------------------------
SD_GSCRATCH = "/gscratch/aims/diffusion-attr"
SD_N_GROUPS, SD_FIT, SD_TEST = 258, 120, 30


def sd_lds_rows():
    sys.path[:0] = [p for p in (os.path.join(ROOT, "group-attribution-for-diffusion-models_amd"), ROOT) if p not in sys.path]
    import numpy as np
    from src.datasets import remove_data_by_datamodel, remove_data_by_shapley
    from text_to_image import compute_model_behaviors as M
    units = list(range(SD_N_GROUPS))
    rng = np.random.RandomState(7)
    w = rng.standard_normal(SD_N_GROUPS) * 0.05                      # ground-truth contribution of every artist

    def row(exp_name, remaining, noise_seed):
        r = np.random.RandomState(noise_seed)
        base = 5.0 + float(w[np.asarray(remaining, dtype=int)].sum()) if len(remaining) else 5.0
        args = M.parse_args(["--reference_lora_dir", "ref", "--lora_dir", exp_name, "--db", "db.jsonl", "--num_images", "2",
                             "--exp_name", exp_name])
        lists = {b: [base + 0.02 * r.standard_normal() + 0.001 * j for _ in range(2)] for j, b in enumerate(M.BEHAVIOURS)}
        times = {b: [1.0, 2.0] for b in M.BEHAVIOURS}
        rem = [int(i) for i in remaining]
        return M.assemble_row(args, lists, times, rem, [i for i in units if i not in set(rem)])
    out = {"fit": [row(f"sparse_gd_artist_shapley_seed_{k}", remove_data_by_shapley(units, k)[0], 1000 + k) for k in range(SD_FIT)],
           "baseline_fit": [row(f"retrain_artist_shapley_seed_{k}", remove_data_by_shapley(units, k)[0], 2000 + k) for k in range(SD_FIT)],
           "full": [row("retrain_full", units, 1)], "null": [row("pretrained", [], 2)], "test": {}}
    for s in (42, 43, 44):
        out["test"][str(s)] = [row(f"retrain_artist_datamodel_alpha=0.5_seed_{k}",
                                   remove_data_by_datamodel(units, alpha=0.5, seed=k)[0], 100 * s + k) for k in range(SD_TEST)]
    return out


def stage_write_sd(work):
    rows = sd_lds_rows()

    def dump(path, rs):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            for r in rs:
                f.write(json.dumps(r) + "\n")
    for k in ("fit", "baseline_fit", "full", "null"):
        dump(os.path.join(work, f"sd_{k}.jsonl"), rows[k])
    for s, rs in rows["test"].items():
        dump(os.path.join(work, "gscratch_sd", f"seed{s}", "artbench_post_impressionism", "retrain_artist_datamodel_alpha=0.5.jsonl"), rs)


def stage_read_sd(work, out_json):
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
    import runpy
    real_read_json = pd.read_json

    def mapped_read_json(path, *a, **k):                          # shapley_lds.py:158-166 hard-codes /gscratch/aims/diffusion-attr
        if isinstance(path, str) and path.startswith(SD_GSCRATCH):
            path = os.path.join(work, "gscratch_sd") + path[len(SD_GSCRATCH):]
        return real_read_json(path, *a, **k)
    pd.read_json = mapped_read_json
    outdir = os.path.join(work, "sd_out")
    os.makedirs(outdir, exist_ok=True)
    argv = ["shapley_lds.py", "--fit_db", os.path.join(work, "sd_fit.jsonl"), "--baseline_fit_db", os.path.join(work, "sd_baseline_fit.jsonl"),
            "--null_db", os.path.join(work, "sd_null.jsonl"), "--full_db", os.path.join(work, "sd_full.jsonl"),
            "--test_size", str(SD_TEST), "--fit_size", "60", str(SD_FIT), "--model_behavior_key", "aesthetic_score_avg",
            "--output_dir", outdir, "--outfile_prefix", "toy"]
    buf = io.StringIO()
    old_argv, old_stdout = sys.argv, sys.stdout
    sys.argv, sys.stdout = argv, buf
    try:
        runpy.run_path(os.path.join(REF, "text_to_image", "shapley_lds.py"), run_name="__main__")
    finally:
        sys.argv, sys.stdout = old_argv, old_stdout
    text = buf.getvalue()
    print(text)
    import numpy as np
    lds = [(float(a), float(b)) for a, b in re.findall(r"\tLDS: (-?[\d.]+) \((-?[\d.]+)\)", text)]
    base = [(float(a), float(b)) for a, b in re.findall(r"Baseline LDS: (-?[\d.]+) \((-?[\d.]+)\)", text)]
    assert len(lds) == 2 and len(base) == 2, text
    attrs = np.load(os.path.join(outdir, f"artist_toy_fit_size={SD_FIT}.npy"))
    rank = np.load(os.path.join(outdir, f"all_generated_images_artist_rank_toy_fit_size={SD_FIT}.npy"))
    json.dump({"argv": argv[1:], "fit_sizes": [60, SD_FIT], "lds_mean_ci": lds, "baseline_lds_mean_ci": base,
               "printed": [l for l in text.splitlines() if "LDS" in l or "fit size" in l],
               "attrs_shape": list(attrs.shape), "attrs_first8": attrs[:8, 0].tolist(), "rank_first16": rank[:16].tolist()},
              open(out_json, "w"), separators=(",", ":"))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        {"write": stage_write, "read": stage_read, "write_sd": stage_write_sd, "read_sd": stage_read_sd}[sys.argv[1]](*sys.argv[2:])
    else:
        import tempfile
        work = tempfile.mkdtemp(prefix="lds_main_")
        subprocess.run([sys.executable, __file__, "write", work], check=True)
        subprocess.run([sys.executable, __file__, "read", work, os.path.join(HERE, "lds_main.json")], check=True)
        subprocess.run([sys.executable, __file__, "write_sd", work], check=True)
        subprocess.run([sys.executable, __file__, "read_sd", work, os.path.join(HERE, "sd_lds_main.json")], check=True)

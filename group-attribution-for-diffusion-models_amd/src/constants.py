"""Site constants (the reference keeps this file git-ignored and user-supplied:
README.md:23-26 lists DATASET_DIR/OUTDIR/LOGDIR/MAX_NUM_SAMPLE_IMAGES_TO_SAVE; the
code also reads DATASET, METHOD - main.py:51,95).  Here they are environment
driven so the same tree runs in the build container and on the GPU box."""
import os

_ROOT = os.environ.get("GAD_ROOT", os.path.join(os.path.expanduser("~"), ".cache", "gad_amd"))
DATASET_DIR = os.environ.get("GAD_DATASET_DIR", os.path.join(_ROOT, "datasets"))
OUTDIR = os.environ.get("GAD_OUTDIR", os.path.join(_ROOT, "results"))
LOGDIR = os.environ.get("GAD_LOGDIR", os.path.join(_ROOT, "logs"))
MAX_NUM_SAMPLE_IMAGES_TO_SAVE = 64

DATASET = ["cifar", "cifar2", "cifar100", "cifar100_f", "celeba", "mnist", "imagenette"]
METHOD = ["retrain", "prune_fine_tune", "gd", "gd_u", "ga", "ga_u", "esd", "iu", "lora", "lora_u"]
REMOVAL_DIST = ["uniform", "shapley", "shapley_uniform", "datamodel", "loo", "add_one_in"]

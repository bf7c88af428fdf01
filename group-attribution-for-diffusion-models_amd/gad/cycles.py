"""Coalition cycles of BASELINE configs 3-5 for the one-coalition-per-GPU scheduler (`gad.coalition.run_sharded`,
`python -m gad.launch --cycle ...`): each `run_coalition(seed)` is one in-process run of the kept entry point(s) the
reference would have submitted as one SLURM array task (text_to_image/experiments/setup_unlearn_commands.py:160-214;
unconditional_generation/experiments/... for CelebA), so the rows are byte-for-byte what those entry points write.

  CelebaCycle   config 3: unconditional_generation/unlearn.py --dataset celeba --method gd --removal_dist shapley
                (sFT on the remaining celebrities, latent sampling, entropy / cluster_count / cluster_proportions)
  SDLoRACycle   configs 4 / 5: text_to_image/train_text_to_image_lora.py on the coalition (LoRA sFT from the full or the
                pruned LoRA) followed by text_to_image/compute_model_behaviors.py against the reference LoRA
                (per-image behaviours + aesthetic / CLIP quantiles)

The scalar behaviours travel in the record's `extra` slots through the final all_gather; the full row (argument dump,
index lists, per-image columns) is appended to the owning rank's shard the moment the coalition finishes."""
from __future__ import annotations

import json
import os
import tempfile
from typing import List, Optional, Sequence

import torch

from .coalition import CoalitionRecord

_NAN = float("nan")


def _last_row(path: str) -> dict:
    with open(path) as f:
        lines = [l for l in f if l.strip()]
    return json.loads(lines[-1])


class _EntryPointCycle:
    extra_keys: Sequence[str] = ()
    n_groups = 0

    def __init__(self, device):
        self.device = torch.device(device)
        self._rows = {}
        self._tmp = tempfile.mkdtemp(prefix="gad_cycle_")

    def jsonl_row(self, rec: CoalitionRecord) -> Optional[dict]:
        return self._rows.get(rec.removal_seed)

    def _record(self, seed, row, remaining_groups, fid=_NAN) -> CoalitionRecord:
        row.setdefault("removal_seed", seed)                  # the scheduler's key (SD rows carry the seed only in exp_name)
        self._rows[seed] = row
        rem, rmv = row.get("remaining_idx") or [], row.get("removed_idx") or row.get("removal_idx") or []
        return CoalitionRecord(seed, len(rem), len(rmv), fid, _NAN, float(row.get("total_steps_time", _NAN)),
                               float(row.get("total_sampling_time", _NAN)), int(row.get("trained_steps", 0) or 0),
                               sorted(int(g) for g in remaining_groups), extra=[row.get(k) for k in self.extra_keys])


class CelebaCycle(_EntryPointCycle):
    """BASELINE config 3.  `base_args`: the unlearn.py flags shared by every coalition (--load, --outdir, --gd_steps,
    --n_samples, --batch_size, --num_inference_steps, --precompute_stage reuse, ...)."""
    NUM_CLUSTER = 20
    extra_keys = ["entropy"] + [f"cluster_count_{i}" for i in range(NUM_CLUSTER)]

    def __init__(self, device, base_args: Sequence[str]):
        super().__init__(device)
        from src.datasets import create_dataset
        self.base_args = list(base_args)
        self.dataset = create_dataset("celeba", train=True)
        self.groups = sorted(set(self.dataset.targets))
        self.n_groups = len(self.groups)
        self._gid = {g: i for i, g in enumerate(self.groups)}

    def run_coalition(self, seed: int, verbose=False) -> CoalitionRecord:
        from unconditional_generation import unlearn
        db = os.path.join(self._tmp, f"celeba_seed{seed}.jsonl")
        args = unlearn.parse_args(["--dataset", "celeba", "--method", "gd", "--removal_dist", "shapley", "--model_behavior", "global",
                                   "--removal_seed", str(seed), "--db", db, "--device", str(self.device),
                                   "--exp_name", f"gd_shapley_seed_{seed}"] + self.base_args)
        unlearn.main(args)
        row = _last_row(db)
        for i, c in enumerate(row.get("cluster_count", [])):
            row.setdefault(f"cluster_count_{i}", c)
        rec = self._record(seed, row, {self._gid[self.dataset.targets[i]] for i in row["remaining_idx"]})
        for i in range(self.NUM_CLUSTER):                      # the flattened copies were only for the record's extras
            row.pop(f"cluster_count_{i}", None)
        if verbose:
            print(f"[celeba coalition {seed}] |S|={rec.n_remaining} entropy {row.get('entropy')}", flush=True)
        return rec


class SDLoRACycle(_EntryPointCycle):
    """BASELINE configs 4 / 5.  `train_args`: the train_text_to_image_lora.py flags shared by every coalition (data dir,
    output dir, --method, --lora_dir / --lora_steps of the starting LoRA, --max_train_steps 200 ...); `behaviour_args`:
    the compute_model_behaviors.py flags (--reference_lora_dir, --num_images, --resolution, ...)."""
    extra_keys = ["aesthetic_score_0.5", "aesthetic_score_0.75", "aesthetic_score_0.9", "aesthetic_score_avg",
                  "clip_prompt_score_0.5", "clip_prompt_score_0.75", "clip_prompt_score_0.9", "clip_prompt_score_avg"]

    def __init__(self, device, train_args: Sequence[str], behaviour_args: Sequence[str], n_groups: int,
                 removal_unit="artist"):
        super().__init__(device)
        self.train_args, self.behaviour_args = list(train_args), list(behaviour_args)
        self.n_groups, self.removal_unit = n_groups, removal_unit

    def run_coalition(self, seed: int, verbose=False) -> CoalitionRecord:
        import time
        from text_to_image import compute_model_behaviors as M
        from text_to_image import train_text_to_image_lora as T
        t0 = time.time()
        targs = T.parse_args(self.train_args + ["--removal_dist", "shapley", "--removal_unit", self.removal_unit,
                                                "--removal_seed", str(seed), "--device", str(self.device)])
        T.main(targs)
        lora_dir = targs.model_outdir                         # set by main(): {output_dir}/{dataset}/{method}/models/{removal_dir}
        t_train = time.time() - t0
        db = os.path.join(self._tmp, f"sd_seed{seed}.jsonl")
        margs = M.parse_args(self.behaviour_args + ["--lora_dir", lora_dir, "--db", db, "--device", str(self.device),
                                                    "--exp_name", f"{targs.method}_{self.removal_unit}_shapley_seed_{seed}"])
        t1 = time.time()
        M.main(margs)
        row = _last_row(db)
        row.setdefault("total_steps_time", t_train)
        row.setdefault("total_sampling_time", time.time() - t1)
        row.setdefault("trained_steps", getattr(targs, "max_train_steps", None) or 0)
        rec = self._record(seed, row, row.get("remaining_idx") or [])
        if verbose:
            print(f"[sd coalition {seed}] |S|={rec.n_remaining} aesthetic_0.9 {row.get('aesthetic_score_0.9')}", flush=True)
        return rec

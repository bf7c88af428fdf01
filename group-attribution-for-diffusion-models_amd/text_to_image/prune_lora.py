"""Magnitude pruning of LoRA ranks (reference text_to_image/prune_lora.py:121-210), torch_pruning-free.

The reference scores every rank of every LoRALinearLayer with torch_pruning 1.3.2's MagnitudeImportance over
the (down out-channel, up in-channel) dependency group - p = 2, group reduction "mean", normalizer "mean"
(restated from the published torch_pruning algorithm; the package is not installable offline, so this
scoring is parity-unpinned) - then walks the ranks in ascending stable order, removing (module, rank) pairs
until the LoRA parameter count is <= pruning_ratio * total (:146-162), which leaves a different rank per
projection.  Works directly on `pytorch_lora_weights.safetensors`; writes the pruned file and `info.csv` into
the sibling directory `pruned_ratio={r}` (:190-209)."""
import argparse
import os

import numpy as np
import torch


def parse_args(argv=None):
    p = argparse.ArgumentParser(description="Pruning LoRA weights")
    p.add_argument("--lora_dir", type=str, required=True)
    p.add_argument("--pruning_ratio", type=float, default=0.5, help="fraction of LoRA parameters to keep")
    p.add_argument("--weight_name", type=str, default="pytorch_lora_weights.safetensors")
    return p.parse_args(argv)


def rank_scores(down: torch.Tensor, up: torch.Tensor) -> np.ndarray:
    """Group magnitude importance of each rank: mean of the squared L2 norms of down's row and up's column,
    normalised by the module's mean score."""
    local = torch.stack([down.double().pow(2).sum(dim=1), up.double().pow(2).sum(dim=0)]).mean(dim=0)
    return (local / local.mean()).numpy()


def select_pairs(modules, pruning_ratio):
    """modules: ordered {name: (down [r,in], up [out,r])} -> {name: sorted ranks to remove}."""
    nodes, scores, sizes = [], [], []
    for name, (down, up) in modules.items():
        s = rank_scores(down, up).tolist()
        r = down.shape[0]
        nodes += [(name, i) for i in range(r)]        # "down" nodes of the module ...
        scores += s
        sizes += [down.shape[1]] * r
        nodes += [(name, i) for i in range(r)]        # ... then its "up" nodes, same group scores
        scores += s
        sizes += [up.shape[0]] * r
    remaining = float(sum(sizes))
    target = pruning_ratio * remaining
    removed = set()
    for i in np.argsort(scores, kind="stable"):
        pair = nodes[i]
        if pair not in removed:
            removed.add(pair)
            remaining -= sizes[i] * 2                  # the pair is charged twice the node's size (:155-159)
        if remaining <= target:
            break
    out = {}
    for name, idx in removed:
        out.setdefault(name, []).append(idx)
    return {k: sorted(v) for k, v in out.items()}


def main(args):
    from safetensors.torch import load_file, save_file
    sd = load_file(os.path.join(args.lora_dir, args.weight_name))
    modules = {}
    for k in sd:                                       # file order == module order
        if k.endswith(".down.weight"):
            base = k[: -len(".down.weight")]
            modules[base] = (sd[k], sd[base + ".up.weight"])
    total = sum(d.numel() + u.numel() for d, u in modules.values())
    removed = select_pairs(modules, args.pruning_ratio)
    out = {}
    for base, (down, up) in modules.items():
        keep = [i for i in range(down.shape[0]) if i not in set(removed.get(base, []))]
        out[base + ".down.weight"] = down[keep].contiguous()
        out[base + ".up.weight"] = up[:, keep].contiguous()
    kept = sum(t.numel() for t in out.values())
    parts = args.lora_dir.rstrip("/").split("/")
    parts[-3] = f"pruned_ratio={args.pruning_ratio}"
    outdir = "/".join(parts)
    os.makedirs(outdir, exist_ok=True)
    save_file(out, os.path.join(outdir, args.weight_name))
    with open(os.path.join(outdir, "info.csv"), "w") as f:
        f.write("metric,value\n")
        f.write(f"lora_params,{total:.0f}\n")
        f.write(f"pruned_lora_params,{kept:.0f}\n")
        f.write(f"target_pruning_ratio,{args.pruning_ratio}\n")
        f.write(f"actual_pruning_ratio,{kept / total:.5f}\n")
    print(f"Pruned LoRA weights saved to {outdir} ({kept}/{total} parameters kept)")
    return outdir


if __name__ == "__main__":
    main(parse_args())
    print("Pruning done!")

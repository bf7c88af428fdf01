"""Small helpers (reference src/utils.py): checkpoint discovery by file name, norms, image grids."""
import glob
import math
import os
import re

import numpy as np
import torch


def get_max_steps(folder_path):
    """Largest N among `ckpt_steps_{N:0>8}.pt` in a directory, or None (reference :64-76)."""
    best = None
    for path in glob.glob(os.path.join(folder_path, "ckpt_steps_*.pt")):
        m = re.search(r"ckpt_steps_(\d+)\.pt$", os.path.basename(path))
        if m:
            best = int(m.group(1)) if best is None else max(best, int(m.group(1)))
    return best


def compute_grad_norm(model):
    """Global L2 norm of the gradients (reference :15-24)."""
    sq = [p.grad.detach().float().pow(2).sum() for p in model.parameters() if p.grad is not None]
    return float(torch.stack(sq).sum().sqrt()) if sq else 0.0


def compute_param_norm(model):
    sq = [p.detach().float().pow(2).sum() for p in model.parameters()]
    return float(torch.stack(sq).sum().sqrt())


def get_module(model, name):
    for part in name.split("."):
        model = getattr(model, part)
    return model


def save_image_grid(images, path, nrow=8, padding=2):
    """torchvision.utils.save_image stand-in: images [N,C,H,W] in [0,1] -> PNG grid."""
    from PIL import Image

    x = torch.as_tensor(images).detach().float().cpu().clamp(0, 1)
    n, c, h, w = x.shape
    ncol = min(nrow, n)
    nr = int(math.ceil(n / ncol))
    grid = torch.zeros(c, nr * (h + padding) + padding, ncol * (w + padding) + padding)
    for i in range(n):
        r, col = divmod(i, ncol)
        y0, x0 = padding + r * (h + padding), padding + col * (w + padding)
        grid[:, y0:y0 + h, x0:x0 + w] = x[i]
    arr = grid.mul(255).add_(0.5).clamp_(0, 255).permute(1, 2, 0).to(torch.uint8).numpy()
    if c == 1:
        arr = arr[:, :, 0]
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    Image.fromarray(arr).save(path)

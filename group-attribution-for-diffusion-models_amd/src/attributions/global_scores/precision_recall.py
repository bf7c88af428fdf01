"""Improved precision / recall arithmetic (reference precision_recall.py:54-72,208-295): k-th neighbour radii
of each manifold (fp16 storage, as the reference keeps them) and the two coverage tests.  Works on
whatever device the feature tensors live on (the cdist tiles are 10k x 10k in the reference call,
unlearn.py:811-820)."""
from collections import namedtuple

import torch

Manifold = namedtuple("Manifold", ["features", "kth"])


def compute_distance(row_features, col_features, row_batch_size, col_batch_size, device):
    rows = []
    for rb in row_features.split(row_batch_size, dim=0):
        # distances are accumulated in fp32 and stored in the features' dtype (fp16 in the reference call);
        # ROCm has no half cdist kernel and an fp16 accumulation would only add noise
        cols = [torch.cdist(rb.to(device).float().unsqueeze(0), cb.to(device).float().unsqueeze(0)).squeeze(0)
                .to(rb.dtype) for cb in col_features.split(col_batch_size, dim=0)]      # tiles stay on `device`
        rows.append(torch.cat(cols, dim=1))
    return torch.cat(rows, dim=0)


def compute_kth(features, nhood_size, row_batch_size, col_batch_size, device):
    kth = []
    for rb in features.split(row_batch_size, dim=0):
        d = compute_distance(rb, features, row_batch_size, col_batch_size, device)
        kth.append(d.to(torch.float32).kthvalue(nhood_size + 1, dim=1).values.to(torch.float16))  # +1: skip itself
    return torch.cat(kth)


def make_manifold(features, nhood_size=3, row_batch_size=10000, col_batch_size=10000, device="cpu"):
    features = features.to(torch.float16)            # the reference extracts VGG features in fp16
    return Manifold(features, compute_kth(features, nhood_size, row_batch_size, col_batch_size, device))


def calc_pr(manifold_1, manifold_2, row_batch_size, col_batch_size, device):
    """manifold_1 = generated, manifold_2 = reference -> (precision, recall)."""
    def covered(probe, target):
        hits = []
        for pb in probe.features.split(row_batch_size):
            d = compute_distance(pb, target.features, row_batch_size, col_batch_size, device)
            hits.append((d <= target.kth.to(d.device).unsqueeze(0)).any(dim=1))
        return torch.cat(hits).to(torch.float32).mean().item()
    return covered(manifold_1, manifold_2), covered(manifold_2, manifold_1)

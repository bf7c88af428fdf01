"""Model-behaviour score tail for the CIFAR configuration (reference
src/attributions/global_scores/fid_score.py:23-107).

The reference extracts pool3 features with pytorch_fid's InceptionV3, whose weights are
fetched from a URL (fid_score.py:28) and are unobtainable offline; the feature extractor is
therefore a *seeded random-weight* conv stack built from the same HIP operators (the tail is
~1 % of a coalition's FLOPs), while the score arithmetic - float64 mean / covariance and the
Frechet distance - is restated exactly (fid_score.py:104-105 and
pytorch_fid.calculate_frechet_distance)."""
from __future__ import annotations

import os

import numpy as np
import torch
import torch.nn as nn

from . import nn as gnn
from . import ops


class FeatureNet(nn.Module):
    """[B,3,32,32] in [0,1] -> [B,2048] features (GroupNorm/SiLU conv pyramid + global average pool)."""

    def __init__(self, dims=2048, seed=1234):
        super().__init__()
        rng = torch.random.get_rng_state()
        torch.manual_seed(seed)
        chans = [(3, 64, 1), (64, 128, 2), (128, 256, 2), (256, 512, 2)]
        self.convs = nn.ModuleList([gnn.Conv2d(ci, co, 3, stride=s, pad=(1, 1, 1, 1)) for ci, co, s in chans])
        self.norms = nn.ModuleList([gnn.GroupNorm(32, co, 1e-5) for _, co, _ in chans])
        self.head = gnn.Linear(512, dims)
        torch.random.set_rng_state(rng)
        self.dims, self.seed = dims, seed

    @torch.no_grad()
    def forward(self, images_nchw01):
        x = ops.nchw_to_nhwc_raw((images_nchw01.float() * 2 - 1).contiguous())
        for conv, norm in zip(self.convs, self.norms):
            x = norm(conv(x), silu=True)
        b, h, w, c = x.shape
        pooled = ops.colsum_raw(x.view(b * h * w, c), segments=b) / float(h * w)
        return self.head(pooled)


def extractor_tag(net) -> str:
    """What produced the features behind a score row.  The reference's rows come from InceptionV3 / VGG16 / BLIP with
    URL-fetched weights (fid_score.py:28, precision_recall.py:31, diversity_score.py:89); rows written under the seeded
    stand-in must never be mistaken for them, so every row carries this tag as `feature_extractor`."""
    return getattr(net, "tag", f"standin-seed{getattr(net, 'seed', '?')}-d{getattr(net, 'dims', '?')}")


class ScriptedExtractor(nn.Module):
    """A real extractor supplied as a TorchScript file ([B,3,H,W] in [0,1] -> [B,dims]; e.g. pytorch_fid's pool3 trunk
    scripted where its weights exist): `GAD_FEATURE_NET_TS=/path/extractor.pt`.  Runs with stock torch ops - the score
    tail is ~1 % of a coalition and off the hand-written path."""

    def __init__(self, path, device):
        super().__init__()
        import hashlib
        self.mod = torch.jit.load(path, map_location=device).eval()
        with open(path, "rb") as f:
            self.tag = f"torchscript:{os.path.basename(path)}:{hashlib.sha256(f.read()).hexdigest()[:12]}"
        with torch.no_grad():
            self.dims = int(self.mod(torch.zeros(1, 3, 32, 32, device=device)).shape[1])

    @torch.no_grad()
    def forward(self, images_nchw01):
        return self.mod(images_nchw01.float()).float()


def default_extractor(dims, device, seed=1234):
    path = os.environ.get("GAD_FEATURE_NET_TS")
    if path and os.path.exists(path):
        return ScriptedExtractor(path, device)
    return FeatureNet(dims, seed=seed).to(device)


from src.attributions.global_scores.fid_score import (calculate_frechet_distance as frechet_distance,  # noqa: E402,F401
                                                      compute_features_stats as feature_stats)


def feature_stats_torch(features: torch.Tensor):
    """float64 mean / unbiased covariance on the tensor's device (np.mean / np.cov(rowvar=False) semantics)."""
    f = features.double()
    mu = f.mean(dim=0)
    d = f - mu
    return mu, d.T @ d / (f.shape[0] - 1)


def frechet_distance_torch(mu1, sigma1, mu2, sigma2):
    """Same quantity as frechet_distance without the host sqrtm: Tr sqrt(S1 S2) = sum sqrt(eig(S1^1/2 S2 S1^1/2)),
    two symmetric float64 eigendecompositions on the tensors' device (the score-tail-on-device step of SURVEY
    section 8f-2; agrees with scipy's sqrtm route to ~1e-9 relative on well-conditioned inputs)."""
    diff = mu1 - mu2
    w, v = torch.linalg.eigh(sigma1)
    root1 = (v * w.clamp_min(0).sqrt()) @ v.T
    inner = root1 @ sigma2 @ root1
    ev = torch.linalg.eigvalsh((inner + inner.T) / 2).clamp_min(0)
    return float(diff @ diff + torch.trace(sigma1) + torch.trace(sigma2) - 2 * ev.sqrt().sum())


def compute_features(net: FeatureNet, images: torch.Tensor, batch_size: int, device) -> np.ndarray:
    """images [N,3,H,W] float in [0,1] (host or device) -> float64 [N, dims] (fid_score.py:74-102)."""
    out = np.empty((len(images), net.dims))
    for s in range(0, len(images), batch_size):
        f = net(images[s:s + batch_size].to(device))
        out[s:s + f.shape[0]] = f.double().cpu().numpy()
    return out


_REF_STATS = {}


def fid_against_dataset(images01, dataset, device, batch_size=512, feature_dims=2048):
    """calculate_fid (fid_score.py:23-71) with the training set as the reference distribution (its
    mu/sigma play the role of the precomputed stats.pkl, :42-58; cached per dataset object)."""
    net = _REF_STATS.get("net")
    if net is None:
        net = default_extractor(feature_dims, device)
        _REF_STATS["net"] = net
    key = id(dataset)
    if key not in _REF_STATS:
        ref = dataset.device_tensor(device).add_(1).div_(2)
        _REF_STATS[key] = feature_stats(compute_features(net, ref, max(batch_size, 256), device))
    mu, sigma = feature_stats(compute_features(net, images01, max(batch_size, 256), device))
    return frechet_distance(mu, sigma, *_REF_STATS[key])


def compute_features_torch(net: FeatureNet, images: torch.Tensor, batch_size: int, device) -> torch.Tensor:
    """Like compute_features but the [N, dims] fp32 feature matrix stays in HBM."""
    return torch.cat([net(images[s:s + batch_size].to(device)) for s in range(0, len(images), batch_size)], 0)


def global_scores_against_dataset(images01, dataset, device, batch_size=512, feature_dims=2048, nhood_size=3):
    """All four global behaviours unlearn.py writes for the CIFAR family (:807-837): fid_value, is, precision,
    recall - with the seeded stand-in extractor in place of Inception / VGG16 (URL-fetched weights):
    IS uses the softmax of the first 1000 feature dims as class probabilities, P/R the fp16 features.
    Features, float64 moments, the Frechet eigendecompositions and the P/R distance tiles all stay on the
    device; only the four scalars come back to the host."""
    from src.attributions.global_scores.inception_score import inception_score_from_probs
    from src.attributions.global_scores.precision_recall import calc_pr, make_manifold
    net = _REF_STATS.get("net")
    if net is None:
        net = default_extractor(feature_dims, device)
        _REF_STATS["net"] = net
    key = ("dev", id(dataset))
    if key not in _REF_STATS:
        ref = dataset.device_tensor(device).add_(1).div_(2)
        ref_f = compute_features_torch(net, ref, max(batch_size, 256), device)
        _REF_STATS[key] = (feature_stats_torch(ref_f), make_manifold(ref_f, nhood_size, 10000, 10000, device))
    (mu_r, sig_r), m_ref = _REF_STATS[key]
    gen_f = compute_features_torch(net, images01, max(batch_size, 256), device)
    mu, sig = feature_stats_torch(gen_f)
    fid = frechet_distance_torch(mu, sig, mu_r, sig_r)
    probs = torch.softmax(gen_f[:, :1000].double(), dim=1).cpu().numpy()
    is_value = inception_score_from_probs(probs)
    precision, recall = calc_pr(make_manifold(gen_f, nhood_size, 10000, 10000, device), m_ref, 10000, 10000, device)
    return {"fid_value": fid, "is": is_value, "precision": precision, "recall": recall,
            "feature_extractor": extractor_tag(net)}


def diversity_against_dataset(images01, dataset, device, num_cluster=20, batch_size=256, feature_dims=768, max_ref=2000):
    """The CelebA global behaviour unlearn.py writes (:787-803): entropy / cluster_count / cluster_proportions of
    calculate_diversity_score (diversity_score.py:82-188) - Ward clusters of the reference embeddings, generated samples
    assigned to the nearest cluster mean, log2 entropy of the proportions.  The BLIP-VQA vision tower (hub-fetched, :89-90)
    is replaced by the seeded stand-in extractor; the reference set ({OUTDIR}/celeba/cluster_imgs there) is the training
    set itself (its first `max_ref` items), mapped to [0,1] like the pipeline output.  Embeddings are computed on the
    device; the Ward linkage stays scipy on the host, as in the reference."""
    from src.attributions.global_scores.diversity_score import diversity_from_embeddings
    key = ("div_net", feature_dims)
    net = _REF_STATS.get(key)
    if net is None:
        net = FeatureNet(feature_dims, seed=4321).to(device)
        _REF_STATS[key] = net
    rkey = ("div_ref", id(dataset))
    if rkey not in _REF_STATS:
        idx = list(range(min(len(dataset), max_ref)))
        ref = dataset.device_tensor(device, idx).add(1).div(2).clamp(0, 1)
        f = compute_features_torch(net, ref, batch_size, device)
        _REF_STATS[rkey] = torch.nn.functional.normalize(f.double(), dim=1).cpu().numpy()
    emb_ref = _REF_STATS[rkey]
    f = compute_features_torch(net, images01.to(device), batch_size, device)
    emb_gen = torch.nn.functional.normalize(f.double(), dim=1).cpu().numpy()
    entropy, cluster_count, proportions, _, _ = diversity_from_embeddings(emb_ref, emb_gen, num_cluster)
    return {"entropy": entropy, "cluster_count": cluster_count, "cluster_proportions": proportions,
            "feature_extractor": extractor_tag(net)}

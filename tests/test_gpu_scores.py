"""The DEVICE score tail that writes the jsonl numbers (gad/scoring.py), fed from CUDA
tensors and compared with (i) tests/golden/scores.npz - what the reference's own ManifoldBuilder / calc_pr / eval_is
returned - and (ii) the host routes through the libraries the reference calls (np.cov, scipy sqrtm, scipy ward)."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
dev = torch.device("cuda:0")
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _feats(n, d, seed, shift=0.0, scale=1.0):
    rng = np.random.RandomState(seed)
    return (rng.randn(n, d) @ (rng.randn(d, d) / np.sqrt(d)) * scale + shift).astype(np.float32)


@pytest.mark.parametrize("d,n", [(64, 700), (2048, 4096)])
def test_device_moments_and_frechet_match_host_route(d, n):
    """feature_stats_torch / frechet_distance_torch on cuda vs np.mean / np.cov / scipy.linalg.sqrtm
    (fid_score.py:60-71,104-105).  Tolerances: moments 1e-10 relative (both float64), distance 1e-6 relative."""
    from gad.scoring import feature_stats_torch, frechet_distance_torch
    from src.attributions.global_scores.fid_score import calculate_frechet_distance, compute_features_stats
    a, b = _feats(n, d, 1), _feats(n - 100, d, 2, shift=0.15, scale=1.1)
    mu_a, sig_a = compute_features_stats(a)
    mu_b, sig_b = compute_features_stats(b)
    (ma, sa), (mb, sb) = feature_stats_torch(torch.from_numpy(a).to(dev)), feature_stats_torch(torch.from_numpy(b).to(dev))
    assert ma.is_cuda and sa.dtype == torch.float64
    np.testing.assert_allclose(ma.cpu().numpy(), mu_a, rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sa.cpu().numpy(), sig_a, rtol=1e-9, atol=1e-11)
    want = calculate_frechet_distance(mu_a, sig_a, mu_b, sig_b)
    got = frechet_distance_torch(ma, sa, mb, sb)
    assert abs(got - want) < 1e-6 * max(1.0, abs(want)), (got, want)
    assert abs(frechet_distance_torch(ma, sa, ma, sa)) < 1e-6 * float(torch.trace(sa))


def test_device_precision_recall_matches_the_reference_functions_output():
    """make_manifold / calc_pr with every tile on the GPU vs what the reference's ManifoldBuilder / calc_pr produced
    (scores.npz).  Radii: <= 2 fp16 ulps (the reference takes native-fp16 distances, here fp32 rounded once);
    precision / recall: at most two boundary samples may flip."""
    from src.attributions.global_scores.precision_recall import calc_pr, make_manifold
    z = np.load(os.path.join(GOLD, "scores.npz"))
    m_ref = make_manifold(torch.from_numpy(z["pr_ref"]).to(dev), 3, 128, 100, dev)
    m_gen = make_manifold(torch.from_numpy(z["pr_gen"]).to(dev), 3, 128, 100, dev)
    assert m_ref.features.is_cuda and m_ref.kth.is_cuda
    for got, want in ((m_ref.kth, z["kth_ref"]), (m_gen.kth, z["kth_gen"])):
        rel = np.abs(got.float().cpu().numpy() - want) / want
        assert rel.max() < 2.0 ** -9 and (rel == 0).mean() > 0.8
    p, r = calc_pr(m_gen, m_ref, 128, 100, dev)
    assert abs(p - float(z["precision"])) <= 2 / 200 + 1e-9 and abs(r - float(z["recall"])) <= 2 / 300 + 1e-9
    # and the device tiles agree with the same functions on the host, bit for bit in the radii
    c_ref = make_manifold(torch.from_numpy(z["pr_ref"]), 3, 128, 100, "cpu")
    assert (c_ref.kth.float() - m_ref.kth.float().cpu()).abs().max().item() <= float(np.abs(z["kth_ref"]).max()) * 2.0 ** -10


def test_device_inception_score_matches_reference_output():
    from src.attributions.global_scores.inception_score import inception_score_from_probs
    sys.path.insert(0, GOLD)
    from make_scores_golden import tiny_classifier
    z = np.load(os.path.join(GOLD, "scores.npz"))
    with torch.no_grad():
        logits = tiny_classifier().to(dev)(torch.from_numpy(z["is_images"]).to(dev))
        probs = torch.softmax(logits.double(), dim=1).cpu().numpy()          # the engine's device softmax (scoring.py)
    assert inception_score_from_probs(probs, splits=1) == pytest.approx(float(z["is_splits1"]), rel=1e-5)
    assert inception_score_from_probs(probs, splits=4) == pytest.approx(float(z["is_splits4"]), rel=1e-5)


def test_global_scores_against_dataset_equals_the_host_route():
    """The engine's on-device tail (features, float64 moments, eigh Frechet, P/R tiles in HBM) vs the reference-named
    host modules on the SAME features: fid to 1e-6 relative, precision / recall exactly, IS to 1e-9."""
    from gad import scoring
    from src.attributions.global_scores.fid_score import calculate_frechet_distance, compute_features_stats
    from src.attributions.global_scores.inception_score import inception_score_from_probs
    from src.attributions.global_scores.precision_recall import calc_pr, make_manifold
    from src.datasets import create_dataset
    ds = create_dataset("toy2", train=True)
    net = scoring.FeatureNet(256).to(dev)
    scoring._REF_STATS.clear()
    scoring._REF_STATS["net"] = net
    g = torch.Generator().manual_seed(0)
    gen = (ds.device_tensor("cpu")[:96].add(1).div(2) * 0.8 + 0.1 * torch.rand(96, 3, 32, 32, generator=g)).clamp(0, 1)
    got = scoring.global_scores_against_dataset(gen.to(dev), ds, dev, 64, 256)
    assert got["feature_extractor"].startswith("standin-seed1234")
    ref_f = scoring.compute_features_torch(net, ds.device_tensor(dev).add_(1).div_(2), 256, dev).cpu()
    gen_f = scoring.compute_features_torch(net, gen.to(dev), 256, dev).cpu()
    fid = calculate_frechet_distance(*compute_features_stats(gen_f.numpy()), *compute_features_stats(ref_f.numpy()))
    assert abs(got["fid_value"] - fid) < 1e-6 * max(1.0, abs(fid))
    p, r = calc_pr(make_manifold(gen_f, 3, 10000, 10000, "cpu"), make_manifold(ref_f, 3, 10000, 10000, "cpu"), 10000, 10000, "cpu")
    assert abs(got["precision"] - p) <= 1 / 96 + 1e-9 and abs(got["recall"] - r) <= 1 / len(ref_f) + 1e-9
    probs = torch.softmax(gen_f[:, :1000].double(), dim=1).numpy()
    assert got["is"] == pytest.approx(inception_score_from_probs(probs), rel=1e-9)
    scoring._REF_STATS.clear()


def test_device_diversity_matches_scipy_recomputation():
    """diversity_against_dataset (device embeddings -> Ward clusters -> entropy, diversity_score.py:122-171) vs an
    independent recomputation from the same embeddings with scipy's ward / fcluster and scipy.stats.entropy."""
    from scipy.cluster.hierarchy import fcluster, ward
    from scipy.spatial.distance import squareform
    from gad import scoring
    from src.datasets import create_dataset
    ds = create_dataset("toy2", train=True)
    scoring._REF_STATS.clear()
    g = torch.Generator().manual_seed(1)
    gen = (ds.device_tensor("cpu")[:80].add(1).div(2) + 0.05 * torch.randn(80, 3, 32, 32, generator=g)).clamp(0, 1)
    got = scoring.diversity_against_dataset(gen.to(dev), ds, dev, num_cluster=5, feature_dims=64)
    net = scoring._REF_STATS[("div_net", 64)]
    idx = list(range(min(len(ds), 2000)))
    e_ref = torch.nn.functional.normalize(scoring.compute_features_torch(net, ds.device_tensor(dev, idx).add(1).div(2).clamp(0, 1), 256, dev).double(), dim=1).cpu().numpy()
    e_gen = torch.nn.functional.normalize(scoring.compute_features_torch(net, gen.to(dev), 256, dev).double(), dim=1).cpu().numpy()
    sim = e_ref @ e_ref.T
    dist = sim.max() - sim
    np.fill_diagonal(dist, 0)
    labels = fcluster(ward(squareform(dist, checks=False)), 5, criterion="maxclust")
    d_gen = sim.max() - e_gen @ e_ref.T
    assigned = np.array([np.nanargmin([d_gen[i, labels == c].mean() if (labels == c).any() else np.nan for c in range(1, 6)]) + 1
                         for i in range(len(e_gen))])
    count = np.array([(assigned == c).sum() for c in range(1, 6)], dtype=float)
    prop = count / len(assigned)
    want = float(-np.sum(prop * np.log2(prop + np.finfo(float).eps)))
    assert got["cluster_count"] == count.tolist()
    assert got["entropy"] == pytest.approx(want, abs=1e-12)
    scoring._REF_STATS.clear()


def test_antithetic_timesteps_device_draw_follows_the_reference_formula():
    """main.py:684-696 on the device generator: the first B//2+1 entries are the uniform draw t1, the rest N-1-t1."""
    from gad.coalition import antithetic_timesteps
    from oracle import diffusers_ref as R
    for B in (128, 127, 2, 1):
        t = antithetic_timesteps(1000, B, dev, generator=torch.Generator(device=dev).manual_seed(B))
        assert t.dtype == torch.int64 and t.shape == (B,) and int(t.min()) >= 0 and int(t.max()) < 1000
        t1 = torch.randint(0, 1000, (B // 2 + 1,), device=dev, generator=torch.Generator(device=dev).manual_seed(B)).long()
        assert torch.equal(t, R.antithetic_timesteps(t1.cpu(), 1000, B).to(dev))


def test_device_loader_epoch_coverage_and_flip_statistics():
    """DeviceLoader = DataLoader(Subset(ds, idx), B, shuffle=True) + RandomHorizontalFlip (unlearn.py:373-379,
    datasets.py:444-457): every remaining index exactly once per epoch, short last batch kept, labels travel with their
    images, each sample is the original or its mirror image, flips ~ Bernoulli(1/2), order changes between epochs."""
    from gad.coalition import DeviceLoader
    from src.datasets import create_dataset
    ds = create_dataset("toy2", train=True)
    idx = [i for i in range(len(ds)) if i % 3 != 1]
    x_all = ds.device_tensor(dev, idx)
    torch.manual_seed(0)
    loader = DeviceLoader(ds, idx, 16, dev)
    assert len(loader) == (len(idx) + 15) // 16
    orders, flips, total = [], 0, 0
    for _ in range(4):
        seen = []
        sizes = []
        for xb, yb in loader:
            sizes.append(xb.shape[0])
            for img, lab in zip(xb, yb):
                same = (x_all == img).flatten(1).all(1)
                mirr = (x_all.flip(-1) == img).flatten(1).all(1)
                hit = (same | mirr).nonzero().flatten()
                assert hit.numel() >= 1
                j = int(hit[0])
                assert int(lab) == ds.targets[idx[j]]
                seen.append(j)
                if not bool(same[j]):
                    flips += 1
                total += 1
        assert sorted(seen) == list(range(len(idx)))
        assert sizes[:-1] == [16] * (len(sizes) - 1) and sizes[-1] == len(idx) - 16 * (len(sizes) - 1)
        orders.append(seen)
    assert orders[0] != orders[1] and orders[1] != orders[2]
    assert abs(flips / total - 0.5) < 4 * 0.5 / total ** 0.5                 # 4 sigma
    noflip = DeviceLoader(ds, idx, 16, dev, flip=False)
    for xb, _ in noflip:
        for img in xb:
            assert bool((x_all == img).flatten(1).all(1).any())

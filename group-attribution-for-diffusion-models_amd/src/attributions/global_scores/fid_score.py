"""FID score arithmetic (reference src/attributions/global_scores/fid_score.py:23-107).

The feature extractor is supplied by the caller (the reference's pytorch_fid InceptionV3 weights are
fetched from a URL, :28); everything after it - float64 statistics (:104-105) and the Frechet distance
(pytorch_fid.calculate_frechet_distance, called at :60-71) - is here.  `gad.scoring` holds the on-device
(eigh) route of the same quantity and is checked against this one."""
import numpy as np


def compute_features_stats(features: np.ndarray):
    """mu, sigma exactly as fid_score.compute_features_stats (:104-105): float64 mean and np.cov."""
    f = np.asarray(features, dtype=np.float64)
    return np.mean(f, axis=0), np.cov(f, rowvar=False)


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """d^2 = |mu1-mu2|^2 + Tr(s1 + s2 - 2 sqrt(s1 s2))  (pytorch_fid.calculate_frechet_distance)."""
    from scipy import linalg

    mu1, mu2 = np.atleast_1d(mu1), np.atleast_1d(mu2)
    sigma1, sigma2 = np.atleast_2d(sigma1), np.atleast_2d(sigma2)
    diff = mu1 - mu2
    covmean, _ = linalg.sqrtm(sigma1.dot(sigma2), disp=False)
    if not np.isfinite(covmean).all():
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError(f"Imaginary component {np.max(np.abs(covmean.imag))}")
        covmean = covmean.real
    return float(diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * np.trace(covmean))


def calculate_fid_from_features(features, mu_ref, sigma_ref):
    mu, sigma = compute_features_stats(np.asarray(features))
    return calculate_frechet_distance(mu, sigma, mu_ref, sigma_ref)

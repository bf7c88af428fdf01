"""FID score arithmetic (reference src/attributions/global_scores/fid_score.py:23-107).

The feature extractor is supplied by the caller (the reference's pytorch_fid InceptionV3 weights are
fetched from a URL, :28); everything after it - float64 statistics and the Frechet distance - is here."""
import numpy as np

from gad.scoring import feature_stats as compute_features_stats  # noqa: F401  (np.mean / np.cov, :104-105)
from gad.scoring import frechet_distance as calculate_frechet_distance  # noqa: F401


def calculate_fid_from_features(features, mu_ref, sigma_ref):
    mu, sigma = compute_features_stats(np.asarray(features))
    return calculate_frechet_distance(mu, sigma, mu_ref, sigma_ref)

"""TEST INFRASTRUCTURE ONLY (imported by tests/ alone).  Direct-loop restatement of the two scikit-image metrics the
reference calls at text_to_image/compute_model_behaviors.py:338-354 - `structural_similarity(im1, im2,
channel_axis=-1, data_range=255)` and `normalized_root_mse(image_true, image_test)` - to check the vectorised product
functions against.  scikit-image (unpinned in requirements.txt) is not installable here: PARITY UNPINNED; the algorithm
is the published one (Wang, Bovik, Sheikh, Simoncelli 2004) with skimage's documented defaults: win_size 7, uniform
window, K1 0.01, K2 0.03, sample covariance, 'reflect' borders cropped by (win_size-1)//2."""
import numpy as np


def ssim_loops(im1, im2, data_range=255.0, win=7, K1=0.01, K2=0.03):
    a, b = np.asarray(im1, dtype=np.float64), np.asarray(im2, dtype=np.float64)
    H, W, C = a.shape
    pad = (win - 1) // 2
    NP = win * win
    C1, C2 = (K1 * data_range) ** 2, (K2 * data_range) ** 2
    per_channel = []
    for c in range(C):
        vals = []
        for i in range(pad, H - pad):            # windows fully inside the image: the cropped region needs no border rule
            for j in range(pad, W - pad):
                x = a[i - pad:i + pad + 1, j - pad:j + pad + 1, c].ravel()
                y = b[i - pad:i + pad + 1, j - pad:j + pad + 1, c].ravel()
                ux, uy = x.mean(), y.mean()
                vx = ((x - ux) ** 2).sum() / (NP - 1)
                vy = ((y - uy) ** 2).sum() / (NP - 1)
                vxy = ((x - ux) * (y - uy)).sum() / (NP - 1)
                vals.append(((2 * ux * uy + C1) * (2 * vxy + C2)) / ((ux * ux + uy * uy + C1) * (vx + vy + C2)))
        per_channel.append(np.mean(vals))
    return float(np.mean(per_channel))


def nrmse_loops(image_true, image_test):
    t, x = np.asarray(image_true, dtype=np.float64).ravel(), np.asarray(image_test, dtype=np.float64).ravel()
    se = sum((ti - xi) ** 2 for ti, xi in zip(t.tolist(), x.tolist()))
    tt = sum(ti * ti for ti in t.tolist())
    return float(np.sqrt(se / len(t)) / np.sqrt(tt / len(t)))

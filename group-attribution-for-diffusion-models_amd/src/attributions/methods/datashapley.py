"""Shapley aggregation API (reference src/attributions/methods/datashapley.py).

Same call signatures and results as the reference so lds.py / shapley_lds.py
evaluate this tree's jsonl unchanged:
  data_shapley(dataset_size, x_train, y_train, v1, v0)   reference :8-48
  kernel_shap_ridge(...)                                  reference :51-83
  kernel_shap(...)                                        reference :86-133
Host-side numpy (d = 20..258 unknowns: microseconds, not a device kernel).
Checked against vectors produced by the reference functions
(tests/golden/shapley.npz)."""
import warnings

import numpy as np


def _gram(x_train, resid):
    n = len(x_train)
    return x_train.T @ x_train / n, x_train.T @ resid.reshape(-1, 1) / n


def data_shapley(dataset_size, x_train, y_train, v1, v0):
    """Closed-form KernelSHAP estimator, eq. (7) of Covert & Lee (AISTATS'21):
    least squares on the sampled coalitions subject to the efficiency constraint
    sum(coef) = v1 - v0, solved with a Lagrange correction on the pseudo-inverse."""
    a_hat, b_hat = _gram(x_train, y_train - v0)
    a_inv = np.linalg.pinv(a_hat)           # pinv: a_hat is singular for few coalitions
    ones = np.ones((dataset_size, 1))
    excess = ones.T @ a_inv @ b_hat - v1 + v0
    denom = ones.T @ a_inv @ ones
    coef = a_inv @ (b_hat - ones @ (excess / denom))
    coef[np.abs(coef) < 1e-10] = 0
    return coef


def _augmented(dataset_size, x_train, y_train, v1, v0, anchor_weight):
    """Append the grand (all ones) and null (all zeros) coalitions with a large weight."""
    X = np.concatenate((x_train, np.ones((1, dataset_size)), np.zeros((1, dataset_size))), axis=0)
    y = np.concatenate((y_train, np.asarray([v1, v0])), axis=0)
    w = np.concatenate((np.ones(len(x_train)), np.asarray([anchor_weight, anchor_weight])), axis=0)
    return X, y, w


def kernel_shap_ridge(dataset_size, x_train, y_train, v1, v0):
    from sklearn.linear_model import RidgeCV

    X, y, w = _augmented(dataset_size, x_train, y_train, v1, v0, 10000.0)
    return RidgeCV(alphas=np.linspace(1e-20, 1e-15, 5)).fit(w[:, None] * X, y).coef_


def kernel_shap(dataset_size, x_train, y_train, v1, v0):
    X, y, w = _augmented(dataset_size, x_train, y_train, v1, v0, 1e10)
    WX = w[:, None] * X
    try:
        return np.linalg.solve(X.T @ WX, WX.T @ y)
    except np.linalg.LinAlgError:
        warnings.warn("kernel_shap: normal equations singular; falling back to weighted least squares "
                      "(add coalitions or group contributors to avoid this).")
        rw = np.sqrt(w)
        return np.linalg.lstsq(rw[:, None] * X, rw * y, rcond=None)[0]

"""Linear datamodeling score (LDS) of Shapley attributions, as the reference's evaluator computes it
(lds.py:158-170 `evaluate_lds`, :268-452 `main` for `--removal_dist shapley`): fit per-behaviour attributions on the
training coalitions with `data_shapley` (full / null model behaviours as the efficiency constraint), predict every test
coalition's behaviour as mask @ attribution, Spearman-correlate with the measured behaviours (x100), average over
behaviours, then mean and 1.96 sigma / sqrt(n) over the test sets.

The reference's lds.py itself evaluates this build's jsonl databases unchanged (tests/golden/make_lds_main_golden.py
runs its __main__); this restatement exists so that the same number can be computed where the reference cannot travel
(the GPU box) - tests/test_lds_cpu.py pins it to what the reference's __main__ printed."""
from __future__ import annotations

import numpy as np


def masks_and_behaviours(rows, group_of_index, n_groups, key="fid_value"):
    """rows (dicts with removal_seed, remaining_idx, <key>) -> (masks [n, n_groups], behaviours [n, 1], seeds [n]); the
    by-class mask of lds.py:216-231 (a group is `remaining` if any of its items is), first row per seed wins (:243)."""
    masks, ys, seeds = [], [], []
    for r in rows:
        s = int(r["removal_seed"])
        if s in seeds:
            continue
        m = np.zeros(n_groups)
        m[sorted({int(group_of_index[i]) for i in r["remaining_idx"]})] = 1
        masks.append(m)
        ys.append([float(r[key])])
        seeds.append(s)
    return np.stack(masks), np.stack(ys), np.array(seeds)


def evaluate_lds(attrs_all, test_data_list):
    """lds.py:158-170; attrs_all[k] = attribution vector of behaviour k."""
    from scipy.stats import spearmanr
    lds_list = []
    for x_test, y_test in test_data_list:
        lds_list.append(np.mean([spearmanr(x_test @ attrs_all[k], y_test[:, k]).statistic * 100
                                 for k in range(len(attrs_all))]))
    return float(np.mean(lds_list)), float(np.std(lds_list) / np.sqrt(len(lds_list)) * 1.96)


def shapley_lds(train_masks, train_targets, test_data_list, full_targets, null_targets, n_fit=None, train_order=None):
    """LDS of `data_shapley` attributions fitted on the first n_fit training coalitions of `train_order` (default: all,
    in the given order) - the body of lds.py:394-452 for one subset size."""
    from src.attributions.methods.datashapley import data_shapley
    order = np.arange(len(train_masks)) if train_order is None else np.asarray(train_order)
    idx = order[:n_fit] if n_fit is not None else order
    x, y = train_masks[idx], train_targets[idx]
    attrs = [data_shapley(x.shape[-1], x, y[:, k], float(np.ravel(full_targets)[k]), float(np.ravel(null_targets)[k]))
             for k in range(y.shape[-1])]
    return evaluate_lds(attrs, test_data_list), attrs


def reference_train_order(train_masks, last_test_masks, seed=42):
    """The order in which lds.py:347-383 consumes the training coalitions: those whose mask equals one of the (last
    loaded) test masks are dropped, the rest are shuffled by numpy's global generator seeded with 42."""
    matches = np.all(train_masks[:, None, :] == last_test_masks[None, :, :], axis=2)
    idx = np.where(~np.any(matches, axis=1))[0]
    rng = np.random.RandomState(seed)       # np.random.seed(42); np.random.shuffle(...) draws from the same MT19937 stream
    rng.shuffle(idx)
    return idx

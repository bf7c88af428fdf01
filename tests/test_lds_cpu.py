"""LDS evidence: tests/golden/lds_main.json holds databases written by this build's
kept entry points (main.py, unlearn.py, calculate_global_scores.py on the CPU oracle backend) and what the reference's
`lds.py` printed when its __main__ was run UNCHANGED on them (tests/golden/make_lds_main_golden.py).  Here the same
numbers are recomputed from the stored rows with this build's restatement (gad/lds.py + src/attributions)."""
import json
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "lds_main.json")


@pytest.fixture(scope="module")
def gold():
    return json.load(open(GOLD))


def _select(rows, **cond):
    return [r for r in rows if all(r.get(k) == v for k, v in cond.items())]


def test_reference_lds_main_ran_on_this_builds_databases(gold):
    assert gold["fit_sizes"] == [20, 10, 20, 30, 40] and len(gold["lds_mean_ci"]) == 5
    assert all(np.isfinite(m) and np.isfinite(c) for m, c in gold["lds_mean_ci"])
    assert len(gold["train"]) == 40 and all(len(v) == 8 for v in gold["test"].values())
    r = gold["train"][0]
    assert r["method"] == "gd" and r["removal_dist"] == "shapley" and r["exp_name"] == "toy_train" and r["gd_steps"] == 3
    t = gold["test"]["42"][0]
    assert t["method"] == "retrain" and t["removal_dist"] == "datamodel" and t["datamodel_alpha"] == 0.5


def test_restated_lds_equals_what_the_reference_printed(gold):
    from gad.lds import masks_and_behaviours, reference_train_order, shapley_lds
    group_of = {i: i // 16 for i in range(320)}                       # cifar100 at GAD_SYNTH_SCALE=0.032
    tr = _select(gold["train"], dataset="cifar100", removal_dist="shapley", method="gd", exp_name="toy_train")
    train_masks, train_y, _ = masks_and_behaviours(tr, group_of, 20)
    tests = []
    for s in ("42", "43", "44"):
        rows = _select(gold["test"][s], dataset="cifar100", removal_dist="datamodel", method="retrain", exp_name="toy_test", datamodel_alpha=0.5)
        m, y, _ = masks_and_behaviours(rows, group_of, 20)
        tests.append((m, y))
    _, full_y, _ = masks_and_behaviours(_select(gold["full"], method="retrain"), group_of, 20)
    _, null_y, _ = masks_and_behaviours(_select(gold["null"], method="retrain"), group_of, 20)
    order = reference_train_order(train_masks, tests[-1][0][:8])
    for n, (want_mean, want_ci) in zip(gold["fit_sizes"], gold["lds_mean_ci"]):
        (mean, ci), attrs = shapley_lds(train_masks, train_y, tests, full_y, null_y, n_fit=n, train_order=order)
        assert mean == pytest.approx(want_mean, abs=0.006) and ci == pytest.approx(want_ci, abs=0.006), (n, mean, ci)
        # efficiency: the attributions sum to v(full) - v(null) (datashapley.py:33-44)
        assert float(np.sum(attrs[0])) == pytest.approx(float(full_y[0, 0] - null_y[0, 0]), rel=1e-6)


def test_restated_sd_lds_equals_what_shapley_lds_main_printed():
    """tests/golden/sd_lds_main.json = what the reference's text_to_image/shapley_lds.py __main__ printed on rows assembled by
    this build's compute_model_behaviors.assemble_row over this build's coalition samplers (258 artists); the rows are
    regenerated here (deterministic) and the LDS recomputed with src/attributions + gad/lds.py."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_lds_main_golden import SD_FIT, SD_N_GROUPS, sd_lds_rows
    from gad.lds import evaluate_lds
    from src.attributions.methods.datashapley import data_shapley
    gold = json.load(open(os.path.join(os.path.dirname(GOLD), "sd_lds_main.json")))
    rows = sd_lds_rows()

    def collect(rs):                                                 # shapley_lds.py:83-121, aggregate behaviour
        seeds = [int(r["exp_name"].split("seed_")[1]) for r in rs]
        rs = [r for _, r in sorted(zip(seeds, rs), key=lambda t: t[0])]
        m = np.zeros((len(rs), SD_N_GROUPS))
        for i, r in enumerate(rs):
            m[i, r["remaining_idx"]] = 1
        return m, np.array([[r["aesthetic_score_avg"]] for r in rs])
    tests = [collect(rows["test"][s]) for s in ("42", "43", "44")]
    v0, v1 = rows["null"][0]["aesthetic_score_avg"], rows["full"][0]["aesthetic_score_avg"]
    for which, key in (("fit", "lds_mean_ci"), ("baseline_fit", "baseline_lds_mean_ci")):
        x, y = collect(rows[which])
        for n, (want_mean, want_ci) in zip(gold["fit_sizes"], gold[key]):
            attrs = data_shapley(dataset_size=SD_N_GROUPS, x_train=x[:n], y_train=y[:n, 0], v0=v0, v1=v1)   # keywords, as shapley_lds.py:247-262
            mean, ci = evaluate_lds([np.ravel(attrs)], tests)
            assert mean == pytest.approx(want_mean, abs=0.006) and ci == pytest.approx(want_ci, abs=0.006), (which, n, mean, ci)
            if which == "fit" and n == SD_FIT:
                assert list(np.shape(attrs)) in ([258], [258, 1]) and gold["attrs_shape"] == [258, 1]
                np.testing.assert_allclose(np.ravel(attrs)[:8], gold["attrs_first8"], rtol=1e-9, atol=1e-12)
                assert np.argsort(-np.ravel(attrs), kind="stable")[:16].tolist() == gold["rank_first16"]

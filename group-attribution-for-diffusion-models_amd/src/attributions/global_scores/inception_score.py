"""Inception-score arithmetic (reference inception_score.py:56-76): exp(mean_i KL(p(y|x_i) || p(y)))
per split, averaged over splits.  `preds` are the softmax outputs [N, classes]."""
import numpy as np


def inception_score_from_probs(preds, splits=1):
    preds = np.asarray(preds, dtype=np.float64)
    n = preds.shape[0]
    out = []
    for k in range(splits):
        part = preds[k * (n // splits):(k + 1) * (n // splits)]
        py = part.mean(axis=0)
        # scipy.stats.entropy(pk, qk) normalises both arguments and sums pk*log(pk/qk) with 0 log 0 = 0
        pk = part / part.sum(axis=1, keepdims=True)
        qk = py / py.sum()
        with np.errstate(divide="ignore", invalid="ignore"):
            kl = np.where(pk > 0, pk * np.log(pk / qk), 0.0).sum(axis=1)
        out.append(np.exp(kl.mean()))
    return float(np.mean(out))

"""Bootstrap RidgeCV datamodel (reference src/attributions/methods/datamodel.py:8-36)."""
import numpy as np


def datamodel(x_train, y_train, num_runs):
    from sklearn.linear_model import RidgeCV

    n = len(x_train)
    coefs = []
    for _ in range(num_runs):
        pick = np.random.choice(n, n, replace=True)   # global numpy RNG, as the reference
        coefs.append(RidgeCV(cv=5, alphas=[0.1, 1.0, 1e1]).fit(x_train[pick], y_train[pick]).coef_)
    return np.stack(coefs)

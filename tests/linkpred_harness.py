"""TEST INFRASTRUCTURE: seeded restatement of the reference's link-prediction scorer
(performancescores/runlinkpredict.py:50-145): per vertex u the edges (u, n) with n > u are positives, twice as
many random non-neighbours are negatives (:57-99; the 'more than half the graph' special case :78-80 kept),
features are the Hadamard product of the two embeddings, the shuffled set is split 50/50 and scored with
LogisticRegression accuracy / F1.  The reference draws negatives and shuffles unseeded; here both are seeded so
two embeddings are scored on the IDENTICAL pair set and split."""
import warnings

import numpy as np


def pair_set(rowptr, colids, seed=0):
    n = len(rowptr) - 1
    rng = np.random.RandomState(seed)
    us, vs, ys = [], [], []
    for u in range(n):
        nu = np.unique(colids[rowptr[u]:rowptr[u + 1]])
        pos = nu[nu > u]
        total = 2 * len(pos)
        if len(nu) > n // 2:
            total = (n - len(nu)) // 2
        us += [u] * len(pos); vs += pos.tolist(); ys += [1] * len(pos)
        taken = set(nu.tolist())
        cn = 0
        while cn < total:
            nn = int(rng.randint(0, n))
            if nn not in taken:
                taken.add(nn)
                us.append(u); vs.append(nn); ys.append(0)
                cn += 1
    us, vs, ys = np.array(us), np.array(vs), np.array(ys)
    perm = rng.permutation(len(ys))
    return us[perm], vs[perm], ys[perm]


def link_scores(X, pairs, train_frac=0.5):
    """-> (accuracy, F1-macro, F1-micro) in percent with Hadamard features (runlinkpredict.py:128-139)."""
    from sklearn.exceptions import ConvergenceWarning
    from sklearn.linear_model import LogisticRegression
    from sklearn.metrics import accuracy_score, f1_score
    us, vs, ys = pairs
    X = np.asarray(X, dtype=np.float64)
    feats = X[us] * X[vs]
    cv = int(len(ys) * train_frac)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", ConvergenceWarning)
        model = LogisticRegression().fit(feats[:cv], ys[:cv])
    pred = model.predict(feats[cv:])
    labels = np.unique(pred)
    return (100.0 * accuracy_score(pred, ys[cv:]), 100.0 * f1_score(pred, ys[cv:], average="macro", labels=labels),
            100.0 * f1_score(pred, ys[cv:], average="micro", labels=labels))

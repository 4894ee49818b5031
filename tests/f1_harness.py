"""TEST INFRASTRUCTURE: seeded restatement of the reference's node-classification scorer
(performancescores/runnodeclassclust.py): label join with 1-based ids (:173-190), per train
fraction a shuffle, split at int(len*tf), OneVsRestClassifier(LogisticRegression(random_state=0)),
top-k label prediction (:160-171), micro/macro F1 (:289-309).  The reference script itself breaks
on current scikit-learn (MultiLabelBinarizer(range(labs)) positional, :254) and shuffles unseeded;
here classes=range(labs) and every split is seeded so two embeddings are scored on IDENTICAL splits."""
import warnings

import numpy as np


def load_labels(path, n):
    labels = [[] for _ in range(n)]
    for line in open(path):
        t = line.split()
        if len(t) >= 2:
            labels[int(t[0]) - 1].append(int(t[1]))
    return labels


def f1_scores(X, labels, train_fracs=(0.05, 0.10, 0.15, 0.20, 0.25), n_splits=10, seed=0):
    """-> {tf: (mean micro-F1, mean macro-F1)} in percent, over n_splits seeded shuffles per fraction."""
    from sklearn.exceptions import ConvergenceWarning
    from sklearn.linear_model import LogisticRegression
    from sklearn.metrics import f1_score
    from sklearn.multiclass import OneVsRestClassifier
    from sklearn.preprocessing import MultiLabelBinarizer

    keep = [i for i, l in enumerate(labels) if l]
    Xt = np.asarray(X, dtype=np.float64)[keep]
    Yt = [labels[i] for i in keep]
    labs = len({v for l in Yt for v in l})
    mlb = MultiLabelBinarizer(classes=list(range(labs)))
    out = {}
    for tf in train_fracs:
        mic, mac = [], []
        for s in range(n_splits):
            idx = np.random.RandomState(seed * 1000003 + s * 101 + int(tf * 100)).permutation(len(Yt))
            cv = int(len(Yt) * tf)
            tr, te = idx[:cv], idx[cv:]
            Ytr = mlb.fit_transform([Yt[i] for i in tr])
            with warnings.catch_warnings():
                warnings.simplefilter("ignore", ConvergenceWarning)
                warnings.simplefilter("ignore", UserWarning)
                model = OneVsRestClassifier(LogisticRegression(random_state=0)).fit(Xt[tr], Ytr)
                ps = np.asarray(model.predict_proba(Xt[te]))
            pred = [model.classes_[ps[r].argsort()[-len(Yt[i]):]].tolist() for r, i in enumerate(te)]
            Yte = mlb.fit_transform([Yt[i] for i in te])
            Yp = mlb.fit_transform(pred)
            mic.append(100.0 * f1_score(Yp, Yte, average="micro"))
            mac.append(100.0 * f1_score(Yp, Yte, average="macro"))
        out[tf] = (float(np.mean(mic)), float(np.mean(mac)))
    return out

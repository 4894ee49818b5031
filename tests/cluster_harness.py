"""TEST INFRASTRUCTURE: seeded restatement of the clustering half of the reference's scorer (performancescores/runnodeclassclust.py:311-331):
KMeans(n_clusters = c, random_state = 0) on the embedding for c = 2 ... 49, the Newman modularity of each clustering on the input graph
(the script calls python-louvain's `community_louvain.modularity`, which is not installed here: restated below for an unweighted simple graph),
the best modularity and its cluster count.  A subset of cluster counts keeps the test short; every fit is seeded."""
import warnings

import numpy as np


def modularity(rowptr, colids, labels):
    """Q = sum over communities of L_c / m - (d_c / 2m)^2 on the SIMPLE undirected graph of the CSR (L_c: edges inside c, d_c: total
    degree of c, m: edges): duplicate entries -- the reference's reader keeps them, cora has 302 -- are collapsed as networkx.Graph does,
    a self-loop counts once as an edge and twice in its vertex's degree.  Equal to networkx.algorithms.community.modularity (checked
    in tests/test_oracle_golden.py)."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    labels = np.asarray(labels, dtype=np.int64)
    n = len(rowptr) - 1
    src = np.repeat(np.arange(n), np.diff(rowptr))
    dst = np.asarray(colids, dtype=np.int64)
    lo, hi = np.minimum(src, dst), np.maximum(src, dst)
    edges = np.unique(lo * n + hi)               # every undirected edge once
    u, v = edges // n, edges % n
    m = float(len(edges))
    if m == 0:
        return 0.0
    k = int(labels.max()) + 1
    inside = np.bincount(labels[u][labels[u] == labels[v]], minlength=k).astype(np.float64)
    degree = np.bincount(labels[u], minlength=k).astype(np.float64) + np.bincount(labels[v], minlength=k).astype(np.float64)
    return float((inside / m - (degree / (2.0 * m)) ** 2).sum())


def modularity_table(X, rowptr, colids, cluster_counts=(2, 4, 7, 10, 16, 25, 40)):
    """-> {clusters: modularity} of KMeans(random_state = 0) clusterings of the embedding."""
    from sklearn.cluster import KMeans
    X = np.asarray(X, dtype=np.float64)
    out = {}
    for c in cluster_counts:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            km = KMeans(n_clusters=c, random_state=0, n_init=10).fit(X)
        out[int(c)] = modularity(rowptr, colids, km.labels_)
    return out

#!/usr/bin/env python3
"""Device time of single epochs: (1) right after initialisation, epoch by epoch; (2) with lr = 0 (the matrix does not
change) on the initial random matrix, on a trained one, and on a constant one -- is the epoch time data dependent?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import force2vec_amd as F

rowptr, colids = bench.load_graph(20, 16, 1)
n = len(rowptr) - 1
eng = F.Engine(rowptr, colids, 128)
eng.srand(1); eng.init_embeddings(0)
X0 = eng.get_embeddings()
print("after init:", " ".join("%.3f" % (eng.train(5, 1, 65536) * 1e3) for _ in range(30)), flush=True)
eng.train(5, 30, 65536)
X1 = eng.get_embeddings()


def frozen(name, X):
    eng.set_embeddings(X)
    eng.train(5, 3, 65536, 5, 0.0)
    t = [eng.train(5, 1, 65536, 5, 0.0) * 1e3 for _ in range(8)]
    print("lr=0 on %-34s %s" % (name + ":", " ".join("%.3f" % x for x in t)), flush=True)


frozen("the initial U[-1,1) matrix", X0)
frozen("the matrix after 60 epochs", X1)
frozen("a constant matrix (all 0.25)", np.full_like(X0, 0.25))
frozen("the initial matrix scaled by 0.01", X0 * np.float32(0.01))
frozen("the initial matrix scaled by 8", X0 * np.float32(8))
frozen("the trained matrix, rows permuted", X1[np.random.default_rng(0).permutation(n)])

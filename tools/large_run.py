#!/usr/bin/env python3
"""Functional + throughput check of one BIG configuration on one GPU (BASELINE configs[4] graph:
RMAT scale-24, ~16.8 M vertices, N*D = 2^31 -> 64-bit row offsets, 2 x 8 GiB of embeddings):
a few epochs of f2v_train, then ONE extra minibatch checked on sampled rows (the largest hubs
included) against the oracle's row function applied to the downloaded pre-step matrix."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import force2vec_amd as F
from force2vec_amd.graph import rmat_csr
from oracle import oracle as O

ap = argparse.ArgumentParser()
ap.add_argument("--scale", type=int, default=24)
ap.add_argument("--batch", type=int, default=262144)
ap.add_argument("--option", type=int, default=5)
ap.add_argument("--epochs", type=int, default=3)
ap.add_argument("--edge-factor", type=int, default=16)
args = ap.parse_args()

t0 = time.time()
rowptr, colids = rmat_csr(args.scale, args.edge_factor, seed=1)
n, nnz = len(rowptr) - 1, len(colids)
deg = np.diff(rowptr.astype(np.int64))
print("graph: n=%d nnz=%d max degree %d, generated in %.0fs" % (n, nnz, deg.max(), time.time() - t0), flush=True)
t0 = time.time()
eng = F.Engine(rowptr, colids, 128)
eng.srand(1)
eng.init_embeddings(0 if args.option in (5, 8, 11) else 1)
print("engine + init (N*D = %d rand() draws): %.0fs" % (n * 128, time.time() - t0), flush=True)
eng.train(args.option, 1, args.batch)
sec = eng.train(args.option, args.epochs, args.batch)
st = eng.stats()
print("train: %d epochs in %.3fs device time -> %.2f G edges/s, %.0f GB/s algorithmic, hub_chunk %d, %d hub rows/epoch"
      % (args.epochs, sec, st["nnz"] / sec / 1e9, st["algorithmic_bytes"] / sec / 1e9, eng.get_param("hub_chunk"), st["hub_rows"] // args.epochs), flush=True)

# one more minibatch in the middle of the vertex range, checked on sampled rows
before = eng.get_embeddings()
lo = (n // 2 // args.batch) * args.batch
hi = min(lo + args.batch, n)
rng = np.random.default_rng(3)
ids = rng.integers(0, n - 1, 5).astype(np.uint32)
walks = None
if args.option in (7, 10):
    walks = eng.generate_walks()
eng.minibatch_step(args.option, lo, hi, ids, 5, 0.02)
after = eng.get_embeddings()
rows = np.concatenate([rng.integers(lo, hi, 24), lo + np.argsort(deg[lo:hi])[-4:]])
chunk = eng.get_param("hub_chunk")
math = {5: 5, 8: 5, 11: 5, 6: 6, 9: 6, 7: 7, 10: 7}[args.option]
bad = 0
for i in rows:
    want = O.row(math, rowptr, colids, before, int(i), ids, 0.02, walks=walks, order=O.ORDER_TREE, chunk=chunk)
    if not np.array_equal(after[i], want):
        bad += 1
        print("MISMATCH row %d (degree %d): max abs %g" % (i, deg[i], np.abs(after[i] - want).max()))
outside = np.ones(n, bool)
outside[lo:hi] = False
same = bool(np.array_equal(after[outside], before[outside]))
print("sampled rows: %d checked (degrees up to %d), %d mismatches; rows outside the batch untouched: %s" % (len(rows), deg[rows].max(), bad, same))
eng.close()
sys.exit(0 if bad == 0 and same else 1)

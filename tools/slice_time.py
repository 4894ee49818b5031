#!/usr/bin/env python3
"""Compute-only time of ONE rank's share of an epoch at world sizes 1, 2, 4, 8 (no exchange): rank 0 steps its slice
of every minibatch, with the hub chunk chosen for the whole batch ("batch") or for the slice ("slice").  This is the
floor the multi-GPU epoch cannot go below.   usage: slice_time.py [batch] [epochs] [scale]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
import force2vec_amd as F
from force2vec_amd.dist import shard_bounds

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 5
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 20
rowptr, colids = bench.load_graph(scale, 16, 1)
n = len(rowptr) - 1
nb = -(-n // batch)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1)
eng.init_embeddings(0)
ids = np.random.default_rng(1).integers(0, n, size=nb * 5, dtype=np.uint32)
eng.upload_sample_ids(ids)
for world in (1, 2, 4, 8):
    for mode in ("batch", "slice"):
        eng.set_param("hub_chunk_for_batch", batch if mode == "batch" else -(-batch // world))
        for rep in range(2):
            eng.synchronize()
            t0 = time.perf_counter()
            for _ in range(epochs):
                for b in range(nb):
                    lo, hi = b * batch, min((b + 1) * batch, n)
                    _, a, z = shard_bounds(lo, hi, 0, world)
                    eng.minibatch_step_at(5, lo, hi, b * 5, 5, 0.02, 0, row_lo=a, row_hi=z)
            eng.synchronize()
            dt = (time.perf_counter() - t0) / epochs
        print("world %d  chunk for %-5s (%3d): rank 0's share of an epoch %.3f ms  (%.1f us per minibatch)" %
              (world, mode, eng.get_param("hub_chunk"), dt * 1e3, dt / nb * 1e6), flush=True)
eng.close()

# imbalance between the ranks' equal-row slices: every rank's own share at world 8 / 4 (max is what a minibatch waits for)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1)
eng.init_embeddings(0)
eng.upload_sample_ids(ids)
deg = np.diff(rowptr.astype(np.int64))
for world in (4, 8):
    eng.set_param("hub_chunk_for_batch", -(-batch // world))
    shares, worst = [], 0.0
    per_batch = np.zeros((nb, world))
    for r in range(world):
        for rep in range(2):
            eng.synchronize()
            t0 = time.perf_counter()
            for _ in range(epochs):
                for b in range(nb):
                    lo, hi = b * batch, min((b + 1) * batch, n)
                    _, a, z = shard_bounds(lo, hi, r, world)
                    eng.minibatch_step_at(5, lo, hi, b * 5, 5, 0.02, 0, row_lo=a, row_hi=z)
            eng.synchronize()
            dt = (time.perf_counter() - t0) / epochs
        shares.append(dt * 1e3)
        for b in range(nb):
            lo, hi = b * batch, min((b + 1) * batch, n)
            _, a, z = shard_bounds(lo, hi, r, world)
            per_batch[b, r] = deg[a:z].sum() + 3 * (z - a)
    print("world %d: per-rank share of an epoch (ms): %s; max/mean %.3f; work (nnz + 3 rows) of the heaviest slice / mean slice, "
          "averaged over minibatches: %.3f" % (world, " ".join("%.3f" % x for x in shares), max(shares) / (sum(shares) / world),
                                               float((per_batch.max(1) / per_batch.mean(1)).mean())), flush=True)
eng.close()

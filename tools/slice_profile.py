#!/usr/bin/env python3
"""rocprofv3 target: rank 0's share of RMAT-20 epochs at world 8 (slice chunk), no exchange -- kernel durations and
gaps of the per-minibatch launch chain.  usage: rocprofv3 --kernel-trace --stats ... -- python3 tools/slice_profile.py [world]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np

import bench
import force2vec_amd as F
from force2vec_amd.dist import shard_bounds

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
batch = 65536
rowptr, colids = bench.load_graph(20, 16, 1)
n = len(rowptr) - 1
nb = -(-n // batch)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1)
eng.init_embeddings(0)
eng.upload_sample_ids(np.random.default_rng(1).integers(0, n, size=nb * 5, dtype=np.uint32))
eng.set_param("hub_chunk_for_batch", -(-batch // world))
for _ in range(6):
    for b in range(nb):
        lo, hi = b * batch, min((b + 1) * batch, n)
        _, a, z = shard_bounds(lo, hi, 0, world)
        eng.minibatch_step_at(5, lo, hi, b * 5, 5, 0.02, 0, row_lo=a, row_hi=z)
eng.synchronize()
eng.close()

#!/usr/bin/env python3
"""Option 7 (rForce2Vec), parity mode: the epoch's walks come from the one serial rand() stream and are drawn on the host.
How long is an epoch, how long is the host generation alone, how long is the device work alone (fast_rng: same kernels, walks
generated on the device)?  With the producer thread an epoch should cost max(host, device), not their sum."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
rowptr, colids = bench.load_graph(scale, 16, 1)
n = len(rowptr) - 1
eng = F.Engine(rowptr, colids, 128)
eng.srand(1)
eng.init_embeddings(1)
t0 = time.perf_counter()
for _ in range(3):
    eng.generate_walks()
host = (time.perf_counter() - t0) / 3
eng.train(7, 2, batch)
t0 = time.perf_counter()
eng.train(7, 8, batch)
wall = (time.perf_counter() - t0) / 8
eng.set_param("fast_rng", 1)
eng.train(7, 2, batch)
dev = eng.train(7, 8, batch) / 8
print("option 7, RMAT-%d (n=%d), batch %d: epoch %.2f ms wall; host walk generation + upload alone %.2f ms; device work alone %.3f ms"
      % (scale, n, batch, wall * 1e3, host * 1e3, dev * 1e3))
eng.close()

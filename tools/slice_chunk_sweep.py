import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import bench, force2vec_amd as F
from force2vec_amd.dist import shard_bounds
batch = 65536
rowptr, colids = bench.load_graph(20, 16, 1)
n = len(rowptr) - 1
nb = -(-n // batch)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1); eng.init_embeddings(0)
eng.upload_sample_ids(np.random.default_rng(1).integers(0, n, size=nb * 5, dtype=np.uint32))
for world, chunks in ((8, (8, 16, 24, 32, 48, 64)), (4, (16, 32, 48, 64, 96)), (2, (32, 64, 96, 128))):
    out = []
    for ch in chunks:
        eng.set_param("hub_chunk", ch)
        best = 1e9
        for rep in range(4):
            eng.synchronize(); t0 = time.perf_counter()
            for _ in range(5):
                for b in range(nb):
                    lo, hi = b * batch, min((b + 1) * batch, n)
                    _, a, z = shard_bounds(lo, hi, 0, world)
                    eng.minibatch_step_at(5, lo, hi, b * 5, 5, 0.02, 0, row_lo=a, row_hi=z)
            eng.synchronize(); best = min(best, (time.perf_counter() - t0) / 5)
        out.append("%d: %.3f" % (ch, best * 1e3))
    print("world %d (auto %d): %s" % (world, 0, "; ".join(out)), flush=True)

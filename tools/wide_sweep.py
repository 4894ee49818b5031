#!/usr/bin/env python3
"""Chained minibatches: the wide form (pieces of a row meet in LDS, qwide_chain_kernel) against the round-2 form (partial sums
through HBM + tree nodes, qstep_chain_kernel) on RMAT-20, D=128, option 5; then the wide form's tunables.
Usage: wide_sweep.py [option] [batches, comma separated] [key=v1,v2,... sweeps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F

option = int(sys.argv[1]) if len(sys.argv) > 1 else 5
batches = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "256,384,1024,2048,4096").split(",")]
rowptr, colids = bench.load_graph(20, 16, 1)
nnz = len(colids)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1)
eng.init_embeddings(0 if option in (5, 8, 11) else 1)
eng.set_param("wide_max_batch", 1 << 20)  # the comparison decides where the threshold belongs


def t(batch):
    eng.train(option, 2, batch)
    return min(eng.train(option, 3, batch) / 3 for _ in range(3))


for b in batches:
    out = []
    for wide in (0, 1):
        eng.set_param("chain_wide", wide)
        s = t(b)
        out.append(s)
    print("batch %5d: round-2 chained %8.3f ms/epoch %6.2f G edges/s | wide %8.3f ms/epoch %6.2f G edges/s  (form %d, hub chunk %d)"
          % (b, out[0] * 1e3, nnz / out[0] * 1e-9, out[1] * 1e3, nnz / out[1] * 1e-9, eng.get_param("last_train_form"), eng.get_param("hub_chunk")), flush=True)
eng.set_param("chain_wide", 1)
for sweep in sys.argv[3:]:
    key, vals = sweep.split("=")
    default = eng.get_param(key)
    for v in vals.split(","):
        eng.set_param(key, int(v))
        print("  %s=%s:" % (key, v), "  ".join("batch %d %7.3f ms" % (b, t(b) * 1e3) for b in batches), flush=True)
    eng.set_param(key, default)
eng.close()

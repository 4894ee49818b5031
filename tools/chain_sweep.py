#!/usr/bin/env python3
"""Chained minibatches: epoch time over (batch, rows per launch, hub chunk) on the bench graph.
usage: chain_sweep.py [scale]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rowptr, colids = bench.load_graph(scale, 16, 1)
eng = F.Engine(rowptr, colids, 128)
eng.set_param("chain_max_batch", 1 << 20)
eng.srand(1)
eng.init_embeddings(0)
for batch in (256, 1024, 4096):
    for rows in (16384, 65536, 262144):
        if rows < 2 * batch:
            continue
        eng.set_param("chain_rows", rows)
        for chunk in (4, 8, 16, 32):
            eng.set_param("hub_chunk", chunk)
            eng.train(5, 2, batch)
            best = min(eng.train(5, 3, batch) / 3 for _ in range(2))
            print("batch %5d rows/launch %6d chunk %3d: %8.3f ms/epoch  %5.2f G edges/s" % (batch, rows, chunk, best * 1e3, len(colids) / best / 1e9), flush=True)
eng.close()

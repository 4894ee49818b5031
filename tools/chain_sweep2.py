#!/usr/bin/env python3
"""Chained minibatches at batch 256 / 384: rows in flight per item (4 | 8), hub chunk, fan-in."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F

rowptr, colids = bench.load_graph(20, 16, 1)
eng = F.Engine(rowptr, colids, 128)
eng.set_param("chain_rows", 16384)
eng.srand(1)
eng.init_embeddings(0)
for batch in (256, 384):
    for u in (4, 8):
        eng.set_param("rows_in_flight", u)
        for chunk, fanin in ((4, 32), (8, 32), (8, 64), (8, 16), (12, 32)):
            eng.set_param("hub_chunk", chunk)
            eng.set_param("hub_fanin", fanin)
            eng.train(5, 2, batch)
            best = min(eng.train(5, 3, batch) / 3 for _ in range(2))
            print("batch %5d rows in flight %d chunk %3d fanin %3d: %8.3f ms/epoch  %5.2f G edges/s" % (batch, u, chunk, fanin, best * 1e3, len(colids) / best / 1e9), flush=True)
eng.close()

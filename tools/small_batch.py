#!/usr/bin/env python3
"""Epoch time over small batch sizes with and without chained minibatches ("chain_batches") on the bench graph, for the two
row-in-flight depths of the chained kernel.   usage: small_batch.py [scale [option]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
option = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rowptr, colids = bench.load_graph(scale, 16, 1)
eng = F.Engine(rowptr, colids, 128)
eng.set_param("chain_max_batch", 1 << 20)
if len(sys.argv) > 3:
    eng.set_param("chain_rows", int(sys.argv[3]))
eng.srand(1)
eng.init_embeddings(0 if option == 5 else 1)
for batch in (256, 384, 1024, 2048, 4096, 8192, 16384):
    for chain, u in ((0, 0), (1, 0)):
        eng.set_param("chain_batches", chain)
        eng.set_param("rows_in_flight", u)
        eng.train(option, 3, batch)
        best = min(eng.train(option, 4, batch) / 4 for _ in range(3))
        st = eng.stats()
        print("batch %6d %s: %8.3f ms/epoch  %6.2f G edges/s  (%d launches/epoch, hub_chunk %d)"
              % (batch, "chained (%d rows per launch)" % eng.get_param("chain_rows") if chain else "one launch per minibatch ", best * 1e3, len(colids) / best / 1e9, st["step_launches"] // 4, eng.get_param("hub_chunk")), flush=True)
eng.close()

#!/usr/bin/env python3
"""Where the chained launch form stops paying: RMAT-20, option 5, D = 128, batches 4096 ... 32768 with "chain_max_batch" below / above the batch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F

rowptr, colids = bench.load_graph(20, 16, 1)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1)
eng.init_embeddings(0)
for batch in (4096, 6144, 8192, 12288, 16384, 32768):
    for rnd in range(2):
        for limit in (2048, 65536):
            eng.set_param("chain_max_batch", limit)
            eng.train(5, 3, batch)
            t = min(eng.train(5, 5, batch) / 5 for _ in range(3))
            print("batch %6d chain_max_batch %6d (form %d, hub chunk %d): %.3f ms/epoch  %.2f G edges/s" % (
                batch, limit, eng.get_param("last_train_form"), eng.get_param("hub_chunk"), t * 1e3, len(colids) / t / 1e9), flush=True)
eng.close()

#!/usr/bin/env python3
"""Device time per epoch over (batch, piece_affinity, hub chunk) on the bench graph: does placing hub pieces on the XCD
that owns their neighbours' id range pay at every batch size, and does it move the best chunk?
usage: affinity_sweep.py [scale [option]]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
option = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rowptr, colids = bench.load_graph(scale, 16, 1)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1)
eng.init_embeddings(0 if option == 5 else 1)
for batch in (4096, 16384, 65536, 262144):
    eng.set_param("hub_chunk_for_batch", batch)
    auto = eng.get_param("hub_chunk")
    for aff in (0, 1):
        eng.set_param("piece_affinity", aff)
        for chunk in sorted({max(8, auto // 4), max(8, auto // 2), auto, auto * 2}):
            eng.set_param("hub_chunk", chunk)
            eng.train(option, 12, batch)  # plans + settle
            best = min(eng.train(option, 10, batch) / 10 for _ in range(3))
            print("batch %6d affinity %d chunk %4d%s: %.4f ms/epoch  %.2f G edges/s" % (batch, aff, chunk, " (auto)" if chunk == auto else "", best * 1e3, len(colids) / best / 1e9), flush=True)
eng.close()

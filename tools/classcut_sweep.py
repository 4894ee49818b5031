#!/usr/bin/env python3
"""EXPERIMENT: pieces of split rows also end at the XCDs' id-range boundaries ("class_cut" = minimum degree): epoch time."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F

rowptr, colids = bench.load_graph(20, 16, 1)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1)
eng.init_embeddings(0)
for batch in (16384, 65536, 262144):
    eng.set_param("hub_chunk_for_batch", batch)
    auto = eng.get_param("hub_chunk")
    for chunk in (auto, max(8, auto // 2)):
        eng.set_param("hub_chunk", chunk)
        for cut in (0, 1, 2 * chunk + 1, 4 * chunk + 1, 8 * chunk + 1):
            eng.set_param("class_cut", cut)
            eng.train(5, 12, batch)
            best = min(eng.train(5, 10, batch) / 10 for _ in range(3))
            st = eng.stats()
            print("batch %6d chunk %4d class_cut >= %5d: %.4f ms/epoch  %.2f G edges/s  (%d pieces/epoch)"
                  % (batch, chunk, cut, best * 1e3, len(colids) / best / 1e9, st["hub_chunks"] // 10), flush=True)
eng.close()

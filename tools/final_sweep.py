#!/usr/bin/env python3
"""Round-2 re-check of r01's tunables under the XCD-affine, class-cut layout: rows in flight, waves per workgroup, hub chunk."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F

rowptr, colids = bench.load_graph(20, 16, 1)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1)
eng.init_embeddings(0)


def t(batch, reps=10):
    eng.train(5, 12, batch)
    return min(eng.train(5, reps, batch) / reps for _ in range(3))


for batch, chunks in ((65536, (64, 96, 128, 192, 256)), (262144, (256, 384, 512, 768, 1024)), (16384, (16, 24, 32, 48))):
    for ch in chunks:
        eng.set_param("hub_chunk", ch)
        b = t(batch)
        print("batch %6d chunk %4d: %.4f ms  %.2f G" % (batch, ch, b * 1e3, len(colids) / b / 1e9), flush=True)
eng.set_param("hub_chunk_for_batch", 65536)
for rif in (4, 8):
    eng.set_param("rows_in_flight", rif)
    for wpb in (4, 2, 1):
        eng.set_param("waves_per_block", wpb)
        b = t(65536)
        print("batch 65536 rows_in_flight %d waves_per_block %d: %.4f ms  %.2f G" % (rif, wpb, b * 1e3, len(colids) / b / 1e9), flush=True)
eng.set_param("rows_in_flight", 4)
eng.set_param("waves_per_block", 4)
for fanin in (8, 16, 32, 64):
    eng.set_param("hub_fanin", fanin)
    b = t(65536)
    print("batch 65536 fanin %d: %.4f ms  %.2f G" % (fanin, b * 1e3, len(colids) / b / 1e9), flush=True)
eng.close()

#!/usr/bin/env python3
"""Headline kernel (batch 65536, RMAT-20, option 5, D = 128): workgroups of 4 / 2 / 1 wavefronts.  A workgroup's wave slots are only
re-used when a whole new workgroup fits, so smaller workgroups keep more slots busy; they also stage the negative samples more often."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F

rowptr, colids = bench.load_graph(20, 16, 1)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1)
eng.init_embeddings(0)
batches = [int(b) for b in sys.argv[1:]] or [65536, 262144, 16384]


def t(batch, reps=10):
    return min(eng.train(5, reps, batch) / reps for _ in range(3))


for batch in batches:
    eng.train(5, 12, batch)
    for rnd in range(3):
        for wpb in (4, 2, 1):
            eng.set_param("waves_per_block", wpb)
            eng.train(5, 2, batch)
            b = t(batch)
            print("batch %6d round %d waves_per_block %d: %.4f ms  %.2f G" % (batch, rnd, wpb, b * 1e3, len(colids) / b / 1e9), flush=True)
eng.set_param("waves_per_block", 4)
eng.close()

#!/usr/bin/env python3
""""split rows combined in LDS for the plain launch form" -- measured with the machinery that exists.  With
"wide_single" = 1 the wide form (qwide_chain_kernel: a split row's pieces meet in LDS inside a finisher workgroup, helpers' group sums
through HBM, tree nodes only above fanin^2 pieces) also runs launches of ONE minibatch: nothing is handed on inside such a launch (no
flag is ever polled), so what is measured is the LDS-combine structure itself against the plain launch form (qstep_kernel: pieces
placed by the id range of their neighbours on the XCD that caches it, partial sums through HBM, tree nodes at the end of the grid).
Same pieces, same fan-in groups, same bits (checked).   usage: wide_single_probe.py [BATCH ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
import force2vec_amd as F

batches = [int(x) for x in sys.argv[1:]] or [16384, 65536, 262144]
rowptr, colids = bench.load_graph(20, 16, 1)
nnz = len(colids)
for batch in batches:
    res = {}
    for form in ("plain", "wide_single"):
        eng = F.Engine(rowptr, colids, 128)
        eng.set_param("hub_chunk_for_batch", batch)
        chunk = eng.get_param("hub_chunk")
        eng.set_param("hub_chunk", chunk)  # both forms cut rows alike: the bits must agree
        if form == "wide_single":
            eng.set_param("chain_max_batch", 1 << 20)
            eng.set_param("wide_max_batch", 1 << 20)
            eng.set_param("wide_rows", batch)
            eng.set_param("wide_single", 1)
        eng.srand(1)
        eng.init_embeddings(0)
        eng.train(5, 2, batch)
        t = min(eng.train(5, 4, batch) / 4 for _ in range(3))
        res[form] = (t, eng.get_param("last_train_form"), eng.get_embeddings(), eng.stats()["step_launches"] // 4)
        eng.close()
    same = bool(np.array_equal(res["plain"][2], res["wide_single"][2]))
    print("batch %7d (chunk %3d): plain launches %.3f ms/epoch (%.2f G edges/s, form %d, %d launches); LDS-combined split rows %.3f ms/epoch (%.2f G edges/s, form %d, %d launches): %+.1f %%; bits identical: %s"
          % (batch, chunk, res["plain"][0] * 1e3, nnz / res["plain"][0] * 1e-9, res["plain"][1], res["plain"][3], res["wide_single"][0] * 1e3, nnz / res["wide_single"][0] * 1e-9,
             res["wide_single"][1], res["wide_single"][3], (res["wide_single"][0] / res["plain"][0] - 1) * 100, same), flush=True)

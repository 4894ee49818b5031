#!/usr/bin/env python3
"""BASELINE configs[0] / [1] (the reference's own dataset): cora, batch 256 (and the CLI default 384), 1200 epochs -- device time of
f2v_train with chained minibatches and with one launch per minibatch, options 5 / 6 / 7."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import force2vec_amd as F

rowptr, colids = F.read_mtx(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cora.mtx"))
for dim in (16, 128):
    for option in (5, 6, 7):
        for batch in (256, 384):
            out = []
            for chain in (1, 0):
                eng = F.Engine(rowptr, colids, dim)
                eng.set_param("chain_batches", chain)
                eng.srand(1)
                eng.init_embeddings(0 if option == 5 else 1)
                eng.train(option, 50, batch)
                out.append(eng.train(option, 1200, batch))
                eng.close()
            print("cora D=%3d option %d batch %d, 1200 epochs: %.3f s chained (%s), %.3f s one launch per minibatch"
                  % (dim, option, batch, out[0], "D not a multiple of 32: not chained" if dim % 32 else "chained", out[1]), flush=True)

#!/usr/bin/env python3
"""What bounds a chained launch at small batches: the launch structure or the dependency chain?  Self-test build: the same
launches with their row waits switched off (results wrong, timing only) against the real thing, for several launch lengths,
next to one launch per minibatch."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F
from force2vec_amd import _lib

rowptr, colids = bench.load_graph(20, 16, 1)
T = _lib.selftest_lib()
eng = F.Engine(rowptr, colids, 128, selftest=True)
eng.set_param("chain_max_batch", 1 << 20)
eng.srand(1)
eng.init_embeddings(0)
chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 8
eng.set_param("hub_chunk", chunk)
for batch in (256, 4096):
    eng.set_param("chain_batches", 0)
    eng.train(5, 2, batch)
    print("batch %5d chunk %d one launch per minibatch: %8.3f ms/epoch" % (batch, chunk, min(eng.train(5, 3, batch) / 3 for _ in range(2)) * 1e3), flush=True)
    eng.set_param("chain_batches", 1)
    for k in (2, 4, 16, 64):
        eng.set_param("chain_rows", k * batch)
        out = []
        for nowait in (0, 1):
            _lib.check(T.f2v_test_chain_nowait(eng._h, nowait), T)
            eng.train(5, 2, batch)
            out.append(min(eng.train(5, 3, batch) / 3 for _ in range(2)) * 1e3)
        print("batch %5d chunk %d, %2d minibatches per launch: %8.3f ms/epoch with row waits, %8.3f without (timing only)" % (batch, chunk, k, out[0], out[1]), flush=True)
eng.close()

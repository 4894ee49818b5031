#!/usr/bin/env python3
"""Timing experiments on the wide form of chained minibatches (self-test build; every variant but the first gives WRONG
results): what a hop of the dependency chain is made of.  f2v_test_chain_nowait bits: 1 no row waits at all, 2 no
acknowledgement wait before a flag, 4 plain loads of handed-off rows, 8 the jobs' sums skipped."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F
from force2vec_amd import _lib
rowptr, colids = bench.load_graph(20, 16, 1)
T = _lib.selftest_lib()
eng = F.Engine(rowptr, colids, 128, selftest=True)
eng.srand(1); eng.init_embeddings(0); eng.set_param('wide_max_batch', 1 << 20)
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    eng.set_param(k, int(v))
for batch in (256, 1024):
    for mode, what in ((0, "as shipped"), (2, "no acknowledgement wait before the flags"), (4, "plain loads of handed-off rows"), (8, "job sums skipped"),
                       (14, "all three"), (1, "no row waits at all")):
        _lib.check(T.f2v_test_chain_nowait(eng._h, mode), T)
        eng.train(5, 2, batch)
        ms = min(eng.train(5, 3, batch) / 3 for _ in range(2)) * 1e3
        print("batch %5d: %8.3f ms/epoch  %s" % (batch, ms, what), flush=True)
    _lib.check(T.f2v_test_chain_nowait(eng._h, 0), T)
eng.close()

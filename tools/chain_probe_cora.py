#!/usr/bin/env python3
"""The timing-only variants of tools/chain_probe_wide.py on BASELINE configs[0] / [1] (cora, batch 256, 1200 epochs, option 5): what the 11
dependent minibatches of an epoch are made of on an otherwise idle chip.  Self-test build; every line but the first computes WRONG results."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import force2vec_amd as F
from force2vec_amd import _lib
rowptr, colids = F.read_mtx(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cora.mtx"))
T = _lib.selftest_lib()
for dim in (16, 128):
    eng = F.Engine(rowptr, colids, dim, selftest=True)
    eng.srand(1); eng.init_embeddings(0)
    for kv in sys.argv[1:]:
        k, v = kv.split("=")
        eng.set_param(k, int(v))
    for mode, what in ((0, "as shipped"), (2, "no acknowledgement wait before the flags"), (4, "plain loads of handed-off rows"), (8, "job sums skipped"),
                       (14, "all three"), (1, "no row waits at all")):
        _lib.check(T.f2v_test_chain_nowait(eng._h, mode), T)
        eng.train(5, 100, 256)
        s = min(eng.train(5, 1200, 256) for _ in range(2))
        print("cora D=%3d batch 256: %7.4f s / 1200 epochs = %6.2f us per epoch, %5.2f us per minibatch  %s" % (dim, s, s / 1200 * 1e6, s / 1200 / 11 * 1e6, what), flush=True)
    _lib.check(T.f2v_test_chain_nowait(eng._h, 0), T)
    eng.close()

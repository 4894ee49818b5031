#!/usr/bin/env python3
"""Do the eight XCDs finish a launch together?  Self-test build, f2v_test_xcd_times: single minibatches of RMAT-20 (option 5, D = 128, batch 65536 by default),
per XCD the time its last workgroup ended (after the launch's first workgroup started), the sum of its workgroups' durations and their number.
usage: xcd_balance_probe.py [batch = 65536] [key=value engine params ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F
from force2vec_amd import _lib

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rowptr, colids = bench.load_graph(20, 16, 1)
n = len(rowptr) - 1
T = _lib.selftest_lib()
eng = F.Engine(rowptr, colids, 128, selftest=True)
for kv in sys.argv[2:]:
    k, v = kv.split("=")
    eng.set_param(k, int(v))
eng.srand(1)
eng.init_embeddings(0)
eng.train(5, 4, batch)           # plans built, clocks up
ids = np.array([5, 77, 1234, 99999, 500000], dtype=np.uint32)
out = np.zeros(32, dtype=np.uint64)
print("RMAT-20, option 5, D = 128, batch %d, hub chunk %d; microseconds after the launch's first workgroup started" % (batch, eng.get_param("hub_chunk")))
for b in (0, 1, 5, 11, 15):
    lo, hi = b * batch, min((b + 1) * batch, n)
    if lo >= n:
        continue
    for rep in range(3):   # the third repetition is reported
        _lib.check(T.f2v_test_xcd_times(eng._h, 1, None), T)
        eng.minibatch_step(5, lo, hi, ids, 5, 0.02)
        eng.flush()
        _lib.check(T.f2v_test_xcd_times(eng._h, 1, out.ctypes.data_as(C.POINTER(C.c_uint64))), T)
    end, start, busy, cnt = out[0:8].astype(np.int64), out[8:16].astype(np.int64), out[16:24].astype(np.int64), out[24:32].astype(np.int64)
    t0 = start.min()
    print("minibatch %2d: last workgroup ends  %s" % (b, " ".join("%6.1f" % ((e - t0) / 100.0) for e in end)))
    print("              mean workgroup time  %s   workgroups %s" % (" ".join("%6.1f" % (bu / max(c, 1) / 100.0) for bu, c in zip(busy, cnt)), " ".join("%5d" % c for c in cnt)))
    print("              busy (sum / 160 slots) %s" % " ".join("%6.1f" % (bu / 160.0 / 100.0) for bu in busy), flush=True)
_lib.check(T.f2v_test_xcd_times(eng._h, 0, None), T)
eng.close()

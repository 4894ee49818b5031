#!/usr/bin/env python3
"""How much of a launch is its memory side?  The real launch plans of RMAT-20 at one batch size (option 5, D = 128) replayed by a kernel that only
gathers (self-test build, f2v_test_plan_gather: same items, same order, same lockstep, 4 rows in flight, nothing computed) against the step kernel's
own time for the same minibatches.   usage: plan_gather_probe.py [batch = 65536] [key=value engine params ...]"""
import ctypes as C
import os
import sys

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
eng.train(5, 6, batch)
nb = -(-n // batch)
t_step = min(eng.train(5, 10, batch) / 10 for _ in range(3)) / nb * 1e6
print("RMAT-20, option 5, D = 128, batch %d (hub chunk %d): the step kernel %.1f us per launch (self-test build; epoch %.3f ms)" % (batch, eng.get_param("hub_chunk"), t_step, t_step * nb * 1e-3), flush=True)
us = C.c_double()
tot = {}
MODES = ((0, "neighbour rows only"), (1, "+ own rows"), (3, "+ own rows + row stores"), (3 | 4, "the same, 2 groups per wave"), (3 | 8, "the same, 4 groups per wave"))
if os.environ.get("PLAN_GATHER_MODES"):
    MODES = tuple(m for m in MODES if str(m[0]) in os.environ["PLAN_GATHER_MODES"].split(","))
for mode, what in MODES:
    per = []
    for b in range(nb):
        lo, hi = b * batch, min((b + 1) * batch, n)
        _lib.check(T.f2v_test_plan_gather(eng._h, lo, hi, mode, 5, C.byref(us)), T)
        per.append(us.value)
    tot[mode] = sum(per) / nb
    print("gather only, %-30s: %.1f us per launch (min %.1f, max %.1f over the epoch's %d minibatches) = %.2f of the step kernel" % (
        what, tot[mode], min(per), max(per), nb, tot[mode] / t_step), flush=True)
eng.close()

#!/usr/bin/env python3
"""Host only: why option 7's walk generation (parity mode) does not parallelise.  Walk i consumes one rand() draw per step at a vertex
of degree > 2, so the stream offset of walk i + 1 is known only when walk i's path is: threads started at PREDICTED offsets are right only
as long as no walk before them drew fewer numbers than predicted.  This counts, on the benchmark graph, how often that happens and how
long the runs of correctly predicted walks are -- the most any in-order validation (one thread with 32 walks in flight, or T threads
with far-ahead segments) can commit per speculation.   usage: walk_deficits.py [SCALE]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import bench
from force2vec_amd import _lib

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rowptr, colids = bench.load_graph(scale, 16, 1)
n = len(rowptr) - 1
L = _lib.lib()
g = L.f2v_rng_create(1)
walks = np.empty(5 * n, dtype=np.uint32)
rp = np.ascontiguousarray(rowptr, dtype=np.uint32)
ci = np.ascontiguousarray(colids, dtype=np.uint32)
t0 = time.perf_counter()
_lib.check(L.f2v_rng_walks(g, rp.ctypes.data_as(_lib.u32p), ci.ctypes.data_as(_lib.u32p), n, len(ci), walks.ctypes.data_as(_lib.u32p)))
dt = time.perf_counter() - t0
L.f2v_rng_destroy(g)
deg = np.diff(rowptr.astype(np.int64))
W = walks.reshape(n, 5).astype(np.int64)
visited = np.concatenate([np.arange(n)[:, None], W[:, :4]], axis=1)      # the vertex each of the five steps starts from
draws = (deg[visited] > 2)                                                 # a step draws iff its vertex has more than two neighbours
used = draws.sum(axis=1)
# what walks_host predicts without knowing the path: a draw at every step, except the start vertex (known) and, where the first step draws
# nothing, the second vertex (known too)
pred = np.full(n, 5)
low0 = deg <= 2
pred[low0] = 3 + draws[low0, 1]
miss = used != pred
runs = np.diff(np.flatnonzero(np.concatenate([[True], miss])))
print("RMAT-%d: n=%d walks, %.1f M steps; f2v_rng_walks here: %.0f ms" % (scale, n, 5 * n / 1e6, dt * 1e3))
print("steps that start at a vertex of degree <= 2 (no draw): %.2f %% of all steps; walks that draw fewer numbers than predicted: %d = one in %.1f"
      % (100.0 * (1 - draws.mean()), int(miss.sum()), n / max(int(miss.sum()), 1)))
print("runs of consecutive correctly predicted walks: mean %.1f, median %d, 90th percentile %d, longest %d" % (runs.mean(), int(np.median(runs)), int(np.percentile(runs, 90)), int(runs.max())))
cum = np.cumsum(pred - used)
print("cumulative shortfall of the stream offset against the prediction after 1/64, 1/8, 1/2, all of the walks: %d, %d, %d, %d draws"
      % (cum[n // 64], cum[n // 8], cum[n // 2], cum[-1]))
print("=> a thread that starts the second half of an epoch's walks from the predicted offset is %d draws off: every one of its walks is wrong;" % cum[n // 2])
print("   any in-order scheme commits ~%.0f walks per speculation, i.e. ~%d dependent speculations per epoch, each at least one 5-step chain of cache misses long" % (runs.mean(), len(runs)))

#!/usr/bin/env python3
"""The push exchange inside ONE process: `world` engines on GPU 0, each driven by its own host thread, peers attached
by direct pointers (f2v_test_push_attach_local).  Separates what the protocol costs (push kernels, flag barriers,
launch gaps) from what several PROCESSES sharing a GPU cost; runs under rocprofv3.
usage: push_local.py [world] [batch] [epochs] [scale]"""
import ctypes as C
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# every engine's stream needs a hardware queue of its own: two streams multiplexed onto one queue would put one
# engine's step kernels BEHIND the other's spinning barrier kernel (the barrier could then only time out)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np

import bench
import force2vec_amd as F
from force2vec_amd import _lib

world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 10
scale = int(sys.argv[4]) if len(sys.argv) > 4 else 20
rowptr, colids = bench.load_graph(scale, 16, 1)
ref = F.Engine(rowptr, colids, 128)
ref.srand(1); ref.init_embeddings(0)
ref.train(5, 2, batch)
single = ref.train(5, epochs, batch) / epochs
want = ref.get_embeddings()
ref.close()
engs = [F.Engine(rowptr, colids, 128, selftest=True) for _ in range(world)]  # libf2v_selftest.so: the local attach hook
for e in engs:
    e.srand(1); e.init_embeddings(0); e.push_export(); e.set_param("push_timeout_ms", 3000)
arr = (C.c_void_p * world)(*[e._h for e in engs])
for r, e in enumerate(engs):
    _lib.check(_lib.selftest_lib().f2v_test_push_attach_local(e._h, r, world, arr), _lib.selftest_lib())
res = [None] * world


def run(r, k):
    res[r] = engs[r].train_sharded(5, k, batch)


for k in (2, epochs):
    th = [threading.Thread(target=run, args=(r, k)) for r in range(world)]
    t0 = time.perf_counter()
    [t.start() for t in th]; [t.join() for t in th]
    wall = (time.perf_counter() - t0) / k
nb = -(-(len(rowptr) - 1) // batch)
same = all(np.array_equal(e.get_embeddings(), want) for e in engs)
st = engs[0].push_stats()
print("world %d batch %d (one process): single %.3f ms/epoch; sharded %.3f ms/epoch device, %.3f wall (+%.1f us per minibatch); "
      "pushed/all-gather rows %.3f; bit-identical to the single engine: %s" %
      (world, batch, single * 1e3, max(res) / epochs * 1e3, wall * 1e3, (max(res) / epochs - single) / nb * 1e6, st["rows_pushed"] / max(st["rows_allgather"], 1), same), flush=True)
for e in engs:
    e.push_detach(); e.close()

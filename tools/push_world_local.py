#!/usr/bin/env python3
"""The push exchange at world W (default 8 = F2V_PUSH_MAX_RANKS) inside ONE process on ONE card: W engines (self-test build), each
driven by its own host thread on its own stream, peers attached by direct pointers (f2v_test_push_attach_local).  The GPU boxes of
this pool admit at most 6 processes on a card, so world 7 / 8 cannot be rehearsed with one process per rank there; this runs what
a world-8 run runs on the device -- kMaxRanks peer tables, reader-mask bit 7, the 8-lane xgmi_barrier_kernel, pushes into 7 peers,
the completion pass -- and compares every replica with the single-engine f2v_train bit for bit.  (hipIpc attach of 7 peers is the
one piece left out: tests/test_gpu_dist.py and tools/world_rehearsal.sh cover IPC up to the box's process limit.)
usage: push_world_local.py [world [graph.mtx option iters batch dim [landing [fused]]]]     exit code 0 = every replica identical"""
import ctypes as C
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# every engine's stream needs a hardware queue of its own: streams multiplexed onto one queue would put one engine's step
# kernels BEHIND another's spinning barrier kernel (the barrier could then only time out)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np

import force2vec_amd as F
from force2vec_amd import _lib

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cases = [(os.path.join(ROOT, "tests", "golden", "cora.mtx"), 5, 3, 256, 128, 0, 1),
         (os.path.join(ROOT, "tests", "golden", "pubmed.mtx"), 6, 2, 4096, 128, 0, 1),
         (os.path.join(ROOT, "tests", "golden", "karate.mtx"), 7, 3, 5, 64, 0, 1),      # 5 rows over 8 ranks: empty slices
         (os.path.join(ROOT, "tests", "golden", "cora.mtx"), 11, 2, 300, 100, 1, 0)]    # landing buffers, separate push kernel
if len(sys.argv) > 2:
    a = sys.argv[2:]
    cases = [(a[0], int(a[1]), int(a[2]), int(a[3]), int(a[4]), int(a[5]) if len(a) > 5 else 0, int(a[6]) if len(a) > 6 else 1)]
T = _lib.selftest_lib()
failed = 0
for graph, option, iters, batch, dim, landing, fused in cases:
    if not os.path.exists(graph) and os.path.exists(graph + ".gz"):  # (the larger goldens are committed gzipped)
        import gzip
        plain = os.path.join("/tmp", "f2v_golden_" + os.path.basename(graph))
        if not os.path.exists(plain):
            with gzip.open(graph + ".gz", "rb") as fi, open(plain + ".tmp%d" % os.getpid(), "wb") as fo:
                fo.write(fi.read())
            os.replace(plain + ".tmp%d" % os.getpid(), plain)
        graph = plain
    rowptr, colids = F.read_mtx(graph)
    kind = 0 if option in (5, 8, 11) else 1
    ref = F.Engine(rowptr, colids, dim)
    ref.srand(1); ref.init_embeddings(kind)
    ref.train(option, iters, batch)
    ref.train(option, 1, batch)
    chunk = ref.get_param("hub_chunk")
    want = ref.get_embeddings()
    ref.close()
    engs = [F.Engine(rowptr, colids, dim, selftest=True) for _ in range(world)]
    for e in engs:
        e.srand(1); e.init_embeddings(kind); e.set_param("push_timeout_ms", 20000)
        e.set_param("hub_chunk", chunk)  # the chunk is part of the summation order: pinned to the single engine's
        if landing:
            e.set_param("push_landing", 1)
        e.set_param("push_fused", fused)
        e.push_export()
    arr = (C.c_void_p * world)(*[e._h for e in engs])
    for r, e in enumerate(engs):
        _lib.check(T.f2v_test_push_attach_local(e._h, r, world, arr), T)
    errs = [None] * world

    def run(r, k):
        try:
            engs[r].train_sharded(option, k, batch)
        except Exception as ex:  # noqa: BLE001
            errs[r] = ex

    for k in (iters, 1):  # a second run on the same attachment continues where the first stopped
        th = [threading.Thread(target=run, args=(r, k)) for r in range(world)]
        [t.start() for t in th]
        [t.join() for t in th]
    same = [errs[r] is None and np.array_equal(engs[r].get_embeddings(), want) for r in range(world)]
    st = engs[world - 1].push_stats()
    print("world %d in one process, %s option %d batch %d D %d iters %d+1%s%s: replicas identical to the single engine: %s  (rank %d pushed %d of %d all-gather rows)%s"
          % (world, os.path.basename(graph), option, batch, dim, iters, ", landing buffers" if landing else "", "" if fused else ", separate push kernel",
             "all %d" % world if all(same) else "NO: ranks %s differ" % [r for r in range(world) if not same[r]], world - 1, st["rows_pushed"], st["rows_allgather"],
             "".join("\n  rank %d: %s" % (r, errs[r]) for r in range(world) if errs[r] is not None)), flush=True)
    failed += 0 if all(same) else 1
    for e in engs:
        try:
            e.push_detach()
        except Exception:  # noqa: BLE001
            pass
        e.close()
sys.exit(1 if failed else 0)

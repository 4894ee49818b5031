#!/usr/bin/env python3
"""Host-side cost of the launch plans: the first f2v_train call of a batch size builds and uploads the epoch's plans (wide form: on
the host's threads, wide_plans_for_epoch; F2V_IO_THREADS bounds them), later calls find them resident.  Wall time of the first and the
second call, device time, resident plan bytes.   usage: plan_build_time.py [SCALE [BATCH ...]]   (F2V_IO_THREADS=1: the serial build)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
batches = [int(x) for x in sys.argv[2:]] or [256, 384, 2048, 4096, 65536]
rowptr, colids = bench.load_graph(scale, 16, 1)
print("RMAT-%d: n=%d nnz=%d; F2V_IO_THREADS=%s, %d host cores" % (scale, len(rowptr) - 1, len(colids), os.environ.get("F2V_IO_THREADS", "(unset)"), os.cpu_count()), flush=True)
for batch in batches:
    eng = F.Engine(rowptr, colids, 128)
    eng.set_param("fast_rng", 1)  # (initial values from the device: this tool times plans, not the 2^31 rand() draws of RMAT-24's init)
    eng.srand(1); eng.init_embeddings(0)
    t0 = time.perf_counter(); dev = eng.train(5, 1, batch); w1 = time.perf_counter() - t0
    t0 = time.perf_counter(); dev2 = eng.train(5, 1, batch); w2 = time.perf_counter() - t0
    print("batch %6d: first call wall %.3f s (device %.4f) -> plans built + uploaded in ~%.3f s; second call wall %.4f s (device %.4f); form %d; resident plans %.1f MB"
          % (batch, w1, dev, w1 - w2, w2, dev2, eng.get_param("last_train_form"), eng.get_param("plan_resident_bytes") / 1e6), flush=True)
    eng.close()

import os, sys, time
sys.path.insert(0, '.')
import numpy as np
import bench, force2vec_amd as F
scale, batch, chunk, epochs = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
tag = sys.argv[5] if len(sys.argv) > 5 else "solo"
rowptr, colids = bench.load_graph(scale, 16, 1)
eng = F.Engine(rowptr, colids, 128)
eng.set_param("hub_chunk", chunk)
eng.srand(1); eng.init_embeddings(0)
t0 = time.time()
try:
    sched = [int(x) for x in os.environ.get("SCHED", "").split(",") if x] or [1] * epochs
    for i, k in enumerate(sched):
        s = eng.train(5, k, batch)
        print("[%s] call %d (%d epochs): %.3f ms per epoch" % (tag, i, k, s * 1e3 / max(k, 1)), flush=True)
    print("[%s] ok in %.1fs" % (tag, time.time() - t0), flush=True)
except Exception as e:
    print("[%s] FAILED after %.1fs: %s" % (tag, time.time() - t0, e), flush=True)

import sys, time, numpy as np
sys.path.insert(0, '.')
import bench, force2vec_amd as F
rowptr, colids = bench.load_graph(20, 16, 1)
n = len(rowptr) - 1
eng = F.Engine(rowptr, colids, 128)
eng.srand(1); eng.init_embeddings(0)
eng.train(5, 5, 65536)
ids = np.array([1, 2, 3, 4, 5], dtype=np.uint32)
lo = (n // 2 // 65536) * 65536
for rep in range(4):
    if rep % 2 == 0:
        x = eng.get_embeddings()
    else:
        time.sleep(0.05)
    eng.synchronize()
    t = time.perf_counter()
    eng.minibatch_step(5, lo, lo + 65536, ids, 5, 0.02)
    eng.synchronize()
    print("rep %d (%s before): minibatch_step + sync %.3f ms" % (rep, "get_embeddings" if rep % 2 == 0 else "50 ms sleep", (time.perf_counter() - t) * 1e3), flush=True)
    t = time.perf_counter()
    eng.minibatch_step(5, lo + 65536, lo + 2 * 65536, ids, 5, 0.02)
    eng.synchronize()
    print("   the next minibatch right after: %.3f ms" % ((time.perf_counter() - t) * 1e3), flush=True)

import sys, time
sys.path.insert(0, '.')
import bench, force2vec_amd as F
rowptr, colids = bench.load_graph(20, 16, 1)
nnz = len(colids)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1); eng.init_embeddings(0)
for batch in (65536, 262144):
    eng.train(5, 2, batch)
    for K in (0, 1, 5, 10, 20, 40):
        res = []
        for rep in range(3):
            eng.synchronize()
            t0 = time.perf_counter()
            dev = eng.train(5, K, batch)
            eng.synchronize()
            wall = time.perf_counter() - t0
            res.append((wall, dev))
        wall, dev = min(res)
        print("batch %d K=%d: wall %.3f ms, device %.3f ms, wall-device %.3f ms" % (batch, K, wall * 1e3, dev * 1e3, (wall - dev) * 1e3), flush=True)

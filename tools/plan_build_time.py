import sys, time
sys.path.insert(0, "/root/repo")
import bench, force2vec_amd as F
rowptr, colids = bench.load_graph(20, 16, 1)
for batch in (256, 384, 2048, 4096, 65536):
    eng = F.Engine(rowptr, colids, 128)
    eng.srand(1); eng.init_embeddings(0)
    t0 = time.perf_counter(); dev = eng.train(5, 1, batch); w1 = time.perf_counter() - t0
    t0 = time.perf_counter(); dev2 = eng.train(5, 1, batch); w2 = time.perf_counter() - t0
    print("batch %6d: first call wall %.3f s (device %.4f), second call wall %.4f s (device %.4f); form %d" % (batch, w1, dev, w2, dev2, eng.get_param("last_train_form")), flush=True)
    eng.close()

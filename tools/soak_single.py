import sys, time
sys.path.insert(0, '.')
import bench, force2vec_amd as F
rowptr, colids = bench.load_graph(20, 16, 1)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1); eng.init_embeddings(0)
for batch, epochs in ((65536, 3000), (4096, 600), (256, 150), (16384, 1000)):
    t0 = time.time()
    dev = eng.train(5, epochs, batch)
    print("batch %d: %d epochs, %d launches, device %.2fs (%.3f ms/epoch), no give-ups" % (batch, epochs, eng.stats()["step_launches"], dev, dev / epochs * 1e3), flush=True)

"""Soak: many epochs of every launch style on one GPU -- one launch per minibatch with in-grid combine trees (batch 65536, 16384),
chained minibatches (batch 4096 ... 256, options 5 and 6) -- without a single bounded wait giving up (f2v_train would fail)."""
import sys, time
sys.path.insert(0, '.')
import bench, force2vec_amd as F
rowptr, colids = bench.load_graph(20, 16, 1)
for option in (5, 6):
    eng = F.Engine(rowptr, colids, 128)
    eng.srand(1); eng.init_embeddings(0 if option == 5 else 1)
    for batch, epochs in ((65536, 1500), (16384, 600), (4096, 500), (1024, 300), (384, 250), (256, 250)):
        t0 = time.time()
        dev = eng.train(option, epochs, batch)
        print("option %d batch %d: %d epochs, %d launches, device %.2fs (%.3f ms/epoch), no give-ups" % (option, batch, epochs, eng.stats()["step_launches"], dev, dev / epochs * 1e3), flush=True)
    eng.close()

"""Soak: many epochs of every launch style on one GPU -- one launch per minibatch with in-grid combine trees (batch 65536, 16384),
chained minibatches (batch 4096 ... 256, options 5 and 6) -- without a single bounded wait giving up (f2v_train would fail)."""
import sys, time
sys.path.insert(0, '.')
import numpy as np
import bench, force2vec_amd as F
print("library: %s" % F._lib.lib().f2v_version().decode(), flush=True)
rowptr, colids = bench.load_graph(20, 16, 1)
for option in (5, 6):
    eng = F.Engine(rowptr, colids, 128)
    eng.srand(1); eng.init_embeddings(0 if option == 5 else 1)
    for batch, epochs in ((65536, 1500), (16384, 600), (4096, 500), (1024, 300), (384, 250), (256, 250)):
        t0 = time.time()
        dev = eng.train(option, epochs, batch)
        print("option %d batch %d: %d epochs, %d launches, device %.2fs (%.3f ms/epoch), no give-ups (recoveries %d)" % (option, batch, epochs, eng.stats()["step_launches"], dev, dev / epochs * 1e3, eng.get_param("recoveries")), flush=True)
    eng.close()
# small graph (the kernel's EARLY form: a launch is one dependency chain)
import os
rp, ci = F.read_mtx(os.path.join("tests", "golden", "cora.mtx"))
for option, dim in ((5, 16), (5, 128), (6, 128), (7, 64)):
    eng = F.Engine(rp, ci, dim)
    eng.srand(1); eng.init_embeddings(0 if option == 5 else 1)
    dev = eng.train(option, 20000, 256)
    print("cora option %d D=%d batch 256: 20000 epochs, device %.2fs, form %d, epochs per launch %d (qwide_chain_kernel MODE %s), recoveries %d"
          % (option, dim, dev, eng.get_param("last_train_form"), eng.get_param("last_wide_epochs"), "2" if eng.get_param("last_wide_epochs") > 1 else "0/1", eng.get_param("recoveries")), flush=True)
    eng.close()
# epochs chained in one launch (MODE 2) on a graph with helper and finisher workgroups: RMAT-13, 5000 epochs, and the end state against
# one epoch per launch ("wide_epochs" = 1) bit for bit
from force2vec_amd.graph import rmat_csr
rp, ci = rmat_csr(13, 8, seed=6)
for option, dim, batch in ((5, 128, 128), (6, 64, 256)):
    end = {}
    for E in (0, 1):
        eng = F.Engine(rp, ci, dim)
        eng.set_param("hub_chunk", 4)
        if E:
            eng.set_param("wide_epochs", E)
        eng.srand(1); eng.init_embeddings(0 if option == 5 else 1)
        dev = eng.train(option, 5000, batch)
        end[E] = eng.get_embeddings()
        print("RMAT-13 option %d D=%d batch %d: 5000 epochs, device %.2fs, form %d, epochs per launch %d, recoveries %d"
              % (option, dim, batch, dev, eng.get_param("last_train_form"), eng.get_param("last_wide_epochs"), eng.get_param("recoveries")), flush=True)
        eng.close()
    print("  end state of the chained-epochs run == one epoch per launch: %s" % bool(np.array_equal(end[0], end[1])), flush=True)

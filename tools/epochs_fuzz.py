#!/usr/bin/env python3
"""Randomised check of the epochs-chained-in-one-launch form against one launch per epoch: random RMAT graphs (scale 9 ... 14, edge
factor 2 ... 24), D, option, batch, hub chunk / fan-in, epochs per launch -- the embeddings must agree bit for bit.  usage: epochs_fuzz.py [cases [seed]]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import force2vec_amd as F
from force2vec_amd.graph import rmat_csr
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
chained = 0
for k in range(cases):
    scale, ef = int(rng.integers(9, 15)), int(rng.integers(2, 25))
    dim = int(rng.choice([16, 20, 32, 48, 64, 96, 100, 128, 256]))
    option = int(rng.choice([5, 6]))
    n = 1 << scale
    batch = int(rng.choice([32, 64, 96, 128, 256, 384, 512, 1000, 2048]))
    if (batch * dim) % 32:
        batch = 256
    chunk, fanin = int(rng.choice([2, 3, 4, 8])), int(rng.choice([4, 8, 16, 32, 32]))
    E = int(rng.choice([2, 3, 5, 8, 32]))
    iters = int(rng.integers(3, 40))
    rp, ci = rmat_csr(scale, ef, int(rng.integers(1, 1000)))
    res = []
    for e in (E, 1):
        eng = F.Engine(rp, ci, dim)
        eng.set_param("hub_chunk", chunk); eng.set_param("hub_fanin", fanin); eng.set_param("wide_epochs", e)
        eng.srand(7); eng.init_embeddings(0 if option == 5 else 1)
        eng.train(option, iters, batch)
        got = eng.get_param("last_wide_epochs")
        eng.train(option, 1, batch)
        res.append(eng.get_embeddings())
        if e == E:
            form, used = eng.get_param("last_train_form"), got
        assert eng.get_param("recoveries") == 0
        eng.close()
    ok = np.array_equal(res[0], res[1]) and np.isfinite(res[0]).all()
    chained += used > 1
    print("case %2d: rmat%d ef%d n=%d nnz=%d D=%d option %d batch %d chunk %d fanin %d, %d epochs, E=%d (form %d, a launch carried %d): %s"
          % (k, scale, ef, n, len(ci), dim, option, batch, chunk, fanin, iters, E, form, used, "identical" if ok else "DIFFERENT"), flush=True)
    if not ok:
        sys.exit(1)
print("%d cases, %d of them with epochs chained: all identical" % (cases, chained))

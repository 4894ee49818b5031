#!/usr/bin/env python3
"""Where chaining epochs in one launch ("wide_epochs") pays: option 5, D = 128, batch 256, graphs from cora to RMAT-16 (2 M nonzeros) --
seconds per 200 epochs with 1 / 8 / 32 epochs per launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import force2vec_amd as F
from force2vec_amd.graph import rmat_csr
graphs = [("cora", F.read_mtx(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "cora.mtx")))]
for scale, ef in ((12, 8), (13, 8), (14, 8), (15, 8), (16, 8), (16, 16)):
    graphs.append(("rmat%d ef%d" % (scale, ef), rmat_csr(scale, ef, 3)))
for name, (rp, ci) in graphs:
    out = []
    for e in (1, 8, 32):
        eng = F.Engine(rp, ci, 128)
        eng.set_param("wide_epochs", e)
        eng.srand(1); eng.init_embeddings(0)
        eng.train(5, 40, 256)
        s = min(eng.train(5, 200, 256) for _ in range(2))
        out.append("E=%d %.4f s (%d)" % (e, s, eng.get_param("last_wide_epochs")))
        eng.close()
    print("%-12s n=%6d nnz=%8d: %s" % (name, len(rp) - 1, len(ci), "  ".join(out)), flush=True)

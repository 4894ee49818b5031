import os, sys
sys.path.insert(0, os.getcwd())
import bench
import force2vec_amd as F
from force2vec_amd import _lib
rowptr, colids = bench.load_graph(20, 16, 1)
T = _lib.selftest_lib()
eng = F.Engine(rowptr, colids, 128, selftest=True)
eng.set_param("chain_max_batch", 1 << 20)
eng.srand(1); eng.init_embeddings(0)
for batch in (8192, 16384, 65536):
    for aff in (1, 0):
        eng.set_param("piece_affinity", aff)
        eng.set_param("chain_batches", 0)
        eng.train(5, 6, batch)
        plain = min(eng.train(5, 6, batch) / 6 for _ in range(3)) * 1e3
        eng.set_param("chain_batches", 1)
        eng.set_param("chain_rows", 4 * batch)
        out = []
        for nowait in (0, 1):
            _lib.check(T.f2v_test_chain_nowait(eng._h, nowait), T)
            eng.train(5, 6, batch)
            out.append(min(eng.train(5, 6, batch) / 6 for _ in range(3)) * 1e3)
        _lib.check(T.f2v_test_chain_nowait(eng._h, 0), T)
        print("batch %6d piece_affinity %d: plain %.3f ms; 4 minibatches per launch: %.3f with row waits, %.3f without (timing only)" % (batch, aff, plain, out[0], out[1]), flush=True)

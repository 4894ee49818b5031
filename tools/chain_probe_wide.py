import os, sys
sys.path.insert(0, "/root/repo")
import bench
import force2vec_amd as F
from force2vec_amd import _lib
rowptr, colids = bench.load_graph(20, 16, 1)
T = _lib.selftest_lib()
eng = F.Engine(rowptr, colids, 128, selftest=True)
eng.srand(1); eng.init_embeddings(0); eng.set_param('wide_max_batch', 1 << 20)
for batch in (256, 1024, 4096):
    for wide in (0, 1):
        eng.set_param("chain_wide", wide)
        out = []
        for nowait in (0, 1):
            _lib.check(T.f2v_test_chain_nowait(eng._h, nowait), T)
            eng.train(5, 2, batch)
            out.append(min(eng.train(5, 3, batch) / 3 for _ in range(2)) * 1e3)
        print("batch %5d wide %d: %8.3f ms/epoch with row waits, %8.3f without (timing only)" % (batch, wide, out[0], out[1]), flush=True)
    _lib.check(T.f2v_test_chain_nowait(eng._h, 0), T)
eng.close()

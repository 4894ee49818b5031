import os, sys
sys.path.insert(0, os.getcwd())
import bench
import force2vec_amd as F
from force2vec_amd import _lib
rowptr, colids = bench.load_graph(20, 16, 1)
T = _lib.selftest_lib()
eng = F.Engine(rowptr, colids, 128, selftest=True)
eng.set_param("chain_max_batch", 1 << 20)
eng.set_param("hub_chunk", 8)
eng.srand(1); eng.init_embeddings(0)
batch = 4096
for u in (4, 8):
    eng.set_param("rows_in_flight", u)
    eng.set_param("chain_batches", 0)
    eng.train(5, 2, batch)
    plain = min(eng.train(5, 3, batch) / 3 for _ in range(2)) * 1e3
    eng.set_param("chain_batches", 1)
    eng.set_param("chain_rows", 4 * batch)
    out = []
    for nowait in (0, 1):
        _lib.check(T.f2v_test_chain_nowait(eng._h, nowait), T)
        eng.train(5, 2, batch)
        out.append(min(eng.train(5, 3, batch) / 3 for _ in range(2)) * 1e3)
    print("%s rows_in_flight %d: plain %.3f ms, chained(4) %.3f with waits, %.3f without" % (os.environ.get("F2V_SELFTEST_LIBRARY", "default"), u, plain, out[0], out[1]), flush=True)

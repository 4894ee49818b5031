import os, sys
sys.path.insert(0, "/root/repo")
import bench
import force2vec_amd as F
rowptr, colids = bench.load_graph(20, 16, 1)
nnz = len(colids)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1); eng.init_embeddings(0)
eng.set_param("chain_max_batch", 1 << 20)
def t(batch):
    eng.train(5, 2, batch)
    return min(eng.train(5, 3, batch) / 3 for _ in range(3)) * 1e3
for batch in (4096, 8192, 16384):
    eng.set_param("hub_chunk_for_batch", batch)
    auto = eng.get_param("hub_chunk")
    eng.set_param("chain_batches", 0)
    plain = t(batch)
    eng.set_param("chain_batches", 1); eng.set_param("chain_wide", 0)
    old = t(batch)
    out = []
    eng.set_param("chain_wide", 1); eng.set_param("wide_max_batch", 1 << 20)
    for ch in (4, 8, 16, 32):
        eng.set_param("hub_chunk", ch)
        for rows in (262144,):
            eng.set_param("wide_rows", rows)
            out.append("chunk %d: %.3f" % (ch, t(batch)))
    print("batch %5d (auto chunk %d): plain %.3f ms, round-2 chained %.3f ms, wide: %s" % (batch, auto, plain, old, "  ".join(out)), flush=True)
    eng.set_param("hub_chunk_for_batch", batch)

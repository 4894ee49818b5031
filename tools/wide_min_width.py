import sys
sys.path.insert(0, "/root/repo")
import bench, force2vec_amd as F
rowptr, colids = bench.load_graph(20, 16, 1)
for dim in (16, 32, 64):
    eng = F.Engine(rowptr, colids, dim)
    eng.srand(1); eng.init_embeddings(0)
    for mw in (16, 32, 64):
        eng.set_param("wide_min_width", mw)
        out = []
        for b in (256, 1024, 2048):
            eng.train(5, 2, b)
            out.append(min(eng.train(5, 3, b) / 3 for _ in range(2)) * 1e3)
        print("RMAT-20 D=%d wide_min_width=%d: batch 256 / 1024 / 2048: %.3f %.3f %.3f ms/epoch" % (dim, mw, *out), flush=True)
    eng.close()

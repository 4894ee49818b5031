import sys
sys.path.insert(0, '.')
import bench, force2vec_amd as F
rowptr, colids = bench.load_graph(20, 16, 1)
nnz = len(colids)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1); eng.init_embeddings(0)
eng.train(5, 20, 65536)
for batch, chunks in ((65536, (80, 96, 112, 128, 144, 160, 192)), (262144, (384, 448, 512, 576, 640, 768))):
    out = []
    for ch in chunks:
        eng.set_param("hub_chunk", ch)
        eng.train(5, 4, batch)
        best = min(eng.train(5, 8, batch) / 8 for _ in range(3))
        out.append("%d: %.3f" % (ch, best * 1e3))
    print("batch %d: %s" % (batch, "; ".join(out)), flush=True)
for fanin in (8, 16, 32, 64, 128):
    eng.set_param("hub_chunk", 128)
    eng.set_param("hub_fanin", fanin)
    eng.train(5, 4, 65536)
    best = min(eng.train(5, 8, 65536) / 8 for _ in range(3))
    print("batch 65536 chunk 128 fanin %d: %.3f ms" % (fanin, best * 1e3), flush=True)

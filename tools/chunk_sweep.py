"""Device time per epoch (HIP events, best of 3) over hub chunk sizes around the automatic choice; SWEEP=large for the ends of the range."""
import sys, time
sys.path.insert(0, '.')
import bench, force2vec_amd as F
rowptr, colids = bench.load_graph(20, 16, 1)
nnz = len(colids)
eng = F.Engine(rowptr, colids, 128)
eng.srand(1); eng.init_embeddings(0)
import os
SWEEP = ((262144, (128, 256, 512)), (65536, (32, 64, 128, 256)), (16384, (8, 16, 32, 64)), (4096, (8, 16, 32)), (8192, (8, 16, 32)))
if os.environ.get('SWEEP') == 'large':
    SWEEP = ((262144, (512, 1024, 2048)), (1048576, (512, 1024, 2048, 4096)), (4096, (4, 8)), (1024, (4, 8, 16)), (256, (4, 8, 16)))
for batch, chunks in SWEEP:
    out = []
    for ch in chunks:
        eng.set_param("hub_chunk", ch)
        eng.train(5, 2, batch)
        best = min(eng.train(5, 6, batch) / 6 for _ in range(3))
        out.append("chunk %d: %.3f ms (%.2f G)" % (ch, best * 1e3, nnz / best / 1e9))
    eng.set_param("hub_chunk_for_batch", batch)
    print("batch %d (auto %d): %s" % (batch, eng.get_param("hub_chunk"), "; ".join(out)), flush=True)

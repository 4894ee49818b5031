import sys, time
sys.path.insert(0, '.')
import numpy as np
import force2vec_amd as F
from oracle import oracle as O
for rows in (1 << 20, 3 << 20, 16 << 20):
    rp = np.zeros(rows + 1, dtype=np.uint32); ci = np.zeros(0, dtype=np.uint32)
    eng = F.Engine(rp, ci, 128)
    eng.srand(1)
    t0 = time.time(); eng.init_embeddings(0); dt = time.time() - t0
    print("rows %d (%.1f GiB): init %.2fs" % (rows, rows * 512 / 2**30, dt), flush=True)
    if rows <= 3 << 20:
        X = eng.get_embeddings()
        want = O.Rng(1).init_embeddings(rows, 128, 0)
        print("  bit-identical to the serial stream:", bool(np.array_equal(X, want)), flush=True)
        # the stream continues correctly after the init
        nxt = eng.draw_samples(1000, 3, 3)
        g = O.Rng(1); g.init_embeddings(rows, 128, 0)
        print("  next draws agree:", [int(v) for v in nxt])
    eng.close()

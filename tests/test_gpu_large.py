"""-m gpu: BASELINE configs[3] and configs[4] at FULL size on one MI355X, and the push exchange at the matrix size
that forces its landing-buffer path.

  configs[4]  synthetic RMAT scale-24 (16.8 M vertices, ~521 M directed nonzeros, N*D = 2^31: 64-bit row offsets,
              2 x 8 GiB of embeddings), option 11 (load-balanced t-distribution), D = 128
  configs[3]  com-Orkut-sized power-law graph (3 072 441 vertices, ~234 M nonzeros; the real file is not in the
              container: force2vec_amd.graph.orkut_like_csr), option 6 (sigmoid), D = 128

The oracle cannot run an epoch at these sizes in seconds, so parity is checked through size-independent properties:
two engines give identical bits (determinism under load), and ONE more minibatch is compared, on sampled rows that
include the batch's largest hubs and zero-degree rows, with the oracle's row function applied to the downloaded
pre-step matrix (bit for bit, ORDER_TREE with the engine's hub chunk); rows outside the minibatch must not change."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def cached_graph(name, build):
    """CSR under /tmp (generation is setup; a second test or a child process reads the cache)."""
    path = "/tmp/f2v_test_%s.npz" % name
    if os.path.exists(path):
        try:
            z = np.load(path)
            return z["rowptr"], z["colids"]
        except Exception:
            pass
    rowptr, colids = build()
    tmp = path + ".%d.tmp.npz" % os.getpid()
    np.savez(tmp, rowptr=rowptr, colids=colids)
    os.replace(tmp, path)
    return rowptr, colids


def full_size_check(F, rowptr, colids, option, batch, n_sample=32):
    n, dim = len(rowptr) - 1, 128
    math = {5: 5, 8: 5, 11: 5, 6: 6, 9: 6}[option]
    deg = np.diff(rowptr.astype(np.int64))
    engs = []
    for _ in range(2):
        e = F.Engine(rowptr, colids, dim)
        e.srand(1)
        e.init_embeddings(0 if math == 5 else 1)
        e.train(option, 1, batch)
        engs.append(e)
    before = engs[0].get_embeddings()
    other = engs[1].get_embeddings()
    engs[1].close()
    assert np.array_equal(before, other)
    del other
    assert np.isfinite(before[:: max(1, n // 4096)]).all()
    eng = engs[0]
    st = eng.stats()
    assert st["nnz"] == len(colids) and st["rows"] == n and st["hub_rows"] > 0
    chunk = eng.get_param("hub_chunk")
    rng = np.random.default_rng(9)
    lo = (n // 2 // batch) * batch
    hi = min(lo + batch, n)
    ids = rng.integers(0, n - 1, 5).astype(np.uint32)
    ids[0] = lo + int(np.argmax(deg[lo:hi]))  # the batch's largest hub samples itself: NaN -> -5 rule (option 5 maths)
    eng.minibatch_step(option, lo, hi, ids, 5, 0.02)
    after = eng.get_embeddings()
    zero_rows = lo + np.flatnonzero(deg[lo:hi] == 0)[:6]
    rows = np.concatenate([rng.integers(lo, hi, n_sample), lo + np.argsort(deg[lo:hi])[-6:], zero_rows])
    assert deg[rows].max() > 8 * chunk and len(zero_rows) == 6 and len(rows) >= 32
    for i in rows:
        want = O.row(math, rowptr, colids, before, int(i), ids, 0.02, order=O.ORDER_TREE, chunk=chunk)
        assert np.array_equal(after[i], want), (int(i), int(deg[i]), float(np.abs(after[i] - want).max()))
    assert np.array_equal(after[:lo], before[:lo]) and np.array_equal(after[hi:], before[hi:])
    eng.close()
    return len(rows)


def test_config5_rmat24_option11_full_size():
    """BASELINE configs[4]: RMAT scale-24, option 11, D = 128, batch 262144 (64 minibatches per epoch)."""
    import force2vec_amd as F
    from force2vec_amd.graph import rmat_csr
    rowptr, colids = cached_graph("rmat24", lambda: rmat_csr(24, 16, seed=1))
    assert len(rowptr) - 1 == 1 << 24 and len(colids) > 500_000_000
    assert full_size_check(F, rowptr, colids, 11, 262144) >= 32


def test_config4_orkut_sized_option6_full_size():
    """BASELINE configs[3]: com-Orkut's size (3 072 441 vertices, 117.1 M undirected edges), option 6, D = 128."""
    import force2vec_amd as F
    from force2vec_amd.graph import ORKUT_N, orkut_like_csr
    rowptr, colids = cached_graph("orkut_like", orkut_like_csr)
    assert len(rowptr) - 1 == ORKUT_N and abs(len(colids) / 2 - 117_185_083) < 0.01 * 117_185_083
    assert full_size_check(F, rowptr, colids, 6, 131072) >= 32


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _landing_worker(rank, world, port, outdir):
    import torch.distributed as dist
    import force2vec_amd as F
    from force2vec_amd import dist as fdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    z = np.load("/tmp/f2v_test_rmat22e8.npz")
    eng = F.Engine(z["rowptr"], z["colids"], 128, device=0)
    eng.set_param("hub_chunk", 64)
    eng.srand(1)
    eng.init_embeddings(0)
    eng.set_param("push_timeout_ms", 20000)
    comm = fdist.PushExchange(dist, rank, world)
    fdist.ShardedTrainer(eng, rank, world, comm).train(5, 1, 262144, 5, 0.02, 0)
    assert eng.get_param("push_landing") == 1  # chosen by the engine itself: the matrices are over 2 GiB
    X = eng.get_embeddings()
    # a digest per rank instead of 2 GiB files: row sums in float64 and a strided sample of raw rows
    np.savez(os.path.join(outdir, "r%d.npz" % rank), sums=X.sum(axis=1, dtype=np.float64), sample=X[::997].copy())
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def test_push_exchange_lands_at_2gib(tmp_path):
    """Matrices of 2 GiB and more cannot be mapped through HIP IPC: two ranks (sharing the card) on RMAT scale-22
    (4.19 M vertices x 128 floats = 2.15 GB per matrix) exchange through the landing buffer WITHOUT being told to, and
    both replicas equal the single-GPU run."""
    import torch.multiprocessing as mp
    import force2vec_amd as F
    from force2vec_amd.graph import rmat_csr
    rowptr, colids = cached_graph("rmat22e8", lambda: rmat_csr(22, 8, seed=1))
    assert (len(rowptr) - 1 + 4096) * 128 * 4 >= 1 << 31
    mp.spawn(_landing_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    eng = F.Engine(rowptr, colids, 128)
    eng.set_param("hub_chunk", 64)
    eng.srand(1)
    eng.init_embeddings(0)
    eng.train(5, 1, 262144)
    X = eng.get_embeddings()
    eng.close()
    sums, sample = X.sum(axis=1, dtype=np.float64), X[::997]
    for r in range(2):
        z = np.load(str(tmp_path / ("r%d.npz" % r)))
        assert np.array_equal(z["sums"], sums) and np.array_equal(z["sample"], sample)


def _boundary_worker(rank, world, port, n, expect_landing, outdir):
    import torch.distributed as dist
    import force2vec_amd as F
    from force2vec_amd import dist as fdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    z = np.load("/tmp/f2v_test_ring%d.npz" % n)
    eng = F.Engine(z["rowptr"], z["colids"], 128, device=0)
    eng.set_param("fast_rng", 1)  # 2 GiB of rand() draws are not the point here
    eng.srand(1)
    eng.init_embeddings(0)
    eng.set_param("push_timeout_ms", 20000)
    comm = fdist.PushExchange(dist, rank, world)
    comm.attach(eng)
    assert eng.get_param("push_landing") == expect_landing
    fdist.ShardedTrainer(eng, rank, world, comm).train(5, 1, 1 << 20, 5, 0.02, 0)
    X = eng.get_embeddings()
    np.savez(os.path.join(outdir, "r%d.npz" % rank), sums=X.sum(axis=1, dtype=np.float64), sample=X[::1009].copy())
    comm.detach(eng)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("where", ["just_below_2GiB", "just_above_2GiB"])
def test_ipc_mapping_boundary(where, tmp_path):
    """hipIpcOpenMemHandle cannot open allocations of 2^31 bytes and more (f2v_push_export: kIpcMaxBytes).  A matrix ONE ROW
    below that size is attached directly (the peers map both matrices), one row above it goes through the landing buffer --
    chosen by the engine from the size alone -- and both give the single-GPU result.  A mapping call that hangs is cut off
    by the join time-out instead of stalling the suite."""
    import torch.multiprocessing as mp
    import force2vec_amd as F
    n = (1 << 22) - 4096 + (-1 if where == "just_below_2GiB" else 1)  # (n + 4096 padding rows) * 512 bytes = 2^31 -/+ 512
    assert ((n + 4096) * 512 < 1 << 31) == (where == "just_below_2GiB")

    def ring():  # every vertex linked to its two neighbours on a ring and to one far vertex: cheap to build, rows all alike
        v = np.arange(n, dtype=np.int64)
        far = (v * 7919 + 13) % n
        r = np.concatenate([v, v, v, far])
        c = np.concatenate([(v + 1) % n, (v - 1) % n, far, v])
        keep = r != c
        r, c = r[keep], c[keep]
        order = np.lexsort((c, r))
        rowptr = np.zeros(n + 1, dtype=np.int64)
        rowptr[1:] = np.cumsum(np.bincount(r, minlength=n))
        return rowptr.astype(np.uint32), c[order].astype(np.uint32)
    rowptr, colids = cached_graph("ring%d" % n, ring)
    ctx = mp.spawn(_boundary_worker, args=(2, _free_port(), n, 0 if where == "just_below_2GiB" else 1, str(tmp_path)), nprocs=2, join=False)
    import time
    t0 = time.time()
    while not ctx.join(timeout=5):
        if time.time() - t0 > 240:
            for p in ctx.processes:
                p.kill()
            pytest.fail("attaching a matrix %s did not finish in 240 s (a hipIpcOpenMemHandle that never returns?)" % where)
    eng = F.Engine(rowptr, colids, 128)
    eng.set_param("fast_rng", 1)
    eng.srand(1)
    eng.init_embeddings(0)
    eng.set_param("hub_chunk_for_batch", (1 << 20) // 2)  # the chunk a rank of two picks for its slice (no row is split here anyway)
    eng.train(5, 1, 1 << 20)
    X = eng.get_embeddings()
    eng.close()
    for r in range(2):
        z = np.load(str(tmp_path / ("r%d.npz" % r)))
        assert np.array_equal(z["sums"], X.sum(axis=1, dtype=np.float64)) and np.array_equal(z["sample"], X[::1009])


def test_ingest_at_config4_size(tmp_path):
    """SURVEY 8f row 1 at BASELINE configs[3]'s size: the com-Orkut-sized graph written as a `pattern symmetric` MatrixMarket
    text (117 M entries, ~1.9 GB: the reader slurps the file whole and cuts it into per-thread byte ranges -- offsets and
    counters past 2^30 are what this is about) and parsed back by f2v_read_mtx must give the identical CSR; the
    binary CSR cache round-trips it.  (Marked gpu because of its size, not because it touches the card.)"""
    import time
    import force2vec_amd as F
    from force2vec_amd.graph import edges_from_csr, orkut_like_csr
    import pyarrow as pa
    import pyarrow.csv as pacsv
    rowptr, colids = cached_graph("orkut_like", orkut_like_csr)
    n = len(rowptr) - 1
    src, dst = edges_from_csr(rowptr, colids)
    mtx = "/tmp/f2v_test_orkut_like.mtx"
    with open(mtx, "wb") as f:
        f.write(b"%%MatrixMarket matrix coordinate pattern symmetric\n%d %d %d\n" % (n, n, len(src)))
        pacsv.write_csv(pa.table({"r": pa.array(src + 1), "c": pa.array(dst + 1)}), f, pacsv.WriteOptions(include_header=False, delimiter=" "))
    del src, dst
    size = os.path.getsize(mtx)
    assert size > 1_500_000_000
    t0 = time.time()
    rp, ci = F.read_mtx(mtx)
    took = time.time() - t0
    print("f2v_read_mtx: %.2f GB, %d entries in %.1f s" % (size / 1e9, len(colids) // 2, took))
    assert np.array_equal(rp, rowptr) and np.array_equal(ci, colids)
    F.write_csr_bin(mtx + ".f2vcsr", rp, ci)
    rp2, ci2 = F.read_csr_bin(mtx + ".f2vcsr")
    assert np.array_equal(rp2, rowptr) and np.array_equal(ci2, colids)
    os.remove(mtx)
    os.remove(mtx + ".f2vcsr")

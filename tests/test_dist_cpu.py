"""The N>1 path on CPU: force2vec_amd.dist.ShardedTrainer over torch.distributed (gloo,
world_size 2 and 3) with the oracle standing in for the device (tests/fake_engine.py).
Checks that sharding each minibatch's rows + one all-gather per minibatch reproduces the
single-process run bit for bit, for every option and for ragged / empty shards."""
import os
import socket

import numpy as np
import pytest

from conftest import golden_graph_path
from oracle import oracle as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, outdir, exchange="allgather"):
    import torch.distributed as dist
    from fake_engine import OracleEngine
    from force2vec_amd import dist as fdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    graph, option, iters, batch, dim, bs = case
    rp, ci = O.read_mtx(golden_graph_path(graph))
    eng = OracleEngine(rp, ci, dim, chunk=4)
    eng.srand(1)
    eng.init_embeddings(0 if option == 5 else 1)
    comm = fdist.HostStageComm(dist, rank, world) if exchange == "allgather" else fdist.NeedExchange(dist, rank, world, backend="host")
    fdist.ShardedTrainer(eng, rank, world, comm).train(option, iters, batch, 5, 0.02, bs)
    np.save(os.path.join(outdir, "r%d.npy" % rank), eng.get_embeddings())
    if exchange == "need":
        np.save(os.path.join(outdir, "sent%d.npy" % rank), np.array([comm.rows_sent_per_epoch]))
    dist.barrier()
    dist.destroy_process_group()


CASES = [("karate.mtx", 5, 3, 16, 16, 0), ("karate.mtx", 5, 2, 7, 16, 1), ("karate.mtx", 6, 3, 16, 32, 0),
         ("karate.mtx", 7, 3, 5, 16, 0), ("karate.mtx", 5, 2, 64, 16, 0), ("cora.mtx", 5, 2, 256, 16, 0)]


def _ids(cases):
    return ["%s-opt%d-B%d-bs%d" % (c[0][:-4], c[1], c[3], c[5]) for c in cases]


# world 3 (ragged and empty shards) on the small cases only: every spawn costs seconds
PAIRS = [(c, 2) for c in CASES] + [(c, 3) for c in CASES[:4]]
# world 8 = F2V_PUSH_MAX_RANKS, what the driver's scaling run uses: 16-row minibatches of karate (2 rows per rank, a 2-row tail: six
# empty slices), 5-row minibatches (three empty slices in every minibatch, option 7), cora (32-row slices, a 148-row tail)
PAIRS += [(CASES[0], 8), (CASES[3], 8), (CASES[5], 8)]


@pytest.mark.parametrize("case,world", PAIRS, ids=["%s-w%d" % (i, w) for i, (c, w) in zip(_ids([p[0] for p in PAIRS]), PAIRS)])
def test_sharded_training_equals_single_process(case, world, tmp_path):
    import torch.multiprocessing as mp
    graph, option, iters, batch, dim, bs = case
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    rp, ci = O.read_mtx(golden_graph_path(graph))
    want = O.train(option, rp, ci, dim, iters, batch, bs_mode=bs, order=O.ORDER_TREE, chunk=4)
    for r in range(world):
        got = np.load(str(tmp_path / ("r%d.npy" % r)))
        assert np.array_equal(got, want), (r, float(np.nanmax(np.abs(got - want))))


NEED_PAIRS = [(c, w) for c, w in PAIRS if c[1] != 7]


@pytest.mark.parametrize("case,world", NEED_PAIRS, ids=["%s-w%d" % (i, w) for i, (c, w) in zip(_ids([p[0] for p in NEED_PAIRS]), NEED_PAIRS)])
def test_per_destination_exchange_equals_single_process(case, world, tmp_path):
    """NeedExchange: a rank receives only the rows it reads (CSR neighbours of its rows + every sampled vertex);
    rows nobody reads stay NaN in the stand-in engine, so a missing row would poison the result."""
    import torch.multiprocessing as mp
    graph, option, iters, batch, dim, bs = case
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path), "need"), nprocs=world, join=True)
    rp, ci = O.read_mtx(golden_graph_path(graph))
    want = O.train(option, rp, ci, dim, iters, batch, bs_mode=bs, order=O.ORDER_TREE, chunk=4)
    sent = 0
    for r in range(world):
        got = np.load(str(tmp_path / ("r%d.npy" % r)))
        assert np.array_equal(got, want), (r, int(np.isnan(got).sum()))
        sent += int(np.load(str(tmp_path / ("sent%d.npy" % r)))[0])
    assert sent <= (world - 1) * (len(rp) - 1)   # never more than the all-gather moves


def test_row_owner_matches_shard_bounds():
    from force2vec_amd.dist import row_owner, shard_bounds
    for n, batch, world in ((34, 16, 2), (34, 7, 3), (100, 100, 8), (2708, 256, 3), (9, 4, 4)):
        b, owner = row_owner(n, batch, world)
        for lo in range(0, n, batch):
            hi = min(lo + batch, n)
            for r in range(world):
                _, a, e = shard_bounds(lo, hi, r, world)
                assert np.all(owner[a:e] == r) and np.all(b[a:e] == lo // batch)


def test_shard_bounds_cover_the_batch():
    from force2vec_amd.dist import shard_bounds
    for lo, hi in ((0, 1), (0, 7), (10, 26), (5, 5 + 65536), (3, 3 + 100)):
        for world in (1, 2, 3, 8):
            seen = []
            pers = set()
            for r in range(world):
                per, a, b = shard_bounds(lo, hi, r, world)
                pers.add(per)
                assert lo <= a <= b <= hi and b - a <= per
                seen += list(range(a, b))
            assert seen == list(range(lo, hi)) and len(pers) == 1


def _partition_worker(rank, world, port, graph, option, iters, batch, dim, outdir):
    import torch.distributed as dist
    from fake_engine import OracleEngine
    from force2vec_amd import dist as fdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rp, ci = O.read_mtx(golden_graph_path(graph))
    start, new_id = fdist.vertex_partition(len(rp) - 1, world, batch)
    rp2, ci2 = fdist.relabel_csr(rp, ci, new_id)
    eng = OracleEngine(rp2, ci2, dim, chunk=4)
    eng.srand(1)
    eng.init_embeddings(0 if option == 5 else 1)
    fdist.ShardedTrainer(eng, rank, world, fdist.HostStageComm(dist, rank, world)).train(option, iters, world * batch, 5, 0.02, 0)
    np.save(os.path.join(outdir, "r%d.npy" % rank), eng.get_embeddings())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("graph,option,batch,world", [("cora.mtx", 5, 128, 2), ("karate.mtx", 6, 5, 3), ("cora.mtx", 6, 300, 2)])
def test_vertex_partition_mode_is_the_reference_on_the_relabelled_graph(graph, option, batch, world, tmp_path):
    """SURVEY 8e, second mode: every rank owns a contiguous vertex range and steps its own local minibatch k while the others
    step theirs.  force2vec_amd.dist.vertex_partition gives the relabelling under which that is the reference's algorithm
    with batch world*B; checked as the survey prescribes: (1) the sharded run on the relabelled graph gives, on every rank,
    exactly what the oracle gives on the relabelled graph with -batch world*B; (2) in every minibatch, rank r's slice is a
    contiguous run of ITS OWN original vertices; (3) where the genuine reference binary is present, it agrees on the permuted
    .mtx (fp32 order tolerance)."""
    import torch.multiprocessing as mp
    from force2vec_amd import dist as fdist
    from force2vec_amd.graph import edges_from_csr, write_mtx_symmetric
    dim, iters = 16, 3
    rp, ci = O.read_mtx(golden_graph_path(graph))
    n = len(rp) - 1
    start, new_id = fdist.vertex_partition(n, world, batch)
    assert sorted(new_id.tolist()) == list(range(n)) and start[0] == 0 and start[-1] == n
    old_of_new = np.empty(n, dtype=np.int64)
    old_of_new[new_id] = np.arange(n)
    big = world * batch
    for lo in range(0, n, big):
        hi = min(lo + big, n)
        for r in range(world):
            _, a, b = fdist.shard_bounds(lo, hi, r, world)
            mine = old_of_new[a:b]
            assert np.all((mine >= start[r]) & (mine < start[r + 1])) and np.all(np.diff(mine) == 1)
    rp2, ci2 = fdist.relabel_csr(rp, ci, new_id)
    # the relabelled CSR is the same graph: edge (u, v) <-> (new_id[u], new_id[v]), neighbour ids ascending
    deg2 = np.diff(rp2.astype(np.int64))
    assert np.array_equal(deg2[new_id], np.diff(rp.astype(np.int64)))
    for v in (0, n // 2, n - 1):
        assert sorted(new_id[ci[rp[v]:rp[v + 1]].astype(np.int64)].tolist()) == ci2[rp2[new_id[v]]:rp2[new_id[v] + 1]].tolist()
    mp.spawn(_partition_worker, args=(world, _free_port(), graph, option, iters, batch, dim, str(tmp_path)), nprocs=world, join=True)
    want = O.train(option, rp2, ci2, dim, iters, big, order=O.ORDER_TREE, chunk=4)
    for r in range(world):
        assert np.array_equal(np.load(str(tmp_path / ("r%d.npy" % r))), want)
    if O.ref_binary() is not None:
        mtx = str(tmp_path / "permuted.mtx")
        src, dst = edges_from_csr(rp2, ci2)
        write_mtx_symmetric(mtx, n, src, dst)
        rp3, ci3 = O.read_mtx(mtx)
        assert np.array_equal(rp3, rp2) and np.array_equal(ci3, ci2)  # what the reference will read is the relabelled graph
        path, _ = O.run_reference(mtx, str(tmp_path), option, iters, big, dim)
        assert np.abs(O.read_embd(path) - want).max() < 1e-5

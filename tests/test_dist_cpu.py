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


def _worker(rank, world, port, case, outdir):
    import torch.distributed as dist
    from fake_engine import OracleEngine
    from force2vec_amd import dist as fdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    graph, option, iters, batch, dim, bs = case
    rp, ci = O.read_mtx(golden_graph_path(graph))
    eng = OracleEngine(rp, ci, dim, chunk=4)
    eng.srand(1)
    eng.init_embeddings(0 if option == 5 else 1)
    fdist.ShardedTrainer(eng, rank, world, fdist.HostStageComm(dist, rank, world)).train(option, iters, batch, 5, 0.02, bs)
    np.save(os.path.join(outdir, "r%d.npy" % rank), eng.get_embeddings())
    dist.barrier()
    dist.destroy_process_group()


CASES = [("karate.mtx", 5, 3, 16, 16, 0), ("karate.mtx", 5, 2, 7, 16, 1), ("karate.mtx", 6, 3, 16, 32, 0),
         ("karate.mtx", 7, 3, 5, 16, 0), ("karate.mtx", 5, 2, 64, 16, 0), ("cora.mtx", 5, 2, 256, 16, 0)]


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", CASES, ids=["%s-opt%d-B%d-bs%d" % (c[0][:-4], c[1], c[3], c[5]) for c in CASES])
def test_sharded_training_equals_single_process(case, world, tmp_path):
    import torch.multiprocessing as mp
    graph, option, iters, batch, dim, bs = case
    mp.spawn(_worker, args=(world, _free_port(), case, str(tmp_path)), nprocs=world, join=True)
    rp, ci = O.read_mtx(golden_graph_path(graph))
    want = O.train(option, rp, ci, dim, iters, batch, bs_mode=bs, order=O.ORDER_TREE, chunk=4)
    for r in range(world):
        got = np.load(str(tmp_path / ("r%d.npy" % r)))
        assert np.array_equal(got, want), (r, float(np.nanmax(np.abs(got - want))))


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

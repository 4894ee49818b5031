"""-m gpu: the multi-process path with the real HIP engines.  The GPU box has one card, so
(a) two gloo ranks share GPU 0 and exchange staged rows through the host, and (b) a
single-rank RCCL group exercises the zero-copy in-place all-gather on the engine's stream.
Both must reproduce the single-engine f2v_train bit for bit."""
import os
import socket

import numpy as np
import pytest

from conftest import golden_graph_path

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _single(case):
    import force2vec_amd as F
    graph, option, iters, batch, dim, bs = case
    rp, ci = F.read_mtx(golden_graph_path(graph))
    a = F.algorithms((rp, ci), dim=dim)
    a.srand(1)
    a._run(option, bs, iters, batch, 5, 0.02, write=False)
    X = a.nCoordinates
    a.engine.close()
    return X


def _gloo_worker(rank, world, port, case, outdir, exchange="allgather"):
    import torch.distributed as dist
    import force2vec_amd as F
    from force2vec_amd import dist as fdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    graph, option, iters, batch, dim, bs = case
    rp, ci = F.read_mtx(golden_graph_path(graph))
    eng = F.Engine(rp, ci, dim, device=0)
    eng.srand(1)
    eng.init_embeddings(0 if option in (5, 8, 11) else 1)
    replicated = exchange == "push_replicated"
    if replicated:
        exchange = "push"
    if exchange in ("push", "push_unfused", "push_landing", "push_landing_unfused"):
        # the engine's own exchange: peers' matrices mapped through HIP IPC (here: other processes on the same GPU),
        # rows pushed by a HIP kernel, device-side flag barrier; gloo only carries the handles
        comm = fdist.PushExchange(dist, rank, world)
        eng.set_param("push_timeout_ms", 8000)
        if exchange.startswith("push_landing"):  # what matrices of 2 GiB and more use: rows travel through a landing buffer
            eng.set_param("push_landing", 1)
        if exchange.endswith("unfused"):  # rows pushed by a kernel of their own behind the step instead of by the step itself
            eng.set_param("push_fused", 0)
        exchange = "push"
        if replicated:  # (1, the default, replicates only where no two ranks share a card: here they all do)
            assert eng.get_param("replicate_small") == 1
            eng.set_param("replicate_small", 2)
    elif exchange == "allgather":
        comm = fdist.HostStageComm(dist, rank, world)
    elif exchange == "need":
        comm = fdist.NeedExchange(dist, rank, world, backend="host")
    elif exchange == "nccl_refused":
        # the insurance path of NcclStageComm: the device exchange raises on first use -> host bounce over the gloo group
        import torch
        torch.cuda.set_device(0)
        comm = fdist.NcclStageComm(dist, rank, world, 0, host_group=dist.group.WORLD)

        def refuse(*a, **k):
            raise RuntimeError("simulated RCCL refusal")
        comm._device_exchange = refuse
    else:  # device-side pack / unpack on the engine's stream, collectives bounced through the host (gloo)
        import torch
        torch.cuda.set_device(0)
        comm = fdist.NeedExchange(dist, rank, world, device=0, backend="device")
    tr = fdist.ShardedTrainer(eng, rank, world, comm)
    tr.train(option, iters, batch, 5, 0.02, bs)
    if replicated:
        # a minibatch small enough to chain: every rank ran the whole epochs itself, in the chained form, and pushed nothing ...
        assert eng.get_param("last_train_replicated") == 1 and eng.get_param("last_train_form") in (1, 2)
        assert eng.push_stats() == {"rows_pushed": 0, "rows_allgather": 0}
        # ... and the same attachment shards the next call (one minibatch per epoch cannot chain): the ranks still hold the same matrix
        tr.train(option, 1, eng.n, 5, 0.02, bs)
        st = eng.push_stats()
        assert eng.get_param("last_train_replicated") == 0 and 0 < st["rows_pushed"] <= st["rows_allgather"]
    elif exchange == "push":
        # a second run on the same attachment continues where the first stopped, exactly like two f2v_train calls
        tr.train(option, 1, batch, 5, 0.02, bs)
        st = eng.push_stats()
        assert 0 < st["rows_pushed"] <= st["rows_allgather"]
    np.save(os.path.join(outdir, "r%d.npy" % rank), eng.get_embeddings())
    if exchange == "push":
        comm.detach(eng)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def _single_twice(case):
    import force2vec_amd as F
    graph, option, iters, batch, dim, bs = case
    rp, ci = F.read_mtx(golden_graph_path(graph))
    eng = F.Engine(rp, ci, dim, device=0)
    eng.srand(1)
    eng.init_embeddings(0 if option in (5, 8, 11) else 1)
    eng.train(option, iters, batch, 5, 0.02, bs)
    eng.train(option, 1, batch, 5, 0.02, bs)
    X = eng.get_embeddings()
    eng.close()
    return X


@pytest.mark.parametrize("world", [2, 3])
@pytest.mark.parametrize("case", [("cora.mtx", 5, 3, 256, 128, 0), ("cora.mtx", 6, 2, 300, 64, 0), ("pubmed.mtx", 5, 2, 4096, 128, 0),
                                  ("karate.mtx", 7, 3, 16, 64, 0), ("cora.mtx", 5, 2, 300, 100, 1), ("citeseer.mtx", 11, 2, 1000, 32, 0),
                                  ("cora.mtx", 6, 2, 256, 66, 0)])
def test_push_exchange_between_processes_sharing_the_gpu(case, world, tmp_path):
    """f2v_train_sharded: every rank maps its peers' matrices and flags through HIP IPC, pushes its new rows into
    the peers that read them and passes the device-side flag barrier -- here between processes on ONE card (the
    peer mappings then alias memory of the same GPU; IPC, masks, push kernel, barrier protocol and the final
    completion pass are the code that runs across xGMI).  Bit-identical to the single-engine run on every rank."""
    import torch.multiprocessing as mp
    if case[0] == "pubmed.mtx" and world == 3:
        pytest.skip("covered by world 2")
    mp.spawn(_gloo_worker, args=(world, _free_port(), case, str(tmp_path), "push"), nprocs=world, join=True)
    want = _single_twice(case)
    for r in range(world):
        assert np.array_equal(np.load(str(tmp_path / ("r%d.npy" % r))), want)


def _single_then_whole(case):
    import force2vec_amd as F
    graph, option, iters, batch, dim, bs = case
    rp, ci = F.read_mtx(golden_graph_path(graph))
    eng = F.Engine(rp, ci, dim, device=0)
    eng.srand(1)
    eng.init_embeddings(0 if option in (5, 8, 11) else 1)
    eng.train(option, iters, batch, 5, 0.02, bs)
    eng.train(option, 1, eng.n, 5, 0.02, bs)
    X = eng.get_embeddings()
    eng.close()
    return X


@pytest.mark.parametrize("case", [("cora.mtx", 5, 40, 256, 128, 0), ("cora.mtx", 6, 3, 300, 64, 0), ("karate.mtx", 7, 3, 16, 64, 0)])
def test_sharded_call_at_a_chainable_batch_runs_replicated(case, tmp_path):
    """f2v_train_sharded at a batch size that chains ("replicate_small"): an epoch of small minibatches is one row-to-row dependency chain,
    which hops over xGMI can only lengthen, so every rank runs the whole call itself in the chained form and nothing is exchanged; the
    ranks end with the same matrix -- f2v_train's, bit for bit -- and the same rand() state, and the next, sharded call of the same
    attachment (whose pushes land in matrices that must therefore already agree) continues from it."""
    import torch.multiprocessing as mp
    world = 2
    mp.spawn(_gloo_worker, args=(world, _free_port(), case, str(tmp_path), "push_replicated"), nprocs=world, join=True)
    want = _single_then_whole(case)
    for r in range(world):
        assert np.array_equal(np.load(str(tmp_path / ("r%d.npy" % r))), want)


def _rmat_push_worker(rank, world, port, outdir):
    import torch.distributed as dist
    import force2vec_amd as F
    from force2vec_amd import dist as fdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rp, ci = F.rmat_csr(14, 16, 3)
    eng = F.Engine(rp, ci, 128, device=0)
    eng.srand(1)
    eng.init_embeddings(0)
    comm = fdist.PushExchange(dist, rank, world)
    fdist.ShardedTrainer(eng, rank, world, comm).train(5, 3, 16384, 5, 0.02, 0)
    np.save(os.path.join(outdir, "r%d.npy" % rank), eng.get_embeddings())
    open(os.path.join(outdir, "chunk%d.txt" % rank), "w").write(str(eng.get_param("hub_chunk")))
    comm.detach(eng)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


def test_push_exchange_hub_chunk_follows_the_slice(tmp_path):
    """A rank's launch covers batch/world rows, so the automatic hub chunk is chosen for the slice (a whole batch's
    chunk would keep the step kernel as long as on one GPU).  The chunk is part of the summation order: the sharded
    run equals the single engine run with THAT chunk bit for bit."""
    import torch.multiprocessing as mp
    import force2vec_amd as F
    mp.spawn(_rmat_push_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    chunk = int(open(str(tmp_path / "chunk0.txt")).read())
    rp, ci = F.rmat_csr(14, 16, 3)
    eng = F.Engine(rp, ci, 128, device=0)
    eng.set_param("hub_chunk_for_batch", 16384)
    assert eng.get_param("hub_chunk") > chunk  # the case really exercises a slice-sized chunk
    eng.set_param("hub_chunk", chunk)
    eng.srand(1)
    eng.init_embeddings(0)
    eng.train(5, 3, 16384, 5, 0.02, 0)
    want = eng.get_embeddings()
    eng.close()
    for r in range(2):
        assert np.array_equal(np.load(str(tmp_path / ("r%d.npy" % r))), want)


def _push_timeout_worker(rank, world, port, outdir):
    import torch.distributed as dist
    import force2vec_amd as F
    from force2vec_amd import dist as fdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rp, ci = F.read_mtx(golden_graph_path("karate.mtx"))
    eng = F.Engine(rp, ci, 64, device=0)
    eng.srand(1)
    eng.init_embeddings(0)
    eng.set_param("push_timeout_ms", 300)
    comm = fdist.PushExchange(dist, rank, world)
    comm.attach(eng)
    msg = "ok"
    if rank == 0:  # rank 1 never trains: rank 0's first barrier must give up, not hang the GPU
        try:
            eng.train_sharded(5, 1, 16, 5, 0.02, 0)
        except F.F2VError as ex:
            msg = str(ex)
    open(os.path.join(outdir, "r%d.txt" % rank), "w").write(msg)
    dist.barrier()
    eng.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("case", [("cora.mtx", 5, 3, 256, 128, 0), ("pubmed.mtx", 6, 2, 4096, 128, 0), ("cora.mtx", 5, 2, 300, 100, 1),
                                  ("cora.mtx", 5, 2, 256, 260, 0), ("cora.mtx", 6, 2, 256, 20, 0), ("cora.mtx", 5, 2, 256, 512, 0),
                                  ("cora.mtx", 5, 2, 256, 30, 0)])
def test_push_exchange_with_a_separate_push_kernel(case, tmp_path):
    """ "push_fused" = 0: the same exchange with the rows copied to the peers by push_rows_kernel after the step."""
    import torch.multiprocessing as mp
    mp.spawn(_gloo_worker, args=(2, _free_port(), case, str(tmp_path), "push_unfused"), nprocs=2, join=True)
    want = _single_twice(case)
    for r in range(2):
        assert np.array_equal(np.load(str(tmp_path / ("r%d.npy" % r))), want)


@pytest.mark.parametrize("mode", ["push_landing", "push_landing_unfused"])
@pytest.mark.parametrize("case", [("cora.mtx", 5, 3, 256, 128, 0), ("pubmed.mtx", 6, 2, 4096, 128, 0), ("karate.mtx", 7, 3, 16, 64, 0),
                                  ("cora.mtx", 5, 2, 300, 100, 1)])
def test_push_exchange_through_landing_buffers(case, mode, tmp_path):
    """ "push_landing" = 1 (automatic for matrices of 2 GiB and more, which HIP IPC cannot map): the peers push a
    minibatch's rows into a small mapped buffer, alternating halves, and unpack_rows_kernel moves them into the
    matrix behind the barrier; the final completion pass goes minibatch by minibatch through the same buffer."""
    import torch.multiprocessing as mp
    world = 3 if case[0] == "cora.mtx" and mode == "push_landing" else 2
    mp.spawn(_gloo_worker, args=(world, _free_port(), case, str(tmp_path), mode), nprocs=world, join=True)
    want = _single_twice(case)
    for r in range(world):
        assert np.array_equal(np.load(str(tmp_path / ("r%d.npy" % r))), want)


@pytest.mark.parametrize("mode", ["push", "push_unfused", "push_landing", "push_landing_unfused"])
def test_push_exchange_with_skewed_ranks(mode, tmp_path, monkeypatch):
    """F2V_PUSH_CHAOS: every rank drains its stream and sleeps up to 3 ms at random minibatches (other ones on every
    rank), so ranks run ahead of and behind each other by whole minibatches: the barrier, the two landing-buffer halves
    and the epoch's matrix swap must keep every replica bit-identical all the same."""
    import torch.multiprocessing as mp
    from force2vec_amd import _lib
    monkeypatch.setenv("F2V_LIBRARY", _lib.SELFTEST_LIB_PATH)  # the spawned ranks load the self-test build: fault injection lives there
    monkeypatch.setenv("F2V_PUSH_CHAOS", "7")
    case = ("pubmed.mtx", 5, 4, 1024, 64, 0)
    mp.spawn(_gloo_worker, args=(3, _free_port(), case, str(tmp_path), mode), nprocs=3, join=True)
    monkeypatch.delenv("F2V_PUSH_CHAOS")
    monkeypatch.delenv("F2V_LIBRARY")
    want = _single_twice(case)
    for r in range(3):
        assert np.array_equal(np.load(str(tmp_path / ("r%d.npy" % r))), want)


def test_push_barrier_times_out_instead_of_hanging(tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_push_timeout_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    assert "did not reach the xGMI barrier" in open(str(tmp_path / "r0.txt")).read()


@pytest.mark.parametrize("mode", ["need", "need_device", "nccl_refused"])
@pytest.mark.parametrize("case", [("cora.mtx", 5, 3, 256, 128, 0), ("cora.mtx", 6, 2, 300, 64, 1)])
def test_two_gloo_ranks_per_destination_exchange(case, mode, tmp_path):
    """The same with NeedExchange: rows travel only to the ranks that read them, replicas are completed at the end.
    "need_device" packs / unpacks with torch ops on zero-copy views of the engine's matrices, on its stream."""
    import torch.multiprocessing as mp
    mp.spawn(_gloo_worker, args=(2, _free_port(), case, str(tmp_path), mode), nprocs=2, join=True)
    want = _single(case)
    for r in range(2):
        assert np.array_equal(np.load(str(tmp_path / ("r%d.npy" % r))), want)


@pytest.mark.parametrize("case", [("cora.mtx", 5, 3, 256, 128, 0), ("cora.mtx", 6, 2, 384, 128, 0), ("karate.mtx", 7, 3, 16, 64, 0),
                                  ("cora.mtx", 5, 2, 300, 64, 1)])
def test_two_gloo_ranks_share_the_gpu(case, tmp_path):
    import torch.multiprocessing as mp
    mp.spawn(_gloo_worker, args=(2, _free_port(), case, str(tmp_path)), nprocs=2, join=True)
    want = _single(case)
    for r in range(2):
        assert np.array_equal(np.load(str(tmp_path / ("r%d.npy" % r))), want)


def _nccl_worker(rank, world, port, case, outdir, exchange="allgather"):
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    import force2vec_amd as F
    from force2vec_amd import dist as fdist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    graph, option, iters, batch, dim, bs = case
    rp, ci = F.read_mtx(golden_graph_path(graph))
    eng = F.Engine(rp, ci, dim, device=0)
    eng.srand(1)
    eng.init_embeddings(0)
    comm = fdist.NcclStageComm(dist, rank, world, 0) if exchange == "allgather" else fdist.NeedExchange(dist, rank, world, device=0, backend="device")
    fdist.ShardedTrainer(eng, rank, world, comm, exchange_when_single=True).train(option, iters, batch, 5, 0.02, bs)
    np.save(os.path.join(outdir, "r%d.npy" % rank), eng.get_embeddings())
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["allgather", "need"])
def test_rccl_in_place_all_gather_on_engine_stream(exchange, tmp_path):
    """Single-rank RCCL group: the zero-copy views, the engine's stream as torch ExternalStream, the in-place
    all-gather (per minibatch for "allgather", the final completion pass for "need")."""
    import torch.multiprocessing as mp
    case = ("cora.mtx", 5, 3, 256, 128, 0)
    mp.spawn(_nccl_worker, args=(1, _free_port(), case, str(tmp_path), exchange), nprocs=1, join=True)
    assert np.array_equal(np.load(str(tmp_path / "r0.npy")), _single(case))


def test_push_exchange_at_world_8_in_one_process():
    """World 8 = F2V_PUSH_MAX_RANKS (the driver's scaling run) had never executed anywhere: kMaxRanks peer tables, reader-mask bit 7,
    the 8-lane xgmi_barrier_kernel, pushes into 7 peers.  A GPU box admits at most 6 processes on its card, so the rehearsal runs 8
    ENGINES in one process (self-test build, peers attached by direct pointers, one host thread and one stream per rank):
    tools/push_world_local.py -- cora option 5, pubmed option 6, karate option 7 with 5-row minibatches (empty slices), cora option
    11 at D = 100 through landing buffers with the separate push kernel; every replica bit-identical to the single-engine run."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "push_world_local.py"), "8"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("replicas identical to the single engine: all 8") == 4, r.stdout


def test_push_exchange_at_world_5_through_ipc(tmp_path):
    """The most ranks-as-processes one card of this pool takes (6 GPU processes, the test runner being one): every rank attaches
    FOUR peers through hipIpcOpenMemHandle; bit-identical on every rank."""
    import torch.multiprocessing as mp
    case = ("cora.mtx", 5, 3, 256, 128, 0)
    mp.spawn(_gloo_worker, args=(5, _free_port(), case, str(tmp_path), "push"), nprocs=5, join=True)
    want = _single_twice(case)
    for r in range(5):
        assert np.array_equal(np.load(str(tmp_path / ("r%d.npy" % r))), want)


def test_bench_launches_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with no launcher around it: bench.py starts torch.distributed.run itself as a child process
    (before it has touched the GPU), relays rank 0's JSON line as its ONLY stdout line and returns the ranks' exit code.
    Rehearsed here with two gloo ranks sharing the one card; on an 8-GPU node the same path runs over RCCL / xGMI."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--scale", "15", "--batch", "8192",
                        "--steps", "2", "--warmup", "1", "--settle-ms", "0", "--config5-scale", "0", "--config4", "0", "--dist-extra-batches", ""],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["scaling"] == "strong" and "failed" not in res
    assert res["config"]["replicas_bit_identical_to_1gpu_run"] is True

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
    if exchange == "allgather":
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
    fdist.ShardedTrainer(eng, rank, world, comm).train(option, iters, batch, 5, 0.02, bs)
    np.save(os.path.join(outdir, "r%d.npy" % rank), eng.get_embeddings())
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


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

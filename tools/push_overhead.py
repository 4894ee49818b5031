#!/usr/bin/env python3
"""What the push exchange costs besides the xGMI transfer itself, measured on ONE GPU: `world` processes share
GPU 0, map each other's matrices through HIP IPC and run f2v_train_sharded.  The ranks' step kernels then share
one card, so the ideal epoch time equals the single-GPU epoch; what is measured on top is push kernels (local
copies here), flag barriers and launch gaps.   usage: push_overhead.py [world] [batch] [epochs] [scale]"""
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, batch, epochs, scale):
    import torch.distributed as dist
    import bench
    import force2vec_amd as F
    from force2vec_amd import dist as fdist
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    rowptr, colids = bench.load_graph(scale, 16, 1)
    eng = F.Engine(rowptr, colids, 128, device=0)
    eng.srand(1)
    eng.init_embeddings(0)
    single = None
    if rank == 0:
        eng.train(5, 2, batch)
        single = eng.train(5, epochs, batch) / epochs
        eng.srand(1)
        eng.init_embeddings(0)
    comm = fdist.PushExchange(dist, rank, world)
    comm.attach(eng)
    eng.train_sharded(5, 2, batch)
    dist.barrier()
    t0 = time.perf_counter()
    dev = eng.train_sharded(5, epochs, batch) / epochs
    eng.synchronize()
    dist.barrier()
    wall = (time.perf_counter() - t0) / epochs
    st = eng.push_stats()
    if rank == 0:
        nb = -(-(len(rowptr) - 1) // batch)
        print("world %d batch %d: single GPU %.3f ms/epoch; %d ranks sharing the GPU: %.3f ms/epoch device, %.3f ms wall "
              "(+%.1f us per minibatch); rows pushed / all-gather rows = %.3f" %
              (world, batch, single * 1e3, world, dev * 1e3, wall * 1e3, (dev - single) / nb * 1e6, st["rows_pushed"] / max(st["rows_allgather"], 1)), flush=True)
    comm.detach(eng)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 10
    scale = int(sys.argv[4]) if len(sys.argv) > 4 else 20
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(world, port, batch, epochs, scale), nprocs=world, join=True)

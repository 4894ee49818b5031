#!/usr/bin/env python3
"""How large an embedding matrix can be mapped into a peer process through HIP IPC?  Two processes on GPU 0, an edgeless
graph of `rows` vertices (D = 128: rows * 512 bytes per matrix); every step of the attachment is timed and printed.
usage: ipc_probe.py [rows ...]"""
import os
import socket
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, port, rows_list):
    import numpy as np
    import torch.distributed as dist
    import force2vec_amd as F
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)

    def say(msg):
        print("[rank %d] %s" % (rank, msg), flush=True)

    for rows in rows_list:
        rp = np.zeros(rows + 1, dtype=np.uint32)
        ci = np.zeros(0, dtype=np.uint32)
        eng = F.Engine(rp, ci, 128, device=0)
        eng.set_param("fast_rng", 1)
        eng.init_embeddings(0)
        eng.set_param("push_timeout_ms", 3000)
        say("rows %d (%.2f GiB per matrix): engine ready" % (rows, rows * 512 / 2**30))
        t0 = time.time()
        mine = eng.push_export()
        say("  export %.3fs" % (time.time() - t0))
        allh = [None] * world
        dist.all_gather_object(allh, mine)
        t0 = time.time()
        eng.push_attach(rank, world, allh)
        say("  attach %.3fs" % (time.time() - t0))
        dist.barrier()
        t0 = time.time()
        eng.push_selftest()
        say("  selftest %.3fs" % (time.time() - t0))
        dist.barrier()
        eng.push_detach()
        eng.close()
        dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    rows_list = [int(x) for x in sys.argv[1:]] or [1 << 20, 3 << 20, 4 << 20, 4194304 + 8, 6 << 20]
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    import torch.multiprocessing as mp
    mp.spawn(worker, args=(2, port, rows_list), nprocs=2, join=True)

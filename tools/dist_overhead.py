#!/usr/bin/env python3
"""Measures what the Python multi-GPU driver costs per epoch against the native f2v_train loop, on ONE
GPU: a single-rank RCCL group with the exchange forced on (in-place all-gather of world size 1)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

import bench
import force2vec_amd as F
from force2vec_amd import dist as fdist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
rowptr, colids = bench.load_graph(20, 16, 1)
nnz = len(colids)
for batch in (65536, 16384, 4096):
    eng = F.Engine(rowptr, colids, 128)
    eng.srand(1)
    eng.init_embeddings(0)
    eng.train(5, 2, batch)
    eng.synchronize()
    t0 = time.perf_counter(); eng.train(5, 5, batch); eng.synchronize(); t_native = (time.perf_counter() - t0) / 5
    for exch in (False, True):
        tr = fdist.ShardedTrainer(eng, 0, 1, fdist.NcclStageComm(dist, 0, 1, 0), exchange_when_single=exch)
        tr.train(5, 1, batch); eng.synchronize(); torch.cuda.synchronize()
        t0 = time.perf_counter(); tr.train(5, 5, batch); eng.synchronize(); torch.cuda.synchronize(); t = (time.perf_counter() - t0) / 5
        print("batch %6d  native %.3f ms/epoch   python driver%s %.3f ms/epoch  (+%.1f us per minibatch)" %
              (batch, t_native * 1e3, " + all-gather" if exch else "", t * 1e3, (t - t_native) / (-(-len(rowptr) // batch)) * 1e6), flush=True)
    eng.close()
dist.destroy_process_group()

"""Multi-GPU driver: one process per GPU, every minibatch's rows sharded across the ranks.

How the path shards ("exact mode"): forces only ever update the SOURCE row, rows of
one minibatch are independent given the pre-batch matrix, and the next minibatch must see all
of them.  So the graph and the N x D matrix are replicated in every GPU's HBM (8 GiB at
RMAT-24 against 288 GB), rank r computes the r-th contiguous slice of each minibatch's rows,
and ONE exchange per minibatch -- an all-gather of the staged new rows, B x D fp32 in total --
makes every replica identical again.  Results are bit-identical to the single-GPU run for any
world size.  Every rank draws the same rand() stream on its host.

The exchange is an in-place RCCL all-gather on those rows of the second matrix (torch.distributed backend
"nccl", zero-copy on the engine's own HIP stream); a host-bounce variant serves gloo."""
import numpy as np


def shard_bounds(lo, hi, rank, world):
    """Contiguous equal slices of minibatch [lo,hi): -> (per, my_lo, my_hi); trailing ranks may be empty."""
    per = -(-(hi - lo) // world)
    my_lo = min(lo + rank * per, hi)
    return per, my_lo, min(my_lo + per, hi)


def math_of_option(option):
    return {5: 5, 8: 5, 11: 5, 6: 6, 9: 6, 7: 7, 10: 7}[option]


class HostStageComm:
    """All-gather of the staged rows through host memory (any torch.distributed backend, e.g. gloo)."""

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world

    def prepare(self, engine, max_batch_rows):
        pass

    def exchange(self, engine, lo, hi):
        import torch
        per, my_lo, my_hi = shard_bounds(lo, hi, self.rank, self.world)
        mine = torch.zeros((per, engine.dim), dtype=torch.float32)
        if my_hi > my_lo:
            mine[: my_hi - my_lo] = torch.from_numpy(engine.stage_read(my_lo, my_hi))
        parts = [torch.empty_like(mine) for _ in range(self.world)]
        self.dist.all_gather(parts, mine)
        for r in range(self.world):
            if r == self.rank:
                continue
            _, r_lo, r_hi = shard_bounds(lo, hi, r, self.world)
            if r_hi > r_lo:
                engine.stage_write(r_lo, r_hi, parts[r][: r_hi - r_lo].numpy())


class _DevBuf:
    def __init__(self, ptr, nfloats):
        self.__cuda_array_interface__ = {"shape": (nfloats,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class NcclStageComm:
    """In-place RCCL all-gather over xGMI on the new rows inside the engine's second matrix, on its HIP stream."""

    def __init__(self, dist, rank, world, device):
        import torch
        self.torch, self.dist, self.rank, self.world, self.device = torch, dist, rank, world, device
        self._views = {}
        self._stream = None

    def prepare(self, engine, max_batch_rows):
        per = -(-max_batch_rows // self.world)
        engine.stage_reserve(per * self.world)  # room for the padded all-gather
        self._stream = self.torch.cuda.ExternalStream(engine.stream(), device=self.device)

    def _matrix_view(self, engine, lo):
        """torch view of the whole matrix the staged rows live in (two matrices alternate: two views, made once)."""
        ptr, cap = engine.stage_device_ptr()          # address of staged row `lo`, rows available from there
        base = ptr - lo * engine.dim * 4
        v = self._views.get(base)
        if v is None:
            v = self.torch.as_tensor(_DevBuf(base, (cap + lo) * engine.dim), device=self.torch.device("cuda", self.device))
            self._views[base] = v
        return v

    def exchange(self, engine, lo, hi):
        per, _, _ = shard_bounds(lo, hi, self.rank, self.world)
        d = engine.dim
        full = self._matrix_view(engine, lo)[lo * d: (lo + per * self.world) * d]
        mine = full[self.rank * per * d: (self.rank + 1) * per * d]
        with self.torch.cuda.stream(self._stream):
            self.dist.all_gather_into_tensor(full, mine)


class ShardedTrainer:
    """AlgoForce2Vec* over `world` engines: same epochs, minibatches and rand() order as f2v_train."""

    def __init__(self, engine, rank, world, comm, exchange_when_single=False):
        self.engine, self.rank, self.world, self.comm = engine, rank, world, comm
        self.exchange_when_single = exchange_when_single  # self-test of the exchange path on one rank

    def train(self, option, iters, batch, ns=5, lr=0.02, bs_mode=0):
        e = self.engine
        n = e.n
        math = math_of_option(option)
        if math == 7 and bs_mode:
            raise ValueError("option 7 has no -bs 1 variant")
        nb = -(-n // batch)
        if e.get_param("hub_chunk_auto"):
            e.set_param("hub_chunk_for_batch", batch)  # same chunk as the single-GPU f2v_train, whatever the world size
        self.comm.prepare(e, min(batch, n))
        ndraw = ns * batch if bs_mode else ns
        stride = (min(batch, n) + ns) if bs_mode else ns
        for _ in range(iters):
            if math == 7:
                e.generate_walks()  # same stream on every rank: sample/algorithms.cpp:1097-1118
            # the epoch's sample ids do not depend on the embeddings: draw them now (option 7 draws them after
            # the walks, as the reference's rand() order has it) and keep them in HBM -- no sync per minibatch
            ids = np.zeros(nb * stride, dtype=np.uint32)
            for b in range(nb):
                maxv = min((b + 1) * batch, n - 1) if math == 7 else n - 1  # algorithms.cpp:1125
                keep = min(stride, ndraw)  # -bs 1 draws ns*BATCH ids of which rows+ns-1 are ever read
                ids[b * stride: b * stride + keep] = e.draw_samples(maxv, ndraw, keep)
            e.upload_sample_ids(ids)
            for b in range(nb):
                lo, hi = b * batch, min((b + 1) * batch, n)
                _, my_lo, my_hi = shard_bounds(lo, hi, self.rank, self.world)
                e.minibatch_step_at(option, lo, hi, b * stride, ns, lr, bs_mode, row_lo=my_lo, row_hi=my_hi)
                if self.world > 1 or self.exchange_when_single:
                    self.comm.exchange(e, lo, hi)
        e.flush()

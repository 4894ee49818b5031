"""Multi-GPU driver: one process per GPU, every minibatch's rows sharded across the ranks.

How the path shards ("exact mode"): forces only ever update the SOURCE row, rows of
one minibatch are independent given the pre-batch matrix, and the next minibatch must see all
of them.  So the graph and the N x D matrix are replicated in every GPU's HBM (8 GiB at
RMAT-24 against 288 GB), rank r computes the r-th contiguous slice of each minibatch's rows,
and ONE exchange per minibatch -- an all-gather of the staged new rows, B x D fp32 in total --
makes every replica identical again.  Results are bit-identical to the single-GPU run for any
world size.  Every rank draws the same rand() stream on its host.

Four interchangeable exchanges, all bit-identical in result: PushExchange (the engine's own: new rows are stored
straight into the reading peers' matrices over xGMI by a HIP kernel, a device-side flag barrier separates
minibatches, the whole epoch loop runs inside libf2v -- torch.distributed only carries the IPC handles once),
NcclStageComm (an in-place RCCL all-gather on those rows of the second matrix, zero-copy on the engine's own HIP
stream), HostStageComm (the same through host memory: gloo, tests, and NcclStageComm's insurance path) and
NeedExchange (per-destination all-to-all-v: a rank receives only the rows it reads)."""
import numpy as np


def shard_bounds(lo, hi, rank, world):
    """Contiguous equal slices of minibatch [lo,hi): -> (per, my_lo, my_hi); trailing ranks may be empty."""
    per = -(-(hi - lo) // world)
    my_lo = min(lo + rank * per, hi)
    return per, my_lo, min(my_lo + per, hi)


def math_of_option(option):
    return {5: 5, 8: 5, 11: 5, 6: 6, 9: 6, 7: 7, 10: 7}[option]


def vertex_partition(n, world, batch):
    """1-D vertex partition (second mode): rank g OWNS a contiguous range of vertices and steps its own local
    minibatch k (`batch` rows) at the same time as every other rank steps ITS local minibatch k; after the step the world*batch
    new rows are exchanged.  That is, row for row, the reference's algorithm on a RELABELLED graph -- super-minibatch k is the
    union of the ranks' local minibatches k, laid out rank after rank -- with batch size world*batch and the same sample-index
    stream; and it is what ShardedTrainer does on that graph, whose equal-count slice r of every minibatch is rank r's own local
    minibatch.  -> (start uint64[world+1]: rank g owns original vertices [start[g], start[g+1]),
                    new_id int64[n]: position of original vertex v in the relabelled graph).
    All super-minibatches but the last are full (world*batch rows); the last one's rows are split like any minibatch's
    (shard_bounds), so rank sizes differ by at most one minibatch's rounding."""
    big = world * batch
    nfull = (n - 1) // big if n > 0 else 0          # full super-minibatches
    rest = n - nfull * big                          # rows of the last one (1..big)
    sizes, last = [], []
    for g in range(world):
        _, a, b = shard_bounds(0, rest, g, world)
        last.append((a, b))
        sizes.append(nfull * batch + (b - a))
    start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    new_id = np.empty(n, dtype=np.int64)
    for g in range(world):
        local = np.arange(sizes[g], dtype=np.int64)
        k, p = local // batch, local % batch
        full = k < nfull
        ids = np.where(full, k * big + g * batch + p, nfull * big + last[g][0] + (local - nfull * batch))
        new_id[start[g]: start[g + 1]] = ids
    return start, new_id


def relabel_csr(rowptr, colids, new_id):
    """The same graph with vertex v renamed new_id[v]: rows in the new order, neighbour ids ascending inside a row, duplicates kept."""
    n = len(rowptr) - 1
    deg = np.diff(rowptr.astype(np.int64))
    old_of_new = np.empty(n, dtype=np.int64)
    old_of_new[new_id] = np.arange(n, dtype=np.int64)
    rows_new = np.repeat(new_id, deg)                  # new id of every nonzero's row, in the old order
    cols_new = new_id[colids.astype(np.int64)]
    order = np.lexsort((cols_new, rows_new))
    rp = np.zeros(n + 1, dtype=np.int64)
    rp[1:] = np.cumsum(deg[old_of_new])
    return rp.astype(np.uint32), cols_new[order].astype(np.uint32)


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
    """In-place RCCL all-gather over xGMI on the new rows inside the engine's second matrix, on its HIP stream.

    Insurance for the first use on a new machine: if the very first exchange raises (an API-level refusal is the
    same on every rank), the ranks fall back -- consistently, it is decided by the exception every one of them got --
    to the host-bounce exchange over a gloo group created up-front; slow, but the run completes and says so."""

    def __init__(self, dist, rank, world, device, host_group=None):
        import torch
        self.torch, self.dist, self.rank, self.world, self.device = torch, dist, rank, world, device
        self._views = {}
        self._stream = None
        self._host_group = host_group
        self._fallback = None
        self._exchanges = 0

    def prepare(self, engine, max_batch_rows):
        per = -(-max_batch_rows // self.world)
        engine.stage_reserve(per * self.world)  # room for the padded all-gather
        if self._stream is None:
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

    def _device_exchange(self, engine, lo, hi):
        per, _, _ = shard_bounds(lo, hi, self.rank, self.world)
        d = engine.dim
        full = self._matrix_view(engine, lo)[lo * d: (lo + per * self.world) * d]
        mine = full[self.rank * per * d: (self.rank + 1) * per * d]
        with self.torch.cuda.stream(self._stream):
            self.dist.all_gather_into_tensor(full, mine)

    def exchange(self, engine, lo, hi):
        if self._fallback is not None:
            return self._fallback.exchange(engine, lo, hi)
        self._exchanges += 1
        if self._exchanges > 1 or self._host_group is None:
            return self._device_exchange(engine, lo, hi)
        try:
            self._device_exchange(engine, lo, hi)
            engine.synchronize()
            self.torch.cuda.synchronize()
        except Exception as ex:  # noqa: BLE001 -- see the class docstring
            import sys
            print("force2vec_amd.dist: RCCL exchange refused (%r); falling back to the host-bounce exchange" % (ex,), file=sys.stderr, flush=True)
            self._fallback = HostStageComm(_GroupDist(self.dist, self._host_group), self.rank, self.world)
            self._fallback.exchange(engine, lo, hi)


class _GroupDist:
    """torch.distributed restricted to one process group (what HostStageComm calls)."""

    def __init__(self, dist, group):
        self._dist, self._group = dist, group

    def all_gather(self, out, inp):
        return self._dist.all_gather(out, inp, group=self._group)


def row_owner(n, batch, world):
    """(batch index, owning rank) of every row under shard_bounds' contiguous equal slices."""
    rows = np.arange(n, dtype=np.int64)
    b = rows // batch
    lo = b * batch
    per = -(-(np.minimum(lo + batch, n) - lo) // world)
    return b, (rows - lo) // per


class NeedExchange:
    """Per-destination exchange: a rank receives only the rows it reads.

    Which rows those are is static: the CSR neighbours of the rows the rank computes, plus every vertex the
    run's (pre-drawn) sample ids name.  After each minibatch a rank packs, for every other rank, the rows of
    its slice that rank needs, and one all-to-all-v moves them; rows nobody else reads never travel (on RMAT-20
    that is 0.52 / 0.43 / 0.34 of the all-gather volume at 2 / 4 / 8 ranks).  Replicas are then complete only
    in the rows their rank reads, so `finish` all-gathers the final matrix once.  Options 5/6 only (option 7's
    neighbours are the epoch's walks).  Results are bit-identical to the all-gather exchange.

    backend "device": pack/unpack are index_select / index_copy_ on a zero-copy torch view of the engine's
    matrices, the collective is RCCL's all_to_all_single on the engine's stream; backend "host": rows travel
    through f2v_rows_read / f2v_rows_write and CPU tensors (gloo) -- tests.
    """

    def __init__(self, dist, rank, world, device=None, backend="device"):
        self.dist, self.rank, self.world, self.device, self.backend = dist, rank, world, device, backend
        if backend == "device":
            import torch
            self.torch = torch
            self._views = {}
            self._stream = None
            # a process group that cannot move device tensors (gloo) still exercises the device-side pack / unpack:
            # the collectives then go through host copies (two ranks sharing the one GPU of a test box)
            self.bounce = dist.get_backend() != "nccl"
        self.plan_key = None

    # -- setup -----------------------------------------------------------------------------------------
    def prepare(self, engine, batch, sample_ids):
        import torch
        n, w = engine.n, self.world
        key = (n, batch, w, hash(np.asarray(sample_ids).tobytes()))
        if self.backend == "device" and self._stream is None:
            self._stream = torch.cuda.ExternalStream(engine.stream(), device=self.device)
        if key == self.plan_key:
            return
        self.batch = batch
        self.nb = -(-n // batch)
        b_of, owner = row_owner(n, batch, w)
        self.owner = owner
        deg = np.diff(engine.rowptr.astype(np.int64))
        mine_edges = np.repeat(owner == self.rank, deg)
        need = np.union1d(np.unique(engine.colids[mine_edges]).astype(np.int64), np.unique(np.asarray(sample_ids, dtype=np.int64)))
        need = need[owner[need] != self.rank]
        order = np.lexsort((need, b_of[need], owner[need]))        # by source rank, then batch, then row
        need = need[order]
        recv_cnt = np.zeros((self.nb, w), dtype=np.int64)
        np.add.at(recv_cnt, (b_of[need], owner[need]), 1)
        # tell every source which of its rows this rank needs (counts per batch, then the ids)
        dev = torch.device("cuda", self.device) if (self.backend == "device" and not self.bounce) else torch.device("cpu")
        cnt_in = torch.from_numpy(np.ascontiguousarray(recv_cnt.T).reshape(-1)).to(dev)
        cnt_out = torch.empty_like(cnt_in)
        self.dist.all_to_all_single(cnt_out, cnt_in)
        send_cnt = cnt_out.cpu().numpy().reshape(w, self.nb).T.copy()             # [batch, destination]
        ids_in = torch.from_numpy(need).to(dev)
        ids_out = torch.empty(int(send_cnt.sum()), dtype=torch.int64, device=dev)
        self.dist.all_to_all_single(ids_out, ids_in, output_split_sizes=send_cnt.sum(0).tolist(), input_split_sizes=recv_cnt.sum(0).tolist())
        send_ids = ids_out.cpu().numpy()                                             # by destination, then batch, then row
        # regroup both lists per batch: batch b sends [dst 0 rows | dst 1 rows | ...], receives [src 0 | src 1 | ...]
        def per_batch(ids, cnt):
            out, start = [[] for _ in range(self.nb)], 0
            for r in range(w):
                for b in range(self.nb):
                    out[b].append(ids[start: start + cnt[b, r]])
                    start += cnt[b, r]
            return [np.concatenate(x) if x else np.zeros(0, np.int64) for x in out]
        self.send_ids, self.recv_ids = per_batch(send_ids, send_cnt), per_batch(need, recv_cnt)
        self.send_cnt, self.recv_cnt = send_cnt, recv_cnt
        assert all(np.all(owner[x] == self.rank) for x in self.send_ids)
        if self.backend == "device":
            gpu = torch.device("cuda", self.device)
            self.send_idx = [torch.from_numpy(x).to(gpu) for x in self.send_ids]
            self.recv_idx = [torch.from_numpy(x).to(gpu) for x in self.recv_ids]
        self.rows_sent_per_epoch = int(send_cnt.sum())
        self.plan_key = key

    def _matrix2d(self, engine, base_ptr, rows):
        v = self._views.get(base_ptr)
        if v is None:
            v = self.torch.as_tensor(_DevBuf(base_ptr, rows * engine.dim), device=self.torch.device("cuda", self.device)).view(rows, engine.dim)
            self._views[base_ptr] = v
        return v

    # -- per minibatch ---------------------------------------------------------------------------------
    def exchange(self, engine, lo, hi):
        import torch
        b = lo // self.batch
        s_split, r_split = self.send_cnt[b].tolist(), self.recv_cnt[b].tolist()
        if self.world == 1:
            return  # a single rank owns and reads everything
        if self.backend == "device":
            ptr, cap = engine.stage_device_ptr()
            xn = self._matrix2d(engine, ptr - lo * engine.dim * 4, cap + lo)
            with torch.cuda.stream(self._stream):
                sendbuf = xn.index_select(0, self.send_idx[b])
                if self.bounce:
                    host = torch.empty((int(sum(r_split)), engine.dim), dtype=torch.float32)
                    self.dist.all_to_all_single(host, sendbuf.cpu(), output_split_sizes=r_split, input_split_sizes=s_split)
                    recvbuf = host.to(xn.device)
                else:
                    recvbuf = torch.empty((int(sum(r_split)), engine.dim), dtype=torch.float32, device=xn.device)
                    self.dist.all_to_all_single(recvbuf, sendbuf, output_split_sizes=r_split, input_split_sizes=s_split)
                xn.index_copy_(0, self.recv_idx[b], recvbuf)
        else:
            sendbuf = torch.from_numpy(engine.rows_read(self.send_ids[b]) if len(self.send_ids[b]) else np.zeros((0, engine.dim), np.float32))
            recvbuf = torch.empty((int(sum(r_split)), engine.dim), dtype=torch.float32)
            self.dist.all_to_all_single(recvbuf, sendbuf, output_split_sizes=r_split, input_split_sizes=s_split)
            if len(self.recv_ids[b]):
                engine.rows_write(self.recv_ids[b], recvbuf.numpy())

    # -- once, after the last epoch -----------------------------------------------------------------------
    def finish(self, engine):
        """Make every replica the complete final matrix (the caller has flushed the engine)."""
        import torch
        n, d, w = engine.n, engine.dim, self.world
        if self.backend == "device":
            x = self._matrix2d(engine, engine.embeddings_device_ptr(), n + 4096).view(-1)
            with torch.cuda.stream(self._stream):
                for b in range(self.nb):
                    lo, hi = b * self.batch, min((b + 1) * self.batch, n)
                    per, _, _ = shard_bounds(lo, hi, self.rank, w)
                    full = x[lo * d: (lo + per * w) * d]
                    mine = full[self.rank * per * d: (self.rank + 1) * per * d]
                    if self.bounce:
                        host = torch.empty(per * w * d, dtype=torch.float32)
                        self.dist.all_gather_into_tensor(host, mine.cpu())
                        full.copy_(host.to(x.device))
                    else:
                        self.dist.all_gather_into_tensor(full, mine)
            engine.synchronize()
            torch.cuda.current_stream().synchronize()
        else:
            X = engine.get_embeddings()
            mine = np.flatnonzero(self.owner == self.rank)
            parts = [None] * w
            self.dist.all_gather_object(parts, (mine, X[mine]))
            for ids, rows in parts:
                X[ids] = rows
            engine.set_embeddings(X)


class PushExchange:
    """The engine's push exchange (include/f2v.h "multi-GPU: the push exchange over xGMI").

    attach(): every rank exports the HIP IPC handles of its two matrices and its flag array, the process group
    gathers them (the only use of torch.distributed on this path), every rank maps its peers and the ranks run the
    self-test together.  Raises if any rank failed -- on every rank, so that all of them fall back together."""

    def __init__(self, dist, rank, world, group=None):
        self.dist, self.rank, self.world, self.group = dist, rank, world, group
        self.attached = False

    def _all_ok(self, ok):
        flags = [None] * self.world
        self.dist.all_gather_object(flags, bool(ok), group=self.group)
        return all(flags)

    def attach(self, engine, selftest=True):
        err = None
        try:
            mine = engine.push_export()
        except Exception as ex:  # noqa: BLE001 -- reported after the ranks have agreed
            err, mine = ex, b""
        exports = [None] * self.world
        self.dist.all_gather_object(exports, mine, group=self.group)
        if all(len(e) == len(exports[0]) and len(e) > 0 for e in exports):
            try:
                engine.push_attach(self.rank, self.world, exports)
            except Exception as ex:  # noqa: BLE001
                err = ex
        elif err is None:
            err = RuntimeError("a peer could not export its IPC handles")
        ok = self._all_ok(err is None)  # also: nobody launches a barrier kernel before everybody has mapped everybody
        if ok and selftest:
            try:
                engine.push_selftest()
            except Exception as ex:  # noqa: BLE001
                err = ex
            ok = self._all_ok(err is None)
        if not ok:
            try:
                engine.push_detach()
            except Exception:  # noqa: BLE001
                pass
            raise RuntimeError("push exchange unavailable on rank %d: %r" % (self.rank, err))
        self.attached = True

    def detach(self, engine):
        if self.attached:
            engine.synchronize()
            self._all_ok(True)  # no peer unmaps while another still pushes
            engine.push_detach()
            self.attached = False


class ShardedTrainer:
    """AlgoForce2Vec* over `world` engines: same epochs, minibatches and rand() order as f2v_train."""

    def __init__(self, engine, rank, world, comm, exchange_when_single=False):
        self.engine, self.rank, self.world, self.comm = engine, rank, world, comm
        self.exchange_when_single = exchange_when_single  # self-test of the exchange path on one rank
        self.need_based = isinstance(comm, NeedExchange)

    def train(self, option, iters, batch, ns=5, lr=0.02, bs_mode=0):
        e = self.engine
        if isinstance(self.comm, PushExchange):
            if not self.comm.attached:
                self.comm.attach(e)
            return e.train_sharded(option, iters, batch, ns, lr, bs_mode)  # the whole loop runs inside libf2v
        n = e.n
        math = math_of_option(option)
        if math == 7 and bs_mode:
            raise ValueError("option 7 has no -bs 1 variant")
        nb = -(-n // batch)
        if e.get_param("hub_chunk_auto"):
            # the chunk follows the rows ONE launch covers (a slice), exactly as f2v_train_sharded picks it: every
            # exchange then gives the same bits; "hub_chunk" pins it for bits that do not depend on the world size
            e.set_param("hub_chunk_for_batch", -(-batch // self.world))
        if self.need_based and math == 7:
            raise ValueError("the per-destination exchange needs static neighbour lists: options 5/6 only")
        ndraw = ns * batch if bs_mode else ns
        stride = (min(batch, n) + ns) if bs_mode else ns

        def draw_epoch():
            # the epoch's sample ids do not depend on the embeddings (option 7 draws them after the walks, as the
            # reference's rand() order has it); kept in HBM for the whole epoch -- no sync per minibatch
            ids = np.zeros(nb * stride, dtype=np.uint32)
            for b in range(nb):
                maxv = min((b + 1) * batch, n - 1) if math == 7 else n - 1  # algorithms.cpp:1125
                if option == 9 and b < n // batch:
                    maxv = (b + 1) * batch  # option 9's own range for full minibatches (algorithms.cpp:1700-1704)
                keep = min(stride, ndraw)  # -bs 1 draws ns*BATCH ids of which rows+ns-1 are ever read
                ids[b * stride: b * stride + keep] = e.draw_samples(maxv, ndraw, keep)
            return ids

        all_ids = None
        if self.need_based:
            all_ids = [draw_epoch() for _ in range(iters)]  # who samples what decides which rows must travel
            self.comm.prepare(e, batch, np.concatenate(all_ids) if all_ids else np.zeros(0, np.uint32))
        else:
            self.comm.prepare(e, min(batch, n))
        exchanging = self.world > 1 or self.exchange_when_single
        for it in range(iters):
            if math == 7:
                e.generate_walks()  # same stream on every rank: sample/algorithms.cpp:1097-1118
            e.upload_sample_ids(all_ids[it] if all_ids is not None else draw_epoch())
            for b in range(nb):
                lo, hi = b * batch, min((b + 1) * batch, n)
                _, my_lo, my_hi = shard_bounds(lo, hi, self.rank, self.world)
                e.minibatch_step_at(option, lo, hi, b * stride, ns, lr, bs_mode, row_lo=my_lo, row_hi=my_hi)
                if exchanging:
                    self.comm.exchange(e, lo, hi)
        e.flush()
        if self.need_based and exchanging:
            self.comm.finish(e)

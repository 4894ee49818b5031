#!/usr/bin/env python3
"""bench.py -- edges/sec of the Force2Vec hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one epoch = one pass of the hot path over every minibatch of the synthetic
graph (BASELINE.json configs[2]: RMAT scale-20, ~1 M vertices / ~16 M undirected edges,
option 5, D = 128, ns = 5).  edges/sec = nnz * K / T (nnz = directed CSR nonzeros, the
reference's unit).  Inputs are resident in HBM when the timed region starts.
N > 1 (launched by torch.distributed.run, one rank per GPU): the SAME graph and epochs, each
minibatch's rows sharded over the ranks -- strong scaling.  Default exchange: the engine's push over
xGMI (f2v_train_sharded: new rows stored straight into the reading peers' HBM, device-side flag
barrier; torch.distributed only carries the IPC handles); if it cannot be set up on this machine the
run falls back to an RCCL all-gather per minibatch and says so.  After the timed region every rank
checks its replica bit for bit against a single-GPU run of the same epochs.

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` (HBM bound,
algorithmic bytes / launch over the live HIP-event launch time) and `cpu_baseline` (the genuine
reference binary oracle/_ref timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


T0 = time.time()


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def note(rank, msg):
    """Progress on stderr (every rank: a stuck rank shows where it stopped)."""
    if os.environ.get("F2V_BENCH_QUIET") != "1":
        print("bench[rank %d +%.1fs]: %s" % (rank, time.time() - T0, msg), file=sys.stderr, flush=True)


def load_graph(scale, edge_factor, seed):
    """RMAT CSR, cached under /tmp (generation is setup, not measured)."""
    from force2vec_amd.graph import rmat_csr
    cache = "/tmp/f2v_rmat_s%d_e%d_seed%d.npz" % (scale, edge_factor, seed)
    if os.path.exists(cache):
        try:
            z = np.load(cache)
            return z["rowptr"], z["colids"]
        except Exception:
            pass
    t0 = time.time()
    rowptr, colids = rmat_csr(scale, edge_factor, seed)
    log("bench: generated RMAT scale %d: n=%d nnz=%d in %.1fs" % (scale, len(rowptr) - 1, len(colids), time.time() - t0))
    try:
        tmp = cache + ".%d.tmp.npz" % os.getpid()
        np.savez(tmp, rowptr=rowptr, colids=colids)
        os.replace(tmp, cache)
    except Exception:
        pass
    return rowptr, colids


def cpu_baseline(args):
    """The reference on this box's host cores, on a bounded sample of the same workload: a smaller RMAT
    graph (scale args.cpu_scale), same D / ns / lr / batch.  T(iters=k) - T(iters=0) strips the reference's
    in-timer initialisation.  kind 'reference' = oracle/_ref (the genuine reference, built from
    /root/reference by oracle/build_ref.sh); fallback kind 'port' = the single-thread C oracle."""
    from force2vec_amd.graph import edges_from_csr, write_mtx_symmetric
    from oracle import oracle as O
    rowptr, colids = load_graph(args.cpu_scale, 16, 1)
    n, nnz = len(rowptr) - 1, len(colids)
    cores = os.cpu_count() or 1
    flags = open("/proc/cpuinfo").read()
    avx512 = (" avx512f" in flags) and (" avx512dq" in flags) and O.ref_binary(True) is not None
    exe_ok = O.ref_binary(avx512) is not None
    sample = "RMAT scale-%d (n=%d, nnz=%d), D=%d ns=5 lr=0.02 batch=%d" % (args.cpu_scale, n, nnz, args.dim, args.batch)
    if exe_ok:
        option = 11 if avx512 else 5
        with tempfile.TemporaryDirectory() as td:
            mtx = os.path.join(td, "sample.mtx")
            src, dst = edges_from_csr(rowptr, colids)
            write_mtx_symmetric(mtx, n, src, dst)

            def run(iters, threads):
                out = run_ref(O, mtx, td, option, iters, args, threads, avx512)
                for line in out.splitlines():
                    if "Wall time required" in line:
                        return float(line.split(":")[-1].split()[0])
                raise RuntimeError("reference output not understood:\n" + out)

            # the reference does not scale to every core count on every box: a short scan picks the
            # thread count that serves it best, the long run below is what is reported
            cands = sorted({c for c in (16, 32, 48, 64, 128, cores) if c <= cores})
            best, best_rate = cores, 0.0
            t0 = run(0, 1)  # the in-timer initialisation is serial (N*D rand() calls): measured once
            for th in cands:
                rate = 3.0 / max(run(3, th) - t0, 1e-6)
                if rate > best_rate:
                    best, best_rate = th, rate
            threads = best
            k = int(min(2000, max(args.cpu_iters, 12.0 * best_rate)))  # about 12 s of CPU work
            tk = run(k, threads)
        val = nnz * k / max(tk - t0, 1e-9)
        return {"value": val, "unit": "edges/s", "cores": threads, "kind": "reference",
                "sample": "%s; oracle/_ref option %d%s, %d epochs in %.2fs, -threads %d (best of %s on a %d-core host)"
                          % (sample, option, " (AVX512 build)" if avx512 else " (scalar build)", k, tk - t0, threads, cands, cores)}
    # port: single-thread oracle, a few epochs
    X = O.Rng(1).init_embeddings(n, args.dim, 0)
    t0 = time.time()
    O.train(5, rowptr, colids, args.dim, 1, args.batch, X0=X)
    dt = time.time() - t0
    return {"value": nnz / dt, "unit": "edges/s", "cores": 1, "kind": "port", "sample": sample + "; oracle/f2v_oracle.c, 1 epoch in %.2fs" % dt}


def kernel_name(args, pushing):
    """The step kernel a run of this shape launches (force2vec_amd/csrc/f2v_engine.hip launch_step)."""
    if args.dim % 4 or args.dim > 256:
        return "f2v::step_kernel"
    w = 16
    while w < args.dim:
        w <<= 1
    return "f2v::qstep_kernel<%d, %d, %d, %d, %s, %s>" % (5 if args.option in (5, 8, 11) else 6, min(16, w // 4), max(1, w // 64), 4 if w >= 128 else 8,
                                                          "true" if pushing else "false", "true" if w == args.dim else "false")


def run_ref(O, mtx, td, option, iters, args, cores, avx512):
    exe = O.ref_binary(avx512)
    cmd = [exe, "-input", mtx, "-output", "/nonexistent_dir_so_no_embd_is_written/", "-iter", str(iters), "-batch", str(args.batch),
           "-dim", str(args.dim), "-nsamples", "5", "-lr", "0.02", "-option", str(option), "-threads", str(cores)]
    out = subprocess.run(cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--scale", type=int, default=20, help="RMAT scale (2^scale vertices, edge factor 16)")
    ap.add_argument("--dim", type=int, default=128)
    ap.add_argument("--batch", type=int, default=65536)
    ap.add_argument("--option", type=int, default=5)
    ap.add_argument("--hub-chunk", type=int, default=-1)
    ap.add_argument("--param", action="append", default=[], help="engine tunable name=value (repeatable)")
    ap.add_argument("--cpu-scale", type=int, default=17)
    ap.add_argument("--cpu-iters", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--exchange", choices=["push", "allgather", "need"], default="push",
                    help="N>1: push = the engine's own xGMI exchange (rows stored into the reading peers' HBM by a HIP kernel); "
                         "allgather = RCCL all-gather of every minibatch's new rows; need = RCCL all-to-all-v of only the rows each rank reads")
    ap.add_argument("--push-fused", choices=["auto", "1", "0"], default="auto",
                    help="push exchange: rows pushed by the step kernels themselves (1), by a kernel behind them (0), or whichever "
                         "is faster on this machine over a few untimed epochs (auto)")
    ap.add_argument("--settle-ms", type=float, default=40.0,
                    help="before the W warmup steps: epochs of the same workload until this much time has passed -- after any idle "
                         "gap the GPU's power controller needs ~15 ms of steady load before epoch times stop moving "
                         "(profiles/r01_idle_effect.txt); 0 = none")
    ap.add_argument("--no-preflight", action="store_true", help="N>1: skip the child process that rehearses IPC mapping and remote stores first")
    ap.add_argument("--no-verify", action="store_true", help="N>1: skip the bit-for-bit check of every replica against a single-GPU run")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: self-test of the N>1 plumbing on a one-GPU box (all ranks on device 0, exchange through the host)")
    ap.add_argument("--force-dist", action="store_true", help="drive even a single rank through the multi-GPU path (RCCL group of 1): self-test")
    ap.add_argument("--extra-batches", type=str, default="256,4096,16384,262144", help="N=1: comma list of further batch sizes to time (reported under 'extra')")
    ap.add_argument("--dist-extra-batches", type=str, default="262144", help="N>1: the same for the sharded run (fewer, larger exchanges)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run --nproc-per-node %d" % (args.gpus, args.gpus))
        args.gpus = world

    import force2vec_amd as F
    from force2vec_amd import dist as fdist

    dist = None
    torch = None
    host_group = None
    use_dist = world > 1 or args.force_dist
    preflight_ok = True
    if world > 1 and args.exchange == "push" and not args.no_preflight:
        # Before this process touches the GPU: a throw-away child rehearses the IPC mapping and the remote stores the push
        # exchange needs (the ranks' children meet through files).  If it fails, faults or never returns, the run uses RCCL.
        mbytes = ((1 << args.scale) + 4096) * args.dim * 4
        if mbytes >= 0x7FF00000:
            mbytes = 2 * min(512 << 20, mbytes // 2)  # such engines map a landing buffer instead of the matrices
        pre_dev = 0 if args.dist_backend == "gloo" else local_rank
        # where the ranks' children meet: all ranks are children of one launcher (torch.distributed.run), whose pid makes
        # the name unique to this launch; with another launcher the children time out and the run falls back
        meet = "/tmp/f2v_preflight_%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid())
        try:
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ipc_preflight.py"), str(pre_dev), str(rank), str(world), meet,
                                str(mbytes), "60"], timeout=150, stdout=subprocess.DEVNULL)
            preflight_ok = r.returncode == 0
        except subprocess.TimeoutExpired:
            preflight_ok = False
        note(rank, "IPC preflight (%d MiB buffers between %d processes): %s" % (mbytes >> 20, world, "passed" if preflight_ok else "FAILED"))
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        import torch
        import torch.distributed as dist
        if args.dist_backend == "gloo":
            local_rank = 0
            dist.init_process_group("gloo")
        else:
            if torch.cuda.device_count() <= local_rank:
                local_rank = 0  # the launcher gave every rank its own single visible device
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            try:  # handle exchange of the push path, insurance path of NcclStageComm
                host_group = dist.new_group(backend="gloo") if world > 1 else None
            except Exception as exn:  # the RCCL group carries the (pickled) handles just as well
                log("bench[rank %d]: no gloo group (%r); using the RCCL group for the handle exchange" % (rank, exn))
                host_group = None

    if use_dist and world > 1 and args.exchange == "push" and not args.no_preflight:
        oks = [None] * world
        dist.all_gather_object(oks, bool(preflight_ok), group=host_group)
        if rank == 0:  # every rank's child has ended
            shutil.rmtree(meet, ignore_errors=True)
        if not all(oks):
            state_note = "IPC preflight failed on rank(s) %s: RCCL all-gather instead of the push exchange" % ([i for i, o in enumerate(oks) if not o],)
            log("bench[rank %d]: %s" % (rank, state_note))
            args.exchange = "allgather"
            args.preflight_note = state_note
    if use_dist and world > 1:
        # one rank generates (or finds) the cached graph, the others read the cache
        if rank == 0:
            load_graph(args.scale, 16, 1)
        dist.barrier(group=host_group) if host_group is not None else dist.barrier()
    rowptr, colids = load_graph(args.scale, 16, 1)
    n, nnz = len(rowptr) - 1, len(colids)
    note(rank, "graph ready: n=%d nnz=%d" % (n, nnz))
    eng = F.Engine(rowptr, colids, args.dim, device=local_rank)
    if args.hub_chunk >= 0:
        eng.set_param("hub_chunk", args.hub_chunk)
    for kv in args.param:
        k, v = kv.split("=")
        eng.set_param(k, int(v))
    eng.srand(1)
    eng.init_embeddings(F._lib.INIT_SYMMETRIC if args.option in (5, 8, 11) else F._lib.INIT_UNIT)
    note(rank, "engine ready, embeddings initialised")

    def barrier():
        eng.synchronize()
        if use_dist:
            if args.dist_backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()
            if args.dist_backend == "nccl":
                torch.cuda.synchronize()

    state = {"exchange": args.exchange, "comm": None, "note": getattr(args, "preflight_note", None)}

    def make_comm():
        """The exchange this run uses; the push exchange is attached (and self-tested) here, once."""
        if state["comm"] is not None:
            return state["comm"]
        ex = state["exchange"]
        if ex == "push":
            comm = fdist.PushExchange(dist, rank, world, group=host_group)
            try:
                comm.attach(eng)
                note(rank, "push exchange attached, self-test passed")
            except Exception as exn:  # raised on every rank alike (the ranks agree inside attach)
                state["note"] = "push exchange unavailable (%s): fell back to the RCCL all-gather" % (str(exn)[:200],)
                log("bench[rank %d]: %s" % (rank, state["note"]))
                state["exchange"] = ex = "allgather"
        if ex == "allgather":
            comm = (fdist.HostStageComm(dist, rank, world) if args.dist_backend == "gloo"
                    else fdist.NcclStageComm(dist, rank, world, local_rank, host_group=host_group))
        elif ex == "need":
            comm = (fdist.NeedExchange(dist, rank, world, backend="host") if args.dist_backend == "gloo"
                    else fdist.NeedExchange(dist, rank, world, device=local_rank, backend="device"))
        state["comm"] = comm
        return comm

    schedule = []  # every (epochs, batch) trained so far: the verification replays it on one engine

    def run_epochs(k, batch):
        """-> per-rank statistics of these k epochs (launches, algorithmic bytes, device seconds where known)."""
        schedule.append((k, batch))
        if not use_dist:
            eng.train(args.option, k, batch, 5, 0.02, 0)
            return eng.stats()
        s0 = eng.stats()
        comm = make_comm()
        fdist.ShardedTrainer(eng, rank, world, comm, exchange_when_single=True).train(args.option, k, batch, 5, 0.02, 0)
        s1 = eng.stats()
        if state["exchange"] == "push":
            st = dict(s1)  # f2v_train_sharded restarts the statistics like f2v_train
        else:
            st = {key: s1[key] - s0[key] for key in s1}
        st["device_seconds"] = None  # the sharded loop is timed by the wall clock below
        return st

    def settle(batch):
        """Same workload, untimed, until the device has been under steady load for --settle-ms."""
        if args.settle_ms <= 0:
            return 0
        run_epochs(2, batch)  # launch plans are built here, on the host: not load yet
        eng.synchronize()
        done, t0 = 2, time.perf_counter()
        while max_over_ranks((time.perf_counter() - t0) * 1e3) < args.settle_ms and done < 1000:
            run_epochs(4, batch)
            eng.synchronize()
            done += 4
        return done

    def timed(k, w, batch):
        state["settle_epochs"] = settle(batch)
        if w > 0:
            run_epochs(w, batch)
        barrier()
        t0 = time.perf_counter()
        st = run_epochs(k, batch)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        return dt, st

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def tune_push():
        """Untimed: which of the two push variants is faster here?  Every rank sees the same (max-over-ranks) times."""
        if args.push_fused != "auto":
            eng.set_param("push_fused", int(args.push_fused))
            return
        run_epochs(1, args.batch)  # launch plans and reader masks exist from here on
        note(rank, "first sharded epoch done")
        settle(args.batch)         # both variants are timed on a GPU that has left the power-management transient behind
        took = {1: float("inf"), 0: float("inf")}
        for fused in (1, 0, 1, 0):  # alternating, best of two each
            eng.set_param("push_fused", fused)
            barrier()
            t0 = time.perf_counter()
            run_epochs(5, args.batch)
            barrier()
            took[fused] = min(took[fused], max_over_ranks(time.perf_counter() - t0) / 5)
        best = 1 if took[1] <= took[0] else 0
        eng.set_param("push_fused", best)
        note(rank, "push variants timed: fused %.3f ms, separate kernel %.3f ms per epoch" % (took[1] * 1e3, took[0] * 1e3))
        state["tuned"] = {"push_fused": best, "ms_per_epoch_fused": took[1] * 1e3, "ms_per_epoch_separate_kernel": took[0] * 1e3}

    def verify(schedule):
        """Every rank: the same epochs on ONE engine (no sharding, no exchange) must give this replica bit for bit."""
        ref = F.Engine(rowptr, colids, args.dim, device=local_rank)
        for kv in args.param:
            pk, pv = kv.split("=")
            if not pk.startswith("push_"):
                ref.set_param(pk, int(pv))
        ref.set_param("hub_chunk", eng.get_param("hub_chunk"))  # the chunk is part of the summation order
        ref.srand(1)
        ref.init_embeddings(F._lib.INIT_SYMMETRIC if args.option in (5, 8, 11) else F._lib.INIT_UNIT)
        for i, (k, b) in enumerate(schedule):
            try:
                ref.train(args.option, k, b, 5, 0.02, 0)
            except Exception:
                note(rank, "single-GPU replay failed in call %d of %d (%d epochs at batch %d)" % (i, len(schedule), k, b))
                raise
        same = bool(np.array_equal(ref.get_embeddings(), eng.get_embeddings()))
        ref.close()
        t = torch.tensor([1 if same else 0], dtype=torch.int32, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    if use_dist and world > 1 and state["exchange"] == "push":
        make_comm()
        if state["exchange"] == "push":
            tune_push()
    dt, st = timed(args.steps, args.warmup, args.batch)
    main_settle = state["settle_epochs"]
    note(rank, "timed region done: %.3f ms per epoch" % (dt / args.steps * 1e3))
    verified = None
    if use_dist and not args.no_verify:
        verified = verify(schedule)
        note(rank, "replica compared with a single-GPU run of the same %d epochs: %s" % (sum(k for k, _ in schedule), "identical" if verified else "DIFFERENT"))
        if not verified and state["exchange"] == "push":
            # never report a number for wrong results: redo the whole measurement over the RCCL all-gather
            state["note"] = "push exchange gave a replica that differs from the single-GPU run: measured again over the RCCL all-gather"
            log("bench[rank %d]: %s" % (rank, state["note"]))
            state["comm"].detach(eng)
            state["comm"], state["exchange"] = None, "allgather"
            eng.srand(1)
            eng.init_embeddings(F._lib.INIT_SYMMETRIC if args.option in (5, 8, 11) else F._lib.INIT_UNIT)
            del schedule[:]
            dt, st = timed(args.steps, args.warmup, args.batch)
            verified = verify(schedule)
    # rForce2Vec attracts along 5 walk samples per vertex, not along the CSR's nonzeros
    units = 5 * n if args.option in (7, 10) else nnz
    value = units * args.steps / dt
    res = {
        "metric": "embedding edges/sec at D=%d, option %d" % (args.dim, args.option),
        "value": value, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": "RMAT scale-%d edge-factor 16 (n=%d, nnz=%d directed CSR nonzeros), option %d, D=%d, ns=5, lr=0.02, batch=%d; step = 1 epoch"
                   % (args.scale, n, nnz, args.option, args.dim, args.batch),
                   "batch": args.batch, "hub_chunk": eng.get_param("hub_chunk"),
                   "settle": "%d untimed epochs of the same workload (>= %g ms) before the %d warmup steps" % (main_settle, args.settle_ms, args.warmup),
                   "parallelism": "1 GPU" if not use_dist else "minibatch rows sharded over %d GPUs, replicated graph+matrix, %s" % (world, {
                       "push": "new rows pushed over xGMI into the HBM of the peers that read them (HIP kernel + device-side flag barrier)",
                       "allgather": "%s all-gather of the new rows" % ("RCCL" if args.dist_backend == "nccl" else "gloo (host-bounce self-test)"),
                       "need": "%s all-to-all-v of the rows each rank reads" % ("RCCL" if args.dist_backend == "nccl" else "gloo (host-bounce self-test)")}[state["exchange"]])},
    }
    if use_dist:
        res["config"]["exchange"] = state["exchange"]
        res["config"]["replicas_bit_identical_to_1gpu_run"] = verified
        if state["note"]:
            res["config"]["note"] = state["note"]
        if state.get("tuned"):
            res["config"]["push_autotune"] = state["tuned"]
        if state["exchange"] == "push":
            res["config"]["push_fused"] = eng.get_param("push_fused")
            ps = eng.push_stats()
            res["config"]["rows_pushed_over_allgather_rows"] = ps["rows_pushed"] / max(ps["rows_allgather"], 1)
    if st is not None:
        # dominant kernel = qstep_kernel, one launch per minibatch; launch time from HIP events on the
        # engine's stream around the epoch loop (f2v_train), algorithmic bytes per SURVEY 8d.
        # N > 1: rank 0's share of the bytes over the wall time of the sharded loop (exchange included).
        per_launch = st["algorithmic_bytes"] / max(st["step_launches"], 1)
        t_launch = (st["device_seconds"] if st["device_seconds"] else dt) / max(st["step_launches"], 1)
        ach = per_launch / t_launch * 1e-9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                if not use_dist and tj.get("batch") == args.batch and tj.get("scale") == args.scale and tj.get("dim") == args.dim:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                pass
        res["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                           "traffic": traffic, "kernel": kernel_name(args, use_dist and world > 1 and state["exchange"] == "push" and eng.get_param("push_fused")), "algorithmic_bytes_per_launch": per_launch,
                           "avg_launch_us": t_launch * 1e6, "launches": st["step_launches"],
                           # frac counts ALGORITHMIC bytes, so XCD-L2 hits on hub rows can push it past 1; what actually crossed
                           # the fabric (PMC) over the same time is the honest HBM utilisation
                           "hbm_frac_from_traffic": (traffic / t_launch * 1e-9 / HBM_PEAK_GBS) if traffic else None}
        if not use_dist:
            import ctypes
            g = ctypes.c_double()
            if F._lib.lib().f2v_diag_stream_copy(local_rank, 1 << 30, 5, ctypes.byref(g)) == 0:
                res["roofline"]["stream_copy_GBs_on_this_box"] = g.value  # 1-GiB copy kernel, read + written bytes
    if rank == 0 and not use_dist:
        extra = {}
        for b in [int(x) for x in args.extra_batches.split(",") if x]:
            dtb, stb = timed(max(1, args.steps // 2), 1, b)
            k = max(1, args.steps // 2)
            extra["batch_%d" % b] = {"edges_per_s": units * k / dtb, "ms_per_epoch": dtb / k * 1e3,
                                     "hbm_GBs": stb["algorithmic_bytes"] / stb["device_seconds"] * 1e-9}
        if extra:
            res["extra"] = extra
    if use_dist and world > 1:
        extra = {}
        for b in [int(x) for x in args.dist_extra_batches.split(",") if x]:
            k = max(1, args.steps // 2)
            dtb, _ = timed(k, 1, b)
            extra["batch_%d" % b] = {"edges_per_s": units * k / dtb, "ms_per_epoch": dtb / k * 1e3}
        if extra:
            res["extra"] = extra
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(args)
        except Exception as ex:  # the baseline is reported, never required for the GPU number
            res["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": os.cpu_count(), "kind": "reference", "sample": "failed: %r" % (ex,)}
    if use_dist and state["exchange"] == "push" and state["comm"] is not None:
        state["comm"].detach(eng)
    eng.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()

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

Prints ONE JSON line on rank 0 (contract in the task statement), with `roofline` and `cpu_baseline` (the genuine
reference binary oracle/_ref timed on this box's host cores on a bounded sample of the same workload).

roofline, bound "hbm": `achieved` = COMPULSORY bytes per launch of the dominant kernel (distinct embedding rows read +
rows written + neighbour ids + work items: what must cross HBM even if every re-read inside the launch hit a cache,
counted on the host by the engine) / that kernel's average launch time (HIP events on the engine's stream around the
epoch loop); `frac` = achieved / 8 TB/s and cannot exceed 1.  The SURVEY 8d figure (every neighbour row charged to HBM,
"no credit for cache reuse") is kept as `algorithmic_GBs`: it is NOT a fraction of anything -- on a power-law graph hub
rows are cache hits and it passes the HBM peak.  `traffic` = bytes that left the XCDs' L2s per launch (PMC passes of
tools/profile_traffic.sh, committed under profiles/): HBM + Infinity Cache together; rocprofv3 exposes no counter on
the memory side of the Infinity Cache on this part, so HBM bytes lie between `compulsory` ... `traffic`.
After the timed region (N = 1) one more minibatch is checked on sampled rows against the CPU oracle (`config.verified_rows`).
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


def write_mtx_fast(path, n, src, dst):
    """`pattern symmetric` MatrixMarket file of a 10^7-edge list in seconds (pyarrow's CSV writer; np.savetxt takes a minute)."""
    try:
        import pyarrow as pa
        import pyarrow.csv as pacsv
        with open(path, "wb") as f:
            f.write(b"%%MatrixMarket matrix coordinate pattern symmetric\n")
            f.write(b"%d %d %d\n" % (n, n, len(src)))
            tbl = pa.table({"r": pa.array(src + 1), "c": pa.array(dst + 1)})
            pacsv.write_csv(tbl, f, pacsv.WriteOptions(include_header=False, delimiter=" "))
    except ImportError:
        from force2vec_amd.graph import write_mtx_symmetric
        write_mtx_symmetric(path, n, src, dst)


def cpu_baseline(args, opt7=None):
    """The reference on this box's host cores, on a bounded sample of the same workload: the north-star's 10 M-edge
    power-law graph (RMAT scale args.cpu_scale = 20 by default: the benchmark's own graph, 15.7 M undirected edges), same
    D / ns / lr / batch, a few epochs.  T(iters=k) - T(iters=0) strips the reference's in-timer initialisation and leaves
    its file parsing out.  kind 'reference' = oracle/_ref (the genuine reference, built from /root/reference by
    oracle/build_ref.sh); fallback kind 'port' = the single-thread C oracle."""
    from force2vec_amd.graph import edges_from_csr
    from oracle import oracle as O
    cores = os.cpu_count() or 1
    flags = open("/proc/cpuinfo").read()
    avx512 = (" avx512f" in flags) and (" avx512dq" in flags) and O.ref_binary(True) is not None
    exe_ok = O.ref_binary(avx512) is not None
    rowptr, colids = load_graph(args.cpu_scale, 16, 1)
    n, nnz = len(rowptr) - 1, len(colids)
    sample = "RMAT scale-%d (n=%d, nnz=%d = %.1f M undirected edges), D=%d ns=5 lr=0.02 batch=%d" % (args.cpu_scale, n, nnz, nnz / 2e6, args.dim, args.batch)
    if exe_ok:
        option = 11 if avx512 else 5
        with tempfile.TemporaryDirectory() as td:
            def run(mtx, iters, threads):
                out = run_ref(O, mtx, td, option, iters, args, threads, avx512)
                for line in out.splitlines():
                    if "Wall time required" in line:
                        return float(line.split(":")[-1].split()[0])
                raise RuntimeError("reference output not understood:\n" + out)

            # the reference does not scale to every core count on every box (its best was 32 threads on one 256-core host of the
            # pool, 128 on another): a scan of three epochs per candidate ON THIS GRAPH picks the thread count, the long run
            # below is what is reported
            mtx = os.path.join(td, "sample.mtx")
            src, dst = edges_from_csr(rowptr, colids)
            write_mtx_fast(mtx, n, src, dst)
            del src, dst
            cands = sorted({c for c in (32, 48, 64, 128) if c <= cores} or {cores})  # (48: the thread count the north star quotes)
            t0 = run(mtx, 0, 1)  # the in-timer initialisation is serial (N*D rand() calls): measured once
            best, best_rate = cands[0], 0.0
            for th in cands:
                rate = 3.0 / max(run(mtx, 3, th) - t0, 1e-6)
                if rate > best_rate:
                    best, best_rate = th, rate
            threads = best
            k = max(2, args.cpu_iters)
            tk = run(mtx, k, threads)
            if opt7 is not None:
                # the reference's own option 7 (rForce2Vec; option 10 is its AVX512 twin) on the same graph: T(6) - T(0), 5 walk samples per vertex and epoch
                try:
                    o7 = 10 if avx512 else 7

                    def run7(iters, th):
                        out = run_ref(O, mtx, td, o7, iters, args, th, avx512)
                        return float([l for l in out.splitlines() if "Wall time required" in l][0].split(":")[-1].split()[0])
                    t70 = run7(0, 1)
                    t7 = run7(6, threads) - t70
                    opt7.update({"reference_cpu_option": o7, "reference_cpu_threads": threads, "reference_cpu_ms_per_epoch": t7 / 6 * 1e3,
                                 "reference_cpu_pairs_per_s": 5.0 * n * 6 / max(t7, 1e-9)})
                except Exception as ex:  # noqa: BLE001
                    opt7["reference_cpu_ms_per_epoch"] = "failed: %r" % (ex,)
        val = nnz * k / max(tk - t0, 1e-9)
        return {"value": val, "unit": "edges/s", "cores": threads, "kind": "reference",
                "sample": "%s; oracle/_ref option %d%s, %d epochs in %.2fs, -threads %d (best of %s in a 3-epoch scan on the same graph; %d-core host)"
                          % (sample, option, " (AVX512 build)" if avx512 else " (scalar build)", k, tk - t0, threads, cands, cores)}
    # port: single-thread oracle, one epoch of the small graph
    rowptr, colids = load_graph(min(17, args.cpu_scale), 16, 1)
    n, nnz = len(rowptr) - 1, len(colids)
    X = O.Rng(1).init_embeddings(n, args.dim, 0)
    t0 = time.time()
    O.train(5, rowptr, colids, args.dim, 1, args.batch, X0=X)
    dt = time.time() - t0
    return {"value": nnz / dt, "unit": "edges/s", "cores": 1, "kind": "port",
            "sample": "RMAT scale-17 (n=%d, nnz=%d), D=%d; oracle/f2v_oracle.c (single thread), 1 epoch in %.2fs" % (n, nnz, args.dim, dt)}


def verify_rows(F, eng, rowptr, colids, args, n_rows=32):
    """After the timed region: one more minibatch, checked on sampled rows (the batch's largest hubs and zero-degree rows
    included) against the CPU oracle's row function applied to the downloaded pre-step matrix, bit for bit; rows outside the
    minibatch must not change.  -> number of rows verified (raises on a mismatch)."""
    from oracle import oracle as O
    n = len(rowptr) - 1
    math = {5: 5, 8: 5, 11: 5, 6: 6, 9: 6}.get(args.option)
    if math is None:
        return None  # option 7's walks are per epoch: covered by the tests
    deg = np.diff(rowptr.astype(np.int64))
    before = eng.get_embeddings()
    rng = np.random.default_rng(2)
    lo = (n // 2 // args.batch) * args.batch
    hi = min(lo + args.batch, n)
    ids = rng.integers(0, n - 1, 5).astype(np.uint32)
    eng.minibatch_step(args.option, lo, hi, ids, 5, 0.02)
    after = eng.get_embeddings()
    chunk = eng.get_param("hub_chunk")
    rows = np.concatenate([rng.integers(lo, hi, n_rows - 8), lo + np.argsort(deg[lo:hi])[-4:], lo + np.flatnonzero(deg[lo:hi] == 0)[:4]])
    for i in rows:
        want = O.row(math, rowptr, colids, before, int(i), ids, 0.02, order=O.ORDER_TREE, chunk=chunk)
        if not np.array_equal(after[i], want):
            raise SystemExit("bench: row %d (degree %d) of the verification minibatch differs from the oracle by %g"
                             % (i, deg[i], float(np.abs(after[i] - want).max())))
    if not (np.array_equal(after[:lo], before[:lo]) and np.array_equal(after[hi:], before[hi:])):
        raise SystemExit("bench: the verification minibatch changed rows outside [%d,%d)" % (lo, hi))
    return int(len(rows))


def kernel_name(args, pushing, form, wide_width=0, wide_early=0):
    """The step kernel the last f2v_train of this shape launched: `form` is the engine's own answer ("last_train_form": 0 one
    launch per minibatch, 1 chained, 2 chained in the wide form) -- the launch rules live in f2v_engine.hip, not here."""
    if args.dim % 4 or args.dim > 256:
        return "f2v::step_kernel"
    w = 16
    while w < args.dim:
        w <<= 1
    if form == 2 and wide_width:
        w = max(w, wide_width)  # the wide form may run narrow rows on a wider layout ("wide_min_width")
    opt, lpi, nb, u, full = 5 if args.option in (5, 8, 11) else 6, min(16, w // 4), max(1, w // 64), 4 if w >= 128 else 8, "true" if w == args.dim else "false"
    if form == 2 and not pushing:
        return "f2v::qwide_chain_kernel<%d, %d, %d, %d, %s, %d>" % (opt, lpi, nb, u, full, wide_early)  # MODE: 0 plain, 1 EARLY, 2 EARLY + epochs chained
    if form == 1 and not pushing:
        return "f2v::qstep_chain_kernel<%d, %d, %d, %d, %s>" % (opt, lpi, nb, u, full)
    return "f2v::qstep_kernel<%d, %d, %d, %d, %s, %s>" % (opt, lpi, nb, u, "true" if pushing else "false", full)


def live_traffic(args, kname, budget_s):
    """`roofline.traffic` measured BY THIS RUN: separate `rocprofv3 --pmc` passes (FETCH_SIZE; WRITE_SIZE; TCC_HIT / TCC_MISS -- never
    combined with a trace domain other than --kernel-trace) over a short child run of this very script on the same workload, read per
    launch of the dominant kernel.  Corrections as MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE (KiB) x 2 for 16-byte-per-lane
    reads (128-B requests are tallied at 64 B), WRITE_SIZE (KiB) exact.  -> dict, or a string saying why it was not collected."""
    import glob
    if shutil.which("rocprofv3") is None:
        return "rocprofv3 is not on PATH"
    if any(k in os.environ for k in ("ROCPROFILER_LIBRARY_CTOR", "ROCP_TOOL_LIBRARIES", "ROCPROF_OUTPUT_PATH")) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return "this run is itself under a profiler"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from parse_pmc import counters
    child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--settle-ms", "0", "--extra-batches=", "--config5-scale", "0", "--config4", "0",
             "--cora", "0", "--sustained-s", "0", "--verify-rows", "0", "--no-cpu-baseline", "--live-pmc", "0", "--option7", "0", "--no-ceilings",
             "--batch", str(args.batch), "--scale", str(args.scale), "--dim", str(args.dim), "--option", str(args.option), "--hub-chunk", str(args.hub_chunk)]
    for kv in args.param:
        child += ["--param", kv]
    short = kname.split("::")[-1].split("<")[0]
    out, t_start = {}, time.time()
    td = tempfile.mkdtemp(prefix="f2v_pmc_")
    try:
        for name, ctrs in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("l2", ["TCC_HIT_sum", "TCC_MISS_sum"])):
            left = budget_s - (time.time() - t_start)
            if left < 20:
                out["note"] = "stopped before the %s pass: the time budget of %d s was used up" % (name, budget_s)
                break
            d = os.path.join(td, name)
            env = dict(os.environ, TMPDIR="/tmp", F2V_BENCH_QUIET="1")
            try:
                r = subprocess.run(["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                                   stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=left)
            except subprocess.TimeoutExpired:
                out["note"] = "the %s pass did not end within the time budget" % name
                break
            if r.returncode != 0:
                out["note"] = "the %s pass ended with code %d: %s" % (name, r.returncode, r.stderr[-300:])
                break
            c, nd = counters(d, kname.split("::")[-1], skip_first=0)  # (the name as the profiler prints it: template arguments included)
            if not nd:
                c, nd = counters(d, short, skip_first=0)
            # the child's launches of the dominant kernel at THIS batch size only (its verification / other kernels are switched off)
            out.update(c)
            out["launches_" + name] = nd
        if "FETCH_SIZE" in out and "WRITE_SIZE" in out:
            res = {"traffic": out["FETCH_SIZE"] * 1024.0 * 2.0 + out["WRITE_SIZE"] * 1024.0, "fetch_bytes": out["FETCH_SIZE"] * 1024.0 * 2.0, "write_bytes": out["WRITE_SIZE"] * 1024.0,
                   "launches_sampled": out.get("launches_fetch"), "seconds": time.time() - t_start}
            if out.get("TCC_MISS_sum"):
                res["l2_hit_rate"] = out["TCC_HIT_sum"] / (out["TCC_HIT_sum"] + out["TCC_MISS_sum"])
                res["l2_request_bytes"] = (out["TCC_HIT_sum"] + out["TCC_MISS_sum"]) * 128.0  # 128-byte requests, hits and misses
            if "note" in out:
                res["note"] = out["note"]
            return res
        return out.get("note", "no counters of %s in the profiler's output" % short)
    finally:
        shutil.rmtree(td, ignore_errors=True)


def run_ref(O, mtx, td, option, iters, args, cores, avx512):
    exe = O.ref_binary(avx512)
    cmd = [exe, "-input", mtx, "-output", "/nonexistent_dir_so_no_embd_is_written/", "-iter", str(iters), "-batch", str(args.batch),
           "-dim", str(args.dim), "-nsamples", "5", "-lr", "0.02", "-option", str(option), "-threads", str(cores)]
    out = subprocess.run(cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
    return out


def self_launch(n_ranks, out):
    """-> exit code.  One rank per GPU through torch.distributed.run on 127.0.0.1 with a free port; the child ranks are this very
    script with the same arguments (they find WORLD_SIZE set and run the benchmark)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("bench: --gpus %d without a launcher: starting %s" % (n_ranks, " ".join(cmd[1:8])))
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for text in child.stdout:  # rank 0's JSON line is the only thing the ranks write to stdout; anything else goes to stderr
        text = text.rstrip("\n")
        if text.startswith("{") and text.endswith("}"):
            line = text
        elif text:
            log(text)
    rc = child.wait()
    if line is not None:
        print(line, file=out, flush=True)
    elif rc == 0:
        log("bench: the ranks ended without a result line")
        rc = 4
    return rc


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
    ap.add_argument("--cpu-scale", type=int, default=20, help="cpu_baseline graph: RMAT scale (20 = the benchmark's own 15.7 M-edge graph)")
    ap.add_argument("--cpu-iters", type=int, default=16, help="cpu_baseline: epochs of the reference on the large graph")
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
    ap.add_argument("--extra-batches", type=str, default="256,384,1024,2048,4096,16384,262144", help="N=1: comma list of further batch sizes to time (reported under 'extra')")
    ap.add_argument("--dist-extra-batches", type=str, default="262144", help="N>1: the same for the sharded run (fewer, larger exchanges)")
    ap.add_argument("--verify-rows", type=int, default=32, help="N=1: sampled rows of one more minibatch checked against the CPU oracle after the timed region (0 = skip)")
    ap.add_argument("--config5-scale", type=int, default=24, help="also measure BASELINE configs[4] (RMAT of this scale, option 11) and report it under 'extra' (0 = skip)")
    ap.add_argument("--config5-batch", type=int, default=1048576)
    ap.add_argument("--config4", type=int, default=1, help="also measure BASELINE configs[3] (com-Orkut-sized graph, option 6) and report it under 'extra' (0 = skip)")
    ap.add_argument("--config4-batch", type=int, default=262144)
    ap.add_argument("--config4-mtx", type=str, default=os.environ.get("F2V_ORKUT_MTX", ""),
                    help="the real com-orkut.ungraph as a MatrixMarket file (or its .f2vcsr cache), where it has been supplied on this box "
                         "(also: environment variable F2V_ORKUT_MTX); otherwise a synthetic graph of com-Orkut's size stands in and the line says so")
    ap.add_argument("--cora", type=int, default=1, help="N=1: also time BASELINE configs[0] and [1] (cora, option 5, batch 256, 1200 epochs at D = 16 / 128) and "
                                                        "the reference's single-threaded option 5 beside config 0 (reported under 'extra'; 0 = skip)")
    ap.add_argument("--live-pmc", type=int, default=60, help="N=1: roofline.traffic measured by this run -- separate rocprofv3 --pmc passes over a short child run of the same workload, "
                                                             "within this many seconds (0 = take the figure of the committed builder-side session, profiles/traffic.json, where its key matches)")
    ap.add_argument("--no-ceilings", action="store_true", help="skip the on-box ceilings (stream copy, row gathers) of the roofline object")
    ap.add_argument("--option7", type=int, default=1, help="N=1: also time option 7 (rForce2Vec, parity mode: walks from the one rand() stream on the host) on the main graph, "
                                                           "with its host walk generation alone and the reference's own option 7 beside it (reported under 'extra'; 0 = skip)")
    ap.add_argument("--sustained-s", type=float, default=5.0, help="N=1: the headline workload for this many seconds in ONE f2v_train call, with the rate second by second (0 = skip)")
    args = ap.parse_args()

    # stdout carries exactly ONE line, the JSON result: whatever libraries print there meanwhile (gloo announces its
    # connections on stdout) goes to stderr
    sys.stdout.flush()
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            # `python bench.py --gpus N` by itself: this process becomes the launcher.  It starts torch.distributed.run as a
            # CHILD before anything here has touched the GPU (no exec of a process that has), passes the ranks' stderr
            # through, relays rank 0's JSON line as its only stdout line and exits with the child's code.
            sys.exit(self_launch(args.gpus, real_stdout))
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
        if mbytes >= 1 << 31:  # kIpcMaxBytes of f2v_engine.hip
            mbytes = 2 * min(512 << 20, mbytes // 2)  # such engines map a landing buffer instead of the matrices
        pre_dev = 0 if args.dist_backend == "gloo" else local_rank
        # where the ranks' children meet: all ranks are children of one launcher (torch.distributed.run), whose pid makes
        # the name unique to this launch; with another launcher the children time out and the run falls back
        # (the ranks cannot hand each other a mkdtemp name before they can talk: a private per-user parent directory instead)
        parent = os.path.join(tempfile.gettempdir(), "f2v_preflight_uid%d" % os.getuid())
        os.makedirs(parent, mode=0o700, exist_ok=True)
        st = os.lstat(parent)
        if st.st_uid != os.getuid() or (st.st_mode & 0o077) or not os.path.isdir(parent) or os.path.islink(parent):
            raise SystemExit("bench: %s is not a private directory of this user" % parent)
        meet = os.path.join(parent, "%s_%d" % (os.environ.get("MASTER_PORT", "0"), os.getppid()))
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
    def sync_all():
        if use_dist:
            if args.dist_backend == "nccl":
                torch.cuda.synchronize()
            dist.barrier()
            if args.dist_backend == "nccl":
                torch.cuda.synchronize()

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def all_ranks_agree(flag):
        if not use_dist:
            return bool(flag)
        t = torch.tensor([1 if flag else 0], dtype=torch.int32, device="cuda" if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item())

    class Session:
        """One engine on one graph / option: epochs (sharded over the ranks when N > 1), timing, verification."""

        def __init__(self, rowptr, colids, option, exchange, note_text=None):
            self.rowptr, self.colids, self.option = rowptr, colids, option
            self.n, self.nnz = len(rowptr) - 1, len(colids)
            self.init_kind = F._lib.INIT_SYMMETRIC if option in (5, 8, 11) else F._lib.INIT_UNIT
            self.eng = F.Engine(rowptr, colids, args.dim, device=local_rank)
            if args.hub_chunk >= 0:
                self.eng.set_param("hub_chunk", args.hub_chunk)
            for kv in args.param:
                k, v = kv.split("=")
                self.eng.set_param(k, int(v))
            self.eng.set_param("count_compulsory", 1)
            self.state = {"exchange": exchange, "comm": None, "note": note_text}
            self.schedule = []  # every (epochs, batch) trained so far: the verification replays it on one engine
            self.reset()

        def reset(self):
            self.eng.srand(1)
            self.eng.init_embeddings(self.init_kind)
            del self.schedule[:]

        def barrier(self):
            self.eng.synchronize()
            sync_all()

        def make_comm(self):
            """The exchange this run uses; the push exchange is attached (and self-tested) here, once."""
            st = self.state
            if st["comm"] is not None:
                return st["comm"]
            ex = st["exchange"]
            if ex == "push":
                comm = fdist.PushExchange(dist, rank, world, group=host_group)
                try:
                    comm.attach(self.eng)
                    note(rank, "push exchange attached, self-test passed")
                except Exception as exn:  # raised on every rank alike (the ranks agree inside attach)
                    st["note"] = "push exchange unavailable (%s): fell back to the RCCL all-gather" % (str(exn)[:200],)
                    log("bench[rank %d]: %s" % (rank, st["note"]))
                    st["exchange"] = ex = "allgather"
            if ex == "allgather":
                comm = (fdist.HostStageComm(dist, rank, world) if args.dist_backend == "gloo"
                        else fdist.NcclStageComm(dist, rank, world, local_rank, host_group=host_group))
            elif ex == "need":
                comm = (fdist.NeedExchange(dist, rank, world, backend="host") if args.dist_backend == "gloo"
                        else fdist.NeedExchange(dist, rank, world, device=local_rank, backend="device"))
            st["comm"] = comm
            return comm

        def run_epochs(self, k, batch):
            """-> per-rank statistics of these k epochs (launches, bytes, device seconds where known)."""
            eng = self.eng
            if not use_dist:
                eng.train(self.option, k, batch, 5, 0.02, 0)
                self.schedule.append((k, batch, eng.get_param("hub_chunk")))
                return eng.stats()
            s0 = eng.stats()
            comm = self.make_comm()
            fdist.ShardedTrainer(eng, rank, world, comm, exchange_when_single=True).train(self.option, k, batch, 5, 0.02, 0)
            self.schedule.append((k, batch, eng.get_param("hub_chunk")))  # the chunk this call ran with: part of the summation order
            s1 = eng.stats()
            if self.state["exchange"] == "push":
                st = dict(s1)  # f2v_train_sharded restarts the statistics like f2v_train
            else:
                st = {key: s1[key] - s0[key] for key in s1}
            st["device_seconds"] = None  # the sharded loop is timed by the wall clock
            return st

        def settle(self, batch):
            """Same workload, untimed, until the device has been under steady load for --settle-ms."""
            if args.settle_ms <= 0:
                return 0
            self.run_epochs(2, batch)  # launch plans are built here, on the host: not load yet
            self.eng.synchronize()
            done, t0 = 2, time.perf_counter()
            while max_over_ranks((time.perf_counter() - t0) * 1e3) < args.settle_ms and done < 1000:
                self.run_epochs(4, batch)
                self.eng.synchronize()
                done += 4
            return done

        def timed(self, k, w, batch):
            self.state["settle_epochs"] = self.settle(batch)
            if w > 0:
                self.run_epochs(w, batch)
            self.barrier()
            t0 = time.perf_counter()
            st = self.run_epochs(k, batch)
            self.barrier()
            dt = max_over_ranks(time.perf_counter() - t0)
            return dt, st

        def tune_push(self, batch):
            """Untimed: which of the two push variants is faster here?  Every rank sees the same (max-over-ranks) times."""
            eng = self.eng
            if args.push_fused != "auto":
                eng.set_param("push_fused", int(args.push_fused))
                return
            self.run_epochs(1, batch)  # launch plans and reader masks exist from here on
            note(rank, "first sharded epoch done")
            self.settle(batch)         # both variants are timed on a GPU that has left the power-management transient behind
            took = {1: float("inf"), 0: float("inf")}
            for fused in (1, 0, 1, 0):  # alternating, best of two each
                eng.set_param("push_fused", fused)
                self.barrier()
                t0 = time.perf_counter()
                self.run_epochs(5, batch)
                self.barrier()
                took[fused] = min(took[fused], max_over_ranks(time.perf_counter() - t0) / 5)
            best = 1 if took[1] <= took[0] else 0
            eng.set_param("push_fused", best)
            note(rank, "push variants timed: fused %.3f ms, separate kernel %.3f ms per epoch" % (took[1] * 1e3, took[0] * 1e3))
            self.state["tuned"] = {"push_fused": best, "ms_per_epoch_fused": took[1] * 1e3, "ms_per_epoch_separate_kernel": took[0] * 1e3}

        def verify_replica(self):
            """Every rank: the same epochs on ONE engine (no sharding, no exchange) must give this replica bit for bit."""
            ref = F.Engine(self.rowptr, self.colids, args.dim, device=local_rank)
            for kv in args.param:
                pk, pv = kv.split("=")
                if not pk.startswith("push_"):
                    ref.set_param(pk, int(pv))
            ref.srand(1)
            ref.init_embeddings(self.init_kind)
            for i, (k, b, chunk) in enumerate(self.schedule):
                try:
                    ref.set_param("hub_chunk", chunk)  # the chunk is part of the summation order (a sharded run picks it per slice)
                    ref.train(self.option, k, b, 5, 0.02, 0)
                except Exception:
                    note(rank, "single-GPU replay failed in call %d of %d (%d epochs at batch %d)" % (i, len(self.schedule), k, b))
                    raise
            same = bool(np.array_equal(ref.get_embeddings(), self.eng.get_embeddings()))
            ref.close()
            return all_ranks_agree(same)

        def close(self):
            if use_dist and self.state["exchange"] == "push" and self.state["comm"] is not None:
                self.state["comm"].detach(self.eng)
            self.eng.close()

    failed = []  # anything here makes the line carry "failed": true and the process exit non-zero

    def measure(sess, steps, warmup, batch, tune):
        """The timed region + (N > 1) the bit-for-bit replica check.  A push replica that differs from the single-GPU
        run is a FAILURE of the run; the RCCL re-measurement afterwards is additional data, not a substitute."""
        if use_dist and world > 1 and sess.state["exchange"] == "push":
            sess.make_comm()
            if sess.state["exchange"] == "push" and tune:
                sess.tune_push(batch)
        dt, st = sess.timed(steps, warmup, batch)
        if st is not None and st.get("recovered"):
            # a give-up inside the timed region: f2v_train ran the call again with one launch per minibatch and tree level -- correct
            # bits, but not the launch forms this line is about
            failed.append("a launch of the timed region was lost and recovered from (option %d, batch %d): its time is the slow launch forms'" % (sess.option, batch))
        verified = None
        if use_dist and not args.no_verify:
            verified = sess.verify_replica()
            note(rank, "replica compared with a single-GPU run of the same %d epochs: %s" % (sum(e[0] for e in sess.schedule), "identical" if verified else "DIFFERENT"))
            if not verified:
                failed.append("%s exchange: replica differs from the single-GPU run (option %d, batch %d, n=%d)" % (sess.state["exchange"], sess.option, batch, sess.n))
                if sess.state["exchange"] == "push":
                    sess.state["note"] = "FAILED: the push exchange gave a replica that differs from the single-GPU run; the numbers below were measured again over the RCCL all-gather"
                    log("bench[rank %d]: %s" % (rank, sess.state["note"]))
                    sess.state["comm"].detach(sess.eng)
                    sess.state["comm"], sess.state["exchange"] = None, "allgather"
                    sess.reset()
                    dt, st = sess.timed(steps, warmup, batch)
                    verified_again = sess.verify_replica()
                    if not verified_again:
                        failed.append("RCCL all-gather: replica differs from the single-GPU run too")
        return dt, st, verified

    rowptr, colids = load_graph(args.scale, 16, 1)
    n, nnz = len(rowptr) - 1, len(colids)
    note(rank, "graph ready: n=%d nnz=%d" % (n, nnz))
    sess = Session(rowptr, colids, args.option, args.exchange, getattr(args, "preflight_note", None))
    eng, state = sess.eng, sess.state
    note(rank, "engine ready, embeddings initialised")
    dt, st, verified = measure(sess, args.steps, args.warmup, args.batch, tune=True)
    main_settle = state["settle_epochs"]
    main_form = eng.get_param("last_train_form")
    main_wide_width = eng.get_param("last_wide_width") if main_form == 2 else 0
    main_wide_early = (2 if eng.get_param("last_wide_epochs") > 1 else eng.get_param("last_wide_early")) if main_form == 2 else 0
    note(rank, "timed region done: %.3f ms per epoch" % (dt / args.steps * 1e3))
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
    if st is not None:
        now = eng.stats()  # (the handle's own answers: a sharded run's `st` holds differences)
        res["config"]["recoveries"] = int(now.get("recoveries", 0))          # give-ups f2v_train recovered from on this handle so far (0 expected)
        res["config"]["merge_finalize"] = int(now.get("merge_finalize", 1))  # 0: the handle has fallen back to launches without in-grid waits
        if not use_dist:
            res["config"]["snapshot_copy_ms_per_train_call"] = st.get("snapshot_seconds", 0.0) * 1e3  # "recover": outside ms_per_step's device time, inside the wall time
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
    if not use_dist and args.verify_rows > 0:
        # (before the ceilings below: the first launch behind their 4-GiB allocation and release took 19-28 ms under rocprofv3 and
        # lifted the kernel's profiled average; the figures of the timed region come from `st`, taken above)
        res["config"]["verified_rows"] = verify_rows(F, eng, rowptr, colids, args, args.verify_rows)
        note(rank, "verification minibatch: %s sampled rows bit-identical to the oracle" % res["config"]["verified_rows"])
    if st is not None:
        # dominant kernel = the step kernel, one launch per minibatch; launch time from HIP events on the engine's stream
        # around the epoch loop (f2v_train).  N > 1: rank 0's share of the bytes over the wall time of the sharded loop.
        launches = max(st["step_launches"], 1)
        t_launch = (st["device_seconds"] if st["device_seconds"] else dt) / launches
        comp = st["compulsory_bytes"] / launches
        alg = st["algorithmic_bytes"] / launches
        kname = kernel_name(args, use_dist and world > 1 and state["exchange"] == "push" and eng.get_param("push_fused"), main_form, main_wide_width, main_wide_early)
        ach = comp / t_launch * 1e-9
        roof = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None,
                "kernel": kname, "compulsory_bytes": comp, "compulsory_bytes_per_launch": comp, "avg_launch_us": t_launch * 1e6, "launches": st["step_launches"],
                "definition": "achieved = compulsory bytes per launch (distinct embedding rows read + rows written + neighbour ids + work items, "
                              "counted by the engine) / average launch time; frac <= 1 by construction",
                # SURVEY 8d's count charges EVERY neighbour row to HBM: a rate of useful bytes delivered to the CUs, served by L2 + Infinity Cache + HBM together
                "algorithmic_bytes_per_launch": alg, "algorithmic_GBs": alg / t_launch * 1e-9,
                "algorithmic_frac_note": ("SURVEY 8d bytes / launch time / peak = %.2f%s: every neighbour row is charged to HBM there although re-reads of hub rows are served by "
                                          "the caches -- a delivery rate of useful bytes (L2 + Infinity Cache + HBM together), not a fraction of the HBM roofline")
                                         % (alg / t_launch * 1e-9 / HBM_PEAK_GBS, " (> 1)" if alg / t_launch * 1e-9 > HBM_PEAK_GBS else "")}
        live = None
        if not use_dist and args.live_pmc > 0:
            note(rank, "roofline.traffic: rocprofv3 --pmc passes over a child run of the same workload (at most %d s)" % args.live_pmc)
            live = live_traffic(args, kname, args.live_pmc)
            if isinstance(live, dict):
                roof["traffic"] = live["traffic"]
                roof["traffic_is"] = "bytes that left the XCDs' L2s per launch: FETCH_SIZE x 2 (the guide's gfx950 correction for 16-byte-per-lane reads) + WRITE_SIZE, HBM + Infinity Cache together (rocprofv3 cannot split them on gfx950)"
                roof["traffic_source"] = ("measured by this run: separate rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE; TCC_HIT_sum TCC_MISS_sum) over a child run of the same workload on this box, "
                                          "%s launches of the kernel sampled per pass, %.0f s" % (live.get("launches_sampled"), live["seconds"]))
                roof["traffic_fetch_bytes"], roof["traffic_write_bytes"] = live["fetch_bytes"], live["write_bytes"]
                roof["l2_miss_GBs"] = live["traffic"] / t_launch * 1e-9
                if "l2_hit_rate" in live:
                    roof["l2_hit_rate"] = live["l2_hit_rate"]
                if "l2_request_bytes" in live:
                    # what the L2s SERVED (hits and misses, TCC_HIT + TCC_MISS requests of 128 B); row_gather_from_l2_GBs_on_this_box below is the
                    # rate at which they deliver row gathers that all hit (profiles/r04_launch_time_decomposition.txt)
                    roof["l2_request_bytes_per_launch"] = live["l2_request_bytes"]
                    roof["l2_request_GBs"] = live["l2_request_bytes"] / t_launch * 1e-9
                roof["traffic_over_compulsory"] = live["traffic"] / comp
                roof["hbm_bytes_per_launch_between"] = [comp, live["traffic"]]
                if "note" in live:
                    roof["traffic_note"] = live["note"]
            else:
                roof["traffic_note"] = "no live PMC pass: %s" % (live,)
        tp = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tp) and not use_dist and not isinstance(live, dict):
            try:
                tj = json.load(open(tp))
                key = {"batch": args.batch, "scale": args.scale, "dim": args.dim, "option": args.option, "hub_chunk": eng.get_param("hub_chunk"),
                       "kernel": kname, "library": F._lib.lib().f2v_version().decode()}
                if all(tj.get(k) == v for k, v in key.items()):
                    roof["traffic"] = tj.get("l2_miss_bytes_per_launch")
                    # NOT measured by this run: replayed from the committed PMC session of the same workload / kernel / library version
                    roof["traffic_source"] = ("profiles/traffic.json: builder-side rocprofv3 --pmc session (tools/profile_traffic.sh, separate passes), collected %s for library '%s'; "
                                              "replayed here because its workload / kernel / library key matches -- not a measurement of this run"
                                              % (tj.get("collected", "in an earlier session"), tj.get("library")))
                    roof["traffic_is"] = "bytes that left the XCDs' L2s per launch (PMC: TCC_EA0_RDREQ/WRREQ, calibrated): HBM + Infinity Cache, which rocprofv3 cannot split on gfx950"
                    roof["l2_miss_GBs"] = roof["traffic"] / t_launch * 1e-9
                    roof["l2_hit_rate"] = tj.get("l2_hit_rate")
                    roof["hbm_bytes_per_launch_between"] = [comp, roof["traffic"]]
                else:
                    roof["traffic_note"] = (roof.get("traffic_note", "") + "; " if roof.get("traffic_note") else "") + "profiles/traffic.json was collected for another workload / kernel / library build: not applied"
            except Exception:
                pass
        if not use_dist and not args.no_ceilings:
            import ctypes
            g = ctypes.c_double()
            L = F._lib.lib()
            if L.f2v_diag_stream_copy(local_rank, 1 << 30, 5, ctypes.byref(g)) == 0:
                roof["stream_copy_GBs_on_this_box"] = g.value   # 1-GiB copy, read + written bytes
            if L.f2v_diag_gather_rate(local_rank, 4 << 30, 2, ctypes.byref(g)) == 0:
                roof["row_gather_from_hbm_GBs_on_this_box"] = g.value  # random 512-B rows of a 4-GiB table, each once
            if L.f2v_diag_gather_rate(local_rank, 64 << 20, 2, ctypes.byref(g)) == 0:
                roof["row_gather_from_infinity_cache_GBs_on_this_box"] = g.value  # the same from a 64-MiB table
            if L.f2v_diag_gather_rate(local_rank, 1 << 20, 2, ctypes.byref(g)) == 0:
                roof["row_gather_from_l2_GBs_on_this_box"] = g.value  # the same from a 1-MiB table: every XCD's L2 holds it -- the L2s' delivery rate for row gathers
                if roof.get("l2_request_GBs"):
                    roof["l2_request_frac_of_l2_gather_rate"] = roof["l2_request_GBs"] / g.value
                    roof["launch_time_note"] = ("the launch's L2 requests (hits + misses) move at %.2f of the rate at which this box's L2s deliver row gathers that all hit, its misses at %.2f "
                                                "of the box's pure-miss gather rate from the Infinity Cache: neither path is saturated; a gather-only replay of the same launch plans "
                                                "takes 0.88 of the kernel's time, 0.71 with the hub chunk cut to 32 (no long dependent chains) -- profiles/r04_launch_time_decomposition.txt"
                                                % (roof["l2_request_GBs"] / g.value, roof.get("l2_miss_GBs", 0.0) / max(roof.get("row_gather_from_infinity_cache_GBs_on_this_box", 0.0), 1e-9)))
        res["roofline"] = roof
    extra = {}
    if rank == 0 and not use_dist:
        for b in [int(x) for x in args.extra_batches.split(",") if x]:
            k = max(1, args.steps // 2)
            dtb, stb = sess.timed(k, 1, b)
            if stb.get("recovered"):
                failed.append("batch %d: a launch of the timed region was lost and recovered from" % b)
            extra["batch_%d" % b] = {"edges_per_s": units * k / dtb, "ms_per_epoch": dtb / k * 1e3,
                                     "algorithmic_GBs": stb["algorithmic_bytes"] / stb["device_seconds"] * 1e-9,
                                     "compulsory_GBs": stb["compulsory_bytes"] / stb["device_seconds"] * 1e-9}
    if use_dist and world > 1:
        for b in [int(x) for x in args.dist_extra_batches.split(",") if x]:
            k = max(1, args.steps // 2)
            dtb, _ = sess.timed(k, 1, b)
            extra["batch_%d" % b] = {"edges_per_s": units * k / dtb, "ms_per_epoch": dtb / k * 1e3}
        if extra and not args.no_verify:  # everything trained since the first check is checked too
            ok = sess.verify_replica()
            extra["replicas_bit_identical_to_1gpu_run"] = ok
            if not ok:
                failed.append("extra batches: replica differs from the single-GPU run")
    if rank == 0 and not use_dist and args.sustained_s > 0:
        # the headline workload for seconds, in ONE f2v_train call (no host synchronisation inside), with the rate second by
        # second from HIP events every few epochs ("epoch_marks"): power capping or a clock ramp would show as a drifting rate
        per_epoch = (st["device_seconds"] if st and st.get("device_seconds") else dt) / args.steps
        iters = max(8, int(1.05 * args.sustained_s / per_epoch) + 1)
        every = max(1, iters // 400)
        eng.set_param("epoch_marks", every)
        t0 = time.perf_counter()
        eng.train(args.option, iters, args.batch, 5, 0.02, 0)
        wall = time.perf_counter() - t0
        marks = eng.train_marks()
        eng.set_param("epoch_marks", 0)
        sess.schedule.append((iters, args.batch, eng.get_param("hub_chunk")))
        dev = eng.stats()["device_seconds"]
        rates = []
        if len(marks) >= 2:
            sec, last_t, last_k = 1.0, 0.0, 0
            for k, tm in enumerate(marks):
                if tm >= sec or k == len(marks) - 1:
                    rates.append(units * (k + 1 - last_k) * every / max(tm - last_t, 1e-9))
                    last_t, last_k, sec = tm, k + 1, sec + 1.0
        extra["sustained"] = {"seconds_device": dev, "seconds_wall": wall, "epochs": iters, "edges_per_s": units * iters / dev,
                              "per_second_edges_per_s_min": min(rates) if rates else None, "per_second_edges_per_s_max": max(rates) if rates else None,
                              "per_second_edges_per_s": rates}
        note(rank, "sustained: %d epochs in %.2f s of device time, %.2f G edges/s" % (iters, dev, units * iters / dev * 1e-9))
    if rank == 0 and not use_dist and args.option7:
        # option 7 (rForce2Vec) in parity mode on the same graph: its walks are draws of the ONE serial rand() stream, generated on the
        # host (a producer thread, one epoch ahead of the device): epoch time, host generation alone, device work alone ("fast_rng":
        # the same kernels, walks from a device-side generator -- non-parity numbers)
        e7 = F.Engine(rowptr, colids, args.dim, device=local_rank)
        e7.srand(1)
        e7.init_embeddings(F._lib.INIT_UNIT)
        t0 = time.perf_counter()
        for _ in range(3):
            e7.generate_walks()
        host_ms = (time.perf_counter() - t0) / 3 * 1e3
        e7.train(7, 2, args.batch, 5, 0.02, 0)
        t0 = time.perf_counter()
        e7.train(7, 6, args.batch, 5, 0.02, 0)
        wall_ms = (time.perf_counter() - t0) / 6 * 1e3
        e7.set_param("fast_rng", 1)
        e7.train(7, 2, args.batch, 5, 0.02, 0)
        dev_ms = e7.train(7, 6, args.batch, 5, 0.02, 0) / 6 * 1e3
        e7.close()
        extra["option7"] = {"workload": "RMAT scale-%d (n=%d), option 7, D=%d, batch=%d, parity mode (walks from the serial rand() stream)" % (args.scale, n, args.dim, args.batch),
                            "ms_per_epoch": wall_ms, "pairs_per_s": 5.0 * n / (wall_ms * 1e-3), "host_walk_generation_ms_per_epoch": host_ms,
                            "device_only_ms_per_epoch_fast_rng": dev_ms}
        note(rank, "option 7: %.1f ms per epoch (host walks alone %.1f ms, device alone %.2f ms)" % (wall_ms, host_ms, dev_ms))
    sess.close()
    del sess, eng
    if rank == 0 and not use_dist and args.cora:
        # BASELINE configs[0] / [1]: the reference's own bundled graph (tests/golden/cora.mtx is that file), option 5, batch 256,
        # 1200 epochs, D = 16 and 128: device seconds of the epoch loop; first a 10-epoch run compared with the reference's
        # committed output (tests/golden, 3e-5) and, bit for bit, with the CPU oracle in the kernels' summation order
        import gzip
        from oracle import oracle as O
        gold = os.path.join(ROOT, "tests", "golden")
        crp, cci = F.read_mtx(os.path.join(gold, "cora.mtx"))
        for key, dim in (("config0_cora_D16", 16), ("config1_cora_D128", 128)):
            ce = F.Engine(crp, cci, dim, device=local_rank)
            ce.srand(1)
            ce.init_embeddings(F._lib.INIT_SYMMETRIC)
            ce.train(5, 10, 256, 5, 0.02, 0)
            got = ce.get_embeddings()
            want = O.train(5, crp, cci, dim, 10, 256, order=O.ORDER_TREE, chunk=ce.get_param("hub_chunk"))
            with gzip.open(os.path.join(gold, "cora_opt5_it10_B256_D%d_bs0.embd.gz" % dim), "rb") as f:
                tmp = tempfile.NamedTemporaryFile(suffix=".embd", delete=False)
                tmp.write(f.read())
                tmp.close()
                ref = F.read_embd(tmp.name)
                os.unlink(tmp.name)
            err = float(np.abs(got - ref).max())
            if not np.array_equal(got, want) or not err < 3e-5:
                failed.append("%s: 10 epochs differ from the oracle (bit-identical: %s) or from the reference's committed output by %g" % (key, np.array_equal(got, want), err))
            best = None
            for _ in range(3):
                ce.srand(1)
                ce.init_embeddings(F._lib.INIT_SYMMETRIC)
                sec = ce.train(5, 1200, 256, 5, 0.02, 0)
                best = sec if best is None else min(best, sec)
            extra[key] = {"workload": "cora.mtx (n=%d, nnz=%d), option 5, D=%d, batch 256, 1200 epochs" % (len(crp) - 1, len(cci), dim),
                          "seconds_device": best, "edges_per_s": len(cci) * 1200 / best, "launch_form": {0: "one launch per minibatch", 1: "chained", 2: "chained, wide form", 3: "hipGraph replay"}[ce.get_param("last_train_form")],
                          "epochs_per_launch": ce.get_param("last_wide_epochs"), "recoveries": ce.get_param("recoveries"),
                          "epochs10_max_abs_vs_reference_output": err, "epochs10_bit_identical_to_oracle": bool(np.array_equal(got, want))}
            ce.close()
        try:  # config 0 as BASELINE words it: the reference CPU path, single thread (oracle/_ref option 5 -threads 1; T(1200) - T(0) strips its in-timer init)
            if O.ref_binary(False) is not None:
                with tempfile.TemporaryDirectory() as td:
                    def ref_seconds(iters):
                        exe = O.ref_binary(False)
                        cmd = [exe, "-input", os.path.join(gold, "cora.mtx"), "-output", "/nonexistent_dir_so_no_embd_is_written/", "-iter", str(iters), "-batch", "256",
                               "-dim", "16", "-nsamples", "5", "-lr", "0.02", "-option", "5", "-threads", "1"]
                        out = subprocess.run(cmd, cwd=td, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
                        return float([l for l in out.splitlines() if "Wall time required" in l][0].split(":")[-1].split()[0])
                    t_ref = ref_seconds(1200) - ref_seconds(0)
                extra["config0_cora_D16"]["reference_cpu_1_thread_seconds"] = t_ref
                extra["config0_cora_D16"]["reference_cpu_1_thread_edges_per_s"] = len(cci) * 1200 / t_ref
        except Exception as ex:
            extra["config0_cora_D16"]["reference_cpu_1_thread_seconds"] = "failed: %r" % (ex,)
    def load_orkut_like():
        """BASELINE configs[3]: com-Orkut itself where its file has been supplied (--config4-mtx / F2V_ORKUT_MTX: parsed once by
        libf2v's reader, kept as a binary CSR next to it when the directory is writable, else under /tmp), otherwise a
        synthetic graph of its size (3 072 441 vertices, 117 185 083 edges), cached like the RMAT graphs."""
        if args.config4_mtx and os.path.exists(args.config4_mtx):
            path = args.config4_mtx
            if path.endswith(".f2vcsr"):
                return F.graph.read_csr_bin(path)
            for cache in (path + ".f2vcsr", "/tmp/f2v_orkut_real.f2vcsr"):
                if os.path.exists(cache) and os.path.getmtime(cache) >= os.path.getmtime(path):
                    return F.graph.read_csr_bin(cache)
            t0 = time.time()
            rp, ci = F.read_mtx(path)
            log("bench: parsed %s: n=%d nnz=%d in %.1fs" % (path, len(rp) - 1, len(ci), time.time() - t0))
            for cache in (path + ".f2vcsr", "/tmp/f2v_orkut_real.f2vcsr"):
                try:
                    F.graph.write_csr_bin(cache, rp, ci)
                    break
                except Exception:
                    continue
            return rp, ci
        from force2vec_amd.graph import orkut_like_csr
        cache = "/tmp/f2v_orkut_like_seed1.npz"
        if os.path.exists(cache):
            try:
                z = np.load(cache)
                return z["rowptr"], z["colids"]
            except Exception:
                pass
        t0 = time.time()
        rp, ci = orkut_like_csr(1)
        log("bench: generated the Orkut-sized graph: n=%d nnz=%d in %.1fs" % (len(rp) - 1, len(ci), time.time() - t0))
        try:
            tmp = cache + ".%d.tmp.npz" % os.getpid()
            np.savez(tmp, rowptr=rp, colids=ci)
            os.replace(tmp, cache)
        except Exception:
            pass
        return rp, ci

    def extra_config(label, load, option, batch):
        """One more BASELINE configuration, same measurement and the same checks as the headline: a few epochs, then (N > 1) the
        bit-for-bit replica check or (N = 1) the sampled-row oracle check."""
        if not use_dist or rank == 0:
            load()  # one rank generates (or finds) the cached graph, the others read the cache
        sync_all()
        rp, ci = load()
        note(rank, "%s: graph ready: n=%d nnz=%d" % (label, len(rp) - 1, len(ci)))
        sx = Session(rp, ci, option, state["exchange"] if use_dist else args.exchange)
        kx = max(2, min(args.steps, 5))
        opt_saved, args.option = args.option, option
        dtx, stx, verx = measure(sx, kx, 1, batch, tune=False)
        out = {"workload": "%s (n=%d, nnz=%d), option %d, D=%d, batch=%d" % (label, len(rp) - 1, len(ci), option, args.dim, batch),
               "edges_per_s": len(ci) * kx / dtx, "ms_per_epoch": dtx / kx * 1e3, "epochs": kx, "hub_chunk": sx.eng.get_param("hub_chunk")}
        if use_dist:
            out["exchange"] = sx.state["exchange"]
            out["replicas_bit_identical_to_1gpu_run"] = verx
            if sx.state["note"]:
                out["note"] = sx.state["note"]
        else:
            out["compulsory_GBs"] = stx["compulsory_bytes"] / stx["device_seconds"] * 1e-9
            out["algorithmic_GBs"] = stx["algorithmic_bytes"] / stx["device_seconds"] * 1e-9
            if args.verify_rows > 0:
                ax = argparse.Namespace(**vars(args))
                ax.batch = batch
                out["verified_rows"] = verify_rows(F, sx.eng, rp, ci, ax, args.verify_rows)
        args.option = opt_saved
        sx.close()
        return out

    if args.config5_scale > 0:
        # BASELINE configs[4]: RMAT scale-24, option 11, D = 128, minibatches of 1 M rows -- the size at which a minibatch is
        # milliseconds of work per rank, i.e. where the strong-scaling target lives
        extra["config5_rmat%d_option11" % args.config5_scale] = extra_config(
            "RMAT scale-%d" % args.config5_scale, lambda: load_graph(args.config5_scale, 16, 1), 11, args.config5_batch)
    if args.config4:
        # BASELINE configs[3]: com-Orkut's size (3.07 M vertices, 117 M edges; synthetic stand-in), option 6 (sigmoid)
        real = bool(args.config4_mtx and os.path.exists(args.config4_mtx))
        extra["config4_orkut_option6" if real else "config4_orkut_sized_option6"] = extra_config(
            "com-Orkut (%s)" % args.config4_mtx if real else "Orkut-sized synthetic power-law graph (the real com-orkut.ungraph file was not supplied: --config4-mtx / F2V_ORKUT_MTX)",
            load_orkut_like, 6, args.config4_batch)
    if extra:
        res["extra"] = extra
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            res["cpu_baseline"] = cpu_baseline(args, extra.get("option7") if args.cpu_scale == args.scale else None)
        except Exception as ex:  # the baseline is reported, never required for the GPU number
            res["cpu_baseline"] = {"value": None, "unit": "edges/s", "cores": os.cpu_count(), "kind": "reference", "sample": "failed: %r" % (ex,)}
    if failed:
        res["failed"] = True
        res["failures"] = failed
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), file=real_stdout, flush=True)
    if failed:
        sys.exit(3)


if __name__ == "__main__":
    main()

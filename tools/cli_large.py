#!/usr/bin/env python3
"""The drop-in CLI end to end on a large graph: bin/Force2Vec on an RMAT graph written as a MatrixMarket file -- whole-process wall time and its
parts (the reader's stage times with F2V_IO_TRACE, the printed training time, the text .embd written by the host's threads).
usage: cli_large.py [SCALE [ITER [BATCH]]]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from force2vec_amd.graph import edges_from_csr

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
iters = sys.argv[2] if len(sys.argv) > 2 else "100"
batch = sys.argv[3] if len(sys.argv) > 3 else "384"
rowptr, colids = bench.load_graph(scale, 16, 1)
mtx = "/tmp/f2v_cli_rmat%d.mtx" % scale
if not os.path.exists(mtx):
    src, dst = edges_from_csr(rowptr, colids)
    bench.write_mtx_fast(mtx, len(rowptr) - 1, src, dst)
out = "/tmp/f2v_cli_out/"
os.makedirs(out, exist_ok=True)
for threads in (None, "1"):
    env = dict(os.environ, F2V_IO_TRACE="1")
    if threads:
        env["F2V_IO_THREADS"] = threads
    t0 = time.perf_counter()
    r = subprocess.run([os.path.join(ROOT, "bin", "Force2Vec"), "-input", mtx, "-output", out, "-iter", iters, "-batch", batch, "-dim", "128", "-option", "5"],
                       cwd=out, env=env, capture_output=True, text=True)
    wall = time.perf_counter() - t0
    files = [f for f in os.listdir(out) if f.endswith(".embd")]
    size = sum(os.path.getsize(os.path.join(out, f)) for f in files) / 1e9
    lines = [l for l in (r.stdout + r.stderr).splitlines() if "GPU epoch loop" in l or "Wall time" in l or "f2v_read_mtx" in l]
    print("bin/Force2Vec RMAT-%d (%.0f MB .mtx) -iter %s -batch %s -dim 128 -option 5, F2V_IO_THREADS=%s: exit %d, whole process %.2f s, .embd %.2f GB"
          % (scale, os.path.getsize(mtx) / 1e6, iters, batch, threads or "(unset)", r.returncode, wall, size))
    for l in lines:
        print("    " + l.strip())
    for f in files:
        os.remove(os.path.join(out, f))

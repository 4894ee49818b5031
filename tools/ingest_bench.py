#!/usr/bin/env python3
"""Ingest timing: MatrixMarket text -> CSR with libf2v's reader at 1 and many threads, the
binary CSR cache, and -- where oracle/_ref exists -- the reference's own ReadASCII + CSC + CSR path
(its -iter 0 run on the same file, wall clock of the whole process minus its embedding init)."""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import force2vec_amd as F
from force2vec_amd.graph import rmat_edges, write_mtx_symmetric

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n, s, d = rmat_edges(scale, 16, 1)
path = "/tmp/f2v_ingest_s%d.mtx" % scale
t = time.time()
write_mtx_symmetric(path, n, s, d)
print("wrote %s: %d entries, %.0f MB in %.0fs" % (path, len(s), os.path.getsize(path) / 1e6, time.time() - t), flush=True)
ref = None
for T in ("1", str(min(os.cpu_count() or 1, 64))):
    os.environ["F2V_IO_THREADS"] = T
    best = 1e9
    for _ in range(2):
        t = time.time()
        rp, ci = F.read_mtx(path)
        best = min(best, time.time() - t)
    print("f2v_read_mtx, %2s threads: %.2fs  (%.1f M entries/s)" % (T, best, len(s) / best / 1e6), flush=True)
    if ref is None:
        ref = (rp, ci)
    else:
        assert np.array_equal(ref[0], rp) and np.array_equal(ref[1], ci)
F.write_csr_bin(path + ".f2vcsr", rp, ci)
t = time.time()
rp2, ci2 = F.read_csr_bin(path + ".f2vcsr")
print("binary CSR cache: %.2fs" % (time.time() - t))
assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
exe = os.path.join(ROOT, "oracle", "_ref", "Force2Vec")
if os.path.exists(exe):
    t = time.time()
    out = subprocess.run([exe, "-input", path, "-output", "/nonexistent_dir/", "-iter", "0", "-dim", "1", "-option", "5", "-threads", str(os.cpu_count())],
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, cwd="/tmp").stdout
    wall = time.time() - t
    coo = [l for l in out.splitlines() if "ASCII to COO" in l]
    print("reference reader (ReadASCII + CSC + CSR, whole -iter 0 -dim 1 process): %.2fs wall; %s" % (wall, coo[0] if coo else ""))

#!/usr/bin/env python3
"""Host only: how long the text .embd of a large run takes to write (the drop-in boundary's writeToFile: "<N> <D>" then one line of %g values
per vertex), on one thread and on the host's threads (f2v_write_embd formats slices of rows side by side and writes them in order: same bytes).
usage: embd_write_time.py [ROWS [DIM]]"""
import hashlib
import os
import subprocess
import sys
import time

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
dim = int(sys.argv[2]) if len(sys.argv) > 2 else 128
code = r'''
import sys, time, hashlib
sys.path.insert(0, %r)
import numpy as np
import force2vec_amd as F
rng = np.random.default_rng(0)
X = rng.uniform(-3, 3, (%d, %d)).astype(np.float32)
t0 = time.perf_counter(); F.write_embd("/tmp/f2v_embd_time.embd", X); dt = time.perf_counter() - t0
h = hashlib.md5()
with open("/tmp/f2v_embd_time.embd", "rb") as f:
    for blk in iter(lambda: f.read(1 << 24), b""):
        h.update(blk)
import os
print("%%.2f s, %%.2f GB, md5 %%s" %% (dt, os.path.getsize("/tmp/f2v_embd_time.embd") / 1e9, h.hexdigest()[:12]))
''' % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), rows, dim)
for threads in ("1", None):
    env = dict(os.environ)
    if threads:
        env["F2V_IO_THREADS"] = threads
    else:
        env.pop("F2V_IO_THREADS", None)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print("%d x %d, F2V_IO_THREADS=%s (%d host cores): %s" % (rows, dim, threads or "(unset)", os.cpu_count(), (out.stdout.strip() or out.stderr.strip()[-300:])), flush=True)
os.remove("/tmp/f2v_embd_time.embd")

#!/usr/bin/env python3
"""Child process of bench.py (N > 1), started BEFORE the parent touches the GPU: rehearses what the push exchange needs
from this machine -- HIP IPC mapping of a buffer as large as an embedding matrix and of fine-grained flags between the
ranks' processes, remote stores from a kernel -- so that a mapping call that never returns or a faulting store happens
here and not in the benchmark.  No torch in this process (its first import on a fresh box takes a minute or two): the
ranks' children meet through files in `dir`, a directory the parents agree on (bench.py: the launcher's pid + port).
usage: ipc_preflight.py device rank world dir bytes timeout_s      -> exit code 0 = usable"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from force2vec_amd import _lib  # ctypes only

device, rank, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
d, nbytes, timeout_s = sys.argv[4], int(sys.argv[5]), float(sys.argv[6])
os.makedirs(d, exist_ok=True)
rc = _lib.lib().f2v_diag_ipc_preflight(device, rank, world, d.encode(), nbytes, timeout_s)
if rc != 0:
    print("ipc_preflight[rank %d]: %s" % (rank, _lib.lib().f2v_last_error().decode(errors="replace")), file=sys.stderr, flush=True)
sys.exit(0 if rc == 0 else 1)

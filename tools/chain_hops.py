#!/usr/bin/env python3
"""Where a chained launch spends its time (self-test build, f2v_test_stamps): one epoch of RMAT-20 with per-row device
time stamps, then the dependency chain that ends in each launch's last row is walked back -- from a row to the neighbour (or
sample) row of an earlier minibatch of the same launch whose flag was stored last -- and every hop is split into
    wait   the dependency's flag stored -> the row's last hub piece announced (flag hand-off + row load + the piece's gathers + store)
    tree   last piece announced -> last inner combine-tree node announced (rows of more than `fanin` pieces)
    root   -> the row's own flag stored (root node: loads the sums, adds, stores the row, waits for the acknowledgement)
Rows that are one item (degree <= hub chunk) have a single figure.  Usage: chain_hops.py [batch] [option] [key=value engine params ...]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import force2vec_amd as F
from force2vec_amd import _lib

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
option = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rowptr, colids = bench.load_graph(20, 16, 1)
n = len(rowptr) - 1
T = _lib.selftest_lib()
eng = F.Engine(rowptr, colids, 128, selftest=True)
for kv in sys.argv[3:]:
    k, v = kv.split("=")
    eng.set_param(k, int(v))
eng.srand(1)
eng.init_embeddings(0 if option in (5, 8, 11) else 1)
eng.train(option, 3, batch)
t_plain = min(eng.train(option, 3, batch) / 3 for _ in range(2))
_lib.check(T.f2v_test_stamps(eng._h, 1, None), T)
t_stamped = eng.train(option, 1, batch)
st = np.zeros(4 * n, dtype=np.uint64)
_lib.check(T.f2v_test_stamps(eng._h, 0, st.ctypes.data_as(C.POINTER(C.c_uint64))), T)
chunk = eng.get_param("hub_chunk")
chain_rows = eng.get_param("wide_rows" if eng.get_param("last_train_form") == 2 else "chain_rows")
fanin = eng.get_param("hub_fanin")
eng.close()
st = st.reshape(n, 4)
t_piece, t_node, t_row = st[:, 0].astype(np.int64), st[:, 1].astype(np.int64), st[:, 2].astype(np.int64)
t_seen = np.where(st[:, 3] != 0, (~st[:, 3]).astype(np.int64), 0)
deg = np.diff(rowptr.astype(np.int64))
pieces = np.maximum(1, (deg + chunk - 1) // chunk)
print("RMAT-20 option %d batch %d: hub chunk %d fan-in %d, %d rows per launch; epoch %.3f ms (%.3f ms with the stamps on)"
      % (option, batch, chunk, fanin, chain_rows, t_plain * 1e3, t_stamped * 1e3))
if not t_row.any():
    sys.exit("no stamps were recorded: the run was not chained")
K = max(1, chain_rows // batch)
edges = [1, fanin, 8 * fanin, fanin * fanin, 1 << 30]
names = ["one item", "2..%d pieces" % fanin, "..%d" % (8 * fanin), "..%d" % (fanin * fanin), "more"]
hops = {k: [] for k in range(len(edges))}
path_us, span_us, nhops = 0.0, 0.0, 0
rp = rowptr.astype(np.int64)
flag_lat = []
for lo in range(0, n, K * batch):
    hi = min(lo + K * batch, n)
    i = lo + int(np.argmax(t_row[lo:hi]))
    first = int(t_row[lo:hi][t_row[lo:hi] > 0].min())
    span_us += (int(t_row[i]) - first) / 100.0
    while True:
        mb_lo = (i // batch) * batch
        nb = colids[rp[i]:rp[i + 1]].astype(np.int64)
        nb = nb[(nb >= lo) & (nb < mb_lo)]
        if len(nb) == 0:
            break
        j = int(nb[np.argmax(t_row[nb])])
        cls = int(np.searchsorted(edges, pieces[i], side="left"))
        total = (t_row[i] - t_row[j]) / 100.0
        if pieces[i] <= 1:
            hops[cls].append((total, total, 0.0, 0.0))
        else:
            a = (t_piece[i] - t_row[j]) / 100.0
            b = (t_node[i] - t_piece[i]) / 100.0 if t_node[i] else 0.0
            c = (t_row[i] - max(t_node[i], t_piece[i])) / 100.0
            hops[cls].append((total, a, b, c))
        if t_seen[j]:
            flag_lat.append((t_seen[j] - t_row[j]) / 100.0)
        path_us += total
        nhops += 1
        i = j
print("sum over the %d launches: last row flag - first row flag %.2f ms; the walked chains: %d hops, %.2f ms"
      % ((n + K * batch - 1) // (K * batch), span_us * 1e-3, nhops, path_us * 1e-3))
print("%-16s %6s %9s %9s | %9s %9s %9s   (microseconds: mean; median of the total)" % ("row class", "hops", "total", "median", "wait", "tree", "root"))
for k in range(len(edges)):
    h = np.array(hops[k]).reshape(-1, 4)
    if len(h):
        print("%-16s %6d %9.2f %9.2f | %9.2f %9.2f %9.2f" % (names[k], len(h), h[:, 0].mean(), np.median(h[:, 0]), h[:, 1].mean(), h[:, 2].mean(), h[:, 3].mean()))
if flag_lat:
    fl = np.array(flag_lat)
    print("flag stored -> first seen by a waiter that was already waiting: mean %.2f us, median %.2f us (%d rows)" % (fl.mean(), np.median(fl), len(fl)))

#!/usr/bin/env python3
"""Host-side model of the headline kernel's neighbour-row gathers against the eight XCD L2s (no GPU): one minibatch of RMAT-20 is cut
into the engine's work items (piece_cuts + plan_for's placement: pieces of split rows to the XCD that owns their id eighth, whole rows
anywhere, longest first, 16 items per workgroup, workgroup b on XCD b mod 8), every XCD runs its workgroups in index order `conc` at a
time (their gathers interleaved four rows per item and turn), and each XCD's L2 is an LRU of `cap` 512-byte rows.  Prints the hit rate
of the row gathers and how many times a distinct row is fetched, for the shipped schedule and for variants, so that a schedule is
worth building only if the model says it misses less.  Usage: l2_sim.py [batch_index ...]   (default: minibatches 1, 5, 12 of 16)"""
import os
import sys
from collections import OrderedDict

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

XCDS, IPB, CHUNK, BATCH, U = 8, 16, 128, 65536, 4
rowptr, colids = bench.load_graph(20, 16, 1)
rowptr = rowptr.astype(np.int64)
colids = colids.astype(np.int64)
n = len(rowptr) - 1


def items_of(lo, hi, classes=8, split_min=CHUNK, sub=1, sub_min=0, sub_order=1):
    """-> (pieces per class: list of (first, cnt)), whole rows [(first, cnt)].  A row of more than split_min neighbours is cut every
    CHUNK neighbours and at the boundaries of the `classes` x `sub` id ranges; a piece's class is that of its id range."""
    queues = [[] for _ in range(classes)]
    whole = []
    fine = classes * sub
    for i in range(lo, hi):
        rp, deg = rowptr[i], rowptr[i + 1] - rowptr[i]
        if deg <= split_min:
            whole.append((rp, deg))
            continue
        f = fine if deg > sub_min else classes  # finer cuts only for rows of more than sub_min neighbours
        cls = colids[rp:rp + deg] * f // n
        cut = np.flatnonzero(np.diff(cls)) + 1
        bounds = np.concatenate(([0], cut, [deg]))
        for b, e in zip(bounds[:-1], bounds[1:]):
            for s in range(b, e, CHUNK):
                # sort key: the (ordering) sub-range of the piece's first neighbour
                queues[int(cls[b]) * classes // f].append((rp + s, min(CHUNK, e - s), int(colids[rp + s] * classes * sub_order // n)))
    return queues, whole


def schedule(queues, whole, order="longest", whole_affine=False):
    """plan_for's rounds: one workgroup per class per round; a class's workgroup takes its own pieces while they are at least as long
    as the longest waiting whole row, else whole rows.  -> per XCD the list of workgroups (each a list of (first, cnt))."""
    if order == "longest":
        for q in queues:
            q.sort(key=lambda x: -x[1])
    elif order == "subrange":  # id sub-range major, longest first inside one
        for q in queues:
            q.sort(key=lambda x: (x[2], -x[1]))
    whole = sorted(whole, key=lambda x: -x[1])
    wq = None
    if whole_affine:  # a whole row to the XCD that owns most of its neighbours
        wq = [[] for _ in range(XCDS)]
        for rp, deg in whole:
            if deg == 0:
                wq[0].append((rp, deg))
                continue
            c = np.bincount(colids[rp:rp + deg] * XCDS // n, minlength=XCDS)
            wq[int(np.argmax(c))].append((rp, deg))
    per_xcd = [[] for _ in range(XCDS)]
    pos = [0] * XCDS
    wpos = [0] * XCDS if whole_affine else [0]
    more = True
    while more:
        more = False
        for k in range(XCDS):
            wg = []
            wl = wq[k] if whole_affine else whole
            wi = k if whole_affine else 0
            for _ in range(IPB):
                mine = pos[k] < len(queues[k])
                anyw = wpos[wi] < len(wl)
                if mine and (order == "subrange" or not anyw or queues[k][pos[k]][1] >= wl[wpos[wi]][1]):
                    wg.append(queues[k][pos[k]][:2])
                    pos[k] += 1
                elif anyw:
                    wg.append(wl[wpos[wi]])
                    wpos[wi] += 1
            if wg:
                per_xcd[k].append(wg)
            more = more or pos[k] < len(queues[k])
    # the remaining whole rows, workgroup by workgroup round robin
    k = 0
    if whole_affine:
        for x in range(XCDS):
            rest = wq[x][wpos[x]:]
            for a in range(0, len(rest), IPB):
                per_xcd[x].append(rest[a:a + IPB])
    else:
        rest = whole[wpos[0]:]
        for a in range(0, len(rest), IPB):
            per_xcd[k % XCDS].append(rest[a:a + IPB])
            k += 1
    return per_xcd


def simulate(per_xcd, cap, conc):
    """-> (row reads, hits, distinct rows over all XCDs, fetches)"""
    reads = hits = 0
    for wgs in per_xcd:
        lru = OrderedDict()
        for w0 in range(0, len(wgs), conc):
            window = wgs[w0:w0 + conc]
            longest = max((cnt for wg in window for _, cnt in wg), default=0)
            for g in range(0, longest, U):
                for wg in window:
                    for first, cnt in wg:
                        for j in colids[first + g:first + min(cnt, g + U)]:
                            reads += 1
                            if j in lru:
                                hits += 1
                                lru.move_to_end(j)
                            else:
                                lru[j] = True
                                if len(lru) > cap:
                                    lru.popitem(last=False)
    return reads, hits


def report(name, per_xcd, lo, hi):
    pieces = sum(len(wg) for x in per_xcd for wg in x) - (hi - lo)
    name = "%s [+%d pieces]" % (name, pieces)
    distinct = len(np.unique(colids[rowptr[lo]:rowptr[hi]]))
    out = []
    for cap, conc in ((6144, 160), (1 << 30, 160)):
        reads, hits = simulate(per_xcd, cap, conc)
        out.append("%s %.3f (x%.2f)" % ("inf" if cap > 1 << 20 else "%dk/%d" % (cap >> 10, conc), hits / reads, (reads - hits) / distinct))
    print("  %-66s hit rate (fetches per distinct row): %s" % (name, "  ".join(out)), flush=True)


for bi in [int(x) for x in sys.argv[1:]] or [1, 5, 12]:
    lo, hi = bi * BATCH, min(n, (bi + 1) * BATCH)
    print("minibatch %d: rows [%d, %d), %d neighbour reads, %d distinct rows; L2 model: rows per XCD / workgroups in flight per XCD" % (
        bi, lo, hi, rowptr[hi] - rowptr[lo], len(np.unique(colids[rowptr[lo]:rowptr[hi]]))), flush=True)
    q, w = items_of(lo, hi)
    report("shipped (8 classes, longest first)", schedule(q, w), lo, hi)
    for so in (4, 16):
        q, w = items_of(lo, hi, sub_order=so)
        report("same pieces, ordered by their first neighbour's 1/%d sub-range" % so, schedule(q, w, order="subrange"), lo, hi)
    for sub, sub_min in ((4, 0), (4, 512), (4, 2048), (16, 2048), (16, 8192)):
        q, w = items_of(lo, hi, sub=sub, sub_min=sub_min, sub_order=sub)
        report("rows of > %d neighbours cut at %d sub-ranges, sub-range major" % (max(sub_min, CHUNK), sub), schedule(q, w, order="subrange"), lo, hi)

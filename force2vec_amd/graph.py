"""Graph inputs of the hot path: MatrixMarket ingest (through libf2v's reader, which keeps
the reference's ReadASCII/CSC/CSR semantics) and the synthetic power-law (RMAT) generator
BASELINE.md section 3 prescribes for the roofline configs."""
import ctypes as C

import numpy as np

from . import _lib


def _read_with(fn, path):
    L = _lib.lib()
    n = C.c_uint32()
    nnz = C.c_uint64()
    rp = _lib.u32p()
    ci = _lib.u32p()
    _lib.check(getattr(L, fn)(str(path).encode(), C.byref(n), C.byref(nnz), C.byref(rp), C.byref(ci)))
    try:
        rowptr = np.ctypeslib.as_array(rp, shape=(n.value + 1,)).copy()
        colids = np.ctypeslib.as_array(ci, shape=(max(nnz.value, 1),)).copy()[: nnz.value]
    finally:
        L.f2v_free(rp)
        L.f2v_free(ci)
    return rowptr, colids


def read_mtx(path):
    """-> (rowptr u32[n+1], colids u32[nnz]) as SetInputMatricesAsCSR builds them (sample/commonutility.h:44-54)."""
    return _read_with("f2v_read_mtx", path)


def read_csr_bin(path):
    """Binary CSR cache written by write_csr_bin / `Force2Vec -cache 1` (format in include/f2v.h)."""
    return _read_with("f2v_read_csr_bin", path)


def write_csr_bin(path, rowptr, colids):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.uint32)
    colids = np.ascontiguousarray(colids, dtype=np.uint32)
    _lib.check(_lib.lib().f2v_write_csr_bin(str(path).encode(), rowptr.ctypes.data_as(_lib.u32p), colids.ctypes.data_as(_lib.u32p),
                                           len(rowptr) - 1, len(colids)))


def _symmetric_csr_dedup(n, src, dst):
    """Symmetric CSR of the undirected pairs (src, dst), duplicate pairs MERGED, neighbour ids ascending: scipy's COO -> CSR
    conversion (linear-time bucketing + per-row sort + duplicate merge, all compiled) -- the same arrays the all-numpy path
    (np.unique on the pair keys, then a global key sort) builds, in a fifth of the time.  No self-loops in the input."""
    import scipy.sparse as sp
    r = np.concatenate([src, dst])
    c = np.concatenate([dst, src])
    m = sp.coo_matrix((np.ones(len(r), dtype=np.int8), (r, c)), shape=(n, n)).tocsr()  # tocsr() sums (= merges) duplicates
    del r, c
    m.sort_indices()
    assert m.indptr[-1] == len(m.indices) and m.indptr[-1] < 2**32
    return m.indptr.astype(np.uint32), m.indices.astype(np.uint32)


def csr_from_undirected_edges(n, src, dst):
    """Symmetric CSR (both directions, ascending colids) from an undirected edge list without self-loops."""
    r = np.concatenate([src, dst]).astype(np.int64)
    c = np.concatenate([dst, src]).astype(np.int64)
    key = r * n + c
    key.sort(kind="stable")
    rows = (key // n).astype(np.int64)
    colids = (key % n).astype(np.uint32)
    rowptr = np.zeros(n + 1, dtype=np.int64)
    rowptr[1:] = np.cumsum(np.bincount(rows, minlength=n))
    assert rowptr[-1] == len(colids) and rowptr[-1] < 2**32
    return rowptr.astype(np.uint32), colids


def _rmat_bits(levels, m, seed, a, b, c):
    """(src, dst) bit strings of `m` RMAT edges over `levels` levels, drawn exactly as the serial loop
        rng = default_rng(seed); for level: u = rng.random(m); src = src << 1 | (u >= a+b); dst = dst << 1 | (a <= u < a+b or u >= a+b+c)
    draws them -- level-major, one 64-bit draw per value -- but by edge CHUNKS in parallel threads: PCG64 can jump, so the chunk
    [s, e) of level l starts l*m + s draws into the stream.  -> src, dst, and the generator as the serial loop leaves it."""
    from concurrent.futures import ThreadPoolExecutor
    import os
    src = np.empty(m, dtype=np.int64)
    dst = np.empty(m, dtype=np.int64)
    ab, abc = a + b, a + b + c
    chunk = 1 << 22

    def work(s0):
        e0 = min(s0 + chunk, m)
        sv = np.zeros(e0 - s0, dtype=np.int64)
        dv = np.zeros(e0 - s0, dtype=np.int64)
        for level in range(levels):
            bg = np.random.PCG64(seed)
            bg.advance(level * m + s0)
            u = np.random.Generator(bg).random(e0 - s0)
            sv = (sv << 1) | (u >= ab)
            dv = (dv << 1) | (((u >= a) & (u < ab)) | (u >= abc))
        src[s0:e0] = sv
        dst[s0:e0] = dv

    with ThreadPoolExecutor(max_workers=max(1, min(16, os.cpu_count() or 1))) as pool:
        list(pool.map(work, range(0, m, chunk)))
    bg = np.random.PCG64(seed)
    bg.advance(levels * m)
    return src, dst, np.random.Generator(bg)


def _rmat_pairs(scale, edge_factor, seed, a, b, c):
    """The raw RMAT draws: n * edge_factor (src, dst) pairs over permuted vertex ids, self-loops removed, duplicates kept."""
    n = 1 << scale
    src, dst, rng = _rmat_bits(scale, n * edge_factor, seed, a, b, c)
    perm = rng.permutation(n)
    src, dst = perm[src], perm[dst]
    keep = src != dst
    return n, src[keep], dst[keep]


def rmat_edges(scale, edge_factor=16, seed=1, a=0.57, b=0.19, c=0.19):
    """RMAT (a,b,c,d)=(0.57,0.19,0.19,0.05) edge list of 2^scale vertices, vertex ids permuted,
    self-loops and duplicate undirected pairs removed.  -> (n, src, dst) with src > dst."""
    n, src, dst = _rmat_pairs(scale, edge_factor, seed, a, b, c)
    hi, lo = np.maximum(src, dst), np.minimum(src, dst)
    key = np.unique(hi * n + lo)
    return n, (key // n).astype(np.int64), (key % n).astype(np.int64)


def rmat_edges_n(n, m, seed=1, a=0.57, b=0.19, c=0.19):
    """The same generator for a vertex count that is not a power of two: `m` RMAT edges over 2^ceil(log2 n) ids,
    permuted, folded into [0, n) by `id mod n`, self-loops and duplicate undirected pairs removed."""
    n, src, dst = _rmat_pairs_n(n, m, seed, a, b, c)
    key = np.unique(np.maximum(src, dst) * n + np.minimum(src, dst))
    return n, key // n, key % n


def _rmat_pairs_n(n, m, seed, a, b, c):
    scale = max(1, int(np.ceil(np.log2(n))))
    src, dst, rng = _rmat_bits(scale, m, seed, a, b, c)
    perm = rng.permutation(1 << scale)
    src, dst = perm[src] % n, perm[dst] % n
    keep = src != dst
    return n, src[keep], dst[keep]


# com-Orkut: 3 072 441 vertices, 117 185 083 undirected edges.  The file is not in the container
# (no network), so BASELINE configs[3] runs on a synthetic power-law graph of that size: ORKUT_M generated edges leave
# 117.2 M distinct ones (nnz = 234.4 M directed CSR nonzeros).
ORKUT_N, ORKUT_M = 3072441, 125300000


def orkut_like_csr(seed=1):
    n, s, d = _rmat_pairs_n(ORKUT_N, ORKUT_M, seed, 0.57, 0.19, 0.19)
    return _symmetric_csr_dedup(n, s, d)


def rmat_csr(scale, edge_factor=16, seed=1):
    """The CSR of rmat_edges' graph (tests/test_host_boundary.py checks the two paths against each other)."""
    n, s, d = _rmat_pairs(scale, edge_factor, seed, 0.57, 0.19, 0.19)
    return _symmetric_csr_dedup(n, s, d)


def write_mtx_symmetric(path, n, src, dst):
    """`pattern symmetric` MatrixMarket file (lower triangle, 1-based), the form the reference's datasets use."""
    with open(path, "w") as f:
        f.write("%%MatrixMarket matrix coordinate pattern symmetric\n")
        f.write("%d %d %d\n" % (n, n, len(src)))
        np.savetxt(f, np.stack([src + 1, dst + 1], axis=1), fmt="%d %d")


def edges_from_csr(rowptr, colids):
    """Lower-triangle edge list (src > dst) of a symmetric CSR, duplicates kept."""
    n = len(rowptr) - 1
    rows = np.repeat(np.arange(n, dtype=np.int64), np.diff(rowptr.astype(np.int64)))
    cols = colids.astype(np.int64)
    m = rows > cols
    return rows[m], cols[m]

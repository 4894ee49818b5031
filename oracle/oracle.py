"""ctypes front-end of the CPU parity oracle (oracle/f2v_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under force2vec_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORDER_REF = 0   # reference's sequential d-sum (bit-pinned against oracle/_ref)
ORDER_TREE = 1  # canonical wavefront tree order (what the HIP kernels compute)

_lib = None


def build():
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    subprocess.check_call(["make", "-s", "-C", HERE, "all"])


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-s", "-C", HERE, "liboracle.so"])
        L = C.CDLL(so)
        u32p = C.POINTER(C.c_uint32)
        f32p = C.POINTER(C.c_float)
        L.orc_rng_new.restype = C.c_void_p
        L.orc_rng_new.argtypes = [C.c_uint]
        L.orc_rng_free.argtypes = [C.c_void_p]
        L.orc_rand.restype = C.c_int
        L.orc_rand.argtypes = [C.c_void_p]
        L.orc_rand_index.restype = C.c_uint32
        L.orc_rand_index.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
        L.orc_init_embeddings.argtypes = [C.c_void_p, f32p, C.c_uint32, C.c_uint32, C.c_int]
        L.orc_read_mtx.restype = C.c_int
        L.orc_read_mtx.argtypes = [C.c_char_p, u32p, C.POINTER(C.c_uint64), C.POINTER(u32p), C.POINTER(u32p)]
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_sm_table.argtypes = [f32p]
        L.orc_sm_table_as_compiled.restype = C.c_int
        L.orc_sm_table_as_compiled.argtypes = [f32p]
        L.orc_set_sm_table.argtypes = [f32p]
        L.orc_set_fanin.argtypes = [C.c_uint32]
        L.orc_fast_sm.restype = C.c_float
        L.orc_fast_sm.argtypes = [f32p, C.c_float]
        L.orc_minibatch.restype = C.c_int
        L.orc_minibatch.argtypes = [C.c_int, C.c_int, u32p, u32p, C.c_uint32, C.c_uint32, f32p,
                                    C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, C.c_uint32,
                                    C.c_float, u32p, C.c_int, C.c_uint32]
        L.orc_generate_walks.argtypes = [C.c_void_p, u32p, u32p, C.c_uint32, C.c_uint64, u32p]
        L.orc_train.restype = C.c_int
        L.orc_train.argtypes = [C.c_int, C.c_int, u32p, u32p, C.c_uint32, C.c_uint64, C.c_uint32, f32p,
                                C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_int,
                                C.c_uint32, C.c_int]
        L.orc_row.restype = C.c_int
        L.orc_set_class_cut.argtypes = [C.c_uint32]
        L.orc_row.argtypes = [C.c_int, u32p, u32p, C.c_uint32, C.c_uint32, f32p, C.c_uint32, u32p, C.c_uint32,
                              C.c_float, u32p, C.c_int, C.c_uint32, f32p]
        L.orc_write_embd.restype = C.c_int
        L.orc_write_embd.argtypes = [C.c_char_p, f32p, C.c_uint32, C.c_uint32]
        _lib = L
    return _lib


def _u32(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32)) if a is not None else None


def _f32(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Rng:
    """glibc rand() restatement (state after srand(seed))."""

    def __init__(self, seed=1):
        self._h = lib().orc_rng_new(seed)

    def rand(self):
        return lib().orc_rand(self._h)

    def rand_index(self, max_num, min_num=0):
        return lib().orc_rand_index(self._h, max_num, min_num)

    def init_embeddings(self, n, d, kind):
        X = np.empty((n, d), dtype=np.float32)
        lib().orc_init_embeddings(self._h, _f32(X), n, d, kind)
        return X

    def __del__(self):
        try:
            lib().orc_rng_free(self._h)
        except Exception:
            pass


def read_mtx(path):
    n = C.c_uint32()
    nnz = C.c_uint64()
    rp = C.POINTER(C.c_uint32)()
    ci = C.POINTER(C.c_uint32)()
    rc = lib().orc_read_mtx(path.encode(), C.byref(n), C.byref(nnz), C.byref(rp), C.byref(ci))
    if rc != 0:
        raise IOError("orc_read_mtx(%s) -> %d" % (path, rc))
    rowptr = np.ctypeslib.as_array(rp, shape=(n.value + 1,)).copy()
    colids = np.ctypeslib.as_array(ci, shape=(max(nnz.value, 1),)).copy()[: nnz.value]
    lib().orc_free(rp)
    lib().orc_free(ci)
    return rowptr.astype(np.uint32), colids.astype(np.uint32)


def sm_table():
    t = np.empty(2048, dtype=np.float32)
    lib().orc_sm_table(_f32(t))
    return t


def sm_table_as_compiled():
    """The reference's table as g++ -O3 -ffast-math compiled it, on THIS host (pinning only)."""
    t = np.empty(2048, dtype=np.float32)
    rc = lib().orc_sm_table_as_compiled(_f32(t))
    return t, rc == 0


def set_sm_table(table=None):
    """Test hook: table used by the option-6/7 routines (None = source-level default)."""
    if table is None:
        lib().orc_set_sm_table(None)
    else:
        table = np.ascontiguousarray(table, dtype=np.float32)
        assert table.shape == (2048,)
        lib().orc_set_sm_table(_f32(table))


def set_fanin(fanin=32):
    """Hub combine fan-in of the chunked order (0 = one sequential pass); the engine's "hub_fanin"."""
    lib().orc_set_fanin(fanin)


def set_class_cut(classes=8):
    """Pieces of a split row also end where its neighbour ids cross into the next of `classes` equal id ranges (the engine's
    "class_cut", default on = 8); 0 = every `chunk` neighbours only."""
    lib().orc_set_class_cut(classes)


def minibatch(option, rowptr, colids, X, lo, hi, sample_ids, ns, lr, bs_mode=0, walks=None,
              order=ORDER_TREE, chunk=0, row_lo=None, row_hi=None):
    """Update rows [row_lo,row_hi) of minibatch [lo,hi) of X in place."""
    n, d = X.shape
    assert X.dtype == np.float32 and X.flags.c_contiguous
    sample_ids = np.ascontiguousarray(sample_ids, dtype=np.uint32)
    need = (hi - lo) + ns - 1 if bs_mode else ns
    assert len(sample_ids) >= need
    rc = lib().orc_minibatch(option, bs_mode, _u32(rowptr), _u32(colids), n, d, _f32(X), lo, hi,
                             lo if row_lo is None else row_lo, hi if row_hi is None else row_hi,
                             _u32(sample_ids), ns, lr, _u32(walks), order, chunk)
    if rc != 0:
        raise RuntimeError("orc_minibatch -> %d" % rc)


def generate_walks(rng, rowptr, colids):
    n = len(rowptr) - 1
    walks = np.empty(n * 5, dtype=np.uint32)
    lib().orc_generate_walks(rng._h, _u32(rowptr), _u32(colids), n, len(colids), _u32(walks))
    return walks


def train(option, rowptr, colids, dim, iters, batch, ns=5, lr=0.02, bs_mode=0, seed=1,
          order=ORDER_REF, chunk=0, X0=None, rng=None):
    """Whole run as the reference's AlgoForce2Vec* methods; returns the N x D embeddings."""
    n = len(rowptr) - 1
    rng = rng or Rng(seed)
    if X0 is None:
        X = np.empty((n, dim), dtype=np.float32)
        do_init = 1
    else:
        X = np.array(X0, dtype=np.float32, order="C")
        do_init = 0
    rc = lib().orc_train(option, bs_mode, _u32(rowptr), _u32(colids), n, len(colids), dim, _f32(X),
                         rng._h, iters, batch, ns, lr, order, chunk, do_init)
    if rc != 0:
        raise RuntimeError("orc_train -> %d" % rc)
    return X


def row(option, rowptr, colids, X, i, sample_ids, lr, walks=None, order=ORDER_TREE, chunk=0):
    n, d = X.shape
    sample_ids = np.ascontiguousarray(sample_ids, dtype=np.uint32)
    out = np.empty(d, dtype=np.float32)
    rc = lib().orc_row(option, _u32(rowptr), _u32(colids), n, d, _f32(X), i, _u32(sample_ids),
                       len(sample_ids), lr, _u32(walks), order, chunk, _f32(out))
    if rc != 0:
        raise RuntimeError("orc_row -> %d" % rc)
    return out


def write_embd(path, X):
    X = np.ascontiguousarray(X, dtype=np.float32)
    rc = lib().orc_write_embd(path.encode(), _f32(X), X.shape[0], X.shape[1])
    if rc != 0:
        raise IOError("orc_write_embd(%s) -> %d" % (path, rc))


def read_embd(path):
    """Reader for the text .embd format (1-based ids, performancescores/runnodeclassclust.py:57-79)."""
    with open(path) as f:
        n, d = (int(t) for t in f.readline().split())
        X = np.zeros((n, d), dtype=np.float32)
        for line in f:
            p = line.split()
            if not p:
                continue
            X[int(p[0]) - 1] = np.array(p[1:1 + d], dtype=np.float32)
    return X


def read_embd_roundtrip(X):
    """X as a reader of the 6-significant-digit .embd text would see it."""
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".embd") as f:
        write_embd(f.name, X)
        return read_embd(f.name)


# ---------------------------------------------------------------------------------------
# The genuine reference binary (oracle/_ref), when present
# ---------------------------------------------------------------------------------------
def ref_binary(avx512=False):
    p = os.path.join(HERE, "_ref", "Force2Vec_avx512" if avx512 else "Force2Vec")
    return p if os.path.exists(p) else None


def run_reference(mtx, outdir, option, iters, batch, dim, ns=5, lr=0.02, bs=0, threads=1, avx512=False):
    """Run the genuine reference CLI; returns (embd_path, stdout)."""
    exe = ref_binary(avx512)
    if exe is None:
        raise FileNotFoundError("oracle/_ref binary missing (run oracle/build_ref.sh where /root/reference exists)")
    if not outdir.endswith("/"):
        outdir += "/"
    cmd = [exe, "-input", mtx, "-output", outdir, "-iter", str(iters), "-batch", str(batch), "-dim", str(dim),
           "-nsamples", str(ns), "-lr", repr(lr), "-option", str(option), "-bs", str(bs), "-threads", str(threads)]
    out = subprocess.run(cmd, cwd=outdir, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, check=True).stdout
    path = None
    for line in out.splitlines():
        if line.startswith("Creating output file in following directory:"):
            path = line.split(":", 1)[1].strip()
    return path, out

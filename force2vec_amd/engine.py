"""Host-side mirror of the reference's operator interface over libf2v (include/f2v.h).

`Engine` is the thin handle wrapper; `algorithms` mirrors `class algorithms` of
sample/algorithms.h:51-137 -- same constructor meaning, same AlgoForce2Vec* method names and
(ITERATIONS, NUMOFTHREADS, BATCHSIZE, ns, lr) arguments, result [seconds], side effect = the
.embd file -- so the parity tests read like a user of the reference.  All force arithmetic
runs in the HIP kernels; nothing here (or anywhere in this package) computes on the CPU."""
import ctypes as C
import os
import time

import numpy as np

from . import _lib
from ._lib import INIT_SYMMETRIC, INIT_UNIT, check


def _u32(a):
    return a.ctypes.data_as(_lib.u32p)


def _f32(a):
    return a.ctypes.data_as(_lib.f32p)


class Engine:
    """One HBM-resident graph + embedding matrix on one MI355X."""

    def __init__(self, rowptr, colids, dim, device=0, selftest=False):
        # selftest: bind this engine to libf2v_selftest.so (include/f2v_test.h), e.g. for fault injection
        self._L = _lib.selftest_lib() if selftest else _lib.lib()
        rowptr = np.ascontiguousarray(rowptr, dtype=np.uint32)
        colids = np.ascontiguousarray(colids, dtype=np.uint32)
        self.n = len(rowptr) - 1
        self.nnz = int(rowptr[-1])
        self.dim = int(dim)
        self.rowptr, self.colids = rowptr, colids
        h = C.c_void_p()
        self._ck(self._L.f2v_create(_u32(rowptr), _u32(colids), self.n, self.nnz, self.dim, device, C.byref(h)))
        self._h = h

    def _ck(self, rc):
        check(rc, self._L)  # the error text lives in the library that returned the code

    def close(self):
        if getattr(self, "_h", None):
            self._L.f2v_destroy(self._h)
            self._h = None

    __del__ = close

    # -- rand() stream / embeddings ---------------------------------------------------------
    def srand(self, seed=1):
        self._ck(self._L.f2v_srand(self._h, seed))

    def init_embeddings(self, kind):
        self._ck(self._L.f2v_init_embeddings(self._h, kind))

    def set_embeddings(self, X):
        X = np.ascontiguousarray(X, dtype=np.float32)
        assert X.shape == (self.n, self.dim)
        self._ck(self._L.f2v_set_embeddings(self._h, _f32(X)))

    def get_embeddings(self):
        X = np.empty((self.n, self.dim), dtype=np.float32)
        self._ck(self._L.f2v_get_embeddings(self._h, _f32(X)))
        return X

    def rand_index(self, max_num, min_num=0):
        out = C.c_uint32()
        self._ck(self._L.f2v_rand_index(self._h, max_num, min_num, C.byref(out)))
        return out.value

    def draw_samples(self, max_num, count, keep=None):
        """`count` randIndex(max_num, 0) draws from the handle's rand() stream; the first `keep` are returned."""
        keep = count if keep is None else keep
        out = np.empty(max(keep, 1), dtype=np.uint32)
        self._ck(self._L.f2v_rand_indices(self._h, max_num, 0, count, keep, _u32(out)))
        return out[:keep]

    def set_param(self, name, value):
        self._ck(self._L.f2v_set_param(self._h, name.encode(), int(value)))

    def get_param(self, name):
        v = C.c_int64()
        self._ck(self._L.f2v_get_param(self._h, name.encode(), C.byref(v)))
        return v.value

    # -- training -----------------------------------------------------------------------------
    def train(self, option, iters, batch, ns=5, lr=0.02, bs_mode=0):
        """-> device seconds of the epoch loop."""
        sec = C.c_double()
        self._ck(self._L.f2v_train(self._h, option, iters, batch, ns, lr, bs_mode, C.byref(sec)))
        return sec.value

    # -- multi-GPU push exchange over xGMI (include/f2v.h) ----------------------------------------
    def push_export(self):
        """-> bytes: this rank's IPC handles (gather them from all ranks, then push_attach)."""
        buf = C.create_string_buffer(_lib.PUSH_EXPORT_BYTES)
        self._ck(self._L.f2v_push_export(self._h, buf))
        return buf.raw

    def push_attach(self, rank, world, exports):
        blob = b"".join(exports)
        assert len(blob) == world * _lib.PUSH_EXPORT_BYTES
        self._ck(self._L.f2v_push_attach(self._h, rank, world, C.c_char_p(blob)))

    def push_selftest(self):
        self._ck(self._L.f2v_push_selftest(self._h))

    def push_detach(self):
        self._ck(self._L.f2v_push_detach(self._h))

    def push_stats(self):
        a, b = C.c_uint64(), C.c_uint64()
        self._ck(self._L.f2v_push_stats(self._h, C.byref(a), C.byref(b)))
        return {"rows_pushed": a.value, "rows_allgather": b.value}

    def train_sharded(self, option, iters, batch, ns=5, lr=0.02, bs_mode=0):
        """f2v_train over the attached ranks -> device seconds of the epoch loop (exchange included)."""
        sec = C.c_double()
        self._ck(self._L.f2v_train_sharded(self._h, option, iters, batch, ns, lr, bs_mode, C.byref(sec)))
        return sec.value

    def minibatch_step(self, option, batch_lo, batch_hi, sample_ids, ns, lr, bs_mode=0, row_lo=None, row_hi=None):
        ids = np.ascontiguousarray(sample_ids, dtype=np.uint32)
        self._ck(self._L.f2v_minibatch_step(self._h, option, batch_lo, batch_hi,
                                         batch_lo if row_lo is None else row_lo, batch_hi if row_hi is None else row_hi,
                                         _u32(ids), len(ids), ns, lr, bs_mode))

    def upload_sample_ids(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        self._ck(self._L.f2v_upload_sample_ids(self._h, _u32(ids), len(ids)))

    def minibatch_step_at(self, option, batch_lo, batch_hi, ids_offset, ns, lr, bs_mode=0, row_lo=None, row_hi=None):
        self._ck(self._L.f2v_minibatch_step_at(self._h, option, batch_lo, batch_hi,
                                            batch_lo if row_lo is None else row_lo, batch_hi if row_hi is None else row_hi,
                                            ids_offset, ns, lr, bs_mode))

    def flush(self):
        self._ck(self._L.f2v_flush(self._h))

    def synchronize(self):
        self._ck(self._L.f2v_synchronize(self._h))

    def set_walks(self, walks):
        w = np.ascontiguousarray(walks, dtype=np.uint32)
        assert w.size == 5 * self.n
        self._ck(self._L.f2v_set_walks(self._h, _u32(w)))

    def generate_walks(self):
        w = np.empty(5 * self.n, dtype=np.uint32)
        self._ck(self._L.f2v_generate_walks(self._h, _u32(w)))
        return w

    def stage_reserve(self, rows):
        self._ck(self._L.f2v_stage_reserve(self._h, rows))

    def stage_read(self, row_lo, row_hi):
        out = np.empty((row_hi - row_lo, self.dim), dtype=np.float32)
        self._ck(self._L.f2v_stage_read(self._h, row_lo, row_hi, _f32(out)))
        return out

    def stage_write(self, row_lo, row_hi, rows):
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        assert rows.shape == (row_hi - row_lo, self.dim)
        self._ck(self._L.f2v_stage_write(self._h, row_lo, row_hi, _f32(rows)))

    def rows_read(self, ids):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        out = np.empty((len(ids), self.dim), dtype=np.float32)
        self._ck(self._L.f2v_rows_read(self._h, _u32(ids), len(ids), _f32(out)))
        return out

    def rows_write(self, ids, rows):
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        rows = np.ascontiguousarray(rows, dtype=np.float32)
        assert rows.shape == (len(ids), self.dim)
        self._ck(self._L.f2v_rows_write(self._h, _u32(ids), len(ids), _f32(rows)))

    def embeddings_device_ptr(self):
        p = C.c_uint64()
        self._ck(self._L.f2v_embeddings_device_ptr(self._h, C.byref(p)))
        return p.value

    def stage_device_ptr(self):
        p = C.c_uint64()
        cap = C.c_uint32()
        self._ck(self._L.f2v_stage_device_ptr(self._h, C.byref(p), C.byref(cap)))
        return p.value, cap.value

    def stream(self):
        s = C.c_uint64()
        self._ck(self._L.f2v_stream(self._h, C.byref(s)))
        return s.value

    def train_marks(self):
        """Device seconds from the start of the last f2v_train's epoch loop to every "epoch_marks"-th epoch's end."""
        n = C.c_uint32()
        self._ck(self._L.f2v_train_marks(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=np.float64)
        self._ck(self._L.f2v_train_marks(self._h, out.ctypes.data_as(C.POINTER(C.c_double)), n.value, C.byref(n)))
        return out[: n.value]

    def stats(self):
        s = _lib.Stats()
        self._ck(self._L.f2v_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}


def shard_bounds_balanced(rowptr, lo, hi, world):
    """f2v_shard_bounds (host only): uint32[world+1], slice r = rows [b[r], b[r+1]) of minibatch [lo,hi), balanced by degree + 4."""
    rp = np.ascontiguousarray(rowptr, dtype=np.uint32)
    out = np.zeros(world + 1, dtype=np.uint32)
    check(_lib.lib().f2v_shard_bounds(_u32(rp), lo, hi, world, _u32(out)))
    return out


def push_masks(rowptr, colids, batch, world, sample_ids=()):
    """f2v_push_masks (host only): uint32[n], bit r = rank r reads the row without owning it."""
    rp = np.ascontiguousarray(rowptr, dtype=np.uint32)
    ci = np.ascontiguousarray(colids, dtype=np.uint32)
    ids = np.ascontiguousarray(sample_ids, dtype=np.uint32)
    out = np.zeros(len(rp) - 1, dtype=np.uint32)
    check(_lib.lib().f2v_push_masks(_u32(rp), _u32(ci), len(rp) - 1, batch, world, _u32(ids), len(ids), _u32(out)))
    return out


def write_embd(path, X):
    X = np.ascontiguousarray(X, dtype=np.float32)
    check(_lib.lib().f2v_write_embd(str(path).encode(), _f32(X), X.shape[0], X.shape[1]))


def write_embd_bin(path, X):
    """Raw fp32 N x D file (the scorers' readBinEmbeddings format, runnodeclassclust.py:81-100)."""
    X = np.ascontiguousarray(X, dtype=np.float32)
    check(_lib.lib().f2v_write_embd_bin(str(path).encode(), _f32(X), X.shape[0], X.shape[1]))


def read_embd(path):
    """Text .embd (writeToFile's format) -> float32 [N, D]."""
    L = _lib.lib()
    n, d, x = C.c_uint32(), C.c_uint32(), _lib.f32p()
    check(L.f2v_read_embd(str(path).encode(), C.byref(n), C.byref(d), C.byref(x)))
    try:
        return np.ctypeslib.as_array(x, shape=(n.value, d.value)).copy()
    finally:
        L.f2v_free(x)


def read_embd_bin(path, n, dim):
    """Raw fp32 N x D file (write_embd_bin) -> float32 [N, D]."""
    X = np.empty((n, dim), dtype=np.float32)
    check(_lib.lib().f2v_read_embd_bin(str(path).encode(), n, dim, _f32(X)))
    return X


def output_name(input_path, outdir, option, bs_mode, batch, dim, iters, ns):
    buf = C.create_string_buffer(4096)
    check(_lib.lib().f2v_output_name(str(input_path).encode(), str(outdir).encode(), option, bs_mode, batch, dim, iters, ns, buf, len(buf)))
    return buf.value.decode()


def sm_table():
    t = np.empty(2048, dtype=np.float32)
    check(_lib.lib().f2v_sm_table(_f32(t)))
    return t


class algorithms:
    """Mirror of `class algorithms` (sample/algorithms.h:51-137) on the GPU engine."""

    def __init__(self, graph, input_path="", outputdir="", dim=128, gamma=1.0, bsize=384, device=0):
        rowptr, colids = graph
        self.engine = Engine(rowptr, colids, dim, device)
        self.DIM = dim
        self.filename = input_path
        self.outputdir = outputdir
        self.gpu_train_seconds = 0.0
        self.nCoordinates = None  # filled after a run (host copy of the HBM matrix)

    def srand(self, seed=1):  # Test/Force2Vec.cpp:126
        self.engine.srand(seed)

    def _run(self, option, bs, ITER, BATCH, ns, lr, write=True):
        t0 = time.perf_counter()
        self.engine.init_embeddings(INIT_SYMMETRIC if option in (5, 8, 11) else INIT_UNIT)
        self.gpu_train_seconds = self.engine.train(option, ITER, BATCH, ns, lr, bs)
        sec = time.perf_counter() - t0
        self.nCoordinates = self.engine.get_embeddings()
        if write and self.filename:
            self.writeToFile(output_name(self.filename, self.outputdir, option, bs, BATCH, self.DIM, ITER, ns))
        return [sec]

    def AlgoForce2VecNS(self, ITERATIONS, NUMOFTHREADS, BATCHSIZE, ns, lr):
        return self._run(5, 0, ITERATIONS, BATCHSIZE, ns, lr)

    def AlgoForce2VecNSBS(self, ITERATIONS, NUMOFTHREADS, BATCHSIZE, ns, lr):
        return self._run(5, 1, ITERATIONS, BATCHSIZE, ns, lr)

    def AlgoForce2VecNSRW(self, ITERATIONS, NUMOFTHREADS, BATCHSIZE, ns, lr):
        return self._run(6, 0, ITERATIONS, BATCHSIZE, ns, lr)

    def AlgoForce2VecNSRWBS(self, ITERATIONS, NUMOFTHREADS, BATCHSIZE, ns, lr):
        return self._run(6, 1, ITERATIONS, BATCHSIZE, ns, lr)

    def AlgoForce2VecNSRWEFF(self, ITERATIONS, NUMOFTHREADS, BATCHSIZE, ns, lr):
        return self._run(7, 0, ITERATIONS, BATCHSIZE, ns, lr)

    def AlgoForce2VecNS_SREAL_D128_AVXZ(self, ITERATIONS, NUMOFTHREADS, BATCHSIZE, ns, lr):
        return self._run(8, 0, ITERATIONS, BATCHSIZE, ns, lr)

    def AlgoForce2VecNSRW_SREAL_D128_AVXZ(self, ITERATIONS, NUMOFTHREADS, BATCHSIZE, ns, lr):
        return self._run(9, 0, ITERATIONS, BATCHSIZE, ns, lr)

    def AlgoForce2VecNSRWEFF_SREAL_D128_AVXZ(self, ITERATIONS, NUMOFTHREADS, BATCHSIZE, ns, lr):
        return self._run(10, 0, ITERATIONS, BATCHSIZE, ns, lr)

    def AlgoForce2VecNSLB_SREAL_D128_AVXZ(self, ITERATIONS, NUMOFTHREADS, BATCHSIZE, ns, lr):
        return self._run(11, 0, ITERATIONS, BATCHSIZE, ns, lr)

    def writeToFile(self, path):
        print("Creating output file in following directory:" + path)
        write_embd(path, self.nCoordinates)
        self.last_output = path

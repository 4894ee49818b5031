"""ctypes binding of libf2v.so (include/f2v.h).  The library is built in-tree by `make`
(or __graft_entry__.build()); there is no fallback implementation -- a missing library
is an ImportError, a missing GPU is F2V_ENODEV from f2v_create."""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libf2v.so")
SELFTEST_LIB_PATH = os.path.join(HERE, "libf2v_selftest.so")

F2V_OK, F2V_EINVAL, F2V_ENODEV, F2V_ENOMEM, F2V_EIO, F2V_ESTATE = 0, -1, -2, -3, -4, -5
INIT_SYMMETRIC, INIT_UNIT = 0, 1

u32p = C.POINTER(C.c_uint32)
f32p = C.POINTER(C.c_float)


class Stats(C.Structure):
    _fields_ = [("step_launches", C.c_uint64), ("rows", C.c_uint64), ("nnz", C.c_uint64),
                ("algorithmic_bytes", C.c_uint64), ("device_seconds", C.c_double),
                ("hub_rows", C.c_uint64), ("hub_chunks", C.c_uint64), ("compulsory_bytes", C.c_uint64),
                ("snapshot_seconds", C.c_double), ("recoveries", C.c_uint64), ("recovered", C.c_uint32), ("merge_finalize", C.c_uint32)]


# every entry point declared in include/f2v.h: name -> (restype, argtypes)
SIGNATURES = {
    "f2v_last_error": (C.c_char_p, []),
    "f2v_version": (C.c_char_p, []),
    "f2v_create": (C.c_int, [u32p, u32p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "f2v_destroy": (C.c_int, [C.c_void_p]),
    "f2v_srand": (C.c_int, [C.c_void_p, C.c_uint32]),
    "f2v_init_embeddings": (C.c_int, [C.c_void_p, C.c_int]),
    "f2v_set_embeddings": (C.c_int, [C.c_void_p, f32p]),
    "f2v_get_embeddings": (C.c_int, [C.c_void_p, f32p]),
    "f2v_set_param": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "f2v_get_param": (C.c_int, [C.c_void_p, C.c_char_p, C.POINTER(C.c_int64)]),
    "f2v_train": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.POINTER(C.c_double)]),
    "f2v_minibatch_step": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, u32p, C.c_uint32, C.c_uint32, C.c_float, C.c_int]),
    "f2v_upload_sample_ids": (C.c_int, [C.c_void_p, u32p, C.c_uint64]),
    "f2v_minibatch_step_at": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint32, C.c_float, C.c_int]),
    "f2v_flush": (C.c_int, [C.c_void_p]),
    "f2v_set_walks": (C.c_int, [C.c_void_p, u32p]),
    "f2v_generate_walks": (C.c_int, [C.c_void_p, u32p]),
    "f2v_rand_index": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, u32p]),
    "f2v_rand_indices": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, u32p]),
    "f2v_stage_device_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), u32p]),
    "f2v_stage_read": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, f32p]),
    "f2v_stage_write": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, f32p]),
    "f2v_stage_reserve": (C.c_int, [C.c_void_p, C.c_uint32]),
    "f2v_rows_read": (C.c_int, [C.c_void_p, u32p, C.c_uint32, f32p]),
    "f2v_rows_write": (C.c_int, [C.c_void_p, u32p, C.c_uint32, f32p]),
    "f2v_embeddings_device_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "f2v_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64)]),
    "f2v_synchronize": (C.c_int, [C.c_void_p]),
    "f2v_get_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "f2v_train_marks": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.c_uint32, u32p]),
    "f2v_push_export": (C.c_int, [C.c_void_p, C.c_void_p]),
    "f2v_push_attach": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]),
    "f2v_push_selftest": (C.c_int, [C.c_void_p]),
    "f2v_push_detach": (C.c_int, [C.c_void_p]),
    "f2v_train_sharded": (C.c_int, [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_int, C.POINTER(C.c_double)]),
    "f2v_shard_bounds": (C.c_int, [u32p, C.c_uint32, C.c_uint32, C.c_uint32, u32p]),
    "f2v_push_masks": (C.c_int, [u32p, u32p, C.c_uint32, C.c_uint32, C.c_uint32, u32p, C.c_uint64, u32p]),
    "f2v_push_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "f2v_read_mtx": (C.c_int, [C.c_char_p, u32p, C.POINTER(C.c_uint64), C.POINTER(u32p), C.POINTER(u32p)]),
    "f2v_free": (None, [C.c_void_p]),
    "f2v_write_embd": (C.c_int, [C.c_char_p, f32p, C.c_uint32, C.c_uint32]),
    "f2v_write_csr_bin": (C.c_int, [C.c_char_p, u32p, u32p, C.c_uint32, C.c_uint64]),
    "f2v_read_csr_bin": (C.c_int, [C.c_char_p, u32p, C.POINTER(C.c_uint64), C.POINTER(u32p), C.POINTER(u32p)]),
    "f2v_write_embd_bin": (C.c_int, [C.c_char_p, f32p, C.c_uint32, C.c_uint32]),
    "f2v_read_embd": (C.c_int, [C.c_char_p, u32p, u32p, C.POINTER(f32p)]),
    "f2v_read_embd_bin": (C.c_int, [C.c_char_p, C.c_uint32, C.c_uint32, f32p]),
    "f2v_output_name": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t]),
    "f2v_rng_create": (C.c_void_p, [C.c_uint32]),
    "f2v_rng_destroy": (None, [C.c_void_p]),
    "f2v_rng_next": (C.c_int, [C.c_void_p]),
    "f2v_rng_jump": (None, [C.c_void_p, C.c_uint64]),
    "f2v_rng_fill": (C.c_int, [C.c_void_p, f32p, C.c_uint64, C.c_int]),
    "f2v_rng_walks": (C.c_int, [C.c_void_p, u32p, u32p, C.c_uint32, C.c_uint64, u32p]),
    "f2v_sm_table": (C.c_int, [f32p]),
    "f2v_diag_ipc_preflight": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32, C.c_char_p, C.c_uint64, C.c_double]),
    "f2v_diag_stream_copy": (C.c_int, [C.c_int, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
    "f2v_diag_gather_rate": (C.c_int, [C.c_int, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
}

# include/f2v_test.h: only in libf2v_selftest.so (the same sources built with -DF2V_TEST_HOOKS), for tests/ and tools/
TEST_SIGNATURES = {
    "f2v_test_push_attach_local": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "f2v_test_gather_calibration": (C.c_int, [C.c_int, C.c_uint32, C.c_uint32]),
    "f2v_test_wave_reduce": (C.c_int, [C.c_int, f32p, C.c_uint32, C.c_uint32, f32p]),
    "f2v_test_withhold_flag": (C.c_int, [C.c_void_p, C.c_uint32]),
    "f2v_test_chain_nowait": (C.c_int, [C.c_void_p, C.c_int]),
    "f2v_test_withhold_row": (C.c_int, [C.c_void_p, C.c_uint32]),
    "f2v_test_stamps": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]),
    "f2v_test_xcd_times": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint64)]),
    "f2v_test_plan_gather": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]),
    "f2v_test_wide_plan_check": (C.c_int, [u32p, u32p, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_int64), C.c_uint32, C.POINTER(C.c_uint64)]),
}

PUSH_EXPORT_BYTES = 256  # F2V_PUSH_EXPORT_BYTES
PUSH_MAX_RANKS = 8       # F2V_PUSH_MAX_RANKS

_lib = None
_selftest = None


class F2VError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libf2v error %d: %s" % (code, msg))
        self.code = code


def lib():
    global _lib
    if _lib is None:
        path = os.environ.get("F2V_LIBRARY", LIB_PATH)  # A/B runs of two builds of the library
        if not os.path.exists(path):
            raise ImportError("%s is missing: build it with `make` (hipcc --offload-arch=gfx950); "
                              "force2vec_amd has no fallback implementation" % path)
        L = C.CDLL(path)
        for name, (res, args) in SIGNATURES.items():
            if path != LIB_PATH and not hasattr(L, name):
                continue  # an older build in an A/B run
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def selftest_lib():
    """The self-test build (fault injection, look-inside hooks): a library of its own, with its own handles."""
    global _selftest
    if _selftest is None:
        path = os.environ.get("F2V_SELFTEST_LIBRARY", SELFTEST_LIB_PATH)  # A/B runs of two self-test builds
        if not os.path.exists(path):
            raise ImportError("%s is missing: build it with `make`" % path)
        L = C.CDLL(path)
        for table in (SIGNATURES, TEST_SIGNATURES):
            for name, (res, args) in table.items():
                fn = getattr(L, name)
                fn.restype = res
                fn.argtypes = args
        _selftest = L
    return _selftest


def check(rc, L=None):
    if rc != F2V_OK:
        raise F2VError(rc, (L or lib()).f2v_last_error().decode(errors="replace"))

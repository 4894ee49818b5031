"""CPU tests of the host side of the drop-in boundary (no GPU): the C ABI surface, the libc
rand() restatement, MatrixMarket ingest, the .embd writer and naming, the CLI's argument
behaviour.  libf2v.so must load without a GPU; compute entry points must refuse to run."""
import ctypes as C
import gzip
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLD, ROOT, golden_graph_path
from oracle import oracle as O

import force2vec_amd as F
from force2vec_amd import _lib


def _declared(header):
    hdr = open(os.path.join(ROOT, "include", header)).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return set(re.findall(r"\b(f2v_[a-z_0-9]+)\s*\(", hdr))


def test_abi_exports_every_declared_symbol():
    declared = _declared("f2v.h")
    assert len(declared) >= 30
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), name
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    assert b"gfx950" in _lib.lib().f2v_version()
    # the self-test hooks of include/f2v_test.h exist in the self-test build only, never in the product library
    hooks = _declared("f2v_test.h")
    assert hooks == set(_lib.TEST_SIGNATURES) and len(hooks) >= 4
    T = C.CDLL(_lib.SELFTEST_LIB_PATH)
    for name in hooks:
        assert hasattr(T, name) and not hasattr(L, name), name
    for name in declared:
        assert hasattr(T, name), name


def test_dynamic_symbol_table_is_the_abi_and_nothing_else():
    """libf2v.so is linked with hidden visibility, -Bsymbolic and a version script (Makefile: LIBFLAGS): its dynamic symbol table
    is exactly include/f2v.h, the self-test build's exactly f2v.h + f2v_test.h.  Round 3 exported 465 more -- among them every
    kernel's host-side handle variable, weak: two builds of the library in one process then shared ONE handle address for two
    different code objects (the `address 0x1000` GPU fault of the CLI under LD_PRELOAD=libf2v_selftest.so)."""
    def exported(path):
        out = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        return {ln.split()[-1] for ln in out.splitlines() if ln.strip()}
    assert exported(_lib.LIB_PATH) == _declared("f2v.h")
    assert exported(_lib.SELFTEST_LIB_PATH) == _declared("f2v.h") | _declared("f2v_test.h")
    # the two builds' kernels are different symbols altogether (inline namespace f2v::selftest): no stub or handle can be shared
    names = subprocess.run(["nm", "-C", _lib.SELFTEST_LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "f2v::selftest::qstep_kernel<" in names and "f2v::qstep_kernel<" not in names.replace("f2v::selftest::", "")
    names = subprocess.run(["nm", "-C", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "f2v::qstep_kernel<" in names and "f2v::selftest::" not in names


def test_no_cpu_fallback():
    rp, ci = F.read_mtx(golden_graph_path("karate.mtx"))
    try:
        e = F.Engine(rp, ci, 16)
    except _lib.F2VError as ex:  # no GPU here: the product refuses, loudly
        assert ex.code == _lib.F2V_ENODEV and "no CPU fallback" in str(ex)
    else:
        e.close()  # on a GPU box the engine simply exists
    src = ""
    for root, _, files in os.walk(os.path.join(ROOT, "force2vec_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".hpp")):
                src += open(os.path.join(root, f)).read()
    # the product never imports, links or loads the oracle
    assert not re.search(r"^\s*(import|from)\s+oracle\b", src, flags=re.M) and "liboracle" not in src


def test_rng_matches_libc_and_oracle(manifest):
    L = _lib.lib()
    libc = C.CDLL("libc.so.6")
    for seed in (1, 7, 0, 4242424242):
        g = L.f2v_rng_create(seed)
        libc.srand(seed)
        o = O.Rng(seed)
        for _ in range(3000):
            v = L.f2v_rng_next(g)
            assert v == libc.rand() == o.rand()
        L.f2v_rng_destroy(g)
    g = L.f2v_rng_create(1)
    assert [L.f2v_rng_next(g) for _ in range(16)] == manifest["rand_after_srand1"]
    L.f2v_rng_destroy(g)


def test_sigmoid_table_is_the_source_level_one():
    assert np.array_equal(F.sm_table(), O.sm_table())


@pytest.mark.parametrize("g", ["karate.mtx", "cora.mtx", "citeseer.mtx"])
def test_read_mtx_matches_oracle_reader(g):
    rp, ci = F.read_mtx(golden_graph_path(g))
    rp2, ci2 = O.read_mtx(golden_graph_path(g))
    assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
    if g == "cora.mtx":  # N=2708, nnz=10858 (duplicates kept, no self-loops)
        assert len(rp) - 1 == 2708 and len(ci) == 10858
    for i in range(len(rp) - 1):
        assert np.all(np.diff(ci[rp[i]:rp[i + 1]].astype(np.int64)) >= 0)


def test_read_mtx_edge_cases(tmp_path):
    # symmetric: mirrored, self-loop dropped, duplicate kept, isolated vertex 5, comment lines, values ignored
    p = tmp_path / "s.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real symmetric\n% a comment\n5 5 5\n2 1 0.5\n3 1 2\n3 3 9\n2 1 7\n4 2 1\n")
    rp, ci = F.read_mtx(str(p))
    assert rp.tolist() == [0, 3, 6, 7, 8, 8] and ci.tolist() == [1, 1, 2, 0, 0, 3, 0, 1]
    rp2, ci2 = O.read_mtx(str(p))
    assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
    # general: kept as directed, diagonal kept (sample/IO.h:122-134 only drops it for symmetric files)
    p = tmp_path / "g.mtx"
    p.write_text("%%MatrixMarket matrix coordinate pattern general\n3 3 4\n1 2\n2 2\n3 1\n1 3\n")
    rp, ci = F.read_mtx(str(p))
    assert rp.tolist() == [0, 2, 3, 4] and ci.tolist() == [1, 2, 1, 0]
    rp2, ci2 = O.read_mtx(str(p))
    assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
    with pytest.raises(_lib.F2VError):
        F.read_mtx(str(tmp_path / "missing.mtx"))
    p = tmp_path / "bad.mtx"
    p.write_text("%%MatrixMarket matrix coordinate pattern general\n3 3 1\n1 9\n")
    with pytest.raises(_lib.F2VError):
        F.read_mtx(str(p))


def test_write_embd_reproduces_reference_text(tmp_path, manifest):
    for case in manifest["cases"]:
        if "file" not in case:
            continue
        with gzip.open(os.path.join(GOLD, case["file"]), "rb") as f:
            txt = f.read()
        ref = tmp_path / "ref.embd"
        ref.write_bytes(txt)
        X = O.read_embd(str(ref))
        out = tmp_path / "out.embd"
        F.write_embd(str(out), X)
        assert out.read_bytes() == txt, case["name"]
    X = np.array([[0.0, -0.0, 1e-30, -1e30, 123456.7, 1234567.0, 0.1, np.float32(1 / 3)]], dtype=np.float32)
    F.write_embd(str(tmp_path / "a.embd"), X)
    O.write_embd(str(tmp_path / "b.embd"), X)
    assert (tmp_path / "a.embd").read_bytes() == (tmp_path / "b.embd").read_bytes()


def test_output_names_match_reference(manifest):
    for case in manifest["cases"]:
        got = F.output_name("/some/dir/" + case["graph"], "/out/", case["option"], case["bs"], case["batch"], case["dim"], case["iters"], case["ns"])
        assert got == "/out/" + case["embd_name"]
    assert F.output_name("g.mtx", "", 11, 0, 256, 128, 10, 5) == "g.mtxF2VNSLB_AVXZ256D128IT10NS5.embd"
    assert F.output_name("g.mtx", "", 11, 0, 256, 64, 10, 5) == "g.mtxF2VNSLB_AVXZ64256D64IT10NS5.embd"
    with pytest.raises(_lib.F2VError):
        F.output_name("g.mtx", "", 3, 0, 256, 64, 10, 5)


def test_cli_argument_behaviour(tmp_path):
    exe = os.path.join(ROOT, "bin", "Force2Vec")
    r = subprocess.run([exe, "-h"], capture_output=True, text=True)
    assert r.returncode == 1 and "Usage of Force2Vec tool" in r.stdout and "-nsamples" in r.stdout
    r = subprocess.run([exe, "-iter", "3"], capture_output=True, text=True)
    assert r.returncode == 1 and "Valid input file needed" in r.stdout
    r = subprocess.run([exe, "-input", golden_graph_path("karate.mtx"), "-option", "3"], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 1 and "out of scope" in r.stdout
    r = subprocess.run([exe, "-input", str(tmp_path / "nope.mtx")], capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 2 and "cannot open" in r.stderr


def test_binary_csr_cache_and_binary_embd(tmp_path):
    rp, ci = F.read_mtx(golden_graph_path("cora.mtx"))
    p = tmp_path / "cora.f2vcsr"
    F.write_csr_bin(p, rp, ci)
    rp2, ci2 = F.read_csr_bin(p)
    assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
    open(p, "r+b").truncate(os.path.getsize(p) - 8)
    with pytest.raises(_lib.F2VError):
        F.read_csr_bin(p)
    with pytest.raises(_lib.F2VError):
        F.read_csr_bin(golden_graph_path("cora.mtx"))
    X = np.random.default_rng(0).standard_normal((37, 16)).astype(np.float32)
    F.write_embd_bin(tmp_path / "x.bin", X)
    D = np.fromfile(tmp_path / "x.bin", np.float32)  # readBinEmbeddings, runnodeclassclust.py:81-86
    assert np.array_equal(D.reshape(-1, 16), X)


def test_embedding_readers_round_trip(tmp_path):
    """f2v_read_embd / f2v_read_embd_bin read back what the writers (and the reference's writeToFile) produce: the text form to the
    printed 6 digits and in ANY row order, the binary form bit for bit; malformed files are refused."""
    rng = np.random.default_rng(0)
    X = rng.uniform(-3, 3, (37, 24)).astype(np.float32)
    txt, binf = str(tmp_path / "x.embd"), str(tmp_path / "x.bin")
    F.write_embd(txt, X)
    F.write_embd_bin(binf, X)
    assert np.array_equal(F.read_embd_bin(binf, 37, 24), X)
    got = F.read_embd(txt)
    assert got.shape == X.shape and np.array_equal(got, O.read_embd(txt)) and np.abs(got - X).max() < 5e-5
    lines = open(txt).read().splitlines()
    open(txt, "w").write("\n".join([lines[0]] + lines[:0:-1]) + "\n")     # rows in reverse order: ids place them
    assert np.array_equal(F.read_embd(txt), got)
    with pytest.raises(F.F2VError):
        F.read_embd_bin(binf, 37, 25)                                        # not exactly n * dim floats
    open(txt, "w").write("\n".join(lines[:-1]) + "\n")                        # a row missing
    with pytest.raises(F.F2VError):
        F.read_embd(txt)
    open(txt, "w").write("\n".join([lines[0], lines[1], lines[1]] + lines[3:]) + "\n")  # a row twice
    with pytest.raises(F.F2VError):
        F.read_embd(txt)
    open(txt, "w").write("\n".join(lines) + "\n\n  \n")                                # trailing white space is not content ...
    assert np.array_equal(F.read_embd(txt), got)
    open(txt, "w").write("\n".join(lines + [lines[1]]) + "\n")                          # ... a row beyond the announced N is
    with pytest.raises(F.F2VError):
        F.read_embd(txt)
    wide = rng.uniform(-1, 1, (3, 700)).astype(np.float32)                              # the reference's writeToFile has no bound on D
    F.write_embd(txt, wide)
    assert F.read_embd(txt).shape == (3, 700) and np.abs(F.read_embd(txt) - wide).max() < 5e-6


def test_threaded_embd_reader_equals_single_thread(tmp_path):
    """f2v_read_embd splits a file above 4 MB over the host's threads at white space (ranges may start inside a row): the matrix is the
    single-threaded one bit for bit, for any thread count, with rows shuffled, with tokens wrapped over lines, and with the special values
    "%g" prints; one bad token anywhere (a value, an id, a repeated id) is refused and the FIRST bad row is the one named."""
    rng = np.random.default_rng(11)
    n, dim = 9000, 64
    X = rng.uniform(-3, 3, (n, dim)).astype(np.float32)
    X[5, 3], X[17, 0], X[8000, 63], X[44, 1] = np.inf, -np.inf, np.float32(1e-42), np.float32(3.3e38)
    path = str(tmp_path / "big.embd")
    F.write_embd(path, X)
    assert os.path.getsize(path) > (4 << 20)
    lines = open(path).read().splitlines()
    order = rng.permutation(n)
    body = [lines[1 + i] for i in order]
    body[100] = body[100].replace(" ", "\n", 7)         # tokens are white-space separated, line ends included (what fscanf read)
    body[101] = body[101].replace(" ", "\t  ", 3)
    text = "\n".join([lines[0]] + body) + "\n"
    open(path, "w").write(text)

    def read(threads):
        os.environ["F2V_IO_THREADS"] = str(threads)
        try:
            return F.read_embd(path)
        finally:
            del os.environ["F2V_IO_THREADS"]

    one = read(1)
    grid = np.array(text.split()[2:], dtype=np.float64).reshape(n, dim + 1)           # numpy's own parse of the same tokens
    want = np.empty((n, dim), np.float32)
    want[grid[:, 0].astype(np.int64) - 1] = grid[:, 1:].astype(np.float32)
    assert np.array_equal(one, want) and np.isinf(one[5, 3]) and one[17, 0] == -np.inf
    assert np.allclose(one[np.isfinite(X)], X[np.isfinite(X)], rtol=1e-5, atol=1e-44)
    for threads in (2, 3, 7, 16):
        assert np.array_equal(read(threads), one, equal_nan=True), threads
    toks = text.split(" ")
    for what, k, repl in (("value", len(toks) // 2, "1.5x"), ("value", len(toks) - 40, "abc"), ("id", None, None)):
        bad = list(toks)
        if what == "value":
            bad[k] = repl
            open(path, "w").write(" ".join(bad))
        else:
            rows = [lines[0]] + list(body)
            rows[6001] = "%d %s" % (order[10] + 1, rows[6001].split(" ", 1)[1])   # row 6001 repeats an earlier id
            open(path, "w").write("\n".join(rows) + "\n")
        for threads in (1, 5):
            with pytest.raises(F.F2VError) as e:
                read(threads)
            if what == "id":
                assert "row 6001 of 9000" in str(e.value) or "row 11 of 9000" in str(e.value), str(e.value)


def test_fast_g_formatter_equals_printf(tmp_path):
    """f2v_write_embd formats without printf where it can prove the digits (1e-4 <= |v| < 1e6: the scaled value is an exact double) and through
    sprintf elsewhere: the text must be what "%g " gives for EVERY float -- here the edge cases (exact ties, carries into the next power of ten
    and into 1e+06, the 1e-4 boundary from both sides, denormals, zeros, inf, nan) and 400 000 random bit patterns, against Python's own
    "%g" (C semantics), in one- and many-threaded runs.  (All 2^32 patterns: tools/src/fmt_check.cpp.)"""
    rng = np.random.default_rng(7)
    edge = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 0.25, 0.125, 1e-4, 9.9999997e-05, 1.00000005e-04, 9.99999e-05, 0.00010000005, 999999.5, 999999.44, 999999.4375,
                     999999.9, 1e6, 1e-5, 123456.5, 123457.5, 0.1234565, 1.5, 2.5, 100000.5, 100001.5, 9.9999995, 99999.95, 0.999999523, 0.99999994,
                     3.4028235e38, 1.17549435e-38, 1e-45, np.inf, -np.inf, np.nan, 65504.0, 0.333333343, 2.0 / 3.0, 1234567.0, 0.000123456789], dtype=np.float32)
    ties = (np.arange(100000, 100064, dtype=np.float64) + 0.5).astype(np.float32) / np.float32(8.0)   # .0625 steps: exact ties at the 6th digit
    bits = rng.integers(0, 1 << 32, 400000, dtype=np.uint64).astype(np.uint32).view(np.float32)
    vals = np.concatenate([edge, ties, bits, rng.uniform(-3, 3, 100000).astype(np.float32)])
    vals[np.isnan(vals)] = np.float32(np.nan)  # (glibc prints a NaN's sign, Python's "%g" does not: both go through sprintf anyway)
    dim = 8
    vals = vals[: len(vals) // dim * dim].reshape(-1, dim)
    want_lines = ["%d %d" % vals.shape] + ["%d " % (i + 1) + "".join("%g " % float(v) for v in row) for i, row in enumerate(vals)]
    want = ("\n".join(want_lines) + "\n").encode()
    for threads in ("1", "5"):
        os.environ["F2V_IO_THREADS"] = threads
        try:
            path = str(tmp_path / ("g%s.embd" % threads))
            F.write_embd(path, vals)
            got = open(path, "rb").read()
        finally:
            del os.environ["F2V_IO_THREADS"]
        if got != want:
            gl, wl = got.split(b"\n"), want.split(b"\n")
            k = next(i for i in range(min(len(gl), len(wl))) if gl[i] != wl[i])
            raise AssertionError("line %d differs:\n got  %r\n want %r" % (k, gl[k][:200], wl[k][:200]))


def test_header_is_plain_c_and_links(tmp_path):
    """include/f2v.h is a C header (no C++ or torch types in the signatures) and a C program links against libf2v."""
    src = tmp_path / "t.c"
    src.write_text('#include "f2v.h"\n#include <stdio.h>\nint main(void){ f2v_rng *g = f2v_rng_create(1); int v = f2v_rng_next(g); f2v_rng_destroy(g);'
                   ' printf("%d %s\\n", v, f2v_version()); f2v_handle h = 0; unsigned rp[3] = {0,0,0};'
                   ' int rc = f2v_create(rp, 0, 2, 0, 8, 0, &h); if (rc == 0) f2v_destroy(h); else printf("%s\\n", f2v_last_error()); return 0; }\n')
    exe = tmp_path / "t"
    libdir = os.path.dirname(_lib.LIB_PATH)
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                           "-L", libdir, "-lf2v", "-Wl,-rpath," + libdir])
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("1804289383 f2v-mi355x"), out.stdout + out.stderr


def test_parallel_init_equals_the_serial_rand_stream():
    """N*D initial values are draws of ONE serial rand() stream; libf2v cuts it into chunks with a 31x31 jump-ahead
    matrix (the generator is linear over Z/2^32) and fills them with threads -- same bits as the serial draw, same
    stream position afterwards."""
    L = _lib.lib()
    libc = C.CDLL("libc.so.6")
    for k in (0, 1, 2, 30, 31, 32, 1000, 123457):
        g = L.f2v_rng_create(5)
        L.f2v_rng_jump(g, k)
        libc.srand(5)
        for _ in range(k):
            libc.rand()
        assert [L.f2v_rng_next(g) for _ in range(40)] == [libc.rand() for _ in range(40)], k
        L.f2v_rng_destroy(g)
    for kind in (0, 1):
        for count in (1000, (1 << 20) + 12345, 3000001):  # below and above the parallel threshold, ragged chunks
            g = L.f2v_rng_create(1)
            out = np.empty(count, dtype=np.float32)
            _lib.check(L.f2v_rng_fill(g, out.ctypes.data_as(_lib.f32p), count, kind))
            o = O.Rng(1)
            want = o.init_embeddings(1, count, kind)[0]
            assert np.array_equal(out, want), (kind, count)
            assert L.f2v_rng_next(g) == o.rand()
            L.f2v_rng_destroy(g)


def test_shard_bounds_balance_work_not_rows():
    """f2v_shard_bounds: contiguous slices covering the minibatch, each within one row's weight of an equal share of
    sum(degree + 4); a hub heavier than a whole share gets a slice of its own (its neighbours' slices may be empty)."""
    from force2vec_amd.engine import shard_bounds_balanced
    from force2vec_amd.graph import rmat_csr
    rp, _ = rmat_csr(14, 16, 2)
    n = len(rp) - 1
    deg = np.diff(rp.astype(np.int64))
    w = deg + 4
    for lo, hi, world in [(0, n, 8), (1000, 9192, 4), (5, 6, 3), (7, 7, 2), (n - 100, n, 5), (0, 4096, 1)]:
        b = shard_bounds_balanced(rp, lo, hi, world).astype(np.int64)
        assert b[0] == lo and b[-1] == hi and np.all(np.diff(b) >= 0)
        total = w[lo:hi].sum()
        for r in range(world):
            mine = w[b[r]:b[r + 1]].sum()
            heaviest = w[b[r]:b[r + 1]].max() if b[r + 1] > b[r] else 0
            assert mine <= total / world + heaviest + 1, (lo, hi, world, r)
    # a graph whose first row is one enormous hub: the hub is a slice, the rest is shared out behind it
    rp2 = np.concatenate([[0, 10000], 10000 + 3 * np.arange(1, 64)]).astype(np.uint32)
    b = shard_bounds_balanced(rp2, 0, 64, 4)
    assert b[1] == 1 and list(b) == sorted(b) and b[-1] == 64


def test_push_masks_name_exactly_the_ranks_that_read_a_row():
    """f2v_push_masks against a brute-force restatement: rank r reads v iff v is a CSR neighbour of a row in one of
    r's minibatch slices (f2v_shard_bounds), or v is sampled; the owner's own bit is never set."""
    from force2vec_amd.engine import push_masks, shard_bounds_balanced
    from force2vec_amd.graph import rmat_csr
    rp, ci = rmat_csr(11, 8, 3)
    n = len(rp) - 1
    for batch, world in [(256, 2), (300, 3), (4096, 8), (100, 5)]:
        samples = np.array([1, 7, n - 1, 7], dtype=np.uint32)
        got = push_masks(rp, ci, batch, world, samples)
        owner = np.zeros(n, dtype=np.int64)
        for lo in range(0, n, batch):
            hi = min(lo + batch, n)
            b = shard_bounds_balanced(rp, lo, hi, world)
            for r in range(world):
                owner[b[r]:b[r + 1]] = r
        want = np.zeros(n, dtype=np.uint32)
        for u in range(n):
            want[ci[rp[u]:rp[u + 1]]] |= np.uint32(1 << owner[u])
        want[samples] = (1 << world) - 1
        want &= ~(np.uint32(1) << owner.astype(np.uint32))
        assert np.array_equal(got, want), (batch, world)
    # the threaded path (>= 2^20 nonzeros) agrees with the serial one
    rp, ci = rmat_csr(16, 16, 5)
    os.environ["F2V_IO_THREADS"] = "1"
    a = push_masks(rp, ci, 8192, 8)
    os.environ["F2V_IO_THREADS"] = "7"
    b = push_masks(rp, ci, 8192, 8)
    del os.environ["F2V_IO_THREADS"]
    assert np.array_equal(a, b) and a.any()


def test_fast_graph_generators_equal_the_plain_numpy_path():
    """rmat_csr / orkut_like_csr build their CSR through scipy's compiled COO -> CSR conversion (duplicates merged there); the
    arrays must be exactly what unique edge list + global key sort give -- the graphs every recorded number refers to."""
    from force2vec_amd import graph as G
    for scale, ef in ((10, 16), (14, 16), (13, 4)):
        n, s, d = G.rmat_edges(scale, ef, 1)
        rp, ci = G.csr_from_undirected_edges(n, s, d)
        rp2, ci2 = G.rmat_csr(scale, ef, 1)
        assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
    n, s, d = G.rmat_edges_n(3001, 40000, 1)
    rp, ci = G.csr_from_undirected_edges(n, s, d)
    n2, s2, d2 = G._rmat_pairs_n(3001, 40000, 1, 0.57, 0.19, 0.19)
    rp2, ci2 = G._symmetric_csr_dedup(n2, s2, d2)
    assert n == n2 and np.array_equal(rp, rp2) and np.array_equal(ci, ci2)


def test_bench_self_launch_propagates_failure():
    """`python bench.py --gpus 2` without a launcher starts its own ranks (torch.distributed.run as a child).  Without a GPU
    the ranks fail (F2V_ENODEV: there is no CPU fallback): the launcher must pass the non-zero exit code on and print no
    result line."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["F2V_BENCH_QUIET"] = "1"
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible: the ranks would run")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--scale", "10", "--batch", "256",
                        "--steps", "1", "--warmup", "0", "--no-preflight", "--config5-scale", "0", "--config4", "0"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0
    assert r.stdout.strip() == "", r.stdout
    assert "torch.distributed.run" in r.stderr


def _walk_graphs():
    """(name, rowptr, colids): real, power-law, and graphs made to defeat the walk generator's predictions"""
    from force2vec_amd.graph import rmat_csr
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ("karate", "cora"):
        rp, ci = F.read_mtx(os.path.join(here, "golden", name + ".mtx"))
        yield name, rp, ci
    rp, ci = rmat_csr(13, 8, 3)
    yield "rmat13", rp, ci
    rng = np.random.default_rng(5)
    # degrees 0..4 mixed at random (draw / no draw alternate unpredictably), ids random: every block mispredicts
    n = 3000
    deg = rng.integers(0, 5, size=n)
    rp = np.concatenate(([0], np.cumsum(deg))).astype(np.uint32)
    ci = np.concatenate([np.sort(rng.choice(n, size=d, replace=False)) for d in deg] + [np.zeros(0, dtype=np.int64)]).astype(np.uint32)
    yield "mixed", rp, ci
    # a ring (every degree 2: nothing is ever drawn), and a graph of one edge among many isolated vertices
    n = 100
    ring = np.stack([(np.arange(n) - 1) % n, (np.arange(n) + 1) % n], axis=1)
    yield "ring", (2 * np.arange(n + 1)).astype(np.uint32), np.sort(ring, axis=1).reshape(-1).astype(np.uint32)
    rp = np.zeros(41, dtype=np.uint32)
    rp[8:] = 1
    rp[31:] = 2
    yield "one-edge", rp, np.array([30, 7], dtype=np.uint32)
    # fewer vertices than one block of walks, all of degree 3
    yield "k4", (3 * np.arange(5)).astype(np.uint32), np.array([1, 2, 3, 0, 2, 3, 0, 1, 3, 0, 1, 2], dtype=np.uint32)


@pytest.mark.parametrize("in_flight", [None, "1", "3", "7", "64"])
def test_walks_from_the_rand_stream_equal_the_reference_loop(in_flight, monkeypatch):
    """f2v_rng_walks (up to 64 walks in flight, each at its own step, from predicted stream positions; the younger ones issued again behind
    a walk that drew fewer numbers than predicted) = the oracle's serial loop (algorithms.cpp:1097-1118): same samples, and the stream
    stands where the serial loop leaves it -- two epochs in a row, several seeds; with the adaptive number of walks in flight and with
    fixed ones (F2V_WALKS_IN_FLIGHT: 1 = the serial loop itself, odd widths, the maximum)."""
    if in_flight is None:
        monkeypatch.delenv("F2V_WALKS_IN_FLIGHT", raising=False)
    else:
        monkeypatch.setenv("F2V_WALKS_IN_FLIGHT", in_flight)
    L = _lib.lib()
    for name, rp, ci in _walk_graphs():
        n = len(rp) - 1
        for seed in (1, 99):
            g = L.f2v_rng_create(seed)
            o = O.Rng(seed)
            for epoch in range(2):
                want = O.generate_walks(o, rp, ci)
                got = np.empty(5 * n, dtype=np.uint32)
                _lib.check(L.f2v_rng_walks(g, rp.ctypes.data_as(_lib.u32p), ci.ctypes.data_as(_lib.u32p), n, len(ci), got.ctypes.data_as(_lib.u32p)))
                assert np.array_equal(got, want), (name, seed, epoch, int(np.flatnonzero(got != want)[0]))
                assert [L.f2v_rng_next(g) for _ in range(40)] == [o.rand() for _ in range(40)], (name, seed, epoch)
            L.f2v_rng_destroy(g)
    g = L.f2v_rng_create(1)
    bad = np.array([0, 1, 2], dtype=np.uint32)
    out = np.empty(10, dtype=np.uint32)
    off_graph = np.array([1, 5], dtype=np.uint32)  # a column id that is no vertex: refused, nothing drawn
    assert L.f2v_rng_walks(g, bad.ctypes.data_as(_lib.u32p), off_graph.ctypes.data_as(_lib.u32p), 2, 2, out.ctypes.data_as(_lib.u32p)) == _lib.F2V_EINVAL
    assert L.f2v_rng_next(g) == O.Rng(1).rand()
    L.f2v_rng_destroy(g)

"""CPU: the launch plans of the WIDE form of chained minibatches (round 3), checked without a device by the self-test build's
f2v_test_wide_plan_check -- the host logic the kernel's correctness and LIVENESS rest on: every row finished exactly once, every
neighbour of a split row in exactly one piece, rounds / phases / LDS slots / passes well-formed, a job adding consecutive pieces of
one row in neighbour order, and every in-grid wait (an imported group sum, a combine-tree node's inputs) pointing at a workgroup
with a SMALLER index (workgroups start in index order: what a wave waits for is then running or done)."""
import ctypes as C

import numpy as np
import pytest

from conftest import golden_graph_path
from oracle import oracle as O

import force2vec_amd as F
from force2vec_amd import _lib


def check(rowptr, colids, dim, batch, walk=0, **params):
    T = _lib.selftest_lib()
    rowptr = np.ascontiguousarray(rowptr, dtype=np.uint32)
    colids = np.ascontiguousarray(colids, dtype=np.uint32)
    names = (C.c_char_p * len(params))(*[k.encode() for k in params])
    values = (C.c_int64 * len(params))(*[int(v) for v in params.values()])
    stats = (C.c_uint64 * 7)()
    rc = T.f2v_test_wide_plan_check(rowptr.ctypes.data_as(_lib.u32p), colids.ctypes.data_as(_lib.u32p), len(rowptr) - 1, len(colids), dim, batch, walk,
                                    names, values, len(params), stats)
    if rc != 0:
        raise AssertionError(T.f2v_last_error().decode())
    return dict(zip(("workgroups", "helpers", "finishers", "packed", "node_workgroups", "slots", "checksum"), [int(x) for x in stats]))


@pytest.fixture(scope="module")
def rmat():
    from force2vec_amd.graph import rmat_csr
    return rmat_csr(13, 16, seed=7)  # 8192 vertices, hubs of more than a thousand neighbours


@pytest.mark.parametrize("dim,batch,params", [
    (128, 256, {}), (128, 384, {"hub_chunk": 2}), (128, 100, {"hub_fanin": 4, "hub_chunk": 4}), (128, 64, {"hub_fanin": 2, "hub_chunk": 3, "wide_finish": 1}),
    (64, 512, {"wide_span": 1, "wide_finish": 2}), (32, 128, {"wide_phases": 3, "wide_rounds": 2}), (16, 256, {}), (16, 64, {"wide_min_width": 16}),
    (256, 200, {"hub_fanin": 8, "wide_finish": 8, "wide_span": 8}), (128, 2048, {"hub_chunk": 4}), (128, 37, {"wide_rows": 370, "wide_order": 2}),
    (96, 256, {"class_cut": 0}), (128, 256, {"hub_chunk": 0}), (48, 128, {"wide_min_width": 128, "hub_fanin": 32, "hub_chunk": 2}),
])
def test_wide_plans_are_well_formed_on_a_power_law_graph(rmat, dim, batch, params):
    rowptr, colids = rmat
    st = check(rowptr, colids, dim, batch, **params)
    n = len(rowptr) - 1
    assert st["workgroups"] >= -(-n // batch)
    deg = np.diff(rowptr.astype(np.int64))
    chunk = params.get("hub_chunk", 4)
    fanin = params.get("hub_fanin", 32)
    if chunk and deg.max() > chunk * fanin:          # rows of several fan-in groups exist: finishers
        assert st["finishers"] > 0
    if chunk and deg.max() > chunk * fanin * fanin:   # rows of more than fanin^2 pieces: combine-tree nodes above their units
        assert st["node_workgroups"] > 0 and st["slots"] > 0
    if chunk == 0:
        assert st["finishers"] == 0 and st["helpers"] == 0 and st["slots"] == 0


@pytest.mark.parametrize("graph,dim,batch", [("karate.mtx", 128, 16), ("karate.mtx", 16, 8), ("cora.mtx", 128, 256), ("cora.mtx", 16, 256), ("cora.mtx", 64, 384)])
def test_wide_plans_on_the_reference_graphs(graph, dim, batch):
    rowptr, colids = O.read_mtx(golden_graph_path(graph))
    check(rowptr, colids, dim, batch)
    check(rowptr, colids, dim, batch, walk=1)      # option 7: every row is one item of five walk samples
    check(rowptr, colids, dim, batch, hub_fanin=4, wide_finish=2, wide_span=1)


@pytest.mark.parametrize("dim,batch,params", [(128, 100, {"wide_rows": 1000}), (64, 37, {"wide_rows": 370, "hub_fanin": 4}), (16, 64, {"wide_rows": 512})])
def test_plans_built_on_several_host_threads_are_the_serial_ones(rmat, dim, batch, params):
    """wide_plans_for_epoch: the plans of an epoch are independent of each other and are built side by side on host threads, then appended
    in launch order -- the resident arrays (items, jobs, workgroup descriptors, tree nodes) are byte for byte the serial build's, and
    the checker passes on them."""
    rowptr, colids = rmat
    serial = check(rowptr, colids, dim, batch, **params)
    for threads in (2, 5):
        assert check(rowptr, colids, dim, batch, plan_threads=threads, **params) == serial
    walk = check(rowptr, colids, dim, batch, walk=1, **params)
    assert check(rowptr, colids, dim, batch, walk=1, plan_threads=3, **params) == walk


def test_wide_plan_check_sees_a_broken_plan_shape():
    """(the checker itself: shapes the wide form does not run are refused, not waved through)"""
    rowptr, colids = O.read_mtx(golden_graph_path("karate.mtx"))
    with pytest.raises(AssertionError):
        check(rowptr, colids, 128, 16, hub_fanin=64)   # fan-in groups must fit the 32 piece slots
    with pytest.raises(AssertionError):
        check(rowptr, colids, 130, 16)                 # no sub-wave layout for D = 130

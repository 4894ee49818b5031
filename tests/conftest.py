import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # a fresh checkout has no built artefacts: build the product library / CLI and the test oracle once
    # (hipcc cross-compiles without a GPU; on the GPU box the prebuilt files travel with the snapshot)
    import subprocess
    if not all(os.path.exists(os.path.join(ROOT, p)) for p in ("force2vec_amd/libf2v.so", "force2vec_amd/libf2v_selftest.so", "bin/Force2Vec")):
        subprocess.check_call(["make", "-s", "-C", ROOT, "all"])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])


@pytest.fixture(scope="session")
def manifest():
    with open(os.path.join(GOLD, "manifest.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLD


def golden_graph_path(name, tmpdir_factory=None):
    """Path of a bundled graph; citeseer is stored gzipped and unpacked on demand."""
    p = os.path.join(GOLD, name)
    if os.path.exists(p):
        return p
    gz = p + ".gz"
    out = os.path.join("/tmp", "f2v_golden_" + name)
    if not os.path.exists(out):
        with gzip.open(gz, "rb") as fi, open(out + ".tmp%d" % os.getpid(), "wb") as fo:
            fo.write(fi.read())
        os.replace(out + ".tmp%d" % os.getpid(), out)
    return out

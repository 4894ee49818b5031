"""glibc rand() restatement (oracle side) against the C library itself and the committed
known-answer values (first outputs after srand(1): 1804289383, 846930886, ... SURVEY 8a-a2)."""
import ctypes

import numpy as np

from oracle import oracle as O


def test_kat(manifest):
    g = O.Rng(1)
    assert [g.rand() for _ in range(16)] == manifest["rand_after_srand1"]
    assert manifest["rand_after_srand1"][:3] == [1804289383, 846930886, 1681692777]
    g = O.Rng(1)
    for _ in range(1000000):
        v = g.rand()
    assert v == manifest["rand_1000000th"]


def test_against_libc():
    libc = ctypes.CDLL("libc.so.6")
    for seed in (1, 2, 12345, 0, 2**31 + 7):
        libc.srand(seed)
        g = O.Rng(seed)
        assert all(libc.rand() == g.rand() for _ in range(5000)), seed


def test_init_ranges():
    g = O.Rng(1)
    X = g.init_embeddings(100, 16, 0)
    assert X.min() >= -1.0 and X.max() < 1.0
    assert X[0, 0] == np.float32(-1.0 + 2.0 * 1804289383 / 2147483648.0)
    Y = g.init_embeddings(100, 16, 1)
    assert Y.min() >= 0.0 and Y.max() <= 1.0

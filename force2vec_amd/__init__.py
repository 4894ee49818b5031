"""force2vec_amd -- MI355X (gfx950) Force2Vec embedding engine.

csrc/ holds the HIP kernels and the C ABI (libf2v.so, include/f2v.h); this package is the
host-side mirror of the reference's interface for the one hot path (options 5-11 force
kernels + SGD row update).  Importing it never imports anything from oracle/."""
from . import _lib  # noqa: F401
from ._lib import F2VError  # noqa: F401
from .engine import Engine, algorithms, output_name, push_masks, read_embd, read_embd_bin, sm_table, write_embd, write_embd_bin  # noqa: F401
from .graph import read_csr_bin, read_mtx, rmat_csr, write_csr_bin  # noqa: F401

__all__ = ["Engine", "F2VError", "algorithms", "read_mtx", "rmat_csr", "write_embd", "output_name", "sm_table"]

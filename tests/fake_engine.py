"""TEST INFRASTRUCTURE: an oracle-backed stand-in for force2vec_amd.Engine with the same
methods the multi-GPU driver uses, so the sharding / exchange logic of force2vec_amd/dist.py
can be verified on CPU (gloo, world_size 2).  Never used by the product."""
import numpy as np

from oracle import oracle as O


class OracleEngine:
    def __init__(self, rowptr, colids, dim, chunk=0, order=O.ORDER_TREE):
        self.rowptr, self.colids = rowptr, colids
        self.n, self.dim = len(rowptr) - 1, dim
        self.chunk, self.order = chunk, order
        self.rng = O.Rng(1)
        self.X = None
        self.walks = None
        self.pending = None  # (lo, hi, rows)

    def srand(self, seed=1):
        self.rng = O.Rng(seed)

    def init_embeddings(self, kind):
        self.X = self.rng.init_embeddings(self.n, self.dim, kind)

    def get_param(self, name):
        return {"hub_chunk_auto": 0, "hub_chunk": self.chunk}[name]

    def set_param(self, name, value):
        raise AssertionError("not expected: " + name)

    def stage_reserve(self, rows):
        pass

    def draw_samples(self, max_num, count, keep=None):
        keep = count if keep is None else keep
        out = [self.rng.rand_index(max_num, 0) for _ in range(count)]
        return np.array(out[:keep], dtype=np.uint32)

    def generate_walks(self):
        self.walks = O.generate_walks(self.rng, self.rowptr, self.colids)
        return self.walks

    def _commit(self):
        if self.pending is not None:
            lo, hi, rows = self.pending
            self.X[lo:hi] = rows
            self.pending = None

    def minibatch_step(self, option, lo, hi, ids, ns, lr, bs_mode=0, row_lo=None, row_hi=None):
        self._commit()
        row_lo = lo if row_lo is None else row_lo
        row_hi = hi if row_hi is None else row_hi
        work = self.X.copy()
        math = {5: 5, 8: 5, 11: 5, 6: 6, 9: 6, 7: 7, 10: 10}[option]  # (option 10: option 7 without the division by deg + 1, as the engine)
        O.minibatch(math, self.rowptr, self.colids, work, lo, hi, ids, ns, lr, bs_mode=bs_mode, walks=self.walks,
                    order=self.order, chunk=self.chunk, row_lo=row_lo, row_hi=row_hi)
        rows = np.full((hi - lo, self.dim), np.nan, dtype=np.float32)  # rows of other ranks must arrive by exchange
        rows[row_lo - lo: row_hi - lo] = work[row_lo:row_hi]
        self.pending = (lo, hi, rows)

    def upload_sample_ids(self, ids):
        self.ids = np.array(ids, dtype=np.uint32)

    def minibatch_step_at(self, option, lo, hi, ids_offset, ns, lr, bs_mode=0, row_lo=None, row_hi=None):
        need = (hi - lo) + ns - 1 if bs_mode else ns
        self.minibatch_step(option, lo, hi, self.ids[ids_offset: ids_offset + need], ns, lr, bs_mode, row_lo, row_hi)

    def stage_read(self, row_lo, row_hi):
        lo, hi, rows = self.pending
        return rows[row_lo - lo: row_hi - lo].copy()

    def stage_write(self, row_lo, row_hi, data):
        lo, hi, rows = self.pending
        rows[row_lo - lo: row_hi - lo] = data

    def rows_read(self, ids):
        lo, hi, rows = self.pending
        ids = np.asarray(ids, dtype=np.int64)
        assert np.all((ids >= lo) & (ids < hi))
        return rows[ids - lo].copy()

    def rows_write(self, ids, data):
        lo, hi, rows = self.pending
        ids = np.asarray(ids, dtype=np.int64)
        assert np.all((ids >= lo) & (ids < hi))
        rows[ids - lo] = data

    def set_embeddings(self, X):
        self.pending = None
        self.X = np.array(X, dtype=np.float32)

    def flush(self):
        self._commit()

    def get_embeddings(self):
        self._commit()
        return self.X.copy()

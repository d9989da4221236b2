"""The callers' side of the PD/PI path at PubMed scale.

The reference's harness (loaddatas.py:38-54, TLCGNN.py:80-107) materialises
  * `1. - adj.toarray()`            dense N x N float64                      (3.1 GB for PubMed)
  * `neg_edges`                      [n_neg, 2] int64, shuffled               (3.1 GB)
  * `pi_sg`                          float64 [n_pairs, 25], cached as .npy    (39 GB)
of which only the shuffle ORDER and the 0.3 % of rows with d(u,v) <= hop carry information.  Here

  ShuffledNegatives   the same list, as (seeded permutation of list numbers) + device-side enumeration of the numbered
                      non-edges (engine.ComplementIndex): any slice of the reference's shuffled `neg_edges` in O(slice).
  sweep_images        streams the list through tlc_pd_pi_batch in chunks and keeps the informative rows only.
  SparseImages        status-coded sparse store with the dense array's indexing, `.npz` on disk.

The permutation is drawn by numpy's legacy MT19937 `shuffle` exactly where the reference draws it, so every slice is
identical to the reference's (`tests/golden/adj_split.npz`).  No CPU fallback for the enumeration or the images.
"""
import numpy as np

from . import _lib, engine


class ShuffledNegatives:
    """`neg_edges` after `np.random.shuffle(neg_edges)` (loaddatas.py:45-46) without the array.

    Must be constructed at the point of the reference's shuffle: it consumes the global numpy RNG stream exactly like
    `np.random.shuffle` of the [n_neg, 2] array does (one `random_interval` draw per row, last row first -- the legacy
    shuffle of an ndarray draws the same numbers whatever the row width)."""

    def __init__(self, rowptr, col, device=None):
        self.index = engine.ComplementIndex(rowptr, col, device=device)
        self.perm = np.arange(len(self.index), dtype=np.int64)
        np.random.shuffle(self.perm)

    def __len__(self):
        return len(self.perm)

    def device_pairs(self, lo=0, hi=None):
        """int32 CUDA [hi-lo, 2] = neg_edges[lo:hi]."""
        import torch
        hi = len(self) if hi is None else min(hi, len(self))
        ranks = torch.from_numpy(self.perm[lo:hi]).to(self.index.rowptr.device)
        return self.index.pairs(ranks=ranks)

    def __getitem__(self, sl):
        """numpy int64 [k, 2], like slicing the reference's array (use for the val/test slices, not for the whole list)."""
        if isinstance(sl, slice):
            lo, hi, step = sl.indices(len(self))
            assert step == 1
            return self.device_pairs(lo, hi).cpu().numpy().astype(np.int64)
        import torch
        ranks = torch.from_numpy(self.perm[np.asarray(sl)]).to(self.index.rowptr.device)
        return self.index.pairs(ranks=ranks.reshape(-1)).cpu().numpy().astype(np.int64)


class LazyPairList:
    """`total_edges` of TLCGNN.py:80 -- the six pair lists back to back -- with the negative list kept as a ShuffledNegatives.

    segments: list of (pairs, label) with pairs an int array [k, 2] or a ShuffledNegatives; label 1 = positive."""

    def __init__(self, segments):
        self.segments = list(segments)
        self.bounds = np.concatenate([[0], np.cumsum([len(p) for p, _ in self.segments])]).astype(np.int64)

    def __len__(self):
        return int(self.bounds[-1])

    def device_pairs(self, lo, hi, device=None):
        import torch
        out = []
        for k, (p, _) in enumerate(self.segments):
            a, b = max(lo, int(self.bounds[k])), min(hi, int(self.bounds[k + 1]))
            if a >= b:
                continue
            a, b = a - int(self.bounds[k]), b - int(self.bounds[k])
            if isinstance(p, ShuffledNegatives):
                out.append(p.device_pairs(a, b))
            else:
                dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
                out.append(torch.from_numpy(np.ascontiguousarray(np.asarray(p)[a:b], dtype=np.int32)).to(dev))
        return out[0] if len(out) == 1 else torch.cat(out)

    def gather(self, index):
        """numpy int64 [k, 2] = total_edges[index]."""
        index = np.asarray(index, dtype=np.int64).reshape(-1)
        seg = np.searchsorted(self.bounds, index, side="right") - 1
        out = np.empty((len(index), 2), dtype=np.int64)
        for k, (p, _) in enumerate(self.segments):
            m = seg == k
            if m.any():
                out[m] = p[index[m] - self.bounds[k]] if isinstance(p, ShuffledNegatives) else np.asarray(p)[index[m] - self.bounds[k]]
        return out

    def labels(self, index):
        index = np.asarray(index, dtype=np.int64).reshape(-1)
        seg = np.searchsorted(self.bounds, index, side="right") - 1
        return np.asarray([lab for _, lab in self.segments], dtype=np.int64)[seg]


class SparseImages:
    """float64 [n_rows, width] image array of which only the rows that carry information are stored.

    idx int64[nnz] ascending, rows float64[nnz, width], status uint8[nnz] (TLC_ST_* of the stored rows: 0 for every non-zero
    row; a stored zero row with status != 0 is a pair the reference swallowed an exception for -- kept only on request),
    status_counts int64[8]: how many of ALL n_rows rows ended with each status byte (the reference keeps just that:
    `cnt_compute`).  Rows not stored are zero."""

    def __init__(self, n_rows, width, idx, rows, status, status_counts=None):
        self.shape = (int(n_rows), int(width))
        order = np.argsort(idx, kind="stable")
        self.idx = np.ascontiguousarray(np.asarray(idx, dtype=np.int64)[order])
        self.rows = np.ascontiguousarray(np.asarray(rows, dtype=np.float64).reshape(-1, width)[order])
        self.status = np.ascontiguousarray(np.asarray(status, dtype=np.uint8)[order])
        if status_counts is None:                       # every failed row is stored: the rest is TLC_ST_OK
            status_counts = np.bincount(self.status, minlength=8).astype(np.int64)
            status_counts[0] += self.shape[0] - len(self.idx)
        self.status_counts = np.asarray(status_counts, dtype=np.int64)
        self.unclassified = 0               # rows whose status byte was never computed (sweep_near: zero by the distance rule)
        self.near_pairs = None              # sweep_near: how many pairs passed the distance pre-filter
        self._dev = None

    @property
    def cnt_compute(self):
        """graph2pi.cnt_compute (riccidist2dgm.py:355): pairs whose image was computed without an exception (a lower bound when
        `unclassified` rows exist: the pre-filtered sweep does not run the pairs it knows to be zero)."""
        return int(self.status_counts[_lib.ST_OK])

    def __len__(self):
        return self.shape[0]

    def gather(self, index):
        """Dense float64 [k, width] rows for an int64 index array (numpy) -- what `PI[index]` gives on the dense array."""
        index = np.asarray(index, dtype=np.int64).reshape(-1)
        out = np.zeros((len(index), self.shape[1]), dtype=np.float64)
        if len(self.idx):
            pos = np.searchsorted(self.idx, index)
            pos_c = np.minimum(pos, len(self.idx) - 1)
            hit = self.idx[pos_c] == index
            out[hit] = self.rows[pos_c[hit]]
        return out

    def __getitem__(self, sl):
        if isinstance(sl, slice):
            lo, hi, step = sl.indices(self.shape[0])
            return self.gather(np.arange(lo, hi, step))
        return self.gather(sl)

    def gather_device(self, index):
        """Same on the GPU: index int64 CUDA tensor -> float64 CUDA [k, width] (the decode's `torch.Tensor(PI)` slice)."""
        import torch
        dev = index.device
        if self._dev is None or self._dev[0].device != dev:
            self._dev = (torch.from_numpy(self.idx).to(dev), torch.from_numpy(self.rows).to(dev))
        didx, drows = self._dev
        out = torch.zeros((index.numel(), self.shape[1]), dtype=torch.float64, device=dev)
        if didx.numel():
            pos = torch.searchsorted(didx, index.reshape(-1)).clamp_(max=didx.numel() - 1)
            hit = didx[pos] == index.reshape(-1)
            out[hit] = drows[pos[hit]]
        return out

    def to_dense(self):
        out = np.zeros(self.shape, dtype=np.float64)
        out[self.idx] = self.rows
        return out

    def save(self, path):
        np.savez(path, shape=np.asarray(self.shape, dtype=np.int64), idx=self.idx, rows=self.rows, status=self.status,
                 status_counts=self.status_counts)

    @classmethod
    def load(cls, path):
        d = np.load(path)
        return cls(int(d["shape"][0]), int(d["shape"][1]), d["idx"], d["rows"], d["status"], d["status_counts"])

    @classmethod
    def from_dense(cls, pi, status=None):
        pi = np.asarray(pi, dtype=np.float64)
        status = np.zeros(len(pi), dtype=np.uint8) if status is None else np.asarray(status, dtype=np.uint8)
        keep = (status != 0) | (pi != 0).any(1)
        idx = np.nonzero(keep)[0]
        return cls(pi.shape[0], pi.shape[1], idx, pi[idx], status[idx])


def sweep_images(graph, pair_source, n_pairs, hop, flags=0, res=5, chunk=1 << 22, index_base=0, store=None, keep_failed=False,
                 expect_fraction=0.004):
    """Images of a long pair list, streamed: pair_source(lo, hi) -> int32 CUDA [hi-lo, 2] for list positions lo..hi-1.

    Every chunk goes through tlc_pd_pi_batch (graph: engine.DeviceGraph) and tlc_select_rows, which appends the non-zero rows (and,
    with keep_failed, the zero rows whose pair failed) to ONE device-side store and adds the status bytes to a histogram; nothing is
    synchronised or copied until the list is through (the store is sized from `expect_fraction`; if it turns out too small the
    sweep is repeated with the size it reported).  Returns SparseImages over positions index_base .. index_base + n_pairs - 1 of
    a store with n_pairs rows (or extends `store`, a list of (idx, rows, status, status_counts) pieces, and returns None)."""
    import torch
    width = res * res
    dev = torch.device("cuda", graph.device)
    pieces = [] if store is None else store
    cap = max(1 << 16, int(n_pairs * expect_fraction) + (1 << 14))
    if keep_failed:
        cap = max(cap, n_pairs)
    with torch.cuda.device(dev):
        out = torch.empty((min(chunk, max(n_pairs, 1)), width), dtype=torch.float64, device=dev)
        st = torch.empty(out.shape[0], dtype=torch.uint8, device=dev)
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        hist = torch.zeros(8, dtype=torch.int64, device=dev)
        while True:
            sel_idx = torch.empty(cap, dtype=torch.int64, device=dev)
            sel_st = torch.empty(cap, dtype=torch.uint8, device=dev)
            sel_rows = torch.empty((cap, width), dtype=torch.float64, device=dev)
            count.zero_()
            hist.zero_()
            for lo in range(0, n_pairs, chunk):
                hi = min(n_pairs, lo + chunk)
                pairs = pair_source(lo, hi)
                graph.pd_pi_batch(pairs, hop, flags=flags, res=res, out=out[: hi - lo], status=st[: hi - lo])
                engine.select_rows(out[: hi - lo], st[: hi - lo], index_base + lo, count, sel_idx, sel_st, sel_rows, hist=hist,
                                   keep_failed=keep_failed)
            k = int(count.item())                                        # the only synchronisation of the sweep
            if k <= cap:
                break
            del sel_idx, sel_st, sel_rows
            cap = k + (k >> 3) + 1024                                    # the store was too small: once more, with what it needs
        pieces.append((sel_idx[:k].cpu().numpy(), sel_rows[:k].cpu().numpy(), sel_st[:k].cpu().numpy(), hist.cpu().numpy()))
    if store is not None:
        return None
    return assemble(pieces, n_pairs, width)


def sweep_near(graph, index, hop, positions=None, flags=0, res=5, chunk=1 << 20):
    """The images of a whole negative list through the distance pre-filter (PI-C): only the non-edges with
    d(u,v) <= hop can have a non-zero row (A.6), so tlc_near_pairs lists those (with their numbers in the complement list of
    `index`, an engine.ComplementIndex) and only they go through tlc_pd_pi_batch.

    positions: None -> a row's index is its list number; or a callable mapping list numbers (int64 numpy) to row indices, e.g.
    the inverse of the reference's shuffle.  Returns SparseImages over len(index) rows whose status_counts cover the near
    pairs only; the rows left out are zero by the distance rule and their status bytes are not computed (`unclassified`)."""
    import torch
    width = res * res
    dev = torch.device("cuda", graph.device)
    with torch.cuda.device(dev):
        pairs, ranks = engine.near_pairs(index, hop)
        k = pairs.shape[0]
        out = torch.empty((max(k, 1), width), dtype=torch.float64, device=dev)
        st = torch.empty(max(k, 1), dtype=torch.uint8, device=dev)
        if k:
            graph.pd_pi_batch(pairs, hop, flags=flags, res=res, out=out[:k], status=st[:k])
        keep = (out[:k] != 0).any(1)
        rows = out[:k][keep].cpu().numpy()
        rk = ranks[keep].cpu().numpy()
        counts = np.bincount(st[:k].cpu().numpy(), minlength=8).astype(np.int64) if k else np.zeros(8, dtype=np.int64)
    idx = rk if positions is None else np.asarray(positions(rk), dtype=np.int64)
    images = SparseImages(len(index), width, idx, rows, np.zeros(len(rk), dtype=np.uint8), counts)
    images.unclassified = len(index) - k
    images.near_pairs = k
    return images


def assemble(pieces, n_rows, width):
    idx = np.concatenate([p[0] for p in pieces]) if pieces else np.zeros(0, dtype=np.int64)
    rows = np.concatenate([p[1] for p in pieces]) if pieces else np.zeros((0, width))
    status = np.concatenate([p[2] for p in pieces]) if pieces else np.zeros(0, dtype=np.uint8)
    counts = np.sum([p[3] for p in pieces], axis=0) if pieces else np.zeros(8, dtype=np.int64)
    return SparseImages(n_rows, width, idx, rows, status, counts)

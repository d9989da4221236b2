"""Device-side entry points of the PD/PI path: thin, typed wrappers over the C ABI (include/tlcgnn.h).

Everything here takes/returns torch CUDA tensors (device memory + stream plumbing only) and calls
libtlcgnn_hip.so through ctypes.  The reference-named drop-ins (sg2dgm/, baselines/, Knowledge_Distillation/)
are built on these.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import KEEP_ZERO_PERS, INCLUDE_ROOTS, NORM_EPS, PI_ORD0_EXT1, NO_EXT1, UNREACHABLE_100  # noqa: F401


class DeviceGraph:
    """graph2pi.__init__ (sg2dgm/riccidist2dgm.py:216-226): weighted graph resident on one GPU.

    rowptr/col/w: symmetric CSR (numpy), w = kappa + 1 > 0.
    """

    def __init__(self, rowptr, col, w, device=None):
        torch = _lib.require_gpu()
        self.device = torch.cuda.current_device() if device is None else int(device)
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        w = np.ascontiguousarray(w, dtype=np.float64)
        self.n_nodes = len(rowptr) - 1
        self.nnz = int(rowptr[-1])
        h = C.c_void_p()
        rc = _lib.lib().tlc_graph_create(C.c_int32(self.n_nodes), rowptr.ctypes.data_as(C.c_void_p),
                                         col.ctypes.data_as(C.c_void_p), w.ctypes.data_as(C.c_void_p),
                                         C.c_int(self.device), C.byref(h))
        _lib.check(rc, "tlc_graph_create")
        self._h = h
        self._inflight = []

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().tlc_graph_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- P2-P9 -------------------------------------------------------------------------------------------
    def pd_pi_batch(self, pairs, hop, flags=0, res=5, out=None, status=None, async_=False):
        """pairs: int32 CUDA tensor [E,2] -> (pi float64[E,res*res], status uint8[E]) on the current stream.
        async_: submit without making the current stream wait (tlc_pd_pi_batch_async): batches submitted back to back overlap;
        the outputs are complete for work that follows join() on its stream."""
        import torch
        assert pairs.is_cuda and pairs.dtype == torch.int32 and pairs.dim() == 2 and pairs.shape[1] == 2
        assert pairs.device.index == self.device, "pairs live on cuda:%s, the graph on cuda:%d" % (pairs.device.index, self.device)
        pairs = pairs.contiguous()
        E = pairs.shape[0]
        if out is None:
            out = torch.empty((E, res * res), dtype=torch.float64, device=pairs.device)
        if status is None:
            status = torch.empty((E,), dtype=torch.uint8, device=pairs.device)
        fn = _lib.lib().tlc_pd_pi_batch_async if async_ else _lib.lib().tlc_pd_pi_batch
        rc = fn(self._h, _lib.ptr(pairs), C.c_int64(E), C.c_int(hop), C.c_uint32(flags),
                C.c_int(res), _lib.ptr(out), _lib.ptr(status), _lib.stream_ptr(self.device))
        _lib.check(rc, "tlc_pd_pi_batch")
        if async_:
            self._inflight.append((pairs, out, status))          # (the buffers of a batch in flight must stay alive)
            del self._inflight[:-6]
        return out, status

    def join(self):
        """The current stream waits for every batch submitted with async_=True (nothing is waited for on the host)."""
        _lib.check(_lib.lib().tlc_pd_pi_batch_join(self._h, _lib.stream_ptr(self.device)), "tlc_pd_pi_batch_join")

    def stats(self):
        out = (C.c_int64 * 10)()
        rc = _lib.lib().tlc_pd_pi_batch_stats(self._h, C.cast(out, C.c_void_p), _lib.stream_ptr(self.device))
        _lib.check(rc, "tlc_pd_pi_batch_stats")
        v = list(out)
        return {"tier_small": v[0], "tier_medium": v[1], "tier_large": v[2], "tier_huge": v[3],
                "tier_mid": v[7], "tier_tiny": v[8], "tier_medium_many_pos": v[9],
                "induced_entries": v[4], "tie_fallback_sources": v[5], "chunks": v[6]}

    # ---- diagnostics (tests, A/B timing) --------------------------------------------------------------------------
    def set_option(self, name, value):
        """'extract' (ball-list extraction) / 'heavy' (its hub-row skipping) / 'tiny' (lane-per-subgraph kernel):
        1 on (default), 0 off.  Results do not depend on them (tests/test_gpu_extract.py)."""
        _lib.check(_lib.lib().tlc_debug_set_option(self._h, name.encode(), C.c_int(int(value))), "tlc_debug_set_option")

    def dc_stats(self):
        """(subgraphs whose cycle swap ran as the divide and conquer, subgraphs it gave back to the serial walk) since the last
        pd_pi_batch call began."""
        out = (C.c_longlong * 2)()
        _lib.check(_lib.lib().tlc_debug_dc_stats(self._h, C.cast(out, C.c_void_p), _lib.stream_ptr(self.device)), "tlc_debug_dc_stats")
        return int(out[0]), int(out[1])

    def tier_counts(self):
        """The tier lists of the last pd_pi_batch call as the device cut them (tlc_debug_tier_counts)."""
        out = (C.c_longlong * 8)()
        _lib.check(_lib.lib().tlc_debug_tier_counts(self._h, C.cast(out, C.c_void_p), _lib.stream_ptr(self.device)), "tlc_debug_tier_counts")
        return dict(zip(("small", "medium", "large", "huge", "mid", "tiny", "medium_many_pos", "medium_wide"), (int(v) for v in out)))

    # ---- measurement helpers (bench.py) -----------------------------------------------------------------------
    KERNELS = ["vicinity_count", "scan_bin", "vicinity_fill", "pd_tier_small", "pd_tier_medium", "pd_tier_large", "pd_tier_huge",
               "pd_tier_mid"]

    def set_timing(self, on=True, only=None):
        """on: bracket every kernel of later batches with HIP events; only=[names of KERNELS]: just those (the event records
        cost time themselves: 48 us of the 1.06 ms PubMed batch for all eight)."""
        mask = (1 if on else 0) if only is None else sum(1 << (self.KERNELS.index(k) + 1) for k in only)
        _lib.check(_lib.lib().tlc_pd_pi_batch_set_timing(self._h, C.c_int(mask)), "set_timing")

    def timings(self):
        """ms per kernel of the last batch (HIP events on the stream each kernel ran on); -1 = not launched."""
        out = (C.c_double * 8)()
        _lib.check(_lib.lib().tlc_pd_pi_batch_timings(self._h, C.cast(out, C.c_void_p), _lib.stream_ptr(self.device)), "timings")
        return dict(zip(self.KERNELS, list(out)[:8]))

    def timing_history(self, kernel, cap=64):
        """ms of `kernel` (a name of KERNELS) over the most recent chunks, oldest first -- read once after a run of batches
        that were enqueued without synchronising (the library keeps the events of the last 64 chunks)."""
        out = (C.c_double * cap)()
        n = C.c_int32(0)
        _lib.check(_lib.lib().tlc_pd_pi_batch_timing_history(self._h, C.c_int(self.KERNELS.index(kernel)), C.cast(out, C.c_void_p),
                                                             C.c_int32(cap), C.byref(n), _lib.stream_ptr(self.device)), "timing_history")
        return list(out)[:n.value]

    def sizes(self, n_pairs):
        n = np.zeros(n_pairs, dtype=np.int32)
        m2 = np.zeros(n_pairs, dtype=np.int32)
        _lib.check(_lib.lib().tlc_pd_pi_batch_sizes(self._h, n.ctypes.data_as(C.c_void_p), m2.ctypes.data_as(C.c_void_p),
                                                    C.c_int64(n_pairs), _lib.stream_ptr(self.device)), "sizes")
        return n, m2

    def _capacity_offsets(self, E, cap, dev):
        """arange(E + 1) * cap, kept for the last few (E, cap): a loop over equally sized chunks asks for the same two tensors every time
        (read-only: callers get the cached tensor itself)."""
        cache = self.__dict__.setdefault("_cap_offs", {})
        key = (E, cap, str(dev))
        t = cache.get(key)
        if t is None:
            import torch
            if len(cache) >= 8:
                cache.clear()
            t = cache[key] = torch.arange(E + 1, dtype=torch.int64, device=dev) * cap
        return t

    def vicinity_sizes(self, pairs, hop, flags=0):
        """-> (n int32[E], m int32[E]) CUDA tensors: |S| and the induced edge count of every pair's vicinity (tlc_vicinity_sizes)."""
        import torch
        assert pairs.device.index == self.device, "pairs live on cuda:%s, the graph on cuda:%d" % (pairs.device.index, self.device)
        pairs = pairs.contiguous()
        E = pairs.shape[0]
        n = torch.zeros(max(E, 1), dtype=torch.int32, device=pairs.device)
        m = torch.zeros(max(E, 1), dtype=torch.int32, device=pairs.device)
        rc = _lib.lib().tlc_vicinity_sizes(self._h, _lib.ptr(pairs), C.c_int64(E), C.c_int(hop), C.c_uint32(flags), _lib.ptr(n), _lib.ptr(m),
                                           _lib.stream_ptr(self.device))
        _lib.check(rc, "tlc_vicinity_sizes")
        return n[:E], m[:E]

    def vicinity_filtration(self, pairs, hop, flags=0, cap=None, edge_cap=None, zero=True, offsets=None):
        """-> (node_offs int64[E+1], ids int32[E*cap], f float64[E*cap], n int32[E], status uint8[E])
        with edge_cap: additionally (edge_offs int64[E+1], edges int32[E*edge_cap,2] local ids, m int32[E]).
        zero=False: the capacity buffers are not zero-filled (what lies beyond a pair's n / m entries is never read by a caller
        that slices by n / m: 290 MB of fill per 4 096 pairs at node_cap 512 / edge_cap 8 192).
        offsets=(node_offs, edge_offs, total_nodes, total_edges): the caller's own offsets (e.g. exact ones from vicinity_sizes +
        pack_offsets) instead of a capacity per pair; the outputs then have total_nodes / total_edges rows."""
        import torch
        assert pairs.device.index == self.device, "pairs live on cuda:%s, the graph on cuda:%d" % (pairs.device.index, self.device)
        pairs = pairs.contiguous()
        E = pairs.shape[0]
        dev = pairs.device
        mk = torch.zeros if zero else torch.empty
        if offsets is not None:
            offs, eoffs_x, tot_n, tot_m = offsets
            ids = mk(max(int(tot_n), 1), dtype=torch.int32, device=dev)
            f = mk(max(int(tot_n), 1), dtype=torch.float64, device=dev)
            n = torch.zeros(max(E, 1), dtype=torch.int32, device=dev)
            st = torch.zeros(max(E, 1), dtype=torch.uint8, device=dev)
            edges = mk((max(int(tot_m), 1), 2), dtype=torch.int32, device=dev)
            m = torch.zeros(max(E, 1), dtype=torch.int32, device=dev)
            rc = _lib.lib().tlc_vicinity_filtration(self._h, _lib.ptr(pairs), C.c_int64(E), C.c_int(hop), C.c_uint32(flags),
                                                    _lib.ptr(offs), _lib.ptr(ids), _lib.ptr(f), _lib.ptr(n), _lib.ptr(st),
                                                    _lib.ptr(eoffs_x), _lib.ptr(edges), _lib.ptr(m), _lib.stream_ptr(self.device))
            _lib.check(rc, "tlc_vicinity_filtration")
            return offs, ids, f, n[:E], st[:E], eoffs_x, edges, m[:E]
        cap = self.n_nodes if cap is None else int(cap)
        offs = self._capacity_offsets(E, cap, dev)
        ids = mk(max(E * cap, 1), dtype=torch.int32, device=dev)
        f = mk(max(E * cap, 1), dtype=torch.float64, device=dev)
        n = torch.zeros(max(E, 1), dtype=torch.int32, device=dev)
        st = torch.zeros(max(E, 1), dtype=torch.uint8, device=dev)
        eoffs = edges = m = None
        if edge_cap is not None:
            eoffs = self._capacity_offsets(E, int(edge_cap), dev)
            edges = mk((max(E * int(edge_cap), 1), 2), dtype=torch.int32, device=dev)
            m = torch.zeros(max(E, 1), dtype=torch.int32, device=dev)
        rc = _lib.lib().tlc_vicinity_filtration(self._h, _lib.ptr(pairs), C.c_int64(E), C.c_int(hop), C.c_uint32(flags),
                                                _lib.ptr(offs), _lib.ptr(ids), _lib.ptr(f), _lib.ptr(n), _lib.ptr(st),
                                                _lib.ptr(eoffs), _lib.ptr(edges), _lib.ptr(m), _lib.stream_ptr(self.device))
        _lib.check(rc, "tlc_vicinity_filtration")
        if edge_cap is not None:
            return offs, ids, f, n[:E], st[:E], eoffs, edges, m[:E]
        return offs, ids, f, n[:E], st[:E]


@_lib.on_device_of
def pack_vicinities(node_offs, ids, f, edge_offs, edges, node_ptr, edge_ptr, tot_n, tot_m, label=None, owners=True):
    """The capacity layout of vicinity_filtration -> packed (ids int64 [tot_n], f [tot_n], edges int32 [tot_m, 2], pair_of_node,
    pair_of_edge) at node_ptr / edge_ptr (tlc_pack_vicinities: one kernel, no host synchronisation)."""
    torch = _lib.require_gpu()
    dev = f.device
    E = node_ptr.numel() - 1
    out_ids = torch.empty(max(tot_n, 1), dtype=torch.int64, device=dev)
    out_f = torch.empty(max(tot_n, 1), dtype=torch.float64, device=dev)
    out_e = torch.empty((max(tot_m, 1), 2), dtype=torch.int32, device=dev)
    pn = torch.empty(max(tot_n, 1), dtype=torch.int64, device=dev) if owners else None
    pe = torch.empty(max(tot_m, 1), dtype=torch.int64, device=dev) if owners else None
    rc = _lib.lib().tlc_pack_vicinities(C.c_int64(E), _lib.ptr(node_offs), _lib.ptr(ids), _lib.ptr(f), _lib.ptr(edge_offs),
                                        _lib.ptr(edges), _lib.ptr(node_ptr), _lib.ptr(edge_ptr), _lib.ptr(label), _lib.ptr(out_ids),
                                        _lib.ptr(out_f), _lib.ptr(out_e), _lib.ptr(pn), _lib.ptr(pe), _lib.stream_ptr())
    _lib.check(rc, "tlc_pack_vicinities")
    return out_ids[:tot_n], out_f[:tot_n], out_e[:tot_m], (pn[:tot_n] if owners else None), (pe[:tot_m] if owners else None)


@_lib.on_device_of
def stack_batch(node_ptr, edge_ptr, edges, f=None):
    """A packed batch (node_ptr / edge_ptr int64[B+1], edges int32 [m,2] local ids, f float64 [n]) -> (edge_index int64 [2, m+n] with
    global ids and the n self loops last, x float32 [n,1] or None): the operands of Teacher_Model.forward, one launch
    (tlc_stack_batch); the sizes are the tensors' own, nothing is read back."""
    torch = _lib.require_gpu()
    B, m = node_ptr.numel() - 1, int(edges.shape[0])
    if f is None:
        raise ValueError("stack_batch: f (its length is the node count) is needed")
    n = int(f.numel())
    ei = torch.empty((2, m + n), dtype=torch.int64, device=edges.device)
    x = torch.empty((n, 1), dtype=torch.float32, device=edges.device)
    rc = _lib.lib().tlc_stack_batch(C.c_int64(B), _lib.ptr(node_ptr), _lib.ptr(edge_ptr), _lib.ptr(edges.contiguous()),
                                    _lib.ptr(f.contiguous()), C.c_int64(n), C.c_int64(m), _lib.ptr(ei), _lib.ptr(x), _lib.stream_ptr())
    _lib.check(rc, "tlc_stack_batch")
    return ei, x


@_lib.on_device_of
def pack_offsets(n, m):
    """Per-pair counts of vicinity_filtration (int32[E], negative = did not fit) -> (node_ptr int64[E+1], edge_ptr int64[E+1],
    totals int64[4] = min n, min m, sum n, sum m): tlc_pack_offsets, one launch; vicinities without an edge are left out."""
    torch = _lib.require_gpu()
    E = n.numel()
    dev = n.device
    node_ptr = torch.empty(E + 1, dtype=torch.int64, device=dev)
    edge_ptr = torch.empty(E + 1, dtype=torch.int64, device=dev)
    totals = torch.empty(4, dtype=torch.int64, device=dev)
    rc = _lib.lib().tlc_pack_offsets(C.c_int64(E), _lib.ptr(n.contiguous()), _lib.ptr(m.contiguous()), _lib.ptr(node_ptr), _lib.ptr(edge_ptr),
                                     _lib.ptr(totals), _lib.stream_ptr())
    _lib.check(rc, "tlc_pack_offsets")
    return node_ptr, edge_ptr, totals


@_lib.on_device_of
def pd_from_filtration(node_offs, edge_offs, edges, f, flags=0, want_rank=True):
    """Batched perturb_filter_function + Union_find + Accelerate_PD (sg2dgm/accelerated_PD.py:6-178).

    All arguments CUDA tensors: node_offs/edge_offs int64[B+1], edges int32[sum m,2], f float64[sum n].
    Returns dict(up, down, one, ext0, counts, edge_rank) of CUDA tensors (layout: include/tlcgnn.h).
    """
    torch = _lib.require_gpu()
    dev = f.device
    B = node_offs.numel() - 1
    sn, sm = int(f.numel()), int(edges.shape[0])
    up = torch.zeros((max(sn, 1), 2), dtype=torch.float64, device=dev)
    down = torch.zeros((max(sn, 1), 2), dtype=torch.float64, device=dev)
    one = torch.zeros((max(sm, 1), 2), dtype=torch.float64, device=dev)
    ext0 = torch.zeros((max(B, 1), 2), dtype=torch.float64, device=dev)
    counts = torch.zeros((max(B, 1), 4), dtype=torch.int32, device=dev)
    rank = torch.zeros(max(sm, 1), dtype=torch.int32, device=dev) if want_rank else None
    edges = edges.contiguous()
    rc = _lib.lib().tlc_pd_from_filtration(C.c_int32(B), _lib.ptr(node_offs.contiguous()), _lib.ptr(edge_offs.contiguous()),
                                           _lib.ptr(edges), _lib.ptr(f.contiguous()), C.c_uint32(flags), _lib.ptr(up),
                                           _lib.ptr(down), _lib.ptr(one), _lib.ptr(ext0), _lib.ptr(counts),
                                           _lib.ptr(rank), _lib.stream_ptr())
    _lib.check(rc, "tlc_pd_from_filtration")
    return dict(up=up, down=down, one=one, ext0=ext0[:B], counts=counts[:B], edge_rank=None if rank is None else rank[:sm])


@_lib.on_device_of
def pi_raster(offs, pts, res=5):
    """Batched PersistenceImager(resolution=res).transform (sg2dgm/PersistenceImager.pyx:352-388).

    offs int64[B+1], pts float64[sum k, 2] (birth, death) CUDA tensors -> float64[B, res*res].
    """
    torch = _lib.require_gpu()
    B = offs.numel() - 1
    out = torch.empty((max(B, 1), res * res), dtype=torch.float64, device=offs.device)
    pts = pts.contiguous()
    rc = _lib.lib().tlc_pi_raster(C.c_int32(B), _lib.ptr(offs.contiguous()), _lib.ptr(pts) if pts.numel() else None,
                                  C.c_int(res), _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "tlc_pi_raster")
    return out[:B]


@_lib.on_device_of
def pi_raster_wgrad(offs, pts, grad_img, res=5):
    """d images / d points of the reference's differentiable imager (pimg.py:354-400: through the weights only) contracted with
    grad_img float64[B, res*res] -> float64[N, 2] (tlc_pi_raster_wgrad)."""
    torch = _lib.require_gpu()
    B = offs.numel() - 1
    pts = pts.contiguous()
    grad_img = grad_img.contiguous()
    out = torch.zeros((max(pts.shape[0], 1), 2), dtype=torch.float64, device=offs.device)
    rc = _lib.lib().tlc_pi_raster_wgrad(C.c_int32(B), C.c_int64(pts.shape[0]), _lib.ptr(offs.contiguous()), _lib.ptr(pts) if pts.numel() else None,
                                        C.c_int(res), _lib.ptr(grad_img), _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "tlc_pi_raster_wgrad")
    return out[:pts.shape[0]]


# size tiers of the PD kernel (csrc/tlc_kernels.h) and what the library's timing slots bracket
TIER_LIMITS = [("pd_tier_small", 64, 128), ("pd_tier_mid", 128, 256), ("pd_tier_medium", 512, 1024), ("pd_tier_large", 2048, 4096)]
TINY_LIMITS = (16, 24)          # TLC_T_NMAX / TLC_T_MMAX: lane-per-subgraph kernel (plain image batches at resolution 5)
MEDIUM_MANY_POS = 120           # TLC_MH_MIN_POS: MEDIUM-sized vicinities with at least this many Pos edges (m - n + 1)
MEDIUM_COMPACT = (384, 512)     # TLC_C_NMAX / TLC_C_MMAX: the compact configuration of the MEDIUM-sized tiers (beyond it: MEDWIDE)


def tier_of(n, m2, tiny=True):
    """Which kernel's TIMING SLOT covers each pair, from (|S|, induced directed entries), mirroring tlc_scan_bin and run_chunk:
    'pd_tier_small' = the wavefront-per-subgraph SMALL kernel only -- the pairs of the lane-per-subgraph kernel are
    'pd_tier_tiny' (no slot of its own); 'pd_tier_medium' = the MEDIUM-sized vicinities with many Pos edges or beyond the compact
    kernel configuration (the launch that slot brackets), the rest of the MEDIUM tier is 'pd_tier_medium_rest' (not bracketed);
    '' for pairs finished early.
    So bytes summed over the pairs of a name and the time of that name's slot cover the same work."""
    n = np.asarray(n)
    m = np.asarray(m2) // 2
    out = np.full(n.shape, "pd_tier_huge", dtype=object)
    for name, nm, mm in reversed(TIER_LIMITS):
        out[(n <= nm) & (m <= mm)] = name
    med = out == "pd_tier_medium"
    out[med & (m - n + 1 < MEDIUM_MANY_POS) & (n <= MEDIUM_COMPACT[0]) & (m <= MEDIUM_COMPACT[1])] = "pd_tier_medium_rest"
    if tiny:
        out[(n <= TINY_LIMITS[0]) & (m <= TINY_LIMITS[1])] = "pd_tier_tiny"
    out[n <= 0] = ""
    return out


def algorithmic_bytes(rowptr, col, pairs, hop, res=5):
    """SURVEY.md 8(d) byte model per pair (host-side accounting through the C ABI; no GPU needed)."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    out = np.zeros(len(pairs), dtype=np.float64)
    rc = _lib.lib().tlc_pd_pi_algorithmic_bytes(C.c_int32(len(rowptr) - 1), rowptr.ctypes.data_as(C.c_void_p),
                                                col.ctypes.data_as(C.c_void_p), pairs.ctypes.data_as(C.c_void_p),
                                                C.c_int64(len(pairs)), C.c_int(hop), C.c_int(res),
                                                out.ctypes.data_as(C.c_void_p))
    _lib.check(rc, "tlc_pd_pi_algorithmic_bytes")
    return out


# ---- the negative list of loaddatas.py:44-45 and the sparse image store ----------------------------
class ComplementIndex:
    """Non-edges of a symmetric 0/1 adjacency in the order of `sp.triu(sp.csr_matrix(1. - adj.toarray())).nonzero()`
    (loaddatas.py:44), addressed by list number on the device -- no N x N matrix, no [n_neg, 2] array.

    rowptr/col: numpy CSR of the symmetric adjacency, columns ascending and unique inside a row (checked here)."""

    def __init__(self, rowptr, col, device=None):
        torch = _lib.require_gpu()
        rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        n = len(rowptr) - 1
        if len(col):
            inner = np.ones(len(col), dtype=bool)
            inner[rowptr[:-1][rowptr[:-1] < len(col)]] = False          # first entry of every non-empty row
            if not np.all(np.diff(col.astype(np.int64))[inner[1:]] > 0):
                raise ValueError("ComplementIndex: CSR columns must be ascending and unique inside every row")
        dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
        self.n_nodes = n
        self.rowptr = torch.from_numpy(rowptr).to(dev)
        self.col = torch.from_numpy(col).to(dev) if len(col) else torch.zeros(1, dtype=torch.int32, device=dev)
        self.row_start = torch.empty(n + 1, dtype=torch.int64, device=dev)
        with torch.cuda.device(dev):
            rc = _lib.lib().tlc_complement_rows(C.c_int32(n), _lib.ptr(self.rowptr), _lib.ptr(self.col),
                                                _lib.ptr(self.row_start), _lib.stream_ptr())
        _lib.check(rc, "tlc_complement_rows")
        self.n_neg = int(self.row_start[-1].item())

    def __len__(self):
        return self.n_neg

    def pairs(self, ranks=None, first=0, count=None, out=None):
        """int32 CUDA [count, 2]: the pairs numbered ranks[...] (int64 CUDA tensor) or first .. first+count-1."""
        import torch
        if ranks is not None:
            assert ranks.is_cuda and ranks.dtype == torch.int64
            ranks = ranks.contiguous()
            count = ranks.numel()
        elif count is None:
            count = self.n_neg - first
        if out is None:
            out = torch.empty((count, 2), dtype=torch.int32, device=self.rowptr.device)
        with torch.cuda.device(self.rowptr.device):
            rc = _lib.lib().tlc_complement_pairs(C.c_int32(self.n_nodes), _lib.ptr(self.rowptr), _lib.ptr(self.col),
                                                 _lib.ptr(self.row_start), _lib.ptr(ranks), C.c_int64(first), C.c_int64(count),
                                                 _lib.ptr(out), _lib.stream_ptr())
        _lib.check(rc, "tlc_complement_pairs")
        return out


def near_pairs(index, hop, cap=None):
    """Non-adjacent pairs u <= v with d(u,v) <= hop (tlc_near_pairs) of a ComplementIndex: (pairs int32 CUDA [k,2], ranks int64
    CUDA [k]) with ranks = their numbers in the complement list.  Unordered."""
    import torch
    dev = index.rowptr.device
    cap = max(1 << 16, 64 * index.n_nodes) if cap is None else int(cap)
    with torch.cuda.device(dev):
        count = torch.zeros(1, dtype=torch.int64, device=dev)
        while True:
            ranks = torch.empty(cap, dtype=torch.int64, device=dev)
            pairs = torch.empty((cap, 2), dtype=torch.int32, device=dev)
            count.zero_()
            rc = _lib.lib().tlc_near_pairs(C.c_int32(index.n_nodes), _lib.ptr(index.rowptr), _lib.ptr(index.col),
                                           _lib.ptr(index.row_start), C.c_int(hop), C.c_int64(cap), _lib.ptr(count), _lib.ptr(ranks),
                                           _lib.ptr(pairs), _lib.stream_ptr())
            _lib.check(rc, "tlc_near_pairs")
            k = int(count.item())
            if k <= cap:
                return pairs[:k], ranks[:k]
            del ranks, pairs
            cap = k + 1024


@_lib.on_device_of
def select_rows(pi, status, index_base, count, out_idx, out_status, out_rows, hist=None, keep_failed=False):
    """Append the non-zero rows of an image block (keep_failed: and the zero rows with status != 0) to a sparse store and add
    the block's status bytes to `hist` (int64 CUDA [8]) -- tlc_select_rows.  count: int64 CUDA scalar tensor the caller zeroed
    (read it after synchronising; it may exceed the capacity)."""
    rc = _lib.lib().tlc_select_rows(C.c_int64(pi.shape[0]), C.c_int32(pi.shape[1]), _lib.ptr(pi), _lib.ptr(status),
                                    C.c_int64(index_base), C.c_int64(out_idx.shape[0]), C.c_uint32(1 if keep_failed else 0),
                                    _lib.ptr(count), _lib.ptr(hist), _lib.ptr(out_idx), _lib.ptr(out_status), _lib.ptr(out_rows),
                                    _lib.stream_ptr())
    _lib.check(rc, "tlc_select_rows")


def ollivier_ricci_sinkhorn(rowptr, col, edges, alpha=0.5, reg=0.1, max_iter=1000, stop_thr=1e-9, device=None, want_iters=False):
    """Ollivier-Ricci curvature of the given edges with the Sinkhorn transport distance -- GraphRicciCurvature's
    OllivierRicci(G, alpha, method="Sinkhorn") as loaddatas.py:105-123 calls it (tlc_ollivier_ricci_sinkhorn).

    rowptr/col: numpy CSR of the symmetric, loop-free, unit-weight graph (columns ascending); edges: int [E,2] adjacent pairs.
    Returns float64 numpy [E] (and the iteration counts)."""
    torch = _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device() if device is None else int(device))
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    edges = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
    E, n = len(edges), len(rowptr) - 1
    if E == 0:
        return (np.zeros(0), np.zeros(0, dtype=np.int32)) if want_iters else np.zeros(0)
    deg = np.diff(rowptr).astype(np.int64)
    ok = (edges >= 0).all(1) & (edges < n).all(1)
    if not ok.all():
        raise ValueError("ollivier_ricci_sinkhorn: edge endpoint out of range")
    # the 0/1/2/3 hop-distance rule of the kernel holds for ADJACENT pairs only: every edge must be in the CSR (self pairs get 0)
    keys = np.repeat(np.arange(n, dtype=np.int64), deg) * n + col.astype(np.int64)
    want = edges[:, 0].astype(np.int64) * n + edges[:, 1].astype(np.int64)
    loops = edges[:, 0] == edges[:, 1]
    pos = np.searchsorted(keys, want)
    found = (pos < len(keys)) & (keys[np.minimum(pos, max(len(keys) - 1, 0))] == want) if len(keys) else np.zeros(E, dtype=bool)
    if not (found | loops).all():
        raise ValueError("ollivier_ricci_sinkhorn: every pair must be an edge of the CSR (columns ascending)")
    na = np.where(ok, deg[np.clip(edges[:, 0], 0, n - 1)] + 1, 1)
    nb = np.where(ok, deg[np.clip(edges[:, 1], 0, n - 1)] + 1, 1)
    prod = na * nb
    big = (prod > 8192) | (na + nb > 256)                         # what the wavefront kernel leaves to the workgroup kernel
    max_support = int(max(2, (na + nb).max()))
    max_product = int(prod[big].max()) if big.any() else 16
    slots = int(min(max(int(big.sum()), 1), 512))
    work_bytes = 16 + ((4 * E + 15) // 16) * 16 + slots * ((((max_product + 3) // 4 + 15) // 16) * 16)
    with torch.cuda.device(dev):
        d_rowptr, d_col = torch.from_numpy(rowptr).to(dev), torch.from_numpy(col).to(dev)
        d_edges = torch.from_numpy(edges).to(dev)
        kappa = torch.empty(E, dtype=torch.float64, device=dev)
        iters = torch.empty(E, dtype=torch.int32, device=dev)
        work = torch.empty(work_bytes, dtype=torch.uint8, device=dev)
        rc = _lib.lib().tlc_ollivier_ricci_sinkhorn(C.c_int32(n), _lib.ptr(d_rowptr), _lib.ptr(d_col), C.c_int64(E), _lib.ptr(d_edges),
                                                    C.c_double(alpha), C.c_double(reg), C.c_int32(max_iter), C.c_double(stop_thr),
                                                    _lib.ptr(kappa), _lib.ptr(iters), _lib.ptr(work), C.c_int64(work_bytes),
                                                    C.c_int32(max_support), C.c_int64(max_product), _lib.stream_ptr())
        _lib.check(rc, "tlc_ollivier_ricci_sinkhorn")
        out = kappa.cpu().numpy()
        it = iters.cpu().numpy()
    if np.isnan(out).any():
        raise _lib.TlcError("tlc_ollivier_ricci_sinkhorn: an edge exceeded the workspace (max_support / max_product)")
    return (out, it) if want_iters else out

"""Typed wrappers over the LP-forward / PDGNN entry points of the C ABI (include/tlcgnn.h).

torch tensors in, torch tensors out; every call enqueues hand-written HIP kernels on the current stream.
Forward only (no autograd): the path in scope is the link-prediction FORWARD.
"""
import ctypes as C

from . import _lib


def _f32(t):
    import torch
    assert t.is_cuda and t.dtype == torch.float32, "expected a float32 CUDA tensor"
    return t.contiguous()


@_lib.on_device_of
def gcn_norm_csr(edge_index, num_nodes):
    """gcn_norm of GCNConv(cached=True) (Knowledge_Distillation/PD_conv.py:35-70) as CSR by target.

    edge_index: int64 CUDA [2,E] (row 0 source, row 1 target).  Returns (rowptr int32[N+1], col int32[nnz], val f32[nnz]).
    """
    torch = _lib.require_gpu()
    assert edge_index.is_cuda and edge_index.dtype == torch.int64 and edge_index.shape[0] == 2
    ei = edge_index.contiguous()
    E = ei.shape[1]
    dev = ei.device
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    col = torch.empty(E + num_nodes, dtype=torch.int32, device=dev)
    val = torch.empty(E + num_nodes, dtype=torch.float32, device=dev)
    nnz = torch.zeros(1, dtype=torch.int32, device=dev)
    rc = _lib.lib().tlc_gcn_norm_csr(C.c_int32(num_nodes), C.c_int64(E), _lib.ptr(ei), _lib.ptr(rowptr), _lib.ptr(col),
                                     _lib.ptr(val), _lib.ptr(nnz), _lib.stream_ptr())
    _lib.check(rc, "tlc_gcn_norm_csr")
    k = int(nnz.item())
    return rowptr, col[:k].contiguous(), val[:k].contiguous()


@_lib.on_device_of
def csr_by_target(edge_index, num_nodes):
    """remove_self_loops + add_self_loops + grouping by target (Knowledge_Distillation/gat_conv.py:146-152), the structure alone:
    (rowptr int32[N+1], col int32[E+N] of which the first rowptr[N] are written).  tlc_csr_by_target: temporaries from the caching
    allocator, no host read -- the per-batch form of gcn_norm_csr (whose one-off build allocates, waits and frees)."""
    torch = _lib.require_gpu()
    assert edge_index.is_cuda and edge_index.dtype == torch.int64 and edge_index.shape[0] == 2
    ei = edge_index.contiguous()
    E, dev = ei.shape[1], ei.device
    rowptr = torch.empty(num_nodes + 1, dtype=torch.int32, device=dev)
    col = torch.empty(E + num_nodes, dtype=torch.int32, device=dev)
    work = torch.empty(3 * num_nodes + E + 1, dtype=torch.int32, device=dev)             # [temporaries | nnz]
    rc = _lib.lib().tlc_csr_by_target(C.c_int32(num_nodes), C.c_int64(E), _lib.ptr(ei), _lib.ptr(rowptr), _lib.ptr(col),
                                      _lib.ptr(work[3 * num_nodes + E:]), _lib.ptr(work), _lib.stream_ptr())
    _lib.check(rc, "tlc_csr_by_target")
    return rowptr, col


@_lib.on_device_of
def pdgnn_forward(x, edge_index, params, edge_ptr, res=5, hidden=32, rowptr=None, col=None, tiles=None):
    """Teacher_Model.forward(compute_loss=False) without gradients as one library call (tlc_pdgnn_forward): x float32 [n,1],
    edge_index int64 [2, m+n] (self loops last), params = the 20 float32 tensors in the header's order, edge_ptr int64 [B+1]
    -> (points float32 [m,2], images float64 [B, res*res]).  rowptr / col / tiles: the structure of a batch the caller holds."""
    torch = _lib.require_gpu()
    n, E, B = int(x.shape[0]), int(edge_index.shape[1]), int(edge_ptr.numel()) - 1
    ei = edge_index.contiguous()
    L = _lib.lib()
    nbytes = int(L.tlc_pdgnn_forward_work_bytes(C.c_int32(n), C.c_int64(E), C.c_int32(hidden)))
    if nbytes < 0:
        raise ValueError("pdgnn_forward: needs n >= 1 nodes and an edge_index that ends in the n self loops (train_Teacher_Model.py:43-44)")
    work = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
    points = torch.empty((E - n, 2), dtype=torch.float32, device=x.device)
    img = torch.empty((max(B, 1), res * res), dtype=torch.float64, device=x.device)
    keep = [_f32(p) for p in params]
    arr = (C.c_void_p * 20)(*[p.data_ptr() for p in keep])
    nt = 0 if tiles is None else tiles.numel() - 1
    rc = L.tlc_pdgnn_forward(C.c_int32(n), C.c_int64(E), _lib.ptr(ei), _lib.ptr(_f32(x)), C.c_int32(hidden), arr, C.c_int64(B),
                             _lib.ptr(edge_ptr.contiguous()), C.c_int32(res), _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(tiles), C.c_int32(nt),
                             _lib.ptr(work), C.c_int64(nbytes), _lib.ptr(points), _lib.ptr(img), _lib.stream_ptr())
    _lib.check(rc, "tlc_pdgnn_forward")
    return points, img[:B]


@_lib.on_device_of
def gemm(a, b, bias=None, relu=False, out=None):
    """C = A @ B (+bias)(ReLU) on the f32 MFMA; A [M,K], B [K,N<=128] float32 CUDA."""
    torch = _lib.require_gpu()
    a, b = _f32(a), _f32(b)
    M, K = a.shape
    K2, N = b.shape
    assert K == K2
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=a.device)
    rc = _lib.lib().tlc_gemm_f32(C.c_int32(M), C.c_int32(N), C.c_int32(K), _lib.ptr(a), _lib.ptr(b),
                                 _lib.ptr(_f32(bias)) if bias is not None else None, C.c_int(1 if relu else 0),
                                 _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "tlc_gemm_f32")
    return out


class SparseRows:
    """CSR of a feature matrix (the zeros of bag-of-words / TF-IDF features dropped), built once: rowptr int32[M+1],
    col int32[nnz], val f32[nnz] on the matrix's device.  The conversion is a data-format step (torch index plumbing)."""

    def __init__(self, x):
        import torch
        x = _f32(x)
        self.shape = tuple(x.shape)
        nz = x != 0
        counts = nz.sum(dim=1, dtype=torch.int64)
        self.rowptr = torch.zeros(x.shape[0] + 1, dtype=torch.int32, device=x.device)
        self.rowptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        idx = nz.nonzero(as_tuple=False)                                  # row-major: rows ascending, columns ascending
        self.nnz = int(idx.shape[0])
        if self.nnz:
            self.col = idx[:, 1].to(torch.int32).contiguous()
            self.val = x[nz].contiguous()
        else:                                                              # (an all-zero matrix: the arrays still need an address)
            self.col = torch.zeros(1, dtype=torch.int32, device=x.device)
            self.val = torch.zeros(1, dtype=torch.float32, device=x.device)

    @property
    def density(self):
        return self.nnz / float(max(self.shape[0] * self.shape[1], 1))


SPARSE_GEMM_MAX_K = 636          # a 64-column slice of B in LDS: K * 256 B + 1 KiB <= 160 KiB (narrower slices take more: sparse_gemm_fits)
SPARSE_GEMM_MAX_NNZ = 1 << 29    # the kernel addresses entries by 32-bit BYTE offsets through a buffer descriptor (include/tlcgnn.h)


def _check_sparse_rows(xs):
    """Entries beyond 2^29 would read back as zeros (offsets behind the descriptor's range): refused, never truncated."""
    if xs.nnz >= SPARSE_GEMM_MAX_NNZ:
        raise _lib.TlcError("sparse feature projection: %d stored entries, the kernel takes fewer than 2^29 (use the dense ops.gemm)" % xs.nnz)


def sparse_gemm_fits(k, n):
    """Whether tlc_spgemm_csr_dense_f32 takes a [*, k] @ [k, n] product: its column slice of B ([k][<= 64], equal widths, a
    multiple of four: 52 + 48 for n = 100) and 1 KiB must fit the 160 KiB of LDS."""
    slices = (n + 63) // 64
    sw = (((n + slices - 1) // slices) + 3) & ~3
    return k > 0 and n > 0 and k * sw * 4 + 1024 <= 160 * 1024


@_lib.on_device_of
def sparse_gemm(xs, b, bias=None, relu=False, out=None):
    """C = X @ B (+bias)(ReLU) for X given as SparseRows: the zeros of X cost nothing (tlc_spgemm_csr_dense_f32; fp32 sums in an
    order of its own)."""
    torch = _lib.require_gpu()
    b = _f32(b)
    M, K = xs.shape
    assert b.shape[0] == K
    _check_sparse_rows(xs)
    N = b.shape[1]
    if out is None:
        out = torch.empty((M, N), dtype=torch.float32, device=b.device)
    rc = _lib.lib().tlc_spgemm_csr_dense_f32(C.c_int32(M), C.c_int32(K), C.c_int32(N), _lib.ptr(xs.rowptr), _lib.ptr(xs.col),
                                             _lib.ptr(xs.val), _lib.ptr(b), _lib.ptr(_f32(bias)) if bias is not None else None,
                                             C.c_int(1 if relu else 0), _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "tlc_spgemm_csr_dense_f32")
    return out


@_lib.on_device_of
def spmm(rowptr, col, val, x, bias=None, relu=False, out=None, renorm=False):
    """Y = act(CSR @ X + bias): the normalised scatter-add of GCNConv as a row-owned gather; renorm=True also applies
    emb.renorm_(2, 0, 1) (TLCGNN.py:48) to every output row in the same pass."""
    torch = _lib.require_gpu()
    x = _f32(x)
    n = rowptr.numel() - 1
    k = x.shape[1]
    if out is None:
        out = torch.empty((n, k), dtype=torch.float32, device=x.device)
    rc = _lib.lib().tlc_spmm_csr_f32(C.c_int32(n), _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val), _lib.ptr(x), C.c_int32(k),
                                     _lib.ptr(_f32(bias)) if bias is not None else None, C.c_int((1 if relu else 0) | (2 if renorm else 0)),
                                     _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "tlc_spmm_csr_f32")
    return out


_GCN2_WS = {}


def gcn2_encode(rowptr, col, val, x, w1, b1, w2, b2, relu=True, renorm=False, out=None, x_sparse=None):
    """Net.encode in eval mode behind one library call (tlc_gcn2_encode_f32): relu(A (relu(A (x w1) + b1)) w2 + b2), rows
    renormalised when renorm=True (TLCGNN.py:19-26,48).  Same four kernels as gemm / spmm / gemm / spmm; the scratch for
    the intermediates is kept per (device, stream, sizes): two forwards in flight on two streams do not share it.
    x_sparse: SparseRows of x -- the first projection then runs over the stored entries (tlc_gcn2_encode_csr_f32)."""
    torch = _lib.require_gpu()
    w1, w2 = _f32(w1), _f32(w2)
    if x_sparse is None:
        x = _f32(x)
        n, f_in = x.shape[0], x.shape[1]
    else:
        n, f_in = x_sparse.shape
        _check_sparse_rows(x_sparse)
    hidden, d = w1.shape[1], w2.shape[1]
    dev = w1.device
    key = (dev, torch.cuda.current_stream(dev).cuda_stream, n, hidden, d)
    ws = _GCN2_WS.get(key)
    if ws is None:
        if len(_GCN2_WS) >= 4:
            _GCN2_WS.clear()
        ws = _GCN2_WS[key] = torch.empty(((2 * hidden + d) * n + 12,), dtype=torch.float32, device=dev)
    if out is None:
        out = torch.empty((n, d), dtype=torch.float32, device=dev)
    flags = C.c_int((1 if relu else 0) | (2 if renorm else 0))
    pb1 = _lib.ptr(_f32(b1)) if b1 is not None else None
    pb2 = _lib.ptr(_f32(b2)) if b2 is not None else None
    with torch.cuda.device(dev):
        if x_sparse is None:
            rc = _lib.lib().tlc_gcn2_encode_f32(C.c_int32(n), _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val), _lib.ptr(x), C.c_int32(f_in),
                                                _lib.ptr(w1), pb1, C.c_int32(hidden), _lib.ptr(w2), pb2, C.c_int32(d),
                                                flags, _lib.ptr(ws), _lib.ptr(out), _lib.stream_ptr(dev))
        else:
            rc = _lib.lib().tlc_gcn2_encode_csr_f32(C.c_int32(n), _lib.ptr(rowptr), _lib.ptr(col), _lib.ptr(val), _lib.ptr(x_sparse.rowptr),
                                                    _lib.ptr(x_sparse.col), _lib.ptr(x_sparse.val), C.c_int32(f_in),
                                                    _lib.ptr(w1), pb1, C.c_int32(hidden), _lib.ptr(w2), pb2, C.c_int32(d),
                                                    flags, _lib.ptr(ws), _lib.ptr(out), _lib.stream_ptr(dev))
    _lib.check(rc, "tlc_gcn2_encode_f32" if x_sparse is None else "tlc_gcn2_encode_csr_f32")
    return out


@_lib.on_device_of
def renorm_rows_(emb):
    """emb.renorm_(2, 0, 1) in place (baselines/TLCGNN.py:48)."""
    emb_c = _f32(emb)
    assert emb_c.data_ptr() == emb.data_ptr(), "renorm_rows_ needs a contiguous tensor (in place)"
    rc = _lib.lib().tlc_renorm_rows_f32(C.c_int32(emb.shape[0]), C.c_int32(emb.shape[1]), _lib.ptr(emb), _lib.stream_ptr())
    _lib.check(rc, "tlc_renorm_rows_f32")
    return emb


@_lib.on_device_of
def lp_decode(pairs, emb, pi, w1, b1, w2, b2, out=None):
    """Fused Net.decode tail (baselines/TLCGNN.py:52-61).  pairs int32 [E,2], emb f32 [N,D], pi [E,P] float64 (the raw output of
    pd_pi_batch: cast to float32 on load) or float32 (cast once by the caller, as Net._tables does: the reference's
    torch.Tensor(PI) of :52-53 hoisted out of the per-decode path; same values, half the bytes)."""
    torch = _lib.require_gpu()
    assert pairs.dtype == torch.int32 and pi.dtype in (torch.float64, torch.float32)
    E = pairs.shape[0]
    if out is None:
        out = torch.empty(E, dtype=torch.float32, device=emb.device)
    fn = _lib.lib().tlc_lp_decode_fused if pi.dtype == torch.float64 else _lib.lib().tlc_lp_decode_fused_f32
    rc = fn(C.c_int64(E), _lib.ptr(pairs.contiguous()), _lib.ptr(_f32(emb)), C.c_int32(emb.shape[1]),
            _lib.ptr(pi.contiguous()), C.c_int32(pi.shape[1]), _lib.ptr(_f32(w1)), _lib.ptr(_f32(b1)),
            _lib.ptr(_f32(w2).reshape(-1)), _lib.ptr(_f32(b2)), _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "tlc_lp_decode_fused")
    return out


@_lib.on_device_of
def gat_layer(rowptr, src, x, wl, att, wij, bias, prelu_slope=-1.0, out=None):
    """One PDGNN layer (Knowledge_Distillation/gat_conv.py:113-216) on a CSR-by-target batch."""
    torch = _lib.require_gpu()
    x = _f32(x)
    n, c_in = x.shape
    c_out = wl.shape[0]
    if out is None:
        out = torch.empty((n, 2 * c_out), dtype=torch.float32, device=x.device)
    work = torch.empty(max(n, 1) * (3 * c_out + 4) + c_in * c_out + c_out * (2 * c_out + 4), dtype=torch.float32, device=x.device)
    rc = _lib.lib().tlc_gat_layer_fwd(C.c_int32(n), _lib.ptr(rowptr), _lib.ptr(src), _lib.ptr(x), C.c_int32(c_in),
                                      C.c_int32(c_out), _lib.ptr(_f32(wl)), _lib.ptr(_f32(att).reshape(-1)), _lib.ptr(_f32(wij)),
                                      _lib.ptr(_f32(bias)), C.c_float(prelu_slope), _lib.ptr(work), _lib.ptr(out),
                                      _lib.stream_ptr())
    _lib.check(rc, "tlc_gat_layer_fwd")
    return out


GAT_TILE_NODES = 192          # GT_TM of csrc/gat_forward.hip


def gat_tiles(rowptr, col, n, tile_nodes=GAT_TILE_NODES):
    """Cuts a block-diagonal batch (CSR by target: rowptr int32 [n+1], col = the sources) into self-contained tiles for
    tlc_gat_layer_tiled_fwd: int32 [T+1] node offsets of tiles of at most `tile_nodes` consecutive nodes, cut only at positions no
    edge crosses -- or None when the batch has no such cuts close enough together (one big graph: the two-kernel layer serves it).
    All on the device (tlc_gat_tile_cut; one host read, of the tile count)."""
    if not 2 <= int(tile_nodes) <= GAT_TILE_NODES:
        raise ValueError("gat_tiles: tile_nodes %r; gat_tile_kernel's LDS tile holds at most %d rows" % (tile_nodes, GAT_TILE_NODES))
    torch = _lib.require_gpu()
    n = int(n)
    if n == 0:
        return None
    dev = rowptr.device
    work = torch.empty(n // 32 + 6, dtype=torch.int32, device=dev)
    tiles = torch.empty(2 * n // tile_nodes + 3, dtype=torch.int32, device=dev)
    nt = C.c_int32(0)
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().tlc_gat_tile_cut(C.c_int32(n), _lib.ptr(rowptr), _lib.ptr(col), C.c_int32(tile_nodes), _lib.ptr(work),
                                               _lib.ptr(tiles), C.byref(nt), _lib.stream_ptr(dev)), "tlc_gat_tile_cut")
    return tiles[:nt.value + 1] if nt.value > 0 else None


@_lib.on_device_of
def gat_layer_tiled(rowptr, src, tiles, x, wl, att, wij, bias, prelu_slope=-1.0, out=None):
    """One PDGNN layer (gat_conv.py:113-216) on a block-diagonal batch cut by gat_tiles: the node rows stay in LDS
    (tlc_gat_layer_tiled_fwd).  Shapes it does not take raise _lib.TlcError(TLC_ERR_UNSUPPORTED): callers check `gat_tiled_ok`."""
    torch = _lib.require_gpu()
    x = _f32(x)
    n, c_in = x.shape
    c_out = wl.shape[0]
    if out is None:
        out = torch.empty((n, 2 * c_out), dtype=torch.float32, device=x.device)
    n2 = 2 * c_out + 4
    work = torch.empty(c_in * c_out + c_out * n2 + c_in * n2 + 2 * c_out + 8, dtype=torch.float32, device=x.device)
    rc = _lib.lib().tlc_gat_layer_tiled_fwd(C.c_int32(n), _lib.ptr(rowptr), _lib.ptr(src), C.c_int32(tiles.numel() - 1), _lib.ptr(tiles),
                                            _lib.ptr(x), C.c_int32(c_in), C.c_int32(c_out), _lib.ptr(_f32(wl)),
                                            _lib.ptr(_f32(att).reshape(-1)), _lib.ptr(_f32(wij)), _lib.ptr(_f32(bias)),
                                            C.c_float(prelu_slope), _lib.ptr(work), _lib.ptr(out), _lib.stream_ptr())
    _lib.check(rc, "tlc_gat_layer_tiled_fwd")
    return out


def gat_tiled_ok(c_in, c_out):
    return c_in in (1, 64) and c_out in (16, 32)


@_lib.on_device_of
def edge_head(src, dst, x, w5, b5, prelu_slope, w6, b6, out=None):
    """Edge head of Teacher_Model.forward (Knowledge_Distillation/Teacher_model.py:54-59) -> f32 [E,2]."""
    torch = _lib.require_gpu()
    x = _f32(x)
    E = src.numel()
    if out is None:
        out = torch.empty((E, 2), dtype=torch.float32, device=x.device)
    n, c, hidden = x.shape[0], x.shape[1], w5.shape[0]
    work = torch.empty((n + c) * 2 * hidden, dtype=torch.float32, device=x.device)      # per-node projections + packed W5
    rc = _lib.lib().tlc_edge_head_fwd(C.c_int64(E), _lib.ptr(src), _lib.ptr(dst), _lib.ptr(x), C.c_int32(c),
                                      _lib.ptr(_f32(w5)), _lib.ptr(_f32(b5)), C.c_int32(hidden), C.c_float(prelu_slope),
                                      _lib.ptr(_f32(w6)), _lib.ptr(_f32(b6)), _lib.ptr(out), C.c_int32(n), _lib.ptr(work),
                                      _lib.stream_ptr())
    _lib.check(rc, "tlc_edge_head_fwd")
    return out


@_lib.on_device_of
def gat_layer_bwd(rowptr, src, x, wl, att, wij, prelu_slope, out, gout, need_gx=True):
    """Gradients of `gat_layer` (what autograd runs through Knowledge_Distillation/gat_conv.py:113-216):
    -> (gX or None, gWl, gatt, gWij, gbias).  `out` is the forward's output (read for the fused PReLU only)."""
    torch = _lib.require_gpu()
    x = _f32(x)
    n, c_in = x.shape
    c_out = wl.shape[0]
    dev = x.device
    gx = torch.empty((n, c_in), dtype=torch.float32, device=dev) if need_gx else None
    gwl = torch.empty((c_out, c_in), dtype=torch.float32, device=dev)
    gatt = torch.empty(c_out, dtype=torch.float32, device=dev)
    gwij = torch.empty((c_out, 2 * c_out), dtype=torch.float32, device=dev)
    gbias = torch.empty(2 * c_out, dtype=torch.float32, device=dev)
    work = torch.empty(max(n, 1) * (8 * c_out + 5) + 2 * c_out * c_out + c_in * c_out + 4 + c_out * (2 * c_out + 4), dtype=torch.float32, device=dev)
    rc = _lib.lib().tlc_gat_layer_bwd(C.c_int32(n), _lib.ptr(rowptr), _lib.ptr(src), _lib.ptr(x), C.c_int32(c_in), C.c_int32(c_out),
                                      _lib.ptr(_f32(wl)), _lib.ptr(_f32(att).reshape(-1)), _lib.ptr(_f32(wij)), C.c_float(prelu_slope),
                                      _lib.ptr(_f32(out)) if out is not None else None, _lib.ptr(_f32(gout)), _lib.ptr(gx),
                                      _lib.ptr(gwl), _lib.ptr(gatt), _lib.ptr(gwij), _lib.ptr(gbias), _lib.ptr(work), _lib.stream_ptr())
    _lib.check(rc, "tlc_gat_layer_bwd")
    return gx, gwl, gatt, gwij, gbias


@_lib.on_device_of
def edge_head_bwd(src, dst, x, w5, b5, prelu_slope, w6, gpd):
    """Gradients of `edge_head` (Teacher_model.py:54-59): -> (gX, gW5, gb5, gW6, gb6)."""
    torch = _lib.require_gpu()
    x = _f32(x)
    E = src.numel()
    n, c, hidden = x.shape[0], x.shape[1], w5.shape[0]
    dev = x.device
    gx = torch.zeros((n, c), dtype=torch.float32, device=dev)
    gw5 = torch.empty((hidden, 2 * c), dtype=torch.float32, device=dev)
    gb5 = torch.empty(hidden, dtype=torch.float32, device=dev)
    gw6 = torch.empty((2, hidden), dtype=torch.float32, device=dev)
    gb6 = torch.empty(2, dtype=torch.float32, device=dev)
    work = torch.empty(max(E, 1) * (2 * c + 2 * hidden), dtype=torch.float32, device=dev)
    rc = _lib.lib().tlc_edge_head_bwd(C.c_int64(E), _lib.ptr(src), _lib.ptr(dst), _lib.ptr(x), C.c_int32(c), _lib.ptr(_f32(w5)),
                                      _lib.ptr(_f32(b5)), C.c_int32(hidden), C.c_float(prelu_slope), _lib.ptr(_f32(w6)),
                                      _lib.ptr(_f32(gpd)), _lib.ptr(gx), _lib.ptr(gw5), _lib.ptr(gb5), _lib.ptr(gw6), _lib.ptr(gb6),
                                      _lib.ptr(work), _lib.stream_ptr())
    _lib.check(rc, "tlc_edge_head_bwd")
    return gx, gw5, gb5, gw6, gb6


REDUCE = {"add": 0, "sum": 0, "mean": 1, "min": 2, "max": 3}


@_lib.on_device_of
def scatter(src, index, dim_size, reduce="sum"):
    """torch_scatter.scatter(src, index, dim=0, dim_size=dim_size, reduce=...) (message_passing.py:292).

    src float32 CUDA [E, ...] (trailing dims flattened), index int64 CUDA [E] -> [dim_size, ...]; empty segments are 0."""
    torch = _lib.require_gpu()
    src = _f32(src)
    E = src.shape[0]
    tail = tuple(src.shape[1:])
    k = 1
    for d in tail:
        k *= int(d)
    k = max(k, 1)
    out = torch.empty((dim_size,) + tail, dtype=torch.float32, device=src.device)
    r = REDUCE[reduce]
    cnt = torch.empty(max(dim_size, 1), dtype=torch.int32, device=src.device) if r != 0 else None
    rc = _lib.lib().tlc_scatter_f32(C.c_int64(E), _lib.ptr(index.contiguous()), _lib.ptr(src), C.c_int32(k), C.c_int(r),
                                    C.c_int32(dim_size), _lib.ptr(out), _lib.ptr(cnt), _lib.stream_ptr())
    _lib.check(rc, "tlc_scatter_f32")
    return out


@_lib.on_device_of
def w2_partial_matching(xoff, X, yoff, Y, order=2, want_grad=True, max_points=None):
    """PDGNN's diagram loss for a batch of (predicted, target) diagram pairs (Knowledge_Distillation/wasserstein.py:198-379 with
    num_models = 1): xoff / yoff int64[B+1] offsets, X / Y float64[., 2] CUDA tensors.
    -> dict(loss[B], wxy[B], wxd[B], assign[sum n] (target index or -1 = diagonal), grad[sum n, 2] (d loss / d X), status[B])."""
    torch = _lib.require_gpu()
    B = xoff.numel() - 1
    dev = X.device
    X = X.to(torch.float64).contiguous()
    Y = Y.to(torch.float64).contiguous()
    nx = int(X.shape[0])
    if B > 0:
        # (offsets beyond the arrays would be read out of bounds on the device; one host read, with max_points below)
        ends = torch.stack([xoff[-1], yoff[-1], (xoff[1:] - xoff[:-1]).max()]).tolist()
        if ends[0] > nx or ends[1] > int(Y.shape[0]):
            raise ValueError("w2_partial_matching: offsets end at (%d, %d) but X / Y hold (%d, %d) points"
                             % (ends[0], ends[1], nx, int(Y.shape[0])))
        if max_points is None:
            max_points = int(ends[2])
    elif max_points is None:
        max_points = 0
    loss = torch.zeros(max(B, 1), dtype=torch.float64, device=dev)
    wxy = torch.zeros_like(loss)
    wxd = torch.zeros_like(loss)
    assign = torch.full((max(nx, 1),), -1, dtype=torch.int32, device=dev)
    grad = torch.zeros((max(nx, 1), 2), dtype=torch.float64, device=dev) if want_grad else None
    status = torch.zeros(max(B, 1), dtype=torch.uint8, device=dev)
    rc = _lib.lib().tlc_w2_partial_matching(C.c_int32(B), _lib.ptr(xoff.to(torch.int64).contiguous()), _lib.ptr(X) if nx else None,
                                            _lib.ptr(yoff.to(torch.int64).contiguous()), _lib.ptr(Y) if Y.numel() else None,
                                            C.c_int(order), C.c_int32(max_points), _lib.ptr(loss), _lib.ptr(wxy), _lib.ptr(wxd),
                                            _lib.ptr(assign), _lib.ptr(grad), _lib.ptr(status), _lib.stream_ptr())
    _lib.check(rc, "tlc_w2_partial_matching")
    return dict(loss=loss[:B], wxy=wxy[:B], wxd=wxd[:B], assign=assign[:nx], grad=None if grad is None else grad[:nx], status=status[:B])


@_lib.on_device_of
def w2_inference_matching(xoff, X, yoff, Y, order=2, want_grad=False, max_points=None):
    """The evaluation distance of PDGNN (`wasserstein_distance_inference`, Knowledge_Distillation/wasserstein.py:93-195: both
    diagrams may use the diagonal) for a batch of (predicted, target) pairs.
    -> dict(loss[B], wxy[B], wxd[B], wyd[B], assign_x[sum n], assign_y[sum m], grad or None, status[B])."""
    torch = _lib.require_gpu()
    B = xoff.numel() - 1
    dev = X.device
    X = X.to(torch.float64).contiguous()
    Y = Y.to(torch.float64).contiguous()
    xoff = xoff.to(torch.int64).contiguous()
    yoff = yoff.to(torch.int64).contiguous()
    nx, ny = int(X.shape[0]), int(Y.shape[0])
    if B > 0:
        ends = torch.stack([xoff[-1], yoff[-1]]).tolist()
        if ends[0] > nx or ends[1] > ny:
            raise ValueError("w2_inference_matching: offsets end at (%d, %d) but X / Y hold (%d, %d) points" % (ends[0], ends[1], nx, ny))
    if max_points is None:
        max_points = int(((xoff[1:] - xoff[:-1]) + (yoff[1:] - yoff[:-1])).max().item()) if B > 0 else 0
    loss = torch.zeros(max(B, 1), dtype=torch.float64, device=dev)
    wxy, wxd, wyd = torch.zeros_like(loss), torch.zeros_like(loss), torch.zeros_like(loss)
    ax = torch.full((max(nx, 1),), -1, dtype=torch.int32, device=dev)
    ay = torch.full((max(ny, 1),), -1, dtype=torch.int32, device=dev)
    grad = torch.zeros((max(nx, 1), 2), dtype=torch.float64, device=dev) if want_grad else None
    status = torch.zeros(max(B, 1), dtype=torch.uint8, device=dev)
    rc = _lib.lib().tlc_w2_inference_matching(C.c_int32(B), _lib.ptr(xoff), _lib.ptr(X) if nx else None, _lib.ptr(yoff),
                                              _lib.ptr(Y) if ny else None, C.c_int(order), C.c_int32(max_points), _lib.ptr(loss),
                                              _lib.ptr(wxy), _lib.ptr(wxd), _lib.ptr(wyd), _lib.ptr(ax), _lib.ptr(ay), _lib.ptr(grad),
                                              _lib.ptr(status), _lib.stream_ptr())
    _lib.check(rc, "tlc_w2_inference_matching")
    return dict(loss=loss[:B], wxy=wxy[:B], wxd=wxd[:B], wyd=wyd[:B], assign_x=ax[:nx], assign_y=ay[:ny],
                grad=None if grad is None else grad[:nx], status=status[:B])


def capture(fn, warmup=2):
    """HIP graph of a forward closure: `fn` (C-ABI launches on the current stream, outputs in caller-held or graph-pool buffers,
    no host synchronisation inside) is warmed up on a side stream, captured once, and replayed with `.replay()`.
    torch.cuda.CUDAGraph is hipStreamBeginCapture / hipGraphLaunch plumbing.  Every LP-forward entry point of the C ABI is
    capturable (tests/test_gpu_lp_forward.py).  Measured on the PubMed forward (five kernels, 70 us of kernel time): replay
    86 us vs 78 us launched kernel by kernel from Python -- ROCm's graph launch costs more than the launch gaps it removes, so
    bench.py does not use it."""
    torch = _lib.require_gpu()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(warmup):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fn()
    return graph

"""ctypes front end of the CPU oracle (oracle/tlc_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product package (tlc-gnn_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# TLC_ORACLE_ASAN=1: load the AddressSanitizer/UBSan build (make -C oracle asan); the process must have been started with
# LD_PRELOAD=$(gcc -print-file-name=libasan.so) -- tests/test_oracle_asan.py does that for a child pytest
ASAN = os.environ.get("TLC_ORACLE_ASAN") == "1"
LIB_PATH = os.path.join(HERE, "libtlc_oracle_asan.so" if ASAN else "libtlc_oracle.so")

ST_OK, ST_MISSING_NODE, ST_DISCONNECTED, ST_ZERO_RANGE, ST_NO_TREE_EDGE = range(5)
KEEP_ZERO_PERS, INCLUDE_ROOTS, NORM_EPS, PI_ORD0_EXT1, NO_EXT1, UNREACHABLE_100 = 1, 2, 4, 8, 16, 32

_lib = None


def build(force=False):
    src = os.path.join(HERE, "tlc_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s", "-B"] + (["asan"] if ASAN else []))
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.tlco_algorithmic_bytes.restype = C.c_double
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def pi_raster(offs, pts, res=5):
    offs = np.ascontiguousarray(offs, dtype=np.int64)
    pts = np.ascontiguousarray(pts, dtype=np.float64).reshape(-1, 2)
    n = len(offs) - 1
    out = np.empty((n, res * res), dtype=np.float64)
    rc = lib().tlco_pi_raster(C.c_int32(n), _p(offs, C.c_int64), _p(pts, C.c_double), C.c_int(res),
                              _p(out, C.c_double))
    assert rc == 0
    return out


def pd_from_filtration(node_offs, edge_offs, edges, f, flags=0):
    """Returns dict of per-graph lists: up, down, one (float64[k,2]), ext0[B,2], counts[B,4], edge_rank."""
    node_offs = np.ascontiguousarray(node_offs, dtype=np.int64)
    edge_offs = np.ascontiguousarray(edge_offs, dtype=np.int64)
    edges = np.ascontiguousarray(edges, dtype=np.int32).reshape(-1, 2)
    f = np.ascontiguousarray(f, dtype=np.float64)
    B = len(node_offs) - 1
    sn, sm = int(node_offs[-1]), int(edge_offs[-1])
    up = np.zeros((sn, 2))
    down = np.zeros((sn, 2))
    one = np.zeros((max(sm, 1), 2))
    ext0 = np.zeros((B, 2))
    counts = np.zeros((B, 4), dtype=np.int32)
    rank = np.zeros(max(sm, 1), dtype=np.int32)
    rc = lib().tlco_pd_from_filtration(C.c_int32(B), _p(node_offs, C.c_int64), _p(edge_offs, C.c_int64),
                                       _p(edges, C.c_int32), _p(f, C.c_double), C.c_uint32(flags),
                                       _p(up, C.c_double), _p(down, C.c_double), _p(one, C.c_double),
                                       _p(ext0, C.c_double), _p(counts, C.c_int32), _p(rank, C.c_int32))
    assert rc == 0
    return dict(up=up, down=down, one=one, ext0=ext0, counts=counts, edge_rank=rank[:sm])


def pd_pi_batch(rowptr, col, w, pairs, hop, flags=0, res=5, n_threads=1):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float64)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    E = len(pairs)
    out = np.empty((E, res * res), dtype=np.float64)
    st = np.empty(E, dtype=np.uint8)
    used = lib().tlco_pd_pi_batch(C.c_int32(len(rowptr) - 1), _p(rowptr, C.c_int32), _p(col, C.c_int32),
                                  _p(w, C.c_double), _p(pairs, C.c_int32), C.c_int64(E), C.c_int(hop),
                                  C.c_uint32(flags), C.c_int(res), _p(out, C.c_double), _p(st, C.c_uint8),
                                  C.c_int(n_threads))
    return out, st, used


def vicinity_filtration(rowptr, col, w, pairs, hop, flags=0, cap=None, edge_cap=None):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    w = np.ascontiguousarray(w, dtype=np.float64)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    E = len(pairs)
    n_nodes = len(rowptr) - 1
    cap = n_nodes if cap is None else cap
    node_offs = np.arange(E + 1, dtype=np.int64) * cap
    ids = np.zeros(E * cap, dtype=np.int32)
    f = np.zeros(E * cap, dtype=np.float64)
    n = np.zeros(E, dtype=np.int32)
    m = np.zeros(E, dtype=np.int32)
    st = np.zeros(E, dtype=np.uint8)
    edge_offs = edges = None
    if edge_cap is not None:
        edge_offs = np.arange(E + 1, dtype=np.int64) * edge_cap
        edges = np.zeros((max(E * edge_cap, 1), 2), dtype=np.int32)
    lib().tlco_vicinity_filtration(C.c_int32(n_nodes), _p(rowptr, C.c_int32), _p(col, C.c_int32), _p(w, C.c_double),
                                   _p(pairs, C.c_int32), C.c_int64(E), C.c_int(hop), C.c_uint32(flags),
                                   _p(node_offs, C.c_int64), _p(ids, C.c_int32), _p(f, C.c_double), _p(n, C.c_int32),
                                   _p(m, C.c_int32), _p(st, C.c_uint8), _p(edge_offs, C.c_int64), _p(edges, C.c_int32))
    if edge_cap is not None:
        return node_offs, ids, f, n, m, st, edge_offs, edges
    return node_offs, ids, f, n, m, st


def algorithmic_bytes(rowptr, col, pairs, hop, res=5):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    col = np.ascontiguousarray(col, dtype=np.int32)
    pairs = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
    sn, sm = C.c_int64(0), C.c_int64(0)
    b = lib().tlco_algorithmic_bytes(C.c_int32(len(rowptr) - 1), _p(rowptr, C.c_int32), _p(col, C.c_int32),
                                     _p(pairs, C.c_int32), C.c_int64(len(pairs)), C.c_int(hop), C.c_int(res),
                                     C.byref(sn), C.byref(sm))
    return float(b), int(sn.value), int(sm.value)


def max_threads():
    return int(lib().tlco_max_threads())


def complement_pairs_dense(adj_dense):
    """The reference's negative list before the shuffle, loaddatas.py:44: `sp.triu(sp.csr_matrix(1. - adj.toarray())).nonzero()`
    -- row-major over x <= y with 1 - adj[x, y] != 0.  Dense: test sizes only.  (Checker for tlc_complement_pairs.)"""
    c = np.triu(1.0 - np.asarray(adj_dense, dtype=np.float64))
    x, y = np.nonzero(c)
    return np.stack([x, y], 1).astype(np.int64)

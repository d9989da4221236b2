"""CPU: host logic, the C-ABI library surface, generators, sharding helpers (no GPU compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_c_abi_library_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    from tlc_gnn_amd import _lib
    L = _lib.lib()
    header = open(os.path.join(ROOT, "include", "tlcgnn.h")).read()
    declared = sorted(set(re.findall(r"\b(tlc_[a-z0-9_]+)\s*\(", header)))
    assert declared, "no declarations parsed"
    for sym in declared:
        assert hasattr(L, sym), "libtlcgnn_hip.so does not export %s" % sym
    assert sorted(_lib.SYMBOLS) == declared
    assert L.tlc_version().startswith(b"tlcgnn-hip")


def test_product_path_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from tlc_gnn_amd import engine, _lib
    rowptr = np.array([0, 1, 2], dtype=np.int32)
    with pytest.raises(_lib.TlcError):
        engine.DeviceGraph(rowptr, np.array([1, 0], dtype=np.int32), np.array([1.0, 1.0]))
    # the C ABI itself reports "no device" instead of computing on the CPU
    h = C.c_void_p()
    col = np.array([1, 0], dtype=np.int32)
    w = np.array([1.0, 1.0])
    rc = _lib.lib().tlc_graph_create(C.c_int32(2), rowptr.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p),
                                     w.ctypes.data_as(C.c_void_p), C.c_int(0), C.byref(h))
    assert rc == 3 and not h.value          # TLC_ERR_NO_DEVICE


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "tlc-gnn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in txt.replace("no oracle", ""), "%s mentions the oracle" % f


def test_get_adj_split_matches_reference_golden_g7():
    import scipy.sparse as sp
    from tlc_gnn_amd import loaddatas
    d = np.load(os.path.join(G, "adj_split.npz"))
    n, edges = int(d["n_nodes"]), d["edges"]
    a = sp.coo_matrix((np.ones(len(edges)), (edges[:, 0], edges[:, 1])), shape=(n, n))
    parts = loaddatas.get_adj_split(sp.csr_matrix(a + a.T), val_prop=0.05, test_prop=0.1, seed=1234)
    names = ["train_edges", "train_edges_false", "val_edges", "val_edges_false", "test_edges", "test_edges_false"]
    for name, p in zip(names, parts):
        assert np.array_equal(np.asarray(p), d[name]), name


def test_get_adj_split_ppi_proportions_match_reference_golden_g7b():
    """The PPI configuration's split (val_prop = test_prop = 0.2, baselines/TLCGNN.py:73-75) on the three graphs of golden G7b."""
    import scipy.sparse as sp
    from tlc_gnn_amd import loaddatas
    d = np.load(os.path.join(G, "adj_split_ppi.npz"))
    names = ["train_edges", "train_edges_false", "val_edges", "val_edges_false", "test_edges", "test_edges_false"]
    for gi in range(int(d["n_graphs"])):
        n, edges = int(d["g%d_n_nodes" % gi]), d["g%d_edges" % gi]
        a = sp.coo_matrix((np.ones(len(edges)), (edges[:, 0], edges[:, 1])), shape=(n, n))
        parts = loaddatas.get_adj_split(sp.csr_matrix(a + a.T), val_prop=0.2, test_prop=0.2, seed=1234)
        for name, p in zip(names, parts):
            assert np.array_equal(np.asarray(p), d["g%d_%s" % (gi, name)]), (gi, name)


def test_remove_pairs_both_directions_like_list_remove():
    import torch
    from tlc_gnn_amd.baselines.TLCGNN import remove_pairs_both_directions
    ei = torch.tensor([[0, 1, 1, 2, 0, 1, 3, 2], [1, 0, 2, 1, 1, 0, 2, 3]])
    out = remove_pairs_both_directions(ei, np.array([[0, 1], [2, 3], [7, 7]]))
    # reference: edge_list.remove([0,1]); edge_list.remove([1,0]) removes the FIRST occurrences only
    ref = np.array(ei).T.tolist()
    for e in ([0, 1], [2, 3]):
        ref.remove(e)
        ref.remove(e[::-1])
    assert out.T.tolist() == ref


def test_synthetic_generator_is_deterministic_and_shaped():
    from tlc_gnn_amd import synth
    n, e, k, hop, f = synth.shaped_graph("PubMed", scale=0.05)
    n2, e2, k2, _, _ = synth.shaped_graph("PubMed", scale=0.05)
    assert np.array_equal(e, e2) and np.array_equal(k, k2)
    assert len(np.unique(e, axis=0)) == len(e) and (e[:, 0] < e[:, 1]).all()
    rowptr, col, w = synth.edges_to_csr(n, e, k)
    assert rowptr[-1] == 2 * len(e) and (w > 0).all()
    ricci = synth.synthetic_curvature(e)
    assert len(ricci) == 2 * len(e) and ricci == sorted(ricci)


def test_algorithmic_bytes_model_matches_oracle_accounting():
    from tlc_gnn_amd import synth, engine
    from oracle import oracle
    n, e, k, hop, _ = synth.shaped_graph("PubMed", scale=0.1)
    rowptr, col, w = synth.edges_to_csr(n, e, k)
    pairs = np.concatenate([e[:300], np.random.RandomState(0).randint(0, n, size=(100, 2))]).astype(np.int32)
    b = engine.algorithmic_bytes(rowptr, col, pairs, 2)
    tot, sn, sm = oracle.algorithmic_bytes(rowptr, col, pairs, 2)
    assert abs(b.sum() - tot) < 1e-6
    assert (b >= 8 + 200).all()


def test_tier_classification_mirrors_kernel_constants():
    from tlc_gnn_amd import engine
    hdr = open(os.path.join(ROOT, "tlc-gnn_amd", "csrc", "tlc_kernels.h")).read()
    vals = {k: int(v) for k, v in re.findall(r"#define (TLC_[SDML]_[NM]MAX) (\d+)", hdr)}
    assert engine.TIER_LIMITS == [("pd_tier_small", vals["TLC_S_NMAX"], vals["TLC_S_MMAX"]),
                                  ("pd_tier_mid", vals["TLC_D_NMAX"], vals["TLC_D_MMAX"]),
                                  ("pd_tier_medium", vals["TLC_M_NMAX"], vals["TLC_M_MMAX"]),
                                  ("pd_tier_large", vals["TLC_L_NMAX"], vals["TLC_L_MMAX"])]
    tiny = {k: int(v) for k, v in re.findall(r"#define (TLC_T_[NM]MAX|TLC_MH_MIN_POS) (\d+)", hdr)}
    assert engine.TINY_LIMITS == (tiny["TLC_T_NMAX"], tiny["TLC_T_MMAX"]) and engine.MEDIUM_MANY_POS == tiny["TLC_MH_MIN_POS"]
    comp = {k: int(v) for k, v in re.findall(r"#define (TLC_C_[NM]MAX) (\d+)", hdr)}
    assert engine.MEDIUM_COMPACT == (comp["TLC_C_NMAX"], comp["TLC_C_MMAX"])
    n = np.array([0, 10, 16, 17, 16, 64, 65, 128, 129, 512, 300, 300, 513, 3000, 384, 385, 300])
    m2 = np.array([0, 20, 48, 48, 50, 256, 10, 512, 10, 2048, 2 * (299 + 119), 2 * (299 + 120), 10, 10, 2 * 512, 2 * 512, 2 * 513])
    t = engine.tier_of(n, m2)
    # (the vicinities beyond the compact configuration -- 512 nodes, 385 nodes, 513 edges -- go with the many-Pos ones)
    assert t.tolist() == ["", "pd_tier_tiny", "pd_tier_tiny", "pd_tier_small", "pd_tier_small", "pd_tier_small", "pd_tier_mid", "pd_tier_mid",
                          "pd_tier_medium_rest", "pd_tier_medium", "pd_tier_medium_rest", "pd_tier_medium", "pd_tier_large", "pd_tier_huge",
                          "pd_tier_medium", "pd_tier_medium", "pd_tier_medium"]
    assert engine.tier_of(n, m2, tiny=False).tolist()[1:3] == ["pd_tier_small", "pd_tier_small"]


def test_shard_helpers():
    from tlc_gnn_amd import dist as tdist
    for n, w in ((10, 3), (19717, 8), (5, 8), (0, 2)):
        b = [tdist.shard_bounds(n, w, r) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1
    cost = np.array([1, 1, 1, 100, 1, 1, 1, 1])
    bounds = tdist.shard_pairs_by_cost(cost, 2)
    assert bounds[0] == 0 and bounds[-1] == len(cost) and 0 < bounds[1] < len(cost)


def test_simplex_filter_keys_are_the_reference_expressions():
    """perturb_filter_function (accelerated_PD.py:6-23): node entries {'old','new'}, edge entries keyed (a, b) with
    asc = hi + (lo + 1)*1e-6, desc = lo - (101 - hi)*1e-6 evaluated in that association in float64 (KAT: scalar CPython)."""
    from tlc_gnn_amd.sg2dgm.accelerated_PD import build_simplex_filter
    rs = np.random.RandomState(3)
    f = np.concatenate([rs.rand(40), [0.0, 1.0, 1 / 3, 2 / 3, 1 / 3]]).tolist()
    nodes = ["n%d" % i for i in range(len(f))]
    edges = [(nodes[a], nodes[b]) for a, b in rs.randint(0, len(f), size=(120, 2)).tolist()]
    sf = build_simplex_filter(nodes, f, edges)
    assert list(sf)[:len(nodes)] == nodes                       # nodes first, in order; then the edges
    for nd, v in zip(nodes, f):
        assert sf[nd] == {'old': v, 'new': v}
    for a, b in edges:
        hi, lo = max(sf[a]['old'], sf[b]['old']), min(sf[a]['old'], sf[b]['old'])
        assert sf[(a, b)]['asc'] == hi + (lo + 1) * 1e-6 and sf[(a, b)]['desc'] == lo - (101 - hi) * 1e-6
        assert type(sf[(a, b)]['asc']) is float


def _bipartite_case(seed=5):
    import torch
    g = torch.Generator().manual_seed(seed)
    n_src, n_dst, E, k = 23, 11, 160, 6
    xs = torch.randn(n_src, k, generator=g)
    xd = torch.randn(n_dst, k, generator=g) + 100.0          # (far from xs: a swapped element cannot pass)
    a_l = torch.randn(n_src, generator=g)
    a_r = torch.randn(n_dst, generator=g)
    ei = torch.stack([torch.randint(0, n_src, (E,), generator=g), torch.randint(0, n_dst, (E,), generator=g)])
    return xs, xd, a_l, a_r, ei


def test_message_passing_tuple_arguments_follow_the_reference():
    """`propagate(ei, x=(x_src, x_dst), alpha=(a_l, a_r))` -- the reference's own hot-path call form
    (Knowledge_Distillation/gat_conv.py:160-161): `_j` reads element 0 through edge_index[0], `_i` element 1 through
    edge_index[1] (message_passing.py:147-158), sizes come from the two elements.  Distinct tensors with different
    node counts; the aggregate is replaced by torch so the whole propagate runs on the CPU."""
    import torch
    from tlc_gnn_amd.Knowledge_Distillation.message_passing import MessagePassing
    xs, xd, a_l, a_r, ei = _bipartite_case()

    class Conv(MessagePassing):
        def __init__(self, flow):
            super().__init__(aggr="add", flow=flow, node_dim=0)
            self.seen = {}

        def message(self, x_j, x_i, alpha_j, alpha_i, index, size_i, size_j):
            self.seen = dict(x_j=x_j, x_i=x_i, alpha_j=alpha_j, alpha_i=alpha_i, index=index, size_i=size_i, size_j=size_j)
            return (x_j - x_i) * (alpha_j + alpha_i).view(-1, 1)

        def aggregate(self, inputs, index, dim_size=None):
            return torch.zeros(dim_size, inputs.size(1)).index_add_(0, index, inputs)

    c = Conv("source_to_target")
    out = c.propagate(ei, x=(xs, xd), alpha=(a_l, a_r))
    s = c.seen
    assert torch.equal(s["x_j"], xs[ei[0]]) and torch.equal(s["x_i"], xd[ei[1]])
    assert torch.equal(s["alpha_j"], a_l[ei[0]]) and torch.equal(s["alpha_i"], a_r[ei[1]])
    assert torch.equal(s["index"], ei[1]) and s["size_i"] == xd.size(0) and s["size_j"] == xs.size(0)
    ref = torch.zeros(xd.size(0), xs.size(1)).index_add_(0, ei[1], (xs[ei[0]] - xd[ei[1]]) * (a_l[ei[0]] + a_r[ei[1]]).view(-1, 1))
    assert out.shape == ref.shape and torch.allclose(out, ref)
    # an explicit size that contradicts an element is an error (:115-122)
    with pytest.raises(ValueError):
        c.propagate(ei, size=(xs.size(0) + 1, xd.size(0)), x=(xs, xd), alpha=(a_l, a_r))
    # flow 'target_to_source': the suffix still picks the element (0 for _j, 1 for _i); the rows swap (i, j) = (0, 1)
    c = Conv("target_to_source")
    ei_t = torch.stack([ei[1], ei[0]])                      # row 0 = i indexes element 1, row 1 = j indexes element 0
    c.propagate(ei_t, x=(xs, xd), alpha=(a_l, a_r))
    s = c.seen
    assert torch.equal(s["x_j"], xs[ei_t[1]]) and torch.equal(s["x_i"], xd[ei_t[0]]) and torch.equal(s["index"], ei_t[0])
    # a single tensor is used for both sides, a None element is passed through untouched
    class One(MessagePassing):
        def __init__(self):
            super().__init__(aggr="add", node_dim=0)

        def message(self, x_j, e_i):
            assert e_i is None
            return x_j

        def aggregate(self, inputs, index, dim_size=None):
            return torch.zeros(dim_size, inputs.size(1)).index_add_(0, index, inputs)

    sq = torch.stack([ei[0] % 11, ei[1]])
    out = One().propagate(sq, x=xd, e=(xd, None))
    assert torch.allclose(out, torch.zeros(11, xd.size(1)).index_add_(0, sq[1], xd[sq[0]]))


def test_sparse_gemm_fits_follows_the_kernels_lds_budget():
    """ops.sparse_gemm_fits restates the launcher's rule (lp_forward.hip, tlc_spgemm_csr_dense_f32): equal column slices of at most 64,
    a multiple of four, K * slice * 4 B + 1 KiB within 160 KiB."""
    from tlc_gnn_amd import ops
    assert ops.sparse_gemm_fits(500, 100) and ops.sparse_gemm_fits(640, 100)          # 52-column slices
    assert ops.sparse_gemm_fits(636, 64) and not ops.sparse_gemm_fits(637, 64)        # a full 64-column slice
    assert ops.sparse_gemm_fits(636, 128) and not ops.sparse_gemm_fits(1433, 100)     # Cora's 1 433 features: the dense kernel
    assert ops.sparse_gemm_fits(ops.SPARSE_GEMM_MAX_K, 64) and not ops.sparse_gemm_fits(0, 4)


def test_sparse_rows_beyond_the_kernels_offset_range_are_refused():
    """tlc_spgemm_csr_dense_f32 addresses entries by 32-bit byte offsets (nnz < 2^29, include/tlcgnn.h): the wrappers refuse a
    larger matrix instead of reading its tail back as zeros; gat_tiles refuses a tile wider than the kernel's LDS tile."""
    from types import SimpleNamespace
    from tlc_gnn_amd import ops, _lib
    ops._check_sparse_rows(SimpleNamespace(nnz=(1 << 29) - 1))
    with pytest.raises(_lib.TlcError):
        ops._check_sparse_rows(SimpleNamespace(nnz=1 << 29))
    with pytest.raises(ValueError):
        ops.gat_tiles(None, None, 10, tile_nodes=ops.GAT_TILE_NODES + 1)

"""Negative enumeration on the device, the streamed split and the sparse image store."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")
pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _csr(a):
    import scipy.sparse as sp
    m = sp.csr_matrix(a)
    m.sort_indices()
    return m.indptr.astype(np.int32), m.indices.astype(np.int32)


def test_complement_pairs_vs_oracle(torch_cuda):
    """Every list number, contiguous and through a rank list, on graphs with the corner cases: no edges, complete graph
    (only the diagonal is left), self loops (the diagonal entry is an edge), a node adjacent to everything, one node."""
    torch = torch_cuda
    from tlc_gnn_amd import engine
    from oracle import oracle
    rs = np.random.RandomState(5)
    cases = []
    for n, p in ((1, 0.0), (2, 1.0), (7, 0.3), (64, 0.05), (64, 1.0), (65, 0.0), (300, 0.02), (300, 0.5), (1500, 0.004)):
        a = np.triu((rs.rand(n, n) < p).astype(np.float64), 1)
        a = a + a.T
        cases.append(a)
    loops = cases[3].copy()
    loops[np.arange(0, 64, 3), np.arange(0, 64, 3)] = 1                      # self loops
    hub = cases[6].copy()
    hub[17, :] = 1; hub[:, 17] = 1                                            # row 17 (and the tail of every row) without non-edges
    full_loops = np.ones((9, 9))                                              # nothing left at all
    cases += [loops, hub, full_loops]
    for a in cases:
        n = len(a)
        ref = oracle.complement_pairs_dense(a)
        rowptr, col = _csr(a)
        ci = engine.ComplementIndex(rowptr, col)
        assert len(ci) == len(ref)
        got = ci.pairs().cpu().numpy()
        assert got.shape == (len(ref), 2) and np.array_equal(got, ref)
        ranks = np.concatenate([rs.randint(0, max(len(ref), 1), 500), [-1, len(ref), len(ref) + 5, 0, len(ref) - 1]]).astype(np.int64)
        got = ci.pairs(ranks=torch.from_numpy(ranks).cuda()).cpu().numpy()
        ok = (ranks >= 0) & (ranks < len(ref))
        assert np.array_equal(got[ok], ref[ranks[ok]]) and np.all(got[~ok] == -1)
        if len(ref) > 10:
            assert np.array_equal(ci.pairs(first=3, count=6).cpu().numpy(), ref[3:9])
    with pytest.raises(ValueError):
        engine.ComplementIndex(np.array([0, 2, 2], dtype=np.int32), np.array([1, 0], dtype=np.int32))     # unsorted row


def test_streamed_split_matches_reference_golden_g7(torch_cuda):
    import scipy.sparse as sp
    from tlc_gnn_amd import loaddatas
    d = np.load(os.path.join(G, "adj_split.npz"))
    n, edges = int(d["n_nodes"]), d["edges"]
    a = sp.coo_matrix((np.ones(len(edges)), (edges[:, 0], edges[:, 1])), shape=(n, n))
    tr, neg, va, vaf, te, tef = loaddatas.get_adj_split_streamed(sp.csr_matrix(a + a.T), val_prop=0.05, test_prop=0.1, seed=1234)
    for name, got in (("train_edges", tr), ("val_edges", va), ("val_edges_false", vaf), ("test_edges", te), ("test_edges_false", tef)):
        assert np.array_equal(got, d[name]), name
    ref_false = d["train_edges_false"]
    assert len(neg) + len(va) + len(te) == len(ref_false)
    assert np.array_equal(neg[:], ref_false[: len(neg)])                       # the whole shuffled negative list
    assert np.array_equal(np.concatenate([va, te]), ref_false[len(neg):])
    pick = np.array([0, 5, len(neg) - 1, 77, 77])
    assert np.array_equal(neg[pick], ref_false[pick])


def test_full_pubmed_complement_by_properties(torch_cuda):
    """All 1.9e8 non-edges of the PubMed-shaped graph, in chunks: strictly increasing in (x, y) within and across chunks,
    x <= y, none of them an edge, and as many as N(N+1)/2 - M -- together: exactly the reference's list, in its order."""
    torch = torch_cuda
    from tlc_gnn_amd import engine, synth
    n, edges, _, _, _ = synth.shaped_graph("PubMed")
    rowptr, col, _ = synth.edges_to_csr(n, edges)
    ci = engine.ComplementIndex(rowptr, col)
    m = len(np.unique(np.sort(edges, 1), axis=0))
    assert len(ci) == n * (n + 1) // 2 - m
    ekeys = torch.from_numpy(np.sort(np.minimum(edges[:, 0], edges[:, 1]).astype(np.int64) * n + np.maximum(edges[:, 0], edges[:, 1]))).cuda()
    last = -1
    chunk = 1 << 25
    for lo in range(0, len(ci), chunk):
        p = ci.pairs(first=lo, count=min(chunk, len(ci) - lo)).long()
        assert bool((p[:, 0] <= p[:, 1]).all()) and bool((p[:, 0] >= 0).all()) and bool((p[:, 1] < n).all())
        key = p[:, 0] * n + p[:, 1]
        assert int(key[0]) > last and bool((key[1:] > key[:-1]).all())
        last = int(key[-1])
        pos = torch.searchsorted(ekeys, key).clamp_(max=ekeys.numel() - 1)
        assert not bool((ekeys[pos] == key).any())
        del p, key, pos


def test_streamed_images_equal_the_dense_path(torch_cuda, tmp_path):
    """compute_persistence_image_streamed (sparse store over the lazy pair list) against tlc_pd_pi_batch on the materialised
    list and against the CPU oracle, on a graph with isolated nodes (status 1 rows must be kept although they are zero)."""
    torch = torch_cuda
    import scipy.sparse as sp
    from tlc_gnn_amd import loaddatas, engine, synth
    from tlc_gnn_amd.data import Data
    from tlc_gnn_amd.baselines.TLCGNN import remove_pairs_both_directions
    from tlc_gnn_amd.pi_cache import SparseImages
    from oracle import oracle
    n, edges, kappa, hop, _ = synth.shaped_graph("PubMed", scale=0.03)
    n += 3                                                                     # three isolated nodes
    a = sp.coo_matrix((np.ones(len(edges)), (edges[:, 0], edges[:, 1])), shape=(n, n))
    tr, neg, va, vaf, te, tef = loaddatas.get_adj_split_streamed(sp.csr_matrix(a + a.T), seed=1234)
    ei = torch.from_numpy(np.concatenate([edges, edges[:, ::-1]]).T.copy()).long()
    ei = remove_pairs_both_directions(ei, np.concatenate([va, te]))
    und = np.unique(np.sort(ei.numpy().T, axis=1), axis=0)
    kap = dict(zip(map(tuple, np.sort(edges, 1).tolist()), kappa.tolist()))
    ricci = sorted([[a_, b_, kap[(min(a_, b_), max(a_, b_))]] for a_, b_ in und.tolist()] +
                   [[b_, a_, kap[(min(a_, b_), max(a_, b_))]] for a_, b_ in und.tolist()])
    data = Data(x=None, edge_index=ei, y=torch.zeros(n), ricci_list=ricci)
    images, total = loaddatas.compute_persistence_image_streamed(data, tr, neg, va, vaf, te, tef, hop=hop, chunk=100003, keep_failed=True)
    assert len(total) == len(tr) + len(neg) + len(va) + len(te) + len(va) + len(vaf) + len(te) + len(tef)
    assert images.shape == (len(total), 25)
    # dense path on the materialised list
    allp = total.gather(np.arange(len(total)))
    assert np.array_equal(allp[: len(tr)], tr) and np.array_equal(allp[len(tr): len(tr) + len(neg)], neg[:])
    kund = np.array([kap[(int(a_), int(b_))] for a_, b_ in und.tolist()])
    rowptr, col, w = synth.edges_to_csr(n, und, kund)
    g = engine.DeviceGraph(rowptr, col, w)
    dense, st = g.pd_pi_batch(torch.from_numpy(allp.astype(np.int32)).cuda(), hop)
    dense, st = dense.cpu().numpy(), st.cpu().numpy()
    assert np.array_equal(images.to_dense(), dense)
    full_status = np.zeros(len(total), dtype=np.uint8)
    full_status[images.idx] = images.status
    assert np.array_equal(full_status, st) and (st == 1).any()
    assert images.cnt_compute == int((st == 0).sum())
    assert np.array_equal(images.status_counts, np.bincount(st, minlength=8))
    # default store: non-zero rows only, the failures as a histogram
    lean, _ = loaddatas.compute_persistence_image_streamed(data, tr, neg, va, vaf, te, tef, hop=hop, chunk=50000)
    assert np.array_equal(lean.idx, np.nonzero((dense != 0).any(1))[0]) and np.array_equal(lean.to_dense(), dense)
    assert np.array_equal(lean.status_counts, images.status_counts) and not lean.status.any()
    # through the distance pre-filter: the same rows; the far negatives' status bytes are left uncomputed
    pre, _ = loaddatas.compute_persistence_image_streamed(data, tr, neg, va, vaf, te, tef, hop=hop, prefilter=True)
    assert np.array_equal(pre.idx, lean.idx) and np.array_equal(pre.rows, lean.rows)
    assert pre.unclassified > 0 and pre.status_counts.sum() + pre.unclassified == len(total)
    assert len(images.idx) < len(total)                                         # (a 600-node graph at hop 2 is not sparse; PubMed's sweep keeps 0.3 %)
    # and against the oracle on a sample
    pick = np.concatenate([np.arange(0, len(total), 97), images.idx[:: max(1, len(images.idx) // 400)]])
    ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, allp[pick].astype(np.int32), hop, n_threads=0)
    got = images[pick]
    assert np.array_equal(rst, full_status[pick]) and np.array_equal(got == 0, ref == 0)
    nz = ref != 0
    assert (np.abs(got[nz] - ref[nz]) / np.abs(ref[nz])).max() < 1e-8
    # device gather and the on-disk format
    idx = torch.from_numpy(pick).cuda()
    assert np.array_equal(images.gather_device(idx).cpu().numpy(), got)
    f = str(tmp_path / "PubMed_small.npz")
    images.save(f)
    assert np.array_equal(SparseImages.load(f)[pick], got)
    g.close()


def test_select_rows_overflow_reports_the_needed_capacity(torch_cuda):
    torch = torch_cuda
    from tlc_gnn_amd import engine
    pi = torch.zeros((1000, 25), dtype=torch.float64, device="cuda")
    pi[::10, 3] = 1.0
    st = torch.zeros(1000, dtype=torch.uint8, device="cuda")
    st[5] = 2
    count = torch.zeros(1, dtype=torch.int64, device="cuda")
    idx = torch.full((16,), -7, dtype=torch.int64, device="cuda")
    ost = torch.zeros(16, dtype=torch.uint8, device="cuda")
    rows = torch.zeros((16, 25), dtype=torch.float64, device="cuda")
    hist = torch.zeros(8, dtype=torch.int64, device="cuda")
    engine.select_rows(pi, st, 100, count, idx, ost, rows, hist=hist, keep_failed=True)
    assert int(count.item()) == 101                                             # needed, not written
    assert hist.cpu().numpy().tolist() == [999, 0, 1, 0, 0, 0, 0, 0]
    got = idx.cpu().numpy()
    assert np.all(got >= 100) and len(set(got.tolist())) == 16
    idx = torch.empty(128, dtype=torch.int64, device="cuda"); ost = torch.empty(128, dtype=torch.uint8, device="cuda")
    rows = torch.empty((128, 25), dtype=torch.float64, device="cuda")
    count.zero_()
    engine.select_rows(pi, st, 100, count, idx, ost, rows, keep_failed=True)
    k = int(count.item())
    o = np.argsort(idx[:k].cpu().numpy())
    assert np.array_equal(idx[:k].cpu().numpy()[o], np.sort(np.concatenate([np.arange(0, 1000, 10), [5]])) + 100)
    assert np.array_equal(rows[:k].cpu().numpy()[o], pi.cpu().numpy()[idx[:k].cpu().numpy()[o] - 100])
    assert ost[:k].cpu().numpy()[o][1] == 2
    count.zero_()
    engine.select_rows(pi, st, 100, count, idx, ost, rows)                       # default: non-zero rows only
    assert int(count.item()) == 100


def test_call_streamed_equals_call_dense(torch_cuda):
    """TLCGNN.call with the streamed tables (lazy pair list + sparse images) against the dense harness: same counts, same
    pairs, same images, same probabilities for val / test / a train draw."""
    torch = torch_cuda
    import tempfile
    from tlc_gnn_amd import synth
    from tlc_gnn_amd.baselines import TLCGNN
    from tlc_gnn_amd.data import Data
    n, m, F_ = 260, 700, 40
    edges = synth.holme_kim_edges(n, m, triad_p=0.5, seed=21)

    def make():
        ei = torch.from_numpy(np.concatenate([edges, edges[:, ::-1]]).T.copy()).long()
        d = Data(x=torch.from_numpy(synth.synthetic_features(n, F_, seed=2)), edge_index=ei, y=torch.zeros(n, dtype=torch.long))
        d.ricci_list = synth.synthetic_curvature(edges, seed=21)
        return d
    out = {}
    cwd = os.getcwd()
    for streamed in (False, True):
        with tempfile.TemporaryDirectory() as tmp:
            os.chdir(tmp)
            try:
                torch.manual_seed(3)
                model, data = TLCGNN.call(make(), "Cora", F_, 2, 0, streamed=streamed)
            finally:
                os.chdir(cwd)
        model.eval()
        res = {"counts": (data.train_pos, data.train_neg, data.val_pos, data.val_neg, data.test_pos, data.test_neg),
               "ei": data.edge_index.cpu().numpy()}
        with torch.no_grad():
            emb = model.encode(data)
            for t in ("val", "test", "train"):
                np.random.seed(11)
                p, y = model.decode(data, emb.clone(), t)
                res[t] = (p.cpu().numpy(), y.cpu().numpy())
        out[streamed] = (res, model, data)
    a, b = out[False][0], out[True][0]
    assert a["counts"] == b["counts"] and np.array_equal(a["ei"], b["ei"])
    dense_pairs = np.asarray(out[False][2].total_edges)
    lazy = out[True][2].total_edges
    assert np.array_equal(lazy.gather(np.arange(len(lazy))), dense_pairs)
    assert np.array_equal(out[True][1].PI.to_dense(), np.asarray(out[False][1].PI))
    for t in ("val", "test", "train"):
        assert np.array_equal(a[t][1], b[t][1]), t
        assert np.array_equal(a[t][0], b[t][0]), t


@pytest.mark.parametrize("hop", [1, 2, 3])
def test_near_pairs_prefilter_equals_the_full_sweep(torch_cuda, hop):
    """tlc_near_pairs against a BFS on the host (exactly the non-adjacent pairs u <= v within `hop`, with their list numbers),
    and the pre-filtered sweep against the full one: the same non-zero rows at the same indices."""
    torch = torch_cuda
    import scipy.sparse as sp
    from scipy.sparse.csgraph import shortest_path
    from tlc_gnn_amd import engine, synth, pi_cache
    n, edges, kappa, _, _ = synth.shaped_graph("PubMed", scale=0.04)
    n += 2                                                                        # two isolated nodes
    rowptr, col, w = synth.edges_to_csr(n, edges, kappa)
    ci = engine.ComplementIndex(rowptr, col)
    pairs, ranks = engine.near_pairs(ci, hop)
    pairs, ranks = pairs.cpu().numpy(), ranks.cpu().numpy()
    a = sp.coo_matrix((np.ones(len(edges)), (edges[:, 0], edges[:, 1])), shape=(n, n))
    adj = ((a + a.T) > 0).astype(np.float64).tocsr()
    dist = shortest_path(adj, method="D", unweighted=True)
    uu, vv = np.nonzero(np.triu((dist <= hop) & (adj.toarray() == 0)))
    want = set(zip(uu.tolist(), vv.tolist()))
    assert len(pairs) == len(want) and set(map(tuple, pairs.tolist())) == want
    back = ci.pairs(ranks=torch.from_numpy(ranks).cuda()).cpu().numpy()
    assert np.array_equal(back, pairs)                                            # the list numbers are the pairs' own
    small_cap, _ = engine.near_pairs(ci, hop, cap=10)                             # a store that is too small: counted, then redone
    assert set(map(tuple, small_cap.cpu().numpy().tolist())) == want
    g = engine.DeviceGraph(rowptr, col, w)
    full = pi_cache.sweep_images(g, lambda lo, hi: ci.pairs(first=lo, count=hi - lo), len(ci), hop, chunk=70001)
    near = pi_cache.sweep_near(g, ci, hop)
    assert np.array_equal(near.idx, full.idx) and np.array_equal(near.rows, full.rows)
    assert near.near_pairs == len(want) and near.unclassified == len(ci) - len(want)
    # with the reference's shuffle: positions through the inverse permutation
    perm = np.random.RandomState(3).permutation(len(ci))
    inv = np.empty_like(perm); inv[perm] = np.arange(len(ci))
    shuffled = pi_cache.sweep_near(g, ci, hop, positions=lambda r: inv[r])
    dense_shuffled = full.to_dense()[perm]                                        # row p of the shuffled list = list number perm[p]
    assert np.array_equal(shuffled.to_dense(), dense_shuffled)
    g.close()


def test_near_pairs_on_a_graph_beyond_four_bitmaps_per_workgroup(torch_cuda):
    """150 000 nodes: the pre-filter runs one wavefront per workgroup (three 18 KB bitmaps); against a host BFS on a sample of
    source nodes, and every returned list number maps back to its pair."""
    torch = torch_cuda
    from tlc_gnn_amd import engine
    rs = np.random.RandomState(2)
    n = 150000
    par = (rs.rand(n - 1) * np.arange(1, n)).astype(np.int64)                  # a random recursive tree + a few extra edges
    edges = np.concatenate([np.stack([par, np.arange(1, n)], 1), rs.randint(0, n, size=(20000, 2))])
    edges = edges[edges[:, 0] != edges[:, 1]]
    edges = np.unique(np.sort(edges, 1), axis=0)
    import scipy.sparse as sp
    a = sp.coo_matrix((np.ones(len(edges)), (edges[:, 0], edges[:, 1])), shape=(n, n))
    adj = ((a + a.T) > 0).astype(np.int8).tocsr()
    adj.sort_indices()
    ci = engine.ComplementIndex(adj.indptr.astype(np.int32), adj.indices.astype(np.int32))
    pairs, ranks = engine.near_pairs(ci, 2)
    assert np.array_equal(ci.pairs(ranks=ranks).cpu().numpy(), pairs.cpu().numpy())
    pairs = pairs.cpu().numpy()
    order = np.lexsort((pairs[:, 1], pairs[:, 0]))
    pairs = pairs[order]
    starts = np.searchsorted(pairs[:, 0], np.arange(n + 1))
    for u in rs.randint(0, n, 300).tolist() + [0, n - 1]:
        n1 = adj.indices[adj.indptr[u]:adj.indptr[u + 1]]
        ball = {u} | set(n1.tolist())
        for x in n1.tolist():
            ball |= set(adj.indices[adj.indptr[x]:adj.indptr[x + 1]].tolist())
        want = sorted(v for v in ball if v >= u and v not in set(n1.tolist()))
        assert pairs[starts[u]:starts[u + 1], 1].tolist() == want, u


def test_degenerate_inputs_of_the_sweep_entry_points(torch_cuda):
    """no nodes, one node, no edges, empty requests: sizes come out right and nothing faults"""
    torch = torch_cuda
    from tlc_gnn_amd import engine, pi_cache
    ci0 = engine.ComplementIndex(np.array([0], dtype=np.int32), np.zeros(0, dtype=np.int32))
    assert len(ci0) == 0 and ci0.pairs().shape == (0, 2)
    ci1 = engine.ComplementIndex(np.array([0, 0], dtype=np.int32), np.zeros(0, dtype=np.int32))
    assert len(ci1) == 1 and ci1.pairs().cpu().numpy().tolist() == [[0, 0]]          # the diagonal pair
    p, r = engine.near_pairs(ci1, 2)
    assert p.cpu().numpy().tolist() == [[0, 0]] and r.cpu().numpy().tolist() == [0]
    ci5 = engine.ComplementIndex(np.zeros(6, dtype=np.int32), np.zeros(0, dtype=np.int32))   # five isolated nodes
    assert len(ci5) == 15
    p, r = engine.near_pairs(ci5, 1)
    assert sorted(map(tuple, p.cpu().numpy().tolist())) == [(i, i) for i in range(5)]  # only the diagonal is within any hop
    assert ci5.pairs(first=3, count=0).shape == (0, 2)
    assert ci5.pairs(ranks=torch.zeros(0, dtype=torch.int64, device="cuda")).shape == (0, 2)
    empty = pi_cache.SparseImages(0, 25, np.zeros(0, dtype=np.int64), np.zeros((0, 25)), np.zeros(0, dtype=np.uint8))
    assert empty.to_dense().shape == (0, 25) and empty.gather(np.zeros(0, dtype=np.int64)).shape == (0, 25)

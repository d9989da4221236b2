"""GPU: TLCGNN link-prediction forward (HIP, through the C ABI / the drop-in classes) against the pure-torch fp32
restatement of oracle/lp_forward_ref.py.  Tolerance: 1e-5 relative (north_star), with an absolute floor of 1e-6."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-6


@pytest.fixture(scope="module")
def setup():
    import torch
    assert torch.cuda.is_available()
    from tlc_gnn_amd import synth
    n, m, F_ = 1500, 5200, 233
    edges = synth.holme_kim_edges(n, m, triad_p=0.4, seed=3)
    x = torch.from_numpy(synth.synthetic_features(n, F_, seed=3))
    ei = torch.from_numpy(np.concatenate([edges, edges[:, ::-1]]).T.copy()).long()
    # a few self loops and a duplicate edge: gcn_norm must treat them like add_remaining_self_loops does
    ei = torch.cat([ei, torch.tensor([[5, 9, 9], [5, 9, 9]]), ei[:, :3]], dim=1)
    return torch, n, F_, x, ei


def _close(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    err = (a - b).abs()
    return bool((err <= ATOL + RTOL * b.abs()).all()), float((err / (b.abs() + ATOL)).max())


def test_gcn_norm_csr(setup):
    torch, n, F_, x, ei = setup
    from tlc_gnn_amd import ops
    from oracle import lp_forward_ref as ref
    rowptr, col, val = ops.gcn_norm_csr(ei.cuda(), n)
    rei, norm = ref.gcn_norm(ei, n)
    dense_ref = torch.zeros(n, n, dtype=torch.float64)
    dense_ref.index_put_((rei[1], rei[0]), norm.double(), accumulate=True)
    rp, c, v = rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().double().numpy()
    dense = np.zeros((n, n))
    for i in range(n):
        assert np.all(np.diff(c[rp[i]:rp[i + 1]]) >= 0)         # sources ascending inside a row
        np.add.at(dense[i], c[rp[i]:rp[i + 1]], v[rp[i]:rp[i + 1]])
    assert np.abs(dense - dense_ref.numpy()).max() < 1e-6
    assert len(c) == rei.shape[1]


# vectorised (K, N multiples of 4) and element-wise load paths, K a multiple of the 32-chunk or not, odd chunk counts,
# row counts around the 80-row tile, every column-tile count of the 16x16x4 kernel
def test_gcn_norm_csr_large_graph_multi_block_scan(setup):
    """above 8 192 nodes the row-pointer prefix is a three-kernel scan: a 50 000-node graph (isolated nodes, self loops,
    duplicates) against numpy."""
    torch = setup[0]
    from tlc_gnn_amd import ops
    rs = np.random.RandomState(8)
    n, E = 50000, 160000
    src, dst = rs.randint(0, n - 500, E), rs.randint(0, n - 500, E)          # the last 500 nodes stay isolated
    src[:100] = dst[:100]                                                     # some self loops
    ei = torch.from_numpy(np.stack([np.concatenate([src, src[:50]]), np.concatenate([dst, dst[:50]])])).long()
    rowptr, col, val = ops.gcn_norm_csr(ei.cuda(), n)
    rp, c, v = rowptr.cpu().numpy(), col.cpu().numpy(), val.cpu().numpy()
    # reference: every non-loop entry (duplicates kept), one self loop per node; CSR by target, sources ascending
    keep = src != dst
    s2 = np.concatenate([src[keep], src[:50][keep[:50]], np.arange(n)])
    d2 = np.concatenate([dst[keep], dst[:50][keep[:50]], np.arange(n)])
    order = np.lexsort((s2, d2))
    s2, d2 = s2[order], d2[order]
    deg = np.bincount(d2, minlength=n).astype(np.float64)
    assert np.array_equal(rp, np.concatenate([[0], np.cumsum(np.bincount(d2, minlength=n))]))
    assert np.array_equal(c, s2)
    ref = (1.0 / np.sqrt(deg[s2])) * (1.0 / np.sqrt(deg[d2]))
    assert np.abs(v - ref).max() < 1e-6


@pytest.mark.parametrize("shape", [(1500, 233, 100), (1500, 100, 16), (77, 5, 3), (130, 64, 128), (1, 1, 1), (19717, 500, 100),
                                   (2708, 1433, 7), (80, 32, 16), (81, 33, 17), (79, 96, 48), (161, 31, 65), (400, 768, 80),
                                   (333, 160, 112), (5, 4, 4)])
def test_gemm_f32_mfma(setup, shape):
    torch = setup[0]
    from tlc_gnn_amd import ops
    M, K, N = shape
    g = torch.Generator().manual_seed(M * 7 + N)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    out = ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), relu=True)
    ref = torch.relu(a.double() @ b.double() + bias.double())
    # per ELEMENT: an fp32 accumulation chain is off by at most ~K * 2^-24 * sum_k |a_ik b_kj| whatever the order, and the
    # observed error is far below that; 5e-7 of the element's own magnitude sum is a bound no element may exceed
    err = (out.cpu().double() - ref).abs()
    mag = a.abs().double() @ b.abs().double() + bias.abs().double()
    assert bool((err <= 5e-7 * mag + 1e-30).all()), float((err / mag).max())
    # without cancellation (non-negative operands, like TF-IDF features against a non-negative weight block) that is the
    # north_star bound on the OUTPUT itself: 1e-5 relative, element by element
    ap, bp = a.abs(), b.abs()
    outp = ops.gemm(ap.cuda(), bp.cuda())
    refp = ap.double() @ bp.double()
    relp = ((outp.cpu().double() - refp).abs() / refp.clamp_min(1e-300))
    assert float(relp.max()) < 1e-5, float(relp.max())
    # asymmetric check of the C layout: A = I picks rows of B
    eye = torch.eye(K)
    out = ops.gemm(eye.cuda(), b.cuda())
    assert torch.equal(out.cpu(), b)


@pytest.mark.parametrize("shape", [(4096, 32, 68), (5000, 16, 36), (100003, 64, 32), (7777, 64, 16), (4097, 32, 64), (9001, 16, 5),
                                   (8192, 64, 80), (4100, 32, 1)])
def test_gemm_skinny_k(setup, shape):
    """the streaming kernel for K = 16 / 32 / 64 and M >= 4096 (the PDGNN shapes): ragged M and N, bias and ReLU, no bias, and
    an asymmetric layout check (A with one 1 per row picks rows of B)."""
    torch = setup[0]
    from tlc_gnn_amd import ops
    M, K, N = shape
    g = torch.Generator().manual_seed(M + 13 * N)
    a = torch.randn(M, K, generator=g)
    b = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    out = ops.gemm(a.cuda(), b.cuda(), bias=bias.cuda(), relu=True)
    ref = torch.relu(a.double() @ b.double() + bias.double())
    scale = (a.abs().double() @ b.abs().double()).max().item()
    assert (out.cpu().double() - ref).abs().max().item() <= 1e-6 * scale + 1e-6
    out = ops.gemm(a.cuda(), b.cuda())
    assert (out.cpu().double() - a.double() @ b.double()).abs().max().item() <= 1e-6 * scale + 1e-6
    pick = torch.randint(0, K, (M,), generator=g)
    sel = torch.zeros(M, K)
    sel[torch.arange(M), pick] = 1.0
    assert torch.equal(ops.gemm(sel.cuda(), b.cuda()).cpu(), b[pick])


def test_encode_decode_vs_torch_reference(setup):
    torch, n, F_, x, ei = setup
    from tlc_gnn_amd.baselines import TLCGNN
    from tlc_gnn_amd.data import Data
    from oracle import lp_forward_ref as ref
    torch.manual_seed(1234)
    E = 4000
    rs = np.random.RandomState(5)
    pairs = rs.randint(0, n, size=(E, 2))
    PI = rs.uniform(0, 0.3, size=(E, 25))
    PI[::7] = 0.0
    data = Data(x=x.clone(), edge_index=ei.clone(), y=torch.zeros(n), total_edges=pairs,
                total_edges_y=torch.from_numpy((rs.rand(E) < 0.5).astype(np.int64)),
                train_pos=1000, train_neg=1500, val_pos=300, val_neg=300, test_pos=450, test_neg=450)
    model = TLCGNN.Net(data, F_, 2, PI=PI)
    with torch.no_grad():
        model.conv1.bias.uniform_(-0.1, 0.1)      # reference init is zeros; exercise the bias path
        model.conv2.bias.uniform_(-0.1, 0.1)
        model.conv2.weight.mul_(3.0)              # push some embedding rows over norm 1 so that renorm acts
    model = model.cuda().eval()
    data = data.to("cuda")
    with torch.no_grad():
        emb = model.encode(data)
    w1, b1 = model.conv1.weight.detach().cpu(), model.conv1.bias.detach().cpu()
    w2, b2 = model.conv2.weight.detach().cpu(), model.conv2.bias.detach().cpu()
    emb_ref = ref.tlcgnn_encode(x, ei, w1, b1, w2, b2)
    ok, worst = _close(emb, emb_ref)
    assert ok, worst
    assert (emb_ref.norm(dim=1) > 1).any() and (emb_ref.norm(dim=1) < 1).any()
    for typ, sl in (("val", slice(2500, 3100)), ("test", slice(3100, None))):
        emb_in = emb.clone()
        prob, y = model.decode(data, emb_in, typ)
        emb_r = emb_ref.clone()
        prob_ref = ref.tlcgnn_decode(emb_r, torch.from_numpy(pairs[sl]), torch.from_numpy(PI[sl]),
                                     model.linear_1.weight.detach().cpu(), model.linear_1.bias.detach().cpu(),
                                     model.linear.weight.detach().cpu(), model.linear.bias.detach().cpu())
        ok, worst = _close(prob, prob_ref)
        assert ok, (typ, worst)
        ok, worst = _close(emb_in, emb_r)              # renorm_ acted in place, like the reference
        assert ok, worst
        assert torch.equal(y.cpu(), data.total_edges_y[sl].float().cpu())
    # train: same RNG stream as the reference's np.random.randint
    np.random.seed(77)
    prob, y = model.decode(data, emb.clone(), "train")
    np.random.seed(77)
    index = np.random.randint(0, 1500, 1000)
    sel = np.concatenate([np.arange(1000), 1000 + index])
    prob_ref = ref.tlcgnn_decode(emb_ref.clone(), torch.from_numpy(pairs[sel]), torch.from_numpy(PI[sel]),
                                 model.linear_1.weight.detach().cpu(), model.linear_1.bias.detach().cpu(),
                                 model.linear.weight.detach().cpu(), model.linear.bias.detach().cpu())
    ok, worst = _close(prob, prob_ref)
    assert ok, worst
    assert prob.shape[0] == 2000 and y.shape[0] == 2000


def test_decode_matches_reference_golden_g10():
    """The drop-in Net.decode (HIP: renorm + lp_decode8) against Net.decode of the IMPORTED reference
    (/root/reference/baselines/TLCGNN.py:27-62; tests/golden/decode.npz, make_golden.py --only g10): probabilities within 1e-5
    relative for 'train' / 'val' / 'test', labels and the np.random.randint negatives equal, the in-place renorm of the
    embedding equal.  Also: rebinding model.PI / data.total_edges is honoured by the next decode (the reference re-slices
    every call, :35-36)."""
    import os
    import torch
    from tlc_gnn_amd.baselines import TLCGNN
    from tlc_gnn_amd.data import Data
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "decode.npz"))
    tp, tn, vp, vn, sp_, sn = d["counts"].tolist()
    n = int(d["n_nodes"])
    data = Data(x=torch.zeros(n, 4), edge_index=torch.zeros(2, 0, dtype=torch.long), y=torch.zeros(n),
                total_edges=d["pairs"].copy(), total_edges_y=torch.from_numpy(d["y"]),
                train_pos=tp, train_neg=tn, val_pos=vp, val_neg=vn, test_pos=sp_, test_neg=sn)
    model = TLCGNN.Net(data, 4, 2, PI=d["PI"].copy())
    with torch.no_grad():
        model.linear_1.weight.copy_(torch.from_numpy(d["lin1_w"])); model.linear_1.bias.copy_(torch.from_numpy(d["lin1_b"]))
        model.linear.weight.copy_(torch.from_numpy(d["lin_w"])); model.linear.bias.copy_(torch.from_numpy(d["lin_b"]))
    model = model.cuda().eval()
    data = data.to("cuda")
    for kind in ("train", "val", "test"):
        np.random.seed(int(d["np_seed"]))
        emb = torch.from_numpy(d["emb"].copy()).cuda()
        with torch.no_grad():
            prob, y = model.decode(data, emb, kind)
        want = d["prob_" + kind]
        got = prob.cpu().numpy()
        assert got.shape == want.shape
        # 1e-5 relative (north_star) ON THE DECODER'S OUTPUT d = clamp(|W2 h + b2|, 0, 40); the link function
        # prob = 1 / (exp(d - 2) + 1) turns a relative error eps of d into a relative error eps * d of prob on its exponential
        # tail (d up to 40 here: the huge image rows), so the bound on prob is 1e-5 * max(1, d) relative
        w64 = want.astype(np.float64)
        d_want = np.maximum(2.0 + np.log(np.maximum(1.0 / w64 - 1.0, 1e-300)), 0.0)
        tol = 1e-5 * np.maximum(1.0, d_want) * w64 + 1e-25
        bad = np.abs(got.astype(np.float64) - w64) > tol
        assert not bad.any(), (kind, int(bad.sum()), got[bad][:4], want[bad][:4])
        assert d_want.max() > 39.0 and d_want.min() < 0.5                      # both ends of the clamp are in the fixture
        assert np.array_equal(y.cpu().numpy(), d["y_" + kind])
        after = d["emb_after_" + kind]
        assert np.all(np.abs(emb.cpu().numpy() - after) <= 1e-6 * np.abs(after) + 1e-12)
        assert np.array_equal(emb.cpu().numpy()[np.linalg.norm(d["emb"], axis=1) <= 1.0 - 1e-4],
                              d["emb"][np.linalg.norm(d["emb"], axis=1) <= 1.0 - 1e-4])          # short rows untouched, bit for bit
    # a rebound table is used by the next call
    model.PI = np.zeros_like(d["PI"])
    emb = torch.from_numpy(d["emb"].copy()).cuda()
    with torch.no_grad():
        p0, _ = model.decode(data, emb, "val")
    assert np.abs(p0.cpu().numpy() - d["prob_val"]).max() > 1e-3
    model.PI = d["PI"].copy()
    data.total_edges = d["pairs"][::-1].copy()
    emb = torch.from_numpy(d["emb"].copy()).cuda()
    with torch.no_grad():
        p1, _ = model.decode(data, emb, "val")
    assert np.abs(p1.cpu().numpy() - d["prob_val"]).max() > 1e-3


def test_full_size_pubmed_encode_decode_vs_restatement():
    """BASELINE configs[1] at full size: the PubMed-shaped graph (N = 19 717, F = 500, hub rows in the SpMM) through
    Net.encode and Net.decode over the 75 352-pair training batch (TLCGNN.py:29-32), against the torch restatement:
    embedding and probabilities within 1e-5 relative, element by element."""
    import torch
    from tlc_gnn_amd import synth
    from tlc_gnn_amd.baselines import TLCGNN
    from tlc_gnn_amd.data import Data
    from oracle import lp_forward_ref as ref
    n, edges, kappa, hop, F_ = synth.shaped_graph("PubMed")
    x = torch.from_numpy(synth.synthetic_features(n, F_))
    ei = torch.from_numpy(np.concatenate([edges, edges[:, ::-1]]).T.copy()).long()
    rs = np.random.RandomState(11)
    n_pos = 37676
    pairs = np.concatenate([edges[rs.permutation(len(edges))[:n_pos]], rs.randint(0, n, size=(n_pos + 500, 2))])
    PI = rs.uniform(0, 0.3, size=(len(pairs), 25))
    PI[::5] = 0.0
    E = len(pairs)
    data = Data(x=x.clone(), edge_index=ei.clone(), y=torch.zeros(n), total_edges=pairs,
                total_edges_y=torch.from_numpy((np.arange(E) < n_pos).astype(np.int64)),
                train_pos=n_pos, train_neg=n_pos + 500, val_pos=0, val_neg=0, test_pos=0, test_neg=0)
    torch.manual_seed(1234)
    model = TLCGNN.Net(data, F_, 2, PI=PI)
    with torch.no_grad():
        for mod in (model.linear, model.linear_1):                    # weights_init of pipelines.py:42-46
            torch.nn.init.xavier_uniform_(mod.weight)
            torch.nn.init.zeros_(mod.bias)
        model.conv2.weight.mul_(4.0)              # some rows over norm 1: renorm_ acts on part of the embedding
    model = model.cuda().eval()
    data = data.to("cuda")
    with torch.no_grad():
        emb = model.encode(data)
    w1, b1 = model.conv1.weight.detach().cpu(), model.conv1.bias.detach().cpu()
    w2, b2 = model.conv2.weight.detach().cpu(), model.conv2.bias.detach().cpu()
    emb_ref = ref.tlcgnn_encode(x, ei, w1, b1, w2, b2)
    assert tuple(emb.shape) == (n, 16)
    ok, worst = _close(emb, emb_ref)
    assert ok, worst
    norms = emb_ref.norm(dim=1)
    assert (norms > 1).any() and (norms < 1).any()
    np.random.seed(5)
    with torch.no_grad():
        prob, y = model.decode(data, emb.clone(), "train")
    np.random.seed(5)
    index = np.random.randint(0, n_pos + 500, n_pos)                                       # TLCGNN.py:30
    sel = np.concatenate([np.arange(n_pos), n_pos + index])
    prob_ref = ref.tlcgnn_decode(emb_ref.clone(), torch.from_numpy(pairs[sel]), torch.from_numpy(PI[sel]),
                                 model.linear_1.weight.detach().cpu(), model.linear_1.bias.detach().cpu(),
                                 model.linear.weight.detach().cpu(), model.linear.bias.detach().cpu())
    assert prob.shape[0] == 2 * n_pos == 75352
    ok, worst = _close(prob, prob_ref)
    assert ok, worst


@pytest.mark.parametrize("shape,density", [((19717, 500, 100), 0.1), ((2708, 640, 100), 0.013), ((777, 33, 7), 0.3), ((64, 500, 130), 0.1),
                                           ((5, 4, 1), 0.5), ((300, 256, 64), 0.0), ((40, 600, 20), 0.5), ((300000, 8, 4), 0.5),
                                           ((1000, 636, 64), 0.05), ((64, 5000, 4), 0.02), ((64, 4500, 8), 0.02),
                                           ((33, 10176, 4), 0.01), ((33, 4096, 4), 0.01)])
def test_sparse_feature_projection(setup, shape, density):
    """x @ W over the stored entries of x (tlc_spgemm_csr_dense_f32): against the float64 product, element by element, and
    against the dense MFMA kernel; empty rows, N not a multiple of 64, more than one column slice, an all-zero matrix, rows of
    several 64-entry chunks (300 entries), more than 64 rows per wavefront (300 000 rows), the largest K of a 64-column slice, and
    narrow slices with K >= 4 096 (N = 4 / 8: the staging's item -> (row, quad) split needs a 64-bit product there; rows of the
    weight slice from 4 096 on were staged as zeros before)."""
    torch = setup[0]
    from tlc_gnn_amd import ops
    M, K, N = shape
    g = torch.Generator().manual_seed(M + K)
    x = torch.rand(M, K, generator=g) * (torch.rand(M, K, generator=g) < density).float()
    if M > 10:
        x[3] = 0.0                                                     # an empty row
    w = torch.randn(K, N, generator=g)
    bias = torch.randn(N, generator=g)
    xs = ops.SparseRows(x.cuda())
    assert xs.nnz == int((x != 0).sum()) and abs(xs.density - xs.nnz / float(M * K)) < 1e-12
    out = ops.sparse_gemm(xs, w.cuda(), bias=bias.cuda(), relu=True).cpu()
    ref = torch.relu(x.double() @ w.double() + bias.double())
    mag = x.abs().double() @ w.abs().double() + bias.abs().double()
    err = (out.double() - ref).abs()
    assert bool((err <= 5e-7 * mag + 1e-30).all()), float((err / mag).max())
    if N <= 128:                                                       # (the dense kernel's limit: GCN hidden sizes)
        dense = ops.gemm(x.cuda(), w.cuda(), bias=bias.cuda(), relu=True).cpu()
        assert torch.allclose(out, dense, rtol=1e-5, atol=1e-5)
    plain = ops.sparse_gemm(xs, w.cuda()).cpu()
    assert bool(((plain.double() - x.double() @ w.double()).abs() <= 5e-7 * (mag + 1e-12) + 1e-30).all())


def test_encode_uses_the_sparse_projection_and_tracks_feature_edits(setup):
    """Net.encode keeps the CSR of data.x while x is the same unmodified tensor; an in-place edit or a new tensor rebuilds it."""
    torch, n, F_, x, ei = setup
    from tlc_gnn_amd.baselines import TLCGNN
    from tlc_gnn_amd.data import Data
    from oracle import lp_forward_ref as ref
    torch.manual_seed(5)
    model = TLCGNN.Net(None, F_, 2, PI=None).cuda().eval()
    x = x * (torch.rand(x.shape, generator=torch.Generator().manual_seed(9)) < 0.2).float()      # ~2 % dense, like Cora's bag of words
    data = Data(x=x.clone().cuda(), edge_index=ei.cuda(), y=torch.zeros(n))
    w1, b1 = model.conv1.weight.detach().cpu(), model.conv1.bias.detach().cpu()
    w2, b2 = model.conv2.weight.detach().cpu(), model.conv2.bias.detach().cpu()
    with torch.no_grad():
        e1 = model.encode(data)
        assert model._xs[2] is not None and model._xs[2].density < model.SPARSE_FEATURES_BELOW
        ok, worst = _close(e1, ref.tlcgnn_encode(x, ei, w1, b1, w2, b2))
        assert ok, worst
        data.x[:50] += 0.5                                                        # in place: same tensor, new version
        e2 = model.encode(data)
        x2 = x.clone(); x2[:50] += 0.5
        ok, worst = _close(e2, ref.tlcgnn_encode(x2, ei, w1, b1, w2, b2))
        assert ok, worst
        data.x = torch.ones(n, F_).cuda()                                         # dense features: the MFMA path
        e3 = model.encode(data)
        assert model._xs[2] is None
        ok, worst = _close(e3, ref.tlcgnn_encode(torch.ones(n, F_), ei, w1, b1, w2, b2))
        assert ok, worst


def test_decode_generic_dims(setup):
    torch = setup[0]
    from tlc_gnn_amd import ops
    from oracle import lp_forward_ref as ref
    g = torch.Generator().manual_seed(0)
    n, D, P, E = 50, 8, 9, 333
    emb = torch.randn(n, D, generator=g) * 0.3
    pairs = torch.randint(0, n, (E, 2), generator=g)
    pi = torch.rand(E, P, generator=g, dtype=torch.float64)
    w1, b1 = torch.randn(P, D + P, generator=g) * 0.3, torch.randn(P, generator=g) * 0.1
    w2, b2 = torch.randn(1, P, generator=g) * 0.3, torch.randn(1, generator=g) * 0.1
    emb_d = ops.renorm_rows_(emb.cuda())
    out = ops.lp_decode(pairs.int().cuda(), emb_d, pi.cuda(), w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
    r = ref.tlcgnn_decode(emb.clone(), pairs, pi, w1, b1, w2, b2)
    ok, worst = _close(out, r)
    assert ok, worst


def test_decode_float32_table_equals_float64_table(setup):
    """tlc_lp_decode_fused_f32 on an image table cast once (Net._tables; the reference's `torch.Tensor(PI)` of TLCGNN.py:52-53
    hoisted out of the decode) gives the bits of tlc_lp_decode_fused on the float64 table, for pair counts around the sixteen-pair
    tiles of the MFMA kernel and for the generic dimensions; both within the bar of the restatement."""
    torch = setup[0]
    from tlc_gnn_amd import ops
    from oracle import lp_forward_ref as ref
    g = torch.Generator().manual_seed(3)
    for n, D, P, E in ((300, 16, 25, 1), (300, 16, 25, 15), (300, 16, 25, 16), (300, 16, 25, 17), (300, 16, 25, 70001), (40, 8, 9, 100)):
        emb = torch.randn(n, D, generator=g) * 0.3
        pairs = torch.randint(0, n, (E, 2), generator=g)
        pi = torch.rand(E, P, generator=g, dtype=torch.float64) * 3.0
        w1, b1 = torch.randn(P, D + P, generator=g) * 0.3, torch.randn(P, generator=g) * 0.1
        w2, b2 = torch.randn(1, P, generator=g) * 0.3, torch.randn(1, generator=g) * 0.1
        emb_d = ops.renorm_rows_(emb.cuda())
        args = (w1.cuda(), b1.cuda(), w2.cuda(), b2.cuda())
        o64 = ops.lp_decode(pairs.int().cuda(), emb_d, pi.cuda(), *args)
        o32 = ops.lp_decode(pairs.int().cuda(), emb_d, pi.cuda().float(), *args)
        assert torch.equal(o64, o32), (E, float((o64 - o32).abs().max()))
        r = ref.tlcgnn_decode(emb.clone(), pairs, pi, w1, b1, w2, b2)
        ok, worst = _close(o32, r)
        assert ok, (E, worst)


def test_spmm_hub_rows_and_clustered_hubs(setup):
    """Rows longer than the per-group limit (32) go through the cooperative hub path; one row is longer than its 2048-entry
    staging chunk; hubs have neighbouring ids (they must not serialise in one workgroup); k covers every lane-group width."""
    torch = setup[0]
    from tlc_gnn_amd import ops
    rs = np.random.RandomState(5)
    n = 6000
    deg = rs.randint(1, 9, size=n)
    deg[:40] = rs.randint(33, 400, size=40)          # clustered hubs
    deg[7] = 5000                                    # longer than one staging chunk
    deg[n - 1] = 70
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = np.concatenate([np.sort(rs.choice(n, size=d, replace=False)) for d in deg]).astype(np.int32)
    val = rs.randn(len(col)).astype(np.float32)
    for k in (4, 16, 24, 64, 100, 128, 200, 7, 33):
        x = torch.from_numpy(rs.randn(n, k).astype(np.float32))
        bias = torch.from_numpy(rs.randn(k).astype(np.float32))
        y = ops.spmm(torch.from_numpy(rowptr).cuda(), torch.from_numpy(col).cuda(), torch.from_numpy(val).cuda(), x.cuda(),
                     bias=bias.cuda(), relu=True)
        A = torch.sparse_csr_tensor(torch.from_numpy(rowptr).long(), torch.from_numpy(col).long(), torch.from_numpy(val).double(), (n, n))
        ref = torch.relu(A @ x.double() + bias.double())
        mag = torch.sparse_csr_tensor(torch.from_numpy(rowptr).long(), torch.from_numpy(col).long(), torch.from_numpy(np.abs(val)).double(), (n, n)) @ x.abs().double()
        err = (y.cpu().double() - ref).abs()
        assert bool((err <= ATOL + RTOL * (mag + bias.abs().double())).all()), (k, float(err.max()))


def test_spmm_fused_renorm_equals_separate_pass(setup):
    """relu | renorm in one SpMM pass == SpMM then tlc_renorm_rows_f32 (emb.renorm_(2, 0, 1), TLCGNN.py:48), for the lane-group
    widths of the vector kernels, the scalar fallback (k % 4 != 0) and rows that go through the hub path."""
    torch = setup[0]
    from tlc_gnn_amd import ops
    rs = np.random.RandomState(9)
    n = 3000
    deg = rs.randint(1, 9, size=n)
    deg[:5] = [40, 200, 33, 3000, 64]
    rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
    col = np.concatenate([np.sort(rs.choice(n, size=d, replace=False)) for d in deg]).astype(np.int32)
    val = (rs.rand(len(col)).astype(np.float32) - 0.3)
    rp, c, v = [torch.from_numpy(a).cuda() for a in (rowptr, col, val)]
    for k in (16, 100, 8, 128, 7):
        x = torch.from_numpy((0.6 / np.sqrt(k) * rs.randn(n, k)).astype(np.float32)).cuda()
        bias = torch.from_numpy((0.1 / np.sqrt(k) * rs.randn(k)).astype(np.float32)).cuda()
        fused = ops.spmm(rp, c, v, x, bias=bias, relu=True, renorm=True)
        two = ops.spmm(rp, c, v, x, bias=bias, relu=True)
        assert float(two.norm(dim=1).max()) > 1.0 and float(two.norm(dim=1).min()) < 1.0    # both branches of renorm_ occur
        ops.renorm_rows_(two)
        ok, worst = _close(fused, two)
        assert ok, (k, worst)
        assert float(fused.norm(dim=1).max()) <= 1.0 + 1e-5


def test_gcn2_encode_one_call_equals_the_four_calls(setup):
    """tlc_gcn2_encode_f32 (Net.encode in eval mode behind one library call, TLCGNN.py:19-26,48) against gemm / spmm / gemm / spmm,
    with and without the fused renorm_, for sizes whose scratch blocks need padding: identical bits where it submits the same four
    kernels; for out_dim == 16 (round 5: conv2's projection runs in the epilogue of conv1's aggregate, fp32 sums in another
    order) within 1e-5 relative + 1e-6."""
    torch = setup[0]
    from tlc_gnn_amd import ops
    rs = np.random.RandomState(21)
    for n, f_in, hidden, d in ((3001, 500, 100, 16), (517, 33, 12, 7), (64, 8, 128, 16)):
        deg = rs.randint(1, 7, size=n)
        deg[:3] = [min(n - 1, 300), 40, 1]
        rowptr = np.concatenate([[0], np.cumsum(deg)]).astype(np.int32)
        col = np.concatenate([np.sort(rs.choice(n, size=k, replace=False)) for k in deg]).astype(np.int32)
        val = (rs.rand(len(col)).astype(np.float32) * 0.5)
        rp, c, v = [torch.from_numpy(a).cuda() for a in (rowptr, col, val)]
        x = torch.from_numpy((rs.rand(n, f_in) < 0.1).astype(np.float32) * rs.rand(n, f_in).astype(np.float32)).cuda()
        w1 = torch.from_numpy((rs.randn(f_in, hidden) / np.sqrt(f_in)).astype(np.float32)).cuda()
        b1 = torch.from_numpy((0.1 * rs.randn(hidden)).astype(np.float32)).cuda()
        w2 = torch.from_numpy((rs.randn(hidden, d) / np.sqrt(hidden) * 3).astype(np.float32)).cuda()
        b2 = torch.from_numpy((0.1 * rs.randn(d)).astype(np.float32)).cuda()
        for renorm in (False, True):
            h = ops.spmm(rp, c, v, ops.gemm(x, w1), bias=b1, relu=True)
            want = ops.spmm(rp, c, v, ops.gemm(h, w2), bias=b2, relu=True, renorm=renorm)
            got = ops.gcn2_encode(rp, c, v, x, w1, b1, w2, b2, relu=True, renorm=renorm)
            # the features as CSR (tlc_gcn2_encode_csr_f32): the first projection over the stored entries, sums in another order
            got_csr = ops.gcn2_encode(rp, c, v, None, w1, b1, w2, b2, relu=True, renorm=renorm, x_sparse=ops.SparseRows(x))
            ok_csr, worst_csr = _close(got_csr, want)
            assert ok_csr, (n, renorm, worst_csr)
            torch.cuda.synchronize()
            if d == 16 and hidden % 4 == 0:
                assert bool(((got - want).abs() <= 1e-5 * want.abs() + 1e-6).all()), (n, f_in, hidden, d, renorm, float((got - want).abs().max()))
            else:
                assert torch.equal(got, want), (n, f_in, hidden, d, renorm, float((got - want).abs().max()))


def test_hip_graph_replay_equals_eager_forward(setup):
    """ops.capture: the encode + decode chain as one HIP graph; replays give the eager launch's bits, also after the inputs
    held in the captured buffers change."""
    torch, n, F_, x, ei = setup
    from tlc_gnn_amd import ops
    dev = torch.device("cuda")
    torch.manual_seed(5)
    xd = x.to(dev)
    w1, b1 = torch.randn(F_, 100, device=dev) * 0.1, torch.randn(100, device=dev) * 0.1
    w2, b2 = torch.randn(100, 16, device=dev) * 0.1, torch.randn(16, device=dev) * 0.1
    l1w, l1b = torch.randn(25, 41, device=dev) * 0.3, torch.randn(25, device=dev) * 0.1
    l2w, l2b = torch.randn(1, 25, device=dev) * 0.3, torch.randn(1, device=dev) * 0.1
    rowptr, col, val = ops.gcn_norm_csr(ei.to(dev), n)
    E = 4000
    pairs = torch.randint(0, n, (E, 2), device=dev, dtype=torch.int32)
    pi = torch.rand((E, 25), device=dev, dtype=torch.float64)
    prob = torch.empty(E, dtype=torch.float32, device=dev)

    def fwd():
        h = ops.spmm(rowptr, col, val, ops.gemm(xd, w1), bias=b1, relu=True)
        emb = ops.spmm(rowptr, col, val, ops.gemm(h, w2), bias=b2, relu=True, renorm=True)
        ops.lp_decode(pairs, emb, pi, l1w, l1b, l2w, l2b, out=prob)

    fwd()
    torch.cuda.synchronize()
    eager = prob.clone()
    g = ops.capture(fwd)
    prob.zero_()
    g.replay()
    torch.cuda.synchronize()
    assert torch.equal(prob, eager)
    xd.mul_(0.5)                                                 # same buffers, new contents
    pi.mul_(2.0)
    g.replay()
    torch.cuda.synchronize()
    replayed = prob.clone()
    fwd()
    torch.cuda.synchronize()
    assert torch.equal(prob, replayed) and not torch.equal(replayed, eager)

"""GPU: the HIP PD/PI path (through the C ABI) against the CPU oracle and the reference goldens."""
import os

import numpy as np
import pytest

from helpers import same_multiset, ragged_slice, rel_err, csr_from_golden

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return torch


def _dev(torch, a, dtype):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=dtype).cuda()


def test_pi_raster_goldens(torch_cuda):
    torch = torch_cuda
    from tlc_gnn_amd import engine
    d = np.load(os.path.join(G, "pi_kat.npz"))
    out = engine.pi_raster(_dev(torch, [0, len(d["pd"])], torch.int64), _dev(torch, d["pd"], torch.float64), 5).cpu().numpy()[0]
    assert np.abs(out - d["gt_4dp"]).max() < 6e-5          # the reference's own printed known-answer vector
    assert rel_err(out, d["ref_fp64"]).max() < 1e-9         # north_star tolerance: 1e-5 relative
    d = np.load(os.path.join(G, "pi_random.npz"))
    out = engine.pi_raster(_dev(torch, d["offs"], torch.int64), _dev(torch, d["pts"], torch.float64), 5).cpu().numpy()
    ref = d["out"]
    assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max())
    nz = np.abs(ref) > 1e-9
    assert rel_err(out[nz], ref[nz]).max() < 1e-8
    for res in (3, 7):
        o = engine.pi_raster(_dev(torch, d["offs"][:21], torch.int64), _dev(torch, d["pts"][: d["offs"][20]], torch.float64), res).cpu().numpy()
        r = d["out_res%d" % res]
        assert np.abs(o - r).max() <= 1e-11 * max(1.0, np.abs(r).max())


def test_pi_raster_ragged_batch_vs_oracle(torch_cuda):
    """Lane-per-point raster: empty diagrams, lengths around the 16-lane round, the whole-wavefront (96) and the whole-workgroup threshold (1024),
    points outside [0,1]^2 (erfc branch), below the diagonal (weight 0) and with persistence > 1 (weight 1), every res."""
    torch = torch_cuda
    from tlc_gnn_amd import engine
    from oracle import oracle
    rs = np.random.RandomState(11)
    lens = [0, 1, 15, 16, 17, 0, 63, 64, 65, 95, 96, 97, 511, 512, 513, 1023, 1024, 1025, 2000, 3, 0, 700, 31, 5000] + rs.randint(0, 120, size=203).tolist()
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    k = int(offs[-1])
    b = rs.uniform(-0.5, 1.5, size=k)
    pts = np.stack([b, b + rs.uniform(-0.3, 2.5, size=k)], 1)
    inside = rs.rand(k) < 0.6                                  # most points as the pipeline produces them
    pts[inside, 0] = rs.rand(inside.sum())
    pts[inside, 1] = pts[inside, 0] + rs.rand(inside.sum()) * (1 - pts[inside, 0])
    for res in (1, 2, 3, 4, 5, 6, 7, 8):
        ref = oracle.pi_raster(offs, pts, res)
        out = engine.pi_raster(_dev(torch, offs, torch.int64), _dev(torch, pts, torch.float64), res).cpu().numpy()
        assert out.shape == ref.shape
        empty = np.diff(offs) == 0
        assert np.all(out[empty] == 0.0)
        assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), res
        nz = np.abs(ref) > 1e-9
        assert rel_err(out[nz], ref[nz]).max() < 1e-8, res
    # sixteen diagrams per workgroup (batches of 16 384 diagrams and more; smaller ones above ran one diagram per workgroup)
    reps = 75
    offs_l = np.concatenate([[0], np.cumsum(np.tile(np.diff(offs), reps))]).astype(np.int64)
    pts_l = np.tile(pts, (reps, 1)) * np.repeat(np.linspace(0.9, 1.1, reps), len(pts))[:, None]
    assert len(offs_l) - 1 >= 16384
    for res in (5, 2):
        ref = oracle.pi_raster(offs_l, pts_l, res)
        out = engine.pi_raster(_dev(torch, offs_l, torch.int64), _dev(torch, pts_l, torch.float64), res).cpu().numpy()
        assert np.abs(out - ref).max() <= 1e-11 * max(1.0, np.abs(ref).max()), res
        nz = np.abs(ref) > 1e-9
        assert rel_err(out[nz], ref[nz]).max() < 1e-8, res
    # a batch that is not a multiple of four diagrams, one diagram only, and a strided grid is not needed below 2^24 diagrams
    for B in (1, 2, 3, 5):
        o = engine.pi_raster(_dev(torch, offs[: B + 1], torch.int64), _dev(torch, pts[: offs[B]] if offs[B] else pts[:1], torch.float64), 5).cpu().numpy()
        assert np.abs(o - oracle.pi_raster(offs[: B + 1], pts[: max(int(offs[B]), 1)], 5)).max() <= 1e-11


@pytest.mark.parametrize("fork", ["tlc", "kd"])
def test_pd_from_filtration_golden(torch_cuda, fork):
    torch = torch_cuda
    from tlc_gnn_amd import engine
    d = np.load(os.path.join(G, "pd_from_f.npz"))
    flags = engine.KEEP_ZERO_PERS if fork == "kd" else 0
    r = engine.pd_from_filtration(_dev(torch, d["f_offs"], torch.int64), _dev(torch, d["e_offs"], torch.int64),
                                  _dev(torch, d["edges"], torch.int32), _dev(torch, d["f"], torch.float64), flags)
    r = {k: v.cpu().numpy() for k, v in r.items()}
    node_offs, edge_offs = d["f_offs"], d["e_offs"]
    for g in range(len(d["n"])):
        no, eo = node_offs[g], edge_offs[g]
        n, m = int(d["n"][g]), int(edge_offs[g + 1] - eo)
        c = r["counts"][g]
        up, down, one = r["up"][no:no + c[0]], r["down"][no:no + c[1]], r["one"][eo:eo + c[2]]
        ext0 = r["ext0"][g]
        if fork == "tlc":
            pd0 = np.concatenate([up, ext0[None, :], down, ext0[None, ::-1]])
            assert same_multiset(pd0, ragged_slice(d["tlc_pd0"], d["tlc_pd0_offs"], g)), g     # bit-exact
            assert same_multiset(one, ragged_slice(d["tlc_pd1"], d["tlc_pd1_offs"], g)), g
            rank = r["edge_rank"][eo:eo + m]
            assert (rank >= 0).sum() == d["npos"][g] and (rank < 0).sum() == d["nneg"][g]
            assert sorted(rank[rank >= 0].tolist()) == list(range(int(d["npos"][g])))
        else:
            assert same_multiset(up, ragged_slice(d["kd_ord0"], d["kd_ord0_offs"], g)), g
            assert same_multiset(ext0[None, :], ragged_slice(d["kd_ext0"], d["kd_ext0_offs"], g)), g
            assert same_multiset(down, ragged_slice(d["kd_rel1"], d["kd_rel1_offs"], g)), g
            assert same_multiset(one, ragged_slice(d["kd_ext1"], d["kd_ext1_offs"], g)), g
        assert c[3] == 1


def test_filtration_golden_bit_exact(torch_cuda):
    torch = torch_cuda
    from tlc_gnn_amd import engine
    d = np.load(os.path.join(G, "filtration.npz"))
    rowptr, col, w = csr_from_golden(d)
    g = engine.DeviceGraph(rowptr, col, w)
    for hop in (1, 2, 3):
        sel = np.nonzero(d["hop"] == hop)[0]
        pairs = _dev(torch, d["pairs"][sel], torch.int32)
        offs, ids, f, n, st = [t.cpu().numpy() for t in g.vicinity_filtration(pairs, hop)]
        for k, gi in enumerate(sel):
            ref_ids = ragged_slice(d["ids"], d["offs"], gi)
            ref_f = ragged_slice(d["f"], d["offs"], gi)
            assert n[k] == len(ref_ids), (hop, k)
            assert np.array_equal(ids[offs[k]:offs[k] + n[k]], ref_ids)
            assert np.array_equal(f[offs[k]:offs[k] + n[k]], ref_f), (hop, k)   # bit-exact f
    g.close()


@pytest.mark.parametrize("hop", [1, 2, 3])
def test_end_to_end_golden(torch_cuda, hop):
    torch = torch_cuda
    from tlc_gnn_amd import engine
    d = np.load(os.path.join(G, "e2e.npz"))
    rowptr, col, w = csr_from_golden(d)
    g = engine.DeviceGraph(rowptr, col, w)
    out, st = g.pd_pi_batch(_dev(torch, d["pairs"], torch.int32), hop)
    out, st = out.cpu().numpy(), st.cpu().numpy()
    ref = d["pi_hop%d" % hop]
    assert np.array_equal(st.astype(np.int64), d["cls_hop%d" % hop])        # the reference's exception classes
    assert np.array_equal(out == 0, ref == 0)
    nz = ref != 0
    assert rel_err(out[nz], ref[nz]).max() < 1e-8                           # north_star: 1e-5 relative
    g.close()


def test_pubmed_sample_vs_reference(torch_cuda):
    torch = torch_cuda
    from tlc_gnn_amd import engine, synth
    d = np.load(os.path.join(G, "pubmed_sample.npz"))
    n, e, k, hop, _ = synth.shaped_graph("PubMed")
    rowptr, col, w = synth.edges_to_csr(n, e, k)
    g = engine.DeviceGraph(rowptr, col, w)
    out, st = g.pd_pi_batch(_dev(torch, d["pairs"], torch.int32), 2)
    out, st = out.cpu().numpy(), st.cpu().numpy()
    ref = d["pi"]
    assert (st == 0).all()
    nz = ref != 0
    assert np.array_equal(out == 0, ref == 0)
    assert rel_err(out[nz], ref[nz]).max() < 1e-8
    g.close()


def test_full_pubmed_batch_vs_oracle(torch_cuda):
    """BASELINE configs[1] at full size: bench.py's own PI-A batch (bench.build_workload(0): the training graph of the PubMed-shaped
    synthetic, all 37 676 train-positive pairs, hop 2) -- the stream-ordered call against the oracle, and then the PIPELINED mode
    the bench's timed region uses: seven batches through tlc_pd_pi_batch_async (three workspaces in turn, second halves deferred,
    early pass + LARGE tier + divide and conquer active) + one join, every output buffer bit-equal to the stream-ordered rows."""
    torch = torch_cuda
    import bench
    from tlc_gnn_amd import engine
    from oracle import oracle
    wl = bench.build_workload(0)
    rowptr, col, w, pairs = wl["rowptr"], wl["col"], wl["w"], wl["pi_pairs"]
    assert pairs.shape == (37676, 2) and wl["hop"] == 2
    g = engine.DeviceGraph(rowptr, col, w)
    d_pairs = _dev(torch, pairs, torch.int32)
    out_t, st_t = g.pd_pi_batch(d_pairs, 2)
    out2, st2 = g.pd_pi_batch(d_pairs, 2)
    assert torch.equal(out_t, out2) and torch.equal(st_t, st2)              # deterministic, bit for bit
    out, st = out_t.cpu().numpy(), st_t.cpu().numpy()
    ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, 2, n_threads=0)
    assert np.array_equal(st, rst)
    assert np.array_equal(out == 0, ref == 0)
    nz = ref != 0
    assert rel_err(out[nz], ref[nz]).max() < 1e-8
    stats = g.stats()
    assert stats["tier_small"] + stats["tier_mid"] + stats["tier_medium"] + stats["tier_large"] + stats["tier_huge"] == len(pairs)
    assert stats["tier_large"] > 0 and g.dc_stats()[0] > 0                  # the LARGE tier and its divide and conquer are in play
    # pipelined: seven batches in a row, a buffer of its own each (a batch in flight owns its buffers until the join)
    bufs = [(torch.empty_like(out_t), torch.empty_like(st_t)) for _ in range(7)]
    for o, s in bufs:
        o.fill_(-1.0); s.fill_(77)
    torch.cuda.synchronize()
    for o, s in bufs:
        g.pd_pi_batch(d_pairs, 2, out=o, status=s, async_=True)
    g.join()
    torch.cuda.synchronize()
    for k, (o, s) in enumerate(bufs):
        assert torch.equal(s, st_t), k
        assert torch.equal(o, out_t), k                                     # bit for bit == the stream-ordered call (<= 1e-8 of the oracle above)
    got = bufs[-1][0].cpu().numpy()
    assert rel_err(got[nz], ref[nz]).max() < 1e-8
    g.close()


def test_ties_take_the_exact_fallback(torch_cuda):
    """All-equal weights: every shortest path is tied, the chain walk must hand over to the node-sourced
    Bellman-Ford and still match the oracle bit for bit (filtration) / 1e-8 (image)."""
    torch = torch_cuda
    from tlc_gnn_amd import engine, synth
    from oracle import oracle
    n = 600
    e = synth.holme_kim_edges(n, 2400, triad_p=0.5, seed=11)
    kappa = np.round(np.random.RandomState(3).uniform(-0.5, 0.9, size=len(e)), 1)   # one decimal: many exact ties
    rowptr, col, w = synth.edges_to_csr(n, e, kappa)
    g = engine.DeviceGraph(rowptr, col, w)
    pairs = e[:500].astype(np.int32)
    for hop in (1, 2):
        offs, ids, f, nn, st = [t.cpu().numpy() for t in g.vicinity_filtration(_dev(torch, pairs, torch.int32), hop)]
        o_offs, o_ids, o_f, o_n, o_m, o_st = oracle.vicinity_filtration(rowptr, col, w, pairs, hop)
        assert np.array_equal(nn, o_n) and np.array_equal(st, o_st)
        for k in range(len(pairs)):
            assert np.array_equal(f[offs[k]:offs[k] + nn[k]], o_f[o_offs[k]:o_offs[k] + o_n[k]]), (hop, k)
        out, st = g.pd_pi_batch(_dev(torch, pairs, torch.int32), hop)
        ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, hop, n_threads=0)
        out = out.cpu().numpy()
        assert np.array_equal(st.cpu().numpy(), rst)
        nz = ref != 0
        assert np.array_equal(out == 0, ref == 0)
        assert rel_err(out[nz], ref[nz]).max() < 1e-8
    assert g.stats()["tie_fallback_sources"] > 0
    g.close()


def test_variant_flags_vs_oracle(torch_cuda):
    torch = torch_cuda
    from tlc_gnn_amd import engine
    from oracle import oracle
    d = np.load(os.path.join(G, "e2e.npz"))
    rowptr, col, w = csr_from_golden(d)
    g = engine.DeviceGraph(rowptr, col, w)
    pairs = d["pairs"].astype(np.int32)
    for flags in (engine.INCLUDE_ROOTS | engine.NORM_EPS, engine.KEEP_ZERO_PERS | engine.PI_ORD0_EXT1, engine.NO_EXT1):
        for hop in (1, 2):
            out, st = g.pd_pi_batch(_dev(torch, pairs, torch.int32), hop, flags=flags)
            ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, hop, flags=flags, n_threads=0)
            out = out.cpu().numpy()
            assert np.array_equal(st.cpu().numpy(), rst), (flags, hop)
            nz = ref != 0
            assert np.array_equal(out == 0, ref == 0), (flags, hop)
            assert rel_err(out[nz], ref[nz]).max() < 1e-8
    # other resolutions
    for res in (3, 8):
        out, st = g.pd_pi_batch(_dev(torch, pairs, torch.int32), 2, res=res)
        ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, 2, res=res, n_threads=0)
        out = out.cpu().numpy()
        nz = ref != 0
        assert np.array_equal(out == 0, ref == 0)
        assert rel_err(out[nz], ref[nz]).max() < 1e-8
    g.close()


def test_empty_and_degenerate_batches(torch_cuda):
    torch = torch_cuda
    from tlc_gnn_amd import engine
    rowptr = np.array([0, 1, 2, 2], dtype=np.int32)      # 0-1 edge, node 2 isolated
    col = np.array([1, 0], dtype=np.int32)
    w = np.array([1.5, 1.5])
    g = engine.DeviceGraph(rowptr, col, w)
    out, st = g.pd_pi_batch(torch.zeros((0, 2), dtype=torch.int32, device="cuda"), 1)
    assert out.shape == (0, 25)
    pairs = torch.tensor([[0, 1], [0, 2], [2, 2], [0, 0], [5, 0], [-1, 1]], dtype=torch.int32, device="cuda")
    out, st = g.pd_pi_batch(pairs, 1)
    from oracle import oracle
    ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs.cpu().numpy(), 1)
    assert st.cpu().tolist() == rst.tolist() == [3, 1, 1, 0, 1, 1]   # ZeroDivision / KeyError classes; (0,0) is a real row
    out = out.cpu().numpy()
    assert np.array_equal(out == 0, ref == 0)
    assert rel_err(out[ref != 0], ref[ref != 0]).max() < 1e-8
    g.close()


def test_huge_tier_and_photo_shaped_hop1(torch_cuda):
    """Dense graph (Photo-shaped core): hop-1 vicinities with thousands of nodes / >4096 edges take the HUGE tier
    (state in an HBM scratch slot) and the LARGE tier; both must match the oracle."""
    torch = torch_cuda
    from tlc_gnn_amd import engine, synth
    from oracle import oracle
    n = 3000
    e = synth.holme_kim_edges(n, 90000, triad_p=0.6, seed=13)        # mean degree 60, hubs in the hundreds
    kappa = synth.curvature_array(e, seed=13)
    rowptr, col, w = synth.edges_to_csr(n, e, kappa)
    deg = np.diff(rowptr)
    hubs = np.argsort(-deg)[:12]
    pairs = np.array([[a, b] for a in hubs[:6] for b in hubs[6:]] + e[:150].tolist(), dtype=np.int32)
    g = engine.DeviceGraph(rowptr, col, w)
    for hop in (1, 2):
        sel = pairs if hop == 1 else pairs[:6]
        out, st = g.pd_pi_batch(_dev(torch, sel, torch.int32), hop)
        stats = g.stats()
        ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, sel, hop, n_threads=0)
        out = out.cpu().numpy()
        assert np.array_equal(st.cpu().numpy(), rst), hop
        nz = ref != 0
        assert np.array_equal(out == 0, ref == 0)
        assert rel_err(out[nz], ref[nz]).max() < 1e-8
        if hop == 2:
            assert stats["tier_huge"] > 0, stats
    # the same sizes through tlc_pd_from_filtration (HUGE tier of the pdf kernel)
    offs, ids, f, nn, mm, sst = oracle.vicinity_filtration(rowptr, col, w, pairs[:3], 2)
    for k in range(3):
        S = ids[offs[k]:offs[k] + nn[k]]
        fv = f[offs[k]:offs[k] + nn[k]]
        loc = -np.ones(n, dtype=np.int64); loc[S] = np.arange(len(S))
        mask = (loc[e[:, 0]] >= 0) & (loc[e[:, 1]] >= 0)
        le = np.stack([loc[e[mask, 0]], loc[e[mask, 1]]], axis=1).astype(np.int32)
        r = engine.pd_from_filtration(_dev(torch, [0, len(S)], torch.int64), _dev(torch, [0, len(le)], torch.int64),
                                      _dev(torch, le, torch.int32), _dev(torch, fv, torch.float64), 0)
        o = oracle.pd_from_filtration([0, len(S)], [0, len(le)], le, fv, 0)
        c, oc = r["counts"][0].cpu().numpy(), o["counts"][0]
        assert np.array_equal(c, oc)
        assert same_multiset(r["up"][:c[0]].cpu().numpy(), o["up"][:oc[0]])
        assert same_multiset(r["down"][:c[1]].cpu().numpy(), o["down"][:oc[1]])
        assert same_multiset(r["one"][:c[2]].cpu().numpy(), o["one"][:oc[2]])
    g.close()


def test_more_than_one_chunk_of_pairs(torch_cuda):
    """> 2^20 pairs: the batch is processed in chunks that reuse the handle's workspace; rows must not depend on it."""
    torch = torch_cuda
    from tlc_gnn_amd import engine
    d = np.load(os.path.join(G, "e2e.npz"))
    rowptr, col, w = csr_from_golden(d)
    g = engine.DeviceGraph(rowptr, col, w)
    base = d["pairs"].astype(np.int32)
    reps = (1 << 20) // len(base) + 2
    pairs = np.tile(base, (reps, 1))
    assert len(pairs) > (1 << 20)
    out, st = g.pd_pi_batch(_dev(torch, pairs, torch.int32), 2)
    ref = torch.from_numpy(d["pi_hop2"])
    one = out[: len(base)].cpu()
    assert rel_err(one.numpy()[ref.numpy() != 0], ref.numpy()[ref.numpy() != 0]).max() < 1e-8
    tiled = out.view(reps, len(base), 25)
    assert bool((tiled == tiled[0:1]).all())                      # every repetition bit-identical to the first
    assert bool((st.view(reps, len(base)) == st[: len(base)].view(1, -1)).all())
    assert g.stats()["chunks"] == 2
    g.close()


def _long_cycle_graphs(seed):
    """Graphs whose spanning trees are deep: rings, rings with a few chords, ladders and lollipops.  Their Pos edges close
    loops of tens to hundreds of tree edges, so the cycle swap runs past its 64-step path records (serial fallback), parks a
    walker above the root and everts long paths -- none of which the hub-centred vicinities of the other tests reach."""
    rs = np.random.RandomState(seed)
    graphs = []
    for n in (3, 5, 64, 65, 66, 130, 300, 700, 1500, 3000):
        ring = np.stack([np.arange(n), (np.arange(n) + 1) % n], 1)
        graphs.append((n, ring))
        if n >= 64:
            ch = rs.randint(0, n, size=(max(2, n // 40), 2))
            ch = ch[(ch[:, 0] != ch[:, 1]) & (np.abs(ch[:, 0] - ch[:, 1]) % n > 1) & (np.abs(ch[:, 0] - ch[:, 1]) % n < n - 1)]
            e = np.concatenate([ring, ch])
            e = np.unique(np.sort(e, 1), axis=0)
            graphs.append((n, e))
    for k in (40, 200, 480):                                  # ladder: two paths joined by rungs
        a = np.arange(k)
        e = np.concatenate([np.stack([a[:-1], a[1:]], 1), np.stack([a[:-1] + k, a[1:] + k], 1), np.stack([a, a + k], 1)[::3]])
        graphs.append((2 * k, e))
    for k in (100, 400):                                      # lollipop: a long tail into a dense blob
        tail = np.stack([np.arange(k - 1), np.arange(1, k)], 1)
        blob = np.array([(i, j) for i in range(k, k + 12) for j in range(i + 1, k + 12)])
        e = np.concatenate([tail, [[k - 1, k]], blob, [[0, k + 5]]])
        graphs.append((k + 12, e))
    out = []
    for n, e in graphs:
        for mode in ("random", "monotone", "ties"):
            if mode == "random":
                f = rs.rand(n)
            elif mode == "monotone":
                f = np.arange(n) / max(n - 1, 1) + 1e-3 * rs.rand(n)
            else:
                f = rs.randint(0, 4, size=n) / 4.0
            out.append((n, e.astype(np.int32), f.astype(np.float64)))
    return out


@pytest.mark.parametrize("flags", [0, 1])
def test_cycle_swap_long_loops_vs_oracle(torch_cuda, flags):
    torch = torch_cuda
    from tlc_gnn_amd import engine
    from oracle import oracle
    gs = _long_cycle_graphs(11)
    node_offs = np.concatenate([[0], np.cumsum([g[0] for g in gs])]).astype(np.int64)
    edge_offs = np.concatenate([[0], np.cumsum([len(g[1]) for g in gs])]).astype(np.int64)
    edges = np.concatenate([g[1] for g in gs]).astype(np.int32)
    f = np.concatenate([g[2] for g in gs])
    ref = oracle.pd_from_filtration(node_offs, edge_offs, edges, f, flags)
    got = engine.pd_from_filtration(_dev(torch, node_offs, torch.int64), _dev(torch, edge_offs, torch.int64),
                                    _dev(torch, edges, torch.int32), _dev(torch, f, torch.float64), flags)
    got = {k: v.cpu().numpy() for k, v in got.items()}
    assert np.array_equal(got["counts"], ref["counts"])
    assert np.array_equal(got["ext0"], ref["ext0"])
    for g in range(len(gs)):
        no, eo = node_offs[g], edge_offs[g]
        c = ref["counts"][g]
        for key, base, k in (("up", no, c[0]), ("down", no, c[1]), ("one", eo, c[2])):
            assert same_multiset(got[key][base:base + k], ref[key][base:base + k]), (g, key, gs[g][0])   # bit-exact


@pytest.mark.parametrize("name", ["Cora", "PPI", "Photo", "Computers"])
def test_other_baseline_shapes_hop1_vs_oracle(torch_cuda, name):
    """BASELINE.json's other configurations (Cora plumbing case, PPI graphs, Amazon Photo / Computers; TLCGNN.py:102 gives them
    hop 1): positives and sampled negatives of the shaped synthetic graph against the oracle -- status bytes equal, zero
    patterns equal, images within 1e-8 relative."""
    torch = torch_cuda
    from tlc_gnn_amd import engine, synth
    from oracle import oracle
    n, edges, kappa, hop, _ = synth.shaped_graph(name)
    assert hop == 1
    rowptr, col, w = synth.edges_to_csr(n, edges, kappa)
    rs = np.random.RandomState(5)
    k = 1500 if name in ("Photo", "Computers") else 3000
    pos = edges[rs.permutation(len(edges))[:k]]
    neg = rs.randint(0, n, size=(k // 2, 2))
    pairs = np.concatenate([pos, pos[: k // 4, ::-1], neg, [[0, 0], [n - 1, n - 1]]]).astype(np.int32)
    g = engine.DeviceGraph(rowptr, col, w)
    out, st = g.pd_pi_batch(torch.from_numpy(pairs).cuda(), hop)
    out, st = out.cpu().numpy(), st.cpu().numpy()
    ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, hop, n_threads=0)
    assert np.array_equal(st, rst)
    assert np.array_equal(out == 0, ref == 0)
    nz = ref != 0
    assert nz.any() and rel_err(out[nz], ref[nz]).max() < 1e-8
    g.close()


@pytest.mark.parametrize("flags", [0, 1])
def test_edge_counts_just_above_a_power_of_two(torch_cuda, flags):
    """The edge sorts split a count just above a power of two into two bitonic runs merged by rank (pd_pipeline.hip,
    sort_padded): counts around 2^k and 2^k + 2^(k-2) in every tier, with continuous and with heavily tied filtration values
    (equal keys across the two runs), both forks, bit-exact against the oracle."""
    torch = torch_cuda
    from tlc_gnn_amd import engine
    from oracle import oracle
    rs = np.random.RandomState(17)
    gs = []
    for m in (255, 256, 257, 300, 319, 320, 321, 512, 513, 600, 640, 641, 1023, 1025, 1100, 2049, 2270, 2560, 2561, 4000):
        n = max(8, int(m * 0.68))                              # the batch's heaviest vicinity: 1 539 nodes, 2 270 edges
        par = np.array([rs.randint(0, k) for k in range(1, n)])
        tree = np.stack([par, np.arange(1, n)], 1)
        extra = set()
        while len(extra) < m - (n - 1):
            a, b = rs.randint(0, n, 2)
            if a != b and (min(a, b), max(a, b)) not in extra and par[max(a, b) - 1] != min(a, b):
                extra.add((min(a, b), max(a, b)))
        e = np.concatenate([tree, np.array(sorted(extra)).reshape(-1, 2)]).astype(np.int32)
        assert len(e) == m
        e = e[rs.permutation(m)]
        for mode in ("random", "ties"):
            f = rs.rand(n) if mode == "random" else rs.randint(0, 5, size=n) / 4.0
            gs.append((n, e, f.astype(np.float64)))
    node_offs = np.concatenate([[0], np.cumsum([g[0] for g in gs])]).astype(np.int64)
    edge_offs = np.concatenate([[0], np.cumsum([len(g[1]) for g in gs])]).astype(np.int64)
    edges = np.concatenate([g[1] for g in gs]).astype(np.int32)
    f = np.concatenate([g[2] for g in gs])
    ref = oracle.pd_from_filtration(node_offs, edge_offs, edges, f, flags)
    got = engine.pd_from_filtration(_dev(torch, node_offs, torch.int64), _dev(torch, edge_offs, torch.int64),
                                    _dev(torch, edges, torch.int32), _dev(torch, f, torch.float64), flags)
    got = {k: v.cpu().numpy() for k, v in got.items()}
    assert np.array_equal(got["counts"], ref["counts"])
    assert np.array_equal(got["ext0"], ref["ext0"])
    for g in range(len(gs)):
        no, eo = node_offs[g], edge_offs[g]
        c = ref["counts"][g]
        for key, base, k in (("up", no, c[0]), ("down", no, c[1]), ("one", eo, c[2])):
            assert same_multiset(got[key][base:base + k], ref[key][base:base + k]), (g, key, gs[g][0], len(gs[g][1]))


def test_pd_from_filtration_rejects_oversized_graph(torch_cuda):
    """More than 65 535 nodes do not fit the packed local ids: the graph is skipped and says so (counts row = -1); its
    neighbours in the batch are still computed."""
    torch = torch_cuda
    from tlc_gnn_amd import engine
    from oracle import oracle
    n_big = 70000
    big = np.stack([np.arange(n_big - 1), np.arange(1, n_big)], 1)
    tri = np.array([[0, 1], [1, 2], [0, 2]])
    node_offs = np.array([0, 3, 3 + n_big, 6 + n_big], dtype=np.int64)
    edge_offs = np.array([0, 3, 3 + len(big), 6 + len(big)], dtype=np.int64)
    edges = np.concatenate([tri, big, tri]).astype(np.int32)
    f = np.concatenate([[0.1, 0.5, 0.9], np.linspace(0, 1, n_big), [0.3, 0.2, 0.7]])
    got = engine.pd_from_filtration(_dev(torch, node_offs, torch.int64), _dev(torch, edge_offs, torch.int64),
                                    _dev(torch, edges, torch.int32), _dev(torch, f, torch.float64), 0)
    c = got["counts"].cpu().numpy()
    assert (c[1] == -1).all()
    ref = oracle.pd_from_filtration(node_offs[[0, 1]], edge_offs[[0, 1]], tri.astype(np.int32), f[:3], 0)
    assert np.array_equal(c[0], ref["counts"][0])
    assert same_multiset(got["one"].cpu().numpy()[:c[0][2]], ref["one"][:c[0][2]])
    assert c[2][2] == 1 and c[2][3] == 1


def test_early_pass_overflow_takes_the_ordinary_path(torch_cuda):
    """A batch whose every pair is predicted heavy and LARGE-tier: the early pass (the predicted-heavy pairs counted and written
    ahead of the batch, api.hip run_chunk) has 512 candidate and 256 arena slots, everything beyond them must come out of
    the ordinary COUNT -> scan -> FILL -> tier path with the same rows."""
    torch = torch_cuda
    from tlc_gnn_amd import engine, synth
    from oracle import oracle
    rs = np.random.RandomState(11)
    L = 640                                                   # hub 0 + a ring of leaves with a few chords
    e = [(0, k) for k in range(1, L + 1)] + [(k, k % L + 1) for k in range(1, L + 1)]
    e += [(int(a), int(b)) for a, b in rs.randint(1, L + 1, size=(200, 2)) if a != b]
    e = np.unique(np.sort(np.array(e, dtype=np.int64), axis=1), axis=0)
    kappa = rs.uniform(-0.5, 0.9, size=len(e))
    rowptr, col, w = synth.edges_to_csr(L + 1, e, kappa)
    g = engine.DeviceGraph(rowptr, col, w)
    distinct = e[rs.permutation(len(e))[:192]].astype(np.int32)
    reps = 24
    pairs = np.tile(distinct, (reps, 1))                      # 4 608 pairs, every vicinity = the whole graph (641 nodes)
    perm = rs.permutation(len(pairs))
    out, st = g.pd_pi_batch(_dev(torch, pairs[perm], torch.int32), 2)
    out, st = out.cpu().numpy(), st.cpu().numpy()
    stats = g.stats()
    assert stats["tier_large"] == len(pairs)
    ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, distinct, 2, n_threads=0)
    inv = np.empty_like(perm)
    inv[perm] = np.arange(len(perm))
    out, st = out[inv].reshape(reps, len(distinct), 25), st[inv].reshape(reps, len(distinct))
    assert (st == rst[None, :]).all()
    assert (out == out[0:1]).all()                            # early slot or ordinary path: bit-identical rows
    nz = ref != 0
    assert nz.any() and np.array_equal(out[0] == 0, ~nz)
    assert rel_err(out[0][nz], ref[nz]).max() < 1e-8
    g.close()


@pytest.mark.parametrize("extract", [1, 0])
def test_count_pass_arena_overflow_falls_back_to_scan_and_fill(torch_cuda, extract):
    """COUNT writes the vicinities below the heavy tiers into the arena itself (extraction: at the cursor of a per-workgroup
    region, then into blocks from a bump counter; breadth-first kernels: MID / MEDIUM at bump-allocated offsets); the arena is
    sized from a guess before the sizes are known.  First call on a fresh handle: the guess (made far too small here) runs
    out, the chunk must take the scan + FILL path; second call: the arena has grown, COUNT's own writes are used.  Same rows
    both times, equal to the oracle's."""
    torch = torch_cuda
    from tlc_gnn_amd import engine, synth
    from oracle import oracle
    rs = np.random.RandomState(5)
    L = 300                                                   # hub 0 + a ring of leaves + chords: every vicinity is the graph
    e = [(0, k) for k in range(1, L + 1)] + [(k, k % L + 1) for k in range(1, L + 1)]
    e += [(int(a), int(b)) for a, b in rs.randint(1, L + 1, size=(120, 2)) if a != b]
    e = np.unique(np.sort(np.array(e, dtype=np.int64), axis=1), axis=0)
    rowptr, col, w = synth.edges_to_csr(L + 1, e, rs.uniform(-0.5, 0.9, size=len(e)))
    g = engine.DeviceGraph(rowptr, col, w)
    g.set_option("extract", extract)
    g.set_option("x_arena", 256)                               # (extraction: 400 regions of 256 entries + a bump area of 12 800)
    distinct = e[rs.permutation(len(e))[:100]].astype(np.int32)
    pairs = np.tile(distinct, (4, 1))                         # 400 pairs x ~1 400 directed entries >> 65 536
    dp = _dev(torch, pairs, torch.int32)
    out1, st1 = g.pd_pi_batch(dp, 2)
    s1 = g.stats()
    out2, st2 = g.pd_pi_batch(dp, 2)
    s2 = g.stats()
    assert s1["tier_medium"] == len(pairs) and s2["tier_medium"] == len(pairs)
    assert s1["induced_entries"] == s2["induced_entries"] > 65536
    assert bool((out1 == out2).all()) and bool((st1 == st2).all())
    ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, distinct, 2, n_threads=0)
    o = out2.cpu().numpy().reshape(4, len(distinct), 25)
    assert (o == o[0:1]).all() and (st2.cpu().numpy().reshape(4, -1) == rst[None, :]).all()
    nz = ref != 0
    assert nz.any() and rel_err(o[0][nz], ref[nz]).max() < 1e-8
    g.close()


@pytest.mark.parametrize("res", [5, 8])
def test_variant_flags_through_the_early_and_handoff_paths(torch_cuda, res):
    """The flag variants and a non-default resolution on a batch large enough for the early pass (>= 4 096 pairs, hub pairs
    whose vicinities reach the LARGE tier) and with MID / MEDIUM vicinities that take the hand-off to the swap kernel."""
    torch = torch_cuda
    from tlc_gnn_amd import engine, synth
    from oracle import oracle
    e = synth.holme_kim_edges(3000, 9000, triad_p=0.4, seed=77)
    rs = np.random.RandomState(3)
    rowptr, col, w = synth.edges_to_csr(3000, e, rs.uniform(-0.5, 0.9, size=len(e)))
    g = engine.DeviceGraph(rowptr, col, w)
    pairs = np.concatenate([e[rs.permutation(len(e))[:4300]], rs.randint(0, 3000, size=(200, 2))]).astype(np.int32)
    seen_large = False
    for flags in (engine.INCLUDE_ROOTS | engine.NORM_EPS | engine.UNREACHABLE_100, engine.KEEP_ZERO_PERS | engine.PI_ORD0_EXT1):
        out, st = g.pd_pi_batch(_dev(torch, pairs, torch.int32), 2, flags=flags, res=res)
        stats = g.stats()
        seen_large |= stats["tier_large"] > 0
        assert stats["tier_mid"] > 0 and stats["tier_medium"] > 0
        ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, 2, flags=flags, res=res, n_threads=0)
        out = out.cpu().numpy()
        assert np.array_equal(st.cpu().numpy(), rst), flags
        nz = ref != 0
        assert np.array_equal(out == 0, ref == 0), flags
        assert rel_err(out[nz], ref[nz]).max() < 1e-8
    assert seen_large
    g.close()


def test_hiv_sized_batch_of_molecule_graphs_vs_oracle(torch_cuda):
    """BASELINE config 5 at its real size outside bench.py: 41 127 HIV-shaped molecule graphs (synth.hiv_shaped_molecules, the batch of
    bench.py's pdgnn block; Knowledge_Distillation/data_utils_GC.py:98-170, degree filtration) in ONE tlc_pd_from_filtration call,
    both forks' flags: the counts of every graph equal the oracle's, and the diagrams of 2 000 graphs spread over the batch --
    Ord0 / Rel1 / Ext0 / Ext1 -- are the oracle's as sorted multisets, bit for bit."""
    torch = torch_cuda
    from tlc_gnn_amd import engine, synth
    from oracle import oracle
    edges, f, node_offs, edge_offs = synth.hiv_shaped_molecules(41127, 1234)
    assert len(node_offs) == 41128 and node_offs[-1] > 1000000
    d_no, d_eo = _dev(torch, node_offs, torch.int64), _dev(torch, edge_offs, torch.int64)
    d_e, d_f = _dev(torch, edges, torch.int32), _dev(torch, f, torch.float64)
    check = np.random.RandomState(0).permutation(41127)[:2000]
    for flags in (0, engine.KEEP_ZERO_PERS):
        ref = oracle.pd_from_filtration(node_offs, edge_offs, edges, f, flags)
        got = {k: v.cpu().numpy() for k, v in engine.pd_from_filtration(d_no, d_eo, d_e, d_f, flags).items()}
        assert np.array_equal(got["counts"], ref["counts"]), flags
        assert np.array_equal(got["ext0"], ref["ext0"]), flags
        for gph in check.tolist():
            no, eo = node_offs[gph], edge_offs[gph]
            c = ref["counts"][gph]
            for key, base, cnt in (("up", no, c[0]), ("down", no, c[1]), ("one", eo, c[2])):
                assert same_multiset(got[key][base:base + cnt], ref[key][base:base + cnt]), (flags, gph, key)

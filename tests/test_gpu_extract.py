"""GPU: the ball-list extraction of the vicinities (csrc/extract.hip) against the breadth-first kernels it replaces
(csrc/vicinity.hip) and against the CPU oracle.

The packed subgraph of a pair may list its directed entries in any order (the tier kernels do not care), so what has to agree
is: |S| and the entry count per pair, the status byte (= the reference's exception class), and the image rows -- bit-exact
status / sizes, images within 1e-8 relative of the oracle (north_star: 1e-5) and within 1e-12 between the modes (the order of
the diagram points, hence of the fp64 image sum, may differ where keys tie)."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu


def _run(g, torch, pairs, hop, **opts):
    for k, v in opts.items():
        g.set_option(k, v)
    out, st = g.pd_pi_batch(torch.as_tensor(np.ascontiguousarray(pairs, dtype=np.int32)).cuda(), hop)
    n, m2 = g.sizes(len(pairs))
    stats = g.stats()
    for k in opts:
        g.set_option(k, 1)
    return out.cpu().numpy(), st.cpu().numpy(), n, m2, stats


def _mixed_pairs(n, edges, rs, k_pos, k_neg):
    pos = edges[rs.permutation(len(edges))[:k_pos]]
    neg = rs.randint(0, n, size=(k_neg, 2))
    selfp = np.stack([np.arange(0, n, max(1, n // 50))] * 2, 1)
    pairs = np.concatenate([pos, pos[:200, ::-1], neg, selfp]).astype(np.int32)
    return pairs[rs.permutation(len(pairs))]


@pytest.mark.parametrize("shape,scale,hop", [("PubMed", 0.35, 2), ("PubMed", 0.35, 1), ("Photo", 0.2, 1), ("Cora", 1.0, 2),
                                             ("Cora", 1.0, 3), ("PubMed", 0.1, 3), ("Cora", 0.5, 4)])     # (round 5: hop >= 3 through the ball lists too)
def test_extraction_modes_agree_and_match_the_oracle(shape, scale, hop):
    import torch
    from tlc_gnn_amd import engine, synth
    from oracle import oracle
    n, edges, kappa, _, _ = synth.shaped_graph(shape, scale=scale)
    rowptr, col, w = synth.edges_to_csr(n, edges, kappa)
    rs = np.random.RandomState(3)
    pairs = _mixed_pairs(n, edges, rs, 5000, 1500)
    # a node without edges and ids outside the graph: KeyError rows
    iso = np.flatnonzero(np.diff(rowptr) == 0)
    extra = [[-1, 3], [n, 0]] + ([[int(iso[0]), 1]] if len(iso) else [])
    pairs = np.concatenate([pairs, np.array(extra, dtype=np.int32)])
    g = engine.DeviceGraph(rowptr, col, w)
    new = _run(g, torch, pairs, hop)
    light = _run(g, torch, pairs, hop, heavy=0)
    old = _run(g, torch, pairs, hop, extract=0)
    g.close()
    for other, name in ((light, "heavy=0"), (old, "extract=0")):
        assert np.array_equal(new[1], other[1]), name                   # status bytes
        assert np.array_equal(new[2], other[2]), name                   # |S|
        assert np.array_equal(new[3], other[3]), name                   # induced directed entries
        assert np.abs(new[0] - other[0]).max() <= 1e-12 * max(1.0, np.abs(other[0]).max()), name
        assert new[4]["tier_small"] == other[4]["tier_small"] and new[4]["tier_medium"] == other[4]["tier_medium"], name
    ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, hop, n_threads=0)
    assert np.array_equal(new[1], rst)
    assert np.array_equal(new[0] == 0, ref == 0)
    nz = ref != 0
    assert rel_err(new[0][nz], ref[nz]).max() < 1e-8
    assert (rst == 1).sum() >= 2 and (rst == 0).sum() > 1000


@pytest.mark.parametrize("shape,scale,hop", [("Photo", 0.3, 1), ("PubMed", 0.3, 2), ("Cora", 1.0, 2)])
@pytest.mark.parametrize("kd", [True, False])
def test_filtration_outputs_through_the_extraction(shape, scale, hop, kd):
    """tlc_vicinity_filtration (ids, f, induced edges; the PDGNN fork's flags: roots always members, sentinel 100, eps
    normalisation -- data_utils_LP.py:35-65,105-118) through the ball-list extraction and through the breadth-first kernels
    (option extract=0): every output array bit for bit, and the oracle's.  Pairs: edges, far pairs (roots not in the intersection:
    merged into the id list), (u, u), ids outside the graph."""
    import torch
    from tlc_gnn_amd import engine, synth, _lib
    from oracle import oracle
    n, edges, kappa, _, _ = synth.shaped_graph(shape, scale=scale)
    rowptr, col, w = synth.edges_to_csr(n, edges, kappa)
    rs = np.random.RandomState(11)
    pairs = np.concatenate([_mixed_pairs(n, edges, rs, 1500, 1200), np.array([[-1, 3], [n, 0]], dtype=np.int32)])
    flags = (_lib.INCLUDE_ROOTS | _lib.NORM_EPS | _lib.UNREACHABLE_100) if kd else 0
    oflags = (oracle.INCLUDE_ROOTS | oracle.NORM_EPS | oracle.UNREACHABLE_100) if kd else 0
    g = engine.DeviceGraph(rowptr, col, w)
    dev = torch.as_tensor(pairs).cuda()
    res = {}
    for x in (1, 0):
        g.set_option("extract", x)
        res[x] = [t.cpu().numpy() for t in g.vicinity_filtration(dev, hop, flags=flags, cap=600, edge_cap=6000)]
    g.set_option("extract", 1)
    g.close()
    offs, ids, f, nn, st, eoffs, e, mm = res[1]
    for a, b, name in zip(res[1], res[0], ("offs", "ids", "f", "n", "status", "eoffs", "edges", "m")):
        if name in ("ids", "f"):
            for k in range(len(pairs)):
                if nn[k] > 0:
                    assert np.array_equal(a[offs[k]:offs[k] + nn[k]], b[offs[k]:offs[k] + nn[k]]), (name, k)
        elif name == "edges":
            # (the two kernels list a vicinity's edges in different orders: compared as sorted sets)
            for k in range(len(pairs)):
                if mm[k] > 0:
                    ea = a[eoffs[k]:eoffs[k] + mm[k]]; eb = b[eoffs[k]:eoffs[k] + mm[k]]
                    assert np.array_equal(ea[np.lexsort((ea[:, 1], ea[:, 0]))], eb[np.lexsort((eb[:, 1], eb[:, 0]))]), (name, k)
        else:
            assert np.array_equal(a, b), name
    o_offs, o_ids, o_f, o_n, o_m, o_st, o_eoffs, o_e = oracle.vicinity_filtration(rowptr, col, w, pairs, hop, oflags, edge_cap=6000)
    assert np.array_equal(st, o_st)
    fits = (o_n <= 600) & (o_m <= 6000)
    okm = fits & (st == 0)                                                 # (a failed pair's edge count is reported as 0 here)
    assert np.array_equal(nn[fits], o_n[fits]) and np.array_equal(mm[okm], o_m[okm]) and fits.sum() > 2000
    for k in np.flatnonzero(fits & (o_n > 0))[:1500]:
        assert np.array_equal(ids[offs[k]:offs[k] + nn[k]], o_ids[o_offs[k]:o_offs[k] + o_n[k]]), k
        assert np.array_equal(f[offs[k]:offs[k] + nn[k]], o_f[o_offs[k]:o_offs[k] + o_n[k]]), k
    if kd:
        far = np.flatnonzero((nn == 2) & (mm == 0) & (pairs[:, 0] != pairs[:, 1]))
        assert len(far) > 20                                              # two isolated roots: nothing in the intersection


@pytest.mark.parametrize("n,reach,hop", [(60, 12, 2), (150, -8, 2), (70, 15, 1)])
def test_sweep_segments_of_dense_vicinities(n, reach, hop):
    """The sweep deals the eight-entry segments behind the node records of a batch of 64 members to the lanes: circulant graphs
    (every node 2 * reach neighbours, 20 - 30: three or four segments per member, up to 250 per batch of members, several batches)
    make it go round more than once per batch.  Same rows as the breadth-first kernels, which walk row by row, and as the oracle."""
    import torch
    from tlc_gnn_amd import engine, synth
    from oracle import oracle
    rs = np.random.RandomState(n)
    # (reach < 0: offsets 1..8 and 20..26 instead of 1..reach -- 30 neighbours again, but hop-2 balls of more than 64 nodes)
    offs = list(range(1, reach + 1)) if reach > 0 else list(range(1, 9)) + list(range(20, 27))
    edges = np.array([(i, (i + d) % n) for i in range(n) for d in offs], dtype=np.int64)
    kappa = rs.rand(len(edges)) * 1.6 - 0.8
    rowptr, col, w = synth.edges_to_csr(n, edges, kappa)
    assert 20 <= int(np.diff(rowptr).max()) == 2 * len(offs) < 32
    pairs = np.concatenate([edges[rs.permutation(len(edges))[:300]], rs.randint(0, n, size=(100, 2))]).astype(np.int32)
    g = engine.DeviceGraph(rowptr, col, w)
    new = _run(g, torch, pairs, hop)
    old = _run(g, torch, pairs, hop, extract=0)
    g.close()
    assert np.array_equal(new[1], old[1]) and np.array_equal(new[2], old[2]) and np.array_equal(new[3], old[3])
    assert int(new[2].max()) > (64 if reach < 0 else 25)                 # vicinities of that many such members
    assert np.abs(new[0] - old[0]).max() <= 1e-12 * max(1.0, np.abs(old[0]).max())
    ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, hop, n_threads=0)
    assert np.array_equal(new[1], rst)
    nz = ref != 0
    assert np.array_equal(new[0] == 0, ref == 0) and rel_err(new[0][nz], ref[nz]).max() < 1e-8


def test_heavy_rows_hub_vicinities_and_dense_heavy_core():
    """A graph built around the heavy-row path: a clique-like core of 40 hubs (every heavy-heavy entry comes from the dense
    table), each hub with a fan of leaves, leaves cross-linked (light rows that find several heavy members), pairs hub-hub,
    hub-leaf, leaf-leaf across fans, at hop 1 and 2 -- vicinities from 3 nodes to > 1 000."""
    import torch
    from tlc_gnn_amd import engine, synth
    from oracle import oracle
    rs = np.random.RandomState(9)
    H, fan = 40, 45
    edges = set()
    for a in range(H):
        for b in range(a + 1, H):
            if rs.rand() < 0.6:
                edges.add((a, b))
    nxt = H
    leaves = []
    for h in range(H):
        for _ in range(fan):
            edges.add((h, nxt))
            leaves.append(nxt)
            nxt += 1
    leaves = np.array(leaves)
    for _ in range(1500):                                  # leaves attached to a second hub / to each other
        a = int(leaves[rs.randint(len(leaves))])
        b = int(rs.randint(H)) if rs.rand() < 0.5 else int(leaves[rs.randint(len(leaves))])
        if a != b:
            edges.add((min(a, b), max(a, b)))
    n = nxt + 3                                            # three isolated nodes at the end
    e = np.array(sorted(edges), dtype=np.int64)
    kappa = rs.uniform(-0.5, 0.9, size=len(e))
    rowptr, col, w = synth.edges_to_csr(n, e, kappa)
    assert (np.diff(rowptr) >= 32).sum() >= H
    pairs = np.concatenate([e[rs.permutation(len(e))[:3000]], rs.randint(0, n, size=(500, 2)),
                            np.stack([np.arange(H), (np.arange(H) + 1) % H], 1)]).astype(np.int32)
    g = engine.DeviceGraph(rowptr, col, w)
    for hop in (1, 2):
        new = _run(g, torch, pairs, hop)
        old = _run(g, torch, pairs, hop, extract=0)
        assert np.array_equal(new[1], old[1]) and np.array_equal(new[2], old[2]) and np.array_equal(new[3], old[3])
        ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, hop, n_threads=0)
        assert np.array_equal(new[1], rst)
        nz = ref != 0
        assert np.array_equal(new[0] == 0, ref == 0)
        assert rel_err(new[0][nz], ref[nz]).max() < 1e-8
        assert new[2].max() > (500 if hop == 2 else 20)
    g.close()


def test_asymmetric_or_repeated_heavy_rows_switch_the_heavy_set_off():
    """The mirror emission needs heavy rows that are symmetric and free of repeated columns (csrc/api.hip, build_heavy_set):
    a CSR that violates this at a hub is still processed -- every row is read -- and equals the breadth-first kernels."""
    import torch
    from tlc_gnn_amd import engine
    rs = np.random.RandomState(4)
    n = 300
    rows = [[] for _ in range(n)]
    for x in range(1, 120):                               # hub 0 with 119 neighbours
        rows[0].append((x, 1.0 + 0.001 * x)); rows[x].append((0, 1.0 + 0.001 * x))
    for _ in range(600):
        a, b = rs.randint(1, n, size=2)
        if a != b and all(c != b for c, _ in rows[a]):
            wt = float(rs.uniform(0.5, 1.9))
            rows[a].append((b, wt)); rows[b].append((a, wt))
    rows[0].append((5, 1.005))                            # column 5 twice in the hub's row (and once more in row 5)
    rows[5].append((0, 1.005))
    rowptr = np.zeros(n + 1, dtype=np.int32)
    col, w = [], []
    for x in range(n):
        r = sorted(rows[x])
        col += [c for c, _ in r]; w += [v for _, v in r]
        rowptr[x + 1] = len(col)
    pairs = np.array([[0, x] for x in range(1, 60)] + [[x, x + 1] for x in range(1, 200)], dtype=np.int32)
    g = engine.DeviceGraph(rowptr, np.array(col, dtype=np.int32), np.array(w))
    new = _run(g, torch, pairs, 2)
    old = _run(g, torch, pairs, 2, extract=0)
    g.close()
    assert np.array_equal(new[1], old[1]) and np.array_equal(new[2], old[2]) and np.array_equal(new[3], old[3])
    assert np.abs(new[0] - old[0]).max() <= 1e-12


def test_async_batches_overlap_and_equal_the_stream_ordered_call():
    """tlc_pd_pi_batch_async + tlc_pd_pi_batch_join: batches submitted back to back run on the handle's workspaces in turn (three;
    a fourth submission waits on the host for the first) and must give, bit for bit, what the stream-ordered call gives for the
    same pairs -- whatever workspace a batch lands on, whatever is in flight beside it.  A batch's tier launches are submitted
    behind the NEXT batch's first half (or by the join): the rows must not depend on that either (options defer / n_ws)."""
    import torch
    from tlc_gnn_amd import engine, synth
    n, edges, kappa, _, _ = synth.shaped_graph("PubMed", scale=0.3)
    rowptr, col, w = synth.edges_to_csr(n, edges, kappa)
    rs = np.random.RandomState(12)
    g = engine.DeviceGraph(rowptr, col, w)
    batches = []
    for k in range(7):
        m = [6000, 300, 5000, 4100, 17, 7000, 4500][k]                  # with and without the early pass (>= 4 096 pairs)
        b = np.concatenate([edges[rs.permutation(len(edges))[:m]], rs.randint(0, n, size=(m // 5 + 1, 2))]).astype(np.int32)
        batches.append(torch.as_tensor(b[rs.permutation(len(b))]).cuda())
    want = []
    for b in batches:
        o, s = g.pd_pi_batch(b, 2)
        want.append((o.clone(), s.clone()))
    torch.cuda.synchronize()
    got = [g.pd_pi_batch(b, 2, async_=True) for b in batches]           # seven in a row: the host waits for the oldest as needed
    g.join()
    torch.cuda.synchronize()
    for k, ((o, s), (wo, wst)) in enumerate(zip(got, want)):
        assert torch.equal(s, wst), k
        assert torch.equal(o, wo), k
    # a consumer on the stream after join() sees complete rows without any host synchronisation in between
    o2, s2 = g.pd_pi_batch(batches[0], 2, async_=True)
    o3, s3 = g.pd_pi_batch(batches[2], 2, async_=True)
    g.join()
    tot = o2.sum() + o3.sum()
    assert abs(float(tot) - float(want[0][0].sum() + want[2][0].sum())) <= 1e-9 * abs(float(tot))
    # the stream-ordered call still works with batches in flight, and statistics follow the last call
    o4, _ = g.pd_pi_batch(batches[1], 2, async_=True)
    o5, s5 = g.pd_pi_batch(batches[3], 2)
    assert torch.equal(o5, want[3][0]) and torch.equal(s5, want[3][1])
    g.join()
    torch.cuda.synchronize()
    assert torch.equal(o4, want[1][0])
    assert g.stats()["chunks"] == 1
    # with the speculative launches' reserved slots cut to 8, with kernel events on every 2nd chunk only and with two or four
    # workspaces in turn, the rows are the same
    for opts in ({"spec_cap": 8}, {"timing_every": 2}, {"n_ws": 2}, {"n_ws": 4}):
        for k, v in opts.items():
            g.set_option(k, v)
        if "timing_every" in opts:
            g.set_timing(True)
        got = [g.pd_pi_batch(b, 2, async_=True) for b in batches]
        g.join()
        torch.cuda.synchronize()
        for k, ((o, s), (wo, wst)) in enumerate(zip(got, want)):
            assert torch.equal(s, wst) and torch.equal(o, wo), (opts, k)
        g.set_timing(False)
        for k in opts:
            g.set_option(k, {"timing_every": 1, "n_ws": 3}.get(k, 0))
    # a batch whose second half is still owed: statistics and sizes ask for it themselves, a stream-ordered call submits it first,
    # and a handle may be closed with one pending
    o6, s6 = g.pd_pi_batch(batches[0], 2, async_=True)
    assert g.stats()["chunks"] == 1
    torch.cuda.synchronize()
    assert torch.equal(o6, want[0][0]) and torch.equal(s6, want[0][1])
    o7, _ = g.pd_pi_batch(batches[2], 2, async_=True)
    o8, s8 = g.pd_pi_batch(batches[5], 2)                                # (stream-ordered: behind the pending one)
    torch.cuda.synchronize()
    assert torch.equal(o8, want[5][0]) and torch.equal(s8, want[5][1])
    g.join()
    torch.cuda.synchronize()
    assert torch.equal(o7, want[2][0])
    g.pd_pi_batch(batches[3], 2, async_=True)
    g.close()


def test_async_batches_of_alternating_hop_rebuild_the_ball_lists_behind_the_pending_chunk():
    """A pipelined chunk's second half is deferred; the ball lists belong to ONE hop value and are rebuilt when the next batch
    asks for another.  The rebuild must first submit the pending chunk's second half and wait for it: that half may relaunch
    the extraction (arena overflow -> FILL from the lists the chunk was started with).  Arena regions shrunk so that every
    chunk overflows; with and without the ball-list extraction (the breadth-first path rebuilds the ball-size bounds)."""
    import torch
    from tlc_gnn_amd import engine, synth
    n, edges, kappa, _, _ = synth.shaped_graph("PubMed", scale=0.3)
    rowptr, col, w = synth.edges_to_csr(n, edges, kappa)
    rs = np.random.RandomState(21)
    g = engine.DeviceGraph(rowptr, col, w)
    batches = []
    for k in range(6):
        b = np.concatenate([edges[rs.permutation(len(edges))[:4500]], rs.randint(0, n, size=(600, 2))]).astype(np.int32)
        batches.append((torch.as_tensor(b[rs.permutation(len(b))]).cuda(), 2 if k % 2 == 0 else 1))
    want = []
    for b, hop in batches:
        o, s = g.pd_pi_batch(b, hop)
        want.append((o.clone(), s.clone()))
    torch.cuda.synchronize()
    for opts in ({"x_arena": 64}, {"extract": 0}, {}):
        for k, v in opts.items():
            g.set_option(k, v)
        got = [g.pd_pi_batch(b, hop, async_=True) for b, hop in batches]
        g.join()
        torch.cuda.synchronize()
        for k, ((o, s), (wo, wst)) in enumerate(zip(got, want)):
            assert torch.equal(s, wst), (opts, k)
            assert (o - wo).abs().max() <= 1e-12, (opts, k)
        g.set_option("x_arena", 0); g.set_option("extract", 1)
    # tlc_pd_pi_batch == async + join: a stream-ordered call also makes the stream wait for batches still in flight
    o1, s1 = g.pd_pi_batch(batches[0][0], 2, async_=True)
    o2, s2 = g.pd_pi_batch(batches[2][0], 2, async_=True)
    o3, s3 = g.pd_pi_batch(batches[1][0], 1)
    tot = (o1.sum() + o2.sum() + o3.sum()).item()                      # (.item(): a read on the current stream, no device-wide sync before it)
    ref_tot = (want[0][0].sum() + want[2][0].sum() + want[1][0].sum()).item()
    assert abs(tot - ref_tot) <= 1e-9 * abs(ref_tot)
    g.close()


@pytest.mark.parametrize("N", [400_000, 1_000_000])
def test_graphs_of_several_hundred_thousand_nodes(N):
    """The vicinity kernels keep bitmaps of N bits in LDS.  400 000 nodes: the extraction asks for more than 64 KB per workgroup
    (80 KB: the opt-in for large dynamic LDS) and its rows are the oracle's.  A million nodes do not fit the 160 KB of a CU:
    tlc_graph_create says so (TLC_ERR_UNSUPPORTED) instead of a failed launch later."""
    import torch
    from tlc_gnn_amd import engine, synth, _lib
    from oracle import oracle
    i = np.arange(N - 1, dtype=np.int64)
    ring = np.stack([i, i + 1], 1)
    j = np.arange(0, N - 7, 3, dtype=np.int64)
    chords = np.stack([j, j + 7], 1)
    e = np.concatenate([ring, chords])
    rs = np.random.RandomState(5)
    rowptr, col, w = synth.edges_to_csr(N, e, rs.uniform(-0.5, 0.9, size=len(e)))
    if N > 500_000:
        with pytest.raises(_lib.TlcError, match="UNSUPPORTED"):
            engine.DeviceGraph(rowptr, col, w)
        return
    g = engine.DeviceGraph(rowptr, col, w)
    pairs = e[rs.permutation(len(e))[:96]].astype(np.int32)
    pairs = np.concatenate([pairs, [[0, 1], [N - 2, N - 1], [5, 5], [10, N // 2]]]).astype(np.int32)
    for hop in (2, 1):
        out, st = g.pd_pi_batch(torch.as_tensor(pairs).cuda(), hop)
        out, st = out.cpu().numpy(), st.cpu().numpy()
        ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, hop, n_threads=0)
        assert np.array_equal(st, rst)
        assert np.array_equal(out == 0, ref == 0)
        nz = ref != 0
        assert rel_err(out[nz], ref[nz]).max() < 1e-8
    g.close()


@pytest.mark.parametrize("shape,scale,hop,n_pos", [("PubMed", 0.35, 2, 5000), ("PubMed", 0.2, 2, 900), ("Photo", 0.2, 1, 5000), ("Computers", 0.1, 1, 4500)])
def test_subgraph_list_extraction_equals_the_row_sweep(shape, scale, hop, n_pos):
    """Round 5: a pair whose smaller ball has <= 128 nodes takes its vicinity from that ball's SUBGRAPH LIST (x_sweep_ball) instead
    of sweeping the members' rows, in a launch of its own (tlc_extract_kernel<64, true>) when the batch has an early pass; the
    LARGE tier runs its divide and conquer in place.  Against the row sweep (ball_edges=0): status bytes, |S| and entry counts
    equal, images within 1e-12 (the entries of a vicinity come in another order); fast_split=0 (the same lists inside the general
    launch), dc_inplace=0 (tlc_pd_dc_kernel from the same record) and ball_bits=0 (round 6: that launch tests membership in the larger
    ball against the per-node ball bitmaps instead of a bitmap it marks in LDS) and plain_kernels=0 (round 6: the tier / swap kernels without
    the plain batch's parameters as compile-time constants): the SAME bits.  Batches above and below the early pass's
    minimum size, hop 1 and 2, sparse and dense graphs; and against the oracle."""
    import torch
    from tlc_gnn_amd import engine, synth
    from oracle import oracle
    n, edges, kappa, _, _ = synth.shaped_graph(shape, scale=scale)
    rowptr, col, w = synth.edges_to_csr(n, edges, kappa)
    rs = np.random.RandomState(11)
    pairs = _mixed_pairs(n, edges, rs, n_pos, n_pos // 4)
    pairs = np.concatenate([pairs, np.array([[-1, 3], [n, 0]], dtype=np.int32)])
    g = engine.DeviceGraph(rowptr, col, w)
    new = _run(g, torch, pairs, hop)
    sweep = _run(g, torch, pairs, hop, ball_edges=0)
    one = _run(g, torch, pairs, hop, fast_split=0)
    dck = _run(g, torch, pairs, hop, dc_inplace=0)
    nbb = _run(g, torch, pairs, hop, ball_bits=0)                 # (round 6) the larger ball marked in LDS instead of the ball bitmaps
    gen = _run(g, torch, pairs, hop, plain_kernels=0)             # (round 6) the general tier / swap kernel instances
    g.close()
    for other, name, bits in ((sweep, "ball_edges=0", False), (one, "fast_split=0", True), (dck, "dc_inplace=0", True), (nbb, "ball_bits=0", True),
                              (gen, "plain_kernels=0", True)):
        assert np.array_equal(new[1], other[1]), name
        assert np.array_equal(new[2], other[2]), name
        assert np.array_equal(new[3], other[3]), name
        if bits:
            assert np.array_equal(new[0], other[0]), name
        else:
            assert np.abs(new[0] - other[0]).max() <= 1e-12 * max(1.0, np.abs(other[0]).max()), name
    ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, hop, n_threads=0)
    assert np.array_equal(new[1], rst)
    nz = ref != 0
    assert np.array_equal(new[0] == 0, ref == 0) and rel_err(new[0][nz], ref[nz]).max() < 1e-8


def test_pairs_at_the_early_candidate_boundary_have_one_owner():
    """Round 5: with subgraph lists up to 512-node balls, a pair whose smaller ball has 511 or 512 nodes is served by a list AND is a
    candidate of the early pass (smaller ball >= 511).  The subgraph-list launch runs beside the classification, so it must leave every
    such pair alone (TlcVicParams::early_min_ball): before, both launches extracted it and wrote its header (equal sizes, two arena copies
    -- rows came out right whichever copy the tier kernel read, but a pair has one owner).  Hubs whose hop-1 balls have 509 .. 514 nodes, each paired with a larger hub, in batches large enough for
    the early pass, stream-ordered and pipelined, against the breadth-first kernels and the oracle."""
    import torch
    from tlc_gnn_amd import engine, synth
    from oracle import oracle
    rs = np.random.RandomState(21)
    n_leaf, sizes, copies = 700, (509, 510, 511, 512, 513, 514), 6
    edges, pairs, base = [], [], 0
    for c in range(copies):
        H = base
        hubs = [base + 1 + k for k in range(len(sizes))]
        leaves = np.arange(base + 1 + len(sizes), base + 1 + len(sizes) + n_leaf)
        edges += [[H, int(x)] for x in leaves] + [[H, h] for h in hubs]
        for h, sz in zip(hubs, sizes):
            edges += [[h, int(x)] for x in leaves[:sz - 2]]            # ball_1(h) = h + H + (sz - 2) leaves = sz nodes
            pairs.append([h, H])
        pairs += [[int(x), H] for x in leaves]
        base += 1 + len(sizes) + n_leaf
    e = np.array(edges, dtype=np.int64)
    rowptr, col, w = synth.edges_to_csr(base, e, rs.uniform(-0.5, 0.9, size=len(e)))
    pairs = np.array(pairs, dtype=np.int32)[rs.permutation(len(pairs))]
    assert len(pairs) >= 4096
    deg = np.diff(rowptr)
    assert sorted(set((deg[pairs[:, 0]] + 1).tolist()) & set(sizes)) == list(sizes)
    g = engine.DeviceGraph(rowptr, col, w)
    dev = torch.as_tensor(pairs).cuda()
    want, want_st = g.pd_pi_batch(dev, 1)
    ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, 1, n_threads=0)
    assert np.array_equal(want_st.cpu().numpy(), rst)
    nz = ref != 0
    assert np.array_equal(want.cpu().numpy() == 0, ref == 0) and rel_err(want.cpu().numpy()[nz], ref[nz]).max() < 1e-8
    outs = [torch.empty_like(want) for _ in range(3)]
    sts = [torch.empty_like(want_st) for _ in range(3)]
    for rnd in range(4):
        for k in range(3):
            g.pd_pi_batch(dev, 1, out=outs[k], status=sts[k], async_=True)
        g.join()
        torch.cuda.synchronize()
        for k in range(3):
            assert torch.equal(outs[k], want) and torch.equal(sts[k], want_st), (rnd, k)
    g.set_option("extract", 0)
    old, old_st = g.pd_pi_batch(dev, 1)
    assert torch.equal(old_st, want_st) and float((old - want).abs().max()) <= 1e-12
    g.close()

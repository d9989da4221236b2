"""GPU: the tier plumbing of the PD/PI batch at its thresholds -- which kernel a vicinity takes must never change its row.

  * the divide-and-conquer cycle swap (tlc_pd_dc_kernel, csrc/ext1_dc.h): LARGE-tier vicinities with 159 / 160 / 161 Pos edges
    (TLC_DC_MIN_POS = 160), with long runs of equal keys (not attempted), and with every solve forced to fail (the give-back
    to the serial walk) -- asserted through tlc_debug_dc_stats, rows against the oracle;
  * the MEDIUM tier's cut by Pos-edge count (TLC_MH_MIN_POS = 120): 119 / 120 / 121; its compact / wide kernel configurations at
    384 | 385 nodes and 512 | 513 edges;
  * the lane-per-subgraph kernel (pd_tiny.hip): vicinities at 16 / 17 nodes and 24 / 25 edges, tied weights, on and off;
  * a seeded random sweep over graph families, weight styles, hops and flags (the former tests/aids/fuzz_parity.py).
Every vicinity here is built as "hub + leaves + chords": the pair (hub, leaf) sees the whole component at hop 2, so its node,
edge and Pos-edge (= m - n + 1) counts are what the test asks for."""
import numpy as np
import pytest

from helpers import rel_err

pytestmark = pytest.mark.gpu


def hub_component(n, m, rs, base=0):
    """a hub, n - 1 leaves and m - (n - 1) distinct chords among the leaves; ids from `base` on; returns edges int64[m, 2]"""
    assert m >= n - 1 and m <= (n - 1) + (n - 1) * (n - 2) // 2
    e = set((0, k) for k in range(1, n))
    while len(e) < m:
        a, b = rs.randint(1, n, size=2)
        if a != b:
            e.add((min(a, b), max(a, b)))
    return np.array(sorted(e), dtype=np.int64) + base


def _check(g, torch, rowptr, col, w, pairs, hop=2, tol=1e-8):
    from oracle import oracle
    out, st = g.pd_pi_batch(torch.as_tensor(np.ascontiguousarray(pairs, dtype=np.int32)).cuda(), hop)
    out, st = out.cpu().numpy(), st.cpu().numpy()
    uniq, inv = np.unique(pairs, axis=0, return_inverse=True)
    ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, uniq.astype(np.int32), hop, n_threads=0)
    ref, rst = ref[inv.ravel()], rst[inv.ravel()]
    assert np.array_equal(st, rst)
    assert np.array_equal(out == 0, ref == 0)
    nz = ref != 0
    if nz.any():
        assert rel_err(out[nz], ref[nz]).max() < tol
    return out, st


@pytest.mark.parametrize("k_pos,expect_dc", [(159, False), (160, True), (161, True), (700, True)])
@pytest.mark.parametrize("n_pairs", [8, 4608])
def test_divide_and_conquer_threshold(k_pos, expect_dc, n_pairs):
    """LARGE-tier vicinities (601 nodes) around TLC_DC_MIN_POS: below it the tier kernel keeps its serial walk, from it on every
    subgraph goes through tlc_pd_dc_kernel -- on the ordinary heavy path (8 pairs) and through the early pass (4 608 pairs)."""
    import torch
    from tlc_gnn_amd import engine, synth
    rs = np.random.RandomState(k_pos)
    n = 601
    e = hub_component(n, n - 1 + k_pos, rs)
    rowptr, col, w = synth.edges_to_csr(n, e, rs.uniform(-0.5, 0.9, size=len(e)))
    pairs = np.tile(np.array([[0, 1], [0, 2], [3, 0], [0, 7], [5, 0], [0, 11], [0, 13], [2, 0]]), (n_pairs // 8, 1))
    g = engine.DeviceGraph(rowptr, col, w)
    _check(g, torch, rowptr, col, w, pairs)
    ran, back = g.dc_stats()
    stats = g.stats()
    assert stats["tier_large"] == len(pairs)
    assert (ran, back) == ((len(pairs), 0) if expect_dc else (0, 0))
    g.close()


def test_divide_and_conquer_give_back_and_long_tie_runs():
    """(i) Every solve forced to fail (debug option dc_force_fail): tlc_pd_dc_kernel gives each subgraph back to the serial walk it
    carries -- same rows.  The real trigger, ranks that are no minimum-spanning-tree order because of keys that differ in the
    last bits only, was not reached by 40 constructed weight families (tools/explore_dc.py), so the path is driven this way.
    (ii) Unit weights: hundreds of equal keys in a row (> TLC_DC_MAX_TIE_RUN) -- the divide and conquer is not attempted."""
    import torch
    from tlc_gnn_amd import engine, synth
    rs = np.random.RandomState(3)
    n = 701
    e = hub_component(n, n - 1 + 400, rs)
    pairs = np.array([[0, k] for k in range(1, 25)])
    rowptr, col, w = synth.edges_to_csr(n, e, rs.uniform(-0.5, 0.9, size=len(e)))
    g = engine.DeviceGraph(rowptr, col, w)
    a, _ = _check(g, torch, rowptr, col, w, pairs)
    assert g.dc_stats() == (len(pairs), 0)
    g.set_option("dc_force_fail", 1)
    b, _ = _check(g, torch, rowptr, col, w, pairs)
    assert g.dc_stats() == (0, len(pairs))
    g.set_option("dc_force_fail", 0)
    assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()
    g.close()
    rowptr, col, w = synth.edges_to_csr(n, e, np.zeros(len(e)))
    g = engine.DeviceGraph(rowptr, col, w)
    _check(g, torch, rowptr, col, w, pairs)
    assert g.dc_stats() == (0, 0) and g.stats()["tier_large"] == len(pairs)
    g.close()


@pytest.mark.parametrize("k_pos,many", [(119, False), (120, True), (121, True)])
def test_medium_tier_cut_by_pos_edges(k_pos, many):
    """MEDIUM-sized vicinities (301 nodes) around TLC_MH_MIN_POS: from 120 Pos edges on they go to the list that is launched
    first; either way the same kernels, the same rows."""
    import torch
    from tlc_gnn_amd import engine, synth
    rs = np.random.RandomState(k_pos)
    n = 301
    e = hub_component(n, n - 1 + k_pos, rs)
    rowptr, col, w = synth.edges_to_csr(n, e, rs.uniform(-0.5, 0.9, size=len(e)))
    pairs = np.array([[0, k] for k in range(1, 41)] + [[k, 0] for k in range(41, 61)])
    g = engine.DeviceGraph(rowptr, col, w)
    _check(g, torch, rowptr, col, w, pairs)
    stats = g.stats()
    assert stats["tier_medium"] == len(pairs)
    assert stats["tier_medium_many_pos"] == (len(pairs) if many else 0)
    g.close()


@pytest.mark.parametrize("n,m", [(384, 512), (385, 512), (384, 513), (300, 513), (512, 1024), (450, 600)])
def test_medium_compact_and_wide_configurations(n, m):
    """The MEDIUM tier runs kernels sized for 384 nodes / 512 edges (26 KB of LDS, six workgroups per CU); one node or one edge
    more and a vicinity goes to a list of the 512 / 1024 configuration (the many-Pos list of a chunk on its own, a list of its own
    in pipelined chunks).  At the cut, either way the oracle's rows -- and in one batch with the other kind, in pipelined chunks
    and with the speculative slots exhausted."""
    import torch
    from tlc_gnn_amd import engine, synth
    rs = np.random.RandomState(n * 7 + m)
    other = (450, 640) if (n <= 384 and m <= 512) else (301, 420)
    comps, pairs, base = [], [], 0
    for nn, mm, cnt in ((n, m, 24), (other[0], other[1], 8)):
        comps.append(hub_component(nn, mm, rs, base))
        pairs += [[base, base + k] for k in range(1, cnt + 1)]
        base += nn
    e = np.concatenate(comps)
    rowptr, col, w = synth.edges_to_csr(base, e, rs.uniform(-0.5, 0.9, size=len(e)))
    pairs = np.array(pairs)[rs.permutation(len(pairs))]
    g = engine.DeviceGraph(rowptr, col, w)
    ref_out, ref_st = _check(g, torch, rowptr, col, w, pairs)
    tc = g.tier_counts()
    wide = 24 if (n > 384 or m > 512) else 8
    many = 24 if (n <= 384 and m <= 512 and m - n + 1 >= 120) else (8 if (other[0] <= 384 and other[1] - other[0] + 1 >= 120) else 0)
    # a chunk on its own: the wide ones go with the many-Pos ones (the first, speculative launch; the wide kernels)
    assert tc["medium_wide"] == 0 and tc["medium_many_pos"] == wide + many and tc["medium"] == 32 - wide - many, tc
    assert g.stats()["tier_medium"] == 32
    g.set_option("spec_cap", 4)                                  # most of both lists beyond their reserved hand-off slots
    for _ in range(2):
        out, st = _check(g, torch, rowptr, col, w, pairs)
        assert np.array_equal(out, ref_out) and np.array_equal(st, ref_st)
    g.set_option("spec_cap", 0)
    # pipelined chunks (no split by Pos edges, no speculative launch): the same rows
    dev = torch.as_tensor(np.ascontiguousarray(pairs, dtype=np.int32)).cuda()
    outs = [torch.empty((len(pairs), 25), dtype=torch.float64, device="cuda") for _ in range(3)]
    sts = [torch.empty(len(pairs), dtype=torch.uint8, device="cuda") for _ in range(3)]
    for k in range(3):
        g.pd_pi_batch(dev, 2, out=outs[k], status=sts[k], async_=True)
    g.join()
    torch.cuda.synchronize()
    tc = g.tier_counts()                                         # (of the last call = the last chunk submitted)
    assert tc["medium_wide"] == wide and tc["medium"] == 32 - wide and tc["medium_many_pos"] == 0, tc
    for k in range(3):
        assert np.array_equal(outs[k].cpu().numpy(), ref_out) and np.array_equal(sts[k].cpu().numpy(), ref_st)
    g.close()


@pytest.mark.parametrize("n,k_pos", [(301, 319), (301, 320), (301, 400), (83, 600), (450, 560), (100, 350), (150, 330)])
def test_divide_and_conquer_of_the_wide_medium_configuration(n, k_pos):
    """MEDIUM-sized vicinities with hundreds of Pos edges (the dense hop-1 vicinities of the Amazon shapes: 83 nodes / 680 edges)
    sit in the 512 / 1 024 configuration for their edge count; from TLC_DC_MIN_POS_SHARED = 320 Pos edges the scan counts them and
    tlc_pd_dc_kernel goes between their tier and swap kernels -- in the chain of a lone chunk from the second call on (the
    speculative launch goes by the previous chunk's count), in pipelined chunks at once.  Rows: the oracle's, and equal to the
    serial walk's (option dc_force_fail = 1: every solve given back to the serial walk the kernel carries)."""
    import torch
    from tlc_gnn_amd import engine, synth
    rs = np.random.RandomState(n + k_pos)
    m = n - 1 + k_pos
    comps = [hub_component(n, m, rs, 0), hub_component(200, 260, rs, n)]                 # (and MEDIUM vicinities beside them)
    e = np.concatenate(comps)
    rowptr, col, w = synth.edges_to_csr(n + 200, e, rs.uniform(-0.5, 0.9, size=len(e)))
    pairs = np.array([[0, k] for k in range(1, 25)] + [[n, n + k] for k in range(1, 9)])
    g = engine.DeviceGraph(rowptr, col, w)
    g.set_option("dc_force_fail", 1)
    _check(g, torch, rowptr, col, w, pairs)                            # (first call: no previous chunk's count to size the launch by)
    ref_out, ref_st = _check(g, torch, rowptr, col, w, pairs)
    assert g.dc_stats()[0] == 0
    g.set_option("dc_force_fail", 0)
    expect = 24 if k_pos >= 320 else 0
    for _ in range(2):
        out, st = _check(g, torch, rowptr, col, w, pairs)
        assert np.array_equal(st, ref_st) and np.abs(out - ref_out).max() <= 1e-12 * np.abs(ref_out).max()
        assert g.dc_stats() == (expect, 0), g.dc_stats()
    dev = torch.as_tensor(np.ascontiguousarray(pairs, dtype=np.int32)).cuda()
    outs = [torch.empty((len(pairs), 25), dtype=torch.float64, device="cuda") for _ in range(3)]
    sts = [torch.empty(len(pairs), dtype=torch.uint8, device="cuda") for _ in range(3)]
    for k in range(3):
        g.pd_pi_batch(dev, 2, out=outs[k], status=sts[k], async_=True)
    g.join()
    torch.cuda.synchronize()
    for k in range(3):
        assert np.array_equal(sts[k].cpu().numpy(), ref_st) and np.abs(outs[k].cpu().numpy() - ref_out).max() <= 1e-12 * np.abs(ref_out).max()
        # (100, 350), (150, 330): small enough for the compact kernels, which do not mark -- routed to the wide ones by their Pos edges,
        # alone and pipelined alike: the same row, bit for bit
        assert np.array_equal(outs[k].cpu().numpy(), out), np.abs(outs[k].cpu().numpy() - out).max()
    g.close()


def test_tier_mask_leaves_exactly_the_masked_tiers_rows_unwritten():
    """Option tier_mask (the cost tables of profiles/: a tier's kernels are not launched): the rows and status bytes of the masked
    tier's pairs keep what the output buffers held, every other pair's row is the unmasked call's, bit for bit -- stream-ordered
    and pipelined.  (The TINY list goes to its kernel in size classes, largest first: every pair's row is its own whatever the order.)"""
    import torch
    from tlc_gnn_amd import engine, synth
    n, edges, kappa, hop, _ = synth.shaped_graph("PubMed", scale=0.3)
    rowptr, col, w = synth.edges_to_csr(n, edges, kappa)
    rs = np.random.RandomState(12)
    pairs = torch.as_tensor(edges[rs.permutation(len(edges))[:9000]].astype(np.int32)).cuda()
    g = engine.DeviceGraph(rowptr, col, w)
    ref, rst = g.pd_pi_batch(pairs, hop)
    assert g.stats()["tier_tiny"] > 3000
    nn, m2 = g.sizes(len(pairs))
    tier = engine.tier_of(nn, m2)
    names = {"pd_tier_tiny": 5, "pd_tier_small": 0, "pd_tier_mid": 4}
    for name, bit in names.items():
        sel = torch.as_tensor(tier == name).cuda() & (rst == 0)
        assert int(sel.sum()) > 50, name
        g.set_option("tier_mask", 255 & ~(1 << bit))
        out = torch.full_like(ref, -7.0)
        st = torch.full_like(rst, 99)
        g.pd_pi_batch(pairs, hop, out=out, status=st)
        torch.cuda.synchronize()
        assert bool((out[sel] == -7.0).all()) and bool((st[sel] == 99).all()), name
        assert torch.equal(out[~sel], ref[~sel]) and torch.equal(st[~sel], rst[~sel]), name
        outs = [torch.full_like(ref, -7.0) for _ in range(3)]
        sts = [torch.full_like(rst, 99) for _ in range(3)]
        for k in range(3):
            g.pd_pi_batch(pairs, hop, out=outs[k], status=sts[k], async_=True)
        g.join()
        torch.cuda.synchronize()
        for k in range(3):
            assert torch.equal(outs[k], out) and torch.equal(sts[k], st), (name, k)
        g.set_option("tier_mask", 255)
    g.close()


def test_speculative_launches_beyond_their_reserved_slots():
    """The MEDIUM-many-Pos tier kernel is submitted before the tier sizes are known, one workgroup per slot reserved from the
    previous chunk's count; list positions beyond the slots are completed by a second launch, and MEDIUM / MID vicinities beyond
    theirs run their cycle swap inside the tier kernel.  `spec_cap` = 8 slots puts most of this batch on those paths: same rows,
    bit for bit, as with the slots of the default sizing -- and the oracle's."""
    import torch
    from tlc_gnn_amd import engine, synth
    rs = np.random.RandomState(77)
    comps, pairs, base = [], [], 0
    for n, k_pos in ((301, 130), (301, 50), (100, 30)):          # MEDIUM with many Pos edges, MEDIUM, MID
        comps.append(hub_component(n, n - 1 + k_pos, rs, base))
        pairs += [[base, base + k] for k in range(1, 31)]
        base += n
    e = np.concatenate(comps)
    rowptr, col, w = synth.edges_to_csr(base, e, rs.uniform(-0.5, 0.9, size=len(e)))
    pairs = np.array(pairs)[rs.permutation(len(pairs))]
    g = engine.DeviceGraph(rowptr, col, w)
    ref_out, ref_st = _check(g, torch, rowptr, col, w, pairs)
    stats = g.stats()
    assert stats["tier_medium_many_pos"] == 30 and stats["tier_medium"] == 60 and stats["tier_mid"] == 30
    g.set_option("spec_cap", 8)
    for _ in range(2):                                           # (twice: the second chunk sizes its launch from the first)
        out, st = _check(g, torch, rowptr, col, w, pairs)
        assert np.array_equal(out, ref_out) and np.array_equal(st, ref_st)
    g.close()


@pytest.mark.parametrize("decimals", [None, 1])
def test_tiny_tier_boundaries_on_and_off(decimals):
    """Components at the limits of the lane-per-subgraph kernel (16 nodes / 24 edges) and just beyond, and the smallest ones:
    the kernel takes exactly the vicinities within both limits; switched off, the wavefront-per-subgraph kernel gives the same
    rows (the two break ties between equal keys differently: 1e-12, and 1e-8 against the oracle).  decimals=1: weights with
    one decimal, i.e. many tied paths and keys."""
    import torch
    from tlc_gnn_amd import engine, synth
    rs = np.random.RandomState(5)
    shapes = [(16, 24), (17, 24), (16, 25), (17, 25), (16, 15), (15, 24), (12, 16), (3, 3), (3, 2), (2, 1), (4, 6), (9, 20), (16, 23), (10, 9)]
    shapes = shapes * 6
    edges, pairs, base, expect_tiny = [], [], 0, 0
    for (n, m) in shapes:
        edges.append(hub_component(n, m, rs, base))
        pairs.append([base, base + 1 + rs.randint(n - 1)])
        if n >= 3:
            pairs.append([base + 1, base])
            expect_tiny += 1 if (n <= 16 and m <= 24) else 0
        expect_tiny += 1 if (n <= 16 and m <= 24) else 0
        base += n
    e = np.concatenate(edges)
    kappa = rs.uniform(-0.5, 0.9, size=len(e))
    if decimals is not None:
        kappa = np.round(kappa, decimals)
    rowptr, col, w = synth.edges_to_csr(base, e, kappa)
    pairs = np.array(pairs)
    g = engine.DeviceGraph(rowptr, col, w)
    on, st_on = _check(g, torch, rowptr, col, w, pairs)
    assert g.stats()["tier_tiny"] == expect_tiny
    g.set_option("tiny", 0)
    off, st_off = _check(g, torch, rowptr, col, w, pairs)
    assert g.stats()["tier_tiny"] == 0
    g.set_option("tiny", 1)
    assert np.array_equal(st_on, st_off)
    assert np.abs(on - off).max() <= 1e-12 * max(1.0, np.abs(off).max())
    g.close()


def _family(kind, n, rs):
    from tlc_gnn_amd import synth
    if kind == "er":
        e = rs.randint(0, n, size=(int(n * rs.uniform(1.0, 6.0)), 2))
    elif kind == "ba":
        return synth.holme_kim_edges(n, int(n * rs.uniform(1.5, 5.0)), triad_p=rs.uniform(0, 0.8), seed=int(rs.randint(1 << 30)))
    elif kind == "grid":
        w = int(np.sqrt(n)); idx = np.arange(w * w).reshape(w, w)
        e = np.concatenate([np.stack([idx[:, :-1].ravel(), idx[:, 1:].ravel()], 1), np.stack([idx[:-1].ravel(), idx[1:].ravel()], 1),
                            rs.randint(0, w * w, size=(w, 2))])
    elif kind == "caveman":
        k = 12; c = n // k
        e = np.array([(q * k + i, q * k + j) for q in range(c) for i in range(k) for j in range(i + 1, k) if rs.rand() < 0.7] +
                     [(q * k, ((q + 1) % c) * k + 1) for q in range(c)])
    else:                                                   # star: a few hubs
        hubs = rs.randint(0, n, size=5)
        e = np.concatenate([np.stack([rs.choice(hubs, size=3 * n), rs.randint(0, n, size=3 * n)], 1), rs.randint(0, n, size=(n, 2))])
    e = e[e[:, 0] != e[:, 1]]
    return np.unique(np.sort(e, 1), axis=0).astype(np.int64)


@pytest.mark.parametrize("seed", range(10))
def test_seeded_random_sweep_vs_oracle(seed):
    """graph family x weight style (continuous / one decimal / unweighted) x hop 1-3 x flags (0, NO_EXT1): status bytes equal,
    images within 1e-8 of the oracle; batches of > 4 096 pairs where the graph has them (early pass, bins, divide and conquer)."""
    import torch
    from tlc_gnn_amd import engine, synth, _lib
    from oracle import oracle
    rs = np.random.RandomState(1000 + seed)
    kind = ["er", "ba", "grid", "caveman", "star"][seed % 5]
    n = int(rs.choice([150, 600, 2500]))
    e = _family(kind, n, rs)
    n = int(e.max()) + 1
    kappa = rs.uniform(-0.5, 0.9, size=len(e))
    if seed % 3 == 1:
        kappa = np.round(kappa, 1)
    if seed % 3 == 2:
        kappa = np.zeros(len(e))
    rowptr, col, w = synth.edges_to_csr(n, e, kappa)
    g = engine.DeviceGraph(rowptr, col, w)
    for hop in (1, 2, 3):
        if hop == 3 and n > 700:
            continue
        pos = e[rs.permutation(len(e))[:4500]]
        pairs = np.concatenate([pos, rs.randint(0, n, size=(300, 2)), [[0, 0], [n + 5, 1], [-1, 2]]]).astype(np.int32)
        for flags in (0, _lib.NO_EXT1):
            got, st = g.pd_pi_batch(torch.as_tensor(pairs).cuda(), hop, flags=flags)
            got, st = got.cpu().numpy(), st.cpu().numpy()
            ref, rst, _ = oracle.pd_pi_batch(rowptr, col, w, pairs, hop, flags=flags, n_threads=0)
            assert np.array_equal(st, rst), (kind, hop, flags)
            scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-300
            assert (np.abs(got - ref) / scale).max() < 1e-8, (kind, hop, flags)
    g.close()

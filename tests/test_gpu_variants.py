"""GPU: the variant flags of the vicinity / filtration stage against goldens generated from the imported reference
(tests/golden/make_golden.py: G8 variants.npz, G4c kd_nc.npz).

  descriptor 'min' / 'max' and norm=False        sg2dgm/riccidist2dgm.py:20-61,310-329
  node-centred, single-root vicinity (KD-NC)     Knowledge_Distillation/data_utils_NC.py:27-50,95-187
Filtration values and diagrams bit-exact, status bytes = the reference's exception classes, images <= 1e-8 relative
(north_star: 1e-5)."""
import os

import numpy as np
import pytest

from helpers import same_multiset, ragged_slice, rel_err, csr_from_golden

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
EXC = {1: KeyError, 2: AssertionError, 3: ZeroDivisionError, 4: IndexError}


def _ricci(edges, kappa):
    return sorted([[int(a), int(b), float(k)] for (a, b), k in zip(edges.tolist(), kappa.tolist())] +
                  [[int(b), int(a), float(k)] for (a, b), k in zip(edges.tolist(), kappa.tolist())])


@pytest.mark.parametrize("hop", [1, 2])
@pytest.mark.parametrize("desc", ["min", "max"])
def test_descriptor_min_max_batch_path_golden(desc, hop):
    import torch
    from tlc_gnn_amd import engine, _lib
    d, v = np.load(os.path.join(G, "e2e.npz")), np.load(os.path.join(G, "variants.npz"))
    g = engine.DeviceGraph(*csr_from_golden(d))
    tag = "%s_norm_hop%d" % (desc, hop)
    pairs = torch.from_numpy(v["pairs"].astype(np.int32)).cuda()
    flag = _lib.DESCRIPTOR_FLAG[desc]
    out, st = g.pd_pi_batch(pairs, hop, flags=flag)
    out, st = out.cpu().numpy(), st.cpu().numpy()
    ref = v["pi_" + tag]
    assert np.array_equal(st.astype(np.int64), v["cls_" + tag])              # the reference's exception classes
    assert np.array_equal(out == 0, ref == 0)
    nz = ref != 0
    assert nz.any() and rel_err(out[nz], ref[nz]).max() < 1e-8
    # the node values themselves, bit for bit
    sel = v["fsel_" + tag]
    offs, ids, f, n, _ = [t.cpu().numpy() for t in g.vicinity_filtration(pairs[torch.from_numpy(sel).cuda()], hop, flags=flag)]
    for k in range(len(sel)):
        ref_ids, ref_f = ragged_slice(v["ids_" + tag], v["offs_" + tag], k), ragged_slice(v["f_" + tag], v["offs_" + tag], k)
        assert n[k] == len(ref_ids) and np.array_equal(ids[offs[k]:offs[k] + n[k]], ref_ids)
        assert np.array_equal(f[offs[k]:offs[k] + n[k]], ref_f), (tag, k)
    # the fused image stage assumes [0, 1]: raw values are refused there, never silently rasterised
    with pytest.raises(_lib.TlcError):
        g.pd_pi_batch(pairs[:4], hop, flags=flag | _lib.NO_NORM)
    g.close()


@pytest.mark.parametrize("hop", [1, 2])
@pytest.mark.parametrize("desc", ["sum", "min", "max"])
def test_norm_false_filtration_and_dropin_golden(desc, hop):
    """norm=False (sg2dgm_accelerate's own default): raw distances bit-exact through TLC_NO_NORM; images and exception classes
    through the drop-in, which chains tlc_vicinity_filtration -> tlc_pd_from_filtration -> tlc_pi_raster."""
    import torch
    from tlc_gnn_amd import engine, _lib
    from tlc_gnn_amd.sg2dgm import riccidist2dgm as sg2dgm
    from test_gpu_dropins import TinyGraph
    d, v = np.load(os.path.join(G, "e2e.npz")), np.load(os.path.join(G, "variants.npz"))
    tag = "%s_raw_hop%d" % (desc, hop)
    g = engine.DeviceGraph(*csr_from_golden(d))
    sel = v["fsel_" + tag]
    pairs = torch.from_numpy(v["pairs"][sel].astype(np.int32)).cuda()
    offs, ids, f, n, _ = [t.cpu().numpy() for t in g.vicinity_filtration(pairs, hop, flags=_lib.DESCRIPTOR_FLAG[desc] | _lib.NO_NORM)]
    for k in range(len(sel)):
        ref_ids, ref_f = ragged_slice(v["ids_" + tag], v["offs_" + tag], k), ragged_slice(v["f_" + tag], v["offs_" + tag], k)
        assert n[k] == len(ref_ids) and np.array_equal(ids[offs[k]:offs[k] + n[k]], ref_ids)
        assert np.array_equal(f[offs[k]:offs[k] + n[k]], ref_f), (tag, k)
    g.close()
    pi = sg2dgm.graph2pi(TinyGraph(int(d["n_nodes"]), d["edges"]), ricci_curv=_ricci(d["edges"], d["kappa"]))
    cls, ref = v["cls_" + tag], v["pi_" + tag]
    seen = set()
    step = 3 if hop == 1 else 4
    for i in list(range(0, len(cls), step)) + np.nonzero(cls == 1)[0][:6].tolist():
        u, w = v["pairs"][i].tolist()
        c = int(cls[i])
        seen.add(c)
        if c == 1 and (u not in pi.dict_node or w not in pi.dict_node):
            continue                                    # KeyError of the caller's dict_node lookup (:353), not of this function
        if c != 0:
            with pytest.raises(EXC[c]):
                pi.sg2dgm_accelerate(pi.dict_node[u], pi.dict_node[w], hop, extended_flag=True, descriptor=desc, norm=False)
            continue
        img = pi.sg2dgm_accelerate(pi.dict_node[u], pi.dict_node[w], hop, extended_flag=True, descriptor=desc, norm=False).reshape(-1)
        assert np.array_equal(img == 0, ref[i] == 0), (tag, i)
        nz = ref[i] != 0
        if nz.any():
            assert rel_err(img[nz], ref[i][nz]).max() < 1e-8, (tag, i)
    assert 0 in seen and 2 in seen


def test_graph2pi_default_descriptor_is_min_like_the_reference():
    from tlc_gnn_amd.sg2dgm import riccidist2dgm as sg2dgm
    from test_gpu_dropins import TinyGraph
    d, v = np.load(os.path.join(G, "e2e.npz")), np.load(os.path.join(G, "variants.npz"))
    pi = sg2dgm.graph2pi(TinyGraph(int(d["n_nodes"]), d["edges"]), ricci_curv=_ricci(d["edges"], d["kappa"]))
    pi.get_pimg_for_all_edges(v["pairs"].tolist(), cores=16, hop=2, norm=True, extended_flag=True, resolution=5)      # descriptor='min' (:362)
    ref = v["pi_min_norm_hop2"]
    assert pi.cnt_compute == int((v["cls_min_norm_hop2"] == 0).sum())
    assert np.array_equal(pi.pi_sg == 0, ref == 0) and rel_err(pi.pi_sg[ref != 0], ref[ref != 0]).max() < 1e-8
    with pytest.raises(KeyError):
        pi.get_pimg_for_all_edges(v["pairs"][:3].tolist(), cores=1, hop=2, descriptor='seal')     # no such node attribute


def test_kd_nc_node_centred_vicinities_golden_g4c():
    """data_utils_NC.py:95-187, filt='ricci': ball(u), ONE root, f = d(x, u) / (max + 1e-10): node sets, values (bit-exact),
    induced edges; then Ord0 / Ext1 (bit-exact multisets) and the three images."""
    from tlc_gnn_amd.Knowledge_Distillation import data_utils_NC as kd
    d, g5 = np.load(os.path.join(G, "kd_nc.npz")), np.load(os.path.join(G, "e2e.npz"))
    edges, ricci = g5["edges"], _ricci(g5["edges"], g5["kappa"])
    vic = kd.NodeVicinities(edges, ricci)
    for hop in (1, 2):
        sel = np.nonzero(d["hop"] == hop)[0]
        b = vic.batch(d["roots"][sel], hop)
        node_ptr, edge_ptr = b["node_ptr"].cpu().numpy(), b["edge_ptr"].cpu().numpy()
        ids, f, e = b["ids"].cpu().numpy(), b["f"].cpu().numpy(), b["edges"].cpu().numpy()
        for k, gi in enumerate(sel):
            ref_ids, ref_f = ragged_slice(d["ids"], d["offs"], gi), ragged_slice(d["f"], d["offs"], gi)
            my_ids = ids[node_ptr[k]:node_ptr[k + 1]]
            assert np.array_equal(my_ids, ref_ids), (hop, k)
            assert np.array_equal(f[node_ptr[k]:node_ptr[k + 1]], ref_f), (hop, k)          # bit-exact f
            got = np.sort(my_ids[e[edge_ptr[k]:edge_ptr[k + 1]]], axis=1)
            got = got[np.lexsort((got[:, 1], got[:, 0]))]
            assert np.array_equal(got, ragged_slice(d["edges"], d["e_offs"], gi))
    # the reference signature, mode 'PI': diagrams and images
    for gi in list(range(0, len(d["roots"]), 9)):
        u, hop = int(d["roots"][gi]), int(d["hop"][gi])
        o0, e1, img, fv, ei, pi0, pi1, _, _ = kd.compute_persistence_image(edges, u, filt='ricci', hop=hop, ricci_curv=ricci, mode='PI')
        assert same_multiset(o0, ragged_slice(d["ord0"], d["ord0_offs"], gi)), gi
        assert same_multiset(e1, ragged_slice(d["ext1"], d["ext1_offs"], gi)), gi
        for got, ref in ((img, d["pi"][gi]), (pi0, d["pi0"][gi]), (pi1, d["pi1"][gi])):
            assert np.array_equal(np.asarray(got) == 0, ref == 0)
            nz = ref != 0
            if nz.any():
                assert rel_err(np.asarray(got)[nz], ref[nz]).max() < 1e-8
        assert np.array_equal(np.array(fv), ragged_slice(d["f"], d["offs"], gi)) and tuple(ei.shape) == (2, len(ragged_slice(d["edges"], d["e_offs"], gi)))
    fv, ei = kd.compute_persistence_image(edges, int(d["roots"][0]), filt='ricci', hop=int(d["hop"][0]), ricci_curv=ricci, mode='filtration')
    assert np.array_equal(np.array(fv), ragged_slice(d["f"], d["offs"], 0))
    with pytest.raises(SystemExit):                                       # the reference prints and calls sys.exit() (:154-155)
        kd.compute_persistence_image(edges, 0, filt='no such filtration', ricci_curv=ricci)


def test_kd_structural_filtrations_golden_g4d():
    """data_utils_NC.py:124-135 (filt 'degree' / 'centrality' / 'clustering': the last two are the shipped training script's,
    train_Teacher_Model.py:158-159) and data_utils_LP.py:131-133 (filt 'degree'): the vicinities from the device, the node
    function from `structural_filtration` -- values bit-exact against the imported reference (G4d), then Ord0 / Ext1 as multisets
    and the three images; single calls through the reference signatures and one batched call per (kind, hop)."""
    from tlc_gnn_amd.Knowledge_Distillation import data_utils_NC as kd_nc, data_utils_LP as kd_lp
    d, g5 = np.load(os.path.join(G, "kd_struct.npz")), np.load(os.path.join(G, "e2e.npz"))
    edges = g5["edges"]
    names = ("degree", "centrality", "clustering", "degree")
    checked = 0
    for gi in range(0, len(d["kind"]), 3):
        kind, hop, u, v = int(d["kind"][gi]), int(d["hop"][gi]), int(d["u"][gi]), int(d["v"][gi])
        if kind < 3:
            res = kd_nc.compute_persistence_image(edges, u, filt=names[kind], hop=hop, mode='PI')
        else:
            res = kd_lp.compute_persistence_image(edges, u, v, filt='degree', hop=hop, mode='PI')
        o0, e1, img, fv, ei, pi0, pi1, _, _ = res
        assert np.array_equal(np.array(fv), ragged_slice(d["f"], d["offs"], gi)), (gi, kind)        # bit-exact f (ascending-id order)
        assert same_multiset(o0, ragged_slice(d["ord0"], d["ord0_offs"], gi)), gi
        assert same_multiset(e1, ragged_slice(d["ext1"], d["ext1_offs"], gi)), gi
        for got, ref in ((img, d["pi"][gi]), (pi0, d["pi0"][gi]), (pi1, d["pi1"][gi])):
            assert np.array_equal(np.asarray(got) == 0, ref == 0)
            nz = ref != 0
            if nz.any():
                assert rel_err(np.asarray(got)[nz], ref[nz]).max() < 1e-8
        checked += 1
    assert checked >= 50
    # batched: every case of a (kind, hop) in one call
    vic_n, vic_l = kd_nc.NodeVicinities(edges, None), kd_lp.Vicinities(edges, None)
    for kind in range(4):
        for hop in (1, 2):
            sel = np.nonzero((d["kind"] == kind) & (d["hop"] == hop))[0]
            if not len(sel):
                continue
            b = vic_n.batch(d["u"][sel], hop, filt=names[kind]) if kind < 3 else \
                vic_l.batch(np.stack([d["u"][sel], d["v"][sel]], 1), hop, filt='degree')
            node_ptr, ids, f = b["node_ptr"].cpu().numpy(), b["ids"].cpu().numpy(), b["f"].cpu().numpy()
            for k, gi in enumerate(sel):
                assert np.array_equal(ids[node_ptr[k]:node_ptr[k + 1]], ragged_slice(d["ids"], d["offs"], gi)), (kind, hop, k)
                assert np.array_equal(f[node_ptr[k]:node_ptr[k + 1]], ragged_slice(d["f"], d["offs"], gi)), (kind, hop, k)
    with pytest.raises(SystemExit):                                       # the reference prints and calls sys.exit() (:152-153)
        kd_lp.compute_persistence_image(edges, 0, 1, filt='no such filtration')


def test_kd_hks_filtration_golden_g4e():
    """filt='hks', the default of the three compute_persistence_image signatures (data_utils_GC.py:114-116, data_utils_NC.py:
    120-122, data_utils_LP.py:128-130): vicinities from the device, `hks_signature` on the host (scipy's eigh, like the reference),
    diagrams and images on the device.  Values bit-exact for whole graphs (same node order), 1e-11 for vicinities; images 1e-7
    (symmetric nodes tie up to rounding: which of them a pair lands on is noise on either side)."""
    from tlc_gnn_amd.Knowledge_Distillation import data_utils_NC as kd_nc, data_utils_LP as kd_lp, data_utils_GC as kd_gc
    d, g5 = np.load(os.path.join(G, "kd_hks.npz")), np.load(os.path.join(G, "e2e.npz"))
    edges = g5["edges"]
    n_done = [0, 0, 0]
    for gi in range(0, len(d["kind"]), 2):
        kind, hop, u, v, t = int(d["kind"][gi]), int(d["hop"][gi]), int(d["u"][gi]), int(d["v"][gi]), float(d["time"][gi])
        ref_f = ragged_slice(d["f"], d["offs"], gi)
        if kind == 2:
            res = kd_gc.compute_persistence_image((int(d["n"][gi]), ragged_slice(d["edges"], d["e_offs"], gi)), filt='hks', hks_time=t, mode='PI')
        elif kind == 0:
            res = kd_nc.compute_persistence_image(edges, u, filt='hks', hks_time=t, hop=hop, mode='PI')
        else:
            res = kd_lp.compute_persistence_image(edges, u, v, filt='hks', hks_time=t, hop=hop, mode='PI')
        o0, e1, img, fv, ei, pi0, pi1, _, _ = res
        if kind == 2:
            assert np.array_equal(np.array(fv), ref_f), gi
        else:
            assert np.abs(np.array(fv) - ref_f).max() <= 1e-11, gi
        assert len(o0) == len(ragged_slice(d["ord0"], d["ord0_offs"], gi)) and len(e1) == len(ragged_slice(d["ext1"], d["ext1_offs"], gi)), gi
        for got, ref in ((img, d["pi"][gi]), (pi0, d["pi0"][gi]), (pi1, d["pi1"][gi])):
            assert np.abs(np.asarray(got) - ref).max() <= 1e-7 * max(1.0, np.abs(ref).max()), (gi, kind)
        n_done[kind] += 1
    assert min(n_done) >= 8
    # batched GC call == single calls
    sel = [gi for gi in range(len(d["kind"])) if int(d["kind"][gi]) == 2 and float(d["time"][gi]) == 10.0][:12]
    graphs = [(int(d["n"][gi]), ragged_slice(d["edges"], d["e_offs"], gi)) for gi in sel]
    outs = kd_gc.compute_persistence_image_batch(graphs, filt='hks', hks_time=10.0)
    for gi, o in zip(sel, outs):
        assert np.array_equal(np.array(o[3]), ragged_slice(d["f"], d["offs"], gi))
        assert np.abs(np.asarray(o[2]) - d["pi"][gi]).max() <= 1e-7

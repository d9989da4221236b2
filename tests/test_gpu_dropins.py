"""GPU: the reference-named drop-in classes/functions (same signatures) end to end."""
import os

import numpy as np
import pytest

from helpers import same_multiset, ragged_slice, rel_err

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class _Nodes(dict):
    def __call__(self):
        return list(self.keys())


class TinyGraph:
    """the slice of the networkx API the reference functions touch: g.nodes(), g.nodes[n][key], g.edges()"""

    def __init__(self, n, edges, f=None, key="sum"):
        self.nodes = _Nodes({i: ({key: float(f[i])} if f is not None else {}) for i in range(n)})
        self._edges = [(int(a), int(b)) for a, b in edges]

    def edges(self):
        return list(self._edges)


def test_graph2pi_dropin_matches_reference_golden():
    from tlc_gnn_amd.sg2dgm import riccidist2dgm as sg2dgm
    d = np.load(os.path.join(G, "e2e.npz"))
    edges, kappa = d["edges"], d["kappa"]
    ricci = sorted([[int(a), int(b), float(k)] for (a, b), k in zip(edges.tolist(), kappa.tolist())] +
                   [[int(b), int(a), float(k)] for (a, b), k in zip(edges.tolist(), kappa.tolist())])
    pi = sg2dgm.graph2pi(TinyGraph(int(d["n_nodes"]), edges), ricci_curv=ricci)
    for hop in (1, 2):
        pi.get_pimg_for_all_edges(d["pairs"].tolist(), cores=16, hop=hop, norm=True, extended_flag=True, resolution=5,
                                  descriptor='sum')
        ref = d["pi_hop%d" % hop]
        assert pi.pi_sg.shape == ref.shape and pi.pi_sg.dtype == np.float64
        assert pi.cnt_compute == int((d["cls_hop%d" % hop] == 0).sum())
        assert np.array_equal(pi.pi_sg == 0, ref == 0)
        assert rel_err(pi.pi_sg[ref != 0], ref[ref != 0]).max() < 1e-8
    # the exception classes of sg2dgm_accelerate
    cls = d["cls_hop1"]
    names = {1: KeyError, 2: AssertionError, 3: ZeroDivisionError, 4: IndexError}
    seen = set()
    for i, c in enumerate(cls.tolist()):
        if c in names and c not in seen and c != 1:
            u, v = d["pairs"][i].tolist()
            with pytest.raises(names[c]):
                pi.sg2dgm_accelerate(pi.dict_node[u], pi.dict_node[v], 1, extended_flag=True, descriptor='sum', norm=True)
            seen.add(c)
    assert seen == {2, 3, 4}


def test_accelerated_pd_dropins():
    from tlc_gnn_amd.sg2dgm import accelerated_PD as apd
    from tlc_gnn_amd.Knowledge_Distillation import accelerated_PD as kd
    d = np.load(os.path.join(G, "pd_from_f.npz"))
    for g in (0, 7, 13, 101, 250):
        n = int(d["n"][g])
        edges = ragged_slice(d["edges"], d["e_offs"], g)
        f = ragged_slice(d["f"], d["f_offs"], g)
        gr = TinyGraph(n, edges, f, "sum")
        sf = apd.perturb_filter_function(gr, "sum")
        PD, pos, neg = apd.Union_find(sf)
        assert same_multiset(PD, ragged_slice(d["tlc_pd0"], d["tlc_pd0_offs"], g))
        assert len(pos) == d["npos"][g] and len(neg) == d["nneg"][g]
        PD1 = apd.Accelerate_PD(pos, neg, sf)
        assert same_multiset(PD1, ragged_slice(d["tlc_pd1"], d["tlc_pd1_offs"], g))
        sf = kd.perturb_filter_function(TinyGraph(n, edges), [float(v) for v in f])
        o0, e0, r1, pos, neg = kd.Union_find(sf)
        assert same_multiset(o0, ragged_slice(d["kd_ord0"], d["kd_ord0_offs"], g))
        assert same_multiset(r1, ragged_slice(d["kd_rel1"], d["kd_rel1_offs"], g))
        e1 = kd.Accelerate_PD(pos, neg, sf)
        assert same_multiset(e1, ragged_slice(d["kd_ext1"], d["kd_ext1_offs"], g))
        # the split is re-derived on the device: lists that are not Union_find's own are refused, never silently ignored
        if len(pos) >= 2:
            with pytest.raises(ValueError):
                kd.Accelerate_PD(pos[::-1], neg, sf)
            with pytest.raises(ValueError):
                apd.Accelerate_PD(pos[:-1], neg, sf)
            with pytest.raises(ValueError):
                apd.Accelerate_PD(pos, neg[:-1] + [pos[0]], sf)
        # an edge handed back with its endpoints swapped is the same edge
        assert same_multiset(kd.Accelerate_PD([[b, a] for a, b in pos], neg, sf), e1)


def test_persistence_imager_dropin():
    from tlc_gnn_amd.sg2dgm.PersistenceImager import PersistenceImager
    d = np.load(os.path.join(G, "pi_kat.npz"))
    im = PersistenceImager(resolution=5)
    assert np.array_equal(im._bpnts, d["bpnts"]) and np.array_equal(im._ppnts, d["ppnts"])
    out = im.transform(d["pd"])
    assert out.shape == (5, 5)
    assert np.abs(out.reshape(-1) - d["gt_4dp"]).max() < 6e-5
    with pytest.raises(NotImplementedError):
        PersistenceImager(resolution=5, birth_range=(0.0, 2.0))


def test_tlcgnn_call_harness():
    """call(): split, pair order, edge removal, hop rule, PI through the HIP path, model + data on the device."""
    import torch
    from tlc_gnn_amd import synth
    from tlc_gnn_amd.baselines import TLCGNN
    from tlc_gnn_amd.data import Data
    n, m, F_ = 260, 700, 40
    edges = synth.holme_kim_edges(n, m, triad_p=0.5, seed=21)
    ei = torch.from_numpy(np.concatenate([edges, edges[:, ::-1]]).T.copy()).long()
    data = Data(x=torch.from_numpy(synth.synthetic_features(n, F_, seed=2)), edge_index=ei, y=torch.zeros(n, dtype=torch.long))
    data.ricci_list = synth.synthetic_curvature(edges, seed=21)
    cwd = os.getcwd()
    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        os.chdir(tmp)
        try:
            model, data = TLCGNN.call(data, "Cora", F_, 2, 0)
        finally:
            os.chdir(cwd)
    n_val, n_test = int(m * 0.05), int(m * 0.1)
    assert (data.val_pos, data.test_pos, data.train_pos) == (n_val, n_test, m - n_val - n_test)
    assert data.edge_index.shape[1] == 2 * (m - n_val - n_test)
    assert model.PI.shape == (len(data.total_edges), 25)
    model.eval()
    with torch.no_grad():
        emb = model.encode(data)
        prob, y = model.decode(data, emb, "test")
    assert prob.shape[0] == data.test_pos + data.test_neg and torch.isfinite(prob).all()
    assert float(prob.min()) >= 0 and float(prob.max()) <= 1
    # the image rows of adjacent training pairs are mostly non-zero at hop 1 on a clustered graph
    assert (np.abs(model.PI[:data.train_pos]).sum(1) > 0).mean() > 0.3

#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by IMPORTING the reference (build container only).

Run from the repo root:  python tests/golden/make_golden.py [--time]

What it does: builds a throw-away shadow package in a temp directory that points at the
reference's own files under /root/reference (symlinks; `PersistenceImager.pyx` exposed as a `.py` module,
which is what the reference README recommends; `dionysus`/`gudhi`/... stubbed in `sys.modules` because
the accelerated path imports but never calls them), runs the reference single-threaded (`cores=1`, the
only deterministic semantics) on seeded inputs, and stores INPUTS + EXPECTED OUTPUTS as
small .npz files.  No reference source is copied into this repository; only data is written.

Fixtures:
  G1 pi_kat.npz          PI known-answer vector held (commented) in Knowledge_Distillation/pimg.py:451-501
  G2 pi_random.npz       random diagrams -> 5x5 images            (sg2dgm/PersistenceImager.pyx:352-388)
  G3 pd_from_f.npz       graphs + filtration -> PD pieces, both forks (sg2dgm|Knowledge_Distillation/accelerated_PD.py)
  G4 filtration.npz      (graph,u,v,hop) -> S, f[n]               (sg2dgm/riccidist2dgm.py:20-61,310-320)
  G5 e2e.npz             pairs -> pi_sg rows + exception class    (sg2dgm/riccidist2dgm.py:348-370)
  G4b kd_lp_filtration.npz  PDGNN LP vicinity: ids, f, induced edges  (Knowledge_Distillation/data_utils_LP.py:105-200)
  G6 kd_gc.npz           PDGNN ground-truth tuples, degree filtration (Knowledge_Distillation/data_utils_GC.py:98-166)
  G7 adj_split.npz       get_adj_split outputs, seed 1234         (loaddatas.py:38-54)
  G7b adj_split_ppi.npz  get_adj_split with the PPI configuration's proportions (0.2 / 0.2), three graphs (baselines/TLCGNN.py:73-75)
  G8 variants.npz        descriptor 'min' / 'max' and norm=False of sg2dgm_accelerate: f[n] + image rows + exception class
                         (sg2dgm/riccidist2dgm.py:20-61,310-329)
  G4e kd_hks.npz         filt='hks' (the signatures' default): data_utils_GC :114-116 (times 0.1 / 10), data_utils_NC :120-122, data_utils_LP
                         :128-130: values, Ord0 / Ext1, images
  G4d kd_struct.npz      PDGNN fork, structural filtrations: data_utils_NC filt 'degree' / 'centrality' / 'clustering' (:124-135),
                         data_utils_LP filt 'degree' (:131-133): values, Ord0 / Ext1, images
  G4c kd_nc.npz          PDGNN node-centred vicinity: ball(u), single root, f = d(x,u)/(max + 1e-10); Ord0 / Ext1 / images
                         (Knowledge_Distillation/data_utils_NC.py:27-50,95-187)
  G10 decode.npz         Net.decode('train' | 'val' | 'test') of the imported baselines/TLCGNN.py on seeded embeddings (rows with
                         norm above and below 1), images and weights: probabilities, labels, sampled negatives, the renormed
                         embedding (baselines/TLCGNN.py:27-62; GCNConv is stubbed -- decode never calls it)
"""
import argparse
import importlib
import os
import sys
import tempfile
import time
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference():
    """Shadow package: symlinks to the reference files, nothing is copied into the repo."""
    tmp = tempfile.mkdtemp(prefix="tlc_ref_shadow_")
    os.makedirs(os.path.join(tmp, "sg2dgm"))
    os.makedirs(os.path.join(tmp, "Knowledge_Distillation"))
    for pkg in ("sg2dgm", "Knowledge_Distillation"):
        open(os.path.join(tmp, pkg, "__init__.py"), "w").close()
    for f in ("accelerated_PD.py", "riccidist2dgm.py", "dgformat.py"):
        os.symlink(os.path.join(REF, "sg2dgm", f), os.path.join(tmp, "sg2dgm", f))
    os.symlink(os.path.join(REF, "sg2dgm", "PersistenceImager.pyx"),
               os.path.join(tmp, "sg2dgm", "PersistenceImager.py"))
    for f in ("accelerated_PD.py", "pimg.py", "data_utils_GC.py", "data_utils_LP.py", "data_utils_NC.py",
              "SBM_Model.py"):
        os.symlink(os.path.join(REF, "Knowledge_Distillation", f),
                   os.path.join(tmp, "Knowledge_Distillation", f))
    os.symlink(os.path.join(REF, "loaddatas.py"), os.path.join(tmp, "loaddatas.py"))
    sys.path.insert(0, tmp)
    sys.path.insert(0, os.path.join(tmp, "Knowledge_Distillation"))
    # imported-but-unused third-party modules
    _stub("dionysus")
    _stub("gudhi")
    _stub("sklearn_stub")
    _stub("learnable_filter")
    _stub("learnable_filter.loaddatas_LP")
    _stub("loaddatas_LP_arxiv", get_edges_split=None)
    _stub("spectral", SpectralClustering=None)
    _stub("Knowledge_Distillation.spectral", SpectralClustering=None)
    tg = _stub("torch_geometric")
    tgu = _stub("torch_geometric.utils", remove_self_loops=lambda ei, ea=None: (ei, ea))
    tgd = _stub("torch_geometric.datasets", TUDataset=None, ZINC=None, Planetoid=None, Amazon=None, PPI=None)
    tgdata = _stub("torch_geometric.data", Data=object)
    tgt = _stub("torch_geometric.transforms")
    tg.utils, tg.datasets, tg.data, tg.transforms = tgu, tgd, tgdata, tgt
    _stub("ogb")
    _stub("ogb.graphproppred", PygGraphPropPredDataset=None)
    mods = {}
    mods["apd"] = importlib.import_module("sg2dgm.accelerated_PD")
    mods["pimg"] = importlib.import_module("sg2dgm.PersistenceImager")
    mods["r2d"] = importlib.import_module("sg2dgm.riccidist2dgm")
    mods["kd_apd"] = importlib.import_module("Knowledge_Distillation.accelerated_PD")
    try:
        mods["kd_gc"] = importlib.import_module("Knowledge_Distillation.data_utils_GC")
    except Exception as e:  # pragma: no cover
        print("WARN: data_utils_GC not importable:", repr(e))
        mods["kd_gc"] = None
    try:
        mods["kd_lp"] = importlib.import_module("Knowledge_Distillation.data_utils_LP")
    except Exception as e:  # pragma: no cover
        print("WARN: data_utils_LP not importable:", repr(e))
        mods["kd_lp"] = None
    try:
        mods["kd_nc"] = importlib.import_module("Knowledge_Distillation.data_utils_NC")
    except Exception as e:  # pragma: no cover
        print("WARN: data_utils_NC not importable:", repr(e))
        mods["kd_nc"] = None
    try:
        mods["lds"] = importlib.import_module("loaddatas")
    except Exception as e:  # pragma: no cover
        print("WARN: loaddatas not importable:", repr(e))
        mods["lds"] = None
    return mods


# ----------------------------------------------------------------------------------------------- helpers
def ragged(list_of_arrays, width=2, dtype=np.float64):
    offs = np.zeros(len(list_of_arrays) + 1, dtype=np.int64)
    for i, a in enumerate(list_of_arrays):
        offs[i + 1] = offs[i] + len(a)
    if offs[-1] == 0:
        flat = np.zeros((0, width) if width else (0,), dtype=dtype)
    else:
        flat = np.concatenate([np.asarray(a, dtype=dtype).reshape((-1, width) if width else (-1,))
                               for a in list_of_arrays if len(a)])
    return flat, offs


def random_connected_graph(rs, n, m_extra):
    """random spanning tree + m_extra distinct extra edges; returns int64[m,2]."""
    perm = rs.permutation(n)
    es = set()
    for i in range(1, n):
        a = perm[i]
        b = perm[rs.randint(i)]
        es.add((min(a, b), max(a, b)))
    tries = 0
    while m_extra > 0 and tries < 20 * m_extra + 100:
        tries += 1
        a, b = rs.randint(n), rs.randint(n)
        if a == b:
            continue
        e = (min(a, b), max(a, b))
        if e in es:
            continue
        es.add(e)
        m_extra -= 1
    es = np.array(sorted(es), dtype=np.int64).reshape(-1, 2)
    # random orientation + order: the PD multisets must not depend on it
    flip = rs.rand(len(es)) < 0.5
    es[flip] = es[flip][:, ::-1]
    return es[rs.permutation(len(es))]


# ----------------------------------------------------------------------------------------------- G1/G2
PI_KAT_PD = [
    [0.0913, 0.0913], [0.1294, 0.1294], [0.1606, 0.1606], [0.1628, 0.1628], [0.0801, 0.1628], [0.1993, 0.1993],
    [0.1186, 0.1993], [0.1189, 0.1993], [0.2081, 0.2081], [0.1294, 0.2081], [0.0800, 0.2081], [0.3562, 0.3562],
    [0.0784, 0.3562], [0.0784, 0.3562], [0.0784, 0.3562], [0.0798, 0.3562], [1.0000, 1.0000], [0.0391, 1.0000],
    [0.0391, 1.0000], [0.0391, 1.0000], [0.0391, 1.0000], [0.0391, 1.0000], [0.0391, 1.0000], [0.0391, 1.0000],
    [0.0391, 1.0000], [0.0784, 1.0000], [0.0913, 1.0000], [0.0798, 0.2081], [0.0784, 0.3562], [0.0784, 0.3562],
    [0.0784, 0.3562]] + [[0.0391, 1.0000]] * 15
PI_KAT_GT = [0.1209, 0.1381, 0.1520, 0.1610, 0.1642, 0.1173, 0.1340, 0.1474, 0.1561, 0.1592, 0.1093, 0.1249,
             0.1374, 0.1455, 0.1483, 0.0979, 0.1119, 0.1230, 0.1303, 0.1328, 0.0843, 0.0963, 0.1059, 0.1121, 0.1143]


def make_g1_g2(mods):
    pimg = mods["pimg"]
    imager = pimg.PersistenceImager(resolution=5)
    pd = np.array(PI_KAT_PD, dtype=np.float64)
    assert pd.shape == (46, 2)
    ref = imager.transform(pd).reshape(-1)
    np.savez(os.path.join(HERE, "pi_kat.npz"), pd=pd, gt_4dp=np.array(PI_KAT_GT), ref_fp64=ref,
             bpnts=imager._bpnts, ppnts=imager._ppnts)
    print("G1 max|ref-gt| =", np.abs(ref - np.array(PI_KAT_GT)).max())

    rs = np.random.RandomState(20240101)
    dgms, outs, ress = [], [], []
    for i in range(200):
        k = int(rs.randint(1, 301))
        b = rs.uniform(-0.2, 1.2, size=k)
        kind = i % 4
        if kind == 0:
            d = b + rs.uniform(0, 1.0, size=k)
        elif kind == 1:
            d = b + rs.uniform(-0.5, 1.5, size=k)       # below-diagonal and pers>1 points
        elif kind == 2:
            d = b.copy()                                # all on the diagonal
            d[: k // 2] += rs.uniform(0, 0.3, size=k // 2)
        else:
            b = np.round(b * 3) / 3
            d = b + np.round(rs.uniform(0, 1, size=k) * 3) / 3
        dg = np.stack([b, d], axis=1)
        dgms.append(dg)
        outs.append(imager.transform(dg).reshape(-1))
    flat, offs = ragged(dgms)
    # other resolutions (API parity: `resolution` is a parameter of get_pimg_for_all_edges)
    res_outs = {}
    for res in (3, 7):
        im = pimg.PersistenceImager(resolution=res)
        res_outs[res] = np.stack([im.transform(dg).reshape(-1) for dg in dgms[:20]])
    np.savez_compressed(os.path.join(HERE, "pi_random.npz"), pts=flat, offs=offs, out=np.stack(outs),
                        out_res3=res_outs[3], out_res7=res_outs[7])
    print("G2 diagrams:", len(dgms), "points:", len(flat))


# ----------------------------------------------------------------------------------------------- G3
def run_tlc_pd(apd, n, edges, f):
    import networkx as nx
    g = nx.Graph()
    g.add_nodes_from(range(n))
    for i in range(n):
        g.nodes[i]["sum"] = float(f[i])
    g.add_edges_from([(int(a), int(b)) for a, b in edges])
    sf = apd.perturb_filter_function(g, "sum")
    pd0, pos, neg = apd.Union_find(sf)
    pd1 = apd.Accelerate_PD(pos, neg, sf)
    return np.array(pd0, dtype=np.float64).reshape(-1, 2), np.array(pd1, dtype=np.float64).reshape(-1, 2), \
        len(pos), len(neg)


def run_kd_pd(kd, n, edges, f):
    import networkx as nx
    g = nx.Graph()
    g.add_nodes_from(range(n))
    g.add_edges_from([(int(a), int(b)) for a, b in edges])
    sf = kd.perturb_filter_function(g, [float(x) for x in f])
    ord0, ext0, rel1, pos, neg = kd.Union_find(sf)
    ext1 = kd.Accelerate_PD(pos, neg, sf)
    return (np.asarray(ord0, dtype=np.float64).reshape(-1, 2), np.asarray(ext0, dtype=np.float64).reshape(-1, 2),
            np.asarray(rel1, dtype=np.float64).reshape(-1, 2), np.asarray(ext1, dtype=np.float64).reshape(-1, 2))


def make_g3(mods):
    rs = np.random.RandomState(777)
    ns, es, fs = [], [], []
    tlc0, tlc1, npos, nneg = [], [], [], []
    k_ord0, k_ext0, k_rel1, k_ext1 = [], [], [], []
    sizes = [2, 3, 4, 5] + [int(x) for x in rs.randint(2, 201, size=496)]
    for gi, n in enumerate(sizes):
        max_extra = max(0, min(3 * n, n * (n - 1) // 2 - (n - 1)))
        extra = int(rs.randint(0, max_extra + 1))
        if gi % 10 == 0:
            extra = 0                                   # trees: no Pos edges at all
        edges = random_connected_graph(rs, n, extra)
        kind = gi % 5
        if kind == 0:
            f = rs.uniform(0, 1, size=n)
        elif kind == 1:
            f = rs.randint(0, 4, size=n) / 3.0          # heavy ties
        elif kind == 2:
            f = rs.randint(0, 8, size=n) / 7.0
        elif kind == 3:
            f = rs.uniform(0, 1, size=n)
            f = f / f.max()
            f[rs.randint(n)] = 0.0                      # what build_fv produces: min 0, max 1
        else:
            f = np.full(n, 1.0)                         # constant (the d(u,v)>hop case)
        p0, p1, a, b = run_tlc_pd(mods["apd"], n, edges, f)
        o0, e0, r1, e1 = run_kd_pd(mods["kd_apd"], n, edges, f)
        ns.append(n)
        es.append(edges)
        fs.append(f)
        tlc0.append(p0)
        tlc1.append(p1)
        npos.append(a)
        nneg.append(b)
        k_ord0.append(o0)
        k_ext0.append(e0)
        k_rel1.append(r1)
        k_ext1.append(e1)
    e_flat, e_offs = ragged(es, 2, np.int64)
    f_flat, f_offs = ragged(fs, 0, np.float64)
    out = dict(n=np.array(ns), edges=e_flat, e_offs=e_offs, f=f_flat, f_offs=f_offs,
               npos=np.array(npos), nneg=np.array(nneg))
    for name, lst in (("tlc_pd0", tlc0), ("tlc_pd1", tlc1), ("kd_ord0", k_ord0), ("kd_ext0", k_ext0),
                      ("kd_rel1", k_rel1), ("kd_ext1", k_ext1)):
        flat, offs = ragged(lst)
        out[name] = flat
        out[name + "_offs"] = offs
    np.savez_compressed(os.path.join(HERE, "pd_from_f.npz"), **out)
    print("G3 graphs:", len(ns), "edges:", len(e_flat))


# ----------------------------------------------------------------------------------------------- G4/G5
EXC_CLASS = {"ok": 0, "KeyError": 1, "AssertionError": 2, "ZeroDivisionError": 3, "IndexError": 4}


def build_ref_graph2pi(mods, n_nodes, edges, kappa):
    import networkx as nx
    from tlc_gnn_amd import synth
    g = nx.Graph()
    g.add_edges_from([(int(a), int(b)) for a, b in edges])           # edges only: loaddatas.py:88-92
    ricci = []
    for (a, b), k in zip(edges.tolist(), kappa.tolist()):
        ricci.append([a, b, k])
        ricci.append([b, a, k])
    ricci = sorted(ricci)
    return mods["r2d"].graph2pi(g, ricci_curv=ricci)


def ref_one_pair(mods, pi, u, v, hop, want_f=False):
    """mirror of get_pimg_for_one_edge (riccidist2dgm.py:348-357) that reports the exception class."""
    import networkx as nx
    r2d = mods["r2d"]
    try:
        uu, vv = pi.dict_node[u], pi.dict_node[v]
        img = pi.sg2dgm_accelerate(uu, vv, hop, norm=True, extended_flag=True, resolution=5, descriptor="sum")
        return img.reshape(-1), 0
    except BaseException as e:  # noqa: the reference swallows everything, classify for the status byte
        return np.zeros(25), EXC_CLASS.get(type(e).__name__, 9)


def ref_filtration(mods, pi, u, v, hop):
    """S and f for one pair, straight from the reference classes (riccidist2dgm.py:311-320)."""
    import networkx as nx
    r2d = mods["r2d"]
    uu, vv = pi.dict_node[u], pi.dict_node[v]
    nodes_u = [uu] + [x for _, x in nx.bfs_edges(pi.graph, uu, depth_limit=hop)]
    nodes_v = [vv] + [x for _, x in nx.bfs_edges(pi.graph, vv, depth_limit=hop)]
    nodes = list(set(nodes_u) & set(nodes_v))
    sub = pi.graph.subgraph(nodes).copy()                # copy: isolate from the shared parent
    fil = r2d.filtration(sub, uu, vv, hop, ricci_curv=pi.ricci_curv)
    g = fil.build_fv(weight_graph=True, norm=True)
    inv = {new: old for old, new in pi.dict_node.items()}
    ids = np.array(sorted(inv[x] for x in g.nodes()), dtype=np.int64)
    back = {inv[x]: x for x in g.nodes()}
    f = np.array([g.nodes[back[i]]["sum"] for i in ids.tolist()], dtype=np.float64)
    return ids, f


def make_g4_g5(mods, do_time=False):
    from tlc_gnn_amd import synth
    rs = np.random.RandomState(4242)
    n_nodes, m_edges = 320, 900
    edges = synth.holme_kim_edges(n_nodes - 6, m_edges, triad_p=0.5, seed=99)   # last 6 ids stay isolated
    # add a pendant path and a far-away component so that every zero-row class occurs
    extra = np.array([[n_nodes - 6, n_nodes - 5], [n_nodes - 5, n_nodes - 4]], dtype=np.int64)   # small component
    edges = np.concatenate([edges, extra])
    kappa = synth.curvature_array(edges, seed=99)
    pi = build_ref_graph2pi(mods, n_nodes, edges, kappa)
    pos = edges[rs.permutation(len(edges))[:260]]
    neg = np.stack([rs.randint(0, n_nodes, size=200), rs.randint(0, n_nodes, size=200)], axis=1)
    special = np.array([[0, 0], [5, 5], [n_nodes - 1, 3], [3, n_nodes - 2], [n_nodes - 6, n_nodes - 4],
                        [n_nodes - 6, n_nodes - 5], [n_nodes - 5, n_nodes - 5], [n_nodes - 1, n_nodes - 1]],
                       dtype=np.int64)
    pairs = np.concatenate([pos, pos[:40, ::-1], neg, special]).astype(np.int64)
    out = {"n_nodes": n_nodes, "edges": edges, "kappa": kappa, "pairs": pairs}
    for hop in (1, 2, 3):
        rows, cls = [], []
        for u, v in pairs.tolist():
            r, c = ref_one_pair(mods, pi, u, v, hop)
            rows.append(r)
            cls.append(c)
        rows = np.stack(rows)
        cls = np.array(cls, dtype=np.int64)
        # the shipped entry point must agree with the per-pair mirror above when run single-threaded
        pi.get_pimg_for_all_edges(pairs.tolist(), cores=1, hop=hop, norm=True, extended_flag=True, resolution=5,
                                  descriptor="sum")
        assert np.array_equal(pi.pi_sg, rows), "mirror disagrees with get_pimg_for_all_edges"
        assert pi.cnt_compute == int((cls == 0).sum())
        out["pi_hop%d" % hop] = rows
        out["cls_hop%d" % hop] = cls
        print("G5 hop", hop, "classes:", np.bincount(cls, minlength=5), "nonzero rows:",
              int((np.abs(rows).sum(1) > 0).sum()))
    np.savez_compressed(os.path.join(HERE, "e2e.npz"), **out)

    # G4: filtration values for the pairs that get that far
    ids_l, f_l, pr, hops = [], [], [], []
    for hop in (1, 2, 3):
        cls = out["cls_hop%d" % hop]
        ok = np.nonzero((cls == 0) | (cls == 4))[0]
        for i in ok[:: 3 if hop > 1 else 1][:120]:
            u, v = pairs[i].tolist()
            try:
                ids, f = ref_filtration(mods, pi, u, v, hop)
            except BaseException:
                continue
            ids_l.append(ids)
            f_l.append(f)
            pr.append([u, v])
            hops.append(hop)
    ids_flat, offs = ragged(ids_l, 0, np.int64)
    f_flat, _ = ragged(f_l, 0, np.float64)
    np.savez_compressed(os.path.join(HERE, "filtration.npz"), n_nodes=n_nodes, edges=edges, kappa=kappa,
                        pairs=np.array(pr, dtype=np.int64), hop=np.array(hops), ids=ids_flat, f=f_flat, offs=offs)
    print("G4 cases:", len(pr))

    if do_time:
        n, e, k, hop, _ = synth.shaped_graph("PubMed")
        pi = build_ref_graph2pi(mods, n, e, k)
        sel = e[np.random.RandomState(1).permutation(len(e))[:300]]
        t0 = time.time()
        pi.get_pimg_for_all_edges(sel.tolist(), cores=1, hop=2, norm=True, extended_flag=True, resolution=5,
                                  descriptor="sum")
        dt = time.time() - t0
        print("TIMING reference python, PubMed-shaped, hop=2, 300 positive pairs, cores=1: %.2f s = %.1f PI/s"
              % (dt, 300 / dt))
        np.savez_compressed(os.path.join(HERE, "pubmed_sample.npz"), pairs=sel, pi=pi.pi_sg,
                            ref_seconds=dt)


# ----------------------------------------------------------------------------------------------- G4b
def make_g4b(mods):
    """PDGNN link-prediction vicinity (Knowledge_Distillation/data_utils_LP.py:105-200, filt='ricci', mode='filtration'):
    nodes = ball(u) & ball(v) + [u, v], NO connectivity assert (unreachable roots -> 100), normalised by max + 1e-10."""
    import networkx as nx
    kd = mods["kd_lp"]
    if kd is None:
        print("G4b skipped")
        return
    d = np.load(os.path.join(HERE, "e2e.npz"))
    edges, kappa, pairs = d["edges"], d["kappa"], d["pairs"]
    g = nx.Graph()
    g.add_edges_from([(int(a), int(b)) for a, b in edges])
    ricci = sorted([[int(a), int(b), float(k)] for (a, b), k in zip(edges.tolist(), kappa.tolist())] +
                   [[int(b), int(a), float(k)] for (a, b), k in zip(edges.tolist(), kappa.tolist())])
    ids_l, f_l, e_l, pr, hops, none = [], [], [], [], [], []
    for hop in (1, 2):
        for i, (u, v) in enumerate(pairs.tolist()):
            if u not in g or v not in g:
                continue                                        # networkx raises on a missing root: not a data case
            fv, ei = kd.compute_persistence_image(g, u, v, filt="ricci", hop=hop, ricci_curv=ricci, mode="filtration")
            if fv is None:
                none.append([u, v, hop])
                continue
            # the same two statements the function runs, to recover which node each value belongs to (:108-114)
            nodes_u = [u] + [x for _, x in nx.bfs_edges(g, u, depth_limit=hop)]
            nodes_v = [v] + [x for _, x in nx.bfs_edges(g, v, depth_limit=hop)]
            nodes = list(set(nodes_u) & set(nodes_v)) + [u] + [v]
            sub = nx.convert_node_labels_to_integers(g.subgraph(nodes), label_attribute="old_label")
            old = np.array([sub._node[k]["old_label"] for k in range(len(sub))], dtype=np.int64)
            order = np.argsort(old)
            ids_l.append(old[order])
            f_l.append(np.asarray(fv, dtype=np.float64)[order])
            ee = old[np.asarray(ei).T.reshape(-1, 2)]
            ee = np.sort(ee, axis=1)
            e_l.append(ee[np.lexsort((ee[:, 1], ee[:, 0]))])
            pr.append([u, v])
            hops.append(hop)
    ids_flat, offs = ragged(ids_l, 0, np.int64)
    f_flat, _ = ragged(f_l, 0, np.float64)
    e_flat, e_offs = ragged(e_l, 2, np.int64)
    np.savez_compressed(os.path.join(HERE, "kd_lp_filtration.npz"), pairs=np.array(pr, dtype=np.int64), hop=np.array(hops),
                        ids=ids_flat, f=f_flat, offs=offs, edges=e_flat, e_offs=e_offs, none_cases=np.array(none, dtype=np.int64))
    print("G4b cases:", len(pr), "none:", len(none), "with a 100-sentinel:", sum(1 for f in f_l if len(f) and f.max() > 0 and (np.isclose(f * 0 + 1, 1).all())))


# ----------------------------------------------------------------------------------------------- G6
def make_g6(mods):
    import networkx as nx
    kd = mods["kd_gc"]
    if kd is None:
        print("G6 skipped")
        return
    rs = np.random.RandomState(31337)
    ns, es, fs, ord0, ext1, pis, pi0s, pi1s = [], [], [], [], [], [], [], []
    for gi in range(60):
        n = int(max(3, rs.poisson(25)))
        edges = random_connected_graph(rs, n, int(rs.randint(0, 4)))
        g = nx.Graph()
        g.add_nodes_from(range(n))
        g.add_edges_from([(int(a), int(b)) for a, b in edges])
        res = kd.compute_persistence_image(g, filt="degree", mode="PI")
        d0, d1, img, fv, ei, pi0, pi1 = res[:7]
        ns.append(n)
        es.append(np.asarray(ei).T.astype(np.int64))
        fs.append(np.asarray(fv, dtype=np.float64))
        ord0.append(np.asarray(d0, dtype=np.float64).reshape(-1, 2))
        ext1.append(np.asarray(d1, dtype=np.float64).reshape(-1, 2))
        pis.append(np.asarray(img, dtype=np.float64).reshape(-1))
        pi0s.append(np.asarray(pi0, dtype=np.float64).reshape(-1))
        pi1s.append(np.asarray(pi1, dtype=np.float64).reshape(-1))
    e_flat, e_offs = ragged(es, 2, np.int64)
    f_flat, f_offs = ragged(fs, 0, np.float64)
    o_flat, o_offs = ragged(ord0)
    x_flat, x_offs = ragged(ext1)
    np.savez_compressed(os.path.join(HERE, "kd_gc.npz"), n=np.array(ns), edges=e_flat, e_offs=e_offs, f=f_flat,
                        f_offs=f_offs, ord0=o_flat, ord0_offs=o_offs, ext1=x_flat, ext1_offs=x_offs,
                        pi=np.stack(pis), pi0=np.stack(pi0s), pi1=np.stack(pi1s))
    print("G6 graphs:", len(ns))


# ----------------------------------------------------------------------------------------------- G7
def make_g7(mods):
    import scipy.sparse as sp
    lds = mods["lds"]
    if lds is None:
        print("G7 skipped")
        return
    from tlc_gnn_amd import synth
    n = 200
    edges = synth.holme_kim_edges(n, 520, triad_p=0.4, seed=5)
    a = sp.coo_matrix((np.ones(len(edges)), (edges[:, 0], edges[:, 1])), shape=(n, n))
    adj = sp.csr_matrix(a + a.T)
    parts = lds.get_adj_split(adj, val_prop=0.05, test_prop=0.1, seed=1234)
    names = ["train_edges", "train_edges_false", "val_edges", "val_edges_false", "test_edges", "test_edges_false"]
    np.savez_compressed(os.path.join(HERE, "adj_split.npz"), n_nodes=n, edges=edges,
                        **{k: np.asarray(v, dtype=np.int64) for k, v in zip(names, parts)})
    print("G7 sizes:", [len(p) for p in parts])


# ----------------------------------------------------------------------------------------------- G7b
def make_g7b(mods):
    """The PPI configuration's split (baselines/TLCGNN.py:73-75: val_prop = test_prop = 0.2) of get_adj_split, on three small
    PPI-shaped graphs (the reference loops over PPI's 20 graphs, pipelines.py:81-111), seed 1234 as everywhere."""
    import scipy.sparse as sp
    lds = mods["lds"]
    if lds is None:
        print("G7b skipped")
        return
    from tlc_gnn_amd import synth
    out = {}
    for gi, (n, m, seed) in enumerate(((260, 1900, 21), (300, 2600, 22), (220, 1500, 23))):
        edges = synth.holme_kim_edges(n, m, triad_p=0.5, seed=seed)
        a = sp.coo_matrix((np.ones(len(edges)), (edges[:, 0], edges[:, 1])), shape=(n, n))
        adj = sp.csr_matrix(a + a.T)
        parts = lds.get_adj_split(adj, val_prop=0.2, test_prop=0.2, seed=1234)
        names = ["train_edges", "train_edges_false", "val_edges", "val_edges_false", "test_edges", "test_edges_false"]
        out["g%d_n_nodes" % gi] = n
        out["g%d_edges" % gi] = edges
        for k, v in zip(names, parts):
            out["g%d_%s" % (gi, k)] = np.asarray(v, dtype=np.int64)
        print("G7b graph %d sizes:" % gi, [len(p) for p in parts])
    np.savez_compressed(os.path.join(HERE, "adj_split_ppi.npz"), n_graphs=3, **out)


# ----------------------------------------------------------------------------------------------- G8 / G4c
def make_g8(mods):
    """descriptor 'min' / 'max' (and 'sum') with norm=True, and norm=False, of sg2dgm_accelerate on the G5 graph: the node
    values build_fv leaves on the subgraph (riccidist2dgm.py:47-56), the image row and the exception class per pair."""
    import networkx as nx
    r2d = mods["r2d"]
    d = np.load(os.path.join(HERE, "e2e.npz"))
    n_nodes, edges, kappa, pairs = int(d["n_nodes"]), d["edges"], d["kappa"], d["pairs"]
    pi = build_ref_graph2pi(mods, n_nodes, edges, kappa)
    inv = {new: old for old, new in pi.dict_node.items()}
    out = {"pairs": pairs}
    for hop in (1, 2):
        for desc in ("min", "max", "sum"):
            for norm in (True, False):
                if desc == "sum" and norm:
                    continue                                     # that is G5
                rows, cls, ids_l, f_l, sel = [], [], [], [], []
                for i, (u, v) in enumerate(pairs.tolist()):
                    try:
                        uu, vv = pi.dict_node[u], pi.dict_node[v]
                        img = pi.sg2dgm_accelerate(uu, vv, hop, norm=norm, extended_flag=True, resolution=5, descriptor=desc)
                        rows.append(img.reshape(-1))
                        cls.append(0)
                    except BaseException as e:  # noqa
                        rows.append(np.zeros(25))
                        cls.append(EXC_CLASS.get(type(e).__name__, 9))
                    if cls[-1] in (0, 4) and i % 4 == 0:
                        nodes_u = [uu] + [x for _, x in nx.bfs_edges(pi.graph, uu, depth_limit=hop)]
                        nodes_v = [vv] + [x for _, x in nx.bfs_edges(pi.graph, vv, depth_limit=hop)]
                        sub = pi.graph.subgraph(list(set(nodes_u) & set(nodes_v))).copy()
                        g = r2d.filtration(sub, uu, vv, hop, ricci_curv=pi.ricci_curv).build_fv(weight_graph=True, norm=norm)
                        ids = np.array(sorted(inv[x] for x in g.nodes()), dtype=np.int64)
                        back = {inv[x]: x for x in g.nodes()}
                        ids_l.append(ids)
                        f_l.append(np.array([g.nodes[back[k]][desc] for k in ids.tolist()], dtype=np.float64))
                        sel.append(i)
                tag = "%s_%s_hop%d" % (desc, "norm" if norm else "raw", hop)
                out["pi_" + tag] = np.stack(rows)
                out["cls_" + tag] = np.array(cls, dtype=np.int64)
                out["fsel_" + tag] = np.array(sel, dtype=np.int64)
                out["ids_" + tag], out["offs_" + tag] = ragged(ids_l, 0, np.int64)
                out["f_" + tag], _ = ragged(f_l, 0, np.float64)
                print("G8", tag, "classes:", np.bincount(cls, minlength=5), "f cases:", len(sel))
        # the shipped entry point with its own default descriptor ('min', :362) agrees with the mirror
        pi.get_pimg_for_all_edges(pairs.tolist(), cores=1, hop=hop, norm=True, extended_flag=True, resolution=5)
        assert np.array_equal(pi.pi_sg, out["pi_min_norm_hop%d" % hop])
    np.savez_compressed(os.path.join(HERE, "variants.npz"), **out)


def make_g4c(mods):
    """PDGNN node-centred vicinity (data_utils_NC.py:95-187, filt='ricci'): ball_hop(u), ONE root, unreachable -> 100,
    f / (max + 1e-10) (:27-50); mode='filtration' values and induced edges, mode='PI' diagrams and images."""
    import networkx as nx
    kd = mods["kd_nc"]
    if kd is None:
        print("G4c skipped")
        return
    d = np.load(os.path.join(HERE, "e2e.npz"))
    edges, kappa = d["edges"], d["kappa"]
    g = nx.Graph()
    g.add_edges_from([(int(a), int(b)) for a, b in edges])
    ricci = sorted([[int(a), int(b), float(k)] for (a, b), k in zip(edges.tolist(), kappa.tolist())] +
                   [[int(b), int(a), float(k)] for (a, b), k in zip(edges.tolist(), kappa.tolist())])
    rs = np.random.RandomState(77)
    roots = sorted(set(rs.choice(sorted(g.nodes()), size=70, replace=False).tolist()) | {int(d["n_nodes"]) - 6, int(d["n_nodes"]) - 5})
    ids_l, f_l, e_l, o0_l, e1_l, pis, pi0s, pi1s, rt, hops = [], [], [], [], [], [], [], [], [], []
    for hop in (1, 2):
        for u in roots:
            fv, ei = kd.compute_persistence_image(g, u, filt="ricci", hop=hop, ricci_curv=ricci, mode="filtration")
            if fv is None:
                continue
            res = kd.compute_persistence_image(g, u, filt="ricci", hop=hop, ricci_curv=ricci, mode="PI")
            nodes = [u] + [x for _, x in nx.bfs_edges(g, u, depth_limit=hop)]                     # :97-99
            sub = nx.convert_node_labels_to_integers(g.subgraph(nodes), label_attribute="old_label")
            old = np.array([sub._node[k]["old_label"] for k in range(len(sub))], dtype=np.int64)
            order = np.argsort(old)
            ids_l.append(old[order])
            f_l.append(np.asarray(fv, dtype=np.float64)[order])
            ee = np.sort(old[np.asarray(ei).T.reshape(-1, 2)], axis=1)
            e_l.append(ee[np.lexsort((ee[:, 1], ee[:, 0]))])
            o0_l.append(np.asarray(res[0], dtype=np.float64).reshape(-1, 2))
            e1_l.append(np.asarray(res[1], dtype=np.float64).reshape(-1, 2))
            pis.append(np.asarray(res[2], dtype=np.float64).reshape(-1))
            pi0s.append(np.asarray(res[5], dtype=np.float64).reshape(-1))
            pi1s.append(np.asarray(res[6], dtype=np.float64).reshape(-1))
            rt.append(u)
            hops.append(hop)
    ids_flat, offs = ragged(ids_l, 0, np.int64)
    f_flat, _ = ragged(f_l, 0, np.float64)
    e_flat, e_offs = ragged(e_l, 2, np.int64)
    o0, o0_offs = ragged(o0_l, 2)
    e1, e1_offs = ragged(e1_l, 2)
    np.savez_compressed(os.path.join(HERE, "kd_nc.npz"), roots=np.array(rt, dtype=np.int64), hop=np.array(hops), ids=ids_flat,
                        f=f_flat, offs=offs, edges=e_flat, e_offs=e_offs, ord0=o0, ord0_offs=o0_offs, ext1=e1, ext1_offs=e1_offs,
                        pi=np.stack(pis), pi0=np.stack(pi0s), pi1=np.stack(pi1s))
    print("G4c cases:", len(rt), "largest ball:", max(len(x) for x in ids_l))


def make_g10(mods):
    """Net.decode of the reference itself (baselines/TLCGNN.py:27-62): plain torch + numpy.  The module imports
    torch_geometric.nn.GCNConv / ChebConv at the top (third-party, absent: SURVEY 8c) but decode never touches the two conv
    layers, so they are stubbed as empty modules; the `.cuda()` at :52-53 is a device hop with no arithmetic and is made the
    identity for the duration of the call (this container has no GPU).  Everything else -- the slicing by split, the
    np.random.randint negatives, renorm_, the 41->25->1 head, the Fermi-Dirac link function -- is the reference's own code."""
    import torch

    class _Conv(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    tgnn = _stub("torch_geometric.nn", GCNConv=_Conv, ChebConv=_Conv)
    sys.modules["torch_geometric"].nn = tgnn
    tmp = os.path.dirname(mods["lds"].__file__)
    os.makedirs(os.path.join(tmp, "baselines"), exist_ok=True)
    open(os.path.join(tmp, "baselines", "__init__.py"), "w").close()
    link = os.path.join(tmp, "baselines", "TLCGNN.py")
    if not os.path.exists(link):
        os.symlink(os.path.join(REF, "baselines", "TLCGNN.py"), link)
    ref = importlib.import_module("baselines.TLCGNN")

    rs = np.random.RandomState(2024)
    n_nodes, dim = 400, 5
    tp, tn, vp, vn, sp_, sn = 300, 900, 40, 40, 60, 60
    n_pairs = tp + tn + vp + vn + sp_ + sn
    pairs = rs.randint(0, n_nodes, size=(n_pairs, 2)).astype(np.int64)
    pairs[5] = (7, 7)                                                    # a self pair: (a - b)^2 = 0
    y = np.concatenate([np.ones(tp), np.zeros(tn), np.ones(vp), np.zeros(vn), np.ones(sp_), np.zeros(sn)]).astype(np.int64)
    PI = rs.gamma(0.6, 0.4, size=(n_pairs, dim * dim))                  # non-negative, heavy near 0 like real images
    PI[rs.rand(n_pairs) < 0.3] = 0.0                                     # zero rows (far pairs)
    PI[rs.rand(n_pairs) < 0.05] *= 400.0                                 # a few huge rows: |d| beyond the clamp at 40
    emb = rs.randn(n_nodes, 16).astype(np.float32) * 0.35               # norms around 1.4: most rows are rescaled
    emb[::3] *= 0.2                                                      # a third of the rows with norm < 1 (untouched)
    emb[11] = 0.0

    class _Data:
        pass

    data = _Data()
    data.total_edges = pairs
    data.total_edges_y = torch.from_numpy(y)
    data.train_pos, data.train_neg, data.val_pos, data.val_neg, data.test_pos, data.test_neg = tp, tn, vp, vn, sp_, sn
    torch.manual_seed(1234)
    net = ref.Net(data, 32, 2, PI, dimension=dim)
    net.eval()
    with torch.no_grad():
        net.linear.weight.mul_(6.0)                                      # spread |d| over (0, 40): both clamp ends are hit
        net.linear.bias.add_(0.5)
        net.linear_1.weight.mul_(3.0)
    out = dict(n_nodes=n_nodes, pairs=pairs, y=y, PI=PI, emb=emb, counts=np.array([tp, tn, vp, vn, sp_, sn]),
               lin1_w=net.linear_1.weight.detach().numpy().copy(), lin1_b=net.linear_1.bias.detach().numpy().copy(),
               lin_w=net.linear.weight.detach().numpy().copy(), lin_b=net.linear.bias.detach().numpy().copy(), np_seed=4321)
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        for kind in ("train", "val", "test"):
            np.random.seed(4321)
            e = torch.from_numpy(emb.copy())
            with torch.no_grad():
                prob, yy = net.decode(data, e, kind)
            out["prob_" + kind] = prob.numpy().astype(np.float32)
            out["y_" + kind] = yy.numpy().astype(np.float32)
            out["emb_after_" + kind] = e.numpy().copy()                  # renorm_ is in place (:48)
        np.random.seed(4321)
        out["train_index"] = np.random.randint(0, tn, tp)
    finally:
        torch.Tensor.cuda = real_cuda
    sat = [(float(out["prob_" + k].min()), float(out["prob_" + k].max())) for k in ("train", "val", "test")]
    print("G10 decode: rows", [len(out["prob_" + k]) for k in ("train", "val", "test")], "prob ranges", sat,
          "rows rescaled:", int((np.linalg.norm(emb, axis=1) > 1).sum()), "of", n_nodes)
    np.savez_compressed(os.path.join(HERE, "decode.npz"), **out)


def make_g4d(mods):
    """The structural filtrations of the PDGNN fork: data_utils_NC.compute_persistence_image with filt 'degree' / 'centrality' /
    'clustering' (:124-135; the last two are what train_Teacher_Model.py:158-159 trains on) and data_utils_LP with filt 'degree'
    (:131-133): filtration values, Ord0 / Ext1 and the three images per vicinity."""
    import networkx as nx
    kd_nc, kd_lp = mods["kd_nc"], mods["kd_lp"]
    if kd_nc is None or kd_lp is None:
        print("G4d skipped")
        return
    d = np.load(os.path.join(HERE, "e2e.npz"))
    edges = d["edges"]
    g = nx.Graph()
    g.add_edges_from([(int(a), int(b)) for a, b in edges])
    rs = np.random.RandomState(91)
    roots = sorted(rs.choice(sorted(g.nodes()), size=24, replace=False).tolist())
    pairs = [tuple(int(x) for x in edges[i]) for i in rs.choice(len(edges), size=16, replace=False)] + \
            [tuple(int(x) for x in rs.choice(sorted(g.nodes()), size=2, replace=False)) for _ in range(8)]
    out = dict(kind=[], hop=[], u=[], v=[])
    ids_l, f_l, o0_l, e1_l, pis, pi0s, pi1s = [], [], [], [], [], [], []

    def take(kind, hop, u, v, nodes, fv, res):
        sub = nx.convert_node_labels_to_integers(g.subgraph(nodes), label_attribute="old_label")
        old = np.array([sub._node[k]["old_label"] for k in range(len(sub))], dtype=np.int64)
        order = np.argsort(old)
        ids_l.append(old[order]); f_l.append(np.asarray(fv, dtype=np.float64)[order])
        o0_l.append(np.asarray(res[0], dtype=np.float64).reshape(-1, 2)); e1_l.append(np.asarray(res[1], dtype=np.float64).reshape(-1, 2))
        pis.append(np.asarray(res[2], dtype=np.float64).reshape(-1)); pi0s.append(np.asarray(res[5], dtype=np.float64).reshape(-1))
        pi1s.append(np.asarray(res[6], dtype=np.float64).reshape(-1))
        out["kind"].append(kind); out["hop"].append(hop); out["u"].append(u); out["v"].append(v)

    for ki, filt in enumerate(("degree", "centrality", "clustering")):
        for hop in (1, 2):
            for u in roots:
                fv, ei = kd_nc.compute_persistence_image(g, u, filt=filt, hop=hop, mode="filtration")
                if fv is None or max(fv) == 0:       # (clustering of a tree is all zero: every key ties, nothing to compare)
                    continue
                res = kd_nc.compute_persistence_image(g, u, filt=filt, hop=hop, mode="PI")
                nodes = [u] + [x for _, x in nx.bfs_edges(g, u, depth_limit=hop)]
                take(ki, hop, u, -1, nodes, fv, res)
    for hop in (1, 2):
        for (u, v) in pairs:
            fv, ei = kd_lp.compute_persistence_image(g, u, v, filt="degree", hop=hop, mode="filtration")
            if fv is None:
                continue
            res = kd_lp.compute_persistence_image(g, u, v, filt="degree", hop=hop, mode="PI")
            nu = [u] + [x for _, x in nx.bfs_edges(g, u, depth_limit=hop)]
            nv = [v] + [x for _, x in nx.bfs_edges(g, v, depth_limit=hop)]
            # data_utils_LP.py:107-111, the very expression: a subgraph view iterates the SET built from this list, whose order
            # depends on the insertion order where hashes collide -- and `fv` is listed in that order
            nodes = list(set(nu) & set(nv)) + [u] + [v]
            take(3, hop, u, v, nodes, fv, res)
    ids_flat, offs = ragged(ids_l, 0, np.int64)
    f_flat, _ = ragged(f_l, 0, np.float64)
    o0, o0_offs = ragged(o0_l, 2)
    e1, e1_offs = ragged(e1_l, 2)
    np.savez_compressed(os.path.join(HERE, "kd_struct.npz"), kind=np.array(out["kind"]), hop=np.array(out["hop"]), u=np.array(out["u"]),
                        v=np.array(out["v"]), ids=ids_flat, f=f_flat, offs=offs, ord0=o0, ord0_offs=o0_offs, ext1=e1, ext1_offs=e1_offs,
                        pi=np.stack(pis), pi0=np.stack(pi0s), pi1=np.stack(pi1s))
    print("G4d cases:", len(out["kind"]), "by kind:", np.bincount(out["kind"]).tolist(), "largest:", max(len(x) for x in ids_l))


def make_g4e(mods):
    """filt='hks' -- the default of all three compute_persistence_image signatures (data_utils_GC.py:114-116 with the shipped
    loop's times 0.1 and 10, :315-319; data_utils_NC.py:120-122; data_utils_LP.py:128-130): values, Ord0 / Ext1, images.
    GC graphs have nodes 0..n-1 (values comparable bit for bit); NC / LP values are stored in ascending-id order."""
    import networkx as nx
    kd_gc, kd_nc, kd_lp = mods["kd_gc"], mods["kd_nc"], mods["kd_lp"]
    if kd_gc is None or kd_nc is None or kd_lp is None:
        print("G4e skipped")
        return
    d = np.load(os.path.join(HERE, "e2e.npz"))
    edges = d["edges"]
    g = nx.Graph()
    g.add_edges_from([(int(a), int(b)) for a, b in edges])
    rs = np.random.RandomState(17)
    meta = dict(kind=[], hop=[], u=[], v=[], time=[], n=[])
    ids_l, f_l, e_l, o0_l, e1_l, pis, pi0s, pi1s = [], [], [], [], [], [], [], []

    def take(kind, hop, u, v, t, old, fv, ei_old, res):
        order = np.argsort(old)
        ids_l.append(old[order]); f_l.append(np.asarray(fv, dtype=np.float64)[order]); e_l.append(ei_old)
        o0_l.append(np.asarray(res[0], dtype=np.float64).reshape(-1, 2)); e1_l.append(np.asarray(res[1], dtype=np.float64).reshape(-1, 2))
        pis.append(np.asarray(res[2], dtype=np.float64).reshape(-1)); pi0s.append(np.asarray(res[5], dtype=np.float64).reshape(-1))
        pi1s.append(np.asarray(res[6], dtype=np.float64).reshape(-1))
        for k, x in zip(("kind", "hop", "u", "v", "time", "n"), (kind, hop, u, v, t, len(old))):
            meta[k].append(x)

    for gi in range(24):                                                  # GC: whole graphs, nodes 0..n-1
        n = int(max(3, rs.poisson(22)))
        ee = random_connected_graph(rs, n, int(rs.randint(0, 4)))
        gg = nx.Graph()
        gg.add_nodes_from(range(n))
        gg.add_edges_from([(int(a), int(b)) for a, b in ee])
        for t in (0.1, 10):
            res = kd_gc.compute_persistence_image(gg, filt="hks", hks_time=t, mode="PI")
            take(2, 0, -1, -1, t, np.arange(n), res[3], np.asarray(res[4]).T.astype(np.int64), res)
    roots = sorted(rs.choice(sorted(g.nodes()), size=10, replace=False).tolist())
    for hop in (1, 2):
        for u in roots:
            for t in (0.1, 10):
                fv, ei = kd_nc.compute_persistence_image(g, u, filt="hks", hks_time=t, hop=hop, mode="filtration")
                if fv is None:
                    continue
                res = kd_nc.compute_persistence_image(g, u, filt="hks", hks_time=t, hop=hop, mode="PI")
                nodes = [u] + [x for _, x in nx.bfs_edges(g, u, depth_limit=hop)]
                sub = nx.convert_node_labels_to_integers(g.subgraph(nodes), label_attribute="old_label")
                old = np.array([sub._node[k]["old_label"] for k in range(len(sub))], dtype=np.int64)
                take(0, hop, u, -1, t, old, fv, np.zeros((0, 2), dtype=np.int64), res)
    pairs = [tuple(int(x) for x in edges[i]) for i in rs.choice(len(edges), size=10, replace=False)]
    for hop in (1, 2):
        for (u, v) in pairs:
            fv, ei = kd_lp.compute_persistence_image(g, u, v, filt="hks", hks_time=0.1, hop=hop, mode="filtration")
            if fv is None:
                continue
            res = kd_lp.compute_persistence_image(g, u, v, filt="hks", hks_time=0.1, hop=hop, mode="PI")
            nu = [u] + [x for _, x in nx.bfs_edges(g, u, depth_limit=hop)]
            nv = [v] + [x for _, x in nx.bfs_edges(g, v, depth_limit=hop)]
            nodes = list(set(nu) & set(nv)) + [u] + [v]
            sub = nx.convert_node_labels_to_integers(g.subgraph(nodes), label_attribute="old_label")
            old = np.array([sub._node[k]["old_label"] for k in range(len(sub))], dtype=np.int64)
            take(1, hop, u, v, 0.1, old, fv, np.zeros((0, 2), dtype=np.int64), res)
    ids_flat, offs = ragged(ids_l, 0, np.int64)
    f_flat, _ = ragged(f_l, 0, np.float64)
    e_flat, e_offs = ragged(e_l, 2, np.int64)
    o0, o0_offs = ragged(o0_l, 2)
    e1, e1_offs = ragged(e1_l, 2)
    np.savez_compressed(os.path.join(HERE, "kd_hks.npz"), ids=ids_flat, f=f_flat, offs=offs, edges=e_flat, e_offs=e_offs, ord0=o0,
                        ord0_offs=o0_offs, ext1=e1, ext1_offs=e1_offs, pi=np.stack(pis), pi0=np.stack(pi0s), pi1=np.stack(pi1s),
                        **{k: np.array(v) for k, v in meta.items()})
    print("G4e cases:", len(meta["kind"]), "by kind (NC, LP, GC):", np.bincount(meta["kind"]).tolist(), "largest:", max(meta["n"]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true", help="also time the reference on the PubMed-shaped graph")
    ap.add_argument("--only", default="", help="regenerate a single fixture (g4b | g7b | g8 | g4c | g4d | g4e | g10)")
    args = ap.parse_args()
    assert sys.version_info[:2] < (3, 12), "python>=3.12 sums with compensation: goldens would differ (SURVEY A.2)"
    mods = import_reference()
    if args.only:
        {"g4b": make_g4b, "g7b": make_g7b, "g8": make_g8, "g4c": make_g4c, "g4d": make_g4d, "g4e": make_g4e, "g10": make_g10}[args.only](mods)
        return
    make_g1_g2(mods)
    make_g3(mods)
    make_g4_g5(mods, do_time=args.time)
    make_g4b(mods)
    make_g6(mods)
    make_g7(mods)
    make_g7b(mods)
    make_g8(mods)
    make_g4c(mods)
    make_g4d(mods)
    make_g4e(mods)
    make_g10(mods)


if __name__ == "__main__":
    main()

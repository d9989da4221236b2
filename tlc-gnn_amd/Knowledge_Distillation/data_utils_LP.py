"""Drop-in for the vicinity / filtration part of the reference's Knowledge_Distillation/data_utils_LP.py (PDGNN, link
prediction), filt='ricci'.

  ricci_filtration.build_fv :35-65, compute_persistence_image :105-200 (modes 'filtration' and 'PI').

Differences from the TLC-GNN vicinity (sg2dgm/riccidist2dgm.py) that the HIP kernels take as flags: the two roots are always
part of the subgraph (:111), there is no connectivity assert (an unreachable root costs the sentinel 100, :41-49) and the
normalisation divides by max + 1e-10 (:64).  Node labels of a vicinity are positions in ASCENDING original id (the reference's
`convert_node_labels_to_integers` order is arbitrary); edges are listed once, lower label first.
filt='degree' (:131-133) and 'hks' (:128-130, the signature's default): the same vicinities, f from `structural_filtration` (host side:
networkx's arithmetic, scipy's eigh).  The CBGNN cycle helpers (:256-448, dead code in the reference) and `call` are not reproduced.
"""
import sys

import numpy as np

from .. import engine, _lib

KD_LP_FLAGS = _lib.INCLUDE_ROOTS | _lib.NORM_EPS | _lib.UNREACHABLE_100


STRUCTURAL_FILTS = ("degree", "centrality", "clustering", "hks")


def hks_signature(n, edges, time):
    """hks_signature of the reference (data_utils_LP.py:96-100, data_utils_NC.py:88-92, data_utils_GC.py:90-94): the heat-kernel
    signature sum_k exp(-t lambda_k) phi_k(x)^2 of the normalised Laplacian -- the same scipy calls on the same kind of matrix
    (`csgraph.laplacian(A, normed=True)`, dense `eigh`), n nodes, edges int[m, 2] local ids.  Host side: a dense symmetric
    eigenproblem per graph.  Node order: ascending id here, networkx's subgraph order there -- the values of a node agree to
    rounding (1e-12), bit for bit when the orders coincide (data_utils_GC: nodes 0..n-1)."""
    import scipy.sparse as sp
    from scipy.sparse import csgraph
    from scipy.linalg import eigh
    e = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    A = sp.csr_matrix((np.ones(2 * len(e), dtype=np.int64), (np.concatenate([e[:, 0], e[:, 1]]), np.concatenate([e[:, 1], e[:, 0]]))), shape=(n, n))
    L = csgraph.laplacian(A, normed=True)
    egvals, egvectors = eigh(L.toarray())
    return np.square(egvectors).dot(np.diag(np.exp(-time * egvals))).sum(axis=1)


def structural_filtration(kind, node_ptr, edge_ptr, edges, hks_time=0.1):
    """The node functions of the induced subgraph that the reference computes with networkx (data_utils_LP.py:131-133 'degree';
    data_utils_NC.py:124-135 'centrality', 'clustering', 'degree'), each divided by (max + 1e-10), for a packed batch of
    vicinities: node_ptr / edge_ptr int64[B+1], edges int[sum m, 2] local ids -> float64[sum n].  Host side (numpy + scipy.sparse):
    they are functions of a few hundred nodes; the vicinities themselves come from the device.  Same operations in the same
    order as networkx, so the values are the reference's bit for bit -- for SIMPLE graphs, which is what the extraction hands over
    (graph2pi strips self loops): an edge (a, a) would count into `degree` and the triangle counts here, networkx's clustering drops
    it; a one-node vicinity gets centrality 1 like nx.degree_centrality (not the division by n - 1 = 0):
      degree      d                                  (subgraph.degree())
      centrality  d * (1.0 / (n - 1.0))              (nx.degree_centrality)
      clustering  t / (d * (d - 1)), t = sum over the neighbours w of |N(v) & N(w)| (each triangle twice), 0 where t == 0
                                                      (nx.clustering, unweighted)
      hks         `hks_signature` at time hks_time    (a dense eigenproblem per vicinity)"""
    import scipy.sparse as sp
    if kind not in STRUCTURAL_FILTS:
        raise ValueError("filt should be one of %s" % (STRUCTURAL_FILTS,))
    node_ptr = np.asarray(node_ptr, dtype=np.int64)
    edge_ptr = np.asarray(edge_ptr, dtype=np.int64)
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    N = int(node_ptr[-1])
    if kind == "hks":                                        # (:128-130 / :120-122: hks_signature, then /= max + 1e-10, per graph)
        out = np.zeros(N, dtype=np.float64)
        for k in range(len(node_ptr) - 1):
            a, b = int(node_ptr[k]), int(node_ptr[k + 1])
            if b > a:
                v = hks_signature(b - a, edges[int(edge_ptr[k]):int(edge_ptr[k + 1])], hks_time)
                out[a:b] = v / (max(v) + 1e-10)
        return out
    owner = np.repeat(np.arange(len(node_ptr) - 1), np.diff(node_ptr))
    base = np.repeat(node_ptr[:-1], np.diff(edge_ptr))
    a, b = edges[:, 0] + base, edges[:, 1] + base
    deg = np.bincount(np.concatenate([a, b]), minlength=N).astype(np.int64)
    if kind == "degree":
        raw = deg.astype(np.float64)
    elif kind == "centrality":
        n_of = np.diff(node_ptr)[owner].astype(np.float64)
        raw = np.where(n_of > 1.0, deg.astype(np.float64) * (1.0 / np.where(n_of > 1.0, n_of - 1.0, 1.0)), 1.0)   # (n == 1: nx returns 1)
    else:
        A = sp.csr_matrix((np.ones(2 * len(a), dtype=np.int64), (np.concatenate([a, b]), np.concatenate([b, a]))), shape=(N, N))
        t = np.asarray((A @ A).multiply(A).sum(axis=1)).reshape(-1).astype(np.int64)
        den = (deg * (deg - 1)).astype(np.float64)
        raw = np.where(t > 0, t.astype(np.float64) / np.where(den > 0, den, 1.0), 0.0)
    mx = np.zeros(len(node_ptr) - 1, dtype=np.float64)
    if N:
        np.maximum.at(mx, owner, raw)
    return raw / (mx[owner] + 1e-10) if N else raw


class Vicinities:
    """Device-resident weighted graph for PDGNN's edge-centred vicinities; build once, query many pairs.
    ricci_curv=None: no curvature (the structural filtrations need only the vicinities): unit edge weights."""

    def __init__(self, g, ricci_curv):
        from ..sg2dgm.riccidist2dgm import graph2pi, _edge_array
        if ricci_curv is None:
            e = _edge_array(g)
            ricci_curv = np.concatenate([np.concatenate([e, e[:, ::-1]]).astype(np.float64), np.zeros((2 * len(e), 1))], axis=1)
        self._g2p = graph2pi(g, ricci_curv, keep_labels=True)   # kappa+1 weights (riccidist2dgm.py:216-226)
        self.dict_node = self._g2p.dict_node
        self.inv = np.arange(self._g2p.n_nodes, dtype=np.int64)
        for old, new in self.dict_node.items():
            self.inv[new] = old

    def batch(self, pairs, hop, node_cap=None, edge_cap=None, flags=None, filt='ricci', hks_time=0.1):
        """pairs: [E,2] original labels -> dict of CUDA tensors: node_ptr int64[E+1], edge_ptr int64[E+1], ids int64 (original
        labels, ascending inside a vicinity), f float64, edges int32 [sum m, 2] (local ids, lower first), status uint8[E].
        Vicinities without an edge have empty slices (the reference returns (None, None) for them, :117-118).
        filt: 'ricci' (the weighted-distance filtration of the device kernels) or one of STRUCTURAL_FILTS (f replaced by
        `structural_filtration` of the extracted vicinities).
        node_cap / edge_cap: per-pair capacities of an intermediate layout (one extraction; raises if a vicinity is larger); neither
        given: sizes first, exact offsets, two extractions (`tlc_vicinity_sizes` + `tlc_pack_offsets`)."""
        import torch
        dev_graph = self._g2p._device_graph()
        mapped = torch.from_numpy(self._g2p._map_pairs(pairs)).cuda()
        fl = KD_LP_FLAGS if flags is None else flags
        if getattr(self, "_inv_dev", None) is None or self._inv_dev.device != mapped.device:
            self._inv_dev = torch.from_numpy(self.inv).to(mapped.device)
        if node_cap is None and edge_cap is None:
            # no capacities given: sizes first (the extraction alone), exact offsets from them, then the filtration writes the packed
            # batch directly -- nothing to guess, nothing to overflow (a capacity of "whole graph" per pair would be 60 GB for 100 000
            # PubMed pairs), at the price of extracting twice
            n0, m0 = dev_graph.vicinity_sizes(mapped, hop, flags=fl)
            node_ptr, edge_ptr, totals = engine.pack_offsets(n0, m0)
            E = len(n0)
            lo_n, _, tot_n, tot_m = totals.tolist() if E else (0, 0, 0, 0)
            if lo_n < 0:
                raise RuntimeError("a vicinity has more nodes than the packed local ids hold (status TLC_ST_TOO_LARGE)")
            _, ids, f, n, st, _, edges, m = dev_graph.vicinity_filtration(mapped, hop, flags=fl, zero=False,
                                                                         offsets=(node_ptr, edge_ptr, tot_n, tot_m))
            counts_n, counts_m = node_ptr[1:] - node_ptr[:-1], edge_ptr[1:] - edge_ptr[:-1]
            # the second pass wrote what the first one counted?  (a vicinity that fails inside the tier kernel writes no edges: its
            # slice of the exact-size arrays would be uninitialised memory)
            ok_n = torch.where(edge_ptr[1:] > edge_ptr[:-1], n.long() == counts_n, torch.ones_like(counts_n, dtype=torch.bool))
            if E and not bool((ok_n & (m.long() == counts_m)).all()):
                raise RuntimeError("Vicinities.batch: the filtration pass wrote other sizes than the sizes pass counted "
                                   "(a vicinity failed in its tier kernel, or the graph has self loops / one-sided entries)")
            owner = torch.arange(E, device=mapped.device)
            pn = torch.repeat_interleave(owner, counts_n, output_size=int(tot_n))
            pe = torch.repeat_interleave(owner, counts_m, output_size=int(tot_m))
            out_ids, out_f, out_e = self._inv_dev[ids[:int(tot_n)].long()], f[:int(tot_n)], edges[:int(tot_m)]
            if filt != 'ricci':
                out_f = torch.from_numpy(structural_filtration(filt, node_ptr.cpu().numpy(), edge_ptr.cpu().numpy(), out_e.cpu().numpy(),
                                                               hks_time=hks_time)).to(out_f.device)
            return dict(node_ptr=node_ptr, edge_ptr=edge_ptr, ids=out_ids, f=out_f, edges=out_e, status=st, pair_of_node=pn, pair_of_edge=pe)
        n_cap = dev_graph.n_nodes if node_cap is None else int(node_cap)
        e_cap = max(dev_graph.nnz // 2, 1) if edge_cap is None else int(edge_cap)
        offs, ids, f, n, st, eoffs, edges, m = dev_graph.vicinity_filtration(mapped, hop, flags=fl, cap=n_cap, edge_cap=e_cap, zero=False)
        # packed offsets on the device (one kernel; no edge -> (None, None): an empty slice); ONE host read: the two totals (to size
        # the packed arrays) and the two capacity checks
        E = len(n)
        node_ptr, edge_ptr, totals = engine.pack_offsets(n, m)
        lo_n, lo_m, tot_n, tot_m = totals.tolist() if E else (0, 0, 0, 0)
        if lo_n < 0 or lo_m < 0:
            raise RuntimeError("vicinity larger than the requested node_cap / edge_cap")
        out_ids, out_f, out_e, pn, pe = engine.pack_vicinities(offs, ids, f, eoffs, edges, node_ptr, edge_ptr, int(tot_n), int(tot_m),
                                                               label=self._inv_dev)
        if filt != 'ricci':
            out_f = torch.from_numpy(structural_filtration(filt, node_ptr.cpu().numpy(), edge_ptr.cpu().numpy(), out_e.cpu().numpy(),
                                                           hks_time=hks_time)).to(out_f.device)
        return dict(node_ptr=node_ptr, edge_ptr=edge_ptr, ids=out_ids, f=out_f, edges=out_e, status=st, pair_of_node=pn, pair_of_edge=pe)


def stacked(batch):
    """A `Vicinities.batch` result -> (x float32 [n,1], edge_index int64 [2, m+n]): all vicinities as ONE block-diagonal input of
    Teacher_Model.forward (pass graph_ptr=batch['node_ptr'], edge_ptr=batch['edge_ptr']) -- what gcn_LP_GIN.Net.compute_PI (:43-64)
    builds per candidate edge: the vicinity's edge_index plus self loops (last), its filtration as a float32 column."""
    ei, x = engine.stack_batch(batch["node_ptr"], batch["edge_ptr"], batch["edges"], batch["f"])
    return x, ei


_CACHE = {}


def _vicinities(g, ricci_curv):
    """One device graph kept between calls, reused only for the SAME graph and curvature objects (held here, so their ids
    cannot pass to other objects) with an unchanged content stamp -- the rule of data_utils_NC._vicinities."""
    from .data_utils_NC import _graph_stamp
    ent = _CACHE.get("entry")
    stamp = _graph_stamp(g, ricci_curv)
    if ent is None or ent[0] is not g or ent[1] is not ricci_curv or ent[2] != stamp:
        _CACHE["entry"] = ent = (g, ricci_curv, stamp, Vicinities(g, ricci_curv))
    return ent[3]


def compute_persistence_image(g, u, v, filt='hks', hks_time=0.1, hop=2, ricci_curv=None, mode='PI', num_models=5,
                              max_loop_len=10, cycle_the=2):
    """Reference signature (:105).  filt='hks' (:128-130), 'degree' (:131-133) or 'ricci'; mode 'filtration' -> (filtration_val
    list, edge_index LongTensor[2,m]) or (None, None); mode 'PI' -> the reference's 9-tuple (times are 0)."""
    import torch
    if filt not in ('ricci', 'degree', 'hks'):
        print("Error: 'filt' should be 'hks', 'degree' or 'ricci'! ")          # :152-153
        sys.exit()
    b = _vicinities(g, ricci_curv).batch([[u, v]], hop, filt=filt, hks_time=hks_time)
    if int(b["edge_ptr"][-1]) == 0:
        return None, None
    fv = b["f"].cpu().numpy()
    edge_index = b["edges"].t().contiguous().long().cpu()
    if mode == 'filtration':
        return fv.tolist(), edge_index
    if mode != 'PI':
        raise ValueError("mode must be 'PI' or 'filtration'")
    return diagrams_and_images(b, fv, edge_index)


def diagrams_and_images(b, fv, edge_index):
    """mode 'PI' tail shared by the edge- and node-centred vicinities (data_utils_LP.py:178-197, data_utils_NC.py:155-183):
    original_extended_persistence (Knowledge_Distillation fork: zero-persistence pairs kept) -> Ord0, Ext1, then the three
    images PI(Ord0 ++ Ext1), PI0, PI1; the reference's 9-tuple (times are 0)."""
    import torch
    n, m = len(fv), edge_index.shape[1]
    r = engine.pd_from_filtration(torch.tensor([0, n], dtype=torch.int64, device="cuda"),
                                  torch.tensor([0, m], dtype=torch.int64, device="cuda"),
                                  b["edges"].contiguous(), b["f"].contiguous(), _lib.KEEP_ZERO_PERS)
    c = r["counts"][0].cpu().numpy()
    if c[3] != 1 and int(c[2]) != m - n + int(c[3]):
        # not connected, and a Pos edge lies outside the component of the tree's root: Parent[...] of the reference's walk has no
        # such node (accelerated_PD.py:132-148, KeyError).  (Components without a cycle -- the isolated roots of a far pair -- are
        # no obstacle there either: every one of the m - n + #components Pos edges was walked.)
        raise KeyError("vicinity is not connected and a cycle lies outside the root's component: the reference's Accelerate_PD raises "
                       "here (accelerated_PD.py:132-148)")
    ord0, ext1 = r["up"][:c[0]], r["one"][:c[2]]

    def img(pts):
        if pts.shape[0] == 0:
            return np.zeros(25)
        return engine.pi_raster(torch.tensor([0, pts.shape[0]], dtype=torch.int64, device="cuda"), pts.contiguous(), 5)[0].cpu().numpy()
    PI0, PI1 = img(ord0), img(ext1)
    both = torch.cat((ord0, ext1))
    pers_img = PI1 if c[0] == 0 else (PI0 if c[2] == 0 else img(both))
    return ord0.cpu().numpy(), ext1.cpu().numpy(), pers_img, fv.tolist(), edge_index, PI0, PI1, 0.0, 0.0


def load_ricci_file(filename):
    """The reference's curvature cache: one `u v kappa` line per directed edge (data_utils_LP.py:233-240 writes it)."""
    ricci_list = []
    with open(filename) as f:
        for line in f:
            a, b, k = line.split()
            ricci_list.append([int(a), int(b), float(k)])
    return ricci_list


def compute_ricci_curvature(data, data_name, cache_dir='./data/curvature'):
    """data_utils_LP.py:202-242: Ollivier-Ricci curvature (alpha 0.5, Sinkhorn) of data.edge_index, cached as a text file of
    `u v kappa` lines.  The reference delegates to GraphRicciCurvature and writes under a hard-wired absolute path; here the
    curvature comes from the GPU (loaddatas.compute_ricci_curvature -> tlc_ollivier_ricci_sinkhorn) and the path is a
    parameter (None disables the cache)."""
    import os
    from ..loaddatas import compute_ricci_curvature as _compute
    filename = os.path.join(cache_dir, 'graph_' + data_name + '_removevaltest.edge_list') if cache_dir else None
    if filename and os.path.exists(filename):
        print("curvature file exists, directly loading")
        return load_ricci_file(filename)
    ricci_list = _compute(data)
    if filename:
        os.makedirs(os.path.dirname(filename), exist_ok=True)
        with open(filename, 'w') as f:
            for a, b, k in ricci_list:
                f.write(str(a) + " " + str(b) + " " + repr(float(k)) + "\n")
    return ricci_list

"""Drop-in for the reference's sg2dgm/accelerated_PD.py (TLC-GNN fork: zero-persistence pairs dropped).

  perturb_filter_function :6-23, Union_find :26-113, Accelerate_PD :115-178.

The simplex_filter dict keeps the reference's exact structure (it is the interface between the three functions);
the union-find passes and the cycle swap run in `tlc_pd_from_filtration` (include/tlcgnn.h).
"""
import numpy as np

from .. import engine, _lib


EE = 1e-6            # accelerated_PD.py:9
MAX_FILTER = 101     # accelerated_PD.py:10


def build_simplex_filter(nodes, values, edges):
    """The dict the reference's three functions pass around: node -> {'old', 'new'} (both the filtration value) and
    (a, b) -> {'asc': hi + (lo + 1)*EE, 'desc': lo - (MAX_FILTER - hi)*EE} with hi/lo the larger/smaller endpoint value
    (accelerated_PD.py:12-22).  Keys for all edges are evaluated in one numpy pass: elementwise float64 add/mul round
    exactly like CPython's (two roundings, no FMA), the association is the reference's."""
    nodes = list(nodes)
    values = list(values)
    sf = {nd: {'old': v, 'new': v} for nd, v in zip(nodes, values)}
    edges = [(e[0], e[1]) for e in edges]
    if edges:
        slot = {nd: i for i, nd in enumerate(nodes)}
        f = np.asarray(values, dtype=np.float64)
        fa = f[[slot[a] for a, _ in edges]]
        fb = f[[slot[b] for _, b in edges]]
        hi, lo = np.maximum(fa, fb), np.minimum(fa, fb)
        asc = hi + (lo + 1) * EE
        desc = lo - (MAX_FILTER - hi) * EE
        for e, up, down in zip(edges, asc.tolist(), desc.tolist()):
            sf[e] = {'asc': up, 'desc': down}
    return sf


def perturb_filter_function(g, descriptor='seal'):
    """:6-23.  g: networkx-like graph whose nodes carry g.nodes[n][descriptor]."""
    nodes = list(g.nodes())
    return build_simplex_filter(nodes, [g.nodes[nd][descriptor] for nd in nodes], g.edges())


def _unpack(simplex_filter):
    nodes = [s for s in simplex_filter if not isinstance(s, tuple)]
    edges = [s for s in simplex_filter if isinstance(s, tuple)]
    local = {nd: i for i, nd in enumerate(nodes)}
    f = np.array([simplex_filter[nd]['old'] for nd in nodes], dtype=np.float64)
    e = np.array([[local[a], local[b]] for a, b in edges], dtype=np.int32).reshape(-1, 2)
    return nodes, edges, f, e


def _run(simplex_filter, flags):
    import torch
    nodes, edges, f, e = _unpack(simplex_filter)
    if len(nodes) > 65535:
        raise ValueError("graphs with more than 65535 nodes are not supported by the HIP PD kernel")
    dev = "cuda"
    r = engine.pd_from_filtration(torch.tensor([0, len(nodes)], dtype=torch.int64, device=dev),
                                  torch.tensor([0, len(edges)], dtype=torch.int64, device=dev),
                                  torch.from_numpy(e).to(dev), torch.from_numpy(f).to(dev), flags)
    c = r["counts"][0].cpu().numpy()
    out = dict(up=r["up"][:c[0]].cpu().numpy(), down=r["down"][:c[1]].cpu().numpy(), one=r["one"][:c[2]].cpu().numpy(),
               ext0=r["ext0"][0].cpu().numpy(), rank=r["edge_rank"].cpu().numpy(), edges=edges)
    return out


def _pos_neg(res):
    rank, edges = res["rank"], res["edges"]
    pos = sorted([(int(rank[i]), i) for i in range(len(edges)) if rank[i] >= 0])
    neg = sorted([(-int(rank[i]) - 1, i) for i in range(len(edges)) if rank[i] < 0])
    return [[edges[i][0], edges[i][1]] for _, i in pos], [[edges[i][0], edges[i][1]] for _, i in neg]


def Union_find(simplex_filter):
    """:26-113 -> (PD, Pos_edges, Neg_edges); PD = PD_up + [[min,max]] + PD_down + [[max,min]] (:110)."""
    res = _run(simplex_filter, 0)
    mn, mx = float(res["ext0"][0]), float(res["ext0"][1])
    PD = res["up"].tolist() + [[mn, mx]] + res["down"].tolist() + [[mx, mn]]
    Pos_edges, Neg_edges = _pos_neg(res)
    return PD, Pos_edges, Neg_edges


def check_split(res, Pos_edges, Neg_edges):
    """The cycle swap runs on the device from the filtration alone: it re-derives the tree (Neg) and the query order (Pos)
    that Union_find reported.  A caller that hands in anything else (edited lists, lists of another filtration) would get
    the diagram of the unedited split -- refuse instead of answering a different question."""
    pos, neg = _pos_neg(res)
    as_sets = lambda edges: [frozenset((a, b)) for a, b in edges]
    if as_sets(Pos_edges) != as_sets(pos):
        raise ValueError("Accelerate_PD: Pos_edges is not the cycle-edge sequence Union_find yields for this simplex_filter "
                         "(the HIP kernel derives the split itself; edited lists are not supported)")
    if set(as_sets(Neg_edges)) != set(as_sets(neg)):
        raise ValueError("Accelerate_PD: Neg_edges is not the spanning tree Union_find yields for this simplex_filter")


def Accelerate_PD(Pos_edges, Neg_edges, simplex_filter):
    """:115-178 -> PD_one (list of [low, large]).  Pos/Neg must be what Union_find returned for the same
    simplex_filter (checked: ValueError otherwise)."""
    if len(Neg_edges) == 0:
        raise IndexError("list index out of range")          # list(Nodes)[0] on an empty graph (:122)
    res = _run(simplex_filter, 0)
    check_split(res, Pos_edges, Neg_edges)
    return res["one"].tolist()

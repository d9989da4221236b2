"""Drop-in for the reference's sg2dgm/accelerated_PD.py (TLC-GNN fork: zero-persistence pairs dropped).

  perturb_filter_function :6-23, Union_find :26-113, Accelerate_PD :115-178.

The simplex_filter dict keeps the reference's exact structure (it is the interface between the three functions);
the union-find passes and the cycle swap run in `tlc_pd_from_filtration` (include/tlcgnn.h).
"""
import numpy as np

from .. import engine, _lib


def perturb_filter_function(g, descriptor='seal'):
    """:6-23.  g: networkx-like graph whose nodes carry g.nodes[n][descriptor]."""
    simplex_filter = {}
    ee = 1e-6
    max_filter = 101
    for node in g.nodes():
        temp = {}
        temp['old'] = g.nodes[node][descriptor]
        temp['new'] = g.nodes[node][descriptor]
        simplex_filter[node] = temp
    for edge in g.edges():
        temp = {}
        max_node, min_node = max(simplex_filter[edge[0]]['old'], simplex_filter[edge[1]]['old']), min(
            simplex_filter[edge[0]]['old'], simplex_filter[edge[1]]['old'])
        temp['asc'] = max_node + (min_node + 1) * ee
        temp['desc'] = min_node - (max_filter - max_node) * ee
        simplex_filter[(edge[0], edge[1])] = temp
    return simplex_filter


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


def Accelerate_PD(Pos_edges, Neg_edges, simplex_filter):
    """:115-178 -> PD_one (list of [low, large]).  Pos/Neg must come from Union_find on the same simplex_filter."""
    if len(Neg_edges) == 0:
        raise IndexError("list index out of range")          # list(Nodes)[0] on an empty graph (:122)
    res = _run(simplex_filter, 0)
    return res["one"].tolist()

"""Drop-in for the PDGNN fork Knowledge_Distillation/accelerated_PD.py: array-valued filtration, zero-persistence
pairs KEPT, pieces returned separately (:6,12-13,68-69,108-109,118,169-170,183).  Same HIP kernel as the TLC-GNN
fork with the TLC_KEEP_ZERO_PERS flag."""
import numpy as np

from .. import _lib
from ..sg2dgm.accelerated_PD import _run, _pos_neg


def perturb_filter_function(g, filtration_val):
    """:6-22.  filtration_val[node] for every node of g."""
    simplex_filter = {}
    ee = 1e-6
    max_filter = 101
    for node in g.nodes():
        temp = {}
        temp['old'] = filtration_val[node]
        temp['new'] = filtration_val[node]
        simplex_filter[node] = temp
    for edge in g.edges():
        temp = {}
        max_node, min_node = max(simplex_filter[edge[0]]['old'], simplex_filter[edge[1]]['old']), min(
            simplex_filter[edge[0]]['old'], simplex_filter[edge[1]]['old'])
        temp['asc'] = max_node + (min_node + 1) * ee
        temp['desc'] = min_node - (max_filter - max_node) * ee
        simplex_filter[(edge[0], edge[1])] = temp
    return simplex_filter


def Union_find(simplex_filter):
    """:26-118 -> (Ord0, Ext0, Rel1, Pos_edges, Neg_edges) as numpy arrays / lists."""
    res = _run(simplex_filter, _lib.KEEP_ZERO_PERS)
    Pos_edges, Neg_edges = _pos_neg(res)
    return (np.array(res["up"]), np.array([[float(res["ext0"][0]), float(res["ext0"][1])]]), np.array(res["down"]),
            Pos_edges, Neg_edges)


def Accelerate_PD(Pos_edges, Neg_edges, simplex_filter):
    """:120-183 -> Ext1 as np.array([[low, large], ...])."""
    if len(Neg_edges) == 0:
        raise IndexError("list index out of range")
    res = _run(simplex_filter, _lib.KEEP_ZERO_PERS)
    return np.array(res["one"])

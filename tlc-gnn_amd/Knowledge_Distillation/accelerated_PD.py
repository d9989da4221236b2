"""Drop-in for the PDGNN fork Knowledge_Distillation/accelerated_PD.py: array-valued filtration, zero-persistence
pairs KEPT, pieces returned separately (:6,12-13,68-69,108-109,118,169-170,183).  Same HIP kernel as the TLC-GNN
fork with the TLC_KEEP_ZERO_PERS flag."""
import numpy as np

from .. import _lib
from ..sg2dgm.accelerated_PD import _run, _pos_neg, build_simplex_filter, check_split


def perturb_filter_function(g, filtration_val):
    """:6-22.  filtration_val[node] for every node of g (array or dict)."""
    nodes = list(g.nodes())
    return build_simplex_filter(nodes, [filtration_val[nd] for nd in nodes], g.edges())


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
    check_split(res, Pos_edges, Neg_edges)
    return np.array(res["one"])

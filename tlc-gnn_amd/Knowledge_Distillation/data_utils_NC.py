"""Drop-in for the vicinity / filtration part of the reference's Knowledge_Distillation/data_utils_NC.py (PDGNN, node
classification: one diagram per NODE), filt='ricci'.

  ricci_filtration.build_fv :27-50, compute_persistence_image :95-187 (modes 'filtration' and 'PI').

The node-centred variant of the vicinity kernels (SURVEY.md A.7 ii): the subgraph is the whole ball_hop(u) (:97-99), there is
ONE root, f(x) = weighted distance x -> u (sentinel 100 if unreachable, :41-42) divided by max + 1e-10 (:47-49).  On the
device that is the pair (u, u) -- ball(u) & ball(u) -- with TLC_DESC_ROOT1 | TLC_INCLUDE_ROOTS | TLC_NORM_EPS |
TLC_UNREACHABLE_100.  Node labels of a vicinity are positions in ASCENDING original id (the reference's
`convert_node_labels_to_integers` order is arbitrary); edges are listed once, lower label first.
filt='degree' / 'centrality' / 'clustering' (:124-135; the shipped training script's filtrations, train_Teacher_Model.py:158-159): the
same vicinities from the device, f from `data_utils_LP.structural_filtration` (host side, networkx's arithmetic: bit-equal values);
filt='hks' (:120-122): `hks_signature` with scipy's eigh, host side (values to rounding: the node order inside a vicinity differs).
`call` is not reproduced.
"""
import sys

import numpy as np

from .. import _lib
from .data_utils_LP import Vicinities, diagrams_and_images

KD_NC_FLAGS = _lib.INCLUDE_ROOTS | _lib.NORM_EPS | _lib.UNREACHABLE_100 | _lib.DESC_ROOT1


class NodeVicinities(Vicinities):
    """Device-resident weighted graph for PDGNN's node-centred vicinities; build once, query many nodes."""

    def batch(self, nodes, hop, node_cap=None, edge_cap=None, filt='ricci', hks_time=0.1):
        """nodes: [B] original labels -> the dict of Vicinities.batch (one vicinity per node).  filt: 'ricci', or 'degree' /
        'centrality' / 'clustering' (:124-135; `data_utils_LP.structural_filtration`)."""
        nodes = np.asarray(nodes, dtype=np.int64).reshape(-1)
        return super().batch(np.stack([nodes, nodes], 1), hop, node_cap=node_cap, edge_cap=edge_cap, flags=KD_NC_FLAGS, filt=filt, hks_time=hks_time)


_CACHE = {}


def _graph_stamp(g, ricci_curv):
    """Cheap content stamp of what the device copy was built from: node / edge counts, the length of the curvature list and its
    first and last 64 rows.  (The function is called once per node: a full hash of a 44 000-edge graph per call would cost
    more than the vicinities.)"""
    k = 0 if ricci_curv is None else len(ricci_curv)
    sample = () if not k else tuple(tuple(float(x) for x in row) for row in (list(ricci_curv[:64]) + list(ricci_curv[-64:])))
    if hasattr(g, "number_of_nodes"):
        counts = (g.number_of_nodes(), g.number_of_edges())
    else:                                                  # an edge array [m, 2] (what Vicinities also accepts)
        counts = tuple(np.asarray(g).shape)
    return counts + (k, hash(sample))


def _vicinities(g, ricci_curv):
    """One device graph is kept between calls (the reference rebuilds everything per call, :95-187).  It is reused only for
    the SAME graph and curvature objects (held here, so that their ids cannot be handed to other objects) whose stamp has not
    changed; a caller that edits a graph in place without changing any count should build `NodeVicinities(g, ricci_curv)`
    itself and call `.batch(nodes, hop)` -- which is also the faster way: many nodes per launch."""
    ent = _CACHE.get("entry")
    stamp = _graph_stamp(g, ricci_curv)
    if ent is None or ent[0] is not g or ent[1] is not ricci_curv or ent[2] != stamp:
        _CACHE["entry"] = ent = (g, ricci_curv, stamp, NodeVicinities(g, ricci_curv))
    return ent[3]


def compute_persistence_image(g, u, filt='hks', hks_time=0.1, hop=2, ricci_curv=None, mode='PI', num_models=5, max_loop_len=10,
                              cycle_the=2):
    """Reference signature (:95).  filt='hks' (the default), 'ricci', 'degree', 'centrality' or 'clustering' (the last two are what the shipped
    train_Teacher_Model.py:158-159 trains on); mode 'filtration' -> (filtration_val list, edge_index LongTensor[2,m]) or
    (None, None) for a ball without an edge (:103-104); mode 'PI' -> the reference's 9-tuple (:183; times are 0)."""
    from .data_utils_LP import STRUCTURAL_FILTS
    if filt != 'ricci' and filt not in STRUCTURAL_FILTS:
        print("Error: 'filt' should be 'hks', 'clustering',' centrality', 'degree' or 'ricci'! ")      # :154-155
        sys.exit()
    b = _vicinities(g, ricci_curv).batch([u], hop, filt=filt, hks_time=hks_time)
    if int(b["edge_ptr"][-1]) == 0:
        return None, None
    fv = b["f"].cpu().numpy()
    edge_index = b["edges"].t().contiguous().long().cpu()
    if mode == 'filtration':
        return fv.tolist(), edge_index
    if mode != 'PI':
        raise ValueError("mode must be 'PI' or 'filtration'")
    return diagrams_and_images(b, fv, edge_index)

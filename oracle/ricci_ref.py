"""CPU restatement of the curvature step before the path (test infrastructure only; never imported by the product).

PARITY UNPINNED.  loaddatas.py:105-123 calls the third-party `GraphRicciCurvature` (`OllivierRicci(Gd, alpha=0.5,
method="Sinkhorn")`), which calls POT's `ot.sinkhorn2(x, y, d, 1e-1, method='sinkhorn')`.  Neither package is in the
reference tree or its requirements.txt (no version pin) nor installable here, and the reference holds no golden values for the
curvature, so this file restates the published algorithms:

  GraphRicciCurvature OllivierRicci.py (0.5.x): `_get_single_node_neighbors_distributions` -- weights base**(-w**exp_power)
  with base=e, exp_power=2 over the (top-3000) neighbours, normalised to (1 - alpha), the node itself gets alpha;
  `_distribute_densities` -- cost = all-pairs shortest path lengths between the two supports (edge weight 1.0 when the graph
  has no "weight" attribute); `_compute_ricci_curvature_single_edge` -- kappa = 1 - m / weight(source, target).
  POT bregman.py `sinkhorn_knopp` (0.7-0.9): u, v start uniform; K = exp(-M / reg); loop { v = b / (K^T u); u = 1 / (Kp v) with
  Kp = K / a[:, None]; every 10th iteration err = || v * (K^T u) - b ||_2 }, until err <= stopThr (1e-9) or numItermax (1000);
  `sinkhorn2` returns sum(u[:, None] * K * v[None, :] * M).

Distances here come from scipy's BFS over the whole graph, NOT from the 0/1/2/3 shortcut of the HIP kernel.
"""
import numpy as np
import scipy.sparse as sp
from scipy.sparse.csgraph import shortest_path


def sinkhorn2(a, b, M, reg=1e-1, num_iter_max=1000, stop_thr=1e-9):
    a, b, M = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), np.asarray(M, dtype=np.float64)
    u = np.ones(len(a)) / len(a)
    v = np.ones(len(b)) / len(b)
    K = np.exp(M / (-reg))
    Kp = (1.0 / a).reshape(-1, 1) * K
    ii, err = 0, 1.0
    while err > stop_thr and ii < num_iter_max:
        uprev, vprev = u, v
        ktu = K.T @ u
        v = b / ktu
        u = 1.0 / (Kp @ v)
        if np.any(ktu == 0) or not (np.all(np.isfinite(u)) and np.all(np.isfinite(v))):
            u, v = uprev, vprev
            break
        if ii % 10 == 0:
            err = np.linalg.norm(np.einsum('i,ij,j->j', u, K, v) - b)
        ii += 1
    return float(np.sum(u[:, None] * K * v[None, :] * M)), ii


def ollivier_ricci_sinkhorn(n_nodes, edges, alpha=0.5, reg=1e-1, num_iter_max=1000, stop_thr=1e-9):
    """edges: undirected, loop-free [E,2].  Returns (kappa float64[E], iterations int[E]) in the order of `edges`."""
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    a = sp.coo_matrix((np.ones(len(edges)), (edges[:, 0], edges[:, 1])), shape=(n_nodes, n_nodes))
    adj = ((a + a.T) > 0).astype(np.float64).tocsr()
    dist = shortest_path(adj, method="D", unweighted=True)
    nbrs = [adj.indices[adj.indptr[i]:adj.indptr[i + 1]] for i in range(n_nodes)]

    def dens(node):
        nb = nbrs[node]
        if len(nb) == 0:
            return np.array([1.0]), np.array([node])
        w = np.full(len(nb), np.e ** (-(1.0 ** 2)))
        return np.concatenate([(1.0 - alpha) * w / w.sum(), [alpha]]), np.concatenate([nb, [node]])

    kappa, iters = np.zeros(len(edges)), np.zeros(len(edges), dtype=np.int64)
    for k, (s, t) in enumerate(edges.tolist()):
        if s == t:
            continue
        x, xs = dens(s)
        y, ys = dens(t)
        m, it = sinkhorn2(x, y, dist[np.ix_(xs, ys)], reg, num_iter_max, stop_thr)
        kappa[k], iters[k] = 1.0 - m / 1.0, it
    return kappa, iters

"""Synthetic inputs for the per-edge PD/PI + LP-forward path (numpy only, seeded).

There is no network on the build or GPU boxes, so the PubMed/Cora/Photo datasets the reference loads
(`/root/reference/loaddatas.py:15-24`) are replaced by graphs of the same shape:
preferential attachment with triad closure (Holme-Kim style, so that hop-1 vicinities are not empty),
topped up with uniform random edges to the exact edge count, plus a seeded stand-in for the
Ollivier-Ricci curvature the reference computes with a third-party package
(`loaddatas.py:105-123`); curvature is an *input* of the path, not part of it.
"""
import numpy as np

SHAPES = {
    # name: (N, M undirected, F features, hop, triad probability)
    "PubMed": (19717, 44324, 500, 2, 0.3),
    "Cora": (2708, 5278, 1433, 1, 0.3),
    "Photo": (7650, 119081, 745, 1, 0.6),
    "Computers": (13752, 245861, 767, 1, 0.6),
    "PPI": (2300, 30000, 50, 1, 0.5),
}


def holme_kim_edges(n_nodes, n_edges, triad_p=0.3, seed=1234):
    """Undirected simple graph with exactly ``n_edges`` edges on nodes 0..n_nodes-1.

    Returns int64[n_edges, 2] with u < v, sorted lexicographically.
    """
    rs = np.random.RandomState(seed)
    m_attach = max(1, min(n_edges // max(n_nodes, 1), 64))
    adj = [set() for _ in range(n_nodes)]
    # `targets` holds one entry per edge endpoint: sampling it uniformly is preferential attachment
    targets = []
    n0 = m_attach + 1
    for a in range(n0):
        for b in range(a + 1, n0):
            adj[a].add(b)
            adj[b].add(a)
            targets += [a, b]
    count = n0 * (n0 - 1) // 2
    for v in range(n0, n_nodes):
        prev = -1
        made = 0
        guard = 0
        while made < m_attach and guard < 8 * m_attach:
            guard += 1
            w = -1
            if prev >= 0 and rs.rand() < triad_p and adj[prev]:
                cand = list(adj[prev])
                w = cand[rs.randint(len(cand))]
            if w < 0 or w == v or w in adj[v]:
                w = targets[rs.randint(len(targets))]
            if w == v or w in adj[v]:
                continue
            adj[v].add(w)
            adj[w].add(v)
            targets += [v, w]
            prev = w
            made += 1
            count += 1
    # top up (or trim) to the exact edge count with uniform random edges
    while count < n_edges:
        a, b = rs.randint(n_nodes), rs.randint(n_nodes)
        if a == b or b in adj[a]:
            continue
        adj[a].add(b)
        adj[b].add(a)
        count += 1
    edges = np.array([(a, b) for a in range(n_nodes) for b in adj[a] if a < b], dtype=np.int64)
    if len(edges) > n_edges:
        keep = np.sort(rs.permutation(len(edges))[:n_edges])
        edges = edges[keep]
    order = np.lexsort((edges[:, 1], edges[:, 0]))
    return edges[order]


def synthetic_curvature(edges, seed=1234, lo=-0.5, hi=0.9):
    """Stand-in for `compute_ricci_curvature` (`loaddatas.py:105-123`): one kappa per undirected edge,
    kappa ~ U(lo, hi) so that the path's edge weights kappa+1 are strictly positive.

    Returns the reference's format: a sorted list of ``[u, v, kappa]`` with both directions present.
    """
    rs = np.random.RandomState(seed)
    kappa = rs.uniform(lo, hi, size=len(edges))
    out = []
    for (a, b), k in zip(edges.tolist(), kappa.tolist()):
        out.append([a, b, k])
        out.append([b, a, k])
    return sorted(out)


def curvature_array(edges, seed=1234, lo=-0.5, hi=0.9):
    """Same values as :func:`synthetic_curvature`, as float64[len(edges)] aligned with ``edges``."""
    rs = np.random.RandomState(seed)
    return rs.uniform(lo, hi, size=len(edges))


def edges_to_csr(n_nodes, edges, kappa=None):
    """Symmetric CSR (rowptr int32[n+1], col int32[2M], weight float64[2M] = kappa+1), columns ascending."""
    edges = np.asarray(edges, dtype=np.int64).reshape(-1, 2)
    src = np.concatenate([edges[:, 0], edges[:, 1]])
    dst = np.concatenate([edges[:, 1], edges[:, 0]])
    if kappa is None:
        w = np.ones(len(src), dtype=np.float64)
    else:
        kappa = np.asarray(kappa, dtype=np.float64)
        w = np.concatenate([kappa, kappa]) + 1.0
    order = np.lexsort((dst, src))
    src, dst, w = src[order], dst[order], w[order]
    rowptr = np.zeros(n_nodes + 1, dtype=np.int64)
    np.add.at(rowptr, src + 1, 1)
    rowptr = np.cumsum(rowptr)
    return rowptr.astype(np.int32), dst.astype(np.int32), np.ascontiguousarray(w)


def synthetic_features(n_nodes, n_feat, seed=1234, density=0.1):
    """fp32[n_nodes, n_feat], ``density`` non-zeros ~ U(0, 0.2) (bag-of-words-like)."""
    rs = np.random.RandomState(seed)
    x = rs.uniform(0.0, 0.2, size=(n_nodes, n_feat)).astype(np.float32)
    mask = rs.rand(n_nodes, n_feat) < density
    return x * mask


def shaped_graph(name="PubMed", seed=1234, scale=1.0):
    """(n_nodes, edges[M,2], kappa[M], hop, n_feat) for one of SHAPES, optionally scaled down."""
    n, m, f, hop, tp = SHAPES[name]
    n = max(8, int(n * scale))
    m = max(n, int(m * scale))
    edges = holme_kim_edges(n, m, triad_p=tp, seed=seed)
    kappa = curvature_array(edges, seed=seed)
    return n, edges, kappa, hop, f


def hiv_shaped_molecules(n_graphs=41127, seed=1234):
    """As many small sparse graphs as ogbg-molhiv holds (41 127, Knowledge_Distillation/data_utils_GC.py:284; config 5 of
    BASELINE.json): a random tree of ~25 nodes plus up to three extra edges each, degree filtration normalised per graph
    (data_utils_GC.py:118-121).  Returns (edges int32 [M,2] in graph-local ids, f float64 [N], node_offs, edge_offs int64)."""
    rs = np.random.RandomState(seed)
    ns = np.maximum(3, rs.poisson(25, size=n_graphs))
    edges, fs, node_offs, edge_offs = [], [], [0], [0]
    for n in ns:
        par = np.array([rs.randint(0, k) for k in range(1, n)])
        e = np.stack([par, np.arange(1, n)], 1)
        extra = rs.randint(0, n, size=(int(rs.randint(0, 4)), 2))
        extra = extra[extra[:, 0] != extra[:, 1]]
        e = np.unique(np.sort(np.concatenate([e, extra]), 1), axis=0)
        deg = np.bincount(e.ravel(), minlength=n).astype(np.float64)
        fs.append(deg / (deg.max() + 1e-10))
        edges.append(e)
        node_offs.append(node_offs[-1] + n)
        edge_offs.append(edge_offs[-1] + len(e))
    return (np.concatenate(edges).astype(np.int32), np.concatenate(fs), np.asarray(node_offs, dtype=np.int64),
            np.asarray(edge_offs, dtype=np.int64))

"""Drop-in for the path-facing part of the reference's loaddatas.py (/root/reference/loaddatas.py).

  get_edges_split :26-35, get_adj_split :38-54, compute_persistence_image :56-103, compute_ricci_curvature :105-123.

`compute_persistence_image` is the caller of the hot path: it fixes the order of the six
pair lists, builds the graph FROM EDGES ONLY (isolated nodes are "missing"), takes weights kappa+1 and calls
graph2pi(...).get_pimg_for_all_edges(hop, norm=True, extended_flag=True, resolution=5, descriptor='sum').
Dataset download (`loaddatas`) and the Ollivier-Ricci solver are out of scope: curvature is an INPUT here
(`data.ricci_list`, the reference's sorted [u, v, kappa] list).
"""
import os

import numpy as np
import scipy.sparse as sp

from .sg2dgm import riccidist2dgm as sg2dgm


def _edge_index_numpy(edge_index):
    try:
        import torch
        if isinstance(edge_index, torch.Tensor):
            return edge_index.detach().cpu().numpy()
    except ImportError:  # pragma: no cover
        pass
    return np.asarray(edge_index)


def get_edges_split(data, val_prop=0.2, test_prop=0.2, seed=1234):
    """loaddatas.py:26-35: adjacency of the graph on nodes 0..len(data.y)-1 -> get_adj_split."""
    n = len(data.y)
    ei = _edge_index_numpy(data.edge_index).astype(np.int64)
    # nx.Graph().add_edges_from + nx.adjacency_matrix: symmetric 0/1 matrix (a self loop is a single diagonal 1)
    a = sp.coo_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(n, n)).tocsr()
    a = ((a + a.T) > 0).astype(np.int64)
    return get_adj_split(sp.csr_matrix(a), val_prop=val_prop, test_prop=test_prop, seed=seed)


def _upper_positives(adj):
    """Stored entries of the upper triangle in CSR (row-major) order -- the order `sp.triu(adj).nonzero()` walks."""
    up = sp.triu(adj).tocoo()
    stored = up.data != 0
    return np.stack([up.row[stored], up.col[stored]], 1).astype(np.int64)


def _upper_non_edges(adj, block_cells=1 << 24):
    """[n_neg, 2] int64: every (x, y), y >= x, with adj[x, y] != 1, row-major -- what `sp.triu(1. - adj.toarray()).nonzero()`
    enumerates (loaddatas.py:44), built from boolean row blocks instead of the N x N float64 array."""
    n = adj.shape[0]
    cols = np.arange(n)
    rows_per_block = max(1, block_cells // max(n, 1))
    parts = []
    for r0 in range(0, n, rows_per_block):
        r1 = min(n, r0 + rows_per_block)
        free = cols[None, :] >= np.arange(r0, r1)[:, None]
        ones = adj[r0:r1].tocoo()
        is_one = ones.data == 1
        free[ones.row[is_one], ones.col[is_one]] = False
        x, y = np.nonzero(free)
        parts.append(np.stack([x + r0, y], 1))
    return np.concatenate(parts).astype(np.int64) if parts else np.zeros((0, 2), np.int64)


def _cut(pos_edges, negatives, val_prop, test_prop):
    """The three slices of both lists (loaddatas.py:48-52); `negatives` only needs slicing."""
    n_val, n_test = int(len(pos_edges) * val_prop), int(len(pos_edges) * test_prop)
    a, b = n_val, n_val + n_test
    return (pos_edges[b:], pos_edges[:a], pos_edges[a:b]), (negatives[:a], negatives[a:b])


def get_adj_split(adj, val_prop=0.05, test_prop=0.1, seed=1234):
    """loaddatas.py:38-54 with the reference's array outputs: the same two legacy-MT19937 shuffles in the same order
    (positives, then negatives), `train_edges_false` = the whole shuffled negative list ++ val ++ test positives (:53).
    The host-side twin of get_adj_split_streamed for plumbing-sized graphs: the negatives are enumerated by
    `_upper_non_edges` (n_neg x 2 int64 -- 3.1 GB for PubMed, which is what the streamed form is for)."""
    adj = sp.csr_matrix(adj)
    np.random.seed(seed)
    pos_edges = _upper_positives(adj)
    np.random.shuffle(pos_edges)
    neg_edges = _upper_non_edges(adj)
    np.random.shuffle(neg_edges)
    (train, val, test), (val_false, test_false) = _cut(pos_edges, neg_edges, val_prop, test_prop)
    return train, np.concatenate([neg_edges, val, test]), val, val_false, test, test_false


def get_edges_split_streamed(data, val_prop=0.2, test_prop=0.2, seed=1234, device=None):
    """get_edges_split (loaddatas.py:26-35) on top of get_adj_split_streamed."""
    n = len(data.y)
    ei = _edge_index_numpy(data.edge_index).astype(np.int64)
    a = sp.coo_matrix((np.ones(ei.shape[1]), (ei[0], ei[1])), shape=(n, n)).tocsr()
    a = ((a + a.T) > 0).astype(np.int64)
    return get_adj_split_streamed(sp.csr_matrix(a), val_prop=val_prop, test_prop=test_prop, seed=seed, device=device)


def get_adj_split_streamed(adj, val_prop=0.05, test_prop=0.1, seed=1234, device=None):
    """get_adj_split (loaddatas.py:38-54) without the dense complement: same RNG stream, same six lists, but the negative
    list stays on the device side as (seeded permutation, CSR enumeration) -- SURVEY.md 8(f) item 2.

    Returns (train_edges, negatives, val_edges, val_edges_false, test_edges, test_edges_false) where `negatives` is a
    pi_cache.ShuffledNegatives standing for the reference's shuffled `neg_edges`; the reference's
    `train_edges_false` is negatives ++ val_edges ++ test_edges (:53)."""
    from .pi_cache import ShuffledNegatives
    adj = sp.csr_matrix(adj)
    adj.sum_duplicates()
    adj.eliminate_zeros()
    adj.sort_indices()
    if adj.nnz and not np.all(adj.data == 1):
        raise ValueError("get_adj_split_streamed: the adjacency must be 0/1 (`1. - adj` of the reference is a non-edge "
                         "only where adj == 1)")
    np.random.seed(seed)
    pos_edges = _upper_positives(adj)
    np.random.shuffle(pos_edges)
    negatives = ShuffledNegatives(adj.indptr, adj.indices, device=device)     # draws the second shuffle
    (train, val, test), (val_false, test_false) = _cut(pos_edges, negatives, val_prop, test_prop)
    return train, negatives, val, val_false, test, test_false


def compute_persistence_image_streamed(data, train_edges, negatives, val_edges, val_edges_false, test_edges, test_edges_false,
                                       hop=1, chunk=1 << 22, keep_failed=False, prefilter=False):
    """compute_persistence_image (loaddatas.py:56-103) for the streamed split: the images of all six lists in the reference's
    order (:65-66) as a pi_cache.SparseImages, plus the LazyPairList standing for `total_edges`.
    data.edge_index must already have lost the val/test positives (TLCGNN.py:88-100)."""
    import torch
    from . import engine, synth
    from .pi_cache import LazyPairList, sweep_images, assemble
    tail = np.concatenate([val_edges, test_edges]).astype(np.int64)
    total = LazyPairList([(train_edges, 1), (negatives, 0), (tail, 0), (val_edges, 1), (val_edges_false, 0),
                          (test_edges, 1), (test_edges_false, 0)])
    ei = _edge_index_numpy(data.edge_index)
    ei = ei[:, ei[0] != ei[1]]
    und = np.unique(np.sort(ei.T, axis=1), axis=0)
    ricci = compute_ricci_curvature(data)
    kap = {(int(a), int(b)): float(k) for a, b, k in ricci}
    kappa = np.array([kap[(int(a), int(b))] for a, b in und.tolist()])
    n = int(max(len(data.y), und.max() + 1)) if len(und) else len(data.y)
    rowptr, col, w = synth.edges_to_csr(n, und, kappa)
    g = engine.DeviceGraph(rowptr, col, w)
    pieces = []
    dev = torch.device("cuda", g.device)
    if prefilter and not keep_failed:
        # the negative list through the distance <= hop pre-filter (pi_cache.sweep_near): its zero rows are known without running
        # them (their status bytes stay uncomputed: `unclassified`); the five short lists go through the pipeline as they are
        from .pi_cache import sweep_near, SparseImages
        lo_neg, hi_neg = int(total.bounds[1]), int(total.bounds[2])
        inv = np.empty(len(negatives), dtype=np.int64)
        inv[negatives.perm] = np.arange(len(negatives), dtype=np.int64)
        near = sweep_near(g, negatives.index, hop, positions=lambda r: inv[r] + lo_neg)
        sweep_images(g, lambda lo, hi: total.device_pairs(lo, hi, device=dev), lo_neg, hop, chunk=chunk, store=pieces)
        sweep_images(g, lambda lo, hi: total.device_pairs(hi_neg + lo, hi_neg + hi, device=dev), len(total) - hi_neg, hop,
                     chunk=chunk, index_base=hi_neg, store=pieces)
        g.close()
        rest = assemble(pieces, len(total), 25)
        out = SparseImages(len(total), 25, np.concatenate([rest.idx, near.idx]), np.concatenate([rest.rows, near.rows]),
                           np.concatenate([rest.status, near.status]), rest.status_counts + near.status_counts)
        out.unclassified = near.unclassified
        return out, total
    sweep_images(g, lambda lo, hi: total.device_pairs(lo, hi, device=dev), len(total), hop, chunk=chunk, store=pieces,
                 keep_failed=keep_failed)
    g.close()
    return assemble(pieces, len(total), 25), total


def compute_ricci_curvature(data):
    """loaddatas.py:105-123: Ollivier-Ricci curvature (alpha 0.5, Sinkhorn) of every edge of data.edge_index as the sorted list
    [[u, v, kappa], [v, u, kappa], ...].  The reference delegates to the third-party GraphRicciCurvature (absent here); this
    runs the same computation on the GPU (tlc_ollivier_ricci_sinkhorn; parity unpinned, see the checker's header in
    tests/).  A caller-supplied `data.ricci_list` (the reference's format) takes precedence -- e.g. curvature computed
    elsewhere, or the seeded stand-in of tlc_gnn_amd.synth.synthetic_curvature."""
    ricci = getattr(data, "ricci_list", None)
    if ricci is not None:
        return ricci
    from . import engine, synth
    ei = _edge_index_numpy(data.edge_index).astype(np.int64)
    ei = ei[:, ei[0] != ei[1]]                                   # the library drops self loops
    und = np.unique(np.sort(ei.T, axis=1), axis=0)
    n = int(max(len(data.y), und.max() + 1)) if len(und) else len(data.y)
    rowptr, col, _ = synth.edges_to_csr(n, und)
    # (source, target) as networkx's G.edges() yields it after add_edges_from(edge list): the endpoint that entered the graph
    # first is the source.  The Sinkhorn loop stops on the target marginal, so the orientation shows at the 1e-6 level.
    flat = ei.T.reshape(-1)
    first = np.full(n, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(first, flat, np.arange(len(flat)))
    swap = first[und[:, 0]] > first[und[:, 1]]
    oriented = np.where(swap[:, None], und[:, ::-1], und)
    kappa = engine.ollivier_ricci_sinkhorn(rowptr, col, oriented, alpha=0.5)
    ricci_list = []
    for (n1, n2), k in zip(oriented.tolist(), kappa.tolist()):
        ricci_list.append([n1, n2, k])
        ricci_list.append([n2, n1, k])
    return sorted(ricci_list)


def compute_persistence_image(data, train_edges, train_edges_false, val_edges, val_edges_false, test_edges,
                              test_edges_false, data_name, hop=1, cache_dir='./data/TLCGNN'):
    """loaddatas.py:56-103.  Returns float64 [n_pairs, 25] (numpy, like the reference)."""
    import torch
    if data_name == "photo":
        data_name = "Photo"
    if data_name == "computers":
        data_name = "Computers"

    filename = os.path.join(cache_dir, data_name + '.npy') if cache_dir else None
    if filename and os.path.exists(filename):
        return np.load(filename)
    total_edges = np.concatenate(
        (train_edges, train_edges_false, val_edges, val_edges_false, test_edges, test_edges_false))
    data.train_pos, data.train_neg = len(train_edges), len(train_edges_false)
    data.val_pos, data.val_neg = len(val_edges), len(val_edges_false)
    data.test_pos, data.test_neg = len(test_edges), len(test_edges_false)
    data.total_edges = total_edges

    # delete val_pos and test_pos (:73-86) -- a no-op when TLCGNN.call already did it
    from .baselines.TLCGNN import remove_pairs_both_directions
    data.edge_index = remove_pairs_both_directions(data.edge_index, np.concatenate((val_edges, test_edges)))
    ei = data.edge_index.cpu().numpy()
    ei = ei[:, ei[0] != ei[1]]                                   # remove_self_loops (:86,90)
    data.edge_index = torch.from_numpy(ei).long()

    # generate graph for computing persistence diagram: edges only, so isolated nodes do not exist (:88-92)
    edges = ei.T
    print(len(set(map(tuple, np.sort(edges, axis=1).tolist()))))  # len(g.edges()) (:93)

    ricci_cur = compute_ricci_curvature(data)

    # compute sg2dgm and save in a dict
    pi = sg2dgm.graph2pi(edges, ricci_curv=ricci_cur)
    pi.get_pimg_for_all_edges(total_edges, cores=16, hop=hop, norm=True, extended_flag=True,
                              resolution=5, descriptor='sum')
    if filename:
        os.makedirs(os.path.dirname(filename), exist_ok=True)    # the reference never creates it and np.save fails
        np.save(filename, pi.pi_sg)
    return pi.pi_sg

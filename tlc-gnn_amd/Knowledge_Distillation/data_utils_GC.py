"""Drop-in for the ground-truth generation of the reference's Knowledge_Distillation/data_utils_GC.py (PDGNN, graph
classification: whole graph = one diagram) and for the forward-only loop of train_Teacher_Model_GC.evaluate_time.

  compute_persistence_image :98-170 (filt='degree' and 'hks': node functions on the host, the reference's own numpy / scipy calls),
  original_extended_persistence :78-82, call :228-279 (largest connected component, relabel), evaluate_time :118-143.

The extended persistence of every graph runs in `tlc_pd_from_filtration` (Knowledge_Distillation fork: zero-persistence pairs
kept), the three images (Ord0 ++ Ext1, Ord0, Ext1; :155-163) in `tlc_pi_raster`; `*_batch` processes a whole dataset in one
launch of each instead of the reference's per-graph Python loop.
"""
import numpy as np

from .. import engine, _lib


def _edges_nodes(g):
    if hasattr(g, "edges") and callable(getattr(g, "edges")):
        nodes = list(g.nodes())
        e = np.array([(a, b) for a, b in g.edges()], dtype=np.int64).reshape(-1, 2)
        return len(nodes), e
    n, e = g
    return int(n), np.asarray(e, dtype=np.int64).reshape(-1, 2)


def degree_filtration(n, edges):
    """:117-119  degree / (max degree + 1e-10), in fp64 like the reference's Python floats."""
    deg = np.bincount(edges.reshape(-1), minlength=n).astype(np.float64)
    return deg / (deg.max() + 1e-10)


def _connected(n, edges):
    parent = list(range(n))

    def find(x):
        while parent[x] != x:
            parent[x] = parent[parent[x]]
            x = parent[x]
        return x
    for a, b in edges.tolist():
        ra, rb = find(a), find(b)
        if ra != rb:
            parent[ra] = rb
    return len({find(i) for i in range(n)}) == 1


def hks_filtration(n, edges, hks_time):
    """:114-116  hks_signature / (max + 1e-10): the reference's scipy calls on the same matrix (nodes 0..n-1), host side."""
    from .data_utils_LP import hks_signature
    v = hks_signature(n, edges, hks_time)
    return v / (max(v) + 1e-10)


def compute_persistence_image_batch(graphs, filt='degree', filtrations=None, hks_time=0.1):
    """graphs: list of networkx-like graphs with nodes 0..n-1, or (n, edges[m,2]) tuples.
    Returns a list with the reference's 9-tuple per graph (:166), or (None, None) for graphs without an edge / not
    connected (:101-103).  filt: 'degree' or 'hks' (host side, :114-119); `filtrations` supplies f per graph for anything else."""
    import torch
    if filt not in ('degree', 'hks') and filtrations is None:
        raise NotImplementedError("data_utils_GC (HIP): filt='degree' and 'hks' are computed here; pass `filtrations` for anything else")
    parsed, keep = [], []
    for gi, g in enumerate(graphs):
        n, e = _edges_nodes(g)
        ok = len(e) > 0 and _connected(n, e)
        parsed.append((n, e))
        if ok:
            keep.append(gi)
    out = [(None, None)] * len(graphs)
    if not keep:
        return out
    fs = [np.asarray(filtrations[gi], dtype=np.float64) if filtrations is not None else
          (hks_filtration(parsed[gi][0], parsed[gi][1], hks_time) if filt == 'hks' else degree_filtration(*parsed[gi])) for gi in keep]
    node_offs = np.concatenate([[0], np.cumsum([parsed[gi][0] for gi in keep])]).astype(np.int64)
    edge_offs = np.concatenate([[0], np.cumsum([len(parsed[gi][1]) for gi in keep])]).astype(np.int64)
    edges = np.concatenate([parsed[gi][1] for gi in keep]).astype(np.int32)
    f = np.concatenate(fs)
    dev = "cuda"
    r = engine.pd_from_filtration(torch.from_numpy(node_offs).to(dev), torch.from_numpy(edge_offs).to(dev),
                                  torch.from_numpy(edges).to(dev), torch.from_numpy(f).to(dev), _lib.KEEP_ZERO_PERS, want_rank=False)
    counts = r["counts"].cpu().numpy()
    up, one = r["up"], r["one"]
    # gather the ragged diagrams: Ord0 of graph k = up[node_offs[k] : +counts[k,0]], Ext1 = one[edge_offs[k] : +counts[k,2]]
    idx0 = torch.cat([torch.arange(int(node_offs[k]), int(node_offs[k]) + int(counts[k, 0]), device=dev) for k in range(len(keep))])
    idx1 = torch.cat([torch.arange(int(edge_offs[k]), int(edge_offs[k]) + int(counts[k, 2]), device=dev) for k in range(len(keep))])
    p0, p1 = up[idx0], one[idx1]
    o0 = np.concatenate([[0], np.cumsum(counts[:, 0])]).astype(np.int64)
    o1 = np.concatenate([[0], np.cumsum(counts[:, 2])]).astype(np.int64)
    # Ord0 ++ Ext1 per graph (:163)
    both_idx, ob = [], [0]
    for k in range(len(keep)):
        both_idx.append(torch.arange(o0[k], o0[k + 1], device=dev))
        both_idx.append(len(p0) + torch.arange(o1[k], o1[k + 1], device=dev))
        ob.append(ob[-1] + int(counts[k, 0] + counts[k, 2]))
    pall = torch.cat((p0, p1))[torch.cat(both_idx)]
    to = lambda a: torch.from_numpy(np.asarray(a, dtype=np.int64)).to(dev)
    img = engine.pi_raster(to(ob), pall, 5).cpu().numpy()
    img0 = engine.pi_raster(to(o0), p0, 5).cpu().numpy()
    img1 = engine.pi_raster(to(o1), p1, 5).cpu().numpy()
    p0n, p1n = p0.cpu().numpy(), p1.cpu().numpy()
    for k, gi in enumerate(keep):
        n, e = parsed[gi]
        d0, d1 = p0n[o0[k]:o0[k + 1]], p1n[o1[k]:o1[k + 1]]
        PI0 = img0[k] if len(d0) else np.zeros(25)
        PI1 = img1[k] if len(d1) else np.zeros(25)
        pers_img = PI1 if len(d0) == 0 else (PI0 if len(d1) == 0 else img[k])
        edge_index = torch.from_numpy(e.T.copy()).long()
        out[gi] = (d0, d1, pers_img, fs[k].tolist(), edge_index, PI0, PI1, 0.0, 0.0)
    return out


def compute_persistence_image(g, filt='hks', hks_time=0.1, hop=2, ricci_curv=None, mode='PI', num_models=5, max_loop_len=10,
                              cycle_the=2):
    """Reference signature (:98).  filt='hks' (the default) or 'degree' ('ricci' needs curvatures per graph: pass `filtrations`
    to compute_persistence_image_batch); mode 'PI' -> 9-tuple, 'filtration' -> (filtration_val, edge_index)."""
    import torch
    if filt not in ('degree', 'hks'):
        raise NotImplementedError("data_utils_GC (HIP): filt='hks' and 'degree' are implemented; for 'ricci' pass the values as "
                                  "`filtrations` to compute_persistence_image_batch")
    n, e = _edges_nodes(g)
    if len(e) == 0 or not _connected(n, e):
        return None, None
    if mode == 'filtration':
        f = hks_filtration(n, e, hks_time) if filt == 'hks' else degree_filtration(n, e)
        return f.tolist(), torch.from_numpy(e.T.copy()).long()
    return compute_persistence_image_batch([(n, e)], filt=filt, hks_time=hks_time)[0]


def evaluate_batch(model, samples):
    """The forward-only loop of train_Teacher_Model_GC.evaluate_time (:118-143) as ONE block-diagonal PDGNN forward.
    samples: list of 9-tuples / (None, None) as produced above; returns float32 CUDA [n_kept, 25] images + kept indices."""
    import torch
    xs, eis, gptr, eptr, kept = [], [], [0], [0], []
    for si, data in enumerate(samples):
        if len(data) <= 2:                       # :127-128
            continue
        f, ei = np.asarray(data[3], dtype=np.float32), data[4]
        ei = ei[:, ei[0] != ei[1]]               # remove_self_loops (:132)
        eis.append(ei + gptr[-1])
        xs.append(f)
        gptr.append(gptr[-1] + len(f))
        eptr.append(eptr[-1] + ei.shape[1])
        kept.append(si)
    if not kept:
        return torch.zeros(0, 25, device="cuda"), kept
    n = gptr[-1]
    loops = torch.arange(n)
    edge_index = torch.cat([torch.cat(eis, dim=1), torch.stack([loops, loops])], dim=1).cuda()       # add_self_loops (:133)
    x = torch.from_numpy(np.concatenate(xs)).view(-1, 1).cuda()
    with torch.no_grad():
        _, img, *_ = model(x, edge_index, None, compute_loss=False, grad_PI=False, graph_ptr=torch.tensor(gptr).cuda(),
                           edge_ptr=torch.tensor(eptr).cuda())
    return img, kept

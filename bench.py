#!/usr/bin/env python3
"""bench.py -- persistence-images/sec + LP-forward edges/sec on the PubMed-shaped synthetic graph (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W            (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W   (N>1)

One "step" = one pass of the hot path over one batch per GPU:
  leg 1  PD/PI : all 37 676 train-positive pairs (hop 2) -> 5x5 persistence images        (tlc_pd_pi_batch)
  leg 2  LP fwd: TLCGNN encode (2-layer GCN) + fused decode over 2*37 676 = 75 352 pairs, image rows resident in HBM
Inputs are resident in HBM before the timed region.  Weak scaling: every rank owns a full batch (its own permutation of
the training pairs, the small CSR replicated, no data-path collective for leg 1; the encoder of leg 2 is node-row
sharded with one RCCL all-gather per layer).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense f32-input MFMA peak (same guide: 64 FLOP/clk/SIMD)


def build_workload(rank, seed=1234):
    """PubMed-shaped graph, the reference's split of the positives (loaddatas.py:38-53: same RNG stream up to the
    shuffle of the positives), training graph = graph minus val/test positives (TLCGNN.py:88-100)."""
    from tlc_gnn_amd import synth
    n, edges, kappa, hop, n_feat = synth.shaped_graph("PubMed", seed=seed)
    np.random.seed(seed)
    order = np.lexsort((edges[:, 1], edges[:, 0]))          # sp.triu(adj).nonzero(): row-major order of (x, y), x < y
    pos = edges[order].copy()
    np.random.shuffle(pos)
    m_pos = len(pos)
    n_val, n_test = int(m_pos * 0.05), int(m_pos * 0.1)
    train = pos[n_val + n_test:]
    # training graph: curvature is an input of the path; seeded stand-in on the surviving edges
    key = {(int(a), int(b)): float(k) for (a, b), k in zip(edges.tolist(), kappa.tolist())}
    tr_sorted = train[np.lexsort((train[:, 1], train[:, 0]))]
    tr_kappa = np.array([key[(int(a), int(b))] for a, b in tr_sorted.tolist()])
    rowptr, col, w = synth.edges_to_csr(n, tr_sorted, tr_kappa)
    rs = np.random.RandomState(seed + 1000 * rank)
    pi_pairs = train.copy()
    if rank > 0:                                             # weak scaling: same work, different order/orientation
        pi_pairs = pi_pairs[rs.permutation(len(pi_pairs))]
        pi_pairs = pi_pairs[:, ::-1] if rank % 2 else pi_pairs
    # negatives for the decode: uniformly sampled non-adjacent pairs, as many as positives (TLCGNN.py:29-32)
    adj = set(map(tuple, tr_sorted.tolist()))
    neg = []
    while len(neg) < len(train):
        a, b = rs.randint(0, n, size=2 * len(train)), rs.randint(0, n, size=2 * len(train))
        for x, y in zip(a.tolist(), b.tolist()):
            if x != y and (min(x, y), max(x, y)) not in adj:
                neg.append((x, y))
                if len(neg) == len(train):
                    break
    neg = np.array(neg, dtype=np.int64)
    x = synth.synthetic_features(n, n_feat, seed=seed)
    return dict(n=n, hop=hop, n_feat=n_feat, rowptr=rowptr, col=col, w=w, train_edges=tr_sorted,
                pi_pairs=np.ascontiguousarray(pi_pairs, dtype=np.int32), neg=neg, x=x)


def pdgnn_aux(torch, dev, n_graphs=41127, seed=1234):
    """PDGNN forward vs exact PD on HIV-shaped molecules -- as many graphs as ogbg-molhiv holds (41 127, data_utils_GC.py:284; config 5
    of BASELINE.json): graphs/s of each (device-resident inputs, median of 5)."""
    from tlc_gnn_amd import engine
    from tlc_gnn_amd.Knowledge_Distillation.Teacher_model import Teacher_Model
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
    f = np.concatenate(fs)
    e_all = np.concatenate(edges).astype(np.int32)
    d_no = torch.tensor(node_offs, dtype=torch.int64, device=dev)
    d_eo = torch.tensor(edge_offs, dtype=torch.int64, device=dev)
    d_e = torch.from_numpy(e_all).to(dev)
    d_f = torch.from_numpy(f).to(dev)

    def med_ms(fn, reps=5):
        fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3

    exact_ms = med_ms(lambda: engine.pd_from_filtration(d_no, d_eo, d_e, d_f, 0, want_rank=False))
    # block-diagonal PDGNN batch: both directions of every edge, self loops appended last (train_Teacher_Model.py:43-44)
    glob = np.concatenate([edges[k] + node_offs[k] for k in range(n_graphs)])
    both = np.concatenate([glob, glob[:, ::-1]])
    order = np.argsort(np.searchsorted(np.asarray(node_offs[1:]), both[:, 0], side="right"), kind="stable")
    both = both[order]
    eptr = np.concatenate([[0], np.cumsum(2 * np.diff(edge_offs))]).astype(np.int64)
    n_tot = node_offs[-1]
    loops = np.arange(n_tot)
    ei = torch.from_numpy(np.concatenate([both, np.stack([loops, loops], 1)]).T.copy()).to(dev)
    x = torch.from_numpy(f.astype(np.float32)).view(-1, 1).to(dev)
    torch.manual_seed(seed)
    model = Teacher_Model(type='GAT').eval().to(dev)
    gptr = torch.tensor(node_offs, dtype=torch.int64, device=dev)
    d_eptr = torch.from_numpy(eptr).to(dev)
    with torch.no_grad():
        pd_ms = med_ms(lambda: model(x, ei, None, compute_loss=False, grad_PI=False, graph_ptr=gptr, edge_ptr=d_eptr))
    return {"graphs": int(n_graphs), "nodes": int(n_tot), "edges": int(len(e_all)),
            "pdgnn_forward_graphs_per_sec": n_graphs / (pd_ms * 1e-3), "pdgnn_forward_ms": pd_ms,
            "exact_pd_graphs_per_sec": n_graphs / (exact_ms * 1e-3), "exact_pd_ms": exact_ms,
            "note": "HIV-shaped synthetic molecules in one block-diagonal batch; PDGNN = 4 GAT layers + edge head + 5x5 image "
                    "(random-init weights, seed 1234), exact = tlc_pd_from_filtration on the same graphs; host wall clock around "
                    "device-resident calls; not part of `value`"}


def _ricci_cpu_sample(ricci_ref, n, edges, sample):
    """CPU restatement (numpy Sinkhorn of oracle/ricci_ref.py) for a sample of edges; hop distances by the 0/1/2/3 rule from
    adjacency sets (an all-pairs BFS of the 19 717-node graph would dominate the timing)."""
    nb = [[] for _ in range(n)]
    for a, b in edges.tolist():
        nb[a].append(b)
        nb[b].append(a)
    sets = [set(x) for x in nb]
    out = []
    for s, t in sample.tolist():
        xs, ys = nb[s] + [s], nb[t] + [t]
        M = np.empty((len(xs), len(ys)))
        for i, a in enumerate(xs):
            for j, b in enumerate(ys):
                M[i, j] = 0 if a == b else (1 if b in sets[a] else (2 if sets[a] & sets[b] else 3))
        x = np.concatenate([np.full(len(xs) - 1, 0.5 / (len(xs) - 1)), [0.5]])
        y = np.concatenate([np.full(len(ys) - 1, 0.5 / (len(ys) - 1)), [0.5]])
        out.append(1.0 - ricci_ref.sinkhorn2(x, y, M)[0])
    return np.array(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the auxiliary random-pair sweep (profiling runs)")
    ap.add_argument("--dist-backend", default="nccl", help="rehearsal only: 'gloo' lets N ranks share ONE GPU with --single-device "
                    "(RCCL refuses two ranks on a device); the measured configuration is always nccl = RCCL, one rank per GPU")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="development: no per-kernel HIP events in the timed region (the roofline block is then meaningless)")
    args = ap.parse_args()

    import torch
    from tlc_gnn_amd import engine, ops, dist as tdist, _lib
    from tlc_gnn_amd.baselines import TLCGNN

    rank, local_rank, world = tdist.env_world()
    if args.gpus != world and world == 1 and args.gpus > 1:
        raise SystemExit("--gpus %d needs torch.distributed.run with --nproc-per-node %d" % (args.gpus, args.gpus))
    _lib.require_gpu()
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)

    wl = build_workload(rank)
    n, hop = wl["n"], wl["hop"]
    g = engine.DeviceGraph(wl["rowptr"], wl["col"], wl["w"], device=local_rank)
    pi_pairs = torch.from_numpy(wl["pi_pairs"]).to(dev)
    E = pi_pairs.shape[0]
    pi_out = torch.empty((E, 25), dtype=torch.float64, device=dev)
    pi_status = torch.empty(E, dtype=torch.uint8, device=dev)

    # ---- LP leg setup: model, graph operator, decode tables (all resident before the timed region) ---------------------
    torch.manual_seed(1234)
    model = TLCGNN.Net(None, wl["n_feat"], 2, PI=None)
    for mod in (model.linear, model.linear_1):                       # weights_init of pipelines.py:42-46 (xavier on Linear)
        torch.nn.init.xavier_uniform_(mod.weight)
        torch.nn.init.zeros_(mod.bias)
    model = model.to(dev).eval()
    te = wl["train_edges"]
    edge_index = torch.from_numpy(np.concatenate([te, te[:, ::-1]]).T.copy()).long().to(dev)
    rowptr_n, col_n, val_n = ops.gcn_norm_csr(edge_index, n)          # cached=True: one-off
    enc = tdist.ShardedGCNEncoder(rowptr_n, col_n, val_n, n, world, rank,
                                  gemm=lambda a, b: ops.gemm(a, b),
                                  spmm=lambda rp, c, v, xx, bias, relu, renorm=False: ops.spmm(rp, c, v, xx, bias=bias, relu=relu, renorm=renorm))
    x_local = torch.from_numpy(wl["x"][enc.lo:enc.hi]).to(dev).contiguous()
    dec_pairs_np = np.concatenate([wl["pi_pairs"].astype(np.int64), wl["neg"]]).astype(np.int32)
    dec_pairs = torch.from_numpy(dec_pairs_np).to(dev)
    dec_pi, _ = g.pd_pi_batch(dec_pairs, hop)                         # image rows of the decode batch: resident
    w1, b1 = model.conv1.weight.detach(), model.conv1.bias.detach()
    w2, b2 = model.conv2.weight.detach(), model.conv2.bias.detach()
    l1w, l1b = model.linear_1.weight.detach(), model.linear_1.bias.detach()
    l2w, l2b = model.linear.weight.detach(), model.linear.bias.detach()
    prob = torch.empty(dec_pairs.shape[0], dtype=torch.float32, device=dev)

    def leg_pi():
        g.pd_pi_batch(pi_pairs, hop, out=pi_out, status=pi_status)

    def leg_lp():
        emb = enc.encode(x_local, w1, b1, w2, b2, renorm=True)         # renorm_ of TLCGNN.py:48 fused into the last SpMM
        ops.lp_decode(dec_pairs, emb, dec_pi, l1w, l1b, l2w, l2b, out=prob)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # Per-kernel HIP events (recorded inside the library on the stream each kernel runs on) cost time themselves: all eight
    # pairs add 48 us to the 1.06 ms batch.  So every kernel is timed during the warm-up steps (-> `kernel_ms` of the others,
    # and which kernel dominates), and inside the timed region only the dominant kernel carries events (-> `roofline`).
    ktimes = {k: [] for k in engine.DeviceGraph.KERNELS}
    g.set_timing(not args.no_kernel_events)
    for _ in range(max(args.warmup, 1)):
        leg_pi()
        leg_lp()
        for k, v in g.timings().items():
            ktimes[k].append(v)
    warm_avg = {k: float(np.mean([x for x in v if x >= 0])) if any(x >= 0 for x in v) else -1.0 for k, v in ktimes.items()}
    dom_warm = max(warm_avg, key=lambda k: warm_avg[k])
    if not args.no_kernel_events:
        g.set_timing(True, only=[dom_warm])
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    ktimes_timed = []
    barrier()
    t0 = time.perf_counter()
    for s in range(args.steps):
        ev[s][0].record()
        leg_pi()
        ev[s][1].record()
        leg_lp()
        ev[s][2].record()
        ktimes_timed.append(g.timings()[dom_warm])   # HIP events on the stream the kernel ran on (synchronises: part of the step)
    barrier()
    wall = time.perf_counter() - t0
    t_pi = sum(ev[s][0].elapsed_time(ev[s][1]) for s in range(args.steps)) * 1e-3
    t_lp = sum(ev[s][1].elapsed_time(ev[s][2]) for s in range(args.steps)) * 1e-3
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([wall, t_pi, t_lp], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, t_pi, t_lp = [float(v) for v in t.tolist()]

    # auxiliary (untimed region, single pass): the negative sweep of loaddatas.py:44-53 is dominated by pairs with
    # d(u,v) > hop; report the throughput on 2^20 uniformly random pairs (PI-C of SURVEY.md 8d) next to the headline
    sweep = None
    if rank == 0 and not args.no_sweep:
        g.set_timing(False)
        rsw = np.random.RandomState(99)
        sw_pairs = torch.from_numpy(rsw.randint(0, n, size=(1 << 20, 2)).astype(np.int32)).to(dev)
        sw_out = torch.empty((1 << 20, 25), dtype=torch.float64, device=dev)
        sw_st = torch.empty(1 << 20, dtype=torch.uint8, device=dev)
        g.pd_pi_batch(sw_pairs, hop, out=sw_out, status=sw_st)
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        g.pd_pi_batch(sw_pairs, hop, out=sw_out, status=sw_st)
        torch.cuda.synchronize()
        sdt = time.perf_counter() - c0
        sweep = {"pairs": 1 << 20, "pairs_per_sec": (1 << 20) / sdt, "nonzero_rows": int((sw_out.abs().sum(1) > 0).sum()),
                 "note": "uniformly random pairs, hop 2, one pass; not part of `value`"}
        del sw_out, sw_pairs, sw_st
        g.pd_pi_batch(pi_pairs, hop, out=pi_out, status=pi_status)      # restore the headline batch's header for sizes()/stats()
        torch.cuda.synchronize()
    # auxiliary (untimed region): the reference's whole negative sweep (loaddatas.py:44-53 + TLCGNN.py:80-107: every non-edge of
    # the graph gets an image) -- pairs enumerated on the device by list number (tlc_complement_pairs), streamed through
    # tlc_pd_pi_batch in 2^22-pair chunks, informative rows kept by tlc_select_rows.  List order (the reference's seeded
    # shuffle permutes the same list; it costs ~20 s of host MT19937 and does not change the device work).
    full_sweep = None
    if rank == 0 and not args.no_sweep:
        try:
            from tlc_gnn_amd import pi_cache
            ci = engine.ComplementIndex(wl["rowptr"], wl["col"], device=local_rank)
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            store = pi_cache.sweep_images(g, lambda lo, hi: ci.pairs(first=lo, count=hi - lo), len(ci), hop, chunk=1 << 22)
            torch.cuda.synchronize()
            fdt = time.perf_counter() - c0
            full_sweep = {"pairs": len(ci), "seconds": fdt, "pairs_per_sec": len(ci) / fdt, "stored_rows": int(len(store.idx)),
                          "stored_fraction": len(store.idx) / len(ci), "dense_bytes": len(ci) * 200,
                          "sparse_bytes": int(store.idx.nbytes + store.rows.nbytes + store.status.nbytes),
                          "note": "all N(N+1)/2 - M non-edges of the training graph incl. the diagonal, host wall clock incl. the "
                                  "D2H of the kept rows; not part of `value`"}
            # the same sweep through the distance <= hop pre-filter (PI-C): only the non-edges inside each other's
            # hop-ball can have a non-zero row, so only they go through the pipeline; same stored rows
            pi_cache.sweep_near(g, ci, hop)
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            near = pi_cache.sweep_near(g, ci, hop)
            torch.cuda.synchronize()
            ndt = time.perf_counter() - c0
            full_sweep["prefiltered"] = {"seconds": ndt, "near_pairs": int(near.near_pairs), "list_pairs_per_sec": len(ci) / ndt,
                                         "nontrivial_pi_per_sec": near.near_pairs / ndt, "stored_rows": int(len(near.idx)),
                                         "same_rows_as_full_sweep": bool(np.array_equal(near.idx, store.idx) and np.array_equal(near.rows, store.rows))}
            del store, ci, near
            g.pd_pi_batch(pi_pairs, hop, out=pi_out, status=pi_status)      # restore the headline batch's header
            torch.cuda.synchronize()
        except Exception as ex:
            full_sweep = {"error": repr(ex)}
    # auxiliary (untimed region): BASELINE.json's other graph shapes (hop 1, TLCGNN.py:102) -- all positive pairs of the
    # shaped synthetic graph through tlc_pd_pi_batch, one call, device-resident; host wall clock
    other_shapes = None
    if rank == 0 and not args.no_sweep:
        other_shapes = {}
        for shape_name in ("Cora", "PPI", "Photo", "Computers"):
            try:
                from tlc_gnn_amd import synth as _synth
                n_s, e_s, k_s, hop_s, _ = _synth.shaped_graph(shape_name)
                rp_s, col_s, w_s = _synth.edges_to_csr(n_s, e_s, k_s)
                g_s = engine.DeviceGraph(rp_s, col_s, w_s, device=local_rank)
                p_s = torch.from_numpy(np.ascontiguousarray(e_s, dtype=np.int32)).to(dev)
                o_s = torch.empty((len(e_s), 25), dtype=torch.float64, device=dev)
                st_s = torch.empty(len(e_s), dtype=torch.uint8, device=dev)
                g_s.pd_pi_batch(p_s, hop_s, out=o_s, status=st_s)
                torch.cuda.synchronize()
                c0 = time.perf_counter()
                g_s.pd_pi_batch(p_s, hop_s, out=o_s, status=st_s)
                torch.cuda.synchronize()
                sdt_ = time.perf_counter() - c0
                other_shapes[shape_name] = {"nodes": int(n_s), "pairs": int(len(e_s)), "hop": int(hop_s), "ms": sdt_ * 1e3,
                                            "images_per_sec": len(e_s) / sdt_,
                                            "tiers": {k: int(v) for k, v in g_s.stats().items() if k.startswith("tier")}}
                g_s.close()
                del p_s, o_s, st_s
            except Exception as ex:
                other_shapes[shape_name] = {"error": repr(ex)}
    # auxiliary (untimed region): the step before the path -- Ollivier-Ricci curvature (alpha 0.5, Sinkhorn reg 0.1) of every
    # edge of the training graph (loaddatas.py:105-123), the producer of the path's edge weights
    ricci = None
    if rank == 0 and not args.no_sweep:
        try:
            te_ = wl["train_edges"]
            rp_, col_ = wl["rowptr"], wl["col"]
            engine.ollivier_ricci_sinkhorn(rp_, col_, te_)                      # warm-up: module load, LDS attribute, allocator
            torch.cuda.synchronize()
            c0 = time.perf_counter()
            kap_, it_ = engine.ollivier_ricci_sinkhorn(rp_, col_, te_, want_iters=True)
            rdt = time.perf_counter() - c0
            ricci = {"edges": int(len(te_)), "seconds": rdt, "edges_per_sec": len(te_) / rdt, "mean_iterations": float(it_.mean()),
                     "max_iterations": int(it_.max()), "kappa_min": float(kap_.min()), "kappa_max": float(kap_.max()),
                     "note": "host wall clock incl. H2D of the CSR and D2H of kappa; not part of `value`"}
            if not args.no_cpu_baseline:
                from oracle import ricci_ref
                sub = te_[:: max(1, len(te_) // 300)][:300]
                c0 = time.perf_counter()
                ref_k = _ricci_cpu_sample(ricci_ref, wl["n"], te_, sub)
                cdt = time.perf_counter() - c0
                got_k = kap_[:: max(1, len(te_) // 300)][:300]
                ricci["cpu_restatement_edges_per_sec_1thread"] = len(sub) / cdt
                ricci["max_abs_diff_vs_cpu_sample"] = float(np.abs(ref_k - got_k).max())
        except Exception as ex:
            ricci = {"error": repr(ex)}
    # auxiliary (untimed region): configs 3/5 of BASELINE.json -- the per-graph PDGNN forward next to the exact PD of the same
    # graphs (Knowledge_Distillation evaluate_time, train_Teacher_Model_GC.py:118-143) on HIV-shaped synthetic molecules
    # (n ~ Poisson(25), a random tree plus a few ring-closing edges, degree filtration / (max + 1e-10), data_utils_GC.py:117-119),
    # one block-diagonal batch, weights random-init with a fixed seed.  Reported beside `value`, never part of it.
    pdgnn = None
    if rank == 0 and not args.no_sweep:
        try:
            pdgnn = pdgnn_aux(torch, dev)
        except Exception as ex:                                   # the headline line must not depend on the auxiliary
            pdgnn = {"error": repr(ex)}
    # auxiliary (untimed): the LP leg's two bounded kernels on their own -- the feature GEMM against the f32 MFMA peak and
    # the scatter-add SpMM against HBM (north_star); torch events on the current stream, which is where ops.* enqueue
    lp_roof = None
    if rank == 0:
        def _avg_us(fn, reps=20):
            fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        Mr, Kf, Nh = x_local.shape[0], x_local.shape[1], w1.shape[1]
        xw = torch.empty((Mr, Nh), dtype=torch.float32, device=dev)
        us_g = _avg_us(lambda: ops.gemm(x_local, w1, out=xw))
        hfull = torch.empty((n, Nh), dtype=torch.float32, device=dev).normal_()
        yfull = torch.empty((n, Nh), dtype=torch.float32, device=dev)
        us_s = _avg_us(lambda: ops.spmm(rowptr_n, col_n, val_n, hfull, bias=b1, relu=True, out=yfull))
        nnz = int(col_n.shape[0])
        spmm_bytes = nnz * (Nh * 4 + 8) + n * (Nh * 4 + 4)            # gathered rows + col/val + output rows + rowptr
        lp_roof = {"feature_gemm": {"bound": "mfma", "shape_mkn": [int(Mr), int(Kf), int(Nh)], "kernel_us": us_g,
                                    "achieved": 2.0 * Mr * Kf * Nh / us_g / 1e6, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": 2.0 * Mr * Kf * Nh / us_g / 1e6 / MFMA_F32_PEAK_TFLOPS, "dtype": "f32 (v_mfma_f32_16x16x4_f32)"},
                   "scatter_add_spmm": {"bound": "hbm", "rows": int(n), "nnz": nnz, "k": int(Nh), "kernel_us": us_s,
                                        "achieved": spmm_bytes / us_s / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": spmm_bytes / us_s / 1e3 / HBM_PEAK_GBS}}
        del xw, hfull, yfull
        # the stand-alone PI raster (tlc_pi_raster = PersistenceImager.transform) on diagrams shaped like this batch's:
        # one diagram per pair with as many points as the vicinity has edges (Ord0 + ext0 + Ext1 points), values in [0,1]
        n_sz_r, m2_sz_r = g.sizes(E)
        kpts = np.maximum(m2_sz_r // 2, 1).astype(np.int64)
        offs_r = torch.from_numpy(np.concatenate([[0], np.cumsum(kpts)])).to(dev)
        gen = torch.Generator(device=dev).manual_seed(7)
        bd = torch.rand((int(kpts.sum()), 2), generator=gen, device=dev, dtype=torch.float64)
        bd[:, 1] = bd[:, 0] + bd[:, 1] * (1.0 - bd[:, 0])                     # death >= birth
        us_r = _avg_us(lambda: engine.pi_raster(offs_r, bd, 5), reps=20)
        raster_bytes = 16.0 * float(kpts.sum()) + 8.0 * 25 * E + 8.0 * (E + 1)
        lp_roof["pi_raster"] = {"bound": "hbm", "diagrams": int(E), "points": int(kpts.sum()), "kernel_us": us_r,
                                "achieved": raster_bytes / us_r / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": raster_bytes / us_r / 1e3 / HBM_PEAK_GBS,
                                "note": "fp64 FMA issue-bound: 2*(res+1) CDF series of 19 terms + res^2 pixel terms per point "
                                        "(~370 fp64 instructions per point), not HBM"}
        del bd, offs_r
    if rank == 0:
        stats = g.stats()
        n_sz, m2_sz = g.sizes(E)
        tiers = engine.tier_of(n_sz, m2_sz)
        bytes_pp = engine.algorithmic_bytes(wl["rowptr"], wl["col"], wl["pi_pairs"], hop)
        kavg = dict(warm_avg)                                   # all kernels: warm-up steps (every kernel carried events there)
        dom = dom_warm
        live = [x for x in ktimes_timed if x >= 0]
        if live:
            kavg[dom] = float(np.mean(live))                    # the dominant kernel: live, inside the timed region
        if dom.startswith("pd_tier"):
            dom_bytes = float(bytes_pp[tiers == dom].sum())
            dom_units = int((tiers == dom).sum())
        else:
            dom_bytes, dom_units = float(bytes_pp.sum()), E
        achieved = dom_bytes / (kavg[dom] * 1e-3) / 1e9
        traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get(dom)
            except Exception:
                traffic = None
        out = {
            "metric": "persistence-images/sec + LP-forward edges/sec, PubMed-scale, 1/2/4/8 GPU",
            "value": world * E * args.steps / t_pi,
            "unit": "persistence-images/sec",
            "lp_forward_edges_per_sec": world * dec_pairs.shape[0] * args.steps / t_lp,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall / args.steps * 1e3,
            "pi_ms_per_step": t_pi / args.steps * 1e3, "lp_ms_per_step": t_lp / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "PubMed-shaped synthetic graph (N=19717, M=44324, F=500, seed 1234), hop=2; per GPU: "
                                   "PI-A = all %d train-positive pairs -> 5x5 persistence images, then TLCGNN forward "
                                   "(GCN 500->100->16 encode + fused decode of %d pairs, image rows resident)" % (E, dec_pairs.shape[0]),
                       "pairs_per_gpu": E, "decode_pairs_per_gpu": int(dec_pairs.shape[0]),
                       "parallelism": "pair shards per GPU (no collective); encoder node-row sharded, 1 all-gather per layer",
                       "vicinity_tiers": {k: int(v) for k, v in stats.items() if k.startswith("tier")},
                       "tie_fallback_sources": int(stats["tie_fallback_sources"])},
            "roofline": {"bound": "hbm", "kernel": dom, "kernel_ms": kavg[dom], "units_per_launch": dom_units,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic},
            "roofline_chain": {"bound": "hbm", "algorithmic_bytes_per_pi": float(bytes_pp.mean()),
                               "achieved": float(bytes_pp.sum()) * world * args.steps / t_pi / 1e9, "peak": HBM_PEAK_GBS * world,
                               "unit": "GB/s", "frac": float(bytes_pp.sum()) * args.steps / t_pi / 1e9 / HBM_PEAK_GBS},
            "kernel_ms": kavg,
            "kernel_ms_note": "'%s' (the roofline kernel): HIP events inside the timed region, mean of %d steps; the others: events "
                              "during the %d warm-up steps (all eight event pairs inside the timed region cost 48 us per batch)"
                              % (dom, len(live), max(args.warmup, 1)),
            "roofline_lp": lp_roof,
            "sweep": sweep,
            "full_sweep": full_sweep,
            "ricci": ricci,
            "other_shapes": other_shapes,
            "pdgnn": pdgnn,
        }
        if world == 1 and not args.no_cpu_baseline:
            # the CPU restatement (oracle/tlc_oracle.c, a port of the reference's algorithm) on this box's host cores,
            # same batch, bounded sample.  A reported baseline, not a target.
            from oracle import oracle
            sample = wl["pi_pairs"][: min(E, 20000)]
            oracle.pd_pi_batch(wl["rowptr"], wl["col"], wl["w"], sample[:256], hop, n_threads=0)    # warm up / build
            c0 = time.perf_counter()
            ref, rst, used = oracle.pd_pi_batch(wl["rowptr"], wl["col"], wl["w"], sample, hop, n_threads=0)
            cdt = time.perf_counter() - c0
            out["cpu_baseline"] = {"value": len(sample) / cdt, "unit": "persistence-images/sec", "cores": int(used),
                                   "kind": "port", "sample": "first %d pairs of the same PI-A batch, OpenMP over pairs, "
                                   "%.2f s wall" % (len(sample), cdt)}
            # the same restatement on ONE host thread (SURVEY.md 8d asks for both), first 2 000 pairs
            c0 = time.perf_counter()
            oracle.pd_pi_batch(wl["rowptr"], wl["col"], wl["w"], sample[:2000], hop, n_threads=1)
            out["cpu_baseline"]["value_1thread"] = min(len(sample), 2000) / (time.perf_counter() - c0)
            got = pi_out[: len(sample)].cpu().numpy()
            nz = ref != 0
            out["cpu_baseline"]["max_rel_diff_vs_gpu"] = float((np.abs(got[nz] - ref[nz]) / np.abs(ref[nz])).max()) if nz.any() else 0.0
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

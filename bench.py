#!/usr/bin/env python3
"""bench.py -- persistence-images/sec + LP-forward edges/sec on the PubMed-shaped synthetic graph (BASELINE.json).

  python bench.py --gpus N --steps K --warmup W            (N=1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W   (N>1)

One "step" = one pass of the hot path over one batch per GPU:
  leg 1  PD/PI : all 37 676 train-positive pairs (hop 2) -> 5x5 persistence images        (tlc_pd_pi_batch)
  leg 2  LP fwd: TLCGNN encode (2-layer GCN) + fused decode over 2*37 676 = 75 352 pairs, image rows resident in HBM
Inputs are resident in HBM before the timed region.  The timed region is K steps with no host synchronisation inside: the K
image batches enqueued back to back, then the K forwards (the two legs of a step do not depend on each other); `value` is
the image throughput of that region, `pi_latency_ms` the time of one synchronised batch.  --rotate-batches feeds a different
batch every step.  Weak scaling: every rank owns a full batch (the small CSR replicated, no data-path collective in leg 1;
the encoder of leg 2 either node-row sharded with one RCCL all-gather per layer or replicated, whichever is faster);
`strong_scaling` beside it cuts ONE list (the 504 514 non-edges within hop distance) into cost-balanced shards.
`--gpus N` without a launcher starts its own N ranks (torch.distributed.run, before any GPU call).  Rank 0 prints ONE JSON line.
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
    return dict(n=n, hop=hop, n_feat=n_feat, rowptr=rowptr, col=col, w=w, train_edges=tr_sorted, all_pos=pos,
                pi_pairs=np.ascontiguousarray(pi_pairs, dtype=np.int32), neg=neg, x=x)


def measure_traffic():
    """roofline.traffic measured INSIDE this run: two child processes of this script under `rocprofv3 --pmc` (FETCH_SIZE and
    WRITE_SIZE need a pass each: MI355X_MICROARCH.md, HBM / counters), 3 batches each, folded by tools/pmc_to_traffic.fold into
    bytes per batch of every kernel's launches -> (dict, detail) or (None, reason); the caller then falls back to
    profiles/pmc_traffic.json.  Called BEFORE this process touches the GPU (a process that has initialised the GPU must not start
    other programs on this pool); the children run one after the other and are killed as a process group on time-out."""
    import shutil, signal, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_to_traffic
    except Exception as ex:
        return None, "tools/pmc_to_traffic.py not importable: %r" % (ex,)
    tmp = tempfile.mkdtemp(prefix="tlc_pmc_")
    env = dict(os.environ, TLC_BENCH_CHILD="1")
    try:
        for cname, sub in (("FETCH_SIZE", "fetch"), ("WRITE_SIZE", "write")):
            cmd = [exe, "--pmc", cname, "--output-format", "csv", "-d", os.path.join(tmp, sub), "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-sweep"]
            pr = subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=150)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)
                except Exception:
                    pass
                pr.wait()
                return None, "the %s pass timed out" % cname
            if rc != 0:
                return None, "the %s pass failed (rc %d)" % (cname, rc)
        res, detail, _ = pmc_to_traffic.fold(os.path.join(tmp, "fetch"), os.path.join(tmp, "write"))
        if not res:
            return None, "no counter rows collected"
        return res, detail
    except Exception as ex:
        return None, repr(ex)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def measure_issue():
    """roofline_issue measured INSIDE this run: two more child passes of this script under `rocprofv3 --pmc` (counters only, program
    directly behind `--`), 3 batches each -- SQ_INSTS_VALU / _LDS / _SALU, then SQ_INSTS (every instruction) / _SMEM / _VMEM_RD /
    _VMEM_WR / _BRANCH; -> (dict of wave-instructions per image batch summed over the PD/PI kernels, detail) or (None, reason).
    Called BEFORE this process touches the GPU, like measure_traffic."""
    import csv, glob, shutil, signal, subprocess, tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    tmp = tempfile.mkdtemp(prefix="tlc_pmc_issue_")
    env = dict(os.environ, TLC_BENCH_CHILD="1")
    try:
        out, per_kernel_valu, batches = {}, {}, 0
        for pi_, counters in enumerate((("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_SALU"),
                                        ("SQ_INSTS", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_BRANCH"))):
            sub = os.path.join(tmp, "issue%d" % pi_)
            cmd = [exe, "--pmc"] + list(counters) + ["--output-format", "csv", "-d", sub, "--",
                   sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-sweep"]
            pr = subprocess.Popen(cmd, cwd=tmp, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                rc = pr.wait(timeout=240)
            except subprocess.TimeoutExpired:
                os.killpg(pr.pid, signal.SIGKILL)
                pr.wait()
                return (out or None), ("the %s pass timed out" % counters[0] if not out else {"batches_in_pass": batches, "per_kernel_valu": per_kernel_valu,
                                                                                               "second_pass": "timed out"})
            if rc != 0:
                if out:
                    return out, {"batches_in_pass": batches, "per_kernel_valu": per_kernel_valu, "second_pass": "failed (rc %d)" % rc}
                return None, "the %s pass failed (rc %d)" % (counters[0], rc)
            tot, per_kernel, scans = {}, {}, 0
            for f in glob.glob(sub + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(f)):
                    k, c = row["Kernel_Name"], row.get("Counter_Name")
                    if not k.startswith(("tlc_", "void tlc_")) or "tlc_pi_raster" in k or "ball_" in k:
                        continue                                      # (the image batch's kernels; the one-off ball lists are set-up)
                    v = float(row["Counter_Value"])
                    tot[c] = tot.get(c, 0.0) + v
                    per_kernel.setdefault(k[:60], {}).setdefault(c, 0.0)
                    per_kernel[k[:60]][c] += v
                    if c == counters[0] and "tlc_scan_bin" in k:
                        scans += 1
            if not tot or scans == 0:
                if out:
                    return out, {"batches_in_pass": batches, "per_kernel_valu": per_kernel_valu, "second_pass": "no counter rows"}
                return None, "no counter rows collected"
            out.update({c: v / scans for c, v in tot.items()})
            if pi_ == 0:
                batches = scans
                per_kernel_valu = {k: v.get("SQ_INSTS_VALU", 0.0) / scans for k, v in per_kernel.items()}
        return out, {"batches_in_pass": batches, "per_kernel_valu": per_kernel_valu}
    except Exception as ex:
        return None, repr(ex)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def pdgnn_aux(torch, dev, n_graphs=41127, seed=1234):
    """PDGNN forward vs exact PD on HIV-shaped molecules -- as many graphs as ogbg-molhiv holds (41 127, data_utils_GC.py:284; config 5
    of BASELINE.json): graphs/s of each (device-resident inputs, median of 5)."""
    from tlc_gnn_amd import engine
    from tlc_gnn_amd.Knowledge_Distillation.Teacher_model import Teacher_Model
    from tlc_gnn_amd import synth
    e_all, f, node_offs, edge_offs = synth.hiv_shaped_molecules(n_graphs, seed)
    node_offs, edge_offs = node_offs.tolist(), edge_offs.tolist()
    edges = [e_all[edge_offs[k]:edge_offs[k + 1]].astype(np.int64) for k in range(n_graphs)]
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
    from tlc_gnn_amd.Knowledge_Distillation.gat_conv import GraphBatch
    with torch.no_grad():
        # the evaluate-time loop of the reference (train_Teacher_Model.py:124-151) runs the model over fixed graphs: the caller
        # builds the batch's structure once (GraphBatch) and every forward reads it; `..._with_csr_build`: built inside each call
        gb = GraphBatch(ei, n_tot)
        pd_ms = med_ms(lambda: model(x, ei, None, compute_loss=False, grad_PI=False, graph_ptr=gptr, edge_ptr=d_eptr, csr=gb))
        pd_build_ms = med_ms(lambda: model(x, ei, None, compute_loss=False, grad_PI=False, graph_ptr=gptr, edge_ptr=d_eptr))
        csr_ms = med_ms(lambda: GraphBatch(ei, n_tot))
    # roofline of the forward (review of round 4, item 3): algorithmic bytes per layer from DESIGN.md's formula
    # 4n(c_in + 3C + 4) + nnz(4 + 4(C + 1)) + 8nC (node rows in, [P|Q|alpha] out and gathered per in-edge, the 2C-wide output), the
    # edge head 4n(c + 2h) + E(8 + 8h + 8), the raster 16 per point + 8 res^2 per graph; f32 MFMA flops of the node products
    nnz = int(ei.shape[1])
    E_dir = int(2 * len(e_all))
    layers = [(1, 32), (64, 32), (64, 32), (64, 16)]
    lay_bytes = [4.0 * n_tot * (ci + 3 * C_ + 4) + nnz * (4.0 + 4.0 * (C_ + 1)) + 8.0 * n_tot * C_ for ci, C_ in layers]
    head_bytes = 4.0 * n_tot * (32 + 2 * 32) + E_dir * (8.0 + 8.0 * 32 + 8.0)
    raster_bytes = 16.0 * len(e_all) + 8.0 * 25 * n_graphs
    alg_bytes = sum(lay_bytes) + head_bytes + raster_bytes
    flops = sum(2.0 * n_tot * ci * (2 * C_ + 1) for ci, C_ in layers[1:]) + 2.0 * n_tot * 32 * 64
    roof = {"bound": "hbm", "algorithmic_bytes": alg_bytes, "bytes_per_layer": lay_bytes, "edge_head_bytes": head_bytes,
            "achieved": alg_bytes / (pd_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg_bytes / (pd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "node_product_flops": flops, "node_product_time_at_f32_mfma_peak_ms": flops / (MFMA_F32_PEAK_TFLOPS * 1e12) * 1e3,
            "note": "whole forward (4 layers + edge head + raster) against HBM; the f32 node products alone would take the time given at the "
                    "157.3 TF f32 MFMA peak; per-kernel durations: profiles/r05_pdgnn_kernel_stats*.csv"}
    return {"graphs": int(n_graphs), "nodes": int(n_tot), "edges": int(len(e_all)), "roofline_pdgnn": roof,
            "pdgnn_forward_graphs_per_sec": n_graphs / (pd_ms * 1e-3), "pdgnn_forward_ms": pd_ms,
            "pdgnn_forward_ms_with_csr_build": pd_build_ms, "csr_by_target_build_ms": csr_ms,
            "exact_pd_graphs_per_sec": n_graphs / (exact_ms * 1e-3), "exact_pd_ms": exact_ms,
            "note": "HIV-shaped synthetic molecules in one block-diagonal batch; PDGNN = 4 GAT layers + edge head + 5x5 image "
                    "(random-init weights, seed 1234), exact = tlc_pd_from_filtration on the same graphs; host wall clock around "
                    "device-resident calls; not part of `value`"}


def pdgnn_amazon_aux(torch, dev, n_pairs=4096, seed=1234):
    """BASELINE configs[2]: PDGNN (gat_conv.py) forward on the hop-1 vicinities of the Amazon-shaped graphs, the caller shape of
    gcn_LP_GIN.Net.compute_PI (:43-64) -- vicinity + filtration of every candidate pair on the device (tlc_vicinity_filtration),
    all vicinities of the sample stacked block-diagonally, ONE Teacher_Model forward, one image per vicinity -- beside the exact
    diagrams + images of the same pairs (tlc_pd_pi_batch with the PDGNN fork's flags).  Device-resident, median of 5."""
    from tlc_gnn_amd import synth, _lib
    from tlc_gnn_amd.Knowledge_Distillation.data_utils_LP import Vicinities, KD_LP_FLAGS, stacked
    from tlc_gnn_amd.Knowledge_Distillation.Teacher_model import Teacher_Model
    out = {}
    torch.manual_seed(seed)
    model = Teacher_Model(type='GAT').eval().to(dev)

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

    for shape in ("Photo", "Computers"):
        n, edges, kappa, hop, _ = synth.shaped_graph(shape)
        ricci = np.concatenate([np.concatenate([edges, kappa[:, None]], 1), np.concatenate([edges[:, ::-1], kappa[:, None]], 1)]).tolist()
        vic = Vicinities(edges, ricci)
        rs = np.random.RandomState(seed)
        pairs = edges[rs.permutation(len(edges))[:n_pairs]]
        state = {}

        def extract():
            state["b"] = vic.batch(pairs, hop, node_cap=512, edge_cap=8192)

        def forward():
            b = state["b"]
            x, ei = stacked(b)                   # (the vicinities' edge_index + self loops and float32 filtration, gcn_LP_GIN.py:43-64)
            with torch.no_grad():
                state["img"] = model(x, ei, None, compute_loss=False, grad_PI=False, graph_ptr=b["node_ptr"], edge_ptr=b["edge_ptr"])[1]

        ex_ms = med_ms(extract)
        fw_ms = med_ms(forward)
        g = vic._g2p._device_graph()
        mapped = torch.from_numpy(vic._g2p._map_pairs(pairs)).to(dev)
        flags = KD_LP_FLAGS | _lib.KEEP_ZERO_PERS | _lib.PI_ORD0_EXT1
        exact_ms = med_ms(lambda: g.pd_pi_batch(mapped, hop, flags=flags))
        b = state["b"]
        sizes = (b["node_ptr"][1:] - b["node_ptr"][:-1]).cpu().numpy()
        out[shape] = {"pairs": int(len(pairs)), "hop": int(hop), "nodes": int(b["node_ptr"][-1]), "edges": int(b["edge_ptr"][-1]),
                      "largest_vicinity_nodes": int(sizes.max()), "vicinity_extraction_ms": ex_ms, "pdgnn_forward_ms": fw_ms,
                      "pdgnn_vicinities_per_sec": len(pairs) / ((ex_ms + fw_ms) * 1e-3),
                      "pdgnn_forward_only_vicinities_per_sec": len(pairs) / (fw_ms * 1e-3),
                      "exact_pd_pi_ms": exact_ms, "exact_vicinities_per_sec": len(pairs) / (exact_ms * 1e-3)}
        g.close()
    out["note"] = ("Amazon-shaped synthetic graphs, hop 1, a sample of the positive pairs; PDGNN = vicinity extraction + 4 GAT layers + "
                   "edge head + 5x5 image per vicinity (random-init weights, seed 1234, one edge per undirected pair as in "
                   "gcn_LP_GIN.compute_PI); exact = tlc_pd_pi_batch on the same pairs with the PDGNN fork's flags; host wall clock "
                   "around device-resident calls; not part of `value`")
    return out


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


def _self_launch(n_gpus, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves -- a child
    `python -m torch.distributed.run` (one process per GPU, rendezvous on 127.0.0.1), BEFORE this process has touched the
    GPU (nothing here has even imported torch), and hand its exit code back.  Never an exec of a GPU-initialised process."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def rotated_batches(wl, count, seed=4321):
    """`count` different batches of the headline size: samples without replacement of ALL positive pairs of the graph
    (train + val + test positives: 44 324), random orientation -- tier counts, arena size and the heavy-pair list differ
    from batch to batch, so the library's guesses from the previous chunk (arena size, speculative grid sizes) are
    never exactly right, as in a sweep over changing chunks."""
    E = len(wl["pi_pairs"])
    out = []
    for j in range(count):
        rs = np.random.RandomState(seed + j)
        b = wl["all_pos"][rs.choice(len(wl["all_pos"]), size=E, replace=False)]
        flip = rs.rand(E) < 0.5
        b = np.where(flip[:, None], b[:, ::-1], b)
        out.append(np.ascontiguousarray(b, dtype=np.int32))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-sweep", action="store_true", help="skip the auxiliary blocks (profiling runs)")
    ap.add_argument("--rotate-batches", action="store_true",
                    help="timed region over different 37 676-pair batches (see rotated_batches) instead of the same one")
    ap.add_argument("--encoder", default="auto", choices=["auto", "allgather", "replicated"],
                    help="N > 1: node-row sharded encoder with one RCCL all-gather per layer, or the whole encoder on every "
                         "rank; auto = time both before the timed region and keep the faster")
    ap.add_argument("--dist-backend", default="nccl", help="rehearsal only: 'gloo' lets N ranks share ONE GPU with --single-device "
                    "(RCCL refuses two ranks on a device); the measured configuration is always nccl = RCCL, one rank per GPU")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--sync-batches", action="store_true",
                    help="timed region with stream-ordered tlc_pd_pi_batch calls (one batch at a time) instead of "
                         "tlc_pd_pi_batch_async + one join (three batches in flight)")
    ap.add_argument("--lp-graph", default="auto", choices=["auto", "on", "off"],
                    help="replay the LP forward from a HIP graph (one launch per step); auto = only when the host needs more than "
                         "85 %% of a forward's device time to submit it kernel by kernel")
    ap.add_argument("--no-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes that measure roofline.traffic (it is then read from profiles/)")
    ap.add_argument("--no-kernel-events", action="store_true",
                    help="development: no per-kernel HIP events in the timed region (the roofline block is then meaningless)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args.gpus, sys.argv[1:]))

    # roofline.traffic: two child passes of this script under rocprofv3 --pmc, BEFORE this process initialises the GPU
    pre_traffic, pre_traffic_detail = None, "not attempted"
    pre_issue, pre_issue_detail = None, "not attempted"
    if args.gpus == 1 and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not args.no_sweep and not args.no_traffic \
            and not os.environ.get("TLC_BENCH_CHILD"):
        pre_traffic, pre_traffic_detail = measure_traffic()
        pre_issue, pre_issue_detail = measure_issue()

    import torch
    from tlc_gnn_amd import engine, ops, dist as tdist, _lib
    from tlc_gnn_amd.baselines import TLCGNN

    rank, local_rank, world = tdist.env_world()
    if args.gpus != world:
        raise SystemExit("--gpus %d but the launcher started %d rank(s): `value` is the aggregate over WORLD_SIZE ranks, "
                         "the two must agree" % (args.gpus, world))
    _lib.require_gpu()
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    wl = build_workload(rank)
    n, hop = wl["n"], wl["hop"]
    # (the handle -- and with it its streams -- ahead of the communicator: hardware queues are dealt to streams in creation order,
    # and the image pipeline wants one queue per stream; RCCL's own streams then take what is left)
    if args.single_device and world > 1:
        # (rehearsal: several ranks on ONE card.  A process holds up to twelve hardware queues; two such processes oversubscribe
        # the card's queue slots and the firmware time-slices them -- seconds per batch.  Two workspaces, streams on demand.)
        os.environ["TLC_LAZY_STREAMS"] = "1"
    g = engine.DeviceGraph(wl["rowptr"], wl["col"], wl["w"], device=local_rank)
    if args.single_device and world > 1:
        g.set_option("n_ws", 2)
    if world > 1:
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)
    pi_pairs = torch.from_numpy(wl["pi_pairs"]).to(dev)
    E = pi_pairs.shape[0]
    pi_out = torch.empty((E, 25), dtype=torch.float64, device=dev)
    pi_status = torch.empty(E, dtype=torch.uint8, device=dev)
    rot = [torch.from_numpy(b).to(dev) for b in rotated_batches(wl, 8, seed=4321 + 100 * rank)]

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
    x_full = torch.from_numpy(wl["x"]).to(dev).contiguous()
    dec_pairs_np = np.concatenate([wl["pi_pairs"].astype(np.int64), wl["neg"]]).astype(np.int32)
    dec_pairs = torch.from_numpy(dec_pairs_np).to(dev)
    dec_pi, _ = g.pd_pi_batch(dec_pairs, hop)                         # image rows of the decode batch: resident
    dec_pi = dec_pi.float()                                           # ... as float32, cast once like Net._tables (the reference's torch.Tensor(PI), TLCGNN.py:52-53)
    w1, b1 = model.conv1.weight.detach(), model.conv1.bias.detach()
    w2, b2 = model.conv2.weight.detach(), model.conv2.bias.detach()
    l1w, l1b = model.linear_1.weight.detach(), model.linear_1.bias.detach()
    l2w, l2b = model.linear.weight.detach(), model.linear.bias.detach()
    prob = torch.empty(dec_pairs.shape[0], dtype=torch.float32, device=dev)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    def make_encoder(mode, prob_out=None):
        box = {}
        prob_out = prob if prob_out is None else prob_out

        def gemm(a, b, out=None):
            # the feature projection runs over the stored entries of the (10 % dense, never changing) feature rows, like
            # Net.encode does; every other product is the dense MFMA kernel
            if a is box.get("x") and box["xs"] is not None:
                return ops.sparse_gemm(box["xs"], b, out=out)
            return ops.gemm(a, b, out=out)
        enc = tdist.ShardedGCNEncoder(rowptr_n, col_n, val_n, n, world, rank, mode=mode, gemm=gemm,
                                      spmm=lambda rp, c, v, xx, bias, relu, renorm=False, out=None:
                                      ops.spmm(rp, c, v, xx, bias=bias, relu=relu, renorm=renorm, out=out))
        x_local = x_full[enc.lo:enc.hi].contiguous()
        xs_local = ops.SparseRows(x_local)
        box["x"], box["xs"] = x_local, (xs_local if xs_local.density < TLCGNN.Net.SPARSE_FEATURES_BELOW and ops.sparse_gemm_fits(x_local.shape[1], w1.shape[1]) else None)
        pairs_mapped = enc.row_map(dec_pairs.long()).to(torch.int32).contiguous()      # decode pairs index encode()'s layout

        fused = enc.rows is None                                        # no exchange step: one call for the encoder
        emb_buf = torch.empty((n, w2.shape[1]), dtype=torch.float32, device=dev) if fused else None

        def leg():
            if fused:
                emb = ops.gcn2_encode(rowptr_n, col_n, val_n, x_local, w1, b1, w2, b2, relu=True, renorm=True, out=emb_buf, x_sparse=box["xs"])
            else:
                emb = enc.encode(x_local, w1, b1, w2, b2, renorm=True)  # renorm_ of TLCGNN.py:48 fused into the last SpMM
            ops.lp_decode(pairs_mapped, emb, dec_pi, l1w, l1b, l2w, l2b, out=prob_out)
        return leg

    def time_leg(fn, reps):
        fn()
        barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        barrier()
        t = torch.tensor([e0.elapsed_time(e1) / reps], dtype=torch.float64, device=dev)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # N > 1: the encoder either shards node rows (3 all-gathers per forward) or runs whole on every rank (no collective);
    # both are timed here, outside the timed region, and the faster one is the LP leg (same result either way)
    encoder_ms = {}
    modes = ["replicated"] if world == 1 else (["allgather", "replicated"] if args.encoder == "auto" else [args.encoder])
    legs = {}
    for mode in modes:
        legs[mode] = make_encoder(mode)
        encoder_ms[mode] = time_leg(legs[mode], 10)
    enc_mode = min(encoder_ms, key=lambda k: encoder_ms[k])
    leg_lp = legs[enc_mode]
    # The forward is five small kernels (79 us on the device): submitted one by one through the Python wrappers it needs a host
    # that keeps up (a box with a slow host measured 0.18 ms per forward, submission-bound).  On such a host, and without a
    # collective inside, the whole forward is captured once into a HIP graph -- same kernels, same buffers, one launch per step --
    # and replayed; kept only if it reproduces the eager output bit for bit and is within 5 % of the eager time on the device.
    lp_submit = "eager"
    # (how long the host takes to SUBMIT a forward, without waiting for it)
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    for _ in range(10):
        leg_lp()
    lp_host_ms = (time.perf_counter() - h0) * 1e2
    torch.cuda.synchronize()
    encoder_ms["host submission, eager"] = lp_host_ms
    # Only when the host is that slow: replayed from a graph the forward takes 85 instead of 79 us on the device.
    want_graph = args.lp_graph == "on" or (args.lp_graph == "auto" and lp_host_ms > 0.85 * encoder_ms[enc_mode])
    if enc_mode == "replicated" and want_graph and world == 1:       # (N > 1: the communicator's watchdog thread must not meet a capture)
        try:
            leg_lp()
            torch.cuda.synchronize()
            want_prob = prob.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                leg_lp()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                leg_lp()
            prob.zero_()
            graph.replay()
            torch.cuda.synchronize()
            if torch.equal(prob, want_prob):
                encoder_ms["replicated, HIP graph"] = time_leg(graph.replay, 10)
                if encoder_ms["replicated, HIP graph"] <= 1.05 * encoder_ms["replicated"]:   # (one launch per step: robust against a slow host)
                    leg_lp, lp_submit = graph.replay, "hipGraph"
        except Exception as exc:                                       # (capture refused: the eager leg stays)
            lp_submit = "eager (graph capture failed: %s)" % str(exc).splitlines()[0][:120]
            torch.cuda.synchronize()

    # Beside the leg (not part of it, nothing below reads it): the same forward with TWO in flight, each on its own stream and
    # its own buffers.  One forward is four small kernels back to back, each too short to fill 256 CUs; what two of them
    # together take per forward says how much of the leg's time is the machine standing half empty.
    if world == 1 and enc_mode == "replicated":
        try:
            prob_b = torch.empty_like(prob)
            pair_of_legs = [legs["replicated"], make_encoder("replicated", prob_out=prob_b)]
            lanes = [torch.cuda.Stream(), torch.cuda.Stream()]

            def two_in_flight(reps=20):
                cur = torch.cuda.current_stream()
                for lane in lanes:
                    lane.wait_stream(cur)
                for r in range(reps):
                    with torch.cuda.stream(lanes[r & 1]):
                        pair_of_legs[r & 1]()
                for lane in lanes:
                    cur.wait_stream(lane)
            two_in_flight(4)
            torch.cuda.synchronize()
            if torch.equal(prob_b, prob):
                encoder_ms["replicated, two forwards in flight (per forward)"] = time_leg(lambda: two_in_flight(20), 5) / 20
        except Exception as exc:
            encoder_ms["two forwards in flight"] = "failed: %s" % str(exc).splitlines()[0][:120]
            torch.cuda.synchronize()

    def leg_pi(step=0):
        g.pd_pi_batch(rot[step % len(rot)] if args.rotate_batches else pi_pairs, hop, out=pi_out, status=pi_status)

    # The timed region submits its K image batches with tlc_pd_pi_batch_async and joins once behind the last one: the
    # library keeps three batches in flight on its workspaces and submits a batch's tier launches behind the NEXT batch's first
    # half, so the lead-in of a batch (classification, early extraction, the main extraction: latency-bound, half-empty machine)
    # runs under the tier kernels of the batch before.  Three output
    # buffers in turn: a batch in flight owns its buffers until the join.  --sync-batches: stream-ordered calls instead.
    pi_outs = [pi_out] + [torch.empty_like(pi_out) for _ in range(2)]
    pi_sts = [pi_status] + [torch.empty_like(pi_status) for _ in range(2)]

    def leg_pi_submit(step, rotate):
        g.pd_pi_batch(rot[step % len(rot)] if rotate else pi_pairs, hop, out=pi_outs[step % 3], status=pi_sts[step % 3],
                      async_=not args.sync_batches)

    # ---- warm-up + latency line -------------------------------------------------------------------------------------------
    # Per-kernel HIP events (recorded inside the library on the stream each kernel runs on) cost time themselves: all eight
    # pairs add 48 us to the 1.06 ms batch.  Every kernel is timed during the warm-up steps (-> `kernel_ms` and which kernel
    # dominates); in the timed region only the dominant kernel carries events, kept by the library in a ring and read AFTER
    # the region.  The warm-up steps synchronise after every batch: their median is the LATENCY of one batch
    # (`pi_latency_ms`); the timed region below measures THROUGHPUT (batches enqueued back to back, no host sync).
    ktimes = {k: [] for k in engine.DeviceGraph.KERNELS}
    g.set_timing(not args.no_kernel_events)
    lat = []
    for s in range(max(args.warmup, 1)):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        leg_pi(s)
        e1.record()
        leg_lp()
        for k, v in g.timings().items():
            ktimes[k].append(v)
        lat.append(e0.elapsed_time(e1))
    warm_avg = {k: float(np.median([x for x in v if x >= 0])) if any(x >= 0 for x in v) else -1.0 for k, v in ktimes.items()}
    dom_warm = max(warm_avg, key=lambda k: warm_avg[k])
    # (the LARGE tier's chain and the MEDIUM tier's are within a few per cent of each other in a batch alone -- 0.43 against 0.41 -
    # 0.44 ms -- and took turns at being "the longest" from run to run: the LARGE chain is the roofline kernel unless another one
    # is longer by more than a tenth, so that two runs report the same kernel)
    if warm_avg.get("pd_tier_large", -1.0) >= 0.9 * warm_avg[dom_warm]:
        dom_warm = "pd_tier_large"
    g.set_timing(False)
    lat_plain = []
    for s in range(5):                                                # latency without any kernel events
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        leg_pi(s)
        e1.record()
        torch.cuda.synchronize()
        lat_plain.append(e0.elapsed_time(e1))
    if not args.sync_batches:
        # the second workspace of the handle (buffers, streams) comes into being with the first batch that lands on it
        for s in range(4):
            leg_pi_submit(s, args.rotate_batches)
        g.join()
        torch.cuda.synchronize()
    if not args.no_kernel_events:
        # (a timed event record holds up the stream it sits on for a few us: every batch carrying the pair costs the region
        # 2 %, so one batch in four does -- `roofline.launches_timed` says how many durations the average is over)
        g.set_timing(True, only=[dom_warm])
        g.set_option("timing_every", 4)

    # ---- timed region: EXACTLY K steps; the two legs of a step are independent (the decode reads resident image rows), so the
    # K image batches go first, back to back, then the K forwards -- no host synchronisation anywhere inside -------------------
    K = args.steps
    ev_pi = [torch.cuda.Event(enable_timing=True) for _ in range(K + 1)]
    ev_lp = [torch.cuda.Event(enable_timing=True) for _ in range(K)]
    barrier()
    t0 = time.perf_counter()
    ev_pi[0].record()
    for s in range(K):
        leg_pi_submit(s, args.rotate_batches)
        if args.sync_batches:
            ev_pi[s + 1].record()
    if not args.sync_batches:
        g.join()                                                        # the stream waits for the batches in flight
        ev_pi[K].record()
    for s in range(K):
        leg_lp()
        ev_lp[s].record()
    barrier()
    wall = time.perf_counter() - t0
    t_pi = ev_pi[0].elapsed_time(ev_pi[K]) * 1e-3
    t_lp = ev_pi[K].elapsed_time(ev_lp[K - 1]) * 1e-3
    pi_steps = [ev_pi[s].elapsed_time(ev_pi[s + 1]) for s in range(K)] if args.sync_batches else [t_pi * 1e3 / K]
    lp_steps = [(ev_pi[K] if s == 0 else ev_lp[s - 1]).elapsed_time(ev_lp[s]) for s in range(K)]
    ktimes_timed = [] if args.no_kernel_events else g.timing_history(dom_warm, cap=min((K + 3) // 4, 64))
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([wall, t_pi, t_lp], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, t_pi, t_lp = [float(v) for v in t.tolist()]
    g.set_timing(False)
    g.set_option("timing_every", 1)
    # What the timed region WROTE (three batches in flight, second halves deferred, LARGE tier + divide and conquer + early pass
    # active) against the stream-ordered call on the same pairs: the last min(K, 3) steps' buffers, bit for bit, statuses too.
    # The snapshot of the last step's rows is what `cpu_baseline` checks against the oracle below.
    timed_equal, timed_checked = True, 0
    timed_snapshot = None
    for s_ in range(max(K - 3, 0), K):
        b_ = rot[s_ % len(rot)] if args.rotate_batches else pi_pairs
        ref_o, ref_s = g.pd_pi_batch(b_, hop)
        torch.cuda.synchronize()
        timed_equal = timed_equal and bool(torch.equal(pi_outs[s_ % 3], ref_o)) and bool(torch.equal(pi_sts[s_ % 3], ref_s))
        timed_checked += 1
        if s_ == K - 1:
            timed_snapshot = (pi_outs[s_ % 3].clone(), pi_sts[s_ % 3].clone(), b_)
        del ref_o, ref_s
    if world > 1:
        import torch.distributed as dist
        te_ = torch.tensor([1.0 if timed_equal else 0.0], dtype=torch.float64, device=dev)
        dist.all_reduce(te_, op=dist.ReduceOp.MIN)
        timed_equal = bool(te_.item() > 0.5)
        # every rank's own device time of its K image batches: proof that N ranks ran, and how evenly
        pr_ = torch.zeros(world, dtype=torch.float64, device=dev)
        pr_[rank] = ev_pi[0].elapsed_time(ev_pi[K])
        dist.all_reduce(pr_, op=dist.ReduceOp.SUM)
        rank_pi_ms = [float(v) for v in pr_.tolist()]
        dist_info = {"backend": dist.get_backend(), "world_size": int(dist.get_world_size()), "rank_pi_region_ms": rank_pi_ms,
                     "devices": None}
        names = [None] * world
        dist.all_gather_object(names, "%s cuda:%d" % (torch.cuda.get_device_name(dev), local_rank))
        dist_info["devices"] = names
    else:
        dist_info = {"backend": None, "world_size": 1, "rank_pi_region_ms": [ev_pi[0].elapsed_time(ev_pi[K])],
                     "devices": ["%s cuda:%d" % (torch.cuda.get_device_name(dev), local_rank)]}

    # ---- the other mode of the same region (same batch every step <-> rotating batches), K steps, reported beside `value` ----
    def pi_region(rotate):
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        barrier()
        ev0.record()
        for s in range(K):
            leg_pi_submit(s, rotate)
        if not args.sync_batches:
            g.join()
        ev1.record()
        barrier()
        t = torch.tensor([ev0.elapsed_time(ev1) * 1e-3], dtype=torch.float64, device=dev)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    t_other = pi_region(not args.rotate_batches)
    g.pd_pi_batch(pi_pairs, hop, out=pi_out, status=pi_status)          # the headline batch's header for sizes()/stats() below
    torch.cuda.synchronize()

    # ---- strong scaling (every N): ONE list -- the non-edges of the graph inside each other's hop-ball, the part of the
    # reference's negative sweep that has non-zero images (loaddatas.py:44-53, riccidist2dgm.py:362-370) -- cut into N
    # contiguous shards of equal estimated cost, each rank runs its shard, the image rows are exchanged (one all-gather) ------
    strong = None
    if not args.no_sweep:
        try:
            ci = engine.ComplementIndex(wl["rowptr"], wl["col"], device=local_rank)
            near, ranks = engine.near_pairs(ci, hop)
            near = near[torch.argsort(ranks)].contiguous()                                # list order: the same on every rank
            near_np = near.cpu().numpy()
            cost = tdist.pair_cost(tdist.ball_bound(wl["rowptr"], wl["col"], hop), near_np)
            run = lambda shard: g.pd_pi_batch(shard.contiguous(), hop)
            gather = tdist.gather_shards if world > 1 else None
            reps, dev_ms = [], []

            def run_timed(shard):                                   # the rank's own device time, beside the host wall clock
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                res = run(shard)
                e1.record()
                dev_ms.append((e0, e1))
                return res
            for it in range(6):
                barrier()
                c0 = time.perf_counter()
                rows_s, st_s, part = tdist.pd_pi_batch_sharded(run_timed, near, world, rank, cost=cost, gather=gather)
                barrier()
                reps.append(time.perf_counter() - c0)
            my_ms = float(np.median([a.elapsed_time(b) for a, b in dev_ms[1:]]))
            strong_runs = [float(r) for r in reps]
            tt = torch.tensor([float(np.median(reps[1:]))], dtype=torch.float64, device=dev)
            per_rank = torch.zeros(world, dtype=torch.float64, device=dev)
            per_rank[rank] = my_ms
            if world > 1:
                import torch.distributed as dist
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                dist.all_reduce(per_rank, op=dist.ReduceOp.SUM)
            sdt = float(tt.item())
            parts = tdist.shard_pairs_interleaved(cost, world) if world > 1 else [np.arange(len(near_np))]
            strong = {"pairs": int(len(near_np)), "seconds": sdt, "images_per_sec": len(near_np) / sdt,
                      "shard_pairs": [int(len(p_)) for p_ in parts],
                      "shard_cost": [float(cost[p_].sum()) for p_ in parts],
                      "rank_device_ms": [float(v) for v in per_rank.tolist()], "runs_s": strong_runs,
                      "rows_gathered_on_every_rank": bool(world > 1), "nonzero_rows": int((rows_s.abs().sum(1) > 0).sum()),
                      "note": "strong scaling: fixed list (all non-adjacent pairs with d(u,v) <= hop) dealt to the ranks by descending "
                              "estimated cost (cost = smaller ball-size bound of the endpoints; boustrophedon, so no rank owns the "
                              "heavy tail), rows gathered back to list order with one all-gather; `seconds`: host wall clock, median "
                              "of 5 (the first call, which sizes the buffers, left out; `runs_s` has all six of rank 0), max over ranks; rank_device_ms: every rank's own HIP-event time of its shard"}
            del rows_s, st_s, near, ranks, ci
            g.pd_pi_batch(pi_pairs, hop, out=pi_out, status=pi_status)
            torch.cuda.synchronize()
        except Exception as ex:
            strong = {"error": repr(ex)}

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
    pdgnn_amazon = None
    if rank == 0 and not args.no_sweep:
        try:
            pdgnn_amazon = pdgnn_amazon_aux(torch, dev)
        except Exception as ex:
            pdgnn_amazon = {"error": repr(ex)}
    # auxiliary (untimed): the LP leg's two bounded kernels on their own -- the feature GEMM against the f32 MFMA peak and
    # the scatter-add SpMM against HBM (north_star); torch events on the current stream, which is where ops.* enqueue
    lp_roof = None
    if rank == 0:
        def _avg_us(fn, reps=20):
            """device time per call: `reps` calls replayed from a captured graph where that works (one rank, no collective inside) -- a
            call through the Python wrappers takes the host 10 - 25 us, longer than most of these kernels run, so that timed over
            eager calls a 16 us kernel read 19.8 us on a box with a slow host -- else over eager calls"""
            fn()
            torch.cuda.synchronize()
            if world == 1:
                try:
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        fn()
                    torch.cuda.current_stream().wait_stream(side)
                    torch.cuda.synchronize()
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr):
                        for _ in range(reps):
                            fn()
                    gr.replay()
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        gr.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    return e0.elapsed_time(e1) / (3 * reps) * 1e3
                except Exception:
                    torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps * 1e3
        Mr, Kf, Nh = x_full.shape[0], x_full.shape[1], w1.shape[1]
        xw = torch.empty((Mr, Nh), dtype=torch.float32, device=dev)
        us_g = _avg_us(lambda: ops.gemm(x_full, w1, out=xw))
        hfull = torch.empty((n, Nh), dtype=torch.float32, device=dev).normal_()
        yfull = torch.empty((n, Nh), dtype=torch.float32, device=dev)
        us_s = _avg_us(lambda: ops.spmm(rowptr_n, col_n, val_n, hfull, bias=b1, relu=True, out=yfull))
        nnz = int(col_n.shape[0])
        spmm_bytes = nnz * (Nh * 4 + 8) + n * (Nh * 4 + 4)            # gathered rows + col/val + output rows + rowptr
        xs_full = ops.SparseRows(x_full)
        us_sg = _avg_us(lambda: ops.sparse_gemm(xs_full, w1, out=xw))
        sp_bytes = xs_full.nnz * 8.0 + 4.0 * (Mr + 1) + 4.0 * Kf * Nh + 4.0 * Mr * Nh          # CSR once per column slice is L2 traffic
        # what the kernel is bound by: a step of a row (four entries, one per 16-lane row) is one ds_read_b128 of the whole wavefront
        # = 1 KiB of the weight slice in LDS, per column slice; the LDS pipe moves 128 B per clock and CU
        row_cnt = (xs_full.rowptr[1:] - xs_full.rowptr[:-1]).long()
        sp_slices = (Nh + 63) // 64
        sp_lds_bytes = float(((row_cnt + 3) // 4).sum().item()) * sp_slices * 1024.0
        LDS_PEAK_GBS = 256 * 128 * 2.4                                                          # 256 CUs x 128 B/clk x 2.4 GHz = 78.6 TB/s
        lp_roof = {"feature_gemm_sparse": {"bound": "hbm", "shape_mkn": [int(Mr), int(Kf), int(Nh)], "nnz": int(xs_full.nnz),
                                           "density": xs_full.density, "kernel_us": us_sg,
                                           "achieved": sp_bytes / us_sg / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": sp_bytes / us_sg / 1e3 / HBM_PEAK_GBS,
                                           "useful_tflops": 2.0 * xs_full.nnz * Nh / us_sg / 1e6,
                                           "lds": {"bytes": sp_lds_bytes, "achieved": sp_lds_bytes / us_sg / 1e3, "peak": LDS_PEAK_GBS, "unit": "GB/s",
                                                   "frac": sp_lds_bytes / us_sg / 1e3 / LDS_PEAK_GBS,
                                                   "note": "LDS reads of the weight slice over the WHOLE kernel time (launch, staging and the "
                                                           "start-up chain included): the resource the kernel is bound by"},
                                           "note": "x @ W over the stored entries of x (tlc_spgemm_csr_dense_f32); bytes = CSR + W + "
                                                   "output, each once (bound in fact by the LDS reads of W's slice: DESIGN.md).  Used by Net.encode / "
                                                   "the LP leg below %.0f %% density (the dense MFMA kernel is faster above): %s on this workload"
                                                   % (100 * TLCGNN.Net.SPARSE_FEATURES_BELOW, "used" if xs_full.density < TLCGNN.Net.SPARSE_FEATURES_BELOW else "not used")},
                   "feature_gemm": {"bound": "mfma", "shape_mkn": [int(Mr), int(Kf), int(Nh)], "kernel_us": us_g,
                                    "achieved": 2.0 * Mr * Kf * Nh / us_g / 1e6, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                    "frac": 2.0 * Mr * Kf * Nh / us_g / 1e6 / MFMA_F32_PEAK_TFLOPS, "dtype": "f32 (v_mfma_f32_16x16x4_f32)"},
                   "scatter_add_spmm": {"bound": "hbm", "rows": int(n), "nnz": nnz, "k": int(Nh), "kernel_us": us_s,
                                        "achieved": spmm_bytes / us_s / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                        "frac": spmm_bytes / us_s / 1e3 / HBM_PEAK_GBS}}
        # round 5: the kernels the leg actually runs behind the feature GEMM -- conv1's aggregate with conv2's projection in its
        # epilogue (tlc_gcn2_encode_f32 minus the GEMM), the k = 16 aggregate, and the decode on the f32 MFMA over the float32 table
        ws_e = torch.empty(((2 * Nh + 16) * n + 12,), dtype=torch.float32, device=dev)
        emb_e = torch.empty((n, 16), dtype=torch.float32, device=dev)
        us_enc = _avg_us(lambda: ops.gcn2_encode(rowptr_n, col_n, val_n, x_full, w1, b1, w2, b2, relu=True, renorm=True, out=emb_e))
        h16 = torch.empty((n, 16), dtype=torch.float32, device=dev).normal_()
        y16 = torch.empty((n, 16), dtype=torch.float32, device=dev)
        us_s16 = _avg_us(lambda: ops.spmm(rowptr_n, col_n, val_n, h16, bias=b2, relu=True, renorm=True, out=y16))
        us_dec = _avg_us(lambda: ops.lp_decode(dec_pairs, emb_e, dec_pi, l1w, l1b, l2w, l2b, out=prob))
        Ed = int(dec_pairs.shape[0])
        s16_bytes = nnz * (16 * 4 + 8) + n * (16 * 4 + 4)
        fused_bytes = nnz * (Nh * 4 + 8) + n * (16 * 4 + 4) + Nh * 16 * 4
        dec_bytes = Ed * (8 + 2 * 16 * 4 + dec_pi.shape[1] * dec_pi.element_size() + 4)
        lp_roof["scatter_add_spmm_k16"] = {"bound": "hbm", "rows": int(n), "nnz": nnz, "k": 16, "kernel_us": us_s16,
                                           "achieved": s16_bytes / us_s16 / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                           "frac": s16_bytes / us_s16 / 1e3 / HBM_PEAK_GBS}
        lp_roof["encode_minus_feature_gemm"] = {"bound": "hbm", "kernel_us": us_enc - us_g, "algorithmic_bytes": int(fused_bytes + s16_bytes),
                                                "achieved": (fused_bytes + s16_bytes) / max(us_enc - us_g, 1e-3) / 1e3, "peak": HBM_PEAK_GBS,
                                                "unit": "GB/s", "frac": (fused_bytes + s16_bytes) / max(us_enc - us_g, 1e-3) / 1e3 / HBM_PEAK_GBS,
                                                "encode_us": us_enc,
                                                "note": "tlc_gcn2_encode_f32 (three launches: feature GEMM, conv1 aggregate + bias + ReLU + @W2 in one "
                                                        "kernel, k = 16 aggregate + bias + ReLU + renorm) minus the feature GEMM timed alone"}
        lp_roof["decode"] = {"bound": "hbm", "pairs": Ed, "image_table_dtype": str(dec_pi.dtype).replace("torch.", ""), "kernel_us": us_dec,
                             "algorithmic_bytes": int(dec_bytes), "achieved": dec_bytes / us_dec / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": dec_bytes / us_dec / 1e3 / HBM_PEAK_GBS,
                             "note": "lp_decode_mfma_kernel: pair ids + two 64-byte embedding rows + the image row + the probability per pair; "
                                     "back-to-back launches timed with events carry ~5 us of launch each (rocprofv3 durations: profiles/)"}
        del xw, hfull, yfull, ws_e, emb_e, h16, y16
        # the stand-alone PI raster (tlc_pi_raster = PersistenceImager.transform) on diagrams shaped like this batch's:
        # one diagram per pair with as many points as the vicinity has edges (Ord0 + ext0 + Ext1 points), values in [0,1]
        g.pd_pi_batch(pi_pairs, hop, out=pi_out, status=pi_status)
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
        # per-pair byte model and tier of every pair of the batches the timed region ran
        used = [rot[j] for j in range(min(K, len(rot)))] if args.rotate_batches else [pi_pairs]
        weight = [len(range(j, K, len(rot))) for j in range(len(used))] if args.rotate_batches else [K]
        dom = dom_warm
        live = [x for x in ktimes_timed if x >= 0]
        dom_bytes = dom_units = all_bytes = 0.0
        for b_dev, wgt in zip(used, weight):
            g.pd_pi_batch(b_dev, hop, out=pi_out, status=pi_status)
            n_sz, m2_sz = g.sizes(E)
            tiers = engine.tier_of(n_sz, m2_sz)
            bpp = engine.algorithmic_bytes(wl["rowptr"], wl["col"], b_dev.cpu().numpy(), hop)
            sel = (tiers == dom) if dom.startswith("pd_tier") else np.ones(E, dtype=bool)
            if dom == "pd_tier_medium" and not args.sync_batches and live:
                # (pipelined chunks do not split the MEDIUM tier by Pos-edge count: the timed launches took all of it)
                sel = (tiers == "pd_tier_medium") | (tiers == "pd_tier_medium_rest")
            dom_bytes += wgt * float(bpp[sel].sum()) / K
            dom_units += wgt * float(sel.sum()) / K
            all_bytes += wgt * float(bpp.sum()) / K
        g.pd_pi_batch(pi_pairs, hop, out=pi_out, status=pi_status)
        stats = g.stats()
        kavg = dict(warm_avg)                                   # all kernels: warm-up steps (every kernel carried events there)
        if live:
            kavg[dom] = float(np.mean(live))                    # the dominant kernel: live, inside the timed region
        achieved = dom_bytes / (kavg[dom] * 1e-3) / 1e9
        traffic, traffic_source = None, None
        if pre_traffic is not None and dom in pre_traffic:
            d_ = pre_traffic_detail[dom]
            traffic = float(pre_traffic[dom])
            traffic_source = ("measured in this run: two child passes of this script under rocprofv3 --pmc before the timed region "
                              "(FETCH_SIZE, WRITE_SIZE; 3 batches each); bytes per batch of this kernel's launches = (FETCH_SIZE %.0f KB + "
                              "WRITE_SIZE %.0f KB) * 1024; gfx950: FETCH_SIZE tallies 64 B per 128-B request of a wide stream and is "
                              "uncalibrated for 4/8-byte gathers, so the read side is a lower bound" % (d_["FETCH_SIZE_KB"], d_["WRITE_SIZE_KB"]))
        elif pre_traffic is None:
            traffic_source = str(pre_traffic_detail)
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if traffic is None and os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                traffic = tj.get(dom)
                why = (" [in-run measurement: %s]" % traffic_source) if traffic_source else ""
                traffic_source = "NOT measured in this run: profiles/pmc_traffic.json (%s), rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE " \
                                 "passes over this command; bytes per launch%s" % (tj.get("_collected", "round-1 profile"), why)
            except Exception:
                traffic = None
        # what actually bounds the image leg: vector issue.  VALU wave-instructions per batch (counter pass above) x 4 cycles (a
        # wave64 instruction occupies its SIMD16 for four cycles) against the 1 024 SIMDs x clock x the time a batch takes.
        roofline_issue = None
        if pre_issue is not None:
            props = torch.cuda.get_device_properties(dev)
            clk = float(getattr(props, "clock_rate", 2400000)) * 1e3                      # Hz (the device's maximum shader clock)
            simds = int(props.multi_processor_count) * 4
            valu = float(pre_issue.get("SQ_INSTS_VALU", 0.0))
            per_batch_s = t_pi / K
            roofline_issue = {"bound": "valu-issue", "valu_wave_insts_per_batch": valu, "lds_wave_insts_per_batch": pre_issue.get("SQ_INSTS_LDS"),
                              "salu_wave_insts_per_batch": pre_issue.get("SQ_INSTS_SALU"), "cycles_per_wave_inst": 4,
                              "simds": simds, "clock_hz": clk, "clock_source": "device property (maximum shader clock)",
                              "achieved": valu * 4.0 / per_batch_s, "peak": simds * clk, "unit": "SIMD-cycles/s",
                              "frac": valu * 4.0 / (simds * clk * per_batch_s),
                              "frac_one_batch_alone": valu * 4.0 / (simds * clk * float(np.median(lat_plain)) * 1e-3),
                              "source": "measured in this run: child pass under rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU, "
                                        "%d batches; sums over every kernel of the image batch" % pre_issue_detail["batches_in_pass"],
                              # issue rates measured on this GPU with tools/probes/salu_probe.hip (profiles/r04_issue_rates_probe.txt):
                              # 1.33 vector and 0.86 scalar-ALU wave-instructions per cycle and CU (one scalar unit per CU), independent
                              "measured_issue_rates_per_cycle_and_cu": {"valu": 1.33, "salu": 0.86},
                              "frac_valu_of_measured_rate": valu / (1.33 * (simds / 4) * clk * per_batch_s),
                              "frac_salu_of_measured_rate": float(pre_issue.get("SQ_INSTS_SALU", 0.0)) / (0.86 * (simds / 4) * clk * per_batch_s),
                              "per_kernel_valu_wave_insts": pre_issue_detail["per_kernel_valu"],
                              # every instruction class (second counter pass): SQ_INSTS counts all wave-instructions; the classes do
                              # not share one issue port (a SIMD issues up to one instruction PER CLASS per cycle, from different
                              # wavefronts), so this fraction is an upper bound of how full any one port can be
                              "all_wave_insts_per_batch": pre_issue.get("SQ_INSTS"),
                              "smem_wave_insts_per_batch": pre_issue.get("SQ_INSTS_SMEM"),
                              "vmem_rd_wave_insts_per_batch": pre_issue.get("SQ_INSTS_VMEM_RD"),
                              "vmem_wr_wave_insts_per_batch": pre_issue.get("SQ_INSTS_VMEM_WR"),
                              "branch_wave_insts_per_batch": pre_issue.get("SQ_INSTS_BRANCH"),
                              "frac_all_classes": (float(pre_issue["SQ_INSTS"]) * 4.0 / (simds * clk * per_batch_s)) if pre_issue.get("SQ_INSTS") else None,
                              "second_pass": pre_issue_detail.get("second_pass", "ok"),
                              # tools/probes/mix_probe.hip (profiles/r05_mix_probe.txt): a loop with the extraction's own instruction mix
                              # issues 0.37 / 0.73 / 1.30 / 1.72 wave-instructions per cycle and CU at 1 / 2 / 4 / 8 wavefronts per SIMD --
                              # the CU's ceiling for such code is ~1.75, and below ~6 wavefronts per SIMD the rate is latency, not a port
                              "measured_mix_ceiling_per_cycle_and_cu": 1.75,
                              "all_classes_per_cycle_and_cu": (float(pre_issue["SQ_INSTS"]) / ((simds / 4) * clk * per_batch_s)) if pre_issue.get("SQ_INSTS") else None,
                              "frac_all_classes_of_measured_ceiling": (float(pre_issue["SQ_INSTS"]) / (1.75 * (simds / 4) * clk * per_batch_s)) if pre_issue.get("SQ_INSTS") else None,
                              "note": "share of the machine's vector issue slots the image leg uses in the timed region (pipelined) / for one "
                                      "batch alone; the rest is dependent-latency time (LDS / L2 round trips of serial graph code)"}
        else:
            roofline_issue = {"error": str(pre_issue_detail)}
        out = {
            "metric": "persistence-images/sec + LP-forward edges/sec, PubMed-scale, 1/2/4/8 GPU",
            "value": world * E * K / t_pi,
            "unit": "persistence-images/sec",
            "lp_forward_edges_per_sec": world * dec_pairs.shape[0] * K / t_lp,
            "n_gpus": world, "steps": K, "warmup": args.warmup,
            "ms_per_step": wall / K * 1e3,
            "pi_ms_per_step": t_pi / K * 1e3, "lp_ms_per_step": t_lp / K * 1e3,
            "pi_ms_per_step_median": float(np.median(pi_steps)), "lp_ms_per_step_median": float(np.median(lp_steps)),
            "pi_latency_ms": float(np.median(lat_plain)),
            "timed_outputs_equal": bool(timed_equal), "timed_outputs_checked": int(timed_checked),
            "dist": dist_info,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "PubMed-shaped synthetic graph (N=19717, M=44324, F=500, seed 1234), hop=2; per GPU and step: "
                                   "PI-A = %s -> 5x5 persistence images, and one TLCGNN forward "
                                   "(GCN 500->100->16 encode + fused decode of %d pairs, image rows resident)"
                                   % ("a different %d-pair sample of the graph's positive pairs every step (--rotate-batches)" % E
                                      if args.rotate_batches else "all %d train-positive pairs" % E, dec_pairs.shape[0]),
                       "timed_region": ("K image batches as stream-ordered tlc_pd_pi_batch calls (one at a time), " if args.sync_batches else
                                        "K image batches submitted with tlc_pd_pi_batch_async (three in flight on the handle's "
                                        "workspaces, a batch's tier launches submitted behind the next batch's first half; three output buffers in turn) and ONE join behind the last, ") +
                                       "then K forwards, no host synchronisation inside (throughput); pi_latency_ms = one "
                                       "stream-ordered batch with a synchronisation after it",
                       "pi_submit": "sync" if args.sync_batches else "async",
                       "pairs_per_gpu": E, "decode_pairs_per_gpu": int(dec_pairs.shape[0]), "rotate_batches": bool(args.rotate_batches),
                       "parallelism": "every rank its own batch (weak; no collective in the image leg); encoder: " + enc_mode,
                       "vicinity_tiers": {k: int(v) for k, v in stats.items() if k.startswith("tier")},
                       "tie_fallback_sources": int(stats["tie_fallback_sources"])},
            "roofline": {"bound": "hbm", "kernel": dom, "kernel_ms": kavg[dom], "launches_timed": len(live), "units_per_launch": dom_units,
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source,
                         "kernel_ms_one_batch_alone": warm_avg.get(dom, -1.0),
                         # busy time of the chain (review of round 4: the span of a two-kernel chain held the queueing between its
                         # kernels).  Since round 5 the LARGE tier runs its divide and conquer in place: the chain IS one kernel, and the
                         # events bracket that one launch on the stream it runs on.
                         "kernel_busy_ms": kavg[dom], "chain_kernels": 1 if dom == "pd_tier_large" else None,
                         "batches_in_flight": 1 if args.sync_batches else 3,
                         "busy_within_batches_in_flight": bool(kavg[dom] <= (t_pi / K * 1e3) * (1 if args.sync_batches else 3)),
                         "note": "kernel_ms: this kernel chain's launches INSIDE the timed region, where three batches are in flight and "
                                 "its whole-CU workgroups wait for room among the other batches' kernels; kernel_ms_one_batch_alone: "
                                 "the same chain in a stream-ordered batch with the machine to itself (warm-up steps)"},
            "roofline_issue": roofline_issue,
            "roofline_chain": {"bound": "hbm", "algorithmic_bytes_per_pi": all_bytes / E,
                               "achieved": all_bytes * world * K / t_pi / 1e9, "peak": HBM_PEAK_GBS * world,
                               "unit": "GB/s", "frac": all_bytes * K / t_pi / 1e9 / HBM_PEAK_GBS},
            ("same_batch_every_step" if args.rotate_batches else "rotated_batches"): {
                "value": world * E * K / t_other, "pi_ms_per_step": t_other / K * 1e3,
                "note": "the same K-step region with %s" % ("the one train-positive batch every step" if args.rotate_batches else
                                                            "8 different 37 676-pair samples of the positive pairs in turn: the "
                                                            "previous chunk's sizes never match exactly")},
            "encoder": {"mode": enc_mode, "lp_submit": lp_submit, "lp_leg_ms": encoder_ms,
                        "note": "LP leg (encode + decode) timed per mode before the timed region, max over ranks; N=1 has no exchange"},
            "strong_scaling": strong,
            "kernel_ms": kavg,
            "kernel_ms_note": "'%s' (the roofline kernel): HIP events kept by the library in a ring and read after the timed region, "
                              "mean of %d steps; the others: median of the %d warm-up steps (all eight event pairs inside the timed "
                              "region cost 48 us per batch)" % (dom, len(live), max(args.warmup, 1)),
            "roofline_lp": lp_roof,
            "sweep": sweep,
            "full_sweep": full_sweep,
            "ricci": ricci,
            "other_shapes": other_shapes,
            "pdgnn": pdgnn,
            "pdgnn_amazon": pdgnn_amazon,
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
            # the rows the TIMED REGION wrote (its last step's buffer), not a later stream-ordered call
            snap_o, snap_s, snap_b = timed_snapshot
            if args.rotate_batches:
                ref, rst, _ = oracle.pd_pi_batch(wl["rowptr"], wl["col"], wl["w"], snap_b[: len(sample)].cpu().numpy(), hop, n_threads=0)
            got = snap_o[: len(sample)].cpu().numpy()
            nz = ref != 0
            out["cpu_baseline"]["max_rel_diff_vs_gpu"] = float((np.abs(got[nz] - ref[nz]) / np.abs(ref[nz])).max()) if nz.any() else 0.0
            out["cpu_baseline"]["status_equal_vs_gpu"] = bool(np.array_equal(snap_s[: len(sample)].cpu().numpy(), rst))
            out["cpu_baseline"]["checked"] = "rows written by the last step of the timed region (pipelined submission)"
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

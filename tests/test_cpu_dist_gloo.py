"""CPU, 2 / 3 / 8 processes, gloo: the N>1 layout of the LP forward (node-row sharded encoder + one all-gather per layer,
pair-sharded decode) reproduces the single-rank result.  The compute callables are the torch fp32 restatement
(the checker) because HIP kernels cannot run here; what is under test is the sharding/exchange logic of dist.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _csr_from_norm(ei, norm, n):
    order = torch.argsort(ei[1], stable=True)
    col = ei[0][order].to(torch.int32)
    val = norm[order]
    rowptr = torch.zeros(n + 1, dtype=torch.int32)
    rowptr[1:] = torch.cumsum(torch.bincount(ei[1], minlength=n), 0).to(torch.int32)
    return rowptr, col, val


def _spmm_cpu(rowptr, col, val, x, bias, relu, renorm=False, out=None):
    n = rowptr.numel() - 1
    rows = torch.repeat_interleave(torch.arange(n), (rowptr[1:] - rowptr[:-1]).long())
    res = torch.zeros(n, x.shape[1]).index_add_(0, rows, val[:, None] * x[col.long()])
    res = res + bias
    res = torch.relu(res) if relu else res
    if out is not None:
        out.copy_(res)
        return out
    return res


def _gemm_cpu(a, b, out=None):
    if out is not None:
        return torch.matmul(a, b, out=out)
    return a @ b


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from tlc_gnn_amd import dist as tdist, synth
    from oracle import lp_forward_ref as ref
    torch.manual_seed(0)
    n, F_ = 211, 37
    edges = synth.holme_kim_edges(n, 600, seed=4)
    ei = torch.from_numpy(np.concatenate([edges, edges[:, ::-1]]).T.copy()).long()
    x = torch.from_numpy(synth.synthetic_features(n, F_, seed=4))
    w1, b1 = torch.randn(F_, 20) * 0.2, torch.randn(20) * 0.1
    w2, b2 = torch.randn(20, 16) * 0.2, torch.randn(16) * 0.1
    rei, norm = ref.gcn_norm(ei, n)
    rowptr, col, val = _csr_from_norm(rei, norm, n)
    full = ref.tlcgnn_encode(x, ei, w1, b1, w2, b2)
    # replicated encoder: the whole forward on every rank, no collective
    rep = tdist.ShardedGCNEncoder(rowptr, col, val, n, world, rank, gemm=_gemm_cpu, spmm=_spmm_cpu, mode="replicated")
    emb_rep = rep.encode(x, w1, b1, w2, b2)
    assert (emb_rep - full).abs().max() < 1e-5 and rep.row_map(torch.arange(n)).equal(torch.arange(n))
    # node-row sharded encoder: one all-gather per layer into the padded block layout, read in place through row_map
    enc = tdist.ShardedGCNEncoder(rowptr, col, val, n, world, rank, gemm=_gemm_cpu, spmm=_spmm_cpu)
    for _ in range(2):                                  # twice: the persistent exchange buffers are reused
        emb_padded = enc.encode(x[enc.lo:enc.hi].contiguous(), w1, b1, w2, b2)
    assert emb_padded.shape[0] == world * enc.rows.blk
    emb = emb_padded[enc.row_map(torch.arange(n))]
    # pair-sharded decode: every rank owns a contiguous shard, no collective
    pairs = torch.randint(0, n, (501, 2))
    pi = torch.rand(501, 25, dtype=torch.float64)
    l1w, l1b, l2w, l2b = torch.randn(25, 41) * 0.2, torch.zeros(25), torch.randn(1, 25) * 0.2, torch.zeros(1)
    lo, hi = tdist.shard_bounds(501, world, rank)
    prob = ref.tlcgnn_decode(emb.clone(), pairs[lo:hi], pi[lo:hi], l1w, l1b, l2w, l2b)
    prob_full = ref.tlcgnn_decode(full.clone(), pairs, pi, l1w, l1b, l2w, l2b)
    # the same decode reading the padded buffer through remapped pair ids
    prob_p = ref.tlcgnn_decode(emb_padded.clone(), enc.row_map(pairs[lo:hi]), pi[lo:hi], l1w, l1b, l2w, l2b)
    assert torch.equal(prob_p, prob)
    # PD/PI pair shards: cost-balanced contiguous cuts, rows exchanged with ONE all-gather, identical to the single-rank rows
    from oracle import oracle
    from tlc_gnn_amd import synth as _s
    kap = np.random.RandomState(4).uniform(-0.5, 0.9, size=len(edges))
    rp, cl, ww = _s.edges_to_csr(n, edges, kap)
    pr = np.concatenate([edges[:150], np.random.RandomState(5).randint(0, n, size=(60, 2))]).astype(np.int32)
    single_rows, single_st, _ = oracle.pd_pi_batch(rp, cl, ww, pr, 2, n_threads=1)

    def run(shard):                                     # the checker stands in for the HIP path (no GPU in this test)
        r_, s_, _ = oracle.pd_pi_batch(rp, cl, ww, np.ascontiguousarray(shard), 2, n_threads=1)
        return torch.from_numpy(r_), torch.from_numpy(s_)

    cost = tdist.pair_cost(tdist.ball_bound(rp, cl, 2), pr)
    rows, st, (slo, shi) = tdist.pd_pi_batch_sharded(run, pr, world, rank, cost=cost, gather=tdist.gather_shards, scheme="contiguous")
    assert 0 <= shi - slo < len(pr)
    assert np.array_equal(rows.numpy(), single_rows) and np.array_equal(st.numpy(), single_st)      # bit for bit
    # the default: pairs dealt to the ranks by descending cost; the rows come back to list order through the index arrays
    rows, st, mine = tdist.pd_pi_batch_sharded(run, pr, world, rank, cost=cost, gather=tdist.gather_shards)
    parts = tdist.shard_pairs_interleaved(cost, world)
    assert np.array_equal(mine, parts[rank]) and 0 < len(mine) < len(pr)
    sizes = [len(q_) for q_ in parts]
    assert max(sizes) - min(sizes) <= 1                                                              # uneven remainders: at most one apart
    assert np.array_equal(np.sort(np.concatenate(parts)), np.arange(len(pr)))                       # a partition of the list
    heavy = np.argsort(-cost, kind="stable")[:world]                                                 # the heaviest pairs: one per rank
    assert sorted(int(np.flatnonzero([h in p for p in parts])[0]) for h in heavy) == list(range(world))
    sums = [cost[q_].sum() for q_ in parts]
    assert max(sums) - min(sums) <= cost.max()
    assert np.array_equal(rows.numpy(), single_rows) and np.array_equal(st.numpy(), single_st)      # bit for bit
    # fewer pairs than ranks: some shards are EMPTY (their run() sees zero pairs, their block of the gather is all padding)
    few = pr[:2] if world > 2 else pr[:1]
    few_rows, few_st, _ = oracle.pd_pi_batch(rp, cl, ww, few, 2, n_threads=1)
    for scheme in ("interleaved", "contiguous"):
        rows, st, part = tdist.pd_pi_batch_sharded(run, few, world, rank, cost=cost[:len(few)], gather=tdist.gather_shards, scheme=scheme)
        assert rows.shape == (len(few), 25) and np.array_equal(rows.numpy(), few_rows) and np.array_equal(st.numpy(), few_st), scheme
    # PaddedRows with rem != 0 (211 rows over 2 / 3 / 8 ranks): the padded row of every node is inside its owner's block
    pr_rows = tdist.PaddedRows(n, world, rank)
    padded = pr_rows.remap(torch.arange(n))
    for r_ in range(world):
        lo_r, hi_r = tdist.shard_bounds(n, world, r_)
        assert padded[lo_r:hi_r].equal(torch.arange(hi_r - lo_r) + r_ * pr_rows.blk)
    q.put((rank, float((emb - full).abs().max()), float((prob - prob_full[lo:hi]).abs().max()), tuple(emb.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_forward_gloo(world):
    """world 3 and 8: uneven interleaved shards, node blocks with a remainder (211 rows), empty pair shards."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    try:
        res = [q.get(timeout=600) for _ in range(world)]
    finally:
        for p in procs:
            p.join(60)
            if p.is_alive():
                p.kill()
    for p in procs:
        assert p.exitcode == 0
    for rank, e_emb, e_prob, shape in res:
        assert shape == (211, 16)
        assert e_emb < 1e-5 and e_prob < 1e-6, (rank, e_emb, e_prob)


def test_single_rank_layout_is_the_identity():
    from tlc_gnn_amd import dist as tdist
    x = torch.arange(12.0).view(4, 3)
    assert tdist.gather_shards(x, [0, 4]) is x
    rows = tdist.PaddedRows(4, 1, 0)
    assert rows.remap(torch.arange(4)).equal(torch.arange(4)) and rows.blk == 4
    got = tdist.pd_pi_batch_sharded(lambda p: (p * 2, p[:, 0]), x, 1, 0, cost=np.ones(4))
    assert got[2] == (0, 4) and torch.equal(got[0], x * 2)

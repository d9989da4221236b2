"""CPU, 2 processes, gloo: the N>1 layout of the LP forward (node-row sharded encoder + one all-gather per layer,
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


def _spmm_cpu(rowptr, col, val, x, bias, relu):
    n = rowptr.numel() - 1
    rows = torch.repeat_interleave(torch.arange(n), (rowptr[1:] - rowptr[:-1]).long())
    out = torch.zeros(n, x.shape[1]).index_add_(0, rows, val[:, None] * x[col.long()])
    out = out + bias
    return torch.relu(out) if relu else out


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
    enc = tdist.ShardedGCNEncoder(rowptr, col, val, n, world, rank, gemm=lambda a, b: a @ b, spmm=_spmm_cpu)
    emb = enc.encode(x[enc.lo:enc.hi].contiguous(), w1, b1, w2, b2)
    full = ref.tlcgnn_encode(x, ei, w1, b1, w2, b2)
    # pair-sharded decode: every rank owns a contiguous shard, no collective
    pairs = torch.randint(0, n, (501, 2))
    pi = torch.rand(501, 25, dtype=torch.float64)
    l1w, l1b, l2w, l2b = torch.randn(25, 41) * 0.2, torch.zeros(25), torch.randn(1, 25) * 0.2, torch.zeros(1)
    lo, hi = tdist.shard_bounds(501, world, rank)
    prob = ref.tlcgnn_decode(emb.clone(), pairs[lo:hi], pi[lo:hi], l1w, l1b, l2w, l2b)
    prob_full = ref.tlcgnn_decode(full.clone(), pairs, pi, l1w, l1b, l2w, l2b)
    q.put((rank, float((emb - full).abs().max()), float((prob - prob_full[lo:hi]).abs().max()), tuple(emb.shape)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_forward_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, e_emb, e_prob, shape in res:
        assert shape == (211, 16)
        assert e_emb < 1e-5 and e_prob < 1e-6, (rank, e_emb, e_prob)


def test_all_gather_rows_single_rank_is_identity():
    from tlc_gnn_amd import dist as tdist
    x = torch.arange(12.0).view(4, 3)
    assert tdist.all_gather_rows(x, 4, 1, 0) is x

"""Multi-GPU layout of the path (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI).

The reference is single-process, single-device (no torch.distributed, no DataParallel: SURVEY.md §2.1), so this is
new design, kept to what the path needs:

  * PD/PI per pair and the decode are independent units: the pair list is cut into contiguous shards, the (tiny) CSR is
    replicated, NO data-path collective.
  * The GCN encoder is node-row sharded: every rank projects and aggregates its block of rows and the blocks are
    exchanged with ONE all-gather per layer ([N,100] and [N,16] fp32: ~1 MB per link on PubMed, latency-bound on the
    7 point-to-point xGMI links, hence a single fused collective per layer rather than per-bucket traffic).

The compute callables are injected so that the sharding/exchange logic is testable on CPU with gloo (tests/), while the
product passes the HIP ops.
"""
import os

import numpy as np


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(n_items, world, rank):
    """contiguous, size-balanced shard [lo, hi) of n_items for `rank`."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_pairs_by_cost(cost, world):
    """Cut the pair list into `world` contiguous chunks of roughly equal summed cost (e.g. degree products);
    returns the world+1 boundaries."""
    c = np.cumsum(np.asarray(cost, dtype=np.float64))
    total = c[-1] if len(c) else 0.0
    bounds = [0]
    for r in range(1, world):
        bounds.append(int(np.searchsorted(c, total * r / world)))
    bounds.append(len(c))
    return bounds


def all_gather_rows(local_rows, n_total, world, rank, group=None):
    """All-gather of row blocks laid out by shard_bounds: local [hi-lo, k] -> full [n_total, k] on every rank.

    Blocks are padded to the largest block so that a single all_gather_into_tensor (one RCCL collective) moves them."""
    import torch
    import torch.distributed as dist
    if world == 1:
        return local_rows
    k = local_rows.shape[1]
    blk = (n_total + world - 1) // world
    send = torch.zeros((blk, k), dtype=local_rows.dtype, device=local_rows.device)
    send[:local_rows.shape[0]] = local_rows
    recv = torch.empty((world * blk, k), dtype=local_rows.dtype, device=local_rows.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    out = torch.empty((n_total, k), dtype=local_rows.dtype, device=local_rows.device)
    for r in range(world):
        lo, hi = shard_bounds(n_total, world, r)
        out[lo:hi] = recv[r * blk:r * blk + (hi - lo)]
    return out


class ShardedGCNEncoder:
    """Net.encode (baselines/TLCGNN.py:19-26, eval mode) with node rows sharded over ranks.

    gemm(x_rows, W) -> rows @ W;  spmm(rowptr_local, col, val, X_full, bias, relu) -> aggregated local rows.
    rowptr/col/val: the gcn-normalised CSR by target of the WHOLE graph (replicated); each rank slices its rows.
    """

    def __init__(self, rowptr, col, val, n_nodes, world, rank, gemm, spmm, group=None):
        self.n, self.world, self.rank, self.group = n_nodes, world, rank, group
        self.gemm, self.spmm = gemm, spmm
        self.lo, self.hi = shard_bounds(n_nodes, world, rank)
        lo, hi = self.lo, self.hi
        rp = rowptr[lo:hi + 1]
        base = int(rp[0])
        self.rowptr = (rp - base).contiguous()
        end = int(rp[-1])
        self.col = col[base:end].contiguous()
        self.val = val[base:end].contiguous()

    def layer(self, x_local, weight, bias, relu, renorm=False):
        xw_local = self.gemm(x_local, weight)                                           # rows of X @ W
        xw = all_gather_rows(xw_local, self.n, self.world, self.rank, self.group)       # exchange step
        if renorm:                                                                      # aggregate own rows (+ renorm_)
            return self.spmm(self.rowptr, self.col, self.val, xw, bias, relu, True)
        return self.spmm(self.rowptr, self.col, self.val, xw, bias, relu)

    def encode(self, x_local, w1, b1, w2, b2, renorm=False):
        """renorm=True: the emb.renorm_(2, 0, 1) that Net.decode applies (TLCGNN.py:48) is fused into the last aggregation
        (a row's norm does not depend on the other ranks' rows, so it commutes with the final all-gather)."""
        h = self.layer(x_local, w1, b1, True)
        h = self.layer(h, w2, b2, True, renorm)
        return all_gather_rows(h, self.n, self.world, self.rank, self.group)            # every rank decodes its pair shard

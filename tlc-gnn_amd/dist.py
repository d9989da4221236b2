"""Multi-GPU layout of the path (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI).

The reference is single-process, single-device (no torch.distributed, no DataParallel: SURVEY.md §2.1), so this is
new design, kept to what the path needs:

  * PD/PI per pair and the decode are independent units: the pair list is dealt to the ranks by descending estimated cost
    (or cut into contiguous shards), the (tiny) CSR is replicated, NO data-path collective.
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


def ball_bound(rowptr, col, hop):
    """Host twin of tlc_ball_bound_kernel: per-node upper bound of |ball_hop(x)| (ub_1 = 1 + deg, ub_h(x) = 1 + sum_y ub_{h-1}(y)
    over the neighbours y).  A vicinity is a subset of both endpoints' balls, so min(ub[u], ub[v]) bounds its size: the cost
    estimate the pair list is cut by."""
    rowptr = np.asarray(rowptr, dtype=np.int64)
    col = np.asarray(col, dtype=np.int64)
    deg = np.diff(rowptr)
    ub = 1.0 + deg
    rows = np.repeat(np.arange(len(deg)), deg)
    for _ in range(1, hop):
        ub = 1.0 + np.bincount(rows, weights=ub[col], minlength=len(deg))
    return ub


def pair_cost(ub, pairs):
    """Work estimate of one pair: the smaller of the two ball bounds (the vicinity's size bound) plus a constant for the
    extraction itself; far pairs (most of a negative sweep) cost the constant."""
    pairs = np.asarray(pairs).reshape(-1, 2)
    return 8.0 + np.minimum(ub[pairs[:, 0]], ub[pairs[:, 1]])


def shard_pairs_interleaved(cost, world):
    """Deal the pair list to `world` ranks by DESCENDING cost, boustrophedon (0 1 .. w-1 w-1 .. 1 0 0 1 ..): every rank gets
    its share of the heavy tail instead of whoever owns the stretch of the list where the hub pairs sit -- a contiguous cut of
    504 514 near pairs at equal summed cost gave the ranks 218 090 / 286 424 pairs and very different slowest vicinities.
    Returns `world` ascending index arrays (a partition of range(len(cost))); deterministic, so every rank computes all of them."""
    cost = np.asarray(cost, dtype=np.float64)
    order = np.argsort(-cost, kind="stable")
    pos = np.arange(len(order))
    lap, k = np.divmod(pos, world)
    owner = np.where(lap % 2 == 0, k, world - 1 - k)
    return [np.sort(order[owner == r]) for r in range(world)]


def pd_pi_batch_sharded(run, pairs, world, rank, cost=None, gather=None, scheme="interleaved"):
    """The pair list of get_pimg_for_all_edges (sg2dgm/riccidist2dgm.py:362-370; the reference maps its ThreadPool over exactly
    this list) cut into `world` shards; rank `rank` runs its own shard, no data-path collective.

    scheme "interleaved" (with a cost): shard_pairs_interleaved, the rows come back to list order through the index arrays;
    "contiguous": contiguous shards of equal summed cost (or of equal length without a cost).
    run(pairs_shard) -> (rows [k, 25], status [k]) is the per-GPU path (DeviceGraph.pd_pi_batch in the product).
    Returns (rows, status, part) with part = the (lo, hi) of a contiguous shard or the index array of an interleaved one; with
    gather=callable the shards of all ranks are exchanged by it (gather_shards / gather_shards_indexed, chosen by the
    scheme) and the full arrays come back in list order."""
    n = len(pairs)
    if cost is not None and scheme == "interleaved" and world > 1:
        parts = shard_pairs_interleaved(cost, world)
        mine = parts[rank]
        import torch
        idx = torch.as_tensor(mine, device=pairs.device) if type(pairs).__module__.startswith("torch") else mine
        rows, status = run(pairs[idx])
        if gather is None:
            return rows, status, mine
        g = gather_shards_indexed if gather is gather_shards else gather
        return g(rows, parts), g(status, parts), mine
    bounds = shard_pairs_by_cost(cost, world) if cost is not None else [shard_bounds(n, world, r)[0] for r in range(world)] + [n]
    lo, hi = bounds[rank], bounds[rank + 1]
    rows, status = run(pairs[lo:hi])
    if gather is None:
        return rows, status, (lo, hi)
    return gather(rows, bounds), gather(status, bounds), (lo, hi)


def gather_shards_indexed(local, parts, group=None, always_collective=False):
    """Interleaved shards -> the whole array in list order on every rank: ONE all_gather_into_tensor over blocks padded to the
    largest shard, then every block is scattered to its rows' list positions (parts[r] = the positions of rank r's rows).
    always_collective: a one-rank group goes through the collective too (the RCCL test on the one-GPU box)."""
    import torch
    import torch.distributed as dist
    world = len(parts)
    if world == 1 and not always_collective:
        return local
    sizes = [len(p) for p in parts]
    blk = max(max(sizes), 1)
    tail = tuple(local.shape[1:])
    send = torch.zeros((blk,) + tail, dtype=local.dtype, device=local.device)
    send[:local.shape[0]] = local
    recv = torch.empty((world * blk,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    out = torch.empty((sum(sizes),) + tail, dtype=local.dtype, device=local.device)
    for r in range(world):
        out[torch.as_tensor(parts[r], device=local.device)] = recv[r * blk:r * blk + sizes[r]]
    return out


def gather_shards(local, bounds, group=None):
    """Ragged contiguous shards -> the whole array on every rank: one all_gather_into_tensor over blocks padded to the largest
    shard (RCCL over xGMI with backend nccl), then the blocks are laid end to end."""
    import torch
    import torch.distributed as dist
    world = len(bounds) - 1
    if world == 1:
        return local
    sizes = [bounds[r + 1] - bounds[r] for r in range(world)]
    blk = max(max(sizes), 1)
    tail = tuple(local.shape[1:])
    send = torch.zeros((blk,) + tail, dtype=local.dtype, device=local.device)
    send[:local.shape[0]] = local
    recv = torch.empty((world * blk,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(recv, send, group=group)
    return torch.cat([recv[r * blk:r * blk + sizes[r]] for r in range(world)])


class PaddedRows:
    """Row blocks of shard_bounds laid out for ONE all_gather_into_tensor and read in place afterwards.

    Rank r's rows sit at [r*blk, r*blk + count_r) of a [world*blk, k] buffer (blk = the largest block).  Nothing is copied
    around the collective: the producer writes its rows straight into `send(k)` (a view of persistent storage whose pad rows
    stay zero), the consumers index the gathered buffer through `remap(ids)` (global row id -> padded row id, computed once
    for the CSR columns and the decode pairs)."""

    def __init__(self, n_total, world, rank):
        self.n, self.world, self.rank = n_total, world, rank
        self.blk = (n_total + world - 1) // world if world > 1 else n_total
        self.lo, self.hi = shard_bounds(n_total, world, rank)
        self._send, self._recv = {}, {}

    def remap(self, ids):
        """global row ids (tensor or ndarray of ints) -> row ids of the padded buffer."""
        if self.world == 1:
            return ids
        base, rem = divmod(self.n, self.world)
        # rows [0, rem*(base+1)) belong to the first `rem` ranks (base+1 rows each), the rest to ranks of `base` rows
        big = rem * (base + 1)
        if type(ids).__module__.startswith("torch"):
            import torch
            owner = torch.where(ids < big, ids // (base + 1), rem + (ids - big) // max(base, 1))
            start = torch.where(owner < rem, owner * (base + 1), big + (owner - rem) * base)
        else:
            ids = np.asarray(ids)
            owner = np.where(ids < big, ids // (base + 1), rem + (ids - big) // max(base, 1))
            start = np.where(owner < rem, owner * (base + 1), big + (owner - rem) * base)
        return owner * self.blk + (ids - start)

    def send(self, key, k, like):
        """[hi-lo, k] view to write the local rows into (the pad rows of the block were zeroed once).  `key` names the buffer
        pair: one per exchange step of a forward, so that a block is never rewritten while a later kernel still reads it."""
        import torch
        buf = self._send.get(key)
        if buf is None or buf.shape[1] != k or buf.dtype != like.dtype or buf.device != like.device:
            # (a buffer is tied to the width, dtype and device it was made for: a later call with others gets its own)
            self._send[key] = torch.zeros((self.blk, k), dtype=like.dtype, device=like.device)
            self._recv[key] = torch.empty((self.world * self.blk, k), dtype=like.dtype, device=like.device)
        return self._send[key][:self.hi - self.lo]

    def gather(self, key, group=None):
        """all-gather of the block last written through send(key, ...) -> [world*blk, k] padded buffer (persistent)."""
        import torch.distributed as dist
        dist.all_gather_into_tensor(self._recv[key], self._send[key], group=group)
        return self._recv[key]


class ShardedGCNEncoder:
    """Net.encode (baselines/TLCGNN.py:19-26, eval mode) over `world` ranks.

    mode "allgather": node rows sharded; per layer every rank projects its rows, ONE all-gather of the projected rows, then
      aggregates its rows; a last all-gather hands every rank the whole embedding for its pair shard.  The gathered buffers
      keep the padded block layout (PaddedRows): the CSR columns are remapped once, `row_map` remaps the decode pairs.
    mode "replicated": every rank runs the whole (tiny) encoder, no collective at all (the PubMed layers move
      ~1 MB per link, i.e. the all-gathers are latency, not bandwidth).
    gemm(x_rows, W, out=None) -> rows @ W;  spmm(rowptr, col, val, X, bias, relu[, renorm][, out=]) -> aggregated rows.
    rowptr/col/val: the gcn-normalised CSR by target of the WHOLE graph (replicated); each rank slices its rows.
    """

    def __init__(self, rowptr, col, val, n_nodes, world, rank, gemm, spmm, group=None, mode="allgather"):
        assert mode in ("allgather", "replicated")
        self.n, self.world, self.rank, self.group = n_nodes, world, rank, group
        self.gemm, self.spmm = gemm, spmm
        self.mode = mode if world > 1 else "replicated"
        if self.mode == "replicated":
            self.lo, self.hi = 0, n_nodes
            self.rowptr, self.col, self.val = rowptr, col, val
            self.rows = None
            return
        self.rows = PaddedRows(n_nodes, world, rank)
        self.lo, self.hi = lo, hi = self.rows.lo, self.rows.hi
        rp = rowptr[lo:hi + 1]
        base = int(rp[0])
        self.rowptr = (rp - base).contiguous()
        end = int(rp[-1])
        self.col = self.rows.remap(col[base:end].long()).to(col.dtype).contiguous()      # columns index the padded buffers
        self.val = val[base:end].contiguous()

    def row_map(self, ids):
        """node ids -> rows of the tensor encode() returns (identity unless the padded all-gather layout is in use)."""
        return ids if self.rows is None else self.rows.remap(ids)

    def _aggregate(self, xw, bias, relu, renorm, out=None):
        kw = {} if out is None else {"out": out}
        if renorm:
            return self.spmm(self.rowptr, self.col, self.val, xw, bias, relu, True, **kw)
        return self.spmm(self.rowptr, self.col, self.val, xw, bias, relu, **kw)

    def layer(self, x_local, weight, bias, relu, renorm=False, key=0, out=None):
        if self.rows is None:
            return self._aggregate(self.gemm(x_local, weight), bias, relu, renorm)
        k = weight.shape[1]
        self.gemm(x_local, weight, out=self.rows.send(("xw", key), k, x_local))         # rows of X @ W, written in place
        xw = self.rows.gather(("xw", key), self.group)                                  # exchange step
        return self._aggregate(xw, bias, relu, renorm, out=out)

    def encode(self, x_local, w1, b1, w2, b2, renorm=False):
        """renorm=True: the emb.renorm_(2, 0, 1) that Net.decode applies (TLCGNN.py:48) is fused into the last aggregation
        (a row's norm does not depend on the other ranks' rows, so it commutes with the final all-gather).
        Returns the embedding of ALL nodes; index it through row_map()."""
        h = self.layer(x_local, w1, b1, True, key=1)
        if self.rows is None:
            return self.layer(h, w2, b2, True, renorm)
        # the last aggregation writes straight into the send block of the final exchange
        self.layer(h, w2, b2, True, renorm, key=2, out=self.rows.send("emb", w2.shape[1], x_local))
        return self.rows.gather("emb", self.group)                                      # every rank decodes its pair shard

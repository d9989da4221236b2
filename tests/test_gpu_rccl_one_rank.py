"""GPU: RCCL (torch.distributed backend "nccl") as a ONE-rank group on the one-GPU box, in the order bench.py uses at N > 1: the
handle -- and with it its eleven streams -- first, then the communicator; the repository's own collectives
(dist.gather_shards_indexed, dist.PaddedRows.gather: all_gather_into_tensor) run beside three pipelined image batches, and the
rows that come out equal a stream-ordered call's, bit for bit.  It proves nothing about scaling; it keeps the first real
multi-GPU run from being the first time RCCL's streams meet the handle's."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_rccl_one_rank_group_beside_pipelined_batches():
    import torch
    import torch.distributed as dist
    from tlc_gnn_amd import engine, synth
    from tlc_gnn_amd import dist as tdist
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0") == "0"
    n, edges, kappa, hop, _ = synth.shaped_graph("PubMed", scale=0.5)
    rowptr, col, w = synth.edges_to_csr(n, edges, kappa)
    g = engine.DeviceGraph(rowptr, col, w, device=0)                      # the handle and its streams BEFORE the communicator
    pairs = torch.as_tensor(np.ascontiguousarray(edges[:12000], dtype=np.int32)).cuda()
    E = len(pairs)
    want, want_st = g.pd_pi_batch(pairs, hop)                             # stream-ordered reference rows
    torch.cuda.synchronize()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    assert not dist.is_initialized()
    dist.init_process_group(backend="nccl", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    try:
        assert dist.get_backend() == "nccl" and dist.get_world_size() == 1
        t = torch.ones(1 << 18, device="cuda")
        dist.all_reduce(t)
        dist.barrier()
        outs = [torch.empty((E, 25), dtype=torch.float64, device="cuda") for _ in range(3)]
        sts = [torch.empty(E, dtype=torch.uint8, device="cuda") for _ in range(3)]
        parts = tdist.shard_pairs_interleaved(tdist.ball_bound(rowptr, col, hop)[pairs.cpu().numpy()].min(1), 1)
        pr = tdist.PaddedRows(E, 1, 0)
        for rnd in range(3):
            for k in range(3):                                            # three batches in flight on the handle's workspaces ...
                g.pd_pi_batch(pairs, hop, out=outs[k], status=sts[k], async_=True)
            # ... and the collectives of the sharded path beside them (rows of the reference call: resident, independent buffers)
            rows = tdist.gather_shards_indexed(want, parts, always_collective=True)
            pr.send("rows", 25, want).copy_(want)
            padded = pr.gather("rows")
            g.join()
            torch.cuda.synchronize()
            assert torch.equal(rows, want) and torch.equal(padded[:E], want)
            for k in range(3):
                if not (torch.equal(outs[k], want) and torch.equal(sts[k], want_st)):
                    # (which rows, how far off, which sizes: a flake here in round 5 left no trace of its cause)
                    bad = torch.nonzero((outs[k] != want).any(dim=1) | (sts[k] != want_st)).view(-1)
                    nn, mm = g.sizes(E)
                    idx = bad[:8].cpu().numpy()
                    raise AssertionError("round %d buffer %d: %d rows differ; first %s  n %s  m2 %s  max |diff| %.3e  status %s / %s" % (
                        rnd, k, bad.numel(), idx.tolist(), nn[idx].tolist(), mm[idx].tolist(), float((outs[k] - want).abs().max()),
                        sts[k][bad[:8]].tolist(), want_st[bad[:8]].tolist()))
        assert float(t[0]) == 1.0
    finally:
        dist.destroy_process_group()
        g.close()

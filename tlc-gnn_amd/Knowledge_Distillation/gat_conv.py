"""Drop-in for the reference's Knowledge_Distillation/gat_conv.py ("the PDGNN layer", README.md:70).

  GATConv.__init__ :62-104, forward :113-181, message :183-200, aggregate :202-216.

Tensor input, heads=1, new_node_feat=True, use_edge_attn=True, add_self_loops=True (what Teacher_model.py:182-189 builds)
run as two HIP kernels behind `tlc_gat_layer_fwd` (include/tlcgnn.h); other configurations raise.  With gradients enabled and
an input or parameter that requires them the call goes through `autograd.GatLayer`, whose backward is `tlc_gat_layer_bwd`.
"""
import math

import torch
from torch.nn import Linear, Parameter

from .. import ops, autograd


def glorot(tensor):
    if tensor is not None:
        stdv = math.sqrt(6.0 / (tensor.size(-2) + tensor.size(-1)))
        tensor.data.uniform_(-stdv, stdv)


class GraphBatch:
    """The graph structure of one (block-diagonal) batch, built ONCE and held by the caller: remove_self_loops + add_self_loops
    (:146-152) grouped by target (`GATConv.csr_by_target`).  The reference rebuilds nothing here because its layers gather through
    edge_index directly; the HIP layers read a CSR by target, and building it (a sort of 2 M edges for ogbg-molhiv's 41 127 graphs:
    0.45 ms) on every forward of an evaluate-time loop (train_Teacher_Model.py:124-151) costs a quarter of the forward.

    Reuse rule (the one data_utils_NC._vicinities uses): the object holds the edge_index tensor it was built from, so that
    tensor's storage cannot pass to another tensor, and compares its `_version` (in-place edits) -- `check()` raises on a
    different or edited tensor instead of silently using a stale structure."""

    def __init__(self, edge_index, n, tiled=True):
        self.edge_index, self.n, self.version = edge_index, int(n), edge_index._version
        self.rowptr, self.col = GATConv.csr_by_target(edge_index, n)
        self._tiled, self._tiles, self._cut = bool(tiled and edge_index.is_cuda), None, False

    @property
    def tiles(self):
        """A batch of small graphs cut into self-contained tiles (ops.gat_tiles): its layers then run as one kernel each with the node
        rows in LDS (tlc_gat_layer_tiled_fwd); None for a batch without close enough cuts (one big graph).  Cut on FIRST USE by a
        forward without gradients: the cut is three launches and one blocking host read, and the autograd path (a training step's
        Base_Model.forward builds a GraphBatch per call) never looks at it."""
        if self._tiled and not self._cut:
            self._tiles, self._cut = ops.gat_tiles(self.rowptr, self.col, self.n), True
        return self._tiles

    def edge_ends(self, m):
        """int32 (src, dst) of the first m edges -- the batch without the self loops appended last (train_Teacher_Model.py:43-44,
        Teacher_model.py:54-55): the edge head's operands, converted once per batch instead of once per forward."""
        if getattr(self, "_ends", None) is None or self._ends[0] != m:
            ei = self.edge_index
            self._ends = (m, ei[0, :m].to(torch.int32).contiguous(), ei[1, :m].to(torch.int32).contiguous())
        return self._ends[1], self._ends[2]

    def check(self, edge_index, n):
        if edge_index is not self.edge_index or edge_index._version != self.version or int(n) != self.n:
            raise ValueError("GraphBatch: built for another (or since edited) edge_index / node count; build a new one")
        return self.rowptr, self.col


class GATConv(torch.nn.Module):
    def __init__(self, in_channels, out_channels, double_input=False, new_node_feat=True, use_edge_attn=True, heads=1,
                 concat=True, negative_slope=0.2, dropout=0., add_self_loops=True, bias=True, **kwargs):
        super(GATConv, self).__init__()
        if not isinstance(in_channels, int) or heads != 1 or not new_node_feat or not use_edge_attn or not add_self_loops \
                or not bias or abs(negative_slope - 0.2) > 1e-12:
            raise NotImplementedError("GATConv (HIP): only the PDGNN configuration is implemented: int in_channels, heads=1, "
                                      "new_node_feat, use_edge_attn, add_self_loops, bias, negative_slope=0.2")
        if concat:
            # the reference's concat=True path views [N,1,2C] as [-1, C] and then adds a 2C bias (:166-172): a shape error
            raise NotImplementedError("GATConv (HIP): concat=True is not usable in the reference either; use concat=False")
        self.in_channels, self.out_channels, self.heads, self.concat = in_channels, out_channels, heads, concat
        self.negative_slope, self.dropout, self.add_self_loops = negative_slope, dropout, add_self_loops
        self.new_node_feat, self.use_edge_attn = new_node_feat, use_edge_attn
        self.lin_ij = Linear(2 * out_channels, out_channels, bias=False)                        # :80-81
        self.lin_l = Linear((2 if double_input else 1) * in_channels, heads * out_channels, bias=False)   # :83-86
        self.lin_r = self.lin_l
        self.att_l = Parameter(torch.Tensor(1, heads, out_channels))
        self.att_r = Parameter(torch.Tensor(1, heads, out_channels))                            # unused on the tensor path (:135-136)
        self.bias = Parameter(torch.Tensor(2 * out_channels))                                   # :99-101
        self.reset_parameters()

    def reset_parameters(self):
        glorot(self.lin_l.weight)
        glorot(self.lin_r.weight)          # lin_r IS lin_l (:86): the reference draws the shared weight twice (:107-108); the
        glorot(self.att_l)                 # second draw is kept so that every later parameter sees the same RNG stream
        glorot(self.att_r)
        self.bias.data.zero_()

    @staticmethod
    def csr_by_target(edge_index, n):
        """remove_self_loops + add_self_loops (:146-152) and grouping by target == the structure gcn_norm builds.  Built per
        call, like the reference (no cache: an address-keyed cache is stale as soon as the allocator reuses a block or
        the caller edits edge_index in place); a caller that runs several layers on one graph builds it once and passes
        `csr=` down, as Base_Model.forward does."""
        return ops.csr_by_target(edge_index, n)

    def forward(self, x, edge_index, size=None, return_attention_weights=None, prelu_slope=-1.0, csr=None):
        if not isinstance(x, torch.Tensor) or size is not None or return_attention_weights is not None:
            raise NotImplementedError("GATConv (HIP): tensor input, size=None, return_attention_weights=None only")
        assert x.dim() == 2, 'Static graphs not supported in `GATConv`.'
        if self.training and self.dropout > 0:
            raise NotImplementedError("GATConv (HIP): attention dropout in training mode is not implemented")
        batch = None
        if isinstance(csr, GraphBatch):
            batch = csr
            csr = csr.check(edge_index, x.shape[0])
        rowptr, col = csr if csr is not None else self.csr_by_target(edge_index, x.shape[0])
        if torch.is_grad_enabled() and (x.requires_grad or self.lin_l.weight.requires_grad or self.att_l.requires_grad
                                        or self.lin_ij.weight.requires_grad or self.bias.requires_grad):
            return autograd.gat_layer(x, self.lin_l.weight, self.att_l, self.lin_ij.weight, self.bias, rowptr, col, prelu_slope)
        tiles = batch.tiles if (batch is not None and ops.gat_tiled_ok(x.shape[1], self.out_channels)) else None   # (cut on first use)
        if tiles is not None:
            return ops.gat_layer_tiled(rowptr, col, tiles, x, self.lin_l.weight.detach(), self.att_l.detach().reshape(-1),
                                       self.lin_ij.weight.detach(), self.bias.detach(), prelu_slope=prelu_slope)
        return ops.gat_layer(rowptr, col, x, self.lin_l.weight.detach(), self.att_l.detach().reshape(-1),
                             self.lin_ij.weight.detach(), self.bias.detach(), prelu_slope=prelu_slope)

    def __repr__(self):
        return '{}({}, {}, heads={})'.format(self.__class__.__name__, self.in_channels, self.out_channels, self.heads)

"""Pure-torch fp32 CPU restatement of the model side of the path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED against the reference at this boundary: torch_geometric / torch_scatter / torch_sparse are not
installed in the build container and cannot be installed, so the reference's own GCNConv / scatter /
softmax kernels cannot be run.  This file restates them from the in-tree sources:

  gcn_norm, GCNConv.forward     Knowledge_Distillation/PD_conv.py:35-70,146-148,179-188 (verbatim PyG 1.6.1 fork)
  Net.encode / Net.decode       baselines/TLCGNN.py:19-62
  MessagePassing gather/scatter Knowledge_Distillation/message_passing.py:124-183,231-261,275-293
  GATConv ("PDGNN layer")       Knowledge_Distillation/gat_conv.py:62-216
  Teacher/Base model (GAT)      Knowledge_Distillation/Teacher_model.py:46-59,182-189,213-229
plus the PyG-1.6.1 conventions: softmax = exp(x - segment max) / (segment sum + 1e-16); add_self_loops appends the
loops at the end; scatter min/max leave 0 in empty segments.
"""
import torch
import torch.nn.functional as F


# ---- GCNConv -------------------------------------------------------------------------------------------------------
def add_remaining_self_loops(edge_index, num_nodes):
    """torch_geometric.utils.add_remaining_self_loops with unit weights: existing loops are dropped and exactly one
    loop per node is appended at the end."""
    row, col = edge_index[0], edge_index[1]
    mask = row != col
    loop = torch.arange(num_nodes, dtype=edge_index.dtype)
    return torch.cat([edge_index[:, mask], torch.stack([loop, loop])], dim=1)


def gcn_norm(edge_index, num_nodes):
    """PD_conv.py:35-70, dense edge_index branch, edge_weight=None, improved=False."""
    ei = add_remaining_self_loops(edge_index, num_nodes)
    w = torch.ones(ei.shape[1], dtype=torch.float32)
    row, col = ei[0], ei[1]
    deg = torch.zeros(num_nodes, dtype=torch.float32).index_add_(0, col, w)      # scatter_add(w, col)
    dis = deg.pow(-0.5)
    dis = dis.masked_fill(dis == float("inf"), 0.0)
    return ei, dis[row] * w * dis[col]


def gcn_conv(x, edge_index, weight, bias):
    """PD_conv.py:179-188: x @ W, propagate(add) at the target edge_index[1], + bias."""
    ei, norm = gcn_norm(edge_index, x.shape[0])
    xw = x @ weight
    out = torch.zeros(x.shape[0], weight.shape[1], dtype=torch.float32)
    out.index_add_(0, ei[1], norm[:, None] * xw[ei[0]])
    return out + bias


def tlcgnn_encode(x, edge_index, w1, b1, w2, b2):
    """TLCGNN.py:19-26 in eval mode (dropout is the identity)."""
    h = F.relu(gcn_conv(x, edge_index, w1, b1))
    return F.relu(gcn_conv(h, edge_index, w2, b2))


def tlcgnn_decode(emb, pairs, pi, lin1_w, lin1_b, lin_w, lin_b):
    """TLCGNN.py:48-61.  emb is renormed IN PLACE like the reference; returns prob."""
    emb = emb.renorm_(2, 0, 1)
    new_x = pi.to(torch.float32)                                     # torch.Tensor(PI)
    a, b = emb[pairs[:, 0].long()], emb[pairs[:, 1].long()]
    sq = (a - b).pow(2)
    h = F.leaky_relu(F.linear(torch.cat((sq, new_x), dim=1), lin1_w, lin1_b), 0.2)
    d = torch.abs(F.linear(h, lin_w, lin_b)).reshape(-1)
    d = torch.clamp(d, min=0, max=40)
    return 1.0 / (torch.exp((d - 2.0) / 1.0) + 1.0)


# ---- GATConv (the PDGNN layer) -----------------------------------------------------------------------------------------
def remove_self_loops(edge_index):
    m = edge_index[0] != edge_index[1]
    return edge_index[:, m]


def add_self_loops(edge_index, num_nodes):
    loop = torch.arange(num_nodes, dtype=edge_index.dtype)
    return torch.cat([edge_index, torch.stack([loop, loop])], dim=1)


def segment_softmax(src, index, num_nodes):
    """torch_geometric.utils.softmax (1.6.1): subtract the segment max, exp, divide by (segment sum + 1e-16)."""
    mx = torch.full((num_nodes,) + src.shape[1:], float("-inf"), dtype=src.dtype)
    mx = mx.scatter_reduce(0, index.view(-1, *([1] * (src.dim() - 1))).expand_as(src), src, reduce="amax", include_self=True)
    out = (src - mx[index]).exp()
    den = torch.zeros((num_nodes,) + src.shape[1:], dtype=src.dtype).index_add_(0, index, out)
    return out / (den[index] + 1e-16)


def scatter_min_max(src, index, num_nodes):
    """torch_scatter.scatter(..., reduce='min'|'max'): empty segments stay 0."""
    idx = index.view(-1, 1).expand_as(src)
    mn = torch.zeros(num_nodes, src.shape[1], dtype=src.dtype).scatter_reduce(0, idx, src, reduce="amin", include_self=False)
    mx = torch.zeros(num_nodes, src.shape[1], dtype=src.dtype).scatter_reduce(0, idx, src, reduce="amax", include_self=False)
    return mn, mx


def gat_conv(x, edge_index, lin_l_w, att_l, lin_ij_w, bias, negative_slope=0.2):
    """gat_conv.py:113-216 with heads=1, concat=True, new_node_feat=True, use_edge_attn=True, add_self_loops=True,
    tensor input (x_l = x_r = lin_l(x), alpha_r = alpha_l, :129-136).

    lin_l_w [C, in] (no bias, :79), att_l [C] (:87), lin_ij_w [C, 2C] (no bias, :81), bias [2C] (:97).
    Returns [n, 2C].
    """
    n = x.shape[0]
    C = lin_l_w.shape[0]
    x_l = F.linear(x, lin_l_w)                                   # [n, C]
    alpha = (x_l * att_l.view(1, C)).sum(dim=-1)                 # [n]      (:135)
    ei = add_self_loops(remove_self_loops(edge_index), n)        # :146-152
    src, dst = ei[0], ei[1]                                      # x_j = x[src], x_i = x[dst], aggregated at dst
    a = F.leaky_relu(alpha[src] + alpha[dst], negative_slope)    # :184-185
    a = segment_softmax(a, dst, n)                               # :188
    m = F.leaky_relu(F.linear(torch.cat((x_l[dst], x_l[src]), dim=-1), lin_ij_w), 0.2)    # :193-195
    m = m * a.view(-1, 1)                                        # :198-200
    s = torch.zeros(n, C, dtype=x.dtype).index_add_(0, dst, m)   # scatter sum
    mn, mx = scatter_min_max(m, dst, n)
    out = torch.cat((s, mn + mx), dim=1)                         # :216
    return out + bias                                            # mean over 1 head, + bias (:166-172)


def teacher_forward(f, edge_index, params, masks=None):
    """Teacher_Model.forward with type='GAT', compute_loss=False, grad_PI=False (Teacher_model.py:46-59,213-229).

    f: [n,1] filtration; edge_index: [2, m+n] with the n self loops appended at the end (the caller convention of
    train_Teacher_Model.py:43-44).  Returns (x [n,32], pd_hat [m,2]).
    masks (train mode, dropout > 0): the five F.dropout masks in call order, already scaled by 1 / (1 - p) -- on the input
    (:218), after the prelu of conv1 / conv2 / conv4 (:221,224,227) and between lin5 and lin6 (:58); None = eval mode.
    """
    n = f.shape[0]
    x = f if masks is None else f * masks[0]
    for k, name in enumerate(("conv1", "conv2", "conv4")):
        p = params[name]
        x = F.prelu(gat_conv(x, edge_index, p["lin_l"], p["att_l"], p["lin_ij"], p["bias"]), params["prelu"])
        if masks is not None:
            x = x * masks[1 + k]
    p = params["conv3"]
    x = gat_conv(x, edge_index, p["lin_l"], p["att_l"], p["lin_ij"], p["bias"])
    ei = edge_index[:, :-n]                                       # :54-55  strips the appended self loops
    x_in, x_out = x[ei[0]], x[ei[1]]
    h = F.prelu(F.linear(torch.cat((x_in, x_out), dim=1), params["lin5_w"], params["lin5_b"]), params["prelu"])
    if masks is not None:
        h = h * masks[4]
    pd_hat = F.linear(h, params["lin6_w"], params["lin6_b"])
    return x, pd_hat

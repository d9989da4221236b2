"""Drop-in for the reference's Knowledge_Distillation/Teacher_model.py (PDGNN), type='GAT'.

  Teacher_Model.__init__ :21-44, forward :46-104 (grad_PI=False: :53-59, :83-84), compute_PD_loss :106-137 (kernel='wasserstein'),
  Base_Model.__init__ :146-211 (GAT branch :182-189), Base_Model.forward :213-229.

The whole forward stays on the GPU: four fused GAT layers, the edge head and the persistence image raster
(`tlc_gat_layer_fwd`, `tlc_edge_head_fwd`, `tlc_pi_raster`); the reference copies the predicted diagram to the host and
rasterises it with the Cython CPU code (:84).

Training: with `compute_loss=True, kernel='wasserstein'` the diagram loss is the device matching
(`tlc_w2_partial_matching`; the reference calls POT's ot.emd on the host, wasserstein.py:303) and `loss.backward()` runs
`tlc_edge_head_bwd` and `tlc_gat_layer_bwd` (autograd.py).  Train mode with dropout > 0 (the shipped script trains with dropout = 0,
train_Teacher_Model.py:163): torch's F.dropout at the reference's five points (:58, :218-227), on tensors of the same shapes, so a seed
draws the reference's masks; the edge head then runs unfused (the mask sits between its two GEMMs).  What is NOT here: the 'sliced'
kernel (the reference's own branch cannot run: it ends in `return loss, ind_tmp_test, loss_xy, ...` with those names never bound,
:110-123,137, and hands lists of CUDA tensors to scipy's cityblock) and draw_fig.  grad_PI=True (the signature's default; the
training script passes False, train_Teacher_Model.py:51): the image of the differentiable imager (pimg.py:354-400) with its gradient
-- through the points' weights only, because the reference detaches the coordinates inside the normal-CDF factors (:392,395);
`tlc_pi_raster_wgrad` (autograd.DiagramImage).

Evaluation (train_Teacher_Model.py:85-113: `model(filt, edge_index, PD, p=p, kernel=kernel, pair_diagonal=True, grad_PI=False)`):
pair_diagonal=True -> compute_PD_loss(type='inference') :66,134-136 -> `wasserstein_distance_inference` (wasserstein.py:93-195)
-> `tlc_w2_inference_matching`; loss_yd0 is then the targets left to the diagonal.
"""
import time

import torch
import torch.nn.functional as F
from torch.nn import Linear

from .. import ops, engine, autograd
from .gat_conv import GATConv, GraphBatch


class Base_Model(torch.nn.Module):
    def __init__(self, in_dim=1, hidden_dim=32, dropout=0.2, type='GCN', out_dim=2, new_node_feat=True, use_edge_attn=True):
        super(Base_Model, self).__init__()
        if type != 'GAT':
            raise NotImplementedError("Base_Model (HIP): only type='GAT' (the PDGNN layer) is implemented")
        self.conv1 = GATConv(in_dim, hidden_dim, concat=False, new_node_feat=new_node_feat, use_edge_attn=use_edge_attn)
        self.conv2 = GATConv(hidden_dim, hidden_dim, double_input=True, concat=False, new_node_feat=new_node_feat, use_edge_attn=use_edge_attn)
        self.conv3 = GATConv(hidden_dim, int(out_dim / 2), double_input=True, concat=False, new_node_feat=new_node_feat, use_edge_attn=use_edge_attn)
        self.conv4 = GATConv(hidden_dim, hidden_dim, double_input=True, concat=False, new_node_feat=new_node_feat, use_edge_attn=use_edge_attn)
        self.conv5 = GATConv(hidden_dim, hidden_dim, double_input=True, concat=False, new_node_feat=new_node_feat, use_edge_attn=use_edge_attn)
        self.dropout = dropout
        self.type = type
        self.out_dim = out_dim

    def forward(self, x, edge_index, csr=None):
        """csr: a `GraphBatch` of this edge_index held by the caller (a loop that runs the model on the same batch again, e.g.
        evaluate_time of train_Teacher_Model.py:124-151, builds it once); default: built here, once for the four layers."""
        if x.size()[0] == 0:
            return torch.zeros([0, 2], device=x.device)
        # F.dropout of :218-227 (train mode only): on the input and behind the prelu of conv1 / conv2 / conv4.  torch's own op, at the
        # same points and on tensors of the same shapes as the reference, so the same seed draws the same masks
        drop = (lambda t: F.dropout(t, p=self.dropout, training=True)) if (self.training and self.dropout > 0) else (lambda t: t)
        if csr is None:
            csr = GraphBatch(edge_index, x.shape[0])                   # one CSR (and tile cut) for the four layers of THIS call
        elif isinstance(csr, GraphBatch):
            csr.check(edge_index, x.shape[0])                          # (handed down whole: the layers take its tiles)
        x = self.conv1(drop(x), edge_index, prelu_slope=0.1, csr=csr)  # conv -> F.prelu(0.1) fused (:218-219)
        x = self.conv2(drop(x), edge_index, prelu_slope=0.1, csr=csr)
        x = self.conv4(drop(x), edge_index, prelu_slope=0.1, csr=csr)
        x = self.conv3(drop(x), edge_index, csr=csr)
        return x


class Teacher_Model(torch.nn.Module):
    def __init__(self, hidden_dim=32, out_dim=25, num_models=3, dropout=0.2, type='GIN', max_loop_len=10, new_node_feat=True,
                 use_edge_attn=True):
        super(Teacher_Model, self).__init__()
        self.DIM0_Model = Base_Model(1, hidden_dim, dropout, type, out_dim=hidden_dim, new_node_feat=new_node_feat,
                                     use_edge_attn=use_edge_attn)
        self.lin5 = Linear(2 * hidden_dim, hidden_dim)
        self.lin6 = Linear(hidden_dim, 2)
        self.num_models = num_models
        self.lin1 = Linear(2, hidden_dim)
        self.lin2 = Linear(hidden_dim, out_dim)
        self.lin3 = Linear(hidden_dim, hidden_dim)
        self.lin4 = Linear(hidden_dim, hidden_dim)
        self.dropout = dropout

    def _one_call_params(self):
        d = self.DIM0_Model
        ps = []
        for conv in (d.conv1, d.conv2, d.conv4, d.conv3):
            ps += [conv.lin_l.weight, conv.att_l, conv.lin_ij.weight, conv.bias]
        return [p.detach() for p in ps + [self.lin5.weight, self.lin5.bias, self.lin6.weight, self.lin6.bias]]

    def _one_call_ok(self, x0, csr):
        """tlc_pdgnn_forward serves the reference's configuration (in_dim 1, hidden 32, float32) when no gradient is wanted and no
        dropout is drawn; everything else takes the layer-by-layer path below."""
        if not x0.is_cuda or x0.dim() != 2 or x0.shape[1] != 1 or x0.shape[0] == 0 or self.lin5.out_features != 32 \
                or self.DIM0_Model.conv1.out_channels != 32 or self.DIM0_Model.conv3.out_channels != 16 or self.lin5.weight.dtype != torch.float32 \
                or (self.training and self.dropout > 0) or not (csr is None or isinstance(csr, GraphBatch)):
            return False
        return not (torch.is_grad_enabled() and (x0.requires_grad or any(p.requires_grad for p in self.parameters())))

    def forward(self, x0, edge_index0, PD, kernel='sliced', M=50, p=1, pair_diagonal=False, draw_fig=False, fig_name='',
                compute_loss=True, grad_PI=True, graph_ptr=None, edge_ptr=None, pd_ptr=None, csr=None):
        """Reference signature.  compute_loss=True needs kernel='wasserstein' (p = 1 or 2) and returns
        loss0 (differentiable), loss_xy0, loss_xd0, loss_yd0 like :63-66.  pair_diagonal=False (training, :64): every target
        point is matched, loss_yd0 is 0 (wasserstein.py:330-372, num_models=1); pair_diagonal=True (evaluation, :66): the
        distance in which both diagrams may use the diagonal, loss_yd0 = the targets left to it.

        x0 [n,1] filtration, edge_index0 [2, m+n] with the n self loops LAST (train_Teacher_Model.py:43-44).
        Block-diagonal batches: pass graph_ptr (int64 [B+1] node offsets) and edge_ptr (int64 [B+1] offsets into the
        non-self-loop edges) to get one image per graph [B,25]; otherwise one image [25] for the whole input.  With a loss,
        pd_ptr (int64 [B+1]) gives the rows of PD that belong to each graph; without it every graph must have exactly as
        many target points as edges (PD = Ord0 + Ext1 of the same graph: n - 1 + m - n + 1 = m points, data_utils_GC.py:166).
        csr: a `GraphBatch(edge_index0, n)` the caller built once for this batch (else the CSR by target is built per call).
        """
        if draw_fig:
            raise NotImplementedError("Teacher_Model (HIP): draw_fig=False only")
        if compute_loss and kernel != 'wasserstein':
            raise NotImplementedError("Teacher_Model (HIP): compute_loss needs kernel='wasserstein' (the reference's 'sliced' branch cannot "
                                      "run either: Teacher_model.py:110-123 ends in names that were never bound, :137)")
        t1 = time.time()
        if not compute_loss and self._one_call_ok(x0, csr):
            # inference: the whole forward is one library call (tlc_pdgnn_forward: the same kernels, submitted natively)
            n = x0.shape[0]
            m = edge_index0.shape[1] - n
            offs = torch.tensor([0, m], dtype=torch.int64, device=x0.device) if edge_ptr is None else edge_ptr.to(torch.int64)
            held = csr if isinstance(csr, GraphBatch) else None
            if held is not None:
                held.check(edge_index0, n)
            x, img = ops.pdgnn_forward(x0, edge_index0, self._one_call_params(), offs, res=5, hidden=self.lin5.out_features,
                                       rowptr=None if held is None else held.rowptr, col=None if held is None else held.col,
                                       tiles=None if held is None else held.tiles)
            if grad_PI:
                img = img.to(x.dtype)
            if edge_ptr is None:
                img = img[0]
            t2 = time.time()
            return x, img, None, None, None, None, t2 - t1, 0.0
        x = self.DIM0_Model(x0, edge_index0, csr=csr)
        n = x0.shape[0]
        m = edge_index0.shape[1] - n
        if isinstance(csr, GraphBatch):
            src, dst = csr.edge_ends(m)                               # (int32 copies made once per batch, like its CSR)
        else:
            src = edge_index0[0, :m].to(torch.int32).contiguous()     # strips the appended self loops (:54-55)
            dst = edge_index0[1, :m].to(torch.int32).contiguous()
        if self.training and self.dropout > 0:
            # :56-59 with the dropout mask between the head's two GEMMs: the fused kernel (tlc_edge_head) has no place for it, so this
            # one case runs lin5 -> prelu -> F.dropout -> lin6 as separate device ops (torch's linear = rocBLAS; its autograd)
            h = F.linear(torch.cat((x[src.long()], x[dst.long()]), dim=-1), self.lin5.weight, self.lin5.bias)
            h = F.dropout(F.prelu(h, torch.tensor(0.1, device=h.device, dtype=h.dtype)), p=self.dropout, training=True)
            x = F.linear(h, self.lin6.weight, self.lin6.bias)
        elif torch.is_grad_enabled() and (x.requires_grad or self.lin5.weight.requires_grad or self.lin6.weight.requires_grad):
            x = autograd.edge_head(x, self.lin5.weight, self.lin5.bias, self.lin6.weight, self.lin6.bias, src, dst, 0.1)
        else:
            x = ops.edge_head(src, dst, x, self.lin5.weight.detach(), self.lin5.bias.detach(), 0.1,
                              self.lin6.weight.detach(), self.lin6.bias.detach())                  # :56-59 (no dropout)
        t2 = time.time()
        loss0 = loss_xy0 = loss_xd0 = loss_yd0 = None
        if compute_loss:                                                                           # :61-66, compute_PD_loss :124-136
            xoff = None if edge_ptr is None else edge_ptr.to(torch.int64)
            yoff = xoff if pd_ptr is None else pd_ptr.to(torch.int64)
            if pd_ptr is None and edge_ptr is not None and int(PD.shape[0]) != m:
                # (one target point per edge is what makes edge_ptr double as the target offsets)
                raise ValueError("Teacher_Model: PD has %d points for %d edges; pass pd_ptr (target offsets per graph)" % (int(PD.shape[0]), m))
            if pd_ptr is not None and (edge_ptr is None or pd_ptr.numel() != edge_ptr.numel()):
                raise ValueError("Teacher_Model: pd_ptr needs edge_ptr with the same number of graphs")
            parts = autograd.diagram_loss(x, PD.to(torch.float64), order=p, xoff=xoff, yoff=yoff, infer=bool(pair_diagonal))
            loss0, loss_xy0, loss_xd0 = parts[0].sum().reshape(1), parts[1].sum().reshape(1), parts[2].sum().reshape(1)
            loss_yd0 = parts[3].sum().reshape(1) if pair_diagonal else torch.zeros(1, dtype=loss0.dtype, device=loss0.device)
        x0_out = x
        offs = torch.tensor([0, m], dtype=torch.int64, device=x.device) if edge_ptr is None else edge_ptr.to(torch.int64)
        if grad_PI:
            # :80-81, the differentiable imager (pimg.py:354-400): the same image, float32 like the reference's, with the gradient the
            # reference's graph carries -- through the points' weights only (it detaches the CDF factors)
            img = autograd.diagram_image(x, offs, 5) if (torch.is_grad_enabled() and x.requires_grad) else \
                engine.pi_raster(offs, x.detach().to(torch.float64), 5).to(x.dtype)
        else:
            img = engine.pi_raster(offs, x.detach().to(torch.float64), 5)                          # :84, on the device
        if edge_ptr is None:
            img = img[0]
        t3 = time.time()
        return x0_out, img, loss0, loss_xy0, loss_xd0, loss_yd0, t2 - t1, t3 - t2

"""Drop-in for the forward of the reference's Knowledge_Distillation/Teacher_model.py (PDGNN), type='GAT'.

  Teacher_Model.__init__ :21-44, forward :46-104 (compute_loss=False, grad_PI=False: :53-59, :83-84),
  Base_Model.__init__ :146-211 (GAT branch :182-189), Base_Model.forward :213-229.

The whole forward stays on the GPU: four fused GAT layers, the edge head and the persistence image raster
(`tlc_gat_layer_fwd`, `tlc_edge_head_fwd`, `tlc_pi_raster`); the reference copies the predicted diagram to the host and
rasterises it with the Cython CPU code (:84).  Losses (Wasserstein / sliced) are training-only and out of scope.
"""
import time

import torch
import torch.nn.functional as F
from torch.nn import Linear

from .. import ops, engine
from .gat_conv import GATConv


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

    def forward(self, x, edge_index):
        if x.size()[0] == 0:
            return torch.zeros([0, 2], device=x.device)
        if self.training:
            raise NotImplementedError("Base_Model (HIP): forward/eval only (dropout is the identity)")
        csr = GATConv.csr_by_target(edge_index, x.shape[0])            # one CSR for the four layers of THIS call
        x = self.conv1(x, edge_index, prelu_slope=0.1, csr=csr)       # conv -> F.prelu(0.1) fused (:218-219)
        x = self.conv2(x, edge_index, prelu_slope=0.1, csr=csr)
        x = self.conv4(x, edge_index, prelu_slope=0.1, csr=csr)
        x = self.conv3(x, edge_index, csr=csr)
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

    def forward(self, x0, edge_index0, PD, kernel='sliced', M=50, p=1, pair_diagonal=False, draw_fig=False, fig_name='',
                compute_loss=True, grad_PI=True, graph_ptr=None, edge_ptr=None):
        """Reference signature; compute_loss / grad_PI must be False (forward only).

        x0 [n,1] filtration, edge_index0 [2, m+n] with the n self loops LAST (train_Teacher_Model.py:43-44).
        Block-diagonal batches: pass graph_ptr (int64 [B+1] node offsets) and edge_ptr (int64 [B+1] offsets into the
        non-self-loop edges) to get one image per graph [B,25]; otherwise one image [25] for the whole input.
        """
        if compute_loss or grad_PI or draw_fig:
            raise NotImplementedError("Teacher_Model (HIP): forward only (compute_loss=False, grad_PI=False)")
        t1 = time.time()
        x = self.DIM0_Model(x0, edge_index0)
        n = x0.shape[0]
        m = edge_index0.shape[1] - n
        src = edge_index0[0, :m].to(torch.int32).contiguous()         # strips the appended self loops (:54-55)
        dst = edge_index0[1, :m].to(torch.int32).contiguous()
        x = ops.edge_head(src, dst, x, self.lin5.weight.detach(), self.lin5.bias.detach(), 0.1,
                          self.lin6.weight.detach(), self.lin6.bias.detach())                      # :56-59 (eval: no dropout)
        t2 = time.time()
        x0_out = x
        pts = x.to(torch.float64)
        if edge_ptr is None:
            offs = torch.tensor([0, m], dtype=torch.int64, device=x.device)
            img = engine.pi_raster(offs, pts, 5)[0]                                                # :84, on the device
        else:
            img = engine.pi_raster(edge_ptr.to(torch.int64), pts, 5)
        t3 = time.time()
        return x0_out, img, None, None, None, None, t2 - t1, t3 - t2

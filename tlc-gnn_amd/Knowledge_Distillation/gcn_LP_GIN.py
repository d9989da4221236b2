"""Drop-in for the forward of the reference's Knowledge_Distillation/gcn_LP_GIN.py: the TLCGNN link predictor whose
persistence images come from a frozen PDGNN (Teacher_Model) instead of the exact PD/PI path.

  Net.compute_PI :43-64 (per candidate edge: vicinity + filtration -> Teacher_Model -> PI[25]); encode/decode = TLCGNN's.

The reference loops over the candidate edges one by one (one tiny GAT forward each).  Here all vicinities of a chunk are
extracted by the HIP vicinity kernels, stacked block-diagonally and pushed through ONE batched PDGNN forward; the image
raster runs on the device per vicinity.  Edge convention: each undirected edge once (lower label first) followed by one
self loop per node, i.e. the training script's convention (train_Teacher_Model.py:43-44), so that Teacher_Model's
`edge_index0[:, :-len(x0)]` strips exactly the self loops; the reference's inference caller omits the self loops and thereby
drops the last n real edges -- that latent bug is not reproduced.
"""
import torch

from ..baselines import TLCGNN
from .data_utils_LP import Vicinities


class Net(TLCGNN.Net):
    def __init__(self, data, num_features, num_classes, teacher, g=None, ricci_curv=None, dimension=5):
        super().__init__(data, num_features, num_classes, PI=None, dimension=dimension)
        self.modelGIN = teacher
        self.total_edges = getattr(data, "total_edges", None)
        self._vic = Vicinities(g, ricci_curv) if g is not None else None

    @torch.no_grad()
    def compute_PI(self, data, name, chunk=4096):
        """:43-64, batched.  Fills self.PI (float32 [n_pairs, 25] on the device); rows of vicinities without an edge or
        with a single node stay zero, as in the reference (:55-59)."""
        self.modelGIN.eval()
        hop = 2 if name in ["Cora", "Citeseer", "PubMed"] else 1
        E = len(self.total_edges)
        PI = torch.zeros(E, 25, device="cuda")
        for lo in range(0, E, chunk):
            pairs = self.total_edges[lo:lo + chunk]
            b = self._vic.batch(pairs, hop)
            n_tot = int(b["node_ptr"][-1])
            if n_tot == 0:
                continue
            node_ptr, edge_ptr = b["node_ptr"], b["edge_ptr"]
            e = b["edges"].long() + node_ptr[b["pair_of_edge"]].view(-1, 1)       # block-diagonal node ids
            loops = torch.arange(n_tot, device=e.device)
            edge_index = torch.cat([e.t(), torch.stack([loops, loops])], dim=1)
            x = b["f"].to(torch.float32).view(-1, 1)
            _, img, *_ = self.modelGIN(x, edge_index, None, compute_loss=False, grad_PI=False, graph_ptr=node_ptr, edge_ptr=edge_ptr)
            ok = (node_ptr[1:] - node_ptr[:-1]) > 1
            PI[lo:lo + len(pairs)][ok] = img[ok].to(torch.float32)
        self.PI = PI.double()
        self._pi_dev = None
        return self.PI

"""development: one PDGNN training step (forward with the diagram loss + backward) on HIV-shaped molecules in one block-diagonal batch,
per-kernel via rocprofv3 if run under it.  python tools/time_pdgnn_train.py [n_graphs]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from tlc_gnn_amd.Knowledge_Distillation.Teacher_model import Teacher_Model
n_graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rs = np.random.RandomState(1234)
ns = np.maximum(3, rs.poisson(25, size=n_graphs))
edges, fs, node_offs = [], [], [0]
for n in ns:
    par = np.array([rs.randint(0, k) for k in range(1, n)])
    e = np.stack([par, np.arange(1, n)], 1)
    extra = rs.randint(0, n, size=(int(rs.randint(0, 4)), 2)); extra = extra[extra[:, 0] != extra[:, 1]]
    e = np.unique(np.sort(np.concatenate([e, extra]), 1), axis=0)
    deg = np.bincount(e.ravel(), minlength=n).astype(np.float64)
    fs.append(deg / (deg.max() + 1e-10)); edges.append(e); node_offs.append(node_offs[-1] + n)
f = np.concatenate(fs)
per = [np.concatenate([e, e[:, ::-1]]) + node_offs[k] for k, e in enumerate(edges)]
eptr = np.concatenate([[0], np.cumsum([len(p) for p in per])]).astype(np.int64)
both = np.concatenate(per)
N = node_offs[-1]
loops = np.arange(N)
ei = torch.from_numpy(np.concatenate([both, np.stack([loops, loops], 1)]).T.copy()).cuda()
x = torch.from_numpy(f.astype(np.float32)).view(-1, 1).cuda()
b = rs.rand(len(both)); PD = torch.tensor(np.stack([b, b + rs.uniform(0, 0.5, size=len(both))], 1), dtype=torch.float32).cuda()
gptr = torch.tensor(node_offs, dtype=torch.int64).cuda(); d_eptr = torch.from_numpy(eptr).cuda()
torch.manual_seed(1234)
model = Teacher_Model(type='GAT', dropout=0.0).cuda().train()
def step(loss):
    model.zero_grad(set_to_none=True)
    out = model(x, ei, PD, kernel='wasserstein', p=2, grad_PI=False, compute_loss=loss, graph_ptr=gptr, edge_ptr=d_eptr)
    if loss:
        out[2].backward()
def med(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3
fwd = med(lambda: step(False))
full = med(lambda: step(True))
print("%d graphs, %d nodes, %d directed edges: forward %.2f ms; forward + diagram loss + backward %.2f ms (%.2f M graphs/s)" % (n_graphs, N, len(both), fwd, full, n_graphs / full / 1e3))

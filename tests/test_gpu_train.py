"""GPU: SURVEY.md 8(f) item 4 -- the training side of PDGNN: the partial-matching Wasserstein loss on the device
(tlc_w2_partial_matching) against scipy's assignment solver (oracle/w2_ref.py), its gradient against torch.autograd of the
restated loss expression.  PARITY UNPINNED at this boundary: the reference solves the transport with POT's ot.emd
(third-party, absent); the optimal cost is unique and is what is compared."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _problems(rs, B, nmax, style):
    xs, ys = [], []
    for b in range(B):
        n = int(rs.randint(0, nmax + 1))
        m = int(rs.randint(0, n + 1)) if style != "equal" else n
        if style == "grid":                                  # coordinates on a coarse grid: many tied costs
            X = rs.randint(0, 6, size=(n, 2)) / 5.0
            Y = rs.randint(0, 6, size=(m, 2)) / 5.0
        else:
            b0 = rs.rand(n)
            X = np.stack([b0, b0 + rs.uniform(-0.1, 0.6, size=n)], 1)       # a few predicted points BELOW the diagonal
            b1 = rs.rand(m)
            Y = np.stack([b1, b1 + rs.uniform(0.0, 0.7, size=m)], 1)
        xs.append(X.astype(np.float64)); ys.append(Y.astype(np.float64))
    return xs, ys


@pytest.mark.parametrize("order", [2, 1])
@pytest.mark.parametrize("style,nmax", [("random", 40), ("equal", 30), ("grid", 24), ("random", 130), ("random", 300)])
def test_partial_matching_loss_cost_and_gradient(order, style, nmax):
    import torch
    from tlc_gnn_amd import ops
    from oracle import w2_ref
    rs = np.random.RandomState(nmax + order)
    B = 60 if nmax <= 40 else (12 if nmax <= 130 else 4)
    xs, ys = _problems(rs, B, nmax, style)
    xoff = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.int64)
    yoff = np.concatenate([[0], np.cumsum([len(y) for y in ys])]).astype(np.int64)
    X = torch.as_tensor(np.concatenate(xs) if xoff[-1] else np.zeros((0, 2))).cuda()
    Y = torch.as_tensor(np.concatenate(ys) if yoff[-1] else np.zeros((0, 2))).cuda()
    r = ops.w2_partial_matching(torch.as_tensor(xoff).cuda(), X, torch.as_tensor(yoff).cuda(), Y, order=order)
    loss, wxy, wxd = r["loss"].cpu().numpy(), r["wxy"].cpu().numpy(), r["wxd"].cpu().numpy()
    assign, grad, status = r["assign"].cpu().numpy(), r["grad"].cpu().numpy(), r["status"].cpu().numpy()
    assert (status == 0).all()
    for b in range(B):
        Xb, Yb = xs[b], ys[b]
        a = assign[xoff[b]:xoff[b + 1]]
        n, m = len(Xb), len(Yb)
        # a valid partial matching: every target exactly once, the rest on the diagonal
        assert sorted(a[a >= 0].tolist()) == list(range(m)) and (a >= -1).all() and (a < max(m, 1)).all()
        ref_loss, ref_wxy, ref_wxd, ref_assign, ref_cost = w2_ref.partial_matching(Xb, Yb, order)
        # optimal: the cost of the device's assignment equals scipy's optimum
        M = w2_ref.cost_matrix(Xb, Yb, order)
        cost = float(sum(M[i, a[i] if a[i] >= 0 else m] for i in range(n)))
        assert abs(cost - ref_cost) <= 1e-9 * max(1.0, abs(ref_cost)), (b, cost, ref_cost)
        # the loss and its pieces are those of the restated expression on the device's assignment
        l2, wxy2, wxd2 = w2_ref.loss_from_assignment(Xb, Yb, a, order)
        assert abs(loss[b] - l2) <= 1e-12 * max(1.0, l2) and abs(wxy[b] - wxy2) <= 1e-12 * max(1.0, wxy2) and abs(wxd[b] - wxd2) <= 1e-12 * max(1.0, wxd2)
        if order == 2 and style != "grid":
            assert abs(loss[b] - ref_loss) <= 1e-9 * max(1.0, ref_loss)        # = sqrt(optimal cost): unique
        # gradient: torch.autograd of the same expression
        if n:
            Xt = torch.tensor(Xb, requires_grad=True)
            w2_ref.loss_torch(Xt, torch.tensor(Yb), a, order).backward()
            g = grad[xoff[b]:xoff[b + 1]]
            if style != "grid":                                # (on the grid |dx| == |dy| ties: autograd splits, the kernel picks one)
                assert np.abs(g - Xt.grad.numpy()).max() <= 1e-12, b


def test_partial_matching_status_codes_and_empty_diagrams():
    import torch
    from tlc_gnn_amd import ops
    rs = np.random.RandomState(1)
    X = [rs.rand(3, 2), np.zeros((0, 2)), rs.rand(5, 2), rs.rand(4100, 2), np.zeros((0, 2))]
    Y = [rs.rand(5, 2), np.zeros((0, 2)), np.zeros((0, 2)), rs.rand(2, 2), rs.rand(2, 2)]
    xoff = np.concatenate([[0], np.cumsum([len(x) for x in X])]).astype(np.int64)
    yoff = np.concatenate([[0], np.cumsum([len(y) for y in Y])]).astype(np.int64)
    r = ops.w2_partial_matching(torch.as_tensor(xoff).cuda(), torch.as_tensor(np.concatenate(X)).cuda(), torch.as_tensor(yoff).cuda(),
                                torch.as_tensor(np.concatenate(Y)).cuda(), order=2)
    st = r["status"].cpu().numpy().tolist()
    assert st == [1, 0, 0, 2, 1]          # fewer predicted than target; both empty; no targets; too many points; none predicted
    loss = r["loss"].cpu().numpy()
    assert loss[1] == 0.0
    d = (X[2][:, 1] - X[2][:, 0]) * 0.5
    assert abs(loss[2] - np.sqrt((d ** 2).sum())) < 1e-14 and (r["assign"].cpu().numpy()[3:8] == -1).all()
    assert loss[3] == 0.0 and (r["assign"].cpu().numpy()[8:8 + 4100] == -1).all()          # more than 4 096 predicted points: refused


@pytest.mark.parametrize("order", [2, 1])
def test_partial_matching_above_512_points_takes_a_workgroup_per_problem(order):
    """513 .. 4 096 predicted points: the second launch (512 threads per problem); small problems of the same batch still take the
    one-wavefront kernel.  Optimal cost against scipy, loss pieces and gradient against the restated expression."""
    import torch
    from tlc_gnn_amd import ops
    from oracle import w2_ref
    rs = np.random.RandomState(90 + order)
    sizes = [(513, 513), (40, 11), (700, 350), (1500, 1499), (0, 0), (2300, 2300), (512, 100), (1024, 0)]
    xs, ys = [], []
    for n, m in sizes:
        b0 = rs.rand(n); xs.append(np.stack([b0, b0 + rs.uniform(-0.1, 0.6, size=n)], 1))
        b1 = rs.rand(m); ys.append(np.stack([b1, b1 + rs.uniform(0.0, 0.7, size=m)], 1))
    xoff = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.int64)
    yoff = np.concatenate([[0], np.cumsum([len(y) for y in ys])]).astype(np.int64)
    r = ops.w2_partial_matching(torch.as_tensor(xoff).cuda(), torch.as_tensor(np.concatenate(xs)).cuda(), torch.as_tensor(yoff).cuda(),
                                torch.as_tensor(np.concatenate(ys)).cuda(), order=order)
    assert (r["status"].cpu().numpy() == 0).all()
    loss, assign, grad = r["loss"].cpu().numpy(), r["assign"].cpu().numpy(), r["grad"].cpu().numpy()
    for b, (n, m) in enumerate(sizes):
        a = assign[xoff[b]:xoff[b + 1]]
        assert sorted(a[a >= 0].tolist()) == list(range(m)) and (a >= -1).all()
        if n == 0:
            assert loss[b] == 0.0
            continue
        ref_cost = w2_ref.partial_matching(xs[b], ys[b], order)[4]
        M = w2_ref.cost_matrix(xs[b], ys[b], order)
        cost = float(M[np.arange(n), np.where(a >= 0, a, m)].sum())
        assert abs(cost - ref_cost) <= 1e-9 * max(1.0, abs(ref_cost)), (b, n, m, cost, ref_cost)
        l2 = w2_ref.loss_from_assignment(xs[b], ys[b], a, order)[0]
        assert abs(loss[b] - l2) <= 1e-12 * max(1.0, l2)
        Xt = torch.tensor(xs[b], requires_grad=True)
        w2_ref.loss_torch(Xt, torch.tensor(ys[b]), a, order).backward()
        assert np.abs(grad[xoff[b]:xoff[b + 1]] - Xt.grad.numpy()).max() <= 1e-12, b


# ---- backward of the PDGNN layer, the edge head and the whole training step -------------------------------------------------
# against torch.autograd of the CPU restatement (oracle/lp_forward_ref.py), fp32, tolerance 1e-4 of the gradient's largest entry
# (float atomics on the device, a different summation order on the CPU).
def _rel(a, b):
    a = a.detach().cpu().double().reshape(-1); b = b.detach().cpu().double().reshape(-1)
    return float((a - b).abs().max() / max(1e-12, float(b.abs().max())))


def _random_graph(rs, n, m, torch):
    s = rs.randint(0, n, size=m); t = rs.randint(0, n, size=m)
    keep = s != t
    e = np.unique(np.stack([s[keep], t[keep]]), axis=1)          # no duplicate edges (add_self_loops convention of the callers)
    loops = np.arange(n)
    return torch.tensor(np.concatenate([e, np.stack([loops, loops])], axis=1), dtype=torch.int64)


@pytest.mark.parametrize("c_in,c_out,slope", [(1, 32, 0.1), (64, 32, 0.1), (64, 16, -1.0), (5, 8, 0.1), (64, 64, -1.0)])
@pytest.mark.parametrize("n,m", [(300, 2400), (7, 12), (1, 0)])
def test_gat_layer_backward_matches_autograd_of_the_restatement(c_in, c_out, slope, n, m):
    import torch
    import torch.nn.functional as F
    from oracle import lp_forward_ref as ref
    from tlc_gnn_amd import ops
    from tlc_gnn_amd.Knowledge_Distillation.gat_conv import GATConv
    rs = np.random.RandomState(c_in * 100 + c_out + n)
    torch.manual_seed(c_in + c_out + n)
    ei = _random_graph(rs, n, m, torch)
    x = torch.randn(n, c_in)
    wl = (torch.randn(c_out, c_in) * (0.5 / np.sqrt(c_in))).requires_grad_()
    att = (torch.randn(c_out) * 0.5).requires_grad_()
    wij = (torch.randn(c_out, 2 * c_out) * (0.7 / np.sqrt(c_out))).requires_grad_()
    bias = (torch.randn(2 * c_out) * 0.2).requires_grad_()
    xr = x.clone().requires_grad_()
    out_ref = ref.gat_conv(xr, ei, wl, att, wij, bias)
    if slope >= 0:
        out_ref = F.prelu(out_ref, torch.tensor(slope))
    G = torch.randn(n, 2 * c_out)
    (out_ref * G).sum().backward()
    rowptr, col = GATConv.csr_by_target(ei.cuda(), n)
    out = ops.gat_layer(rowptr, col, x.cuda(), wl.detach().cuda(), att.detach().cuda(), wij.detach().cuda(), bias.detach().cuda(), prelu_slope=slope)
    assert _rel(out, out_ref) <= 2e-5
    gx, gwl, gatt, gwij, gbias = ops.gat_layer_bwd(rowptr, col, x.cuda(), wl.detach().cuda(), att.detach().cuda(), wij.detach().cuda(),
                                                   slope, out, G.cuda())
    for name, got, want in (("x", gx, xr.grad), ("lin_l", gwl, wl.grad), ("att", gatt, att.grad), ("lin_ij", gwij, wij.grad),
                            ("bias", gbias, bias.grad)):
        assert got.shape == want.shape, name
        assert _rel(got, want) <= 1e-4, (name, _rel(got, want))


@pytest.mark.parametrize("n,m,c,hidden", [(200, 1500, 32, 32), (9, 20, 32, 32), (50, 300, 16, 16), (5, 0, 32, 32)])
def test_edge_head_backward_matches_autograd(n, m, c, hidden):
    import torch
    import torch.nn.functional as F
    from tlc_gnn_amd import ops
    rs = np.random.RandomState(n + m)
    torch.manual_seed(n + m)
    src = torch.tensor(rs.randint(0, n, size=m), dtype=torch.int64); dst = torch.tensor(rs.randint(0, n, size=m), dtype=torch.int64)
    x = torch.randn(n, c, requires_grad=True)
    w5 = (torch.randn(hidden, 2 * c) * 0.2).requires_grad_(); b5 = (torch.randn(hidden) * 0.2).requires_grad_()
    w6 = (torch.randn(2, hidden) * 0.3).requires_grad_(); b6 = (torch.randn(2) * 0.1).requires_grad_()
    h = F.prelu(F.linear(torch.cat((x[src], x[dst]), dim=1), w5, b5), torch.tensor(0.1))      # Teacher_model.py:54-59
    pd = F.linear(h, w6, b6)
    G = torch.randn(m, 2)
    (pd * G).sum().backward()
    d = lambda t: t.detach().cuda()
    got = ops.edge_head_bwd(src.to(torch.int32).cuda(), dst.to(torch.int32).cuda(), d(x), d(w5), d(b5), 0.1, d(w6), G.cuda())
    for name, g, want in zip(("x", "lin5.weight", "lin5.bias", "lin6.weight", "lin6.bias"), got, (x.grad, w5.grad, b5.grad, w6.grad, b6.grad)):
        if m == 0:
            assert float(g.abs().max()) == 0.0, name
        else:
            assert _rel(g, want) <= 1e-4, (name, _rel(g, want))


def _teacher_params(model, torch):
    params = {"prelu": torch.tensor(0.1)}
    leaves = {}
    for name in ("conv1", "conv2", "conv3", "conv4"):
        c = getattr(model.DIM0_Model, name)
        params[name] = {"lin_l": c.lin_l.weight.detach().clone().requires_grad_(), "att_l": c.att_l.detach().reshape(-1).clone().requires_grad_(),
                        "lin_ij": c.lin_ij.weight.detach().clone().requires_grad_(), "bias": c.bias.detach().clone().requires_grad_()}
        leaves["DIM0_Model.%s.lin_l.weight" % name] = params[name]["lin_l"]
        leaves["DIM0_Model.%s.att_l" % name] = params[name]["att_l"]
        leaves["DIM0_Model.%s.lin_ij.weight" % name] = params[name]["lin_ij"]
        leaves["DIM0_Model.%s.bias" % name] = params[name]["bias"]
    for k, mod, attr in (("lin5_w", "lin5", "weight"), ("lin5_b", "lin5", "bias"), ("lin6_w", "lin6", "weight"), ("lin6_b", "lin6", "bias")):
        params[k] = getattr(getattr(model, mod), attr).detach().clone().requires_grad_()
        leaves["%s.%s" % (mod, attr)] = params[k]
    return params, leaves


@pytest.mark.parametrize("order", [2, 1])
def test_teacher_training_step_gradients_match_autograd_of_the_restatement(order):
    """model(filt, edge_index, PD, p, kernel='wasserstein', grad_PI=False); loss_0.backward() (train_Teacher_Model.py:51-62):
    every parameter's .grad against autograd through oracle/lp_forward_ref.teacher_forward and the restated loss expression
    evaluated on the matching the device found (the matching is a constant of the differentiation, wasserstein.py:303-372)."""
    import torch
    from oracle import lp_forward_ref as ref
    from oracle import w2_ref
    from tlc_gnn_amd.Knowledge_Distillation.Teacher_model import Teacher_Model
    torch.manual_seed(3)
    rs = np.random.RandomState(3)
    n, m = 60, 170
    ei = _random_graph(rs, n, m, torch)
    m = ei.shape[1] - n
    f = torch.rand(n, 1)
    b = rs.rand(m)
    PD = torch.tensor(np.stack([b, b + rs.uniform(0, 0.6, size=m)], 1), dtype=torch.float32)
    model = Teacher_Model(type='GAT', dropout=0.0)
    with torch.no_grad():
        for conv in (model.DIM0_Model.conv1, model.DIM0_Model.conv2, model.DIM0_Model.conv3, model.DIM0_Model.conv4):
            conv.bias.uniform_(-0.2, 0.2)
            torch.nn.init.xavier_uniform_(conv.lin_ij.weight)
    params, leaves = _teacher_params(model, torch)
    model = model.cuda().train()
    x0, img, loss0, lxy, lxd, lyd, _, _ = model(f.cuda(), ei.cuda(), PD.cuda(), kernel='wasserstein', p=order, grad_PI=False)
    assert loss0.shape == (1,) and loss0.requires_grad and img.shape == (25,) and not img.requires_grad
    loss0.backward()
    # the restatement: same forward, the loss expression on the device's matching
    _, pd_ref = ref.teacher_forward(f, ei, params)
    assert _rel(x0, pd_ref) <= 2e-5
    from tlc_gnn_amd import ops
    r = ops.w2_partial_matching(torch.tensor([0, m]).cuda(), x0.detach().double(), torch.tensor([0, m]).cuda(), PD.double().cuda(), order=order)
    assign = r["assign"].cpu().numpy()
    loss_ref = w2_ref.loss_torch(pd_ref, PD, assign, order)
    want_cost = w2_ref.partial_matching(pd_ref.detach().double().numpy(), PD.double().numpy(), order)[0]
    assert abs(float(loss0.detach()) - float(want_cost)) <= 1e-5 * max(1.0, abs(float(want_cost)))          # the optimum (scipy), not just this matching
    assert abs(float(loss0.detach()) - float(loss_ref.detach())) <= 1e-5 * max(1.0, abs(float(loss_ref.detach())))
    loss_ref.backward()
    named = dict(model.named_parameters())
    checked = 0
    for name, leaf in leaves.items():
        got = named[name].grad
        assert got is not None, name
        assert _rel(got.reshape(leaf.shape), leaf.grad) <= 1e-4, (name, _rel(got.reshape(leaf.shape), leaf.grad))
        checked += 1
    assert checked == 20
    for name in ("lin1.weight", "lin2.weight", "DIM0_Model.conv5.bias"):                             # not on the path (:86-101 commented out)
        assert named[name].grad is None


def test_teacher_train_mode_refuses_what_is_not_implemented():
    import torch
    from tlc_gnn_amd.Knowledge_Distillation.Teacher_model import Teacher_Model
    ei = torch.tensor([[0, 1, 0, 1], [1, 0, 0, 1]]).cuda()
    f = torch.rand(2, 1).cuda()
    PD = torch.tensor([[0.1, 0.5], [0.2, 0.3]]).cuda()
    m = Teacher_Model(type='GAT', dropout=0.0).cuda().train()
    with pytest.raises(NotImplementedError):
        m(f, ei, PD, kernel='sliced', grad_PI=False)
    with pytest.raises(NotImplementedError):
        m(f, ei, PD, kernel='wasserstein', draw_fig=True)
    out = m(f, ei, PD, kernel='wasserstein', p=2, grad_PI=False)
    assert out[2].requires_grad and not out[1].requires_grad
    out = m(f, ei, PD, kernel='wasserstein', p=2)                       # grad_PI=True, the signature's default
    assert out[1].requires_grad and out[1].dtype == torch.float32 and out[1].shape == (25,)


def test_differentiable_imager_gradient_goes_through_the_weights_only():
    """Teacher_Model.forward(grad_PI=True) -> pimg.PersistenceImager.transform (pimg.py:354-400): the image of the nograd imager,
    and a gradient that reaches a point through linear_ramp(death - birth) alone -- the reference detaches the coordinates inside
    the normal CDFs (:392,395).  autograd.diagram_image against that expression written in torch (float64), single diagram and a
    batch; points with pers < 0 and pers > 1 get no gradient."""
    import torch
    from tlc_gnn_amd import autograd, engine

    def ncdf(x):
        return 0.5 * torch.erfc(-x / np.sqrt(2.0))

    def restated(pd):
        b, pers = pd[:, 0], pd[:, 1] - pd[:, 0]
        inside = (pers >= 0) & (pers <= 1)
        w = torch.where(inside, pers, (pers > 1).to(pd.dtype).detach())
        grid = torch.arange(6, dtype=pd.dtype) * 0.2
        cb = ncdf(grid[None, :] - b.detach()[:, None])
        cp = ncdf(grid[None, :] - pers.detach()[:, None])
        return torch.einsum('k,ki,kj->ij', w, cb[:, 1:] - cb[:, :-1], cp[:, 1:] - cp[:, :-1]).reshape(-1)

    rs = np.random.RandomState(2)
    counts = [37, 0, 5, 120]
    pts = rs.uniform(0, 1, size=(sum(counts), 2))
    pts[:, 1] = pts[:, 0] + rs.uniform(-0.2, 1.3, size=len(pts))          # some below the diagonal, some with persistence above 1
    pts[3] = [0.25, 0.25]; pts[4] = [0.0, 1.0]                             # persistence exactly 0 and exactly 1: the linear branch
    offs = np.concatenate([[0], np.cumsum(counts)])
    coef = rs.uniform(-1, 1, size=(len(counts), 25))
    x = torch.tensor(pts, dtype=torch.float64, device="cuda", requires_grad=True)
    img = autograd.diagram_image(x, torch.tensor(offs, dtype=torch.int64, device="cuda"), 5)
    assert img.shape == (4, 25) and bool((img[1] == 0).all())
    assert torch.equal(img.detach(), engine.pi_raster(torch.tensor(offs, dtype=torch.int64, device="cuda"), x.detach(), 5))
    (img * torch.tensor(coef, device="cuda")).sum().backward()
    xr = torch.tensor(pts, dtype=torch.float64, requires_grad=True)
    total = 0
    for k in range(len(counts)):
        seg = xr[offs[k]:offs[k + 1]]
        im = restated(seg) if counts[k] else torch.zeros(25, dtype=torch.float64)
        assert (im.detach() - img[k].detach().cpu()).abs().max() <= 1e-12 * max(1.0, float(im.detach().abs().max()))
        total = total + (im * torch.tensor(coef[k])).sum()
    total.backward()
    assert (x.grad.cpu() - xr.grad).abs().max() <= 1e-12 * float(xr.grad.abs().max())
    pers = pts[:, 1] - pts[:, 0]
    out = (pers < 0) | (pers > 1)
    assert out.sum() > 10 and bool((x.grad.cpu()[torch.tensor(out)] == 0).all())
    assert bool((x.grad[:, 0] == -x.grad[:, 1]).all()) and float(x.grad[3].abs().sum()) > 0 and float(x.grad[4].abs().sum()) > 0
    # float32 points (the model's): the image and the gradient in float32
    x32 = torch.tensor(pts[:37], dtype=torch.float32, device="cuda", requires_grad=True)
    im32 = autograd.diagram_image(x32)
    assert im32.dtype == torch.float32 and im32.shape == (1, 25)
    im32.sum().backward()
    assert x32.grad.dtype == torch.float32 and bool(torch.isfinite(x32.grad).all())


def test_teacher_train_mode_dropout_draws_the_masks_of_the_same_seed():
    """Train mode with dropout 0.2 (Teacher_model.py:58, :218-227): F.dropout at the reference's five points on tensors of the
    same shapes.  The forward and every gradient against the restatement with the masks the same seed draws (F.dropout on ones
    of those shapes, in call order, from the same generator state)."""
    import torch
    import torch.nn.functional as F
    from oracle import lp_forward_ref as ref
    from oracle import w2_ref
    from tlc_gnn_amd import ops
    from tlc_gnn_amd.Knowledge_Distillation.Teacher_model import Teacher_Model
    torch.manual_seed(5)
    rs = np.random.RandomState(5)
    n, m = 50, 140
    ei = _random_graph(rs, n, m, torch)
    m = ei.shape[1] - n
    f = torch.rand(n, 1)
    b = rs.rand(m)
    PD = torch.tensor(np.stack([b, b + rs.uniform(0, 0.6, size=m)], 1), dtype=torch.float32)
    model = Teacher_Model(type='GAT', dropout=0.2)
    params, leaves = _teacher_params(model, torch)
    model = model.cuda().train()
    torch.cuda.manual_seed(1234)
    x0, img, loss0, _, _, _, _, _ = model(f.cuda(), ei.cuda(), PD.cuda(), kernel='wasserstein', p=2, grad_PI=False)
    loss0.backward()
    torch.cuda.manual_seed(1234)
    masks = [F.dropout(torch.ones(shape, device="cuda"), p=0.2, training=True).cpu() for shape in ((n, 1), (n, 64), (n, 64), (n, 64), (m, 32))]
    assert all(0.05 < float((mk == 0).float().mean()) < 0.5 for mk in masks[1:])
    _, pd_ref = ref.teacher_forward(f, ei, params, masks=masks)
    assert _rel(x0, pd_ref) <= 2e-5
    _, pd_eval = ref.teacher_forward(f, ei, params)
    assert _rel(x0, pd_eval) > 1e-2                                     # (the masks did something)
    r = ops.w2_partial_matching(torch.tensor([0, m]).cuda(), x0.detach().double(), torch.tensor([0, m]).cuda(), PD.double().cuda(), order=2)
    loss_ref = w2_ref.loss_torch(pd_ref, PD, r["assign"].cpu().numpy(), 2)
    assert abs(float(loss0.detach()) - float(loss_ref.detach())) <= 1e-5 * max(1.0, abs(float(loss_ref.detach())))
    loss_ref.backward()
    named = dict(model.named_parameters())
    for name, leaf in leaves.items():
        got = named[name].grad
        assert got is not None, name
        # (fp32 sums in another order, scaled by 1 / (1 - p) five times over: 5e-4 of the largest entry; a wrong mask is off by 1e-1)
        assert _rel(got.reshape(leaf.shape), leaf.grad) <= 5e-4, (name, _rel(got.reshape(leaf.shape), leaf.grad))
    # eval mode: no mask, the fused head
    model.eval()
    with torch.no_grad():
        xe = model(f.cuda(), ei.cuda(), PD.cuda(), compute_loss=False, grad_PI=False)[0]
    assert _rel(xe, pd_eval) <= 2e-5


def test_teacher_training_step_on_a_block_diagonal_batch_equals_the_sum_over_its_graphs():
    """The reference steps graph by graph and lets the gradients of `batch_size` samples accumulate before `optimizer.step()`
    (train_Teacher_Model.py:35-64).  The drop-in takes the same graphs as ONE block-diagonal batch (graph_ptr / edge_ptr): its loss
    is the sum of the per-graph losses and its gradients are the accumulated ones -- checked against autograd of the restatement run
    graph by graph."""
    import torch
    from oracle import lp_forward_ref as ref
    from oracle import w2_ref
    from tlc_gnn_amd import ops
    from tlc_gnn_amd.Knowledge_Distillation.Teacher_model import Teacher_Model
    torch.manual_seed(11)
    rs = np.random.RandomState(11)
    graphs, fs, pds = [], [], []
    for b in range(6):
        n = int(rs.randint(5, 40))
        ei = _random_graph(rs, n, int(rs.randint(n, 4 * n)), torch)
        m = ei.shape[1] - n
        graphs.append((n, ei[:, :m]))
        fs.append(torch.rand(n, 1))
        bb = rs.rand(m)
        pds.append(torch.tensor(np.stack([bb, bb + rs.uniform(0, 0.6, size=m)], 1), dtype=torch.float32))
    gptr = np.concatenate([[0], np.cumsum([g[0] for g in graphs])])
    eptr = np.concatenate([[0], np.cumsum([g[1].shape[1] for g in graphs])])
    N = int(gptr[-1])
    ei_all = torch.cat([g[1] + int(gptr[b]) for b, g in enumerate(graphs)], dim=1)
    loops = torch.arange(N, dtype=torch.int64)
    ei_full = torch.cat([ei_all, torch.stack([loops, loops])], dim=1)                  # self loops LAST (train_Teacher_Model.py:43-44)
    model = Teacher_Model(type='GAT', dropout=0.0)
    with torch.no_grad():
        for conv in (model.DIM0_Model.conv1, model.DIM0_Model.conv2, model.DIM0_Model.conv3, model.DIM0_Model.conv4):
            conv.bias.uniform_(-0.2, 0.2)
            torch.nn.init.xavier_uniform_(conv.lin_ij.weight)
    params, leaves = _teacher_params(model, torch)
    model = model.cuda().train()
    x0, img, loss0, lxy, lxd, lyd, _, _ = model(torch.cat(fs).cuda(), ei_full.cuda(), torch.cat(pds).cuda(), kernel='wasserstein', p=2, grad_PI=False,
                                                graph_ptr=torch.tensor(gptr).cuda(), edge_ptr=torch.tensor(eptr).cuda())
    assert img.shape == (6, 25) and loss0.shape == (1,)
    loss0.backward()
    # graph by graph through the restatement, the loss expression on the device's matching of that graph
    xoff = torch.tensor(eptr).cuda()
    r = ops.w2_partial_matching(xoff, x0.detach().double(), xoff, torch.cat(pds).double().cuda(), order=2)
    assign = r["assign"].cpu().numpy()
    total = 0.0
    for b, (n, e) in enumerate(graphs):
        m = e.shape[1]
        lp = torch.arange(n, dtype=torch.int64)
        _, pd_ref = ref.teacher_forward(fs[b], torch.cat([e, torch.stack([lp, lp])], dim=1), params)
        assert _rel(x0[eptr[b]:eptr[b + 1]], pd_ref) <= 5e-5, b
        lb = w2_ref.loss_torch(pd_ref, pds[b], assign[eptr[b]:eptr[b + 1]], 2)
        total = total + lb
    assert abs(float(loss0.detach()) - float(total.detach())) <= 1e-5 * max(1.0, abs(float(total.detach())))
    total.backward()
    named = dict(model.named_parameters())
    for name, leaf in leaves.items():
        got = named[name].grad
        assert got is not None, name
        assert _rel(got.reshape(leaf.shape), leaf.grad) <= 1e-4, (name, _rel(got.reshape(leaf.shape), leaf.grad))


def test_gat_layer_backward_with_tied_minima_and_maxima():
    """Several in-neighbours of a node with the same input row send the same message: the minimum / maximum is attained more than
    once.  The kernel gives the tied gradient to the first edge of the row, torch's scatter-reduce splits it evenly -- the inputs'
    gradients then differ between the tied sources, but every PARAMETER gradient is the same (tied sources have equal x_l rows)."""
    import torch
    import torch.nn.functional as F
    from oracle import lp_forward_ref as ref
    from tlc_gnn_amd import ops
    from tlc_gnn_amd.Knowledge_Distillation.gat_conv import GATConv
    rs = np.random.RandomState(4)
    torch.manual_seed(4)
    n, c_in, c_out = 120, 1, 32                               # the first layer: the filtration value is the only input
    ei = _random_graph(rs, n, 900, torch)
    x = torch.tensor(rs.randint(0, 4, size=(n, 1)) / 4.0, dtype=torch.float32)        # four distinct values: ties everywhere
    wl = (torch.randn(c_out, c_in) * 0.5).requires_grad_()
    att = (torch.randn(c_out) * 0.5).requires_grad_()
    wij = (torch.randn(c_out, 2 * c_out) * 0.15).requires_grad_()
    bias = (torch.randn(2 * c_out) * 0.2).requires_grad_()
    out_ref = F.prelu(ref.gat_conv(x, ei, wl, att, wij, bias), torch.tensor(0.1))
    G = torch.randn(n, 2 * c_out)
    (out_ref * G).sum().backward()
    rowptr, col = GATConv.csr_by_target(ei.cuda(), n)
    d = lambda t: t.detach().cuda()
    out = ops.gat_layer(rowptr, col, x.cuda(), d(wl), d(att), d(wij), d(bias), prelu_slope=0.1)
    assert _rel(out, out_ref) <= 2e-5
    _, gwl, gatt, gwij, gbias = ops.gat_layer_bwd(rowptr, col, x.cuda(), d(wl), d(att), d(wij), 0.1, out, G.cuda(), need_gx=False)
    for name, got, want in (("lin_l", gwl, wl.grad), ("att", gatt, att.grad), ("lin_ij", gwij, wij.grad), ("bias", gbias, bias.grad)):
        assert _rel(got, want) <= 1e-4, (name, _rel(got, want))


def test_a_few_optimizer_steps_reduce_the_diagram_loss():
    """The reference's loop (train_Teacher_Model.py:30-66: forward, loss_0.backward(), optimizer.step()) on one small batch: with the
    gradients coming from the device kernels Adam brings the diagram loss down."""
    import torch
    from tlc_gnn_amd.Knowledge_Distillation.Teacher_model import Teacher_Model
    torch.manual_seed(5)
    rs = np.random.RandomState(5)
    n = 50
    ei = _random_graph(rs, n, 160, torch)
    m = ei.shape[1] - n
    f = torch.rand(n, 1).cuda()
    bb = rs.rand(m)
    PD = torch.tensor(np.stack([bb, bb + rs.uniform(0, 0.5, size=m)], 1), dtype=torch.float32).cuda()
    model = Teacher_Model(type='GAT', dropout=0.0).cuda().train()
    opt = torch.optim.Adam(model.parameters(), lr=5e-3)
    losses = []
    for step in range(25):
        opt.zero_grad()
        _, _, loss0, _, _, _, _, _ = model(f, ei.cuda(), PD, kernel='wasserstein', p=2, grad_PI=False)
        loss0.backward()
        opt.step()
        losses.append(float(loss0.detach()))
    assert all(np.isfinite(losses))
    assert min(losses[-5:]) < 0.7 * losses[0], losses


# ---- the evaluation distance: pair_diagonal=True -> wasserstein_distance_inference (wasserstein.py:93-195) ------------------
def _free_problems(rs, B, nmax, style):
    """like _problems but with no relation between n and m (either diagram may be the longer one, either may be empty)"""
    xs, ys = [], []
    for b in range(B):
        n = int(rs.randint(0, nmax + 1)); m = int(rs.randint(0, nmax + 1))
        if style == "grid":
            X = rs.randint(0, 6, size=(n, 2)) / 5.0; Y = rs.randint(0, 6, size=(m, 2)) / 5.0
        else:
            b0 = rs.rand(n); X = np.stack([b0, b0 + rs.uniform(-0.1, 0.6, size=n)], 1)
            b1 = rs.rand(m); Y = np.stack([b1, b1 + rs.uniform(0.0, 0.7, size=m)], 1)
        xs.append(X.astype(np.float64)); ys.append(Y.astype(np.float64))
    return xs, ys


@pytest.mark.parametrize("order", [2, 1])
@pytest.mark.parametrize("style,nmax", [("random", 20), ("grid", 14), ("random", 70), ("random", 200), ("random", 330)])
def test_inference_matching_cost_pieces_and_gradient(order, style, nmax):
    """tlc_w2_inference_matching against scipy's assignment solver on the (n + m)-square expansion of the reference's
    (n+1) x (m+1) transport (oracle/w2_ref.inference_matching): optimal cost, the four returned norms, both maps, the gradient.
    330 points a side: n + m > 512 takes the workgroup-per-problem kernel."""
    import torch
    from tlc_gnn_amd import ops
    from oracle import w2_ref
    rs = np.random.RandomState(7 * nmax + order)
    B = 50 if nmax <= 20 else (10 if nmax <= 70 else 4)
    xs, ys = _free_problems(rs, B, nmax, style)
    xoff = np.concatenate([[0], np.cumsum([len(x) for x in xs])]).astype(np.int64)
    yoff = np.concatenate([[0], np.cumsum([len(y) for y in ys])]).astype(np.int64)
    X = torch.as_tensor(np.concatenate(xs)).cuda()
    Y = torch.as_tensor(np.concatenate(ys)).cuda()
    r = ops.w2_inference_matching(torch.as_tensor(xoff).cuda(), X, torch.as_tensor(yoff).cuda(), Y, order=order, want_grad=True)
    assert (r["status"].cpu().numpy() == 0).all()
    loss, wxy, wxd, wyd = (r[k].cpu().numpy() for k in ("loss", "wxy", "wxd", "wyd"))
    ax_all, ay_all, grad = r["assign_x"].cpu().numpy(), r["assign_y"].cpu().numpy(), r["grad"].cpu().numpy()
    for b in range(B):
        Xb, Yb = xs[b], ys[b]
        n, m = len(Xb), len(Yb)
        ax, ay = ax_all[xoff[b]:xoff[b + 1]], ay_all[yoff[b]:yoff[b + 1]]
        ref = w2_ref.inference_matching(Xb, Yb, order)
        if n == 0 or m == 0:
            # (:98-113) total persistence of the other diagram, the parts reported as 0, nothing matched
            assert abs(loss[b] - ref[0]) <= 1e-12 * max(1.0, ref[0]) and wxy[b] == 0 and wxd[b] == 0 and wyd[b] == 0
            assert (ax == -1).all() and (ay == -1).all()
            continue
        # the two maps are each other's inverse on the matched points
        on = ax >= 0
        assert (ay[ax[on]] == np.flatnonzero(on)).all() and (ay >= 0).sum() == on.sum()
        # optimal: the transport cost of the device's matching equals scipy's optimum
        C = w2_ref.inference_cost_matrix(Xb, Yb, order)
        cost = float(C[np.flatnonzero(on), ax[on]].sum() + C[np.flatnonzero(~on), m].sum() + C[n, np.flatnonzero(ay < 0)].sum())
        assert abs(cost - ref[6]) <= 1e-9 * max(1.0, abs(ref[6])), (b, n, m, cost, ref[6])
        # the returned norms are those of the restated expression on the device's matching
        l2, a2, d2, e2 = w2_ref.inference_loss_from_assignment(Xb, Yb, ax, ay, order)
        for got, want in ((loss[b], l2), (wxy[b], a2), (wxd[b], d2), (wyd[b], e2)):
            assert abs(got - want) <= 1e-12 * max(1.0, want)
        if order == 2 and style != "grid":
            assert abs(loss[b] - ref[0]) <= 1e-9 * max(1.0, ref[0])                 # sqrt(optimal cost): unique (no point below the diagonal counts negative at p = 2)
        if style != "grid":
            Xt = torch.tensor(Xb, requires_grad=True)
            Yt = torch.tensor(Yb)
            parts = []
            if on.any():
                parts.append((Yt[torch.as_tensor(ax[on]).long()] - Xt[torch.as_tensor(on)]).abs().amax(dim=1))
            if (~on).any():
                q = Xt[torch.as_tensor(~on)]
                parts.append(((q[:, 1] - q[:, 0]) * 0.5).abs())
            if (ay < 0).any():
                q = Yt[torch.as_tensor(ay < 0)]
                parts.append(((q[:, 1] - q[:, 0]) * 0.5).abs())
            ((torch.cat(parts) ** order).sum() ** (1.0 / order)).backward()
            assert np.abs(grad[xoff[b]:xoff[b + 1]] - Xt.grad.numpy()).max() <= 1e-12, b


def test_matching_refuses_non_finite_points_and_too_many_points():
    """A NaN / Inf coordinate (a diverging training step) used to leave no column as the minimum of a step: status 3, loss NaN, no
    matching, zero gradient -- and the other problems of the batch are untouched.  n + m > 4 096 in the evaluation form: status 2."""
    import torch
    from tlc_gnn_amd import ops
    rs = np.random.RandomState(5)
    X = [rs.rand(6, 2), rs.rand(5, 2), rs.rand(700, 2), rs.rand(4, 2), rs.rand(3000, 2)]
    Y = [rs.rand(4, 2), rs.rand(5, 2), rs.rand(600, 2), rs.rand(2, 2), rs.rand(1200, 2)]
    X[0][2, 1] = np.nan; Y[1][0, 0] = np.inf; X[2][650, 0] = -np.inf
    xoff = torch.as_tensor(np.concatenate([[0], np.cumsum([len(x) for x in X])]).astype(np.int64)).cuda()
    yoff = torch.as_tensor(np.concatenate([[0], np.cumsum([len(y) for y in Y])]).astype(np.int64)).cuda()
    Xc, Yc = torch.as_tensor(np.concatenate(X)).cuda(), torch.as_tensor(np.concatenate(Y)).cuda()
    r = ops.w2_partial_matching(xoff, Xc, yoff, Yc, order=2)
    torch.cuda.synchronize()
    assert r["status"].cpu().tolist() == [3, 3, 3, 0, 0]
    loss = r["loss"].cpu().numpy()
    assert np.isnan(loss[:3]).all() and np.isfinite(loss[3:]).all() and loss[3] > 0
    assert (r["assign"].cpu().numpy()[:6 + 5 + 700] == -1).all() and float(r["grad"][:6 + 5 + 700].abs().max()) == 0.0
    r = ops.w2_inference_matching(xoff, Xc, yoff, Yc, order=1, want_grad=True)
    torch.cuda.synchronize()
    assert r["status"].cpu().tolist() == [3, 3, 3, 0, 2]
    assert np.isnan(r["loss"].cpu().numpy()[:3]).all() and float(r["loss"][4]) == 0.0
    with pytest.raises(ValueError):                                    # offsets beyond the arrays never reach the device
        ops.w2_partial_matching(xoff, Xc[:100], yoff, Yc, order=2)


def test_teacher_evaluation_with_pair_diagonal_and_target_offsets():
    """The reference scores a trained model with pair_diagonal=True (train_Teacher_Model.py:99 -> Teacher_model.py:66 ->
    wasserstein_distance_inference): loss_0 and the three logged parts against the restatement on the predicted diagram.
    Also: a block-diagonal batch whose targets are NOT one point per edge needs pd_ptr (shorter targets used to be read out of
    bounds), and the per-graph sums equal the graph-by-graph calls."""
    import torch
    from oracle import w2_ref
    from tlc_gnn_amd.Knowledge_Distillation.Teacher_model import Teacher_Model
    torch.manual_seed(5)
    rs = np.random.RandomState(5)
    model = Teacher_Model(type='GAT', dropout=0.2).cuda().eval()
    graphs, fs, pds = [], [], []
    for b in range(5):
        n = int(rs.randint(4, 30))
        ei = _random_graph(rs, n, int(rs.randint(n, 3 * n)), torch)
        m = ei.shape[1] - n
        graphs.append((n, ei[:, :m])); fs.append(torch.rand(n, 1))
        k = int(rs.randint(0, m + 6))                                   # more or fewer target points than edges, or none
        bb = rs.rand(k)
        pds.append(torch.tensor(np.stack([bb, bb + rs.uniform(0, 0.6, size=k)], 1).reshape(k, 2), dtype=torch.float32))
    per_graph = []
    with torch.no_grad():
        for (n, e), f, PD in zip(graphs, fs, pds):
            lp = torch.arange(n, dtype=torch.int64)
            ei = torch.cat([e, torch.stack([lp, lp])], dim=1).cuda()
            for p in (1, 2):
                x0, img, l0, lxy, lxd, lyd, _, _ = model(f.cuda(), ei, PD.cuda(), kernel='wasserstein', p=p, pair_diagonal=True, grad_PI=False)
                want = w2_ref.inference_matching(x0.double().cpu().numpy(), PD.double().numpy(), p)
                assert abs(float(l0) - want[0]) <= 1e-5 * max(1.0, want[0]), (p, float(l0), want[0])
                if p == 2:
                    per_graph.append([float(l0), float(lxy), float(lxd), float(lyd)])
                    tot = float(lxy) ** 2 + float(lxd) ** 2 + float(lyd) ** 2
                    assert abs(tot - float(l0) ** 2) <= 1e-5 * max(1.0, tot)
        gptr = np.concatenate([[0], np.cumsum([g[0] for g in graphs])])
        eptr = np.concatenate([[0], np.cumsum([g[1].shape[1] for g in graphs])])
        pptr = np.concatenate([[0], np.cumsum([len(q) for q in pds])])
        N = int(gptr[-1])
        ei_all = torch.cat([g[1] + int(gptr[b]) for b, g in enumerate(graphs)], dim=1)
        loops = torch.arange(N, dtype=torch.int64)
        ei_full = torch.cat([ei_all, torch.stack([loops, loops])], dim=1).cuda()
        kw = dict(kernel='wasserstein', p=2, pair_diagonal=True, grad_PI=False, graph_ptr=torch.tensor(gptr).cuda(), edge_ptr=torch.tensor(eptr).cuda())
        if int(pptr[-1]) != int(eptr[-1]):
            with pytest.raises(ValueError):
                model(torch.cat(fs).cuda(), ei_full, torch.cat(pds).cuda(), **kw)
        out = model(torch.cat(fs).cuda(), ei_full, torch.cat(pds).cuda(), pd_ptr=torch.tensor(pptr).cuda(), **kw)
        want = np.array(per_graph).sum(axis=0)
        for got, w in zip(out[2:6], want):
            assert abs(float(got) - w) <= 1e-4 * max(1.0, w)
    # the training form cannot take fewer predicted than target points: a clear error, not a device fault
    short = [i for i, (g, q) in enumerate(zip(graphs, pds)) if len(q) > g[1].shape[1]]
    if short:
        n, e = graphs[short[0]]
        lp = torch.arange(n, dtype=torch.int64)
        with pytest.raises(ValueError):
            model(fs[short[0]].cuda(), torch.cat([e, torch.stack([lp, lp])], dim=1).cuda(), pds[short[0]].cuda(), kernel='wasserstein', p=2, grad_PI=False)

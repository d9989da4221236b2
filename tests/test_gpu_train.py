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
    X = [rs.rand(3, 2), np.zeros((0, 2)), rs.rand(5, 2), rs.rand(600, 2), np.zeros((0, 2))]
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

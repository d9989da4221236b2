"""CPU restatement of PDGNN's diagram loss.  TEST INFRASTRUCTURE ONLY (tests/, never imported by the product).

Restates /root/reference/Knowledge_Distillation/wasserstein.py:
  _dist_to_diag :30-42, _build_dist_matrix :45-67, wasserstein_distance(X, Y, order, internal_p=inf, enable_autodiff=True,
  num_models=1) :198-379  (the call of Teacher_model.py:131, compute_PD_loss(kernel='wasserstein'))
  wasserstein_distance_inference(X, Y, order, internal_p=inf, enable_autodiff=True) :93-195  (Teacher_model.py:134-136,
  compute_PD_loss(type='inference'), reached with pair_diagonal=True from train_Teacher_Model.py:99)

PARITY UNPINNED: the reference solves the transport with POT's `ot.emd` (third-party, not installed, not installable here).
With unit masses on the points the transport polytope's vertices are assignments, so the optimum is an assignment problem; it is
solved here with scipy.optimize.linear_sum_assignment on the cost matrix expanded by n - m copies of the diagonal column.  The
optimal COST is unique; which optimal assignment `ot.emd` would return among ties is not reproduced.
"""
import numpy as np
from scipy.optimize import linear_sum_assignment


def dist_to_diag(X):
    """:30-42 with internal_p = inf: (death - birth) * 2 ** (1/inf - 1), signed."""
    X = np.asarray(X, dtype=np.float64).reshape(-1, 2)
    return (X[:, 1] - X[:, 0]) * 0.5


def cost_matrix(X, Y, order):
    """:45-67 without the last row (the call passes M[:-1, :], :289): [n, m + 1], last column = the diagonal."""
    X = np.asarray(X, dtype=np.float64).reshape(-1, 2)
    Y = np.asarray(Y, dtype=np.float64).reshape(-1, 2)
    C = np.abs(X[:, None, :] - Y[None, :, :]).max(axis=2) ** order if len(Y) else np.zeros((len(X), 0))
    return np.hstack([C, (dist_to_diag(X) ** order)[:, None]])


def partial_matching(X, Y, order):
    """-> (loss, wxy, wxd, assign[n] (target index or -1), optimal cost) for n >= m >= 0, num_models = 1."""
    X = np.asarray(X, dtype=np.float64).reshape(-1, 2)
    Y = np.asarray(Y, dtype=np.float64).reshape(-1, 2)
    n, m = len(X), len(Y)
    assert n >= m, "the diagonal would get negative mass (:264)"
    if n == 0:
        return 0.0, 0.0, 0.0, np.zeros(0, dtype=np.int64), 0.0
    M = cost_matrix(X, Y, order)
    big = np.hstack([M[:, :m], np.repeat(M[:, m:], n - m, axis=1)]) if n > m else M[:, :m]
    rows, cols = linear_sum_assignment(big)
    assign = np.full(n, -1, dtype=np.int64)
    assign[rows] = np.where(cols < m, cols, -1)
    cost = float(big[rows, cols].sum())
    loss, wxy, wxd = loss_from_assignment(X, Y, assign, order)
    return loss, wxy, wxd, assign, cost


def loss_from_assignment(X, Y, assign, order):
    """:303-372: d_k = ||Y_j - X_i||_inf for the matched pairs, |dist_to_diag(X_i)| for the points sent to the diagonal (all of them
    are kept when num_models = 1); loss = (sum d_k^p)^(1/p), wxy / wxd = the same norm over either group."""
    X = np.asarray(X, dtype=np.float64).reshape(-1, 2)
    Y = np.asarray(Y, dtype=np.float64).reshape(-1, 2)
    assign = np.asarray(assign)
    on = assign >= 0
    dxy = np.abs(Y[assign[on]] - X[on]).max(axis=1) if on.any() else np.zeros(0)
    dxd = np.abs(dist_to_diag(X[~on]))
    norm = lambda d: float((d ** order).sum() ** (1.0 / order)) if len(d) else 0.0
    return norm(np.concatenate([dxy, dxd])), norm(dxy), norm(dxd)


def loss_torch(X, Y, assign, order):
    """The same expression on torch tensors (X requires grad): what autograd differentiates in the reference (:311-372)."""
    import torch
    assign = torch.as_tensor(np.asarray(assign))
    on = assign >= 0
    parts = []
    if bool(on.any()):
        parts.append((Y[assign[on]] - X[on]).abs().amax(dim=1))
    if bool((~on).any()):
        parts.append(((X[~on][:, 1] - X[~on][:, 0]) * 0.5).abs())
    d = torch.cat(parts)
    return (d ** order).sum() ** (1.0 / order)


# ---- the evaluation distance: wasserstein_distance_inference (:93-195), reached with pair_diagonal=True ----------------------------
def inference_cost_matrix(X, Y, order):
    """_build_dist_matrix :45-67 in full: [(n+1), (m+1)], last column = X to the diagonal, last row = Y to the diagonal, corner 0."""
    X = np.asarray(X, dtype=np.float64).reshape(-1, 2)
    Y = np.asarray(Y, dtype=np.float64).reshape(-1, 2)
    n, m = len(X), len(Y)
    C = np.zeros((n + 1, m + 1))
    if n and m:
        C[:n, :m] = np.abs(X[:, None, :] - Y[None, :, :]).max(axis=2) ** order
    C[:n, m] = dist_to_diag(X) ** order
    C[n, :m] = dist_to_diag(Y) ** order
    return C


def inference_matching(X, Y, order):
    """-> (loss, wxy, wxd, wyd, assign_x[n], assign_y[m], optimal cost).  Masses a = [1]*n + [m], b = [1]*m + [n] (:127-131): with
    integer masses the transport has an integral optimum = an assignment of n + m rows (points of X, then m copies of the diagonal)
    onto n + m columns (points of Y, then n copies of the diagonal).  Empty diagrams (:98-113): the total persistence of the other
    one, the three parts 0."""
    X = np.asarray(X, dtype=np.float64).reshape(-1, 2)
    Y = np.asarray(Y, dtype=np.float64).reshape(-1, 2)
    n, m = len(X), len(Y)
    norm = lambda d: float((np.abs(d) ** order).sum() ** (1.0 / order)) if len(d) else 0.0
    if n == 0 or m == 0:
        return norm(dist_to_diag(X if n else Y)), 0.0, 0.0, 0.0, np.full(n, -1, dtype=np.int64), np.full(m, -1, dtype=np.int64), None
    C = inference_cost_matrix(X, Y, order)
    big = np.zeros((n + m, n + m))
    big[:n, :m] = C[:n, :m]
    big[:n, m:] = C[:n, m][:, None]
    big[n:, :m] = C[n, :m][None, :]
    rows, cols = linear_sum_assignment(big)
    ax = np.full(n, -1, dtype=np.int64)
    ay = np.full(m, -1, dtype=np.int64)
    for r, c in zip(rows, cols):
        if r < n and c < m:
            ax[r] = c
            ay[c] = r
    cost = float(big[rows, cols].sum())
    return inference_loss_from_assignment(X, Y, ax, ay, order) + (ax, ay, cost)


def inference_loss_from_assignment(X, Y, ax, ay, order):
    """:140-181: matched pairs ||Y_j - X_i||_inf, |dist_to_diag| of the X points and of the Y points sent to the diagonal;
    loss = the order-norm of all of them, wxy / wxd / wyd = of each group."""
    X = np.asarray(X, dtype=np.float64).reshape(-1, 2)
    Y = np.asarray(Y, dtype=np.float64).reshape(-1, 2)
    ax, ay = np.asarray(ax), np.asarray(ay)
    on = ax >= 0
    dxy = np.abs(Y[ax[on]] - X[on]).max(axis=1) if on.any() else np.zeros(0)
    dxd = np.abs(dist_to_diag(X[~on]))
    dyd = np.abs(dist_to_diag(Y[ay < 0]))
    norm = lambda d: float((d ** order).sum() ** (1.0 / order)) if len(d) else 0.0
    return norm(np.concatenate([dxy, dxd, dyd])), norm(dxy), norm(dxd), norm(dyd)

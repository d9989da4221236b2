"""The CPU restatement of the curvature step (oracle/ricci_ref.py, parity unpinned) against facts that do not depend on it:
the exact transport cost from a linear programme, and the closed forms of Ollivier's curvature on small graphs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _emd(a, b, M):
    from scipy.optimize import linprog
    na, nb = len(a), len(b)
    A = np.zeros((na + nb, na * nb))
    for i in range(na):
        A[i, i * nb:(i + 1) * nb] = 1
    for j in range(nb):
        A[na + j, j::nb] = 1
    r = linprog(M.reshape(-1), A_eq=A, b_eq=np.concatenate([a, b]), bounds=(0, None), method="highs")
    assert r.success
    return r.fun


def test_sinkhorn_cost_brackets_the_exact_transport_cost():
    from oracle import ricci_ref
    rs = np.random.RandomState(0)
    for _ in range(20):
        na, nb = rs.randint(2, 9), rs.randint(2, 9)
        a = np.concatenate([np.full(na - 1, 0.5 / (na - 1)), [0.5]])
        b = np.concatenate([np.full(nb - 1, 0.5 / (nb - 1)), [0.5]])
        M = rs.randint(0, 4, size=(na, nb)).astype(float)
        w, it = ricci_ref.sinkhorn2(a, b, M)
        exact = _emd(a, b, M)
        assert 0 < it <= 1000
        assert w >= exact - 2e-3 and w - exact < 0.12            # entropic plan (feasible up to the stopping rule): close to, essentially never below, the optimum


def test_curvature_of_small_graphs():
    from oracle import ricci_ref
    # complete graph K5, alpha 0.5: m_s and m_t differ only by (alpha - (1-alpha)/4) at s and t -> W = 0.375, kappa = 0.625
    k5 = np.array([[i, j] for i in range(5) for j in range(i + 1, 5)])
    kap, _ = ricci_ref.ollivier_ricci_sinkhorn(5, k5)
    assert np.abs(kap - 0.625).max() < 0.02 and np.ptp(kap) < 1e-6
    # a long path: interior edges are flat (kappa = 0), the end edges have kappa = 0.5 at alpha 0.5 ... W_1 = 0.5
    path = np.stack([np.arange(9), np.arange(1, 10)], 1)
    kap, _ = ricci_ref.ollivier_ricci_sinkhorn(10, path)
    assert np.abs(kap[2:-2]).max() < 0.02 and abs(kap[0] - kap[-1]) < 1e-6
    # a star: every edge has the same curvature; tree edges with a high-degree endpoint are negative or zero
    star = np.stack([np.zeros(8, dtype=int), np.arange(1, 9)], 1)
    kap, _ = ricci_ref.ollivier_ricci_sinkhorn(9, star)
    assert np.ptp(kap) < 1e-6

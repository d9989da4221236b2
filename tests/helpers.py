"""Shared helpers for the parity tests."""
import numpy as np


def sorted_points(a):
    """Diagram as a sorted multiset: point order carries no meaning."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, 2)
    if len(a) == 0:
        return a
    return a[np.lexsort((a[:, 1], a[:, 0]))]


def same_multiset(a, b):
    a, b = sorted_points(a), sorted_points(b)
    return a.shape == b.shape and np.array_equal(a, b)


def ragged_slice(flat, offs, i):
    return flat[offs[i]:offs[i + 1]]


def rel_err(a, b, floor=1e-300):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b) / np.maximum(np.abs(b), floor)


def csr_from_golden(d):
    from tlc_gnn_amd import synth
    return synth.edges_to_csr(int(d["n_nodes"]), d["edges"], d["kappa"])

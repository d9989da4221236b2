"""Host logic of the streamed split / sparse image store -- no GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
G = os.path.join(ROOT, "tests", "golden")


def test_legacy_shuffle_draws_do_not_depend_on_row_width():
    """ShuffledNegatives shuffles list NUMBERS where the reference shuffles the [n_neg, 2] array (loaddatas.py:46): numpy's
    legacy shuffle draws one random_interval per row either way, so the permutation is the same."""
    for n in (1, 2, 3, 17, 1000, 70001):
        np.random.seed(1234)
        np.random.shuffle(np.zeros((5, 2)))                       # something before it, as in the reference
        rows = np.stack([np.arange(n), np.arange(n) * 7 + 1], 1)
        np.random.shuffle(rows)
        after_rows = np.random.randint(0, 1 << 30)
        np.random.seed(1234)
        np.random.shuffle(np.zeros((5, 2)))
        perm = np.arange(n, dtype=np.int64)
        np.random.shuffle(perm)
        after_perm = np.random.randint(0, 1 << 30)
        assert np.array_equal(rows[:, 0], perm) and np.array_equal(rows[:, 1], perm * 7 + 1)
        assert after_rows == after_perm                           # and the stream continues at the same place


def test_oracle_complement_matches_the_golden_negative_list_g7():
    from oracle import oracle
    d = np.load(os.path.join(G, "adj_split.npz"))
    n, edges = int(d["n_nodes"]), d["edges"]
    a = np.zeros((n, n))
    a[edges[:, 0], edges[:, 1]] = 1
    a[edges[:, 1], edges[:, 0]] = 1
    comp = oracle.complement_pairs_dense(a)
    n_neg = len(comp)
    neg_shuffled = d["train_edges_false"][:n_neg]                 # neg_edges ++ val ++ test (loaddatas.py:53)
    assert n_neg == n * (n + 1) // 2 - len(np.unique(np.sort(edges, 1), axis=0))
    key = lambda p: p[:, 0] * n + p[:, 1]
    assert np.array_equal(np.sort(key(neg_shuffled)), key(comp))  # same set; comp is the sorted (row-major) order
    assert np.array_equal(d["val_edges_false"], neg_shuffled[: len(d["val_edges_false"])])


def test_sparse_images_behave_like_the_dense_array(tmp_path):
    from tlc_gnn_amd.pi_cache import SparseImages
    rs = np.random.RandomState(0)
    dense = np.zeros((1000, 25))
    rows = rs.choice(1000, 37, replace=False)
    dense[rows] = rs.rand(37, 25)
    status = np.zeros(1000, dtype=np.uint8)
    status[[5, 999]] = [2, 4]
    dense[[5, 999]] = 0
    sp_ = SparseImages.from_dense(dense, status)
    assert sp_.shape == (1000, 25) and len(sp_.idx) == len(set(rows.tolist()) | {5, 999})
    assert np.array_equal(sp_.to_dense(), dense)
    idx = rs.randint(0, 1000, 500)
    assert np.array_equal(sp_[idx], dense[idx])
    assert np.array_equal(sp_[10:200], dense[10:200])
    assert sp_.cnt_compute == 998 and sp_.status_counts.tolist() == [998, 0, 1, 0, 1, 0, 0, 0]
    f = str(tmp_path / "pi.npz")
    sp_.save(f)
    back = SparseImages.load(f)
    assert np.array_equal(back.to_dense(), dense) and np.array_equal(back.status, sp_.status)
    assert np.array_equal(back.status_counts, sp_.status_counts)
    empty = SparseImages.from_dense(np.zeros((7, 25)))
    assert np.array_equal(empty[np.array([0, 6])], np.zeros((2, 25)))


def test_lazy_pair_list_over_plain_arrays():
    from tlc_gnn_amd.pi_cache import LazyPairList
    a = np.arange(10).reshape(5, 2)
    b = np.arange(100, 106).reshape(3, 2)
    lp = LazyPairList([(a, 1), (b, 0)])
    full = np.concatenate([a, b])
    assert len(lp) == 8
    idx = np.array([7, 0, 4, 5, 5])
    assert np.array_equal(lp.gather(idx), full[idx])
    assert np.array_equal(lp.labels(idx), np.array([0, 1, 1, 0, 0]))

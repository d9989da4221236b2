"""Development aid: tlc_pd_from_filtration on graphs with many Pos edges (divide-and-conquer cycle swap) vs the CPU checker."""
import sys
import numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))); sys.path.insert(0, "tests")
from tlc_gnn_amd import engine
from oracle import oracle
from helpers import same_multiset
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
def graph(n, extra):
    perm = rs.permutation(n)
    E = {(min(int(perm[i]), int(perm[j])), max(int(perm[i]), int(perm[j]))) for i in range(1, n) for j in [rs.randint(0, i)]}
    while len(E) < n - 1 + extra:
        a, b = rs.randint(0, n, 2)
        if a != b: E.add((min(a, b), max(a, b)))
    return np.array(sorted(E), dtype=np.int32)
for trial, (n, extra, ties) in enumerate([(1500, 750, False), (1200, 1200, False), (1000, 400, True), (600, 3000, False), (2000, 170, False), (300, 200, False)]):
    e = graph(n, extra)
    f = (rs.randint(0, 50, n) / 49.0) if ties else rs.rand(n)
    no = np.array([0, n], dtype=np.int64); eo = np.array([0, len(e)], dtype=np.int64)
    for flags in (0, 1):
        ro = oracle.pd_from_filtration(no, eo, e, f, flags)
        ref = {"one": [ro["one"][:ro["counts"][0][2]]]}
        res = []
        for it in range(3):
            r = engine.pd_from_filtration(torch.from_numpy(no).cuda(), torch.from_numpy(eo).cuda(), torch.from_numpy(e).cuda(), torch.from_numpy(f).cuda(), flags)
            c = r["counts"][0].cpu().numpy()
            one = r["one"][:c[2]].cpu().numpy()
            res.append(one)
        ok = [same_multiset(x, ref["one"][0]) for x in res]
        det = [np.array_equal(res[0], x) for x in res[1:]]
        print("n=%d m=%d K=%d ties=%s flags=%d: multiset ok %s, identical order across runs %s, count %d vs %d" % (n, len(e), len(e) - n + 1, ties, flags, ok, det, len(res[0]), len(ref["one"][0])))
        if not all(ok):
            a = res[0][np.lexsort((res[0][:, 1], res[0][:, 0]))]; b = np.asarray(ref["one"][0]); b = b[np.lexsort((b[:, 1], b[:, 0]))]
            if a.shape == b.shape:
                d = np.nonzero((a != b).any(1))[0]
                print("   differing sorted rows:", len(d), a[d[:3]].tolist(), b[d[:3]].tolist())

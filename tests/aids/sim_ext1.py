"""CPU simulation of the cycle swap on the heaviest vicinities of the bench graph: distribution of the two walk lengths
(a = steps from p to the meeting node, b = from q) under the GPU's tree orientation (root = lowest f) -- development aid."""
import sys, collections
import numpy as np
sys.path.insert(0, ".")
import bench
from oracle import oracle

wl = bench.build_workload(0)
rowptr, col, w, pairs, hop = wl["rowptr"], wl["col"], wl["w"], wl["pi_pairs"], wl["hop"]
deg = np.diff(rowptr)
score = deg[pairs[:, 0]] + deg[pairs[:, 1]]
heavy = pairs[np.argsort(-score)[:40]]
no, ids, f, n_, m_, st, eo, edges = oracle.vicinity_filtration(rowptr, col, w, heavy, hop, cap=4096, edge_cap=16384)
best = np.argsort(-m_)[:3]
for gi in best:
    n, m = int(n_[gi]), int(m_[gi])
    fv = f[no[gi]:no[gi] + n]
    E = edges[eo[gi]:eo[gi] + m]
    rank = np.argsort(np.argsort(fv, kind="stable"), kind="stable")
    lo = np.minimum(rank[E[:, 0]], rank[E[:, 1]]); hi = np.maximum(rank[E[:, 0]], rank[E[:, 1]])
    fr = np.sort(fv, kind="stable")
    asc = fr[hi] + (fr[lo] + 1) * 1e-6
    desc = fr[lo] - (101 - fr[hi]) * 1e-6
    order = np.lexsort((np.arange(m), -desc))          # descending pass order
    comp = list(range(n))
    def find(x):
        while comp[x] != x:
            comp[x] = comp[comp[x]]; x = comp[x]
        return x
    neg, pos = [], []
    for e in order:
        a, b = find(lo[e]), find(hi[e])
        if a != b: comp[a] = b; neg.append(e)
        else: pos.append(e)
    adj = collections.defaultdict(list)
    for e in neg: adj[lo[e]].append((hi[e], e)); adj[hi[e]].append((lo[e], e))
    par = [-1] * n; pk = [0.0] * n
    par[0] = 0
    q = [0]
    for x in q:
        for y, e in adj[x]:
            if par[y] < 0: par[y] = x; pk[y] = asc[e]; q.append(y)
    hist = collections.Counter(); hl = collections.Counter()
    for e in pos:
        p, qq = int(lo[e]), int(hi[e])
        mark = {}
        x, s = p, 0
        while True:
            mark[x] = s
            if x == 0: break
            x = par[x]; s += 1
        x, b = qq, 0
        while x not in mark: x = par[x]; b += 1
        meet, a = x, mark[x]
        hist[(min(a, 3), min(b, 3))] += 1
        hl[min(a + b, 12)] += 1
        # max edge on loop
        bestn, bv, side = -1, -1.0, 0
        x = p
        while x != meet:
            if pk[x] > bv: bv, bestn, side = pk[x], x, 0
            x = par[x]
        x = qq
        while x != meet:
            if pk[x] > bv: bv, bestn, side = pk[x], x, 1
            x = par[x]
        node, nodec, kin = (p, qq, asc[e]) if side == 0 else (qq, p, asc[e])
        while nodec != bestn:
            pp, kk = par[node], pk[node]
            par[node], pk[node] = nodec, kin
            nodec, node, kin = node, pp, kk
    tot = sum(hist.values())
    print("n=%d m=%d pos=%d" % (n, m, len(pos)))
    print("  (a,b) capped at 3:", {k: round(v / tot, 3) for k, v in sorted(hist.items())})
    print("  a+b:", {k: round(v / tot, 3) for k, v in sorted(hl.items())})

"""CPU model of the divide-and-conquer cycle swap (development aid; the device code in pd_pipeline.hip follows it array for array).

Accelerate_PD (accelerated_PD.py:115-178) inserts the Pos edges e_1..e_K one at a time into the spanning tree of the Neg edges and
removes the heaviest ('asc') tree edge of the cycle each closes: an incremental minimum spanning tree w.r.t. the ascending ranks.
The edge removed by e_k is what the diagram point of e_k needs.  Offline, every edge's DELETION TIME can be found by a binary
search that all edges run together: a segment [l, r) of insertion times keeps
   P: edges alive at l that die inside the segment, Q: the edges inserted inside it,
on supernodes = the tree edges alive throughout the segment, contracted.  T_mid = MSF(P + Q[l, mid)) splits both lists between
the two halves; after log2 K levels every segment holds one query and the one P edge it removes.
"""
import sys
import numpy as np


def serial_swap(n, ends, asc, neg, pos):
    """the reference's loop: returns removed[k] = edge id removed by pos[k]"""
    adj = {}
    for e in neg:
        a, b = ends[e]
        adj.setdefault(a, []).append((b, e)); adj.setdefault(b, []).append((a, e))
    root = ends[neg[0]][0]
    par, pe = {root: root}, {}
    q = [root]
    for x in q:
        for y, e in adj.get(x, []):
            if y not in par:
                par[y] = x; pe[y] = e; q.append(y)
    out = []
    for e in pos:
        p, qq = ends[e]
        seen = {}
        x = p
        while True:
            seen[x] = True
            if x == root: break
            x = par[x]
        x = qq
        while x not in seen: x = par[x]
        meet = x
        best, side = None, 0
        for s, start in ((0, p), (1, qq)):
            x = start
            while x != meet:
                if best is None or asc[pe[x]] > asc[pe[best]]: best, side = x, s
                x = par[x]
        out.append(pe[best])
        node, nodec, ein = (p, qq, e) if side == 0 else (qq, p, e)
        while True:
            pp, ee = par[node], pe.get(node)
            par[node], pe[node] = nodec, ein
            if node == best: break
            nodec, node, ein = node, pp, ee
    return out


def msf(n_nodes, items):
    """items: list of (w, a, b, tag); returns set of tags in the minimum spanning forest (Kruskal; the device runs Boruvka)"""
    comp = list(range(n_nodes))
    def find(x):
        while comp[x] != x:
            comp[x] = comp[comp[x]]; x = comp[x]
        return x
    keep = set()
    for w, a, b, tag in sorted(items):
        ra, rb = find(a), find(b)
        if ra != rb:
            comp[ra] = rb; keep.add(tag)
    return keep


def components(n_nodes, edges):
    comp = list(range(n_nodes))
    def find(x):
        while comp[x] != x:
            comp[x] = comp[comp[x]]; x = comp[x]
        return x
    for a, b in edges:
        ra, rb = find(a), find(b)
        if ra != rb: comp[max(ra, rb)] = min(ra, rb)
    return [find(x) for x in range(n_nodes)]


def dc_swap(n, ends, asc, neg, pos, in_final):
    """level-synchronous binary search on deletion times.  in_final[e]: e is in the ascending-pass spanning tree of the whole graph."""
    K = len(pos)
    if K == 0:
        return []
    # root segment: supernodes = components of the Neg edges that are never removed
    lab = components(n, [ends[e] for e in neg if in_final[e]])
    # slots: Q[k] for query k, P[d] for each edge that is removed at some time (d = 0..K-1)
    Qab = [(lab[ends[e][0]], lab[ends[e][1]]) for e in pos]
    Qalive = [bool(in_final[e]) for e in pos]           # alive at the END of its segment
    Qseg = [0] * K
    dying = [e for e in neg if not in_final[e]] + [e for e in pos if not in_final[e]]
    assert len(dying) == K, (len(dying), K)
    pslot = {e: i for i, e in enumerate(dying)}
    Pab = [None] * K
    Pseg = [None] * K                                   # None: not spawned yet
    for e in neg:
        if not in_final[e]:
            Pab[pslot[e]] = (lab[ends[e][0]], lab[ends[e][1]]); Pseg[pslot[e]] = 0
    n_sup = n                                           # supernode id space of this level
    level, nseg = 0, 1
    bound = lambda lvl, j: (j * K) >> lvl               # segment j of level lvl = [bound(j), bound(j+1))
    while (K >> level) > 1 or any(bound(level, j + 1) - bound(level, j) > 1 for j in range(nseg)):
        mid = [bound(level + 1, 2 * j + 1) for j in range(nseg)]
        items = []
        for d in range(K):
            if Pseg[d] is not None:
                items.append((asc[dying[d]], Pab[d][0], Pab[d][1], ('P', d)))
        for k in range(K):
            if k < mid[Qseg[k]]:
                items.append((asc[pos[k]], Qab[k][0], Qab[k][1], ('Q', k)))
        tmid = msf(n_sup, items)
        # contraction sets and routing
        FL, FR = [], []
        newP = []
        for d in range(K):
            if Pseg[d] is None: continue
            if ('P', d) in tmid: FL.append(Pab[d])              # alive at l and at mid: contracted in the left child
        for k in range(K):
            if k < mid[Qseg[k]] and ('Q', k) in tmid and Qalive[k]:
                FR.append(Qab[k])                               # alive at mid and at r: contracted in the right child
        labL, labR = components(n_sup, FL), components(n_sup, FR)
        nxt_Pab, nxt_Pseg = list(Pab), list(Pseg)
        for d in range(K):
            if Pseg[d] is None: continue
            j = Pseg[d]
            if ('P', d) in tmid:
                nxt_Pseg[d] = 2 * j + 1; nxt_Pab[d] = (labR[Pab[d][0]] + n_sup, labR[Pab[d][1]] + n_sup)
            else:
                nxt_Pseg[d] = 2 * j; nxt_Pab[d] = (labL[Pab[d][0]], labL[Pab[d][1]])
        for k in range(K):
            j = Qseg[k]
            if k < mid[j]:
                intm = ('Q', k) in tmid
                if intm and not Qalive[k]:                      # dies in the right half: a P copy starts there
                    d = pslot[pos[k]]
                    assert nxt_Pseg[d] is None
                    nxt_Pseg[d] = 2 * j + 1; nxt_Pab[d] = (labR[Qab[k][0]] + n_sup, labR[Qab[k][1]] + n_sup)
                Qalive[k] = intm
                Qseg[k] = 2 * j; Qab[k] = (labL[Qab[k][0]], labL[Qab[k][1]])
            else:
                Qseg[k] = 2 * j + 1; Qab[k] = (labR[Qab[k][0]] + n_sup, labR[Qab[k][1]] + n_sup)
        Pab, Pseg = nxt_Pab, nxt_Pseg
        # dense renumbering of the supernodes in use (left copies live in [0, n_sup), right copies in [n_sup, 2 n_sup))
        used = sorted({x for ab in Qab for x in ab} | {x for d in range(K) if Pseg[d] is not None for x in Pab[d]})
        ren = {x: i for i, x in enumerate(used)}
        Qab = [(ren[a], ren[b]) for a, b in Qab]
        Pab = [None if Pseg[d] is None else (ren[Pab[d][0]], ren[Pab[d][1]]) for d in range(K)]
        for k in range(K): assert Qab[k][0] != Qab[k][1]
        n_sup = len(used)
        level += 1; nseg *= 2
        assert n_sup <= 2 * K + nseg, (n_sup, K, nseg)
    # every segment now holds one query and the one P edge it removes
    out = [None] * K
    for d in range(K):
        assert Pseg[d] is not None
        k = bound(level, Pseg[d])
        assert bound(level, Pseg[d] + 1) - k == 1 and out[k] is None
        out[k] = dying[d]
    return out


def instance(rs, n, m_extra, ties):
    perm = rs.permutation(n)
    E = [(int(perm[i]), int(perm[rs.randint(0, i)])) for i in range(1, n)]
    have = set(tuple(sorted(e)) for e in E)
    while len(E) < n - 1 + m_extra:
        a, b = rs.randint(0, n, 2)
        if a != b and tuple(sorted((a, b))) not in have:
            have.add(tuple(sorted((a, b)))); E.append((int(a), int(b)))
    f = rs.randint(0, 4, n) / 3.0 if ties else rs.rand(n)
    rank = np.argsort(np.argsort(f, kind="stable"), kind="stable")
    fr = np.sort(f, kind="stable")
    ends = [(int(min(rank[a], rank[b])), int(max(rank[a], rank[b]))) for a, b in E]
    m = len(E)
    asc_k = np.array([fr[h] + (fr[l] + 1) * 1e-6 for l, h in ends]); desc_k = np.array([fr[l] - (101 - fr[h]) * 1e-6 for l, h in ends])
    order = np.lexsort((rs.permutation(m), -desc_k))            # descending pass order, ties in any order
    dpos = np.empty(m, dtype=np.int64); dpos[order] = np.arange(m)
    # ascending ranks: equal keys are ranked by DESCENDING position in the descending pass (an edge that closes a cycle then
    # ranks below every equal-key edge of that cycle, all of which come earlier in the descending pass): this is what makes
    # the reference's swap an incremental-MST update
    asc = np.empty(m, dtype=np.int64); asc[np.lexsort((-dpos, asc_k))] = np.arange(m)
    comp = list(range(n))
    def find(x):
        while comp[x] != x:
            comp[x] = comp[comp[x]]; x = comp[x]
        return x
    neg, pos = [], []
    for e in order:
        a, b = find(ends[e][0]), find(ends[e][1])
        if a != b: comp[a] = b; neg.append(int(e))
        else: pos.append(int(e))
    fin = msf(n, [(asc[e], ends[e][0], ends[e][1], e) for e in range(m)])
    in_final = [e in fin for e in range(m)]
    return ends, asc, neg, pos, in_final


if __name__ == "__main__":
    rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    for trial in range(300):
        n = int(rs.randint(2, 60)); extra = int(rs.randint(0, 3 * n))
        extra = min(extra, n * (n - 1) // 2 - (n - 1))
        ends, asc, neg, pos, fin = instance(rs, n, extra, ties=bool(trial & 1))
        ref = serial_swap(n, ends, asc, neg, pos)
        got = dc_swap(n, ends, asc, neg, pos, fin)
        assert ref == got, (trial, n, extra, ref, got)
    print("ok: 300 instances, divide-and-conquer == serial swap, query for query")

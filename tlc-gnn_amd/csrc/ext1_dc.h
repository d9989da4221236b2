// ext1_dc.h -- Accelerate_PD (accelerated_PD.py:115-178) without the serial loop over the Pos edges.
//
// The reference inserts the Pos edges e_1 .. e_K (descending-pass order) one at a time into the spanning tree of the Neg
// edges and removes the heaviest ('asc') tree edge of the cycle each one closes.  With the ascending RANKS as weights --
// equal keys ranked so that the edge that comes earlier in the descending pass is the heavier one, see fix_desc_ties --
// the inserted edge is never the heaviest of its own cycle, so every step is an incremental minimum-spanning-tree update
// and the edge removed at step k is "the edge whose deletion time is k".  Deletion times can be found OFFLINE by a binary
// search that all edges run together (the classic offline-dynamic-MST contraction): a segment [l, r) of insertion times
// holds
//     P: edges alive at time l that die inside the segment      Q: the edges inserted inside the segment
// on supernodes = the tree edges alive throughout the segment, contracted.  T_mid = MSF(P + Q[l, mid)) sends every P
// edge to the half it dies in, the Q edges to the half they are inserted in, and a Q edge that survives to mid but not to
// r starts a P copy in the right half.  After ceil(log2 K) levels every segment is one query with the one P edge it
// removes.  All segments of a level live in one id space and are processed by the same Boruvka rounds, so a level costs a
// few dozen barriers whatever K is: ~log2 K * 30k cycles for the workgroup instead of K * 2k for one lane pair.
// tests/aids/sim_dc_ext1.py is the CPU model of exactly these arrays (checked query for query against the serial loop).
//
// If the ranks do not make the process an MST update (only possible through floating-point rounding of keys that differ in
// the last bits) some query ends without its edge: the stage reports failure and the caller runs the serial walk.
#pragma once

#define TLC_DC_MIN_POS 160          /* below this many Pos edges the serial walk wins */
#define TLC_DC_MAX_TIE_RUN 64       /* longer runs of equal descending keys: no fix-up, serial walk */

namespace {

struct DcMem {
    unsigned *Qab, *Pab;                          // [K] endpoints (a << 16 | b) in the level's supernode ids
    unsigned short *Qseg, *Qw, *Pseg, *Pw, *Phin; // [K]; Qseg bit 15: alive at the END of its segment; Pseg 0xFFFF: not started
    unsigned* inT;                                // bit per item code (P slot d: d, Q slot k: K + k): item is in T_mid
    unsigned* best;                               // [S]
    unsigned short *comp, *hook, *labL, *labR;    // [S] each; comp|hook double as the 2S renumbering flags
    int S_cap;
};

__host__ __device__ constexpr size_t dc_bytes(int K, int S_cap) {
    return al16((size_t)K * 8) + 5 * al16((size_t)K * 2) + al16((size_t)((2 * K + 31) / 32) * 4) + al16((size_t)S_cap * 4) +
           4 * al16((size_t)S_cap * 2);
}

__device__ __forceinline__ DcMem dc_carve(unsigned char* base, int K, int S_cap) {
    DcMem D;
    size_t o = 0;
    D.Qab = (unsigned*)(base + o);  o += al16((size_t)K * 4);
    D.Pab = (unsigned*)(base + o);  o += al16((size_t)K * 4);
    D.Qseg = (unsigned short*)(base + o); o += al16((size_t)K * 2);
    D.Qw = (unsigned short*)(base + o);   o += al16((size_t)K * 2);
    D.Pseg = (unsigned short*)(base + o); o += al16((size_t)K * 2);
    D.Pw = (unsigned short*)(base + o);   o += al16((size_t)K * 2);
    D.Phin = (unsigned short*)(base + o); o += al16((size_t)K * 2);
    D.inT = (unsigned*)(base + o);  o += al16((size_t)((2 * K + 31) / 32) * 4);
    D.best = (unsigned*)(base + o); o += al16((size_t)S_cap * 4);
    D.comp = (unsigned short*)(base + o); o += al16((size_t)S_cap * 2);
    D.hook = (unsigned short*)(base + o); o += al16((size_t)S_cap * 2);     // (directly behind comp: together the 2S flags)
    D.labL = (unsigned short*)(base + o); o += al16((size_t)S_cap * 2);
    D.labR = (unsigned short*)(base + o); o += al16((size_t)S_cap * 2);
    D.S_cap = S_cap;
    return D;
}

// exclusive prefix sum over a[0..N) in place (values are small counts); returns the total.  All W threads.
template <int W>
__device__ __forceinline__ int block_exscan_u16(unsigned short* a, int N, int* wcnt) {
    const int tid = threadIdx.x;
    const int per = (N + W - 1) / W;
    const int lo = tid * per, hi = lo + per < N ? lo + per : N;
    int s = 0;
    for (int i = lo; i < hi; ++i) s += a[i];
    const int inc = tlc_wave_iscan_i32(s);
    if (W > 64) {
        if (tlc_lane() == 63) wcnt[tid >> 6] = inc;
        __syncthreads();
    }
    int before = inc - s, total;
    if (W > 64) {
        int t = 0;
#pragma unroll
        for (int k = 0; k < W / 64; ++k) {
            const int c = wcnt[k];
            if (k < (tid >> 6)) before += c;
            t += c;
        }
        total = t;
    } else {
        total = __builtin_amdgcn_readlane(inc, 63);
    }
    for (int i = lo; i < hi; ++i) {
        const int v = a[i];
        a[i] = (unsigned short)before;
        before += v;
    }
    __syncthreads();
    return total;
}

// Components of the forest given by the marked items, as labels = the smallest id of the component.  `want(code, ab)`
// says whether an item is an edge of the forest.  Roots hook under ANY smaller neighbouring root (plain 16-bit stores: the
// race only decides which smaller root wins), then everything is flattened; a root that survives a round is a local
// minimum among the roots, so the number of roots at least halves per round.
template <int W, class Items>
__device__ __forceinline__ void dc_components(unsigned short* lab, int S, const Items& items, int* ctl) {
    const int tid = threadIdx.x;
    for (int x = tid; x < S; x += W) lab[x] = (unsigned short)x;
    __syncthreads();
    for (int round = 0; round < 64; ++round) {
        bool ch = false;
        items([&](unsigned ab) {
            const unsigned ra = lab[ab >> 16], rb = lab[ab & 0xffffu];
            if (ra < rb) { lab[rb] = (unsigned short)ra; ch = true; }
            else if (rb < ra) { lab[ra] = (unsigned short)rb; ch = true; }
        });
        __syncthreads();
        if (!block_any<W>(ch, ctl, 0)) break;
        flatten<W>(lab, S, ctl);
    }
}

// The divide-and-conquer cycle swap on the subgraph in M (rank space): Pos list M.pn[0..K), Neg list from the back, ends in
// M.dir, ascending ranks in M.arank, `finb` = bit per edge id: the edge is in the ascending pass's spanning tree (= alive at
// the end).  `out_hin[k]` (K u16, caller's) receives the higher endpoint of the edge query k removes.  Returns false if it
// does not apply (scratch too small) or the ranks turned out not to be an MST order; nothing the serial walk needs has been
// touched then.
template <int W, typename idx_t>
__device__ __noinline__ bool ext1_dc_solve(Mem<idx_t>& M, int n, int MMcap, const unsigned* finb, unsigned char* scratch, size_t scratch_bytes,
                              unsigned short* out_hin) {
    const int tid = threadIdx.x;
    const int K = M.ctl[3], nneg = M.ctl[4];
    const int S_cap = (n > 2 * K + 2 ? n : 2 * K + 2);
    if (K < 2 || K > 16000 || S_cap > 65000 || dc_bytes(K, S_cap) > scratch_bytes) return false;
    const DcMem D = dc_carve(scratch, K, S_cap);
    const unsigned* ends = M.dir;
    int* ctl = M.ctl;
    int* n_pslot = &ctl[10];                      // P slots handed out so far
    int* bad = &ctl[11];
    // ---- root segment [0, K): supernodes = components of the Neg edges that are never removed -----------------------------
    if (tid == 0) { *n_pslot = 0; *bad = 0; }
    dc_components<W>(D.labL, n, [&](auto edge) {
        for (int k = tid; k < nneg; k += W) {
            const unsigned e = M.pn[MMcap - 1 - k];
            if ((finb[e >> 5] >> (e & 31)) & 1u) edge(ends[e]);
        }
    }, ctl);
    for (int d = tid; d < K; d += W) D.Pseg[d] = 0xFFFFu;
    __syncthreads();
    for (int k = tid; k < nneg; k += W) {         // Neg edges that die: P copies of the root segment
        const unsigned e = M.pn[MMcap - 1 - k];
        if (!((finb[e >> 5] >> (e & 31)) & 1u)) {
            const int d = atomicAdd(n_pslot, 1);
            if (d < K) {
                const unsigned ab = ends[e];
                D.Pab[d] = ((unsigned)D.labL[ab >> 16] << 16) | D.labL[ab & 0xffffu];
                D.Pw[d] = (unsigned short)M.arank[e];
                D.Phin[d] = (unsigned short)(ab & 0xffffu);
                D.Pseg[d] = 0;
            }
        }
    }
    for (int k = tid; k < K; k += W) {
        const unsigned e = M.pn[k];
        const unsigned ab = ends[e];
        D.Qab[k] = ((unsigned)D.labL[ab >> 16] << 16) | D.labL[ab & 0xffffu];
        D.Qw[k] = (unsigned short)M.arank[e];
        D.Qseg[k] = (unsigned short)(((finb[e >> 5] >> (e & 31)) & 1u) ? 0x8000u : 0u);
    }
    __syncthreads();
    // (#dying Neg + #dying Pos == K exactly when the ranks are an MST order; checked at the end through the bijection)
    int S = n;
    int levels = 0;
    while ((1 << levels) < K) ++levels;
    for (int L = 0; L < levels; ++L) {
        const int n_p = *n_pslot < K ? *n_pslot : K;
        auto mid_of = [&](int j) { return (int)(((long long)(2 * j + 1) * K) >> (L + 1)); };
        // ---- A. T_mid = MSF(P + Q[l, mid)) for all segments at once (Boruvka on the ranks) -------------------------------
        for (int x = tid; x < S; x += W) D.comp[x] = (unsigned short)x;
        for (int w = tid; w < (2 * K + 31) / 32; w += W) D.inT[w] = 0u;
        __syncthreads();
        for (int round = 0; round < 48; ++round) {
            for (int x = tid; x < S; x += W) { D.best[x] = 0xFFFFFFFFu; D.hook[x] = 0xFFFFu; }
            __syncthreads();
            bool found = false;
            for (int d = tid; d < n_p; d += W) {
                if (D.Pseg[d] == 0xFFFFu) continue;
                const unsigned ab = D.Pab[d];
                const unsigned ra = D.comp[ab >> 16], rb = D.comp[ab & 0xffffu];
                if (ra != rb) {
                    const unsigned key = ((unsigned)D.Pw[d] << 16) | (unsigned)d;
                    atomicMin(&D.best[ra], key); atomicMin(&D.best[rb], key);
                    found = true;
                }
            }
            for (int k = tid; k < K; k += W) {
                if (k >= mid_of(D.Qseg[k] & 0x7FFF)) continue;
                const unsigned ab = D.Qab[k];
                const unsigned ra = D.comp[ab >> 16], rb = D.comp[ab & 0xffffu];
                if (ra != rb) {
                    const unsigned key = ((unsigned)D.Qw[k] << 16) | (unsigned)(K + k);
                    atomicMin(&D.best[ra], key); atomicMin(&D.best[rb], key);
                    found = true;
                }
            }
            __syncthreads();
            if (!block_any<W>(found, ctl, 0)) break;
            for (int x = tid; x < S; x += W) {
                const unsigned key = D.best[x];
                if (key == 0xFFFFFFFFu || D.comp[x] != x) continue;
                const int code = (int)(key & 0xffffu);
                const unsigned ab = code < K ? D.Pab[code] : D.Qab[code - K];
                const unsigned ra = D.comp[ab >> 16], rb = D.comp[ab & 0xffffu];
                const unsigned other = (ra == (unsigned)x) ? rb : ra;
                atomicOr(&D.inT[code >> 5], 1u << (code & 31));
                // unique ranks => the pick graph has only 2-cycles; the larger root of a mutual pick hooks
                if (D.best[other] != key || (unsigned)x > other) D.hook[x] = (unsigned short)other;
            }
            __syncthreads();
            for (int x = tid; x < S; x += W)
                if (D.hook[x] != 0xFFFFu) D.comp[x] = D.hook[x];
            __syncthreads();
            flatten<W>(D.comp, S, ctl);
        }
        auto in_t = [&](int code) { return ((D.inT[code >> 5] >> (code & 31)) & 1u) != 0u; };
        // ---- B. what each half contracts: left = P edges alive at mid, right = Q[l, mid) edges alive at mid and at r -----
        dc_components<W>(D.labL, S, [&](auto edge) {
            for (int d = tid; d < n_p; d += W)
                if (D.Pseg[d] != 0xFFFFu && in_t(d)) edge(D.Pab[d]);
        }, ctl);
        dc_components<W>(D.labR, S, [&](auto edge) {
            for (int k = tid; k < K; k += W) {
                const unsigned sg = D.Qseg[k];
                if (k < mid_of(sg & 0x7FFF) && (sg & 0x8000u) && in_t(K + k)) edge(D.Qab[k]);
            }
        }, ctl);
        // ---- C. the supernodes the children use: left copies keep [0, S), right copies move to [S, 2S) ---------------------
        unsigned short* flag = D.comp;                                    // comp | hook = 2S flags, then the new ids
        for (int x = tid; x < 2 * S; x += W) flag[x] = 0;
        __syncthreads();
        auto child_ab = [&](unsigned ab, bool right) -> unsigned {
            const unsigned a = ab >> 16, b = ab & 0xffffu;
            return right ? (((unsigned)(S + D.labR[a]) << 16) | (unsigned)(S + D.labR[b]))
                         : (((unsigned)D.labL[a] << 16) | (unsigned)D.labL[b]);
        };
        for (int d = tid; d < n_p; d += W) {
            if (D.Pseg[d] == 0xFFFFu) continue;
            const unsigned c = child_ab(D.Pab[d], in_t(d));
            flag[c >> 16] = 1; flag[c & 0xffffu] = 1;
        }
        for (int k = tid; k < K; k += W) {
            const bool left = k < mid_of(D.Qseg[k] & 0x7FFF);
            const unsigned c = child_ab(D.Qab[k], !left);
            flag[c >> 16] = 1; flag[c & 0xffffu] = 1;
            if (left && in_t(K + k) && !(D.Qseg[k] & 0x8000u)) {            // its P copy starts in the right half
                const unsigned c2 = child_ab(D.Qab[k], true);
                flag[c2 >> 16] = 1; flag[c2 & 0xffffu] = 1;
            }
        }
        __syncthreads();
        const int S_next = block_exscan_u16<W>(flag, 2 * S, M.wcnt);
        if (S_next > S_cap) { if (tid == 0) *bad = 1; __syncthreads(); break; }
        // ---- D. move every copy to its child -----------------------------------------------------------------------------------
        auto renum = [&](unsigned c) -> unsigned { return ((unsigned)flag[c >> 16] << 16) | (unsigned)flag[c & 0xffffu]; };
        for (int d = tid; d < n_p; d += W) {
            const unsigned sg = D.Pseg[d];
            if (sg == 0xFFFFu) continue;
            const bool right = in_t(d);
            D.Pab[d] = renum(child_ab(D.Pab[d], right));
            D.Pseg[d] = (unsigned short)(2 * sg + (right ? 1 : 0));
        }
        for (int k = tid; k < K; k += W) {
            const unsigned sg = D.Qseg[k];
            const int j = (int)(sg & 0x7FFF);
            const unsigned ab = D.Qab[k];
            if (k < mid_of(j)) {
                const bool alive_mid = in_t(K + k);
                if (alive_mid && !(sg & 0x8000u)) {
                    const int d = atomicAdd(n_pslot, 1);
                    if (d < K) {
                        D.Pab[d] = renum(child_ab(ab, true));
                        D.Pw[d] = D.Qw[k];
                        D.Phin[d] = (unsigned short)(ends[M.pn[k]] & 0xffffu);
                        D.Pseg[d] = (unsigned short)(2 * j + 1);
                    } else *bad = 1;
                }
                D.Qab[k] = renum(child_ab(ab, false));
                D.Qseg[k] = (unsigned short)((2 * j) | (alive_mid ? 0x8000u : 0u));
            } else {
                D.Qab[k] = renum(child_ab(ab, true));
                D.Qseg[k] = (unsigned short)((2 * j + 1) | (sg & 0x8000u));
            }
        }
        __syncthreads();
        S = S_next;
        if (*bad) break;
    }
    // ---- every segment is one query now: the P copy in it is the edge that query removes ----------------------------------------
    bool fail = (*bad != 0) || (*n_pslot != K);
    __syncthreads();
    if (!fail) {
        for (int w = tid; w < (K + 31) / 32; w += W) D.inT[w] = 0u;
        __syncthreads();
        for (int d = tid; d < K; d += W) {
            const unsigned sg = D.Pseg[d];
            if (sg == 0xFFFFu) { fail = true; continue; }
            const int k = (int)(((long long)sg * K) >> levels);
            const int k1 = (int)(((long long)(sg + 1) * K) >> levels);
            if (k1 - k != 1 || k >= K) { fail = true; continue; }
            const unsigned old = atomicOr(&D.inT[k >> 5], 1u << (k & 31));
            if ((old >> (k & 31)) & 1u) { fail = true; continue; }
            if (D.Pw[d] <= D.Qw[k]) { fail = true; continue; }          // the removed edge must be heavier than the inserted one
            out_hin[k] = D.Phin[d];
        }
        __syncthreads();
    }
    return !block_any<W>(fail, ctl, 0);
}

// Descending sort, equal keys: the edge with the HIGHER ascending rank first (module comment).  valS/keyS hold the sorted
// order; runs of equal keys are short (two edges from the two roots to a common neighbour are the usual case).  Returns false
// if some run is longer than TLC_DC_MAX_TIE_RUN (nothing is changed then).
template <int W, typename idx_t>
__device__ __forceinline__ bool fix_desc_ties(Mem<idx_t>& M, int m) {
    const int tid = threadIdx.x;
    constexpr int PER = 8;                                   // positions per thread (callers guarantee m <= PER * W)
    int np_[PER];
    unsigned ne_[PER];
    bool too_long = false;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int pos = tid + q * W;
        np_[q] = -1;
        if (pos < m) {
            const ull k = M.keyS[pos];
            const bool tl = pos > 0 && M.keyS[pos - 1] == k, tr = pos + 1 < m && M.keyS[pos + 1] == k;
            if (tl || tr) {
                int s = pos, t = pos + 1;
                while (s > 0 && M.keyS[s - 1] == k && pos - s <= TLC_DC_MAX_TIE_RUN) --s;
                while (t < m && M.keyS[t] == k && t - pos <= TLC_DC_MAX_TIE_RUN) ++t;
                if (t - s > TLC_DC_MAX_TIE_RUN) too_long = true;
                else {
                    const unsigned e = M.valS[pos];
                    const unsigned ar = M.arank[e];
                    int c = 0;
                    for (int j = s; j < t; ++j) c += (M.arank[M.valS[j]] > ar);
                    np_[q] = s + c;
                    ne_[q] = e;
                }
            }
        }
    }
    const bool bad = block_any<W>(too_long, M.ctl, 0);
    if (!bad) {
#pragma unroll
        for (int q = 0; q < PER; ++q)
            if (np_[q] >= 0) M.valS[np_[q]] = ne_[q];
    }
    __syncthreads();
    return !bad;
}

}  // namespace

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

#ifndef TLC_DC_MIN_POS             /* (overridable for the threshold sweep: tools/gpu_threshold_sweep.sh) */
#define TLC_DC_MIN_POS 160          /* below this many Pos edges the serial walk wins (a workgroup with a CU to itself) */
#endif
/* 256-thread tiers (MEDIUM): measured on the PubMed batch, a level costs ~30k cycles there as well (phases of two or three
 * dependent LDS round trips, no latency hiding) -- 8 levels = 100 us for 190 queries against 185 us for the serial walk in
 * tlc_pd_swap_kernel, and the kernel sits between the tier kernel and the serial kernel on the same stream: the chain gets
 * longer, not shorter.  Only the sizes the tier cannot reach in this batch take it. */
#ifndef TLC_DC_MIN_POS_SHARED
#define TLC_DC_MIN_POS_SHARED 320
#endif
#ifndef TLC_DC_LARGE_MODE
#define TLC_DC_LARGE_MODE 2          /* LARGE tier: 1 = in the tier kernel, 2 = hand the subgraph to tlc_pd_dc_kernel */
#endif
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

// "Did any thread raise the flag" with ONE barrier per call (block_any of pd_pipeline.hip takes three): three rotating LDS
// slots; call i uses slot i % 3 and clears the slot of call i - 1, which every thread has read before it reached this
// call's barrier and which is not written again before the barrier of call i + 1.  `turn` is uniform across the workgroup.
// The barrier also orders the LDS traffic of the phase that computed `v`.
template <int W>
struct AnyFlag {
    int* slots;
    int turn;
    __device__ __forceinline__ void init(int* three_ints) {
        slots = three_ints; turn = 0;
        if (threadIdx.x < 3) slots[threadIdx.x] = 0;
        __syncthreads();
    }
    __device__ __forceinline__ bool any(bool v) {
        if (W == 64) { __syncthreads(); return __ballot(v) != 0ull; }
        const int sl = turn % 3;
        if (v) slots[sl] = 1;
        __syncthreads();
        const bool r = slots[sl] != 0;
        if (threadIdx.x == 0) slots[(sl + 2) % 3] = 0;
        ++turn;
        return r;
    }
};

// Every entry to its root.  The arrays are flat before a hooking round, so an entry's depth afterwards is the depth of its old
// root in the hook forest -- a handful.  A sweep follows up to four pointers per entry (dependent LDS reads, but one barrier)
// and reports whether anybody is still short of a root; long hook chains shrink at least fourfold per sweep.
__device__ __forceinline__ bool dc_jump4(unsigned short* a, int y) {
    const unsigned c = a[y];
    const unsigned p1 = a[c];
    if (p1 == c) return false;
    const unsigned p2 = a[p1];
    const unsigned p3 = a[p2];
    a[y] = (unsigned short)p3;
    return a[p3] != p3;
}
template <int W>
__device__ __forceinline__ void dc_flatten(unsigned short* a, int S, AnyFlag<W>& af) {
    for (int it = 0; it < 32; ++it) {
        bool more = false;
        for (int y = threadIdx.x; y < S; y += W) more |= dc_jump4(a, y);
        if (!af.any(more)) break;
    }
}
template <int W>
__device__ __forceinline__ void dc_flatten2(unsigned short* a, unsigned short* b, int S, AnyFlag<W>& af) {
    for (int it = 0; it < 32; ++it) {
        bool more = false;
        for (int y = threadIdx.x; y < S; y += W) { more |= dc_jump4(a, y); more |= dc_jump4(b, y); }
        if (!af.any(more)) break;
    }
}

// Components of the forests given by two sets of marked items, as labels = the smallest id of the component, both label
// arrays in the same sweeps.  itemsA / itemsB call edge(ab) for every edge of their forest (itemsB may be empty).  Roots hook
// under ANY smaller neighbouring root (plain 16-bit stores: the race only decides which smaller root wins), then everything is
// flattened; a root that survives a round is a local minimum among the roots, so the number of roots at least halves per round.
template <int W, class ItemsA, class ItemsB>
__device__ __forceinline__ void dc_components2(unsigned short* labA, unsigned short* labB, int S, const ItemsA& itemsA,
                                               const ItemsB& itemsB, AnyFlag<W>& af) {
    const int tid = threadIdx.x;
    for (int x = tid; x < S; x += W) { labA[x] = (unsigned short)x; labB[x] = (unsigned short)x; }
    __syncthreads();
    for (int round = 0; round < 64; ++round) {
        bool ch = false;
        itemsA([&](unsigned ab) {
            const unsigned ra = labA[ab >> 16], rb = labA[ab & 0xffffu];
            if (ra < rb) { labA[rb] = (unsigned short)ra; ch = true; }
            else if (rb < ra) { labA[ra] = (unsigned short)rb; ch = true; }
        });
        itemsB([&](unsigned ab) {
            const unsigned ra = labB[ab >> 16], rb = labB[ab & 0xffffu];
            if (ra < rb) { labB[rb] = (unsigned short)ra; ch = true; }
            else if (rb < ra) { labB[ra] = (unsigned short)rb; ch = true; }
        });
        if (!af.any(ch)) break;
        dc_flatten2<W>(labA, labB, S, af);
    }
}

// Where the edges come from.  neg(i, ab, w, fin) -> false if item i is not a Neg edge; pos(k, ...): query k.  ab = endpoints
// (rank space, any orientation), w = ascending rank, fin = the edge is in the ascending pass's spanning tree (alive at the end).
// LdsSrc: the lists a tier kernel holds in LDS.  `finb` is the ascending pass's tree bitmap, indexed by ascending rank.
struct LdsSrc {
    const unsigned *pn, *ends, *arank, *finb;
    int MMcap;
    __device__ __forceinline__ bool neg(int i, unsigned& ab, unsigned& w, bool& fin) const {
        const unsigned e = pn[MMcap - 1 - i];
        ab = ends[e]; w = arank[e]; fin = ((finb[w >> 5] >> (w & 31)) & 1u) != 0u;
        return true;
    }
    __device__ __forceinline__ void pos(int k, unsigned& ab, unsigned& w, bool& fin) const {
        const unsigned e = pn[k];
        ab = ends[e]; w = arank[e]; fin = ((finb[w >> 5] >> (w & 31)) & 1u) != 0u;
    }
};
// HandoffSrc: the record a tier kernel left in HBM (pd_pipeline.hip, Handoff): the oriented tree -- node i hangs under
// par[i] & 0x7fffffff behind the edge of key[i] = (rank + 1) << 8, bit 31 of par = fin; the root has key 0 -- and the queries
// ((rank + 1) << 8 | fin) << 32 | p << 16 | q.
struct HandoffSrc {
    const unsigned *par, *key;
    const ull* query;
    __device__ __forceinline__ bool neg(int i, unsigned& ab, unsigned& w, bool& fin) const {
        const unsigned k = key[i];
        if (k == 0u) return false;
        const unsigned pr = par[i];
        ab = ((unsigned)i << 16) | (pr & 0xffffu); w = (k >> 8) - 1u; fin = (pr >> 31) != 0u;
        return true;
    }
    __device__ __forceinline__ void pos(int k, unsigned& ab, unsigned& w, bool& fin) const {
        const ull q = query[k];
        ab = (unsigned)q; w = ((unsigned)(q >> 40)) - 1u; fin = ((q >> 32) & 1ull) != 0ull;
    }
};

// The divide-and-conquer cycle swap: K queries, n_neg_items candidates for Neg edges (Src::neg), n nodes (rank space).
// `out_hin[k]` (K u16, caller's) receives the higher endpoint of the edge query k removes.  ctl: 16 ints, wcnt: 32 ints of
// LDS.  Returns false if it does not apply (scratch too small) or the ranks turned out not to be an MST order; nothing of the
// caller's has been touched then.
template <int W, class Src>
__device__ __forceinline__ bool ext1_dc_solve(const Src& src, int n, int K, int n_neg_items, int* ctl, int* wcnt,
                                              unsigned char* scratch, size_t scratch_bytes, unsigned short* out_hin,
                                              unsigned long long* prof = nullptr) {
    const int tid = threadIdx.x;
    // prof (diagnostics, thread 0): [0] setup [1] MSF [2] contraction labels [3] renumber + move, cycles; [4] Boruvka rounds [5] levels
    unsigned long long t_prev = prof ? clock64() : 0ull;
#define DC_STAMP(k) do { if (prof && tid == 0) { const unsigned long long _t = clock64(); prof[(k)] += _t - t_prev; t_prev = _t; } } while (0)
    const int S_cap = (n > 2 * K + 2 ? n : 2 * K + 2);
    if (K < 2 || K > 16000 || S_cap > 65000 || dc_bytes(K, S_cap) > scratch_bytes) return false;
    const DcMem D = dc_carve(scratch, K, S_cap);
    int* n_pslot = &ctl[10];                      // P slots handed out so far
    int* bad = &ctl[11];
    // ---- root segment [0, K): supernodes = components of the Neg edges that are never removed -----------------------------
    if (tid == 0) { *n_pslot = 0; *bad = 0; }
    AnyFlag<W> af;
    af.init(&ctl[12]);
    dc_components2<W>(D.labL, D.labR, n, [&](auto edge) {
        for (int i = tid; i < n_neg_items; i += W) {
            unsigned ab, w; bool fin;
            if (src.neg(i, ab, w, fin) && fin) edge(ab);
        }
    }, [&](auto) {}, af);
    for (int d = tid; d < K; d += W) D.Pseg[d] = 0xFFFFu;
    __syncthreads();
    for (int i = tid; i < n_neg_items; i += W) {  // Neg edges that die: P copies of the root segment
        unsigned ab, w; bool fin;
        if (src.neg(i, ab, w, fin) && !fin) {
            const int d = atomicAdd(n_pslot, 1);
            if (d < K) {
                const unsigned a = ab >> 16, b = ab & 0xffffu;
                D.Pab[d] = ((unsigned)D.labL[a] << 16) | D.labL[b];
                D.Pw[d] = (unsigned short)w;
                D.Phin[d] = (unsigned short)(a > b ? a : b);
                D.Pseg[d] = 0;
            }
        }
    }
    for (int k = tid; k < K; k += W) {
        unsigned ab, w; bool fin;
        src.pos(k, ab, w, fin);
        D.Qab[k] = ((unsigned)D.labL[ab >> 16] << 16) | D.labL[ab & 0xffffu];
        D.Qw[k] = (unsigned short)w;
        D.Qseg[k] = (unsigned short)(fin ? 0x8000u : 0u);
    }
    __syncthreads();
    // (#dying Neg + #dying Pos == K exactly when the ranks are an MST order; checked at the end through the bijection)
    DC_STAMP(0);
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
            if (!af.any(found)) break;
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
            dc_flatten<W>(D.comp, S, af);
            if (prof && tid == 0) prof[4] += 1;
        }
        DC_STAMP(1);
        auto in_t = [&](int code) { return ((D.inT[code >> 5] >> (code & 31)) & 1u) != 0u; };
        // ---- B. what each half contracts: left = P edges alive at mid, right = Q[l, mid) edges alive at mid and at r -----
        dc_components2<W>(D.labL, D.labR, S, [&](auto edge) {
            for (int d = tid; d < n_p; d += W)
                if (D.Pseg[d] != 0xFFFFu && in_t(d)) edge(D.Pab[d]);
        }, [&](auto edge) {
            for (int k = tid; k < K; k += W) {
                const unsigned sg = D.Qseg[k];
                if (k < mid_of(sg & 0x7FFF) && (sg & 0x8000u) && in_t(K + k)) edge(D.Qab[k]);
            }
        }, af);
        DC_STAMP(2);
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
        const int S_next = block_exscan_u16<W>(flag, 2 * S, wcnt);
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
                        { unsigned qab, qw; bool qf; src.pos(k, qab, qw, qf);
                          D.Phin[d] = (unsigned short)((qab >> 16) > (qab & 0xffffu) ? (qab >> 16) : (qab & 0xffffu)); }
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
        DC_STAMP(3);
        if (prof && tid == 0) prof[5] += 1;
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
#undef DC_STAMP
    return !af.any(fail);
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

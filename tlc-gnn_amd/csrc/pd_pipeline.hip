// pd_pipeline.hip -- P5..P9 of the hot path on one vicinity subgraph per workgroup.
//
//   P5  filtration.build_fv           sg2dgm/riccidist2dgm.py:20-61     (node-sourced shortest paths, normalise)
//   P6  perturb_filter_function       sg2dgm/accelerated_PD.py:6-23     (fp64 keys, no FMA contraction)
//   P7  Union_find                    sg2dgm/accelerated_PD.py:26-113   (two sorted passes, elder rule)
//   P8  Accelerate_PD                 sg2dgm/accelerated_PD.py:115-178  (spanning-tree cycle swap)
//   P9  PersistenceImager.transform   sg2dgm/PersistenceImager.pyx:352-388
//
// All per-subgraph state (distances, keys, union-find parents, tree parents, diagram points, image
// table) is staged in LDS; the size tier picks the workgroup width and the LDS footprint (tlc_kernels.h).
// The HUGE tier runs the same code on a per-workgroup HBM scratch slot.
//
// Exactness of the filtration: the reference's distance from x to a root r is the minimum
// over paths of the LEFT-TO-RIGHT fp64 sum starting at x.  We run one root-sourced Bellman-Ford per root
// (lanes over directed entries, ds_min_u64 on the bit patterns), mark the entries that are tight within
// 1e-10 relative, and let one lane per node x walk its tight chain towards the root accumulating the
// weights in the reference's order.  If every node on the chain has exactly one tight successor, every
// other path is longer by far more than any rounding error, so the chain's sum IS the reference value.
// Otherwise (exact or near ties) that source falls back to a Bellman-Ford sourced at x itself, which
// converges to the same minimum by monotonicity of fp64 addition.  Compile with -ffp-contract=off.
#include "tlc_common.h"
#include "tlc_kernels.h"

#define TLC_INF_BITS 0x7FF0000000000000ull
// diagnostics (make PHASE_DEBUG=1; tools/phase_profile.py): accumulate the cycles thread 0 spent since the previous stamp
// into phase slot k.  Compiled out by default: the counters cost registers in every tier kernel.
#ifdef TLC_PHASE_DEBUG
#define TLC_STAMP(k)                                                                 \
    do {                                                                             \
        if (pc && threadIdx.x == 0) {                                                \
            const unsigned long long _t = clock64();                                 \
            ph[(k)] += _t - t_prev;             /* (summed into pc[] once, at the end) */ \
            t_prev = _t;                                                             \
        }                                                                            \
    } while (0)
#else
#define TLC_STAMP(k) do { } while (0)
#endif
#define TLC_NONE16 0xFFFFu
// Timing experiment (tools/gpu_r6_knockout.sh): -DTLC_STOP_AFTER=k compiles the tier kernels' stages behind stage k out -- 1 staging,
// 2 Bellman-Ford, 3 tight chains (+ fallback) and normalisation, 4 edge compaction, 5 rank relabel + ascending sort, 6 ascending pass,
// 7 descending sort, 8 descending pass + Pos / Neg split, 9 cycle swap (hand-off / walk / divide and conquer), 10 image = everything.
// Rows are garbage below 10; what a pipelined batch then costs, stage by stage, is the cost table of DESIGN.md section 0.
#ifndef TLC_STOP_AFTER
#define TLC_STOP_AFTER 10
#endif
#define TLC_STOPPED 99          /* a status nobody tests for: every later stage is guarded by status == TLC_ST_OK */
// the SMALL tier keeps its entry weights in LDS (true) or reads them from the arena like the larger tiers (false: 6.9 -> 6.0 KB per
// workgroup; A/B on one box, tools/gpu_build_ab.sh: pipelined batch 0.744 -> 0.736 ms, one batch alone 0.804 -> 0.792 ms)
#ifndef TLC_SMALL_LWL
#define TLC_SMALL_LWL false
#endif

namespace {

typedef unsigned long long ull;

__host__ __device__ constexpr size_t al16(size_t x) { return (x + 15) & ~(size_t)15; }
__host__ __device__ constexpr int pow2ceil(int x) {
    int p = 1;
    while (p < x) p <<= 1;
    return p;
}
__host__ __device__ constexpr size_t smax(size_t a, size_t b) { return a > b ? a : b; }

// Region layout of one subgraph's state; constexpr for the LDS tiers, evaluated at run time for HUGE.
// Regions are aliased by phase:
//   x region : [dist from v | tight-successor tables]  (filtration)
//              [sort keys u64 | sort payload u32]       (the three sorts)
//              [first/hook u32 | best u32 | comp2 | tree bits | R bits] in the key half between the sorts (MST passes),
//              then [edge rank of each node's parent edge u32] (cycle swap)
//   dir region: directed entries, then [undirected edges (lo<<16|hi in rank space) | ascending rank of every edge],
//              then (with lw) the image table
struct Layout {
    size_t o_f, o_dir, o_lw, o_x, o_vals, o_amb, o_par, o_mark, o_pn, o_pts, o_ctl, o_rec, total;
    int P;   // capacity of the sort buffers (power of two >= max(NM, MM))
};
__host__ __device__ constexpr size_t aux_bytes(int NM, int MM, int idxb) {
    return (size_t)NM * 8 + al16((size_t)NM * idxb) + 2 * al16((size_t)((MM + 31) / 32) * 4);
}
__host__ __device__ constexpr Layout make_layout(int NM, int MM, bool lwl, int idxb, size_t min_table = 0) {
    Layout L{};
    size_t o = 0;
    L.P = pow2ceil(NM > MM ? NM : MM);
    L.o_f = o;    o += al16((size_t)(NM + 2) * 8);                      // BF distances from u, then f
    L.o_dir = o;  o += al16((size_t)2 * MM * 4);                        // directed entries | later: edges + asc ranks
    L.o_lw = o;   o += lwl ? al16((size_t)2 * MM * 8) : 0;              // entry weights (dir|lw also host the PI table)
    if (o - L.o_dir < min_table) o = L.o_dir + al16(min_table);         // the image table needs >= 1 point
    const size_t keys = al16(smax((size_t)L.P * 8, aux_bytes(NM, MM, idxb)));
    L.o_x = o;
    L.o_vals = o + keys;
    o += al16(smax((size_t)NM * 8 + (size_t)4 * NM * 4, keys + (size_t)L.P * 4));
    // the cycle swap re-uses everything from o_x up to o_pn: par, key, mark (u32) and two u64 tables per node
    {
        const size_t have = (o - L.o_x) + 3 * al16((size_t)NM * idxb);
        const size_t need = (size_t)(NM + 1) * 28 + 16;     // one spare node slot for the idle walker
        if (have < need) o += al16(need - have);
    }
    L.o_amb = o;  o += al16((size_t)NM * idxb);                         // tie-fallback list, then union-find parents
    L.o_par = o;  o += al16((size_t)NM * idxb);                         // spanning-tree parents
    L.o_mark = o; o += al16((size_t)NM * idxb);                         // node ranks, then path stamps
    // (both also host one tight-successor weight table of the filtration stage, NM doubles, when the weights are not in LDS)
    L.o_pn = o;   o += al16(smax((size_t)MM * 4, lwl ? 0 : (size_t)NM * 8));             // Pos from the front, Neg from the back (edge ids)
    L.o_pts = o;  o += al16(smax((size_t)(MM + 2) * 4, lwl ? 0 : (size_t)NM * 8));       // diagram points (birth node<<16 | death node)
    L.o_ctl = o;  o += 320;                                             // 16 ints | 16 doubles | 32 ints
    // cycle swap: the two walks' path records, [2][65] nodes + [2][65] keys (the entry weights are dead by then)
    if (lwl && (size_t)2 * MM * 8 >= 1280) L.o_rec = L.o_lw;
    else { L.o_rec = o; o += 1280; }
    L.total = o;
    return L;
}

template <typename idx_t>
struct Mem {
    double* f;
    unsigned* dir;     // phase 1: directed entries; afterwards ends[e] = lo<<16|hi (rank space)
    unsigned* arank;   // ascending-sort position of edge e
    double* lw;
    ull* dv;
    unsigned *cntU, *nxtU, *cntV, *nxtV;
    ull* keyS;
    unsigned* valS;
    // MST-pass scratch inside the key half of the sort buffers
    unsigned *first, *best;
    idx_t* comp2;
    unsigned *tbits, *rbits;
    unsigned* ekey;    // cycle swap: ascending rank of the edge (node, parent[node])
    idx_t *amb, *comp, *par, *mark;
    unsigned *pn, *pts;
    int* ctl;      // [0] flag [1] namb [2] npts [3] npos [4] nneg [5] flag2 [6] n_up [7] n_down [8] n_one
    double* red;   // 16 doubles for block reductions
    int* wcnt;     // 32 ints for block compaction
    unsigned* rec; // cycle swap: path records of the two walks
    unsigned char* table;  // PI table: spans dir (+lw)
    size_t table_bytes;
    unsigned char* xbase;  // everything between the edge tables and the Pos/Neg lists: free once the passes are done
    size_t xbytes;
    unsigned long long* stats;   // batch statistics (device, may be null): [1] subgraphs whose cycle swap ran as a divide and conquer, [3] fell back
    // HUGE tier (state in an HBM scratch slot): LDS for the tables of the serial cycle swap when they fit (else null / 0)
    unsigned char* lds_swap;
    size_t lds_swap_bytes;
    int P;
};

template <typename idx_t>
__device__ __forceinline__ Mem<idx_t> carve(unsigned char* base, const Layout& L, int NM, int MM, bool lwl) {
    Mem<idx_t> m;
    m.f = (double*)(base + L.o_f);
    m.dir = (unsigned*)(base + L.o_dir);
    m.arank = m.dir + MM;
    m.lw = (double*)(base + L.o_lw);
    m.dv = (ull*)(base + L.o_x);
    m.cntU = (unsigned*)(base + L.o_x + (size_t)NM * 8);
    m.nxtU = m.cntU + NM;
    m.cntV = m.nxtU + NM;
    m.nxtV = m.cntV + NM;
    m.keyS = (ull*)(base + L.o_x);
    m.valS = (unsigned*)(base + L.o_vals);
    m.first = (unsigned*)(base + L.o_x);
    m.best = m.first + NM;
    m.comp2 = (idx_t*)(base + L.o_x + (size_t)NM * 8);
    m.tbits = (unsigned*)(base + L.o_x + (size_t)NM * 8 + al16((size_t)NM * sizeof(idx_t)));
    m.rbits = m.tbits + al16((size_t)((MM + 31) / 32) * 4) / 4;
    m.ekey = m.first;
    m.amb = (idx_t*)(base + L.o_amb);
    m.comp = m.amb;
    m.par = (idx_t*)(base + L.o_par);
    m.mark = (idx_t*)(base + L.o_mark);
    m.pn = (unsigned*)(base + L.o_pn);
    m.pts = (unsigned*)(base + L.o_pts);
    m.ctl = (int*)(base + L.o_ctl);
    m.red = (double*)(base + L.o_ctl + 64);
    m.wcnt = (int*)(base + L.o_ctl + 192);
    m.rec = (unsigned*)(base + L.o_rec);
    m.table = base + L.o_dir;
    // (round 6, measured: with the table up to the diagram points -- 33 instead of 9 points per round of the image stage in the SMALL tier --
    // the rows are the same bits and the batch takes the same time; the whole in-kernel image stage is 1.8 % of a pipelined batch
    // (-DTLC_STOP_AFTER=9).  The per-phase cycle stamps of the PHASE_DEBUG build had said 39 % of the SMALL kernel: not to be trusted.)
    m.table_bytes = L.o_x - L.o_dir;
    m.xbase = base + L.o_x;
    m.xbytes = L.o_pn - L.o_x;
    m.stats = nullptr;
    m.lds_swap = nullptr; m.lds_swap_bytes = 0;
    m.P = L.P;
    return m;
}

// ---- block helpers (W threads) -------------------------------------------------------------------------------
template <int W>
__device__ __forceinline__ double block_max(double v, double* red) {
    v = tlc_wave_max_f64(v);
    if (W == 64) return v;
    if (tlc_lane() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int k = 1; k < W / 64; ++k) r = red[k] > r ? red[k] : r;
    __syncthreads();
    return r;
}
template <int W>
__device__ __forceinline__ double block_min(double v, double* red) {
    v = tlc_wave_min_f64(v);
    if (W == 64) return v;
    if (tlc_lane() == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double r = red[0];
#pragma unroll
    for (int k = 1; k < W / 64; ++k) r = red[k] < r ? red[k] : r;
    __syncthreads();
    return r;
}

// monotone map double -> u64 (ascending); ~key gives descending order
__device__ __forceinline__ ull f64_key(double x) {
    const ull b = (ull)__double_as_longlong(x);
    return b ^ ((b >> 63) ? ~0ull : (1ull << 63));
}

// perturb_filter_function (accelerated_PD.py:18-21); evaluated in exactly this association, no FMA
__device__ __forceinline__ double key_asc(double fa, double fb) {
    const double hi = fa > fb ? fa : fb, lo = fa < fb ? fa : fb;
    return hi + (lo + 1.0) * 1e-6;
}
__device__ __forceinline__ double key_desc(double fa, double fb) {
    const double hi = fa > fb ? fa : fb, lo = fa < fb ? fa : fb;
    return lo - (101.0 - hi) * 1e-6;
}

// ---- Bellman-Ford over the directed entries: dist = min over paths of the fp64 sum accumulated from the source --
// CE > 0: a thread's (at most CE) entries and their weights stay in registers over the rounds.  The tiers that leave the
// weights in HBM/L2 would otherwise pay a global round trip per round -- several microseconds each while the vicinity
// kernels of the batch saturate the memory pipeline beside an early-started tier kernel.
template <int W, bool TWO, int CE, class LWF>
__device__ __forceinline__ void bellman_ford(ull* d0, ull* d1, const unsigned* dir, int m2, int n, LWF LW, int* ctl) {
    const int tid = threadIdx.x;
    if constexpr (CE > 0) {
        unsigned ec[CE];
        double wc[CE];
#pragma unroll
        for (int c = 0; c < CE; ++c) {
            const int j = tid + c * W;
            ec[c] = j < m2 ? dir[j] : 0u;
            wc[c] = j < m2 ? LW(j) : 0.0;
        }
        for (int iter = 0; iter <= n; ++iter) {
            if (tid == 0) ctl[0] = 0;
            __syncthreads();
            int ch = 0;
#pragma unroll
            for (int c = 0; c < CE; ++c) {
                if (tid + c * W < m2) {
                    const int a = ec[c] >> 16, b = ec[c] & 0xffffu;
                    {
                        const double cd = __longlong_as_double((long long)d0[a]) + wc[c];
                        const ull cb = (ull)__double_as_longlong(cd);
                        const ull old = atomicMin(&d0[b], cb);
                        ch |= (cb < old);
                    }
                    if (TWO) {
                        const double cd = __longlong_as_double((long long)d1[a]) + wc[c];
                        const ull cb = (ull)__double_as_longlong(cd);
                        const ull old = atomicMin(&d1[b], cb);
                        ch |= (cb < old);
                    }
                }
            }
            if (ch) ctl[0] = 1;
            __syncthreads();
            const int any = ctl[0];
            __syncthreads();
            if (!any) break;
        }
        return;
    }
    for (int iter = 0; iter <= n; ++iter) {
        if (tid == 0) ctl[0] = 0;
        __syncthreads();
        int ch = 0;
        for (int j = tid; j < m2; j += W) {
            const unsigned e = dir[j];
            const int a = e >> 16, b = e & 0xffffu;
            const double w = LW(j);
            {
                const double c = __longlong_as_double((long long)d0[a]) + w;
                const ull cb = (ull)__double_as_longlong(c);
                const ull old = atomicMin(&d0[b], cb);
                ch |= (cb < old);
            }
            if (TWO) {
                const double c = __longlong_as_double((long long)d1[a]) + w;
                const ull cb = (ull)__double_as_longlong(c);
                const ull old = atomicMin(&d1[b], cb);
                ch |= (cb < old);
            }
        }
        if (ch) ctl[0] = 1;
        __syncthreads();
        const int any = ctl[0];
        __syncthreads();
        if (!any) break;
    }
}

// ---- bitonic sort of (key u64, payload u32), ascending by key, P a power of two ------------------------------------
// One compare-exchange of the bitonic network between lanes l and l^J of a wavefront (element i = base + lane against
// i^J).  Same rule as the LDS pass below: the pair is swapped iff (key_lo > key_hi) == up, equal keys stay put.
template <int J>
__device__ __forceinline__ void bitonic_lane_step(ull& kk, unsigned& vv, bool up) {
    const ull ok = ((ull)tlc_lane_xor_u32<J>((unsigned)(kk >> 32)) << 32) | (ull)tlc_lane_xor_u32<J>((unsigned)kk);
    const unsigned ov = tlc_lane_xor_u32<J>(vv);
    const bool lo = (tlc_lane() & J) == 0;
    const ull klo = lo ? kk : ok, khi = lo ? ok : kk;
    // (round 6, measured and not kept: the same rule as two compares and arithmetic on the lane masks -- `(kk != ok) && ((kk > ok) != hi) == up`
    // -- without these four selects: -7.7 M vector instructions per batch (-3 %), pipelined batch +0.5 % in three rounds of two libraries in
    // turn: the v_cmp -> s_xor / s_xnor / s_and -> v_cndmask chain is longer than the selects it replaces, and the sorts wait on latency)
    if (((klo > khi) == up) && (klo != khi)) { kk = ok; vv = ov; }
}
// sub-stages j = JMAX, JMAX/2, ..., 1 of stage k on one 64-element block held one element per lane
template <int JMAX>
__device__ __forceinline__ void bitonic_lane_tail(ull& kk, unsigned& vv, bool up) {
    if (JMAX >= 32) bitonic_lane_step<32>(kk, vv, up);
    if (JMAX >= 16) bitonic_lane_step<16>(kk, vv, up);
    if (JMAX >= 8) bitonic_lane_step<8>(kk, vv, up);
    if (JMAX >= 4) bitonic_lane_step<4>(kk, vv, up);
    if (JMAX >= 2) bitonic_lane_step<2>(kk, vv, up);
    bitonic_lane_step<1>(kk, vv, up);
}

// Bitonic sort of (key, payload) pairs in LDS, P a power of two.  Every compare-exchange whose partners sit in the same
// 64-element block (j <= 32) runs in registers with cross-lane moves; only the sub-stages with j >= 64 go through LDS
// with a barrier each: P = 1024 needs 15 barriers instead of 55, P <= 64 one.
template <int W>
__device__ __forceinline__ void bitonic_sort(ull* key, unsigned* val, int P) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int NWV = W / 64;
    const int nblk = (P + 63) >> 6;
    // stages k = 2 .. min(P, 64): entirely inside a block
    for (int b = wave; b < nblk; b += NWV) {
        const int i = b * 64 + lane;
        ull kk = i < P ? key[i] : 0ull;
        unsigned vv = i < P ? val[i] : 0u;
        if (P >= 2) bitonic_lane_tail<1>(kk, vv, (i & 2) == 0);
        if (P >= 4) bitonic_lane_tail<2>(kk, vv, (i & 4) == 0);
        if (P >= 8) bitonic_lane_tail<4>(kk, vv, (i & 8) == 0);
        if (P >= 16) bitonic_lane_tail<8>(kk, vv, (i & 16) == 0);
        if (P >= 32) bitonic_lane_tail<16>(kk, vv, (i & 32) == 0);
        if (P >= 64) bitonic_lane_tail<32>(kk, vv, (i & 64) == 0);
        if (i < P) { key[i] = kk; val[i] = vv; }
    }
    __syncthreads();
    for (int k = 128; k <= P; k <<= 1) {
        for (int j = k >> 1; j >= 64; j >>= 1) {
            for (int t = tid; t < (P >> 1); t += W) {
                const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
                const int x = i | j;
                const bool up = ((i & k) == 0);
                const ull ki = key[i], kx = key[x];
                if ((ki > kx) == up && ki != kx) {
                    key[i] = kx; key[x] = ki;
                    const unsigned vi = val[i], vx = val[x];
                    val[i] = vx; val[x] = vi;
                }
            }
            __syncthreads();
        }
        for (int b = wave; b < nblk; b += NWV) {
            const int i = b * 64 + lane;
            ull kk = key[i];
            unsigned vv = val[i];
            bitonic_lane_tail<32>(kk, vv, (i & k) == 0);
            key[i] = kk; val[i] = vv;
        }
        __syncthreads();
    }
}

// a tier sorts at most MM (a power of two) edges: runs of P/2 and at most P/4 elements, 3 MM / 4 in all; HUGE (MM = 0): 8
__host__ __device__ constexpr int sort_hold(int MM, int W) { return MM <= 0 ? 8 : ((3 * MM / 4 + W - 1) / W < 1 ? 1 : ((3 * MM / 4 + W - 1) / W > 8 ? 8 : (3 * MM / 4 + W - 1) / W)); }

// Sort of `cnt` pairs whose array is padded with ~0 keys up to P = pow2ceil(cnt) (callers read the first cnt positions).
// The bitonic network costs P log^2 P whatever cnt is, and a count just above a power of two pays for twice its size (the
// batch's heaviest vicinity has 2 270 edges: P = 4 096).  When the part above P/2 fits a quarter of P, the two parts are
// sorted as runs of P/2 and R = pow2ceil(cnt - P/2) elements and merged by rank: an element's final position is its index in
// its own run plus the number of elements of the other run ordered before it (binary search in LDS; equal keys: the lower
// run first), elements held in registers across the barrier.  Falls back to the plain network when the runs do not pay
// or a thread would have to hold more than eight elements (HUGE tier).
// QS: the most elements a thread can be asked to hold (8 unless the caller knows its cap: 3/4 of the largest P over W)
template <int W, int QS = 8>
__device__ __forceinline__ void sort_padded(ull* key, unsigned* val, int cnt) {
    const int P = pow2ceil(cnt < 2 ? 2 : cnt);
    const int H = P >> 1, r = cnt - H;
    const int R = r > 0 ? pow2ceil(r < 2 ? 2 : r) : P;
    const int total = H + R;
    if (P < 256 || 4 * R > P || total > QS * W) {
        bitonic_sort<W>(key, val, P);
        return;
    }
    bitonic_sort<W>(key, val, H);
    bitonic_sort<W>(key + H, val + H, R);
    ull kk[QS];
    unsigned vv[QS];
    int pp[QS];
#pragma unroll
    for (int q = 0; q < QS; ++q) {
        const int i = (int)threadIdx.x + q * W;
        pp[q] = -1;
        if (i < total) {
            const ull k = key[i];
            kk[q] = k;
            vv[q] = val[i];
            int lo, hi;
            if (i < H) {                               // strictly smaller elements of the upper run
                lo = 0; hi = R;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[H + mid] < k) lo = mid + 1; else hi = mid; }
                pp[q] = i + lo;
            } else {                                   // smaller or equal elements of the lower run
                lo = 0; hi = H;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (key[mid] <= k) lo = mid + 1; else hi = mid; }
                pp[q] = (i - H) + lo;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < QS; ++q)
        if (pp[q] >= 0) { key[pp[q]] = kk[q]; val[pp[q]] = vv[q]; }
    __syncthreads();
}

template <typename idx_t>
__device__ __forceinline__ int uf_find(idx_t* comp, int p) {
    // path halving exactly as accelerated_PD.py:53-58
    int c = comp[p];
    while (p != c) {
        const int g = comp[c];
        comp[p] = (idx_t)g;
        p = g;
        c = comp[p];
    }
    return p;
}

__device__ __forceinline__ double key_to_f64(ull k) {
    const ull b = (k >> 63) ? (k ^ (1ull << 63)) : ~k;
    return __longlong_as_double((long long)b);
}

// W-thread barrier-separated "did any thread set the flag" on ctl[slot]
template <int W>
__device__ __forceinline__ bool block_any(bool v, int* ctl, int slot) {
    if (W == 64) return __ballot(v) != 0ull;
    if (threadIdx.x == 0) ctl[slot] = 0;
    __syncthreads();
    if (v) ctl[slot] = 1;
    __syncthreads();
    const int r = ctl[slot];
    __syncthreads();
    return r != 0;
}

// pointer jumping until every node points at its root
template <int W, typename idx_t>
__device__ __forceinline__ void flatten(idx_t* comp, int n, int* ctl) {
    for (int it = 0; it < 32; ++it) {
        bool ch = false;
        for (int y = threadIdx.x; y < n; y += W) {
            const int c = comp[y];
            const int cc = comp[c];
            if (cc != c) { comp[y] = (idx_t)cc; ch = true; }
        }
        __syncthreads();
        if (!block_any<W>(ch, ctl, 0)) break;
    }
}

// Where the diagram points go.  Counters live in LDS (ctl[6..8], ctl[2]) so that parallel and serial code share them.
// Batch path: node-index pairs (rank space) into LDS: every PD coordinate is a copy of some f[node].
struct PtsSink {
    unsigned* pts;
    int* ctl;
    static constexpr bool want_down = false, is_global = false;
    __device__ __forceinline__ void up(const double*, int b, int d) { pts[ctl[2]++] = ((unsigned)b << 16) | (unsigned)d; ctl[6]++; }
    __device__ __forceinline__ void ext0(const double*, int mn, int mx) { pts[ctl[2]++] = ((unsigned)mn << 16) | (unsigned)mx; }
    __device__ __forceinline__ void down(const double*, int, int) {}
    __device__ __forceinline__ void one_at(const double*, int slot, int, int b, int d) { pts[slot] = ((unsigned)b << 16) | (unsigned)d; }
};
// development check (TLC_DC_VERIFY): the higher endpoint per query
struct VerifySink {
    unsigned short* o;
    [[maybe_unused]] static constexpr bool want_down = false, is_global = false;
    __device__ __forceinline__ void one_at(const double*, int, int k, int, int d) { o[k] = (unsigned short)d; }
};
// tlc_pd_from_filtration: values straight to the caller's arrays.
struct GlobalSink {
    double *pd_up, *pd_down, *pd_one, *e0;
    int* ctl;
    static constexpr bool want_down = true, is_global = true;
    __device__ __forceinline__ void up(const double* f, int b, int d) { const int k = ctl[6]++; pd_up[2 * k] = f[b]; pd_up[2 * k + 1] = f[d]; }
    __device__ __forceinline__ void ext0(const double* f, int mn, int mx) { e0[0] = f[mn]; e0[1] = f[mx]; }
    __device__ __forceinline__ void down(const double* f, int b, int d) { const int k = ctl[7]++; pd_down[2 * k] = f[b]; pd_down[2 * k + 1] = f[d]; }
    __device__ __forceinline__ void one_at(const double* f, int, int k, int b, int d) { pd_one[2 * k] = f[b]; pd_one[2 * k + 1] = f[d]; }
};

// ---- relabel the nodes by ascending f (stable for ties) and move the edges to rank space -------------------------------
// After this f[0] = min, f[n-1] = max, every comparison of f values between nodes is an index comparison, and
// ends[e] = lo<<16 | hi with lo < hi the two ranks of edge e.  M.mark holds rank[node] until the cycle swap reuses it.
template <int W, typename idx_t>
__device__ __forceinline__ void relabel_by_rank(Mem<idx_t>& M, int n, int m) {
    const int tid = threadIdx.x;
    const int Pn = pow2ceil(n < 2 ? 2 : n);
    for (int i = tid; i < Pn; i += W) {
        M.keyS[i] = i < n ? f64_key(M.f[i]) : ~0ull;
        M.valS[i] = (unsigned)i;
    }
    __syncthreads();
    bitonic_sort<W>(M.keyS, M.valS, Pn);
    for (int r = tid; r < n; r += W) {
        M.f[r] = key_to_f64(M.keyS[r]);
        M.mark[M.valS[r]] = (idx_t)r;
    }
    __syncthreads();
    for (int e = tid; e < m; e += W) {
        const unsigned ab = M.dir[e];
        const unsigned ra = M.mark[ab >> 16], rb = M.mark[ab & 0xffffu];
        M.dir[e] = ra < rb ? ((ra << 16) | rb) : ((rb << 16) | ra);
    }
    __syncthreads();
}

// sort the m edges by the ascending (DESC=false) or descending (DESC=true) perturbed key; valS[pos] = edge id
template <int W, typename idx_t, bool DESC, int QS = 8>
__device__ __forceinline__ void sort_edges(Mem<idx_t>& M, int m) {
    const int tid = threadIdx.x;
    const int P = pow2ceil(m < 2 ? 2 : m);
    for (int e = tid; e < P; e += W) {
        if (e < m) {
            const unsigned ab = M.dir[e];
            const double flo = M.f[ab >> 16], fhi = M.f[ab & 0xffffu];
            M.keyS[e] = DESC ? ~f64_key(key_desc(flo, fhi)) : f64_key(key_asc(flo, fhi));
        } else {
            M.keyS[e] = ~0ull;
        }
        M.valS[e] = (unsigned)e;
    }
    __syncthreads();
    sort_padded<W, QS>(M.keyS, M.valS, m);
}

// ---- one filtration pass as a minimum-spanning-forest computation in sorted-position order -----------------------------
// The reference runs Kruskal over the sorted simplices with the elder rule (accelerated_PD.py:46-68 ascending, :83-109
// descending).  Which edges join two components (the "Neg"/tree edges) depends only on the total order, so they are found
// with Boruvka rounds (every component picks its earliest outgoing edge: atomicMin on the sorted position), all lanes busy.
// The elder-rule PAIRS additionally depend on the merge order, but only through the component minima (maxima):
//   * the first edge (in sorted order) at a node y whose other endpoint is older attaches the singleton {y} to a component
//     that already holds an older vertex; the pair it emits has zero persistence and it never changes a component's
//     oldest vertex, nor can it bridge two components earlier than the reference would (no other edge at y precedes it);
//     all of these are applied at once ("pre-merge"), leaving one basin per local extremum;
//   * Boruvka started from the basins yields the remaining tree edges R (|R| = #basins - 1, a handful);
//   * only R is replayed serially, in sorted order, with the reference's elder rule.
// WANT_PAIRS=false: tree bits only (batch path, descending pass: only the Pos/Neg split and #Neg are consumed).
template <int W, typename idx_t, bool DESC, bool WANT_PAIRS, class Sink>
__device__ __forceinline__ void mst_pass(Mem<idx_t>& M, Sink& sink, int n, int m, unsigned flags) {
    const int tid = threadIdx.x;
    const bool keep0 = (flags & TLC_KEEP_ZERO_PERS) != 0;
    const unsigned INF = 0xFFFFFFFFu;
    const int nwords = (m + 31) >> 5;
    const unsigned* ord = M.valS;
    const unsigned* ends = M.dir;
    double* f = M.f;
    for (int w = tid; w < nwords; w += W) { M.tbits[w] = 0u; M.rbits[w] = 0u; }
    if (WANT_PAIRS) {
        for (int y = tid; y < n; y += W) M.first[y] = INF;
        __syncthreads();
        for (int pos = tid; pos < m; pos += W) {
            const unsigned ab = ends[ord[pos]];
            atomicMin(&M.first[ab >> 16], (unsigned)pos);
            atomicMin(&M.first[ab & 0xffffu], (unsigned)pos);
        }
        __syncthreads();
        // pre-merge: y joins the older endpoint of its first edge (ascending: the lower rank; descending: the higher)
        int npre = 0;
        for (int y0 = 0; y0 < n; y0 += W) {
            const int y = y0 + tid;
            int c = y;
            bool pre = false;
            if (y < n) {
                const unsigned p = M.first[y];
                if (p != INF) {
                    const unsigned ab = ends[ord[p]];
                    const int lo = ab >> 16, hi = ab & 0xffffu;
                    if (DESC ? (lo == y) : (hi == y)) {
                        c = DESC ? hi : lo;
                        pre = true;
                        atomicOr(&M.tbits[p >> 5], 1u << (p & 31));
                    }
                }
                M.comp[y] = (idx_t)c;
            }
            if (Sink::is_global && keep0) {
                // the Knowledge_Distillation fork keeps these zero-persistence pairs (:68-69 / :108-109): [f[y], f[y]]
                const ull mk = __ballot(pre);
                int off;
                if (W == 64) {
                    off = npre + __popcll(mk & tlc_lanemask_lt());
                    npre += __popcll(mk);
                } else {
                    if (tlc_lane() == 0) M.wcnt[tid >> 6] = __popcll(mk);
                    __syncthreads();
                    int before = 0, tot = 0;
#pragma unroll
                    for (int k = 0; k < W / 64; ++k) {
                        const int cc = M.wcnt[k];
                        if (k < (tid >> 6)) before += cc;
                        tot += cc;
                    }
                    off = npre + before + __popcll(mk & tlc_lanemask_lt());
                    npre += tot;
                    __syncthreads();
                }
                if (pre) {
                    if constexpr (Sink::is_global) {
                        double* dst = DESC ? sink.pd_down : sink.pd_up;
                        dst[2 * off] = f[y];
                        dst[2 * off + 1] = f[y];
                    }
                }
            }
        }
        __syncthreads();
        if (Sink::is_global && keep0 && tid == 0) M.ctl[DESC ? 7 : 6] = npre;
        flatten<W>(M.comp, n, M.ctl);
        for (int y = tid; y < n; y += W) M.comp2[y] = M.comp[y];
    } else {
        for (int y = tid; y < n; y += W) M.comp2[y] = (idx_t)y;
    }
    __syncthreads();
    // ---- Boruvka rounds on comp2 (kept flat: comp2[y] is y's root) ---------------------------------------------------
    unsigned* hook = M.first;
    for (int round = 0; round < 40; ++round) {
        for (int y = tid; y < n; y += W) { M.best[y] = INF; hook[y] = INF; }
        __syncthreads();
        bool found = false;
        for (int pos = tid; pos < m; pos += W) {
            if ((M.tbits[pos >> 5] >> (pos & 31)) & 1u) continue;
            const unsigned ab = ends[ord[pos]];
            const int ra = M.comp2[ab >> 16], rb = M.comp2[ab & 0xffffu];
            if (ra != rb) {
                atomicMin(&M.best[ra], (unsigned)pos);
                atomicMin(&M.best[rb], (unsigned)pos);
                found = true;
            }
        }
        __syncthreads();
        if (!block_any<W>(found, M.ctl, 0)) break;
        for (int y = tid; y < n; y += W) {
            const unsigned pos = M.best[y];
            if (pos == INF || M.comp2[y] != y) continue;
            const unsigned ab = ends[ord[pos]];
            const int ra = M.comp2[ab >> 16], rb = M.comp2[ab & 0xffffu];
            const int other = (ra == y) ? rb : ra;
            atomicOr(&M.tbits[pos >> 5], 1u << (pos & 31));
            if (WANT_PAIRS) atomicOr(&M.rbits[pos >> 5], 1u << (pos & 31));
            // unique positions => the pick graph has only 2-cycles; the larger root of a mutual pick hooks
            if (M.best[other] != pos || y > other) hook[y] = (unsigned)other;
        }
        __syncthreads();
        for (int y = tid; y < n; y += W)
            if (hook[y] != INF) M.comp2[y] = (idx_t)hook[y];
        __syncthreads();
        flatten<W>(M.comp2, n, M.ctl);
    }
    // ---- replay the few remaining tree edges with the elder rule -------------------------------------------------------
    if (WANT_PAIRS) {
        if (tid == 0) {
            for (int w = 0; w < nwords; ++w) {
                unsigned bits = M.rbits[w];
                while (bits) {
                    const int pos = (w << 5) + __builtin_ctz(bits);
                    bits &= bits - 1;
                    const unsigned ab = ends[ord[pos]];
                    const int lo = ab >> 16, hi = ab & 0xffffu;
                    const int pa = uf_find(M.comp, lo), pb = uf_find(M.comp, hi);
                    if (pa == pb) continue;
                    const int small = pa < pb ? pa : pb, large = pa + pb - small;      // rank order == f order
                    if (!DESC) {
                        // :63-68  the younger root `large` dies at the higher endpoint of the edge
                        if (keep0 || f[large] < f[hi]) sink.up(f, large, hi);
                        M.comp[large] = (idx_t)small;
                    } else {
                        // :101-107  the root with the smaller f dies at the lower endpoint
                        if (Sink::want_down && (keep0 || f[small] > f[lo])) sink.down(f, small, lo);
                        M.comp[small] = (idx_t)large;
                    }
                }
            }
        }
        __syncthreads();
    }
}

// ---- Pos / Neg lists from the tree bits of the descending pass, in descending-pass order (:99,109) --------------------
template <int W, typename idx_t>
__device__ __forceinline__ void split_pos_neg(Mem<idx_t>& M, int m, int MMcap) {
    const int tid = threadIdx.x;
    int npos = 0, nneg = 0;
    for (int p0 = 0; p0 < m; p0 += W) {
        const int pos = p0 + tid;
        bool isneg = false, valid = pos < m;
        unsigned eid = 0;
        if (valid) {
            isneg = (M.tbits[pos >> 5] >> (pos & 31)) & 1u;
            eid = M.valS[pos];
        }
        const ull mp = __ballot(valid && !isneg), mn = __ballot(valid && isneg);
        int offp, offn;
        if (W == 64) {
            offp = npos + __popcll(mp & tlc_lanemask_lt());
            offn = nneg + __popcll(mn & tlc_lanemask_lt());
            npos += __popcll(mp);
            nneg += __popcll(mn);
        } else {
            if (tlc_lane() == 0) { M.wcnt[tid >> 6] = __popcll(mp); M.wcnt[16 + (tid >> 6)] = __popcll(mn); }
            __syncthreads();
            int bp = 0, bn = 0, tp = 0, tn = 0;
#pragma unroll
            for (int k = 0; k < W / 64; ++k) {
                const int cp = M.wcnt[k], cn = M.wcnt[16 + k];
                if (k < (tid >> 6)) { bp += cp; bn += cn; }
                tp += cp; tn += cn;
            }
            offp = npos + bp + __popcll(mp & tlc_lanemask_lt());
            offn = nneg + bn + __popcll(mn & tlc_lanemask_lt());
            npos += tp; nneg += tn;
            __syncthreads();
        }
        if (valid) {
            if (isneg) M.pn[MMcap - 1 - offn] = eid;
            else M.pn[offp] = eid;
        }
    }
    if (tid == 0) { M.ctl[3] = npos; M.ctl[4] = nneg; }
    __syncthreads();
}

// Accelerate_PD (accelerated_PD.py:115-178).
//
// Tree state per node (u32 arrays carved from the dead sort / union-find regions): par[x], key[x] = (ascending rank + 1)
// << 8 of the edge (x, par x), mark[x] = stamp of the last walk through x, pmP[x] / pmQ[x] = (running maximum rank+1, child endpoint
// of that edge) of the p- resp. q-walk on arrival at x.  Every tree edge carries the ascending rank of its key, so "the
// first maximum of 'asc' over the loop" (:155-159) is an integer maximum.
// The loop of a Pos edge (p,q) is found by TWO LANES of one wavefront walking up from p and from q in the same instruction
// stream.  A step is: publish the running maximum at the next node, then ds_wrxchg the stamp there; the exchange returns
// the other walk's stamp exactly once, at the lowest common ancestor (two lanes hitting the same word in one instruction
// are serialised by the LDS, so simultaneous arrival is detected too).  One LDS round trip per step, no second pass over
// the loop, cost per Pos edge ~ cycle length instead of the two root paths of :131-148.  The path eversion that swaps the
// edges (:168-176) re-parents the whole path in one parallel step.

// the cycle swap's per-node tables: NMcap + 1 slots each (slot NMcap is a spare node: the target of an idle walker's traffic)
struct SwapTables {
    unsigned *par, *key, *mark;
    ull *pmP, *pmQ;
};
__host__ __device__ constexpr size_t swap_table_bytes(int NMcap) { return (size_t)(NMcap + 1) * 28 + 16; }
__device__ __forceinline__ SwapTables carve_swap(void* base, int NMcap) {
    const int NS = NMcap + 1;
    SwapTables T;
    T.par = (unsigned*)base;
    T.key = T.par + NS;
    T.mark = T.key + NS;
    T.pmP = (ull*)(T.mark + NS + (NS & 1));
    T.pmQ = T.pmP + NS;
    return T;
}

// Orient the spanning tree of the Neg edges (:119-125) into T.par / T.key and clear the stamps; all W threads.  Returns whether
// some node was not reached (disconnected input: only then does a query have to test its endpoints).
template <int W, typename idx_t>
__device__ __forceinline__ bool ext1_build_tree(Mem<idx_t>& M, const SwapTables& T, int n, int MMcap, int NMcap) {
    const int tid = threadIdx.x;
    const unsigned NONE = 0xffffffffu;
    const int nneg = M.ctl[4];
    const unsigned* ends = M.dir;
    unsigned *par = T.par, *key = T.key, *mark = T.mark;
    // spanning tree of the Neg edges (:119-125).  In a connected subgraph any root gives the same diagram; rank 0 (a root of the
    // vicinity) keeps the tree shallow.  The reference roots it at the first endpoint of its FIRST Neg edge (:126-127): in a
    // disconnected subgraph that decides which component's Pos edges can be walked at all (the others raise KeyError there), so
    // when rank 0 turns out to sit in another component than that edge -- or is not incident to a Neg edge -- the tree is built
    // again from the reference's choice.  (the Neg list in M.pn and the edge tables in M.dir are outside the regions re-used here)
    const int ref_root = (int)(ends[M.pn[MMcap - 1]] >> 16);
    int root = 0;
    {
        bool has0 = false;
        for (int k = tid; k < nneg; k += W) has0 |= ((ends[M.pn[MMcap - 1 - k]] >> 16) == 0u);
        if (!block_any<W>(has0, M.ctl, 5)) root = ref_root;
    }
    bool any_unreached = false;
    for (int attempt = 0; attempt < 2; ++attempt) {
        __syncthreads();
        for (int i = tid; i < n; i += W) { par[i] = NONE; key[i] = 0u; mark[i] = 0u; }
        __syncthreads();
        if (tid == 0) par[root] = (unsigned)root;
        __syncthreads();
        if (nneg <= 4 * W) {
            // (every LDS tier: at most four Neg edges per thread) the edges stay in registers over the rounds and retire once
            // oriented, so a round costs the two parent reads of the edges still open, not the whole list again
            unsigned eab[4], ekey[4];
            bool open[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = tid + q * W;
                open[q] = k < nneg;
                eab[q] = 0u; ekey[q] = 0u;
                if (open[q]) {
                    const unsigned eid = M.pn[MMcap - 1 - k];
                    eab[q] = ends[eid];
                    ekey[q] = (M.arank[eid] + 1u) << 8;
                }
            }
            for (int round = 0; round <= n; ++round) {
                bool prog = false;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (open[q]) {
                        const int a = eab[q] >> 16, b = eab[q] & 0xffffu;
                        const unsigned pa = par[a], pb = par[b];
                        if (pa != NONE && pb == NONE) { par[b] = (unsigned)a; key[b] = ekey[q]; prog = true; open[q] = false; }
                        else if (pb != NONE && pa == NONE) { par[a] = (unsigned)b; key[a] = ekey[q]; prog = true; open[q] = false; }
                    }
                }
                __syncthreads();
                if (!block_any<W>(prog, M.ctl, 5)) break;
            }
        } else {
            for (int round = 0; round <= n; ++round) {
                bool prog = false;
                for (int k = tid; k < nneg; k += W) {
                    const unsigned eid = M.pn[MMcap - 1 - k];
                    const unsigned ab = ends[eid];
                    const int a = ab >> 16, b = ab & 0xffffu;
                    const unsigned pa = par[a], pb = par[b];
                    if (pa != NONE && pb == NONE) { par[b] = (unsigned)a; key[b] = (M.arank[eid] + 1u) << 8; prog = true; }
                    else if (pb != NONE && pa == NONE) { par[a] = (unsigned)b; key[a] = (M.arank[eid] + 1u) << 8; prog = true; }
                }
                __syncthreads();
                if (!block_any<W>(prog, M.ctl, 5)) break;
            }
        }
        // nodes the tree did not reach (disconnected input): only then does a query have to test its endpoints
        bool unreached = false;
        for (int i = tid; i < n; i += W) unreached |= (par[i] == NONE);
        any_unreached = block_any<W>(unreached, M.ctl, 5);
        if (!any_unreached || root == ref_root || par[ref_root] != NONE) break;      // (uniform)
        root = ref_root;
    }
    // above the root sits the spare slot, its own parent, behind edges of key 0: a walk that passes the root keeps
    // stepping in place there, so a step needs no "at the root" case
    if (tid == 0) { par[root] = (unsigned)NMcap; key[root] = 0u; par[NMcap] = (unsigned)NMcap; key[NMcap] = 0u; mark[NMcap] = 0u; }
    __syncthreads();
    return any_unreached;
}

// The queries of the cycle swap -- (endpoints p<<16|q in rank space, (ascending rank + 1) << 8) of the Pos edges in
// descending-pass order -- do not depend on the tree, so they are fetched ahead of their use: no load of this chain is ever
// waited for inside a query.  QueryLds: from the Pos list / edge tables in LDS (edge id two queries ahead, endpoints and rank
// one ahead).
struct QueryLds {
    const unsigned *pn, *ends, *arank;
    int npos;
    unsigned eid1, n_pq, n_ar;
    __device__ __forceinline__ void init(int) {
        eid1 = npos > 0 ? pn[npos > 1 ? 1 : 0] : 0u;
        n_pq = 0u; n_ar = 0u;
        if (npos > 0) { const unsigned e0 = pn[0]; n_pq = ends[e0]; n_ar = (arank[e0] + 1u) << 8; }
    }
    // (unconditional: past the end it re-reads the last query's operands, which nobody uses)
    __device__ __forceinline__ void advance(int pi, int) {
        n_pq = ends[eid1];
        n_ar = (arank[eid1] + 1u) << 8;
        eid1 = pn[pi + 2 < npos ? pi + 2 : npos - 1];
    }
};
// QueryGlobal: from the packed list (rank word << 32 | endpoints) a tier kernel left in HBM: lane l of the wavefront holds
// query 64 b + l of the current block b and of the next one, a query is a v_readlane away, and a block is loaded (one
// coalesced 512-byte read) 64 queries before its first use.
struct QueryGlobal {
    const ull* q;
    int npos;
    unsigned cur_lo, cur_hi, nxt_lo, nxt_hi, n_pq, n_ar;
    __device__ __forceinline__ ull load(int idx) const { return q[idx < npos ? idx : (npos > 0 ? npos - 1 : 0)]; }
    __device__ __forceinline__ void init(int lane) {
        const ull c = load(lane), x = load(64 + lane);
        cur_lo = (unsigned)c; cur_hi = (unsigned)(c >> 32);
        nxt_lo = (unsigned)x; nxt_hi = (unsigned)(x >> 32);
        n_pq = (unsigned)__builtin_amdgcn_readlane((int)cur_lo, 0);
        n_ar = (unsigned)__builtin_amdgcn_readlane((int)cur_hi, 0) & ~0xffu;     // (bit 0: the divide and conquer's flag)
    }
    __device__ __forceinline__ void advance(int pi, int lane) {
        const int idx = pi + 1;
        if ((idx & 63) == 0) {
            cur_lo = nxt_lo; cur_hi = nxt_hi;
            const ull x = load(idx + 64 + lane);
            nxt_lo = (unsigned)x; nxt_hi = (unsigned)(x >> 32);
        }
        n_pq = (unsigned)__builtin_amdgcn_readlane((int)cur_lo, idx & 63);
        n_ar = (unsigned)__builtin_amdgcn_readlane((int)cur_hi, idx & 63) & ~0xffu;
    }
};

// The serial part of Accelerate_PD: one wavefront (all 64 lanes call; lanes 0 and 1 walk).  `recs` = 2 x 65 path records.
// Points go to sink.one_at(f, out0 + k, k, ...); returns the number of points emitted.
template <class Sink, class Query>
__device__ __forceinline__ int ext1_walk(const SwapTables& T, ull* recs, Query& qs, int npos, int NMcap, bool any_unreached,
                                         Sink& sink, const double* f, bool keep0, int out0, ull* pc) {
    const unsigned NONE = 0xffffffffu;
    unsigned *par = T.par, *key = T.key, *mark = T.mark;
    ull *pmP = T.pmP, *pmQ = T.pmQ;
    int n_out = 0;                         // points emitted by this stage (wave-uniform)
    {
        // this wave carries the critical serial chain of the batch: let it win issue arbitration against the other
        // kernels' waves that share the SIMD
        __builtin_amdgcn_s_setprio(3);
        const int lane = tlc_lane();
        const bool qside = (lane & 1) != 0;
        ull* pmMine = qside ? pmQ : pmP;
        const ull* pmTheirs = qside ? pmP : pmQ;
        // path records: (key << 32 | node) of the edge crossed at step s of either walk (slot 64 = spare); a swap then
        // re-parents its whole path in one parallel step instead of one dependent LDS round trip per node
        ull* rec = recs + (qside ? 65 : 0);
        unsigned stamp = 0;
        qs.init(lane);
        unsigned n_pq = qs.n_pq, n_ar = qs.n_ar;                   // endpoints / ascending rank of query pi
        // ... and so are the walkers' first nodes, read under the previous query's swap
        const unsigned side_shift = qside ? 0u : 16u;           // a walker's first node: p in the high, q in the low half
        int n_cur = (int)((n_pq >> side_shift) & 0xffffu);
        unsigned n_pcur = par[n_cur], n_kcur = key[n_cur];
        // (PHASE_DEBUG=2: per-query clocks inside the walk as well; they slow it down by half, so level 1 leaves them out)
#if defined(TLC_PHASE_DEBUG) && TLC_PHASE_DEBUG > 1
        ull dbg_a = 0, dbg_b = 0, dbg_c = 0, dbg_t0 = 0, dbg_t1 = 0, dbg_t2 = 0;
#define DBG_T(v) v = clock64()
#else
#define DBG_T(v)
#endif
        for (int pi = 0; pi < npos; ++pi) {
            DBG_T(dbg_t0);
            const unsigned pq = n_pq, ar = n_ar;
            int cur = n_cur;
            unsigned pcur = n_pcur, kcur = n_kcur;
            qs.advance(pi, lane);
            n_pq = qs.n_pq;
            n_ar = qs.n_ar;
            const int p = pq >> 16, q = pq & 0xffffu;              // f[p] <= f[q]: low_value = f[p] (:162)
            // winner's record: hi = (max rank + 1) << 8 | step of that edge, lo = child << 16 | parent (ranks are unique,
            // so comparing hi words compares ranks)
            unsigned res_hi = 0, res_lo = 0;
            unsigned res_s = 0;                // side of the winner: 0 = p-walk, 1 = q-walk
            if (!any_unreached || __ballot(lane < 2 && pcur == NONE) == 0ull) {   // else other component (the reference raises KeyError)
                stamp += 2;                                        // < 2^26: at most 2^24 Pos edges per subgraph (the key packing)
                int fl = 0;
                DBG_T(dbg_t1);
                if (lane < 2) {
                    // The walk is straight-line code for both lanes; the only branch in a step is the uniform exit.
                    const unsigned mine = stamp + (unsigned)(lane & 1), theirs = stamp + (unsigned)((lane & 1) ^ 1);
                    unsigned mx_hi = 0, mx_lo = 0;                 // heaviest edge this walk has crossed (0 = none)
                    unsigned step = 0, o_hi, o_lo;
                    mark[cur] = mine;
                    pmMine[cur] = 0ull;
                    for (;;) {
                        const unsigned c_hi = kcur | (step < 255u ? step : 255u);
                        const bool upd = c_hi > mx_hi;
                        mx_hi = upd ? c_hi : mx_hi;
                        mx_lo = upd ? (((unsigned)cur << 16) | pcur) : mx_lo;
                        rec[step < 64u ? step : 64u] = ((unsigned long long)kcur << 32) | (unsigned)cur;
                        pmMine[pcur] = ((unsigned long long)mx_hi << 32) | mx_lo;  // publish before taking the stamp
                        unsigned old = atomicExch(&mark[pcur], mine);              // ds_wrxchg_rtn_b32
                        unsigned pn = par[pcur], kn = key[pcur];
                        const unsigned long long o = pmTheirs[pcur];               // only meaningful at the meeting node
                        o_hi = (unsigned)(o >> 32);
                        o_lo = (unsigned)o;
                        // one LDS round trip per step: without this the compiler sinks the reads below the test on `old`
                        asm volatile("" : "+v"(old), "+v"(pn), "+v"(kn), "+v"(o_hi), "+v"(o_lo));
                        const ull fm = __ballot(old == theirs);
                        cur = (int)pcur;                                           // (unused once the walks have met)
                        pcur = pn;
                        kcur = kn;
                        ++step;
                        if (fm) { fl = __builtin_ctzll(fm); break; }               // the step's only branch
                    }
                    const bool mwin = mx_hi >= o_hi;
                    res_hi = mwin ? mx_hi : o_hi;
                    res_lo = mwin ? mx_lo : o_lo;
                    res_s = (unsigned)(lane & 1) ^ (mwin ? 0u : 1u);
                }
                DBG_T(dbg_t2);
#if defined(TLC_PHASE_DEBUG) && TLC_PHASE_DEBUG > 1
                dbg_a += dbg_t1 - dbg_t0; dbg_b += dbg_t2 - dbg_t1;
#endif
                fl = __builtin_amdgcn_readfirstlane(fl);           // lanes 0 and 1 agree; the rest follow lane 0
                res_hi = __builtin_amdgcn_readlane(res_hi, fl);
                res_lo = __builtin_amdgcn_readlane(res_lo, fl);
                res_s = __builtin_amdgcn_readlane(res_s, fl);
            }
            // the next query's first nodes (their parent and key are read after the swap's writes, below)
            n_cur = (int)((n_pq >> side_shift) & 0xffffu);
            if (res_hi == 0u) { n_pcur = par[n_cur]; n_kcur = key[n_cur]; continue; }
            const int best = (int)(res_lo >> 16), bp = (int)(res_lo & 0xffffu);
            const unsigned bstep = res_hi & 0xffu;
            const int hin = best > bp ? best : bp;                                // large_value  (:160)
            // low_value > = large_value is dropped by the TLC fork (:164).  The batch path defers that test: a point with
            // zero persistence has weight 0 in the image, so it may stay in the list.
            bool emit = true;
            if (Sink::is_global && !keep0) emit = f[hin] > f[p];
            if (emit) {
                if (lane == 0) sink.one_at(f, out0 + n_out, n_out, p, hin);
                ++n_out;
            }
            // evert the path so that (p,q) replaces the removed tree edge (:168-176): node x_i of the winner's walk gets
            // x_{i-1} as parent (x_{-1} = the other endpoint) and inherits the key of the edge below it
            if (bstep < 64u) {
                const ull* rr = recs + (res_s ? 65 : 0);
                const bool mine_i = (unsigned)lane <= bstep;
                const int li = mine_i ? lane : 0;
                const ull r1 = rr[li], r0 = rr[li > 0 ? li - 1 : 0];
                const unsigned xi = mine_i ? (unsigned)r1 : 0xffffffffu;
                const unsigned xprev = lane ? (unsigned)r0 : (unsigned)(res_s == 0 ? q : p);
                const unsigned kprev = lane ? (unsigned)(r0 >> 32) : ar;
                if (mine_i) { par[xi] = xprev; key[xi] = kprev; }
            } else {
                if (lane == 0) {
                    int node = res_s == 0 ? p : q, nodec = res_s == 0 ? q : p;
                    unsigned kin = ar;
                    while (nodec != best) {
                        const unsigned pp = par[node], kk = key[node];
                        par[node] = (unsigned)nodec;
                        key[node] = kin;
                        nodec = node;
                        node = (int)pp;
                        kin = kk;
                    }
                }
            }
            // LDS executes in order: these reads see the swap; their latency hides under the next query's set-up
            n_pcur = par[n_cur];
            n_kcur = key[n_cur];
#if defined(TLC_PHASE_DEBUG) && TLC_PHASE_DEBUG > 1
            dbg_c += clock64() - dbg_t2;
#endif
        }
#if defined(TLC_PHASE_DEBUG) && TLC_PHASE_DEBUG > 1
        if (pc && lane == 0) { atomicAdd(&pc[15], dbg_a); atomicAdd(&pc[30], dbg_b); atomicAdd(&pc[31], dbg_c); }
#endif
        __builtin_amdgcn_s_setprio(0);
    }
    return n_out;
}


// ---- the same walk on COMPACT tables (tlc_pd_swap_kernel: capacities <= 1 024 edges, so ranks and stamps fit 16 / 12 bits) ----------
// Per node 12 bytes instead of 28, and per step three LDS instructions instead of six:
//   pk[x]   u32 = (ascending rank + 1 of the edge x -> parent) << 16 | parent      (parent 0xffff: not in the tree; NMcap: above the root)
//   slot[x] u64 = stamp << 52 | (heaviest key so far | step) << 32 | child << 16 | parent
// A walker EXCHANGES its stamp and its running maximum into the node it steps on: what comes back is the other walker's stamp and
// maximum when that one has been there (the meeting), so "publish, take the stamp, read theirs" of ext1_walk is one ds_wrxchg_rtn_b64.
// In the swap kernels these steps are a seventh of a pipelined batch's wave-instructions (profiles/r05_stream_assignment.txt, 3).
#ifndef TLC_SWAP_COMPACT
#define TLC_SWAP_COMPACT 1        /* make EXTRA="-DTLC_SWAP_COMPACT=0": the swap kernels on ext1_walk's tables (A/B) */
#endif
struct SwapTablesC {
    unsigned* pk;
    ull* slot;
};
__host__ __device__ constexpr size_t swap_table_c_bytes(int NMcap) { return (size_t)(NMcap + 2) * 12 + 16; }
__device__ __forceinline__ SwapTablesC carve_swap_c(void* base, int NMcap) {
    const int NS = NMcap + 1;
    SwapTablesC T;
    T.slot = (ull*)base;
    T.pk = (unsigned*)(T.slot + NS);
    return T;
}
// `recs` = 2 x 65 path records u32 (rank + 1) << 16 | node.  Same contract as ext1_walk (PtsSink only: the swap kernel's).
template <class Sink, class Query>
__device__ __forceinline__ int ext1_walk_c(const SwapTablesC& T, unsigned* recs, Query& qs, int npos, int NMcap, bool any_unreached,
                                           Sink& sink, int out0) {
    const unsigned NONE16 = 0xffffu;
    unsigned* pk = T.pk;
    ull* slot = T.slot;
    int n_out = 0;
    __builtin_amdgcn_s_setprio(3);
    const int lane = tlc_lane();
    const bool qside = (lane & 1) != 0;
    unsigned* rec = recs + (qside ? 65 : 0);
    unsigned stamp = 0;
    qs.init(lane);
    unsigned n_pq = qs.n_pq, n_ar = qs.n_ar;
    const unsigned side_shift = qside ? 0u : 16u;
    int n_cur = (int)((n_pq >> side_shift) & 0xffffu);
    unsigned n_pk = pk[n_cur];
    for (int pi = 0; pi < npos; ++pi) {
        const unsigned pq = n_pq, ar = n_ar;
        int cur = n_cur;
        unsigned pcur = n_pk & 0xffffu, kcur = (n_pk >> 16) << 8;
        qs.advance(pi, lane);
        n_pq = qs.n_pq;
        n_ar = qs.n_ar;
        const int p = pq >> 16, q = pq & 0xffffu;
        unsigned res_hi = 0, res_lo = 0, res_s = 0;
        if (!any_unreached || __ballot(lane < 2 && pcur == NONE16) == 0ull) {
            stamp += 2;                                            // <= 2 * 1 024 + 1: twelve bits
            int fl = 0;
            if (lane < 2) {
                const unsigned mine = (stamp + (unsigned)(lane & 1)) << 20, theirs = stamp + (unsigned)((lane & 1) ^ 1);
                unsigned mx_hi = 0, mx_lo = 0, step = 0;
                ull old;
                slot[cur] = (ull)mine << 32;
                for (;;) {
                    const unsigned c_hi = kcur | (step < 255u ? step : 255u);
                    const bool upd = c_hi > mx_hi;
                    mx_hi = upd ? c_hi : mx_hi;
                    mx_lo = upd ? (((unsigned)cur << 16) | pcur) : mx_lo;
                    rec[step < 64u ? step : 64u] = (kcur << 8) | (unsigned)cur;
                    old = atomicExch(&slot[pcur], ((ull)(mine | mx_hi) << 32) | mx_lo);      // ds_wrxchg_rtn_b64
                    unsigned pn = pk[pcur];
                    unsigned o_st = (unsigned)(old >> 52);
                    // one LDS round trip per step (see ext1_walk)
                    asm volatile("" : "+v"(o_st), "+v"(pn));
                    const ull fm = __ballot(o_st == theirs);
                    cur = (int)pcur;
                    pcur = pn & 0xffffu;
                    kcur = (pn >> 16) << 8;
                    ++step;
                    if (fm) { fl = __builtin_ctzll(fm); break; }
                }
                const unsigned o_hi = (unsigned)(old >> 32) & 0xfffffu, o_lo = (unsigned)old;
                const bool mwin = mx_hi >= o_hi;
                res_hi = mwin ? mx_hi : o_hi;
                res_lo = mwin ? mx_lo : o_lo;
                res_s = (unsigned)(lane & 1) ^ (mwin ? 0u : 1u);
            }
            fl = __builtin_amdgcn_readfirstlane(fl);
            res_hi = __builtin_amdgcn_readlane(res_hi, fl);
            res_lo = __builtin_amdgcn_readlane(res_lo, fl);
            res_s = __builtin_amdgcn_readlane(res_s, fl);
        }
        n_cur = (int)((n_pq >> side_shift) & 0xffffu);
        if (res_hi == 0u) { n_pk = pk[n_cur]; continue; }
        const int best = (int)(res_lo >> 16), bp = (int)(res_lo & 0xffffu);
        const unsigned bstep = res_hi & 0xffu;
        const int hin = best > bp ? best : bp;
        if (lane == 0) sink.one_at(nullptr, out0 + n_out, n_out, p, hin);
        ++n_out;
        const unsigned ar16 = (ar >> 8) << 16;
        if (bstep < 64u) {
            const unsigned* rr = recs + (res_s ? 65 : 0);
            const bool mine_i = (unsigned)lane <= bstep;
            const int li = mine_i ? lane : 0;
            const unsigned r1 = rr[li], r0 = rr[li > 0 ? li - 1 : 0];
            const unsigned xprev = lane ? (r0 & 0xffffu) : (unsigned)(res_s == 0 ? q : p);
            const unsigned kprev = lane ? (r0 & 0xffff0000u) : ar16;
            if (mine_i) pk[r1 & 0xffffu] = kprev | xprev;
        } else {
            if (lane == 0) {
                int node = res_s == 0 ? p : q, nodec = res_s == 0 ? q : p;
                unsigned kin = ar16;
                while (nodec != best) {
                    const unsigned old_pk = pk[node];
                    pk[node] = kin | (unsigned)nodec;
                    nodec = node;
                    node = (int)(old_pk & 0xffffu);
                    kin = old_pk & 0xffff0000u;
                }
            }
        }
        n_pk = pk[n_cur];
    }
    __builtin_amdgcn_s_setprio(0);
    return n_out;
}

}  // namespace
#include "ext1_dc.h"
namespace {

// Hand-off record of one subgraph between a tier kernel (parallel stages, W threads, the tier's full LDS footprint) and
// tlc_pd_swap_kernel (the serial cycle swap + the image: one wavefront and a quarter of the LDS, so that the long serial
// tails of a tier run at several times the tier kernel's residency).  Slot layout, NM / MM = the tier's capacities:
//   int hdr[8]  : [0] pending  [1] n  [2] #Pos  [3] #0-dim points  [4] #PD_up points  [5] tree has unreached nodes
//   f64 f[NM] | u32 par[NM] | u32 key[NM] | u32 pts[NM] | u64 query[MM]   (query = (asc rank + 1) << 40 | p << 16 | q)
__host__ __device__ constexpr size_t handoff_bytes(int NM, int MM) { return al16(32 + (size_t)20 * NM + (size_t)8 * MM); }
struct Handoff {
    int* hdr;
    double* f;
    unsigned *par, *key, *pts;
    ull* query;
};
__device__ __forceinline__ Handoff carve_handoff(unsigned char* slot, int NM) {
    Handoff H;
    H.hdr = (int*)slot;
    H.f = (double*)(slot + 32);
    H.par = (unsigned*)(slot + 32 + (size_t)8 * NM);
    H.key = H.par + NM;
    H.pts = H.key + NM;
    H.query = (ull*)(slot + 32 + (size_t)20 * NM);
    return H;
}

// The parallel half of Accelerate_PD, then everything the serial half and the image need goes to the hand-off slot.
// finb != null: the subgraph is meant for tlc_pd_dc_kernel (hdr[6] = 1): its descending ties are fixed and the record carries,
// per tree edge (bit 31 of par) and per query (bit 32), whether the edge is in the ascending pass's spanning tree.
template <int W, typename idx_t>
__device__ __forceinline__ void ext1_handoff(Mem<idx_t>& M, int n, int MMcap, int NMcap, unsigned char* slot, const unsigned* finb) {
    const int tid = threadIdx.x;
    const SwapTables T = carve_swap(M.keyS, NMcap);
    const bool any_unreached = ext1_build_tree<W>(M, T, n, MMcap, NMcap);
    const Handoff H = carve_handoff(slot, NMcap);
    const int npos = M.ctl[3], np0 = M.ctl[2];
    auto fin_of = [&](unsigned rank) -> unsigned { return finb ? ((finb[rank >> 5] >> (rank & 31)) & 1u) : 0u; };
    for (int i = tid; i < n; i += W) {
        const unsigned k = T.key[i];
        H.f[i] = M.f[i];
        H.par[i] = T.par[i] | ((k != 0u && fin_of((k >> 8) - 1u)) ? 0x80000000u : 0u);
        H.key[i] = k;
    }
    for (int i = tid; i < np0; i += W) H.pts[i] = M.pts[i];
    for (int k = tid; k < npos; k += W) {
        const unsigned e = M.pn[k];
        const unsigned ar = M.arank[e];
        H.query[k] = ((ull)(((ar + 1u) << 8) | fin_of(ar)) << 32) | (ull)M.dir[e];
    }
    if (tid == 0) {
        H.hdr[1] = n; H.hdr[2] = npos; H.hdr[3] = np0; H.hdr[4] = M.ctl[6]; H.hdr[5] = any_unreached ? 1 : 0;
        H.hdr[6] = (finb != nullptr && !any_unreached) ? 1 : 0;
        H.hdr[0] = 1;
    }
    __syncthreads();
}

// Accelerate_PD (accelerated_PD.py:115-178) on a subgraph whose Pos / Neg lists sit in M.pn.  Requires ctl[4] (#Neg) >= 1.
// dc: the descending order has its ties fixed and M.rec holds the spanning-tree bits of the ascending pass (pd_all_stages):
// try the divide-and-conquer form (ext1_dc.h); the serial walk below is the fallback and the form for few Pos edges.
template <int W, typename idx_t, class Sink>
__device__ __forceinline__ void ext1_stage(Mem<idx_t>& M, Sink& sink, int n, unsigned flags, int MMcap, int NMcap, ull* pc,
                                           ull& t_prev, ull* ph, bool dc) {
    if (dc) {
        const int K = M.ctl[3];
        const size_t hin_bytes = al16((size_t)K * 2);
        unsigned short* hin = (unsigned short*)(M.xbase + (M.xbytes - hin_bytes));
        const LdsSrc src{M.pn, M.dir, M.arank, (const unsigned*)M.rec, MMcap};
        const bool ok = M.xbytes > hin_bytes &&
                        ext1_dc_solve<W>(src, n, K, M.ctl[4], M.ctl, M.wcnt, M.xbase, M.xbytes - hin_bytes, hin);
        if (threadIdx.x == 0 && M.stats) atomicAdd(&M.stats[ok ? 1 : 3], 1ull);
#ifdef TLC_DC_VERIFY
        if (ok && (size_t)(M.ctl[2] + K) * 4 + (size_t)K * 2 + 16 <= (size_t)(MMcap + 2) * 4) {
            // development check: the serial walk on the same input, answer for answer (mismatches are counted in stats[3])
            unsigned short* v_hin = (unsigned short*)(M.pts + M.ctl[2] + K + 2);
            VerifySink vs{v_hin};
            unsigned short keep[8];
            for (int q = 0; q < 8; ++q) { const int k = (int)threadIdx.x + q * W; keep[q] = k < K ? hin[k] : 0; }
            __syncthreads();
            const SwapTables Tv = carve_swap(M.keyS, NMcap);
            const bool anyu = ext1_build_tree<W>(M, Tv, n, MMcap, NMcap);
            if (threadIdx.x < 64) {
                QueryLds qv{M.pn, M.dir, M.arank, K, 0u, 0u, 0u};
                ext1_walk(Tv, (ull*)M.rec, qv, K, NMcap, anyu, vs, M.f, true, 0, nullptr);
            }
            __syncthreads();
            int mism = 0;
            for (int q = 0; q < 8; ++q) { const int k = (int)threadIdx.x + q * W; if (k < K) { mism += (keep[q] != v_hin[k]); hin[k] = keep[q]; } }
            if (mism && M.stats) atomicAdd(&M.stats[3], (unsigned long long)mism);
            __syncthreads();
        }
#endif
        if (ok) {
            const bool keep0 = (flags & TLC_KEEP_ZERO_PERS) != 0;
            const int out0 = M.ctl[2];
            __syncthreads();                       // (every wavefront has read out0 before thread 0 moves the counter)
            if (!(Sink::is_global && !keep0)) {
                for (int k = (int)threadIdx.x; k < K; k += W)
                    sink.one_at(M.f, out0 + k, k, (int)(M.dir[M.pn[k]] >> 16), (int)hin[k]);
                if (threadIdx.x == 0) { M.ctl[2] = out0 + K; M.ctl[8] = K; }
            } else {
                // the TLC fork drops low_value >= large_value (:164): deterministic compaction in query order
                unsigned short* pos = (unsigned short*)M.xbase;
                for (int k = (int)threadIdx.x; k < K; k += W)
                    pos[k] = (unsigned short)(M.f[hin[k]] > M.f[M.dir[M.pn[k]] >> 16] ? 1 : 0);
                __syncthreads();
                const int total = block_exscan_u16<W>(pos, K, M.wcnt);
                for (int k = (int)threadIdx.x; k < K; k += W) {
                    const int p = (int)(M.dir[M.pn[k]] >> 16);
                    if (M.f[hin[k]] > M.f[p]) sink.one_at(M.f, out0 + pos[k], pos[k], p, (int)hin[k]);
                }
                if (threadIdx.x == 0) { M.ctl[2] = out0 + total; M.ctl[8] = total; }
            }
            __syncthreads();
            TLC_STAMP(10);
            return;
        }
        __syncthreads();
    }
    // HUGE tier: everything of the subgraph lives in HBM, and the walk below is a chain of dependent reads and exchanges on the
    // per-node tables -- 4.7 of the 6.8 ms of a 4 000-node vicinity with ~1 800 Pos edges went there at L2 latency.  The tables (28
    // bytes per node) and the path records take the workgroup's LDS when they fit (n <= ~5 200 with 144 KB), sized for n itself.
    const bool t_lds = M.lds_swap != nullptr && al16(swap_table_bytes(n)) + 1280 <= M.lds_swap_bytes;
    const int NMs = t_lds ? n : NMcap;
    const SwapTables T = carve_swap(t_lds ? (void*)M.lds_swap : (void*)M.keyS, NMs);
    ull* recs = t_lds ? (ull*)(M.lds_swap + al16(swap_table_bytes(n))) : (ull*)M.rec;
    const bool any_unreached = ext1_build_tree<W>(M, T, n, MMcap, NMs);
    if (t_lds) {
        // (and the queries: Pos list -> edge -> endpoints / rank are three dependent reads per query in HBM, two queries of
        // look-ahead do not cover them.  Packed by all threads into the sort keys' region, which the tables no longer use, the walk
        // reads them 64 at a time -- the form tlc_pd_swap_kernel takes them in.)
        ull* qg = M.keyS;
        const int npos = M.ctl[3];
        for (int k = threadIdx.x; k < npos; k += W) {
            const unsigned e = M.pn[k];
            qg[k] = ((ull)((M.arank[e] + 1u) << 8) << 32) | (ull)M.dir[e];
        }
        __syncthreads();
        TLC_STAMP(9);
        if (threadIdx.x < 64) {
            QueryGlobal qs{qg, npos, 0u, 0u, 0u, 0u, 0u, 0u};
            const int out0 = M.ctl[2];
            const int n_out = ext1_walk(T, recs, qs, npos, NMs, any_unreached, sink, M.f, (flags & TLC_KEEP_ZERO_PERS) != 0, out0, pc);
            if (threadIdx.x == 0) { M.ctl[2] = out0 + n_out; M.ctl[8] = n_out; }
        }
        __syncthreads();
        TLC_STAMP(10);
        return;
    }
    TLC_STAMP(9);
    if (threadIdx.x < 64) {                                          // first wavefront
        QueryLds qs{M.pn, M.dir, M.arank, M.ctl[3], 0u, 0u, 0u};
        const int out0 = M.ctl[2];
        const int n_out = ext1_walk(T, recs, qs, M.ctl[3], NMs, any_unreached, sink, M.f,
                                    (flags & TLC_KEEP_ZERO_PERS) != 0, out0, pc);
        if (threadIdx.x == 0) { M.ctl[2] = out0 + n_out; M.ctl[8] = n_out; }
    }
    __syncthreads();
    TLC_STAMP(10);
}

// All PD stages on a subgraph whose f[0..n) is final and whose m undirected edges sit in M.dir[0..m) as node-id pairs.
// `slot` != null: a subgraph with Pos edges leaves its cycle swap to tlc_pd_swap_kernel (deferred = true).
template <int W, int QS = 8, typename idx_t, class Sink>
__device__ __forceinline__ int pd_all_stages(Mem<idx_t>& M, Sink& sink, int n, int m, unsigned flags, int MMcap, int NMcap,
                                            ull* pc, ull& t_prev, ull* ph, unsigned char* slot, bool& deferred,
                                            int dc_mode = 0, bool handoff_all = true) {
    relabel_by_rank<W>(M, n, m);
    sort_edges<W, idx_t, false, QS>(M, m);
    for (int pos = threadIdx.x; pos < m; pos += W) M.arank[M.valS[pos]] = (unsigned)pos;
    __syncthreads();
    TLC_STAMP(5);
    if (TLC_STOP_AFTER <= 5) return TLC_STOPPED;
    mst_pass<W, idx_t, false, true>(M, sink, n, m, flags);
    if (threadIdx.x == 0) sink.ext0(M.f, 0, n - 1);                  // [min, max]  (:110)
    __syncthreads();
    TLC_STAMP(6);
    if (TLC_STOP_AFTER <= 6) return TLC_STOPPED;
    // A connected vicinity with n - 1 edges is a tree (30 % of a PubMed batch): every edge is a Neg edge, there is no Pos
    // edge and no 1-dimensional point, and the descending pass would only add Rel1 points, which weigh 0 in the image.
    // (The batch path is entered for connected vicinities only; tlc_pd_from_filtration reports Rel1 and takes the long way.)
    if (!Sink::is_global && m == n - 1) return TLC_ST_OK;
    // Many Pos edges (m - n + 1 of them in a connected graph): the cycle swap runs as a divide and conquer (ext1_dc.h), which
    // needs the ascending pass's spanning tree per edge id -- kept in M.rec, the serial walk's record area -- and the descending
    // order with its ties fixed
    // dc_mode 1: here, in this workgroup (LARGE: a CU to itself);  2: through the hand-off record, by tlc_pd_dc_kernel
    const int k_pos = m - n + 1;
    // (`elig` is a property of the subgraph alone: the tie fix-up below changes the order of equal descending keys, and a row must not
    // depend on whether this launch had a hand-off slot for the subgraph -- a speculative launch runs out of slots, another does not)
    bool elig = dc_mode != 0 && !(flags & TLC_NO_EXT1) && k_pos >= (W >= 512 ? TLC_DC_MIN_POS : TLC_DC_MIN_POS_SHARED) && m <= 8 * W &&
                m < 65536 && (size_t)((m + 31) / 32) * 4 <= 1280 &&
                (dc_mode == 2 || dc_bytes(k_pos, n > 2 * k_pos + 2 ? n : 2 * k_pos + 2) + al16((size_t)k_pos * 2) < M.xbytes);
    if (elig) {
        unsigned* finb = (unsigned*)M.rec;                            // the tree bitmap is indexed by ascending position = rank
        for (int w = threadIdx.x; w < (m + 31) / 32; w += W) finb[w] = M.tbits[w];
        __syncthreads();
    }
    sort_edges<W, idx_t, true, QS>(M, m);
    if (elig) elig = fix_desc_ties<W>(M, m);
    const bool dc = elig && (dc_mode == 1 || slot != nullptr);
    TLC_STAMP(7);
    if (TLC_STOP_AFTER <= 7) return TLC_STOPPED;
    if (Sink::want_down) mst_pass<W, idx_t, true, true>(M, sink, n, m, flags);
    else mst_pass<W, idx_t, true, false>(M, sink, n, m, flags);
    split_pos_neg<W>(M, m, MMcap);
    TLC_STAMP(8);
    if (TLC_STOP_AFTER <= 8) return TLC_STOPPED;
    int status = TLC_ST_OK;
    if (!(flags & TLC_NO_EXT1)) {
        if (M.ctl[4] == 0) status = TLC_ST_NO_TREE_EDGE;              // list(Nodes)[0] -> IndexError (:122)
        else if (slot != nullptr && M.ctl[3] > 0 && (handoff_all || (dc && dc_mode == 2 && M.ctl[3] == k_pos))) {
            ext1_handoff<W>(M, n, MMcap, NMcap, slot, (dc && dc_mode == 2 && M.ctl[3] == k_pos) ? (const unsigned*)M.rec : nullptr);
            deferred = true;
            TLC_STAMP(9);
        } else ext1_stage<W>(M, sink, n, flags, MMcap, NMcap, pc, t_prev, ph, dc && dc_mode == 1 && M.ctl[3] == k_pos);
    }
    return status;
}

// Standard normal CDF, _norm_cdf of PersistenceImager.pyx:54-60 = 0.5 * erfc(-x / sqrt 2).  The images of the batch path
// only ever ask for |x| <= 1.2 (births, persistences and grid lines all lie in [0, 1.2]); there the Maclaurin series of erf
// (19 terms, |z| <= 0.95, relative error 1.2e-15 against erfc) is a quarter of the device library's erfc, which is what
// the image stage of a small vicinity spends its time on.  Outside that range (BOUNDED = false: tlc_pi_raster, whose
// diagrams are the caller's): erfc.  The tier kernels instantiate the series alone -- erfc inlined beside it costs them
// registers they do not have.
template <bool BOUNDED>
__device__ __forceinline__ double tlc_norm_cdf(double x) {
    const double z = x * 0.70710678118654752440;
    if (BOUNDED || fabs(z) <= 0.95) {
        const double t = z * z;
        double a = 4.22140728880708822e-18;
        a = fma(a, t, -8.03273501241577328e-17);
        a = fma(a, t, 1.44832646435981379e-15);
        a = fma(a, t, -2.46682701026445706e-14);
        a = fma(a, t, 3.95542951645852569e-13);
        a = fma(a, t, -5.94779401363763541e-12);
        a = fma(a, t, 8.35070279514723971e-11);
        a = fma(a, t, -1.08922210371485731e-09);
        a = fma(a, t, 1.31225329638028058e-08);
        a = fma(a, t, -1.45038522231504685e-07);
        a = fma(a, t, 1.45891690009337058e-06);
        a = fma(a, t, -1.32275132275132281e-05);
        a = fma(a, t, 1.06837606837606838e-04);
        a = fma(a, t, -7.57575757575757575e-04);
        a = fma(a, t, 4.62962962962962937e-03);
        a = fma(a, t, -2.38095238095238082e-02);
        a = fma(a, t, 1.00000000000000006e-01);
        a = fma(a, t, -3.33333333333333315e-01);
        a = fma(a, t, 1.0);
        return fma(0.5 * 1.1283791670955126, z * a, 0.5);
    }
    return 0.5 * erfc(-z);
}

// PersistenceImager.transform (PersistenceImager.pyx:352-388) over points first..last: Gaussian sigma=1 on [0,1]^2,
// linear-ramp weight.  Phase A: one lane per (point, grid line) evaluates the normal CDF into an LDS table; phase B:
// one lane per pixel sums w * dPhi_b * dPhi_p over the points in diagram order (the factored form of the reference's
// 4-term inclusion-exclusion).  `get(k, b, d)` yields the k-th (birth, death).  Returns this thread's
// pixel value (threads >= res*res carry `acc` through unchanged).
// SW = the width the pixel sums are sliced for (S = SW / res^2 point slices per pixel, a point's slice = its index mod S, so the
// summation order -- and with it the last bit of the image -- depends on SW only, not on W or on the table size: the MID /
// MEDIUM tier kernels pass 64 so that a subgraph gives the same bits whether its image is made here or in tlc_pd_swap_kernel).
// RES: the resolution as a compile-time constant (0: `res` at run time).  The stage divides by 2 res + 3, res^2 and res per table entry
// and per thread; with the reference's resolution 5 known they are multiplications (pi_stage below picks the instance).
template <int W, bool BOUNDED, int SW, int RES, class Get>
__device__ __forceinline__ double pi_stage_impl(double* tbl, size_t table_bytes, Get get, int first, int last, int res_rt,
                                                double acc) {
    const int tid = threadIdx.x;
    const int res = RES > 0 ? RES : res_rt;
    const int G = res + 1, stride = 2 * G + 1, res2 = res * res;
    int batch = (int)(table_bytes / ((size_t)stride * 8));
    if (batch > W) batch = W;
    // phase B layout: S slices of the point list per pixel; thread = slice * res^2 + pixel
    const int S = SW / res2 > 0 ? SW / res2 : 1;
    const int sl = tid / res2, pix = tid - sl * res2;
    const int pi = pix / res, pj = pix - pi * res;
    const double acc_in = acc;
    acc = 0.0;
    const double pixel = 1.0 / (double)res;
    const double step = ((1.0 + pixel) - 0.0) / (double)(res + 1);       // _create_mesh (:302-314)
    for (int b0 = first; b0 < last; b0 += batch) {
        const int nb = (last - b0) < batch ? (last - b0) : batch;
        for (int t = tid; t < nb * stride; t += W) {
            const int pt = t / stride, g = t - pt * stride;
            double b, d;
            get(b0 + pt, b, d);
            const double pers = d - b;                                         // skew (:367-368)
            const double wgt = pers < 0.0 ? 0.0 : (pers > 1.0 ? 1.0 : pers);   // linear_ramp (:9-30)
            double val;
            if (g == 2 * G) val = wgt;
            else if (wgt == 0.0) val = 0.0;
            else {
                const double x = (g < G) ? ((double)g * step - b) : ((double)(g - G) * step - pers);
                val = tlc_norm_cdf<BOUNDED>(x);                                // _norm_cdf (:54-60)
            }
            tbl[t] = val;
        }
        __syncthreads();
        if (tid < res2 * S) {
            const int o = (b0 - first) % S;                            // slice of this batch's first point
            for (int pt = (sl - o + S) % S; pt < nb; pt += S) {
                const double* r = tbl + pt * stride;
                const double wgt = r[2 * G];
                if (wgt != 0.0) acc += wgt * ((r[pi + 1] - r[pi]) * (r[G + pj + 1] - r[G + pj]));
            }
        }
        __syncthreads();
    }
    if (S > 1) {
        // fold the S point-slices of every pixel in a fixed order (deterministic)
        if (tid < res2 * S) tbl[tid] = acc;
        __syncthreads();
        if (tid < res2) {
            double t = 0.0;
            for (int s = 0; s < S; ++s) t += tbl[s * res2 + tid];
            acc = t;
        }
        __syncthreads();
    }
    return acc + acc_in;
}
template <int W, bool BOUNDED, int SW, class Get>
__device__ __forceinline__ double pi_stage(double* tbl, size_t table_bytes, Get get, int first, int last, int res, double acc) {
#ifndef TLC_NO_RES5
    if (res == 5) return pi_stage_impl<W, BOUNDED, SW, 5>(tbl, table_bytes, get, first, last, res, acc);   // (uniform)
#endif
    return pi_stage_impl<W, BOUNDED, SW, 0>(tbl, table_bytes, get, first, last, res, acc);
}

}  // namespace

// (defined behind tlc_pd_swap_kernel: one subgraph of a divide-and-conquer list from its hand-off record)
template <int NM, int MM, int W>
__device__ __forceinline__ void dc_subgraph(const TlcPdParams& p, int wi, unsigned char* lds_raw);

// ======================================================================================================================
// Batch kernel: one workgroup per vicinity subgraph of one size tier.
// ======================================================================================================================
// PLAIN: the launch is the plain TLC-GNN image batch -- flags == 0, images at resolution 5, none of tlc_vicinity_filtration's outputs
// (tlc_launch_pd_tier checks): those parameters are constants of the instance, and every branch on a variant flag, the filtration /
// edge outputs and the divisions by the resolution drop out of the kernel the bench batch runs.
template <int NM, int MM, int W, bool LWL, bool HUGE, bool PLAIN = false>
#ifdef TLC_PHASE_DEBUG
__global__ __launch_bounds__(W) void tlc_pd_tier_kernel(TlcPdParams p) {       // (the counters need the registers)
#else
// 128 VGPRs for the SMALL and MEDWIDE tiers: four wavefronts per SIMD (16 resp. 4 workgroups per CU); 80 for the compact MEDIUM
// configuration: six (its 26 KB of LDS let six workgroups share a CU)
#ifndef TLC_M_WPE
#define TLC_M_WPE 4
#endif
__global__ __launch_bounds__(W, (W == 256 && !HUGE ? (NM == TLC_C_NMAX ? 6 : TLC_M_WPE) : (W <= 128 ? 4 : 1))) void tlc_pd_tier_kernel(TlcPdParams p) {
#endif
    typedef unsigned short idx_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if constexpr (PLAIN) {
        p.flags = 0u; p.res = 5; p.pi_enabled = 1;
        p.out_f = nullptr; p.out_n = nullptr; p.out_edges = nullptr; p.out_m = nullptr; p.ids_off = nullptr; p.edges_off = nullptr;
    }
    const int tid = threadIdx.x;
    unsigned char* base;
    Layout L;
    int NMr, MMr;
    if constexpr (HUGE) {
        NMr = p.huge_nmax; MMr = p.huge_mmax;
        L = make_layout(NMr, MMr, false, sizeof(idx_t), TLC_HUGE_MIN_TABLE);
        base = p.huge_scratch + (size_t)blockIdx.x * p.huge_stride;
    } else {
        NMr = NM; MMr = MM;
        constexpr Layout Lc = make_layout(NM, MM, LWL, sizeof(idx_t));
        L = Lc;
        base = lds_raw;
    }
    Mem<idx_t> M = carve<idx_t>(base, L, NMr, MMr, LWL);
    M.stats = p.stats;
    if constexpr (HUGE) { M.lds_swap = p.huge_lds > 0 ? lds_raw : nullptr; M.lds_swap_bytes = (size_t)p.huge_lds; }
    const int res = p.res, res2 = res * res;

    // Issue priority by LDS per wavefront (development A/B, make EXTRA="-DTLC_PRIO_LARGE=3 ..."; default 0 = none): in a pipelined
    // batch a tier costs LDS capacity x time, so a wavefront that holds much LDS should get through first.  Measured
    // (tools/gpu_build_ab3.sh, TINY 3 / LARGE 3 / MEDIUM 2 / MID 1 / SMALL 1 and subsets, four builds in turn): 0.649 - 0.664 ms per
    // pipelined batch whatever the setting -- the SIMDs are not contended enough for issue arbitration to matter
#ifndef TLC_PRIO_LARGE
#define TLC_PRIO_LARGE 0
#endif
#ifndef TLC_PRIO_MEDIUM
#define TLC_PRIO_MEDIUM 0
#endif
#ifndef TLC_PRIO_MID
#define TLC_PRIO_MID 0
#endif
#ifndef TLC_PRIO_SMALL
#define TLC_PRIO_SMALL 0
#endif
    if constexpr (!HUGE) {
        constexpr int prio = (NM == TLC_L_NMAX) ? TLC_PRIO_LARGE : ((NM == TLC_M_NMAX || NM == TLC_C_NMAX) ? TLC_PRIO_MEDIUM : (NM == TLC_D_NMAX ? TLC_PRIO_MID : TLC_PRIO_SMALL));
        if constexpr (prio > 0) __builtin_amdgcn_s_setprio(prio);
    }
    // this workgroup is resident: tell the launcher's gate (api.hip, tlc_wait_started)
    if (p.abort_flag && *p.abort_flag) return;
    int tier_count = p.tier_count;
    if (p.tier_count_dev) { const int c = *p.tier_count_dev; tier_count = c < tier_count ? c : tier_count; }
    // (both launches of a split LARGE tier count: the gate's target is the list's length, reached as soon as the first one is resident)
    if ((NM == TLC_L_NMAX) && !HUGE && p.started && tid == 0 && (int)blockIdx.x < tier_count) atomicAdd(p.started, 1);
    // HUGE: the workgroups stride over the list (one scratch slot each).  The LDS tiers: ONE subgraph per workgroup, list position
    // wi_base + blockIdx.x -- no loop around the body, so that nothing of it is hoisted in front of it and kept in registers
    // across all of its phases (that was the tiers' register spilling: thread-id arithmetic and image constants of every phase,
    // live from the first line on); a launch with fewer workgroups than subgraphs is completed by a second launch (api.hip).
    int wi = HUGE ? (int)blockIdx.x : p.wi_base + (int)blockIdx.x;
    if (wi >= tier_count) return;
    do {
        const int i = wi < p.n_hi ? p.tier_list_hi[wi] : p.tier_list[wi - p.n_hi];
        // hand-off slot of this subgraph (tiers whose cycle swap runs in tlc_pd_swap_kernel); "nothing pending" until decided
        unsigned char* slot = (!HUGE && p.handoff && wi < p.handoff_cap) ? p.handoff + (size_t)wi * (size_t)p.handoff_stride : nullptr;
        bool deferred = false;
        if (slot && tid == 0) *(int*)slot = 0;
        if (i < 0) continue;              // (a slot of the early arena whose vicinity turned out not to be of this tier: extract.hip)
        const int n = p.hdr_n[i], m2 = p.hdr_m2[i], lu = p.hdr_lu[i], lv = p.hdr_lv[i];
#ifndef TLC_NO_SIZE_ASSUME
        // (the scan bins a vicinity by these sizes: told to the compiler, a `for (k = tid; k < n; k += W)` of a tier with NM <= W is
        // an `if`, and the entry loops have a known maximum trip count)
        if constexpr (!HUGE) { __builtin_assume(n >= 1 && n <= NM); __builtin_assume(m2 >= 0 && m2 <= 2 * MM); }
#endif
        const long long eo = p.slot_entries ? (long long)wi * p.slot_entries : p.edge_off[i];
        const int m = m2 >> 1;
        const bool far = (lu < 0);        // u in S <=> v in S <=> d(u,v) <= hop  
        int status = TLC_ST_OK;
#ifdef TLC_PHASE_DEBUG
        ull* pc = p.phase_cycles;
        ull t_prev = pc ? clock64() : 0ull;
        ull ph[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        const ull t_begin = t_prev;
#else
        ull* pc = nullptr;
        ull t_prev = 0ull;
        ull* ph = nullptr;
#endif
        // ---- stage the subgraph ------------------------------------------------------------------------------------
        const bool in_small = (NM == TLC_S_NMAX && !HUGE && p.small_dir != nullptr);
        const unsigned* adir = in_small ? p.small_dir + (size_t)i * (2 * TLC_S_MMAX) : p.A_dir + eo;
        const double* alw = in_small ? p.small_lw + (size_t)i * (2 * TLC_S_MMAX) : p.A_lw + eo;
        for (int j = tid; j < m2; j += W) {
            M.dir[j] = adir[j];
            if (LWL) M.lw[j] = alw[j];
        }
        ull* du = (ull*)M.f;
        for (int k = tid; k < n; k += W) { du[k] = TLC_INF_BITS; M.dv[k] = TLC_INF_BITS; }
        if (tid == 0) M.ctl[1] = 0;
        __syncthreads();
        const double* glw = alw;
        auto LW = [&](int j) -> double { return LWL ? M.lw[j] : glw[j]; };
        // tiers whose weights stay in HBM/L2: Bellman-Ford keeps a thread's entries in registers, and the weight of every
        // node's tight successor entry goes to LDS beside its index (the Pos/Neg and point lists are not live yet), so that
        // neither the rounds nor the chain walks wait for global memory
        constexpr int BF_CE = (!LWL && !HUGE) ? (2 * MM + W - 1) / W : 0;
        constexpr bool WTAB = !LWL && !HUGE;
        // image slicing: the tiers that may hand a subgraph to tlc_pd_swap_kernel slice like that kernel does
        constexpr int PSW = (!HUGE && (NM == TLC_D_NMAX || NM == TLC_M_NMAX || NM == TLC_C_NMAX)) ? 64 : W;
        // (the tight-successor weight tables alias pn / pts: make_layout gives both NM doubles at least)
        double* wU = (double*)M.pn;
        double* wV = (double*)M.pts;

        const unsigned desc = p.flags & TLC_DESC_MASK;
        if (TLC_STOP_AFTER <= 1) status = TLC_STOPPED;
        if (status == TLC_ST_OK && !far) {
            // ---- P5: filtration.build_fv, weighted branch (riccidist2dgm.py:20-61); descriptor by flag ---------------------
            if (tid == 0) { du[lu] = 0ull; M.dv[lv] = 0ull; }
            __syncthreads();
            TLC_STAMP(0);
            bellman_ford<W, true, BF_CE>(du, M.dv, M.dir, m2, n, LW, M.ctl);
            TLC_STAMP(1);
            if (TLC_STOP_AFTER <= 2) status = TLC_STOPPED;
            // assert one connected component (:318): everything must be reachable from u
            double dmx = 0.0;
            int unreach = 0;
            for (int k = tid; k < n; k += W) {
                const ull a = du[k], b = M.dv[k];
                unreach |= (a == TLC_INF_BITS);
                const double da = __longlong_as_double((long long)a), db = __longlong_as_double((long long)b);
                if (a != TLC_INF_BITS && da > dmx) dmx = da;
                if (b != TLC_INF_BITS && db > dmx) dmx = db;
            }
            dmx = block_max<W>(dmx, M.red);
            const double un = block_max<W>(unreach ? 1.0 : 0.0, M.red);
            const bool sentinel = (p.flags & TLC_UNREACHABLE_100) != 0;     // data_utils_LP.py:41-49: NetworkXNoPath -> 100
            if (un != 0.0 && !sentinel) status = TLC_ST_DISCONNECTED;
            if (status == TLC_ST_OK) {
                // tight entries: a -> b lies on a (near-)shortest path from a to the root
                const double tol = 1e-10 * (1.0 + dmx);
                for (int k = tid; k < n; k += W) { M.cntU[k] = 0u; M.cntV[k] = 0u; }
                __syncthreads();
                for (int j = tid; j < m2; j += W) {
                    const unsigned e = M.dir[j];
                    const int a = e >> 16, b = e & 0xffffu;
                    const double w = LW(j);
                    if (a != lu) {
                        const double s = (w + __longlong_as_double((long long)du[b])) - __longlong_as_double((long long)du[a]);
                        if (s <= tol) { atomicAdd(&M.cntU[a], 1u); M.nxtU[a] = (unsigned)j; if (WTAB) wU[a] = w; }
                    }
                    if (a != lv) {
                        const double s = (w + __longlong_as_double((long long)M.dv[b])) - __longlong_as_double((long long)M.dv[a]);
                        if (s <= tol) { atomicAdd(&M.cntV[a], 1u); M.nxtV[a] = (unsigned)j; if (WTAB) wV[a] = w; }
                    }
                }
                __syncthreads();
                // chain walks: sum the weights from x towards the root, left to right (:29-30, :34-35).
                // f aliases du, which is dead from here on: the walks only read cnt/nxt/dir/lw.
                double nrm_min = 0.0;                  // descriptor 'min': this thread's share of max_S max(d1, d2)
                for (int x0 = 0; x0 < n; x0 += W) {
                    const int x = x0 + tid;
                    double fr = 0.0;
                    bool amb = false;
                    if (x < n && x != lu && x != lv) {
                        double d1 = 0.0, d2 = 0.0;
                        int a = x, steps = 0;
                        const bool no_u = (du[x] == TLC_INF_BITS), no_v = (M.dv[x] == TLC_INF_BITS);   // only with `sentinel`
                        if (no_u) { d1 = 100.0; a = lu; }
                        while (a != lu) {
                            if (M.cntU[a] != 1u || ++steps > n) { amb = true; break; }
                            const int j = (int)M.nxtU[a];
                            d1 = d1 + (WTAB ? wU[a] : LW(j));
                            a = (int)(M.dir[j] & 0xffffu);
                        }
                        a = x; steps = 0;
                        if (no_v) { d2 = 100.0; a = lv; }
                        if (desc == TLC_DESC_ROOT1) a = lv;               // single root: no second distance
                        while (!amb && a != lv) {
                            if (M.cntV[a] != 1u || ++steps > n) { amb = true; break; }
                            const int j = (int)M.nxtV[a];
                            d2 = d2 + (WTAB ? wV[a] : LW(j));
                            a = (int)(M.dir[j] & 0xffffu);
                        }
                        // 'sum' = dist_1 + dist_2, 'min', 'max' (:47-49); single root: dist_1 (data_utils_NC.py:46)
                        fr = desc == 0u ? d1 + d2 : (desc == TLC_DESC_MIN ? (d2 < d1 ? d2 : d1) : (desc == TLC_DESC_MAX ? (d2 > d1 ? d2 : d1) : d1));
                        // 'min' is normalised by the maximum of 'max' (:51), not by its own
                        if (desc == TLC_DESC_MIN && !amb) { const double dm = d2 > d1 ? d2 : d1; nrm_min = dm > nrm_min ? dm : nrm_min; }
                    }
                    if (x < n) {
                        if (amb) M.amb[atomicAdd(&M.ctl[1], 1)] = (idx_t)x;
                        else M.f[x] = fr;
                    }
                }
                __syncthreads();
                TLC_STAMP(2);
                // exact fallback for sources with (near-)tied paths: Bellman-Ford sourced at x itself
                const int namb = M.ctl[1];
                for (int q = 0; q < namb; ++q) {
                    const int x = M.amb[q];
                    for (int k = tid; k < n; k += W) M.dv[k] = TLC_INF_BITS;
                    __syncthreads();
                    if (tid == 0) M.dv[x] = 0ull;
                    __syncthreads();
                    bellman_ford<W, false, 0>(M.dv, M.dv, M.dir, m2, n, LW, M.ctl);
                    if (tid == 0) {
                        const double e1 = M.dv[lu] == TLC_INF_BITS ? 100.0 : __longlong_as_double((long long)M.dv[lu]);
                        const double e2 = M.dv[lv] == TLC_INF_BITS ? 100.0 : __longlong_as_double((long long)M.dv[lv]);
                        M.f[x] = desc == 0u ? e1 + e2 : (desc == TLC_DESC_MIN ? (e2 < e1 ? e2 : e1) : (desc == TLC_DESC_MAX ? (e2 > e1 ? e2 : e1) : e1));
                        if (desc == TLC_DESC_MIN) { const double dm = e2 > e1 ? e2 : e1; nrm_min = dm > nrm_min ? dm : nrm_min; }
                    }
                    __syncthreads();
                }
                if (namb && tid == 0 && p.stats) atomicAdd(&p.stats[0], (ull)namb);
                TLC_STAMP(3);
                // normalise (:50-56): plain division by the maximum
                if (!(p.flags & TLC_NO_NORM)) {                           // (norm=False: the raw distances stand)
                    double mx = 0.0;
                    if (desc == TLC_DESC_MIN) mx = nrm_min;                   // norm_scaler = max of 'max' (:51)
                    else for (int k = tid; k < n; k += W) mx = M.f[k] > mx ? M.f[k] : mx;
                    mx = block_max<W>(mx, M.red);
                    double scaler = mx;
                    if (p.flags & TLC_NORM_EPS) scaler = mx + 1e-10;          // data_utils_LP.py:64
                    else if (mx == 0.0) status = TLC_ST_ZERO_RANGE;           // ZeroDivisionError (:54)
                    if (status == TLC_ST_OK)
                        for (int k = tid; k < n; k += W) M.f[k] = M.f[k] / scaler;
                }
                __syncthreads();
            }
        } else if (status == TLC_ST_OK) {
            // d(u,v) > hop: every distance is the sentinel 100 (:31-37) => f == 200/200; only connectivity matters
            if (tid == 0) du[0] = 0ull;
            __syncthreads();
            bellman_ford<W, false, 0>(du, du, M.dir, m2, n, LW, M.ctl);
            int unreach = 0;
            for (int k = tid; k < n; k += W) unreach |= (du[k] == TLC_INF_BITS);
            const double un = block_max<W>(unreach ? 1.0 : 0.0, M.red);
            if (un != 0.0) status = TLC_ST_DISCONNECTED;
            __syncthreads();
            // both sentinels: 'sum' = 200, 'min' = 'max' = 100 (:31-37,47-49), then the same division as above
            const double raw = desc == 0u ? 200.0 : 100.0;
            const double one = (p.flags & TLC_NO_NORM) ? raw : ((p.flags & TLC_NORM_EPS) ? (raw / (raw + 1e-10)) : (raw / raw));
            for (int k = tid; k < n; k += W) M.f[k] = one;
            __syncthreads();
        }
        if (TLC_STOP_AFTER <= 3 && status == TLC_ST_OK) status = TLC_STOPPED;
        // optional filtration output (tlc_vicinity_filtration)
        if (p.out_f) {
            const long long no = p.ids_off[i];
            const long long cap = p.ids_off[i + 1] - no;
            if (n <= cap) {
                for (int k = tid; k < n; k += W) p.out_f[no + k] = (status == TLC_ST_OK) ? M.f[k] : 0.0;
                if (tid == 0) p.out_n[i] = n;
            } else if (tid == 0) p.out_n[i] = -n;
        }
        const bool want_edges = (p.out_edges != nullptr);
        if (status == TLC_ST_OK && ((p.pi_enabled && !far) || want_edges)) {
        // ---- undirected edge list: directed entries with src < dst, in CSR order (deterministic compaction) ------
            {
                int run = 0;
                for (int j0 = 0; j0 < m2; j0 += W) {
                    const int j = j0 + tid;
                    unsigned e = 0;
                    bool keep = false;
                    if (j < m2) { e = M.dir[j]; keep = (e >> 16) < (e & 0xffffu); }
                    const ull mk = __ballot(keep);
                    int off;
                    if (W == 64) {
                        off = run + __popcll(mk & tlc_lanemask_lt());
                        run += __popcll(mk);
                    } else {
                        if (tlc_lane() == 0) M.wcnt[tid >> 6] = __popcll(mk);
                        __syncthreads();
                        int before = 0, tot = 0;
#pragma unroll
                        for (int k = 0; k < W / 64; ++k) {
                            const int c = M.wcnt[k];
                            if (k < (tid >> 6)) before += c;
                            tot += c;
                        }
                        off = run + before + __popcll(mk & tlc_lanemask_lt());
                        run += tot;
                        __syncthreads();
                    }
                    // in place: off <= j, and every lane of this chunk has read its entry already
                    if (keep) M.dir[off] = e;
                }
                __syncthreads();
            }
        }
        if (TLC_STOP_AFTER <= 4 && status == TLC_ST_OK) status = TLC_STOPPED;
        if (want_edges) {
            const long long eo2 = p.edges_off[i];
            const long long ecap = p.edges_off[i + 1] - eo2;
            if (status == TLC_ST_OK && m <= ecap) {
                for (int e = tid; e < m; e += W) {
                    const unsigned ab = M.dir[e];
                    p.out_edges[2 * (eo2 + e)] = (int)(ab >> 16);
                    p.out_edges[2 * (eo2 + e) + 1] = (int)(ab & 0xffffu);
                }
                if (tid == 0) p.out_m[i] = m;
            } else if (tid == 0) p.out_m[i] = status == TLC_ST_OK ? -m : 0;
        }
        double acc = 0.0;
        if (p.pi_enabled && status == TLC_ST_OK && far) {
            // constant f: no strict pair, [1,1] has persistence 0 => weight 0 => exact zero image;
            // IndexError for a single node (accelerated_PD.py:122)
            if (n == 1 && !(p.flags & TLC_NO_EXT1)) status = TLC_ST_NO_TREE_EDGE;
        } else if (p.pi_enabled && status == TLC_ST_OK) {
            if (tid == 0) { M.ctl[2] = 0; M.ctl[6] = 0; M.ctl[7] = 0; M.ctl[8] = 0; }
            __syncthreads();
            PtsSink sink{M.pts, M.ctl};
            TLC_STAMP(4);
            // (the compact LARGE kernels sort like the wide ones -- same run / merge decisions, so a vicinity's row does not depend on which of
            // the two took it)
            status = pd_all_stages<W, sort_hold(MM, W)>(M, sink, n, m, p.flags, MMr, NMr, pc, t_prev, ph, slot, deferred,
                                      /*dc_mode=*/(!HUGE && (NM == TLC_L_NMAX)) ? TLC_DC_LARGE_MODE
                                                  : ((!HUGE && NM == TLC_M_NMAX) ? 2 : 0),     // (whether or not the launch carries a
                                      // dc list: marking a subgraph for the divide and conquer fixes the order of its tied descending
                                      // keys, and a row must not depend on whether tlc_pd_dc_kernel or the serial walk then answers)
                                      /*handoff_all=*/NM != TLC_L_NMAX);
            bool dc_here = false;
            if constexpr (!HUGE && NM == TLC_L_NMAX) dc_here = p.dc_inplace != 0;
            if (deferred && slot && !dc_here && tid == 0 && p.dc_count && ((const int*)slot)[6] != 0) {
                const int li = atomicAdd(p.dc_count, 1);              // meant for tlc_pd_dc_kernel
                p.dc_list[li] = wi;
            }
            if constexpr (!HUGE && NM == TLC_L_NMAX) {
                // the divide and conquer of this subgraph by this workgroup, from the record it has just written (ext1_handoff ends
                // with a barrier; the layout of tlc_pd_dc_kernel fits the tier's LDS): no second launch, no second placement
                if (deferred && slot && dc_here && ((const int*)slot)[6] != 0) dc_subgraph<NM, MM, W>(p, wi, lds_raw);
            }
            if (TLC_STOP_AFTER >= 10 && status == TLC_ST_OK && !deferred) {
                const int np = M.ctl[2], n_up = M.ctl[6];
                auto get = [&](int k, double& b, double& d) {
                    const unsigned bd = M.pts[k];
                    b = M.f[bd >> 16];
                    d = M.f[bd & 0xffffu];
                };
                // transform(np.array(PD_zero + PD_one)) (riccidist2dgm.py:327-328); Rel1 and [max,min] have negative
                // persistence => weight 0 (PersistenceImager.pyx:23-24), so they are never materialised here
                if (p.flags & TLC_PI_ORD0_EXT1) {
                    acc = pi_stage<W, true, PSW>((double*)M.table, M.table_bytes, get, 0, n_up, res, acc);
                    acc = pi_stage<W, true, PSW>((double*)M.table, M.table_bytes, get, n_up + 1, np, res, acc);
                } else {
                    acc = pi_stage<W, true, PSW>((double*)M.table, M.table_bytes, get, 0, np, res, acc);
                }
            }
        }
        TLC_STAMP(11);
#ifdef TLC_PHASE_DEBUG
        if (pc && tid == 0) {
            const ull tot = clock64() - t_begin;
            // (one round of atomics per workgroup, behind its last phase: a stamp that adds to a global counter makes the phase behind it
            // pay for the atomic at its first barrier -- with 10 000 SMALL workgroups on twelve addresses that was most of what it showed)
            for (int k = 0; k < 12; ++k) if (ph[k]) atomicAdd(&pc[k], ph[k]);
            atomicAdd(&pc[12], tot);
            atomicAdd(&pc[14], 1ull);
            if (atomicMax(&pc[13], tot) < tot) {          // slowest workgroup so far: its own phase split, n and m
                for (int k = 0; k < 12; ++k) pc[16 + k] = ph[k];
                pc[28] = (ull)n; pc[29] = (ull)m;
            }
        }
#endif
        if (status != TLC_ST_OK) acc = 0.0;
        if (!deferred) {                  // (else tlc_pd_swap_kernel writes the image row and the status)
            if (p.out_pi && tid < res2) p.out_pi[(size_t)i * res2 + tid] = acc;
            if (p.out_status && tid == 0) p.out_status[i] = (unsigned char)status;
        }
        __syncthreads();
    } while (HUGE && (wi += (int)gridDim.x) < tier_count);
}

// ======================================================================================================================
// The serial half of a tier: cycle swap + image of the subgraphs a tier kernel handed off.  One wavefront per subgraph.
// ======================================================================================================================
struct SwapLayout {
    size_t o_rec, o_pts, o_ctl, total, table_bytes;
};
__host__ __device__ constexpr SwapLayout make_swap_layout(int NM, int MM) {
    SwapLayout L{};
    // [0, o_rec): the swap tables; afterwards f[NM] and the image table (64 points per round at res 5)
    // (capacities up to 1 024 edges: the compact tables of ext1_walk_c)
    size_t o = al16(smax((MM <= 1024 && TLC_SWAP_COMPACT) ? swap_table_c_bytes(NM) : swap_table_bytes(NM), (size_t)8 * NM + (size_t)64 * 13 * 8));
    L.table_bytes = o - (size_t)8 * NM;
    L.o_rec = o;  o += 1280;
    L.o_pts = o;  o += al16((size_t)(MM + 2) * 4);
    L.o_ctl = o;  o += 64;
    L.total = o;
    return L;
}

// The serial cycle swap + the image of the handed-off subgraph at list position wi; W threads (the walk itself is the first
// wavefront's), LDS as make_swap_layout(NM, MM).
template <int NM, int MM, int W>
__device__ __forceinline__ void swap_subgraph(const TlcPdParams& p, int wi, unsigned char* lds_raw) {
    constexpr SwapLayout L = make_swap_layout(NM, MM);
    const int tid = threadIdx.x;
    const Handoff H = carve_handoff(p.handoff + (size_t)wi * (size_t)p.handoff_stride, NM);
    if (H.hdr[0] == 0) return;                                        // finished by the tier kernel itself
    const int i = wi < p.n_hi ? p.tier_list_hi[wi] : p.tier_list[wi - p.n_hi];
    const int n = H.hdr[1], npos = H.hdr[2], np0 = H.hdr[3], n_up = H.hdr[4];
    const bool any_unreached = H.hdr[5] != 0;
    unsigned* pts = (unsigned*)(lds_raw + L.o_pts);
    int* ctl = (int*)(lds_raw + L.o_ctl);
#ifdef TLC_PHASE_DEBUG
    const ull t_begin = clock64();
#endif
    for (int k = tid; k < np0; k += W) pts[k] = H.pts[k];
    PtsSink sink{pts, ctl};
    if constexpr (MM <= 1024 && TLC_SWAP_COMPACT) {
        // compact tables (ext1_walk_c): (rank + 1) << 16 | parent per node, "not in the tree" = 0xffff; slot NM = the spare node above the root
        const SwapTablesC T = carve_swap_c(lds_raw, NM);
        unsigned* recs = (unsigned*)(lds_raw + L.o_rec);
        for (int k = tid; k < n; k += W) {
            const unsigned pr = H.par[k];
            T.pk[k] = ((H.key[k] >> 8) << 16) | (pr == 0xffffffffu ? 0xffffu : (pr & 0xffffu));
            T.slot[k] = 0ull;
        }
        if (tid == 0) { T.pk[NM] = (unsigned)NM; T.slot[NM] = 0ull; }
        __syncthreads();
        if (tid < 64) {
            QueryGlobal qs{H.query, npos, 0u, 0u, 0u, 0u, 0u, 0u};
            const int n_out = ext1_walk_c(T, recs, qs, npos, NM, any_unreached, sink, np0);
            if (tid == 0) ctl[2] = np0 + n_out;
        }
    } else {
        const SwapTables T = carve_swap(lds_raw, NM);
        ull* recs = (ull*)(lds_raw + L.o_rec);
        for (int k = tid; k < n; k += W) {
            const unsigned pr = H.par[k];
            T.par[k] = pr == 0xffffffffu ? pr : (pr & 0x7fffffffu);
            T.key[k] = H.key[k];
            T.mark[k] = 0u;
        }
        if (tid == 0) { T.par[NM] = (unsigned)NM; T.key[NM] = 0u; T.mark[NM] = 0u; }
        __syncthreads();
        if (tid < 64) {
            QueryGlobal qs{H.query, npos, 0u, 0u, 0u, 0u, 0u, 0u};
            const int n_out = ext1_walk(T, recs, qs, npos, NM, any_unreached, sink, (const double*)nullptr, false, np0, nullptr);
            if (tid == 0) ctl[2] = np0 + n_out;
        }
    }
    __syncthreads();
    const int np = ctl[2];
    // the tables are dead: f and the image table take their place
    double* f = (double*)lds_raw;
    for (int k = tid; k < n; k += W) f[k] = H.f[k];
    __syncthreads();
    auto get = [&](int k, double& b, double& d) {
        const unsigned bd = pts[k];
        b = f[bd >> 16];
        d = f[bd & 0xffffu];
    };
    const int res = p.res, res2 = res * res;
    double* table = (double*)(lds_raw + (size_t)8 * NM);
    double acc = 0.0;
    if (p.flags & TLC_PI_ORD0_EXT1) {
        acc = pi_stage<W, true, 64>(table, L.table_bytes, get, 0, n_up, res, acc);
        acc = pi_stage<W, true, 64>(table, L.table_bytes, get, n_up + 1, np, res, acc);
    } else {
        acc = pi_stage<W, true, 64>(table, L.table_bytes, get, 0, np, res, acc);
    }
    if (p.out_pi && tid < res2) p.out_pi[(size_t)i * res2 + tid] = acc;
    if (p.out_status && tid == 0) p.out_status[i] = (unsigned char)TLC_ST_OK;
#ifdef TLC_PHASE_DEBUG
    if (p.phase_cycles && tid == 0) { atomicAdd(&p.phase_cycles[10], clock64() - t_begin); atomicAdd(&p.phase_cycles[12], clock64() - t_begin); }
#endif
    __syncthreads();
}

template <int NM, int MM, bool PLAIN = false>
__global__ __launch_bounds__(64, 4) void tlc_pd_swap_kernel(TlcPdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if constexpr (PLAIN) { p.flags = 0u; p.res = 5; p.pi_enabled = 1; }           // (see tlc_pd_tier_kernel)
    if (p.abort_flag && *p.abort_flag) return;
    int tier_count = p.tier_count;
    if (p.tier_count_dev) { const int c = *p.tier_count_dev; tier_count = c < tier_count ? c : tier_count; }
    if (tier_count > p.handoff_cap) tier_count = p.handoff_cap;       // (list positions beyond it were not handed off)
    // (one subgraph per workgroup -- every launch has at least min(tier_count, handoff_cap) of them -- and no loop: see tlc_pd_tier_kernel)
    if ((int)blockIdx.x < tier_count) swap_subgraph<NM, MM, 64>(p, (int)blockIdx.x, lds_raw);
}

// ======================================================================================================================
// The cycle swap of the subgraphs with MANY Pos edges as a divide and conquer (ext1_dc.h), from the hand-off record, then the
// image.  One workgroup of W threads per subgraph; runs between a tier kernel and its tlc_pd_swap_kernel on the same stream.
// A record it finishes is marked "nothing pending"; whatever it does not take (few Pos edges, not marked for it, does not fit
// its LDS budget, or -- rounding of nearly equal keys -- the ranks turn out not to be an MST order) stays for the serial kernel.
// ======================================================================================================================
__host__ __device__ constexpr size_t dc_kernel_lds(int NM, int MM) {
    // LDS budget: the solver's arrays for a subgraph of the tier's typical heavy shape (K = MM/2 queries on NM nodes), the
    // answers, the points; f and the image table re-use the solver's arrays afterwards.  The wide MEDIUM configuration: K = 7/8 MM --
    // what it gets are dense little vicinities, 600 Pos edges on 80 nodes (52 KB; there are a few dozen of them in a batch)
    const int KB = NM <= TLC_M_NMAX ? (MM / 8) * 7 : MM / 2;
    const size_t dc = al16(dc_bytes(KB, NM > 2 * KB + 2 ? NM : 2 * KB + 2) + (size_t)MM + 16) + al16((size_t)(MM + 2) * 4) + 256;
    return dc > make_swap_layout(NM, MM).total ? dc : make_swap_layout(NM, MM).total;     // (the serial fallback's layout fits too)
}
// One subgraph (position wi of the launch's list) from its hand-off record: the divide and conquer, its points, the image; what the
// solver cannot take goes through the serial walk here.  Called by tlc_pd_dc_kernel, and by the LARGE tier kernel IN PLACE
// (TlcPdParams::dc_inplace, round 5: the workgroup that wrote the record re-carves its own LDS and carries on -- the separate
// launch had to find 71 CUs with 107 KB of free LDS again while the other tiers' kernels were being placed: 262 -> 523 us for one
// batch alone once the extraction in front of them got shorter).  lds_raw: at least dc_kernel_lds(NM, MM) bytes.
template <int NM, int MM, int W>
__device__ __forceinline__ void dc_subgraph(const TlcPdParams& p, int wi, unsigned char* lds_raw) {
    constexpr size_t total = dc_kernel_lds(NM, MM);
    constexpr size_t o_pts = total - 256 - al16((size_t)(MM + 2) * 4), o_ctl = total - 256;
    const int tid = threadIdx.x;
    unsigned* pts = (unsigned*)(lds_raw + o_pts);
    int* ctl = (int*)(lds_raw + o_ctl);               // 16 ints, then 32 ints for block scans
    int* wcnt = ctl + 16;
    do {
        const Handoff H = carve_handoff(p.handoff + (size_t)wi * (size_t)p.handoff_stride, NM);
        if (H.hdr[0] == 0 || H.hdr[6] == 0) continue;                 // finished by the tier kernel / not meant for this kernel
        const int i = wi < p.n_hi ? p.tier_list_hi[wi] : p.tier_list[wi - p.n_hi];
        const int n = H.hdr[1], K = H.hdr[2], np0 = H.hdr[3], n_up = H.hdr[4];
        const size_t hin_bytes = al16((size_t)K * 2);
        // what the divide and conquer cannot take (does not fit, or the ranks are no MST order): the serial walk, here
        if (hin_bytes + 16 >= o_pts || np0 + K > MM + 2) { __syncthreads(); swap_subgraph<NM, MM, W>(p, wi, lds_raw); continue; }
        unsigned short* hin = (unsigned short*)(lds_raw + (o_pts - hin_bytes));
        const HandoffSrc src{H.par, H.key, H.query};
        __syncthreads();
#ifdef TLC_PHASE_DEBUG
        // (diagnostics build only: an array whose address is passed on lives in scratch, 80 bytes per lane)
        unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const unsigned long long t_begin = p.phase_cycles ? clock64() : 0ull;
        const bool ok = ext1_dc_solve<W>(src, n, K, n, ctl, wcnt, lds_raw, o_pts - hin_bytes, hin, p.phase_cycles ? prof : nullptr) && !p.dc_force_fail;
        if (p.phase_cycles && tid == 0) {                             // the slowest subgraph's split
            const unsigned long long tot = clock64() - t_begin;
            if (atomicMax(&p.phase_cycles[13], tot) < tot) {
                for (int k = 0; k < 6; ++k) p.phase_cycles[16 + k] = prof[k];
                p.phase_cycles[22] = tot; p.phase_cycles[28] = (unsigned long long)n; p.phase_cycles[29] = (unsigned long long)K;
            }
        }
#else
        const bool ok = ext1_dc_solve<W>(src, n, K, n, ctl, wcnt, lds_raw, o_pts - hin_bytes, hin, nullptr) && !p.dc_force_fail;
#endif
        if (!ok) {
            if (p.stats && tid == 0) atomicAdd(&p.stats[3], 1ull);
            __syncthreads();
            swap_subgraph<NM, MM, W>(p, wi, lds_raw);
            continue;
        }
        if (p.stats && tid == 0) atomicAdd(&p.stats[1], 1ull);
        for (int k = tid; k < np0; k += W) pts[k] = H.pts[k];
        for (int k = tid; k < K; k += W) pts[np0 + k] = (((unsigned)(H.query[k] >> 16) & 0xffffu) << 16) | (unsigned)hin[k];
        __syncthreads();
        // the solver's arrays are dead: f and the image table take their place (same slicing as tlc_pd_swap_kernel: same bits)
        double* f = (double*)lds_raw;
        for (int k = tid; k < n; k += W) f[k] = H.f[k];
        __syncthreads();
        auto get = [&](int k, double& b, double& d) {
            const unsigned bd = pts[k];
            b = f[bd >> 16];
            d = f[bd & 0xffffu];
        };
        const int res = p.res, res2 = res * res, np = np0 + K;
        double* table = (double*)(lds_raw + al16((size_t)8 * n));
        const size_t table_bytes = (o_pts - hin_bytes) - al16((size_t)8 * n);
        double acc = 0.0;
        if (p.flags & TLC_PI_ORD0_EXT1) {
            acc = pi_stage<W, true, 64>(table, table_bytes, get, 0, n_up, res, acc);
            acc = pi_stage<W, true, 64>(table, table_bytes, get, n_up + 1, np, res, acc);
        } else {
            acc = pi_stage<W, true, 64>(table, table_bytes, get, 0, np, res, acc);
        }
        if (p.out_pi && tid < res2) p.out_pi[(size_t)i * res2 + tid] = acc;
        if (tid == 0) {
            if (p.out_status) p.out_status[i] = (unsigned char)TLC_ST_OK;
            H.hdr[0] = 0;                                             // nothing pending for tlc_pd_swap_kernel
        }
        __syncthreads();
    } while (false);
}

template <int NM, int MM, int W>
__global__ __launch_bounds__(W, (W <= 256 ? 4 : 1)) void tlc_pd_dc_kernel(TlcPdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    if (p.abort_flag && *p.abort_flag) return;
    int tier_count = p.tier_count;
    if (p.tier_count_dev) { const int c = *p.tier_count_dev; tier_count = c < tier_count ? c : tier_count; }
    if (tier_count > p.handoff_cap) tier_count = p.handoff_cap;
    // a chain of short barrier-separated phases: its wavefronts should win issue arbitration against the throughput kernels
    // of the other tiers that share the CU
    __builtin_amdgcn_s_setprio(3);
    const int n_list = p.dc_count ? *p.dc_count : 0;
    // (one subgraph per workgroup and no loop around the body, like the tier kernels: see tlc_pd_tier_kernel)
    const int li = blockIdx.x;
    if (li >= n_list) return;
    const int wi = p.dc_list[li];
    if (wi < 0 || wi >= tier_count) return;
    dc_subgraph<NM, MM, W>(p, wi, lds_raw);
}

// ======================================================================================================================
// tlc_pd_from_filtration: caller-supplied graphs and filtration values; diagrams written out as values.
// ======================================================================================================================
template <int NM, int MM, int W, bool HUGE>
__global__ __launch_bounds__(W) void tlc_pdf_tier_kernel(TlcPdfParams p) {
    typedef unsigned short idx_t;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int tid = threadIdx.x;
    unsigned char* base;
    Layout L;
    int NMr, MMr;
    if constexpr (HUGE) {
        NMr = p.huge_nmax; MMr = p.huge_mmax;
        L = make_layout(NMr, MMr, false, sizeof(idx_t), TLC_HUGE_MIN_TABLE);
        base = p.huge_scratch + (size_t)blockIdx.x * p.huge_stride;
    } else {
        NMr = NM; MMr = MM;
        constexpr Layout Lc = make_layout(NM, MM, false, sizeof(idx_t));
        L = Lc;
        base = lds_raw;
    }
    Mem<idx_t> M = carve<idx_t>(base, L, NMr, MMr, false);
    // (no loop around the body for the LDS tiers, whose launches have one workgroup per graph: see tlc_pd_tier_kernel)
    int wi = blockIdx.x;
    if (wi >= p.count) return;
    do {
        const int g = p.list[wi];
        const long long no = p.node_offs[g], eo = p.edge_offs[g];
        const int n = (int)(p.node_offs[g + 1] - no), m = (int)(p.edge_offs[g + 1] - eo);
        for (int k = tid; k < n; k += W) M.f[k] = p.f[no + k];
        for (int e = tid; e < m; e += W)
            M.dir[e] = ((unsigned)p.edges[2 * (eo + e)] << 16) | (unsigned)p.edges[2 * (eo + e) + 1];
        if (tid == 0) { M.ctl[2] = 0; M.ctl[3] = 0; M.ctl[4] = 0; M.ctl[6] = 0; M.ctl[7] = 0; M.ctl[8] = 0; }
        __syncthreads();
        GlobalSink sink{p.pd_up + 2 * no, p.pd_down + 2 * no, p.pd_one + 2 * eo, p.ext0 + 2 * (size_t)g, M.ctl};
        ull* pc = nullptr;
        ull t_prev = 0;
        ull* ph = nullptr;
        if (m > 0) {
            bool deferred = false;
            pd_all_stages<W>(M, sink, n, m, p.flags, MMr, NMr, pc, t_prev, ph, nullptr, deferred,
                             /*dc_mode=*/(NM == TLC_L_NMAX && !HUGE) ? 1 : 0);
        } else if (tid == 0) {
            double mn = 99999999.0, mx = -99999999.0;
            for (int k = 0; k < n; ++k) { mn = M.f[k] < mn ? M.f[k] : mn; mx = M.f[k] > mx ? M.f[k] : mx; }
            sink.e0[0] = mn; sink.e0[1] = mx;
        }
        __syncthreads();
        const int npos = M.ctl[3], nneg = M.ctl[4];
        if (tid == 0) {
            p.counts[4 * (size_t)g + 0] = M.ctl[6];
            p.counts[4 * (size_t)g + 1] = M.ctl[7];
            p.counts[4 * (size_t)g + 2] = M.ctl[8];
            p.counts[4 * (size_t)g + 3] = n - nneg;
        }
        if (p.edge_rank) {
            for (int k = tid; k < npos; k += W) p.edge_rank[eo + M.pn[k]] = k;
            for (int k = tid; k < nneg; k += W) p.edge_rank[eo + M.pn[MMr - 1 - k]] = -k - 1;
        }
        __syncthreads();
    } while (HUGE && (wi += (int)gridDim.x) < p.count);
}

// tier binning for tlc_pd_from_filtration
__global__ void tlc_pdf_bin_kernel(int n_graphs, const long long* node_offs, const long long* edge_offs, int* tier_count,
                                   int* tier_list, int* counts) {
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = g < n_graphs;
    const long long n = live ? node_offs[g + 1] - node_offs[g] : 0, m = live ? edge_offs[g + 1] - edge_offs[g] : 0;
    int tier = TLC_TIER_HUGE;
    if (n <= 0) tier = -1;
    else if (n > TLC_MAX_SUBGRAPH_NODES || m > TLC_MAX_SUBGRAPH_EDGES) {
        // does not fit the packed node ids / edge ranks: reported through the counts row, never computed wrongly
        tier = -1;
        for (int c = 0; c < 4; ++c) counts[4 * (size_t)g + c] = -1;
    }
    else if (n <= TLC_S_NMAX && m <= TLC_S_MMAX) tier = TLC_TIER_SMALL;
    else if (n <= TLC_M_NMAX && m <= TLC_M_MMAX) tier = TLC_TIER_MEDIUM;
    else if (n <= TLC_L_NMAX && m <= TLC_L_MMAX) tier = TLC_TIER_LARGE;
    // wave-aggregated append: one returning atomic per wavefront and tier (one per graph on the same counter is served at
    // ~90/us: 8 192 molecule graphs took 96 us to bin)
#pragma unroll
    for (int tt = 0; tt <= TLC_TIER_HUGE; ++tt) {
        const unsigned long long mk = __ballot(tier == tt);
        if (mk == 0ull) continue;
        int base = 0;
        const int leader = __builtin_ctzll(mk);
        if (tlc_lane() == leader) base = atomicAdd(&tier_count[tt], __popcll(mk));
        base = __builtin_amdgcn_readlane(base, leader);
        if (tier == tt) tier_list[(size_t)tt * n_graphs + base + __popcll(mk & tlc_lanemask_lt())] = g;
    }
}

// ======================================================================================================================
// tlc_pi_raster (PersistenceImager.transform, PersistenceImager.pyx:352-388): one LANE per diagram point.
// Sixteen lanes share a diagram (four diagrams per wavefront; mean 41 points on the PubMed batch).  A lane evaluates the
// 2*(res+1) normal CDFs of its point in registers, forms w * dPhi_b[i] and dPhi_p[j] (the factored form of the reference's
// 4-term inclusion-exclusion) and accumulates its res^2 pixel terms in registers; the sixteen partial
// images are folded with DPP butterflies on the vector ALU and stored as whole rows.  No LDS, no barrier: the former
// table version (one lane per (point, grid line) into LDS, then one lane per pixel) spent 0.37 ms on 1.5 M points,
// most of it in LDS round trips and in a wavefront that executed the erfc and the series branch of every table entry.
// Diagrams of TLC_RASTER_MID points or more are taken by the whole wavefront, of TLC_RASTER_BIG or more by the whole
// workgroup (256 lanes), one after the other (a 500-point diagram on sixteen lanes kept its wavefront for 32 rounds).
// Summation order: a lane adds its points in index order (stride 16 / 64 / 256), the lanes are folded in a fixed order
// -- a function of the diagram length only, so results are reproducible.
// ======================================================================================================================
#define TLC_RASTER_MID 96
#define TLC_RASTER_BIG 1024

// Normal CDF at the G grid lines g * step - base, all G Maclaurin series advanced together (independent chains: evaluated one
// after the other inside per-value branches they were 12 x 19 dependent fp64 FMAs per point and the kernel ran at the FMA
// latency).  Arguments beyond the series' range get 0 here and set `slow`; raster_point patches them with erfc.
template <int G>
__device__ __forceinline__ void raster_cdf_row(double base, double step, double (&c)[G], bool& slow) {
    double z[G], t[G], a[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
        const double zz = ((double)g * step - base) * 0.70710678118654752440;
        const bool in = fabs(zz) <= 0.95;
        slow |= !in;
        z[g] = in ? zz : 0.0;
        t[g] = z[g] * z[g];
        a[g] = 4.22140728880708822e-18;
    }
#define TLC_SERIES_TERM(C)                         \
    _Pragma("unroll") for (int g = 0; g < G; ++g) a[g] = fma(a[g], t[g], (C));
    TLC_SERIES_TERM(-8.03273501241577328e-17) TLC_SERIES_TERM(1.44832646435981379e-15) TLC_SERIES_TERM(-2.46682701026445706e-14)
    TLC_SERIES_TERM(3.95542951645852569e-13) TLC_SERIES_TERM(-5.94779401363763541e-12) TLC_SERIES_TERM(8.35070279514723971e-11)
    TLC_SERIES_TERM(-1.08922210371485731e-09) TLC_SERIES_TERM(1.31225329638028058e-08) TLC_SERIES_TERM(-1.45038522231504685e-07)
    TLC_SERIES_TERM(1.45891690009337058e-06) TLC_SERIES_TERM(-1.32275132275132281e-05) TLC_SERIES_TERM(1.06837606837606838e-04)
    TLC_SERIES_TERM(-7.57575757575757575e-04) TLC_SERIES_TERM(4.62962962962962937e-03) TLC_SERIES_TERM(-2.38095238095238082e-02)
    TLC_SERIES_TERM(1.00000000000000006e-01) TLC_SERIES_TERM(-3.33333333333333315e-01) TLC_SERIES_TERM(1.0)
#undef TLC_SERIES_TERM
#pragma unroll
    for (int g = 0; g < G; ++g) c[g] = fma(0.5 * 1.1283791670955126, z[g] * a[g], 0.5);     // same closing step as tlc_norm_cdf
}

// erfc for the grid lines whose argument lies beyond the series' range (c[0..G) birth lines, c[G..2G) persistence lines)
__device__ __attribute__((noinline)) void raster_cdf_slow(double b, double pers, double step, int G, double* c) {
#pragma unroll 1
    for (int g = 0; g < 2 * G; ++g) {
        const double x = (g < G) ? ((double)g * step - b) : ((double)(g - G) * step - pers);
        const double z = x * 0.70710678118654752440;
        if (!(fabs(z) <= 0.95)) c[g] = 0.5 * erfc(-z);
    }
}

template <int RES>
__device__ __forceinline__ void raster_point(double b, double d, double (&acc)[RES * RES]) {
    const double pers = d - b;                                         // skew (:367-368)
    const double wgt = pers < 0.0 ? 0.0 : (pers > 1.0 ? 1.0 : pers);   // linear_ramp (:9-30)
    if (!(wgt != 0.0)) return;                                         // weight 0 (NaN stays in: it propagates as in the reference)
    const double pixel = 1.0 / (double)RES;
    const double step = ((1.0 + pixel) - 0.0) / (double)(RES + 1);     // _create_mesh (:302-314)
    constexpr int G = RES + 1;
    double cb[G], cp[G];                                               // _norm_cdf (:54-60) at the grid lines
    bool slow = false;
    raster_cdf_row<G>(b, step, cb, slow);
    raster_cdf_row<G>(pers, step, cp, slow);
    if (slow) {
        // out of line, through scratch: the library's erfc inlined here costs the whole kernel 70 VGPRs (2 instead of 4
        // wavefronts per SIMD) for a branch that diagrams inside [0,1]^2 never take
        double cc[2 * G];
#pragma unroll
        for (int q = 0; q < G; ++q) { cc[q] = cb[q]; cc[G + q] = cp[q]; }
        raster_cdf_slow(b, pers, step, G, cc);
#pragma unroll
        for (int q = 0; q < G; ++q) { cb[q] = cc[q]; cp[q] = cc[G + q]; }
    }
    double wb[RES], dp[RES];
#pragma unroll
    for (int g = 0; g < RES; ++g) {
        wb[g] = wgt * (cb[g + 1] - cb[g]);
        dp[g] = cp[g + 1] - cp[g];
    }
#pragma unroll
    for (int i = 0; i < RES; ++i)
#pragma unroll
        for (int j = 0; j < RES; ++j) acc[i * RES + j] = fma(wb[i], dp[j], acc[i * RES + j]);
}

// Fold the partial images of a wavefront's lanes through LDS, sixteen pixels at a time: lane `sub` of each 16-lane group sums
// pixel c*16 + sub over its group's lanes (one 128-byte store per diagram instead of 25 single-lane stores behind 100 DPP
// exchanges).  Rows of 17 doubles: writes and reads are bank-conflict free.  WIDE: the four groups hold parts of ONE
// diagram and are added on top (group order fixed).  LDS operations of one wavefront execute in order; the fences only
// keep the compiler from moving them.
template <int R2, bool WIDE>
__device__ __forceinline__ void raster_fold(const double (&acc)[R2], double (*st)[17], bool store, double* __restrict__ row) {
    const int lane = tlc_lane(), sub = lane & 15;
#pragma unroll
    for (int c = 0; c < (R2 + 15) / 16; ++c) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if (c * 16 + q < R2) st[lane][q] = acc[c * 16 + q];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        if (c * 16 + sub < R2) {
            const int l0 = lane & 48;
            double v0 = st[l0][sub], v1 = st[l0 + 1][sub], v2 = st[l0 + 2][sub], v3 = st[l0 + 3][sub];
#pragma unroll
            for (int l = 4; l < 16; l += 4) {
                v0 += st[l0 + l][sub];
                v1 += st[l0 + l + 1][sub];
                v2 += st[l0 + l + 2][sub];
                v3 += st[l0 + l + 3][sub];
            }
            double v = (v0 + v1) + (v2 + v3);
            if (WIDE) {
                v += tlc_lane_xor_f64<16>(v);
                v += tlc_lane_xor_f64<32>(v);
                if (store && lane < 16) row[c * 16 + sub] = v;
            } else if (store) row[c * 16 + sub] = v;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
}

// (three workgroups per CU need <= 168 VGPRs: at resolution 5 that costs 18 spilled VGPRs and 176 B of scratch per lane.  Two per CU
// and no scratch -- make EXTRA=-DTLC_RASTER_OCC3_MAXRES=4 -- was measured in round 4 (tools/time_raster.py, same box): uniform
// 48-point diagrams 39.1 vs 38.8 us, the batch-shaped mix 86.0 vs 81.1, the erfc range 334 vs 300: the spills sit outside the
// series loops and the third workgroup hides more latency than they cost.  Three stay.)
#ifndef TLC_RASTER_OCC3_MAXRES
#define TLC_RASTER_OCC3_MAXRES 5
#endif
template <int RES>
__global__ __launch_bounds__(256, (RES <= TLC_RASTER_OCC3_MAXRES ? 3 : 2)) void tlc_pi_raster_kernel(int n_dgms, int dpb, const long long* __restrict__ offs,
                                                            const double* __restrict__ pts, double* __restrict__ out) {
    constexpr int R2 = RES * RES;
    __shared__ long long s_k[16], s_o[16];
    __shared__ double s_red[4][R2];
    __shared__ double s_t[4][64][17];
    const int tid = threadIdx.x, lane = tlc_lane(), sub = lane & 15, slot = tid >> 4, wv = tid >> 6;
    const double2* __restrict__ pts2 = reinterpret_cast<const double2*>(pts);
    // a workgroup takes dpb = sixteen consecutive diagrams per iteration -- or one, when the batch has too few diagrams to fill
    // the machine that way (block-uniform trip count: the long-diagram path has barriers)
    for (long long base = (long long)blockIdx.x * dpb; base < n_dgms; base += (long long)gridDim.x * dpb) {
        const long long d = slot < dpb ? base + slot : (long long)n_dgms;
        long long o = 0, kl = 0;
        if (d < n_dgms) { o = offs[d]; kl = offs[d + 1] - o; }
        if (kl < 0) kl = 0;
        if (sub == 0) { s_k[slot] = kl; s_o[slot] = o; }
        // ---- diagrams below TLC_RASTER_MID points: sixteen lanes each, four diagrams per wavefront ---------------------
        {
            const int k = kl < TLC_RASTER_MID ? (int)kl : 0;
            int rounds = (k + 15) >> 4;
            rounds = max(max(__builtin_amdgcn_readlane(rounds, 0), __builtin_amdgcn_readlane(rounds, 16)),
                         max(__builtin_amdgcn_readlane(rounds, 32), __builtin_amdgcn_readlane(rounds, 48)));
            double acc[R2];
#pragma unroll
            for (int q = 0; q < R2; ++q) acc[q] = 0.0;
            double2 nxt = make_double2(0.0, 0.0);
            if (sub < k) nxt = pts2[o + sub];
            for (int r = 0; r < rounds; ++r) {
                const int idx = r * 16 + sub;
                const double2 bd = nxt;
                if (idx + 16 < k) nxt = pts2[o + idx + 16];             // next round's point while this one is rasterised
                if (idx < k) raster_point<RES>(bd.x, bd.y, acc);
            }
            raster_fold<R2, false>(acc, s_t[wv], kl < TLC_RASTER_MID && d < n_dgms, out + (size_t)(d < n_dgms ? d : 0) * R2);
        }
        // ---- TLC_RASTER_MID .. TLC_RASTER_BIG - 1 points: the whole wavefront, one diagram after the other --------------
        {
            const int kmid = (kl >= TLC_RASTER_MID && kl < TLC_RASTER_BIG) ? (int)kl : 0;
            if (__builtin_amdgcn_ballot_w64(kmid != 0)) {
                for (int g = 0; g < 4; ++g) {
                    const int kg = __builtin_amdgcn_readlane(kmid, g * 16);
                    if (kg == 0) continue;                              // wavefront-uniform
                    const unsigned olo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)o, g * 16);
                    const unsigned ohi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(o >> 32), g * 16);
                    const long long og = (long long)(((unsigned long long)ohi << 32) | olo);
                    double acc[R2];
#pragma unroll
                    for (int q = 0; q < R2; ++q) acc[q] = 0.0;
                    double2 nxt = make_double2(0.0, 0.0);
                    if (lane < kg) nxt = pts2[og + lane];
                    for (int idx = lane; idx < kg; idx += 64) {
                        const double2 bd = nxt;
                        if (idx + 64 < kg) nxt = pts2[og + idx + 64];
                        raster_point<RES>(bd.x, bd.y, acc);
                    }
                    raster_fold<R2, true>(acc, s_t[wv], true, out + (size_t)(base + wv * 4 + g) * R2);
                }
            }
        }
        // ---- TLC_RASTER_BIG points and more: the whole workgroup, one diagram after the other ---------------------------
        __syncthreads();
        for (int g = 0; g < 16; ++g) {
            const long long kg = s_k[g];
            if (kg < TLC_RASTER_BIG) continue;                          // block-uniform
            const long long og = s_o[g];
            double acc[R2];
#pragma unroll
            for (int q = 0; q < R2; ++q) acc[q] = 0.0;
            for (long long idx = tid; idx < kg; idx += 256) {
                const double2 bd = pts2[og + idx];
                raster_point<RES>(bd.x, bd.y, acc);
            }
#pragma unroll
            for (int q = 0; q < R2; ++q) {
                double v = acc[q];
                v += tlc_lane_xor_f64<1>(v);
                v += tlc_lane_xor_f64<2>(v);
                v += tlc_lane_xor_f64<4>(v);
                v += tlc_lane_xor_f64<8>(v);
                v += tlc_lane_xor_f64<16>(v);
                v += tlc_lane_xor_f64<32>(v);
                if (lane == (q & 63)) s_red[wv][q] = v;
            }
            __syncthreads();
            if (tid < R2) out[(size_t)(base + g) * R2 + tid] = ((s_red[0][tid] + s_red[1][tid]) + s_red[2][tid]) + s_red[3][tid];
            __syncthreads();
        }
        __syncthreads();                                                // s_k / s_o are rewritten by the next iteration
    }
}

// ---- host launchers ------------------------------------------------------------------------------------------------------
size_t tlc_handoff_slot_bytes(int tier) {
    switch (tier) {
        // (MID: none since round 6 -- its tier kernel walks the Pos edges itself.  A 128-thread workgroup with 10.6 KB that keeps a
        // second wavefront waiting during the walk costs a pipelined batch and the long list nothing against the hand-off record +
        // tlc_pd_swap_kernel<128, 256> on the same stream, and one batch alone 0.618 -> 0.589 ms: one launch and 4.6 KB out and in
        // again per vicinity less.  The compact MEDIUM configuration the same way: pipelined batch +5 %, long list +13 %
        // -- four wavefronts and 26 KB wait for one walker there.  gpurun_out A/Bs: profiles/r06_fused_tiers_ab.txt)
        case TLC_TIER_MID: return 0;
        case TLC_TIER_MEDIUM: return handoff_bytes(TLC_C_NMAX, TLC_C_MMAX);
        case TLC_TIER_MEDHI:
        case TLC_TIER_MEDWIDE: return handoff_bytes(TLC_M_NMAX, TLC_M_MMAX);
        // (LARGE keeps the serial cycle swap of its subgraphs with few Pos edges: they are few and the batch waits for the
        // slowest of them, which runs fastest with a CU to itself -- measured 0.91 vs 1.07 ms with the swap in the shared
        // one-wavefront kernel; only the subgraphs meant for tlc_pd_dc_kernel are handed off, into a buffer of their own)
        case TLC_TIER_LARGE: return TLC_DC_LARGE_MODE == 2 ? handoff_bytes(TLC_L_NMAX, TLC_L_MMAX) : 0;
        default: return 0;
    }
}
size_t tlc_huge_slot_bytes(int nmax, int mmax) { return al16(make_layout(nmax, mmax, false, 2, TLC_HUGE_MIN_TABLE).total); }

template <class K>
static int set_lds_limit(K kernel, size_t bytes) {
    if (bytes > 64 * 1024) TLC_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
    return TLC_OK;
}

#ifndef TLC_C_THREADS
#define TLC_C_THREADS 256          /* threads of the compact MEDIUM tier kernel (128 measured: see DESIGN_HISTORY.md, round 6) */
#endif
int tlc_launch_pd_tier(int tier, const TlcPdParams& p, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (p.tier_count <= 0) return TLC_OK;
    // tiers with a hand-off buffer leave the cycle swap + image of their subgraphs to tlc_pd_swap_kernel, same stream
    const bool deferring = p.handoff != nullptr && p.pi_enabled && !(p.flags & TLC_NO_EXT1);
    const int grid = p.grid > 0 ? p.grid : p.tier_count;
    // (the instances with the plain image batch's parameters as constants: tlc_pd_tier_kernel's PLAIN; handle option plain_kernels = 0: the general ones)
    const bool plain = !p.no_plain && p.flags == 0u && p.res == 5 && p.pi_enabled && !p.out_f && !p.out_n && !p.out_edges && !p.out_m;
    static const bool host_trace = getenv("TLC_HOST_TRACE") != nullptr;
    if (host_trace) {                                 // (development: LDS bytes per workgroup of every kernel of the tiers)
        static int once = 0;
        if (!once++)
            fprintf(stderr, "[tlc] LDS per workgroup: SMALL %zu | MEDIUM tier %zu swap %zu | MEDWIDE tier %zu swap %zu | MID tier %zu swap %zu | LARGE tier %zu dc %zu\n",
                    (size_t)make_layout(TLC_S_NMAX, TLC_S_MMAX, TLC_SMALL_LWL, 2).total, (size_t)make_layout(TLC_C_NMAX, TLC_C_MMAX, false, 2).total,
                    (size_t)make_swap_layout(TLC_C_NMAX, TLC_C_MMAX).total, (size_t)make_layout(TLC_M_NMAX, TLC_M_MMAX, false, 2).total,
                    (size_t)make_swap_layout(TLC_M_NMAX, TLC_M_MMAX).total, (size_t)make_layout(TLC_D_NMAX, TLC_D_MMAX, false, 2).total,
                    (size_t)make_swap_layout(TLC_D_NMAX, TLC_D_MMAX).total, (size_t)make_layout(TLC_L_NMAX, TLC_L_MMAX, false, 2).total,
                    (size_t)dc_kernel_lds(TLC_L_NMAX, TLC_L_MMAX));
    }
    switch (tier) {
        case TLC_TIER_SMALL: {
            constexpr Layout L = make_layout(TLC_S_NMAX, TLC_S_MMAX, TLC_SMALL_LWL, 2);
            // (development: TLC_SMALL_LDS_PAD=bytes inflates this tier's footprint -- the experiment behind DESIGN.md's "the tier
            // phase is bound by LDS capacity x time": +4 KB here costs the batch 2.5 %, +16 KB 20 %)
            static const size_t pad = getenv("TLC_SMALL_LDS_PAD") ? (size_t)atoi(getenv("TLC_SMALL_LDS_PAD")) : 0;
            if (plain)
                hipLaunchKernelGGL((tlc_pd_tier_kernel<TLC_S_NMAX, TLC_S_MMAX, 64, TLC_SMALL_LWL, false, true>), dim3(p.tier_count), dim3(64),
                                   L.total + pad, s, p);
            else
                hipLaunchKernelGGL((tlc_pd_tier_kernel<TLC_S_NMAX, TLC_S_MMAX, 64, TLC_SMALL_LWL, false>), dim3(p.tier_count), dim3(64),
                                   L.total + pad, s, p);
            break;
        }
        case TLC_TIER_MEDIUM: {
            constexpr Layout L = make_layout(TLC_C_NMAX, TLC_C_MMAX, false, 2);
            static const size_t mpad = getenv("TLC_MEDIUM_LDS_PAD") ? (size_t)atoi(getenv("TLC_MEDIUM_LDS_PAD")) : 0;
            if (p.phase != 2) {
                if (plain)
                    hipLaunchKernelGGL((tlc_pd_tier_kernel<TLC_C_NMAX, TLC_C_MMAX, TLC_C_THREADS, false, false, true>), dim3(grid),
                                       dim3(TLC_C_THREADS), L.total + mpad, s, p);
                else
                    hipLaunchKernelGGL((tlc_pd_tier_kernel<TLC_C_NMAX, TLC_C_MMAX, TLC_C_THREADS, false, false>), dim3(grid),
                                       dim3(TLC_C_THREADS), L.total + mpad, s, p);
            }
            if (deferring && p.phase != 1) {
                constexpr SwapLayout SL = make_swap_layout(TLC_C_NMAX, TLC_C_MMAX);
                if (plain) hipLaunchKernelGGL((tlc_pd_swap_kernel<TLC_C_NMAX, TLC_C_MMAX, true>), dim3(grid), dim3(64), SL.total, s, p);
                else hipLaunchKernelGGL((tlc_pd_swap_kernel<TLC_C_NMAX, TLC_C_MMAX>), dim3(grid), dim3(64), SL.total, s, p);
            }
            break;
        }
        case TLC_TIER_MEDHI:
        case TLC_TIER_MEDWIDE: {
            constexpr Layout L = make_layout(TLC_M_NMAX, TLC_M_MMAX, false, 2);
            // (development: TLC_MEDIUM_LDS_PAD=bytes -- how much does this tier's footprint cost?  37.5 KB = four workgroups per CU)
            static const size_t mpad = getenv("TLC_MEDIUM_LDS_PAD") ? (size_t)atoi(getenv("TLC_MEDIUM_LDS_PAD")) : 0;
            if (p.phase != 2) {
                if (plain)
                    hipLaunchKernelGGL((tlc_pd_tier_kernel<TLC_M_NMAX, TLC_M_MMAX, 256, false, false, true>), dim3(grid),
                                       dim3(256), L.total + mpad, s, p);
                else
                    hipLaunchKernelGGL((tlc_pd_tier_kernel<TLC_M_NMAX, TLC_M_MMAX, 256, false, false>), dim3(grid),
                                       dim3(256), L.total + mpad, s, p);
            }
            // (a dc list on the launch: the scan counted vicinities with Pos edges enough -- dense hop-1 vicinities of the Amazon
            // shapes, 600 Pos edges on 80 nodes -- and the tier kernel marked them; the rest stay for the swap kernel behind)
            if (deferring && p.phase != 1 && p.dc_count) {
                constexpr size_t dcm = dc_kernel_lds(TLC_M_NMAX, TLC_M_MMAX);
                hipLaunchKernelGGL((tlc_pd_dc_kernel<TLC_M_NMAX, TLC_M_MMAX, 256>), dim3(grid), dim3(256), dcm, s, p);
            }
            if (deferring && p.phase != 1) {
                constexpr SwapLayout SL = make_swap_layout(TLC_M_NMAX, TLC_M_MMAX);
                if (plain) hipLaunchKernelGGL((tlc_pd_swap_kernel<TLC_M_NMAX, TLC_M_MMAX, true>), dim3(grid), dim3(64), SL.total, s, p);
                else hipLaunchKernelGGL((tlc_pd_swap_kernel<TLC_M_NMAX, TLC_M_MMAX>), dim3(grid), dim3(64), SL.total, s, p);
            }
            break;
        }
        case TLC_TIER_MID: {
            constexpr Layout L = make_layout(TLC_D_NMAX, TLC_D_MMAX, false, 2);
            if (p.phase != 2) {
                if (plain)
                    hipLaunchKernelGGL((tlc_pd_tier_kernel<TLC_D_NMAX, TLC_D_MMAX, TLC_D_THREADS, false, false, true>), dim3(grid),
                                       dim3(TLC_D_THREADS), L.total, s, p);
                else
                    hipLaunchKernelGGL((tlc_pd_tier_kernel<TLC_D_NMAX, TLC_D_MMAX, TLC_D_THREADS, false, false>), dim3(grid),
                                       dim3(TLC_D_THREADS), L.total, s, p);
            }
            break;                                                    // (no hand-off: tlc_handoff_slot_bytes)
        }
        case TLC_TIER_LARGE: {
            constexpr Layout L = make_layout(TLC_L_NMAX, TLC_L_MMAX, false, 2);
            // the whole CU: no SMALL workgroup beside the wavefront that carries the batch's longest serial chain
            // (bit 0: this kernel, bit 1: the divide-and-conquer kernel behind it.  The latter no longer does: with 107 of 160 KB a
            // TINY workgroup fits beside it, 0.838 -> 0.818 ms for the PubMed batch; development A/B: tools/gpu_large_excl.sh)
            // Round 5: off by default.  With the divide and conquer in place a LARGE workgroup holds its CU for ~0.49 ms of a 0.53 ms
            // pipelined batch (71 of them: 0.065 ms of the batch, profiles/r05_tier_cost_pipelined.txt); the 15 KB its 145 KB leave are
            // room for two SMALL or one MID / swap workgroup: pipelined batch 0.526 -> 0.517 ms (two alternating runs), one batch alone equal.
            static const int excl = getenv("TLC_LARGE_EXCL") ? atoi(getenv("TLC_LARGE_EXCL")) : 0;
            const size_t lds_bytes = (L.total > 156 * 1024 || !(excl & 1)) ? L.total : 156 * 1024;
            int rc = plain ? set_lds_limit(tlc_pd_tier_kernel<TLC_L_NMAX, TLC_L_MMAX, TLC_L_THREADS, false, false, true>, lds_bytes)
                           : set_lds_limit(tlc_pd_tier_kernel<TLC_L_NMAX, TLC_L_MMAX, TLC_L_THREADS, false, false>, lds_bytes);
            if (rc) return rc;
            if (plain)
                hipLaunchKernelGGL((tlc_pd_tier_kernel<TLC_L_NMAX, TLC_L_MMAX, TLC_L_THREADS, false, false, true>), dim3(p.tier_count),
                                   dim3(TLC_L_THREADS), lds_bytes, s, p);
            else
                hipLaunchKernelGGL((tlc_pd_tier_kernel<TLC_L_NMAX, TLC_L_MMAX, TLC_L_THREADS, false, false>), dim3(p.tier_count),
                                   dim3(TLC_L_THREADS), lds_bytes, s, p);
            if (deferring) {
                // (only the subgraphs marked for the divide and conquer were handed off; tlc_pd_dc_kernel runs the serial walk itself
                // for those it gives back)
                // (its own 107 KB only, see above)
                const size_t dcl = (dc_kernel_lds(TLC_L_NMAX, TLC_L_MMAX) > 156 * 1024 || !(excl & 2)) ? dc_kernel_lds(TLC_L_NMAX, TLC_L_MMAX) : 156 * 1024;
                if (host_trace) { static int once = 0; if (!once++) fprintf(stderr, "[tlc] LARGE tier LDS %zu (layout %zu), dc %zu (layout %zu)\n", lds_bytes, (size_t)L.total, dcl, (size_t)dc_kernel_lds(TLC_L_NMAX, TLC_L_MMAX)); }
                rc = set_lds_limit(tlc_pd_dc_kernel<TLC_L_NMAX, TLC_L_MMAX, TLC_L_THREADS>, dcl);
                if (rc) return rc;
                if (p.dc_count && !p.dc_inplace)
                    hipLaunchKernelGGL((tlc_pd_dc_kernel<TLC_L_NMAX, TLC_L_MMAX, TLC_L_THREADS>), dim3(p.tier_count),
                                       dim3(TLC_L_THREADS), dcl, s, p);
            }
            break;
        }
        case TLC_TIER_HUGE: {
            if (p.huge_slots <= 0 || !p.huge_scratch) { tlc_set_error("HUGE tier without scratch"); return TLC_ERR_INVALID_ARG; }
            const int grid = p.tier_count < p.huge_slots ? p.tier_count : p.huge_slots;
            // (LDS for the cycle swap's tables: a HUGE workgroup takes a CU's LDS -- there are a handful of them in a batch, each
            // the longest serial chain of it; TLC_HUGE_LDS=0: the tables stay in the scratch slot, development A/B)
            static const int huge_lds = getenv("TLC_HUGE_LDS") ? atoi(getenv("TLC_HUGE_LDS")) : 144 * 1024;
            TlcPdParams q = p;
            q.huge_lds = huge_lds;
            if (huge_lds > 0) { int rc = set_lds_limit(tlc_pd_tier_kernel<0, 0, 256, false, true>, (size_t)huge_lds); if (rc) return rc; }
            hipLaunchKernelGGL((tlc_pd_tier_kernel<0, 0, 256, false, true>), dim3(grid), dim3(256), (size_t)huge_lds, s, q);
            break;
        }
        default:
            tlc_set_error("bad tier");
            return TLC_ERR_INVALID_ARG;
    }
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

int tlc_launch_pdf_tier(int tier, const TlcPdfParams& p, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (p.count <= 0) return TLC_OK;
    switch (tier) {
        case TLC_TIER_SMALL: {
            constexpr Layout L = make_layout(TLC_S_NMAX, TLC_S_MMAX, false, 2);
            hipLaunchKernelGGL((tlc_pdf_tier_kernel<TLC_S_NMAX, TLC_S_MMAX, 64, false>), dim3(p.count), dim3(64), L.total, s, p);
            break;
        }
        case TLC_TIER_MEDIUM: {
            constexpr Layout L = make_layout(TLC_M_NMAX, TLC_M_MMAX, false, 2);
            hipLaunchKernelGGL((tlc_pdf_tier_kernel<TLC_M_NMAX, TLC_M_MMAX, 256, false>), dim3(p.count), dim3(256), L.total, s, p);
            break;
        }
        case TLC_TIER_LARGE: {
            constexpr Layout L = make_layout(TLC_L_NMAX, TLC_L_MMAX, false, 2);
            int rc = set_lds_limit(tlc_pdf_tier_kernel<TLC_L_NMAX, TLC_L_MMAX, 512, false>, L.total);
            if (rc) return rc;
            hipLaunchKernelGGL((tlc_pdf_tier_kernel<TLC_L_NMAX, TLC_L_MMAX, 512, false>), dim3(p.count), dim3(512), L.total, s, p);
            break;
        }
        case TLC_TIER_HUGE: {
            if (p.huge_slots <= 0 || !p.huge_scratch) { tlc_set_error("HUGE tier without scratch"); return TLC_ERR_INVALID_ARG; }
            const int grid = p.count < p.huge_slots ? p.count : p.huge_slots;
            hipLaunchKernelGGL((tlc_pdf_tier_kernel<0, 0, 256, true>), dim3(grid), dim3(256), 0, s, p);
            break;
        }
        default:
            tlc_set_error("bad tier");
            return TLC_ERR_INVALID_ARG;
    }
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

int tlc_launch_pdf_bin(int n_graphs, const long long* node_offs, const long long* edge_offs, int* counts, int* tier_count,
                       int* tier_list, void* stream) {
    if (n_graphs <= 0) return TLC_OK;
    hipLaunchKernelGGL(tlc_pdf_bin_kernel, dim3((n_graphs + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_graphs,
                       node_offs, edge_offs, tier_count, tier_list, counts);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

template <int RES>
static void launch_pi_raster(int n_dgms, const long long* offs, const double* pts, double* out, hipStream_t s) {
    // four diagrams per wavefront, four wavefronts per workgroup; the grid is capped and strided beyond 2^20 workgroups
    const int dpb = n_dgms >= 16 * 1024 ? 16 : 1;
    long long blocks = ((long long)n_dgms + dpb - 1) / dpb;
    if (blocks > (1 << 20)) blocks = 1 << 20;
    hipLaunchKernelGGL(tlc_pi_raster_kernel<RES>, dim3((unsigned)blocks), dim3(256), 0, s, n_dgms, dpb, offs, pts, out);
}

// d image / d points of the DIFFERENTIABLE imager, Knowledge_Distillation/pimg.py:354-400 (Teacher_Model.forward(grad_PI=True),
// Teacher_model.py:80-81): there the two normal-CDF factors of a point are computed from detached coordinates (`.detach()`, :392,395),
// so the only path from a point to the image is its weight, linear_ramp(pers) (:11-30): slope 1 for 0 <= pers <= 1, constant outside.
// Hence d L / d pers_i = sum over the pixels of dL/d image x (dPhi_b x dPhi_p of point i), and with pers = death - birth (:371)
// d L / d death_i = that, d L / d birth_i = minus that.  One thread per point; its diagram by bisection in offs.
__global__ void tlc_pi_raster_wgrad_kernel(int n_dgms, long long n_pts, const long long* __restrict__ offs, const double* __restrict__ pts,
                                           int res, const double* __restrict__ grad_img, double* __restrict__ grad_pts) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pts) return;
    const double b = pts[2 * i], pers = pts[2 * i + 1] - pts[2 * i];
    double g = 0.0;
    if (pers >= 0.0 && pers <= 1.0) {
        int lo = 0, hi = n_dgms;                                   // offs[lo] <= i < offs[hi]
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (offs[mid] <= i) lo = mid; else hi = mid; }
        const double* gi = grad_img + (size_t)lo * res * res;
        const double step = (1.0 + 1.0 / (double)res) / (double)(res + 1);        // _create_mesh (:304-316)
        double cp[9];
        for (int q = 0; q <= res; ++q) cp[q] = tlc_norm_cdf<false>((double)q * step - pers);
        double cb0 = tlc_norm_cdf<false>(0.0 - b);
        for (int pi = 0; pi < res; ++pi) {
            const double cb1 = tlc_norm_cdf<false>((double)(pi + 1) * step - b);
            const double db = cb1 - cb0;
            for (int pj = 0; pj < res; ++pj) g += gi[pi * res + pj] * (db * (cp[pj + 1] - cp[pj]));
            cb0 = cb1;
        }
    }
    grad_pts[2 * i] = -g;
    grad_pts[2 * i + 1] = g;
}

int tlc_launch_pi_raster_wgrad(int n_dgms, long long n_pts, const long long* offs, const double* pts, int res, const double* grad_img,
                               double* grad_pts, void* stream) {
    if (n_pts <= 0) return TLC_OK;
    hipLaunchKernelGGL(tlc_pi_raster_wgrad_kernel, dim3((unsigned)((n_pts + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       n_dgms, n_pts, offs, pts, res, grad_img, grad_pts);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

int tlc_launch_pi_raster(int n_dgms, const long long* offs, const double* pts, int res, double* out, void* stream) {
    if (n_dgms <= 0) return TLC_OK;
    hipStream_t s = (hipStream_t)stream;
    switch (res) {
        case 1: launch_pi_raster<1>(n_dgms, offs, pts, out, s); break;
        case 2: launch_pi_raster<2>(n_dgms, offs, pts, out, s); break;
        case 3: launch_pi_raster<3>(n_dgms, offs, pts, out, s); break;
        case 4: launch_pi_raster<4>(n_dgms, offs, pts, out, s); break;
        case 5: launch_pi_raster<5>(n_dgms, offs, pts, out, s); break;
        case 6: launch_pi_raster<6>(n_dgms, offs, pts, out, s); break;
        case 7: launch_pi_raster<7>(n_dgms, offs, pts, out, s); break;
        case 8: launch_pi_raster<8>(n_dgms, offs, pts, out, s); break;
        default: tlc_set_error("res must be in 1..8"); return TLC_ERR_INVALID_ARG;
    }
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// extract_lane.hip -- P4 of the hot path for the SMALLEST vicinities, ONE LANE per pair.
//
// sg2dgm_accelerate's BFS + set intersection + graph.subgraph (sg2dgm/riccidist2dgm.py:310-316) for the pairs whose smaller
// ball has at most xl_cut (24) nodes: on the PubMed-shaped batch that is 20 236 of 37 676 pairs, 18 780 of which end up in the
// lane-per-subgraph PD kernel (pd_tiny.hip: <= 16 nodes, <= 24 edges).  tlc_extract_kernel gives such a pair a whole wavefront:
// a 6 KB bitmap to clear, ~1 000 vector instructions and 8 us for a vicinity of nine nodes and nine edges, 36 % of that
// kernel's wavefront time -- and an arena round trip into the kernel that then runs the pair on one lane anyway.  Here a lane
//   * reads the smaller ball list (<= 32 ids) and looks each id up in the MEMBERSHIP TABLE of the other endpoint's ball (per node
//     a small two-choice bucket hash of its ball, built once per graph and hop beside the ball lists: two 16-byte loads per id,
//     no dependent round -- a binary search of the sorted list was ten dependent gathers per id and 29 % of this kernel);
//   * keeps S (<= 16 sorted ids = the local ids) in a lane-interleaved LDS array; "is y a member, which" is a four-step search,
//     32 of them in flight at a time (one after the other they cost 600 cycles each);
//   * walks the members' node records and row segments in the canonical entry order of extract.hip (round 0: the four entries
//     of every record in member order; then the eight-entry segments round by round; then heavy x heavy through the dense
//     table), skipping heavy members' rows and mirroring what is found at the other end exactly as x_sweep_wave does -- so the
//     undirected edge list (the entries with src < dst, in that order) is the one the lane-per-subgraph kernel used to read
//     from the arena, and images agree with the wavefront path to the last bit wherever keys do not tie differently;
//   * writes the edge list and weights as a fixed-size RECORD (lane-interleaved per 64 pairs: coalesced both ways) that
//     tlc_pd_tiny_kernel<REC> reads in place of the arena: no arena bytes, no scan entry, no tier list.
// A pair that does not fit (more than 16 nodes or 24 edges, a member with a long row that must be scanned) is GIVEN BACK:
// appended to bin 3 of the classification lists, which tlc_extract_kernel takes after its other bins; a pair without any
// common node is finished here (zero row, status "disconnected").  Pairs with an id that the graph does not contain are left
// to the main pass (its predicate for "candidate of this pass" excludes them).
#include "vicinity_dev.h"

namespace {

constexpr int XC = TLC_XL_MAXCUT;     // candidates (ids of the smaller ball) a lane looks up
constexpr int XN = TLC_T_NMAX;        // members kept
constexpr int XM = TLC_T_MMAX;        // edges a record holds

template <typename T>
struct XlArr {
    T* base;
    __device__ __forceinline__ T& operator[](int k) const { return base[k * 64]; }
};

// LDS per wavefront (lane-interleaved rows of 64): sid u32[16] | r0 u32[16] (rest0 = row start + inline entries; heavy: index in the
// heavy set) | re u32[16] | edge u32[XM + 1] | weight f64[XM + 1] | seg u8[65]   (the extra row takes the writes of entries that are
// not kept: the sweep is straight-line code, "keep" only advances the cursor)
constexpr size_t XL_O_SID = 0;
constexpr size_t XL_O_R0 = XL_O_SID + (size_t)XN * 256;
constexpr size_t XL_O_RE = XL_O_R0 + (size_t)XN * 256;
constexpr size_t XL_O_EDGE = XL_O_RE + (size_t)XN * 256;
constexpr size_t XL_O_W = XL_O_EDGE + (size_t)(XM + 1) * 256;
constexpr size_t XL_O_SEG = XL_O_W + (size_t)(XM + 1) * 512;
constexpr size_t XL_LDS = XL_O_SEG + 65 * 64;

}  // namespace

// Everything between the head of a slot and its verdict is written WITHOUT divergent branches: loads take a harmless address when a
// lane has nothing to load, look-ups are computed and masked, an entry is written to the cursor row and the cursor moves only if
// the entry is kept.  (The first version branched around every load and every entry: 1 400 branches and 1 200 exec-mask saves in
// 16 500 instructions, global stores inside the sweep that the next loads' s_waitcnt vmcnt then waited for: 167 k cycles per slot.)
// XL_WPB wavefronts per workgroup, each with its own slots and its own LDS rows (no barrier anywhere): FOUR, so that the ~300
// wavefronts of a chunk sit on ~80 CUs, one per SIMD, instead of one or two on every CU of the machine -- a workgroup of the LARGE
// tier needs a whole CU's LDS, and with a 34 KB tenant on every CU none of them could be placed until this kernel had drained
// (measured: its residency gate ran into its 50 us bound, pipelined batches 0.67 -> 0.72 ms).
#ifndef XL_WPB
#define XL_WPB 4
#endif
__global__ __launch_bounds__(64 * XL_WPB) void tlc_xlane_kernel(TlcXlParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xl_lds_all[];
    unsigned char* xl_lds = xl_lds_all + (size_t)(threadIdx.x >> 6) * XL_LDS;
    const int lane = tlc_lane();
    XlArr<unsigned> sid{(unsigned*)(xl_lds + XL_O_SID) + lane};
    XlArr<unsigned> r0L{(unsigned*)(xl_lds + XL_O_R0) + lane};
    XlArr<unsigned> reL{(unsigned*)(xl_lds + XL_O_RE) + lane};
    XlArr<unsigned> edgeL{(unsigned*)(xl_lds + XL_O_EDGE) + lane};
    XlArr<double> wL{(double*)(xl_lds + XL_O_W) + lane};
    XlArr<unsigned char> segL{xl_lds + XL_O_SEG + lane};
    int count = *p.xl_count;
    count = count < p.xl_cap ? count : p.xl_cap;
    const int nslots = (count + 63) >> 6;
    unsigned long long* pc = p.dbg;
#define XL_STAMP(k) do { if (pc) { const unsigned long long _t = clock64(); if (lane == 0) atomicAdd(&pc[(k)], _t - t_prev); t_prev = _t; } } while (0)
    for (int slot = (int)blockIdx.x * XL_WPB + (int)(threadIdx.x >> 6); slot < nslots; slot += (int)gridDim.x * XL_WPB) {
        unsigned long long t_prev = pc ? clock64() : 0ull;
        const int wi = slot * 64 + lane;
        unsigned char* rec = p.rec + (size_t)slot * TLC_XL_REC_BYTES;
        bool act = wi < count;
        const int i = p.xl_list[act ? wi : 0];                                     // (count >= 1 here: xl_list[0] is a pair)
        const int u = p.pairs[2 * (size_t)i], v = p.pairs[2 * (size_t)i + 1];      // (ids in range: tlc_classify_kernel)
        int a0, a1, b0, b1;
        {
            int ru0, ru1, rv0, rv1;
            row_bounds(p.rowptr, u, ru0, ru1);
            row_bounds(p.rowptr, v, rv0, rv1);
            row_bounds(p.bptr, u, a0, a1);
            row_bounds(p.bptr, v, b0, b1);
            // KeyError on dict_node (riccidist2dgm.py:353): a node without edges is not in the edge-built graph -- the main pass's row
            act = act && ru1 != ru0 && rv1 != rv0;
        }
        const bool v_big = a1 - a0 < b1 - b0;                                      // [b0, b1) becomes the smaller ball
        const int big_node = v_big ? v : u;
        if (v_big) { b0 = a0; b1 = a1; }
        int nB = act ? b1 - b0 : 0;
        nB = nB < XC ? nB : XC;                                                    // (<= xl_cut <= XC by classification)
        XL_STAMP(0);
        // ---- S = ball(u) & ball(v) (:315): every id of the smaller list looked up in the other ball's membership table ---------------
        int cand[XC];
#pragma unroll
        for (int c = 0; c < XC; ++c) cand[c] = p.bcol[b0 + (c < nB ? c : 0)];
        unsigned eq = 0u;
        bool no_table;
        {
            const long long d = p.hptr[big_node];
            const int lg = (int)(d & 63);
            no_table = lg == 63;
            const unsigned mask = no_table ? 0u : ((1u << lg) - 1u);
            const TlcI4* tab = reinterpret_cast<const TlcI4*>(p.htab) + (d >> 6);
#pragma unroll
            for (int g = 0; g < XC; g += 16) {
                TlcI4 ba[16], bb[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const unsigned y = (unsigned)cand[g + q];
                    ba[q] = tab[TLC_XL_H1(y) & mask];
                    bb[q] = tab[TLC_XL_H2(y) & mask];
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int y = cand[g + q];
                    const bool h = ba[q].v[0] == y || ba[q].v[1] == y || ba[q].v[2] == y || ba[q].v[3] == y ||
                                   bb[q].v[0] == y || bb[q].v[1] == y || bb[q].v[2] == y || bb[q].v[3] == y;
                    eq |= (h && g + q < nB) ? (1u << (g + q)) : 0u;
                }
            }
        }
        XL_STAMP(1);
        int n = __popc(eq);
        bool give_back = act && (n > p.ncut || no_table);
        act = act && !give_back;
        int lu = -1, lv = -1;
        {
            // the members in ascending id = list order; row k of `sid` for a lane with fewer members: INT_MAX
            int k = 0;
#pragma unroll
            for (int c = 0; c < XC; ++c) {
                const bool h = (eq >> c) & 1u;
                const int kk = k < XN ? k : XN - 1;                                  // (more than 16 hits: given back above, rows are junk)
                sid[kk] = h ? (unsigned)cand[c] : sid[kk];
                lu = (h && cand[c] == u) ? k : lu;
                lv = (h && cand[c] == v) ? k : lv;
                k += h ? 1 : 0;
            }
#pragma unroll
            for (int kk = 0; kk < XN; ++kk) sid[kk] = kk < n ? sid[kk] : 0x7fffffffu;
        }
        const bool empty = act && n == 0;                 // AssertionError, zero connected components (:318): the row is exactly zero
        act = act && n > 0;
        n = act ? n : 0;
        // local id of node y: position in the sorted member list (padded with INT_MAX), four halvings; branch-free, so that the 32
        // look-ups of a batch overlap their LDS round trips.  -> position | 16 if y is a member
        const unsigned s7 = sid[7];
        auto find = [&](int y) -> int {
            int pos = s7 < (unsigned)y ? 8 : 0;
            pos += sid[pos + 3] < (unsigned)y ? 4 : 0;
            pos += sid[pos + 1] < (unsigned)y ? 2 : 0;
            pos += sid[pos] < (unsigned)y ? 1 : 0;
            return pos | (sid[pos] == (unsigned)y ? 16 : 0);
        };
        XL_STAMP(2);
        // ---- the members' record heads: row start, degree, heavy index --------------------------------------------------------------
        unsigned hv = 0u;                                  // heavy members (their rows are never read)
        unsigned long long segpack = 0ull;                 // eight-entry segments behind the record of member k: 3 bits each
        {
            TlcI4 hd[XN];
#pragma unroll
            for (int k = 0; k < XN; ++k) hd[k] = *reinterpret_cast<const TlcI4*>(p.nrec + (k < n ? sid[k] : 0u));
#pragma unroll
            for (int k = 0; k < XN; ++k) {
                const bool valid = k < n;
                const int x = (int)sid[k];
                const int rb = hd[k].v[0], deg = hd[k].v[1], hidx = hd[k].v[2], n_in = hd[k].v[3];
                const bool heavy = valid && p.hh_k > 0 && hidx >= 0 && x != u && x != v;
                const bool scan = valid && !heavy;
                hv |= heavy ? (1u << k) : 0u;
                give_back = give_back || (scan && deg >= 32);               // a long row that must be scanned: the wavefront kernel's job
                r0L[k] = heavy ? (unsigned)hidx : (unsigned)(rb + n_in);
                reL[k] = scan ? (unsigned)(rb + deg) : 0u;
                const int segs = scan ? (deg - n_in + 7) >> 3 : 0;
                segpack |= (unsigned long long)(segs < 4 ? segs : 4) << (3 * k);
            }
        }
        if (give_back) { act = false; n = 0; segpack = 0ull; hv = 0u; }
        // one found entry k -> ly (canonical order: extract.hip, x_sweep_wave): counted as a directed entry, with its mirror when ly
        // is a heavy member; the undirected edge is kept once, lower local id first.  Written at the cursor, kept by moving it.
        int m = 0, m2 = 0;
        auto emit = [&](bool found, int k, int ly, double w) {
            const unsigned hb = (hv >> ly) & 1u;
            m2 += found ? 1 + (int)hb : 0;
            const bool keep = found && k != ly && (hb != 0u || k < ly);
            const int a = k < ly ? k : ly, b = k < ly ? ly : k;
            const int at = m < XM ? m : XM;
            edgeL[at] = ((unsigned)a << 8) | (unsigned)b;
            wL[at] = w;
            m += keep ? 1 : 0;
        };
        XL_STAMP(3);
        // ---- round 0: the entries held in the records, member by member (eight records in flight) ------------------------------------
#pragma unroll
        for (int g = 0; g < XN; g += 8) {
            if (__ballot(g < n) == 0ull) break;
            TlcI4 hc[8], cc[8];
            TlcD2 wa[8], wb[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const TlcNodeRec* r = p.nrec + (g + q < n ? sid[g + q] : 0u);
                hc[q] = *reinterpret_cast<const TlcI4*>(r);
                cc[q] = *reinterpret_cast<const TlcI4*>(&r->col[0]);
                wa[q] = *reinterpret_cast<const TlcD2*>(&r->w[0]);
                wb[q] = *reinterpret_cast<const TlcD2*>(&r->w[2]);
            }
            int fr[8][4];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int k = g + q;
                const int n_in = (k < n && !((hv >> k) & 1u)) ? hc[q].v[3] : 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) { const int f = find(cc[q].v[e]); fr[q][e] = e < n_in ? f : 0; }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
#pragma unroll
                for (int e = 0; e < 4; ++e) emit((fr[q][e] & 16) != 0, g + q, fr[q][e] & 15, e < 2 ? wa[q].v[e] : wb[q].v[e - 2]);
            }
        }
        XL_STAMP(4);
        // ---- the rows beyond the records, in eight-entry segments: round r of member k, by round, then by member -------------------------
        int T = 0;
        if (__ballot(segpack != 0ull) != 0ull) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int k = 0; k < XN; ++k) {
                    segL[T] = (unsigned char)(k | (r << 4));
                    T += (int)((segpack >> (3 * k)) & 7ull) > r ? 1 : 0;
                }
            }
            int tmax = T;
            for (int o = 32; o; o >>= 1) { const int t = __shfl_xor(tmax, o); tmax = t > tmax ? t : tmax; }
            for (int t0 = 0; t0 < tmax; t0 += 4) {
                int bb[4][8], kk[4], jb[4], je[4];
                double ww[4][8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool valid = t0 + q < T;
                    const int sg = segL[valid ? t0 + q : 64];
                    kk[q] = sg & 15;
                    jb[q] = valid ? (int)r0L[kk[q]] + 8 * (sg >> 4) : 0;
                    je[q] = valid ? (int)reL[kk[q]] : 0;
                    load_row8(p.col, jb[q], bb[q]);
                    load_row8w(p.w, jb[q], ww[q]);
                }
                int fr[4][8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const int f = find(bb[q][e]); fr[q][e] = jb[q] + e < je[q] ? f : 0; }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) emit((fr[q][e] & 16) != 0, kk[q], fr[q][e] & 15, ww[q][e]);
                }
            }
        }
        XL_STAMP(5);
        // ---- heavy x heavy: ordered pairs of heavy members in member order, through the dense table (0 = not adjacent) -----------------
        // (two heavy members at least -- or one, if some heavy node of this graph has a self loop)
        const unsigned hvx = (__popc(hv) >= 2 || (p.hh_diag && hv != 0u)) ? hv : 0u;
        if (__ballot(hvx != 0u) != 0ull) {
            for (int ka = 0; ka < XN; ++ka) {
                if (__ballot((hvx >> ka) & 1u) == 0ull) continue;
                double wr[XN];
#pragma unroll
                for (int kb = 0; kb < XN; ++kb) {
                    const bool both = ((hvx >> ka) & 1u) && ((hvx >> kb) & 1u);
                    wr[kb] = p.hh_w[both ? (size_t)r0L[ka] * p.hh_k + r0L[kb] : (size_t)0];
                    wr[kb] = both ? wr[kb] : 0.0;
                }
#pragma unroll
                for (int kb = 0; kb < XN; ++kb) {
                    const bool adj = wr[kb] > 0.0;
                    m2 += adj ? 1 : 0;
                    const int at = m < XM ? m : XM;
                    edgeL[at] = ((unsigned)ka << 8) | (unsigned)kb;
                    wL[at] = wr[kb];
                    m += (adj && ka < kb) ? 1 : 0;
                }
            }
        }
        XL_STAMP(6);
        if (act && m > p.mcut) { give_back = true; act = false; }
        // ---- verdict ------------------------------------------------------------------------------------------------------------------
        if (empty) {
            p.hdr_n[i] = 0; p.hdr_m2[i] = 0; p.hdr_lu[i] = -1; p.hdr_lv[i] = -1;
            if (p.out_status) p.out_status[i] = (unsigned char)TLC_ST_DISCONNECTED;
            double* out = p.out_pi + (size_t)i * 25;
#pragma unroll
            for (int c = 0; c < 25; ++c) out[c] = 0.0;
        }
        unsigned hdr = 0xffffffffu;
        if (act) {
            hdr = (unsigned)n | ((unsigned)m << 8) | ((unsigned)(lu & 0xff) << 16) | ((unsigned)(lv & 0xff) << 24);
            p.hdr_n[i] = n; p.hdr_m2[i] = m2 | TLC_XL_DONE_FLAG; p.hdr_lu[i] = lu; p.hdr_lv[i] = lv;
        }
        // the record: whole rows, as many as the longest edge list of the slot (a lane's rows beyond its own count are not read)
        ((unsigned*)rec)[lane] = hdr;
        {
            int mmax = act ? m : 0;
            for (int o = 32; o; o >>= 1) { const int t = __shfl_xor(mmax, o); mmax = t > mmax ? t : mmax; }
            unsigned* re_ = (unsigned*)(rec + TLC_XL_REC_EDGE_OFF) + lane;
            double* rw_ = (double*)(rec + TLC_XL_REC_W_OFF) + lane;
            for (int e = 0; e < mmax; ++e) { re_[e * 64] = edgeL[e]; rw_[e * 64] = wL[e]; }
        }
        const unsigned long long gb = __ballot(give_back), dn = __ballot(act);
        if (gb) {
            int base = 0;
            const int leader = __builtin_ctzll(gb);
            if (lane == leader) base = atomicAdd(p.ovf_count, __popcll(gb));
            base = __builtin_amdgcn_readlane(base, leader);
            if (give_back) p.ovf_list[base + __popcll(gb & tlc_lanemask_lt())] = i;
        }
        if (dn && lane == (int)__builtin_ctzll(dn)) atomicAdd(p.done_count, __popcll(dn));
        XL_STAMP(7);
        if (pc && lane == 0) atomicAdd(&pc[15], 1ull);
        // (the next slot reuses the LDS rows: every lane reads and writes its own column only, nothing to wait for)
    }
#undef XL_STAMP
}

// ---- the membership tables: one wavefront per node, a lane per ball entry ------------------------------------------------------
// An id goes to the emptier of its two buckets (the other one if that is full); a node where both are full gets fail[x] = 1.
__global__ __launch_bounds__(64) void tlc_ball_hash_kernel(int n_nodes, const int* __restrict__ bptr, const int* __restrict__ bcol,
                                                           const long long* __restrict__ hdesc, int* __restrict__ htab,
                                                           int* __restrict__ fail) {
    const int lane = tlc_lane();
    for (int x = blockIdx.x; x < n_nodes; x += gridDim.x) {
        const int j0 = bptr[x], j1 = bptr[x + 1];
        const long long d = hdesc[x];
        const unsigned mask = (1u << (int)(d & 63)) - 1u;
        int* tab = htab + 4 * (d >> 6);
        for (int j = j0 + lane; j < j1; j += TLC_WAVE) {
            const int y = bcol[j];
            int* p1 = tab + 4 * (size_t)(TLC_XL_H1(y) & mask);
            int* p2 = tab + 4 * (size_t)(TLC_XL_H2(y) & mask);
            int c1 = 0, c2 = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                c1 += __hip_atomic_load(&p1[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != -1;
                c2 += __hip_atomic_load(&p2[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != -1;
            }
            int* first = c1 <= c2 ? p1 : p2;
            int* second = c1 <= c2 ? p2 : p1;
            bool placed = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (!placed && atomicCAS(&first[q], -1, y) == -1) placed = true;
#pragma unroll
            for (int q = 0; q < 4; ++q) if (!placed && atomicCAS(&second[q], -1, y) == -1) placed = true;
            if (!placed) atomicOr(&fail[x], 1);
        }
    }
}

int tlc_launch_ball_hash(int n_nodes, const int* bptr, const int* bcol, const long long* hptr, int* htab, int* fail, void* stream) {
    if (n_nodes <= 0) return TLC_OK;
    hipLaunchKernelGGL(tlc_ball_hash_kernel, dim3(n_nodes < 8192 ? n_nodes : 8192), dim3(64), 0, (hipStream_t)stream, n_nodes, bptr, bcol,
                       hptr, htab, fail);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// grid: slots (of 64 candidates) the launch should cover at once
int tlc_launch_xlane(const TlcXlParams& p, int grid, void* stream) {
    if (grid <= 0) return TLC_OK;
    const size_t lds = XL_LDS * XL_WPB;
    if (lds > 64 * 1024)
        TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_xlane_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(tlc_xlane_kernel, dim3((grid + XL_WPB - 1) / XL_WPB), dim3(64 * XL_WPB), lds, (hipStream_t)stream, p);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// vicinity_dev.h -- device code of the vicinity pass (S = ball(u) & ball(v), induced subgraph), shared by
// vicinity.hip (COUNT / FILL kernels; kept in a header so that a tier kernel could extract a vicinity itself).
#pragma once
#include "tlc_common.h"
#include "tlc_kernels.h"

#define TLC_BIG_ROWS 4   /* long CSR rows a wavefront keeps in flight at once */

namespace {

// exclusive scan of one int per thread over a BW-thread workgroup (BW multiple of 64); *total = sum.  `xw` is an LDS array
// of BW/64 ints.  All threads must call.
template <int BW>
__device__ __forceinline__ int block_escan_i32(int v, int* xw, int* total) {
    const int inc = tlc_wave_iscan_i32(v);
    if (BW == 64) {
        *total = __builtin_amdgcn_readlane(inc, 63);
        return inc - v;
    }
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if (tlc_lane() == 63) xw[wv] = inc;
    __syncthreads();
    int before = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < BW / 64; ++k) {
        const int c = xw[k];
        if (k < wv) before += c;
        tot += c;
    }
    *total = tot;
    return before + inc - v;
}

__device__ __forceinline__ bool bit_test(const unsigned* bits, int b) { return (bits[b >> 5] >> (b & 31)) & 1u; }

// A lane that scans its own short CSR row costs the texture addresser one cycle per load, whatever the load's width: the
// row scans are bound by that rate (64 scattered lanes = 64 cycles per wave instruction), so they fetch 16 bytes per
// load from 4-byte aligned addresses (global memory is in unaligned-access mode; the CSR arrays carry 8 padding entries).
struct __attribute__((packed, aligned(4))) TlcI4 { int v[4]; };
struct __attribute__((packed, aligned(4))) TlcI2 { int x, y; };
struct __attribute__((packed, aligned(8))) TlcD2 { double v[2]; };
__device__ __forceinline__ void load_row8(const int* __restrict__ col, int j0, int (&bb)[8]) {
    const TlcI4 a = *reinterpret_cast<const TlcI4*>(col + j0), b = *reinterpret_cast<const TlcI4*>(col + j0 + 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) { bb[q] = a.v[q]; bb[4 + q] = b.v[q]; }
}
__device__ __forceinline__ void load_row8w(const double* __restrict__ w, int j0, double (&ww)[8]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const TlcD2 d = *reinterpret_cast<const TlcD2*>(w + j0 + 2 * q);
        ww[2 * q] = d.v[0]; ww[2 * q + 1] = d.v[1];
    }
}
__device__ __forceinline__ void row_bounds(const int* __restrict__ rowptr, int a, int& beg, int& end) {
    const TlcI2 r = *reinterpret_cast<const TlcI2*>(rowptr + a);
    beg = r.x; end = r.y;
}

// Set the bits of every neighbour of every node in list[0..count); optionally append newly set nodes to
// `next` (global scratch) through the LDS counter *s_cnt.
__device__ __forceinline__ void expand_rows(const int* __restrict__ list, int count, unsigned* bits,
                                            const int* __restrict__ rowptr, const int* __restrict__ col,
                                            int* next, int* s_cnt) {
    const int lane = tlc_lane();
    for (int base = 0; base < count; base += TLC_WAVE) {
        const int k = base + lane;
        int beg = 0, end = 0;
        if (k < count) {
            const int a = list[k];
            row_bounds(rowptr, a, beg, end);
        }
        const bool big = (end - beg) >= 32;
        if (!big) {
            // eight entries per round trip (a load-use loop would pay one global latency per entry)
            for (int j0 = beg; j0 < end; j0 += 8) {
                int bb[8];
                load_row8(col, j0, bb);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    if (j0 + q < end) {
                        const int b = bb[q];
                        const unsigned bit = 1u << (b & 31);
                        const unsigned old = atomicOr(&bits[b >> 5], bit);
                        if (next && !(old & bit)) next[atomicAdd(s_cnt, 1)] = b;
                    }
                }
            }
        }
        unsigned long long mask = __ballot(big);
        while (mask) {
            const int L = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int bb = __builtin_amdgcn_readlane(beg, L), ee = __builtin_amdgcn_readlane(end, L);
            for (int j = bb + lane; j < ee; j += TLC_WAVE) {
                const int b = col[j];
                const unsigned bit = 1u << (b & 31);
                const unsigned old = atomicOr(&bits[b >> 5], bit);
                if (next && !(old & bit)) next[atomicAdd(s_cnt, 1)] = b;
            }
        }
    }
}

// nx.bfs_edges(graph, root, depth_limit=hop) as a bitmap (riccidist2dgm.py:311-314)
__device__ __forceinline__ void mark_ball(unsigned* bits, int root, int hop, const TlcVicParams& p, int* frontA,
                                          int* frontB, int* s_cnt) {
    const int lane = tlc_lane();
    if (lane == 0) atomicOr(&bits[root >> 5], 1u << (root & 31));
    const int rb = p.rowptr[root], re = p.rowptr[root + 1];
    // level 0: the only frontier node is the root; the next frontier is its CSR row itself
    for (int j = rb + lane; j < re; j += TLC_WAVE) {
        const int b = p.col[j];
        atomicOr(&bits[b >> 5], 1u << (b & 31));
    }
    const int* list = p.col + rb;
    int count = re - rb;
    for (int level = 1; level < hop; ++level) {
        const bool last = (level == hop - 1);
        int* next = last ? nullptr : ((level & 1) ? frontA : frontB);
        if (lane == 0) *s_cnt = 0;
        __syncthreads();
        expand_rows(list, count, bits, p.rowptr, p.col, next, s_cnt);
        __syncthreads();
        if (last) break;
        list = next;
        count = *s_cnt;
        __syncthreads();
        if (count == 0) break;
    }
    __syncthreads();
}

__device__ __forceinline__ int local_id(const unsigned* S, const unsigned short* pref, int b) {
    const int w = b >> 5;
    return (int)pref[w] + __popc(S[w] & ((1u << (b & 31)) - 1u));
}

// One wavefront's batch of rows (lane L = row kbase + L * kstride with CSR range [beg, end), empty for idle lanes): counts
// the entries whose column is in S and, with WRITE, stores them as (k << 16 | local id, weight) from offset t on.  Short
// rows are scanned by their own lane, rows of >= 32 entries by the whole wavefront.  Returns the lane's induced degree.
template <bool WRITE>
__device__ __forceinline__ int induced_batch(int k, int kbase, int kstride, int beg, int end, int t, const unsigned* S,
                                             const unsigned short* pref, const TlcVicParams& p, unsigned* dir, double* lw) {
    const int lane = tlc_lane();
    const bool big = (end - beg) >= 32;
    int cnt = 0;
    if (!big) {
        // eight entries (and, when writing, their weights) per round trip
        for (int j0 = beg; j0 < end; j0 += 8) {
            int bb[8];
            double ww[8];
            load_row8(p.col, j0, bb);
            if (WRITE) load_row8w(p.w, j0, ww);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (j0 + q < end && bit_test(S, bb[q])) {
                    if (WRITE) {
                        dir[t + cnt] = ((unsigned)k << 16) | (unsigned)local_id(S, pref, bb[q]);
                        lw[t + cnt] = ww[q];
                    }
                    ++cnt;
                }
            }
        }
    }
    // Long rows as a sequence of 64-entry chunks, four chunks in flight: all four are requested before any is looked at, so
    // the long rows of a batch (the vicinities of hub pairs are full of them, and a hub's own row is several chunks) cost
    // one global round trip per four chunks instead of one per chunk.
    unsigned long long mask = __ballot(big);
    int curL = -1, cur_j = 0, cur_e = 0, cur_t = 0;       // the row being chunked (wave-uniform)
    int run = 0;
    for (;;) {
        int cL[TLC_BIG_ROWS], cj[TLC_BIG_ROWS], ce[TLC_BIG_ROWS], ct[TLC_BIG_ROWS], bv[TLC_BIG_ROWS];
        bool first[TLC_BIG_ROWS];
        double wv[TLC_BIG_ROWS];
#pragma unroll
        for (int q = 0; q < TLC_BIG_ROWS; ++q) {
            first[q] = false;
            if (curL < 0 || cur_j >= cur_e) {
                curL = -1;
                if (mask) {
                    curL = __builtin_ctzll(mask);
                    mask &= mask - 1;
                    cur_j = __builtin_amdgcn_readlane(beg, curL);
                    cur_e = __builtin_amdgcn_readlane(end, curL);
                    cur_t = __builtin_amdgcn_readlane(t, curL);
                    first[q] = true;
                }
            }
            cL[q] = curL; cj[q] = cur_j; ce[q] = cur_e; ct[q] = cur_t;
            cur_j += TLC_WAVE;
        }
        if (cL[0] < 0) break;
#pragma unroll
        for (int q = 0; q < TLC_BIG_ROWS; ++q) {
            const int j = cj[q] + lane;
            const bool ok = cL[q] >= 0 && j < ce[q];
            bv[q] = ok ? p.col[j] : -1;
            if (WRITE) wv[q] = ok ? p.w[j] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < TLC_BIG_ROWS; ++q) {
            if (cL[q] < 0) continue;                                  // (uniform)
            if (first[q]) run = 0;
            const int b = bv[q];
            const bool in = b >= 0 && bit_test(S, b);
            const unsigned long long m = __ballot(in);
            if (WRITE && in) {
                const int pos = ct[q] + run + __popcll(m & tlc_lanemask_lt());
                dir[pos] = ((unsigned)(kbase + cL[q] * kstride) << 16) | (unsigned)local_id(S, pref, b);
                lw[pos] = wv[q];
            }
            run += __popcll(m);
            if (lane == cL[q]) cnt = run;                             // (the row's last chunk leaves the total)
        }
    }
    return cnt;
}

// Walk the CSR rows of ids[0..n).  WRITE=false: ldeg[k] = induced degree of row k (if ldeg), returns the
// wave-uniform total.  WRITE=true: row k's entries go to dir/lw starting at lrow[k].
// Rows are dealt to the wavefronts round-robin (row k -> wavefront k mod NW): the ids are ascending and, in a graph grown by
// preferential attachment, the low ids are the hubs -- contiguous batches would hand every long row to the first wavefront.
template <bool WRITE, int BW>
__device__ __forceinline__ int induced_rows(const int* __restrict__ ids, int n, const unsigned* S, const unsigned short* pref,
                                            const TlcVicParams& p, int* ldeg_or_lrow, unsigned* dir, double* lw, int* xw) {
    const int lane = tlc_lane();
    constexpr int NW = BW / TLC_WAVE;
    const int wv = (int)(threadIdx.x >> 6);
    int total = 0;
    // four batches of 64 rows per wavefront at a time: their node ids (and write offsets), then their row bounds, are
    // requested together, so the chain id -> row bounds -> columns is paid once per four batches
    constexpr int R = 4;
    for (int b0 = 0; b0 * TLC_WAVE * NW + wv < n; b0 += R) {
        int a[R], beg[R], end[R], t[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int k = ((b0 + r) * TLC_WAVE + lane) * NW + wv;
            a[r] = k < n ? ids[k] : -1;
            t[r] = (WRITE && k < n) ? ldeg_or_lrow[k] : 0;
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            beg[r] = 0; end[r] = 0;
            if (a[r] >= 0) row_bounds(p.rowptr, a[r], beg[r], end[r]);
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int kbase = (b0 + r) * TLC_WAVE * NW + wv;          // row of lane 0
            if (kbase >= n) continue;                                 // (uniform)
            const int k = kbase + lane * NW;
            const int cnt = induced_batch<WRITE>(k, kbase, NW, beg[r], end[r], t[r], S, pref, p, dir, lw);
            if (!WRITE) {
                if (ldeg_or_lrow && k < n) ldeg_or_lrow[k] = cnt;
                total += cnt;
            }
        }
    }
    if (WRITE) return 0;
    int tot = 0;
    block_escan_i32<BW>(total, xw, &tot);
    return tot;
}

}  // namespace

// Both balls at once for hop <= 2 (the only values the reference uses, TLCGNN.py:102): the two root rows are read
// together and their concatenation is the level-1 frontier, so the pair pays ONE chain of dependent global loads
// (root rows -> neighbour rows) instead of two.
template <int BW>
__device__ __forceinline__ void mark_two_balls_hop2(unsigned* bitsU, unsigned* bitsV, int u, int v, int hop, const TlcVicParams& p) {
    const int lane = tlc_lane();
    const int ub = p.rowptr[u], ue = p.rowptr[u + 1], vb = p.rowptr[v], ve = p.rowptr[v + 1];
    const int du = ue - ub, dv = ve - vb, tot = du + dv;
    if (threadIdx.x == 0) {
        atomicOr(&bitsU[u >> 5], 1u << (u & 31));
        atomicOr(&bitsV[v >> 5], 1u << (v & 31));
    }
    // frontier entries dealt to the wavefronts round-robin (see induced_rows: the low ids, first in every row, are the hubs)
    constexpr int NW = BW / TLC_WAVE;
    const int wv = (int)(threadIdx.x >> 6);
    for (int kbase = wv; kbase < tot; kbase += BW) {
        const int k = kbase + lane * NW;
        int beg = 0, end = 0;
        unsigned* bits = bitsU;
        if (k < tot) {
            const bool isu = k < du;
            const int a = p.col[isu ? ub + k : vb + (k - du)];
            bits = isu ? bitsU : bitsV;
            atomicOr(&bits[a >> 5], 1u << (a & 31));
            if (hop >= 2) row_bounds(p.rowptr, a, beg, end);
        }
        if (hop < 2) continue;
        const bool big = (end - beg) >= 32;
        if (!big) {
            for (int j0 = beg; j0 < end; j0 += 8) {         // eight entries per round trip
                int bb[8];
                load_row8(p.col, j0, bb);
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (j0 + q < end) atomicOr(&bits[bb[q] >> 5], 1u << (bb[q] & 31));
            }
        }
        // long rows as 64-entry chunks, four in flight (see induced_batch)
        unsigned long long mask = __ballot(big);
        int curL = -1, cur_j = 0, cur_e = 0;
        for (;;) {
            int cL[TLC_BIG_ROWS], bv[TLC_BIG_ROWS], cj[TLC_BIG_ROWS], ce[TLC_BIG_ROWS];
#pragma unroll
            for (int q = 0; q < TLC_BIG_ROWS; ++q) {
                if (curL < 0 || cur_j >= cur_e) {
                    curL = -1;
                    if (mask) {
                        curL = __builtin_ctzll(mask);
                        mask &= mask - 1;
                        cur_j = __builtin_amdgcn_readlane(beg, curL);
                        cur_e = __builtin_amdgcn_readlane(end, curL);
                    }
                }
                cL[q] = curL; cj[q] = cur_j; ce[q] = cur_e;
                cur_j += TLC_WAVE;
            }
            if (cL[0] < 0) break;
#pragma unroll
            for (int q = 0; q < TLC_BIG_ROWS; ++q) {
                const int j = cj[q] + lane;
                bv[q] = (cL[q] >= 0 && j < ce[q]) ? p.col[j] : -1;
            }
#pragma unroll
            for (int q = 0; q < TLC_BIG_ROWS; ++q) {
                if (cL[q] < 0) continue;
                unsigned* wb = ((kbase + cL[q] * NW) < du) ? bitsU : bitsV;
                if (bv[q] >= 0) atomicOr(&wb[bv[q] >> 5], 1u << (bv[q] & 31));
            }
        }
    }
    __syncthreads();
}

// One pair: S = ball(u) & ball(v), the induced rows, and (FILL; or COUNT for a SMALL-tier vicinity -> its fixed slot, a MID / MEDIUM one -> a bump-allocated arena offset, a LARGE one in the
// early pass -> a slot of the early arena) the packed subgraph.
// `lds` = the workgroup's bitmap area (TlcVicParams::nw words x 2.5 + 24), `slot` = its global scratch slot.
template <bool FILL, int BW>
__device__ __forceinline__ void vicinity_pair(const TlcVicParams& p, int i, unsigned* lds, int* slot) {
    const int nw4 = (p.nw + 3) & ~3;          // bitmaps padded to 16 bytes
    unsigned* bitsU = lds;
    unsigned* bitsV = lds + nw4;   // becomes S
    unsigned short* pref = (unsigned short*)(lds + 2 * nw4);   // 16 bits: a vicinity has at most 65 535 nodes
    int* s_cnt = (int*)(lds + 2 * nw4 + nw4 / 2);
    int* xw = s_cnt + 4;                       // BW/64 ints for the block scans
    const int tid = threadIdx.x;
    const int cap = p.n_nodes < TLC_MAX_SUBGRAPH_NODES + 1 ? p.n_nodes : TLC_MAX_SUBGRAPH_NODES + 1;
    int* ids = slot;                             // ascending node ids of S (a vicinity has at most 65 535 nodes)
    int* lrow = slot + cap;                      // induced degree, then row offsets
    int* frontA = slot + 2 * (size_t)cap;        // BFS frontiers: hop >= 3 only (the host allocates them then)
    int* frontB = frontA + p.n_nodes;
    const int res2 = p.res * p.res;
    const int wpl = (p.nw + BW - 1) / BW;      // bitmap words per thread (contiguous chunk)

#ifdef TLC_PHASE_DEBUG
#define VSTAMP(k) do { if (!FILL && p.dbg && tid == 0) { const unsigned long long _t = clock64(); atomicAdd(&p.dbg[k], _t - vt); vt = _t; } } while (0)
#else
#define VSTAMP(k) do { } while (0)
#endif
#ifdef TLC_PHASE_DEBUG
    unsigned long long vt = clock64();
#endif
    const int u = p.pairs[2 * (size_t)i], v = p.pairs[2 * (size_t)i + 1];
    if (FILL) {
        const int hn = p.hdr_n[i];
        if (hn <= 0) return;    // finished by the COUNT pass
        const int hm = p.hdr_m2[i] >> 1;
        if (p.small_dir && hn <= TLC_S_NMAX && hm <= TLC_S_MMAX) return;                         // written by COUNT
        if (p.fill_mode == 2 && (hn > TLC_M_NMAX || hm > TLC_M_MMAX)) return;                    // filled by the early pass
    } else {
        // KeyError on dict_node (riccidist2dgm.py:353): ids the edge-built graph does not contain
        bool missing = u < 0 || v < 0 || u >= p.n_nodes || v >= p.n_nodes;
        if (!missing) missing = (p.rowptr[u + 1] == p.rowptr[u]) || (p.rowptr[v + 1] == p.rowptr[v]);
        if (missing) {
            if (tid == 0) {
                p.hdr_n[i] = 0; p.hdr_m2[i] = 0; p.hdr_lu[i] = -1; p.hdr_lv[i] = -1;
                if (p.out_status) p.out_status[i] = TLC_ST_MISSING_NODE;
                if (p.out_n) p.out_n[i] = 0;
                if (p.out_m) p.out_m[i] = 0;
            }
            if (p.out_pi) for (int c = tid; c < res2; c += BW) p.out_pi[(size_t)i * res2 + c] = 0.0;
            return;
        }
    }
    // ---- the two balls -------------------------------------------------------------------------
    {
        uint4* z = reinterpret_cast<uint4*>(lds);
        for (int w = tid; w < (2 * nw4) / 4; w += BW) z[w] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    if (p.hop <= 2) {
        mark_two_balls_hop2<BW>(bitsU, bitsV, u, v, p.hop, p);
    } else if (BW == 64) {                              // generic depth: single-wavefront workgroups only (host enforces)
        mark_ball(bitsV, v, p.hop, p, frontA, frontB, s_cnt);
        mark_ball(bitsU, u, p.hop, p, frontA, frontB, s_cnt);
    }
    if ((p.flags & TLC_INCLUDE_ROOTS) && tid == 0) {   // data_utils_LP.py:111  nodes + [u, v]
        bitsU[u >> 5] |= 1u << (u & 31); bitsV[u >> 5] |= 1u << (u & 31);
        bitsU[v >> 5] |= 1u << (v & 31); bitsV[v >> 5] |= 1u << (v & 31);
    }
    __syncthreads();
    VSTAMP(0);
    // ---- S = ball(u) & ball(v)  (:315), popcount prefix and the ascending id list in one sweep: every lane owns a
    //      contiguous chunk of bitmap words, one wave scan links the chunks
    const int w0 = tid * wpl < p.nw ? tid * wpl : p.nw, w1 = (w0 + wpl) < p.nw ? (w0 + wpl) : p.nw;
    int mycnt = 0;
    for (int w = w0; w < w1; ++w) {
        const unsigned s = bitsU[w] & bitsV[w];
        bitsV[w] = s;
        mycnt += __popc(s);
    }
    int n = 0;
    const int excl = block_escan_i32<BW>(mycnt, xw, &n);
    {
        int o = excl;
        for (int w = w0; w < w1; ++w) {
            unsigned s = bitsV[w];
            pref[w] = (unsigned short)o;
            while (s && n <= TLC_MAX_SUBGRAPH_NODES) {          // (an oversized vicinity is rejected below; its list has no room)
                const int b = __builtin_ctz(s);
                s &= s - 1;
                ids[o++] = (w << 5) + b;
            }
        }
    }
    __syncthreads();
    VSTAMP(1);
    const unsigned* S = bitsV;
    if (!FILL) {
        int lu = -1, lv = -1;
        if (n > 0) {
            if (bit_test(S, u)) lu = local_id(S, pref, u);
            if (bit_test(S, v)) lv = local_id(S, pref, v);
        }
        if (n == 0 || n > 65535) {
            // n == 0: AssertionError, zero connected components (:318).  n > 65535 does not fit the packed local ids:
            // reported as its own status so that it cannot pass silently.
            if (tid == 0) {
                p.hdr_n[i] = 0; p.hdr_m2[i] = 0; p.hdr_lu[i] = lu; p.hdr_lv[i] = lv;
                if (p.out_status) p.out_status[i] = (n == 0) ? TLC_ST_DISCONNECTED : TLC_ST_TOO_LARGE;
                if (p.out_n) p.out_n[i] = (n == 0) ? 0 : -n;
                if (p.out_m) p.out_m[i] = 0;
            }
            if (p.out_pi) for (int c = tid; c < res2; c += BW) p.out_pi[(size_t)i * res2 + c] = 0.0;
            return;
        }
        if (tid == 0) { p.hdr_lu[i] = lu; p.hdr_lv[i] = lv; }
    }
    if (p.out_ids && (FILL || p.small_dir)) {
        // id output of tlc_vicinity_filtration: by whichever pass finishes the pair's subgraph (COUNT: the SMALL tier's, and with a
        // bump allocator every vicinity up to the MEDIUM tier's node count -- one whose edges then do not fit is written again by FILL)
        const bool mine = FILL ? true : (n <= TLC_S_NMAX || (p.bump_top != nullptr && n <= TLC_M_NMAX));
        if (mine) {
            const long long no = p.ids_off[i];
            const long long cap = p.ids_off[i + 1] - no;
            if (n <= cap) for (int k = tid; k < n; k += BW) p.out_ids[no + k] = ids[k];
        }
    }
    // ---- induced subgraph (graph.subgraph(nodes), :316) -------------------------------------------------
    // A vicinity of <= 64 nodes is one row per lane of a single wavefront: the row bounds and induced degrees stay in
    // registers between the counting and the writing pass (no row-offset hand-off through global scratch, no second
    // read of the id list and the row pointers); 9 of 10 pairs take this path in COUNT.
    const bool reg_rows = (BW == 64) && n <= TLC_WAVE;
    int rb = 0, re = 0, rcnt = 0;
    int m2;
    if (reg_rows) {
        if (tid < n) row_bounds(p.rowptr, ids[tid], rb, re);
        rcnt = induced_batch<false>(tid, 0, 1, rb, re, 0, S, pref, p, nullptr, nullptr);
        m2 = tlc_wave_sum_i32(rcnt);
    } else {
        m2 = induced_rows<false, BW>(ids, n, S, pref, p, lrow, nullptr, nullptr, xw);
    }
    VSTAMP(2);
    bool write = FILL;
    unsigned* wdir = nullptr;
    double* wlw = nullptr;
    if (!FILL) {
        if ((m2 >> 1) > TLC_MAX_SUBGRAPH_EDGES) {            // edge ranks are packed in 24 bits (pd_pipeline.hip, cycle swap)
            if (tid == 0) {
                p.hdr_n[i] = 0; p.hdr_m2[i] = 0;
                if (p.out_status) p.out_status[i] = TLC_ST_TOO_LARGE;
                if (p.out_n) p.out_n[i] = -n;
                if (p.out_m) p.out_m[i] = 0;
            }
            if (p.out_pi) for (int c = tid; c < res2; c += BW) p.out_pi[(size_t)i * res2 + c] = 0.0;
            __syncthreads();
            return;
        }
        if (tid == 0) { p.hdr_n[i] = n; p.hdr_m2[i] = m2; }
        // small vicinities are finished right here: fixed-size slot, no second kernel pass over this pair
        if (p.small_dir && n <= TLC_S_NMAX && (m2 >> 1) <= TLC_S_MMAX) {
            write = true;
            wdir = p.small_dir + (size_t)i * (2 * TLC_S_MMAX);
            wlw = p.small_lw + (size_t)i * (2 * TLC_S_MMAX);
        } else if (p.bump_top && n <= TLC_M_NMAX && (m2 >> 1) <= TLC_M_MMAX) {
            // MID / MEDIUM tier: written right here at a bump-allocated arena offset
            long long* s_off = (long long*)(s_cnt + 2);
            __syncthreads();
            if (tid == 0) {
                long long off = (long long)atomicAdd(p.bump_top, (unsigned long long)m2);
                if (off + m2 > p.bump_cap) { off = -1; atomicAdd(p.bump_overflow, 1); }
                p.edge_off[i] = off;
                *s_off = off;
            }
            __syncthreads();
            const long long off = *s_off;
            if (off >= 0) {
                write = true;
                wdir = p.A_dir + off;
                wlw = p.A_lw + off;
            }
        } else if (p.early_list) {
            // early pass: a LARGE-tier vicinity takes a slot of the early arena and is written right away
            const int m = m2 >> 1;
            if ((n > TLC_M_NMAX || m > TLC_M_MMAX) && n <= TLC_L_NMAX && m <= TLC_L_MMAX) {
                __syncthreads();
                if (tid == 0) s_cnt[3] = atomicAdd(p.early_count, 1);
                __syncthreads();
                const int es = s_cnt[3];
                if (es < p.early_cap) {
                    if (tid == 0) p.early_list[es] = i;
                    write = true;
                    wdir = p.early_dir + (size_t)es * (2 * TLC_L_MMAX);
                    wlw = p.early_lw + (size_t)es * (2 * TLC_L_MMAX);
                }
            }
        }
    } else {
        const long long eo = p.edge_off[i];
        wdir = p.A_dir + eo;
        wlw = p.A_lw + eo;
    }
    if (write && reg_rows) {
        const int t0 = tlc_wave_iscan_i32(rcnt) - rcnt;
        induced_batch<true>(tid, 0, 1, rb, re, t0, S, pref, p, wdir, wlw);
    } else if (write) {
        __syncthreads();
        int run = 0;
        for (int base = 0; base < n; base += BW) {
            const int k = base + tid;
            const int d = k < n ? lrow[k] : 0;
            int tot = 0;
            const int ex = block_escan_i32<BW>(d, xw, &tot);
            if (k < n) lrow[k] = run + ex;
            run += tot;
        }
        __syncthreads();
        induced_rows<true, BW>(ids, n, S, pref, p, lrow, wdir, wlw, xw);
    }
    __syncthreads();
    VSTAMP(3);
#ifdef TLC_PHASE_DEBUG
    if (!FILL && p.dbg && tid == 0) atomicAdd(&p.dbg[4], 1ull);
#endif
}
#undef VSTAMP


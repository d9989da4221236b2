// extract.hip -- P4 of the hot path (hop <= 2 are the only values the reference uses, TLCGNN.py:102; round 5: any hop), second form:
// S = ball(u) & ball(v) from PRECOMPUTED BALL LISTS, the induced subgraph in one sweep that skips hub rows.
//
// Replaces sg2dgm_accelerate's BFS + set intersection + graph.subgraph (sg2dgm/riccidist2dgm.py:311-316), like
// vicinity.hip (which stays as the `extract` = 0 path, for graphs whose bitmaps do not fit the LDS, the FILL pass of the heavy tiers and the variant entry points).
//
// What the per-phase counters of the first form said (PubMed-shaped batch, 37 676 pairs): 28 % of COUNT is the two
// breadth-first searches, 70 % the two sweeps over the CSR rows of the members of S -- and 80 % of the entries those
// sweeps read are not in S: half of them sit in the rows of a few dozen hub nodes that happen to be members.
//   * ball_hop(x) is a property of the graph: it is listed once per (graph, hop) as sorted node ids (1.06 M entries =
//     4 MB for PubMed, hop 2).  A pair marks the larger list in an LDS bitmap and filters the smaller one through it:
//     two coalesced streams instead of a two-level row expansion, S comes out as an ascending id list (= local ids).
//   * the order of the packed directed entries carries no meaning downstream (the tier kernels run Bellman-Ford over the
//     entry list and sort the undirected edges by key), so an entry y -> x need not come from row y.  Rows of HEAVY
//     members (the <= 256 highest degrees >= 32, fixed per graph; the two roots excepted -- all of N(u) lies in S when
//     v is adjacent) are not read at all: a scanned row that finds a heavy member emits both directions, and
//     heavy-heavy entries come from a dense K x K weight table.
//   * one sweep: entries are written as they are found (positions from wave scans) at the cursor of an arena region
//     that belongs to the workgroup (a fixed region per workgroup, further blocks from a bump counter when it is full): the
//     place is known before the size is, so no vicinity up to the MEDIUM tier is swept twice, no per-pair allocation
//     atomic, and the fixed 3 KB slots of the SMALL tier are gone (every tier kernel reads arena + edge_off).
//   * longest first: per-pair wall-clock stamps (tools/pair_times.py) showed 90 % of the pairs done after 55 % of the
//     kernel's time, the rest being a thin tail of MEDIUM-tier vicinities (50-70 us on one wavefront) that started late.
//     min(|ball(u)|, |ball(v)|) bounds |S| and predicts it well, so tlc_classify_kernel bins the pairs by it ahead of
//     the extraction and the workgroups take the bins in descending order before the rest.
#include "vicinity_dev.h"

#define TLC_X_HV_CAP 2048     /* local ids covered by the heavy-member bits; larger vicinities scan every row */
#define TLC_X_H_CAP 128       /* heavy members listed per vicinity; more: every row is scanned */
#define TLC_X_BLOCK 4096      /* arena entries a workgroup takes from the bump counter when its region is full (>= 2 * TLC_M_MMAX) */
#define TLC_X_BIN_MIN 64      /* pairs whose smaller ball has at least this many nodes are binned and extracted first */
#define TLC_X_COUNTERS 8      /* work counters of the main pass (tlc_extract_kernel) */
#define TLC_X_COUNTER_STRIDE 64 /* ints between them: atomics on ONE 128-byte line are served one after the other (~90/us) */

namespace {

template <int BW>
struct XCfg {
    static constexpr int SID_CAP = (BW == 64) ? 256 : 2048;   // member ids kept in LDS; beyond: the global scratch slot
};

#define TLC_X_UNITS (1024 + TLC_X_H_CAP * TLC_X_H_CAP / 64)    /* units of a sweep: 64-member batches + 64-pair heavy chunks */
#ifndef TLC_X_STAGE
#define TLC_X_STAGE 128       /* entries a wavefront stages in LDS before it writes them out in order (0: every lane stores its own) */
#endif
struct XLayout {
    size_t o_pref, o_sid, o_hvy, o_hl, o_ctl, o_ucnt, o_stage, o_mt, total;
};
#define TLC_X_MT_BYTES ((TLC_BE_CAP / 64) * 16)      /* member table of x_sweep_ball: per 64 positions {mask u64, members below u32, pad} */
__host__ __device__ __forceinline__ constexpr XLayout x_layout(int nw, int sid_cap, int bw, bool fast = false) {
    XLayout L{};
    const size_t nw4 = (size_t)((nw + 3) & ~3);
    if (fast) {
        // tlc_extract_kernel<64, true>: the bitmap of the larger ball, the member ids, the control words -- no word ranks, no heavy
        // bits, no staging buffer (x_sweep_ball needs none of them): 3.6 KB for PubMed, so that LDS does not cap its wavefronts
        L.o_pref = nw4 * 4;
        L.o_sid = (L.o_pref + 15) & ~(size_t)15;
        L.o_hvy = L.o_hl = L.o_ctl = L.o_sid + (size_t)sid_cap * 4;
        L.o_ucnt = L.o_ctl + 64 + 4 + 16;
        L.o_stage = (L.o_ucnt + 15) & ~(size_t)15;
        L.o_mt = L.o_stage;
        L.total = L.o_mt + TLC_X_MT_BYTES;
        return L;
    }
    L.o_pref = nw4 * 4;
    L.o_sid = (L.o_pref + nw4 * 2 + 15) & ~(size_t)15;
    L.o_hvy = L.o_sid + (size_t)sid_cap * 4;
    L.o_hl = L.o_hvy + TLC_X_HV_CAP / 8;
    L.o_ctl = L.o_hl + (size_t)TLC_X_H_CAP * 4;
    L.o_ucnt = L.o_ctl + 64 + (size_t)(bw / 64) * 4 + 16;
    L.o_stage = (L.o_ucnt + (bw > 64 ? (size_t)TLC_X_UNITS * 4 : 0) + 15) & ~(size_t)15;
    L.o_mt = L.o_stage + (size_t)(bw / 64) * TLC_X_STAGE * 12;       // per wavefront: [TLC_X_STAGE doubles | TLC_X_STAGE words]
    L.total = L.o_mt + TLC_X_MT_BYTES;
    return L;
}

struct XState {
    unsigned* bits;
    unsigned short* pref;
    unsigned* hvy;
    unsigned* hl;          // (local id << 16) | heavy index
    int* ucnt;             // several wavefronts: entries / first entry number of every unit of a sweep
    unsigned char* stage;  // the wavefronts' staging buffers (x_sweep_wave)
    int* ctl;              // [0] lu [1] lv [2] entry counter (BW > 64) [3] heavy members [4] early slot [6..7] block grab; [8..11] region cursor / end (2 x i64, live across pairs)
    int* xw;
};

__device__ __forceinline__ bool x_member(const XState& X, int y, int& ly) {
    const unsigned word = X.bits[y >> 5];
    const unsigned bit = 1u << (y & 31);
    if (!(word & bit)) return false;
    ly = (int)X.pref[y >> 5] + __popc(word & (bit - 1u));
    return true;
}
// a member's node record in registers (one 64-byte line: four 16-byte loads).  Scalars only: a struct with arrays that is
// assigned as a whole goes through scratch memory.
struct XRec {
    int rb, deg, hidx, n_in;
    int c0, c1, c2, c3;
    double w0, w1, w2, w3;
    __device__ __forceinline__ int c(int q) const { return q == 0 ? c0 : (q == 1 ? c1 : (q == 2 ? c2 : c3)); }
    __device__ __forceinline__ double w(int q) const { return q == 0 ? w0 : (q == 1 ? w1 : (q == 2 ? w2 : w3)); }
};
__device__ __forceinline__ XRec x_empty_rec() {
    XRec R;
    R.rb = R.deg = R.n_in = 0; R.hidx = -1;
    R.c0 = R.c1 = R.c2 = R.c3 = 0;
    R.w0 = R.w1 = R.w2 = R.w3 = 0.0;
    return R;
}
__device__ __forceinline__ XRec x_load_rec(const TlcNodeRec* r) {
    XRec R;
    const TlcI4 h = *reinterpret_cast<const TlcI4*>(r), c = *reinterpret_cast<const TlcI4*>(&r->col[0]);
    const TlcD2 wa = *reinterpret_cast<const TlcD2*>(&r->w[0]), wb = *reinterpret_cast<const TlcD2*>(&r->w[2]);
    R.rb = h.v[0]; R.deg = h.v[1]; R.hidx = h.v[2]; R.n_in = h.v[3];
    R.c0 = c.v[0]; R.c1 = c.v[1]; R.c2 = c.v[2]; R.c3 = c.v[3];
    R.w0 = wa.v[0]; R.w1 = wa.v[1]; R.w2 = wb.v[0]; R.w3 = wb.v[1];
    return R;
}

// Workgroup-wide ordering of LDS traffic.  A single wavefront executes its LDS instructions in issue order, so it needs neither
// a barrier nor -- what __syncthreads() also costs -- an s_waitcnt vmcnt(0): the global stores of one pair's entries would be
// waited for (0.7 us under load) at every synchronisation point of the next one.  Only the compiler must keep the order.
template <int BW>
__device__ __forceinline__ void x_sync() {
    if (BW == 64) {
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}
// arena entries: ordinary stores.  (Nontemporal stores -- the entries are written once and read once, by another kernel, and
// 37 MB of them per batch pass through a 4 MB L2 beside the graph's 6 MB of lists and rows -- were measured: same kernel time,
// but WRITE_SIZE 142 MB instead of 51 MB per batch: the partial lines a wavefront's few entries make are not merged on the way
// out.  make X_NT=1 brings them back for an A/B.)
template <typename T>
#ifdef TLC_X_NT_STORE
__device__ __forceinline__ void x_store(T* p, T v) { __builtin_nontemporal_store(v, p); }
#else
__device__ __forceinline__ void x_store(T* p, T v) { *p = v; }
#endif
__device__ __forceinline__ bool x_heavy(const XState& X, int ly) { return (X.hvy[ly >> 5] >> (ly & 31)) & 1u; }

// One sweep over the rows of the members ids[0..n): every directed entry of the induced subgraph exactly once, as
// (src local id << 16 | dst local id, weight).
//
// The ORDER of the entries carries no meaning downstream, but it decides how equal sort keys fall and with that the last bits
// of an image, so it is made canonical: the same for every workgroup width, for the early pass, the main pass and the FILL
// pass.  The sweep is a sequence of UNITS -- batch b of 64 consecutive members, then chunk c of 64 consecutive ordered pairs of
// heavy members -- and a unit's entries are numbered the way ONE wavefront finds them (short rows by eight-entry round,
// lane-major inside a round; then the long rows in lane order).  A single wavefront walks the units in order and writes as
// it counts.  Several wavefronts take the units round-robin: they count every unit into ucnt[], a prefix over the units
// places them, and a second walk writes.
// Walks the units of this wavefront; ucnt (several wavefronts): WR ? the first entry number of every unit : receives the
// count of every unit.  Returns the entries this wavefront counted.
template <int BW, bool WR, class PB>
__device__ __forceinline__ int x_sweep_wave(const PB& p, const int* ids, int n, const XState& X, bool use_hvy, int nH,
                                            unsigned* dir, double* lw, int cap, int* ucnt, const XRec pre, bool has_pre, int dbg_i = -1) {
    const int lane = tlc_lane(), wv = (int)(threadIdx.x >> 6);
    constexpr int NW = BW / 64;
    int run = 0, counted = 0;
#ifdef TLC_PAIR_TIMES
#define SSTAMP(k) do { if (WR && dbg_i >= 0 && p.dbg_pair_t && threadIdx.x == 0) p.dbg_pair_t[16 * (size_t)dbg_i + (k)] = wall_clock64(); } while (0)
#else
#define SSTAMP(k) do { } while (0)
#endif
    SSTAMP(8);
    // Staged writes.  A lane that finds q entries in its row writes them at ITS offsets: one store instruction per position of the
    // row round, every active lane at an address of its own -- and the texture addresser takes one cycle per such lane (counters:
    // 41.6 M store lane-accesses per batch, half of everything the kernel asks of it, at 0.85 lane-accesses per cycle and CU).  So the
    // entries of the short-row rounds go to an LDS buffer at their entry number and leave it in order, 64 consecutive entries per
    // store.  [st_fb, st_fb + st_fill) are the entry numbers the buffer holds; the dense paths (long rows, heavy pairs) number
    // their entries by ballot already and store directly.
    double* const st_w = reinterpret_cast<double*>(X.stage + (size_t)wv * TLC_X_STAGE * 12);
    unsigned* const st_d = reinterpret_cast<unsigned*>(st_w + TLC_X_STAGE);
    int st_fb = 0, st_fill = 0;
    auto st_flush = [&]() {
        if (!WR || TLC_X_STAGE == 0 || st_fill == 0) return;
        x_sync<64>();
        for (int k = lane; k < st_fill; k += TLC_WAVE) {
            const int e = st_fb + k;
            if (e < cap) { x_store(&dir[e], st_d[k]); x_store(&lw[e], st_w[k]); }
        }
        x_sync<64>();
        st_fill = 0;
    };
    // the round's entries are numbered [r0, r0 + tot): true = they go through the buffer
    auto st_round = [&](int r0, int tot) -> bool {
        if (!WR || TLC_X_STAGE == 0) return false;
        if (st_fill && (r0 != st_fb + st_fill || st_fill + tot > TLC_X_STAGE)) st_flush();
        if (tot > TLC_X_STAGE) return false;
        if (!st_fill) st_fb = r0;
        st_fill += tot;
        return true;
    };
    auto st_put = [&](bool staged, int off, unsigned d, double w) {
        if (staged) { st_d[off - st_fb] = d; st_w[off - st_fb] = w; }
        else if (off < cap) { x_store(&dir[off], d); x_store(&lw[off], w); }
    };
    const int nb = (n + 63) >> 6;
    for (int b = wv; b < nb; b += NW) {
        if (NW > 1) { if (WR) run = ucnt[b]; else run = 0; }
        const int k = (b << 6) + lane;
        const bool act = k < n;
        // the member's node record: row start, degree and its first four entries in one 64-byte line (most nodes have no more);
        // batch 0 of a single wavefront arrives with it (loaded when the member bits were set)
        int rb = 0, re = 0, n_in = 0;
        XRec R = x_empty_rec();
        if (act && !(use_hvy && x_heavy(X, k))) {
            if (has_pre && b == 0) R = pre; else R = x_load_rec(p.nrec + ids[k]);
            rb = R.rb; re = rb + R.deg; n_in = R.n_in;
        }
        const bool big = (re - rb) >= 32;
        // ---- short rows, round 0: the entries held in the record --------------------------------------------------------
        {
            int ly[4];
            unsigned hit = 0u, rev = 0u;
            if (!big) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    ly[q] = 0;
                    if (q < n_in && x_member(X, R.c(q), ly[q])) {
                        hit |= 1u << q;
                        if (use_hvy && x_heavy(X, ly[q])) rev |= 1u << q;
                    }
                }
            }
            const int cnt = __popc(hit) + __popc(rev);
            const int incl = tlc_wave_iscan_i32(cnt);
            int off = run + incl - cnt;
            const int tot = __builtin_amdgcn_readlane(incl, 63);
            const bool staged = st_round(run, tot);
            run += tot;
            if (WR && cnt) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if ((hit >> q) & 1u) {
                        const double wq = R.w(q);
                        st_put(staged, off, ((unsigned)k << 16) | (unsigned)ly[q], wq);
                        ++off;
                        if ((rev >> q) & 1u) {
                            st_put(staged, off, ((unsigned)ly[q] << 16) | (unsigned)k, wq);
                            ++off;
                        }
                    }
                }
            }
        }
        SSTAMP(10);
        // ---- short rows beyond the record, in eight-entry SEGMENTS (segment r of a member = entries 8r .. 8r+7 behind the record's;
        //      a short row has at most four).  The entries are numbered by round r, by member inside a round, by position inside
        //      a segment (what a lane that walks its own row round after round finds).  That walk costs one global round trip per
        //      round of the LONGEST row while most lanes idle (per-pair stamps: 2.8 of the 8.3 us of a <= 16-node vicinity, 5.6 of
        //      12 us up to 64 nodes), so the segments of the whole batch of members are dealt to the lanes in exactly that order
        //      -- lane s takes the s-th segment -- and a plain scan over the lanes numbers the entries: one round trip per 64
        //      segments, typically one per batch of members.
        {
            const int rest0 = big ? re : rb + n_in;                       // (heavy / inactive members: rb == re == 0)
            const int segs = (re - rest0 + 7) >> 3;
            const unsigned long long m0 = __ballot(segs > 0), m1 = __ballot(segs > 1), m2 = __ballot(segs > 2), m3 = __ballot(segs > 3);
            const int b1 = __popcll(m0), b2 = b1 + __popcll(m1), b3 = b2 + __popcll(m2), total = b3 + __popcll(m3);
            for (int s0 = 0; s0 < total; s0 += TLC_WAVE) {
                const int sx = s0 + lane;
                const bool have = sx < total;
                const int r = (sx >= b1) + (sx >= b2) + (sx >= b3);
                const unsigned long long mm = r == 0 ? m0 : (r == 1 ? m1 : (r == 2 ? m2 : m3));
                int kth = sx - (r == 0 ? 0 : (r == 1 ? b1 : (r == 2 ? b2 : b3)));
                // the lane of the kth member that has a segment r: kth set bit of mm
                int src = 0;
                {
                    unsigned w32 = (unsigned)mm;
                    const int c32 = __popc(w32);
                    if (kth >= c32) { kth -= c32; src = 32; w32 = (unsigned)(mm >> 32); }
#pragma unroll
                    for (int sh = 16; sh; sh >>= 1) {
                        const unsigned low = w32 & ((1u << sh) - 1u);
                        const int c = __popc(low);
                        if (kth >= c) { kth -= c; w32 >>= sh; src += sh; } else w32 = low;
                    }
                }
                if (!have) src = lane;
                const int m_rest = __shfl(rest0, src), m_re = __shfl(re, src);
                const int kk = (b << 6) + src;
                const int j0 = m_rest + 8 * r;
                int bb[8], ly[8];
                double ww[8];
                unsigned hit = 0u, rev = 0u;
                if (have) {
                    load_row8(p.col, j0, bb);
                    if (WR) load_row8w(p.w, j0, ww);
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        ly[q] = 0;
                        if (j0 + q < m_re && x_member(X, bb[q], ly[q])) {
                            hit |= 1u << q;
                            if (use_hvy && x_heavy(X, ly[q])) rev |= 1u << q;
                        }
                    }
                }
                const int cnt = __popc(hit) + __popc(rev);
                const int incl = tlc_wave_iscan_i32(cnt);
                int off = run + incl - cnt;
                const int tot = __builtin_amdgcn_readlane(incl, 63);
                const bool staged = st_round(run, tot);
                run += tot;
                if (WR && cnt) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        if ((hit >> q) & 1u) {
                            st_put(staged, off, ((unsigned)kk << 16) | (unsigned)ly[q], ww[q]);
                            ++off;
                            if ((rev >> q) & 1u) {
                                st_put(staged, off, ((unsigned)ly[q] << 16) | (unsigned)kk, ww[q]);
                                ++off;
                            }
                        }
                    }
                }
            }
        }
        SSTAMP(11);
        // ---- long scanned rows (a root that is a hub; a high degree outside the heavy set): the wavefront streams the row,
        //      four 64-entry chunks in flight ---------------------------------------------------------------------------
        unsigned long long mask = __ballot(act && big);
        while (mask) {
            const int L = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int jb = __builtin_amdgcn_readlane(rb, L), je = __builtin_amdgcn_readlane(re, L);
            const int kk = (b << 6) + L;
            for (int j = jb; j < je; j += 4 * TLC_WAVE) {
                int cv[4];
                double wvv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int jj = j + r * TLC_WAVE + lane;
                    cv[r] = jj < je ? p.col[jj] : -1;
                    wvv[r] = (WR && jj < je) ? p.w[jj] : 0.0;
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (j + r * TLC_WAVE >= je) break;                 // (uniform)
                    int l = 0;
                    const bool h = cv[r] >= 0 && x_member(X, cv[r], l);
                    const bool rv = h && use_hvy && x_heavy(X, l);
                    const unsigned long long m1 = __ballot(h), m2 = __ballot(rv);
                    const int c1 = __popcll(m1);
                    if (WR && h) {
                        const double wq = wvv[r];
                        const int o1 = run + __popcll(m1 & tlc_lanemask_lt());
                        if (o1 < cap) { x_store(&dir[o1], ((unsigned)kk << 16) | (unsigned)l); x_store(&lw[o1], wq); }
                        if (rv) {
                            const int o2 = run + c1 + __popcll(m2 & tlc_lanemask_lt());
                            if (o2 < cap) { x_store(&dir[o2], ((unsigned)l << 16) | (unsigned)kk); x_store(&lw[o2], wq); }
                        }
                    }
                    run += c1 + __popcll(m2);
                }
            }
        }
        if (NW > 1) {
            if (!WR) { if (lane == 0) ucnt[b] = run; counted += run; }
        }
    }
    st_flush();
    SSTAMP(12);
    // ---- heavy x heavy: ordered pairs of listed heavy members through the dense weight table (0 = not adjacent) ------------
    if (use_hvy && nH > 0) {
        const int np = nH * nH, nc = (np + 63) >> 6;
        for (int c = wv; c < nc; c += NW) {
            if (NW > 1) { if (WR) run = ucnt[nb + c]; else run = 0; }
            const int t = (c << 6) + lane;
            double w = 0.0;
            unsigned ea = 0u, eb = 0u;
            if (t < np) {
                ea = X.hl[t / nH]; eb = X.hl[t % nH];
                w = p.hh_w[(size_t)(ea & 0xffffu) * p.hh_k + (eb & 0xffffu)];
            }
            const bool h = w > 0.0;
            const unsigned long long m1 = __ballot(h);
            if (WR && h) {
                const int o1 = run + __popcll(m1 & tlc_lanemask_lt());
                if (o1 < cap) { x_store(&dir[o1], ((ea >> 16) << 16) | (eb >> 16)); x_store(&lw[o1], w); }
            }
            run += __popcll(m1);
            if (NW > 1 && !WR) { if (lane == 0) ucnt[nb + c] = run; counted += run; }
        }
    }
    SSTAMP(13);
#undef SSTAMP
    return NW > 1 ? counted : run;
}

// Round 5.  The same entries from the SUBGRAPH LIST of the smaller ball (TlcVicParams::be_ptr): S is a subset of that ball, so
// the induced subgraph is the list's entries whose two ends are members -- M0, M1 hold the members as bit masks over the positions
// of the ball list (|ball| <= 128: the two ballots of the filter; up to TLC_BE_CAP = 512: a table of the ballots in LDS), a local id is
// the number of members below a position.  One coalesced 4 + 8-byte stream instead of a node record per member, its row segments, the heavy-member bookkeeping and the member bitmap:
// a lane's entry is kept or dropped by two bit tests, numbered by a ballot and stored -- 99.6 % of the PubMed batch's pairs
// (smaller ball <= 512 nodes; for the 84 % up to 64 nodes: 51 directed list entries on average, up to ~1 300 at 512 nodes).  Order: sources ascending, a source's
// entries in CSR order -- a function of the pair alone, whichever pass (main, FILL) sweeps it.  One wavefront; dir may be null.
// TAB: the masks come from the LDS table mt (|ball| > TLC_BE_REG_CAP: up to eight ballots): entry p >> 6 = {mask, members below}.
template <bool TAB>
__device__ __forceinline__ int x_sweep_ball(const unsigned* __restrict__ be_pos, const double* __restrict__ be_w, int e0, int e1,
                                            unsigned long long M0, unsigned long long M1, const uint4* mt, unsigned* dir, double* lw, int cap) {
    const int lane = tlc_lane();
    const unsigned c0 = (unsigned)__popcll(M0);
    int run = 0;
    // position p of the ball list: is it a member, and how many members stand below it (its local id)
    auto look = [&](unsigned p, bool& mem, unsigned& rk) {
        const unsigned long long below = (1ull << (p & 63u)) - 1ull;
        if (TAB) {
            const uint4 e = mt[p >> 6];
            const unsigned long long m = ((unsigned long long)e.y << 32) | e.x;
            mem = ((m >> (p & 63u)) & 1ull) != 0ull;
            rk = e.z + (unsigned)__popcll(m & below);
        } else {
            const unsigned long long m = (p & 64u) ? M1 : M0;
            mem = ((m >> (p & 63u)) & 1ull) != 0ull;
            rk = ((p & 64u) ? c0 : 0u) + (unsigned)__popcll(m & below);
        }
    };
    // TLC_X_BALL_UNROLL chunks of 64 entries are requested together (a list of 1 300 entries was 20 trips to L2 one after the
    // other; 84 % of the pairs have one chunk and are not touched by this)
#ifndef TLC_X_BALL_UNROLL
#define TLC_X_BALL_UNROLL 4
#endif
    constexpr int UN = TLC_X_BALL_UNROLL;
    for (int j0 = e0; j0 < e1; j0 += UN * TLC_WAVE) {
        unsigned pos_[UN];
        double w_[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int j = j0 + u * TLC_WAVE + lane;
            const bool in = j < e1;
            pos_[u] = in ? be_pos[j] : 0u;
            w_[u] = (in && dir) ? be_w[j] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            if (j0 + u * TLC_WAVE >= e1) break;                      // (uniform)
            const bool in = j0 + u * TLC_WAVE + lane < e1;
            const unsigned pos = pos_[u];
            bool ma, mb;
            unsigned ra, rb;
            look(pos >> 16, ma, ra);
            look(pos & 0xffffu, mb, rb);
            const bool keep = in && ma && mb;
            const unsigned long long K = __ballot(keep);
            if (dir && keep) {
                const int o = run + __popcll(K & tlc_lanemask_lt());
                if (o < cap) { x_store(&dir[o], (ra << 16) | rb); x_store(&lw[o], w_[u]); }
            }
            run += __popcll(K);
        }
    }
    return run;
}

// The sweep of a workgroup; dir may be null (count only).  Returns the number of entries (uniform over the workgroup).
template <int BW, class PB>
__device__ __forceinline__ int x_sweep(const PB& p, const int* ids, int n, const XState& X, bool use_hvy, int nH,
                                       unsigned* dir, double* lw, int cap, const XRec pre, bool has_pre, int dbg_i = -1) {
    if (BW == 64) {
        if (dir) return x_sweep_wave<BW, true>(p, ids, n, X, use_hvy, nH, dir, lw, cap, nullptr, pre, has_pre, dbg_i);
        return x_sweep_wave<BW, false>(p, ids, n, X, use_hvy, nH, nullptr, nullptr, 0, nullptr, pre, has_pre);
    }
    int* ucnt = X.ucnt;
    x_sweep_wave<BW, false>(p, ids, n, X, use_hvy, nH, nullptr, nullptr, 0, ucnt, pre, false);
    x_sync<BW>();
    // exclusive prefix over the units (at most TLC_X_UNITS of them), a few consecutive units per thread
    const int nu = ((n + 63) >> 6) + ((use_hvy && nH > 0) ? ((nH * nH + 63) >> 6) : 0);
    const int per = (nu + BW - 1) / BW;
    const int u0 = (int)threadIdx.x * per;
    int mine = 0;
    for (int q = 0; q < per; ++q) if (u0 + q < nu) mine += ucnt[u0 + q];
    int total = 0;
    int off = block_escan_i32<BW>(mine, X.xw, &total);
    for (int q = 0; q < per; ++q) if (u0 + q < nu) { const int c = ucnt[u0 + q]; ucnt[u0 + q] = off; off += c; }
    x_sync<BW>();
    if (dir) x_sweep_wave<BW, true>(p, ids, n, X, use_hvy, nH, dir, lw, cap, ucnt, pre, false);
    x_sync<BW>();
    return total;
}

template <int BW, class PB>
__device__ __forceinline__ void x_zero_row(const PB& p, int i, int status, int n_report, int lu, int lv) {
    const int tid = threadIdx.x, res2 = p.res * p.res;
    if (tid == 0) {
        p.hdr_n[i] = 0; p.hdr_m2[i] = 0; p.hdr_lu[i] = lu; p.hdr_lv[i] = lv;
        if (p.out_status) p.out_status[i] = (unsigned char)status;
        if (p.out_n) p.out_n[i] = n_report;
        if (p.out_m) p.out_m[i] = 0;
    }
    if (p.out_pi) for (int c = tid; c < res2; c += BW) p.out_pi[(size_t)i * res2 + c] = 0.0;
}

// One pair.  The bitmap is all zero on entry and on exit.
// The head of an item -- the pair and the bounds of its rows and ball lists -- as scalars: loaded by the kernel's loop one item
// ahead of the item's body (tlc_extract_kernel).  Bounds are zero for a pair with an id out of range.
struct XHead {
    int u, v;
    int ru0, ru1, rv0, rv1, a0, a1, b0, b1;
    int eu0, eu1, ev0, ev1;            // the ball-subgraph lists of u and v (TlcVicParams::be_ptr; zero without them)
};
// The first chunks of a pair's two ball lists, asked for while the pair BEFORE it is swept (single-wavefront workgroups): lane l
// holds entry l of the smaller list (b) and entries l, 64 + l of the larger one (a), -1 beyond their ends.  (Three registers: with
// four chunks of each the kernel spilled 40 -- and a spilled prefetch register is a wait for the load in front of the sweep.)
struct XNoPrefetch {
    __device__ __forceinline__ void operator()() const {}
};
// FAST (BW == 64): the launch that takes the pairs x_sweep_ball serves (TlcVicParams::fast_split) and nothing else -- the general
// sweep, the member bitmap and the heavy-member bookkeeping are compiled out, which is what lets it run at twice the wavefronts.
template <int BW, bool FAST = false, class PB, class PF = XNoPrefetch>
__device__ __forceinline__ void extract_pair(const PB& p, int i, bool from_rest, const XHead& H, unsigned char* lds, int* slot,
                                             bool has_pre = false, int pb0 = -1, int pa0 = -1, int pa1 = -1,
                                             const PF& prefetch_next = PF()) {
    static_assert(!FAST || BW == 64, "the FAST launch is one wavefront per pair");
    constexpr int SID_CAP = FAST ? TLC_BE_CAP + 8 : XCfg<BW>::SID_CAP;   // (the FAST launch never touches its scratch slot)
    const XLayout L = x_layout(p.nw, SID_CAP, BW, FAST);
    const int nw4 = (p.nw + 3) & ~3;
    XState X;
    X.bits = (unsigned*)lds;
    X.pref = (unsigned short*)(lds + L.o_pref);
    int* sid = (int*)(lds + L.o_sid);
    X.hvy = (unsigned*)(lds + L.o_hvy);
    X.hl = (unsigned*)(lds + L.o_hl);
    X.ctl = (int*)(lds + L.o_ctl);
    X.xw = X.ctl + 16;
    X.ucnt = (int*)(lds + L.o_ucnt);
    X.stage = lds + L.o_stage;
    // (laundered: values derived from the thread index -- shifted copies, list addresses -- are otherwise computed once in front of
    // the item loop and held in registers across the whole body, where the sweep then spills)
    int tid_ = threadIdx.x;
    asm volatile("" : "+v"(tid_));
    const int tid = tid_;
#ifdef TLC_PAIR_TIMES
    // (per-pair wall-clock stamps only: sums through global atomics on one address would serialise the whole kernel)
#define XSTAMP(k) do { if (p.dbg_pair_t && tid == 0) p.dbg_pair_t[16 * (size_t)i + (k)] = wall_clock64(); } while (0)
#else
#define XSTAMP(k) do { } while (0)
#endif
#ifdef TLC_PAIR_TIMES
    const unsigned long long t_start = wall_clock64();
#endif
#ifndef TLC_NO_FAST_ASSUME
    if constexpr (FAST) {
        // what api.hip's subgraph-list launch always passes (run_chunk_front, `fp`): told to the compiler, the branches on them and the
        // scalar loads of the fields behind them leave the per-pair code (every field of p is re-read per pair, see the kernel)
        __builtin_assume(p.x_fill == 0); __builtin_assume(p.fill_mode == 0); __builtin_assume(p.early_list == nullptr);
        __builtin_assume(p.skip_count == nullptr); __builtin_assume(p.big_count == nullptr); __builtin_assume(p.be_ptr != nullptr);
        __builtin_assume((p.flags & TLC_INCLUDE_ROOTS) == 0u); __builtin_assume(p.bump_top != nullptr);
    }
#endif
    const int u = H.u, v = H.v;
    // KeyError on dict_node (riccidist2dgm.py:353): ids the edge-built graph does not contain
    bool missing = u < 0 || v < 0 || u >= p.n_nodes || v >= p.n_nodes;
    int a0 = 0, a1 = 0, b0 = 0, b1 = 0;
    if (!missing) {
        a0 = H.a0; a1 = H.a1; b0 = H.b0; b1 = H.b1;
        missing = (H.ru1 == H.ru0) || (H.rv1 == H.rv0);
        if (!missing && from_rest && !FAST) {
            const int na = a1 - a0, nb = b1 - b0, mn = na < nb ? na : nb;
            // same predicates as tlc_classify_kernel: the early pass owns its candidates (as long as its list held them all;
            // else they are extracted here); a binned pair was taken from its bin
            if (p.skip_count && mn >= p.skip_threshold) {
                if (*p.skip_count <= p.skip_cap) return;
            } else if (p.big_count && mn >= TLC_X_BIN_MIN) return;
        }
    }
    // FILL pass (the scan has laid the chunk out after an arena overflow; a heavy tier the early pass did not take): the
    // headers exist, the entries go to edge_off[i] -- in the same canonical order as everywhere else
    int fill_m2 = 0;
    if (p.x_fill) {
        const int hn = p.hdr_n[i];
        if (hn <= 0) return;                                             // finished by the COUNT pass
        fill_m2 = p.hdr_m2[i];
        if (p.fill_mode == 2 && (hn > TLC_M_NMAX || (fill_m2 >> 1) > TLC_M_MMAX)) return;   // filled from the heavy tiers' lists
    }
    if (missing) {
        // (one owner: the FAST launch when the chunk has one)
        if (FAST || !p.fast_split || p.x_fill) x_zero_row<BW>(p, i, TLC_ST_MISSING_NODE, 0, -1, -1);
        return;
    }
#ifdef TLC_PAIR_TIMES
    if (p.dbg_pair_t && tid == 0) { p.dbg_pair_t[16 * (size_t)i] = t_start; p.dbg_pair_t[16 * (size_t)i + 1] = wall_clock64(); p.dbg_pair_t[16 * (size_t)i + 14] = blockIdx.x; }
#endif
    if (a1 - a0 < b1 - b0) { int t = a0; a0 = b0; b0 = t; t = a1; a1 = b1; b1 = t; }    // [a0,a1): the larger ball
    const int nA = a1 - a0, nB = b1 - b0;
    // the subgraph list of the smaller ball's node (b is v's list unless u's is strictly shorter), when there is one: x_sweep_ball
    int be0 = 0, be1 = 0;
    bool fast = false;
    if constexpr (BW == 64) {
        // (not a candidate of the early pass: with lists up to 512 nodes a smaller ball of 511 or 512 nodes is both -- the FAST launch runs
        // beside the classification and cannot wait for its verdict, so it leaves every such pair to the early pass / the general launch)
        if (p.be_ptr && nB <= TLC_BE_CAP && !(p.flags & TLC_INCLUDE_ROOTS) && !(p.early_min_ball > 0 && nB >= p.early_min_ball)) {
            const bool b_is_u = (H.a1 - H.a0) < (H.b1 - H.b0);
            be0 = b_is_u ? H.eu0 : H.ev0; be1 = b_is_u ? H.eu1 : H.ev1;
            fast = true;
        }
    }
    // two launches share a chunk's pairs (fast_split): the FAST one takes exactly the pairs the subgraph lists serve, whatever bin
    // they are in; this one the rest (and, as the FILL pass, everything: x_fill)
    if constexpr (FAST) { if (!fast) return; }
    else if (fast && p.fast_split && !p.x_fill) return;
    unsigned long long M0 = 0ull, M1 = 0ull;               // (fast) the members of S as a mask over the smaller ball's positions
    uint4* const mtab = reinterpret_cast<uint4*>(lds + L.o_mt);
    // ---- S = ball(u) & ball(v) (:315): the smaller list filtered through a bitmap of the larger -----------------------------
    int bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {                       // (the first chunks of the smaller list are in flight while the larger is marked)
        const int j = r * BW + tid;
        if (has_pre && r == 0) bv[r] = pb0;             // (uniform) came with the call
        else bv[r] = j < nB ? p.bcol[b0 + j] : -1;
    }
    // (FAST launch with ball bitmaps: the larger ball is not marked at all -- its row of p.bbits answers the membership test below)
    const unsigned* bbrow = nullptr;
    if constexpr (FAST) {
        if (p.bbits) bbrow = p.bbits + (size_t)(((H.a1 - H.a0) < (H.b1 - H.b0)) ? v : u) * (size_t)p.bb_nw;   // (the node whose ball is [a0, a1))
    }
    bool first = has_pre;
    for (int j = tid; j < (bbrow ? 0 : nA); j += 4 * BW) {
        int av[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (first && r < 2) av[r] = r == 0 ? pa0 : pa1;
            else av[r] = (j + r * BW < nA) ? p.bcol[a0 + j + r * BW] : -1;
        }
        first = false;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (av[r] >= 0) atomicOr(&X.bits[av[r] >> 5], 1u << (av[r] & 31));
    }
    if (tid == 0) { X.ctl[0] = -1; X.ctl[1] = -1; }
    x_sync<BW>();
    XSTAMP(2);
    const int capg = p.n_nodes < TLC_MAX_SUBGRAPH_NODES + 1 ? p.n_nodes : TLC_MAX_SUBGRAPH_NODES + 1;
    // (TLC_INCLUDE_ROOTS may add the two roots to the list: room for them)
    const bool incl = (p.flags & TLC_INCLUDE_ROOTS) != 0u;
    const bool in_lds = nB + (incl ? 2 : 0) <= SID_CAP;
    int* ids = in_lds ? sid : slot;
    const int cap_ids = in_lds ? SID_CAP : capg;
    int n = 0;
    for (int base = 0; base < nB; base += 4 * BW) {
        if (base > 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int j = base + r * BW + tid;
                bv[r] = j < nB ? p.bcol[b0 + j] : -1;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (base + r * BW >= nB) break;                              // (uniform)
            const int b = bv[r];
            const bool h = b >= 0 && (bbrow ? bit_test(bbrow, b) : bit_test(X.bits, b));
            int pos, tot;
            if (BW == 64) {
                const unsigned long long m = __ballot(h);
                pos = n + __popcll(m & tlc_lanemask_lt());
                tot = __popcll(m);
                if (base == 0 && r == 0) M0 = m;
                if (base == 0 && r == 1) M1 = m;
                if (fast && nB > TLC_BE_REG_CAP && tid == 0)             // (uniform) the member table of x_sweep_ball<true>
                    mtab[(base >> 6) + r] = make_uint4((unsigned)m, (unsigned)(m >> 32), (unsigned)n, 0u);
            } else {
                pos = n + block_escan_i32<BW>(h ? 1 : 0, X.xw, &tot);
            }
            if (h) {
                if (pos < cap_ids) ids[pos] = b;
                if (b == u) X.ctl[0] = pos;
                if (b == v) X.ctl[1] = pos;
            }
            n += tot;
        }
    }
    x_sync<BW>();
    XSTAMP(3);
    // the marks of the larger ball go (by list when that is the shorter way)
    if (bbrow) {                                          // (nothing was marked)
    } else if (nw4 > 4096 && nA * 8 < nw4) {              // (re-reading the list is a global round trip: only for a large bitmap)
        for (int j = tid; j < nA; j += BW) X.bits[p.bcol[a0 + j] >> 5] = 0u;
    } else {
        uint4* z = reinterpret_cast<uint4*>(X.bits);
        for (int w = tid; w < nw4 / 4; w += BW) z[w] = make_uint4(0u, 0u, 0u, 0u);
    }
    int lu = X.ctl[0], lv = X.ctl[1];
    x_sync<BW>();
    if (incl && lu < 0 && u != v) {
        // data_utils_LP.py:111, nodes + [u, v]: u is in ball(v) exactly when v is in ball(u), so either both roots are members
        // already or neither is (d(u, v) > hop) -- then both go into the ascending list: element k moves up by the number of roots
        // below it (top chunk first: a chunk is read, then written, and lands on slots whose elements have moved already)
        const int r1 = u < v ? u : v, r2 = u < v ? v : u;
        int c1 = 0, c2 = 0;
        for (int k = tid; k < n && k < cap_ids; k += BW) { const int x = ids[k]; c1 += x < r1 ? 1 : 0; c2 += x < r2 ? 1 : 0; }
        if (BW == 64) { c1 = tlc_wave_sum_i32(c1); c2 = tlc_wave_sum_i32(c2); }
        else {
            int t1, t2;
            block_escan_i32<BW>(c1, X.xw, &t1);
            x_sync<BW>();
            block_escan_i32<BW>(c2, X.xw, &t2);
            x_sync<BW>();
            c1 = t1; c2 = t2;
        }
        if (n + 2 <= cap_ids) {
            for (int base = n > 0 ? ((n - 1) / BW) * BW : -1; base >= 0; base -= BW) {
                const int k = base + tid;
                const int x = k < n ? ids[k] : 0;
                x_sync<BW>();
                if (k < n) ids[k + (x > r1 ? 1 : 0) + (x > r2 ? 1 : 0)] = x;
                x_sync<BW>();
            }
            if (tid == 0) { ids[c1] = r1; ids[c2 + 1] = r2; }
        }
        n += 2;
        lu = u < v ? c1 : c2 + 1;
        lv = u < v ? c2 + 1 : c1;
        x_sync<BW>();
    }
    XSTAMP(4);
    if (n == 0 || n > TLC_MAX_SUBGRAPH_NODES) {
        // n == 0: AssertionError, zero connected components (:318).  n > 65535 does not fit the packed local ids: its own status
        x_zero_row<BW>(p, i, n == 0 ? TLC_ST_DISCONNECTED : TLC_ST_TOO_LARGE, n == 0 ? 0 : -n, lu, lv);
        return;
    }
    // id output of tlc_vicinity_filtration: by whichever pass finishes the pair's subgraph -- this one every vicinity up to the
    // MEDIUM tier's node count (one whose edges then do not fit is written again by the FILL pass), FILL all of its own
    if (p.out_ids && (p.x_fill || n <= TLC_M_NMAX)) {
        const long long no = p.ids_off[i];
        const long long cap_o = p.ids_off[i + 1] - no;
        if (n <= cap_o) for (int k = tid; k < n; k += BW) p.out_ids[no + k] = ids[k];
    }
    // ---- member bits, the rank of the first member of every touched bitmap word, heavy members ------------------------------
    const bool heavy_ok = !FAST && !fast && p.hh_k > 0 && n <= TLC_X_HV_CAP;
    if (heavy_ok) for (int w = tid; w < TLC_X_HV_CAP / 32; w += BW) X.hvy[w] = 0u;
    x_sync<BW>();
    int nH = 0;
    XRec rec0 = x_empty_rec();
    for (int base = 0; base < ((FAST || fast) ? 0 : n); base += BW) {
        const int k = base + tid;
        int hi = -1;
        if (k < n) {
            const int x = ids[k];
            if (BW == 64 && base == 0) {                       // (kept for the sweep)
                rec0 = x_load_rec(p.nrec + x);
                if (heavy_ok && x != u && x != v) hi = rec0.hidx;
            } else if (heavy_ok && x != u && x != v) hi = p.nrec[x].hidx;
            const int w = x >> 5;
            atomicOr(&X.bits[w], 1u << (x & 31));
            if (k == 0 || (ids[k - 1] >> 5) != w) X.pref[w] = (unsigned short)k;
        }
        if (heavy_ok) {                                      // (uniform) the heavy members, listed in member order
            int q, tot;
            if (BW == 64) {
                const unsigned long long mk = __ballot(hi >= 0);
                q = nH + __popcll(mk & tlc_lanemask_lt());
                tot = __popcll(mk);
            } else {
                q = nH + block_escan_i32<BW>(hi >= 0 ? 1 : 0, X.xw, &tot);
            }
            if (hi >= 0) {
                atomicOr(&X.hvy[k >> 5], 1u << (k & 31));
                if (q < TLC_X_H_CAP) X.hl[q] = ((unsigned)k << 16) | (unsigned)hi;
            }
            nH += tot;
        }
    }
    x_sync<BW>();
    const bool use_hvy = heavy_ok && nH <= TLC_X_H_CAP;      // (more heavy members than the list holds: every row is scanned)
    XSTAMP(5);
    // (one place for the two forms of the sweep)
    auto sweep = [&](unsigned* d_, double* w_, int cap_, int dbg_i) -> int {
        if (FAST || fast) {
            if (nB > TLC_BE_REG_CAP) return x_sweep_ball<true>(p.be_pos, p.be_w, be0, be1, 0ull, 0ull, mtab, d_, w_, cap_);
            return x_sweep_ball<false>(p.be_pos, p.be_w, be0, be1, M0, M1, nullptr, d_, w_, cap_);
        }
        if constexpr (!FAST) return x_sweep<BW>(p, ids, n, X, use_hvy, nH, d_, w_, cap_, rec0, BW == 64, dbg_i);
        return 0;
    };
    if (p.x_fill) {
        const long long eo = p.edge_off[i];
        sweep(p.A_dir + eo, p.A_lw + eo, fill_m2, -1);
        if (fast) return;                                                // (no member bits were set)
        x_sync<BW>();
        for (int k = tid; k < n; k += BW) X.bits[ids[k] >> 5] = 0u;
        x_sync<BW>();
        return;
    }
    // ---- induced subgraph (graph.subgraph(nodes), :316): one sweep, written at the cursor of this workgroup's arena region ----
    // (room for the largest vicinity the region serves -- the MEDIUM tier's 2 x 1024 entries, or n^2 -- is secured first)
    long long* cur = (long long*)(X.ctl + 8);            // [0] cursor [1] end of the current block
    unsigned* wdir = nullptr;
    double* wlw = nullptr;
    int cap = 0;
    long long at = -1;
    const bool early_large_ok = p.early_list != nullptr && n <= TLC_L_NMAX;
    if (p.bump_top && n <= TLC_M_NMAX) {
        const long long nn = (long long)n * n;
        cap = nn < 2 * TLC_M_MMAX ? (int)nn : 2 * TLC_M_MMAX;
        if (fast && be1 - be0 < cap) cap = be1 - be0;     // (the subgraph list bounds the vicinity's entries: its region lasts longer)
        if (cur[1] - cur[0] < cap) {                     // (uniform: LDS state)
            x_sync<BW>();
            if (tid == 0) {
                long long off = p.bump_base + (long long)atomicAdd(p.bump_top, (unsigned long long)TLC_X_BLOCK);
                if (off + TLC_X_BLOCK > p.bump_cap) { off = -1; atomicAdd(p.bump_overflow, 1); }
                cur[0] = off; cur[1] = off < 0 ? -1 : off + TLC_X_BLOCK;
            }
            x_sync<BW>();
        }
        at = cur[1] - cur[0] >= cap ? cur[0] : -1;
        if (at >= 0) { wdir = p.A_dir + at; wlw = p.A_lw + at; } else cap = 0;
    }
    int es = -1;
    if (!wdir && early_large_ok && n > TLC_M_NMAX) {
        // early pass: a vicinity beyond the MEDIUM tier takes a slot of the early arena and is written right away
        if (tid == 0) X.ctl[4] = atomicAdd(p.early_count, 1);
        x_sync<BW>();
        es = X.ctl[4];
        if (es < p.early_cap) {
            wdir = p.early_dir + (size_t)es * (2 * TLC_L_MMAX);
            wlw = p.early_lw + (size_t)es * (2 * TLC_L_MMAX);
            cap = 2 * TLC_L_MMAX;
        }
    }
    prefetch_next();                                      // (the next pair's ball lists travel while this one's rows are swept)
    const int m2 = sweep(wdir, wlw, cap, i);
    XSTAMP(6);
    const int m = m2 >> 1;
    if (m > TLC_MAX_SUBGRAPH_EDGES) {                     // edge ranks are packed in 24 bits (pd_pipeline.hip, cycle swap)
        x_zero_row<BW>(p, i, TLC_ST_TOO_LARGE, -n, lu, lv);
    } else {
        if (tid == 0) { p.hdr_n[i] = n; p.hdr_m2[i] = m2; p.hdr_lu[i] = lu; p.hdr_lv[i] = lv; }
        if (at >= 0 && m <= TLC_M_MMAX) {
            // the entries stand where they were written
            if (tid == 0) { p.edge_off[i] = at; cur[0] = at + m2; }
        } else if (p.bump_top && n <= TLC_M_NMAX && m <= TLC_M_MMAX) {
            if (tid == 0) p.edge_off[i] = -1;             // (no room: the overflow flag is up, the chunk is redone by scan + FILL)
        } else if (n <= TLC_M_NMAX && m > TLC_M_MMAX && early_large_ok && m <= TLC_L_MMAX) {
            // few nodes, many edges: beyond the MEDIUM tier by its edge count only -- swept again into an early slot
            if (tid == 0) X.ctl[4] = atomicAdd(p.early_count, 1);
            x_sync<BW>();
            es = X.ctl[4];
            if (es < p.early_cap) {
                sweep(p.early_dir + (size_t)es * (2 * TLC_L_MMAX), p.early_lw + (size_t)es * (2 * TLC_L_MMAX), 2 * TLC_L_MMAX, -1);
                if (tid == 0) p.early_list[es] = i;
            }
        } else if (es >= 0 && es < p.early_cap) {
            // (an early slot holds a vicinity of the LARGE tier only; one that turned out HUGE by its edges leaves the slot unused)
            if (m <= TLC_L_MMAX) { if (tid == 0) p.early_list[es] = i; }
            else if (tid == 0) p.early_list[es] = -1;
        }
    }
    // ---- the member bits go, by list -----------------------------------------------------------------------------------------
    if (!FAST && !fast) {
        x_sync<BW>();
        for (int k = tid; k < n; k += BW) X.bits[ids[k] >> 5] = 0u;
        x_sync<BW>();
    }
    XSTAMP(7);
#undef XSTAMP
}

}  // namespace

#ifndef TLC_X_WPE
#define TLC_X_WPE 4
#endif
#ifndef TLC_X_NOPF
#define TLC_X_NOPF 0
#endif
#ifndef TLC_X_PIPE
#define TLC_X_PIPE 0          /* 0: the item loop without the stream / list prefetch (A/B) */
#endif
typedef const __attribute__((address_space(4))) TlcVicParams XParams;
typedef const __attribute__((address_space(4))) int XCInt;
template <int BW, bool FAST>
__global__ __launch_bounds__(BW, BW == 64 ? (FAST ? 8 : TLC_X_WPE) : 1) void tlc_extract_kernel(TlcVicParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char xlds[];
    int* slot = p.scratch + (size_t)(p.scratch_base_slot + blockIdx.x) * p.scratch_stride;
    {
        const int nw4 = (p.nw + 3) & ~3;
        uint4* z = reinterpret_cast<uint4*>(xlds);
        for (int w = threadIdx.x; w < nw4 / 4; w += BW) z[w] = make_uint4(0u, 0u, 0u, 0u);
        __syncthreads();
    }
    // this workgroup's arena region (cursor / end live in LDS across its pairs)
    {
        const XLayout L = x_layout(p.nw, FAST ? TLC_BE_CAP + 8 : XCfg<BW>::SID_CAP, BW, FAST);
        long long* cur = (long long*)((int*)(xlds + L.o_ctl) + 8);
        if (threadIdx.x == 0) {
            const long long b = p.region_base_entries + (long long)(p.region_base_wg + (int)blockIdx.x) * p.region_entries;
            cur[0] = b; cur[1] = b + p.region_entries;
            if (!p.bump_top) cur[1] = cur[0] = 0;
        }
        __syncthreads();
    }
    // Work order.  Early pass: the candidate list (length on the device).  Main pass: the bins of tlc_classify_kernel in
    // descending order, then every pair by index (binned pairs and the early pass's candidates drop out at once); items are
    // dealt in strided chunks, the first chunk of a workgroup being its own index and the rest coming from a counter (one
    // dequeue per chunk: a single counter serves ~90 dequeues/us, MI355X_MICROARCH.md) -- so the big items, which sit at
    // the lowest item numbers, are the FIRST item of the first workgroups.
    int c0 = 0, c1 = 0, c2 = 0;
    int n_work = p.n_pairs;
    if (p.fill_mode == 1) {
        n_work = p.fill_count;
        if (p.work_count_dev) { const int c = *p.work_count_dev; n_work = c < n_work ? c : n_work; }
    } else if (p.big_count && !FAST) {
        c0 = p.big_count[0]; c1 = p.big_count[1]; c2 = p.big_count[2];
        // behind a FAST launch only the pairs of bins 0 and 1 are left (smaller ball >= 128 nodes): bin 2 and every pair outside
        // the bins have a smaller ball of < 128 nodes, which the FAST launch owns (TLC_BE_CAP >= 128 >= TLC_X_BIN_MIN)
        // (unless the early pass's candidate list overflowed: the candidates it could not hold are reached by index only)
        if (p.fast_split && !p.x_fill && !(p.skip_count && *p.skip_count > p.skip_cap)) { c2 = 0; n_work = 0; }
        n_work += c0 + c1 + c2;
    }
    // The parameter block is read through the kernel-argument segment pointer, laundered per item: every field is a scalar load
    // inside the body instead of a value hoisted in front of the persistent loop (which had cost 110 spilled SGPRs).
    // The head of an item is a chain of dependent loads -- list position -> pair index -> pair -> four row bounds -- in front of the
    // first byte of the ball lists: three round trips of ~1-2 us under load in an item of ~12 us.  All of it is uniform over the
    // workgroup, so it is read with SCALAR loads (constant address space: the lists, the pairs and the row pointers do not change
    // while this kernel runs), one stage per item ahead: while item k runs, its own bounds, the pair of item k+1 and the list
    // entry of item k+2 are fetched together -- one exposed round trip per item instead of three.
    XParams* kp = (XParams*)__builtin_amdgcn_kernarg_segment_ptr();   // (p is the kernel's only argument: offset 0)
    auto idx_of = [&](XParams* q, int w, bool& from_rest) -> int {
        from_rest = false;
        if (q->fill_mode == 1) return ((XCInt*)q->fill_list)[w];
        if (w < c0) return ((XCInt*)q->big_list)[w];
        if (w < c0 + c1) return ((XCInt*)q->big_list)[(size_t)q->n_pairs + (w - c0)];
        if (w < c0 + c1 + c2) return ((XCInt*)q->big_list)[2 * (size_t)q->n_pairs + (w - c0 - c1)];
        from_rest = true;
        return w - c0 - c1 - c2;
    };
    auto pair_of = [&](XParams* q, int i, int& u, int& v) {
        XCInt* pr = (XCInt*)q->pairs + 2 * (size_t)i;
        u = pr[0]; v = pr[1];
    };
    auto bounds_of = [&](XParams* q, XHead& H) {
        H.ru0 = H.ru1 = H.rv0 = H.rv1 = H.a0 = H.a1 = H.b0 = H.b1 = 0;
        H.eu0 = H.eu1 = H.ev0 = H.ev1 = 0;
        if (H.u < 0 || H.v < 0 || H.u >= q->n_nodes || H.v >= q->n_nodes) return;
        XCInt* rp = (XCInt*)q->rowptr;
        XCInt* bp = (XCInt*)q->bptr;
        H.ru0 = rp[H.u]; H.ru1 = rp[H.u + 1]; H.rv0 = rp[H.v]; H.rv1 = rp[H.v + 1];
        H.a0 = bp[H.u]; H.a1 = bp[H.u + 1]; H.b0 = bp[H.v]; H.b1 = bp[H.v + 1];
        if (BW == 64 && q->be_ptr) {
            XCInt* ep = (XCInt*)q->be_ptr;
            H.eu0 = ep[H.u]; H.eu1 = ep[H.u + 1]; H.ev0 = ep[H.v]; H.ev1 = ep[H.v + 1];
        }
    };
    if (p.started && threadIdx.x == 0 && (int)blockIdx.x < n_work) atomicAdd(p.started, 1);
    // (one loop for both schedules -- static: a chunk is one item and the next chunk is gridDim.x further on)
    // Dynamic: the first chunk of a workgroup is its own index; the rest come from TLC_X_COUNTERS counters (256 bytes apart), counter k handing out the
    // chunks gridDim.x + k, + TLC_X_COUNTERS, ... -- a workgroup starts at counter blockIdx % TLC_X_COUNTERS and moves on when one runs
    // dry.  (One counter serves ~90 dequeues/us: with it the chunks had to be ~9 pairs = ~100 us, about one per resident wavefront,
    // i.e. no balancing at all -- per-pair stamps showed the last 30 % of the kernel's span at under half occupancy.  Eight counters
    // carry chunks of two pairs.)  The next chunk is requested BEFORE the current one is worked on: its round trip is hidden.
    __shared__ int s_chunk;
    const bool dyn = p.work_counter != nullptr;
    const int n_chunks = dyn ? (n_work + p.work_chunk - 1) / p.work_chunk : n_work;
    int ck = (int)(blockIdx.x % TLC_X_COUNTERS), dry = 0;
    if constexpr (BW == 64 && TLC_X_PIPE != 0) {
        // Single-wavefront workgroups: the items of this workgroup form ONE stream (its chunks one after the other) and the stages
        // run across chunk boundaries, three items ahead: while item k is extracted, the list entry of item k+3, the pair of k+2 and
        // the row bounds of k+1 are fetched (scalar loads), and once item k's own lists have been consumed -- in front of its
        // sweep -- the first 64 / 128 entries of the two ball lists of item k+1 are asked for (vector loads that nobody waits for
        // before the next item starts).  Per-pair stamps had shown 2.4 of the 9.8 us of a <= 16-node pair in front of the first
        // list entry, and the three-load chain at the head of every CHUNK (two pairs) exposed.
        int g_c = (int)blockIdx.x;
        int g_w = g_c < n_chunks ? g_c : -1;                   // the next item the stream hands out
        int g_nxt = 0;                                        // (lane 0) the counter's answer for the chunk after g_c
        if (dyn && g_w >= 0 && threadIdx.x == 0) g_nxt = atomicAdd(p.work_counter + ck * TLC_X_COUNTER_STRIDE, 1);
        auto gen = [&]() -> int {
            const int w = g_w;
            if (w < 0) return -1;
            if (w + n_chunks < n_work) { g_w = w + n_chunks; return w; }
            if (!dyn) g_c += (int)gridDim.x;
            else {
                g_c = (int)gridDim.x + TLC_X_COUNTERS * __builtin_amdgcn_readfirstlane(g_nxt) + ck;
                while (g_c >= n_chunks && ++dry < TLC_X_COUNTERS) {          // (dry counters: see the loop below)
                    ck = (ck + 1) % TLC_X_COUNTERS;
                    int* cnt = p.work_counter + ck * TLC_X_COUNTER_STRIDE;
                    int t = 0;
                    if (threadIdx.x == 0) t = __atomic_load_n(cnt, __ATOMIC_RELAXED);
                    if ((int)gridDim.x + TLC_X_COUNTERS * __builtin_amdgcn_readfirstlane(t) + ck >= n_chunks) continue;
                    if (threadIdx.x == 0) t = atomicAdd(cnt, 1);
                    g_c = (int)gridDim.x + TLC_X_COUNTERS * __builtin_amdgcn_readfirstlane(t) + ck;
                }
            }
            g_w = g_c < n_chunks ? g_c : -1;
            if (dyn && g_w >= 0 && threadIdx.x == 0) g_nxt = atomicAdd(p.work_counter + ck * TLC_X_COUNTER_STRIDE, 1);
            return w;
        };
        bool fr0 = false, fr1 = false, fr2 = false;
        int i0 = -1, i1 = -1, i2 = -1, u1 = -1, v1 = -1;
        XHead H0;
        H0.u = H0.v = -1;
        {
            XParams* q = kp;
            asm volatile("" : "+s"(q));
            const int w0 = gen(), w1 = gen(), w2 = gen();
            if (w0 >= 0) i0 = idx_of(q, w0, fr0);
            if (w1 >= 0) i1 = idx_of(q, w1, fr1);
            if (w2 >= 0) i2 = idx_of(q, w2, fr2);
            if (i0 >= 0) pair_of(q, i0, H0.u, H0.v);
            if (i1 >= 0) pair_of(q, i1, u1, v1);
            bounds_of(q, H0);
        }
        bool pre_ok = false;
        int pb0 = -1, pa0 = -1, pa1 = -1;
        while (i0 >= 0) {
            XParams* q = kp;
            asm volatile("" : "+s"(q));
            bool fr3 = false;
            int i3 = -1, u2 = -1, v2 = -1;
            const int w3 = gen();
            if (w3 >= 0) i3 = idx_of(q, w3, fr3);
            if (i2 >= 0) pair_of(q, i2, u2, v2);
            XHead H1;
            H1.u = u1; H1.v = v1;
            bounds_of(q, H1);
            bool pf_done = false, pre_n = false;
            int npb0 = -1, npa0 = -1, npa1 = -1;
            auto pf = [&]() {
                pf_done = true;
                if (TLC_X_NOPF || i1 < 0 || q->x_fill) return;
                int a0 = H1.a0, a1 = H1.a1, b0 = H1.b0, b1 = H1.b1;          // (all zero for ids outside the graph)
                if (a1 - a0 < b1 - b0) { int t = a0; a0 = b0; b0 = t; t = a1; a1 = b1; b1 = t; }
                const int nA = a1 - a0, nB = b1 - b0;
                if (fr1) {                                                   // (a pair another pass owns: extract_pair's predicates)
                    if (q->skip_count && nB >= q->skip_threshold) return;
                    else if (q->big_count && nB >= TLC_X_BIN_MIN) return;
                }
                const int* bc = q->bcol;
                const int j = (int)threadIdx.x;
                npb0 = j < nB ? bc[b0 + j] : -1;
                npa0 = j < nA ? bc[a0 + j] : -1;
                npa1 = j + 64 < nA ? bc[a0 + j + 64] : -1;
                pre_n = true;
            };
            extract_pair<BW, FAST>(*q, i0, fr0, H0, xlds, slot, pre_ok, pb0, pa0, pa1, pf);
            if (!pf_done) pf();
            pre_ok = pre_n;
            pb0 = npb0; pa0 = npa0; pa1 = npa1;
            // (the carried heads are uniform; said explicitly, or the compiler keeps the eight bounds in vector registers across the body)
#define XU(x) __builtin_amdgcn_readfirstlane(x)
            i0 = XU(i1); fr0 = fr1;
            H0.u = XU(H1.u); H0.v = XU(H1.v);
            H0.ru0 = XU(H1.ru0); H0.ru1 = XU(H1.ru1); H0.rv0 = XU(H1.rv0); H0.rv1 = XU(H1.rv1);
            H0.a0 = XU(H1.a0); H0.a1 = XU(H1.a1); H0.b0 = XU(H1.b0); H0.b1 = XU(H1.b1);
            H0.eu0 = XU(H1.eu0); H0.eu1 = XU(H1.eu1); H0.ev0 = XU(H1.ev0); H0.ev1 = XU(H1.ev1);
            i1 = XU(i2); fr1 = fr2; u1 = XU(u2); v1 = XU(v2);
            i2 = XU(i3); fr2 = fr3;
#undef XU
        }
        return;
    }
    for (int c = blockIdx.x; c < n_chunks;) {
        int nxt = 0;
        if (dyn && BW == 64 && threadIdx.x == 0) nxt = atomicAdd(p.work_counter + ck * TLC_X_COUNTER_STRIDE, 1);
        // the items of a chunk: c, c + n_chunks, ...; stage registers: [0] the item about to run, [1] the one after it
        bool fr0 = false, fr1 = false;
        int i0 = -1, i1 = -1, u0 = -1, v0 = -1;
        {
            XParams* q = kp;
            asm volatile("" : "+s"(q));
            i0 = idx_of(q, c, fr0);
            if (c + n_chunks < n_work) i1 = idx_of(q, c + n_chunks, fr1);
            pair_of(q, i0, u0, v0);
        }
        for (int w = c; w < n_work; w += n_chunks) {
            XParams* q = kp;
            asm volatile("" : "+s"(q));
            bool fr2 = false;
            int i2 = -1, u1 = -1, v1 = -1;
            if (w + 2 * n_chunks < n_work) i2 = idx_of(q, w + 2 * n_chunks, fr2);
            if (w + n_chunks < n_work) pair_of(q, i1, u1, v1);
            XHead H;
            H.u = u0; H.v = v0;
            bounds_of(q, H);
            extract_pair<BW, FAST>(*q, i0, fr0, H, xlds, slot);
            i0 = i1; fr0 = fr1; u0 = u1; v0 = v1;
            i1 = i2; fr1 = fr2;
        }
        if (!dyn) { c += gridDim.x; continue; }
        if (BW == 64) {
            c = (int)gridDim.x + TLC_X_COUNTERS * __builtin_amdgcn_readfirstlane(nxt) + ck;
            // this counter is dry: the next one (at most once round), looked at with a plain load first -- at the end of the kernel
            // every wavefront comes through here, and 4 096 x 7 atomics on dry counters took longer than the extraction itself
            while (c >= n_chunks && ++dry < TLC_X_COUNTERS) {
                ck = (ck + 1) % TLC_X_COUNTERS;
                int* cnt = p.work_counter + ck * TLC_X_COUNTER_STRIDE;
                int t = 0;
                if (threadIdx.x == 0) t = __atomic_load_n(cnt, __ATOMIC_RELAXED);
                if ((int)gridDim.x + TLC_X_COUNTERS * __builtin_amdgcn_readfirstlane(t) + ck >= n_chunks) continue;
                if (threadIdx.x == 0) t = atomicAdd(cnt, 1);
                c = (int)gridDim.x + TLC_X_COUNTERS * __builtin_amdgcn_readfirstlane(t) + ck;
            }
        } else {
            if (threadIdx.x == 0) s_chunk = (int)gridDim.x + atomicAdd(p.work_counter, 1);
            __syncthreads();
            c = __builtin_amdgcn_readfirstlane(s_chunk);
            __syncthreads();
        }
    }
}

template __global__ void tlc_extract_kernel<64, false>(TlcVicParams);
template __global__ void tlc_extract_kernel<64, true>(TlcVicParams);
template __global__ void tlc_extract_kernel<512, false>(TlcVicParams);

size_t tlc_extract_lds_bytes(int nw, int bw, bool fast) { return x_layout(nw, fast ? TLC_BE_CAP + 8 : (bw == 64 ? XCfg<64>::SID_CAP : XCfg<512>::SID_CAP), bw, fast).total; }

int tlc_launch_extract(int bw, int grid, size_t lds, const TlcVicParams& p, void* stream, bool fast) {
    if (grid <= 0) return TLC_OK;
    if (fast) {
        if (lds > 64 * 1024) TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_extract_kernel<64, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((tlc_extract_kernel<64, true>), dim3(grid), dim3(64), lds, (hipStream_t)stream, p);
    } else if (bw == 64) {
        if (lds > 64 * 1024) TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_extract_kernel<64, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((tlc_extract_kernel<64, false>), dim3(grid), dim3(64), lds, (hipStream_t)stream, p);
    } else {
        if (lds > 64 * 1024) TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_extract_kernel<512, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((tlc_extract_kernel<512, false>), dim3(grid), dim3(512), lds, (hipStream_t)stream, p);
    }
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// ---- which pairs go first ------------------------------------------------------------------------------------------------
// k = min(|ball(u)|, |ball(v)|) >= |S|.  k >= cand_threshold: candidate of the early pass (at most cand_cap are listed; the
// count keeps running so that the main pass knows whether the list is complete).  Else k >= 256 / 128 / 64: bins 0 / 1 / 2.
__global__ void tlc_classify_kernel(int n_pairs, const int* __restrict__ pairs, int n_nodes, const int* __restrict__ bptr,
                                    int cand_threshold, int cand_cap, int* __restrict__ cand_count, int* __restrict__ cand_list,
                                    int* __restrict__ big_count, int* __restrict__ big_list) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1;                                         // 3 = candidate, 0..2 = bin
    if (i < n_pairs) {
        const int u = pairs[2 * (size_t)i], v = pairs[2 * (size_t)i + 1];
        if (u >= 0 && v >= 0 && u < n_nodes && v < n_nodes) {
            int a0, a1, b0, b1;
            row_bounds(bptr, u, a0, a1);
            row_bounds(bptr, v, b0, b1);
            const int a = a1 - a0, b = b1 - b0;
            const int k = a < b ? a : b;
            if (cand_list && k >= cand_threshold) cls = 3;
            else if (k >= 256) cls = 0;
            else if (k >= 128) cls = 1;
            else if (k >= TLC_X_BIN_MIN) cls = 2;
        }
    }
    // one round of atomics per wavefront: lane c reserves the room of class c, then the bases go round
    unsigned long long mk[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) mk[c] = __ballot(cls == c);
    const int lane = tlc_lane();
    int base = 0;
    if (lane < 4) {
        const unsigned long long m = lane == 0 ? mk[0] : (lane == 1 ? mk[1] : (lane == 2 ? mk[2] : mk[3]));
        if (m) base = atomicAdd(lane == 3 ? cand_count : &big_count[lane], __popcll(m));
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int bc = __builtin_amdgcn_readlane(base, c);
        if (cls == c) {
            const int pos = bc + __popcll(mk[c] & tlc_lanemask_lt());
            if (c == 3) { if (pos < cand_cap) cand_list[pos] = i; }
            else big_list[(size_t)c * n_pairs + pos] = i;
        }
    }
}
int tlc_launch_classify(int n_pairs, const int* pairs, int n_nodes, const int* bptr, int cand_threshold, int cand_cap,
                        int* cand_count, int* cand_list, int* big_count, int* big_list, void* stream) {
    if (n_pairs <= 0) return TLC_OK;
    hipLaunchKernelGGL(tlc_classify_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_pairs, pairs, n_nodes,
                       bptr, cand_threshold, cand_cap, cand_count, cand_list, big_count, big_list);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// ---- the ball lists: ball_hop(x) for every node, ascending ids, x itself included ------------------------------------------------
// One wavefront per node (grid-stride): x, its row and (hop 2) the rows of its neighbours are marked in an LDS bitmap;
// COUNT stores the population, FILL the ids from bptr[x] on.  One-off per (graph, hop).
template <bool FILL>
__global__ __launch_bounds__(64) void tlc_ball_list_kernel(int n_nodes, int nw, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                           int hop, int* __restrict__ bsize, const int* __restrict__ bptr, int* __restrict__ bcol) {
    extern __shared__ __attribute__((aligned(16))) unsigned bbits[];
    const int lane = tlc_lane();
    const int wpl = (nw + 63) / 64;
    const int w0 = lane * wpl < nw ? lane * wpl : nw, w1 = (w0 + wpl) < nw ? (w0 + wpl) : nw;
    for (int w = w0; w < w1; ++w) bbits[w] = 0u;
    __syncthreads();
    // hop >= 3 (round 5; not a value the reference's configurations use): breadth-first levels over two more bitmaps behind the first
    unsigned* cur = bbits + nw;
    unsigned* nxt = cur + nw;
    if (hop >= 3) {
        for (int w = w0; w < w1; ++w) { cur[w] = 0u; nxt[w] = 0u; }
        __syncthreads();
    }
    for (int x = blockIdx.x; x < n_nodes; x += gridDim.x) {
        const int rb = rowptr[x], re = rowptr[x + 1];
        if (lane == 0) atomicOr(&bbits[x >> 5], 1u << (x & 31));
        if (hop <= 2) {
            for (int j = rb + lane; j < re; j += TLC_WAVE) {
                const int a = col[j];
                atomicOr(&bbits[a >> 5], 1u << (a & 31));
                if (hop >= 2) {
                    const int ab = rowptr[a], ae = rowptr[a + 1];
                    for (int t = ab; t < ae; ++t) { const int c = col[t]; atomicOr(&bbits[c >> 5], 1u << (c & 31)); }
                }
            }
        } else {
            if (lane == 0) cur[x >> 5] = 1u << (x & 31);
            __syncthreads();
            for (int h = 0; h < hop; ++h) {
                // every lane walks the frontier members in its words; a node enters the next frontier when its visited bit was clear
                for (int w = w0; w < w1; ++w) {
                    unsigned f = cur[w];
                    while (f) {
                        const int a = (w << 5) + __builtin_ctz(f);
                        f &= f - 1;
                        for (int t = rowptr[a], te = rowptr[a + 1]; t < te; ++t) {
                            const int c = col[t];
                            const unsigned bit = 1u << (c & 31);
                            if (!(atomicOr(&bbits[c >> 5], bit) & bit)) atomicOr(&nxt[c >> 5], bit);
                        }
                    }
                }
                __syncthreads();
                for (int w = w0; w < w1; ++w) { cur[w] = nxt[w]; nxt[w] = 0u; }
                __syncthreads();
            }
            for (int w = w0; w < w1; ++w) cur[w] = 0u;
        }
        __syncthreads();
        int cnt = 0;
        for (int w = w0; w < w1; ++w) cnt += __popc(bbits[w]);
        const int incl = tlc_wave_iscan_i32(cnt);
        if (!FILL) {
            if (lane == 63) bsize[x] = incl;
        } else {
            int o = bptr[x] + incl - cnt;
            for (int w = w0; w < w1; ++w) {
                unsigned s = bbits[w];
                while (s) { bcol[o++] = (w << 5) + __builtin_ctz(s); s &= s - 1; }
            }
        }
        for (int w = w0; w < w1; ++w) bbits[w] = 0u;
        __syncthreads();
    }
}

// ---- the ball subgraphs: for every node x with |ball(x)| <= TLC_BE_CAP, the directed entries a -> b of the graph with a and b in
// ball(x), as positions in the (ascending) ball list, sources ascending, a source's entries in CSR order.  One wavefront per node
// (grid-stride): the members are marked in an LDS bitmap with the rank of every word's first member beside it (as the extraction
// marks a vicinity), then member after member the wavefront streams the member's row, 64 entries per step, and keeps the entries
// whose column is a member.  COUNT stores the number of entries, FILL writes them from be_ptr[x] on.  One-off per (graph, hop).
template <bool FILL>
__global__ __launch_bounds__(64) void tlc_ball_edges_kernel(int n_nodes, int nw, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                            const double* __restrict__ w, const int* __restrict__ bptr,
                                                            const int* __restrict__ bcol, int* __restrict__ esize,
                                                            const int* __restrict__ be_ptr, unsigned* __restrict__ be_pos,
                                                            double* __restrict__ be_w) {
    extern __shared__ __attribute__((aligned(16))) unsigned ebits[];
    const int lane = tlc_lane();
    unsigned short* pref = (unsigned short*)(ebits + ((nw + 3) & ~3));
    __shared__ int eid[TLC_BE_CAP];
    for (int k = lane; k < nw; k += TLC_WAVE) ebits[k] = 0u;
    __syncthreads();
    for (int x = blockIdx.x; x < n_nodes; x += gridDim.x) {
        const int b0 = bptr[x], nb = bptr[x + 1] - b0;
        if (nb > TLC_BE_CAP) { if (!FILL && lane == 0) esize[x] = 0; continue; }        // (uniform)
        for (int k = lane; k < nb; k += TLC_WAVE) eid[k] = bcol[b0 + k];
        __syncthreads();
        for (int k = lane; k < nb; k += TLC_WAVE) {
            const int id = eid[k];
            atomicOr(&ebits[id >> 5], 1u << (id & 31));
            if (k == 0 || (eid[k - 1] >> 5) != (id >> 5)) pref[id >> 5] = (unsigned short)k;
        }
        __syncthreads();
        int run = FILL ? be_ptr[x] : 0;
        for (int k = 0; k < nb; ++k) {
            const int y = eid[k];
            const int rb = rowptr[y], re = rowptr[y + 1];
            for (int j = rb; j < re; j += TLC_WAVE) {
                const int jj = j + lane;
                const int c = jj < re ? col[jj] : -1;
                bool h = false;
                int pos = 0;
                if (c >= 0) {
                    const unsigned word = ebits[c >> 5], bit = 1u << (c & 31);
                    h = (word & bit) != 0u;
                    pos = (int)pref[c >> 5] + __popc(word & (bit - 1u));
                }
                const unsigned long long m = __ballot(h);
                if (FILL && h) {
                    const int o = run + __popcll(m & tlc_lanemask_lt());
                    be_pos[o] = ((unsigned)k << 16) | (unsigned)pos;
                    be_w[o] = w[jj];
                }
                run += __popcll(m);
            }
        }
        if (!FILL && lane == 0) esize[x] = run;
        __syncthreads();
        for (int k = lane; k < nb; k += TLC_WAVE) ebits[eid[k] >> 5] = 0u;
        __syncthreads();
    }
}

// ---- ball bitmaps (TlcVicParams::bbits): row x gets the bits of ball(x)'s sorted list; one wavefront per node, rows zeroed before ----
__global__ __launch_bounds__(64) void tlc_ball_bits_kernel(int n_nodes, int nw, const int* __restrict__ bptr, const int* __restrict__ bcol,
                                                           unsigned* __restrict__ bbits) {
    for (int x = blockIdx.x; x < n_nodes; x += gridDim.x) {
        const int b0 = bptr[x], b1 = bptr[x + 1];
        unsigned* row = bbits + (size_t)x * (size_t)nw;
        for (int j = b0 + (int)threadIdx.x; j < b1; j += 64) {
            const int y = bcol[j];
            atomicOr(&row[y >> 5], 1u << (y & 31));
        }
    }
}
int tlc_launch_ball_bits(int n_nodes, int nw, const int* bptr, const int* bcol, unsigned* bbits, void* stream) {
    if (n_nodes <= 0) return TLC_OK;
    hipLaunchKernelGGL(tlc_ball_bits_kernel, dim3(n_nodes < 16384 ? n_nodes : 16384), dim3(64), 0, (hipStream_t)stream, n_nodes, nw, bptr, bcol, bbits);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

int tlc_launch_ball_edges(bool fill, int n_nodes, int nw, const int* rowptr, const int* col, const double* w, const int* bptr, const int* bcol,
                          int* esize, const int* be_ptr, unsigned* be_pos, double* be_w, int grid, void* stream) {
    if (n_nodes <= 0) return TLC_OK;
    const size_t lds = (size_t)((nw + 3) & ~3) * 6 + 16;
    if (fill) {
        if (lds > 64 * 1024) TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_ball_edges_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((tlc_ball_edges_kernel<true>), dim3(grid), dim3(64), lds, (hipStream_t)stream, n_nodes, nw, rowptr, col, w, bptr, bcol, esize, be_ptr, be_pos, be_w);
    } else {
        if (lds > 64 * 1024) TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_ball_edges_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((tlc_ball_edges_kernel<false>), dim3(grid), dim3(64), lds, (hipStream_t)stream, n_nodes, nw, rowptr, col, w, bptr, bcol, esize, be_ptr, be_pos, be_w);
    }
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

int tlc_launch_ball_list(bool fill, int n_nodes, int nw, const int* rowptr, const int* col, int hop, int* bsize, const int* bptr,
                         int* bcol, int grid, void* stream) {
    if (n_nodes <= 0) return TLC_OK;
    const size_t lds = (size_t)(hop >= 3 ? 3 : 1) * nw * 4 + 16;
    if (lds > (size_t)160 * 1024) { tlc_set_error("ball lists: %zu B of LDS bitmaps", lds); return TLC_ERR_UNSUPPORTED; }
    if (fill) {
        if (lds > 64 * 1024) TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_ball_list_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((tlc_ball_list_kernel<true>), dim3(grid), dim3(64), lds, (hipStream_t)stream, n_nodes, nw, rowptr, col, hop, bsize, bptr, bcol);
    } else {
        if (lds > 64 * 1024) TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_ball_list_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((tlc_ball_list_kernel<false>), dim3(grid), dim3(64), lds, (hipStream_t)stream, n_nodes, nw, rowptr, col, hop, bsize, bptr, bcol);
    }
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// pd_tiny.hip -- P5..P9 for the smallest vicinities, ONE LANE per subgraph.
//
// 60 % of the pairs of a PubMed-shaped batch have vicinities of at most 16 nodes and 24 edges (median: 14 nodes).  A
// wavefront per subgraph (tlc_pd_tier_kernel, SMALL tier) runs every vector instruction of its sorts, passes and image stage
// with a quarter of its lanes in use, and that kernel is issue-bound: 2 250 vector instructions per subgraph.  Here a lane
// runs the whole chain for its own subgraph as plain serial code -- the reference's algorithm almost line for line -- with its
// arrays in LDS, interleaved so that lane l's element k sits at [k][l] (no bank conflicts while the lanes walk in step).  A
// lane spends ~10-25 k instructions, i.e. a few hundred wavefront instructions per subgraph, and 64 subgraphs share them.
//
//   P5  filtration.build_fv            sg2dgm/riccidist2dgm.py:20-61     one search per root + tight-chain walks; a search
//                                       SOURCED AT x where the chain is ambiguous (the reference's value is the minimum over
//                                       paths of the left-to-right fp64 sum that starts at x)
//   P6  perturb_filter_function        sg2dgm/accelerated_PD.py:6-23     keys evaluated in the reference's association
//   P7  Union_find                     sg2dgm/accelerated_PD.py:26-113   sorted edge lists, path halving, elder rule
//   P8  Accelerate_PD                  sg2dgm/accelerated_PD.py:115-178  root paths as bit masks, first maximum, eversion
//   P9  PersistenceImager.transform    sg2dgm/PersistenceImager.pyx:352-388  25 accumulators in registers
// Input: the fixed-size SMALL-tier slot the COUNT pass wrote (directed entries src<<16|dst + weight).  Plain TLC-GNN batch
// path only (flags == 0, res == 5): every variant flag goes through the wavefront kernels.  Compile with -ffp-contract=off.
#include "tlc_common.h"
#include "tlc_kernels.h"

namespace {

// lane-interleaved array in LDS: element k of this lane
template <typename T>
struct LaneArr {
    T* base;
    __device__ __forceinline__ T& operator[](int k) const { return base[k * 64]; }
};

// byte j of a lane inside the lane's own 8 bytes of every row of an f64 lane-interleaved array (row = 64 lanes x 8 bytes)
struct LaneBytes8 {
    unsigned char* base;       // the lane's first byte of row 0
    __device__ __forceinline__ unsigned char& operator[](int j) const { return base[(j >> 3) * 512 + (j & 7)]; }
};
struct LaneBytes8Slice {
    LaneBytes8 b;
    int off;
    __device__ __forceinline__ unsigned char& operator[](int j) const { return b[off + j]; }
};

__device__ __forceinline__ double tiny_key_asc(double fa, double fb) {
    const double hi = fa > fb ? fa : fb, lo = fa < fb ? fa : fb;
    return hi + (lo + 1.0) * 1e-6;
}
__device__ __forceinline__ double tiny_key_desc(double fa, double fb) {
    const double hi = fa > fb ? fa : fb, lo = fa < fb ? fa : fb;
    return lo - (101.0 - hi) * 1e-6;
}

// _norm_cdf on |x| <= 1.2 (all the image stage of a normalised filtration asks for): the 19-term Maclaurin series of erf used by
// the tier kernels (pd_pipeline.hip, tlc_norm_cdf<true>), same coefficients
__device__ __forceinline__ double tiny_norm_cdf(double x) {
    const double z = x * 0.70710678118654752440;
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

struct TinyImage {
    double px[25];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < 25; ++k) px[k] = 0.0;
    }
    // one diagram point (PersistenceImager.pyx:367-388, factored form)
    __device__ __forceinline__ void add(double b, double d) {
        const double pers = d - b;
        const double wgt = pers < 0.0 ? 0.0 : (pers > 1.0 ? 1.0 : pers);
        if (wgt == 0.0) return;
        const double step = ((1.0 + 0.2) - 0.0) / 6.0;                 // _create_mesh (:302-314), resolution 5
        double cb[6], cp[6];
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            cb[g] = tiny_norm_cdf((double)g * step - b);
            cp[g] = tiny_norm_cdf((double)g * step - pers);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const double db = cb[i + 1] - cb[i];
#pragma unroll
            for (int j = 0; j < 5; ++j) px[i * 5 + j] += wgt * (db * (cp[j + 1] - cp[j]));
        }
    }
};

constexpr int TN = TLC_T_NMAX, TM = TLC_T_MMAX;
constexpr int TP = TLC_T_NMAX + TLC_T_MMAX;          // diagram points with a weight: < n from the ascending pass, [min,max], <= m - n + 1 loops
// LDS per workgroup, regions reused as the stages go by (576 bytes per lane = 36 KB: four workgroups per CU; the first layout,
// one region per array, took 66 KB -- two per CU -- and its workgroups waited for the other tiers to drain, 0.44 ms for
// 0.15 ms of work):
//   A  weights -> sort keys                                    f64[TM]
//   B  distances from u -> filtration values f                 f64[TN]   (the chain walks read the tight-successor tables only)
//   C  distances from v -> per-source fallback search; then    f64[TN]
//      points (birth node, death node), tree parent / parent edge          u8[TP] x 2, u8[TN] x 2
//   D  edge endpoints                                          u16[TM]
//   E  tight-successor counts / entries, ambiguous sources; then sorted order, components, Pos (front) / Neg (back) list
constexpr size_t T_OA = 0;
constexpr size_t T_OB = T_OA + (size_t)TM * 64 * 8;
constexpr size_t T_OC = T_OB + (size_t)TN * 64 * 8;
constexpr size_t T_OD = T_OC + (size_t)TN * 64 * 8;
constexpr size_t T_OE = T_OD + (size_t)TM * 64 * 2;
constexpr size_t TINY_WG_BYTES = T_OE + 5 * (size_t)TN * 64;
static_assert(2 * TP + 2 * TN <= TN * 8, "points + tree tables alias the second distance array (8 bytes per lane and row)");
static_assert(2 * (size_t)TM * 64 + (size_t)TN * 64 <= 5 * (size_t)TN * 64, "order / components / Pos-Neg list alias the tight-successor tables");

}  // namespace

// The subgraphs come from the arena through a tier list (slot = 64 list positions).
// one slot of 64 subgraphs on one wavefront; `lds`: the wavefront's TINY_WG_BYTES
__device__ __forceinline__ void tiny_slot(const TlcPdParams p, unsigned char* lds, int lane, int slot) {
    // carve (see the region table above)
    LaneArr<double> ew{(double*)(lds + T_OA) + lane};
    const LaneArr<double>& key = ew;                  // (the sort keys take the weights' place once f is final)
    LaneArr<double> du{(double*)(lds + T_OB) + lane};
    const LaneArr<double>& f = du;                    // (f[x] is written while the walks read cnt / nxt / eab / ew only)
    LaneArr<double> dv{(double*)(lds + T_OC) + lane};
    const LaneArr<double>& dist = dv;                 // (per-source searches run after the tight-successor tables are built)
    // (byte arrays inside the lane's OWN eight bytes of each f64 row of region C: a byte array interleaved one byte per lane would
    // spread over the doubles of other lanes, and lanes are not always in the same stage -- the lanes of far pairs run their
    // one search after the others have been through every stage)
    LaneBytes8 cbytes{lds + T_OC + 8 * (size_t)lane};
    LaneBytes8Slice ptb{cbytes, 0};                                             // diagram points as (birth node, death node):
    LaneBytes8Slice ptd{cbytes, TP};                                            // every coordinate is a copy of some f[node]
    LaneBytes8Slice par{cbytes, 2 * TP};
    LaneBytes8Slice pedge{cbytes, 2 * TP + TN};
    LaneArr<unsigned short> eab{(unsigned short*)(lds + T_OD) + lane};
    LaneArr<unsigned char> cntU{lds + T_OE + lane};
    LaneArr<unsigned char> nxtU{lds + T_OE + (size_t)TN * 64 + lane};
    LaneArr<unsigned char> cntV{lds + T_OE + 2 * (size_t)TN * 64 + lane};
    LaneArr<unsigned char> nxtV{lds + T_OE + 3 * (size_t)TN * 64 + lane};
    LaneArr<unsigned char> ambl{lds + T_OE + 4 * (size_t)TN * 64 + lane};    // sources whose tight chain is ambiguous
    LaneArr<unsigned char> ord{lds + T_OE + lane};
    LaneArr<unsigned char> comp{lds + T_OE + (size_t)TM * 64 + lane};
    LaneArr<unsigned char> pn{lds + T_OE + (size_t)TM * 64 + (size_t)TN * 64 + lane};   // Pos edges from the front, Neg edges from the back
    int npts = 0;
    int i, n, m2 = 0, lu, lv, m = 0;
    const unsigned* adir = nullptr;
    const double* alw = nullptr;
    {
        if (p.tiny_bin_list) {
            // the list by size class: this wavefront's 64 entries come from ONE bin (slots are dealt to the bins in order, largest first)
            int s = slot, b = 0;
            while (b < TLC_TINY_BINS - 1 && s >= ((p.tiny_bin_cnt[b] + 63) >> 6)) { s -= (p.tiny_bin_cnt[b] + 63) >> 6; ++b; }
            const int wi = s * 64 + lane;
            if (wi >= p.tiny_bin_cnt[b]) return;
            i = p.tiny_bin_list[(size_t)b * p.tiny_bin_stride + wi];
        } else {
            int tier_count = p.tier_count;
            if (p.tier_count_dev) { const int c = *p.tier_count_dev; tier_count = c < tier_count ? c : tier_count; }
            const int wi = slot * 64 + lane;
            if (wi >= tier_count) return;
            i = p.tier_list[wi];
        }
        n = p.hdr_n[i]; m2 = p.hdr_m2[i]; lu = p.hdr_lu[i]; lv = p.hdr_lv[i];
        // (fixed slots written by the breadth-first COUNT pass, or wherever the extraction left the subgraph in the arena)
        adir = p.small_dir ? p.small_dir + (size_t)i * (2 * TLC_S_MMAX) : p.A_dir + p.edge_off[i];
        alw = p.small_dir ? p.small_lw + (size_t)i * (2 * TLC_S_MMAX) : p.A_lw + p.edge_off[i];
    }
    int status = TLC_ST_OK;
    TinyImage img;
    img.clear();
    // diagnostics (tools/tiny_profile.py): cycles of the wavefront per phase, summed over the wavefronts by their first lanes
    unsigned long long* pc = p.phase_cycles;
    unsigned long long t_prev = pc ? clock64() : 0ull;
#define TINY_STAMP(k) do { if (pc) { const unsigned long long _t = clock64(); if (lane == 0) atomicAdd(&pc[(k)], _t - t_prev); t_prev = _t; } } while (0)
    {
        // undirected edge list: the directed entries with src < dst, in CSR order.  Eight entries (and their weights, wanted or not)
        // are requested before the first is looked at: every lane reads its own subgraph, so a load is 64 scattered lines, and one
        // entry per round trip made this loop 22 % of the kernel (tools/tiny_profile.py: 105 k of 479 k cycles per wavefront)
        for (int j0 = 0; j0 < m2; j0 += 8) {
            unsigned ee[8];
            double wq[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const bool in = j0 + q < m2;
                ee[q] = in ? adir[j0 + q] : 0u;
                wq[q] = in ? alw[j0 + q] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const unsigned a = ee[q] >> 16, b = ee[q] & 0xffffu;
                if (j0 + q < m2 && a < b && m < TM) { eab[m] = (unsigned short)((a << 8) | b); ew[m] = wq[q]; ++m; }
            }
        }
    }
    TINY_STAMP(0);
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    // label-correcting shortest paths sourced at s: dist[y] = min over paths of the fp64 sum accumulated from s
    auto bellman_ford_from = [&](int s, const LaneArr<double>& d) {
        for (int y = 0; y < n; ++y) d[y] = INF;
        d[s] = 0.0;
        for (int round = 0; round <= n; ++round) {
            bool ch = false;
            for (int e = 0; e < m; ++e) {
                const unsigned ab = eab[e];
                const int a = (int)(ab >> 8), b = (int)(ab & 0xffu);
                const double w = ew[e];
                const double da = d[a], db = d[b];
                const double ca = da + w, cb = db + w;
                if (ca < db) { d[b] = ca; ch = true; }
                if (cb < da) { d[a] = cb; ch = true; }
            }
            if (!ch) break;
        }
    };
    const bool far = lu < 0;                          // d(u,v) > hop: every value is the double sentinel => f == 1
    if (far) {
        bellman_ford_from(0, dist);
        for (int y = 0; y < n; ++y) if (dist[y] == INF) status = TLC_ST_DISCONNECTED;           // assert (:318)
        if (status == TLC_ST_OK && n == 1) status = TLC_ST_NO_TREE_EDGE;                        // IndexError (accelerated_PD.py:122)
        // constant f: no strict pair, [1,1] has persistence 0 => the image is exactly zero
    } else {
        // ---- P5: 'sum' = dist_1 + dist_2 (:27-49).  The reference's distance from x is the minimum over paths of the fp64 sum
        // accumulated FROM x.  As in the wavefront kernels (pd_pipeline.hip): one search per root, the entries that are tight
        // within 1e-10 relative, and a walk from x along its tight chain summing the weights in the reference's order -- a
        // unique chain is the float-minimal path (every other path is longer by far more than any rounding); a node with
        // more than one tight successor on the way gets its own search sourced at x.
        bellman_ford_from(lu, du);
        for (int y = 0; y < n; ++y) if (du[y] == INF) status = TLC_ST_DISCONNECTED;
        if (status == TLC_ST_OK) {
            bellman_ford_from(lv, dv);
            double dmx = 0.0;
            for (int y = 0; y < n; ++y) { const double a = du[y], b = dv[y]; dmx = a > dmx ? a : dmx; dmx = b > dmx ? b : dmx; cntU[y] = 0; cntV[y] = 0; }
            const double tol = 1e-10 * (1.0 + dmx);
            for (int e = 0; e < m; ++e) {
                const unsigned ab = eab[e];
                const int a = (int)(ab >> 8), b = (int)(ab & 0xffu);
                const double w = ew[e];
                const double ua = du[a], ub = du[b], va = dv[a], vb = dv[b];
                if (a != lu && (w + ub) - ua <= tol) { cntU[a] = (unsigned char)(cntU[a] + 1); nxtU[a] = (unsigned char)e; }
                if (b != lu && (w + ua) - ub <= tol) { cntU[b] = (unsigned char)(cntU[b] + 1); nxtU[b] = (unsigned char)e; }
                if (a != lv && (w + vb) - va <= tol) { cntV[a] = (unsigned char)(cntV[a] + 1); nxtV[a] = (unsigned char)e; }
                if (b != lv && (w + va) - vb <= tol) { cntV[b] = (unsigned char)(cntV[b] + 1); nxtV[b] = (unsigned char)e; }
            }
            double mx = 0.0;
            int namb = 0;
            for (int x = 0; x < n; ++x) {
                double fr = 0.0;
                if (x != lu && x != lv) {
                    double d1 = 0.0, d2 = 0.0;
                    bool amb = false;
                    int a = x, steps = 0;
                    while (a != lu) {
                        if (cntU[a] != 1 || ++steps > n) { amb = true; break; }
                        const int e = nxtU[a];
                        const unsigned ab = eab[e];
                        d1 = d1 + ew[e];
                        a = ((int)(ab >> 8) == a) ? (int)(ab & 0xffu) : (int)(ab >> 8);
                    }
                    a = x; steps = 0;
                    while (!amb && a != lv) {
                        if (cntV[a] != 1 || ++steps > n) { amb = true; break; }
                        const int e = nxtV[a];
                        const unsigned ab = eab[e];
                        d2 = d2 + ew[e];
                        a = ((int)(ab >> 8) == a) ? (int)(ab & 0xffu) : (int)(ab >> 8);
                    }
                    if (amb) ambl[namb++] = (unsigned char)x;       // (resolved below: the lanes of a wavefront must not each drag
                    fr = d1 + d2;                                   //  the others through a search of their own inside this loop)
                }
                f[x] = fr;
            }
            for (int q = 0; q < namb; ++q) {
                const int x = ambl[q];
                bellman_ford_from(x, dist);
                f[x] = dist[lu] + dist[lv];
            }
            if (namb && p.stats) atomicAdd(&p.stats[0], (unsigned long long)namb);
            for (int x = 0; x < n; ++x) { const double v = f[x]; mx = v > mx ? v : mx; }
            TINY_STAMP(1);
            if (mx == 0.0) status = TLC_ST_ZERO_RANGE;                                          // ZeroDivisionError (:54)
            else for (int x = 0; x < n; ++x) f[x] = f[x] / mx;                                  // (:50-56)
        }
        if (status == TLC_ST_OK) {
            // ---- P6 + P7 ascending pass (:27-68): an edge's key exceeds both endpoints' values, so every node is made
            // before any of its edges: only the edges need sorting (stable insertion sort: equal keys keep list order)
            for (int e = 0; e < m; ++e) { const unsigned ab = eab[e]; key[e] = tiny_key_asc(f[ab >> 8], f[ab & 0xffu]); }
            for (int e = 0; e < m; ++e) {                  // stable: position = #smaller keys + #equal keys in front
                const double k = key[e];
                int r = 0;
                for (int e2 = 0; e2 < m; ++e2) { const double k2 = key[e2]; r += (k2 < k || (k2 == k && e2 < e)) ? 1 : 0; }
                ord[r] = (unsigned char)e;
            }
            auto find = [&](int x) {
                int c = comp[x];
                while (x != c) { const int g = comp[c]; comp[x] = (unsigned char)g; x = g; c = comp[x]; }   // path halving (:53-58)
                return x;
            };
            for (int y = 0; y < n; ++y) comp[y] = (unsigned char)y;
            int imin = 0, imax = 0;
            {
                double fmin = f[0], fmax = f[0];
                for (int y = 1; y < n; ++y) { const double v = f[y]; if (v < fmin) { fmin = v; imin = y; } if (v > fmax) { fmax = v; imax = y; } }
            }
            for (int t = 0; t < m; ++t) {
                const unsigned ab = eab[ord[t]];
                const int a = (int)(ab >> 8), b = (int)(ab & 0xffu);
                const int pu = find(a), pv = find(b);
                if (pu != pv) {
                    const int small = (f[pu] <= f[pv]) ? pu : pv, large = pu + pv - small;     // (:63-64)
                    const int max_node = (f[a] > f[b]) ? a : b;                                  // (:65)
                    if (f[large] < f[max_node]) { ptb[npts] = (unsigned char)large; ptd[npts] = (unsigned char)max_node; ++npts; }   // (:66-67)
                    comp[large] = (unsigned char)small;
                }
            }
            TINY_STAMP(2);
            ptb[npts] = (unsigned char)imin; ptd[npts] = (unsigned char)imax; ++npts;            // [min, max] (:110); [max, min] weighs 0
            // ---- descending pass (:70-109): value descending, stable; Rel1 points weigh 0 in the image ------------------------
            for (int e = 0; e < m; ++e) { const unsigned ab = eab[e]; key[e] = tiny_key_desc(f[ab >> 8], f[ab & 0xffu]); }
            for (int e = 0; e < m; ++e) {                  // descending, stable
                const double k = key[e];
                int r = 0;
                for (int e2 = 0; e2 < m; ++e2) { const double k2 = key[e2]; r += (k2 > k || (k2 == k && e2 < e)) ? 1 : 0; }
                ord[r] = (unsigned char)e;
            }
            for (int y = 0; y < n; ++y) comp[y] = (unsigned char)y;
            int npos = 0, nneg = 0;
            for (int t = 0; t < m; ++t) {
                const int e = ord[t];
                const unsigned ab = eab[e];
                const int pu = find((int)(ab >> 8)), pv = find((int)(ab & 0xffu));
                if (pu != pv) {
                    pn[TM - 1 - nneg++] = (unsigned char)e;
                    const int small = (f[pu] <= f[pv]) ? pu : pv, large = pu + pv - small;     // (:101-102)
                    comp[small] = (unsigned char)large;
                } else {
                    pn[npos++] = (unsigned char)e;
                }
            }
            TINY_STAMP(3);
            if (nneg == 0) status = TLC_ST_NO_TREE_EDGE;                                        // list(Nodes)[0] (:122)
            else if (npos > 0) {
                // ---- P8 (:115-178): spanning tree of the Neg edges, rooted at the first endpoint of the first one -------------
                for (int y = 0; y < n; ++y) par[y] = 0xFFu;
                const int root = (int)(eab[pn[TM - 1]] >> 8);
                par[root] = (unsigned char)root;
                for (int round = 0; round < n; ++round) {
                    bool ch = false;
                    for (int t = 0; t < nneg; ++t) {
                        const int e = pn[TM - 1 - t];
                        const unsigned ab = eab[e];
                        const int a = (int)(ab >> 8), b = (int)(ab & 0xffu);
                        const unsigned pa = par[a], pb = par[b];
                        if (pa != 0xFFu && pb == 0xFFu) { par[b] = (unsigned char)a; pedge[b] = (unsigned char)e; ch = true; }
                        else if (pb != 0xFFu && pa == 0xFFu) { par[a] = (unsigned char)b; pedge[a] = (unsigned char)e; ch = true; }
                    }
                    if (!ch) break;
                }
                for (int e = 0; e < m; ++e) { const unsigned ab = eab[e]; key[e] = tiny_key_asc(f[ab >> 8], f[ab & 0xffu]); }
                for (int t = 0; t < npos; ++t) {
                    const int e = pn[t];
                    const unsigned ab = eab[e];
                    const int pnode = (int)(ab >> 8), q = (int)(ab & 0xffu);
                    if (par[pnode] == 0xFFu || par[q] == 0xFFu) continue;      // (other component: cannot happen, the vicinity is connected)
                    unsigned mask = 0u;                                         // path_0 as a bit mask (at most 16 nodes)
                    for (int a = pnode; a != root; a = par[a]) mask |= 1u << a;
                    int meet = root;
                    for (int a = q; a != root; a = par[a]) if ((mask >> a) & 1u) { meet = a; break; }
                    int best = -1, side = 0;
                    double bv = 0.0;
                    for (int a = pnode; a != meet; a = par[a]) { const double v = key[pedge[a]]; if (best < 0 || v > bv) { best = a; bv = v; side = 0; } }
                    for (int a = q; a != meet; a = par[a]) { const double v = key[pedge[a]]; if (best < 0 || v > bv) { best = a; bv = v; side = 1; } }
                    if (best < 0) continue;
                    const unsigned lab = eab[pedge[best]];
                    const double fa = f[lab >> 8], fb = f[lab & 0xffu];
                    const double large_value = fa > fb ? fa : fb;                                // (:160)
                    const double fp = f[pnode], fq = f[q];
                    const double low_value = fp < fq ? fp : fq;                                  // (:162)
                    if (large_value > low_value && npts < TP) {                                  // (:164)
                        ptb[npts] = (unsigned char)(fp < fq ? pnode : q);
                        ptd[npts] = (unsigned char)(fa > fb ? (lab >> 8) : (lab & 0xffu));
                        ++npts;
                    }
                    int node = side == 0 ? pnode : q, nodec = side == 0 ? q : pnode, e_in = e;   // evert (:168-176)
                    while (nodec != best) {
                        const int tp = par[node], te = pedge[node];
                        par[node] = (unsigned char)nodec; pedge[node] = (unsigned char)e_in;
                        nodec = node; node = tp; e_in = te;
                    }
                }
            }
        }
    }
    TINY_STAMP(4);
    // ---- P9: transform(np.array(PD_zero + PD_one)) (riccidist2dgm.py:327-328), one tight loop over the points ---------------
    if (status == TLC_ST_OK)
        for (int t = 0; t < npts; ++t) img.add(f[ptb[t]], f[ptd[t]]);
    TINY_STAMP(5);
    double* out = p.out_pi + (size_t)i * 25;
    if (status != TLC_ST_OK) img.clear();
#pragma unroll
    for (int k = 0; k < 25; ++k) out[k] = img.px[k];
    if (p.out_status) p.out_status[i] = (unsigned char)status;
    TINY_STAMP(6);
    if (pc && lane == 0) atomicAdd(&pc[14], 1ull);
#undef TINY_STAMP
}

// from the arena through a tier list: one workgroup = one wavefront = 64 list positions, no loop around the body
__global__ __launch_bounds__(64) void tlc_pd_tiny_kernel(TlcPdParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_all[];
#ifndef TLC_PRIO_TINY
#define TLC_PRIO_TINY 0
#endif
    if (TLC_PRIO_TINY > 0) __builtin_amdgcn_s_setprio(TLC_PRIO_TINY);      // (development A/B: 36 KB of LDS on one wavefront)
    tiny_slot(p, lds_all, (int)(threadIdx.x & 63), (int)blockIdx.x);
}

int tlc_launch_pd_tiny(const TlcPdParams& p, void* stream) {
    if (p.tier_count <= 0) return TLC_OK;
    const size_t lds = TINY_WG_BYTES;
    int grid = (p.tier_count + 63) / 64;
    if (p.tiny_bin_list) {                    // one wavefront per 64 entries of a size bin
        grid = 0;
        for (int b = 0; b < TLC_TINY_BINS; ++b) grid += (p.tiny_bin_cnt[b] + 63) / 64;
        if (grid == 0) return TLC_OK;
    }
    // (above the 64 KiB default of dynamic LDS; the attribute is per device, so it is set per launch: ~1 us)
    if (lds > 64 * 1024)
        TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_pd_tiny_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(tlc_pd_tiny_kernel, dim3(grid), dim3(64), lds, (hipStream_t)stream, p);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// A tier list by descending size: counting sort by (n + m) >> shift in one workgroup (histogram, prefix, scatter; which of two
// equal-sized vicinities comes first is left to the atomics -- every vicinity's row is its own).  Development option tier_sort of
// the SMALL / MID / MEDIUM lists (no gain there); the TINY list gets its size classes from the scan itself (TLC_TINY_BINS:
// this kernel took 110 us in front of the lane-per-subgraph kernel, one workgroup fighting over twenty LDS counters).
__global__ __launch_bounds__(1024) void tlc_tiny_sort_kernel(int count, const int* __restrict__ list, const int* __restrict__ hdr_n,
                                                            const int* __restrict__ hdr_m2, int* __restrict__ out, int shift) {
    __shared__ int hist[64], base[64];
    const int tid = (int)threadIdx.x;
    if (tid < 64) hist[tid] = 0;
    __syncthreads();
    for (int k = tid; k < count; k += 1024) {
        const int i = list[k];
        int key = (hdr_n[i] + (hdr_m2[i] >> 1)) >> shift;
        key = key < 0 ? 0 : (key > 63 ? 63 : key);
        atomicAdd(&hist[63 - key], 1);                         // (largest first)
    }
    __syncthreads();
    if (tid == 0) { int run = 0; for (int b = 0; b < 64; ++b) { base[b] = run; run += hist[b]; } }
    __syncthreads();
    for (int k = tid; k < count; k += 1024) {
        const int i = list[k];
        int key = (hdr_n[i] + (hdr_m2[i] >> 1)) >> shift;
        key = key < 0 ? 0 : (key > 63 ? 63 : key);
        out[atomicAdd(&base[63 - key], 1)] = i;
    }
}

int tlc_launch_tiny_sort(int count, const int* list, const int* hdr_n, const int* hdr_m2, int* out, void* stream, int shift) {
    if (count <= 0) return TLC_OK;
    hipLaunchKernelGGL(tlc_tiny_sort_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, count, list, hdr_n, hdr_m2, out, shift);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

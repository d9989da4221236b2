// w2_match.hip -- the diagram loss of PDGNN training: partial-matching Wasserstein distance between a predicted and a target
// persistence diagram.
//
// Replaces Knowledge_Distillation/wasserstein.py:198-379 (`wasserstein_distance(X, Y, order=p, enable_autodiff=True,
// num_models=1)`, called from Teacher_model.py:107-139 `compute_PD_loss(kernel='wasserstein')`): the reference builds the
// cost matrix  C[i, j] = ||X_i - Y_j||_inf ^ p,  C[i, m] = ((X_i.y - X_i.x) / 2) ^ p  (:45-67), gives every predicted point
// mass 1, every target point mass 1 and the diagonal mass n - m (:262-264), solves the transport with POT's `ot.emd`
// (third-party, absent here), and sums the matched distances:  loss = (sum_k |d_k|^p)^(1/p)  over the X-Y pairs and the
// points sent to the diagonal (:303-372; with num_models = 1 every diagonal point is kept, :329-337).
//
// With unit masses the transport is an ASSIGNMENT: n rows (predicted points) onto m target columns plus n - m copies of the
// diagonal.  One wavefront per diagram pair solves it with the shortest-augmenting-path (Hungarian) method, lanes = columns:
// the cost of (row, column) is recomputed from the two points (no matrix is stored), column duals / slack in registers, row
// duals, the column -> row map and the alternating-path links in LDS; O(n^2) wavefront steps of ~60 instructions (n <= 512;
// 513 .. 4 096 points: the same with a workgroup per problem, tlc_w2_match_wide_kernel).
// Also returned: which target every predicted point went to, the two partial sums the reference logs (wxy, wxd) and the
// gradient of the loss with respect to the predicted points (what `loss.backward()` would put on PD-hat).
//
// Evaluation (INFER = true): `wasserstein_distance_inference` (wasserstein.py:93-195, called with pair_diagonal=True from
// train_Teacher_Model.py:99 -> Teacher_model.py:66,134-136) is the classic transport in which BOTH diagrams may use the
// diagonal: masses a = [1]*n + [m], b = [1]*m + [n], cost matrix (n+1) x (m+1) with C[n, j] = ((Y_j.y - Y_j.x) / 2) ^ p and
// C[n, m] = 0 (:45-67,127-131).  With integer masses that is the assignment of n + m rows (the predicted points, then m copies
// of the diagonal) onto n + m columns (the targets, then n copies of the diagonal): same kernels, matrix dimension N = n + m,
// a third partial sum (wyd, the targets sent to the diagonal, :170-176) and the target -> predicted map beside the other one.
// Empty diagrams: the total persistence of the other one and zero partial sums (:98-113).
//
// PARITY UNPINNED: `ot.emd` is not available; the optimal COST is unique and is checked against
// scipy.optimize.linear_sum_assignment (tests/test_gpu_train.py); among several optimal assignments
// (ties) `ot.emd` may pick another one than this kernel.
#include "tlc_common.h"

namespace {

struct W2Params {
    int n_pairs;
    const long long* xoff;     // [B+1] rows of X per problem
    const double* X;           // [sum n, 2] predicted (birth, death)
    const long long* yoff;     // [B+1]
    const double* Y;           // [sum m, 2] target
    int order;                 // p: 1 or 2
    double* loss;              // [B]
    double* wxy;               // [B]
    double* wxd;               // [B]
    int* assign;               // [sum n]: target index (problem-local), -1 = diagonal
    double* gradX;             // [sum n, 2] or null
    unsigned char* status;     // [B]: 0 ok, 1 fewer predicted than target points, 2 too many points, 3 non-finite coordinates
    int leave_big;             // problems beyond this kernel's capacity are left alone (the workgroup kernel takes them)
    // INFER only
    double* wyd;               // [B]
    int* assign_y;             // [sum m]: predicted index (problem-local) a target is matched to, -1 = diagonal
};

__device__ __forceinline__ double w2_pow(double d, int order) { return order == 2 ? d * d : d; }

// A problem that cannot be solved: its outputs are defined (zero loss or NaN, no matching, zero gradient)
template <bool INFER, int STRIDE>
__device__ __forceinline__ void w2_fail(const W2Params& p, int b, long long x0, long long y0, int n, int m, int code, double lossv, int tid) {
    if (tid == 0) {
        p.status[b] = (unsigned char)code; p.loss[b] = lossv; p.wxy[b] = 0.0; p.wxd[b] = 0.0;
        if (INFER) p.wyd[b] = 0.0;
    }
    for (int i = tid; i < n; i += STRIDE) { p.assign[x0 + i] = -1; if (p.gradX) { p.gradX[2 * (x0 + i)] = 0.0; p.gradX[2 * (x0 + i) + 1] = 0.0; } }
    if (INFER) for (int j = tid; j < m; j += STRIDE) p.assign_y[y0 + j] = -1;
}

// One diagram against the empty diagram (wasserstein.py:98-113): the total persistence of the other one, (sum |d/2|^p)^(1/p);
// the three logged parts are returned as 0 there.  side = 0: the points are X (gradient!), 1: Y.
template <int STRIDE>
__device__ __forceinline__ double w2_perstot_partial(const double* P, long long o, int cnt, int order, int tid) {
    double s = 0.0;
    for (int i = tid; i < cnt; i += STRIDE) {
        const double d = fabs((P[2 * (o + i) + 1] - P[2 * (o + i)]) * 0.5);
        s += order == 2 ? d * d : d;
    }
    return s;
}

// INFER = false: n rows (predicted) onto m targets + (n - m) diagonal copies.  INFER = true: N = n + m rows (predicted, then m
// diagonal rows) onto N columns (targets, then n diagonal columns).  cost(i, j):
//     i < n, j < m: ||X_i - Y_j||_inf ^ p      i < n, j >= m: cxd_i      i >= n, j < m: cdy_j      i >= n, j >= m: 0
template <int CPL, bool INFER>
__global__ __launch_bounds__(64) void tlc_w2_match_kernel(W2Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char w2_lds[];
    constexpr int NMAX = 64 * CPL;
    double* xs = (double*)w2_lds;              // [NMAX] predicted births
    double* ys = xs + NMAX;                    // [NMAX] predicted deaths
    double* cxd = ys + NMAX;                   // [NMAX] cost of the diagonal for row i
    double* u = cxd + NMAX;                    // [NMAX] row duals
    int* pcol = (int*)(u + NMAX);              // [NMAX] row assigned to column j, -1 = free
    int* way = pcol + NMAX;                    // [NMAX] previous column on the alternating path, -1 = the start
    int* rdone = way + NMAX;                   // [NMAX] row already assigned by the column reduction
    const int lane = tlc_lane();
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    const double QNAN = __longlong_as_double(0x7FF8000000000000ll);
    auto fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    for (int b = blockIdx.x; b < p.n_pairs; b += gridDim.x) {
        const long long x0 = p.xoff[b], y0 = p.yoff[b];
        const int n = (int)(p.xoff[b + 1] - x0), m = (int)(p.yoff[b + 1] - y0);
        const int N = INFER ? n + m : n;       // dimension of the assignment
        if (N > NMAX && p.leave_big) continue;
        if ((!INFER && n < m) || N > NMAX) {
            // (n < m: the diagonal would need negative mass, wasserstein.py:264 -- the reference's transport has no solution)
            w2_fail<INFER, 64>(p, b, x0, y0, n, m, (!INFER && n < m) ? 1 : 2, 0.0, lane);
            continue;
        }
        if (INFER && (n == 0 || m == 0)) {
            // empty diagram(s) (:98-113): distance to the empty diagram, the parts are reported as 0
            double s = n ? w2_perstot_partial<64>(p.X, x0, n, p.order, lane) : w2_perstot_partial<64>(p.Y, y0, m, p.order, lane);
            for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
            const double L = p.order == 2 ? sqrt(s) : s;
            w2_fail<INFER, 64>(p, b, x0, y0, n, m, 0, L, lane);
            if (p.gradX)
                for (int i = lane; i < n; i += 64) {
                    const double sd = (p.X[2 * (x0 + i) + 1] - p.X[2 * (x0 + i)]) * 0.5;
                    const double w = p.order == 2 ? (L > 0.0 ? sd / L : 0.0) : (sd > 0.0 ? 1.0 : (sd < 0.0 ? -1.0 : 0.0));
                    p.gradX[2 * (x0 + i)] = -0.5 * w; p.gradX[2 * (x0 + i) + 1] = 0.5 * w;
                }
            continue;
        }
        bool finite = true;
        for (int i = lane; i < N; i += 64) {
            double bx = 0.0, by = 0.0;
            if (i < n) { bx = p.X[2 * (x0 + i)]; by = p.X[2 * (x0 + i) + 1]; }
            finite = finite && (bx - bx == 0.0) && (by - by == 0.0);
            xs[i] = bx; ys[i] = by;
            cxd[i] = w2_pow((by - bx) * 0.5, p.order);        // _dist_to_diag, internal_p = inf (:30-42)
            u[i] = 0.0; pcol[i] = -1;
        }
        double yx[CPL], yy[CPL], cdy[CPL], v[CPL], minv[CPL];
        bool used[CPL];
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = lane + 64 * c;
            yx[c] = yy[c] = 0.0;
            if (j < m) { yx[c] = p.Y[2 * (y0 + j)]; yy[c] = p.Y[2 * (y0 + j) + 1]; }
            finite = finite && (yx[c] - yx[c] == 0.0) && (yy[c] - yy[c] == 0.0);
            cdy[c] = w2_pow((yy[c] - yx[c]) * 0.5, p.order);
            v[c] = 0.0;
        }
        if (__ballot(!finite)) {
            // NaN / Inf coordinates (a diverging training step): every reduced cost would be NaN, no column would ever be the
            // minimum and the augmenting loop would not end
            w2_fail<INFER, 64>(p, b, x0, y0, n, m, 3, QNAN, lane);
            continue;
        }
        fence();
        // the cost of (row, column) from the two points; row values in scalars
        auto cost_of = [&](bool row_real, double xi, double yi, double cd, int j, int c) -> double {
            if (j < m) {
                if (!INFER || row_real) {
                    const double dx = fabs(xi - yx[c]), dy = fabs(yi - yy[c]);
                    return w2_pow(dx > dy ? dx : dy, p.order);             // chebyshev ^ order (:59-60)
                }
                return cdy[c];
            }
            return (!INFER || row_real) ? cd : 0.0;
        };
        // ---- column reduction (the usual start of the shortest-augmenting-path method): v_j = min_i c_ij, and a row that is the
        // minimiser of some column takes the lowest such column -- duals feasible (u = 0), every such pair tight; on random
        // diagrams this assigns ~60 % of the rows before the first augmentation
        {
            int rj[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) { v[c] = INF; rj[c] = 0; }
            for (int i = 0; i < N; ++i) {
                const double xi = xs[i], yi = ys[i], cd = cxd[i];
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const double cost = cost_of(i < n, xi, yi, cd, lane + 64 * c, c);
                    if (cost < v[c]) { v[c] = cost; rj[c] = i; }
                }
            }
            for (int i = lane; i < N; i += 64) way[i] = 0x7fffffff;
            fence();
#pragma unroll
            for (int c = 0; c < CPL; ++c) { const int j = lane + 64 * c; if (j < N) atomicMin(&way[rj[c]], j); else v[c] = 0.0; }
            fence();
#pragma unroll
            for (int c = 0; c < CPL; ++c) { const int j = lane + 64 * c; if (j < N && way[rj[c]] == j) pcol[j] = rj[c]; }
            for (int i = lane; i < N; i += 64) rdone[i] = way[i] != 0x7fffffff;
            fence();
        }
        bool failed = false;
        for (int i = 0; i < N && !failed; ++i) {
            if (rdone[i]) continue;                           // (uniform: LDS)
#pragma unroll
            for (int c = 0; c < CPL; ++c) { minv[c] = INF; used[c] = false; }
            int i0 = i, j0 = -1;
            for (;;) {
                const double xi = xs[i0], yi = ys[i0], ui = u[i0], cd = cxd[i0];
                double best = INF;
                int bj = -1;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const int j = lane + 64 * c;
                    if (j < N && !used[c]) {
                        const double cur = cost_of(i0 < n, xi, yi, cd, j, c) - ui - v[c];
                        if (cur < minv[c]) { minv[c] = cur; way[j] = j0; }
                        if (minv[c] < best) { best = minv[c]; bj = j; }
                    }
                }
                const double delta = tlc_wave_min_f64(best);
                const unsigned long long who = __ballot(best == delta && bj >= 0);
                if (who == 0ull) { failed = true; break; }    // (cannot happen with finite inputs; never index with lane 64)
                const int j1 = __builtin_amdgcn_readlane(bj, __builtin_ctzll(who));
                // dual update: rows of the used columns and the row that started the path go up, used columns go down,
                // the slack of the others shrinks
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                    const int j = lane + 64 * c;
                    if (j < N) {
                        if (used[c]) { u[pcol[j]] += delta; v[c] -= delta; }
                        else minv[c] -= delta;
                    }
                }
                if (lane == 0) u[i] += delta;
                j0 = j1;
#pragma unroll
                for (int c = 0; c < CPL; ++c) if (lane + 64 * c == j1) used[c] = true;
                fence();
                i0 = pcol[j1];
                if (i0 < 0) break;
            }
            if (failed) break;
            // augment along the alternating path (one lane: at most N links)
            if (lane == 0) {
                int j = j0;
                while (j >= 0) {
                    const int jp = way[j];
                    pcol[j] = jp < 0 ? i : pcol[jp];
                    j = jp;
                }
            }
            fence();
        }
        if (failed) {
            w2_fail<INFER, 64>(p, b, x0, y0, n, m, 3, QNAN, lane);
            fence();
            continue;
        }
        // ---- the loss and its pieces (:303-372 / :140-195): matched distances d_k, loss = (sum |d_k|^p)^(1/p) ----------------
        double sxy = 0.0, sxd = 0.0, syd = 0.0;
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = lane + 64 * c;
            if (j < N) {
                const int i = pcol[j];
                if (i < n) {
                    if (j < m) {
                        const double dx = fabs(xs[i] - yx[c]), dy = fabs(ys[i] - yy[c]);
                        const double d = dx > dy ? dx : dy;
                        sxy += p.order == 2 ? d * d : d;
                        p.assign[x0 + i] = j;
                        if (INFER) p.assign_y[y0 + j] = i;
                    } else {
                        const double d = fabs((ys[i] - xs[i]) * 0.5);
                        sxd += p.order == 2 ? d * d : d;
                        p.assign[x0 + i] = -1;
                    }
                } else if (j < m) {                            // (INFER) a target sent to the diagonal
                    const double d = fabs((yy[c] - yx[c]) * 0.5);
                    syd += p.order == 2 ? d * d : d;
                    p.assign_y[y0 + j] = -1;
                }
            }
        }
        for (int o = 32; o; o >>= 1) { sxy += __shfl_xor(sxy, o); sxd += __shfl_xor(sxd, o); syd += __shfl_xor(syd, o); }
        const double tot = sxy + sxd + syd;
        const double L = p.order == 2 ? sqrt(tot) : tot;
        if (lane == 0) {
            p.status[b] = 0;
            p.loss[b] = L;
            p.wxy[b] = p.order == 2 ? sqrt(sxy) : sxy;
            p.wxd[b] = p.order == 2 ? sqrt(sxd) : sxd;
            if (INFER) p.wyd[b] = p.order == 2 ? sqrt(syd) : syd;
        }
        // ---- d loss / d X: through the matched distances only (the matching is piecewise constant) -------------------------
        if (p.gradX) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int j = lane + 64 * c;
                if (j < N) {
                    const int i = pcol[j];
                    if (i >= n) continue;
                    double gx = 0.0, gy = 0.0;
                    if (j < m) {
                        const double ex = yx[c] - xs[i], ey = yy[c] - ys[i];          // Y - X (:311)
                        const double ax = fabs(ex), ay = fabs(ey);
                        const double d = ax > ay ? ax : ay;
                        const double w = p.order == 2 ? (L > 0.0 ? d / L : 0.0) : 1.0;   // dL/dd
                        if (ax >= ay) gx = -w * (ex > 0.0 ? 1.0 : (ex < 0.0 ? -1.0 : 0.0));
                        else gy = -w * (ey > 0.0 ? 1.0 : (ey < 0.0 ? -1.0 : 0.0));
                    } else {
                        const double s = (ys[i] - xs[i]) * 0.5;                          // signed distance to the diagonal
                        const double w = p.order == 2 ? (L > 0.0 ? s / L : 0.0) : (s > 0.0 ? 1.0 : (s < 0.0 ? -1.0 : 0.0));
                        gx = -0.5 * w; gy = 0.5 * w;
                    }
                    p.gradX[2 * (x0 + i)] = gx;
                    p.gradX[2 * (x0 + i) + 1] = gy;
                }
            }
        }
        fence();
    }
}

// ---- the same method for up to 4 096 rows: one WORKGROUP per problem -------------------------------------------------------
// 512 threads x 8 columns; a step's minimum is found per wavefront (DPP), then over the eight wavefronts through LDS: two barriers
// per step instead of none, ~1.5 us per step -- a 2 000-point problem takes a few hundred thousand steps at worst.  Everything
// else (costs recomputed from the points, duals, links, the loss and its gradient) as above, so that a problem gives the same
// numbers whichever kernel takes it.
constexpr int W2_WIDE_THREADS = 512, W2_WIDE_CPL = 8, W2_WIDE_NMAX = W2_WIDE_THREADS * W2_WIDE_CPL;
template <bool INFER>
__global__ __launch_bounds__(W2_WIDE_THREADS) void tlc_w2_match_wide_kernel(W2Params p, int min_points) {
    extern __shared__ __attribute__((aligned(16))) unsigned char w2_lds[];
    constexpr int NMAX = W2_WIDE_NMAX, CPL = W2_WIDE_CPL, W = W2_WIDE_THREADS, NWV = W / 64;
    double* xs = (double*)w2_lds;              // [NMAX] predicted births
    double* ys = xs + NMAX;                    // [NMAX] predicted deaths
    double* u = ys + NMAX;                     // [NMAX] row duals
    int* pcol = (int*)(u + NMAX);              // [NMAX] row assigned to column j, -1 = free
    int* way = pcol + NMAX;                    // [NMAX] previous column on the alternating path, -1 = the start
    double* red = (double*)(way + NMAX);       // [3 NWV] wavefront minima (at the end: the partial sums per wavefront)
    double* sums = red + 3 * NWV;              // [3]
    int* redj = (int*)(sums + 3);              // [NWV] the columns of the minima; [NWV]: non-finite input flag
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double INF = __longlong_as_double(0x7FF0000000000000ll);
    const double QNAN = __longlong_as_double(0x7FF8000000000000ll);
    const int b = blockIdx.x;
    if (b >= p.n_pairs) return;
    const long long x0 = p.xoff[b], y0 = p.yoff[b];
    const int n = (int)(p.xoff[b + 1] - x0), m = (int)(p.yoff[b + 1] - y0);
    const int N = INFER ? n + m : n;
    if (N <= min_points) return;                                      // (the one-wavefront kernel took it)
    if ((!INFER && n < m) || N > NMAX) {
        w2_fail<INFER, W>(p, b, x0, y0, n, m, (!INFER && n < m) ? 1 : 2, 0.0, tid);
        return;
    }
    if (INFER && (n == 0 || m == 0)) {
        double s = n ? w2_perstot_partial<W>(p.X, x0, n, p.order, tid) : w2_perstot_partial<W>(p.Y, y0, m, p.order, tid);
        for (int o = 32; o; o >>= 1) s += __shfl_xor(s, o);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        double t = 0.0;
        for (int k = 0; k < NWV; ++k) t += red[k];
        const double L = p.order == 2 ? sqrt(t) : t;
        w2_fail<INFER, W>(p, b, x0, y0, n, m, 0, L, tid);
        if (p.gradX)
            for (int i = tid; i < n; i += W) {
                const double sd = (p.X[2 * (x0 + i) + 1] - p.X[2 * (x0 + i)]) * 0.5;
                const double w = p.order == 2 ? (L > 0.0 ? sd / L : 0.0) : (sd > 0.0 ? 1.0 : (sd < 0.0 ? -1.0 : 0.0));
                p.gradX[2 * (x0 + i)] = -0.5 * w; p.gradX[2 * (x0 + i) + 1] = 0.5 * w;
            }
        return;
    }
    bool finite = true;
    if (tid == 0) redj[NWV] = 0;
    for (int i = tid; i < N; i += W) {
        double bx = 0.0, by = 0.0;
        if (i < n) { bx = p.X[2 * (x0 + i)]; by = p.X[2 * (x0 + i) + 1]; }
        finite = finite && (bx - bx == 0.0) && (by - by == 0.0);
        xs[i] = bx; ys[i] = by;
        u[i] = 0.0; pcol[i] = -1;
    }
    double yx[CPL], yy[CPL], cdy[CPL], v[CPL], minv[CPL];
    bool used[CPL];
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int j = tid + W * c;
        yx[c] = yy[c] = 0.0;
        if (j < m) { yx[c] = p.Y[2 * (y0 + j)]; yy[c] = p.Y[2 * (y0 + j) + 1]; }
        finite = finite && (yx[c] - yx[c] == 0.0) && (yy[c] - yy[c] == 0.0);
        cdy[c] = w2_pow((yy[c] - yx[c]) * 0.5, p.order);
        v[c] = 0.0;
    }
    __syncthreads();
    if (!finite) redj[NWV] = 1;
    __syncthreads();
    if (redj[NWV]) {                                                  // (uniform)
        w2_fail<INFER, W>(p, b, x0, y0, n, m, 3, QNAN, tid);
        return;
    }
    auto cost_of = [&](bool row_real, double xi, double yi, double cd, int j, int c) -> double {
        if (j < m) {
            if (!INFER || row_real) {
                const double dx = fabs(xi - yx[c]), dy = fabs(yi - yy[c]);
                return w2_pow(dx > dy ? dx : dy, p.order);
            }
            return cdy[c];
        }
        return (!INFER || row_real) ? cd : 0.0;
    };
    for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) { minv[c] = INF; used[c] = false; }
        int i0 = i, j0 = -1;
        for (;;) {
            const double xi = xs[i0], yi = ys[i0], ui = u[i0];
            const double cd = w2_pow((yi - xi) * 0.5, p.order);
            double best = INF;
            int bj = -1;
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int j = tid + W * c;
                if (j < N && !used[c]) {
                    const double cur = cost_of(i0 < n, xi, yi, cd, j, c) - ui - v[c];
                    if (cur < minv[c]) { minv[c] = cur; way[j] = j0; }
                    if (minv[c] < best) { best = minv[c]; bj = j; }
                }
            }
            // the minimum over the workgroup; among equal minima the lowest column (the one-wavefront kernel's rule)
            const double wmin = tlc_wave_min_f64(best);
            unsigned long long who = __ballot(best == wmin && bj >= 0);
            int wj = 0x7fffffff;
            if (who) {
                // lowest column among the wavefront's candidates
                int cand = (best == wmin && bj >= 0) ? bj : 0x7fffffff;
                for (int o = 32; o; o >>= 1) { const int t = __shfl_xor(cand, o); cand = t < cand ? t : cand; }
                wj = cand;
            }
            if (lane == 0) { red[wave] = wmin; redj[wave] = wj; }
            __syncthreads();
            double delta = red[0];
            int j1 = redj[0];
#pragma unroll
            for (int k = 1; k < NWV; ++k) {
                const double d = red[k];
                const int jj = redj[k];
                if (d < delta || (d == delta && jj < j1)) { delta = d; j1 = jj; }
            }
            if (j1 == 0x7fffffff) {                                   // (uniform; cannot happen with finite inputs)
                __syncthreads();
                w2_fail<INFER, W>(p, b, x0, y0, n, m, 3, QNAN, tid);
                return;
            }
#pragma unroll
            for (int c = 0; c < CPL; ++c) {
                const int j = tid + W * c;
                if (j < N) {
                    if (used[c]) { u[pcol[j]] += delta; v[c] -= delta; }
                    else minv[c] -= delta;
                }
            }
            if (tid == 0) u[i] += delta;
            j0 = j1;
#pragma unroll
            for (int c = 0; c < CPL; ++c) if (tid + W * c == j1) used[c] = true;
            __syncthreads();
            i0 = pcol[j1];
            if (i0 < 0) break;
        }
        if (tid == 0) {
            int j = j0;
            while (j >= 0) {
                const int jp = way[j];
                pcol[j] = jp < 0 ? i : pcol[jp];
                j = jp;
            }
        }
        __syncthreads();
    }
    // ---- the loss and its pieces ------------------------------------------------------------------------------------------
    double sxy = 0.0, sxd = 0.0, syd = 0.0;
#pragma unroll
    for (int c = 0; c < CPL; ++c) {
        const int j = tid + W * c;
        if (j < N) {
            const int i = pcol[j];
            if (i < n) {
                if (j < m) {
                    const double dx = fabs(xs[i] - yx[c]), dy = fabs(ys[i] - yy[c]);
                    const double d = dx > dy ? dx : dy;
                    sxy += p.order == 2 ? d * d : d;
                    p.assign[x0 + i] = j;
                    if (INFER) p.assign_y[y0 + j] = i;
                } else {
                    const double d = fabs((ys[i] - xs[i]) * 0.5);
                    sxd += p.order == 2 ? d * d : d;
                    p.assign[x0 + i] = -1;
                }
            } else if (j < m) {
                const double d = fabs((yy[c] - yx[c]) * 0.5);
                syd += p.order == 2 ? d * d : d;
                p.assign_y[y0 + j] = -1;
            }
        }
    }
    for (int o = 32; o; o >>= 1) { sxy += __shfl_xor(sxy, o); sxd += __shfl_xor(sxd, o); syd += __shfl_xor(syd, o); }
    if (lane == 0) { red[wave] = sxy; red[NWV + wave] = sxd; red[2 * NWV + wave] = syd; }
    __syncthreads();
    if (tid == 0) {
        double a = 0.0, d = 0.0, e = 0.0;
        for (int k = 0; k < NWV; ++k) { a += red[k]; d += red[NWV + k]; e += red[2 * NWV + k]; }
        sums[0] = a; sums[1] = d; sums[2] = e;
    }
    __syncthreads();
    sxy = sums[0]; sxd = sums[1]; syd = sums[2];
    const double tot = sxy + sxd + syd;
    const double L = p.order == 2 ? sqrt(tot) : tot;
    if (tid == 0) {
        p.status[b] = 0;
        p.loss[b] = L;
        p.wxy[b] = p.order == 2 ? sqrt(sxy) : sxy;
        p.wxd[b] = p.order == 2 ? sqrt(sxd) : sxd;
        if (INFER) p.wyd[b] = p.order == 2 ? sqrt(syd) : syd;
    }
    if (p.gradX) {
#pragma unroll
        for (int c = 0; c < CPL; ++c) {
            const int j = tid + W * c;
            if (j < N) {
                const int i = pcol[j];
                if (i >= n) continue;
                double gx = 0.0, gy = 0.0;
                if (j < m) {
                    const double ex = yx[c] - xs[i], ey = yy[c] - ys[i];
                    const double ax = fabs(ex), ay = fabs(ey);
                    const double d = ax > ay ? ax : ay;
                    const double w = p.order == 2 ? (L > 0.0 ? d / L : 0.0) : 1.0;
                    if (ax >= ay) gx = -w * (ex > 0.0 ? 1.0 : (ex < 0.0 ? -1.0 : 0.0));
                    else gy = -w * (ey > 0.0 ? 1.0 : (ey < 0.0 ? -1.0 : 0.0));
                } else {
                    const double s = (ys[i] - xs[i]) * 0.5;
                    const double w = p.order == 2 ? (L > 0.0 ? s / L : 0.0) : (s > 0.0 ? 1.0 : (s < 0.0 ? -1.0 : 0.0));
                    gx = -0.5 * w; gy = 0.5 * w;
                }
                p.gradX[2 * (x0 + i)] = gx;
                p.gradX[2 * (x0 + i) + 1] = gy;
            }
        }
    }
}

template <bool INFER>
static int launch_w2_wide(const W2Params& p, int min_points, hipStream_t s) {
    const size_t lds = (size_t)W2_WIDE_NMAX * (3 * 8 + 2 * 4) + 64 * 8;
    TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_w2_match_wide_kernel<INFER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((tlc_w2_match_wide_kernel<INFER>), dim3(p.n_pairs), dim3(W2_WIDE_THREADS), lds, s, p, min_points);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

template <int CPL, bool INFER>
static int launch_w2(const W2Params& p, hipStream_t s) {
    const size_t lds = (size_t)64 * CPL * (4 * 8 + 3 * 4);
    if (lds > 64 * 1024) TLC_HIP_CHECK(hipFuncSetAttribute((const void*)tlc_w2_match_kernel<CPL, INFER>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int grid = p.n_pairs < 16384 ? p.n_pairs : 16384;
    hipLaunchKernelGGL((tlc_w2_match_kernel<CPL, INFER>), dim3(grid), dim3(64), lds, s, p);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

template <bool INFER>
static int dispatch_w2(W2Params& p, int max_points, hipStream_t s) {
    p.leave_big = max_points > 512;
    if (max_points <= 64) return launch_w2<1, INFER>(p, s);
    if (max_points <= 128) return launch_w2<2, INFER>(p, s);
    if (max_points <= 256) return launch_w2<4, INFER>(p, s);
    int rc = launch_w2<8, INFER>(p, s);
    if (rc != TLC_OK || max_points <= 512) return rc;
    return launch_w2_wide<INFER>(p, 512, s);    // problems of 513 .. 4 096 rows: a workgroup each; beyond: status 2
}

}  // namespace

// max_points: an upper bound of the predicted points of one problem (the caller knows its offsets): picks the kernel variant
// (64 / 128 / 256 / 512 columns per wavefront); above 512 a second launch gives every larger problem a 512-thread workgroup
// (up to 4 096 predicted points); beyond that status 2.
extern "C" int tlc_w2_partial_matching(int32_t n_problems, const int64_t* d_xoff, const double* d_X, const int64_t* d_yoff,
                                       const double* d_Y, int order, int32_t max_points, double* d_loss, double* d_wxy,
                                       double* d_wxd, int32_t* d_assign, double* d_gradX, uint8_t* d_status, void* stream) {
    TLC_REQUIRE(n_problems >= 0, "n_problems < 0");
    TLC_REQUIRE(order == 1 || order == 2, "order must be 1 or 2");
    if (n_problems == 0) return TLC_OK;
    TLC_REQUIRE(d_xoff && d_yoff && d_loss && d_wxy && d_wxd && d_assign && d_status, "null pointer");
    W2Params p;
    memset(&p, 0, sizeof(p));
    p.n_pairs = n_problems; p.xoff = (const long long*)d_xoff; p.X = d_X; p.yoff = (const long long*)d_yoff; p.Y = d_Y;
    p.order = order; p.loss = d_loss; p.wxy = d_wxy; p.wxd = d_wxd; p.assign = d_assign; p.gradX = d_gradX; p.status = d_status;
    return dispatch_w2<false>(p, max_points, (hipStream_t)stream);
}

// The evaluation form (wasserstein_distance_inference, wasserstein.py:93-195): both diagrams may use the diagonal.
// max_points: an upper bound of n + m (predicted + target points) of one problem; beyond 4 096: status 2.
extern "C" int tlc_w2_inference_matching(int32_t n_problems, const int64_t* d_xoff, const double* d_X, const int64_t* d_yoff,
                                         const double* d_Y, int order, int32_t max_points, double* d_loss, double* d_wxy,
                                         double* d_wxd, double* d_wyd, int32_t* d_assign_x, int32_t* d_assign_y, double* d_gradX,
                                         uint8_t* d_status, void* stream) {
    TLC_REQUIRE(n_problems >= 0, "n_problems < 0");
    TLC_REQUIRE(order == 1 || order == 2, "order must be 1 or 2");
    if (n_problems == 0) return TLC_OK;
    TLC_REQUIRE(d_xoff && d_yoff && d_loss && d_wxy && d_wxd && d_wyd && d_assign_x && d_assign_y && d_status, "null pointer");
    W2Params p;
    memset(&p, 0, sizeof(p));
    p.n_pairs = n_problems; p.xoff = (const long long*)d_xoff; p.X = d_X; p.yoff = (const long long*)d_yoff; p.Y = d_Y;
    p.order = order; p.loss = d_loss; p.wxy = d_wxy; p.wxd = d_wxd; p.wyd = d_wyd; p.assign = d_assign_x; p.assign_y = d_assign_y;
    p.gradX = d_gradX; p.status = d_status;
    return dispatch_w2<true>(p, max_points, (hipStream_t)stream);
}

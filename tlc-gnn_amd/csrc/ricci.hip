// ricci.hip -- the producer of the path's edge weights: Ollivier-Ricci curvature of every edge with
// the entropic (Sinkhorn) transport distance, as loaddatas.py:105-123 asks of the third-party GraphRicciCurvature
// (`OllivierRicci(G, alpha=0.5, method="Sinkhorn")`, absent from the reference tree and from this image: the algorithm is
// restated from its published form; the CPU checker under tests/ says which -- parity unpinned).
//
// Per edge (s, t):  m_s = alpha at s, (1 - alpha)/deg(s) on each neighbour (unit weights: base^(-w^p) is the same for all);
// m_t likewise;  cost d(a, b) = hop distance in the whole graph between a in N[s] and b in N[t];  W = <P, d> for the Sinkhorn
// plan P = diag(u) K diag(v), K = exp(-d / reg), iterated like POT's sinkhorn_knopp (v = b / K^T u, u = a / K v, marginal
// violation checked every 10th iteration against stopThr, at most numItermax);  kappa = 1 - W / d(s, t) = 1 - W.
//
// Because s ~ t, every such pair is within three hops (a - s - t - b), so d is 0 (same node), 1 (adjacent), 2 (a common
// neighbour) or else exactly 3, and K has four distinct values: the cost matrix is a byte code per entry, staged in LDS (one
// wavefront per edge; one 256-thread workgroup per hub edge, with an HBM slot for supports beyond the LDS).  u, v and the marginals live in LDS.  Integer/latency-bound graph work plus short fp64 mat-vecs: no MFMA.
#include "tlc_common.h"

namespace {

struct RicciParams {
    int n_nodes;
    const int* rowptr;
    const int* col;
    long long n_edges;
    const int* edges;            // [n_edges, 2]
    double alpha, reg, stop_thr;
    int max_iter;
    double* kappa;               // [n_edges]
    int* iters;                  // [n_edges] or null
    // split between the two kernels
    int small_cap;               // the wavefront kernel takes edges with (deg s + 1) * (deg t + 1) <= small_cap
    int* big_count;              // device counter + list of the edges left to the workgroup kernel
    int* big_list;
    unsigned int* big_codes;     // [slots][slot_bytes / 4]: packed hop codes of the supports beyond the LDS
    long long slot_bytes;
    int max_support;             // LDS capacity of the workgroup kernel for u, v, a, b (entries of each)
};

template <int W>
__device__ __forceinline__ void group_sync() {
    if (W == 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    } else {
        __syncthreads();
    }
}

// sum of one double per thread over the group; red: W doubles of LDS
template <int W>
__device__ __forceinline__ double group_sum(double v, double* red, int tid) {
    v += tlc_lane_xor_f64<1>(v);
    v += tlc_lane_xor_f64<2>(v);
    v += tlc_lane_xor_f64<4>(v);
    v += tlc_lane_xor_f64<8>(v);
    v += tlc_lane_xor_f64<16>(v);
    v += tlc_lane_xor_f64<32>(v);
    if (W == 64) return v;
    group_sync<W>();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    group_sync<W>();
    double s = 0.0;
    for (int w = 0; w < W / 64; ++w) s += red[w];
    return s;
}

// hop codes are packed sixteen to a 32-bit word (a 437 x 404 hub-hub edge: 44 KB instead of 176 KB -- LDS instead of HBM)
__device__ __forceinline__ int code_at(const unsigned int* codes, int q) { return (int)((codes[q >> 4] >> ((q & 15) * 2)) & 3u); }

// out-of-place mat-vec over the coded cost matrix with ALL threads of the group: s(r) = sum_c kval(code(r, c)) * x[c] for
// r < rows.  A short side (a 172 x 5 hub edge has five columns) would leave most lanes idle with one thread per row, so every
// row is cut into P = 2^k <= 64 parts held by P consecutive lanes and folded with lane exchanges.  TRANS: r indexes columns of
// the stored matrix (K^T u).  fin(r, s) is called once per row by the lane holding the folded sum.
template <int W, bool TRANS, class KV, class Fin>
__device__ __forceinline__ void coded_matvec(const unsigned int* codes, int nb, int rows, int cols, const double* x, KV kval,
                                             Fin fin, int tid) {
    int P = 1;
    while (P < 64 && rows * P * 2 <= W) P *= 2;
    const int part = tid & (P - 1);
    const int chunk = (cols + P - 1) / P;
    const int c0 = part * chunk, c1 = min(cols, c0 + chunk);
    const int rows_per_pass = W / P;
    for (int r0 = 0; r0 < rows; r0 += rows_per_pass) {                 // uniform trip count: the exchanges need every lane
        const int r = r0 + tid / P;
        double sum = 0.0;
        if (r < rows) {
            if (TRANS) { for (int c = c0; c < c1; ++c) sum += kval(code_at(codes, c * nb + r)) * x[c]; }
            else       { for (int c = c0; c < c1; ++c) sum += kval(code_at(codes, r * nb + c)) * x[c]; }
        }
        double t;
        t = tlc_lane_xor_f64<1>(sum);  if (P > 1) sum += t;
        t = tlc_lane_xor_f64<2>(sum);  if (P > 2) sum += t;
        t = tlc_lane_xor_f64<4>(sum);  if (P > 4) sum += t;
        t = tlc_lane_xor_f64<8>(sum);  if (P > 8) sum += t;
        t = tlc_lane_xor_f64<16>(sum); if (P > 16) sum += t;
        t = tlc_lane_xor_f64<32>(sum); if (P > 32) sum += t;
        if (r < rows && part == 0) fin(r, sum);
    }
}

// One edge by one group of W threads.  codes: na*nb bytes (LDS, or an HBM slot for the largest supports); u, v: LDS doubles;
// idx: na + nb + 1 ints of LDS.  The marginals are uniform apart from the node's own mass, so they are recomputed on the fly.
template <int W>
__device__ void ricci_edge(const RicciParams& p, long long e, unsigned int* codes, double* u, double* v, int* idx, double* red,
                           int tid) {
    const int s = p.edges[2 * e], t = p.edges[2 * e + 1];
    const int sl = p.rowptr[s], tl = p.rowptr[t];
    const int ds = p.rowptr[s + 1] - sl, dt = p.rowptr[t + 1] - tl;
    const int na = ds + 1, nb = dt + 1;
    // support a_i: the neighbours of s, then s itself (mass alpha); the target support likewise (its ids are staged in LDS below)
    auto sup_a = [&](int i) { return i < ds ? p.col[sl + i] : s; };
    const double ma = ds > 0 ? (1.0 - p.alpha) / (double)ds : 0.0, mb = dt > 0 ? (1.0 - p.alpha) / (double)dt : 0.0;
    auto mass_a = [&](int i) { return i < ds ? ma : (ds > 0 ? p.alpha : 1.0); };
    auto mass_b = [&](int j) { return j < dt ? mb : (dt > 0 ? p.alpha : 1.0); };
    // Hop codes.  One entry at a time (two dependent binary searches over global rows per entry) took 4.7 ms on a 172 x 172 hub
    // edge; instead the rows around the source support are streamed once: the target support's ids sit in LDS (sorted), every
    // entry starts at 3, and for every (a_i, y in row(a_i)) unit -- dealt to the threads through a prefix over deg(a_i) -- the
    // row of y marks distance 2, y itself distance 1, a_i itself distance 0 (later passes overwrite earlier ones).
    int* const bid = idx;                    // [nb - 1] neighbours of t, ascending
    int* const off = idx + nb;               // [na + 1] prefix of deg(a_i)
    for (int j = tid; j < dt; j += W) bid[j] = p.col[tl + j];
    for (int i = tid; i < na; i += W) { const int a = sup_a(i); off[i + 1] = p.rowptr[a + 1] - p.rowptr[a]; }
    for (int q = tid; q < (na * nb + 15) / 16; q += W) codes[q] = 0xffffffffu;         // every entry 3
    group_sync<W>();
    if (tid == 0) {
        int run = 0;
        off[0] = 0;
        for (int i = 0; i < na; ++i) { run += off[i + 1]; off[i + 1] = run; }
    }
    group_sync<W>();
    const int units = off[na];
    auto pos_b = [&](int z) -> int {         // index of z in the target support, or -1
        if (z == t) return dt;
        int lo = 0, hi = dt;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const int c = bid[mid];
            if (c == z) return mid;
            if (c < z) lo = mid + 1; else hi = mid;
        }
        return -1;
    };
    auto unit_row = [&](int k) -> int {      // the i whose row holds unit k: last i with off[i] <= k
        int lo = 0, hi = na;
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (off[mid] <= k) lo = mid; else hi = mid;
        }
        return lo;
    };
    for (int k = tid; k < units; k += W) {
        const int i = unit_row(k);
        const int a = sup_a(i);
        const int y = p.col[p.rowptr[a] + (k - off[i])];
        const int yl = p.rowptr[y], yh = p.rowptr[y + 1];
        for (int q = yl; q < yh; ++q) {
            const int j = pos_b(p.col[q]);
            if (j >= 0) { const int q2 = i * nb + j; atomicAnd(&codes[q2 >> 4], ~(1u << ((q2 & 15) * 2))); }        // 3 -> 2
        }
    }
    group_sync<W>();
    for (int k = tid; k < units; k += W) {
        const int i = unit_row(k);
        const int j = pos_b(p.col[p.rowptr[sup_a(i)] + (k - off[i])]);
        if (j >= 0) {                                                                           // -> 1
            const int q1 = i * nb + j;
            atomicAnd(&codes[q1 >> 4], ~(3u << ((q1 & 15) * 2)));
            atomicOr(&codes[q1 >> 4], 1u << ((q1 & 15) * 2));
        }
    }
    group_sync<W>();
    for (int i = tid; i < na; i += W) {
        const int j = pos_b(sup_a(i));
        if (j >= 0) { const int q0 = i * nb + j; atomicAnd(&codes[q0 >> 4], ~(3u << ((q0 & 15) * 2))); }            // -> 0
    }
    for (int i = tid; i < na; i += W) u[i] = 1.0 / (double)na;
    for (int j = tid; j < nb; j += W) v[j] = 1.0 / (double)nb;
    const double kv0 = 1.0, kv1 = exp(-1.0 / p.reg), kv2 = exp(-2.0 / p.reg), kv3 = exp(-3.0 / p.reg);
    auto kval = [&](int c) { return c == 0 ? kv0 : (c == 1 ? kv1 : (c == 2 ? kv2 : kv3)); };
    group_sync<W>();
    int ii = 0;
    double err = 1.0;
    while (err > p.stop_thr && ii < p.max_iter) {
        // v = b / (K^T u)
        coded_matvec<W, true>(codes, nb, nb, na, u, kval, [&](int j, double ktu) { v[j] = mass_b(j) / ktu; }, tid);
        group_sync<W>();
        // u = 1 / ((K / a) v) = a / (K v)
        coded_matvec<W, false>(codes, nb, na, nb, v, kval, [&](int i, double kvs) { u[i] = mass_a(i) / kvs; }, tid);
        group_sync<W>();
        if (ii % 10 == 0) {
            // marginal violation || v * (K^T u) - b ||_2
            double part = 0.0;
            coded_matvec<W, true>(codes, nb, nb, na, u, kval, [&](int j, double ktu) {
                const double dlt = ktu * v[j] - mass_b(j);
                part += dlt * dlt;
            }, tid);
            err = sqrt(group_sum<W>(part, red, tid));
            if (!(err == err)) break;                                   // NaN: stop with what there is
        }
        ++ii;
    }
    // W = sum_ij u_i K_ij v_j d_ij
    double part = 0.0;
    auto kd = [&](int c) { return kval(c) * (double)c; };
    coded_matvec<W, false>(codes, nb, na, nb, v, kd, [&](int i, double row) { part += u[i] * row; }, tid);
    const double wd = group_sum<W>(part, red, tid);
    if (tid == 0) {
        p.kappa[e] = 1.0 - wd;                                           // / d(s, t), which is 1
        if (p.iters) p.iters[e] = ii;
    }
    group_sync<W>();
}

#define RICCI_SMALL_CODES 8192          /* hop codes (2 bits each) per wavefront: 2 KB */
#define RICCI_SMALL_SUPPORT 256         /* na + nb limit of the wavefront kernel */

// one wavefront per edge, four per workgroup, everything in LDS; larger edges go to the list of the workgroup kernel
__global__ __launch_bounds__(256) void ricci_small_kernel(RicciParams p) {
    __shared__ __attribute__((aligned(16))) unsigned int s_codes[4][RICCI_SMALL_CODES / 16];
    __shared__ double s_uv[4][RICCI_SMALL_SUPPORT];
    __shared__ int s_idx[4][RICCI_SMALL_SUPPORT + 2];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long long n_waves = (long long)gridDim.x * 4;
    for (long long e = (long long)blockIdx.x * 4 + wv; e < p.n_edges; e += n_waves) {
        const int s = p.edges[2 * e], t = p.edges[2 * e + 1];
        const bool bad = s < 0 || t < 0 || s >= p.n_nodes || t >= p.n_nodes || s == t;
        if (bad) {
            if (lane == 0) { p.kappa[e] = 0.0; if (p.iters) p.iters[e] = 0; }   // self loop: curvature 0 (the library's convention)
            continue;
        }
        const int na = p.rowptr[s + 1] - p.rowptr[s] + 1, nb = p.rowptr[t + 1] - p.rowptr[t] + 1;
        if ((long long)na * nb > p.small_cap || na + nb > RICCI_SMALL_SUPPORT) {
            if (lane == 0) p.big_list[atomicAdd(p.big_count, 1)] = (int)e;
            continue;
        }
        ricci_edge<64>(p, e, s_codes[wv], s_uv[wv], s_uv[wv] + na, s_idx[wv], nullptr, lane);
    }
}

// hub edges: RICCI_BIG_THREADS threads per edge (a 437 x 404 hub-hub edge with 181 iterations kept a 256-thread workgroup for 44 ms), u / v and (when they fit: lds_codes bytes) the codes in LDS, else the codes in an HBM slot
#define RICCI_BIG_THREADS 1024
__global__ __launch_bounds__(RICCI_BIG_THREADS) void ricci_big_kernel(RicciParams p, int lds_codes) {
    extern __shared__ __attribute__((aligned(16))) double s_big[];
    __shared__ double s_red[RICCI_BIG_THREADS / 64];
    int* const s_idx = reinterpret_cast<int*>(s_big + p.max_support);                    // [max_support + 2]
    unsigned int* lds_code_base = reinterpret_cast<unsigned int*>(s_idx + ((p.max_support + 2 + 3) & ~3));
    const int n_big = *p.big_count;
    for (int k = blockIdx.x; k < n_big; k += gridDim.x) {
        const long long e = p.big_list[k];
        const int s = p.edges[2 * e], t = p.edges[2 * e + 1];
        const int na = p.rowptr[s + 1] - p.rowptr[s] + 1, nb = p.rowptr[t + 1] - p.rowptr[t] + 1;
        const bool in_lds = (long long)na * nb <= (long long)lds_codes * 4;                     // four codes per byte
        if (na + nb > p.max_support || (!in_lds && (long long)na * nb > p.slot_bytes * 4)) {      // beyond the caller's workspace: loud
            if (threadIdx.x == 0) { p.kappa[e] = __longlong_as_double(0x7ff8000000000000LL); if (p.iters) p.iters[e] = -1; }
            continue;
        }
        ricci_edge<RICCI_BIG_THREADS>(p, e, in_lds ? lds_code_base : p.big_codes + (size_t)blockIdx.x * (size_t)(p.slot_bytes / 4), s_big, s_big + na, s_idx,
                        s_red, threadIdx.x);
    }
}

}  // namespace

extern "C" int tlc_ollivier_ricci_sinkhorn(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, int64_t n_edges,
                                           const int32_t* d_edges, double alpha, double reg, int32_t max_iter, double stop_thr,
                                           double* d_kappa, int32_t* d_iters, void* d_work, int64_t work_bytes,
                                           int32_t max_support, int64_t max_product, void* stream) {
    TLC_REQUIRE(n_nodes >= 0 && n_edges >= 0, "negative size");
    if (n_edges == 0) return TLC_OK;
    TLC_REQUIRE(d_rowptr && d_col && d_edges && d_kappa, "null pointer");
    TLC_REQUIRE(alpha >= 0.0 && alpha <= 1.0 && reg > 0.0 && max_iter >= 0, "bad parameter");
    TLC_REQUIRE(n_edges < (1ll << 31), "more than 2^31 - 1 edges");
    // workspace: [counter 16 B][big list int32[n_edges]][slots x slot_bytes]
    const int64_t list_bytes = ((int64_t)n_edges * 4 + 15) & ~15ll;
    const int64_t slot_bytes = (((max_product + 3) / 4) + 15) & ~15ll;                 // 2 bits per hop code
    TLC_REQUIRE(d_work && work_bytes >= 16 + list_bytes + slot_bytes, "workspace too small: 16 + 4*n_edges + >= 1 slot of max_product/4 bytes");
    TLC_REQUIRE(max_support >= 2 && (int64_t)max_support * 12 + 16 <= 150 * 1024, "max_support beyond the LDS of a workgroup (12 800 entries)");
    hipStream_t s = (hipStream_t)stream;
    RicciParams p;
    p.n_nodes = n_nodes; p.rowptr = d_rowptr; p.col = d_col; p.n_edges = n_edges; p.edges = d_edges;
    p.alpha = alpha; p.reg = reg; p.stop_thr = stop_thr; p.max_iter = max_iter; p.kappa = d_kappa; p.iters = d_iters;
    p.small_cap = RICCI_SMALL_CODES;
    p.big_count = (int*)d_work;
    p.big_list = (int*)((char*)d_work + 16);
    p.big_codes = (unsigned int*)((unsigned char*)d_work + 16 + list_bytes);
    p.slot_bytes = slot_bytes;
    p.max_support = max_support;
    int64_t slots = slot_bytes > 0 ? (work_bytes - 16 - list_bytes) / slot_bytes : 1;
    if (slots > 1024) slots = 1024;
    TLC_HIP_CHECK(hipMemsetAsync(d_work, 0, 16, s));
    long long blocks = (n_edges + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ricci_small_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    // LDS of the workgroup kernel: u and v (max_support doubles) + as many code bytes as the largest product needs, up to 150 KB
    const int64_t idx_bytes = 4 * (((int64_t)max_support + 2 + 3) & ~3ll);
    int64_t lds_codes = 150 * 1024 - (int64_t)max_support * 8 - idx_bytes;
    if (lds_codes > slot_bytes) lds_codes = slot_bytes;
    if (lds_codes < 0) lds_codes = 0;
    const size_t lds = (size_t)max_support * 8 + (size_t)idx_bytes + (size_t)lds_codes;
    if (lds > 64 * 1024) TLC_HIP_CHECK(hipFuncSetAttribute((const void*)ricci_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ricci_big_kernel, dim3((unsigned)slots), dim3(RICCI_BIG_THREADS), lds, s, p, (int)lds_codes);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

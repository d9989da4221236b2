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
// Because s ~ t, every such pair is within three hops (a - s - t - b), so d is 0 (same node), 1 (adjacent: binary search in
// the sorted row), 2 (the two sorted rows intersect) or else exactly 3, and K has four distinct values: the cost matrix is a
// byte code per entry, staged in LDS (one wavefront per edge) or, for hub edges, in an HBM slot (one 256-thread workgroup
// per edge).  u, v and the marginals live in LDS.  Integer/latency-bound graph work plus short fp64 mat-vecs: no MFMA.
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
    unsigned char* big_codes;    // [slots][slot_bytes]
    long long slot_bytes;
    int max_support;             // LDS capacity of the workgroup kernel for u, v, a, b (entries of each)
};

__device__ __forceinline__ bool row_has(const int* __restrict__ col, int lo, int hi, int x) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        const int c = col[mid];
        if (c == x) return true;
        if (c < x) lo = mid + 1; else hi = mid;
    }
    return false;
}

// hop distance between a in N[s] and b in N[t] for adjacent s, t: 0, 1, 2 or 3
__device__ __forceinline__ int support_distance(const int* __restrict__ rowptr, const int* __restrict__ col, int a, int b) {
    if (a == b) return 0;
    int al = rowptr[a], ah = rowptr[a + 1], bl = rowptr[b], bh = rowptr[b + 1];
    if (ah - al > bh - bl) { int t = al; al = bl; bl = t; t = ah; ah = bh; bh = t; t = a; a = b; b = t; }   // a = shorter row
    if (row_has(col, al, ah, b)) return 1;
    for (int i = al; i < ah; ++i)
        if (row_has(col, bl, bh, col[i])) return 2;
    return 3;
}

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

// One edge by one group of W threads.  codes: na*nb bytes (LDS or HBM); fl: 2*(na+nb) doubles of LDS (u, v) -- the marginals
// are uniform apart from the node's own mass, so they are recomputed on the fly.
template <int W>
__device__ void ricci_edge(const RicciParams& p, long long e, unsigned char* codes, double* u, double* v, double* red, int tid) {
    const int s = p.edges[2 * e], t = p.edges[2 * e + 1];
    const int sl = p.rowptr[s], tl = p.rowptr[t];
    const int ds = p.rowptr[s + 1] - sl, dt = p.rowptr[t + 1] - tl;
    const int na = ds + 1, nb = dt + 1;
    // support a_i: the neighbours of s, then s itself (mass alpha); same for t
    auto sup_a = [&](int i) { return i < ds ? p.col[sl + i] : s; };
    auto sup_b = [&](int j) { return j < dt ? p.col[tl + j] : t; };
    const double ma = ds > 0 ? (1.0 - p.alpha) / (double)ds : 0.0, mb = dt > 0 ? (1.0 - p.alpha) / (double)dt : 0.0;
    auto mass_a = [&](int i) { return i < ds ? ma : (ds > 0 ? p.alpha : 1.0); };
    auto mass_b = [&](int j) { return j < dt ? mb : (dt > 0 ? p.alpha : 1.0); };
    for (int q = tid; q < na * nb; q += W) {
        const int i = q / nb, j = q - i * nb;
        codes[q] = (unsigned char)support_distance(p.rowptr, p.col, sup_a(i), sup_b(j));
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
        for (int j = tid; j < nb; j += W) {
            double ktu = 0.0;
            for (int i = 0; i < na; ++i) ktu += kval(codes[i * nb + j]) * u[i];
            v[j] = mass_b(j) / ktu;
        }
        group_sync<W>();
        // u = 1 / ((K / a) v)
        for (int i = tid; i < na; i += W) {
            double kvs = 0.0;
            const double inv_a = 1.0 / mass_a(i);
            for (int j = 0; j < nb; ++j) kvs += (inv_a * kval(codes[i * nb + j])) * v[j];
            u[i] = 1.0 / kvs;
        }
        group_sync<W>();
        if (ii % 10 == 0) {
            // marginal violation || v * (K^T u) - b ||_2
            double part = 0.0;
            for (int j = tid; j < nb; j += W) {
                double ktu = 0.0;
                for (int i = 0; i < na; ++i) ktu += u[i] * kval(codes[i * nb + j]);
                const double dlt = ktu * v[j] - mass_b(j);
                part += dlt * dlt;
            }
            err = sqrt(group_sum<W>(part, red, tid));
            if (!(err == err)) break;                                   // NaN: stop with what there is
        }
        ++ii;
    }
    // W = sum_ij u_i K_ij v_j d_ij
    double part = 0.0;
    for (int i = tid; i < na; i += W) {
        double row = 0.0;
        for (int j = 0; j < nb; ++j) {
            const int c = codes[i * nb + j];
            row += (kval(c) * v[j]) * (double)c;
        }
        part += u[i] * row;
    }
    const double wd = group_sum<W>(part, red, tid);
    if (tid == 0) {
        p.kappa[e] = 1.0 - wd;                                           // / d(s, t), which is 1
        if (p.iters) p.iters[e] = ii;
    }
    group_sync<W>();
}

#define RICCI_SMALL_CODES 3072          /* bytes of codes per wavefront */
#define RICCI_SMALL_SUPPORT 256         /* na + nb limit of the wavefront kernel */

// one wavefront per edge, four per workgroup, everything in LDS; larger edges go to the list of the workgroup kernel
__global__ __launch_bounds__(256) void ricci_small_kernel(RicciParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char s_codes[4][RICCI_SMALL_CODES];
    __shared__ double s_uv[4][RICCI_SMALL_SUPPORT];
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
        ricci_edge<64>(p, e, s_codes[wv], s_uv[wv], s_uv[wv] + na, nullptr, lane);
    }
}

// hub edges: 256 threads per edge, codes in an HBM slot (L2 resident), u / v in LDS
__global__ __launch_bounds__(256) void ricci_big_kernel(RicciParams p) {
    extern __shared__ __attribute__((aligned(16))) double s_big[];
    __shared__ double s_red[4];
    const int n_big = *p.big_count;
    for (int k = blockIdx.x; k < n_big; k += gridDim.x) {
        const long long e = p.big_list[k];
        const int s = p.edges[2 * e], t = p.edges[2 * e + 1];
        const int na = p.rowptr[s + 1] - p.rowptr[s] + 1, nb = p.rowptr[t + 1] - p.rowptr[t] + 1;
        if (na + nb > p.max_support || (long long)na * nb > p.slot_bytes) {        // does not fit the caller's workspace: loud
            if (threadIdx.x == 0) { p.kappa[e] = __longlong_as_double(0x7ff8000000000000LL); if (p.iters) p.iters[e] = -1; }
            continue;
        }
        ricci_edge<256>(p, e, p.big_codes + (size_t)blockIdx.x * p.slot_bytes, s_big, s_big + na, s_red, threadIdx.x);
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
    const int64_t slot_bytes = (max_product + 15) & ~15ll;
    TLC_REQUIRE(d_work && work_bytes >= 16 + list_bytes + slot_bytes, "workspace too small: 16 + 4*n_edges + >= 1 slot of max_product bytes");
    TLC_REQUIRE(max_support >= 2 && (int64_t)max_support * 8 <= 150 * 1024, "max_support beyond the LDS of a workgroup (19 200 entries)");
    hipStream_t s = (hipStream_t)stream;
    RicciParams p;
    p.n_nodes = n_nodes; p.rowptr = d_rowptr; p.col = d_col; p.n_edges = n_edges; p.edges = d_edges;
    p.alpha = alpha; p.reg = reg; p.stop_thr = stop_thr; p.max_iter = max_iter; p.kappa = d_kappa; p.iters = d_iters;
    p.small_cap = RICCI_SMALL_CODES;
    p.big_count = (int*)d_work;
    p.big_list = (int*)((char*)d_work + 16);
    p.big_codes = (unsigned char*)d_work + 16 + list_bytes;
    p.slot_bytes = slot_bytes;
    p.max_support = max_support;
    int64_t slots = slot_bytes > 0 ? (work_bytes - 16 - list_bytes) / slot_bytes : 1;
    if (slots > 1024) slots = 1024;
    TLC_HIP_CHECK(hipMemsetAsync(d_work, 0, 16, s));
    long long blocks = (n_edges + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ricci_small_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p);
    const size_t lds = (size_t)max_support * 8;
    if (lds > 64 * 1024) TLC_HIP_CHECK(hipFuncSetAttribute((const void*)ricci_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(ricci_big_kernel, dim3((unsigned)slots), dim3(256), lds, s, p);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

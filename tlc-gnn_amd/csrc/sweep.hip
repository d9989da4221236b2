// sweep.hip -- the steps either side of the PD/PI batch when the pair list is the reference's NEGATIVE SWEEP
// (tlc_near_pairs, the distance pre-filter of SURVEY.md 8d, is at the end of the file):
//
//   tlc_complement_rows / tlc_complement_pairs
//       loaddatas.py:44-45 enumerates the non-edges as `sp.triu(sp.csr_matrix(1. - adj.toarray())).nonzero()`: a dense
//       N x N float64 matrix (3.1 GB for PubMed) and a [1.9e8, 2] int64 pair array, only to be shuffled and sliced.  The
//       same list is a function of the CSR: pair number r (row-major over x <= y, adj[x,y] == 0, diagonal included) is
//       found by two binary searches -- over the per-row complement prefix, then over the row's sorted neighbours for
//       "the t-th column >= x that is not a neighbour".  One thread per requested rank; ranks are either a contiguous
//       range or the caller's (shuffled) index list, so the pairs of any slice of the reference's shuffled negative list
//       are produced on the device without ever materialising the list.
//   tlc_select_rows
//       the sweep's images are zero for every pair with d(u,v) > hop (99.7 % of PubMed's 1.9e8): keeps (index, status, row)
//       of the non-zero rows plus a histogram of the status bytes (all the reference keeps of its swallowed exceptions is
//       their number, `cnt_compute`), which is what the sparse image cache stores (the reference caches the dense
//       float64[n_pairs, 25]: 39 GB).
//
// Integer work on cache-resident tables: no MFMA; the searches run out of L2 (the CSR is ~1 MB), the pre-filter's
// breadth-first levels on LDS bitmaps.
#include "tlc_common.h"

namespace {

// first position in [lo, hi) whose column is >= x (columns ascending)
__device__ __forceinline__ int lower_bound_col(const int* __restrict__ col, int lo, int hi, int x) {
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (col[mid] < x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// row x: columns x..n-1 minus the stored neighbours among them
__global__ void complement_count_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                        long long* __restrict__ row_start) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= n) return;
    const int rs = rowptr[x], re = rowptr[x + 1];
    const int lb = lower_bound_col(col, rs, re, x);
    row_start[x + 1] = (long long)(n - x) - (long long)(re - lb);
    if (x == 0) row_start[0] = 0;
}

// in-place inclusive scan of row_start[1..n] by ONE workgroup (a one-off per graph; n <= ~5e5)
__global__ __launch_bounds__(1024) void complement_scan_kernel(int n, long long* __restrict__ row_start) {
    __shared__ long long part[1024];
    const int tid = threadIdx.x;
    const int chunk = (n + 1023) / 1024;
    const int lo = tid * chunk, hi = min(n, lo + chunk);
    long long s = 0;
    for (int i = lo; i < hi; ++i) s += row_start[i + 1];
    part[tid] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const long long v = tid >= off ? part[tid - off] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    long long run = tid ? part[tid - 1] : 0;
    for (int i = lo; i < hi; ++i) {
        run += row_start[i + 1];
        row_start[i + 1] = run;
    }
}

__global__ void complement_pairs_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                        const long long* __restrict__ row_start, const long long* __restrict__ ranks,
                                        long long first, long long count, int* __restrict__ pairs) {
    const long long total = row_start[n];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (long long)gridDim.x * blockDim.x) {
        const long long r = ranks ? ranks[i] : first + i;
        int x = -1, y = -1;
        if (r >= 0 && r < total) {
            // row: the last x with row_start[x] <= r (rows without a non-edge share their start with the next row)
            int lo = 0, hi = n;                    // invariant: row_start[lo] <= r < row_start[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (row_start[mid] <= r) lo = mid; else hi = mid;
            }
            x = lo;
            const long long t = r - row_start[x];  // t-th missing column among x, x+1, ...
            const int re = rowptr[x + 1];
            const int lb = lower_bound_col(col, rowptr[x], re, x);
            // j = number of neighbours below the answer: the first j with (col[lb+j] - x) - j > t; that count is monotone
            int jl = 0, jh = re - lb;
            while (jl < jh) {
                const int mid = (jl + jh) >> 1;
                if ((long long)(col[lb + mid] - x) - mid > t) jh = mid; else jl = mid + 1;
            }
            y = (int)((long long)x + t + jl);
        }
        reinterpret_cast<int2*>(pairs)[i] = make_int2(x, y);
    }
}

// rows of a float64 [n, width] image block that carry information: any entry != 0 (keep_failed: or status != 0); the status
// histogram counts every row (wave-aggregated: one atomic per status value and wavefront)
__global__ void select_rows_kernel(long long n, int width, const double* __restrict__ pi, const unsigned char* __restrict__ status,
                                   long long index_base, long long cap, int keep_failed, unsigned long long* __restrict__ count,
                                   unsigned long long* __restrict__ hist, long long* __restrict__ out_idx,
                                   unsigned char* __restrict__ out_status, double* __restrict__ out_rows) {
    for (long long i0 = (long long)blockIdx.x * blockDim.x; i0 < n; i0 += (long long)gridDim.x * blockDim.x) {
        const long long i = i0 + threadIdx.x;
        bool keep = false;
        unsigned char st = 0;
        if (i < n) {
            st = status ? status[i] : 0;
            keep = keep_failed && st != 0;
            const double* __restrict__ row = pi + (size_t)i * width;
            for (int q = 0; q < width && !keep; ++q) keep = row[q] != 0.0;
        }
        if (hist) {
            for (int v = 0; v < 8; ++v) {
                const unsigned long long hm = __builtin_amdgcn_ballot_w64(i < n && (st & 7) == v);
                if (hm && tlc_lane() == __builtin_ctzll(hm)) atomicAdd(&hist[v], (unsigned long long)__popcll(hm));
            }
        }
        const unsigned long long mk = __builtin_amdgcn_ballot_w64(keep);
        if (mk == 0) continue;
        unsigned long long base = 0;
        const int leader = __builtin_ctzll(mk);
        if (tlc_lane() == leader) base = atomicAdd(count, (unsigned long long)__popcll(mk));
        base = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(base >> 32), leader) << 32) |
               (unsigned)__builtin_amdgcn_readlane((int)(unsigned)base, leader);
        if (keep) {
            const long long o = (long long)base + __popcll(mk & tlc_lanemask_lt());
            if (o < cap) {
                out_idx[o] = index_base + i;
                if (out_status) out_status[o] = st;
                const double* __restrict__ row = pi + (size_t)i * width;
                for (int q = 0; q < width; ++q) out_rows[(size_t)o * width + q] = row[q];
            }
        }
    }
}

}  // namespace

extern "C" int tlc_complement_rows(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, int64_t* d_row_start,
                                   void* stream) {
    TLC_REQUIRE(n_nodes >= 0, "n_nodes < 0");
    TLC_REQUIRE(d_row_start != nullptr, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    if (n_nodes == 0) {
        TLC_HIP_CHECK(hipMemsetAsync(d_row_start, 0, sizeof(int64_t), s));
        return TLC_OK;
    }
    TLC_REQUIRE(d_rowptr != nullptr, "null pointer");
    hipLaunchKernelGGL(complement_count_kernel, dim3((n_nodes + 255) / 256), dim3(256), 0, s, n_nodes, d_rowptr, d_col,
                       (long long*)d_row_start);
    hipLaunchKernelGGL(complement_scan_kernel, dim3(1), dim3(1024), 0, s, n_nodes, (long long*)d_row_start);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

extern "C" int tlc_complement_pairs(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, const int64_t* d_row_start,
                                    const int64_t* d_ranks, int64_t first, int64_t count, int32_t* d_pairs, void* stream) {
    TLC_REQUIRE(n_nodes >= 0 && count >= 0, "negative size");
    if (count == 0) return TLC_OK;
    TLC_REQUIRE(n_nodes > 0, "no nodes");
    TLC_REQUIRE(d_rowptr && d_row_start && d_pairs, "null pointer");
    long long blocks = (count + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(complement_pairs_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, n_nodes, d_rowptr,
                       d_col, (const long long*)d_row_start, (const long long*)d_ranks, (long long)first, (long long)count,
                       d_pairs);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

extern "C" int tlc_select_rows(int64_t n_rows, int32_t width, const double* d_pi, const uint8_t* d_status, int64_t index_base,
                               int64_t cap, uint32_t flags, uint64_t* d_count, uint64_t* d_status_hist, int64_t* d_out_idx,
                               uint8_t* d_out_status, double* d_out_rows, void* stream) {
    TLC_REQUIRE(n_rows >= 0 && width >= 1 && cap >= 0, "bad size");
    TLC_REQUIRE(d_count != nullptr, "null counter");
    if (n_rows == 0) return TLC_OK;
    TLC_REQUIRE(d_pi && (cap == 0 || (d_out_idx && d_out_rows)), "null pointer");
    long long blocks = (n_rows + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(select_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long long)n_rows, width,
                       d_pi, d_status, (long long)index_base, (long long)cap, (int)(flags & TLC_SELECT_KEEP_FAILED),
                       (unsigned long long*)d_count, (unsigned long long*)d_status_hist, (long long*)d_out_idx, d_out_status,
                       d_out_rows);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// ======================================================================================================================
// tlc_near_pairs: the distance <= hop pre-filter of the sweep (PI-C).  An image row can be non-zero only when
// d(u,v) <= hop (otherwise u, v are outside their own vicinity, every filtration value is the sentinel and every point has
// persistence 0: SURVEY.md A.6, Z0), so of PubMed's 1.9e8 non-edges only the few million inside each other's hop-ball need the
// PD/PI pipeline at all.  One wavefront per source node u: breadth-first levels over three LDS bitmaps (visited / current /
// next, atomicOr), then every visited v >= u that is not adjacent to u is appended -- with its number in the reference's
// negative list (row_start[u] + (v - u) - #neighbours of u in [u, v)) -- through one wave-aggregated atomic per 64 pairs.
// ======================================================================================================================
namespace {

__global__ __launch_bounds__(256) void near_pairs_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                         const long long* __restrict__ row_start, int hop, long long cap,
                                                         unsigned long long* __restrict__ count, long long* __restrict__ out_rank,
                                                         int* __restrict__ out_pairs) {
    extern __shared__ unsigned int bm_all[];
    const int W = (n + 31) >> 5;
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63, wpb = blockDim.x >> 6;     // 4 wavefronts per workgroup, or 1
    unsigned int* visited = bm_all + (size_t)wv * 3 * W;
    unsigned int* cur = visited + W;
    unsigned int* nxt = cur + W;
    const long long n_waves = (long long)gridDim.x * wpb;
    auto fence = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    for (long long u = (long long)blockIdx.x * wpb + wv; u < n; u += n_waves) {
        for (int w = lane; w < W; w += 64) { visited[w] = 0u; cur[w] = 0u; }
        fence();
        if (lane == 0) { visited[u >> 5] = 1u << (u & 31); cur[u >> 5] = 1u << (u & 31); }
        fence();
        for (int level = 0; level < hop; ++level) {
            for (int w = lane; w < W; w += 64) nxt[w] = 0u;
            fence();
            for (int w = lane; w < W; w += 64) {
                unsigned bits = cur[w];
                while (bits) {
                    const int x = (w << 5) + __builtin_ctz(bits);
                    bits &= bits - 1;
                    const int xe = rowptr[x + 1];
                    for (int j = rowptr[x]; j < xe; ++j) {
                        const int y = col[j];
                        const unsigned bit = 1u << (y & 31);
                        if (!(atomicOr(&visited[y >> 5], bit) & bit)) atomicOr(&nxt[y >> 5], bit);
                    }
                }
            }
            fence();
            unsigned int* t = cur; cur = nxt; nxt = t;
        }
        // emit the visited v >= u that are not neighbours of u, in lockstep rounds (one wave-aggregated append per round)
        const int ub = rowptr[u], ue = rowptr[u + 1];
        const int lb_u = lower_bound_col(col, ub, ue, (int)u);
        int w = (int)(u >> 5) + lane;
        unsigned bits = 0u;
        if (w < W) { bits = visited[w]; if (w == (int)(u >> 5)) bits &= ~((1u << (u & 31)) - 1u); }
        while (true) {
            while (bits == 0u && w < W) { w += 64; if (w < W) bits = visited[w]; }
            const bool have = w < W && bits != 0u;
            if (__builtin_amdgcn_ballot_w64(have) == 0ull) break;
            int v = -1;
            long long rank = -1;
            bool emit = false;
            if (have) {
                v = (w << 5) + __builtin_ctz(bits);
                bits &= bits - 1;
                const int lb_v = lower_bound_col(col, ub, ue, v);
                const bool adjacent = lb_v < ue && col[lb_v] == v;
                if (!adjacent) { emit = true; rank = row_start[u] + (long long)(v - (int)u) - (long long)(lb_v - lb_u); }
            }
            const unsigned long long mk = __builtin_amdgcn_ballot_w64(emit);
            if (mk) {
                unsigned long long base = 0;
                const int leader = __builtin_ctzll(mk);
                if (lane == leader) base = atomicAdd(count, (unsigned long long)__popcll(mk));
                base = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(base >> 32), leader) << 32) |
                       (unsigned)__builtin_amdgcn_readlane((int)(unsigned)base, leader);
                if (emit) {
                    const long long o = (long long)base + __popcll(mk & tlc_lanemask_lt());
                    if (o < cap) {
                        out_rank[o] = rank;
                        reinterpret_cast<int2*>(out_pairs)[o] = make_int2((int)u, v);
                    }
                }
            }
        }
        fence();
    }
}

}  // namespace

extern "C" int tlc_near_pairs(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, const int64_t* d_row_start, int hop,
                              int64_t cap, uint64_t* d_count, int64_t* d_out_rank, int32_t* d_out_pairs, void* stream) {
    TLC_REQUIRE(n_nodes >= 0 && hop >= 1 && cap >= 0, "bad argument");
    TLC_REQUIRE(d_count != nullptr, "null counter");
    if (n_nodes == 0) return TLC_OK;
    TLC_REQUIRE(d_rowptr && d_row_start && (cap == 0 || (d_out_rank && d_out_pairs)), "null pointer");
    const size_t words = ((size_t)n_nodes + 31) / 32;
    // three bitmaps per wavefront: four wavefronts per workgroup while they fit the LDS (~100 000 nodes), else one (~400 000)
    int wpb = 4;
    size_t lds = (size_t)wpb * 3 * words * 4;
    if (lds > 150 * 1024) { wpb = 1; lds = 3 * words * 4; }
    if (lds > 150 * 1024) { tlc_set_error("tlc_near_pairs: graph too large for the LDS bitmaps (%d nodes; limit ~400 000)", n_nodes); return TLC_ERR_UNSUPPORTED; }
    if (lds > 64 * 1024) TLC_HIP_CHECK(hipFuncSetAttribute((const void*)near_pairs_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    long long blocks = ((long long)n_nodes + wpb - 1) / wpb;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(near_pairs_kernel, dim3((unsigned)blocks), dim3(64 * wpb), lds, (hipStream_t)stream, n_nodes, d_rowptr, d_col,
                       (const long long*)d_row_start, hop, (long long)cap, (unsigned long long*)d_count, (long long*)d_out_rank,
                       d_out_pairs);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// ---- tlc_pack_vicinities: the capacity layout of tlc_vicinity_filtration -> one packed block-diagonal batch ------------------
namespace {
// one wavefront per vicinity: its n ids (through the label table), n values and m edges from the per-pair capacity slots to
// the packed arrays at node_ptr[i] / edge_ptr[i]; the owner of every node / edge beside them
__global__ __launch_bounds__(256) void pack_vicinities_kernel(long long n_pairs, const long long* __restrict__ node_offs,
                                                            const int* __restrict__ ids, const double* __restrict__ f,
                                                            const long long* __restrict__ edge_offs, const int* __restrict__ edges,
                                                            const long long* __restrict__ node_ptr, const long long* __restrict__ edge_ptr,
                                                            const long long* __restrict__ label, long long* __restrict__ out_ids,
                                                            double* __restrict__ out_f, int* __restrict__ out_edges,
                                                            long long* __restrict__ pair_of_node, long long* __restrict__ pair_of_edge) {
    const int lane = (int)(threadIdx.x & 63);
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwave = (long long)gridDim.x * (blockDim.x >> 6);
    for (long long i = wave; i < n_pairs; i += nwave) {
        const long long no = node_ptr[i], eo = edge_ptr[i];
        const int n = (int)(node_ptr[i + 1] - no), m = (int)(edge_ptr[i + 1] - eo);
        const long long so = node_offs[i], se = edge_offs[i];
        for (int k = lane; k < n; k += 64) {
            const int x = ids[so + k];
            out_ids[no + k] = label ? label[x] : (long long)x;
            out_f[no + k] = f[so + k];
            if (pair_of_node) pair_of_node[no + k] = i;
        }
        const int2* src = reinterpret_cast<const int2*>(edges) + se;
        int2* dst = reinterpret_cast<int2*>(out_edges) + eo;
        for (int k = lane; k < m; k += 64) {
            dst[k] = src[k];
            if (pair_of_edge) pair_of_edge[eo + k] = i;
        }
    }
}
}  // namespace

// ---- the per-pair sizes of a chunk from its headers (tlc_vicinity_sizes): n as counted, m = directed entries / 2 -------------------
namespace {
__global__ void copy_sizes_kernel(int n_pairs, const int* __restrict__ hdr_n, const int* __restrict__ hdr_m2, int* __restrict__ out_n,
                                  int* __restrict__ out_m) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    // a vicinity that does not fit the packed ids (TLC_ST_TOO_LARGE) has hdr_n = 0 and its negated size in out_n already (x_zero_row /
    // the COUNT kernels wrote it there): that marker stays, as the header documents -- it is not "no vicinity"
    if (out_n && !(out_n[i] < 0 && hdr_n[i] == 0)) out_n[i] = hdr_n[i];
    if (out_m) out_m[i] = hdr_n[i] > 0 ? (hdr_m2[i] >> 1) : 0;
}
}  // namespace
int tlc_launch_copy_sizes(int n_pairs, const int* hdr_n, const int* hdr_m2, int* out_n, int* out_m, void* stream) {
    if (n_pairs <= 0) return TLC_OK;
    hipLaunchKernelGGL(copy_sizes_kernel, dim3((n_pairs + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_pairs, hdr_n, hdr_m2, out_n, out_m);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// ---- tlc_pack_offsets: the packed batch's offsets from the per-pair counts of tlc_vicinity_filtration, ONE launch ----------------
// node_ptr / edge_ptr = exclusive prefix sums of (m > 0 ? n : 0) and max(m, 0) -- a vicinity without an edge is left out of the packed
// batch, the reference returns (None, None) for it (data_utils_LP.py:117-118) -- and totals = {min n, min m, sum n, sum m}: a negative
// count says a vicinity did not fit the caller's capacity.  One workgroup: a contiguous slice of the pairs per thread, block scan of the
// slice sums, second walk writes.  (It replaces a dozen torch launches on a 0.4 ms call.)
namespace {
__global__ __launch_bounds__(1024) void pack_offsets_kernel(long long n_pairs, const int* __restrict__ n, const int* __restrict__ m,
                                                           long long* __restrict__ node_ptr, long long* __restrict__ edge_ptr,
                                                           long long* __restrict__ totals) {
    __shared__ long long s_n[1024], s_m[1024];
    __shared__ int s_mn[1024], s_mm[1024];
    const int tid = (int)threadIdx.x;
    const long long per = (n_pairs + 1023) / 1024, lo = tid * per, hi = lo + per < n_pairs ? lo + per : n_pairs;
    long long an = 0, am = 0;
    int mn = 0x7fffffff, mm = 0x7fffffff;
    for (long long i = lo; i < hi; ++i) {
        const int ni = n[i], mi = m[i];
        an += mi > 0 ? (ni > 0 ? ni : 0) : 0;
        am += mi > 0 ? mi : 0;
        mn = ni < mn ? ni : mn; mm = mi < mm ? mi : mm;
    }
    s_n[tid] = an; s_m[tid] = am; s_mn[tid] = mn; s_mm[tid] = mm;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1) {                               // inclusive scan / min reduction over the 1 024 slice sums
        const long long vn = tid >= d ? s_n[tid - d] : 0, vm = tid >= d ? s_m[tid - d] : 0;
        const int v1 = tid >= d ? s_mn[tid - d] : 0x7fffffff, v2 = tid >= d ? s_mm[tid - d] : 0x7fffffff;
        __syncthreads();
        s_n[tid] += vn; s_m[tid] += vm;
        s_mn[tid] = v1 < s_mn[tid] ? v1 : s_mn[tid]; s_mm[tid] = v2 < s_mm[tid] ? v2 : s_mm[tid];
        __syncthreads();
    }
    long long bn = s_n[tid] - an, bm = s_m[tid] - am;
    for (long long i = lo; i < hi; ++i) {
        node_ptr[i] = bn; edge_ptr[i] = bm;
        const int ni = n[i], mi = m[i];
        bn += mi > 0 ? (ni > 0 ? ni : 0) : 0;
        bm += mi > 0 ? mi : 0;
    }
    if (tid == 1023) {
        node_ptr[n_pairs] = s_n[1023]; edge_ptr[n_pairs] = s_m[1023];
        totals[0] = s_mn[1023]; totals[1] = s_mm[1023]; totals[2] = s_n[1023]; totals[3] = s_m[1023];
    }
}
}  // namespace

extern "C" int tlc_pack_offsets(int64_t n_pairs, const int32_t* d_n, const int32_t* d_m, int64_t* d_node_ptr, int64_t* d_edge_ptr,
                                int64_t* d_totals, void* stream) {
    TLC_REQUIRE(n_pairs >= 0, "n_pairs < 0");
    TLC_REQUIRE(d_node_ptr && d_edge_ptr && d_totals && (n_pairs == 0 || (d_n && d_m)), "null pointer");
    hipLaunchKernelGGL(pack_offsets_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, (long long)n_pairs, d_n, d_m,
                       (long long*)d_node_ptr, (long long*)d_edge_ptr, (long long*)d_totals);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

extern "C" int tlc_pack_vicinities(int64_t n_pairs, const int64_t* d_node_offs, const int32_t* d_ids, const double* d_f,
                                   const int64_t* d_edge_offs, const int32_t* d_edges, const int64_t* d_node_ptr,
                                   const int64_t* d_edge_ptr, const int64_t* d_label, int64_t* d_out_ids, double* d_out_f,
                                   int32_t* d_out_edges, int64_t* d_pair_of_node, int64_t* d_pair_of_edge, void* stream) {
    TLC_REQUIRE(n_pairs >= 0, "n_pairs < 0");
    if (n_pairs == 0) return TLC_OK;
    TLC_REQUIRE(d_node_offs && d_edge_offs && d_node_ptr && d_edge_ptr, "null offsets");
    long long blocks = (n_pairs + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(pack_vicinities_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long long)n_pairs,
                       (const long long*)d_node_offs, d_ids, d_f, (const long long*)d_edge_offs, d_edges, (const long long*)d_node_ptr,
                       (const long long*)d_edge_ptr, (const long long*)d_label, (long long*)d_out_ids, d_out_f, d_out_edges,
                       (long long*)d_pair_of_node, (long long*)d_pair_of_edge);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// ---- tlc_stack_batch: the packed batch -> the model's operands (x, edge_index), ONE launch ------------------------------------------
// What the PDGNN fork's caller does per vicinity (gcn_LP_GIN.py:43-64: edge_index of the vicinity + its self loops, the filtration as
// a float32 column) for the whole block-diagonal batch: global node ids = local id + node_ptr[owner], the n self loops LAST
// (train_Teacher_Model.py:43-44).  One wavefront per vicinity.  (The torch formulation -- long(), gather, add, arange, stack, cat,
// float() -- is eight launches: 0.08 ms of a 0.5 ms forward on 4 096 vicinities.)
namespace {
__global__ __launch_bounds__(256) void stack_batch_kernel(long long n_graphs, const long long* __restrict__ node_ptr, const long long* __restrict__ edge_ptr,
                                                        const int* __restrict__ edges, const double* __restrict__ f, long long tot_n, long long tot_m,
                                                        long long* __restrict__ ei, float* __restrict__ x) {
    const int lane = (int)(threadIdx.x & 63);
    const long long wave = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), nwave = (long long)gridDim.x * (blockDim.x >> 6);
    const long long width = tot_m + tot_n;
    for (long long i = wave; i < n_graphs; i += nwave) {
        const long long no = node_ptr[i], eo = edge_ptr[i];
        const int n = (int)(node_ptr[i + 1] - no), m = (int)(edge_ptr[i + 1] - eo);
        const int2* src = reinterpret_cast<const int2*>(edges) + eo;
        for (int k = lane; k < m; k += 64) {
            const int2 e = src[k];
            ei[eo + k] = no + e.x;
            ei[width + eo + k] = no + e.y;
        }
        for (int k = lane; k < n; k += 64) {
            ei[tot_m + no + k] = no + k;
            ei[width + tot_m + no + k] = no + k;
            if (x) x[no + k] = (float)f[no + k];
        }
    }
}
}  // namespace
extern "C" int tlc_stack_batch(int64_t n_graphs, const int64_t* d_node_ptr, const int64_t* d_edge_ptr, const int32_t* d_edges, const double* d_f,
                               int64_t tot_n, int64_t tot_m, int64_t* d_edge_index, float* d_x, void* stream) {
    TLC_REQUIRE(n_graphs >= 0 && tot_n >= 0 && tot_m >= 0, "bad sizes");
    if (n_graphs == 0 || tot_n + tot_m == 0) return TLC_OK;
    TLC_REQUIRE(d_node_ptr && d_edge_ptr && d_edge_index && (tot_m == 0 || d_edges) && (!d_x || d_f), "null pointer");
    long long blocks = (n_graphs + 3) / 4;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(stack_batch_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long long)n_graphs, (const long long*)d_node_ptr,
                       (const long long*)d_edge_ptr, d_edges, d_f, (long long)tot_n, (long long)tot_m, (long long*)d_edge_index, d_x);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// gat_forward.hip -- M4..M6: the PDGNN layer and edge head, forward only.
//
//   tlc_gat_layer_fwd   GATConv(heads=1, concat=False, new_node_feat, use_edge_attn) on a block-diagonal batch
//                       (Knowledge_Distillation/gat_conv.py:113-216; gather/scatter: message_passing.py:124-183,275-293)
//   tlc_edge_head_fwd   lin6(prelu(lin5([x_src || x_dst])))  (Knowledge_Distillation/Teacher_model.py:54-59)
//
// The reference evaluates lin_ij on the concatenation [x_i || x_j] for every edge (gat_conv.py:193-194): an
// [E, 2C] x [2C, C] product.  Because lin_ij has no bias, W_ij [x_i || x_j] = W_ij[:, :C] x_i + W_ij[:, C:] x_j, so the
// two halves are projected ONCE PER NODE (kernel 1, together with lin_l and the attention logit) and the per-edge work
// collapses to leaky_relu(P_i + Q_j) * softmax weight.  The three scatters (sum, min, max at the target,
// gat_conv.py:216) become one pass over a CSR row per target: every output row has one owner, no atomics.
#include <algorithm>
#include <mutex>
#include <type_traits>

#include "tlc_common.h"
#include "gat_internal.h"

namespace {

// kernel 1: per node  x_l = Wl x,  alpha = x_l . att,  P = Wij[:, :C] x_l,  Q = Wij[:, C:] x_l
// work layout per node: [x_l (C) | P (C) | Q (C) | alpha (1)]   (stride 3C+1)
template <int C>
__global__ __launch_bounds__(256) void gat_node_kernel(int n, const float* __restrict__ X, int c_in, const float* __restrict__ Wl,
                                                       const float* __restrict__ att, const float* __restrict__ Wij,
                                                       float* __restrict__ work) {
    extern __shared__ float sm[];
    float* sWl = sm;                       // [C][c_in]
    float* sWij = sm + C * c_in;           // [C][2C]
    float* sXl = sWij + C * 2 * C;         // [256/C nodes][C]
    // transposed in LDS: the C lanes of a node read C consecutive words per k (row-major [c][k] put all of them on one bank)
    for (int t = threadIdx.x; t < C * c_in; t += 256) { const int c = t / c_in, k = t - c * c_in; sWl[k * C + c] = Wl[t]; }
    for (int t = threadIdx.x; t < C * 2 * C; t += 256) { const int c = t / (2 * C), k = t - c * 2 * C; sWij[k * C + c] = Wij[t]; }
    __syncthreads();
    constexpr int NPB = 256 / C;           // nodes per block
    const int ln = threadIdx.x / C, c = threadIdx.x % C;
    for (int base = blockIdx.x * NPB; base < n; base += gridDim.x * NPB) {
        const int i = base + ln;
        float xl = 0.0f;
        if (i < n) {
            const float* xr = X + (size_t)i * c_in;
            for (int k = 0; k < c_in; ++k) xl += sWl[k * C + c] * xr[k];
        }
        sXl[ln * C + c] = xl;
        // alpha = (x_l * att_l).sum(-1)  (gat_conv.py:135): reduce over the C lanes of this node
        float a = xl * att[c];
#pragma unroll
        for (int o = C / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        __syncthreads();
        if (i < n) {
            const float* xs = sXl + ln * C;
            float p = 0.0f, q = 0.0f;
#pragma unroll 8
            for (int k = 0; k < C; ++k) {
                p += sWij[k * C + c] * xs[k];          // target half:  x_i
                q += sWij[(C + k) * C + c] * xs[k];    // source half:  x_j
            }
            float* w = work + (size_t)i * (3 * C + 1);
            w[c] = xl;
            w[C + c] = p;
            w[2 * C + c] = q;
            if (c == 0) w[3 * C] = a;
        }
        __syncthreads();
    }
}

// kernel 2: one group of C lanes per target node
// NPG nodes per lane group: the kernel is a chain of three dependent round trips per node (row bounds -> source ids -> alpha / Q
// of the sources) and its duration was (wavefronts / resident wavefronts) x that chain: 336 us for a million nodes.  With the loads
// of NPG nodes issued together a wavefront's lifetime serves NPG times the nodes.
// RANK1 (c_in == 1, the first PDGNN layer: the node feature is the filtration value alone): x_l = f_i * Wl[:,0], so the whole row
// [P | Q | alpha] of node i is f_i * v with ONE vector v = Wl[:,0]^T [Wij_t^T | Wij_s^T | att] -- neither x_l nor the rows are
// materialised (no 70 us narrow product, no 120 us GEMM, 4-byte instead of 132-byte gathers): pqa then points at f (stride S = 1)
// and `vec` at v.
template <int C, int NPG, bool RANK1>
__global__ __launch_bounds__(256) void gat_aggregate_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ src,
                                                            const float* __restrict__ pqa, int S, const float* __restrict__ bias,
                                                            float prelu_slope, float* __restrict__ out, const float* __restrict__ vec) {
    // pqa: per node [P (C) | Q (C) | alpha (1)] with row stride S
    const int grp = (blockIdx.x * 256 + threadIdx.x) / C, c = threadIdx.x % C;
    float vP = 0.0f, vQ = 0.0f, vA = 0.0f;
    if (RANK1) { vP = vec[c]; vQ = vec[C + c]; vA = vec[2 * C]; }
    int b[NPG], e[NPG];
    float ai[NPG], pi[NPG];
#pragma unroll
    for (int u = 0; u < NPG; ++u) {
        const int i = grp * NPG + u;
        b[u] = 0; e[u] = 0; ai[u] = 0.0f; pi[u] = 0.0f;
        if (i < n) {
            b[u] = rowptr[i]; e[u] = rowptr[i + 1];
            if (RANK1) { const float fi = pqa[i]; ai[u] = fi * vA; pi[u] = fi * vP; }
            else { ai[u] = pqa[(size_t)i * S + 2 * C]; pi[u] = pqa[(size_t)i * S + c]; }
        }
    }
    // the usual case (molecules, vicinity graphs: a handful of neighbours + the self loop): all source ids, then all alphas and
    // Q values, of all NPG nodes are requested together -- one dependent round trip each instead of one per edge and pass
    int sj[NPG][8];
    float tj[NPG][8], qj[NPG][8];
#pragma unroll
    for (int u = 0; u < NPG; ++u)
#pragma unroll
        for (int q = 0; q < 8; ++q) sj[u][q] = (e[u] - b[u] <= 8 && b[u] + q < e[u]) ? src[b[u] + q] : -1;
#pragma unroll
    for (int u = 0; u < NPG; ++u)
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            tj[u][q] = 0.0f; qj[u][q] = 0.0f;
            if (sj[u][q] >= 0) {
                if (RANK1) { const float fj = pqa[sj[u][q]]; tj[u][q] = fj * vA; qj[u][q] = fj * vQ; }
                else {
                    const float* wj = pqa + (size_t)sj[u][q] * S;
                    tj[u][q] = wj[2 * C];
                    qj[u][q] = wj[C + c];
                }
            }
        }
#pragma unroll
    for (int u = 0; u < NPG; ++u) {
        const int i = grp * NPG + u;
        if (i >= n) continue;
        // softmax over the incoming edges (torch_geometric.utils.softmax: subtract the segment max)
        float mx = -INFINITY;
        float den = 0.0f, sum = 0.0f, mn = INFINITY, mxv = -INFINITY;
        if (e[u] - b[u] <= 8) {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (sj[u][q] >= 0) {
                    float t = tj[u][q] + ai[u];                   // alpha_j + alpha_i  (gat_conv.py:184)
                    t = t > 0.0f ? t : 0.2f * t;                  // leaky_relu(negative_slope=0.2) (:185)
                    tj[u][q] = t;
                    mx = t > mx ? t : mx;
                }
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (sj[u][q] >= 0) {
                    const float ex = expf(tj[u][q] - mx);
                    den += ex;
                    float m = pi[u] + qj[u][q];                   // lin_ij([x_i || x_j]) (:193-194)
                    m = m > 0.0f ? m : 0.2f * m;                  // leaky_relu (:195)
                    m *= ex;                                      // * alpha (:198-200), normalised below
                    sum += m;
                    mn = m < mn ? m : mn;
                    mxv = m > mxv ? m : mxv;
                }
            }
        } else {
            for (int j = b[u]; j < e[u]; ++j) {
                float t = (RANK1 ? pqa[src[j]] * vA : pqa[(size_t)src[j] * S + 2 * C]) + ai[u];
                t = t > 0.0f ? t : 0.2f * t;
                mx = t > mx ? t : mx;
            }
            for (int j = b[u]; j < e[u]; ++j) {
                const float* wj = pqa + (size_t)src[j] * S;
                const float fj = RANK1 ? pqa[src[j]] : 0.0f;
                float t = (RANK1 ? fj * vA : wj[2 * C]) + ai[u];
                t = t > 0.0f ? t : 0.2f * t;
                const float ex = expf(t - mx);
                den += ex;
                float m = pi[u] + (RANK1 ? fj * vQ : wj[C + c]);
                m = m > 0.0f ? m : 0.2f * m;
                m *= ex;
                sum += m;
                mn = m < mn ? m : mn;
                mxv = m > mxv ? m : mxv;
            }
        }
        float o_sum = 0.0f, o_mm = 0.0f;                          // empty segment: scatter leaves zeros
        if (e[u] > b[u]) {
            const float inv = 1.0f / (den + 1e-16f);
            o_sum = sum * inv;
            o_mm = mn * inv + mxv * inv;                          // scatter min + scatter max (:216)
        }
        o_sum += bias[c];                                         // mean over the single head, + bias (:166-172)
        o_mm += bias[C + c];
        if (prelu_slope >= 0.0f) {                                // F.prelu(x, 0.1) between layers (Teacher_model.py:219-227)
            o_sum = o_sum > 0.0f ? o_sum : prelu_slope * o_sum;
            o_mm = o_mm > 0.0f ? o_mm : prelu_slope * o_mm;
        }
        out[(size_t)i * 2 * C + c] = o_sum;
        out[(size_t)i * 2 * C + C + c] = o_mm;
    }
}

// edge head: one thread per edge; W5 / W6 staged in LDS
__global__ __launch_bounds__(256) void edge_head_kernel(long long n_edges, const int* __restrict__ src, const int* __restrict__ dst,
                                                        const float* __restrict__ X, int c, const float* __restrict__ W5,
                                                        const float* __restrict__ b5, int hidden, float slope,
                                                        const float* __restrict__ W6, const float* __restrict__ b6,
                                                        float* __restrict__ pd) {
    extern __shared__ float sm[];
    float* sW5 = sm;                       // [hidden][2c]
    float* sW6 = sm + hidden * 2 * c;      // [2][hidden]
    for (int t = threadIdx.x; t < hidden * 2 * c; t += 256) sW5[t] = W5[t];
    for (int t = threadIdx.x; t < 2 * hidden; t += 256) sW6[t] = W6[t];
    __syncthreads();
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_edges) return;
    const float* xs = X + (size_t)src[e] * c;
    const float* xd = X + (size_t)dst[e] * c;
    float o0 = b6[0], o1 = b6[1];
    for (int h = 0; h < hidden; ++h) {
        const float* wr = sW5 + h * 2 * c;
        float v = b5[h];
        for (int k = 0; k < c; ++k) v += wr[k] * xs[k];
        for (int k = 0; k < c; ++k) v += wr[c + k] * xd[k];
        v = v > 0.0f ? v : slope * v;                          // F.prelu(x, 0.1) (Teacher_model.py:57)
        o0 += sW6[h] * v;
        o1 += sW6[hidden + h] * v;
    }
    pd[2 * e] = o0;
    pd[2 * e + 1] = o1;
}

// The edge head per NODE first: lin5([x_s || x_t]) = W5[:, :c] x_s + W5[:, c:] x_t, so U = X @ W5[:, :c]^T and V = X @ W5[:, c:]^T
// are computed once per node on the MFMA GEMM (a molecule batch has as many nodes as edges, but every node is the endpoint of
// several edges, and a 32 x 64 product per EDGE on the vector ALU was 0.34 ms of a 1 ms PDGNN forward); the per-edge kernel
// then gathers two rows of 2*hidden floats, adds, applies b5 / PReLU and the 2 x hidden output layer.
__global__ void edge_head_pack_kernel(int c, int hidden, const float* __restrict__ W5, float* __restrict__ Bt) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;               // Bt[k][j], k < c, j < 2*hidden
    if (t >= c * 2 * hidden) return;
    const int k = t / (2 * hidden), j = t - k * 2 * hidden;
    Bt[t] = j < hidden ? W5[(size_t)j * 2 * c + k] : W5[(size_t)(j - hidden) * 2 * c + c + k];
}
// IT: int (tlc_edge_head_fwd) or long long (tlc_pdgnn_forward reads the caller's int64 edge_index as it lies); pd64 (may be null): the
// points once more as float64, what tlc_pi_raster takes -- a conversion launch of their own cost 12.6 + 10.5 us of a 1.09 ms forward
template <typename IT>
__global__ __launch_bounds__(256) void edge_head_gather_kernel(long long n_edges, const IT* __restrict__ src, const IT* __restrict__ dst,
                                                               const float* __restrict__ UV, int hidden, const float* __restrict__ b5,
                                                               float slope, const float* __restrict__ W6, const float* __restrict__ b6,
                                                               float* __restrict__ pd, double* __restrict__ pd64) {
    extern __shared__ float sm[];                                      // [b5 (hidden) | W6 (2*hidden)]
    for (int t = threadIdx.x; t < hidden; t += 256) sm[t] = b5[t];
    for (int t = threadIdx.x; t < 2 * hidden; t += 256) sm[hidden + t] = W6[t];
    __syncthreads();
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= n_edges) return;
    const float4* us = reinterpret_cast<const float4*>(UV + (size_t)src[e] * 2 * hidden);
    const float4* vd = reinterpret_cast<const float4*>(UV + (size_t)dst[e] * 2 * hidden + hidden);
    float o0 = b6[0], o1 = b6[1];
    for (int h4 = 0; h4 < hidden / 4; ++h4) {
        const float4 a = us[h4], b = vd[h4];
        float v[4] = {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = 4 * h4 + q;
            float t = v[q] + sm[h];
            t = t > 0.0f ? t : slope * t;                              // F.prelu(x, 0.1) (Teacher_model.py:57)
            o0 += sm[hidden + h] * t;
            o1 += sm[2 * hidden + h] * t;
        }
    }
    pd[2 * e] = o0;
    pd[2 * e + 1] = o1;
    if (pd64) { pd64[2 * e] = (double)o0; pd64[2 * e + 1] = (double)o1; }
}

// The same with hidden / 4 lanes per edge (hidden = 8, 16, 32, 64): a lane loads ONE float4 of U[src] and of V[dst] -- the edge's two
// rows are two coalesced 4 * hidden-byte reads instead of 2 * hidden / 4 sixteen-byte reads of one lane (a wavefront touched 64 x 2
// separate rows per instruction: 90 - 104 us for the million edges of 41 127 molecules) --, its four hidden units, and the two output
// sums are folded over the lanes of the edge.
template <typename IT, int LPE>
__global__ __launch_bounds__(256) void edge_head_gather_lanes_kernel(long long n_edges, const IT* __restrict__ src, const IT* __restrict__ dst,
                                                                     const float* __restrict__ UV, const float* __restrict__ b5, float slope,
                                                                     const float* __restrict__ W6, const float* __restrict__ b6,
                                                                     float* __restrict__ pd, double* __restrict__ pd64) {
    constexpr int hidden = 4 * LPE;
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long e = t / LPE;
    const int q = (int)(t % LPE);
    if (e >= n_edges) return;                                          // (256 is a multiple of LPE: an edge's lanes leave together)
    const float4 a = reinterpret_cast<const float4*>(UV + (size_t)src[e] * 2 * hidden)[q];
    const float4 b = reinterpret_cast<const float4*>(UV + (size_t)dst[e] * 2 * hidden + hidden)[q];
    const float4 bb = reinterpret_cast<const float4*>(b5)[q];
    const float4 w0 = reinterpret_cast<const float4*>(W6)[q], w1 = reinterpret_cast<const float4*>(W6 + hidden)[q];
    float v[4] = {a.x + b.x + bb.x, a.y + b.y + bb.y, a.z + b.z + bb.z, a.w + b.w + bb.w};
    const float ww0[4] = {w0.x, w0.y, w0.z, w0.w}, ww1[4] = {w1.x, w1.y, w1.z, w1.w};
    float o0 = 0.0f, o1 = 0.0f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float x = v[j] > 0.0f ? v[j] : slope * v[j];             // F.prelu(x, 0.1) (Teacher_model.py:57)
        o0 += ww0[j] * x;
        o1 += ww1[j] * x;
    }
#pragma unroll
    for (int o = LPE / 2; o; o >>= 1) { o0 += __shfl_xor(o0, o); o1 += __shfl_xor(o1, o); }
    if (q == 0) {
        o0 += b6[0]; o1 += b6[1];
        pd[2 * e] = o0;
        pd[2 * e + 1] = o1;
        if (pd64) { pd64[2 * e] = (double)o0; pd64[2 * e + 1] = (double)o1; }
    }
}

// The head in ONE kernel for the PDGNN sizes (c = 32 node channels, hidden = 32; round 5): the 2c x hidden layer on the f32 MFMA with
// the EDGES as rows -- a 16-edge tile's A operand is gathered straight from X (lane (edge r, group g) reads the sixteen contiguous
// floats g & 1 of x[src] for g < 2, of x[dst] for g >= 2: k = 16 g + s at step s, W5 in registers in the same permuted order, as
// gemm_skinny_f32_kernel does), 32 MFMAs per tile, then bias, PReLU, the 2 x hidden output layer and a fold over the sixteen lanes that
// hold an edge's hidden units.  Instead of: pack W5, per-node projections U | V [n, 64] on the MFMA (131 MB read, 263 MB written for a
// million nodes), gather of two 128-byte rows of U | V per edge -- 92 + 96 us and two launches for the 1.04 M edges of 41 127 molecules.
typedef float eh_f32x4 __attribute__((ext_vector_type(4)));
template <typename IT>
__global__ __launch_bounds__(256) void edge_head_mfma_kernel(long long n_edges, const IT* __restrict__ src, const IT* __restrict__ dst,
                                                             const float* __restrict__ X, const float* __restrict__ W5, const float* __restrict__ b5,
                                                             float slope, const float* __restrict__ W6, const float* __restrict__ b6,
                                                             float* __restrict__ pd, double* __restrict__ pd64) {
    constexpr int CX = 32, H = 32, K2 = 64, KQ = 16;
    const int lane = threadIdx.x & 63, l16 = lane & 15, g = lane >> 4;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    float b[KQ][2];
#pragma unroll
    for (int sidx = 0; sidx < KQ; ++sidx)
#pragma unroll
        for (int t = 0; t < 2; ++t) b[sidx][t] = W5[(size_t)(t * 16 + l16) * K2 + g * KQ + sidx];       // W5[h][k], k = 16 g + s
    float bb[2], w0[2], w1[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) { bb[t] = b5[t * 16 + l16]; w0[t] = W6[t * 16 + l16]; w1[t] = W6[H + t * 16 + l16]; }
    const float c0 = b6[0], c1 = b6[1];
    const long long n_tiles = (n_edges + 15) >> 4;
    auto node_of = [&](long long tile) -> long long {
        long long e = tile * 16 + l16;
        if (e >= n_edges) e = n_edges - 1;                              // (edges past the end are computed and not stored)
        return (long long)(g < 2 ? src[e] : dst[e]);
    };
    auto load_tile = [&](long long node, eh_f32x4 (&a)[4]) {
        const eh_f32x4* p = reinterpret_cast<const eh_f32x4*>(X + (size_t)node * CX + (g & 1) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) a[q] = p[q];
    };
    eh_f32x4 a_cur[4], a_nxt[4];
    long long tile = wave, node_nxt = 0;
    if (tile < n_tiles) load_tile(node_of(tile), a_cur);
    if (tile + n_waves < n_tiles) node_nxt = node_of(tile + n_waves);
    for (; tile < n_tiles; tile += n_waves) {
        const long long nt = tile + n_waves;
        if (nt < n_tiles) load_tile(node_nxt, a_nxt);
        if (nt + n_waves < n_tiles) node_nxt = node_of(nt + n_waves);
        eh_f32x4 acc[2] = {(eh_f32x4){0.f, 0.f, 0.f, 0.f}, (eh_f32x4){0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int sidx = 0; sidx < KQ; ++sidx)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[sidx >> 2][sidx & 3], b[sidx][t], acc[t], 0, 0, 0);
        // C / D layout: column (hidden unit of tile t) = lane & 15, row (edge of the tile) = 4 * (lane >> 4) + reg
        float o0[4], o1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0[r] = 0.f; o1[r] = 0.f;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                float v = acc[t][r] + bb[t];
                v = v > 0.0f ? v : slope * v;                          // F.prelu(x, 0.1) (Teacher_model.py:57)
                o0[r] += w0[t] * v;
                o1[r] += w1[t] * v;
            }
        }
#pragma unroll
        for (int o = 8; o; o >>= 1)
#pragma unroll
            for (int r = 0; r < 4; ++r) { o0[r] += __shfl_xor(o0[r], o); o1[r] += __shfl_xor(o1[r], o); }
        // lane r of the sixteen writes edge 4 g + r
        if (l16 < 4) {
            const long long e = tile * 16 + 4 * g + l16;
            const float y0 = (l16 == 0 ? o0[0] : l16 == 1 ? o0[1] : l16 == 2 ? o0[2] : o0[3]) + c0;
            const float y1 = (l16 == 0 ? o1[0] : l16 == 1 ? o1[1] : l16 == 2 ? o1[2] : o1[3]) + c1;
            if (e < n_edges) {
                pd[2 * e] = y0; pd[2 * e + 1] = y1;
                if (pd64) { pd64[2 * e] = (double)y0; pd64[2 * e + 1] = (double)y1; }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) a_cur[q] = a_nxt[q];
    }
}

// weights of one layer packed for the MFMA GEMM path: Bt1[k][c] = Wl[c][k]  (c_in x C);  Bt2[k][j] (C x (2C+4), rows a multiple
// of 16 bytes so that the GEMM takes its vector path): j < C: Wij[j][k] (target half), C <= j < 2C: Wij[j-C][C+k] (source
// half), j = 2C: att[k], then zeros
__global__ void gat_pack_kernel(int C, int c_in, const float* __restrict__ Wl, const float* __restrict__ att,
                                const float* __restrict__ Wij, float* __restrict__ Bt1, float* __restrict__ Bt2) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < c_in * C) { const int k = t / C, c = t - k * C; Bt1[t] = Wl[(size_t)c * c_in + k]; }
    const int N2 = 2 * C + 4;
    if (t < C * N2) {
        const int k = t / N2, j = t - k * N2;
        Bt2[t] = j < C ? Wij[(size_t)j * 2 * C + k] : (j < 2 * C ? Wij[(size_t)(j - C) * 2 * C + C + k] : (j == 2 * C ? att[k] : 0.0f));
    }
}
// c_in == 1: v[j] = sum_k Wl[k] * B2[k][j] with B2 = [Wij_t^T | Wij_s^T | att]  (j < 2C + 1)
__global__ void gat_rank1_vec_kernel(int C, const float* __restrict__ Wl, const float* __restrict__ att, const float* __restrict__ Wij,
                                     float* __restrict__ vec) {
    const int j = threadIdx.x;
    if (j > 2 * C) return;
    float a = 0.0f;
    for (int k = 0; k < C; ++k) {
        const float bkj = j < C ? Wij[(size_t)j * 2 * C + k] : (j < 2 * C ? Wij[(size_t)(j - C) * 2 * C + C + k] : att[k]);
        a += Wl[k] * bkj;                                    // Wl is [C, 1]
    }
    vec[j] = a;
}
// x_l = X Wl^T for a narrow input (1 < c_in < 16): one thread per (node, channel)
__global__ void gat_xl_narrow_kernel(int n, int C, int c_in, const float* __restrict__ X, const float* __restrict__ Wl,
                                     float* __restrict__ XL) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * C) return;
    const int i = (int)(t / C), c = (int)(t - (long long)i * C);
    float a = 0.0f;
    for (int k = 0; k < c_in; ++k) a += Wl[(size_t)c * c_in + k] * X[(size_t)i * c_in + k];
    XL[t] = a;
}

template <int C>
int launch_gat(int n, const int* rowptr, const int* src, const float* X, int c_in, const float* Wl, const float* att,
               const float* Wij, const float* bias, float slope, float* work, float* out, hipStream_t s) {
    const unsigned agrid = (unsigned)(((size_t)n * C + 255) / 256);
    constexpr int NPG = 2;                                   // nodes per lane group in the aggregation (4: 321 us, 1: 336 us, 2: 243 us)
    const unsigned agrid2 = (unsigned)((((size_t)n + NPG - 1) / NPG * C + 255) / 256);
    if (c_in == 1) {
        // the first PDGNN layer: rows are f_i * v (see gat_aggregate_kernel, RANK1)
        float* vec = work;                                   // [2C + 1]
        hipLaunchKernelGGL(gat_rank1_vec_kernel, dim3(1), dim3(256), 0, s, C, Wl, att, Wij, vec);
        hipLaunchKernelGGL((gat_aggregate_kernel<C, NPG, true>), dim3(agrid2), dim3(256), 0, s, n, rowptr, src, X, 1, bias, slope, out,
                           (const float*)vec);
        TLC_HIP_CHECK(hipGetLastError());
        return TLC_OK;
    }
    if (2 * C + 4 <= 128 && C % 4 == 0) {
        // x_l = X Wl^T and [P | Q | alpha] = x_l [Wij_t^T | Wij_s^T | att] as two products on the f32 MFMA (per node on the
        // vector ALU this was 95 us of a layer; the products are 2 x ~15 us)
        float* XL = work;
        float* PQA = work + (size_t)n * C;
        float* Bt1 = work + (size_t)n * (3 * C + 4);
        float* Bt2 = Bt1 + (size_t)c_in * C;
        const int N2 = 2 * C + 4;
        const int np = std::max(c_in * C, C * N2);
        hipLaunchKernelGGL(gat_pack_kernel, dim3((unsigned)((np + 255) / 256)), dim3(256), 0, s, C, c_in, Wl, att, Wij, Bt1, Bt2);
        TLC_HIP_CHECK(hipGetLastError());
        int rc = TLC_OK;
        if (c_in >= 16 && c_in % 4 == 0 && c_in <= 64 && (size_t)n * C >= (size_t)c_in * N2) {
            // [P | Q | alpha] = X (Wl^T B2): the two weight matrices are multiplied first (c_in x N2, in the space x_l would
            // have taken), so the node features make ONE pass through the MFMA and x_l [n, C] is never written or read back
            float* Wf = XL;
            rc = tlc_gemm_f32(c_in, N2, C, Bt1, Bt2, nullptr, 0, Wf, s);
            if (rc != TLC_OK) return rc;
            rc = tlc_gemm_f32(n, N2, c_in, X, Wf, nullptr, 0, PQA, s);
            if (rc != TLC_OK) return rc;
        } else {
            if (c_in >= 16 && c_in % 4 == 0) rc = tlc_gemm_f32(n, C, c_in, X, Bt1, nullptr, 0, XL, s);
            else hipLaunchKernelGGL(gat_xl_narrow_kernel, dim3(agrid), dim3(256), 0, s, n, C, c_in, X, Wl, XL);
            if (rc != TLC_OK) return rc;
            rc = tlc_gemm_f32(n, N2, C, XL, Bt2, nullptr, 0, PQA, s);
            if (rc != TLC_OK) return rc;
        }
        hipLaunchKernelGGL((gat_aggregate_kernel<C, NPG, false>), dim3(agrid2), dim3(256), 0, s, n, rowptr, src, (const float*)PQA, N2, bias,
                           slope, out, (const float*)nullptr);
        TLC_HIP_CHECK(hipGetLastError());
        return TLC_OK;
    }
    const size_t lds = ((size_t)C * c_in + (size_t)C * 2 * C + 256) * sizeof(float);
    if (lds > 64 * 1024) { tlc_set_error("tlc_gat_layer_fwd: c_in too large for the LDS-staged weights"); return TLC_ERR_UNSUPPORTED; }
    constexpr int NPB = 256 / C;
    int grid = (n + NPB - 1) / NPB;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(gat_node_kernel<C>, dim3(grid), dim3(256), lds, s, n, X, c_in, Wl, att, Wij, work);
    hipLaunchKernelGGL((gat_aggregate_kernel<C, NPG, false>), dim3(agrid2), dim3(256), 0, s, n, rowptr, src, (const float*)(work + C), 3 * C + 1, bias,
                       slope, out, (const float*)nullptr);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// ======================================================================================================================
// Round 5: the layer on a BLOCK-DIAGONAL batch in one kernel, tile by tile, with [P | Q | alpha] never leaving the CU.
//
// The two-kernel form writes the node rows [P | Q | alpha] (272 B per node) to HBM and gathers them back per in-edge: for
// ogbg-molhiv's million nodes that is 280 MB out, ~1 GB of 128-byte lines in, and an aggregation whose every node waits for
// three dependent global round trips (row bounds -> source ids -> rows): 183 + 231 us per layer.  In a batch of small graphs
// every in-edge of a node comes from its own graph, so a TILE of consecutive whole graphs (tile_ptr: node offsets, at most
// GT_TM nodes, closed under in-edges -- the caller cuts the batch at positions no edge crosses) is self-contained:
//   phase 0  the tile's row bounds and source ids (as tile-local ids) go to LDS: one coalesced pass;
//   phase 1  [P | Q | alpha] = X_tile (Wl^T [Wij_t^T | Wij_s^T | att]) on the f32 MFMA, 16-row tiles dealt to the wavefronts, A
//            straight from global memory (lane (row, g) reads the sixteen contiguous floats k = 16 g .. of its row), the combined
//            weights in registers for the workgroup's whole life, results into the LDS tile (row stride 2C + 4 words);
//            c_in == 1 (first layer): the row is f_i * v, written by the vector ALU;
//   phase 2  a group of C lanes per node: softmax statistics, sum / min / max of leaky_relu(P_i + Q_j) * a over the in-edges --
//            every operand an LDS read -- bias, PReLU, the 2C-wide output row in two coalesced stores.
// Persistent workgroups (two per CU: 78 KB of LDS each, so that one's MFMA phase runs under the other's aggregation), tiles strided;
// the combined weights sit in LDS in MFMA operand layout for the workgroup's whole life.
// ======================================================================================================================
#define GT_TM 192            /* nodes per tile (twelve 16-row MFMA tiles) */
#define GT_EM 2048           /* in-edge slots of a tile staged in LDS; a tile with more reads its source ids from global memory */
#define GT_THREADS 512
typedef float gt_f32x4 __attribute__((ext_vector_type(4)));

template <int C, int CIN>
__global__ __launch_bounds__(GT_THREADS, 4) void gat_tile_kernel(int n_tiles, const int* __restrict__ tile_ptr, const int* __restrict__ rowptr,
                                                              const int* __restrict__ src, const float* __restrict__ X,
                                                              const float* __restrict__ Wf, const float* __restrict__ vec,
                                                              const float* __restrict__ bias, float prelu_slope, float* __restrict__ out) {
    constexpr int N2 = 2 * C + 4, NT = (N2 + 15) / 16;
    static_assert(CIN == 1 || (CIN % 16 == 0 && CIN <= 64), "c_in: 1 (rank-one rows) or a multiple of 16 up to 64");
    extern __shared__ __attribute__((aligned(16))) float gt_lds[];
    float* const pqa = gt_lds;                                   // [GT_TM][N2]
    int* const rp = (int*)(gt_lds + GT_TM * N2);                 // [GT_TM + 1] row bounds relative to the tile's first in-edge slot
    unsigned short* const ls = (unsigned short*)(rp + GT_TM + 4); // [GT_EM] tile-local source ids
    // [GT_TM] node values (c_in == 1) | the combined weights in MFMA operand layout (c_in >= 16): [t][k / 16][lane][4] -- lane
    // (l16, g) reads the four steps k = 16 q + 4 g .. of column tile t in one ds_read_b128
    float* const wop = reinterpret_cast<float*>(ls + GT_EM);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l16 = lane & 15, g = lane >> 4;
    constexpr int NW = GT_THREADS / 64;
    if constexpr (CIN >= 16) {
        // step s of lane group g consumes k = 16 q + 4 g + (s & 3), q = s >> 2, for A and B alike (a sum over k does not care)
        for (int i = tid; i < NT * (CIN / 16) * 64 * 4; i += GT_THREADS) {
            const int e = i & 3, ln = (i >> 2) & 63, q = (i >> 8) % (CIN / 16), t = (i >> 8) / (CIN / 16);
            const int k = 16 * q + 4 * (ln >> 4) + e, c = 16 * t + (ln & 15);
            wop[i] = c < N2 ? Wf[(size_t)k * N2 + c] : 0.0f;
        }
        __syncthreads();
    }
    // the head of a tile -- its node range and first in-edge slot: two dependent loads -- is fetched one tile ahead
    // (a caller-made cut with an empty tile or one of more rows than the LDS tile holds: the tile is run as ONE row -- every index below
    // stays inside the tile's regions -- and its rows come back as NaN: loud, nothing overrun.  A cut of tlc_gat_tile_cut never has one.
    // Kept to two scalars and one predicate on the stores: a separate path for such a tile cost the kernel 12 registers and spills.)
    int nx_base = 0, nx_tn = 0, nx_eb0 = 0, nx_ne = 0, nx_raw = 0;
    auto head = [&](int t) {
        if (t < n_tiles) {
            nx_base = tile_ptr[t]; nx_raw = tile_ptr[t + 1] - nx_base;
            nx_tn = (nx_raw < 1 || nx_raw > GT_TM) ? 1 : nx_raw;
            nx_eb0 = rowptr[nx_base]; nx_ne = rowptr[nx_base + nx_tn] - nx_eb0;
        }
    };
    head(blockIdx.x);
    // What phase 0 and the first MFMA row tile need from global memory is requested one tile AHEAD, into registers, at the start of
    // the tile before's aggregation (six registers, + the first A operands): a tile's stores drain behind a barrier anyway, and the
    // next tile's loads now travel beside them instead of behind them.
    constexpr int NQ_ = CIN >= 16 ? CIN / 16 : 1;
    int st_rp = 0, st_ls[4] = {0, 0, 0, 0};
    float st_fl = 0.0f;
    gt_f32x4 st_a[NQ_];
    auto load_a0 = [&](int base_, int tn_, int rt, gt_f32x4 (&a)[NQ_]) {
        if constexpr (CIN >= 16) {
            int row = rt * 16 + l16;
            if (row >= tn_) row = tn_ - 1;                        // (rows past the tile: computed, not stored)
            const gt_f32x4* ap = reinterpret_cast<const gt_f32x4*>(X + (size_t)(base_ + row) * CIN + 4 * g);
#pragma unroll
            for (int q = 0; q < NQ_; ++q) a[q] = ap[4 * q];
        }
    };
    // (every load unconditional at a clamped index, the raw value kept: a load inside `if (k < ne)` is waited for inside its own
    // branch, and the six loads of a tile's phase 0 became six round trips one after the other)
    auto fetch = [&]() {                                          // (of the tile whose head is in nx_*)
        st_rp = rowptr[nx_base + (tid <= nx_tn ? tid : nx_tn)];
        if (nx_ne > 0 && nx_ne <= GT_EM) {                        // (uniform)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = tid + q * GT_THREADS;
                st_ls[q] = src[nx_eb0 + (k < nx_ne ? k : nx_ne - 1)];
            }
        }
        if constexpr (CIN == 1) st_fl = X[nx_base + (tid < nx_tn ? tid : nx_tn - 1)];
        else load_a0(nx_base, nx_tn, wave < ((nx_tn + 15) >> 4) ? wave : 0, st_a);
    };
    if ((int)blockIdx.x < n_tiles) fetch();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int base = nx_base, tn = nx_tn, eb0 = nx_eb0, ne = nx_ne;
        const bool bad = nx_raw != tn;                            // (uniform)
        if (bad) {
            const long long left = (long long)tile_ptr[n_tiles] - base;
            const long long rows = nx_raw < 1 ? 0 : ((long long)nx_raw < left ? (long long)nx_raw : left);
            for (long long i = tid; i < rows * 2 * C; i += GT_THREADS) out[(size_t)base * 2 * C + i] = __builtin_nanf("");
        }
        const bool staged = ne <= GT_EM;
        // ---- phase 0: row bounds and tile-local source ids (fetched during the tile before) --------------------------------
        if (tid <= tn) rp[tid] = st_rp - eb0;
        if (staged) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int k = tid + q * GT_THREADS; if (k < ne) ls[k] = (unsigned short)(st_ls[q] - base); }
        }
        if constexpr (CIN == 1) { if (tid < tn) wop[tid] = st_fl; }
        head(tile + (int)gridDim.x);
        __syncthreads();
        // ---- phase 1: the tile's node rows ----------------------------------------------------------------------------------
        if constexpr (CIN == 1) {
            // (rows dealt to the wavefronts, lane j writes columns j and 64 + j: no division per element)
            const float v0 = lane <= 2 * C ? vec[lane] : 0.0f, v1 = 64 + lane <= 2 * C ? vec[64 + lane] : 0.0f;
            const float* const fl = wop;                                // [GT_TM] the tile's node values
            for (int li = wave; li < tn; li += NW) {
                const float fi = fl[li];
                if (lane <= 2 * C) pqa[li * N2 + lane] = fi * v0;
                if (64 + lane <= 2 * C) pqa[li * N2 + 64 + lane] = fi * v1;
            }
        } else {
            // 16-row tiles dealt to the wavefronts; lane (row l16, g) holds A[row][16 q + 4 g ..] for q = 0 .. CIN/16 (one 16-byte load
            // each), the NEXT row tile's loads are in flight during this one's MFMAs; B comes from LDS, one 16-byte read per four MFMAs
            const int nrt = (tn + 15) >> 4;
            constexpr int NQ = CIN / 16;
            auto load_a = [&](int rt, gt_f32x4 (&a)[NQ]) { load_a0(base, tn, rt, a); };
            gt_f32x4 a_cur[NQ], a_nxt[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) a_cur[q] = st_a[q];

            for (int rt = wave; rt < nrt; rt += NW) {
                if (rt + NW < nrt) load_a(rt + NW, a_nxt);
                gt_f32x4 acc[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = (gt_f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    gt_f32x4 bq[NT];
#pragma unroll
                    for (int t = 0; t < NT; ++t) bq[t] = *reinterpret_cast<const gt_f32x4*>(wop + ((t * NQ + q) * 64 + lane) * 4);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int t = 0; t < NT; ++t)
                            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[q][e], bq[t][e], acc[t], 0, 0, 0);
                }
                // C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int c = t * 16 + l16;
                    if (c < N2) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int lr = rt * 16 + 4 * g + r;
                            if (lr < tn) pqa[lr * N2 + c] = acc[t][r];
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < NQ; ++q) a_cur[q] = a_nxt[q];
            }
        }
        __syncthreads();
        fetch();                                                  // the next tile's phase-0 data: in flight during this aggregation
        // ---- phase 2: one group of C / 4 lanes per node, four channels per lane ----------------------------------------------
        // (a lane per channel made all C lanes of a node compute the same attention logits, exponentials and denominator: 27 of
        // its ~35 instructions per in-edge; with four channels per lane that part is shared four ways and every LDS read is 16 bytes)
        {
            constexpr int LPN = C / 4, NG = GT_THREADS / LPN;
            const int grp = tid / LPN, c = 4 * (tid % LPN);
            const gt_f32x4 bs = *reinterpret_cast<const gt_f32x4*>(bias + c), bm = *reinterpret_cast<const gt_f32x4*>(bias + C + c);
            auto nodes = [&](auto staged_c) {
                constexpr bool ST = decltype(staged_c)::value;
                for (int li = grp; li < tn; li += NG) {
                    const int bq = rp[li], eq = rp[li + 1], d = eq - bq;
                    const float* const ri = pqa + __mul24(li, N2);        // (24-bit multiplies: v_mul_lo_u32 is a quarter-rate instruction)
                    const float ai = ri[2 * C];
                    const gt_f32x4 pi = *reinterpret_cast<const gt_f32x4*>(ri + c);
                    float mx = -INFINITY, den = 0.0f;
                    gt_f32x4 sum = {0.f, 0.f, 0.f, 0.f}, mn = {INFINITY, INFINITY, INFINITY, INFINITY}, mxv = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
                    auto slot = [&](int j) -> int { return ST ? (int)ls[j] : src[eb0 + j] - base; };
                    // two walks over the node's in-edges (a molecule's node has three or four): the largest logit, then the weights
                    for (int j = bq; j < eq; ++j) {
                        const float t0 = pqa[__mul24(slot(j), N2) + 2 * C] + ai;      // alpha_j + alpha_i  (gat_conv.py:184)
                        mx = fmaxf(mx, fmaxf(t0, 0.2f * t0));                         // leaky_relu(negative_slope=0.2) (:185)
                    }
                    for (int j = bq; j < eq; ++j) {
                        const float* const rj = pqa + __mul24(slot(j), N2);
                        const float t0 = rj[2 * C] + ai;
                        const float ex = __expf(fmaxf(t0, 0.2f * t0) - mx);           // (argument <= 0: one v_exp_f32)
                        den += ex;
                        const gt_f32x4 q = *reinterpret_cast<const gt_f32x4*>(rj + C + c);
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            float m = pi[k] + q[k];                                   // lin_ij([x_i || x_j]) (:193-194)
                            m = fmaxf(m, 0.2f * m) * ex;                              // leaky_relu (:195) * alpha (:198-200), normalised below
                            sum[k] += m;
                            mn[k] = fminf(mn[k], m);
                            mxv[k] = fmaxf(mxv[k], m);
                        }
                    }
                    gt_f32x4 o_sum = {0.f, 0.f, 0.f, 0.f}, o_mm = {0.f, 0.f, 0.f, 0.f};   // empty segment: scatter leaves zeros
                    if (d > 0) {
                        const float inv = 1.0f / (den + 1e-16f);
#pragma unroll
                        for (int k = 0; k < 4; ++k) { o_sum[k] = sum[k] * inv; o_mm[k] = mn[k] * inv + mxv[k] * inv; }   // scatter min + scatter max (:216)
                    }
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        o_sum[k] += bs[k];                                    // mean over the single head, + bias (:166-172)
                        o_mm[k] += bm[k];
                        if (prelu_slope >= 0.0f) {                            // F.prelu(x, 0.1) between layers (Teacher_model.py:219-227)
                            o_sum[k] = o_sum[k] > 0.0f ? o_sum[k] : prelu_slope * o_sum[k];
                            o_mm[k] = o_mm[k] > 0.0f ? o_mm[k] : prelu_slope * o_mm[k];
                        }
                    }
                    if (!bad) {
                        *reinterpret_cast<gt_f32x4*>(out + (size_t)(base + li) * 2 * C + c) = o_sum;
                        *reinterpret_cast<gt_f32x4*>(out + (size_t)(base + li) * 2 * C + C + c) = o_mm;
                    }
                }
            };
            if (staged) nodes(std::true_type{}); else nodes(std::false_type{});
        }
        __syncthreads();                                                 // (the next tile overwrites the LDS tile)
    }
}

template <int C, int CIN>
static int launch_gat_tiled(int n_tiles, const int* tile_ptr, const int* rowptr, const int* src, const float* X, const float* Wf,
                            const float* vec, const float* bias, float slope, float* out, hipStream_t s) {
    constexpr int N2 = 2 * C + 4;
    constexpr int NT = (N2 + 15) / 16;
    const size_t wop_bytes = CIN >= 16 ? (size_t)NT * (CIN / 16) * 64 * 16 : (size_t)GT_TM * 4;
    const size_t lds = (size_t)GT_TM * N2 * 4 + (size_t)(GT_TM + 4) * 4 + (size_t)GT_EM * 2 + wop_bytes + 16;
    auto kern = gat_tile_kernel<C, CIN>;
    // per device, behind a lock: a process may drive several GPUs from several threads (one CU count and one attribute flag for all of
    // them sized every grid from the first device and raced on the bookkeeping)
    static std::mutex mu;
    static bool attr_set[64] = {};
    static int cus_of[64] = {};
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TLC_ERR_HIP;
    {
        std::lock_guard<std::mutex> lock(mu);
        if (!attr_set[dev]) {
            TLC_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set[dev] = true;
        }
        if (!cus_of[dev]) {
            int v = 0;
            TLC_HIP_CHECK(hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev));
            cus_of[dev] = v;
        }
        cus = cus_of[dev];
    }
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(2, (160 * 1024) / lds));
    const int grid = std::min(n_tiles, cus * per_cu);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(GT_THREADS), lds, s, n_tiles, tile_ptr, rowptr, src, X, Wf, vec, bias, slope, out);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// ---- generic scatter (message_passing.py:275-293 -> torch_scatter.scatter, reduce sum / mean / min / max) ----------------
// float atomics at the memory side; min / max order floats through their monotone integer image.
__device__ __forceinline__ int f32_ord(float x) {
    const int b = __float_as_int(x);
    return b >= 0 ? b : (b ^ 0x7fffffff);
}
__device__ __forceinline__ float ord_f32(int o) { return __int_as_float(o >= 0 ? o : (o ^ 0x7fffffff)); }

__global__ void scatter_init_kernel(long long n, int reduce, float* __restrict__ out, int* __restrict__ cnt, long long n_rows) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) {
        if (reduce == 2) reinterpret_cast<int*>(out)[t] = 0x7fffffff;          // +max of the ordered image
        else if (reduce == 3) reinterpret_cast<int*>(out)[t] = (int)0x80000000;
        else out[t] = 0.0f;
    }
    if (cnt && t < n_rows) cnt[t] = 0;
}
__global__ void scatter_kernel(long long n_src, const long long* __restrict__ index, const float* __restrict__ src, int k, int reduce,
                               int n_out, float* __restrict__ out, int* __restrict__ cnt) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_src * k) return;
    const long long e = t / k;
    const int c = (int)(t - e * k);
    const long long i = index[e];
    if (i < 0 || i >= n_out) return;
    const float v = src[t];
    float* dst = out + i * k + c;
    if (reduce <= 1) atomicAdd(dst, v);
    else if (reduce == 2) atomicMin(reinterpret_cast<int*>(dst), f32_ord(v));
    else atomicMax(reinterpret_cast<int*>(dst), f32_ord(v));
    if (cnt && c == 0) atomicAdd(&cnt[i], 1);
}
__global__ void scatter_finish_kernel(long long n, int k, int reduce, float* __restrict__ out, const int* __restrict__ cnt) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n) return;
    const int c = cnt[t / k];
    if (reduce == 1) out[t] = c > 0 ? out[t] / (float)c : 0.0f;
    else out[t] = c > 0 ? ord_f32(reinterpret_cast<int*>(out)[t]) : 0.0f;      // empty segments stay 0 (torch_scatter)
}

}  // namespace

extern "C" int tlc_scatter_f32(int64_t n_src, const int64_t* d_index, const float* d_src, int32_t k, int reduce, int32_t n_out,
                               float* d_out, int32_t* d_count_work, void* stream) {
    TLC_REQUIRE(n_src >= 0 && k > 0 && n_out >= 0 && reduce >= 0 && reduce <= 3, "bad arguments");
    if (n_out == 0) return TLC_OK;
    TLC_REQUIRE(d_out && (n_src == 0 || (d_index && d_src)), "null pointer");
    TLC_REQUIRE(reduce == 0 || d_count_work, "mean / min / max need the int32[n_out] count workspace");
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)n_out * k;
    int* cnt = reduce == 0 ? nullptr : d_count_work;
    hipLaunchKernelGGL(scatter_init_kernel, dim3((unsigned)((std::max<long long>(total, n_out) + 255) / 256)), dim3(256), 0, s, total, reduce,
                       d_out, cnt, (long long)n_out);
    if (n_src > 0)
        hipLaunchKernelGGL(scatter_kernel, dim3((unsigned)((n_src * k + 255) / 256)), dim3(256), 0, s, (long long)n_src,
                           (const long long*)d_index, d_src, k, reduce, n_out, d_out, cnt);
    if (reduce != 0)
        hipLaunchKernelGGL(scatter_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, total, k, reduce, d_out,
                           (const int*)cnt);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

extern "C" int tlc_gat_layer_fwd(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_src, const float* d_X, int32_t c_in,
                                 int32_t c_out, const float* d_Wl, const float* d_att, const float* d_Wij, const float* d_bias,
                                 float prelu_slope, float* d_work, float* d_out, void* stream) {
    TLC_REQUIRE(n_nodes >= 0 && c_in > 0, "bad sizes");
    TLC_REQUIRE(c_out == 8 || c_out == 16 || c_out == 32 || c_out == 64, "c_out must be 8, 16, 32 or 64");
    if (n_nodes == 0) return TLC_OK;
    TLC_REQUIRE(d_rowptr && d_src && d_X && d_Wl && d_att && d_Wij && d_bias && d_work && d_out, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    switch (c_out) {
        case 8: return launch_gat<8>(n_nodes, d_rowptr, d_src, d_X, c_in, d_Wl, d_att, d_Wij, d_bias, prelu_slope, d_work, d_out, s);
        case 16: return launch_gat<16>(n_nodes, d_rowptr, d_src, d_X, c_in, d_Wl, d_att, d_Wij, d_bias, prelu_slope, d_work, d_out, s);
        case 32: return launch_gat<32>(n_nodes, d_rowptr, d_src, d_X, c_in, d_Wl, d_att, d_Wij, d_bias, prelu_slope, d_work, d_out, s);
        default: return launch_gat<64>(n_nodes, d_rowptr, d_src, d_X, c_in, d_Wl, d_att, d_Wij, d_bias, prelu_slope, d_work, d_out, s);
    }
}

template <typename IT>
static int edge_head_fwd_impl(int64_t n_edges, const IT* d_src, const IT* d_dst, const float* d_X, int32_t c, const float* d_W5, const float* d_b5,
                              int32_t hidden, float prelu_slope, const float* d_W6, const float* d_b6, float* d_pd, double* d_pd64, int32_t n_nodes,
                              float* d_work, void* stream) {
    TLC_REQUIRE(n_edges >= 0 && c > 0 && hidden > 0, "bad sizes");
    if (n_edges == 0) return TLC_OK;
    TLC_REQUIRE(d_src && d_dst && d_X && d_W5 && d_b5 && d_W6 && d_b6 && d_pd, "null pointer");
    if (c == 32 && hidden == 32 && (reinterpret_cast<uintptr_t>(d_X) & 15) == 0) {
        const long long tiles = (n_edges + 15) / 16;
        long long blocks = (tiles + 3) / 4;
        if (blocks > 256 * 4) blocks = 256 * 4;                       // persistent wavefronts: four per SIMD (98 registers)
        hipLaunchKernelGGL(edge_head_mfma_kernel<IT>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (long long)n_edges, d_src, d_dst, d_X, d_W5,
                           d_b5, prelu_slope, d_W6, d_b6, d_pd, d_pd64);
        TLC_HIP_CHECK(hipGetLastError());
        return TLC_OK;
    }
    if (d_work && n_nodes > 0 && hidden % 4 == 0 && 2 * hidden <= 128) {
        // per-node projections U | V (work[0 .. n_nodes * 2 * hidden)) on the MFMA GEMM, then the per-edge gather
        float* d_UV = d_work;
        float* d_Bt = d_work + (size_t)n_nodes * 2 * hidden;
        hipLaunchKernelGGL(edge_head_pack_kernel, dim3((unsigned)((c * 2 * hidden + 255) / 256)), dim3(256), 0, (hipStream_t)stream, c,
                           hidden, d_W5, d_Bt);
        TLC_HIP_CHECK(hipGetLastError());
        int rc = tlc_gemm_f32(n_nodes, 2 * hidden, c, d_X, d_Bt, nullptr, 0, d_UV, stream);
        if (rc != TLC_OK) return rc;
        const bool al16 = ((reinterpret_cast<uintptr_t>(d_b5) | reinterpret_cast<uintptr_t>(d_W6) | reinterpret_cast<uintptr_t>(d_UV)) & 15) == 0;
#define TLC_EH_LANES(LPE_)                                                                                                                   \
    hipLaunchKernelGGL((edge_head_gather_lanes_kernel<IT, LPE_>), dim3((unsigned)((n_edges * LPE_ + 255) / 256)), dim3(256), 0, (hipStream_t)stream, \
                       (long long)n_edges, d_src, d_dst, (const float*)d_UV, d_b5, prelu_slope, d_W6, d_b6, d_pd, d_pd64)
        if (al16 && hidden == 32) TLC_EH_LANES(8);
        else if (al16 && hidden == 64) TLC_EH_LANES(16);
        else if (al16 && hidden == 16) TLC_EH_LANES(4);
        else if (al16 && hidden == 8) TLC_EH_LANES(2);
        else
            hipLaunchKernelGGL(edge_head_gather_kernel<IT>, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), 3 * (size_t)hidden * sizeof(float),
                               (hipStream_t)stream, (long long)n_edges, d_src, d_dst, (const float*)d_UV, hidden, d_b5, prelu_slope, d_W6,
                               d_b6, d_pd, d_pd64);
#undef TLC_EH_LANES
        TLC_HIP_CHECK(hipGetLastError());
        return TLC_OK;
    }
    if constexpr (std::is_same<IT, int>::value) {
        if (!d_pd64) {
            const size_t lds = ((size_t)hidden * 2 * c + 2 * (size_t)hidden) * sizeof(float);
            TLC_REQUIRE(lds <= 64 * 1024, "edge head weights do not fit LDS");
            hipLaunchKernelGGL(edge_head_kernel, dim3((unsigned)((n_edges + 255) / 256)), dim3(256), lds, (hipStream_t)stream,
                               (long long)n_edges, d_src, d_dst, d_X, c, d_W5, d_b5, hidden, prelu_slope, d_W6, d_b6, d_pd);
            TLC_HIP_CHECK(hipGetLastError());
            return TLC_OK;
        }
    }
    tlc_set_error("edge head: this form needs the per-node scratch (hidden a multiple of 4, at most 64)");
    return TLC_ERR_UNSUPPORTED;
}

extern "C" int tlc_edge_head_fwd(int64_t n_edges, const int32_t* d_src, const int32_t* d_dst, const float* d_X, int32_t c,
                                 const float* d_W5, const float* d_b5, int32_t hidden, float prelu_slope, const float* d_W6,
                                 const float* d_b6, float* d_pd, int32_t n_nodes, float* d_work, void* stream) {
    return edge_head_fwd_impl<int>(n_edges, d_src, d_dst, d_X, c, d_W5, d_b5, hidden, prelu_slope, d_W6, d_b6, d_pd, nullptr, n_nodes, d_work, stream);
}
// (gat_internal.h) the head over an int64 edge_index [2][width] as it lies (sources in the first row, targets in the second), the
// points also as float64
int tlc_edge_head_fwd_i64(int64_t n_edges, const long long* d_src, const long long* d_dst, const float* d_X, int c, const float* d_W5, const float* d_b5,
                          int hidden, float prelu_slope, const float* d_W6, const float* d_b6, float* d_pd, double* d_pd64, int n_nodes, float* d_work,
                          hipStream_t stream) {
    return edge_head_fwd_impl<long long>(n_edges, d_src, d_dst, d_X, c, d_W5, d_b5, hidden, prelu_slope, d_W6, d_b6, d_pd, d_pd64, n_nodes, d_work, (void*)stream);
}

// One PDGNN layer on a block-diagonal batch cut into self-contained tiles (see gat_tile_kernel): d_tile_ptr int32[n_tiles + 1], node
// offsets of tiles of at most 192 consecutive nodes such that every in-edge of a tile's node has its source in the same tile (the
// caller's business: Knowledge_Distillation/gat_conv.py, GraphBatch).  c_in = 1 or 64 with c_out = 32, 16 (the PDGNN layers); any
// other shape: TLC_ERR_UNSUPPORTED (the caller takes tlc_gat_layer_fwd).  d_work: float32[c_in * c_out + c_out * (2 c_out + 4) +
// c_in * (2 c_out + 4) + 2 c_out + 8] of scratch for the packed and combined weights.  Same results as tlc_gat_layer_fwd up to the
// rounding of the node rows' fp32 sums.
extern "C" int tlc_gat_layer_tiled_fwd(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_src, int32_t n_tiles,
                                       const int32_t* d_tile_ptr, const float* d_X, int32_t c_in, int32_t c_out, const float* d_Wl,
                                       const float* d_att, const float* d_Wij, const float* d_bias, float prelu_slope, float* d_work,
                                       float* d_out, void* stream) {
    TLC_REQUIRE(n_nodes >= 0 && c_in > 0 && n_tiles >= 0, "bad sizes");
    if (n_nodes == 0 || n_tiles == 0) return TLC_OK;
    TLC_REQUIRE(d_rowptr && d_src && d_tile_ptr && d_X && d_Wl && d_att && d_Wij && d_bias && d_work && d_out, "null pointer");
    if (!((c_in == 1 || c_in == 64) && (c_out == 32 || c_out == 16))) { tlc_set_error("tlc_gat_layer_tiled_fwd: c_in %d / c_out %d not built", c_in, c_out); return TLC_ERR_UNSUPPORTED; }
    hipStream_t s = (hipStream_t)stream;
    const TlcGatPrepLayer layer = {c_in, c_out, d_Wl, d_att, d_Wij, d_work};
    const int rc = tlc_gat_tiled_prepare(1, &layer, s);
    if (rc != TLC_OK) return rc;
    return tlc_gat_tiled_run(n_tiles, d_tile_ptr, d_rowptr, d_src, d_X, c_in, c_out, d_work, d_bias, prelu_slope, d_out, s);
}

// The combined weights of up to four tiled layers, one workgroup per layer (see gat_internal.h).  The products are 64 x 68 x 32 at
// most: summed per output in LDS, in the order k = 0 .. C-1 -- a forward's weights are ready after ONE launch (the MFMA GEMM the
// layers used before was a launch per layer behind a packing launch per layer).
namespace {
struct GatPrepArgs { TlcGatPrepLayer l[TLC_GAT_PREP_MAX]; };
constexpr int GAT_PREP_SPLIT = 8;                              // workgroups per layer (one: 19.6 us in front of the first layer)
__global__ __launch_bounds__(256) void gat_tiled_prep_kernel(GatPrepArgs a) {
    __shared__ float sWl[32 * 64];                             // [C][c_in], as stored
    __shared__ float sB2[32 * 68];                             // [C][2C + 4] = [Wij_t^T | Wij_s^T | att | 0 0 0]
    const TlcGatPrepLayer L = a.l[blockIdx.x / GAT_PREP_SPLIT];
    const int part = blockIdx.x % GAT_PREP_SPLIT;
    const int C = L.c_out, N2 = 2 * C + 4, c_in = L.c_in;
    for (int t = threadIdx.x; t < C * c_in; t += 256) sWl[t] = L.Wl[t];
    for (int t = threadIdx.x; t < C * N2; t += 256) {
        const int k = t / N2, j = t - k * N2;
        sB2[t] = j < C ? L.Wij[(size_t)j * 2 * C + k] : (j < 2 * C ? L.Wij[(size_t)(j - C) * 2 * C + C + k] : (j == 2 * C ? L.att[k] : 0.0f));
    }
    __syncthreads();
    const int total = c_in == 1 ? 2 * C + 1 : c_in * N2;
    for (int t = part * 256 + threadIdx.x; t < total; t += 256 * GAT_PREP_SPLIT) {
        const int k = t / N2, j = t - k * N2;                 // (c_in == 1: k = 0, j = t)
        float acc = 0.0f;
        for (int c = 0; c < C; ++c) acc += sWl[c * c_in + k] * sB2[c * N2 + j];
        L.prep[t] = acc;
    }
}
}  // namespace
int tlc_gat_tiled_prepare(int n_layers, const TlcGatPrepLayer* layers, hipStream_t s) {
    TLC_REQUIRE(n_layers > 0 && n_layers <= TLC_GAT_PREP_MAX && layers, "bad sizes");
    GatPrepArgs a = {};
    for (int i = 0; i < n_layers; ++i) {
        const TlcGatPrepLayer& L = layers[i];
        if (!((L.c_in == 1 || L.c_in == 64) && (L.c_out == 32 || L.c_out == 16))) { tlc_set_error("tlc_gat_layer_tiled_fwd: c_in %d / c_out %d not built", L.c_in, L.c_out); return TLC_ERR_UNSUPPORTED; }
        TLC_REQUIRE(L.Wl && L.att && L.Wij && L.prep, "null pointer");
        a.l[i] = L;
    }
    hipLaunchKernelGGL(gat_tiled_prep_kernel, dim3(n_layers * GAT_PREP_SPLIT), dim3(256), 0, s, a);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}
int tlc_gat_tiled_run(int n_tiles, const int* tile_ptr, const int* rowptr, const int* src, const float* X, int c_in, int c_out, const float* prep,
                      const float* bias, float prelu_slope, float* out, hipStream_t s) {
    if (n_tiles <= 0) return TLC_OK;
    if (!((c_in == 1 || c_in == 64) && (c_out == 32 || c_out == 16))) { tlc_set_error("tlc_gat_layer_tiled_fwd: c_in %d / c_out %d not built", c_in, c_out); return TLC_ERR_UNSUPPORTED; }
    if (c_in == 1)
        return c_out == 32 ? launch_gat_tiled<32, 1>(n_tiles, tile_ptr, rowptr, src, X, nullptr, prep, bias, prelu_slope, out, s)
                           : launch_gat_tiled<16, 1>(n_tiles, tile_ptr, rowptr, src, X, nullptr, prep, bias, prelu_slope, out, s);
    return c_out == 32 ? launch_gat_tiled<32, 64>(n_tiles, tile_ptr, rowptr, src, X, prep, nullptr, bias, prelu_slope, out, s)
                       : launch_gat_tiled<16, 64>(n_tiles, tile_ptr, rowptr, src, X, prep, nullptr, bias, prelu_slope, out, s);
}

// The tile cut of tlc_gat_layer_tiled_fwd (Knowledge_Distillation/gat_conv.py GraphBatch -> ops.gat_tiles), all on the device
// and with ONE host read at its end (the torch formulation it replaces -- a segmented min/max, a prefix sum, nonzero, searchsorted,
// unique -- cost 0.43 ms and four host round trips per batch, more than the four layers of a 4096-vicinity batch).
//   position k (the cut between node k-1 and node k, k = 0 .. n) is FREE when no edge crosses it: row i with smallest / largest
//   column a / b (i itself included) forbids a+1 .. b.  A bitmap of the forbidden positions (mark), the largest distance between
//   consecutive free positions `gap` (gap), then with step = tile_nodes - gap the first free position at or behind every multiple
//   of step starts a tile (pick): consecutive picks are less than step + gap = tile_nodes apart.  No cut when 2 gap > tile_nodes.
namespace {
enum { TC_BAD = 0, TC_GAP = 1, TC_NT = 2, TC_HEAD = 4 };
constexpr int TC_HALF_MAX = 512;                               // tile_nodes <= 1024
// 256 rows per workgroup: their forbidden positions lie within `half` of the rows, so they are OR-ed in LDS first and go out as one
// atomic per touched word (one atomic per row and word made 80 us of same-address traffic on a million molecule nodes)
__global__ void __launch_bounds__(256) tile_cut_mark_kernel(int n, int half, const int* __restrict__ rowptr, const int* __restrict__ col, unsigned* __restrict__ bits, int* __restrict__ st) {
    __shared__ unsigned sb[(256 + 2 * TC_HALF_MAX) / 32 + 2];
    const int i0 = blockIdx.x * 256, i = i0 + threadIdx.x;
    const int wb = max(i0 - half, 0) >> 5, nw = ((min(i0 + 255, n - 1) + half) >> 5) - wb + 1;
    if ((int)threadIdx.x < nw) sb[threadIdx.x] = 0u;
    __syncthreads();
    if (i < n) {
        int a = i, b = i;
        for (int j = rowptr[i]; j < rowptr[i + 1]; ++j) { const int c = col[j]; a = c < a ? c : a; b = c > b ? c : b; }
        if (b - a >= half) st[TC_BAD] = 1;                     // b - a forbidden positions in a row: gap > half
        else
            for (int w = (a + 1) >> 5; w <= (b >> 5) && a < b; ++w) {
                const int p0 = max(a + 1, w << 5), p1 = min(b, (w << 5) + 31), len = p1 - p0 + 1;
                atomicOr(&sb[w - wb], (len == 32 ? ~0u : ((1u << len) - 1u)) << (p0 & 31));
            }
    }
    __syncthreads();
    if ((int)threadIdx.x < nw && sb[threadIdx.x]) atomicOr(&bits[wb + threadIdx.x], sb[threadIdx.x]);
}
// the largest free position below f, looking back `limit` positions at most; -1 when there is none that close
__device__ __forceinline__ int tile_cut_prev_free(const unsigned* __restrict__ bits, int f, int limit) {
    int p = f - 1;
    const int stop = max(f - limit, 0);
    while (p >= stop) {
        const int sh = p & 31;
        const unsigned w = ~bits[p >> 5] & (sh == 31 ? ~0u : ((2u << sh) - 1u));
        if (w) { const int q = (p & ~31) + 31 - __clz(w); return q >= stop ? q : -1; }
        p = (p & ~31) - 1;
    }
    return -1;
}
constexpr int TC_GAP_PER_WG = 2048;
__global__ void __launch_bounds__(256) tile_cut_gap_kernel(int n, int half, const unsigned* __restrict__ bits, int* __restrict__ st) {
    __shared__ int wmax[4];
    int d = 0;
    for (int f = blockIdx.x * TC_GAP_PER_WG + threadIdx.x + 1, e = min(n, (blockIdx.x + 1) * TC_GAP_PER_WG); f <= e; f += 256)
        if (!((bits[f >> 5] >> (f & 31)) & 1u)) {
            const int pf = tile_cut_prev_free(bits, f, half);
            if (pf < 0) st[TC_BAD] = 1; else d = max(d, f - pf);
        }
    for (int o = 32; o; o >>= 1) d = max(d, __shfl_xor(d, o));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = d;
    __syncthreads();
    // one atomic per workgroup at most, and only for a new maximum (read at the L2: a plain load stays the zero this CU cached
    // first, and an atomic per wave on the one address took 170 us on a million nodes)
    if (threadIdx.x == 0) {
        d = max(max(wmax[0], wmax[1]), max(wmax[2], wmax[3]));
        if (d > __hip_atomic_load(&st[TC_GAP], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&st[TC_GAP], d);
    }
}
__global__ void tile_cut_pick_kernel(int n, int tile_nodes, const unsigned* __restrict__ bits, int* __restrict__ st, int* __restrict__ tiles) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    const int gap = st[TC_GAP];
    if (st[TC_BAD] || 2 * gap > tile_nodes) return;            // st[TC_NT] stays 0: no cut
    if (f > n || ((bits[f >> 5] >> (f & 31)) & 1u)) return;
    if (f == 0) { tiles[0] = 0; return; }
    const int step = tile_nodes - gap;
    const int pf = tile_cut_prev_free(bits, f, gap);
    if (f == n) { tiles[pf / step + 1] = n; st[TC_NT] = pf / step + 1; }
    else if (f / step > pf / step) tiles[f / step] = f;
}
}  // namespace
extern "C" int tlc_gat_tile_cut(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, int32_t tile_nodes, int32_t* d_work,
                                int32_t* d_tile_ptr, int32_t* n_tiles, void* stream) {
    TLC_REQUIRE(n_nodes >= 0 && tile_nodes >= 2 && tile_nodes <= 2 * TC_HALF_MAX && n_tiles, "bad sizes");
    *n_tiles = 0;
    if (n_nodes == 0) return TLC_OK;
    TLC_REQUIRE(d_rowptr && d_col && d_work && d_tile_ptr, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    int* st = d_work;
    unsigned* bits = (unsigned*)(d_work + TC_HEAD);
    const size_t words = (size_t)n_nodes / 32 + 2;
    TLC_HIP_CHECK(hipMemsetAsync(d_work, 0, (TC_HEAD + words) * sizeof(int), s));
    const unsigned g = (unsigned)((n_nodes + 1 + 255) / 256);
    hipLaunchKernelGGL(tile_cut_mark_kernel, dim3(g), dim3(256), 0, s, n_nodes, tile_nodes / 2, d_rowptr, d_col, bits, st);
    hipLaunchKernelGGL(tile_cut_gap_kernel, dim3((unsigned)(n_nodes / TC_GAP_PER_WG + 1)), dim3(256), 0, s, n_nodes, tile_nodes / 2, bits, st);
    hipLaunchKernelGGL(tile_cut_pick_kernel, dim3(g), dim3(256), 0, s, n_nodes, tile_nodes, bits, st, d_tile_ptr);
    TLC_HIP_CHECK(hipGetLastError());
    int h[TC_HEAD] = {0, 0, 0, 0};
    TLC_HIP_CHECK(hipMemcpyAsync(h, st, sizeof(h), hipMemcpyDeviceToHost, s));
    TLC_HIP_CHECK(hipStreamSynchronize(s));
    *n_tiles = h[TC_NT];
    return TLC_OK;
}

// gat_backward.hip -- SURVEY.md 8(f) item 4: backward of the PDGNN layer and of the edge head (what `loss.backward()` does
// in the reference's training loop, Knowledge_Distillation/train_Teacher_Model.py:55-62, through
// Knowledge_Distillation/gat_conv.py:113-216 and Teacher_model.py:53-59).
//
//   tlc_gat_layer_bwd   d(out)/d(X, Wl, att, Wij, bias) of one GATConv(heads=1, new_node_feat, use_edge_attn) layer
//   tlc_edge_head_bwd   d(pd)/d(X, W5, b5, W6, b6) of lin6(prelu(lin5([x_s || x_t])))
//
// Written for clarity first: nothing of the forward is kept, every kernel recomputes what it needs from the layer's input and
// weights (the forward's fused node rows [P | Q | alpha] are not stored either).  The three scatters of the forward
// (sum / min / max at the target, gat_conv.py:216) become one wavefront per target row again: the row's softmax, its channel-wise
// minima and maxima, the gradient of every message -- four passes over the row's in-edges, lanes = channels; what flows to the
// SOURCE of an edge (d alpha_j, d Q_j) goes through float atomics.  Weight gradients are sums over nodes / edges of outer
// products: one reduction kernel (tlc_xty) with per-workgroup partial sums and float atomics.
// fp32 throughout, like the forward; FMA contraction welcome (Makefile).
#include "tlc_common.h"

extern "C" int tlc_gemm_f32(int32_t M, int32_t N, int32_t K, const float* d_A, const float* d_B, const float* d_bias, int relu,
                            float* d_C, void* stream);

namespace {

__device__ __forceinline__ float lrelu(float x, float s) { return x > 0.f ? x : s * x; }
__device__ __forceinline__ float wave_sum_f32(float v) {
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// out[a][b] += sum_i A[i][a] * B[i][b]   (A [n, lda >= na], B [n, ldb >= nb]; out [na, nb] zeroed by the caller; na, nb <= 64)
// A workgroup takes slabs of 1 024 rows, 32 at a time through LDS; its 256 threads are four row groups (eight of the 32 rows each) of
// 64 threads, and a thread owns 4 x 4 blocks of the output: four values of A and four of B per row feed sixteen FMAs (one value
// of each per FMA made the kernel LDS-bound: 0.2 ms per call on 200 k rows).  Partial sums leave by float atomics.
__global__ __launch_bounds__(256) void xty_kernel(long long n, const float* __restrict__ A, int lda, int na, const float* __restrict__ B,
                                                  int ldb, int nb, float* __restrict__ out) {
    __shared__ float sa[32][68], sb[32][68];            // (rows padded: 68 floats keep the float4 reads 16-byte aligned and off one bank)
    const int tid = threadIdx.x, grp = tid >> 6, t64 = tid & 63;
    const int ba = (na + 3) >> 2, bb = (nb + 3) >> 2, nblk = ba * bb;         // <= 256 blocks of 4 x 4
    float acc[4][16];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
    const long long slab = 1024;
    // (the next 32-row tile is loaded into registers while this one is multiplied: a workgroup has 32 tiles and nothing else to hide
    // their round trips behind)
    float pa[8], pb[8];
    auto fetch = [&](long long t0, long long r1) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = tid + 256 * q, r = k >> 6, c = k & 63;
            const bool in = t0 + r < r1;
            pa[q] = (in && c < na) ? A[(t0 + r) * lda + c] : 0.f;
            pb[q] = (in && c < nb) ? B[(t0 + r) * ldb + c] : 0.f;
        }
    };
    for (long long r0 = (long long)blockIdx.x * slab; r0 < n; r0 += (long long)gridDim.x * slab) {
        const long long r1 = r0 + slab < n ? r0 + slab : n;
        fetch(r0, r1);
        for (long long t0 = r0; t0 < r1; t0 += 32) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int k = tid + 256 * q; sa[k >> 6][k & 63] = pa[q]; sb[k >> 6][k & 63] = pb[q]; }
            __syncthreads();
            if (t0 + 32 < r1) fetch(t0 + 32, r1);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int blk = t64 + 64 * j;
                if (blk < nblk) {                                           // (uniform per j beyond the last full 64)
                    const int a0 = (blk / bb) * 4, b0 = (blk % bb) * 4;
#pragma unroll
                    for (int rr = 0; rr < 8; ++rr) {
                        const int r = grp * 8 + rr;
                        const float4 av = *reinterpret_cast<const float4*>(&sa[r][a0]);
                        const float4 bv = *reinterpret_cast<const float4*>(&sb[r][b0]);
                        const float a4[4] = {av.x, av.y, av.z, av.w}, b4[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
                        for (int x = 0; x < 4; ++x)
#pragma unroll
                            for (int y = 0; y < 4; ++y) acc[j][x * 4 + y] += a4[x] * b4[y];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int blk = t64 + 64 * j;
        if (blk < nblk) {
            const int a0 = (blk / bb) * 4, b0 = (blk % bb) * 4;
#pragma unroll
            for (int x = 0; x < 4; ++x)
#pragma unroll
                for (int y = 0; y < 4; ++y)
                    if (a0 + x < na && b0 + y < nb && acc[j][x * 4 + y] != 0.f) atomicAdd(&out[(a0 + x) * nb + b0 + y], acc[j][x * 4 + y]);
        }
    }
}

static int xty(long long n, const float* A, int lda, int na, const float* B, int ldb, int nb, float* out, hipStream_t s) {
    if (n <= 0 || na <= 0 || nb <= 0) return TLC_OK;
    if (na > 64 || nb > 64) { tlc_set_error("xty: %d x %d outputs (at most 64 x 64)", na, nb); return TLC_ERR_UNSUPPORTED; }
    const int grid = (int)std::min<long long>((n + 1023) / 1024, 2048);
    hipLaunchKernelGGL(xty_kernel, dim3(grid), dim3(256), 0, s, n, A, lda, na, B, ldb, nb, out);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// ---- PDGNN layer ----------------------------------------------------------------------------------------------------------
// node rows recomputed for the backward: xl [n, C], then pqa [n, 2C + 1] = [P | Q | alpha]
__global__ void gatb_xl_kernel(int n, int C, int c_in, const float* __restrict__ X, const float* __restrict__ Wl, float* __restrict__ xl) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * C) return;
    const int i = (int)(t / C), c = (int)(t % C);
    float s = 0.f;
    for (int k = 0; k < c_in; ++k) s += X[(size_t)i * c_in + k] * Wl[c * c_in + k];
    xl[t] = s;
}
__global__ void gatb_pqa_kernel(int n, int C, const float* __restrict__ xl, const float* __restrict__ att, const float* __restrict__ Wij,
                                float* __restrict__ pqa) {
    const int S = 2 * C + 1;                              // (the fallback's rows are unpadded)
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * S) return;
    const int i = (int)(t / S), j = (int)(t % S);
    const float* x = xl + (size_t)i * C;
    float s = 0.f;
    if (j < C) for (int k = 0; k < C; ++k) s += Wij[j * 2 * C + k] * x[k];                    // P: target half of lin_ij
    else if (j < 2 * C) for (int k = 0; k < C; ++k) s += Wij[(j - C) * 2 * C + C + k] * x[k];  // Q: source half
    else for (int k = 0; k < C; ++k) s += att[k] * x[k];
    pqa[t] = s;
}

// the layer's weights as the right-hand sides of the two products that recompute the node rows:
//   Bt1 [c_in, C] = Wl^T;   Bt2 [C, 2C + 4] = [Wij[:, :C]^T | Wij[:, C:]^T | att | 0 0 0]
__global__ void gatb_pack_kernel(int C, int c_in, const float* __restrict__ Wl, const float* __restrict__ att, const float* __restrict__ Wij,
                                 float* __restrict__ Bt1, float* __restrict__ Bt2) {
    const int N2 = 2 * C + 4;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < c_in * C) { const int q = t / C, c = t % C; Bt1[t] = Wl[c * c_in + q]; }
    if (t < C * N2) {
        const int k = t / N2, j = t % N2;
        Bt2[t] = j < C ? Wij[j * 2 * C + k] : (j < 2 * C ? Wij[(j - C) * 2 * C + C + k] : (j == 2 * C ? att[k] : 0.f));
    }
}

// One wavefront per target row.  G = d loss / d out [n, 2C] (of the layer's OUTPUT: with a fused PReLU its derivative is taken
// here from the sign of out).  Writes gP [n, C] (row-owned), adds into gQ [n, C] and gAlpha [n] (atomics: a node is the source of
// many edges), and stores the activated gradient Gz [n, 2C] (= d loss / d (pre-activation), whose column sums are d bias).
template <int C>
__global__ __launch_bounds__(64) void gatb_edge_kernel(int n, const int* __restrict__ rowptr, const int* __restrict__ src,
                                                       const float* __restrict__ pqa, int S, const float* __restrict__ out, float prelu_slope,
                                                       const float* __restrict__ G, float* __restrict__ Gz, float* __restrict__ gP,
                                                       float* __restrict__ gQ, float* __restrict__ gAlpha) {
    const int lane = tlc_lane();                      // (S: floats per row of pqa = [P | Q | alpha | padding])
    const bool ch = lane < C;
    for (int i = blockIdx.x; i < n; i += gridDim.x) {
        const int rb = rowptr[i], re = rowptr[i + 1];
        const float ai = pqa[(size_t)i * S + 2 * C];
        const float Pi = ch ? pqa[(size_t)i * S + lane] : 0.f;
        // gradient at the pre-activation (PReLU fused into the forward's output)
        float gs = 0.f, gmm = 0.f;
        if (ch) {
            gs = G[(size_t)i * 2 * C + lane]; gmm = G[(size_t)i * 2 * C + C + lane];
            if (prelu_slope >= 0.f) {
                if (out[(size_t)i * 2 * C + lane] <= 0.f) gs *= prelu_slope;
                if (out[(size_t)i * 2 * C + C + lane] <= 0.f) gmm *= prelu_slope;
            }
            Gz[(size_t)i * 2 * C + lane] = gs; Gz[(size_t)i * 2 * C + C + lane] = gmm;
        }
        // pass 1: softmax statistics of the row (lanes over edges)
        float tmax = -INFINITY;
        for (int e = rb + lane; e < re; e += 64) tmax = fmaxf(tmax, lrelu(pqa[(size_t)src[e] * S + 2 * C] + ai, 0.2f));
        tmax = wave_max_f32(tmax);
        float den = 0.f;
        for (int e = rb + lane; e < re; e += 64) den += __expf(lrelu(pqa[(size_t)src[e] * S + 2 * C] + ai, 0.2f) - tmax);
        den = wave_sum_f32(den) + 1e-16f;
        // pass 2: channel-wise minimum and maximum of the messages (lanes over channels, edges in turn)
        // (the FIRST edge of the row that attains it takes the gradient, like torch_scatter's arg output on the CPU)
        float mn = INFINITY, mx = -INFINITY;
        int emn = -1, emx = -1;
        for (int e = rb; e < re; ++e) {
            const int j = src[e];
            const float a = __expf(lrelu(pqa[(size_t)j * S + 2 * C] + ai, 0.2f) - tmax) / den;
            const float m = ch ? lrelu(Pi + pqa[(size_t)j * S + C + lane], 0.2f) * a : 0.f;
            if (m < mn) { mn = m; emn = e; }
            if (m > mx) { mx = m; emx = e; }
        }
        // pass 3: sum_e a_e * (d loss / d a_e), for the softmax backward
        float Ssum = 0.f;
        for (int e = rb; e < re; ++e) {
            const int j = src[e];
            const float a = __expf(lrelu(pqa[(size_t)j * S + 2 * C] + ai, 0.2f) - tmax) / den;
            const float h = ch ? lrelu(Pi + pqa[(size_t)j * S + C + lane], 0.2f) : 0.f;
            const float gm = ch ? gs + gmm * ((e == emn ? 1.f : 0.f) + (e == emx ? 1.f : 0.f)) : 0.f;
            Ssum += a * wave_sum_f32(gm * h);
        }
        // pass 4: the gradients
        float gPi = 0.f, gai = 0.f;
        for (int e = rb; e < re; ++e) {
            const int j = src[e];
            const float t = pqa[(size_t)j * S + 2 * C] + ai;
            const float a = __expf(lrelu(t, 0.2f) - tmax) / den;
            const float z = ch ? Pi + pqa[(size_t)j * S + C + lane] : 0.f;
            const float h = lrelu(z, 0.2f);
            const float gm = ch ? gs + gmm * ((e == emn ? 1.f : 0.f) + (e == emx ? 1.f : 0.f)) : 0.f;
            const float ga = wave_sum_f32(gm * h);
            const float gt = a * (ga - Ssum) * (t > 0.f ? 1.f : 0.2f);
            gai += gt;
            if (lane == 0) atomicAdd(&gAlpha[j], gt);
            if (ch) {
                const float gz = gm * a * (z > 0.f ? 1.f : 0.2f);
                gPi += gz;
                atomicAdd(&gQ[(size_t)j * C + lane], gz);
            }
        }
        if (ch) gP[(size_t)i * C + lane] = gPi;
        if (lane == 0) atomicAdd(&gAlpha[i], gai);
    }
}

// gxl[i][k] = sum_c gP[i][c] Wij[c][k] + gQ[i][c] Wij[c][C + k] + gAlpha[i] att[k]
__global__ void gatb_gxl_kernel(int n, int C, const float* __restrict__ gP, const float* __restrict__ gQ, const float* __restrict__ gAlpha,
                                const float* __restrict__ Wij, const float* __restrict__ att, float* __restrict__ gxl) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * C) return;
    const int i = (int)(t / C), k = (int)(t % C);
    float s = gAlpha[i] * att[k];
    for (int c = 0; c < C; ++c) s += gP[(size_t)i * C + c] * Wij[c * 2 * C + k] + gQ[(size_t)i * C + c] * Wij[c * 2 * C + C + k];
    gxl[t] = s;
}
// gX[i][q] = sum_c gxl[i][c] Wl[c][q]
__global__ void gatb_gx_kernel(int n, int C, int c_in, const float* __restrict__ gxl, const float* __restrict__ Wl, float* __restrict__ gX) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long long)n * c_in) return;
    const int i = (int)(t / c_in), q = (int)(t % c_in);
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += gxl[(size_t)i * C + c] * Wl[c * c_in + q];
    gX[t] = s;
}
// column sums: out[b] += sum_i A[i][b]   (nb <= 256 and a divisor of 256; four rows per thread in flight)
__global__ __launch_bounds__(256) void colsum_kernel(long long n, const float* __restrict__ A, int nb, float* __restrict__ out) {
    const int b = threadIdx.x % nb, lanes = 256 / nb;
    const int r = threadIdx.x / nb;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const long long step = (long long)gridDim.x * lanes;
    long long i = (long long)blockIdx.x * lanes + r;
    for (; i + 3 * step < n; i += 4 * step) {
        s0 += A[i * nb + b]; s1 += A[(i + step) * nb + b]; s2 += A[(i + 2 * step) * nb + b]; s3 += A[(i + 3 * step) * nb + b];
    }
    for (; i < n; i += step) s0 += A[i * nb + b];
    // one atomic per column and workgroup (per thread they pile up on nb addresses: 0.42 instead of 0.15 ms)
    __shared__ float red[256];
    red[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if ((int)threadIdx.x < nb) {
        float sum = 0.f;
        for (int k = 0; k < lanes; ++k) sum += red[k * nb + threadIdx.x];
        if (sum != 0.f) atomicAdd(&out[threadIdx.x], sum);
    }
}

template <int C>
static int launch_gat_bwd(int n, const int* rowptr, const int* src, const float* X, int c_in, const float* Wl, const float* att,
                          const float* Wij, float prelu_slope, const float* out, const float* G, float* gX, float* gWl, float* gAtt,
                          float* gWij, float* gBias, float* work, hipStream_t s) {
    // work: xl [n,C] | pqa [n,2C+4] | gP [n,C] | gQ [n,C] | gxl [n,C] | Gz [n,2C] | tmp [2,C,C] | Bt1 [c_in,C] | Bt2 [C,2C+4] | gAlpha [n]
    // (S = 2C + 4 when the rows come from the MFMA product -- the forward's layout -- else 2C + 1)
    const bool by_gemm = 2 * C + 4 <= 128;
    const int S = by_gemm ? 2 * C + 4 : 2 * C + 1;
    float* xl = work;
    float* pqa = xl + (size_t)n * C;
    float* gP = pqa + (size_t)n * (2 * C + 4);
    float* gQ = gP + (size_t)n * C;
    float* gxl = gQ + (size_t)n * C;
    float* Gz = gxl + (size_t)n * C;
    float* tmp0 = Gz + (size_t)n * 2 * C;
    float* Bt1 = tmp0 + (size_t)2 * C * C;
    float* Bt2 = Bt1 + (((size_t)c_in * C + 3) & ~(size_t)3);
    float* gAl = Bt2 + (size_t)C * (2 * C + 4);              // (last: n floats would put what follows off its 16-byte alignment)
    TLC_HIP_CHECK(hipMemsetAsync(gQ, 0, (size_t)n * C * sizeof(float), s));
    TLC_HIP_CHECK(hipMemsetAsync(gAl, 0, (size_t)n * sizeof(float), s));
    TLC_HIP_CHECK(hipMemsetAsync(gWl, 0, (size_t)C * c_in * sizeof(float), s));
    TLC_HIP_CHECK(hipMemsetAsync(gAtt, 0, (size_t)C * sizeof(float), s));
    TLC_HIP_CHECK(hipMemsetAsync(gWij, 0, (size_t)C * 2 * C * sizeof(float), s));
    TLC_HIP_CHECK(hipMemsetAsync(gBias, 0, (size_t)2 * C * sizeof(float), s));
    auto blocks = [](long long t) { return dim3((unsigned)((t + 255) / 256)); };
    // the node rows again: x_l = X Wl^T and [P | Q | alpha] = x_l Bt2, on the f32 MFMA like the forward (per node on the vector ALU
    // these two took 1.2 ms of a layer's backward on 200 k nodes)
    if (by_gemm) {
        const int np = std::max(c_in * C, C * (2 * C + 4));
        hipLaunchKernelGGL(gatb_pack_kernel, blocks(np), dim3(256), 0, s, C, c_in, Wl, att, Wij, Bt1, Bt2);
        TLC_HIP_CHECK(hipGetLastError());
        int rc2;
        if (c_in >= 16 && c_in % 4 == 0) { if ((rc2 = tlc_gemm_f32(n, C, c_in, X, Bt1, nullptr, 0, xl, s)) != TLC_OK) return rc2; }
        else hipLaunchKernelGGL(gatb_xl_kernel, blocks((long long)n * C), dim3(256), 0, s, n, C, c_in, X, Wl, xl);
        if ((rc2 = tlc_gemm_f32(n, 2 * C + 4, C, xl, Bt2, nullptr, 0, pqa, s)) != TLC_OK) return rc2;
    } else {
        hipLaunchKernelGGL(gatb_xl_kernel, blocks((long long)n * C), dim3(256), 0, s, n, C, c_in, X, Wl, xl);
        hipLaunchKernelGGL(gatb_pqa_kernel, blocks((long long)n * (2 * C + 1)), dim3(256), 0, s, n, C, (const float*)xl, att, Wij, pqa);
    }
    hipLaunchKernelGGL((gatb_edge_kernel<C>), dim3(std::min(n, 65536)), dim3(64), 0, s, n, rowptr, src, (const float*)pqa, S, out, prelu_slope, G,
                       Gz, gP, gQ, gAl);
    hipLaunchKernelGGL(gatb_gxl_kernel, blocks((long long)n * C), dim3(256), 0, s, n, C, (const float*)gP, (const float*)gQ,
                       (const float*)gAl, Wij, att, gxl);
    if (gX) hipLaunchKernelGGL(gatb_gx_kernel, blocks((long long)n * c_in), dim3(256), 0, s, n, C, c_in, (const float*)gxl, Wl, gX);
    TLC_HIP_CHECK(hipGetLastError());
    int rc;
    // d Wij[c][k] = sum_i gP[i][c] xl[i][k];  d Wij[c][C + k] = sum_i gQ[i][c] xl[i][k]: two [C, C] blocks of the [C, 2C] matrix
    float* tmp = tmp0;                                // [C, C] x 2: the two halves, interleaved into gWij below
    TLC_HIP_CHECK(hipMemsetAsync(tmp, 0, (size_t)2 * C * C * sizeof(float), s));
    if ((rc = xty(n, gP, C, C, xl, C, C, tmp, s)) != TLC_OK) return rc;
    if ((rc = xty(n, gQ, C, C, xl, C, C, tmp + C * C, s)) != TLC_OK) return rc;
    TLC_HIP_CHECK(hipMemcpy2DAsync(gWij, (size_t)2 * C * sizeof(float), tmp, (size_t)C * sizeof(float), (size_t)C * sizeof(float), C, hipMemcpyDeviceToDevice, s));
    TLC_HIP_CHECK(hipMemcpy2DAsync(gWij + C, (size_t)2 * C * sizeof(float), tmp + C * C, (size_t)C * sizeof(float), (size_t)C * sizeof(float), C, hipMemcpyDeviceToDevice, s));
    if ((rc = xty(n, gAl, 1, 1, xl, C, C, gAtt, s)) != TLC_OK) return rc;                          // d att[k] = sum_i gAlpha[i] xl[i][k]
    if ((rc = xty(n, gxl, C, C, X, c_in, c_in, gWl, s)) != TLC_OK) return rc;                      // d Wl[c][q] = sum_i gxl[i][c] X[i][q]
    hipLaunchKernelGGL(colsum_kernel, dim3(1024), dim3(256), 0, s, (long long)n, (const float*)Gz, 2 * C, gBias);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// ---- edge head -----------------------------------------------------------------------------------------------------------------
// one thread per edge: pre = W5 [x_s || x_t] + b5, h = prelu(pre), pd = W6 h + b6.  Stores cat [E, 2c], h [E, H], gpre [E, H] for the
// weight-gradient reductions and adds d loss / d x into gX (atomics: a node is an endpoint of several edges).
template <int H>
__global__ __launch_bounds__(128) void edge_head_bwd_kernel(long long n_edges, const int* __restrict__ src, const int* __restrict__ dst,
                                                            const float* __restrict__ X, int c, const float* __restrict__ W5,
                                                            const float* __restrict__ b5, float slope, const float* __restrict__ W6,
                                                            const float* __restrict__ Gpd, float* __restrict__ cat, float* __restrict__ hbuf,
                                                            float* __restrict__ gpre, float* __restrict__ gX) {
    extern __shared__ float w[];                  // W5 [H, 2c] | b5 [H] | W6 [2, H]
    float* sW5 = w;
    float* sb5 = w + H * 2 * c;
    float* sW6 = sb5 + H;
    for (int k = threadIdx.x; k < H * 2 * c; k += blockDim.x) sW5[k] = W5[k];
    for (int k = threadIdx.x; k < H; k += blockDim.x) sb5[k] = b5[k];
    for (int k = threadIdx.x; k < 2 * H; k += blockDim.x) sW6[k] = W6[k];
    __syncthreads();
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const int u = src[e], v = dst[e];
    const float* xs = X + (size_t)u * c;
    const float* xt = X + (size_t)v * c;
    float pre[H];
#pragma unroll
    for (int h = 0; h < H; ++h) pre[h] = sb5[h];
    for (int k = 0; k < c; ++k) {
        const float a = xs[k], b = xt[k];
        cat[(size_t)e * 2 * c + k] = a; cat[(size_t)e * 2 * c + c + k] = b;
#pragma unroll
        for (int h = 0; h < H; ++h) pre[h] += sW5[h * 2 * c + k] * a + sW5[h * 2 * c + c + k] * b;
    }
    const float g0 = Gpd[2 * e], g1 = Gpd[2 * e + 1];
#pragma unroll
    for (int h = 0; h < H; ++h) {
        const float hv = pre[h] > 0.f ? pre[h] : slope * pre[h];
        const float gh = g0 * sW6[h] + g1 * sW6[H + h];
        const float gp = gh * (pre[h] > 0.f ? 1.f : slope);
        hbuf[(size_t)e * H + h] = hv;
        gpre[(size_t)e * H + h] = gp;
        pre[h] = gp;
    }
    for (int k = 0; k < c; ++k) {
        float ga = 0.f, gb = 0.f;
#pragma unroll
        for (int h = 0; h < H; ++h) { ga += pre[h] * sW5[h * 2 * c + k]; gb += pre[h] * sW5[h * 2 * c + c + k]; }
        atomicAdd(&gX[(size_t)u * c + k], ga);
        atomicAdd(&gX[(size_t)v * c + k], gb);
    }
}

}  // namespace

// d_work: float32[n * (8 * c_out + 5) + 2 * c_out * c_out + c_in * c_out + 4 + c_out * (2 * c_out + 4)] scratch, 16-byte aligned.  d_out: the layer's forward output (only read when prelu_slope >= 0: the sign of the
// output is the sign of the pre-activation).  d_gX may be null (first layer: the input is data).
extern "C" int tlc_gat_layer_bwd(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_src, const float* d_X, int32_t c_in,
                                 int32_t c_out, const float* d_Wl, const float* d_att, const float* d_Wij, float prelu_slope,
                                 const float* d_out, const float* d_gout, float* d_gX, float* d_gWl, float* d_gatt, float* d_gWij,
                                 float* d_gbias, float* d_work, void* stream) {
    TLC_REQUIRE(n_nodes >= 0 && c_in > 0, "bad sizes");
    TLC_REQUIRE(c_out == 8 || c_out == 16 || c_out == 32 || c_out == 64, "c_out must be 8, 16, 32 or 64");
    TLC_REQUIRE(d_rowptr && d_src && d_X && d_Wl && d_att && d_Wij && d_gout && d_gWl && d_gatt && d_gWij && d_gbias && d_work, "null pointer");
    TLC_REQUIRE(prelu_slope < 0.f || d_out != nullptr, "out is needed for the fused PReLU");
    TLC_REQUIRE(c_in <= 64, "c_in <= 64");
    hipStream_t s = (hipStream_t)stream;
    if (n_nodes == 0) {
        TLC_HIP_CHECK(hipMemsetAsync(d_gWl, 0, (size_t)c_out * c_in * sizeof(float), s));
        TLC_HIP_CHECK(hipMemsetAsync(d_gatt, 0, (size_t)c_out * sizeof(float), s));
        TLC_HIP_CHECK(hipMemsetAsync(d_gWij, 0, (size_t)c_out * 2 * c_out * sizeof(float), s));
        TLC_HIP_CHECK(hipMemsetAsync(d_gbias, 0, (size_t)2 * c_out * sizeof(float), s));
        return TLC_OK;
    }
    switch (c_out) {
        case 8: return launch_gat_bwd<8>(n_nodes, d_rowptr, d_src, d_X, c_in, d_Wl, d_att, d_Wij, prelu_slope, d_out, d_gout, d_gX, d_gWl, d_gatt, d_gWij, d_gbias, d_work, s);
        case 16: return launch_gat_bwd<16>(n_nodes, d_rowptr, d_src, d_X, c_in, d_Wl, d_att, d_Wij, prelu_slope, d_out, d_gout, d_gX, d_gWl, d_gatt, d_gWij, d_gbias, d_work, s);
        case 32: return launch_gat_bwd<32>(n_nodes, d_rowptr, d_src, d_X, c_in, d_Wl, d_att, d_Wij, prelu_slope, d_out, d_gout, d_gX, d_gWl, d_gatt, d_gWij, d_gbias, d_work, s);
        default: return launch_gat_bwd<64>(n_nodes, d_rowptr, d_src, d_X, c_in, d_Wl, d_att, d_Wij, prelu_slope, d_out, d_gout, d_gX, d_gWl, d_gatt, d_gWij, d_gbias, d_work, s);
    }
}

// d_work: float32[n_edges * (2 * c + 2 * hidden)] scratch.  d_gX [n_nodes, c] is ADDED to (zero it first).
extern "C" int tlc_edge_head_bwd(int64_t n_edges, const int32_t* d_src, const int32_t* d_dst, const float* d_X, int32_t c,
                                 const float* d_W5, const float* d_b5, int32_t hidden, float prelu_slope, const float* d_W6,
                                 const float* d_gpd, float* d_gX, float* d_gW5, float* d_gb5, float* d_gW6, float* d_gb6,
                                 float* d_work, void* stream) {
    TLC_REQUIRE(n_edges >= 0 && c > 0, "bad sizes");
    TLC_REQUIRE(hidden == 16 || hidden == 32 || hidden == 64, "hidden must be 16, 32 or 64");
    TLC_REQUIRE(d_gW5 && d_gb5 && d_gW6 && d_gb6, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    TLC_HIP_CHECK(hipMemsetAsync(d_gW5, 0, (size_t)hidden * 2 * c * sizeof(float), s));
    TLC_HIP_CHECK(hipMemsetAsync(d_gb5, 0, (size_t)hidden * sizeof(float), s));
    TLC_HIP_CHECK(hipMemsetAsync(d_gW6, 0, (size_t)2 * hidden * sizeof(float), s));
    TLC_HIP_CHECK(hipMemsetAsync(d_gb6, 0, (size_t)2 * sizeof(float), s));
    if (n_edges == 0) return TLC_OK;
    TLC_REQUIRE(d_src && d_dst && d_X && d_W5 && d_b5 && d_W6 && d_gpd && d_gX && d_work, "null pointer");
    float* cat = d_work;
    float* hbuf = cat + (size_t)n_edges * 2 * c;
    float* gpre = hbuf + (size_t)n_edges * hidden;
    const size_t lds = ((size_t)hidden * 2 * c + 3 * (size_t)hidden) * sizeof(float);
    TLC_REQUIRE(lds <= 64 * 1024, "edge head weights do not fit LDS");
    const dim3 grid((unsigned)((n_edges + 127) / 128));
    if (hidden == 16) hipLaunchKernelGGL((edge_head_bwd_kernel<16>), grid, dim3(128), lds, s, (long long)n_edges, d_src, d_dst, d_X, c, d_W5, d_b5, prelu_slope, d_W6, d_gpd, cat, hbuf, gpre, d_gX);
    else if (hidden == 32) hipLaunchKernelGGL((edge_head_bwd_kernel<32>), grid, dim3(128), lds, s, (long long)n_edges, d_src, d_dst, d_X, c, d_W5, d_b5, prelu_slope, d_W6, d_gpd, cat, hbuf, gpre, d_gX);
    else hipLaunchKernelGGL((edge_head_bwd_kernel<64>), grid, dim3(128), lds, s, (long long)n_edges, d_src, d_dst, d_X, c, d_W5, d_b5, prelu_slope, d_W6, d_gpd, cat, hbuf, gpre, d_gX);
    TLC_HIP_CHECK(hipGetLastError());
    int rc;
    if ((rc = xty(n_edges, gpre, hidden, hidden, cat, 2 * c, 2 * c, d_gW5, s)) != TLC_OK) return rc;      // d W5 = gpre^T cat
    if ((rc = xty(n_edges, d_gpd, 2, 2, hbuf, hidden, hidden, d_gW6, s)) != TLC_OK) return rc;              // d W6 = gpd^T h
    hipLaunchKernelGGL(colsum_kernel, dim3(1024), dim3(256), 0, s, (long long)n_edges, (const float*)gpre, hidden, d_gb5);
    hipLaunchKernelGGL(colsum_kernel, dim3(1024), dim3(256), 0, s, (long long)n_edges, d_gpd, 2, d_gb6);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

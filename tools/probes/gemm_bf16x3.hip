// PROBE, NOT BUILT INTO THE LIBRARY (round 3).  The first-layer projection as three-piece bf16 products on the bf16 matrix cores.
// Numerically it does what it says (tests/test_gpu_lp_forward.py passed with it wired in: 38/38; max error 8e-7 of max |C| against
// 5e-7 for the f32 MFMA kernel), but it is SLOWER than the f32 kernel of lp_forward.hip on PubMed's 19 717 x 500 x 100:
//   f32 MFMA kernel                                             30.3 us
//   B in registers (192 VGPRs), A through LDS                   31.6 us  (VGPR limit: 55 spills once the loads were made branch-free)
//   this file: A in registers, B pre-cut into a workspace       31.9 us  (+ 1.5-5 us for the cutting kernel)
// Why (per-step stamps, one wavefront per SIMD because M / 16 = 1 233 row tiles meet 1 024 SIMDs): a k step of 42 MFMAs (672
// cycles) takes 0.85 us = 2 000 cycles -- the 21 ds_read_b128 of the step's B fragments are waited for in front of the MFMAs
// and nothing else runs on the SIMD; interleaving the accumulators did not change it (it is not MFMA dependency); the C store
// at the end costs another 3.9 us.  A later ablation of a fourth variant (A in a ring of six steps, all of a step's fragments read
// before its products; 29.9 us) with rocprofv3 kernel times on a one-round launch (64 workgroups): everything 23.2 us; without
// the MFMAs 19.8; without the B copies, the barriers and the A ring 15.7; without the MFMAs as well 12.8 -- i.e. the 672 MFMAs of
// a wavefront cost 3 us, the per-step copy + barrier 7 us, and 11-13 us are launch, the first HBM round trip, the piece cutting
// and the C store drain of a kernel that lives for one round.  The f32 kernel pays the same fixed costs inside its 30 us; a
// bf16-pieces kernel only wins if those are overlapped (several short-lived workgroups per CU), which PubMed's 1 233 row tiles
// do not give with B shared through LDS.  Two workgroups per CU overlap perfectly (two waves per SIMD: same time for twice the rows), so
// the kernel needs either (row tile, column half) wavefronts -- 2 466 of them, two per SIMD -- or the next group's fragments
// prefetched under the current group's MFMAs (36 more VGPRs: past 256 with the whole A row tile in registers).  Estimated
// 15-17 us with either; not done.  What cost most before that: cutting B in every workgroup (VALU: 1.8 us per step), loads behind
// branches (waited for at the join), arrays of HIP's uint4 struct (kept in scratch; native ext_vector_type arrays are not).
//
// gemm_bf16x3.hip -- the feature projection  C[M, N] = A[M, K] @ B[K, N] (+ bias)(ReLU)  of TLCGNN's first GCN layer
// (baselines/TLCGNN.py:23: GCNConv(F -> 100); x @ W of Knowledge_Distillation/PD_conv.py:179-181) on the bf16 matrix cores,
// with f32 accuracy.
//
// gfx950 has no reduced-precision f32 matrix instruction, and the f32 MFMA runs at 1/16 of the bf16 rate (157 against
// 2 500 TFLOP/s); the f32 kernel of lp_forward.hip reaches 41 % of that peak: 30 us for PubMed's 19 717 x 500 x 100.
// Here every f32 operand is cut into THREE bf16 pieces by truncation,  x = x1 + x2 + x3  EXACTLY (8 + 8 + 8 significant bits;
// each remainder is an exact f32 subtraction), and the product is summed over the six piece pairs of weight >= 2^-16:
//     x w  ~  x1 w1 + x1 w2 + x2 w1 + x1 w3 + x3 w1 + x2 w2          (dropped: x2 w3 + x3 w2 + x3 w3 <= 3 * 2^-24 |x w|)
// Every piece product is exact in f32 (8 x 8 bits) and the matrix core accumulates in f32: the result differs from an f32 dot
// product by a few 2^-24 of sum |x_k w_k| -- the size of that dot product's own rounding error, two orders below the 1e-5
// parity bound.  Six bf16 MFMAs replace one f32 MFMA of the same shape at 16 x the rate.
//
// Layout of the work.  A workgroup of four wavefronts owns 64 rows of C; a wavefront owns one 16-row tile and ALL column tiles
// (NT <= 8 of 16 columns): NT accumulators of four registers.  K (<= 512) is walked in steps of 32:
//   * A: a lane's operand of a step is eight consecutive k of one row (two 16-byte loads).  PubMed has 1 233 row tiles for 1 024
//     SIMDs -- one wavefront per SIMD, nothing to hide a load behind -- so a wavefront asks for its WHOLE row tile (16 x K f32 =
//     128 registers) before the first product and consumes it in the order it arrives; the pieces are cut step by step;
//   * B: cut once, by a small kernel of its own, into a workspace in MFMA operand order ([step][tile][piece][lane] of 16 bytes:
//     344 KB for 500 x 112) -- cutting it in every workgroup cost more vector instructions per step than the products' 42 MFMAs
//     (1.8 us per step with one wavefront per SIMD and nothing to overlap with); a workgroup copies the step's 21 KB block into
//     LDS (two buffers, the blocks of the next two steps in flight in registers) and its four wavefronts read it from there
//     with conflict-free ds_read_b128;
//   * per step and wavefront: NT x (3 LDS reads + 6 MFMAs).
// One barrier per step.
#include "tlc_common.h"

#ifdef TLC_GX_STAMPS
__device__ unsigned long long gx_dbg[64];
#define GX_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) gx_dbg[(k)] = wall_clock64(); } while (0)
extern "C" int tlc_debug_gx_stamps(unsigned long long* h_out) { return hipMemcpyFromSymbol(h_out, HIP_SYMBOL(gx_dbg), sizeof(gx_dbg)) == hipSuccess ? 0 : 2; }
#else
#define GX_STAMP(k) do { } while (0)
#endif

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;      // (native vector: arrays of HIP's uint4 struct stay in scratch memory)

// x = h + m + l exactly (truncating cuts; the remainders are exact f32 subtractions)
__device__ __forceinline__ void cut3(float x, unsigned& h, unsigned& m, unsigned& l) {
    const unsigned uh = __float_as_uint(x) & 0xffff0000u;
    const float r = x - __uint_as_float(uh);
    const unsigned um = __float_as_uint(r) & 0xffff0000u;
    const float r2 = r - __uint_as_float(um);
    h = uh >> 16; m = um >> 16; l = __float_as_uint(r2) >> 16;
}
// eight floats -> three fragments of eight bf16 (element j in the low / high half of word j / 2)
__device__ __forceinline__ void cut3x8(const float (&x)[8], uint4& H, uint4& Mi, uint4& L) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) cut3(x[j], h[j], m[j], l[j]);
    H = make_uint4(h[0] | (h[1] << 16), h[2] | (h[3] << 16), h[4] | (h[5] << 16), h[6] | (h[7] << 16));
    Mi = make_uint4(m[0] | (m[1] << 16), m[2] | (m[3] << 16), m[4] | (m[5] << 16), m[6] | (m[7] << 16));
    L = make_uint4(l[0] | (l[1] << 16), l[2] | (l[3] << 16), l[4] | (l[5] << 16), l[6] | (l[7] << 16));
}
__device__ __forceinline__ f32x4 mma(const uint4& a, const uint4& b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf8, a), __builtin_bit_cast(bf8, b), acc, 0, 0, 0);
}

constexpr int GX_WAVES = 4;                                // wavefronts per workgroup, one 16-row tile each
constexpr int GX_ROWS = GX_WAVES * 16;                     // rows of C per workgroup
constexpr int GX_KS = 16;                                  // k steps held in registers: K <= 512

// NT column tiles; VEC: rows of A are 16-byte aligned and K % 4 == 0 (else scalar loads)
template <int NT, bool VEC>
__global__ __launch_bounds__(GX_WAVES * 64) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemm_bf16x3_kernel(int M, int N, int K, const float* __restrict__ A,
                                                                    const uint4* __restrict__ Bw, const float* __restrict__ bias, int relu,
                                                                    float* __restrict__ C) {
    extern __shared__ uint4 b_lds[];                       // [2][NT][3][64]
    constexpr int W = GX_WAVES * 64;
    constexpr int ITEMS = NT * 3 * 64;                     // uint4 of a step's block of B
    constexpr int PER = (ITEMS + W - 1) / W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_steps = (K + 31) >> 5;
    const int row0 = (int)blockIdx.x * GX_ROWS + wave * 16;
    GX_STAMP(0);

    // ---- B: the step's block of the workspace ([NT][3][64] uint4) -> LDS, PER uint4 per thread, two steps in flight
    u32x4 br0[PER], br1[PER];                              // (two arrays, not [2][PER]: see the note on A below)
    auto load_b = [&](int s, u32x4 (&dst)[PER]) __attribute__((always_inline)) {
        const u32x4* blk = reinterpret_cast<const u32x4*>(Bw) + (size_t)s * ITEMS;
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int item = tid + q * W;
            dst[q] = blk[item < ITEMS ? item : ITEMS - 1];
        }
    };
    auto store_b = [&](int buf, const u32x4 (&srcv)[PER]) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < PER; ++q) {
            const int item = tid + q * W;
            if (item < ITEMS) reinterpret_cast<u32x4*>(b_lds)[(size_t)buf * ITEMS + item] = srcv[q];
        }
    };
    load_b(0, br0);
    if (n_steps > 1) load_b(1, br1);

    // ---- A: lane (r = lane & 15, g = lane >> 4) reads A[row0 + r][32 s + 8 g .. + 8) for every step s, all at once.  Addresses are
    // clamped into the matrix: a row >= M only feeds rows of C that are never stored, and k >= K meets rows of B that are zero.
    // (Sixteen separate 8-float arrays, not one [16][8]: the compiler keeps an array of that size in scratch memory.)
    int arow = row0 + (lane & 15);
    arow = arow < M ? arow : M - 1;
    const float* abase = A + (size_t)arow * K;
    auto load_a = [&](int s, float (&x)[8]) __attribute__((always_inline)) {
        const int k0 = 32 * s + 8 * (lane >> 4);
        if (VEC) {
            const int ka = k0 + 4 <= K ? k0 : K - 4, kb = k0 + 8 <= K ? k0 + 4 : K - 4;
            const float4 x0 = *reinterpret_cast<const float4*>(abase + ka), x1 = *reinterpret_cast<const float4*>(abase + kb);
            x[0] = x0.x; x[1] = x0.y; x[2] = x0.z; x[3] = x0.w;
            x[4] = x1.x; x[5] = x1.y; x[6] = x1.z; x[7] = x1.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const int k = k0 + j; x[j] = abase[k < K ? k : K - 1]; }
        }
    };
#define GX_ALL(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(8) F(9) F(10) F(11) F(12) F(13) F(14) F(15)
#define GX_LOAD(s) float ar##s[8]; load_a(s, ar##s);
    GX_ALL(GX_LOAD)
#undef GX_LOAD
    static_assert(GX_KS == 16, "GX_ALL lists sixteen steps");
    f32x4 acc[NT];
#pragma unroll
    for (int c = 0; c < NT; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};

    store_b(0, br0);
    if (n_steps > 2) load_b(2, br0);
    __syncthreads();
    GX_STAMP(1);
    auto step = [&](int s, const float (&x)[8], u32x4 (&bnext)[PER]) __attribute__((always_inline)) {
        const int buf = s & 1;
        uint4 ah, am, al;
        cut3x8(x, ah, am, al);
        GX_STAMP(32 + s);
        const uint4* src = b_lds + (size_t)buf * NT * 3 * 64 + lane;
        // Tiles in groups of four, the six piece products of a group interleaved over its accumulators: a wavefront is alone on
        // its SIMD here, and six back-to-back MFMAs into ONE accumulator wait for each other (measured: 1.4 us per step that way)
        constexpr int GRP = 4;
#pragma unroll
        for (int c0 = 0; c0 < NT; c0 += GRP) {
            uint4 bh[GRP], bm[GRP], bl[GRP];
#pragma unroll
            for (int c = 0; c < GRP; ++c)
                if (c0 + c < NT) { bh[c] = src[((c0 + c) * 3 + 0) * 64]; bm[c] = src[((c0 + c) * 3 + 1) * 64]; bl[c] = src[((c0 + c) * 3 + 2) * 64]; }
#pragma unroll
            for (int c = 0; c < GRP; ++c) if (c0 + c < NT) acc[c0 + c] = mma(al, bh[c], acc[c0 + c]);
#pragma unroll
            for (int c = 0; c < GRP; ++c) if (c0 + c < NT) acc[c0 + c] = mma(ah, bl[c], acc[c0 + c]);
#pragma unroll
            for (int c = 0; c < GRP; ++c) if (c0 + c < NT) acc[c0 + c] = mma(am, bm[c], acc[c0 + c]);
#pragma unroll
            for (int c = 0; c < GRP; ++c) if (c0 + c < NT) acc[c0 + c] = mma(am, bh[c], acc[c0 + c]);
#pragma unroll
            for (int c = 0; c < GRP; ++c) if (c0 + c < NT) acc[c0 + c] = mma(ah, bm[c], acc[c0 + c]);
#pragma unroll
            for (int c = 0; c < GRP; ++c) if (c0 + c < NT) acc[c0 + c] = mma(ah, bh[c], acc[c0 + c]);
        }
        // the next step's block (asked for two steps ago) goes to the other buffer; its registers take the block after the next
        if (s + 1 < n_steps) {
            store_b(buf ^ 1, bnext);
            if (s + 3 < n_steps) load_b(s + 3, bnext);
        }
        __syncthreads();
        GX_STAMP(8 + s);
    };
#define GX_STEP(s) if (s < n_steps) step(s, ar##s, ((s + 1) & 1) ? br1 : br0);
    GX_ALL(GX_STEP)
#undef GX_STEP
#undef GX_ALL
    GX_STAMP(2);
    // C[row0 + 4 (lane >> 4) + i][16 c + (lane & 15)]
#pragma unroll
    for (int c = 0; c < NT; ++c) {
        const int col = 16 * c + (lane & 15);
        if (col < N) {
            const float bc = bias ? bias[col] : 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = row0 + 4 * (lane >> 4) + i;
                float v = acc[c][i] + bc;
                if (relu) v = v > 0.f ? v : 0.f;
                if (row < M) C[(size_t)row * N + col] = v;
            }
        }
    }
    GX_STAMP(3);
}

// B [K, N] f32 -> workspace [n_steps][NT][3][64] uint4: thread (step s, tile c, lane l) cuts B[32 s + 8 (l >> 4) + j][16 c + (l & 15)]
__global__ __launch_bounds__(64) void gemm_bf16x3_cut_b_kernel(int N, int K, int NT, const float* __restrict__ B, uint4* __restrict__ Bw) {
    const int s = blockIdx.x / NT, c = blockIdx.x % NT, l = threadIdx.x;
    const int col = 16 * c + (l & 15);
    const int cc = col < N ? col : N - 1;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 32 * s + 8 * (l >> 4) + j;
        v[j] = B[(size_t)(k < K ? k : K - 1) * N + cc];
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) if (32 * s + 8 * (l >> 4) + j >= K || col >= N) v[j] = 0.f;
    uint4 H, Mi, L;
    cut3x8(v, H, Mi, L);
    uint4* dst = Bw + ((size_t)(s * NT + c) * 3) * 64 + l;
    dst[0] = H; dst[64] = Mi; dst[128] = L;
}

template <int NT>
static hipError_t launch_n(int M, int N, int K, const float* A, const float* B, uint4* Bw, const float* bias, int relu, float* C, hipStream_t s) {
    const int n_steps = (K + 31) / 32;
    hipLaunchKernelGGL(gemm_bf16x3_cut_b_kernel, dim3(n_steps * NT), dim3(64), 0, s, N, K, NT, B, Bw);
    const size_t lds = (size_t)2 * NT * 3 * 64 * sizeof(uint4);
    const bool vec = (K % 4) == 0 && (reinterpret_cast<uintptr_t>(A) & 15) == 0;
    const int grid = (M + GX_ROWS - 1) / GX_ROWS;
    if (vec) hipLaunchKernelGGL((gemm_bf16x3_kernel<NT, true>), dim3(grid), dim3(GX_WAVES * 64), lds, s, M, N, K, A, (const uint4*)Bw, bias, relu, C);
    else hipLaunchKernelGGL((gemm_bf16x3_kernel<NT, false>), dim3(grid), dim3(GX_WAVES * 64), lds, s, M, N, K, A, (const uint4*)Bw, bias, relu, C);
    return hipGetLastError();
}

}  // namespace

// Workspace bytes for the cut copy of B, or 0 if the shape is not one this kernel is built for (many rows, 128 <= K <= 512,
// 64 < N <= 128): the caller then runs the f32 kernel.
size_t tlc_gemm_bf16x3_work_bytes(int M, int N, int K) {
    if (M < 4096 || K < 128 || K > 32 * GX_KS || N <= 64 || N > 128) return 0;
    return (size_t)((K + 31) / 32) * ((N + 15) / 16) * 3 * 64 * sizeof(uint4);
}

int tlc_gemm_bf16x3(int M, int N, int K, const float* d_A, const float* d_B, const float* d_bias, int relu, float* d_C, void* d_work,
                    void* stream) {
    hipStream_t s = (hipStream_t)stream;
    uint4* Bw = (uint4*)d_work;
    hipError_t e;
    switch ((N + 15) / 16) {
        case 5: e = launch_n<5>(M, N, K, d_A, d_B, Bw, d_bias, relu, d_C, s); break;
        case 6: e = launch_n<6>(M, N, K, d_A, d_B, Bw, d_bias, relu, d_C, s); break;
        case 7: e = launch_n<7>(M, N, K, d_A, d_B, Bw, d_bias, relu, d_C, s); break;
        default: e = launch_n<8>(M, N, K, d_A, d_B, Bw, d_bias, relu, d_C, s); break;
    }
    TLC_HIP_CHECK(e);
    return TLC_OK;
}

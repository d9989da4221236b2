// lp_forward.hip -- M1..M3: the TLCGNN link-prediction forward (baselines/TLCGNN.py:19-62).
//
//   tlc_gcn_norm_csr     gcn_norm of GCNConv(cached=True)         (spec in-tree: Knowledge_Distillation/PD_conv.py:35-70)
//   tlc_gemm_f32         x @ W on the f32 MFMA (v_mfma_f32_16x16x4_f32), bias/ReLU fused   (PD_conv.py:179-181)
//   tlc_spmm_csr_f32     propagate = normalised scatter-add at the target, as a row-owned CSR SpMM,
//                        bias + ReLU fused (PD_conv.py:183-188; message_passing.py:275-293 aggr='add')
//   tlc_renorm_rows_f32  emb.renorm_(2, 0, 1)                      (TLCGNN.py:48)
//   tlc_lp_decode_fused  gather, (a-b)^2 || PI, Linear(41->25), LeakyReLU, Linear(25->1), |.|, clamp, Fermi-Dirac
//                        in one pass per pair                      (TLCGNN.py:52-61)
//
// fp32 throughout (the 1e-5 parity bound rules out bf16 MFMA inputs); the scatter-add is restated as a
// gather over a CSR sorted by target so that every output row has one owner: no atomics, bitwise
// reproducible, and the bias/activation fuse into the same pass.
#include "tlc_common.h"
#include <stdlib.h>


// ======================================================================================================================
// gcn_norm -> CSR by target
// ======================================================================================================================
namespace {

// (the count an edge's atomic returns is its place in its row: kept per edge, so that the fill needs no second round of atomics)
__global__ void gcn_count_kernel(long long n_edges, const long long* __restrict__ ei, int n_nodes, int* __restrict__ cnt, int* __restrict__ place) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    const long long r = ei[e], c = ei[n_edges + e];
    if (r != c && r >= 0 && c >= 0 && r < n_nodes && c < n_nodes) place[e] = atomicAdd(&cnt[c], 1);   // self loops are re-added once per node
}

// rowptr[i+1] = sum_{j<=i} (cnt[j] + 1): single workgroup, chunks of 1024 with a running carry
__global__ __launch_bounds__(1024) void gcn_scan_kernel(int n_nodes, const int* __restrict__ cnt, int* __restrict__ rowptr,
                                                        int* __restrict__ nnz_out) {
    __shared__ int s[1024];
    __shared__ int carry;
    const int t = threadIdx.x;
    if (t == 0) { carry = 0; rowptr[0] = 0; }
    __syncthreads();
    for (int base = 0; base < n_nodes; base += 1024) {
        const int i = base + t;
        const int v = i < n_nodes ? cnt[i] + 1 : 0;
        s[t] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int a = t >= o ? s[t - o] : 0;
            __syncthreads();
            s[t] += a;
            __syncthreads();
        }
        if (i < n_nodes) rowptr[i + 1] = carry + s[t];
        __syncthreads();
        if (t == 1023) carry += s[t];
        __syncthreads();
    }
    if (t == 0 && nnz_out) *nnz_out = carry;
}

// The same prefix for graphs beyond a few thousand nodes (a PDGNN batch of 41 127 molecule graphs has a million: the single
// workgroup above took 1.75 ms, longer than the four-layer forward): blocks of 1024 scan locally and leave their totals, one
// workgroup scans the totals, the blocks add their offsets.
__global__ __launch_bounds__(1024) void gcn_scan_block_kernel(int n_nodes, const int* __restrict__ cnt, int* __restrict__ rowptr,
                                                              int* __restrict__ block_sum) {
    __shared__ int s[1024];
    const int t = threadIdx.x, i = blockIdx.x * 1024 + t;
    s[t] = i < n_nodes ? cnt[i] + 1 : 0;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int a = t >= o ? s[t - o] : 0;
        __syncthreads();
        s[t] += a;
        __syncthreads();
    }
    if (i < n_nodes) rowptr[i + 1] = s[t];
    if (t == 1023) block_sum[blockIdx.x] = s[t];
}
__global__ __launch_bounds__(1024) void gcn_scan_top_kernel(int n_blocks, int* __restrict__ block_sum, int* __restrict__ rowptr,
                                                            int* __restrict__ nnz_out) {
    __shared__ int s[1024];
    __shared__ int carry;
    const int t = threadIdx.x;
    if (t == 0) { carry = 0; rowptr[0] = 0; }
    __syncthreads();
    for (int base = 0; base < n_blocks; base += 1024) {
        const int i = base + t;
        const int v = i < n_blocks ? block_sum[i] : 0;
        s[t] = v;
        __syncthreads();
        for (int o = 1; o < 1024; o <<= 1) {
            const int a = t >= o ? s[t - o] : 0;
            __syncthreads();
            s[t] += a;
            __syncthreads();
        }
        if (i < n_blocks) block_sum[i] = carry + s[t] - v;             // exclusive prefix of the block totals
        __syncthreads();
        if (t == 1023) carry += s[t];
        __syncthreads();
    }
    if (t == 0 && nnz_out) *nnz_out = carry;
}
__global__ __launch_bounds__(1024) void gcn_scan_add_kernel(int n_nodes, int* __restrict__ rowptr, const int* __restrict__ block_sum) {
    const int i = blockIdx.x * 1024 + threadIdx.x;
    if (i < n_nodes) rowptr[i + 1] += block_sum[blockIdx.x];
}

__global__ void gcn_fill_kernel(long long n_edges, const long long* __restrict__ ei, int n_nodes,
                                const int* __restrict__ rowptr, const int* __restrict__ place, int* __restrict__ col, int* __restrict__ long_count) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n_edges) return;
    if (e == 0) *long_count = 0;                            // (the list of long rows starts empty: see gcn_csr_launch)
    const long long r = ei[e], c = ei[n_edges + e];
    if (r != c && r >= 0 && c >= 0 && r < n_nodes && c < n_nodes) col[rowptr[c] + place[e]] = (int)r;
}

// rows of at most GCN_SHORT_ROW entries (all of a molecule batch, nearly all of a citation graph): one THREAD per row, the
// same rank sort in registers -- a wavefront per three-entry row made the million-node PDGNN batch's build 0.2 ms
#define GCN_SHORT_ROW 8
__global__ __launch_bounds__(256) void gcn_finish_short_kernel(int n_nodes, const int* __restrict__ rowptr, const int* __restrict__ raw,
                                                               int* __restrict__ col, int* __restrict__ long_rows, int long_cap) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int b = c < n_nodes ? rowptr[c] : 0, d = c < n_nodes ? rowptr[c + 1] - b : 0;
    // the longer rows go on a list for gcn_finish_kernel (long_rows[0]: their number; one atomic per workgroup that has any): a
    // wavefront per ROW that only finds out that its row is short was 58 us on a million-node molecule batch without one long row
    {
        __shared__ int s_cnt[4], s_base;
        const bool lng = d > GCN_SHORT_ROW;
        const unsigned long long m = __ballot(lng);
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        if (lane == 0) s_cnt[wave] = __popcll(m);
        __syncthreads();
        const int total = s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3];
        if (total) {                                            // (uniform over the workgroup)
            if (threadIdx.x == 0) s_base = atomicAdd(&long_rows[0], total);
            __syncthreads();
            if (lng) {
                int pos = s_base + __popcll(m & ((1ull << lane) - 1ull));
                for (int w = 0; w < wave; ++w) pos += s_cnt[w];
                if (pos < long_cap) long_rows[1 + pos] = c;
            }
        }
    }
    if (c >= n_nodes || d > GCN_SHORT_ROW) return;
    int v[GCN_SHORT_ROW];
#pragma unroll
    for (int i = 0; i < GCN_SHORT_ROW; ++i) v[i] = i < d - 1 ? raw[b + i] : c;      // entry d-1 is the self loop
#pragma unroll
    for (int i = 0; i < GCN_SHORT_ROW; ++i) {
        if (i < d) {
            int rank = 0;
#pragma unroll
            for (int j = 0; j < GCN_SHORT_ROW; ++j)
                if (j < d) rank += (v[j] < v[i]) || (v[j] == v[i] && j < i);
            col[b + rank] = v[i];
        }
    }
}

// one wave per target row: the self loop joins the sources and the row is rank-sorted ascending out of place (every
// lane counts the entries below its own; hub rows would serialise an in-place sort on the global-memory latency)
__global__ __launch_bounds__(256) void gcn_finish_kernel(int n_nodes, const int* __restrict__ rowptr, const int* __restrict__ raw,
                                                         int* __restrict__ col, const int* __restrict__ long_rows, int long_cap) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, n_waves = (gridDim.x * 256) >> 6, lane = threadIdx.x & 63;
    const int n_long = long_rows[0];
    const bool listed = n_long <= long_cap;                // (more long rows than the list holds: every row is looked at)
    const int n_items = listed ? n_long : n_nodes;
    for (int it = wave; it < n_items; it += n_waves) {
        const int c = listed ? long_rows[1 + it] : it;
        const int b = rowptr[c], d = rowptr[c + 1] - b;     // entry d-1 is the self loop (add_remaining_self_loops, weight 1)
        if (d <= GCN_SHORT_ROW) continue;                   // short rows: one thread each, gcn_finish_short_kernel
        for (int i = lane; i < d; i += 64) {
            const int vi = i < d - 1 ? raw[b + i] : c;
            int rank = 0;
            for (int j = 0; j < d; ++j) {
                const int vj = j < d - 1 ? raw[b + j] : c;
                rank += (vj < vi) || (vj == vi && j < i);
            }
            col[b + rank] = vi;
        }
    }
}
__global__ void gcn_val_kernel(int n_nodes, const int* __restrict__ rowptr, const int* __restrict__ col, float* __restrict__ val) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_nodes) return;
    const int b = rowptr[c], e = rowptr[c + 1];
    // deg = scatter_add(w, col) over targets (PD_conv.py:66); deg^-1/2, inf -> 0 (:67-68)
    const float dc = 1.0f / sqrtf((float)(e - b));
    for (int j = b; j < e; ++j) {
        const int r = col[j];
        const float dr = 1.0f / sqrtf((float)(rowptr[r + 1] - rowptr[r]));
        val[j] = dr * 1.0f * dc;                       // deg_inv_sqrt[row] * edge_weight * deg_inv_sqrt[col] (:69)
    }
}

// ======================================================================================================================
// f32 MFMA GEMM, 16x16x4 tiles: C[M,N] = A[M,K] @ B[K,N] (+bias)(ReLU).  N <= 128.
// Workgroup = 80 rows x NT*16 columns (PubMed: 19717 rows -> 247 workgroups, one per CU in a single round; N=100 pads to
// 112, not 128).  Wave w of a group of four owns row tile w across all NT column tiles; the fifth row tile is split by
// columns over the four waves.  KS such groups split every K chunk between them (two waves per SIMD) and are summed
// through LDS at the end.  K runs in chunks of KC through double-buffered LDS with two chunks of global loads in flight
// in registers (one chunk of MFMAs is shorter than an HBM round trip).
// k is permuted inside every 16-block (lane group g = lane/16 takes k = 4g+s at step s; a sum over k does not care), so a
// lane's four A operands are contiguous: one ds_read_b128 from rows of stride KC+8 words (conflict-free for b128's lane
// groups); B stays row-major with stride NP+4 (4 rows apart = 16 banks apart for the two halves of a ds_read_b32).
// Loads are buffer loads: rows past M, k past K and columns past N fall outside the descriptor and read as zeros, so the
// pipeline has no branches and the compiler can count the loads in flight (vmcnt(n) instead of vmcnt(0)).
// ======================================================================================================================
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
#define G16_BM 80

template <int NT, int KC>
struct G16Layout {
    static constexpr int NP = NT * 16, SB = NP + 4, AS = KC + 8;
    static constexpr int A_WORDS = G16_BM * AS, B_WORDS = KC * SB;
    static constexpr int STAGE_WORDS = 2 * (A_WORDS + B_WORDS), C_WORDS = G16_BM * SB;
    static constexpr int LDS_BYTES = 4 * (STAGE_WORDS > C_WORDS ? STAGE_WORDS : C_WORDS);
};

template <int NT, bool VEC, int KS, int KC>
__global__ __launch_bounds__(256 * KS) void gemm16_f32_kernel(int M, int N, int K, const float* __restrict__ A,
                                                              const float* __restrict__ B, const float* __restrict__ bias, int relu,
                                                              float* __restrict__ C) {
    using L = G16Layout<NT, KC>;
    constexpr int TH = 256 * KS;
    constexpr int NP = L::NP, SB = L::SB, AS = L::AS;
    constexpr int XT = (NT + 3) / 4;                        // column tiles of the fifth row tile per wave
    constexpr int NB = NT + XT;
    constexpr int A4 = G16_BM * KC / 4, B4 = KC * NP / 4;   // float4 slots of the A and B tiles of one chunk
    constexpr int AQ = (A4 + TH - 1) / TH, BQ = (B4 + TH - 1) / TH;
    constexpr int A_WORDS = L::A_WORDS, B_WORDS = L::B_WORDS;
    constexpr int NBLK = KC / 16 / KS;                      // 16-k blocks per wave and chunk
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const As = smem;                                 // [2][A_WORDS]
    float* const Bs = smem + 2 * A_WORDS;                   // [2][B_WORDS]
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, ks = tid >> 8;
    const int l16 = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * G16_BM;

    f32x4 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int xcol[XT];                                           // clamped: a wave without a real extra tile repeats the last one
#pragma unroll
    for (int x = 0; x < XT; ++x) xcol[x] = (wave * XT + x < NT ? wave * XT + x : NT - 1) * 16;

    // The A descriptor covers only this workgroup's rows: no 4 GiB limit on A, and rows past M are out of range.
    const int rows_here = min(G16_BM, M - row0);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (size_t)row0 * K), 0, rows_here * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, K * N * 4, 0x00020000);
    constexpr int OOB = 0x7ffffff0;
    u32x4 ra[2][AQ], rb[2][BQ];                             // two chunks in flight
    auto load_chunk = [&](int k0, int set) {
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int idx = tid + q * TH;
            const int r = idx / (KC / 4), gk = k0 + (idx % (KC / 4)) * 4;
            const int off = (r * K + gk) * 4;
            if (VEC) {
                ra[set][q] = __builtin_amdgcn_raw_buffer_load_b128(rsA, idx < A4 && gk < K ? off : OOB, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    ra[set][q][j] = __builtin_amdgcn_raw_buffer_load_b32(rsA, idx < A4 && gk + j < K ? off + 4 * j : OOB, 0, 0);
            }
        }
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const int idx = tid + q * TH;
            const int kk = idx / (NP / 4), c = (idx - kk * (NP / 4)) * 4;
            const int gk = k0 + kk, off = (gk * N + c) * 4;
            if (VEC) {
                rb[set][q] = __builtin_amdgcn_raw_buffer_load_b128(rsB, idx < B4 && gk < K && c < N ? off : OOB, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    rb[set][q][j] = __builtin_amdgcn_raw_buffer_load_b32(rsB, idx < B4 && gk < K && c + j < N ? off + 4 * j : OOB, 0, 0);
            }
        }
    };
    auto store_chunk = [&](int buf, int set) {
#pragma unroll
        for (int q = 0; q < AQ; ++q) {
            const int idx = tid + q * TH;
            const int r = idx / (KC / 4), kq = (idx % (KC / 4)) * 4;
            if (idx < A4) *reinterpret_cast<u32x4*>(&As[buf * A_WORDS + r * AS + kq]) = ra[set][q];
        }
#pragma unroll
        for (int q = 0; q < BQ; ++q) {
            const int idx = tid + q * TH;
            const int kk = idx / (NP / 4), c = (idx - kk * (NP / 4)) * 4;
            if (idx < B4) *reinterpret_cast<u32x4*>(&Bs[buf * B_WORDS + kk * SB + c]) = rb[set][q];
        }
    };
    // operands of one 16-k block: read ahead of the MFMAs of the block before it
    struct Ops { f32x4 a0, a1; float b[4][NB]; };
    const int a0_off = (wave * 16 + l16) * AS + g * 4, a1_off = (64 + l16) * AS + g * 4;
    auto read_block = [&](Ops& o, int buf, int kb) {
        o.a0 = *reinterpret_cast<const f32x4*>(&As[buf * A_WORDS + a0_off + kb * 16]);
        o.a1 = *reinterpret_cast<const f32x4*>(&As[buf * A_WORDS + a1_off + kb * 16]);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const float* brow = &Bs[buf * B_WORDS + (kb * 16 + g * 4 + s) * SB + l16];
#pragma unroll
            for (int t = 0; t < NT; ++t) o.b[s][t] = brow[t * 16];
#pragma unroll
            for (int x = 0; x < XT; ++x) o.b[s][NT + x] = brow[xcol[x]];
        }
    };
    auto mfma_block = [&](const Ops& o) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a0[s], o.b[s][t], acc[t], 0, 0, 0);
#pragma unroll
            for (int x = 0; x < XT; ++x) acc[NT + x] = __builtin_amdgcn_mfma_f32_16x16x4f32(o.a1[s], o.b[s][NT + x], acc[NT + x], 0, 0, 0);
        }
    };

    // Iteration ch (parity p): LDS buffer p holds chunk ch, register set p^1 holds chunk ch+1 (in flight), set p is free
    // and takes chunk ch+2.  Chunk count rounded up to even; chunks past K load zeros.  Every load, store and barrier is
    // unconditional (the compiler then counts the loads in flight: vmcnt(n), not vmcnt(0)).
    const int nchunks = ((K + KC - 1) / KC + 1) & ~1;
    auto step = [&](int ch, int p) {
        load_chunk((ch + 2) * KC, p);
        Ops o[2];
        read_block(o[0], p, ks * NBLK);
#pragma unroll
        for (int i = 0; i < NBLK; ++i) {
            if (i + 1 < NBLK) read_block(o[(i + 1) & 1], p, ks * NBLK + i + 1);
            __builtin_amdgcn_sched_barrier(0);              // keep the reads ahead of the matrix pipe
            mfma_block(o[i & 1]);
        }
        store_chunk(p ^ 1, p ^ 1);
        __syncthreads();
    };
    load_chunk(0, 0);
    load_chunk(KC, 1);
    store_chunk(0, 0);
    __syncthreads();
    for (int ch = 0; ch < nchunks; ch += 2) {
        step(ch, 0);
        step(ch + 1, 1);
    }
    // accumulators -> LDS tile (C/D layout of 16x16x4: col = lane&15, row = 4*(lane>>4) + reg), summed over the k-split
    // wave groups, then streamed out as whole rows
    float* const Cs = smem;
    auto tile_rows = [&](auto&& fn) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) fn(acc[t][r], (wave * 16 + 4 * g + r) * SB + t * 16 + l16);
#pragma unroll
        for (int x = 0; x < XT; ++x)
            if (wave * XT + x < NT)
#pragma unroll
                for (int r = 0; r < 4; ++r) fn(acc[NT + x][r], (64 + 4 * g + r) * SB + xcol[x] + l16);
    };
#pragma unroll
    for (int part = KS - 1; part >= 0; --part) {
        if (ks == part) {
            if (part == KS - 1) tile_rows([&](float v, int off) { Cs[off] = v; });
            else tile_rows([&](float v, int off) { Cs[off] += v; });
        }
        __syncthreads();
    }
    if (VEC) {
        const int n4 = N >> 2;
        for (int idx = tid; idx < G16_BM * n4; idx += TH) {
            const int r = idx / n4, c = (idx - r * n4) * 4;
            if (row0 + r >= M) break;
            float4 v = *reinterpret_cast<const float4*>(&Cs[r * SB + c]);
            if (bias) {
                const float4 bb = *reinterpret_cast<const float4*>(bias + c);
                v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            }
            if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
            *reinterpret_cast<float4*>(C + (size_t)(row0 + r) * N + c) = v;
        }
    } else {
        for (int idx = tid; idx < G16_BM * N; idx += TH) {
            const int r = idx / N, c = idx - r * N;
            if (row0 + r >= M) break;
            float v = Cs[r * SB + c] + (bias ? bias[c] : 0.0f);
            if (relu) v = fmaxf(v, 0.f);
            C[(size_t)(row0 + r) * N + c] = v;
        }
    }
}

template <int NT, bool VEC, int KS, int KC>
static hipError_t gemm16_launch(int M, int N, int K, const float* A, const float* B, const float* bias, int relu, float* C, hipStream_t s) {
    static bool attr_set[64] = {};                          // dynamic LDS opt-in, once per kernel and device
    auto kern = gemm16_f32_kernel<NT, VEC, KS, KC>;
    constexpr int lds = G16Layout<NT, KC>::LDS_BYTES;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3((M + G16_BM - 1) / G16_BM), dim3(256 * KS), lds, s, M, N, K, A, B, bias, relu, C);
    return hipGetLastError();
}

// ======================================================================================================================
// Round 4: the same product with B RESIDENT in LDS and no barrier inside the K loop.  (gemm16_f32_kernel above synchronises its
// eight wavefronts after every 32-k chunk -- one 16-k block of MFMAs per wavefront between barriers -- and stages A through
// LDS although every A row tile belongs to exactly one wavefront: its matrix pipes were busy 58 % of the time a CU was busy,
// 31 us = 40 % of the f32 MFMA peak on PubMed's 19 717 x 500 x 100.)  Same workgroup tile (80 rows x NT*16 columns, eight
// wavefronts = two K groups of four, 247 workgroups = one round on 256 CUs), same accumulator layout and epilogue, but:
//   * K runs in PHASES of 128: the phase's slice of B (128 x NP, stored [k/4][col][k%4] so that a lane's four MFMA steps of a
//     column tile are ONE ds_read_b128) sits in one of two LDS buffers; the next phase's slice goes from global memory STRAIGHT
//     into the other buffer (buffer_load ... lds, no registers) while this phase's MFMAs run: one barrier per 128 k, four for K = 500;
//   * A never touches LDS: lane (row, g) loads the 16 bytes  A[row][16 kb + 4 g ..]  of each of its blocks straight into the
//     MFMA operand registers, a whole phase ahead (row tile w of the wavefront and the shared fifth row tile);
//   * inside a phase a wavefront runs its four 16-k blocks back to back: nine ds_read_b128 (read one block ahead) and 36 MFMAs
//     per block, with the other wavefront of its SIMD filling the gaps.
// ======================================================================================================================
#define GBR_KH 128                                           /* k per phase */
#ifdef TLC_GBR_DEBUG
__device__ unsigned long long g_gbr_dbg[16];                 // cycle sums of wavefront 0 of every workgroup: see GBR_STAMP
#define GBR_STAMP(k) do { const unsigned long long _t = clock64(); if (tid == 0) atomicAdd(&g_gbr_dbg[(k)], _t - t_prev); t_prev = _t; } while (0)
#else
#define GBR_STAMP(k) do { } while (0)
#endif
template <int NT>
struct GbrLayout {
    static constexpr int NP = NT * 16;
    // one phase of B = the 128 rows of the slice exactly as they lie in global memory (row stride N <= NP), rounded up to the 1 KiB
    // pieces it arrives in, plus the reach of the last row's column tiles beyond N
    static constexpr int B_WORDS = ((GBR_KH * NP * 4 + 1023) / 1024) * 256 + 64;
    static constexpr int SB = NP + 4, C_WORDS = G16_BM * SB;
    static constexpr int LDS_BYTES = 4 * (2 * B_WORDS > C_WORDS ? 2 * B_WORDS : C_WORDS);
};

template <int NT, int KS>
__global__ __launch_bounds__(256 * KS) void gemm_bres_f32_kernel(int M, int N, int K, const float* __restrict__ A, const float* __restrict__ B,
                                                            const float* __restrict__ bias, int relu, float* __restrict__ C) {
    using L = GbrLayout<NT>;
    constexpr int TH = 256 * KS, NP = L::NP, SB = L::SB;
    constexpr int XT = (NT + 3) / 4;                        // column tiles of the fifth row tile per wave
    constexpr int NB = NT + XT;
    constexpr int BLK = GBR_KH / 16 / KS;                   // 16-k blocks per wavefront and phase (KS K groups)
    constexpr int NWV = 4 * KS;                             // wavefronts of the workgroup
    constexpr int BITEMS = (GBR_KH / 4) * NP;               // (k-quad, column) items of one phase of B
    constexpr int BQ = (BITEMS + TH - 1) / TH;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = (tid >> 6) & 3, ks = tid >> 8;
    const int l16 = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * G16_BM;

#ifdef TLC_GBR_DEBUG
    unsigned long long t_prev = clock64();
#endif
    f32x4 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    int xcol[XT];                                           // clamped: a wave without a real extra tile repeats the last one
#pragma unroll
    for (int x = 0; x < XT; ++x) xcol[x] = (wave * XT + x < NT ? wave * XT + x : NT - 1) * 16;

    const int rows_here = min(G16_BM, M - row0);
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A + (size_t)row0 * K), 0, rows_here * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, K * N * 4, 0x00020000);
    constexpr int OOB = 0x7ffffff0;
    const int nph = (K + GBR_KH - 1) / GBR_KH;
    // A operands of one phase: blocks ks*BLK .. of the phase, row tile `wave` (a0) and the fifth row tile (a1)
    const int ra0 = (wave * 16 + l16) * K, ra1 = (64 + l16) * K;
    auto load_a0 = [&](int ph, int i) __attribute__((always_inline)) -> u32x4 {                        // block i of this wavefront's share of phase ph: row tile `wave`
        const int k = ph * GBR_KH + (ks * BLK + i) * 16 + g * 4;
        return __builtin_amdgcn_raw_buffer_load_b128(rsA, k < K ? (ra0 + k) * 4 : OOB, 0, 0);
    };
    auto load_a1 = [&](int ph, int i) __attribute__((always_inline)) -> u32x4 {                        // ... the shared fifth row tile
        const int k = ph * GBR_KH + (ks * BLK + i) * 16 + g * 4;
        return __builtin_amdgcn_raw_buffer_load_b128(rsA, k < K ? (ra1 + k) * 4 : OOB, 0, 0);
    };
    // One phase of B straight from global memory into LDS (buffer_load_dwordx4 ... lds: no registers in between, so the whole next
    // slice is requested at the start of a phase and lands under its MFMAs).  The slice's 128 rows are contiguous in global memory
    // (B is row-major [K][N]) and are copied as they are, in 1 KiB pieces (64 lanes x 16 bytes): ceil(N / 2) pieces per slice, dealt
    // to the wavefronts round-robin.  (Dword pieces into a [k/4][col][k%4] image -- one ds_read_b128 per operand tile -- were
    // measured first: 224 pieces per slice at ~100 cycles of issue each, the wavefronts waited 6 500 cycles per phase for them.)
    // Rows >= K fall behind the descriptor: zeros.  Columns >= N of an operand tile read the next row's start: never stored.
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int npieces = (GBR_KH * N * 4 + 1023) >> 10;
    auto issue_b = [&](int ph, int buf) __attribute__((always_inline)) {
        const int vbase = ph * GBR_KH * N * 4 + lane * 16;
        for (int q = wave8; q < npieces; q += NWV)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)&smem[buf * L::B_WORDS + q * 256], 16, vbase, q * 1024, 0, 0);
    };
    constexpr int BQ_UNUSED = BQ;
    (void)BQ_UNUSED;
    struct Bops { f32x4 b[NB]; };
    auto read_b = [&](Bops& o, int buf, int blk) __attribute__((always_inline)) {           // blk: block of the phase (0 .. KH/16)
        const float* base = &smem[buf * L::B_WORDS + (blk * 16 + g * 4) * N + l16];
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) {
#pragma unroll
            for (int t = 0; t < NT; ++t) o.b[t][sidx] = base[sidx * N + t * 16];
#pragma unroll
            for (int x = 0; x < XT; ++x) o.b[NT + x][sidx] = base[sidx * N + xcol[x]];
        }
    };

    // Two named operand sets (an array indexed by the phase parity would go to scratch).  A phase's A operands are requested TOGETHER
    // at the start of the phase before it: the BLK loads of a row are 64 BLK contiguous bytes issued back to back (one DRAM page
    // while it is open); issued one per block, 0.5 us apart, every 64-byte piece paid an activation of its own and the wavefronts
    // waited 6 500 cycles per phase for them.
    u32x4 xa0[BLK], xa1[BLK], ya0[BLK], ya1[BLK];
    auto phase = [&](int ph, int buf, const u32x4 (&ca0)[BLK], const u32x4 (&ca1)[BLK], u32x4 (&na0)[BLK], u32x4 (&na1)[BLK]) __attribute__((always_inline)) {
#ifndef TLC_GBR_SKIP_B                                       /* (diagnostic builds: which operand stream the wavefronts wait for) */
        issue_b(ph + 1, buf ^ 1);                            // (the other buffer: its last readers passed the barrier of the phase before)
#endif
#ifndef TLC_GBR_SKIP_A
#pragma unroll
        for (int i = 0; i < BLK; ++i) { na0[i] = load_a0(ph + 1, i); na1[i] = load_a1(ph + 1, i); }
#else
#pragma unroll
        for (int i = 0; i < BLK; ++i) { na0[i] = ca0[i]; na1[i] = ca1[i]; }
#endif
        constexpr int NO = KS <= 2 ? 2 : 1;                  // (four wavefronts per SIMD hide an LDS round trip themselves: one operand set)
        Bops o[NO];
        read_b(o[0], buf, ks * BLK);
#pragma unroll
        for (int i = 0; i < BLK; ++i) {
            if (NO == 2 && i + 1 < BLK) read_b(o[(i + 1) & 1], buf, ks * BLK + i + 1);
            if (NO == 1 && i > 0) read_b(o[0], buf, ks * BLK + i);
            const f32x4 fa0 = __builtin_bit_cast(f32x4, ca0[i]), fa1 = __builtin_bit_cast(f32x4, ca1[i]);
            __builtin_amdgcn_sched_barrier(0);              // keep the reads and requests ahead of the matrix pipe
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa0[sidx], o[i & (NO - 1)].b[t][sidx], acc[t], 0, 0, 0);
#pragma unroll
                for (int x = 0; x < XT; ++x) acc[NT + x] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa1[sidx], o[i & (NO - 1)].b[NT + x][sidx], acc[NT + x], 0, 0, 0);
            }
        }
        GBR_STAMP(1);                                        // [1] a phase's MFMAs issued
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wavefront's part of the next slice is in LDS ...
        GBR_STAMP(2);                                        // [2] waiting for its own loads
        __syncthreads();                                     // ... and everybody's
        GBR_STAMP(3);                                        // [3] barrier
    };
#pragma unroll
    for (int i = 0; i < BLK; ++i) { xa0[i] = load_a0(0, i); xa1[i] = load_a1(0, i); }
    issue_b(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    GBR_STAMP(0);                                            // [0] prologue: first operands in
    for (int ph = 0; ph < nph; ph += 2) {
        phase(ph, 0, xa0, xa1, ya0, ya1);
        if (ph + 1 < nph) phase(ph + 1, 1, ya0, ya1, xa0, xa1);      // (uniform)
    }
    // accumulators -> LDS tile (C/D layout of 16x16x4: col = lane&15, row = 4*(lane>>4) + reg), summed over the two K groups,
    // then streamed out as whole rows (as in gemm16_f32_kernel)
    float* const Cs = smem;
    auto tile_rows = [&](auto&& fn) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) fn(acc[t][r], (wave * 16 + 4 * g + r) * SB + t * 16 + l16);
#pragma unroll
        for (int x = 0; x < XT; ++x)
            if (wave * XT + x < NT)
#pragma unroll
                for (int r = 0; r < 4; ++r) fn(acc[NT + x][r], (64 + 4 * g + r) * SB + xcol[x] + l16);
    };
#pragma unroll
    for (int part = KS - 1; part >= 0; --part) {
        if (ks == part) {
            if (part == KS - 1) tile_rows([&](float v, int off) { Cs[off] = v; });
            else tile_rows([&](float v, int off) { Cs[off] += v; });
        }
        __syncthreads();
    }
    const int n4 = N >> 2;
    for (int idx = tid; idx < G16_BM * n4; idx += TH) {
        const int r = idx / n4, c = (idx - r * n4) * 4;
        if (row0 + r >= M) break;
        float4 v = *reinterpret_cast<const float4*>(&Cs[r * SB + c]);
        if (bias) {
            const float4 bb = *reinterpret_cast<const float4*>(bias + c);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
        }
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(C + (size_t)(row0 + r) * N + c) = v;
    }
    GBR_STAMP(4);                                            // [4] epilogue (K-group sums through LDS, row stores issued)
#ifdef TLC_GBR_DEBUG
    if (tid == 0) atomicAdd(&g_gbr_dbg[15], 1ull);
#endif
}

#ifndef TLC_GBR_KS
#define TLC_GBR_KS 2
#endif
template <int NT>
static hipError_t gemm_bres_launch(int M, int N, int K, const float* A, const float* B, const float* bias, int relu, float* C, hipStream_t s) {
    static bool attr_set[64] = {};
    constexpr int KS = TLC_GBR_KS;
    auto kern = gemm_bres_f32_kernel<NT, KS>;
    constexpr int lds = GbrLayout<NT>::LDS_BYTES;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(kern, dim3((M + G16_BM - 1) / G16_BM), dim3(256 * KS), lds, s, M, N, K, A, B, bias, relu, C);
#ifdef TLC_GBR_DEBUG
    {
        static int calls = 0;
        if (++calls == 50) {                                  // (one report, once warm)
            unsigned long long h[16], z[16] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gbr_dbg), z, sizeof(z));
            hipLaunchKernelGGL(kern, dim3((M + G16_BM - 1) / G16_BM), dim3(256 * KS), lds, s, M, N, K, A, B, bias, relu, C);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_gbr_dbg), sizeof(h));
            const double w = (double)(h[15] ? h[15] : 1);
            fprintf(stderr, "[gemm_bres %d x %d x %d] cycles of wavefront 0, mean over %llu workgroups: prologue %.0f | MFMA phases %.0f | own loads %.0f | barrier %.0f | epilogue %.0f\n",
                    M, K, N, h[15], h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w);
        }
    }
#endif
    return hipGetLastError();
}

// ======================================================================================================================
// Skinny-K variant of the same product: K = 16 / 32 / 64 (the PDGNN layers: x_l, [P|Q|alpha], the edge head), M in the millions.
// With one or two K chunks the tiled kernel above is all prologue and epilogue (274 us for [1M,32] @ [32,68], 1.5 TB/s of its
// 411 MB); here nothing goes through LDS: the whole B sits in registers in MFMA operand layout (K/4 x NT floats per lane),
// persistent wavefronts stream 16-row tiles of A straight from global memory -- lane (row, g) reads the K/4 CONTIGUOUS floats
// k = g*K/4 .. of its row (MFMA step s of lane group g consumes k = g*K/4 + s for A and B alike: a sum over k does not care
// about the order), so a tile is one fully coalesced 16 x K block -- the next tile's loads are issued before this tile's
// MFMAs, and the accumulators go out as 64-byte row segments.
// ======================================================================================================================
template <int KQ, int NT>
__global__ __launch_bounds__(256) void gemm_skinny_f32_kernel(int M, int N, const float* __restrict__ A, const float* __restrict__ B,
                                                              const float* __restrict__ bias, int relu, float* __restrict__ C) {
    constexpr int K = KQ * 4;
    const int lane = threadIdx.x & 63, l16 = lane & 15, g = lane >> 4;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    float b[KQ][NT];
#pragma unroll
    for (int sidx = 0; sidx < KQ; ++sidx)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c = t * 16 + l16;
            b[sidx][t] = c < N ? B[(size_t)(g * KQ + sidx) * N + c] : 0.0f;
        }
    float bs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) bs[t] = (bias && t * 16 + l16 < N) ? bias[t * 16 + l16] : 0.0f;
    const long long n_tiles = ((long long)M + 15) >> 4;
    auto load_tile = [&](long long tile, f32x4 (&a)[KQ / 4]) {
        long long r = tile * 16 + l16;
        if (r >= M) r = M - 1;                                          // (rows past M are computed and not stored)
        const f32x4* src = reinterpret_cast<const f32x4*>(A + (size_t)r * K + g * KQ);
#pragma unroll
        for (int q = 0; q < KQ / 4; ++q) a[q] = src[q];
    };
    f32x4 a_cur[KQ / 4], a_nxt[KQ / 4];
    long long tile = wave;
    if (tile < n_tiles) load_tile(tile, a_cur);
    for (; tile < n_tiles; tile += n_waves) {
        const long long nt = tile + n_waves;
        if (nt < n_tiles) load_tile(nt, a_nxt);
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int sidx = 0; sidx < KQ; ++sidx)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[sidx >> 2][sidx & 3], b[sidx][t], acc[t], 0, 0, 0);
        // C/D layout: col = lane & 15, row = 4 * (lane >> 4) + reg
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c = t * 16 + l16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long long row = tile * 16 + 4 * g + r;
                float v = acc[t][r] + bs[t];
                if (relu) v = fmaxf(v, 0.f);
                if (row < M && c < N) C[(size_t)row * N + c] = v;
            }
        }
#pragma unroll
        for (int q = 0; q < KQ / 4; ++q) a_cur[q] = a_nxt[q];
    }
}

template <int KQ>
static bool gemm_skinny_launch(int M, int N, const float* A, const float* B, const float* bias, int relu, float* C, hipStream_t s) {
    const long long tiles = ((long long)M + 15) / 16;
    long long blocks = (tiles + 3) / 4;
    if (blocks > 256 * 8) blocks = 256 * 8;                             // persistent: at most 8 workgroups per CU
#define TLC_SK(NT_) case NT_: hipLaunchKernelGGL((gemm_skinny_f32_kernel<KQ, NT_>), dim3((unsigned)blocks), dim3(256), 0, s, M, N, A, B, bias, relu, C); return true;
    switch ((N + 15) / 16) {
        TLC_SK(1) TLC_SK(2) TLC_SK(3) TLC_SK(4) TLC_SK(5)
        default: return false;
    }
#undef TLC_SK
}

// ======================================================================================================================
// CSR SpMM: Y[i,:] = act(sum_j val[j] * X[col[j],:] + bias).  Each row is owned by G lanes, every lane covering 4 adjacent
// feature columns with 16-byte gathers; 8 neighbour rows are in flight per lane (hub rows of the graph have hundreds of
// entries and would otherwise serialise on the gather latency).  k % 4 == 0; k <= 4*G.
// ======================================================================================================================
#define SPMM_HUB 32
#define SPMM_CAP 2048
template <int G>
__global__ __launch_bounds__(256) void spmm_csr_v4_kernel(int n_rows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                          const float* __restrict__ val, const float* __restrict__ X, int k,
                                                          const float* __restrict__ bias, int relu, float* __restrict__ Y) {
    constexpr int NG = 256 / G;                         // rows (lane groups) per workgroup
    __shared__ int s_col[NG * SPMM_HUB];                // per group: the indices/values of its own row
    __shared__ float s_val[NG * SPMM_HUB];
    __shared__ int h_col[SPMM_CAP];                     // hub rows: one chunk of the row, staged by the whole workgroup
    __shared__ float h_val[SPMM_CAP];
    __shared__ int hub_rows[NG];
    __shared__ int n_hub;
    __shared__ float4 part[NG][G];
    const int tid = threadIdx.x, grp = tid / G, gl = tid % G;
    // rows are dealt to workgroups round-robin (row = grp * gridDim.x + blockIdx.x): hub rows with neighbouring ids
    // (old nodes of a preferential-attachment graph, degree-sorted inputs) land in different workgroups
    const int row = grp * gridDim.x + blockIdx.x;
    const int c = gl * 4;
    const bool lane_ok = c < k;
    if (tid == 0) n_hub = 0;
    __syncthreads();
    int b = 0, e = 0;
    if (row < n_rows) { b = rowptr[row]; e = rowptr[row + 1]; }
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);                     // (with the first loads: read where it is added it is one more dependent trip)
    if (bias && lane_ok) { bias4.x = bias[c]; bias4.y = bias[c + 1]; bias4.z = bias[c + 2]; bias4.w = bias[c + 3]; }
    const int d = e - b;
    // rows above SPMM_HUB entries are left to the whole workgroup below: one lane group walking a hub row alone
    // (dozens of dependent gather rounds) would be the tail of the launch
    const bool hub = d > SPMM_HUB;
    if (hub && gl == 0) hub_rows[atomicAdd(&n_hub, 1)] = row;
    if (!hub)
        for (int i = gl; i < d; i += G) { s_col[grp * SPMM_HUB + i] = col[b + i]; s_val[grp * SPMM_HUB + i] = val[b + i]; }
    __syncthreads();
    // epilogue of one row, called by all G lanes of the row's group (lanes past k carry zeros and do not store):
    // bias, ReLU (relu & 1) and, with relu & 2, emb.renorm_(2, 0, 1) of TLCGNN.py:48 -- rows with an L2 norm above 1 are
    // scaled by 1 / (norm + 1e-7); the norm is a shuffle reduction over the group
    auto finish = [&](float4 acc, int r, bool store) {
        if (store) {
            acc.x += bias4.x; acc.y += bias4.y; acc.z += bias4.z; acc.w += bias4.w;
            if (relu & 1) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
        }
        if (relu & 2) {
            float ss = store ? acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w : 0.0f;
#pragma unroll
            for (int o = G >> 1; o > 0; o >>= 1) ss += __shfl_xor(ss, o, G);
            const float nrm = sqrtf(ss);
            if (nrm > 1.0f) {
                const float sc = 1.0f / (nrm + 1e-7f);
                acc.x *= sc; acc.y *= sc; acc.z *= sc; acc.w *= sc;
            }
        }
        if (store) *reinterpret_cast<float4*>(Y + (size_t)r * k + c) = acc;
    };
    // staged entries j0, j0+step, ... < j1, eight gathers in flight per round: a gather depends on one global round
    // trip (the X row); the tail round is predicated (entries past j1 re-read the round's first row and are zeroed)
    auto gather = [&](float4 acc, const int* sc, const float* sv, int j0, int j1, int step) {
        for (int j = j0; j < j1; j += 8 * step) {
            float4 x[8];
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jj = j + u * step;
                const bool ok = jj < j1;
                const int js = ok ? jj : j;
                v[u] = ok ? sv[js] : 0.0f;
                x[u] = *reinterpret_cast<const float4*>(X + (size_t)sc[js] * k + c);
                if (!ok) x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u] * x[u].x; acc.y += v[u] * x[u].y; acc.z += v[u] * x[u].z; acc.w += v[u] * x[u].w; }
        }
        return acc;
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const bool mine = row < n_rows && !hub;             // (uniform over the group)
        float4 acc = zero4;
        if (mine && lane_ok) acc = gather(zero4, s_col + grp * SPMM_HUB, s_val + grp * SPMM_HUB, 0, d, 1);
        if (mine) finish(acc, row, lane_ok);
    }
    const int nh = n_hub;
    for (int h = 0; h < nh; ++h) {                      // hub row: group g takes entries g, g+NG, ...; fixed-order sum
        const int r = hub_rows[h];
        const int hb = rowptr[r], he = rowptr[r + 1];
        float4 acc = zero4;
        for (int cb = hb; cb < he; cb += SPMM_CAP) {
            const int cn = min(SPMM_CAP, he - cb);
            __syncthreads();                            // the previous chunk / row is done with h_col, h_val and part
            for (int i = tid; i < cn; i += 256) { h_col[i] = col[cb + i]; h_val[i] = val[cb + i]; }
            __syncthreads();
            if (lane_ok) acc = gather(acc, h_col, h_val, grp, cn, NG);
        }
        part[grp][gl] = acc;
        __syncthreads();
        if (grp == 0) {
            float4 t = part[0][gl];
#pragma unroll
            for (int q = 1; q < NG; ++q) { const float4 p = part[q][gl]; t.x += p.x; t.y += p.y; t.z += p.z; t.w += p.w; }
            finish(t, r, lane_ok);
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// Round 5: the same aggregate for k == 16 (the second GCN layer: 64-byte rows) with SIXTEEN lanes per row -- four entry slots x
// four 16-byte column lanes -- four rows per wavefront, no LDS and no barrier.  (spmm_csr_v4_kernel<4> gave a row four lanes:
// 308 workgroups for PubMed's 19 717 rows = 1.2 wavefronts per SIMD, every row's entries staged through LDS behind a barrier and
// every hub row walked by its whole workgroup, one after the other: 10.4 us for 8.7 MB, 0.10 of HBM.)  A lane loads the indices
// and values of its slot's entries itself (the four column lanes of a slot read the same words: one request) and has eight
// gathers in flight; rows above 64 entries are taken by the whole wavefront afterwards (sixteen slots: 128 entries per round
// trip).  The sums are folded over the slots in a fixed order, so the result depends on the row alone.
// ----------------------------------------------------------------------------------------------------------------------
#define SPMM16_WAVE_ROW 64
template <int J>
__device__ __forceinline__ float lane_xor_f32(float v) { return __builtin_bit_cast(float, tlc_lane_xor_u32<J>(__builtin_bit_cast(unsigned, v))); }
template <int J>
__device__ __forceinline__ float4 fold4(float4 a) {                   // a += the value of lane ^ J (vector-ALU lane exchanges, no LDS crossbar)
    a.x += lane_xor_f32<J>(a.x); a.y += lane_xor_f32<J>(a.y); a.z += lane_xor_f32<J>(a.z); a.w += lane_xor_f32<J>(a.w);
    return a;
}
__global__ __launch_bounds__(256) void spmm16_kernel(int n_rows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                     const float* __restrict__ val, const float* __restrict__ X,
                                                     const float* __restrict__ bias, int relu, float* __restrict__ Y) {
    const int tid = threadIdx.x, lane = tid & 63, cl = lane & 3, c = cl * 4;
    const int row = (tid >> 4) * gridDim.x + blockIdx.x;                 // rows dealt round-robin (hubs with neighbouring ids spread out)
    int b = 0, d = 0;
    if (row < n_rows) { b = rowptr[row]; d = rowptr[row + 1] - b; }
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 bias4 = zero4;                                               // (with the first loads: read where it is added it is one more dependent trip)
    if (bias) { bias4.x = bias[c]; bias4.y = bias[c + 1]; bias4.z = bias[c + 2]; bias4.w = bias[c + 3]; }
    // entries j0 + slot, + nslot, ... of row segment [rb, rb + rd): eight gathers in flight, tail predicated
    auto gather = [&](float4 acc, int rb, int rd, int slot, int nslot) {
        for (int j = slot; j < rd; j += 8 * nslot) {
            int cj[8];
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jj = j + u * nslot;
                const bool ok = jj < rd;
                cj[u] = col[rb + (ok ? jj : j)];
                v[u] = ok ? val[rb + jj] : 0.0f;
            }
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) x[u] = *reinterpret_cast<const float4*>(X + (size_t)cj[u] * 16 + c);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u] * x[u].x; acc.y += v[u] * x[u].y; acc.z += v[u] * x[u].z; acc.w += v[u] * x[u].w; }
        }
        return acc;
    };
    // bias, ReLU (relu & 1), emb.renorm_(2, 0, 1) (relu & 2; TLCGNN.py:48): the four column lanes hold the row
    auto finish = [&](float4 acc, int r, bool store) {
        acc.x += bias4.x; acc.y += bias4.y; acc.z += bias4.z; acc.w += bias4.w;
        if (relu & 1) { acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f); }
        if (relu & 2) {
            float ss = acc.x * acc.x + acc.y * acc.y + acc.z * acc.z + acc.w * acc.w;
            ss += __builtin_bit_cast(float, tlc_lane_xor_u32<1>(__builtin_bit_cast(unsigned, ss)));
            ss += __builtin_bit_cast(float, tlc_lane_xor_u32<2>(__builtin_bit_cast(unsigned, ss)));
            const float nrm = sqrtf(ss);
            if (nrm > 1.0f) {
                const float sc = 1.0f / (nrm + 1e-7f);
                acc.x *= sc; acc.y *= sc; acc.z *= sc; acc.w *= sc;
            }
        }
        if (store) *reinterpret_cast<float4*>(Y + (size_t)r * 16 + c) = acc;
    };
    const bool wide = d > SPMM16_WAVE_ROW;
    {
        float4 acc = gather(zero4, b, wide ? 0 : d, (lane >> 2) & 3, 4);
        acc = fold4<8>(fold4<4>(acc));
        finish(acc, row, row < n_rows && !wide && (lane & 12) == 0);
    }
    if (__ballot(wide)) {                                            // (rare: a wavefront that owns a hub row)
#pragma unroll 1
        for (int q = 0; q < 4; ++q) {
            const int dq = __builtin_amdgcn_readlane(d, q * 16);
            if (dq <= SPMM16_WAVE_ROW) continue;                         // (uniform)
            const int bq = __builtin_amdgcn_readlane(b, q * 16), rq = __builtin_amdgcn_readlane(row, q * 16);
            float4 acc = gather(zero4, bq, dq, lane >> 2, 16);
            acc = fold4<32>(fold4<16>(fold4<8>(fold4<4>(acc))));
            finish(acc, rq, lane < 4);
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// Round 5: the first GCN layer's aggregate with the second layer's projection in its epilogue (TLCGNN.py:22-25:
// conv2's  x @ W  applied to  relu(conv1(x))  row by row):  Y2[i, :] = relu(sum_j val[j] X[col[j], :] + bias) @ W2,  W2 [k, 16].
// The gather is spmm_csr_v4_kernel<32>'s (32 lanes per row, four columns per lane); the 100-wide activation row never leaves the
// registers: lane gl multiplies its four values with rows 4 gl .. of W2 (in LDS, lane stride 68 words: conflict-free 16-byte
// reads) into sixteen partial outputs, and a halving butterfly over the 32 lanes (8 + 4 + 2 + 1 + 1 exchanges instead of 16 x 5)
// leaves output o on the lane whose low four bits are o reversed.  Saves the separate 19 717 x 100 x 16 product's launch and the
// 15.8 MB that the activation matrix cost to write and read back.
// ----------------------------------------------------------------------------------------------------------------------
#define SPMM_W2_CAP 512
__global__ __launch_bounds__(256) void spmm_w2_kernel(int n_rows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                      const float* __restrict__ val, const float* __restrict__ X, int k,
                                                      const float* __restrict__ bias, const float* __restrict__ W2, float* __restrict__ Y2) {
    constexpr int G = 32, NG = 256 / G, WS = 68;
    __shared__ int s_col[NG * SPMM_HUB];
    __shared__ float s_val[NG * SPMM_HUB];
    __shared__ int h_col[SPMM_W2_CAP];
    __shared__ float h_val[SPMM_W2_CAP];
    __shared__ int hub_flag[NG];                         // the group's row if it is a hub row, else -1
    __shared__ float4 part[NG][G];
    __shared__ __attribute__((aligned(16))) float s_w2[G * WS];
    const int tid = threadIdx.x, grp = tid / G, gl = tid % G;
    const int row = grp * gridDim.x + blockIdx.x;
    const int c = gl * 4;
    const bool lane_ok = c < k;
    // one barrier in the common case: row bounds -> the row's entries into LDS, W2 into LDS beside them (its loads depend on
    // nothing and are in flight with the row bounds), hub rows flagged per group
    int b = 0, e = 0;
    if (row < n_rows) { b = rowptr[row]; e = rowptr[row + 1]; }
    // (the bias with the first loads: read where it is added -- after the gather -- it was a fourth dependent trip to memory:
    // cycle stamps per wavefront showed more than half of a wavefront's time BEHIND the gather)
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias && lane_ok) bias4 = *reinterpret_cast<const float4*>(bias + c);
    float wreg[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {                                        // W2 rows 4 gl + a -> s_w2[gl][a][0..16); rows >= k: zeros
        const int i = tid + q * 256, r = i >> 4;
        wreg[q] = r < k ? W2[i] : 0.0f;
    }
    const int d = e - b;
    const bool hub = d > SPMM_HUB;
    if (gl == 0) hub_flag[grp] = hub ? row : -1;
    if (!hub)
        for (int i = gl; i < d; i += G) { s_col[grp * SPMM_HUB + i] = col[b + i]; s_val[grp * SPMM_HUB + i] = val[b + i]; }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int i = tid + q * 256, r = i >> 4, o = i & 15;
        s_w2[(r >> 2) * WS + (r & 3) * 16 + o] = wreg[q];
    }
    __syncthreads();
    auto gather = [&](float4 acc, const int* sc, const float* sv, int j0, int j1, int step) {
        for (int j = j0; j < j1; j += 8 * step) {
            float4 x[8];
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int jj = j + u * step;
                const bool ok = jj < j1;
                const int js = ok ? jj : j;
                v[u] = ok ? sv[js] : 0.0f;
                x[u] = *reinterpret_cast<const float4*>(X + (size_t)sc[js] * k + c);
                if (!ok) x[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u] * x[u].x; acc.y += v[u] * x[u].y; acc.z += v[u] * x[u].z; acc.w += v[u] * x[u].w; }
        }
        return acc;
    };
    // called by all 32 lanes of a group (lanes past k carry zeros): bias + ReLU, the row times W2, store by the low sixteen lanes
    auto finish = [&](float4 acc, int r, bool store) {
        if (lane_ok) {
            acc.x += bias4.x; acc.y += bias4.y; acc.z += bias4.z; acc.w += bias4.w;
            acc.x = fmaxf(acc.x, 0.f); acc.y = fmaxf(acc.y, 0.f); acc.z = fmaxf(acc.z, 0.f); acc.w = fmaxf(acc.w, 0.f);
        } else {
            acc = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const float hv[4] = {acc.x, acc.y, acc.z, acc.w};
        float p[16];
#pragma unroll
        for (int o = 0; o < 16; ++o) p[o] = 0.0f;
        const float* wr = s_w2 + gl * WS;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int o4 = 0; o4 < 4; ++o4) {
                const float4 w = *reinterpret_cast<const float4*>(wr + a * 16 + o4 * 4);
                p[o4 * 4] += hv[a] * w.x; p[o4 * 4 + 1] += hv[a] * w.y; p[o4 * 4 + 2] += hv[a] * w.z; p[o4 * 4 + 3] += hv[a] * w.w;
            }
        // halving butterfly: after the step with mask m a lane keeps the half of its outputs selected by its bit m
#pragma unroll
        for (int j = 0; j < 8; ++j) { const bool hi = gl & 1; const float snd = hi ? p[j] : p[j + 8], kp = hi ? p[j + 8] : p[j]; p[j] = kp + __builtin_bit_cast(float, tlc_lane_xor_u32<1>(__builtin_bit_cast(unsigned, snd))); }
#pragma unroll
        for (int j = 0; j < 4; ++j) { const bool hi = gl & 2; const float snd = hi ? p[j] : p[j + 4], kp = hi ? p[j + 4] : p[j]; p[j] = kp + __builtin_bit_cast(float, tlc_lane_xor_u32<2>(__builtin_bit_cast(unsigned, snd))); }
#pragma unroll
        for (int j = 0; j < 2; ++j) { const bool hi = gl & 4; const float snd = hi ? p[j] : p[j + 2], kp = hi ? p[j + 2] : p[j]; p[j] = kp + __builtin_bit_cast(float, tlc_lane_xor_u32<4>(__builtin_bit_cast(unsigned, snd))); }
        { const bool hi = gl & 8; const float snd = hi ? p[0] : p[1], kp = hi ? p[1] : p[0]; p[0] = kp + __builtin_bit_cast(float, tlc_lane_xor_u32<8>(__builtin_bit_cast(unsigned, snd))); }
        p[0] += __builtin_bit_cast(float, tlc_lane_xor_u32<16>(__builtin_bit_cast(unsigned, p[0])));
        const int o = ((gl & 1) << 3) | ((gl & 2) << 1) | ((gl & 4) >> 1) | ((gl & 8) >> 3);
        if (store && gl < 16) Y2[(size_t)r * 16 + o] = p[0];
    };
    const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
    {
        const bool mine = row < n_rows && !hub;             // (uniform over the group)
        float4 acc = zero4;
        if (mine && lane_ok) acc = gather(zero4, s_col + grp * SPMM_HUB, s_val + grp * SPMM_HUB, 0, d, 1);
        finish(acc, row, mine);                             // (every lane takes part in the exchanges)
    }
    for (int h = 0; h < NG; ++h) {
        const int r = hub_flag[h];
        if (r < 0) continue;                                // (uniform over the workgroup)
        const int hb = rowptr[r], he = rowptr[r + 1];
        float4 acc = zero4;
        for (int cb = hb; cb < he; cb += SPMM_W2_CAP) {
            const int cn = min(SPMM_W2_CAP, he - cb);
            __syncthreads();
            for (int i = tid; i < cn; i += 256) { h_col[i] = col[cb + i]; h_val[i] = val[cb + i]; }
            __syncthreads();
            if (lane_ok) acc = gather(acc, h_col, h_val, grp, cn, NG);
        }
        part[grp][gl] = acc;
        __syncthreads();
        if (grp == 0) {                                     // (a whole wavefront half: uniform for the exchanges of its 32 lanes)
            float4 t = part[0][gl];
#pragma unroll
            for (int q = 1; q < NG; ++q) { const float4 pq = part[q][gl]; t.x += pq.x; t.y += pq.y; t.z += pq.z; t.w += pq.w; }
            finish(t, r, true);
        }
    }
}

// scalar fallback (k not a multiple of 4): G lanes per row, one column per lane and pass
template <int G>
__global__ __launch_bounds__(256) void spmm_csr_kernel(int n_rows, const int* __restrict__ rowptr, const int* __restrict__ col,
                                                       const float* __restrict__ val, const float* __restrict__ X, int k,
                                                       const float* __restrict__ bias, int relu, float* __restrict__ Y) {
    const int gid = (blockIdx.x * 256 + threadIdx.x) / G, gl = threadIdx.x % G;
    if (gid >= n_rows) return;
    const int b = rowptr[gid], e = rowptr[gid + 1];
    for (int c0 = 0; c0 < k; c0 += G) {
        const int c = c0 + gl;
        float acc = 0.0f;
        if (c < k) {
            int j = b;
            for (; j + 1 < e; j += 2) {     // two independent gathers in flight
                const float x0 = X[(size_t)col[j] * k + c], x1 = X[(size_t)col[j + 1] * k + c];
                acc += val[j] * x0;
                acc += val[j + 1] * x1;
            }
            if (j < e) acc += val[j] * X[(size_t)col[j] * k + c];
            if (bias) acc += bias[c];
            if (relu & 1) acc = acc > 0.0f ? acc : 0.0f;
            Y[(size_t)gid * k + c] = acc;
        }
    }
}

// emb.renorm_(2, 0, 1): rows with ||row||_2 > 1 are scaled by 1 / (norm + 1e-7)
__global__ void renorm_rows_kernel(int n_rows, int k, float* __restrict__ emb) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    float* row = emb + (size_t)r * k;
    float s = 0.0f;
    for (int c = 0; c < k; ++c) s += row[c] * row[c];
    const float nrm = sqrtf(s);
    if (nrm > 1.0f) {
        const float sc = 1.0f / (nrm + 1e-7f);
        for (int c = 0; c < k; ++c) row[c] *= sc;
    }
}

// Net.decode after the renorm (baselines/TLCGNN.py:52-61), emb 16 / image 25, on the f32 MFMA (round 5; rounds 1-4: eight lanes per
// pair with W1 staged in LDS by each of 2 355 workgroups behind a barrier, 15.9 us).  One LANE per pair, 64 pairs per wavefront:
// the hidden layer is 41 rank-one updates  H[32 x 64] += W1[:, k] (x) X[k, :]  of v_mfma_f32_32x32x1_2b_f32 (two 32 x 32 blocks:
// rows = hidden units, columns = the pairs of lanes 0..31 / 32..63).  A K = 1 instruction is one fused multiply-add per element, so
// hidden unit o of a pair is accumulated as  b1[o], += W1[o][0] x[0], += W1[o][1] x[1], ...  -- the reference's (and the earlier
// kernel's) order, bit for bit, which a K = 4 tile would not give (the G10 fixture holds image rows of 1e2..1e3 whose terms cancel:
// another order moves the result by more than the 1e-5 bar).  Operands, with no LDS at all:
//   * B operand k of lane l = input k of ITS pair: (emb[u] - emb[v])^2 from eight 16-byte gathers, the image row from 100
//     contiguous bytes;
//   * A operand k of lane l = W1[l % 32][k] (zero for rows >= 25), 41 registers loaded once per wavefront from the 4 KB matrix;
//   * accumulators start at b1.  A block's 32 x 32 result is spread over the two half-wavefronts (rows 8 a + 4 half + b); sixteen
//     v_permlane32_swap leave every lane with all 32 rows of its own pair; LeakyReLU, W2 . h in the earlier kernel's summation
//     tree (eight strided partial sums folded 4, 2, 1), |.|, clamp, Fermi-Dirac, one coalesced 256-byte store per wavefront.
// PT: the image table's type -- float when the caller cast it once at set-up (what the reference's torch.Tensor(PI) does on every
// decode, :52-53), double for the raw output of tlc_pd_pi_batch.
typedef float f32x32 __attribute__((ext_vector_type(32)));
template <typename PT>
__global__ __launch_bounds__(256) void lp_decode_mfma_kernel(long long n_pairs, const int* __restrict__ pairs, const float* __restrict__ emb,
                                                             const PT* __restrict__ pi, const float* __restrict__ W1,
                                                             const float* __restrict__ b1, const float* __restrict__ W2,
                                                             const float* __restrict__ b2, float* __restrict__ prob) {
    constexpr int ED = 16, PD = 25, IN = ED + PD;
    const int lane = threadIdx.x & 63, hrow = lane & 31, hf = lane >> 5;
    const long long wave = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long n_waves = ((long long)gridDim.x * blockDim.x) >> 6;
    const long long n_tiles = (n_pairs + 63) >> 6;
    float a[IN];
#pragma unroll
    for (int k = 0; k < IN; ++k) {
        const float w = W1[(hrow < PD ? hrow : 0) * IN + k];
        a[k] = hrow < PD ? w : 0.0f;
    }
    f32x32 acc0;                                                    // register v of the result: row 8 ((v % 16) / 4) + 4 half + v % 4 of block v / 16
#pragma unroll
    for (int v = 0; v < 32; ++v) {
        const int o = 8 * ((v & 15) >> 2) + 4 * hf + (v & 3);
        const float bv = b1[o < PD ? o : 0];
        acc0[v] = o < PD ? bv : 0.0f;
    }
    const float bias2 = b2[0];
    for (long long tile = wave; tile < n_tiles; tile += n_waves) {
        const long long i = tile * 64 + lane;
        const bool live = i < n_pairs;
        const long long ic = live ? i : n_pairs - 1;              // (pairs past the end: computed on the last pair, not stored)
        const int2 uv = *reinterpret_cast<const int2*>(pairs + 2 * ic);
        const float4* eu = reinterpret_cast<const float4*>(emb + (size_t)uv.x * ED);
        const float4* ev = reinterpret_cast<const float4*>(emb + (size_t)uv.y * ED);
        const PT* pr = pi + (size_t)ic * PD;
        float x[IN];
#pragma unroll
        for (int q = 0; q < 4; ++q) {                              // (emb_in - emb_out).pow(2)  (:57)
            const float4 ea = eu[q], eb = ev[q];
            x[4 * q] = (ea.x - eb.x) * (ea.x - eb.x); x[4 * q + 1] = (ea.y - eb.y) * (ea.y - eb.y);
            x[4 * q + 2] = (ea.z - eb.z) * (ea.z - eb.z); x[4 * q + 3] = (ea.w - eb.w) * (ea.w - eb.w);
        }
#pragma unroll
        for (int c = 0; c < PD; ++c) x[ED + c] = (float)pr[c];      // torch.Tensor(PI): float64 -> float32 (:52-53)
        // every load of the tile is requested before the first MFMA (left alone the scheduler dealt the loads out between the
        // MFMAs, eight exposed round trips in a row: 15 us)
        __builtin_amdgcn_sched_barrier(0);
        f32x32 acc = acc0;
#pragma unroll
        for (int k = 0; k < IN; ++k) acc = __builtin_amdgcn_mfma_f32_32x32x1f32(a[k], x[k], acc, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // lower half keeps block 0 (its own pairs), upper half block 1: each sends the other half's rows across (one
        // v_permlane32_swap per register pair; the two-operand form of the builtin was miscompiled by this toolchain -- every
        // element read from acc[0] -- so the exchange goes through the one-operand helper and two selects)
        float hr[32];                                              // (scalars: element inserts into the 32-wide vector copied all of it)
#pragma unroll
        for (int v = 0; v < 32; ++v) hr[v] = acc[v];
#pragma unroll
        for (int w = 0; w < 16; ++w) {
            const float keep = hf ? hr[16 + w] : hr[w], send = hf ? hr[w] : hr[16 + w];
            const float recv = lane_xor_f32<32>(send);
            hr[w] = hf ? recv : keep;
            hr[16 + w] = hf ? keep : recv;
        }
        // now hr[w], w < 16, is row 8 (w / 4) + w % 4 of the lane's own pair and hr[16 + w] row 8 (w / 4) + 4 + w % 4
        float pg[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float part = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int o = g + 8 * q;
                if (o < PD) {
                    float h = hr[(g < 4 ? 0 : 16) + 4 * q + (g & 3)];
                    h = h > 0.0f ? h : 0.2f * h;                  // LeakyReLU(0.2) (:58)
                    part += W2[o] * h;
                }
            }
            pg[g] = part;
        }
        const float s40 = pg[0] + pg[4], s41 = pg[1] + pg[5], s42 = pg[2] + pg[6], s43 = pg[3] + pg[7];
        const float s20 = s40 + s42, s21 = s41 + s43;
        if (live) {
            float d = bias2 + (s20 + s21);
            d = fabsf(d);                                         // :59
            d = d < 0.0f ? 0.0f : (d > 40.0f ? 40.0f : d);        // clamp (:60)
            prob[i] = 1.0f / (expf((d - 2.0f) / 1.0f) + 1.0f);    // Fermi-Dirac (:61)
        }
    }
}

// generic fallback for other dimensions: one thread per pair, everything from global
template <typename PT>
__global__ __launch_bounds__(256) void lp_decode_generic_kernel(long long n_pairs, const int* __restrict__ pairs,
                                                                const float* __restrict__ emb, int ED, const PT* __restrict__ pi,
                                                                int PD, const float* __restrict__ W1, const float* __restrict__ b1,
                                                                const float* __restrict__ W2, const float* __restrict__ b2,
                                                                float* __restrict__ prob) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_pairs) return;
    const int u = pairs[2 * i], v = pairs[2 * i + 1];
    const float* eu = emb + (size_t)u * ED;
    const float* ev = emb + (size_t)v * ED;
    const PT* pr = pi + (size_t)i * PD;
    float d = b2[0];
    for (int o = 0; o < PD; ++o) {
        const float* wr = W1 + (size_t)o * (ED + PD);
        float h = b1[o];
        for (int c = 0; c < ED; ++c) { const float t = eu[c] - ev[c]; h += wr[c] * (t * t); }
        for (int c = 0; c < PD; ++c) h += wr[ED + c] * (float)pr[c];
        h = h > 0.0f ? h : 0.2f * h;
        d += W2[o] * h;
    }
    d = fabsf(d);
    d = d < 0.0f ? 0.0f : (d > 40.0f ? 40.0f : d);
    prob[i] = 1.0f / (expf((d - 2.0f) / 1.0f) + 1.0f);
}

}  // namespace

// ======================================================================================================================
// C ABI
// ======================================================================================================================
// the launches of the CSR-by-target build; tmp int32[3 n + E] = [cnt n | cursor n | unsorted col E + n]; d_val NULL: structure only
static int gcn_csr_launch(int32_t n_nodes, int64_t n_edges, const int64_t* d_edge_index, int32_t* d_rowptr, int32_t* d_col, float* d_val,
                          int32_t* d_nnz, int* tmp, hipStream_t s) {
    // tmp: [n] counts | [n] unused | [E + n] entries as they arrive.  The counts are dead once the row pointers exist: their words
    // then hold the list of the rows above GCN_SHORT_ROW entries (word 0: how many; zeroed by the fill kernel, which runs behind
    // the scans and ahead of the kernel that appends).  An edge's place in its row (what its counting atomic returned: E words)
    // waits in d_col from word n on (d_col has E + n words and is written for good only by the finish kernels; its head is the
    // scan's scratch).
    if (hipMemsetAsync(tmp, 0, (size_t)n_nodes * sizeof(int), s) != hipSuccess) return TLC_ERR_HIP;
    int* long_rows = tmp;
    const int long_cap = n_nodes - 1;
    int* raw = tmp + 2 * (size_t)n_nodes;
    int* place = d_col + n_nodes;                             // (E words behind the scan's scratch: d_col holds E + n)
    const int eb = (int)((n_edges + 255) / 256), nb = (n_nodes + 255) / 256;
    if (n_edges) hipLaunchKernelGGL(gcn_count_kernel, dim3(eb), dim3(256), 0, s, (long long)n_edges, (const long long*)d_edge_index, n_nodes, tmp, place);
    if (n_nodes <= 8192) {
        hipLaunchKernelGGL(gcn_scan_kernel, dim3(1), dim3(1024), 0, s, n_nodes, (const int*)tmp, d_rowptr, d_nnz);
    } else {
        // block totals in the head of d_col (written for good only by gcn_finish_kernel, after the fill)
        const int sb = (n_nodes + 1023) / 1024;
        hipLaunchKernelGGL(gcn_scan_block_kernel, dim3(sb), dim3(1024), 0, s, n_nodes, (const int*)tmp, d_rowptr, d_col);
        hipLaunchKernelGGL(gcn_scan_top_kernel, dim3(1), dim3(1024), 0, s, sb, d_col, d_rowptr, d_nnz);
        hipLaunchKernelGGL(gcn_scan_add_kernel, dim3(sb), dim3(1024), 0, s, n_nodes, d_rowptr, (const int*)d_col);
    }
    if (n_edges) hipLaunchKernelGGL(gcn_fill_kernel, dim3(eb), dim3(256), 0, s, (long long)n_edges, (const long long*)d_edge_index, n_nodes, (const int*)d_rowptr, (const int*)place, raw, long_rows);
    hipLaunchKernelGGL(gcn_finish_short_kernel, dim3(nb), dim3(256), 0, s, n_nodes, (const int*)d_rowptr, (const int*)raw, d_col, long_rows, long_cap);
    const int fin_wg = std::min((n_nodes + 3) / 4, 2048);
    hipLaunchKernelGGL(gcn_finish_kernel, dim3(fin_wg), dim3(256), 0, s, n_nodes, (const int*)d_rowptr, (const int*)raw, d_col, (const int*)long_rows, long_cap);
    if (d_val) hipLaunchKernelGGL(gcn_val_kernel, dim3(nb), dim3(256), 0, s, n_nodes, (const int*)d_rowptr, (const int*)d_col, d_val);
    return hipGetLastError() == hipSuccess ? TLC_OK : TLC_ERR_HIP;
}

extern "C" int tlc_gcn_norm_csr(int32_t n_nodes, int64_t n_edges, const int64_t* d_edge_index, int32_t* d_rowptr,
                                int32_t* d_col, float* d_val, int32_t* d_nnz, void* stream) {
    TLC_REQUIRE(n_nodes > 0 && n_edges >= 0, "bad sizes");
    TLC_REQUIRE(d_rowptr && d_col && d_val && (n_edges == 0 || d_edge_index), "null pointer");
    hipStream_t s = (hipStream_t)stream;
    int* tmp = nullptr;
    TLC_HIP_CHECK(hipMalloc(&tmp, (3 * (size_t)n_nodes + (size_t)n_edges) * sizeof(int)));
    const int rc = gcn_csr_launch(n_nodes, n_edges, d_edge_index, d_rowptr, d_col, d_val, d_nnz, tmp, s);
    hipStreamSynchronize(s);      // cached=True: one-off preprocessing; the temporaries must outlive the kernels
    hipFree(tmp);
    if (rc != TLC_OK) tlc_set_error("tlc_gcn_norm_csr: HIP failure");
    return rc;
}

// The structure alone, per BATCH (Knowledge_Distillation/gat_conv.py:146-152 runs remove_self_loops + add_self_loops on every forward):
// the caller brings the temporaries, nothing is allocated and nothing waits for the stream.
extern "C" int tlc_csr_by_target(int32_t n_nodes, int64_t n_edges, const int64_t* d_edge_index, int32_t* d_rowptr, int32_t* d_col,
                                 int32_t* d_nnz, int32_t* d_work, void* stream) {
    TLC_REQUIRE(n_nodes > 0 && n_edges >= 0, "bad sizes");
    TLC_REQUIRE(d_rowptr && d_col && d_nnz && d_work && (n_edges == 0 || d_edge_index), "null pointer");
    const int rc = gcn_csr_launch(n_nodes, n_edges, d_edge_index, d_rowptr, d_col, nullptr, d_nnz, d_work, (hipStream_t)stream);
    if (rc != TLC_OK) tlc_set_error("tlc_csr_by_target: HIP failure");
    return rc;
}

extern "C" int tlc_gemm_f32(int32_t M, int32_t N, int32_t K, const float* d_A, const float* d_B, const float* d_bias, int relu,
                            float* d_C, void* stream) {
    TLC_REQUIRE(M >= 0 && N > 0 && K > 0, "bad sizes");
    TLC_REQUIRE(N <= 128, "tlc_gemm_f32 supports N <= 128 (GCN hidden sizes)");
    if (M == 0) return TLC_OK;
    TLC_REQUIRE(d_A && d_B && d_C, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    TLC_REQUIRE((long long)K * N * 4 < (1ll << 31) && (long long)G16_BM * K * 4 < (1ll << 31), "K too large for the 32-bit buffer offsets");
    const bool vec = (K % 4) == 0 && (N % 4) == 0 &&
                     ((reinterpret_cast<uintptr_t>(d_A) | reinterpret_cast<uintptr_t>(d_B) | reinterpret_cast<uintptr_t>(d_C) |
                       reinterpret_cast<uintptr_t>(d_bias)) & 15) == 0;
    // skinny K (the PDGNN shapes) with enough rows to stream: B in registers, A straight from global memory
    if (M >= 4096 && (K == 16 || K == 32 || K == 64) && (K / 4) * ((N + 15) / 16) <= 80 &&
        (reinterpret_cast<uintptr_t>(d_A) & 15) == 0) {
        bool done = false;
        if (K == 16) done = gemm_skinny_launch<4>(M, N, d_A, d_B, d_bias, relu, d_C, s);
        else if (K == 32) done = gemm_skinny_launch<8>(M, N, d_A, d_B, d_bias, relu, d_C, s);
        else done = gemm_skinny_launch<16>(M, N, d_A, d_B, d_bias, relu, d_C, s);
        if (done) {
            TLC_HIP_CHECK(hipGetLastError());
            return TLC_OK;
        }
    }
    hipError_t le = hipSuccess;
    // enough K for phases of 128 and 16-byte operands: B resident in LDS, no barrier inside the K loop (TLC_GEMM_BRES=0: the chunked kernel)
    static const bool bres_on = !(getenv("TLC_GEMM_BRES") && getenv("TLC_GEMM_BRES")[0] == '0');
    if (bres_on && vec && K >= 128 && (long long)G16_BM * K * 4 < (1ll << 31)) {
#define TLC_GBR(NT_) case NT_: le = gemm_bres_launch<NT_>(M, N, K, d_A, d_B, d_bias, relu, d_C, s); break;
        switch ((N + 15) / 16) {
            TLC_GBR(1) TLC_GBR(2) TLC_GBR(3) TLC_GBR(4) TLC_GBR(5) TLC_GBR(6) TLC_GBR(7) TLC_GBR(8)
            default: break;
        }
#undef TLC_GBR
        TLC_HIP_CHECK(le);
        TLC_HIP_CHECK(hipGetLastError());
        return TLC_OK;
    }
#define TLC_G16(NT_)                                                                              \
    case NT_:                                                                                     \
        le = vec ? gemm16_launch<NT_, true, 2, 32>(M, N, K, d_A, d_B, d_bias, relu, d_C, s)       \
                 : gemm16_launch<NT_, false, 2, 32>(M, N, K, d_A, d_B, d_bias, relu, d_C, s);     \
        break;
    switch ((N + 15) / 16) {
        TLC_G16(1) TLC_G16(2) TLC_G16(3) TLC_G16(4) TLC_G16(5) TLC_G16(6) TLC_G16(7) TLC_G16(8)
        default: break;
    }
#undef TLC_G16
    TLC_HIP_CHECK(le);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// ---- C = CSR(A) @ B (+bias)(ReLU): the feature projection x @ W when x is sparse --------------------------------------------
// PubMed's TF-IDF features are 90 % zeros (Cora's 98.7 %): the product over the stored entries alone does a tenth of the flops
// and bytes of the dense GEMM.  B (K x N, the weight matrix: 200 KB for 500 x 100) is what every entry gathers from, so a column
// slice of it lives in LDS (one workgroup per CU), and a wavefront owns a row of A at a time.  gridDim.y = column slices.
// HBM: the CSR once per slice (8 B per entry), B once per workgroup (from L2), C once.
// (Round 2's kernel -- lanes = the 64 columns of a slice, the row's entries handed round with v_readlane, one entry per step:
// four vector instructions and a 256-byte LDS read per entry and slice, 1.97 M such steps for PubMed = 31 M SIMD cycles of
// vector issue alone -- measured 38 us against the dense MFMA kernel's 28.5 and was used below 5 % density only; replaced by:)
// Four entries of a row at a time (round 5): the four 16-lane rows of the wavefront work on four entries at once: lane (g, l) holds the columns 4 l .. 4 l + 3
// of the slice (a slice is `sw` <= 64 columns, a multiple of four: 52 + 48 for N = 100, K sw 4 B = 104 KB of LDS) and adds up the
// entries  j = g (mod 4)  of the row.  A chunk of 64 entries is loaded coalesced (lane j: entry j) and turned round once (two
// ds_bpermute) so that lane (g, l) holds entry 4 l + g: step i of the chunk then needs, in every 16-lane row, what that row's
// lane i holds -- one DPP row broadcast folded into the address add (v_add_u32_dpp row_newbcast:i), one for the value, one
// ds_read_b128 and two v_pk_fma_f32 per FOUR entries: 4 vector instructions and a 1 KB LDS read where round 2's kernel needed 16
// and four 256-byte reads.  (Loading the chunk already turned round -- lane (g, l) asks for entry 4 l + g -- was measured first:
// every 16-lane quarter of such a load touches all of the chunk's cache lines, four times the tag look-ups, and the wavefronts
// waited 2 000 cycles per row for their entries.)  The four partial sums of a row meet without LDS: v_permlane16_swap on (x, y)
// and (z, w), v_permlane32_swap on the two sums -- three swaps, three adds -- leave column 4 l + g of the row in lane (g, l), and
// the row goes out as one dword per lane, 4 sw contiguous bytes.  Sum order: entries j = g (mod 4) ascending per partial sum,
// then (g0 + g1) + (g2 + g3) -- fixed, not the dense kernel's.
// A row whose entry count is not a multiple of four ends in a step with idle 16-lane rows: their value is 0 and their column the
// row's first entry's, so the only weights multiplied by that zero are ones the row uses anyway (finite weights: exact; a
// non-finite weight the row touches gives a non-finite output either way, possibly NaN where the sum is an infinity).
//
// What bounds it (cycle stamps per wavefront, TLC_SQ_DEBUG): the LDS reads of the slice -- 13 ds_read_b128 of 1 KB per row of 50
// entries, 16 wavefronts x 10 rows per CU = 16 000 cycles of the LDS pipe at 128 B per clock -- then the slice's way into LDS
// (26 MB from L2 over all CUs, asked for with buffer_load ... lds: no registers, no ds_write, every piece in flight at once).
// The entries of a row are requested four rows ahead into a ring of four register pairs.  The loads are issued from inline
// assembly and the wait before a pair's first use is written by hand (vmcnt(6): vector memory loads return in order, the six
// loads of the three rows requested since may still be in flight; anything else issued in between -- the row stores, a long
// row's later chunks -- only makes the count conservative): the compiler's own s_waitcnt bookkeeping was tried first, in three
// loop shapes, and always ended up waiting for all but the newest two loads.  Every path through the row loop issues the same
// requests into the same registers (a request under a branch gets routed through a temporary the compiler copies from before
// the data is there): a block's rows are rounded up to a multiple of four with empty ones, and loads have no branch around them
// -- a lane without an entry asks for an offset behind the descriptor's range and gets zero.
#ifndef TLC_SPGEMM_NW
#define TLC_SPGEMM_NW 16
#endif
#ifdef TLC_SQ_DEBUG
__device__ unsigned long long g_sq_dbg[8];                   // cycle sums over all wavefronts
#define SQ_CLK() clock64()
#else
#define SQ_CLK() 0ull
#endif
template <int I>
__device__ __forceinline__ int row_bcast(int x) { return __builtin_amdgcn_update_dpp(0, x, 0x150 + I, 0xf, 0xf, true); }

template <int NW>
__global__ __launch_bounds__(NW * 64, 1) void spgemm_quad_kernel(int M, int K, int N, int sw, const int* __restrict__ rowptr,
                                                                const int* __restrict__ col, const float* __restrict__ val,
                                                                const float* __restrict__ B, const float* __restrict__ bias,
                                                                int relu, float* __restrict__ C) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sq_lds[];
    float* Bs = (float*)sq_lds;                                   // [K][sw] (+ 1 KB: the last piece of the staging, zeros)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, l = lane & 15;
    const int c0 = (int)blockIdx.y * sw;
    const int cw = N - c0 < sw ? N - c0 : sw;
    const int rb = sw * 4;                                        // bytes of a row of the slice
    const int turn = (4 * l + g) * 4;                             // ds_bpermute address: the lane holding entry 4 l + g
    const int loff = l * 16;
    const int stride = (int)gridDim.x * NW;
    const int r_first = (int)blockIdx.x * NW + wave;
    // the rows of this wavefront are r_first + i stride: lane i keeps the bounds of row i (of a block of 64 rows)
#if defined(TLC_SQ_DIAG) && (TLC_SQ_DIAG & 1)                    /* (diagnostic builds: 1 no rows, 2 no staging, 4 no row stores) */
    const int n_mine = 0;
#else
    const int n_mine = r_first < M ? (M - r_first + stride - 1) / stride : 0;
#endif
    unsigned long long dbg_t0 = SQ_CLK(), dbg_wait = 0, dbg_chunk = 0, dbg_tail = 0, dbg_stage = 0;
    (void)dbg_t0; (void)dbg_wait; (void)dbg_chunk; (void)dbg_tail; (void)dbg_stage;
    int rb0 = 0, rb1 = 0;
    if (lane < n_mine) { rb0 = rowptr[r_first + lane * stride]; rb1 = rowptr[r_first + lane * stride + 1]; }
    constexpr int OOB = 0x7ffffff0;
    typedef int sq_v4i __attribute__((ext_vector_type(4)));
    const unsigned long long a_col = (unsigned long long)col, a_val = (unsigned long long)val;
    const sq_v4i rs_col = {(int)(unsigned)a_col, (int)((unsigned)(a_col >> 32) & 0xffffu), OOB, 0x00020000};
    const sq_v4i rs_val = {(int)(unsigned)a_val, (int)((unsigned)(a_val >> 32) & 0xffffu), OOB, 0x00020000};
    auto request = [&](int i, int& k, int& v) __attribute__((always_inline)) {   // first chunk of row i of the block (uniform i; rows >= 64 or without bounds: empty)
        const int b0 = __builtin_amdgcn_readlane(rb0, i & 63), b1 = __builtin_amdgcn_readlane(rb1, i & 63);
        const int off = (i < 64 && lane < b1 - b0) ? (b0 + lane) * 4 : OOB;
        // ("+v": the destination is the ring register itself, not a temporary the compiler would copy from before the data is there)
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(k) : "v"(off), "s"(rs_col) : "memory");
        asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(v) : "v"(off), "s"(rs_val) : "memory");
    };
    int k0 = 0, k1 = 0, k2 = 0, k3 = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0;       // the ring (value bits in v*)
    request(0, k0, v0); request(1, k1, v1); request(2, k2, v2); request(3, k3, v3);

    if ((N & 3) == 0 && (((uintptr_t)B) & 15) == 0) {
        // The slice into LDS without registers: LDS is [k][sw] = 16-byte items in order, item id = k (sw/4) + q, and a piece is 64
        // consecutive items (one buffer_load_dwordx4 ... lds: lane j's 16 bytes land at piece base + 16 j).  Items behind the
        // slice, and the columns >= N of the last slice, ask for an offset out of range: zeros.
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, K * N * 4, 0x00020000);
        const int q4 = sw >> 2, items = K * q4;
        const unsigned inv = (1u << 20) / (unsigned)q4 + 1u;      // id / q4 = (id inv) >> 20 for id q4 < 2^20 (items <= 10 240 fit LDS, q4 <= 16);
                                                                  // the product in 64 bits: q4 = 1 or 2 (N = 4, 8) puts inv near 2^20 and id above 4 095
        const int npieces = (items + 63) >> 6;
#if defined(TLC_SQ_DIAG) && (TLC_SQ_DIAG & 2)
        for (int pc = wave; pc < 0; pc += NW) {
#else
        for (int pc = wave; pc < npieces; pc += NW) {
#endif
            const int id = pc * 64 + lane;
            const int k = (int)(((unsigned long long)(unsigned)id * inv) >> 20), q = id - k * q4;
            const int goff = (id < items && 4 * q < cw) ? (k * N + c0 + 4 * q) * 4 : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (__attribute__((address_space(3))) void*)(sq_lds + pc * 1024), 16, goff, 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        for (int t = tid; t < K * 64; t += NW * 64) {
            const int k = t >> 6, c = t & 63;
            if (c < sw) Bs[k * sw + c] = c < cw ? B[(size_t)k * N + c0 + c] : 0.f;
        }
        if (tid < 16) Bs[K * sw + tid] = 0.f;
    }
    __syncthreads();
    dbg_stage = SQ_CLK() - dbg_t0;
    const int my_col = 4 * l + g;                                 // the column of the slice this lane stores
    const float bvs = (bias && my_col < cw) ? bias[c0 + my_col] : 0.f;
    const unsigned char* bsb = sq_lds + loff;

    // one chunk (<= 64 entries, lane (g, l) holds entry 4 l + g as (column, value); lanes without one: a used column, value 0)
    auto do_chunk = [&](int kk, int vvi, int cnt, float4& acc) __attribute__((always_inline)) {
        kk *= rb;                                                 // column -> byte offset of its row of the slice
        const int steps = (cnt + 3) >> 2;
#define TLC_SQ_LOAD(I, b_, x_)                                                                                     \
        const float4 b_ = *reinterpret_cast<const float4*>(bsb + row_bcast<I>(kk));                                \
        const float x_ = __builtin_bit_cast(float, row_bcast<I>(vvi));
#define TLC_SQ_FMA(b_, x_)                                                                                         \
        acc.x = fmaf(x_, b_.x, acc.x); acc.y = fmaf(x_, b_.y, acc.y);                                              \
        acc.z = fmaf(x_, b_.z, acc.z); acc.w = fmaf(x_, b_.w, acc.w);
#define TLC_SQ_STEP(I) { TLC_SQ_LOAD(I, b0_, x0_) TLC_SQ_FMA(b0_, x0_) }
        // four steps (16 entries) per block: the four LDS reads of a block are in flight together
#define TLC_SQ_BLOCK(J)                                                                                            \
        if (steps >= 4 * J + 4) {                                                                                  \
            TLC_SQ_LOAD(4 * J, b0_, x0_) TLC_SQ_LOAD(4 * J + 1, b1_, x1_)                                          \
            TLC_SQ_LOAD(4 * J + 2, b2_, x2_) TLC_SQ_LOAD(4 * J + 3, b3_, x3_)                                      \
            TLC_SQ_FMA(b0_, x0_) TLC_SQ_FMA(b1_, x1_) TLC_SQ_FMA(b2_, x2_) TLC_SQ_FMA(b3_, x3_)                    \
        } else {                                                                                                   \
            if (steps > 4 * J) TLC_SQ_STEP(4 * J)                                                                  \
            if (steps > 4 * J + 1) TLC_SQ_STEP(4 * J + 1)                                                          \
            if (steps > 4 * J + 2) TLC_SQ_STEP(4 * J + 2)                                                          \
            break;                                                                                                 \
        }
        do { TLC_SQ_BLOCK(0) TLC_SQ_BLOCK(1) TLC_SQ_BLOCK(2) TLC_SQ_BLOCK(3) } while (0);
#undef TLC_SQ_BLOCK
#undef TLC_SQ_STEP
#undef TLC_SQ_FMA
#undef TLC_SQ_LOAD
    };
    // a chunk as loaded (lane j: entry j) turned round: lane (g, l) gets entry 4 l + g; lanes without an entry the first one's column
    auto turn_round = [&](int& kk, int& vvi, int cnt) __attribute__((always_inline)) {
        const int kfirst = __builtin_amdgcn_readfirstlane(kk);
        kk = __builtin_amdgcn_ds_bpermute(turn, kk);
        vvi = __builtin_amdgcn_ds_bpermute(turn, vvi);
        if (my_col >= cnt) kk = kfirst;                           // (my_col = 4 l + g is also this lane's entry number)
    };

    // one row: its first chunk in the ring pair (kr, vr), requested four rows ago; the pair then takes row i + 4
    auto do_row = [&](int r, int i, int& kr, int& vr) __attribute__((always_inline)) {
        const unsigned long long ta = SQ_CLK();
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(kr), "+v"(vr) : : "memory");
        const unsigned long long tb = SQ_CLK();
        int kk = kr, vvi = vr;
        asm volatile("" : "+v"(kk), "+v"(vvi) : "v"(kr), "v"(vr) : "memory");   // (the copies are made here, before the pair is requested again)
        request(i + 4, kr, vr);
        const int e0 = __builtin_amdgcn_readlane(rb0, i), e1 = __builtin_amdgcn_readlane(rb1, i);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int base = e0; base < e1; base += 64) {
            const int cnt = e1 - base < 64 ? e1 - base : 64;
            if (base != e0) {                                     // (rows longer than 64 entries: the later chunks on demand)
                const int off = lane < cnt ? (base + lane) * 4 : OOB;
                asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(kk) : "v"(off), "s"(rs_col) : "memory");
                asm volatile("buffer_load_dword %0, %1, %2, 0 offen" : "+v"(vvi) : "v"(off), "s"(rs_val) : "memory");
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(kk), "+v"(vvi) : : "memory");
            }
            turn_round(kk, vvi, cnt);
#ifdef TLC_SQ_DUMP
            if (r == 0) { C[64 + lane] = __builtin_bit_cast(float, vvi); C[128 + lane] = (float)kk; }
#endif
            do_chunk(kk, vvi, cnt, acc);
        }
#ifdef TLC_SQ_DUMP
        if (r == 0) { C[192 + lane] = acc.x; C[256 + lane] = acc.y; }
#endif
#ifdef TLC_SQ_DEBUG
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w) : : "memory");
#endif
        const unsigned long long tc = SQ_CLK();
        dbg_wait += tb - ta; dbg_chunk += tc - tb;
        if (r < M) {                                              // (uniform; the padding rows of a trip have no row)
            // rows of 16 lanes: [x | y] -> x' = (x0, y0, x2, y2), y' = (x1, y1, x3, y3); the same for [z | w]; then the halves.
            // (From assembly: with __builtin_amdgcn_permlane16_swap this compiler adds the FIRST result to itself -- v_add v, v9, v9.
            // The s_nop cover the wait states between a vector write and a swap that the compiler would otherwise insert.)
            asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 1"
                         : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w));
            float s_xy = acc.x + acc.y;                           // (x01, y01, x23, y23)
            float s_zw = acc.z + acc.w;                           // (z01, w01, z23, w23)
            asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(s_xy), "+v"(s_zw));
            float y = (s_xy + s_zw) + bvs;                        // (x, y, z, w): column 4 l + g
            if (relu & 1) y = y > 0.f ? y : 0.f;
#if defined(TLC_SQ_DIAG) && (TLC_SQ_DIAG & 4)
            if (my_col < cw && y == 123.456f) C[(size_t)r * N + c0 + my_col] = y;
#else
            if (my_col < cw) C[(size_t)r * N + c0 + my_col] = y;
#endif
        }
        dbg_tail += SQ_CLK() - tc;
    };

    for (int blk = 0; blk < n_mine; blk += 64) {                  // (more than 64 rows per wavefront: the bounds 64 rows at a time)
        const int nrows = n_mine - blk < 64 ? n_mine - blk : 64;
        const int rbase = r_first + blk * stride;
#pragma unroll 1
        for (int i = 0; i < nrows; i += 4) {
#define TLC_SQ_ROW(J, kr, vr) do_row(i + J < nrows ? rbase + (i + J) * stride : M, i + J, kr, vr);
            TLC_SQ_ROW(0, k0, v0) TLC_SQ_ROW(1, k1, v1) TLC_SQ_ROW(2, k2, v2) TLC_SQ_ROW(3, k3, v3)
#undef TLC_SQ_ROW
        }
        // the next block's bounds and first requests, needed or not (see above)
        const int nb = rbase + 64 * stride;
        const int left = n_mine - blk - 64;
        rb0 = rb1 = 0;
        if (lane < left) { rb0 = rowptr[nb + lane * stride]; rb1 = rowptr[nb + lane * stride + 1]; }
        request(0, k0, v0); request(1, k1, v1); request(2, k2, v2); request(3, k3, v3);
    }
    // (the ring's last requests -- rows behind the end: empty -- land before the wavefront ends)
    asm volatile("s_waitcnt vmcnt(0)" : : : "memory");
#ifdef TLC_SQ_DEBUG
    if (lane == 0) {
        const unsigned long long whole = SQ_CLK() - dbg_t0;
        atomicAdd(&g_sq_dbg[0], dbg_stage); atomicAdd(&g_sq_dbg[1], dbg_wait); atomicAdd(&g_sq_dbg[2], dbg_chunk);
        atomicAdd(&g_sq_dbg[3], dbg_tail); atomicAdd(&g_sq_dbg[4], whole); atomicAdd(&g_sq_dbg[5], (unsigned long long)n_mine);
        atomicAdd(&g_sq_dbg[7], 1ull);
    }
#endif
}

extern "C" int tlc_spgemm_csr_dense_f32(int32_t M, int32_t K, int32_t N, const int32_t* d_rowptr, const int32_t* d_col,
                                        const float* d_val, const float* d_B, const float* d_bias, int relu, float* d_C,
                                        void* stream) {
    TLC_REQUIRE(M >= 0 && K > 0 && N > 0, "bad sizes");
    if (M == 0) return TLC_OK;
    TLC_REQUIRE(d_rowptr && d_col && d_val && d_B && d_C, "null pointer");
    // column slices of equal width, a multiple of four and at most 64 (N = 100: 52 + 48)
    const int slices = (N + 63) / 64;
    const int sw = (((N + slices - 1) / slices) + 3) & ~3;
    constexpr int NW = TLC_SPGEMM_NW;
    const size_t lds = (size_t)K * sw * sizeof(float) + 1024;
    if (lds > 160 * 1024) {
        tlc_set_error("tlc_spgemm_csr_dense_f32: K = %d needs %zu B of LDS per %d-column slice (max 160 KiB)", K, lds, sw);
        return TLC_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    int gx = 256 / slices;                           // one round of workgroups over all slices
    if (gx < 1) gx = 1;
    const int need = (M + NW - 1) / NW;
    if (gx > need) gx = need;
    // (per call: the attribute is per device and a process may drive several; a host-side call of a few hundred nanoseconds)
    if (lds > 64 * 1024)
        TLC_HIP_CHECK(hipFuncSetAttribute((const void*)spgemm_quad_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(spgemm_quad_kernel<NW>, dim3(gx, slices), dim3(NW * 64), lds, s, M, K, N, sw, d_rowptr, d_col, d_val, d_B,
                       d_bias, relu, d_C);
#ifdef TLC_SQ_DEBUG
    {
        static int calls = 0;
        if (++calls % 50 == 0) {                             // (a report every 50 calls, warm)
            unsigned long long h[8], z[8] = {};
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_sq_dbg), z, sizeof(z));
            hipLaunchKernelGGL(spgemm_quad_kernel<NW>, dim3(gx, slices), dim3(NW * 64), lds, s, M, K, N, sw, d_rowptr, d_col, d_val, d_B,
                               d_bias, relu, d_C);
            (void)hipStreamSynchronize(s);
            (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_sq_dbg), sizeof(h));
            const double w = (double)(h[7] ? h[7] : 1);
            fprintf(stderr, "[spgemm_quad %d x %d x %d] mean cycles per wavefront (%llu wavefronts, %.1f rows each): to the barrier %.0f | rows: "
                    "waiting for entries %.0f, chunks %.0f, butterfly + store %.0f | whole %.0f\n",
                    M, K, N, h[7], h[5] / w, h[0] / w, h[1] / w, h[2] / w, h[3] / w, h[4] / w);
        }
    }
#endif
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

extern "C" int tlc_spmm_csr_f32(int32_t n_rows, const int32_t* d_rowptr, const int32_t* d_col, const float* d_val,
                                const float* d_X, int32_t k, const float* d_bias, int relu, float* d_Y, void* stream) {
    TLC_REQUIRE(n_rows >= 0 && k > 0, "bad sizes");
    if (n_rows == 0) return TLC_OK;
    TLC_REQUIRE(d_rowptr && d_col && d_val && d_X && d_Y, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (k % 4) == 0 && k <= 256 && ((reinterpret_cast<uintptr_t>(d_X) | reinterpret_cast<uintptr_t>(d_Y)) & 15) == 0;
    if (vec && k == 16) {
        hipLaunchKernelGGL(spmm16_kernel, dim3((unsigned)((n_rows + 15) / 16)), dim3(256), 0, s, n_rows, d_rowptr, d_col, d_val, d_X, d_bias, relu, d_Y);
    } else if (vec && k <= 16) {
        hipLaunchKernelGGL(spmm_csr_v4_kernel<4>, dim3((unsigned)(((size_t)n_rows * 4 + 255) / 256)), dim3(256), 0, s, n_rows, d_rowptr, d_col, d_val, d_X, k, d_bias, relu, d_Y);
    } else if (vec && k <= 32) {
        hipLaunchKernelGGL(spmm_csr_v4_kernel<8>, dim3((unsigned)(((size_t)n_rows * 8 + 255) / 256)), dim3(256), 0, s, n_rows, d_rowptr, d_col, d_val, d_X, k, d_bias, relu, d_Y);
    } else if (vec && k <= 64) {
        hipLaunchKernelGGL(spmm_csr_v4_kernel<16>, dim3((unsigned)(((size_t)n_rows * 16 + 255) / 256)), dim3(256), 0, s, n_rows, d_rowptr, d_col, d_val, d_X, k, d_bias, relu, d_Y);
    } else if (vec && k <= 128) {
        hipLaunchKernelGGL(spmm_csr_v4_kernel<32>, dim3((unsigned)(((size_t)n_rows * 32 + 255) / 256)), dim3(256), 0, s, n_rows, d_rowptr, d_col, d_val, d_X, k, d_bias, relu, d_Y);
    } else if (vec) {
        hipLaunchKernelGGL(spmm_csr_v4_kernel<64>, dim3((unsigned)(((size_t)n_rows * 64 + 255) / 256)), dim3(256), 0, s, n_rows, d_rowptr, d_col, d_val, d_X, k, d_bias, relu, d_Y);
    } else if (k <= 16) {
        hipLaunchKernelGGL(spmm_csr_kernel<16>, dim3((n_rows * 16 + 255) / 256), dim3(256), 0, s, n_rows, d_rowptr, d_col, d_val, d_X, k, d_bias, relu, d_Y);
    } else if (k <= 32) {
        hipLaunchKernelGGL(spmm_csr_kernel<32>, dim3((n_rows * 32 + 255) / 256), dim3(256), 0, s, n_rows, d_rowptr, d_col, d_val, d_X, k, d_bias, relu, d_Y);
    } else {
        hipLaunchKernelGGL(spmm_csr_kernel<64>, dim3((n_rows * 64 + 255) / 256), dim3(256), 0, s, n_rows, d_rowptr, d_col, d_val, d_X, k, d_bias, relu, d_Y);
    }
    // the vectorised kernels renormalise the rows in their epilogue (relu & 2); the scalar fallback takes a second pass
    if ((relu & 2) && !vec) hipLaunchKernelGGL(renorm_rows_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, s, n_rows, k, d_Y);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

extern "C" int tlc_renorm_rows_f32(int32_t n_rows, int32_t k, float* d_emb, void* stream) {
    TLC_REQUIRE(n_rows >= 0 && k > 0, "bad sizes");
    if (n_rows == 0) return TLC_OK;
    TLC_REQUIRE(d_emb != nullptr, "null pointer");
    hipLaunchKernelGGL(renorm_rows_kernel, dim3((n_rows + 255) / 256), dim3(256), 0, (hipStream_t)stream, n_rows, k, d_emb);
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

// The whole two-layer encoder of Net.encode in eval mode (baselines/TLCGNN.py:19-26: conv1 -> ReLU -> conv2, dropout off) behind
// ONE call: x@W1, aggregate + b1 + ReLU, h@W2, aggregate + b2 with `flags` as in tlc_spmm_csr_f32 (bit 0 ReLU, bit 1 the renorm_ of
// TLCGNN.py:48) submitted back to back, without five trips through the caller's language between them (a Python host needs ~70 us
// to submit what the device runs in 79 us).  For out_dim == 16 (TLCGNN) the second and third step are ONE kernel (spmm_w2_kernel:
// h@W2 in the aggregate's epilogue, fp32 sums in another order than the separate product's); otherwise the four kernels of the
// separate calls.
// d_ws: (2 * hidden + out_dim) * n floats (+ 12) of scratch.
static int gcn2_encode_impl(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, const float* d_val,
                            const float* d_x, const int32_t* d_xs_rowptr, const int32_t* d_xs_col, const float* d_xs_val,
                            int32_t f_in, const float* d_w1, const float* d_b1, int32_t hidden,
                            const float* d_w2, const float* d_b2, int32_t out_dim, int flags, float* d_ws, float* d_emb,
                            void* stream) {
    auto pad4 = [](size_t v) { return (v + 3) & ~(size_t)3; };
    float* t1 = d_ws;
    float* t2 = t1 + pad4((size_t)n_nodes * hidden);
    float* t3 = t2 + pad4((size_t)n_nodes * hidden);
    int rc;
    if (d_x) rc = tlc_gemm_f32(n_nodes, hidden, f_in, d_x, d_w1, nullptr, 0, t1, stream);
    else rc = tlc_spgemm_csr_dense_f32(n_nodes, f_in, hidden, d_xs_rowptr, d_xs_col, d_xs_val, d_w1, nullptr, 0, t1, stream);
    if (rc != TLC_OK) return rc;
    if (out_dim == 16 && hidden % 4 == 0 && hidden <= 128 && ((reinterpret_cast<uintptr_t>(t1) | reinterpret_cast<uintptr_t>(d_b1)) & 15) == 0) {
        // conv2's projection in the epilogue of conv1's aggregate: three launches, the [n, hidden] activations stay in registers
        hipLaunchKernelGGL(spmm_w2_kernel, dim3((unsigned)((n_nodes + 7) / 8)), dim3(256), 0, (hipStream_t)stream, n_nodes, d_rowptr, d_col, d_val,
                           t1, hidden, d_b1, d_w2, t3);
        TLC_HIP_CHECK(hipGetLastError());
    } else {
        if ((rc = tlc_spmm_csr_f32(n_nodes, d_rowptr, d_col, d_val, t1, hidden, d_b1, 1, t2, stream)) != TLC_OK) return rc;
        if ((rc = tlc_gemm_f32(n_nodes, out_dim, hidden, t2, d_w2, nullptr, 0, t3, stream)) != TLC_OK) return rc;
    }
    return tlc_spmm_csr_f32(n_nodes, d_rowptr, d_col, d_val, t3, out_dim, d_b2, flags, d_emb, stream);
}

extern "C" int tlc_gcn2_encode_f32(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, const float* d_val,
                                   const float* d_x, int32_t f_in, const float* d_w1, const float* d_b1, int32_t hidden,
                                   const float* d_w2, const float* d_b2, int32_t out_dim, int flags, float* d_ws, float* d_emb,
                                   void* stream) {
    TLC_REQUIRE(n_nodes >= 0 && f_in > 0 && hidden > 0 && out_dim > 0, "bad sizes");
    if (n_nodes == 0) return TLC_OK;
    TLC_REQUIRE(d_rowptr && d_col && d_val && d_x && d_w1 && d_w2 && d_ws && d_emb, "null pointer");
    return gcn2_encode_impl(n_nodes, d_rowptr, d_col, d_val, d_x, nullptr, nullptr, nullptr, f_in, d_w1, d_b1, hidden, d_w2, d_b2, out_dim,
                            flags, d_ws, d_emb, stream);
}

// The same with the features given as CSR (bag-of-words / TF-IDF rows: tlc_spgemm_csr_dense_f32 in place of the dense projection).
extern "C" int tlc_gcn2_encode_csr_f32(int32_t n_nodes, const int32_t* d_rowptr, const int32_t* d_col, const float* d_val,
                                       const int32_t* d_xs_rowptr, const int32_t* d_xs_col, const float* d_xs_val, int32_t f_in,
                                       const float* d_w1, const float* d_b1, int32_t hidden, const float* d_w2, const float* d_b2,
                                       int32_t out_dim, int flags, float* d_ws, float* d_emb, void* stream) {
    TLC_REQUIRE(n_nodes >= 0 && f_in > 0 && hidden > 0 && out_dim > 0, "bad sizes");
    if (n_nodes == 0) return TLC_OK;
    TLC_REQUIRE(d_rowptr && d_col && d_val && d_xs_rowptr && d_xs_col && d_xs_val && d_w1 && d_w2 && d_ws && d_emb, "null pointer");
    return gcn2_encode_impl(n_nodes, d_rowptr, d_col, d_val, nullptr, d_xs_rowptr, d_xs_col, d_xs_val, f_in, d_w1, d_b1, hidden, d_w2, d_b2,
                            out_dim, flags, d_ws, d_emb, stream);
}

template <typename PT>
static int lp_decode_launch(int64_t n_pairs, const int32_t* d_pairs, const float* d_emb, int32_t emb_dim, const PT* d_pi,
                            int32_t pi_dim, const float* d_W1, const float* d_b1, const float* d_W2, const float* d_b2,
                            float* d_prob, void* stream) {
    TLC_REQUIRE(n_pairs >= 0 && emb_dim > 0 && pi_dim > 0, "bad sizes");
    if (n_pairs == 0) return TLC_OK;
    TLC_REQUIRE(d_pairs && d_emb && d_pi && d_W1 && d_b1 && d_W2 && d_b2 && d_prob, "null pointer");
    hipStream_t s = (hipStream_t)stream;
    const bool aligned = ((reinterpret_cast<uintptr_t>(d_emb) & 15) | (reinterpret_cast<uintptr_t>(d_pairs) & 7)) == 0;
    if (emb_dim == 16 && pi_dim == 25 && aligned) {
        const long long tiles = (n_pairs + 63) / 64;
        long long blocks = (tiles + 3) / 4;                       // one 64-pair tile per wavefront up to eight workgroups per CU
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(lp_decode_mfma_kernel<PT>, dim3((unsigned)blocks), dim3(256), 0, s, (long long)n_pairs, d_pairs, d_emb, d_pi,
                           d_W1, d_b1, d_W2, d_b2, d_prob);
    } else {
        hipLaunchKernelGGL(lp_decode_generic_kernel<PT>, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, s, (long long)n_pairs, d_pairs,
                           d_emb, emb_dim, d_pi, pi_dim, d_W1, d_b1, d_W2, d_b2, d_prob);
    }
    TLC_HIP_CHECK(hipGetLastError());
    return TLC_OK;
}

extern "C" int tlc_lp_decode_fused(int64_t n_pairs, const int32_t* d_pairs, const float* d_emb, int32_t emb_dim, const double* d_pi,
                                   int32_t pi_dim, const float* d_W1, const float* d_b1, const float* d_W2, const float* d_b2,
                                   float* d_prob, void* stream) {
    return lp_decode_launch<double>(n_pairs, d_pairs, d_emb, emb_dim, d_pi, pi_dim, d_W1, d_b1, d_W2, d_b2, d_prob, stream);
}

extern "C" int tlc_lp_decode_fused_f32(int64_t n_pairs, const int32_t* d_pairs, const float* d_emb, int32_t emb_dim, const float* d_pi,
                                       int32_t pi_dim, const float* d_W1, const float* d_b1, const float* d_W2, const float* d_b2,
                                       float* d_prob, void* stream) {
    return lp_decode_launch<float>(n_pairs, d_pairs, d_emb, emb_dim, d_pi, pi_dim, d_W1, d_b1, d_W2, d_b2, d_prob, stream);
}

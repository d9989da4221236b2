// Probe: what v_mfma_f32_16x16x4_f32 delivers with this GEMM's launch shape and instruction mix, nothing else in the way.
//   mode 0: bare MFMA loop, NB independent accumulators, operands in registers
//   mode 1: + nine ds_read_b128 per 36 MFMAs (operands from LDS, as the GEMM's inner block)
// hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_f32_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NB, int MODE>
__global__ __launch_bounds__(512) void probe(float* out, int iters, float seed) {
    __shared__ f32x4 lds[2048];
    const int lane = threadIdx.x & 63;
    for (int k = threadIdx.x; k < 2048; k += blockDim.x) lds[k] = (f32x4){seed, seed + 1, seed + 2, seed + 3};
    __syncthreads();
    f32x4 acc[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 a = (f32x4){seed, seed * 2, seed * 3, seed * 4};
    f32x4 b[NB];
#pragma unroll
    for (int t = 0; t < NB; ++t) b[t] = (f32x4){seed + t, seed - t, seed * t, seed};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1) {
#pragma unroll
            for (int t = 0; t < NB; ++t) b[t] = lds[(it * 7 + t * 64 + lane) & 2047];
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int t = 0; t < NB; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], b[t][s], acc[t], 0, 0, 0);
    }
    float r = 0.f;
#pragma unroll
    for (int t = 0; t < NB; ++t) r += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int NB, int MODE>
static void run(const char* name, int grid, int threads) {
    float* out; hipMalloc(&out, (size_t)grid * threads * 4);
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<NB, MODE><<<grid, threads>>>(out, 100, 1.0f);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        probe<NB, MODE><<<grid, threads>>>(out, iters, 1.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double mfma = (double)grid * (threads / 64) * iters * 4 * NB;
        printf("%-34s grid %4d x %4d thr: %.3f ms  %.1f TFLOP/s  (%.1f cycles per MFMA and SIMD at 2.4 GHz)\n", name, grid, threads, ms,
               mfma * 2048 / ms / 1e9, ms * 1e-3 * 2.4e9 / (mfma / (256.0 * 4)));
    }
    hipFree(out);
}
int main() {
    run<9, 0>("bare, 9 acc, 2 waves/SIMD", 247, 512);
    run<9, 0>("bare, 9 acc, 2 waves/SIMD, 256 WG", 256, 512);
    run<9, 0>("bare, 9 acc, 1 wave/SIMD", 256, 256);
    run<9, 0>("bare, 9 acc, 4 waves/SIMD", 256, 1024);
    run<9, 1>("LDS operands, 9 acc, 2 waves/SIMD", 247, 512);
    run<4, 0>("bare, 4 acc, 2 waves/SIMD", 256, 512);
    return 0;
}

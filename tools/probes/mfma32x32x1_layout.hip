// Prints the register/lane -> (block, row, column) layout of v_mfma_f32_32x32x1_2b_f32's result on this GPU
// (round 5: lp_decode_mfma_kernel relies on it).  hipcc --offload-arch=gfx950 -o mfma_layout mfma32x32x1_layout.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x32 __attribute__((ext_vector_type(32)));
__global__ void probe(float* out) {
    const int lane = threadIdx.x;
    f32x32 acc;
    for (int v = 0; v < 32; ++v) acc[v] = 0.f;
    f32x32 r1 = __builtin_amdgcn_mfma_f32_32x32x1f32((float)lane, 1.0f, acc, 0, 0, 0);        // D = 32 block + i
    f32x32 r2 = __builtin_amdgcn_mfma_f32_32x32x1f32(1.0f, (float)lane, acc, 0, 0, 0);        // D = 32 block + j
    for (int v = 0; v < 32; ++v) { out[v * 64 + lane] = r1[v]; out[2048 + v * 64 + lane] = r2[v]; }
}
int main() {
    float* d; float h[4096];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int v = 0; v < 32; ++v) {
        printf("reg %2d: A-lane (32 block + row) at lanes 0,1,31,32,33,63: %g %g %g %g %g %g | B-lane (32 block + col): %g %g %g %g %g %g\n", v,
               h[v * 64], h[v * 64 + 1], h[v * 64 + 31], h[v * 64 + 32], h[v * 64 + 33], h[v * 64 + 63],
               h[2048 + v * 64], h[2048 + v * 64 + 1], h[2048 + v * 64 + 31], h[2048 + v * 64 + 32], h[2048 + v * 64 + 33], h[2048 + v * 64 + 63]);
    }
    return 0;
}

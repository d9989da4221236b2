// Probe: does hipExtStreamCreateWithCUMask partition the CUs of an MI355X, and how do mask bits map to (XCC, SE, CU)?
// Build + run on the GPU box: hipcc --offload-arch=gfx950 -O2 tools/probes/cumask_probe.hip -o /tmp/cumask_probe && /tmp/cumask_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <set>
#define CK(e) do { hipError_t _e = (e); if (_e != hipSuccess) { printf("%s -> %s\n", #e, hipGetErrorString(_e)); exit(1); } } while (0)

__global__ void where(unsigned* out, int spin) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
__global__ void big_lds(unsigned* out, int spin) {
    extern __shared__ unsigned sm[];
    sm[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw + sm[1] - 1; out[2 * blockIdx.x + 1] = xcc; }
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) __builtin_amdgcn_s_sleep(8);
}
static void report(const char* name, const std::vector<unsigned>& h, int n) {
    std::set<unsigned> cus;
    int per_xcc[16] = {0};
    for (int b = 0; b < n; ++b) {
        const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
        const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
        cus.insert((xcc << 16) | (se << 8) | (sh << 4) | cu);
    }
    for (unsigned c : cus) per_xcc[c >> 16]++;
    printf("%s: %d blocks on %zu distinct (xcc,se,sh,cu); per xcc:", name, n, cus.size());
    for (int x = 0; x < 8; ++x) printf(" %d", per_xcc[x]);
    printf("\n");
}
int main() {
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    printf("CUs %d\n", prop.multiProcessorCount);
    const int NB = 2048;
    unsigned* d; CK(hipMalloc(&d, NB * 2 * 4));
    std::vector<unsigned> h(NB * 2);
    // 1. unmasked
    hipLaunchKernelGGL(where, dim3(NB), dim3(64), 0, 0, d, 2000);
    CK(hipDeviceSynchronize()); CK(hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost));
    report("unmasked", h, NB);
    // 2. masks
    for (int variant = 0; variant < 4; ++variant) {
        unsigned mask[8] = {0};
        const char* nm = "";
        if (variant == 0) { for (int i = 0; i < 2; ++i) mask[i] = 0xffffffffu; nm = "bits 0..63"; }
        if (variant == 1) { for (int i = 6; i < 8; ++i) mask[i] = 0xffffffffu; nm = "bits 192..255"; }
        if (variant == 2) { for (int i = 0; i < 8; ++i) mask[i] = 0x000000ffu; nm = "low 8 of every 32"; }
        if (variant == 3) { for (int i = 0; i < 8; ++i) mask[i] = 0x11111111u; nm = "every 4th bit"; }
        hipStream_t s;
        hipError_t e = hipExtStreamCreateWithCUMask(&s, 8, mask);
        if (e != hipSuccess) { printf("hipExtStreamCreateWithCUMask(%s) -> %s\n", nm, hipGetErrorString(e)); continue; }
        hipLaunchKernelGGL(where, dim3(NB), dim3(64), 0, s, d, 2000);
        CK(hipStreamSynchronize(s)); CK(hipMemcpy(h.data(), d, NB * 8, hipMemcpyDeviceToHost));
        report(nm, h, NB);
        CK(hipStreamDestroy(s));
    }
    // 3. partition: a flood of small blocks on the complement while 64 whole-CU blocks (156 KB LDS) go to the reserved CUs
    {
        unsigned m_res[8] = {0}, m_rest[8];
        for (int i = 0; i < 8; ++i) { m_res[i] = 0x000000ffu; m_rest[i] = ~m_res[i]; }
        hipStream_t s_res, s_rest;
        CK(hipExtStreamCreateWithCUMask(&s_res, 8, m_res));
        CK(hipExtStreamCreateWithCUMask(&s_rest, 8, m_rest));
        CK(hipFuncSetAttribute((const void*)big_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024));
        unsigned* d2; CK(hipMalloc(&d2, 64 * 2 * 4));
        hipEvent_t e0, e1, e2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&e2));
        // flood first (20000 blocks x 100 us), then the whole-CU blocks: how long until they have run (10 us each)?
        CK(hipEventRecord(e0, s_res));
        hipLaunchKernelGGL(where, dim3(20000), dim3(64), 0, s_rest, d, 10000);
        hipLaunchKernelGGL(big_lds, dim3(64), dim3(512), 156 * 1024, s_res, d2, 1000);
        CK(hipEventRecord(e1, s_res));
        CK(hipEventRecord(e2, s_rest));
        CK(hipDeviceSynchronize());
        float t_res = 0, t_rest = 0;
        CK(hipEventElapsedTime(&t_res, e0, e1));
        CK(hipEventElapsedTime(&t_rest, e0, e2));
        std::vector<unsigned> h2(128);
        CK(hipMemcpy(h2.data(), d2, 128 * 4, hipMemcpyDeviceToHost));
        report("whole-CU blocks on the reserved mask", h2, 64);
        printf("reserved stream done after %.3f ms, flood stream after %.3f ms\n", t_res, t_rest);
        // the same without masks
        hipStream_t a, b; CK(hipStreamCreate(&a)); CK(hipStreamCreate(&b));
        CK(hipEventRecord(e0, a));
        hipLaunchKernelGGL(where, dim3(20000), dim3(64), 0, b, d, 10000);
        hipLaunchKernelGGL(big_lds, dim3(64), dim3(512), 156 * 1024, a, d2, 1000);
        CK(hipEventRecord(e1, a));
        CK(hipEventRecord(e2, b));
        CK(hipDeviceSynchronize());
        CK(hipEventElapsedTime(&t_res, e0, e1));
        CK(hipEventElapsedTime(&t_rest, e0, e2));
        printf("no masks: whole-CU stream done after %.3f ms, flood stream after %.3f ms\n", t_res, t_rest);
    }
    return 0;
}

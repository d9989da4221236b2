// What bounds a wavefront-per-item kernel with tlc_extract_kernel<64>'s own instruction mix (round 4 counters: ~46 % vector ALU,
// ~35 % scalar ALU, ~11 % branches, ~8 % memory, dependent chains, in order)?  One wavefront per workgroup; 1, 2, 4 (and 8)
// wavefronts per SIMD run the same loop; the body below is 100 instructions in that mix:
//   46 VALU in one dependent chain, 35 SALU in one dependent chain, 11 taken uniform branches, 8 memory instructions (4 dependent
//   LDS reads, 2 scalar loads, 2 vector loads of a cache-resident line, each waited for where its value is used).
// If the instructions per cycle and CU stop growing between 2 and 4 wavefronts per SIMD near ~0.7, the per-SIMD issue port is the
// bound and only the instruction count matters; if they keep doubling, latency is (review of round 4, item 1a).
//   hipcc --offload-arch=gfx950 -O2 tools/probes/mix_probe.hip -o tools/probes/bin/mix_probe && tools/probes/bin/mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define V4 "v_mad_u32_u24 %0, %0, %0, %1\n v_xor_b32 %0, %0, %1\n v_add_u32 %0, %0, %1\n v_lshrrev_b32 %0, 1, %0\n"
#define S3 "s_mul_i32 %2, %2, 3\n s_add_u32 %2, %2, 1\n s_lshr_b32 %2, %2, 1\n"
#define BR(n) "s_cmp_lg_u32 %2, -1\n s_cbranch_scc1 " #n "f\n s_nop 0\n " #n ":\n"        /* (the compare is one of the SALU instructions) */

// MODE 0: the full mix; 1: VALU + SALU only (no branches, no memory); 2: without the memory instructions; 3: without the branches
template <int MODE>
__global__ __launch_bounds__(64) void k_mix(int iters, const int* __restrict__ g, int* out) {
    __shared__ int lds[256];
    int v = threadIdx.x, w = 3, s = blockIdx.x | 1, a = threadIdx.x & 63;
    lds[threadIdx.x] = (threadIdx.x * 7) & 63; lds[threadIdx.x + 64] = threadIdx.x; lds[threadIdx.x + 128] = 1; lds[threadIdx.x + 192] = 2;
    __syncthreads();
    const int* gp = g + (threadIdx.x & 15);
    for (int i = 0; i < iters; ++i) {
        // 11 groups of (4 VALU, 3 SALU incl. the branch's compare, 1 branch) = 44 V + 33 S + 11 B; + 2 V + 2 S below = 46 / 35 / 11
        asm volatile(V4 S3 : "+v"(v), "+v"(w), "+s"(s) : : "scc");
        if (MODE == 0 || MODE == 2) asm volatile(BR(1) : "+v"(v), "+v"(w), "+s"(s) : : "scc"); else asm volatile("s_cmp_lg_u32 %2, -1\n" : "+v"(v), "+v"(w), "+s"(s) : : "scc");
#define GRP(n) asm volatile(V4 "s_mul_i32 %2, %2, 3\n s_add_u32 %2, %2, 1\n" : "+v"(v), "+v"(w), "+s"(s) : : "scc"); \
               if (MODE == 0 || MODE == 2) asm volatile(BR(n) : "+v"(v), "+v"(w), "+s"(s) : : "scc"); else asm volatile("s_cmp_lg_u32 %2, -1\n" : "+v"(v), "+v"(w), "+s"(s) : : "scc");
        GRP(2) GRP(3) GRP(4) GRP(5) GRP(6) GRP(7) GRP(8) GRP(9) GRP(10) GRP(11)
        asm volatile("v_add_u32 %0, %0, %1\n v_and_b32 %1, 63, %0\n s_add_u32 %2, %2, 2\n s_and_b32 %2, %2, 0xffff\n" : "+v"(v), "+v"(w), "+s"(s) : : "scc");
        if (MODE == 0 || MODE == 3) {
            // 4 dependent LDS reads (a pointer chase), 2 scalar loads, 2 vector loads; each consumed
            a = lds[a & 63]; a = lds[(a & 63) + 64]; a = lds[(a & 63)]; a = lds[(a & 63) + 64];
            int s0, s1;
            asm volatile("s_load_dword %0, %2, 0x0\n s_load_dword %1, %2, 0x4\n s_waitcnt lgkmcnt(0)\n" : "=s"(s0), "=s"(s1) : "s"(g) : "memory");
            s ^= (s0 ^ s1) & 1;
            const int x0 = gp[0], x1 = gp[16];
            v += (x0 ^ x1 ^ a) & 1;
        }
    }
    if (v == 0x7fffffff) out[0] = v + w + s + a;
}

template <int MODE>
static int run(const char* name, int per_iter, const int* d_g, int* d_out) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, iters = 4000;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%s (%d instructions per iteration)\n", name, per_iter);
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int grid = cus * 4 * wps;
        hipLaunchKernelGGL(k_mix<MODE>, dim3(grid), dim3(64), 0, 0, 10, d_g, d_out);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_mix<MODE>, dim3(grid), dim3(64), 0, 0, iters, d_g, d_out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double clk = prop.clockRate * 1e3;                         // Hz (maximum shader clock)
        const double insts = (double)grid * iters * per_iter;
        printf("  %d wavefront(s) per SIMD: %.3f ms  -> %.3f wave-instructions per cycle and CU (at %.2f GHz), %.1f cycles per iteration and wavefront\n",
               wps, ms, insts / (ms * 1e-3 * clk * cus), clk / 1e9, ms * 1e-3 * clk / iters);
    }
    return 0;
}

int main() {
    int* d_g; int* d_out;
    CHECK(hipMalloc(&d_g, 4096)); CHECK(hipMalloc(&d_out, 64));
    CHECK(hipMemset(d_g, 0, 4096));
    if (run<0>("full mix: 46 VALU + 35 SALU + 11 taken branches + 8 memory (4 LDS chase, 2 SMEM, 2 VMEM)", 100 + 12, d_g, d_out)) return 1;   // (+ address / wait instructions of the memory part)
    if (run<2>("without the memory instructions", 92, d_g, d_out)) return 1;
    if (run<3>("without the branches", 89 + 12, d_g, d_out)) return 1;
    if (run<1>("VALU + SALU only", 81, d_g, d_out)) return 1;
    return 0;
}

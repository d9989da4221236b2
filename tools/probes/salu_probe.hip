// How many scalar instructions does a CU take per cycle, and what does a taken branch / an exec-mask if cost?  One wavefront per
// workgroup, W workgroups per CU; every wavefront runs the same loop of scalar (or vector) instructions.  Development aid:
//   hipcc --offload-arch=gfx950 -O2 tools/probes/salu_probe.hip -o /tmp/salu_probe && /tmp/salu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_salu(int iters, int* out) {
    int a = blockIdx.x, b = 1, c = 2, d = 3;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n"
            "s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n"
            "s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n"
            "s_add_u32 %0, %0, %1\n s_add_u32 %1, %1, %2\n s_add_u32 %2, %2, %3\n s_add_u32 %3, %3, %0\n"
            : "+s"(a), "+s"(b), "+s"(c), "+s"(d) : : "scc");
    }
    if (a == 0x7fffffff) out[0] = a + b + c + d;
}
__global__ void k_valu(int iters, int* out) {
    int a = threadIdx.x, b = 1, c = 2, d = 3;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
            "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
            "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
            "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n"
            : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    }
    if (a == 0x7fffffff) out[0] = a + b + c + d;
}
// half scalar, half vector, interleaved: do they share an issue port?
__global__ void k_mix(int iters, int* out) {
    int a = blockIdx.x, b = 1, c = threadIdx.x, d = 3;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "s_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n s_add_u32 %1, %1, %0\n v_add_u32 %3, %3, %2\n"
            "s_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n s_add_u32 %1, %1, %0\n v_add_u32 %3, %3, %2\n"
            "s_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n s_add_u32 %1, %1, %0\n v_add_u32 %3, %3, %2\n"
            "s_add_u32 %0, %0, %1\n v_add_u32 %2, %2, %3\n s_add_u32 %1, %1, %0\n v_add_u32 %3, %3, %2\n"
            : "+s"(a), "+s"(b), "+v"(c), "+v"(d) : : "scc");
    }
    if (c == 0x7fffffff) out[0] = a + b + c + d;
}
// sixteen taken uniform branches per iteration (compare + s_cbranch_scc1 over one instruction)
__global__ void k_ubr(int iters, int* out) {
    int a = blockIdx.x;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n 1:\n" "s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 2f\n s_add_u32 %0, %0, 1\n 2:\n"
            "s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 3f\n s_add_u32 %0, %0, 1\n 3:\n" "s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 4f\n s_add_u32 %0, %0, 1\n 4:\n"
            "s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 5f\n s_add_u32 %0, %0, 1\n 5:\n" "s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 6f\n s_add_u32 %0, %0, 1\n 6:\n"
            "s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 7f\n s_add_u32 %0, %0, 1\n 7:\n" "s_cmp_eq_u32 %0, %0\n s_cbranch_scc1 8f\n s_add_u32 %0, %0, 1\n 8:\n"
            : "+s"(a) : : "scc");
    }
    if (a == 0x7fffffff) out[0] = a;
}
// eight NOT taken uniform branches per iteration
__global__ void k_unt(int iters, int* out) {
    int a = blockIdx.x;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 1f\n s_add_u32 %0, %0, 1\n 1:\n" "s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 2f\n s_add_u32 %0, %0, 1\n 2:\n"
            "s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 3f\n s_add_u32 %0, %0, 1\n 3:\n" "s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 4f\n s_add_u32 %0, %0, 1\n 4:\n"
            "s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 5f\n s_add_u32 %0, %0, 1\n 5:\n" "s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 6f\n s_add_u32 %0, %0, 1\n 6:\n"
            "s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 7f\n s_add_u32 %0, %0, 1\n 7:\n" "s_cmp_lg_u32 %0, %0\n s_cbranch_scc1 8f\n s_add_u32 %0, %0, 1\n 8:\n"
            : "+s"(a) : : "scc");
    }
    if (a == 0x7fffffff) out[0] = a;
}
// sixteen divergent ifs per iteration (exec-mask save / branch / restore around one vector add): the shape of `if (hit) store`
__global__ void k_if(int iters, int* out) {
    int v = threadIdx.x, acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if ((v >> (q & 3)) & 1) { asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(v)); }
            asm volatile("" : "+v"(v));
        }
    }
    if (acc == 0x7fffffff) out[0] = acc;
}
// the same work without control flow: a select per add
__global__ void k_sel(int iters, int* out) {
    int v = threadIdx.x, acc = 0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int t = ((v >> (q & 3)) & 1) ? v : 0;
            asm volatile("v_add_u32 %0, %0, %1" : "+v"(acc) : "v"(t));
            asm volatile("" : "+v"(v));
        }
    }
    if (acc == 0x7fffffff) out[0] = acc;
}

int main() {
    int* out;
    CHECK(hipMalloc(&out, 64));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const double ghz = prop.clockRate * 1e-6;
    printf("CUs %d, clock %.2f GHz\n", cus, ghz);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int iters = 20000;
    struct { const char* name; void (*k)(int, int*); double per_iter; } ks[] = {
        {"scalar adds", k_salu, 16}, {"vector adds", k_valu, 16}, {"scalar + vector interleaved", k_mix, 16},
        {"8 taken uniform branches", k_ubr, 8}, {"8 untaken uniform branches", k_unt, 8}, {"16 divergent ifs", k_if, 16}, {"16 selects", k_sel, 16}};
    for (auto& K : ks) {
        for (int w : {1, 4, 8, 16, 32}) {          // wavefronts per CU
            hipLaunchKernelGGL(K.k, dim3(cus * w), dim3(64), 0, 0, 100, out);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(K.k, dim3(cus * w), dim3(64), 0, 0, iters, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double cyc = ms * 1e-3 * ghz * 1e9;
            printf("%-30s %2d wavefronts/CU: %8.3f ms  %6.2f cycles per iteration-item and wavefront  %6.2f items per cycle and CU\n", K.name, w, ms,
                   cyc / (iters * K.per_iter), w * iters * K.per_iter / cyc);
        }
    }
    return 0;
}

// Does an MFMA occupy the vector issue port of its SIMD, and for how long?  16x16x16 against 16x16x32 bf16, alone and interleaved with
// plain VALU work, at 1 / 2 / 3 waves per SIMD (DESIGN.md §4.4 rule 35).  hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_port.hip -o /tmp/mfma_port
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
#define REP8(x) x x x x x x x x
template <int OP>
__global__ void k(float* sink, int iters) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    s16x4 a4 = {1, 2, 3, 4}; s16x8 a8 = {1, 2, 3, 4, 5, 6, 7, 8};
    float f0 = threadIdx.x * 1e-3f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == 0) { REP8(asm volatile("v_mfma_f32_16x16x16_bf16 %0, %4, %4, %0\n v_mfma_f32_16x16x16_bf16 %1, %4, %4, %1\n v_mfma_f32_16x16x16_bf16 %2, %4, %4, %2\n v_mfma_f32_16x16x16_bf16 %3, %4, %4, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a4));) }
        if constexpr (OP == 1) { REP8(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %4, %4, %0\n v_mfma_f32_16x16x32_bf16 %1, %4, %4, %1\n v_mfma_f32_16x16x32_bf16 %2, %4, %4, %2\n v_mfma_f32_16x16x32_bf16 %3, %4, %4, %3" : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a8));) }
        // one MFMA : four independent fma
        if constexpr (OP == 2) { REP8(asm volatile("v_mfma_f32_16x16x16_bf16 %0, %8, %8, %0\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %4\n"
                                               "v_mfma_f32_16x16x16_bf16 %1, %8, %8, %1\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %4"
                                               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a4));) }
        if constexpr (OP == 3) { REP8(asm volatile("v_mfma_f32_16x16x32_bf16 %0, %8, %8, %0\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %4\n"
                                               "v_mfma_f32_16x16x32_bf16 %1, %8, %8, %1\n v_fma_f32 %4, %4, %4, %5\n v_fma_f32 %5, %5, %5, %6\n v_fma_f32 %6, %6, %6, %7\n v_fma_f32 %7, %7, %7, %4"
                                               : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a8));) }
        // the fma stream alone (8 per group)
        if constexpr (OP == 4) { REP8(asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0\n v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));) }
    }
    sink[threadIdx.x + blockIdx.x * blockDim.x] = c0[0] + c1[1] + c2[2] + c3[3] + f0 + f1 + f2 + f3;
}
int main() {
    float* s; hipMalloc(&s, 4096 * 1024 * 4);
    const char* names[] = {"4 x mfma 16x16x16 bf16", "4 x mfma 16x16x32 bf16", "2 x (mfma x16 + 4 fma)", "2 x (mfma x32 + 4 fma)", "8 x fma"};
    void (*ks[])(float*, int) = {k<0>, k<1>, k<2>, k<3>, k<4>};
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 2048;
    const double ghz = prop.clockRate * 1e-6;
    printf("CUs %d, nameplate %.2f GHz: cycles per group at that clock\n", cus, ghz);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfg[3][2] = {{1, 256}, {2, 512}, {3, 768}};
    for (int op = 0; op < 5; ++op) {
        printf("%-26s", names[op]);
        for (int c = 0; c < 3; ++c) {
            hipLaunchKernelGGL(ks[op], dim3(cus), dim3(cfg[c][1]), 0, 0, s, 16);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(ks[op], dim3(cus), dim3(cfg[c][1]), 0, 0, s, iters);
            hipEventRecord(e1, 0); hipEventSynchronize(e1);
            float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("  %dw/SIMD: %6.1f", cfg[c][0], ms * 1e-3 * ghz * 1e9 / (8.0 * iters * cfg[c][0]));
        }
        printf("\n");
    }
    return 0;
}

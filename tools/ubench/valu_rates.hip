// Issue rate of the VALU ops the BiMAU kernels are made of, per SIMD, against the number of resident waves per SIMD
// (host-timed with HIP events on a full-chip grid; 4 independent dependency chains per wave).
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(uint64_t* out, float* sink, int iters) {
    float a0 = threadIdx.x * 1e-3f + 0.5f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    float b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    uint32_t u0 = threadIdx.x + 1, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    const uint32_t c = 0x9E3779B1u;
    uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == 0) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(c));) }
        if constexpr (OP == 1) { REP16(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(c));) }
        if constexpr (OP == 2) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (OP == 3) { REP16(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (OP == 4) { REP16(asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (OP == 5) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %1\n v_pk_fma_f32 %1, %1, %1, %0\n v_pk_fma_f32 %0, %0, %1, %1\n v_pk_fma_f32 %1, %1, %0, %0" : "+v"(*(double*)&a0), "+v"(*(double*)&a2));) }
        if constexpr (OP == 6) { REP16(asm volatile("v_xor_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_xor_b32 %2, %2, %3\n v_xor_b32 %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if constexpr (OP == 7) { REP16(asm volatile("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) :: "vcc");) }
        if constexpr (OP == 8) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %4, %1\n v_mad_u32_u24 %1, %1, %4, %2\n v_mad_u32_u24 %2, %2, %4, %3\n v_mad_u32_u24 %3, %3, %4, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(c));) }
        if constexpr (OP == 9) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1\n v_cvt_pk_bf16_f32 %1, %1, %2\n v_cvt_pk_bf16_f32 %2, %2, %3\n v_cvt_pk_bf16_f32 %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (OP == 10) { REP16(asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (OP == 11) { REP16(asm volatile("v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %0\n v_pk_mul_f32 %0, %0, %1\n v_pk_add_f32 %1, %1, %0" : "+v"(*(double*)&a0), "+v"(*(double*)&a2));) }
        if constexpr (OP == 12) { REP16(asm volatile("v_max3_f32 %0, %0, %1, %2\n v_max3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_max3_f32 %3, %3, %0, %1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (OP == 13) { REP16(asm volatile("v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %0\n v_lshl_add_u64 %0, %0, 1, %1\n v_lshl_add_u64 %1, %1, 1, %0" : "+v"(*(uint64_t*)&b0), "+v"(*(uint64_t*)&b2));) }
    }
    uint64_t t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    sink[threadIdx.x + blockIdx.x * 64] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + (float)(u0 ^ u1 ^ u2 ^ u3);
}

int main() {
    uint64_t* d; float* s;
    hipMalloc(&d, 4096 * 8); hipMalloc(&s, 4096 * 1024 * 4);
    const char* names[] = {"v_mul_lo_u32", "v_mul_u32_u24", "v_exp_f32", "v_rcp_f32", "v_fma_f32", "v_pk_fma_f32", "v_xor_b32", "v_cndmask_b32",
                           "v_mad_u32_u24", "v_cvt_pk_bf16_f32", "v_log_f32", "v_pk_mul/add_f32", "v_max3_f32", "v_lshl_add_u64"};
    void (*ks[])(uint64_t*, float*, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>, k<13>};
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 2048;
    const double ghz = prop.clockRate * 1e-6;
    printf("CUs %d, clock %.2f GHz\n", cus, ghz);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    // waves per SIMD: 1 (256 threads x 1 block per CU), 2, 4 (1024 threads), 8 (2 blocks of 1024 per CU)
    const int cfg[4][3] = {{1, 256, 1}, {2, 512, 1}, {4, 1024, 1}, {8, 1024, 2}};
    for (int op = 0; op < 14; ++op) {
        printf("%-20s", names[op]);
        for (int c = 0; c < 4; ++c) {
            hipLaunchKernelGGL(ks[op], dim3(cus * cfg[c][2]), dim3(cfg[c][1]), 0, 0, d, s, 16);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(ks[op], dim3(cus * cfg[c][2]), dim3(cfg[c][1]), 0, 0, d, s, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
            const double instr_per_simd = 64.0 * iters * cfg[c][0];
            printf("  %dw/SIMD: %.2f cyc/instr", cfg[c][0], ms * 1e-3 * ghz * 1e9 / instr_per_simd);
        }
        printf("\n");
    }
    return 0;
}

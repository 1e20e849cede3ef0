// Issue cost of the instruction MIXES the BiMAU kernels are made of: does a transcendental overlap with plain VALU work of
// the same / the other resident wave, what do the 64-bit multiply and the SDWA compare cost.  Full-chip grid, HIP-event timed.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_mix.hip -o /tmp/valu_mix && /tmp/valu_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP16(x) x x x x x x x x x x x x x x x x
template <int OP>
__global__ void k(float* sink, int iters) {
    float a0 = threadIdx.x * 1e-3f + 0.5f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
    float b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    uint32_t u0 = threadIdx.x + 1, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    uint64_t w0 = u0, w1 = u1;
    const uint32_t c = 0x9E3779B1u;
    for (int it = 0; it < iters; ++it) {
        if constexpr (OP == 0) { REP16(asm volatile("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if constexpr (OP == 1) { REP16(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        // 1 transcendental : 1 plain (independent registers)
        if constexpr (OP == 2) { REP16(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %2, %2, %2, %3\n v_exp_f32 %1, %1\n v_fma_f32 %3, %3, %3, %2" : "+v"(a0), "+v"(a1), "+v"(b2), "+v"(b3));) }
        // 1 : 3
        if constexpr (OP == 3) { REP16(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %1" : "+v"(a0), "+v"(b1), "+v"(b2), "+v"(b3));) }
        // 1 : 7
        if constexpr (OP == 4) { REP16(asm volatile("v_exp_f32 %0, %0\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %1\n v_fma_f32 %1, %1, %1, %2" : "+v"(a0), "+v"(b1), "+v"(b2), "+v"(b3));) }
        // sigmoid . weight as the forward has it: exp, add, rcp, fma (4 independent chains -> 16 instructions per asm)
        if constexpr (OP == 5) { REP16(asm volatile(
            "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n"
            "v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3\n"
            "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
            "v_fma_f32 %4, %0, %5, %4\n v_fma_f32 %4, %1, %5, %4\n v_fma_f32 %4, %2, %5, %4\n v_fma_f32 %4, %3, %5, %4"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0) : "v"(b1));) }
        if constexpr (OP == 6) { REP16(asm volatile("v_mad_u64_u32 %0, vcc, %2, %4, %1\n v_mad_u64_u32 %1, vcc, %3, %4, %0\n v_mad_u64_u32 %0, vcc, %2, %4, %1\n v_mad_u64_u32 %1, vcc, %3, %4, %0" : "+v"(w0), "+v"(w1) : "v"(u0), "v"(u1), "s"(c) : "vcc");) }
        if constexpr (OP == 7) { REP16(asm volatile("v_cmp_ge_u32_sdwa vcc, %0, %4 src0_sel:WORD_1 src1_sel:DWORD\n v_cndmask_b32 %1, 0, %1, vcc\n v_cmp_ge_u32_sdwa vcc, %0, %4 src0_sel:WORD_0 src1_sel:DWORD\n v_cndmask_b32 %2, 0, %2, vcc" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "v"(c) : "vcc");) }
        if constexpr (OP == 8) { REP16(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(c));) }
        // packed f16 arithmetic (two elements per lane and instruction)
        if constexpr (OP == 9) { REP16(asm volatile("v_pk_fma_f16 %0, %0, %0, %1\n v_pk_fma_f16 %1, %1, %1, %2\n v_pk_fma_f16 %2, %2, %2, %3\n v_pk_fma_f16 %3, %3, %3, %0" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));) }
        if constexpr (OP == 10) { REP16(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3) : "s"(c));) }
        // 2 transcendentals : 2 plain, the transcendentals back to back
        if constexpr (OP == 11) { REP16(asm volatile("v_exp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %2" : "+v"(a0), "+v"(a1), "+v"(b2), "+v"(b3));) }
        if constexpr (OP == 12) { REP16(asm volatile("v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
    }
    sink[threadIdx.x + blockIdx.x * blockDim.x] = a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 + (float)(u0 ^ u1 ^ u2 ^ u3) + (float)(w0 ^ w1);
}

int main() {
    float* s;
    hipMalloc(&s, 4096 * 1024 * 4);
    const char* names[] = {"fma x4", "exp x4", "exp:fma 1:1", "exp:fma 1:3", "exp:fma 1:7", "sigmoid.w (16 instr)", "v_mad_u64_u32", "cmp_sdwa+cndmask x2", "v_mul_lo_u32",
                           "v_pk_fma_f16", "v_mul_hi_u32", "exp,rcp,fma,fma", "v_exp_f16"};
    const int per_asm[] = {4, 4, 4, 4, 8, 16, 4, 4, 4, 4, 4, 4, 4};
    void (*ks[])(float*, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>};
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, iters = 1024;
    const double ghz = prop.clockRate * 1e-6;
    printf("CUs %d, clock %.2f GHz (nameplate; cycles below are at that clock)\n", cus, ghz);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int cfg[3][3] = {{1, 256, 1}, {2, 512, 1}, {4, 1024, 1}};
    for (int op = 0; op < 13; ++op) {
        printf("%-24s", names[op]);
        for (int c = 0; c < 3; ++c) {
            hipLaunchKernelGGL(ks[op], dim3(cus * cfg[c][2]), dim3(cfg[c][1]), 0, 0, s, 16);
            hipEventRecord(e0, 0);
            hipLaunchKernelGGL(ks[op], dim3(cus * cfg[c][2]), dim3(cfg[c][1]), 0, 0, s, iters);
            hipEventRecord(e1, 0);
            hipEventSynchronize(e1);
            float ms = 0.f; hipEventElapsedTime(&ms, e0, e1);
            const double groups_per_simd = 16.0 * iters * cfg[c][0];
            printf("  %dw/SIMD: %6.2f cyc/group(%d)", cfg[c][0], ms * 1e-3 * ghz * 1e9 / groups_per_simd, per_asm[op]);
        }
        printf("\n");
    }
    return 0;
}

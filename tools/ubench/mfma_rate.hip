// Issue rate of v_mfma_f32_32x32x16_bf16 from ONE wave per SIMD (256-thread workgroups, one per CU) by register file of the
// operands: which of  D/C in AGPR | VGPR,  B in VGPR | AGPR  costs matrix-pipe cycles?  Reports shader-clock ticks per MFMA
// (s_memtime) and the wall-clock rate.     hipcc --offload-arch=gfx950 -O3 -o tools/ubench/mfma_rate tools/ubench/mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int v4i __attribute__((ext_vector_type(4)));
constexpr int NIT = 512;

template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const int* src, float* out, unsigned long long* ticks) {
    const int lane = threadIdx.x;
    v4i a = *reinterpret_cast<const v4i*>(src + lane * 4), b[8];
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) {
        b[i] = *reinterpret_cast<const v4i*>(src + ((lane * 4 + i * 1024) & 4095));
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    }
    if (MODE == 0) for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(acc[i]), "+v"(b[i]));
    if (MODE == 1) for (int i = 0; i < 8; ++i) asm volatile("" : "+a"(acc[i]), "+a"(b[i]));
    if (MODE == 2) for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(acc[i]), "+a"(b[i]));
    if (MODE == 3) for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(acc[i]), "+v"(b[i]));
    asm volatile("s_nop 15\n\ts_nop 15");
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b[i]));
            if (MODE == 1) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "a"(b[i]));
            if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "a"(b[i]));
            if (MODE == 3) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b[i]));
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0.f;
    for (int i = 0; i < 8; ++i) {
        if (MODE <= 1) asm volatile("" : "+a"(acc[i])); else asm volatile("" : "+v"(acc[i]));
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    out[blockIdx.x * 256 + lane] = s;
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}

// dependent accumulation: NACC independent chains, the same accumulator every NACC-th MFMA
template <int NACC>
__global__ __launch_bounds__(256, 1) void kdep(const int* src, float* out, unsigned long long* ticks) {
    const int lane = threadIdx.x;
    v4i a = *reinterpret_cast<const v4i*>(src + lane * 4), b = *reinterpret_cast<const v4i*>(src + ((lane * 4 + 1024) & 4095));
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) { for (int r = 0; r < 16; ++r) acc[i][r] = 0.f; asm volatile("" : "+v"(acc[i])); }
    asm volatile("" : "+a"(b));
    asm volatile("s_nop 15\n\ts_nop 15");
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < NIT * 8 / NACC; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "a"(b));
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) { asm volatile("" : "+v"(acc[i])); for (int r = 0; r < 16; ++r) s += acc[i][r]; }
    out[blockIdx.x * 256 + lane] = s;
    if (lane == 0) ticks[blockIdx.x] = t1 - t0;
}
template <int NACC>
void rundep(const int* src, float* out, unsigned long long* ticks, int nblk) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(kdep<NACC>, dim3(nblk), dim3(256), 0, 0, src, out, ticks);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk);
    hipMemcpy(h.data(), ticks, nblk * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= nblk;
    printf("%d accumulation chains (D/C VGPR, B AGPR): %6.1f ticks / MFMA\n", NACC, avg / (NIT * 8.0));
}

template <int MODE>
void run(const char* name, const int* src, float* out, unsigned long long* ticks, int nblk) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, src, out, ticks);
    hipEventRecord(e0);
    const int reps = 20;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k<MODE>, dim3(nblk), dim3(256), 0, 0, src, out, ticks);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nblk);
    hipMemcpy(h.data(), ticks, nblk * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= nblk;
    const double per = avg / (NIT * 8.0);
    const double flop = 2.0 * 32 * 32 * 16 * NIT * 8 * 4 * (double)nblk;
    printf("%-34s %6.1f ticks / MFMA   kernel %7.1f us   %7.0f TF/s\n", name, per, ms / reps * 1e3, flop / (ms / reps * 1e-3) / 1e12);
}

int main(int argc, char** argv) {
    const int nblk = argc > 1 ? atoi(argv[1]) : 256;
    const bool zero = argc > 2 && atoi(argv[2]);
    int* src; float* out; unsigned long long* ticks;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, nblk * 256 * 4); hipMalloc(&ticks, nblk * 8);
    std::vector<int> h(4096);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; unsigned e0 = 0x3c00 + ((x >> 8) & 0x3ff), e1 = 0xbc00 + ((x >> 20) & 0x3ff); v = zero ? 0 : (int)(e0 | (e1 << 16)); }
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    printf("%d workgroups x 4 waves, %s operands\n", nblk, zero ? "zero" : "random bf16");
    run<0>("D/C AGPR, B VGPR", src, out, ticks, nblk);
    run<1>("D/C AGPR, B AGPR", src, out, ticks, nblk);
    run<2>("D/C VGPR, B AGPR", src, out, ticks, nblk);
    run<3>("D/C VGPR, B VGPR", src, out, ticks, nblk);
    rundep<1>(src, out, ticks, nblk); rundep<2>(src, out, ticks, nblk); rundep<3>(src, out, ticks, nblk); rundep<4>(src, out, ticks, nblk);
    return 0;
}

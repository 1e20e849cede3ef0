// What fits beside a v_mfma_f32_32x32x16_bf16 issued by ONE wave per SIMD?  Pure MFMA streams run at 32.0 cycles per MFMA
// (mfma_rate.hip); this adds, per MFMA: NF v_fma_f32, NE v_exp_f32, NA v_add_f32, NC v_cvt_pk (every other MFMA), and per MFMA
// PAIR one ds_read_b128 (LB = 1) or two ds_read_b64_tr_b16 (LT = 1) consumed two pairs later — the filler mix of the strip
// scoring kernels — with the accumulators in VGPRs (DV = 1) or AGPRs.  Prints shader-clock cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((ext_vector_type(4))) short s4;
constexpr int NIT = 256;

template <int NF, int NE, int NA, int NC, int LB, int LT, int DV, int ST = 0>
__global__ __launch_bounds__(256, 1) void k(const int* src, float* out, unsigned long long* ticks) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<int*>(lds)[i] = src[i & 4095];
    __syncthreads();
    v4i b[8], zf[3];
    f32x16 acc[8];
    float t[8], sum = 0.f;
    int pk = 0;
    for (int i = 0; i < 8; ++i) {
        b[i] = *reinterpret_cast<const v4i*>(src + ((lane * 4 + i * 1024) & 4095));
        t[i] = (float)(lane + i) * 1e-3f;
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        if (DV) asm volatile("" : "+v"(acc[i]), "+a"(b[i])); else asm volatile("" : "+a"(acc[i]), "+v"(b[i]));
    }
    for (int i = 0; i < 3; ++i) zf[i] = *reinterpret_cast<const v4i*>(lds + lane * 16 + i * 1024);
    asm volatile("s_nop 15\n\ts_nop 15");
    // the strip kernels' layouts: row z at z*512 + rot(z)*16 — row-fragment reads and transpose reads are conflict-free
    auto rot16 = [](int z) { return (((z & 3) << 2) | ((z >> 2) & 3)) * 16; };
    const int zr = lane & 31, hi = lane >> 5, G = lane >> 4, s_ = lane & 15, tz = 4 * hi + (s_ >> 2);
    const char* lp = lds + (LT ? tz * 512 + rot16(tz) + (16 * (G & 1) + 4 * (s_ & 3)) * 2 : zr * 512 + rot16(zr) + hi * 16);
    const int vrow = threadIdx.x >> 4, vcv = threadIdx.x & 15;
    char* sp = lds + vrow * 512 + rot16(vrow) + vcv * 16;      // staged-tile store: 16 lanes per row, 16 rows per store
    const uint4 sdata = make_uint4(lane, lane + 1, lane + 2, lane + 3);
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < NIT; ++it) {
#pragma unroll
        for (int pr = 0; pr < 4; ++pr) {     // 4 pairs = 8 MFMAs
            if (LB) zf[(pr + 2) % 3] = *reinterpret_cast<const v4i*>(lp + pr * 32 + (it & 1) * 16384);
            if (LT) {
                const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lp + pr * 64 + (it & 1) * 16384));
                const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lp + pr * 64 + 4128 + (it & 1) * 16384));
                const uint2 a_ = __builtin_bit_cast(uint2, v0), b_ = __builtin_bit_cast(uint2, v1);
                zf[(pr + 2) % 3] = v4i{(int)a_.x, (int)a_.y, (int)b_.x, (int)b_.y};
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int i = pr * 2 + h;
                if (DV) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(zf[pr % 3]), "a"(b[i]));
                else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(zf[pr % 3]), "v"(b[i]));
                if (NE) asm volatile("v_exp_f32 %0, %0" : "+v"(t[i]));
                if (NF) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t[(i + 1) & 7]) : "s"(1.0001f), "v"(t[(i + 5) & 7]));
                if (NF > 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(t[(i + 2) & 7]) : "s"(1.0001f), "v"(t[(i + 6) & 7]));
                if (NA) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum) : "v"(t[(i + 7) & 7]));
                if (NC && h) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(t[(i + 6) & 7]), "v"(t[(i + 7) & 7]));
                if (ST == 1 && h == 0 && (it & 1)) *reinterpret_cast<uint4*>(sp + pr * 8192) = sdata;                 // 4 stores per 16 MFMAs
                if (ST == 2 && h == 0 && (it & 1) && pr == (int)(threadIdx.x >> 6)) *reinterpret_cast<uint4*>(sp) = sdata;   // one wave per slot
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_nop 15\n\ts_nop 15");
    float s = sum + (float)pk;
    for (int i = 0; i < 8; ++i) {
        if (DV) asm volatile("" : "+v"(acc[i])); else asm volatile("" : "+a"(acc[i]));
        for (int r = 0; r < 16; ++r) s += acc[i][r];
        s += t[i];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int NF, int NE, int NA, int NC, int LB, int LT, int DV, int ST = 0>
void run(const int* src, float* out, unsigned long long* ticks, int nblk) {
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<NF, NE, NA, NC, LB, LT, DV, ST>), dim3(nblk), dim3(256), 0, 0, src, out, ticks);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nblk);
    hipMemcpy(h.data(), ticks, nblk * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= nblk;
    printf("fma %d exp %d add %d cvt %d | ds_read_b128 %d tr_b16 %d | ds_write_b128 %s | acc in %s : %6.1f cycles / MFMA\n", NF, NE, NA,
           NC, LB, LT, ST == 0 ? "none" : (ST == 1 ? "4 per 16 MFMAs, lock step" : "1 per 16 MFMAs per wave, staggered"), DV ? "VGPR" : "AGPR",
           avg / (NIT * 8.0));
}

int main(int argc, char** argv) {
    const int nblk = argc > 1 ? atoi(argv[1]) : 256;
    int* src; float* out; unsigned long long* ticks;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, nblk * 256 * 4); hipMalloc(&ticks, nblk * 8);
    std::vector<int> h(4096);
    unsigned x = 12345;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; unsigned e0 = 0x3c00 + ((x >> 8) & 0x3ff), e1 = 0xbc00 + ((x >> 20) & 0x3ff); v = (int)(e0 | (e1 << 16)); }
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    run<0, 0, 0, 0, 0, 0, 1>(src, out, ticks, nblk);
    run<1, 1, 1, 1, 0, 0, 1>(src, out, ticks, nblk);
    run<1, 1, 1, 1, 1, 0, 1>(src, out, ticks, nblk);
    run<1, 1, 0, 1, 1, 0, 1>(src, out, ticks, nblk);
    run<1, 1, 0, 0, 1, 0, 1>(src, out, ticks, nblk);
    run<1, 0, 0, 0, 1, 0, 1>(src, out, ticks, nblk);
    run<0, 1, 0, 0, 1, 0, 1>(src, out, ticks, nblk);
    run<2, 0, 1, 1, 1, 0, 1>(src, out, ticks, nblk);
    run<1, 1, 1, 1, 0, 1, 0>(src, out, ticks, nblk);
    run<1, 1, 0, 1, 0, 1, 0>(src, out, ticks, nblk);
    run<1, 1, 0, 0, 0, 1, 0>(src, out, ticks, nblk);
    return 0;
}

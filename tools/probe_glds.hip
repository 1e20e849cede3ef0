// Probe (gfx950): where does global_load_lds_dwordx4 / _dword land for an M0 base above 64 KB, and does the instruction offset add?
//   hipcc --offload-arch=gfx950 -O2 tools/probe_glds.hip -o tools/probe_glds && tools/probe_glds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int LDSB = 160 * 1024;
__global__ void k(const uint32_t* src, uint32_t* out, unsigned base16, unsigned base4) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < LDSB / 4; i += blockDim.x) reinterpret_cast<uint32_t*>(smem)[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem;
    if (threadIdx.x < 64) {
        const uint32_t* g16 = src + threadIdx.x * 4;          // lane l: 16 bytes l
        const uint32_t* g4 = src + 1024 + threadIdx.x;        // lane l: dword 1024 + l
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g16), "s"(lds0 + base16) : "memory");
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g4), "s"(lds0 + base4) : "memory");
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(g16), "s"(lds0 + 4096) : "memory");     // does `offset` move the SOURCE, the DESTINATION or both?
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LDSB / 4; i += blockDim.x) out[i] = reinterpret_cast<uint32_t*>(smem)[i];
    if (threadIdx.x == 0) out[LDSB / 4] = lds0;
}
int main() {
    std::vector<uint32_t> h(4096);
    for (int i = 0; i < 4096; ++i) h[i] = 0x10000000u + i;
    uint32_t *src, *out;
    hipMalloc(&src, 4096 * 4); hipMalloc(&out, LDSB + 4);
    hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDSB);
    const unsigned base16 = 100000 / 16 * 16, base4 = 150000 / 4 * 4;
    hipLaunchKernelGGL(k, dim3(1), dim3(256), LDSB, 0, src, out, base16, base4);
    std::vector<uint32_t> o(LDSB / 4 + 1);
    hipError_t e = hipMemcpy(o.data(), out, LDSB + 4, hipMemcpyDeviceToHost);
    printf("status %d, lds0 = %u, base16 = %u, base4 = %u\n", (int)e, o[LDSB / 4], base16, base4);
    int first = -1, last = -1, n = 0;
    for (int i = 0; i < LDSB / 4; ++i)
        if (o[i] != 0xdeadbeefu) {
            if (first < 0 || i != last + 1) printf("%srun starts at byte %d: value 0x%x (src dword %d)\n", first < 0 ? "" : "", i * 4, o[i], (int)(o[i] - 0x10000000u));
            if (first < 0) first = i;
            last = i; ++n;
        }
    printf("%d dwords written, last at byte %d\n", n, last * 4);
    return 0;
}

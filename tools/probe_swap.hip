// Probe of v_permlane32_swap / v_permlane16_swap lane semantics on gfx950 (same idea as probe_tr.hip).
// hipcc --offload-arch=gfx950 -O2 tools/probe_swap.hip -o tools/probe_swap && ./tools/probe_swap
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    const unsigned l = threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
    auto q = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
    auto r2 = __builtin_amdgcn_permlane32_swap(l, l, false, false);   // same value on both sides
    auto q2 = __builtin_amdgcn_permlane16_swap(l, l, false, false);
    out[l] = r[0]; out[64 + l] = r[1]; out[128 + l] = q[0]; out[192 + l] = q[1];
    out[256 + l] = r2[0]; out[320 + l] = r2[1]; out[384 + l] = q2[0]; out[448 + l] = q2[1];
}
int main() {
    unsigned* d; unsigned h[512];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const char* nm[8] = {"swap32 vdst'", "swap32 vsrc'", "swap16 vdst'", "swap16 vsrc'", "swap32(l,l) 1st", "swap32(l,l) 2nd", "swap16(l,l) 1st", "swap16(l,l) 2nd"};
    for (int a = 0; a < 8; ++a) {
        printf("%s:", nm[a]);
        for (int l = 0; l < 64; ++l) printf(" %u", h[a * 64 + l]);
        printf("\n");
    }
    return 0;
}

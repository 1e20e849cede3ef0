// Probe: lane semantics of ds_read_b64_tr_b16 on gfx950 (which element does lane l, slot j receive?).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(int mode, unsigned short* out) {
    __shared__ unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;  // value = element index
    __syncthreads();
    const int l = threadIdx.x;
    // LDS viewed as a row-major matrix with 64 columns (row stride 128 B): element (r, c) = r*64 + c
    int r, c;
    if (mode == 0) { r = l & 15; c = (l >> 4) * 4; }          // lane -> row (l&15), cols 4g..4g+3
    else if (mode == 1) { r = (l >> 4) * 4 + 0; c = (l & 15) * 4; }
    else { r = (l & 3) + 4 * (l >> 4); c = ((l >> 2) & 3) * 4; }   // lane -> row within 16: 4*(l>>4)+(l&3), col block (l>>2)&3
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + r * 64 + c));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    unsigned short h[256];
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, mode, d);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) printf(" (r%2d,c%2d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
            printf("\n");
        }
    }
    return 0;
}

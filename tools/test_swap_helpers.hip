// Device self-test of the permlane-swap exchange helpers (reduce_scatter16 / all_gather16 / group_sum4 / group_max4).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/test_swap_helpers.hip -o /tmp/t_swap && /tmp/t_swap
#include "../easydgl_amd/csrc/bimau_common.h"
#include <cstdio>
using namespace bimau;
__global__ void k(float* out) {
    const int lane = threadIdx.x;
    float z[16];
    for (int e = 0; e < 16; ++e) z[e] = (float)(lane * 16 + e);
    float o4[4];
    reduce_scatter16(z, o4, lane);
    for (int i = 0; i < 4; ++i) out[lane * 4 + i] = o4[i];
    float g[16];
    all_gather16(o4, g, lane);
    for (int e = 0; e < 16; ++e) out[256 + lane * 16 + e] = g[e];
    out[256 + 1024 + lane] = group_sum4((float)lane);
    out[256 + 1024 + 64 + lane] = group_max4((float)lane);
}
int main() {
    float* d; static float h[256 + 1024 + 128];
    hipMalloc(&d, sizeof(h));
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int lane = 0; lane < 64; ++lane) {
        const int g = lane >> 4, l15 = lane & 15;
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * g + i;
            float want = 0;
            for (int gg = 0; gg < 4; ++gg) want += (float)((gg * 16 + l15) * 16 + e);
            if (h[lane * 4 + i] != want) { if (bad < 5) printf("rs lane %d i %d got %f want %f\n", lane, i, h[lane*4+i], want); ++bad; }
        }
        for (int e = 0; e < 16; ++e) {
            float want = 0;
            for (int gg = 0; gg < 4; ++gg) want += (float)((gg * 16 + l15) * 16 + e);
            if (h[256 + lane * 16 + e] != want) { if (bad < 10) printf("ag lane %d e %d got %f want %f\n", lane, e, h[256+lane*16+e], want); ++bad; }
        }
        float ws = 0, wm = 0;
        for (int gg = 0; gg < 4; ++gg) { ws += gg * 16 + l15; wm = fmaxf(wm, gg * 16 + l15); }
        if (h[1280 + lane] != ws) { if (bad < 15) printf("sum lane %d got %f want %f\n", lane, h[1280+lane], ws); ++bad; }
        if (h[1344 + lane] != wm) { if (bad < 20) printf("max lane %d got %f want %f\n", lane, h[1344+lane], wm); ++bad; }
    }
    printf("bad = %d\n", bad);
    return 0;
}

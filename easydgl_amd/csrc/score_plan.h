// Device-side launch plan of the row-side scoring passes (shared by k_score.hip and k_score_strip.hip).
#pragma once
// The number of weighted rows is only known on the device (edgl_compact_rows), so the x-block / item-chunk split of
// a launch of G workgroups is derived there: nx x-blocks cover the valid rows, the G/nx chunks share the z range.
struct DevPlan { int nx, nchunk, zchunk; };
__host__ __device__ __forceinline__ DevPlan dev_plan(int x_eff, int xb, int G, int ztotal, int ZB) {
    DevPlan d;
    d.nx = (x_eff + xb - 1) / xb;
    if (d.nx < 1) d.nx = 1;
    int ntiles = (ztotal + ZB - 1) / ZB;
    if (ntiles < 1) ntiles = 1;
    int nchunk = G / d.nx;
    if (nchunk < 1) nchunk = 1;
    if (nchunk > ntiles) nchunk = ntiles;
    const int per = (ntiles + nchunk - 1) / nchunk;
    d.nchunk = (ntiles + per - 1) / per;
    d.zchunk = per * ZB;
    return d;
}

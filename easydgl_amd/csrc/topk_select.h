// Per-row top-K selection shared by k_score.hip (K6: mask_topk_kernel) and k_eval_topk.hip (the exact fallback of the fused
// evaluation scoring): Base.py:181 tf.nn.top_k — descending, ties to the lower index; -0.0 and +0.0 tie.
#pragma once
#include "edgl_common.h"

__device__ __forceinline__ uint32_t float_key(float f) {  // monotone map float -> uint32 (larger = larger); -0.0 and +0.0 tie
    uint32_t u = __float_as_uint(f);
    u = u == 0x80000000u ? 0u : u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ float key_float(uint32_t k) {  // inverse of float_key
    return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k);
}


// 4-pass radix select of the K-th largest key of x[0 .. n) (any address space: global logits row or an LDS image of it), then the
// ordered compaction and a bitonic sort of the <= 128 winners.  One 256-thread workgroup; K <= 128.  Writes K (value, index + i0)
// pairs, -inf / -1 behind the n-th.
__device__ __forceinline__ void radix_select_row(const float* x, int n, int i0, int K, float* out_val_row, int32_t* out_idx_row) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel_prefix, sel_remaining;
    __shared__ int cnt_gt, cnt_eq;
    __shared__ float cval[128];
    __shared__ int cidx[128];
    __shared__ int wave_eq[4];
    const int tid = threadIdx.x;
    const int Keff = min(K, n);
    uint32_t prefix = 0u, mask = 0u;
    int remaining = Keff;
    for (int pass = 3; pass >= 0; --pass) {
        hist[tid] = 0u;
        __syncthreads();
        for (int i = tid; i < n; i += blockDim.x) {
            const uint32_t k = float_key(x[i]);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> (pass * 8)) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0, b = 255;
            for (; b > 0; --b) {
                if (acc + (int)hist[b] >= remaining) break;
                acc += hist[b];
            }
            sel_prefix = prefix | ((uint32_t)b << (pass * 8));
            sel_remaining = remaining - acc;
        }
        __syncthreads();
        prefix = sel_prefix;
        remaining = sel_remaining;
        mask |= 255u << (pass * 8);
        __syncthreads();
    }
    // prefix = key of the K-th largest value; `remaining` of the elements equal to it are taken (lowest index first)
    if (tid == 0) { cnt_gt = 0; cnt_eq = 0; }
    for (int i = tid; i < 128; i += blockDim.x) { cval[i] = -INFINITY; cidx[i] = 0x7fffffff; }
    __syncthreads();
    const int n_gt = Keff - remaining;
    // elements strictly greater: any order (sorted afterwards); equal: need the `remaining` lowest indices
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + tid;
        bool is_gt = false, is_eq = false;
        float v = 0.f;
        if (i < n) {
            v = x[i];
            const uint32_t k = float_key(v);
            is_gt = k > prefix; is_eq = k == prefix;
        }
        if (is_gt) {
            const int pos = atomicAdd(&cnt_gt, 1);
            cval[pos] = v; cidx[pos] = i;
        }
        // equal elements in index order: ballot-based ordered append within the block pass
        const unsigned long long bal = __ballot(is_eq);
        const int lane = tid & 63, w = tid >> 6;
        if (lane == 0) wave_eq[w] = __popcll(bal);
        __syncthreads();
        int offs = cnt_eq;
        for (int ww = 0; ww < w; ++ww) offs += wave_eq[ww];
        if (is_eq) {
            const int pos = offs + __popcll(bal & ((1ull << lane) - 1ull));
            if (pos < remaining) { cval[n_gt + pos] = v; cidx[n_gt + pos] = i; }
        }
        __syncthreads();
        if (tid == 0) cnt_eq += wave_eq[0] + wave_eq[1] + wave_eq[2] + wave_eq[3];
        __syncthreads();
    }
    // bitonic sort of 128 candidates by (value desc, index asc)
    for (int k = 2; k <= 128; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (tid < 128) {
                const int ixj = tid ^ j;
                if (ixj > tid) {
                    const float a = cval[tid], b = cval[ixj];
                    const int ia = cidx[tid], ib = cidx[ixj];
                    const bool a_first = (a > b) || (a == b && ia < ib);  // a should precede b
                    const bool up = (tid & k) == 0;
                    if (up ? !a_first : a_first) { cval[tid] = b; cval[ixj] = a; cidx[tid] = ib; cidx[ixj] = ia; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < K; i += blockDim.x) {
        out_val_row[i] = i < Keff ? cval[i] : -INFINITY;
        out_idx_row[i] = i < Keff ? cidx[i] + i0 : -1;
    }
}


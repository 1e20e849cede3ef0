// Batch preparation of a training step — the row compaction map of the scoring and the slot data of the TPP regulariser: both read
// labels / masked positions / timestamps only.  As device functions, so that they run either as kernels of their own
// (edgl_compact_scan_labels: k_score.hip; edgl_tpp_prep: k_misc.hip) or as extra workgroups of the encoder's launch
// (edgl_encode_fwd_prep: k_encode.hip) — on the main stream, in front of everything that reads them, without a side stream whose
// join the first attention kernel would wait for.
#pragma once
#include "bimau_common.h"
#include "edgl_common.h"

namespace batch_prep {

__device__ __forceinline__ float raw_span(const float* ts_row, int pos, int T) {
    // EasyDGL.py:161-162 on RAW seconds: span[t] = clip(ts[t]-ts[t-1], 0, 100), span[0] := span[1]
    if (T < 2) return 0.f;
    const int t1 = pos == 0 ? 1 : pos;
    return fminf(fmaxf(ts_row[t1] - ts_row[t1 - 1], 0.f), 100.f);
}

// slot data of sample b (256 threads; tpp_smem: T * 16 + 2 * 256 * 4 bytes, 16-byte aligned) — see edgl_tpp_prep
__device__ __forceinline__ void tpp_prep_sample(const int64_t* mpos, const int64_t* labels, const float* ts, const uint8_t* mtab, int B,
                                                int T, int M, char* desc, int b, char* tpp_smem) {
    uint4* nm_s = reinterpret_cast<uint4*>(tpp_smem);                 // [T] mark rows of the positions' first effective slots
    int* pos_s = reinterpret_cast<int*>(nm_s + T);                    // [256] position of an effective slot, -1 otherwise
    int* ovf_s = pos_s + 256;                                         // [256] 1: effective, not the first of its position
    const bimau::TppLayout lay = bimau::tpp_layout(B, T, M);
    const int m = threadIdx.x;
    for (int t = m; t < T; t += 256) nm_s[t] = make_uint4(0u, 0u, 0u, 0u);
    int pos = -1;
    uint4 nm = make_uint4(0u, 0u, 0u, 0u);
    if (m < M) {
        const int pin = (int)mpos[(long)b * M + m];
        const int64_t lab = labels[(long)b * M + m];
        nm = *reinterpret_cast<const uint4*>(mtab + lab * 16);
        if (pin >= 0 && pin < T && (nm.x | nm.y | nm.z | nm.w) != 0u) pos = pin;
    }
    pos_s[m] = pos;
    {   // marks of this slot's label (valid position or not, as edgl_tpp_norm counts them)
        const uint32_t ws[4] = {nm.x, nm.y, nm.z, nm.w};
        int c = 0;
        if (m < M) {
#pragma unroll
            for (int q = 0; q < 4; ++q) c += (int)((ws[q] & 0xffu) + ((ws[q] >> 8) & 0xffu) + ((ws[q] >> 16) & 0xffu) + (ws[q] >> 24));
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((m & 63) == 0) ovf_s[m >> 6] = c;     // (ovf_s is written for real behind the next barrier)
    }
    __syncthreads();
    const int cnt_b = ovf_s[0] + ovf_s[1] + ovf_s[2] + ovf_s[3];
    __syncthreads();
    bool first = pos >= 0;
    for (int q = 0; q < m; ++q) first = first && pos_s[q] != pos;
    ovf_s[m] = (pos >= 0 && !first) ? 1 : 0;
    if (pos >= 0 && first) nm_s[pos] = nm;
    __syncthreads();
    uint4* nmw = reinterpret_cast<uint4*>(desc) + (long)b * T;
    float* spr = reinterpret_cast<float*>(desc + lay.off_spr) + (long)b * T;
    for (int t = m; t < T; t += 256) {
        const uint4 w = nm_s[t];
        nmw[t] = w;
        spr[t] = (w.x | w.y | w.z | w.w) != 0u ? raw_span(ts + (long)b * T, t, T) : -1.0f;
    }
    int rank = 0, total = 0;
    for (int q = 0; q < M; ++q) { rank += q < m ? ovf_s[q] : 0; total += ovf_s[q]; }
    if (m == 0) reinterpret_cast<int*>(desc + lay.off_novf)[b] = total;
    if (m < M && ovf_s[m]) {
        reinterpret_cast<int*>(desc + lay.off_ovf_pos)[(long)b * M + rank] = pos;
        reinterpret_cast<uint4*>(desc + lay.off_ovf_nm)[(long)b * M + rank] = nm;
    }
    if (m == 0) reinterpret_cast<int*>(desc + lay.off_cntp)[b] = cnt_b;
}

// Row compaction map (perm[j] = original row of compact row j, inv[r] = compact index of row r or -1, nvalid = #weighted rows,
// labels_c = the labels of the weighted rows first, 0 behind them) by ONE workgroup of NT threads (NT / 64 <= 16 waves): groups of 16
// chunks of NT consecutive rows — one coalesced label per thread and chunk, all 16 fetched together, wave ballots, the wave counts
// of the 16 chunks through LDS behind ONE barrier per group (two count buffers in turn; the running base is the same number in
// every thread: nothing shared to update).  wcnt: 2 * 16 * (NT / 64) ints of LDS.
template <int NT>
__device__ __forceinline__ void compact_scan_body(const int64_t* labels, int R, int32_t* perm, int32_t* inv, int32_t* nvalid,
                                                  int64_t* labels_c, int* wcnt) {
    constexpr int NW = NT / 64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const unsigned long long below = (1ull << lane) - 1ull;
    int base = 0, buf = 0;
    for (int g0 = 0; g0 < R; g0 += 16 * NT, buf ^= 1) {
        int64_t labk[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) labk[k] = labels[min(g0 + k * NT + t, R - 1)];      // (clamped, unconditional)
        unsigned long long bal[16];
        int* wc = wcnt + buf * 16 * NW;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int r = g0 + k * NT + t;
            bal[k] = __ballot(r < R && labk[k] != 0);
            if (lane == 0) wc[k * NW + w] = __popcll(bal[k]);
        }
        lds_barrier();
        int run = base;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            int before = run, tot = 0;
#pragma unroll
            for (int i = 0; i < NW; ++i) { const int c = wc[k * NW + i]; before += i < w ? c : 0; tot += c; }
            const int r = g0 + k * NT + t;
            if (r < R) {
                if ((bal[k] >> lane) & 1ull) {
                    const int pos = before + __popcll(bal[k] & below);
                    perm[pos] = r; inv[r] = pos;
                    if (labels_c) labels_c[pos] = labk[k];
                } else {
                    inv[r] = -1;
                }
            }
            run += tot;
        }
        base = run;
    }
    for (int j = base + t; j < R; j += NT) { perm[j] = -1; if (labels_c) labels_c[j] = 0; }
    if (t == 0) nvalid[0] = base;
}

}  // namespace batch_prep

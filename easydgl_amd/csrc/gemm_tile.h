// 128-row operand tile staging shared by the dense GEMM (k_gemm.hip) and the scoring kernels
// (k_score.hip): HBM -> registers (coalesced 16-byte vectors, guarded at the edges) -> LDS as
// S[row][k] with 16-byte padded rows, so that every MFMA fragment is a single ds_read_b128.
#pragma once
#include "edgl_common.h"

namespace tile {

constexpr int BM = 128, BN = 128, NT = 256;

template <typename T>
__device__ __forceinline__ Vec16<T> guarded_vec(const T* base, long off, int avail, bool vec_ok) {
    // `avail` = number of valid contiguous elements starting at base+off (may be <= 0)
    constexpr int VEC = ElemTraits<T>::VEC;
    if (avail >= VEC && vec_ok) return ld16<T>(base + off);
    Vec16<T> r = zero16<T>();
#pragma unroll
    for (int j = 0; j < VEC; ++j)
        if (j < avail) r.v[j] = base[off + j];
    return r;
}

// One operand tile: ROWS(=128) x BK, "row" = output index (m or n), staged as S[row][k].
template <typename T, bool KC>
struct Stager {
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int BK = 2 * ElemTraits<T>::KB;
    static constexpr int LDK = BK + VEC;  // padded LDS row (elements)
    static constexpr int KV = BK / VEC;   // vectors along k per row (= 8)
    // KC: 128*KV vectors / 256 threads = 4 per thread.
    // !KC: (128/VEC) x KV blocks of VECxVEC; per thread ceil(blocks/256) blocks of VEC vectors.
    static constexpr int NBLK = (128 / VEC) * KV;
    static constexpr int BPT = (NBLK + NT - 1) / NT;
    static constexpr int NREG = KC ? 4 : BPT * VEC;
    Vec16<T> reg[NREG];

    // rows_total: extent of the row index (M or N); K: contraction extent (k < kend valid)
    // zero_row0: global row 0 reads as zeros (the zero-padded embedding row, coding.py:56-57)
    __device__ __forceinline__ void load(const T* base, int ld, int row0, int rows_total, int k0, int kend,
                                         bool vec_ok, bool zero_row0 = false) {
        const int tid = threadIdx.x;
        if constexpr (KC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = tid + i * NT;
                const int row = v / KV, kv = v % KV;
                const int gr = row0 + row, gk = k0 + kv * VEC;
                const int avail = (gr < rows_total && !(zero_row0 && gr == 0)) ? (kend - gk) : 0;
                reg[i] = guarded_vec<T>(base, (long)gr * ld + gk, avail, vec_ok);
            }
        } else {
#pragma unroll
            for (int b = 0; b < BPT; ++b) {
                const int bid = tid + b * NT;
                const int rb = bid % (128 / VEC), kb = bid / (128 / VEC);
#pragma unroll
                for (int kk = 0; kk < VEC; ++kk) {
                    const int gk = k0 + kb * VEC + kk, gr = row0 + rb * VEC;
                    const int avail = (bid < NBLK && gk < kend) ? (rows_total - gr) : 0;
                    reg[b * VEC + kk] = guarded_vec<T>(base, (long)gk * ld + gr, avail, vec_ok);
                }
            }
        }
    }
    __device__ __forceinline__ void store(T* S) {
        const int tid = threadIdx.x;
        if constexpr (KC) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int v = tid + i * NT;
                const int row = v / KV, kv = v % KV;
                st16<T>(S + row * LDK + kv * VEC, reg[i]);
            }
        } else {
#pragma unroll
            for (int b = 0; b < BPT; ++b) {
                const int bid = tid + b * NT;
                if (bid < NBLK) {
                    const int rb = bid % (128 / VEC), kb = bid / (128 / VEC);
#pragma unroll
                    for (int rr = 0; rr < VEC; ++rr) {
                        Vec16<T> o;
#pragma unroll
                        for (int kk = 0; kk < VEC; ++kk) o.v[kk] = reg[b * VEC + kk].v[rr];
                        st16<T>(S + (rb * VEC + rr) * LDK + kb * VEC, o);
                    }
                }
            }
        }
    }
};


}  // namespace tile

// K5/K6/K7: tied-embedding scoring against the item table.
//   logits[r, n] = rows[r,:] . table[n,:] + bias[n]   (EasyDGL.py:149-150; table row 0 acts as zeros,
//   bias = concat([-1000], output_bias) — coding.py:56-57, Base.py:106-110)
// Training never materialises the [R, I] logits: the forward keeps an online (max, sum-exp) per row
// (EasyDGL.py:155), the backward recomputes logit tiles and feeds dl = coef*(p - onehot) straight back into
// MFMA as an operand.  All three kernels share one structure:
//   * a workgroup = 8 waves (512 threads) owns 256 "x" vectors (rows, or table rows) whose MFMA fragments
//     stay in REGISTERS for the whole kernel; each wave owns a 32-x strip and the full width of every
//     streamed tile, so no cross-wave reduction is ever needed;
//   * the other operand ("z": item tiles, or row tiles) streams through LDS in 128-z tiles, double buffered,
//     prefetched through registers, one barrier per tile;
//   * logit tile in the swapped orientation  D[z][x]  (register layout L(first=z, second=x)), so that dl is
//     directly the A operand of the second product  out[x][c] += sum_z dl[x][z] Z[z][c]; its B operand (z along
//     the contraction) comes out of the SAME row-major LDS tile through gfx950's transpose read
//     (ds_read_b64_tr_b16, bf16) — one image per tile, no transposed copies of the operands in HBM.  f32 has no
//     32-bit transpose read: there a second LDS image in [c][z] order is filled from a pre-transposed copy
//     (tableT [C][ldt] / rowsT [C][ldr]).
//   score_fwd_kernel : x = rows,  z = items   -> per-chunk (max, sumexp) [+ optional logits]
//   score_bwd<ROLE_Y>: x = rows,  z = items   -> d_rows slabs
//   score_bwd<ROLE_W>: x = items, z = rows    -> d_table slabs, d_bias slabs
#include <cstdlib>
#include <mutex>
#include <unordered_map>

#include "edgl_common.h"
#include "score_plan.h"
#include "topk_select.h"
#include "batch_prep.h"

// csrc/k_score_strip.hip: one-wave-per-SIMD form of the two product passes (bf16, C = 128)
bool edgl_strip_enabled();
int edgl_strip_rows(const void* rows, const void* table, const float* out_bias, int R, int I, int i0, int i1, const int32_t* nvalid,
                    float* slabs, float* part, int G, int slab16, hipStream_t st);
int edgl_strip_table(const void* rows, const void* table, const float* out_bias, const float* coef, const float* row_lse, int R,
                     int I, int i0, int i1, const int32_t* nvalid, float* slabs, float* bias_slabs, int nchunk, float* acc_table,
                     float* acc_bias, hipStream_t st);
// k_score_stripw.hip: the same passes at C = 256 (32 x vectors per wave, 128 per workgroup)
bool edgl_stripw_enabled();
bool edgl_stripw_supports(int C);
long edgl_stripw_info_floats(long n);
int edgl_stripw_rows(const void* rows, const void* table, const float* out_bias, int R, int C, int I, int i0, int i1,
                     const int32_t* nvalid, float* slabs, float* part, int G, float* info_ws, hipStream_t st);
int edgl_stripw_table(const void* rows, const void* table, const float* out_bias, const float* coef, const float* row_lse, int R,
                      int C, int I, int i0, int i1, const int32_t* nvalid, float* slabs, float* bias_slabs, int nchunk, float* info_ws,
                      hipStream_t st);
int edgl_stripw_label_scatter(const void* rows, const int64_t* labels, const float* coef, const int32_t* nvalid, int R, int C, int i0,
                              int i1, const float* gscale, float* d_table, float* d_bias, hipStream_t st);
int edgl_strip_label_scatter(const void* rows, const int64_t* labels, const float* coef, const int32_t* nvalid, int R, int i0, int i1,
                             const float* gscale, float* d_table, float* d_bias, hipStream_t st);

namespace {

constexpr int SNT = 512;   // threads per workgroup of the forward kernel

// Tile shape per (dtype, C).  C <= 128 (and bf16 C = 256): a wave keeps TWO 16-x tiles in registers and 128-z tiles stream
// through LDS.  Wider rows (bf16 C = 512 — every published recipe, runme.sh:15-115; f32 C = 256 / 512 for parity runs) keep
// ONE x tile per wave (its fragments alone are 64-128 registers) and stream narrower z tiles so that the Z / Z^T images of
// a tile still fit the 160 KB of LDS.
template <typename T, int CT>
struct ScoreCfg {
    static constexpr bool BF = sizeof(T) == 2;
    static constexpr bool STD = CT <= 8 || (BF && CT == 16);
    static constexpr int IX = STD ? 2 : 1;                                   // 16-x tiles per wave
    static constexpr int ZB = STD ? 128 : ((BF || CT == 16) ? 64 : 32);      // z vectors per streamed tile
    static constexpr int NWB = (!BF && CT == 32) ? 4 : 8;                    // waves per workgroup, backward kernels
    // output channel tiles per backward pass: bf16 C = 256 / 512 accumulate half the channels per pass (the logits are
    // recomputed per pass) — all [IX][CT] accumulator tiles next to the x fragments spill ~100-300 registers
    static constexpr int CO = (BF && CT >= 16) ? CT / 2 : CT;
};
struct RtCfg { int ix, zb, nwb; };
inline RtCfg rt_cfg(int C, size_t esize) {   // the same table for the host-side planners
    const int ct = C / 16;
    const bool bf = esize == 2, stdc = ct <= 8 || (bf && ct == 16);
    return RtCfg{stdc ? 2 : 1, stdc ? 128 : ((bf || ct == 16) ? 64 : 32), (!bf && ct == 32) ? 4 : 8};
}

template <typename T, int CT, int NTHR = SNT, int ZBT = ScoreCfg<T, CT>::ZB>
struct SC {
    static constexpr int ZB = ZBT;
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int KB = ElemTraits<T>::KB;
    static constexpr int C = 16 * CT;
    static constexpr int NKB = C / KB;
    static constexpr int LDC = C + 2 * VEC;   // Z  image: [ZB][LDC]   (row z, k contiguous); +32 B: conflict-free b128 reads
    static constexpr int LDZ = ZB + VEC;      // ZT image: [C][LDZ]    (row c, z contiguous)
    static constexpr int CV = C / VEC;
    static constexpr int PER_Z = ZB * CV / NTHR;          // 16-byte vectors per thread, Z image
    static constexpr int PER_ZT = C * (ZB / VEC) / NTHR;  // same count, ZT image
    static constexpr size_t Z_BYTES = (size_t)ZB * LDC * sizeof(T);
    // second image of the tile in [c][z] order: f32 only — bf16 fetches the z-contracting operand of the second product from
    // the Z image itself with the LDS transpose read (ds_read_b64_tr_b16; row stride LDC * 2 B = 8 banks mod 64: conflict-free)
    static constexpr bool WT = sizeof(T) == 4;
    static constexpr size_t ZT_BYTES = WT ? (size_t)C * LDZ * sizeof(T) : 0;
    static constexpr size_t INFO_BYTES = 3 * ZB * sizeof(float);
    static constexpr int JH = ZB >= 64 ? 4 : ZB / 16;     // 16-z tiles per "half" (the unit whose logits are live at once)
    static constexpr int NH = ZB / 16 / JH;               // halves per streamed tile
    static_assert(PER_Z >= 1 && PER_ZT >= 1 && ZB % 16 == 0, "tile does not divide over the workgroup");
};

typedef __attribute__((ext_vector_type(4))) short sc_s16x4;
// B operand of a 16x16x32 MFMA whose contraction index runs along the ROWS of a row-major bf16 LDS tile (rows k0 .. k0+31,
// columns z0 .. z0+15): slots 0-3 <-> k0 + 4G + j, slots 4-7 <-> k0 + 16 + 4G + j (G = lane >> 4) — the order in which the
// dl fragment is packed from the logit accumulators (see csrc/k_gemm2.hip for the measured lane map of the instruction)
__device__ __forceinline__ bf16x8 sc_tr_frag32(const bf16* tile, int ld, int k0, int z0, int lane) {
    const int G = lane >> 4, sl = lane & 15;
    const bf16* p = tile + (k0 + 4 * G + (sl >> 2)) * ld + z0 + 4 * (sl & 3);
    bf16x8 f;
    sc_s16x4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sc_s16x4*)p);
    sc_s16x4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) sc_s16x4*)(p + 16 * ld));
    *reinterpret_cast<uint2*>(&f) = *reinterpret_cast<uint2*>(&v0);
    *(reinterpret_cast<uint2*>(&f) + 1) = *reinterpret_cast<uint2*>(&v1);
    return f;
}

struct ScoreP {
    const void* rows; const void* rowsT; int ldr;      // rows [R][C], rowsT [C][ldr]
    const void* table; const void* tableT; int ldt;    // table [I][C], tableT [C][ldt]
    const float* out_bias; const int64_t* labels;
    int R, C, I, i0, i1;
    const int32_t* nvalid;                              // optional device count: only rows < *nvalid carry weight
    float* row_lse; float* part; float* logits;         // fwd
    int zchunk, nchunk;                                 // z vectors per block (multiple of ZB), #chunks
    const float* coef; const float* gscale;             // bwd
    float* slabs; float* bias_slabs;
    float* lab_out;                                     // flash forward: label logits [R]
    float* coef_out;                                    // flash forward, compacted rows, whole table: loss coefficients [R]
    int table_ready;                                    // tableT already written by edgl_score_prepare_table
    const int32_t* wtotal;                              // data parallel: weighted rows of the GLOBAL batch (denominator of the loss)
    bool defer_label;                                   // strip path: the caller applies the one-hot term (edgl_score_flash_label_term)
    bool acc_atomic;                                    // strip path: d_table / d_bias are zero-filled and take the chunks as f32 atomics
    bool keep_slabs;                                    // the row-chunk slabs of d_table / d_bias stay in the workspace: no slab_reduce launch (edgl_adam_apply_ex sums them)
    float* ce_part;                                     // one-launch row finish: per-workgroup sums of -log(p_label + 1e-5) (edgl_score_ce_nparts)
};

// ---- streamed tile: global -> registers -> LDS -------------------------------------------------
template <typename T, int CT, bool WITH_T, int NTHR = SNT>
struct ZStream {
    using S = SC<T, CT, NTHR>;
    Vec16<T> rz[S::PER_Z];
    Vec16<T> rt[WITH_T ? S::PER_ZT : 1];
    int z0_, zend_;      // tile start / chunk end of the rows in flight
    bool zero0_;
    // Z image rows [z0, z0+ZB) of src [*, C]; rows >= zend (or global row 0 when zero_row0) read as zeros.
    // The loads are UNCONDITIONAL (addresses clamped into the valid range) and the zeroing happens in store(): a
    // per-lane branch around a load makes the compiler wait for that load inside the branch, which serialises the
    // whole prefetch (one full memory latency per slot).
    __device__ __forceinline__ void load(const T* src, const T* srcT, int ldT, int z0, int zend, bool zero_row0) {
        load_z(src, z0, zend, zero_row0);
        load_t(srcT, ldT, z0, zend, zero_row0);
    }
    __device__ __forceinline__ void load_z(const T* src, int z0, int zend, bool zero_row0) {
        z0_ = z0; zend_ = zend; zero0_ = zero_row0;
#pragma unroll
        for (int i = 0; i < S::PER_Z; ++i) {
            const int v = threadIdx.x + i * NTHR;
            const int row = v / S::CV, cv = v % S::CV, gz = min(z0 + row, zend - 1);
            rz[i] = ld16<T>(src + (long)gz * S::C + cv * S::VEC);
        }
    }
    __device__ __forceinline__ void load_t(const T* srcT, int ldT, int z0, int zend, bool zero_row0) {
        if constexpr (WITH_T) {
            z0_ = z0; zend_ = zend; zero0_ = zero_row0;
            constexpr int ZV = S::ZB / S::VEC;
#pragma unroll
            for (int i = 0; i < S::PER_ZT; ++i) {
                const int v = threadIdx.x + i * NTHR;
                const int c = v / ZV, zv = v % ZV, gz = min(z0 + zv * S::VEC, ldT - S::VEC);   // ldT is a multiple of VEC
                rt[i] = ld16<T>(srcT + (long)c * ldT + gz);
            }
        }
    }
    __device__ __forceinline__ void store(T* Zs, T* ZTs) {
#pragma unroll
        for (int i = 0; i < S::PER_Z; ++i) {
            const int v = threadIdx.x + i * NTHR;
            const int gz = z0_ + v / S::CV;
            const bool ok = gz < zend_ && !(zero0_ && gz == 0);
            st16<T>(Zs + (v / S::CV) * S::LDC + (v % S::CV) * S::VEC, ok ? rz[i] : zero16<T>());
        }
        if constexpr (WITH_T) {
            constexpr int ZV = S::ZB / S::VEC;
#pragma unroll
            for (int i = 0; i < S::PER_ZT; ++i) {
                const int v = threadIdx.x + i * NTHR;
                const int gz = z0_ + (v % ZV) * S::VEC;
                Vec16<T> t = rt[i];
                if (gz + S::VEC > zend_ || (zero0_ && gz == 0)) {   // only the last tile of a chunk / the tile holding row 0
#pragma unroll
                    for (int j = 0; j < S::VEC; ++j)
                        if (gz + j >= zend_ || (zero0_ && gz + j == 0)) t.v[j] = from_f32<T>(0.f);
                }
                st16<T>(ZTs + (v / ZV) * S::LDZ + (v % ZV) * S::VEC, t);
            }
        }
    }
};

// x fragments of this wave's 32-x strip, kept in registers.  Two halves: issue_xfrags() sends every load (row clamped,
// unconditional), finish_xfrags() — called after the first streamed tile's loads have been issued as well — zeroes the
// fragments of rows past the end.  `ok ? load : zero` compiles to a branch around each load with its own wait: IX * NKB
// dependent round trips at the head of the kernel (8 at C = 128).
template <typename T, int CT, int IX>
__device__ __forceinline__ void issue_xfrags(const T* X, int x0, int xend, int lane, Vec16<T> (&xf)[IX][SC<T, CT>::NKB]) {
    using S = SC<T, CT>;
#pragma unroll
    for (int ix = 0; ix < IX; ++ix) {
        const int gx = max(min(x0 + ix * 16 + (lane & 15), xend - 1), 0);
#pragma unroll
        for (int kb = 0; kb < S::NKB; ++kb) xf[ix][kb] = ld16<T>(X + (long)gx * S::C + kb * S::KB + (lane >> 4) * S::VEC);
    }
}
template <typename T, int CT, int IX>
__device__ __forceinline__ void finish_xfrags(int x0, int xend, bool zero_row0, int lane, Vec16<T> (&xf)[IX][SC<T, CT>::NKB]) {
    using S = SC<T, CT>;
#pragma unroll
    for (int ix = 0; ix < IX; ++ix)
#pragma unroll
        for (int kb = 0; kb < S::NKB; ++kb) {      // the loaded values are "used" here: the loads cannot sink into a branch
            uint4& u = *reinterpret_cast<uint4*>(&xf[ix][kb]);
            asm volatile("" : "+v"(u.x), "+v"(u.y), "+v"(u.z), "+v"(u.w));
        }
#pragma unroll
    for (int ix = 0; ix < IX; ++ix) {
        const int gx = x0 + ix * 16 + (lane & 15);
        const bool ok = gx < xend && !(zero_row0 && gx == 0);
#pragma unroll
        for (int kb = 0; kb < S::NKB; ++kb) xf[ix][kb] = ok ? xf[ix][kb] : zero16<T>();
    }
}

// D[z][x] half tile of the wave (64 z = 4 jz tiles starting at jz0): acc[j][ix], L(first = z, second = x).
// The 128-z tile is processed in two halves so that only 32 accumulator registers are live at a time.
template <typename T, int CT, int IX>
__device__ __forceinline__ void logit_half(const T* Zs, int jz0, const Vec16<T> (&xf)[IX][SC<T, CT>::NKB], int lane,
                                           f32x4 (&acc)[SC<T, CT>::JH][IX]) {
    using S = SC<T, CT>;
#pragma unroll
    for (int j = 0; j < S::JH; ++j)
#pragma unroll
        for (int ix = 0; ix < IX; ++ix) acc[j][ix] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < S::NKB; ++kb) {
#pragma unroll
        for (int j = 0; j < S::JH; ++j) {
            const Vec16<T> zf = ld16<T>(Zs + ((jz0 + j) * 16 + (lane & 15)) * S::LDC + kb * S::KB + (lane >> 4) * S::VEC);
#pragma unroll
            for (int ix = 0; ix < IX; ++ix) acc[j][ix] = mma_kblock(zf, xf[ix][kb], acc[j][ix]);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward: online log-sum-exp per row over this block's item chunk (+ optional logits)
// ---------------------------------------------------------------------------------------------
template <typename T, int CT>
__global__ __launch_bounds__(SNT) void score_fwd_kernel(ScoreP p) {
    using S = SC<T, CT>;
    constexpr int IX = ScoreCfg<T, CT>::IX, ZB = S::ZB, XB = 16 * IX * (SNT / 64), JH = S::JH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr size_t BUF = S::Z_BYTES + ZB * sizeof(float);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int Reff = p.nvalid ? min(p.R, p.nvalid[0]) : p.R;
    const DevPlan dp = dev_plan(Reff, XB, gridDim.x, p.i1 - p.i0, ZB);
    if ((int)blockIdx.x >= dp.nx * dp.nchunk || Reff <= 0) return;
    const int bx = blockIdx.x % dp.nx, by = blockIdx.x / dp.nx;
    const int m0 = bx * XB + wave * 16 * IX;
    const int c_lo = p.i0 + by * dp.zchunk, c_hi = min(p.i1, c_lo + dp.zchunk);
    const T* rows = reinterpret_cast<const T*>(p.rows);
    const T* table = reinterpret_cast<const T*>(p.table);

    Vec16<T> xf[IX][S::NKB];
    issue_xfrags<T, CT, IX>(rows, m0, Reff, lane, xf);
    ZStream<T, CT, false> zs;
    const int ntile = (c_hi - c_lo + ZB - 1) / ZB;
    zs.load(table, nullptr, 0, c_lo, c_hi, true);
    finish_xfrags<T, CT, IX>(m0, Reff, false, lane, xf);
    {
        T* Zs = reinterpret_cast<T*>(smem);
        float* info = reinterpret_cast<float*>(smem + S::Z_BYTES);
        zs.store(Zs, nullptr);
        if (tid < ZB) {
            const int n = c_lo + tid;
            const float ob = p.out_bias[min(max(n, 1), p.I - 1) - 1];
            info[tid] = (n < c_hi && n > 0) ? ob : 0.f;
        }
    }
    __syncthreads();
    float rmax[IX], rsum[IX];
#pragma unroll
    for (int ix = 0; ix < IX; ++ix) { rmax[ix] = -INFINITY; rsum[ix] = 0.f; }
    for (int it = 0; it < ntile; ++it) {
        const int n0 = c_lo + it * ZB;
        const bool more = it + 1 < ntile;
        const char* cur = smem + (size_t)(it & 1) * BUF;
        char* nxt = smem + (size_t)((it + 1) & 1) * BUF;
        float ob_next = 0.f;      // bias of the next tile's items: loaded with the tile, stored after it (no wait in between)
        if (more) {
            zs.load(table, nullptr, 0, n0 + ZB, c_hi, true);
            if (tid < ZB) ob_next = p.out_bias[min(max(n0 + ZB + tid, 1), p.I - 1) - 1];
        }
        const T* Zs = reinterpret_cast<const T*>(cur);
        const float* info = reinterpret_cast<const float*>(cur + S::Z_BYTES);
        const bool edge = (n0 == 0) || (n0 + ZB > c_hi);
#pragma unroll
        for (int half = 0; half < S::NH; ++half) {
            f32x4 acc[JH][IX];
            logit_half<T, CT, IX>(Zs, half * JH, xf, lane, acc);
#pragma unroll
            for (int ix = 0; ix < IX; ++ix) {
                const int m = m0 + ix * 16 + l15;
                float tmax = -INFINITY;
#pragma unroll
                for (int j = 0; j < JH; ++j) {
                    const int jz = half * JH + j;
                    const float4 bz = *reinterpret_cast<const float4*>(info + jz * 16 + g4);
                    const float bb[4] = {bz.x, bz.y, bz.z, bz.w};
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = acc[j][ix][r] + bb[r];
                        if (edge) {   // only the tile holding column 0 or the chunk tail needs per-element tests
                            const int n = n0 + jz * 16 + g4 + r;
                            x = (n == 0) ? -1000.0f : x;       // zero-padded row . y + (-1000)  (Base.py:110)
                            x = (n < c_hi) ? x : -INFINITY;
                        }
                        acc[j][ix][r] = x;
                        tmax = fmaxf(tmax, x);
                    }
                }
                if (p.logits && m < Reff) {
                    float* dst = p.logits + (long)m * (p.i1 - p.i0) + (n0 - p.i0);
#pragma unroll
                    for (int j = 0; j < JH; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int zz = (half * JH + j) * 16 + g4 + r;
                            if (n0 + zz < c_hi) dst[zz] = acc[j][ix][r];
                        }
                }
                const float nm = fmaxf(rmax[ix], tmax);
                if (nm > -INFINITY) {
                    float sacc = rsum[ix] * __expf(rmax[ix] - nm);
#pragma unroll
                    for (int j = 0; j < JH; ++j)
#pragma unroll
                        for (int r = 0; r < 4; ++r) sacc += __expf(acc[j][ix][r] - nm);
                    rsum[ix] = sacc;
                    rmax[ix] = nm;
                }
            }
        }
        if (more) {
            zs.store(reinterpret_cast<T*>(nxt), nullptr);
            if (tid < ZB) {
                const int n = n0 + ZB + tid;
                reinterpret_cast<float*>(nxt + S::Z_BYTES)[tid] = (n < c_hi && n > 0) ? ob_next : 0.f;
            }
        }
        __syncthreads();
    }
    // the 4 lane groups of a wave hold disjoint z subsets of the same x: combine them
#pragma unroll
    for (int ix = 0; ix < IX; ++ix) {
        const float mx = group_max4(rmax[ix]);
        float s = (rmax[ix] > -INFINITY) ? rsum[ix] * __expf(rmax[ix] - mx) : 0.f;
        s = group_sum4(s);
        const int m = m0 + ix * 16 + l15;
        if (lane < 16 && m < Reff) {
            p.part[((long)m * dp.nchunk + by) * 2] = mx;
            p.part[((long)m * dp.nchunk + by) * 2 + 1] = s;
        }
    }
}

__global__ void lse_combine_kernel(const float* part, int R, const int32_t* nvalid, int xb, int zb, int G, int ztotal, float* row_lse) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= R) return;
    const int Reff = nvalid ? min(R, nvalid[0]) : R;
    if (m >= Reff) { row_lse[m] = 0.f; return; }
    const int nchunk = dev_plan(Reff, xb, G, ztotal, zb).nchunk;
    float mx = -INFINITY;
    for (int c = 0; c < nchunk; ++c) mx = fmaxf(mx, part[((long)m * nchunk + c) * 2]);
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) {
        const float pm = part[((long)m * nchunk + c) * 2];
        if (pm > -INFINITY) s += part[((long)m * nchunk + c) * 2 + 1] * __expf(pm - mx);
    }
    row_lse[m] = mx + __logf(s);
}

// lse_combine_kernel + label_logit_kernel in one launch (flash form): one wave per row
template <typename T>
__global__ __launch_bounds__(256) void lse_label_kernel(const float* part, int R, const int32_t* nvalid, int xb, int zb, int G, int ztotal,
                                                        float* row_lse, const T* rows, const T* table, const float* out_bias,
                                                        const int64_t* labels, int C, int i0, int i1, float* lab_out, float* coef_out,
                                                        const int32_t* wtotal) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    // two round trips: {row count, label}, then everything that hangs on them — chunk partials, row, label row of the table, bias —
    // as unconditional loads (indices clamped); guarded loads made this kernel six dependent round trips
    const int Reff = nvalid ? min(R, nvalid[0]) : R;
    const int64_t lab = labels[r];
    const int nchunk = dev_plan(max(Reff, 1), xb, G, ztotal, zb).nchunk;
    const bool own = lab >= i0 && lab < i1;
    const int64_t labc = own ? lab : (int64_t)i0;
    const long pidx = ((long)min(r, max(Reff - 1, 0)) * nchunk + min(lane, nchunk - 1)) * 2;
    float pm = part[pidx], ps = part[pidx + 1];
    float ob = out_bias[max(labc, (int64_t)1) - 1];
    float a = 0.f;
    if (C <= 128) {
        const int c0 = min(lane, C - 1), c1 = min(lane + 64, C - 1);
        float x0 = to_f32(rows[(long)r * C + c0]), x1 = to_f32(rows[(long)r * C + c1]);
        float t0 = to_f32(table[labc * C + c0]), t1 = to_f32(table[labc * C + c1]);
        asm volatile("" : "+v"(pm), "+v"(ps), "+v"(ob), "+v"(x0), "+v"(x1), "+v"(t0), "+v"(t1));
        a = lane < C ? x0 * t0 : 0.f;                       // the summation order of label_logit_kernel (bit-equal results)
        a = lane + 64 < C ? fmaf(x1, t1, a) : a;
    } else {
        asm volatile("" : "+v"(pm), "+v"(ps), "+v"(ob));
        for (int c = lane; c < C; c += 64) a += to_f32(rows[(long)r * C + c]) * to_f32(table[labc * C + c]);
    }
    a = (lab != 0 && own) ? a : 0.f;
    float lse = 0.f;
    if (r < Reff) {
        float mx = lane < nchunk ? pm : -INFINITY;
        for (int c = lane + 64; c < nchunk; c += 64) mx = fmaxf(mx, part[((long)r * nchunk + c) * 2]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        float sm = (lane < nchunk && pm > -INFINITY) ? ps * __expf(pm - mx) : 0.f;
        for (int c = lane + 64; c < nchunk; c += 64) {
            const float qm = part[((long)r * nchunk + c) * 2];
            if (qm > -INFINITY) sm += part[((long)r * nchunk + c) * 2 + 1] * __expf(qm - mx);
        }
        sm = wave_sum(sm);
        lse = mx + __logf(sm);
    }
    a = wave_sum(a);
    if (lane == 0) {
        row_lse[r] = lse;
        const float ll = (lab == 0) ? -1000.0f : a + ob;
        if (own) lab_out[r] = ll;
        if (coef_out) {   // d loss / d logit scale of the row: ce_loss_kernel's coefficient, its row count known as *nvalid
            const float v = __expf(ll - lse), W = (float)(wtotal ? wtotal[0] : Reff) + 1e-5f;   // EasyDGL.py:184 over the global batch
            coef_out[r] = (r < Reff && lab != 0) ? (1.f / W) * (v / (v + 1e-5f)) : 0.f;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void label_logit_kernel(const T* rows, const T* table, const float* out_bias,
                                                          const int64_t* labels, int R, int C, int i0, int i1, float* out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    const int64_t lab = labels[r];
    if (lab < i0 || lab >= i1) return;
    float a = 0.f;
    if (lab != 0)
        for (int c = lane; c < C; c += 64) a += to_f32(rows[(long)r * C + c]) * to_f32(table[lab * C + c]);
    a = wave_sum(a);
    if (lane == 0) out[r] = (lab == 0) ? -1000.0f : a + out_bias[lab - 1];
}

// ---------------------------------------------------------------------------------------------
// backward: ROLE_Y (x = rows, z = items) -> d_rows ; ROLE_W (x = items, z = rows) -> d_table, d_bias
// ---------------------------------------------------------------------------------------------
// ROLE_YF ("flash" form of ROLE_Y, used by edgl_score_flash_*): the SAME pass also produces the row log-sum-exp, so the separate
// forward LSE kernel disappears — dl is exp(logit - running row max) and the accumulators are rescaled whenever the running
// max of a row moves (rare after the first tiles); the finish kernel divides by the row sum, applies the loss coefficient
// and subtracts the label row.
enum { ROLE_Y = 0, ROLE_W = 1, ROLE_YF = 2 };

// NW = 8: one workgroup per CU, double-buffered LDS, register prefetch across the compute phase.
// NW = 4: two independent 4-wave workgroups per CU, single LDS buffer, next tile fetched at the tile boundary —
//         the two workgroups drift apart, so one's MFMA phase overlaps the other's exp/VALU and load phases.
// CO = output channel tiles accumulated per workgroup: CT (one pass) or CT/2 (C = 256: two passes over the channel halves —
// the logits are recomputed per pass, but [2][16] accumulator tiles + operands do not fit 256 registers and spill ~300)
template <typename T, int CT, int ROLE, int NW, int CO = CT>
__global__ __launch_bounds__(64 * NW) void score_bwd_kernel(ScoreP p) {
    constexpr int IX = ScoreCfg<T, CT>::IX;
    constexpr int NTHR = 64 * NW, XBW = 16 * IX * NW;
    constexpr bool YS = ROLE != ROLE_W, FLASH = ROLE == ROLE_YF;   // YS: x = rows, z = items
    const int ct0 = (CO == CT) ? 0 : (YS ? (int)blockIdx.y : (int)blockIdx.z) * CO;
    using S = SC<T, CT, NTHR>;
    constexpr int ZB = S::ZB, JH = S::JH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr size_t BUF = S::Z_BYTES + S::ZT_BYTES + S::INFO_BYTES;
    constexpr bool DOUBLE = (NW == 8) && (2 * BUF <= 160 * 1024);
    constexpr bool PREFETCH = (NW == 8) && ScoreCfg<T, CT>::STD;   // the wide shapes have no registers to spare for the next tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const T* rows = reinterpret_cast<const T*>(p.rows);
    const T* rowsT = reinterpret_cast<const T*>(p.rowsT);
    const T* table = reinterpret_cast<const T*>(p.table);
    const T* tableT = reinterpret_cast<const T*>(p.tableT);
    const int Reff = p.nvalid ? min(p.R, p.nvalid[0]) : p.R;
    // ROLE_Y: 1-D launch, (x-block, item chunk) derived from the valid row count; ROLE_W: x = items (blockIdx.x),
    // the valid rows are split evenly over gridDim.y chunks
    int bx, by, zchunk, nchunk_dev = 1;
    long slab_stride;
    if (YS) {
        const DevPlan dp = dev_plan(Reff, XBW, gridDim.x, p.i1 - p.i0, ZB);
        if ((int)blockIdx.x >= dp.nx * dp.nchunk || Reff <= 0) return;
        bx = blockIdx.x % dp.nx; by = blockIdx.x / dp.nx; zchunk = dp.zchunk; nchunk_dev = dp.nchunk;
        slab_stride = (long)dp.nx * XBW * S::C;
    } else {
        bx = blockIdx.x; by = blockIdx.y;
        const int ntiles = (Reff + ZB - 1) / ZB;
        zchunk = (ntiles + (int)gridDim.y - 1) / (int)gridDim.y * ZB;
        slab_stride = (long)p.I * S::C;
    }
    // x side
    const int xbase = (YS ? 0 : p.i0) + bx * XBW + wave * 16 * IX;
    const int xend = YS ? Reff : p.i1;
    Vec16<T> xf[IX][S::NKB];
    issue_xfrags<T, CT, IX>(YS ? rows : table, xbase, xend, lane, xf);
    float x_lse[IX], x_cf[IX], x_bias[IX];
    int x_lab[IX];
    float m_run[IX], s_run[IX];   // ROLE_YF: running row max (uniform over the 4 lane groups) and this lane's part of the row sum
#pragma unroll
    for (int ix = 0; ix < IX; ++ix) {
        const int gx = xbase + ix * 16 + l15;
        const bool ok = gx < xend;
        m_run[ix] = -INFINITY; s_run[ix] = 0.f;
        if (FLASH) {
            x_lse[ix] = 0.f; x_cf[ix] = 0.f; x_lab[ix] = -1; x_bias[ix] = 0.f;
        } else if (ROLE == ROLE_Y) {
            const int gc = max(min(gx, xend - 1), 0);
            const float cf = p.coef[gc], ls = p.row_lse[gc];
            const int64_t lb = p.labels[gc];
            x_cf[ix] = ok ? cf : 0.f;
            // coef*exp(x - lse) = exp(x - (lse - log coef)); coef == 0 (label 0 / row past the end) -> +inf -> 0
            x_lse[ix] = x_cf[ix] > 0.f ? ls - __logf(x_cf[ix]) : INFINITY;
            x_lab[ix] = ok ? (int)lb : -1;
            x_bias[ix] = 0.f;
        } else {
            const float ob = p.out_bias[min(max(gx, 1), p.I - 1) - 1];     // unconditional load, then the select
            x_bias[ix] = (ok && gx > 0) ? ob : 0.f;
            x_lse[ix] = 0.f; x_cf[ix] = 0.f; x_lab[ix] = ok ? gx : -2;   // x_lab = own item id
        }
    }
    // z side
    const int z_lo = (YS ? p.i0 : 0) + by * zchunk;
    const int z_hi = min(YS ? p.i1 : Reff, z_lo + zchunk);
    const T* zsrc = YS ? table : rows;
    const T* zsrcT = YS ? tableT : rowsT;
    const int ldT = YS ? p.ldt : p.ldr;

    // per-z scalars of a streamed tile (ROLE_Y*: bias[z]; ROLE_W: lse[z] - log coef[z], coef[z], label[z]), threads < ZB: the
    // loads are unconditional (index clamped) and issued with the tile's own loads, the LDS stores follow the tile's —
    // a load inside `ok ? .. : 0` waits for its data on the spot, one exposed round trip per streamed tile
    float zi_a = 0.f, zi_b = 0.f;
    int64_t zi_c = 0;
    auto load_info = [&](int z0) {
        if (tid < ZB) {
            const int gz = z0 + tid;
            if (YS) {
                zi_a = p.out_bias[min(max(gz, 1), p.I - 1) - 1];
            } else {
                const int gc = min(gz, Reff - 1);
                zi_a = p.coef[gc]; zi_b = p.row_lse[gc]; zi_c = p.labels[gc];
            }
        }
    };
    auto store_info = [&](char* buf, int z0) {
        float* info = reinterpret_cast<float*>(buf + S::Z_BYTES + S::ZT_BYTES);
        if (tid < ZB) {
            const int gz = z0 + tid;
            const bool ok = gz < z_hi;
            if (YS) {
                info[tid] = (ok && gz > 0) ? zi_a : 0.f;
            } else {
                const float cf = ok ? zi_a : 0.f;
                info[tid] = cf > 0.f ? zi_b - __logf(cf) : INFINITY;
                info[ZB + tid] = cf;
                reinterpret_cast<int*>(info)[2 * ZB + tid] = ok ? (int)zi_c : -1;
            }
        }
    };

    f32x4 out[IX][CO];
#pragma unroll
    for (int ix = 0; ix < IX; ++ix)
#pragma unroll
        for (int ct = 0; ct < CO; ++ct) out[ix][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float dbias[IX];
#pragma unroll
    for (int ix = 0; ix < IX; ++ix) dbias[ix] = 0.f;

    ZStream<T, CT, S::WT, NTHR> zs;
    const int ntile = z_hi > z_lo ? (z_hi - z_lo + ZB - 1) / ZB : 0;
    if (ntile > 0) { zs.load(zsrc, zsrcT, ldT, z_lo, z_hi, YS); load_info(z_lo); }
    finish_xfrags<T, CT, IX>(xbase, xend, ROLE == ROLE_W, lane, xf);
    if (ntile > 0) {
        zs.store(reinterpret_cast<T*>(smem), reinterpret_cast<T*>(smem + S::Z_BYTES));
        store_info(smem, z_lo);
    }
    __syncthreads();
    for (int it = 0; it < ntile; ++it) {
        const int z0 = z_lo + it * ZB;
        const bool more = it + 1 < ntile;
        char* cur = smem + (DOUBLE ? (size_t)(it & 1) * BUF : 0);
        char* nxt = smem + (DOUBLE ? (size_t)((it + 1) & 1) * BUF : 0);
        if (PREFETCH && more) { zs.load_z(zsrc, z0 + ZB, z_hi, YS); load_info(z0 + ZB); }
        const T* Zs = reinterpret_cast<const T*>(cur);
        const T* ZTs = reinterpret_cast<const T*>(cur + S::Z_BYTES);
        const float* info = reinterpret_cast<const float*>(cur + S::Z_BYTES + S::ZT_BYTES);
        const bool edge = YS && ((z0 == 0) || (z0 + ZB > z_hi));
#pragma unroll
        for (int half = 0; half < S::NH; ++half) {
            f32x4 acc[JH][IX];
            logit_half<T, CT, IX>(Zs, half * JH, xf, lane, acc);
            // ---- dl[z][x] in place ------------------------------------------------------------------
            if constexpr (FLASH) {
                float tmax[IX];
#pragma unroll
                for (int ix = 0; ix < IX; ++ix) tmax[ix] = -INFINITY;
#pragma unroll
                for (int j = 0; j < JH; ++j) {
                    const int jz = half * JH + j;
                    const float4 t4 = *reinterpret_cast<const float4*>(info + jz * 16 + g4);
                    const float zb[4] = {t4.x, t4.y, t4.z, t4.w};
#pragma unroll
                    for (int ix = 0; ix < IX; ++ix)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            float x = acc[j][ix][r] + zb[r];
                            if (edge) {
                                const int gz = z0 + jz * 16 + g4 + r;
                                x = (gz == 0) ? -1000.0f : x;          // zero-padded row . y + (-1000)  (Base.py:110)
                                x = (gz < z_hi) ? x : -INFINITY;
                            }
                            acc[j][ix][r] = x;
                            tmax[ix] = fmaxf(tmax[ix], x);
                        }
                }
                bool moved = false;
                float corr[IX];
#pragma unroll
                for (int ix = 0; ix < IX; ++ix) {
                    const float m_new = fmaxf(m_run[ix], group_max4(tmax[ix]));
                    corr[ix] = (m_run[ix] > -INFINITY) ? __expf(m_run[ix] - m_new) : 0.f;
                    moved = moved || (m_new > m_run[ix]);
                    m_run[ix] = m_new;
                }
                if (__any(moved)) {     // rescale what was accumulated against the old maxima (wave-uniform branch)
#pragma unroll
                    for (int ix = 0; ix < IX; ++ix) {
                        s_run[ix] *= corr[ix];
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float f = __shfl(corr[ix], g4 + r, 64);   // the factor of row x = ix*16 + g4 + r sits in lane g4 + r
#pragma unroll
                            for (int ct = 0; ct < CO; ++ct) out[ix][ct][r] *= f;
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < JH; ++j)
#pragma unroll
                    for (int ix = 0; ix < IX; ++ix)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const float d = __expf(acc[j][ix][r] - m_run[ix]);
                            s_run[ix] += d;
                            acc[j][ix][r] = d;
                        }
            } else
#pragma unroll
            for (int j = 0; j < JH; ++j) {
                const int jz = half * JH + j;
                float zb[4];   // ROLE_Y: bias[z] ; ROLE_W: lse[z] - log coef[z]
                int zlab[4];
                {
                    const float4 t = *reinterpret_cast<const float4*>(info + jz * 16 + g4);
                    zb[0] = t.x; zb[1] = t.y; zb[2] = t.z; zb[3] = t.w;
                }
                if (ROLE == ROLE_W) {
                    const int4 c4 = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(info) + 2 * ZB + jz * 16 + g4);
                    zlab[0] = c4.x; zlab[1] = c4.y; zlab[2] = c4.z; zlab[3] = c4.w;
                }
#pragma unroll
                for (int ix = 0; ix < IX; ++ix)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int gz = z0 + jz * 16 + g4 + r;
                        float d;
                        if (ROLE == ROLE_Y) {
                            float x = acc[j][ix][r] + zb[r];
                            if (edge) x = (gz == 0) ? -1000.0f : x;
                            d = __expf(x - x_lse[ix]);
                            if (gz == x_lab[ix]) d -= x_cf[ix];
                            if (edge) d = (gz < z_hi) ? d : 0.f;
                        } else {
                            float x = acc[j][ix][r] + x_bias[ix];
                            x = (x_lab[ix] == 0) ? -1000.0f : x;
                            d = __expf(x - zb[r]);
                            if (x_lab[ix] == zlab[r]) d -= info[ZB + jz * 16 + g4 + r];   // rare: ~1 hit per tile
                            d = (x_lab[ix] >= 0) ? d : 0.f;     // x beyond the table shard
                            dbias[ix] += d;
                        }
                        acc[j][ix][r] = d;
                    }
            }
            // ---- out[x][c] += sum_z dl[x][z] Z[z][c] --------------------------------------------------
            if constexpr (sizeof(T) == 2) {
#pragma unroll
                for (int jp = 0; jp < JH / 2; ++jp) {
                    bf16x8 af[IX];
#pragma unroll
                    for (int ix = 0; ix < IX; ++ix)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            af[ix][r] = (bf16)acc[2 * jp][ix][r];
                            af[ix][4 + r] = (bf16)acc[2 * jp + 1][ix][r];
                        }
#pragma unroll
                    for (int ct = 0; ct < CO; ++ct) {
                        const bf16x8 bfr = sc_tr_frag32(Zs, S::LDC, (half * (JH / 2) + jp) * 32, (ct0 + ct) * 16, lane);
#pragma unroll
                        for (int ix = 0; ix < IX; ++ix) out[ix][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ix], bfr, out[ix][ct], 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
                for (int j = 0; j < JH; ++j) {
                    Frag4<T> a[IX];
#pragma unroll
                    for (int ix = 0; ix < IX; ++ix) a[ix] = frag_from_acc<T>(acc[j][ix]);
#pragma unroll
                    for (int ct = 0; ct < CO; ++ct) {
                        const Frag4<T> bfr = frag_ld<T>(ZTs + ((ct0 + ct) * 16 + l15) * S::LDZ + (half * JH + j) * 16 + g4);
#pragma unroll
                        for (int ix = 0; ix < IX; ++ix) out[ix][ct] = mma16(a[ix], bfr, out[ix][ct]);
                    }
                }
            }
        }
        if (!DOUBLE) __syncthreads();
        if (more) {
            if (!PREFETCH) { zs.load_z(zsrc, z0 + ZB, z_hi, YS); load_info(z0 + ZB); }
            zs.load_t(zsrcT, ldT, z0 + ZB, z_hi, YS);   // short-lived: the other wave of the SIMD covers it
            zs.store(reinterpret_cast<T*>(nxt), reinterpret_cast<T*>(nxt + S::Z_BYTES));
            store_info(nxt, z0 + ZB);
        }
        __syncthreads();
    }
    // ---- write the slab: out regs r <-> x = ix*16 + g4 + r, lane l15 <-> c = ct*16 + l15 ----------------
    float* slab = p.slabs + (long)by * slab_stride;
#pragma unroll
    for (int ix = 0; ix < IX; ++ix)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int gx = xbase + ix * 16 + g4 + r;
            if (gx < xend) {
#pragma unroll
                for (int ct = 0; ct < CO; ++ct) slab[(long)gx * S::C + (ct0 + ct) * 16 + l15] = out[ix][ct][r];
            }
        }
    if (FLASH && ct0 == 0) {   // (row max, row sum) of this item chunk: lse_combine_kernel / flash_finish_kernel merge the chunks
#pragma unroll
        for (int ix = 0; ix < IX; ++ix) {
            const float sm = group_sum4(s_run[ix]);
            const int gx = xbase + ix * 16 + l15;
            if (lane < 16 && gx < xend) {
                p.part[((long)gx * nchunk_dev + by) * 2] = m_run[ix];
                p.part[((long)gx * nchunk_dev + by) * 2 + 1] = sm;
            }
        }
    }
    if (ROLE == ROLE_W && ct0 == 0) {
#pragma unroll
        for (int ix = 0; ix < IX; ++ix) {
            const float v = group_sum4(dbias[ix]);
            const int gx = xbase + ix * 16 + l15;
            if (lane < 16 && gx < xend && gx > 0) p.bias_slabs[(long)by * (p.I - 1) + gx - 1] = v;
        }
    }
}

// out[i] = (T) sum_s slabs[s][i]; `zero_first` elements at the front are forced to 0 (table row 0).  A second job (the bias
// gradient next to the table gradient) rides in the same launch: workgroups [nb0, gridDim.x) belong to it.
struct SlabJob { const float* slabs; long n, lo, hi, zero_first; void* out; };
template <typename TO>
__global__ void slab_reduce_kernel(SlabJob j0, SlabJob j1, int nb0, int nslab, const float* gscale) {
    const float gs = gscale ? gscale[0] : 1.0f;   // upstream d(loss) scalar
    const bool first = (int)blockIdx.x < nb0;
    const SlabJob& j = first ? j0 : j1;
    const long bid = first ? blockIdx.x : blockIdx.x - nb0, nblk = first ? nb0 : (long)gridDim.x - nb0;
    TO* out = reinterpret_cast<TO*>(j.out);
    if constexpr (sizeof(TO) == 4) {
        // f32 output with everything a multiple of 4 (the table gradient): 16-byte vectors, the slabs of a group of four fetched
        // together (slab index clamped, the surplus weighted 0) — one element per thread and slab was a round trip per slab
        if (((j.lo | j.hi | j.n | j.zero_first) & 3) == 0 && ((uintptr_t)j.slabs & 15) == 0 && ((uintptr_t)out & 15) == 0) {
            for (long i = j.lo + (bid * blockDim.x + threadIdx.x) * 4; i < j.hi; i += nblk * blockDim.x * 4) {
                float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int s0 = 0; s0 < nslab; s0 += 4) {
                    float4 x[4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) x[k] = *reinterpret_cast<const float4*>(j.slabs + (long)min(s0 + k, nslab - 1) * j.n + i);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float w = s0 + k < nslab ? 1.f : 0.f;
                        a.x = s0 + k < nslab ? a.x + x[k].x : a.x; a.y = s0 + k < nslab ? a.y + x[k].y : a.y;
                        a.z = s0 + k < nslab ? a.z + x[k].z : a.z; a.w = s0 + k < nslab ? a.w + x[k].w : a.w;
                        (void)w;
                    }
                }
                const bool z = i < j.zero_first;
                *reinterpret_cast<float4*>(out + i) = z ? make_float4(0.f, 0.f, 0.f, 0.f) : make_float4(a.x * gs, a.y * gs, a.z * gs, a.w * gs);
            }
            return;
        }
    }
    for (long i = j.lo + bid * blockDim.x + threadIdx.x; i < j.hi; i += nblk * blockDim.x) {
        float a = 0.f;
        for (int s = 0; s < nslab; ++s) a += j.slabs[(long)s * j.n + i];
        out[i] = from_f32<TO>(i < j.zero_first ? 0.f : a * gs);
    }
}

// d_rows[r] = (T) gs * sum over the item chunks' slabs; the slab geometry follows dev_plan of the producing launch.
template <typename TO>
__global__ void slab_reduce_rows_kernel(const float* slabs, const int32_t* nvalid, int R, int C, int xb, int zb, int G, int ztotal,
                                        const float* gscale, TO* out) {
    const float gs = gscale ? gscale[0] : 1.0f;
    const int Reff = nvalid ? min(R, nvalid[0]) : R;
    const DevPlan dp = dev_plan(Reff, xb, G, ztotal, zb);
    const long stride = (long)dp.nx * xb * C, nval = (long)Reff * C, n = (long)R * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float a = 0.f;
        if (i < nval)
            for (int s = 0; s < dp.nchunk; ++s) a += slabs[(long)s * stride + i];
        out[i] = from_f32<TO>(a * gs);
    }
}

// four consecutive slab entries at element offset `off`: f32 slabs, or the bf16 slabs of the strip row pass (StripP::slab16)
template <bool S16>
__device__ __forceinline__ float4 ld_slab4(const float* slabs, long off) {
    if constexpr (S16) {
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16*>(slabs) + off);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    } else {
        return *reinterpret_cast<const float4*>(slabs + off);
    }
}

// lse_label_kernel + flash_finish_kernel in ONE launch (the training engine: both run back to back between the two product passes,
// each a few microseconds of work behind a kernel boundary).  LPR = C / 4 lanes own a row (4 consecutive channels each): every lane
// forms the row's log-sum-exp from the chunk partials itself (<= 16 exponentials), the label logit is the LPR-lane sum of the
// lanes' 4-channel dot products with the label row of the table — the row this kernel loads anyway for the -table[label] term —
// then coefficient and d_rows as in the two kernels.  Chunk counts above MAXCH take the rolled loops.
template <typename TO, int LPR, bool S16 = false>
__global__ __launch_bounds__(256) void flash_finish_lse_kernel(const float* slabs, const float* part, const TO* rows, const TO* table,
                                                               const float* out_bias, const int64_t* labels, const int32_t* nvalid,
                                                               const int32_t* wtotal, int R, int xb, int zb, int G, int ztotal,
                                                               const float* gscale, float* row_lse, float* lab_out, float* coef_out,
                                                               TO* out, float* ce_part) {
    constexpr int C = 4 * LPR, MAXCH = 16;
    __shared__ float ce_red[8];
    float ce_num = 0.f;       // this thread's rows: -log(p_label + 1e-5) of the weighted ones (EasyDGL.py:181-185), one lane per row
    const float gs = gscale ? gscale[0] : 1.0f;
    const int Reff = nvalid ? min(R, nvalid[0]) : R;
    const float W = (float)(wtotal ? wtotal[0] : Reff) + 1e-5f;       // EasyDGL.py:184 over the global batch
    const DevPlan dp = dev_plan(max(Reff, 1), xb, G, ztotal, zb);
    const long stride = (long)dp.nx * xb * C;
    const int nch = dp.nchunk;
    const int nrow_blk = 256 / LPR;
    for (long r0 = (long)blockIdx.x * nrow_blk; r0 < R; r0 += (long)gridDim.x * nrow_blk) {
        const int r = (int)min(r0 + threadIdx.x / LPR, (long)R - 1), c = (threadIdx.x % LPR) * 4, rc = min(r, max(Reff - 1, 0));
        const bool live = r0 + threadIdx.x / LPR < R;
        if (r0 >= Reff) {      // (uniform) rows behind the weighted ones — label 0 by construction of the compaction: nothing to load
            if (live) {
                if (c == 0) { row_lse[r] = 0.f; lab_out[r] = labels[r] == 0 ? -1000.0f : 0.f; coef_out[r] = 0.f; }
                if constexpr (sizeof(TO) == 2) *reinterpret_cast<uint2*>(out + (long)r * C + c) = make_uint2(0u, 0u);
                else *reinterpret_cast<float4*>(out + (long)r * C + c) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            continue;
        }
        int64_t lab = labels[r];
        asm volatile("" : "+v"(lab));
        const int64_t labc = lab > 0 ? lab : 0;
        float ob = out_bias[max(labc, (int64_t)1) - 1];
        float xr[4], tb[4];
        if constexpr (sizeof(TO) == 2) {
            const Frag4<TO> fx = frag_ld<TO>(rows + (long)r * C + c), ft = frag_ld<TO>(table + labc * C + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) { xr[q] = to_f32(fx.v[q]); tb[q] = to_f32(ft.v[q]); }
        } else {
            const float4 fx = *reinterpret_cast<const float4*>(rows + (long)r * C + c), ft = *reinterpret_cast<const float4*>(table + labc * C + c);
            xr[0] = fx.x; xr[1] = fx.y; xr[2] = fx.z; xr[3] = fx.w; tb[0] = ft.x; tb[1] = ft.y; tb[2] = ft.z; tb[3] = ft.w;
        }
        // ---- chunk partials in batches of MAXCH (all loads of a batch in flight together; index clamped, the surplus weighted 0):
        //      the row maximum first, then  sm = sum_c l_c e_c,  a = sum_c slab_c e_c  with e_c = exp(m_c - max);
        //      lse = max + log sm,  sum_c slab_c exp(m_c - lse) = a / sm
        float mx = -INFINITY;
        float4 sl0[MAXCH];                      // the first batch stays in registers (the benchmark has 11-12 chunks): ONE round trip
        float pm0[MAXCH], ps0[MAXCH];
#pragma unroll
        for (int s = 0; s < MAXCH; ++s) {
            const int sc = min(s, nch - 1);
            sl0[s] = ld_slab4<S16>(slabs, (long)sc * stride + (long)rc * C + c);
            pm0[s] = part[((long)rc * nch + sc) * 2];
            ps0[s] = part[((long)rc * nch + sc) * 2 + 1];
        }
#pragma unroll
        for (int s = 0; s < MAXCH; ++s) mx = s < nch ? fmaxf(mx, pm0[s]) : mx;
        for (int s0 = MAXCH; s0 < nch; s0 += MAXCH) {
            float pm[MAXCH];
#pragma unroll
            for (int s = 0; s < MAXCH; ++s) pm[s] = part[((long)rc * nch + min(s0 + s, nch - 1)) * 2];
#pragma unroll
            for (int s = 0; s < MAXCH; ++s) mx = s0 + s < nch ? fmaxf(mx, pm[s]) : mx;
        }
        float sm = 0.f, acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < MAXCH; ++s) {
            const float e = (s < nch && pm0[s] > -INFINITY) ? __expf(pm0[s] - mx) : 0.f;
            sm = fmaf(ps0[s], e, sm);
            acc[0] = fmaf(sl0[s].x, e, acc[0]); acc[1] = fmaf(sl0[s].y, e, acc[1]);
            acc[2] = fmaf(sl0[s].z, e, acc[2]); acc[3] = fmaf(sl0[s].w, e, acc[3]);
        }
        for (int s0 = MAXCH; s0 < nch; s0 += MAXCH) {
            float4 sl[MAXCH];
            float pm[MAXCH], ps[MAXCH];
#pragma unroll
            for (int s = 0; s < MAXCH; ++s) {
                const int sc = min(s0 + s, nch - 1);
                sl[s] = ld_slab4<S16>(slabs, (long)sc * stride + (long)rc * C + c);
                pm[s] = part[((long)rc * nch + sc) * 2];
                ps[s] = part[((long)rc * nch + sc) * 2 + 1];
            }
#pragma unroll
            for (int s = 0; s < MAXCH; ++s) {
                const float e = (s0 + s < nch && pm[s] > -INFINITY) ? __expf(pm[s] - mx) : 0.f;
                sm = fmaf(ps[s], e, sm);
                acc[0] = fmaf(sl[s].x, e, acc[0]); acc[1] = fmaf(sl[s].y, e, acc[1]);
                acc[2] = fmaf(sl[s].z, e, acc[2]); acc[3] = fmaf(sl[s].w, e, acc[3]);
            }
        }
        asm volatile("" : "+v"(ob));
        const float lse = r < Reff ? mx + __logf(sm) : 0.f;
        const float inv_sm = (r < Reff && sm > 0.f) ? 1.0f / sm : 0.f;
        // ---- label logit: LPR-lane sum (a row's lanes are LPR consecutive lanes of one wave)
        float a = (xr[0] * tb[0] + xr[1] * tb[1]) + (xr[2] * tb[2] + xr[3] * tb[3]);
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        const float ll = lab == 0 ? -1000.0f : a + ob;
        const float v = __expf(ll - lse);
        const float cf = (r < Reff && lab != 0) ? (1.f / W) * (v / (v + 1e-5f)) : 0.f;
        if (live && c == 0) {
            row_lse[r] = lse; lab_out[r] = ll; coef_out[r] = cf;
            ce_num += (r < Reff && lab != 0) ? -__logf(v + 1e-5f) : 0.f;
        }
        // ---- d_rows = gs * coef * ( sum_chunks slab_c * exp(m_c - lse)  -  table[label] )
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] *= inv_sm;
        float o4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) o4[q] = cf != 0.f ? gs * cf * (acc[q] - tb[q]) : 0.f;
        if (live) {
            if constexpr (sizeof(TO) == 2) { const Frag4<TO> f = frag_from_acc<TO>(f32x4{o4[0], o4[1], o4[2], o4[3]}); *reinterpret_cast<uint2*>(out + (long)r * C + c) = *reinterpret_cast<const uint2*>(&f); }
            else *reinterpret_cast<float4*>(out + (long)r * C + c) = make_float4(o4[0], o4[1], o4[2], o4[3]);
        }
    }
    if (ce_part) {      // the loss numerator of this workgroup's rows: the loss kernel then adds <= 4096 numbers instead of sweeping the rows
        ce_num = block_sum(ce_num, ce_red);
        if (threadIdx.x == 0) {
            ce_part[blockIdx.x] = ce_num;
            if (blockIdx.x == 0) ce_part[gridDim.x] = (float)Reff;      // ... and the weighted-row count behind the sums (exact below 2^24)
        }
    }
}

// The same launch at C = 512: eight channels per lane, so that a row's 64 lanes stay within one wave (the label logit is a wave
// reduction).  A kernel of its own: the instances above — the headline's among them — keep their code.
template <typename TO, int LPR, bool S16 = false, int CPL = 8>
__global__ __launch_bounds__(256) void flash_finish_lse_wide_kernel(const float* slabs, const float* part, const TO* rows, const TO* table,
                                                               const float* out_bias, const int64_t* labels, const int32_t* nvalid,
                                                               const int32_t* wtotal, int R, int xb, int zb, int G, int ztotal,
                                                               const float* gscale, float* row_lse, float* lab_out, float* coef_out,
                                                               TO* out, float* ce_part) {
    constexpr int C = CPL * LPR, MAXCH = CPL == 4 ? 16 : 8, NQ = CPL / 4;
    __shared__ float ce_red[8];
    float ce_num = 0.f;       // this thread's rows: -log(p_label + 1e-5) of the weighted ones (EasyDGL.py:181-185), one lane per row
    const float gs = gscale ? gscale[0] : 1.0f;
    const int Reff = nvalid ? min(R, nvalid[0]) : R;
    const float W = (float)(wtotal ? wtotal[0] : Reff) + 1e-5f;       // EasyDGL.py:184 over the global batch
    const DevPlan dp = dev_plan(max(Reff, 1), xb, G, ztotal, zb);
    const long stride = (long)dp.nx * xb * C;
    const int nch = dp.nchunk;
    const int nrow_blk = 256 / LPR;
    for (long r0 = (long)blockIdx.x * nrow_blk; r0 < R; r0 += (long)gridDim.x * nrow_blk) {
        const int r = (int)min(r0 + threadIdx.x / LPR, (long)R - 1), c = (threadIdx.x % LPR) * CPL, rc = min(r, max(Reff - 1, 0));
        const bool live = r0 + threadIdx.x / LPR < R;
        if (r0 >= Reff) {      // (uniform) rows behind the weighted ones — label 0 by construction of the compaction: nothing to load
            if (live) {
                if (c == 0) { row_lse[r] = 0.f; lab_out[r] = labels[r] == 0 ? -1000.0f : 0.f; coef_out[r] = 0.f; }
#pragma unroll
                for (int h = 0; h < NQ; ++h) {
                    if constexpr (sizeof(TO) == 2) *reinterpret_cast<uint2*>(out + (long)r * C + c + 4 * h) = make_uint2(0u, 0u);
                    else *reinterpret_cast<float4*>(out + (long)r * C + c + 4 * h) = make_float4(0.f, 0.f, 0.f, 0.f);
                }
            }
            continue;
        }
        int64_t lab = labels[r];
        asm volatile("" : "+v"(lab));
        const int64_t labc = lab > 0 ? lab : 0;
        float ob = out_bias[max(labc, (int64_t)1) - 1];
        float xr[CPL], tb[CPL];
#pragma unroll
        for (int h = 0; h < NQ; ++h) {
            if constexpr (sizeof(TO) == 2) {
                const Frag4<TO> fx = frag_ld<TO>(rows + (long)r * C + c + 4 * h), ft = frag_ld<TO>(table + labc * C + c + 4 * h);
#pragma unroll
                for (int q = 0; q < 4; ++q) { xr[4 * h + q] = to_f32(fx.v[q]); tb[4 * h + q] = to_f32(ft.v[q]); }
            } else {
                const float4 fx = *reinterpret_cast<const float4*>(rows + (long)r * C + c + 4 * h), ft = *reinterpret_cast<const float4*>(table + labc * C + c + 4 * h);
                xr[4 * h] = fx.x; xr[4 * h + 1] = fx.y; xr[4 * h + 2] = fx.z; xr[4 * h + 3] = fx.w;
                tb[4 * h] = ft.x; tb[4 * h + 1] = ft.y; tb[4 * h + 2] = ft.z; tb[4 * h + 3] = ft.w;
            }
        }
        // ---- chunk partials in batches of MAXCH (all loads of a batch in flight together; index clamped, the surplus weighted 0):
        //      the row maximum first, then  sm = sum_c l_c e_c,  a = sum_c slab_c e_c  with e_c = exp(m_c - max);
        //      lse = max + log sm,  sum_c slab_c exp(m_c - lse) = a / sm
        float mx = -INFINITY;
        float4 sl0[MAXCH][NQ];                  // the first batch stays in registers (the benchmark has 11-12 chunks): ONE round trip
        float pm0[MAXCH], ps0[MAXCH];
#pragma unroll
        for (int s = 0; s < MAXCH; ++s) {
            const int sc = min(s, nch - 1);
#pragma unroll
            for (int h = 0; h < NQ; ++h) sl0[s][h] = ld_slab4<S16>(slabs, (long)sc * stride + (long)rc * C + c + 4 * h);
            pm0[s] = part[((long)rc * nch + sc) * 2];
            ps0[s] = part[((long)rc * nch + sc) * 2 + 1];
        }
#pragma unroll
        for (int s = 0; s < MAXCH; ++s) mx = s < nch ? fmaxf(mx, pm0[s]) : mx;
        for (int s0 = MAXCH; s0 < nch; s0 += MAXCH) {
            float pm[MAXCH];
#pragma unroll
            for (int s = 0; s < MAXCH; ++s) pm[s] = part[((long)rc * nch + min(s0 + s, nch - 1)) * 2];
#pragma unroll
            for (int s = 0; s < MAXCH; ++s) mx = s0 + s < nch ? fmaxf(mx, pm[s]) : mx;
        }
        float sm = 0.f, acc[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) acc[q] = 0.f;
#pragma unroll
        for (int s = 0; s < MAXCH; ++s) {
            const float e = (s < nch && pm0[s] > -INFINITY) ? __expf(pm0[s] - mx) : 0.f;
            sm = fmaf(ps0[s], e, sm);
#pragma unroll
            for (int h = 0; h < NQ; ++h) {
                acc[4 * h] = fmaf(sl0[s][h].x, e, acc[4 * h]); acc[4 * h + 1] = fmaf(sl0[s][h].y, e, acc[4 * h + 1]);
                acc[4 * h + 2] = fmaf(sl0[s][h].z, e, acc[4 * h + 2]); acc[4 * h + 3] = fmaf(sl0[s][h].w, e, acc[4 * h + 3]);
            }
        }
        for (int s0 = MAXCH; s0 < nch; s0 += MAXCH) {
            float4 sl[MAXCH][NQ];
            float pm[MAXCH], ps[MAXCH];
#pragma unroll
            for (int s = 0; s < MAXCH; ++s) {
                const int sc = min(s0 + s, nch - 1);
#pragma unroll
                for (int h = 0; h < NQ; ++h) sl[s][h] = ld_slab4<S16>(slabs, (long)sc * stride + (long)rc * C + c + 4 * h);
                pm[s] = part[((long)rc * nch + sc) * 2];
                ps[s] = part[((long)rc * nch + sc) * 2 + 1];
            }
#pragma unroll
            for (int s = 0; s < MAXCH; ++s) {
                const float e = (s0 + s < nch && pm[s] > -INFINITY) ? __expf(pm[s] - mx) : 0.f;
                sm = fmaf(ps[s], e, sm);
#pragma unroll
                for (int h = 0; h < NQ; ++h) {
                    acc[4 * h] = fmaf(sl[s][h].x, e, acc[4 * h]); acc[4 * h + 1] = fmaf(sl[s][h].y, e, acc[4 * h + 1]);
                    acc[4 * h + 2] = fmaf(sl[s][h].z, e, acc[4 * h + 2]); acc[4 * h + 3] = fmaf(sl[s][h].w, e, acc[4 * h + 3]);
                }
            }
        }
        asm volatile("" : "+v"(ob));
        const float lse = r < Reff ? mx + __logf(sm) : 0.f;
        const float inv_sm = (r < Reff && sm > 0.f) ? 1.0f / sm : 0.f;
        // ---- label logit: LPR-lane sum (a row's lanes are LPR consecutive lanes of one wave)
        float a = (xr[0] * tb[0] + xr[1] * tb[1]) + (xr[2] * tb[2] + xr[3] * tb[3]);
        if constexpr (CPL == 8) a += (xr[4] * tb[4] + xr[5] * tb[5]) + (xr[6] * tb[6] + xr[7] * tb[7]);
#pragma unroll
        for (int o = LPR / 2; o > 0; o >>= 1) a += __shfl_xor(a, o, 64);
        const float ll = lab == 0 ? -1000.0f : a + ob;
        const float v = __expf(ll - lse);
        const float cf = (r < Reff && lab != 0) ? (1.f / W) * (v / (v + 1e-5f)) : 0.f;
        if (live && c == 0) {
            row_lse[r] = lse; lab_out[r] = ll; coef_out[r] = cf;
            ce_num += (r < Reff && lab != 0) ? -__logf(v + 1e-5f) : 0.f;
        }
        // ---- d_rows = gs * coef * ( sum_chunks slab_c * exp(m_c - lse)  -  table[label] )
#pragma unroll
        for (int q = 0; q < CPL; ++q) acc[q] *= inv_sm;
        float o4[CPL];
#pragma unroll
        for (int q = 0; q < CPL; ++q) o4[q] = cf != 0.f ? gs * cf * (acc[q] - tb[q]) : 0.f;
        if (live) {
#pragma unroll
            for (int h = 0; h < NQ; ++h) {
                if constexpr (sizeof(TO) == 2) { const Frag4<TO> f = frag_from_acc<TO>(f32x4{o4[4 * h], o4[4 * h + 1], o4[4 * h + 2], o4[4 * h + 3]}); *reinterpret_cast<uint2*>(out + (long)r * C + c + 4 * h) = *reinterpret_cast<const uint2*>(&f); }
                else *reinterpret_cast<float4*>(out + (long)r * C + c + 4 * h) = make_float4(o4[4 * h], o4[4 * h + 1], o4[4 * h + 2], o4[4 * h + 3]);
            }
        }
    }
    if (ce_part) {      // the loss numerator of this workgroup's rows: the loss kernel then adds <= 4096 numbers instead of sweeping the rows
        ce_num = block_sum(ce_num, ce_red);
        if (threadIdx.x == 0) {
            ce_part[blockIdx.x] = ce_num;
            if (blockIdx.x == 0) ce_part[gridDim.x] = (float)Reff;      // ... and the weighted-row count behind the sums (exact below 2^24)
        }
    }
}

// ROLE_YF finish: d_rows[r] = gs * coef[r] * ( sum_chunks slab_c[r] * exp(m_c - lse[r])  -  table[label[r]] )
//   = gs * coef * (sum_z p_z T_z - T_label)   (Appendix C: dy_rows = dl . table, dl = coef (p - onehot))
template <typename TO>
__global__ void flash_finish_kernel(const float* slabs, const float* part, const float* row_lse, const float* coef,
                                    const int64_t* labels, const TO* table, const int32_t* nvalid, int R, int C, int xb, int zb,
                                    int G, int ztotal, const float* gscale, TO* out, int i0, int i1) {
    // (i0, i1): the item range the slabs were formed over — the label row leaves only where the label lies in it (an item shard of
    //  the vocab-parallel loss: every rank finishes its share of d_rows with the GLOBAL log-sum-exp, the shares add up)
    const float gs = gscale ? gscale[0] : 1.0f;
    const int Reff = nvalid ? min(R, nvalid[0]) : R;
    const DevPlan dp = dev_plan(max(Reff, 1), xb, G, ztotal, zb);
    const long stride = (long)dp.nx * xb * C;
    const int nch = dp.nchunk, C4 = C >> 2;
    constexpr int MAXCH = 16;
    const bool vec = (((uintptr_t)slabs | (uintptr_t)out | (uintptr_t)table) & 15) == 0 && (stride & 3) == 0;
    // a thread = 4 consecutive channels of a row; rounds of loads: the row's scalars with the slab values and maxima of up to
    // MAXCH chunks at a time (chunk index clamped, the surplus weighted 0), then the label row of the table.  One element per
    // thread with the chunk loop rolled was a dependent round trip per chunk (the 512-unit recipe has > 16 chunks: 60 us).
    for (long i4 = (long)blockIdx.x * blockDim.x + threadIdx.x; i4 < (long)R * C4; i4 += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i4 / C4), c = (int)(i4 % C4) * 4, rc = min(r, max(Reff - 1, 0));
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (vec) {
            float cf = coef[rc], lse = row_lse[rc];
            int64_t lab = labels[rc];
            asm volatile("" : "+v"(cf), "+v"(lse), "+v"(lab));
            float tb[4];
            {
                const TO* trow = table + (lab > 0 ? lab : 0) * C + c;
                if constexpr (sizeof(TO) == 2) { const Frag4<TO> f = frag_ld<TO>(trow); for (int q = 0; q < 4; ++q) tb[q] = to_f32(f.v[q]); }
                else { const float4 f = *reinterpret_cast<const float4*>(trow); tb[0] = f.x; tb[1] = f.y; tb[2] = f.z; tb[3] = f.w; }
            }
            float a[4] = {0.f, 0.f, 0.f, 0.f};
            for (int s0 = 0; s0 < nch; s0 += MAXCH) {
                float4 sl[MAXCH];
                float pm[MAXCH];
#pragma unroll
                for (int s = 0; s < MAXCH; ++s) {
                    const int sc = min(s0 + s, nch - 1);
                    sl[s] = *reinterpret_cast<const float4*>(slabs + (long)sc * stride + (long)rc * C + c);
                    pm[s] = part[((long)rc * nch + sc) * 2];
                }
#pragma unroll
                for (int s = 0; s < MAXCH; ++s) {
                    const float e = s0 + s < nch ? __expf(pm[s] - lse) : 0.f;
                    const float x[4] = {sl[s].x, sl[s].y, sl[s].z, sl[s].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) a[q] = s0 + s < nch ? fmaf(x[q], e, a[q]) : a[q];
                }
            }
            const bool on = r < Reff && cf != 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = on ? gs * cf * (a[q] - ((lab > 0 && lab >= i0 && lab < i1) ? tb[q] : 0.f)) : 0.f;
        } else if (r < Reff) {
            const float cf = coef[r];
            if (cf != 0.f) {
                const float lse = row_lse[r];
                const int64_t lab = labels[r];
                for (int q = 0; q < 4; ++q) {
                    float a = 0.f;
                    for (int s = 0; s < nch; ++s) a += slabs[(long)s * stride + (long)r * C + c + q] * __expf(part[((long)r * nch + s) * 2] - lse);
                    v[q] = gs * cf * (a - ((lab > 0 && lab >= i0 && lab < i1) ? to_f32(table[lab * C + c + q]) : 0.f));
                }
            }
        }
        if constexpr (sizeof(TO) == 2) { const Frag4<TO> f = frag_from_acc<TO>(f32x4{v[0], v[1], v[2], v[3]}); *reinterpret_cast<uint2*>(out + (long)r * C + c) = *reinterpret_cast<const uint2*>(&f); }
        else *reinterpret_cast<float4*>(out + (long)r * C + c) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

// Rows whose label is 0 (a masked position that fell on padding) have weight 0 in the loss (EasyDGL.py:180): they
// contribute nothing to the loss or to any gradient.  compact_scan orders the weighted rows first
// (perm[j] = original row of compact row j, inv[r] = compact index of row r or -1, nvalid = #weighted rows), so the
// scoring kernels can skip the rest exactly.
__global__ __launch_bounds__(1024) void compact_scan_kernel(const int64_t* labels, int R, int32_t* perm, int32_t* inv,
                                                            int32_t* nvalid, int64_t* labels_c) {
    __shared__ int wcnt[2 * 16 * 16];
    batch_prep::compact_scan_body<1024>(labels, R, perm, inv, nvalid, labels_c, wcnt);
}
template <typename T>
__global__ void compact_gather_kernel(const T* rows, const int64_t* labels, const int32_t* perm, int R, int C, T* rows_c,
                                      int64_t* labels_c) {
    const int vpr = C / ElemTraits<T>::VEC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)R * vpr; i += (long)gridDim.x * blockDim.x) {
        const int j = (int)(i / vpr), cv = (int)(i % vpr);
        const int r = perm[j];
        st16<T>(rows_c + (long)j * C + cv * ElemTraits<T>::VEC, r >= 0 ? ld16<T>(rows + (long)r * C + cv * ElemTraits<T>::VEC) : zero16<T>());
        if (cv == 0) labels_c[j] = r >= 0 ? labels[r] : 0;
    }
}
template <typename T>
__global__ void scatter_rows_kernel(const T* rows_c, const int32_t* inv, int R, int C, T* rows) {
    const int vpr = C / ElemTraits<T>::VEC;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (long)R * vpr; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / vpr), cv = (int)(i % vpr);
        const int j = inv[r];
        st16<T>(rows + (long)r * C + cv * ElemTraits<T>::VEC, j >= 0 ? ld16<T>(rows_c + (long)j * C + cv * ElemTraits<T>::VEC) : zero16<T>());
    }
}

// [rows, C] -> [C, ld] transposed copy (operand images for the second MFMA product).  Two matrices per launch (the compacted
// rows and the item table): blocks [0, nbx0) of the x grid belong to the first.
template <typename T>
__global__ __launch_bounds__(256) void transpose_kernel(const T* src0, long rows0, T* dst0, long ld0, int nbx0, const T* src1, long rows1,
                                                        T* dst1, long ld1, int C) {
    __shared__ T tile[64][65];
    const bool first = (int)blockIdx.x < nbx0;
    const T* src = first ? src0 : src1;
    T* dst = first ? dst0 : dst1;
    const long rows = first ? rows0 : rows1, ld = first ? ld0 : ld1;
    const long r0 = (long)((int)blockIdx.x - (first ? 0 : nbx0)) * 64;
    const int c0 = blockIdx.y * 64;
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int r = i / 64, c = i % 64;
        tile[r][c] = (r0 + r < rows && c0 + c < C) ? src[(r0 + r) * C + c0 + c] : from_f32<T>(0.f);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 64; i += 256) {
        const int c = i / 64, r = i % 64;
        if (c0 + c < C && r0 + r < ld) dst[(long)(c0 + c) * ld + r0 + r] = tile[r][c];
    }
}

// ---------------------------------------------------------------------------------------------
// CE loss from (lse, label logit) — EasyDGL.py:155,177-185
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void ce_loss_kernel(const float* row_lse, const float* label_logit, const int64_t* labels, int R,
                                                       float* loss_out, float* coef, const float* add_in, const float* add_in2,
                                                       const int32_t* wtotal) {
    // one workgroup; a thread keeps up to CE_KEEP of its rows' probabilities in registers between the two passes (the loads of
    // a pass are independent, so they overlap instead of paying one memory round trip per row)
    constexpr int CE_KEEP = 16;
    __shared__ float red[16];
    float py[CE_KEEP];
    float num = 0.f, den = 0.f;
    {   // all 3 * CE_KEEP loads issued before the first use (`m < R && labels[..]` is a guarded load: a branch and a wait per row
        // — 16 dependent round trips, 10 us for this kernel)
        int64_t lb[CE_KEEP];
        float lg[CE_KEEP], ls[CE_KEEP];
#pragma unroll
        for (int i = 0; i < CE_KEEP; ++i) {
            const int mc = min((int)threadIdx.x + i * 1024, R - 1);
            lb[i] = labels[mc]; lg[i] = label_logit[mc]; ls[i] = row_lse[mc];
        }
#pragma unroll
        for (int i = 0; i < CE_KEEP; ++i) asm volatile("" : "+v"(lb[i]), "+v"(lg[i]), "+v"(ls[i]));
#pragma unroll
        for (int i = 0; i < CE_KEEP; ++i) {
            const int m = threadIdx.x + i * 1024;
            const bool on = (m < R) & (lb[i] != 0);   // weight 0 (EasyDGL.py:180): lse may be undefined for such rows
            const float v = __expf(lg[i] - ls[i]);
            py[i] = on ? v : -1.f;
            num += on ? -__logf(v + 1e-5f) : 0.f;
            den += on ? 1.f : 0.f;
        }
    }
    for (int m = threadIdx.x + CE_KEEP * 1024; m < R; m += 1024) {
        if (labels[m] == 0) continue;
        num += -__logf(__expf(label_logit[m] - row_lse[m]) + 1e-5f);
        den += 1.f;
    }
    num = block_sum(num, red);
    den = block_sum(den, red);
    const float W = (wtotal ? (float)wtotal[0] : den) + 1e-5f;     // data parallel: the weighted rows of all ranks
    if (threadIdx.x == 0) loss_out[0] = num / W + (add_in ? add_in[0] : 0.f) + (add_in2 ? add_in2[0] : 0.f);   // + regularisation terms
    if (!coef) return;    // (the flash forward wrote the coefficients: edgl_score_flash_fwd_coef)
#pragma unroll
    for (int i = 0; i < CE_KEEP; ++i) {
        const int m = threadIdx.x + i * 1024;
        if (m < R) coef[m] = py[i] >= 0.f ? (1.f / W) * (py[i] / (py[i] + 1e-5f)) : 0.f;
    }
    for (int m = threadIdx.x + CE_KEEP * 1024; m < R; m += 1024) {
        float cf = 0.f;
        if (labels[m] != 0) {
            const float v = __expf(label_logit[m] - row_lse[m]);
            cf = (1.f / W) * (v / (v + 1e-5f));
        }
        coef[m] = cf;
    }
}

// ce_loss_kernel for a forward that left per-workgroup sums of -log(p_label + 1e-5) (flash_finish_lse_kernel: ce_part): the loss is
// their sum over the weighted-row count (+ the regularisation terms) — a few thousand numbers instead of three arrays of R rows,
// and nothing of the batch (labels, lse, label logits) is read: the launch may run any time before the next forward rewrites ce_part.
__global__ __launch_bounds__(1024) void ce_loss_parts_kernel(const float* ce_part, int nparts, float* loss_out,
                                                             const float* add_in, const float* add_in2, const int32_t* wtotal) {
    __shared__ float red[16];
    float num = 0.f;
    for (int i0 = threadIdx.x; i0 < nparts; i0 += 1024 * 4) {
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ce_part[min(i0 + j * 1024, nparts - 1)];
#pragma unroll
        for (int j = 0; j < 4; ++j) num += i0 + j * 1024 < nparts ? v[j] : 0.f;
    }
    num = block_sum(num, red);
    if (threadIdx.x != 0) return;
    const float W = (wtotal ? (float)wtotal[0] : ce_part[nparts]) + 1e-5f;     // (the count rides behind the sums)
    loss_out[0] = num / W + (add_in ? add_in[0] : 0.f) + (add_in2 ? add_in2[0] : 0.f);
}

// ---------------------------------------------------------------------------------------------
// K6: mask seen items + per-row top-K (Base.py:156-163,181); K7 merge; metrics (Base.py:181-201)
// ---------------------------------------------------------------------------------------------
// one workgroup per row: 4-pass radix select of the K-th largest key, then ordered compaction (topk_select.h)
__global__ __launch_bounds__(256) void mask_topk_kernel(float* logits, int R, int n, int i0, const int64_t* seen,
                                                        int T, int K, float* out_val, int32_t* out_idx) {
    const int row = blockIdx.x, tid = threadIdx.x;
    float* x = logits + (long)row * n;
    if (seen) {
        for (int t = tid; t < T; t += blockDim.x) {
            const long id = seen[(long)row * T + t] - i0;
            if (id >= 0 && id < n) x[id] = -INFINITY;
        }
        __syncthreads();
    }
    radix_select_row(x, n, i0, K, out_val + (long)row * K, out_idx + (long)row * K);
}

// The same selection with the row in REGISTERS (n <= 256 * NJ): the 4-pass form above re-reads the row five times and builds its
// histograms with LDS atomics that pile up on a handful of bins (the logits of a row share their exponent bits) — 194 us for 512
// rows of 20 001 items, none of it memory time.  Here a thread keeps its NJ keys.
//   Fast path: the K-th largest of the 256 THREAD MAXIMA is a lower bound L of the K-th largest key (K distinct elements lie at or
//   above it), so the top K are among the elements >= L — a few hundred for any row without heavy ties.  They are compacted
//   into LDS and sorted there by (key descending, index ascending): no pass over all keys per bit, no tie logic.
//   Otherwise (more than TK_CAP candidates: tie blocks, degenerate rows) the K-th largest key is found bit by bit over ALL keys (32
//   rounds of "how many keys are >= prefix | bit": one compare per key, the wave total from the ballot's population count), then
//   the elements above it and the ties at it are gathered — ties in index order only when there are more of them than places left.
constexpr int TK_CAP = 1024;
template <int NJ>
__global__ __launch_bounds__(256) void mask_topk_reg_kernel(float* logits, int R, int n, int i0, const int64_t* seen,
                                                            int T, int K, float* out_val, int32_t* out_idx) {
    __shared__ int wcnt[2][4];
    __shared__ int cnt_gt, cnt_eq;
    __shared__ float cval[128];
    __shared__ int cidx[128];
    __shared__ int wave_eq[4];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float* x = logits + (long)row * n;
#ifndef TK_NOMASK
    if (seen) {
        for (int t = tid; t < T; t += 256) {
            const long id = seen[(long)row * T + t] - i0;
            if (id >= 0 && id < n) x[id] = -INFINITY;
        }
        __syncthreads();
    }
#endif
    // The row as aligned 16-byte pieces (a row of odd length starts anywhere): piece q = tid + 256 jq of the pieces from the aligned
    // address below the row's first element; register j = 4 jq + c holds element  i = 4 q + c - d  (d = the row's offset into its
    // first piece), or key 0 — below every float's key — outside the row.  (4-byte loads: 256 bytes per wave instruction, 1.3 TB/s.)
    static_assert(NJ % 4 == 0, "registers in groups of one piece");
    const long e0 = (long)row * n;
    const int d = (int)(e0 & 3);
    const float4* xa = reinterpret_cast<const float4*>(logits + (e0 - d));
    const int npiece = (n + d + 3) >> 2;
    auto elem = [&](int j) { return 4 * (tid + 256 * (j >> 2)) + (j & 3) - d; };
    uint32_t key[NJ];
#pragma unroll
    for (int jq = 0; jq < NJ / 4; ++jq) {
        const int q = tid + 256 * jq;
        const float4 v = xa[min(q, npiece - 1)];
        const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int i = 4 * q + c - d;
            key[4 * jq + c] = (i >= 0 && i < n) ? float_key(vv[c]) : 0u;
        }
    }
    const int Keff = min(K, n);
    // ---- fast path: candidates at or above the K-th largest thread maximum ----------------------------------------------------------
    __shared__ __attribute__((aligned(16))) uint32_t ckey[TK_CAP];
    __shared__ __attribute__((aligned(16))) int cix[TK_CAP];
    __shared__ int ccount;
    {
        uint32_t tmax = key[0];
#pragma unroll
        for (int j = 1; j < NJ; ++j) tmax = max(tmax, key[j]);
        // L = the Keff-th largest of the 256 maxima: ONE wave searches it bit by bit on four values per lane (ballots only, no
        // workgroup barrier per bit)
        __shared__ uint32_t tmx[256];
        __shared__ uint32_t Lsh;
        tmx[tid] = tmax;
        __syncthreads();
        if (w == 0) {
            const uint32_t m0 = tmx[lane], m1 = tmx[lane + 64], m2 = tmx[lane + 128], m3 = tmx[lane + 192];
            uint32_t Lw = 0u;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = Lw | (1u << bit);
                const int c = __popcll(__ballot(m0 >= cand)) + __popcll(__ballot(m1 >= cand)) + __popcll(__ballot(m2 >= cand)) +
                              __popcll(__ballot(m3 >= cand));
                if (c >= Keff) Lw = cand;
            }
            if (lane == 0) Lsh = Lw;
        }
        __syncthreads();
        const uint32_t L = Lsh;
        int c = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) c += __popcll(__ballot(key[j] >= L));
        if (tid == 0) ccount = 0;
        if (lane == 0) wcnt[0][w] = c;
        __syncthreads();
        const int C = wcnt[0][0] + wcnt[0][1] + wcnt[0][2] + wcnt[0][3];
#ifdef TK_ONLYLOAD
        if (C >= 0) { if (tid == 0) out_idx[(long)row * K] = C; return; }
#endif
        if (C <= TK_CAP) {
#pragma unroll
            for (int j = 0; j < NJ; ++j)
                if (key[j] >= L) {
                    const int pos = atomicAdd(&ccount, 1);
                    ckey[pos] = key[j]; cix[pos] = elem(j);
                }
            __syncthreads();
            if (C <= 256) {
                // few candidates (the usual case): every thread ranks its own candidate against all of them — its rank in (key
                // descending, index ascending) is its place in the output; no sort, no further barrier
                for (int i = tid; i < K; i += 256) { if (i >= Keff) { out_val[(long)row * K + i] = -INFINITY; out_idx[(long)row * K + i] = -1; } }
                // (the list is read four candidates per LDS instruction, two such groups in flight: one 32-bit read per candidate
                //  and iteration was a chain of ~C LDS round trips)
                if (tid < 4) { ckey[C + tid] = 0u; cix[C + tid] = 0x7fffffff; }      // (C <= 256 < TK_CAP - 4) padding: precedes nobody
                __syncthreads();
                if (tid < C) {
                    const uint32_t a = ckey[tid];
                    const int ia = cix[tid];
                    int rank = 0;
#pragma unroll 2
                    for (int q = 0; q < C; q += 4) {
                        const uint4 b = *reinterpret_cast<const uint4*>(ckey + q);
                        const int4 ib = *reinterpret_cast<const int4*>(cix + q);
                        rank += ((b.x > a) || (b.x == a && ib.x < ia)) ? 1 : 0;
                        rank += ((b.y > a) || (b.y == a && ib.y < ia)) ? 1 : 0;
                        rank += ((b.z > a) || (b.z == a && ib.z < ia)) ? 1 : 0;
                        rank += ((b.w > a) || (b.w == a && ib.w < ia)) ? 1 : 0;
                    }
                    if (rank < Keff) { out_val[(long)row * K + rank] = key_float(a); out_idx[(long)row * K + rank] = ia + i0; }
                }
                return;
            }
            int P2 = 512;
            while (P2 < C) P2 <<= 1;
            for (int t = C + tid; t < P2; t += 256) { ckey[t] = 0u; cix[t] = 0x7fffffff; }
            __syncthreads();
            for (int k = 2; k <= P2; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    for (int t = tid; t < P2; t += 256) {
                        const int ixj = t ^ j;
                        if (ixj > t) {
                            const uint32_t a = ckey[t], b = ckey[ixj];
                            const int ia = cix[t], ib = cix[ixj];
                            const bool a_first = (a > b) || (a == b && ia < ib);
                            const bool up = (t & k) == 0;
                            if (up ? !a_first : a_first) { ckey[t] = b; ckey[ixj] = a; cix[t] = ib; cix[ixj] = ia; }
                        }
                    }
                    __syncthreads();
                }
            for (int i = tid; i < K; i += 256) {
                out_val[(long)row * K + i] = i < Keff ? key_float(ckey[i]) : -INFINITY;
                out_idx[(long)row * K + i] = i < Keff ? cix[i] + i0 : -1;
            }
            return;
        }
        __syncthreads();
    }
    // ---- the K-th largest key: the largest P with |{key >= P}| >= K -------------------------------------------------------------
    uint32_t prefix = 0u;
#ifdef TK_NOSEARCH   // timing experiments only
    prefix = 0xfff00000u;
#else
    for (int bit = 31; bit >= 0; --bit) {
        const uint32_t cand = prefix | (1u << bit);
        int c = 0;
#pragma unroll
        for (int j = 0; j < NJ; ++j) c += __popcll(__ballot(key[j] >= cand));     // wave total (scalar unit)
        if (lane == 0) wcnt[bit & 1][w] = c;
        __syncthreads();                                                             // (two buffers: one barrier per round)
        const int tot = wcnt[bit & 1][0] + wcnt[bit & 1][1] + wcnt[bit & 1][2] + wcnt[bit & 1][3];
        if (tot >= Keff) prefix = cand;
    }
#endif
    // ---- gather: keys above the prefix (any order), ties at the prefix (the `remaining` lowest indices) ----------------------------
    if (tid == 0) { cnt_gt = 0; cnt_eq = 0; }
    if (tid < 128) { cval[tid] = -INFINITY; cidx[tid] = 0x7fffffff; }
    int c_gt = 0, c_eq = 0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) { c_gt += __popcll(__ballot(key[j] > prefix)); c_eq += __popcll(__ballot(key[j] == prefix)); }
    __syncthreads();
    if (lane == 0) { atomicAdd(&cnt_gt, c_gt); atomicAdd(&cnt_eq, c_eq); }
    __syncthreads();
    const int n_gt = cnt_gt, n_eq = cnt_eq, remaining = Keff - n_gt;
    __syncthreads();
    if (tid == 0) { cnt_gt = 0; cnt_eq = 0; }
    __syncthreads();
    const bool all_ties = n_eq <= remaining;     // (n_eq >= remaining by construction: == means every tie is taken)
#ifndef TK_NOGATHER
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int i = elem(j);
        if (key[j] > prefix) {
            const int pos = atomicAdd(&cnt_gt, 1);
            if (pos < 128) { cval[pos] = key_float(key[j]); cidx[pos] = i; }
        } else if (all_ties && key[j] == prefix && i >= 0 && i < n) {
            const int pos = n_gt + atomicAdd(&cnt_eq, 1);
            if (pos < 128) { cval[pos] = key_float(key[j]); cidx[pos] = i; }
        }
    }
#endif
    if (!all_ties) {      // more ties than places: index order — one pass per 1024 consecutive elements (a thread's piece = 4 of them)
#pragma unroll 1
        for (int jq = 0; jq < NJ / 4; ++jq) {
            uint32_t k4[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int q = 0; q < NJ / 4; ++q)
#pragma unroll
                for (int c = 0; c < 4; ++c) k4[c] = q == jq ? key[4 * q + c] : k4[c];
            const int ibase = 4 * (tid + 256 * jq) - d;
            bool eq[4];
            int below = 0, wtot = 0, own = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                eq[c] = k4[c] == prefix && ibase + c >= 0 && ibase + c < n;
                const unsigned long long bal = __ballot(eq[c]);
                below += __popcll(bal & ((1ull << lane) - 1ull));
                wtot += __popcll(bal);
            }
            if (lane == 0) wave_eq[w] = wtot;
            __syncthreads();
            int offs = cnt_eq + below;
            for (int ww = 0; ww < w; ++ww) offs += wave_eq[ww];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (eq[c]) {
                    const int pos = offs + own;
                    if (pos < remaining) { cval[n_gt + pos] = key_float(prefix); cidx[n_gt + pos] = ibase + c; }
                    ++own;
                }
            }
            __syncthreads();
            if (tid == 0) cnt_eq += wave_eq[0] + wave_eq[1] + wave_eq[2] + wave_eq[3];
            __syncthreads();
        }
    }
    __syncthreads();
#ifndef TK_NOSORT
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
#endif
    for (int i = tid; i < K; i += 256) {
        out_val[(long)row * K + i] = i < Keff ? cval[i] : -INFINITY;
        out_idx[(long)row * K + i] = i < Keff ? cidx[i] + i0 : -1;
    }
}

// candidates [S][R][K] -> global top-K by (value desc, index asc); S*K <= 1024
// counts (optional, [R]): the first counts[row] slots of a row's S*K (slot j = list j / K, place j % K) hold candidates, the rest
// is unwritten memory (the fused evaluation scoring appends candidates: k_eval_topk.hip); a row whose count exceeds S*K is left
// to that file's exact fallback kernel and not written here.
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* cand_val, const int32_t* cand_idx, int S, int R,
                                                         int K, float* out_val, int32_t* out_idx, const int32_t* counts) {
    __shared__ float v[1024];
    __shared__ int ix[1024];
    const int row = blockIdx.x, tid = threadIdx.x;
    int n = S * K;
    if (counts) {
        const int c = counts[row];
        if (c < 0 || c > n) return;
        n = c;
    }
    {   // the four slots of a thread: all eight loads out before the first use (slot index clamped: no load behind a branch)
        int id4[4]; float v4[4];
        const int SK = S * K;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = min(tid + 256 * j, SK - 1), s = i / K, k = i - s * K;
            id4[j] = cand_idx[((long)s * R + row) * K + k];
            v4[j] = cand_val[((long)s * R + row) * K + k];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(id4[j]), "+v"(v4[j]));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = tid + 256 * j;
            const bool ok = i < n && id4[j] >= 0;
            v[i] = ok ? v4[j] : -INFINITY;
            ix[i] = ok ? id4[j] : 0x7fffffff;
        }
    }
    // Fast path (any input order; as in mask_topk_reg_kernel): a thread looks at its <= 4 candidates, the K-th largest of the 256 thread
    // maxima is a lower bound of the K-th largest candidate, the candidates at or above it — a few hundred — are compacted and each
    // one's rank among them, in (value descending, index ascending), is its place in the output.  ~3 us instead of the 42 us of the
    // 1024-element bitonic sort below, which stays for inputs with more than 256 candidates above the bound (tie blocks).
    {
        __shared__ uint32_t tmx[256];
        __shared__ uint32_t Lsh;
        __shared__ __attribute__((aligned(16))) uint32_t ckey[264];
        __shared__ __attribute__((aligned(16))) int cix[264];
        __shared__ int ccount, wsum[4];
        const int lane = tid & 63, w = tid >> 6;
        const int Keff = min(K, n);
        uint32_t key[4];
        int kid[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = tid + 256 * q;      // (n <= 1024)
            const bool ok = i < n && ix[i] != 0x7fffffff;
            key[q] = ok ? float_key(v[i]) : 0u;
            kid[q] = ok ? ix[i] : 0x7fffffff;
        }
        tmx[tid] = max(max(key[0], key[1]), max(key[2], key[3]));
        if (tid == 0) ccount = 0;
        __syncthreads();
        if (w == 0) {
            const uint32_t m0 = tmx[lane], m1 = tmx[lane + 64], m2 = tmx[lane + 128], m3 = tmx[lane + 192];
            uint32_t Lw = 0u;
            for (int bit = 31; bit >= 0; --bit) {
                const uint32_t cand = Lw | (1u << bit);
                const int c = __popcll(__ballot(m0 >= cand)) + __popcll(__ballot(m1 >= cand)) + __popcll(__ballot(m2 >= cand)) +
                              __popcll(__ballot(m3 >= cand));
                if (c >= Keff) Lw = cand;
            }
            if (lane == 0) Lsh = Lw;
        }
        __syncthreads();
        const uint32_t L = max(Lsh, 1u);      // (key 0: invalid / absent candidates never qualify)
        int c = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) c += __popcll(__ballot(key[q] >= L));
        if (lane == 0) wsum[w] = c;
        __syncthreads();
        const int C = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (C <= 256) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (key[q] >= L) {
                    const int pos = atomicAdd(&ccount, 1);
                    ckey[pos] = key[q]; cix[pos] = kid[q];
                }
            for (int i = tid; i < K; i += blockDim.x) { out_val[(long)row * K + i] = -INFINITY; out_idx[(long)row * K + i] = -1; }
            __syncthreads();
            if (tid < 4) { ckey[C + tid] = 0u; cix[C + tid] = 0x7fffffff; }
            __syncthreads();
            if (tid < C) {
                const uint32_t a = ckey[tid];
                const int ia = cix[tid];
                int rank = 0;
#pragma unroll 2
                for (int q = 0; q < C; q += 4) {
                    const uint4 b = *reinterpret_cast<const uint4*>(ckey + q);
                    const int4 ib = *reinterpret_cast<const int4*>(cix + q);
                    // (a repeated (value, index) pair: the copy that sits first in the list goes first)
                    rank += ((b.x > a) || (b.x == a && (ib.x < ia || (ib.x == ia && q < tid)))) ? 1 : 0;
                    rank += ((b.y > a) || (b.y == a && (ib.y < ia || (ib.y == ia && q + 1 < tid)))) ? 1 : 0;
                    rank += ((b.z > a) || (b.z == a && (ib.z < ia || (ib.z == ia && q + 2 < tid)))) ? 1 : 0;
                    rank += ((b.w > a) || (b.w == a && (ib.w < ia || (ib.w == ia && q + 3 < tid)))) ? 1 : 0;
                }
                if (rank < K) { out_val[(long)row * K + rank] = key_float(a); out_idx[(long)row * K + rank] = ia; }
            }
            return;
        }
        __syncthreads();
    }
    for (int k = 2; k <= 1024; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < 1024; t += blockDim.x) {
                const int ixj = t ^ j;
                if (ixj > t) {
                    const float a = v[t], b = v[ixj];
                    const int ia = ix[t], ib = ix[ixj];
                    const bool a_first = (a > b) || (a == b && ia < ib);
                    const bool up = (t & k) == 0;
                    if (up ? !a_first : a_first) { v[t] = b; v[ixj] = a; ix[t] = ib; ix[ixj] = ia; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < K; i += blockDim.x) {
        out_val[(long)row * K + i] = v[i];
        out_idx[(long)row * K + i] = ix[i] == 0x7fffffff ? -1 : ix[i];
    }
}

__global__ void rank_metrics_kernel(const int32_t* topk, int R, int K, const int64_t* label, float* metrics) {
    __shared__ float red[8];
    float h10 = 0, h50 = 0, h100 = 0, n10 = 0, n50 = 0, n100 = 0;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const int lab = (int)label[r];
        int rank = -1;
        for (int k = 0; k < K && k < 100; ++k)
            if (topk[(long)r * K + k] == lab) { rank = k; break; }
        if (rank >= 0) {
            const float gain = 1.0f / log2f((float)rank + 2.0f);
            if (rank < 10) { h10 += 1; n10 += gain; }
            if (rank < 50) { h50 += 1; n50 += gain; }
            h100 += 1; n100 += gain;
        }
    }
    float vals[6] = {h10, h50, h100, n10, n50, n100};
    for (int i = 0; i < 6; ++i) {
        const float t = block_sum(vals[i], red);
        if (threadIdx.x == 0) metrics[i] += t;
    }
}


// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct Chunking { int nchunk, zchunk; };
// `l2_tiles` (0 = no limit): most z tiles one chunk may span.  Every workgroup of a chunk streams the same z range; the
// workgroups drift apart, and once that range no longer fits the 4 MB L2 of an XCD each of them re-reads it from HBM
// (measured at R = 51200 rows: d_rows kernel 917 us with one chunk of 157 item tiles, 474 us with 5 chunks of 32).
inline Chunking pick_chunks(int xblocks, int ztotal, int target_blocks, int ZB, int l2_tiles = 0) {
    const int ntiles = std::max(1, (ztotal + ZB - 1) / ZB);
    int nchunk = std::min(ntiles, std::max(1, target_blocks / std::max(1, xblocks)));
    if (l2_tiles > 0) nchunk = std::min(ntiles, std::max(nchunk, (ntiles + l2_tiles - 1) / l2_tiles));
    const int per = (ntiles + nchunk - 1) / nchunk;
    nchunk = (ntiles + per - 1) / per;
    return Chunking{nchunk, per * ZB};
}
inline int xblocks_of(int n, int xb) { return (n + xb - 1) / xb; }
// waves per workgroup of the generic product passes where the shape allows both: 8 (two waves per SIMD, bf16 C >= 256: half the
// output channels per pass, the logits recomputed per half) or 4 (one wave per SIMD, all channels in one pass).  Measured in the
// engine step (tools/try_shape.py, round 5): C = 256 — equal at 20 K items, 4 waves - 10 % at 100 K, - 13 % at 300 K / 1 M items
// (config 3: 53.0 -> 45.4 ms); C = 512 — 8 waves - 5..8 % (recipe 1.94 against 2.03 ms).  EDGL_SCORE_NW overrides.
inline int score_nw(int C, size_t esize) {
    static const int env = getenv("EDGL_SCORE_NW") ? atoi(getenv("EDGL_SCORE_NW")) : 0;
    if (env == 4 || env == 8) return env;
    return (C == 256 && esize == 2) ? 4 : 8;
}
inline int score_ftarget() { static const int t = getenv("EDGL_SCORE_FTARGET") ? atoi(getenv("EDGL_SCORE_FTARGET")) : 256; return t; }
inline int score_target() { static const int t = getenv("EDGL_SCORE_TARGET") ? atoi(getenv("EDGL_SCORE_TARGET")) : 256; return t; }
inline long up8(long v) { return (v + 7) / 8 * 8; }

// z tiles per chunk that keep a chunk's streamed range within ~2 MB (half an XCD's L2); `images` = 2 when the kernel reads the
// tile and its transposed copy (d_rows), 1 for the forward
inline int l2_tiles_for(int C, size_t esize, int images, int ZB) { return std::max(8, (int)((2u << 20) / ((size_t)ZB * C * esize * images))); }
constexpr int F_L2_TILES_MIN = 32;   // smallest value l2_tiles_for(C, esize, 1, zb) takes over the supported (C, dtype): workspace bound
struct BwdPlan {
    Chunking y, w;
    long off_rowsT, off_tableT, off_slabY, off_slabW, off_slabB, off_part, off_infoY, off_infoW, total;  // float offsets
};
// The flash form (edgl_score_flash_*) at bf16 / C = 128 runs the strip kernels (k_score_strip.hip): 256 x vectors per
// workgroup as well, 64-z tiles handed out in pairs.
// 0: the generic kernels, 1: k_score_strip.hip (C = 128, 256 x vectors per workgroup), 2: k_score_stripw.hip (C = 256, 128 per workgroup)
inline int use_strip(int C, size_t esize) {
    if (esize != 2) return 0;
    if (C == 128 && edgl_strip_enabled()) return 1;
    if (edgl_stripw_supports(C) && edgl_stripw_enabled()) return 2;
    return 0;
}
inline int strip_xb(int strip) { return strip == 2 ? 128 : 256; }
inline BwdPlan bwd_plan(int R, int C, int I, int n_items, size_t esize, int strip = 0) {
    BwdPlan b;
    const RtCfg cf = rt_cfg(C, esize);
    const int nw = strip ? 8 : (cf.nwb == 8 ? score_nw(C, esize) : cf.nwb), zb = strip ? 128 : cf.zb;   // strip: pairs of its 64-z tiles
    const int xb = strip ? strip_xb(strip) : 16 * cf.ix * nw;
    b.y = pick_chunks(xblocks_of(R, xb), n_items, score_target(), zb, l2_tiles_for(C, esize, 2, zb));
    {   // every chunk writes an [R, C] f32 slab that a later kernel sums: keep that side traffic bounded (1M-item tables would
        // otherwise ask for hundreds of chunks)
        const long slab_bytes = (long)xblocks_of(R, xb) * xb * C * 4;
        const int cap = (int)std::max<long>(score_target() / std::max(1, xblocks_of(R, xb)), (256L << 20) / std::max(1L, slab_bytes));
        if (b.y.nchunk > cap) b.y = pick_chunks(xblocks_of(R, xb), n_items, cap * xblocks_of(R, xb), zb);
    }
    b.w = pick_chunks(xblocks_of(n_items, xb), R, score_target(), zb);
    long o = 0;
    auto take = [&](long floats) { const long at = o; o += (floats + 63) / 64 * 64; return at; };
    b.off_rowsT = take(((long)C * up8(R) * (long)esize + 3) / 4);
    b.off_tableT = take(((long)C * up8(I) * (long)esize + 3) / 4);
    // G workgroups x one [xb, C] tile each (strip: the grid is at least score_target() workgroups, split on the device)
    const long gy = strip ? std::max<long>((long)b.y.nchunk * xblocks_of(R, xb), score_target()) : (long)b.y.nchunk * xblocks_of(R, xb);
    b.off_slabY = take(gy * xb * C);
    b.off_slabW = take((long)b.w.nchunk * I * C);
    b.off_slabB = take((long)b.w.nchunk * (I - 1));
    b.off_part = take(2L * gy * xb);   // (max, sum) per (row, item chunk) of the ROLE_YF pass
    b.off_infoY = b.off_infoW = 0;
    if (strip == 2) {                  // k_score_stripw.hip: the C operands of the z rows of a pass (items / rows), -inf padded
        b.off_infoY = take(edgl_stripw_info_floats(n_items));
        b.off_infoW = take(edgl_stripw_info_floats(R));
    }
    b.total = o;
    return b;
}

template <typename T, int CT>
int run_fwd(ScoreP p, hipStream_t st) {
    using S = SC<T, CT>;
    constexpr int XB = 16 * ScoreCfg<T, CT>::IX * (SNT / 64);
    const size_t smem = 2 * (S::Z_BYTES + S::ZB * sizeof(float));
    static_assert(2 * (S::Z_BYTES + S::ZB * sizeof(float)) <= 160 * 1024, "forward tile images exceed the LDS");
    auto k = score_fwd_kernel<T, CT>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, dim3(xblocks_of(p.R, XB) * p.nchunk), dim3(SNT), smem, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

// MODE 0: edgl_score_ce_bwd (transposes, d_rows with the known lse, d_table / d_bias)
// MODE 1: edgl_score_flash_fwd (transposes, ROLE_YF pass: unnormalised d_rows slabs + per-chunk (max, sum) -> row lse)
// MODE 2: edgl_score_flash_bwd (finish d_rows from the slabs of MODE 1, then d_table / d_bias)
// Format of the d_rows slabs a MODE 1 call left in a flash workspace (host-side record keyed by the workspace address: the
// calls of one step are issued by one host thread in order, and a captured graph replays the checked sequence).  The strip row
// pass writes bf16 slabs when the rows are finished in the same call; flash_finish_kernel (MODE 2 with d_rows) reads f32 slabs —
// pairing the two is refused instead of returning garbage row gradients.
static std::mutex g_slabfmt_mu;
static std::unordered_map<const void*, int> g_slabfmt;
static void slab_format_set(const void* ws, int bf16_slabs) {
    std::lock_guard<std::mutex> lk(g_slabfmt_mu);
    g_slabfmt[ws] = bf16_slabs;
}
static int slab_format_get(const void* ws) {
    std::lock_guard<std::mutex> lk(g_slabfmt_mu);
    auto it = g_slabfmt.find(ws);
    return it == g_slabfmt.end() ? 0 : it->second;
}

template <typename T, int CT, int MODE>
int run_bwd_mode(ScoreP p, const BwdPlan& plan, float* ws, void* d_rows, float* d_table, float* d_bias, hipStream_t st) {
    using S = SC<T, CT>;
    constexpr size_t BUF = S::Z_BYTES + S::ZT_BYTES + S::INFO_BYTES;
    const size_t smem = (2 * BUF <= 160 * 1024) ? 2 * BUF : BUF;
    EDGL_REQUIRE(smem <= 160 * 1024, EDGL_ERR_SHAPE, "edgl_score_ce_bwd: C=%d needs %zu B of LDS", p.C, smem);
    // operand images for the second product: rowsT [C][ldr], tableT [C][ldt]
    T* rowsT = reinterpret_cast<T*>(ws + plan.off_rowsT);
    T* tableT = reinterpret_cast<T*>(ws + plan.off_tableT);
    p.ldr = (int)up8(p.R); p.ldt = (int)up8(p.I);
    if (MODE != 2 && S::WT) {   // bf16 reads the transposed operand out of the row-major LDS tile: no transposed copies
        const int nbx0 = (p.ldr + 63) / 64, nbx1 = p.table_ready ? 0 : (p.ldt + 63) / 64;
        hipLaunchKernelGGL((transpose_kernel<T>), dim3((unsigned)(nbx0 + nbx1), (p.C + 63) / 64), dim3(256), 0, st,
                           reinterpret_cast<const T*>(p.rows), (long)p.R, rowsT, (long)p.ldr, nbx0,
                           reinterpret_cast<const T*>(p.table), (long)p.I, tableT, (long)p.ldt, p.C);
        EDGL_LAUNCH_CHECK();
    }
    p.rowsT = rowsT; p.tableT = tableT;
    using Cfg = ScoreCfg<T, CT>;
    constexpr int CO = Cfg::CO;
    const int strip = MODE != 0 ? use_strip(p.C, sizeof(T)) : 0;     // plan built with the same flag by the callers
    const int ZBK = strip ? 128 : S::ZB;
    const int nw = Cfg::NWB == 8 ? score_nw(p.C, sizeof(T)) : Cfg::NWB;
    const size_t smem_nw = nw == 8 ? smem : BUF;
    const int xb = strip ? strip_xb(strip) : 16 * Cfg::IX * nw;
    const int G = strip ? std::max(xblocks_of(p.R, xb) * plan.y.nchunk, score_target()) : xblocks_of(p.R, xb) * plan.y.nchunk;
    float* part = ws + plan.off_part;
    // rows finished in one launch (flash_finish_lse_kernel): the strip row pass then leaves bf16 slabs
    const bool one_launch = MODE == 1 && d_rows && (p.C == 128 || p.C == 64 || p.C == 256 || p.C == 512) && p.i0 == 0 && p.i1 == p.I;
    const bool slab16 = strip == 1 && one_launch;
    if (MODE == 1) slab_format_set(ws, slab16 ? 1 : 0);
    if (MODE == 2 && d_rows)
        EDGL_REQUIRE(slab_format_get(ws) == 0, EDGL_ERR_WORKSPACE,
                     "edgl_score_flash_bwd: d_rows != NULL, but this workspace holds the bf16 slabs of edgl_score_flash_fwd_rows_w "
                     "(which has already written d_rows): call with d_rows = NULL, or run edgl_score_flash_fwd_coef for the two-call form");
    if (MODE != 2) {   // the row-side pass
        constexpr int RY = MODE == 1 ? ROLE_YF : ROLE_Y;
        ScoreP q = p;
        q.zchunk = plan.y.zchunk; q.nchunk = plan.y.nchunk; q.slabs = ws + plan.off_slabY; q.part = part;
        edgl_prof_begin(EDGL_KERNEL_SCORE_BWD_ROWS, st);
        if (strip == 2) {
            const int rc = edgl_stripw_rows(p.rows, p.table, p.out_bias, p.R, p.C, p.I, p.i0, p.i1, p.nvalid, q.slabs, part, G, ws + plan.off_infoY, st);
            if (rc) return rc;
        } else if (strip) {
            const int rc = edgl_strip_rows(p.rows, p.table, p.out_bias, p.R, p.I, p.i0, p.i1, p.nvalid, q.slabs, part, G, slab16 ? 1 : 0, st);
            if (rc) return rc;
        } else if (nw == 8) {
            auto k = score_bwd_kernel<T, CT, RY, 8, CO>;
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_nw);
            hipLaunchKernelGGL(k, dim3(G, CT / CO), dim3(512), smem_nw, st, q);
        } else {
            auto k = score_bwd_kernel<T, CT, RY, 4>;
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_nw);
            hipLaunchKernelGGL(k, dim3(G), dim3(256), smem_nw, st, q);
        }
        edgl_prof_end(EDGL_KERNEL_SCORE_BWD_ROWS, st);
        EDGL_LAUNCH_CHECK();
    }
    const long nrc = (long)p.R * p.C;
    if (MODE == 0) {
        hipLaunchKernelGGL((slab_reduce_rows_kernel<T>), dim3((unsigned)std::min<long>((nrc + 255) / 256, 2048)), dim3(256), 0, st,
                           ws + plan.off_slabY, p.nvalid, p.R, p.C, xb, ZBK, G, p.i1 - p.i0, p.gscale, reinterpret_cast<T*>(d_rows));
        EDGL_LAUNCH_CHECK();
    } else if (MODE == 1) {
        if (one_launch) {
            // rows finished in one launch: log-sum-exp, label logit, coefficient and d_rows (edgl_score_flash_fwd_rows_w)
            const T* rows_t = reinterpret_cast<const T*>(p.rows);
            const T* tab_t = reinterpret_cast<const T*>(p.table);
            const float* slabY = ws + plan.off_slabY;
            T* out_t = reinterpret_cast<T*>(d_rows);
#define EDGL_FFL(LPR, S16)                                                                                                     \
    hipLaunchKernelGGL((flash_finish_lse_kernel<T, LPR, S16>), dim3((unsigned)std::min<long>(((long)p.R * LPR + 255) / 256, 4096)), \
                       dim3(256), 0, st, slabY, part, rows_t, tab_t, p.out_bias, p.labels, p.nvalid, p.wtotal, p.R, xb, ZBK, G, \
                       p.i1 - p.i0, p.gscale, p.row_lse, p.lab_out, p.coef_out, out_t, p.ce_part)
            if (p.C == 128 && slab16) EDGL_FFL(32, true); else if (p.C == 128) EDGL_FFL(32, false); else if (p.C == 64) EDGL_FFL(16, false); else if (p.C == 512) hipLaunchKernelGGL((flash_finish_lse_wide_kernel<T, 64, false, 8>), dim3((unsigned)std::min<long>(((long)p.R * 64 + 255) / 256, 4096)), dim3(256), 0, st, slabY, part, rows_t, tab_t, p.out_bias, p.labels, p.nvalid, p.wtotal, p.R, xb, ZBK, G, p.i1 - p.i0, p.gscale, p.row_lse, p.lab_out, p.coef_out, out_t, p.ce_part); else EDGL_FFL(64, false);
#undef EDGL_FFL
            EDGL_LAUNCH_CHECK();
            return EDGL_OK;
        }
        hipLaunchKernelGGL((lse_label_kernel<T>), dim3((p.R + 3) / 4), dim3(256), 0, st, part, p.R, p.nvalid, xb, ZBK, G, p.i1 - p.i0,
                           p.row_lse, reinterpret_cast<const T*>(p.rows), reinterpret_cast<const T*>(p.table), p.out_bias, p.labels,
                           p.C, p.i0, p.i1, p.lab_out, p.coef_out, p.wtotal);
        EDGL_LAUNCH_CHECK();
        if (d_rows) {   // other widths: the two kernels, one after the other
            hipLaunchKernelGGL((flash_finish_kernel<T>), dim3((unsigned)std::min<long>((nrc / 4 + 255) / 256, 2048)), dim3(256), 0, st,
                               ws + plan.off_slabY, part, p.row_lse, p.coef_out, p.labels, reinterpret_cast<const T*>(p.table), p.nvalid,
                               p.R, p.C, xb, ZBK, G, p.i1 - p.i0, p.gscale, reinterpret_cast<T*>(d_rows), p.i0, p.i1);
            EDGL_LAUNCH_CHECK();
        }
        return EDGL_OK;
    } else if (d_rows) {
        hipLaunchKernelGGL((flash_finish_kernel<T>), dim3((unsigned)std::min<long>((nrc / 4 + 255) / 256, 2048)), dim3(256), 0, st,
                           ws + plan.off_slabY, part, p.row_lse, p.coef, p.labels, reinterpret_cast<const T*>(p.table), p.nvalid,
                           p.R, p.C, xb, ZBK, G, p.i1 - p.i0, p.gscale, reinterpret_cast<T*>(d_rows), p.i0, p.i1);
        EDGL_LAUNCH_CHECK();
    }
    // d_table, d_bias
    {
        ScoreP q = p;
        q.zchunk = plan.w.zchunk; q.nchunk = plan.w.nchunk; q.slabs = ws + plan.off_slabW; q.bias_slabs = ws + plan.off_slabB;
        if (strip == 2) {
            const int rc = edgl_stripw_table(p.rows, p.table, p.out_bias, p.coef, p.row_lse, p.R, p.C, p.I, p.i0, p.i1, p.nvalid, q.slabs,
                                             q.bias_slabs, q.nchunk, ws + plan.off_infoW, st);
            if (rc) return rc;
        } else if (strip) {
            const bool acc = p.acc_atomic && !p.gscale;
            const int rc = edgl_strip_table(p.rows, p.table, p.out_bias, p.coef, p.row_lse, p.R, p.I, p.i0, p.i1, p.nvalid, q.slabs,
                                            q.bias_slabs, q.nchunk, acc ? d_table : nullptr, acc ? d_bias : nullptr, st);
            if (rc) return rc;
            if (acc) {      // no slabs, no slab reduction
                if (!p.defer_label) return edgl_strip_label_scatter(p.rows, p.labels, p.coef, p.nvalid, p.R, p.i0, p.i1, p.gscale, d_table, d_bias, st);
                return EDGL_OK;
            }
        } else if (nw == 8) {
            auto k = score_bwd_kernel<T, CT, ROLE_W, 8, CO>;
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_nw);
            hipLaunchKernelGGL(k, dim3(xblocks_of(p.i1 - p.i0, xb), q.nchunk, CT / CO), dim3(512), smem_nw, st, q);
        } else {
            auto k = score_bwd_kernel<T, CT, ROLE_W, 4>;
            hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_nw);
            hipLaunchKernelGGL(k, dim3(xblocks_of(p.i1 - p.i0, xb), q.nchunk), dim3(256), smem_nw, st, q);
        }
        EDGL_LAUNCH_CHECK();
        if (p.keep_slabs && !p.gscale) return EDGL_OK;      // (the caller asked edgl_score_flash_slab_info where they are; the one-hot term is deferred too)
        const long n = (long)p.I * p.C, lo = (long)p.i0 * p.C, hi = (long)p.i1 * p.C;
        const long nb = p.I - 1, blo = std::max(p.i0, 1) - 1, bhi = p.i1 - 1;
        const int nb0 = (int)std::min<long>(((hi - lo) / 4 + 255) / 256 + 1, 2048), nb1 = bhi > blo ? (int)std::min<long>((bhi - blo + 255) / 256, 256) : 0;
        hipLaunchKernelGGL((slab_reduce_kernel<float>), dim3((unsigned)(nb0 + nb1)), dim3(256), 0, st,
                           SlabJob{q.slabs, n, lo, hi, (long)p.C, d_table}, SlabJob{q.bias_slabs, nb, blo, bhi, 0L, d_bias}, nb0,
                           q.nchunk, p.gscale);
        EDGL_LAUNCH_CHECK();
        if (strip && !p.defer_label) {   // the one-hot part of dl, which the strip product pass leaves out
            const int rc = strip == 2 ? edgl_stripw_label_scatter(p.rows, p.labels, p.coef, p.nvalid, p.R, p.C, p.i0, p.i1, p.gscale, d_table, d_bias, st)
                                      : edgl_strip_label_scatter(p.rows, p.labels, p.coef, p.nvalid, p.R, p.i0, p.i1, p.gscale, d_table, d_bias, st);
            if (rc) return rc;
        }
    }
    return EDGL_OK;
}

int check_score(const void* rows, const void* table, const float* out_bias, int R, int C, int I, int i0, int i1,
                int dtype, const char* who) {
    EDGL_REQUIRE(rows && table && out_bias, EDGL_ERR_NULL, "%s: null pointer", who);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "%s: bad dtype %d", who, dtype);
    EDGL_REQUIRE(R > 0 && I > 1 && i0 >= 0 && i1 <= I && i0 < i1 && (i0 % 8) == 0, EDGL_ERR_SHAPE,
                 "%s: bad shape R=%d I=%d [%d,%d) (i0 must be a multiple of 8)", who, R, I, i0, i1);
    EDGL_REQUIRE(((uintptr_t)rows & 15) == 0 && ((uintptr_t)table & 15) == 0, EDGL_ERR_SHAPE,
                 "%s: operands must be 16-byte aligned", who);
    EDGL_REQUIRE(C >= 32 && C <= 512 && (C & (C - 1)) == 0, EDGL_ERR_SHAPE, "%s: C=%d unsupported (power of two in [32, 512])",
                 who, C);
    return EDGL_OK;
}

#define SCORE_DISPATCH(T, CALL)                                      \
    switch (C / 16) {                                                \
        case 2: return CALL(T, 2);                                   \
        case 4: return CALL(T, 4);                                   \
        case 8: return CALL(T, 8);                                   \
        case 16: return CALL(T, 16);                                 \
        case 32: return CALL(T, 32);                                 \
    }                                                                \
    edgl_set_error("edgl_score: C=%d unsupported", C);               \
    return EDGL_ERR_SHAPE;

template <typename T>
int fwd_dispatch(ScoreP p, int C, hipStream_t st) {
#define CALL_FWD(T, CT) run_fwd<T, CT>(p, st)
    SCORE_DISPATCH(T, CALL_FWD)
#undef CALL_FWD
}
template <typename T, int MODE>
int bwd_dispatch(ScoreP p, int C, const BwdPlan& plan, float* ws, void* d_rows, float* d_table, float* d_bias, hipStream_t st) {
#define CALL_BWD(T, CT) run_bwd_mode<T, CT, MODE>(p, plan, ws, d_rows, d_table, d_bias, st)
    SCORE_DISPATCH(T, CALL_BWD)
#undef CALL_BWD
}

}  // namespace

extern "C" int edgl_score_flash_fwd_coef_w(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R,
                                           int C, int I, const int32_t* nvalid, const int32_t* wtotal, float* row_lse,
                                           float* label_logit, float* coef, float* workspace, int dtype, void* stream);

// workspace rule of the forward: 2 * R * edgl_score_chunks floats >= 2 * XB * (#workgroups), the most (row, chunk)
// partial pairs any device-side split of the launch can produce
extern "C" int edgl_score_chunks(int R, int n_items) {
    long best = 1;   // C is not an argument: take the largest need over the tile shapes of ScoreCfg
    for (int xb : {256, 128})
        for (int zb : {128, 64, 32}) {
            const long g = (long)xblocks_of(R, xb) * pick_chunks(xblocks_of(R, xb), n_items, score_ftarget(), zb, F_L2_TILES_MIN).nchunk;
            best = std::max(best, (g * xb + R - 1) / R);
        }
    return (int)best;
}

// edgl_compact_rows = edgl_compact_scan (labels only: may run as soon as the batch is known, e.g. on a side stream at the start
// of the step) + edgl_compact_gather (needs the rows)
extern "C" int edgl_compact_scan(const int64_t* labels, int R, int32_t* perm, int32_t* inv, int32_t* nvalid, void* stream) {
    EDGL_REQUIRE(labels && perm && inv && nvalid, EDGL_ERR_NULL, "edgl_compact_scan: null pointer");
    EDGL_REQUIRE(R > 0, EDGL_ERR_SHAPE, "edgl_compact_scan: bad shape R=%d", R);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, R, perm, inv, nvalid, (int64_t*)nullptr);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
// The scan that also writes the compacted labels (labels_c [R]: the labels of the weighted rows first, 0 behind them) — with
// it and edgl_tail_fwd's row map the rows reach the scoring kernels without edgl_compact_gather.
extern "C" int edgl_compact_scan_labels(const int64_t* labels, int R, int32_t* perm, int32_t* inv, int32_t* nvalid,
                                        int64_t* labels_c, void* stream) {
    EDGL_REQUIRE(labels && perm && inv && nvalid && labels_c, EDGL_ERR_NULL, "edgl_compact_scan_labels: null pointer");
    EDGL_REQUIRE(R > 0, EDGL_ERR_SHAPE, "edgl_compact_scan_labels: bad shape R=%d", R);
    hipLaunchKernelGGL(compact_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, labels, R, perm, inv, nvalid, labels_c);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
extern "C" int edgl_compact_gather(const void* rows, const int64_t* labels, const int32_t* perm, int R, int C, void* rows_c,
                                   int64_t* labels_c, int dtype, void* stream) {
    EDGL_REQUIRE(rows && labels && perm && rows_c && labels_c, EDGL_ERR_NULL, "edgl_compact_gather: null pointer");
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_compact_gather: bad dtype %d", dtype);
    EDGL_REQUIRE(R > 0 && C % (dtype == EDGL_BF16 ? 8 : 4) == 0, EDGL_ERR_SHAPE, "edgl_compact_gather: bad shape R=%d C=%d", R, C);
    hipStream_t st = (hipStream_t)stream;
    const long nv = (long)R * C / (dtype == EDGL_BF16 ? 8 : 4);
    const unsigned nb = (unsigned)std::min<long>((nv + 255) / 256, 2048);
    if (dtype == EDGL_F32) hipLaunchKernelGGL((compact_gather_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)rows, labels, perm, R, C, (float*)rows_c, labels_c);
    else hipLaunchKernelGGL((compact_gather_kernel<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)rows, labels, perm, R, C, (bf16*)rows_c, labels_c);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
extern "C" int edgl_compact_rows(const void* rows, const int64_t* labels, int R, int C, int32_t* perm, int32_t* inv,
                                 int32_t* nvalid, void* rows_c, int64_t* labels_c, int dtype, void* stream) {
    EDGL_REQUIRE(rows && labels && perm && inv && nvalid && rows_c && labels_c, EDGL_ERR_NULL, "edgl_compact_rows: null pointer");
    const int rc = edgl_compact_scan(labels, R, perm, inv, nvalid, stream);
    return rc ? rc : edgl_compact_gather(rows, labels, perm, R, C, rows_c, labels_c, dtype, stream);
}

extern "C" int edgl_scatter_rows(const void* rows_c, const int32_t* inv, int R, int C, void* rows, int dtype, void* stream) {
    EDGL_REQUIRE(rows_c && inv && rows, EDGL_ERR_NULL, "edgl_scatter_rows: null pointer");
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_scatter_rows: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const long nv = (long)R * C / (dtype == EDGL_BF16 ? 8 : 4);
    const unsigned nb = (unsigned)std::min<long>((nv + 255) / 256, 2048);
    if (dtype == EDGL_F32) hipLaunchKernelGGL((scatter_rows_kernel<float>), dim3(nb), dim3(256), 0, st, (const float*)rows_c, inv, R, C, (float*)rows);
    else hipLaunchKernelGGL((scatter_rows_kernel<bf16>), dim3(nb), dim3(256), 0, st, (const bf16*)rows_c, inv, R, C, (bf16*)rows);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_score_lse_fwd(const void* rows, const void* table, const float* out_bias, const int64_t* labels,
                                  int R, int C, int I, int i0, int i1, const int32_t* nvalid, float* row_lse,
                                  float* label_logit, float* logits, float* workspace, int dtype, void* stream) {
    int rc = check_score(rows, table, out_bias, R, C, I, i0, i1, dtype, "edgl_score_lse_fwd");
    if (rc) return rc;
    // row_lse == NULL (with logits): the logits tile only — the evaluation path (score_topk) has no use for the normaliser, and the
    // one-thread-per-row merge of the chunk partials is a chain of dependent loads (46 us per tile at 512 rows)
    EDGL_REQUIRE((row_lse || logits) && workspace && (!labels || label_logit), EDGL_ERR_NULL, "edgl_score_lse_fwd: null output");
    ScoreP p{};
    p.rows = rows; p.table = table; p.out_bias = out_bias; p.labels = labels; p.R = R; p.C = C; p.I = I; p.i0 = i0;
    p.i1 = i1; p.nvalid = nvalid; p.row_lse = row_lse; p.part = workspace; p.logits = logits;
    const size_t esize = dtype == EDGL_BF16 ? 2 : 4;
    const RtCfg cf = rt_cfg(C, esize);
    const int xbf = 16 * cf.ix * (SNT / 64);
    const Chunking ch = pick_chunks(xblocks_of(R, xbf), i1 - i0, score_ftarget(), cf.zb, l2_tiles_for(C, esize, 1, cf.zb));
    p.nchunk = ch.nchunk; p.zchunk = ch.zchunk;
    hipStream_t st = (hipStream_t)stream;
    rc = dtype == EDGL_F32 ? fwd_dispatch<float>(p, C, st) : fwd_dispatch<bf16>(p, C, st);
    if (rc) return rc;
    if (row_lse) {
        hipLaunchKernelGGL(lse_combine_kernel, dim3((R + 255) / 256), dim3(256), 0, st, workspace, R, nvalid, xbf, cf.zb,
                           xblocks_of(R, xbf) * p.nchunk, i1 - i0, row_lse);
        EDGL_LAUNCH_CHECK();
    }
    if (labels) {
        if (dtype == EDGL_F32)
            hipLaunchKernelGGL((label_logit_kernel<float>), dim3((R + 3) / 4), dim3(256), 0, st, (const float*)rows, (const float*)table, out_bias, labels, R, C, i0, i1, label_logit);
        else
            hipLaunchKernelGGL((label_logit_kernel<bf16>), dim3((R + 3) / 4), dim3(256), 0, st, (const bf16*)rows, (const bf16*)table, out_bias, labels, R, C, i0, i1, label_logit);
        EDGL_LAUNCH_CHECK();
    }
    return EDGL_OK;
}

extern "C" int edgl_ce_loss_fwd_add_w(const float* row_lse, const float* label_logit, const int64_t* labels, int R, float* loss_out,
                                      float* coef, const float* add_in, const float* add_in2, const int32_t* wtotal, void* stream) {
    EDGL_REQUIRE(row_lse && label_logit && labels && loss_out, EDGL_ERR_NULL, "edgl_ce_loss_fwd: null pointer");   // coef may be NULL
    hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_lse, label_logit, labels, R, loss_out, coef,
                       add_in, add_in2, wtotal);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
// number of per-workgroup loss sums edgl_score_flash_fwd_rows_wp writes (0: this shape has no one-launch row finish)
extern "C" int edgl_score_ce_nparts(int R, int C) {
    if (!(C == 128 || C == 64 || C == 256 || C == 512) || R <= 0) return 0;
    const long lpr = C == 512 ? 64 : C / 4;      // (C = 512: eight channels per lane)
    return (int)std::min<long>(((long)R * lpr + 255) / 256, 4096);
}
// loss (EasyDGL.py:181-188) from the sums edgl_score_flash_fwd_rows_wp left in ce_part: sum / (weighted rows + 1e-5) + add_in + add_in2
// (ce_part: edgl_score_ce_nparts(R, C) sums and, behind them, the weighted-row count — nparts + 1 floats; wtotal: the global count
// under data parallelism)
extern "C" int edgl_ce_loss_parts(const float* ce_part, int nparts, float* loss_out, const float* add_in, const float* add_in2,
                                  const int32_t* wtotal, void* stream) {
    EDGL_REQUIRE(ce_part && loss_out && nparts > 0, EDGL_ERR_NULL, "edgl_ce_loss_parts: null pointer / bad size");
    hipLaunchKernelGGL(ce_loss_parts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ce_part, nparts, loss_out, add_in, add_in2,
                       wtotal);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
extern "C" int edgl_ce_loss_fwd_add(const float* row_lse, const float* label_logit, const int64_t* labels, int R, float* loss_out,
                                    float* coef, const float* add_in, const float* add_in2, void* stream) {
    return edgl_ce_loss_fwd_add_w(row_lse, label_logit, labels, R, loss_out, coef, add_in, add_in2, nullptr, stream);
}
extern "C" int edgl_ce_loss_fwd(const float* row_lse, const float* label_logit, const int64_t* labels, int R,
                                float* loss_out, float* coef, void* stream) {
    return edgl_ce_loss_fwd_add(row_lse, label_logit, labels, R, loss_out, coef, nullptr, nullptr, stream);
}

extern "C" long edgl_score_bwd_workspace(int R, int C, int I, int n_items, int dtype) {
    return bwd_plan(R, C, I, n_items, dtype == EDGL_BF16 ? 2 : 4).total;
}

extern "C" int edgl_score_ce_bwd(const void* rows, const void* table, const float* out_bias, const int64_t* labels,
                                 const float* row_lse, const float* coef, const float* gscale, int R, int C, int I,
                                 int i0, int i1, const int32_t* nvalid, void* d_rows, float* d_table, float* d_bias,
                                 float* workspace, int dtype, void* stream) {
    int rc = check_score(rows, table, out_bias, R, C, I, i0, i1, dtype, "edgl_score_ce_bwd");
    if (rc) return rc;
    EDGL_REQUIRE(labels && row_lse && coef && d_rows && d_table && d_bias && workspace, EDGL_ERR_NULL,
                 "edgl_score_ce_bwd: null pointer");
    ScoreP p{};
    p.rows = rows; p.table = table; p.out_bias = out_bias; p.labels = labels; p.R = R; p.C = C; p.I = I; p.i0 = i0;
    p.i1 = i1; p.nvalid = nvalid; p.row_lse = const_cast<float*>(row_lse); p.coef = coef; p.gscale = gscale;
    const BwdPlan plan = bwd_plan(R, C, I, i1 - i0, dtype == EDGL_BF16 ? 2 : 4);
    hipStream_t st = (hipStream_t)stream;
    return dtype == EDGL_F32 ? bwd_dispatch<float, 0>(p, C, plan, workspace, d_rows, d_table, d_bias, st)
                             : bwd_dispatch<bf16, 0>(p, C, plan, workspace, d_rows, d_table, d_bias, st);
}

// ---- "flash" form: the forward scoring pass already accumulates the row gradients -------------------------------------------
// edgl_score_flash_fwd: ONE pass over the item table computes, per weighted row, the log-sum-exp of its logits (as
// edgl_score_lse_fwd) AND sum_z exp(logit_z - max) table[z] — the row gradient up to the loss coefficient — with flash-style
// running maxima; the label logits come from the same small gather kernel.  edgl_score_flash_bwd then finishes d_rows
// (divide by the row sum, apply coef, subtract the label row) and computes d_table / d_bias as edgl_score_ce_bwd does.
// The logits are computed 2x per step (here and in the d_table pass) instead of 3x.  `workspace` (edgl_score_flash_workspace
// floats) carries the slabs from the forward to the backward call and must not be touched in between.
extern "C" long edgl_score_flash_workspace(int R, int C, int I, int n_items, int dtype) {
    return bwd_plan(R, C, I, n_items, dtype == EDGL_BF16 ? 2 : 4, use_strip(C, dtype == EDGL_BF16 ? 2 : 4)).total;
}

extern "C" int edgl_score_flash_fwd(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R,
                                    int C, int I, int i0, int i1, const int32_t* nvalid, float* row_lse, float* label_logit,
                                    float* workspace, int dtype, void* stream) {
    int rc = check_score(rows, table, out_bias, R, C, I, i0, i1, dtype, "edgl_score_flash_fwd");
    if (rc) return rc;
    EDGL_REQUIRE(labels && row_lse && label_logit && workspace, EDGL_ERR_NULL, "edgl_score_flash_fwd: null pointer");
    ScoreP p{};
    p.rows = rows; p.table = table; p.out_bias = out_bias; p.labels = labels; p.R = R; p.C = C; p.I = I; p.i0 = i0;
    p.i1 = i1; p.nvalid = nvalid; p.row_lse = row_lse;
    const BwdPlan plan = bwd_plan(R, C, I, i1 - i0, dtype == EDGL_BF16 ? 2 : 4, use_strip(C, dtype == EDGL_BF16 ? 2 : 4));
    hipStream_t st = (hipStream_t)stream;
    p.lab_out = label_logit;   // row LSE and label logits come out of one small kernel behind the scoring pass
    return dtype == EDGL_F32 ? bwd_dispatch<float, 1>(p, C, plan, workspace, nullptr, nullptr, nullptr, st)
                             : bwd_dispatch<bf16, 1>(p, C, plan, workspace, nullptr, nullptr, nullptr, st);
}

// The transposed image of the item table depends on the weights only: edgl_score_prepare_table writes it into the flash
// workspace ahead of time (e.g. on a side stream at the start of the step) and edgl_score_flash_fwd_pre then transposes the
// rows alone.  Same R, C, I, [i0, i1) and workspace as the forward call that follows.
extern "C" int edgl_score_prepare_table(const void* table, int R, int C, int I, int i0, int i1, float* workspace, int dtype,
                                        void* stream) {
    EDGL_REQUIRE(table && workspace, EDGL_ERR_NULL, "edgl_score_prepare_table: null pointer");
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_score_prepare_table: bad dtype %d", dtype);
    const BwdPlan plan = bwd_plan(R, C, I, i1 - i0, dtype == EDGL_BF16 ? 2 : 4);
    const long ldt = (long)up8(I);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((unsigned)((ldt + 63) / 64), (C + 63) / 64);
    if (dtype != EDGL_F32) return EDGL_OK;      // the bf16 kernels need no transposed image
    hipLaunchKernelGGL((transpose_kernel<float>), grid, dim3(256), 0, st, (const float*)table, (long)I, reinterpret_cast<float*>(workspace + plan.off_tableT), ldt,
                       (int)grid.x, (const float*)nullptr, 0L, (float*)nullptr, 0L, C);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
extern "C" int edgl_score_flash_fwd_pre(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R,
                                        int C, int I, int i0, int i1, const int32_t* nvalid, float* row_lse, float* label_logit,
                                        float* workspace, int table_ready, int dtype, void* stream) {
    int rc = check_score(rows, table, out_bias, R, C, I, i0, i1, dtype, "edgl_score_flash_fwd");
    if (rc) return rc;
    EDGL_REQUIRE(labels && row_lse && label_logit && workspace, EDGL_ERR_NULL, "edgl_score_flash_fwd: null pointer");
    ScoreP p{};
    p.rows = rows; p.table = table; p.out_bias = out_bias; p.labels = labels; p.R = R; p.C = C; p.I = I; p.i0 = i0;
    p.i1 = i1; p.nvalid = nvalid; p.row_lse = row_lse; p.lab_out = label_logit; p.table_ready = table_ready;
    const BwdPlan plan = bwd_plan(R, C, I, i1 - i0, dtype == EDGL_BF16 ? 2 : 4, use_strip(C, dtype == EDGL_BF16 ? 2 : 4));
    hipStream_t st = (hipStream_t)stream;
    return dtype == EDGL_F32 ? bwd_dispatch<float, 1>(p, C, plan, workspace, nullptr, nullptr, nullptr, st)
                             : bwd_dispatch<bf16, 1>(p, C, plan, workspace, nullptr, nullptr, nullptr, st);
}
// edgl_score_flash_fwd over the whole table for COMPACTED rows (edgl_compact_*: the *nvalid weighted rows first, labels 0 behind
// them) that also writes the loss coefficients coef[r] = (1 / (n + 1e-5)) * p_y / (p_y + 1e-5), n = *nvalid — what edgl_ce_loss_fwd
// computes after a reduction over the rows; with them the backward does not wait for the loss kernel (EasyDGL.py:177-185).
extern "C" int edgl_score_flash_fwd_coef(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R,
                                         int C, int I, const int32_t* nvalid, float* row_lse, float* label_logit, float* coef,
                                         float* workspace, int dtype, void* stream) {
    return edgl_score_flash_fwd_coef_w(rows, table, out_bias, labels, R, C, I, nvalid, nullptr, row_lse, label_logit, coef, workspace,
                                       dtype, stream);
}
// Data parallel form: `wtotal` (device, may be NULL) = weighted rows of the GLOBAL batch; the coefficients then are those of the
// global loss (EasyDGL.py:183-185 with the denominator summed over the ranks), and the ranks' gradients ADD UP to its gradient.
extern "C" int edgl_score_flash_fwd_coef_w(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R,
                                           int C, int I, const int32_t* nvalid, const int32_t* wtotal, float* row_lse,
                                           float* label_logit, float* coef, float* workspace, int dtype, void* stream) {
    int rc = check_score(rows, table, out_bias, R, C, I, 0, I, dtype, "edgl_score_flash_fwd_coef");
    if (rc) return rc;
    EDGL_REQUIRE(labels && row_lse && label_logit && workspace && nvalid && coef, EDGL_ERR_NULL,
                 "edgl_score_flash_fwd_coef: null pointer (the row count of the compaction is required)");
    ScoreP p{};
    p.rows = rows; p.table = table; p.out_bias = out_bias; p.labels = labels; p.R = R; p.C = C; p.I = I; p.i0 = 0;
    p.i1 = I; p.nvalid = nvalid; p.row_lse = row_lse; p.lab_out = label_logit; p.coef_out = coef; p.wtotal = wtotal;
    const BwdPlan plan = bwd_plan(R, C, I, I, dtype == EDGL_BF16 ? 2 : 4, use_strip(C, dtype == EDGL_BF16 ? 2 : 4));
    hipStream_t st = (hipStream_t)stream;
    return dtype == EDGL_F32 ? bwd_dispatch<float, 1>(p, C, plan, workspace, nullptr, nullptr, nullptr, st)
                             : bwd_dispatch<bf16, 1>(p, C, plan, workspace, nullptr, nullptr, nullptr, st);
}

// edgl_score_flash_fwd_coef_w that also finishes the rows: d_rows (= gscale * d loss / d rows, what edgl_score_flash_bwd would write)
// comes out of the same launch as lse / label logits / coefficients; edgl_score_flash_bwd is then called with d_rows = NULL and
// only runs the table side.
extern "C" int edgl_score_flash_fwd_rows_wp(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R,
                                            int C, int I, const int32_t* nvalid, const int32_t* wtotal, const float* gscale,
                                            float* row_lse, float* label_logit, float* coef, void* d_rows, float* ce_part,
                                            float* workspace, int dtype, void* stream);
extern "C" int edgl_score_flash_fwd_rows_w(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R,
                                           int C, int I, const int32_t* nvalid, const int32_t* wtotal, const float* gscale,
                                           float* row_lse, float* label_logit, float* coef, void* d_rows, float* workspace,
                                           int dtype, void* stream) {
    return edgl_score_flash_fwd_rows_wp(rows, table, out_bias, labels, R, C, I, nvalid, wtotal, gscale, row_lse, label_logit, coef, d_rows,
                                        nullptr, workspace, dtype, stream);
}
// ... that also leaves the loss numerator as edgl_score_ce_nparts(R, C) per-workgroup sums in ce_part (NULL: none), for
// edgl_ce_loss_parts — the loss launch then reads nothing of the batch
extern "C" int edgl_score_flash_fwd_rows_wp(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R,
                                            int C, int I, const int32_t* nvalid, const int32_t* wtotal, const float* gscale,
                                            float* row_lse, float* label_logit, float* coef, void* d_rows, float* ce_part,
                                            float* workspace, int dtype, void* stream) {
    int rc = check_score(rows, table, out_bias, R, C, I, 0, I, dtype, "edgl_score_flash_fwd_rows");
    if (rc) return rc;
    EDGL_REQUIRE(!ce_part || edgl_score_ce_nparts(R, C) > 0, EDGL_ERR_SHAPE, "edgl_score_flash_fwd_rows_wp: no per-workgroup loss sums at C=%d", C);
    EDGL_REQUIRE(labels && row_lse && label_logit && workspace && nvalid && coef && d_rows, EDGL_ERR_NULL,
                 "edgl_score_flash_fwd_rows: null pointer (the row count of the compaction is required)");
    ScoreP p{};
    p.rows = rows; p.table = table; p.out_bias = out_bias; p.labels = labels; p.R = R; p.C = C; p.I = I; p.i0 = 0;
    p.i1 = I; p.nvalid = nvalid; p.row_lse = row_lse; p.lab_out = label_logit; p.coef_out = coef; p.wtotal = wtotal; p.gscale = gscale;
    p.ce_part = ce_part;
    const BwdPlan plan = bwd_plan(R, C, I, I, dtype == EDGL_BF16 ? 2 : 4, use_strip(C, dtype == EDGL_BF16 ? 2 : 4));
    hipStream_t st = (hipStream_t)stream;
    return dtype == EDGL_F32 ? bwd_dispatch<float, 1>(p, C, plan, workspace, d_rows, nullptr, nullptr, st)
                             : bwd_dispatch<bf16, 1>(p, C, plan, workspace, d_rows, nullptr, nullptr, st);
}

extern "C" int edgl_score_flash_bwd_ex(const void* rows, const void* table, const float* out_bias, const int64_t* labels,
                                       const float* row_lse, const float* coef, const float* gscale, int R, int C, int I, int i0,
                                       int i1, const int32_t* nvalid, void* d_rows, float* d_table, float* d_bias,
                                       float* workspace, int defer_label_term, int dtype, void* stream);
extern "C" int edgl_score_flash_bwd(const void* rows, const void* table, const float* out_bias, const int64_t* labels,
                                    const float* row_lse, const float* coef, const float* gscale, int R, int C, int I, int i0,
                                    int i1, const int32_t* nvalid, void* d_rows, float* d_table, float* d_bias,
                                    float* workspace, int dtype, void* stream) {
    return edgl_score_flash_bwd_ex(rows, table, out_bias, labels, row_lse, coef, gscale, R, C, I, i0, i1, nvalid, d_rows, d_table, d_bias,
                                   workspace, 0, dtype, stream);
}
// The one-hot part of dl = coef (p - onehot(label)) as its own call:  d_table[label[r]] -= gscale coef[r] rows[r],
// d_bias[label[r] - 1] -= gscale coef[r]  over the weighted rows (f32 atomics, commutes with every other accumulation into the two
// arrays).  edgl_score_flash_bwd_ex(defer_label_term = 1) leaves exactly this out WHERE ITS PRODUCT PASS DOES NOT CONTAIN IT (bf16,
// C = 128: the strip kernels) — edgl_score_flash_label_term is then a launch, and a no-op (return 0) for every other configuration,
// whose product kernels subtract the one-hot in place.  The training engine runs it on its side stream under the embedding backward.
extern "C" int edgl_score_flash_label_term(const void* rows, const int64_t* labels, const float* coef, const float* gscale, int R, int C,
                                           int I, int i0, int i1, const int32_t* nvalid, float* d_table, float* d_bias, int dtype,
                                           void* stream) {
    EDGL_REQUIRE(rows && labels && coef && d_table && d_bias, EDGL_ERR_NULL, "edgl_score_flash_label_term: null pointer");
    EDGL_REQUIRE(R > 0 && I > 1 && i0 >= 0 && i1 <= I && i0 < i1, EDGL_ERR_SHAPE, "edgl_score_flash_label_term: bad shape");
    const int strip = use_strip(C, dtype == EDGL_BF16 ? 2 : 4);
    if (!strip) return EDGL_OK;
    if (strip == 2) return edgl_stripw_label_scatter(rows, labels, coef, nvalid, R, C, i0, i1, gscale, d_table, d_bias, (hipStream_t)stream);
    return edgl_strip_label_scatter(rows, labels, coef, nvalid, R, i0, i1, gscale, d_table, d_bias, (hipStream_t)stream);
}
extern "C" int edgl_score_flash_bwd_ex(const void* rows, const void* table, const float* out_bias, const int64_t* labels,
                                       const float* row_lse, const float* coef, const float* gscale, int R, int C, int I, int i0,
                                       int i1, const int32_t* nvalid, void* d_rows, float* d_table, float* d_bias,
                                       float* workspace, int defer_label_term, int dtype, void* stream) {
    int rc = check_score(rows, table, out_bias, R, C, I, i0, i1, dtype, "edgl_score_flash_bwd");
    if (rc) return rc;
    // d_rows == NULL: the rows were finished by edgl_score_flash_fwd_rows_w; only d_table / d_bias are computed
    EDGL_REQUIRE(labels && row_lse && coef && d_table && d_bias && workspace, EDGL_ERR_NULL,
                 "edgl_score_flash_bwd: null pointer");
    ScoreP p{};
    p.rows = rows; p.table = table; p.out_bias = out_bias; p.labels = labels; p.R = R; p.C = C; p.I = I; p.i0 = i0;
    p.i1 = i1; p.nvalid = nvalid; p.row_lse = const_cast<float*>(row_lse); p.coef = coef; p.gscale = gscale;
    p.defer_label = (defer_label_term & 1) != 0;
    p.acc_atomic = (defer_label_term & 2) != 0;      // (bf16, C = 128 strip path only; elsewhere the slabs)
    p.keep_slabs = (defer_label_term & 4) != 0 && (defer_label_term & 1) != 0 && !gscale && i0 == 0 && i1 == I;
    const BwdPlan plan = bwd_plan(R, C, I, i1 - i0, dtype == EDGL_BF16 ? 2 : 4, use_strip(C, dtype == EDGL_BF16 ? 2 : 4));
    hipStream_t st = (hipStream_t)stream;
    return dtype == EDGL_F32 ? bwd_dispatch<float, 2>(p, C, plan, workspace, d_rows, d_table, d_bias, st)
                             : bwd_dispatch<bf16, 2>(p, C, plan, workspace, d_rows, d_table, d_bias, st);
}

// Where edgl_score_flash_bwd_ex(defer_label_term & 4) leaves the partial table / bias gradients: out[0] / out[1] = float offsets of the
// [nslab][I * C] and [nslab][I - 1] slabs inside the flash workspace, out[2] = nslab.  Depends on the shape only.
extern "C" int edgl_score_flash_slab_info(int R, int C, int I, int n_items, int dtype, long* out) {
    EDGL_REQUIRE(out && (dtype == EDGL_F32 || dtype == EDGL_BF16), EDGL_ERR_NULL, "edgl_score_flash_slab_info: bad arguments");
    const BwdPlan plan = bwd_plan(R, C, I, n_items, dtype == EDGL_BF16 ? 2 : 4, use_strip(C, dtype == EDGL_BF16 ? 2 : 4));
    out[0] = plan.off_slabW; out[1] = plan.off_slabB; out[2] = plan.w.nchunk;
    return EDGL_OK;
}

extern "C" int edgl_mask_topk(float* logits, int R, int n, int i0, const int64_t* seen, int T, int K, float* out_val,
                              int32_t* out_idx, void* stream) {
    EDGL_REQUIRE(logits && out_val && out_idx, EDGL_ERR_NULL, "edgl_mask_topk: null pointer");
    EDGL_REQUIRE(R > 0 && n > 0 && K > 0 && K <= 128, EDGL_ERR_SHAPE, "edgl_mask_topk: bad shape R=%d n=%d K=%d", R, n, K);
    static const int reg_form = getenv("EDGL_TOPK_REG") ? atoi(getenv("EDGL_TOPK_REG")) : 1;
    hipStream_t st = (hipStream_t)stream;
    if (reg_form && n <= 256 * 16 - 8) hipLaunchKernelGGL(mask_topk_reg_kernel<16>, dim3(R), dim3(256), 0, st, logits, R, n, i0, seen, T, K, out_val, out_idx);
    else if (reg_form && n <= 256 * 40 - 8) hipLaunchKernelGGL(mask_topk_reg_kernel<40>, dim3(R), dim3(256), 0, st, logits, R, n, i0, seen, T, K, out_val, out_idx);
    else if (reg_form && n <= 256 * 80 - 8) hipLaunchKernelGGL(mask_topk_reg_kernel<80>, dim3(R), dim3(256), 0, st, logits, R, n, i0, seen, T, K, out_val, out_idx);
    else hipLaunchKernelGGL(mask_topk_kernel, dim3(R), dim3(256), 0, st, logits, R, n, i0, seen, T, K, out_val, out_idx);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_topk_merge(const float* cand_val, const int32_t* cand_idx, int S, int R, int K, float* out_val,
                               int32_t* out_idx, void* stream) {
    EDGL_REQUIRE(cand_val && cand_idx && out_val && out_idx, EDGL_ERR_NULL, "edgl_topk_merge: null pointer");
    EDGL_REQUIRE(S > 0 && R > 0 && K > 0 && S * K <= 1024, EDGL_ERR_SHAPE, "edgl_topk_merge: S*K=%d > 1024", S * K);
    hipLaunchKernelGGL(topk_merge_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, cand_val, cand_idx, S, R, K, out_val, out_idx,
                       (const int32_t*)nullptr);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
// the merge over appended candidate lists (k_eval_topk.hip)
int edgl_topk_merge_counted(const float* cand_val, const int32_t* cand_idx, const int32_t* counts, int S, int R, int K, float* out_val,
                            int32_t* out_idx, hipStream_t st) {
    hipLaunchKernelGGL(topk_merge_kernel, dim3(R), dim3(256), 0, st, cand_val, cand_idx, S, R, K, out_val, out_idx, counts);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_rank_metrics(const int32_t* topk_idx, int R, int K, const int64_t* label, float* metrics, void* stream) {
    EDGL_REQUIRE(topk_idx && label && metrics, EDGL_ERR_NULL, "edgl_rank_metrics: null pointer");
    hipLaunchKernelGGL(rank_metrics_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, topk_idx, R, K, label, metrics);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

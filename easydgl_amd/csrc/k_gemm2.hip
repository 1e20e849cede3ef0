// bf16 fast paths of edgl_gemm for the shapes of the hot path (M = B*T rows huge; K, N in {128..512}).
// Both kernels rely on gfx950's LDS transpose read (ds_read_b64_tr_b16): when lane (G=l>>4, s=l&15)
// points at T[k0 + 4G + (s>>2)][z0 + 4*(s&3) ..+3] of a row-major LDS tile, it receives
// T[k0 + 4G + j][z0 + (l&15)], j = 0..3 — i.e. the MFMA A/B fragment whose CONTRACTION index runs along the
// tile's rows (measured on MI355X with tools/probe_tr.hip).  A 16x16x32 MFMA takes two such reads
// (rows k0..k0+15 and k0+16..k0+31); both operands are read with the same k-slot order, so it is consistent.
//
//   strip_gemm_kernel : C[M,N] = epi(A[M,K] . B)   (dense forward and dX)
//       a wave keeps a 32-row strip of A as MFMA fragments in registers for the whole kernel (A is read from
//       HBM exactly once), B streams through LDS in 64-column tiles (double buffered, one barrier per tile);
//       B is either [N][K] (k contiguous: plain ds_read_b128) or [K][N] (n contiguous: transpose reads), so
//       neither the forward (x.W) nor dX (dz.W^T) needs a transposed copy of the weights.
//   tn_gemm_kernel    : C[Kf,N] (+ bias-gradient row) = X[R,Kf]^T . dY[R,N]   (dW, db)
//       contraction over the R rows with both operands row-major: tiles are staged row-major with coalesced
//       16-byte copies and BOTH operands are fetched with transpose reads; split over R with f32 partial slabs
//       reduced in a fixed order; the column sums of dY (bias gradient) ride along as one extra output row.
#include <cstdlib>

#include "edgl_common.h"

typedef __attribute__((ext_vector_type(4))) short tr_s16x4;

namespace gemm2 {

__device__ __forceinline__ uint2 tr_read(const bf16* tile, int ld, int k0, int z0, int lane) {
    const int G = lane >> 4, s = lane & 15;
    const bf16* p = tile + (k0 + 4 * G + (s >> 2)) * ld + z0 + 4 * (s & 3);
    tr_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_s16x4*)p);
    return *reinterpret_cast<uint2*>(&v);
}
// 8-slot fragment for the 16x16x32 MFMA: slots 0-3 <-> k0+4G+j, slots 4-7 <-> k0+16+4G+j
__device__ __forceinline__ bf16x8 tr_frag32(const bf16* tile, int ld, int k0, int z0, int lane) {
    bf16x8 f;
    *reinterpret_cast<uint2*>(&f) = tr_read(tile, ld, k0, z0, lane);
    *(reinterpret_cast<uint2*>(&f) + 1) = tr_read(tile, ld, k0 + 16, z0, lane);
    return f;
}

struct EpiP {
    const float* bias; void* aux; int flags;
};

// XCD-aware workgroup order.  The hardware deals workgroup b of a launch to XCD b % 8 (observed; speed only), and each XCD has its
// own 4 MB L2: consecutive ids — which these kernels give to the tiles that stream the SAME operand rows — would land on eight
// different L2s and every one of them would fetch those rows again (QKVT projection: 162 MB fetched for 40 MB of activations, its
// dX 167 for 53, the grouped dW 434 for ~110).  With a grid padded to a multiple of 8 the virtual id  v = (b % 8) * (G / 8) + b / 8
// hands every XCD one contiguous run of virtual ids; ids >= n_real are padding.  Returns -1 for padding workgroups.
__device__ __forceinline__ int xcd_virtual_id(int b, int grid, int n_real, int on) {
    if (!on || (grid & 7)) return b < n_real ? b : -1;
    const int v = (b & 7) * (grid >> 3) + (b >> 3);
    return v < n_real ? v : -1;
}
static inline int xcd_grid(int n_real) { return (n_real + 7) / 8 * 8; }
static inline int xcd_on() {
    static const int on = getenv("EDGL_XCD_ORDER") ? atoi(getenv("EDGL_XCD_ORDER")) : 1;
    return on;
}
__device__ __forceinline__ void epi_store4(const EpiP& e, bf16* C, void* Cany, long idx, int n, float x[4]) {
    if (e.flags & EDGL_EPI_BIAS) {
        const float4 b = *reinterpret_cast<const float4*>(e.bias + n);
        x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w;
    }
    if (e.flags & EDGL_EPI_SAVE_PRE) {
        Frag4<bf16> f = frag_from_acc<bf16>(f32x4{x[0], x[1], x[2], x[3]});
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(e.aux) + idx) = *reinterpret_cast<uint2*>(&f);
    }
    if (e.flags & EDGL_EPI_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = gelu_t<bf16>(x[r]);
    }
    if (e.flags & EDGL_EPI_RELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
    }
    if (e.flags & EDGL_EPI_MUL_DGELU) {
        const Frag4<bf16> a = frag_ld<bf16>(reinterpret_cast<const bf16*>(e.aux) + idx);
#pragma unroll
        for (int r = 0; r < 4; ++r) x[r] *= dgelu_t<bf16>(to_f32(a.v[r]));
    }
    if (e.flags & EDGL_EPI_OUT_F32) {
        float* c = reinterpret_cast<float*>(Cany) + idx;
        if (e.flags & EDGL_EPI_ACCUM) {
            const float4 o = *reinterpret_cast<const float4*>(c);
            x[0] += o.x; x[1] += o.y; x[2] += o.z; x[3] += o.w;
        }
        *reinterpret_cast<float4*>(c) = make_float4(x[0], x[1], x[2], x[3]);
    } else {
        if (e.flags & EDGL_EPI_ACCUM) {
            const Frag4<bf16> o = frag_ld<bf16>(C + idx);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] += to_f32(o.v[r]);
        }
        Frag4<bf16> f = frag_from_acc<bf16>(f32x4{x[0], x[1], x[2], x[3]});
        *reinterpret_cast<uint2*>(C + idx) = *reinterpret_cast<uint2*>(&f);
    }
}

// ------------------------------------------------------------------------------------------------
// strip GEMM
// ------------------------------------------------------------------------------------------------
constexpr int S_NT = 256;   // 4 waves x 32 rows = 128 rows per workgroup
constexpr int S_ZB = 64;    // output columns per streamed tile

struct StripP {
    const bf16* A; const bf16* B; void* C;
    int M, N, K, lda, ldb, ldc;
    EpiP epi;
    int dbg;   // EDGL_DBG ablation bits (profiling only): 1 skip epilogue stores, 2 skip MFMA loop, 4 skip B streaming
    int xcd;   // tile_nn_kernel: XCD-aware workgroup order (xcd_virtual_id)
};

// Epilogue of one [32 rows x 64 columns] accumulator block of a wave.  acc[jz][ix] = L(first = n, second = m): a lane holds 4
// consecutive n of one row, which would make every global store a 16-row x 32-byte scatter; the values therefore go
// through a wave-private LDS image and leave as whole 128-byte row segments.  bias64: LDS, the 64 bias values of the block.
struct StripEpi {
    bf16* Ostage; float* OstageF; const float* bias64;
    bool staged, fstage;
    uint4 pre_aux[2][2], pre_c[2][2];
};
__device__ __forceinline__ void strip_prefetch_combine(const StripP& p, StripEpi& e, int m0, int n0, int lane) {
    // ·GELU'(aux) / += C epilogues: the 16-byte pieces this lane will combine with are fetched before the MFMA loop
    // (whole 128-byte row segments) and used after it
    if (e.fstage) {
#pragma unroll
        for (int ix = 0; ix < 2; ++ix)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int m = min(m0 + ix * 16 + q * 8 + (lane >> 3), p.M - 1);
                const long idx = (long)m * p.ldc + n0 + (lane & 7) * 8;
                if (p.epi.flags & EDGL_EPI_MUL_DGELU) e.pre_aux[ix][q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.epi.aux) + idx);
                if (p.epi.flags & EDGL_EPI_ACCUM) e.pre_c[ix][q] = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16*>(p.C) + idx);
            }
    }
}
__device__ __forceinline__ void strip_epilogue(const StripP& p, StripEpi& e, f32x4 (&acc)[4][2], int m0, int n0, int lane) {
    constexpr int LDO = 64 + 8, LDOF = 64 + 4;
    const int g4 = (lane >> 4) * 4, l15 = lane & 15;
    bf16* const Ostage = e.Ostage;
    float* const OstageF = e.OstageF;
    if (p.dbg & 1) return;
    if (e.staged) {
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
#pragma unroll
            for (int jz = 0; jz < 4; ++jz) {
                const int n = n0 + jz * 16 + g4;
                float x[4] = {acc[jz][ix][0], acc[jz][ix][1], acc[jz][ix][2], acc[jz][ix][3]};
                {   // bias from the LDS copy: a global load here would put one memory round trip per tile on the
                    // critical path of the epilogue
                    const float4 bb = *reinterpret_cast<const float4*>(e.bias64 + jz * 16 + g4);
                    x[0] += bb.x; x[1] += bb.y; x[2] += bb.z; x[3] += bb.w;
                }
                if (p.epi.flags & EDGL_EPI_SAVE_PRE) {   // pre-activation image (2 GEMMs per step)
                    const int m = m0 + ix * 16 + l15;
                    if (m < p.M) {
                        Frag4<bf16> f = frag_from_acc<bf16>(f32x4{x[0], x[1], x[2], x[3]});
                        *reinterpret_cast<uint2*>(reinterpret_cast<bf16*>(p.epi.aux) + (long)m * p.ldc + n) = *reinterpret_cast<uint2*>(&f);
                    }
                }
                if (p.epi.flags & EDGL_EPI_GELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] = gelu_t<bf16>(x[r]);
                }
                if (p.epi.flags & EDGL_EPI_RELU) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
                }
                Frag4<bf16> f = frag_from_acc<bf16>(f32x4{x[0], x[1], x[2], x[3]});
                *reinterpret_cast<uint2*>(Ostage + l15 * LDO + jz * 16 + g4) = *reinterpret_cast<uint2*>(&f);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // 16 rows x 128 B: 8 lanes per row, 8 rows per instruction, 2 instructions
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int lrow = q * 8 + (lane >> 3), cv = lane & 7, m = m0 + ix * 16 + lrow;
                const uint4 d = *reinterpret_cast<const uint4*>(Ostage + lrow * LDO + cv * 8);
                if (m < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.C) + (long)m * p.ldc + n0 + cv * 8) = d;
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else if (e.fstage) {
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
#pragma unroll
            for (int jz = 0; jz < 4; ++jz) {
                const float4 bb = *reinterpret_cast<const float4*>(e.bias64 + jz * 16 + g4);
                *reinterpret_cast<float4*>(OstageF + l15 * LDOF + jz * 16 + g4) =
                    make_float4(acc[jz][ix][0] + bb.x, acc[jz][ix][1] + bb.y, acc[jz][ix][2] + bb.z, acc[jz][ix][3] + bb.w);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int lrow = q * 8 + (lane >> 3), cv = lane & 7, m = m0 + ix * 16 + lrow;
                const float4 lo = *reinterpret_cast<const float4*>(OstageF + lrow * LDOF + cv * 8);
                const float4 hi = *reinterpret_cast<const float4*>(OstageF + lrow * LDOF + cv * 8 + 4);
                float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                if (p.epi.flags & EDGL_EPI_MUL_DGELU) {
                    const bf16x8 a = *reinterpret_cast<const bf16x8*>(&e.pre_aux[ix][q]);
#pragma unroll
                    for (int r = 0; r < 8; ++r) x[r] *= dgelu_t<bf16>(to_f32(a[r]));
                }
                if (p.epi.flags & EDGL_EPI_ACCUM) {
                    const bf16x8 o = *reinterpret_cast<const bf16x8*>(&e.pre_c[ix][q]);
#pragma unroll
                    for (int r = 0; r < 8; ++r) x[r] += to_f32(o[r]);
                }
                const Frag4<bf16> f0 = frag_from_acc<bf16>(f32x4{x[0], x[1], x[2], x[3]});
                const Frag4<bf16> f1 = frag_from_acc<bf16>(f32x4{x[4], x[5], x[6], x[7]});
                uint4 d;
                *reinterpret_cast<uint2*>(&d) = *reinterpret_cast<const uint2*>(&f0);
                *(reinterpret_cast<uint2*>(&d) + 1) = *reinterpret_cast<const uint2*>(&f1);
                if (m < p.M) *reinterpret_cast<uint4*>(reinterpret_cast<bf16*>(p.C) + (long)m * p.ldc + n0 + cv * 8) = d;
            }
            __builtin_amdgcn_wave_barrier();
        }
    } else {
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
            const int m = m0 + ix * 16 + l15;
            if (m < p.M) {
#pragma unroll
                for (int jz = 0; jz < 4; ++jz) {
                    const int n = n0 + jz * 16 + g4;
                    float x[4] = {acc[jz][ix][0], acc[jz][ix][1], acc[jz][ix][2], acc[jz][ix][3]};
                    epi_store4(p.epi, reinterpret_cast<bf16*>(p.C), p.C, (long)m * p.ldc + n, n, x);
                }
            }
        }
    }
}

// Weights-resident strip GEMM.  A workgroup (8 waves) loads ONE column slice of B (<= 128 output columns, all K)
// into LDS once, then its waves independently walk 32-row strips of A: strip fragments -> registers, MFMA against
// the LDS-resident weights, epilogue, next strip.  There is no barrier and no shared traffic in the main loop, so
// the two waves of a SIMD hide each other's memory latency, and A is touched only by the (<= 4) column-slice
// workgroups that run side by side on the same rows.
// NKB = K/32 ; B_KC: B stored [N][ldb] (k contiguous) else [K][ldb] (n contiguous, read with transpose reads)
constexpr int W_NT = 512;
#ifdef EDGL_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[16];
#endif
template <int NKB, bool B_KC>
__global__ __launch_bounds__(W_NT) void strip_gemm_kernel(StripP p) {
    constexpr int K = 32 * NKB;
    constexpr int NP = 128;                 // columns per slice (gridDim.y slices)
    constexpr int LDW_KC = K + 16;          // [NP][K+16]   rows n (+32 B: conflict-free b128 reads)
    constexpr int LDW_TR = NP + 16;         // [K][NP+16]   rows k
    constexpr int WELEMS = B_KC ? NP * LDW_KC : K * LDW_TR;
    constexpr int LDO = 64 + 8;             // per-wave output staging [16][64+8] (one 16-row tile at a time)
    constexpr int LDOF = 64 + 4;            // f32 staging [16][64+4] for the epilogues that combine with another tensor
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PH_DECL
    bf16* const Ws = reinterpret_cast<bf16*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4, g4 = G * 4, l15 = lane & 15;
    const bool fstage = (p.epi.flags & (EDGL_EPI_ACCUM | EDGL_EPI_MUL_DGELU)) && !(p.epi.flags & EDGL_EPI_OUT_F32);
    float* const biasS = reinterpret_cast<float*>(Ws + WELEMS);                   // this slice's bias (zeros if none)
    bf16* const Ostage = Ws + WELEMS + 256 + wave * 16 * LDO;                     // (256 bf16 = the 128 bias floats)
    float* const OstageF = reinterpret_cast<float*>(Ws + WELEMS + 256) + wave * 16 * LDOF;
    const int nbase = blockIdx.y * NP;
    const int ncols = min(NP, p.N - nbase);          // multiple of 64
    if (p.dbg & 16) { if (tid < NP) biasS[tid] = 0.f; __syncthreads(); goto main_loop; }   // ablation: no weight staging
    if (tid < NP) biasS[tid] = ((p.epi.flags & EDGL_EPI_BIAS) && nbase + tid < p.N) ? p.epi.bias[nbase + tid] : 0.f;
    // ---- weights slice -> LDS (once).  Unconditional loads (row / column clamped into the slice: the clamped copies land
    //      in rows / columns >= ncols that no strip reads) issued in batches of 4, so the slice arrives in a few memory
    //      round trips instead of one per 16-byte piece. ------------------------------------------------------------
    {
        constexpr int TOTAL = B_KC ? NP * (K / 8) : K * (NP / 8);
        constexpr int ITER = (TOTAL + W_NT - 1) / W_NT;
#pragma unroll
        for (int i0 = 0; i0 < ITER; i0 += 4) {
            uint4 d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = min(tid + (i0 + j) * W_NT, TOTAL - 1);
                if constexpr (B_KC) {
                    const int row = min(v / (K / 8), ncols - 1), kv = v % (K / 8);
                    d[j] = *reinterpret_cast<const uint4*>(p.B + (long)(nbase + row) * p.ldb + kv * 8);
                } else {
                    const int row = v / (NP / 8), nv = min(v % (NP / 8), ncols / 8 - 1);
                    d[j] = *reinterpret_cast<const uint4*>(p.B + (long)row * p.ldb + nbase + nv * 8);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int v = tid + (i0 + j) * W_NT;
                if (i0 + j < ITER && v < TOTAL) {
                    if constexpr (B_KC) *reinterpret_cast<uint4*>(Ws + (v / (K / 8)) * LDW_KC + (v % (K / 8)) * 8) = d[j];
                    else *reinterpret_cast<uint4*>(Ws + (v / (NP / 8)) * LDW_TR + (v % (NP / 8)) * 8) = d[j];
                }
            }
        }
    }
    __syncthreads();
main_loop:
    PH_MARK(0);   // weight slice staged
    const bool staged = !(p.epi.flags & EDGL_EPI_OUT_F32) && !fstage;
    StripEpi se;
    se.Ostage = Ostage; se.OstageF = OstageF; se.staged = staged; se.fstage = fstage;
    const int nstrip = (p.M + 31) / 32;
    for (int strip = blockIdx.x * 8 + wave; strip < nstrip; strip += gridDim.x * 8) {
        const int m0 = strip * 32;
        // ---- A strip fragments (k-slot order = B fragment order) ---------------------------------------------------
        bf16x8 xf[2][NKB];
#pragma unroll
        for (int ix = 0; ix < 2; ++ix) {
            // rows past M are clamped, not branched around: their results are never stored, and a per-lane branch
            // around a load makes the compiler wait for it inside the branch (one memory round trip per fragment)
            const int m = min(m0 + ix * 16 + l15, p.M - 1);
            const bf16* row = p.A + (long)m * p.lda;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                bf16x8 f;
                if (p.dbg & 8) { *reinterpret_cast<uint4*>(&f) = make_uint4(lane, kb, 0, 0); xf[ix][kb] = f; continue; }   // ablation: no A loads
                if constexpr (B_KC) {
                    *reinterpret_cast<uint4*>(&f) = *reinterpret_cast<const uint4*>(row + kb * 32 + G * 8);
                } else {
                    *reinterpret_cast<uint2*>(&f) = *reinterpret_cast<const uint2*>(row + kb * 32 + g4);
                    *(reinterpret_cast<uint2*>(&f) + 1) = *reinterpret_cast<const uint2*>(row + kb * 32 + 16 + g4);
                }
                xf[ix][kb] = f;
            }
        }
        PH_MARK(1);   // strip loads issued (+ waited where the first MFMA needs them: counted in phase 2)
        for (int nc = 0; nc < ncols; nc += 64) {
            f32x4 acc[4][2];
#pragma unroll
            for (int jz = 0; jz < 4; ++jz) { acc[jz][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[jz][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
            strip_prefetch_combine(p, se, m0, nbase + nc, lane);
            if (!(p.dbg & 2))
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
                for (int jz = 0; jz < 4; ++jz) {
                    bf16x8 zf;
                    if constexpr (B_KC) zf = *reinterpret_cast<const bf16x8*>(Ws + (nc + jz * 16 + l15) * LDW_KC + kb * 32 + G * 8);
                    else zf = tr_frag32(Ws, LDW_TR, kb * 32, nc + jz * 16, lane);
                    acc[jz][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, xf[0][kb], acc[jz][0], 0, 0, 0);
                    acc[jz][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, xf[1][kb], acc[jz][1], 0, 0, 0);
                }
            }
            PH_MARK(2);   // MFMA loop (incl. the wait for the strip)
            se.bias64 = biasS + nc;
            strip_epilogue(p, se, acc, m0, nbase + nc, lane);
            PH_MARK(3);   // epilogue
        }
    }
    PH_FLUSH(0);
}

// Activations-in-registers, weights-streamed variant for wide outputs (N >= 3 column slices).  The weights-resident
// kernel above makes every column slice re-read all of A (the dominant cost once A no longer fits the L2 working set:
// 4 x 40 MB for the QKVT projection); here a workgroup owns 256 rows — 8 waves x one 32-row strip held in registers for
// the whole kernel — and walks ALL output columns in 64-column chunks of the weights, double-buffered through LDS (the
// weights are small and L2-resident, so re-reading THEM per workgroup is cheap).  One LDS-scoped barrier per chunk.
template <int NKB, bool B_KC>
__global__ __launch_bounds__(W_NT) void strip_stream_kernel(StripP p) {
    constexpr int K = 32 * NKB;
    constexpr int NCH = 64;                 // output columns per chunk
    constexpr int LDW_KC = K + 16;          // [NCH][K+16]
    constexpr int LDW_TR = NCH + 16;        // [K][NCH+16]
    constexpr int CELEMS = B_KC ? NCH * LDW_KC : K * LDW_TR;
    constexpr int LDO = 64 + 8, LDOF = 64 + 4;
    constexpr int TOTAL = B_KC ? NCH * (K / 8) : K * (NCH / 8);   // 16-byte pieces per chunk
    constexpr int ITER = (TOTAL + W_NT - 1) / W_NT;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    PH_DECL
    bf16* const Wbuf = reinterpret_cast<bf16*>(smem);              // [2][CELEMS]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4, g4 = G * 4, l15 = lane & 15;
    float* const biasS = reinterpret_cast<float*>(Wbuf + 2 * CELEMS);   // [N] (zeros if none)
    const int nb4 = (p.N + 3) / 4 * 4;
    bf16* const stage0 = reinterpret_cast<bf16*>(biasS + nb4);
    const bool fstage = (p.epi.flags & (EDGL_EPI_ACCUM | EDGL_EPI_MUL_DGELU)) && !(p.epi.flags & EDGL_EPI_OUT_F32);
    StripEpi se;
    se.Ostage = stage0 + wave * 16 * LDO;
    se.OstageF = reinterpret_cast<float*>(stage0) + wave * 16 * LDOF;
    se.fstage = fstage;
    se.staged = !(p.epi.flags & EDGL_EPI_OUT_F32) && !fstage;
    for (int i = tid; i < p.N; i += W_NT) biasS[i] = (p.epi.flags & EDGL_EPI_BIAS) ? p.epi.bias[i] : 0.f;

    // chunk loader: global -> registers (unconditional, coalesced 16-byte pieces) ... -> LDS
    uint4 wreg[ITER];
    auto load_chunk = [&](int nc) {
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int v = min(tid + i * W_NT, TOTAL - 1);
            if constexpr (B_KC) wreg[i] = *reinterpret_cast<const uint4*>(p.B + (long)(nc + v / (K / 8)) * p.ldb + (v % (K / 8)) * 8);
            else wreg[i] = *reinterpret_cast<const uint4*>(p.B + (long)(v / (NCH / 8)) * p.ldb + nc + (v % (NCH / 8)) * 8);
        }
    };
    auto store_chunk = [&](bf16* W) {
#pragma unroll
        for (int i = 0; i < ITER; ++i) {
            const int v = tid + i * W_NT;
            if (v < TOTAL) {
                if constexpr (B_KC) *reinterpret_cast<uint4*>(W + (v / (K / 8)) * LDW_KC + (v % (K / 8)) * 8) = wreg[i];
                else *reinterpret_cast<uint4*>(W + (v / (NCH / 8)) * LDW_TR + (v % (NCH / 8)) * 8) = wreg[i];
            }
        }
    };
    load_chunk(0);
    // ---- this wave's strip of A: registers, for the whole kernel -------------------------------------------------------
    const int m0 = (blockIdx.x * 8 + wave) * 32;
    bf16x8 xf[2][NKB];
#pragma unroll
    for (int ix = 0; ix < 2; ++ix) {
        const int m = min(m0 + ix * 16 + l15, p.M - 1);   // clamped: rows past M are never stored
        const bf16* row = p.A + (long)m * p.lda;
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
            bf16x8 f;
            if constexpr (B_KC) {
                *reinterpret_cast<uint4*>(&f) = *reinterpret_cast<const uint4*>(row + kb * 32 + G * 8);
            } else {
                *reinterpret_cast<uint2*>(&f) = *reinterpret_cast<const uint2*>(row + kb * 32 + g4);
                *(reinterpret_cast<uint2*>(&f) + 1) = *reinterpret_cast<const uint2*>(row + kb * 32 + 16 + g4);
            }
            xf[ix][kb] = f;
        }
    }
    store_chunk(Wbuf);
    lds_barrier();
    PH_MARK(4);   // prologue: strip + first chunk
    const int nchunks = p.N / NCH;
    for (int c = 0; c < nchunks; ++c) {
        const int nc = c * NCH;
        const bf16* Ws = Wbuf + (c & 1) * CELEMS;
        strip_prefetch_combine(p, se, m0, nc, lane);
        f32x4 acc[4][2];
#pragma unroll
        for (int jz = 0; jz < 4; ++jz) { acc[jz][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[jz][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb) {
#pragma unroll
            for (int jz = 0; jz < 4; ++jz) {
                bf16x8 zf;
                if constexpr (B_KC) zf = *reinterpret_cast<const bf16x8*>(Ws + (jz * 16 + l15) * LDW_KC + kb * 32 + G * 8);
                else zf = tr_frag32(Ws, LDW_TR, kb * 32, jz * 16, lane);
                acc[jz][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, xf[0][kb], acc[jz][0], 0, 0, 0);
                acc[jz][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(zf, xf[1][kb], acc[jz][1], 0, 0, 0);
            }
        }
        PH_MARK(5);   // MFMA loop issue
        if (c + 1 < nchunks) load_chunk(nc + NCH);          // next chunk's pieces fly during the epilogue (issuing them
                                                            // before the MFMA loop costs 24-32 more live registers)
        se.bias64 = biasS + nc;
        strip_epilogue(p, se, acc, m0, nc, lane);
        PH_MARK(6);   // MFMA drain + epilogue
        if (c + 1 < nchunks) store_chunk(Wbuf + ((c + 1) & 1) * CELEMS);   // that buffer was last read in iteration c - 1
        lds_barrier();
        PH_MARK(7);   // next chunk -> LDS, barrier
    }
    PH_FLUSH(0);
}

// ------------------------------------------------------------------------------------------------
// 128 x 128 tiled GEMM for the wide projections (QKVT forward: [M, 384] . [384, 512] + bias; its dX: [M, 512] . [512, 384])
// ------------------------------------------------------------------------------------------------
// The strip kernels give every wave 32 rows x ALL columns: 8 epilogues and 8 workgroup barriers per strip, the 8 waves of a
// workgroup in lock-step, 202 workgroups for 256 CUs.  Here a 4-wave workgroup (2 x 2 waves, 64 x 64 outputs each = 16
// accumulator tiles) owns one 128 x 128 output tile and walks K in 64-wide steps: register prefetch of the next step; the general-
// epilogue form double-buffers both operand tiles in LDS (one LDS-scoped barrier per step, two workgroups per CU), the bias-only
// form keeps one buffer and runs three workgroups per CU (see the step loop); 1616 workgroups
// for the QKVT shape; the output tile leaves through LDS as whole 256-byte row segments.  Epilogue: optional bias.
constexpr int G_BM = 128, G_BN = 128, G_BK = 64, G_NT = 256;
constexpr int G_LDK = G_BK + 8;      // [rows][k] images (A; B when k-contiguous)
constexpr int G_LDN = G_BN + 16;     // [k][n] image (B when n-contiguous: transpose reads)
constexpr int G_LDO = G_BN + 8;      // output staging [128][128]
// EPI = false: bias-only epilogue, bf16 output staged through LDS.  EPI = true: every epilogue of epi_store4 (GELU / ReLU with
// the saved pre-activation, x GELU', accumulate, f32 output) written straight from the accumulator fragments (8 / 16 bytes per
// lane) — the wide layers of the 512-unit recipes (K = 1024 / 1536 / 2048), which the register-strip kernels (K <= 512) do not take.
template <bool B_KC, bool EPI = false>
__global__ __launch_bounds__(G_NT, EPI ? 2 : 3) void tile_nn_kernel(StripP p) {
    constexpr int A_EL = G_BM * G_LDK, B_EL = B_KC ? G_BN * G_LDK : G_BK * G_LDN;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* As = reinterpret_cast<bf16*>(smem);                 // [2][A_EL]
    bf16* Bs = As + (EPI ? 2 : 1) * A_EL;                      // [buffers][B_EL]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int G = lane >> 4, g4 = G * 4, l15 = lane & 15;
    // consecutive workgroups share the row block (its A rows stay in L2 across the N / 128 column tiles)
    const int nbn = p.N / G_BN;
    const int vid = xcd_virtual_id((int)blockIdx.x, (int)gridDim.x, ((p.M + G_BM - 1) / G_BM) * nbn, p.xcd);
    if (vid < 0) return;
    const int m0 = (vid / nbn) * G_BM, n0 = (vid % nbn) * G_BN;
    uint4 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3;   // named registers (arrays captured by a lambda end up in scratch here)
    // piece v of an operand tile: A (and B when k-contiguous): row v >> 3, k offset (v & 7) * 8; B n-contiguous: k row v >> 4, n offset (v & 15) * 8
#define G_LOAD_ONE(PA, PB, I)                                                                                                  \
    {                                                                                                                          \
        const int v = tid + (I) * G_NT, ar = v >> 3, ac = (v & 7) * 8;                                                         \
        PA = *reinterpret_cast<const uint4*>(p.A + (long)min(m0 + ar, p.M - 1) * p.lda + k0_ + ac);                            \
        if constexpr (B_KC) PB = *reinterpret_cast<const uint4*>(p.B + (long)(n0 + ar) * p.ldb + k0_ + ac);                    \
        else PB = *reinterpret_cast<const uint4*>(p.B + (long)(k0_ + (v >> 4)) * p.ldb + n0 + (v & 15) * 8);                   \
    }
#define G_LOAD(K0) { const int k0_ = (K0); G_LOAD_ONE(pa0, pb0, 0) G_LOAD_ONE(pa1, pb1, 1) G_LOAD_ONE(pa2, pb2, 2) G_LOAD_ONE(pa3, pb3, 3) }
#define G_STORE_ONE(PA, PB, I, BUF)                                                                                            \
    {                                                                                                                          \
        const int v = tid + (I) * G_NT, ar = v >> 3, ac = (v & 7) * 8;                                                         \
        *reinterpret_cast<uint4*>(As + (BUF) * A_EL + ar * G_LDK + ac) = PA;                                                   \
        if constexpr (B_KC) *reinterpret_cast<uint4*>(Bs + (BUF) * B_EL + ar * G_LDK + ac) = PB;                               \
        else *reinterpret_cast<uint4*>(Bs + (BUF) * B_EL + (v >> 4) * G_LDN + (v & 15) * 8) = PB;                              \
    }
#define G_STORE(BUF) { G_STORE_ONE(pa0, pb0, 0, BUF) G_STORE_ONE(pa1, pb1, 1, BUF) G_STORE_ONE(pa2, pb2, 2, BUF) G_STORE_ONE(pa3, pb3, 3, BUF) }
    f32x4 acc[4][4];   // [i: m tile][j: n tile], L(first = n, second = m)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nstep = p.K / G_BK;
    G_LOAD(0)
    G_STORE(0)
    lds_barrier();
    for (int st = 0; st < nstep; ++st) {
        // Bias-only form (the QKVT projection pair of the headline shape): ONE LDS buffer — 37 KB and <= 168 registers, i.e. THREE
        // workgroups per CU instead of two — and a second barrier per step.  The TA, LDS and MFMA phases of a step take turns within a
        // workgroup (a deeper register pipeline changed nothing); a third workgroup is what fills them: 43.0 -> 37.3 and 37.2 -> 34.9 us.
        const int buf = EPI ? (st & 1) : 0;
        if (st + 1 < nstep) G_LOAD((st + 1) * G_BK)
        const bf16* Ab = As + buf * A_EL + (wm * 64) * G_LDK;
        const bf16* Bb = Bs + buf * B_EL;
#pragma unroll
        for (int kb = 0; kb < G_BK / 32; ++kb) {
            bf16x8 af[4], bfr[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bf16* ar = Ab + (i * 16 + l15) * G_LDK + kb * 32;
                if constexpr (B_KC) {
                    af[i] = *reinterpret_cast<const bf16x8*>(ar + G * 8);
                } else {   // k-slot order of the transpose-read B fragment: slots 0-3 <-> 4G + j, slots 4-7 <-> 16 + 4G + j
                    *reinterpret_cast<uint2*>(&af[i]) = *reinterpret_cast<const uint2*>(ar + g4);
                    *(reinterpret_cast<uint2*>(&af[i]) + 1) = *reinterpret_cast<const uint2*>(ar + 16 + g4);
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (B_KC) bfr[j] = *reinterpret_cast<const bf16x8*>(Bb + (wn * 64 + j * 16 + l15) * G_LDK + kb * 32 + G * 8);
                else bfr[j] = tr_frag32(Bb, G_LDN, kb * 32, wn * 64 + j * 16, lane);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
        }
        if constexpr (!EPI) lds_barrier();     // every wave is done reading the buffer
        if (st + 1 < nstep) G_STORE(EPI ? (buf ^ 1) : 0)   // double-buffered: that buffer was last read in step st - 1
        lds_barrier();
    }
    if constexpr (EPI) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + l15;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + g4;
                float x[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                if (m < p.M) epi_store4(p.epi, reinterpret_cast<bf16*>(p.C), p.C, (long)m * p.ldc + n, n, x);
            }
        }
        return;
    }
    // ---- epilogue: bias, bf16, through LDS, whole 256-byte row segments -------------------------------------------------
    bf16* Os = reinterpret_cast<bf16*>(smem);   // [128][G_LDO] over the operand buffers (all reads are behind the last barrier)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int nl = wn * 64 + j * 16 + g4;
        float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.epi.flags & EDGL_EPI_BIAS) b4 = *reinterpret_cast<const float4*>(p.epi.bias + n0 + nl);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int ml = wm * 64 + i * 16 + l15;
            const Frag4<bf16> f = frag_from_acc<bf16>(f32x4{acc[i][j][0] + b4.x, acc[i][j][1] + b4.y, acc[i][j][2] + b4.z, acc[i][j][3] + b4.w});
            *reinterpret_cast<uint2*>(Os + ml * G_LDO + nl) = *reinterpret_cast<const uint2*>(&f);
        }
    }
    lds_barrier();
    bf16* C = reinterpret_cast<bf16*>(p.C);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int v = tid + i * G_NT, r = v >> 4, cpc = (v & 15) * 8;   // 128 rows x 16 pieces
        if (m0 + r < p.M) *reinterpret_cast<uint4*>(C + (long)(m0 + r) * p.ldc + n0 + cpc) = *reinterpret_cast<const uint4*>(Os + r * G_LDO + cpc);
    }
#undef G_LOAD_ONE
#undef G_LOAD
#undef G_STORE_ONE
#undef G_STORE
}

template <bool B_KC, bool EPI = false>
static int launch_tile_nn(const StripP& p, hipStream_t st) {
    constexpr size_t a_el = (size_t)G_BM * G_LDK, b_el = B_KC ? (size_t)G_BN * G_LDK : (size_t)G_BK * G_LDN;
    const size_t smem = std::max((EPI ? 2 : 1) * (a_el + b_el) * sizeof(bf16), (size_t)G_BM * G_LDO * sizeof(bf16));
    auto k = tile_nn_kernel<B_KC, EPI>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int nbm = (p.M + G_BM - 1) / G_BM, nbn = p.N / G_BN;
    StripP q = p;
    q.xcd = xcd_on();
    hipLaunchKernelGGL(k, dim3((unsigned)(q.xcd ? xcd_grid(nbm * nbn) : nbm * nbn)), dim3(G_NT), smem, st, q);
    EDGL_LAUNCH_CHECK();
    return 1;
}

// ------------------------------------------------------------------------------------------------
// TN GEMM (dW / db)
// ------------------------------------------------------------------------------------------------
constexpr int T_NT = 256, T_BR = 64, T_LD = 128 + 16;   // T_BR rows per step (128-row steps measured: grouped dW 70 -> 81 us)

struct TnP {
    const bf16* X; const bf16* Y; int R, Kf, N, ldx, ldy;
    int rows_per_split;
    float* partial;   // [splits][Kf + 1][N]  (row Kf = column sums of Y)
    int with_colsum;
    int tiles_n, tiles_k, nblocks, xcd;   // tn_gemm_kernel: 1-D grid (padded to a multiple of 8 with the XCD-aware order)
};

// TM = output tile edge of a workgroup (4 waves in a 2 x 2 arrangement): 128 (default) or 64 (see tn_tile)
template <int TM>
__device__ __forceinline__ void tn_body(const TnP& p, int bx, int by, int bz, bf16* Xs, bf16* Ys) {
    constexpr int NI = TM / 32, LDT = TM + 16, PCS = TM / 8, NLD = T_BR * PCS / T_NT;   // 16-byte pieces per row / per thread
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int kf0 = by * TM, n0 = bx * TM;
    const int r_lo = bz * p.rows_per_split, r_hi = min(p.R, r_lo + p.rows_per_split);
    f32x4 acc[NI][NI];   // [i: kf tile][j: n tile], L(first = kf, second = n)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // Column sums of Y (db) ride on the matrix pipe: a ones operand against the Y fragments this wave holds anyway, two of the four
    // n-tiles per wave (wm picks the pair) — +12 % matrix instructions instead of a 64-deep LDS loop on two of the four waves
    constexpr int NCS = NI / 2;
    f32x4 acc_cs[NCS];
#pragma unroll
    for (int j = 0; j < NCS; ++j) acc_cs[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bool do_cs = p.with_colsum && by == 0;
    bf16x8 ones;
#pragma unroll
    for (int i = 0; i < 8; ++i) ones[i] = (__bf16)1.0f;
    // (A second row tile in flight in registers behind the one in LDS was measured: grouped dW 72 -> 79 us, with these column sums
    // 106 us — the ~500 workgroups already keep the memory system's queues full, more requests only lengthen them.)
    uint4 px[NLD], py[NLD];
    int r0_cur = 0;
    // unconditional loads (row clamped into the split, column offset into the matrix); rows past the split are zeroed
    // when the tile is written to LDS — a branch around a load would serialise the prefetch on memory latency
    auto load = [&](int r0) {
        r0_cur = r0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int v = tid + i * T_NT, row = v / PCS, cv = v % PCS, gr = min(r0 + row, r_hi - 1);
            const int cx = min(kf0 + cv * 8, p.Kf - 8), cy = min(n0 + cv * 8, p.N - 8);
            px[i] = *reinterpret_cast<const uint4*>(p.X + (long)gr * p.ldx + cx);
            py[i] = *reinterpret_cast<const uint4*>(p.Y + (long)gr * p.ldy + cy);
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int v = tid + i * T_NT, row = v / PCS, cv = v % PCS;
            const bool ok = r0_cur + row < r_hi;
            *reinterpret_cast<uint4*>(Xs + row * LDT + cv * 8) = ok ? px[i] : make_uint4(0, 0, 0, 0);
            *reinterpret_cast<uint4*>(Ys + row * LDT + cv * 8) = ok ? py[i] : make_uint4(0, 0, 0, 0);
        }
    };
    const int nstep = (r_hi - r_lo + T_BR - 1) / T_BR;
    if (nstep > 0) { load(r_lo); store(); }
    __syncthreads();
    for (int st = 0; st < nstep; ++st) {
        const bool more = st + 1 < nstep;
        if (more) load(r_lo + (st + 1) * T_BR);
#pragma unroll
        for (int rb = 0; rb < T_BR / 32; ++rb) {
            bf16x8 af[NI], bf_[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) af[i] = tr_frag32(Xs, LDT, rb * 32, wm * (TM / 2) + i * 16, lane);
#pragma unroll
            for (int j = 0; j < NI; ++j) bf_[j] = tr_frag32(Ys, LDT, rb * 32, wn * (TM / 2) + j * 16, lane);
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bf_[j], acc[i][j], 0, 0, 0);
            if (do_cs) {
                if (wm) {
#pragma unroll
                    for (int j = 0; j < NCS; ++j) acc_cs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bf_[NCS + j], acc_cs[j], 0, 0, 0);
                } else {
#pragma unroll
                    for (int j = 0; j < NCS; ++j) acc_cs[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, bf_[j], acc_cs[j], 0, 0, 0);
                }
            }
        }
        lds_barrier();       // all reads of this tile done; the next tile's global loads stay in flight across it
        if (more) store();
        lds_barrier();
    }
    float* out = p.partial + (long)bz * (p.Kf + 1) * p.N;
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kf = kf0 + wm * (TM / 2) + i * 16 + g4 + r;
            if (kf < p.Kf) {
#pragma unroll
                for (int j = 0; j < NI; ++j) {
                    const int n = n0 + wn * (TM / 2) + j * 16 + l15;
                    if (n < p.N) out[(long)kf * p.N + n] = acc[i][j][r];
                }
            }
        }
    if (do_cs && lane < 16) {      // (every row of the ones product holds the sums: row 0 writes them)
#pragma unroll
        for (int j = 0; j < NCS; ++j) {
            const int n = n0 + wn * (TM / 2) + (wm * NCS + j) * 16 + lane;
            if (n < p.N) out[(long)p.Kf * p.N + n] = acc_cs[j][0];
        }
    }
}

template <int TM>
__global__ __launch_bounds__(T_NT) void tn_gemm_kernel(TnP p) {
    extern __shared__ __attribute__((aligned(16))) char tn_smem[];
    bf16* Xs = reinterpret_cast<bf16*>(tn_smem);
    bf16* Ys = Xs + T_BR * (TM + 16);
    // tiles of one row split are consecutive virtual ids: they stream the same rows of X and Y and share one XCD's L2
    const int vid = xcd_virtual_id((int)blockIdx.x, (int)gridDim.x, p.nblocks, p.xcd);
    if (vid < 0) return;
    tn_body<TM>(p, vid % p.tiles_n, (vid / p.tiles_n) % p.tiles_k, vid / (p.tiles_n * p.tiles_k), Xs, Ys);
}

// Several weight-gradient GEMMs in ONE launch (edgl_gemm_dw_defer): the dW products of a block are independent of each
// other and individually too small for the chip — the 128 x 128 layers run two 64-row steps per workgroup and leave 25 MB of
// partial slabs each.  Grouped, they share the launch with the wide QKVT product and get row splits of similar length.
constexpr int TN_MAX_JOBS = 8;
struct TnGroupP { TnP job[TN_MAX_JOBS]; int tiles_n[TN_MAX_JOBS], tiles_k[TN_MAX_JOBS], blk0[TN_MAX_JOBS + 1]; int n, xcd; };
__global__ __launch_bounds__(T_NT) void tn_gemm_group_kernel(TnGroupP g) {
    extern __shared__ __attribute__((aligned(16))) char tn_smem[];
    bf16* Xs = reinterpret_cast<bf16*>(tn_smem);
    bf16* Ys = Xs + T_BR * (128 + 16);
    // (the tiles of one row split are consecutive ids: with the XCD-aware order they share one L2)
    const int vid = xcd_virtual_id((int)blockIdx.x, (int)gridDim.x, g.blk0[g.n], g.xcd);
    if (vid < 0) return;
    int j = 0;
    for (int i = 1; i < g.n; ++i)
        if (vid >= g.blk0[i]) j = i;
    const int local = vid - g.blk0[j];
    const int tn = g.tiles_n[j], tk = g.tiles_k[j];
    tn_body<128>(g.job[j], local % tn, (local / tn) % tk, local / (tn * tk), Xs, Ys);
}

}  // namespace gemm2

// ------------------------------------------------------------------------------------------------
// host entry points used by edgl_gemm (k_gemm.hip) and exported for the training engine
// ------------------------------------------------------------------------------------------------
using namespace gemm2;

template <int NKB, bool B_KC>
static int launch_stream(const StripP& p, hipStream_t st) {
    constexpr int K = 32 * NKB;
    constexpr size_t cel = B_KC ? (size_t)64 * (K + 16) : (size_t)K * (64 + 16);
    const bool fstage = (p.epi.flags & (EDGL_EPI_ACCUM | EDGL_EPI_MUL_DGELU)) && !(p.epi.flags & EDGL_EPI_OUT_F32);
    const size_t smem = 2 * cel * sizeof(bf16) + (size_t)((p.N + 3) / 4 * 4) * sizeof(float) +
                        (fstage ? (size_t)8 * 16 * (64 + 4) * sizeof(float) : (size_t)8 * 16 * (64 + 8) * sizeof(bf16));
    if (smem > 160 * 1024) return 0;
    auto k = strip_stream_kernel<NKB, B_KC>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, dim3((p.M + 255) / 256), dim3(W_NT), smem, st, p);
    EDGL_LAUNCH_CHECK();
    return 1;
}

template <int NKB, bool B_KC>
static int launch_strip(const StripP& p, hipStream_t st) {
    constexpr int K = 32 * NKB;
    constexpr size_t wel = B_KC ? (size_t)128 * (K + 16) : (size_t)K * (128 + 16);
    const bool fstage = (p.epi.flags & (EDGL_EPI_ACCUM | EDGL_EPI_MUL_DGELU)) && !(p.epi.flags & EDGL_EPI_OUT_F32);
    const size_t smem = wel * sizeof(bf16) + 128 * sizeof(float) +
                        (fstage ? (size_t)8 * 16 * (64 + 4) * sizeof(float) : (size_t)8 * 16 * (64 + 8) * sizeof(bf16));
    if (smem > 160 * 1024) return 0;
    auto k = strip_gemm_kernel<NKB, B_KC>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const int nparts = (p.N + 127) / 128;
    const int per_cu = std::max(1, (int)((160 * 1024) / smem));
    const int nstrip8 = (p.M + 255) / 256;
    const int gx = std::max(1, std::min(nstrip8, (256 * std::min(per_cu, 2)) / nparts));
    hipLaunchKernelGGL(k, dim3(gx, nparts), dim3(W_NT), smem, st, p);
    EDGL_LAUNCH_CHECK();
    return 1;
}

// returns 1 if the fast path was taken, 0 if the shape does not qualify, <0 on error
int edgl_gemm2_try_strip(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int b_kc,
                         const float* bias, void* aux, int flags, hipStream_t st) {
    const bool ok = (K % 32 == 0) && K >= 32 && (N % 64 == 0) && (lda % 8 == 0) && (ldb % 8 == 0) &&
                    (ldc % 4 == 0) && (((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0 &&
                    (!aux || ((uintptr_t)aux & 7) == 0) && (!bias || ((uintptr_t)bias & 15) == 0);
    if (!ok) return 0;
    static const int dbg = getenv("EDGL_DBG") ? atoi(getenv("EDGL_DBG")) : 0;
    StripP p{(const bf16*)A, (const bf16*)B, C, M, N, K, lda, ldb, ldc, EpiP{bias, aux, flags}, dbg, 0};
    int rc;
    // wide projections with a plain (bias-only) epilogue: the 128 x 128 tiled kernel
    static const int use_tile = getenv("EDGL_GEMM_TILE") ? atoi(getenv("EDGL_GEMM_TILE")) : 1;
    if (use_tile && M >= 4096 && N >= 256 && (N % G_BN) == 0 && (K % G_BK) == 0 && K >= 128 && (flags & ~EDGL_EPI_BIAS) == 0 &&
        (ldc % 8) == 0)
        return b_kc ? launch_tile_nn<true>(p, st) : launch_tile_nn<false>(p, st);
    // beyond the register strips (K <= 512): the tiled kernel with the general epilogue
    if (K > 512) {
        if (use_tile && M >= 1024 && (N % G_BN) == 0 && (K % G_BK) == 0)
            return b_kc ? launch_tile_nn<true, true>(p, st) : launch_tile_nn<false, true>(p, st);
        return 0;
    }
    static const int stream_min_n = getenv("EDGL_GEMM_STREAM_N") ? atoi(getenv("EDGL_GEMM_STREAM_N")) : 384;
    if (N >= stream_min_n && (K == 384 || K == 512) && M >= 4096) {   // A would be re-read by >= 3 column slices
        rc = 0;
        if (K == 384) rc = b_kc ? launch_stream<12, true>(p, st) : launch_stream<12, false>(p, st);
        else rc = b_kc ? launch_stream<16, true>(p, st) : launch_stream<16, false>(p, st);
        if (rc) return rc;
    }
#define STRIP_CASE(NKB)                                                             \
    case NKB: rc = b_kc ? launch_strip<NKB, true>(p, st) : launch_strip<NKB, false>(p, st); break;
    switch (K / 32) {
        STRIP_CASE(1) STRIP_CASE(2) STRIP_CASE(4) STRIP_CASE(8) STRIP_CASE(12) STRIP_CASE(16)
        default: rc = 0;
    }
#undef STRIP_CASE
    // shapes the strips decline (their weight slice + staging exceed the LDS, e.g. K = 512 with N = 1024 and an epilogue)
    if (rc == 0 && use_tile && M >= 1024 && (N % G_BN) == 0 && (K % G_BK) == 0 && K >= 128)
        return b_kc ? launch_tile_nn<true, true>(p, st) : launch_tile_nn<false, true>(p, st);
    return rc;
}

// C[Kf,N] = X^T . Y (f32, overwritten or accumulated); if dbias != nullptr also dbias[N] = colsum(Y).
// workspace floats: edgl_gemm2_tn_workspace(R, Kf, N).
static int tn_tile(int Kf, int N) {
    // 128-tiles by default.  64-tiles (EDGL_TN_TILE=64, experiment) quarter the partial slabs of the 128 x 128 layers but
    // every operand column block is then read by twice as many workgroups: measured +18 us in the GEMMs for -12 us in
    // the slab reduction.
    (void)Kf; (void)N;
    static const int force = getenv("EDGL_TN_TILE") ? atoi(getenv("EDGL_TN_TILE")) : 0;
    return force == 64 ? 64 : 128;
}
// Row splits of a TN product (or of a group of them: `tiles` output tiles in all, `out_elems` f32 of output in all, `flop` =
// 2 R sum(Kf N)).  The chip takes workgroups in rounds of its CU count: 192 tiles x 2 splits = 384 workgroups run as TWO
// rounds (the 512-unit QKVT weight gradient: 229 us), x 4 = 768 as three of half the length (143 us) — but every split
// leaves a [Kf+1, N] f32 slab that a second kernel sums (384 workgroups over a 128 x 128 output meant 25 MB of partials for a
// 64 KB result).  Both sides priced (rates measured on the part: ~0.6 PFLOP/s of this kernel when every round is full, 3 us
// of prologue / slab write per workgroup, x1.25 below two resident workgroups per CU, ~4 TB/s of slab reduction), the
// cheapest split count wins: 4 for that product (768 workgroups), 28 for the headline's grouped launch (504, as before).
static int tn_choose_splits(int R, int tiles, double flop, double out_elems, int cap) {
    static const int forced = getenv("EDGL_TN_TARGET") ? atoi(getenv("EDGL_TN_TARGET")) : 0;   // experiments: fixed workgroup target
    const int smax = std::max(1, std::min(cap, R / 128));
    if (forced > 0) return std::max(1, std::min(forced / std::max(1, tiles), smax));
    static const int cus = [] {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    int best = 1;
    double best_t = 1e30;
    for (int sp = 1; sp <= smax; ++sp) {
        const long wg = (long)tiles * sp;
        const long rounds = (wg + cus - 1) / cus;
        double t = (double)rounds * (flop / (double)wg / (0.6e15 / cus) + 3e-6);   // + a workgroup's fixed part (its 64 KB slab)
        if (4 * wg < 7L * cus) t *= 1.25;   // fewer than ~two workgroups per CU: nothing hides a workgroup's barriers
        if (sp > 1) t += (double)sp * out_elems * 4.0 / 4e12 + 3e-6;
        if (t < best_t * 0.999) { best_t = t; best = sp; }
    }
    return best;
}
static int tn_splits(int R, int Kf, int N) {
    const int tm = tn_tile(Kf, N);
    const int tiles = ((Kf + tm - 1) / tm) * ((N + tm - 1) / tm);
    return tn_choose_splits(R, tiles, 2.0 * R * Kf * N, (double)(Kf + 1) * N, 1 << 20);
}
long edgl_gemm2_tn_workspace(int R, int Kf, int N) { return (long)tn_splits(R, Kf, N) * (Kf + 1) * N; }

// reductions of one TN product's split slabs into (dW, db)
static int tn_reduce(const float* workspace, int splits, float* C, int Kf, int N, float* dbias, int accumulate, hipStream_t st) {
    if (dbias && dbias == C + (long)Kf * N)   // (dW, db) contiguous, as in the flat gradient arena: one reduction
        return edgl_reduce_rows(workspace, splits, (Kf + 1) * N, (long)(Kf + 1) * N, C, accumulate, st);
    int rc = edgl_reduce_rows(workspace, splits, Kf * N, (long)(Kf + 1) * N, C, accumulate, st);
    if (rc) return rc;
    if (dbias) rc = edgl_reduce_rows(workspace + (long)Kf * N, splits, N, (long)(Kf + 1) * N, dbias, accumulate, st);
    return rc;
}

// ---- deferred / grouped mode -------------------------------------------------------------------------------------------
struct TnQueued { TnP p; float* C; float* dbias; int accumulate, max_splits; };
thread_local bool g_tn_defer = false;
thread_local int g_tn_n = 0;
thread_local TnQueued g_tn_q[TN_MAX_JOBS];

static int tn_flush(hipStream_t st) {
    const int n = g_tn_n;
    g_tn_n = 0;
    if (n == 0) return EDGL_OK;
    TnGroupP g;
    g.n = n;
    int total_tiles = 0;
    for (int i = 0; i < n; ++i) {
        g.tiles_n[i] = (g_tn_q[i].p.N + 127) / 128;
        g.tiles_k[i] = (g_tn_q[i].p.Kf + 127) / 128;
        total_tiles += g.tiles_n[i] * g.tiles_k[i];
    }
    // one split count for the whole group (row ranges of similar length), capped by what each job's workspace was sized for
    static const int target = getenv("EDGL_TN_GROUP_TARGET") ? atoi(getenv("EDGL_TN_GROUP_TARGET")) : 0;
    int group_splits;
    if (target > 0) {
        group_splits = std::max(1, target / std::max(1, total_tiles));
    } else {
        double flop = 0.0, elems = 0.0;
        int rmin = 1 << 30;
        for (int i = 0; i < n; ++i) {
            const TnP& q = g_tn_q[i].p;
            flop += 2.0 * q.R * q.Kf * q.N;
            elems += (double)(q.Kf + 1) * q.N;
            rmin = std::min(rmin, q.R);
        }
        group_splits = tn_choose_splits(rmin, total_tiles, flop, elems, 1 << 20);
    }
    int splits[TN_MAX_JOBS], blocks = 0;
    for (int i = 0; i < n; ++i) {
        TnP& p = g_tn_q[i].p;
        int sp = std::max(1, std::min(std::min(group_splits, g_tn_q[i].max_splits), p.R / 128));
        const int rps = ((p.R + sp - 1) / sp + T_BR - 1) / T_BR * T_BR;
        sp = (p.R + rps - 1) / rps;
        p.rows_per_split = rps;
        splits[i] = sp;
        g.job[i] = p;
        g.blk0[i] = blocks;
        blocks += g.tiles_n[i] * g.tiles_k[i] * sp;
    }
    g.blk0[n] = blocks;
    g.xcd = xcd_on();
    const size_t tn_lds = (size_t)2 * T_BR * (128 + 16) * sizeof(bf16);
    hipFuncSetAttribute((const void*)tn_gemm_group_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)tn_lds);
    hipLaunchKernelGGL(tn_gemm_group_kernel, dim3(g.xcd ? xcd_grid(blocks) : blocks), dim3(T_NT), tn_lds, st, g);
    EDGL_LAUNCH_CHECK();
    for (int i = 0; i < n; ++i) {
        const TnP& p = g_tn_q[i].p;
        const int rc = tn_reduce(p.partial, splits[i], g_tn_q[i].C, p.Kf, p.N, g_tn_q[i].dbias, g_tn_q[i].accumulate, st);
        if (rc) return rc;
    }
    return EDGL_OK;
}

// on = 1: the following edgl_gemm_dw calls (bf16, 128-tiles) are queued; on = 0: they run as one grouped launch on `stream`
// followed by their slab reductions; on < 0: forget the queue.  The operands and workspaces of queued calls must stay
// untouched until the flush; a full queue flushes itself.
int edgl_gemm2_tn_defer(int on, hipStream_t st) {
    if (on < 0) { g_tn_n = 0; g_tn_defer = false; return EDGL_OK; }
    if (!on && g_tn_defer) {
        g_tn_defer = false;
        return tn_flush(st);
    }
    g_tn_defer = on != 0;
    return EDGL_OK;
}

int edgl_gemm2_try_tn(const void* X, const void* Y, float* C, int R, int Kf, int N, int ldx, int ldy, int ldc, float* dbias,
                      int accumulate, float* workspace, hipStream_t st) {
    const bool ok = (Kf % 8 == 0) && (N % 8 == 0) && (ldx % 8 == 0) && (ldy % 8 == 0) && ldc == N &&
                    (((uintptr_t)X | (uintptr_t)Y) & 15) == 0 && workspace;
    if (!ok) return 0;
    int splits = tn_splits(R, Kf, N);
    int rps = ((R + splits - 1) / splits + T_BR - 1) / T_BR * T_BR;
    splits = (R + rps - 1) / rps;
    TnP p{(const bf16*)X, (const bf16*)Y, R, Kf, N, ldx, ldy, rps, workspace, dbias ? 1 : 0, 0, 0, 0, 0};
    if (g_tn_defer && tn_tile(Kf, N) == 128) {
        if (g_tn_n == TN_MAX_JOBS) {
            const int rc = tn_flush(st);
            if (rc) return rc;
        }
        g_tn_q[g_tn_n++] = TnQueued{p, C, dbias, accumulate, splits};
        return 1;
    }
    if (tn_tile(Kf, N) == 64) {
        const size_t lds = (size_t)2 * T_BR * (64 + 16) * sizeof(bf16);
        hipFuncSetAttribute((const void*)tn_gemm_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        p.tiles_n = (N + 63) / 64; p.tiles_k = (Kf + 63) / 64; p.nblocks = p.tiles_n * p.tiles_k * splits; p.xcd = xcd_on();
        hipLaunchKernelGGL(tn_gemm_kernel<64>, dim3(p.xcd ? xcd_grid(p.nblocks) : p.nblocks), dim3(T_NT), lds, st, p);
    } else {
        const size_t lds = (size_t)2 * T_BR * (128 + 16) * sizeof(bf16);
        hipFuncSetAttribute((const void*)tn_gemm_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        p.tiles_n = (N + 127) / 128; p.tiles_k = (Kf + 127) / 128; p.nblocks = p.tiles_n * p.tiles_k * splits; p.xcd = xcd_on();
        hipLaunchKernelGGL(tn_gemm_kernel<128>, dim3(p.xcd ? xcd_grid(p.nblocks) : p.nblocks), dim3(T_NT), lds, st, p);
    }
    EDGL_LAUNCH_CHECK();
    const int rc = tn_reduce(workspace, splits, C, Kf, N, dbias, accumulate, st);
    return rc ? rc : 1;
}

#ifdef EDGL_PHASE_TIMING
extern "C" int edgl_debug_phase_cycles_gemm(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(gemm2::g_phase_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(gemm2::g_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

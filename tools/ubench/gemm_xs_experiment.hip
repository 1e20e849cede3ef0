// EXPERIMENT, not part of the library (DESIGN.md §4.5 rule 45): the two wide projections of a block with the rows of A stationary
// in registers.  Measured 30.5 us for both products against 36.5 / 32.4 us of gemm2::tile_nn_kernel in isolation and +-0 inside the
// optimizer step; kept here with its ablation switches (XS_NOFLUSH / XS_NOOUT / XS_NOSTAGE / XS_NOMFMA / XS_NOXLOAD) for the record.
// To try it: copy to easydgl_amd/csrc/k_gemm_xs.hip, add it to build.py's SOURCES (flags -fno-slp-vectorize), declare
// edgl_gemm_xs_try in edgl_common.h and call it from edgl_gemm2_try_strip in front of the tiled kernel.
// "x-stationary" bf16 GEMM for the two wide projections of a block (BiMAU.__call__, temporal.py:409: QKVT = dense(queries, 4C)):
//     forward   C[M, 512] = A[M, 384] . W[384, 512] + bias        (W n-contiguous: transpose reads)
//     dX        C[M, 384] = A[M, 512] . W[384, 512]^T             (W k-contiguous: row-fragment reads)
// M = B*T rows is huge (51 712 at the headline), K and N are a few hundred: the product is a STREAM over the rows of A, and the
// 128 x 128 tiled kernel (k_gemm2.hip) pays a prologue and an epilogue per tile for only 6-8 K steps in between.  Here, as in the
// scoring kernels (k_score_strip.hip), a wave keeps its 64 rows of A in registers for the whole kernel — all K of them, as the B
// operands of v_mfma_f32_32x32x16_bf16, loaded straight into AGPRs — and the weights stream through LDS in units of 32 output
// columns:  D[z = column][x = row] = sum_k W(z, k) A[x][k]  — a lane ends up with 4 consecutive columns of one row (8-byte
// stores).  One workgroup = 4 waves = one wave per SIMD (512 registers) = 256 rows; an iteration = one unit = 2 K/16 MFMAs per
// wave, with the conversion / stores of the previous unit's results, the LDS stores of the next unit and the global loads of the
// one after it placed between the MFMAs: no phase of a wave waits for another.
#include <type_traits>

#include "edgl_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int v4i __attribute__((ext_vector_type(4)));

namespace xs {

constexpr int NTHR = 256, XW = 64, XB = 256, ZU = 32;

struct XsP {
    const bf16* A; const bf16* W; bf16* C; const float* bias;
    int M, N, lda, ldw, ldc;
};

#define XS_SPIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {   // one v_cvt_pk_bf16_f32 (RNE)
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
__device__ __forceinline__ v4i lds_b128(const char* p) { return *reinterpret_cast<const v4i*>(p); }
// A operand contracting along the ROWS of a [k][32 columns] image (64 bytes per row): two transpose reads, k-slots 0-3 <- rows
// +0..3, slots 4-7 <- rows +4..7 of this lane half's 8 rows — the order of a plain 16-byte row fragment of A
__device__ __forceinline__ v4i lds_tr(const char* p) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p + 4 * 64));
    const uint2 a = __builtin_bit_cast(uint2, v0), b = __builtin_bit_cast(uint2, v1);
    return v4i{(int)a.x, (int)a.y, (int)b.x, (int)b.y};
}

// MFMAs from asm with the register FILE chosen by constraint (k_score_strip.hip, rule 23 of DESIGN.md): results D in VGPRs (the
// VALU converts them), the A-row fragments XF in AGPRs (only ever a B operand), the weight fragment in VGPRs.
__device__ __forceinline__ void mfma_first(f32x16& d, const v4i& a, const v4i& b, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(d) : "v"(a), "a"(b), "v"(c));
}
__device__ __forceinline__ void mfma_first0(f32x16& d, const v4i& a, const v4i& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_acc(f32x16& d, const v4i& a, const v4i& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b));
}
// MFMA results read by compiler code: the reader may not be scheduled above this statement, and the wait states are inside it
__device__ __forceinline__ void settle(f32x16& d0, f32x16& d1) { asm volatile("s_nop 15\n\ts_nop 15" : "+v"(d0), "+v"(d1)); }

template <int OFF>
__device__ __forceinline__ void xload(v4i& dst, const bf16* p) {
    asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(dst) : "v"(p), "n"(OFF) : "memory");
}
template <int KS, int K0>
__device__ __forceinline__ void xload_row(v4i (&XF)[KS], const bf16* p) {
    if constexpr (K0 < KS) {
        xload<K0 * 32>(XF[K0], p);
        xload_row<KS, K0 + 1>(XF, p);
    }
}
template <int KS>
__device__ __forceinline__ void xwait(v4i (&XF)[2][KS]) {   // the asm loads above are invisible to the compiler's wait counts
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int xt = 0; xt < 2; ++xt)
#pragma unroll
        for (int ks = 0; ks < KS; ks += 8)
            asm volatile("" : "+a"(XF[xt][ks]), "+a"(XF[xt][ks + 1]), "+a"(XF[xt][ks + 2]), "+a"(XF[xt][ks + 3]), "+a"(XF[xt][ks + 4]),
                         "+a"(XF[xt][ks + 5]), "+a"(XF[xt][ks + 6]), "+a"(XF[xt][ks + 7]));
}

// KS = K / 16.  TR: W is [K][ldw] (n contiguous; forward) — unit image [K][32 columns], 64 bytes per row, transpose reads;
// otherwise W is [N][ldw] (k contiguous; dX) — unit image [32 rows][K + 8], row-fragment reads.
template <int KS, bool TR>
struct Cfg {
    static constexpr int K = 16 * KS;
    static constexpr int ROWB = TR ? 64 : K * 2 + 16;
    static constexpr int IMGB = TR ? K * 64 : ZU * ROWB;
    static constexpr int SLOTB = IMGB;
    static constexpr int NPC = (TR ? K * 4 : ZU * (K / 8)) / NTHR;   // 16-byte pieces per thread and unit
    static_assert((TR ? K * 4 : ZU * (K / 8)) % NTHR == 0, "unit pieces must divide over the workgroup");
    static constexpr int OUTB = 128 + 16;                      // output staging: [64 rows][64 columns] bf16 per wave, padded rows
    static constexpr int OUT0 = 2 * SLOTB;
    static constexpr int BIAS0 = OUT0 + 4 * XW * OUTB;         // bias of ALL columns (f32), once per workgroup
};

template <int KS, bool TR, bool BIAS>
__global__ __launch_bounds__(NTHR, 1) void xs_gemm_kernel(XsP p) {
    using G = Cfg<KS, TR>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, l31 = lane & 31;
    const int xbase = blockIdx.x * XB + wave * XW;
    const int NU = p.N / ZU;
    // per-lane LDS offsets of the weight fragments
    int zoff;
    if constexpr (TR) {
        const int s = lane & 15, Gq = lane >> 4;
        zoff = (8 * hi + (s >> 2)) * 64 + (16 * (Gq & 1) + 4 * (s & 3)) * 2;     // (+ ks * 16 * 64)
    } else {
        zoff = l31 * G::ROWB + hi * 16;                                          // (+ ks * 32)
    }
    // ---- the wave's 64 rows of A: AGPRs, for the whole kernel -----------------------------------------------------------------
    v4i XF[2][KS];
    {
        const bf16* x0 = p.A + (long)min(xbase + l31, p.M - 1) * p.lda + hi * 8;
        const bf16* x1 = p.A + (long)min(xbase + 32 + l31, p.M - 1) * p.lda + hi * 8;
#ifndef XS_NOXLOAD
        xload_row<KS, 0>(XF[0], x0);
        xload_row<KS, 0>(XF[1], x1);
#else
        xload<0>(XF[0][0], x0);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) asm volatile("" : "=a"(XF[xt][ks]));
#endif
    }
    // ---- staging of a unit: global -> registers -> LDS in single-instruction pieces placed between the MFMAs.  ONE register set:
    // a piece of unit u+1 goes to the LDS in iteration u and its registers are reloaded with the same piece of unit u+2 in the next
    // slot — loaded a whole iteration before it is stored.  (Named registers: a struct / array of them ends up in scratch.)
    static_assert(G::NPC <= 8, "staging registers");
    uint4 g0, g1, g2, g3, g4, g5, g6, g7;
    int st_n0 = 0;
    auto stage_begin = [&](int n0) { st_n0 = min(n0, p.N - ZU); };      // (behind the last unit: a harmless reload of it)
    auto load_piece = [&](int i) __attribute__((always_inline)) {
        if (i < G::NPC) {
            const int v = tid + NTHR * i;
            uint4 t;
            if constexpr (TR) t = *reinterpret_cast<const uint4*>(p.W + (long)(v >> 2) * p.ldw + st_n0 + (v & 3) * 8);
            else t = *reinterpret_cast<const uint4*>(p.W + (long)(st_n0 + v / (G::K / 8)) * p.ldw + (v % (G::K / 8)) * 8);
            if (i == 0) g0 = t; else if (i == 1) g1 = t; else if (i == 2) g2 = t; else if (i == 3) g3 = t;
            else if (i == 4) g4 = t; else if (i == 5) g5 = t; else if (i == 6) g6 = t; else g7 = t;
        }
    };
    auto store_piece = [&](char* slot, int i) __attribute__((always_inline)) {
        if (i < G::NPC) {
            const int v = tid + NTHR * i;
            const uint4 t = i == 0 ? g0 : i == 1 ? g1 : i == 2 ? g2 : i == 3 ? g3 : i == 4 ? g4 : i == 5 ? g5 : i == 6 ? g6 : g7;
            if constexpr (TR) *reinterpret_cast<uint4*>(slot + v * 16) = t;
            else *reinterpret_cast<uint4*>(slot + (v / (G::K / 8)) * G::ROWB + (v % (G::K / 8)) * 16) = t;
        }
    };
    stage_begin(0);
#pragma unroll
    for (int i = 0; i < G::NPC; ++i) load_piece(i);
#pragma unroll
    for (int i = 0; i < G::NPC; ++i) store_piece(smem, i);
    stage_begin(ZU);
#pragma unroll
    for (int i = 0; i < G::NPC; ++i) load_piece(i);
    if constexpr (BIAS) {
        for (int i = tid; i < p.N; i += NTHR) reinterpret_cast<float*>(smem + G::BIAS0)[i] = p.bias[i];
    }
    xwait<KS>(XF);
    lds_barrier();

    auto frag = [&](const char* slot, int ks) -> v4i {
        if constexpr (TR) return lds_tr(slot + zoff + ks * 16 * 64);
        else return lds_b128(slot + zoff + ks * 32);
    };
    auto bias_c = [&](int n0) -> f32x16 {     // C rows of the unit: column n0 + 8 g + 4 hi + j of register 4 g + j
        f32x16 ci;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(smem + G::BIAS0 + n0 * 4 + hi * 16 + g * 32);
            ci[4 * g] = t[0]; ci[4 * g + 1] = t[1]; ci[4 * g + 2] = t[2]; ci[4 * g + 3] = t[3];
        }
        return ci;
    };
    // results of a unit -> bf16 -> the wave's staging image (a lane holds 4 consecutive columns of one row: 8 bytes); every
    // second unit the image — 64 rows x 64 columns — leaves as 16 bytes per lane, 128 contiguous bytes per row (8-byte pieces
    // straight from the registers were 32 partial lines per store instruction: the kernel was bound by its store path)
    char* const wout = smem + G::OUT0 + wave * XW * G::OUTB;
    auto out_piece = [&](const f32x16 (&D)[2], int half, int pc) __attribute__((always_inline)) {
        const int xt = pc >> 2, g = pc & 3;
        const uint2 v = make_uint2(pack_bf16(D[xt][4 * g], D[xt][4 * g + 1]), pack_bf16(D[xt][4 * g + 2], D[xt][4 * g + 3]));
        *reinterpret_cast<uint2*>(wout + (32 * xt + l31) * G::OUTB + half * 64 + (8 * g + 4 * hi) * 2) = v;
    };
    // (rows past M: loads and stores are clamped to row M-1 — such a lane computes and writes that row's own values once more)
    // (two steps a slot apart: the wait for the LDS read would otherwise sit in front of the store, ~100 cycles per piece)
    uint4 fl0, fl1;
    auto flush_read = [&](int q) __attribute__((always_inline)) {
        const uint4 t = *reinterpret_cast<const uint4*>(wout + (8 * q + (lane >> 3)) * G::OUTB + (lane & 7) * 16);
        if (q & 1) fl1 = t; else fl0 = t;
    };
    auto flush_write = [&](int n0, int q) __attribute__((always_inline)) {
        const int r = 8 * q + (lane >> 3), cv = lane & 7;
#ifdef XS_NOFLUSH   // timing experiments (tools/build_variant.sh): bounds of the kernel's phases, wrong results
        if (p.M < 0)
#endif
        *reinterpret_cast<uint4*>(p.C + (long)min(xbase + r, p.M - 1) * p.ldc + n0 + cv * 8) = (q & 1) ? fl1 : fl0;
    };

    f32x16 Da[2], Db[2];
    constexpr int NSL = 2 * KS, STEP = (NSL - 8) / 8;
    // MFMAs of unit u (-> Dn) beside the conversion of unit u-1 (Dc) into half (u-1) & 1 of the staging image, the LDS stores of unit
    // u+1 and the loads of unit u+2.  The image of a unit pair (u-2, u-1), complete in the first half of an even iteration, leaves
    // SPREAD over the rest of that iteration (3 pieces) and the first half of the next one (5 pieces, in front of its conversions,
    // which overwrite half 0): all workgroups of the launch run in step, and eight stores per wave in eight consecutive slots were a
    // burst of 6.6 MB every second iteration that the kernel waited for.
    // MODE 0: first unit (nothing to convert); 1: odd unit, no image pending (u = 1); 2: even unit; 3: odd unit with a pending image
    constexpr int OS_E = 4, OS_O = (NSL - 28) / 8 >= 4 ? 4 : 2;
    auto iter = [&](f32x16 (&Dn)[2], const f32x16 (&Dc)[2], int u, auto mode) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode)::value;
        const char* slot = smem + (u & 1) * G::SLOTB;
        char* wslot = smem + ((u + 1) & 1) * G::SLOTB;
        f32x16 ci;
        if constexpr (BIAS) ci = bias_c(u * ZU);
        v4i zf[4];
        zf[0] = frag(slot, 0); zf[1] = frag(slot, 1); zf[2] = frag(slot, 2);
        const int n_pair = (MODE == 2 ? u - 2 : u - 3) * ZU, n_load = (u + 2) * ZU;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            if (ks + 3 < KS) zf[(ks + 3) & 3] = frag(slot, ks + 3);
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) {
#ifndef XS_NOMFMA
                if (ks == 0) { if constexpr (BIAS) mfma_first(Dn[xt], zf[0], XF[xt][0], ci); else mfma_first0(Dn[xt], zf[0], XF[xt][0]); }
                else mfma_acc(Dn[xt], zf[ks & 3], XF[xt][ks]);
                if constexpr (BIAS) { if (ks == 1 || ks == 2) asm volatile("" ::"v"(ci)); }
#else
                asm volatile("" : "+v"(Dn[xt]) : "v"(zf[ks & 3]));
#endif
                const int sl = 2 * ks + xt;
                // one piece of side work per slot
#ifndef XS_NOOUT
                if constexpr (MODE == 2) {          // conversions of the odd unit u-1 (half 1), then image pieces 0 .. 2
                    if (sl >= 2 && sl < 2 + 8 * OS_E && (sl - 2) % OS_E == 0) out_piece(Dc, 1, (sl - 2) / OS_E);
                    if (sl == NSL - 14 || sl == NSL - 9 || sl == NSL - 4) flush_read((sl - (NSL - 14)) / 5);
                    if (sl == NSL - 13 || sl == NSL - 8 || sl == NSL - 3) flush_write(n_pair, (sl - (NSL - 13)) / 5);
                } else if constexpr (MODE == 1 || MODE == 3) {   // image pieces 3 .. 7 (MODE 3), then conversions of the even unit u-1 (half 0)
                    if (MODE == 3 && sl >= 1 && sl <= 21 && (sl - 1) % 5 == 0) flush_read(3 + (sl - 1) / 5);
                    if (MODE == 3 && sl >= 2 && sl <= 22 && (sl - 2) % 5 == 0) flush_write(n_pair, 3 + (sl - 2) / 5);
                    if (sl >= 26 && sl < 26 + 8 * OS_O && (sl - 26) % OS_O == 0) out_piece(Dc, 0, (sl - 26) / OS_O);
                }
#endif
#ifndef XS_NOSTAGE
                if (sl >= 5 && (sl - 5) % STEP == 0 && (sl - 5) / STEP < G::NPC) store_piece(wslot, (sl - 5) / STEP);
                if (sl == 5) stage_begin(n_load);      // (behind the first store piece, which still belongs to unit u+1)
                if (sl >= 6 && (sl - 6) % STEP == 0 && (sl - 6) / STEP < G::NPC) load_piece((sl - 6) / STEP);
#endif
                XS_SPIN();
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    typedef std::integral_constant<int, 0> M0; typedef std::integral_constant<int, 1> M1;
    typedef std::integral_constant<int, 2> M2; typedef std::integral_constant<int, 3> M3;
    // N is a multiple of 64 and >= 128: an even number (>= 4) of units, even units in Da, odd ones in Db
    iter(Da, Da, 0, M0{});
    settle(Da[0], Da[1]);
    iter(Db, Da, 1, M1{});
    iter(Da, Db, 2, M2{});
#pragma clang loop unroll(disable)
    for (int u = 3; u + 1 < NU; u += 2) {
        iter(Db, Da, u, M3{});
        iter(Da, Db, u + 1, M2{});
    }
    iter(Db, Da, NU - 1, M3{});
    settle(Db[0], Db[1]);
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) out_piece(Db, 1, pc);
#pragma unroll
    for (int q = 0; q < 8; ++q) { flush_read(q); flush_write((NU - 2) * ZU, q); }
}

template <int KS, bool TR>
static int launch(const XsP& p, hipStream_t st) {
    using G = Cfg<KS, TR>;
    auto k = p.bias ? xs_gemm_kernel<KS, TR, true> : xs_gemm_kernel<KS, TR, false>;
    const int smem = G::BIAS0 + p.N * (int)sizeof(float);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(k, dim3((p.M + XB - 1) / XB), dim3(NTHR), smem, st, p);
    EDGL_LAUNCH_CHECK();
    return 1;
}

}  // namespace xs

// returns 1 if taken, 0 if the shape does not qualify (k_gemm2.hip: edgl_gemm2_try_strip asks first)
int edgl_gemm_xs_try(const void* A, const void* W, void* C, int M, int N, int K, int lda, int ldw, int ldc, int b_kc, const float* bias,
                     hipStream_t st) {
    static const int on = getenv("EDGL_GEMM_XS") ? atoi(getenv("EDGL_GEMM_XS")) : 1;
    if (!on || M < 4096 || (N % 64) != 0 || N < 128 || (lda % 8) || (ldw % 8) || (ldc % 4) || lda < K) return 0;
    if ((((uintptr_t)A | (uintptr_t)W) & 15) || ((uintptr_t)C & 7) || (bias && ((uintptr_t)bias & 3))) return 0;
    xs::XsP p{(const bf16*)A, (const bf16*)W, (bf16*)C, bias, M, N, lda, ldw, ldc};
    if (!b_kc && K == 384) return xs::launch<24, true>(p, st);
    if (b_kc && K == 512) return xs::launch<32, false>(p, st);
    return 0;
}

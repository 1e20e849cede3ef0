// K5 "strip" kernels: the two passes of the fused scoring / cross-entropy (EasyDGL.py:149-155,177-185) at the headline width
// (bf16, C = 128) in the one-wave-per-SIMD form.
//
//   ROLE_YF (x = compacted rows, z = items): one sweep over an item chunk computes the logits D[z][x] = Z[z].X[x] + bias[z], a running
//            row reference m (natural-log units, deferred: it only moves when a unit's maximum exceeds it by more than 8) and the
//            unnormalised row gradient  O[x] = sum_z exp(D[z][x] - m[x]) Z[z]  together with  l[x] = sum_z exp(D[z][x] - m[x]).
//   ROLE_W  (x = items, z = rows): with the row log-sum-exp known,  P[z][x] = coef[z] softmax(z)[x] = exp(D - lse'[z]),
//            lse' = lse - log coef;  O[x] = sum_z P[z][x] Z[z]  (d_table without the label term) and  sum_z P[z][x]  (d_bias
//            without the label term).  The label term  -coef[z] onehot(label[z])  is a scatter of R_w rows and is applied by
//            label_scatter_kernel after the slab reduction: no compare / select per logit in the product loop.
//
// Geometry: a workgroup = 4 waves = one wave per SIMD, 512 registers each; a wave owns 64 x vectors whose fragments (64
// registers) and whose [64 x 128] f32 accumulator (128 registers) never leave the register file; z streams through LDS in
// 64-row tiles (ring of 4, one barrier per tile), processed as 32-row units with v_mfma_f32_32x32x16_bf16:
//     S(u)  = Z(u) . X^T     16 MFMAs   (A = Z rows from LDS, ds_read_b128;  B = X fragments;  C = bias / -lse' per z row)
//     O    += P(u)^T . Z(u)  16 MFMAs   (A = P packed from the S registers in place;  B = Z through ds_read_b64_tr_b16)
// so every LDS operand fragment feeds two MFMAs (64 B / clk / CU of LDS reads at full MFMA rate, a quarter of the LDS peak).
// The three stages of a unit run one unit apart: iteration u issues S(u+1) and O(u-1) on the matrix pipe while the VALU turns
// S(u) into P(u) (one fma + one v_exp + one add per logit, one v_cvt_pk per pair), one logit per MFMA slot.
//
// LDS image of a tile: row z at byte z*512 + rot(z)*16, rot(z) = ((z&3)<<2) | ((z>>2)&3): both the row-fragment reads (32 rows
// x 16 B per half wave group) and the transpose reads (4 rows x 64 B per half wave) are bank-conflict free, and every
// address is one lane register + an immediate.
#include <atomic>
#include <cstdlib>

#include "edgl_common.h"
#include "score_plan.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int v4i __attribute__((ext_vector_type(4)));

namespace strip {

constexpr int NTHR = 256, XW = 64, XB = 256, ZT = 64, ZU = 32, C = 128;
constexpr int ROWB = 512;                   // LDS bytes per z row
constexpr int UNITB = ZU * ROWB;            // 16 KB
constexpr int TILEB = ZT * ROWB;            // 32 KB
constexpr int INFOB = ZT * 4;               // one float per z row: the C operand of its logit row
constexpr int SLOTB = TILEB + INFOB;
constexpr int NSLOT = 4;
constexpr int OSTR = 132;                   // floats per staged output row (epilogue)
constexpr int SMEM_LOOP = NSLOT * SLOTB, SMEM_EPI = 4 * XW * OSTR * 4;
constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;
constexpr float L2E = 1.4426950408889634f;
constexpr float THR2 = 8.0f * L2E;          // deferral threshold of the row reference, log2 units

enum { ROLE_YF = 0, ROLE_W = 1 };

struct StripP {
    const bf16* rows; const bf16* table; const float* out_bias;
    int R, I, i0, i1;
    const int32_t* nvalid;
    const float* coef; const float* row_lse;     // ROLE_W
    float* slabs; float* bias_slabs; float* part;
    float* acc_table; float* acc_bias;           // ROLE_W, optional: the outputs go into these zero-initialised arrays as f32 atomics (no slabs)
    int slab16;                                  // ROLE_YF: the row slabs are written as bf16 (the one-launch row finish reads them so)
    unsigned long long* stamps;     // -DSTRIP_TIMING builds only: [workgroup][8] shader-clock stamps of wave 0
};
#ifdef STRIP_TIMING
__device__ unsigned long long g_ph[48];
#define PHT(i) do { SPIN(); const unsigned long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t; ph_t = t_; SPIN(); } while (0)
#define STAMP(i) do { if (p.stamps && g.tid == 0) p.stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = __builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i)
#define PHT(i)
#endif

__device__ __forceinline__ int rot16(int z) { return (((z & 3) << 2) | ((z >> 2) & 3)) * 16; }

#define SPIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {   // one v_cvt_pk_bf16_f32 (RNE)
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){a, b}, bf16x2_t));
}
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }   // v_max3_f32

__device__ __forceinline__ v4i lds_b128(const char* p) { return *reinterpret_cast<const v4i*>(p); }
__device__ __forceinline__ f32x4 lds_f4(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
// B operand of a 32x32x16 MFMA contracting along the rows of the tile: two transpose reads (slots 0-3: rows +0..3, slots 4-7:
// rows +8..11 of this lane half's row group — the order in which P is packed from the logit registers)
__device__ __forceinline__ v4i lds_tr(const char* p) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p + 8 * ROWB + 32));
    const uint2 a = __builtin_bit_cast(uint2, v0), b = __builtin_bit_cast(uint2, v1);
    return v4i{(int)a.x, (int)a.y, (int)b.x, (int)b.y};
}

// Per-lane LDS offsets (bytes, relative to a unit's first row)
struct LaneOff {
    int zf;   // row-fragment read: row l&31, k-slot hi        (+ ks*32)
    int tr;   // transpose read: row 4hi + (s>>2), columns 16*(G&1) + 4*(s&3)   (+ ks2*16*ROWB + ct*64)
    int ci;   // C operand of the logit rows: info floats 4hi .. 4hi+3   (+ g*32)
};
__device__ __forceinline__ LaneOff lane_off(int lane) {
    LaneOff o;
    const int zr = lane & 31, hi = lane >> 5, G = lane >> 4, s = lane & 15;
    o.zf = zr * ROWB + rot16(zr) + hi * 16;
    const int tz = 4 * hi + (s >> 2);
    o.tr = tz * ROWB + rot16(tz) + (16 * (G & 1) + 4 * (s & 3)) * 2;
    o.ci = 4 * hi * 4;
    return o;
}

// global -> registers -> LDS staging of one 64-row tile (+ its per-row C operand), in single-instruction pieces (load_piece /
// store_piece) that the main loop places one per MFMA slot.  One wave per SIMD issues ~one instruction per 4-5 cycles, i.e.
// about six besides each 32-cycle MFMA: the loop is ISSUE bound, and every instruction of the staging counts.  Hence
//   * thread-constant offsets (row = tid/16 + 16 i keeps rot(row): the four pieces of a thread are 8 KB apart in LDS, 4 KB in
//     global memory): a piece is one min + one address add + the access;
//   * NO zero-filling of rows past the chunk / of table row 0: their C operand (-inf / -1000) makes every exponential of such a
//     row exactly 0 (exp2 underflows below -1000 log2 e for any finite reference), so the row's DATA only has to be finite —
//     the loads are clamped to the last valid row of the chunk.
template <int ROLE>
struct Stage {
    uint4 g0, g1, g2, g3;      // (named members, not an array: the array form ended up in scratch)
    float cinfo, cinfo2;
    int z0_, zend_;
    const bf16* Z_;
    const float* bias_; const float* coef_; const float* lse_;     // (by value: a pointer to the kernel's parameter struct would
    int I_, Reff_, tid_, row0_, goff_, loff_;                       //  put that struct — and every access to it — into scratch)
    __device__ __forceinline__ void init(const StripP& p, const bf16* Z, int zend, int Reff, int tid) {
        Z_ = Z; bias_ = p.out_bias; coef_ = p.coef; lse_ = p.row_lse; I_ = p.I; zend_ = zend; Reff_ = Reff; tid_ = tid;
        cinfo = 0.f; cinfo2 = 0.f; z0_ = 0;
        row0_ = tid >> 4;
        goff_ = (tid & 15) * 8;                                      // elements
        loff_ = row0_ * ROWB + rot16(row0_) + (tid & 15) * 16;       // bytes
    }
    __device__ __forceinline__ void begin(int z0) { z0_ = z0; }
    __device__ __forceinline__ void load_piece(int i) {      // i = 0..3: 16 bytes of the tile;  i = 4: the per-row scalars
        if (i < 4) {
            const int gz = max(min(z0_ + row0_ + 16 * i, zend_ - 1), 0);
            const uint4 v = *reinterpret_cast<const uint4*>(Z_ + (long)gz * C + goff_);
            if (i == 0) g0 = v; else if (i == 1) g1 = v; else if (i == 2) g2 = v; else g3 = v;
        } else {
            const int z = z0_ + (tid_ & (ZT - 1));      // every wave loads them (no divergent branch); wave 0 stores them
            if (ROLE == ROLE_YF) {
                cinfo = bias_[min(max(z, 1), I_ - 1) - 1];
            } else {
                const int gc = max(min(z, Reff_ - 1), 0);
                cinfo = coef_[gc]; cinfo2 = lse_[gc];
            }
        }
    }
    __device__ __forceinline__ void load(int z0) {
        begin(z0);
#pragma unroll
        for (int i = 0; i < 5; ++i) load_piece(i);
    }
    __device__ __forceinline__ void store_piece(char* slot, int i) {
#ifdef STRIP_T_WAITONLY    // (timing experiments only: the staged registers are waited for and consumed, nothing is written)
        if (z0_ > 128) { const uint4& g = i == 0 ? g0 : (i == 1 ? g1 : (i == 2 ? g2 : g3)); if (i < 4) asm volatile("" ::"v"(g.x), "v"(g.y), "v"(g.z), "v"(g.w)); else asm volatile("" ::"v"(cinfo), "v"(cinfo2)); return; }
#endif
        if (i < 4) {
            *reinterpret_cast<uint4*>(slot + loff_ + i * 16 * ROWB) = i == 0 ? g0 : (i == 1 ? g1 : (i == 2 ? g2 : g3));
        } else {
            const int z = z0_ + (tid_ & (ZT - 1));
            float c;
            if (ROLE == ROLE_YF) c = z >= zend_ ? -INFINITY : (z == 0 ? -1000.0f : cinfo);   // pad logit -1000 (Base.py:110)
            else c = (z < zend_ && cinfo > 0.f) ? __logf(cinfo) - cinfo2 : -INFINITY;        // -(lse - log coef)
            if (tid_ < ZT) reinterpret_cast<float*>(slot + TILEB)[tid_] = c;
        }
    }
    __device__ __forceinline__ void store(char* slot) {
#pragma unroll
        for (int i = 0; i < 5; ++i) store_piece(slot, i);
    }
};

struct Carry {            // operands fetched one iteration ahead: first three row fragments and the C rows of the next S unit
    v4i zf0, zf1, zf2;
    f32x16 ci;
};
__device__ __forceinline__ void fetch_ci(f32x16& ci, const char* info, const LaneOff& lo) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const f32x4 t = lds_f4(info + lo.ci + g * 32);
        ci[4 * g] = t[0]; ci[4 * g + 1] = t[1]; ci[4 * g + 2] = t[2]; ci[4 * g + 3] = t[3];
    }
}
__device__ __forceinline__ void fetch_carry(Carry& cy, const char* unit, const char* info, const LaneOff& lo) {
    cy.zf0 = lds_b128(unit + lo.zf);
    cy.zf1 = lds_b128(unit + lo.zf + 32);
    cy.zf2 = lds_b128(unit + lo.zf + 64);
    fetch_ci(cy.ci, info, lo);
}

// The MFMAs and the per-logit VALU work are asm statements, for two reasons.
// (1) Register FILES: with 512 registers per wave the compiler selects the AGPR form for every builtin MFMA and then moves each
//     logit through v_accvgpr_read before the VALU can touch it (144 moves per 32 MFMAs in the first build of this kernel;
//     -amdgpu-mfma-vgpr-form crashes hipcc 7.2 here).  Fixed here:  logits S in VGPRs (exponentiated in place), x fragments XF
//     in AGPRs (only ever an MFMA B operand; loaded straight into them), output O in AGPRs (only touched by MFMAs until the
//     epilogue), P and the Z fragments in VGPRs.
// (2) Placement: one wave per SIMD issues one instruction per ~4 cycles, so a 32-cycle MFMA hides ~7 other instructions and only
//     if they sit next to it.  IR passes otherwise sink the conversions to the end of the iteration and pack the row sums into
//     v_pk_add_f32 (slow beside MFMAs).  asm volatile statements keep their program order.
// Hazards (guide §5.7): the compiler neither sees nor pads an instruction inside asm.  Every consumer of an MFMA result here is
// either the next MFMA of the same accumulator chain (no wait states) or more than a full slot group later; a v_exp result is
// first read one slot later (no trans -> VALU forwarding hazard); the places that read MFMA results directly (prologue maxima,
// epilogue) sit behind settle_s() / settle_o().
#ifdef STRIP_SAFE
#define MFMA_PAD "\n\ts_nop 15\n\ts_nop 15"
#else
#define MFMA_PAD ""
#endif
__device__ __forceinline__ void mfma_s0(f32x16& d, const v4i& a, const v4i& b, const f32x16& c) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" MFMA_PAD : "=&v"(d) : "v"(a), "a"(b), "v"(c));
}
__device__ __forceinline__ void mfma_s(f32x16& d, const v4i& a, const v4i& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" MFMA_PAD : "+v"(d) : "v"(a), "a"(b));
}
__device__ __forceinline__ void mfma_o(f32x16& d, const v4i& a, const v4i& b) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" MFMA_PAD : "+a"(d) : "v"(a), "v"(b));
}
// Wait states before compiler code reads MFMA results.  The results are operands of the statement: a reader cannot be scheduled
// above it (a bare asm volatile orders against memory operations only — the first build read the prologue logits 4 instructions
// after their MFMA).
__device__ __forceinline__ void settle_s(f32x16& s0, f32x16& s1) { asm volatile("s_nop 15\n\ts_nop 15" : "+v"(s0), "+v"(s1)); }
__device__ __forceinline__ void settle_o(f32x16 (&O)[2][4]) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(O[0][0]), "+a"(O[0][1]), "+a"(O[0][2]), "+a"(O[0][3]), "+a"(O[1][0]), "+a"(O[1][1]),
                 "+a"(O[1][2]), "+a"(O[1][3]));
}

// One MFMA slot = ONE asm statement: the MFMA and the VALU work on logit e (0..15) of the x tile T being exponentiated; no
// instruction depends on a result of the same slot (one wave per SIMD: a dependent pair costs the full VALU latency), and separate
// statements would draw a compiler s_nop between them (an issue slot each).
//   T[e]   <- exp2(T[e])                      T[e] already holds  logit * log2(e) + add  (written one slot earlier)
//   T[e+1] <- T[e+1] * log2(e) + add          (e = 0 also scales itself first)
//   row sum += T[e-1];  after every odd logit the pair before it is packed
// slot_tail() finishes the tile (sum / pack of logits 14, 15).
#define VALU_E0 "v_fma_f32 %[cur], %[cur], %[l2e], %[add]\n\tv_fma_f32 %[nxt], %[nxt], %[l2e], %[add]\n\tv_exp_f32 %[cur], %[cur]"
#ifdef STRIP_T_NOSUM      // (timing experiment only: the row sums are not formed — bounds what a sum on the matrix pipe could give)
#define VALU_ODD "v_exp_f32 %[cur], %[cur]\n\tv_fma_f32 %[nxt], %[nxt], %[l2e], %[add]"
#else
#define VALU_ODD "v_exp_f32 %[cur], %[cur]\n\tv_fma_f32 %[nxt], %[nxt], %[l2e], %[add]\n\tv_add_f32 %[sum], %[sum], %[p1]"
#endif
#define VALU_EVEN VALU_ODD "\n\tv_cvt_pk_bf16_f32 %[pk], %[p2], %[p1]"
#define VALU_E15 "v_exp_f32 %[cur], %[cur]\n\tv_add_f32 %[sum], %[sum], %[p1]"
#define MF_S0 "v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], %[c]\n\t"
#define MF_S "v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], %[d]\n\t"
// kind 0: S MFMA with C = ci (D early-clobber VGPR), 1: S MFMA accumulating (D VGPR, B AGPR), 2: O MFMA (D AGPR, B VGPR)
template <int KIND>
__device__ __forceinline__ void slot(f32x16& d, const v4i& a, const v4i& b, const f32x16& c, f32x16& T, int (&pk)[8], float& lsum,
                                     float add, int e) {
#ifdef STRIP_NOVALU
    if (KIND == 0) mfma_s0(d, a, b, c); else if (KIND == 1) mfma_s(d, a, b); else mfma_o(d, a, b);
    if (e & 1) pk[e >> 1] = __builtin_bit_cast(int, T[e]);
    return;
#else
    float cur = T[e], nxt = T[e < 15 ? e + 1 : 15];
    int r = 0;
#define SLOT_ASM(MF, VA, DC, BC)                                                                                                  \
    asm volatile(MF VA : [d] DC(d), [cur] "+v"(cur), [nxt] "+v"(nxt), [sum] "+v"(lsum), [pk] "=&v"(r)                          \
                 : [a] "v"(a), [b] BC(b), [l2e] "s"(L2E), [add] "v"(add), [p1] "v"(T[e >= 1 ? e - 1 : 0]), [p2] "v"(T[e >= 2 ? e - 2 : 0]))
#define SLOT_ASM_C(MF, VA, DC, BC)                                                                                                \
    asm volatile(MF VA : [d] DC(d), [cur] "+v"(cur), [nxt] "+v"(nxt), [sum] "+v"(lsum), [pk] "=&v"(r)                          \
                 : [a] "v"(a), [b] BC(b), [c] "v"(c), [l2e] "s"(L2E), [add] "v"(add), [p1] "v"(T[e >= 1 ? e - 1 : 0]),             \
                   [p2] "v"(T[e >= 2 ? e - 2 : 0]))
    if (KIND == 0) {
        if (e == 0) SLOT_ASM_C(MF_S0, VALU_E0, "=&v", "a");
        else SLOT_ASM_C(MF_S0, VALU_ODD, "=&v", "a");          // (kind 0 occupies slots 0 and 1 only)
    } else if (KIND == 1) {
        if (e == 15) SLOT_ASM(MF_S, VALU_E15, "+v", "a");
        else if (e & 1) SLOT_ASM(MF_S, VALU_ODD, "+v", "a");
        else SLOT_ASM(MF_S, VALU_EVEN, "+v", "a");
    } else {
        if (e == 0) SLOT_ASM(MF_S, VALU_E0, "+a", "v");
        else if (e == 15) SLOT_ASM(MF_S, VALU_E15, "+a", "v");
        else if (e & 1) SLOT_ASM(MF_S, VALU_ODD, "+a", "v");
        else SLOT_ASM(MF_S, VALU_EVEN, "+a", "v");
    }
#undef SLOT_ASM_C
#undef SLOT_ASM
    T[e] = cur;
    if (e < 15) T[e + 1] = nxt;
    if (e >= 2 && (e & 1) == 0) pk[(e - 2) >> 1] = r;
#endif
}
__device__ __forceinline__ void slot_tail(f32x16& T, int (&pk)[8], float& lsum) {
#ifndef STRIP_NOVALU
    int r;
    asm volatile("v_add_f32 %0, %0, %2\n\tv_cvt_pk_bf16_f32 %1, %3, %2" : "+v"(lsum), "=&v"(r) : "v"(T[15]), "v"(T[14]));
    pk[7] = r;
#endif
}

// One pipeline iteration u: S(u+1) -> Sn, P(u) <- exp of Sc, O += P(u-1) . Z(u-1).
//   s_unit / o_unit: LDS rows of unit u+1 / unit u-1;  nx_unit / nx_info: unit u+2 (operands of the next iteration's first MFMAs).
//   STAGE: the staged tile is written to `st_slot` in the S half and the workgroup barrier sits between the halves.
//   LOAD : the loads of the tile after next are issued in the S half (stg.begin() called by the caller).
template <int ROLE, bool STAGE, bool LOAD>
__device__ __forceinline__ void unit_iter(f32x16 (&O)[2][4], const v4i (&XF)[2][8], f32x16 (&Sc)[2], f32x16 (&Sn)[2],
                                          v4i (&Pc)[2][2], const v4i (&Pp)[2][2], const float (&add)[2], float (&lsum)[2],
                                          Carry& cy, const char* s_unit, const char* o_unit, const char* nx_unit,
                                          const char* nx_info, const LaneOff& lo, Stage<ROLE>& stg, char* st_slot, int tid
#ifdef STRIP_TIMING
                                          , unsigned long long (&ph_acc)[12], unsigned long long& ph_t
#endif
) {
    PHT(11);
    // ---- S half: 16 MFMAs beside the 16 logits of x tile 0 --------------------------------------------------------------------
    // LDS operands are fetched THREE MFMA pairs ahead (rings of 4): with the four waves of the workgroup in lock step their reads
    // queue up behind each other, and two pairs (~130 cycles) did not cover it.
    v4i zf[4];
    zf[0] = cy.zf0; zf[1] = cy.zf1; zf[2] = cy.zf2;
    v4i tf[4];
    int pk[2][8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
#ifndef STRIP_T_NOZF
        if (ks + 3 < 8) zf[(ks + 3) % 4] = lds_b128(s_unit + lo.zf + (ks + 3) * 32);
#else
        if (ks == 0) zf[3] = cy.zf0;
#endif
#ifndef STRIP_T_NOTF
        if (ks >= 5) tf[ks - 5] = lds_tr(o_unit + lo.tr + (ks - 5) * 64);     // (tile u-1: resident long before this iteration)
#else
        if (ks == 6) { tf[0] = cy.zf0; tf[1] = cy.zf1; tf[2] = cy.zf0; tf[3] = cy.zf1; }
#endif
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            // (kind 0: slot 0 starts the x tile: VALU_E0; slot 1 is an odd slot)
            if (ks == 0) slot<0>(Sn[xt], zf[0], XF[xt][0], cy.ci, Sc[0], pk[0], lsum[0], add[0], xt);
            else slot<1>(Sn[xt], zf[ks % 4], XF[xt][ks], cy.ci, Sc[0], pk[0], lsum[0], add[0], 2 * ks + xt);
            // SrcC of the two ks = 0 MFMAs is read late in their passes: nothing may be allocated over `ci` until they are done
            // (the first build loaded the next row fragment over it: the x tile 1 logits came out with errors of 1e-3..1e-1) —
            // it stays an operand of an (empty) statement for two more MFMA pairs.
            if (ks == 1 || ks == 2) asm volatile("" ::"v"(cy.ci));
            const int sl = 2 * ks + xt;                          // one staging piece per slot
#ifndef STRIP_T_NOSTORE
            if (STAGE && sl >= 4 && sl <= 12 && (sl & 1) == 0) stg.store_piece(st_slot, (sl - 4) >> 1);
#endif
#ifndef STRIP_T_NOLOADS
            if (LOAD && sl >= 2 && sl < 7) stg.load_piece(sl - 2);
#endif
            SPIN();
            if (sl == 3) PHT(STAGE ? 0 : 4);
            if (sl == 12) PHT(STAGE ? 1 : 5);
        }
    }
    slot_tail(Sc[0], pk[0], lsum[0]);
    PHT(STAGE ? 2 : 6);
#ifndef STRIP_T_NOBAR
    // Workgroup barrier behind the staged stores.  Only the stores have to be complete: LDS operations retire in order, so
    // lgkmcnt(2) lets the two youngest ones — the transpose reads of tf[2], issued at ks = 7 behind the last store piece (slot 12)
    // — stay in flight; a full drain exposed their whole latency once per tile.  The count must never exceed the number of LDS
    // operations issued after the last store piece: 2.
    if (STAGE) asm volatile("s_waitcnt lgkmcnt(2)\n\ts_barrier" ::: "memory");
#endif
    SPIN();
    PHT(STAGE ? 3 : 7);
    // ---- O half: 16 MFMAs beside the 16 logits of x tile 1 --------------------------------------------------------------------
#pragma unroll
    for (int f = 0; f < 8; ++f) {          // f = ks2 * 4 + ct
        const int ks2 = f >> 2, ct = f & 3;
#ifndef STRIP_T_NOTF
        if (f + 3 < 8) tf[(f + 3) % 4] = lds_tr(o_unit + lo.tr + ((f + 3) >> 2) * 16 * ROWB + ((f + 3) & 3) * 64);
#endif
#ifndef STRIP_T_NOCARRY
        if (f == 4) cy.zf0 = lds_b128(nx_unit + lo.zf);
        if (f == 5) { fetch_ci(cy.ci, nx_info, lo); cy.zf1 = lds_b128(nx_unit + lo.zf + 32); }
        if (f == 6) cy.zf2 = lds_b128(nx_unit + lo.zf + 64);
#endif
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            slot<2>(O[xt][ct], Pp[xt][ks2], tf[f % 4], Sc[1], Sc[1], pk[1], lsum[1], add[1], 2 * f + xt);
            SPIN();
        }
    }
    slot_tail(Sc[1], pk[1], lsum[1]);
#pragma unroll
    for (int xt = 0; xt < 2; ++xt)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
            Pc[xt][k2] = v4i{pk[xt][4 * k2], pk[xt][4 * k2 + 1], pk[xt][4 * k2 + 2], pk[xt][4 * k2 + 3]};
    SPIN();
    PHT(STAGE ? 8 : 9);
}

// S-only sweep over the chunk (ROLE_YF fallback): the exact maximum of every row's logits, per lane (16 of the 32 rows of a unit)
__device__ __forceinline__ void max_unit(float (&mx)[2], const v4i (&XF)[2][8], const char* unit, const char* info, const LaneOff& lo) {
    f32x16 ci, S[2];
    fetch_ci(ci, info, lo);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const v4i zf = lds_b128(unit + lo.zf + ks * 32);
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            if (ks == 0) mfma_s0(S[xt], zf, XF[xt][0], ci);
            else mfma_s(S[xt], zf, XF[xt][ks]);
        }
    }
    settle_s(S[0], S[1]);
    asm volatile("" ::"v"(ci));
#pragma unroll
    for (int xt = 0; xt < 2; ++xt)
#pragma unroll
        for (int r = 0; r < 16; ++r) mx[xt] = fmaxf(mx[xt], S[xt][r]);
}

// Reference of the row exponentials (ROLE_YF).  A flash-style running maximum would have to rescale the [64 x 128] accumulator
// whenever it moves — code that touches O outside an MFMA, which drags the accumulators through VGPRs in every iteration (128
// v_accvgpr moves + spills in the first build).  Instead the reference of a row is FIXED per item chunk: the maximum of the
// chunk's first 32 logits.  exp(logit - ref) then exceeds 1 for larger logits, which f32 (and bf16: same exponent range, relative
// precision) absorbs up to 2^100; a row sum beyond that makes the WORKGROUP redo its chunk with the exact row maxima from an
// S-only sweep (attempt 1: exp <= 1, cannot overflow).  The finish kernels merge chunks from (reference, sum) pairs and do not
// care which reference a chunk used.
constexpr float LSUM_LIMIT = 1.2676506e30f;   // 2^100

struct Geo {     // per-wave geometry of a launch
    const bf16* Z;
    int tid, lane, wave, hi, l31;
    int Reff, xbase, xend, z_lo, z_hi, ntile, by, nchunk_dev;
    long slab_stride;
    LaneOff lo;
};

// x fragments X[x = 32 xt + l31][16 ks + 8 hi ..+7] straight into AGPRs.  Rows past the end (and the pad item 0 of ROLE_W, whose
// table row is not zero in memory) are NOT zeroed: their outputs are never stored and `add` = -inf / -1000 makes every exponential
// of theirs 0, so no other row sees them.
__device__ __forceinline__ void load_xfrags(v4i (&XF)[2][8], const bf16* X, const Geo& g) {
    const bf16* x0 = X + (long)max(min(g.xbase + g.l31, g.xend - 1), 0) * C + g.hi * 8;
    const bf16* x1 = X + (long)max(min(g.xbase + 32 + g.l31, g.xend - 1), 0) * C + g.hi * 8;
    asm volatile(
        "global_load_dwordx4 %0, %16, off\n\tglobal_load_dwordx4 %1, %16, off offset:32\n\t"
        "global_load_dwordx4 %2, %16, off offset:64\n\tglobal_load_dwordx4 %3, %16, off offset:96\n\t"
        "global_load_dwordx4 %4, %16, off offset:128\n\tglobal_load_dwordx4 %5, %16, off offset:160\n\t"
        "global_load_dwordx4 %6, %16, off offset:192\n\tglobal_load_dwordx4 %7, %16, off offset:224\n\t"
        "global_load_dwordx4 %8, %17, off\n\tglobal_load_dwordx4 %9, %17, off offset:32\n\t"
        "global_load_dwordx4 %10, %17, off offset:64\n\tglobal_load_dwordx4 %11, %17, off offset:96\n\t"
        "global_load_dwordx4 %12, %17, off offset:128\n\tglobal_load_dwordx4 %13, %17, off offset:160\n\t"
        "global_load_dwordx4 %14, %17, off offset:192\n\tglobal_load_dwordx4 %15, %17, off offset:224\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&a"(XF[0][0]), "=&a"(XF[0][1]), "=&a"(XF[0][2]), "=&a"(XF[0][3]), "=&a"(XF[0][4]), "=&a"(XF[0][5]), "=&a"(XF[0][6]),
          "=&a"(XF[0][7]), "=&a"(XF[1][0]), "=&a"(XF[1][1]), "=&a"(XF[1][2]), "=&a"(XF[1][3]), "=&a"(XF[1][4]), "=&a"(XF[1][5]),
          "=&a"(XF[1][6]), "=&a"(XF[1][7])
        : "v"(x0), "v"(x1)
        : "memory");
}

// One sweep of the wave's 64 x vectors over the workgroup's z chunk: O, lsum (and, ROLE_YF with !EXACT, the reference m2 / add
// from the chunk's first unit).
// LOADX: the x fragments are fetched here, behind the first tile loads (one memory round trip for both instead of two).
template <int ROLE, bool EXACT, bool LOADX>
__device__ __forceinline__ void main_pass(const StripP& p, const Geo& g, char* smem, const bf16* X, v4i (&XF)[2][8], f32x16 (&O)[2][4],
                                          float (&add)[2], float (&m2)[2], float (&lsum)[2]) {
    constexpr bool YS = ROLE == ROLE_YF;
    const int tid = g.tid;
    const LaneOff lo = g.lo;
#pragma unroll
    for (int xt = 0; xt < 2; ++xt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) O[xt][ct][r] = 0.f;
            // the zeros must sit in their AGPRs long before the first MFMA reads them (the compiler otherwise writes them with
            // v_accvgpr_write right in front of that MFMA: a write -> MFMA-read hazard it cannot see inside asm)
            asm volatile("" : "+a"(O[xt][ct]));
        }
    lsum[0] = 0.f; lsum[1] = 0.f;
    if (g.ntile == 0) {
        if (LOADX) load_xfrags(XF, X, g);
        return;
    }
    STAMP(1);
    // Two staging register sets: the loads of a tile are issued THREE iterations (~5 k cycles) before its LDS stores — with one set
    // and one iteration of distance every tile waited ~400 cycles for its data (global-load latency under this load > 1 us).
    Stage<ROLE> stg0, stg1;      // stg1: odd tiles, stg0: even tiles
    stg0.init(p, g.Z, g.z_hi, g.Reff, tid);
    stg1.init(p, g.Z, g.z_hi, g.Reff, tid);
    // ---- prologue: tile 0 (tiles 1 and 2 in flight), S(0) -------------------------------------------------------------------
    stg0.load(g.z_lo);
    stg1.load(g.z_lo + ZT);         // (rows past the chunk: clamped to its last row; their C operand is -inf)
    if (LOADX) load_xfrags(XF, X, g);     // waits for everything issued so far
    stg0.store(smem);
    stg0.load(g.z_lo + 2 * ZT);
    // iteration 0 multiplies P(-1) = 0 into "tile -1" = ring slot 3: its second unit must hold finite numbers
#pragma unroll
    for (int i = 0; i < 4; ++i) *reinterpret_cast<uint4*>(smem + 3 * SLOTB + UNITB + (tid + NTHR * i) * 16) = make_uint4(0, 0, 0, 0);
    lds_barrier();
    f32x16 Sa[2], Sb[2];
    v4i Pa[2][2], Pb[2][2];
#pragma unroll
    for (int xt = 0; xt < 2; ++xt)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            Pa[xt][k2] = v4i{0, 0, 0, 0}; Pb[xt][k2] = v4i{0, 0, 0, 0};
            asm volatile("" : "+v"(Pa[xt][k2]), "+v"(Pb[xt][k2]));
        }
    Carry cy;
    fetch_carry(cy, smem, smem + TILEB, lo);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const v4i zf = ks == 0 ? cy.zf0 : (ks == 1 ? cy.zf1 : (ks == 2 ? cy.zf2 : lds_b128(smem + lo.zf + ks * 32)));
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            if (ks == 0) mfma_s0(Sa[xt], zf, XF[xt][0], cy.ci);
            else mfma_s(Sa[xt], zf, XF[xt][ks]);
        }
    }
    settle_s(Sa[0], Sa[1]);
    asm volatile("" ::"v"(cy.ci));
    if (YS && !EXACT) {   // reference of the chunk: maximum of the row's first 32 logits
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            float t = Sa[xt][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) t = fmaxf(t, Sa[xt][r]);
            m2[xt] = fmaxf(t, __shfl_xor(t, 32, 64)) * L2E;
            add[xt] = -m2[xt];
        }
    }
    fetch_carry(cy, smem + UNITB, smem + TILEB + ZU * 4, lo);   // unit 1
    STAMP(2);
#ifdef STRIP_TIMING
    unsigned long long ph_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, ph_t = __builtin_readcyclecounter();
#endif
    // ---- main loop: two iterations per tile -------------------------------------------------------------------------------
    // (no unrolling / peeling: a peeled first trip gets its own register assignment for the accumulators, i.e. 192 v_accvgpr_mov
    // next to MFMAs that the compiler cannot pad)
#ifdef STRIP_TIMING
#define PH_ARGS , ph_acc, ph_t
#else
#define PH_ARGS
#endif
    // Tiles are processed in pairs (static names for the two staging sets); an odd tile count runs one all-padding tile (rows
    // past the chunk are zero-filled and carry C = -inf: their exponentials are 0).  edgl's planners hand out chunks of an even
    // number of tiles, so only the last chunk of a launch can be odd.
    const int ntile2 = (g.ntile + 1) & ~1;
#pragma clang loop unroll(disable)
    for (int t = 0; t < ntile2; t += 2) {
        char* s0 = smem + (t & 3) * SLOTB;            // tile t
        char* sm = smem + ((t + 3) & 3) * SLOTB;      // tile t-1 (t = 0: the zeroed unit of slot 3, P(-1) = 0)
        char* s1 = smem + ((t + 1) & 3) * SLOTB;      // tile t+1
        char* s2 = smem + ((t + 2) & 3) * SLOTB;      // tile t+2
        // iteration A (u = 2t): S(2t+1) from the second unit of tile t, O(2t-1) from the second unit of tile t-1; writes the
        // staged tile t+1 (behind the last tile: a harmless rewrite of stale rows — ONE code path, so that the accumulators
        // keep their registers: an if / else over two instances made the allocator shuffle 128 AGPRs per trip)
        unit_iter<ROLE, true, false>(O, XF, Sa, Sb, Pa, Pb, add, lsum, cy, s0 + UNITB, sm + UNITB, s1, s1 + TILEB, lo, stg1, s1, tid PH_ARGS);
        // iteration B (u = 2t+1): S(2t+2) from the first unit of tile t+1, O(2t) from tile t; issues the loads of tile t+3
        stg1.begin(g.z_lo + (t + 3) * ZT);
        unit_iter<ROLE, false, true>(O, XF, Sb, Sa, Pb, Pa, add, lsum, cy, s1, s0, s1 + UNITB, s1 + TILEB + ZU * 4, lo, stg1, s1, tid PH_ARGS);
        // the same for tile t+1: A writes tile t+2, B loads tile t+4
        unit_iter<ROLE, true, false>(O, XF, Sa, Sb, Pa, Pb, add, lsum, cy, s1 + UNITB, s0 + UNITB, s2, s2 + TILEB, lo, stg0, s2, tid PH_ARGS);
        stg0.begin(g.z_lo + (t + 4) * ZT);
        unit_iter<ROLE, false, true>(O, XF, Sb, Sa, Pb, Pa, add, lsum, cy, s2, s1, s2 + UNITB, s2 + TILEB + ZU * 4, lo, stg0, s2, tid PH_ARGS);
    }
    STAMP(3);
#ifdef STRIP_TIMING
    if (g.lane == 0 && blockIdx.x == 3 && blockIdx.y == 0) for (int i = 0; i < 12; ++i) g_ph[g.wave * 12 + i] = ph_acc[i];
#endif
    // ---- drain: O(2 ntile - 1) ----------------------------------------------------------------------------------------------
    {
        const char* o_unit = smem + ((ntile2 - 1) & 3) * SLOTB + UNITB;
#pragma unroll
        for (int f = 0; f < 8; ++f) {
            const v4i tf = lds_tr(o_unit + lo.tr + (f >> 2) * 16 * ROWB + (f & 3) * 64);
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) mfma_o(O[xt][f & 3], Pb[xt][f >> 2], tf);
        }
    }
    settle_o(O);
    STAMP(4);
}

// the wave's [64 x 128] accumulator -> LDS -> whole 512-byte rows of the slab; row sums / references
template <int ROLE>
__device__ __forceinline__ void epilogue(const StripP& p, const Geo& g, char* smem, const f32x16 (&O)[2][4], const float (&m2)[2],
                                         const float (&lsum)[2]) {
    constexpr bool YS = ROLE == ROLE_YF;
    __syncthreads();
    float* stg_o = reinterpret_cast<float*>(smem) + g.wave * XW * OSTR;
#pragma unroll
    for (int xt = 0; xt < 2; ++xt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                stg_o[(32 * xt + (r & 3) + 8 * (r >> 2) + 4 * g.hi) * OSTR + 32 * ct + g.l31] = O[xt][ct][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    float* slab = p.slabs + (long)g.by * g.slab_stride;
    if (!YS && p.acc_table) {
        // d_table without slabs: the row chunks of an item block add up in the gradient itself (edgl_score_flash_bwd_ex, bit 1 of
        // defer_label_term: the caller zero-filled d_table / d_bias; 3 chunks at the headline shape)
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const int xr = 2 * i + g.hi, gx = g.xbase + xr;
            const float4 v = *reinterpret_cast<const float4*>(stg_o + xr * OSTR + 4 * g.l31);
            if (gx < g.xend) {
                float* d = p.acc_table + (long)gx * C + 4 * g.l31;
                atomicAdd(d, v.x); atomicAdd(d + 1, v.y); atomicAdd(d + 2, v.z); atomicAdd(d + 3, v.w);
            }
        }
#pragma unroll
        for (int xt = 0; xt < 2; ++xt) {
            const float sb = lsum[xt] + __shfl_xor(lsum[xt], 32, 64);
            const int gx = g.xbase + 32 * xt + g.l31;
            if (g.hi == 0 && gx < g.xend && gx > 0) atomicAdd(p.acc_bias + gx - 1, sb);
        }
        return;
    }
    if (YS && p.slab16) {
        // bf16 slabs: half of the 33 MB burst that all workgroups write at once, and half of the row finish's read-back; the finish
        // rounds d_rows to bf16 anyway (one more rounding of each chunk's partial: 2^-9 relative before the weighted sum)
        bf16* slab_h = reinterpret_cast<bf16*>(p.slabs) + (long)g.by * g.slab_stride;
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const int xr = 2 * i + g.hi, gx = g.xbase + xr;
            const float4 v = *reinterpret_cast<const float4*>(stg_o + xr * OSTR + 4 * g.l31);
            const Frag4<bf16> h = frag_from_acc<bf16>(f32x4{v.x, v.y, v.z, v.w});
            if (gx < g.xend) *reinterpret_cast<uint2*>(slab_h + (long)gx * C + 4 * g.l31) = *reinterpret_cast<const uint2*>(&h);
        }
    } else {
#pragma unroll 4
        for (int i = 0; i < 32; ++i) {
            const int xr = 2 * i + g.hi, gx = g.xbase + xr;
            const float4 v = *reinterpret_cast<const float4*>(stg_o + xr * OSTR + 4 * g.l31);
            if (gx < g.xend) *reinterpret_cast<float4*>(slab + (long)gx * C + 4 * g.l31) = v;
        }
    }
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) {
        const float s = lsum[xt] + __shfl_xor(lsum[xt], 32, 64);
        const int gx = g.xbase + 32 * xt + g.l31;
        if (g.hi == 0 && gx < g.xend) {
            if (YS) {
                p.part[((long)gx * g.nchunk_dev + g.by) * 2] = m2[xt] * (1.0f / L2E);
                p.part[((long)gx * g.nchunk_dev + g.by) * 2 + 1] = s;
            } else if (gx > 0) {
                p.bias_slabs[(long)g.by * (p.I - 1) + gx - 1] = s;
            }
        }
    }
}

// ROLE_YF, rare: a row sum left the f32-safe range.  Exact row maxima over the whole chunk (S-only sweep), then the sweep again with
// them as references (exp <= 1).  NOT inlined: as part of the kernel body its live ranges cost the hot sweep 60 spilled registers
// and a copy of the x fragments per MFMA; as a function it has its own allocation and reloads what it needs.
__device__ __attribute__((noinline)) void fallback_exact(const StripP* pp, const Geo* gp, char* smem) {
    const StripP p = *pp;
    const Geo g = *gp;
    v4i XF[2][8];
    load_xfrags(XF, p.rows, g);
    Stage<ROLE_YF> stg;
    stg.init(p, g.Z, g.z_hi, g.Reff, g.tid);
    float mx[2] = {-INFINITY, -INFINITY};
    for (int t = 0; t < g.ntile; ++t) {
        __syncthreads();
        stg.load(g.z_lo + t * ZT);
        stg.store(smem);
        __syncthreads();
        max_unit(mx, XF, smem, smem + TILEB, g.lo);
        max_unit(mx, XF, smem + UNITB, smem + TILEB + ZU * 4, g.lo);
    }
    __syncthreads();
    float add[2], m2[2], lsum[2];
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) {
        m2[xt] = fmaxf(mx[xt], __shfl_xor(mx[xt], 32, 64)) * L2E;
        add[xt] = -m2[xt];
    }
    f32x16 O[2][4];
    main_pass<ROLE_YF, true, false>(p, g, smem, p.rows, XF, O, add, m2, lsum);
    epilogue<ROLE_YF>(p, g, smem, O, m2, lsum);
}

template <int ROLE>
__global__ __launch_bounds__(NTHR, 1) void strip_kernel(StripP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool YS = ROLE == ROLE_YF;
    Geo g;
    g.tid = threadIdx.x; g.lane = g.tid & 63; g.wave = g.tid >> 6; g.hi = g.lane >> 5; g.l31 = g.lane & 31;
    g.Reff = p.nvalid ? min(p.R, p.nvalid[0]) : p.R;
    int bx, zchunk;
    g.nchunk_dev = 1;
    if (YS) {
        const DevPlan dp = dev_plan(g.Reff, XB, gridDim.x, p.i1 - p.i0, 2 * ZT);     // chunks of whole tile pairs
        // XCD-aware order: workgroup b runs on XCD b % 8 (observed placement; speed only), and the nx workgroups of an item chunk
        // stream the SAME table rows — chunk-major ids are dealt to the XCDs in contiguous runs, so that one XCD's L2 holds 2-3
        // chunks (~1 MB) instead of seeing the whole table (5 MB > 4 MB: every tile came from the Infinity Cache)
        int id = blockIdx.x;
        if ((gridDim.x & 7) == 0) id = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
        if (id >= dp.nx * dp.nchunk || g.Reff <= 0) return;
        bx = id % dp.nx; g.by = id / dp.nx; zchunk = dp.zchunk; g.nchunk_dev = dp.nchunk;
        g.slab_stride = (long)dp.nx * XB * C;
    } else {
        bx = blockIdx.x; g.by = blockIdx.y;
        const int ntiles = (g.Reff + 2 * ZT - 1) / (2 * ZT);
        zchunk = (ntiles + (int)gridDim.y - 1) / (int)gridDim.y * 2 * ZT;
        g.slab_stride = (long)p.I * C;
    }
    g.xbase = (YS ? 0 : p.i0) + bx * XB + g.wave * XW;
    g.xend = YS ? g.Reff : p.i1;
    const bf16* X = YS ? p.rows : p.table;
    g.Z = YS ? p.table : p.rows;
    g.z_lo = (YS ? p.i0 : 0) + g.by * zchunk;
    g.z_hi = min(YS ? p.i1 : g.Reff, g.z_lo + zchunk);
    g.ntile = g.z_hi > g.z_lo ? (g.z_hi - g.z_lo + ZT - 1) / ZT : 0;
    g.lo = lane_off(g.lane);
    STAMP(0);
    v4i XF[2][8];
    float add[2], lsum[2], m2[2];
#pragma unroll
    for (int xt = 0; xt < 2; ++xt) {
        const int gx = g.xbase + 32 * xt + g.l31;
        // ROLE_W: logit + bias[x] rides in the exponent's fma; the pad item's logit is -1000 (Base.py:110), items past the shard give 0
        const float ob = YS ? 0.f : p.out_bias[min(max(gx, 1), p.I - 1) - 1];
        add[xt] = YS ? 0.f : (gx >= g.xend ? -INFINITY : (gx == 0 ? -1000.0f * L2E : ob * L2E));
        lsum[xt] = 0.f; m2[xt] = 0.f;
    }
    {
        f32x16 O[2][4];
        main_pass<ROLE, false, true>(p, g, smem, X, XF, O, add, m2, lsum);
        bool bad = false;
        if (YS) {
#pragma unroll
            for (int xt = 0; xt < 2; ++xt) {
                const float s = lsum[xt] + __shfl_xor(lsum[xt], 32, 64);
                bad = bad || ((g.xbase + 32 * xt + g.l31 < g.xend) && !(s < LSUM_LIMIT));
            }
        }
        if (!YS || !__syncthreads_or(bad ? 1 : 0)) {
            epilogue<ROLE>(p, g, smem, O, m2, lsum);
            STAMP(5);
#ifdef STRIP_TIMING
            if (p.stamps && g.tid == 0) p.stamps[(blockIdx.y * gridDim.x + blockIdx.x) * 8 + 6] = (unsigned long long)g.ntile;
#endif
            return;
        }
    }
#ifndef STRIP_TEST_NOFALLBACK
    if (YS) {   // copies: the structs the hot path reads must not be address-taken (they would live in scratch)
        const StripP p2 = p;
        const Geo g2 = g;
        fallback_exact(&p2, &g2, smem);
    }
#endif
}

// d_table[label[r]] -= coef[r] rows[r];  d_bias[label[r] - 1] -= coef[r]   over the weighted rows (label != 0): the one-hot part of
// dl = coef (p - onehot) (Appendix C) that the ROLE_W product pass leaves out.  A block = 32 rows x 128 channels: rows with equal
// labels are summed in LDS first, in row order by one thread per channel (hot items would otherwise serialise their atomics and
// the sums of a block are formed in a fixed order); the leaders' sums leave as f32 atomics.
__global__ __launch_bounds__(128) void label_scatter_kernel(const bf16* rows, const int64_t* labels, const float* coef,
                                                            const int32_t* nvalid, int R, int i0, int i1, const float* gscale,
                                                            float* d_table, float* d_bias) {
    constexpr int RB = 32;
    __shared__ float acc[RB][C];
    __shared__ float accb[RB];
    __shared__ int lab_s[RB], lead_s[RB];
    __shared__ float cf_s[RB];
    const int Reff = nvalid ? min(R, nvalid[0]) : R;
    const int r0 = blockIdx.x * RB, tid = threadIdx.x;
    if (r0 >= Reff) return;
    const float gs = gscale ? gscale[0] : 1.0f;
    float xv[RB];      // every row value of this thread's channel in flight before the first use
#pragma unroll
    for (int j = 0; j < RB; ++j) xv[j] = (float)rows[(long)min(r0 + j, Reff - 1) * C + tid];
    if (tid < RB) {
        const int r = r0 + tid;
        const int64_t lb = labels[min(r, Reff - 1)];
        const float cf = coef[min(r, Reff - 1)];
        const bool on = r < Reff && lb != 0 && lb >= i0 && lb < i1 && cf != 0.f;
        lab_s[tid] = on ? (int)lb : -1;
        cf_s[tid] = on ? cf * gs : 0.f;
        accb[tid] = 0.f;
    }
#pragma unroll
    for (int j = 0; j < RB; ++j) acc[j][tid] = 0.f;
    __syncthreads();
    if (tid < RB) {   // leader = first row of the block with the same label
        int lead = tid;
        const int lb = lab_s[tid];
        for (int j = tid - 1; j >= 0; --j)
            if (lab_s[j] == lb) lead = j;
        lead_s[tid] = lead;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < RB; ++j) acc[lead_s[j]][tid] += cf_s[j] * xv[j];     // column `tid` is private to this thread
    if (tid == 0)
        for (int j = 0; j < RB; ++j) accb[lead_s[j]] += cf_s[j];
    __syncthreads();
#pragma unroll 4
    for (int j = 0; j < RB; ++j)
        if (lab_s[j] >= 0 && lead_s[j] == j) atomicAdd(d_table + (long)lab_s[j] * C + tid, -acc[j][tid]);
    if (tid < RB && lab_s[tid] >= 0 && lead_s[tid] == tid) atomicAdd(d_bias + lab_s[tid] - 1, -accb[tid]);
}

}  // namespace strip

// ---- host side (called from k_score.hip) --------------------------------------------------------------------------------------
static unsigned long long* g_strip_stamps = nullptr;
extern "C" void edgl_debug_strip_stamps(void* buf) { g_strip_stamps = (unsigned long long*)buf; }   // STRIP_TIMING builds (tools/)
#ifdef STRIP_TIMING
extern "C" void edgl_debug_strip_phases(unsigned long long* out4) { hipMemcpyFromSymbol(out4, HIP_SYMBOL(strip::g_ph), 384); }
#endif
bool edgl_strip_enabled() {
    static const int on = getenv("EDGL_SCORE_STRIP") ? atoi(getenv("EDGL_SCORE_STRIP")) : 1;
    return on != 0;
}

// The dynamic-LDS attribute of a kernel is PER DEVICE: one flag per (kernel, device ordinal), set with an atomic so that two host
// threads driving different GPUs neither skip nor race it (a process-wide bool left the second device without the attribute).
static void strip_set_smem_attr(const void* kern, int which) {
    static std::atomic<uint64_t> done[2];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {   // unknown ordinal: set it on every launch
        hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, strip::SMEM);
        return;
    }
    const uint64_t bit = 1ull << dev;
    if (done[which].load(std::memory_order_acquire) & bit) return;
    hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, strip::SMEM);
    done[which].fetch_or(bit, std::memory_order_release);
}

int edgl_strip_rows(const void* rows, const void* table, const float* out_bias, int R, int I, int i0, int i1, const int32_t* nvalid,
                    float* slabs, float* part, int G, int slab16, hipStream_t st) {
    strip::StripP p{};
    p.slab16 = slab16;
    p.rows = (const bf16*)rows; p.table = (const bf16*)table; p.out_bias = out_bias; p.R = R; p.I = I; p.i0 = i0; p.i1 = i1;
    p.nvalid = nvalid; p.slabs = slabs; p.part = part; p.stamps = g_strip_stamps;
    auto k = strip::strip_kernel<strip::ROLE_YF>;
    strip_set_smem_attr((const void*)k, 0);
    hipLaunchKernelGGL(k, dim3(G), dim3(strip::NTHR), strip::SMEM, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

int edgl_strip_table(const void* rows, const void* table, const float* out_bias, const float* coef, const float* row_lse, int R,
                     int I, int i0, int i1, const int32_t* nvalid, float* slabs, float* bias_slabs, int nchunk, float* acc_table,
                     float* acc_bias, hipStream_t st) {
    strip::StripP p{};
    p.acc_table = acc_table; p.acc_bias = acc_bias;
    p.rows = (const bf16*)rows; p.table = (const bf16*)table; p.out_bias = out_bias; p.R = R; p.I = I; p.i0 = i0; p.i1 = i1;
    p.nvalid = nvalid; p.coef = coef; p.row_lse = row_lse; p.slabs = slabs; p.bias_slabs = bias_slabs; p.stamps = g_strip_stamps;
    auto k = strip::strip_kernel<strip::ROLE_W>;
    strip_set_smem_attr((const void*)k, 1);
    hipLaunchKernelGGL(k, dim3((i1 - i0 + strip::XB - 1) / strip::XB, nchunk), dim3(strip::NTHR), strip::SMEM, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

int edgl_strip_label_scatter(const void* rows, const int64_t* labels, const float* coef, const int32_t* nvalid, int R, int i0, int i1,
                             const float* gscale, float* d_table, float* d_bias, hipStream_t st) {
    hipLaunchKernelGGL(strip::label_scatter_kernel, dim3((R + 31) / 32), dim3(128), 0, st, (const bf16*)rows, labels, coef, nvalid, R,
                       i0, i1, gscale, d_table, d_bias);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

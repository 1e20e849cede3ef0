// K5 "wide strip" kernels: the two passes of the fused scoring / cross-entropy (EasyDGL.py:149-155,177-185) at the widths above the
// headline's — bf16, C = 256 (BASELINE.json configs[2]: 1 M items) — in the one-wave-per-SIMD form of k_score_strip.hip.
//
//   ROLE_YF (x = compacted rows, z = items) / ROLE_W (x = items, z = rows): the same two sweeps as strip::strip_kernel — logits
//   D[z][x] = Z[z].X[x] + c[z], P = exp(D - reference), O[x] += P^T Z, l[x] += sum_z P — with the same outputs (row slabs +
//   (reference, sum) pairs / table slabs + bias slabs), so every finishing kernel of k_score.hip serves both.
//
// What the width changes.  A wave's [x][C] f32 accumulator and its x fragments must stay in the register file: at C = 256 that is
// 32 x vectors per wave (128 + 64 registers, as 64 x vectors are at C = 128), i.e. ONE 32-column MFMA tile, 128 x vectors per
// workgroup.  A 32-row z unit then costs 16 + 16 MFMAs (v_mfma_f32_32x32x16_bf16) for 16 logits per lane instead of 32: half a
// logit per MFMA slot, so the VALU stream that bounds the C = 128 loop (54 cycles per 32-cycle MFMA, DESIGN rule 27) fits with
// room.  What becomes scarce instead is the way INTO the LDS: a unit (16 KB) feeds only 128 MFMAs, and staging it through
// registers (global_load -> ds_write_b128, the C = 128 kernel's form) measured 34 % of the kernel — 17 % the loads and their
// address arithmetic, 25 % the stores (13 cycles of the SIMD pair's LDS path each) — with the MFMA pipe at 55 %.  Hence:
//
// LDS-direct staging (global_load_lds_dwordx4): a wave instruction moves 64 x 16 bytes from per-lane global addresses to ONE
// contiguous KB of LDS at M0 — no staging registers, no ds_write, no compiler-visible load in the loop (every vmcnt is placed
// by hand).  The LDS image is built for that: BLOCK b = 0..15 of a unit holds rows b and b + 16 back to back (1 KB = one
// instruction: lanes 0-31 fetch row b, lanes 32-63 row b + 16) at byte b*1280 + rot(b)*16, rot(b) = ((b&3)<<2) | ((b>>2)&3) —
// rows b and b + 16 share rot, 1280 = 5 x 256 keeps the banks of the C = 128 layout (row-fragment reads of 16 rows and transpose
// reads of 4 rows x 64 bytes conflict free), every read address is one lane register + an immediate.  The per-row C operands
// (bias / -1000 / -inf;  log coef - lse) come from a small array a pre-kernel writes (info_kernel: -inf padded, so that no
// masking is left in the loop) through global_load_lds_dword.
//
// Geometry: 4 waves = one per SIMD, 512 registers each; z streams through LDS in 32-row units, ring of SEVEN: the loads of unit
// u+5 are issued in the second half of iteration u (behind the barrier: the slot of unit u-2 is free), unit u+2 must have
// landed at iteration u's barrier (s_waitcnt vmcnt(8): the two younger units stay in flight) — three iterations ~ 3 us of cover.
#include <atomic>
#include <cstdlib>

#include "edgl_common.h"
#include "score_plan.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef int v4i __attribute__((ext_vector_type(4)));

namespace stripw {

constexpr int NTHR = 256, XW = 32, XB = 128, ZU = 32, ZQ = 128;     // ZQ: chunk granularity of the planners
constexpr int INFOB = 256;                 // one global_load_lds_dword: 64 floats (the unit's 32 + the next unit's, unused)
#ifndef STRIPW_AHEAD
#define STRIPW_AHEAD 5
#endif
constexpr int AHEAD = STRIPW_AHEAD, NSLOT = AHEAD + 2;     // iteration u issues the loads of unit u + AHEAD; ring slots
static_assert(AHEAD >= 3 && AHEAD <= 5, "ring of 5 .. 7 units");
constexpr int CPAD = 512;                  // -inf entries behind the C-operand arrays (units past a chunk's end read them)
#ifndef STRIPW_PF
#define STRIPW_PF 6
#endif
#ifndef STRIPW_SPREAD
#define STRIPW_SPREAD 1
#endif
constexpr int PF = STRIPW_PF, RING = 8;    // operand prefetch distance (MFMA slots) / ring size
static_assert(PF >= 2 && PF < RING, "operand rings");
constexpr float L2E = 1.4426950408889634f;
constexpr float LSUM_LIMIT = 1.2676506e30f;   // 2^100

template <int CW>
struct W {
    static constexpr int C = CW;
    static constexpr int KS = CW / 16;                 // S MFMAs of a unit
    static constexpr int CT = CW / 32;                 // 32-channel tiles of the accumulator
    static constexpr int BLKB = 4 * CW + 256;          // LDS bytes per block = rows b, b + 16 (+ the rotation range)
    static constexpr int UNITB = 16 * BLKB;
    static constexpr int SLOTB = UNITB + INFOB;
    static constexpr int OSTR = CW + 4;                // floats per staged output row (epilogue)
    static constexpr int SMEM_LOOP = NSLOT * SLOTB, SMEM_EPI = 4 * XW * OSTR * 4;
    static constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;
    static_assert(SMEM <= 160 * 1024, "LDS");
    static_assert(KS == 16 && CT == 8 && 2 * CW * 2 == 1024, "the slot schedule and the one-KB blocks below are written for C = 256");
};

enum { ROLE_YF = 0, ROLE_W = 1 };

struct StripP {
    const bf16* rows; const bf16* table; const float* out_bias;
    int R, I, i0, i1;
    const int32_t* nvalid;
    const float* cinfo;       // C operand per z row, relative to the pass's first z (info_kernel; CPAD entries of -inf behind the range)
    int ncinfo;               // its length incl. the padding
    float* slabs; float* bias_slabs; float* part;
};

__device__ __forceinline__ int rot16(int z) { return (((z & 3) << 2) | ((z >> 2) & 3)) * 16; }
#define SPIN() __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ v4i lds_b128(const char* p) { return *reinterpret_cast<const v4i*>(p); }
__device__ __forceinline__ f32x4 lds_f4(const char* p) { return *reinterpret_cast<const f32x4*>(p); }
// B operand of a 32x32x16 MFMA contracting along the rows of the unit: two transpose reads (slots 0-3: rows +0..3, slots 4-7:
// rows +8..11 of this lane half's row group — the order in which P is packed from the logit registers)
template <int BLKB>
__device__ __forceinline__ v4i lds_tr(const char* p) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p + 8 * BLKB + 32));   // rows + 8: rot + 2
    const uint2 a = __builtin_bit_cast(uint2, v0), b = __builtin_bit_cast(uint2, v1);
    return v4i{(int)a.x, (int)a.y, (int)b.x, (int)b.y};
}

// Per-lane LDS offsets (bytes, relative to a unit's first block).  Row z of a unit: block z & 15, second half of it for z >= 16.
struct LaneOff {
    int zf;   // row-fragment read: row l&31, k-slot hi        (+ ks*32)
    int tr;   // transpose read: row 4hi + (s>>2), columns 16*(G&1) + 4*(s&3)   (+ ks2*512 + ct*64; rows + 8: lds_tr)
    int ci;   // C operand of the logit rows: info floats 4hi .. 4hi+3   (+ g*32)
};
template <int BLKB>
__device__ __forceinline__ LaneOff lane_off(int lane) {
    LaneOff o;
    const int zr = lane & 31, hi = lane >> 5, G = lane >> 4, s = lane & 15;
    o.zf = (zr & 15) * BLKB + rot16(zr & 15) + (zr >> 4) * 512 + hi * 16;
    const int tz = 4 * hi + (s >> 2);
    o.tr = tz * BLKB + rot16(tz) + (16 * (G & 1) + 4 * (s & 3)) * 2;
    o.ci = 4 * hi * 4;
    return o;
}

// LDS-direct staging of one 32-row unit: wave w moves blocks 4w .. 4w+3 (one instruction each: lanes 0-31 row b, lanes 32-63
// row b + 16; rows past the chunk are clamped to its last row — their C operand is -inf, the data only has to be finite), wave 0
// also the unit's C operands.  M0 is saved and restored inside each statement (the compiler owns it).
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
__device__ __forceinline__ void glds4(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
template <int CW>
struct Dma {
    using Cf = W<CW>;
    const char* Z_;           // first byte of the z operand
    const float* ci_;         // C operands, entry 0 = the pass's first z
    int zend_, zrel_, nci_, wave_, lrow_, lcol_, lane_;
    unsigned boff_[4];        // LDS offset of block 4 wave + j inside a unit (wave-uniform)
    int z0_;
    __device__ __forceinline__ void init(const StripP& p, const bf16* Z, int zend, int zfirst, int wave, int lane) {
        Z_ = reinterpret_cast<const char*>(Z); ci_ = p.cinfo; nci_ = p.ncinfo; zend_ = zend; zrel_ = zfirst; wave_ = __builtin_amdgcn_readfirstlane(wave); lane_ = lane;
        lrow_ = 16 * (lane >> 5); lcol_ = (lane & 31) * 16; z0_ = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int b = 4 * wave + j; boff_[j] = (unsigned)__builtin_amdgcn_readfirstlane(b * Cf::BLKB + rot16(b)); }
    }
    __device__ __forceinline__ void begin(int z0) { z0_ = z0; }
    __device__ __forceinline__ void piece(unsigned slot_lds, int k) {       // k = 0..3: one block;  k = 4: the C operands (wave 0)
#if defined(STRIPW_NOSTAGE)
        return;
#endif
        if (k < 4) {
            const int gz = max(min(z0_ + 4 * wave_ + k + lrow_, zend_ - 1), 0);
            glds16(Z_ + (long)gz * (CW * 2) + lcol_, slot_lds + boff_[k]);
        } else if (wave_ == 0) {
            const int gi = max(min(z0_ - zrel_ + lane_, nci_ - 1), 0);
            glds4(ci_ + gi, slot_lds + Cf::UNITB);
        }
    }
    __device__ __forceinline__ void issue(unsigned slot_lds, int z0) {
        begin(z0);
#pragma unroll
        for (int k = 0; k < 5; ++k) piece(slot_lds, k);
    }
};
// vmcnt(N): at most N of this wave's loads still in flight (they retire in order); never more than a wave without the C-operand
// load has issued since the unit waited for
#define VM_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")

__device__ __forceinline__ void fetch_ci_part(f32x16& ci, const char* info, const LaneOff& lo, int g) {
    const f32x4 t = lds_f4(info + lo.ci + g * 32);
    ci[4 * g] = t[0]; ci[4 * g + 1] = t[1]; ci[4 * g + 2] = t[2]; ci[4 * g + 3] = t[3];
}

// MFMAs and the per-logit VALU work as asm statements (register files and placement: see k_score_strip.hip).  Logits S in VGPRs
// (exponentiated in place), x fragments XF and the output O in AGPRs, P and the Z fragments in VGPRs.  Every consumer of an MFMA
// result is either the next MFMA of the same accumulator chain (no wait states) or more than a full slot group later; the places
// that read MFMA results directly (prologue maxima, epilogue) sit behind settle_s() / settle_o().
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
__device__ __forceinline__ void settle_s(f32x16& s0) { asm volatile("s_nop 15\n\ts_nop 15" : "+v"(s0)); }
__device__ __forceinline__ void settle_o(f32x16 (&O)[8]) {
    asm volatile("s_nop 15\n\ts_nop 15" : "+a"(O[0]), "+a"(O[1]), "+a"(O[2]), "+a"(O[3]), "+a"(O[4]), "+a"(O[5]), "+a"(O[6]), "+a"(O[7]));
}

// One S slot = ONE asm statement: the MFMA and the VALU work on logit e (0..15) of the unit being exponentiated (the stream of
// strip::slot: T[e] <- exp2(T[e]); T[e+1] scaled; row sum += T[e-1]; after every odd logit the pair before it is packed).
#define VALU_E0 "v_fma_f32 %[cur], %[cur], %[l2e], %[add]\n\tv_fma_f32 %[nxt], %[nxt], %[l2e], %[add]\n\tv_exp_f32 %[cur], %[cur]"
#define VALU_ODD "v_exp_f32 %[cur], %[cur]\n\tv_fma_f32 %[nxt], %[nxt], %[l2e], %[add]\n\tv_add_f32 %[sum], %[sum], %[p1]"
#define VALU_EVEN VALU_ODD "\n\tv_cvt_pk_bf16_f32 %[pk], %[p2], %[p1]"
#define VALU_E15 "v_exp_f32 %[cur], %[cur]\n\tv_add_f32 %[sum], %[sum], %[p1]"
#define MF_S0 "v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], %[c]\n\t"
#define MF_S "v_mfma_f32_32x32x16_bf16 %[d], %[a], %[b], %[d]\n\t"
// kind 0: S MFMA with C = ci (D early-clobber VGPR; logit 0), 1: S MFMA accumulating, 2: O MFMA (D AGPR, B VGPR)
template <int KIND>
__device__ __forceinline__ void slot(f32x16& d, const v4i& a, const v4i& b, const f32x16& c, f32x16& T, int (&pk)[8], float& lsum,
                                     float add, int e) {
    float cur = T[e], nxt = T[e < 15 ? e + 1 : 15];
    int r = 0;
#define SLOT_ASM(MF, VA, DC, BC)                                                                                               \
    asm volatile(MF VA : [d] DC(d), [cur] "+v"(cur), [nxt] "+v"(nxt), [sum] "+v"(lsum), [pk] "=&v"(r)                          \
                 : [a] "v"(a), [b] BC(b), [l2e] "s"(L2E), [add] "v"(add), [p1] "v"(T[e >= 1 ? e - 1 : 0]), [p2] "v"(T[e >= 2 ? e - 2 : 0]))
#define SLOT_ASM_C(MF, VA, DC, BC)                                                                                             \
    asm volatile(MF VA : [d] DC(d), [cur] "+v"(cur), [nxt] "+v"(nxt), [sum] "+v"(lsum), [pk] "=&v"(r)                          \
                 : [a] "v"(a), [b] BC(b), [c] "v"(c), [l2e] "s"(L2E), [add] "v"(add), [p1] "v"(T[e >= 1 ? e - 1 : 0]),             \
                   [p2] "v"(T[e >= 2 ? e - 2 : 0]))
    if (KIND == 0) {
        SLOT_ASM_C(MF_S0, VALU_E0, "=&v", "a");
    } else if (KIND == 1) {
        if (e == 15) SLOT_ASM(MF_S, VALU_E15, "+v", "a");
        else if (e & 1) SLOT_ASM(MF_S, VALU_ODD, "+v", "a");
        else SLOT_ASM(MF_S, VALU_EVEN, "+v", "a");
    } else {
        if (e == 15) SLOT_ASM(MF_S, VALU_E15, "+a", "v");
        else if (e & 1) SLOT_ASM(MF_S, VALU_ODD, "+a", "v");
        else SLOT_ASM(MF_S, VALU_EVEN, "+a", "v");
    }
#undef SLOT_ASM_C
#undef SLOT_ASM
    T[e] = cur;
    if (e < 15) T[e + 1] = nxt;
    if (e >= 2 && (e & 1) == 0) pk[(e - 2) >> 1] = r;
}
__device__ __forceinline__ void slot_tail(f32x16& T, int (&pk)[8], float& lsum) {
    int r;
    asm volatile("v_add_f32 %0, %0, %2\n\tv_cvt_pk_bf16_f32 %1, %3, %2" : "+v"(lsum), "=&v"(r) : "v"(T[15]), "v"(T[14]));
    pk[7] = r;
}

struct Carry {            // operands of the next iteration's first PF S slots and the C rows of its logits, fetched in the O half
    v4i zf[PF];
    f32x16 ci;
};

// One pipeline iteration u: S(u+1) -> Sn, P(u) <- exp of Sc, O += P(u-1) . Z(u-1).
//   s_unit / o_unit: LDS rows of unit u+1 / unit u-1;  nx_unit: unit u+2 — its loads (issued three iterations ago) must have landed
//   at the barrier between the halves, behind which the O half fetches the next iteration's first operands from it and issues the
//   loads of unit u + AHEAD into `ld_slot` (the slot of unit u-2: every wave is past its last read).
template <int CW>
__device__ __forceinline__ void unit_iter(f32x16 (&O)[8], const v4i (&XF)[16], f32x16& Sc, f32x16& Sn, v4i (&Pc)[2], const v4i (&Pp)[2],
                                          float add, float& lsum, Carry& cy, const char* s_unit, const char* o_unit, const char* nx_unit,
                                          const LaneOff& lo, Dma<CW>& dma, unsigned ld_slot) {
    using Cf = W<CW>;
    constexpr int BLKB = Cf::BLKB;
    v4i zf[RING], tf[RING];
#pragma unroll
    for (int i = 0; i < PF; ++i) zf[i] = cy.zf[i];
    int pk[8];
    // The 16 logits of P(u) are exponentiated beside every SECOND MFMA of the iteration's 32 (logit e in slot 2e).
    // ---- S half: 16 MFMAs ----------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        if (ks + PF < 16) zf[(ks + PF) % RING] = lds_b128(s_unit + lo.zf + (ks + PF) * 32);
        else { const int f = ks + PF - 16; tf[f % RING] = lds_tr<BLKB>(o_unit + lo.tr + (f >> 3) * 512 + (f & 7) * 64); }
#if STRIPW_SPREAD
        if (ks == 0) slot<0>(Sn, zf[0], XF[0], cy.ci, Sc, pk, lsum, add, 0);
        else if ((ks & 1) == 0) slot<1>(Sn, zf[ks % RING], XF[ks], cy.ci, Sc, pk, lsum, add, ks >> 1);
        else mfma_s(Sn, zf[ks % RING], XF[ks]);
#else
        if (ks == 0) slot<0>(Sn, zf[0], XF[0], cy.ci, Sc, pk, lsum, add, 0);
        else slot<1>(Sn, zf[ks % RING], XF[ks], cy.ci, Sc, pk, lsum, add, ks);
#endif
        // SrcC of the ks = 0 MFMA is read late in its passes: nothing may be allocated over `ci` until it is done (16 wait states:
        // the slots without a logit are two instructions long)
        if (ks >= 1 && ks <= 8) asm volatile("" ::"v"(cy.ci));
        SPIN();
    }
#if !STRIPW_SPREAD
    slot_tail(Sc, pk, lsum);
#endif
    // Unit u+2 landed (this wave's share: the loads of units u+3 and u+4 — 2 x 4 — may stay in flight), then the workgroup barrier:
    // everybody's share landed, and everybody is past the O half of iteration u-1 (the slot the loads below overwrite).
    // (LDS reads in flight cross the barrier freely: they target other slots.)
#ifndef STRIPW_NOBAR
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(4 * (AHEAD - 3)) : "memory");
#endif
    SPIN();
    // ---- O half: 16 MFMAs; operands of the next S half, the loads of unit u + AHEAD -----------------------------------------------
#pragma unroll
    for (int f = 0; f < 16; ++f) {          // f = ks2 * 8 + ct
        const int fn = f + PF;
        if (fn < 16) tf[fn % RING] = lds_tr<BLKB>(o_unit + lo.tr + (fn >> 3) * 512 + (fn & 7) * 64);
        else cy.zf[fn - 16] = lds_b128(nx_unit + lo.zf + (fn - 16) * 32);
        if (f >= 4 && f < 8) fetch_ci_part(cy.ci, nx_unit + Cf::UNITB, lo, f - 4);
        if (f < 10 && (f & 1)) dma.piece(ld_slot, f >> 1);
#if STRIPW_SPREAD
        if ((f & 1) == 0) slot<2>(O[f & 7], Pp[f >> 3], tf[f % RING], cy.ci, Sc, pk, lsum, add, 8 + (f >> 1));
        else mfma_o(O[f & 7], Pp[f >> 3], tf[f % RING]);
#else
        mfma_o(O[f & 7], Pp[f >> 3], tf[f % RING]);
#endif
        SPIN();
    }
#if STRIPW_SPREAD
    slot_tail(Sc, pk, lsum);
#endif
    Pc[0] = v4i{pk[0], pk[1], pk[2], pk[3]};
    Pc[1] = v4i{pk[4], pk[5], pk[6], pk[7]};
    SPIN();
}

struct Geo {     // per-wave geometry of a launch
    const bf16* Z;
    int tid, lane, wave, hi, l31;
    int Reff, xbase, xend, z_lo, z_hi, z_first, nunit, by, nchunk_dev;
    long slab_stride;
    LaneOff lo;
};

// x fragments X[x = l31][16 ks + 8 hi ..+7] straight into AGPRs (rows past the end are clamped, not zeroed: their outputs are
// never stored and `add` = -inf makes every exponential of theirs 0).  Ends with vmcnt(0): everything this wave issued has landed.
template <int CW>
__device__ __forceinline__ void load_xfrags(v4i (&XF)[16], const bf16* X, const Geo& g) {
    const bf16* x0 = X + (long)max(min(g.xbase + g.l31, g.xend - 1), 0) * CW + g.hi * 8;
    asm volatile(
        "global_load_dwordx4 %0, %16, off\n\tglobal_load_dwordx4 %1, %16, off offset:32\n\t"
        "global_load_dwordx4 %2, %16, off offset:64\n\tglobal_load_dwordx4 %3, %16, off offset:96\n\t"
        "global_load_dwordx4 %4, %16, off offset:128\n\tglobal_load_dwordx4 %5, %16, off offset:160\n\t"
        "global_load_dwordx4 %6, %16, off offset:192\n\tglobal_load_dwordx4 %7, %16, off offset:224\n\t"
        "global_load_dwordx4 %8, %16, off offset:256\n\tglobal_load_dwordx4 %9, %16, off offset:288\n\t"
        "global_load_dwordx4 %10, %16, off offset:320\n\tglobal_load_dwordx4 %11, %16, off offset:352\n\t"
        "global_load_dwordx4 %12, %16, off offset:384\n\tglobal_load_dwordx4 %13, %16, off offset:416\n\t"
        "global_load_dwordx4 %14, %16, off offset:448\n\tglobal_load_dwordx4 %15, %16, off offset:480\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&a"(XF[0]), "=&a"(XF[1]), "=&a"(XF[2]), "=&a"(XF[3]), "=&a"(XF[4]), "=&a"(XF[5]), "=&a"(XF[6]), "=&a"(XF[7]), "=&a"(XF[8]),
          "=&a"(XF[9]), "=&a"(XF[10]), "=&a"(XF[11]), "=&a"(XF[12]), "=&a"(XF[13]), "=&a"(XF[14]), "=&a"(XF[15])
        : "v"(x0)
        : "memory");
}

__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ int ring_next(int q, int k) { const int r = q + k; return r >= NSLOT ? r - NSLOT : r; }

// One sweep of the wave's 32 x vectors over the workgroup's z chunk: O, lsum (and, ROLE_YF with !EXACT, the reference m2 / add from
// the chunk's first unit).  LOADX: the x fragments are fetched here, between the first unit loads and their wait.
template <int ROLE, int CW, bool EXACT, bool LOADX>
__device__ __forceinline__ void main_pass(const StripP& p, const Geo& g, char* smem, const bf16* X, v4i (&XF)[16], f32x16 (&O)[8],
                                          float& add, float& m2, float& lsum) {
    using Cf = W<CW>;
    constexpr bool YS = ROLE == ROLE_YF;
    constexpr int SLOTB = Cf::SLOTB, UNITB = Cf::UNITB;
    const int tid = g.tid;
    const LaneOff lo = g.lo;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r) O[ct][r] = 0.f;
        // the zeros must sit in their AGPRs long before the first MFMA reads them (a write -> MFMA-read hazard hipcc cannot see inside asm)
        asm volatile("" : "+a"(O[ct]));
    }
    lsum = 0.f;
    if (g.nunit == 0) {
        if (LOADX) load_xfrags<CW>(XF, X, g);
        return;
    }
    Dma<CW> dma;
    dma.init(p, g.Z, g.z_hi, g.z_first, g.wave, g.lane);
    const unsigned lds0 = lds_addr(smem);
    // ---- prologue: units 0 .. AHEAD-1 on their way (unit v lives in ring slot v % 7), S(0) --------------------------------------
    // (the previous user of the LDS — an earlier sweep's epilogue or the fallback's probe — ended with a workgroup barrier)
#pragma unroll
    for (int v = 0; v < AHEAD; ++v) dma.issue(lds0 + v * SLOTB, g.z_lo + v * ZU);
    if (LOADX) load_xfrags<CW>(XF, X, g);     // (waits for everything issued so far)
    // iteration 0 multiplies P(-1) = 0 into "unit -1" = ring slot 6: it must hold finite numbers
#pragma unroll
    for (int i = 0; i < UNITB / 16 / NTHR; ++i) *reinterpret_cast<uint4*>(smem + (NSLOT - 1) * SLOTB + (tid + NTHR * i) * 16) = make_uint4(0, 0, 0, 0);
    VM_WAIT(4 * (AHEAD - 2));                 // units 0 and 1 (2, 3, 4 may stay in flight)
    lds_barrier();
    f32x16 Sa, Sb;
    v4i Pa[2], Pb[2];
#pragma unroll
    for (int k2 = 0; k2 < 2; ++k2) {
        Pa[k2] = v4i{0, 0, 0, 0}; Pb[k2] = v4i{0, 0, 0, 0};
        asm volatile("" : "+v"(Pa[k2]), "+v"(Pb[k2]));
    }
    Carry cy;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) fetch_ci_part(cy.ci, smem + UNITB, lo, gq);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const v4i zf = lds_b128(smem + lo.zf + ks * 32);
        if (ks == 0) mfma_s0(Sa, zf, XF[0], cy.ci);
        else mfma_s(Sa, zf, XF[ks]);
    }
    settle_s(Sa);
    asm volatile("" ::"v"(cy.ci));
    if (YS && !EXACT) {   // reference of the chunk: maximum of the row's first 32 logits
        float t = Sa[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) t = fmaxf(t, Sa[r]);
        m2 = fmaxf(t, __shfl_xor(t, 32, 64)) * L2E;
        add = -m2;
    }
    // carry for iteration 0: unit 1
#pragma unroll
    for (int i = 0; i < PF; ++i) cy.zf[i] = lds_b128(smem + SLOTB + lo.zf + i * 32);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) fetch_ci_part(cy.ci, smem + SLOTB + UNITB, lo, gq);
    // ---- main loop: two iterations per trip (static names for the two logit / P buffers) ------------------------------------------
    const int nunit2 = (g.nunit + 1) & ~1;
    int sl = 0;                         // ring slot of unit u
#pragma clang loop unroll(disable)
    for (int u = 0; u < nunit2; u += 2) {
        const int qm = ring_next(sl, NSLOT - 1), q1 = ring_next(sl, 1), q2 = ring_next(sl, 2), q3 = ring_next(sl, 3),
                  q5 = ring_next(sl, AHEAD), q6 = ring_next(sl, AHEAD + 1 >= NSLOT ? AHEAD + 1 - NSLOT : AHEAD + 1);
        // iteration u: S(u+1), O(u-1), carry from unit u+2; loads unit u+5 into its slot (= the slot of unit u-2)
        dma.begin(g.z_lo + (u + AHEAD) * ZU);
        unit_iter<CW>(O, XF, Sa, Sb, Pa, Pb, add, lsum, cy, smem + q1 * SLOTB, smem + qm * SLOTB, smem + q2 * SLOTB, lo, dma, lds0 + q5 * SLOTB);
        // iteration u+1: S(u+2), O(u), carry from unit u+3; loads unit u+6 (the slot of unit u-1)
        dma.begin(g.z_lo + (u + 1 + AHEAD) * ZU);
        unit_iter<CW>(O, XF, Sb, Sa, Pb, Pa, add, lsum, cy, smem + q2 * SLOTB, smem + sl * SLOTB, smem + q3 * SLOTB, lo, dma, lds0 + q6 * SLOTB);
        sl = q2;
    }
    // ---- drain: O(nunit2 - 1) ----------------------------------------------------------------------------------------------------
    {
        const char* o_unit = smem + ring_next(sl, NSLOT - 1) * SLOTB;
#pragma unroll
        for (int f = 0; f < 16; ++f) {
            const v4i tf = lds_tr<Cf::BLKB>(o_unit + lo.tr + (f >> 3) * 512 + (f & 7) * 64);
            mfma_o(O[f & 7], Pb[f >> 3], tf);
        }
    }
    settle_o(O);
    VM_WAIT(0);      // the loads issued for units past the chunk's end: nothing of this sweep may land in the LDS after it
}

// the wave's [32 x C] accumulator -> LDS -> whole rows of the slab; row sums / references
template <int ROLE, int CW>
__device__ __forceinline__ void epilogue(const StripP& p, const Geo& g, char* smem, const f32x16 (&O)[8], float m2, float lsum) {
    using Cf = W<CW>;
    constexpr bool YS = ROLE == ROLE_YF;
    constexpr int OSTR = Cf::OSTR;
    __syncthreads();
    float* stg_o = reinterpret_cast<float*>(smem) + g.wave * XW * OSTR;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) stg_o[((r & 3) + 8 * (r >> 2) + 4 * g.hi) * OSTR + 32 * ct + g.l31] = O[ct][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    float* slab = p.slabs + (long)g.by * g.slab_stride;
#pragma unroll 4
    for (int xr = 0; xr < XW; ++xr) {
        const int gx = g.xbase + xr;
        const float4 v = *reinterpret_cast<const float4*>(stg_o + xr * OSTR + 4 * g.lane);
        if (gx < g.xend) *reinterpret_cast<float4*>(slab + (long)gx * CW + 4 * g.lane) = v;
    }
    const float s = lsum + __shfl_xor(lsum, 32, 64);
    const int gx = g.xbase + g.l31;
    if (g.hi == 0 && gx < g.xend) {
        if (YS) {
            p.part[((long)gx * g.nchunk_dev + g.by) * 2] = m2 * (1.0f / L2E);
            p.part[((long)gx * g.nchunk_dev + g.by) * 2 + 1] = s;
        } else if (gx > 0) {
            p.bias_slabs[(long)g.by * (p.I - 1) + gx - 1] = s;
        }
    }
}

// S-only sweep over a resident unit (ROLE_YF fallback): the exact maximum of every row's logits, per lane (16 of the unit's 32 rows)
template <int CW>
__device__ __forceinline__ void max_unit(float& mx, const v4i (&XF)[16], const char* unit, const LaneOff& lo) {
    f32x16 ci, S;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) fetch_ci_part(ci, unit + W<CW>::UNITB, lo, gq);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        const v4i zf = lds_b128(unit + lo.zf + ks * 32);
        if (ks == 0) mfma_s0(S, zf, XF[0], ci);
        else mfma_s(S, zf, XF[ks]);
    }
    settle_s(S);
    asm volatile("" ::"v"(ci));
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[r]);
}

// ROLE_YF, rare: a row sum left the f32-safe range (the chunk's reference is the maximum of its FIRST 32 logits).  Exact row maxima
// over the whole chunk (S-only sweep), then the sweep again with them as references (exp <= 1).  Not inlined: own register allocation.
template <int CW>
__device__ __attribute__((noinline)) void fallback_exact(const StripP* pp, const Geo* gp, char* smem) {
    const StripP p = *pp;
    const Geo g = *gp;
    v4i XF[16];
    load_xfrags<CW>(XF, p.rows, g);
    Dma<CW> dma;
    dma.init(p, g.Z, g.z_hi, g.z_first, g.wave, g.lane);
    const unsigned lds0 = lds_addr(smem);
    float mx = -INFINITY;
    for (int u = 0; u < g.nunit; ++u) {
        __syncthreads();
        dma.issue(lds0, g.z_lo + u * ZU);
        VM_WAIT(0);
        __syncthreads();
        max_unit<CW>(mx, XF, smem, g.lo);
    }
    __syncthreads();
    float m2 = fmaxf(mx, __shfl_xor(mx, 32, 64)) * L2E;
    float add = -m2, lsum;
    f32x16 O[8];
    main_pass<ROLE_YF, CW, true, false>(p, g, smem, p.rows, XF, O, add, m2, lsum);
    epilogue<ROLE_YF, CW>(p, g, smem, O, m2, lsum);
}

template <int ROLE, int CW>
__global__ __launch_bounds__(NTHR, 1) void stripw_kernel(StripP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool YS = ROLE == ROLE_YF;
    Geo g;
    g.tid = threadIdx.x; g.lane = g.tid & 63; g.wave = __builtin_amdgcn_readfirstlane(g.tid >> 6); g.hi = g.lane >> 5; g.l31 = g.lane & 31;
    g.Reff = p.nvalid ? min(p.R, p.nvalid[0]) : p.R;
    int bx, zchunk;
    g.nchunk_dev = 1;
    if (YS) {
        const DevPlan dp = dev_plan(g.Reff, XB, gridDim.x, p.i1 - p.i0, ZQ);
        // XCD-aware order (see strip::strip_kernel): chunk-major ids dealt to the XCDs in contiguous runs — the x blocks of an item
        // chunk stream the same table rows out of one XCD's L2
        int id = blockIdx.x;
        if ((gridDim.x & 7) == 0) id = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
        if (id >= dp.nx * dp.nchunk || g.Reff <= 0) return;
        bx = id % dp.nx; g.by = id / dp.nx; zchunk = dp.zchunk; g.nchunk_dev = dp.nchunk;
        g.slab_stride = (long)dp.nx * XB * CW;
    } else {
        bx = blockIdx.x; g.by = blockIdx.y;
        const int nq = (g.Reff + ZQ - 1) / ZQ;
        zchunk = (nq + (int)gridDim.y - 1) / (int)gridDim.y * ZQ;
        g.slab_stride = (long)p.I * CW;
    }
    g.xbase = (YS ? 0 : p.i0) + bx * XB + g.wave * XW;
    g.xend = YS ? g.Reff : p.i1;
    const bf16* X = YS ? p.rows : p.table;
    g.Z = YS ? p.table : p.rows;
    g.z_lo = (YS ? p.i0 : 0) + g.by * zchunk;
    g.z_hi = min(YS ? p.i1 : g.Reff, g.z_lo + zchunk);
    g.nunit = g.z_hi > g.z_lo ? (g.z_hi - g.z_lo + ZU - 1) / ZU : 0;
    g.z_first = YS ? p.i0 : 0;
    g.lo = lane_off<W<CW>::BLKB>(g.lane);
    v4i XF[16];
    float add, lsum = 0.f, m2 = 0.f;
    {
        const int gx = g.xbase + g.l31;
        // ROLE_W: logit + bias[x] rides in the exponent's fma; the pad item's logit is -1000 (Base.py:110), items past the shard give 0
        const float ob = YS ? 0.f : p.out_bias[min(max(gx, 1), p.I - 1) - 1];
        add = YS ? 0.f : (gx >= g.xend ? -INFINITY : (gx == 0 ? -1000.0f * L2E : ob * L2E));
    }
    {
        f32x16 O[8];
        main_pass<ROLE, CW, false, true>(p, g, smem, X, XF, O, add, m2, lsum);
        bool bad = false;
        if (YS) {
            const float s = lsum + __shfl_xor(lsum, 32, 64);
            bad = (g.xbase + g.l31 < g.xend) && !(s < LSUM_LIMIT);
        }
#ifdef STRIPW_T_NOFALLBACK
        bad = false;
#endif
        if (!YS || !__syncthreads_or(bad ? 1 : 0)) {
            epilogue<ROLE, CW>(p, g, smem, O, m2, lsum);
            return;
        }
    }
    if (YS) {   // copies: the structs the hot path reads must not be address-taken (they would live in scratch)
        const StripP p2 = p;
        const Geo g2 = g;
        fallback_exact<CW>(&p2, &g2, smem);
    }
}

// ====================================================================================================================================
// C = 512 (the published recipe, runme.sh:15-23).  32 x-rows x 512 channels of f32 accumulator are every AGPR a wave has, so a workgroup
// owns ONE 256-channel half of the output and computes the logits (K = 512) for it: the halves of an x block are two workgroups that both
// form S — 32 + 16 MFMAs per unit instead of 32 + 32 / 2, 1.5 x the algorithmic products and two exponentials per logit (rule 68: bounded
// by 2/3 of what the C = 256 form reaches; the generic 8-wave kernels it replaces run at 0.13 of the MFMA peak).  x fragments (128) and the
// accumulator (128) fill the AGPRs.  A unit is 32 rows x 1 KB: block b = rows b and b + 16, one global_load_lds_dwordx4 each, at byte
// b*2304 + rot(b)*16 (2304 = 9 x 256: the banks of the other images) — 36 KB, so the ring holds FOUR units and the pipeline is one unit
// shorter than at C = 256: iteration u forms S(u+1) beside the exponentials of S(u) (one per second slot of 32), and its O half multiplies
// the P(u) it has just packed into unit u — units u, u+1, u+2 are alive, unit u+3 is issued behind the barrier into the slot of unit u-1.
template <int DUMMY = 0>
struct W5 {
    static constexpr int CW = 512, CO = 256, KS = 32;
    static constexpr int BLKB = 2 * 1024 + 256;
    static constexpr int UNITB = 16 * BLKB;
    static constexpr int SLOTB = UNITB + INFOB;
    static constexpr int NSLOT5 = 4;
    static constexpr int OSTR = CO + 4;
    static constexpr int SMEM_LOOP = NSLOT5 * SLOTB, SMEM_EPI = 4 * XW * OSTR * 4;
    static constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;
    static_assert(SMEM <= 160 * 1024, "LDS");
};
using C5 = W5<0>;

struct Geo5 {
    const bf16* Z;
    int tid, lane, wave, hi, l31;
    int Reff, xbase, xend, z_lo, z_hi, z_first, nunit, by, nchunk_dev, half;
    long slab_stride;
    LaneOff lo;
};
__device__ __forceinline__ LaneOff lane_off5(int lane, int half) {
    LaneOff o;
    const int zr = lane & 31, hi = lane >> 5, G = lane >> 4, s = lane & 15;
    o.zf = (zr & 15) * C5::BLKB + rot16(zr & 15) + (zr >> 4) * 1024 + hi * 16;
    const int tz = 4 * hi + (s >> 2);
    o.tr = tz * C5::BLKB + rot16(tz) + (16 * (G & 1) + 4 * (s & 3)) * 2 + half * (C5::CO * 2);
    o.ci = 4 * hi * 4;
    return o;
}
struct Dma5 {      // wave w moves blocks 4w .. 4w+3 of a unit, two instructions each (row b, row b + 16): pieces 0 .. 7; piece 8: the C operands
    const char* Z_;
    const float* ci_;
    int zend_, zrel_, nci_, wave_, lane_, z0_;
    unsigned boff_[4];
    __device__ __forceinline__ void init(const StripP& p, const bf16* Z, int zend, int zfirst, int wave, int lane) {
        Z_ = reinterpret_cast<const char*>(Z); ci_ = p.cinfo; nci_ = p.ncinfo; zend_ = zend; zrel_ = zfirst;
        wave_ = __builtin_amdgcn_readfirstlane(wave); lane_ = lane; z0_ = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int b = 4 * wave + j; boff_[j] = (unsigned)__builtin_amdgcn_readfirstlane(b * C5::BLKB + rot16(b)); }
    }
    __device__ __forceinline__ void begin(int z0) { z0_ = z0; }
    __device__ __forceinline__ void piece(unsigned slot_lds, int k) {
        if (k < 8) {
            const int gz = max(min(z0_ + 4 * wave_ + (k >> 1) + 16 * (k & 1), zend_ - 1), 0);      // (wave-uniform)
            glds16(Z_ + (long)gz * (C5::CW * 2) + lane_ * 16, slot_lds + boff_[k >> 1] + (k & 1) * 1024);
        } else if (wave_ == 0) {
            const int gi = max(min(z0_ - zrel_ + lane_, nci_ - 1), 0);
            glds4(ci_ + gi, slot_lds + C5::UNITB);
        }
    }
    __device__ __forceinline__ void issue(unsigned slot_lds, int z0) {
        begin(z0);
#pragma unroll
        for (int k = 0; k < 9; ++k) piece(slot_lds, k);
    }
};

// x fragments X[x = l31][16 ks + 8 hi ..+7], ks = 0 .. 31, straight into AGPRs (two statements: an asm takes 30 operands)
__device__ __forceinline__ void load_xfrags5(v4i (&XF)[32], const bf16* X, const Geo5& g) {
    const bf16* x0 = X + (long)max(min(g.xbase + g.l31, g.xend - 1), 0) * C5::CW + g.hi * 8;
#define EDGL_XF16(BASE, PTR)                                                                                                          \
    asm volatile(                                                                                                                     \
        "global_load_dwordx4 %0, %16, off\n\tglobal_load_dwordx4 %1, %16, off offset:32\n\t"                                          \
        "global_load_dwordx4 %2, %16, off offset:64\n\tglobal_load_dwordx4 %3, %16, off offset:96\n\t"                               \
        "global_load_dwordx4 %4, %16, off offset:128\n\tglobal_load_dwordx4 %5, %16, off offset:160\n\t"                             \
        "global_load_dwordx4 %6, %16, off offset:192\n\tglobal_load_dwordx4 %7, %16, off offset:224\n\t"                             \
        "global_load_dwordx4 %8, %16, off offset:256\n\tglobal_load_dwordx4 %9, %16, off offset:288\n\t"                             \
        "global_load_dwordx4 %10, %16, off offset:320\n\tglobal_load_dwordx4 %11, %16, off offset:352\n\t"                           \
        "global_load_dwordx4 %12, %16, off offset:384\n\tglobal_load_dwordx4 %13, %16, off offset:416\n\t"                           \
        "global_load_dwordx4 %14, %16, off offset:448\n\tglobal_load_dwordx4 %15, %16, off offset:480\n\t"                           \
        "s_waitcnt vmcnt(0)"                                                                                                          \
        : "=&a"(XF[BASE + 0]), "=&a"(XF[BASE + 1]), "=&a"(XF[BASE + 2]), "=&a"(XF[BASE + 3]), "=&a"(XF[BASE + 4]), "=&a"(XF[BASE + 5]),  \
          "=&a"(XF[BASE + 6]), "=&a"(XF[BASE + 7]), "=&a"(XF[BASE + 8]), "=&a"(XF[BASE + 9]), "=&a"(XF[BASE + 10]),                    \
          "=&a"(XF[BASE + 11]), "=&a"(XF[BASE + 12]), "=&a"(XF[BASE + 13]), "=&a"(XF[BASE + 14]), "=&a"(XF[BASE + 15])                 \
        : "v"(PTR)                                                                                                                    \
        : "memory")
    EDGL_XF16(0, x0);
    const bf16* x1 = x0 + 256;
    EDGL_XF16(16, x1);
#undef EDGL_XF16
}

// One iteration u: S half — S(u+1) -> Sn (32 MFMAs) beside the 16 exponentials of Sc = S(u) (logit e in slot 2e) -> P(u); barrier (unit u+2
// has landed: vmcnt(0), it is the youngest load); O half — O += P(u) . Z(u) (16 MFMAs), the first operands of the next S half from unit
// u+2, the loads of unit u+3 into `ld_slot`.
__device__ __forceinline__ void unit_iter5(f32x16 (&O)[8], const v4i (&XF)[32], f32x16& Sc, f32x16& Sn, float add, float& lsum, Carry& cy,
                                           const char* s_unit, const char* o_unit, const char* nx_unit, const LaneOff& lo, Dma5& dma,
                                           unsigned ld_slot) {
    constexpr int BLKB = C5::BLKB;
    v4i zf[RING], tf[RING];
#pragma unroll
    for (int i = 0; i < PF; ++i) zf[i] = cy.zf[i];
    int pk[8];
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        if (ks + PF < 32) zf[(ks + PF) % RING] = lds_b128(s_unit + lo.zf + (ks + PF) * 32);
        else { const int f = ks + PF - 32; tf[f % RING] = lds_tr<BLKB>(o_unit + lo.tr + (f >> 3) * 1024 + (f & 7) * 64); }
        if (ks == 0) slot<0>(Sn, zf[0], XF[0], cy.ci, Sc, pk, lsum, add, 0);
        else if ((ks & 1) == 0) slot<1>(Sn, zf[ks % RING], XF[ks], cy.ci, Sc, pk, lsum, add, ks >> 1);
        else mfma_s(Sn, zf[ks % RING], XF[ks]);
        if (ks >= 1 && ks <= 8) asm volatile("" ::"v"(cy.ci));      // SrcC of the ks = 0 MFMA: see unit_iter
        SPIN();
    }
    slot_tail(Sc, pk, lsum);
    v4i P[2];
    P[0] = v4i{pk[0], pk[1], pk[2], pk[3]};
    P[1] = v4i{pk[4], pk[5], pk[6], pk[7]};
    // unit u+2 landed (it is this wave's youngest load), everybody past the O half of iteration u-1; the s_nop keeps the wait states
    // between the last v_cvt_pk above and the first MFMA that reads P
    asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier\n\ts_nop 4" : "+v"(P[0]), "+v"(P[1]) : : "memory");
    SPIN();
#pragma unroll
    for (int f = 0; f < 16; ++f) {          // f = ks2 * 8 + ct
        const int fn = f + PF;
        if (fn < 16) tf[fn % RING] = lds_tr<BLKB>(o_unit + lo.tr + (fn >> 3) * 1024 + (fn & 7) * 64);
        else cy.zf[fn - 16] = lds_b128(nx_unit + lo.zf + (fn - 16) * 32);
        if (f >= 4 && f < 8) fetch_ci_part(cy.ci, nx_unit + C5::UNITB, lo, f - 4);
        if (f < 9) dma.piece(ld_slot, f);
        mfma_o(O[f & 7], P[f >> 3], tf[f % RING]);
        SPIN();
    }
}

template <int ROLE, bool EXACT, bool LOADX>
__device__ __forceinline__ void main_pass5(const StripP& p, const Geo5& g, char* smem, const bf16* X, v4i (&XF)[32], f32x16 (&O)[8],
                                           float& add, float& m2, float& lsum) {
    constexpr bool YS = ROLE == ROLE_YF;
    constexpr int SLOTB = C5::SLOTB, UNITB = C5::UNITB, NS = C5::NSLOT5;
    const LaneOff lo = g.lo;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct) {
#pragma unroll
        for (int r = 0; r < 16; ++r) O[ct][r] = 0.f;
        asm volatile("" : "+a"(O[ct]));
    }
    lsum = 0.f;
    if (g.nunit == 0) {
        if (LOADX) load_xfrags5(XF, X, g);
        return;
    }
    Dma5 dma;
    dma.init(p, g.Z, g.z_hi, g.z_first, g.wave, g.lane);
    const unsigned lds0 = lds_addr(smem);
#pragma unroll
    for (int v = 0; v < 3; ++v) dma.issue(lds0 + v * SLOTB, g.z_lo + v * ZU);       // units 0 .. 2 -> slots 0 .. 2
    if (LOADX) load_xfrags5(XF, X, g);
    VM_WAIT(0);
    lds_barrier();
    f32x16 Sa, Sb;
    Carry cy;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) fetch_ci_part(cy.ci, smem + UNITB, lo, gq);
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        const v4i zf = lds_b128(smem + lo.zf + ks * 32);
        if (ks == 0) mfma_s0(Sa, zf, XF[0], cy.ci);
        else mfma_s(Sa, zf, XF[ks]);
    }
    settle_s(Sa);
    asm volatile("" ::"v"(cy.ci));
    if (YS && !EXACT) {
        float t = Sa[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) t = fmaxf(t, Sa[r]);
        m2 = fmaxf(t, __shfl_xor(t, 32, 64)) * L2E;
        add = -m2;
    }
#pragma unroll
    for (int i = 0; i < PF; ++i) cy.zf[i] = lds_b128(smem + SLOTB + lo.zf + i * 32);
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) fetch_ci_part(cy.ci, smem + SLOTB + UNITB, lo, gq);
    const int nunit2 = (g.nunit + 1) & ~1;
    int sl = 0;                         // ring slot of unit u
#pragma clang loop unroll(disable)
    for (int u = 0; u < nunit2; u += 2) {
        const int q1 = (sl + 1) & (NS - 1), q2 = (sl + 2) & (NS - 1), q3 = (sl + 3) & (NS - 1);
        // iteration u: S(u+1) from q1, O(u) from sl, carry from q2 (unit u+2); loads unit u+3 into q3 (the slot of unit u-1)
        dma.begin(g.z_lo + (u + 3) * ZU);
        unit_iter5(O, XF, Sa, Sb, add, lsum, cy, smem + q1 * SLOTB, smem + sl * SLOTB, smem + q2 * SLOTB, lo, dma, lds0 + q3 * SLOTB);
        // iteration u+1: S(u+2) from q2, O(u+1) from q1, carry from q3 (unit u+3); loads unit u+4 into sl (the slot of unit u)
        dma.begin(g.z_lo + (u + 4) * ZU);
        unit_iter5(O, XF, Sb, Sa, add, lsum, cy, smem + q2 * SLOTB, smem + q1 * SLOTB, smem + q3 * SLOTB, lo, dma, lds0 + sl * SLOTB);
        sl = q2;
    }
    settle_o(O);
    VM_WAIT(0);
}

template <int ROLE>
__device__ __forceinline__ void epilogue5(const StripP& p, const Geo5& g, char* smem, const f32x16 (&O)[8], float m2, float lsum) {
    constexpr bool YS = ROLE == ROLE_YF;
    constexpr int OSTR = C5::OSTR;
    __syncthreads();
    float* stg_o = reinterpret_cast<float*>(smem) + g.wave * XW * OSTR;
#pragma unroll
    for (int ct = 0; ct < 8; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) stg_o[((r & 3) + 8 * (r >> 2) + 4 * g.hi) * OSTR + 32 * ct + g.l31] = O[ct][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    float* slab = p.slabs + (long)g.by * g.slab_stride + g.half * C5::CO;
#pragma unroll 4
    for (int xr = 0; xr < XW; ++xr) {
        const int gx = g.xbase + xr;
        const float4 v = *reinterpret_cast<const float4*>(stg_o + xr * OSTR + 4 * g.lane);
        if (gx < g.xend) *reinterpret_cast<float4*>(slab + (long)gx * C5::CW + 4 * g.lane) = v;
    }
    const float s = lsum + __shfl_xor(lsum, 32, 64);
    const int gx = g.xbase + g.l31;
    if (g.hi == 0 && gx < g.xend && g.half == 0) {      // (both halves of an x block form the same sums: the first one writes them)
        if (YS) {
            p.part[((long)gx * g.nchunk_dev + g.by) * 2] = m2 * (1.0f / L2E);
            p.part[((long)gx * g.nchunk_dev + g.by) * 2 + 1] = s;
        } else if (gx > 0) {
            p.bias_slabs[(long)g.by * (p.I - 1) + gx - 1] = s;
        }
    }
}

__device__ __forceinline__ void max_unit5(float& mx, const v4i (&XF)[32], const char* unit, const LaneOff& lo) {
    f32x16 ci, S;
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) fetch_ci_part(ci, unit + C5::UNITB, lo, gq);
#pragma unroll
    for (int ks = 0; ks < 32; ++ks) {
        const v4i zf = lds_b128(unit + lo.zf + ks * 32);
        if (ks == 0) mfma_s0(S, zf, XF[0], ci);
        else mfma_s(S, zf, XF[ks]);
    }
    settle_s(S);
    asm volatile("" ::"v"(ci));
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[r]);
}

__device__ __attribute__((noinline)) void fallback_exact5(const StripP* pp, const Geo5* gp, char* smem) {
    const StripP p = *pp;
    const Geo5 g = *gp;
    v4i XF[32];
    load_xfrags5(XF, p.rows, g);
    Dma5 dma;
    dma.init(p, g.Z, g.z_hi, g.z_first, g.wave, g.lane);
    const unsigned lds0 = lds_addr(smem);
    float mx = -INFINITY;
    for (int u = 0; u < g.nunit; ++u) {
        __syncthreads();
        dma.issue(lds0, g.z_lo + u * ZU);
        VM_WAIT(0);
        __syncthreads();
        max_unit5(mx, XF, smem, g.lo);
    }
    __syncthreads();
    float m2 = fmaxf(mx, __shfl_xor(mx, 32, 64)) * L2E;
    float add = -m2, lsum;
    f32x16 O[8];
    main_pass5<ROLE_YF, true, false>(p, g, smem, p.rows, XF, O, add, m2, lsum);
    epilogue5<ROLE_YF>(p, g, smem, O, m2, lsum);
}

// grid: ROLE_YF 2 G workgroups (id & 1 = channel half, id >> 1 = the (x block, item chunk) of dev_plan over G); ROLE_W (x blocks, row chunks, 2)
template <int ROLE>
__global__ __launch_bounds__(NTHR, 1) void stripw5_kernel(StripP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool YS = ROLE == ROLE_YF;
    Geo5 g;
    g.tid = threadIdx.x; g.lane = g.tid & 63; g.wave = __builtin_amdgcn_readfirstlane(g.tid >> 6); g.hi = g.lane >> 5; g.l31 = g.lane & 31;
    g.Reff = p.nvalid ? min(p.R, p.nvalid[0]) : p.R;
    int bx, zchunk;
    g.nchunk_dev = 1;
    if (YS) {
        const int G = (int)gridDim.x >> 1;
        const DevPlan dp = dev_plan(g.Reff, XB, G, p.i1 - p.i0, ZQ);
        int id2 = blockIdx.x;
        if ((gridDim.x & 7) == 0) id2 = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);      // XCD-aware order (stripw_kernel)
        g.half = id2 & 1;
        const int id = id2 >> 1;
        if (id >= dp.nx * dp.nchunk || g.Reff <= 0) return;
        bx = id % dp.nx; g.by = id / dp.nx; zchunk = dp.zchunk; g.nchunk_dev = dp.nchunk;
        g.slab_stride = (long)dp.nx * XB * C5::CW;
    } else {
        bx = blockIdx.x; g.by = blockIdx.y; g.half = blockIdx.z;
        const int nq = (g.Reff + ZQ - 1) / ZQ;
        zchunk = (nq + (int)gridDim.y - 1) / (int)gridDim.y * ZQ;
        g.slab_stride = (long)p.I * C5::CW;
    }
    g.xbase = (YS ? 0 : p.i0) + bx * XB + g.wave * XW;
    g.xend = YS ? g.Reff : p.i1;
    const bf16* X = YS ? p.rows : p.table;
    g.Z = YS ? p.table : p.rows;
    g.z_lo = (YS ? p.i0 : 0) + g.by * zchunk;
    g.z_hi = min(YS ? p.i1 : g.Reff, g.z_lo + zchunk);
    g.nunit = g.z_hi > g.z_lo ? (g.z_hi - g.z_lo + ZU - 1) / ZU : 0;
    g.z_first = YS ? p.i0 : 0;
    g.lo = lane_off5(g.lane, g.half);
    v4i XF[32];
    float add, lsum = 0.f, m2 = 0.f;
    {
        const int gx = g.xbase + g.l31;
        const float ob = YS ? 0.f : p.out_bias[min(max(gx, 1), p.I - 1) - 1];
        add = YS ? 0.f : (gx >= g.xend ? -INFINITY : (gx == 0 ? -1000.0f * L2E : ob * L2E));
    }
    {
        f32x16 O[8];
        main_pass5<ROLE, false, true>(p, g, smem, X, XF, O, add, m2, lsum);
        bool bad = false;
        if (YS) {
            const float s = lsum + __shfl_xor(lsum, 32, 64);
            bad = (g.xbase + g.l31 < g.xend) && !(s < LSUM_LIMIT);
        }
        if (!YS || !__syncthreads_or(bad ? 1 : 0)) {
            epilogue5<ROLE>(p, g, smem, O, m2, lsum);
            return;
        }
    }
    if (YS) {
        const StripP p2 = p;
        const Geo5 g2 = g;
        fallback_exact5(&p2, &g2, smem);
    }
}

// d_table[label[r]] -= coef[r] rows[r];  d_bias[label[r] - 1] -= coef[r]   over the weighted rows: the one-hot part of
// dl = coef (p - onehot) that the ROLE_W product pass leaves out (strip::label_scatter_kernel at any width: a block = RB rows x CW
// channels, equal labels summed in LDS first in row order by one thread per channel, the leaders' sums leave as f32 atomics).
template <int CW, int RB>
__global__ __launch_bounds__(CW) void label_scatter_kernel(const bf16* rows, const int64_t* labels, const float* coef,
                                                           const int32_t* nvalid, int R, int i0, int i1, const float* gscale,
                                                           float* d_table, float* d_bias) {
    __shared__ float acc[RB][CW];
    __shared__ float accb[RB];
    __shared__ int lab_s[RB], lead_s[RB];
    __shared__ float cf_s[RB];
    const int Reff = nvalid ? min(R, nvalid[0]) : R;
    const int r0 = blockIdx.x * RB, tid = threadIdx.x;
    if (r0 >= Reff) return;
    const float gs = gscale ? gscale[0] : 1.0f;
    float xv[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) xv[j] = (float)rows[(long)min(r0 + j, Reff - 1) * CW + tid];
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
        if (lab_s[j] >= 0 && lead_s[j] == j) atomicAdd(d_table + (long)lab_s[j] * CW + tid, -acc[j][tid]);
    if (tid < RB && lab_s[tid] >= 0 && lead_s[tid] == tid) atomicAdd(d_bias + lab_s[tid] - 1, -accb[tid]);
}

// C operand of every z row of a pass, entry k = row z_first + k, CPAD entries of -inf behind the range (units past a chunk's end):
//   ROLE_YF: z = i0 + k:  -1000 for the pad item (Base.py:110), out_bias[z - 1] otherwise;
//   ROLE_W:  row k: log coef - lse (the row's softmax scale times its loss coefficient, Appendix C); -inf for rows without weight.
__global__ __launch_bounds__(256) void info_items_kernel(const float* out_bias, int i0, int i1, float* cz) {
    const int k = blockIdx.x * 256 + threadIdx.x, n = i1 - i0;
    if (k >= n + CPAD) return;
    const int z = i0 + k;
    cz[k] = k >= n ? -INFINITY : (z == 0 ? -1000.0f : out_bias[z - 1]);
}
__global__ __launch_bounds__(256) void info_rows_kernel(const float* coef, const float* lse, const int32_t* nvalid, int R, float* rc) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= R + CPAD) return;
    const int Reff = nvalid ? min(R, nvalid[0]) : R;
    float c = -INFINITY;
    if (k < Reff) { const float cf = coef[k]; if (cf > 0.f) c = __logf(cf) - lse[k]; }
    rc[k] = c;
}

}  // namespace stripw

// ---- host side (called from k_score.hip) --------------------------------------------------------------------------------------
bool edgl_stripw_enabled() {
    static const int on = getenv("EDGL_SCORE_STRIPW") ? atoi(getenv("EDGL_SCORE_STRIPW")) : 1;
    return on != 0;
}
bool edgl_stripw_supports(int C) {      // 256: stripw_kernel; 512: stripw5_kernel (EDGL_SCORE_STRIPW512=0: the generic kernels, the A/B switch)
    static const int on512 = getenv("EDGL_SCORE_STRIPW512") ? atoi(getenv("EDGL_SCORE_STRIPW512")) : 1;
    return C == 256 || (C == 512 && on512 != 0);
}

// the dynamic-LDS attribute of a kernel is per device (see k_score_strip.hip)
static void stripw_set_smem_attr(const void* kern, int which, int bytes) {
    static std::atomic<uint64_t> done[4];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) {
        hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        return;
    }
    const uint64_t bit = 1ull << dev;
    if (done[which].load(std::memory_order_acquire) & bit) return;
    hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    done[which].fetch_or(bit, std::memory_order_release);
}

// floats of scratch a pass needs for its C operands (edgl_stripw_rows: n = i1 - i0; edgl_stripw_table: n = R)
long edgl_stripw_info_floats(long n) { return n + stripw::CPAD; }

int edgl_stripw_rows(const void* rows, const void* table, const float* out_bias, int R, int C, int I, int i0, int i1,
                     const int32_t* nvalid, float* slabs, float* part, int G, float* info_ws, hipStream_t st) {
    EDGL_REQUIRE(edgl_stripw_supports(C), EDGL_ERR_SHAPE, "edgl_stripw_rows: C=%d unsupported", C);
    EDGL_REQUIRE(info_ws, EDGL_ERR_NULL, "edgl_stripw_rows: no scratch for the C operands");
    const int n = i1 - i0 + stripw::CPAD;
    hipLaunchKernelGGL(stripw::info_items_kernel, dim3((n + 255) / 256), dim3(256), 0, st, out_bias, i0, i1, info_ws);
    stripw::StripP p{};
    p.rows = (const bf16*)rows; p.table = (const bf16*)table; p.out_bias = out_bias; p.R = R; p.I = I; p.i0 = i0; p.i1 = i1;
    p.nvalid = nvalid; p.slabs = slabs; p.part = part; p.cinfo = info_ws; p.ncinfo = n;
    if (C == 512) {
        auto k5 = stripw::stripw5_kernel<stripw::ROLE_YF>;
        stripw_set_smem_attr((const void*)k5, 2, stripw::C5::SMEM);
        hipLaunchKernelGGL(k5, dim3(2 * G), dim3(stripw::NTHR), stripw::C5::SMEM, st, p);
        EDGL_LAUNCH_CHECK();
        return EDGL_OK;
    }
    auto k = stripw::stripw_kernel<stripw::ROLE_YF, 256>;
    stripw_set_smem_attr((const void*)k, 0, stripw::W<256>::SMEM);
    hipLaunchKernelGGL(k, dim3(G), dim3(stripw::NTHR), stripw::W<256>::SMEM, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

int edgl_stripw_table(const void* rows, const void* table, const float* out_bias, const float* coef, const float* row_lse, int R,
                      int C, int I, int i0, int i1, const int32_t* nvalid, float* slabs, float* bias_slabs, int nchunk, float* info_ws,
                      hipStream_t st) {
    EDGL_REQUIRE(edgl_stripw_supports(C), EDGL_ERR_SHAPE, "edgl_stripw_table: C=%d unsupported", C);
    EDGL_REQUIRE(info_ws, EDGL_ERR_NULL, "edgl_stripw_table: no scratch for the C operands");
    const int n = R + stripw::CPAD;
    hipLaunchKernelGGL(stripw::info_rows_kernel, dim3((n + 255) / 256), dim3(256), 0, st, coef, row_lse, nvalid, R, info_ws);
    stripw::StripP p{};
    p.rows = (const bf16*)rows; p.table = (const bf16*)table; p.out_bias = out_bias; p.R = R; p.I = I; p.i0 = i0; p.i1 = i1;
    p.nvalid = nvalid; p.slabs = slabs; p.bias_slabs = bias_slabs; p.cinfo = info_ws; p.ncinfo = n;
    if (C == 512) {
        auto k5 = stripw::stripw5_kernel<stripw::ROLE_W>;
        stripw_set_smem_attr((const void*)k5, 3, stripw::C5::SMEM);
        hipLaunchKernelGGL(k5, dim3((i1 - i0 + stripw::XB - 1) / stripw::XB, nchunk, 2), dim3(stripw::NTHR), stripw::C5::SMEM, st, p);
        EDGL_LAUNCH_CHECK();
        return EDGL_OK;
    }
    auto k = stripw::stripw_kernel<stripw::ROLE_W, 256>;
    stripw_set_smem_attr((const void*)k, 1, stripw::W<256>::SMEM);
    hipLaunchKernelGGL(k, dim3((i1 - i0 + stripw::XB - 1) / stripw::XB, nchunk), dim3(stripw::NTHR), stripw::W<256>::SMEM, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

int edgl_stripw_label_scatter(const void* rows, const int64_t* labels, const float* coef, const int32_t* nvalid, int R, int C, int i0,
                              int i1, const float* gscale, float* d_table, float* d_bias, hipStream_t st) {
    EDGL_REQUIRE(edgl_stripw_supports(C), EDGL_ERR_SHAPE, "edgl_stripw_label_scatter: C=%d unsupported", C);
    if (C == 512)
        hipLaunchKernelGGL((stripw::label_scatter_kernel<512, 16>), dim3((R + 15) / 16), dim3(512), 0, st, (const bf16*)rows, labels, coef,
                           nvalid, R, i0, i1, gscale, d_table, d_bias);
    else
        hipLaunchKernelGGL((stripw::label_scatter_kernel<256, 32>), dim3((R + 31) / 32), dim3(256), 0, st, (const bf16*)rows, labels, coef,
                           nvalid, R, i0, i1, gscale, d_table, d_bias);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

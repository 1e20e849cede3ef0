// K3 backward sweep kernels (X and Z of k_bimau_bwd.hip's three passes), shared by k_bimau_bwd.hip (head dims 16 / 32) and
// k_bimau_big.hip (head dims 64 / 128, where the intensity MLP backward runs as two GEMM-shaped kernels in between).
#pragma once
#include <algorithm>

#include "bimau_common.h"

#ifndef EDGL_EXP_SKIP_TILES
#define EDGL_EXP_SKIP_TILES 0   // timing experiment: query tiles left out at the end (wrong results, bounds the cost of the remainder tile)
#endif

#ifdef EDGL_PHASE_TIMING
extern __device__ unsigned long long g_phase_cycles[16];
#endif

namespace bimau {


constexpr int KY_ECH = 8;        // marks per workgroup row of kernel Y (gridDim.y = 16 / KY_ECH dH partials)
constexpr int KY_NY = EP / KY_ECH;
constexpr int KY_BLOCKS = 384;   // workgroups of kernel Y per mark group (x KY_NY = one resident round at 3 WGs/CU)

// head dims >= 64 (k_bimau_big.hip): the weight-gradient kernel runs on a (row splits, channel groups) grid; a group is
// big_nj(dh) channel tiles of 16.  Enough row splits for ~512 workgroups, at least 16.  Head dim 64: four tiles — 64
// accumulator registers, 253 in all, two waves per SIMD; with eight (402 registers, one wave per SIMD, nothing to hide its
// LDS / MFMA / sigmoid chain behind) the kernel measured 167 us at the recipe shape against 122.
#ifndef EDGL_WG_NJ64
#define EDGL_WG_NJ64 4
#endif
constexpr int big_nj(int dh) { return dh == 64 ? EDGL_WG_NJ64 : 32 / (dh / 16); }   // channel tiles of a weight-gradient workgroup
inline int big_row_splits(int dh, int E) {
    const int nc = big_nj(dh) * 16;
    const int groups = std::max(1, (dh * E + nc - 1) / nc);
    return std::max(16, 512 / groups);
}

// workspace of edgl_bimau_bwd: dz [R,16] | dH partial slabs | row term [R] | dscaling partials | weight-gradient partials
struct WsLayout { size_t dz, dh, rowdot, dsc, wpart, total; };
inline WsLayout ws_layout(int B, int T_, int C, int H, int E) {
    const int dh = C / H;
    const bool big = dh >= 64;
    const size_t R = (size_t)B * H * T_;
    WsLayout w;
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    w.dz = take(R * EP * sizeof(float));
    w.dh = take((size_t)(big ? 1 : KY_NY) * R * dh * sizeof(float));
    w.rowdot = take(R * sizeof(float));
    w.dsc = take((size_t)B * H * EP * sizeof(float));
    w.wpart = take((size_t)(big ? big_row_splits(dh, E) : KY_BLOCKS) * ((size_t)(dh + 3) * dh * E + EP) * sizeof(float));
    w.total = o;
    return w;
}

struct BwdP {
    const void* qkvt; const int64_t* ids; const float* spans; const uint8_t* marks; const char* pack;
    const void* d_out; const float* d_lam_ext; const float* lam; const float* z; const void* hin;
    int B, T, C, H, E;
    float rate; const uint64_t* rng; uint32_t stream_id;
    void* d_qkvt;
    float* dz_ws; float* dh_ws; float* rowdot_ws; float* dsc_part; float* wpart;
    int waves;
    int flags;   // MAU_CAUSAL | MAU_NO_DIAG | MAU_DIAG_ZERO
    int noskip;  // EDGL_MAU_NO_SKIP (host side: the unskipped kernels)
    const uint32_t* dbits;   // optional: keep bits of the attention dropout (edgl_bimau_dropbits, bimau_common.h)
    float qk_scale;          // score scale (0: 1 / sqrt(dh)); a zero-padded head of true width d < dh passes 1 / sqrt(d)
    // optional (edgl_tpp_prep, bimau_common.h): sweep 1 recomputes the TPP regulariser's d lambda from the slot data instead of
    // reading d_lam_ext; tpp_sums[4] (int) = next-event mark count of the batch (NULL: the sum of edgl_tpp_prep's per-sample counts),
    // tpp_coef = ct_reg / H
    const void* tpp_desc; int tpp_M; const float* tpp_sums; float tpp_coef;
    const int32_t* order;   // optional [B]: the samples in launch order (edgl_bimau_job_order); NULL: 0 .. B-1
    float* tpp_part;   // [B*H + 1, 2]: the wave's share of the regulariser's two loss sums (sum log event intensity | sum non-event
                       // term); the last pair's first word = the normaliser's mark count (int)
};

template <typename T>
__device__ __forceinline__ void st_frag(T* dst, const f32x4& a) {
    Frag4<T> f = frag_from_acc<T>(a);
    if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(&f);
    else *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(&f);
}

// ------------------------------------------------------------------------------------------------------------------
// X) sweep 1
// ------------------------------------------------------------------------------------------------------------------
// PREF: fetch the next query tile's operands one iteration ahead (head dims <= 32); at head dims 64 / 128 the doubled
// operand set would not fit the register file, so the next tile is fetched at the end of the iteration instead.
// FL: -1 = MAU_CAUSAL / MAU_NO_DIAG read from p.flags at run time; 0 = the BiMAU configuration compiled in (bidirectional,
// diagonal set): no per-element causal compares / selects in the softmax, the diagonal as one select on a scalar-and-ed mask
// DB: stored keep bits of the attention dropout instead of the hash (same decisions: bimau_common.h)
// TP: d lambda of the TPP regulariser recomputed from p.tpp_desc (bimau_common.h) instead of loaded from p.d_lam_ext
// SK: the all-padding key tiles in front of the sequence's first real key are left out (FL == 0 only; KeyMask::kt0, bimau_common.h):
//     the query loop and the dV accumulators exist once per key-tile count NK, dV of the skipped key rows is written as zeros
template <typename T, int DT, int NT, int EC, bool PREF = true, int FL = -1, bool DB = false, bool TP = false, bool SK = false>
__global__ __launch_bounds__(256) void bimau_bwd_sweep1_kernel(BwdP p) {   // (forcing 3 waves / SIMD: 83 spilled registers, 76 -> 239 us)
    constexpr int dh = 16 * DT, Tp = 16 * NT, LDT = Tp + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int E = EC ? EC : p.E;
    // the wave index through the scalar unit: (b, head) and every base pointer derived from them are then wave-uniform VALUES for
    // the compiler too — global addresses become scalar base + 32-bit lane offset instead of 64-bit vector arithmetic per load
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // launch slot -> (b, head): p.order (edgl_bimau_job_order) lists the samples by falling key-tile count, so that the long jobs
    // start first (a launch is two rounds of workgroups: in launch order the slowest CU would get two long ones)
    const long slot = (long)blockIdx.x * p.waves + wave;
    if (slot >= (long)p.B * p.H) return;
    const int sb = (int)(slot / p.H), head = (int)(slot % p.H);
    const int b = p.order ? __builtin_amdgcn_readfirstlane(p.order[sb]) : sb;
    const long job = (long)b * p.H + head;   // index of the per-(b, head) partials: independent of the launch order
    const long bp = (long)head * p.B + b;
    const int g4 = (lane >> 4) * 4, l15 = lane & 15;
    constexpr bool SWZ = img_swz<T>();                // bank-swizzled row-major images (bimau_common.h: swz_col)
    const int g4s = SWZ ? swz_col<16>(l15, g4) : g4;  // column of this lane's row fragment in the mark image
    int cw[DT];                                       // ... in the K / T_ / V images (row 16 kt + l15), per 16-channel block
#pragma unroll
    for (int ub = 0; ub < DT; ++ub) cw[ub] = SWZ ? swz_col<16 * DT>(l15, ub * 16 + g4) : ub * 16 + g4;

    // wave-private LDS: K, V row-major [Tp][dh], marks [Tp][16] (+ transposed marks for f32), additive key mask
    constexpr bool TR = sizeof(T) == 2;
    constexpr size_t EXTRA = TR ? 0 : (size_t)EP * LDT;
    constexpr size_t MASK_ELEMS = (size_t)Tp * sizeof(float) / sizeof(T);
    constexpr size_t WAVE_ELEMS = 2 * (size_t)Tp * dh + (size_t)Tp * EP + EXTRA + MASK_ELEMS;
    T* Ks = reinterpret_cast<T*>(smem) + (size_t)wave * WAVE_ELEMS;
    T* Vs = Ks + Tp * dh;
    T* Ms = Vs + Tp * dh;
    T* MTs = Ms + Tp * EP;   // f32 only
    const T* qkvt = reinterpret_cast<const T*>(p.qkvt) + (long)b * p.T * 4 * p.C;
    const T* dout = reinterpret_cast<const T*>(p.d_out) + (long)b * p.T * p.C;
    T* dqkvt = reinterpret_cast<T*>(p.d_qkvt) + (long)b * p.T * 4 * p.C;
    const int ldq = 4 * p.C;
    const KeyMask<NT> km = stage_wave<T, DT, NT, EC>(
        qkvt + p.C + head * dh, Ks, nullptr, static_cast<const T*>(nullptr), static_cast<T*>(nullptr), static_cast<T*>(nullptr),
        qkvt + 2 * p.C + head * dh, Vs, nullptr, ldq, p.marks + (long)b * p.T * E, E, Ms, TR ? nullptr : MTs,
        p.ids + (long)b * p.T, reinterpret_cast<float*>(Ks + WAVE_ELEMS - MASK_ELEMS), p.T, LDT, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const PackDims pd = pack_dims<T>(dh, E);
    const float* iscs_g = reinterpret_cast<const float*>(p.pack + pd.off_f32) + 3 * pd.JE + EP;   // 1 / exp(scaling)
    float isc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) isc[i] = iscs_g[g4 + i];

    const float cscale = p.qk_scale > 0.f ? p.qk_scale : rsqrtf((float)dh);
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const Frag4<T> ident = identity_frag<T>(lane);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // Per-query-tile global operands are fetched one tile ahead.  The loads are unconditional (row / mark index clamped)
    // so that the prefetch is straight-line code: with per-lane branches around them the wait-count insertion falls
    // back to near-zero counts and every iteration would stall on the loads it has just issued.  Lanes past the end of
    // the sequence (or marks >= E) are zeroed when the values are consumed.
    struct QOps { Frag4<T> qf[DT], dof[DT]; float4 z; float lam[4], dlx[4]; uint32_t kb; uint32_t nmw; float spr; };
    const float* dlx_src = p.d_lam_ext ? p.d_lam_ext : p.lam;   // always a readable [rows, E] array
    const float dlx_on = p.d_lam_ext ? 1.0f : 0.0f;
    TppDesc td{};
    int tp_novf = 0;
    float tp_k = 0.f;
    if constexpr (TP) {
        td = tpp_desc(p.tpp_desc, p.B, p.T, p.tpp_M);
        tp_novf = __builtin_amdgcn_readfirstlane(td.novf[b]);
        int cnt;    // the regulariser's normaliser: given (edgl_tpp_norm / a data-parallel all-reduce), or the sum of the samples' counts
        if (p.tpp_sums) {
            cnt = reinterpret_cast<const int*>(p.tpp_sums)[4];
        } else {
            cnt = 0;
            for (int i = lane; i < p.B; i += 64) cnt += td.cntp[i];
            for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
        }
        tp_k = -p.tpp_coef / ((float)cnt * (float)p.H);   // temporal.py:331-332
        // ... and behind the partial sums for the launch that finishes the loss term (it then reads nothing of the batch)
        if (job == 0 && lane == 0) reinterpret_cast<int*>(p.tpp_part)[2 * p.B * p.H] = cnt;
    }
    auto load_q = [&](int qt) {
        QOps o;
        const int q = min(qt * 16 + l15, p.T - 1);
        const long row = bp * p.T + q;
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) {
            o.qf[ub] = frag_ld<T>(qkvt + (long)q * ldq + head * dh + ub * 16 + g4);
            o.dof[ub] = frag_ld<T>(dout + (long)q * p.C + head * dh + ub * 16 + g4);
        }
        o.z = *reinterpret_cast<const float4*>(p.z + row * EP + g4);
        if constexpr (DB) o.kb = p.dbits[(bp * NT + qt) * 64 + lane];
        if constexpr (TP) {   // slot data of the row's position instead of a d lambda row
            const float4 l4 = *reinterpret_cast<const float4*>(p.lam + row * EC + g4);
            o.lam[0] = l4.x; o.lam[1] = l4.y; o.lam[2] = l4.z; o.lam[3] = l4.w;
            o.dlx[0] = 0.f; o.dlx[1] = 0.f; o.dlx[2] = 0.f; o.dlx[3] = 0.f;
            o.nmw = td.nmw[((long)b * p.T + q) * 4 + (lane >> 4)];
            o.spr = td.spr[(long)b * p.T + q];
        } else if constexpr (EC == 16) {   // one 16-byte load each
            const float4 l4 = *reinterpret_cast<const float4*>(p.lam + row * EC + g4), d4 = *reinterpret_cast<const float4*>(dlx_src + row * EC + g4);
            o.lam[0] = l4.x; o.lam[1] = l4.y; o.lam[2] = l4.z; o.lam[3] = l4.w;
            o.dlx[0] = d4.x; o.dlx[1] = d4.y; o.dlx[2] = d4.z; o.dlx[3] = d4.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = min(g4 + i, E - 1);
                o.lam[i] = p.lam[row * E + e];
                o.dlx[i] = dlx_src[row * E + e];
            }
        }
        return o;
    };
    // zero what a lane past the sequence end (or a mark >= E) must not contribute
    auto mask_q = [&](QOps& o, bool ok) {
#pragma unroll
        for (int ub = 0; ub < DT; ++ub)
            if (!ok) o.dof[ub] = frag_zero<T>();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool oke = ok && (EC == 16 || (g4 + i) < E);
            o.lam[i] = oke ? o.lam[i] : 0.f;
            o.dlx[i] = oke ? o.dlx[i] * dlx_on : 0.f;
        }
    };
    PH_DECL
    auto run = [&](auto nk_c) {
    constexpr int NK = decltype(nk_c)::value, K0 = NT - NK;   // this copy of the loop walks the key tiles K0 .. NT-1
    const KeyMask<NK> kmk = keymask_tail<NK, NT>(km);
    float dsc_acc[4] = {0.f, 0.f, 0.f, 0.f}, tp_a = 0.f, tp_b = 0.f;   // (declared in here: state a lambda captures by reference may end up in scratch)
    f32x4 dVa[DT][NK];  // L(first=v, second=k)
#pragma unroll
    for (int u = 0; u < DT; ++u)
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) dVa[u][kt] = zero4;
    QOps qcur = load_q(0);
    touch_regs(qcur);   // complete before the loop (edgl_common.h)
    // Results of a query tile are stored at the TOP of the next iteration: the loop-carried prefetch makes the compiler
    // drain vmcnt to 0 on the back edge, and a store issued just before it would expose its full write latency there.
    float pend_dz[4] = {0.f, 0.f, 0.f, 0.f}, pend_rowdot = 0.f;
    int pend_q = p.T;   // >= T: nothing pending
    auto flush_pending = [&]() {
        if (pend_q < p.T) {
            *reinterpret_cast<float4*>(p.dz_ws + (bp * p.T + pend_q) * EP + g4) = make_float4(pend_dz[0], pend_dz[1], pend_dz[2], pend_dz[3]);
            if (lane < 16) p.rowdot_ws[bp * p.T + pend_q] = pend_rowdot;
        }
    };
    for (int qt = 0; qt < NT - EDGL_EXP_SKIP_TILES; ++qt) {
        asm volatile("" ::: "memory");   // keep loop-invariant LDS operands from being hoisted into registers
        const int q = qt * 16 + l15;
        const bool qok = q < p.T;
        QOps qnext;
        if constexpr (PREF) qnext = load_q(qt + 1 < NT ? qt + 1 : qt);
        asm volatile("" ::: "memory");   // the prefetch loads stay here (otherwise they are sunk to their use at the loop end)
        mask_q(qcur, qok);
        if constexpr (TP) {   // d lambda of the TPP regulariser at the masked positions among these rows (temporal.py:317-333)
            float gr[4] = {0.f, 0.f, 0.f, 0.f};
            tpp_slot(qcur.lam, qcur.nmw, qcur.spr, qok && qcur.spr >= 0.f, tp_k, gr, tp_a, tp_b);
            for (int j = 0; j < tp_novf; ++j) {   // further slots on an already taken position (normally none)
                const int tj = td.ovf_pos[(long)b * td.M + j];
                if ((tj >> 4) != qt) continue;
                tpp_slot(qcur.lam, td.ovf_nm[((long)b * td.M + j) * 4 + (lane >> 4)], qcur.spr, qok && q == tj, tp_k, gr, tp_a, tp_b);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) qcur.dlx[i] = gr[i];
        }
        const float zq4[4] = {qcur.z.x, qcur.z.y, qcur.z.z, qcur.z.w};
        // ---- recompute S, P --------------------------------------------------------------------------------------
        f32x4 s[NK];
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            f32x4 a = zero4;
#pragma unroll
            for (int ub = 0; ub < DT; ++ub)
                a = mma16(frag_ld<T>(Ks + ((K0 + kt) * 16 + l15) * dh + cw[ub]), qcur.qf[ub], a);
            s[kt] = a;
        }
        // s = P^T * (dropout scale), L(first=k, second=q): every use of P in this sweep carries the scale, so it rides on the softmax
        // normalisation (no multiply of its own)
        if constexpr (FL == 0) masked_softmax_impl<NK, false, false>(s, kmk, cscale, lane, q, dk.scale);
        else {
            masked_softmax<NK, 0>(s, kmk, cscale, lane, q, (p.flags & MAU_CAUSAL) != 0);
#pragma unroll
            for (int kt = 0; kt < NK; ++kt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[kt][r] *= dk.scale;
        }
        // The previous tile's results leave HERE, behind the first use of this tile's operands: stores count in vmcnt like loads, and
        // stores issued at the top of the iteration (in front of the prefetch) sat between the loads of the previous iteration and
        // the wait for them — that wait then also waited for the store to be acknowledged (~2 us per query tile in the stamps).
        asm volatile("" ::: "memory");
        flush_pending();
        asm volatile("" ::: "memory");
        PH_MARK(0);
        Frag4<T> lf;
#pragma unroll
        for (int i = 0; i < 4; ++i) lf.v[i] = from_f32<T>(qcur.lam[i]);
        Frag4<T> dOT[DT];  // L(first=q, second=v): A operand contracting over q
#pragma unroll
        for (int vt = 0; vt < DT; ++vt) dOT[vt] = frag_from_acc<T>(mma16(qcur.dof[vt], ident, zero4));
        // ---- G', dA, dG -> dlambda, row term, dV ------------------------------------------------------------------
        float rowdot = 0.f;   // this lane's part of  sum_k dP1[q][k] P[q][k]
        f32x4 dlamT = zero4;  // L(first=e, second=q)
        const uint32_t dbase = (uint32_t)((bp * p.T + q) * p.T);   // dropout element index of (b', q, k=0)
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            const f32x4 gacc = mma16(frag_ld<T>(Ms + ((K0 + kt) * 16 + l15) * EP + g4s), lf, zero4);
            f32x4 da = zero4;
#pragma unroll
            for (int vb = 0; vb < DT; ++vb)
                da = mma16(frag_ld<T>(Vs + ((K0 + kt) * 16 + l15) * dh + cw[vb]), qcur.dof[vb], da);
            f32x4 ap, dg, gv = gacc;
            const bool dtile = K0 + kt == qt && (FL == 0 || !(p.flags & MAU_NO_DIAG));   // only this key tile can hold k == q
            if constexpr (FL == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gv[r] = (dtile && g4 + r == l15) ? 1.0f : gacc[r];   // G' diag := 1 (temporal.py:438-439)
            } else if (dtile) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gv[r] = (g4 + r == l15) ? ((p.flags & MAU_DIAG_ZERO) ? 0.0f : 1.0f) : gacc[r];
            }
            // dropout: one hash per four neighbouring elements (drop_hash_quad; rate 0: threshold 0, everything kept, scale 1)
            uint64_t hw = 0ull;
            if constexpr (!DB) hw = drop_hash_quad(dk, dbase + (K0 + kt) * 16 + g4);
            const bool keep[4] = {drop_quad_keep<0>(dk, hw), drop_quad_keep<1>(dk, hw), drop_quad_keep<2>(dk, hw), drop_quad_keep<3>(dk, hw)};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float fp;                                   // D*P (with the scale)
                if constexpr (DB) fp = keep_bit(qcur.kb, (K0 + kt) * 4 + r, s[kt][r]);
                else fp = keep[r] ? s[kt][r] : 0.f;
                ap[r] = gv[r] * fp;                         // A' = D*G'*P
                dg[r] = da[r] * fp;                         // dG' = dA' * D * P
                rowdot = fmaf(dg[r], gv[r], rowdot);        // dP1 * P, dP1 = dA' * D * G'  (before the diagonal of dG' is blocked)
            }
            if constexpr (FL == 0) {   // set_diag blocks the gradient into lambda
#pragma unroll
                for (int r = 0; r < 4; ++r) dg[r] = (dtile && g4 + r == l15) ? 0.f : dg[r];
            } else if (dtile) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dg[r] = (g4 + r == l15) ? 0.f : dg[r];
            }
            dlamT = mma16(kfrag<T, SWZ ? 16 : 0>(Ms, EP, MTs, LDT, (K0 + kt) * 16, 0, lane), frag_from_acc<T>(dg), dlamT);
            const Frag4<T> apT = frag_from_acc<T>(transpose_tile<T>(ap, ident));  // L(first=q, second=k)
#pragma unroll
            for (int vt = 0; vt < DT; ++vt) dVa[vt][kt] = mma16(dOT[vt], apT, dVa[vt][kt]);
            if (kt & 1) __builtin_amdgcn_sched_barrier(0);   // at most two key tiles' temporaries interleaved
        }
        PH_MARK(1);
        // ---- dlambda -> dz, dscaling; row term --------------------------------------------------------------------
        float dz4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float dl = dlamT[i] + qcur.dlx[i];
            const float sg = sigmoid_f(zq4[i] * isc[i]);   // softplus'
            dz4[i] = dl * sg;
            if (qok && (EC == 16 || (g4 + i) < E)) dsc_acc[i] += dl * (qcur.lam[i] - zq4[i] * sg);
        }
        pend_rowdot = group_sum4(rowdot);
#pragma unroll
        for (int i = 0; i < 4; ++i) pend_dz[i] = dz4[i];
        pend_q = q;
        if constexpr (PREF) qcur = qnext;
        else if (qt + 1 < NT) qcur = load_q(qt + 1);
        PH_MARK(2);
    }
    // ---- dscaling partial: sum over the 16 query lanes (before the stores: nothing behind them waits on vmcnt) ------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = dsc_acc[i];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        dsc_acc[i] = v;
    }
    if constexpr (TP) {   // the wave's share of the two loss sums (every lane of a row's four holds the row's terms: one lane writes)
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { tp_a += __shfl_xor(tp_a, o, 64); tp_b += __shfl_xor(tp_b, o, 64); }
    }
    flush_pending();
    if constexpr (TP) {
        if (lane == 0) *reinterpret_cast<float2*>(p.tpp_part + job * 2) = make_float2(tp_a, tp_b);
    }
    if (l15 == 0) *reinterpret_cast<float4*>(p.dsc_part + job * EP + g4) = make_float4(dsc_acc[0], dsc_acc[1], dsc_acc[2], dsc_acc[3]);
    // ---- write dV (L(first=v, second=k): 4 consecutive channels of key row k) -----------------------------------------
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const int k = kt * 16 + l15;
        if (k < p.T) {
#pragma unroll
            for (int ut = 0; ut < DT; ++ut) {
                T* dst = dqkvt + (long)k * ldq + 2 * p.C + head * dh + ut * 16 + g4;
                if constexpr (K0 > 0) {   // (kt < K0: a key tile of padding only — its dV rows are exactly 0)
                    if (kt < K0) st_frag<T>(dst, zero4);
                    else st_frag<T>(dst, dVa[ut][kt < K0 ? 0 : kt - K0]);
                } else {
                    st_frag<T>(dst, dVa[ut][kt]);
                }
            }
        }
    }
    };   // run
    if constexpr (SK) {
        static_assert(FL == 0 && sizeof(T) == 2, "the key-tile skip exists for the bidirectional bf16 form");
        dispatch_nk<NT>(NT - km.kt0, run);
    } else {
        run(std::integral_constant<int, NT>{});
    }
    PH_MARK(3);
    PH_FLUSH(0);
}

// ------------------------------------------------------------------------------------------------------------------
// Z) sweep 2
// ------------------------------------------------------------------------------------------------------------------
// NYP: number of dH partial slabs the intensity backward left in dh_ws (KY_NY mark groups at head dims <= 32, 1 above)
// US / SL: channel slices.  The per-(b, head) accumulators dK, dT_ are DT x NT register tiles each — at head dim 128 and more than
// four key tiles (T > 64) they alone exceed the register file.  A launch with US = 2 keeps the tiles of channel slice SL only
// (dQ, dK, dT_ of those dh / US channels); everything that contracts over the whole head (S, P, dP, the row term) is computed by
// every slice.  One launch per slice (the slice index selects REGISTERS: it has to be a compile-time constant).
// SK: as in sweep 1 — the all-padding key tiles in front of the first real key are left out, dK / dT_ of their rows are written as zeros
template <typename T, int DT, int NT, int EC, int NYP = KY_NY, bool PREF = true, int FL = -1, bool DB = false, int US = 1, int SL = 0, bool SK = false>
__global__ __launch_bounds__(256) void bimau_bwd_sweep2_kernel(BwdP p) {
    constexpr int dh = 16 * DT, Tp = 16 * NT, LDT = Tp + 4;
    static_assert(DT % US == 0 && SL < US, "channel slices");
    constexpr int DTS = DT / US, U0 = SL * DTS;   // this launch's channel tiles: U0 .. U0 + DTS - 1
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int E = EC ? EC : p.E;
    // the wave index through the scalar unit: (b, head) and every base pointer derived from them are then wave-uniform VALUES for
    // the compiler too — global addresses become scalar base + 32-bit lane offset instead of 64-bit vector arithmetic per load
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // launch slot -> (b, head): p.order (edgl_bimau_job_order) lists the samples by falling key-tile count, so that the long jobs
    // start first (a launch is two rounds of workgroups: in launch order the slowest CU would get two long ones)
    const long slot = (long)blockIdx.x * p.waves + wave;
    if (slot >= (long)p.B * p.H) return;
    const int sb = (int)(slot / p.H), head = (int)(slot % p.H);
    const int b = p.order ? __builtin_amdgcn_readfirstlane(p.order[sb]) : sb;
    const long job = (long)b * p.H + head;   // index of the per-(b, head) partials: independent of the launch order
    const long bp = (long)head * p.B + b;
    const int g4 = (lane >> 4) * 4, l15 = lane & 15;
    constexpr bool SWZ = img_swz<T>();                // bank-swizzled row-major images (bimau_common.h: swz_col)
    const int g4s = SWZ ? swz_col<16>(l15, g4) : g4;  // column of this lane's row fragment in the mark image
    int cw[DT];                                       // ... in the K / T_ / V images (row 16 kt + l15), per 16-channel block
#pragma unroll
    for (int ub = 0; ub < DT; ++ub) cw[ub] = SWZ ? swz_col<16 * DT>(l15, ub * 16 + g4) : ub * 16 + g4;

    // wave-private LDS: K, T_, V row-major [Tp][dh], marks [Tp][16]; f32 additionally K^T (no 32-bit transpose read)
    constexpr bool TR = sizeof(T) == 2;
    constexpr size_t EXTRA = TR ? 0 : (size_t)dh * LDT;
    constexpr size_t MASK_ELEMS = (size_t)Tp * sizeof(float) / sizeof(T);
    constexpr size_t WAVE_ELEMS = 3 * (size_t)Tp * dh + (size_t)Tp * EP + EXTRA + MASK_ELEMS;
    T* Ks = reinterpret_cast<T*>(smem) + (size_t)wave * WAVE_ELEMS;
    T* Ts = Ks + Tp * dh;
    T* Vs = Ts + Tp * dh;
    T* Ms = Vs + Tp * dh;
    T* KTs = Ms + Tp * EP;   // f32 only
    const T* qkvt = reinterpret_cast<const T*>(p.qkvt) + (long)b * p.T * 4 * p.C;
    const T* dout = reinterpret_cast<const T*>(p.d_out) + (long)b * p.T * p.C;
    const T* hin = reinterpret_cast<const T*>(p.hin);
    T* dqkvt = reinterpret_cast<T*>(p.d_qkvt) + (long)b * p.T * 4 * p.C;
    const int ldq = 4 * p.C;
    const KeyMask<NT> km = stage_wave<T, DT, NT, EC>(
        qkvt + p.C + head * dh, Ks, TR ? nullptr : KTs, qkvt + 3 * p.C + head * dh, Ts, static_cast<T*>(nullptr),
        qkvt + 2 * p.C + head * dh, Vs, static_cast<T*>(nullptr), ldq, p.marks + (long)b * p.T * E, E, Ms, static_cast<T*>(nullptr),
        p.ids + (long)b * p.T, reinterpret_cast<float*>(Ks + WAVE_ELEMS - MASK_ELEMS), p.T, LDT, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const float cscale = p.qk_scale > 0.f ? p.qk_scale : rsqrtf((float)dh);
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    // 1/sqrt(dh) of dS, or 0 when every key of the sequence is padded (uniform softmax: no score takes a gradient, temporal.py:425-426)
    float cz = cscale;
    if constexpr (FL == 0) {
        bool real = false;
        for (int k = lane; k < Tp; k += 64) real |= km.madd[k] == 0.f;
        cz = __any(real) ? cscale : 0.f;
    }
    (void)cz;
    const Frag4<T> ident = identity_frag<T>(lane);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const long R = (long)p.B * p.H * p.T;

    // branch-free one-tile-ahead prefetch (see kernel X); kernel Y's dH partials are summed when consumed
    struct QOps { Frag4<T> qf[DT], dof[DT], hf[DT]; float4 dHp[DT][NYP]; float lam[4], rowdot; uint32_t kb; };
    auto load_q = [&](int qt) {
        QOps o;
        const int q = min(qt * 16 + l15, p.T - 1);
        const long row = bp * p.T + q;
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) {
            o.qf[ub] = frag_ld<T>(qkvt + (long)q * ldq + head * dh + ub * 16 + g4);
            o.dof[ub] = frag_ld<T>(dout + (long)q * p.C + head * dh + ub * 16 + g4);
            o.hf[ub] = frag_ld<T>(hin + row * dh + ub * 16 + g4);
#pragma unroll
            for (int y = 0; y < NYP; ++y)
                o.dHp[ub][y] = *reinterpret_cast<const float4*>(p.dh_ws + ((long)y * R + row) * dh + ub * 16 + g4);
        }
        if constexpr (EC == 16) {
            const float4 l4 = *reinterpret_cast<const float4*>(p.lam + row * EC + g4);
            o.lam[0] = l4.x; o.lam[1] = l4.y; o.lam[2] = l4.z; o.lam[3] = l4.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) o.lam[i] = p.lam[row * E + min(g4 + i, E - 1)];
        }
        o.rowdot = p.rowdot_ws[row];
        if constexpr (DB) o.kb = p.dbits[(bp * NT + qt) * 64 + lane];
        return o;
    };
    PH_DECL
#ifdef EDGL_PHASE_TIMING
    const unsigned long long clk_m0 = __builtin_readcyclecounter(), clk_r0 = wall_clock64();
#endif
    auto run = [&](auto nk_c) {
    constexpr int NK = decltype(nk_c)::value, K0 = NT - NK;   // this copy of the loop walks the key tiles K0 .. NT-1
    const KeyMask<NK> kmk = keymask_tail<NK, NT>(km);
    f32x4 dKa[DTS][NK], dTa[DTS][NK];  // L(first=u, second=k), channel tiles U0 ..
#pragma unroll
    for (int u = 0; u < DTS; ++u)
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) { dKa[u][kt] = zero4; dTa[u][kt] = zero4; }
    QOps qcur = load_q(0);
    touch_regs(qcur);   // complete before the loop (edgl_common.h)
    // dQ of a query tile is stored at the top of the next iteration (see kernel X: the back edge drains vmcnt)
    Frag4<T> pend_dq[DTS];
    int pend_q = p.T;
    auto flush_pending = [&]() {
        if (pend_q < p.T) {
#pragma unroll
            for (int ut = 0; ut < DTS; ++ut) {
                T* dst = dqkvt + (long)pend_q * ldq + head * dh + (U0 + ut) * 16 + g4;
                if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(&pend_dq[ut]);
                else *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(&pend_dq[ut]);
            }
        }
    };
#pragma unroll
    for (int ut = 0; ut < DTS; ++ut) pend_dq[ut] = frag_zero<T>();
    for (int qt = 0; qt < NT - EDGL_EXP_SKIP_TILES; ++qt) {
        asm volatile("" ::: "memory");
        const int q = qt * 16 + l15;
        const bool qok = q < p.T;
        (void)qok;
        QOps qnext;
        if constexpr (PREF) qnext = load_q(qt + 1 < NT ? qt + 1 : qt);
        asm volatile("" ::: "memory");   // the prefetch loads stay here (otherwise they are sunk to their use at the loop end)
        PH_MARK(3);
        // consume the tile fetched one iteration ago; rows past the sequence end contribute nothing to dK / dT_
        f32x4 dHq[DT];   // dH^T[u][q], L(first=u, second=q): sum of the mark-group partials in a fixed order
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) {
            f32x4 a = zero4;
#pragma unroll
            for (int y = 0; y < NYP; ++y) {
                a[0] += qcur.dHp[ub][y].x; a[1] += qcur.dHp[ub][y].y; a[2] += qcur.dHp[ub][y].z; a[3] += qcur.dHp[ub][y].w;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) dHq[ub][r] = qok ? a[r] : 0.f;
            if (!qok) qcur.dof[ub] = frag_zero<T>();
        }
        const float rowdot1 = qok ? qcur.rowdot : 0.f;
        PH_MARK(4);
        // ---- recompute S, P --------------------------------------------------------------------------------------
        f32x4 s[NK];
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            f32x4 a = zero4;
#pragma unroll
            for (int ub = 0; ub < DT; ++ub)
                a = mma16(frag_ld<T>(Ks + ((K0 + kt) * 16 + l15) * dh + cw[ub]), qcur.qf[ub], a);
            s[kt] = a;
        }
        PH_MARK(5);
        if constexpr (FL == 0) masked_softmax_impl<NK, false, false>(s, kmk, cscale, lane, q);
        else masked_softmax<NK, 0>(s, kmk, cscale, lane, q, (p.flags & MAU_CAUSAL) != 0);  // s = P^T, L(first=k, second=q)
        asm volatile("" ::: "memory");
        flush_pending();   // behind the first use of this tile's operands (see kernel X)
        asm volatile("" ::: "memory");
        PH_MARK(0);
        // G' carries the dropout scale here (lambda * scale, diagonal := scale): dP1 = D * dA' * G' is then one multiply and one select
        Frag4<T> lf;
#pragma unroll
        for (int i = 0; i < 4; ++i) lf.v[i] = from_f32<T>((EC == 16 || (g4 + i) < E) ? qcur.lam[i] * dk.scale : 0.f);
        // rowsum(dP*P) = sum_k dP1*P (kernel X) + sum_k P[q][k] (dH[q].T_[k]) = ... + dH[q].H[q]   (H = P.T_, saved)
        float rowdot = 0.f;
        Frag4<T> dhf[DT], QT[DTS], dHT[DTS];
#pragma unroll
        for (int ut = 0; ut < DT; ++ut) {
#pragma unroll
            for (int r = 0; r < 4; ++r) rowdot = fmaf(dHq[ut][r], to_f32(qcur.hf[ut].v[r]), rowdot);
            dhf[ut] = frag_from_acc<T>(dHq[ut]);
        }
#pragma unroll
        for (int ut = 0; ut < DTS; ++ut) {
            QT[ut] = frag_from_acc<T>(mma16(qcur.qf[U0 + ut], ident, zero4));      // L(first=q, second=u)
            dHT[ut] = frag_from_acc<T>(transpose_tile<T>(dHq[U0 + ut], ident));
        }
        rowdot = group_sum4(rowdot) + rowdot1;
        // ---- dP = dP1 + dH.T_^T ; dS = P*(dP - rowsum(dP*P)) * c ; dQ, dK, dT_ -------------------------------------
        f32x4 dQ[DTS];
#pragma unroll
        for (int ut = 0; ut < DTS; ++ut) dQ[ut] = zero4;
        const uint32_t dbase = (uint32_t)((bp * p.T + q) * p.T);
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            const f32x4 gacc = mma16(frag_ld<T>(Ms + ((K0 + kt) * 16 + l15) * EP + g4s), lf, zero4);
            f32x4 da = zero4;
#pragma unroll
            for (int vb = 0; vb < DT; ++vb)
                da = mma16(frag_ld<T>(Vs + ((K0 + kt) * 16 + l15) * dh + cw[vb]), qcur.dof[vb], da);
            f32x4 gv = gacc;
            if constexpr (FL == 0) {
                const bool dtile = K0 + kt == qt;
#pragma unroll
                for (int r = 0; r < 4; ++r) gv[r] = (dtile && g4 + r == l15) ? dk.scale : gacc[r];
            } else if (kt == qt && !(p.flags & MAU_NO_DIAG)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gv[r] = (g4 + r == l15) ? ((p.flags & MAU_DIAG_ZERO) ? 0.0f : dk.scale) : gacc[r];
            }
            uint64_t hw = 0ull;
            if constexpr (!DB) hw = drop_hash_quad(dk, dbase + (K0 + kt) * 16 + g4);
            const bool keep[4] = {drop_quad_keep<0>(dk, hw), drop_quad_keep<1>(dk, hw), drop_quad_keep<2>(dk, hw), drop_quad_keep<3>(dk, hw)};
            f32x4 a;
#pragma unroll
            for (int r = 0; r < 4; ++r) {                                       // dP1 = D * dA' * G'  (kernel X's dP1)
                if constexpr (DB) a[r] = keep_bit(qcur.kb, (K0 + kt) * 4 + r, da[r] * gv[r]);
                else a[r] = keep[r] ? da[r] * gv[r] : 0.f;
            }
#pragma unroll
            for (int ub = 0; ub < DT; ++ub)
                a = mma16(frag_ld<T>(Ts + ((K0 + kt) * 16 + l15) * dh + cw[ub]), dhf[ub], a);
            f32x4 ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // tf.where(mask==0, paddings, S) (temporal.py:425-426) passes no gradient to a padded key's
                // score; P is non-zero there only for fully padded rows (uniform softmax)
                // BiMAU (FL == 0): the masked keys are the same for every query row, and P is EXACTLY 0 on them unless all keys of the
                // sequence are padded — the select and the 1/sqrt(dh) factor are one per-wave factor `cz` on dQ and dK instead.
                if constexpr (FL == 0) {
                    ds[r] = s[kt][r] * (a[r] - rowdot);
                } else {
                    const bool padded = ((km.pad >> (kt * 4 + r)) & 1ull) || ((p.flags & MAU_CAUSAL) && kt * 16 + g4 + r > q);
                    ds[r] = padded ? 0.f : s[kt][r] * (a[r] - rowdot) * cscale;
                }
            }
            const Frag4<T> dsf = frag_from_acc<T>(ds);
#pragma unroll
            for (int ut = 0; ut < DTS; ++ut)
                dQ[ut] = mma16(kfrag<T, SWZ ? 16 * DT : 0>(Ks, dh, KTs, LDT, (K0 + kt) * 16, (U0 + ut) * 16, lane), dsf, dQ[ut]);
            const Frag4<T> dsT = frag_from_acc<T>(transpose_tile<T>(ds, ident));
            const Frag4<T> pT = frag_from_acc<T>(transpose_tile<T>(s[kt], ident));
#pragma unroll
            for (int ut = 0; ut < DTS; ++ut) {
                dKa[ut][kt] = mma16(QT[ut], dsT, dKa[ut][kt]);
                dTa[ut][kt] = mma16(dHT[ut], pT, dTa[ut][kt]);
            }
        }
#pragma unroll
        for (int ut = 0; ut < DTS; ++ut) {
            if constexpr (FL == 0) {
#pragma unroll
                for (int r = 0; r < 4; ++r) dQ[ut][r] *= cz;
            }
            pend_dq[ut] = frag_from_acc<T>(dQ[ut]);
        }
        pend_q = q;
        if constexpr (PREF) qcur = qnext;
        else if (qt + 1 < NT) qcur = load_q(qt + 1);
        PH_MARK(1);
    }
    flush_pending();
    // ---- write dK / dT_ (L(first=u, second=k): 4 consecutive channels of key row k) ----------------------------------
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const int k = kt * 16 + l15;
        if (k < p.T) {
#pragma unroll
            for (int ut = 0; ut < DTS; ++ut) {
                T* row = dqkvt + (long)k * ldq + head * dh + (U0 + ut) * 16 + g4;
                if (kt < K0) {   // a key tile of padding only: dK and dT_ of its rows are exactly 0
                    st_frag<T>(row + p.C, zero4);
                    st_frag<T>(row + 3 * p.C, zero4);
                } else {
                    const int kk = kt < K0 ? 0 : kt - K0;
                    if constexpr (FL == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) dKa[ut][kk][r] *= cz;
                    }
                    st_frag<T>(row + p.C, dKa[ut][kk]);
                    st_frag<T>(row + 3 * p.C, dTa[ut][kk]);
                }
            }
        }
    }
    };   // run
    if constexpr (SK) {
        static_assert(FL == 0 && sizeof(T) == 2, "the key-tile skip exists for the bidirectional bf16 form");
        dispatch_nk<NT>(NT - km.kt0, run);
    } else {
        run(std::integral_constant<int, NT>{});
    }
    PH_MARK(2);
    PH_FLUSH(8);
#ifdef EDGL_PHASE_TIMING   // shader cycles and 100 MHz ticks of this wave's life: their ratio is the clock the kernel ran at
    if (lane == 0) { atomicAdd(&g_phase_cycles[14], __builtin_readcyclecounter() - clk_m0); atomicAdd(&g_phase_cycles[15], wall_clock64() - clk_r0); }
#endif
}


// head dims 64 / 128 (k_bimau_big.hip)
int big_bwd(const BwdP& p, char* ws, float* dW1, float* db1, float* dw, float* dsc, int dtype, hipStream_t st);

}  // namespace bimau

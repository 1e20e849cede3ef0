// K3 forward kernel template, shared by k_bimau_fwd.hip (head dims 16 / 32: one fused launch) and k_bimau_big.hip (head dims
// 64 / 128: the intensity MLP runs as its own row-tile kernel between two phases of this one).
//   PHASE 0  fused:   S, softmax, H = P.T_, intensity MLP (weights in LDS), G = lambda.marks^T, diag := 1, dropout, O = A.V
//   PHASE 1  scores:  S, softmax, H = P.T_  -> H rows (saved->hin); no weights, no marks
//   PHASE 2  values:  S, softmax recomputed, lambda rows read back, G, diag := 1, dropout, O = A.V + residual
// BiMAU.__call__ (temporal.py:404-452) with MAU.intensity (temporal.py:281-315); see bimau_common.h for the layout scheme.
#pragma once
#include <type_traits>

#include "bimau_common.h"

#ifndef EDGL_EXP_SKIP_TILES
#define EDGL_EXP_SKIP_TILES 0   // timing experiment: query tiles left out at the end (wrong results, bounds the cost of the remainder tile)
#endif

#ifndef EDGL_BIMAU_FWD_WAVES
#define EDGL_BIMAU_FWD_WAVES 2   // waves per SIMD the headline instance (bf16, head dim 16, E = 16) is compiled for (3: 168 registers, 4 spilled — 79 -> 84 us)
#endif

namespace bimau {


struct FwdP {
    const void* qkvt; const void* resid; int ld_res;
    const int64_t* ids; const float* spans; const uint8_t* marks;
    const char* pack;
    int B, T, C, H, E;
    float rate; const uint64_t* rng; uint32_t stream_id;
    void* out; float* lam;
    void* hin_out; float* z_out;   // saved for the backward (NULL: inference)
    float* zero_rows;              // optional [H*B, T, E] f32 array that this launch fills with zeros (the engine's d lambda buffer:
                                   //  edgl_tpp_fwd_bwd_rows then writes the masked positions only); head dims 16 / 32 only
    int waves;
    int flags;   // MAU_CAUSAL | MAU_NO_DIAG | MAU_DIAG_ZERO
    const uint32_t* dbits;   // optional: keep bits of the attention dropout (edgl_bimau_dropbits, bimau_common.h)
    float qk_scale;          // score scale (0: 1 / sqrt(dh), temporal.py:422); a zero-padded head of true width d < dh passes 1 / sqrt(d)
    const int32_t* order;    // optional [B]: the samples in launch order (edgl_bimau_job_order); NULL: 0 .. B-1
    int noskip;              // EDGL_MAU_NO_SKIP: launch the kernels that walk every key tile (the caller knows there is nothing to skip)
};

// wave-private LDS bytes of a phase (K always; T_ unless values phase; V and marks unless scores phase; the f32 key mask)
template <typename T, int DT, int NT, int PHASE>
__host__ __device__ constexpr size_t fwd_wave_bytes() {
    constexpr size_t dh = 16 * DT, Tp = 16 * NT, LDT = Tp + 4;
    constexpr size_t kv = sizeof(T) == 2 ? Tp * dh : dh * LDT;
    return (Tp * dh + (PHASE != 2 ? kv : 0) + (PHASE != 1 ? kv : 0) + (PHASE != 1 ? Tp * EP : 0)) * sizeof(T) + Tp * sizeof(float);
}

// DT = dh/16, NT = ceil(T/16); EC = compile-time mark count (16: all LDS offsets are immediates and the mark loop is
// one straight-line block) or 0 (runtime p.E)
// DB: the attention dropout reads stored keep bits (p.dbits) instead of hashing — same decisions (bimau_common.h)
// SK: the key tiles in front of the first real key (left padding: KeyMask::kt0) are left out — exact, see bimau_common.h.  The
//     query loop exists once per key-tile count NK = 1 .. NT (straight-line code with NK-entry register arrays each); the wave
//     picks its copy by a scalar branch.  Bidirectional flags only (a causal row's score replacement depends on q).
template <typename T, int DT, int NT, int EC, int PHASE = 0, bool DB = false, bool SK = false>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && DT == 1 && EC == 16 && PHASE == 0) ? EDGL_BIMAU_FWD_WAVES : 1) void bimau_fwd_kernel(FwdP p) {
    constexpr int dh = 16 * DT, Tp = 16 * NT, LDT = Tp + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int E = EC ? EC : p.E;
    constexpr bool FUSED = PHASE == 0, HAS_T = PHASE != 2, HAS_V = PHASE != 1, HAS_M = PHASE != 1;
    const PackDims pd = pack_dims<T>(dh, E);
    const size_t pack_bytes = FUSED ? pd.fwd_bytes : 0;   // the backward-only W1R image stays in HBM
    // ---- workgroup-shared intensity weights (fused form only) ---------------------------------
    if constexpr (FUSED) copy_pack_to_lds(smem, p.pack, pd.fwd_bytes);
    const T* W1T = reinterpret_cast<const T*>(smem);
    const T* W1X = reinterpret_cast<const T*>(smem + pd.off_w1x);
    const float* fW = reinterpret_cast<const float*>(smem + pd.off_f32);
    const float* w1s = fW; const float* b1s = fW + pd.JE; const float* wvs = fW + 2 * pd.JE;
    const float* scs = fW + 3 * pd.JE; const float* iscs = scs + EP;
    (void)W1T; (void)W1X; (void)w1s; (void)b1s; (void)wvs; (void)scs; (void)iscs;
    if constexpr (FUSED) __syncthreads();

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
    const long bp = (long)head * p.B + b;  // head-major index b' (temporal.py:413-416)

    // ---- wave-private LDS.  bf16: K, T_, V row-major [Tp][dh] + marks [Tp][16]; products that contract over the
    //      KEY index fetch their operand with transpose reads (kfrag).  f32: T_ and V are staged transposed instead.
    constexpr bool TR = sizeof(T) == 2;
    constexpr size_t KV_ELEMS = TR ? (size_t)Tp * dh : (size_t)dh * LDT;
    constexpr size_t MASK_ELEMS = (size_t)Tp * sizeof(float) / sizeof(T);   // additive key mask, f32 [Tp]
    constexpr size_t WAVE_ELEMS = (size_t)Tp * dh + (HAS_T ? KV_ELEMS : 0) + (HAS_V ? KV_ELEMS : 0) + (HAS_M ? (size_t)Tp * EP : 0) + MASK_ELEMS;
    static_assert(WAVE_ELEMS * sizeof(T) == fwd_wave_bytes<T, DT, NT, PHASE>(), "host / device LDS layouts differ");
    T* Ks = reinterpret_cast<T*>(smem + pack_bytes) + (size_t)wave * WAVE_ELEMS;
    T* Ts = Ks + Tp * dh;                       // T_ : row-major (bf16) or transposed [dh][LDT] (f32)
    T* Vs = Ts + (HAS_T ? KV_ELEMS : 0);        // V  : same
    T* Ms = Vs + (HAS_V ? KV_ELEMS : 0);
    const T* qkvt = reinterpret_cast<const T*>(p.qkvt) + (long)b * p.T * 4 * p.C;
    const int ldq = 4 * p.C;
    // K, T_, V (split order Q,K,V,T: temporal.py:410), marks, key mask — one round trip at the small shapes (stage_wave)
    const KeyMask<NT> km = stage_wave<T, DT, NT, EC>(
        qkvt + p.C + head * dh, Ks, nullptr,
        HAS_T ? qkvt + 3 * p.C + head * dh : nullptr, TR ? Ts : nullptr, TR ? nullptr : Ts,
        HAS_V ? qkvt + 2 * p.C + head * dh : nullptr, TR ? Vs : nullptr, TR ? nullptr : Vs, ldq,
        HAS_M ? p.marks + (long)b * p.T * E : nullptr, E, Ms, nullptr, p.ids + (long)b * p.T,
        reinterpret_cast<float*>(Ms + (HAS_M ? Tp * EP : 0)), p.T, LDT, lane);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#ifdef EDGL_PROLOGUE_ONLY   // diagnostic build (not the product): time of the LDS staging alone — rule 9 of DESIGN.md §4.2
    if (p.B > 0) { if (km.pad == 0x1234567ull) reinterpret_cast<T*>(p.out)[0] = Ks[lane] + Ms[lane]; return; }
#endif

    const float cscale = p.qk_scale > 0.f ? p.qk_scale : rsqrtf((float)dh);
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const int g4 = (lane >> 4) * 4, l15 = lane & 15;
    constexpr bool SWZ = img_swz<T>();                // bank-swizzled row-major images (bimau_common.h: swz_col)
    const int g4s = SWZ ? swz_col<16>(l15, g4) : g4;  // column of this lane's row fragment in the mark image
    int cw[DT];                                       // ... in the K / T_ / V images (row 16 kt + l15), per 16-channel block
#pragma unroll
    for (int ub = 0; ub < DT; ++ub) cw[ub] = SWZ ? swz_col<16 * DT>(l15, ub * 16 + g4) : ub * 16 + g4;

    // per-query-tile global operands (Q rows, interval, residual rows) are fetched one tile ahead: their HBM/L2 latency
    // overlaps the previous tile's compute instead of opening every iteration with a stall
    struct QOps { Frag4<T> qf[DT], rf[DT]; float span; float lam[4]; uint32_t kb; };
    auto load_q = [&](int qt) {   // unconditional (row clamped): branch-free, so the wait counts around it stay exact
        QOps o;
        const int q = min(qt * 16 + l15, p.T - 1);
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) {
            o.qf[ub] = frag_ld<T>(qkvt + (long)q * ldq + head * dh + ub * 16 + g4);
            o.rf[ub] = frag_ld<T>(reinterpret_cast<const T*>(p.resid) + ((long)b * p.T + q) * p.ld_res + head * dh + ub * 16 + g4);
        }
        o.span = p.spans[(long)b * p.T + q];
        if constexpr (DB) o.kb = p.dbits[(bp * NT + qt) * 64 + lane];
        if constexpr (PHASE == 2) {   // lambda rows written by the intensity kernel between the two phases
#pragma unroll
            for (int i = 0; i < 4; ++i) o.lam[i] = p.lam[(bp * p.T + q) * E + min(g4 + i, E - 1)];
        }
        return o;
    };
    auto run = [&](auto nk_c) {
    constexpr int NK = decltype(nk_c)::value, K0 = NT - NK;   // this copy of the loop walks the key tiles K0 .. NT-1
    const KeyMask<NK> kmk = keymask_tail<NK, NT>(km);
    QOps qcur = load_q(0);
    touch_regs(qcur);   // complete before the loop (edgl_common.h)
    // The output rows of a query tile (computed last) are stored at the TOP of the next iteration: the loop-carried
    // prefetch makes the compiler drain vmcnt to 0 on the back edge, and a store issued just before it would expose its
    // full write latency there.  (H rows, z and lambda are stored mid-iteration and have landed by then.)
    Frag4<T> pend_o[DT];
#pragma unroll
    for (int ut = 0; ut < DT; ++ut) pend_o[ut] = frag_zero<T>();
    int pend_q = p.T;   // >= T: nothing pending
    auto flush_pending = [&]() {
        if (pend_q < p.T) {
#pragma unroll
            for (int ut = 0; ut < DT; ++ut) {
                T* dst = reinterpret_cast<T*>(p.out) + ((long)b * p.T + pend_q) * p.C + head * dh + ut * 16 + g4;
                if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(&pend_o[ut]);
                else *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(&pend_o[ut]);
            }
        }
    };
    for (int qt = 0; qt < NT - EDGL_EXP_SKIP_TILES; ++qt) {
        const int q = qt * 16 + l15;
        const bool qok = q < p.T;
        const QOps qnext = load_q(qt + 1 < NT ? qt + 1 : qt);
        asm volatile("" ::: "memory");   // the prefetch loads stay here (otherwise they are sunk to their use at the loop end)
        // ---- S^T[k][q] = sum_u K[k][u] Q[q][u] ------------------------------------------------
        Frag4<T> qf[DT];
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) qf[ub] = qcur.qf[ub];
        f32x4 s[NK];
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ub = 0; ub < DT; ++ub)
                a = mma16(frag_ld<T>(Ks + ((K0 + kt) * 16 + l15) * dh + cw[ub]), qf[ub], a);
            s[kt] = a;
        }
        // bf16: s := exp(v - max), UNNORMALISED; 1 / sum (`pinv`) rides on the H rows (4 values) and, together with the dropout scale,
        // on lambda (G = lambda . marks^T is linear in it) — 2 x 28 multiplies per query tile that the tile itself never sees.
        // f32: s := P^T (pinv = 1).
        float pinv = 1.0f;
        if constexpr (TR) {
            if constexpr (K0 == 0) {
                if (p.flags & MAU_CAUSAL) pinv = masked_softmax_impl<NT, true, false, false>(s, kmk, cscale, lane, q);
                else pinv = masked_softmax_impl<NT, false, false, false>(s, kmk, cscale, lane, q);
            } else {
                pinv = masked_softmax_impl<NK, false, false, false>(s, kmk, cscale, lane, q);
            }
        } else {
            masked_softmax<NT, 0>(s, kmk, cscale, lane, q, (p.flags & MAU_CAUSAL) != 0);
        }
        const float gfac = pinv * dk.scale;   // factor of G (and of its diagonal)
        // the previous tile's output rows leave here, behind the first use of this tile's operands: a store in front of the prefetch
        // sits between the previous iteration's loads and the wait for them, and that wait then waits for the store's acknowledge
        asm volatile("" ::: "memory");
        flush_pending();
        asm volatile("" ::: "memory");
        // ---- H^T[u][q] = sum_k T_[k][u] P[q][k] -----------------------------------------------
        Frag4<T> pf[NK];
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) pf[kt] = frag_from_acc<T>(s[kt]);
        Frag4<T> hf[DT];
        if constexpr (HAS_T) {
#pragma unroll
            for (int ut = 0; ut < DT; ++ut) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kt = 0; kt < NK; ++kt)
                    a = mma16(kfrag<T, SWZ ? 16 * DT : 0>(Ts, dh, Ts, LDT, (K0 + kt) * 16, ut * 16, lane), pf[kt], a);
                if constexpr (TR) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) a[r] *= pinv;
                }
                hf[ut] = frag_from_acc<T>(a);
                if (p.hin_out && qok) {   // H rows in the activation dtype: exactly what the intensity MLP consumed
                    T* dst = reinterpret_cast<T*>(p.hin_out) + (bp * p.T + q) * dh + ut * 16 + g4;
                    if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(&hf[ut]);
                    else *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(&hf[ut]);
                }
            }
        }
        if constexpr (PHASE == 1) {   // scores phase: the intensity kernel takes over from the H rows
            qcur = qnext;
            continue;
        }
        Frag4<T> lf;
        if constexpr (FUSED) {
        // ---- intensity MLP (temporal.py:287-306): Zpre^T[j][q], channel j = e*dh + u' ------------
        const float span = qcur.span;
        float zp[16];
        // operands of channel tile jt: W1T rows (A operand), output weight, and the interval weight / bias — as the A rows of
        // a second MFMA (bf16: span * w1s + b1 comes out of the matrix pipe, see span_frag) or as f32 vectors (f32).  The tile
        // after the one being computed is always in flight (explicit double buffer: the LDS latency hides behind one tile
        // of MFMA + sigmoid work instead of stalling every tile).
        struct MlpOps { Frag4<T> w[DT]; Frag4<T> wx; f32x4 ws, bs, wv; };
        Frag4<T> sf;
        if constexpr (TR) sf = span_frag(span, lane);
        auto load_ops = [&](int jt) {
            MlpOps o;
#pragma unroll
            for (int ub = 0; ub < DT; ++ub) o.w[ub] = frag_ld<T>(W1T + (jt * 16 + l15) * pd.LDW + ub * 16 + g4);
            if constexpr (TR) {
                o.wx = frag_ld<T>(W1X + (jt * 16 + l15) * XW + (g4 & 4));
            } else {
                o.ws = *reinterpret_cast<const f32x4*>(w1s + jt * 16 + g4);
                o.bs = *reinterpret_cast<const f32x4*>(b1s + jt * 16 + g4);
            }
            o.wv = *reinterpret_cast<const f32x4*>(wvs + jt * 16 + g4);
            return o;
        };
        // pre-activation of one channel tile (matrix pipe) and sigmoid . output weight (VALU); `acc` is the running pair sum
        auto tile_pre = [&](const MlpOps& o) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            if constexpr (TR) a = mma16(o.wx, sf, a);
#pragma unroll
            for (int ub = 0; ub < DT; ++ub) a = mma16(o.w[ub], hf[ub], a);
            // f32: (span*ws + a) + bs, chained on the MFMA result so that nothing of this tile can be hoisted into the
            // prefetch slot above (which would wait on the loads just issued)
            if constexpr (!TR) {
#pragma unroll
                for (int r = 0; r < 4; ++r) a[r] = fmaf(span, o.ws[r], a[r]) + o.bs[r];
            }
            return a;
        };
        auto tile_post = [&](const f32x4& a, const MlpOps& o, float& acc, bool first) {
            float sg[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) sg[r] = fast_rcp(1.0f + __builtin_amdgcn_exp2f(a[r]));
            acc = first ? sg[0] * o.wv[0] : fmaf(sg[0], o.wv[0], acc);
#pragma unroll
            for (int r = 1; r < 4; ++r) acc = fmaf(sg[r], o.wv[r], acc);
        };
        if constexpr (EC == 16) {
            // two channel tiles per step: the exp / rcp chains of one tile fill the MFMA and transcendental latencies of the
            // other (a wave has only one partner on its SIMD); the pair after the one being computed is in flight
            // ... and software-pipelined: the MFMAs of the next pair are issued before the sigmoids of the current one, the
            // operands of the pair after that are in flight
            MlpOps c0 = load_ops(0), c1 = load_ops(1);
            MlpOps n0 = load_ops(2), n1 = load_ops(3);
            f32x4 a0 = tile_pre(c0), a1 = tile_pre(c1);
            float zacc;
#pragma unroll
            for (int jt = 0; jt < 16 * DT; jt += 2) {
                const int jn = jt + 4 < 16 * DT ? jt + 4 : jt;
                const MlpOps m0 = load_ops(jn), m1 = load_ops(jn + 1);
                EDGL_PIN();   // keep the prefetch at the top of this pair's work
                f32x4 b0 = a0, b1 = a1;
                if (jt + 2 < 16 * DT) { b0 = tile_pre(n0); b1 = tile_pre(n1); }
                const int e0 = jt / DT, e1 = (jt + 1) / DT;
                tile_post(a0, c0, zacc, jt % DT == 0);
                if (e1 != e0) zp[e0] = zacc;
                tile_post(a1, c1, zacc, (jt + 1) % DT == 0);
                if ((jt + 2) % DT == 0) zp[e1] = zacc;
                c0 = n0; c1 = n1; n0 = m0; n1 = m1; a0 = b0; a1 = b1;
                EDGL_PIN();
            }
        } else {
            MlpOps cur = load_ops(0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                zp[e] = 0.f;
                if (e < E) {
                    float zacc;
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        const int jt = e * DT + d;
                        const MlpOps nxt = load_ops(jt + 1 < E * DT ? jt + 1 : jt);
                        EDGL_PIN();   // keep the prefetch at the top of this tile's work
                        const f32x4 a = tile_pre(cur);
                        tile_post(a, cur, zacc, d == 0);
                        cur = nxt;
                        EDGL_PIN();
                    }
                    zp[e] = zacc;
                }
            }
        }
        float z4[4];
        reduce_scatter16(zp, z4, lane);  // lane group g now owns e = 4g + i
        if (p.z_out && qok) *reinterpret_cast<float4*>(p.z_out + (bp * p.T + q) * EP + g4) = make_float4(z4[0], z4[1], z4[2], z4[3]);
        float lam4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sc = scs[g4 + i], isc = iscs[g4 + i];
            lam4[i] = sc * __logf(1.0f + __expf(z4[i] * isc));  // temporal.py:305-306
            lf.v[i] = from_f32<T>(lam4[i] * gfac);
        }
        if (qok) {
            float* dst = p.lam + (bp * p.T + q) * E + g4;
            if constexpr (EC == 16) {
                *reinterpret_cast<float4*>(dst) = make_float4(lam4[0], lam4[1], lam4[2], lam4[3]);
                if (p.zero_rows) *reinterpret_cast<float4*>(p.zero_rows + (bp * p.T + q) * E + g4) = make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (g4 + i < E) { dst[i] = lam4[i]; if (p.zero_rows) p.zero_rows[(bp * p.T + q) * E + g4 + i] = 0.f; }
            }
        }
        } else {   // values phase: lambda of this query tile was prefetched with the Q rows
#pragma unroll
            for (int i = 0; i < 4; ++i) lf.v[i] = from_f32<T>(qcur.lam[i] * gfac);
        }
        // ---- G^T[k][q] = sum_e marks[k][e] lam[q][e]; diag := 1; A' = dropout(G * P) ------------
        const uint32_t dbase = (uint32_t)((bp * p.T + q) * p.T);   // element index of (b', q, k=0); < 2^32 (host-checked)
        // One straight-line block for all key tiles (the dropout decision is taken once, outside; the diagonal is a select
        // on a scalar-and-ed lane mask): the scheduler can issue the seven MFMAs ahead and run the hashes in their shadow.
        const bool set_diag = !(p.flags & MAU_NO_DIAG);
        const float dval = (p.flags & MAU_DIAG_ZERO) ? 0.0f : gfac;   // later mark groups of a split call (bimau_common.h): 0
        auto modulate = [&](auto drop_on) {
#pragma unroll
            for (int kt = 0; kt < NK; ++kt) {
                f32x4 gacc = mma16(frag_ld<T>(Ms + ((K0 + kt) * 16 + l15) * EP + g4s), lf, f32x4{0.f, 0.f, 0.f, 0.f});
                const bool dtile = set_diag && K0 + kt == qt;   // only this key tile can contain k == q (temporal.py:438-439)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    gacc[r] = (dtile && g4 + r == l15) ? dval : gacc[r];
                    s[kt][r] = gacc[r] * s[kt][r];     // temporal.py:441
                }
                if constexpr (decltype(drop_on)::value) {                       // temporal.py:442 (the scale is in G already)
                    if constexpr (DB) {   // stored decisions: bit kt*4 + r of this lane's word
#pragma unroll
                        for (int r = 0; r < 4; ++r) s[kt][r] = keep_bit(qcur.kb, (K0 + kt) * 4 + r, s[kt][r]);
                    } else {
                        const uint64_t hw = drop_hash_quad(dk, dbase + (K0 + kt) * 16 + g4);
                        s[kt][0] = drop_quad_keep<0>(dk, hw) ? s[kt][0] : 0.f;
                        s[kt][1] = drop_quad_keep<1>(dk, hw) ? s[kt][1] : 0.f;
                        s[kt][2] = drop_quad_keep<2>(dk, hw) ? s[kt][2] : 0.f;
                        s[kt][3] = drop_quad_keep<3>(dk, hw) ? s[kt][3] : 0.f;
                    }
                }
                pf[kt] = frag_from_acc<T>(s[kt]);
            }
        };
        if (dk.thresh != 0u) modulate(std::true_type{});
        else modulate(std::false_type{});
        // ---- O^T[v][q] = sum_k V[k][v] A'[q][k]; + residual (temporal.py:443-447) ----------------
#pragma unroll
        for (int vt = 0; vt < DT; ++vt) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kt = 0; kt < NK; ++kt)
                a = mma16(kfrag<T, SWZ ? 16 * DT : 0>(Vs, dh, Vs, LDT, (K0 + kt) * 16, vt * 16, lane), pf[kt], a);
            const Frag4<T> rf = qcur.rf[vt];
            f32x4 o4;
#pragma unroll
            for (int r = 0; r < 4; ++r) o4[r] = a[r] + to_f32(rf.v[r]);
            pend_o[vt] = frag_from_acc<T>(o4);
        }
        pend_q = q;
        qcur = qnext;
    }
    flush_pending();
    };   // run
    if constexpr (SK) {
        static_assert(TR && PHASE == 0, "the key-tile skip exists for the fused bf16 form");
        dispatch_nk<NT>((p.flags & MAU_CAUSAL) ? NT : NT - km.kt0, run);
    } else {
        run(std::integral_constant<int, NT>{});
    }
}

template <typename T, int DT, int NT, int EC, int PHASE = 0>
int launch_fwd_e(FwdP p, hipStream_t st) {
    constexpr int dh = 16 * DT;
    const size_t pack_bytes = PHASE == 0 ? pack_dims<T>(dh, p.E).fwd_bytes : 0;
    constexpr size_t wave_bytes = fwd_wave_bytes<T, DT, NT, PHASE>();
    int waves = 4;
    while (waves > 1 && pack_bytes + waves * wave_bytes > 80 * 1024) waves >>= 1;
    // Long sequences at head dim 32 (config 3: T = 201 -> 47 KB per wave + 39 KB of intensity weights): not even ONE wave stays under
    // the two-workgroups-per-CU line, so the CU holds one workgroup whatever its size — then as many waves as the LDS takes beside the
    // shared weights (one wave per CU measured 1.11 ms at B = 512; two: see DESIGN rule 70)
    if (waves == 1 && pack_bytes + wave_bytes > 80 * 1024)
        while (waves < 4 && pack_bytes + (waves + 1) * wave_bytes <= 160 * 1024) ++waves;
    const size_t smem = pack_bytes + waves * wave_bytes;
    EDGL_REQUIRE(smem <= 160 * 1024, EDGL_ERR_SHAPE, "edgl_bimau_fwd: needs %zu B of LDS (dh=%d E=%d T=%d)", smem, dh,
                 p.E, p.T);
    p.waves = waves;
    auto kern = bimau_fwd_kernel<T, DT, NT, EC, PHASE>;
    // stored keep bits: the headline family (bf16, head dim 16, 16 marks, <= 8 key tiles, fused form); elsewhere the hash.
    // The same family leaves out the all-padding key tiles in front of a sequence's first real key (SK; EDGL_BIMAU_SKIP=0: the
    // unskipped kernels — the A/B switch and the reference of the bit-equality test)
    if constexpr (sizeof(T) == 2 && DT == 1 && EC == 16 && PHASE == 0 && NT <= 8) {
        const bool sk = NT >= 2 && bimau_skip_enabled() && !p.noskip;
        if (p.dbits && p.rate > 0.f) kern = sk ? bimau_fwd_kernel<T, DT, NT, EC, PHASE, true, (NT >= 2)> : bimau_fwd_kernel<T, DT, NT, EC, PHASE, true>;
        else if (sk) kern = bimau_fwd_kernel<T, DT, NT, EC, PHASE, false, (NT >= 2)>;
    }
    if (smem > 48 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const long jobs = (long)p.B * p.H;
    hipLaunchKernelGGL(kern, dim3((unsigned)((jobs + waves - 1) / waves)), dim3(64 * waves), smem, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

template <typename T, int DT, int NT>
int launch_fwd(FwdP p, hipStream_t st) {
    return p.E == 16 ? launch_fwd_e<T, DT, NT, 16>(p, st) : launch_fwd_e<T, DT, NT, 0>(p, st);
}

template <typename T, int DT>
int dispatch_nt(FwdP p, hipStream_t st) {
    const int nt = (p.T + 15) / 16;
    switch (nt) {
        case 1: return launch_fwd<T, DT, 1>(p, st);
        case 2: return launch_fwd<T, DT, 2>(p, st);
        case 3: return launch_fwd<T, DT, 3>(p, st);
        case 4: return launch_fwd<T, DT, 4>(p, st);
        case 5: return launch_fwd<T, DT, 5>(p, st);
        case 6: return launch_fwd<T, DT, 6>(p, st);
        case 7: return launch_fwd<T, DT, 7>(p, st);
        case 8: return launch_fwd<T, DT, 8>(p, st);
        case 9: return launch_fwd<T, DT, 9>(p, st);
        case 10: return launch_fwd<T, DT, 10>(p, st);
        case 11: return launch_fwd<T, DT, 11>(p, st);
        case 12: return launch_fwd<T, DT, 12>(p, st);
        case 13: return launch_fwd<T, DT, 13>(p, st);
    }
    edgl_set_error("edgl_bimau_fwd: T=%d not supported (T <= 208)", p.T);
    return EDGL_ERR_SHAPE;
}


// head dims 64 / 128 (k_bimau_big.hip): scores phase -> intensity kernel -> values phase
int big_fwd(const FwdP& p, int dtype, hipStream_t st);

}  // namespace bimau

// K3 backward.  Two kernels:
//   A) attention part, one wave per (b, head): recomputes S, P, H, Z, lambda, G in registers, then
//      dA -> dG/dP1 -> dlambda -> dz -> du -> dH -> dP2 -> dS -> dQ, and accumulates dK/dV/dT_ over
//      its query tiles in MFMA accumulators (written once, no atomics).  Products that contract
//      over the QUERY index use operands transposed by one MFMA against the identity.
//   B) intensity weight gradients: row tiles of (Hin, dz) written by A are re-expanded
//      (Zpre = Hin.W1, sigmoid) in the orientation whose MFMA contraction runs over rows, giving
//      dW1 / db1 / dw partials per workgroup, reduced deterministically by a third tiny kernel.
#include "bimau_common.h"

#ifndef EDGL_BWD_OCC
#define EDGL_BWD_OCC
#endif
// -DEDGL_PHASE_TIMING builds a diagnostic variant: every wave accumulates s_memtime deltas per phase of the query loop
// and lane 0 adds them to g_phase_cycles (read back with edgl_debug_phase_cycles).  Not part of the product build.
#ifdef EDGL_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[16];
#define PH_DECL unsigned long long ph_t0 = __builtin_readcyclecounter(), ph_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define PH_MARK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t0; ph_t0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PH_FLUSH() do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 10; ++i_) atomicAdd(&g_phase_cycles[i_], ph_acc[i_]); } while (0)
#else
#define PH_DECL
#define PH_MARK(i)
#define PH_FLUSH()
#endif

namespace {
using namespace bimau;

constexpr int KB_BLOCKS = 192;  // workgroups of kernel B per mark group (x4 groups = 768 = 3 resident per CU: one full wave of WGs)

struct BwdP {
    const void* qkvt; const int64_t* ids; const float* spans; const uint8_t* marks; const char* pack;
    const void* d_out; const float* d_lam_ext;
    int B, T, C, H, E;
    float rate; const uint64_t* rng; uint32_t stream_id;
    void* d_qkvt;
    void* hin_ws; float* dz_ws; float* dsc_part; float* wpart;
    int waves;
};

template <typename T>
__device__ __forceinline__ void st_frag(T* dst, const f32x4& a) {
    Frag4<T> f = frag_from_acc<T>(a);
    if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(&f);
    else *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(&f);
}

template <typename T, int DT, int NT, int EC>
__global__ __launch_bounds__(256) EDGL_BWD_OCC void bimau_bwd_kernel(BwdP p) {
    constexpr int dh = 16 * DT, Tp = 16 * NT, LDT = Tp + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int E = EC ? EC : p.E;   // EC = 16: LDS offsets are immediates, the mark loops are straight-line code
    const PackDims pd = pack_dims<T>(dh, E);
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.pack);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = threadIdx.x; i < (int)(pd.bytes / 16); i += blockDim.x) dst[i] = src[i];
    }
    const T* W1T = reinterpret_cast<const T*>(smem);
    const T* W1R = reinterpret_cast<const T*>(smem + pd.off_w1r);
    const float* fW = reinterpret_cast<const float*>(smem + pd.off_f32);
    const float* w1s = fW; const float* b1s = fW + pd.JE; const float* wvs = fW + 2 * pd.JE;
    const float* scs = fW + 3 * pd.JE; const float* iscs = scs + EP;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long job = (long)blockIdx.x * p.waves + wave;
    if (job >= (long)p.B * p.H) return;
    const int b = (int)(job / p.H), head = (int)(job % p.H);
    const long bp = (long)head * p.B + b;

    // bf16: K, T_, V row-major [Tp][dh] + marks [Tp][16]; every product that contracts over the KEY index reads its
    // operand with transpose reads (kfrag).  f32: additionally K^T, T_^T, marks^T images (no 32-bit transpose read).
    constexpr bool TR = sizeof(T) == 2;
    constexpr size_t EXTRA = TR ? 0 : 2 * (size_t)dh * LDT + (size_t)EP * LDT;
    constexpr size_t MASK_ELEMS = (size_t)Tp * sizeof(float) / sizeof(T);   // additive key mask, f32 [Tp]
    constexpr size_t WAVE_ELEMS = 3 * (size_t)Tp * dh + (size_t)Tp * EP + EXTRA + MASK_ELEMS;
    T* Ks = reinterpret_cast<T*>(smem + pd.bytes) + (size_t)wave * WAVE_ELEMS;  // K  [Tp][dh]
    T* Ts = Ks + Tp * dh;                                                       // T_ [Tp][dh]
    T* Vs = Ts + Tp * dh;                                                       // V  [Tp][dh]
    T* Ms = Vs + Tp * dh;                                                       // marks   [Tp][16]
    T* KTs = Ms + Tp * EP;                                                      // f32 only: K^T  [dh][LDT]
    T* TTs = KTs + dh * LDT;                                                    // f32 only: T_^T [dh][LDT]
    T* MTs = TTs + dh * LDT;                                                    // f32 only: marks^T [16][LDT]
    const T* qkvt = reinterpret_cast<const T*>(p.qkvt) + (long)b * p.T * 4 * p.C;
    const T* dout = reinterpret_cast<const T*>(p.d_out) + (long)b * p.T * p.C;
    T* dqkvt = reinterpret_cast<T*>(p.d_qkvt) + (long)b * p.T * 4 * p.C;
    const int ldq = 4 * p.C;
    stage_rows<T>(qkvt + p.C + head * dh, ldq, p.T, Tp, dh, Ks, TR ? nullptr : KTs, LDT, lane);
    stage_rows<T>(qkvt + 3 * p.C + head * dh, ldq, p.T, Tp, dh, Ts, TR ? nullptr : TTs, LDT, lane);
    stage_rows<T>(qkvt + 2 * p.C + head * dh, ldq, p.T, Tp, dh, Vs, nullptr, LDT, lane);
    stage_marks<T>(p.marks + (long)b * p.T * E, E, p.T, Tp, Ms, TR ? nullptr : MTs, LDT, lane);
    const KeyMask<NT> km = load_keymask<NT>(p.ids + (long)b * p.T, p.T, lane, reinterpret_cast<float*>(Ks + WAVE_ELEMS - MASK_ELEMS));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    const float cscale = rsqrtf((float)dh);
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const int g4 = (lane >> 4) * 4, l15 = lane & 15;
    const Frag4<T> ident = identity_frag<T>(lane);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    f32x4 dKa[DT][NT], dVa[DT][NT], dTa[DT][NT];  // L(first=u, second=k)
#pragma unroll
    for (int u = 0; u < DT; ++u)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) { dKa[u][kt] = zero4; dVa[u][kt] = zero4; dTa[u][kt] = zero4; }
    float dsc_acc[4] = {0.f, 0.f, 0.f, 0.f};
    PH_DECL
    PH_MARK(0);   // staging

    // per-query-tile global operands (Q rows, dO rows, interval, upstream d lambda) are fetched one tile ahead so that
    // their HBM/L2 latency overlaps the previous tile's work
    struct QOps { Frag4<T> qf[DT], dof[DT]; float span, dlx[4]; };
    auto load_q = [&](int qt) {
        QOps o;
        const int q = qt * 16 + l15;
        const bool ok = q < p.T;
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) {
            o.qf[ub] = ok ? frag_ld<T>(qkvt + (long)q * ldq + head * dh + ub * 16 + g4) : frag_zero<T>();
            o.dof[ub] = ok ? frag_ld<T>(dout + (long)q * p.C + head * dh + ub * 16 + g4) : frag_zero<T>();
        }
        o.span = ok ? p.spans[(long)b * p.T + q] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            o.dlx[i] = (p.d_lam_ext && ok && (g4 + i) < E) ? p.d_lam_ext[(bp * p.T + q) * E + g4 + i] : 0.f;
        return o;
    };
    QOps qcur = load_q(0);
    for (int qt = 0; qt < NT; ++qt) {
        // compiler-level memory barrier: without it every loop-invariant LDS operand (intensity weights, key mask) is
        // hoisted out of the query loop and parked in ~190 extra registers
        asm volatile("" ::: "memory");
        const int q = qt * 16 + l15;
        const bool qok = q < p.T;
        const QOps qnext = load_q(qt + 1 < NT ? qt + 1 : qt);
        Frag4<T> qf[DT], dof[DT];
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) { qf[ub] = qcur.qf[ub]; dof[ub] = qcur.dof[ub]; }
        // ---- recompute S, P ---------------------------------------------------------------------
        f32x4 s[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4 a = zero4;
#pragma unroll
            for (int ub = 0; ub < DT; ++ub)
                a = mma16(frag_ld<T>(Ks + (kt * 16 + l15) * dh + ub * 16 + g4), qf[ub], a);
            s[kt] = a;
        }
        masked_softmax<NT>(s, km, cscale, lane);  // s = P^T, L(first=k, second=q)
        PH_MARK(1);   // q loads + S + softmax
        // From here on P lives in the activation dtype only (bf16: 2 registers per key tile instead of 4).  That is
        // the precision the forward pass used for P.T_ and A'.V anyway.
        Frag4<T> pf[NT];
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) pf[kt] = frag_from_acc<T>(s[kt]);
        __builtin_amdgcn_sched_barrier(0);
        // ---- H^T and the intensity MLP ----------------------------------------------------------
        Frag4<T> hf[DT];
        f32x4 hacc[DT];   // H^T in f32, L(first=u, second=q): used again by the softmax row-dot
        {
#pragma unroll
            for (int ut = 0; ut < DT; ++ut) {
                f32x4 a = zero4;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
                    a = mma16(kfrag<T>(Ts, dh, TTs, LDT, kt * 16, ut * 16, lane), pf[kt], a);
                hacc[ut] = a;
                hf[ut] = frag_from_acc<T>(a);
                if (qok) {  // Hin rows for kernel B (T-rounded, identical to what Z is computed from)
                    T* dst = reinterpret_cast<T*>(p.hin_ws) + (bp * p.T + q) * dh + ut * 16 + g4;
                    if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(&hf[ut]);
                    else *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(&hf[ut]);
                }
            }
        }
        PH_MARK(2);   // H
        const float span = qcur.span;
        // zq[e][d] = wv[j] * z (1 - z) for channel j = e*dh + d*16 + g4 + r (z = sigmoid output): all the du step needs.
        // Kept in the activation dtype (bf16: 2 registers per tile instead of 4).
        Frag4<T> zq[16][DT];
        float zp[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            zp[e] = 0.f;
            if constexpr (EC != 16) {
#pragma unroll
                for (int d = 0; d < DT; ++d) zq[e][d] = frag_zero<T>();
            }
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (EC == 16 || e < E) {
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const int jt = e * DT + d;
                    f32x4 a = zero4;
#pragma unroll
                    for (int ub = 0; ub < DT; ++ub)
                        a = mma16(frag_ld<T>(W1T + (jt * 16 + l15) * pd.LDW + ub * 16 + g4), hf[ub], a);
                    const float4 ws = *reinterpret_cast<const float4*>(w1s + jt * 16 + g4);
                    const float4 bs = *reinterpret_cast<const float4*>(b1s + jt * 16 + g4);
                    const float4 wv = *reinterpret_cast<const float4*>(wvs + jt * 16 + g4);
                    f32x4 zz, zw;
                    zz[0] = sigmoid_pre(a[0] + fmaf(span, ws.x, bs.x));
                    zz[1] = sigmoid_pre(a[1] + fmaf(span, ws.y, bs.y));
                    zz[2] = sigmoid_pre(a[2] + fmaf(span, ws.z, bs.z));
                    zz[3] = sigmoid_pre(a[3] + fmaf(span, ws.w, bs.w));
                    zw[0] = zz[0] * wv.x; zw[1] = zz[1] * wv.y; zw[2] = zz[2] * wv.z; zw[3] = zz[3] * wv.w;
                    zp[e] += (zw[0] + zw[1]) + (zw[2] + zw[3]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) zw[r] = zw[r] - zw[r] * zz[r];   // wv z (1 - z)
                    zq[e][d] = frag_from_acc<T>(zw);
                }
            }
            if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // at most four marks' operand loads in flight
        }
        PH_MARK(3);   // intensity MLP
        float z4[4], lam4[4], sg4[4];
        reduce_scatter16(zp, z4, lane);
        Frag4<T> lf;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float sc = scs[g4 + i], isc = iscs[g4 + i];
            const float x = z4[i] * isc;
            lam4[i] = sc * __logf(1.0f + __expf(x));
            sg4[i] = sigmoid_f(x);
            lf.v[i] = from_f32<T>(lam4[i]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- G, dA, dG -> dlambda, dP1, and dV accumulation ----------------------------------------
        Frag4<T> dOT[DT];  // L(first=q, second=v): A operand contracting over q
#pragma unroll
        for (int vt = 0; vt < DT; ++vt) dOT[vt] = frag_from_acc<T>(mma16(dof[vt], ident, zero4));
        PH_MARK(4);   // lambda, dOT
        Frag4<T> d1f[NT];     // dP through A' (dA * D * G'), activation dtype
        float rowdot = 0.f;   // sum_k dP[q][k] P[q][k]: this lane's part of  sum_k d1 * P
        f32x4 dlamT = zero4;  // L(first=e, second=q)
        const uint32_t dbase = (uint32_t)((bp * p.T + q) * p.T);   // dropout element index of (b', q, k=0)
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const f32x4 gacc = mma16(frag_ld<T>(Ms + (kt * 16 + l15) * EP + g4), lf, zero4);
            f32x4 da = zero4;
#pragma unroll
            for (int vb = 0; vb < DT; ++vb)
                da = mma16(frag_ld<T>(Vs + (kt * 16 + l15) * dh + vb * 16 + g4), dof[vb], da);
            f32x4 ap, dg, d1, gv = gacc;
            if (kt == qt) {   // only this key tile can hold k == q: G' diag := 1 (temporal.py:438-439)
#pragma unroll
                for (int r = 0; r < 4; ++r) gv[r] = (g4 + r == l15) ? 1.0f : gacc[r];
            }
            float facs[4] = {1.0f, 1.0f, 1.0f, 1.0f};
            if (dk.thresh != 0u) {   // same paired hash as the forward kernel
                const uint32_t h0 = drop_hash_pair(dk, dbase + kt * 16 + g4), h1 = drop_hash_pair(dk, dbase + kt * 16 + g4 + 2);
                facs[0] = (h0 & 0xffffu) >= dk.t16 ? dk.scale : 0.f;
                facs[1] = (h0 >> 16) >= dk.t16 ? dk.scale : 0.f;
                facs[2] = (h1 & 0xffffu) >= dk.t16 ? dk.scale : 0.f;
                facs[3] = (h1 >> 16) >= dk.t16 ? dk.scale : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = to_f32(pf[kt].v[r]);
                const float fac = facs[r];
                ap[r] = fac * gv[r] * pv;                 // A' = D*G'*P
                const float dad = da[r] * fac;
                dg[r] = dad * pv;
                d1[r] = dad * gv[r];                      // dP through A'
                rowdot = fmaf(d1[r], pv, rowdot);
            }
            if (kt == qt) {   // set_diag blocks the gradient into lambda
#pragma unroll
                for (int r = 0; r < 4; ++r) dg[r] = (g4 + r == l15) ? 0.f : dg[r];
            }
            d1f[kt] = frag_from_acc<T>(d1);
            dlamT = mma16(kfrag<T>(Ms, EP, MTs, LDT, kt * 16, 0, lane), frag_from_acc<T>(dg), dlamT);
            const Frag4<T> apT = frag_from_acc<T>(transpose_tile<T>(ap, ident));  // L(first=q, second=k)
#pragma unroll
            for (int vt = 0; vt < DT; ++vt) dVa[vt][kt] = mma16(dOT[vt], apT, dVa[vt][kt]);
            if (kt == 3) __builtin_amdgcn_sched_barrier(0);   // bound the live ranges: at most four key tiles interleaved
        }
        __builtin_amdgcn_sched_barrier(0);
        PH_MARK(5);   // G / dA / dV sweep
        // ---- dlambda -> dz, dscaling ---------------------------------------------------------------
        float dz4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float dl = dlamT[i];
            dl += qcur.dlx[i];   // upstream d lambda (0 when absent / padded)
            dz4[i] = dl * sg4[i];
            if (qok && (g4 + i) < E) dsc_acc[i] += dl * (lam4[i] - z4[i] * sg4[i]);
        }
        if (qok) *reinterpret_cast<float4*>(p.dz_ws + (bp * p.T + q) * EP + g4) = make_float4(dz4[0], dz4[1], dz4[2], dz4[3]);
        float dz16[16];
        all_gather16(dz4, dz16, lane);
        __builtin_amdgcn_sched_barrier(0);
        PH_MARK(6);   // dz
        // ---- du -> dH^T[u][q] = sum_j W1[u][j] du[q][j] -----------------------------------------------
        f32x4 dH[DT];
#pragma unroll
        for (int ut = 0; ut < DT; ++ut) dH[ut] = zero4;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            if (EC == 16 || e < E) {
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const int jt = e * DT + d;
                    const Frag4<T> zf = zq[e][d];
                    f32x4 du;
#pragma unroll
                    for (int r = 0; r < 4; ++r) du[r] = dz16[e] * to_f32(zf.v[r]);
                    const Frag4<T> duf = frag_from_acc<T>(du);
#pragma unroll
                    for (int ut = 0; ut < DT; ++ut)
                        dH[ut] = mma16(frag_ld<T>(W1R + (ut * 16 + l15) * pd.LDR + jt * 16 + g4), duf, dH[ut]);
                }
            }
            if ((e & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        PH_MARK(7);   // du / dH
        // ---- dP = d1 + dH.T_^T ; dS = P*(dP - rowsum(dP*P)) * c -----------------------------------------
        // rowsum(dP*P) = sum_k d1*P + sum_k P[q][k] (dH[q].T_[k]) = sum_k d1*P + dH[q].H[q]   (H = P.T_), so the row
        // term is known before the key sweep and dP never has to be kept for all key tiles.
        Frag4<T> dhf[DT];
#pragma unroll
        for (int ut = 0; ut < DT; ++ut) {
            dhf[ut] = frag_from_acc<T>(dH[ut]);
#pragma unroll
            for (int r = 0; r < 4; ++r) rowdot = fmaf(dH[ut][r], hacc[ut][r], rowdot);
        }
        rowdot = group_sum4(rowdot);
        // transposed operands for the query-contracting products
        Frag4<T> QT[DT], dHT[DT];
#pragma unroll
        for (int ut = 0; ut < DT; ++ut) {
            QT[ut] = frag_from_acc<T>(mma16(qf[ut], ident, zero4));
            dHT[ut] = frag_from_acc<T>(transpose_tile<T>(dH[ut], ident));
        }
        f32x4 dQ[DT];
#pragma unroll
        for (int ut = 0; ut < DT; ++ut) dQ[ut] = zero4;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            f32x4 a;
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r] = to_f32(d1f[kt].v[r]);
#pragma unroll
            for (int ub = 0; ub < DT; ++ub)
                a = mma16(frag_ld<T>(Ts + (kt * 16 + l15) * dh + ub * 16 + g4), dhf[ub], a);
            f32x4 ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // tf.where(mask==0, paddings, S) (temporal.py:425-426) passes no gradient to a padded key's
                // score; P is non-zero there only for fully padded rows (uniform softmax)
                const bool padded = (km.pad >> (kt * 4 + r)) & 1u;
                ds[r] = padded ? 0.f : to_f32(pf[kt].v[r]) * (a[r] - rowdot) * cscale;
            }
            const Frag4<T> dsf = frag_from_acc<T>(ds);
#pragma unroll
            for (int ut = 0; ut < DT; ++ut)
                dQ[ut] = mma16(kfrag<T>(Ks, dh, KTs, LDT, kt * 16, ut * 16, lane), dsf, dQ[ut]);
            const Frag4<T> dsT = frag_from_acc<T>(transpose_tile<T>(ds, ident));
            const Frag4<T> pT = frag_from_acc<T>(mma16(pf[kt], ident, zero4));   // P^T tile: L(first=q, second=k)
#pragma unroll
            for (int ut = 0; ut < DT; ++ut) {
                dKa[ut][kt] = mma16(QT[ut], dsT, dKa[ut][kt]);
                dTa[ut][kt] = mma16(dHT[ut], pT, dTa[ut][kt]);
            }
            if (kt == 3) __builtin_amdgcn_sched_barrier(0);
        }
        if (qok) {
#pragma unroll
            for (int ut = 0; ut < DT; ++ut) st_frag<T>(dqkvt + (long)q * ldq + head * dh + ut * 16 + g4, dQ[ut]);
        }
        PH_MARK(8);   // dS sweep, dQ, dK, dT
        qcur = qnext;
    }
    // ---- write dK / dV / dT_ (L(first=u, second=k): 4 consecutive channels of key row k) ----------
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const int k = kt * 16 + l15;
        if (k < p.T) {
#pragma unroll
            for (int ut = 0; ut < DT; ++ut) {
                T* row = dqkvt + (long)k * ldq + head * dh + ut * 16 + g4;
                st_frag<T>(row + p.C, dKa[ut][kt]);
                st_frag<T>(row + 2 * p.C, dVa[ut][kt]);
                st_frag<T>(row + 3 * p.C, dTa[ut][kt]);
            }
        }
    }
    PH_MARK(9);   // epilogue stores
    PH_FLUSH();
    // ---- dscaling partial: sum over the 16 query lanes -----------------------------------------
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float v = dsc_acc[i];
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        if (l15 == 0) p.dsc_part[job * EP + g4 + i] = v;
    }
}

// ---------------- kernel B: intensity weight gradients ---------------------------------------------
struct WgP {
    const void* hin_ws; const float* dz_ws; const float* spans; const char* pack;
    long R; int B, T, E; float* wpart; const float* dsc_part; long njobs;
};

template <typename T, int DT>
__global__ __launch_bounds__(256) void intensity_wgrad_kernel(WgP p) {
    constexpr int dh = 16 * DT;
    constexpr int ECH = 4;  // marks per workgroup row (gridDim.y = 16/ECH): 4*DT*DT accumulator tiles per wave
    const int e0 = blockIdx.y * ECH;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PackDims pd = pack_dims<T>(dh, p.E);
    {
        const uint4* src = reinterpret_cast<const uint4*>(p.pack);
        uint4* dst = reinterpret_cast<uint4*>(smem);
        for (int i = threadIdx.x; i < (int)(pd.bytes / 16); i += blockDim.x) dst[i] = src[i];
    }
    const int JE = pd.JE, NPAR = (dh + 3) * JE, NPARX = NPAR + EP;
    float* accs = reinterpret_cast<float*>(smem + pd.bytes);  // [NPAR + 16] block accumulator
    for (int i = threadIdx.x; i < NPARX; i += blockDim.x) accs[i] = 0.f;
    const T* W1T = reinterpret_cast<const T*>(smem);
    const float* fW = reinterpret_cast<const float*>(smem + pd.off_f32);
    const float* w1s = fW; const float* b1s = fW + JE; const float* wvs = fW + 2 * JE;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const Frag4<T> ident = identity_frag<T>(lane);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // row tiles follow kernel A's (b', query tile) split: 16 consecutive queries of ONE b' = head*B + b, so the
    // interval of a row needs no per-row division
    const int ntq = (p.T + 15) / 16;
    const int ntile = (int)(p.R / p.T) * ntq;
    const T* hin = reinterpret_cast<const T*>(p.hin_ws);

    f32x4 dW[ECH][DT][DT];  // [e-e0][d][ub]: tile (j-tile = e*DT+d, u-tile = ub), L(first=j, second=u)
    float adb[ECH][DT], adws[ECH][DT], adw[ECH][DT];
#pragma unroll
    for (int e = 0; e < ECH; ++e)
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            adb[e][d] = 0.f; adws[e][d] = 0.f; adw[e][d] = 0.f;
#pragma unroll
            for (int ub = 0; ub < DT; ++ub) dW[e][d][ub] = zero4;
        }

    for (int t = (int)blockIdx.x * 4 + wave; t < ntile; t += (int)gridDim.x * 4) {
        const int bpq = t / ntq, qt = t - bpq * ntq, bb = bpq % p.B;
        const long row0 = (long)bpq * p.T + qt * 16;
        const bool okA = qt * 16 + l15 < p.T;   // row on the lane axis (A operand)
        Frag4<T> hA[DT], hB[DT];
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) {
            hA[ub] = okA ? frag_ld<T>(hin + (row0 + l15) * dh + ub * 16 + g4) : frag_zero<T>();
            hB[ub] = frag_from_acc<T>(mma16(hA[ub], ident, zero4));  // L(first=row, second=u)
        }
        float spn[4];
        float4 dz4[4];   // dz[row g4+r][e0 .. e0+3]
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int qq = qt * 16 + g4 + r;
            const bool ok = qq < p.T;
            spn[r] = ok ? p.spans[(long)bb * p.T + qq] : 0.f;
            dz4[r] = ok ? *reinterpret_cast<const float4*>(p.dz_ws + (row0 + g4 + r) * EP + e0) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int ee = 0; ee < ECH; ++ee) {
            const int e = e0 + ee;
            if (e < p.E) {
                float dzr[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) dzr[r] = ee == 0 ? dz4[r].x : ee == 1 ? dz4[r].y : ee == 2 ? dz4[r].z : dz4[r].w;
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const int jt = e * DT + d, j = jt * 16 + l15;
                    f32x4 a = zero4;  // Zpre[row][j], L(first=row, second=j)
#pragma unroll
                    for (int ub = 0; ub < DT; ++ub)
                        a = mma16(hA[ub], frag_ld<T>(W1T + (jt * 16 + l15) * pd.LDW + ub * 16 + g4), a);
                    const float ws = w1s[j], bs = b1s[j], wv = wvs[j];
                    f32x4 du;
                    float sdb = 0.f, sdws = 0.f, sdw = 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = sigmoid_pre(a[r] + fmaf(spn[r], ws, bs));
                        const float t2 = dzr[r] * z;
                        du[r] = t2 * wv * (1.0f - z);
                        sdb += du[r]; sdws += du[r] * spn[r]; sdw += t2;
                    }
                    adb[ee][d] += sdb; adws[ee][d] += sdws; adw[ee][d] += sdw;
                    const Frag4<T> duf = frag_from_acc<T>(du);  // as A operand: A[m=j][kk=row]
#pragma unroll
                    for (int ub = 0; ub < DT; ++ub) dW[ee][d][ub] = mma16(duf, hB[ub], dW[ee][d][ub]);
                }
            }
        }
    }
    // ---- block reduction: waves take turns adding into the LDS accumulator (deterministic) -----------
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int ee = 0; ee < ECH; ++ee) {
                const int e = e0 + ee;
                if (e < p.E) {
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        const int jt = e * DT + d;
                        const float sb = group_sum4(adb[ee][d]), sws = group_sum4(adws[ee][d]), sw = group_sum4(adw[ee][d]);
                        if (lane < 16) {
                            const int j = jt * 16 + l15;
                            accs[dh * JE + j] += sws;          // dW1[dh][j]   (interval row)
                            accs[(dh + 1) * JE + j] += sb;     // db1[j]
                            accs[(dh + 2) * JE + j] += sw;     // dw.flatten()[j]
                        }
#pragma unroll
                        for (int ub = 0; ub < DT; ++ub)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int j = jt * 16 + g4 + r, u = ub * 16 + l15;
                                accs[u * JE + j] += dW[ee][d][ub][r];  // dW1[u][j]
                            }
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < NPAR; i += blockDim.x) {
        const int e = (i % JE) / dh;  // every entry belongs to exactly one mark e -> one blockIdx.y
        if (e >= e0 && e < e0 + ECH) p.wpart[(long)blockIdx.x * NPARX + i] = accs[i];
    }
    // dscaling: fold this block's slice of kernel A's per-(b,head) partials into the same partial row
    if (blockIdx.y == 0 && threadIdx.x < EP) {
        const long per = (p.njobs + gridDim.x - 1) / gridDim.x;
        const long j0 = blockIdx.x * per, j1 = min(p.njobs, j0 + per);
        float a = 0.f;
        for (long j = j0; j < j1; ++j) a += p.dsc_part[j * EP + threadIdx.x];
        p.wpart[(long)blockIdx.x * NPARX + NPAR + threadIdx.x] = a;
    }
}

struct WsLayout { size_t hin, dz, dsc, wpart, total; };
template <typename T>
WsLayout ws_layout(int B, int T_, int C, int H, int E) {
    const int dh = C / H;
    const size_t R = (size_t)B * H * T_;
    WsLayout w;
    size_t o = 0;
    w.hin = o; o += (R * dh * sizeof(T) + 255) & ~(size_t)255;
    w.dz = o; o += (R * EP * sizeof(float) + 255) & ~(size_t)255;
    w.dsc = o; o += ((size_t)B * H * EP * sizeof(float) + 255) & ~(size_t)255;
    w.wpart = o; o += ((size_t)KB_BLOCKS * ((dh + 3) * dh * E + EP) * sizeof(float) + 255) & ~(size_t)255;
    w.total = o;
    return w;
}

template <typename T, int DT, int NT>
int launch_bwd(BwdP p, char* ws, float* dW1, float* db1, float* dw, float* dscaling, hipStream_t st) {
    constexpr int dh = 16 * DT, Tp = 16 * NT, LDT = Tp + 4;
    const PackDims pd = pack_dims<T>(dh, p.E);
    const WsLayout wl = ws_layout<T>(p.B, p.T, p.C, p.H, p.E);
    p.hin_ws = ws + wl.hin; p.dz_ws = reinterpret_cast<float*>(ws + wl.dz);
    p.dsc_part = reinterpret_cast<float*>(ws + wl.dsc); p.wpart = reinterpret_cast<float*>(ws + wl.wpart);
    const size_t wave_bytes = (3 * (size_t)Tp * dh + (size_t)Tp * EP + (sizeof(T) == 2 ? 0 : 2 * (size_t)dh * LDT + (size_t)EP * LDT)) * sizeof(T) +
                              (size_t)Tp * sizeof(float);
    int waves = 4;
    while (waves > 1 && pd.bytes + waves * wave_bytes > 80 * 1024) waves >>= 1;
    const size_t smem = pd.bytes + waves * wave_bytes;
    EDGL_REQUIRE(smem <= 160 * 1024, EDGL_ERR_SHAPE, "edgl_bimau_bwd: needs %zu B of LDS", smem);
    p.waves = waves;
    auto kern = p.E == 16 ? bimau_bwd_kernel<T, DT, NT, 16> : bimau_bwd_kernel<T, DT, NT, 0>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const long jobs = (long)p.B * p.H;
    edgl_prof_begin(EDGL_KERNEL_BIMAU_BWD, st);
    hipLaunchKernelGGL(kern, dim3((unsigned)((jobs + waves - 1) / waves)), dim3(64 * waves), smem, st, p);
    edgl_prof_end(EDGL_KERNEL_BIMAU_BWD, st);
    EDGL_LAUNCH_CHECK();

    WgP wp{p.hin_ws, p.dz_ws, p.spans, p.pack, (long)p.B * p.H * p.T, p.B, p.T, p.E, p.wpart, p.dsc_part, jobs};
    const int JE = dh * p.E, NPAR = (dh + 3) * JE, NPARX = NPAR + EP;
    const size_t smem_b = pd.bytes + (size_t)NPARX * sizeof(float);
    EDGL_REQUIRE(smem_b <= 160 * 1024, EDGL_ERR_SHAPE, "edgl_bimau_bwd: weight-grad kernel needs %zu B of LDS", smem_b);
    auto kb = intensity_wgrad_kernel<T, DT>;
    hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b);
    hipLaunchKernelGGL(kb, dim3(KB_BLOCKS, 4), dim3(256), smem_b, st, wp);   // gridDim.y = 16 marks / ECH
    EDGL_LAUNCH_CHECK();
    if (db1 == dW1 + (dh + 1) * JE && dw == db1 + JE && dscaling == dw + JE) {   // flat-arena layout: one reduction
        return edgl_reduce_rows(p.wpart, KB_BLOCKS, NPAR + p.E, NPARX, dW1, 0, st);
    }
    int rc = edgl_reduce_rows(p.wpart, KB_BLOCKS, (dh + 1) * JE, NPARX, dW1, 0, st);
    if (rc) return rc;
    rc = edgl_reduce_rows(p.wpart + (dh + 1) * JE, KB_BLOCKS, JE, NPARX, db1, 0, st);
    if (rc) return rc;
    rc = edgl_reduce_rows(p.wpart + (dh + 2) * JE, KB_BLOCKS, JE, NPARX, dw, 0, st);
    if (rc) return rc;
    rc = edgl_reduce_rows(p.wpart + NPAR, KB_BLOCKS, p.E, NPARX, dscaling, 0, st);
    if (rc) return rc;
    return EDGL_OK;
}

template <typename T, int DT>
int dispatch_nt(BwdP p, char* ws, float* dW1, float* db1, float* dw, float* dsc, hipStream_t st) {
    switch ((p.T + 15) / 16) {
        case 1: return launch_bwd<T, DT, 1>(p, ws, dW1, db1, dw, dsc, st);
        case 2: return launch_bwd<T, DT, 2>(p, ws, dW1, db1, dw, dsc, st);
        case 3: return launch_bwd<T, DT, 3>(p, ws, dW1, db1, dw, dsc, st);
        case 4: return launch_bwd<T, DT, 4>(p, ws, dW1, db1, dw, dsc, st);
        case 5: return launch_bwd<T, DT, 5>(p, ws, dW1, db1, dw, dsc, st);
        case 6: return launch_bwd<T, DT, 6>(p, ws, dW1, db1, dw, dsc, st);
        case 7: return launch_bwd<T, DT, 7>(p, ws, dW1, db1, dw, dsc, st);
        case 8: return launch_bwd<T, DT, 8>(p, ws, dW1, db1, dw, dsc, st);
    }
    edgl_set_error("edgl_bimau_bwd: T=%d not supported (T <= 128)", p.T);
    return EDGL_ERR_SHAPE;
}

}  // namespace

extern "C" long edgl_bimau_bwd_workspace(int B, int T, int C, int H, int E, int dtype) {
    if (H <= 0 || C % H) return -1;
    return (long)(dtype == EDGL_BF16 ? ws_layout<bf16>(B, T, C, H, E).total : ws_layout<float>(B, T, C, H, E).total);
}

extern "C" int edgl_bimau_bwd(const void* qkvt, const int64_t* ids, const float* spans, const uint8_t* marks,
                              const void* pack, const void* d_out, const float* d_lam_ext, int B, int T, int C, int H,
                              int E, float drop_rate, const uint64_t* rng_state, uint32_t stream_id, void* d_qkvt,
                              float* dW1, float* db1, float* dw, float* dscaling, void* workspace, int dtype,
                              void* stream) {
    EDGL_REQUIRE(qkvt && ids && spans && marks && pack && d_out && d_qkvt && dW1 && db1 && dw && dscaling && workspace,
                 EDGL_ERR_NULL, "edgl_bimau_bwd: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0 && H > 0 && C % H == 0 && E >= 1 && E <= bimau::EP, EDGL_ERR_SHAPE,
                 "edgl_bimau_bwd: bad shape B=%d T=%d C=%d H=%d E=%d", B, T, C, H, E);
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_bimau_bwd: dropout without rng_state");
    EDGL_REQUIRE((double)B * H * T * T < 4294967296.0, EDGL_ERR_SHAPE, "edgl_bimau_bwd: H*B*T*T must be < 2^32");
    BwdP p{};
    p.qkvt = qkvt; p.ids = ids; p.spans = spans; p.marks = marks; p.pack = (const char*)pack; p.d_out = d_out;
    p.d_lam_ext = d_lam_ext; p.B = B; p.T = T; p.C = C; p.H = H; p.E = E; p.rate = drop_rate; p.rng = rng_state;
    p.stream_id = stream_id; p.d_qkvt = d_qkvt;
    hipStream_t st = (hipStream_t)stream;
    const int dh = C / H;
    char* ws = (char*)workspace;
    if (dtype == EDGL_F32) {
        if (dh == 16) return dispatch_nt<float, 1>(p, ws, dW1, db1, dw, dscaling, st);
        if (dh == 32) return dispatch_nt<float, 2>(p, ws, dW1, db1, dw, dscaling, st);
    } else if (dtype == EDGL_BF16) {
        if (dh == 16) return dispatch_nt<bf16, 1>(p, ws, dW1, db1, dw, dscaling, st);
        if (dh == 32) return dispatch_nt<bf16, 2>(p, ws, dW1, db1, dw, dscaling, st);
    } else {
        edgl_set_error("edgl_bimau_bwd: bad dtype %d", dtype);
        return EDGL_ERR_DTYPE;
    }
    edgl_set_error("edgl_bimau_bwd: head dim %d not supported (16 or 32)", dh);
    return EDGL_ERR_SHAPE;
}

#ifdef EDGL_PHASE_TIMING
extern "C" int edgl_debug_phase_cycles(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

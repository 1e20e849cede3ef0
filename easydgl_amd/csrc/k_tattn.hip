// K11: causal multi-head attention with a time feature map — TfMultiHeadAttention.__call__ (temporal.py:126-184, TGAT)
// with TimeFunctionCoding.code (coding.py:104-122) folded into the operands.
//
// The reference scores a (query q, key k) pair with
//     S[q,k] = sum_d Q[q,d] * ( K[k,d] + Pos[k,d] + cos(dt[q,k]*w_d + phi_d) ) / sqrt(dh),   dt[q,k] = max(t[q+1] - t[k], 0)
// and materialises the cosine as a [B,T,T,C] tensor.  For every pair the causal + key masks keep (k <= q, key not padding)
// the clamp is inactive when timestamps do not decrease, and cos(a-b) = cos a cos b + sin a sin b turns the time term
// into two more inner products:
//     q~[q] = [ Q | Q*cos(a_q w + phi) | Q*sin(a_q w + phi) ],  k~[k] = [ K + Pos | cos(b_k w) | sin(b_k w) ],  S = q~ . k~^T
// with a_q = t[q+1] - base, b_k = t[k] - base (base = the sequence's last timestamp, so the arguments stay as small as the
// reference's).  T*C transcendentals per sequence instead of T*T*C, and the T*T*C contraction runs on the matrix cores.
// timefn_fwd counts sequences whose timestamps decrease at an unmasked position (the host checks the counter).
//
// The attention itself is generic in (Dq, Dv): one wave per (sample, head, 16-row tile); operands are read straight from
// L2 as 4-element row chunks (A/B fragments of the 16x16 MFMA); a tile that has to be contracted along its rows is turned
// with one MFMA against the identity (edgl_common.h transpose_tile) instead of an LDS round trip.
//   fwd   : per query tile, online softmax over the key tiles, O^T += V^T . P^T; saves (max, sum) per row and O in f32
//   bwd_q : per query tile, D = dO.O, dS = P*(dP - D), dQ~^T += K~^T . dS^T
//   bwd_k : per key tile, loops the query tiles, dV^T += dO^T . A, dK~^T += Q~^T . dS
// Fully masked query rows (left padding) are uniform over ALL keys in the reference (every score is replaced by the same
// constant, temporal.py:158,166) and the joint LayerNorm that follows reads them, so no tile is skipped.
#include <algorithm>

#include "edgl_common.h"

namespace {

constexpr int TA_CAUSAL = 1;
constexpr float PADV = -4294967296.0f;   // float32(-2**32 + 1)

struct TaP {
    const void *qx, *kx, *v, *resid;
    int ldq, ldk, ldv, ldr;
    const int64_t* ids;
    int B, T, H, Dq, Dv;
    float cscale, rate;
    const uint64_t* rng; uint32_t stream_id;
    void* out; int ldo;
    float *st_m, *st_l, *st_d, *oatt;     // [H*B*T] each, oatt [B,T,H*Dv] f32
    const void* d_out; int ld_do;
    void *d_qx, *d_kx, *d_v; int ld_dq, ld_dk, ld_dv;
    int flags, NT;
};

template <typename T> __device__ __forceinline__ void frag_st(T* dst, const Frag4<T>& f) {
    if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(&f);
    else *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(&f);
}
// registers of a row-chunk load (lane = row l&15, 4 columns at (l>>4)*4) -> the same tile with rows on (l>>4)*4+r
template <typename T> __device__ __forceinline__ Frag4<T> turn(const Frag4<T>& f, const Frag4<T>& ident) {
    return frag_from_acc<T>(mma16(f, ident, f32x4{0.f, 0.f, 0.f, 0.f}));
}

struct Job { int b, head, tile; long bp; };
__device__ __forceinline__ bool get_job(const TaP& p, Job& j) {
    const long job = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (job >= (long)p.B * p.H * p.NT) return false;
    j.tile = (int)(job % p.NT);
    const long bh = job / p.NT;
    j.head = (int)(bh % p.H);
    j.b = (int)(bh / p.H);
    j.bp = (long)j.head * p.B + j.b;   // head-major b' (temporal.py:139-141)
    return true;
}

// First position of the sequence with ids != 0 (T if none).  Rows before it are fully masked: every score is the same
// replaced constant and the softmax is uniform over ALL keys, so their tiles must be computed.  Every later row has an
// unmasked key, its replaced scores underflow to exactly 0 after the softmax, and key tiles above the diagonal contribute
// exactly nothing — they are skipped (half of the causal work).
__device__ __forceinline__ int first_unpadded(const int64_t* idr, int T, int lane) {
    int f = T;
    for (int k = lane; k < T; k += 64)
        if (idr[k] != 0) f = min(f, k);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) f = min(f, __shfl_xor(f, o, 64));
    return f;
}

// ---- forward -----------------------------------------------------------------------------------------------------------
template <typename T, int DQT, int DVT>
__device__ __forceinline__ void tattn_fwd_body(const TaP& p);
template <typename T, int DQT, int DVT>
__global__ __launch_bounds__(256) void tattn_fwd_kernel(TaP p) { tattn_fwd_body<T, DQT, DVT>(p); }
// same body under a 128-register budget (4 waves per SIMD): with one head of 128 channels (TGAT, h = 1) a launch has only
// B * T/16 waves — 3584 at the headline sizes — and 3 waves per SIMD leave a second, nearly empty round
template <typename T, int DQT, int DVT>
__global__ __launch_bounds__(256, 4) void tattn_fwd_kernel_w4(TaP p) { tattn_fwd_body<T, DQT, DVT>(p); }

template <typename T, int DQT, int DVT>
__device__ __forceinline__ void tattn_fwd_body(const TaP& p) {
    Job j;
    if (!get_job(p, j)) return;
    const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int q = j.tile * 16 + l15, qc = min(q, p.T - 1);
    const bool qok = q < p.T, causal = (p.flags & TA_CAUSAL) != 0;
    const T* Qr = reinterpret_cast<const T*>(p.qx) + ((long)j.b * p.T + qc) * p.ldq + j.head * p.Dq;
    const T* Kb = reinterpret_cast<const T*>(p.kx) + (long)j.b * p.T * p.ldk + j.head * p.Dq;
    const T* Vb = reinterpret_cast<const T*>(p.v) + (long)j.b * p.T * p.ldv + j.head * p.Dv;
    const int64_t* idr = p.ids + (long)j.b * p.T;
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const Frag4<T> ident = identity_frag<T>(lane);
    const uint32_t dbase = (uint32_t)((j.bp * p.T + qc) * p.T);
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc[DVT];
#pragma unroll
    for (int ut = 0; ut < DVT; ++ut) acc[ut] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kt_end = (causal && j.tile * 16 >= first_unpadded(idr, p.T, lane)) ? j.tile + 1 : p.NT;
    // compile-time contraction length: the operand loads of a tile are issued together instead of one per dependent MFMA
    Frag4<T> qf[DQT];
#pragma unroll
    for (int dt = 0; dt < DQT; ++dt) qf[dt] = frag_ld<T>(Qr + dt * 16 + g4);
    for (int kt = 0; kt < kt_end; ++kt) {
        const int kr = min(kt * 16 + l15, p.T - 1);
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        Frag4<T> kf[DQT];
#pragma unroll
        for (int dt = 0; dt < DQT; ++dt) kf[dt] = frag_ld<T>(Kb + (long)kr * p.ldk + dt * 16 + g4);
        Frag4<T> vraw[DVT];
#pragma unroll
        for (int ut = 0; ut < DVT; ++ut) vraw[ut] = frag_ld<T>(Vb + (long)kr * p.ldv + ut * 16 + g4);
#pragma unroll
        for (int dt = 0; dt < DQT; ++dt) s = mma16(kf[dt], qf[dt], s);
        float x[4], tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kt * 16 + g4 + r;
            const float madd = k >= p.T ? -INFINITY : (idr[min(k, p.T - 1)] == 0 ? PADV : 0.f);   // temporal.py:153-158
            float v = fmaf(s[r], p.cscale, madd);
            if (causal && k > q && k < p.T) v = PADV;                                            // :161-166
            x[r] = v;
            tmax = fmaxf(tmax, v);
        }
        tmax = group_max4(tmax);
        const float m_new = fmaxf(m_run, tmax), corr = __expf(m_run - m_new);
        f32x4 e;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = __expf(x[r] - m_new); psum += e[r]; }
        l_run = l_run * corr + group_sum4(psum);
        m_run = m_new;
        if (dk.thresh != 0u) {   // :172 — the keep mask; the 1/(1-rate) factor is applied once in the epilogue
            const uint32_t h0 = drop_hash_pair(dk, dbase + kt * 16 + g4), h1 = drop_hash_pair(dk, dbase + kt * 16 + g4 + 2);
            e[0] = (h0 & 0xffffu) >= dk.t16 ? e[0] : 0.f;
            e[1] = (h0 >> 16) >= dk.t16 ? e[1] : 0.f;
            e[2] = (h1 & 0xffffu) >= dk.t16 ? e[2] : 0.f;
            e[3] = (h1 >> 16) >= dk.t16 ? e[3] : 0.f;
        }
        const Frag4<T> pf = frag_from_acc<T>(e);
#pragma unroll
        for (int ut = 0; ut < DVT; ++ut) {
            const Frag4<T> vt = turn<T>(vraw[ut], ident);   // V[k][u] with k on the registers
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ut][r] *= corr;
            acc[ut] = mma16(vt, pf, acc[ut]);                                                          // :175
        }
    }
    const float inv = dk.scale / l_run;
    if (p.st_m && qok && g4 == 0) { p.st_m[j.bp * p.T + q] = m_run; p.st_l[j.bp * p.T + q] = l_run; }
    if (!qok) return;
    const long orow = (long)j.b * p.T + q;
#pragma unroll
    for (int ut = 0; ut < DVT; ++ut) {
        const int col = j.head * p.Dv + ut * 16 + g4;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = acc[ut][r] * inv;
        if (p.oatt) *reinterpret_cast<float4*>(p.oatt + orow * ((long)p.H * p.Dv) + col) = make_float4(o[0], o[1], o[2], o[3]);
        const Frag4<T> rf = frag_ld<T>(reinterpret_cast<const T*>(p.resid) + orow * p.ldr + col);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += to_f32(rf.v[r]);                                           // :181 residual
        frag_st<T>(reinterpret_cast<T*>(p.out) + orow * p.ldo + col, frag_from_acc<T>(o));
    }
}

// ---- backward, query side ---------------------------------------------------------------------------------------------
template <typename T, int DQT, int DVT>
__global__ __launch_bounds__(256) void tattn_bwd_q_kernel(TaP p) {
    Job j;
    if (!get_job(p, j)) return;
    const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int q = j.tile * 16 + l15, qc = min(q, p.T - 1);
    const bool qok = q < p.T, causal = (p.flags & TA_CAUSAL) != 0;
    const T* Qr = reinterpret_cast<const T*>(p.qx) + ((long)j.b * p.T + qc) * p.ldq + j.head * p.Dq;
    const T* Kb = reinterpret_cast<const T*>(p.kx) + (long)j.b * p.T * p.ldk + j.head * p.Dq;
    const T* Vb = reinterpret_cast<const T*>(p.v) + (long)j.b * p.T * p.ldv + j.head * p.Dv;
    const T* dOr = reinterpret_cast<const T*>(p.d_out) + ((long)j.b * p.T + qc) * p.ld_do + j.head * p.Dv;
    const int64_t* idr = p.ids + (long)j.b * p.T;
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const Frag4<T> ident = identity_frag<T>(lane);
    const uint32_t dbase = (uint32_t)((j.bp * p.T + qc) * p.T);
    // D[q] = sum_k dA[q,k] A[q,k] = dO[q] . O_att[q]
    float dsum = 0.f;
    {
        const float* oa = p.oatt + ((long)j.b * p.T + qc) * ((long)p.H * p.Dv) + j.head * p.Dv;
        for (int u = 0; u < p.Dv; u += 16) {
            const Frag4<T> g = frag_ld<T>(dOr + u + g4);
            const float4 o = *reinterpret_cast<const float4*>(oa + u + g4);
            dsum += to_f32(g.v[0]) * o.x + to_f32(g.v[1]) * o.y + to_f32(g.v[2]) * o.z + to_f32(g.v[3]) * o.w;
        }
        dsum = group_sum4(dsum);
        if (qok && g4 == 0) p.st_d[j.bp * p.T + q] = dsum;
    }
    const float m = p.st_m[j.bp * p.T + qc], invl = 1.0f / p.st_l[j.bp * p.T + qc];
    f32x4 acc[DQT];
#pragma unroll
    for (int dt = 0; dt < DQT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kt_end = (causal && j.tile * 16 >= first_unpadded(idr, p.T, lane)) ? j.tile + 1 : p.NT;
    Frag4<T> qf[DQT], gf[DVT];
#pragma unroll
    for (int dt = 0; dt < DQT; ++dt) qf[dt] = frag_ld<T>(Qr + dt * 16 + g4);
#pragma unroll
    for (int ut = 0; ut < DVT; ++ut) gf[ut] = frag_ld<T>(dOr + ut * 16 + g4);
    for (int kt = 0; kt < kt_end; ++kt) {
        const int kr = min(kt * 16 + l15, p.T - 1);
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
        Frag4<T> kf[DQT], vf[DVT];
#pragma unroll
        for (int dt = 0; dt < DQT; ++dt) kf[dt] = frag_ld<T>(Kb + (long)kr * p.ldk + dt * 16 + g4);
#pragma unroll
        for (int ut = 0; ut < DVT; ++ut) vf[ut] = frag_ld<T>(Vb + (long)kr * p.ldv + ut * 16 + g4);
#pragma unroll
        for (int dt = 0; dt < DQT; ++dt) s = mma16(kf[dt], qf[dt], s);
#pragma unroll
        for (int ut = 0; ut < DVT; ++ut) da = mma16(vf[ut], gf[ut], da);
        uint32_t h0 = 0xffffffffu, h1 = 0xffffffffu;
        if (dk.thresh != 0u) { h0 = drop_hash_pair(dk, dbase + kt * 16 + g4); h1 = drop_hash_pair(dk, dbase + kt * 16 + g4 + 2); }
        const uint32_t hb[4] = {h0 & 0xffffu, h0 >> 16, h1 & 0xffffu, h1 >> 16};
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kt * 16 + g4 + r;
            const bool pad = k >= p.T || idr[min(k, p.T - 1)] == 0, fut = causal && k > q;
            float v = fmaf(s[r], p.cscale, k >= p.T ? -INFINITY : (pad ? PADV : 0.f));
            if (fut && k < p.T) v = PADV;
            const float P = __expf(v - m) * invl;
            const float dP = (dk.thresh == 0u || hb[r] >= dk.t16) ? da[r] * dk.scale : 0.f;
            // a replaced score is a constant (tf.where): no gradient, also in a fully masked (uniform) row
            ds[r] = (pad || fut || !qok) ? 0.f : P * (dP - dsum) * p.cscale;
        }
        const Frag4<T> dsf = frag_from_acc<T>(ds);
#pragma unroll
        for (int dt = 0; dt < DQT; ++dt) acc[dt] = mma16(turn<T>(kf[dt], ident), dsf, acc[dt]);
    }
    if (!qok) return;
    T* dst = reinterpret_cast<T*>(p.d_qx) + ((long)j.b * p.T + q) * p.ld_dq + j.head * p.Dq;
#pragma unroll
    for (int dt = 0; dt < DQT; ++dt) frag_st<T>(dst + dt * 16 + g4, frag_from_acc<T>(acc[dt]));
}

// ---- backward, key side -----------------------------------------------------------------------------------------------
template <typename T, int DQT, int DVT>
__global__ __launch_bounds__(256) void tattn_bwd_k_kernel(TaP p) {
    Job j;
    if (!get_job(p, j)) return;
    const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int k = j.tile * 16 + l15, kc = min(k, p.T - 1);
    const bool kok = k < p.T, causal = (p.flags & TA_CAUSAL) != 0;
    const T* Qb = reinterpret_cast<const T*>(p.qx) + (long)j.b * p.T * p.ldq + j.head * p.Dq;
    const T* Kr = reinterpret_cast<const T*>(p.kx) + ((long)j.b * p.T + kc) * p.ldk + j.head * p.Dq;
    const T* Vr = reinterpret_cast<const T*>(p.v) + ((long)j.b * p.T + kc) * p.ldv + j.head * p.Dv;
    const T* dOb = reinterpret_cast<const T*>(p.d_out) + (long)j.b * p.T * p.ld_do + j.head * p.Dv;
    const bool pad = !kok || p.ids[(long)j.b * p.T + kc] == 0;
    const float madd = !kok ? -INFINITY : (pad ? PADV : 0.f);
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const Frag4<T> ident = identity_frag<T>(lane);
    Frag4<T> kf[DQT], vf[DVT];
#pragma unroll
    for (int dt = 0; dt < DQT; ++dt) kf[dt] = frag_ld<T>(Kr + dt * 16 + g4);
#pragma unroll
    for (int ut = 0; ut < DVT; ++ut) vf[ut] = frag_ld<T>(Vr + ut * 16 + g4);
    f32x4 accK[DQT], accV[DVT];
#pragma unroll
    for (int dt = 0; dt < DQT; ++dt) accK[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ut = 0; ut < DVT; ++ut) accV[ut] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fnp = first_unpadded(p.ids + (long)j.b * p.T, p.T, lane);
    for (int qt = 0; qt < p.NT; ++qt) {
        if (causal && qt < j.tile && qt * 16 >= fnp) continue;   // all 16 queries see this key tile as future: exact zeros
        const int ql = min(qt * 16 + l15, p.T - 1);   // this lane's row when it loads an operand chunk
        const T* Qrow = Qb + (long)ql * p.ldq;
        const T* dOrow = dOb + (long)ql * p.ld_do;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};   // [q = g4+r][k = l15]
        Frag4<T> qf[DQT], gf[DVT];
#pragma unroll
        for (int dt = 0; dt < DQT; ++dt) qf[dt] = frag_ld<T>(Qrow + dt * 16 + g4);
#pragma unroll
        for (int ut = 0; ut < DVT; ++ut) gf[ut] = frag_ld<T>(dOrow + ut * 16 + g4);
#pragma unroll
        for (int dt = 0; dt < DQT; ++dt) s = mma16(qf[dt], kf[dt], s);
#pragma unroll
        for (int ut = 0; ut < DVT; ++ut) da = mma16(gf[ut], vf[ut], da);
        f32x4 a4, ds;
        float stm[4], stl[4], std_[4];      // softmax statistics of the tile's four queries: one batch of unconditional loads
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long si = j.bp * p.T + min(qt * 16 + g4 + r, p.T - 1);
            stm[r] = p.st_m[si]; stl[r] = p.st_l[si]; std_[r] = p.st_d[si];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + g4 + r, qc = min(q, p.T - 1);
            const long si = j.bp * p.T + qc;
            const bool fut = causal && k > q;
            float v = fmaf(s[r], p.cscale, madd);
            if (fut && kok) v = PADV;
            const float P = (q < p.T) ? __expf(v - stm[r]) / stl[r] : 0.f;
            bool keep = true;
            if (dk.thresh != 0u) {
                // same pairing as the forward: one hash per (k even, k + 1) of a row
                const uint32_t h = drop_hash_pair(dk, (uint32_t)(si * p.T) + (uint32_t)(k & ~1));
                keep = ((k & 1) ? (h >> 16) : (h & 0xffffu)) >= dk.t16;
            }
            a4[r] = keep ? P * dk.scale : 0.f;
            const float dP = keep ? da[r] * dk.scale : 0.f;
            ds[r] = (pad || fut || q >= p.T) ? 0.f : P * (dP - std_[r]) * p.cscale;
        }
        const Frag4<T> af = frag_from_acc<T>(a4), dsf = frag_from_acc<T>(ds);
#pragma unroll
        for (int ut = 0; ut < DVT; ++ut) accV[ut] = mma16(turn<T>(gf[ut], ident), af, accV[ut]);    // dV^T[u][k] += dO^T[u][q] A[q][k]
#pragma unroll
        for (int dt = 0; dt < DQT; ++dt) accK[dt] = mma16(turn<T>(qf[dt], ident), dsf, accK[dt]);   // dK~^T[d][k] += Q~^T[d][q] dS[q][k]
    }
    if (!kok) return;
    T* dK = reinterpret_cast<T*>(p.d_kx) + ((long)j.b * p.T + k) * p.ld_dk + j.head * p.Dq;
    T* dV = reinterpret_cast<T*>(p.d_v) + ((long)j.b * p.T + k) * p.ld_dv + j.head * p.Dv;
#pragma unroll
    for (int dt = 0; dt < DQT; ++dt) frag_st<T>(dK + dt * 16 + g4, frag_from_acc<T>(accK[dt]));
#pragma unroll
    for (int ut = 0; ut < DVT; ++ut) frag_st<T>(dV + ut * 16 + g4, frag_from_acc<T>(accV[ut]));
}

// =========================================================================================================================
// Interval-bucket form — TiMultiHeadAttention.__call__ (temporal.py:36-105, TiSASRec): the score adds Q[q].Ktime[dt(q,k)]
// and the value adds Vtime[dt(q,k)], dt = int(clip(t[q+1] - t[k], 0, timelen)) (TiSASREC.py:58-62).  The score term of a
// pair is a dh-long dot product against the gathered table row (packed bf16 dot instructions; the tables are L2-resident);
// on the value side the probabilities are binned per interval (W[q][d] += A[q,k], LDS atomics) and one MFMA chain applies
// Vtime to the bins — likewise dQ += dG . Ktime backward, and the table gradients are dG^T . Q and W^T . dO over all rows.
// The reference builds two [B,T,T,C] gathers instead.
// =========================================================================================================================
struct TiP {
    const float* ts; float time_scale; int timelen;
    const void *ktime, *vtime; int ldt, tab_rows;     // [tab_rows, H*dh] activation dtype; bucket >= tab_rows reads as zeros
    void *wbuf, *dgbuf; int NBp;                      // [H*B*T][NBp] activation dtype: binned probabilities / score gradients
};

__device__ __forceinline__ int bucket_of(float xq1, float xk, int timelen) {
    return (int)fminf(fmaxf(xq1 - xk, 0.f), (float)timelen);                       // clip then tf.to_int64 (truncation)
}
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// LDS tile of one wave: bins [16][NBp + 4] f32 (binned probabilities in the forward, binned score gradients backward)
__host__ __device__ constexpr size_t ti_tile_f(int NBp) { return (size_t)16 * (NBp + 4) * sizeof(float); }

// <a, b> over one head slice (16*DT channels), both rows read from global / L1: the interval term of a (query, key) pair,
// Q[q] . Ktime[bucket] or dO[q] . Vtime[bucket].  bf16: packed v_dot2c_f32_bf16 (2 MACs per instruction), f32 accumulate.
template <typename T, int DT>
__device__ __forceinline__ float dot_rows(const T* a, const T* b) {
    float acc = 0.f;
    if constexpr (sizeof(T) == 2) {
        typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
#pragma unroll
        for (int c = 0; c < 2 * DT; ++c) {
            const uint4 x = *reinterpret_cast<const uint4*>(a + c * 8), y = *reinterpret_cast<const uint4*>(b + c * 8);
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x.x), __builtin_bit_cast(bf2, y.x), acc, false);
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x.y), __builtin_bit_cast(bf2, y.y), acc, false);
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x.z), __builtin_bit_cast(bf2, y.z), acc, false);
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x.w), __builtin_bit_cast(bf2, y.w), acc, false);
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4 * DT; ++c) {
            const float4 x = *reinterpret_cast<const float4*>(a + c * 4), y = *reinterpret_cast<const float4*>(b + c * 4);
            acc = fmaf(x.x, y.x, acc); acc = fmaf(x.y, y.y, acc); acc = fmaf(x.z, y.z, acc); acc = fmaf(x.w, y.w, acc);
        }
    }
    return acc;
}
// row of an interval table for `bucket` (clamped address; the caller zeroes the result for bucket >= tab_rows)
template <typename T>
__device__ __forceinline__ const T* tab_row(const T* tab, int ldt, int tab_rows, int bucket) {
    return tab + (long)min(bucket, tab_rows - 1) * ldt;
}
template <typename T> __device__ __forceinline__ Frag4<T> frag_from_f4(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    return frag_from_acc<T>(f32x4{v.x, v.y, v.z, v.w});
}
// acc[dt] += sum_d tab[d][dt*16 + .]^T . Ws[row][d]   (the bucket contraction on the matrix cores)
template <typename T, int DT>
__device__ __forceinline__ void bucket_apply(const float* Ws, int LDG, const T* tab, int ldt, int tab_rows, int NBp,
                                             const Frag4<T>& ident, f32x4 (&acc)[DT], int lane) {
    const int g4 = (lane >> 4) * 4, l15 = lane & 15;
#pragma unroll 4
    for (int d0 = 0; d0 < NBp; d0 += 16) {
        const Frag4<T> bf = frag_from_f4<T>(Ws + l15 * LDG + d0 + g4);
        const int dl = d0 + l15;
        const T* trow = tab + (long)min(dl, tab_rows - 1) * ldt;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            Frag4<T> tf = frag_ld<T>(trow + dt * 16 + g4);
            if (dl >= tab_rows) tf = frag_zero<T>();
            acc[dt] = mma16(turn<T>(tf, ident), bf, acc[dt]);
        }
    }
}
template <typename T>
__device__ __forceinline__ void bucket_store(const float* Ws, int LDG, int NBp, T* dst_row, int lane) {
    const int l15 = lane & 15;
#pragma unroll 4
    for (int c = (lane >> 4) * 4; c < NBp; c += 16) frag_st<T>(dst_row + c, frag_from_f4<T>(Ws + l15 * LDG + c));
}
// Interval dot products of a key tile's four pairs for 16-wide head slices (bf16, DT = 1; at DT = 2 the batches cost a wave of occupancy): all table rows are fetched before the
// first use (clamped addresses; the caller masks out-of-table buckets) — per pair, behind `bucket < tab_rows ? .. : 0`, they
// were a branch and a wait each.  xa: this lane's row a (2*DT 16-byte vectors, loaded once); TWO: also <xb, tab2[bucket]>.
template <int DT, bool TWO>
__device__ __forceinline__ void interval_dots4(const uint4 (&xa)[2 * DT], const uint4 (&xb)[2 * DT], const bf16* tab1, const bf16* tab2,
                                               int ldt, int tab_rows, const int (&bk)[4], float (&g1)[4], float (&g2)[4]) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    uint4 y1[4][2 * DT], y2[TWO ? 4 : 1][2 * DT];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long off = (long)min(bk[r], tab_rows - 1) * ldt;
#pragma unroll
        for (int c = 0; c < 2 * DT; ++c) {
            y1[r][c] = *reinterpret_cast<const uint4*>(tab1 + off + c * 8);
            if constexpr (TWO) y2[r][c] = *reinterpret_cast<const uint4*>(tab2 + off + c * 8);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2 * DT; ++c) {
            asm volatile("" : "+v"(y1[r][c].x), "+v"(y1[r][c].y), "+v"(y1[r][c].z), "+v"(y1[r][c].w));
            if constexpr (TWO) asm volatile("" : "+v"(y2[r][c].x), "+v"(y2[r][c].y), "+v"(y2[r][c].z), "+v"(y2[r][c].w));
        }
    auto dot = [](const uint4& a, const uint4& b, float acc) {
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.x), __builtin_bit_cast(bf2, b.x), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.y), __builtin_bit_cast(bf2, b.y), acc, false);
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.z), __builtin_bit_cast(bf2, b.z), acc, false);
        return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, a.w), __builtin_bit_cast(bf2, b.w), acc, false);
    };
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int c = 0; c < 2 * DT; ++c) {
            a1 = dot(xa[c], y1[r][c], a1);
            if constexpr (TWO) a2 = dot(xb[c], y2[r][c], a2);
        }
        g1[r] = a1; g2[r] = a2;
    }
}

// ... the same with a different left row per pair (the key-side backward: rows of four queries against their buckets' rows)
template <int DT>
__device__ __forceinline__ void interval_dots4_rows(const bf16* a0, long lda, const int (&arow)[4], const bf16* tab, int ldt,
                                                    int tab_rows, const int (&bk)[4], float (&g)[4]) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
    uint4 x[4][2 * DT], y[4][2 * DT];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const long off = (long)min(bk[r], tab_rows - 1) * ldt;
#pragma unroll
        for (int c = 0; c < 2 * DT; ++c) {
            x[r][c] = *reinterpret_cast<const uint4*>(a0 + (long)arow[r] * lda + c * 8);
            y[r][c] = *reinterpret_cast<const uint4*>(tab + off + c * 8);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 2 * DT; ++c) {
            asm volatile("" : "+v"(x[r][c].x), "+v"(x[r][c].y), "+v"(x[r][c].z), "+v"(x[r][c].w));
            asm volatile("" : "+v"(y[r][c].x), "+v"(y[r][c].y), "+v"(y[r][c].z), "+v"(y[r][c].w));
        }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < 2 * DT; ++c) {
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x[r][c].x), __builtin_bit_cast(bf2, y[r][c].x), acc, false);
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x[r][c].y), __builtin_bit_cast(bf2, y[r][c].y), acc, false);
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x[r][c].z), __builtin_bit_cast(bf2, y[r][c].z), acc, false);
            acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, x[r][c].w), __builtin_bit_cast(bf2, y[r][c].w), acc, false);
        }
        g[r] = acc;
    }
}

template <typename T, int DT>
__global__ __launch_bounds__(256) void tiattn_fwd_kernel(TaP p, TiP t) {
    extern __shared__ __attribute__((aligned(16))) char lds_c[];
    Job j;
    if (!get_job(p, j)) return;
    constexpr int dh = 16 * DT;
    const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, l15 = lane & 15, LDG = t.NBp + 4;
    float* Ws = reinterpret_cast<float*>(lds_c + (size_t)(threadIdx.x >> 6) * ti_tile_f(t.NBp));
    const int q = j.tile * 16 + l15, qc = min(q, p.T - 1);
    const bool qok = q < p.T, causal = (p.flags & TA_CAUSAL) != 0;
    const T* Qr = reinterpret_cast<const T*>(p.qx) + ((long)j.b * p.T + qc) * p.ldq + j.head * dh;
    const T* Kb = reinterpret_cast<const T*>(p.kx) + (long)j.b * p.T * p.ldk + j.head * dh;
    const T* Vb = reinterpret_cast<const T*>(p.v) + (long)j.b * p.T * p.ldv + j.head * dh;
    const T* Kt = reinterpret_cast<const T*>(t.ktime) + j.head * dh;
    const T* Vt = reinterpret_cast<const T*>(t.vtime) + j.head * dh;
    const int64_t* idr = p.ids + (long)j.b * p.T;
    const float* tsr = t.ts + (long)j.b * (p.T + 1);
    const float xq1 = tsr[qc + 1] / t.time_scale;                                         // TiSASREC.py:49
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const Frag4<T> ident = identity_frag<T>(lane);
    const uint32_t dbase = (uint32_t)((j.bp * p.T + qc) * p.T);
    const int kt_end = (causal && j.tile * 16 >= first_unpadded(idr, p.T, lane)) ? j.tile + 1 : p.NT;
    for (int i = lane * 4; i < 16 * LDG; i += 256) *reinterpret_cast<float4*>(Ws + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    wave_lds_sync();
    Frag4<T> qf[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) qf[dt] = frag_ld<T>(Qr + dt * 16 + g4);
    uint4 qrow[(sizeof(T) == 2 && DT == 1) ? 2 * DT : 1];   // bf16: this lane's query row (one head slice), kept for the interval dot products
    if constexpr (sizeof(T) == 2 && DT == 1) {
#pragma unroll
        for (int c = 0; c < 2 * DT; ++c) qrow[c] = *reinterpret_cast<const uint4*>(Qr + c * 8);
    }
    // The interval term of a pair hangs on two dependent loads (timestamp -> bucket -> table row).  Written per pair — with the
    // table row behind `bucket < tab_rows ? .. : 0` — that is a branch and two waits per pair: 8 dependent round trips per key
    // tile.  Here the four timestamps / ids of a tile travel together, then the four table rows (clamped, unconditional).
    auto scores = [&](int kt, float (&x)[4], int (&bk)[4]) {
        const int kr = min(kt * 16 + l15, p.T - 1);
        if constexpr (!(sizeof(T) == 2 && DT == 1)) {       // wide head slices / f32: per pair (four rows do not fit in registers)
            f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) s = mma16(frag_ld<T>(Kb + (long)kr * p.ldk + dt * 16 + g4), qf[dt], s);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = kt * 16 + g4 + r, kcl = min(k, p.T - 1);
                bk[r] = bucket_of(xq1, tsr[kcl] / t.time_scale, t.timelen);
                const float madd = k >= p.T ? -INFINITY : (idr[kcl] == 0 ? PADV : 0.f);
                const float g = bk[r] < t.tab_rows ? dot_rows<T, DT>(Qr, tab_row<T>(Kt, t.ldt, t.tab_rows, bk[r])) : 0.f;   // temporal.py:58
                float v = fmaf(s[r] + g, p.cscale, madd);                                     // temporal.py:56-62
                if (causal && k > q && k < p.T) v = PADV;
                x[r] = v;
            }
            return;
        }
        float tsk[4];
        int64_t idk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kcl = min(kt * 16 + g4 + r, p.T - 1);
            tsk[r] = tsr[kcl]; idk[r] = idr[kcl];
        }
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) s = mma16(frag_ld<T>(Kb + (long)kr * p.ldk + dt * 16 + g4), qf[dt], s);
        const T* rowp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bk[r] = bucket_of(xq1, tsk[r] / t.time_scale, t.timelen);
            rowp[r] = tab_row<T>(Kt, t.ldt, t.tab_rows, bk[r]);
        }
        float g[4];
        if constexpr (sizeof(T) == 2 && DT == 1) {     // (wider head slices: the four rows no longer fit in registers)
            typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
            uint4 y[4][2 * DT];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2 * DT; ++c) y[r][c] = *reinterpret_cast<const uint4*>(rowp[r] + c * 8);
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int c = 0; c < 2 * DT; ++c) asm volatile("" : "+v"(y[r][c].x), "+v"(y[r][c].y), "+v"(y[r][c].z), "+v"(y[r][c].w));
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < 2 * DT; ++c) {
                    const uint4 xq = qrow[c];
                    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, xq.x), __builtin_bit_cast(bf2, y[r][c].x), acc, false);
                    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, xq.y), __builtin_bit_cast(bf2, y[r][c].y), acc, false);
                    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, xq.z), __builtin_bit_cast(bf2, y[r][c].z), acc, false);
                    acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2, xq.w), __builtin_bit_cast(bf2, y[r][c].w), acc, false);
                }
                g[r] = acc;
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) g[r] = bk[r] < t.tab_rows ? dot_rows<T, DT>(Qr, rowp[r]) : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kt * 16 + g4 + r;
            const float madd = k >= p.T ? -INFINITY : (idk[r] == 0 ? PADV : 0.f);
            const float gi = bk[r] < t.tab_rows ? g[r] : 0.f;                            // temporal.py:58
            float v = fmaf(s[r] + gi, p.cscale, madd);                                    // temporal.py:56-62
            if (causal && k > q && k < p.T) v = PADV;
            x[r] = v;
        }
    };
    // pass 1: scores of every key tile, kept in registers for pass 2 (LDS, not registers, bounds the occupancy here, and the
    // second pass would otherwise repeat the S MFMA and the dependent timestamp -> bucket -> table-row loads of each pair)
    constexpr int NTM = 16;                   // T <= 256 (host-checked)
    float xs[NTM][4];
    float m = -INFINITY, l = 0.f;
#pragma unroll
    for (int kt = 0; kt < NTM; ++kt) {
        if (kt < kt_end) {
            float x[4]; int bk[4];
            scores(kt, x, bk);
#pragma unroll
            for (int r = 0; r < 4; ++r) xs[kt][r] = x[r];
            const float tmax = group_max4(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
            const float m_new = fmaxf(m, tmax);
            const float ps = __expf(x[0] - m_new) + __expf(x[1] - m_new) + __expf(x[2] - m_new) + __expf(x[3] - m_new);
            l = l * __expf(m - m_new) + group_sum4(ps);
            m = m_new;
        }
    }
    const float invl = 1.0f / l;
    f32x4 acc[DT];
#pragma unroll
    for (int ut = 0; ut < DT; ++ut) acc[ut] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kt = 0; kt < NTM; ++kt) {
        if (kt < kt_end) {
            const int kr = min(kt * 16 + l15, p.T - 1);
            uint32_t h0 = 0xffffffffu, h1 = 0xffffffffu;
            if (dk.thresh != 0u) { h0 = drop_hash_pair(dk, dbase + kt * 16 + g4); h1 = drop_hash_pair(dk, dbase + kt * 16 + g4 + 2); }
            const uint32_t hb[4] = {h0 & 0xffffu, h0 >> 16, h1 & 0xffffu, h1 >> 16};
            f32x4 a4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int kcl = min(kt * 16 + g4 + r, p.T - 1);
                const int bk = bucket_of(xq1, tsr[kcl] / t.time_scale, t.timelen);
                const float P = __expf(xs[kt][r] - m) * invl;
                a4[r] = (dk.thresh == 0u || hb[r] >= dk.t16) ? P * dk.scale : 0.f;            // temporal.py:90
                atomicAdd(&Ws[l15 * LDG + bk], a4[r]);
            }
            const Frag4<T> pf = frag_from_acc<T>(a4);
#pragma unroll
            for (int ut = 0; ut < DT; ++ut)
                acc[ut] = mma16(turn<T>(frag_ld<T>(Vb + (long)kr * p.ldv + ut * 16 + g4), ident), pf, acc[ut]);   // :93-94
        }
    }
    wave_lds_sync();
    bucket_apply<T, DT>(Ws, LDG, Vt, t.ldt, t.tab_rows, t.NBp, ident, acc, lane);                               // :95
    if (t.wbuf && qok) bucket_store<T>(Ws, LDG, t.NBp, reinterpret_cast<T*>(t.wbuf) + (j.bp * p.T + q) * (long)t.NBp, lane);
    if (p.st_m && qok && g4 == 0) { p.st_m[j.bp * p.T + q] = m; p.st_l[j.bp * p.T + q] = l; }
    if (!qok) return;
    const long orow = (long)j.b * p.T + q;
#pragma unroll
    for (int ut = 0; ut < DT; ++ut) {
        const int col = j.head * dh + ut * 16 + g4;
        f32x4 o = acc[ut];
        if (p.oatt) *reinterpret_cast<float4*>(p.oatt + orow * ((long)p.H * dh) + col) = make_float4(o[0], o[1], o[2], o[3]);
        const Frag4<T> rf = frag_ld<T>(reinterpret_cast<const T*>(p.resid) + orow * p.ldr + col);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += to_f32(rf.v[r]);                                                    // :102-104
        frag_st<T>(reinterpret_cast<T*>(p.out) + orow * p.ldo + col, frag_from_acc<T>(o));
    }
}

template <typename T, int DT>
__global__ __launch_bounds__(256) void tiattn_bwd_q_kernel(TaP p, TiP t) {
    extern __shared__ __attribute__((aligned(16))) char lds_c[];
    Job j;
    if (!get_job(p, j)) return;
    constexpr int dh = 16 * DT;
    const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, l15 = lane & 15, LDG = t.NBp + 4;
    float* dGs = reinterpret_cast<float*>(lds_c + (size_t)(threadIdx.x >> 6) * ti_tile_f(t.NBp));
    const int q = j.tile * 16 + l15, qc = min(q, p.T - 1);
    const bool qok = q < p.T, causal = (p.flags & TA_CAUSAL) != 0;
    const T* Qr = reinterpret_cast<const T*>(p.qx) + ((long)j.b * p.T + qc) * p.ldq + j.head * dh;
    const T* Kb = reinterpret_cast<const T*>(p.kx) + (long)j.b * p.T * p.ldk + j.head * dh;
    const T* Vb = reinterpret_cast<const T*>(p.v) + (long)j.b * p.T * p.ldv + j.head * dh;
    const T* dOr = reinterpret_cast<const T*>(p.d_out) + ((long)j.b * p.T + qc) * p.ld_do + j.head * dh;
    const T* Kt = reinterpret_cast<const T*>(t.ktime) + j.head * dh;
    const T* Vt = reinterpret_cast<const T*>(t.vtime) + j.head * dh;
    const int64_t* idr = p.ids + (long)j.b * p.T;
    const float* tsr = t.ts + (long)j.b * (p.T + 1);
    const float xq1 = tsr[qc + 1] / t.time_scale;
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const Frag4<T> ident = identity_frag<T>(lane);
    const uint32_t dbase = (uint32_t)((j.bp * p.T + qc) * p.T);
    const int kt_end = (causal && j.tile * 16 >= first_unpadded(idr, p.T, lane)) ? j.tile + 1 : p.NT;
    for (int i = lane * 4; i < 16 * LDG; i += 256) *reinterpret_cast<float4*>(dGs + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    float dsum = 0.f;
    Frag4<T> qf[DT], gf[DT];
    {
        const float* oa = p.oatt + ((long)j.b * p.T + qc) * ((long)p.H * dh) + j.head * dh;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            qf[dt] = frag_ld<T>(Qr + dt * 16 + g4);
            gf[dt] = frag_ld<T>(dOr + dt * 16 + g4);
            const float4 o = *reinterpret_cast<const float4*>(oa + dt * 16 + g4);
            dsum += to_f32(gf[dt].v[0]) * o.x + to_f32(gf[dt].v[1]) * o.y + to_f32(gf[dt].v[2]) * o.z + to_f32(gf[dt].v[3]) * o.w;
        }
        dsum = group_sum4(dsum);
        if (qok && g4 == 0) p.st_d[j.bp * p.T + q] = dsum;
    }
    const float m = p.st_m[j.bp * p.T + qc], invl = 1.0f / p.st_l[j.bp * p.T + qc];
    constexpr bool NARROW = sizeof(T) == 2 && DT == 1;      // interval dots with batched table rows (interval_dots4)
    uint4 qrow[NARROW ? 2 * DT : 1], dorow[NARROW ? 2 * DT : 1];
    if constexpr (NARROW) {
#pragma unroll
        for (int c = 0; c < 2 * DT; ++c) { qrow[c] = *reinterpret_cast<const uint4*>(Qr + c * 8); dorow[c] = *reinterpret_cast<const uint4*>(dOr + c * 8); }
    }
    wave_lds_sync();
    f32x4 acc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int kt = 0; kt < kt_end; ++kt) {
        const int kr = min(kt * 16 + l15, p.T - 1);
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            s = mma16(frag_ld<T>(Kb + (long)kr * p.ldk + dt * 16 + g4), qf[dt], s);
            da = mma16(frag_ld<T>(Vb + (long)kr * p.ldv + dt * 16 + g4), gf[dt], da);
        }
        uint32_t h0 = 0xffffffffu, h1 = 0xffffffffu;
        if (dk.thresh != 0u) { h0 = drop_hash_pair(dk, dbase + kt * 16 + g4); h1 = drop_hash_pair(dk, dbase + kt * 16 + g4 + 2); }
        const uint32_t hb[4] = {h0 & 0xffffu, h0 >> 16, h1 & 0xffffu, h1 >> 16};
        f32x4 ds;
        float tsk[4];
        int64_t idk[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kcl = min(kt * 16 + g4 + r, p.T - 1);
            tsk[r] = tsr[kcl]; idk[r] = idr[kcl];
        }
        int bks[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bks[r] = bucket_of(xq1, tsk[r] / t.time_scale, t.timelen);
        float gq[4], gd[4];
        if constexpr (NARROW) {
            interval_dots4<DT, true>(qrow, dorow, reinterpret_cast<const bf16*>(Kt), reinterpret_cast<const bf16*>(Vt), t.ldt, t.tab_rows,
                                     bks, gq, gd);
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool inb = bks[r] < t.tab_rows;
                gq[r] = inb ? dot_rows<T, DT>(Qr, tab_row<T>(Kt, t.ldt, t.tab_rows, bks[r])) : 0.f;
                gd[r] = inb ? dot_rows<T, DT>(dOr, tab_row<T>(Vt, t.ldt, t.tab_rows, bks[r])) : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kt * 16 + g4 + r;
            const int bk = bks[r];
            const bool pad = k >= p.T || idk[r] == 0, fut = causal && k > q;
            const bool inb = bk < t.tab_rows;
            const float g = inb ? gq[r] : 0.f;
            const float dw = inb ? gd[r] : 0.f;                                                          // dA[q,k] gets dO[q].Vtime[dt(q,k)]
            float v = fmaf(s[r] + g, p.cscale, k >= p.T ? -INFINITY : (pad ? PADV : 0.f));
            if (fut && k < p.T) v = PADV;
            const float P = __expf(v - m) * invl;
            const float dP = (dk.thresh == 0u || hb[r] >= dk.t16) ? (da[r] + dw) * dk.scale : 0.f;
            ds[r] = (pad || fut || !qok) ? 0.f : P * (dP - dsum) * p.cscale;
            atomicAdd(&dGs[l15 * LDG + bk], ds[r]);
        }
        const Frag4<T> dsf = frag_from_acc<T>(ds);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            acc[dt] = mma16(turn<T>(frag_ld<T>(Kb + (long)kr * p.ldk + dt * 16 + g4), ident), dsf, acc[dt]);
    }
    wave_lds_sync();
    bucket_apply<T, DT>(dGs, LDG, Kt, t.ldt, t.tab_rows, t.NBp, ident, acc, lane);   // dQ[q] += sum_d dG[q][d] Ktime[d]
    if (qok) bucket_store<T>(dGs, LDG, t.NBp, reinterpret_cast<T*>(t.dgbuf) + (j.bp * p.T + q) * (long)t.NBp, lane);
    if (!qok) return;
    T* dst = reinterpret_cast<T*>(p.d_qx) + ((long)j.b * p.T + q) * p.ld_dq + j.head * dh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) frag_st<T>(dst + dt * 16 + g4, frag_from_acc<T>(acc[dt]));
}

template <typename T, int DT>
__global__ __launch_bounds__(256) void tiattn_bwd_k_kernel(TaP p, TiP t) {
    Job j;
    if (!get_job(p, j)) return;
    constexpr int dh = 16 * DT;
    const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int k = j.tile * 16 + l15, kc = min(k, p.T - 1);
    const bool kok = k < p.T, causal = (p.flags & TA_CAUSAL) != 0;
    const T* Qb = reinterpret_cast<const T*>(p.qx) + (long)j.b * p.T * p.ldq + j.head * dh;
    const T* Kr = reinterpret_cast<const T*>(p.kx) + ((long)j.b * p.T + kc) * p.ldk + j.head * dh;
    const T* Vr = reinterpret_cast<const T*>(p.v) + ((long)j.b * p.T + kc) * p.ldv + j.head * dh;
    const T* dOb = reinterpret_cast<const T*>(p.d_out) + (long)j.b * p.T * p.ld_do + j.head * dh;
    const T* Kt = reinterpret_cast<const T*>(t.ktime) + j.head * dh;
    const T* Vt = reinterpret_cast<const T*>(t.vtime) + j.head * dh;
    const float* tsr = t.ts + (long)j.b * (p.T + 1);
    const float xk = tsr[kc] / t.time_scale;
    const bool pad = !kok || p.ids[(long)j.b * p.T + kc] == 0;
    const float madd = !kok ? -INFINITY : (pad ? PADV : 0.f);
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const Frag4<T> ident = identity_frag<T>(lane);
    const int fnp = first_unpadded(p.ids + (long)j.b * p.T, p.T, lane);
    Frag4<T> kf[DT], vf[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { kf[dt] = frag_ld<T>(Kr + dt * 16 + g4); vf[dt] = frag_ld<T>(Vr + dt * 16 + g4); }
    f32x4 accK[DT], accV[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { accK[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; accV[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll 2
    for (int qt = 0; qt < p.NT; ++qt) {
        if (causal && qt < j.tile && qt * 16 >= fnp) continue;
        const int ql = min(qt * 16 + l15, p.T - 1);
        const T* Qrow = Qb + (long)ql * p.ldq;
        const T* dOrow = dOb + (long)ql * p.ld_do;
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, da = {0.f, 0.f, 0.f, 0.f};
        Frag4<T> qf[DT], gf[DT];
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            qf[dt] = frag_ld<T>(Qrow + dt * 16 + g4);
            gf[dt] = frag_ld<T>(dOrow + dt * 16 + g4);
            s = mma16(qf[dt], kf[dt], s);
            da = mma16(gf[dt], vf[dt], da);
        }
        // interval terms of the 4 (q, k) pairs of this lane
        float gq[4], dwq[4];
        int qcs[4], bks[4];
        float tsq[4], stm[4], stl[4], std_[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {      // per-query scalars of the tile: one batch (unconditional, clamped rows)
            qcs[r] = min(qt * 16 + g4 + r, p.T - 1);
            const long si = j.bp * p.T + qcs[r];
            tsq[r] = tsr[qcs[r] + 1]; stm[r] = p.st_m[si]; stl[r] = p.st_l[si]; std_[r] = p.st_d[si];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) bks[r] = bucket_of(tsq[r] / t.time_scale, xk, t.timelen);
        if constexpr (sizeof(T) == 2 && DT == 1) {
            interval_dots4_rows<DT>(reinterpret_cast<const bf16*>(Qb), p.ldq, qcs, reinterpret_cast<const bf16*>(Kt), t.ldt, t.tab_rows, bks, gq);
            interval_dots4_rows<DT>(reinterpret_cast<const bf16*>(dOb), p.ld_do, qcs, reinterpret_cast<const bf16*>(Vt), t.ldt, t.tab_rows, bks, dwq);
#pragma unroll
            for (int r = 0; r < 4; ++r) { const bool inb = bks[r] < t.tab_rows; gq[r] = inb ? gq[r] : 0.f; dwq[r] = inb ? dwq[r] : 0.f; }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const bool inb = bks[r] < t.tab_rows;
                gq[r] = inb ? dot_rows<T, DT>(Qb + (long)qcs[r] * p.ldq, tab_row<T>(Kt, t.ldt, t.tab_rows, bks[r])) : 0.f;
                dwq[r] = inb ? dot_rows<T, DT>(dOb + (long)qcs[r] * p.ld_do, tab_row<T>(Vt, t.ldt, t.tab_rows, bks[r])) : 0.f;
            }
        }
        f32x4 a4, ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + g4 + r;
            const long si = j.bp * p.T + qcs[r];
            const bool fut = causal && k > q;
            float v = fmaf(s[r] + gq[r], p.cscale, madd);
            if (fut && kok) v = PADV;
            const float P = (q < p.T) ? __expf(v - stm[r]) / stl[r] : 0.f;
            bool keep = true;
            if (dk.thresh != 0u) {
                const uint32_t h = drop_hash_pair(dk, (uint32_t)(si * p.T) + (uint32_t)(k & ~1));
                keep = ((k & 1) ? (h >> 16) : (h & 0xffffu)) >= dk.t16;
            }
            a4[r] = keep ? P * dk.scale : 0.f;
            const float dP = keep ? (da[r] + dwq[r]) * dk.scale : 0.f;
            ds[r] = (pad || fut || q >= p.T) ? 0.f : P * (dP - std_[r]) * p.cscale;
        }
        const Frag4<T> af = frag_from_acc<T>(a4), dsf = frag_from_acc<T>(ds);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            accV[dt] = mma16(turn<T>(gf[dt], ident), af, accV[dt]);
            accK[dt] = mma16(turn<T>(qf[dt], ident), dsf, accK[dt]);
        }
    }
    if (!kok) return;
    T* dK = reinterpret_cast<T*>(p.d_kx) + ((long)j.b * p.T + k) * p.ld_dk + j.head * dh;
    T* dV = reinterpret_cast<T*>(p.d_v) + ((long)j.b * p.T + k) * p.ld_dv + j.head * dh;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { frag_st<T>(dK + dt * 16 + g4, frag_from_acc<T>(accK[dt])); frag_st<T>(dV + dt * 16 + g4, frag_from_acc<T>(accV[dt])); }
}

// d_tab[d][head*dh + u] += sum_r Wb[head*BT + r][d] * X[r][head*dh + u]: the interval-table gradients
//   d(Ktime) = dG^T . Q,  d(Vtime) = W^T . dO   (contraction over all B*T rows; f32 atomics of per-split partial tiles)
template <typename T>
__global__ __launch_bounds__(256) void bucket_dtable_kernel(const T* Wb, int NBp, const T* X, int ldx, long BT, int H, int dh,
                                                            int tab_rows, float* d_tab, int ldt, int splits) {
    const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int ndt = NBp / 16, nut = dh / 16;
    long job = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (job >= (long)H * ndt * nut * splits) return;
    const int sp = (int)(job % splits); job /= splits;
    const int ut = (int)(job % nut); job /= nut;
    const int dti = (int)(job % ndt);
    const int head = (int)(job / ndt);
    const Frag4<T> ident = identity_frag<T>(lane);
    const long ntile = (BT + 15) / 16, per = (ntile + splits - 1) / splits;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (long rt = sp * per; rt < min(ntile, (sp + 1) * per); ++rt) {
        const long r = rt * 16 + l15, rc = min(r, BT - 1);
        Frag4<T> wf = frag_ld<T>(Wb + ((long)head * BT + rc) * NBp + dti * 16 + g4);
        if (r >= BT) wf = frag_zero<T>();
        const Frag4<T> xf = frag_ld<T>(X + rc * ldx + head * dh + ut * 16 + g4);
        acc = mma16(turn<T>(wf, ident), turn<T>(xf, ident), acc);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int d = dti * 16 + g4 + r;
        if (d < tab_rows) atomicAdd(d_tab + (long)d * ldt + head * dh + ut * 16 + l15, acc[r]);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void add_pos2_kernel(const T* kv, const float* posK, const float* posV, T* out, long rows, int T_, int C) {
    const int cpr = (2 * C) >> 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long row = gid / cpr;
    if (row >= rows) return;
    const int c0 = (int)(gid % cpr) * 4, tpos = (int)(row % T_);
    const Frag4<T> v = frag_ld<T>(kv + row * 2 * C + c0);
    const float4 pp = *reinterpret_cast<const float4*>((c0 < C ? posK + (long)tpos * C + c0 : posV + (long)tpos * C + c0 - C));
    Frag4<T> o;
    o.v[0] = from_f32<T>(to_f32(v.v[0]) + pp.x); o.v[1] = from_f32<T>(to_f32(v.v[1]) + pp.y);
    o.v[2] = from_f32<T>(to_f32(v.v[2]) + pp.z); o.v[3] = from_f32<T>(to_f32(v.v[3]) + pp.w);
    frag_st<T>(out + row * 2 * C + c0, o);
}

// =========================================================================================================================
// Wide heads (Dv > 128): TGAT's published recipe is ONE head of 512 channels (runme.sh:80-87: num_units 512, num_heads 1), so
// Dq = 1536, Dv = 512 — 128 operand fragments per row, far beyond a wave's registers.  The head is cut into SLICES of 128
// channels: a wave owns (sample, head, 16-row tile, slice); the score of a (q, k) tile needs the FULL contraction and is
// recomputed by every slice (operands re-read per 128-channel chunk from L1/L2 — at T = 30 a sequence has two key tiles),
// while the wide accumulators (O, dQ~, dK~, dV) are split over the slices.  Same arithmetic and the same saved statistics as
// the narrow kernels above.
// =========================================================================================================================
constexpr int SL = 8;   // 16-channel tiles per slice

struct SJob { int b, head, tile, slice; long bp; };
__device__ __forceinline__ bool get_sjob(const TaP& p, int nslice, SJob& j) {
    const long job = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (job >= (long)p.B * p.H * p.NT * nslice) return false;
    j.slice = (int)(job % nslice);
    const long r = job / nslice;
    j.tile = (int)(r % p.NT);
    const long bh = r / p.NT;
    j.head = (int)(bh % p.H);
    j.b = (int)(bh / p.H);
    j.bp = (long)j.head * p.B + j.b;
    return true;
}

// s[k][q] += sum over ALL Dq channels of K~[k] . Q~[q]  (row pointers of this lane's key row / query row)
template <typename T>
__device__ __forceinline__ f32x4 full_score(const T* Krow, const T* Qrow, int Dq, int g4, bool k_is_a) {
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < Dq; c += 16 * SL) {
        Frag4<T> kf[SL], qf[SL];
#pragma unroll
        for (int dt = 0; dt < SL; ++dt) { kf[dt] = frag_ld<T>(Krow + c + dt * 16 + g4); qf[dt] = frag_ld<T>(Qrow + c + dt * 16 + g4); }
#pragma unroll
        for (int dt = 0; dt < SL; ++dt) s = k_is_a ? mma16(kf[dt], qf[dt], s) : mma16(qf[dt], kf[dt], s);
    }
    return s;
}

template <typename T>
__global__ __launch_bounds__(256) void tattn_fwd_sliced_kernel(TaP p) {
    SJob j;
    const int nslice = p.Dv / (16 * SL);
    if (!get_sjob(p, nslice, j)) return;
    const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int q = j.tile * 16 + l15, qc = min(q, p.T - 1);
    const bool qok = q < p.T, causal = (p.flags & TA_CAUSAL) != 0;
    const int v0 = j.slice * 16 * SL;
    const T* Qr = reinterpret_cast<const T*>(p.qx) + ((long)j.b * p.T + qc) * p.ldq + j.head * p.Dq;
    const T* Kb = reinterpret_cast<const T*>(p.kx) + (long)j.b * p.T * p.ldk + j.head * p.Dq;
    const T* Vb = reinterpret_cast<const T*>(p.v) + (long)j.b * p.T * p.ldv + j.head * p.Dv + v0;
    const int64_t* idr = p.ids + (long)j.b * p.T;
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const Frag4<T> ident = identity_frag<T>(lane);
    const uint32_t dbase = (uint32_t)((j.bp * p.T + qc) * p.T);
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc[SL];
#pragma unroll
    for (int ut = 0; ut < SL; ++ut) acc[ut] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kt_end = (causal && j.tile * 16 >= first_unpadded(idr, p.T, lane)) ? j.tile + 1 : p.NT;
    for (int kt = 0; kt < kt_end; ++kt) {
        const int kr = min(kt * 16 + l15, p.T - 1);
        const f32x4 s = full_score<T>(Kb + (long)kr * p.ldk, Qr, p.Dq, g4, true);
        Frag4<T> vraw[SL];
#pragma unroll
        for (int ut = 0; ut < SL; ++ut) vraw[ut] = frag_ld<T>(Vb + (long)kr * p.ldv + ut * 16 + g4);
        float x[4], tmax = -INFINITY;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kt * 16 + g4 + r;
            const float madd = k >= p.T ? -INFINITY : (idr[min(k, p.T - 1)] == 0 ? PADV : 0.f);   // temporal.py:153-158
            float v = fmaf(s[r], p.cscale, madd);
            if (causal && k > q && k < p.T) v = PADV;                                            // :161-166
            x[r] = v;
            tmax = fmaxf(tmax, v);
        }
        tmax = group_max4(tmax);
        const float m_new = fmaxf(m_run, tmax), corr = __expf(m_run - m_new);
        f32x4 e;
        float psum = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) { e[r] = __expf(x[r] - m_new); psum += e[r]; }
        l_run = l_run * corr + group_sum4(psum);
        m_run = m_new;
        if (dk.thresh != 0u) {
            const uint32_t h0 = drop_hash_pair(dk, dbase + kt * 16 + g4), h1 = drop_hash_pair(dk, dbase + kt * 16 + g4 + 2);
            e[0] = (h0 & 0xffffu) >= dk.t16 ? e[0] : 0.f;
            e[1] = (h0 >> 16) >= dk.t16 ? e[1] : 0.f;
            e[2] = (h1 & 0xffffu) >= dk.t16 ? e[2] : 0.f;
            e[3] = (h1 >> 16) >= dk.t16 ? e[3] : 0.f;
        }
        const Frag4<T> pf = frag_from_acc<T>(e);
#pragma unroll
        for (int ut = 0; ut < SL; ++ut) {
            const Frag4<T> vt = turn<T>(vraw[ut], ident);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[ut][r] *= corr;
            acc[ut] = mma16(vt, pf, acc[ut]);
        }
    }
    const float inv = dk.scale / l_run;
    if (p.st_m && qok && g4 == 0 && j.slice == 0) { p.st_m[j.bp * p.T + q] = m_run; p.st_l[j.bp * p.T + q] = l_run; }
    if (!qok) return;
    const long orow = (long)j.b * p.T + q;
#pragma unroll
    for (int ut = 0; ut < SL; ++ut) {
        const int col = j.head * p.Dv + v0 + ut * 16 + g4;
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = acc[ut][r] * inv;
        if (p.oatt) *reinterpret_cast<float4*>(p.oatt + orow * ((long)p.H * p.Dv) + col) = make_float4(o[0], o[1], o[2], o[3]);
        const Frag4<T> rf = frag_ld<T>(reinterpret_cast<const T*>(p.resid) + orow * p.ldr + col);
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] += to_f32(rf.v[r]);
        frag_st<T>(reinterpret_cast<T*>(p.out) + orow * p.ldo + col, frag_from_acc<T>(o));
    }
}

// da[k][q] (or [q][k]) = sum over ALL Dv channels of V[k] . dO[q]
template <typename T>
__device__ __forceinline__ f32x4 full_da(const T* Vrow, const T* dOrow, int Dv, int g4, bool v_is_a) {
    f32x4 da = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
    for (int c = 0; c < Dv; c += 16 * SL) {
        Frag4<T> vf[SL], gf[SL];
#pragma unroll
        for (int ut = 0; ut < SL; ++ut) { vf[ut] = frag_ld<T>(Vrow + c + ut * 16 + g4); gf[ut] = frag_ld<T>(dOrow + c + ut * 16 + g4); }
#pragma unroll
        for (int ut = 0; ut < SL; ++ut) da = v_is_a ? mma16(vf[ut], gf[ut], da) : mma16(gf[ut], vf[ut], da);
    }
    return da;
}

template <typename T>
__global__ __launch_bounds__(256) void tattn_bwd_q_sliced_kernel(TaP p) {
    SJob j;
    const int nslice = p.Dq / (16 * SL);
    if (!get_sjob(p, nslice, j)) return;
    const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int q = j.tile * 16 + l15, qc = min(q, p.T - 1);
    const bool qok = q < p.T, causal = (p.flags & TA_CAUSAL) != 0;
    const int d0 = j.slice * 16 * SL;
    const T* Qr = reinterpret_cast<const T*>(p.qx) + ((long)j.b * p.T + qc) * p.ldq + j.head * p.Dq;
    const T* Kb = reinterpret_cast<const T*>(p.kx) + (long)j.b * p.T * p.ldk + j.head * p.Dq;
    const T* Vb = reinterpret_cast<const T*>(p.v) + (long)j.b * p.T * p.ldv + j.head * p.Dv;
    const T* dOr = reinterpret_cast<const T*>(p.d_out) + ((long)j.b * p.T + qc) * p.ld_do + j.head * p.Dv;
    const int64_t* idr = p.ids + (long)j.b * p.T;
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const Frag4<T> ident = identity_frag<T>(lane);
    const uint32_t dbase = (uint32_t)((j.bp * p.T + qc) * p.T);
    float dsum = 0.f;   // D[q] = dO[q] . O_att[q]
    {
        const float* oa = p.oatt + ((long)j.b * p.T + qc) * ((long)p.H * p.Dv) + j.head * p.Dv;
        for (int u = 0; u < p.Dv; u += 16) {
            const Frag4<T> g = frag_ld<T>(dOr + u + g4);
            const float4 o = *reinterpret_cast<const float4*>(oa + u + g4);
            dsum += to_f32(g.v[0]) * o.x + to_f32(g.v[1]) * o.y + to_f32(g.v[2]) * o.z + to_f32(g.v[3]) * o.w;
        }
        dsum = group_sum4(dsum);
        if (qok && g4 == 0 && j.slice == 0) p.st_d[j.bp * p.T + q] = dsum;
    }
    const float m = p.st_m[j.bp * p.T + qc], invl = 1.0f / p.st_l[j.bp * p.T + qc];
    f32x4 acc[SL];
#pragma unroll
    for (int dt = 0; dt < SL; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int kt_end = (causal && j.tile * 16 >= first_unpadded(idr, p.T, lane)) ? j.tile + 1 : p.NT;
    for (int kt = 0; kt < kt_end; ++kt) {
        const int kr = min(kt * 16 + l15, p.T - 1);
        const f32x4 s = full_score<T>(Kb + (long)kr * p.ldk, Qr, p.Dq, g4, true);
        const f32x4 da = full_da<T>(Vb + (long)kr * p.ldv, dOr, p.Dv, g4, true);
        uint32_t h0 = 0xffffffffu, h1 = 0xffffffffu;
        if (dk.thresh != 0u) { h0 = drop_hash_pair(dk, dbase + kt * 16 + g4); h1 = drop_hash_pair(dk, dbase + kt * 16 + g4 + 2); }
        const uint32_t hb[4] = {h0 & 0xffffu, h0 >> 16, h1 & 0xffffu, h1 >> 16};
        f32x4 ds;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kt * 16 + g4 + r;
            const bool pad = k >= p.T || idr[min(k, p.T - 1)] == 0, fut = causal && k > q;
            float v = fmaf(s[r], p.cscale, k >= p.T ? -INFINITY : (pad ? PADV : 0.f));
            if (fut && k < p.T) v = PADV;
            const float P = __expf(v - m) * invl;
            const float dP = (dk.thresh == 0u || hb[r] >= dk.t16) ? da[r] * dk.scale : 0.f;
            ds[r] = (pad || fut || !qok) ? 0.f : P * (dP - dsum) * p.cscale;
        }
        const Frag4<T> dsf = frag_from_acc<T>(ds);
#pragma unroll
        for (int dt = 0; dt < SL; ++dt)
            acc[dt] = mma16(turn<T>(frag_ld<T>(Kb + (long)kr * p.ldk + d0 + dt * 16 + g4), ident), dsf, acc[dt]);
    }
    if (!qok) return;
    T* dst = reinterpret_cast<T*>(p.d_qx) + ((long)j.b * p.T + q) * p.ld_dq + j.head * p.Dq + d0;
#pragma unroll
    for (int dt = 0; dt < SL; ++dt) frag_st<T>(dst + dt * 16 + g4, frag_from_acc<T>(acc[dt]));
}

template <typename T>
__global__ __launch_bounds__(256) void tattn_bwd_k_sliced_kernel(TaP p) {
    SJob j;
    const int nq = p.Dq / (16 * SL), nslice = nq + p.Dv / (16 * SL);   // slices [0, nq): dK~ channels; the rest: dV channels
    if (!get_sjob(p, nslice, j)) return;
    const int lane = threadIdx.x & 63, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int k = j.tile * 16 + l15, kc = min(k, p.T - 1);
    const bool kok = k < p.T, causal = (p.flags & TA_CAUSAL) != 0;
    const bool for_k = j.slice < nq;
    const int c0 = (for_k ? j.slice : j.slice - nq) * 16 * SL;
    const T* Qb = reinterpret_cast<const T*>(p.qx) + (long)j.b * p.T * p.ldq + j.head * p.Dq;
    const T* Kr = reinterpret_cast<const T*>(p.kx) + ((long)j.b * p.T + kc) * p.ldk + j.head * p.Dq;
    const T* Vr = reinterpret_cast<const T*>(p.v) + ((long)j.b * p.T + kc) * p.ldv + j.head * p.Dv;
    const T* dOb = reinterpret_cast<const T*>(p.d_out) + (long)j.b * p.T * p.ld_do + j.head * p.Dv;
    const bool pad = !kok || p.ids[(long)j.b * p.T + kc] == 0;
    const float madd = !kok ? -INFINITY : (pad ? PADV : 0.f);
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const Frag4<T> ident = identity_frag<T>(lane);
    f32x4 acc[SL];
#pragma unroll
    for (int t = 0; t < SL; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fnp = first_unpadded(p.ids + (long)j.b * p.T, p.T, lane);
    for (int qt = 0; qt < p.NT; ++qt) {
        if (causal && qt < j.tile && qt * 16 >= fnp) continue;
        const int ql = min(qt * 16 + l15, p.T - 1);
        const T* Qrow = Qb + (long)ql * p.ldq;
        const T* dOrow = dOb + (long)ql * p.ld_do;
        const f32x4 s = full_score<T>(Kr, Qrow, p.Dq, g4, false);     // [q = g4+r][k = l15]
        const f32x4 da = full_da<T>(Vr, dOrow, p.Dv, g4, false);
        f32x4 a4, ds;
        float stm[4], stl[4], std_[4];      // softmax statistics of the tile's four queries: one batch of unconditional loads
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long si = j.bp * p.T + min(qt * 16 + g4 + r, p.T - 1);
            stm[r] = p.st_m[si]; stl[r] = p.st_l[si]; std_[r] = p.st_d[si];
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = qt * 16 + g4 + r, qc = min(q, p.T - 1);
            const long si = j.bp * p.T + qc;
            const bool fut = causal && k > q;
            float v = fmaf(s[r], p.cscale, madd);
            if (fut && kok) v = PADV;
            const float P = (q < p.T) ? __expf(v - stm[r]) / stl[r] : 0.f;
            bool keep = true;
            if (dk.thresh != 0u) {
                const uint32_t h = drop_hash_pair(dk, (uint32_t)(si * p.T) + (uint32_t)(k & ~1));
                keep = ((k & 1) ? (h >> 16) : (h & 0xffffu)) >= dk.t16;
            }
            a4[r] = keep ? P * dk.scale : 0.f;
            const float dP = keep ? da[r] * dk.scale : 0.f;
            ds[r] = (pad || fut || q >= p.T) ? 0.f : P * (dP - std_[r]) * p.cscale;
        }
        const Frag4<T> rhs = frag_from_acc<T>(for_k ? ds : a4);
        const T* src = for_k ? Qrow + c0 : dOrow + c0;     // dK~^T += Q~^T . dS ; dV^T += dO^T . A
#pragma unroll
        for (int t = 0; t < SL; ++t) acc[t] = mma16(turn<T>(frag_ld<T>(src + t * 16 + g4), ident), rhs, acc[t]);
    }
    if (!kok) return;
    T* dst = for_k ? reinterpret_cast<T*>(p.d_kx) + ((long)j.b * p.T + k) * p.ld_dk + j.head * p.Dq + c0
                   : reinterpret_cast<T*>(p.d_v) + ((long)j.b * p.T + k) * p.ld_dv + j.head * p.Dv + c0;
#pragma unroll
    for (int t = 0; t < SL; ++t) frag_st<T>(dst + t * 16 + g4, frag_from_acc<T>(acc[t]));
}

template <typename K>
int launch_sliced(K kern, const TaP& p, int nslice, hipStream_t st) {
    const long jobs = (long)p.B * p.H * p.NT * nslice;
    hipLaunchKernelGGL(kern, dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
inline bool wide_head(const TaP& p) { return p.Dv > 128 && p.Dv % (16 * SL) == 0 && p.Dq % (16 * SL) == 0; }

template <typename K>
int launch_jobs(K kern, const TaP& p, hipStream_t st) {
    const long jobs = (long)p.B * p.H * p.NT;
    hipLaunchKernelGGL(kern, dim3((unsigned)((jobs + 3) / 4)), dim3(256), 0, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

template <typename T>
int launch_fwd(const TaP& p, hipStream_t st) {
    const int dqt = p.Dq / 16, dvt = p.Dv / 16;
    if (wide_head(p)) return launch_sliced(tattn_fwd_sliced_kernel<T>, p, p.Dv / (16 * SL), st);
#define EDGL_TA_CASE(DQ, DV) if (dqt == DQ && dvt == DV) return launch_jobs(tattn_fwd_kernel<T, DQ, DV>, p, st);
    if constexpr (sizeof(T) == 2) {
        if (dqt == 24 && dvt == 8) return launch_jobs(tattn_fwd_kernel_w4<T, 24, 8>, p, st);
    }
    EDGL_TA_CASE(1, 1) EDGL_TA_CASE(2, 2) EDGL_TA_CASE(4, 4) EDGL_TA_CASE(8, 8)
    EDGL_TA_CASE(3, 1) EDGL_TA_CASE(6, 2) EDGL_TA_CASE(12, 4) EDGL_TA_CASE(24, 8)
#undef EDGL_TA_CASE
    edgl_set_error("edgl_tattn_fwd: head dims Dq=%d Dv=%d not supported (Dv in {16,32,64,128} with Dq in {Dv, 3 Dv}, or multiples of 128 above)", p.Dq, p.Dv);
    return EDGL_ERR_SHAPE;
}
template <typename T>
int launch_bwd(const TaP& p, hipStream_t st) {
    const int dqt = p.Dq / 16, dvt = p.Dv / 16;
    int rc = EDGL_ERR_SHAPE;
    if (wide_head(p)) {
        rc = launch_sliced(tattn_bwd_q_sliced_kernel<T>, p, p.Dq / (16 * SL), st);
        if (rc == EDGL_OK) rc = launch_sliced(tattn_bwd_k_sliced_kernel<T>, p, (p.Dq + p.Dv) / (16 * SL), st);
        return rc;
    }
#define EDGL_TA_CASE(DQ, DV)                                                     \
    if (dqt == DQ && dvt == DV) {                                                \
        rc = launch_jobs(tattn_bwd_q_kernel<T, DQ, DV>, p, st);                      \
        if (rc == EDGL_OK) rc = launch_jobs(tattn_bwd_k_kernel<T, DQ, DV>, p, st); \
        return rc;                                                               \
    }
    EDGL_TA_CASE(1, 1) EDGL_TA_CASE(2, 2) EDGL_TA_CASE(4, 4) EDGL_TA_CASE(8, 8)        // plain heads (Dq == Dv)
    EDGL_TA_CASE(3, 1) EDGL_TA_CASE(6, 2) EDGL_TA_CASE(12, 4) EDGL_TA_CASE(24, 8)      // time feature map (Dq == 3 Dv)
#undef EDGL_TA_CASE
    edgl_set_error("edgl_tattn_bwd: head dims Dq=%d Dv=%d not supported (Dv in {16,32,64,128} with Dq in {Dv, 3 Dv}, or multiples of 128 above)", p.Dq, p.Dv);
    return rc;
}

// ---- time feature map (coding.py:104-122 through the angle-difference identity) ------------------------------------------
struct TfP {
    const void *q, *k; int ldq, ldk;
    const float *pos, *ts, *omega, *phi;
    const int64_t* ids;
    int B, T, C, H;
    float time_scale;
    void *qx, *kx;
    int* viol;
    const void *d_qx, *d_kx;
    void *d_q, *d_k; int ld_dq, ld_dk;
    float* part;   // [blocks][2C]
};

template <typename T>
__global__ __launch_bounds__(256) void timefn_fwd_kernel(TfP p) {
    const int cpr = p.C >> 2, dh = p.C / p.H;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long row = gid / cpr;
    if (row >= (long)p.B * p.T) return;
    const int c0 = (int)(gid % cpr) * 4, t = (int)(row % p.T);
    const long b = row / p.T;
    const float* tr = p.ts + b * (p.T + 1);
    const float base = tr[p.T] / p.time_scale;                       // TGAT.py:46 (seqs_ts = seqs_t / time_scale)
    const float xt = tr[t] / p.time_scale, xn = tr[t + 1] / p.time_scale;
    const float a = xn - base, bk = xt - base;
    if (c0 == 0 && xn < xt && p.ids[row] != 0) atomicAdd(p.viol, 1);  // max(.,0) of TGAT.py:54 would be active here
    const Frag4<T> qv = frag_ld<T>(reinterpret_cast<const T*>(p.q) + row * p.ldq + c0);
    const Frag4<T> kv = frag_ld<T>(reinterpret_cast<const T*>(p.k) + row * p.ldk + c0);
    const float4 w = *reinterpret_cast<const float4*>(p.omega + c0), ph = *reinterpret_cast<const float4*>(p.phi + c0);
    const float4 pp = *reinterpret_cast<const float4*>(p.pos + (long)t * p.C + c0);
    const float ww[4] = {w.x, w.y, w.z, w.w}, pv[4] = {ph.x, ph.y, ph.z, ph.w}, ps[4] = {pp.x, pp.y, pp.z, pp.w};
    Frag4<T> q0, qc, qs, k0, kc, ks;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float sq, cq, sk, ck;
        sincosf(a * ww[i] + pv[i], &sq, &cq);
        sincosf(bk * ww[i], &sk, &ck);
        const float qf = to_f32(qv.v[i]);
        q0.v[i] = qv.v[i];
        qc.v[i] = from_f32<T>(qf * cq);
        qs.v[i] = from_f32<T>(qf * sq);
        k0.v[i] = from_f32<T>(to_f32(kv.v[i]) + ps[i]);              // temporal.py:143,148: Q . (K + pos)^T
        kc.v[i] = from_f32<T>(ck);
        ks.v[i] = from_f32<T>(sk);
    }
    const int head = c0 / dh, u = c0 % dh;
    T* qo = reinterpret_cast<T*>(p.qx) + (row * p.H + head) * 3 * dh + u;
    T* ko = reinterpret_cast<T*>(p.kx) + (row * p.H + head) * 3 * dh + u;
    frag_st<T>(qo, q0); frag_st<T>(qo + dh, qc); frag_st<T>(qo + 2 * dh, qs);
    frag_st<T>(ko, k0); frag_st<T>(ko + dh, kc); frag_st<T>(ko + 2 * dh, ks);
}

// dQ = dq0 + dqc*cos + dqs*sin; dK = dk0 (also the position-table gradient, summed over the batch by the caller);
// dtheta = Q*(dqs*cos - dqc*sin) -> dphi += dtheta, domega += dtheta * a;  dpsi = dks*cos_k - dkc*sin_k -> domega += dpsi * b
template <typename T>
__global__ __launch_bounds__(256) void timefn_bwd_kernel(TfP p) {
    __shared__ float red[256][8];
    const int cpr = p.C >> 2, dh = p.C / p.H, rpb = 256 / cpr;     // rows per pass of a block
    const int cv = threadIdx.x % cpr, rl = threadIdx.x / cpr, c0 = cv * 4;
    const float4 w = *reinterpret_cast<const float4*>(p.omega + c0), ph = *reinterpret_cast<const float4*>(p.phi + c0);
    const float ww[4] = {w.x, w.y, w.z, w.w}, pv[4] = {ph.x, ph.y, ph.z, ph.w};
    float dw[4] = {0.f, 0.f, 0.f, 0.f}, dph[4] = {0.f, 0.f, 0.f, 0.f};
    const long R = (long)p.B * p.T;
    const int head = c0 / dh, u = c0 % dh;
    for (long row = (long)blockIdx.x * rpb + rl; row < R; row += (long)gridDim.x * rpb) {
        const int t = (int)(row % p.T);
        const long b = row / p.T;
        const float* tr = p.ts + b * (p.T + 1);
        const float base = tr[p.T] / p.time_scale;
        const float a = tr[t + 1] / p.time_scale - base, bk = tr[t] / p.time_scale - base;
        const Frag4<T> qv = frag_ld<T>(reinterpret_cast<const T*>(p.q) + row * p.ldq + c0);
        const T* gq = reinterpret_cast<const T*>(p.d_qx) + (row * p.H + head) * 3 * dh + u;
        const T* gk = reinterpret_cast<const T*>(p.d_kx) + (row * p.H + head) * 3 * dh + u;
        const Frag4<T> g0 = frag_ld<T>(gq), gc = frag_ld<T>(gq + dh), gs = frag_ld<T>(gq + 2 * dh);
        const Frag4<T> h0 = frag_ld<T>(gk), hc = frag_ld<T>(gk + dh), hs = frag_ld<T>(gk + 2 * dh);
        Frag4<T> dq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float sq, cq, sk, ck;
            sincosf(a * ww[i] + pv[i], &sq, &cq);
            sincosf(bk * ww[i], &sk, &ck);
            const float gcf = to_f32(gc.v[i]), gsf = to_f32(gs.v[i]);
            dq.v[i] = from_f32<T>(to_f32(g0.v[i]) + gcf * cq + gsf * sq);
            const float dth = to_f32(qv.v[i]) * (gsf * cq - gcf * sq);
            const float dps = to_f32(hs.v[i]) * ck - to_f32(hc.v[i]) * sk;
            dph[i] += dth;
            dw[i] += dth * a + dps * bk;
        }
        frag_st<T>(reinterpret_cast<T*>(p.d_q) + row * p.ld_dq + c0, dq);
        frag_st<T>(reinterpret_cast<T*>(p.d_k) + row * p.ld_dk + c0, h0);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) { red[threadIdx.x][i] = dw[i]; red[threadIdx.x][4 + i] = dph[i]; }
    __syncthreads();
    if (rl == 0) {
        float o[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int r = 0; r < rpb; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) o[i] += red[r * cpr + cv][i];
        float* dst = p.part + (long)blockIdx.x * 2 * p.C;
#pragma unroll
        for (int i = 0; i < 4; ++i) { dst[c0 + i] = o[i]; dst[p.C + c0 + i] = o[4 + i]; }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void mask_rows_kernel(const T* x, const int64_t* ids, T* y, long rows, int C) {
    const int cpr = C >> 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long row = gid / cpr;
    if (row >= rows) return;
    const int c0 = (int)(gid % cpr) * 4;
    const Frag4<T> v = ids[row] != 0 ? frag_ld<T>(x + row * C + c0) : frag_zero<T>();
    frag_st<T>(y + row * C + c0, v);
}

constexpr int TF_BLOCKS = 256;

struct SavedTa { size_t off_m, off_l, off_d, off_o, bytes; };
SavedTa saved_ta(int B, int T, int H, int Dv) {
    const size_t R = ((size_t)B * H * T + 63) & ~(size_t)63;
    SavedTa s;
    s.off_m = 0; s.off_l = R * 4; s.off_d = 2 * R * 4; s.off_o = 3 * R * 4;
    s.bytes = s.off_o + (size_t)B * T * H * Dv * 4;
    return s;
}

int check_common(const char* who, int B, int T, int H, int Dq, int Dv, int ldq, int ldk, int ldv, int dtype) {
    EDGL_REQUIRE(B > 0 && T > 0 && H > 0 && Dq > 0 && Dv > 0 && Dq % 16 == 0 && Dv % 16 == 0, EDGL_ERR_SHAPE,
                 "%s: bad shape B=%d T=%d H=%d Dq=%d Dv=%d (head dims must be multiples of 16)", who, B, T, H, Dq, Dv);
    EDGL_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0, EDGL_ERR_SHAPE, "%s: row strides must be multiples of 4", who);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "%s: bad dtype %d", who, dtype);
    EDGL_REQUIRE((double)B * H * T * T < 4294967296.0, EDGL_ERR_SHAPE, "%s: H*B*T*T must be < 2^32", who);
    return EDGL_OK;
}

}  // namespace

extern "C" long edgl_tattn_saved_bytes(int B, int T, int H, int Dv) {
    if (B <= 0 || T <= 0 || H <= 0 || Dv <= 0) return -1;
    return (long)saved_ta(B, T, H, Dv).bytes;
}

extern "C" int edgl_tattn_fwd(const void* qx, int ldq, const void* kx, int ldk, const void* v, int ldv, const void* resid,
                              int ldr, const int64_t* ids, int B, int T, int H, int Dq, int Dv, float scale, float drop_rate,
                              const uint64_t* rng_state, uint32_t stream_id, void* out, int ldo, void* saved, int flags,
                              int dtype, void* stream) {
    EDGL_REQUIRE(qx && kx && v && resid && ids && out, EDGL_ERR_NULL, "edgl_tattn_fwd: null pointer");
    if (int rc = check_common("edgl_tattn_fwd", B, T, H, Dq, Dv, ldq, ldk, ldv, dtype)) return rc;
    EDGL_REQUIRE(ldr % 4 == 0 && ldo % 4 == 0, EDGL_ERR_SHAPE, "edgl_tattn_fwd: row strides must be multiples of 4");
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_tattn_fwd: dropout without rng_state");
    TaP p{};
    p.qx = qx; p.kx = kx; p.v = v; p.resid = resid; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldr = ldr;
    p.ids = ids; p.B = B; p.T = T; p.H = H; p.Dq = Dq; p.Dv = Dv; p.cscale = scale; p.rate = drop_rate;
    p.rng = rng_state; p.stream_id = stream_id; p.out = out; p.ldo = ldo; p.flags = flags; p.NT = (T + 15) / 16;
    if (saved) {
        const SavedTa s = saved_ta(B, T, H, Dv);
        p.st_m = reinterpret_cast<float*>((char*)saved + s.off_m);
        p.st_l = reinterpret_cast<float*>((char*)saved + s.off_l);
        p.st_d = reinterpret_cast<float*>((char*)saved + s.off_d);
        p.oatt = reinterpret_cast<float*>((char*)saved + s.off_o);
    }
    return dtype == EDGL_F32 ? launch_fwd<float>(p, (hipStream_t)stream) : launch_fwd<bf16>(p, (hipStream_t)stream);
}

extern "C" int edgl_tattn_bwd(const void* qx, int ldq, const void* kx, int ldk, const void* v, int ldv, const int64_t* ids,
                              const void* d_out, int ld_do, void* saved, int B, int T, int H, int Dq, int Dv, float scale,
                              float drop_rate, const uint64_t* rng_state, uint32_t stream_id, void* d_qx, int ld_dq,
                              void* d_kx, int ld_dk, void* d_v, int ld_dv, int flags, int dtype, void* stream) {
    EDGL_REQUIRE(qx && kx && v && ids && d_out && saved && d_qx && d_kx && d_v, EDGL_ERR_NULL, "edgl_tattn_bwd: null pointer");
    if (int rc = check_common("edgl_tattn_bwd", B, T, H, Dq, Dv, ldq, ldk, ldv, dtype)) return rc;
    EDGL_REQUIRE(ld_do % 4 == 0 && ld_dq % 4 == 0 && ld_dk % 4 == 0 && ld_dv % 4 == 0, EDGL_ERR_SHAPE,
                 "edgl_tattn_bwd: row strides must be multiples of 4");
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_tattn_bwd: dropout without rng_state");
    TaP p{};
    p.qx = qx; p.kx = kx; p.v = v; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv;
    p.ids = ids; p.B = B; p.T = T; p.H = H; p.Dq = Dq; p.Dv = Dv; p.cscale = scale; p.rate = drop_rate;
    p.rng = rng_state; p.stream_id = stream_id; p.flags = flags; p.NT = (T + 15) / 16;
    p.d_out = d_out; p.ld_do = ld_do; p.d_qx = d_qx; p.d_kx = d_kx; p.d_v = d_v; p.ld_dq = ld_dq; p.ld_dk = ld_dk; p.ld_dv = ld_dv;
    const SavedTa s = saved_ta(B, T, H, Dv);
    p.st_m = reinterpret_cast<float*>((char*)saved + s.off_m);
    p.st_l = reinterpret_cast<float*>((char*)saved + s.off_l);
    p.st_d = reinterpret_cast<float*>((char*)saved + s.off_d);
    p.oatt = reinterpret_cast<float*>((char*)saved + s.off_o);
    return dtype == EDGL_F32 ? launch_bwd<float>(p, (hipStream_t)stream) : launch_bwd<bf16>(p, (hipStream_t)stream);
}

// ---- interval-bucket attention (TiSASRec) ---------------------------------------------------------------------------------
namespace {
template <typename K>
int launch_ti(K kern, const TaP& p, const TiP& t, int waves, size_t wave_bytes, hipStream_t st) {
    const size_t smem = (size_t)waves * wave_bytes;
    EDGL_REQUIRE(smem <= 160 * 1024, EDGL_ERR_SHAPE, "edgl_tiattn: timelen %d needs %zu B of LDS", t.timelen, smem);
    if (smem > 48 * 1024) hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const long jobs = (long)p.B * p.H * p.NT;
    hipLaunchKernelGGL(kern, dim3((unsigned)((jobs + waves - 1) / waves)), dim3(64 * waves), smem, st, p, t);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
template <typename T, int DT>
int ti_fwd(const TaP& p, const TiP& t, hipStream_t st) {
    return launch_ti(tiattn_fwd_kernel<T, DT>, p, t, 4, ti_tile_f(t.NBp), st);
}
template <typename T, int DT>
int ti_bwd(const TaP& p, const TiP& t, float* d_ktime, float* d_vtime, hipStream_t st) {
    if (int rc = launch_ti(tiattn_bwd_q_kernel<T, DT>, p, t, 4, ti_tile_f(t.NBp), st)) return rc;
    if (int rc = launch_ti(tiattn_bwd_k_kernel<T, DT>, p, t, 4, 0, st)) return rc;
    const int C = p.H * 16 * DT, splits = 32;
    if (hipMemsetAsync(d_ktime, 0, (size_t)t.tab_rows * C * sizeof(float), st) != hipSuccess ||
        hipMemsetAsync(d_vtime, 0, (size_t)t.tab_rows * C * sizeof(float), st) != hipSuccess) {
        edgl_set_error("edgl_tiattn_bwd: memset failed");
        return EDGL_ERR_LAUNCH;
    }
    const long jobs = (long)p.H * (t.NBp / 16) * DT * splits;
    const dim3 grid((unsigned)((jobs + 3) / 4));
    const long BT = (long)p.B * p.T;
    hipLaunchKernelGGL((bucket_dtable_kernel<T>), grid, dim3(256), 0, st, reinterpret_cast<const T*>(t.dgbuf), t.NBp,
                       reinterpret_cast<const T*>(p.qx), p.ldq, BT, p.H, 16 * DT, t.tab_rows, d_ktime, C, splits);
    hipLaunchKernelGGL((bucket_dtable_kernel<T>), grid, dim3(256), 0, st, reinterpret_cast<const T*>(t.wbuf), t.NBp,
                       reinterpret_cast<const T*>(p.d_out), p.ld_do, BT, p.H, 16 * DT, t.tab_rows, d_vtime, C, splits);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
int ti_check(const char* who, int B, int T, int H, int dh, int timelen, int tab_rows, int dtype) {
    EDGL_REQUIRE(B > 0 && T > 0 && H > 0 && (dh == 16 || dh == 32 || dh == 64 || dh == 128), EDGL_ERR_SHAPE,
                 "%s: bad shape B=%d T=%d H=%d head dim %d (16, 32, 64 or 128)", who, B, T, H, dh);
    EDGL_REQUIRE(timelen >= 1 && timelen <= 256 && tab_rows >= 1 && tab_rows <= timelen + 1, EDGL_ERR_SHAPE,
                 "%s: timelen %d (1..256) / table rows %d not supported", who, timelen, tab_rows);
    EDGL_REQUIRE(T <= 256, EDGL_ERR_SHAPE, "%s: T=%d not supported (T <= 256: 16 key tiles are kept in registers)", who, T);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "%s: bad dtype %d", who, dtype);
    EDGL_REQUIRE((double)B * H * T * T < 4294967296.0, EDGL_ERR_SHAPE, "%s: H*B*T*T must be < 2^32", who);
    return EDGL_OK;
}
}  // namespace

extern "C" long edgl_tiattn_bucket_elems(int B, int T, int H, int timelen) {
    if (B <= 0 || T <= 0 || H <= 0 || timelen < 1) return -1;
    return (long)B * T * H * ((timelen + 1 + 15) / 16 * 16);
}

extern "C" int edgl_tiattn_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* resid, int ldr,
                               const int64_t* ids, const float* ts, const void* ktime, const void* vtime, int tab_rows, int B,
                               int T, int H, int dh, float scale, float time_scale, int timelen, float drop_rate,
                               const uint64_t* rng_state, uint32_t stream_id, void* out, int ldo, void* saved, void* wbuf,
                               int flags, int dtype, void* stream) {
    EDGL_REQUIRE(q && k && v && resid && ids && ts && ktime && vtime && out, EDGL_ERR_NULL, "edgl_tiattn_fwd: null pointer");
    if (int rc = ti_check("edgl_tiattn_fwd", B, T, H, dh, timelen, tab_rows, dtype)) return rc;
    EDGL_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ldr % 4 == 0 && ldo % 4 == 0, EDGL_ERR_SHAPE,
                 "edgl_tiattn_fwd: row strides must be multiples of 4");
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_tiattn_fwd: dropout without rng_state");
    EDGL_REQUIRE((saved != nullptr) == (wbuf != nullptr), EDGL_ERR_NULL, "edgl_tiattn_fwd: saved and wbuf go together");
    TaP p{};
    p.qx = q; p.kx = k; p.v = v; p.resid = resid; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldr = ldr;
    p.ids = ids; p.B = B; p.T = T; p.H = H; p.Dq = dh; p.Dv = dh; p.cscale = scale; p.rate = drop_rate;
    p.rng = rng_state; p.stream_id = stream_id; p.out = out; p.ldo = ldo; p.flags = flags; p.NT = (T + 15) / 16;
    if (saved) {
        const SavedTa s = saved_ta(B, T, H, dh);
        p.st_m = reinterpret_cast<float*>((char*)saved + s.off_m);
        p.st_l = reinterpret_cast<float*>((char*)saved + s.off_l);
        p.st_d = reinterpret_cast<float*>((char*)saved + s.off_d);
        p.oatt = reinterpret_cast<float*>((char*)saved + s.off_o);
    }
    TiP t{ts, time_scale, timelen, ktime, vtime, H * dh, tab_rows, wbuf, nullptr, (timelen + 1 + 15) / 16 * 16};
    hipStream_t st = (hipStream_t)stream;
#define EDGL_TI_FWD(TT)                                        \
    switch (dh / 16) {                                         \
        case 1: return ti_fwd<TT, 1>(p, t, st);                \
        case 2: return ti_fwd<TT, 2>(p, t, st);                \
        case 4: return ti_fwd<TT, 4>(p, t, st);                \
        default: return ti_fwd<TT, 8>(p, t, st);               \
    }
    if (dtype == EDGL_F32) { EDGL_TI_FWD(float) }
    EDGL_TI_FWD(bf16)
#undef EDGL_TI_FWD
}

extern "C" int edgl_tiattn_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const int64_t* ids,
                               const float* ts, const void* ktime, const void* vtime, int tab_rows, const void* d_out, int ld_do,
                               void* saved, void* wbuf, int B, int T, int H, int dh, float scale, float time_scale, int timelen,
                               float drop_rate, const uint64_t* rng_state, uint32_t stream_id, void* d_q, int ld_dq, void* d_k,
                               int ld_dk, void* d_v, int ld_dv, void* dgbuf, float* d_ktime, float* d_vtime, int flags, int dtype,
                               void* stream) {
    EDGL_REQUIRE(q && k && v && ids && ts && ktime && vtime && d_out && saved && wbuf && d_q && d_k && d_v && dgbuf && d_ktime &&
                     d_vtime, EDGL_ERR_NULL, "edgl_tiattn_bwd: null pointer");
    if (int rc = ti_check("edgl_tiattn_bwd", B, T, H, dh, timelen, tab_rows, dtype)) return rc;
    EDGL_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && ld_do % 4 == 0 && ld_dq % 4 == 0 && ld_dk % 4 == 0 && ld_dv % 4 == 0,
                 EDGL_ERR_SHAPE, "edgl_tiattn_bwd: row strides must be multiples of 4");
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_tiattn_bwd: dropout without rng_state");
    TaP p{};
    p.qx = q; p.kx = k; p.v = v; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv;
    p.ids = ids; p.B = B; p.T = T; p.H = H; p.Dq = dh; p.Dv = dh; p.cscale = scale; p.rate = drop_rate;
    p.rng = rng_state; p.stream_id = stream_id; p.flags = flags; p.NT = (T + 15) / 16;
    p.d_out = d_out; p.ld_do = ld_do; p.d_qx = d_q; p.d_kx = d_k; p.d_v = d_v; p.ld_dq = ld_dq; p.ld_dk = ld_dk; p.ld_dv = ld_dv;
    const SavedTa s = saved_ta(B, T, H, dh);
    p.st_m = reinterpret_cast<float*>((char*)saved + s.off_m);
    p.st_l = reinterpret_cast<float*>((char*)saved + s.off_l);
    p.st_d = reinterpret_cast<float*>((char*)saved + s.off_d);
    p.oatt = reinterpret_cast<float*>((char*)saved + s.off_o);
    TiP t{ts, time_scale, timelen, ktime, vtime, H * dh, tab_rows, wbuf, dgbuf, (timelen + 1 + 15) / 16 * 16};
    hipStream_t st = (hipStream_t)stream;
#define EDGL_TI_BWD(TT)                                                        \
    switch (dh / 16) {                                                         \
        case 1: return ti_bwd<TT, 1>(p, t, d_ktime, d_vtime, st);              \
        case 2: return ti_bwd<TT, 2>(p, t, d_ktime, d_vtime, st);              \
        case 4: return ti_bwd<TT, 4>(p, t, d_ktime, d_vtime, st);              \
        default: return ti_bwd<TT, 8>(p, t, d_ktime, d_vtime, st);             \
    }
    if (dtype == EDGL_F32) { EDGL_TI_BWD(float) }
    EDGL_TI_BWD(bf16)
#undef EDGL_TI_BWD
}

extern "C" int edgl_add_pos2(const void* kv, const float* posK, const float* posV, int B, int T, int C, void* out, int dtype,
                             void* stream) {
    EDGL_REQUIRE(kv && posK && posV && out, EDGL_ERR_NULL, "edgl_add_pos2: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0 && C > 0 && C % 4 == 0, EDGL_ERR_SHAPE, "edgl_add_pos2: bad shape B=%d T=%d C=%d", B, T, C);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_add_pos2: bad dtype %d", dtype);
    const long rows = (long)B * T, total = rows * (2 * C / 4);
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == EDGL_F32) hipLaunchKernelGGL((add_pos2_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, (const float*)kv, posK,
                                              posV, (float*)out, rows, T, C);
    else hipLaunchKernelGGL((add_pos2_kernel<bf16>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)kv, posK, posV, (bf16*)out,
                            rows, T, C);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_timefn_fwd(const void* q, int ldq, const void* k, int ldk, const float* pos_tab, const float* ts,
                               const int64_t* ids, const float* omega, const float* phi, int B, int T, int C, int H,
                               float time_scale, void* qx, void* kx, int* violations, int dtype, void* stream) {
    EDGL_REQUIRE(q && k && pos_tab && ts && ids && omega && phi && qx && kx && violations, EDGL_ERR_NULL, "edgl_timefn_fwd: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0 && H > 0 && C % H == 0 && (C / H) % 16 == 0 && ldq % 4 == 0 && ldk % 4 == 0, EDGL_ERR_SHAPE,
                 "edgl_timefn_fwd: bad shape B=%d T=%d C=%d H=%d (head dim must be a multiple of 16)", B, T, C, H);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_timefn_fwd: bad dtype %d", dtype);
    TfP p{};
    p.q = q; p.k = k; p.ldq = ldq; p.ldk = ldk; p.pos = pos_tab; p.ts = ts; p.ids = ids; p.omega = omega; p.phi = phi;
    p.B = B; p.T = T; p.C = C; p.H = H; p.time_scale = time_scale; p.qx = qx; p.kx = kx; p.viol = violations;
    const long total = (long)B * T * (C / 4);
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == EDGL_F32) hipLaunchKernelGGL((timefn_fwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((timefn_fwd_kernel<bf16>), grid, dim3(256), 0, (hipStream_t)stream, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" long edgl_timefn_bwd_workspace(int C) { return (long)TF_BLOCKS * 2 * C; }

extern "C" int edgl_timefn_bwd(const void* q, int ldq, const float* ts, const float* omega, const float* phi, const void* d_qx,
                               const void* d_kx, int B, int T, int C, int H, float time_scale, void* d_q, int ld_dq, void* d_k,
                               int ld_dk, float* d_omega, float* d_phi, float* workspace, int dtype, void* stream) {
    EDGL_REQUIRE(q && ts && omega && phi && d_qx && d_kx && d_q && d_k && d_omega && d_phi && workspace, EDGL_ERR_NULL,
                 "edgl_timefn_bwd: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0 && H > 0 && C % H == 0 && (C / H) % 16 == 0 && C / 4 <= 256 && 256 % (C / 4) == 0 &&
                     ldq % 4 == 0 && ld_dq % 4 == 0 && ld_dk % 4 == 0, EDGL_ERR_SHAPE,
                 "edgl_timefn_bwd: bad shape B=%d T=%d C=%d H=%d (C/4 must divide 256)", B, T, C, H);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_timefn_bwd: bad dtype %d", dtype);
    TfP p{};
    p.q = q; p.ldq = ldq; p.ts = ts; p.omega = omega; p.phi = phi; p.B = B; p.T = T; p.C = C; p.H = H;
    p.time_scale = time_scale; p.d_qx = d_qx; p.d_kx = d_kx; p.d_q = d_q; p.d_k = d_k; p.ld_dq = ld_dq; p.ld_dk = ld_dk;
    p.part = workspace;
    hipStream_t st = (hipStream_t)stream;
    const int rpb = 256 / (C / 4);
    const int blocks = (int)std::min<long>(TF_BLOCKS, ((long)B * T + rpb - 1) / rpb);
    if (dtype == EDGL_F32) hipLaunchKernelGGL((timefn_bwd_kernel<float>), dim3(blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((timefn_bwd_kernel<bf16>), dim3(blocks), dim3(256), 0, st, p);
    EDGL_LAUNCH_CHECK();
    // partial rows are [d_omega | d_phi]: one fixed-order reduction (adjacent outputs) or one each
    if (d_phi == d_omega + C) return edgl_reduce_rows(workspace, blocks, 2 * C, 2L * C, d_omega, 0, st);
    if (int rc = edgl_reduce_rows(workspace, blocks, C, 2L * C, d_omega, 0, st)) return rc;
    return edgl_reduce_rows(workspace + C, blocks, C, 2L * C, d_phi, 0, st);
}

extern "C" int edgl_mask_rows(const void* x, const int64_t* ids, void* y, long rows, int C, int dtype, void* stream) {
    EDGL_REQUIRE(x && ids && y, EDGL_ERR_NULL, "edgl_mask_rows: null pointer");
    EDGL_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, EDGL_ERR_SHAPE, "edgl_mask_rows: bad shape rows=%ld C=%d", rows, C);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_mask_rows: bad dtype %d", dtype);
    const long total = rows * (C / 4);
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == EDGL_F32) hipLaunchKernelGGL((mask_rows_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream,
                                              (const float*)x, ids, (float*)y, rows, C);
    else hipLaunchKernelGGL((mask_rows_kernel<bf16>), grid, dim3(256), 0, (hipStream_t)stream, (const bf16*)x, ids, (bf16*)y, rows, C);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

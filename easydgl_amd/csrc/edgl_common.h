// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of libeasydgl_hip.so.
// Written for MI355X only: 64-wide wavefronts, MFMA 16x16 tiles, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/easydgl_hip.h"

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __bf16 bf16;

#define EDGL_WAVE 64

// ----------------------------------------------------------------------------------------------
// host-side error helpers
// ----------------------------------------------------------------------------------------------
extern "C" void edgl_set_error(const char* fmt, ...);

#define EDGL_REQUIRE(cond, code, ...)            \
    do {                                         \
        if (!(cond)) {                           \
            edgl_set_error(__VA_ARGS__);         \
            return (code);                       \
        }                                        \
    } while (0)

#define EDGL_LAUNCH_CHECK()                                              \
    do {                                                                 \
        hipError_t e_ = hipGetLastError();                               \
        if (e_ != hipSuccess) {                                          \
            edgl_set_error("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return EDGL_ERR_LAUNCH;                                      \
        }                                                                \
    } while (0)

static inline int edgl_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// out[n] (+)= sum_{p<P} part[p*ld + n], fixed summation order (k_misc.hip).  Used for every
// "per-workgroup partials -> parameter gradient" reduction.
int edgl_reduce_rows(const float* part, int P, int N, long ld, float* out, int accumulate, hipStream_t st);
// one-shot HIP-event bracket around a single kernel launch (edgl_profile_next); ids in easydgl_hip.h
void edgl_prof_begin(int kernel_id, hipStream_t st);
void edgl_prof_end(int kernel_id, hipStream_t st);
// bf16 fast paths (k_gemm2.hip): return 1 if taken, 0 if the shape does not qualify, <0 on error
int edgl_gemm2_try_strip(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int b_kc,
                         const float* bias, void* aux, int flags, hipStream_t st);
long edgl_gemm2_tn_workspace(int R, int Kf, int N);
int edgl_gemm2_try_tn(const void* X, const void* Y, float* C, int R, int Kf, int N, int ldx, int ldy, int ldc, float* dbias,
                      int accumulate, float* workspace, hipStream_t st);

// ----------------------------------------------------------------------------------------------
// element types: activations / GEMM operands are float (exact-f32 MFMA path) or bf16
// ----------------------------------------------------------------------------------------------
template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int VEC = 4;   // elements per 16-byte vector
    static constexpr int KB = 16;   // contraction elements consumed by one mma16 K-block
};
template <> struct ElemTraits<bf16> {
    static constexpr int VEC = 8;
    static constexpr int KB = 32;
};

__device__ __forceinline__ float to_f32(float x) { return x; }
__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }

// 16-byte vector of T
template <typename T> struct Vec16;
template <> struct Vec16<float> { float v[4]; };
template <> struct Vec16<bf16> { bf16 v[8]; };

template <typename T> __device__ __forceinline__ Vec16<T> ld16(const T* p) {
    Vec16<T> r;
    *reinterpret_cast<uint4*>(&r) = *reinterpret_cast<const uint4*>(p);
    return r;
}
template <typename T> __device__ __forceinline__ void st16(T* p, const Vec16<T>& r) {
    *reinterpret_cast<uint4*>(p) = *reinterpret_cast<const uint4*>(&r);
}
template <typename T> __device__ __forceinline__ Vec16<T> zero16() {
    Vec16<T> r;
    *reinterpret_cast<uint4*>(&r) = make_uint4(0, 0, 0, 0);
    return r;
}

// ----------------------------------------------------------------------------------------------
// MFMA 16x16 primitives.  One "K-block" contracts KB elements:
//   float : 4 x v_mfma_f32_16x16x4_f32  (exact f32; lane holds A[i=l&15][k=(l>>4)*4+r], r=0..3,
//           MFMA #r pairs register r of A with register r of B, so any k permutation agrees)
//   bf16  : 1 x v_mfma_f32_16x16x32_bf16 (lane holds 8 consecutive k at (l>>4)*8)
// D layout (both): acc[r] = D[row=(l>>4)*4+r][col=l&15].
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ f32x4 mma_kblock(const Vec16<float>& a, const Vec16<float>& b, f32x4 acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[r], b.v[r], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f32x4 mma_kblock(const Vec16<bf16>& a, const Vec16<bf16>& b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                   *reinterpret_cast<const bf16x8*>(&b), acc, 0, 0, 0);
}

// -DEDGL_PHASE_TIMING builds a diagnostic variant of a kernel file: every wave accumulates s_memtime deltas per phase and
// lane 0 adds them to g_phase_cycles (16 slots per translation unit, read back with the file's edgl_debug_phase_cycles*).
// Not part of the product build.
#ifdef EDGL_PHASE_TIMING
#define PH_DECL unsigned long long ph_t0 = __builtin_readcyclecounter(), ph_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define PH_MARK(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_readcyclecounter(); ph_acc[i] += t_ - ph_t0; ph_t0 = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#define PH_FLUSH(base) do { if ((threadIdx.x & 63) == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_phase_cycles[(base) + i_], ph_acc[i_]); } while (0)
#else
#define PH_DECL
#define PH_MARK(i)
#define PH_FLUSH(base)
#endif

// "Use" of every 32-bit word of a register-resident POD (prefetched operands): pins the s_waitcnt for its loads to this point.
// A prefetch that is still pending when a loop is entered makes every wait inside the loop conservative (the entry state is merged
// into the loop header): the waits then also count the stores of the previous iteration, i.e. stall on their acknowledges.
template <typename S>
__device__ __forceinline__ void touch_regs(S& s) {
    static_assert(sizeof(S) % 4 == 0, "32-bit words");
    uint32_t* w = reinterpret_cast<uint32_t*>(&s);
#pragma unroll
    for (int i = 0; i < (int)(sizeof(S) / 4); ++i) asm volatile("" : "+v"(w[i]));
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits for every global
// load a thread has in flight — which turns a register prefetch issued before the barrier into a stall on full memory
// latency.  Use this one when only LDS contents are exchanged across the barrier.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Scheduling fence: memory operations stay on their side at the IR level (asm memory clobber) and the machine scheduler
// moves nothing across (sched_barrier).  Used to keep hand-placed operand prefetches where they were written.
#define EDGL_PIN() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)

// 4-element (K=16) fragments used by the attention kernels: lane holds k=(l>>4)*4+j, j=0..3.
template <typename T> struct Frag4;
template <> struct Frag4<float> { float v[4]; };
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
template <> struct Frag4<bf16> { bf16x4 v; };   // one 64-bit register pair; v[i] indexes the packed vector

__device__ __forceinline__ f32x4 mma16(const Frag4<float>& a, const Frag4<float>& b, f32x4 acc) {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.v[r], b.v[r], acc, 0, 0, 0);
    return acc;
}
__device__ __forceinline__ f32x4 mma16(const Frag4<bf16>& a, const Frag4<bf16>& b, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(__builtin_bit_cast(s16x4, a.v), __builtin_bit_cast(s16x4, b.v), acc, 0, 0, 0);
}
template <typename T> __device__ __forceinline__ Frag4<T> frag_from_acc(const f32x4& c) {
    Frag4<T> f;
    if constexpr (sizeof(T) == 2) {   // one v_cvt_pk_bf16_f32 per pair (RNE, same as the scalar cast)
        // two <2 x float> -> <2 x bfloat> truncations, each moved through a 32-bit integer: this is the form that
        // selects one v_cvt_pk_bf16_f32 per pair (a <4 x bfloat> value feeding an MFMA is scalarised: 4 cvt + 2 perm)
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
        const bf16x2_t lo = __builtin_convertvector((f32x2_t){c[0], c[1]}, bf16x2_t);
        const bf16x2_t hi = __builtin_convertvector((f32x2_t){c[2], c[3]}, bf16x2_t);
        uint2 u;
        u.x = __builtin_bit_cast(uint32_t, lo);
        u.y = __builtin_bit_cast(uint32_t, hi);
        f.v = __builtin_bit_cast(bf16x4, u);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) f.v[r] = from_f32<T>(c[r]);
    }
    return f;
}
template <typename T> __device__ __forceinline__ Frag4<T> frag_ld(const T* p) {  // 4 contiguous elements
    Frag4<T> f;
    if constexpr (sizeof(T) == 4) {
        *reinterpret_cast<uint4*>(&f) = *reinterpret_cast<const uint4*>(p);
    } else {
        *reinterpret_cast<uint2*>(&f) = *reinterpret_cast<const uint2*>(p);
    }
    return f;
}
template <typename T> __device__ __forceinline__ Frag4<T> frag_zero() {
    Frag4<T> f;
#pragma unroll
    for (int r = 0; r < 4; ++r) f.v[r] = from_f32<T>(0.f);
    return f;
}


// ---- helpers on the MFMA register layout L(first,second): reg r of lane l = X[(l>>4)*4+r][l&15] ----
// Row exchanges of a wave (rows = 16 lanes) without LDS traffic: gfx950's v_permlane16_swap / v_permlane32_swap
// (semantics measured by tools/probe_swap.hip).  With rows r0..r3 of a and b:
//   swap16(a, b): first = {a.r0, b.r0, a.r2, b.r2}, second = {a.r1, b.r1, a.r3, b.r3}
//   swap32(a, b): first = {a.r0, a.r1, b.r0, b.r1}, second = {a.r2, a.r3, b.r2, b.r3}
// Written as inline asm: hipcc 7.2 mis-selects the second result of __builtin_amdgcn_permlane*_swap when both results
// feed one expression (it emits first + first).  The leading s_nop covers the VALU-write -> permlane-read wait states
// the compiler would otherwise insert.
struct FPair { float first, second; };
__device__ __forceinline__ FPair swap16(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return FPair{a, b};
}
__device__ __forceinline__ FPair swap32(float a, float b) {
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return FPair{a, b};
}
__device__ __forceinline__ float group_sum4(float v) {  // sum over the 4 lane groups (same lane&15)
    const FPair p = swap16(v, v);        // {r0,r0,r2,r2} , {r1,r1,r3,r3}
    const float t = p.first + p.second;  // pair sums, in both rows of the pair
    const FPair q = swap32(t, t);
    return q.first + q.second;
}
__device__ __forceinline__ float group_max4(float v) {
    const FPair p = swap16(v, v);
    const float t = fmaxf(p.first, p.second);
    const FPair q = swap32(t, t);
    return fmaxf(q.first, q.second);
}
template <typename T>
__device__ __forceinline__ Frag4<T> identity_frag(int lane) {
    Frag4<T> f;
#pragma unroll
    for (int j = 0; j < 4; ++j) f.v[j] = from_f32<T>(((lane >> 4) * 4 + j) == (lane & 15) ? 1.f : 0.f);
    return f;
}
// L(a,b) -> L(b,a): one MFMA against the identity (A operand = the tile itself)
template <typename T>
__device__ __forceinline__ f32x4 transpose_tile(const f32x4& x, const Frag4<T>& ident) {
    return mma16(frag_from_acc<T>(x), ident, f32x4{0.f, 0.f, 0.f, 0.f});
}

// ----------------------------------------------------------------------------------------------
// wave / block reductions
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// all threads get the block-wide sum; `red` holds >= blockDim/64 floats
__device__ __forceinline__ float block_sum(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// ----------------------------------------------------------------------------------------------
// counter-based dropout RNG: keep(idx) is a pure function of (seed, step, stream, idx) so the
// backward kernels regenerate the forward mask instead of storing it.
// rng_state (device): [0] = seed, [1] = step counter (advanced by edgl_rng_advance once per step).
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t edgl_mix32(uint32_t h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
struct DropKey { uint32_t k0, k1, thresh, t16; float scale; };
__device__ __forceinline__ DropKey make_dropkey(const uint64_t* rng_state, uint32_t stream, float rate) {
    DropKey k;
    uint64_t seed = rng_state ? rng_state[0] : 0ull, step = rng_state ? rng_state[1] : 0ull;
    k.k0 = edgl_mix32((uint32_t)seed ^ (stream * 0x9E3779B9u) ^ 0xA511E9B3u);
    k.k1 = edgl_mix32((uint32_t)(seed >> 32) + (uint32_t)step * 0x85ebca6bu + (uint32_t)(step >> 32) + stream);
    // drop iff hash < thresh ; thresh = rate * 2^32
    double t = (double)rate * 4294967296.0;
    k.thresh = rate <= 0.f ? 0u : (t >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)t);
    k.scale = rate <= 0.f ? 1.f : 1.f / (1.f - rate);
    k.t16 = rate <= 0.f ? 0u : (uint32_t)(rate * 65536.0f + 0.5f);   // 16-bit threshold for the paired form
    return k;
}
// Element-wise dropout (hidden dropouts: encoder, block tail, LayerNorm kernels, edgl_dropout): ONE multiply / xor-shift hash per
// ALIGNED GROUP OF FOUR consecutive elements, widened to 64 bits by one 32 x 32 -> 64 multiply; element idx owns the 16-bit field
// (idx & 3) of the hash of idx >> 2 and is kept iff the field is >= t16 (|p_eff - p| <= 8e-6, 2.3e-4 p on the top field;
// tools/dropout_hash_stats.py).  (seed, step, op) enter through k0 / k1, index bits above 2^34 through the additive term.  Every
// kernel that draws or re-derives a mask goes through drop_quad64 — a thread that owns four aligned elements hashes once
// (drop_apply4 / drop_keep4), the per-element forms below give the same decisions.
__device__ __forceinline__ uint64_t drop_quad64(const DropKey& k, uint64_t idx4) {
    const uint64_t qi = idx4 >> 2;
    uint32_t h = ((uint32_t)qi ^ k.k0) * 0x9E3779B1u;
    h += (uint32_t)(qi >> 32) * 0x7feb352du + k.k1;
    h ^= h >> 15; h *= 0x85ebca6bu; h ^= h >> 13;
    return (uint64_t)h * 0xFFF1AFD7u;
}
// keep-decision of element r (compile-time) of a group
template <int R>
__device__ __forceinline__ bool drop_quad_keep(const DropKey& k, uint64_t w) {
    const uint32_t half = R < 2 ? (uint32_t)w : (uint32_t)(w >> 32);
    return ((R & 1) ? (half >> 16) : (half & 0xffffu)) >= k.t16;
}
// x[0..3] = elements idx4 .. idx4 + 3 (idx4 a multiple of 4): dropped ones := 0, kept ones scaled
__device__ __forceinline__ void drop_apply4(const DropKey& k, uint64_t idx4, float (&x)[4]) {
    if (k.thresh == 0u) return;
    const uint64_t w = drop_quad64(k, idx4);
    x[0] = drop_quad_keep<0>(k, w) ? x[0] * k.scale : 0.f;
    x[1] = drop_quad_keep<1>(k, w) ? x[1] * k.scale : 0.f;
    x[2] = drop_quad_keep<2>(k, w) ? x[2] * k.scale : 0.f;
    x[3] = drop_quad_keep<3>(k, w) ? x[3] * k.scale : 0.f;
}
// bit r set: element idx4 + r is kept
__device__ __forceinline__ uint32_t drop_keep4(const DropKey& k, uint64_t idx4) {
    if (k.thresh == 0u) return 0xfu;
    const uint64_t w = drop_quad64(k, idx4);
    return (drop_quad_keep<0>(k, w) ? 1u : 0u) | (drop_quad_keep<1>(k, w) ? 2u : 0u) | (drop_quad_keep<2>(k, w) ? 4u : 0u) |
           (drop_quad_keep<3>(k, w) ? 8u : 0u);
}
__device__ __forceinline__ bool drop_keep(const DropKey& k, uint64_t idx) {
    const uint64_t w = drop_quad64(k, idx & ~3ull);
    return (uint32_t)((w >> (16 * (idx & 3ull))) & 0xffffull) >= k.t16;
}
// Paired form for the attention matrix: ONE hash decides two neighbouring elements (idx_even, idx_even+1) with
// 16-bit thresholds (|p_eff - p| < 8e-6).  Forward and backward kernels must both use it.
__device__ __forceinline__ uint32_t drop_hash_pair(const DropKey& k, uint32_t idx_even) {
    uint32_t h = (idx_even ^ k.k0) * 0x9E3779B1u + k.k1;
    h ^= h >> 15; h *= 0x85ebca6bu; h ^= h >> 13;
    return h;
}
// Quad form (BiMAU kernels): ONE 32-bit hash of the index of the first of four neighbouring elements, widened to 64 bits by one
// 32 x 32 -> 64 multiply (v_mad_u64_u32): element r of the quad owns the 16-bit field [16r, 16r + 16) and is kept iff its field is
// >= t16.  The top field ranges over [0, 0xFFF1] only: |p_eff - p| <= 2.3e-4 p there, 8e-6 on the others (tools/ hash statistics in
// rates, field / lag / step correlations at the noise level of 2^18 samples).  Half the VALU work of two paired hashes.
__device__ __forceinline__ uint64_t drop_hash_quad(const DropKey& k, uint32_t idx0) {
#ifdef EDGL_EXP_NOHASH   // timing experiment (tools/build_variant.sh): the cost of the hash itself — every element kept
    return ~0ull;
#endif
    uint32_t h = (idx0 ^ k.k0) * 0x9E3779B1u + k.k1;
    h ^= h >> 15; h *= 0x85ebca6bu; h ^= h >> 13;
    return (uint64_t)h * 0xFFF1AFD7u;
}
__device__ __forceinline__ float drop_apply(const DropKey& k, uint64_t idx, float x) {
    return k.thresh == 0u ? x : (drop_keep(k, idx) ? x * k.scale : 0.f);
}

// ----------------------------------------------------------------------------------------------
// math
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) {  // EasyDGL.py:31 exact erf GELU
    return x * (0.5f * (1.0f + erff(x * 0.70710678118654752440f)));
}
__device__ __forceinline__ float dgelu_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
    const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// erf with |error| <= 1.5e-7 (Abramowitz & Stegun 7.1.26): one v_rcp, one v_exp and five FMAs instead of libm's ~70
// instructions.  Used by the bf16 kernels only (a bf16 GELU output carries 4e-3 of rounding); the f32 parity path keeps
// erff.  `e_out` = exp(-u*u), which the GELU derivative needs as well.
__device__ __forceinline__ float erf_as(float u, float& e_out) {
    const float a = fabsf(u);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, a, 1.0f));
    const float e = __expf(-a * a);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    e_out = e;
    return copysignf(1.0f - poly * e, u);
}
template <typename T> __device__ __forceinline__ float gelu_t(float x) {     // gelu_f for activation dtype T
    if constexpr (sizeof(T) == 4) return gelu_f(x);
    float e;
    return x * (0.5f * (1.0f + erf_as(x * 0.70710678118654752440f, e)));
}
template <typename T> __device__ __forceinline__ float dgelu_t(float x) {    // dgelu_f for activation dtype T
    if constexpr (sizeof(T) == 4) return dgelu_f(x);
    float e;
    const float cdf = 0.5f * (1.0f + erf_as(x * 0.70710678118654752440f, e));
    return cdf + x * (0.39894228040143267794f * e);
}
// v_rcp_f32 (1 ulp) instead of an IEEE division sequence: sigmoid sits in the inner loop of the intensity MLP
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.0f + __expf(-x)); }

// Shared pieces of the fused BiMAU kernels (K3): BiMAU.__call__ temporal.py:404-452 and
// MAU.intensity temporal.py:281-315.
//
// Execution model: ONE WAVE (64 lanes) owns one (sample b, head) pair and walks its query rows in
// tiles of 16.  Everything is expressed as chains of 16x16 MFMA tiles in the TRANSPOSED orientation
// X^T[k][q] so that the query index q sits on lane&15 and the contraction index (key k, channel u,
// mark e ...) sits on (lane>>4)*4+reg — which is simultaneously the MFMA C/D layout and the MFMA B
// (or A) operand layout.  Consequently the score tile, the softmax, P.T_, the intensity MLP,
// lambda.marks^T, the modulation and (G*P).V all stay in registers; nothing of size [T,T] or
// [T,T,E] ever touches HBM (the reference materialises ~6 [hB,T,T] and 3 [hB,T,T,E] tensors).
//   register layout "L(first,second)":  reg r of lane l holds X[first=(l>>4)*4+r][second=l&15]
//   as MFMA B operand: contracts over `first`, keeps `second` as the output column
//   as MFMA A operand: contracts over `first`, keeps `second` as the output row
//   transpose_tile(): one MFMA against the identity turns L(a,b) into L(b,a).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "edgl_common.h"

namespace bimau {

constexpr int EP = 16;  // marks padded to one MFMA K-block
constexpr float NLOG2E = -1.4426950408889634f;

// sigmoid of a pre-activation that was already multiplied by -log2(e)
__device__ __forceinline__ float sigmoid_pre(float xs) { return fast_rcp(1.0f + __builtin_amdgcn_exp2f(xs)); }

// ---- intensity-weight pack (built once per launch by pack_kernel, copied to LDS by every WG) ---
// T-typed: W1T [JE][LDW]  (W1T[j][u] = -log2e * W1[u][j], u < dh) -> A operand of Zpre'^T = W1T.H^T
//          W1X [JE][8]    (bf16 only) interval weight and bias of channel j as bf16 split terms -> A operand of the
//                         MFMA that adds  span * w1s[j] + b1[j]  to the pre-activation (see span_frag)
// f32:     w1s [JE] = -log2e * W1[dh][j] (interval weight), b1 [JE] = -log2e * b1, wv [JE] = w.flatten(),
//          sc[16] = exp(scaling), isc[16] = 1/sc.
// T-typed: W1R [dh][LDR]  (W1R[u][j] = W1[u][j], unscaled)       -> A operand of dH^T = W1.du^T   (backward only: last,
//                         so that the forward copies `fwd_bytes` and leaves it out)
// The -log2(e) factor turns sigmoid(x) into rcp(1 + exp2(x')) — one v_exp_f32 and one v_rcp_f32, no multiply.
constexpr int XW = 8;   // K slots of the interval / bias MFMA that carry data (lane groups 0 and 1)
struct PackDims {
    int dh, E, JE, LDW, LDR;
    size_t off_w1x, off_f32, off_w1r, fwd_bytes, bytes;  // byte offsets
};
template <typename T>
__host__ __device__ inline PackDims pack_dims(int dh, int E) {
    PackDims d;
    d.dh = dh; d.E = E; d.JE = dh * E; d.LDW = dh + 4; d.LDR = d.JE + 4;
    size_t w1t = (size_t)d.JE * d.LDW * sizeof(T);
    w1t = (w1t + 15) & ~(size_t)15;
    const size_t w1x = sizeof(T) == 2 ? (size_t)d.JE * XW * sizeof(T) : 0;
    size_t f32b = ((size_t)3 * d.JE + 2 * EP) * sizeof(float);
    f32b = (f32b + 15) & ~(size_t)15;
    size_t w1r = (size_t)dh * d.LDR * sizeof(T);
    w1r = (w1r + 15) & ~(size_t)15;
    d.off_w1x = w1t;
    d.off_f32 = w1t + w1x;
    d.off_w1r = d.off_f32 + f32b;
    d.fwd_bytes = d.off_w1r;
    d.bytes = d.off_w1r + w1r;
    return d;
}

// Workgroup copy of the weight pack into LDS, eight 16-byte loads per thread in flight before the first store (a plain
// copy loop pays one L2 round trip per 4 KB of a 256-thread workgroup).
__device__ __forceinline__ void copy_pack_to_lds(char* smem, const char* pack, size_t bytes) {
    const uint4* src = reinterpret_cast<const uint4*>(pack);
    uint4* dst = reinterpret_cast<uint4*>(smem);
    const int n = (int)(bytes / 16), nt = blockDim.x;
    for (int i0 = threadIdx.x; i0 < n; i0 += 8 * nt) {
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[min(i0 + j * nt, n - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j)   // "used" here: otherwise each load sinks into the guard of its store and waits there (8 round trips)
            asm volatile("" : "+v"(v[j].x), "+v"(v[j].y), "+v"(v[j].z), "+v"(v[j].w));
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (i0 + j * nt < n) dst[i0 + j * nt] = v[j];
    }
}

// bf16 split of an f32 value: x = t0 + t1 + t2 up to 2^-24 |x|
struct Split3 { bf16 t0, t1, t2; };
__device__ __forceinline__ Split3 split3(float x) {
    Split3 s;
    s.t0 = from_f32<bf16>(x);
    const float r1 = x - to_f32(s.t0);
    s.t1 = from_f32<bf16>(r1);
    s.t2 = from_f32<bf16>(r1 - to_f32(s.t1));
    return s;
}
// B operand of the interval / bias MFMA for this lane's query row (K slot = 4 * lane group + i):
//   W1X row j : [w0 w1 w0 w2 | w1 w0 b0 b1]        (w = w0 + w1 + w2 interval weight, b = b0 + b1 bias)
//   span_frag : [s0 s0 s1 s0 | s1 s2  1  1 | 0 ...]  (span = s0 + s1 + s2)
// sum of the eight products = span * w + b with every cross term above 2^-24 kept (bf16 x bf16 is exact in the f32
// accumulator).  Lane groups 2 and 3 hold zeros here, so whatever finite values the A operand has there do not matter.
__device__ __forceinline__ Frag4<bf16> span_frag(float span, int lane) {
    const Split3 s = split3(span);
    const bf16 one = from_f32<bf16>(1.f), zero = from_f32<bf16>(0.f);
    Frag4<bf16> f;
    const int g = lane >> 4;
    f.v[0] = g == 0 ? s.t0 : g == 1 ? s.t1 : zero;
    f.v[1] = g == 0 ? s.t0 : g == 1 ? s.t2 : zero;
    f.v[2] = g == 0 ? s.t1 : g == 1 ? one : zero;
    f.v[3] = g == 0 ? s.t0 : g == 1 ? one : zero;
    return f;
}

// Activations the forward keeps for the backward (one buffer): H rows [H*B*T, dh] in the activation dtype (input of the
// intensity MLP) followed by z [H*B*T, 16] f32 (MLP output before scaling/softplus).
struct SavedLayout { size_t off_hin, off_z, bytes; };
inline SavedLayout saved_layout(int B, int T, int C, int H, size_t esize) {
    const size_t R = (size_t)B * H * T, dh = (size_t)(C / H);
    SavedLayout s;
    s.off_hin = 0;
    s.off_z = (R * dh * esize + 255) & ~(size_t)255;
    s.bytes = s.off_z + R * EP * sizeof(float);
    return s;
}

template <typename T>
__global__ void pack_kernel(const float* W1, const float* b1, const float* w, const float* scaling, int dh, int E,
                            char* pack) {
    const PackDims d = pack_dims<T>(dh, E);
    T* w1t = reinterpret_cast<T*>(pack);
    T* w1r = reinterpret_cast<T*>(pack + d.off_w1r);
    float* f = reinterpret_cast<float*>(pack + d.off_f32);
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    if constexpr (sizeof(T) == 2) {
        T* w1x = reinterpret_cast<T*>(pack + d.off_w1x);
        for (int j = tid; j < d.JE; j += nth) {
            const Split3 w = split3(NLOG2E * W1[(long)dh * d.JE + j]), b = split3(NLOG2E * b1[j]);
            T* r = w1x + (long)j * XW;
            r[0] = w.t0; r[1] = w.t1; r[2] = w.t0; r[3] = w.t2; r[4] = w.t1; r[5] = w.t0; r[6] = b.t0; r[7] = b.t1;
        }
    }
    for (int i = tid; i < d.JE * d.LDW; i += nth) {
        const int j = i / d.LDW, u = i % d.LDW;
        w1t[i] = from_f32<T>(u < dh ? NLOG2E * W1[(long)u * d.JE + j] : 0.f);
    }
    for (int i = tid; i < dh * d.LDR; i += nth) {
        const int u = i / d.LDR, j = i % d.LDR;
        w1r[i] = from_f32<T>(j < d.JE ? W1[(long)u * d.JE + j] : 0.f);
    }
    for (int j = tid; j < d.JE; j += nth) {
        f[j] = NLOG2E * W1[(long)dh * d.JE + j];
        f[d.JE + j] = NLOG2E * b1[j];
        f[2 * d.JE + j] = w[j];
    }
    for (int e = tid; e < EP; e += nth) {
        const float s = e < E ? __expf(scaling[e]) : 1.f;
        f[3 * d.JE + e] = s;
        f[3 * d.JE + EP + e] = 1.f / s;
    }
}

// per-lane key mask, ADDITIVE: madd[kt][r] for key k = kt*16 + (lane>>4)*4 + r is
//   0           real, unpadded key
//   -2^32       padded key (ids == 0): c*s + (-2^32) == float32(-2^32+1) exactly for |c*s| < 256, which is what
//               tf.where(mask == 0, -2**32+1, s) leaves (temporal.py:425-426)
//   -inf        k >= T (tile padding: excluded from the softmax)
// `pad` keeps the padded-key bits for the backward (no gradient flows into a padded score).
// kt0: the first key tile that holds a real (unpadded, k < T) key; 0 when the sequence has none.  Sequences are left-padded
// (data/linkpred.py:142-157), so the key tiles in front of kt0 are ENTIRELY padding: their scores are -2^32+1 against a finite row
// maximum, exp() of that is exactly 0 in f32, and P, G.P, dS, dK, dV, dT_ are exactly 0 there — the kernels that take the
// `SK` form leave those tiles out (wave-uniform choice of a code path compiled for NK = NT - kt0 key tiles).  A sequence
// without any real key keeps kt0 = 0: its softmax is uniform over all T keys (temporal.py:425-429).
template <int NT>
struct KeyMask { const float* madd; uint64_t pad; int kt0; };   // pad: one bit per (key tile, register), NT <= 16   // madd: wave-private LDS array [16*NT]
// first key tile with a real key from the ballots of "key k < T and id != 0" (one 64-bit word per round of 64 keys)
template <int NR>
__device__ __forceinline__ int first_real_tile(const uint64_t (&real)[NR]) {
    int first = -1;
#pragma unroll
    for (int i = NR - 1; i >= 0; --i)
        if (real[i] != 0ull) first = 64 * i + (int)__builtin_ctzll(real[i]);
    return first < 0 ? 0 : first >> 4;
}
// the key tiles K0 .. NT-1 of a mask as a mask of NK = NT - K0 tiles
template <int NK, int NT>
__device__ __forceinline__ KeyMask<NK> keymask_tail(const KeyMask<NT>& km) {
    constexpr int K0 = NT - NK;
    return KeyMask<NK>{km.madd + K0 * 16, km.pad >> (K0 * 4), 0};
}
// host side: EDGL_BIMAU_SKIP=0 launches the kernels that walk every key tile (read per launch: tests flip it)
inline bool bimau_skip_enabled() {
    const char* e = getenv("EDGL_BIMAU_SKIP");
    return !(e && e[0] == '0');
}
// run f(std::integral_constant<int, NK>) for the wave-uniform nk in 1 .. NT
template <int N, typename F>
__device__ __forceinline__ void dispatch_nk(int nk, F&& f) {
    if constexpr (N <= 1) {
        f(std::integral_constant<int, 1>{});
    } else {
        if (nk >= N) f(std::integral_constant<int, N>{});
        else dispatch_nk<N - 1>(nk, f);
    }
}
// One id per lane and round (keys lane, lane + 64, ...): the additive mask goes to LDS, the padded-key bits of this lane's
// (key tile, register) slots are cut out of the wave ballots — no per-slot id loads.
template <int NT>
__device__ __forceinline__ KeyMask<NT> load_keymask(const int64_t* ids_row, int T, int lane, float* lds_madd) {
    KeyMask<NT> km;
    km.pad = 0ull;
    km.madd = lds_madd;
    constexpr int NR = (16 * NT + 63) / 64;
    uint64_t padded[NR], real[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int k = lane + 64 * i;
        const int64_t id = ids_row[min(k, T - 1)];
        const bool pd = k < T && id == 0;
        padded[i] = __ballot(pd);
        real[i] = __ballot(k < T && id != 0);
        if (k < 16 * NT) lds_madd[k] = k >= T ? -INFINITY : (pd ? -4294967296.0f : 0.f);
    }
    km.kt0 = first_real_tile<NR>(real);
    const int g4 = (lane >> 4) * 4;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const uint64_t nib = (padded[kt / 4] >> ((kt % 4) * 16 + g4)) & 0xfull;   // keys kt*16 + g4 .. + 3
        km.pad |= nib << (kt * 4);
    }
    return km;
}

// S^T tiles -> normalised P^T tiles (temporal.py:422-429).  s[kt][r] in: raw Q.K; out: softmax.
// `causal` (MAU.__call__ with causality=True, temporal.py:370-375): keys k > q get the same -2^32+1 score as padded keys
// (tf.where(tril == 0, paddings, outputs)); q = this lane's query index.
constexpr int MAU_CAUSAL = 1;    // future blinding (temporal.py:370-375)
constexpr int MAU_NO_DIAG = 2;   // MAU keeps the modulation on the diagonal; BiMAU sets it to 1 (temporal.py:438-439)
// More than 16 mark types run as groups of <= 16 marks, one launch each (module/temporal.py): G = sum_g G_g is linear in the
// groups, so the first group writes the diagonal 1 and the later ones 0 — their outputs then add up to the ungrouped result.
constexpr int MAU_DIAG_ZERO = 4;
// The scale carries log2(e) so that the exponential is one v_exp_f32 (2^x) on (v - max): fma, sub, exp per element — all
// in 2- / 4-wide vector form (v_pk_fma_f32 / v_pk_add_f32).  The additive mask keeps its magnitude: a padded score is
// "-2^32 + something below the f32 resolution there", a fully padded row is uniform exactly as in the reference.
typedef __attribute__((ext_vector_type(2))) float f32x2;
// NORM: s := softmax * post (the factor rides on the normalisation: the dropout scale of sweep 1, whose every use of P carries it);
// !NORM: s := exp(v - max) unnormalised, the return value is 1 / sum (the forward folds it into lambda and the H rows).
template <int NT, bool CAUSAL, bool VEC, bool NORM = true>
__device__ __forceinline__ float masked_softmax_impl(f32x4 (&s)[NT], const KeyMask<NT>& km, float cscale, int lane, int q, float post = 1.0f) {
    const float c2 = cscale * 1.4426950408889634f;
    float mx = -INFINITY;
    const float* mrow = km.madd + (lane >> 4) * 4;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const f32x4 m4 = *reinterpret_cast<const f32x4*>(mrow + kt * 16);
        f32x4 v;
        if constexpr (VEC) {
            v = s[kt] * f32x4{c2, c2, c2, c2} + m4;
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = fmaf(s[kt][r], c2, m4[r]);
        }
        if constexpr (CAUSAL) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (kt * 16 + (lane >> 4) * 4 + r > q && m4[r] != -INFINITY) v[r] = -4294967296.0f;
        }
        s[kt] = v;
        mx = fmaxf(fmaxf(mx, v[0]), fmaxf(v[1], fmaxf(v[2], v[3])));
    }
    mx = group_max4(mx);
    float sum;
    // (v - mx) first: exact 0 for the row maximum even at the -2^32 padding magnitude
    if constexpr (VEC) {
        const f32x4 mx4 = {mx, mx, mx, mx};
        f32x4 sum4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const f32x4 d = s[kt] - mx4;
            f32x4 e;
#pragma unroll
            for (int r = 0; r < 4; ++r) e[r] = __builtin_amdgcn_exp2f(d[r]);
            s[kt] = e;
            sum4 += e;
        }
        sum = (sum4[0] + sum4[1]) + (sum4[2] + sum4[3]);
    } else {
        sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx);
                s[kt][r] = e;
                sum += e;
            }
    }
    sum = group_sum4(sum);
    const float inv = fast_rcp(sum);
    if constexpr (!NORM) return inv;
    const float f = inv * post;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        if constexpr (VEC) {
            s[kt] *= f32x4{f, f, f, f};
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] *= f;
        }
    }
    return inv;
}
// MODE 2: causal / bidirectional code paths behind one wave-uniform branch (no per-element causal selects on the
//         bidirectional path), 2-wide packed f32 arithmetic — the bf16 forward.
// MODE 1: the same branch, scalar arithmetic — the bf16 backward sweeps (packed operands need aligned register pairs;
//         at 7 key tiles that takes the sweeps past 256 registers, i.e. from two waves per SIMD to one).
// MODE 0: one code path with a run-time causal flag — the f32 kernels: at their register pressure (T_ and V staged
//         transposed, up to 13 key tiles) the other forms spill, and hipcc 7.2 crashes on that in its MFMA rewrite pass.
template <int NT, int MODE = 0>
__device__ __forceinline__ void masked_softmax(f32x4 (&s)[NT], const KeyMask<NT>& km, float cscale, int lane, int q = 0,
                                               bool causal = false) {
    if constexpr (MODE != 0) {
        if (causal) masked_softmax_impl<NT, true, false>(s, km, cscale, lane, q);
        else masked_softmax_impl<NT, false, false>(s, km, cscale, lane, q);
    } else {
        const float c2 = cscale * 1.4426950408889634f;
        float mx = -INFINITY;
        const float* mrow = km.madd + (lane >> 4) * 4;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const float4 m4 = *reinterpret_cast<const float4*>(mrow + kt * 16);
            const float mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = fmaf(s[kt][r], c2, mm[r]);
                if (causal && kt * 16 + (lane >> 4) * 4 + r > q && mm[r] != -INFINITY) v = -4294967296.0f;
                s[kt][r] = v;
                mx = fmaxf(mx, v);
            }
        }
        mx = group_max4(mx);
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx);
                s[kt][r] = e;
                sum += e;
            }
        sum = group_sum4(sum);
        const float inv = fast_rcp(sum);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[kt][r] *= inv;
    }
}

// ---- attention dropout as stored keep bits ---------------------------------------------------------------------------------
// The three kernels of a training step (forward, sweep 1, sweep 2) take the SAME keep decisions; hashing them three times is a
// tenth of each kernel (drop_hash_quad: two 32-bit multiplies, a 64-bit multiply-add and an SDWA compare per element).
// edgl_bimau_dropbits evaluates the hash ONCE per step — exactly the decisions of drop_hash_quad / drop_quad_keep on the kernels'
// element index (b', q, k) — and stores them in the kernels' own register layout: word [(b' * NT + qt) * 64 + lane], bit
// kt * 4 + r  <->  query qt*16 + (lane & 15), key kt*16 + (lane >> 4)*4 + r (NT <= 8 key tiles: 32 bits).  A kernel then loads
// one word per lane and query tile with its other per-tile operands and applies an element's decision with two instructions
// (sign-extended bit field, AND).  Same masks as the hashed form by construction: kernels without the bits (other head dims /
// flags / more than 8 key tiles) and kernels with them can be mixed inside one step.
// (the bit-field extract as asm: from the builtin the compiler knows the mask is 0 / -1 and rewrites the pair as v_and (bit
// test) + v_cmp_ne + v_cndmask — three issue slots per element in kernels that are bound by exactly those)
template <int I>
__device__ __forceinline__ float keep_bit_c(uint32_t kb, float x) {
    int m;
    asm("v_bfe_i32 %0, %1, %2, 1" : "=v"(m) : "v"(kb), "n"(I));   // 0 or -1
    return __int_as_float(__float_as_int(x) & m);
}
__device__ __forceinline__ float keep_bit(uint32_t kb, int i, float x) {   // i: a constant after unrolling (0 .. 31)
    switch (i) {
#define EDGL_KB_CASE(n) case n: return keep_bit_c<n>(kb, x);
        EDGL_KB_CASE(0) EDGL_KB_CASE(1) EDGL_KB_CASE(2) EDGL_KB_CASE(3) EDGL_KB_CASE(4) EDGL_KB_CASE(5) EDGL_KB_CASE(6) EDGL_KB_CASE(7)
        EDGL_KB_CASE(8) EDGL_KB_CASE(9) EDGL_KB_CASE(10) EDGL_KB_CASE(11) EDGL_KB_CASE(12) EDGL_KB_CASE(13) EDGL_KB_CASE(14) EDGL_KB_CASE(15)
        EDGL_KB_CASE(16) EDGL_KB_CASE(17) EDGL_KB_CASE(18) EDGL_KB_CASE(19) EDGL_KB_CASE(20) EDGL_KB_CASE(21) EDGL_KB_CASE(22) EDGL_KB_CASE(23)
        EDGL_KB_CASE(24) EDGL_KB_CASE(25) EDGL_KB_CASE(26) EDGL_KB_CASE(27) EDGL_KB_CASE(28) EDGL_KB_CASE(29) EDGL_KB_CASE(30) EDGL_KB_CASE(31)
#undef EDGL_KB_CASE
    }
    return x;
}
template <int NT>
__global__ __launch_bounds__(256) void dropbits_kernel(const uint64_t* rng, uint32_t stream_id, float rate, int T, long njobs, uint32_t* bits) {
    static_assert(NT <= 8, "one 32-bit word per lane and query tile");
    const int lane = threadIdx.x & 63;
    const long job = (long)blockIdx.x * 4 + (threadIdx.x >> 6);   // (b', query tile)
    if (job >= njobs) return;
    const long bp = job / NT;
    const int qt = (int)(job % NT), q = qt * 16 + (lane & 15), g4 = (lane >> 4) * 4;
    const DropKey dk = make_dropkey(rng, stream_id, rate);
    const uint32_t dbase = (uint32_t)((bp * T + q) * T);   // as in the kernels (rows q >= T: never used)
    uint32_t w = 0u;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const uint64_t hw = drop_hash_quad(dk, dbase + kt * 16 + g4);
        w |= (drop_quad_keep<0>(dk, hw) ? 1u : 0u) << (kt * 4);
        w |= (drop_quad_keep<1>(dk, hw) ? 1u : 0u) << (kt * 4 + 1);
        w |= (drop_quad_keep<2>(dk, hw) ? 1u : 0u) << (kt * 4 + 2);
        w |= (drop_quad_keep<3>(dk, hw) ? 1u : 0u) << (kt * 4 + 3);
    }
    bits[job * 64 + lane] = w;
}

// ---- TPP regulariser inside the attention kernels (MAU.biased_likelihood, temporal.py:317-333; EasyDGL.py:157-175) ------------
// The regulariser reads lambda at the masked positions only, and lambda of a query tile is in registers in sweep 1 (qcur.lam:
// lane group G owns marks 4G .. 4G+3 of query row lane&15).  With the slot data
// of a position prepared per (b, t) by edgl_tpp_prep — the 16 next-mark bytes of the position's first effective slot (a slot
// whose label has at least one mark), the raw-second span (negative: the position carries no effective slot) — sweep 1 forms
// its wave's share of the two loss sums and computes d lambda of the term in place of loading it: no [H*B, T, E] f32 d lambda
// array (26 MB written and read at the headline shape) and no TPP launch on the step's main stream.
// A position drawn into several effective slots (padding slots whose label row is not empty, hand-made batches) keeps its
// first slot in the per-position arrays and the others in a per-sample overflow list that both kernels walk (normally empty).
struct TppDesc {
    const uint32_t* nmw;      // [B, T, 4]  next-mark bytes of the first effective slot (mark e = byte e & 3 of word e >> 2)
    const float* spr;         // [B, T]     raw span (EasyDGL.py:161-162), or -1: no effective slot at this position
    const int32_t* novf;      // [B]        overflow slots of the sample
    const int32_t* ovf_pos;   // [B, M]     their positions
    const uint32_t* ovf_nm;   // [B, M, 4]  their next-mark bytes
    const int32_t* cntp;      // [B]        marks of ALL labels of the sample: their sum is the regulariser's normaliser (temporal.py:330)
    int M;
};
struct TppLayout { size_t off_spr, off_novf, off_ovf_pos, off_ovf_nm, off_cntp, bytes; };
__host__ __device__ inline TppLayout tpp_layout(int B, int T, int M) {
    TppLayout l;
    l.off_spr = (size_t)B * T * 16;
    l.off_novf = l.off_spr + (size_t)B * T * 4;
    l.off_ovf_pos = (l.off_novf + (size_t)B * 4 + 15) & ~(size_t)15;
    l.off_ovf_nm = (l.off_ovf_pos + (size_t)B * M * 4 + 15) & ~(size_t)15;
    l.off_cntp = l.off_ovf_nm + (size_t)B * M * 16;      // per-sample mark counts (edgl_tpp_prep's normaliser)
    l.bytes = l.off_cntp + (((size_t)B * 4 + 15) & ~(size_t)15);
    return l;
}
__host__ __device__ inline TppDesc tpp_desc(const void* base, int B, int T, int M) {
    const TppLayout l = tpp_layout(B, T, M);
    const char* c = reinterpret_cast<const char*>(base);
    return TppDesc{reinterpret_cast<const uint32_t*>(c), reinterpret_cast<const float*>(c + l.off_spr),
                   reinterpret_cast<const int32_t*>(c + l.off_novf), reinterpret_cast<const int32_t*>(c + l.off_ovf_pos),
                   reinterpret_cast<const uint32_t*>(c + l.off_ovf_nm), reinterpret_cast<const int32_t*>(c + l.off_cntp), M};
}
__device__ __forceinline__ void tpp_bytes(uint32_t w, float (&f)[4]) {
    f[0] = (float)(w & 0xffu); f[1] = (float)((w >> 8) & 0xffu);      // v_cvt_f32_ubyte0 .. 3
    f[2] = (float)((w >> 16) & 0xffu); f[3] = (float)(w >> 24);
}
// sums of two per-lane partials over the 4 lane groups, both results in every lane (three row swaps instead of four)
__device__ __forceinline__ void group_sum4_pair(float& x, float& y) {
    const FPair p = swap16(x, y);          // {x0, y0, x2, y2}, {x1, y1, x3, y3}
    const float t = p.first + p.second;    // rows 0, 2: pair sums of x; rows 1, 3: of y
    const FPair q = swap32(t, t);          // {t0, t1, t0, t1}, {t2, t3, t2, t3}
    const float u = q.first + q.second;    // even rows: sum of x; odd rows: sum of y
    const FPair r = swap16(u, u);          // {u0, u0, u2, u2}, {u1, u1, u3, u3}
    x = r.first; y = r.second;
}
// One slot of the regulariser on this lane's row (`on`: the row carries the slot and lies inside the sequence; lam4 = this lane's
// four lambda values): a += log(event intensity), bs += entire intensity * span / 2 (temporal.py:322-328) — every lane of a row's
// four ends with the row's terms — and the gradient  gr[i] += k (nm[e] / ev - span / 2),  k = -coef / (count H)  (temporal.py:331-332)
__device__ __forceinline__ void tpp_slot(const float (&lam4)[4], uint32_t w, float sp, bool on, float k, float (&gr)[4], float& a, float& bs) {
    float f[4];
    tpp_bytes(w, f);
    float ev = lam4[0] * f[0], ent = lam4[0] + lam4[1];
    ev = fmaf(lam4[1], f[1], ev); ev = fmaf(lam4[2], f[2], ev); ev = fmaf(lam4[3], f[3], ev);
    ent += lam4[2] + lam4[3];
    group_sum4_pair(ev, ent);
    ev = on ? ev : 0.f;
    const float hs = sp * 0.5f, kg = on ? k : 0.f;
    a += __logf(ev == 0.f ? 1.f : ev);
    bs = fmaf(on ? ent : 0.f, hs, bs);
    const float iev = ev != 0.f ? fast_rcp(ev) : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) gr[i] = fmaf(kg, fmaf(f[i], iev, -hs), gr[i]);
}

// reduce-scatter of 16 per-lane partials over the 4 lane groups: on return lane group g holds the
// complete sums for e = 4g + i (i = 0..3) — exactly the MFMA B-operand layout lambda^T[e][q].
__device__ __forceinline__ void reduce_scatter16(const float (&z)[16], float (&out)[4], int lane) {
    (void)lane;
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {   // lanes < 32 end with index i, lanes >= 32 with index i + 8
        const FPair p = swap32(z[i], z[i + 8]);
        y[i] = p.first + p.second;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // even rows end with index i, odd rows with index i + 4 (of their half)
        const FPair p = swap16(y[i], y[i + 4]);
        out[i] = p.first + p.second;
    }
}
// inverse: every lane ends with all 16 values (value for e = 4g+i lives in lane group g)
__device__ __forceinline__ void all_gather16(const float (&in)[4], float (&out)[16], int lane) {
    (void)lane;
    float y[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const FPair p = swap16(in[i], in[i]);
        y[i] = p.first;        // the even row's value (e = 8*half + i)
        y[i + 4] = p.second;   // the odd row's value  (e = 8*half + 4 + i)
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const FPair p = swap32(y[i], y[i]);
        out[i] = p.first;      // lower half's value
        out[i + 8] = p.second; // upper half's value
    }
}

// Bank swizzle of the wave-private ROW-MAJOR bf16 images (K / T_ / V: rows of dh = 16 .. 128 elements; marks: 16).
// The 8-byte fragment reads of the products that contract over the channel (row 16 kt + (l & 15), columns 16 ub + 4 (l >> 4) ..) are
// served 32 lanes at a time over 64 banks.  Rows of 32 / 64 / 128 / 256 bytes put rows r and r + 8 / 4 / 2 / 1 on the same banks:
// every such read was a 2- / 4- / 8- / 16-way conflict (PMC round 5, head dim 16: SQ_LDS_BANK_CONFLICT 9-14 % of the backward
// sweeps' wave cycles).  The 16-byte chunk c of row r lives at chunk c ^ f(r), f = the bits of r & 15 that the row stride does not
// already turn into a bank offset: (r >> 3) & 1, (r >> 2) & 3, (r >> 1) & 7, r & 15 for 2 / 4 / 8 / 16 chunks per row.  Then the 16
// rows of a half wave's read cover 16 distinct 16-byte slots = all 64 banks.  Writers (RowStage / stage_rows / the mark rows), the
// row fragment reads and the transpose reads (kfrag<T, W>) all go through swz_col.
template <typename T>
__host__ __device__ constexpr bool img_swz() { return sizeof(T) == 2; }
template <int W>      // W: elements per row (16, 32, 64, 128)
__host__ __device__ __forceinline__ constexpr int swz_col(int row, int col) {
    constexpr int CH = W / 8, SH = CH == 2 ? 3 : (CH == 4 ? 2 : (CH == 8 ? 1 : 0));
    static_assert(CH == 2 || CH == 4 || CH == 8 || CH == 16, "row width");
    return col ^ (((row >> SH) & (CH - 1)) << 3);      // (the chunk index is bits 3.. of the column)
}

// A/B fragment with the CONTRACTION index along the rows of a tile: v[j] = X[k0 + 4G + j][z0 + (l&15)].
// W > 0: the image is bank-swizzled with W elements per row (swz_col; k0 a multiple of 16).
//   bf16: one ds_read_b64_tr_b16 on the ROW-MAJOR tile rm[k][z] (ld_rm elements per row)
//   f32 : no 32-bit transpose read exists -> plain read of a separately staged transposed image tr[z][k]
template <typename T, int W = 0>
__device__ __forceinline__ Frag4<T> kfrag(const T* rm, int ld_rm, const T* tr, int ld_tr, int k0, int z0, int lane) {
    if constexpr (sizeof(T) == 2) {
        const int G = lane >> 4, s = lane & 15;
        typedef __attribute__((ext_vector_type(4))) short s4;
        int col = z0 + 4 * (s & 3);
        if constexpr (W > 0) col = swz_col<W>(4 * G + (s >> 2), col);
        const T* p = rm + (k0 + 4 * G + (s >> 2)) * ld_rm + col;
        s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
        Frag4<T> f;
        *reinterpret_cast<uint2*>(&f) = *reinterpret_cast<uint2*>(&v);
        return f;
    } else {
        return frag_ld<T>(tr + (z0 + (lane & 15)) * ld_tr + k0 + (lane >> 4) * 4);
    }
}

// wave-private staging of one (b, head)'s K / T_ / V slices and marks into LDS.
//   row-major [Tp][dh]  (A operand with the channel as contraction index)
//   transposed [dh][LDT] (A operand with the key as contraction index)
// All global loads of a batch are issued before the first LDS store (rows clamped, so the loads are straight-line code):
// the staging costs one memory latency per batch instead of one per 64 chunks — it was 42 of the forward kernel's 98 us.
template <typename T, int DT, int NT>
__device__ __forceinline__ void stage_rows(const T* src, int ld, int Tlen, T* rowmajor, T* transposed, int LDT, int lane) {
    constexpr int cpr = 4 * DT, dh = 16 * DT, NI = NT * DT;   // chunks of 4 elements: 64 * NI of them
    constexpr int BATCH = sizeof(T) == 2 ? 28 : 16;
#pragma unroll
    for (int i0 = 0; i0 < NI; i0 += BATCH) {
        Frag4<T> f[BATCH];
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
            if (i0 + j < NI) {
                const int c = lane + 64 * (i0 + j), k = c / cpr, u4 = (c % cpr) * 4;
                f[j] = frag_ld<T>(src + (long)min(k, Tlen - 1) * ld + u4);
            }
#pragma unroll
        for (int j = 0; j < BATCH; ++j)
            if (i0 + j < NI) {
                const int c = lane + 64 * (i0 + j), k = c / cpr, u4 = (c % cpr) * 4;
                if (k >= Tlen) f[j] = frag_zero<T>();
                if (rowmajor) {
                    if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(rowmajor + k * dh + u4) = *reinterpret_cast<uint4*>(&f[j]);
                    else *reinterpret_cast<uint2*>(rowmajor + k * dh + (img_swz<T>() ? swz_col<dh>(k, u4) : u4)) = *reinterpret_cast<uint2*>(&f[j]);
                }
                if (transposed) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) transposed[(u4 + r) * LDT + k] = f[j].v[r];
                }
            }
    }
}
// marks [T][E] uint8 -> [Tp][16] in the activation dtype: one key row per lane and round, its E bytes loaded together
// (one 16-byte load when E == 16)
template <typename T, int NT, int EC, bool SW = false>
__device__ __forceinline__ void stage_marks(const uint8_t* marks_row, int E, int Tlen, T* rowmajor, T* transposed, int LDT, int lane) {
    constexpr int Tp = 16 * NT, NR = (Tp + 63) / 64;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int k = lane + 64 * i;
        const uint8_t* row = marks_row + (long)min(k, Tlen - 1) * E;
        uint32_t w[4];
        if constexpr (EC == 16) {
            const uint4 v = *reinterpret_cast<const uint4*>(row);
            w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
        } else {
            uint8_t by[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) by[e] = row[min(e, E - 1)];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                w[j] = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) w[j] |= (4 * j + e < E ? (uint32_t)by[4 * j + e] : 0u) << (8 * e);
            }
        }
        if (k < Tp) {
            const bool ok = k < Tlen;
            T vals[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) vals[e] = from_f32<T>(ok ? (float)((w[e / 4] >> (8 * (e % 4))) & 0xffu) : 0.f);
            if (rowmajor) {
#pragma unroll
                for (int j = 0; j < (int)(16 * sizeof(T) / 16); ++j)
                    reinterpret_cast<uint4*>(rowmajor + k * EP)[SW ? (j ^ ((k >> 3) & 1)) : j] = reinterpret_cast<const uint4*>(vals)[j];
            }
            if (transposed) {
#pragma unroll
                for (int e = 0; e < 16; ++e) transposed[e * LDT + k] = vals[e];
            }
        }
    }
}

// ---- one round trip for a wave's whole staging -------------------------------------------------------------------------------
// stage_rows / stage_marks / load_keymask one after the other are 4-5 dependent round trips at the head of every wave (each
// waits for its own loads before its LDS stores).  At the small shapes (bf16, <= 8 fragments per lane and matrix, 16 marks)
// everything fits in registers at once: all loads of K / T_ / V / marks / ids go out first, then the stores.
template <typename T, int DT, int NT>
struct RowStage {
    static constexpr int cpr = 4 * DT, dh = 16 * DT, NI = NT * DT;
    Frag4<T> f[NI];
    __device__ __forceinline__ void load(const T* src, int ld, int Tlen, int lane) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int c = lane + 64 * j, k = c / cpr, u4 = (c % cpr) * 4;
            f[j] = frag_ld<T>(src + (long)min(k, Tlen - 1) * ld + u4);
        }
    }
    __device__ __forceinline__ void pin() {
        static_assert(sizeof(Frag4<T>) == 8, "bf16 fragments");
#pragma unroll
        for (int j = 0; j < NI; ++j) { uint2& u = *reinterpret_cast<uint2*>(&f[j]); asm volatile("" : "+v"(u.x), "+v"(u.y)); }
    }
    __device__ __forceinline__ void store(T* rowmajor, T* transposed, int LDT, int Tlen, int lane) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int c = lane + 64 * j, k = c / cpr, u4 = (c % cpr) * 4;
            if (k >= Tlen) f[j] = frag_zero<T>();
            if (rowmajor) *reinterpret_cast<uint2*>(rowmajor + k * dh + (img_swz<T>() ? swz_col<dh>(k, u4) : u4)) = *reinterpret_cast<uint2*>(&f[j]);
            if (transposed) {
#pragma unroll
                for (int r = 0; r < 4; ++r) transposed[(u4 + r) * LDT + k] = f[j].v[r];
            }
        }
    }
};
template <typename T, int DT, int NT, int EC>
constexpr bool stage_merged() { return sizeof(T) == 2 && NT * DT <= 8 && EC == 16; }

// k / t / v: source (null = not staged), row-major image, transposed image (either may be null)
template <typename T, int DT, int NT, int EC>
__device__ __forceinline__ KeyMask<NT> stage_wave(const T* k_src, T* k_rm, T* k_tr, const T* t_src, T* t_rm, T* t_tr,
                                                  const T* v_src, T* v_rm, T* v_tr, int ld, const uint8_t* marks_row, int E,
                                                  T* m_rm, T* m_tr, const int64_t* ids_row, float* lds_madd, int Tlen, int LDT,
                                                  int lane) {
    if constexpr (stage_merged<T, DT, NT, EC>()) {
        constexpr int Tp = 16 * NT, NR = (Tp + 63) / 64;
        RowStage<T, DT, NT> sk, st, sv;
        uint4 mk[NR];
        int64_t id[NR];
        sk.load(k_src, ld, Tlen, lane);
        if (t_src) st.load(t_src, ld, Tlen, lane);
        if (v_src) sv.load(v_src, ld, Tlen, lane);
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int k = min(lane + 64 * i, Tlen - 1);
            if (marks_row) mk[i] = *reinterpret_cast<const uint4*>(marks_row + (long)k * 16);
            id[i] = ids_row[k];
        }
        sk.pin();
        if (t_src) st.pin();
        if (v_src) sv.pin();
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            if (marks_row) asm volatile("" : "+v"(mk[i].x), "+v"(mk[i].y), "+v"(mk[i].z), "+v"(mk[i].w));
            asm volatile("" : "+v"(id[i]));
        }
        sk.store(k_rm, k_tr, LDT, Tlen, lane);
        if (t_src) st.store(t_rm, t_tr, LDT, Tlen, lane);
        if (v_src) sv.store(v_rm, v_tr, LDT, Tlen, lane);
        if (marks_row) {
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int k = lane + 64 * i;
                if (k < Tp) {
                    const bool ok = k < Tlen;
                    const uint32_t w[4] = {mk[i].x, mk[i].y, mk[i].z, mk[i].w};
                    T vals[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) vals[e] = from_f32<T>(ok ? (float)((w[e / 4] >> (8 * (e % 4))) & 0xffu) : 0.f);
                    if (m_rm) {
#pragma unroll
                        for (int j = 0; j < (int)(16 * sizeof(T) / 16); ++j)
                            reinterpret_cast<uint4*>(m_rm + k * EP)[img_swz<T>() ? (j ^ ((k >> 3) & 1)) : j] = reinterpret_cast<const uint4*>(vals)[j];
                    }
                    if (m_tr) {
#pragma unroll
                        for (int e = 0; e < 16; ++e) m_tr[e * LDT + k] = vals[e];
                    }
                }
            }
        }
        KeyMask<NT> km;
        km.pad = 0ull;
        km.madd = lds_madd;
        uint64_t padded[NR], real[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int k = lane + 64 * i;
            const bool pd = k < Tlen && id[i] == 0;
            padded[i] = __ballot(pd);
            real[i] = __ballot(k < Tlen && id[i] != 0);
            if (k < Tp) lds_madd[k] = k >= Tlen ? -INFINITY : (pd ? -4294967296.0f : 0.f);
        }
        km.kt0 = first_real_tile<NR>(real);
        const int g4 = (lane >> 4) * 4;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
            const uint64_t nib = (padded[kt / 4] >> ((kt % 4) * 16 + g4)) & 0xfull;
            km.pad |= nib << (kt * 4);
        }
        return km;
    } else {
        stage_rows<T, DT, NT>(k_src, ld, Tlen, k_rm, k_tr, LDT, lane);
        if (t_src) stage_rows<T, DT, NT>(t_src, ld, Tlen, t_rm, t_tr, LDT, lane);
        if (v_src) stage_rows<T, DT, NT>(v_src, ld, Tlen, v_rm, v_tr, LDT, lane);
        if (marks_row) stage_marks<T, NT, EC, img_swz<T>()>(marks_row, E, Tlen, m_rm, m_tr, LDT, lane);
        return load_keymask<NT>(ids_row, Tlen, lane, lds_madd);
    }
}

}  // namespace bimau

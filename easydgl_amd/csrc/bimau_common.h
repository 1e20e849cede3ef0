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
#include "edgl_common.h"

namespace bimau {

constexpr int EP = 16;  // marks padded to one MFMA K-block
constexpr float NLOG2E = -1.4426950408889634f;

// sigmoid of a pre-activation that was already multiplied by -log2(e)
__device__ __forceinline__ float sigmoid_pre(float xs) { return fast_rcp(1.0f + __builtin_amdgcn_exp2f(xs)); }

// ---- intensity-weight pack (built once per launch by pack_kernel, copied to LDS by every WG) ---
// T-typed: W1T [JE][LDW]  (W1T[j][u] = -log2e * W1[u][j], u < dh) -> A operand of Zpre'^T = W1T.H^T
//          W1R [dh][LDR]  (W1R[u][j] = W1[u][j], unscaled)       -> A operand of dH^T = W1.du^T
// f32:     w1s [JE] = -log2e * W1[dh][j] (interval weight), b1 [JE] = -log2e * b1, wv [JE] = w.flatten(),
//          sc[16] = exp(scaling), isc[16] = 1/sc.
// The -log2(e) factor turns sigmoid(x) into rcp(1 + exp2(x')) — one v_exp_f32 and one v_rcp_f32, no multiply.
struct PackDims {
    int dh, E, JE, LDW, LDR;
    size_t off_w1r, off_f32, bytes;  // byte offsets
};
template <typename T>
__host__ __device__ inline PackDims pack_dims(int dh, int E) {
    PackDims d;
    d.dh = dh; d.E = E; d.JE = dh * E; d.LDW = dh + 4; d.LDR = d.JE + 4;
    size_t w1t = (size_t)d.JE * d.LDW * sizeof(T);
    w1t = (w1t + 15) & ~(size_t)15;
    size_t w1r = (size_t)dh * d.LDR * sizeof(T);
    w1r = (w1r + 15) & ~(size_t)15;
    d.off_w1r = w1t;
    d.off_f32 = w1t + w1r;
    d.bytes = d.off_f32 + ((size_t)3 * d.JE + 2 * EP) * sizeof(float);
    d.bytes = (d.bytes + 15) & ~(size_t)15;
    return d;
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
template <int NT>
struct KeyMask { const float* madd; uint64_t pad; };   // pad: one bit per (key tile, register), NT <= 16   // madd: wave-private LDS array [16*NT]
template <int NT>
__device__ __forceinline__ KeyMask<NT> load_keymask(const int64_t* ids_row, int T, int lane, float* lds_madd) {
    KeyMask<NT> km;
    km.pad = 0ull;
    km.madd = lds_madd;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kt * 16 + (lane >> 4) * 4 + r;
            if (k < T && ids_row[k] == 0) km.pad |= 1ull << (kt * 4 + r);
        }
    for (int k = lane; k < 16 * NT; k += 64) {
        float v = -INFINITY;
        if (k < T) v = ids_row[k] == 0 ? -4294967296.0f : 0.f;
        lds_madd[k] = v;
    }
    return km;
}

// S^T tiles -> normalised P^T tiles (temporal.py:422-429).  s[kt][r] in: raw Q.K; out: softmax.
// `causal` (MAU.__call__ with causality=True, temporal.py:370-375): keys k > q get the same -2^32+1 score as padded keys
// (tf.where(tril == 0, paddings, outputs)); q = this lane's query index.
constexpr int MAU_CAUSAL = 1;    // future blinding (temporal.py:370-375)
constexpr int MAU_NO_DIAG = 2;   // MAU keeps the modulation on the diagonal; BiMAU sets it to 1 (temporal.py:438-439)
template <int NT>
__device__ __forceinline__ void masked_softmax(f32x4 (&s)[NT], const KeyMask<NT>& km, float cscale, int lane, int q = 0,
                                               bool causal = false) {
    float mx = -INFINITY;
    const float* mrow = km.madd + (lane >> 4) * 4;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
        const float4 m4 = *reinterpret_cast<const float4*>(mrow + kt * 16);
        const float mm[4] = {m4.x, m4.y, m4.z, m4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = fmaf(s[kt][r], cscale, mm[r]);
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
            // (s - mx) first: exact 0 for the row maximum even at the -2^32 padding magnitude
            const float e = __expf(s[kt][r] - mx);
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

// A/B fragment with the CONTRACTION index along the rows of a tile: v[j] = X[k0 + 4G + j][z0 + (l&15)].
//   bf16: one ds_read_b64_tr_b16 on the ROW-MAJOR tile rm[k][z] (ld_rm elements per row)
//   f32 : no 32-bit transpose read exists -> plain read of a separately staged transposed image tr[z][k]
template <typename T>
__device__ __forceinline__ Frag4<T> kfrag(const T* rm, int ld_rm, const T* tr, int ld_tr, int k0, int z0, int lane) {
    if constexpr (sizeof(T) == 2) {
        const int G = lane >> 4, s = lane & 15;
        typedef __attribute__((ext_vector_type(4))) short s4;
        const T* p = rm + (k0 + 4 * G + (s >> 2)) * ld_rm + z0 + 4 * (s & 3);
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
template <typename T>
__device__ __forceinline__ void stage_rows(const T* src, int ld, int Tlen, int Tp, int dh, T* rowmajor, T* transposed,
                                           int LDT, int lane) {
    const int cpr = dh / 4, nchunk = Tp * cpr;
    for (int c = lane; c < nchunk; c += 64) {
        const int k = c / cpr, u4 = (c % cpr) * 4;
        Frag4<T> f = (k < Tlen) ? frag_ld<T>(src + (long)k * ld + u4) : frag_zero<T>();
        if (rowmajor) {
            if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(rowmajor + k * dh + u4) = *reinterpret_cast<uint4*>(&f);
            else *reinterpret_cast<uint2*>(rowmajor + k * dh + u4) = *reinterpret_cast<uint2*>(&f);
        }
        if (transposed) {
#pragma unroll
            for (int i = 0; i < 4; ++i) transposed[(u4 + i) * LDT + k] = f.v[i];
        }
    }
}
template <typename T>
__device__ __forceinline__ void stage_marks(const uint8_t* marks_row, int E, int Tlen, int Tp, T* rowmajor,
                                            T* transposed, int LDT, int lane) {
    for (int i = lane; i < Tp * EP; i += 64) {
        const int k = i / EP, e = i % EP;
        const float v = (k < Tlen && e < E) ? (float)marks_row[(long)k * E + e] : 0.f;
        if (rowmajor) rowmajor[k * EP + e] = from_f32<T>(v);
        if (transposed) transposed[e * LDT + k] = from_f32<T>(v);
    }
}

}  // namespace bimau

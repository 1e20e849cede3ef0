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

// ---- intensity-weight pack (built once per launch by pack_kernel, copied to LDS by every WG) ---
// T-typed: W1T [JE][LDW]  (W1T[j][u] = W1[u][j], u < dh)        -> A operand of Zpre^T = W1^T.H^T
//          W1R [dh][LDR]  (W1R[u][j] = W1[u][j])                -> A operand of dH^T = W1.du^T
// f32:     w1s [JE] = W1[dh][j] (interval weight), b1 [JE], wv [JE] = w.flatten(), sc[16]=exp(scaling)
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
        w1t[i] = from_f32<T>(u < dh ? W1[(long)u * d.JE + j] : 0.f);
    }
    for (int i = tid; i < dh * d.LDR; i += nth) {
        const int u = i / d.LDR, j = i % d.LDR;
        w1r[i] = from_f32<T>(j < d.JE ? W1[(long)u * d.JE + j] : 0.f);
    }
    for (int j = tid; j < d.JE; j += nth) {
        f[j] = W1[(long)dh * d.JE + j];
        f[d.JE + j] = b1[j];
        f[2 * d.JE + j] = w[j];
    }
    for (int e = tid; e < EP; e += nth) {
        const float s = e < E ? __expf(scaling[e]) : 1.f;
        f[3 * d.JE + e] = s;
        f[3 * d.JE + EP + e] = 1.f / s;
    }
}

// per-wave key bookkeeping: bit (kt*4+r) for key k = kt*16 + (lane>>4)*4 + r
struct KeyBits { uint32_t real, pad; };
template <int NT>
__device__ __forceinline__ KeyBits load_keybits(const int64_t* ids_row, int T, int lane) {
    KeyBits kb{0u, 0u};
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int k = kt * 16 + (lane >> 4) * 4 + r;
            if (k < T) {
                kb.real |= 1u << (kt * 4 + r);
                if (ids_row[k] == 0) kb.pad |= 1u << (kt * 4 + r);
            }
        }
    return kb;
}

// S^T tiles -> normalised P^T tiles (temporal.py:422-429).  s[kt][r] in: raw Q.K; out: softmax.
template <int NT>
__device__ __forceinline__ void masked_softmax(f32x4 (&s)[NT], const KeyBits& kb, float cscale) {
    const float PADV = -4294967296.0f;  // float32(-2**32+1), temporal.py:425
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const uint32_t bit = 1u << (kt * 4 + r);
            float v = s[kt][r] * cscale;
            v = (kb.pad & bit) ? PADV : v;
            v = (kb.real & bit) ? v : -INFINITY;
            s[kt][r] = v;
            mx = fmaxf(mx, v);
        }
    mx = group_max4(mx);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
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
    const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0;
    float y[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float send = b5 ? z[i] : z[i + 8];
        const float keep = b5 ? z[i + 8] : z[i];
        y[i] = keep + __shfl_xor(send, 32, 64);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float send = b4 ? y[i] : y[i + 4];
        const float keep = b4 ? y[i + 4] : y[i];
        out[i] = keep + __shfl_xor(send, 16, 64);
    }
}
// inverse: every lane ends with all 16 values (value for e = 4g+i lives in lane group g)
__device__ __forceinline__ void all_gather16(const float (&in)[4], float (&out)[16], int lane) {
    const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0;
    float y[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float other = __shfl_xor(in[i], 16, 64);
        y[i] = b4 ? other : in[i];
        y[i + 4] = b4 ? in[i] : other;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float other = __shfl_xor(y[i], 32, 64);
        out[i] = b5 ? other : y[i];
        out[i + 8] = b5 ? y[i] : other;
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

// K3 at head dims 64 / 128 — the shapes of the reference's published recipes (runme.sh:15-23: num_units 512, 8 heads ->
// dh = 64; runme.sh:107-115 CTSMA: 4 heads -> dh = 128).  The intensity MLP of MAU.intensity (temporal.py:281-315) has
// dh*(dh+1)*E weights — 133 KB at dh = 64, 528 KB at dh = 128 in bf16 — which no longer sit in the LDS of an attention
// workgroup, so the MLP becomes its own GEMM-shaped kernels over row tiles of ALL (b', q) rows, streaming the weights of one
// mark at a time through LDS:
//   forward   scores phase (S, softmax, H = P.T_ -> H rows)  ->  intensity_fwd_big (H rows -> z, lambda)  ->  values phase
//             (S, P recomputed, G = lambda.marks^T, diag := 1, dropout, O = A.V + residual)
//   backward  sweep 1 (dlambda -> dz, row term, dV)  ->  intensity_bwd_rows_big (dH)  +  intensity_bwd_weights_big (dW1, db1,
//             dw partials over row splits)  ->  sweep 2 (dQ, dK, dT_)
// The attention phases are the kernels of bimau_fwd_impl.h / bimau_bwd_impl.h at DT = 4 / 8.
#include "bimau_bwd_impl.h"
#include "bimau_fwd_impl.h"
#include <type_traits>

namespace {
using namespace bimau;

struct IntP {
    const void* hin; const float* spans; const char* pack; const float* dz;
    long R; int B, T, E;
    float* z_out; float* lam;     // forward outputs
    float* dh_out;                // rows kernel: dH [R, dh] f32
    float* wpart; const float* dsc_part; long njobs;   // weights kernel
};

// cooperative copy of `bytes` (multiple of 16) global -> LDS
__device__ __forceinline__ void copy16(const char* src, char* dst, int bytes) {
    for (int i = threadIdx.x * 16; i < bytes; i += blockDim.x * 16)
        *reinterpret_cast<uint4*>(dst + i) = *reinterpret_cast<const uint4*>(src + i);
}

// One mark's weights on their way global -> registers -> LDS (256-thread workgroups): the dh rows of the packed W1^T, the
// (interval weight | bias | output weight) floats and — backward, row side — the mark's dh x dh block of W1 columns.  fetch()
// issues every load (clamped, unconditional), put() writes the LDS chunk [W1^T rows | W1 columns | floats].  The kernels
// fetch mark e + 1 before they compute mark e out of the other of two LDS chunks: one barrier per mark, and the L2 round
// trip of a chunk runs under the arithmetic of the previous one (with one chunk and a copy loop between two barriers every
// mark of every row block opened with that round trip: 16 per block at the recipe shape, ~40 % of the kernels' time).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
template <typename T, int DT, bool WITH_R>
struct MarkChunk {
    static constexpr int dh = 16 * DT, LDC = dh + 4;
    static constexpr int WB = dh * (dh + 4) * (int)sizeof(T);        // bytes of the W1^T rows (LDW = dh + 4)
    static constexpr int NV = (WB / 16 + 255) / 256;
    static constexpr int NR = WITH_R ? dh * (dh / 4) / 256 : 1;      // 4-channel fragments of the W1 column block per thread
    static constexpr int NF = (3 * dh + 255) / 256;
    static constexpr int BYTES = WB + (WITH_R ? WB : 0) + 3 * dh * (int)sizeof(float);
    static_assert(!WITH_R || (dh * (dh / 4)) % 256 == 0, "column block: whole rounds of the workgroup");
    using RFrag = typename std::conditional<sizeof(T) == 4, u32x4, u32x2>::type;   // four channels of the column block
};
// (first-class vector types: arrays of HIP's uint4 structs stayed in scratch here — a store and a reload per prefetch)
template <typename T, int DT, bool WITH_R, int NV, typename RF, int NR, int NF>
__device__ __forceinline__ void mark_fetch(u32x4 (&v)[NV], RF (&r)[NR], float (&f)[NF], const char* pack, const PackDims& pd, int e) {
    using MC = MarkChunk<T, DT, WITH_R>;
    constexpr int dh = MC::dh;
    const u32x4* src = reinterpret_cast<const u32x4*>(pack + (size_t)e * dh * pd.LDW * sizeof(T));
#pragma unroll
    for (int i = 0; i < MC::NV; ++i) v[i] = src[min((int)threadIdx.x + i * 256, MC::WB / 16 - 1)];
    if constexpr (WITH_R) {
        const T* W1R = reinterpret_cast<const T*>(pack + pd.off_w1r);
#pragma unroll
        for (int i = 0; i < MC::NR; ++i) {
            const int k = threadIdx.x + i * 256, u = k / (dh / 4), c4 = (k % (dh / 4)) * 4;
            r[i] = *reinterpret_cast<const typename MC::RFrag*>(W1R + (size_t)u * pd.LDR + e * dh + c4);
        }
    }
    const float* fW = reinterpret_cast<const float*>(pack + pd.off_f32);
#pragma unroll
    for (int i = 0; i < MC::NF; ++i) {
        const int k = min((int)threadIdx.x + i * 256, 3 * dh - 1);
        f[i] = fW[(k / dh) * pd.JE + e * dh + (k % dh)];
    }
}
template <typename T, int DT, bool WITH_R, int NV, typename RF, int NR, int NF>
__device__ __forceinline__ void mark_put(const u32x4 (&v)[NV], const RF (&r)[NR], const float (&f)[NF], char* chunk) {
    using MC = MarkChunk<T, DT, WITH_R>;
    constexpr int dh = MC::dh;
#pragma unroll
    for (int i = 0; i < MC::NV; ++i) {
        const int k = threadIdx.x + i * 256;
        if (k < MC::WB / 16) reinterpret_cast<u32x4*>(chunk)[k] = v[i];
    }
    if constexpr (WITH_R) {
        T* Rc = reinterpret_cast<T*>(chunk + MC::WB);
#pragma unroll
        for (int i = 0; i < MC::NR; ++i) {
            const int k = threadIdx.x + i * 256, u = k / (dh / 4), c4 = (k % (dh / 4)) * 4;
            *reinterpret_cast<typename MC::RFrag*>(Rc + u * MC::LDC + c4) = r[i];
        }
    }
    float* fc = reinterpret_cast<float*>(chunk + MC::WB + (WITH_R ? MC::WB : 0));
#pragma unroll
    for (int i = 0; i < MC::NF; ++i) {
        const int k = threadIdx.x + i * 256;
        if (k < 3 * dh) fc[k] = f[i];
    }
}
// two LDS chunks where the staging registers are affordable: bf16 at head dim 64 (13 / 21 registers; head dim 128 would stage
// 38 / 70 and spill, f32 twice that)
template <typename T, int DT> constexpr bool big_double_buffer() { return sizeof(T) == 2 && DT == 4; }

// ------------------------------------------------------------------------------------------------------------------
// forward MLP: z[row][e] = sum_u sigmoid([H[row], span] . W1[:, e*dh+u] + b1) * w[e][u]; lambda = s_e softplus(z / s_e)
// A workgroup (4 waves) walks blocks of 128 rows (2 row tiles of 16 per wave); per mark e it stages that mark's dh rows
// of the packed W1^T (+ interval weight, bias, output weight) in LDS, so every weight byte read from L2 serves 128 rows.
// Register layout as in the fused kernel: Zpre^T[j][row], L(first = j, second = row).
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int DT>
__global__ __launch_bounds__(256) void intensity_fwd_big_kernel(IntP p) {
    constexpr int dh = 16 * DT, RT = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PackDims pd = pack_dims<T>(dh, p.E);
    const int LDW = pd.LDW;
    constexpr int CH_T = dh * (dh + 4);                   // elements of one mark's W1^T rows
    using MC = MarkChunk<T, DT, false>;
    constexpr bool DBUF = big_double_buffer<T, DT>();
    u32x4 mv[MC::NV]; typename MC::RFrag mr[MC::NR]; float mf[MC::NF];
    int cur = 0;                                          // LDS chunk of the mark being computed
    if constexpr (DBUF) mark_fetch<T, DT, false>(mv, mr, mf, p.pack, pd, 0);
    const char* packW = p.pack;
    const float* fW = reinterpret_cast<const float*>(p.pack + pd.off_f32);
    const float* scs = fW + 3 * pd.JE; const float* iscs = scs + EP;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int ntq = (p.T + 15) / 16;
    const long ntile = (p.R / p.T) * ntq;
    const T* hin = reinterpret_cast<const T*>(p.hin);
    for (long tb = (long)blockIdx.x * 4 * RT; tb < ntile; tb += (long)gridDim.x * 4 * RT) {
        Frag4<T> hf[RT][DT];
        float span[RT];
        long row0[RT]; bool rok[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const long tile = min(tb + wave * RT + t, ntile - 1);
            const long bpq = tile / ntq; const int qt = (int)(tile - bpq * ntq), bb = (int)(bpq % p.B);
            const int q = min(qt * 16 + l15, p.T - 1);
            row0[t] = bpq * p.T;
            rok[t] = (tb + wave * RT + t < ntile) && (qt * 16 + l15 < p.T);
            row0[t] += q;     // this lane's row
#pragma unroll
            for (int ub = 0; ub < DT; ++ub) hf[t][ub] = frag_ld<T>(hin + row0[t] * dh + ub * 16 + g4);
            span[t] = p.spans[(long)bb * p.T + q];
        }
        // z of this lane's row: lane group g keeps the marks e = 4g + i (the layout reduce_scatter16 leaves)
        float z4a[RT][4];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int i = 0; i < 4; ++i) z4a[t][i] = 0.f;
#pragma unroll 1
        for (int e = 0; e < p.E; ++e) {
            const T* Wc = reinterpret_cast<const T*>(smem + (size_t)cur * MC::BYTES);
            const float* fc = reinterpret_cast<const float*>(smem + (size_t)cur * MC::BYTES + MC::WB);   // ws | bs | wv, dh floats each
            if constexpr (DBUF) {
                mark_put<T, DT, false>(mv, mr, mf, smem + (size_t)cur * MC::BYTES);   // (its last readers passed the previous mark's barrier)
                mark_fetch<T, DT, false>(mv, mr, mf, p.pack, pd, e + 1 < p.E ? e + 1 : 0);   // next mark — or mark 0 of the next row block
                __syncthreads();
                cur ^= 1;
            } else {
                __syncthreads();      // the previous mark's chunk is no longer read
                copy16(packW + (size_t)e * dh * LDW * sizeof(T), smem, CH_T * (int)sizeof(T));
                for (int i = threadIdx.x; i < 3 * dh; i += blockDim.x)
                    reinterpret_cast<float*>(smem + MC::WB)[i] = fW[(i / dh) * pd.JE + e * dh + (i % dh)];
                __syncthreads();
            }
            float zc[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) zc[t] = 0.f;
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                Frag4<T> w[DT];
#pragma unroll
                for (int ub = 0; ub < DT; ++ub) w[ub] = frag_ld<T>(Wc + (d * 16 + l15) * LDW + ub * 16 + g4);
                const float4 ws = *reinterpret_cast<const float4*>(fc + d * 16 + g4);
                const float4 bs = *reinterpret_cast<const float4*>(fc + dh + d * 16 + g4);
                const float4 wv = *reinterpret_cast<const float4*>(fc + 2 * dh + d * 16 + g4);
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int ub = 0; ub < DT; ++ub) a = mma16(w[ub], hf[t][ub], a);
                    zc[t] += sigmoid_pre(fmaf(span[t], ws.x, a[0]) + bs.x) * wv.x + sigmoid_pre(fmaf(span[t], ws.y, a[1]) + bs.y) * wv.y +
                             sigmoid_pre(fmaf(span[t], ws.z, a[2]) + bs.z) * wv.z + sigmoid_pre(fmaf(span[t], ws.w, a[3]) + bs.w) * wv.w;
                }
            }
#pragma unroll
            for (int t = 0; t < RT; ++t) {      // sum over the four lane groups (each holds 4 of a tile's 16 channels)
                FPair s = swap32(zc[t], zc[t]);
                const float h = s.first + s.second;
                s = swap16(h, h);
                const float full = s.first + s.second;
#pragma unroll
                for (int i = 0; i < 4; ++i) z4a[t][i] = (e == g4 + i) ? full : z4a[t][i];
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            float z4[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) z4[i] = z4a[t][i];
            if (rok[t]) {
                *reinterpret_cast<float4*>(p.z_out + row0[t] * EP + g4) = make_float4(z4[0], z4[1], z4[2], z4[3]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (g4 + i < p.E) p.lam[row0[t] * p.E + g4 + i] = scs[g4 + i] * __logf(1.0f + __expf(z4[i] * iscs[g4 + i]));   // temporal.py:305-306
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward, row side: dH[row][u] = sum_j du[row][j] W1[u][j], du = dz[row][e(j)] w[j] Z (1 - Z).  Same walk as the forward
// (128 rows per workgroup pass, one mark's W1^T rows and W1 columns in LDS at a time).
// ------------------------------------------------------------------------------------------------------------------
template <typename T, int DT>
__global__ __launch_bounds__(256) void intensity_bwd_rows_big_kernel(IntP p) {
    constexpr int dh = 16 * DT, RT = 2, LDC = dh + 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PackDims pd = pack_dims<T>(dh, p.E);
    const int LDW = pd.LDW;
    using MC = MarkChunk<T, DT, true>;                      // chunk: W1^T rows [dh j][LDW] | W1 columns [dh u][LDC] | floats
    constexpr bool DBUF = big_double_buffer<T, DT>();
    u32x4 mv[MC::NV]; typename MC::RFrag mr[MC::NR]; float mf[MC::NF];
    int cur = 0;
    if constexpr (DBUF) mark_fetch<T, DT, true>(mv, mr, mf, p.pack, pd, 0);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int ntq = (p.T + 15) / 16;
    const long ntile = (p.R / p.T) * ntq;
    const T* hin = reinterpret_cast<const T*>(p.hin);
    for (long tb = (long)blockIdx.x * 4 * RT; tb < ntile; tb += (long)gridDim.x * 4 * RT) {
        Frag4<T> hf[RT][DT];
        float span[RT];
        long row[RT]; bool rok[RT];
        const float* dzp[RT];
        float dzn[RT];
        f32x4 dHt[RT][DT];       // dH^T[u][row], L(first = u, second = row)
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            const long tile = min(tb + wave * RT + t, ntile - 1);
            const long bpq = tile / ntq; const int qt = (int)(tile - bpq * ntq), bb = (int)(bpq % p.B);
            const int q = min(qt * 16 + l15, p.T - 1);
            rok[t] = (tb + wave * RT + t < ntile) && (qt * 16 + l15 < p.T);
            row[t] = bpq * p.T + q;       // this lane's row (clamped; masked through dz)
#pragma unroll
            for (int ub = 0; ub < DT; ++ub) {
                hf[t][ub] = frag_ld<T>(hin + row[t] * dh + ub * 16 + g4);
                dHt[t][ub] = zero4;
            }
            span[t] = p.spans[(long)bb * p.T + q];
            // dz of the lane's row, one mark ahead and requested BEFORE the chunk prefetch (a load behind it could only be
            // waited for together with it)
            dzp[t] = p.dz + row[t] * EP;
            dzn[t] = dzp[t][0];
        }
#pragma unroll 1
        for (int e = 0; e < p.E; ++e) {
            const T* Wc = reinterpret_cast<const T*>(smem + (size_t)cur * MC::BYTES);
            const T* Rc = reinterpret_cast<const T*>(smem + (size_t)cur * MC::BYTES + MC::WB);
            const float* fc = reinterpret_cast<const float*>(smem + (size_t)cur * MC::BYTES + 2 * MC::WB);
            float dzr[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) dzr[t] = rok[t] ? dzn[t] : 0.f;
            if constexpr (DBUF) {
                mark_put<T, DT, true>(mv, mr, mf, smem + (size_t)cur * MC::BYTES);
#pragma unroll
                for (int t = 0; t < RT; ++t) dzn[t] = dzp[t][min(e + 1, p.E - 1)];
                mark_fetch<T, DT, true>(mv, mr, mf, p.pack, pd, e + 1 < p.E ? e + 1 : 0);
                __syncthreads();
                cur ^= 1;
            } else {
                __syncthreads();
                mark_fetch<T, DT, true>(mv, mr, mf, p.pack, pd, e);
                mark_put<T, DT, true>(mv, mr, mf, smem);
#pragma unroll
                for (int t = 0; t < RT; ++t) dzn[t] = dzp[t][min(e + 1, p.E - 1)];
                __syncthreads();
            }
            // the forward's orientation, Zpre^T[j][row] with L(first = j, second = row): du^T leaves the sigmoid block in the
            // B-operand layout of dH^T[u][row] = sum_j W1[u][j] du[row][j] (no transposing product), a lane's four u of its
            // row are one 16-byte store
#pragma unroll
            for (int d = 0; d < DT; ++d) {
                Frag4<T> w[DT];
#pragma unroll
                for (int ub = 0; ub < DT; ++ub) w[ub] = frag_ld<T>(Wc + (d * 16 + l15) * LDW + ub * 16 + g4);
                const float4 ws4 = *reinterpret_cast<const float4*>(fc + d * 16 + g4);
                const float4 bs4 = *reinterpret_cast<const float4*>(fc + dh + d * 16 + g4);
                const float4 wv4 = *reinterpret_cast<const float4*>(fc + 2 * dh + d * 16 + g4);
                const float ws[4] = {ws4.x, ws4.y, ws4.z, ws4.w}, bs[4] = {bs4.x, bs4.y, bs4.z, bs4.w}, wv[4] = {wv4.x, wv4.y, wv4.z, wv4.w};
                Frag4<T> rc[DT];
#pragma unroll
                for (int ut = 0; ut < DT; ++ut) rc[ut] = frag_ld<T>(Rc + (ut * 16 + l15) * LDC + d * 16 + g4);
#pragma unroll
                for (int t = 0; t < RT; ++t) {
                    f32x4 a = zero4;
#pragma unroll
                    for (int ub = 0; ub < DT; ++ub) a = mma16(w[ub], hf[t][ub], a);
                    f32x4 du;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = sigmoid_pre(fmaf(span[t], ws[r], a[r]) + bs[r]);
                        du[r] = dzr[t] * z * wv[r] * (1.0f - z);
                    }
                    const Frag4<T> duB = frag_from_acc<T>(du);
#pragma unroll
                    for (int ut = 0; ut < DT; ++ut) dHt[t][ut] = mma16(rc[ut], duB, dHt[t][ut]);
                }
            }
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            if (!rok[t]) continue;
            float* dst = p.dh_out + row[t] * dh;
#pragma unroll
            for (int ut = 0; ut < DT; ++ut)
                *reinterpret_cast<float4*>(dst + ut * 16 + g4) = make_float4(dHt[t][ut][0], dHt[t][ut][1], dHt[t][ut][2], dHt[t][ut][3]);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------
// backward, weight side: dW1[u][j] = sum_row [H, span][row][u] du[row][j], db1[j] = sum du, dw[j] = sum dz Z.
// grid = (row splits, channel groups): a workgroup owns NJ = big_nj(dh) channel tiles of 16 (their W1^T rows stay in LDS, the
// dW1 tiles in registers) and one slice of the row tiles; the per-split partials are reduced by edgl_reduce_rows.
// ------------------------------------------------------------------------------------------------------------------
template <int DT> struct WGroup { static constexpr int NJ = big_nj(16 * DT); };   // 4 tiles at dh = 64 and at dh = 128

template <typename T, int DT>
__global__ __launch_bounds__(256) void intensity_bwd_weights_big_kernel(IntP p) {
    constexpr int dh = 16 * DT, NJ = WGroup<DT>::NJ, NC = NJ * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PackDims pd = pack_dims<T>(dh, p.E);
    const int LDW = pd.LDW, JE = pd.JE, NPAR = (dh + 3) * JE, NPARX = NPAR + EP;
    const int j0 = blockIdx.y * NC;                         // first channel of the group
    T* Wc = reinterpret_cast<T*>(smem);                     // [NC][LDW]
    float* fc = reinterpret_cast<float*>(smem + (size_t)NC * (dh + 4) * sizeof(T));   // ws | bs | wv [NC] each
    float* accs = fc + 3 * NC;                               // [(dh + 3)][NC]
    const float* fW = reinterpret_cast<const float*>(p.pack + pd.off_f32);
    const int nval = min(NC, JE - j0);                      // the last group may be partial (dh * E not a multiple of NC)
    copy16(p.pack + (size_t)j0 * LDW * sizeof(T), reinterpret_cast<char*>(Wc), nval * (dh + 4) * (int)sizeof(T));
    for (int i = threadIdx.x * 16; i < (NC - nval) * (dh + 4) * (int)sizeof(T); i += blockDim.x * 16)
        *reinterpret_cast<uint4*>(reinterpret_cast<char*>(Wc) + (size_t)nval * (dh + 4) * sizeof(T) + i) = make_uint4(0, 0, 0, 0);
    for (int i = threadIdx.x; i < 3 * NC; i += blockDim.x) fc[i] = (i % NC) < nval ? fW[(i / NC) * JE + j0 + (i % NC)] : 0.f;
    for (int i = threadIdx.x; i < (dh + 3) * NC; i += blockDim.x) accs[i] = 0.f;
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const Frag4<T> ident = identity_frag<T>(lane);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int ntq = (p.T + 15) / 16;
    const long ntile = (p.R / p.T) * ntq;
    const T* hin = reinterpret_cast<const T*>(p.hin);
    f32x4 dW[NJ][DT];   // tile (channel tile jj, u tile ub), L(first = j, second = u)
    float adb[NJ], adws[NJ], adw[NJ];
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj) {
        adb[jj] = 0.f; adws[jj] = 0.f; adw[jj] = 0.f;
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) dW[jj][ub] = zero4;
    }
    // Operands of a row tile — the H rows, the intervals, dz of the (at most two) marks this channel group spans — are fetched ONE
    // TILE AHEAD with unconditional, clamped loads and masked when consumed: with `ok ? load : 0` per value every tile paid three
    // dependent round trips (32 tiles per wave: 235 us at the 512-unit recipe shape, ~6 us per tile for ~2 us of arithmetic).
    struct Pre { Frag4<T> hA[DT]; float spn[4]; float dz0[4], dz1[4]; };
    const int e_lo = min(j0 / dh, EP - 1), e_hi = min(e_lo + 1, EP - 1);
    auto load_tile = [&](long t) {
        Pre o;
        t = min(t, ntile - 1);
        const long bpq = t / ntq; const int qt = (int)(t - bpq * ntq), bb = (int)(bpq % p.B);
        const long rowq = bpq * p.T;
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) o.hA[ub] = frag_ld<T>(hin + (rowq + min(qt * 16 + l15, p.T - 1)) * dh + ub * 16 + g4);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = min(qt * 16 + g4 + r, p.T - 1);
            o.spn[r] = p.spans[(long)bb * p.T + q];
            o.dz0[r] = p.dz[(rowq + q) * EP + e_lo];
            o.dz1[r] = p.dz[(rowq + q) * EP + e_hi];
        }
        return o;
    };
    const long tstep = (long)gridDim.x * 4;
    long t = (long)blockIdx.x * 4 + wave;
    Pre cur = load_tile(t);
    for (; t < ntile; t += tstep) {
        const Pre nxt = load_tile(t + tstep);
        const long bpq = t / ntq; const int qt = (int)(t - bpq * ntq);
        const bool okA = qt * 16 + l15 < p.T;
        Frag4<T> hA[DT], hB[DT];
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) {
            hA[ub] = okA ? cur.hA[ub] : frag_zero<T>();
            hB[ub] = frag_from_acc<T>(mma16(hA[ub], ident, zero4));   // L(first = row, second = u)
        }
        float spn[4];
        bool rok[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            rok[r] = qt * 16 + g4 + r < p.T;
            spn[r] = rok[r] ? cur.spn[r] : 0.f;
        }
#pragma unroll
        for (int jj = 0; jj < NJ; ++jj) {
            const bool hi_mark = (j0 + jj * 16) / dh > e_lo;      // uniform
            f32x4 a = zero4;
#pragma unroll
            for (int ub = 0; ub < DT; ++ub) a = mma16(hA[ub], frag_ld<T>(Wc + (jj * 16 + l15) * LDW + ub * 16 + g4), a);
            const float ws = fc[jj * 16 + l15], bs = fc[NC + jj * 16 + l15], wv = fc[2 * NC + jj * 16 + l15];
            f32x4 du;
            float sdb = 0.f, sdws = 0.f, sdw = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float dz = rok[r] ? (hi_mark ? cur.dz1[r] : cur.dz0[r]) : 0.f;
                const float z = sigmoid_pre(fmaf(spn[r], ws, a[r]) + bs);
                const float t2 = dz * z;
                du[r] = t2 * wv * (1.0f - z);
                sdb += du[r]; sdws += du[r] * spn[r]; sdw += t2;
            }
            adb[jj] += sdb; adws[jj] += sdws; adw[jj] += sdw;
            const Frag4<T> duf = frag_from_acc<T>(du);   // A operand: A[m = j][k = row]
#pragma unroll
            for (int ub = 0; ub < DT; ++ub) dW[jj][ub] = mma16(duf, hB[ub], dW[jj][ub]);
        }
        cur = nxt;
    }
    // block reduction in LDS, waves in turn (fixed order)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int jj = 0; jj < NJ; ++jj) {
                const float sb = group_sum4(adb[jj]), sws = group_sum4(adws[jj]), sw = group_sum4(adw[jj]);
                if (lane < 16) {
                    accs[dh * NC + jj * 16 + l15] += sws;          // dW1[dh][j] (interval row)
                    accs[(dh + 1) * NC + jj * 16 + l15] += sb;     // db1[j]
                    accs[(dh + 2) * NC + jj * 16 + l15] += sw;     // dw.flatten()[j]
                }
#pragma unroll
                for (int ub = 0; ub < DT; ++ub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) accs[(ub * 16 + l15) * NC + jj * 16 + g4 + r] += dW[jj][ub][r];   // dW1[u][j]
            }
        }
        __syncthreads();
    }
    float* dst = p.wpart + (long)blockIdx.x * NPARX;
    for (int i = threadIdx.x; i < (dh + 3) * NC; i += blockDim.x)
        if ((i % NC) < nval) dst[(long)(i / NC) * JE + j0 + (i % NC)] = accs[i];
    if (blockIdx.y == 0 && threadIdx.x < EP) {   // dscaling: fold sweep 1's per-(b, head) partials into the same partial row
        const long per = (p.njobs + gridDim.x - 1) / gridDim.x;
        const long a0 = blockIdx.x * per, a1 = min(p.njobs, a0 + per);
        float a = 0.f;
        for (long j = a0; j < a1; ++j) a += p.dsc_part[j * EP + threadIdx.x];
        dst[NPAR + threadIdx.x] = a;
    }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
inline int row_blocks(long ntile, int tiles_per_wg) {
    return (int)std::max<long>(1, std::min<long>((ntile + tiles_per_wg - 1) / tiles_per_wg, 1024));
}

template <typename T, int DT>
int run_intensity_fwd(const FwdP& p, hipStream_t st) {
    constexpr int dh = 16 * DT;
    IntP ip{};
    ip.hin = p.hin_out; ip.spans = p.spans; ip.pack = p.pack; ip.R = (long)p.B * p.H * p.T; ip.B = p.B; ip.T = p.T; ip.E = p.E;
    ip.z_out = p.z_out; ip.lam = p.lam;
    const size_t smem = (size_t)MarkChunk<T, DT, false>::BYTES * (big_double_buffer<T, DT>() ? 2 : 1);
    auto k = intensity_fwd_big_kernel<T, DT>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    const long ntile = (long)p.B * p.H * ((p.T + 15) / 16);
    hipLaunchKernelGGL(k, dim3(row_blocks(ntile, 8)), dim3(256), smem, st, ip);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

template <typename T, int DT, int NT>
int fwd_big(FwdP p, hipStream_t st) {
    int rc = launch_fwd_e<T, DT, NT, 0, 1>(p, st);        // scores: H rows
    if (rc) return rc;
    rc = run_intensity_fwd<T, DT>(p, st);                 // z, lambda
    if (rc) return rc;
    return launch_fwd_e<T, DT, NT, 0, 2>(p, st);          // values
}

template <typename T, int DT, int NT>
int bwd_big(BwdP p, char* ws, float* dW1, float* db1, float* dw, float* dscaling, hipStream_t st) {
    constexpr int dh = 16 * DT, Tp = 16 * NT, LDT = Tp + 4;
    constexpr bool TR = sizeof(T) == 2;
    const WsLayout wl = ws_layout(p.B, p.T, p.C, p.H, p.E);
    p.dz_ws = reinterpret_cast<float*>(ws + wl.dz); p.dh_ws = reinterpret_cast<float*>(ws + wl.dh);
    p.rowdot_ws = reinterpret_cast<float*>(ws + wl.rowdot);
    p.dsc_part = reinterpret_cast<float*>(ws + wl.dsc); p.wpart = reinterpret_cast<float*>(ws + wl.wpart);
    const long jobs = (long)p.B * p.H;
    edgl_prof_begin(EDGL_KERNEL_BIMAU_BWD_ALL, st);
    {   // X: sweep 1
        const size_t wave_bytes = (2 * (size_t)Tp * dh + (size_t)Tp * EP + (TR ? 0 : (size_t)EP * LDT)) * sizeof(T) + (size_t)Tp * sizeof(float);
        int waves = 4;
        while (waves > 1 && waves * wave_bytes > 64 * 1024) waves >>= 1;
        const size_t smem = waves * wave_bytes;
        EDGL_REQUIRE(smem <= 160 * 1024, EDGL_ERR_SHAPE, "edgl_bimau_bwd: sweep 1 needs %zu B of LDS (dh=%d T=%d)", smem, dh, p.T);
        p.waves = waves;
        auto kern = bimau_bwd_sweep1_kernel<T, DT, NT, 0, false>;
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kern, dim3((unsigned)((jobs + waves - 1) / waves)), dim3(64 * waves), smem, st, p);
        EDGL_LAUNCH_CHECK();
    }
    const int JE = dh * p.E, NPAR = (dh + 3) * JE, NPARX = NPAR + EP;
    const int RS = big_row_splits(dh, p.E);
    {   // Y: dH rows, then weight-gradient partials
        IntP ip{};
        ip.hin = p.hin; ip.spans = p.spans; ip.pack = p.pack; ip.dz = p.dz_ws; ip.R = (long)p.B * p.H * p.T; ip.B = p.B; ip.T = p.T;
        ip.E = p.E; ip.dh_out = p.dh_ws; ip.wpart = p.wpart; ip.dsc_part = p.dsc_part; ip.njobs = jobs;
        const long ntile = jobs * ((p.T + 15) / 16);
        const size_t smem_r = (size_t)MarkChunk<T, DT, true>::BYTES * (big_double_buffer<T, DT>() ? 2 : 1);
        auto kr = intensity_bwd_rows_big_kernel<T, DT>;
        hipFuncSetAttribute((const void*)kr, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_r);
        hipLaunchKernelGGL(kr, dim3(row_blocks(ntile, 8)), dim3(256), smem_r, st, ip);
        EDGL_LAUNCH_CHECK();
        constexpr int NC = WGroup<DT>::NJ * 16;
        const size_t smem_w = (size_t)NC * (dh + 4) * sizeof(T) + (size_t)(3 * NC + (dh + 3) * NC) * sizeof(float);
        auto kw = intensity_bwd_weights_big_kernel<T, DT>;
        hipFuncSetAttribute((const void*)kw, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_w);
        hipLaunchKernelGGL(kw, dim3(RS, (JE + NC - 1) / NC), dim3(256), smem_w, st, ip);
        EDGL_LAUNCH_CHECK();
    }
    {   // Z: sweep 2
        const size_t wave_bytes = (3 * (size_t)Tp * dh + (size_t)Tp * EP + (TR ? 0 : (size_t)dh * LDT)) * sizeof(T) + (size_t)Tp * sizeof(float);
        int waves = 4;
        while (waves > 1 && waves * wave_bytes > 80 * 1024) waves >>= 1;
        const size_t smem = waves * wave_bytes;
        EDGL_REQUIRE(smem <= 160 * 1024, EDGL_ERR_SHAPE, "edgl_bimau_bwd: sweep 2 needs %zu B of LDS (dh=%d T=%d)", smem, dh, p.T);
        p.waves = waves;
        edgl_prof_begin(EDGL_KERNEL_BIMAU_BWD, st);
        if constexpr (DT * NT > 32) {
            // head dim 128 beyond four key tiles (T > 64): the dK / dT_ accumulator tiles of all 128 channels do not fit the
            // register file (378-521 spilled registers) — two launches, each keeping the tiles of one 64-channel slice
            auto k0 = bimau_bwd_sweep2_kernel<T, DT, NT, 0, 1, false, -1, false, 2, 0>;
            auto k1 = bimau_bwd_sweep2_kernel<T, DT, NT, 0, 1, false, -1, false, 2, 1>;
            hipFuncSetAttribute((const void*)k0, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipLaunchKernelGGL(k0, dim3((unsigned)((jobs + waves - 1) / waves)), dim3(64 * waves), smem, st, p);
            hipLaunchKernelGGL(k1, dim3((unsigned)((jobs + waves - 1) / waves)), dim3(64 * waves), smem, st, p);
        } else {
            auto kern = bimau_bwd_sweep2_kernel<T, DT, NT, 0, 1, false>;
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
            hipLaunchKernelGGL(kern, dim3((unsigned)((jobs + waves - 1) / waves)), dim3(64 * waves), smem, st, p);
        }
        edgl_prof_end(EDGL_KERNEL_BIMAU_BWD, st);
        edgl_prof_end(EDGL_KERNEL_BIMAU_BWD_ALL, st);
        EDGL_LAUNCH_CHECK();
    }
    if (db1 == dW1 + (dh + 1) * JE && dw == db1 + JE && dscaling == dw + JE)   // flat-arena layout: one reduction
        return edgl_reduce_rows(p.wpart, RS, NPAR + p.E, NPARX, dW1, 0, st);
    int rc = edgl_reduce_rows(p.wpart, RS, (dh + 1) * JE, NPARX, dW1, 0, st);
    if (rc) return rc;
    rc = edgl_reduce_rows(p.wpart + (dh + 1) * JE, RS, JE, NPARX, db1, 0, st);
    if (rc) return rc;
    rc = edgl_reduce_rows(p.wpart + (dh + 2) * JE, RS, JE, NPARX, dw, 0, st);
    if (rc) return rc;
    return edgl_reduce_rows(p.wpart + NPAR, RS, p.E, NPARX, dscaling, 0, st);
}

// Head dims 64 and 128: T <= 128 (8 key tiles) in bf16.  The per-(b, head) accumulators dK, dT_ of sweep 2 are DT * NT register
// tiles each: up to 32 of them fit (head dim 64 at 8 key tiles: 458 registers, no spills), head dim 128 beyond 4 key tiles runs
// sweep 2 as two channel slices (bwd_big).  f32 stages K / T_ / V in four-byte elements (+ transposed images): head dim 64 up to
// T = 112 and head dim 128 up to T = 64 as before — the wave-private LDS check of the launchers bounds it.
constexpr bool big_ok(size_t esize, int DT, int NT) { return esize == 2 ? NT <= 8 : (DT == 4 ? NT <= 7 : NT <= 4); }
template <typename T, int DT, int NT>
int fwd_big_or_error(const FwdP& p, hipStream_t st) {
    if constexpr (big_ok(sizeof(T), DT, NT)) return fwd_big<T, DT, NT>(p, st);
    edgl_set_error("edgl_bimau_fwd: head dim %d with T=%d in f32 not supported (f32: dh 64: T <= 112, dh 128: T <= 64; bf16: T <= 128)", 16 * DT, p.T);
    return EDGL_ERR_SHAPE;
}
template <typename T>
int fwd_dispatch(const FwdP& p, hipStream_t st) {
    const int dh = p.C / p.H, nt = (p.T + 15) / 16;
#define EDGL_BIG_CASES(DT_)                                                                                              \
    switch (nt) {                                                                                                        \
        case 1: return fwd_big_or_error<T, DT_, 1>(p, st); case 2: return fwd_big_or_error<T, DT_, 2>(p, st);            \
        case 3: return fwd_big_or_error<T, DT_, 3>(p, st); case 4: return fwd_big_or_error<T, DT_, 4>(p, st);            \
        case 5: return fwd_big_or_error<T, DT_, 5>(p, st); case 6: return fwd_big_or_error<T, DT_, 6>(p, st);            \
        case 7: return fwd_big_or_error<T, DT_, 7>(p, st); case 8: return fwd_big_or_error<T, DT_, 8>(p, st);            \
    }
    if (dh == 64) { EDGL_BIG_CASES(4) } else if (dh == 128) { EDGL_BIG_CASES(8) }
#undef EDGL_BIG_CASES
    edgl_set_error("edgl_bimau_fwd: head dim %d with T=%d not supported (head dims 64 / 128: T <= 128)", dh, p.T);
    return EDGL_ERR_SHAPE;
}
template <typename T, int DT, int NT>
int bwd_big_or_error(const BwdP& p, char* ws, float* dW1, float* db1, float* dw, float* dsc, hipStream_t st) {
    if constexpr (big_ok(sizeof(T), DT, NT)) return bwd_big<T, DT, NT>(p, ws, dW1, db1, dw, dsc, st);
    edgl_set_error("edgl_bimau_bwd: head dim %d with T=%d in f32 not supported (f32: dh 64: T <= 112, dh 128: T <= 64; bf16: T <= 128)", 16 * DT, p.T);
    return EDGL_ERR_SHAPE;
}
template <typename T>
int bwd_dispatch(const BwdP& p, char* ws, float* dW1, float* db1, float* dw, float* dsc, hipStream_t st) {
    const int dh = p.C / p.H, nt = (p.T + 15) / 16;
#define EDGL_BIG_CASES(DT_)                                                                                                            \
    switch (nt) {                                                                                                                      \
        case 1: return bwd_big_or_error<T, DT_, 1>(p, ws, dW1, db1, dw, dsc, st); case 2: return bwd_big_or_error<T, DT_, 2>(p, ws, dW1, db1, dw, dsc, st); \
        case 3: return bwd_big_or_error<T, DT_, 3>(p, ws, dW1, db1, dw, dsc, st); case 4: return bwd_big_or_error<T, DT_, 4>(p, ws, dW1, db1, dw, dsc, st); \
        case 5: return bwd_big_or_error<T, DT_, 5>(p, ws, dW1, db1, dw, dsc, st); case 6: return bwd_big_or_error<T, DT_, 6>(p, ws, dW1, db1, dw, dsc, st); \
        case 7: return bwd_big_or_error<T, DT_, 7>(p, ws, dW1, db1, dw, dsc, st); case 8: return bwd_big_or_error<T, DT_, 8>(p, ws, dW1, db1, dw, dsc, st); \
    }
    if (dh == 64) { EDGL_BIG_CASES(4) } else if (dh == 128) { EDGL_BIG_CASES(8) }
#undef EDGL_BIG_CASES
    edgl_set_error("edgl_bimau_bwd: head dim %d with T=%d not supported (head dims 64 / 128: T <= 128)", dh, p.T);
    return EDGL_ERR_SHAPE;
}

}  // namespace

namespace bimau {
int big_fwd(const FwdP& p, int dtype, hipStream_t st) {
    if (!p.hin_out || !p.z_out) {
        edgl_set_error("edgl_bimau_fwd: head dims >= 64 run as three launches and need `saved` (edgl_bimau_saved_bytes) as scratch, also for inference");
        return EDGL_ERR_WORKSPACE;
    }
    return dtype == EDGL_F32 ? fwd_dispatch<float>(p, st) : fwd_dispatch<bf16>(p, st);
}
int big_bwd(const BwdP& p, char* ws, float* dW1, float* db1, float* dw, float* dsc, int dtype, hipStream_t st) {
    return dtype == EDGL_F32 ? bwd_dispatch<float>(p, ws, dW1, db1, dw, dsc, st) : bwd_dispatch<bf16>(p, ws, dW1, db1, dw, dsc, st);
}
}  // namespace bimau

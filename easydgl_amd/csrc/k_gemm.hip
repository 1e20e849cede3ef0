// K2/K4: MFMA tile GEMM for the dense layers of the hot path (tf.layers.dense in temporal.py:409,
// EasyDGL.py:113,120,125,138) and their backward products.
//   C[M,N] = epilogue( sum_k A(m,k) B(k,n) )
// 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each 64x64 = 4x4 MFMA 16x16 tiles),
// operands staged through LDS k-contiguous ([row][k], 16-byte padded rows) so every fragment is one
// ds_read_b128; operands whose contraction index is the slow dimension in HBM are transposed in
// registers (VECxVEC blocks) on the way in, so HBM reads stay coalesced.
// float -> v_mfma_f32_16x16x4_f32 (exact f32), bf16 -> v_mfma_f32_16x16x32_bf16, f32 accumulate.
#include "gemm_tile.h"

namespace {
using namespace tile;

struct GemmP {
    const void* A; const void* B; void* C;
    int M, N, K, lda, ldb, ldc;
    const float* bias; void* aux;
    int flags;
    int kper;          // K range per split (multiple of BK)
    float* partial;    // split-K partials [splits][M][N] or nullptr
    int vec_ok;        // operands 16-byte aligned with ld % VEC == 0
};

template <typename T>
__device__ __forceinline__ void store4(T* dst, const float x[4]) {
    Frag4<T> f;
#pragma unroll
    for (int r = 0; r < 4; ++r) f.v[r] = from_f32<T>(x[r]);
    if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(&f);
    else *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(&f);
}

template <typename T>
__device__ __forceinline__ void epilogue_store4(const GemmP& p, float v[4], int m, int n) {
    // 4 consecutive n for one m
    if (m >= p.M) return;
    const long idx0 = (long)m * p.ldc + n;
    if (n + 3 < p.N && (p.ldc & 3) == 0) {
        // vector path: one 8/16-byte store per lane instead of four scalar stores
        float x[4] = {v[0], v[1], v[2], v[3]};
        if (p.flags & EDGL_EPI_BIAS) {
            const float4 b = *reinterpret_cast<const float4*>(p.bias + n);
            x[0] += b.x; x[1] += b.y; x[2] += b.z; x[3] += b.w;
        }
        if (p.flags & EDGL_EPI_SAVE_PRE) store4<T>(reinterpret_cast<T*>(p.aux) + idx0, x);
        if (p.flags & EDGL_EPI_GELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = gelu_t<T>(x[r]);
        }
        if (p.flags & EDGL_EPI_RELU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] = fmaxf(x[r], 0.f);
        }
        if (p.flags & EDGL_EPI_MUL_DGELU) {
            const Frag4<T> a = frag_ld<T>(reinterpret_cast<const T*>(p.aux) + idx0);
#pragma unroll
            for (int r = 0; r < 4; ++r) x[r] *= dgelu_t<T>(to_f32(a.v[r]));
        }
        if (p.flags & EDGL_EPI_OUT_F32) {
            float* c = reinterpret_cast<float*>(p.C) + idx0;
            if (p.flags & EDGL_EPI_ACCUM) {
                const float4 o = *reinterpret_cast<const float4*>(c);
                x[0] += o.x; x[1] += o.y; x[2] += o.z; x[3] += o.w;
            }
            *reinterpret_cast<float4*>(c) = make_float4(x[0], x[1], x[2], x[3]);
        } else {
            T* c = reinterpret_cast<T*>(p.C) + idx0;
            if (p.flags & EDGL_EPI_ACCUM) {
                const Frag4<T> o = frag_ld<T>(c);
#pragma unroll
                for (int r = 0; r < 4; ++r) x[r] += to_f32(o.v[r]);
            }
            store4<T>(c, x);
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (n + r >= p.N) continue;
        float x = v[r];
        if (p.flags & EDGL_EPI_BIAS) x += p.bias[n + r];
        const long idx = idx0 + r;
        if (p.flags & EDGL_EPI_SAVE_PRE) reinterpret_cast<T*>(p.aux)[idx] = from_f32<T>(x);
        if (p.flags & EDGL_EPI_GELU) x = gelu_t<T>(x);
        if (p.flags & EDGL_EPI_RELU) x = fmaxf(x, 0.f);
        if (p.flags & EDGL_EPI_MUL_DGELU) x *= dgelu_t<T>(to_f32(reinterpret_cast<const T*>(p.aux)[idx]));
        if (p.flags & EDGL_EPI_OUT_F32) {
            float* c = reinterpret_cast<float*>(p.C);
            c[idx] = (p.flags & EDGL_EPI_ACCUM) ? c[idx] + x : x;
        } else {
            T* c = reinterpret_cast<T*>(p.C);
            c[idx] = from_f32<T>((p.flags & EDGL_EPI_ACCUM) ? to_f32(c[idx]) + x : x);
        }
    }
}

template <typename T, bool A_KC, bool B_KC>
__global__ __launch_bounds__(NT) void gemm_kernel(GemmP p) {
    constexpr int VEC = ElemTraits<T>::VEC, KB = ElemTraits<T>::KB, BK = 2 * KB, LDK = BK + VEC;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    T* As = reinterpret_cast<T*>(smem_raw);
    T* Bs = As + BM * LDK;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int kbeg = blockIdx.z * p.kper;
    const int kend = min(p.K, kbeg + p.kper);
    const T* A = reinterpret_cast<const T*>(p.A);
    const T* Bp = reinterpret_cast<const T*>(p.B);
    const bool vok = p.vec_ok != 0;

    f32x4 acc[4][4];  // acc[j][i]: rows <-> n (j), cols <-> m (i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[j][i] = f32x4{0.f, 0.f, 0.f, 0.f};

    Stager<T, A_KC> sa;
    Stager<T, B_KC> sb;
    const int nk = (kend - kbeg + BK - 1) / BK;
    if (nk > 0) {
        sa.load(A, p.lda, m0, p.M, kbeg, kend, vok);
        sb.load(Bp, p.ldb, n0, p.N, kbeg, kend, vok);
        sa.store(As);
        sb.store(Bs);
    }
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = (kt + 1 < nk);
        if (more) {
            sa.load(A, p.lda, m0, p.M, kbeg + (kt + 1) * BK, kend, vok);
            sb.load(Bp, p.ldb, n0, p.N, kbeg + (kt + 1) * BK, kend, vok);
        }
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            Vec16<T> af[4], bf[4];
            const int koff = kb * KB + (lane >> 4) * VEC;
#pragma unroll
            for (int i = 0; i < 4; ++i) af[i] = ld16<T>(As + (wm * 64 + i * 16 + (lane & 15)) * LDK + koff);
#pragma unroll
            for (int j = 0; j < 4; ++j) bf[j] = ld16<T>(Bs + (wn * 64 + j * 16 + (lane & 15)) * LDK + koff);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[j][i] = mma_kblock(bf[j], af[i], acc[j][i]);
        }
        __syncthreads();
        if (more) {
            sa.store(As);
            sb.store(Bs);
        }
        __syncthreads();
    }

    // epilogue: lane holds, for tile (j,i): n = n0+wn*64+j*16+(lane>>4)*4 + r, m = m0+wm*64+i*16+(lane&15)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + (lane & 15);
            const int n = n0 + wn * 64 + j * 16 + (lane >> 4) * 4;
            float v[4] = {acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]};
            if (p.partial) {
                if (m < p.M) {
                    float* dst = p.partial + ((long)blockIdx.z * p.M + m) * p.N + n;
                    if (n + 3 < p.N && (p.N & 3) == 0) {
                        *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (n + r < p.N) dst[r] = v[r];
                    }
                }
            } else {
                epilogue_store4<T>(p, v, m, n);
            }
        }
}

template <typename T>
__global__ void splitk_reduce_kernel(GemmP p, int splits) {
    const long total4 = (long)p.M * ((p.N + 3) / 4);
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total4; q += (long)gridDim.x * blockDim.x) {
        const int m = (int)(q / ((p.N + 3) / 4)), n = (int)(q % ((p.N + 3) / 4)) * 4;
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < splits; ++s) {
            const float* src = p.partial + ((long)s * p.M + m) * p.N + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (n + r < p.N) v[r] += src[r];
        }
        epilogue_store4<T>(p, v, m, n);
    }
}

template <typename T>
int launch_gemm(GemmP p, int a_kc, int b_kc, int splitk, hipStream_t st) {
    constexpr int VEC = ElemTraits<T>::VEC, BK = 2 * ElemTraits<T>::KB, LDK = BK + VEC;
    const size_t smem = (size_t)(BM + BN) * LDK * sizeof(T);
    if (splitk < 1) splitk = 1;
    int kper = ((p.K + splitk - 1) / splitk + BK - 1) / BK * BK;
    if (kper < BK) kper = BK;
    splitk = (p.K + kper - 1) / kper;
    if (splitk < 1) splitk = 1;
    p.kper = kper;
    if (splitk == 1) p.partial = nullptr;
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM, splitk);
    if (a_kc && b_kc) hipLaunchKernelGGL((gemm_kernel<T, true, true>), grid, dim3(NT), smem, st, p);
    else if (a_kc && !b_kc) hipLaunchKernelGGL((gemm_kernel<T, true, false>), grid, dim3(NT), smem, st, p);
    else if (!a_kc && b_kc) hipLaunchKernelGGL((gemm_kernel<T, false, true>), grid, dim3(NT), smem, st, p);
    else hipLaunchKernelGGL((gemm_kernel<T, false, false>), grid, dim3(NT), smem, st, p);
    EDGL_LAUNCH_CHECK();
    if (splitk > 1) {
        const long total4 = (long)p.M * ((p.N + 3) / 4);
        int blocks = (int)std::min<long>((total4 + 255) / 256, 4096);
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, p, splitk);
        EDGL_LAUNCH_CHECK();
    }
    return EDGL_OK;
}

// ---- column sums (bias gradients) -------------------------------------------------------------
// block = (N/VEC column vectors) x (256/(N/VEC) row lanes); 16-byte loads; per-block partial [N]
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* X, int M, int N, int ld, float* part, int rows_per_block) {
    constexpr int VEC = ElemTraits<T>::VEC;
    extern __shared__ float sm[];  // [rows_par][N]
    const int cpv = N / VEC, rows_par = 256 / cpv;
    const int cv = threadIdx.x % cpv, tr = threadIdx.x / cpv;
    const bool on = threadIdx.x < rows_par * cpv;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float acc[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
    if (on)
        for (int r = r0 + tr; r < r1; r += rows_par) {
            const Vec16<T> v = ld16<T>(X + (long)r * ld + cv * VEC);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc[j] += to_f32(v.v[j]);
        }
    if (on) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) sm[tr * N + cv * VEC + j] = acc[j];
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
        float s = 0.f;
        for (int t = 0; t < rows_par; ++t) s += sm[t * N + n];
        part[(long)blockIdx.x * N + n] = s;
    }
}
// generic fallback (N not a multiple of the vector width)
template <typename T>
__global__ void colsum_partial_scalar_kernel(const T* X, int M, int N, int ld, float* part, int rows_per_block) {
    const int n = blockIdx.y * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    float s = 0.f;
    for (int r = r0; r < r1; ++r) s += to_f32(X[(long)r * ld + n]);
    part[(long)blockIdx.x * N + n] = s;
}

}  // namespace

extern "C" int edgl_gemm(const void* A, const void* Bm, void* Cm, int M, int N, int K, int lda, int ldb,
                         int ldc, int a_kc, int b_kc, const float* bias, void* aux, int epi_flags, int splitk,
                         float* workspace, int dtype, void* stream) {
    EDGL_REQUIRE(A && Bm && Cm, EDGL_ERR_NULL, "edgl_gemm: null operand");
    EDGL_REQUIRE(M > 0 && N > 0 && K > 0, EDGL_ERR_SHAPE, "edgl_gemm: bad shape M=%d N=%d K=%d", M, N, K);
    EDGL_REQUIRE(!(epi_flags & EDGL_EPI_BIAS) || bias, EDGL_ERR_NULL, "edgl_gemm: bias flag without bias");
    EDGL_REQUIRE(!(epi_flags & (EDGL_EPI_SAVE_PRE | EDGL_EPI_MUL_DGELU)) || aux, EDGL_ERR_NULL,
                 "edgl_gemm: aux flag without aux");
    EDGL_REQUIRE(splitk <= 1 || workspace, EDGL_ERR_WORKSPACE, "edgl_gemm: split-K needs a workspace");
    GemmP p;
    p.A = A; p.B = Bm; p.C = Cm; p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.bias = bias; p.aux = aux; p.flags = epi_flags; p.kper = K; p.partial = workspace;
    const int vec = (dtype == EDGL_BF16) ? 8 : 4;
    p.vec_ok = (lda % vec == 0) && (ldb % vec == 0) && (((uintptr_t)A & 15) == 0) && (((uintptr_t)Bm & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_BF16 && a_kc) {   // dense forward / dX shapes: register-resident A strips (k_gemm2.hip)
        const int r = edgl_gemm2_try_strip(A, Bm, Cm, M, N, K, lda, ldb, ldc, b_kc, bias, aux, epi_flags, st);
        if (r != 0) return r < 0 ? r : EDGL_OK;
    }
    if (dtype == EDGL_F32) return launch_gemm<float>(p, a_kc, b_kc, splitk, st);
    if (dtype == EDGL_BF16) return launch_gemm<bf16>(p, a_kc, b_kc, splitk, st);
    edgl_set_error("edgl_gemm: bad dtype %d", dtype);
    return EDGL_ERR_DTYPE;
}

template <typename T>
int run_colsum(const T* X, int M, int N, int ld, float* out, int accumulate, float* workspace, hipStream_t st) {
    constexpr int VEC = ElemTraits<T>::VEC;
    int nparts = std::min(256, std::max(1, M / 64));
    const int rpb = (M + nparts - 1) / nparts;
    nparts = (M + rpb - 1) / rpb;
    const bool vec = (N % VEC == 0) && (ld % VEC == 0) && (N / VEC <= 256) && (((uintptr_t)X & 15) == 0);
    if (vec) {
        const int rows_par = 256 / (N / VEC);
        hipLaunchKernelGGL((colsum_partial_kernel<T>), dim3(nparts), dim3(256), (size_t)rows_par * N * sizeof(float), st, X, M, N, ld, workspace, rpb);
    } else {
        hipLaunchKernelGGL((colsum_partial_scalar_kernel<T>), dim3(nparts, (N + 255) / 256), dim3(256), 0, st, X, M, N, ld, workspace, rpb);
    }
    EDGL_LAUNCH_CHECK();
    return edgl_reduce_rows(workspace, nparts, N, N, out, accumulate, st);
}

extern "C" int edgl_colsum(const void* X, int M, int N, int ld, float* out, int accumulate, float* workspace,
                           int x_f32, int dtype, void* stream) {
    EDGL_REQUIRE(X && out && workspace, EDGL_ERR_NULL, "edgl_colsum: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (x_f32 || dtype == EDGL_F32) return run_colsum<float>((const float*)X, M, N, ld, out, accumulate, workspace, st);
    return run_colsum<bf16>((const bf16*)X, M, N, ld, out, accumulate, workspace, st);
}

// dW[Kf,N] (+)= X[R,Kf]^T . dY[R,N] and (optionally) dbias[N] (+)= colsum(dY): the weight/bias gradients of a
// dense layer (tf.layers.dense backward).  bf16: transposing-read TN kernel; f32: generic kernel + colsum.
extern "C" long edgl_gemm_dw_workspace(int R, int Kf, int N, int dtype) {
    const long tn = edgl_gemm2_tn_workspace(R, Kf, N);
    const long generic = 64L * Kf * N + 256L * N;
    return dtype == EDGL_BF16 ? std::max(tn, generic) : generic;
}

int edgl_gemm2_tn_defer(int on, hipStream_t st);
extern "C" int edgl_gemm_dw_defer(int on, void* stream) { return edgl_gemm2_tn_defer(on, (hipStream_t)stream); }

extern "C" int edgl_gemm_dw(const void* X, const void* dY, float* dW, float* dbias, int R, int Kf, int N, int ldx, int ldy,
                            int accumulate, float* workspace, int dtype, void* stream) {
    EDGL_REQUIRE(X && dY && dW && workspace, EDGL_ERR_NULL, "edgl_gemm_dw: null pointer");
    EDGL_REQUIRE(R > 0 && Kf > 0 && N > 0, EDGL_ERR_SHAPE, "edgl_gemm_dw: bad shape R=%d Kf=%d N=%d", R, Kf, N);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_BF16) {
        const int r = edgl_gemm2_try_tn(X, dY, dW, R, Kf, N, ldx, ldy, N, dbias, accumulate, workspace, st);
        if (r != 0) return r < 0 ? r : EDGL_OK;
    }
    const int tiles = ((Kf + 127) / 128) * ((N + 127) / 128);
    const int splitk = std::max(1, std::min(64, std::min(R / 256, 320 / tiles)));
    int rc = edgl_gemm(X, dY, dW, Kf, N, R, ldx, ldy, N, 0, 0, nullptr, nullptr,
                       EDGL_EPI_OUT_F32 | (accumulate ? EDGL_EPI_ACCUM : 0), splitk, workspace, dtype, stream);
    if (rc) return rc;
    if (dbias) rc = edgl_colsum(dY, R, N, ldy, dbias, accumulate, workspace, 0, dtype, stream);
    return rc;
}

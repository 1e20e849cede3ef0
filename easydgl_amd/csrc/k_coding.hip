// Operator-level entry points of src/module/coding.py — the stand-alone forms of the classes whose arithmetic the
// model kernels fuse (k_encode.hip, k_data.hip, k_tattn.hip):
//   Embedding.__call__            coding.py:60-64     edgl_embedding_fwd / edgl_embedding_bwd
//   PositionCoding.code           coding.py:76-79     (the same gather, ids = 0..T-1 per row — built by the host wrapper)
//   TimeIntervalCoding.code       coding.py:93-94     (the same gather on integer intervals)
//   TimeSinusoidCoding.code       coding.py:137-149   edgl_time_sinusoid
//   TimeFunctionCoding.code       coding.py:113-122   edgl_time_function_fwd / edgl_time_function_bwd
// All are HBM-bound row streams: one thread owns 4 consecutive channels of one output row (8/16-byte stores).
#include "edgl_common.h"

namespace {

template <typename T>
__device__ __forceinline__ void st4(T* dst, const float (&v)[4]) {
    Frag4<T> o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o.v[j] = from_f32<T>(v[j]);
    if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<uint4*>(&o);
    else *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<uint2*>(&o);
}

// out[r] = scale * table[ids[r]]; zero_pad: row 0 acts as a zero constant (coding.py:56-57).  An index outside [0, rows)
// reads zeros — what tf.nn.embedding_lookup does on the GPU (TiSASRec relies on it: bucket == timelen, TiSASREC.py:59).
template <typename T>
__global__ __launch_bounds__(256) void embedding_fwd_kernel(const int64_t* ids, long n, const T* table, int rows, int C,
                                                            int zero_pad, float scale, T* out) {
    const int cpr = C >> 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long r = gid / cpr;
    if (r >= n) return;
    const int c0 = (int)(gid % cpr) * 4;
    const int64_t id = ids[r];
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (id >= (zero_pad ? 1 : 0) && id < rows) {
        const Frag4<T> f = frag_ld<T>(table + id * C + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = to_f32(f.v[j]) * scale;
    }
    st4<T>(out + r * C + c0, v);
}

// d_table[ids[r]] += scale * d_out[r] (f32 atomics into the zero-filled table gradient; row 0 skipped when zero_pad)
template <typename T>
__global__ __launch_bounds__(256) void embedding_bwd_kernel(const int64_t* ids, long n, const T* d_out, int rows, int C,
                                                            int zero_pad, float scale, float* d_table) {
    const int cpr = C >> 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long r = gid / cpr;
    if (r >= n) return;
    const int c0 = (int)(gid % cpr) * 4;
    const int64_t id = ids[r];
    if (id < (zero_pad ? 1 : 0) || id >= rows) return;
    const Frag4<T> g = frag_ld<T>(d_out + r * C + c0);
#pragma unroll
    for (int j = 0; j < 4; ++j) atomicAdd(d_table + id * C + c0 + j, scale * to_f32(g.v[j]));
}

// code[r, 2j] = sin(x[r] / scale[j]), code[r, 2j+1] = cos(same) (coding.py:141-148); thread = 2 (sin, cos) pairs.
// float: libm sincosf of the float32 quotient (parity mode); bf16: double-precision reduction to revolutions + the
// hardware sin/cos, as in encode_fwd_kernel.
template <typename T>
__global__ __launch_bounds__(256) void time_sinusoid_kernel(const float* x, long n, const float* tscale, int C, T* out) {
    const int cpr = C >> 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long r = gid / cpr;
    if (r >= n) return;
    const int c0 = (int)(gid % cpr) * 4, j0 = c0 >> 1;
    const float xv = x[r];
    float v[4];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float arg = xv / tscale[j0 + q];
        float sn, cs;
        if constexpr (sizeof(T) == 4) {
            sincosf(arg, &sn, &cs);
        } else {
            const double rev = (double)arg * 0.15915494309189533577;
            const float fr = (float)(rev - __builtin_rint(rev));
            sn = __builtin_amdgcn_sinf(fr);
            cs = __builtin_amdgcn_cosf(fr);
        }
        v[2 * q] = sn; v[2 * q + 1] = cs;
    }
    st4<T>(out + r * C + c0, v);
}

// code[r, c] = cos(x[r] * freq[c] + phase[c]) (coding.py:118-121)
template <typename T>
__global__ __launch_bounds__(256) void time_function_fwd_kernel(const float* x, long n, const float* freq, const float* phase,
                                                                int C, T* out) {
    const int cpr = C >> 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long r = gid / cpr;
    if (r >= n) return;
    const int c0 = (int)(gid % cpr) * 4;
    const float xv = x[r];
    const float4 f = *reinterpret_cast<const float4*>(freq + c0), ph = *reinterpret_cast<const float4*>(phase + c0);
    const float v[4] = {cosf(fmaf(xv, f.x, ph.x)), cosf(fmaf(xv, f.y, ph.y)), cosf(fmaf(xv, f.z, ph.z)), cosf(fmaf(xv, f.w, ph.w))};
    st4<T>(out + r * C + c0, v);
}

// d_freq[c] = sum_r -sin(x f + phi) x dy, d_phase[c] = sum_r -sin(x f + phi) dy: per-block partials [TFB][2C] in a fixed
// order, then edgl_reduce_rows.  A block's threads split (channel quad, row lane); rows strided by the grid.
constexpr int TFB = 256;
template <typename T>
__global__ __launch_bounds__(256) void time_function_bwd_kernel(const float* x, long n, const float* freq, const float* phase,
                                                                int C, const T* d_out, float* part) {
    extern __shared__ float sm[];   // [rows_par][2][C]
    const int cpr = C >> 2, rows_par = 256 / cpr;
    const int cv = threadIdx.x % cpr, rl = threadIdx.x / cpr, c0 = cv * 4;
    float af[4] = {0.f, 0.f, 0.f, 0.f}, ap[4] = {0.f, 0.f, 0.f, 0.f};
    if (rl < rows_par) {
        const float4 f = *reinterpret_cast<const float4*>(freq + c0), ph = *reinterpret_cast<const float4*>(phase + c0);
        const float fv[4] = {f.x, f.y, f.z, f.w}, pv[4] = {ph.x, ph.y, ph.z, ph.w};
        for (long r = (long)blockIdx.x * rows_par + rl; r < n; r += (long)gridDim.x * rows_par) {
            const float xv = x[r];
            const Frag4<T> g = frag_ld<T>(d_out + r * C + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = -sinf(fmaf(xv, fv[j], pv[j])) * to_f32(g.v[j]);
                ap[j] += d;
                af[j] = fmaf(d, xv, af[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) { sm[(rl * 2 + 0) * C + c0 + j] = af[j]; sm[(rl * 2 + 1) * C + c0 + j] = ap[j]; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
        float a = 0.f;
        for (int q = 0; q < rows_par; ++q) a += sm[q * 2 * C + i];
        part[(long)blockIdx.x * 2 * C + i] = a;
    }
}

inline bool coding_shape_ok(long n, int C) { return n > 0 && C > 0 && C % 4 == 0; }

}  // namespace

extern "C" int edgl_embedding_fwd(const int64_t* ids, long n, const void* table, int rows, int C, int zero_pad, float scale,
                                  void* out, int dtype, void* stream) {
    EDGL_REQUIRE(ids && table && out, EDGL_ERR_NULL, "edgl_embedding_fwd: null pointer");
    EDGL_REQUIRE(coding_shape_ok(n, C) && rows > 0, EDGL_ERR_SHAPE, "edgl_embedding_fwd: bad shape n=%ld rows=%d C=%d", n, rows, C);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_embedding_fwd: bad dtype %d", dtype);
    const dim3 grid((unsigned)((n * (C / 4) + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_F32) hipLaunchKernelGGL((embedding_fwd_kernel<float>), grid, dim3(256), 0, st, ids, n, (const float*)table, rows, C, zero_pad, scale, (float*)out);
    else hipLaunchKernelGGL((embedding_fwd_kernel<bf16>), grid, dim3(256), 0, st, ids, n, (const bf16*)table, rows, C, zero_pad, scale, (bf16*)out);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_embedding_bwd(const int64_t* ids, long n, const void* d_out, int rows, int C, int zero_pad, float scale,
                                  float* d_table, int dtype, void* stream) {
    EDGL_REQUIRE(ids && d_out && d_table, EDGL_ERR_NULL, "edgl_embedding_bwd: null pointer");
    EDGL_REQUIRE(coding_shape_ok(n, C) && rows > 0, EDGL_ERR_SHAPE, "edgl_embedding_bwd: bad shape n=%ld rows=%d C=%d", n, rows, C);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_embedding_bwd: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(d_table, 0, (size_t)rows * C * sizeof(float), st) != hipSuccess) {
        edgl_set_error("edgl_embedding_bwd: memset failed");
        return EDGL_ERR_LAUNCH;
    }
    const dim3 grid((unsigned)((n * (C / 4) + 255) / 256));
    if (dtype == EDGL_F32) hipLaunchKernelGGL((embedding_bwd_kernel<float>), grid, dim3(256), 0, st, ids, n, (const float*)d_out, rows, C, zero_pad, scale, d_table);
    else hipLaunchKernelGGL((embedding_bwd_kernel<bf16>), grid, dim3(256), 0, st, ids, n, (const bf16*)d_out, rows, C, zero_pad, scale, d_table);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_time_sinusoid(const float* x, long n, const float* tscale, int C, void* out, int dtype, void* stream) {
    EDGL_REQUIRE(x && tscale && out, EDGL_ERR_NULL, "edgl_time_sinusoid: null pointer");
    EDGL_REQUIRE(coding_shape_ok(n, C), EDGL_ERR_SHAPE, "edgl_time_sinusoid: bad shape n=%ld C=%d", n, C);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_time_sinusoid: bad dtype %d", dtype);
    const dim3 grid((unsigned)((n * (C / 4) + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_F32) hipLaunchKernelGGL((time_sinusoid_kernel<float>), grid, dim3(256), 0, st, x, n, tscale, C, (float*)out);
    else hipLaunchKernelGGL((time_sinusoid_kernel<bf16>), grid, dim3(256), 0, st, x, n, tscale, C, (bf16*)out);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_time_function_fwd(const float* x, long n, const float* freq, const float* phase, int C, void* out,
                                      int dtype, void* stream) {
    EDGL_REQUIRE(x && freq && phase && out, EDGL_ERR_NULL, "edgl_time_function_fwd: null pointer");
    EDGL_REQUIRE(coding_shape_ok(n, C), EDGL_ERR_SHAPE, "edgl_time_function_fwd: bad shape n=%ld C=%d", n, C);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_time_function_fwd: bad dtype %d", dtype);
    const dim3 grid((unsigned)((n * (C / 4) + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_F32) hipLaunchKernelGGL((time_function_fwd_kernel<float>), grid, dim3(256), 0, st, x, n, freq, phase, C, (float*)out);
    else hipLaunchKernelGGL((time_function_fwd_kernel<bf16>), grid, dim3(256), 0, st, x, n, freq, phase, C, (bf16*)out);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" long edgl_time_function_bwd_workspace(int C) { return (long)TFB * 2 * C; }

extern "C" int edgl_time_function_bwd(const float* x, long n, const float* freq, const float* phase, int C, const void* d_out,
                                      float* d_freq, float* d_phase, float* workspace, int dtype, void* stream) {
    EDGL_REQUIRE(x && freq && phase && d_out && d_freq && d_phase && workspace, EDGL_ERR_NULL, "edgl_time_function_bwd: null pointer");
    EDGL_REQUIRE(coding_shape_ok(n, C) && C / 4 <= 256, EDGL_ERR_SHAPE, "edgl_time_function_bwd: bad shape n=%ld C=%d", n, C);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_time_function_bwd: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const int rows_par = 256 / (C / 4);
    const size_t smem = (size_t)rows_par * 2 * C * sizeof(float);
    if (dtype == EDGL_F32) hipLaunchKernelGGL((time_function_bwd_kernel<float>), dim3(TFB), dim3(256), smem, st, x, n, freq, phase, C, (const float*)d_out, workspace);
    else hipLaunchKernelGGL((time_function_bwd_kernel<bf16>), dim3(TFB), dim3(256), smem, st, x, n, freq, phase, C, (const bf16*)d_out, workspace);
    EDGL_LAUNCH_CHECK();
    int rc = edgl_reduce_rows(workspace, TFB, C, 2L * C, d_freq, 0, st);
    if (rc) return rc;
    return edgl_reduce_rows(workspace + C, TFB, C, 2L * C, d_phase, 0, st);
}

// K4-LN: y = LayerNorm_joint(dropout(x) + resid) and its backward.
// The reference's layernorm (Base.py:12-67, begin_norm_axis=1) takes ONE mean/variance per sample
// over all T*C elements (tf.nn.moments over axes [1,2], population variance, eps 1e-12), with
// gamma/beta over the last axis.  One workgroup per sample; each thread owns a fixed 16-byte column
// vector and strides over rows, so per-channel dgamma/dbeta partials stay in registers.
// Two-pass moments (mean, then centred second moment) as tf.nn.moments does.
#include "edgl_common.h"

namespace {

constexpr int LN_THREADS = 512;
constexpr int LN_NR = 4;   // rows a thread can keep in registers between the passes

struct LnP {
    const void* x; const void* resid; int ld_res;
    const float* gamma; const float* beta;
    int B, T, C;
    float rate; const uint64_t* rng; uint32_t stream_id;
    const int64_t* gpos; int Mg;
    void* y; float* stats;
    // backward
    const void* dy; void* dsum; void* dx_drop; float* part;
    const int32_t* dy_rowmap;   // gathered mode: compact row of (b, j), or -1 (row carries no gradient)
    const void* act_pre;        // backward: x = gelu(act_pre); the emitted gradients are multiplied by gelu'(act_pre)
    // zero-padded channels (a head dim the attention kernels do not tile, run at the next supported one): channel c is real iff
    // c % dhp < dht; the moments are those of the real channels (pads hold exact zeros and are left out of the centred second
    // moment and of the divisor), the padded channels of the input gradient are zero.  dhp == 0: no padding.
    int dhp, dht;
};
// 1 for a real channel, 0 for a padded one
__device__ __forceinline__ float chan_on(const LnP& p, int c) { return (p.dhp == 0 || (c % p.dhp) < p.dht) ? 1.f : 0.f; }
__device__ __forceinline__ float ln_count(const LnP& p) {
    return (float)p.T * (float)(p.dhp == 0 ? p.C : (p.C / p.dhp) * p.dht);
}

template <typename T>
__device__ __forceinline__ void load_sum(const LnP& p, const DropKey& dk, int b, int t, int c0, float s[ElemTraits<T>::VEC]) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const long row = (long)b * p.T + t;
    Vec16<T> xv = ld16<T>(reinterpret_cast<const T*>(p.x) + row * p.C + c0);
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = drop_apply(dk, (uint64_t)(row * p.C + c0 + j), to_f32(xv.v[j]));
    if (p.resid) {
        Vec16<T> rv = ld16<T>(reinterpret_cast<const T*>(p.resid) + row * p.ld_res + c0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) s[j] += to_f32(rv.v[j]);
    }
}

template <typename T>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(LnP p) {
    constexpr int VEC = ElemTraits<T>::VEC;
    __shared__ float red[LN_THREADS / 64];
    const int b = blockIdx.x, cpv = p.C / VEC;
    const int rows_par = LN_THREADS / cpv, active = rows_par * cpv;
    const int tid = threadIdx.x;
    const bool on = tid < active;
    const int cv = tid % cpv, tr = tid / cpv, c0 = cv * VEC;
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const float n = ln_count(p);
    float cm[VEC];      // real / padded channel (all ones without padding)
#pragma unroll
    for (int j = 0; j < VEC; ++j) cm[j] = chan_on(p, c0 + j);

    // A thread owns rows tr, tr + rows_par, ...: when there are at most LN_NR of them (T <= LN_NR * rows_par, the usual
    // case) the dropped-out sums are read ONCE into registers and all three passes (mean, variance, output) run from
    // there; otherwise every pass re-reads its rows.
    const bool cached = p.T <= LN_NR * rows_par;
    float sc[LN_NR][VEC];
    if (cached) {
#pragma unroll
        for (int i = 0; i < LN_NR; ++i) {
            const int t = tr + i * rows_par;
            load_sum<T>(p, dk, b, min(t, p.T - 1), c0, sc[i]);   // clamped, unconditional: the loads overlap
            if (!(on && t < p.T)) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) sc[i][j] = 0.f;
            }
        }
    }
    float acc = 0.f;
    if (cached) {
#pragma unroll
        for (int i = 0; i < LN_NR; ++i)
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc += sc[i][j];
    } else if (on) {
        for (int t = tr; t < p.T; t += rows_par) {
            float s[VEC];
            load_sum<T>(p, dk, b, t, c0, s);
#pragma unroll
            for (int j = 0; j < VEC; ++j) acc += s[j];
        }
    }
    const float mean = block_sum(acc, red) / n;
    acc = 0.f;
    if (cached) {
#pragma unroll
        for (int i = 0; i < LN_NR; ++i)
            if (on && tr + i * rows_par < p.T) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) { const float d = sc[i][j] - mean; acc += cm[j] * (d * d); }
            }
    } else if (on) {
        for (int t = tr; t < p.T; t += rows_par) {
            float s[VEC];
            load_sum<T>(p, dk, b, t, c0, s);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const float d = s[j] - mean; acc += cm[j] * (d * d); }
        }
    }
    const float var = block_sum(acc, red) / n;
    const float rstd = rsqrtf(var + 1e-12f);
    if (tid == 0) { p.stats[2 * b] = mean; p.stats[2 * b + 1] = rstd; }
    if (!on) return;
    float g[VEC], be[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { g[j] = p.gamma[c0 + j]; be[j] = p.beta[c0 + j]; }
    T* y = reinterpret_cast<T*>(p.y);
    if (p.gpos) {
        for (int jrow = tr; jrow < p.Mg; jrow += rows_par) {
            const int t = (int)p.gpos[(long)b * p.Mg + jrow];
            float s[VEC];
            load_sum<T>(p, dk, b, t, c0, s);
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.v[j] = from_f32<T>((s[j] - mean) * rstd * g[j] + be[j]);
            st16<T>(y + ((long)b * p.Mg + jrow) * p.C + c0, o);
        }
    } else if (cached) {
#pragma unroll
        for (int i = 0; i < LN_NR; ++i) {
            const int t = tr + i * rows_par;
            if (t < p.T) {
                Vec16<T> o;
#pragma unroll
                for (int j = 0; j < VEC; ++j) o.v[j] = from_f32<T>((sc[i][j] - mean) * rstd * g[j] + be[j]);
                st16<T>(y + ((long)b * p.T + t) * p.C + c0, o);
            }
        }
    } else {
        for (int t = tr; t < p.T; t += rows_par) {
            float s[VEC];
            load_sum<T>(p, dk, b, t, c0, s);
            Vec16<T> o;
#pragma unroll
            for (int j = 0; j < VEC; ++j) o.v[j] = from_f32<T>((s[j] - mean) * rstd * g[j] + be[j]);
            st16<T>(y + ((long)b * p.T + t) * p.C + c0, o);
        }
    }
}

// dy row for position t of sample b (zero vector if no gathered row points at t).  Several gathered rows may name
// the same position (the reference pads masked_pos with 0, Base.py mask_random): their gradients add, so the rows of
// a position form a chain rowmap[t] -> nextj[..] built in a fixed order.
template <typename T>
__device__ __forceinline__ bool load_dy(const LnP& p, const int* rowmap, const int* nextj, int b, int t, int c0,
                                        float d[ElemTraits<T>::VEC]) {
    constexpr int VEC = ElemTraits<T>::VEC;
    if (!p.gpos) {
        Vec16<T> v = ld16<T>(reinterpret_cast<const T*>(p.dy) + ((long)b * p.T + t) * p.C + c0);
#pragma unroll
        for (int q = 0; q < VEC; ++q) d[q] = to_f32(v.v[q]);
        return true;
    }
#pragma unroll
    for (int q = 0; q < VEC; ++q) d[q] = 0.f;
    bool any = false;
    for (int j = rowmap[t]; j >= 0; j = nextj[j]) {
        long row = (long)b * p.Mg + j;
        if (p.dy_rowmap) row = p.dy_rowmap[row];
        if (row < 0) continue;
        Vec16<T> v = ld16<T>(reinterpret_cast<const T*>(p.dy) + row * p.C + c0);
#pragma unroll
        for (int q = 0; q < VEC; ++q) d[q] += to_f32(v.v[q]);
        any = true;
    }
    return any;
}

template <typename T>
__global__ __launch_bounds__(LN_THREADS) void ln_bwd_kernel(LnP p) {
    constexpr int VEC = ElemTraits<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);            // [8]
    float* cred = red + 8;                                      // [rows_par][2][C] channel partials
    const int b = blockIdx.x, cpv = p.C / VEC;
    const int rows_par = LN_THREADS / cpv, active = rows_par * cpv;
    int* rowmap = reinterpret_cast<int*>(cred + (size_t)rows_par * 2 * p.C);  // [T]
    int* nextj = rowmap + p.T;                                                  // [Mg]
    const int tid = threadIdx.x;
    const bool on = tid < active;
    const int cv = tid % cpv, tr = tid / cpv, c0 = cv * VEC;
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const float n = ln_count(p);
    const float mean = p.stats[2 * b], rstd = p.stats[2 * b + 1];
    float cm[VEC];      // real / padded channel: gamma of a padded channel is 0, so the sums below see nothing of it; its dx := 0
#pragma unroll
    for (int j = 0; j < VEC; ++j) cm[j] = chan_on(p, c0 + j);

    if (p.gpos) {
        for (int t = tid; t < p.T; t += LN_THREADS) rowmap[t] = -1;
        __syncthreads();
        if (tid == 0)
            for (int j = p.Mg - 1; j >= 0; --j) {     // chains in ascending j
                const int t = (int)p.gpos[(long)b * p.Mg + j];
                nextj[j] = rowmap[t];
                rowmap[t] = j;
            }
        __syncthreads();
    }
    float g[VEC];
    if (on) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] = p.gamma[c0 + j];
    }
    float s1 = 0.f, s2 = 0.f, dga[VEC], dbe[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { dga[j] = 0.f; dbe[j] = 0.f; }
    // rows owned by this thread are read once (dy and the dropped-out sum) and kept for the second pass when they fit
    const bool cached = p.T <= LN_NR * rows_par;
    float dc[LN_NR][VEC], sc[LN_NR][VEC];
    if (cached) {
#pragma unroll
        for (int i = 0; i < LN_NR; ++i) {
            const int t = tr + i * rows_par;
            const bool ok = on && t < p.T;
            const int tc = min(t, p.T - 1);
            load_sum<T>(p, dk, b, tc, c0, sc[i]);
            const bool hasd = load_dy<T>(p, rowmap, nextj, b, tc, c0, dc[i]) && ok;
            if (!hasd) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) dc[i][j] = 0.f;
            }
            if (ok) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float xh = (sc[i][j] - mean) * rstd, gg = dc[i][j] * g[j];
                    s1 += gg; s2 += gg * xh; dga[j] += dc[i][j] * xh; dbe[j] += dc[i][j];
                }
            }
        }
    } else if (on) {
        for (int t = tr; t < p.T; t += rows_par) {
            float d[VEC];
            if (!load_dy<T>(p, rowmap, nextj, b, t, c0, d)) continue;
            float s[VEC];
            load_sum<T>(p, dk, b, t, c0, s);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float xh = (s[j] - mean) * rstd, gg = d[j] * g[j];
                s1 += gg; s2 += gg * xh; dga[j] += d[j] * xh; dbe[j] += d[j];
            }
        }
    }
    const float m1 = block_sum(s1, red) / n;
    const float m2 = block_sum(s2, red) / n;
    if (on) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            cred[((size_t)tr * 2 + 0) * p.C + c0 + j] = dbe[j];   // [dbeta | dgamma]: the arena stores beta before gamma
            cred[((size_t)tr * 2 + 1) * p.C + c0 + j] = dga[j];
        }
    }
    __syncthreads();
    for (int i = tid; i < 2 * p.C; i += LN_THREADS) {
        float a = 0.f;
        for (int r = 0; r < rows_par; ++r) a += cred[(size_t)r * 2 * p.C + i];
        p.part[(long)b * 2 * p.C + i] = a;
    }
    if (!on) return;
    T* dsum = reinterpret_cast<T*>(p.dsum);
    T* dxd = reinterpret_cast<T*>(p.dx_drop);
    auto emit = [&](int t, const float (&d)[VEC], const float (&s)[VEC]) {
        const long row = (long)b * p.T + t;
        Vec16<T> o, od;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float xh = (s[j] - mean) * rstd;
            float v = cm[j] * (rstd * (d[j] * g[j] - m1 - xh * m2));
            if (p.act_pre) v *= dgelu_t<T>(to_f32(reinterpret_cast<const T*>(p.act_pre)[row * p.C + c0 + j]));
            o.v[j] = from_f32<T>(v);
            od.v[j] = from_f32<T>(drop_apply(dk, (uint64_t)(row * p.C + c0 + j), v));
        }
        if (dsum) st16<T>(dsum + row * p.C + c0, o);
        if (dxd) st16<T>(dxd + row * p.C + c0, od);
    };
    if (cached) {
#pragma unroll
        for (int i = 0; i < LN_NR; ++i) {
            const int t = tr + i * rows_par;
            if (t < p.T) emit(t, dc[i], sc[i]);
        }
    } else {
        for (int t = tr; t < p.T; t += rows_par) {
            float d[VEC], s[VEC];
            load_dy<T>(p, rowmap, nextj, b, t, c0, d);
            load_sum<T>(p, dk, b, t, c0, s);
            emit(t, d, s);
        }
    }
}

int check_ln_shape(int B, int T, int C, int dtype, const char* who) {
    const int vec = dtype == EDGL_BF16 ? 8 : 4;
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "%s: bad dtype %d", who, dtype);
    EDGL_REQUIRE(B > 0 && T > 0 && C > 0 && C % vec == 0 && C / vec <= LN_THREADS, EDGL_ERR_SHAPE,
                 "%s: unsupported shape B=%d T=%d C=%d (C must be a multiple of %d, <= %d)", who, B, T, C, vec,
                 vec * LN_THREADS);
    return EDGL_OK;
}

}  // namespace

static int check_pad(int C, int dhp, int dht, const char* who) {
    EDGL_REQUIRE((dhp == 0 && dht == 0) || (dhp > 0 && dht > 0 && dht <= dhp && C % dhp == 0), EDGL_ERR_SHAPE,
                 "%s: padded-channel spec dh_pad=%d dh_true=%d does not fit C=%d", who, dhp, dht, C);
    return EDGL_OK;
}
extern "C" int edgl_add_layernorm_fwd_ct(const void* x, const void* resid, int ld_res, const float* gamma,
                                         const float* beta, int B, int T, int C, float drop_rate,
                                         const uint64_t* rng_state, uint32_t stream_id, const int64_t* gather_pos,
                                         int Mg, void* y, float* stats, int dh_pad, int dh_true, int dtype, void* stream);
extern "C" int edgl_add_layernorm_fwd(const void* x, const void* resid, int ld_res, const float* gamma,
                                      const float* beta, int B, int T, int C, float drop_rate,
                                      const uint64_t* rng_state, uint32_t stream_id, const int64_t* gather_pos,
                                      int Mg, void* y, float* stats, int dtype, void* stream) {
    return edgl_add_layernorm_fwd_ct(x, resid, ld_res, gamma, beta, B, T, C, drop_rate, rng_state, stream_id, gather_pos, Mg, y, stats,
                                     0, 0, dtype, stream);
}
extern "C" int edgl_add_layernorm_fwd_ct(const void* x, const void* resid, int ld_res, const float* gamma,
                                         const float* beta, int B, int T, int C, float drop_rate,
                                         const uint64_t* rng_state, uint32_t stream_id, const int64_t* gather_pos,
                                         int Mg, void* y, float* stats, int dh_pad, int dh_true, int dtype, void* stream) {
    EDGL_REQUIRE(x && gamma && beta && y && stats, EDGL_ERR_NULL, "edgl_add_layernorm_fwd: null pointer");
    if (int rcp = check_pad(C, dh_pad, dh_true, "edgl_add_layernorm_fwd")) return rcp;
    int rc = check_ln_shape(B, T, C, dtype, "edgl_add_layernorm_fwd");
    if (rc) return rc;
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_add_layernorm_fwd: dropout without rng_state");
    LnP p{};
    p.x = x; p.resid = resid; p.ld_res = ld_res; p.gamma = gamma; p.beta = beta; p.B = B; p.T = T; p.C = C;
    p.rate = drop_rate; p.rng = rng_state; p.stream_id = stream_id; p.gpos = gather_pos; p.Mg = Mg; p.y = y;
    p.stats = stats; p.dhp = dh_pad; p.dht = dh_true;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_F32) hipLaunchKernelGGL((ln_fwd_kernel<float>), dim3(B), dim3(LN_THREADS), 0, st, p);
    else hipLaunchKernelGGL((ln_fwd_kernel<bf16>), dim3(B), dim3(LN_THREADS), 0, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_add_layernorm_bwd(const void* x, const void* resid, int ld_res, const float* gamma,
                                      const float* stats, const void* dy, int B, int T, int C, float drop_rate,
                                      const uint64_t* rng_state, uint32_t stream_id, const int64_t* gather_pos,
                                      int Mg, const int32_t* dy_rowmap, void* dsum, void* dx_drop, float* dgamma,
                                      float* dbeta, float* workspace, int dtype, void* stream) {
    return edgl_add_layernorm_bwd_act(x, resid, ld_res, gamma, stats, dy, B, T, C, drop_rate, rng_state, stream_id, gather_pos, Mg,
                                      dy_rowmap, nullptr, dsum, dx_drop, dgamma, dbeta, workspace, dtype, stream);
}

extern "C" int edgl_add_layernorm_bwd_act_ct(const void* x, const void* resid, int ld_res, const float* gamma,
                                             const float* stats, const void* dy, int B, int T, int C, float drop_rate,
                                             const uint64_t* rng_state, uint32_t stream_id, const int64_t* gather_pos,
                                             int Mg, const int32_t* dy_rowmap, const void* act_pre, void* dsum, void* dx_drop,
                                             float* dgamma, float* dbeta, float* workspace, int dh_pad, int dh_true, int dtype,
                                             void* stream);
extern "C" int edgl_add_layernorm_bwd_act(const void* x, const void* resid, int ld_res, const float* gamma,
                                          const float* stats, const void* dy, int B, int T, int C, float drop_rate,
                                          const uint64_t* rng_state, uint32_t stream_id, const int64_t* gather_pos,
                                          int Mg, const int32_t* dy_rowmap, const void* act_pre, void* dsum, void* dx_drop,
                                          float* dgamma, float* dbeta, float* workspace, int dtype, void* stream) {
    return edgl_add_layernorm_bwd_act_ct(x, resid, ld_res, gamma, stats, dy, B, T, C, drop_rate, rng_state, stream_id, gather_pos, Mg,
                                         dy_rowmap, act_pre, dsum, dx_drop, dgamma, dbeta, workspace, 0, 0, dtype, stream);
}
extern "C" int edgl_add_layernorm_bwd_act_ct(const void* x, const void* resid, int ld_res, const float* gamma,
                                             const float* stats, const void* dy, int B, int T, int C, float drop_rate,
                                             const uint64_t* rng_state, uint32_t stream_id, const int64_t* gather_pos,
                                             int Mg, const int32_t* dy_rowmap, const void* act_pre, void* dsum, void* dx_drop,
                                             float* dgamma, float* dbeta, float* workspace, int dh_pad, int dh_true, int dtype,
                                             void* stream) {
    EDGL_REQUIRE(x && gamma && stats && dy && dgamma && dbeta && workspace, EDGL_ERR_NULL,
                 "edgl_add_layernorm_bwd: null pointer");
    if (int rcp = check_pad(C, dh_pad, dh_true, "edgl_add_layernorm_bwd")) return rcp;
    int rc = check_ln_shape(B, T, C, dtype, "edgl_add_layernorm_bwd");
    if (rc) return rc;
    LnP p{};
    p.x = x; p.resid = resid; p.ld_res = ld_res; p.gamma = gamma; p.B = B; p.T = T; p.C = C;
    p.rate = drop_rate; p.rng = rng_state; p.stream_id = stream_id; p.gpos = gather_pos; p.Mg = Mg;
    p.stats = const_cast<float*>(stats); p.dy = dy; p.dsum = dsum; p.dx_drop = dx_drop; p.part = workspace;
    p.dy_rowmap = dy_rowmap; p.act_pre = act_pre; p.dhp = dh_pad; p.dht = dh_true;
    const int vec = dtype == EDGL_BF16 ? 8 : 4;
    const int rows_par = LN_THREADS / (C / vec);
    const size_t smem = (8 + (size_t)rows_par * 2 * C) * sizeof(float) + (size_t)(T + (gather_pos ? Mg : 0)) * sizeof(int);
    EDGL_REQUIRE(smem <= 150 * 1024, EDGL_ERR_SHAPE, "edgl_add_layernorm_bwd: LDS need %zu too large", smem);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_F32) hipLaunchKernelGGL((ln_bwd_kernel<float>), dim3(B), dim3(LN_THREADS), smem, st, p);
    else hipLaunchKernelGGL((ln_bwd_kernel<bf16>), dim3(B), dim3(LN_THREADS), smem, st, p);
    EDGL_LAUNCH_CHECK();
    if (dgamma == dbeta + C) return edgl_reduce_rows(workspace, B, 2 * C, 2L * C, dbeta, 0, st);
    rc = edgl_reduce_rows(workspace, B, C, 2L * C, dbeta, 0, st);
    if (rc) return rc;
    return edgl_reduce_rows(workspace + C, B, C, 2L * C, dgamma, 0, st);
}

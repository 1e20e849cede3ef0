// Batch construction on the device — MAUPostProcessor of the reference (src/dataloader.py:159-206) without its
// per-example Python py_func: `choice(seqslen - ignore_head, masklen) + ignore_head` (dataloader.py:34-36,183-186),
// token := MASK at the drawn positions, labels = original tokens there (:188-201); evaluation masks the last position
// (:166-179).  One workgroup (256 threads) per sequence.
#include "edgl_common.h"

namespace {

// M DISTINCT positions in [1, T) per row, uniformly at random (any M-subset equally likely, in random order — the
// semantics of np.random.choice(T-1, M, replace=False) + 1): every position gets a counter-based 32-bit key from
// (seed, step, stream, row, position); the M positions with the smallest (key, position) are taken, rank = output slot.
__global__ __launch_bounds__(256) void mask_random_kernel(const int64_t* tokens, int T, int M, int64_t mask_id,
                                                          const uint64_t* rng, uint32_t stream_id, int64_t* masked,
                                                          int64_t* mpos, int64_t* labels) {
    extern __shared__ uint32_t keys[];   // [T] (entry 0 unused: position 0 is never drawn, ignore_head = 1)
    const int b = blockIdx.x;
    const int64_t* row = tokens + (long)b * T;
    const DropKey dk = make_dropkey(rng, stream_id, 0.f);
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        uint32_t h = ((uint32_t)(b * T + t) ^ dk.k0) * 0x9E3779B1u + dk.k1;
        h ^= h >> 15; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
        keys[t] = h;
    }
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) {
        int64_t tok = row[t];
        if (t >= 1) {
            const uint32_t k = keys[t];
            int rank = 0;
            for (int j = 1; j < T; ++j) {
                const uint32_t kj = keys[j];
                rank += (kj < k) || (kj == k && j < t);
            }
            if (rank < M) {
                mpos[(long)b * M + rank] = t;
                labels[(long)b * M + rank] = tok;
                tok = mask_id;
            }
        }
        masked[(long)b * T + t] = tok;
    }
}

__global__ void mask_last_kernel(const int64_t* tokens, long n, int T, int64_t mask_id, int64_t* masked) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        masked[i] = (i % T == T - 1) ? mask_id : tokens[i];
}

// ---- CTSMA input encoding (CTSMA.py:48-58): X0 = dropout(concat(item_tab[ids] * sqrt(C), pos_tab[0..T))) [B,T,2C],
//      spans[t] = ts[t+1]/scale - ts[t]/scale (float32, no clipping), marks = mark_table[ids].  Thread = 4 channels.
struct EmbP {
    const int64_t* ids; const float* ts; const void* item_tab; const float* pos_tab; const uint8_t* mark_table;
    int B, T, C, E; float time_scale; float rate; const uint64_t* rng; uint32_t stream_id;
    void* x0; float* spans; uint8_t* marks;
    const void* dx0; float* d_item; float* d_pos;
    int srows;   // rows per block of the backward (<= ESROWS; fewer when ESROWS*C floats exceed the LDS)
    float sq;    // coding.py:62-63 sqrt(num_units): of the TRUE width for a channel-padded model (the _ct entry points)
};
template <typename T>
__global__ __launch_bounds__(256) void embed_pos_fwd_kernel(EmbP p) {
    const int cpr = p.C >> 2;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long row = gid / cpr;
    if (row >= (long)p.B * p.T) return;
    const int cv = (int)(gid % cpr), c0 = cv * 4, t = (int)(row % p.T);
    const long b = row / p.T;
    const int64_t id = p.ids[row];
    if (cv == 0 && p.spans) {
        const float* tr = p.ts + b * (p.T + 1) + t;
        p.spans[row] = tr[1] / p.time_scale - tr[0] / p.time_scale;   // CTSMA.py:50-51
        for (int e = 0; e < p.E; ++e) p.marks[row * p.E + e] = p.mark_table[id * p.E + e];   // :54
    }
    const int nseg = p.pos_tab ? 2 : 1;   // item | position (CTSMA) or the item embedding alone (TGAT.py:49, TiSASREC.py:52)
    const long ldo = (long)nseg * p.C;
    const float sq = p.sq;   // coding.py:62-63
    float v[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    if (id != 0) {   // coding.py:56-57 zero-padded row 0
        const Frag4<T> it = frag_ld<T>(reinterpret_cast<const T*>(p.item_tab) + id * p.C + c0);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[0][j] = to_f32(it.v[j]) * sq;
    }
    if (p.pos_tab) {
        const float4 pp = *reinterpret_cast<const float4*>(p.pos_tab + (long)t * p.C + c0);
        v[1][0] = pp.x; v[1][1] = pp.y; v[1][2] = pp.z; v[1][3] = pp.w;
    }
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    T* out = reinterpret_cast<T*>(p.x0) + row * ldo;
    for (int s2 = 0; s2 < nseg; ++s2) {
        Frag4<T> o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.v[j] = from_f32<T>(drop_apply(dk, (uint64_t)row * ldo + s2 * p.C + c0 + j, v[s2][j]));
        if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(out + s2 * p.C + c0) = *reinterpret_cast<uint4*>(&o);
        else *reinterpret_cast<uint2*>(out + s2 * p.C + c0) = *reinterpret_cast<uint2*>(&o);
    }
}
// d_item[id] += sqrt(C) * dropmask * dx[:, :C] (rows with id != 0), d_pos[t] += dropmask * dx[:, C:2C] into zero-filled tables.
// Item popularity is heavy-tailed, so raw global atomics serialise on the hot rows: a block takes ESROWS consecutive (b,t)
// rows, finds for every row the first row of the block with the same id (its leader), accumulates per leader in LDS and
// issues ONE global atomic per distinct id and channel (the scheme of k_encode.hip's encode_scatter_kernel).
constexpr int ESROWS = 128;
template <typename T>
__global__ __launch_bounds__(256) void embed_pos_bwd_kernel(EmbP p) {
    extern __shared__ float acc[];  // [ESROWS][C]
    __shared__ int s_id[ESROWS];
    __shared__ int s_lead[ESROWS];
    const int SR = p.srows;
    const long rows = (long)p.B * p.T, r0 = (long)blockIdx.x * SR;
    const int tid = threadIdx.x;
    if (tid < SR) s_id[tid] = (r0 + tid < rows) ? (int)p.ids[r0 + tid] : -1;
    for (int i = tid; i < SR * p.C; i += 256) acc[i] = 0.f;
    __syncthreads();
    if (tid < SR) {
        const int id = s_id[tid];
        int lead = tid;
        for (int j = 0; j < tid; ++j)
            if (s_id[j] == id) { lead = j; break; }
        s_lead[tid] = lead;
    }
    __syncthreads();
    const int cpr = p.C >> 2, rows_par = 256 / cpr;
    const int cv = tid % cpr, rl = tid / cpr, c0 = cv * 4;
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const float sq = p.sq;
    const long ldo = p.d_pos ? 2L * p.C : (long)p.C;
    if (tid < rows_par * cpr)
        for (int r = rl; r < SR; r += rows_par) {
            if (s_id[r] < 0) continue;   // past the end
            const long row = r0 + r;
            const T* d = reinterpret_cast<const T*>(p.dx0) + row * ldo;
            if (s_id[r] != 0) {          // coding.py:56-57: row 0 is a constant
                const Frag4<T> g0 = frag_ld<T>(d + c0);
                float* dst = acc + s_lead[r] * p.C + c0;
#pragma unroll
                for (int j = 0; j < 4; ++j) atomicAdd(dst + j, sq * drop_apply(dk, (uint64_t)row * ldo + c0 + j, to_f32(g0.v[j])));
            }
            if (p.d_pos) {
                const Frag4<T> g1 = frag_ld<T>(d + p.C + c0);
                const int t = (int)(row % p.T);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    atomicAdd(p.d_pos + (long)t * p.C + c0 + j, drop_apply(dk, (uint64_t)row * ldo + p.C + c0 + j, to_f32(g1.v[j])));
            }
        }
    __syncthreads();
    for (int i = tid; i < SR * p.C; i += 256) {
        const int r = i / p.C, c = i % p.C;
        if (s_lead[r] == r && s_id[r] > 0) atomicAdd(p.d_item + (long)s_id[r] * p.C + c, acc[i]);
    }
}

}  // namespace

extern "C" int edgl_embed_pos_fwd_ct(const int64_t* ids, const float* ts, const void* item_tab, const float* pos_tab,
                                  const uint8_t* mark_table, int B, int T, int C, int E, float time_scale, float drop_rate,
                                  const uint64_t* rng_state, uint32_t stream_id, void* x0, float* spans, uint8_t* marks,
                                  int c_true, int dtype, void* stream) {
    EDGL_REQUIRE(ids && item_tab && x0, EDGL_ERR_NULL, "edgl_embed_pos_fwd: null pointer");
    EDGL_REQUIRE((spans != nullptr) == (marks != nullptr) && (!spans || (ts && mark_table)), EDGL_ERR_NULL,
                 "edgl_embed_pos_fwd: spans and marks go together and need ts and mark_table");
    EDGL_REQUIRE(B > 0 && T > 0 && C > 0 && C % 4 == 0 && (E >= 1 || !spans), EDGL_ERR_SHAPE, "edgl_embed_pos_fwd: bad shape B=%d T=%d C=%d E=%d", B, T, C, E);
    EDGL_REQUIRE(c_true >= 0 && c_true <= C, EDGL_ERR_SHAPE, "edgl_embed_pos_fwd: true width %d exceeds C=%d", c_true, C);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_embed_pos_fwd: bad dtype %d", dtype);
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_embed_pos_fwd: dropout without rng_state");
    EmbP p{ids, ts, item_tab, pos_tab, mark_table, B, T, C, E, time_scale, drop_rate, rng_state, stream_id, x0, spans, marks,
           nullptr, nullptr, nullptr, 0, sqrtf((float)(c_true > 0 ? c_true : C))};
    const long total = (long)B * T * (C / 4);
    dim3 grid((unsigned)((total + 255) / 256));
    if (dtype == EDGL_F32) hipLaunchKernelGGL((embed_pos_fwd_kernel<float>), grid, dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((embed_pos_fwd_kernel<bf16>), grid, dim3(256), 0, (hipStream_t)stream, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_embed_pos_fwd(const int64_t* ids, const float* ts, const void* item_tab, const float* pos_tab,
                                  const uint8_t* mark_table, int B, int T, int C, int E, float time_scale, float drop_rate,
                                  const uint64_t* rng_state, uint32_t stream_id, void* x0, float* spans, uint8_t* marks,
                                  int dtype, void* stream) {
    return edgl_embed_pos_fwd_ct(ids, ts, item_tab, pos_tab, mark_table, B, T, C, E, time_scale, drop_rate, rng_state, stream_id, x0,
                                 spans, marks, 0, dtype, stream);
}

extern "C" int edgl_embed_pos_bwd_ct(const int64_t* ids, const void* dx0, int B, int T, int C, int I, float drop_rate,
                                  const uint64_t* rng_state, uint32_t stream_id, float* d_item, float* d_pos, int c_true,
                                  int dtype, void* stream) {
    EDGL_REQUIRE(ids && dx0 && d_item, EDGL_ERR_NULL, "edgl_embed_pos_bwd: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0 && C > 0 && C % 4 == 0 && C / 4 <= 256 && I > 1,
                 EDGL_ERR_SHAPE, "edgl_embed_pos_bwd: bad shape B=%d T=%d C=%d", B, T, C);
    EDGL_REQUIRE(c_true >= 0 && c_true <= C, EDGL_ERR_SHAPE, "edgl_embed_pos_bwd: true width %d exceeds C=%d", c_true, C);
    int srows = ESROWS;
    while (srows > 8 && (size_t)srows * C * sizeof(float) > 150 * 1024) srows >>= 1;   // C = 512: 64 rows per block
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_embed_pos_bwd: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(d_item, 0, (size_t)I * C * sizeof(float), st) != hipSuccess ||
        (d_pos && hipMemsetAsync(d_pos, 0, (size_t)T * C * sizeof(float), st) != hipSuccess)) {
        edgl_set_error("edgl_embed_pos_bwd: memset failed");
        return EDGL_ERR_LAUNCH;
    }
    EmbP p{ids, nullptr, nullptr, nullptr, nullptr, B, T, C, 0, 1.f, drop_rate, rng_state, stream_id, nullptr, nullptr, nullptr,
           dx0, d_item, d_pos, srows, sqrtf((float)(c_true > 0 ? c_true : C))};
    const size_t smem = (size_t)srows * C * sizeof(float);
    dim3 grid((unsigned)(((long)B * T + srows - 1) / srows));
    if (dtype == EDGL_F32) {
        hipFuncSetAttribute((const void*)embed_pos_bwd_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((embed_pos_bwd_kernel<float>), grid, dim3(256), smem, st, p);
    } else {
        hipFuncSetAttribute((const void*)embed_pos_bwd_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL((embed_pos_bwd_kernel<bf16>), grid, dim3(256), smem, st, p);
    }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_embed_pos_bwd(const int64_t* ids, const void* dx0, int B, int T, int C, int I, float drop_rate,
                                  const uint64_t* rng_state, uint32_t stream_id, float* d_item, float* d_pos, int dtype,
                                  void* stream) {
    return edgl_embed_pos_bwd_ct(ids, dx0, B, T, C, I, drop_rate, rng_state, stream_id, d_item, d_pos, 0, dtype, stream);
}

extern "C" int edgl_mask_random(const int64_t* tokens, int B, int T, int M, int64_t mask_id, const uint64_t* rng_state,
                                uint32_t stream_id, int64_t* masked_tokens, int64_t* masked_pos, int64_t* labels,
                                void* stream) {
    EDGL_REQUIRE(tokens && rng_state && masked_tokens && masked_pos && labels, EDGL_ERR_NULL, "edgl_mask_random: null pointer");
    EDGL_REQUIRE(B > 0 && T > 1 && M >= 1 && M <= T - 1, EDGL_ERR_SHAPE,
                 "edgl_mask_random: need 1 <= masklen <= T-1 (B=%d T=%d M=%d)", B, T, M);
    EDGL_REQUIRE((size_t)T * sizeof(uint32_t) <= 64 * 1024, EDGL_ERR_SHAPE, "edgl_mask_random: T=%d too long", T);
    hipLaunchKernelGGL(mask_random_kernel, dim3(B), dim3(256), (size_t)T * sizeof(uint32_t), (hipStream_t)stream, tokens, T,
                       M, mask_id, rng_state, stream_id, masked_tokens, masked_pos, labels);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_mask_last(const int64_t* tokens, int B, int T, int64_t mask_id, int64_t* masked_tokens, void* stream) {
    EDGL_REQUIRE(tokens && masked_tokens, EDGL_ERR_NULL, "edgl_mask_last: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0, EDGL_ERR_SHAPE, "edgl_mask_last: bad shape B=%d T=%d", B, T);
    const long n = (long)B * T;
    hipLaunchKernelGGL(mask_last_kernel, dim3((unsigned)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0,
                       (hipStream_t)stream, tokens, n, T, mask_id, masked_tokens);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

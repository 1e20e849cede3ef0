// K1: input encoding (EasyDGL.py:70-95; coding.py:60-64 Embedding, :76-79 PositionCoding,
// :137-149 TimeSinusoidCoding) and its backward.  HBM-bound gather: one thread owns the channel
// pair (2j, 2j+1) of a (b,t) row in all three C-wide sections, so a 64-lane wave covers a whole
// C=128 row with contiguous 4/8-byte stores and one sincosf per pair.
#include "edgl_common.h"

namespace {

struct EncP {
    const int64_t* ids; const float* ts; const void* item_tab; const float* pos_tab; const float* mark_emb;
    const uint8_t* mark_table; const float* tscale;
    int B, T, C, E, I; int64_t mask_id; float time_scale;
    float rate; const uint64_t* rng; uint32_t stream_id;
    void* x0; float* spans; uint8_t* marks;
};

template <typename T>
__global__ __launch_bounds__(256) void encode_fwd_kernel(EncP p) {
    const int half = p.C >> 1;
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long row = gid / half;
    if (row >= (long)p.B * p.T) return;
    const int j = (int)(gid % half), c = 2 * j;
    const int t = (int)(row % p.T);
    const int64_t id = p.ids[row];
    // EasyDGL.py:71 — float32 division (quantisation point shared with the oracle)
    const float tsx = p.ts[row] / p.time_scale;
    // EasyDGL.py:76-77 — MASK -> row 0 of the mark table
    const int64_t mid = (id == p.mask_id) ? 0 : id;
    const uint8_t* mrow = p.mark_table + mid * p.E;
    int nm = 0;
    for (int e = 0; e < p.E; ++e) nm += mrow[e];

    if (j == 0) {
        // EasyDGL.py:73-74 — span[t] = clip(ts[t]-ts[t-1], 0, 100); span[0] := span[1]
        const int t1 = (t == 0) ? 1 : t;
        float sp = 0.f;
        if (p.T > 1) {
            const long r1 = row - t + t1;
            const float a = p.ts[r1] / p.time_scale, bq = p.ts[r1 - 1] / p.time_scale;
            sp = fminf(fmaxf(a - bq, 0.f), 100.f);
        }
        p.spans[row] = sp;
    }
    if (j < p.E) p.marks[row * p.E + j] = mrow[j];
    if (j == 0 && half < p.E)
        for (int e = half; e < p.E; ++e) p.marks[row * p.E + e] = mrow[e];

    // coding.py:141-145 — x / scale (float32 division), sin on even / cos on odd channels
    const float arg = tsx / p.tscale[j];
    float sn, cs;
    sincosf(arg, &sn, &cs);
    float e0 = 0.f, e1 = 0.f;
    if (id != 0) {  // coding.py:56-57 zero-padded row 0
        const T* it = reinterpret_cast<const T*>(p.item_tab) + id * p.C + c;
        e0 = to_f32(it[0]); e1 = to_f32(it[1]);
    }
    const float sq = sqrtf((float)p.C);  // coding.py:62-63
    float v[6];
    v[0] = e0 * sq + sn; v[1] = e1 * sq + cs;
    v[2] = p.pos_tab[t * p.C + c]; v[3] = p.pos_tab[t * p.C + c + 1];
    // EasyDGL.py:87-88 — 0/1 mark values index the zero-padded mark-embedding table
    const float fn = (float)nm;
    v[4] = (p.E > 1) ? fn * p.mark_emb[p.C + c] : 0.f;
    v[5] = (p.E > 1) ? fn * p.mark_emb[p.C + c + 1] : 0.f;
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    T* out = reinterpret_cast<T*>(p.x0) + row * 3 * p.C;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const uint64_t idx = (uint64_t)row * 3 * p.C + s * p.C + c;
        out[s * p.C + c] = from_f32<T>(drop_apply(dk, idx, v[2 * s]));
        out[s * p.C + c + 1] = from_f32<T>(drop_apply(dk, idx + 1, v[2 * s + 1]));
    }
}

struct EncBwdP {
    const int64_t* ids; const uint8_t* marks; const void* dx0;
    int B, T, C, E, I; float rate; const uint64_t* rng; uint32_t stream_id;
    float* d_item; float* part; int nchunk;
};

// grid (T, nchunk); thread c accumulates over the b-range of its chunk for position t
template <typename T>
__global__ __launch_bounds__(256) void encode_bwd_kernel(EncBwdP p) {
    const int t = blockIdx.x, chunk = blockIdx.y;
    const int bper = (p.B + p.nchunk - 1) / p.nchunk;
    const int b0 = chunk * bper, b1 = min(p.B, b0 + bper);
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const float sq = sqrtf((float)p.C);
    for (int c = threadIdx.x; c < p.C; c += blockDim.x) {
        float apos = 0.f, amk = 0.f;
        for (int b = b0; b < b1; ++b) {
            const long row = (long)b * p.T + t;
            const T* d = reinterpret_cast<const T*>(p.dx0) + row * 3 * p.C;
            const uint64_t base = (uint64_t)row * 3 * p.C;
            const float g0 = drop_apply(dk, base + c, to_f32(d[c]));
            const float g1 = drop_apply(dk, base + p.C + c, to_f32(d[p.C + c]));
            const float g2 = drop_apply(dk, base + 2 * p.C + c, to_f32(d[2 * p.C + c]));
            const int64_t id = p.ids[row];
            if (id != 0) atomicAdd(p.d_item + id * p.C + c, sq * g0);
            apos += g1;
            int nm = 0;
            const uint8_t* mrow = p.marks + row * p.E;
            for (int e = 0; e < p.E; ++e) nm += mrow[e];
            amk += (float)nm * g2;
        }
        float* dst = p.part + (((long)chunk * p.T + t) * 2) * p.C;
        dst[c] = apos;
        dst[p.C + c] = amk;
    }
}

__global__ void encode_bwd_final_kernel(const float* part, int nchunk, int T, int C, int E, float* d_pos, float* d_mark) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    // first T*C outputs: d_pos ; next E*C: d_mark_emb (only row 1 is non-zero)
    if (i < T * C) {
        const int t = i / C, c = i % C;
        float a = 0.f;
        for (int k = 0; k < nchunk; ++k) a += part[(((long)k * T + t) * 2) * C + c];
        d_pos[i] = a;
    } else if (i < T * C + E * C) {
        const int q = i - T * C, e = q / C, c = q % C;
        float a = 0.f;
        if (e == 1)
            for (int k = 0; k < nchunk; ++k)
                for (int t = 0; t < T; ++t) a += part[(((long)k * T + t) * 2 + 1) * C + c];
        d_mark[q] = a;
    }
}

constexpr int ENC_NCHUNK = 8;

}  // namespace

extern "C" long edgl_encode_bwd_workspace(int B, int T, int C) { (void)B; return (long)ENC_NCHUNK * T * 2 * C; }

extern "C" int edgl_encode_fwd(const int64_t* ids, const float* ts, const void* item_tab, const float* pos_tab,
                               const float* mark_emb, const uint8_t* mark_table, const float* tscale, int B, int T,
                               int C, int E, int I, int64_t mask_id, float time_scale, float drop_rate,
                               const uint64_t* rng_state, uint32_t stream_id, void* x0, float* spans,
                               uint8_t* marks, int dtype, void* stream) {
    EDGL_REQUIRE(ids && ts && item_tab && pos_tab && mark_emb && mark_table && tscale && x0 && spans && marks,
                 EDGL_ERR_NULL, "edgl_encode_fwd: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0 && C > 0 && (C % 2) == 0 && E >= 1 && I > 1, EDGL_ERR_SHAPE,
                 "edgl_encode_fwd: bad shape B=%d T=%d C=%d E=%d I=%d", B, T, C, E, I);
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_encode_fwd: dropout without rng_state");
    EncP p{ids, ts, item_tab, pos_tab, mark_emb, mark_table, tscale, B, T, C, E, I, mask_id, time_scale,
           drop_rate, rng_state, stream_id, x0, spans, marks};
    const long total = (long)B * T * (C / 2);
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_F32) hipLaunchKernelGGL((encode_fwd_kernel<float>), grid, dim3(256), 0, st, p);
    else if (dtype == EDGL_BF16) hipLaunchKernelGGL((encode_fwd_kernel<bf16>), grid, dim3(256), 0, st, p);
    else { edgl_set_error("edgl_encode_fwd: bad dtype %d", dtype); return EDGL_ERR_DTYPE; }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_encode_bwd(const int64_t* ids, const uint8_t* marks, const void* dx0, int B, int T, int C,
                               int E, int I, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                               float* d_item, float* d_pos, float* d_mark_emb, float* workspace, int dtype,
                               void* stream) {
    EDGL_REQUIRE(ids && marks && dx0 && d_item && d_pos && d_mark_emb && workspace, EDGL_ERR_NULL,
                 "edgl_encode_bwd: null pointer");
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_encode_bwd: bad dtype %d", dtype);
    EncBwdP p{ids, marks, dx0, B, T, C, E, I, drop_rate, rng_state, stream_id, d_item, workspace, ENC_NCHUNK};
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(T, ENC_NCHUNK);
    if (dtype == EDGL_F32) hipLaunchKernelGGL((encode_bwd_kernel<float>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((encode_bwd_kernel<bf16>), grid, dim3(256), 0, st, p);
    EDGL_LAUNCH_CHECK();
    const int total = T * C + E * C;
    hipLaunchKernelGGL(encode_bwd_final_kernel, dim3((total + 255) / 256), dim3(256), 0, st, workspace, ENC_NCHUNK,
                       T, C, E, d_pos, d_mark_emb);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

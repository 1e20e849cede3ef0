// K1: input encoding (EasyDGL.py:70-95; coding.py:60-64 Embedding, :76-79 PositionCoding,
// :137-149 TimeSinusoidCoding) and its backward.  HBM-bound gather: one thread owns the channel
// pair (2j, 2j+1) of a (b,t) row in all three C-wide sections, so a 64-lane wave covers a whole
// C=128 row with contiguous 4/8-byte stores and one sincosf per pair.
#include "edgl_common.h"
#include "batch_prep.h"

namespace {

struct EncP {
    const int64_t* ids; const float* ts; const void* item_tab; const float* pos_tab; const float* mark_emb;
    const uint8_t* mark_table; const float* tscale;
    int B, T, C, E, I; int64_t mask_id; float time_scale;
    float rate; const uint64_t* rng; uint32_t stream_id;
    void* x0; float* spans; uint8_t* marks;
    // zero-padded channels (a model width whose head dim the attention kernels do not tile: coding.py's sqrt(C) scale and time code
    // are those of the TRUE width): channel c is real iff c % dhp < dht (dhp == 0: no padding); `sq` = sqrt(true width)
    int dhp, dht; float sq;
};

template <typename T>
__device__ __forceinline__ void encode_fwd_block(const EncP& p, long block) {
    // thread = 8 consecutive channels (4 sin/cos pairs) of one (b,t) row in all three C-wide sections: 16-byte item-row
    // load, 16-byte (bf16) / 2 x 16-byte (f32) stores per section; C/8 threads share the row's id / timestamp / marks.
    const int cpr = p.C >> 3;
    const long gid = block * blockDim.x + threadIdx.x;
    const long row = gid / cpr;
    if (row >= (long)p.B * p.T) return;
    const int cv = (int)(gid % cpr), c0 = cv * 8, j0 = c0 >> 1;
    const int t = (int)(row % p.T);
    // Two rounds of loads: everything addressed by (row, t, channel) — id, timestamps, time scales, position row, mark embedding —
    // then the two rows addressed by the id (mark-table row, item row; unconditional, id 0 / MASK handled by selects).  Written
    // where first used, each was a load with its own wait: ~6 dependent round trips in a kernel that lives for 1.6 block rounds.
    const int64_t id = p.ids[row];
    const int t1 = (t == 0) ? 1 : t;
    const long r1 = p.T > 1 ? row - t + t1 : row + 1;     // (T == 1: ts has no neighbour; the span is 0 below)
    const float ts0 = p.ts[row], tsa = p.ts[p.T > 1 ? r1 : row], tsb = p.ts[p.T > 1 ? r1 - 1 : row];
    const float4 sc4 = *reinterpret_cast<const float4*>(p.tscale + j0);
    const float4 p0 = *reinterpret_cast<const float4*>(p.pos_tab + t * p.C + c0), p1 = *reinterpret_cast<const float4*>(p.pos_tab + t * p.C + c0 + 4);
    const float4 m0 = *reinterpret_cast<const float4*>(p.mark_emb + p.C + c0), m1 = *reinterpret_cast<const float4*>(p.mark_emb + p.C + c0 + 4);
    // EasyDGL.py:71 — float32 division (quantisation point shared with the oracle)
    const float tsx = ts0 / p.time_scale;
    // EasyDGL.py:76-77 — MASK -> row 0 of the mark table
    const int64_t mid = (id == p.mask_id) ? 0 : id;
    const uint8_t* mrow = p.mark_table + mid * p.E;
    const T* it = reinterpret_cast<const T*>(p.item_tab) + id * p.C + c0;      // id 0: row 0 is read and dropped
    Vec16<T> iv0 = ld16<T>(it), iv1 = iv0;
    if constexpr (sizeof(T) == 4) iv1 = ld16<T>(it + 4);
    int nm = 0;
    if (p.E == 16) {   // one 16-byte row: four v_sad_u8 instead of 16 dependent byte loads
        const uint4 mv = *reinterpret_cast<const uint4*>(mrow);
        nm = (int)__builtin_amdgcn_sad_u8(mv.x, 0u, 0u);
        nm = (int)__builtin_amdgcn_sad_u8(mv.y, 0u, (uint32_t)nm);
        nm = (int)__builtin_amdgcn_sad_u8(mv.z, 0u, (uint32_t)nm);
        nm = (int)__builtin_amdgcn_sad_u8(mv.w, 0u, (uint32_t)nm);
        if (cv == 0) *reinterpret_cast<uint4*>(p.marks + row * 16) = mv;
    } else {
        for (int e = 0; e < p.E; ++e) nm += mrow[e];
        if (cv == 0)
            for (int e = 0; e < p.E; ++e) p.marks[row * p.E + e] = mrow[e];
    }
    if (cv == 0) {
        // EasyDGL.py:73-74 — span[t] = clip(ts[t]-ts[t-1], 0, 100); span[0] := span[1]
        const float a = tsa / p.time_scale, bq = tsb / p.time_scale;
        p.spans[row] = p.T > 1 ? fminf(fmaxf(a - bq, 0.f), 100.f) : 0.f;
    }
    float v[3][8];
    // coding.py:141-145 — x / scale (float32 division), sin on even / cos on odd channels
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float arg = tsx / (q == 0 ? sc4.x : q == 1 ? sc4.y : q == 2 ? sc4.z : sc4.w);
        float sn, cs;
        if constexpr (sizeof(T) == 4) {
            sincosf(arg, &sn, &cs);   // parity mode: full-precision sin/cos of the float32 argument
        } else {
            // bf16 activations: the argument (up to ~1e4 rad for day-scaled Unix times) is reduced to [-1/2, 1/2]
            // revolutions in double precision (exact to ~1e-12), then the hardware sin/cos (input in revolutions,
            // ~1e-6 absolute) — two transcendental instructions instead of the ~100-instruction libm path; the result
            // is rounded to bf16 anyway.
            const double rev = (double)arg * 0.15915494309189533577;   // 1 / (2 pi)
            const float fr = (float)(rev - __builtin_rint(rev));
            sn = __builtin_amdgcn_sinf(fr);
            cs = __builtin_amdgcn_cosf(fr);
        }
        v[0][2 * q] = sn; v[0][2 * q + 1] = cs;
    }
    if (p.dhp) {   // padded channels carry no time code (their item / position / mark entries are zero in the tables)
#pragma unroll
        for (int q = 0; q < 8; ++q) v[0][q] = ((c0 + q) % p.dhp) < p.dht ? v[0][q] : 0.f;
    }
    const float sq = p.sq;  // coding.py:62-63
    {   // coding.py:56-57 zero-padded row 0
        const float on = id != 0 ? sq : 0.f;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) v[0][q] = id != 0 ? v[0][q] + to_f32(iv0.v[q]) * on : v[0][q];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[0][q] = id != 0 ? v[0][q] + to_f32(iv0.v[q]) * on : v[0][q];
                v[0][4 + q] = id != 0 ? v[0][4 + q] + to_f32(iv1.v[q]) * on : v[0][4 + q];
            }
        }
    }
    {
        v[1][0] = p0.x; v[1][1] = p0.y; v[1][2] = p0.z; v[1][3] = p0.w; v[1][4] = p1.x; v[1][5] = p1.y; v[1][6] = p1.z; v[1][7] = p1.w;
        // EasyDGL.py:87-88 — 0/1 mark values index the zero-padded mark-embedding table
        const float fn = (p.E > 1) ? (float)nm : 0.f;
        v[2][0] = fn * m0.x; v[2][1] = fn * m0.y; v[2][2] = fn * m0.z; v[2][3] = fn * m0.w;
        v[2][4] = fn * m1.x; v[2][5] = fn * m1.y; v[2][6] = fn * m1.z; v[2][7] = fn * m1.w;
    }
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    T* out = reinterpret_cast<T*>(p.x0) + row * 3 * p.C;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const uint64_t idx = (uint64_t)row * 3 * p.C + s * p.C + c0;
        // one hash per four elements (drop_apply4): 24 per-element hashes were half of this kernel's instructions
        float lo[4] = {v[s][0], v[s][1], v[s][2], v[s][3]}, hi[4] = {v[s][4], v[s][5], v[s][6], v[s][7]};
        drop_apply4(dk, idx, lo);
        drop_apply4(dk, idx + 4, hi);
        if constexpr (sizeof(T) == 2) {
            Vec16<T> o;
#pragma unroll
            for (int q = 0; q < 4; ++q) { o.v[q] = from_f32<T>(lo[q]); o.v[4 + q] = from_f32<T>(hi[q]); }
            st16<T>(out + s * p.C + c0, o);
        } else {
            Vec16<T> o0, o1;
#pragma unroll
            for (int q = 0; q < 4; ++q) { o0.v[q] = from_f32<T>(lo[q]); o1.v[q] = from_f32<T>(hi[q]); }
            st16<T>(out + s * p.C + c0, o0);
            st16<T>(out + s * p.C + c0 + 4, o1);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void encode_fwd_kernel(EncP p) { encode_fwd_block<T>(p, (long)blockIdx.x); }

// The encoder's launch with the batch preparation of the step as its first workgroups: workgroup 0 = the row compaction map of the
// scoring, workgroups 1 .. B = the slot data of the TPP regulariser (sample by sample), the rest = the encoder.  What they write is
// read by later kernels of the same stream only.
struct PrepP {
    const int64_t* labels; int R; int32_t* perm; int32_t* inv; int32_t* nvalid; int64_t* labels_c;      // labels [R] = [B * M]
    const int64_t* mpos; const float* ts_raw; int M; char* desc;                                       // desc == NULL: no slot data
    int nprep;                                                                                         // 1 + (desc ? B : 0)
};
template <typename T>
__global__ __launch_bounds__(256) void encode_prep_kernel(EncP p, PrepP q) {
    extern __shared__ __attribute__((aligned(16))) char prep_smem[];
    const int bid = (int)blockIdx.x;
    if (bid == 0) {
        batch_prep::compact_scan_body<256>(q.labels, q.R, q.perm, q.inv, q.nvalid, q.labels_c, reinterpret_cast<int*>(prep_smem));
    } else if (bid < q.nprep) {
        batch_prep::tpp_prep_sample(q.mpos, q.labels, q.ts_raw, p.mark_table, p.B, p.T, q.M, q.desc, bid - 1, prep_smem);
    } else {
        encode_fwd_block<T>(p, (long)(bid - q.nprep));
    }
}

struct EncBwdP {
    const int64_t* ids; const uint8_t* marks; const void* dx0;
    int B, T, C, E, I; float rate; const uint64_t* rng; uint32_t stream_id;
    float* d_item; float* part_pos; float* part_mk; int nchunk;
    int srows;
    const void* add1; const void* add2;   // optional [B*T, C] terms added to the item section of dX0 (residual branches)
    float sq;             // sqrt(true model width): coding.py:62-63 (== sqrt(C) without channel padding)
    // Optional second job of the MFMA scatter launch (encode_scatter_mfma_kernel): the one-hot term of the tied table's scoring
    // gradient, d_table[label[r]] -= coef[r] rows[r], d_bias[label[r] - 1] -= coef[r] over the weighted rows (Appendix C; what
    // edgl_score_flash_label_term applies as a launch of its own) — the same segmented sum over equal ids, on blocks behind the
    // embedding's: lab_blk0 = first such block (0: none)
    const void* lab_rows; const int64_t* lab_ids; const float* lab_coef; const int32_t* lab_nvalid; int lab_R; int lab_blk0;
    float* d_bias;
    float* d_mark_zero;   // [E*C]: cleared by block (0, 0) of the position / mark stage (only row 1 is ever written afterwards)   // rows per block of the scatter stage (<= SROWS; fewer when SROWS*C floats exceed the LDS)
};

// d_pos / d_mark: grid (T, nchunk): the block owns position t for the b-range of its chunk.  Thread = 4
// consecutive channels of the position and mark sections (8/16-byte loads); 256/(C/4) rows in flight; block
// partials reduced in a fixed order.
template <typename T>
__global__ __launch_bounds__(256) void encode_bwd_kernel(EncBwdP p) {
    extern __shared__ float sm[];  // [rows_par][2][C]
    __shared__ float s_nm[64];
    const int t = blockIdx.x, chunk = blockIdx.y;
    if (t == 0 && chunk == 0 && p.d_mark_zero)
        for (int i = threadIdx.x; i < p.E * p.C; i += blockDim.x) p.d_mark_zero[i] = 0.f;
    const int bper = (p.B + p.nchunk - 1) / p.nchunk;
    const int b0 = chunk * bper, b1 = min(p.B, b0 + bper);
    const int cpr = p.C / 4, rows_par = 256 / cpr;
    const int cv = threadIdx.x % cpr, rl = threadIdx.x / cpr, c0 = cv * 4;
    const bool on = threadIdx.x < rows_par * cpr;
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    float apos[4] = {0.f, 0.f, 0.f, 0.f}, amk[4] = {0.f, 0.f, 0.f, 0.f};
    for (int bb = b0; bb < b1; bb += 64) {
        __syncthreads();
        if (threadIdx.x < 64 && bb + threadIdx.x < b1) {
            const uint8_t* mrow = p.marks + ((long)(bb + threadIdx.x) * p.T + t) * p.E;
            int nm = 0;
            for (int e = 0; e < p.E; ++e) nm += mrow[e];
            s_nm[threadIdx.x] = (float)nm;
        }
        __syncthreads();
        if (on)
            for (int b = bb + rl; b < min(b1, bb + 64); b += rows_par) {
                const long row = (long)b * p.T + t;
                const T* d = reinterpret_cast<const T*>(p.dx0) + row * 3 * p.C;
                const uint64_t base = (uint64_t)row * 3 * p.C;
                const Frag4<T> g1 = frag_ld<T>(d + p.C + c0), g2 = frag_ld<T>(d + 2 * p.C + c0);
                const float nm = s_nm[b - bb];
                float v1[4], v2[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) { v1[j] = to_f32(g1.v[j]); v2[j] = to_f32(g2.v[j]); }
                drop_apply4(dk, base + p.C + c0, v1);
                drop_apply4(dk, base + 2 * p.C + c0, v2);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    apos[j] += v1[j];
                    amk[j] += nm * v2[j];
                }
            }
    }
    __syncthreads();
    if (on) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            sm[(rl * 2 + 0) * p.C + c0 + j] = apos[j];
            sm[(rl * 2 + 1) * p.C + c0 + j] = amk[j];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * p.C; i += 256) {
        float a = 0.f;
        for (int r = 0; r < rows_par; ++r) a += sm[r * 2 * p.C + i];
        const long slot = (long)chunk * p.T + t;
        if (i < p.C) p.part_pos[slot * p.C + i] = a;
        else p.part_mk[slot * p.C + (i - p.C)] = a;
    }
}

// d_item[id] += sqrt(C) * dX0[row, :C] (coding.py:62-63).  Item popularity is heavy-tailed (the synthetic
// Zipf(1.1) stream puts ~1/3 of all tokens on one id), so raw global atomics serialise on the hot rows.  Each
// block takes SROWS consecutive (b,t) rows, finds for every row the first row of the block with the same id
// (its "leader"), accumulates into LDS (ds_add_f32) per leader, and issues ONE global atomic per distinct id
// and channel — the hottest row receives (#blocks) adds instead of (#tokens).
constexpr int SROWS = 128;
template <typename T>
__global__ __launch_bounds__(256) void encode_scatter_kernel(EncBwdP p) {
    extern __shared__ float acc[];  // [SROWS][C]
    __shared__ __attribute__((aligned(16))) int s_id[SROWS];
    __shared__ int s_lead[SROWS];
    const int SR = p.srows;
    const long rows = (long)p.B * p.T, r0 = (long)blockIdx.x * SR;
    const int tid = threadIdx.x;
    if (tid < SR) s_id[tid] = (r0 + tid < rows) ? (int)p.ids[r0 + tid] : 0;
    if (tid >= SR && tid < SROWS) s_id[tid] = -1;   // the leader search reads whole int4 groups
    const int cpr = p.C / 4, rows_par = 256 / cpr;
    const int cv = tid % cpr, rl = tid / cpr, c0 = cv * 4;
    // this thread's rows (rl, rl + rows_par, ...): all gradient fragments are fetched up front, rows clamped — a load
    // inside the per-row loop costs one HBM round trip per row (16 of them at C = 128)
    constexpr int MAXR = 16;
    Frag4<T> g[MAXR];
    const bool worker = tid < rows_par * cpr;
    // gradient of the item section: dX0[:, :C] (+ the two residual-branch terms of the first block, summed in f32)
    auto item_grad = [&](long row) {
        Frag4<T> v = frag_ld<T>(reinterpret_cast<const T*>(p.dx0) + row * 3 * p.C + c0);
        if (p.add1) {
            const Frag4<T> a = frag_ld<T>(reinterpret_cast<const T*>(p.add1) + row * p.C + c0);
            const Frag4<T> b = frag_ld<T>(reinterpret_cast<const T*>(p.add2) + row * p.C + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j) v.v[j] = from_f32<T>(to_f32(v.v[j]) + to_f32(a.v[j]) + to_f32(b.v[j]));
        }
        return v;
    };
#pragma unroll
    for (int k = 0; k < MAXR; ++k) g[k] = item_grad(min(r0 + rl + (long)k * rows_par, rows - 1));
    for (int i = tid; i < SR * p.C; i += 256) acc[i] = 0.f;
    __syncthreads();
    if (tid < SR) {
        // leader = first row of the block with the same id: branch-free scan over int4 groups (a per-element scan with an
        // early exit is a chain of up to 127 dependent LDS reads)
        const int id = s_id[tid];
        int lead = tid;
        for (int j4 = (tid >> 2); j4 >= 0; --j4) {
            const int4 v = *reinterpret_cast<const int4*>(s_id + 4 * j4);
            const int j = 4 * j4;
            if (v.w == id && j + 3 < tid) lead = j + 3;
            if (v.z == id && j + 2 < tid) lead = j + 2;
            if (v.y == id && j + 1 < tid) lead = j + 1;
            if (v.x == id && j < tid) lead = j;
        }
        s_lead[tid] = lead;
    }
    __syncthreads();
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
    const float sq = p.sq;
    if (worker) {
#pragma unroll
        for (int k = 0; k < MAXR; ++k) {
            const int r = rl + k * rows_par;
            if (r >= SR || s_id[r] == 0) continue;  // padding rows and rows past the end
            const long row = r0 + r;
            float* dst = acc + s_lead[r] * p.C + c0;
            float gv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = to_f32(g[k].v[j]);
            drop_apply4(dk, (uint64_t)row * 3 * p.C + c0, gv);
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(dst + j, sq * gv[j]);
        }
        for (int r = rl + MAXR * rows_par; r < SR; r += rows_par) {   // (C < 128: more than MAXR rows per thread)
            if (s_id[r] == 0) continue;
            const long row = r0 + r;
            const Frag4<T> g0 = item_grad(row);
            float* dst = acc + s_lead[r] * p.C + c0;
            float gv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) gv[j] = to_f32(g0.v[j]);
            drop_apply4(dk, (uint64_t)row * 3 * p.C + c0, gv);
#pragma unroll
            for (int j = 0; j < 4; ++j) atomicAdd(dst + j, sq * gv[j]);
        }
    }
    __syncthreads();
    for (int i = tid; i < SR * p.C; i += 256) {
        const int r = i / p.C, c = i % p.C;
        if (s_lead[r] == r && s_id[r] != 0) atomicAdd(p.d_item + (long)s_id[r] * p.C + c, acc[i]);
    }
}

// bf16, C = 16 * CT <= 128: the same per-block aggregation as ONE matrix product on the MFMA pipe —
//     acc[leader][c] = sum_row onehot[leader][row] * masked_g[row][c]
// with the dropout mask applied as exact 0 / g in bf16 and the scalar sqrt(C) / (1 - rate) applied to the f32 sums.  It
// replaces 64 KB of LDS zeroing, 64 LDS atomics per thread and a 64-trip read-out loop (the kernel took 65 us with the
// global atomics compiled out: they were never the limit) by 64 MFMAs per wave; the block's sums are formed in one fixed
// order.  Leaders then add their rows to the table gradient straight from the accumulator registers.
typedef __attribute__((ext_vector_type(4))) short enc_s16x4;
__device__ __forceinline__ uint2 enc_tr_read(const bf16* tile, int ld, int k0, int z0, int lane) {
    const int G = lane >> 4, s = lane & 15;
    const bf16* p = tile + (k0 + 4 * G + (s >> 2)) * ld + z0 + 4 * (s & 3);
    enc_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) enc_s16x4*)p);
    return *reinterpret_cast<uint2*>(&v);
}
// CT = channel tiles of one workgroup's slice (C = 16 CT channels starting at blockIdx.y * C of the p.C-wide rows: the 256 / 512-unit
// recipes run as 2 / 4 slices of 128)
template <int CT>
__global__ __launch_bounds__(256) void encode_scatter_mfma_kernel(EncBwdP p) {
    constexpr int C = 16 * CT, LDG = C + 8, SR = SROWS;
    const int CF = p.C, coff = (int)blockIdx.y * C;      // full row width, first channel of this slice
    __shared__ __attribute__((aligned(16))) bf16 Gs[SR * LDG];   // masked gradient rows of the block
    __shared__ __attribute__((aligned(16))) int s_id[SR];
    __shared__ __attribute__((aligned(16))) int s_lead[SR];       // first row with the same id; -1: padding / past the end
    // label job (block-uniform): rows = the compacted head rows, ids = their labels, every row scaled by its loss coefficient
    const bool lab = p.lab_blk0 > 0 && (int)blockIdx.x >= p.lab_blk0;
    const long rows = lab ? (long)(p.lab_nvalid ? min(p.lab_R, p.lab_nvalid[0]) : p.lab_R) : (long)p.B * p.T;
    const long r0 = (long)((int)blockIdx.x - (lab ? p.lab_blk0 : 0)) * SR;
    if (lab && r0 >= rows) return;
    __shared__ float s_cf[SR], s_cb[SR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = lane >> 4, g4 = G * 4, l15 = lane & 15;
    if (tid < SR) {
        const bool in = r0 + tid < rows;
        if (lab) {
            const float cf = in ? p.lab_coef[r0 + tid] : 0.f;
            s_id[tid] = (in && cf != 0.f) ? (int)p.lab_ids[r0 + tid] : 0;      // label 0: weight 0 (EasyDGL.py:180)
            s_cf[tid] = cf;
            s_cb[tid] = 0.f;
        } else {
            s_id[tid] = in ? (int)p.ids[r0 + tid] : 0;
        }
    }
    // gradient fragments of this thread's rows, all in flight before anything else (rows clamped)
    constexpr int cpr = C / 4, rows_par = 256 / cpr, NR = SR / rows_par;
    const int cv = tid % cpr, rl = tid / cpr, c0 = cv * 4;
    const bf16* dx0 = reinterpret_cast<const bf16*>(lab ? p.lab_rows : p.dx0);
    const long ldrow = lab ? CF : 3 * CF;
    const bool adds = p.add1 && !lab;
    Frag4<bf16> g[NR], ga[NR], gb[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const long row = min(r0 + rl + (long)k * rows_par, rows - 1);
        g[k] = frag_ld<bf16>(dx0 + row * ldrow + coff + c0);
        if (adds) {
            ga[k] = frag_ld<bf16>(reinterpret_cast<const bf16*>(p.add1) + row * CF + coff + c0);
            gb[k] = frag_ld<bf16>(reinterpret_cast<const bf16*>(p.add2) + row * CF + coff + c0);
        }
    }
    __syncthreads();
    if (tid < SR) {
        const int id = s_id[tid];
        int lead = tid;
        for (int j4 = (tid >> 2); j4 >= 0; --j4) {
            const int4 v = *reinterpret_cast<const int4*>(s_id + 4 * j4);
            const int j = 4 * j4;
            if (v.w == id && j + 3 < tid) lead = j + 3;
            if (v.z == id && j + 2 < tid) lead = j + 2;
            if (v.y == id && j + 1 < tid) lead = j + 1;
            if (v.x == id && j < tid) lead = j;
        }
        s_lead[tid] = id == 0 ? -1 : lead;
        if (lab && id != 0 && blockIdx.y == 0) atomicAdd(&s_cb[lead], s_cf[tid]);     // bias term of the leader's label
    }
    const DropKey dk = make_dropkey(p.rng, p.stream_id, p.rate);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int r = rl + k * rows_par;
        const long row = r0 + r;
        Frag4<bf16> v = g[k];
        if (adds) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v.v[j] = from_f32<bf16>(to_f32(v.v[j]) + to_f32(ga[k].v[j]) + to_f32(gb[k].v[j]));
        }
        if (lab) {   // coef[r] * rows[r], rounded to the operand dtype of the segmented sum (as the product pass rounds its P)
            const float cf = s_cf[r];
#pragma unroll
            for (int j = 0; j < 4; ++j) v.v[j] = from_f32<bf16>(cf * to_f32(v.v[j]));
        } else if (dk.thresh != 0u) {
            const uint32_t keep = drop_keep4(dk, (uint64_t)row * 3 * CF + coff + c0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (!((keep >> j) & 1u)) v.v[j] = from_f32<bf16>(0.f);
        }
        *reinterpret_cast<uint2*>(Gs + r * LDG + c0) = *reinterpret_cast<const uint2*>(&v);
    }
    __syncthreads();
    // wave w: leaders [32 w, 32 w + 32) x all channels
    f32x4 acc[2][CT];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[mt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    const bf16 one = from_f32<bf16>(1.f), zero = from_f32<bf16>(0.f);
#pragma unroll
    for (int kb = 0; kb < SR / 32; ++kb) {
        // k-slot order of the transpose-read fragments: slots 0-3 <-> row 32 kb + 4G + j, slots 4-7 <-> 32 kb + 16 + 4G + j
        const int4 la = *reinterpret_cast<const int4*>(s_lead + kb * 32 + g4);
        const int4 lb = *reinterpret_cast<const int4*>(s_lead + kb * 32 + 16 + g4);
        const int lk[8] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w};
        bf16x8 af[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int lead = wave * 32 + mt * 16 + l15;
#pragma unroll
            for (int i = 0; i < 8; ++i) af[mt][i] = lk[i] == lead ? one : zero;
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            bf16x8 bfr;
            *reinterpret_cast<uint2*>(&bfr) = enc_tr_read(Gs, LDG, kb * 32, ct * 16, lane);
            *(reinterpret_cast<uint2*>(&bfr) + 1) = enc_tr_read(Gs, LDG, kb * 32 + 16, ct * 16, lane);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) acc[mt][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[mt], bfr, acc[mt][ct], 0, 0, 0);
        }
    }
    // acc[mt][ct][r] = sum for leader 32 w + 16 mt + 4G + r, channel 16 ct + l15
    const float sqs = lab ? -1.0f : p.sq * dk.scale;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int lead = wave * 32 + mt * 16 + g4 + r;
            if (s_lead[lead] != lead) continue;   // not a leader (or padding)
            float* dst = p.d_item + (long)s_id[lead] * CF + coff + l15;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) atomicAdd(dst + ct * 16, sqs * acc[mt][ct][r]);
        }
    if (lab && blockIdx.y == 0 && tid < SR && s_lead[tid] == tid) atomicAdd(p.d_bias + s_id[tid] - 1, -s_cb[tid]);
}

constexpr int ENC_NCHUNK = 16;

}  // namespace

extern "C" long edgl_encode_bwd_workspace(int B, int T, int C) { (void)B; return 2L * ENC_NCHUNK * T * C; }

extern "C" int edgl_encode_fwd_ct(const int64_t* ids, const float* ts, const void* item_tab, const float* pos_tab,
                                  const float* mark_emb, const uint8_t* mark_table, const float* tscale, int B, int T,
                                  int C, int E, int I, int64_t mask_id, float time_scale, float drop_rate,
                                  const uint64_t* rng_state, uint32_t stream_id, void* x0, float* spans,
                                  uint8_t* marks, int dh_pad, int dh_true, int dtype, void* stream);
extern "C" int edgl_encode_fwd(const int64_t* ids, const float* ts, const void* item_tab, const float* pos_tab,
                               const float* mark_emb, const uint8_t* mark_table, const float* tscale, int B, int T,
                               int C, int E, int I, int64_t mask_id, float time_scale, float drop_rate,
                               const uint64_t* rng_state, uint32_t stream_id, void* x0, float* spans,
                               uint8_t* marks, int dtype, void* stream) {
    return edgl_encode_fwd_ct(ids, ts, item_tab, pos_tab, mark_emb, mark_table, tscale, B, T, C, E, I, mask_id, time_scale, drop_rate,
                              rng_state, stream_id, x0, spans, marks, 0, 0, dtype, stream);
}
// edgl_encode_fwd for a channel-padded model: channel c of each C-wide section is real iff c % dh_pad < dh_true (0, 0: none);
// sqrt(C) of coding.py:62-63 is the square root of the TRUE width C / dh_pad * dh_true, padded channels of the time code are 0.
extern "C" int edgl_encode_fwd_ct(const int64_t* ids, const float* ts, const void* item_tab, const float* pos_tab,
                                  const float* mark_emb, const uint8_t* mark_table, const float* tscale, int B, int T,
                                  int C, int E, int I, int64_t mask_id, float time_scale, float drop_rate,
                                  const uint64_t* rng_state, uint32_t stream_id, void* x0, float* spans,
                                  uint8_t* marks, int dh_pad, int dh_true, int dtype, void* stream) {
    EDGL_REQUIRE((dh_pad == 0 && dh_true == 0) || (dh_pad > 0 && dh_true > 0 && dh_true <= dh_pad && C % dh_pad == 0), EDGL_ERR_SHAPE,
                 "edgl_encode_fwd: padded-channel spec dh_pad=%d dh_true=%d does not fit C=%d", dh_pad, dh_true, C);
    EDGL_REQUIRE(ids && ts && item_tab && pos_tab && mark_emb && mark_table && tscale && x0 && spans && marks,
                 EDGL_ERR_NULL, "edgl_encode_fwd: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0 && C > 0 && (C % 8) == 0 && E >= 1 && I > 1, EDGL_ERR_SHAPE,
                 "edgl_encode_fwd: bad shape B=%d T=%d C=%d E=%d I=%d (C must be a multiple of 8)", B, T, C, E, I);
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_encode_fwd: dropout without rng_state");
    EncP p{ids, ts, item_tab, pos_tab, mark_emb, mark_table, tscale, B, T, C, E, I, mask_id, time_scale,
           drop_rate, rng_state, stream_id, x0, spans, marks, dh_pad, dh_true,
           sqrtf((float)(dh_pad ? C / dh_pad * dh_true : C))};
    const long total = (long)B * T * (C / 8);
    dim3 grid((unsigned)((total + 255) / 256));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_F32) hipLaunchKernelGGL((encode_fwd_kernel<float>), grid, dim3(256), 0, st, p);
    else if (dtype == EDGL_BF16) hipLaunchKernelGGL((encode_fwd_kernel<bf16>), grid, dim3(256), 0, st, p);
    else { edgl_set_error("edgl_encode_fwd: bad dtype %d", dtype); return EDGL_ERR_DTYPE; }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

// edgl_encode_fwd_ct + edgl_compact_scan_labels(labels [B * M]) + (tpp_desc != NULL) edgl_tpp_prep(masked_pos, labels, ts: the raw
// timestamps the encoder reads, mark_table) in ONE launch: the batch preparation as the first workgroups of the encoder's grid.
// Same results as the three calls, bit for bit.  tpp_desc needs E = 16, M <= 256, T <= 2048 (edgl_tpp_prep).
extern "C" int edgl_encode_fwd_prep(const int64_t* ids, const float* ts, const void* item_tab, const float* pos_tab,
                                    const float* mark_emb, const uint8_t* mark_table, const float* tscale, int B, int T,
                                    int C, int E, int I, int64_t mask_id, float time_scale, float drop_rate,
                                    const uint64_t* rng_state, uint32_t stream_id, void* x0, float* spans,
                                    uint8_t* marks, int dh_pad, int dh_true, const int64_t* labels, int M, int32_t* perm,
                                    int32_t* inv, int32_t* nvalid, int64_t* labels_c, const int64_t* masked_pos, void* tpp_desc,
                                    int dtype, void* stream) {
    EDGL_REQUIRE((dh_pad == 0 && dh_true == 0) || (dh_pad > 0 && dh_true > 0 && dh_true <= dh_pad && C % dh_pad == 0), EDGL_ERR_SHAPE,
                 "edgl_encode_fwd_prep: padded-channel spec dh_pad=%d dh_true=%d does not fit C=%d", dh_pad, dh_true, C);
    EDGL_REQUIRE(ids && ts && item_tab && pos_tab && mark_emb && mark_table && tscale && x0 && spans && marks,
                 EDGL_ERR_NULL, "edgl_encode_fwd_prep: null pointer");
    EDGL_REQUIRE(labels && perm && inv && nvalid && labels_c, EDGL_ERR_NULL, "edgl_encode_fwd_prep: null pointer (compaction)");
    EDGL_REQUIRE(B > 0 && T > 0 && C > 0 && (C % 8) == 0 && E >= 1 && I > 1 && M > 0, EDGL_ERR_SHAPE,
                 "edgl_encode_fwd_prep: bad shape B=%d T=%d C=%d E=%d I=%d M=%d (C must be a multiple of 8)", B, T, C, E, I, M);
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_encode_fwd_prep: dropout without rng_state");
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_encode_fwd_prep: bad dtype %d", dtype);
    if (tpp_desc) {
        EDGL_REQUIRE(masked_pos, EDGL_ERR_NULL, "edgl_encode_fwd_prep: slot data without masked positions");
        EDGL_REQUIRE(T <= 2048 && E == 16 && M <= 256, EDGL_ERR_SHAPE,
                     "edgl_encode_fwd_prep: slot data needs E = 16, M <= 256, T <= 2048 (B=%d T=%d E=%d M=%d)", B, T, E, M);
        EDGL_REQUIRE((((uintptr_t)mark_table | (uintptr_t)tpp_desc) & 15) == 0, EDGL_ERR_SHAPE,
                     "edgl_encode_fwd_prep: mark_table / tpp_desc must be 16-byte aligned");
    }
    EncP p{ids, ts, item_tab, pos_tab, mark_emb, mark_table, tscale, B, T, C, E, I, mask_id, time_scale,
           drop_rate, rng_state, stream_id, x0, spans, marks, dh_pad, dh_true,
           sqrtf((float)(dh_pad ? C / dh_pad * dh_true : C))};
    PrepP q{labels, B * M, perm, inv, nvalid, labels_c, masked_pos, ts, M, (char*)tpp_desc, 1 + (tpp_desc ? B : 0)};
    const long total = (long)B * T * (C / 8);
    dim3 grid((unsigned)((total + 255) / 256 + q.nprep));
    const size_t smem = tpp_desc ? (size_t)T * 16 + 2 * 256 * sizeof(int) : 2 * 16 * 4 * sizeof(int);     // (slot data >= the scan's counts)
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_F32) hipLaunchKernelGGL((encode_prep_kernel<float>), grid, dim3(256), smem, st, p, q);
    else hipLaunchKernelGGL((encode_prep_kernel<bf16>), grid, dim3(256), smem, st, p, q);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_encode_bwd_add(const int64_t* ids, const uint8_t* marks, const void* dx0, const void* add1, const void* add2, int B,
                                   int T, int C, int E, int I, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                                   float* d_item, float* d_pos, float* d_mark_emb, float* workspace, int dtype, void* stream);
extern "C" int edgl_encode_bwd_add_ct(const int64_t* ids, const uint8_t* marks, const void* dx0, const void* add1, const void* add2, int B,
                                      int T, int C, int E, int I, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                                      float* d_item, float* d_pos, float* d_mark_emb, float* workspace, int c_true, int dtype, void* stream);
extern "C" int edgl_encode_bwd(const int64_t* ids, const uint8_t* marks, const void* dx0, int B, int T, int C,
                               int E, int I, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                               float* d_item, float* d_pos, float* d_mark_emb, float* workspace, int dtype,
                               void* stream) {
    return edgl_encode_bwd_add(ids, marks, dx0, nullptr, nullptr, B, T, C, E, I, drop_rate, rng_state, stream_id, d_item, d_pos,
                               d_mark_emb, workspace, dtype, stream);
}
extern "C" int edgl_encode_bwd_add(const int64_t* ids, const uint8_t* marks, const void* dx0, const void* add1, const void* add2, int B,
                                   int T, int C, int E, int I, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                                   float* d_item, float* d_pos, float* d_mark_emb, float* workspace, int dtype, void* stream) {
    return edgl_encode_bwd_add_ct(ids, marks, dx0, add1, add2, B, T, C, E, I, drop_rate, rng_state, stream_id, d_item, d_pos, d_mark_emb,
                                  workspace, 0, dtype, stream);
}
// edgl_encode_bwd_add for a channel-padded model: c_true (0: = C) is the true model width whose square root scales the item
// gradient (coding.py:62-63); padded channels of dX0 are exact zeros by construction and need no masking here.
// The MFMA scatter launch can carry the one-hot term of the tied table's scoring gradient as a second job (EncBwdP::lab_*): bf16,
// C a multiple of 128 (or 64), 8-byte aligned operands
extern "C" int edgl_encode_bwd_label_fused(int C, int dtype) { return dtype == EDGL_BF16 && (C % 128 == 0 || C == 64) ? 1 : 0; }
static int encode_bwd_impl(const int64_t* ids, const uint8_t* marks, const void* dx0, const void* add1, const void* add2, int B,
                           int T, int C, int E, int I, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                           float* d_item, float* d_pos, float* d_mark_emb, float* workspace, int c_true, const void* lab_rows,
                           const int64_t* lab_ids, const float* lab_coef, const int32_t* lab_nvalid, int lab_R, float* d_bias,
                           int dtype, void* stream);
extern "C" int edgl_encode_bwd_add_ct(const int64_t* ids, const uint8_t* marks, const void* dx0, const void* add1, const void* add2, int B,
                                      int T, int C, int E, int I, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                                      float* d_item, float* d_pos, float* d_mark_emb, float* workspace, int c_true, int dtype, void* stream) {
    return encode_bwd_impl(ids, marks, dx0, add1, add2, B, T, C, E, I, drop_rate, rng_state, stream_id, d_item, d_pos, d_mark_emb, workspace,
                           c_true, nullptr, nullptr, nullptr, nullptr, 0, nullptr, dtype, stream);
}
// edgl_encode_bwd_add_ct that ALSO applies the one-hot term of the scoring gradient which edgl_score_flash_bwd_ex(defer_label_term = 1)
// left out — d_item[label[r]] -= coef[r] rows[r], d_bias[label[r] - 1] -= coef[r] over the first min(R, nvalid) compacted rows
// (EasyDGL.py:177-185 / SURVEY Appendix C: dl = coef (p - onehot)) — as extra blocks of the embedding scatter's MFMA launch: the
// same segmented sum over equal ids, no launch and no stream fork of its own (edgl_score_flash_label_term is the standalone form).
// Requires edgl_encode_bwd_label_fused(C, dtype).
extern "C" int edgl_encode_bwd_add_label(const int64_t* ids, const uint8_t* marks, const void* dx0, const void* add1, const void* add2,
                                         int B, int T, int C, int E, int I, float drop_rate, const uint64_t* rng_state,
                                         uint32_t stream_id, float* d_item, float* d_pos, float* d_mark_emb, float* workspace,
                                         int c_true, const void* lab_rows, const int64_t* lab_ids, const float* lab_coef,
                                         const int32_t* lab_nvalid, int lab_R, float* d_bias, int dtype, void* stream) {
    EDGL_REQUIRE(lab_rows && lab_ids && lab_coef && d_bias && lab_R > 0, EDGL_ERR_NULL, "edgl_encode_bwd_add_label: null label operands");
    EDGL_REQUIRE(edgl_encode_bwd_label_fused(C, dtype) && (((uintptr_t)lab_rows | (uintptr_t)dx0 | (uintptr_t)add1 | (uintptr_t)add2) & 7) == 0,
                 EDGL_ERR_SHAPE, "edgl_encode_bwd_add_label: needs bf16, C %% 128 == 0 (or 64) and 8-byte aligned operands (C=%d)", C);
    return encode_bwd_impl(ids, marks, dx0, add1, add2, B, T, C, E, I, drop_rate, rng_state, stream_id, d_item, d_pos, d_mark_emb, workspace,
                           c_true, lab_rows, lab_ids, lab_coef, lab_nvalid, lab_R, d_bias, dtype, stream);
}
static int encode_bwd_impl(const int64_t* ids, const uint8_t* marks, const void* dx0, const void* add1, const void* add2, int B,
                           int T, int C, int E, int I, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                           float* d_item, float* d_pos, float* d_mark_emb, float* workspace, int c_true, const void* lab_rows,
                           const int64_t* lab_ids, const float* lab_coef, const int32_t* lab_nvalid, int lab_R, float* d_bias,
                           int dtype, void* stream) {
    EDGL_REQUIRE(c_true >= 0 && c_true <= C, EDGL_ERR_SHAPE, "edgl_encode_bwd: true width %d exceeds C=%d", c_true, C);
    EDGL_REQUIRE((add1 == nullptr) == (add2 == nullptr), EDGL_ERR_NULL, "edgl_encode_bwd_add: add1 / add2 go together");
    EDGL_REQUIRE(ids && marks && dx0 && d_item && d_pos && d_mark_emb && workspace, EDGL_ERR_NULL,
                 "edgl_encode_bwd: null pointer");
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_encode_bwd: bad dtype %d", dtype);
    EDGL_REQUIRE(C % 4 == 0 && C / 4 <= 256, EDGL_ERR_SHAPE, "edgl_encode_bwd: C=%d unsupported", C);
    float* part_pos = workspace;
    float* part_mk = workspace + (long)ENC_NCHUNK * T * C;
    int srows = SROWS;
    while (srows > 8 && (size_t)srows * C * sizeof(float) > 150 * 1024) srows >>= 1;   // C = 512: 64 rows per block
    EncBwdP p{ids, marks, dx0, B, T, C, E, I, drop_rate, rng_state, stream_id, d_item, part_pos, part_mk, ENC_NCHUNK, srows, add1, add2,
              sqrtf((float)(c_true > 0 ? c_true : C)), lab_rows, lab_ids, lab_coef, lab_nvalid, lab_R, 0, d_bias, d_mark_emb};
    hipStream_t st = (hipStream_t)stream;
    const int rows_par = 256 / (C / 4);
    dim3 grid(T, ENC_NCHUNK);
    const size_t smem = (size_t)rows_par * 2 * C * sizeof(float);
    if (dtype == EDGL_F32) hipLaunchKernelGGL((encode_bwd_kernel<float>), grid, dim3(256), smem, st, p);
    else hipLaunchKernelGGL((encode_bwd_kernel<bf16>), grid, dim3(256), smem, st, p);
    EDGL_LAUNCH_CHECK();
    {
        const size_t smem_s = (size_t)srows * C * sizeof(float);
        EDGL_REQUIRE(smem_s <= 150 * 1024, EDGL_ERR_SHAPE, "edgl_encode_bwd: C=%d too large for the scatter stage", C);
        const unsigned nb = (unsigned)(((long)B * T + srows - 1) / srows);
        if (dtype == EDGL_BF16 && (C % 128 == 0 || C == 64) && (((uintptr_t)dx0 | (uintptr_t)add1 | (uintptr_t)add2) & 7) == 0) {
            unsigned nbm = (unsigned)(((long)B * T + SROWS - 1) / SROWS);
            if (lab_rows) { p.lab_blk0 = (int)nbm; nbm += (unsigned)((lab_R + SROWS - 1) / SROWS); }    // label blocks behind the embedding's
            if (C % 128 == 0) hipLaunchKernelGGL((encode_scatter_mfma_kernel<8>), dim3(nbm, C / 128), dim3(256), 0, st, p);
            else hipLaunchKernelGGL((encode_scatter_mfma_kernel<4>), dim3(nbm), dim3(256), 0, st, p);
        } else if (dtype == EDGL_F32) {
            hipFuncSetAttribute((const void*)encode_scatter_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_s);
            hipLaunchKernelGGL((encode_scatter_kernel<float>), dim3(nb), dim3(256), smem_s, st, p);
        } else {
            hipFuncSetAttribute((const void*)encode_scatter_kernel<bf16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_s);
            hipLaunchKernelGGL((encode_scatter_kernel<bf16>), dim3(nb), dim3(256), smem_s, st, p);
        }
        EDGL_LAUNCH_CHECK();
    }
    int rc = edgl_reduce_rows(part_pos, ENC_NCHUNK, T * C, (long)T * C, d_pos, 0, st);
    if (rc) return rc;
    if (E > 1) {  // only row 1 of the mark-embedding table is ever indexed (EasyDGL.py:87-88)
        rc = edgl_reduce_rows(part_mk, ENC_NCHUNK * T, C, C, d_mark_emb + C, 0, st);
        if (rc) return rc;
    }
    return EDGL_OK;
}

// Small kernels around the hot path: error plumbing, dropout-RNG step, the TPP likelihood
// regulariser (K8: temporal.py:317-333 + EasyDGL.py:157-175), TF-form Adam over the flat
// parameter arena (Base.py:142-144), the l2 term (coding.py:34-40) and f32 -> dtype casts.
#include <cstdarg>
#include <cstdio>

#include "edgl_common.h"
#include "bimau_common.h"   // TppDesc / tpp_layout: slot data of the fused TPP form
#include "batch_prep.h"     // slot data of a sample (also run inside the encoder's launch)

namespace {
thread_local char g_err[512] = "";
constexpr int RED_BLOCKS = 256;   // per-block partials of the scalar reductions (one row per thread at the headline size)
constexpr int SUMSQ_BLOCKS = 1024;
}  // namespace

extern "C" void edgl_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
extern "C" const char* edgl_last_error(void) { return g_err; }

// ---- one-shot event bracket -----------------------------------------------------------------------------------
namespace {
struct ProfSlot { int id = -1; hipEvent_t e0 = nullptr, e1 = nullptr; };
thread_local ProfSlot g_prof;
}  // namespace
extern "C" int edgl_profile_next(int kernel_id, void* ev_start, void* ev_stop) {
    g_prof.id = kernel_id; g_prof.e0 = (hipEvent_t)ev_start; g_prof.e1 = (hipEvent_t)ev_stop;
    return EDGL_OK;
}
void edgl_prof_begin(int kernel_id, hipStream_t st) {
    if (g_prof.id == kernel_id && g_prof.e0) (void)hipEventRecord(g_prof.e0, st);
}
void edgl_prof_end(int kernel_id, hipStream_t st) {
    if (g_prof.id == kernel_id && g_prof.e1) { (void)hipEventRecord(g_prof.e1, st); g_prof.id = -1; }
}
extern "C" int edgl_version(void) { return 100; }

namespace {

__global__ void rng_advance_kernel(uint64_t* st) { st[1] += 1ull; }

// ---------------------------------------------------------------------------------------------
// K8 TPP regulariser
// ---------------------------------------------------------------------------------------------
struct TppP {
    const float* lam; const int64_t* mpos; const int64_t* labels; const float* ts; const uint8_t* mtab;
    int B, T, H, E, M; float coef;
};

using batch_prep::raw_span;   // EasyDGL.py:161-162 on RAW seconds

// per-row terms; returns (event_ll, non_event, n_marks); optionally the pieces needed by the backward
__device__ __forceinline__ void tpp_row(const TppP& p, long j, float& ev_ll, float& non_ev, float& nmk, float* ev_out,
                                        float* span_out, float* g_out, int* b_out, int* pos_out, int64_t* lab_out) {
    const long bp = j / p.M;
    const int m = (int)(j % p.M), b = (int)(bp % p.B);
    // masked-position mode (EasyDGL.py:157-175) or, with mpos == NULL, every position of the sequence (CTSMA.py:95-108:
    // M == T, ts holds T+1 raw timestamps per row and the interval is the plain forward difference)
    const int pos = p.mpos ? (int)p.mpos[(long)b * p.M + m] : m;
    const int64_t lab = p.labels[(long)b * p.M + m];
    const uint8_t* nm = p.mtab + lab * p.E;
    const float* lm = p.lam + (bp * p.T + pos) * p.E;
    float cnt = 0.f, ev = 0.f, ent = 0.f;
    for (int e = 0; e < p.E; ++e) { const float f = (float)nm[e]; cnt += f; ev += lm[e] * f; ent += lm[e]; }
    const float g = cnt > 0.f ? 1.f : 0.f;  // sign(sum nm), temporal.py:321
    ev *= g; ent *= g;
    const float sp = p.mpos ? raw_span(p.ts + (long)b * p.T, pos, p.T)
                            : p.ts[(long)b * (p.T + 1) + pos + 1] - p.ts[(long)b * (p.T + 1) + pos];
    ev_ll = __logf(ev == 0.f ? 1.f : ev);  // :324
    non_ev = ent * sp * 0.5f;              // :327-328
    nmk = cnt;
    if (ev_out) { *ev_out = ev; *span_out = sp; *g_out = g; *b_out = b; *pos_out = pos; *lab_out = lab; }
}

__global__ __launch_bounds__(256) void tpp_partial_kernel(TppP p, float* part) {
    __shared__ float red[8];
    const long n = (long)p.H * p.B * p.M;
    float a = 0.f, bsum = 0.f, c = 0.f;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long)gridDim.x * blockDim.x) {
        float e, ne, k;
        tpp_row(p, j, e, ne, k, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        a += e; bsum += ne; c += k;
    }
    a = block_sum(a, red); bsum = block_sum(bsum, red); c = block_sum(c, red);
    if (threadIdx.x == 0) { part[blockIdx.x * 3] = a; part[blockIdx.x * 3 + 1] = bsum; part[blockIdx.x * 3 + 2] = c; }
}
__global__ void tpp_final_kernel(const float* part, int nblk, float coef, float* sums, float* reg_out, int accumulate) {
    // one wave: lane i adds partials i, i+64, ... in index order, then a fixed xor tree — deterministic
    float a = 0.f, b = 0.f, c = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 64) { a += part[i * 3]; b += part[i * 3 + 1]; c += part[i * 3 + 2]; }
    a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
    if (threadIdx.x != 0) return;
    sums[0] = a; sums[1] = b; sums[2] = c;
    const float reg = coef * (-(a - b) / c);  // temporal.py:331-332, EasyDGL.py:175
    reg_out[0] = accumulate ? reg_out[0] + reg : reg;
}
__global__ __launch_bounds__(256) void tpp_bwd_kernel(TppP p, const float* sums, const float* gscale, float* d_lam) {
    const long n = (long)p.H * p.B * p.M;
    const float gs = gscale ? gscale[0] : 1.f;
    const float k = -gs * p.coef / sums[2];
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (long)gridDim.x * blockDim.x) {
        float e, ne, cnt, ev, sp, g; int b, pos; int64_t lab;
        tpp_row(p, j, e, ne, cnt, &ev, &sp, &g, &b, &pos, &lab);
        const long bp = j / p.M;
        const uint8_t* nm = p.mtab + lab * p.E;
        float* dst = d_lam + (bp * p.T + pos) * p.E;
        for (int q = 0; q < p.E; ++q) {
            const float dev = (ev != 0.f) ? (float)nm[q] / ev : 0.f;
            dst[q] = k * g * (dev - sp * 0.5f);
        }
    }
}

// Regulariser and its gradient for the training engine (edgl_tpp_fwd_bwd): d lambda is written for EVERY (b', position) row —
// zeros where the position is not masked, so the [H*B*T, E] gradient needs no memset, and the slots of a repeated masked
// position are summed (the gradient of tf.gather).  The normaliser c = sum of mark counts depends on the labels only and is
// computed first by one small workgroup (exact: integer counts), which removes the partial -> final -> gradient chain of
// launches.  (No last-workgroup-done reduction for the scalar sums: on this multi-XCD part every __threadfence() is an L2
// write-back — 2048 of them cost 50 us; a 64-thread final kernel costs 5.)
// one label per thread; integer atomics: exact and order-independent, hence deterministic
__global__ __launch_bounds__(256) void tpp_norm_kernel(TppP p, int* acc) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int cnt = 0;
    if (i < p.B * p.M) {
        const uint8_t* nm = p.mtab + p.labels[i] * p.E;
        if (p.E == 16 && ((uintptr_t)p.mtab & 15) == 0) {
            const uint4 w = *reinterpret_cast<const uint4*>(nm);
            const uint32_t ws[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) cnt += (int)((ws[j] & 0xffu) + ((ws[j] >> 8) & 0xffu) + ((ws[j] >> 16) & 0xffu) + (ws[j] >> 24));
        } else {
            for (int e = 0; e < p.E; ++e) cnt += nm[e];
        }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(acc, cnt);
}
// The same count by ONE workgroup (batches up to 64 K labels): strided labels per thread, integer block sum, a plain store — no
// memset launch in front of it and nothing stale to inherit (the memset of the atomic form was a 6 us launch of its own in the
// side-stream chain the first attention kernel waits for).
// (nz != NULL: also the number of labels != 0 — the weighted rows of the loss, EasyDGL.py:183-185 — for edgl_dp_counts)
__global__ __launch_bounds__(1024) void tpp_norm_one_kernel(TppP p, int* acc, int* nz) {
    __shared__ int red[16];
    int cnt = 0, cnz = 0;
    const int n = p.B * p.M;
    constexpr int NB = 8;   // labels per thread and round: all label loads of a round fly together, then all mark rows (two round
                            // trips per 8 K labels; one label -> one mark row at a time was a chain of 2 x 10 round trips: 20 us)
    for (int i0 = threadIdx.x; i0 < n; i0 += 1024 * NB) {
        int64_t lab[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) lab[j] = p.labels[min(i0 + j * 1024, n - 1)];
#pragma unroll
        for (int j = 0; j < NB; ++j) asm volatile("" : "+v"(lab[j]));
#pragma unroll
        for (int j = 0; j < NB; ++j) cnz += (i0 + j * 1024 < n && lab[j] != 0) ? 1 : 0;
        if (p.E == 16 && ((uintptr_t)p.mtab & 15) == 0) {
            uint4 w[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) w[j] = *reinterpret_cast<const uint4*>(p.mtab + lab[j] * 16);
#pragma unroll
            for (int j = 0; j < NB; ++j) asm volatile("" : "+v"(w[j].x), "+v"(w[j].y), "+v"(w[j].z), "+v"(w[j].w));
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (i0 + j * 1024 < n) {
                    const uint32_t ws[4] = {w[j].x, w[j].y, w[j].z, w[j].w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) cnt += (int)((ws[q] & 0xffu) + ((ws[q] >> 8) & 0xffu) + ((ws[q] >> 16) & 0xffu) + (ws[q] >> 24));
                }
            }
        } else {
            for (int j = 0; j < NB; ++j)
                if (i0 + j * 1024 < n) {
                    const uint8_t* nm = p.mtab + lab[j] * p.E;
                    for (int e = 0; e < p.E; ++e) cnt += nm[e];
                }
        }
    }
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) {
        int t = 0;
        for (int w = 0; w < 16; ++w) t += red[w];
        acc[0] = t;
    }
    if (nz) {
        __syncthreads();
        for (int o = 32; o > 0; o >>= 1) cnz += __shfl_xor(cnz, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnz;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int w = 0; w < 16; ++w) t += red[w];
            nz[0] = t;
        }
    }
}
constexpr int TPP_MAXM = 256, TPP_MAXT = 1024, TPP_FUSED_BLOCKS = 2048;   // (4096 one-sequence workgroups measured slower: 24 vs 19 us)
__global__ __launch_bounds__(128) void tpp_fused_kernel(TppP p, const float* sums, float* part, float* d_lam) {
    __shared__ float red[8];
    __shared__ int s_pos[TPP_MAXM];
    __shared__ int s_lab[TPP_MAXM];
    __shared__ int s_head[TPP_MAXT];
    __shared__ __attribute__((aligned(16))) float s_out[2][64 * 16];   // per wave: 64 gradient rows, for coalesced stores
    const float c = (float)reinterpret_cast<const int*>(sums)[4] * (float)p.H;   // tpp_norm_kernel
    const float k = -p.coef / c;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float a = 0.f, bsum = 0.f;
    for (int bp = blockIdx.x; bp < p.H * p.B; bp += gridDim.x) {
        const int b = bp % p.B;
        __syncthreads();
        for (int m = threadIdx.x; m < p.M; m += blockDim.x) {
            s_pos[m] = p.mpos ? (int)p.mpos[(long)b * p.M + m] : m;
            s_lab[m] = (int)p.labels[(long)b * p.M + m];
        }
        for (int t = threadIdx.x; t < p.T; t += blockDim.x) s_head[t] = 0x7fffffff;
        __syncthreads();
        // first slot of every position (integer atomicMin: order-independent); unmasked positions — two thirds of the rows —
        // then skip the slot search altogether, and further slots of a repeated position are searched from the first one on
        for (int m = threadIdx.x; m < p.M; m += blockDim.x) {
            const int pos = s_pos[m];
            if (pos >= 0 && pos < p.T) atomicMin(&s_head[pos], m);
        }
        __syncthreads();
        for (int t0 = 0; t0 < p.T; t0 += blockDim.x) {
            const int t = t0 + threadIdx.x;
            const long row = (long)bp * p.T + min(t, p.T - 1);
            float lam[16], gr[16];
            bool loaded = false;
#pragma unroll
            for (int e = 0; e < 16; ++e) gr[e] = 0.f;
            // slots of this position: the search is a cheap LDS scan; the heavy part below runs once per FOUND slot, so the
            // lanes of a wave do their first (usually only) slot together instead of one lane per loop iteration
            const int m_hi = t >= p.T ? 0 : p.M;
            int m = -1, nxt = t < p.T ? s_head[t] : 0x7fffffff;
            for (;;) {
                if (nxt >= m_hi) break;
                m = nxt;
                nxt = 0x7fffffff;
                if (p.mpos)                              // another slot of the same position (rare; none in all-position mode)
                    for (int q = m + 1; q < m_hi; ++q)
                        if (s_pos[q] == t) { nxt = q; break; }
                if (!loaded) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) { lam[e] = p.lam[row * p.E + min(e, p.E - 1)]; asm volatile("" : "+v"(lam[e])); }
                    loaded = true;
                }
                const uint8_t* nm = p.mtab + (long)s_lab[m] * p.E;
                // the E mark bytes of the label: one 16-byte load (E == 16), else E clamped byte loads — all unconditional
                // (a select around a load becomes a branch: dependent round trips)
                uint32_t mw[4];
                if (p.E == 16 && ((uintptr_t)p.mtab & 15) == 0) {
                    const uint4 w = *reinterpret_cast<const uint4*>(nm);
                    mw[0] = w.x; mw[1] = w.y; mw[2] = w.z; mw[3] = w.w;
                } else {
                    uint32_t by[16];
#pragma unroll
                    for (int e = 0; e < 16; ++e) { by[e] = nm[min(e, p.E - 1)]; asm volatile("" : "+v"(by[e])); }
#pragma unroll
                    for (int j = 0; j < 4; ++j) mw[j] = by[4 * j] | (by[4 * j + 1] << 8) | (by[4 * j + 2] << 16) | (by[4 * j + 3] << 24);
                }
                float cnt = 0.f, ev = 0.f, ent = 0.f, f[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    f[e] = e < p.E ? (float)((mw[e >> 2] >> (8 * (e & 3))) & 0xffu) : 0.f;
                    cnt += f[e]; ev += lam[e] * f[e]; ent += e < p.E ? lam[e] : 0.f;
                }
                const float g = cnt > 0.f ? 1.f : 0.f;  // sign(sum nm), temporal.py:321
                ev *= g; ent *= g;
                const float sp = p.mpos ? raw_span(p.ts + (long)b * p.T, t, p.T)
                                        : p.ts[(long)b * (p.T + 1) + t + 1] - p.ts[(long)b * (p.T + 1) + t];
                a += __logf(ev == 0.f ? 1.f : ev);      // :324
                bsum += ent * sp * 0.5f;                // :327-328
                const float iev = ev != 0.f ? 1.0f / ev : 0.f, kg = k * g, hs = sp * 0.5f;
#pragma unroll
                for (int e = 0; e < 16; ++e) gr[e] += kg * (f[e] * iev - hs);
            }
            if (d_lam) {
                if (p.E == 16) {
                    // 64 rows x 64 bytes of a wave are contiguous in d_lam: through LDS so that a store instruction
                    // writes 1 KB of consecutive bytes (lane = one 16-byte piece) instead of 64 scattered pieces
                    float* so = s_out[wave];
#pragma unroll
                    for (int e = 0; e < 16; e += 4)
                        *reinterpret_cast<float4*>(so + lane * 16 + e) = make_float4(gr[e], gr[e + 1], gr[e + 2], gr[e + 3]);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int row_w0 = t0 + wave * 64;                       // first position of this wave
                    const int nrows = min(64, p.T - row_w0);                 // <= 0: nothing to store
                    float* dst = d_lam + ((long)bp * p.T + row_w0) * 16;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int piece = i * 64 + lane;                     // 16-byte piece index in the wave's 4 KB
                        if (piece < nrows * 4) *reinterpret_cast<float4*>(dst + piece * 4) = *reinterpret_cast<const float4*>(so + piece * 4);
                    }
                    __builtin_amdgcn_wave_barrier();
                } else if (t < p.T) {
                    float* dst = d_lam + row * p.E;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        if (e < p.E) dst[e] = gr[e];
                }
            }
        }
    }
    a = block_sum(a, red); bsum = block_sum(bsum, red);
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = a; part[blockIdx.x * 2 + 1] = bsum; }
}
// The same regulariser and gradient with ONE THREAD PER MASKED SLOT (b', m), for a d lambda array that already holds zeros
// (edgl_bimau_fwd_zr fills it beside the lambda rows): only the rows of masked positions are written — 5 MB instead of the 26 MB
// of the headline step — and no position of the sequence is visited that carries no slot.  A position drawn into several slots
// (mask_random draws without replacement; padding rows can repeat position 0) is written once, by its FIRST slot, with the sum
// over its slots — the gradient of tf.gather; every slot adds its own term to the two loss sums.
__global__ __launch_bounds__(256) void tpp_rows_kernel(TppP p, const float* sums, float* part, float* d_lam) {
    __shared__ float red[8];
    __shared__ int s_pos[256], s_lab[256];
    const float c = (float)reinterpret_cast<const int*>(sums)[4] * (float)p.H;   // tpp_norm_kernel
    const float k = -p.coef / c;
    // M <= 256: a workgroup takes 256 / M whole (b', .) slot lists, positions and labels staged in LDS (the searches for the other
    // slots of a position are LDS reads; as global loads they were a chain of M dependent round trips: 17 us for 5 MB of work)
    const int G = p.M <= 256 ? 256 / p.M : 0;
    long bp; int m; bool active;
    if (G) {
        bp = (long)blockIdx.x * G + threadIdx.x / p.M; m = threadIdx.x % p.M;
        active = (int)threadIdx.x < G * p.M && bp < (long)p.H * p.B;
    } else {
        const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
        bp = j / p.M; m = (int)(j % p.M); active = j < (long)p.H * p.B * p.M;
    }
    const int b = (int)(bp % p.B);
    const int64_t* mp = (p.mpos && active) ? p.mpos + (long)b * p.M : nullptr;
    const int pos_in = !active ? 0 : (p.mpos ? (int)mp[m] : m);
    // a masked position outside [0, T) (malformed input, an uninitialised row after bind_batch): the slot is skipped, as the
    // fused kernel this one replaced did — never an out-of-bounds row of lambda / d lambda
    active = active && pos_in >= 0 && pos_in < p.T;
    const int pos = active ? pos_in : 0;
    const int lab = active ? (int)p.labels[(long)b * p.M + m] : 0;
    if (G) {
        s_pos[threadIdx.x] = active ? pos : -1 - (int)threadIdx.x;
        s_lab[threadIdx.x] = lab;
        __syncthreads();
    }
    float a = 0.f, bsum = 0.f;
    if (active) {
        const long row = bp * p.T + pos;
        float lam[16];
        if (p.E == 16) {
#pragma unroll
            for (int e = 0; e < 16; e += 4) {
                const float4 v = *reinterpret_cast<const float4*>(p.lam + row * 16 + e);
                lam[e] = v.x; lam[e + 1] = v.y; lam[e + 2] = v.z; lam[e + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) { lam[e] = p.lam[row * p.E + min(e, p.E - 1)]; asm volatile("" : "+v"(lam[e])); }
        }
        const float sp = p.mpos ? raw_span(p.ts + (long)b * p.T, pos, p.T)
                                : p.ts[(long)b * (p.T + 1) + pos + 1] - p.ts[(long)b * (p.T + 1) + pos];
        // other slots of this position: any earlier one (then this slot is not the writer), any later one (rare)
        const int base = G ? (int)threadIdx.x - m : 0;
        bool head = true, later = false;
        if (p.mpos)
            for (int q = 0; q < p.M; ++q) {
                const bool same = (G ? s_pos[base + q] : (int)mp[q]) == pos;
                head = head && !(same && q < m);
                later = later || (same && q > m);
            }
        float gr[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) gr[e] = 0.f;
        // this slot, then (first slot of a repeated position only) the later slots of the same position
        const int q_end = (head && later) ? p.M : m + 1;
        for (int q = m; q < q_end; ++q) {
            if (q != m && (G ? s_pos[base + q] : (int)mp[q]) != pos) continue;
            const int lq = q == m ? lab : (G ? s_lab[base + q] : (int)p.labels[(long)b * p.M + q]);
            const uint8_t* nm = p.mtab + (long)lq * p.E;
            uint32_t mw[4];
            if (p.E == 16 && ((uintptr_t)p.mtab & 15) == 0) {
                const uint4 w = *reinterpret_cast<const uint4*>(nm);
                mw[0] = w.x; mw[1] = w.y; mw[2] = w.z; mw[3] = w.w;
            } else {
                uint32_t by[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) { by[e] = nm[min(e, p.E - 1)]; asm volatile("" : "+v"(by[e])); }
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) mw[jj] = by[4 * jj] | (by[4 * jj + 1] << 8) | (by[4 * jj + 2] << 16) | (by[4 * jj + 3] << 24);
            }
            float cnt = 0.f, ev = 0.f, ent = 0.f, f[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                f[e] = e < p.E ? (float)((mw[e >> 2] >> (8 * (e & 3))) & 0xffu) : 0.f;
                cnt += f[e]; ev += lam[e] * f[e]; ent += e < p.E ? lam[e] : 0.f;
            }
            const float g = cnt > 0.f ? 1.f : 0.f;  // sign(sum nm), temporal.py:321
            ev *= g; ent *= g;
            if (q == m) {
                a = __logf(ev == 0.f ? 1.f : ev);       // :324
                bsum = ent * sp * 0.5f;                 // :327-328
            }
            const float iev = ev != 0.f ? 1.0f / ev : 0.f, kg = k * g, hs = sp * 0.5f;
#pragma unroll
            for (int e = 0; e < 16; ++e) gr[e] += kg * (f[e] * iev - hs);
        }
        if (head && d_lam) {
            float* dst = d_lam + row * p.E;
            if (p.E == 16) {
#pragma unroll
                for (int e = 0; e < 16; e += 4) *reinterpret_cast<float4*>(dst + e) = make_float4(gr[e], gr[e + 1], gr[e + 2], gr[e + 3]);
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (e < p.E) dst[e] = gr[e];
            }
        }
    }
    a = block_sum(a, red); bsum = block_sum(bsum, red);
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = a; part[blockIdx.x * 2 + 1] = bsum; }
}
// tpp_rows_kernel for ANY number of mark types (the static engine with more than 16 marks: a data set's mark.pkl fixes E,
// EasyDGL.py:45-46): same contract — d lambda pre-zeroed, one thread per masked slot, a repeated position written once by its
// first slot with the sum over its slots — as plain loops over the marks (not the benchmarked shape: no staging, no vector loads).
__global__ __launch_bounds__(256) void tpp_rows_wide_kernel(TppP p, const float* sums, float* part, float* d_lam) {
    __shared__ float red[8];
    const float c = (float)reinterpret_cast<const int*>(sums)[4] * (float)p.H;
    const float k = -p.coef / c;
    const long j = (long)blockIdx.x * blockDim.x + threadIdx.x;
    bool active = j < (long)p.H * p.B * p.M;
    const long bp = active ? j / p.M : 0;
    const int m = active ? (int)(j % p.M) : 0, b = (int)(bp % p.B);
    const int64_t* mp = p.mpos ? p.mpos + (long)b * p.M : nullptr;
    const int pos_in = !active ? 0 : (mp ? (int)mp[m] : m);
    active = active && pos_in >= 0 && pos_in < p.T;
    const int pos = active ? pos_in : 0;
    float a = 0.f, bsum = 0.f;
    if (active) {
        const long row = bp * p.T + pos;
        const float* lm = p.lam + row * p.E;
        const float sp = p.mpos ? raw_span(p.ts + (long)b * p.T, pos, p.T)
                                : p.ts[(long)b * (p.T + 1) + pos + 1] - p.ts[(long)b * (p.T + 1) + pos];
        bool head = true;
        if (mp)
            for (int q = 0; q < m; ++q) head = head && (int)mp[q] != pos;
        for (int q = m; q < (head ? p.M : m + 1); ++q) {
            if (q != m && (!mp || (int)mp[q] != pos)) continue;
            const uint8_t* nm = p.mtab + p.labels[(long)b * p.M + q] * p.E;
            float cnt = 0.f, ev = 0.f, ent = 0.f;
            for (int e = 0; e < p.E; ++e) { const float f = (float)nm[e]; cnt += f; ev += lm[e] * f; ent += lm[e]; }
            const float g = cnt > 0.f ? 1.f : 0.f;
            ev *= g; ent *= g;
            if (q == m) { a = __logf(ev == 0.f ? 1.f : ev); bsum = ent * sp * 0.5f; }
            if (head && d_lam) {
                const float iev = ev != 0.f ? 1.0f / ev : 0.f, kg = k * g, hs = sp * 0.5f;
                float* dst = d_lam + row * p.E;
                for (int e = 0; e < p.E; ++e) dst[e] += kg * ((float)nm[e] * iev - hs);   // this thread owns the row
            }
        }
    }
    a = block_sum(a, red); bsum = block_sum(bsum, red);
    if (threadIdx.x == 0) { part[blockIdx.x * 2] = a; part[blockIdx.x * 2 + 1] = bsum; }
}
// Slot data of the fused TPP form (bimau_common.h: TppDesc): one workgroup per sample, one thread per masked slot.  A slot is
// EFFECTIVE when its position lies inside the sequence and its label's mark row is not empty (sign(sum nm) = 1, temporal.py:321) —
// the others add nothing to the regulariser or its gradient.  The first effective slot of a position fills the per-position
// arrays, further ones go to the sample's overflow list in slot order.  Labels, positions and timestamps only: runs ahead of
// the step's forward on a side stream.  E = 16 (one 16-byte mark row per label), M <= 256.
// Also the regulariser's normaliser, the mark count of ALL labels of the batch (temporal.py:330; what edgl_tpp_norm computes), as
// per-sample counts cntp[B] — every workgroup has its sample's mark rows in registers anyway; the consumers (one wave per (b, head) in
// sweep 1, the final reduction) add the B numbers up themselves: no launch, no memset, no atomics for it.
__global__ __launch_bounds__(256) void tpp_prep_kernel(const int64_t* mpos, const int64_t* labels, const float* ts, const uint8_t* mtab,
                                                       int B, int T, int M, char* desc) {
    extern __shared__ __attribute__((aligned(16))) char tpp_smem[];
    batch_prep::tpp_prep_sample(mpos, labels, ts, mtab, B, T, M, desc, (int)blockIdx.x, tpp_smem);
}
__global__ __launch_bounds__(256) void tpp_final2_kernel(const float* part, int nblk, float coef, int H, float* sums, float* reg_out,
                                                         int accumulate) {
    // one workgroup: thread i adds partials i, i+256, ... in index order, then the fixed block_sum tree — deterministic
    __shared__ float red[8];
    float a = 0.f, b = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 256) { a += part[i * 2]; b += part[i * 2 + 1]; }
    a = block_sum(a, red); b = block_sum(b, red);
    if (threadIdx.x != 0) return;
    const float c = (float)reinterpret_cast<const int*>(sums)[4] * (float)H;
    sums[0] = a; sums[1] = b; sums[2] = c;
    const float reg = coef * (-(a - b) / c);  // temporal.py:331-332, EasyDGL.py:175
    reg_out[0] = accumulate ? reg_out[0] + reg : reg;
    reinterpret_cast<int*>(sums)[4] = 0;   // normaliser accumulator back to zero for the next call
}

// tpp_final2_kernel for the per-(b, head) partial sums of the fused form (thousands of pairs): 1024 threads, every thread's loads
// in flight together (the 256-thread loop above takes them one round trip at a time), the same fixed summation order per launch
// shape; the count slot is NOT cleared (sweep 1 of the same step reads it; edgl_tpp_norm stores it afresh every step).
__global__ __launch_bounds__(1024) void tpp_parts_kernel(const float* part, int nparts, float coef, int H, const int* cntp, int ncnt,
                                                         float* sums, float* reg_out, int accumulate) {
    __shared__ float red[16];
    __shared__ int redi[16];
    if (!cntp && ncnt < 0) {   // normaliser behind the partial sums, as edgl_bimau_bwd_tpp leaves it
        if (threadIdx.x == 0) reinterpret_cast<int*>(sums)[4] = reinterpret_cast<const int*>(part)[2 * nparts];
        __syncthreads();
    }
    if (cntp) {   // normaliser from the per-sample counts of edgl_tpp_prep (otherwise sums[4] holds it: edgl_tpp_norm, data parallel)
        int c = 0;
        for (int i = threadIdx.x; i < ncnt; i += 1024) c += cntp[i];
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if ((threadIdx.x & 63) == 0) redi[threadIdx.x >> 6] = c;
        __syncthreads();
        if (threadIdx.x == 0) {
            int t = 0;
            for (int i = 0; i < 16; ++i) t += redi[i];
            reinterpret_cast<int*>(sums)[4] = t;
        }
        __syncthreads();
    }
    float a = 0.f, b = 0.f;
    for (int i0 = threadIdx.x; i0 < nparts; i0 += 1024 * 8) {
        float2 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = reinterpret_cast<const float2*>(part)[min(i0 + j * 1024, nparts - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(v[j].x), "+v"(v[j].y));
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (i0 + j * 1024 < nparts) { a += v[j].x; b += v[j].y; }
    }
    a = block_sum(a, red); b = block_sum(b, red);
    if (threadIdx.x != 0) return;
    const float c = (float)reinterpret_cast<const int*>(sums)[4] * (float)H;
    sums[0] = a; sums[1] = b; sums[2] = c;
    const float reg = coef * (-(a - b) / c);  // temporal.py:331-332, EasyDGL.py:175
    reg_out[0] = accumulate ? reg_out[0] + reg : reg;
}

// ---------------------------------------------------------------------------------------------
// Adam (TF form) over the flat arena
// ---------------------------------------------------------------------------------------------
__global__ void adam_prepare_kernel(uint64_t* st, float lr, float b1, float b2) {
    st[0] += 1ull;
    const double t = (double)st[0];
    const float lr_t = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
    reinterpret_cast<float*>(st + 1)[0] = lr_t;
}
// l2_part (optional, [gridDim.x]): the block's sum of squares of the UPDATED parameters inside the l2 segments — the next step's L2
// loss term (coding.py:40 on the weights that step reads) needs no pass over the arena of its own (edgl_l2_from_parts)
__device__ __forceinline__ void advance_step_state(uint64_t* rng, uint64_t* adam, float lr, float b1, float b2) {
    rng[1] += 1ull;
    adam[0] += 1ull;
    const double t = (double)adam[0];
    reinterpret_cast<float*>(adam + 1)[0] = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
}
template <bool SHADOW>
__global__ __launch_bounds__(256) void adam_kernel(float* w, const float* g, float* m, float* v, long n, float b1,
                                                   float b2, float eps, const uint64_t* st, float l2,
                                                   const int64_t* seg, int nseg, bf16* shadow, float* l2_part,
                                                   uint64_t* rng_adv = nullptr, float lr_adv = 0.f, unsigned* ticket = nullptr) {
    __shared__ float red[8];
    const float lr_t = reinterpret_cast<const float*>(st + 1)[0];
    float sq = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i];
        const float wi = w[i];
        bool in_seg = false;
        if (l2 != 0.f)
            for (int s = 0; s < nseg; ++s)
                if (i >= seg[2 * s] && i < seg[2 * s + 1]) { gi += l2 * wi; in_seg = true; break; }
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float wn = wi - lr_t * mi / (sqrtf(vi) + eps);
        m[i] = mi; v[i] = vi; w[i] = wn;
        if (SHADOW) shadow[i] = (bf16)wn;
        sq = in_seg ? fmaf(wn, wn, sq) : sq;
    }
    if (l2_part) {
        sq = block_sum(sq, red);
        if (threadIdx.x == 0) l2_part[blockIdx.x] = sq;
    }
    if (ticket) {
        // edgl_adam_apply_l2p_next: the step counters of the NEXT step (step_begin_kernel's update) by the last workgroup to finish —
        // every workgroup has read this step's learning rate before it takes its ticket, so nobody reads what the update writes
        // Two levels of tickets (groups of 64 workgroups, then the groups): thousands of atomics on ONE word are served one after
        // the other by the L2 (measured: + 40 us behind an 18 us kernel with a single counter).
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned g = blockIdx.x >> 6, ng = (gridDim.x + 63) >> 6;
            const unsigned gsize = min(64u, gridDim.x - (g << 6));
            if (atomicAdd(ticket + 1 + g, 1u) == gsize - 1) {
                ticket[1 + g] = 0u;
                if (atomicAdd(ticket, 1u) == ng - 1) {
                    ticket[0] = 0u;
                    advance_step_state(rng_adv, const_cast<uint64_t*>(st), lr_adv, b1, b2);
                }
            }
        }
    }
}

// edgl_adam_apply_ex: the optimizer launch of the static engine's eager step.  Beside adam_kernel's update
//   * the gradient of arena elements [lo, hi) of up to two ranges is grad[i] + sum_s slabs[s * stride + (i - lo)] — the row-chunk
//     slabs of the tied table's / the output bias's scoring gradient, which slab_reduce_kernel otherwise sums in a launch of its
//     own between the scoring and the block-tail backward (the embedding scatter adds its rows into the zero-filled grad);
//   * thread 0 of workgroup 0 writes the step counters of the NEXT step (step_begin_kernel's update) into OTHER buffers
//     (rng_next, st_next): nobody reads them during this launch, nobody waits for a last workgroup (tickets measured + 15-40 us,
//     DESIGN.md rule 55), and the host swaps the two pairs of buffers behind the launch — no single-thread launch at the end of a step.
struct SlabSum { const float* slabs; long stride, lo, hi, zero_first; };
// sum of the slabs at arena element i of a range (0 outside it / in its zero_first head): element form, for range borders and
// ranges whose slab stride is not a multiple of 4 (the bias: I - 1 elements)
__device__ __forceinline__ float slab_sum1(const SlabSum& s, int nslab, long i) {
    if (!s.slabs || i < s.lo + s.zero_first || i >= s.hi) return 0.f;
    float a = 0.f;
    for (int k = 0; k < nslab; ++k) a += s.slabs[(long)k * s.stride + (i - s.lo)];
    return a;
}
template <bool SHADOW>
__global__ __launch_bounds__(256) void adam_ex_kernel(float* w, float* g, float* m, float* v, long n, float b1, float b2, float eps,
                                                      const uint64_t* st, float l2, const int64_t* seg, int nseg, bf16* shadow,
                                                      float* l2_part, SlabSum s0, SlabSum s1, int nslab, const uint64_t* rng_cur,
                                                      uint64_t* st_next, uint64_t* rng_next, float lr) {
    __shared__ float red[8];
    const float lr_t = reinterpret_cast<const float*>(st + 1)[0];
    if (st_next && blockIdx.x == 0 && threadIdx.x == 0) {
        rng_next[0] = rng_cur[0]; rng_next[1] = rng_cur[1] + 1ull;
        const uint64_t t1 = st[0] + 1ull;
        st_next[0] = t1;
        const double t = (double)t1;
        st_next[1] = 0ull;
        reinterpret_cast<float*>(st_next + 1)[0] = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
    }
    float sq = 0.f;
    // l2 segments of this launch (<= 4 kept in registers; more: the scalar tail handles everything)
    long slo[4] = {0, 0, 0, 0}, shi[4] = {0, 0, 0, 0};
    const bool seg_regs = l2 == 0.f || nseg <= 4;
    if (l2 != 0.f && seg_regs)
        for (int k = 0; k < 4; ++k) if (k < nseg) { slo[k] = seg[2 * k]; shi[k] = seg[2 * k + 1]; }
    const bool vec_a = s0.slabs && ((s0.lo | s0.hi | s0.zero_first | s0.stride) & 3) == 0 && ((uintptr_t)s0.slabs & 15) == 0 && nslab <= 4;
    const long n4 = seg_regs ? n / 4 : 0;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < n4; q += (long)gridDim.x * blockDim.x) {
        const long i = q * 4;
        const float4 g0 = *reinterpret_cast<const float4*>(g + i);
        float4 g4 = g0;
        const float4 w4 = *reinterpret_cast<const float4*>(w + i), m4 = *reinterpret_cast<const float4*>(m + i), v4 = *reinterpret_cast<const float4*>(v + i);
        const bool in_a = s0.slabs && i >= s0.lo && i < s0.hi, in_b = s1.slabs && i + 3 >= s1.lo && i < s1.hi;
        if (in_a && vec_a) {        // (all multiples of 4: a group lies inside the range or outside it)
            if (i >= s0.lo + s0.zero_first) {
                float4 x[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) x[k] = *reinterpret_cast<const float4*>(s0.slabs + (long)min(k, nslab - 1) * s0.stride + (i - s0.lo));
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (k < nslab) { g4.x += x[k].x; g4.y += x[k].y; g4.z += x[k].z; g4.w += x[k].w; }
            }
        } else if (in_a || in_b || (s0.slabs && i + 3 >= s0.lo && i < s0.hi)) {
            g4.x += slab_sum1(s0, nslab, i) + slab_sum1(s1, nslab, i); g4.y += slab_sum1(s0, nslab, i + 1) + slab_sum1(s1, nslab, i + 1);
            g4.z += slab_sum1(s0, nslab, i + 2) + slab_sum1(s1, nslab, i + 2); g4.w += slab_sum1(s0, nslab, i + 3) + slab_sum1(s1, nslab, i + 3);
        }
        // the partial gradients are consumed: the two ranges start the next step at zero (the embedding scatter / the one-hot term
        // add into them: no zero-fill launch)
        if (s0.slabs || s1.slabs) {
            float4 z4 = g0;
            bool any = false;
#define EDGL_ZR(c, off) if ((s0.slabs && i + off >= s0.lo && i + off < s0.hi) || (s1.slabs && i + off >= s1.lo && i + off < s1.hi)) { z4.c = 0.f; any = true; }
            EDGL_ZR(x, 0) EDGL_ZR(y, 1) EDGL_ZR(z, 2) EDGL_ZR(w, 3)
#undef EDGL_ZR
            if (any) *reinterpret_cast<float4*>(g + i) = z4;
        }
        float gi[4] = {g4.x, g4.y, g4.z, g4.w};
        const float wi[4] = {w4.x, w4.y, w4.z, w4.w}, mo[4] = {m4.x, m4.y, m4.z, m4.w}, vo[4] = {v4.x, v4.y, v4.z, v4.w};
        float wn[4], mn[4], vn[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bool in_seg = false;
            if (l2 != 0.f) {
#pragma unroll
                for (int k = 0; k < 4; ++k) in_seg = in_seg || (i + r >= slo[k] && i + r < shi[k]);
                gi[r] = in_seg ? gi[r] + l2 * wi[r] : gi[r];
            }
            mn[r] = b1 * mo[r] + (1.f - b1) * gi[r];
            vn[r] = b2 * vo[r] + (1.f - b2) * gi[r] * gi[r];
            wn[r] = wi[r] - lr_t * mn[r] / (sqrtf(vn[r]) + eps);
            sq = in_seg ? fmaf(wn[r], wn[r], sq) : sq;
        }
        *reinterpret_cast<float4*>(m + i) = make_float4(mn[0], mn[1], mn[2], mn[3]);
        *reinterpret_cast<float4*>(v + i) = make_float4(vn[0], vn[1], vn[2], vn[3]);
        *reinterpret_cast<float4*>(w + i) = make_float4(wn[0], wn[1], wn[2], wn[3]);
        if (SHADOW) {
            const Frag4<bf16> f = frag_from_acc<bf16>(f32x4{wn[0], wn[1], wn[2], wn[3]});
            *reinterpret_cast<uint2*>(shadow + i) = *reinterpret_cast<const uint2*>(&f);
        }
    }
    // scalar tail (n not a multiple of 4, or more than four l2 segments: everything)
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float gi = g[i] + slab_sum1(s0, nslab, i) + slab_sum1(s1, nslab, i);
        if ((s0.slabs && i >= s0.lo && i < s0.hi) || (s1.slabs && i >= s1.lo && i < s1.hi)) g[i] = 0.f;
        const float wi = w[i];
        bool in_seg = false;
        if (l2 != 0.f)
            for (int s = 0; s < nseg; ++s)
                if (i >= seg[2 * s] && i < seg[2 * s + 1]) { gi += l2 * wi; in_seg = true; break; }
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        const float wn = wi - lr_t * mi / (sqrtf(vi) + eps);
        m[i] = mi; v[i] = vi; w[i] = wn;
        if (SHADOW) shadow[i] = (bf16)wn;
        sq = in_seg ? fmaf(wn, wn, sq) : sq;
    }
    if (l2_part) {
        sq = block_sum(sq, red);
        if (threadIdx.x == 0) l2_part[blockIdx.x] = sq;
    }
}

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* w, const int64_t* seg, int nseg, float* part) {
    __shared__ float red[8];
    float a = 0.f;
    for (int s = 0; s < nseg; ++s) {
        const long lo = seg[2 * s], hi = seg[2 * s + 1];
        for (long i = lo + (long)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (long)gridDim.x * blockDim.x) a += w[i] * w[i];
    }
    a = block_sum(a, red);
    if (threadIdx.x == 0) part[blockIdx.x] = a;
}
__global__ void sumsq_final_kernel(const float* part, int nblk, float scale, float* out, int accumulate) {
    float a = 0.f;
    for (int i = threadIdx.x; i < nblk; i += 64) a += part[i];
    a = wave_sum(a);
    if (threadIdx.x == 0) out[0] = accumulate ? out[0] + scale * a : scale * a;
}

template <typename T>
__global__ void cast_kernel(const float* src, T* dst, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) dst[i] = from_f32<T>(src[i]);
}
template <typename T>
__global__ void cast_back_kernel(const T* src, float* dst, long n, int accumulate) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dst[i] = accumulate ? dst[i] + to_f32(src[i]) : to_f32(src[i]);
}
template <typename T>
__global__ void add_kernel(const T* a, const T* b, T* out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        out[i] = from_f32<T>(to_f32(a[i]) + to_f32(b[i]));
}
// dst[r, :ncols] += src[r, :ncols] with independent row strides (residual gradients into the first C channels)
template <typename T>
__global__ void add_cols_kernel(T* dst, int ld_dst, const T* src, const T* src2, int ld_src, long rows, int ncols) {
    const long total = rows * ncols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / ncols; const int c = (int)(i % ncols);
        float v = to_f32(dst[r * ld_dst + c]) + to_f32(src[r * ld_src + c]);
        if (src2) v += to_f32(src2[r * ld_src + c]);
        dst[r * ld_dst + c] = from_f32<T>(v);
    }
}
// 16-byte vectors along the row (ncols, ld_dst, ld_src multiples of the vector width, 16-byte aligned bases)
template <typename T>
__global__ void add_cols_vec_kernel(T* dst, int ld_dst, const T* src, const T* src2, int ld_src, long rows, int ncols) {
    constexpr int VEC = ElemTraits<T>::VEC;
    const int vpr = ncols / VEC;
    const long total = rows * vpr;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long r = i / vpr; const int c = (int)(i % vpr) * VEC;
        Vec16<T> d = ld16<T>(dst + r * ld_dst + c);
        const Vec16<T> a = ld16<T>(src + r * ld_src + c);
        Vec16<T> b = a;
        if (src2) b = ld16<T>(src2 + r * ld_src + c);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float v = to_f32(d.v[j]) + to_f32(a.v[j]);
            if (src2) v += to_f32(b.v[j]);
            d.v[j] = from_f32<T>(v);
        }
        st16<T>(dst + r * ld_dst + c, d);
    }
}

// 32 columns x 8 row lanes per block; each thread keeps 4 independent partial sums in flight
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* part, int P, int N, long ld, float* out, int accumulate) {
    __shared__ float sm[8][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n = blockIdx.x * 32 + tx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (n < N) {
        int p = ty;
        for (; p + 24 < P; p += 32) {
            a0 += part[(long)p * ld + n];
            a1 += part[(long)(p + 8) * ld + n];
            a2 += part[(long)(p + 16) * ld + n];
            a3 += part[(long)(p + 24) * ld + n];
        }
        for (; p < P; p += 8) a0 += part[(long)p * ld + n];
    }
    sm[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && n < N) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += sm[i][tx];
        out[n] = accumulate ? out[n] + s : s;
    }
}

template <typename T>
__global__ void gelu_bwd_kernel(const T* dy, const T* pre, T* dz, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dz[i] = from_f32<T>(to_f32(dy[i]) * dgelu_t<T>(to_f32(pre[i])));
}

template <typename T>
__global__ void relu_bwd_kernel(const T* dy, const T* y, T* dz, long n) {   // dz = dy * [y > 0]
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        dz[i] = to_f32(y[i]) > 0.f ? dy[i] : from_f32<T>(0.f);
}

inline int grid_for(long n) { return (int)std::min<long>((n + 255) / 256, 4096); }

}  // namespace

// ---- deferred reductions ------------------------------------------------------------------------------------------
// Every backward kernel that produces per-workgroup partials ends with one of these reductions; each is a few
// microseconds of work behind a full kernel launch.  In deferred mode (edgl_reduce_defer(1)) the calls are collected on
// the host and edgl_reduce_flush() runs all of them in ONE launch.  The caller must then keep every partial buffer
// alive (distinct workspaces) until the flush.  Jobs that accumulate into `out` flush first and run immediately, so the
// order of read-modify-write updates is preserved.
constexpr int RED_MAX_JOBS = 24;
constexpr int RED_COL_P = 32;      // jobs with at most this many partial rows run in the column form of the vector kernel
struct RedJob { const float* part; float* out; long ld; int P, N, blk0; };
struct RedBatch { RedJob j[RED_MAX_JOBS]; int n, blocks; };
static inline int red_blocks_scalar(const RedJob& j) { return (j.N + 31) / 32; }
static inline int red_blocks_vec(const RedJob& j) { return j.P <= RED_COL_P ? (j.N + 1023) / 1024 : (j.N + 31) / 32; }
thread_local bool g_red_defer = false;
thread_local RedBatch g_red_batch = {};

// 32 columns per workgroup.  Scalar form: one column per thread x RL partial-row lanes (RL = 8, or 32 when a list holds a deep job:
// the 1616 partial rows of the mark-embedding gradient were a chain of 50 dependent loads per thread with 8 lanes).
// Vector form (every job of the list has N, ld and both pointers multiples of four floats — the layouts of all producers here):
// 8 float4 column groups x 32 row lanes in 256 threads, eight loads in flight per thread: a 512-row job is two rounds of loads
// instead of sixteen (the scalar list of the headline step took 26 us for ~12 MB: latency, not bandwidth).
template <int RL>
__global__ __launch_bounds__(32 * RL) void reduce_rows_multi_kernel(RedBatch b) {
    __shared__ float sm[RL][33];
    int ji = 0;
    for (int i = 1; i < b.n; ++i)
        if ((int)blockIdx.x >= b.j[i].blk0) ji = i;
    const RedJob& jb = b.j[ji];
    const float* part = jb.part;
    const int P = jb.P, N = jb.N;
    const long ld = jb.ld;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int n = ((int)blockIdx.x - jb.blk0) * 32 + tx;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (n < N) {
        int p = ty;
        for (; p + 3 * RL < P; p += 4 * RL) {
            a0 += part[(long)p * ld + n];
            a1 += part[(long)(p + RL) * ld + n];
            a2 += part[(long)(p + 2 * RL) * ld + n];
            a3 += part[(long)(p + 3 * RL) * ld + n];
        }
        for (; p < P; p += RL) a0 += part[(long)p * ld + n];
    }
    sm[ty][tx] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (ty == 0 && n < N) {
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < RL; ++i) s += sm[i][tx];
        jb.out[n] = s;
    }
}

__global__ __launch_bounds__(256) void reduce_rows_multi_vec_kernel(RedBatch b) {
    __shared__ float4 sm[32][9];
    int ji = 0;
    for (int i = 1; i < b.n; ++i)
        if ((int)blockIdx.x >= b.j[i].blk0) ji = i;
    const RedJob& jb = b.j[ji];
    const int P = jb.P, N = jb.N;
    const long ld = jb.ld;
    if (P <= RED_COL_P) {
        // few partial rows (the row splits of a weight-gradient GEMM: 2-24 slabs of up to 3 M elements): a thread owns one
        // float4 column and adds its P values — 4 KB of consecutive bytes per row and workgroup, 32x fewer workgroups than the
        // row-lane form, whose 32 lanes per column would mostly idle (the 512-unit recipe's lists took 213 us for 143 MB)
        const int n = (((int)blockIdx.x - jb.blk0) * 256 + (int)threadIdx.x) * 4;
        if (n >= N) return;
        const float* base = jb.part + n;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int p0 = 0;
        for (; p0 + 8 <= P; p0 += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(base + (long)(p0 + u) * ld);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; p0 < P; ++p0) {
            const float4 v = *reinterpret_cast<const float4*>(base + (long)p0 * ld);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        *reinterpret_cast<float4*>(jb.out + n) = acc;
        return;
    }
    const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;          // float4 column group, row lane
    const int n = ((int)blockIdx.x - jb.blk0) * 32 + 4 * tx;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N) {
        const float* base = jb.part + n;
        int p = ty;
        for (; p + 7 * 32 < P; p += 8 * 32) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(base + (long)(p + 32 * u) * ld);
#pragma unroll
            for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
        for (; p < P; p += 32) {
            const float4 v = *reinterpret_cast<const float4*>(base + (long)p * ld);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    sm[ty][tx] = acc;
    __syncthreads();
    if (ty < 4 && n < N) {       // 32 threads finish the 32 columns: thread (ty, tx) takes component ty of group tx
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const float4 v = sm[i][tx];
            s += ty == 0 ? v.x : (ty == 1 ? v.y : (ty == 2 ? v.z : v.w));
        }
        jb.out[n + ty] = s;
    }
}

int edgl_reduce_flush_impl(hipStream_t st) {
    if (g_red_batch.n == 0) return EDGL_OK;
    int maxp = 0;
    bool vec = true;
    for (int i = 0; i < g_red_batch.n; ++i) {
        const RedJob& j = g_red_batch.j[i];
        maxp = std::max(maxp, j.P);
        vec = vec && (j.N & 3) == 0 && (j.ld & 3) == 0 && (((uintptr_t)j.part | (uintptr_t)j.out) & 15) == 0;
    }
    int blocks = 0;      // first-block index of every job for the kernel form chosen
    for (int i = 0; i < g_red_batch.n; ++i) {
        RedJob& j = g_red_batch.j[i];
        j.blk0 = blocks;
        blocks += vec ? red_blocks_vec(j) : red_blocks_scalar(j);
    }
    g_red_batch.blocks = blocks;
    if (vec) hipLaunchKernelGGL(reduce_rows_multi_vec_kernel, dim3(g_red_batch.blocks), dim3(256), 0, st, g_red_batch);
    else if (maxp > 1024) hipLaunchKernelGGL(reduce_rows_multi_kernel<32>, dim3(g_red_batch.blocks), dim3(1024), 0, st, g_red_batch);
    else hipLaunchKernelGGL(reduce_rows_multi_kernel<8>, dim3(g_red_batch.blocks), dim3(256), 0, st, g_red_batch);
    g_red_batch.n = 0;
    g_red_batch.blocks = 0;
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

int edgl_reduce_rows(const float* part, int P, int N, long ld, float* out, int accumulate, hipStream_t st) {
    if (g_red_defer && !accumulate) {
        if (g_red_batch.n == RED_MAX_JOBS) {
            const int rc = edgl_reduce_flush_impl(st);
            if (rc) return rc;
        }
        RedJob& j = g_red_batch.j[g_red_batch.n++];
        j.part = part; j.out = out; j.ld = ld; j.P = P; j.N = N; j.blk0 = g_red_batch.blocks;
        g_red_batch.blocks += (N + 31) / 32;
        return EDGL_OK;
    }
    if (g_red_defer) {   // read-modify-write job: everything queued before it must have landed
        const int rc = edgl_reduce_flush_impl(st);
        if (rc) return rc;
    }
    hipLaunchKernelGGL(reduce_rows_kernel, dim3((N + 31) / 32), dim3(256), 0, st, part, P, N, ld, out, accumulate);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_reduce_defer(int on, void* stream) {
    if (on < 0) {   // abort: forget the queued jobs (their partial buffers may be gone) and leave deferred mode
        g_red_batch.n = 0;
        g_red_batch.blocks = 0;
        g_red_defer = false;
        return EDGL_OK;
    }
    if (!on && g_red_defer) {
        const int rc = edgl_reduce_flush_impl((hipStream_t)stream);
        if (rc) return rc;
    }
    g_red_defer = on != 0;
    return EDGL_OK;
}

extern "C" int edgl_reduce_flush(void* stream) { return edgl_reduce_flush_impl((hipStream_t)stream); }

extern "C" int edgl_rng_advance(uint64_t* rng_state, void* stream) {
    EDGL_REQUIRE(rng_state, EDGL_ERR_NULL, "edgl_rng_advance: null state");
    hipLaunchKernelGGL(rng_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_tpp_workspace(void) { return std::max(RED_BLOCKS * 3 + 4, 8 + 2 * 4096); }   // edgl_tpp_fwd_bwd: sums[8] + partial pairs

extern "C" int edgl_tpp_fwd(const float* lam, const int64_t* masked_pos, const int64_t* labels, const float* ts_raw,
                            const uint8_t* mark_table, int B, int T, int H, int E, int M, float coef, float* sums,
                            float* reg_out, int accumulate, void* stream) {
    EDGL_REQUIRE(lam && labels && ts_raw && mark_table && sums && reg_out, EDGL_ERR_NULL, "edgl_tpp_fwd: null pointer");
    EDGL_REQUIRE(masked_pos || M == T, EDGL_ERR_SHAPE, "edgl_tpp_fwd: all-position mode needs M == T");
    TppP p{lam, masked_pos, labels, ts_raw, mark_table, B, T, H, E, M, coef};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(tpp_partial_kernel, dim3(RED_BLOCKS), dim3(256), 0, st, p, sums + 4);
    EDGL_LAUNCH_CHECK();
    hipLaunchKernelGGL(tpp_final_kernel, dim3(1), dim3(64), 0, st, sums + 4, RED_BLOCKS, coef, sums, reg_out, accumulate);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_tpp_bwd(const float* lam, const int64_t* masked_pos, const int64_t* labels, const float* ts_raw,
                            const uint8_t* mark_table, int B, int T, int H, int E, int M, float coef,
                            const float* sums, const float* gscale, float* d_lam, void* stream) {
    EDGL_REQUIRE(lam && labels && ts_raw && mark_table && sums && d_lam, EDGL_ERR_NULL, "edgl_tpp_bwd: null pointer");
    EDGL_REQUIRE(masked_pos || M == T, EDGL_ERR_SHAPE, "edgl_tpp_bwd: all-position mode needs M == T");
    TppP p{lam, masked_pos, labels, ts_raw, mark_table, B, T, H, E, M, coef};
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(d_lam, 0, (size_t)H * B * T * E * sizeof(float), st) != hipSuccess) {
        edgl_set_error("edgl_tpp_bwd: memset failed");
        return EDGL_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(tpp_bwd_kernel, dim3(grid_for((long)H * B * M)), dim3(256), 0, st, p, sums, gscale, d_lam);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

// The two batch sums that normalise the loss, in one launch — counts[0] = labels != 0 (weighted rows, EasyDGL.py:183-185), counts[1] =
// marks of all labels (temporal.py:330, what edgl_tpp_norm puts into sums[4]): what a data-parallel step all-reduces before it
// starts (8 bytes).  B * M <= 65536.
extern "C" int edgl_dp_counts(const int64_t* labels, const uint8_t* mark_table, int B, int M, int E, int32_t* counts, void* stream) {
    EDGL_REQUIRE(labels && mark_table && counts, EDGL_ERR_NULL, "edgl_dp_counts: null pointer");
    EDGL_REQUIRE(B > 0 && M > 0 && E > 0 && (long)B * M <= 65536, EDGL_ERR_SHAPE, "edgl_dp_counts: bad shape B=%d M=%d E=%d (B * M <= 65536)",
                 B, M, E);
    TppP p{nullptr, nullptr, labels, nullptr, mark_table, B, 0, 0, E, M, 0.f};
    hipLaunchKernelGGL(tpp_norm_one_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p, counts + 1, counts);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
// normaliser of the regulariser: integer sum of the mark counts of the batch's labels into sums[4] (labels only: may run
// ahead of the step's forward, e.g. on a side stream)
extern "C" int edgl_tpp_norm(const int64_t* labels, const uint8_t* mark_table, int B, int M, int E, float* sums, void* stream) {
    EDGL_REQUIRE(labels && mark_table && sums, EDGL_ERR_NULL, "edgl_tpp_norm: null pointer");
    TppP p{nullptr, nullptr, labels, nullptr, mark_table, B, 0, 0, E, M, 0.f};
    if ((long)B * M <= 65536) {   // one workgroup, plain store
        hipLaunchKernelGGL(tpp_norm_one_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, p, reinterpret_cast<int*>(sums) + 4,
                           (int*)nullptr);
        EDGL_LAUNCH_CHECK();
        return EDGL_OK;
    }
    // the count is formed from zero every time: a step that aborted between this call and tpp_final2_kernel (which zeroes the slot
    // again) must not leave a stale count behind for the next one
    if (hipMemsetAsync(reinterpret_cast<int*>(sums) + 4, 0, sizeof(int), (hipStream_t)stream) != hipSuccess) {
        edgl_set_error("edgl_tpp_norm: memset failed");
        return EDGL_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(tpp_norm_kernel, dim3((B * M + 255) / 256), dim3(256), 0, (hipStream_t)stream, p, reinterpret_cast<int*>(sums) + 4);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
// with_norm = 0: edgl_tpp_norm has already run for this batch on the same `sums`
extern "C" int edgl_tpp_fwd_bwd_ex(const float* lam, const int64_t* masked_pos, const int64_t* labels, const float* ts_raw,
                                   const uint8_t* mark_table, int B, int T, int H, int E, int M, float coef, float* sums,
                                   float* reg_out, int accumulate, float* d_lam, int with_norm, void* stream) {
    EDGL_REQUIRE(lam && labels && ts_raw && mark_table && sums && reg_out, EDGL_ERR_NULL, "edgl_tpp_fwd_bwd: null pointer");
    EDGL_REQUIRE(masked_pos || M == T, EDGL_ERR_SHAPE, "edgl_tpp_fwd_bwd: all-position mode needs M == T");
    EDGL_REQUIRE(E >= 1 && E <= 16, EDGL_ERR_SHAPE, "edgl_tpp_fwd_bwd: E=%d (1..16)", E);
    EDGL_REQUIRE(M <= TPP_MAXM && T <= TPP_MAXT, EDGL_ERR_SHAPE, "edgl_tpp_fwd_bwd: M=%d > %d or T=%d > %d", M, TPP_MAXM, T, TPP_MAXT);
    if (with_norm) {
        const int rc = edgl_tpp_norm(labels, mark_table, B, M, E, sums, stream);
        if (rc) return rc;
    }
    TppP p{lam, masked_pos, labels, ts_raw, mark_table, B, T, H, E, M, coef};
    hipStream_t st = (hipStream_t)stream;
    const int nblk = std::min(TPP_FUSED_BLOCKS, H * B);
    hipLaunchKernelGGL(tpp_fused_kernel, dim3(nblk), dim3(128), 0, st, p, sums, sums + 8, d_lam);
    EDGL_LAUNCH_CHECK();
    hipLaunchKernelGGL(tpp_final2_kernel, dim3(1), dim3(256), 0, st, sums + 8, nblk, coef, H, sums, reg_out, accumulate);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
// edgl_tpp_fwd_bwd_ex (with_norm = 0) for a d_lam array that ALREADY HOLDS ZEROS (edgl_bimau_fwd_zr): one thread per masked
// slot, only the rows of masked positions are written.  `sums` needs edgl_tpp_rows_workspace(B, H, M) floats.
static long tpp_rows_blocks(int B, int H, int M) {   // (also an upper bound of the one-slot-per-thread grid of the wide form)
    return M <= 256 ? ((long)H * B + 256 / M - 1) / (256 / M) : ((long)H * B * M + 255) / 256;
}
extern "C" long edgl_tpp_rows_workspace(int B, int H, int M) {
    return (B > 0 && H > 0 && M > 0) ? 8 + 2 * std::max(tpp_rows_blocks(B, H, M), ((long)H * B * M + 255) / 256) : -1;
}
extern "C" int edgl_tpp_fwd_bwd_rows(const float* lam, const int64_t* masked_pos, const int64_t* labels, const float* ts_raw,
                                     const uint8_t* mark_table, int B, int T, int H, int E, int M, float coef, float* sums,
                                     float* reg_out, int accumulate, float* d_lam, void* stream) {
    EDGL_REQUIRE(lam && labels && ts_raw && mark_table && sums && reg_out, EDGL_ERR_NULL, "edgl_tpp_fwd_bwd_rows: null pointer");
    EDGL_REQUIRE(masked_pos || M == T, EDGL_ERR_SHAPE, "edgl_tpp_fwd_bwd_rows: all-position mode needs M == T");
    EDGL_REQUIRE(E >= 1 && E <= 256 && B > 0 && H > 0 && M > 0, EDGL_ERR_SHAPE, "edgl_tpp_fwd_bwd_rows: bad shape E=%d B=%d H=%d M=%d", E, B, H, M);
    TppP p{lam, masked_pos, labels, ts_raw, mark_table, B, T, H, E, M, coef};
    hipStream_t st = (hipStream_t)stream;
    int nblk = (int)tpp_rows_blocks(B, H, M);
    if (E <= 16) {
        hipLaunchKernelGGL(tpp_rows_kernel, dim3(nblk), dim3(256), 0, st, p, sums, sums + 8, d_lam);
    } else {   // more than 16 mark types: plain loops over the marks, one slot per thread
        nblk = (int)(((long)H * B * M + 255) / 256);
        hipLaunchKernelGGL(tpp_rows_wide_kernel, dim3(nblk), dim3(256), 0, st, p, sums, sums + 8, d_lam);
    }
    EDGL_LAUNCH_CHECK();
    hipLaunchKernelGGL(tpp_final2_kernel, dim3(1), dim3(256), 0, st, sums + 8, nblk, coef, H, sums, reg_out, accumulate);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
// ---- fused TPP form: the regulariser inside the attention kernels (bimau_common.h: TppDesc) -----------------------------------
extern "C" long edgl_tpp_prep_bytes(int B, int T, int M) {
    return (B > 0 && T > 0 && M > 0) ? (long)bimau::tpp_layout(B, T, M).bytes : -1;
}
// slot data of one batch for edgl_bimau_bwd_tpp: `desc` = edgl_tpp_prep_bytes(B, T, M) bytes, 16-byte aligned
extern "C" int edgl_tpp_prep(const int64_t* masked_pos, const int64_t* labels, const float* ts_raw, const uint8_t* mark_table,
                             int B, int T, int E, int M, void* desc, void* stream) {
    EDGL_REQUIRE(masked_pos && labels && ts_raw && mark_table && desc, EDGL_ERR_NULL, "edgl_tpp_prep: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0 && T <= 2048 && E == 16 && M > 0 && M <= 256, EDGL_ERR_SHAPE,
                 "edgl_tpp_prep: bad shape B=%d T=%d E=%d M=%d (E = 16, M <= 256, T <= 2048)", B, T, E, M);
    EDGL_REQUIRE((((uintptr_t)mark_table | (uintptr_t)desc) & 15) == 0, EDGL_ERR_SHAPE, "edgl_tpp_prep: mark_table / desc must be 16-byte aligned");
    const size_t smem = (size_t)T * 16 + 2 * 256 * sizeof(int);
    hipLaunchKernelGGL(tpp_prep_kernel, dim3(B), dim3(256), smem, (hipStream_t)stream, masked_pos, labels, ts_raw, mark_table, B, T, M,
                       (char*)desc);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
// reduction of the per-(b, head) partial sums edgl_bimau_bwd_tpp left in `part` ([nparts, 2]) into the regulariser
// reg_out (=|+=) coef * (-(sum log ev - sum non-event) / (count * H))  (temporal.py:331-332, EasyDGL.py:175); `sums` = the array
// edgl_tpp_norm wrote the batch's mark count into (sums[0..2] receive the two sums and the normaliser, as edgl_tpp_fwd_bwd leaves
// them).  The normaliser: tpp_desc != NULL — the per-sample counts of edgl_tpp_prep(B, T, M), whose total also goes to sums[4];
// tpp_desc == NULL — sums[4] as edgl_tpp_norm (or a data-parallel all-reduce) left it; the slot is not cleared.
extern "C" int edgl_tpp_finish_parts(const float* part, int nparts, float coef, int H, const void* tpp_desc, int B, int T, int M,
                                     float* sums, float* reg_out, int accumulate, void* stream) {
    EDGL_REQUIRE(part && sums && reg_out, EDGL_ERR_NULL, "edgl_tpp_finish_parts: null pointer");
    EDGL_REQUIRE(nparts > 0 && H > 0 && (!tpp_desc || (B > 0 && T > 0 && M > 0)), EDGL_ERR_SHAPE,
                 "edgl_tpp_finish_parts: bad shape nparts=%d H=%d B=%d T=%d M=%d", nparts, H, B, T, M);
    const int* cntp = tpp_desc ? reinterpret_cast<const int*>((const char*)tpp_desc + bimau::tpp_layout(B, T, M).off_cntp) : nullptr;
    hipLaunchKernelGGL(tpp_parts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, nparts, coef, H, cntp, B, sums, reg_out,
                       accumulate);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
// edgl_tpp_finish_parts with the normaliser edgl_bimau_bwd_tpp left behind the partial sums (part: [nparts + 1, 2]; the count it
// used — the per-sample counts' total or the data-parallel sums[4] — as an int in the last pair): reads nothing of the batch.
extern "C" int edgl_tpp_finish_parts_n(const float* part, int nparts, float coef, int H, float* sums, float* reg_out, int accumulate,
                                       void* stream) {
    EDGL_REQUIRE(part && sums && reg_out, EDGL_ERR_NULL, "edgl_tpp_finish_parts_n: null pointer");
    EDGL_REQUIRE(nparts > 0 && H > 0, EDGL_ERR_SHAPE, "edgl_tpp_finish_parts_n: bad shape nparts=%d H=%d", nparts, H);
    hipLaunchKernelGGL(tpp_parts_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, part, nparts, coef, H, (const int*)nullptr, -1, sums,
                       reg_out, accumulate);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
extern "C" int edgl_tpp_fwd_bwd(const float* lam, const int64_t* masked_pos, const int64_t* labels, const float* ts_raw,
                                const uint8_t* mark_table, int B, int T, int H, int E, int M, float coef, float* sums,
                                float* reg_out, int accumulate, float* d_lam, void* stream) {
    return edgl_tpp_fwd_bwd_ex(lam, masked_pos, labels, ts_raw, mark_table, B, T, H, E, M, coef, sums, reg_out, accumulate, d_lam, 1,
                               stream);
}

extern "C" int edgl_adam_step(float* param, const float* grad, float* m, float* v, long n, float lr, float beta1,
                              float beta2, float eps, uint64_t* step_state, float l2, const int64_t* seg, int nseg,
                              void* shadow, void* stream) {
    EDGL_REQUIRE(param && grad && m && v && step_state, EDGL_ERR_NULL, "edgl_adam_step: null pointer");
    EDGL_REQUIRE(l2 == 0.f || nseg == 0 || seg, EDGL_ERR_NULL, "edgl_adam_step: l2 without segments");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(adam_prepare_kernel, dim3(1), dim3(1), 0, st, step_state, lr, beta1, beta2);
    EDGL_LAUNCH_CHECK();
    if (shadow)
        hipLaunchKernelGGL((adam_kernel<true>), dim3(grid_for(n)), dim3(256), 0, st, param, grad, m, v, n, beta1, beta2, eps,
                           step_state, l2, seg, nseg, (bf16*)shadow, (float*)nullptr);
    else
        hipLaunchKernelGGL((adam_kernel<false>), dim3(grid_for(n)), dim3(256), 0, st, param, grad, m, v, n, beta1, beta2, eps,
                           step_state, l2, seg, nseg, (bf16*)nullptr, (float*)nullptr);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

// The two single-thread state updates of a training step in one launch (a launch boundary costs ~4 us on this device
// whatever the kernel does): dropout step counter += 1, Adam step += 1 and its bias-corrected learning rate.
// edgl_adam_apply is then the parameter update alone.  edgl_adam_step == the Adam half of edgl_step_begin + edgl_adam_apply.
__global__ void step_begin_kernel(uint64_t* rng, uint64_t* adam, float lr, float b1, float b2) {
    rng[1] += 1ull;
    adam[0] += 1ull;
    const double t = (double)adam[0];
    reinterpret_cast<float*>(adam + 1)[0] = (float)((double)lr * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
}
extern "C" int edgl_step_begin(uint64_t* rng_state, uint64_t* adam_state, float lr, float beta1, float beta2, void* stream) {
    EDGL_REQUIRE(rng_state && adam_state, EDGL_ERR_NULL, "edgl_step_begin: null state");
    hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rng_state, adam_state, lr, beta1, beta2);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
extern "C" int edgl_adam_apply(float* param, const float* grad, float* m, float* v, long n, float beta1, float beta2, float eps,
                               const uint64_t* step_state, float l2, const int64_t* seg, int nseg, void* shadow, void* stream) {
    EDGL_REQUIRE(param && grad && m && v && step_state, EDGL_ERR_NULL, "edgl_adam_apply: null pointer");
    EDGL_REQUIRE(l2 == 0.f || nseg == 0 || seg, EDGL_ERR_NULL, "edgl_adam_apply: l2 without segments");
    hipStream_t st = (hipStream_t)stream;
    if (shadow)
        hipLaunchKernelGGL((adam_kernel<true>), dim3(grid_for(n)), dim3(256), 0, st, param, grad, m, v, n, beta1, beta2, eps,
                           step_state, l2, seg, nseg, (bf16*)shadow, (float*)nullptr);
    else
        hipLaunchKernelGGL((adam_kernel<false>), dim3(grid_for(n)), dim3(256), 0, st, param, grad, m, v, n, beta1, beta2, eps,
                           step_state, l2, seg, nseg, (bf16*)nullptr, (float*)nullptr);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

// edgl_adam_apply that also leaves the per-block sums of squares of the updated parameters inside the l2 segments in l2_part
// (edgl_adam_l2_parts(n) floats): the next step's L2 loss term is then edgl_l2_from_parts — one tiny launch that reads nothing
// of the arena (which the optimizer at the end of that step rewrites).
extern "C" int edgl_adam_l2_parts(long n) { return n > 0 ? grid_for(n) : -1; }
extern "C" int edgl_adam_apply_l2p(float* param, const float* grad, float* m, float* v, long n, float beta1, float beta2, float eps,
                                   const uint64_t* step_state, float l2, const int64_t* seg, int nseg, void* shadow, float* l2_part,
                                   void* stream) {
    EDGL_REQUIRE(param && grad && m && v && step_state && l2_part, EDGL_ERR_NULL, "edgl_adam_apply_l2p: null pointer");
    EDGL_REQUIRE(nseg == 0 || seg, EDGL_ERR_NULL, "edgl_adam_apply_l2p: segments missing");
    hipStream_t st = (hipStream_t)stream;
    if (shadow)
        hipLaunchKernelGGL((adam_kernel<true>), dim3(grid_for(n)), dim3(256), 0, st, param, grad, m, v, n, beta1, beta2, eps,
                           step_state, l2, seg, nseg, (bf16*)shadow, l2_part);
    else
        hipLaunchKernelGGL((adam_kernel<false>), dim3(grid_for(n)), dim3(256), 0, st, param, grad, m, v, n, beta1, beta2, eps,
                           step_state, l2, seg, nseg, (bf16*)nullptr, l2_part);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
// ... and advances the dropout step counter, the Adam step and its bias-corrected learning rate for the NEXT step behind the
// update (edgl_step_begin's work, by the last workgroup to finish: no single-thread launch at the end of the step's chain).
// ticket: edgl_adam_next_tickets(n) zero-initialised uint32 owned by the caller (left at zero).
extern "C" int edgl_adam_next_tickets(long n) { return n > 0 ? 1 + (grid_for(n) + 63) / 64 : -1; }
extern "C" int edgl_adam_apply_l2p_next(float* param, const float* grad, float* m, float* v, long n, float beta1, float beta2, float eps,
                                        uint64_t* step_state, float l2, const int64_t* seg, int nseg, void* shadow, float* l2_part,
                                        uint64_t* rng_state, float lr, uint32_t* ticket, void* stream) {
    EDGL_REQUIRE(param && grad && m && v && step_state && rng_state && ticket, EDGL_ERR_NULL, "edgl_adam_apply_l2p_next: null pointer");
    EDGL_REQUIRE(nseg == 0 || seg, EDGL_ERR_NULL, "edgl_adam_apply_l2p_next: segments missing");
    hipStream_t st = (hipStream_t)stream;
    if (shadow)
        hipLaunchKernelGGL((adam_kernel<true>), dim3(grid_for(n)), dim3(256), 0, st, param, grad, m, v, n, beta1, beta2, eps,
                           (const uint64_t*)step_state, l2, seg, nseg, (bf16*)shadow, l2_part, rng_state, lr, (unsigned*)ticket);
    else
        hipLaunchKernelGGL((adam_kernel<false>), dim3(grid_for(n)), dim3(256), 0, st, param, grad, m, v, n, beta1, beta2, eps,
                           (const uint64_t*)step_state, l2, seg, nseg, (bf16*)nullptr, l2_part, rng_state, lr, (unsigned*)ticket);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
// The eager engine's optimizer launch (see adam_ex_kernel).  slabs_a / slabs_b (either may be NULL): `nslab` slabs of `stride_*` floats
// whose sum is added to grad[lo_* .. hi_*) (arena element offsets); the first `zero_first_a` elements of range a take no slabs
// (row 0 of the used table is the zero constant: coding.py:56-57).  rng_cur / step_next / rng_next (all three or none): the
// counters of the next step are WRITTEN to step_next (Adam step, learning rate) and rng_next (seed, dropout step) — buffers other
// than step_state / rng_cur, which this launch reads.  l2_part as edgl_adam_apply_l2p (may be NULL).  The two slab ranges of `grad` are
// ZEROED behind their use (the next step's embedding scatter / one-hot term add into them).  Base.py:142-144.
extern "C" int edgl_adam_apply_ex(float* param, float* grad, float* m, float* v, long n, float beta1, float beta2, float eps,
                                  const uint64_t* step_state, float l2, const int64_t* seg, int nseg, void* shadow, float* l2_part,
                                  const float* slabs_a, long stride_a, long lo_a, long hi_a, long zero_first_a,
                                  const float* slabs_b, long stride_b, long lo_b, long hi_b, int nslab,
                                  const uint64_t* rng_cur, uint64_t* step_next, uint64_t* rng_next, float lr, void* stream) {
    EDGL_REQUIRE(param && grad && m && v && step_state, EDGL_ERR_NULL, "edgl_adam_apply_ex: null pointer");
    EDGL_REQUIRE(l2 == 0.f || nseg == 0 || seg, EDGL_ERR_NULL, "edgl_adam_apply_ex: l2 without segments");
    EDGL_REQUIRE((!rng_cur && !step_next && !rng_next) || (rng_cur && step_next && rng_next && step_next != step_state && rng_next != rng_cur),
                 EDGL_ERR_NULL, "edgl_adam_apply_ex: the next step's counters go to buffers of their own (all three pointers, or none)");
    EDGL_REQUIRE(nslab >= 0 && (nslab == 0 || slabs_a || slabs_b) && (!slabs_a || (lo_a >= 0 && lo_a <= hi_a && hi_a <= n && hi_a - lo_a <= stride_a)) &&
                 (!slabs_b || (lo_b >= 0 && lo_b <= hi_b && hi_b <= n && hi_b - lo_b <= stride_b)), EDGL_ERR_SHAPE, "edgl_adam_apply_ex: bad slab ranges");
    const SlabSum s0{slabs_a, stride_a, slabs_a ? lo_a : 0, slabs_a ? hi_a : 0, zero_first_a};
    const SlabSum s1{slabs_b, stride_b, slabs_b ? lo_b : 0, slabs_b ? hi_b : 0, 0};
    hipStream_t st = (hipStream_t)stream;
    if (shadow)
        hipLaunchKernelGGL((adam_ex_kernel<true>), dim3(grid_for(n)), dim3(256), 0, st, param, grad, m, v, n, beta1, beta2, eps, step_state, l2,
                           seg, nseg, (bf16*)shadow, l2_part, s0, s1, nslab, rng_cur, step_next, rng_next, lr);
    else
        hipLaunchKernelGGL((adam_ex_kernel<false>), dim3(grid_for(n)), dim3(256), 0, st, param, grad, m, v, n, beta1, beta2, eps, step_state, l2,
                           seg, nseg, (bf16*)nullptr, l2_part, s0, s1, nslab, rng_cur, step_next, rng_next, lr);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
extern "C" int edgl_l2_from_parts(const float* l2_part, int nparts, float l2, float* out, int accumulate, void* stream) {
    EDGL_REQUIRE(l2_part && out, EDGL_ERR_NULL, "edgl_l2_from_parts: null pointer");
    EDGL_REQUIRE(nparts > 0, EDGL_ERR_SHAPE, "edgl_l2_from_parts: nparts=%d", nparts);
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, l2_part, nparts, 0.5f * l2, out, accumulate);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_l2_loss(const float* param, const int64_t* seg, int nseg, float l2, float* out, int accumulate,
                            float* workspace, void* stream) {
    EDGL_REQUIRE(param && seg && out && workspace, EDGL_ERR_NULL, "edgl_l2_loss: null pointer");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(SUMSQ_BLOCKS), dim3(256), 0, st, param, seg, nseg, workspace);
    EDGL_LAUNCH_CHECK();
    hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(64), 0, st, workspace, SUMSQ_BLOCKS, 0.5f * l2, out, accumulate);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_cast(const float* src, void* dst, long n, int dtype, void* stream) {
    EDGL_REQUIRE(src && dst, EDGL_ERR_NULL, "edgl_cast: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_BF16) hipLaunchKernelGGL((cast_kernel<bf16>), dim3(grid_for(n)), dim3(256), 0, st, src, (bf16*)dst, n);
    else if (dtype == EDGL_F32) hipLaunchKernelGGL((cast_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, src, (float*)dst, n);
    else { edgl_set_error("edgl_cast: bad dtype %d", dtype); return EDGL_ERR_DTYPE; }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_cast_back(const void* src, float* dst, long n, int accumulate, int dtype, void* stream) {
    EDGL_REQUIRE(src && dst, EDGL_ERR_NULL, "edgl_cast_back: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_BF16) hipLaunchKernelGGL((cast_back_kernel<bf16>), dim3(grid_for(n)), dim3(256), 0, st, (const bf16*)src, dst, n, accumulate);
    else if (dtype == EDGL_F32) hipLaunchKernelGGL((cast_back_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, (const float*)src, dst, n, accumulate);
    else { edgl_set_error("edgl_cast_back: bad dtype %d", dtype); return EDGL_ERR_DTYPE; }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_add(const void* a, const void* b, void* out, long n, int dtype, void* stream) {
    EDGL_REQUIRE(a && b && out, EDGL_ERR_NULL, "edgl_add: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_BF16) hipLaunchKernelGGL((add_kernel<bf16>), dim3(grid_for(n)), dim3(256), 0, st, (const bf16*)a, (const bf16*)b, (bf16*)out, n);
    else if (dtype == EDGL_F32) hipLaunchKernelGGL((add_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, n);
    else { edgl_set_error("edgl_add: bad dtype %d", dtype); return EDGL_ERR_DTYPE; }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_add_cols(void* dst, int ld_dst, const void* src, const void* src2, int ld_src, long rows, int ncols,
                             int dtype, void* stream) {
    EDGL_REQUIRE(dst && src, EDGL_ERR_NULL, "edgl_add_cols: null pointer");
    EDGL_REQUIRE(dtype == EDGL_BF16 || dtype == EDGL_F32, EDGL_ERR_DTYPE, "edgl_add_cols: bad dtype %d", dtype);
    hipStream_t st = (hipStream_t)stream;
    const int vec = dtype == EDGL_BF16 ? 8 : 4;
    const bool vok = ncols % vec == 0 && ld_dst % vec == 0 && ld_src % vec == 0 &&
                     (((uintptr_t)dst | (uintptr_t)src | (uintptr_t)src2) & 15) == 0;
    const long n = vok ? rows * (ncols / vec) : rows * ncols;
    if (dtype == EDGL_BF16) {
        if (vok) hipLaunchKernelGGL((add_cols_vec_kernel<bf16>), dim3(grid_for(n)), dim3(256), 0, st, (bf16*)dst, ld_dst, (const bf16*)src, (const bf16*)src2, ld_src, rows, ncols);
        else hipLaunchKernelGGL((add_cols_kernel<bf16>), dim3(grid_for(n)), dim3(256), 0, st, (bf16*)dst, ld_dst, (const bf16*)src, (const bf16*)src2, ld_src, rows, ncols);
    } else {
        if (vok) hipLaunchKernelGGL((add_cols_vec_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, (float*)dst, ld_dst, (const float*)src, (const float*)src2, ld_src, rows, ncols);
        else hipLaunchKernelGGL((add_cols_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, (float*)dst, ld_dst, (const float*)src, (const float*)src2, ld_src, rows, ncols);
    }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

template <typename T>
__global__ void dropout_state_kernel(const T* x, T* y, long n, const uint64_t* rng, uint32_t stream_id, float rate) {
    const DropKey dk = make_dropkey(rng, stream_id, rate);
    const long n4 = n >> 2;   // 4 elements per thread; the mask depends on the element index only
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < n4; g += (long)gridDim.x * blockDim.x) {
        const Frag4<T> v = frag_ld<T>(x + g * 4);
        Frag4<T> o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.v[j] = from_f32<T>(drop_apply(dk, (uint64_t)(g * 4 + j), to_f32(v.v[j])));
        if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(y + g * 4) = *reinterpret_cast<uint4*>(&o);
        else *reinterpret_cast<uint2*>(y + g * 4) = *reinterpret_cast<uint2*>(&o);
    }
    for (long i = n4 * 4 + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = from_f32<T>(drop_apply(dk, (uint64_t)i, to_f32(x[i])));
}

extern "C" int edgl_dropout(const void* x, void* y, long n, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                            int dtype, void* stream) {
    EDGL_REQUIRE(x && y && (drop_rate == 0.f || rng_state), EDGL_ERR_NULL, "edgl_dropout: null pointer");
    hipStream_t st = (hipStream_t)stream;
    EDGL_REQUIRE((((uintptr_t)x | (uintptr_t)y) & 15) == 0, EDGL_ERR_SHAPE, "edgl_dropout: buffers must be 16-byte aligned");
    const dim3 grid(grid_for((n + 3) / 4));
    if (dtype == EDGL_BF16) hipLaunchKernelGGL((dropout_state_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)x, (bf16*)y, n, rng_state, stream_id, drop_rate);
    else if (dtype == EDGL_F32) hipLaunchKernelGGL((dropout_state_kernel<float>), grid, dim3(256), 0, st, (const float*)x, (float*)y, n, rng_state, stream_id, drop_rate);
    else { edgl_set_error("edgl_dropout: bad dtype %d", dtype); return EDGL_ERR_DTYPE; }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

// FeedForward tail (Base.py:83-86 followed by `*= seqs_masks`, TGAT.py:70): out = (dropout(a) + b) * (ids != 0), 4 elements
// per thread; b and ids optional.  With b == NULL it is also the backward of the `a` branch (mask and dropout commute).
template <typename T>
__global__ __launch_bounds__(256) void ff_tail_kernel(const T* a, const T* b, const int64_t* ids, long rows, int C, T* out,
                                                      const uint64_t* rng, uint32_t stream_id, float rate) {
    const DropKey dk = make_dropkey(rng, stream_id, rate);
    const int cpr = C >> 2;
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < rows * cpr; g += (long)gridDim.x * blockDim.x) {
        const long row = g / cpr;
        const long e0 = g * 4;
        const Frag4<T> av = frag_ld<T>(a + e0);
        const Frag4<T> bv = b ? frag_ld<T>(b + e0) : frag_zero<T>();
        const float m = (!ids || ids[row] != 0) ? 1.f : 0.f;
        Frag4<T> o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o.v[j] = from_f32<T>((drop_apply(dk, (uint64_t)(e0 + j), to_f32(av.v[j])) + to_f32(bv.v[j])) * m);
        if constexpr (sizeof(T) == 4) *reinterpret_cast<uint4*>(out + e0) = *reinterpret_cast<uint4*>(&o);
        else *reinterpret_cast<uint2*>(out + e0) = *reinterpret_cast<uint2*>(&o);
    }
}

extern "C" int edgl_ff_tail(const void* a, const void* b, const int64_t* ids, long rows, int C, float drop_rate,
                            const uint64_t* rng_state, uint32_t stream_id, void* out, int dtype, void* stream) {
    EDGL_REQUIRE(a && out && (drop_rate == 0.f || rng_state), EDGL_ERR_NULL, "edgl_ff_tail: null pointer");
    EDGL_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, EDGL_ERR_SHAPE, "edgl_ff_tail: bad shape rows=%ld C=%d", rows, C);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(grid_for(rows * (C / 4)));
    if (dtype == EDGL_BF16) hipLaunchKernelGGL((ff_tail_kernel<bf16>), grid, dim3(256), 0, st, (const bf16*)a, (const bf16*)b, ids, rows, C, (bf16*)out, rng_state, stream_id, drop_rate);
    else if (dtype == EDGL_F32) hipLaunchKernelGGL((ff_tail_kernel<float>), grid, dim3(256), 0, st, (const float*)a, (const float*)b, ids, rows, C, (float*)out, rng_state, stream_id, drop_rate);
    else { edgl_set_error("edgl_ff_tail: bad dtype %d", dtype); return EDGL_ERR_DTYPE; }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_relu_bwd(const void* dy, const void* y, void* dz, long n, int dtype, void* stream) {
    EDGL_REQUIRE(dy && y && dz, EDGL_ERR_NULL, "edgl_relu_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_BF16) hipLaunchKernelGGL((relu_bwd_kernel<bf16>), dim3(grid_for(n)), dim3(256), 0, st, (const bf16*)dy, (const bf16*)y, (bf16*)dz, n);
    else if (dtype == EDGL_F32) hipLaunchKernelGGL((relu_bwd_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, (const float*)dy, (const float*)y, (float*)dz, n);
    else { edgl_set_error("edgl_relu_bwd: bad dtype %d", dtype); return EDGL_ERR_DTYPE; }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_gelu_bwd(const void* dy, const void* pre, void* dz, long n, int dtype, void* stream) {
    EDGL_REQUIRE(dy && pre && dz, EDGL_ERR_NULL, "edgl_gelu_bwd: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_BF16) hipLaunchKernelGGL((gelu_bwd_kernel<bf16>), dim3(grid_for(n)), dim3(256), 0, st, (const bf16*)dy, (const bf16*)pre, (bf16*)dz, n);
    else if (dtype == EDGL_F32) hipLaunchKernelGGL((gelu_bwd_kernel<float>), dim3(grid_for(n)), dim3(256), 0, st, (const float*)dy, (const float*)pre, (float*)dz, n);
    else { edgl_set_error("edgl_gelu_bwd: bad dtype %d", dtype); return EDGL_ERR_DTYPE; }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

// K5/K6/K7: tied-embedding scoring against the item table.
//   logits[r, n] = rows[r,:] . table[n,:] + bias[n]   (EasyDGL.py:149-150; table row 0 acts as zeros,
//   bias = concat([-1000], output_bias) — coding.py:56-57, Base.py:106-110)
// Training never materialises the [R, I] logits: the forward keeps an online (max, sum-exp) per row
// (EasyDGL.py:155 softmax), the backward recomputes logit tiles and feeds dl = coef*(p - onehot)
// straight back into MFMA as an operand:
//   score_fwd    : grid (row tiles, item chunks)  -> per-chunk (max, sumexp), label logit
//   score_bwd_dy : grid (row tiles, item chunks)  -> d_rows partial slabs   (contract over items)
//   score_bwd_dw : grid (item tiles)              -> d_table, d_bias        (contract over rows)
// 128x128 logit tiles, full-K (K = C <= 256) operand tiles resident in LDS, the streamed operand
// prefetched through registers.  Operands needed "contraction-major" for the second product are
// produced from the same LDS tile by one MFMA against the identity (register-layout transpose).
#include "gemm_tile.h"

namespace {
using namespace tile;

constexpr int CHUNK = 2560;  // items per forward/backward chunk (20 tiles of 128)

template <typename T, int CT>
struct FullTile {  // 128 rows x C (=16*CT) elements, LDS row stride LDC
    static constexpr int VEC = ElemTraits<T>::VEC;
    static constexpr int C = 16 * CT;
    static constexpr int LDC = C + VEC;
    static constexpr int CV = C / VEC;            // vectors per row
    static constexpr int PER = 128 * CV / NT;     // vectors per thread
    Vec16<T> reg[PER];
    __device__ __forceinline__ void load(const T* base, int row0, int rows_total, bool zero_row0) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int v = threadIdx.x + i * NT;
            const int row = v / CV, cv = v % CV;
            const int gr = row0 + row;
            reg[i] = (gr < rows_total && !(zero_row0 && gr == 0)) ? ld16<T>(base + (long)gr * C + cv * VEC) : zero16<T>();
        }
    }
    __device__ __forceinline__ void store(T* S) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int v = threadIdx.x + i * NT;
            const int row = v / CV, cv = v % CV;
            st16<T>(S + row * LDC + cv * VEC, reg[i]);
        }
    }
};

// logit tile: SWAP  -> acc[j][i] = L(first = item n, second = row m)
//             !SWAP -> acc[i][j] = L(first = row m, second = item n)
template <typename T, int CT, bool SWAP>
__device__ __forceinline__ void logit_tile(const T* As, const T* Bs, int wm, int wn, int lane, f32x4 (&acc)[4][4]) {
    constexpr int VEC = ElemTraits<T>::VEC, KB = ElemTraits<T>::KB, LDC = 16 * CT + VEC, NKB = 16 * CT / KB;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) {
        Vec16<T> af[4], bf[4];
        const int koff = kb * KB + (lane >> 4) * VEC;
#pragma unroll
        for (int i = 0; i < 4; ++i) af[i] = ld16<T>(As + (wm * 64 + i * 16 + (lane & 15)) * LDC + koff);
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = ld16<T>(Bs + (wn * 64 + j * 16 + (lane & 15)) * LDC + koff);
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                if constexpr (SWAP) acc[a][b] = mma_kblock(bf[a], af[b], acc[a][b]);
                else acc[a][b] = mma_kblock(af[a], bf[b], acc[a][b]);
            }
    }
}

struct ScoreP {
    const void* rows; const void* table; const float* out_bias; const int64_t* labels;
    int R, C, I, i0, i1;
    float* row_lse; float* label_logit; float* part;  // part [R][nchunk][2]
    float* logits;                                    // optional [R, i1-i0]
    int nchunk;
    // backward
    const float* coef; const float* gscale; void* d_rows; float* d_table; float* d_bias; float* slabs;
};

__device__ __forceinline__ float bias_of(const ScoreP& p, int n) { return n == 0 ? -1000.0f : p.out_bias[n - 1]; }

// ---------------------------------------------------------------------------------------------
// forward: online log-sum-exp per row over this block's item chunk
// ---------------------------------------------------------------------------------------------
template <typename T, int CT>
__global__ __launch_bounds__(NT) void score_fwd_kernel(ScoreP p) {
    constexpr int LDC = 16 * CT + ElemTraits<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);
    T* Bs = As + 128 * LDC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int m0 = blockIdx.x * 128;
    const int c_lo = p.i0 + blockIdx.y * CHUNK, c_hi = min(p.i1, c_lo + CHUNK);
    const T* rows = reinterpret_cast<const T*>(p.rows);
    const T* table = reinterpret_cast<const T*>(p.table);

    FullTile<T, CT> ft;
    ft.load(rows, m0, p.R, false);
    ft.store(As);
    ft.load(table, c_lo, c_hi, true);
    ft.store(Bs);
    __syncthreads();

    float rmax[4], rsum[4];
    int64_t lab[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        rmax[i] = -INFINITY; rsum[i] = 0.f;
        const int m = m0 + wm * 64 + i * 16 + l15;
        lab[i] = (m < p.R && p.labels) ? p.labels[m] : -1;
    }
    const int ntile = (c_hi - c_lo + 127) / 128;
    for (int it = 0; it < ntile; ++it) {
        const int n0 = c_lo + it * 128;
        const bool more = it + 1 < ntile;
        if (more) ft.load(table, n0 + 128, c_hi, true);
        f32x4 acc[4][4];
        logit_tile<T, CT, true>(As, Bs, wm, wn, lane, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + wm * 64 + i * 16 + l15;
            float v[16];
            float tmax = -INFINITY;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * 64 + j * 16 + g4 + r;
                    float x = -INFINITY;
                    if (n < c_hi) {
                        x = (n == 0) ? -1000.0f : acc[j][i][r] + p.out_bias[n - 1];
                        if (n == lab[i]) p.label_logit[m] = x;
                        if (p.logits && m < p.R) p.logits[(long)m * (p.i1 - p.i0) + (n - p.i0)] = x;
                    }
                    v[j * 4 + r] = x;
                    tmax = fmaxf(tmax, x);
                }
            const float nm = fmaxf(rmax[i], tmax);
            if (nm > -INFINITY) {
                float s = rsum[i] * __expf(rmax[i] - nm);
#pragma unroll
                for (int q = 0; q < 16; ++q) s += __expf(v[q] - nm);
                rsum[i] = s;
                rmax[i] = nm;
            }
        }
        __syncthreads();
        if (more) ft.store(Bs);
        __syncthreads();
    }
    // combine the 4 lane groups (same row, different columns) and the two wn waves
    float* red = reinterpret_cast<float*>(smem);  // [2 wn][128 rows][2] reuse As (all tiles done)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float mx = group_max4(rmax[i]);
        float s = (rmax[i] > -INFINITY) ? rsum[i] * __expf(rmax[i] - mx) : 0.f;
        s += __shfl_xor(s, 16, 64);
        s += __shfl_xor(s, 32, 64);
        if (lane < 16) {
            const int lr = wm * 64 + i * 16 + l15;
            red[(wn * 128 + lr) * 2] = mx;
            red[(wn * 128 + lr) * 2 + 1] = s;
        }
    }
    __syncthreads();
    if (tid < 128) {
        const int m = m0 + tid;
        if (m < p.R) {
            const float ma = red[tid * 2], sa = red[tid * 2 + 1], mb = red[(128 + tid) * 2], sb = red[(128 + tid) * 2 + 1];
            const float mx = fmaxf(ma, mb);
            float s = 0.f;
            if (ma > -INFINITY) s += sa * __expf(ma - mx);
            if (mb > -INFINITY) s += sb * __expf(mb - mx);
            p.part[((long)m * p.nchunk + blockIdx.y) * 2] = mx;
            p.part[((long)m * p.nchunk + blockIdx.y) * 2 + 1] = s;
        }
    }
}

__global__ void lse_combine_kernel(const float* part, int R, int nchunk, float* row_lse) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= R) return;
    float mx = -INFINITY;
    for (int c = 0; c < nchunk; ++c) mx = fmaxf(mx, part[((long)m * nchunk + c) * 2]);
    float s = 0.f;
    for (int c = 0; c < nchunk; ++c) {
        const float pm = part[((long)m * nchunk + c) * 2];
        if (pm > -INFINITY) s += part[((long)m * nchunk + c) * 2 + 1] * __expf(pm - mx);
    }
    row_lse[m] = mx + __logf(s);
}

// ---------------------------------------------------------------------------------------------
// backward 1: d_rows[m, :] = sum_n dl[m, n] table[n, :]    (block = row tile x item chunk)
// ---------------------------------------------------------------------------------------------
template <typename T, int CT>
__global__ __launch_bounds__(NT) void score_bwd_dy_kernel(ScoreP p) {
    constexpr int LDC = 16 * CT + ElemTraits<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);
    T* Bs = As + 128 * LDC;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int m0 = blockIdx.x * 128;
    const int c_lo = p.i0 + blockIdx.y * CHUNK, c_hi = min(p.i1, c_lo + CHUNK);
    const T* rows = reinterpret_cast<const T*>(p.rows);
    const T* table = reinterpret_cast<const T*>(p.table);
    const Frag4<T> ident = identity_frag<T>(lane);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    FullTile<T, CT> ft;
    ft.load(rows, m0, p.R, false);
    ft.store(As);
    ft.load(table, c_lo, c_hi, true);
    ft.store(Bs);
    __syncthreads();

    const float gs = p.gscale ? p.gscale[0] : 1.0f;
    float lse[4], cf[4];
    int64_t lab[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + l15;
        const bool ok = m < p.R;
        lse[i] = ok ? p.row_lse[m] : INFINITY;
        cf[i] = ok ? p.coef[m] * gs : 0.f;
        lab[i] = ok ? p.labels[m] : -1;
    }
    f32x4 dy[4][CT];  // [row tile i][channel tile ct], L(first = m, second = c)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dy[i][ct] = zero4;

    const int ntile = (c_hi - c_lo + 127) / 128;
    for (int it = 0; it < ntile; ++it) {
        const int n0 = c_lo + it * 128;
        const bool more = it + 1 < ntile;
        if (more) ft.load(table, n0 + 128, c_hi, true);
        f32x4 acc[4][4];
        logit_tile<T, CT, true>(As, Bs, wm, wn, lane, acc);  // acc[j][i] = L(first=n, second=m)
        Frag4<T> dl[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * 64 + j * 16 + g4 + r;
                    float d = 0.f;
                    if (n < c_hi) {
                        const float x = (n == 0) ? -1000.0f : acc[j][i][r] + p.out_bias[n - 1];
                        d = cf[i] * (__expf(x - lse[i]) - (n == lab[i] ? 1.0f : 0.0f));
                    }
                    dl[j][i].v[r] = from_f32<T>(d);
                }
            }
        // dy[m][c] += sum_n dl[m][n] table[n][c]: A = dl (A[m'=m][kk=n]), B = table tile transposed to L(n, c)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const Frag4<T> tf = frag_ld<T>(Bs + (wn * 64 + j * 16 + l15) * LDC + ct * 16 + g4);  // L(first=c, second=n)
                const Frag4<T> tT = frag_from_acc<T>(mma16(tf, ident, zero4));                        // L(first=n, second=c)
#pragma unroll
                for (int i = 0; i < 4; ++i) dy[i][ct] = mma16(dl[j][i], tT, dy[i][ct]);
            }
        __syncthreads();
        if (more) ft.store(Bs);
        __syncthreads();
    }
    // dy regs: reg r <-> m = ...+g4+r, lane l15 <-> c.  Combine the two wn waves through LDS, then write the slab.
    float* red = reinterpret_cast<float*>(smem);  // [2 wm][64 m][C] floats  (<= 2*64*256*4 = 128 KB worst; fits the tile area for CT<=8 ... checked on host)
    if (wn == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[((wm * 64 + i * 16 + g4 + r) * (16 * CT)) + ct * 16 + l15] = dy[i][ct][r];
    }
    __syncthreads();
    if (wn == 0) {
        float* slab = p.slabs + (long)blockIdx.y * p.R * (16 * CT);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + wm * 64 + i * 16 + g4 + r;
                if (m < p.R) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int c = ct * 16 + l15;
                        slab[(long)m * (16 * CT) + c] = dy[i][ct][r] + red[((wm * 64 + i * 16 + g4 + r) * (16 * CT)) + c];
                    }
                }
            }
    }
}

template <typename T>
__global__ void slab_reduce_kernel(const float* slabs, int nslab, long n, T* out) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        float a = 0.f;
        for (int s = 0; s < nslab; ++s) a += slabs[(long)s * n + i];
        out[i] = from_f32<T>(a);
    }
}

// ---------------------------------------------------------------------------------------------
// backward 2: d_table[n, :] = sum_m dl[m, n] rows[m, :],  d_bias[n-1] = sum_m dl[m, n]
//             (block = one item tile, loops over all row tiles)
// ---------------------------------------------------------------------------------------------
template <typename T, int CT>
__global__ __launch_bounds__(NT) void score_bwd_dw_kernel(ScoreP p) {
    constexpr int LDC = 16 * CT + ElemTraits<T>::VEC;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    T* As = reinterpret_cast<T*>(smem);
    T* Bs = As + 128 * LDC;
    float* rowinfo = reinterpret_cast<float*>(Bs + 128 * LDC);  // [128][2] (lse, coef) + labels as int
    int* rowlab = reinterpret_cast<int*>(rowinfo + 256);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int n0 = p.i0 + blockIdx.x * 128;
    const T* rows = reinterpret_cast<const T*>(p.rows);
    const T* table = reinterpret_cast<const T*>(p.table);
    const Frag4<T> ident = identity_frag<T>(lane);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const float gs = p.gscale ? p.gscale[0] : 1.0f;

    FullTile<T, CT> ft;
    ft.load(table, n0, p.i1, true);
    ft.store(Bs);
    ft.load(rows, 0, p.R, false);
    ft.store(As);
    if (tid < 128) {
        const bool ok = tid < p.R;
        rowinfo[tid * 2] = ok ? p.row_lse[tid] : INFINITY;
        rowinfo[tid * 2 + 1] = ok ? p.coef[tid] * gs : 0.f;
        rowlab[tid] = ok ? (int)p.labels[tid] : -1;
    }
    __syncthreads();

    float bj[4];  // bias of this lane's 4 item columns (j tiles), n = n0 + wn*64 + j*16 + l15
    float dbias[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + l15;
        bj[j] = (n < p.i1 && n > 0) ? p.out_bias[n - 1] : 0.f;
    }
    f32x4 dw[4][CT];  // [item tile j][channel tile ct], L(first = n, second = c)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) dw[j][ct] = zero4;

    const int ntile = (p.R + 127) / 128;
    for (int mt = 0; mt < ntile; ++mt) {
        const int m0 = mt * 128;
        const bool more = mt + 1 < ntile;
        if (more) ft.load(rows, m0 + 128, p.R, false);
        f32x4 acc[4][4];
        logit_tile<T, CT, false>(As, Bs, wm, wn, lane, acc);  // acc[i][j] = L(first=m, second=n)
        Frag4<T> dl[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int lr = wm * 64 + i * 16 + g4 + r;
                const float lse = rowinfo[lr * 2], cf = rowinfo[lr * 2 + 1];
                const int lab = rowlab[lr];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = n0 + wn * 64 + j * 16 + l15;
                    float d = 0.f;
                    if (n < p.i1) {
                        const float x = (n == 0) ? -1000.0f : acc[i][j][r] + bj[j];
                        d = cf * (__expf(x - lse) - (n == lab ? 1.0f : 0.0f));
                    }
                    dbias[j] += d;
                    dl[i][j].v[r] = from_f32<T>(d);
                }
            }
        // dw[n][c] += sum_m dl[m][n] rows[m][c]: A = dl (A[m'=n][kk=m]), B = rows tile transposed to L(m, c)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const Frag4<T> yf = frag_ld<T>(As + (wm * 64 + i * 16 + l15) * LDC + ct * 16 + g4);  // L(first=c, second=m)
                const Frag4<T> yT = frag_from_acc<T>(mma16(yf, ident, zero4));                        // L(first=m, second=c)
#pragma unroll
                for (int j = 0; j < 4; ++j) dw[j][ct] = mma16(dl[i][j], yT, dw[j][ct]);
            }
        __syncthreads();
        if (more) {
            ft.store(As);
            const int m = m0 + 128 + tid;
            if (tid < 128) {
                const bool ok = m < p.R;
                rowinfo[tid * 2] = ok ? p.row_lse[m] : INFINITY;
                rowinfo[tid * 2 + 1] = ok ? p.coef[m] * gs : 0.f;
                rowlab[tid] = ok ? (int)p.labels[m] : -1;
            }
        }
        __syncthreads();
    }
    // combine the two wm waves (same items, different rows) through LDS
    float* red = reinterpret_cast<float*>(smem);  // [2 wn][64 n][C] + [128] bias
    float* redb = red + 2 * 64 * 16 * CT;
#pragma unroll
    for (int j = 0; j < 4; ++j) dbias[j] = group_sum4(dbias[j]);
    if (wm == 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    red[((wn * 64 + j * 16 + g4 + r) * (16 * CT)) + ct * 16 + l15] = dw[j][ct][r];
            if (lane < 16) redb[wn * 64 + j * 16 + l15] = dbias[j];
        }
    }
    __syncthreads();
    if (wm == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + j * 16 + g4 + r;
                if (n < p.i1) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) {
                        const int c = ct * 16 + l15;
                        const float v = dw[j][ct][r] + red[((wn * 64 + j * 16 + g4 + r) * (16 * CT)) + c];
                        p.d_table[(long)n * (16 * CT) + c] = (n == 0) ? 0.f : v;
                    }
                }
            }
            if (lane < 16) {
                const int n = n0 + wn * 64 + j * 16 + l15;
                if (n < p.i1 && n > 0) p.d_bias[n - 1] = dbias[j] + redb[wn * 64 + j * 16 + l15];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// CE loss from (lse, label logit) — EasyDGL.py:155,177-185
// ---------------------------------------------------------------------------------------------
__global__ void ce_loss_kernel(const float* row_lse, const float* label_logit, const int64_t* labels, int R,
                               float* loss_out, float* coef) {
    __shared__ float red[8];
    float num = 0.f, den = 0.f;
    for (int m = threadIdx.x; m < R; m += blockDim.x) {
        const float w = labels[m] != 0 ? 1.f : 0.f;
        const float py = __expf(label_logit[m] - row_lse[m]);
        num += w * (-__logf(py + 1e-5f));
        den += w;
    }
    num = block_sum(num, red);
    den = block_sum(den, red);
    const float W = den + 1e-5f;
    if (threadIdx.x == 0) loss_out[0] = num / W;
    for (int m = threadIdx.x; m < R; m += blockDim.x) {
        const float w = labels[m] != 0 ? 1.f : 0.f;
        const float py = __expf(label_logit[m] - row_lse[m]);
        coef[m] = (w / W) * (py / (py + 1e-5f));
    }
}

// ---------------------------------------------------------------------------------------------
// K6: mask seen items + per-row top-K (Base.py:156-163,181); K7 merge; metrics (Base.py:181-201)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t float_key(float f) {  // monotone map float -> uint32 (larger = larger)
    uint32_t u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// one workgroup per row: 4-pass radix select of the K-th largest key, then ordered compaction
__global__ __launch_bounds__(256) void mask_topk_kernel(float* logits, int R, int n, int i0, const int64_t* seen,
                                                        int T, int K, float* out_val, int32_t* out_idx) {
    __shared__ uint32_t hist[256];
    __shared__ uint32_t sel_prefix, sel_remaining;
    __shared__ int cnt_gt, cnt_eq;
    __shared__ float cval[128];
    __shared__ int cidx[128];
    __shared__ int wave_eq[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float* x = logits + (long)row * n;
    if (seen) {
        for (int t = tid; t < T; t += blockDim.x) {
            const long id = seen[(long)row * T + t] - i0;
            if (id >= 0 && id < n) x[id] = -INFINITY;
        }
        __syncthreads();
    }
    const int Keff = min(K, n);
    uint32_t prefix = 0u, mask = 0u;
    int remaining = Keff;
    for (int pass = 3; pass >= 0; --pass) {
        hist[tid] = 0u;
        __syncthreads();
        for (int i = tid; i < n; i += blockDim.x) {
            const uint32_t k = float_key(x[i]);
            if ((k & mask) == prefix) atomicAdd(&hist[(k >> (pass * 8)) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {
            int acc = 0, b = 255;
            for (; b > 0; --b) {
                if (acc + (int)hist[b] >= remaining) break;
                acc += hist[b];
            }
            sel_prefix = prefix | ((uint32_t)b << (pass * 8));
            sel_remaining = remaining - acc;
        }
        __syncthreads();
        prefix = sel_prefix;
        remaining = sel_remaining;
        mask |= 255u << (pass * 8);
        __syncthreads();
    }
    // prefix = key of the K-th largest value; `remaining` of the elements equal to it are taken (lowest index first)
    if (tid == 0) { cnt_gt = 0; cnt_eq = 0; }
    for (int i = tid; i < 128; i += blockDim.x) { cval[i] = -INFINITY; cidx[i] = 0x7fffffff; }
    __syncthreads();
    const int n_gt = Keff - remaining;
    // elements strictly greater: any order (sorted afterwards); equal: need the `remaining` lowest indices
    for (int base = 0; base < n; base += blockDim.x) {
        const int i = base + tid;
        bool is_gt = false, is_eq = false;
        float v = 0.f;
        if (i < n) {
            v = x[i];
            const uint32_t k = float_key(v);
            is_gt = k > prefix; is_eq = k == prefix;
        }
        if (is_gt) {
            const int pos = atomicAdd(&cnt_gt, 1);
            cval[pos] = v; cidx[pos] = i;
        }
        // equal elements in index order: ballot-based ordered append within the block pass
        const unsigned long long bal = __ballot(is_eq);
        const int lane = tid & 63, w = tid >> 6;
        if (lane == 0) wave_eq[w] = __popcll(bal);
        __syncthreads();
        int offs = cnt_eq;
        for (int ww = 0; ww < w; ++ww) offs += wave_eq[ww];
        if (is_eq) {
            const int pos = offs + __popcll(bal & ((1ull << lane) - 1ull));
            if (pos < remaining) { cval[n_gt + pos] = v; cidx[n_gt + pos] = i; }
        }
        __syncthreads();
        if (tid == 0) cnt_eq += wave_eq[0] + wave_eq[1] + wave_eq[2] + wave_eq[3];
        __syncthreads();
    }
    // bitonic sort of 128 candidates by (value desc, index asc)
    for (int k = 2; k <= 128; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (tid < 128) {
                const int ixj = tid ^ j;
                if (ixj > tid) {
                    const float a = cval[tid], b = cval[ixj];
                    const int ia = cidx[tid], ib = cidx[ixj];
                    const bool a_first = (a > b) || (a == b && ia < ib);  // a should precede b
                    const bool up = (tid & k) == 0;
                    if (up ? !a_first : a_first) { cval[tid] = b; cval[ixj] = a; cidx[tid] = ib; cidx[ixj] = ia; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < K; i += blockDim.x) {
        out_val[(long)row * K + i] = i < Keff ? cval[i] : -INFINITY;
        out_idx[(long)row * K + i] = i < Keff ? cidx[i] + i0 : -1;
    }
}

// candidates [S][R][K] -> global top-K by (value desc, index asc); S*K <= 1024
__global__ __launch_bounds__(256) void topk_merge_kernel(const float* cand_val, const int32_t* cand_idx, int S, int R,
                                                         int K, float* out_val, int32_t* out_idx) {
    __shared__ float v[1024];
    __shared__ int ix[1024];
    const int row = blockIdx.x, tid = threadIdx.x, n = S * K;
    for (int i = tid; i < 1024; i += blockDim.x) {
        if (i < n) {
            const int s = i / K, k = i % K;
            const int id = cand_idx[((long)s * R + row) * K + k];
            v[i] = id < 0 ? -INFINITY : cand_val[((long)s * R + row) * K + k];
            ix[i] = id < 0 ? 0x7fffffff : id;
        } else { v[i] = -INFINITY; ix[i] = 0x7fffffff; }
    }
    __syncthreads();
    for (int k = 2; k <= 1024; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < 1024; t += blockDim.x) {
                const int ixj = t ^ j;
                if (ixj > t) {
                    const float a = v[t], b = v[ixj];
                    const int ia = ix[t], ib = ix[ixj];
                    const bool a_first = (a > b) || (a == b && ia < ib);
                    const bool up = (t & k) == 0;
                    if (up ? !a_first : a_first) { v[t] = b; v[ixj] = a; ix[t] = ib; ix[ixj] = ia; }
                }
            }
            __syncthreads();
        }
    for (int i = tid; i < K; i += blockDim.x) {
        out_val[(long)row * K + i] = v[i];
        out_idx[(long)row * K + i] = ix[i] == 0x7fffffff ? -1 : ix[i];
    }
}

__global__ void rank_metrics_kernel(const int32_t* topk, int R, int K, const int64_t* label, float* metrics) {
    __shared__ float red[8];
    float h10 = 0, h50 = 0, h100 = 0, n10 = 0, n50 = 0, n100 = 0;
    for (int r = threadIdx.x; r < R; r += blockDim.x) {
        const int lab = (int)label[r];
        int rank = -1;
        for (int k = 0; k < K && k < 100; ++k)
            if (topk[(long)r * K + k] == lab) { rank = k; break; }
        if (rank >= 0) {
            const float gain = 1.0f / log2f((float)rank + 2.0f);
            if (rank < 10) { h10 += 1; n10 += gain; }
            if (rank < 50) { h50 += 1; n50 += gain; }
            h100 += 1; n100 += gain;
        }
    }
    float vals[6] = {h10, h50, h100, n10, n50, n100};
    for (int i = 0; i < 6; ++i) {
        const float t = block_sum(vals[i], red);
        if (threadIdx.x == 0) metrics[i] += t;
    }
}

template <typename T, int CT>
size_t score_smem() { return (size_t)2 * 128 * (16 * CT + ElemTraits<T>::VEC) * sizeof(T); }

template <typename T, int CT>
int run_fwd(ScoreP p, hipStream_t st) {
    const size_t smem = score_smem<T, CT>();
    auto k = score_fwd_kernel<T, CT>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, dim3((p.R + 127) / 128, p.nchunk), dim3(NT), smem, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
template <typename T, int CT>
int run_bwd(ScoreP p, hipStream_t st) {
    const size_t tile = score_smem<T, CT>();
    const size_t red1 = (size_t)2 * 64 * 16 * CT * sizeof(float);
    const size_t smem1 = std::max(tile, red1);
    auto k1 = score_bwd_dy_kernel<T, CT>;
    hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem1);
    hipLaunchKernelGGL(k1, dim3((p.R + 127) / 128, p.nchunk), dim3(NT), smem1, st, p);
    EDGL_LAUNCH_CHECK();
    const long n = (long)p.R * p.C;
    hipLaunchKernelGGL((slab_reduce_kernel<T>), dim3((unsigned)std::min<long>((n + 255) / 256, 2048)), dim3(256), 0, st,
                       p.slabs, p.nchunk, n, reinterpret_cast<T*>(p.d_rows));
    EDGL_LAUNCH_CHECK();
    const size_t smem2 = std::max(tile + 128 * 3 * sizeof(float), red1 + 128 * sizeof(float));
    auto k2 = score_bwd_dw_kernel<T, CT>;
    hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
    hipLaunchKernelGGL(k2, dim3((p.i1 - p.i0 + 127) / 128), dim3(NT), smem2, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

template <typename T>
int dispatch_ct(ScoreP p, bool bwd, hipStream_t st) {
    const int maxc = sizeof(T) == 4 ? 128 : 256;
    EDGL_REQUIRE(p.C % 32 == 0 && p.C >= 32 && p.C <= maxc && (p.C & (p.C - 1)) == 0, EDGL_ERR_SHAPE,
                 "edgl_score: C=%d unsupported (power of two in [32, %d])", p.C, maxc);
    switch (p.C / 16) {
        case 2: return bwd ? run_bwd<T, 2>(p, st) : run_fwd<T, 2>(p, st);
        case 4: return bwd ? run_bwd<T, 4>(p, st) : run_fwd<T, 4>(p, st);
        case 8: return bwd ? run_bwd<T, 8>(p, st) : run_fwd<T, 8>(p, st);
        case 16:
            if constexpr (sizeof(T) == 2) return bwd ? run_bwd<T, 16>(p, st) : run_fwd<T, 16>(p, st);
    }
    edgl_set_error("edgl_score: C=%d unsupported", p.C);
    return EDGL_ERR_SHAPE;
}

int check_score(const void* rows, const void* table, const float* out_bias, int R, int C, int I, int i0, int i1,
                int dtype, const char* who) {
    EDGL_REQUIRE(rows && table && out_bias, EDGL_ERR_NULL, "%s: null pointer", who);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "%s: bad dtype %d", who, dtype);
    EDGL_REQUIRE(R > 0 && I > 1 && i0 >= 0 && i1 <= I && i0 < i1, EDGL_ERR_SHAPE, "%s: bad shape R=%d I=%d [%d,%d)", who,
                 R, I, i0, i1);
    EDGL_REQUIRE(((uintptr_t)rows & 15) == 0 && ((uintptr_t)table & 15) == 0, EDGL_ERR_SHAPE, "%s: operands must be 16-byte aligned", who);
    (void)C;
    return EDGL_OK;
}

}  // namespace

extern "C" int edgl_score_chunks(int n_items) { return (n_items + CHUNK - 1) / CHUNK; }

extern "C" int edgl_score_lse_fwd(const void* rows, const void* table, const float* out_bias, const int64_t* labels,
                                  int R, int C, int I, int i0, int i1, float* row_lse, float* label_logit,
                                  float* logits, float* workspace, int dtype, void* stream) {
    int rc = check_score(rows, table, out_bias, R, C, I, i0, i1, dtype, "edgl_score_lse_fwd");
    if (rc) return rc;
    EDGL_REQUIRE(row_lse && workspace && (!labels || label_logit), EDGL_ERR_NULL, "edgl_score_lse_fwd: null output");
    ScoreP p{};
    p.rows = rows; p.table = table; p.out_bias = out_bias; p.labels = labels; p.R = R; p.C = C; p.I = I; p.i0 = i0;
    p.i1 = i1; p.row_lse = row_lse; p.label_logit = label_logit; p.part = workspace; p.logits = logits;
    p.nchunk = edgl_score_chunks(i1 - i0);
    hipStream_t st = (hipStream_t)stream;
    rc = dtype == EDGL_F32 ? dispatch_ct<float>(p, false, st) : dispatch_ct<bf16>(p, false, st);
    if (rc) return rc;
    hipLaunchKernelGGL(lse_combine_kernel, dim3((R + 255) / 256), dim3(256), 0, st, workspace, R, p.nchunk, row_lse);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_ce_loss_fwd(const float* row_lse, const float* label_logit, const int64_t* labels, int R,
                                float* loss_out, float* coef, void* stream) {
    EDGL_REQUIRE(row_lse && label_logit && labels && loss_out && coef, EDGL_ERR_NULL, "edgl_ce_loss_fwd: null pointer");
    hipLaunchKernelGGL(ce_loss_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, row_lse, label_logit, labels, R, loss_out, coef);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" long edgl_score_bwd_workspace(int R, int C, int n_items) { return (long)edgl_score_chunks(n_items) * R * C; }

extern "C" int edgl_score_ce_bwd(const void* rows, const void* table, const float* out_bias, const int64_t* labels,
                                 const float* row_lse, const float* coef, const float* gscale, int R, int C, int I,
                                 int i0, int i1, void* d_rows, float* d_table, float* d_bias, float* workspace,
                                 int dtype, void* stream) {
    int rc = check_score(rows, table, out_bias, R, C, I, i0, i1, dtype, "edgl_score_ce_bwd");
    if (rc) return rc;
    EDGL_REQUIRE(labels && row_lse && coef && d_rows && d_table && d_bias && workspace, EDGL_ERR_NULL,
                 "edgl_score_ce_bwd: null pointer");
    ScoreP p{};
    p.rows = rows; p.table = table; p.out_bias = out_bias; p.labels = labels; p.R = R; p.C = C; p.I = I; p.i0 = i0;
    p.i1 = i1; p.row_lse = const_cast<float*>(row_lse); p.coef = coef; p.gscale = gscale; p.d_rows = d_rows;
    p.d_table = d_table; p.d_bias = d_bias; p.slabs = workspace; p.nchunk = edgl_score_chunks(i1 - i0);
    hipStream_t st = (hipStream_t)stream;
    return dtype == EDGL_F32 ? dispatch_ct<float>(p, true, st) : dispatch_ct<bf16>(p, true, st);
}

extern "C" int edgl_mask_topk(float* logits, int R, int n, int i0, const int64_t* seen, int T, int K, float* out_val,
                              int32_t* out_idx, void* stream) {
    EDGL_REQUIRE(logits && out_val && out_idx, EDGL_ERR_NULL, "edgl_mask_topk: null pointer");
    EDGL_REQUIRE(R > 0 && n > 0 && K > 0 && K <= 128, EDGL_ERR_SHAPE, "edgl_mask_topk: bad shape R=%d n=%d K=%d", R, n, K);
    hipLaunchKernelGGL(mask_topk_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, logits, R, n, i0, seen, T, K, out_val, out_idx);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_topk_merge(const float* cand_val, const int32_t* cand_idx, int S, int R, int K, float* out_val,
                               int32_t* out_idx, void* stream) {
    EDGL_REQUIRE(cand_val && cand_idx && out_val && out_idx, EDGL_ERR_NULL, "edgl_topk_merge: null pointer");
    EDGL_REQUIRE(S > 0 && R > 0 && K > 0 && S * K <= 1024, EDGL_ERR_SHAPE, "edgl_topk_merge: S*K=%d > 1024", S * K);
    hipLaunchKernelGGL(topk_merge_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, cand_val, cand_idx, S, R, K, out_val, out_idx);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_rank_metrics(const int32_t* topk_idx, int R, int K, const int64_t* label, float* metrics, void* stream) {
    EDGL_REQUIRE(topk_idx && label && metrics, EDGL_ERR_NULL, "edgl_rank_metrics: null pointer");
    hipLaunchKernelGGL(rank_metrics_kernel, dim3(1), dim3(512), 0, (hipStream_t)stream, topk_idx, R, K, label, metrics);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

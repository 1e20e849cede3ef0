// K6f: fused evaluation scoring — Sequential.eval's  logits = rows . table^T + bias  ->  seen mask  ->  top-K  (Base.py:150-181,
// EasyDGL.py:149-151) WITHOUT the [R, I] logits tile in HBM (the unfused path writes it — 41 MB at 512 x 20 001 — and reads it back).
//
// The logits are cheap (2.6 GFLOP at the headline shape: microseconds of the matrix pipe); what costs is holding them.  So they are
// computed TWICE and never stored:
//   pass 1  eval_gmax_kernel   maxima of the UNMASKED logits over G disjoint item groups per row (16 groups per item slice: the
//                              (wave, lane half, accumulator block) an element lands in) -> gmax [R, G]
//   thr     eval_thr_kernel    L[row] = the (K + T)-th largest group maximum.  At most T groups owe their maximum to a seen item (a row
//                              has T seen ids: Base.py:156-163), so at least K groups hold an UNSEEN item >= L: every one of the
//                              row's top K unseen items is >= L.  Also clears the row's candidate count.
//   pass 2  eval_emit_kernel   the logits again; an element >= L[row] that is not one of the row's seen ids (an LDS bitmap of the
//                              workgroup's 64 rows x its item slice, built under the first tile's loads) goes to an LDS list of the
//                              workgroup; behind the sweep ONE global atomic per row reserves the row's places in its candidate
//                              list [R, cap] and the entries are written — a few hundred per row
//   rank    eval_rank_kernel   one workgroup per row: the candidates above the K-th largest thread maximum are ranked by counting in
//                              (value desc, index asc).  Rows whose list overflowed (heavy ties, clustered maxima — never on real score
//                              distributions) are redone exactly in the same launch: the row's logits into a scratch row, seen
//                              mask, radix select (topk_select.h)
// Four launches; every launch boundary costs ~4.5 us on this device (the first build had six and was slower than the two kernels it
// replaced), so nothing small stands alone: the overflow list and the exact fallback live inside the ranking launch.
// Both sweeps: a workgroup = 64 rows x one item slice; item tiles of NZ rows stream through a double-buffered LDS image, the 64 query
// rows stay in registers as MFMA B fragments; D[item][query] = Z . X^T with v_mfma_f32_32x32x16_bf16 (items on the accumulator
// registers, the query on the lane: a lane's 16 values of a tile belong to ONE row, so group maxima and threshold tests need no
// cross-lane traffic).  bf16, C in {64, 128, 256}.
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "topk_select.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace evk {

constexpr int GPS = 16;                        // groups per (row, slice)
// query rows per workgroup: 64; 128 at C = 256, where the table no longer fits an L2 at the catalogues this width is used with and
// every query block reads every slice again (1 M items x 512 B: the sweeps ran on the L2 -> LDS traffic of 8 query blocks)
__host__ __device__ constexpr int qb_of(int C) { return C == 256 ? 128 : 64; }
constexpr int RSTR = 24;          // pass 2: a workgroup's LDS list of RUNS — 4 logits of consecutive items | hit mask + row | first item
constexpr int OVERFLOW = 0x40000000;           // added to a row's count when a list dropped entries

struct P {
    const bf16* rows; const bf16* table; const float* bias; const int64_t* seen;
    int T, R, I, i0, i1, K;
    int nslices, tps, stride;     // item slices, tiles per slice, pass 1: every stride-th tile
    int G;                        // groups per row = nslices * GPS
    float* gmax; float* thr; int32_t* count; float* cval; int32_t* cidx; int cap;
    float* scratch;               // [R, i1 - i0] f32: logits rows of the exact fallback (touched by overflowed rows only)
    float* out_val; int32_t* out_idx;
};

// NWS: waves across the items of a tile (the workgroup is QB / 32 x NWS waves: 32 query rows x NZ / NWS items each).  Four at C <= 128 —
// two waves per SIMD: the sweep's MFMA chains and the second sweep's hit bookkeeping are one dependent stream per wave, and a lone
// wave per SIMD has nothing to put into the other's shadow.
template <int C, int NZ, int NWS>
struct Geo {
    static constexpr int QB = qb_of(C);
    static constexpr int NTHR = 64 * (QB / 32) * NWS;
    static constexpr int RCAP = QB == 64 ? 1024 : 1536;       // runs a workgroup's list holds (more of them: every row goes the exact way)
    static constexpr int LDZ = C + 8;                          // bf16 elements per LDS row (16-byte shift per row: conflict-free 16-byte fragment reads)
    static constexpr int ZB = NZ * LDZ * 2;
    static constexpr int OFF_Z = 0, OFF_BIAS = 2 * ZB, OFF_LIST = OFF_BIAS + 2 * NZ * 4;
    static constexpr int LISTB = RCAP * RSTR, DUMB = NTHR * RSTR;   // the run list; one dump slot per thread (stores without a branch)
    static constexpr int OFF_DUM = OFF_LIST + LISTB;
    static constexpr int OFF_CNT = OFF_DUM + DUMB;             // list length, then per-row counts [QB] and global bases [QB]
    static constexpr int OFF_BMP = OFF_CNT + 16 + 2 * QB * 4;  // seen bitmap [QB][words]
    static constexpr int BYTES_GMAX = OFF_BIAS + 2 * NZ * 4;
    static constexpr int bytes_emit(int slice_items) { return OFF_BMP + QB * ((slice_items + 31) / 32) * 4; }
    static constexpr int PIECES = NZ * C / 8 / NTHR;           // 16-byte pieces of an item tile per thread
    static_assert(NZ * C / 8 % NTHR == 0 && NZ % (32 * NWS) == 0 && 8 % NWS == 0 && PIECES >= 1, "tile geometry");
};

// one item tile: global -> registers (rows clamped into the table; item 0 is the zero-padded row: coding.py:56-57)
template <int C, int NZ, int NWS>
__device__ __forceinline__ void tile_load(const P& p, int item0, uint4 (&r)[Geo<C, NZ, NWS>::PIECES], float& b) {
    constexpr int CP = C / 8, NTHR = Geo<C, NZ, NWS>::NTHR;
#pragma unroll
    for (int j = 0; j < Geo<C, NZ, NWS>::PIECES; ++j) {
        const int pc = threadIdx.x + NTHR * j, row = pc / CP, c8 = pc % CP;
        const int item = min(item0 + row, p.I - 1);
        r[j] = *reinterpret_cast<const uint4*>(p.table + (long)item * C + c8 * 8);
    }
    const int it = item0 + (int)threadIdx.x;   // (threads < NZ: the tile's bias row — EasyDGL.py:149-151, Base.py:110)
    b = (threadIdx.x < NZ && it < p.i1 && it > 0) ? p.bias[min(it, p.I - 1) - 1] : 0.f;
}
template <int C, int NZ, int NWS>
__device__ __forceinline__ void tile_store(const P& p, int item0, uint4 (&r)[Geo<C, NZ, NWS>::PIECES], float b, char* zbuf, float* bbuf) {
    constexpr int CP = C / 8, NTHR = Geo<C, NZ, NWS>::NTHR;
#pragma unroll
    for (int j = 0; j < Geo<C, NZ, NWS>::PIECES; ++j) {
        const int pc = threadIdx.x + NTHR * j, row = pc / CP, c8 = pc % CP;
        if (item0 + row == 0) r[j] = make_uint4(0u, 0u, 0u, 0u);
        *reinterpret_cast<uint4*>(zbuf + row * (Geo<C, NZ, NWS>::LDZ * 2) + c8 * 16) = r[j];
    }
    if (threadIdx.x < NZ) {
        const int it = item0 + (int)threadIdx.x;
        bbuf[threadIdx.x] = it >= p.i1 ? -INFINITY : (it == 0 ? -1000.0f : b);
    }
}

// EMIT = false: pass 1 (group maxima);  true: pass 2 (candidates)
template <int C, int NZ, int NWS, bool EMIT>
__global__ __launch_bounds__(64 * (qb_of(C) / 32) * NWS) void eval_sweep_kernel(P p) {
    using G_ = Geo<C, NZ, NWS>;
    constexpr int CK = C / 16, NTW = NZ / (32 * NWS), LDZB = G_::LDZ * 2, NTHR = G_::NTHR;
    constexpr int JG = 8 / NWS;       // group maxima per lane (GPS = 16 groups per slice = NWS waves x 2 lane halves x JG)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QB = G_::QB, RCAP = G_::RCAP;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = wave / NWS, sw = wave % NWS, l32 = lane & 31, hi = lane >> 5;
    // XCD-aware order (speed only; observed placement: workgroup id b runs on XCD b % 8): the query blocks of ONE item slice are
    // consecutive workgroups of ONE XCD, so that they run side by side and the slice's table rows come out of that XCD's L2 for all
    // but the first of them — in (query block, slice) grid order the 8 query blocks of a slice sat on 8 XCDs and every one of them
    // pulled the slice from HBM (1 M items: 8 x 512 MB per sweep)
    const int nqb = (p.R + QB - 1) / QB;
    const int xcd = (int)blockIdx.x & 7, jx = (int)blockIdx.x >> 3;
    const int slice = (jx / nqb) * 8 + xcd;
    if (slice >= p.nslices) return;
    const int q0 = (jx % nqb) * QB;
    int* lcount = reinterpret_cast<int*>(smem + G_::OFF_CNT);
    // ---- this wave's 32 query rows as B fragments, straight from global memory (once per workgroup: no LDS image) -------------------
    bf16x8 xf[CK];
    {
        const bf16* xrow = p.rows + (long)min(q0 + 32 * h + l32, p.R - 1) * C + hi * 8;
#pragma unroll
        for (int k = 0; k < CK; ++k) xf[k] = *reinterpret_cast<const bf16x8*>(xrow + k * 16);
    }
    int* rowcnt = lcount + 4;                     // [QB] survivors per row, then [QB] the rows' bases in their global lists
    uint32_t* bmp = reinterpret_cast<uint32_t*>(smem + G_::OFF_BMP);
    const int bw = (p.tps * NZ + 31) / 32;        // bitmap words per row
    if constexpr (EMIT) {
        if (tid == 0) *lcount = 0;
        for (int i = tid; i < 2 * QB; i += NTHR) rowcnt[i] = 0;
        for (int i = tid; i < QB * bw; i += NTHR) bmp[i] = 0u;
    }
    const int tile_lo = slice * p.tps, ntile_all = (p.i1 - p.i0 + NZ - 1) / NZ;
    const int tile_hi = min(ntile_all, tile_lo + p.tps);
    const int tstep = EMIT ? 1 : p.stride;
    // Two register staging sets: the loads of tile t + 2 leave while tile t is multiplied and tile t + 1 waits in the other set for
    // its LDS buffer — with one set a workgroup had ONE 32 KB tile in flight per 2 us memory round trip (the sweep ran at 9 % of
    // the matrix pipe on a 1 M-item table).  Out-of-range tiles are loaded clamped and never stored.
    uint4 stg[G_::PIECES], stg2[G_::PIECES];
    float stb, stb2;
    if (tile_lo < tile_hi) {
        tile_load<C, NZ, NWS>(p, p.i0 + tile_lo * NZ, stg, stb);
        tile_store<C, NZ, NWS>(p, p.i0 + tile_lo * NZ, stg, stb, smem + G_::OFF_Z, reinterpret_cast<float*>(smem + G_::OFF_BIAS));
        tile_load<C, NZ, NWS>(p, p.i0 + min(tile_lo + tstep, tile_hi - 1) * NZ, stg, stb);      // tile 1 (set A)
    }
    __syncthreads();
    if constexpr (EMIT) {
        // the rows' seen ids that fall into this slice as a bitmap (Base.py:156-163): all loads of a batch out before the first LDS
        // atomic (a load -> atomic loop is one memory round trip per id: 25 per thread)
        const int item_lo = p.i0 + tile_lo * NZ, item_hi = min(p.i1, p.i0 + tile_hi * NZ);
        // the 64 rows' ids are one contiguous run of int64 (16-byte aligned: 64 T ids per row block): pairs by 16-byte loads, up to
        // 16 in flight per thread — one memory round trip at T = 101, two at T = 201
        const int nid = min(QB, p.R - q0) * p.T, npair = nid / 2;
        const int64_t* sbase = p.seen + (long)q0 * p.T;
        auto mark = [&](int i, int64_t v) {
            if (v >= item_lo && v < item_hi) {
                const int o = (int)(v - item_lo);
                atomicOr(bmp + (i / p.T) * bw + (o >> 5), 1u << (o & 31));
            }
        };
        for (int i0_ = tid; i0_ < npair; i0_ += 16 * NTHR) {
            ulonglong2 v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) v[j] = *reinterpret_cast<const ulonglong2*>(sbase + 2 * min(i0_ + j * NTHR, npair - 1));
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(v[j].x), "+v"(v[j].y));
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int i = i0_ + j * NTHR;
                if (i < npair) { mark(2 * i, (int64_t)v[j].x); mark(2 * i + 1, (int64_t)v[j].y); }
            }
        }
        if ((nid & 1) && tid == 0) mark(nid - 1, sbase[nid - 1]);
        __syncthreads();
    }
    const int q = q0 + 32 * h + l32;             // this lane's query row
    float gm[JG];
#pragma unroll
    for (int j = 0; j < JG; ++j) gm[j] = -INFINITY;
    float thr = 0.f;
    if constexpr (EMIT) thr = p.thr[min(q, p.R - 1)];
    int buf = 0;
    // one trip = two tiles: tile t from LDS buffer `buf` with set A holding tile t + 1 and set B taking tile t + 2, then the same with
    // the sets swapped (static names: register arrays cannot be indexed by the trip's parity)
    auto one_tile = [&](int t, uint4 (&sa)[G_::PIECES], float& ba, uint4 (&sb)[G_::PIECES], float& bbn) {
        const int tn = t + tstep, tnn = t + 2 * tstep;
        if (tnn < tile_hi) tile_load<C, NZ, NWS>(p, p.i0 + tnn * NZ, sb, bbn);
        asm volatile("" ::: "memory");
        const char* zb = smem + G_::OFF_Z + buf * G_::ZB;
        const float* bb = reinterpret_cast<const float*>(smem + G_::OFF_BIAS) + buf * NZ;
#pragma unroll
        for (int u = 0; u < NTW; ++u) {
            const int m0 = (sw * NTW + u) * 32;               // the MFMA tile's first item row inside the LDS tile
            f32x16 acc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {                     // D[m][n] starts from the item's bias: m = 8 j + 4 hi + i
                const f32x4 b4 = *reinterpret_cast<const f32x4*>(bb + m0 + 8 * j + 4 * hi);
                acc[4 * j] = b4[0]; acc[4 * j + 1] = b4[1]; acc[4 * j + 2] = b4[2]; acc[4 * j + 3] = b4[3];
            }
#pragma unroll
            for (int k = 0; k < CK; ++k) {
                const bf16x8 zf = *reinterpret_cast<const bf16x8*>(zb + (m0 + l32) * LDZB + k * 32 + hi * 16);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(zf, xf[k], acc, 0, 0, 0);
            }
            if constexpr (!EMIT) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    gm[j * JG / 4] = fmaxf(fmaxf(gm[j * JG / 4], fmaxf(acc[4 * j], acc[4 * j + 1])), fmaxf(acc[4 * j + 2], acc[4 * j + 3]));
            } else {
                // (items past the range carry -inf and thr > -inf: never)
                // The lane's 16 logits are 4 runs of 4 consecutive items; a run with a hit goes to the workgroup's list as
                // (4 logits | hit mask + row | first item).  Straight-line code: every run is stored — to its place or to the thread's
                // dump slot —, so the bookkeeping of a tile can sit in the shadow of the next tile's MFMA chain and of the SIMD's
                // other wave.  The runs are taken apart — seen ids dropped — behind the sweep.
                const int item_base = p.i0 + t * NZ + m0 + 4 * hi;
                uint32_t hits = 0u;
#pragma unroll
                for (int r = 0; r < 16; ++r) hits |= (acc[r] >= thr ? 1u : 0u) << r;
                hits = q < p.R ? hits : 0u;
                // places from wave ballots: ONE LDS atomic per wave and tile (64 lanes adding to one LDS word are served one after
                // the other — ~8 cycles a lane: that, not the stores, was two thirds of this sweep)
                uint64_t bal[4];
                int tot = 0, pre[4];
                const uint64_t lt = (1ull << lane) - 1ull;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bal[j] = __ballot(((hits >> (4 * j)) & 0xfu) != 0u);
                    pre[j] = tot + __popcll(bal[j] & lt);
                    tot += __popcll(bal[j]);
                }
                int wbase = 0;
                if (lane == 0 && tot) wbase = atomicAdd(lcount, tot);
                wbase = __builtin_amdgcn_readfirstlane(wbase);
                char* dump = smem + G_::OFF_DUM + tid * RSTR;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int slot = wbase + pre[j];
                    const bool ok = ((bal[j] >> lane) & 1ull) && slot < RCAP;
                    char* e = ok ? smem + G_::OFF_LIST + slot * RSTR : dump;
                    *reinterpret_cast<float2*>(e) = make_float2(acc[4 * j], acc[4 * j + 1]);
                    *reinterpret_cast<float2*>(e + 8) = make_float2(acc[4 * j + 2], acc[4 * j + 3]);
                    *reinterpret_cast<int2*>(e + 16) = make_int2((int)(((hits >> (4 * j)) & 0xfu) | ((32 * h + l32) << 8)), item_base + 8 * j);
                }
            }
        }
        if (tn < tile_hi)
            tile_store<C, NZ, NWS>(p, p.i0 + tn * NZ, sa, ba, smem + G_::OFF_Z + (buf ^ 1) * G_::ZB,
                              reinterpret_cast<float*>(smem + G_::OFF_BIAS) + (buf ^ 1) * NZ);
        __syncthreads();
        buf ^= 1;
    };
    for (int t = tile_lo; t < tile_hi; t += 2 * tstep) {
        one_tile(t, stg, stb, stg2, stb2);
        if (t + tstep < tile_hi) one_tile(t + tstep, stg2, stb2, stg, stb);      // (block-uniform)
    }
    if constexpr (!EMIT) {
        if (q < p.R) {
            float* g = p.gmax + (long)q * p.G + slice * GPS + (sw * 2 + hi) * JG;
#pragma unroll
            for (int j = 0; j < JG; ++j) g[j] = gm[j];
        }
    } else {
        // ---- the runs' hits against the seen bitmap: count per row, ONE global atomic per row for the places, then they go out --------
        __syncthreads();
        const int nrun = *lcount;
        const char* ent = smem + G_::OFF_LIST;
        const int item_lo = p.i0 + tile_lo * NZ;
        auto for_hits = [&](auto&& f) {      // every unseen hit of every run: f(row, item, value)
            for (int i = tid; i < min(nrun, RCAP) * 4; i += NTHR) {
                const char* e = ent + (i >> 2) * RSTR;
                const int r = i & 3;
                const int2 meta = *reinterpret_cast<const int2*>(e + 16);
                if (!((meta.x >> r) & 1)) continue;
                const int row = (meta.x >> 8) & (QB - 1), item = meta.y + r, o = item - item_lo;
                if ((bmp[row * bw + (o >> 5)] >> (o & 31)) & 1u) continue;     // a seen id of this row (Base.py:156-163)
                f(row, item, reinterpret_cast<const float*>(e)[r]);
            }
        };
        if (nrun <= RCAP) for_hits([&](int row, int, float) { atomicAdd(rowcnt + row, 1); });
        __syncthreads();
        if (tid < QB && q0 + tid < p.R) {
            // (dropped runs: every row of the workgroup goes the exact way — a flag bit, not a sum: several workgroups may set it)
            if (nrun > RCAP) atomicOr(p.count + q0 + tid, OVERFLOW);
            else rowcnt[QB + tid] = rowcnt[tid] ? (atomicAdd(p.count + q0 + tid, rowcnt[tid]) & (OVERFLOW - 1)) : 0;
        }
        __syncthreads();
        if (nrun <= RCAP)
            for_hits([&](int row, int item, float v) {
                const int pos = atomicAdd(rowcnt + QB + row, 1);      // (any order inside the row's reserved run)
                if (pos < p.cap) {
                    const long o = (long)(q0 + row) * p.cap + pos;
                    p.cval[o] = v; p.cidx[o] = item;
                }
            });
    }
}

// L[row] = a lower bound of the rank-th largest of the row's G group maxima: its key with the low 16 bits cleared, found bit by
// bit on ballots (16 rounds instead of 32: any value at or below the exact one is a valid bound, and 2^-7 relative slack adds a
// dozen candidates).  Fewer than `rank` finite maxima: every finite element passes and the row goes the exact way.  Clears the row's
// candidate count.
__global__ __launch_bounds__(256) void eval_thr_kernel(const float* gmax, int R, int G, int rank, float* thr, int32_t* count) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= R) return;
    uint32_t key[16];
    float gv[16];
    const int ng = (G + 63) / 64;      // (wave-uniform)
#pragma unroll
    for (int j = 0; j < 16; ++j) gv[j] = gmax[(long)row * G + min(lane + 64 * j, G - 1)];   // unconditional: one round trip, not sixteen
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" : "+v"(gv[j]));
#pragma unroll
    for (int j = 0; j < 16; ++j) key[j] = lane + 64 * j < G ? float_key(gv[j]) : 0u;
    uint32_t L = 0u;
    for (int bit = 31; bit >= 16; --bit) {
        const uint32_t cand = L | (1u << bit);
        int c = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j)
            if (j < ng) c += __popcll(__ballot(key[j] >= cand));
        if (c >= rank) L = cand;
    }
    if (lane == 0) {
        thr[row] = L <= float_key(-INFINITY) ? -3.0e38f : key_float(L);      // (padded items carry -inf and never pass)
        count[row] = 0;
    }
}

// One workgroup per row.  n = the row's candidate count:
//   0 <= n <= cap   the K-th largest of the 256 thread maxima (a thread holds <= 4 candidates) bounds the K-th largest candidate from
//                   below; the candidates at or above it are compacted and each one's rank among them in (value desc, index asc) is
//                   its place in the output (as topk_merge_kernel in k_score.hip).  More than 512 of them (tie blocks): exact way.
//   otherwise       exact: the row's logits into its scratch row (VALU dot products: the operands are bf16, the sum f32), seen
//                   mask, radix select over the row.
template <int C>
__global__ __launch_bounds__(256) void eval_rank_kernel(P p) {
    __shared__ uint32_t tmx[256];
    __shared__ uint32_t Lsh;
    __shared__ __attribute__((aligned(16))) uint32_t ckey[520];
    __shared__ __attribute__((aligned(16))) int cix[520];
    __shared__ int ccount, wsum[4];
    __shared__ float xr[C];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6, K = p.K;
    const int n = p.count[row];
    float* ov = p.out_val + (long)row * K;
    int32_t* oi = p.out_idx + (long)row * K;
    bool exact = (n & OVERFLOW) != 0 || n < 0 || n > p.cap;
    if (!exact) {
        const int Keff = min(K, n);
        uint32_t key[4];
        int kid[4];
        float v4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {       // (cap <= 1024; slot clamped: all eight loads out together)
            const long o = (long)row * p.cap + min(tid + 256 * j, p.cap - 1);
            v4[j] = p.cval[o]; kid[j] = p.cidx[o];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) asm volatile("" : "+v"(v4[j]), "+v"(kid[j]));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = tid + 256 * j < n;
            key[j] = ok ? float_key(v4[j]) : 0u;
            kid[j] = ok ? kid[j] : 0x7fffffff;
        }
        tmx[tid] = max(max(key[0], key[1]), max(key[2], key[3]));
        if (tid == 0) ccount = 0;
        __syncthreads();
        if (w == 0) {
            const uint32_t m0 = tmx[lane], m1 = tmx[lane + 64], m2 = tmx[lane + 128], m3 = tmx[lane + 192];
            uint32_t Lw = 0u;
            for (int bit = 31; bit >= 12; --bit) {       // (a lower bound is enough: the low 12 bits stay 0)
                const uint32_t cand = Lw | (1u << bit);
                const int c = __popcll(__ballot(m0 >= cand)) + __popcll(__ballot(m1 >= cand)) + __popcll(__ballot(m2 >= cand)) +
                              __popcll(__ballot(m3 >= cand));
                if (c >= Keff) Lw = cand;
            }
            if (lane == 0) Lsh = Lw;
        }
        __syncthreads();
        const uint32_t L = max(Lsh, 1u);
        int c = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) c += __popcll(__ballot(key[j] >= L));
        if (lane == 0) wsum[w] = c;
        __syncthreads();
        const int Cn = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        if (Cn <= 512) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (key[j] >= L) {
                    const int pos = atomicAdd(&ccount, 1);
                    ckey[pos] = key[j]; cix[pos] = kid[j];
                }
            for (int i = tid; i < K; i += 256) { ov[i] = -INFINITY; oi[i] = -1; }
            __syncthreads();
            if (tid < 4) { ckey[Cn + tid] = 0u; cix[Cn + tid] = 0x7fffffff; }
            __syncthreads();
            for (int me = tid; me < Cn; me += 256) {
                const uint32_t a = ckey[me];
                const int ia = cix[me];
                int rank = 0;
#pragma unroll 2
                for (int qi = 0; qi < Cn; qi += 4) {
                    const uint4 b = *reinterpret_cast<const uint4*>(ckey + qi);
                    const int4 ib = *reinterpret_cast<const int4*>(cix + qi);
                    rank += ((b.x > a) || (b.x == a && ib.x < ia)) ? 1 : 0;      // (item ids are distinct: no third key)
                    rank += ((b.y > a) || (b.y == a && ib.y < ia)) ? 1 : 0;
                    rank += ((b.z > a) || (b.z == a && ib.z < ia)) ? 1 : 0;
                    rank += ((b.w > a) || (b.w == a && ib.w < ia)) ? 1 : 0;
                }
                if (rank < K) { ov[rank] = key_float(a); oi[rank] = ia; }
            }
            return;
        }
        exact = true;       // (wave-uniform: Cn comes from LDS)
    }
    // ---- exact ------------------------------------------------------------------------------------------------------------------
    __syncthreads();
    const int ni = p.i1 - p.i0;
    float* x = p.scratch + (long)row * ni;
    for (int i = tid; i < C; i += 256) xr[i] = to_f32(p.rows[(long)row * C + i]);
    __syncthreads();
    for (int i = tid; i < ni; i += 256) {
        const int item = p.i0 + i;
        float a = 0.f;
        if (item != 0) {
            const bf16* z = p.table + (long)item * C;
#pragma unroll 4
            for (int k = 0; k < C; k += 8) {
                const bf16x8 zv = *reinterpret_cast<const bf16x8*>(z + k);
#pragma unroll
                for (int e = 0; e < 8; ++e) a = fmaf(to_f32((bf16)zv[e]), xr[k + e], a);
            }
        }
        x[i] = item == 0 ? -1000.0f : a + p.bias[item - 1];
    }
    __syncthreads();
    for (int t = tid; t < p.T; t += 256) {
        const long id = p.seen[(long)row * p.T + t] - p.i0;
        if (id >= 0 && id < ni) x[id] = -INFINITY;
    }
    __syncthreads();
    radix_select_row(x, ni, p.i0, K, ov, oi);
}

constexpr int MAX_ITEMS = 262144;     // per call: 64 item slices of <= 4096 items (the seen bitmap of a workgroup: 32 KB of LDS)
struct Plan { int nz, nslices, tps, stride, G, cap; size_t off_thr, off_count, off_cval, off_cidx, off_scratch, bytes; };
inline bool make_plan(int R, int C, int n_items, int T, int K, Plan& pl) {
    if (!(C == 64 || C == 128 || C == 256) || K < 1 || K > 128 || R < 1 || T < 0 || n_items < 4096 || n_items > (1 << 30)) return false;
    pl.nz = C == 256 ? 64 : 128;
    const int ntile = (n_items + pl.nz - 1) / pl.nz;
    const int QB = qb_of(C);
    const int nrb = (R + QB - 1) / QB;
    // fill the chip AND keep the bound selective: three times as many groups as the rank of the bound among them
    int want = std::max(256 / std::max(nrb, 1), (3 * (K + T) + GPS - 1) / GPS);
    want = std::max(want, (n_items + (C == 256 ? 2047 : 4095)) / (C == 256 ? 2048 : 4096));   // a slice's seen bitmap
    want = std::min(std::min(want, 64), ntile);
    pl.tps = (ntile + want - 1) / want;
    pl.nslices = (ntile + pl.tps - 1) / pl.tps;
    pl.G = pl.nslices * GPS;
    // (the seen bitmap of a workgroup: 64 rows x its slice's items / 8 bytes of LDS beside the tiles and the entry list)
    if (pl.G < K + T + GPS || pl.G > 1024 || pl.tps * pl.nz > (C == 256 ? 2048 : 4096) + pl.nz) return false;
    // Expected candidates per row: a group maximum reaches the bound with probability P = rank / G, so an element does with
    // p = 1 - (1 - P)^(1 / group size) and a row lists n p of them (iid scores; real ones cluster less than the cap's margin).  Pass 1
    // over every second tile halves the groups (the bound stays valid: the sampled groups are disjoint item sets all the same) and
    // doubles the lists: taken for long ranges when the lists stay short.  Lists that would come near the cap of 1024: not taken.
    auto expect = [&](int stride) {
        const double P = (double)(K + T) / pl.G, gsz = std::max(1.0, (double)pl.tps * pl.nz / GPS / stride);
        return n_items * (1.0 - std::pow(1.0 - std::min(P, 0.999), 1.0 / gsz));
    };
    pl.stride = (n_items > 65536 && expect(2) < 480.0) ? 2 : 1;
    if (expect(pl.stride) > 640.0) return false;
    pl.cap = 1024;
    size_t o = 0;
    auto take = [&](size_t b) { const size_t at = o; o += (b + 255) & ~(size_t)255; return at; };
    take((size_t)R * pl.G * 4);
    pl.off_thr = take((size_t)R * 4); pl.off_count = take((size_t)R * 4);
    pl.off_cval = take((size_t)pl.cap * R * 4); pl.off_cidx = take((size_t)pl.cap * R * 4);
    pl.off_scratch = take((size_t)R * n_items * 4);      // (never touched unless a row overflows)
    pl.bytes = o;
    return true;
}

template <int C, int NZ, int NWS>
int launch(P p, const Plan& pl, hipStream_t st) {
    using G_ = Geo<C, NZ, NWS>;
    constexpr int NTHR = G_::NTHR;
    const dim3 grid(8 * ((p.R + G_::QB - 1) / G_::QB) * ((pl.nslices + 7) / 8));      // (query block, slice) from the id: see eval_sweep_kernel
    auto k1 = eval_sweep_kernel<C, NZ, NWS, false>;
    auto k2 = eval_sweep_kernel<C, NZ, NWS, true>;
    const int be = G_::bytes_emit(pl.tps * NZ);
    EDGL_REQUIRE(be <= 160 * 1024, EDGL_ERR_SHAPE, "edgl_score_topk_fused: %d B of LDS", be);
    hipFuncSetAttribute((const void*)k1, hipFuncAttributeMaxDynamicSharedMemorySize, G_::BYTES_GMAX);
    hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, be);
    hipLaunchKernelGGL(k1, grid, dim3(NTHR), G_::BYTES_GMAX, st, p);
    hipLaunchKernelGGL(eval_thr_kernel, dim3((p.R + 3) / 4), dim3(256), 0, st, p.gmax, p.R, p.G, p.K + p.T, p.thr, p.count);
    hipLaunchKernelGGL(k2, grid, dim3(NTHR), be, st, p);
    hipLaunchKernelGGL(eval_rank_kernel<C>, dim3(p.R), dim3(256), 0, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

}  // namespace evk

// 1 when edgl_score_topk_fused takes this shape (bf16; C in {64, 128, 256}; K <= 128; 4096 <= n_items <= 262144 per call; enough
// item slices for K + T group maxima), else 0: the caller runs edgl_score_lse_fwd
// (logits tile) + edgl_mask_topk.  There is no silent fallback inside the library.
extern "C" int edgl_score_topk_fused_supported(int R, int C, int n_items, int T, int K, int dtype) {
    evk::Plan pl;
    return dtype == EDGL_BF16 && n_items <= evk::MAX_ITEMS && evk::make_plan(R, C, n_items, T, K, pl) ? 1 : 0;
}
extern "C" long edgl_score_topk_fused_workspace(int R, int C, int n_items, int T, int K) {
    evk::Plan pl;
    if (!evk::make_plan(R, C, n_items, T, K, pl)) return -1;
    return (long)pl.bytes;
}
extern "C" int edgl_score_topk_fused(const void* rows, const void* table, const float* out_bias, const int64_t* seen, int T, int R, int C,
                                     int I, int i0, int i1, int K, float* out_val, int32_t* out_idx, void* workspace, int dtype,
                                     void* stream) {
    EDGL_REQUIRE(rows && table && out_bias && out_val && out_idx && workspace, EDGL_ERR_NULL, "edgl_score_topk_fused: null pointer");
    EDGL_REQUIRE(seen || T == 0, EDGL_ERR_NULL, "edgl_score_topk_fused: T > 0 without seen ids");
    EDGL_REQUIRE(dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_score_topk_fused: bf16 only (dtype %d)", dtype);
    EDGL_REQUIRE(I > 1 && i0 >= 0 && i1 <= I && i0 < i1 && i0 % 8 == 0, EDGL_ERR_SHAPE, "edgl_score_topk_fused: bad item range [%d, %d) of %d", i0, i1, I);
    evk::Plan pl;
    EDGL_REQUIRE(i1 - i0 <= evk::MAX_ITEMS && evk::make_plan(R, C, i1 - i0, T, K, pl), EDGL_ERR_SHAPE,
                 "edgl_score_topk_fused: shape not taken (R=%d C=%d items=%d T=%d K=%d; see edgl_score_topk_fused_supported)", R, C, i1 - i0, T, K);
    char* ws = (char*)workspace;
    evk::P p{};
    p.rows = (const bf16*)rows; p.table = (const bf16*)table; p.bias = out_bias; p.seen = seen; p.T = T; p.R = R; p.I = I; p.i0 = i0; p.i1 = i1;
    p.K = K; p.nslices = pl.nslices; p.tps = pl.tps; p.stride = pl.stride; p.G = pl.G;
    p.gmax = reinterpret_cast<float*>(ws); p.thr = reinterpret_cast<float*>(ws + pl.off_thr);
    p.count = reinterpret_cast<int32_t*>(ws + pl.off_count); p.cval = reinterpret_cast<float*>(ws + pl.off_cval);
    p.cidx = reinterpret_cast<int32_t*>(ws + pl.off_cidx); p.cap = pl.cap;
    p.scratch = reinterpret_cast<float*>(ws + pl.off_scratch); p.out_val = out_val; p.out_idx = out_idx;
    hipStream_t st = (hipStream_t)stream;
    if (C == 64) return evk::launch<64, 128, 4>(p, pl, st);
    if (C == 128) return evk::launch<128, 128, 4>(p, pl, st);
    return evk::launch<256, 64, 2>(p, pl, st);
}

// Fused per-sample block tail (EasyDGL.py:110-139): the reference's LayerNorm normalises jointly over (T, C) PER SAMPLE
// (Base.py:12-67), so one workgroup that owns a sample can run
//     ao = att.Wo + bo ; a1 = LN1(drop(ao) + x_in) ; f = gelu(a1.Wi + bi) ; o = f.Wout + bout ; y = LN2(drop(o) + a1)
//     [head]  so = gelu(y.Wt + bt) ; rows = LN3(so)[masked positions]
// with the [T, C] activations held in LDS between the steps (T = 101, C = 128: 28 KB per tensor) and the weights streamed
// from L2 as MFMA operands: no HBM round trip between the four dense layers and three LayerNorms — each intermediate is
// written once (the backward needs it) and never read back by this kernel.  The unfused path is seven launches.
//   MFMA orientation: D[n][row] = sum_k W^T[n][k] X[row][k]; a wave owns one 16-wide tile of output channels for ALL rows
//   (A operand = 16 rows of the packed W^T, read from global/L2; B operand = activation rows from LDS), so a lane ends up
//   with 4 consecutive channels of one row: bias, dropout, residual, GELU and the LayerNorm moments all run on registers;
//   results go to the next LDS buffer and leave for HBM as whole 256-byte rows.
// Arithmetic is that of the unfused kernels step for step (GEMM outputs rounded to the activation dtype before the
// LayerNorm reads them, two-pass moments, the same dropout element indices), so both paths produce the same tensors.
// bf16, C in {64, 128}, T <= 112.
#include "edgl_common.h"

#ifdef EDGL_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[16];   // see edgl_common.h (PH_MARK); read back with edgl_debug_phase_cycles_tail
#endif
// -DEDGL_PHASE_TIMING -DEDGL_PHASE_BWD: the slots count the phases of the backward kernel instead of the forward's
#if defined(EDGL_PHASE_TIMING) && defined(EDGL_PHASE_BWD)
#define PHB_DECL PH_DECL
#define PHB_MARK(i) PH_MARK(i)
#define PHB_FLUSH() PH_FLUSH(0)
#else
#define PHB_DECL
#define PHB_MARK(i)
#define PHB_FLUSH()
#endif

namespace {

constexpr int MAXRT = 7;   // 16-row tiles per sample (T <= 112)
// GELU runs as its own pass over an f32 LDS image (gelu_pass), eight elements per thread in a loop that is not unrolled,
// with the one-rcp-one-exp erf of edgl_common.h, and the same pass writes gelu'(pre) — the only thing the backward needs the
// pre-activations for — in place of the pre-activations themselves, so the backward multiplies by a loaded tensor.  Inside
// the epilogues (on the accumulator registers) libm's erff cost ~400 issue cycles per element — 28 % of the forward, 40 %
// of the backward — and the inlined fast form spilled there: the scheduler interleaves all 28 chains of a phase (~80 more
// registers), and with a scheduling fence per row tile the z / acc state still went to scratch (forward 84 -> 88 us,
// backward 120 -> 138 us).  Computing gelu' in the backward's input copy (fast form, 32 per thread and image) still cost
// 7 k of the 60 k cycles of a block, three to five times per step.

struct TailP {
    const bf16* att; const bf16* xin; int ld_x;
    const bf16 *WoT, *WiT, *WoutT, *WtT;            // packed [N][K] images (k contiguous)
    const float *bo, *bi, *bout, *bt, *g1, *b1, *g2, *b2, *g3, *b3;
    int B, T, C;
    float rate; const uint64_t* rng; uint32_t sid1, sid2;
    const int64_t* mpos; int M; int head; const int32_t* hmap;   // hmap: optional destination row of head row b*M + j (< 0: dropped)
    bf16 *ao, *a1, *pre_f, *f, *o, *y, *pre_t, *so, *hrows;   // pre_f / pre_t receive gelu'(pre-activation)
    float *st1, *st2, *st3;
    int dhp, dht;   // channel-padded model (head dim dht stored as dhp, the padded channels all zero): the LayerNorms' moments are
                    // those of the real channels (edgl_tail_fwd_ct); 0, 0: none
    int stagger;    // t2 kernels: the second half of the grid starts this many s_sleep(127) later (the two workgroups of a CU out of phase)
};

template <int CT>
struct TailGeom {
    static constexpr int C = 16 * CT, NW = CT, NTHR = 64 * CT, LD = C + 8, CV = C / 8;
    static constexpr int LDF = C + 4;                                  // f32 staging image [112][LDF] over buffers A + S
    static constexpr size_t BUF = (size_t)MAXRT * 16 * LD * sizeof(bf16);
    static_assert((size_t)MAXRT * 16 * LDF * sizeof(float) <= 2 * BUF, "f32 staging image exceeds two buffers");
    static constexpr size_t SMEM = 4 * BUF + 64 * sizeof(float);
    static constexpr size_t SMEM_BWD = SMEM + (size_t)(MAXRT * 16 + 512) * sizeof(int);   // + rowmap [112] + nextj, positions [<= 256 each]
    // forward: + the parameter vectors bo | bout | g1 | b1 | g2 | b2 | g3 | b3 | bt | bi (2C)   (see "waits" in tail_fwd_kernel)
    static constexpr int PAR_BO = 0, PAR_BOUT = 1, PAR_G1 = 2, PAR_B1 = 3, PAR_G2 = 4, PAR_B2 = 5, PAR_G3 = 6, PAR_B3 = 7, PAR_BT = 8, PAR_BI = 9;
    static constexpr size_t SMEM_FWD = SMEM + (size_t)11 * C * sizeof(float);
};

// rows [0, T) of a [T, C] global tensor (row stride ld) -> LDS image [112][LD]; rows >= T are zero.  load() issues every
// 16-byte vector of the image, store() writes them to the LDS: all loads of a phase — of BOTH its images — go out before the
// first store (a load / store loop pays one HBM round trip per 8 KB of a 512-thread workgroup, two copies one after the
// other pay two).  Straight-line code: rows are clamped, and the threads that have no vector of their own in the last round
// repeat one of the same round (same data to the same address) — a guard there puts the load and its wait inside a branch.
template <int CT>
struct RowImage {
    using G = TailGeom<CT>;
    static constexpr int NV = MAXRT * 16 * G::CV, NI = (NV + G::NTHR - 1) / G::NTHR;
    static_assert(NI * G::NTHR - NV < G::NTHR && NV >= G::NTHR, "last round: at most one repeat per thread");
    uint4 d[NI];
    static __device__ __forceinline__ int vec(int i) {
        const int v = threadIdx.x + i * G::NTHR;
        return v < NV ? v : v - (NI * G::NTHR - NV);
    }
    __device__ __forceinline__ void load(const bf16* src, long ld, int T) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int v = vec(i), row = v / G::CV, cv = v % G::CV;
            d[i] = *reinterpret_cast<const uint4*>(src + (long)min(row, T - 1) * ld + cv * 8);
        }
    }
    __device__ __forceinline__ void store(bf16* dst, int T) const {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int v = vec(i), row = v / G::CV, cv = v % G::CV;
            *reinterpret_cast<uint4*>(dst + row * G::LD + cv * 8) = row < T ? d[i] : make_uint4(0, 0, 0, 0);
        }
    }
};
template <int CT>
__device__ __forceinline__ void copy_in(bf16* dst, const bf16* src, long ld, int T) {
    RowImage<CT> a;
    a.load(src, ld, T);
    a.store(dst, T);
}
template <int CT>
__device__ __forceinline__ void copy_in2(bf16* dst0, const bf16* src0, long ld0, bf16* dst1, const bf16* src1, long ld1, int T) {
    RowImage<CT> a, b;
    a.load(src0, ld0, T);
    b.load(src1, ld1, T);
    a.store(dst0, T);
    b.store(dst1, T);
}
template <int CT>
__device__ __forceinline__ void copy_out(bf16* dst, long ld, const bf16* src, int T) {
    using G = TailGeom<CT>;
    for (int v = threadIdx.x; v < T * G::CV; v += G::NTHR) {
        const int row = v / G::CV, cv = v % G::CV;
        *reinterpret_cast<uint4*>(dst + (long)row * ld + cv * 8) = *reinterpret_cast<const uint4*>(src + row * G::LD + cv * 8);
    }
}

// f32 image stg [112][LDF] of dense outputs (bias not yet added; `bias` = the C biases of these columns) -> with pre = . + bias:
// dact = gelu'(pre) (bf16, global) and act = gelu(pre) (bf16, global + LDS image act_s).  A thread keeps its 8 columns
// over all its rows.
template <int CT>
__device__ __forceinline__ void gelu_pass(const float* stg, const float* bias, bf16* dact_g, bf16* act_g, long ldg, bf16* act_s, int T) {
    using G = TailGeom<CT>;
    static_assert(G::NTHR % G::CV == 0, "a thread's column group must not change from row to row");
    const int cv = threadIdx.x % G::CV;
    const float4 b0 = *reinterpret_cast<const float4*>(bias + cv * 8), b1 = *reinterpret_cast<const float4*>(bias + cv * 8 + 4);   // (LDS)
#pragma unroll 1
    for (int row = threadIdx.x / G::CV; row < MAXRT * 16; row += G::NTHR / G::CV) {
        const float4 x0 = *reinterpret_cast<const float4*>(stg + row * G::LDF + cv * 8);
        const float4 x1 = *reinterpret_cast<const float4*>(stg + row * G::LDF + cv * 8 + 4);
        const float x[8] = {x0.x + b0.x, x0.y + b0.y, x0.z + b0.z, x0.w + b0.w, x1.x + b1.x, x1.y + b1.y, x1.z + b1.z, x1.w + b1.w};
        Vec16<bf16> dact, act;
#pragma unroll
        for (int j = 0; j < 8; ++j) {    // gelu_t / dgelu_t<bf16> of edgl_common.h with the erf shared
            float e;
            const float cdf = 0.5f * (1.0f + erf_as(x[j] * 0.70710678118654752440f, e));
            act.v[j] = from_f32<bf16>(x[j] * cdf);
            dact.v[j] = from_f32<bf16>(cdf + x[j] * (0.39894228040143267794f * e));
        }
        st16<bf16>(act_s + row * G::LD + cv * 8, act);
        if (row < T) {
            st16<bf16>(dact_g + (long)row * ldg + cv * 8, dact);
            st16<bf16>(act_g + (long)row * ldg + cv * 8, act);
        }
    }
}

// acc[rt] += W^T[n-tile rows][k0 .. k0 + 32*NKB) . X[rows of tile rt][same k]   (WT row stride ldw, X image stride LD)
// The weight fragments of a wave's 16 output channels (global, L2-resident) are fetched ahead of the product that uses
// them — one product early, under the epilogue / LayerNorm / copy-out of the previous step — so that no product of the
// chain opens with an L2 round trip (8 waves per CU: nothing else would hide it).
template <int NKB>
struct WFrags { Vec16<bf16> w[NKB]; };
template <int NKB>
__device__ __forceinline__ WFrags<NKB> load_wfrags(const bf16* WTrows, int ldw, int lane) {
    const int l15 = lane & 15, kg = (lane >> 4) * 8;
    WFrags<NKB> f;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) f.w[kb] = ld16<bf16>(WTrows + (long)l15 * ldw + kb * 32 + kg);
    return f;
}
template <int CT, int NKB>
__device__ __forceinline__ void tile_gemm(const WFrags<NKB>& wf, const bf16* Xs, int nrt, int lane, f32x4 (&acc)[MAXRT]) {
    using G = TailGeom<CT>;
    const int l15 = lane & 15, kg = (lane >> 4) * 8;
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) {
        if (rt < nrt) {
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb)
                acc[rt] = mma_kblock(wf.w[kb], ld16<bf16>(Xs + (rt * 16 + l15) * G::LD + kb * 32 + kg), acc[rt]);
        }
        // at most two row tiles' operand reads in flight: unbounded, the scheduler hoists all 7 x NKB LDS reads (112+
        // registers) above the first MFMA
        // (four or all seven tiles per batch: +-0, measured)
        if (rt & 1) __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void st_bf4(bf16* dst, const float (&v)[4]) {
    const Frag4<bf16> f = frag_from_acc<bf16>(f32x4{v[0], v[1], v[2], v[3]});
    *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(&f);
}
__device__ __forceinline__ void ld_bf4(const bf16* src, float (&v)[4]) {
    const Frag4<bf16> f = frag_ld<bf16>(src);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = to_f32(f.v[r]);
}
__device__ __forceinline__ float rbf(float x) { return to_f32(from_f32<bf16>(x)); }   // round through the activation dtype

// Workgroup sum through LDS with LDS-scoped barriers only: __syncthreads() also drains vmcnt, i.e. it would wait for the
// copy_out stores still in flight (a full HBM write latency at every step of the chain).
__device__ __forceinline__ float block_sum_lds(float v, float* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    lds_barrier();
    if (lane == 0) red[w] = v;
    lds_barrier();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}

// joint (T, C) moments of the values z[rt][r] held by the workgroup (rows >= T excluded): two passes, as tf.nn.moments
// Channel-padded models (dhp > 0: head dim dht stored as dhp; edgl_add_layernorm_*_ct in k_layernorm.hip): the moments are those of
// the REAL channels — divisor T * C_true, padded entries (exact zeros) left out of the centred second moment.
struct ChanPad {
    float rm[4];      // 1 for a real channel of this lane's four, 0 for a padded one
    float ctrue;      // real channels of the model width
};
template <int CT>
__device__ __forceinline__ ChanPad chan_pad(int nl, int dhp, int dht) {
    ChanPad cp;
#pragma unroll
    for (int r = 0; r < 4; ++r) cp.rm[r] = (dhp <= 0 || ((nl + r) & (dhp - 1)) < dht) ? 1.f : 0.f;      // (dhp is a power of two)
    cp.ctrue = dhp > 0 ? (float)((16 * CT) / dhp * dht) : (float)(16 * CT);
    return cp;
}
template <int CT>
__device__ __forceinline__ void joint_moments(const float (&z)[MAXRT][4], int nrt, int T, int lane, float* red, float& mean, float& rstd,
                                              const ChanPad& cp) {
    const int l15 = lane & 15;
    const float n = (float)T * cp.ctrue;
    float a = 0.f;
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt)
        if (rt < nrt && rt * 16 + l15 < T) a += (z[rt][0] * cp.rm[0] + z[rt][1] * cp.rm[1]) + (z[rt][2] * cp.rm[2] + z[rt][3] * cp.rm[3]);
    mean = block_sum_lds(a, red) / n;
    a = 0.f;
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt)
        if (rt < nrt && rt * 16 + l15 < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float d = (z[rt][r] - mean) * cp.rm[r]; a += d * d; }
        }
    rstd = rsqrtf(block_sum_lds(a, red) / n + 1e-12f);
}

// Four LDS images per workgroup: A (att -> y), B (x_in -> a1 -> LN3 rows), C (f halves, so), S (staging of the tensors
// that only pass through: ao, o); A + S together hold the f32 pre-activations of a GELU pass.  Everything leaves for HBM as whole 256-byte rows (copy_out); storing
// the pass-through tensors straight from the accumulator registers (8 bytes per lane) measured slower (88 vs 83 us).
// NRT: the row-tile count as a template constant (2, 4 or 7 tiles: T <= 32, 64, 112; tiles past the sequence are padding like
// the rows past it) — every `rt < nrt` guard folds away and the accumulator tiles stay independent tuples.  With a run-time
// count (NRT = 0, not instantiated) the compiler keeps whole 28-register copies of the tile arrays alive across the guards:
// 47 / 129 spilled registers.
template <int CT, int NRT>
__global__ __launch_bounds__(64 * CT) void tail_fwd_kernel(TailP p) {
    using G = TailGeom<CT>;
    constexpr int C = G::C, LD = G::LD, NKB = C / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* bufA = reinterpret_cast<bf16*>(smem);
    bf16* bufS = reinterpret_cast<bf16*>(smem + G::BUF);
    bf16* bufB = reinterpret_cast<bf16*>(smem + 2 * G::BUF);
    bf16* bufC = reinterpret_cast<bf16*>(smem + 3 * G::BUF);
    float* stg = reinterpret_cast<float*>(smem);             // f32 image over A + S while neither holds a tensor
    float* red = reinterpret_cast<float*>(smem + 4 * G::BUF);
    const int b = blockIdx.x, T = p.T, nrt = NRT ? NRT : (T + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int n0 = wave * 16, nl = n0 + g4;                 // this lane's 4 output channels: nl .. nl + 3
    const long row0 = (long)b * T;
    const DropKey dk1 = make_dropkey(p.rng, p.sid1, p.rate), dk2 = make_dropkey(p.rng, p.sid2, p.rate);
    const ChanPad cpad = chan_pad<CT>(nl, p.dhp, p.dht);

    PH_DECL
    // ---- waits ------------------------------------------------------------------------------------------------------------
    // Loads and stores share ONE in-order counter (vmcnt) on this part: waiting for a load also waits for every store issued
    // before it, and behind a store loop of run-time length the compiler can only wait for "everything".  As first written —
    // parameter vectors loaded where they are used, weight fragments one product ahead — eight points of a sample waited for
    // the store batch issued just before them (copy_out / the GELU pass: 28-56 KB per workgroup, all 256 workgroups at once).
    // Now nothing is loaded in front of its use: every parameter vector is requested before the first store of the kernel,
    // and a product's weight fragments at the START of the segment (the stretch between two store batches) BEFORE the one
    // that uses them, with their wait pinned (touch_regs) to that segment's end — a whole segment after the last stores.
    // (The parameter vectors wait in LDS, not in registers: 56 of them per lane spilled the 7-tile instance.)
    // Measured: 68.5 -> 67 us.  The launch is not bound by its memory side at all — without the GELU passes' global stores
    // (49 % of the bytes written) it takes 61.5 us — but by the serial chain of a workgroup's phases: three GELU passes
    // (15 VALU + 2 transcendental instructions per element, ~35 % of a sample), six LDS-fed products (~27 %), the joint
    // moments and copies.
    float* par = red + 64;     // parameter vectors [11][C] (G::PAR_*)
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x;
        const float* bt_or = p.head ? p.bt : p.bo;   // (head parameters: valid pointers to load from when there is no head)
        const float *g3_or = p.head ? p.g3 : p.g1, *b3_or = p.head ? p.b3 : p.b1;
        const float x0 = p.bo[c], x1 = p.bout[c], x2 = p.g1[c], x3 = p.b1[c], x4 = p.g2[c], x5 = p.b2[c], x6 = g3_or[c], x7 = b3_or[c],
                    x8 = bt_or[c], x9 = p.bi[c], x10 = p.bi[C + c];
        par[G::PAR_BO * C + c] = x0; par[G::PAR_BOUT * C + c] = x1; par[G::PAR_G1 * C + c] = x2; par[G::PAR_B1 * C + c] = x3;
        par[G::PAR_G2 * C + c] = x4; par[G::PAR_B2 * C + c] = x5; par[G::PAR_G3 * C + c] = x6; par[G::PAR_B3 * C + c] = x7;
        par[G::PAR_BT * C + c] = x8; par[G::PAR_BI * C + c] = x9; par[(G::PAR_BI + 1) * C + c] = x10;
    }
    // first round of the head's row gather (M * C / 8 vectors over the workgroup; later rounds load theirs in the loop)
    int gat_t = 0; long gat_dst = -1;
    if (p.head) {
        const int v = min((int)threadIdx.x, p.M * G::CV - 1), j = v / G::CV;
        const long src = (long)b * p.M + j;
        gat_t = (int)p.mpos[src];
        gat_dst = p.hmap ? (long)p.hmap[src] : src;
    }
    WFrags<NKB> wA = load_wfrags<NKB>(p.WoT + (long)n0 * C, C, lane);      // att_out dense
    WFrags<NKB> wB = load_wfrags<NKB>(p.WiT + (long)n0 * C, C, lane);      // inner dense, first half
    WFrags<NKB> wC;
    copy_in2<CT>(bufA, p.att + row0 * C, C, bufB, p.xin + row0 * p.ld_x, p.ld_x, T);
    lds_barrier();
    PH_MARK(0);   // inputs in LDS
    float z[MAXRT][4];
    // ---- segment A: ao = att.Wo + bo ; z1 = drop(ao) + x_in (EasyDGL.py:113-115) ------------------------------------------
    {
        f32x4 acc[MAXRT];
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        tile_gemm<CT, NKB>(wA, bufA, nrt, lane, acc);
        const float4 v_bo = *reinterpret_cast<const float4*>(par + G::PAR_BO * C + nl);
        const float bv[4] = {v_bo.x, v_bo.y, v_bo.z, v_bo.w};
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt)
            if (rt < nrt) {
                const int row = rt * 16 + l15;
                float v[4], xr[4];
                ld_bf4(bufB + row * LD + nl, xr);
                float dv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = rbf(acc[rt][r] + bv[r]); dv[r] = v[r]; }
                drop_apply4(dk1, (uint64_t)((row0 + min(row, T - 1)) * C + nl), dv);   // one hash per four channels
#pragma unroll
                for (int r = 0; r < 4; ++r) z[rt][r] = dv[r] + xr[r];
                st_bf4(bufS + row * LD + nl, v);
            }
    }
    PH_MARK(1);   // G1 + epilogue
    lds_barrier();
    touch_regs(wB);     // every load so far has arrived: stores may start
    copy_out<CT>(p.ao + row0 * C, C, bufS, T);
    PH_MARK(2);   // barrier + copy_out(ao)
    // ---- segment B: a1 = LN1(z1) (EasyDGL.py:116) -> B, in place of the residual it consumed -------------------------------
    wA = load_wfrags<NKB>(p.WoutT + (long)n0 * 2 * C, 2 * C, lane);        // out dense, first half of its inputs (segment D)
    {
        float mean, rstd;
        joint_moments<CT>(z, nrt, T, lane, red, mean, rstd, cpad);
        if (threadIdx.x == 0) { p.st1[2 * b] = mean; p.st1[2 * b + 1] = rstd; }
        const float4 v_g = *reinterpret_cast<const float4*>(par + G::PAR_G1 * C + nl), v_b = *reinterpret_cast<const float4*>(par + G::PAR_B1 * C + nl);
        const float gv[4] = {v_g.x, v_g.y, v_g.z, v_g.w}, ev[4] = {v_b.x, v_b.y, v_b.z, v_b.w};
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt)
            if (rt < nrt) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (z[rt][r] - mean) * rstd * gv[r] + ev[r];
                st_bf4(bufB + (rt * 16 + l15) * LD + nl, v);
            }
    }
    PH_MARK(3);   // LN1
    lds_barrier();
    touch_regs(wA);
    copy_out<CT>(p.a1 + row0 * C, C, bufB, T);
    PH_MARK(4);   // barrier + copy_out(a1)
    // ---- f = gelu(a1.Wi + bi) in two halves of C columns; o accumulates f.Wout half by half (EasyDGL.py:120-125) ----------
    f32x4 acc3[MAXRT];
    // segment C: first half of the inner dense
    wC = load_wfrags<NKB>(p.WiT + (long)(C + n0) * C, C, lane);            // inner dense, second half (segment D)
    {
        f32x4 acc[MAXRT];
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        tile_gemm<CT, NKB>(wB, bufB, nrt, lane, acc);
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt)   // all 7 tiles, unconditionally (tiles >= nrt hold zeros)
            *reinterpret_cast<float4*>(stg + (rt * 16 + l15) * G::LDF + nl) = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
    }
    lds_barrier();
    touch_regs(wC);
    gelu_pass<CT>(stg, par + G::PAR_BI * C, p.pre_f + row0 * 2 * C, p.f + row0 * 2 * C, 2 * C, bufC, T);
    PH_MARK(5);   // G2 half + GELU
    lds_barrier();
    // segment D: first half of the out dense, second half of the inner dense
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) acc3[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    tile_gemm<CT, NKB>(wA, bufC, nrt, lane, acc3);
    wA = load_wfrags<NKB>(p.WoutT + (long)n0 * 2 * C + C, 2 * C, lane);    // out dense, second half of its inputs (segment E; a third live set spills)
    EDGL_PIN();
    PH_MARK(6);   // G3 half
    {
        f32x4 acc[MAXRT];
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        tile_gemm<CT, NKB>(wC, bufB, nrt, lane, acc);
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt)
            *reinterpret_cast<float4*>(stg + (rt * 16 + l15) * G::LDF + nl) = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
    }
    lds_barrier();   // (every wave is also past its reads of the first half's f image)
    touch_regs(wA);
    gelu_pass<CT>(stg, par + (G::PAR_BI + 1) * C, p.pre_f + row0 * 2 * C + C, p.f + row0 * 2 * C + C, 2 * C, bufC, T);
    PH_MARK(5);
    lds_barrier();
    // segment E: second half of the out dense
    wB = load_wfrags<NKB>(p.WtT + (long)n0 * C, C, lane);                  // head transform (fetched even without a head: cheap)
    tile_gemm<CT, NKB>(wA, bufC, nrt, lane, acc3);
    PH_MARK(6);
    // ---- o = . + bout ; z2 = drop(o) + a1 ; y = LN2(z2) (EasyDGL.py:126-128) -> A ---------------------------------------------
    {
        const float4 v_bout = *reinterpret_cast<const float4*>(par + G::PAR_BOUT * C + nl);
        const float bv[4] = {v_bout.x, v_bout.y, v_bout.z, v_bout.w};
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt)
            if (rt < nrt) {
                const int row = rt * 16 + l15;
                float v[4], xr[4];
                ld_bf4(bufB + row * LD + nl, xr);
                float dv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) { v[r] = rbf(acc3[rt][r] + bv[r]); dv[r] = v[r]; }
                drop_apply4(dk2, (uint64_t)((row0 + min(row, T - 1)) * C + nl), dv);   // one hash per four channels
#pragma unroll
                for (int r = 0; r < 4; ++r) z[rt][r] = dv[r] + xr[r];
                st_bf4(bufS + row * LD + nl, v);
            }
    }
    lds_barrier();
    touch_regs(wB);
    copy_out<CT>(p.o + row0 * C, C, bufS, T);
    {
        float mean, rstd;
        joint_moments<CT>(z, nrt, T, lane, red, mean, rstd, cpad);
        if (threadIdx.x == 0) { p.st2[2 * b] = mean; p.st2[2 * b + 1] = rstd; }
        const float4 v_g = *reinterpret_cast<const float4*>(par + G::PAR_G2 * C + nl), v_b = *reinterpret_cast<const float4*>(par + G::PAR_B2 * C + nl);
        const float gv[4] = {v_g.x, v_g.y, v_g.z, v_g.w}, ev[4] = {v_b.x, v_b.y, v_b.z, v_b.w};
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt)
            if (rt < nrt) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (z[rt][r] - mean) * rstd * gv[r] + ev[r];
                st_bf4(bufA + (rt * 16 + l15) * LD + nl, v);
            }
    }
    lds_barrier();
    copy_out<CT>(p.y + row0 * C, C, bufA, T);
    PH_MARK(7);   // o, LN2, y
#if defined(EDGL_PHASE_TIMING) && !defined(EDGL_PHASE_BWD)
    PH_FLUSH(0); ph_acc[0] = 0; ph_t0 = __builtin_readcyclecounter();
#endif
    if (!p.head) return;
    // ---- head: so = gelu(y.Wt + bt) ; rows = LN3(so)[masked positions] (EasyDGL.py:136-146) --------------------------------
    {
        f32x4 acc[MAXRT];
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        tile_gemm<CT, NKB>(wB, bufA, nrt, lane, acc);
        lds_barrier();   // y (buffer A) is overwritten by the f32 image
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt)
            *reinterpret_cast<float4*>(stg + (rt * 16 + l15) * G::LDF + nl) = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
    }
    lds_barrier();
    gelu_pass<CT>(stg, par + G::PAR_BT * C, p.pre_t + row0 * C, p.so + row0 * C, C, bufC, T);
    lds_barrier();
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) ld_bf4(bufC + (rt * 16 + l15) * LD + nl, z[rt]);   // so, rounded through the activation dtype
    {
        float mean, rstd;
        joint_moments<CT>(z, nrt, T, lane, red, mean, rstd, cpad);
        if (threadIdx.x == 0) { p.st3[2 * b] = mean; p.st3[2 * b + 1] = rstd; }
        const float4 v_g = *reinterpret_cast<const float4*>(par + G::PAR_G3 * C + nl), v_b = *reinterpret_cast<const float4*>(par + G::PAR_B3 * C + nl);
        const float gv[4] = {v_g.x, v_g.y, v_g.z, v_g.w}, ev[4] = {v_b.x, v_b.y, v_b.z, v_b.w};
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt)
            if (rt < nrt) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (z[rt][r] - mean) * rstd * gv[r] + ev[r];
                st_bf4(bufB + (rt * 16 + l15) * LD + nl, v);
            }
    }
    lds_barrier();
    // batch_gather of the masked positions (EasyDGL.py:142-143); dst: row compaction of the scoring (edgl_compact_scan's `inv`)
    if ((int)threadIdx.x < p.M * G::CV && gat_dst >= 0)
        *reinterpret_cast<uint4*>(p.hrows + gat_dst * C + (threadIdx.x % G::CV) * 8) =
            *reinterpret_cast<const uint4*>(bufB + gat_t * LD + (threadIdx.x % G::CV) * 8);
    for (int v = threadIdx.x + G::NTHR; v < p.M * G::CV; v += G::NTHR) {
        const int j = v / G::CV, cv = v % G::CV;
        const long src = (long)b * p.M + j;
        const int t = (int)p.mpos[src];
        const long dst = p.hmap ? (long)p.hmap[src] : src;
        if (dst >= 0) *reinterpret_cast<uint4*>(p.hrows + dst * C + cv * 8) = *reinterpret_cast<const uint4*>(bufB + t * LD + cv * 8);
    }
#if defined(EDGL_PHASE_TIMING) && !defined(EDGL_PHASE_BWD)
    PH_MARK(0);   // head (slot 8)
    if ((threadIdx.x & 63) == 0) atomicAdd(&g_phase_cycles[8], ph_acc[0]);
#endif
}

// =========================================================================================================================
// Backward of the same chain in one launch per block: LN3' -> GELU' -> dX(Wt) -> LN2' -> dX(Wout) -> GELU' -> dX(Wi) -> LN1' ->
// dX(Wo).  The dX products contract over the layer's OUTPUT index, so their weight operand is the [in, out] kernel itself
// (a lane = one input channel k, 8 consecutive n) — no packed image.  What the weight-gradient GEMMs need (the gradients
// w.r.t. the four dense outputs) is written once; the LayerNorm parameter gradients leave as per-sample partials.
// Rounding points as in the unfused kernels (every tensor those write in the activation dtype is rounded here too).
// =========================================================================================================================
struct TailBwdP {
    // saved by the forward
    const bf16 *xin; int ld_x;
    const bf16 *ao, *a1, *pre_f, *o, *pre_t, *so;   // pre_f / pre_t: gelu'(pre-activation), as edgl_tail_fwd wrote them
    const float *st1, *st2, *st3;
    const bf16 *Wo, *Wi, *Wout, *Wt;          // [in, out] compute copies
    const float *g1, *g2, *g3;
    int B, T, C;
    float rate; const uint64_t* rng; uint32_t sid1, sid2;
    int head;
    const bf16* d_rows; const int64_t* mpos; int M; const int32_t* rowmap;   // head: compact row gradients, positions, inv map
    const bf16* d_y_in;                                                       // head == 0: gradient w.r.t. y [B,T,C]
    // outputs
    bf16 *d_pre_t, *d_o, *d_pre_f, *d_ao, *d_res1, *d_att;
    float *part1, *part2, *part3;             // [B][2C] (dbeta | dgamma) of LN1 / LN2 / LN3
    int dhp, dht;                             // channel-padded model: see TailP
};

// dX tile: acc[rt] += W[k-tile rows][n0 .. n0 + 32*NKB) . G[rows of tile rt][same n]   (W row stride ldw; G image stride LD)
// — the same MFMA pattern as tile_gemm with the [in, out] kernel as the A operand
template <int CT, int NKB>
__device__ __forceinline__ void tile_dx(const WFrags<NKB>& wf, const bf16* Gs, int nrt, int lane, f32x4 (&acc)[MAXRT]) {
    tile_gemm<CT, NKB>(wf, Gs, nrt, lane, acc);
}

// LayerNorm backward on registers: z = LN input sum, dy = upstream gradient; returns d(sum) in dz and writes the
// per-sample (dbeta | dgamma) partials of this wave's channels
template <int CT>
__device__ __forceinline__ void ln_bwd_regs(const float (&z)[MAXRT][4], const float (&dy)[MAXRT][4], float mean, float rstd,
                                            const float (&gv)[4], int nrt, int T, int lane, int nl, float* red, float* part_b,
                                            float (&dz)[MAXRT][4], const ChanPad& cp) {
    const int l15 = lane & 15;
    const float n = (float)T * cp.ctrue;      // (gamma is 0 on padded channels: they add nothing to s1 / s2)
    float s1 = 0.f, s2 = 0.f, dga[4] = {0.f, 0.f, 0.f, 0.f}, dbe[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt)
        if (rt < nrt && rt * 16 + l15 < T) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float xh = (z[rt][r] - mean) * rstd, gg = dy[rt][r] * gv[r];
                s1 += gg; s2 += gg * xh; dga[r] += dy[rt][r] * xh; dbe[r] += dy[rt][r];
            }
        }
    const float m1 = block_sum_lds(s1, red) / n;
    const float m2 = block_sum_lds(s2, red) / n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {   // sum over the 16 rows held by the lanes of this lane group
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) { dga[r] += __shfl_xor(dga[r], o, 64); dbe[r] += __shfl_xor(dbe[r], o, 64); }
    }
    if (l15 == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) { part_b[nl + r] = dbe[r]; part_b[16 * CT + nl + r] = dga[r]; }
    }
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float xh = (z[rt][r] - mean) * rstd;
            dz[rt][r] = cp.rm[r] * (rstd * (dy[rt][r] * gv[r] - m1 - xh * m2));      // nothing flows into a padded channel
        }
}

template <int CT, int NRT>
__global__ __launch_bounds__(64 * CT) void tail_bwd_kernel(TailBwdP p) {
    using G = TailGeom<CT>;
    constexpr int C = G::C, LD = G::LD, NKB = C / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    bf16* bufA = reinterpret_cast<bf16*>(smem);
    bf16* bufS = reinterpret_cast<bf16*>(smem + G::BUF);
    bf16* bufB = reinterpret_cast<bf16*>(smem + 2 * G::BUF);
    bf16* bufC = reinterpret_cast<bf16*>(smem + 3 * G::BUF);
    float* red = reinterpret_cast<float*>(smem + 4 * G::BUF);
    int* rowmap = reinterpret_cast<int*>(red + 64);          // [T] head of the chain of gathered rows naming position t
    int* nextj = rowmap + MAXRT * 16;                        // [M]
    int* mpos_s = nextj + 256;                               // [M] masked positions of this sample
    const int b = blockIdx.x, T = p.T, nrt = NRT ? NRT : (T + 15) / 16;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const int n0 = wave * 16, nl = n0 + g4;
    const long row0 = (long)b * T;
    const ChanPad cpad = chan_pad<CT>(nl, p.dhp, p.dht);
    float z[MAXRT][4], dy[MAXRT][4], dz[MAXRT][4];
    // weight fragments of the next product, fetched one product ahead (see load_wfrags)
    WFrags<NKB> wf = p.head ? load_wfrags<NKB>(p.Wt + (long)n0 * C, C, lane) : load_wfrags<NKB>(p.Wout + (long)n0 * C, C, lane);
    PHB_DECL
    RowImage<CT> im_o, im_a;      // o, a1: loaded one phase ahead

    if (p.head) {
        // ---- LN3' on the gathered rows, GELU' -> d_pre_t (EasyDGL.py:136-146 backward) ------------------------------------
        // the M gathered row gradients of this sample -> buffer C rows [0, M) (compacted-away rows as zeros): walking the
        // chains below against global memory cost one round trip per link and row tile (50 of the head block's 117 k cycles).
        // Their row indices are fetched first, so that the second hop travels with the two images.
        const bool staged = p.M <= MAXRT * 16;
        constexpr int NG = RowImage<CT>::NI;
        int gsrc[NG];
        if (staged && p.rowmap) {      // (uniform branch; straight-line loads inside)
#pragma unroll
            for (int i = 0; i < NG; ++i) gsrc[i] = p.rowmap[(long)b * p.M + min(RowImage<CT>::vec(i) / G::CV, p.M - 1)];
        } else {
#pragma unroll
            for (int i = 0; i < NG; ++i) gsrc[i] = b * p.M + min(RowImage<CT>::vec(i) / G::CV, p.M - 1);
        }
        const int my_pos = (int)p.mpos[(long)b * p.M + min((int)threadIdx.x, p.M - 1)];      // M <= 256 <= threads
        RowImage<CT> im_so, im_dg;
        im_so.load(p.so + row0 * C, C, T);
        im_dg.load(p.pre_t + row0 * C, C, T);
        if (staged) {
            uint4 g[NG];
#pragma unroll
            for (int i = 0; i < NG; ++i)
                g[i] = *reinterpret_cast<const uint4*>(p.d_rows + (long)max(gsrc[i], 0) * C + (RowImage<CT>::vec(i) % G::CV) * 8);
            im_so.store(bufB, T);
            im_dg.store(bufA, T);
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const int v = RowImage<CT>::vec(i), j = v / G::CV, cv = v % G::CV;
                if (j < p.M) *reinterpret_cast<uint4*>(bufC + j * LD + cv * 8) = gsrc[i] >= 0 ? g[i] : make_uint4(0, 0, 0, 0);
            }
        } else {
            im_so.store(bufB, T);
            im_dg.store(bufA, T);
        }
        if ((int)threadIdx.x < p.M) mpos_s[threadIdx.x] = my_pos;
        for (int t = threadIdx.x; t < MAXRT * 16; t += G::NTHR) rowmap[t] = -1;
        lds_barrier();
        // chains of the gathered rows naming one position, in ascending j (the order the unfused kernel sums in): row j links to
        // the next larger j' with the same position; the smallest j of a position is its head
        for (int j = threadIdx.x; j < p.M; j += G::NTHR) {
            const int t = mpos_s[j];
            int nxt = -1;
            bool first = true;
#pragma unroll 8
            for (int k = 0; k < p.M; ++k) {          // uniform trip count: the LDS reads of an unrolled group travel together
                const bool same = mpos_s[k] == t;
                nxt = (same && k > j && nxt < 0) ? k : nxt;
                first = first && !(same && k < j);
            }
            nextj[j] = nxt;
            if (first) rowmap[t] = j;
        }
        lds_barrier();
        // first link of every row tile's chain as straight-line code (heads, rows, next links: three rounds of LDS reads for
        // all seven tiles instead of three dependent reads per tile); longer chains — repeated positions — in the loop below
        int jn[MAXRT];
        {
            int jh[MAXRT];
#pragma unroll
            for (int rt = 0; rt < MAXRT; ++rt) {
                const int row = rt * 16 + l15;
                ld_bf4(bufB + row * LD + nl, z[rt]);
                jh[rt] = (rt < nrt && row < T) ? rowmap[row] : -1;
            }
#pragma unroll
            for (int rt = 0; rt < MAXRT; ++rt) {
                jn[rt] = nextj[max(jh[rt], 0)];
                if (staged) {
                    ld_bf4(bufC + max(jh[rt], 0) * LD + nl, dy[rt]);
#pragma unroll
                    for (int r = 0; r < 4; ++r) dy[rt][r] = jh[rt] >= 0 ? dy[rt][r] : 0.f;
                    jn[rt] = jh[rt] >= 0 ? jn[rt] : -1;
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) dy[rt][r] = 0.f;
                    jn[rt] = jh[rt];
                }
            }
        }
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt)
            for (int j = jn[rt]; j >= 0; j = nextj[j]) {
                float v[4];
                if (staged) {
                    ld_bf4(bufC + j * LD + nl, v);
                } else {
                    long src = (long)b * p.M + j;
                    if (p.rowmap) src = p.rowmap[src];
                    if (src < 0) continue;
                    ld_bf4(p.d_rows + src * C + nl, v);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) dy[rt][r] += v[r];
            }
        PHB_MARK(0);   // head inputs, gathered row gradients
        {
            const float4 gg = *reinterpret_cast<const float4*>(p.g3 + nl);
            const float gv[4] = {gg.x, gg.y, gg.z, gg.w};
            ln_bwd_regs<CT>(z, dy, p.st3[2 * b], p.st3[2 * b + 1], gv, nrt, T, lane, nl, red, p.part3 + (long)b * 2 * C, dz, cpad);
        }
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt)
            if (rt < nrt) {
                const int row = rt * 16 + l15;
                float dg[4];
                ld_bf4(bufA + row * LD + nl, dg);
                const float v[4] = {dz[rt][0] * dg[0], dz[rt][1] * dg[1], dz[rt][2] * dg[2], dz[rt][3] * dg[3]};
                st_bf4(bufC + row * LD + nl, v);
            }
        lds_barrier();
        copy_out<CT>(p.d_pre_t + row0 * C, C, bufC, T);
        // ---- d_y = d_pre_t . Wt^T ----------------------------------------------------------------------------------------------
        // (the next phase's two images are requested here: their round trip runs under this product and its barrier)
        im_o.load(p.o + row0 * C, C, T);
        im_a.load(p.a1 + row0 * C, C, T);
        EDGL_PIN();
        f32x4 acc[MAXRT];
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        tile_dx<CT, NKB>(wf, bufC, nrt, lane, acc);
        wf = load_wfrags<NKB>(p.Wout + (long)n0 * C, C, lane);
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt)
#pragma unroll
            for (int r = 0; r < 4; ++r) dy[rt][r] = rbf(acc[rt][r]);
        lds_barrier();     // bufA / bufB / bufC are rewritten below
        PHB_MARK(1);   // LN3', gelu', dX(Wt)
    } else {
        im_o.load(p.o + row0 * C, C, T);
        im_a.load(p.a1 + row0 * C, C, T);
        EDGL_PIN();
        copy_in<CT>(bufA, p.d_y_in + row0 * C, C, T);
        lds_barrier();
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt) ld_bf4(bufA + (rt * 16 + l15) * LD + nl, dy[rt]);
        lds_barrier();
    }
    // ---- LN2': z2 = drop(o) + a1 ; d_o = drop(d_z2) ; d_a1 (residual part) = d_z2 (EasyDGL.py:126-128 backward) -----------------
    // (the dropout keys hang on a load of the generator state: created here, not ahead of the head's loads)
    const DropKey dk1 = make_dropkey(p.rng, p.sid1, p.rate), dk2 = make_dropkey(p.rng, p.sid2, p.rate);
    im_o.store(bufA, T);
    im_a.store(bufB, T);
    lds_barrier();
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) {
        const int row = rt * 16 + l15;
        float ov[4], av[4];
        ld_bf4(bufA + row * LD + nl, ov);
        ld_bf4(bufB + row * LD + nl, av);
        drop_apply4(dk2, (uint64_t)((row0 + min(row, T - 1)) * C + nl), ov);
#pragma unroll
        for (int r = 0; r < 4; ++r) z[rt][r] = ov[r] + av[r];
    }
    RowImage<CT> im_p;            // gelu'(pre_f) half: requested one phase ahead as well
    im_p.load(p.pre_f + row0 * 2 * C, 2 * C, T);
    EDGL_PIN();
    PHB_MARK(2);   // o, a1 -> z2
    {
        const float4 gg = *reinterpret_cast<const float4*>(p.g2 + nl);
        const float gv[4] = {gg.x, gg.y, gg.z, gg.w};
        ln_bwd_regs<CT>(z, dy, p.st2[2 * b], p.st2[2 * b + 1], gv, nrt, T, lane, nl, red, p.part2 + (long)b * 2 * C, dz, cpad);
    }
    float da1[MAXRT][4];   // gradient w.r.t. a1: the residual branch now, + d_pre_f . Wi^T below
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt)
        if (rt < nrt) {
            const int row = rt * 16 + l15;
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                da1[rt][r] = rbf(dz[rt][r]);
                v[r] = dz[rt][r];
            }
            drop_apply4(dk2, (uint64_t)((row0 + min(row, T - 1)) * C + nl), v);
            st_bf4(bufC + row * LD + nl, v);
        }
    lds_barrier();
    copy_out<CT>(p.d_o + row0 * C, C, bufC, T);
    PHB_MARK(3);   // LN2', d_o
    // ---- d_pre_f = (d_o . Wout^T) * gelu'(pre_f), two halves of the 2C hidden channels; d_a1 += d_pre_f . Wi^T -------------------
    f32x4 acc6[MAXRT];
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) acc6[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int h = 0; h < 2; ++h) {
        im_p.store(bufA, T);
        {
            f32x4 acc[MAXRT];
#pragma unroll
            for (int rt = 0; rt < MAXRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
            tile_dx<CT, NKB>(wf, bufC, nrt, lane, acc);
            wf = load_wfrags<NKB>(p.Wi + (long)n0 * 2 * C + h * C, 2 * C, lane);
            lds_barrier();   // gelu'(pre_f half) in place (and, for h = 1, every wave past its reads of the previous d_pre_f half)
#pragma unroll
            for (int rt = 0; rt < MAXRT; ++rt)
                if (rt < nrt) {
                    const int row = rt * 16 + l15;
                    float dg[4];
                    ld_bf4(bufA + row * LD + nl, dg);
                    const float v[4] = {acc[rt][0] * dg[0], acc[rt][1] * dg[1], acc[rt][2] * dg[2], acc[rt][3] * dg[3]};
                    st_bf4(bufB + row * LD + nl, v);
                }
        }
        lds_barrier();
        if (h == 0) {   // the other half's image / the two images of the LN1' phase travel under the rest of this half
            im_p.load(p.pre_f + row0 * 2 * C + C, 2 * C, T);
        } else {
            im_o.load(p.ao + row0 * C, C, T);
            im_a.load(p.xin + row0 * p.ld_x, p.ld_x, T);
        }
        EDGL_PIN();
        copy_out<CT>(p.d_pre_f + row0 * 2 * C + h * C, 2 * C, bufB, T);
        tile_dx<CT, NKB>(wf, bufB, nrt, lane, acc6);
        wf = h == 0 ? load_wfrags<NKB>(p.Wout + (long)(C + n0) * C, C, lane) : load_wfrags<NKB>(p.Wo + (long)n0 * C, C, lane);
        lds_barrier();
    }
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) dy[rt][r] = rbf(acc6[rt][r] + da1[rt][r]);
    PHB_MARK(4);   // the two halves of the hidden layer
    // ---- LN1': z1 = drop(ao) + x_in ; d_ao = drop(d_z1) ; d_res1 = d_z1 (EasyDGL.py:113-116 backward) ------------------------------
    im_o.store(bufA, T);
    im_a.store(bufB, T);
    lds_barrier();
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) {
        const int row = rt * 16 + l15;
        float ov[4], xv[4];
        ld_bf4(bufA + row * LD + nl, ov);
        ld_bf4(bufB + row * LD + nl, xv);
        drop_apply4(dk1, (uint64_t)((row0 + min(row, T - 1)) * C + nl), ov);
#pragma unroll
        for (int r = 0; r < 4; ++r) z[rt][r] = ov[r] + xv[r];
    }
    PHB_MARK(5);   // ao, x_in -> z1
    {
        const float4 gg = *reinterpret_cast<const float4*>(p.g1 + nl);
        const float gv[4] = {gg.x, gg.y, gg.z, gg.w};
        ln_bwd_regs<CT>(z, dy, p.st1[2 * b], p.st1[2 * b + 1], gv, nrt, T, lane, nl, red, p.part1 + (long)b * 2 * C, dz, cpad);
    }
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt)
        if (rt < nrt) {
            const int row = rt * 16 + l15;
            float v[4], w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                w[r] = dz[rt][r];
                v[r] = dz[rt][r];
            }
            drop_apply4(dk1, (uint64_t)((row0 + min(row, T - 1)) * C + nl), v);
            st_bf4(bufC + row * LD + nl, v);
            st_bf4(bufS + row * LD + nl, w);
        }
    lds_barrier();
    copy_out<CT>(p.d_ao + row0 * C, C, bufC, T);
    copy_out<CT>(p.d_res1 + row0 * C, C, bufS, T);
    PHB_MARK(6);   // LN1', d_ao, d_res1
    // ---- d_att = d_ao . Wo^T ---------------------------------------------------------------------------------------------------------
    {
        f32x4 acc[MAXRT];
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
        tile_dx<CT, NKB>(wf, bufC, nrt, lane, acc);
#pragma unroll
        for (int rt = 0; rt < MAXRT; ++rt)
            if (rt < nrt) {
                const float v[4] = {acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]};
                st_bf4(bufA + (rt * 16 + l15) * LD + nl, v);
            }
    }
    lds_barrier();
    copy_out<CT>(p.d_att + row0 * C, C, bufA, T);
    PHB_MARK(7);   // dX(Wo), d_att
    PHB_FLUSH();
}

// =========================================================================================================================
// Two workgroups per CU ("t2", C = 128, T <= 101): the same chain on HALF the CU.
// The kernels above own a CU — 122 KB of LDS, 256 registers per wave — so the 512 samples of the benchmark run as two lock-stepped
// rounds of 256 workgroups, every phase of a sample a dependent round trip (L2 / HBM / LDS hand-over / barrier) with nothing on
// the CU to run in the gaps: 50-53 % of the wave cycles parked, the matrix pipe 6 % busy (DESIGN.md rule 53).  At C = 64 two of the
// 4-wave workgroups already fit a CU, and there the second one costs 32 % more time, not 100 % (tools/tail_probe.py: 22.9 us for
// 256 samples, 30.2 us for 512).  This form gives the C = 128 kernel the same second, independent chain:
//   * three LDS images [101][128] bf16, UNPADDED (256-byte rows) with an XOR swizzle — 16-byte chunk c of row r at c ^ (r & 15) — in
//     place of four padded ones + an f32 image: 3 x 25 856 + 256 + 4 096 = 81 920 bytes = half the CU's LDS.  The swizzle is
//     conflict-free for all three access shapes: MFMA operand reads (16 lanes = 16 rows of one chunk column), the 8-byte quads of
//     the accumulator layout (32 lanes = 16 rows x 2 halves of one chunk), whole rows (any permutation);
//   * the f32 pre-activations of a GELU pass go through TWO of the images ([101][128] f32, same swizzle on its 16-byte chunks);
//     the pass loads its values, and — where the f image lands in the staging area itself — waits for everybody's loads before
//     storing (one more barrier per half of the hidden layer);
//   * <= 128 registers (4 waves per SIMD): no operand is prefetched a whole segment ahead — the other workgroup of the CU is
//     what runs while a load is in flight —, the LayerNorm inputs of the backward come straight from global memory in the
//     accumulator layout (8 bytes per lane and row tile) instead of through LDS images held in registers one phase ahead.
// Same arithmetic, in the same order, as the kernels above: the outputs are bit-identical (tests/test_gpu_tail2.py).
// =========================================================================================================================
namespace t2 {
constexpr int C = 128, NTHR = 512, NKB = 4, TMAX = 101;
constexpr int IMG = TMAX * 256;                                      // one bf16 image
constexpr int OFF_RED = 3 * IMG, OFF_PAR = OFF_RED + 256;
constexpr int PAR_BOUT = 0, PAR_G2 = 1, PAR_B2 = 2, PAR_G3 = 3, PAR_B3 = 4, PAR_BI = 5, PAR_BT = 7, NPAR = 8;      // (bi: two vectors)
constexpr int SMEM_FWD = OFF_PAR + NPAR * C * 4;
static_assert(SMEM_FWD == 81920, "two workgroups per CU");
constexpr int OFF_ROWMAP = OFF_RED + 256, OFF_NEXTJ = OFF_ROWMAP + MAXRT * 16 * 4, OFF_MPOS = OFF_NEXTJ + 256 * 4;
constexpr int SMEM_BWD = OFF_MPOS + 256 * 4;
static_assert(SMEM_BWD <= 81920, "two workgroups per CU");

struct Lane {
    int lane, wave, q, l15, nl;
    int frag[NKB];      // byte offset of this lane's MFMA operand chunk of k-block kb in row l15 of an image (+ rt * 4096)
    int quad;           // byte offset of this lane's 4 channels in row l15 of an image (+ rt * 4096)
    int squad;          // the same in the f32 staging image (+ rt * 8192)
};
__device__ __forceinline__ Lane make_lane() {
    Lane L;
    L.lane = threadIdx.x & 63; L.wave = threadIdx.x >> 6; L.q = L.lane >> 4; L.l15 = L.lane & 15; L.nl = L.wave * 16 + L.q * 4;
#pragma unroll
    for (int kb = 0; kb < NKB; ++kb) L.frag[kb] = L.l15 * 256 + (((kb * 4 + L.q) ^ L.l15) << 4);
    L.quad = L.l15 * 256 + (((L.wave * 2 + (L.q >> 1)) ^ L.l15) << 4) + ((L.q & 1) << 3);
    L.squad = L.l15 * 512 + (((L.wave * 4 + L.q) ^ L.l15) << 4);
    return L;
}
__device__ __forceinline__ int vec_off(int row, int cv) { return row * 256 + ((cv ^ (row & 15)) << 4); }

// acc[rt] += W^T[16 output channels][128 k] . X[rows of tile rt][k] over a swizzled image.  Rows >= T of the last tile(s) read
// whatever lies there (inside the workgroup's LDS): an MFMA output row depends on its own input row only, and every consumer
// of rows >= T is guarded.
template <int NRT>
__device__ __forceinline__ void gemm(const WFrags<NKB>& wf, const char* img, const Lane& L, f32x4 (&acc)[MAXRT]) {
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) {
#pragma unroll
        for (int kb = 0; kb < NKB; ++kb)
            acc[rt] = mma_kblock(wf.w[kb], *reinterpret_cast<const Vec16<bf16>*>(img + L.frag[kb] + rt * 4096), acc[rt]);
        __builtin_amdgcn_sched_barrier(0);      // one row tile's operand reads in flight (16 registers)
    }
}
template <int NRT>
__device__ __forceinline__ void zero_acc(f32x4 (&acc)[MAXRT]) {
#pragma unroll
    for (int rt = 0; rt < MAXRT; ++rt) acc[rt] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// global [T, 128] (row stride ld) -> image; every load of BOTH images in flight before the first store (clamped tail vectors
// repeat the last one: same data to the same address)
__device__ __forceinline__ void copy_in2(char* d0, const bf16* s0, long ld0, char* d1, const bf16* s1, long ld1, int T) {
    uint4 a[4], b[4];
    const int nv = T * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = min((int)threadIdx.x + i * NTHR, nv - 1), row = v >> 4, cv = v & 15;
        a[i] = *reinterpret_cast<const uint4*>(s0 + (long)row * ld0 + cv * 8);
        b[i] = *reinterpret_cast<const uint4*>(s1 + (long)row * ld1 + cv * 8);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = min((int)threadIdx.x + i * NTHR, nv - 1), row = v >> 4, cv = v & 15;
        *reinterpret_cast<uint4*>(d0 + vec_off(row, cv)) = a[i];
        *reinterpret_cast<uint4*>(d1 + vec_off(row, cv)) = b[i];
    }
}
__device__ __forceinline__ void copy_out(bf16* dst, long ld, const char* img, int T) {
    const int nv = T * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = threadIdx.x + i * NTHR, row = v >> 4, cv = v & 15;
        if (v < nv) *reinterpret_cast<uint4*>(dst + (long)row * ld + cv * 8) = *reinterpret_cast<const uint4*>(img + vec_off(row, cv));
    }
}
// accumulators (bias not yet added) -> f32 staging image, rows < T only (the image is exactly two bf16 images long)
template <int NRT>
__device__ __forceinline__ void stage_acc(char* stg, const f32x4 (&acc)[MAXRT], const Lane& L, int T) {
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt)
        if (rt * 16 + L.l15 < T)
            *reinterpret_cast<float4*>(stg + L.squad + rt * 8192) = make_float4(acc[rt][0], acc[rt][1], acc[rt][2], acc[rt][3]);
}
// GELU pass over the staging image: a thread owns 8 columns (cv) of rows rsub + 32 k.  dact = gelu'(pre) and act = gelu(pre) go to
// global memory at once; act also into image `act_img` — behind a barrier when that image is part of the staging area (SYNC).
template <bool SYNC>
__device__ __forceinline__ void gelu_pass(const char* stg, const float* bias_g, bf16* dact_g, bf16* act_g, long ldg, char* act_img, int T) {
    const int cv = threadIdx.x & 15, rsub = threadIdx.x >> 4;
    const float4 b0 = *reinterpret_cast<const float4*>(bias_g + cv * 8), b1 = *reinterpret_cast<const float4*>(bias_g + cv * 8 + 4);
    const int s0 = rsub * 512 + (((2 * cv) ^ (rsub & 15)) << 4);          // (r & 15 == rsub & 15 for r = rsub + 32 k)
    uint4 keep0 = make_uint4(0, 0, 0, 0), keep1 = keep0, keep2 = keep0, keep3 = keep0;
    // a ROLLED loop: unrolled, the scheduler interleaves the 32 erf chains of a thread (~80 registers; see gelu_pass above)
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
        const int r = rsub + 32 * k;
        if (r < T) {
            const float4 x0 = *reinterpret_cast<const float4*>(stg + s0 + k * 16384), x1 = *reinterpret_cast<const float4*>(stg + ((s0 + k * 16384) ^ 16));
            const float x[8] = {x0.x + b0.x, x0.y + b0.y, x0.z + b0.z, x0.w + b0.w, x1.x + b1.x, x1.y + b1.y, x1.z + b1.z, x1.w + b1.w};
            Vec16<bf16> dact, act;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float e;
                const float cdf = 0.5f * (1.0f + erf_as(x[j] * 0.70710678118654752440f, e));
                act.v[j] = from_f32<bf16>(x[j] * cdf);
                dact.v[j] = from_f32<bf16>(cdf + x[j] * (0.39894228040143267794f * e));
                // two erf chains in flight, not eight: at four waves per SIMD the other waves hide a chain's latency, and the
                // registers of six more chains are what this kernel does not have
                if (j & 1) __builtin_amdgcn_sched_barrier(0);
            }
            st16<bf16>(dact_g + (long)r * ldg + cv * 8, dact);
            st16<bf16>(act_g + (long)r * ldg + cv * 8, act);
            const uint4 a = *reinterpret_cast<const uint4*>(&act);
            if (!SYNC) *reinterpret_cast<uint4*>(act_img + vec_off(r, cv)) = a;
            else if (k == 0) keep0 = a;
            else if (k == 1) keep1 = a;
            else if (k == 2) keep2 = a;
            else keep3 = a;
        }
    }
    if (SYNC) {
        lds_barrier();                      // everybody's staging reads are done: the image may overwrite the staging area
        if (rsub < T) *reinterpret_cast<uint4*>(act_img + vec_off(rsub, cv)) = keep0;
        if (rsub + 32 < T) *reinterpret_cast<uint4*>(act_img + vec_off(rsub + 32, cv)) = keep1;
        if (rsub + 64 < T) *reinterpret_cast<uint4*>(act_img + vec_off(rsub + 64, cv)) = keep2;
        if (rsub + 96 < T) *reinterpret_cast<uint4*>(act_img + vec_off(rsub + 96, cv)) = keep3;
    }
}

// "No load behind a store": loads and stores share ONE in-order counter (vmcnt) on this part, so waiting for a load also waits for every
// store issued before it — behind a copy_out or a GELU pass that is a full HBM write acknowledge (2-3 us with 512 workgroups storing
// at once), and a sample has ten such batches.  Hence: the three parameter vectors used before the first store batches go to
// registers in the prologue, the others wait in the LDS table, and a product's weight fragments are requested BEFORE the store
// batch in front of it and their wait pinned (touch_regs) in front of that batch's first store — except where 16 more registers do
// not exist (the out-dense fragments of the second hidden half: one load behind the GELU pass's stores per sample).
template <int NRT>
__global__ __launch_bounds__(NTHR, 4) void tail2_fwd_kernel(TailP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const P = smem; char* const Q = smem + IMG; char* const R = smem + 2 * IMG;
    float* red = reinterpret_cast<float*>(smem + OFF_RED);
    float* par = reinterpret_cast<float*>(smem + OFF_PAR);
    const int b = blockIdx.x, T = p.T;
    if (p.stagger > 0 && 2 * b >= (int)gridDim.x)
        for (int i = 0; i < p.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    const Lane L = make_lane();
    const int n0 = L.wave * 16, nl = L.nl, l15 = L.l15;
    const long row0 = (long)b * T;
    const ChanPad cpad = chan_pad<8>(nl, p.dhp, p.dht);
    if ((int)threadIdx.x < C) {
        const int c = threadIdx.x;
        const float *g3_or = p.head ? p.g3 : p.g1, *b3_or = p.head ? p.b3 : p.b1, *bt_or = p.head ? p.bt : p.bo;
        const float x0 = p.bout[c], x1 = p.g2[c], x2 = p.b2[c], x3 = g3_or[c], x4 = b3_or[c], x5 = p.bi[c], x6 = p.bi[C + c], x7 = bt_or[c];
        par[PAR_BOUT * C + c] = x0; par[PAR_G2 * C + c] = x1; par[PAR_B2 * C + c] = x2; par[PAR_G3 * C + c] = x3;
        par[PAR_B3 * C + c] = x4; par[PAR_BI * C + c] = x5; par[(PAR_BI + 1) * C + c] = x6; par[PAR_BT * C + c] = x7;
    }
    const float4 v_bo = *reinterpret_cast<const float4*>(p.bo + nl);
    const float4 v_g1 = *reinterpret_cast<const float4*>(p.g1 + nl), v_b1 = *reinterpret_cast<const float4*>(p.b1 + nl);
    WFrags<NKB> wf = load_wfrags<NKB>(p.WoT + (long)n0 * C, C, L.lane);      // att_out dense
    copy_in2(P, p.att + row0 * C, C, Q, p.xin + row0 * p.ld_x, p.ld_x, T);
    const DropKey dk1 = make_dropkey(p.rng, p.sid1, p.rate), dk2 = make_dropkey(p.rng, p.sid2, p.rate);     // (wave-uniform: scalar registers)
    // first round of the head's row gather: position and destination row of this thread's vector (later rounds load theirs in the loop)
    int gat_t = 0; long gat_dst = -1;
    if (p.head) {
        const int j = min((int)threadIdx.x, p.M * 16 - 1) >> 4;
        const long src = (long)b * p.M + j;
        gat_t = (int)p.mpos[src];
        gat_dst = p.hmap ? (long)p.hmap[src] : src;
    }
    const uint64_t idx0 = (uint64_t)((row0 + l15) * C + nl);       // dropout element index of this lane's quad in row tile 0 (+ rt * 16 * C)
    lds_barrier();
    float z[MAXRT][4];
    f32x4 acc[MAXRT];
    // ---- ao = att.Wo + bo ; z1 = drop(ao) + x_in (EasyDGL.py:113-115): att in P, x_in in Q, ao -> R ---------------------------------
    zero_acc<NRT>(acc);
    gemm<NRT>(wf, P, L, acc);
    wf = load_wfrags<NKB>(p.WiT + (long)n0 * C, C, L.lane);                   // inner dense, first half (used behind two store batches)
    {
        const float bv[4] = {v_bo.x, v_bo.y, v_bo.z, v_bo.w};
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            const int row = rt * 16 + l15;
            float v[4], xr[4], dv[4];
            ld_bf4(reinterpret_cast<const bf16*>(Q + L.quad + rt * 4096), xr);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = rbf(acc[rt][r] + bv[r]); dv[r] = v[r]; }
            drop_apply4(dk1, idx0 + (uint64_t)(rt * 16 * C), dv);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[rt][r] = dv[r] + xr[r];
            if (row < T) st_bf4(reinterpret_cast<bf16*>(R + L.quad + rt * 4096), v);
        }
    }
    lds_barrier();
    touch_regs(wf);                  // every load so far has arrived: stores may start
    copy_out(p.ao + row0 * C, C, R, T);
    // ---- a1 = LN1(z1) (EasyDGL.py:116) -> P (att is consumed) ---------------------------------------------------------------------------
    {
        float mean, rstd;
        joint_moments<8>(z, NRT, T, L.lane, red, mean, rstd, cpad);
        if (threadIdx.x == 0) { p.st1[2 * b] = mean; p.st1[2 * b + 1] = rstd; }
        const float gv[4] = {v_g1.x, v_g1.y, v_g1.z, v_g1.w}, ev[4] = {v_b1.x, v_b1.y, v_b1.z, v_b1.w};
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (z[rt][r] - mean) * rstd * gv[r] + ev[r];
            if (rt * 16 + l15 < T) st_bf4(reinterpret_cast<bf16*>(P + L.quad + rt * 4096), v);
        }
    }
    lds_barrier();
    copy_out(p.a1 + row0 * C, C, P, T);
    // ---- f = gelu(a1.Wi + bi), two halves of C hidden channels; o accumulates f.Wout half by half (EasyDGL.py:120-125) ----------------
    f32x4 acc3[MAXRT];
    zero_acc<NRT>(acc3);
    {   // first half: a1 . Wi[:, :C] -> staging Q + R (x_in / ao: consumed) -> f half in R -> acc3 = f . Wout[:C, :]
        zero_acc<NRT>(acc);
        gemm<NRT>(wf, P, L, acc);
        stage_acc<NRT>(Q, acc, L, T);
        WFrags<NKB> wo = load_wfrags<NKB>(p.WoutT + (long)n0 * 2 * C, 2 * C, L.lane);     // out dense, inputs of the first half
        wf = load_wfrags<NKB>(p.WiT + (long)(C + n0) * C, C, L.lane);                     // inner dense, second half
        lds_barrier();
        touch_regs(wo); touch_regs(wf);          // (acc3 is not live yet: both sets fit across the pass)
        gelu_pass<true>(Q, par + PAR_BI * C, p.pre_f + row0 * 2 * C, p.f + row0 * 2 * C, 2 * C, R, T);
        lds_barrier();
        gemm<NRT>(wo, R, L, acc3);
    }
    {   // second half
        zero_acc<NRT>(acc);
        gemm<NRT>(wf, P, L, acc);
        lds_barrier();                               // the first half's f image (R) is still the operand of somebody's product
        stage_acc<NRT>(Q, acc, L, T);
        lds_barrier();
        gelu_pass<true>(Q, par + (PAR_BI + 1) * C, p.pre_f + row0 * 2 * C + C, p.f + row0 * 2 * C + C, 2 * C, R, T);
        wf = load_wfrags<NKB>(p.WoutT + (long)n0 * 2 * C + C, 2 * C, L.lane);       // out dense, second half (behind the pass's stores: no registers for it earlier)
        lds_barrier();
        gemm<NRT>(wf, R, L, acc3);
    }
    if (p.head) wf = load_wfrags<NKB>(p.WtT + (long)n0 * C, C, L.lane);       // head transform
    // ---- o = . + bout ; z2 = drop(o) + a1 ; y = LN2(z2) (EasyDGL.py:126-128): o -> Q, y -> R ------------------------------------------------
    {
        const float4 v_bout = *reinterpret_cast<const float4*>(par + PAR_BOUT * C + nl);
        const float bv[4] = {v_bout.x, v_bout.y, v_bout.z, v_bout.w};
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            const int row = rt * 16 + l15;
            float v[4], xr[4], dv[4];
            ld_bf4(reinterpret_cast<const bf16*>(P + L.quad + rt * 4096), xr);
#pragma unroll
            for (int r = 0; r < 4; ++r) { v[r] = rbf(acc3[rt][r] + bv[r]); dv[r] = v[r]; }
            drop_apply4(dk2, idx0 + (uint64_t)(rt * 16 * C), dv);
#pragma unroll
            for (int r = 0; r < 4; ++r) z[rt][r] = dv[r] + xr[r];
            if (row < T) st_bf4(reinterpret_cast<bf16*>(Q + L.quad + rt * 4096), v);
        }
    }
    lds_barrier();
    if (p.head) touch_regs(wf);
    copy_out(p.o + row0 * C, C, Q, T);
    {
        float mean, rstd;
        joint_moments<8>(z, NRT, T, L.lane, red, mean, rstd, cpad);
        if (threadIdx.x == 0) { p.st2[2 * b] = mean; p.st2[2 * b + 1] = rstd; }
        const float4 v_g = *reinterpret_cast<const float4*>(par + PAR_G2 * C + nl), v_b = *reinterpret_cast<const float4*>(par + PAR_B2 * C + nl);
        const float gv[4] = {v_g.x, v_g.y, v_g.z, v_g.w}, ev[4] = {v_b.x, v_b.y, v_b.z, v_b.w};
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (z[rt][r] - mean) * rstd * gv[r] + ev[r];
            if (rt * 16 + l15 < T) st_bf4(reinterpret_cast<bf16*>(R + L.quad + rt * 4096), v);
        }
    }
    lds_barrier();
    copy_out(p.y + row0 * C, C, R, T);
    if (!p.head) return;
    // ---- head: so = gelu(y.Wt + bt) ; rows = LN3(so)[masked positions] (EasyDGL.py:136-146): staging = P + Q, so -> R, rows -> Q ----------
    zero_acc<NRT>(acc);
    gemm<NRT>(wf, R, L, acc);
    stage_acc<NRT>(P, acc, L, T);                    // (a1 and o are consumed: every wave is past the barrier behind LN2)
    lds_barrier();                                   // ... which also says that nobody reads y (R) any more
    gelu_pass<false>(P, par + PAR_BT * C, p.pre_t + row0 * C, p.so + row0 * C, C, R, T);
    lds_barrier();
#pragma unroll
    for (int rt = 0; rt < NRT; ++rt) ld_bf4(reinterpret_cast<const bf16*>(R + L.quad + rt * 4096), z[rt]);   // so, rounded through the activation dtype
    {
        float mean, rstd;
        joint_moments<8>(z, NRT, T, L.lane, red, mean, rstd, cpad);
        if (threadIdx.x == 0) { p.st3[2 * b] = mean; p.st3[2 * b + 1] = rstd; }
        const float4 v_g = *reinterpret_cast<const float4*>(par + PAR_G3 * C + nl), v_b = *reinterpret_cast<const float4*>(par + PAR_B3 * C + nl);
        const float gv[4] = {v_g.x, v_g.y, v_g.z, v_g.w}, ev[4] = {v_b.x, v_b.y, v_b.z, v_b.w};
#pragma unroll
        for (int rt = 0; rt < NRT; ++rt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (z[rt][r] - mean) * rstd * gv[r] + ev[r];
            if (rt * 16 + l15 < T) st_bf4(reinterpret_cast<bf16*>(Q + L.quad + rt * 4096), v);
        }
    }
    lds_barrier();
    // batch_gather of the masked positions (EasyDGL.py:142-143); dst: row compaction of the scoring (edgl_compact_scan's `inv`)
    if ((int)threadIdx.x < p.M * 16 && gat_dst >= 0)
        *reinterpret_cast<uint4*>(p.hrows + gat_dst * C + (threadIdx.x & 15) * 8) = *reinterpret_cast<const uint4*>(Q + vec_off(gat_t, threadIdx.x & 15));
    for (int v = threadIdx.x + NTHR; v < p.M * 16; v += NTHR) {
        const int j = v >> 4, cv = v & 15;
        const long src = (long)b * p.M + j;
        const int t = (int)p.mpos[src];
        const long dst = p.hmap ? (long)p.hmap[src] : src;
        if (dst >= 0) *reinterpret_cast<uint4*>(p.hrows + dst * C + cv * 8) = *reinterpret_cast<const uint4*>(Q + vec_off(t, cv));
    }
}

}  // namespace t2

// dst[n][k] = src[k][n]  (tf.layers.dense kernels are [in, out]; the MFMA A operand wants k contiguous)
__global__ void tail_pack_kernel(const bf16* Wo, const bf16* Wi, const bf16* Wout, const bf16* Wt, int C, bf16* pack) {
    const long cc = (long)C * C;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < 6 * cc; i += (long)gridDim.x * blockDim.x) {
        const bf16* src; long off; int K, N;
        if (i < cc) { src = Wo; off = 0; K = C; N = C; }
        else if (i < 3 * cc) { src = Wi; off = cc; K = C; N = 2 * C; }
        else if (i < 5 * cc) { src = Wout; off = 3 * cc; K = 2 * C; N = C; }
        else { src = Wt; off = 5 * cc; K = C; N = C; }
        const long j = i - off;
        const int n = (int)(j / K), k = (int)(j % K);
        pack[i] = src[(long)k * N + n];
    }
}

}  // namespace

// EDGL_TAIL2=0 / edgl_tail_variant(0): the one-workgroup-per-CU kernels for every shape (A/B switch)
static int g_tail2 = -1;
static bool tail2_enabled() {
    if (g_tail2 < 0) { const char* e = getenv("EDGL_TAIL2"); g_tail2 = (e && e[0] == '0') ? 0 : 1; }
    return g_tail2 != 0;
}
static int tail2_stagger() {
    static const int v = [] { const char* e = getenv("EDGL_TAIL2_STAGGER"); return e ? atoi(e) : 0; }();
    return v;
}
extern "C" int edgl_tail_variant(int variant) {
    const int prev = tail2_enabled() ? 1 : 0;
    if (variant >= 0) g_tail2 = variant ? 1 : 0;
    return prev;
}

extern "C" long edgl_tail_pack_elems(int C) { return 6L * C * C; }

extern "C" int edgl_tail_supported(int T, int C, int dtype) { return dtype == EDGL_BF16 && (C == 64 || C == 128) && T >= 1 && T <= 16 * MAXRT; }

extern "C" int edgl_tail_pack(const void* Wo, const void* Wi, const void* Wout, const void* Wt, int C, void* pack, void* stream) {
    EDGL_REQUIRE(Wo && Wi && Wout && Wt && pack, EDGL_ERR_NULL, "edgl_tail_pack: null pointer");
    EDGL_REQUIRE(C == 64 || C == 128, EDGL_ERR_SHAPE, "edgl_tail_pack: C=%d unsupported (64 or 128)", C);
    hipLaunchKernelGGL(tail_pack_kernel, dim3(96), dim3(256), 0, (hipStream_t)stream, (const bf16*)Wo, (const bf16*)Wi,
                       (const bf16*)Wout, (const bf16*)Wt, C, (bf16*)pack);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_tail_fwd_ct(const void* att, const void* xin, int ld_x, const void* pack, const float* bo, const float* bi,
                             const float* bout, const float* bt, const float* g1, const float* b1, const float* g2,
                             const float* b2, const float* g3, const float* b3, int B, int T, int C, float drop_rate,
                             const uint64_t* rng_state, uint32_t sid1, uint32_t sid2, const int64_t* masked_pos, int M, int head,
                             void* ao, void* a1, float* st1, void* pre_f, void* f, void* o, void* y, float* st2, void* pre_t,
                             void* so, float* st3, void* hrows, const int32_t* hrow_map, int dh_pad, int dh_true, int dtype,
                                void* stream) {
    EDGL_REQUIRE(att && xin && pack && bo && bi && bout && g1 && b1 && g2 && b2 && ao && a1 && st1 && pre_f && f && o && y && st2,
                 EDGL_ERR_NULL, "edgl_tail_fwd: null pointer");
    EDGL_REQUIRE(!head || (bt && g3 && b3 && masked_pos && pre_t && so && st3 && hrows && M >= 1), EDGL_ERR_NULL,
                 "edgl_tail_fwd: the head needs its weights, positions and outputs");
    EDGL_REQUIRE(B > 0 && edgl_tail_supported(T, C, dtype) && ld_x % 8 == 0, EDGL_ERR_SHAPE,
                 "edgl_tail_fwd: unsupported shape B=%d T=%d C=%d dtype=%d (bf16, C in {64,128}, T <= 112)", B, T, C, dtype);
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_tail_fwd: dropout without rng_state");
    EDGL_REQUIRE((dh_pad == 0 && dh_true == 0) || (dh_pad > 0 && (dh_pad & (dh_pad - 1)) == 0 && dh_true > 0 && dh_true <= dh_pad && C % dh_pad == 0),
                 EDGL_ERR_SHAPE, "edgl_tail_fwd: padded-channel spec dh_pad=%d dh_true=%d does not fit C=%d", dh_pad, dh_true, C);
    const bf16* pk = (const bf16*)pack;
    const long cc = (long)C * C;
    TailP p{(const bf16*)att, (const bf16*)xin, ld_x, pk, pk + cc, pk + 3 * cc, pk + 5 * cc, bo, bi, bout, bt, g1, b1, g2, b2, g3, b3,
            B, T, C, drop_rate, rng_state, sid1, sid2, masked_pos, M, head, hrow_map, (bf16*)ao, (bf16*)a1, (bf16*)pre_f, (bf16*)f, (bf16*)o,
            (bf16*)y, (bf16*)pre_t, (bf16*)so, (bf16*)hrows, st1, st2, st3, dh_pad, dh_true, tail2_stagger()};
    hipStream_t st = (hipStream_t)stream;
    const int nrt = (T + 15) / 16;
    if (tail2_enabled() && C == 128 && T <= t2::TMAX) {      // two workgroups per CU (see namespace t2)
        auto k2 = nrt <= 2 ? t2::tail2_fwd_kernel<2> : nrt <= 4 ? t2::tail2_fwd_kernel<4> : t2::tail2_fwd_kernel<MAXRT>;
        hipFuncSetAttribute((const void*)k2, hipFuncAttributeMaxDynamicSharedMemorySize, t2::SMEM_FWD);
        hipLaunchKernelGGL(k2, dim3(B), dim3(t2::NTHR), t2::SMEM_FWD, st, p);
        EDGL_LAUNCH_CHECK();
        return EDGL_OK;
    }
    auto k = C == 128 ? (nrt <= 2 ? tail_fwd_kernel<8, 2> : nrt <= 4 ? tail_fwd_kernel<8, 4> : tail_fwd_kernel<8, MAXRT>)
                      : (nrt <= 2 ? tail_fwd_kernel<4, 2> : nrt <= 4 ? tail_fwd_kernel<4, 4> : tail_fwd_kernel<4, MAXRT>);
    const size_t smem = C == 128 ? TailGeom<8>::SMEM_FWD : TailGeom<4>::SMEM_FWD;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, dim3(B), dim3(C == 128 ? 512 : 256), smem, st, p);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_tail_fwd(const void* att, const void* xin, int ld_x, const void* pack, const float* bo, const float* bi,
                             const float* bout, const float* bt, const float* g1, const float* b1, const float* g2,
                             const float* b2, const float* g3, const float* b3, int B, int T, int C, float drop_rate,
                             const uint64_t* rng_state, uint32_t sid1, uint32_t sid2, const int64_t* masked_pos, int M, int head,
                             void* ao, void* a1, float* st1, void* pre_f, void* f, void* o, void* y, float* st2, void* pre_t,
                             void* so, float* st3, void* hrows, const int32_t* hrow_map, int dtype, void* stream) {
    return edgl_tail_fwd_ct(att, xin, ld_x, pack, bo, bi, bout, bt, g1, b1, g2, b2, g3, b3, B, T, C, drop_rate, rng_state, sid1, sid2,
                            masked_pos, M, head, ao, a1, st1, pre_f, f, o, y, st2, pre_t, so, st3, hrows, hrow_map, 0, 0, dtype, stream);
}
extern "C" int edgl_tail_bwd_ct(const void* xin, int ld_x, const void* ao, const void* a1, const void* pre_f, const void* o,
                             const void* pre_t, const void* so, const float* st1, const float* st2, const float* st3,
                             const void* Wo, const void* Wi, const void* Wout, const void* Wt, const float* g1, const float* g2,
                             const float* g3, int B, int T, int C, float drop_rate, const uint64_t* rng_state, uint32_t sid1,
                             uint32_t sid2, int head, const void* d_rows, const int64_t* masked_pos, int M,
                             const int32_t* dy_rowmap, const void* d_y_in, void* d_pre_t, void* d_o, void* d_pre_f, void* d_ao,
                             void* d_res1, void* d_att, float* dg1, float* db1, float* dg2, float* db2, float* dg3, float* db3,
                             float* workspace, int dh_pad, int dh_true, int dtype, void* stream) {
    EDGL_REQUIRE(xin && ao && a1 && pre_f && o && st1 && st2 && Wo && Wi && Wout && g1 && g2 && d_o && d_pre_f && d_ao && d_res1 &&
                 d_att && dg1 && db1 && dg2 && db2 && workspace, EDGL_ERR_NULL, "edgl_tail_bwd: null pointer");
    EDGL_REQUIRE(head ? (pre_t && so && st3 && Wt && g3 && d_rows && masked_pos && d_pre_t && dg3 && db3 && M >= 1 && M <= 256) : (d_y_in != nullptr),
                 EDGL_ERR_NULL, "edgl_tail_bwd: head needs its tensors (M <= 256); without it d_y_in is the upstream gradient");
    EDGL_REQUIRE(B > 0 && edgl_tail_supported(T, C, dtype) && ld_x % 8 == 0, EDGL_ERR_SHAPE,
                 "edgl_tail_bwd: unsupported shape B=%d T=%d C=%d dtype=%d", B, T, C, dtype);
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_tail_bwd: dropout without rng_state");
    EDGL_REQUIRE((dh_pad == 0 && dh_true == 0) || (dh_pad > 0 && (dh_pad & (dh_pad - 1)) == 0 && dh_true > 0 && dh_true <= dh_pad && C % dh_pad == 0),
                 EDGL_ERR_SHAPE, "edgl_tail_bwd: padded-channel spec dh_pad=%d dh_true=%d does not fit C=%d", dh_pad, dh_true, C);
    float* part1 = workspace; float* part2 = workspace + (long)B * 2 * C; float* part3 = workspace + (long)B * 4 * C;
    TailBwdP p{(const bf16*)xin, ld_x, (const bf16*)ao, (const bf16*)a1, (const bf16*)pre_f, (const bf16*)o, (const bf16*)pre_t,
               (const bf16*)so, st1, st2, st3, (const bf16*)Wo, (const bf16*)Wi, (const bf16*)Wout, (const bf16*)Wt, g1, g2, g3, B, T, C,
               drop_rate, rng_state, sid1, sid2, head, (const bf16*)d_rows, masked_pos, M, dy_rowmap, (const bf16*)d_y_in,
               (bf16*)d_pre_t, (bf16*)d_o, (bf16*)d_pre_f, (bf16*)d_ao, (bf16*)d_res1, (bf16*)d_att, part1, part2, part3, dh_pad, dh_true};
    hipStream_t st = (hipStream_t)stream;
    const int nrt = (T + 15) / 16;
    auto k = C == 128 ? (nrt <= 2 ? tail_bwd_kernel<8, 2> : nrt <= 4 ? tail_bwd_kernel<8, 4> : tail_bwd_kernel<8, MAXRT>)
                      : (nrt <= 2 ? tail_bwd_kernel<4, 2> : nrt <= 4 ? tail_bwd_kernel<4, 4> : tail_bwd_kernel<4, MAXRT>);
    const size_t smem = C == 128 ? TailGeom<8>::SMEM_BWD : TailGeom<4>::SMEM_BWD;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    hipLaunchKernelGGL(k, dim3(B), dim3(C == 128 ? 512 : 256), smem, st, p);
    EDGL_LAUNCH_CHECK();
    // (dbeta | dgamma) per sample -> parameter gradients, fixed order (deferred-reduction aware)
    auto red2 = [&](float* part, float* dg, float* db) -> int {
        if (dg == db + C) return edgl_reduce_rows(part, B, 2 * C, 2L * C, db, 0, st);
        int rc = edgl_reduce_rows(part, B, C, 2L * C, db, 0, st);
        if (rc) return rc;
        return edgl_reduce_rows(part + C, B, C, 2L * C, dg, 0, st);
    };
    int rc = red2(part1, dg1, db1);
    if (rc) return rc;
    rc = red2(part2, dg2, db2);
    if (rc) return rc;
    if (head) rc = red2(part3, dg3, db3);
    return rc;
}
extern "C" int edgl_tail_bwd(const void* xin, int ld_x, const void* ao, const void* a1, const void* pre_f, const void* o,
                             const void* pre_t, const void* so, const float* st1, const float* st2, const float* st3,
                             const void* Wo, const void* Wi, const void* Wout, const void* Wt, const float* g1, const float* g2,
                             const float* g3, int B, int T, int C, float drop_rate, const uint64_t* rng_state, uint32_t sid1,
                             uint32_t sid2, int head, const void* d_rows, const int64_t* masked_pos, int M,
                             const int32_t* dy_rowmap, const void* d_y_in, void* d_pre_t, void* d_o, void* d_pre_f, void* d_ao,
                             void* d_res1, void* d_att, float* dg1, float* db1, float* dg2, float* db2, float* dg3, float* db3,
                             float* workspace, int dtype, void* stream) {
    return edgl_tail_bwd_ct(xin, ld_x, ao, a1, pre_f, o, pre_t, so, st1, st2, st3, Wo, Wi, Wout, Wt, g1, g2, g3, B, T, C, drop_rate, rng_state,
                            sid1, sid2, head, d_rows, masked_pos, M, dy_rowmap, d_y_in, d_pre_t, d_o, d_pre_f, d_ao, d_res1, d_att, dg1, db1,
                            dg2, db2, dg3, db3, workspace, 0, 0, dtype, stream);
}

extern "C" long edgl_tail_bwd_workspace(int B, int C) { return 6L * B * C; }

#ifdef EDGL_PHASE_TIMING
extern "C" int edgl_debug_phase_cycles_tail(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// EXPERIMENT, NOT BUILT INTO THE LIBRARY (round 6, DESIGN.md rule 73).  Correct (tests/test_gpu_ops.py -k gemm with EDGL_GEMMW_MIN_K=512;
// tools/gemmw_bench.py checks against torch) and faster ALONE at the recipe's largest projection — [15 872, 1536] . [1536, 2048]:
// 146 -> 114 us (0.87 PFLOP/s), its dX 135 -> 125 — equal or slower at K <= 1024 / fewer tiles, and SLOWER INSIDE the recipe's step
// (1.842 -> 1.866 ms, A/B x 2): a workgroup that takes a whole CU (512 registers per wave, 128 KB of LDS) cannot start beside the
// side-stream kernels that run under the projection (rule 45 again).  To try it: add the file to build.py's SOURCES and call
// edgl_gemmw_try() at the top of edgl_gemm2_try_strip() for flags & ~EDGL_EPI_BIAS == 0.
// bf16 GEMM for the LARGE projections (K >= 512, N a multiple of 256: the 512-unit recipes and config 3 — runme.sh:15-23,
// BASELINE.json configs[2]):  C[M,N] = A[M,K] . B (+ bias), f32 accumulate, bf16 out.
//
// Why another tile kernel.  The 128 x 128 tiles of k_gemm2.hip re-read A once per 128 output columns and B once per 128 output rows:
// 64 flop per byte moved from the L2 into the LDS — the QKVT projection of the recipe ([15 872, 1536] . [1536, 2048], 100 GFLOP) moves
// 1.56 GB that way and runs at the L2's ~10 TB/s, not at the matrix pipe's rate (145 us = 0.28 of the MFMA peak).  Here a workgroup
// owns a 256 x 256 tile (128 flop per byte), 4 waves = one per SIMD with a 128 x 128 accumulator each (sixteen 32 x 32 tiles = all 256
// AGPRs), operands staged with LDS-direct loads (global_load_lds_dwordx4, the form of k_score_stripw.hip: no staging registers, no
// ds_write, every vmcnt placed by hand) through a ring of four 32-KB stages (BK = 32), one barrier per stage.
//
// LDS image of a stage.  A tile [256 m][32 k]: 64-byte rows, the 16-byte chunk c of row m at position c ^ ((m >> 2) & 3) — the 16 rows
// of a ds_read_b128 lane group then hit 16 different bank groups; a load instruction moves 16 rows (1 KB), lane i fetching the chunk
// that belongs at its position (the swizzle is on the SOURCE side: the LDS destination of these loads is lane-linear).
// B as [K][N] (n contiguous: y = x . W, tf.layers.dense kernels): tile [32 k][256 n], 512-byte rows, the 64-byte granule g of row k at
// position g ^ (k & 3): the transpose reads (4 k-rows x 64 bytes per half wave) are conflict free.  B as [N][K] (k contiguous: dX =
// dz . W^T): the A layout.
#include <atomic>
#include <cstdlib>

#include "edgl_common.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef int v4i __attribute__((ext_vector_type(4)));

namespace gemmw {

constexpr int BM = 256, BN = 256, BK = 32, NTHR = 256;
constexpr int A_BYTES = BM * BK * 2, B_BYTES = BN * BK * 2, STAGEB = A_BYTES + B_BYTES;     // 16 + 16 KB
constexpr int NS = 4, AHEAD = 3;            // ring of stages; stage s + AHEAD is issued in the second half of iteration s
constexpr int SMEM_LOOP = NS * STAGEB;      // 128 KB
constexpr int OSTR = 132;                   // floats per staged output row of a wave (epilogue: 32 rows x 128 columns per pass)
constexpr int SMEM_EPI = 4 * 32 * OSTR * 4;
constexpr int SMEM = SMEM_LOOP > SMEM_EPI ? SMEM_LOOP : SMEM_EPI;

struct P {
    const bf16* A; const bf16* B; bf16* C; const float* bias;
    int M, N, K, lda, ldb, ldc;
    int ntn;        // column tiles
    int xcd;
};

__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(__builtin_amdgcn_readfirstlane(lds_dst)) : "memory");
}
#define GW_VM_WAIT(n) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(n) : "memory")
__device__ __forceinline__ unsigned lds_addr(const char* p) { return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p; }
__device__ __forceinline__ bf16x8 lds_frag(const char* p) { return *reinterpret_cast<const bf16x8*>(p); }
// B fragment of a 32x32x16 MFMA whose contraction index runs along the tile's rows: two transpose reads (k rows +0..3 | +4..7 of the
// lane half's eight rows: slot j <-> k = 8 hi + j, as the A fragment's plain 16-byte read has it)
__device__ __forceinline__ bf16x8 lds_tr8(const char* p, int row8_off) {
    typedef __attribute__((ext_vector_type(4))) short s4;
    const s4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)p);
    const s4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(p + row8_off));
    const uint2 a = __builtin_bit_cast(uint2, v0), b = __builtin_bit_cast(uint2, v1);
    return __builtin_bit_cast(bf16x8, uint4{a.x, a.y, b.x, b.y});
}

// Staging of one stage (k0 .. k0 + 31): wave w issues A instructions 4w .. 4w+3 (16 rows each) and B instructions 4w .. 4w+3
// (KC: 16 n-rows each; !KC: 2 k-rows each).  Rows past M / N are clamped (their products land in accumulator rows / columns that are
// never stored).
template <bool B_KC>
struct Stager {
    const char* A_; const char* B_;
    long lda2_, ldb2_;
    int m0_, n0_, M_, N_, wave_, lane_;
    __device__ __forceinline__ void init(const P& p, int m0, int n0, int wave, int lane) {
        A_ = reinterpret_cast<const char*>(p.A); B_ = reinterpret_cast<const char*>(p.B); lda2_ = (long)p.lda * 2; ldb2_ = (long)p.ldb * 2;
        m0_ = m0; n0_ = n0; M_ = p.M; N_ = p.N; wave_ = __builtin_amdgcn_readfirstlane(wave); lane_ = lane;
    }
    __device__ __forceinline__ void piece(unsigned stage_lds, int k0, int j) {      // j = 0..3: A, 4..7: B
        if (j < 4) {
            const int q = 4 * wave_ + j, r = 16 * q + (lane_ >> 2), pos = lane_ & 3, c = pos ^ ((r >> 2) & 3);
            const int gm = min(m0_ + r, M_ - 1);
            glds16(A_ + (long)gm * lda2_ + (long)(k0 + 8 * c) * 2, stage_lds + q * 1024);
        } else if (B_KC) {
            const int q = 4 * wave_ + (j - 4), r = 16 * q + (lane_ >> 2), pos = lane_ & 3, c = pos ^ ((r >> 2) & 3);
            const int gn = min(n0_ + r, N_ - 1);
            glds16(B_ + (long)gn * ldb2_ + (long)(k0 + 8 * c) * 2, stage_lds + A_BYTES + q * 1024);
        } else {
            const int q = 4 * wave_ + (j - 4), kr = 2 * q + (lane_ >> 5), p32 = lane_ & 31, gp = p32 >> 2, sub = p32 & 3;
            const int gl = gp ^ (kr & 3);
            glds16(B_ + (long)(k0 + kr) * ldb2_ + (long)(n0_ + 32 * gl + 8 * sub) * 2, stage_lds + A_BYTES + q * 1024);
        }
    }
};

template <bool B_KC>
__global__ __launch_bounds__(NTHR, 1) void gemmw_kernel(P p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31, G = lane >> 4, s = lane & 15;
    const int ntm = (p.M + BM - 1) / BM, ntiles = ntm * p.ntn;
    // XCD-aware order (k_gemm2.hip xcd_virtual_id): consecutive virtual ids — the column tiles of one row block — run on one XCD's L2
    int vid = blockIdx.x;
    if (p.xcd && (gridDim.x & 7) == 0) vid = (int)(blockIdx.x & 7) * (int)(gridDim.x >> 3) + (int)(blockIdx.x >> 3);
    if (vid >= ntiles) return;
    const int tm = vid / p.ntn, tn = vid % p.ntn;
    const int m0 = tm * BM, n0 = tn * BN;
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned lds0 = lds_addr(smem);
    Stager<B_KC> stg;
    stg.init(p, m0, n0, wave, lane);
    const int nst = p.K / BK;

    // ---- per-lane LDS offsets -----------------------------------------------------------------------------------------------------
    // A fragment (mt, ks): row wm*128 + mt*32 + l31, chunk (2 ks + hi) ^ ((l31 >> 2) & 3)
    const int fa = (l31 >> 2) & 3;
    int offA[2], offB[4];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) offA[ks] = (wm * 128 + l31) * 64 + (((2 * ks + hi) ^ fa) * 16);
    if (B_KC) {   // the A layout over the n rows: row wn*128 + nt*32 + l31
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) offB[ks] = A_BYTES + (wn * 128 + l31) * 64 + (((2 * ks + hi) ^ fa) * 16);
        offB[2] = offB[3] = 0;
    } else {      // transpose reads: k row 8 hi + (s >> 2) (+ 4: second read — slots j = 0..7 <-> k = 8 hi + j, the order of the A
                  // fragment's plain read; + 16: ks = 1), granule (wn*4 + nt) ^ (k & 3)
        const int tz = 8 * hi + (s >> 2), q = tz & 3;
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) offB[nt] = A_BYTES + tz * 512 + ((wn * 4 + (nt ^ q)) * 64) + (16 * (G & 1) + 4 * (s & 3)) * 2;
    }

    f32x16 acc[4][4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // ---- prologue: stages 0 .. AHEAD-1 on their way --------------------------------------------------------------------------------
#pragma unroll
    for (int v = 0; v < AHEAD; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) stg.piece(lds0 + v * STAGEB, min(v, nst - 1) * BK, j);
    GW_VM_WAIT(8 * (AHEAD - 1));      // stage 0
    lds_barrier();

    bf16x8 af[2][4], bfr[2][4];       // [buffer][tile]
    auto fetch = [&](int buf, const char* st, int ks) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) af[buf][mt] = lds_frag(st + offA[ks] + mt * 32 * 64);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) {
            if (B_KC) bfr[buf][nt] = lds_frag(st + offB[ks] + nt * 32 * 64);
            else bfr[buf][nt] = lds_tr8(st + offB[nt] + ks * 16 * 512, 4 * 512);
        }
    };
    auto mma = [&](int buf) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[buf][mt], bfr[buf][nt], acc[mt][nt], 0, 0, 0);
    };
    fetch(0, smem, 0);
    int sl = 0;
#pragma clang loop unroll(disable)
    for (int st = 0; st < nst; ++st) {
        const char* cur = smem + sl * STAGEB;
        const int q1 = (sl + 1) & (NS - 1), q3 = (sl + AHEAD) & (NS - 1);
        // k-step 0 of stage st (fragments in buffer 0) beside the fetch of k-step 1
        fetch(1, cur, 1);
        mma(0);
        // stage st+1 has landed (this wave's share: the loads of stage st+2 may stay in flight), everybody is past the reads of stage st-1
        asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        // k-step 1 beside the fetch of stage st+1's first fragments and the loads of stage st+AHEAD (into the slot of stage st-1)
        fetch(0, smem + q1 * STAGEB, 0);
#pragma unroll
        for (int j = 0; j < 8; ++j) stg.piece(lds0 + q3 * STAGEB, min(st + AHEAD, nst - 1) * BK, j);
        mma(1);
        sl = q1;
    }
    GW_VM_WAIT(0);
    __syncthreads();

    // ---- epilogue: + bias, bf16, whole 256-byte row segments; one 32-row band of the wave per pass through its LDS image ---------
    float* so = reinterpret_cast<float*>(smem) + wave * 32 * OSTR;
    float bv[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) { const int gn = n0 + wn * 128 + nt * 32 + l31; bv[nt] = (p.bias && gn < p.N) ? p.bias[gn] : 0.f; }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) so[((r & 3) + 8 * (r >> 2) + 4 * hi) * OSTR + 32 * nt + l31] = acc[mt][nt][r] + bv[nt];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
        // a row = 128 floats = 32 lanes x float4: two rows per wave instruction
#pragma unroll 4
        for (int rr = 0; rr < 32; rr += 2) {
            const int row = rr + hi, gm = m0 + wm * 128 + mt * 32 + row, gn = n0 + wn * 128 + 4 * l31;
            const float4 v = *reinterpret_cast<const float4*>(so + row * OSTR + 4 * l31);
            if (gm < p.M && gn < p.N) {
                const Frag4<bf16> f = frag_from_acc<bf16>(f32x4{v.x, v.y, v.z, v.w});
                *reinterpret_cast<uint2*>(p.C + (long)gm * p.ldc + gn) = *reinterpret_cast<const uint2*>(&f);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    }
}

}  // namespace gemmw

// returns 1 if taken, 0 if the shape does not qualify, < 0 on error.  bf16, bias-only epilogue (flags checked by the caller).
int edgl_gemmw_try(const void* A, const void* B, void* C, int M, int N, int K, int lda, int ldb, int ldc, int b_kc, const float* bias,
                   hipStream_t st) {
    static const int on = getenv("EDGL_GEMMW") ? atoi(getenv("EDGL_GEMMW")) : 1;
    if (!on) return 0;
    // Measured (tools/gemmw_bench.py, DESIGN rule 73): it pays where the K loop is long and the tiles fill the chip more than once —
    // [15 872, 1536] . [1536, 2048]: 146 -> 114 us, its dX 135 -> 125; at K = 512 .. 1024 or with fewer tiles than ~1.5 rounds of the
    // CU count the 128 x 128 kernel (three workgroups per CU, short prologue) is as fast or faster: those shapes stay there.
    static const int min_k = getenv("EDGL_GEMMW_MIN_K") ? atoi(getenv("EDGL_GEMMW_MIN_K")) : 1536;
    if (!(M >= 2048 && K >= min_k && (K % gemmw::BK) == 0 && (N % gemmw::BN) == 0 && (lda % 8) == 0 && (ldb % 8) == 0 && (ldc % 4) == 0 &&
          (((uintptr_t)A | (uintptr_t)B) & 15) == 0 && ((uintptr_t)C & 7) == 0))
        return 0;
    if ((long)((M + gemmw::BM - 1) / gemmw::BM) * (N / gemmw::BN) < 384) return 0;
    gemmw::P p{(const bf16*)A, (const bf16*)B, (bf16*)C, bias, M, N, K, lda, ldb, ldc, N / gemmw::BN, 1};
    const int ntiles = ((M + gemmw::BM - 1) / gemmw::BM) * p.ntn;
    const int grid = (ntiles + 7) / 8 * 8;
    static std::atomic<uint64_t> done[2];
    auto launch = [&](auto kern, int which) {
        int dev = 0;
        const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
        if (!known || !(done[which].load(std::memory_order_acquire) & (1ull << dev))) {
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, gemmw::SMEM);
            if (known) done[which].fetch_or(1ull << dev, std::memory_order_release);
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(gemmw::NTHR), gemmw::SMEM, st, p);
    };
    if (b_kc) launch(gemmw::gemmw_kernel<true>, 0);
    else launch(gemmw::gemmw_kernel<false>, 1);
    EDGL_LAUNCH_CHECK();
    return 1;
}

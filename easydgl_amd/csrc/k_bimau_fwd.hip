// K3 forward: fused BiMAU attention — BiMAU.__call__ (temporal.py:404-452) with MAU.intensity
// (temporal.py:281-315) inlined.  See bimau_common.h for the register-layout scheme.
#include "bimau_fwd_impl.h"

namespace {
using namespace bimau;

template <typename T>
int dispatch_dt(FwdP p, hipStream_t st) {
    const int dh = p.C / p.H;
    if (dh == 16) return dispatch_nt<T, 1>(p, st);
    if (dh == 32) return dispatch_nt<T, 2>(p, st);
    edgl_set_error("edgl_bimau_fwd: head dim %d not supported (16, 32, 64 or 128)", dh);
    return EDGL_ERR_SHAPE;
}

}  // namespace

extern "C" long edgl_bimau_pack_bytes(int C, int H, int E, int dtype) {
    const int dh = C / (H > 0 ? H : 1);
    return (long)(dtype == EDGL_BF16 ? bimau::pack_dims<bf16>(dh, E).bytes : bimau::pack_dims<float>(dh, E).bytes);
}

extern "C" long edgl_bimau_saved_bytes(int B, int T, int C, int H, int dtype) {
    if (H <= 0 || C % H) return -1;
    return (long)bimau::saved_layout(B, T, C, H, dtype == EDGL_BF16 ? 2 : 4).bytes;
}

extern "C" int edgl_bimau_pack(const float* W1, const float* b1, const float* w, const float* scaling, int C, int H,
                               int E, void* pack, int dtype, void* stream) {
    EDGL_REQUIRE(W1 && b1 && w && scaling && pack, EDGL_ERR_NULL, "edgl_bimau_pack: null pointer");
    EDGL_REQUIRE(H > 0 && C % H == 0 && E >= 1 && E <= bimau::EP, EDGL_ERR_SHAPE, "edgl_bimau_pack: bad C=%d H=%d E=%d", C, H, E);
    const int dh = C / H;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == EDGL_F32) hipLaunchKernelGGL((bimau::pack_kernel<float>), dim3(16), dim3(256), 0, st, W1, b1, w, scaling, dh, E, (char*)pack);
    else if (dtype == EDGL_BF16) hipLaunchKernelGGL((bimau::pack_kernel<bf16>), dim3(16), dim3(256), 0, st, W1, b1, w, scaling, dh, E, (char*)pack);
    else { edgl_set_error("edgl_bimau_pack: bad dtype %d", dtype); return EDGL_ERR_DTYPE; }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

extern "C" int edgl_bimau_fwd_zr(const void* qkvt, const void* resid, int ld_res, const int64_t* ids, const float* spans,
                                 const uint8_t* marks, const void* pack, int B, int T, int C, int H, int E,
                                 float drop_rate, const uint64_t* rng_state, uint32_t stream_id, void* out,
                                 float* lam_out, void* saved, float* zero_rows, int flags, int dtype, void* stream);
extern "C" int edgl_bimau_fwd_db(const void* qkvt, const void* resid, int ld_res, const int64_t* ids, const float* spans,
                                 const uint8_t* marks, const void* pack, int B, int T, int C, int H, int E,
                                 float drop_rate, const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale,
                                 void* out, float* lam_out, void* saved, float* zero_rows, int flags, int dtype, void* stream);

extern "C" int edgl_bimau_fwd_ord(const void* qkvt, const void* resid, int ld_res, const int64_t* ids, const float* spans,
                                  const uint8_t* marks, const void* pack, int B, int T, int C, int H, int E,
                                  float drop_rate, const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale,
                                  void* out, float* lam_out, void* saved, float* zero_rows, const int32_t* order, int flags, int dtype,
                                  void* stream);

// Keep bits of the attention dropout of one (rng state, stream id): bimau_common.h.  0 bytes: this shape has no stored-bits form
// (more than 8 key tiles) and the kernels hash.
extern "C" long edgl_bimau_dropbits_bytes(int B, int T, int H) {
    if (B <= 0 || T <= 0 || H <= 0) return -1;
    const int nt = (T + 15) / 16;
    return nt <= 8 ? (long)B * H * nt * 64 * (long)sizeof(uint32_t) : 0;
}
extern "C" int edgl_bimau_dropbits(int B, int T, int H, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                                   uint32_t* bits, void* stream) {
    EDGL_REQUIRE(rng_state && bits, EDGL_ERR_NULL, "edgl_bimau_dropbits: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0 && H > 0 && T <= 128, EDGL_ERR_SHAPE, "edgl_bimau_dropbits: bad shape B=%d T=%d H=%d (T <= 128)", B, T, H);
    EDGL_REQUIRE((double)B * H * T * T < 4294967296.0, EDGL_ERR_SHAPE, "edgl_bimau_dropbits: H*B*T*T must be < 2^32");
    const int nt = (T + 15) / 16;
    const long njobs = (long)B * H * nt;
    const dim3 grid((unsigned)((njobs + 3) / 4));
    hipStream_t st = (hipStream_t)stream;
    switch (nt) {
#define EDGL_DB_CASE(N) case N: hipLaunchKernelGGL((bimau::dropbits_kernel<N>), grid, dim3(256), 0, st, rng_state, stream_id, drop_rate, T, njobs, bits); break;
        EDGL_DB_CASE(1) EDGL_DB_CASE(2) EDGL_DB_CASE(3) EDGL_DB_CASE(4) EDGL_DB_CASE(5) EDGL_DB_CASE(6) EDGL_DB_CASE(7) EDGL_DB_CASE(8)
#undef EDGL_DB_CASE
    }
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}
extern "C" int edgl_bimau_fwd(const void* qkvt, const void* resid, int ld_res, const int64_t* ids, const float* spans,
                              const uint8_t* marks, const void* pack, int B, int T, int C, int H, int E,
                              float drop_rate, const uint64_t* rng_state, uint32_t stream_id, void* out,
                              float* lam_out, void* saved, int flags, int dtype, void* stream) {
    return edgl_bimau_fwd_zr(qkvt, resid, ld_res, ids, spans, marks, pack, B, T, C, H, E, drop_rate, rng_state, stream_id, out, lam_out,
                             saved, nullptr, flags, dtype, stream);
}
// edgl_bimau_fwd that also fills `zero_rows` ([H*B, T, E] f32, may be NULL) with zeros: the forward is VALU bound and its stores are
// free, a memset of the 26 MB d lambda buffer of the headline step is not.
extern "C" int edgl_bimau_fwd_zr(const void* qkvt, const void* resid, int ld_res, const int64_t* ids, const float* spans,
                                 const uint8_t* marks, const void* pack, int B, int T, int C, int H, int E,
                                 float drop_rate, const uint64_t* rng_state, uint32_t stream_id, void* out,
                                 float* lam_out, void* saved, float* zero_rows, int flags, int dtype, void* stream) {
    return edgl_bimau_fwd_db(qkvt, resid, ld_res, ids, spans, marks, pack, B, T, C, H, E, drop_rate, rng_state, stream_id, nullptr, 0.f, out,
                             lam_out, saved, zero_rows, flags, dtype, stream);
}
// edgl_bimau_fwd_zr with the stored keep bits of the attention dropout (edgl_bimau_dropbits on the SAME rng state, stream id, rate
// and shape; NULL = hash): the kernels that have a stored-bits form read them, the others hash — the same masks either way.
// qk_scale: the score scale (0 = 1 / sqrt(dh), temporal.py:422) — a model whose head dim d is not one the kernels tile (the
// reference's default --num_units 50 --num_heads 1) runs with zero-padded channels at the next supported head dim and 1 / sqrt(d).
extern "C" int edgl_bimau_fwd_db(const void* qkvt, const void* resid, int ld_res, const int64_t* ids, const float* spans,
                                 const uint8_t* marks, const void* pack, int B, int T, int C, int H, int E,
                                 float drop_rate, const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale,
                                 void* out, float* lam_out, void* saved, float* zero_rows, int flags, int dtype, void* stream) {
    return edgl_bimau_fwd_ord(qkvt, resid, ld_res, ids, spans, marks, pack, B, T, C, H, E, drop_rate, rng_state, stream_id, dropbits, qk_scale,
                              out, lam_out, saved, zero_rows, nullptr, flags, dtype, stream);
}

// Launch order of the attention kernels' (sample, head) jobs.  The kernels of the headline family leave out the key tiles in front
// of a sequence's first real key (bimau_common.h: KeyMask::kt0), so a job's time falls with its left padding — and a launch is
// two rounds of workgroups (1024 workgroups on 512 resident slots at the headline shape): in index order the slowest slot gets two
// long jobs and the launch takes as long as without the skip.  order[] lists the samples by falling key-tile count (ties: by
// index — a stable counting order, the same for every run): the long jobs start first, the short ones fill up behind them.
// Two small launches: the first real key of every sequence from wave ballots (one wave per sequence), then every sample's rank by
// counting (one workgroup).  (As ONE 1024-thread workgroup the kernel took 38 us beside the encoder: placement, not work.)
namespace {
// a) one wave per sequence: the key tiles from its first real key on (all of them when it has none)
__global__ __launch_bounds__(256) void job_tiles_kernel(const int64_t* ids, int B, int T, int32_t* nk) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int NT = (T + 15) / 16, NR = (T + 63) / 64;
    int first = -1;
    for (int r0 = 0; r0 < NR; r0 += 4) {   // T <= 256 keys per round: four loads in flight
        int64_t v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = ids[(long)b * T + min(lane + 64 * (r0 + j), T - 1)];
#pragma unroll
        for (int j = 3; j >= 0; --j) {
            const int k = lane + 64 * (r0 + j);
            const uint64_t real = __ballot(k < T && v[j] != 0);
            if (real != 0ull && (first < 0 || 64 * (r0 + j) < first)) first = 64 * (r0 + j) + (int)__builtin_ctzll(real);
        }
        if (first >= 0) break;
    }
    if (lane == 0) nk[b] = first < 0 ? NT : NT - (first >> 4);
}
// b) one workgroup: every sample's rank by counting (key-tile count falling, index rising)
constexpr int ORD_THREADS = 256;
__global__ __launch_bounds__(ORD_THREADS) void job_rank_kernel(const int32_t* nk, int B, int32_t* order) {
    extern __shared__ __attribute__((aligned(16))) int nk_s[];
    for (int b = threadIdx.x; b < B; b += ORD_THREADS) nk_s[b] = nk[b];
    __syncthreads();
    for (int b = threadIdx.x; b < B; b += ORD_THREADS) {
        const int mine = nk_s[b];
        int rank = 0;
        int o = 0;
        // broadcast reads of 16 bytes, eight in flight (one LDS latency per 32 samples: read one by one the loop was 128 dependent
        // round trips to an LDS that the GEMM beside it keeps busy — 34 us)
        for (; o + 32 <= B; o += 32) {
            int4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const int4*>(nk_s + o + 4 * j);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int oo = o + 4 * j;
                rank += (v[j].x > mine || (v[j].x == mine && oo < b)) ? 1 : 0;
                rank += (v[j].y > mine || (v[j].y == mine && oo + 1 < b)) ? 1 : 0;
                rank += (v[j].z > mine || (v[j].z == mine && oo + 2 < b)) ? 1 : 0;
                rank += (v[j].w > mine || (v[j].w == mine && oo + 3 < b)) ? 1 : 0;
            }
        }
        for (; o < B; ++o) {
            const int v = nk_s[o];
            rank += (v > mine || (v == mine && o < b)) ? 1 : 0;
        }
        order[rank] = b;
    }
}
}  // namespace
// order: int32 [2 * B] — the launch order in the first B entries, the samples' key-tile counts behind them (scratch of the two launches)
extern "C" int edgl_bimau_job_order(const int64_t* ids, int B, int T, int32_t* order, void* stream) {
    EDGL_REQUIRE(ids && order, EDGL_ERR_NULL, "edgl_bimau_job_order: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0 && B <= 16384, EDGL_ERR_SHAPE, "edgl_bimau_job_order: bad shape B=%d T=%d (B <= 16384)", B, T);
    hipLaunchKernelGGL(job_tiles_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, ids, B, T, order + B);
    hipLaunchKernelGGL(job_rank_kernel, dim3(1), dim3(ORD_THREADS), (size_t)((B + 3) & ~3) * sizeof(int), (hipStream_t)stream, order + B, B, order);
    EDGL_LAUNCH_CHECK();
    return EDGL_OK;
}

// edgl_bimau_fwd_db with the launch order of the samples (edgl_bimau_job_order; NULL: index order).  The order changes WHEN a
// (sample, head) job runs, nothing of what it computes or where it writes.
extern "C" int edgl_bimau_fwd_ord(const void* qkvt, const void* resid, int ld_res, const int64_t* ids, const float* spans,
                                  const uint8_t* marks, const void* pack, int B, int T, int C, int H, int E,
                                  float drop_rate, const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale,
                                  void* out, float* lam_out, void* saved, float* zero_rows, const int32_t* order, int flags, int dtype,
                                  void* stream) {
    EDGL_REQUIRE(qkvt && resid && ids && spans && marks && pack && out && lam_out, EDGL_ERR_NULL,
                 "edgl_bimau_fwd: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0 && H > 0 && C % H == 0 && E >= 1 && E <= bimau::EP, EDGL_ERR_SHAPE,
                 "edgl_bimau_fwd: bad shape B=%d T=%d C=%d H=%d E=%d", B, T, C, H, E);
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_bimau_fwd: dropout without rng_state");
    EDGL_REQUIRE(ld_res % 4 == 0, EDGL_ERR_SHAPE, "edgl_bimau_fwd: ld_res must be a multiple of 4");
    EDGL_REQUIRE((double)B * H * T * T < 4294967296.0, EDGL_ERR_SHAPE, "edgl_bimau_fwd: H*B*T*T must be < 2^32");
    FwdP p{qkvt, resid, ld_res, ids, spans, marks, (const char*)pack, B, T, C, H, E, drop_rate, rng_state, stream_id,
           out, lam_out, nullptr, nullptr, nullptr, 4, flags & ~EDGL_MAU_NO_SKIP, dropbits, qk_scale, order, (flags & EDGL_MAU_NO_SKIP) ? 1 : 0};
    flags &= ~EDGL_MAU_NO_SKIP;
    hipStream_t st = (hipStream_t)stream;
    if (C / H == 64 || C / H == 128) {   // three-launch form: lambda is written by the intensity kernel — plain memset there
        if (zero_rows && hipMemsetAsync(zero_rows, 0, (size_t)H * B * T * E * sizeof(float), st) != hipSuccess) {
            edgl_set_error("edgl_bimau_fwd: memset failed");
            return EDGL_ERR_LAUNCH;
        }
    } else {
        p.zero_rows = zero_rows;
    }
    if (saved) {   // [H*B*T, dh] activation dtype | [H*B*T, 16] f32 (pre-softplus z)
        const bimau::SavedLayout sl = bimau::saved_layout(B, T, C, H, dtype == EDGL_BF16 ? 2 : 4);
        p.hin_out = (char*)saved + sl.off_hin;
        p.z_out = reinterpret_cast<float*>((char*)saved + sl.off_z);
    }
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_bimau_fwd: bad dtype %d", dtype);
    edgl_prof_begin(EDGL_KERNEL_BIMAU_FWD, st);
    int rc;
    if (C / H == 64 || C / H == 128) rc = bimau::big_fwd(p, dtype, st);
    else if (dtype == EDGL_F32) rc = dispatch_dt<float>(p, st);
    else rc = dispatch_dt<bf16>(p, st);
    edgl_prof_end(EDGL_KERNEL_BIMAU_FWD, st);
    return rc;
}

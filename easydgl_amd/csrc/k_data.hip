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

}  // namespace

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

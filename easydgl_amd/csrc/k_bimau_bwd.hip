// K3 backward as three passes.  The dependency chain of one query row is
//     key sweep 1 (dA, dG -> dlambda)  ->  intensity MLP backward (dz -> dH)  ->  key sweep 2 (dP -> dS -> dQ, dK, dT_)
// and holding both sweeps' state plus the MLP's in one wave costs >400 registers (one wave per SIMD).  Split by phase,
// every kernel keeps 2-4 waves per SIMD and the MLP (sigmoids of all dh*E channels) is evaluated once for both the input
// and the weight gradients:
//   X) sweep 1, one wave per (b, head): recomputes S, P, G' (lambda is saved by the forward), dA; writes dz = dlambda *
//      softplus'(z) per row, the row term  sum_k dP1*P  of the softmax backward, dscaling partials, and accumulates dV.
//   Y) intensity MLP backward over row tiles of (Hin, dz): dH partials per mark group and the per-workgroup partials of
//      dW1 / db1 / dw, reduced deterministically by edgl_reduce_rows.
//   Z) sweep 2, one wave per (b, head): recomputes S, P, G', dA and finishes dP = dP1 + dH.T_^T, dS, dQ, dK, dT_.
// Products that contract over the QUERY index use operands transposed by one MFMA against the identity.
#include <algorithm>
#include <cstdlib>

#include "bimau_bwd_impl.h"

#ifdef EDGL_PHASE_TIMING
__device__ unsigned long long g_phase_cycles[16];   // see edgl_common.h (PH_MARK): X uses slots 0-7, Z slots 8-15
#endif

namespace {
using namespace bimau;

// ------------------------------------------------------------------------------------------------------------------
// Y) intensity MLP backward: dH partials + weight-gradient partials
// ------------------------------------------------------------------------------------------------------------------
// registers a wave parks in its slab of the block reduction: dW tiles, the extra tiles (bf16) or db1 / interval sums (f32), dw sums
template <typename T>
constexpr int intensity_bwd_nreg(int DT) {
    return KY_ECH * DT * DT * 4 + (sizeof(T) == 2 ? (KY_ECH / 4) * DT * 4 : 2 * KY_ECH * DT) + KY_ECH * DT;
}

struct MlpP {
    const void* hin; const float* dz_ws; const float* spans; const char* pack;
    long R; int B, T, E; float* dh_ws; float* wpart; const float* dsc_part; long njobs;
    int slab_epi;   // block reduction through four register slabs (they fit the LDS) instead of wave turns on one accumulator
};

// EC: compile-time mark count (16: no guards around the mark blocks, so the scheduler may run the LDS operand reads of a block
// ahead of the previous block's arithmetic) or 0 (run-time p.E).
// bf16: the row sums of du for db1 and for the interval row of dW1 (du * span) come out of ONE more MFMA per channel tile against
// the operand hX = [1 | span_hi | span_lo | 0 ...] (span as two bf16 terms) instead of two VALU instructions per element: the
// kernel is VALU-issue bound, the matrix pipe is idle.
template <typename T, int DT, int EC>
__global__ __launch_bounds__(256, (sizeof(T) == 2 && DT == 1) ? 3 : 1) void intensity_bwd_kernel(MlpP p) {
    constexpr int dh = 16 * DT;
    constexpr int ECH = KY_ECH;
    constexpr bool MX = sizeof(T) == 2;
    const int E = EC ? EC : p.E;
    const int e0 = blockIdx.y * ECH;
    PH_DECL
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const PackDims pd = pack_dims<T>(dh, E);
    copy_pack_to_lds(smem, p.pack, pd.bytes);
    const int JE = pd.JE, NPAR = (dh + 3) * JE, NPARX = NPAR + EP;
    // block accumulator: dW1[u][j] at accs[j * (dh + 1) + u] (u on the lanes: conflict-free adds; the odd stride keeps the
    // j-major read-out conflict-free too), then the interval row, db1 and dw [JE] each
    const int NACC = JE * (dh + 4), WROW = JE * (dh + 1);
    float* accs = reinterpret_cast<float*>(smem + pd.bytes);
    for (int i = threadIdx.x; i < NACC; i += blockDim.x) accs[i] = 0.f;
    // wave-private scratch behind it: dz [16 rows][ECH] and the intervals [16] of the current row tile
    float* dzs = accs + ((NACC + 3) & ~3) + (threadIdx.x >> 6) * (16 * ECH + 16);
    float* sps = dzs + 16 * ECH;
    const T* W1T = reinterpret_cast<const T*>(smem);
    const T* W1X = reinterpret_cast<const T*>(smem + pd.off_w1x);
    const T* W1R = reinterpret_cast<const T*>(smem + pd.off_w1r);
    const float* fW = reinterpret_cast<const float*>(smem + pd.off_f32);
    const float* w1s = fW; const float* b1s = fW + JE; const float* wvs = fW + 2 * JE;
    (void)W1X; (void)w1s; (void)b1s;
    __syncthreads();
    PH_MARK(4);

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, g4 = (lane >> 4) * 4, l15 = lane & 15;
    const Frag4<T> ident = identity_frag<T>(lane);
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    // row tiles are 16 consecutive rows of the flat [H*B*T] row space (dz, H rows and the dH slabs are all indexed that
    // way); a tile may straddle two sequences — only the interval lookup needs (b, q) = row mod (B*T).  32-bit row arithmetic
    // (R < 2^31, host-checked): the head comes from a float product and one correction step instead of a 64-bit division
    // (which was ~125 instructions of every row tile).
    const int R = (int)p.R, ntile = (R + 15) / 16, BT = p.B * p.T;
    const float inv_bt = 1.0f / (float)BT;
    const float cx0 = l15 == 0 ? 1.0f : 0.0f, cx1 = l15 == 1 ? 1.0f : 0.0f, cx2 = l15 == 2 ? 1.0f : 0.0f;   // column selectors of hX
    (void)cx0; (void)cx1; (void)cx2;
    const T* hin = reinterpret_cast<const T*>(p.hin);

    f32x4 dW[ECH][DT][DT];  // [e-e0][d][ub]: tile (j-tile = e*DT+d, u-tile = ub), L(first=j, second=u)
    // bf16: L(first=j_local, second=c): mark ee of a group of four owns the columns c = 3 * (ee % 4) + {0: db1[j], 1 and 2: interval
    // row (hi, lo span terms)} of the group's tile — the B operand of mark ee is hX shifted by 3 * (ee % 4) lanes (one DPP move)
    f32x4 dWx[MX ? ECH / 4 : 1][DT];
    float adb[ECH][DT], adws[ECH][DT], adw[ECH][DT];
#pragma unroll
    for (int e = 0; e < ECH; ++e)
#pragma unroll
        for (int d = 0; d < DT; ++d) {
            adb[e][d] = 0.f; adws[e][d] = 0.f; adw[e][d] = 0.f;
            if constexpr (MX) dWx[e / 4][d] = zero4;
#pragma unroll
            for (int ub = 0; ub < DT; ++ub) dW[e][d][ub] = zero4;
        }
    if constexpr (!MX) dWx[0][0] = zero4;

    // Operands of a row tile — H rows, intervals, dz — are three unconditional loads (rows clamped into the arrays), fetched
    // one tile ahead; dz and the intervals go through the wave's LDS scratch, from where the mark loop reads the four rows
    // of its lane group as broadcasts (instead of 8 + 4 conditional global loads per lane and tile, which cost 35 of the
    // kernel's 100 us in round trips).
    static_assert(ECH == 8, "a lane stages two dz values of its row");
    struct TileOps { Frag4<T> hA[DT]; float span; float2 dz; };
    auto load_tile = [&](int t) {
        TileOps o;
        const int row = min(t * 16 + l15, R - 1);
        int sidx = row - (int)((float)row * inv_bt) * BT;   // row mod (B*T): the estimate of the head is off by at most one
        sidx = sidx < 0 ? sidx + BT : sidx;
        sidx = sidx >= BT ? sidx - BT : sidx;
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) o.hA[ub] = frag_ld<T>(hin + (long)row * dh + ub * 16 + g4);
        o.span = p.spans[sidx];
        o.dz = *reinterpret_cast<const float2*>(p.dz_ws + (long)row * EP + e0 + (g4 >> 1));   // lane group g: marks e0 + 2g, 2g + 1
        return o;
    };
    // "use" of a fetched tile: pins the wait for its loads to this point of the program
    auto touch_tile = [&](TileOps& o) {
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) {
            if constexpr (sizeof(T) == 2) { uint2& h = *reinterpret_cast<uint2*>(&o.hA[ub]); asm volatile("" : "+v"(h.x), "+v"(h.y)); }
            else { uint4& h = *reinterpret_cast<uint4*>(&o.hA[ub]); asm volatile("" : "+v"(h.x), "+v"(h.y), "+v"(h.z), "+v"(h.w)); }
        }
        asm volatile("" : "+v"(o.span), "+v"(o.dz.x), "+v"(o.dz.y));
    };
    const int tstep = (int)gridDim.x * 4, tfirst = (int)blockIdx.x * 4 + wave;
    TileOps cur;
    cur = load_tile(min(tfirst, ntile - 1));
    touch_tile(cur);   // complete before the loop: a tile pending at the loop entry makes every in-loop wait conservative (it would
                       // count the stores of the previous tile, i.e. wait for their acknowledges)
    for (int t = tfirst; t < ntile; t += tstep) {
        TileOps nxt = load_tile(t + tstep < ntile ? t + tstep : t);
        asm volatile("" ::: "memory");   // the prefetch stays at the top of the tile
        const int row0 = t * 16;
        const bool okA = row0 + l15 < R;   // row on the lane axis (A operand)
        // rows past the end of the row space (last tile) contribute nothing: zero H and dz there
        dzs[l15 * ECH + (g4 >> 1)] = okA ? cur.dz.x : 0.f;
        dzs[l15 * ECH + (g4 >> 1) + 1] = okA ? cur.dz.y : 0.f;
        if (lane < 16) sps[l15] = okA ? cur.span : 0.f;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        Frag4<T> hA[DT], hB[DT];
#pragma unroll
        for (int ub = 0; ub < DT; ++ub) {
            hA[ub] = okA ? cur.hA[ub] : frag_zero<T>();
            hB[ub] = frag_from_acc<T>(mma16(hA[ub], ident, zero4));  // L(first=row, second=u)
        }
        // bf16: span * w1s + b1 of every channel comes out of the matrix pipe (span_frag / W1X, bimau_common.h)
        Frag4<T> sfA;
        if constexpr (MX) sfA = span_frag(cur.span, lane);
        const float4 sp4 = *reinterpret_cast<const float4*>(sps + g4);
        const float spn[4] = {sp4.x, sp4.y, sp4.z, sp4.w};
        Frag4<T> hX;   // L(first=row, second=c): branch-free — hi = the span truncated to bf16 (exact), lo = span - hi (rounded by the pack)
        if constexpr (MX) {
            f32x4 hx;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float hi = __uint_as_float(__float_as_uint(spn[r]) & 0xffff0000u);
                hx[r] = fmaf(cx1, hi, fmaf(cx2, spn[r] - hi, cx0));
            }
            hX = frag_from_acc<T>(hx);
        }
        Frag4<T> hXv[4];
        if constexpr (MX) {
            hXv[0] = hX;
            const uint2 h0 = *reinterpret_cast<const uint2*>(&hX);
#define EDGL_ROW_SHR(x, n) (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(x), 0x110 + (n), 0xf, 0xf, true)
            uint2 h;
            h.x = EDGL_ROW_SHR(h0.x, 3); h.y = EDGL_ROW_SHR(h0.y, 3); *reinterpret_cast<uint2*>(&hXv[1]) = h;
            h.x = EDGL_ROW_SHR(h0.x, 6); h.y = EDGL_ROW_SHR(h0.y, 6); *reinterpret_cast<uint2*>(&hXv[2]) = h;
            h.x = EDGL_ROW_SHR(h0.x, 9); h.y = EDGL_ROW_SHR(h0.y, 9); *reinterpret_cast<uint2*>(&hXv[3]) = h;
#undef EDGL_ROW_SHR
        }
        f32x4 dHt[DT];   // dH[row][u] of this mark group, L(first=row, second=u)
#pragma unroll
        for (int ut = 0; ut < DT; ++ut) dHt[ut] = zero4;
        PH_MARK(5);
#pragma unroll
        for (int ee = 0; ee < ECH; ++ee) {
            const int e = e0 + ee;
            if (EC == 16 || e < E) {
                float dzr[4];   // dz[row g4 + r][e]: LDS broadcast
#pragma unroll
                for (int r = 0; r < 4; ++r) dzr[r] = dzs[(g4 + r) * ECH + ee];
#pragma unroll
                for (int d = 0; d < DT; ++d) {
                    const int jt = e * DT + d, j = jt * 16 + l15;
                    f32x4 a = zero4;  // Zpre[row][j], L(first=row, second=j)
                    if constexpr (MX) a = mma16(sfA, frag_ld<T>(W1X + (jt * 16 + l15) * XW + (g4 & 4)), a);
#pragma unroll
                    for (int ub = 0; ub < DT; ++ub)
                        a = mma16(hA[ub], frag_ld<T>(W1T + (jt * 16 + l15) * pd.LDW + ub * 16 + g4), a);
                    const float wv = wvs[j];
                    if constexpr (!MX) {
                        const float ws = w1s[j], bs = b1s[j];
#pragma unroll
                        for (int r = 0; r < 4; ++r) a[r] = fmaf(spn[r], ws, a[r]) + bs;
                    }
                    f32x4 du;
                    // the kernel is VALU-issue bound (3 waves per SIMD): three instructions per element after the sigmoid
                    // (dz*z, wv*(1-z) as one fma, their product) and the sums straight into their accumulators
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float z = sigmoid_pre(a[r]);
                        const float t2 = dzr[r] * z;
                        du[r] = t2 * fmaf(-z, wv, wv);
                        if constexpr (!MX) {
                            adb[ee][d] += du[r];
                            adws[ee][d] = fmaf(du[r], spn[r], adws[ee][d]);
                        }
                        adw[ee][d] += t2;
                    }
                    const Frag4<T> duf = frag_from_acc<T>(du);  // as A operand: A[m=j][kk=row]
#pragma unroll
                    for (int ub = 0; ub < DT; ++ub) dW[ee][d][ub] = mma16(duf, hB[ub], dW[ee][d][ub]);
                    if constexpr (MX) dWx[ee / 4][d] = mma16(duf, hXv[ee % 4], dWx[ee / 4][d]);
                    // dH[row][u] += sum_j du[row][j] W1[u][j]: du^T as A (contracting over j), W1 rows as B
                    const Frag4<T> duT = frag_from_acc<T>(mma16(duf, ident, zero4));   // L(first=j, second=row)
#pragma unroll
                    for (int ut = 0; ut < DT; ++ut)
                        dHt[ut] = mma16(duT, frag_ld<T>(W1R + (ut * 16 + l15) * pd.LDR + jt * 16 + g4), dHt[ut]);
                }
            }
            if (MX && DT == 1 && (ee & 1)) __builtin_amdgcn_sched_barrier(0);   // at most two marks' operands and temporaries in flight (168 registers)
        }
        // The next tile's operands are "used" here, in front of the stores: vmcnt counts loads and stores in issue order, and with the
        // (conditional) stores between the prefetch and its first use the wait for the prefetch also waited for the stores' acknowledges.
        touch_tile(nxt);
        // dH partial of this mark group: lane holds rows g4+r, channel u = ut*16 + l15
        float* dst = p.dh_ws + ((long)blockIdx.y * R + row0) * dh;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (row0 + g4 + r < R) {
#pragma unroll
                for (int ut = 0; ut < DT; ++ut) dst[(long)(g4 + r) * dh + ut * 16 + l15] = dHt[ut][r];
            }
        cur = nxt;
        PH_MARK(6);
    }
    // ---- block reduction ----------------------------------------------------------------------------------------------------------
    // Slab form: once every wave is past the mark loop the whole LDS (weight images + accumulator) is free, and the four waves park
    // their accumulator REGISTERS there as four [register][lane] slabs — independent, conflict-free stores.  The read-out then sums
    // the four slabs in a fixed order straight into the partial row of this workgroup.  (Wave turns on one LDS accumulator were ~80
    // dependent LDS round trips per wave, four turns in series: 40 % of the kernel in the phase stamps; LDS float atomics instead of
    // the read-modify-writes were slower still: 73 -> 94 us.)
    constexpr int RX = ECH * DT * DT * 4, RW = RX + (MX ? (ECH / 4) * DT * 4 : 2 * ECH * DT), NREG = RW + ECH * DT;
    static_assert(NREG == intensity_bwd_nreg<T>(DT), "host / device slab layouts differ");
    if (p.slab_epi) {
        __syncthreads();
        float* slab = reinterpret_cast<float*>(smem) + (size_t)wave * NREG * 64 + lane;
#pragma unroll
        for (int ee = 0; ee < ECH; ++ee)
#pragma unroll
            for (int d = 0; d < DT; ++d) {
#pragma unroll
                for (int ub = 0; ub < DT; ++ub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) slab[((((ee * DT + d) * DT + ub) * 4) + r) * 64] = dW[ee][d][ub][r];
                if constexpr (MX) {
                    if (ee % 4 == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) slab[(RX + ((ee / 4) * DT + d) * 4 + r) * 64] = dWx[ee / 4][d][r];
                    }
                } else {
                    slab[(RX + (ee * DT + d) * 2) * 64] = adb[ee][d];
                    slab[(RX + (ee * DT + d) * 2 + 1) * 64] = adws[ee][d];
                }
                slab[(RW + ee * DT + d) * 64] = adw[ee][d];
            }
        __syncthreads();
        const float* sl = reinterpret_cast<const float*>(smem);
        auto sum4 = [&](int reg, int ln) {   // the four waves' values of (register, lane), fixed order
            const float* q = sl + reg * 64 + ln;
            return (q[0] + q[NREG * 64]) + (q[2 * NREG * 64] + q[3 * NREG * 64]);
        };
        for (int i = threadIdx.x; i < NPAR; i += blockDim.x) {
            const int row = i / JE, j = i - row * JE;   // row < dh: dW1[row][j]; dh: interval row; dh + 1: db1; dh + 2: dw
            const int jt = j >> 4, jl = j & 15, e = jt / DT, d = jt - e * DT, ee = e - e0;
            if (ee < 0 || ee >= ECH) continue;     // every entry belongs to exactly one mark e -> one blockIdx.y
            float v;
            if (row < dh) {
                v = sum4((((ee * DT + d) * DT + (row >> 4)) * 4) + (jl & 3), (jl >> 2) * 16 + (row & 15));
            } else if (row == dh + 2) {
                const int reg = RW + ee * DT + d;
                v = (sum4(reg, jl) + sum4(reg, 16 + jl)) + (sum4(reg, 32 + jl) + sum4(reg, 48 + jl));
            } else if constexpr (MX) {
                const int reg = RX + ((ee / 4) * DT + d) * 4 + (jl & 3), ln = (jl >> 2) * 16 + 3 * (ee % 4);
                v = row == dh + 1 ? sum4(reg, ln) : sum4(reg, ln + 1) + sum4(reg, ln + 2);
            } else {
                const int reg = RX + (ee * DT + d) * 2 + (row == dh ? 1 : 0);
                v = (sum4(reg, jl) + sum4(reg, 16 + jl)) + (sum4(reg, 32 + jl) + sum4(reg, 48 + jl));
            }
            p.wpart[(long)blockIdx.x * NPARX + i] = v;
        }
    } else {
    // turn form (the slabs do not fit): waves take turns adding into the LDS accumulator (deterministic)
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int ee = 0; ee < ECH; ++ee) {
                const int e = e0 + ee;
                if (EC == 16 || e < E) {
#pragma unroll
                    for (int d = 0; d < DT; ++d) {
                        const int jt = e * DT + d;
                        const float sw = group_sum4(adw[ee][d]);
                        if (lane < 16) accs[WROW + 2 * JE + jt * 16 + l15] += sw;     // dw.flatten()[j]
                        if constexpr (MX) {
                            // columns 3 * (ee % 4) + {0: db1, 1: interval row hi, 2: lo} of the group's extra tile; register r is
                            // channel j = jt*16 + g4 + r.  Lane c0 + 1 takes the lo term from its neighbour (DPP row_shl:1).
                            const int c0 = 3 * (ee % 4);
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float x = dWx[ee / 4][d][r];
                                const float nb = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x101, 0xf, 0xf, true));
                                const int j = jt * 16 + g4 + r;
                                if (l15 == c0 || l15 == c0 + 1) accs[l15 == c0 ? WROW + JE + j : WROW + j] += l15 == c0 ? x : x + nb;   // db1[j] | dW1[dh][j]
                            }
                        } else {
                            const float sb = group_sum4(adb[ee][d]), sws = group_sum4(adws[ee][d]);
                            if (lane < 16) {
                                const int j = jt * 16 + l15;
                                accs[WROW + j] += sws;          // dW1[dh][j]   (interval row)
                                accs[WROW + JE + j] += sb;     // db1[j]
                            }
                        }
#pragma unroll
                        for (int ub = 0; ub < DT; ++ub)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int j = jt * 16 + g4 + r, u = ub * 16 + l15;
                                accs[j * (dh + 1) + u] += dW[ee][d][ub][r];  // dW1[u][j]
                            }
                    }
                }
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < NPAR; i += blockDim.x) {
        const int e = (i % JE) / dh;  // every entry belongs to exactly one mark e -> one blockIdx.y
        const int row = i / JE, j = i - row * JE;
        if (e >= e0 && e < e0 + ECH) p.wpart[(long)blockIdx.x * NPARX + i] = row < dh ? accs[j * (dh + 1) + row] : accs[WROW + (row - dh) * JE + j];
    }
    }
    // dscaling: fold this block's slice of kernel X's per-(b,head) partials into the same partial row.  All 256 threads
    // load (16 jobs x 16 marks per round), the 16 job slices are summed through LDS in a fixed order.
    if (blockIdx.y == 0) {
        const long per = (p.njobs + gridDim.x - 1) / gridDim.x;
        const long j0 = blockIdx.x * per, j1 = min(p.njobs, j0 + per);
        const int e = threadIdx.x & 15, sl = threadIdx.x >> 4;
        float a = 0.f;
        for (long j = j0 + sl; j < j1; j += 16) a += p.dsc_part[j * EP + e];
        __syncthreads();   // accs has been written out
        accs[sl * EP + e] = a;
        __syncthreads();
        if (threadIdx.x < EP) {
            float v = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) v += accs[k * EP + threadIdx.x];
            p.wpart[(long)blockIdx.x * NPARX + NPAR + threadIdx.x] = v;
        }
    }
    PH_MARK(7);
    PH_FLUSH(0);
}

// LDS bytes of the intensity backward kernel: the weight images + the dW1 / db1 / dw accumulators + per-wave tile scratch
template <typename T>
size_t intensity_bwd_lds(int dh, int E) {
    const int NACC = dh * E * (dh + 4);
    return pack_dims<T>(dh, E).bytes + ((size_t)((NACC + 3) & ~3) + 4 * (16 * KY_ECH + 16)) * sizeof(float);
}

template <typename T, int DT, int NT>
int launch_bwd(BwdP p, char* ws, float* dW1, float* db1, float* dw, float* dscaling, hipStream_t st) {
    constexpr int dh = 16 * DT, Tp = 16 * NT, LDT = Tp + 4;
    const PackDims pd = pack_dims<T>(dh, p.E);
    const WsLayout wl = ws_layout(p.B, p.T, p.C, p.H, p.E);
    p.dz_ws = reinterpret_cast<float*>(ws + wl.dz); p.dh_ws = reinterpret_cast<float*>(ws + wl.dh);
    p.rowdot_ws = reinterpret_cast<float*>(ws + wl.rowdot);
    p.dsc_part = reinterpret_cast<float*>(ws + wl.dsc); p.wpart = reinterpret_cast<float*>(ws + wl.wpart);
    const long jobs = (long)p.B * p.H;
    constexpr bool TR = sizeof(T) == 2;
    edgl_prof_begin(EDGL_KERNEL_BIMAU_BWD_ALL, st);
    // ---- X ----
    {
        const size_t wave_bytes = (2 * (size_t)Tp * dh + (size_t)Tp * EP + (TR ? 0 : (size_t)EP * LDT)) * sizeof(T) + (size_t)Tp * sizeof(float);
        int waves = 4;
        while (waves > 1 && waves * wave_bytes > 64 * 1024) waves >>= 1;
        const size_t smem = waves * wave_bytes;
        p.waves = waves;
        // bf16, head dim 16, BiMAU flags (the headline family): the variant with the flags compiled in
        constexpr bool SPEC = sizeof(T) == 2 && DT == 1;
        auto kern = p.E == 16 ? bimau_bwd_sweep1_kernel<T, DT, NT, 16> : bimau_bwd_sweep1_kernel<T, DT, NT, 0>;
        if constexpr (SPEC) {
            if (p.flags == 0) kern = p.E == 16 ? bimau_bwd_sweep1_kernel<T, DT, NT, 16, true, 0> : bimau_bwd_sweep1_kernel<T, DT, NT, 0, true, 0>;
            if constexpr (NT <= 8) {   // stored keep bits of the attention dropout (edgl_bimau_dropbits): the headline family
                // ... which also leaves out the all-padding key tiles in front of the first real key (SK, bimau_common.h)
                constexpr bool SKC = NT >= 2;
                const bool sk = SKC && p.flags == 0 && p.E == 16 && bimau_skip_enabled() && !p.noskip;
                const bool db = p.dbits && p.rate > 0.f;
                if (p.flags == 0 && p.E == 16 && db) kern = sk ? bimau_bwd_sweep1_kernel<T, DT, NT, 16, true, 0, true, false, SKC> : bimau_bwd_sweep1_kernel<T, DT, NT, 16, true, 0, true>;
                else if (sk) kern = bimau_bwd_sweep1_kernel<T, DT, NT, 16, true, 0, false, false, SKC>;
                if (p.tpp_desc) {   // d lambda of the TPP regulariser recomputed in the sweep (edgl_bimau_bwd_tpp checked the shape)
                    if (sk) kern = db ? bimau_bwd_sweep1_kernel<T, DT, NT, 16, true, 0, true, true, SKC> : bimau_bwd_sweep1_kernel<T, DT, NT, 16, true, 0, false, true, SKC>;
                    else kern = db ? bimau_bwd_sweep1_kernel<T, DT, NT, 16, true, 0, true, true> : bimau_bwd_sweep1_kernel<T, DT, NT, 16, true, 0, false, true>;
                }
            }
        }
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        hipLaunchKernelGGL(kern, dim3((unsigned)((jobs + waves - 1) / waves)), dim3(64 * waves), smem, st, p);
        EDGL_LAUNCH_CHECK();
    }
    // ---- Y ----
    {
        MlpP mp{p.hin, p.dz_ws, p.spans, p.pack, (long)p.B * p.H * p.T, p.B, p.T, p.E, p.dh_ws, p.wpart, p.dsc_part, jobs, 0};
        size_t smem_b = intensity_bwd_lds<T>(dh, p.E);
        const size_t slab_b = (size_t)4 * intensity_bwd_nreg<T>(DT) * 64 * sizeof(float);
        mp.slab_epi = slab_b <= std::max(smem_b, (size_t)52 * 1024);   // (three workgroups per CU stay resident up to 53 KB)
        if (mp.slab_epi) smem_b = std::max(smem_b, slab_b);
        EDGL_REQUIRE(smem_b <= 160 * 1024, EDGL_ERR_SHAPE, "edgl_bimau_bwd: intensity kernel needs %zu B of LDS", smem_b);
        EDGL_REQUIRE((long)p.B * p.H * p.T < (1l << 31) - 16, EDGL_ERR_SHAPE, "edgl_bimau_bwd: B*H*T = %ld rows exceed the 32-bit row index", (long)p.B * p.H * p.T);
        // (a compile-time mark count — no guards around the mark blocks — lets the scheduler hoist the LDS operand reads of all eight
        //  blocks: 168 registers no longer hold them, 25-37 spilled inside the tile loop, 81 -> 180 us.  Run-time E it is.)
        auto kb = intensity_bwd_kernel<T, DT, 0>;
        hipFuncSetAttribute((const void*)kb, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem_b);
        hipLaunchKernelGGL(kb, dim3(KY_BLOCKS, KY_NY), dim3(256), smem_b, st, mp);
        EDGL_LAUNCH_CHECK();
    }
    // ---- Z ----
    {
        const size_t wave_bytes = (3 * (size_t)Tp * dh + (size_t)Tp * EP + (TR ? 0 : (size_t)dh * LDT)) * sizeof(T) + (size_t)Tp * sizeof(float);
        int waves = 4;
        while (waves > 1 && waves * wave_bytes > 80 * 1024) waves >>= 1;
        const size_t smem = waves * wave_bytes;
        EDGL_REQUIRE(smem <= 160 * 1024, EDGL_ERR_SHAPE, "edgl_bimau_bwd: needs %zu B of LDS", smem);
        p.waves = waves;
        constexpr bool SPEC = sizeof(T) == 2 && DT == 1;
        auto kern = p.E == 16 ? bimau_bwd_sweep2_kernel<T, DT, NT, 16> : bimau_bwd_sweep2_kernel<T, DT, NT, 0>;
        if constexpr (SPEC) {
            if (p.flags == 0) kern = p.E == 16 ? bimau_bwd_sweep2_kernel<T, DT, NT, 16, KY_NY, true, 0> : bimau_bwd_sweep2_kernel<T, DT, NT, 0, KY_NY, true, 0>;
            if constexpr (NT <= 8) {
                constexpr bool SKC = NT >= 2;
                const bool sk = SKC && p.flags == 0 && p.E == 16 && bimau_skip_enabled() && !p.noskip;
                if (p.flags == 0 && p.E == 16 && p.dbits && p.rate > 0.f) kern = sk ? bimau_bwd_sweep2_kernel<T, DT, NT, 16, KY_NY, true, 0, true, 1, 0, SKC> : bimau_bwd_sweep2_kernel<T, DT, NT, 16, KY_NY, true, 0, true>;
                else if (sk) kern = bimau_bwd_sweep2_kernel<T, DT, NT, 16, KY_NY, true, 0, false, 1, 0, SKC>;
            }
        }
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        edgl_prof_begin(EDGL_KERNEL_BIMAU_BWD, st);
        hipLaunchKernelGGL(kern, dim3((unsigned)((jobs + waves - 1) / waves)), dim3(64 * waves), smem, st, p);
        edgl_prof_end(EDGL_KERNEL_BIMAU_BWD, st);
        edgl_prof_end(EDGL_KERNEL_BIMAU_BWD_ALL, st);
        EDGL_LAUNCH_CHECK();
    }
    const int JE = dh * p.E, NPAR = (dh + 3) * JE, NPARX = NPAR + EP;
    if (db1 == dW1 + (dh + 1) * JE && dw == db1 + JE && dscaling == dw + JE) {   // flat-arena layout: one reduction
        return edgl_reduce_rows(p.wpart, KY_BLOCKS, NPAR + p.E, NPARX, dW1, 0, st);
    }
    int rc = edgl_reduce_rows(p.wpart, KY_BLOCKS, (dh + 1) * JE, NPARX, dW1, 0, st);
    if (rc) return rc;
    rc = edgl_reduce_rows(p.wpart + (dh + 1) * JE, KY_BLOCKS, JE, NPARX, db1, 0, st);
    if (rc) return rc;
    rc = edgl_reduce_rows(p.wpart + (dh + 2) * JE, KY_BLOCKS, JE, NPARX, dw, 0, st);
    if (rc) return rc;
    rc = edgl_reduce_rows(p.wpart + NPAR, KY_BLOCKS, p.E, NPARX, dscaling, 0, st);
    if (rc) return rc;
    return EDGL_OK;
}

template <typename T, int DT>
int dispatch_nt(BwdP p, char* ws, float* dW1, float* db1, float* dw, float* dsc, hipStream_t st) {
    switch ((p.T + 15) / 16) {
        case 1: return launch_bwd<T, DT, 1>(p, ws, dW1, db1, dw, dsc, st);
        case 2: return launch_bwd<T, DT, 2>(p, ws, dW1, db1, dw, dsc, st);
        case 3: return launch_bwd<T, DT, 3>(p, ws, dW1, db1, dw, dsc, st);
        case 4: return launch_bwd<T, DT, 4>(p, ws, dW1, db1, dw, dsc, st);
        case 5: return launch_bwd<T, DT, 5>(p, ws, dW1, db1, dw, dsc, st);
        case 6: return launch_bwd<T, DT, 6>(p, ws, dW1, db1, dw, dsc, st);
        case 7: return launch_bwd<T, DT, 7>(p, ws, dW1, db1, dw, dsc, st);
        case 8: return launch_bwd<T, DT, 8>(p, ws, dW1, db1, dw, dsc, st);
        case 9: return launch_bwd<T, DT, 9>(p, ws, dW1, db1, dw, dsc, st);
        case 10: return launch_bwd<T, DT, 10>(p, ws, dW1, db1, dw, dsc, st);
        case 11: return launch_bwd<T, DT, 11>(p, ws, dW1, db1, dw, dsc, st);
        case 12: return launch_bwd<T, DT, 12>(p, ws, dW1, db1, dw, dsc, st);
        case 13: return launch_bwd<T, DT, 13>(p, ws, dW1, db1, dw, dsc, st);
    }
    edgl_set_error("edgl_bimau_bwd: T=%d not supported (T <= 208)", p.T);
    return EDGL_ERR_SHAPE;
}

}  // namespace

extern "C" long edgl_bimau_bwd_workspace(int B, int T, int C, int H, int E, int dtype) {
    (void)dtype;
    if (H <= 0 || C % H) return -1;
    return (long)ws_layout(B, T, C, H, E).total;
}

// Largest mark count (<= 16) one launch of the fused unit takes at this head dim and dtype — what a caller with more mark types
// splits them by (EDGL_MAU_DIAG_ZERO).  Head dims 16 / 32 are bounded by the LDS of the intensity backward kernel (f32, dh = 32:
// 11 marks); head dims 64 / 128 stream the weights (k_bimau_big.hip) and take the full 16.
extern "C" int edgl_bimau_mark_group(int C, int H, int dtype) {
    if (H <= 0 || C % H || (dtype != EDGL_F32 && dtype != EDGL_BF16)) return -1;
    const int dh = C / H;
    if (dh >= 64) return bimau::EP;
    for (int E = bimau::EP; E >= 1; --E) {
        const size_t b = dtype == EDGL_BF16 ? intensity_bwd_lds<bf16>(dh, E) : intensity_bwd_lds<float>(dh, E);
        if (b <= 160 * 1024) return E;
    }
    return -1;
}

extern "C" int edgl_bimau_bwd_db(const void* qkvt, const int64_t* ids, const float* spans, const uint8_t* marks,
                                 const void* pack, const void* d_out, const float* d_lam_ext, const float* lam,
                                 const void* saved, int B, int T, int C, int H, int E, float drop_rate,
                                 const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale, void* d_qkvt,
                                 float* dW1, float* db1, float* dw, float* dscaling, void* workspace, int flags, int dtype, void* stream);
extern "C" int edgl_bimau_bwd(const void* qkvt, const int64_t* ids, const float* spans, const uint8_t* marks,
                              const void* pack, const void* d_out, const float* d_lam_ext, const float* lam,
                              const void* saved, int B, int T, int C, int H, int E, float drop_rate,
                              const uint64_t* rng_state, uint32_t stream_id, void* d_qkvt, float* dW1, float* db1,
                              float* dw, float* dscaling, void* workspace, int flags, int dtype, void* stream) {
    return edgl_bimau_bwd_db(qkvt, ids, spans, marks, pack, d_out, d_lam_ext, lam, saved, B, T, C, H, E, drop_rate, rng_state, stream_id,
                             nullptr, 0.f, d_qkvt, dW1, db1, dw, dscaling, workspace, flags, dtype, stream);
}
// edgl_bimau_bwd with the stored keep bits the forward used (edgl_bimau_dropbits; NULL = hash — the same masks either way)
static int bimau_bwd_impl(const void* qkvt, const int64_t* ids, const float* spans, const uint8_t* marks,
                          const void* pack, const void* d_out, const float* d_lam_ext, const float* lam,
                          const void* saved, int B, int T, int C, int H, int E, float drop_rate,
                          const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale, void* d_qkvt,
                          float* dW1, float* db1, float* dw, float* dscaling, void* workspace, const void* tpp_desc, int tpp_M,
                          const float* tpp_sums, float tpp_coef, float* tpp_part, const int32_t* order, int flags, int dtype, void* stream) {
    EDGL_REQUIRE(qkvt && ids && spans && marks && pack && d_out && lam && saved && d_qkvt && dW1 && db1 && dw && dscaling && workspace,
                 EDGL_ERR_NULL, "edgl_bimau_bwd: null pointer");
    EDGL_REQUIRE(B > 0 && T > 0 && H > 0 && C % H == 0 && E >= 1 && E <= bimau::EP, EDGL_ERR_SHAPE,
                 "edgl_bimau_bwd: bad shape B=%d T=%d C=%d H=%d E=%d", B, T, C, H, E);
    EDGL_REQUIRE(dtype == EDGL_F32 || dtype == EDGL_BF16, EDGL_ERR_DTYPE, "edgl_bimau_bwd: bad dtype %d", dtype);
    EDGL_REQUIRE(drop_rate == 0.f || rng_state, EDGL_ERR_NULL, "edgl_bimau_bwd: dropout without rng_state");
    EDGL_REQUIRE((double)B * H * T * T < 4294967296.0, EDGL_ERR_SHAPE, "edgl_bimau_bwd: H*B*T*T must be < 2^32");
    BwdP p{};
    p.qkvt = qkvt; p.ids = ids; p.spans = spans; p.marks = marks; p.pack = (const char*)pack; p.d_out = d_out;
    p.d_lam_ext = d_lam_ext; p.lam = lam;
    const bimau::SavedLayout sl = bimau::saved_layout(B, T, C, H, dtype == EDGL_BF16 ? 2 : 4);
    p.hin = (const char*)saved + sl.off_hin;
    p.z = reinterpret_cast<const float*>((const char*)saved + sl.off_z);
    p.B = B; p.T = T; p.C = C; p.H = H; p.E = E; p.rate = drop_rate; p.rng = rng_state;
    p.noskip = (flags & EDGL_MAU_NO_SKIP) ? 1 : 0; flags &= ~EDGL_MAU_NO_SKIP;
    p.stream_id = stream_id; p.d_qkvt = d_qkvt; p.flags = flags; p.dbits = dropbits; p.qk_scale = qk_scale;
    p.tpp_desc = tpp_desc; p.tpp_M = tpp_M; p.tpp_sums = tpp_sums; p.tpp_coef = tpp_coef; p.tpp_part = tpp_part; p.order = order;
    EDGL_REQUIRE(!tpp_desc || (dtype == EDGL_BF16 && C / H == 16 && E == 16 && T <= 128 && flags == 0 && tpp_part && tpp_M > 0 && !d_lam_ext),
                 EDGL_ERR_SHAPE, "edgl_bimau_bwd_tpp: the fused TPP form exists for bf16, head dim 16, 16 marks, T <= 128, BiMAU flags "
                 "(C/H=%d E=%d T=%d flags=%d)", C / H, E, T, flags);
    hipStream_t st = (hipStream_t)stream;
    const int dh = C / H;
    char* ws = (char*)workspace;
    if (dtype == EDGL_F32) {
        if (dh == 16) return dispatch_nt<float, 1>(p, ws, dW1, db1, dw, dscaling, st);
        if (dh == 32) return dispatch_nt<float, 2>(p, ws, dW1, db1, dw, dscaling, st);
    } else {
        if (dh == 16) return dispatch_nt<bf16, 1>(p, ws, dW1, db1, dw, dscaling, st);
        if (dh == 32) return dispatch_nt<bf16, 2>(p, ws, dW1, db1, dw, dscaling, st);
    }
    if (dh == 64 || dh == 128) return bimau::big_bwd(p, ws, dW1, db1, dw, dscaling, dtype, st);
    edgl_set_error("edgl_bimau_bwd: head dim %d not supported (16, 32, 64 or 128)", dh);
    return EDGL_ERR_SHAPE;
}
extern "C" int edgl_bimau_bwd_db(const void* qkvt, const int64_t* ids, const float* spans, const uint8_t* marks,
                                 const void* pack, const void* d_out, const float* d_lam_ext, const float* lam,
                                 const void* saved, int B, int T, int C, int H, int E, float drop_rate,
                                 const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale, void* d_qkvt,
                                 float* dW1, float* db1, float* dw, float* dscaling, void* workspace, int flags, int dtype, void* stream) {
    return bimau_bwd_impl(qkvt, ids, spans, marks, pack, d_out, d_lam_ext, lam, saved, B, T, C, H, E, drop_rate, rng_state, stream_id,
                          dropbits, qk_scale, d_qkvt, dW1, db1, dw, dscaling, workspace, nullptr, 0, nullptr, 0.f, nullptr, nullptr, flags, dtype, stream);
}
// edgl_bimau_bwd_db with the TPP regulariser (MAU.biased_likelihood, temporal.py:317-333, at the masked positions of
// EasyDGL.py:157-175) evaluated by sweep 1, which holds lambda in registers: the term's gradient with respect to lambda
// (coef = ct_reg / H; tpp_sums[4] = the batch's next-event mark count as edgl_tpp_norm leaves it, NULL = the total of edgl_tpp_prep's
// per-sample counts) is computed from the slot data of
// edgl_tpp_prep instead of being read from a d lambda array, and the wave's share of the two loss sums goes to tpp_part [B*H, 2]
// (reduced by edgl_tpp_finish_parts).  bf16, head dim 16, 16 marks, T <= 128, BiMAU flags.
extern "C" int edgl_bimau_bwd_tpp(const void* qkvt, const int64_t* ids, const float* spans, const uint8_t* marks,
                                  const void* pack, const void* d_out, const void* tpp_desc, int M, const float* tpp_sums, float coef,
                                  float* tpp_part, const float* lam, const void* saved, int B, int T, int C, int H, int E, float drop_rate,
                                  const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale, void* d_qkvt,
                                  float* dW1, float* db1, float* dw, float* dscaling, void* workspace, int flags, int dtype, void* stream) {
    EDGL_REQUIRE(tpp_desc && tpp_part, EDGL_ERR_NULL, "edgl_bimau_bwd_tpp: null pointer");
    return bimau_bwd_impl(qkvt, ids, spans, marks, pack, d_out, nullptr, lam, saved, B, T, C, H, E, drop_rate, rng_state, stream_id,
                          dropbits, qk_scale, d_qkvt, dW1, db1, dw, dscaling, workspace, tpp_desc, M, tpp_sums, coef, tpp_part, nullptr, flags, dtype,
                          stream);
}
// edgl_bimau_bwd_db (tpp_desc == NULL: d_lam_ext as given, may be NULL) or edgl_bimau_bwd_tpp (tpp_desc != NULL: d_lam_ext must be
// NULL) with the launch order of the samples the forward used (edgl_bimau_job_order; NULL: index order) for the two sweeps.
extern "C" int edgl_bimau_bwd_ord(const void* qkvt, const int64_t* ids, const float* spans, const uint8_t* marks, const void* pack,
                                  const void* d_out, const float* d_lam_ext, const void* tpp_desc, int M, const float* tpp_sums, float coef,
                                  float* tpp_part, const float* lam, const void* saved, int B, int T, int C, int H, int E, float drop_rate,
                                  const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale, void* d_qkvt,
                                  float* dW1, float* db1, float* dw, float* dscaling, void* workspace, const int32_t* order, int flags,
                                  int dtype, void* stream) {
    EDGL_REQUIRE(!tpp_desc || tpp_part, EDGL_ERR_NULL, "edgl_bimau_bwd_ord: tpp_desc without tpp_part");
    return bimau_bwd_impl(qkvt, ids, spans, marks, pack, d_out, d_lam_ext, lam, saved, B, T, C, H, E, drop_rate, rng_state, stream_id,
                          dropbits, qk_scale, d_qkvt, dW1, db1, dw, dscaling, workspace, tpp_desc, M, tpp_sums, coef, tpp_part, order, flags,
                          dtype, stream);
}

#ifdef EDGL_PHASE_TIMING
extern "C" int edgl_debug_phase_cycles(unsigned long long* out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_phase_cycles), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase_cycles), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

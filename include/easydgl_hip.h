/*
 * libeasydgl_hip.so — C ABI of the MI355X (gfx950) kernels for EasyDGL's temporal-point-process
 * self-modulating-attention hot path.
 *
 * The reference (cchao0116/EasyDGL) has NO FFI: its hot path is a TensorFlow-1.x graph.  Each entry
 * point below therefore replaces a *group of TF ops*; the comment on each cites the reference lines
 * (paths relative to the reference repo root) whose arithmetic it reproduces.  INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all buffers;
 *    kernels never allocate, never synchronise, and launch on `stream` (a hipStream_t);
 *  - returns 0 on success, a negative EDGL_ERR_* otherwise (edgl_last_error() gives the text);
 *  - activations / GEMM operands are `dtype` (EDGL_F32 exact-f32 MFMA path, or EDGL_BF16 with f32
 *    accumulation); statistics, losses, gradients of parameters and optimizer state are always f32;
 *  - tensors are dense row-major; B = batch, T = positions (= seqslen+1), C = num_units,
 *    H = num_heads, dh = C/H, E = num_events, I = item-table rows (= num_items+1), M = masklen;
 *  - head-major stacking b' = head*B + b for every [H*B, ...] tensor (temporal.py:413-416);
 *  - dropout masks are a pure function of (rng_state[0]=seed, rng_state[1]=step, stream_id, element
 *    index); backward kernels regenerate them.  rng_state may be NULL when the rate is 0.
 */
#ifndef EASYDGL_HIP_H
#define EASYDGL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum { EDGL_F32 = 0, EDGL_BF16 = 1 } edgl_dtype;

enum {
    EDGL_OK = 0,
    EDGL_ERR_SHAPE = -1,       /* unsupported / inconsistent shape */
    EDGL_ERR_DTYPE = -2,
    EDGL_ERR_LAUNCH = -3,      /* hipGetLastError() != hipSuccess after a launch */
    EDGL_ERR_NULL = -4,
    EDGL_ERR_WORKSPACE = -5
};

/* GEMM epilogue flags */
enum {
    EDGL_EPI_BIAS = 1,        /* + bias[n]                                   */
    EDGL_EPI_GELU = 2,        /* erf-GELU (EasyDGL.py:19-32)                 */
    EDGL_EPI_SAVE_PRE = 4,    /* aux[m,n] = pre-activation (dtype)           */
    EDGL_EPI_MUL_DGELU = 8,   /* *= gelu'(aux[m,n])                          */
    EDGL_EPI_ACCUM = 16,      /* C += result                                 */
    EDGL_EPI_OUT_F32 = 32,    /* C is f32 regardless of dtype                */
    EDGL_EPI_RELU = 64        /* max(x, 0) (FeedForward inner layer, Base.py:73) */
};

const char* edgl_last_error(void);
int edgl_version(void);

/* Profiling hook (measurement only): the NEXT launch of kernel (group) `kernel_id` on the calling thread records the two
 * hipEvent_t handles immediately before / after it on its launch stream, then the slot clears.
 *   SCORE_BWD_ROWS : the row-side scoring pass (K5);  BIMAU_BWD : sweep 2 of the BiMAU backward alone;
 *   BIMAU_FWD      : every kernel of one edgl_bimau_fwd call (K3 forward: temporal.py:404-452 + 281-315);
 *   BIMAU_BWD_ALL  : the three backward passes X, Y, Z of one edgl_bimau_bwd call (without the parameter-partial reductions) */
enum { EDGL_KERNEL_SCORE_BWD_ROWS = 0, EDGL_KERNEL_BIMAU_BWD = 1, EDGL_KERNEL_BIMAU_FWD = 2, EDGL_KERNEL_BIMAU_BWD_ALL = 3 };
int edgl_profile_next(int kernel_id, void* ev_start, void* ev_stop);

/* ---- dropout RNG state ----------------------------------------------------------------------- */
/* rng_state[1] += 1 on the device (one launch per training step, graph-capturable). */
int edgl_rng_advance(uint64_t* rng_state, void* stream);

/* ---- K0: batch construction — MAUPostProcessor, dataloader.py:159-206 (choice: :34-36) -----------
 * tokens int64 [B,T] (T = seqslen + 1).  edgl_mask_random: per row `masklen` = M DISTINCT positions
 * drawn uniformly from [1, T) (ignore_head = 1, dataloader.py:183-186) by a counter-based generator
 * keyed on (rng_state, stream_id, row); masked_tokens = tokens with MASK = mask_id at those positions
 * (:188-193), masked_pos int64 [B,M], labels int64 [B,M] = the original tokens there (:194-201).
 * edgl_mask_last: evaluation batches — position T-1 := MASK (:166-179); labels are the input tokens. */
int edgl_mask_random(const int64_t* tokens, int B, int T, int M, int64_t mask_id, const uint64_t* rng_state,
                     uint32_t stream_id, int64_t* masked_tokens, int64_t* masked_pos, int64_t* labels,
                     void* stream);
int edgl_mask_last(const int64_t* tokens, int B, int T, int64_t mask_id, int64_t* masked_tokens, void* stream);

/* ---- K1: input encoding — EasyDGL.py:70-95, coding.py:60-64,76-79,137-149 ---------------------
 * x0[b,t,:]   = [ item_tab[id]*sqrt(C) + sincos(ts/time_scale) | pos_tab[t] | nmarks*mark_emb[1] ]
 *               followed by hidden dropout (EasyDGL.py:92);
 * spans[b,t]  = clip(ts'[t]-ts'[t-1],0,100), spans[b,0]=spans[b,1] with ts' = ts/time_scale (:73-74);
 * marks[b,t,:] = mark_table[id (MASK->0)] (:76-77); item row 0 and mark_emb row 0 act as zeros
 *               (coding.py:56-57).
 * item_tab is `dtype` (master f32 or its bf16 shadow); pos_tab / mark_emb are f32; mark_table holds
 * 0/1 bytes [>= num_items, E]; tscale f32 [C/2] = float32(10000^(2j/C)) (coding.py:134-135, computed on
 * the host exactly as the reference does). */
int edgl_encode_fwd(const int64_t* ids, const float* ts, const void* item_tab, const float* pos_tab,
                    const float* mark_emb, const uint8_t* mark_table, const float* tscale, int B, int T, int C,
                    int E, int I, int64_t mask_id, float time_scale, float drop_rate,
                    const uint64_t* rng_state, uint32_t stream_id, void* x0, float* spans, uint8_t* marks,
                    int dtype, void* stream);

/* Backward of K1 (SURVEY Appendix C "Embedding side"): d_item[id] += sqrt(C)*dX0[:, :C] for id != 0
 * (f32 atomics into d_item, which must be pre-initialised by the caller), d_pos[t] and d_mark_emb[1]
 * are written via per-block partials reduced deterministically (d_mark_emb [E,C]: only row 1 is
 * non-zero).  workspace: edgl_encode_bwd_workspace(B,T,C) floats. */
long edgl_encode_bwd_workspace(int B, int T, int C);
int edgl_encode_bwd(const int64_t* ids, const uint8_t* marks, const void* dx0, int B, int T, int C, int E,
                    int I, float drop_rate, const uint64_t* rng_state, uint32_t stream_id, float* d_item,
                    float* d_pos, float* d_mark_emb, float* workspace, int dtype, void* stream);
/* The same with two optional [B*T, C] terms (activation dtype; both or neither) added in f32 to the item section dX0[:, :C]
 * before the scatter: the residual branches of the first block, which the training engine would otherwise add with a
 * separate pass over dX0 (edgl_add_cols). */
int edgl_encode_bwd_add(const int64_t* ids, const uint8_t* marks, const void* dx0, const void* add1, const void* add2, int B,
                        int T, int C, int E, int I, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                        float* d_item, float* d_pos, float* d_mark_emb, float* workspace, int dtype, void* stream);
/* Channel-padded models (a model width whose head dim the attention kernels do not tile — the reference's default --num_units 50
 * --num_heads 1, main.py:35-37 — runs at the next supported head dim with zero-padded channels; channel c of a C-wide section is
 * real iff c % dh_pad < dh_true): edgl_encode_fwd / edgl_encode_bwd_add with coding.py:62-63's sqrt(C) taken of the TRUE width and
 * the time code of the padded channels zeroed (dh_pad = dh_true = 0 / c_true = 0: no padding). */
int edgl_encode_fwd_ct(const int64_t* ids, const float* ts, const void* item_tab, const float* pos_tab, const float* mark_emb,
                       const uint8_t* mark_table, const float* tscale, int B, int T, int C, int E, int I, int64_t mask_id,
                       float time_scale, float drop_rate, const uint64_t* rng_state, uint32_t stream_id, void* x0, float* spans,
                       uint8_t* marks, int dh_pad, int dh_true, int dtype, void* stream);
/* edgl_encode_fwd_ct with the batch preparation of a training step as the first workgroups of its launch (EasyDGL.py:70-95 encoder;
 * EasyDGL.py:177-185: which rows the loss weights — edgl_compact_scan_labels(labels [B * M]); EasyDGL.py:157-175 / temporal.py:317-333:
 * the slots of the TPP term — edgl_tpp_prep(masked_pos, labels, ts, mark_table) into tpp_desc, NULL: none).  Same results as the
 * three calls; what it writes is ordered in front of the stream's later kernels without a second stream and its join. */
int edgl_encode_fwd_prep(const int64_t* ids, const float* ts, const void* item_tab, const float* pos_tab, const float* mark_emb,
                         const uint8_t* mark_table, const float* tscale, int B, int T, int C, int E, int I, int64_t mask_id,
                         float time_scale, float drop_rate, const uint64_t* rng_state, uint32_t stream_id, void* x0, float* spans,
                         uint8_t* marks, int dh_pad, int dh_true, const int64_t* labels, int M, int32_t* perm, int32_t* inv,
                         int32_t* nvalid, int64_t* labels_c, const int64_t* masked_pos, void* tpp_desc, int dtype, void* stream);
int edgl_encode_bwd_add_ct(const int64_t* ids, const uint8_t* marks, const void* dx0, const void* add1, const void* add2, int B,
                           int T, int C, int E, int I, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                           float* d_item, float* d_pos, float* d_mark_emb, float* workspace, int c_true, int dtype, void* stream);
/* edgl_encode_bwd_add_ct that also applies the one-hot term of the tied table's scoring gradient which
 * edgl_score_flash_bwd_ex(defer_label_term = 1) left out (EasyDGL.py:177-185: dl = coef (softmax - onehot)):
 * d_item[label[r]] -= coef[r] rows[r], d_bias[label[r] - 1] -= coef[r] over the first min(lab_R, *lab_nvalid) compacted rows — as
 * extra blocks of the embedding scatter's launch (the same segmented sum over equal ids), instead of the launch and stream fork
 * of edgl_score_flash_label_term.  edgl_encode_bwd_label_fused(C, dtype) says whether this shape has the fused form. */
int edgl_encode_bwd_label_fused(int C, int dtype);
int edgl_encode_bwd_add_label(const int64_t* ids, const uint8_t* marks, const void* dx0, const void* add1, const void* add2, int B,
                              int T, int C, int E, int I, float drop_rate, const uint64_t* rng_state, uint32_t stream_id,
                              float* d_item, float* d_pos, float* d_mark_emb, float* workspace, int c_true, const void* lab_rows,
                              const int64_t* lab_ids, const float* lab_coef, const int32_t* lab_nvalid, int lab_R, float* d_bias,
                              int dtype, void* stream);

/* ---- K1b: CTSMA input encoding — CTSMA.py:48-58, coding.py:60-79 ----------------------------------
 * ids int64 [B,T] (tokens[:-1]), ts f32 [B,T+1] raw seconds.  x0 [B,T,2C] `dtype` =
 * dropout(concat(item_tab[ids] * sqrt(C) (row 0 reads as zeros), pos_tab[0..T))) (PositionCoding.__call__
 * concatenates); spans f32 [B,T] = ts[t+1]/time_scale - ts[t]/time_scale (no clipping); marks u8 [B,T,E] =
 * mark_table[ids].  bwd: d_item f32 [I,C] and d_pos f32 [T,C] are overwritten.
 * pos_tab == NULL (d_pos == NULL in the backward): x0 is [B,T,C], the item embedding alone (TGAT.py:49-56,
 * TiSASREC.py:52-65); spans and marks may be NULL together (then ts and mark_table are not read). */
int edgl_embed_pos_fwd(const int64_t* ids, const float* ts, const void* item_tab, const float* pos_tab,
                       const uint8_t* mark_table, int B, int T, int C, int E, float time_scale, float drop_rate,
                       const uint64_t* rng_state, uint32_t stream_id, void* x0, float* spans, uint8_t* marks,
                       int dtype, void* stream);
int edgl_embed_pos_bwd(const int64_t* ids, const void* dx0, int B, int T, int C, int I, float drop_rate,
                       const uint64_t* rng_state, uint32_t stream_id, float* d_item, float* d_pos, int dtype,
                       void* stream);
/* The same for a channel-padded model (TGAT / TiSASRec / CTSMA at a head dim the attention kernels do not tile, e.g. the
 * reference's default --num_units 50, main.py:35): c_true (0: = C) is the TRUE model width whose square root scales the item
 * embedding (coding.py:62-63); the padded channels of the tables are zero. */
int edgl_embed_pos_fwd_ct(const int64_t* ids, const float* ts, const void* item_tab, const float* pos_tab,
                          const uint8_t* mark_table, int B, int T, int C, int E, float time_scale, float drop_rate,
                          const uint64_t* rng_state, uint32_t stream_id, void* x0, float* spans, uint8_t* marks,
                          int c_true, int dtype, void* stream);
int edgl_embed_pos_bwd_ct(const int64_t* ids, const void* dx0, int B, int T, int C, int I, float drop_rate,
                          const uint64_t* rng_state, uint32_t stream_id, float* d_item, float* d_pos, int c_true,
                          int dtype, void* stream);

/* ---- K2/K4: dense layers — tf.layers.dense (temporal.py:409, EasyDGL.py:113,120,125,138) ------
 * C[M,N] = epilogue( sum_k A(m,k) * B(k,n) ).
 * a_kc != 0: A stored [M, lda] with k contiguous; else stored [K, lda] with m contiguous.
 * b_kc != 0: B stored [N, ldb] with k contiguous; else stored [K, ldb] with n contiguous.
 * A, B are `dtype`; C is `dtype` unless EDGL_EPI_OUT_F32.  bias f32[N].  aux is `dtype` [M,ldc].
 * splitk > 1 needs workspace >= splitk*M*N floats (partials reduced by a second launch). */
int edgl_gemm(const void* A, const void* Bm, void* Cm, int M, int N, int K, int lda, int ldb, int ldc,
              int a_kc, int b_kc, const float* bias, void* aux, int epi_flags, int splitk, float* workspace,
              int dtype, void* stream);

/* Weight / bias gradients of a dense layer: dW[Kf,N] (+)= X[R,Kf]^T . dY[R,N] (f32), and if dbias != NULL
 * dbias[N] (+)= colsum(dY).  workspace >= edgl_gemm_dw_workspace floats; per-split partials are reduced in a
 * fixed order. */
long edgl_gemm_dw_workspace(int R, int Kf, int N, int dtype);
int edgl_gemm_dw(const void* X, const void* dY, float* dW, float* dbias, int R, int Kf, int N, int ldx, int ldy,
                 int accumulate, float* workspace, int dtype, void* stream);
/* Grouped mode of edgl_gemm_dw (bf16): after edgl_gemm_dw_defer(1, stream) the calls are queued on the calling host thread;
 * edgl_gemm_dw_defer(0, stream) runs all of them as ONE launch (up to 8 products, row splits of similar length) followed by
 * their slab reductions.  Operands and workspaces of queued calls must stay untouched until then.  on < 0: drop the queue
 * without launching (error exit). */
int edgl_gemm_dw_defer(int on, void* stream);

/* out[n] (+)= sum_m X[m, n]  — bias gradients.  X `dtype` (or f32 if x_f32) [M, ld]; out f32[N]. */
int edgl_colsum(const void* X, int M, int N, int ld, float* out, int accumulate, float* workspace,
                int x_f32, int dtype, void* stream);

/* ---- K3: fused BiMAU attention — temporal.py:404-452 + MAU.intensity temporal.py:281-315 -------
 * qkvt [B,T,4C] = dense(x) already computed (K2), split order Q,K,V,T (temporal.py:410).  Per
 * (b, head): S=QK^T/sqrt(dh), key mask (ids[b,k]==0 -> -2^32+1), softmax, H=P.T_, intensity MLP ->
 * lam[b',q,:E], G=lam.marks^T with diag:=1, attention dropout, O=(G*P).V;
 * out[b,q,head*dh+:] = O + resid[b,q,head*dh+:] (:447); resid has row stride ld_res (first C channels
 * of the block input).  The intensity weights W1 [dh+1, dh*E], b1 [dh*E], w [E,dh], scaling [E] (f32)
 * are first packed by edgl_bimau_pack into `pack` (edgl_bimau_pack_bytes bytes, reusable by fwd and
 * bwd until the weights change).  lam_out f32 [H*B,T,E].  `saved` (edgl_bimau_saved_bytes bytes, or
 * NULL for inference) receives what the backward re-uses: the H rows fed to the intensity MLP and its
 * pre-softplus output z.  `flags`: 0 = BiMAU as above; EDGL_MAU_CAUSAL adds the future-blinding mask of
 * MAU.__call__(causality=True) (temporal.py:370-375: keys k > q scored -2^32+1 like padded keys, no gradient
 * through them); EDGL_MAU_NO_DIAG keeps the modulation on the diagonal (MAU, temporal.py:383; BiMAU overwrites
 * it with 1, :438-439); EDGL_MAU_DIAG_ZERO writes 0 there instead of 1 — for a unit with more than 16 mark types, run as
 * groups of <= 16 marks whose outputs are summed (G is linear in the marks: group 0 carries the diagonal 1 and the
 * residual, the later groups 0 and a zero residual).  The caller may fill the Q and K|V|T_ column blocks of qkvt from different inputs
 * (MAU: Q = dense(LN(x)), K,V,T_ = dense(x), temporal.py:352-355).  Supported: dh in {16,32}, E<=16, T<=128. */
#define EDGL_MAU_CAUSAL 1
#define EDGL_MAU_NO_DIAG 2
#define EDGL_MAU_DIAG_ZERO 4
/* host-side hint, not a semantic flag: launch the kernels that walk every key tile.  The bf16 / head dim 16 / 16 marks family
 * otherwise leaves out the all-padding key tiles in front of a sequence's first real key (exact); a caller whose batches cannot
 * have any — training batches of the reference's masker carry MASK tokens on padded positions, dataloader.py:187-191 — saves the
 * skip variant's per-tile scalar branch.  Results are identical either way. */
#define EDGL_MAU_NO_SKIP 8
long edgl_bimau_pack_bytes(int C, int H, int E, int dtype);
long edgl_bimau_saved_bytes(int B, int T, int C, int H, int dtype);
/* largest mark count (<= 16) one launch takes at this head dim / dtype (LDS of the intensity backward); -1: bad arguments */
int edgl_bimau_mark_group(int C, int H, int dtype);
int edgl_bimau_pack(const float* W1, const float* b1, const float* w, const float* scaling, int C, int H, int E,
                    void* pack, int dtype, void* stream);
int edgl_bimau_fwd(const void* qkvt, const void* resid, int ld_res, const int64_t* ids, const float* spans,
                   const uint8_t* marks, const void* pack, int B, int T, int C, int H, int E, float drop_rate,
                   const uint64_t* rng_state, uint32_t stream_id, void* out, float* lam_out, void* saved,
                   int flags, int dtype, void* stream);
/* edgl_bimau_fwd that also fills `zero_rows` (f32 [H*B,T,E], may be NULL) with zeros — the training engine's d lambda buffer, of
 * which edgl_tpp_fwd_bwd_rows then writes the masked positions only (the forward is VALU bound: the extra stores are free). */
int edgl_bimau_fwd_zr(const void* qkvt, const void* resid, int ld_res, const int64_t* ids, const float* spans,
                      const uint8_t* marks, const void* pack, int B, int T, int C, int H, int E, float drop_rate,
                      const uint64_t* rng_state, uint32_t stream_id, void* out, float* lam_out, void* saved,
                      float* zero_rows, int flags, int dtype, void* stream);

/* Attention dropout as stored keep bits (temporal.py:442; the three kernels of a training step — forward and the two backward
 * sweeps — take the same decisions): edgl_bimau_dropbits evaluates the counter hash of (rng_state, stream_id, element (b', q, k))
 * ONCE and stores the decisions in the kernels' register layout — edgl_bimau_dropbits_bytes(B, T, H) bytes (0: no stored-bits form
 * at this T, the kernels hash) —, edgl_bimau_fwd_db / edgl_bimau_bwd_db are edgl_bimau_fwd_zr / edgl_bimau_bwd reading them
 * (dropbits NULL = hash).  Identical masks either way: a kernel without a stored-bits form (head dims other than 16, E != 16, MAU
 * flags, f32) ignores the argument and hashes.  qk_scale: the score scale of temporal.py:422 (0 = 1 / sqrt(dh)); a model whose
 * head dim d is none of {16, 32, 64, 128} — the reference's default --num_units 50 --num_heads 1 (main.py:35-37) — runs with
 * zero-padded channels at the next supported head dim and qk_scale = 1 / sqrt(d) (exact: padded Q / K / V / T_ columns are 0). */
long edgl_bimau_dropbits_bytes(int B, int T, int H);
int edgl_bimau_dropbits(int B, int T, int H, float drop_rate, const uint64_t* rng_state, uint32_t stream_id, uint32_t* bits,
                        void* stream);
int edgl_bimau_fwd_db(const void* qkvt, const void* resid, int ld_res, const int64_t* ids, const float* spans,
                      const uint8_t* marks, const void* pack, int B, int T, int C, int H, int E, float drop_rate,
                      const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale, void* out,
                      float* lam_out, void* saved, float* zero_rows, int flags, int dtype, void* stream);

/* Launch order of the attention kernels' (sample, head) jobs (temporal.py:404-452 computes every head of every sample; the order
 * only decides WHEN a job runs).  The bf16 / head dim 16 / 16 marks kernels leave out the key tiles that hold nothing but the
 * left padding of a sequence (data/linkpred.py:142-157; temporal.py:425-429: their probabilities are exactly 0), so a job's time
 * falls with its padding; edgl_bimau_job_order lists the samples by falling key-tile count (first B entries of `order`, ties by index) and
 * edgl_bimau_fwd_ord / edgl_bimau_bwd_ord launch the long jobs first.  order == NULL: index order (= edgl_bimau_fwd_db /
 * edgl_bimau_bwd_db / edgl_bimau_bwd_tpp).  edgl_bimau_bwd_ord: tpp_desc == NULL -> d_lam_ext as in edgl_bimau_bwd_db;
 * tpp_desc != NULL -> the fused TPP form of edgl_bimau_bwd_tpp (d_lam_ext must be NULL).  Results do not depend on the order. */
int edgl_bimau_job_order(const int64_t* ids, int B, int T, int32_t* order, void* stream);   /* order: int32 [2 * B] (order | scratch); B <= 16384 */
int edgl_bimau_fwd_ord(const void* qkvt, const void* resid, int ld_res, const int64_t* ids, const float* spans,
                       const uint8_t* marks, const void* pack, int B, int T, int C, int H, int E, float drop_rate,
                       const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale, void* out,
                       float* lam_out, void* saved, float* zero_rows, const int32_t* order, int flags, int dtype, void* stream);
int edgl_bimau_bwd_ord(const void* qkvt, const int64_t* ids, const float* spans, const uint8_t* marks, const void* pack,
                       const void* d_out, const float* d_lam_ext, const void* tpp_desc, int M, const float* tpp_sums, float coef,
                       float* tpp_part, const float* lam, const void* saved, int B, int T, int C, int H, int E, float drop_rate,
                       const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale, void* d_qkvt,
                       float* dW1, float* db1, float* dw, float* dscaling, void* workspace, const int32_t* order, int flags,
                       int dtype, void* stream);

/* Backward.  d_out [B,T,C] `dtype`; d_lam_ext f32 [H*B,T,E] or NULL (gradient
 * from the TPP regulariser); lam / saved: the forward's lam_out and `saved` buffer.  Writes d_qkvt [B,T,4C] `dtype` and the f32 weight gradients dW1
 * [dh+1,dh*E], db1 [dh*E], dw [E,dh], dscaling [E] (overwritten; per-workgroup partials in
 * `workspace` — edgl_bimau_bwd_workspace BYTES — are reduced in a fixed order).  The residual
 * gradient (d_out itself) is NOT added here — the caller routes it. */
long edgl_bimau_bwd_workspace(int B, int T, int C, int H, int E, int dtype);
int edgl_bimau_bwd(const void* qkvt, const int64_t* ids, const float* spans, const uint8_t* marks,
                   const void* pack, const void* d_out, const float* d_lam_ext, const float* lam,
                   const void* saved, int B, int T, int C, int H, int E, float drop_rate,
                   const uint64_t* rng_state, uint32_t stream_id, void* d_qkvt, float* dW1, float* db1,
                   float* dw, float* dscaling, void* workspace, int flags, int dtype, void* stream);
int edgl_bimau_bwd_db(const void* qkvt, const int64_t* ids, const float* spans, const uint8_t* marks,
                      const void* pack, const void* d_out, const float* d_lam_ext, const float* lam,
                      const void* saved, int B, int T, int C, int H, int E, float drop_rate,
                      const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale, void* d_qkvt,
                      float* dW1, float* db1, float* dw, float* dscaling, void* workspace, int flags, int dtype, void* stream);

/* ---- K4-LN: y = layernorm_joint(dropout(x) + resid) — Base.py:12-67 (moments over (T,C) per
 * sample, eps 1e-12), EasyDGL.py:114-116,126-128,139.  resid may be NULL (ld_res ignored).
 * stats f32 [B,2] = (mean, rstd).  If gather_pos != NULL (int64 [B,Mg]) only rows
 * y_g[b*Mg+j,:] = y[b, gather_pos[b,j], :] are written (EasyDGL.py:142-146 batch_gather). */
int edgl_add_layernorm_fwd(const void* x, const void* resid, int ld_res, const float* gamma,
                           const float* beta, int B, int T, int C, float drop_rate,
                           const uint64_t* rng_state, uint32_t stream_id, const int64_t* gather_pos,
                           int Mg, void* y, float* stats, int dtype, void* stream);

/* Backward: given dy (dense [B,T,C], or gathered [B*Mg,C] when gather_pos != NULL), recomputes
 * xhat from (x, resid, stats) and writes dx (gradient w.r.t. dropout(x)+resid input sum, i.e. the
 * caller uses it for the residual; d_x_pre = dx * dropmask is written to dx_drop) and per-sample
 * partials of dgamma/dbeta into workspace [B,2,C] reduced into dgamma/dbeta (overwritten).
 * Gathered rows naming the same position add up.  dy_rowmap (int32 [B*Mg], may be NULL) redirects
 * gathered row r to row dy_rowmap[r] of dy (-1: the row carries no gradient) — the `inv` map of
 * edgl_compact_rows. */
int edgl_add_layernorm_bwd(const void* x, const void* resid, int ld_res, const float* gamma,
                           const float* stats, const void* dy, int B, int T, int C, float drop_rate,
                           const uint64_t* rng_state, uint32_t stream_id, const int64_t* gather_pos,
                           int Mg, const int32_t* dy_rowmap, void* dsum, void* dx_drop, float* dgamma,
                           float* dbeta, float* workspace, int dtype, void* stream);
/* The same with a fused activation backward: x is gelu(act_pre) (the head transform, EasyDGL.py:136-139); the emitted
 * gradients (dsum, dx_drop) are multiplied by gelu'(act_pre) — the gradient w.r.t. the dense layer's pre-activation. */
int edgl_add_layernorm_bwd_act(const void* x, const void* resid, int ld_res, const float* gamma,
                               const float* stats, const void* dy, int B, int T, int C, float drop_rate,
                               const uint64_t* rng_state, uint32_t stream_id, const int64_t* gather_pos,
                               int Mg, const int32_t* dy_rowmap, const void* act_pre, void* dsum, void* dx_drop,
                               float* dgamma, float* dbeta, float* workspace, int dtype, void* stream);
/* The joint LayerNorm of a channel-padded model (see edgl_encode_fwd_ct): moments over the REAL channels only (divisor T * true
 * width; the padded channels hold exact zeros and stay out of the centred second moment), the input gradient of a padded channel
 * is zero — so nothing ever flows into the padded rows / columns of the dense kernels (Base.py:12-67 on the true width). */
int edgl_add_layernorm_fwd_ct(const void* x, const void* resid, int ld_res, const float* gamma, const float* beta, int B, int T,
                              int C, float drop_rate, const uint64_t* rng_state, uint32_t stream_id, const int64_t* gather_pos,
                              int Mg, void* y, float* stats, int dh_pad, int dh_true, int dtype, void* stream);
int edgl_add_layernorm_bwd_act_ct(const void* x, const void* resid, int ld_res, const float* gamma, const float* stats,
                                  const void* dy, int B, int T, int C, float drop_rate, const uint64_t* rng_state,
                                  uint32_t stream_id, const int64_t* gather_pos, int Mg, const int32_t* dy_rowmap,
                                  const void* act_pre, void* dsum, void* dx_drop, float* dgamma, float* dbeta, float* workspace,
                                  int dh_pad, int dh_true, int dtype, void* stream);

/* ---- K5: tied-embedding scoring + cross-entropy — EasyDGL.py:149-155,177-185, Base.py:106-110 --
 * rows [R,C] `dtype`; table [I,C] `dtype` (row 0 acts as zeros, column 0 logit == -1000); out_bias
 * f32 [I-1]; labels int64 [R] (may be NULL for eval).  The item range [i0, i1) lets a rank score its
 * shard only.  The [R, I] logits are never materialised unless `logits` (f32 [R, i1-i0]) is given.
 * fwd: online (max, sumexp) per row and item chunk -> row_lse f32 [R] (log-sum-exp over [i0,i1)),
 * label_logit f32 [R] (written only for rows whose label lies in [i0,i1); pre-zero when sharded).
 * workspace >= 2*R*edgl_score_chunks(R, i1-i0) floats.  C: power of two, 32..256 (bf16) / 32..128 (f32);
 * i0 must be a multiple of 8. */
int edgl_score_chunks(int R, int n_items);
/* Rows with label 0 have weight 0 in the loss (EasyDGL.py:180) and therefore no effect on the loss
 * or on any gradient.  edgl_compact_rows orders the weighted rows first: perm int32 [R] (original
 * row of compact row j, -1 past the end), inv int32 [R] (compact index of row r, or -1), nvalid
 * int32 [1] (device), rows_c [R,C] / labels_c [R] the gathered copies (zero past nvalid).  The
 * scoring entry points take `nvalid` (device pointer or NULL): rows >= *nvalid are skipped (their
 * row_lse is 0, their d_rows undefined).  edgl_scatter_rows undoes the compaction:
 * rows[r] = inv[r] >= 0 ? rows_c[inv[r]] : 0. */
int edgl_compact_rows(const void* rows, const int64_t* labels, int R, int C, int32_t* perm, int32_t* inv,
                      int32_t* nvalid, void* rows_c, int64_t* labels_c, int dtype, void* stream);
/* The two halves of edgl_compact_rows: the scan needs the labels only (it may run before the rows exist), the gather the rows. */
int edgl_compact_scan(const int64_t* labels, int R, int32_t* perm, int32_t* inv, int32_t* nvalid, void* stream);
/* ... the scan that also writes the compacted labels (labels_c int64 [R]: weighted rows first, 0 behind them) */
int edgl_compact_scan_labels(const int64_t* labels, int R, int32_t* perm, int32_t* inv, int32_t* nvalid, int64_t* labels_c,
                             void* stream);
int edgl_compact_gather(const void* rows, const int64_t* labels, const int32_t* perm, int R, int C, void* rows_c,
                        int64_t* labels_c, int dtype, void* stream);
int edgl_scatter_rows(const void* rows_c, const int32_t* inv, int R, int C, void* rows, int dtype,
                      void* stream);
/* (row_lse may be NULL when logits is given: the evaluation path wants the logits tile only — Base.py:150-163) */
int edgl_score_lse_fwd(const void* rows, const void* table, const float* out_bias, const int64_t* labels,
                       int R, int C, int I, int i0, int i1, const int32_t* nvalid, float* row_lse,
                       float* label_logit, float* logits, float* workspace, int dtype, void* stream);
/* loss = sum_r w_r * (-log(p_y + 1e-5)) / (sum w + 1e-5), w_r = [label != 0]; also writes the
 * per-row coefficient coef[r] = (w_r/W) * p_y/(p_y+1e-5) used by the backward.  loss_out f32[1]. */
int edgl_ce_loss_fwd(const float* row_lse, const float* label_logit, const int64_t* labels, int R,
                     float* loss_out, float* coef, void* stream);
/* The same with loss_out[0] = cross-entropy + add_in[0] + add_in2[0] (device scalars holding the regularisation terms of
 * the step — two, so that terms produced on different streams need no common accumulator; either may be NULL): saves the
 * training engine a separate one-element add launch. */
int edgl_ce_loss_fwd_add(const float* row_lse, const float* label_logit, const int64_t* labels, int R, float* loss_out,
                         float* coef, const float* add_in, const float* add_in2, void* stream);
/* ... with the denominator of the GLOBAL batch (`wtotal`, device int32 or NULL; see edgl_score_flash_fwd_coef_w): loss_out[0] is
 * then this rank's share of the global cross-entropy (+ the terms added in). */
int edgl_ce_loss_fwd_add_w(const float* row_lse, const float* label_logit, const int64_t* labels, int R, float* loss_out,
                           float* coef, const float* add_in, const float* add_in2, const int32_t* wtotal, void* stream);
/* backward of the CE: dl[r,j] = g * coef[r] * (p[r,j] - [j==label_r]) (g = d loss, device scalar or
 * NULL for 1), never materialised:  d_rows[R,C] (`dtype`) = dl . table ;  d_table[I,C] (f32,
 * overwritten for rows [i0,i1), row 0 := 0) = dl^T . rows ;  d_bias[I-1] f32 = colsum(dl)[1:].
 * workspace >= edgl_score_bwd_workspace(R, C, I, i1-i0, dtype) floats (holds the per-chunk partial slabs, reduced in a
 * fixed order, and — f32 only — the transposed operand images rowsT/tableT; the bf16 kernels read the transposed
 * operand out of their row-major LDS tiles). */
long edgl_score_bwd_workspace(int R, int C, int I, int n_items, int dtype);
int edgl_score_ce_bwd(const void* rows, const void* table, const float* out_bias, const int64_t* labels,
                      const float* row_lse, const float* coef, const float* gscale, int R, int C, int I,
                      int i0, int i1, const int32_t* nvalid, void* d_rows, float* d_table, float* d_bias,
                      float* workspace, int dtype, void* stream);

/* ---- K6: evaluation — Base.py:150-207 ----------------------------------------------------------
 * logits f32 [R, n] for the item range starting at i0; seen [R,T] int64 item ids to mask with -inf
 * (Base.py:156-163; NULL = no masking); top-K by (value desc, index asc) (tf.nn.top_k tie rule).
 * out_val f32 [R,K], out_idx int32 [R,K] (GLOBAL item ids = i0 + local). K <= 128. */
int edgl_mask_topk(float* logits, int R, int n, int i0, const int64_t* seen, int T, int K, float* out_val,
                   int32_t* out_idx, void* stream);
/* Fused evaluation scoring (Base.py:150-181 + EasyDGL.py:149-151): logits = rows . table[i0:i1]^T + [-1000, out_bias] -> -inf at the
 * row's seen ids (seen [R,T] int64, may be NULL with T = 0) -> top-K by (value desc, global index asc), WITHOUT the [R, i1-i0] logits
 * tile in HBM: the logits are computed twice on the matrix pipe — group maxima give every row a lower bound of its K-th best unseen
 * logit, the second sweep appends the few hundred elements above it to the row's candidate list, one more launch ranks
 * them; rows whose list overflows (heavy ties) are redone exactly from the row's logits in a scratch row.  bf16; C in {64, 128, 256};
 * K <= 128; 4096 <= i1 - i0 <= 262144 per call (larger catalogues: item chunks + edgl_topk_merge).  edgl_score_topk_fused_supported
 * = 1 when the shape is taken — otherwise the call returns EDGL_ERR_SHAPE (no silent fallback): run edgl_score_lse_fwd +
 * edgl_mask_topk.  workspace: edgl_score_topk_fused_workspace BYTES.  out_val f32 [R,K], out_idx int32 [R,K] (global ids). */
int edgl_score_topk_fused_supported(int R, int C, int n_items, int T, int K, int dtype);
long edgl_score_topk_fused_workspace(int R, int C, int n_items, int T, int K);
int edgl_score_topk_fused(const void* rows, const void* table, const float* out_bias, const int64_t* seen, int T, int R, int C,
                          int I, int i0, int i1, int K, float* out_val, int32_t* out_idx, void* workspace, int dtype, void* stream);
/* ---- K7: merge of per-shard candidates (after an RCCL all-gather): cand_val/idx [S, R, K] ->
 * global top-K per row by (value desc, global index asc); entries with index < 0 are ignored; S*K <= 1024.  (Candidates above the
 * K-th largest thread maximum are ranked by counting; inputs with more than 256 of them — tie blocks — are sorted.) */
int edgl_topk_merge(const float* cand_val, const int32_t* cand_idx, int S, int R, int K, float* out_val,
                    int32_t* out_idx, void* stream);
/* HR@k / NDCG@k sums for k in {10,50,100} (Base.py:181-201): metrics f32 [6] += per-batch sums in
 * the order H10,H50,H100,N10,N50,N100; label int64 [R]. */
int edgl_rank_metrics(const int32_t* topk_idx, int R, int K, const int64_t* label, float* metrics,
                      void* stream);

/* ---- K5, "flash" form (used by the static training engine): the forward scoring pass already accumulates the row
 * gradients, so the [R, I] logits are computed twice per step instead of three times.
 * edgl_score_flash_fwd: one pass over the item rows [i0, i1) gives row_lse (as edgl_score_lse_fwd), label_logit, and keeps
 * sum_z exp(logit_z - max) table[z] per weighted row (flash-style running maxima) in `workspace`.
 * edgl_score_flash_bwd: d_rows = gscale * coef * (that sum / row sum - table[label]) — i.e. dl . table with
 * dl = coef (softmax - onehot), SURVEY Appendix C — then d_table / d_bias exactly as edgl_score_ce_bwd.  `workspace`:
 * edgl_score_flash_workspace floats, untouched between the two calls.  Other arguments as edgl_score_lse_fwd /
 * edgl_score_ce_bwd (EasyDGL.py:149-155,177-185).
 * Item ranges (the item table row-sharded over ranks; i0 a multiple of 8): edgl_score_flash_fwd over [i0, i1) gives the
 * log-sum-exp of the RANGE's logits (the pad item's -1000 only where the range holds item 0) and label_logit where i0 <= label < i1;
 * edgl_score_flash_bwd over the same range takes the GLOBAL row_lse / coef (merged over the ranges by the caller) and returns the
 * range's share of d_rows — its label row subtracted only by the range that owns the label — and writes rows [i0, i1) of d_table /
 * entries [max(i0,1) - 1, i1 - 1) of d_bias, nothing else: the shares add up to the unsharded result (tests/test_gpu_score_strip.py). */
long edgl_score_flash_workspace(int R, int C, int I, int n_items, int dtype);
int edgl_score_flash_fwd(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R, int C,
                         int I, int i0, int i1, const int32_t* nvalid, float* row_lse, float* label_logit,
                         float* workspace, int dtype, void* stream);
/* f32: the transposed operand image of the item table depends on the weights only: edgl_score_prepare_table writes it into
 * the flash workspace ahead of the forward (same R, C, I, [i0, i1), workspace, dtype), and edgl_score_flash_fwd_pre with
 * table_ready != 0 skips it (table_ready == 0: identical to edgl_score_flash_fwd).  bf16 needs no such image:
 * edgl_score_prepare_table returns at once and table_ready is ignored. */
int edgl_score_prepare_table(const void* table, int R, int C, int I, int i0, int i1, float* workspace, int dtype, void* stream);
int edgl_score_flash_fwd_pre(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R, int C,
                             int I, int i0, int i1, const int32_t* nvalid, float* row_lse, float* label_logit,
                             float* workspace, int table_ready, int dtype, void* stream);
/* edgl_score_flash_fwd over the whole table for COMPACTED rows (edgl_compact_*: the *nvalid weighted rows first, labels 0 behind)
 * that also writes the loss coefficients coef[r] = (1 / (n + 1e-5)) * p_y / (p_y + 1e-5), n = *nvalid — exactly what
 * edgl_ce_loss_fwd computes after its reduction over the rows (EasyDGL.py:177-185); the backward then does not wait for the loss
 * kernel, which may be given coef = NULL. */
int edgl_score_flash_fwd_coef(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R, int C,
                              int I, const int32_t* nvalid, float* row_lse, float* label_logit, float* coef, float* workspace,
                              int dtype, void* stream);
/* Data-parallel form: `wtotal` (device int32, may be NULL = the local count) is the number of weighted rows of the
 * GLOBAL batch — the denominator of EasyDGL.py:183-185 summed over the ranks — so that the ranks' gradients add up to the gradient
 * of the global-batch loss. */
int edgl_score_flash_fwd_coef_w(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R, int C,
                                int I, const int32_t* nvalid, const int32_t* wtotal, float* row_lse, float* label_logit, float* coef,
                                float* workspace, int dtype, void* stream);
/* ... that also finishes the rows: d_rows [R, C] `dtype` (= gscale * d loss / d rows; gscale device scalar or NULL = 1) is written by
 * the launch that forms lse / label logits / coefficients (one kernel instead of two between the two product passes);
 * edgl_score_flash_bwd is then called with d_rows = NULL and computes d_table / d_bias only. */
int edgl_score_flash_fwd_rows_w(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R, int C,
                                int I, const int32_t* nvalid, const int32_t* wtotal, const float* gscale, float* row_lse,
                                float* label_logit, float* coef, void* d_rows, float* workspace, int dtype, void* stream);
/* edgl_score_flash_fwd_rows_w that also leaves the loss numerator — sum over the weighted rows of -log(p_label + 1e-5),
 * EasyDGL.py:181-185 — as edgl_score_ce_nparts(R, C) per-workgroup sums in ce_part, the weighted-row count behind them (nparts + 1
 * floats; 0 parts: this width has no one-launch row finish; ce_part NULL: none).  edgl_ce_loss_parts forms the loss from them: sum / (weighted rows + 1e-5) + add_in + add_in2; it
 * reads nothing else of the batch, so it may run any time before the next forward rewrites ce_part. */
int edgl_score_ce_nparts(int R, int C);
int edgl_score_flash_fwd_rows_wp(const void* rows, const void* table, const float* out_bias, const int64_t* labels, int R, int C,
                                 int I, const int32_t* nvalid, const int32_t* wtotal, const float* gscale, float* row_lse,
                                 float* label_logit, float* coef, void* d_rows, float* ce_part, float* workspace, int dtype,
                                 void* stream);
int edgl_ce_loss_parts(const float* ce_part, int nparts, float* loss_out, const float* add_in, const float* add_in2,
                       const int32_t* wtotal, void* stream);      /* ce_part: nparts + 1 floats (the weighted-row count rides last) */
int edgl_score_flash_bwd(const void* rows, const void* table, const float* out_bias, const int64_t* labels,
                         const float* row_lse, const float* coef, const float* gscale, int R, int C, int I, int i0,
                         int i1, const int32_t* nvalid, void* d_rows, float* d_table, float* d_bias, float* workspace,
                         int dtype, void* stream);
/* The one-hot part of dl = coef (p - onehot(label)) as a call of its own: d_table[label[r]] -= gscale coef[r] rows[r], d_bias[label[r]-1]
 * -= gscale coef[r] over the weighted rows (f32 atomics: commutes with every other accumulation into the two arrays).
 * edgl_score_flash_bwd_ex(defer_label_term = 1) leaves exactly this out where its product pass does not contain it (bf16, C = 128 / 256 / 512: the strip kernels);
 * edgl_score_flash_label_term then applies it — and is a no-op for every other configuration.  A training loop may run it off the
 * critical path (any time after the table-side reduction of edgl_score_flash_bwd_ex wrote d_table / d_bias). */
int edgl_score_flash_bwd_ex(const void* rows, const void* table, const float* out_bias, const int64_t* labels,
                            const float* row_lse, const float* coef, const float* gscale, int R, int C, int I, int i0,
                            int i1, const int32_t* nvalid, void* d_rows, float* d_table, float* d_bias, float* workspace,
                            int defer_label_term, int dtype, void* stream);
int edgl_score_flash_label_term(const void* rows, const int64_t* labels, const float* coef, const float* gscale, int R, int C, int I,
                                int i0, int i1, const int32_t* nvalid, float* d_table, float* d_bias, int dtype, void* stream);

/* ---- deferred partial reductions ------------------------------------------------------------------
 * The weight-gradient entry points (edgl_gemm_dw, edgl_add_layernorm_bwd, edgl_encode_bwd,
 * edgl_bimau_bwd, edgl_colsum) finish with a small fixed-order reduction of per-workgroup partials.
 * After edgl_reduce_defer(1, stream) those reductions are queued on the calling host thread instead
 * of launched, and edgl_reduce_flush(stream) (or edgl_reduce_defer(0, stream)) runs all of them in
 * one launch.  While deferred, every call must be given its OWN workspace, kept until the flush,
 * and the gradient outputs are not valid before it.  Accumulating calls flush and run immediately.
 * edgl_reduce_defer(-1, stream) is the error exit: it drops the queued jobs WITHOUT launching them and
 * leaves deferred mode (a caller whose launch sequence failed half-way must not leave the thread deferred). */
int edgl_reduce_defer(int on, void* stream);
int edgl_reduce_flush(void* stream);

/* ---- K8: TPP likelihood regulariser — temporal.py:317-333 + EasyDGL.py:157-175 -----------------
 * lam f32 [H*B,T,E]; masked_pos int64 [B,M]; labels int64 [B,M]; ts_raw f32 [B,T] (raw seconds);
 * mark_table uint8 [NI,E].  reg_out f32[1] (+)= coef * biased_mle with coef = ct_reg/H;
 * sums f32[edgl_tpp_workspace()] scratch (sums[0..2] = event-ll, non-event, #marks are reused by
 * the backward).  bwd zero-fills and writes d_lam f32 [H*B,T,E] (zero except at masked positions), scaled by
 * the device scalar gscale (NULL = 1).
 * masked_pos == NULL selects the all-position form of CTSMA.train (CTSMA.py:95-108): M == T, labels [B,T] (next
 * items), ts_raw f32 [B,T+1] and the interval of position t is ts[t+1]-ts[t] (raw, unclipped). */
int edgl_tpp_workspace(void);
int edgl_tpp_fwd(const float* lam, const int64_t* masked_pos, const int64_t* labels, const float* ts_raw,
                 const uint8_t* mark_table, int B, int T, int H, int E, int M, float coef, float* sums,
                 float* reg_out, int accumulate, void* stream);
int edgl_tpp_bwd(const float* lam, const int64_t* masked_pos, const int64_t* labels, const float* ts_raw,
                 const uint8_t* mark_table, int B, int T, int H, int E, int M, float coef,
                 const float* sums, const float* gscale, float* d_lam, void* stream);
/* edgl_tpp_fwd + edgl_tpp_bwd as one call (three small launches instead of four + a memset): reg_out (+)= regulariser, sums[0..2] as edgl_tpp_fwd, and (d_lam != NULL) the
 * FULL gradient d_lam[H*B*T, E] — zero rows where the position is not masked (no memset needed), contributions of repeated
 * masked positions summed.  sums: edgl_tpp_workspace() floats; sums[4] must be zero before the first call (integer normaliser
 * accumulator; left zero). */
int edgl_tpp_fwd_bwd(const float* lam, const int64_t* masked_pos, const int64_t* labels, const float* ts_raw,
                     const uint8_t* mark_table, int B, int T, int H, int E, int M, float coef, float* sums,
                     float* reg_out, int accumulate, float* d_lam, void* stream);
/* The normaliser alone (labels only; integer sum into sums[4]) and edgl_tpp_fwd_bwd with it optional (with_norm == 0: edgl_tpp_norm
 * already ran for this batch on the same `sums`). */
int edgl_tpp_norm(const int64_t* labels, const uint8_t* mark_table, int B, int M, int E, float* sums, void* stream);
int edgl_tpp_fwd_bwd_ex(const float* lam, const int64_t* masked_pos, const int64_t* labels, const float* ts_raw,
                        const uint8_t* mark_table, int B, int T, int H, int E, int M, float coef, float* sums,
                        float* reg_out, int accumulate, float* d_lam, int with_norm, void* stream);
/* edgl_tpp_fwd_bwd_ex(with_norm = 0) for a d_lam array that ALREADY HOLDS ZEROS (edgl_bimau_fwd_zr): one thread per masked slot,
 * only the rows of masked positions are written (repeated positions: once, summed).  sums: max(edgl_tpp_workspace(),
 * edgl_tpp_rows_workspace(B, H, M)) floats; edgl_tpp_norm must have run on the same `sums`. */
long edgl_tpp_rows_workspace(int B, int H, int M);
int edgl_tpp_fwd_bwd_rows(const float* lam, const int64_t* masked_pos, const int64_t* labels, const float* ts_raw,
                          const uint8_t* mark_table, int B, int T, int H, int E, int M, float coef, float* sums,
                          float* reg_out, int accumulate, float* d_lam, void* stream);   /* E <= 256 (E > 16: plain loops over the marks) */

/* Fused TPP form (bf16, head dim 16, E = 16, T <= 128, M <= 256): the regulariser of MAU.biased_likelihood (temporal.py:317-333)
 * evaluated at the masked positions (EasyDGL.py:157-175) INSIDE sweep 1 of the attention backward, which holds lambda in registers —
 * no [H*B,T,E] d lambda array and no TPP launch between the attention forward and the block tail (the forward runs as
 * edgl_bimau_fwd_db with zero_rows = NULL).
 *   edgl_tpp_prep         slot data of a batch (labels / positions / raw timestamps only, runs ahead of the forward): per position
 *                         the next-mark bytes of its first effective slot (a slot whose label's mark row is not empty:
 *                         temporal.py:321) and the raw span (EasyDGL.py:161-162), further effective slots of a taken position
 *                         in a per-sample overflow list, and the marks of all labels of a sample (their total is the regulariser's
 *                         normaliser, temporal.py:330); desc = edgl_tpp_prep_bytes(B, T, M) bytes, 16-byte aligned.
 *   edgl_bimau_bwd_tpp    edgl_bimau_bwd_db with the regulariser's d lambda (coef = ct_reg / H; count: tpp_sums[4] as edgl_tpp_norm or a
 *                         data-parallel all-reduce left it, tpp_sums == NULL: the total of the slot data's per-sample counts)
 *                         computed by sweep 1 from the slot data, and the two loss sums of the wave's (b, head) in tpp_part f32
 *                         [B*H + 1, 2] — the last pair's first word = the count it used (int32)
 *   edgl_tpp_finish_parts their reduction: reg_out (+)= coef * (-(sum log ev - sum non-event) / (count * H)); sums as edgl_tpp_fwd_bwd
 *                         (count: tpp_desc != NULL — from the slot data (B, T, M), also stored into sums[4]; NULL — sums[4] as given)
 *   edgl_tpp_finish_parts_n  the same with the count edgl_bimau_bwd_tpp left behind the sums (part [nparts + 1, 2]): it reads nothing
 *                         of the batch, a training loop may launch it any time before the next edgl_bimau_bwd_tpp on that array */
/* data parallel: counts[0] = labels != 0, counts[1] = marks of all labels — the two normalisers a step all-reduces (B * M <= 65536) */
int edgl_dp_counts(const int64_t* labels, const uint8_t* mark_table, int B, int M, int E, int32_t* counts, void* stream);
long edgl_tpp_prep_bytes(int B, int T, int M);
int edgl_tpp_prep(const int64_t* masked_pos, const int64_t* labels, const float* ts_raw, const uint8_t* mark_table, int B, int T,
                  int E, int M, void* desc, void* stream);
int edgl_bimau_bwd_tpp(const void* qkvt, const int64_t* ids, const float* spans, const uint8_t* marks, const void* pack,
                       const void* d_out, const void* tpp_desc, int M, const float* tpp_sums, float coef, float* tpp_part,
                       const float* lam, const void* saved, int B, int T, int C, int H, int E, float drop_rate,
                       const uint64_t* rng_state, uint32_t stream_id, const uint32_t* dropbits, float qk_scale, void* d_qkvt, float* dW1,
                       float* db1, float* dw, float* dscaling, void* workspace, int flags, int dtype, void* stream);
int edgl_tpp_finish_parts(const float* part, int nparts, float coef, int H, const void* tpp_desc, int B, int T, int M, float* sums,
                          float* reg_out, int accumulate, void* stream);
int edgl_tpp_finish_parts_n(const float* part, int nparts, float coef, int H, float* sums, float* reg_out, int accumulate, void* stream);

/* ---- optimizer — tf.train.AdamOptimizer (Base.py:142-144) over a flat f32 arena ----------------
 * step_state: device uint64[2]: [0] = step count (incremented by this call, so the first call is
 * t = 1), [1] = scratch holding lr_t = lr*sqrt(1-b2^t)/(1-b1^t).  theta -= lr_t * m/(sqrt(v)+eps).
 * l2 (coding.py:34-40) enters as grad += l2 * w on the element ranges [seg[2s], seg[2s+1]) (device
 * int64 pairs).  If shadow != NULL the updated weights are also written there as bf16. */
int edgl_adam_step(float* param, const float* grad, float* m, float* v, long n, float lr, float beta1,
                   float beta2, float eps, uint64_t* step_state, float l2, const int64_t* seg, int nseg,
                   void* shadow, void* stream);
/* Training-step form of the two state updates: edgl_step_begin = edgl_rng_advance + the step / learning-rate half of
 * edgl_adam_step in ONE single-thread launch at the start of the step; edgl_adam_apply = the parameter-update half (reads
 * the learning rate edgl_step_begin left in step_state). */
int edgl_step_begin(uint64_t* rng_state, uint64_t* adam_state, float lr, float beta1, float beta2, void* stream);
int edgl_adam_apply(float* param, const float* grad, float* m, float* v, long n, float beta1, float beta2, float eps,
                    const uint64_t* step_state, float l2, const int64_t* seg, int nseg, void* shadow, void* stream);
/* l2 part of the loss: out[0] (+)= 0.5*l2*sum(w[seg]^2)  (EasyDGL.py:158); workspace >= 1024 floats. */
/* edgl_adam_apply that also leaves the per-block sums of squares of the UPDATED parameters inside the l2 segments in l2_part
 * (edgl_adam_l2_parts(n) floats): the next step's L2 loss term (coding.py:40) is then edgl_l2_from_parts(l2_part, nparts, l2_reg, ...)
 * = l2_reg / 2 * sum — one tiny launch that reads nothing of the parameter arena. */
int edgl_adam_l2_parts(long n);
int edgl_adam_apply_l2p(float* param, const float* grad, float* m, float* v, long n, float beta1, float beta2, float eps,
                        const uint64_t* step_state, float l2, const int64_t* seg, int nseg, void* shadow, float* l2_part, void* stream);
/* edgl_adam_apply_l2p (l2_part may be NULL) + edgl_step_begin in ONE launch: the last workgroup to finish advances the dropout step
 * counter, the Adam step and its bias-corrected learning rate (Base.py:142-144's global step) for the NEXT step.  ticket:
 * edgl_adam_next_tickets(n) zero-initialised uint32 of the caller's, left at zero.  An A/B switch of the training engine. */
int edgl_adam_next_tickets(long n);
int edgl_adam_apply_l2p_next(float* param, const float* grad, float* m, float* v, long n, float beta1, float beta2, float eps,
                             uint64_t* step_state, float l2, const int64_t* seg, int nseg, void* shadow, float* l2_part,
                             uint64_t* rng_state, float lr, uint32_t* ticket, void* stream);
/* The eager engine's optimizer launch: edgl_adam_apply_l2p that also (a) adds the sum of `nslab` partial-gradient slabs to two ranges of
 * the gradient arena — grad[lo .. hi) += sum_s slabs[s * stride + (i - lo)]; the first `zero_first_a` elements of range a take none (row 0
 * of the used item table is the zero constant, coding.py:56-57): the row-chunk slabs that edgl_score_flash_bwd_ex(defer_label_term & 4)
 * leaves in its workspace (edgl_score_flash_slab_info), so that no slab-reduction launch stands between the scoring and the block-tail
 * backward — and (b) writes the counters of the NEXT step (edgl_step_begin's update: dropout step + 1, Adam step + 1, its bias-corrected
 * learning rate) into step_next / rng_next, buffers OTHER than step_state / rng_cur, which the launch reads: no single-thread launch at
 * the end of a step and no last-workgroup ticket; the caller swaps the buffer pairs behind the launch.  With slabs given, the two ranges
 * of `grad` are ZEROED behind their use: the next step's embedding scatter / one-hot term add into them (no zero-fill launch).  slabs_* / the three counter
 * pointers may be NULL (then that part is left out); l2_part may be NULL.  Base.py:142-144. */
int edgl_adam_apply_ex(float* param, float* grad, float* m, float* v, long n, float beta1, float beta2, float eps,
                       const uint64_t* step_state, float l2, const int64_t* seg, int nseg, void* shadow, float* l2_part,
                       const float* slabs_a, long stride_a, long lo_a, long hi_a, long zero_first_a,
                       const float* slabs_b, long stride_b, long lo_b, long hi_b, int nslab,
                       const uint64_t* rng_cur, uint64_t* step_next, uint64_t* rng_next, float lr, void* stream);
/* out[0] / out[1]: float offsets of the d_table [nslab][I * C] / d_bias [nslab][I - 1] slabs inside the edgl_score_flash_workspace
 * buffer, out[2] = nslab — what edgl_score_flash_bwd_ex leaves there with (defer_label_term & 4) (requires & 1, the whole item range
 * and gscale == NULL; otherwise the flag is ignored and the slabs are reduced as usual).  EasyDGL.py:149-151 backward. */
int edgl_score_flash_slab_info(int R, int C, int I, int n_items, int dtype, long* out);
int edgl_l2_from_parts(const float* l2_part, int nparts, float l2, float* out, int accumulate, void* stream);
int edgl_l2_loss(const float* param, const int64_t* seg, int nseg, float l2, float* out, int accumulate,
                 float* workspace, void* stream);
/* element-wise helpers on `dtype` buffers */
int edgl_cast(const float* src, void* dst, long n, int dtype, void* stream);                 /* dst = (dtype)src      */
int edgl_cast_back(const void* src, float* dst, long n, int accumulate, int dtype, void* stream); /* dst (+)= (f32)src */
int edgl_add(const void* a, const void* b, void* out, long n, int dtype, void* stream);      /* out = a + b           */
/* y = x * keep / (1 - rate) with the counter-based keep mask of (rng_state, stream_id, element index): tf.layers.dropout
 * (Base.py:80,83); calling it on dy with the same arguments is the backward. */
int edgl_dropout(const void* x, void* y, long n, float drop_rate, const uint64_t* rng_state, uint32_t stream_id, int dtype,
                 void* stream);
/* FeedForward tail in one pass (Base.py:83-86, then `seqs_outs *= seqs_masks`, TGAT.py:70 / TiSASREC.py:73):
 * out[r,:] = (dropout(a[r,:]) + b[r,:]) * (ids[r] != 0); b and ids may be NULL (no residual / no row mask).  The dropout
 * mask is the one edgl_dropout draws for the same (rng_state, stream_id, element index).  Backward of the `a` branch:
 * the same call on the upstream gradient with b == NULL; of the `b` branch: edgl_mask_rows. */
int edgl_ff_tail(const void* a, const void* b, const int64_t* ids, long rows, int C, float drop_rate,
                 const uint64_t* rng_state, uint32_t stream_id, void* out, int dtype, void* stream);
int edgl_relu_bwd(const void* dy, const void* y, void* dz, long n, int dtype, void* stream);   /* dz = dy*[y>0] (Base.py:73) */
int edgl_gelu_bwd(const void* dy, const void* pre, void* dz, long n, int dtype, void* stream); /* dz = dy*gelu'(pre), EasyDGL.py:19-32 */
int edgl_add_cols(void* dst, int ld_dst, const void* src, const void* src2, int ld_src, long rows, int ncols,
                  int dtype, void* stream);                                                             /* dst[:, :n] += src[:, :n] (+ src2[:, :n] if not NULL; same ld_src) */

/* ---- K11: causal attention with a time feature map (config 5: TGAT) ---------------------------------
 * Replaces TfMultiHeadAttention.__call__ (src/module/temporal.py:126-184) after its three dense layers, with
 * TimeFunctionCoding.code (src/module/coding.py:104-122) folded into the operands.
 *
 * edgl_timefn_fwd: q [B,T,C] (row stride ldq), k [B,T,C] (ldk), pos_tab f32 [T,C] (pcoding_K), ts f32 [B,T+1] raw seconds
 * (RegressivePostProcessor layout, dataloader.py:95-108), omega/phi f32 [C] (basis_freq, phase) ->
 *   qx [B,T,H,3*dh] = [ Q | Q*cos(a w + phi) | Q*sin(a w + phi) ],  kx [B,T,H,3*dh] = [ K + pos | cos(b w) | sin(b w) ]
 * with a[t] = ts[t+1]/time_scale - base, b[t] = ts[t]/time_scale - base, base = ts[T]/time_scale, so that
 *   qx[q] . kx[k] = sum_d Q[q,d] (K[k,d] + pos[k,d] + cos((ts'[q+1]-ts'[k]) w_d + phi_d))       (temporal.py:143-148)
 * which is the reference's score wherever its max(.,0) clamp (TGAT.py:54) is inactive: every unmasked (k <= q) pair of a
 * sequence with non-decreasing timestamps.  *violations (int32, device) += number of unpadded positions t with
 * ts[t+1] < ts[t]; the caller must treat a non-zero count as an input error.
 * edgl_timefn_bwd: d_qx, d_kx -> d_q (ld_dq), d_k (ld_dk; this is also d(pos_tab) before the sum over the batch),
 * d_omega, d_phi f32 [C] (overwritten).  workspace >= edgl_timefn_bwd_workspace(C) floats.
 *
 * edgl_tattn_fwd: generic masked attention per (sample, head): S = scale * qx_h . kx_h^T with head slices
 * qx[:, :, h*Dq:(h+1)*Dq] (same for kx; v and out use Dv); scores of padded keys (ids == 0) and, with EDGL_TATTN_CAUSAL,
 * of keys k > q are REPLACED by float32(-2^32+1) (temporal.py:153-166: a fully masked row is uniform over all T keys);
 * P = softmax(S); out = dropout(P) . v_h + resid (temporal.py:169-181).  Dv in {16,32,64,128}, Dq in {Dv, 3*Dv}.
 * saved (NULL for inference): edgl_tattn_saved_bytes bytes = row max / sum / dO.O [H*B*T] f32 each + O before the
 * residual [B,T,H*Dv] f32.  edgl_tattn_bwd (Dq in {Dv, 3*Dv}): d_out -> d_qx, d_kx, d_v (the residual gradient is d_out
 * itself); no gradient flows into a replaced score. */
#define EDGL_TATTN_CAUSAL 1
long edgl_tattn_saved_bytes(int B, int T, int H, int Dv);
int edgl_tattn_fwd(const void* qx, int ldq, const void* kx, int ldk, const void* v, int ldv, const void* resid, int ldr,
                   const int64_t* ids, int B, int T, int H, int Dq, int Dv, float scale, float drop_rate,
                   const uint64_t* rng_state, uint32_t stream_id, void* out, int ldo, void* saved, int flags, int dtype,
                   void* stream);
int edgl_tattn_bwd(const void* qx, int ldq, const void* kx, int ldk, const void* v, int ldv, const int64_t* ids,
                   const void* d_out, int ld_do, void* saved, int B, int T, int H, int Dq, int Dv, float scale,
                   float drop_rate, const uint64_t* rng_state, uint32_t stream_id, void* d_qx, int ld_dq, void* d_kx,
                   int ld_dk, void* d_v, int ld_dv, int flags, int dtype, void* stream);
int edgl_timefn_fwd(const void* q, int ldq, const void* k, int ldk, const float* pos_tab, const float* ts,
                    const int64_t* ids, const float* omega, const float* phi, int B, int T, int C, int H,
                    float time_scale, void* qx, void* kx, int* violations, int dtype, void* stream);
long edgl_timefn_bwd_workspace(int C);
int edgl_timefn_bwd(const void* q, int ldq, const float* ts, const float* omega, const float* phi, const void* d_qx,
                    const void* d_kx, int B, int T, int C, int H, float time_scale, void* d_q, int ld_dq, void* d_k,
                    int ld_dk, float* d_omega, float* d_phi, float* workspace, int dtype, void* stream);
/* y = x * (ids != 0) per row: `seqs_outs *= seqs_masks` (TGAT.py:58,70; TiSASREC.py:73); its own backward. */
int edgl_mask_rows(const void* x, const int64_t* ids, void* y, long rows, int C, int dtype, void* stream);

/* ---- K11b: causal attention with interval buckets (config 5: TiSASRec) ------------------------------
 * Replaces TiMultiHeadAttention.__call__ (src/module/temporal.py:36-105) after its three dense layers.
 * q [B,T,H*dh] (ldq); k, v = K + pcoding_K[t], V + pcoding_V[t] (edgl_add_pos2; temporal.py:50-51,57,94), strides ldk, ldv;
 * ts f32 [B,T+1] raw seconds; ktime, vtime [tab_rows, H*dh] in `dtype` (tcoding_K / tcoding_V tables, head slice
 * h*dh..); bucket(q,k) = int(clip(ts[q+1]/time_scale - ts[k]/time_scale, 0, timelen)) (TiSASREC.py:58-62), a bucket
 * >= tab_rows reads a zero row (the GPU embedding lookup of the reference for the index `timelen` of a [timelen, C]
 * table).  S[q,k] = scale * (q.k + q.ktime[bucket]); masks and softmax as edgl_tattn_fwd; out = dropout(P) . (v +
 * vtime[bucket]) + resid.  The query mask of temporal.py:84-88 is the identity for LayerNorm-ed queries and is not
 * applied.  T <= 256.  saved: edgl_tattn_saved_bytes(B,T,H,dh) bytes; wbuf: edgl_tiattn_bucket_elems elements of `dtype` (binned
 * probabilities, read by the backward); both NULL for inference.  timelen <= 256.
 * edgl_tiattn_bwd: d_q, d_k, d_v; d_ktime, d_vtime f32 [tab_rows, H*dh] (overwritten); dgbuf: workspace like wbuf
 * (binned score gradients). */
long edgl_tiattn_bucket_elems(int B, int T, int H, int timelen);
int edgl_tiattn_fwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const void* resid, int ldr,
                    const int64_t* ids, const float* ts, const void* ktime, const void* vtime, int tab_rows, int B, int T,
                    int H, int dh, float scale, float time_scale, int timelen, float drop_rate, const uint64_t* rng_state,
                    uint32_t stream_id, void* out, int ldo, void* saved, void* wbuf, int flags, int dtype, void* stream);
int edgl_tiattn_bwd(const void* q, int ldq, const void* k, int ldk, const void* v, int ldv, const int64_t* ids,
                    const float* ts, const void* ktime, const void* vtime, int tab_rows, const void* d_out, int ld_do,
                    void* saved, void* wbuf, int B, int T, int H, int dh, float scale, float time_scale, int timelen,
                    float drop_rate, const uint64_t* rng_state, uint32_t stream_id, void* d_q, int ld_dq, void* d_k,
                    int ld_dk, void* d_v, int ld_dv, void* dgbuf, float* d_ktime, float* d_vtime, int flags, int dtype,
                    void* stream);
/* out[b,t,:] = kv[b,t,:] + concat(posK[t], posV[t]); kv, out [B,T,2C] `dtype`, posK/posV f32 [>=T, C]. */
int edgl_add_pos2(const void* kv, const float* posK, const float* posV, int B, int T, int C, void* out, int dtype,
                  void* stream);

/* ---- operator-level forms of src/module/coding.py (the classes' own __call__ / code methods) ------
 * The model kernels fuse these (edgl_encode_fwd, edgl_embed_pos_fwd, edgl_timefn_fwd, edgl_tiattn_fwd); the entry
 * points below are what `C.Embedding(...)(ids)`, `C.PositionCoding(...).code(x)`, `C.TimeIntervalCoding(...).code(x)`,
 * `C.TimeSinusoidCoding(C).code(x)` and `C.TimeFunctionCoding(C).code(x)` bind to when called on their own.
 * edgl_embedding_fwd — Embedding.__call__, coding.py:60-64: out[r,:] = scale * table[ids[r],:] over n flat indices;
 *   table [rows, C] in `dtype`; zero_pad != 0: row 0 acts as a zero constant (coding.py:56-57); an index outside
 *   [0, rows) reads zeros (tf.nn.embedding_lookup on the GPU).  scale = sqrt(num_units) or 1 (coding.py:62-63).
 *   PositionCoding.code (coding.py:76-79) is this gather on ids = tile(range(T)); TimeIntervalCoding.code
 *   (coding.py:93-94) on the integer intervals.
 * edgl_embedding_bwd: d_table f32 [rows, C] (overwritten) += scale * d_out at the gathered rows.
 * edgl_time_sinusoid — TimeSinusoidCoding.code, coding.py:137-149: x f32 [n]; out[r, 2j] = sin(x[r] / tscale[j]),
 *   out[r, 2j+1] = cos(same); tscale f32 [C/2] = 10000^(2j/C) (coding.py:134).
 * edgl_time_function_fwd — TimeFunctionCoding.code, coding.py:113-122: out[r, c] = cos(x[r] * freq[c] + phase[c]).
 * edgl_time_function_bwd: d_freq, d_phase f32 [C] (overwritten); workspace: edgl_time_function_bwd_workspace(C) floats. */
int edgl_embedding_fwd(const int64_t* ids, long n, const void* table, int rows, int C, int zero_pad, float scale, void* out,
                       int dtype, void* stream);
int edgl_embedding_bwd(const int64_t* ids, long n, const void* d_out, int rows, int C, int zero_pad, float scale,
                       float* d_table, int dtype, void* stream);
int edgl_time_sinusoid(const float* x, long n, const float* tscale, int C, void* out, int dtype, void* stream);
int edgl_time_function_fwd(const float* x, long n, const float* freq, const float* phase, int C, void* out, int dtype,
                           void* stream);
long edgl_time_function_bwd_workspace(int C);
int edgl_time_function_bwd(const float* x, long n, const float* freq, const float* phase, int C, const void* d_out,
                           float* d_freq, float* d_phase, float* workspace, int dtype, void* stream);

/* ---- fused per-sample block tail — EasyDGL.py:110-139 in ONE launch (csrc/k_tail.hip) ----------------
 * The reference's LayerNorm is joint over (T, C) per sample (Base.py:12-67), so a workgroup owning a sample runs
 *   ao = att.Wo + bo ; a1 = LN1(dropout(ao) + x_in) ; f = gelu(a1.Wi + bi) ; o = f.Wout + bout ; y = LN2(dropout(o) + a1)
 *   head != 0:  so = gelu(y.Wt + bt) ; hrows[b*M + j] = LN3(so)[masked_pos[b, j]]                     (EasyDGL.py:136-146)
 *               hrow_map (optional, int32 [B*M]): the row goes to hrows[hrow_map[b*M + j]] instead, rows with a negative entry
 *               are dropped — with edgl_compact_scan_labels' `inv` the weighted rows land compacted, no edgl_compact_gather
 * with the activations in LDS between the steps; every intermediate the backward reads is written once:
 * ao, a1, o, y, pre_t, so [B,T,C]; pre_f, f [B,T,2C]; st1/st2/st3 f32 [B,2] = (mean, rstd).  pre_f and pre_t receive
 * gelu'(pre-activation), not the pre-activation: the derivative is all edgl_tail_bwd needs them for, and the forward has the
 * erf at hand (the unfused kernels save the pre-activation itself — do not mix the two paths within a block).  Otherwise the
 * same arithmetic, the same dropout element indices (streams sid1 / sid2) and the same saved tensors as edgl_gemm +
 * edgl_add_layernorm_fwd, which remain the path for every other shape: edgl_tail_supported(T, C, dtype) != 0 iff dtype is EDGL_BF16, C in {64, 128},
 * T <= 112.  att [B,T,C]; xin = the block input's first C channels, row stride ld_x.
 * pack: edgl_tail_pack_elems(C) elements of `dtype` written by edgl_tail_pack from the four [in, out] kernels (compute
 * copies) — their [out][in] images, the MFMA operand layout. */
long edgl_tail_pack_elems(int C);
int edgl_tail_supported(int T, int C, int dtype);
/* Launch form of edgl_tail_fwd / edgl_tail_bwd at C = 128, T <= 101 (same results, bit for bit): 1 (default) = TWO 8-wave
 * workgroups per CU (three unpadded, swizzled LDS images = 80 KB, <= 128 registers: a second independent chain of dependent
 * phases on every CU), 0 = one workgroup per CU (four padded images + an f32 image, 256 registers — the form of every other
 * shape).  variant < 0 only queries.  Returns the previous value.  Initial value: environment EDGL_TAIL2 (unset = 1).
 * Process-wide, not thread-safe against concurrent launches: a development / A-B switch. */
int edgl_tail_variant(int variant);
int edgl_tail_pack(const void* Wo, const void* Wi, const void* Wout, const void* Wt, int C, void* pack, void* stream);
int edgl_tail_fwd(const void* att, const void* xin, int ld_x, const void* pack, const float* bo, const float* bi,
                  const float* bout, const float* bt, const float* g1, const float* b1, const float* g2,
                  const float* b2, const float* g3, const float* b3, int B, int T, int C, float drop_rate,
                  const uint64_t* rng_state, uint32_t sid1, uint32_t sid2, const int64_t* masked_pos, int M, int head,
                  void* ao, void* a1, float* st1, void* pre_f, void* f, void* o, void* y, float* st2, void* pre_t,
                  void* so, float* st3, void* hrows, const int32_t* hrow_map, int dtype, void* stream);
/* The same for a channel-padded model (head dim dh_true stored as dh_pad, a power of two; padded channels of every operand exactly
 * zero: model/easydgl.py): the three LayerNorms take their joint (T, C) moments over the REAL channels — divisor T * C_true, padded
 * entries left out of the centred second moment — as edgl_add_layernorm_fwd_ct does.  0, 0 = edgl_tail_fwd. */
int edgl_tail_fwd_ct(const void* att, const void* xin, int ld_x, const void* pack, const float* bo, const float* bi,
                     const float* bout, const float* bt, const float* g1, const float* b1, const float* g2,
                     const float* b2, const float* g3, const float* b3, int B, int T, int C, float drop_rate,
                     const uint64_t* rng_state, uint32_t sid1, uint32_t sid2, const int64_t* masked_pos, int M, int head,
                     void* ao, void* a1, float* st1, void* pre_f, void* f, void* o, void* y, float* st2, void* pre_t,
                     void* so, float* st3, void* hrows, const int32_t* hrow_map, int dh_pad, int dh_true, int dtype,
                     void* stream);

/* Backward of the same chain, one launch per block: given the gradient of the gathered head rows (head != 0: d_rows [*, C]
 * compact, masked_pos [B, M], dy_rowmap = the `inv` map of edgl_compact_rows or NULL) or of y (head == 0: d_y_in [B,T,C]),
 * recomputes the three LayerNorm inputs from the saved tensors (pre_f / pre_t as edgl_tail_fwd wrote them: gelu') and writes
 * what the weight-gradient GEMMs consume:
 * d_pre_t, d_o, d_ao [B,T,C] and d_pre_f [B,T,2C] (gradients w.r.t. the four dense outputs), d_res1 (gradient into the LN1
 * residual x_in[:, :, :C]) and d_att (gradient w.r.t. the block-tail input); LayerNorm parameter gradients dg*, db*
 * (overwritten; reduced from per-sample partials in `workspace`, edgl_tail_bwd_workspace(B, C) floats).  Wo, Wi, Wout, Wt: the
 * [in, out] kernels' compute copies (no packed image needed: dX contracts over the output index). */
long edgl_tail_bwd_workspace(int B, int C);
int edgl_tail_bwd(const void* xin, int ld_x, const void* ao, const void* a1, const void* pre_f, const void* o,
                  const void* pre_t, const void* so, const float* st1, const float* st2, const float* st3,
                  const void* Wo, const void* Wi, const void* Wout, const void* Wt, const float* g1, const float* g2,
                  const float* g3, int B, int T, int C, float drop_rate, const uint64_t* rng_state, uint32_t sid1,
                  uint32_t sid2, int head, const void* d_rows, const int64_t* masked_pos, int M,
                  const int32_t* dy_rowmap, const void* d_y_in, void* d_pre_t, void* d_o, void* d_pre_f, void* d_ao,
                  void* d_res1, void* d_att, float* dg1, float* db1, float* dg2, float* db2, float* dg3, float* db3,
                  float* workspace, int dtype, void* stream);
/* ... of a channel-padded model: the LayerNorm backward divides by T * C_true and returns nothing into a padded channel
 * (edgl_add_layernorm_bwd_act_ct's rule), so every gradient tensor keeps exact zeros there. */
int edgl_tail_bwd_ct(const void* xin, int ld_x, const void* ao, const void* a1, const void* pre_f, const void* o,
                     const void* pre_t, const void* so, const float* st1, const float* st2, const float* st3,
                     const void* Wo, const void* Wi, const void* Wout, const void* Wt, const float* g1, const float* g2,
                     const float* g3, int B, int T, int C, float drop_rate, const uint64_t* rng_state, uint32_t sid1,
                     uint32_t sid2, int head, const void* d_rows, const int64_t* masked_pos, int M,
                     const int32_t* dy_rowmap, const void* d_y_in, void* d_pre_t, void* d_o, void* d_pre_f, void* d_ao,
                     void* d_res1, void* d_att, float* dg1, float* db1, float* dg2, float* db2, float* dg3, float* db3,
                     float* workspace, int dh_pad, int dh_true, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EASYDGL_HIP_H */

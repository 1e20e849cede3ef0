"""Static training engine for the EasyDGL hot path.

`model.train_step` (autograd over the ops) is the flexible path; this engine is the production path for a
fixed batch shape: every activation / gradient / workspace buffer is allocated once (sized for the 288 GB of
an MI355X, nothing is freed or reallocated per step), the forward and backward kernels are issued in a fixed
order straight through the C ABI, parameter gradients land directly in the flat gradient arena (each written
exactly once, except the item table: scoring writes it, the embedding scatter adds to it), GELU' rides in the
epilogue of the dX GEMM, bias gradients ride in the dW kernel, the l2 gradient is folded into the Adam
kernel — and the whole step can be captured into ONE HIP graph (dropout step / Adam step live in device memory).

The arithmetic is the same as `EasyDGL.train_loss` + backward; tests/test_gpu_engine.py checks gradients and
weights against the autograd path and the fp64 oracle.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from . import _lib, ops
from ._lib import check, lib
from .ops import _ptr, _stream

EPI_BIAS, EPI_GELU, EPI_SAVE_PRE, EPI_MUL_DGELU, EPI_ACCUM, EPI_OUT_F32 = 1, 2, 4, 8, 16, 32


class TrainEngine:
    _SIDE_STREAMS = {}

    def __init__(self, model, batch: int, use_graph: bool = True, process_group=None, fused_tail=None, flash_ce=None):
        if use_graph and os.environ.get("EDGL_ENGINE_LEGACY_FORK", "0") == "1":
            raise _lib.EdglError("TrainEngine: EDGL_ENGINE_LEGACY_FORK=1 is an A/B switch of the eager path (the captured sequence "
                                 "relies on the step counters being advanced behind the optimizer)")
        self.m = model
        self.B = batch
        self.use_graph = use_graph
        self.group = process_group
        self.graph: Optional[torch.cuda.CUDAGraph] = None
        m = model
        dev = m._arena.device
        self.dev = dev
        self.dt = m.act_dtype
        self.code = ops._DT[self.dt]
        B, T, C, H, E, M, I = batch, m.seqslen, m.num_units, m.num_heads, m.num_events, m.masklen, m.num_items
        self.T, self.C, self.H, self.E, self.M, self.I = T, C, H, E, M, I
        self.pad = tuple(getattr(m, "pad", (0, 0)))             # (dh_pad, dh_true) of a channel-padded model (model/easydgl.py)
        self.c_true = int(m.width_true) if self.pad[0] else 0
        self.qk_scale = float(getattr(m, "qk_scale", 0.0))      # 0: 1 / sqrt(head dim); a channel-padded model passes 1 / sqrt(true head dim)
        # More mark types than one attention launch takes (16; EasyDGL.py:45-46: E is the width of the data set's mark.pkl): the
        # marks run as groups, as in module/temporal.py modulated_attention — G = sum_e marks.lambda_e is a sum over marks, the
        # output (G * P) V is linear in G and lambda_e reads only its own dh columns of the intensity MLP.  One attention launch
        # per group on its column block of the weights, the SAME dropout stream; group 0 carries the residual and the diagonal 1,
        # the later groups a zero residual and the diagonal 0; outputs and d_qkvt add, lambda / d lambda concatenate.
        gmax = int(lib.edgl_bimau_mark_group(C, H, self.code))
        if gmax <= 0:
            raise _lib.EdglError(f"TrainEngine: head dim {C // H} / {self.dt} unsupported by the attention kernels")
        self.mgroups = [(e0, min(E, e0 + gmax)) for e0 in range(0, E, gmax)] if E > gmax else []
        self.R = B * M
        self.rows = B * T
        nb = len(m.layers)
        # TPP regulariser inside the attention kernels (csrc/bimau_common.h TppDesc): the forward forms the two loss sums from the
        # lambda rows it holds in registers, sweep 1 recomputes d lambda — no [H*B, T, E] d lambda array, no TPP launch between the
        # attention forward and the block tail (the loss kernel needs the term at the end of the backward only: flash_ce form)
        fce = (os.environ.get("EDGL_FLASH_CE", "1") != "0") if flash_ce is None else bool(flash_ce)
        self.fused_tpp = (self.code == _lib.BF16 and C // H == 16 and E == 16 and not self.mgroups and M <= 256 and T <= 128
                          and m.ct_reg != 0.0 and fce and nb > 0 and os.environ.get("EDGL_TPP_FUSED", "1") != "0")
        e = lambda *s, dtype=None: torch.empty(s, device=dev, dtype=dtype or self.dt)  # noqa: E731
        f32 = torch.float32
        # ---- static inputs ----------------------------------------------------------------------------------
        self.ids = torch.zeros((B, T), device=dev, dtype=torch.int64)
        self.ts = torch.zeros((B, T), device=dev, dtype=f32)
        self.mpos = torch.zeros((B, M), device=dev, dtype=torch.int64)
        self.labels = torch.zeros((B, M), device=dev, dtype=torch.int64)
        # ---- forward activations --------------------------------------------------------------------------------
        self.x0 = e(B, T, 3 * C)
        self.spans = e(B, T, dtype=f32)
        self.marks = e(B, T, E, dtype=torch.uint8)
        self.blk = []
        for i in range(nb):
            d = dict(qkvt=e(B, T, 4 * C), att=e(B, T, C), lam=e(H * B, T, E, dtype=f32), ao=e(B, T, C), a1=e(B, T, C),
                     st1=e(B, 2, dtype=f32), pre_f=e(B, T, 2 * C), f=e(B, T, 2 * C), o=e(B, T, C), y=e(B, T, C),
                     st2=e(B, 2, dtype=f32),
                     pack=e(0 if self.mgroups else lib.edgl_bimau_pack_bytes(C, H, E, self.code), dtype=torch.uint8),
                     saved=e(0 if self.mgroups else lib.edgl_bimau_saved_bytes(B, T, C, H, self.code), dtype=torch.uint8),
                     dlam=None if self.fused_tpp else e(H * B, T, E, dtype=f32),
                     tpp_part=torch.zeros((B * H + 1, 2), device=dev, dtype=f32) if self.fused_tpp else None,   # (+ the count)
                     tpp=torch.zeros(max(lib.edgl_tpp_workspace(), lib.edgl_tpp_rows_workspace(B, H, M)), device=dev, dtype=f32))
            # stored keep bits of the attention dropout: hashed once per step on the side stream, read by the forward and both
            # backward sweeps (csrc/bimau_common.h; 0 bytes = no stored-bits form at this T)
            nb_bits = int(lib.edgl_bimau_dropbits_bytes(B, T, H)) if m.attention_probs_dropout_rate > 0 else 0
            d["dbits"] = e(nb_bits // 4, dtype=torch.int32) if nb_bits > 0 and os.environ.get("EDGL_DROPBITS", "1") != "0" and os.environ.get("EDGL_ENGINE_LEGACY_FORK", "0") != "1" else None
            if self.mgroups:
                dh = C // H
                d["grp"] = []
                for (e0, e1) in self.mgroups:
                    eg = e1 - e0
                    d["grp"].append(dict(
                        marks=e(B, T, eg, dtype=torch.uint8), W1=e(dh + 1, dh * eg, dtype=f32), dW1=e(dh + 1, dh * eg, dtype=f32),
                        pack=e(lib.edgl_bimau_pack_bytes(C, H, eg, self.code), dtype=torch.uint8),
                        saved=e(lib.edgl_bimau_saved_bytes(B, T, C, H, self.code), dtype=torch.uint8),
                        lam=e(H * B, T, eg, dtype=f32), dlam=e(H * B, T, eg, dtype=f32),
                        out=e(B, T, C) if e0 else None, dqkvt=e(B, T, 4 * C) if e0 else None))
            self.blk.append(d)
        # launch order of the attention jobs (samples by falling key-tile count: csrc/k_bimau_fwd.hip edgl_bimau_job_order) for the
        # kernels that leave out the all-padding key tiles.  OFF for training unless EDGL_BIMAU_ORDER=1: MAUPostProcessor.mask_random
        # (dataloader.py:187-191) draws the masked positions over ALL positions, the MASK token (id num_items != 0) is a real key
        # (temporal.py:425: the key mask is ids != 0), and with 20 of 100 positions masked hardly any key tile of a training batch is
        # padding only (0.4 % at the headline shape: DESIGN.md rule 50) — nothing to skip, nothing to balance.  Evaluation batches
        # (mask_last) keep their left padding: there the skip is real.
        self.job_order = (torch.arange(2 * B, device=dev, dtype=torch.int32)     # (order | the launches' scratch)
                          if (not self.mgroups and B <= 16384 and nb > 0 and os.environ.get("EDGL_BIMAU_ORDER", "0") == "1") else None)
        self.tpp_desc = torch.zeros(int(lib.edgl_tpp_prep_bytes(B, T, M)), device=dev, dtype=torch.uint8) if self.fused_tpp else None
        self.zero_resid = torch.zeros((B, T, C), device=dev, dtype=self.dt) if self.mgroups else None
        self.pre_t, self.so, self.st3 = e(B, T, C), e(B, T, C), e(B, 2, dtype=f32)
        # fused per-sample block tail (csrc/k_tail.hip): one launch for dense -> LN -> GELU-dense -> dense -> LN (-> head)
        # (the head of the fused tail and the fused TPP kernel hold the masked positions of a sample in LDS: M <= 256, T <= 1024 —
        # beyond what the BiMAU kernels take (T <= 208), checked here so that a future relaxation fails at construction)
        if m.ct_reg != 0.0 and (M > 256 or T > 1024):
            raise _lib.EdglError(f"TrainEngine: masklen {M} > 256 or T {T} > 1024 exceeds the fused TPP kernel (edgl_tpp_fwd_bwd_ex)")
        # (channel-padded models: the _ct forms take the LayerNorms' moments over the real channels; EDGL_FUSED_TAIL_PAD=0: unfused)
        ok = bool(lib.edgl_tail_supported(T, C, self.code)) and M <= 256 and os.environ.get("EDGL_FUSED_TAIL", "1") != "0" \
            and (not self.pad[0] or os.environ.get("EDGL_FUSED_TAIL_PAD", "1") != "0")
        self.fused_tail = ok if fused_tail is None else (bool(fused_tail) and ok)
        self.tail_pack = [e(int(lib.edgl_tail_pack_elems(C))) for _ in range(nb)] if self.fused_tail else []
        if self.fused_tail:   # outputs of the fused backward: the gradients w.r.t. the four dense outputs (operands of the dW GEMMs)
            self.d_pre_t, self.d_o, self.d_pre_f, self.d_ao = e(B, T, C), e(B, T, C), e(B, T, 2 * C), e(B, T, C)
        # "flash" scoring: the forward LSE pass also accumulates the row gradients (edgl_score_flash_fwd / _bwd); its
        # workspace carries the slabs from the forward to the backward and is therefore private
        self.flash_ce = (os.environ.get("EDGL_FLASH_CE", "1") != "0") if flash_ce is None else bool(flash_ce)
        self.ws_flash = e(int(lib.edgl_score_flash_workspace(self.R, C, I, I, self.code)), dtype=f32) if self.flash_ce else None
        self.hrows, self.hrows_c = e(self.R, C), e(self.R, C)
        self.hrows_c.zero_()   # rows behind the weighted ones are never written on the fused path (and never read as data)
        self.labels_c = torch.zeros(self.R, device=dev, dtype=torch.int64)
        self.perm, self.inv = e(self.R, dtype=torch.int32), e(self.R, dtype=torch.int32)
        self.nvalid = torch.zeros(1, device=dev, dtype=torch.int32)
        self.lse, self.lab_logit, self.coef = e(self.R, dtype=f32), e(self.R, dtype=f32), e(self.R, dtype=f32)
        self.loss = e(1, dtype=f32)
        # per-workgroup sums of the loss numerator, left by the one-launch row finish of the scoring forward (0: this width has none)
        self.ce_nparts = int(lib.edgl_score_ce_nparts(self.R, C)) if (self.flash_ce and os.environ.get("EDGL_CE_PARTS", "1") != "0") else 0
        self.ce_part = torch.zeros(self.ce_nparts + 1, device=dev, dtype=f32) if self.ce_nparts > 0 else None   # (+ the row count)
        # L2 + TPP terms (accumulated before the cross-entropy kernel, which adds them to its own term)
        self.loss_aux, self.loss_tpp = torch.zeros(1, device=dev, dtype=f32), torch.zeros(1, device=dev, dtype=f32)
        self.ws_l2 = e(1024, dtype=f32)
        # The L2 term of a step from the sums of squares the PREVIOUS step's optimizer kernel left (edgl_adam_apply_l2p): no pass over
        # the arena on the side stream — a read that nothing ordered against the optimizer launch at the end of the same step once the
        # lazy-loss mode dropped the join in front of it.  Two copies, alternating: the optimizer of step n writes the copy that step
        # n + 1 reads, and step n + 2's main stream is behind that read through the forward's join.  Eager path only (a captured
        # graph bakes one address in); whenever somebody else touched the weights or the step counters (`_state_ahead` False: first
        # step, checkpoint, an autograd-path step) the term is recomputed from the arena.
        self.l2_nparts = int(lib.edgl_adam_l2_parts(m._arena.numel()))
        self.l2_parts = [torch.zeros(self.l2_nparts, device=dev, dtype=f32) for _ in range(2)] \
            if (m.l2_reg != 0.0 and not use_graph and os.environ.get("EDGL_L2_PARTS", "1") != "0") else None
        self._l2p_cur, self._l2p_ready = 0, False
        # two side streams per DEVICE, shared by every engine of the process: the runtime multiplexes streams onto a handful of
        # hardware queues, and a process that builds engine after engine (bench.py's extra rows) otherwise ends up with its main
        # and side streams on ONE queue — no overlap and false dependencies (measured: the later rows 5-13 % slower)
        key = (dev.type, dev.index if dev.index is not None else torch.cuda.current_device())
        if key not in TrainEngine._SIDE_STREAMS:
            TrainEngine._SIDE_STREAMS[key] = (torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev))
        self.side, self.side2 = TrainEngine._SIDE_STREAMS[key]
        self._pending_loss = None
        # a training loop that reads the loss every few hundred steps runs with sync_loss = False (the loss launches of step n ride
        # with step n + 1, nothing in the step waits for them) and lets the engine add the step losses up on the device:
        # join_loss(), then loss_sum / steps (train.py; bench.py times the same mode)
        self.accumulate_loss = False
        self.loss_sum = torch.zeros(1, device=dev, dtype=torch.float64)
        self._pending_label = None
        self._lazy_loss, self._side_has_grads, self._loss_unjoined = False, False, False
        # weight-gradient GEMMs beside the kernels that do not need them (bit 0: the block tail's four products under the attention
        # backward; bit 1: the QKVT product beside the dX GEMM / embedding backward): _issue_backward
        self.dw_overlap = int(os.environ.get("EDGL_DW_OVERLAP", "0"))
        # training batches of the reference's masker leave no key tile of pure padding (MASK tokens sit on padded positions: rule 50),
        # so the engine launches the BiMAU kernels that walk every tile (identical results; EDGL_ENGINE_SKIP=1: the skipping ones)
        # A/B switch: the table-gradient pass adds its row chunks into the zero-filled gradient with f32 atomics (no slabs, no
        # slab_reduce launch between the scoring and the block-tail backward)
        self.score_atomic = (os.environ.get("EDGL_SCORE_ATOMIC", "0") == "1" and self.code == _lib.BF16 and self.C == 128
                             and os.environ.get("EDGL_SCORE_STRIP", "1") != "0" and bool(self.blk))
        self.mau_flags = 0 if os.environ.get("EDGL_ENGINE_SKIP", "0") == "1" else _lib.MAU_NO_SKIP
        # The optimizer launch of the eager step (edgl_adam_apply_ex): (a) it sums the row-chunk slabs of the tied table's / output bias's
        # scoring gradient itself — no slab_reduce launch between the scoring and the block-tail backward (the embedding scatter adds
        # into the gradient zero-filled under the encoder) —, (b) it writes the NEXT step's counters into a second pair of buffers that
        # the host swaps in behind it — no single-thread step_begin launch at the end of the step's chain.  step() only (a bare
        # _issue() leaves complete gradients in the arena), single process only (the all-reduce wants complete gradients), never with a
        # captured graph on the model (a graph bakes the counters' addresses in).  EDGL_ADAM_EX=0: the round-5 launches.
        self.adam_ex = (not use_graph) and os.environ.get("EDGL_ADAM_EX", "1") != "0" and os.environ.get("EDGL_ENGINE_LEGACY_FORK", "0") != "1" \
            and os.environ.get("EDGL_ADAM_NEXT", "0") != "1"
        self._rng_alt, self._adam_alt = torch.zeros_like(m._rng_state), torch.zeros_like(m._adam_state)
        self._slabs = None          # (table slabs ptr, bias slabs ptr, nslab) left by the scoring backward of the current step
        self._slab_info = None
        if self.adam_ex and self.flash_ce:
            import ctypes
            info = (ctypes.c_long * 4)()
            check(lib.edgl_score_flash_slab_info(self.R, C, I, I, self.code, info), "edgl_score_flash_slab_info")
            self._slab_info = (int(info[0]), int(info[1]), int(info[2]))
        self._dw_forked = False
        self.sync_loss = True      # step(): order the returned loss on the caller's stream (a cross-stream wait behind the optimizer)
        # sync_loss False, the loss launches reading nothing of the batch (ce_part): they are not launched at the end of the backward
        # — the fork for them is an event record behind a kernel of the main stream, 6-8 us of idle — but by the NEXT step on the
        # side stream, behind the event its first attention kernel waits for (or by join_loss()).  What they read — the sums of the
        # scoring rows' finish and of sweep 1, the per-sample mark counts — exists twice and alternates step by step: step n + 1
        # writes the other copy, and step n + 2 is ordered behind the launches by the joins it has anyway (main stream: the event of
        # the side stream; second side stream: its wait for the side stream at the start of a step).
        self._deferred_loss = None
        self._alt = None
        if self.ce_part is not None:
            self._alt = dict(ce_part=torch.zeros_like(self.ce_part), loss_aux=torch.zeros_like(self.loss_aux),
                             tpp_desc=torch.zeros_like(self.tpp_desc) if self.tpp_desc is not None else None,
                             tpp_part=[torch.zeros_like(b["tpp_part"]) if b["tpp_part"] is not None else None for b in self.blk])
        # data parallel: weighted rows / TPP normaliser of the GLOBAL batch (filled by _global_counts before a step)
        self.counts = torch.zeros(2, device=dev, dtype=torch.int32)
        self._dp = False
        # ---- backward temporaries -------------------------------------------------------------------------------
        self.d_rows = e(self.R, C)
        self.G1, self.G2, self.G3, self.G4 = e(B, T, C), e(B, T, C), e(B, T, C), e(B, T, C)
        self.G2c, self.G4c, self.G3c = e(B, T, 2 * C), e(B, T, 4 * C), e(B, T, 3 * C)
        # ---- workspaces -----------------------------------------------------------------------------------------------
        wsz = [2 * self.R * lib.edgl_score_chunks(self.R, I),
               lib.edgl_score_bwd_workspace(self.R, C, I, I, self.code),
               lib.edgl_encode_bwd_workspace(B, T, C), B * 2 * C, 1024]
        for (kf, n) in ((3 * C, 4 * C), (C, 4 * C), (C, C), (C, 2 * C), (2 * C, C)):
            wsz.append(lib.edgl_gemm_dw_workspace(self.rows, kf, n, self.code))
        self.ws = e(max(wsz), dtype=f32)
        # The backward runs with deferred partial reductions (edgl_reduce_defer): every call that leaves partials behind
        # gets its own workspace, handed out in call order (the launch sequence is fixed, so the addresses are too).
        self._ws_list, self._ws_i = [], 0
        segs = []
        params = dict(m.named_parameters())
        for n in m.l2_param_names():
            segs += [m._offsets[n], m._offsets[n] + params[n].numel()]
        self.l2_seg = torch.tensor(segs, device=dev, dtype=torch.int64)
        self.nseg = len(segs) // 2

    # ---- thin call helpers ------------------------------------------------------------------------------------------
    def _gemm(self, A, Bm, Cm, M, N, K, lda, ldb, ldc, b_kc, bias=None, aux=None, flags=0):
        check(lib.edgl_gemm(_ptr(A), _ptr(Bm), _ptr(Cm), M, N, K, lda, ldb, ldc, 1, int(b_kc), _ptr(bias), _ptr(aux), flags,
                            1, None, self.code, _stream()), "edgl_gemm")

    def _dense_fwd(self, x, dense_kernel, dense_bias, out, K, N, gelu=False, pre=None):
        flags = EPI_BIAS | ((EPI_GELU | EPI_SAVE_PRE) if gelu else 0)
        self._gemm(x, self.m.compute(dense_kernel), out, self.rows, N, K, K, N, N, False, bias=dense_bias, aux=pre, flags=flags)

    def _dense_dx(self, dz, kernel, out, K_in, N, flags=0, aux=None):
        """out[rows, K_in] (=|+=) dz[rows, N] . kernel[K_in, N]^T"""
        self._gemm(dz, self.m.compute(kernel), out, self.rows, K_in, N, N, N, K_in, True, aux=aux, flags=flags)

    def _ws(self, n, dtype=torch.float32):
        """Private workspace of the next partial-producing call of the fixed backward sequence."""
        i = self._ws_i
        self._ws_i += 1
        if i == len(self._ws_list):
            self._ws_list.append(torch.empty(max(int(n), 1), device=self.m._arena.device, dtype=dtype))
        w = self._ws_list[i]
        if w.numel() < n or w.dtype != dtype:
            raise _lib.EdglError("TrainEngine: backward launch sequence changed between steps")
        return w

    def _dense_dw(self, x, dz, kernel, bias, K_in, N):
        ws = self._ws(lib.edgl_gemm_dw_workspace(self.rows, K_in, N, self.code))
        check(lib.edgl_gemm_dw(_ptr(x), _ptr(dz), _ptr(kernel.grad), _ptr(bias.grad), self.rows, K_in, N, K_in, N, 0,
                               _ptr(ws), self.code, _stream()), "edgl_gemm_dw")

    def _ln_fwd(self, x, resid, ld_res, ln, drop, y, stats, gpos=None):
        B, T, C = self.B, self.T, self.C
        check(lib.edgl_add_layernorm_fwd_ct(_ptr(x), None if resid is None else resid.data_ptr(), ld_res, _ptr(ln.gamma),
                                            _ptr(ln.beta), B, T, C, float(drop.rate), drop.ptr(), drop.stream_id,
                                            _ptr(gpos), 0 if gpos is None else gpos.shape[1], _ptr(y), _ptr(stats),
                                            self.pad[0], self.pad[1], self.code, _stream()), "edgl_add_layernorm_fwd")

    def _ln_bwd(self, x, resid, ld_res, ln, stats, dy, drop, dsum, dx_drop, gpos=None, rowmap=None, act_pre=None):
        B, T, C = self.B, self.T, self.C
        check(lib.edgl_add_layernorm_bwd_act_ct(_ptr(x), None if resid is None else resid.data_ptr(), ld_res, _ptr(ln.gamma),
                                                _ptr(stats), _ptr(dy), B, T, C, float(drop.rate), drop.ptr(), drop.stream_id,
                                                _ptr(gpos), 0 if gpos is None else gpos.shape[1], _ptr(rowmap), _ptr(act_pre),
                                                _ptr(dsum), _ptr(dx_drop), _ptr(ln.gamma.grad), _ptr(ln.beta.grad),
                                                _ptr(self._ws(B * 2 * C)), self.pad[0], self.pad[1], self.code, _stream()),
              "edgl_add_layernorm_bwd_act")

    # ---- one optimizer step, as a fixed launch sequence ----------------------------------------------------------------
    def _issue(self, lazy_loss: bool = False, fold_slabs: bool = False):
        """lazy_loss (step() without data parallelism): nothing the optimizer needs runs on the side stream at the end of the
        backward — every deferred reduction goes into the main stream's last reduction launch — so the main stream does NOT wait
        for the side stream (the TPP partial reduction and the loss kernel) in front of Adam; step() orders the loss behind the
        optimizer instead (or leaves it to the caller: `sync_loss`).  A cross-stream join costs the main stream ~12 us there."""
        m, st = self.m, _stream()
        self._lazy_loss = bool(lazy_loss) and bool(self.blk)
        # (fold_slabs: step() lets the optimizer launch sum the scoring gradient's slabs — see __init__)
        self._fold = bool(fold_slabs) and self._slab_info is not None and bool(self.blk) and not self.score_atomic and not self._dp
        self._slabs = None
        if not self._fold:
            m._table_grad_zero = None      # (this issue's slab reduction ASSIGNS the table / bias gradient)
        B, T, C, H, E, M, I, R = self.B, self.T, self.C, self.H, self.E, self.M, self.I, self.R
        code = self.code
        hd, ad = m.hidden_dropout_rate, m.attention_probs_dropout_rate
        drop = lambda rate, sid: ops.Drop(rate, m._rng_state, sid) if rate > 0 else ops.NO_DROP  # noqa: E731
        tab = m.item_embs.lookup_table
        tab_c = m.compute(tab)
        # dropout step counter, Adam step counter and learning rate of this step (one single-thread launch): normally already in
        # place — _optimizer advances them for the NEXT step behind its last kernel, so that the fork below costs the main stream
        # nothing (an event record behind a kernel idles the stream ~6-13 us before its next launch) while the side stream can hash
        # the keep bits of the attention dropout from the advanced counter.  First step, or an _issue() without _optimizer: here.
        legacy = os.environ.get("EDGL_ENGINE_LEGACY_FORK", "0") == "1"   # A/B switch: round-3 order (one side stream, fork first)
        # the sums of squares the last optimizer launch of THIS engine left are the L2 term's input iff nobody touched the state since
        # (ownership on the MODEL: another engine of the same model, load_tf_variables / _load_padded / a checkpoint — anything that
        #  rewrites the arena goes through sync_shadow / settle_state or another engine's _optimizer and takes the token away)
        l2_from_parts = self.l2_parts is not None and self._l2p_ready and getattr(m, "_state_ahead", False) and not legacy \
            and getattr(m, "_l2_parts_owner", None) == id(self)
        if not legacy:
            if not getattr(m, "_state_ahead", False):
                self._advance_state(st)
            m._state_ahead = False
        # ---- side streams: launches that depend on the weights / labels / the step counter only start with the step and run
        # under the encoder / QKVT projection: few-microsecond kernels that would otherwise sit in the critical path, each behind
        # a full launch.  (Small kernels next to the one-workgroup-per-CU kernels — block tail, scoring — is what NOT to do: the
        # fat workgroups cannot be placed while small ones hold registers of a CU; measured 84 -> 247 us.)
        # The first attention kernel waits for this chain, and a cross-stream edge takes ~12 us to arrive: everything it needs
        # must be through ~25 us before the QKVT projection ends (timeline of round 4: the chain ended 10 us AFTER it and the
        # main stream idled 22 us).  Hence: the row-compaction scan — one 1024-thread workgroup, ~24 us — on a stream of its own,
        # the chain itself without the memset of the normaliser, the L2 term BEHIND the event (its consumer, the loss kernel,
        # runs on this same side stream at the end of the backward).
        main, side, side2 = torch.cuda.current_stream(), self.side, self.side2
        if legacy:
            side2 = side
        sst = side.cuda_stream
        if getattr(self, "_loss_unjoined", False) and self.ce_part is None:
            # the previous step left its loss kernel (the form that sweeps lse / label logits / compacted labels) unjoined on the side
            # stream, and this step's first launch rewrites the compacted labels and the row count on the main stream
            main.wait_stream(side)
            self._loss_unjoined = False
        side.wait_stream(main)
        if self._deferred_loss is not None and not legacy:
            # the previous step's loss launches are still to come (behind ev_pack below): this step writes the other copy of their input
            a = self._alt
            self.ce_part, a["ce_part"] = a["ce_part"], self.ce_part
            self.loss_aux, a["loss_aux"] = a["loss_aux"], self.loss_aux      # (this step's L2 term is written in front of those launches)
            self.tpp_desc, a["tpp_desc"] = a["tpp_desc"], self.tpp_desc
            for j, bj in enumerate(self.blk):
                bj["tpp_part"], a["tpp_part"][j] = a["tpp_part"][j], bj["tpp_part"]
        # The batch preparation (row compaction map, slot data of the regulariser) as the first workgroups of the encoder's launch:
        # on a second side stream its join sat in the chain the first attention kernel waits for — a wait costs the waiting stream
        # ~6 us wherever its event stands (measured without it: -5 us of the step)
        prep_enc = not legacy and os.environ.get("EDGL_PREP_IN_ENCODER", "1") != "0"
        if not legacy:
            if not prep_enc:
                side2.wait_stream(main)
                if getattr(self, "_loss_unjoined", False):
                    side2.wait_stream(side)     # the previous steps' loss kernels (side) read buffers that this stream's first kernels rewrite
        else:
            if not getattr(m, "_state_ahead", False):
                self._advance_state(st)
            m._state_ahead = False
        if not prep_enc:
            with torch.cuda.stream(side2):
                # needed by the scoring: row compaction map (labels only).  Its one 1024-thread workgroup needs a whole CU's worth of
                # free wave slots, which it gets beside the small encoder kernel but not once the QKVT projection fills the chip
                # (512-unit recipe: 5 .. 516 us, the main stream waiting for it at the join)
                check(lib.edgl_compact_scan_labels(_ptr(self.labels), R, _ptr(self.perm), _ptr(self.inv), _ptr(self.nvalid),
                                                   _ptr(self.labels_c), side2.cuda_stream), "edgl_compact_scan_labels")
                if self.fused_tpp:
                    # slot data of the batch for the regulariser inside sweep 1 of the attention backward (labels / positions only), with
                    # the per-sample mark counts whose total is the regulariser's normaliser (data parallel: the all-reduced count of
                    # _global_counts is used instead).  Behind the scan: this stream is joined in front of the first attention kernel.
                    check(lib.edgl_tpp_prep(_ptr(self.mpos), _ptr(self.labels), _ptr(self.ts), _ptr(m.mark_lookup_table), B, T, E, M,
                                            _ptr(self.tpp_desc), side2.cuda_stream), "edgl_tpp_prep")

        def l2_term():
            if m.l2_reg != 0.0 and l2_from_parts:
                check(lib.edgl_l2_from_parts(_ptr(self.l2_parts[self._l2p_cur]), self.l2_nparts, float(m.l2_reg), _ptr(self.loss_aux), 0,
                                             sst), "edgl_l2_from_parts")
            elif m.l2_reg != 0.0:
                check(lib.edgl_l2_loss(_ptr(m._arena), _ptr(self.l2_seg), self.nseg, float(m.l2_reg), _ptr(self.loss_aux), 0,
                                       _ptr(self.ws_l2), sst), "edgl_l2_loss")

        # The fused TPP form has no normaliser launch in this chain (edgl_tpp_prep on the other side stream leaves per-sample counts)
        split = self.fused_tpp and not legacy
        with torch.cuda.stream(side):
            # needed by the first BiMAU forward (the one join of the forward): TPP normaliser (labels only), weight packs, keep bits
            def late(i, blk, b):
                if m.ct_reg != 0.0 and not self._dp and not split:    # (data parallel: _global_counts put the all-reduced count there)
                    check(lib.edgl_tpp_norm(_ptr(self.labels), _ptr(m.mark_lookup_table), B, M, E, _ptr(b["tpp"]), sst),
                          "edgl_tpp_norm")
                if self.fused_tail:
                    check(lib.edgl_tail_pack(_ptr(m.compute(blk.att_out.kernel)), _ptr(m.compute(blk.inter.kernel)),
                                             _ptr(m.compute(blk.out.kernel)), _ptr(m.compute(m.transform.kernel)), C,
                                             _ptr(self.tail_pack[i]), sst), "edgl_tail_pack")
            # (fold: the optimizer launch of the previous step left the two gradients at zero behind its reads — a zero-fill only
            #  when somebody else wrote them since: first step, a bare _issue(), an autograd-path step)
            if self.score_atomic or (self._fold and getattr(m, "_table_grad_zero", None) != id(self)):
                tab.grad.zero_()      # (behind the previous step's optimizer: this stream waited for the main stream above)
                m.output_bias.grad.zero_()
            if self.job_order is not None:   # ids only: under the encoder, in front of everything the first attention kernel waits for
                check(lib.edgl_bimau_job_order(_ptr(self.ids), B, T, _ptr(self.job_order), sst), "edgl_bimau_job_order")
            for i, (blk, b) in enumerate(zip(m.layers, self.blk)):
                att = blk.attention
                late(i, blk, b)
                if self.mgroups:
                    dh = C // H
                    for (e0, e1), gb in zip(self.mgroups, b["grp"]):
                        gb["W1"].copy_(att.st_kernel[:, e0 * dh:e1 * dh])      # the group's column block, contiguous
                        check(lib.edgl_bimau_pack(_ptr(gb["W1"]), _ptr(att.st_bias[e0 * dh:e1 * dh]), _ptr(att.weight[e0:e1]),
                                                  _ptr(att.scaling[e0:e1]), C, H, e1 - e0, _ptr(gb["pack"]), code, sst), "edgl_bimau_pack")
                else:
                    check(lib.edgl_bimau_pack(_ptr(att.st_kernel), _ptr(att.st_bias), _ptr(att.weight), _ptr(att.scaling), C, H, E,
                                              _ptr(b["pack"]), code, sst), "edgl_bimau_pack")
                if b["dbits"] is not None and not legacy:
                    check(lib.edgl_bimau_dropbits(B, T, H, float(ad), _ptr(m._rng_state), 10 + 4 * i, _ptr(b["dbits"]), sst),
                          "edgl_bimau_dropbits")
            # The L2 term reads the parameter arena, which the optimizer at the END of this step rewrites on the main stream — and in
            # the lazy-loss mode nothing joins the side stream in front of the optimizer any more.  In FRONT of the event the first
            # attention kernel waits for, the term is ordered before everything the main stream does from there on (round 4 had it
            # behind the event, when the chain still ended after the QKVT projection; since the batch preparation moved into the
            # encoder's launch the chain has ~17 us of slack: rule 49).  EDGL_L2_EARLY=0: behind the event (the A/B switch).
            l2_early = os.environ.get("EDGL_L2_EARLY", "1") != "0" and not l2_from_parts     # (from the optimizer's sums: no arena read, behind the event)
            if not self.blk or legacy or l2_early:
                l2_term()      # (no block: the loss kernel runs on the main stream behind this one event)
            if not legacy and not prep_enc:
                side.wait_stream(side2)
            ev_pack = side.record_event()
            if self._deferred_loss is not None and not legacy:
                # the previous step's loss: behind the event (nothing of this step waits for it), in front of this step's L2 term
                self._deferred_loss(sst)
                self._deferred_loss = None
            # L2 term: not needed before the loss kernel at the end of the backward (same stream)
            # (the transposed table image is NOT prepared here, although it depends on the weights only: written 200 us before
            # its use it has left the L2 by then and the scoring pass measured 109 -> 118 us — edgl_score_prepare_table)
            if self.blk and not legacy and not l2_early:
                l2_term()
        # ================= forward (EasyDGL.py:70-151) =================
        d0 = drop(hd, 1)
        if prep_enc:
            check(lib.edgl_encode_fwd_prep(_ptr(self.ids), _ptr(self.ts), _ptr(tab_c), _ptr(m.pcoding.pembs.lookup_table),
                                           _ptr(m.mark_embs.lookup_table), _ptr(m.mark_lookup_table), _ptr(m.tcoding.scale), B, T, C,
                                           E, I, int(m.mask), float(m.time_scale), float(d0.rate), d0.ptr(), d0.stream_id,
                                           _ptr(self.x0), _ptr(self.spans), _ptr(self.marks), self.pad[0], self.pad[1],
                                           _ptr(self.labels), M, _ptr(self.perm), _ptr(self.inv), _ptr(self.nvalid), _ptr(self.labels_c),
                                           _ptr(self.mpos), _ptr(self.tpp_desc) if self.fused_tpp else None, code, st),
                  "edgl_encode_fwd_prep")
        else:
            check(lib.edgl_encode_fwd_ct(_ptr(self.ids), _ptr(self.ts), _ptr(tab_c), _ptr(m.pcoding.pembs.lookup_table),
                                         _ptr(m.mark_embs.lookup_table), _ptr(m.mark_lookup_table), _ptr(m.tcoding.scale), B, T, C,
                                         E, I, int(m.mask), float(m.time_scale), float(d0.rate), d0.ptr(), d0.stream_id,
                                         _ptr(self.x0), _ptr(self.spans), _ptr(self.marks), self.pad[0], self.pad[1], code, st),
                  "edgl_encode_fwd")
        x, cin = self.x0, 3 * C
        for i, (blk, b) in enumerate(zip(m.layers, self.blk)):
            att = blk.attention
            self._dense_fwd(x, att.dense_kernel, att.dense_bias, b["qkvt"], cin, 4 * C)
            if i == 0:
                main.wait_event(ev_pack)
            da = drop(ad, 10 + 4 * i)
            # the forward also zero-fills this block's d lambda buffer (free beside its VALU work): the TPP launch then writes
            # the rows of the masked positions only — 5 MB instead of 26 MB at the headline shape
            if self.mgroups:
                self._attention_fwd_groups(b, x, cin, da, st)
            else:
                check(lib.edgl_bimau_fwd_ord(_ptr(b["qkvt"]), x.data_ptr(), cin, _ptr(self.ids), _ptr(self.spans), _ptr(self.marks),
                                             _ptr(b["pack"]), B, T, C, H, E, float(da.rate), da.ptr(), da.stream_id, _ptr(b["dbits"]),
                                             self.qk_scale, _ptr(b["att"]), _ptr(b["lam"]), _ptr(b["saved"]),
                                             _ptr(b["dlam"]) if (m.ct_reg != 0.0 and not self.fused_tpp) else None, _ptr(self.job_order),
                                             self.mau_flags, code, st), "edgl_bimau_fwd_ord")
            if m.ct_reg != 0.0 and not self.fused_tpp:   # TPP regulariser of this block: loss term and d lambda (two small launches)
                check(lib.edgl_tpp_fwd_bwd_rows(_ptr(b["lam"]), _ptr(self.mpos), _ptr(self.labels), _ptr(self.ts),
                                                _ptr(m.mark_lookup_table), B, T, H, E, M, float(m.ct_reg / H), _ptr(b["tpp"]),
                                                _ptr(self.loss_tpp), 1 if i > 0 else 0, _ptr(b["dlam"]), st), "edgl_tpp_fwd_bwd_rows")
            if self.fused_tail:
                last = i == len(self.blk) - 1
                pk = self.tail_pack[i]
                dh1 = drop(hd, 11 + 4 * i)
                check(lib.edgl_tail_fwd_ct(_ptr(b["att"]), x.data_ptr(), cin, _ptr(pk), _ptr(blk.att_out.bias), _ptr(blk.inter.bias),
                                        _ptr(blk.out.bias), _ptr(m.transform.bias), _ptr(blk.att_ln.gamma), _ptr(blk.att_ln.beta),
                                        _ptr(blk.out_ln.gamma), _ptr(blk.out_ln.beta), _ptr(m.transform_ln.gamma),
                                        _ptr(m.transform_ln.beta), B, T, C, float(dh1.rate), dh1.ptr(), 11 + 4 * i, 12 + 4 * i,
                                        _ptr(self.mpos), M, int(last), _ptr(b["ao"]), _ptr(b["a1"]), _ptr(b["st1"]), _ptr(b["pre_f"]),
                                        _ptr(b["f"]), _ptr(b["o"]), _ptr(b["y"]), _ptr(b["st2"]), _ptr(self.pre_t), _ptr(self.so),
                                        _ptr(self.st3), _ptr(self.hrows_c), _ptr(self.inv), self.pad[0], self.pad[1], code, st), "edgl_tail_fwd")
            else:
                self._dense_fwd(b["att"], blk.att_out.kernel, blk.att_out.bias, b["ao"], C, C)
                self._ln_fwd(b["ao"], x, cin, blk.att_ln, drop(hd, 11 + 4 * i), b["a1"], b["st1"])
                self._dense_fwd(b["a1"], blk.inter.kernel, blk.inter.bias, b["f"], C, 2 * C, gelu=True, pre=b["pre_f"])
                self._dense_fwd(b["f"], blk.out.kernel, blk.out.bias, b["o"], 2 * C, C)
                self._ln_fwd(b["o"], b["a1"], C, blk.out_ln, drop(hd, 12 + 4 * i), b["y"], b["st2"])
            x, cin = b["y"], C
        if not self.blk:
            main.wait_event(ev_pack)
        if not (self.fused_tail and self.blk):
            self._dense_fwd(x, m.transform.kernel, m.transform.bias, self.so, cin, C, gelu=True, pre=self.pre_t)   # (no block: cin = 3C)
            self._ln_fwd(self.so, None, 0, m.transform_ln, ops.NO_DROP, self.hrows, self.st3, gpos=self.mpos)
        # rows whose label is 0 have weight 0 (EasyDGL.py:180): score only the weighted ones
        if not (self.fused_tail and self.blk):   # the fused tail writes its head rows compacted (row map = inv)
            check(lib.edgl_compact_gather(_ptr(self.hrows), _ptr(self.labels), _ptr(self.perm), R, C, _ptr(self.hrows_c),
                                          _ptr(self.labels_c), code, st), "edgl_compact_gather")
        lab = self.labels_c
        aux = _ptr(self.loss_aux) if m.l2_reg != 0.0 else None
        tpp = _ptr(self.loss_tpp) if (m.ct_reg != 0.0 and self.blk) else None
        if self.flash_ce:
            # the forward pass writes the loss coefficients itself (the row count of the loss is the compaction's): the backward
            # starts behind it, and the loss kernel — a one-workgroup reduction over the rows — leaves the critical path
            wtot = self.counts.data_ptr() if self._dp else None
            # (... and d_rows: log-sum-exp, label logits, coefficients and the row gradients leave ONE finishing launch)
            check(lib.edgl_score_flash_fwd_rows_wp(_ptr(self.hrows_c), _ptr(tab_c), _ptr(m.output_bias), _ptr(lab), R, C, I,
                                                   _ptr(self.nvalid), wtot, None, _ptr(self.lse), _ptr(self.lab_logit), _ptr(self.coef),
                                                   _ptr(self.d_rows), _ptr(self.ce_part), _ptr(self.ws_flash), code, st),
                  "edgl_score_flash_fwd_rows")
            # (launched on the side stream at the join the backward has anyway: an event record of its own costs the main
            # stream as much as the kernel).  With the row finish's per-workgroup sums the loss launch adds a few thousand numbers
            # and reads nothing of the batch; other widths: the kernel that sweeps lse / label logits / labels
            if self.ce_part is not None:
                self._pending_loss = lambda s, cp=self.ce_part: check(lib.edgl_ce_loss_parts(_ptr(cp), self.ce_nparts, _ptr(self.loss), aux,
                                                                                             tpp, wtot, s), "edgl_ce_loss_parts")
            else:
                self._pending_loss = lambda s: check(lib.edgl_ce_loss_fwd_add_w(_ptr(self.lse), _ptr(self.lab_logit), _ptr(lab), R,
                                                                                _ptr(self.loss), None, aux, tpp, wtot, s),
                                                     "edgl_ce_loss_fwd_add")
        else:
            check(lib.edgl_score_lse_fwd(_ptr(self.hrows_c), _ptr(tab_c), _ptr(m.output_bias), _ptr(lab), R, C, I, 0, I,
                                         _ptr(self.nvalid), _ptr(self.lse), _ptr(self.lab_logit), None, _ptr(self.ws), code, st),
                  "edgl_score_lse_fwd")
            check(lib.edgl_ce_loss_fwd_add_w(_ptr(self.lse), _ptr(self.lab_logit), _ptr(lab), R, _ptr(self.loss), _ptr(self.coef),
                                             aux, tpp, self.counts.data_ptr() if self._dp else None, st), "edgl_ce_loss_fwd_add")
            if self.accumulate_loss:     # (two-pass form: the loss is written inline, on the main stream)
                self.loss_sum.add_(self.loss)
        # ================= backward =================
        self._ws_i = 0
        check(lib.edgl_reduce_defer(1, st), "edgl_reduce_defer")
        try:
            self._issue_backward(st, drop, tab, tab_c, lab)
        except BaseException:
            # never leave the thread in deferred mode: later ops would queue reductions that nobody flushes
            lib.edgl_reduce_defer(-1, st)
            lib.edgl_gemm_dw_defer(-1, st)
            self._pending_loss = None
            self._pending_label = None
            raise
        if self._dw_forked:     # the weight-gradient slabs written on the second side stream
            torch.cuda.current_stream().wait_stream(self.side2)
            self._dw_forked = False
        check(lib.edgl_reduce_defer(0, st), "edgl_reduce_defer")   # runs the remaining queued reductions in one launch
        if self._pending_loss is not None:   # (no block: no side-stream join in the backward)
            self._pending_loss(st)
            self._pending_loss = None
            if self.accumulate_loss:         # (the loss launch of a model without blocks runs on the main stream)
                self.loss_sum.add_(self.loss)
        if self._lazy_loss and not self._side_has_grads:
            self._loss_unjoined = True       # (step() joins behind the optimizer, or the caller does: join_loss())
        else:
            torch.cuda.current_stream().wait_stream(self.side)
            self._loss_unjoined = False

    def join_loss(self) -> None:
        """Orders the current stream behind the kernels that write `self.loss` (launching them first if the last step left them
        for the next one) — needed before the loss is read on the current stream when step() ran with `sync_loss = False`."""
        if self._deferred_loss is not None:
            self.side.wait_stream(torch.cuda.current_stream())
            self._deferred_loss(self.side.cuda_stream)
            self._deferred_loss = None
            self._loss_unjoined = True
        if getattr(self, "_loss_unjoined", False):
            torch.cuda.current_stream().wait_stream(self.side)
            self._loss_unjoined = False

    def _issue_backward(self, st, drop, tab, tab_c, lab):
        m = self.m
        B, T, C, H, E, M, I, R = self.B, self.T, self.C, self.H, self.E, self.M, self.I, self.R
        code = self.code
        hd, ad = m.hidden_dropout_rate, m.attention_probs_dropout_rate
        if self.flash_ce:
            # d_rows: written by the forward call.  The one-hot term of the table / bias gradient (a scatter of the weighted rows:
            # f32 atomics that commute with the embedding scatter's) is deferred to the side stream at the end of the backward
            defer = 1 if self.blk else 0
            if self.score_atomic:
                defer |= 2      # d_table / d_bias zero-filled under the encoder (below): the row chunks add up in them, no slab reduction
            if self._fold:
                defer |= 4      # the slabs stay in the flash workspace: the optimizer launch sums them (no slab_reduce launch)
                o_t, o_b, ns = self._slab_info
                self._slabs = (self.ws_flash.data_ptr() + 4 * o_t, self.ws_flash.data_ptr() + 4 * o_b, ns)
            check(lib.edgl_score_flash_bwd_ex(_ptr(self.hrows_c), _ptr(tab_c), _ptr(m.output_bias), _ptr(lab), _ptr(self.lse),
                                              _ptr(self.coef), None, R, C, I, 0, I, _ptr(self.nvalid), None,
                                              _ptr(tab.grad), _ptr(m.output_bias.grad), _ptr(self.ws_flash), defer, code, st),
                  "edgl_score_flash_bwd")
            # the one-hot term as extra blocks of the embedding scatter's launch at the end of the backward (no launch, no fork)
            self._label_fused = bool((defer & 1) and self.code == _lib.BF16 and C == 128 and lib.edgl_encode_bwd_label_fused(C, code)
                                     and os.environ.get("EDGL_LABEL_FUSED", "1") != "0"
                                     and os.environ.get("EDGL_SCORE_STRIP", "1") != "0")    # (only the strip passes leave the term out)
            if (defer & 1) and not self._label_fused:
                self._pending_label = lambda s: check(lib.edgl_score_flash_label_term(
                    _ptr(self.hrows_c), _ptr(lab), _ptr(self.coef), None, R, C, I, 0, I, _ptr(self.nvalid), _ptr(tab.grad),
                    _ptr(m.output_bias.grad), code, s), "edgl_score_flash_label_term")
                # ... started right here on the side stream, beside the block-tail backward (measured on one box, 3 x 300 steps each:
                # 0.8895 ms against 0.8946 with the launch at the end of the backward, where it sat in front of the slab reductions
                # of the final join; the tail kernel itself takes 88 instead of 82 us next to it).  EDGL_LABEL_EARLY=0: at the end.
                # Default (2): behind the tail backward, beside the BiMAU sweeps (2 waves per SIMD at 256 registers, no LDS
                # pressure from the scatter): 0.868 ms against 0.876 (1) and 0.879 (0), 3 x 300 steps each on one box.
                if os.environ.get("EDGL_LABEL_EARLY", "2") == "1":
                    self.side.wait_stream(torch.cuda.current_stream())
                    self._pending_label(self.side.cuda_stream)
                    self._pending_label = None
        else:
            check(lib.edgl_score_ce_bwd(_ptr(self.hrows_c), _ptr(tab_c), _ptr(m.output_bias), _ptr(lab), _ptr(self.lse),
                                        _ptr(self.coef), None, R, C, I, 0, I, _ptr(self.nvalid), _ptr(self.d_rows), _ptr(tab.grad),
                                        _ptr(m.output_bias.grad), _ptr(self.ws), code, st), "edgl_score_ce_bwd")
        y_last, c_last = (self.blk[-1]["y"], C) if self.blk else (self.x0, 3 * C)    # (no block: the head reads the 3C-wide encoder output)
        if not (self.fused_tail and self.blk):
            # head: LN (gathered rows) -> gelu' -> dense
            # the GELU' of the head transform rides in the LayerNorm backward (G1 = gradient w.r.t. the dense pre-activation)
            self._ln_bwd(self.so, None, 0, m.transform_ln, self.st3, self.d_rows, ops.NO_DROP, self.G1, None, gpos=self.mpos,
                         rowmap=self.inv, act_pre=self.pre_t)
            self._dense_dw(y_last, self.G1, m.transform.kernel, m.transform.bias, c_last, C)
            self._dense_dx(self.G1, m.transform.kernel, self.G2 if self.blk else self.G3c, c_last, C)
        dY = self.G2 if self.blk else self.G3c
        for i in reversed(range(len(self.blk))):
            blk, b = m.layers[i], self.blk[i]
            x_in, cin = (self.x0, 3 * C) if i == 0 else (self.blk[i - 1]["y"], C)
            dh2, dh1 = drop(hd, 12 + 4 * i), drop(hd, 11 + 4 * i)
            if self.fused_tail:
                if self._dw_forked:     # the previous block's products on the side stream read what this block's kernels rewrite
                    torch.cuda.current_stream().wait_stream(self.side2)
                    self._dw_forked = False
                # the five weight-gradient products of the block run as one grouped launch after the BiMAU backward
                check(lib.edgl_gemm_dw_defer(1, st), "edgl_gemm_dw_defer")
                # one launch: LN3' -> GELU' -> dX(Wt) -> LN2' -> dX(Wout) * GELU' -> dX(Wi) -> LN1' -> dX(Wo)  (csrc/k_tail.hip)
                last = i == len(self.blk) - 1
                tl = m.transform_ln
                check(lib.edgl_tail_bwd_ct(x_in.data_ptr(), cin, _ptr(b["ao"]), _ptr(b["a1"]), _ptr(b["pre_f"]), _ptr(b["o"]),
                                        _ptr(self.pre_t), _ptr(self.so), _ptr(b["st1"]), _ptr(b["st2"]), _ptr(self.st3),
                                        _ptr(m.compute(blk.att_out.kernel)), _ptr(m.compute(blk.inter.kernel)),
                                        _ptr(m.compute(blk.out.kernel)), _ptr(m.compute(m.transform.kernel)),
                                        _ptr(blk.att_ln.gamma), _ptr(blk.out_ln.gamma), _ptr(tl.gamma), B, T, C, float(dh1.rate),
                                        dh1.ptr(), 11 + 4 * i, 12 + 4 * i, int(last), _ptr(self.d_rows), _ptr(self.mpos), M,
                                        _ptr(self.inv), None if last else _ptr(dY), _ptr(self.d_pre_t), _ptr(self.d_o),
                                        _ptr(self.d_pre_f), _ptr(self.d_ao), _ptr(self.G1), _ptr(self.G2),
                                        _ptr(blk.att_ln.gamma.grad), _ptr(blk.att_ln.beta.grad), _ptr(blk.out_ln.gamma.grad),
                                        _ptr(blk.out_ln.beta.grad), _ptr(tl.gamma.grad), _ptr(tl.beta.grad),
                                        _ptr(self._ws(lib.edgl_tail_bwd_workspace(B, C))), self.pad[0], self.pad[1], code, st), "edgl_tail_bwd")
                if self._pending_label is not None and os.environ.get("EDGL_LABEL_EARLY", "2") == "2":   # beside the BiMAU sweeps
                    self.side.wait_stream(torch.cuda.current_stream())
                    self._pending_label(self.side.cuda_stream)
                    self._pending_label = None
                if last:
                    self._dense_dw(y_last, self.d_pre_t, m.transform.kernel, m.transform.bias, C, C)
                self._dense_dw(b["f"], self.d_o, blk.out.kernel, blk.out.bias, 2 * C, C)
                self._dense_dw(b["a1"], self.d_pre_f, blk.inter.kernel, blk.inter.bias, C, 2 * C)
                self._dense_dw(b["att"], self.d_ao, blk.att_out.kernel, blk.att_out.bias, C, C)
                if self.dw_overlap & 1:
                    # The tail's weight-gradient products depend on the tail backward only: their grouped launch goes to the second
                    # side stream, UNDER the attention backward (VALU / transcendental bound, matrix pipe > 80 % idle) instead of
                    # behind it on the main stream.  Their split slabs are reduced by the step's one reduction launch (the deferred
                    # reduction queue is not bound to a stream); the main stream joins in front of that launch.
                    self.side2.wait_stream(torch.cuda.current_stream())
                    check(lib.edgl_gemm_dw_defer(0, self.side2.cuda_stream), "edgl_gemm_dw_defer")
                    check(lib.edgl_gemm_dw_defer(1, st), "edgl_gemm_dw_defer")
                    self._dw_forked = True
            else:
                # y = LN(drop(o) + a1)
                self._ln_bwd(b["o"], b["a1"], C, blk.out_ln, b["st2"], dY, dh2, self.G3, self.G4 if dh2.active else None)
                d_o = self.G4 if dh2.active else self.G3
                self._dense_dw(b["f"], d_o, blk.out.kernel, blk.out.bias, 2 * C, C)
                # d_pre_f = (d_o . Wout^T) * gelu'(pre_f)   — GELU' fused into the GEMM epilogue
                self._dense_dx(d_o, blk.out.kernel, self.G2c, 2 * C, C, flags=EPI_MUL_DGELU, aux=b["pre_f"])
                self._dense_dw(b["a1"], self.G2c, blk.inter.kernel, blk.inter.bias, C, 2 * C)
                # d_a1 = dsum (in G3) + d_pre_f . Wi^T        — accumulated by the GEMM epilogue
                self._dense_dx(self.G2c, blk.inter.kernel, self.G3, C, 2 * C, flags=EPI_ACCUM)
                # a1 = LN(drop(ao) + x_in[:, :, :C])
                self._ln_bwd(b["ao"], x_in, cin, blk.att_ln, b["st1"], self.G3, dh1, self.G1, self.G4 if dh1.active else None)
                d_ao = self.G4 if dh1.active else self.G1
                self._dense_dw(b["att"], d_ao, blk.att_out.kernel, blk.att_out.bias, C, C)
                self._dense_dx(d_ao, blk.att_out.kernel, self.G2, C, C)          # G2 = d_att
            att = blk.attention
            da = drop(ad, 10 + 4 * i)
            if self.mgroups:
                self._attention_bwd_groups(att, b, da, st)
            else:
                tp = self.fused_tpp
                check(lib.edgl_bimau_bwd_ord(_ptr(b["qkvt"]), _ptr(self.ids), _ptr(self.spans), _ptr(self.marks), _ptr(b["pack"]),
                                             _ptr(self.G2), None if tp else (_ptr(b["dlam"]) if m.ct_reg != 0.0 else None),
                                             _ptr(self.tpp_desc) if tp else None, M, (_ptr(b["tpp"]) if self._dp else None) if tp else None,
                                             float(m.ct_reg / H), _ptr(b["tpp_part"]) if tp else None,
                                             _ptr(b["lam"]), _ptr(b["saved"]), B, T, C, H, E,
                                             float(da.rate), da.ptr(), da.stream_id, _ptr(b["dbits"]), self.qk_scale, _ptr(self.G4c),
                                             _ptr(att.st_kernel.grad), _ptr(att.st_bias.grad), _ptr(att.weight.grad),
                                             _ptr(att.scaling.grad),
                                             _ptr(self._ws(lib.edgl_bimau_bwd_workspace(B, T, C, H, E, code), torch.uint8)),
                                             _ptr(self.job_order), self.mau_flags, code, st), "edgl_bimau_bwd_ord")
            self._dense_dw(x_in, self.G4c, att.dense_kernel, att.dense_bias, cin, 4 * C)
            if self.fused_tail:
                if self.dw_overlap & 2:
                    # ... and the QKVT product beside the dX GEMM / embedding backward chain that does not need it
                    self.side2.wait_stream(torch.cuda.current_stream())
                    check(lib.edgl_gemm_dw_defer(0, self.side2.cuda_stream), "edgl_gemm_dw_defer")
                    self._dw_forked = True
                else:
                    check(lib.edgl_gemm_dw_defer(0, st), "edgl_gemm_dw_defer")
            d_in = self.G3c if i == 0 else self.G3
            self._dense_dx(self.G4c, att.dense_kernel, d_in, cin, 4 * C)
            if i == 0:
                # every slab reduction queued so far (weight-gradient GEMMs, BiMAU / LayerNorm partials) runs on the side
                # stream under the embedding backward, whose atomics leave the CUs mostly idle
                # (bound of this fork, measured with the loss kernels left out: 6-8 us of the step — the event record behind the dX GEMM)
                pending, self._pending_loss = self._pending_loss, None

                def loss_launches(s, pending=pending, parts=[bj["tpp_part"] for bj in self.blk], desc=self.tpp_desc):
                    if self.fused_tpp:    # the regulariser from sweep 1's partial sums, block by block
                        for j, bj in enumerate(self.blk):
                            # (the count sweep 1 used rides behind the sums: nothing of the batch is read here)
                            check(lib.edgl_tpp_finish_parts_n(_ptr(parts[j]), B * H, float(m.ct_reg / H), H, _ptr(bj["tpp"]),
                                                              _ptr(self.loss_tpp), 1 if j > 0 else 0, s), "edgl_tpp_finish_parts")
                    if pending is not None:
                        pending(s)
                        if self.accumulate_loss:     # running sum of the step losses where they are produced (train.py reads it at its logging points)
                            with torch.cuda.stream(self.side):
                                self.loss_sum.add_(self.loss)

                self._side_has_grads = self._pending_label is not None      # (the one-hot term's atomics: Adam must wait for them)
                if self._lazy_loss and not self.sync_loss and not self._side_has_grads and self.ce_part is not None and \
                        (self.fused_tpp or m.ct_reg == 0.0) and os.environ.get("EDGL_DEFER_LOSS", "1") != "0" and \
                        os.environ.get("EDGL_ENGINE_LEGACY_FORK", "0") != "1":
                    self._deferred_loss = loss_launches      # no fork here: _issue() of the next step, or join_loss()
                else:
                    self.side.wait_stream(torch.cuda.current_stream())
                    loss_launches(self.side.cuda_stream)
                if self._pending_label is not None:
                    self._pending_label(self.side.cuda_stream)
                    self._pending_label = None
                if not self._lazy_loss or self._side_has_grads:
                    if self._dw_forked:
                        self.side.wait_stream(self.side2)
                    check(lib.edgl_reduce_flush(self.side.cuda_stream), "edgl_reduce_flush")
            # both residual branches feed the first C channels of the block input (temporal.py:447, EasyDGL.py:116)
            if i > 0:
                check(lib.edgl_add_cols(_ptr(d_in), cin, _ptr(self.G1), _ptr(self.G2), C, self.rows, C, code, st), "edgl_add_cols")
                # next (earlier) block consumes d_in as its dY; keep it out of the scratch set it will overwrite
                self.G2.copy_(self.G3)
                dY = self.G2
            else:
                dY = d_in   # first block: the embedding backward adds the two branches itself (one pass less over dX0)
        d0 = drop(hd, 1)
        add1, add2 = (self.G1, self.G2) if self.blk else (None, None)
        if getattr(self, "_label_fused", False):
            check(lib.edgl_encode_bwd_add_label(_ptr(self.ids), _ptr(self.marks), _ptr(dY), _ptr(add1), _ptr(add2), B, T, C, E, I,
                                                float(d0.rate), d0.ptr(), d0.stream_id, _ptr(tab.grad),
                                                _ptr(m.pcoding.pembs.lookup_table.grad), _ptr(m.mark_embs.lookup_table.grad),
                                                _ptr(self._ws(lib.edgl_encode_bwd_workspace(B, T, C))), self.c_true,
                                                _ptr(self.hrows_c), _ptr(lab), _ptr(self.coef), _ptr(self.nvalid), R,
                                                _ptr(m.output_bias.grad), code, st), "edgl_encode_bwd_add_label")
        else:
            check(lib.edgl_encode_bwd_add_ct(_ptr(self.ids), _ptr(self.marks), _ptr(dY), _ptr(add1), _ptr(add2), B, T, C, E, I,
                                             float(d0.rate), d0.ptr(), d0.stream_id, _ptr(tab.grad), _ptr(m.pcoding.pembs.lookup_table.grad),
                                             _ptr(m.mark_embs.lookup_table.grad), _ptr(self._ws(lib.edgl_encode_bwd_workspace(B, T, C))),
                                             self.c_true, code, st), "edgl_encode_bwd_add")

    # ---- more than 16 mark types: the attention of a block as mark groups (see __init__) -----------------------------------
    def _attention_fwd_groups(self, b, x, cin, da, st):
        m = self.m
        B, T, C, H, E, code = self.B, self.T, self.C, self.H, self.E, self.code
        for g, ((e0, e1), gb) in enumerate(zip(self.mgroups, b["grp"])):
            gb["marks"].copy_(self.marks[:, :, e0:e1])
            first = g == 0
            check(lib.edgl_bimau_fwd_db(_ptr(b["qkvt"]), x.data_ptr() if first else self.zero_resid.data_ptr(), cin if first else C,
                                        _ptr(self.ids), _ptr(self.spans), _ptr(gb["marks"]), _ptr(gb["pack"]), B, T, C, H, e1 - e0,
                                        float(da.rate), da.ptr(), da.stream_id, _ptr(b["dbits"]), self.qk_scale, _ptr(b["att"] if first else gb["out"]),
                                        _ptr(gb["lam"]), _ptr(gb["saved"]), None, 0 if first else ops.MAU_DIAG_ZERO, code, st),
                  "edgl_bimau_fwd_db")
            if not first:
                check(lib.edgl_add(_ptr(b["att"]), _ptr(gb["out"]), _ptr(b["att"]), b["att"].numel(), code, st), "edgl_add")
        torch.cat([gb["lam"] for gb in b["grp"]], dim=-1, out=b["lam"])
        if m.ct_reg != 0.0:
            b["dlam"].zero_()       # edgl_tpp_fwd_bwd_rows writes the rows of the masked positions only

    def _attention_bwd_groups(self, att, b, da, st):
        m = self.m
        B, T, C, H, code = self.B, self.T, self.C, self.H, self.code
        dh = C // H
        # the group launches reduce their weight-gradient partials at once (their dW1 block is copied into the strided column
        # block of st_kernel.grad right behind them): leave the deferred-reduction mode around them
        check(lib.edgl_reduce_defer(0, st), "edgl_reduce_defer")
        for g, ((e0, e1), gb) in enumerate(zip(self.mgroups, b["grp"])):
            first = g == 0
            if m.ct_reg != 0.0:
                gb["dlam"].copy_(b["dlam"][:, :, e0:e1])
            check(lib.edgl_bimau_bwd_db(_ptr(b["qkvt"]), _ptr(self.ids), _ptr(self.spans), _ptr(gb["marks"]), _ptr(gb["pack"]),
                                     _ptr(self.G2), _ptr(gb["dlam"]) if m.ct_reg != 0.0 else None, _ptr(gb["lam"]), _ptr(gb["saved"]),
                                     B, T, C, H, e1 - e0, float(da.rate), da.ptr(), da.stream_id, _ptr(b["dbits"]), self.qk_scale,
                                     _ptr(self.G4c if first else gb["dqkvt"]), _ptr(gb["dW1"]), _ptr(att.st_bias.grad[e0 * dh:e1 * dh]),
                                     _ptr(att.weight.grad[e0:e1]), _ptr(att.scaling.grad[e0:e1]),
                                     _ptr(self._ws(lib.edgl_bimau_bwd_workspace(B, T, C, H, e1 - e0, code), torch.uint8)),
                                     0 if first else ops.MAU_DIAG_ZERO, code, st), "edgl_bimau_bwd")
            att.st_kernel.grad[:, e0 * dh:e1 * dh].copy_(gb["dW1"])
            if not first:
                check(lib.edgl_add(_ptr(self.G4c), _ptr(gb["dqkvt"]), _ptr(self.G4c), self.G4c.numel(), code, st), "edgl_add")
        check(lib.edgl_reduce_defer(1, st), "edgl_reduce_defer")

    def _advance_state(self, st):
        m = self.m
        check(lib.edgl_step_begin(_ptr(m._rng_state), _ptr(m._adam_state), float(m.learning_rate), 0.9, 0.999, st),
              "edgl_step_begin")

    def _optimizer(self):
        m = self.m
        m.mask_padded_grads()     # (channel-padded models: the one gradient that is not zero on a padded entry by itself)
        if os.environ.get("EDGL_ENGINE_LEGACY_FORK", "0") == "1":     # A/B switch: no look-ahead of the step counters
            seg = self.l2_seg if m.l2_reg != 0.0 else None
            check(lib.edgl_adam_apply(_ptr(m._arena), _ptr(m._grad_arena), _ptr(m._adam_m), _ptr(m._adam_v), m._arena.numel(), 0.9,
                                      0.999, 1e-8, _ptr(m._adam_state), float(m.l2_reg), _ptr(seg),
                                      0 if seg is None else seg.numel() // 2, _ptr(m._shadow), _stream()), "edgl_adam_apply")
            return
        seg = self.l2_seg if m.l2_reg != 0.0 else None
        if os.environ.get("EDGL_ADAM_NEXT", "0") == "1":
            # optimizer + the counters of the next step in ONE launch (the last workgroup to finish advances them)
            l2p = None
            if self.l2_parts is not None and seg is not None:
                self._l2p_cur ^= 1       # the copy the NEXT step reads
                l2p = self.l2_parts[self._l2p_cur]
                self._l2p_ready = True
                m._l2_parts_owner = id(self)
            if getattr(self, "_adam_ticket", None) is None:
                self._adam_ticket = torch.zeros(int(lib.edgl_adam_next_tickets(m._arena.numel())), device=m._arena.device, dtype=torch.int32)
            check(lib.edgl_adam_apply_l2p_next(_ptr(m._arena), _ptr(m._grad_arena), _ptr(m._adam_m), _ptr(m._adam_v), m._arena.numel(),
                                               0.9, 0.999, 1e-8, _ptr(m._adam_state), float(m.l2_reg), _ptr(seg),
                                               0 if seg is None else seg.numel() // 2, _ptr(m._shadow), _ptr(l2p), _ptr(m._rng_state),
                                               float(m.learning_rate), _ptr(self._adam_ticket), _stream()), "edgl_adam_apply_l2p_next")
            m._state_ahead = True
            return
        if self.adam_ex and not getattr(m, "_state_pinned", False):
            import ctypes
            l2p = None
            if self.l2_parts is not None and seg is not None:
                self._l2p_cur ^= 1       # the copy the NEXT step reads
                l2p = self.l2_parts[self._l2p_cur]
            tab, ob = m.item_embs.lookup_table, m.output_bias
            a0 = m._arena.data_ptr()
            sl = self._slabs
            self._slabs = None
            lo_a, lo_b = (tab.data_ptr() - a0) // 4, (ob.data_ptr() - a0) // 4
            check(lib.edgl_adam_apply_ex(_ptr(m._arena), _ptr(m._grad_arena), _ptr(m._adam_m), _ptr(m._adam_v), m._arena.numel(), 0.9, 0.999,
                                         1e-8, _ptr(m._adam_state), float(m.l2_reg), _ptr(seg), 0 if seg is None else seg.numel() // 2,
                                         _ptr(m._shadow), _ptr(l2p),
                                         ctypes.c_void_p(sl[0]) if sl else None, tab.numel(), lo_a, lo_a + tab.numel(), self.C,
                                         ctypes.c_void_p(sl[1]) if sl else None, ob.numel(), lo_b, lo_b + ob.numel(), sl[2] if sl else 0,
                                         _ptr(m._rng_state), _ptr(self._adam_alt), _ptr(self._rng_alt), float(m.learning_rate), _stream()),
                  "edgl_adam_apply_ex")
            if l2p is not None:
                self._l2p_ready = True
                m._l2_parts_owner = id(self)
            else:
                m._l2_parts_owner = None
            m._table_grad_zero = id(self) if sl else None      # (the launch zeroed the two slab ranges of the gradient behind its reads)
            # the counters of the next step were written to the other pair of buffers: swap them in (settle_state undoes the
            # look-ahead on whichever pair is current)
            m._rng_state, self._rng_alt = self._rng_alt, m._rng_state
            m._adam_state, self._adam_alt = self._adam_alt, m._adam_state
            m._state_ahead = True
            return
        if self._slabs is not None:
            raise _lib.EdglError("TrainEngine: the scoring gradient's slabs were left for edgl_adam_apply_ex, which is not in use")
        if self.l2_parts is not None and seg is not None:
            self._l2p_cur ^= 1       # the copy the NEXT step reads
            check(lib.edgl_adam_apply_l2p(_ptr(m._arena), _ptr(m._grad_arena), _ptr(m._adam_m), _ptr(m._adam_v), m._arena.numel(), 0.9,
                                          0.999, 1e-8, _ptr(m._adam_state), float(m.l2_reg), _ptr(seg), seg.numel() // 2, _ptr(m._shadow),
                                          _ptr(self.l2_parts[self._l2p_cur]), _stream()), "edgl_adam_apply_l2p")
            self._l2p_ready = True
            m._l2_parts_owner = id(self)
        else:
            check(lib.edgl_adam_apply(_ptr(m._arena), _ptr(m._grad_arena), _ptr(m._adam_m), _ptr(m._adam_v), m._arena.numel(), 0.9,
                                      0.999, 1e-8, _ptr(m._adam_state), float(m.l2_reg), _ptr(seg),
                                      0 if seg is None else seg.numel() // 2, _ptr(m._shadow), _stream()), "edgl_adam_apply")
            m._l2_parts_owner = None
        # the counters of the next step (Sequential.settle_state undoes this for anyone else who reads them)
        self._advance_state(_stream())
        m._state_ahead = True

    # ---- public API --------------------------------------------------------------------------------------------------------
    def load_batch(self, features: Dict[str, torch.Tensor], labels: torch.Tensor) -> None:
        self.ids.copy_(features["seqs_i"]); self.ts.copy_(features["seqs_t"])
        self.mpos.copy_(features["masked_positions"]); self.labels.copy_(labels)

    def bind_batch(self, features: Dict[str, torch.Tensor], labels: torch.Tensor) -> None:
        """Point the eager launch sequence at device tensors that already hold a batch (no copies: a rotation of resident
        batches costs nothing).  Not for the HIP-graph path, whose captured launches keep the addresses of the static buffers."""
        if self.use_graph:
            raise _lib.EdglError("TrainEngine.bind_batch: the graph path reads the static buffers (use load_batch / step(features, labels))")
        new = (features["seqs_i"], features["seqs_t"], features["masked_positions"], labels)
        for t, ref, nm in zip(new, (self.ids, self.ts, self.mpos, self.labels), ("seqs_i", "seqs_t", "masked_positions", "labels")):
            if t.shape != ref.shape or t.dtype != ref.dtype or t.device != ref.device or not t.is_contiguous():
                raise _lib.EdglError(f"TrainEngine.bind_batch: {nm} must be a contiguous {tuple(ref.shape)} {ref.dtype} tensor on {ref.device}")
        self.ids, self.ts, self.mpos, self.labels = new

    def _global_counts(self) -> None:
        """Data parallel: the two normalisers of the loss that are sums over the BATCH — the number of weighted rows
        (EasyDGL.py:183-185) and the number of next-event marks of the TPP term (temporal.py:331-333) — are all-reduced before the
        step, so that every rank differentiates its share of the GLOBAL-batch loss and the ranks' gradients simply add up.
        Labels only: issued ahead of the step's first kernel (one 8-byte all-reduce, hidden under the encoder)."""
        import torch.distributed as dist
        m = self.m
        if m.ct_reg != 0.0 and self.blk and self.B * self.M <= 65536:
            # both sums in one launch (as torch ops: compare, sum, cast, two slice copies and the normaliser launch — six launches in
            # front of the encoder of every data-parallel step)
            check(lib.edgl_dp_counts(_ptr(self.labels), _ptr(m.mark_lookup_table), self.B, self.M, self.E, _ptr(self.counts), _stream()),
                  "edgl_dp_counts")
        else:
            self.counts[0:1] = (self.labels != 0).sum().to(torch.int32)
            if m.ct_reg != 0.0 and self.blk:
                tpp0 = self.blk[0]["tpp"]
                check(lib.edgl_tpp_norm(_ptr(self.labels), _ptr(m.mark_lookup_table), self.B, self.M, self.E, _ptr(tpp0), _stream()),
                      "edgl_tpp_norm")
                self.counts[1:2] = tpp0.view(torch.int32)[4:5]
        dist.all_reduce(self.counts, op=dist.ReduceOp.SUM, group=self.group)
        if m.ct_reg != 0.0:
            for b in self.blk:
                b["tpp"].view(torch.int32)[4:5] = self.counts[1:2]

    def _dp_allreduce(self) -> torch.Tensor:
        """Data parallel: ONE SUM all-reduce of the gradient arena per step.  The loss a rank's kernels produce mixes per-rank and
        global terms — its share of the cross-entropy and of the TPP term (both already divided by the GLOBAL normalisers) plus the
        full L2 term, which every rank holds — so neither the sum nor the mean of the ranks' values is the loss of the global batch.
        The share (loss - L2) rides in the arena's first trailing slot through the same collective; what comes back, plus L2, is
        the loss of the concatenated batch, identical on every rank and equal to the single-process value
        (tests/test_gpu_distributed.py).  Returns that scalar (also kept as `self.loss_global`)."""
        from . import parallel
        m = self.m
        comm = m._grad_comm
        n = m._grad_arena.numel()
        if m.l2_reg != 0.0:
            torch.sub(self.loss, self.loss_aux, out=comm[n:n + 1])
        else:
            comm[n:n + 1].copy_(self.loss)
        parallel.allreduce_sum_(comm, self.group)
        self.loss_global = comm[n:n + 1] + self.loss_aux if m.l2_reg != 0.0 else comm[n:n + 1].clone()
        return self.loss_global

    def step(self, features=None, labels=None) -> torch.Tensor:
        """One optimizer step on the (optionally refreshed) static batch; returns the loss (device scalar).  Data parallel: the
        normalisers of the loss are global (_global_counts), the flat gradient arena is all-reduced with SUM — the result is the
        gradient of the loss over the concatenated batch, whatever the ranks' numbers of weighted rows — and the returned loss
        is the loss of that concatenated batch, the same number on every rank (_dp_allreduce); `self.loss` stays this rank's
        share + L2 as its kernels wrote it."""
        if features is not None:
            self.load_batch(features, labels)
        distributed = torch.distributed.is_available() and torch.distributed.is_initialized() and \
            torch.distributed.get_world_size(self.group) > 1
        self._dp = distributed
        if not self.use_graph:
            if distributed:
                self._global_counts()
            self._issue(lazy_loss=not distributed and os.environ.get("EDGL_LAZY_LOSS", "1") != "0",
                        fold_slabs=self.adam_ex and not distributed and not getattr(self.m, "_state_pinned", False))
            out = self.loss
            if distributed:
                evs = getattr(self, "_ar_events", None)      # bench.py: HIP events around the step's one collective (its EXPOSED time:
                if evs is not None:                         # nothing of the step runs beside it — the last gradients are final only now)
                    e0 = torch.cuda.Event(enable_timing=True); e0.record()
                out = self._dp_allreduce()
                if evs is not None:
                    e1 = torch.cuda.Event(enable_timing=True); e1.record()
                    evs.append((e0, e1))
            self._optimizer()
            if self.sync_loss:      # the returned loss is ordered on the current stream (False: the caller calls join_loss() / syncs)
                self.join_loss()
            return out
        if self.graph is None:
            # warm-up on a side stream (allocator / lazy module init), then capture
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                if distributed:
                    self._global_counts()
                self._issue()
                if not distributed:
                    self._optimizer()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            # the warm-up was a real step; undo nothing: training simply started one step earlier
            if distributed:
                self._dp_allreduce()
                self._optimizer()
                self._global_counts()      # (the capture below reads the counts of the current batch)
            # (the warm-up's optimizer left the counters of the next step in place: the captured sequence starts without the
            #  single-thread update and — with its own optimizer — ends with it)
            # (the A/B switch EDGL_ENGINE_LEGACY_FORK=1 has no look-ahead: its _issue() carries the update itself)
            assert getattr(self.m, "_state_ahead", False) or os.environ.get("EDGL_ENGINE_LEGACY_FORK", "0") == "1"
            self.m._state_pinned = True      # (the captured launches keep the counters' addresses: no engine swaps them from here on)
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self._issue()
                if not distributed:
                    self._optimizer()
            self.m._state_ahead = True      # nothing ran during the capture: still what the warm-up left
            self._distributed = distributed
            if distributed:                # the captured launches have not run: this call is the warm-up step only
                return self.loss_global
            return self.loss
        if self._distributed:
            self._global_counts()
        if not getattr(self.m, "_state_ahead", False):   # somebody settled the counters (a checkpoint, an autograd-path step)
            self._advance_state(_stream())
        self.graph.replay()
        self.m._state_ahead = not self._distributed       # the captured optimizer ends with the next step's counters
        if self._distributed:
            out = self._dp_allreduce()
            self._optimizer()
            return out
        return self.loss

"""Mirror of src/module/temporal.py: BiMAU (the Bi-level Modulating Attention Unit used by EasyDGL) and MAU (the causal
unit of CTSMA), both on the fused HIP attention kernels, and TfMultiHeadAttention (TGAT) on the generic masked attention
kernel with the time feature map."""
from __future__ import annotations

import torch
from torch import nn

from .. import ops
from .coding import glorot_uniform_


SUPPORTED_HEAD_DIMS = (16, 32, 64, 128)
MARK_GROUP = 16      # mark types one kernel launch takes (one MFMA K-block, csrc/bimau_common.h EP)
MAX_EVENTS = 256     # marks travel as uint8 one-hot columns; beyond 16 they run as groups (modulated_attention)


def _check_head_dim(num_units, num_heads, num_events, who):
    """The fused attention kernels tile the head dim in 16-wide MFMA blocks (csrc/k_bimau_*.hip): fail here, with the
    flag names, instead of at the first step with a low-level shape error."""
    if num_heads <= 0 or num_units % num_heads:
        raise ValueError(f"{who}: num_units={num_units} must be a multiple of num_heads={num_heads}")
    dh = num_units // num_heads
    if dh not in SUPPORTED_HEAD_DIMS:
        raise ValueError(f"{who}: head dim num_units/num_heads = {dh} unsupported; the HIP kernels take {SUPPORTED_HEAD_DIMS} "
                         f"(e.g. --num_units=512 --num_heads=8 as in runme.sh:15-23)")
    if not (1 <= num_events <= MAX_EVENTS):
        raise ValueError(f"{who}: num_events={num_events} unsupported (1..{MAX_EVENTS} mark types)")


def key_ids_from_masks(masks: torch.Tensor, batch: int, seqlen: int, num_heads: int) -> torch.Tensor:
    """The fused kernels take the key mask as a [B, T] int64 vector (a key is masked where it is 0: the item ids serve as is).
    The reference hands ``masks`` = tile(expand_dims(float(ids != 0), 1), [h, T, 1]) — [h*B, T, T] floats, every query row
    the same key vector (EasyDGL.py:94-95, used at temporal.py:425-426).  Accepted here and reduced on the device:
      * int64 [B, T]                      -> as is (ids or 0 / 1);
      * [h*B, T, T], [B, T, T], [B, 1, T] -> head 0, query row 0 (head-major stacking: rows 0..B-1 are head 0);
      * [B, T] of another dtype           -> != 0.
    Anything else raises (a silently mis-read mask is a wrong model, not an error message)."""
    if not torch.is_tensor(masks):
        raise TypeError("BiMAU / MAU: `masks` must be a tensor ([B,T] ids, or the reference's [h*B,T,T] key mask)")
    if masks.dim() == 2 and tuple(masks.shape) == (batch, seqlen):
        return masks.contiguous() if masks.dtype == torch.int64 else (masks != 0).to(torch.int64)
    if masks.dim() == 3 and masks.shape[2] == seqlen and masks.shape[1] in (1, seqlen) and masks.shape[0] in (batch, num_heads * batch):
        return (masks[:batch, 0, :] != 0).to(torch.int64).contiguous()
    raise ValueError(f"BiMAU / MAU: `masks` of shape {tuple(masks.shape)} / {masks.dtype}: expected int64 [B={batch}, T={seqlen}] "
                     f"(item ids or 0/1), or the reference's key mask [h*B, T, T] / [B, T, T] / [B, 1, T]")


def modulated_attention(qkvt, resid, st_kernel, st_bias, weight, scaling, masks, intervals, marks, num_heads, drop, flags=0, qk_scale=0.0):
    """The fused attention of BiMAU / MAU for any number of mark types.  Up to 16 marks: one launch (ops.BiMAUFn).  More: the
    modulation G[q,k] = sum_e marks[k,e] lambda[q,e] (temporal.py:309-313) is a sum over marks and the output (G * P) V is linear
    in G, and lambda_e only reads its own dh columns of the intensity MLP (temporal.py:291: split(Z, dh)), so the marks run as
    groups of <= 16 (edgl_bimau_mark_group) — each launch with its column block of ``st_kernel`` / ``st_bias``, its rows of ``weight`` / ``scaling`` and
    its mark columns, the SAME dropout stream (one mask for the whole of G) — and the outputs add up.  The first group carries
    the residual and, for BiMAU, the diagonal 1 (set_diag, :438-439); the later groups a zero residual and the diagonal 0."""
    E, dh = weight.shape
    group = ops.lib.edgl_bimau_mark_group(dh * num_heads, num_heads, ops._code(qkvt))   # 16, or what the intensity backward's LDS holds
    if group <= 0:
        raise ValueError(f"BiMAU / MAU: head dim {dh} / {qkvt.dtype} unsupported")
    if E <= group:
        return ops.BiMAUFn.apply(qkvt, resid, st_kernel, st_bias, weight, scaling, masks, intervals, marks, num_heads, drop, flags, qk_scale)
    out, lams = None, []
    zero_resid = torch.zeros_like(resid)
    for e0 in range(0, E, group):
        e1 = min(E, e0 + group)
        gflags = flags if e0 == 0 else (flags | (0 if flags & ops.MAU_NO_DIAG else ops.MAU_DIAG_ZERO))
        o, lam = ops.BiMAUFn.apply(qkvt, resid if e0 == 0 else zero_resid, st_kernel[:, e0 * dh:e1 * dh].contiguous(),
                                   st_bias[e0 * dh:e1 * dh].contiguous(), weight[e0:e1].contiguous(), scaling[e0:e1].contiguous(),
                                   masks, intervals, marks[:, :, e0:e1].contiguous(), num_heads, drop, gflags, qk_scale)
        out = o if out is None else out + o
        lams.append(lam)
    return out, torch.cat(lams, dim=-1)


class BiMAU(nn.Module):
    """temporal.py:393-452 + MAU.intensity (temporal.py:281-315).

    Parameters (TF names in brackets): ``dense_kernel`` [Cin, 4C] ~N(0, 0.02) and ``dense_bias`` [4C]
    (TMAU/dense); ``st_kernel`` [dh+1, dh*E] glorot and ``st_bias`` [dh*E]
    (TMAU/sequential_temporal_combined/dense); ``weight`` [E, dh] glorot; ``scaling`` [E] zeros.
    ``__call__(queries, keys, masks, intervals, marks, is_training)`` keeps the reference's signature:
    keys are ignored exactly as in the reference (BiMAU projects Q,K,V,T from ``queries``), ``masks`` is the
    [B,T] item-id tensor (a key is masked where id == 0) or the reference's own [h*B,T,T] float key mask
    (``key_ids_from_masks``)."""

    def __init__(self, num_units, num_heads, num_events, dropout_rate, scope="TMAU", in_units=None, gen=None):
        """Reference signature (temporal.py:401): ``BiMAU(num_units, num_heads, num_events, dropout_rate, scope)``.
        ``in_units`` (width of ``queries``; 3C for EasyDGL's first block) is what tf.layers.dense infers at graph build:
        give it to create the projection eagerly (the models do, so that it lands in the flat arena), or leave it None and
        the projection is created on the first call."""
        super().__init__()
        _check_head_dim(num_units, num_heads, num_events, "BiMAU")
        self.num_units, self.num_heads, self.num_events, self.dropout_rate = num_units, num_heads, num_events, dropout_rate
        self.scope = scope
        self._gen = gen
        dh = num_units // num_heads
        self.dense_kernel = None
        self.dense_bias = None
        if in_units is not None:
            self._make_projection(in_units)
        self.st_kernel = nn.Parameter(glorot_uniform_(torch.empty(dh + 1, dh * num_events), gen))
        self.st_bias = nn.Parameter(torch.zeros(dh * num_events))
        self.weight = nn.Parameter(glorot_uniform_(torch.empty(num_events, dh), gen))
        self.scaling = nn.Parameter(torch.zeros(num_events))
        self.compute = lambda p: p  # replaced by the owning model (bf16 shadow lookup)

    def _make_projection(self, in_units, device=None):
        k = torch.randn(in_units, 4 * self.num_units, generator=self._gen) * 0.02          # temporal.py:393,409
        self.dense_kernel = nn.Parameter(k if device is None else k.to(device))
        self.dense_bias = nn.Parameter(torch.zeros(4 * self.num_units, device=device))

    def forward(self, queries, keys, masks, intervals, marks, is_training, causality=None, *, drop: ops.Drop = None):
        """temporal.py:404: ``(queries, keys, masks, intervals, marks, is_training, causality=None)`` -> (outputs [B,T,C],
        mark_intensity [h*B,T,E]).  ``causality`` is ignored exactly as in the reference (BiMAU is bidirectional).
        ``drop`` (keyword only) names the dropout stream: the owning model passes its device RNG state; a unit used on its
        own with ``is_training`` and ``dropout_rate > 0`` draws from a module-local state (seeded once, advanced per call), so
        that the reference signature alone never silently trains without dropout."""
        if drop is None:
            if is_training and self.dropout_rate > 0:
                if getattr(self, "_own_rng", None) is None or self._own_rng.device != queries.device:
                    self._own_rng = ops.make_rng_state(queries.device, seed=0x42694d4155)
                ops.rng_advance(self._own_rng)
                # a SNAPSHOT of (seed, step): the backward re-derives the mask from the state it is handed, and a second call of
                # this unit before that backward (shared layer, gradient accumulation) must not move it
                drop = ops.Drop(self.dropout_rate, self._own_rng.clone(), 10)
            else:
                drop = ops.NO_DROP
        if self.dense_kernel is None:
            self._make_projection(queries.shape[-1], queries.device)
        C = self.num_units
        masks = key_ids_from_masks(masks, queries.shape[0], queries.shape[1], self.num_heads)
        qkvt = ops.LinearFn.apply(queries, self.dense_kernel, self.dense_bias, self.compute(self.dense_kernel), False)
        resid = queries[:, :, :C]
        return modulated_attention(qkvt, resid, self.st_kernel, self.st_bias, self.weight, self.scaling, masks, intervals,
                                   marks, self.num_heads, drop if is_training else ops.NO_DROP, 0, float(getattr(self, "qk_scale", 0.0)))


class MAU(nn.Module):
    """temporal.py:335-390 (``T.MAU``, ICML'21 CTSMA): four separate projections — Q from ``queries``, K, V, T_ from ``keys``
    (tf.layers.dense x4, glorot kernels; K|V|T_ are stored as one [Cin, 3C] matrix whose column blocks are the three TF
    variables) — then the same fused kernel as BiMAU with ``causality`` (future blinding, :370-375) and the modulation
    kept on the diagonal; the residual adds the queries' first C channels (:383)."""

    def __init__(self, num_units, num_heads, num_events, dropout_rate, scope="modulating_attention", in_units=None, gen=None):
        """Reference signature (temporal.py:274); ``in_units`` = width of queries / keys (default ``num_units``)."""
        super().__init__()
        _check_head_dim(num_units, num_heads, num_events, "MAU")
        self.num_units, self.num_heads, self.num_events, self.dropout_rate = num_units, num_heads, num_events, dropout_rate
        self.scope = scope
        in_units = num_units if in_units is None else in_units
        dh = num_units // num_heads
        self.q_kernel = nn.Parameter(glorot_uniform_(torch.empty(in_units, num_units), gen))        # dense
        self.q_bias = nn.Parameter(torch.zeros(num_units))
        kvt = torch.cat([glorot_uniform_(torch.empty(in_units, num_units), gen) for _ in range(3)], dim=1)
        self.kvt_kernel = nn.Parameter(kvt)                                                           # dense_1 | dense_2 | dense_3
        self.kvt_bias = nn.Parameter(torch.zeros(3 * num_units))
        self.st_kernel = nn.Parameter(glorot_uniform_(torch.empty(dh + 1, dh * num_events), gen))
        self.st_bias = nn.Parameter(torch.zeros(dh * num_events))
        self.weight = nn.Parameter(glorot_uniform_(torch.empty(num_events, dh), gen))
        self.scaling = nn.Parameter(torch.zeros(num_events))
        self.compute = lambda p: p

    def forward(self, queries, keys, masks, intervals, marks, is_training, causality=True, *, drop: ops.Drop = None):
        C = self.num_units
        if drop is None:
            if is_training and self.dropout_rate > 0:
                if getattr(self, "_own_rng", None) is None or self._own_rng.device != queries.device:
                    self._own_rng = ops.make_rng_state(queries.device, seed=0x4d4155)
                ops.rng_advance(self._own_rng)
                drop = ops.Drop(self.dropout_rate, self._own_rng.clone(), 10)   # snapshot: see BiMAU.forward
            else:
                drop = ops.NO_DROP
        masks = key_ids_from_masks(masks, queries.shape[0], queries.shape[1], self.num_heads)
        # column blocks Q | K | V | T_ of the kernel's operand, written in place by the two projections (no concatenation copy)
        qkvt = ops.DualLinearFn.apply(queries, keys, self.q_kernel, self.q_bias, self.compute(self.q_kernel), self.kvt_kernel,
                                      self.kvt_bias, self.compute(self.kvt_kernel))
        flags = ops.MAU_NO_DIAG | (ops.MAU_CAUSAL if causality else 0)
        return modulated_attention(qkvt, queries[:, :, :C], self.st_kernel, self.st_bias, self.weight, self.scaling, masks,
                                   intervals, marks, self.num_heads, drop if is_training else ops.NO_DROP, flags,
                                   float(getattr(self, "qk_scale", 0.0)))


class TfMultiHeadAttention(nn.Module):
    """temporal.py:108-184 (TGAT, ICLR'20).  ``q_kernel`` [C,C] (dense), ``kv_kernel`` [C,2C] (dense_1 | dense_2 as column
    blocks: K and V read the same input, one GEMM); the position table and the time-function parameters are the model's shared
    ``pcoding_K`` / ``tcoding_K`` (TGAT.py:29-30).  ``__call__(queries, keys, intervals, is_training, causality)`` as in the
    reference, except that ``intervals`` is the pair (ids [B,T], seqs_t [B,T+1]) the [B,T,T] interval tensor is a function of —
    it is never materialised — and the key mask is ``ids != 0`` (a key row of the reference is all-zero exactly there)."""

    def __init__(self, num_units, num_heads, dropout_rate, l2_reg, pcoding_K, tcoding_K, gen=None):
        super().__init__()
        self.num_units, self.num_heads, self.dropout_rate, self.l2_reg = num_units, num_heads, dropout_rate, l2_reg
        self.q_kernel = nn.Parameter(glorot_uniform_(torch.empty(num_units, num_units), gen))
        self.q_bias = nn.Parameter(torch.zeros(num_units))
        kv = torch.cat([glorot_uniform_(torch.empty(num_units, num_units), gen) for _ in range(2)], dim=1)
        self.kv_kernel = nn.Parameter(kv)
        self.kv_bias = nn.Parameter(torch.zeros(2 * num_units))
        object.__setattr__(self, "pcoding_K", pcoding_K)   # shared with the model: not registered twice
        object.__setattr__(self, "tcoding_K", tcoding_K)
        self.compute = lambda p: p
        self.time_scale = 1.0
        self.violations = None

    def forward(self, queries, keys, intervals, is_training, causality=True, drop: ops.Drop = ops.NO_DROP):
        if not causality:
            raise NotImplementedError("the HIP time-function attention implements the causal form TGAT uses (TGAT.py:66)")
        ids, ts = intervals
        q = ops.LinearFn.apply(queries, self.q_kernel, self.q_bias, self.compute(self.q_kernel), False)
        kv = ops.LinearFn.apply(keys, self.kv_kernel, self.kv_bias, self.compute(self.kv_kernel), False)
        return ops.TfAttnFn.apply(q, kv, queries, self.pcoding_K.pembs.lookup_table, self.tcoding_K.basis_freq,
                                  self.tcoding_K.phase, ids, ts, self.num_heads, self.time_scale,
                                  drop if is_training else ops.NO_DROP, self.violations, float(getattr(self, "qk_scale", 0.0)))


class TiMultiHeadAttention(nn.Module):
    """temporal.py:15-105 (TiSASRec, WSDM'20).  ``q_kernel`` [C,C], ``kv_kernel`` [C,2C] (dense_1 | dense_2); the position and
    interval tables are the model's shared ``pcoding_K/V`` and ``tcoding_K/V`` (TiSASREC.py:30-33).  ``intervals`` is the pair
    (ids, seqs_t): the integer [B,T,T] interval tensor is computed inside the kernel."""

    def __init__(self, num_units, num_heads, dropout_rate, l2_reg, pcoding_K, pcoding_V, tcoding_K, tcoding_V, gen=None):
        super().__init__()
        self.num_units, self.num_heads, self.dropout_rate, self.l2_reg = num_units, num_heads, dropout_rate, l2_reg
        self.q_kernel = nn.Parameter(glorot_uniform_(torch.empty(num_units, num_units), gen))
        self.q_bias = nn.Parameter(torch.zeros(num_units))
        kv = torch.cat([glorot_uniform_(torch.empty(num_units, num_units), gen) for _ in range(2)], dim=1)
        self.kv_kernel = nn.Parameter(kv)
        self.kv_bias = nn.Parameter(torch.zeros(2 * num_units))
        for n, m in (("pcoding_K", pcoding_K), ("pcoding_V", pcoding_V), ("tcoding_K", tcoding_K), ("tcoding_V", tcoding_V)):
            object.__setattr__(self, n, m)   # shared with the model: not registered twice
        self.compute = lambda p: p
        self.time_scale, self.timelen = 1.0, 256

    def forward(self, queries, keys, intervals, is_training, causality=True, drop: ops.Drop = ops.NO_DROP):
        if not causality:
            raise NotImplementedError("the HIP interval attention implements the causal form TiSASRec uses (TiSASREC.py:70-71)")
        ids, ts = intervals
        q = ops.LinearFn.apply(queries, self.q_kernel, self.q_bias, self.compute(self.q_kernel), False)
        kv = ops.LinearFn.apply(keys, self.kv_kernel, self.kv_bias, self.compute(self.kv_kernel), False)
        kt, vt = self.tcoding_K.pembs.lookup_table, self.tcoding_V.pembs.lookup_table
        return ops.TiAttnFn.apply(q, kv, queries, self.pcoding_K.pembs.lookup_table, self.pcoding_V.pembs.lookup_table, kt, vt,
                                  self.compute(kt), self.compute(vt), ids, ts, self.num_heads, self.time_scale, self.timelen,
                                  drop if is_training else ops.NO_DROP, float(getattr(self, "qk_scale", 0.0)))

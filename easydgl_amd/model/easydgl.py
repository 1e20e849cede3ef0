"""Mirror of src/model/EasyDGL.py — same constructor / call surface, HIP kernels underneath.

    m = EasyDGL(num_items, FLAGS).finalize("cuda")
    logits = m(features, is_training)                 # EasyDGL.__call__  (EasyDGL.py:69-151)
    loss   = m.train_loss(features, labels)           # the scalar EasyDGL.train minimises (:153-188)
    loss   = m.train_step(features, labels)           # + backward + Adam (the reference's train_op)
    m.eval_step(features, labels, mask_seen=True)     # Sequential.eval (Base.py:150-207) accumulators

``features``: dict with ``seqs_i`` int64 [B,T], ``seqs_t`` float32 [B,T] (seconds), ``masked_positions``
int64 [B,M] (training only) — exactly what MAUPostProcessor emits (dataloader.py:159-206).
"""
from __future__ import annotations

import pickle
from typing import Dict

import numpy as np
import torch
from torch import nn

from .. import ops
from ..module import coding as C
from ..module import temporal as T
from ..module.coding import glorot_uniform_
from .base import Sequential


class _Dense(nn.Module):
    """tf.layers.dense parameters: kernel [in,out] glorot_uniform, bias zeros."""

    def __init__(self, n_in, n_out, gen):
        super().__init__()
        self.kernel = nn.Parameter(glorot_uniform_(torch.empty(n_in, n_out), gen))
        self.bias = nn.Parameter(torch.zeros(n_out))


class _LayerNorm(nn.Module):
    """Base.layernorm parameters (Base.py:36-49): beta zeros, gamma ones over the last axis."""

    def __init__(self, n):
        super().__init__()
        self.beta = nn.Parameter(torch.zeros(n))
        self.gamma = nn.Parameter(torch.ones(n))


class _Block(nn.Module):
    def __init__(self, cin, C_, heads, events, att_drop, gen):
        super().__init__()
        self.attention = T.BiMAU(C_, heads, events, att_drop, in_units=cin, gen=gen)   # layer_i/attention/self/TMAU
        self.att_out = _Dense(C_, C_, gen)                                 # layer_i/attention/output/dense
        self.att_ln = _LayerNorm(C_)                                       # layer_i/attention/output/LayerNorm
        self.inter = _Dense(C_, 2 * C_, gen)                               # layer_i/intermediate/dense
        self.out = _Dense(2 * C_, C_, gen)                                 # layer_i/output/dense
        self.out_ln = _LayerNorm(C_)                                       # layer_i/output/LayerNorm


class EasyDGL(Sequential):

    def __init__(self, num_items, FLAGS):
        super().__init__(num_items, FLAGS)
        self.mask = num_items            # EasyDGL.py:39
        self.seqslen += 1                # EasyDGL.py:40
        self.num_items += 1              # EasyDGL.py:41
        self.masklen = FLAGS.masklen
        self.time_scale = float(FLAGS.time_scale)
        self.seed = int(getattr(FLAGS, "seed", 9876))
        table = getattr(FLAGS, "mark_table", None)
        if table is None:
            table = pickle.load(open(FLAGS.mark, "rb")).toarray()   # EasyDGL.py:45
        table = np.asarray(table)
        if table.shape[0] < num_items:
            raise ValueError("mark table needs one row per item id < num_items")
        if not np.isin(table, (0, 1)).all():
            raise ValueError("mark table must be 0/1 multi-hot")
        self.num_events = int(table.shape[-1])                     # EasyDGL.py:46
        if not (2 <= self.num_events <= T.MAX_EVENTS):
            raise ValueError(f"num_events must be in [2, {T.MAX_EVENTS}] (more than 16 run as mark groups: temporal.modulated_attention)")
        self.register_buffer("mark_lookup_table", torch.from_numpy(table.astype(np.uint8)), persistent=False)
        self.ct_reg = float(getattr(FLAGS, "ct_reg", 0.0) or 0.0)

        gen = torch.Generator().manual_seed(self.seed)
        C_ = self.num_units
        # variable scope "CSTMA" (EasyDGL.py:49-57)
        self.item_embs = C.Embedding(self.num_items, C_, self.l2_reg, zero_pad=True, scale=True, gen=gen)
        self.mark_embs = C.Embedding(self.num_events, C_, self.l2_reg, zero_pad=True, scale=False, gen=gen)
        self.pcoding = C.PositionCoding(self.seqslen, C_, self.l2_reg, gen=gen)
        self.tcoding = C.TimeSinusoidCoding(C_)
        self.output_bias = self.make_output_bias()
        self.layers = nn.ModuleList()
        for i in range(FLAGS.num_blocks):
            self.layers.append(_Block(3 * C_ if i == 0 else C_, C_, self.num_heads, self.num_events,
                                      self.attention_probs_dropout_rate, gen))
        self.transform = _Dense(C_, C_, gen)       # cls/predictions/transform/dense
        self.transform_ln = _LayerNorm(C_)         # cls/predictions/transform/LayerNorm
        self._metrics = None

    def l2_param_names(self):
        return ["item_embs.lookup_table", "mark_embs.lookup_table", "pcoding.pembs.lookup_table"]

    def finalize(self, device):
        super().finalize(device)
        for blk in self.layers:
            blk.attention.compute = self.compute
        return self

    # ---- helpers ---------------------------------------------------------------------------------------
    def _drop(self, rate, stream_id, is_training) -> ops.Drop:
        if not is_training or rate <= 0.0:
            return ops.NO_DROP
        return ops.Drop(rate, self._rng_state, stream_id)

    def _linear(self, x, d: _Dense, gelu=False):
        return ops.LinearFn.apply(x, d.kernel, d.bias, self.compute(d.kernel), gelu)

    def encode(self, features, is_training):
        """EasyDGL.py:70-95 -> (X0 [B,T,3C], spans [B,T], marks [B,T,E] uint8)."""
        ids, ts = features["seqs_i"], features["seqs_t"]
        tab = self.item_embs.lookup_table
        return ops.EncodeFn.apply(tab, self.pcoding.pembs.lookup_table, self.mark_embs.lookup_table, self.compute(tab),
                                  ids, ts, self.mark_lookup_table, self.tcoding.scale, self.mask, self.time_scale,
                                  self._drop(self.hidden_dropout_rate, 1, is_training), self.act_dtype)

    def encoder(self, features, is_training, gather_pos):
        """EasyDGL.py:70-146: returns (rows [B*Mg, C] at gather_pos, [lambda per block])."""
        ids = features["seqs_i"]
        C_ = self.num_units
        x, spans, marks = self.encode(features, is_training)
        lams = []
        for i, blk in enumerate(self.layers):
            layer_in = x
            att, lam = blk.attention(layer_in, layer_in, ids, spans, marks, is_training,
                                     drop=self._drop(self.attention_probs_dropout_rate, 10 + 4 * i, is_training))
            att = self._linear(att, blk.att_out)                                                   # :113
            att = ops.AddLayerNormFn.apply(att, layer_in[:, :, :C_], blk.att_ln.gamma, blk.att_ln.beta,
                                           self._drop(self.hidden_dropout_rate, 11 + 4 * i, is_training), None)  # :114-116
            inter = self._linear(att, blk.inter, gelu=True)                                        # :120-121
            out = self._linear(inter, blk.out)                                                     # :125
            x = ops.AddLayerNormFn.apply(out, att, blk.out_ln.gamma, blk.out_ln.beta,
                                         self._drop(self.hidden_dropout_rate, 12 + 4 * i, is_training), None)   # :126-128
            lams.append(lam)
        so = self._linear(x, self.transform, gelu=True)                                           # :138
        rows = ops.AddLayerNormFn.apply(so, None, self.transform_ln.gamma, self.transform_ln.beta, ops.NO_DROP,
                                        gather_pos)                                                # :139,142-146
        return rows, lams

    def _gather_pos(self, features, is_training):
        ids = features["seqs_i"]
        if is_training:
            return features["masked_positions"].contiguous()
        return torch.full((ids.shape[0], 1), ids.shape[1] - 1, device=ids.device, dtype=torch.int64)

    # ---- EasyDGL.__call__ ------------------------------------------------------------------------------
    def forward(self, features: Dict[str, torch.Tensor], is_training: bool):
        """Returns logits [B*M, I] (training) / [B, I] (eval), materialised like the reference (:149-151)."""
        rows, lams = self.encoder(features, is_training, self._gather_pos(features, is_training))
        self._last_lams = lams
        tab = self.item_embs.lookup_table
        return ops.ScoreLogitsFn.apply(rows, tab, self.output_bias, self.compute(tab))

    # ---- EasyDGL.train (the loss it builds) ------------------------------------------------------------------
    def train_loss(self, features, labels):
        """EasyDGL.py:153-188 with the fused scoring/CE path (no [B*M, I] tensor)."""
        rows, lams = self.encoder(features, True, self._gather_pos(features, True))
        tab = self.item_embs.lookup_table
        loss = ops.ScoreCEFn.apply(rows, tab, self.output_bias, self.compute(tab), labels.reshape(-1).contiguous())
        if self.l2_reg != 0.0:                                                                     # :158
            for p in (self.item_embs.lookup_table, self.mark_embs.lookup_table, self.pcoding.pembs.lookup_table):
                loss = loss + ops.L2Fn.apply(p, self.l2_reg)
        if self.ct_reg != 0.0:                                                                     # :159-175
            for lam in lams:
                loss = loss + ops.TppFn.apply(lam, features["masked_positions"].contiguous(), labels.contiguous(),
                                              features["seqs_t"], self.mark_lookup_table, self.num_heads,
                                              self.ct_reg / self.num_heads)
        return loss

    def train_step(self, features, labels):
        """One optimizer step: forward, backward, TF-Adam.  Returns the loss tensor (device scalar)."""
        self.settle_state()
        ops.rng_advance(self._rng_state)
        self.detach_grads()
        loss = self.train_loss(features, labels)
        loss.backward()
        self.collect_grads()
        self.optimizer_step()
        return loss.detach()

    # ---- Sequential.eval (Base.py:150-207) ----------------------------------------------------------------
    @torch.no_grad()
    def eval_topk(self, features, mask_seen=True, K=100):
        rows, _ = self.encoder(features, False, self._gather_pos(features, False))
        tab = self.item_embs.lookup_table
        return ops.score_topk(rows, self.compute(tab), self.output_bias, features["seqs_i"] if mask_seen else None, K, 0,
                              self.num_items)

    @torch.no_grad()
    def eval_topk_sharded(self, features, mask_seen=True, K=100, group=None, world=None, rank=None):
        """Row-sharded full-catalogue scoring: this rank scores the batch against its contiguous
        shard of the item table, masks seen ids that fall in the shard, keeps a local top-K with GLOBAL ids; one
        packed all-gather + merge kernel give the global top-K.  `world`/`rank` may be passed explicitly to
        emulate S shards in one process (tests); with torch.distributed they come from the process group."""
        from .. import parallel
        rows, _ = self.encoder(features, False, self._gather_pos(features, False))
        tab_c = self.compute(self.item_embs.lookup_table)
        seen = features["seqs_i"] if mask_seen else None

        def local_topk(i0, i1):
            if i1 <= i0:
                R = rows.shape[0]
                return (torch.full((R, K), float("-inf"), device=rows.device),
                        torch.full((R, K), -1, device=rows.device, dtype=torch.int32))
            return ops.score_topk(rows, tab_c, self.output_bias, seen, K, i0, i1)

        if world is not None:   # in-process emulation of `world` shards
            vals, idxs = zip(*(local_topk(*parallel.shard_bounds(self.num_items, world, r)) for r in range(world)))
            return ops.topk_merge(torch.stack(vals), torch.stack(idxs))
        return parallel.sharded_topk(local_topk, ops.topk_merge, self.num_items, K, group)

    def reset_metrics(self):
        self._metrics = torch.zeros(6, device=self._arena.device, dtype=torch.float32)
        self._metric_count = 0

    @torch.no_grad()
    def eval_step(self, features, labels, mask_seen=True):
        if self._metrics is None:
            self.reset_metrics()
        _, idx = self.eval_topk(features, mask_seen)
        ops.rank_metrics(idx, labels[:, -1].contiguous(), self._metrics)
        self._metric_count += labels.shape[0]

    def metrics(self) -> Dict[str, float]:
        vals = (self._metrics / max(1, self._metric_count)).tolist()
        return dict(zip(("H10", "H50", "H100", "N10", "N50", "N100"), vals))

    # ---- interop with the oracle's parameter naming (tests) -----------------------------------------------
    def tf_variable_map(self) -> Dict[str, nn.Parameter]:
        """Reference variable name (scope main/...) -> parameter."""
        m = {
            "CSTMA/item_embs/lookup_table": self.item_embs.lookup_table,
            "CSTMA/mark_embs/lookup_table": self.mark_embs.lookup_table,
            "CSTMA/spatial_embs/embedding/lookup_table": self.pcoding.pembs.lookup_table,
            "CSTMA/output_bias": self.output_bias,
            "cls/predictions/transform/dense/kernel": self.transform.kernel,
            "cls/predictions/transform/dense/bias": self.transform.bias,
            "cls/predictions/transform/LayerNorm/beta": self.transform_ln.beta,
            "cls/predictions/transform/LayerNorm/gamma": self.transform_ln.gamma,
        }
        for i, blk in enumerate(self.layers):
            pre = f"layer_{i}/"
            st = pre + "attention/self/TMAU/sequential_temporal_combined/"
            m[pre + "attention/self/TMAU/dense/kernel"] = blk.attention.dense_kernel
            m[pre + "attention/self/TMAU/dense/bias"] = blk.attention.dense_bias
            m[st + "dense/kernel"] = blk.attention.st_kernel
            m[st + "dense/bias"] = blk.attention.st_bias
            m[st + "weight"] = blk.attention.weight
            m[st + "scaling"] = blk.attention.scaling
            m[pre + "attention/output/dense/kernel"] = blk.att_out.kernel
            m[pre + "attention/output/dense/bias"] = blk.att_out.bias
            m[pre + "attention/output/LayerNorm/beta"] = blk.att_ln.beta
            m[pre + "attention/output/LayerNorm/gamma"] = blk.att_ln.gamma
            m[pre + "intermediate/dense/kernel"] = blk.inter.kernel
            m[pre + "intermediate/dense/bias"] = blk.inter.bias
            m[pre + "output/dense/kernel"] = blk.out.kernel
            m[pre + "output/dense/bias"] = blk.out.bias
            m[pre + "output/LayerNorm/beta"] = blk.out_ln.beta
            m[pre + "output/LayerNorm/gamma"] = blk.out_ln.gamma
        return m

    @torch.no_grad()
    def load_tf_variables(self, values: Dict[str, np.ndarray]) -> None:
        for name, p in self.tf_variable_map().items():
            p.copy_(torch.as_tensor(np.asarray(values[name]), dtype=torch.float32).to(p.device))
        self.sync_shadow()

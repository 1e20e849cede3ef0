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
        # A head dim the attention kernels do not tile (they take 16 / 32 / 64 / 128; the reference's own defaults are
        # --num_units 50 --num_heads 1, main.py:35-37): the model runs at the next supported head dim with ZERO-PADDED channels —
        # every parameter is stored at the padded width, its padded rows / columns are zero and stay zero (their gradients are
        # exactly zero: padded Q / K / V / T_ columns, W1 rows and dense rows are 0; the joint LayerNorm takes its moments over
        # the real channels and returns no gradient into the padded ones; the one exception, the intensity MLP's output weights
        # of the padded hidden units, is masked in mask_padded_grads), the score scale 1 / sqrt(dh) and coding.py's sqrt(C) are
        # those of the TRUE width.  tf_values() / load_tf_variables() / tf_gradients() speak the reference's shapes.
        self.width_true = self.num_units
        dh_true = self.num_units // self.num_heads if self.num_heads > 0 and self.num_units % self.num_heads == 0 else 0
        self.pad = (0, 0)
        if dh_true and dh_true not in T.SUPPORTED_HEAD_DIMS:
            dh_pad = next((d for d in T.SUPPORTED_HEAD_DIMS if d >= dh_true), None)
            if dh_pad is None or dh_true % 2:
                raise ValueError(f"EasyDGL: head dim num_units/num_heads = {dh_true}: even head dims up to {T.SUPPORTED_HEAD_DIMS[-1]} "
                                 "run (zero-padded to the next of 16 / 32 / 64 / 128)")
            self.pad = (dh_pad, dh_true)
            self.num_units = self.num_heads * dh_pad
        self.qk_scale = float(dh_true) ** -0.5 if self.pad[0] else 0.0
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
        if FLAGS.num_blocks == 0 and self.pad[0]:
            raise ValueError("EasyDGL: --num_blocks 0 with a channel-padded width is not supported (the head transform would read the "
                             "padded 3C-wide encoder output)")
        # (no block: the head transform reads the 3C-wide encoder output — tf.layers.dense builds its kernel from the input width,
        #  EasyDGL.py:138)
        self.transform = _Dense(3 * C_ if FLAGS.num_blocks == 0 else C_, C_, gen)       # cls/predictions/transform/dense
        self.transform_ln = _LayerNorm(C_)         # cls/predictions/transform/LayerNorm
        self._metrics = None
        if self.pad[0]:
            for blk in self.layers:
                blk.attention.qk_scale = self.qk_scale
            self._init_padded(gen)

    # ---- channel padding (self.pad = (dh_pad, dh_true)) ---------------------------------------------------------------
    def _pad_maps(self):
        """Index maps true -> padded for every axis kind of the parameters."""
        dhp, dht = self.pad
        H, E, Ct, Cp = self.num_heads, self.num_events, self.width_true, self.num_units
        c = torch.arange(Ct)
        cmap = (c // dht) * dhp + (c % dht)
        j = torch.arange(dht * E)
        return dict(c=cmap, c3=torch.cat([cmap + k * Cp for k in range(3)]), c4=torch.cat([cmap + k * Cp for k in range(4)]),
                    h2=torch.arange(2 * Ct), j=(j // dht) * dhp + (j % dht),
                    st_rows=torch.cat([torch.arange(dht), torch.tensor([dhp])]), u=torch.arange(dht))

    def _pad_specs(self):
        """TF variable name -> (parameter, axis maps (None = axis kept), initialiser of the TRUE-shape variable)."""
        mp = self._pad_maps()
        sp = {
            "CSTMA/item_embs/lookup_table": (self.item_embs.lookup_table, (None, mp["c"]), "glorot"),
            "CSTMA/mark_embs/lookup_table": (self.mark_embs.lookup_table, (None, mp["c"]), "glorot"),
            "CSTMA/spatial_embs/embedding/lookup_table": (self.pcoding.pembs.lookup_table, (None, mp["c"]), "glorot"),
            "CSTMA/output_bias": (self.output_bias, (None,), "zeros"),
            "cls/predictions/transform/dense/kernel": (self.transform.kernel, (mp["c"], mp["c"]), "glorot"),
            "cls/predictions/transform/dense/bias": (self.transform.bias, (mp["c"],), "zeros"),
            "cls/predictions/transform/LayerNorm/beta": (self.transform_ln.beta, (mp["c"],), "zeros"),
            "cls/predictions/transform/LayerNorm/gamma": (self.transform_ln.gamma, (mp["c"],), "ones"),
        }
        for i, blk in enumerate(self.layers):
            pre = f"layer_{i}/"
            st = pre + "attention/self/TMAU/sequential_temporal_combined/"
            a = blk.attention
            sp[pre + "attention/self/TMAU/dense/kernel"] = (a.dense_kernel, (mp["c3"] if i == 0 else mp["c"], mp["c4"]), "normal")
            sp[pre + "attention/self/TMAU/dense/bias"] = (a.dense_bias, (mp["c4"],), "zeros")
            sp[st + "dense/kernel"] = (a.st_kernel, (mp["st_rows"], mp["j"]), "glorot")
            sp[st + "dense/bias"] = (a.st_bias, (mp["j"],), "zeros")
            sp[st + "weight"] = (a.weight, (None, mp["u"]), "glorot")
            sp[st + "scaling"] = (a.scaling, (None,), "zeros")
            sp[pre + "attention/output/dense/kernel"] = (blk.att_out.kernel, (mp["c"], mp["c"]), "glorot")
            sp[pre + "attention/output/dense/bias"] = (blk.att_out.bias, (mp["c"],), "zeros")
            sp[pre + "attention/output/LayerNorm/beta"] = (blk.att_ln.beta, (mp["c"],), "zeros")
            sp[pre + "attention/output/LayerNorm/gamma"] = (blk.att_ln.gamma, (mp["c"],), "ones")
            sp[pre + "intermediate/dense/kernel"] = (blk.inter.kernel, (mp["c"], mp["h2"]), "glorot")
            sp[pre + "intermediate/dense/bias"] = (blk.inter.bias, (mp["h2"],), "zeros")
            sp[pre + "output/dense/kernel"] = (blk.out.kernel, (mp["h2"], mp["c"]), "glorot")
            sp[pre + "output/dense/bias"] = (blk.out.bias, (mp["c"],), "zeros")
            sp[pre + "output/LayerNorm/beta"] = (blk.out_ln.beta, (mp["c"],), "zeros")
            sp[pre + "output/LayerNorm/gamma"] = (blk.out_ln.gamma, (mp["c"],), "ones")
        return sp

    @staticmethod
    def _true_shape(p, maps):
        return tuple(p.shape[a] if m is None else len(m) for a, m in enumerate(maps))

    @staticmethod
    def _scatter(dst, maps, src):
        """dst (padded, any device) <- zeros, with src (true shape) at the mapped indices."""
        idx = [torch.arange(dst.shape[a]) if m is None else m for a, m in enumerate(maps)]
        dst.zero_()
        dst[torch.meshgrid(*[i.to(dst.device) for i in idx], indexing="ij")] = src.to(dst.device, dst.dtype)

    @staticmethod
    def _gather(src, maps):
        idx = [torch.arange(src.shape[a]) if m is None else m for a, m in enumerate(maps)]
        return src[torch.meshgrid(*[i.to(src.device) for i in idx], indexing="ij")]

    @torch.no_grad()
    def _init_padded(self, gen):
        """The reference's initialisers on the TRUE shapes (glorot limits of the true fans; temporal.py:393 N(0, 0.02)), scattered
        into the zero-padded storage; the time-code scales are those of the true width (coding.py:132-136)."""
        for name, (p, maps, kind) in self._pad_specs().items():
            shp = self._true_shape(p, maps)
            if kind == "glorot":
                v = glorot_uniform_(torch.empty(shp), gen)
            elif kind == "normal":
                v = torch.randn(shp, generator=gen) * 0.02
            elif kind == "ones":
                v = torch.ones(shp)
            else:
                v = torch.zeros(shp)
            self._scatter(p.data, maps, v)
        dhp, dht = self.pad
        Ct, Cp = self.width_true, self.num_units
        true_scale = np.power(10000, np.arange(0, Ct, 2) * 1.0 / Ct).astype(np.float32)      # pair j of the TRUE channels
        sc = np.ones(Cp // 2, np.float32)
        for c in range(0, Ct, 2):
            cp = (c // dht) * dhp + (c % dht)
            sc[cp // 2] = true_scale[c // 2]
        self.tcoding.scale = torch.from_numpy(sc).to(self.tcoding.scale.device)

    def mask_padded_grads(self) -> None:
        """The one gradient that is not zero on a padded entry by itself: the intensity MLP's output weights w[e, u >= dh_true]
        (their hidden units sit at sigmoid(0) = 1/2 and collect d z).  Zeroed before the optimizer sees them."""
        if not self.pad[0]:
            return
        for blk in self.layers:
            g = blk.attention.weight.grad
            if g is not None:
                g[:, self.pad[1]:].zero_()

    @torch.no_grad()
    def tf_values(self) -> Dict[str, torch.Tensor]:
        """Reference variable name -> value in the REFERENCE's shape (a copy; channel-padded models strip the padding)."""
        if not self.pad[0]:
            return {k: p.detach().clone() for k, p in self.tf_variable_map().items()}
        return {k: self._gather(p.detach(), maps).clone() for k, (p, maps, _) in self._pad_specs().items()}

    @torch.no_grad()
    def tf_gradients(self) -> Dict[str, torch.Tensor]:
        """Reference variable name -> gradient in the reference's shape (a copy)."""
        if not self.pad[0]:
            return {k: p.grad.detach().clone() for k, p in self.tf_variable_map().items()}
        return {k: self._gather(p.grad.detach(), maps).clone() for k, (p, maps, _) in self._pad_specs().items()}

    @torch.no_grad()
    def padded_leak(self) -> float:
        """Largest |value| on a padded entry of any parameter (0.0 for a healthy model; tests)."""
        worst = 0.0
        for k, (p, maps, _) in self._pad_specs().items():
            q = p.detach().clone()
            idx = [torch.arange(q.shape[a]) if m is None else m for a, m in enumerate(maps)]
            q[torch.meshgrid(*[i.to(q.device) for i in idx], indexing="ij")] = 0
            worst = max(worst, float(q.abs().max()))
        return worst

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
                                  self._drop(self.hidden_dropout_rate, 1, is_training), self.act_dtype, self.pad)

    def encoder(self, features, is_training, gather_pos):
        """EasyDGL.py:70-146: returns (rows [B*Mg, C] at gather_pos, [lambda per block])."""
        ids = features["seqs_i"]
        C_ = self.num_units
        x, spans, marks = self.encode(features, is_training)
        lams = []
        fused_eval = self._eval_tail_ok(is_training, x, gather_pos)
        for i, blk in enumerate(self.layers):
            layer_in = x
            att, lam = blk.attention(layer_in, layer_in, ids, spans, marks, is_training,
                                     drop=self._drop(self.attention_probs_dropout_rate, 10 + 4 * i, is_training))
            lams.append(lam)
            if fused_eval:      # inference: the block tail (and the head behind the last block) as ONE launch (csrc/k_tail.hip)
                last = i == len(self.layers) - 1
                x, rows = self._eval_tail(blk, att.contiguous(), layer_in, gather_pos, last)
                if last:
                    return rows, lams
                continue
            att = self._linear(att, blk.att_out)                                                   # :113
            att = ops.AddLayerNormFn.apply(att, layer_in[:, :, :C_], blk.att_ln.gamma, blk.att_ln.beta,
                                           self._drop(self.hidden_dropout_rate, 11 + 4 * i, is_training), None, self.pad)  # :114-116
            inter = self._linear(att, blk.inter, gelu=True)                                        # :120-121
            out = self._linear(inter, blk.out)                                                     # :125
            x = ops.AddLayerNormFn.apply(out, att, blk.out_ln.gamma, blk.out_ln.beta,
                                         self._drop(self.hidden_dropout_rate, 12 + 4 * i, is_training), None, self.pad)   # :126-128
        so = self._linear(x, self.transform, gelu=True)                                           # :138
        rows = ops.AddLayerNormFn.apply(so, None, self.transform_ln.gamma, self.transform_ln.beta, ops.NO_DROP,
                                        gather_pos, self.pad)                                      # :139,142-146
        return rows, lams

    # ---- inference through the fused block tail (the training engine's forward kernel; EasyDGL.py:110-146) ---------------------
    def _eval_tail_ok(self, is_training, x, gather_pos) -> bool:
        import os
        from .. import _lib
        if is_training or torch.is_grad_enabled() or not self.layers or x.dtype != torch.bfloat16:
            return False
        if os.environ.get("EDGL_EVAL_FUSED_TAIL", "1") == "0" or gather_pos.shape[1] > 256:
            return False
        return bool(_lib.lib.edgl_tail_supported(x.shape[1], self.num_units, _lib.BF16))

    def _eval_tail(self, blk, att, layer_in, gather_pos, last):
        """One block tail in one launch: att_out dense -> +residual -> LN1 -> inner dense + GELU -> out dense -> +residual -> LN2
        (-> head transform + GELU + LN3 + row gather behind the last block).  The tensors the kernel saves for a backward go to
        a scratch set kept per (batch, length); dropout rate 0.  Returns (y [B, T, C], head rows [B*Mg, C] or None)."""
        from .. import _lib
        lib, P = _lib.lib, ops._ptr
        B, T, C = att.shape
        Mg = gather_pos.shape[1]
        dev, dt = att.device, att.dtype
        key = (B, T, Mg, str(dev))
        ws = getattr(self, "_eval_tail_ws", None)
        if ws is None or ws["key"] != key:
            e = lambda *sh, d=dt: torch.empty(sh, device=dev, dtype=d)  # noqa: E731
            ws = dict(key=key, pack=e(int(lib.edgl_tail_pack_elems(C))), ao=e(B, T, C), a1=e(B, T, C), pre_f=e(B, T, 2 * C),
                      f=e(B, T, 2 * C), o=e(B, T, C), pre_t=e(B, T, C), so=e(B, T, C), st1=e(B, 2, d=torch.float32),
                      st2=e(B, 2, d=torch.float32), st3=e(B, 2, d=torch.float32))
            self._eval_tail_ws = ws
        st = ops._stream()
        _lib.check(lib.edgl_tail_pack(P(self.compute(blk.att_out.kernel)), P(self.compute(blk.inter.kernel)),
                                      P(self.compute(blk.out.kernel)), P(self.compute(self.transform.kernel)), C, P(ws["pack"]), st),
                   "edgl_tail_pack")
        y = torch.empty((B, T, C), device=dev, dtype=dt)
        rows = torch.empty((B * Mg, C), device=dev, dtype=dt) if last else None
        tl = self.transform_ln
        _lib.check(lib.edgl_tail_fwd_ct(P(att), layer_in.data_ptr(), layer_in.shape[2], P(ws["pack"]), P(blk.att_out.bias),
                                     P(blk.inter.bias), P(blk.out.bias), P(self.transform.bias), P(blk.att_ln.gamma),
                                     P(blk.att_ln.beta), P(blk.out_ln.gamma), P(blk.out_ln.beta), P(tl.gamma), P(tl.beta), B, T, C,
                                     0.0, None, 0, 0, P(gather_pos), Mg, int(last), P(ws["ao"]), P(ws["a1"]), P(ws["st1"]),
                                     P(ws["pre_f"]), P(ws["f"]), P(ws["o"]), P(y), P(ws["st2"]), P(ws["pre_t"]), P(ws["so"]),
                                     P(ws["st3"]), P(rows), None, self.pad[0], self.pad[1], _lib.BF16, st), "edgl_tail_fwd")
        return y, rows

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
        self.mask_padded_grads()
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
        if self.pad[0]:
            for name, (p, maps, _) in self._pad_specs().items():
                v = torch.as_tensor(np.asarray(values[name]), dtype=torch.float32)
                if tuple(v.shape) != self._true_shape(p, maps):
                    raise ValueError(f"{name}: expected shape {self._true_shape(p, maps)}, got {tuple(v.shape)}")
                self._scatter(p.data, maps, v)
        else:
            for name, p in self.tf_variable_map().items():
                p.copy_(torch.as_tensor(np.asarray(values[name]), dtype=torch.float32).to(p.device))
        self.sync_shadow()

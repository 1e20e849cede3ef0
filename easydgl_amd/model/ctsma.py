"""Mirror of src/model/CTSMA.py (ICML'21 continuous-time self-modulating attention) on the HIP kernels — SURVEY §8
row f-4: the causal MAU is the BiMAU kernel with two flags, the rest is LayerNorm / dense / scoring ops that exist.

    m = CTSMA(num_items, FLAGS).finalize("cuda")
    logits = m(features, is_training)          # CTSMA.__call__ (CTSMA.py:48-93): [B*T, I] (train) / [B, I] (eval)
    loss = m.train_loss(features, labels)      # CTSMA.train (CTSMA.py:95-127)

``features`` as RegressivePostProcessor emits them (dataloader.py:88-108, keep_entire): ``seqs_i`` = tokens[:-1] int64
[B,T], ``seqs_t`` float32 [B,T+1]; labels = tokens[1:] [B,T] (training) / tokens [B,T+1] (evaluation).  Unlike EasyDGL
the model keeps T = FLAGS.seqslen positions and a [num_items, C] table (CTSMA.py:23-37).
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
from .base import Sequential
from .easydgl import EasyDGL, _Dense, _LayerNorm


class _FeedForward(nn.Module):
    """Base.FeedForward([C, C]) (Base.py:70-87): Conv1D(k=1, relu) -> dropout -> Conv1D(k=1) -> dropout -> + input."""

    def __init__(self, C_, gen):
        super().__init__()
        self.inner = _Dense(C_, C_, gen)      # dense/Inner
        self.readout = _Dense(C_, C_, gen)    # dense/Readout


class _Block(nn.Module):
    def __init__(self, cin, C_, heads, events, att_drop, gen):
        super().__init__()
        self.att_ln = _LayerNorm(cin)                                    # num_blocks_i/attention/LayerNorm
        self.attention = T.MAU(C_, heads, events, att_drop, in_units=cin, gen=gen)    # num_blocks_i/attention/modulating_attention
        self.ff_ln = _LayerNorm(C_)                                      # num_blocks_i/feed-forward/LayerNorm
        self.ff = _FeedForward(C_, gen)


class CTSMA(Sequential):
    def __init__(self, num_items, FLAGS):
        super().__init__(num_items, FLAGS)
        self.time_scale = float(FLAGS.time_scale)
        self.seed = int(getattr(FLAGS, "seed", 9876))
        table = getattr(FLAGS, "mark_table", None)
        if table is None:
            table = pickle.load(open(FLAGS.mark, "rb")).toarray()     # CTSMA.py:24
        table = np.asarray(table)
        if table.shape[0] < num_items or not np.isin(table, (0, 1)).all():
            raise ValueError("mark table must be a 0/1 multi-hot table with one row per item id")
        self.num_events = int(table.shape[-1])
        if not (2 <= self.num_events <= T.MAX_EVENTS):
            raise ValueError(f"num_events must be in [2, {T.MAX_EVENTS}] (more than 16 run as mark groups: temporal.modulated_attention)")
        self.register_buffer("mark_lookup_table", torch.from_numpy(table.astype(np.uint8)), persistent=False)
        self.ct_reg = float(getattr(FLAGS, "ct_reg", 0.0) or 0.0)
        gen = torch.Generator().manual_seed(self.seed)
        self._setup_channel_pad("CTSMA")      # head dims below 128 that the kernels do not tile run zero-padded (model/base.py)
        C_ = self.num_units
        self.item_embs = C.Embedding(self.num_items, C_, self.l2_reg, zero_pad=True, scale=True, gen=gen)   # CTSMA.py:30-31
        self.pcoding = C.PositionCoding(self.seqslen, C_, self.l2_reg, gen=gen)                              # :32
        self.output_bias = self.make_output_bias()                                                           # :34
        self.layers = nn.ModuleList()
        for i in range(FLAGS.num_blocks):
            self.layers.append(_Block(2 * C_ if i == 0 else C_, C_, self.num_heads, self.num_events,
                                      self.attention_probs_dropout_rate, gen))
        self.out_ln = _LayerNorm(C_)                                                                         # outln
        self._metrics = None
        if self.pad[0]:
            for blk in self.layers:
                blk.attention.qk_scale = self.qk_scale
            self._init_padded(gen)

    def _pad_specs(self):
        """(TF name, parameter, axis maps true -> padded, initialiser) of every variable, for the channel-padded storage."""
        dhp, dht = self.pad
        E, Cp = self.num_events, self.num_units
        c = self._cmap()
        j = torch.arange(dht * E)
        jmap = (j // dht) * dhp + (j % dht)                                   # hidden unit (e, u) of the intensity MLP
        st_rows = torch.cat([torch.arange(dht), torch.tensor([dhp])])        # its inputs: dh channels + the span
        sp = [("CSTMA/item_embs/lookup_table", self.item_embs.lookup_table, (None, c), "glorot"),
              ("CSTMA/spatial_embs/embedding/lookup_table", self.pcoding.pembs.lookup_table, (None, c), "glorot"),
              ("CSTMA/output_bias", self.output_bias, (None,), "zeros"),
              ("outln/LayerNorm/gamma", self.out_ln.gamma, (c,), "ones"), ("outln/LayerNorm/beta", self.out_ln.beta, (c,), "zeros")]
        for i, blk in enumerate(self.layers):
            pre = f"num_blocks_{i}/"
            a = pre + "attention/modulating_attention/"
            st = a + "sequential_temporal_combined/"
            cin = self._cmap(2) if i == 0 else c                               # item | position channels of the first block's input
            att = blk.attention
            sp += [(pre + "attention/LayerNorm/gamma", blk.att_ln.gamma, (cin,), "ones"),
                   (pre + "attention/LayerNorm/beta", blk.att_ln.beta, (cin,), "zeros"),
                   (a + "dense/kernel", att.q_kernel, (cin, c), "glorot"), (a + "dense/bias", att.q_bias, (c,), "zeros")]
            for k in (1, 2, 3):
                sp += [(a + f"dense_{k}/kernel", att.kvt_kernel, (cin, c + (k - 1) * Cp), "glorot"),
                       (a + f"dense_{k}/bias", att.kvt_bias, (c + (k - 1) * Cp,), "zeros")]
            sp += [(st + "dense/kernel", att.st_kernel, (st_rows, jmap), "glorot"), (st + "dense/bias", att.st_bias, (jmap,), "zeros"),
                   (st + "weight", att.weight, (None, torch.arange(dht)), "glorot"), (st + "scaling", att.scaling, (None,), "zeros"),
                   (pre + "feed-forward/LayerNorm/gamma", blk.ff_ln.gamma, (c,), "ones"),
                   (pre + "feed-forward/LayerNorm/beta", blk.ff_ln.beta, (c,), "zeros"),
                   (pre + "feed-forward/Inner/kernel", blk.ff.inner.kernel, (c, c), "glorot"),
                   (pre + "feed-forward/Inner/bias", blk.ff.inner.bias, (c,), "zeros"),
                   (pre + "feed-forward/Readout/kernel", blk.ff.readout.kernel, (c, c), "glorot"),
                   (pre + "feed-forward/Readout/bias", blk.ff.readout.bias, (c,), "zeros")]
        return sp

    def l2_param_names(self):
        return ["item_embs.lookup_table", "pcoding.pembs.lookup_table"]

    def finalize(self, device):
        super().finalize(device)
        for blk in self.layers:
            blk.attention.compute = self.compute
        return self

    _drop = EasyDGL._drop
    reset_metrics = EasyDGL.reset_metrics
    metrics = EasyDGL.metrics

    def _linear(self, x, d: _Dense, act=False):
        return ops.LinearFn.apply(x, d.kernel, d.bias, self.compute(d.kernel), act)

    def encoder(self, features, is_training, gather_pos):
        """CTSMA.py:48-83: (rows [B*Mg, C] of the final LayerNorm at gather_pos (None: all positions), [lambda])."""
        ids, ts = features["seqs_i"].contiguous(), features["seqs_t"].contiguous()
        tab = self.item_embs.lookup_table
        hd = self.hidden_dropout_rate
        pad = self.pad
        x, spans, marks = ops.EmbedPosFn.apply(tab, self.pcoding.pembs.lookup_table, self.compute(tab), ids, ts,
                                               self.mark_lookup_table, self.time_scale, self._drop(hd, 1, is_training),
                                               self.act_dtype, self.width_true if pad[0] else 0)   # :50-58
        lams = []
        for i, blk in enumerate(self.layers):
            q_in = ops.AddLayerNormFn.apply(x, None, blk.att_ln.gamma, blk.att_ln.beta, ops.NO_DROP, None, pad)  # :70
            att, lam = blk.attention(q_in, x, ids, spans, marks, is_training, True,
                                     drop=self._drop(self.attention_probs_dropout_rate, 10 + 4 * i, is_training))
            y = ops.AddLayerNormFn.apply(att, None, blk.ff_ln.gamma, blk.ff_ln.beta, ops.NO_DROP, None, pad)     # :75
            inner = self._linear(y, blk.ff.inner, "relu")                                                        # Base.py:79
            if is_training and hd > 0.0:   # Base.py:80: dropout(inner) — an identity LayerNorm-free path: a scaled copy
                inner = ops.dropout(inner, self._drop(hd, 11 + 4 * i, True))
            out = self._linear(inner, blk.ff.readout)                                                            # Base.py:82
            x = ops.ff_tail(out, y, None, self._drop(hd, 12 + 4 * i, is_training))                               # Base.py:83-86
            lams.append(lam)
        rows = ops.AddLayerNormFn.apply(x, None, self.out_ln.gamma, self.out_ln.beta, ops.NO_DROP, gather_pos, pad)   # :82-83
        return rows, lams

    def forward(self, features: Dict[str, torch.Tensor], is_training: bool):
        ids = features["seqs_i"]
        gp = None if is_training else torch.full((ids.shape[0], 1), ids.shape[1] - 1, device=ids.device, dtype=torch.int64)
        rows, lams = self.encoder(features, is_training, gp)
        self._last_lams = lams
        rows = rows.reshape(-1, self.num_units)
        tab = self.item_embs.lookup_table
        return ops.ScoreLogitsFn.apply(rows, tab, self.output_bias, self.compute(tab))

    def train_loss(self, features, labels):
        """CTSMA.train (CTSMA.py:95-127) with the fused scoring / cross-entropy (no [B*T, I] tensor)."""
        rows, lams = self.encoder(features, True, None)
        tab = self.item_embs.lookup_table
        loss = ops.ScoreCEFn.apply(rows.reshape(-1, self.num_units), tab, self.output_bias, self.compute(tab),
                                   labels.reshape(-1).contiguous())
        if self.l2_reg != 0.0:
            for p in (self.item_embs.lookup_table, self.pcoding.pembs.lookup_table):
                loss = loss + ops.L2Fn.apply(p, self.l2_reg)
        if self.ct_reg != 0.0:                                                                     # :101-112
            for lam in lams:
                loss = loss + ops.TppFn.apply(lam, None, labels.contiguous(), features["seqs_t"].contiguous(),
                                              self.mark_lookup_table, self.num_heads, self.ct_reg)
        return loss

    train_step = EasyDGL.train_step

    @torch.no_grad()
    def eval_topk(self, features, mask_seen=True, K=100):
        logits = self.forward(features, False)
        return ops.mask_topk(logits, 0, features["seqs_i"] if mask_seen else None, K)

    eval_step = EasyDGL.eval_step

    # ---- interop with the reference's variable naming (tests / checkpoints converted from TF) ------------------------
    def load_tf_variables(self, values: Dict[str, np.ndarray]) -> None:
        """`values`: TF variable name (scope `main/` stripped) -> array, as oracle/ctsma_ref.init_params lists them."""
        if self.pad[0]:
            return self._load_padded(values)
        def put(param, arr):
            with torch.no_grad():
                param.copy_(torch.as_tensor(np.asarray(arr), dtype=param.dtype).reshape(param.shape))
        put(self.item_embs.lookup_table, values["CSTMA/item_embs/lookup_table"])
        put(self.pcoding.pembs.lookup_table, values["CSTMA/spatial_embs/embedding/lookup_table"])
        put(self.output_bias, values["CSTMA/output_bias"])
        for i, blk in enumerate(self.layers):
            pre = f"num_blocks_{i}/"
            a = pre + "attention/modulating_attention/"
            put(blk.att_ln.gamma, values[pre + "attention/LayerNorm/gamma"])
            put(blk.att_ln.beta, values[pre + "attention/LayerNorm/beta"])
            put(blk.attention.q_kernel, values[a + "dense/kernel"])
            put(blk.attention.q_bias, values[a + "dense/bias"])
            put(blk.attention.kvt_kernel, np.concatenate([values[a + f"dense_{j}/kernel"] for j in (1, 2, 3)], axis=1))
            put(blk.attention.kvt_bias, np.concatenate([values[a + f"dense_{j}/bias"] for j in (1, 2, 3)]))
            st = a + "sequential_temporal_combined/"
            put(blk.attention.st_kernel, values[st + "dense/kernel"])
            put(blk.attention.st_bias, values[st + "dense/bias"])
            put(blk.attention.weight, values[st + "weight"])
            put(blk.attention.scaling, values[st + "scaling"])
            put(blk.ff_ln.gamma, values[pre + "feed-forward/LayerNorm/gamma"])
            put(blk.ff_ln.beta, values[pre + "feed-forward/LayerNorm/beta"])
            put(blk.ff.inner.kernel, values[pre + "feed-forward/Inner/kernel"])
            put(blk.ff.inner.bias, values[pre + "feed-forward/Inner/bias"])
            put(blk.ff.readout.kernel, values[pre + "feed-forward/Readout/kernel"])
            put(blk.ff.readout.bias, values[pre + "feed-forward/Readout/bias"])
        put(self.out_ln.gamma, values["outln/LayerNorm/gamma"])
        put(self.out_ln.beta, values["outln/LayerNorm/beta"])
        self.sync_shadow()

    def tf_gradients(self) -> Dict[str, np.ndarray]:
        """Gradients of the last backward under the TF variable names (K|V|T_ split back into dense_1..3)."""
        if self.pad[0]:
            return {k: v.float().cpu().numpy() for k, v in self._padded_values(True).items()}
        g = lambda q: q.grad.detach().float().cpu().numpy()
        out = {"CSTMA/item_embs/lookup_table": g(self.item_embs.lookup_table),
               "CSTMA/spatial_embs/embedding/lookup_table": g(self.pcoding.pembs.lookup_table), "CSTMA/output_bias": g(self.output_bias)}
        C_ = self.num_units
        for i, blk in enumerate(self.layers):
            pre = f"num_blocks_{i}/"
            a = pre + "attention/modulating_attention/"
            out[pre + "attention/LayerNorm/gamma"], out[pre + "attention/LayerNorm/beta"] = g(blk.att_ln.gamma), g(blk.att_ln.beta)
            out[a + "dense/kernel"], out[a + "dense/bias"] = g(blk.attention.q_kernel), g(blk.attention.q_bias)
            kk, kb = g(blk.attention.kvt_kernel), g(blk.attention.kvt_bias)
            for j in (1, 2, 3):
                out[a + f"dense_{j}/kernel"], out[a + f"dense_{j}/bias"] = kk[:, (j - 1) * C_:j * C_], kb[(j - 1) * C_:j * C_]
            st = a + "sequential_temporal_combined/"
            out[st + "dense/kernel"], out[st + "dense/bias"] = g(blk.attention.st_kernel), g(blk.attention.st_bias)
            out[st + "weight"], out[st + "scaling"] = g(blk.attention.weight), g(blk.attention.scaling)
            out[pre + "feed-forward/LayerNorm/gamma"], out[pre + "feed-forward/LayerNorm/beta"] = g(blk.ff_ln.gamma), g(blk.ff_ln.beta)
            out[pre + "feed-forward/Inner/kernel"], out[pre + "feed-forward/Inner/bias"] = g(blk.ff.inner.kernel), g(blk.ff.inner.bias)
            out[pre + "feed-forward/Readout/kernel"], out[pre + "feed-forward/Readout/bias"] = g(blk.ff.readout.kernel), g(blk.ff.readout.bias)
        out["outln/LayerNorm/gamma"], out["outln/LayerNorm/beta"] = g(self.out_ln.gamma), g(self.out_ln.beta)
        return out

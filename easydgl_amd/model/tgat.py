"""Mirror of src/model/TGAT.py (Xu et al., ICLR'20, as the reference re-implements it) on the HIP kernels — SURVEY §8 row
a-14, BASELINE.json config 5 ("TGAT ... through the same HIP attention/time-kernel path").

    m = TGAT(num_items, FLAGS).finalize("cuda")
    logits = m(features, is_training)          # TGAT.__call__ (TGAT.py:44-83): [B*T, I] (train) / [B, I] (eval)
    loss = m.train_loss(features, labels)      # Sequential.train (Base.py:119-140)

``features`` as RegressivePostProcessor emits them (dataloader.py:95-108): ``seqs_i`` = tokens[:-1] int64 [B,T] with
T = FLAGS.seqslen, ``seqs_t`` float32 [B,T+1]; labels = tokens[1:] (training) / tokens (evaluation).

Timestamps must not decrease along a sequence (the reference's own data is time-sorted, data/linkpred.py): the kernels use
cos(a-b) = cos a cos b + sin a sin b for the time term, which equals the reference's cos(max(a-b, 0) ...) only then.
``check_inputs()`` raises if a batch violated it."""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
from torch import nn

from .. import ops
from ..module import coding as C
from ..module import temporal as T
from .base import Sequential
from .ctsma import _FeedForward
from .easydgl import EasyDGL, _Dense, _LayerNorm


class _Block(nn.Module):
    def __init__(self, C_, heads, att_drop, l2, pcoding_K, tcoding_K, gen):
        super().__init__()
        self.att_ln = _LayerNorm(C_)                                                       # num_blocks_i/attention/LayerNorm
        self.attention = T.TfMultiHeadAttention(C_, heads, att_drop, l2, pcoding_K, tcoding_K, gen)
        self.ff_ln = _LayerNorm(C_)                                                        # num_blocks_i/feedforward/LayerNorm
        self.ff = _FeedForward(C_, gen)


class TGAT(Sequential):
    def __init__(self, num_items, FLAGS):
        super().__init__(num_items, FLAGS)
        self.time_scale = float(FLAGS.time_scale)
        self.seed = int(getattr(FLAGS, "seed", 9876))
        gen = torch.Generator().manual_seed(self.seed)
        # a head dim below 128 that the kernels do not tile (the reference's default --num_units 50, main.py:35) runs zero-padded
        # to the next of 16 / 32 / 64 / 128 (model/base.py: channel padding)
        self._setup_channel_pad("TGAT")
        C_ = self.num_units
        dh = C_ // max(1, self.num_heads)
        if C_ % self.num_heads or not (dh in (16, 32, 64, 128) or (dh > 128 and dh % 128 == 0)):
            raise ValueError(f"TGAT: head dim num_units/num_heads = {dh} unsupported; the HIP attention kernels take up to 128 "
                             f"(zero-padded to 16 / 32 / 64 / 128) or a multiple of 128 (e.g. --num_units=512 --num_heads=1, runme.sh:80-87)")
        if C_ > 512 or C_ & (C_ - 1) or C_ < 32:
            raise ValueError(f"TGAT: num_units={self.width_true} (stored as {C_}) unsupported: the fused scoring kernels take a "
                             f"power of two in [32, 512] (with padded heads: a power-of-two head count)")
        self.item_embs = C.Embedding(num_items, C_, self.l2_reg, zero_pad=True, scale=True, gen=gen)     # TGAT.py:27-28
        self.pcoding_K = C.PositionCoding(self.seqslen, C_, self.l2_reg, gen=gen)                         # :29
        self.tcoding_K = C.TimeFunctionCoding(C_)                                                         # :30
        self.output_bias = self.make_output_bias()                                                        # :32
        self.layers = nn.ModuleList()
        for _ in range(FLAGS.num_blocks):
            self.layers.append(_Block(C_, self.num_heads, self.attention_probs_dropout_rate, self.l2_reg, self.pcoding_K,
                                      self.tcoding_K, gen))
        self.out_ln = _LayerNorm(C_)
        self._metrics = None
        self._finish_pad(gen)

    def _finish_pad(self, gen):
        if self.pad[0]:
            for blk in self.layers:
                blk.attention.qk_scale = self.qk_scale
            self._init_padded(gen)

    def _pad_specs(self):
        """(TF name, parameter, axis maps, initialiser) of every variable in _tf_map(), for the channel-padded storage."""
        c = self._cmap()
        specs = []
        for name, (param, sl) in self._tf_map().items():
            leaf = name.rsplit("/", 1)[-1]
            if leaf == "lookup_table":
                maps, kind = (None, c), "glorot"
            elif leaf == "output_bias":
                maps, kind = (None,), "zeros"
            elif leaf == "kernel":
                maps, kind = (c, c if sl is None else c + sl.start), "glorot"      # sl: the K / V column block of kv_kernel
            elif leaf == "basis_freq":
                maps, kind = (c,), "linspace9"
            else:   # bias / beta / gamma / phase
                maps, kind = (c if sl is None else c + sl.start,), ("ones" if leaf == "gamma" else "zeros")
            specs.append((name, param, maps, kind))
        return specs

    def l2_param_names(self):
        return ["item_embs.lookup_table", "pcoding_K.pembs.lookup_table"]

    def finalize(self, device):
        super().finalize(device)
        self._violations = torch.zeros(1, device=device, dtype=torch.int32)
        for blk in self.layers:
            blk.attention.compute = self.compute
            blk.attention.time_scale = self.time_scale
            blk.attention.violations = self._violations
        return self

    def check_inputs(self) -> None:
        """Raises if any batch so far had decreasing timestamps at an unpadded position (one device read)."""
        n = int(self._violations.item())
        if n:
            self._violations.zero_()
            raise ValueError(f"{n} positions with decreasing timestamps: sort each sequence by time (see model/tgat.py)")

    _drop = EasyDGL._drop
    reset_metrics = EasyDGL.reset_metrics
    metrics = EasyDGL.metrics

    def _linear(self, x, d: _Dense, act=False):
        return ops.LinearFn.apply(x, d.kernel, d.bias, self.compute(d.kernel), act)

    def encoder(self, features, is_training, gather_pos):
        """TGAT.py:44-73: rows of the final LayerNorm at gather_pos (None: all positions)."""
        ids, ts = features["seqs_i"].contiguous(), features["seqs_t"].contiguous()
        tab = self.item_embs.lookup_table
        hd = self.hidden_dropout_rate
        # :49-62 — `* seqs_masks` is the identity here: row 0 of the table reads as zeros (coding.py:56-57)
        pad = self.pad
        x = ops.EmbedFn.apply(tab, self.compute(tab), ids, self._drop(hd, 1, is_training), self.act_dtype,
                              self.width_true if pad[0] else 0)
        for i, blk in enumerate(self.layers):
            qn = ops.AddLayerNormFn.apply(x, None, blk.att_ln.gamma, blk.att_ln.beta, ops.NO_DROP, None, pad)     # :66
            att = blk.attention(qn, x, (ids, ts), is_training, True,
                                self._drop(self.attention_probs_dropout_rate, 10 + 4 * i, is_training))
            y = ops.AddLayerNormFn.apply(att, None, blk.ff_ln.gamma, blk.ff_ln.beta, ops.NO_DROP, None, pad)      # :69
            inner = self._linear(y, blk.ff.inner, "relu")                                                         # Base.py:79
            if is_training and hd > 0.0:
                inner = ops.dropout(inner, self._drop(hd, 11 + 4 * i, True))                                      # Base.py:80
            out = self._linear(inner, blk.ff.readout)                                                             # Base.py:82
            x = ops.ff_tail(out, y, ids, self._drop(hd, 12 + 4 * i, is_training))                                 # Base.py:83-86, TGAT.py:70
        return ops.AddLayerNormFn.apply(x, None, self.out_ln.gamma, self.out_ln.beta, ops.NO_DROP, gather_pos, pad)   # :72-73

    def forward(self, features: Dict[str, torch.Tensor], is_training: bool):
        ids = features["seqs_i"]
        gp = None if is_training else torch.full((ids.shape[0], 1), ids.shape[1] - 1, device=ids.device, dtype=torch.int64)
        rows = self.encoder(features, is_training, gp).reshape(-1, self.num_units)
        tab = self.item_embs.lookup_table
        return ops.ScoreLogitsFn.apply(rows, tab, self.output_bias, self.compute(tab))                            # :74-83

    def train_loss(self, features, labels):
        """Sequential.train (Base.py:119-131) with the fused scoring / cross-entropy (no [B*T, I] tensor)."""
        rows = self.encoder(features, True, None)
        tab = self.item_embs.lookup_table
        loss = ops.ScoreCEFn.apply(rows.reshape(-1, self.num_units), tab, self.output_bias, self.compute(tab),
                                   labels.reshape(-1).contiguous())
        if self.l2_reg != 0.0:
            for n in self.l2_param_names():
                loss = loss + ops.L2Fn.apply(self.get_parameter(n), self.l2_reg)
        return loss

    train_step = EasyDGL.train_step

    @torch.no_grad()
    def eval_topk(self, features, mask_seen=True, K=100):
        logits = self.forward(features, False)
        return ops.mask_topk(logits, 0, features["seqs_i"] if mask_seen else None, K)

    eval_step = EasyDGL.eval_step

    # ---- interop with the reference's variable naming (tests / checkpoints converted from TF) ------------------------
    def _tf_map(self):
        """TF variable name -> (parameter, column slice or None)."""
        C_ = self.num_units
        m = {"TGAT/item_embs/lookup_table": (self.item_embs.lookup_table, None),
             "TGAT/pcoding_K/embedding/lookup_table": (self.pcoding_K.pembs.lookup_table, None),
             "TGAT/tcoding_K/basis_freq": (self.tcoding_K.basis_freq, None),
             "TGAT/tcoding_K/phase": (self.tcoding_K.phase, None),
             "TGAT/output_bias": (self.output_bias, None),
             "out_ln/LayerNorm/gamma": (self.out_ln.gamma, None), "out_ln/LayerNorm/beta": (self.out_ln.beta, None)}
        for i, blk in enumerate(self.layers):
            pre = f"num_blocks_{i}/"
            a = pre + "attention/attention/timeinterval/"
            m[pre + "attention/LayerNorm/gamma"] = (blk.att_ln.gamma, None)
            m[pre + "attention/LayerNorm/beta"] = (blk.att_ln.beta, None)
            m[a + "dense/kernel"], m[a + "dense/bias"] = (blk.attention.q_kernel, None), (blk.attention.q_bias, None)
            for j in (1, 2):
                sl = slice((j - 1) * C_, j * C_)
                m[a + f"dense_{j}/kernel"], m[a + f"dense_{j}/bias"] = (blk.attention.kv_kernel, sl), (blk.attention.kv_bias, sl)
            m[pre + "feedforward/LayerNorm/gamma"] = (blk.ff_ln.gamma, None)
            m[pre + "feedforward/LayerNorm/beta"] = (blk.ff_ln.beta, None)
            m[pre + "feedforward/Inner/kernel"], m[pre + "feedforward/Inner/bias"] = (blk.ff.inner.kernel, None), (blk.ff.inner.bias, None)
            m[pre + "feedforward/Readout/kernel"], m[pre + "feedforward/Readout/bias"] = (blk.ff.readout.kernel, None), (blk.ff.readout.bias, None)
        return m

    def load_tf_variables(self, values: Dict[str, np.ndarray]) -> None:
        if self.pad[0]:
            return self._load_padded(values)
        with torch.no_grad():
            for name, (param, sl) in self._tf_map().items():
                src = torch.as_tensor(np.asarray(values[name]), dtype=param.dtype)
                (param if sl is None else param[..., sl]).copy_(src)
        self.sync_shadow()

    def tf_values(self) -> Dict[str, np.ndarray]:
        """Reference variable name -> value in the reference's shape (channel-padded models strip the padding)."""
        if self.pad[0]:
            return {k: v.float().cpu().numpy() for k, v in self._padded_values(False).items()}
        return {name: (p if sl is None else p[..., sl]).detach().float().cpu().numpy() for name, (p, sl) in self._tf_map().items()}

    def tf_gradients(self) -> Dict[str, np.ndarray]:
        if self.pad[0]:
            return {k: v.float().cpu().numpy() for k, v in self._padded_values(True).items()}
        out = {}
        for name, (param, sl) in self._tf_map().items():
            g = param.grad.detach().float().cpu().numpy()
            out[name] = g if sl is None else g[..., sl]
        return out

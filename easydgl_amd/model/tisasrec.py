"""Mirror of src/model/TiSASREC.py (Li et al., WSDM'20, as the reference re-implements it) on the HIP kernels — SURVEY §8 row
a-15, BASELINE.json config 5.  Same block wiring as TGAT (model/tgat.py); the attention is T.TiMultiHeadAttention: integer
interval buckets int(clip(ts[q+1] - ts[k], 0, timelen)) index two [timelen, C] tables on the key and on the value side.

Stated quirks kept from the reference: the bucket `timelen` is one past the tables (the GPU lookup returns zeros); positions use
the first seqslen rows of [timelen, C] tables, so seqslen <= timelen is required."""
from __future__ import annotations

import torch
from torch import nn

from ..module import coding as C
from ..module import temporal as T
from .base import Sequential
from .ctsma import _FeedForward
from .easydgl import _LayerNorm
from .tgat import TGAT


class _Block(nn.Module):
    def __init__(self, C_, heads, att_drop, l2, codings, gen):
        super().__init__()
        self.att_ln = _LayerNorm(C_)
        self.attention = T.TiMultiHeadAttention(C_, heads, att_drop, l2, *codings, gen=gen)
        self.ff_ln = _LayerNorm(C_)
        self.ff = _FeedForward(C_, gen)


class TiSASRec(TGAT):
    def __init__(self, num_items, FLAGS):
        Sequential.__init__(self, num_items, FLAGS)
        self.timelen = int(FLAGS.timelen)
        self.time_scale = float(FLAGS.time_scale)
        self.seed = int(getattr(FLAGS, "seed", 9876))
        gen = torch.Generator().manual_seed(self.seed)
        self._setup_channel_pad("TiSASRec")      # head dims below 128 that the kernels do not tile run zero-padded (model/base.py)
        C_ = self.num_units
        if C_ % self.num_heads or (C_ // self.num_heads) not in (16, 32, 64, 128):
            raise ValueError("TiSASRec on the HIP attention kernel needs a head dim of at most 128 (zero-padded to 16 / 32 / 64 / 128)")
        if C_ > 512 or C_ & (C_ - 1) or C_ < 32:
            raise ValueError(f"TiSASRec: num_units={self.width_true} (stored as {C_}) unsupported: the fused scoring kernels take a "
                             f"power of two in [32, 512]")
        if not (self.seqslen <= self.timelen <= 256):
            raise ValueError("need seqslen <= timelen <= 256 (position rows come from [timelen, C] tables, TiSASREC.py:30-31)")
        self.item_embs = C.Embedding(num_items, C_, self.l2_reg, zero_pad=True, scale=True, gen=gen)     # TiSASREC.py:27-28
        self.pcoding_K = C.PositionCoding(self.timelen, C_, self.l2_reg, gen=gen)                         # :30
        self.pcoding_V = C.PositionCoding(self.timelen, C_, self.l2_reg, gen=gen)                         # :31
        self.tcoding_K = C.TimeIntervalCoding(self.timelen, C_, self.l2_reg, gen=gen)                     # :32
        self.tcoding_V = C.TimeIntervalCoding(self.timelen, C_, self.l2_reg, gen=gen)                     # :33
        self.output_bias = self.make_output_bias()
        self.layers = nn.ModuleList()
        codings = (self.pcoding_K, self.pcoding_V, self.tcoding_K, self.tcoding_V)
        for _ in range(FLAGS.num_blocks):
            self.layers.append(_Block(C_, self.num_heads, self.attention_probs_dropout_rate, self.l2_reg, codings, gen))
        self.out_ln = _LayerNorm(C_)
        self._metrics = None
        self._finish_pad(gen)

    def l2_param_names(self):
        return ["item_embs.lookup_table", "pcoding_K.pembs.lookup_table", "pcoding_V.pembs.lookup_table",
                "tcoding_K.pembs.lookup_table", "tcoding_V.pembs.lookup_table"]

    def finalize(self, device):
        Sequential.finalize(self, device)
        for blk in self.layers:
            blk.attention.compute = self.compute
            blk.attention.time_scale = self.time_scale
            blk.attention.timelen = self.timelen
        return self

    def check_inputs(self) -> None:
        """The interval buckets clip negative differences exactly as the reference does: nothing to check."""

    def _tf_map(self):
        C_ = self.num_units
        m = {"TiSASRec/item_embs/lookup_table": (self.item_embs.lookup_table, None),
             "TiSASRec/output_bias": (self.output_bias, None),
             "out_ln/LayerNorm/gamma": (self.out_ln.gamma, None), "out_ln/LayerNorm/beta": (self.out_ln.beta, None)}
        for n in ("pcoding_K", "pcoding_V", "tcoding_K", "tcoding_V"):
            m[f"TiSASRec/{n}/embedding/lookup_table"] = (getattr(self, n).pembs.lookup_table, None)
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

"""Mirror of the reference's src/module/coding.py operator classes (the ones on the EasyDGL path).

Same class names, constructor arguments and ``__call__`` / ``code`` methods as the reference.  On the model's hot path
the arithmetic of Embedding / PositionCoding / TimeSinusoidCoding is fused into ONE kernel (``edgl_encode_fwd``, see
``easydgl_amd/model/easydgl.py``) and the time-function / interval codings are folded into the attention kernels
(``csrc/k_tattn.hip``); called on their own, the methods below run the stand-alone HIP entry points of
``csrc/k_coding.hip`` (``edgl_embedding_fwd/bwd``, ``edgl_time_sinusoid``, ``edgl_time_function_fwd/bwd``).  ``compute`` is
the hook the owning model installs to hand out the bf16 shadow of a parameter (identity in f32 mode)."""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn

from .. import ops


def glorot_uniform_(t: torch.Tensor, gen: torch.Generator) -> torch.Tensor:
    """tf.glorot_uniform_initializer (default of tf.get_variable / tf.layers.dense): U(-l, l),
    l = sqrt(6 / (fan_in + fan_out)), fan_in = shape[0], fan_out = shape[1]."""
    lim = math.sqrt(6.0 / (t.shape[0] + t.shape[1]))
    with torch.no_grad():
        t.copy_((torch.rand(t.shape, generator=gen) * 2 - 1) * lim)
    return t


class Embedding(nn.Module):
    """coding.py:45-64.  ``lookup_table`` is the RAW variable (l2-regularised incl. row 0, coding.py:53-55);
    with ``zero_pad`` row 0 acts as a zero constant wherever the table is used (coding.py:56-57)."""

    def __init__(self, vocab_size, num_units, l2_reg=0.0, zero_pad=True, scale=True, gen=None):
        super().__init__()
        self.num_units, self.l2_reg, self.zero_pad, self.scale = num_units, l2_reg, zero_pad, scale
        self.lookup_table = nn.Parameter(glorot_uniform_(torch.empty(vocab_size, num_units), gen))
        self.compute = lambda p: p

    def forward(self, inputs: torch.Tensor) -> torch.Tensor:
        """coding.py:60-64: ``lookup(table, inputs) * sqrt(num_units)`` (integer ``inputs`` of any shape)."""
        tab = self.lookup_table
        scale = float(self.num_units) ** 0.5 if self.scale else 1.0
        return ops.EmbeddingFn.apply(tab, self.compute(tab), inputs.to(torch.int64), self.zero_pad, scale)


class PositionCoding(nn.Module):
    """coding.py:67-79."""

    def __init__(self, vocab_size, num_units, l2_reg=0.0, gen=None):
        super().__init__()
        self.pembs = Embedding(vocab_size, num_units, l2_reg, zero_pad=False, scale=False, gen=gen)

    def code(self, inputs: torch.Tensor) -> torch.Tensor:
        """coding.py:76-79: the first ``inputs.shape[1]`` table rows, one copy per batch row -> [B, T, C]."""
        B, T = inputs.shape[0], inputs.shape[1]
        pos = torch.arange(T, device=inputs.device, dtype=torch.int64).unsqueeze(0).expand(B, T)
        return self.pembs(pos)

    def forward(self, inputs: torch.Tensor, **kwargs) -> torch.Tensor:
        """coding.py:71-73: ``concat([inputs, code(inputs)], -1)``."""
        code = self.code(inputs)
        return torch.cat([inputs.to(code.dtype), code], dim=-1)


class TimeSinusoidCoding(nn.Module):
    """coding.py:125-149.  ``scale`` = float32(10000^(2j/C)) computed in float64 exactly as the reference."""

    def __init__(self, num_units):
        super().__init__()
        self.num_units = num_units
        scale = np.power(10000, np.arange(0, num_units, 2) * 1.0 / num_units).astype(np.float32)
        self.register_buffer("scale", torch.from_numpy(scale), persistent=False)
        self.act_dtype = torch.float32

    def code(self, inputs: torch.Tensor) -> torch.Tensor:
        """coding.py:137-149: ``inputs`` [B, T] -> [B, T, C] with sin on the even and cos on the odd channels."""
        assert inputs.dim() == 2, "the tensor rank should be 2."           # coding.py:139
        return ops.time_sinusoid(inputs, self.scale, self.num_units, self.act_dtype)


class TimeFunctionCoding(nn.Module):
    """coding.py:104-122 (TGAT): ``basis_freq`` [C] = linspace(0, 9, C), ``phase`` [C] zeros; code(dt) = cos(dt * basis_freq +
    phase).  On the attention path the [B,T,T,C] code tensor is never built (csrc/k_tattn.hip folds it into the operands)."""

    def __init__(self, num_units):
        super().__init__()
        self.num_units = num_units
        self.basis_freq = nn.Parameter(torch.from_numpy(np.linspace(0, 9, num_units).astype(np.float32)))
        self.phase = nn.Parameter(torch.zeros(num_units))
        self.act_dtype = torch.float32

    def code(self, inputs: torch.Tensor) -> torch.Tensor:
        """coding.py:113-122: ``inputs`` [B, T, ...] -> [B, T, K, C] = cos(inputs * basis_freq + phase), K = the flattened
        trailing axes (1 for a rank-2 input)."""
        B, T = inputs.shape[0], inputs.shape[1]
        x = inputs.reshape(B, T, -1)
        return ops.TimeFunctionFn.apply(x, self.basis_freq, self.phase, self.act_dtype)


class TimeIntervalCoding(nn.Module):
    """coding.py:82-94 (TiSASRec): an embedding table indexed by the clipped integer interval."""

    def __init__(self, vocab_size, num_units, l2_reg=0.0, gen=None):
        super().__init__()
        self.pembs = Embedding(vocab_size, num_units, l2_reg, zero_pad=False, scale=False, gen=gen)

    def code(self, inputs: torch.Tensor) -> torch.Tensor:
        """coding.py:93-94: the table row of every integer interval (an index past the table reads zeros, as the
        reference's GPU lookup does for ``timelen``, TiSASREC.py:59)."""
        return self.pembs(inputs)

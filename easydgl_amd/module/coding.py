"""Mirror of the reference's src/module/coding.py operator classes (the ones on the EasyDGL path).

These are parameter containers + thin callables; the arithmetic of Embedding / PositionCoding /
TimeSinusoidCoding on the model's hot path is fused into ONE kernel (edgl_encode_fwd, see
easydgl_amd/model/easydgl.py), so the per-class ``__call__``/``code`` methods below exist for API
parity and run the same kernel on a degenerate input."""
from __future__ import annotations

import math

import numpy as np
import torch
from torch import nn


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


class PositionCoding(nn.Module):
    """coding.py:67-79."""

    def __init__(self, vocab_size, num_units, l2_reg=0.0, gen=None):
        super().__init__()
        self.pembs = Embedding(vocab_size, num_units, l2_reg, zero_pad=False, scale=False, gen=gen)


class TimeSinusoidCoding(nn.Module):
    """coding.py:125-149.  ``scale`` = float32(10000^(2j/C)) computed in float64 exactly as the reference."""

    def __init__(self, num_units):
        super().__init__()
        self.num_units = num_units
        scale = np.power(10000, np.arange(0, num_units, 2) * 1.0 / num_units).astype(np.float32)
        self.register_buffer("scale", torch.from_numpy(scale), persistent=False)


class TimeFunctionCoding(nn.Module):
    """coding.py:104-122 (TGAT): ``basis_freq`` [C] = linspace(0, 9, C), ``phase`` [C] zeros; code(dt) = cos(dt * basis_freq +
    phase).  On the attention path the [B,T,T,C] code tensor is never built (csrc/k_tattn.hip folds it into the operands)."""

    def __init__(self, num_units):
        super().__init__()
        self.num_units = num_units
        self.basis_freq = nn.Parameter(torch.from_numpy(np.linspace(0, 9, num_units).astype(np.float32)))
        self.phase = nn.Parameter(torch.zeros(num_units))


class TimeIntervalCoding(nn.Module):
    """coding.py:82-94 (TiSASRec): an embedding table indexed by the clipped integer interval."""

    def __init__(self, vocab_size, num_units, l2_reg=0.0, gen=None):
        super().__init__()
        self.pembs = Embedding(vocab_size, num_units, l2_reg, zero_pad=False, scale=False, gen=gen)

"""TEST INFRASTRUCTURE — CPU restatement (torch, float64, autograd) of the reference's CTSMA model
(src/model/CTSMA.py:22-127, the causal MAU of src/module/temporal.py:335-390, FeedForward of
src/model/Base.py:70-87, the regressive batch layout of src/dataloader.py:88-108).  PARITY UNPINNED: the reference
ships no tests or golden vectors and TensorFlow cannot be imported here; the restatement follows the source line by
line (cited below) and is pinned by the known-answer / finite-difference tests in tests/test_ctsma_oracle.py.
Only tests/ may import this module.

Conventions (CTSMA.py:23-31): the model keeps FLAGS.seqslen = T positions and a [num_items, C] item table (no "+1" as
in EasyDGL); features are seqs_i = tokens[:-1] [B,T], seqs_t [B,T+1]; labels = tokens[1:] (train) / tokens (eval).
"""
from __future__ import annotations

from typing import Dict, List

import numpy as np
import torch

from . import easydgl_oracle as O
from . import torch_ref as R


def init_params(num_items: int, T: int, C: int, h: int, E: int, num_blocks: int, rng: np.random.Generator) -> Dict[str, np.ndarray]:
    """Variables of CTSMA.__init__/__call__ with the reference initialisers (glorot_uniform for tf.layers.dense /
    Conv1D kernels and tables, zeros for biases, ones/zeros for LayerNorm)."""
    dh = C // h
    g = O.glorot_uniform
    p = {"CSTMA/item_embs/lookup_table": g(rng, (num_items, C)),
         "CSTMA/spatial_embs/embedding/lookup_table": g(rng, (T, C)),
         "CSTMA/output_bias": np.zeros(num_items - 1)}
    for i in range(num_blocks):
        cin = 2 * C if i == 0 else C
        pre = f"num_blocks_{i}/"
        p[pre + "attention/LayerNorm/gamma"] = np.ones(cin)
        p[pre + "attention/LayerNorm/beta"] = np.zeros(cin)
        for nm in ("dense", "dense_1", "dense_2", "dense_3"):   # Q, K, V, T_ (temporal.py:352-355)
            p[pre + f"attention/modulating_attention/{nm}/kernel"] = g(rng, (cin, C))
            p[pre + f"attention/modulating_attention/{nm}/bias"] = np.zeros(C)
        st = pre + "attention/modulating_attention/sequential_temporal_combined/"
        p[st + "dense/kernel"] = g(rng, (dh + 1, dh * E))
        p[st + "dense/bias"] = np.zeros(dh * E)
        p[st + "weight"] = g(rng, (E, dh))
        p[st + "scaling"] = np.zeros(E)
        p[pre + "feed-forward/LayerNorm/gamma"] = np.ones(C)
        p[pre + "feed-forward/LayerNorm/beta"] = np.zeros(C)
        p[pre + "feed-forward/Inner/kernel"] = g(rng, (C, C))
        p[pre + "feed-forward/Inner/bias"] = np.zeros(C)
        p[pre + "feed-forward/Readout/kernel"] = g(rng, (C, C))
        p[pre + "feed-forward/Readout/bias"] = np.zeros(C)
    p["outln/LayerNorm/gamma"] = np.ones(C)
    p["outln/LayerNorm/beta"] = np.zeros(C)
    return p


def encoder(p, mark_table, seqs_i, seqs_t, C: int, h: int, num_blocks: int, time_scale: float, dtype=torch.float64):
    """CTSMA.__call__ up to the final LayerNorm (CTSMA.py:48-80), dropout off.  Returns (out [B,T,C], [lam])."""
    ids = torch.as_tensor(np.asarray(seqs_i), dtype=torch.long)
    B, T = ids.shape
    ts = torch.tensor((np.asarray(seqs_t, dtype=np.float32) / np.float32(time_scale)).astype(np.float64), dtype=dtype)  # :50
    spans = ts[:, 1:] - ts[:, :-1]                                                  # :51
    marks = torch.tensor(np.asarray(mark_table)[np.asarray(seqs_i)], dtype=dtype)   # :54
    x = R.zero_padded(p["CSTMA/item_embs/lookup_table"])[ids] * (C ** 0.5)          # :55, coding.py:60-64
    pos = p["CSTMA/spatial_embs/embedding/lookup_table"][:T].unsqueeze(0).expand(B, T, C)
    x = torch.cat([x, pos], dim=-1)                                                 # :56, PositionCoding.__call__ coding.py:72-74
    keymask3 = (ids != 0).to(dtype).unsqueeze(1).repeat(h, T, 1)                    # :61-62
    lams: List[torch.Tensor] = []
    out = x
    for i in range(num_blocks):
        pre = f"num_blocks_{i}/"
        a = pre + "attention/modulating_attention/"
        q_in = R.layernorm(out, p[pre + "attention/LayerNorm/gamma"], p[pre + "attention/LayerNorm/beta"])   # :70
        Q = q_in @ p[a + "dense/kernel"] + p[a + "dense/bias"]                       # temporal.py:352
        K = out @ p[a + "dense_1/kernel"] + p[a + "dense_1/bias"]                    # :353
        V = out @ p[a + "dense_2/kernel"] + p[a + "dense_2/bias"]                    # :354
        T_ = out @ p[a + "dense_3/kernel"] + p[a + "dense_3/bias"]                   # :355
        pm = {"sequential_temporal_combined/" + k: p[a + "sequential_temporal_combined/" + k]
              for k in ("dense/kernel", "dense/bias", "weight", "scaling")}
        att, lam = R.bimau(C, h, None, keymask3, spans, marks, pm, "", 0.0, False, causal=True, set_diag=False,
                           qkvt=torch.cat([Q, K, V, T_], dim=-1), resid=q_in[:, :, :C])   # :357-383
        y = R.layernorm(att, p[pre + "feed-forward/LayerNorm/gamma"], p[pre + "feed-forward/LayerNorm/beta"])  # :75
        inner = torch.relu(y @ p[pre + "feed-forward/Inner/kernel"] + p[pre + "feed-forward/Inner/bias"])     # Base.py:73,79
        out = inner @ p[pre + "feed-forward/Readout/kernel"] + p[pre + "feed-forward/Readout/bias"] + y       # Base.py:74,82-86
        lams.append(lam)
    out = R.layernorm(out, p["outln/LayerNorm/gamma"], p["outln/LayerNorm/beta"])    # :82-83
    return out, lams


def logits_from(p, rows):
    """CTSMA.py:91-93: rows @ lookup_table^T + concat([-1000], output_bias) (Base.py:106-110, inf_pad).
    `self.item_embs.lookup_table` is the zero-padded tensor (coding.py:56-58), as in EasyDGL."""
    bias = torch.cat([torch.full((1,), -1000.0, dtype=rows.dtype), p["CSTMA/output_bias"]])
    return rows @ R.zero_padded(p["CSTMA/item_embs/lookup_table"]).t() + bias


def train_loss(p, mark_table, features, labels, C: int, h: int, num_blocks: int, time_scale: float, ct_reg: float,
               l2_reg: float, dtype=torch.float64):
    """CTSMA.train (CTSMA.py:95-127)."""
    out, lams = encoder(p, mark_table, features["seqs_i"], features["seqs_t"], C, h, num_blocks, time_scale, dtype)
    B, T, _ = out.shape
    logits = logits_from(p, out.reshape(B * T, C))                                   # :86,91-93
    lp = torch.log(torch.softmax(logits, -1) + 1e-5)                                 # :97
    reg = torch.zeros((), dtype=dtype)
    if l2_reg != 0.0:                                                               # coding.py:34-40 on both tables
        for k in ("CSTMA/item_embs/lookup_table", "CSTMA/spatial_embs/embedding/lookup_table"):
            reg = reg + l2_reg * 0.5 * (p[k] ** 2).sum()
    if ct_reg != 0.0:                                                               # :101-112
        raw = torch.tensor(np.asarray(features["seqs_t"], dtype=np.float32).astype(np.float64), dtype=dtype)
        sp = raw[:, 1:] - raw[:, :-1]
        nm = torch.tensor(np.asarray(mark_table)[np.asarray(labels)], dtype=dtype)
        if h != 1:
            sp, nm = sp.repeat(h, 1), nm.repeat(h, 1, 1)
        for lam in lams:
            reg = reg + ct_reg * R.biased_likelihood(lam, nm, sp)
    lab = torch.as_tensor(np.asarray(labels).reshape(-1), dtype=torch.long)
    w = (lab != 0).to(dtype)
    per = -lp[torch.arange(lab.shape[0]), lab]                                       # :115-119
    ce = (w * per).sum() / (w.sum() + 1e-5)                                          # :120-122
    return ce + reg, dict(ce=ce, reg=reg, logits=logits, lams=lams, out=out)


def eval_logits(p, mark_table, features, C: int, h: int, num_blocks: int, time_scale: float, dtype=torch.float64):
    """CTSMA.__call__(is_training=False): logits of the last position (CTSMA.py:87-93)."""
    out, _ = encoder(p, mark_table, features["seqs_i"], features["seqs_t"], C, h, num_blocks, time_scale, dtype)
    return logits_from(p, out[:, -1])

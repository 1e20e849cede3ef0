"""TEST INFRASTRUCTURE — CPU restatement (torch, float64, autograd) of the reference's two baseline models that BASELINE.json
config 5 runs through the same attention path:

  TGAT      src/model/TGAT.py:20-83, TfMultiHeadAttention src/module/temporal.py:126-184, TimeFunctionCoding coding.py:104-122
  TiSASRec  src/model/TiSASREC.py:20-88, TiMultiHeadAttention temporal.py:36-105, TimeIntervalCoding coding.py:82-94

plus FeedForward (src/model/Base.py:70-87), the all-position loss of Sequential.train (Base.py:119-140) and the regressive
batch layout (src/dataloader.py:95-108).  PARITY UNPINNED: the reference ships no tests or golden vectors and TensorFlow
cannot be imported here; the restatement follows the source line by line — it materialises the [B,T,T,C] time tensors exactly
as the reference does — and is pinned by the known-answer tests in tests/test_baselines_oracle.py.  Only tests/ may import it.

Conventions (TGAT.py:44-46): features seqs_i = tokens[:-1] [B,T] with T = FLAGS.seqslen, seqs_t [B,T+1]; labels = tokens[1:].
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import easydgl_oracle as O
from . import torch_ref as R

PAD = float(np.float32(-2 ** 32 + 1))


def _split_heads(x, h):          # tf.concat(tf.split(x, h, axis=-1), axis=0)
    return torch.cat(torch.split(x, x.shape[-1] // h, dim=-1), dim=0)


def _merge_heads(x, h):          # tf.concat(tf.split(x, h, axis=0), axis=2)
    return torch.cat(torch.split(x, x.shape[0] // h, dim=0), dim=2)


def _feed_forward(p, pre, y):
    """Base.FeedForward([C, C]) (Base.py:70-87), dropout off.  tf.layers.Conv1D creates its variables at the first call, i.e.
    under the caller's scope: num_blocks_i/feedforward/{Inner,Readout}/{kernel,bias}."""
    inner = torch.relu(y @ p[pre + "feedforward/Inner/kernel"] + p[pre + "feedforward/Inner/bias"])
    return inner @ p[pre + "feedforward/Readout/kernel"] + p[pre + "feedforward/Readout/bias"] + y


def _masked_softmax(S, keys, causal=True):
    """temporal.py:153-169: key mask from all-zero key rows, future blinding, softmax."""
    hN, T, _ = S.shape
    h = hN // keys.shape[0]
    key_masks = torch.sign(keys.abs().sum(-1)).repeat(h, 1).unsqueeze(1).expand(hN, T, T)
    S = torch.where(key_masks == 0, torch.full_like(S, PAD), S)
    if causal:
        tril = torch.tril(torch.ones(T, T, dtype=S.dtype)).unsqueeze(0).expand(hN, T, T)
        S = torch.where(tril == 0, torch.full_like(S, PAD), S)
    return torch.softmax(S, dim=-1)


# =====================================================================================================================
# TGAT
# =====================================================================================================================
def tgat_init_params(num_items: int, T: int, C: int, nb: int, rng: np.random.Generator) -> Dict[str, np.ndarray]:
    g = O.glorot_uniform
    p = {"TGAT/item_embs/lookup_table": g(rng, (num_items, C)),
         "TGAT/pcoding_K/embedding/lookup_table": g(rng, (T, C)),
         "TGAT/tcoding_K/basis_freq": np.linspace(0, 9, C).astype(np.float32).astype(np.float64),   # coding.py:115-117
         "TGAT/tcoding_K/phase": np.zeros(C),
         "TGAT/output_bias": np.zeros(num_items - 1)}
    for i in range(nb):
        pre = f"num_blocks_{i}/"
        p[pre + "attention/LayerNorm/gamma"] = np.ones(C)
        p[pre + "attention/LayerNorm/beta"] = np.zeros(C)
        for nm in ("dense", "dense_1", "dense_2"):                    # Q, K, V (temporal.py:134-136)
            p[pre + f"attention/attention/timeinterval/{nm}/kernel"] = g(rng, (C, C))
            p[pre + f"attention/attention/timeinterval/{nm}/bias"] = np.zeros(C)
        p[pre + "feedforward/LayerNorm/gamma"] = np.ones(C)
        p[pre + "feedforward/LayerNorm/beta"] = np.zeros(C)
        for nm in ("Inner", "Readout"):
            p[pre + f"feedforward/{nm}/kernel"] = g(rng, (C, C))
            p[pre + f"feedforward/{nm}/bias"] = np.zeros(C)
    p["out_ln/LayerNorm/gamma"] = np.ones(C)
    p["out_ln/LayerNorm/beta"] = np.zeros(C)
    return p


def tf_attention(p, a, queries, keys, intervals, pos_tab, omega, phi, h):
    """TfMultiHeadAttention.__call__ (temporal.py:126-184), causality=True, dropout off."""
    C = queries.shape[-1]
    B, T, _ = queries.shape
    Q = queries @ p[a + "dense/kernel"] + p[a + "dense/bias"]                       # :134
    K = keys @ p[a + "dense_1/kernel"] + p[a + "dense_1/bias"]                      # :135
    V = keys @ p[a + "dense_2/kernel"] + p[a + "dense_2/bias"]                      # :136
    Q_, K_, V_ = _split_heads(Q, h), _split_heads(K, h), _split_heads(V, h)
    Kp = _split_heads(pos_tab[:T].unsqueeze(0).expand(B, T, C), h)                  # :143, coding.py:76-79
    tcode = torch.cos(intervals.unsqueeze(-1) * omega + phi)                        # coding.py:113-121 -> [B,T,T,C]
    Kt = torch.cat(torch.split(tcode, C // h, dim=3), dim=0)                        # :144 -> [hB,T,T,dh]
    S = Q_ @ K_.transpose(1, 2) + Q_ @ Kp.transpose(1, 2) + (Kt @ Q_.unsqueeze(3)).squeeze(3)   # :147-150
    S = S / (K_.shape[-1] ** 0.5)                                                   # :153
    P = _masked_softmax(S, keys)                                                    # :156-172
    return _merge_heads(P @ V_, h) + queries                                        # :178-184


def tgat_encoder(p, seqs_i, seqs_t, C: int, h: int, nb: int, time_scale: float, dtype=torch.float64):
    """TGAT.__call__ up to the final LayerNorm (TGAT.py:44-72), dropout off."""
    ids = torch.as_tensor(np.asarray(seqs_i), dtype=torch.long)
    ts = torch.tensor((np.asarray(seqs_t, dtype=np.float32) / np.float32(time_scale)).astype(np.float64), dtype=dtype)   # :46
    x = R.zero_padded(p["TGAT/item_embs/lookup_table"])[ids] * (C ** 0.5)                  # :49
    spans = torch.clamp(ts[:, 1:].unsqueeze(2) - ts[:, :-1].unsqueeze(1), min=0.0)        # :51-54  [B,T(q),T(k)]
    masks = (ids != 0).to(dtype).unsqueeze(-1)                                             # :59
    out = x * masks                                                                        # :62
    for i in range(nb):
        pre = f"num_blocks_{i}/"
        qn = R.layernorm(out, p[pre + "attention/LayerNorm/gamma"], p[pre + "attention/LayerNorm/beta"])
        out = tf_attention(p, pre + "attention/attention/timeinterval/", qn, out, spans,
                           p["TGAT/pcoding_K/embedding/lookup_table"], p["TGAT/tcoding_K/basis_freq"],
                           p["TGAT/tcoding_K/phase"], h)                                   # :66
        y = R.layernorm(out, p[pre + "feedforward/LayerNorm/gamma"], p[pre + "feedforward/LayerNorm/beta"])
        out = _feed_forward(p, pre, y) * masks                                             # :69-70
    return R.layernorm(out, p["out_ln/LayerNorm/gamma"], p["out_ln/LayerNorm/beta"])      # :72-73


def _logits(p, table_key, bias_key, rows):
    bias = torch.cat([torch.full((1,), -1000.0, dtype=rows.dtype), p[bias_key]])   # Base.py:106-110
    return rows @ R.zero_padded(p[table_key]).t() + bias


def _all_position_loss(logits, labels, dtype):
    """Sequential.train (Base.py:119-131)."""
    lp = torch.log(torch.softmax(logits, -1) + 1e-5)
    lab = torch.as_tensor(np.asarray(labels).reshape(-1), dtype=torch.long)
    w = (lab != 0).to(dtype)
    per = -lp[torch.arange(lab.shape[0]), lab]
    return (w * per).sum() / (w.sum() + 1e-5)


def tgat_train_loss(p, features, labels, C, h, nb, time_scale, l2_reg, dtype=torch.float64):
    out = tgat_encoder(p, features["seqs_i"], features["seqs_t"], C, h, nb, time_scale, dtype)
    B, T, _ = out.shape
    logits = _logits(p, "TGAT/item_embs/lookup_table", "TGAT/output_bias", out.reshape(B * T, C))   # TGAT.py:74-83
    loss = _all_position_loss(logits, labels, dtype)
    if l2_reg != 0.0:   # tf.losses.get_regularization_loss(): the two tables built with a regulariser (TGAT.py:27-30)
        for k in ("TGAT/item_embs/lookup_table", "TGAT/pcoding_K/embedding/lookup_table"):
            loss = loss + l2_reg * 0.5 * (p[k] ** 2).sum()
    return loss, dict(logits=logits, out=out)


def tgat_eval_logits(p, features, C, h, nb, time_scale, dtype=torch.float64):
    out = tgat_encoder(p, features["seqs_i"], features["seqs_t"], C, h, nb, time_scale, dtype)
    return _logits(p, "TGAT/item_embs/lookup_table", "TGAT/output_bias", out[:, -1])


# =====================================================================================================================
# TiSASRec
# =====================================================================================================================
def tisasrec_init_params(num_items: int, timelen: int, C: int, nb: int, rng: np.random.Generator) -> Dict[str, np.ndarray]:
    g = O.glorot_uniform
    p = {"TiSASRec/item_embs/lookup_table": g(rng, (num_items, C)), "TiSASRec/output_bias": np.zeros(num_items - 1)}
    for nm in ("pcoding_K", "pcoding_V", "tcoding_K", "tcoding_V"):               # TiSASREC.py:30-33: all [timelen, C]
        p[f"TiSASRec/{nm}/embedding/lookup_table"] = g(rng, (timelen, C))
    for i in range(nb):
        pre = f"num_blocks_{i}/"
        p[pre + "attention/LayerNorm/gamma"] = np.ones(C)
        p[pre + "attention/LayerNorm/beta"] = np.zeros(C)
        for nm in ("dense", "dense_1", "dense_2"):
            p[pre + f"attention/attention/timeinterval/{nm}/kernel"] = g(rng, (C, C))
            p[pre + f"attention/attention/timeinterval/{nm}/bias"] = np.zeros(C)
        p[pre + "feedforward/LayerNorm/gamma"] = np.ones(C)
        p[pre + "feedforward/LayerNorm/beta"] = np.zeros(C)
        for nm in ("Inner", "Readout"):
            p[pre + f"feedforward/{nm}/kernel"] = g(rng, (C, C))
            p[pre + f"feedforward/{nm}/bias"] = np.zeros(C)
    p["out_ln/LayerNorm/gamma"] = np.ones(C)
    p["out_ln/LayerNorm/beta"] = np.zeros(C)
    return p


def _lookup_or_zero(tab, idx):
    """tf.nn.embedding_lookup on the GPU kernel returns zeros for an out-of-range index; `clip(.., 0, timelen)` can produce
    the index timelen for a [timelen, C] table (TiSASREC.py:62 with coding.py:89-94) — stated quirk, kept."""
    n = tab.shape[0]
    ext = torch.cat([tab, torch.zeros(1, tab.shape[1], dtype=tab.dtype)], 0)
    return ext[torch.clamp(idx, max=n)]


def ti_attention(p, a, queries, keys, intervals, tabs, h):
    """TiMultiHeadAttention.__call__ (temporal.py:36-105), causality=True, dropout off."""
    C = queries.shape[-1]
    B, T, _ = queries.shape
    Q = queries @ p[a + "dense/kernel"] + p[a + "dense/bias"]
    K = keys @ p[a + "dense_1/kernel"] + p[a + "dense_1/bias"]
    V = keys @ p[a + "dense_2/kernel"] + p[a + "dense_2/bias"]
    Q_, K_, V_ = _split_heads(Q, h), _split_heads(K, h), _split_heads(V, h)
    Kp = _split_heads(tabs["pK"][:T].unsqueeze(0).expand(B, T, C), h)              # :50
    Vp = _split_heads(tabs["pV"][:T].unsqueeze(0).expand(B, T, C), h)              # :51
    Kt = torch.cat(torch.split(_lookup_or_zero(tabs["tK"], intervals), C // h, dim=3), dim=0)   # :52  [hB,T,T,dh]
    Vt = torch.cat(torch.split(_lookup_or_zero(tabs["tV"], intervals), C // h, dim=3), dim=0)   # :53
    S = Q_ @ K_.transpose(1, 2) + Q_ @ Kp.transpose(1, 2) + (Kt @ Q_.unsqueeze(3)).squeeze(3)   # :56-59
    S = S / (K_.shape[-1] ** 0.5)
    P = _masked_softmax(S, keys)                                                   # :64-81
    qmask = torch.sign(queries.abs().sum(-1)).repeat(h, 1).unsqueeze(-1)           # :84-87
    P = P * qmask
    out = P @ V_ + P @ Vp + (P.unsqueeze(2) @ Vt).squeeze(2)                       # :93-96
    return _merge_heads(out, h) + queries                                          # :99-104


def tisasrec_encoder(p, seqs_i, seqs_t, C, h, nb, time_scale, timelen, dtype=torch.float64):
    """TiSASRec.__call__ up to the final LayerNorm (TiSASREC.py:47-77), dropout off."""
    ids = torch.as_tensor(np.asarray(seqs_i), dtype=torch.long)
    ts32 = np.asarray(seqs_t, dtype=np.float32) / np.float32(time_scale)                                    # :49
    d32 = np.clip(ts32[:, 1:, None] - ts32[:, None, :-1], np.float32(0), np.float32(timelen))               # :58-62 (float32)
    intervals = torch.as_tensor(d32.astype(np.int64))                                                       # tf.to_int64: truncation
    x = R.zero_padded(p["TiSASRec/item_embs/lookup_table"])[ids] * (C ** 0.5)
    masks = (ids != 0).to(dtype).unsqueeze(-1)
    out = x * masks
    tabs = {k: p[f"TiSASRec/{n}/embedding/lookup_table"] for k, n in
            (("pK", "pcoding_K"), ("pV", "pcoding_V"), ("tK", "tcoding_K"), ("tV", "tcoding_V"))}
    for i in range(nb):
        pre = f"num_blocks_{i}/"
        qn = R.layernorm(out, p[pre + "attention/LayerNorm/gamma"], p[pre + "attention/LayerNorm/beta"])
        out = ti_attention(p, pre + "attention/attention/timeinterval/", qn, out, intervals, tabs, h)      # :70-71
        y = R.layernorm(out, p[pre + "feedforward/LayerNorm/gamma"], p[pre + "feedforward/LayerNorm/beta"])
        out = _feed_forward(p, pre, y) * masks                                                              # :73-75
    return R.layernorm(out, p["out_ln/LayerNorm/gamma"], p["out_ln/LayerNorm/beta"])


def tisasrec_train_loss(p, features, labels, C, h, nb, time_scale, timelen, l2_reg, dtype=torch.float64):
    out = tisasrec_encoder(p, features["seqs_i"], features["seqs_t"], C, h, nb, time_scale, timelen, dtype)
    B, T, _ = out.shape
    logits = _logits(p, "TiSASRec/item_embs/lookup_table", "TiSASRec/output_bias", out.reshape(B * T, C))
    loss = _all_position_loss(logits, labels, dtype)
    if l2_reg != 0.0:   # all five tables carry the regulariser (TiSASREC.py:27-33)
        for k in p:
            if k.endswith("lookup_table"):
                loss = loss + l2_reg * 0.5 * (p[k] ** 2).sum()
    return loss, dict(logits=logits, out=out)


def tisasrec_eval_logits(p, features, C, h, nb, time_scale, timelen, dtype=torch.float64):
    out = tisasrec_encoder(p, features["seqs_i"], features["seqs_t"], C, h, nb, time_scale, timelen, dtype)
    return _logits(p, "TiSASRec/item_embs/lookup_table", "TiSASRec/output_bias", out[:, -1])

"""fp64 numpy ORACLE for the EasyDGL self-modulating-attention hot path.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it; nothing under ``easydgl_amd/`` does.

PARITY UNPINNED: the reference (cchao0116/EasyDGL) ships no tests, golden
vectors or fixtures, and its arithmetic lives in TensorFlow-1.x graph ops
(``tensorflow-gpu==2.3.4`` per requirements.txt:4 / TF 1.15.3 per the paper
supplement) which is not importable offline.  This file therefore restates the
reference algorithm from source, line by line, in float64 numpy, following the
TF op semantics listed in SURVEY.md Appendix A; it is pinned only by the
known-answer tests derivable from the source (tests/test_oracle_kat.py) and by
a finite-difference check of its own gradients (tests/test_oracle_grad.py).

Every function cites the reference file:line it follows (paths relative to the
reference repo root).  Arithmetic is float64 except where the reference's
*inputs* are float32 by construction (timestamps / time codes): those
quantisation points are reproduced explicitly so that a float32 device kernel
that performs the same IEEE operations sees bit-identical arguments.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import numpy as np
from scipy.special import erf as _erf

F64 = np.float64
F32 = np.float32

PAD_SCORE = float(np.float32(-2 ** 32 + 1))  # temporal.py:425 -> rounds to -4294967296.0 in f32


# --------------------------------------------------------------------------------------
# configuration
# --------------------------------------------------------------------------------------
@dataclass
class Config:
    """Mirror of the FLAGS the model reads (src/main.py:22-75, Base.py:92-104, EasyDGL.py:37-47).

    ``num_items`` / ``seqslen`` are the *FLAGS* values; the model adds one to each
    (EasyDGL.py:40-41): T = seqslen + 1 positions, I = num_items + 1 table rows,
    MASK token id = num_items (EasyDGL.py:39).
    """
    num_items: int
    seqslen: int
    num_units: int
    num_heads: int = 1
    num_blocks: int = 1
    masklen: int = 6
    time_scale: float = 1.0
    ct_reg: float = 0.0
    l2_reg: float = 0.0
    learning_rate: float = 5e-4
    num_events: int = 1

    @property
    def T(self) -> int:
        return self.seqslen + 1

    @property
    def I(self) -> int:
        return self.num_items + 1

    @property
    def mask_id(self) -> int:
        return self.num_items

    @property
    def dh(self) -> int:
        return self.num_units // self.num_heads


# --------------------------------------------------------------------------------------
# parameters (SURVEY.md §8 a-P): same variable names as the reference's scopes
# --------------------------------------------------------------------------------------
def glorot_uniform(rng: np.random.Generator, shape) -> np.ndarray:
    """tf.glorot_uniform_initializer: U(-l, l), l = sqrt(6/(fan_in+fan_out)) (2-D: fan_in=shape[0])."""
    fan_in, fan_out = shape[0], shape[1]
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=shape)


def init_params(cfg: Config, rng: np.random.Generator, perturb: bool = False) -> Dict[str, np.ndarray]:
    """Initialise every variable the reference creates for EasyDGL (float64).

    Names follow the TF variable scopes: EasyDGL.py:49-67 (CSTMA/*), :101-139 (layer_i/*,
    cls/predictions/*), temporal.py:289-303,409 (TMAU/*).  With ``perturb`` the zero/one
    initialised variables (biases, LayerNorm, scaling, output_bias) get small random values so
    that parity tests exercise them.
    """
    C, h, E, T, I, dh = cfg.num_units, cfg.num_heads, cfg.num_events, cfg.T, cfg.I, cfg.dh
    p: Dict[str, np.ndarray] = {}
    p["CSTMA/item_embs/lookup_table"] = glorot_uniform(rng, (I, C))
    p["CSTMA/mark_embs/lookup_table"] = glorot_uniform(rng, (E, C))
    p["CSTMA/spatial_embs/embedding/lookup_table"] = glorot_uniform(rng, (T, C))
    p["CSTMA/output_bias"] = np.zeros(I - 1)

    def small(shape, base=0.0):
        return base + (0.1 * rng.standard_normal(shape) if perturb else np.zeros(shape))

    p["CSTMA/output_bias"] = small((I - 1,))
    for i in range(cfg.num_blocks):
        cin = 3 * C if i == 0 else C
        pre = f"layer_{i}/"
        p[pre + "attention/self/TMAU/dense/kernel"] = 0.02 * rng.standard_normal((cin, 4 * C))
        p[pre + "attention/self/TMAU/dense/bias"] = small((4 * C,))
        st = pre + "attention/self/TMAU/sequential_temporal_combined/"
        p[st + "dense/kernel"] = glorot_uniform(rng, (dh + 1, dh * E))
        p[st + "dense/bias"] = small((dh * E,))
        p[st + "weight"] = glorot_uniform(rng, (E, dh))
        p[st + "scaling"] = small((E,))
        p[pre + "attention/output/dense/kernel"] = glorot_uniform(rng, (C, C))
        p[pre + "attention/output/dense/bias"] = small((C,))
        p[pre + "attention/output/LayerNorm/beta"] = small((C,))
        p[pre + "attention/output/LayerNorm/gamma"] = small((C,), 1.0)
        p[pre + "intermediate/dense/kernel"] = glorot_uniform(rng, (C, 2 * C))
        p[pre + "intermediate/dense/bias"] = small((2 * C,))
        p[pre + "output/dense/kernel"] = glorot_uniform(rng, (2 * C, C))
        p[pre + "output/dense/bias"] = small((C,))
        p[pre + "output/LayerNorm/beta"] = small((C,))
        p[pre + "output/LayerNorm/gamma"] = small((C,), 1.0)
    # (EasyDGL.py:138: tf.layers.dense sizes its kernel by the input — the 3C-wide encoder output when there is no block)
    p["cls/predictions/transform/dense/kernel"] = glorot_uniform(rng, (3 * C if cfg.num_blocks == 0 else C, C))
    p["cls/predictions/transform/dense/bias"] = small((C,))
    p["cls/predictions/transform/LayerNorm/beta"] = small((C,))
    p["cls/predictions/transform/LayerNorm/gamma"] = small((C,), 1.0)
    return {k: np.asarray(v, dtype=F64) for k, v in p.items()}


EMBEDDING_TABLES = (  # variables carrying the l2 regulariser (coding.py:48,53-55)
    "CSTMA/item_embs/lookup_table",
    "CSTMA/mark_embs/lookup_table",
    "CSTMA/spatial_embs/embedding/lookup_table",
)


# --------------------------------------------------------------------------------------
# element-wise pieces
# --------------------------------------------------------------------------------------
def clip_by_value(x):
    """EasyDGL.py:15-16 — tf.clip_by_value(x, 0, 100)."""
    return np.clip(x, 0.0, 100.0)


def gelu(x):
    """EasyDGL.py:19-32 — exact erf GELU: x * 0.5 * (1 + erf(x / sqrt(2)))."""
    return x * (0.5 * (1.0 + _erf(x / math.sqrt(2.0))))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def softmax(x, axis=-1):
    """tf.nn.softmax — max-subtracted (Appendix A)."""
    m = np.max(x, axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / np.sum(e, axis=axis, keepdims=True)


def dense(x, kernel, bias, activation=None):
    """tf.layers.dense: act(x @ kernel[in,out] + bias) on the last axis (Appendix A)."""
    y = x @ kernel + bias
    return activation(y) if activation is not None else y


def layernorm(x, gamma, beta, eps=1e-12):
    """Base.py:12-67 — ``begin_norm_axis=1``: moments over ALL axes but the batch axis,
    i.e. jointly over (T, C) per sample; population variance; gamma/beta over the last axis.
    tf.nn.batch_normalization: y = (x - mean) * rsqrt(var + eps) * gamma + beta.
    """
    axes = tuple(range(1, x.ndim))
    mean = x.mean(axis=axes, keepdims=True)
    var = ((x - mean) ** 2).mean(axis=axes, keepdims=True)
    return (x - mean) / np.sqrt(var + eps) * gamma + beta


# --------------------------------------------------------------------------------------
# input encoding (rows a-2, a-3, a-4)
# --------------------------------------------------------------------------------------
def scaled_times(seqs_t, time_scale) -> np.ndarray:
    """EasyDGL.py:71 — ``features['seqs_t'] / self.time_scale`` on a float32 tensor.
    Quantisation point: the division is a float32 IEEE division in the reference."""
    return (np.asarray(seqs_t, dtype=F32) / F32(time_scale)).astype(F32)


def spans_from_times(ts32: np.ndarray) -> np.ndarray:
    """EasyDGL.py:73-74 (also :161-162 on raw seconds): span[t] = clip(ts[t]-ts[t-1], 0, 100),
    span[0] := span[1].  float32 subtraction as in the reference, returned as float64."""
    ts32 = np.asarray(ts32, dtype=F32)
    d = clip_by_value((ts32[:, 1:] - ts32[:, :-1]).astype(F32))
    d = np.concatenate([d[:, :1], d], axis=-1)
    return d.astype(F64)


def time_sinusoid_scale(num_units: int) -> np.ndarray:
    """coding.py:134-135 — 10000^(2j/C), computed in float64 then cast to float32."""
    return np.power(10000.0, np.arange(0, num_units, 2) * 1.0 / num_units).astype(F32)


def time_sinusoid_code(ts32: np.ndarray, num_units: int) -> np.ndarray:
    """coding.py:137-149 — code[..., 2j] = sin(ts/scale_j), code[..., 2j+1] = cos(ts/scale_j).
    The quotient is a float32 division (quantisation point); sin/cos evaluated in float64 on
    that float32 argument."""
    ts32 = np.asarray(ts32, dtype=F32)
    scale = time_sinusoid_scale(num_units)
    x = (ts32[..., None] / scale[None, None, :]).astype(F32).astype(F64)
    code = np.stack([np.sin(x), np.cos(x)], axis=-1)  # [B,T,C/2,2]
    return code.reshape(ts32.shape[0], ts32.shape[1], num_units)


def zero_padded(table: np.ndarray) -> np.ndarray:
    """coding.py:56-57 — row 0 of the *used* table is a zero constant."""
    return np.concatenate([np.zeros((1, table.shape[1])), table[1:]], axis=0)


def mark_rows(cfg: Config, mark_table: np.ndarray, ids: np.ndarray) -> np.ndarray:
    """EasyDGL.py:76-77 — MASK token -> row 0, then gather the integer multi-hot rows."""
    ids = np.asarray(ids)
    ids0 = np.where(ids == cfg.mask_id, 0, ids)
    return np.asarray(mark_table)[ids0]


def input_encode(cfg: Config, params, mark_table, seqs_i, seqs_t):
    """EasyDGL.py:70-95 (dropout = identity).  Returns X0 [B,T,3C], spans [B,T],
    marks [B,T,E] (ints), keymask [B,T] (1 = real key)."""
    C = cfg.num_units
    ids = np.asarray(seqs_i)
    ts = scaled_times(seqs_t, cfg.time_scale)
    spans = spans_from_times(ts)
    marks = mark_rows(cfg, mark_table, ids)
    tcodes = time_sinusoid_code(ts, C)
    item_tab = zero_padded(params["CSTMA/item_embs/lookup_table"])
    x = item_tab[ids] * (C ** 0.5) + tcodes  # coding.py:60-64 (scale=True), EasyDGL.py:83
    T = ids.shape[1]
    pos = np.broadcast_to(params["CSTMA/spatial_embs/embedding/lookup_table"][:T][None], x.shape)  # coding.py:76-79
    # EasyDGL.py:87-88: the 0/1 multi-hot VALUES index the zero-padded mark embedding table
    mk_tab = zero_padded(params["CSTMA/mark_embs/lookup_table"])
    mk = mk_tab[marks].sum(axis=2)
    x0 = np.concatenate([x, pos, mk], axis=-1)
    keymask = (ids != 0).astype(F64)
    return x0, spans, marks, keymask


# --------------------------------------------------------------------------------------
# BiMAU (rows a-5, a-6)
# --------------------------------------------------------------------------------------
def split_heads(x, h):
    """tf.concat(tf.split(x, h, axis=2), axis=0): head-major stacking b' = head*B + b (temporal.py:413-416)."""
    return np.concatenate(np.split(x, h, axis=2), axis=0)


def merge_heads(x, h):
    """tf.concat(tf.split(x, h, axis=0), axis=2) (temporal.py:444)."""
    return np.concatenate(np.split(x, h, axis=0), axis=2)


def intensity(H, intervals, marks, W1, b1, w, scaling, num_heads):
    """MAU.intensity, temporal.py:281-315.

    H [hB,T,dh]; intervals [B,T]; marks [B,T,E] ints.  Returns (Mint [hB,T,T], lam [hB,T,E]).
    Dense output channel j <-> (e = j // dh, u = j % dh) (tf.split(..., E, axis=2), :291).
    """
    hB, T, dh = H.shape
    E = w.shape[0]
    iv = np.tile(intervals[:, :, None], (num_heads, 1, 1))  # :283
    lin = np.concatenate([H, iv], axis=-1)  # :287
    Z = sigmoid(lin @ W1 + b1)  # :289-290
    Z = Z.reshape(hB, T, E, dh)
    z = np.einsum("bteu,eu->bte", Z, w)  # :291-297,305 (matmul with weight)
    s = np.exp(scaling)[None, None, :]  # :301
    lam = s * np.log(1.0 + np.exp(z / s))  # :305-306
    m = np.tile(np.asarray(marks, dtype=F64), (num_heads, 1, 1))  # :311-312
    Mint = np.einsum("bqe,bke->bqk", lam, m)  # :309-313
    return Mint, lam


def bimau(cfg: Config, x, keymask, spans, marks, Wqkvt, bqkvt, W1, b1, w, scaling, causal=False, set_diag=True,
          qkvt=None, resid=None):
    """BiMAU.__call__, temporal.py:404-452 (dropout = identity; no causal mask).
    x [B,T,Cin] -> (out [B,T,C], lam [hB,T,E]).  causal=True / set_diag=False restate MAU.__call__
    (temporal.py:335-390): future blinding (:370-375), modulation kept on the diagonal; qkvt / resid let the caller
    supply the projections (MAU: Q from LN(x), K,V,T_ from x, :352-355) and the residual (:383 adds the queries)."""
    C, h = cfg.num_units, cfg.num_heads
    B, T = x.shape[0], x.shape[1]
    if qkvt is None:
        qkvt = dense(x, Wqkvt, bqkvt)  # :409
    if resid is None:
        resid = x[:, :, :C]
    Q, K, V, T_ = np.split(qkvt, 4, axis=-1)  # :410
    Q_, K_, V_, T__ = (split_heads(a, h) for a in (Q, K, V, T_))  # :413-416
    S = Q_ @ np.transpose(K_, (0, 2, 1))  # :419
    S = S / (K_.shape[-1] ** 0.5)  # :422
    km = np.tile(keymask[:, None, :], (h, T, 1))  # EasyDGL.py:94-95
    S = np.where(km == 0, PAD_SCORE, S)  # :425-426
    if causal:  # :370-375 — tf.where(tril == 0, paddings, outputs)
        S = np.where(np.tril(np.ones((T, T)))[None] == 0, PAD_SCORE, S)
    P = softmax(S)  # :429
    H = P @ T__  # :434
    Mint, lam = intensity(H, spans, marks, W1, b1, w, scaling, h)  # :435
    if set_diag:
        idx = np.arange(T)
        Mint = Mint.copy()
        Mint[:, idx, idx] = 1.0  # :438-439 set_diag
    A = Mint * P  # :441
    O = A @ V_  # :443
    out = merge_heads(O, h)  # :444
    out = out + resid  # :447
    return out, lam


# --------------------------------------------------------------------------------------
# model forward (rows a-8, a-9)
# --------------------------------------------------------------------------------------
def encoder(cfg: Config, params, mark_table, seqs_i, seqs_t):
    """EasyDGL.__call__ up to the head LayerNorm (EasyDGL.py:70-139), eval/dropout-off semantics.
    Returns (seq_out [B,T,C], [lam per block], aux dict)."""
    C = cfg.num_units
    x0, spans, marks, keymask = input_encode(cfg, params, mark_table, seqs_i, seqs_t)
    prev = x0
    lams = []
    for i in range(cfg.num_blocks):
        pre = f"layer_{i}/"
        st = pre + "attention/self/TMAU/sequential_temporal_combined/"
        layer_in = prev
        att, lam = bimau(cfg, layer_in, keymask, spans, marks,
                         params[pre + "attention/self/TMAU/dense/kernel"],
                         params[pre + "attention/self/TMAU/dense/bias"],
                         params[st + "dense/kernel"], params[st + "dense/bias"],
                         params[st + "weight"], params[st + "scaling"])
        att = dense(att, params[pre + "attention/output/dense/kernel"],
                    params[pre + "attention/output/dense/bias"])  # :113
        att = layernorm(att + layer_in[:, :, :C],
                        params[pre + "attention/output/LayerNorm/gamma"],
                        params[pre + "attention/output/LayerNorm/beta"])  # :116
        inter = dense(att, params[pre + "intermediate/dense/kernel"],
                      params[pre + "intermediate/dense/bias"], gelu)  # :120-121
        out = dense(inter, params[pre + "output/dense/kernel"], params[pre + "output/dense/bias"])  # :125
        out = layernorm(out + att, params[pre + "output/LayerNorm/gamma"],
                        params[pre + "output/LayerNorm/beta"])  # :128
        prev = out
        lams.append(lam)
    so = dense(prev, params["cls/predictions/transform/dense/kernel"],
               params["cls/predictions/transform/dense/bias"], gelu)  # :138
    so = layernorm(so, params["cls/predictions/transform/LayerNorm/gamma"],
                   params["cls/predictions/transform/LayerNorm/beta"])  # :139
    return so, lams, dict(x0=x0, spans=spans, marks=marks, keymask=keymask)


def output_bias_vec(params) -> np.ndarray:
    """Base.py:106-110 — concat([-1000.], output_bias[I-1])."""
    return np.concatenate([[-1000.0], params["CSTMA/output_bias"]])


def score(cfg: Config, params, rows):
    """EasyDGL.py:149-150 — rows @ zero_padded(item_table)^T + output_bias (no sqrt(C) scale)."""
    tab = zero_padded(params["CSTMA/item_embs/lookup_table"])
    return rows @ tab.T + output_bias_vec(params)


def forward(cfg: Config, params, mark_table, features, is_training: bool):
    """EasyDGL.__call__ (EasyDGL.py:69-151).  Returns (logits, lams)."""
    so, lams, _ = encoder(cfg, params, mark_table, features["seqs_i"], features["seqs_t"])
    B = so.shape[0]
    if is_training:
        mp = np.asarray(features["masked_positions"])
        rows = so[np.arange(B)[:, None], mp].reshape(B * mp.shape[1], cfg.num_units)  # :142-143
    else:
        rows = so[:, -1]  # :146
    return score(cfg, params, rows), lams


# --------------------------------------------------------------------------------------
# losses (rows a-10, a-11)
# --------------------------------------------------------------------------------------
def biased_likelihood(lam_g, next_mark, intervals):
    """MAU.biased_likelihood, temporal.py:317-333."""
    lam_g = lam_g * np.sign(next_mark.sum(axis=2, keepdims=True))  # :321
    ev = (lam_g * next_mark).sum(axis=2)  # :322
    event_ll = np.log(np.where(ev == 0, 1.0, ev)).sum()  # :324-325
    entire = lam_g.sum(axis=2)  # :327
    non_event_ll = (entire * intervals * 0.5).sum()  # :328-329
    num_events = next_mark.sum()  # :331
    return -(event_ll - non_event_ll) / num_events  # :332


def regulariser(cfg: Config, params, mark_table, features, labels, lams):
    """EasyDGL.py:157-175: l2 (coding.py:34-40 => l2_reg * sum(w^2)/2 over the raw tables) +
    ct_reg/h * biased_likelihood per block on RAW-second spans."""
    reg = 0.0
    if cfg.l2_reg != 0.0:
        for k in EMBEDDING_TABLES:
            reg += cfg.l2_reg * 0.5 * np.sum(params[k] ** 2)
    if cfg.ct_reg != 0.0:
        h = cfg.num_heads
        mp = np.asarray(features["masked_positions"])
        B = mp.shape[0]
        sp = spans_from_times(np.asarray(features["seqs_t"], dtype=F32))  # :161-162 raw seconds
        sp = sp[np.arange(B)[:, None], mp]  # :163
        nm = np.asarray(mark_table)[np.asarray(labels)].astype(F64)  # :164
        if h != 1:  # :166-169
            sp = np.tile(sp, (h, 1))
            nm = np.tile(nm, (h, 1, 1))
            mp = np.tile(mp, (h, 1))
        for lam in lams:  # :171-175
            lg = lam[np.arange(lam.shape[0])[:, None], mp]
            reg += cfg.ct_reg * biased_likelihood(lg, nm, sp) / h
    return reg


def cross_entropy(cfg: Config, logits, labels):
    """EasyDGL.py:155,177-185: -log(softmax+1e-5)[label], masked mean over label != 0."""
    lp = np.log(softmax(logits, -1) + 1e-5)
    lab = np.asarray(labels).reshape(-1)
    w = (lab != 0).astype(F64)
    per = -lp[np.arange(lab.shape[0]), lab]
    return (w * per).sum() / (w.sum() + 1e-5)


def train_loss(cfg: Config, params, mark_table, features, labels):
    """EasyDGL.train, EasyDGL.py:153-188 (the scalar that is minimised)."""
    logits, lams = forward(cfg, params, mark_table, features, True)
    ce = cross_entropy(cfg, logits, labels)
    reg = regulariser(cfg, params, mark_table, features, labels, lams)
    return ce + reg, dict(ce=ce, reg=reg, logits=logits, lams=lams)


def adam_tf(param, grad, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (Base.py:142-144): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    theta -= lr_t * m / (sqrt(v) + eps).  ``step`` is 1-based."""
    m = beta1 * m + (1 - beta1) * grad
    v = beta2 * v + (1 - beta2) * grad * grad
    lr_t = lr * math.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    return param - lr_t * m / (np.sqrt(v) + eps), m, v


# --------------------------------------------------------------------------------------
# evaluation (row a-13)
# --------------------------------------------------------------------------------------
def top_k(x: np.ndarray, k: int) -> np.ndarray:
    """tf.nn.top_k: descending, ties -> lower index (stable sort on -x)."""
    return np.argsort(-x, axis=-1, kind="stable")[..., :k]


def eval_scores(cfg: Config, params, mark_table, features, mask_seen=True):
    """Base.py:150-165: logits at the last position, -inf at every id in seqs_i (incl. 0 and MASK),
    softmax.  Returns probs [B, I]."""
    logits, _ = forward(cfg, params, mark_table, features, False)
    if mask_seen:
        ids = np.asarray(features["seqs_i"])
        B = ids.shape[0]
        logits = logits.copy()
        logits[np.arange(B)[:, None], ids] = -np.inf  # :156-163
    return softmax(logits, -1)


def ranking_metrics(topk_idx: np.ndarray, real: np.ndarray):
    """Base.py:181-201 per-example HR@k / NDCG@k for k in {10,50,100}; ``topk_idx`` [B,100]."""
    hits = (topk_idx == np.asarray(real).reshape(-1, 1)).astype(F64)
    gain = 1.0 / np.log2(np.arange(2, 100 + 2))
    out = {}
    for k in (10, 50, 100):
        out[f"H{k}"] = np.sign(hits[:, :k].sum(-1))
        out[f"N{k}"] = (hits[:, :k] * gain[:k]).sum(-1)
    return out


def evaluate(cfg: Config, params, mark_table, features, labels, mask_seen=True):
    """Sequential.eval, Base.py:150-207: label = labels[:, -1]; means over the batch."""
    probs = eval_scores(cfg, params, mark_table, features, mask_seen)
    idx = top_k(probs, 100)
    per = ranking_metrics(idx, np.asarray(labels)[:, -1])
    return {k: float(v.mean()) for k, v in per.items()}, idx


# --------------------------------------------------------------------------------------
# batch construction (row a-1)
# --------------------------------------------------------------------------------------
def mask_last(cfg: Config, tokens, timestamps):
    """MAUPostProcessor.mask_last, dataloader.py:166-179 — position T-1 := MASK; labels = tokens."""
    tokens = np.asarray(tokens)
    masked = tokens.copy()
    masked[..., -1] = cfg.mask_id
    return {"seqs_i": masked, "seqs_t": np.asarray(timestamps, dtype=F32)}, tokens


def mask_random(cfg: Config, tokens, timestamps, masked_positions):
    """MAUPostProcessor.mask_random, dataloader.py:181-201, with the positions supplied by the
    caller (the reference draws M distinct positions from [1, T) — dataloader.py:34-36)."""
    tokens = np.asarray(tokens)
    mp = np.asarray(masked_positions)
    B = tokens.shape[0]
    masked = tokens.copy()
    masked[np.arange(B)[:, None], mp] = cfg.mask_id
    labels = tokens[np.arange(B)[:, None], mp]
    return {"seqs_i": masked, "seqs_t": np.asarray(timestamps, dtype=F32),
            "masked_positions": mp}, labels


# --------------------------------------------------------------------------------------
# synthetic data — shared by tests / bench / fixture generator
# --------------------------------------------------------------------------------------
def synthetic_mark_table(num_items: int, num_events: int, multi_hot: bool = False) -> np.ndarray:
    """[num_items, E] int table: one-hot mark i % E for i >= 1, row 0 all-zero (pad)."""
    tab = np.zeros((num_items, num_events), dtype=np.int64)
    idx = np.arange(1, num_items)
    tab[idx, idx % num_events] = 1
    if multi_hot:
        tab[idx, (idx * 7 + 3) % num_events] = 1
    return tab


def synthetic_sequences(cfg: Config, batch: int, rng: np.random.Generator,
                        min_len: int = 5) -> Tuple[np.ndarray, np.ndarray]:
    """Left-padded item ids [B,T] int64 (Zipf(1.1) clipped) and float32 timestamps [B,T]."""
    T = cfg.T
    ids = np.zeros((batch, T), dtype=np.int64)
    ts = np.zeros((batch, T), dtype=F32)
    for b in range(batch):
        n = int(rng.integers(min(min_len, T), T + 1))
        it = np.clip(rng.zipf(1.1, size=n), 1, cfg.num_items - 1)
        t = 9.5e8 + np.cumsum(rng.exponential(3 * 86400.0, size=n))
        ids[b, T - n:] = it
        ts[b, T - n:] = t.astype(F32)
    return ids, ts


def draw_masked_positions(cfg: Config, batch: int, rng: np.random.Generator) -> np.ndarray:
    """M distinct positions in [1, T) per row (dataloader.py:34-36, ignore_head = 1)."""
    T, M = cfg.T, cfg.masklen
    return np.stack([rng.choice(T - 1, M, replace=False) + 1 for _ in range(batch)]).astype(np.int64)

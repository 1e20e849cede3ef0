"""PyTorch-CPU restatement of the reference's EasyDGL graph, in the reference's own op order.

TEST INFRASTRUCTURE (see oracle/easydgl_oracle.py header): used (a) in float64 with autograd as
the GRADIENT oracle for the hand-written HIP backward kernels, after being checked against the
numpy oracle (tests/test_oracle_grad.py), and (b) in float32 as the timed "restated CPU baseline"
of bench.py (TensorFlow is not installable offline).  It deliberately materialises every
intermediate the TensorFlow graph materialises ([hB,T,T] scores, [hB,T,T,E] intensity broadcast,
[B*M, I] logits + one-hot product) so that its cost profile matches the reference's CPU path.

PARITY UNPINNED — same caveat as the numpy oracle.  Nothing under easydgl_amd/ imports this.
"""
from __future__ import annotations

import math
from typing import Dict, List

import numpy as np
import torch

from . import easydgl_oracle as O

PAD_SCORE = O.PAD_SCORE


def to_torch_params(params: Dict[str, np.ndarray], dtype=torch.float64, requires_grad=True):
    return {k: torch.tensor(v, dtype=dtype, requires_grad=requires_grad) for k, v in params.items()}


def gelu(x):  # EasyDGL.py:19-32
    return x * (0.5 * (1.0 + torch.erf(x / math.sqrt(2.0))))


def layernorm(x, gamma, beta, eps=1e-12):  # Base.py:12-67 (joint over all non-batch axes)
    axes = tuple(range(1, x.dim()))
    mean = x.mean(dim=axes, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=axes, keepdim=True)
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


def zero_padded(tab):  # coding.py:56-57
    return torch.cat([torch.zeros_like(tab[:1]), tab[1:]], dim=0)


def split_heads(x, h):  # temporal.py:413-416
    return torch.cat(torch.split(x, x.shape[2] // h, dim=2), dim=0)


def merge_heads(x, h):  # temporal.py:444
    return torch.cat(torch.split(x, x.shape[0] // h, dim=0), dim=2)


def dropout(x, rate, training):
    return torch.nn.functional.dropout(x, rate, training) if (training and rate > 0) else x


def intensity(H, intervals, marks_f, W1, b1, w, scaling, h):  # temporal.py:281-315
    hB, T, dh = H.shape
    E = w.shape[0]
    iv = intervals.unsqueeze(-1).repeat(h, 1, 1)
    lin = torch.cat([H, iv], dim=-1)
    Z = torch.sigmoid(lin @ W1 + b1)
    Z = torch.cat(torch.split(Z, dh, dim=2), dim=0)  # (E*hB, T, dh)  :291
    wt = w.reshape(E, 1, dh, 1).repeat(1, hB, 1, 1).reshape(E * hB, dh, 1)  # :295-297
    sc = torch.exp(scaling).reshape(E, 1, 1, 1).repeat(1, hB, 1, 1).reshape(E * hB, 1, 1)  # :301-303
    mi = torch.matmul(Z, wt) / sc  # :305
    mi = sc * torch.log(1.0 + torch.exp(mi))  # :306
    lam = torch.cat(torch.split(mi, hB, dim=0), dim=2)  # (hB, T, E)  :307
    lam4 = lam.unsqueeze(2).repeat(1, 1, T, 1)  # (hB,T,T,E)  :309-310
    m4 = marks_f.unsqueeze(1).repeat(h, T, 1, 1)  # :311-312
    Mint = (lam4 * m4).sum(-1)  # :313
    return Mint, lam


def bimau(C, h, x, keymask3, spans, marks_f, p, pre, att_drop, training, causal=False, set_diag=True, qkvt=None,
          resid=None):
    """BiMAU.__call__ (temporal.py:404-452).  causal / set_diag=False give MAU.__call__ (temporal.py:335-390: future
    blinding :370-375, no set_diag); qkvt / resid let the caller supply the four projections and the residual (MAU
    projects Q from LN(x) and K, V, T_ from x, and adds the queries back)."""
    st = pre + "sequential_temporal_combined/"
    if qkvt is None:
        qkvt = x @ p[pre + "dense/kernel"] + p[pre + "dense/bias"]
    if resid is None:
        resid = x[:, :, :C]
    Q, K, V, T_ = torch.split(qkvt, C, dim=-1)
    Q_, K_, V_, T__ = (split_heads(a, h) for a in (Q, K, V, T_))
    S = torch.matmul(Q_, K_.transpose(1, 2))
    S = S / (K_.shape[-1] ** 0.5)
    S = torch.where(keymask3 == 0, torch.full_like(S, PAD_SCORE), S)
    T = S.shape[1]
    if causal:  # temporal.py:370-375
        tril = torch.tril(torch.ones(T, T, dtype=torch.bool, device=S.device))
        S = torch.where(tril, S, torch.full_like(S, PAD_SCORE))
    P = torch.softmax(S, dim=-1)
    H = torch.matmul(P, T__)
    Mint, lam = intensity(H, spans, marks_f, p[st + "dense/kernel"], p[st + "dense/bias"],
                          p[st + "weight"], p[st + "scaling"], h)
    if set_diag:
        eye = torch.eye(T, dtype=torch.bool, device=Mint.device)
        Mint = torch.where(eye, torch.ones_like(Mint), Mint)  # set_diag :438-439
    A = dropout(Mint * P, att_drop, training)
    Ovals = torch.matmul(A, V_)
    out = merge_heads(Ovals, h) + resid
    return out, lam


def encoder(cfg: O.Config, p, mark_table: np.ndarray, seqs_i: np.ndarray, seqs_t: np.ndarray,
            dtype=torch.float64, training=False, hidden_drop=0.0, att_drop=0.0):
    C, h = cfg.num_units, cfg.num_heads
    ids = torch.as_tensor(np.asarray(seqs_i), dtype=torch.long)
    ts32 = O.scaled_times(seqs_t, cfg.time_scale)
    spans = torch.tensor(O.spans_from_times(ts32), dtype=dtype)
    marks = torch.as_tensor(O.mark_rows(cfg, mark_table, np.asarray(seqs_i)), dtype=torch.long)
    tcodes = torch.tensor(O.time_sinusoid_code(ts32, C), dtype=dtype)
    x = zero_padded(p["CSTMA/item_embs/lookup_table"])[ids] * (C ** 0.5) + tcodes
    T = ids.shape[1]
    pos = p["CSTMA/spatial_embs/embedding/lookup_table"][:T].unsqueeze(0).expand_as(x)
    mk = zero_padded(p["CSTMA/mark_embs/lookup_table"])[marks].sum(dim=2)
    x0 = torch.cat([x, pos, mk], dim=-1)
    x0 = dropout(x0, hidden_drop, training)
    keymask3 = (ids != 0).to(dtype).unsqueeze(1).repeat(h, T, 1)
    marks_f = marks.to(dtype)
    prev = x0
    lams: List[torch.Tensor] = []
    for i in range(cfg.num_blocks):
        pre = f"layer_{i}/"
        li = prev
        att, lam = bimau(C, h, li, keymask3, spans, marks_f, p, pre + "attention/self/TMAU/",
                         att_drop, training)
        att = att @ p[pre + "attention/output/dense/kernel"] + p[pre + "attention/output/dense/bias"]
        att = dropout(att, hidden_drop, training)
        att = layernorm(att + li[:, :, :C], p[pre + "attention/output/LayerNorm/gamma"],
                        p[pre + "attention/output/LayerNorm/beta"])
        inter = gelu(att @ p[pre + "intermediate/dense/kernel"] + p[pre + "intermediate/dense/bias"])
        out = inter @ p[pre + "output/dense/kernel"] + p[pre + "output/dense/bias"]
        out = dropout(out, hidden_drop, training)
        out = layernorm(out + att, p[pre + "output/LayerNorm/gamma"], p[pre + "output/LayerNorm/beta"])
        prev = out
        lams.append(lam)
    so = gelu(prev @ p["cls/predictions/transform/dense/kernel"] + p["cls/predictions/transform/dense/bias"])
    so = layernorm(so, p["cls/predictions/transform/LayerNorm/gamma"],
                   p["cls/predictions/transform/LayerNorm/beta"])
    return so, lams, dict(x0=x0, spans=spans, marks=marks)


def logits_from(cfg, p, rows):
    tab = zero_padded(p["CSTMA/item_embs/lookup_table"])
    bias = torch.cat([torch.full((1,), -1000.0, dtype=rows.dtype), p["CSTMA/output_bias"]])
    return rows @ tab.t() + bias


def forward(cfg, p, mark_table, features, is_training, dtype=torch.float64,
            hidden_drop=0.0, att_drop=0.0):
    so, lams, aux = encoder(cfg, p, mark_table, features["seqs_i"], features["seqs_t"], dtype,
                            is_training, hidden_drop, att_drop)
    B = so.shape[0]
    if is_training:
        mp = torch.as_tensor(np.asarray(features["masked_positions"]), dtype=torch.long)
        rows = so[torch.arange(B)[:, None], mp].reshape(B * mp.shape[1], cfg.num_units)
    else:
        rows = so[:, -1]
    return logits_from(cfg, p, rows), lams, so


def biased_likelihood(lam_g, nm, iv, nm_total=None):  # temporal.py:317-333
    lam_g = lam_g * torch.sign(nm.sum(dim=2, keepdim=True))
    ev = (lam_g * nm).sum(dim=2)
    event_ll = torch.log(torch.where(ev == 0, torch.ones_like(ev), ev)).sum()
    non_event = (lam_g.sum(dim=2) * iv * 0.5).sum()
    return -(event_ll - non_event) / (nm.sum() if nm_total is None else nm_total)


def train_loss(cfg, p, mark_table, features, labels, dtype=torch.float64,
               hidden_drop=0.0, att_drop=0.0, training=True, w_total=None, nm_total=None):
    """EasyDGL.train (EasyDGL.py:153-188). Returns (loss, dict).
    w_total / nm_total (data-parallel tests): the two batch-sum normalisers — weighted rows (EasyDGL.py:184) and next-event marks
    (temporal.py:333, counted once per sample) — of a LARGER batch this one is a slice of; the result is then this slice's share of
    the larger batch's cross-entropy / TPP term (the l2 term is not a batch sum and is returned in full)."""
    logits, lams, so = forward(cfg, p, mark_table, features, True, dtype,
                               hidden_drop if training else 0.0, att_drop if training else 0.0)
    lp = torch.log(torch.softmax(logits, -1) + 1e-5)
    reg = torch.zeros((), dtype=dtype)
    if cfg.l2_reg != 0.0:
        for k in O.EMBEDDING_TABLES:
            reg = reg + cfg.l2_reg * 0.5 * (p[k] ** 2).sum()
    if cfg.ct_reg != 0.0:
        h = cfg.num_heads
        mp = torch.as_tensor(np.asarray(features["masked_positions"]), dtype=torch.long)
        B = mp.shape[0]
        sp = torch.tensor(O.spans_from_times(np.asarray(features["seqs_t"], dtype=np.float32)), dtype=dtype)
        sp = sp[torch.arange(B)[:, None], mp]
        nm = torch.tensor(np.asarray(mark_table)[np.asarray(labels)], dtype=dtype)
        if h != 1:
            sp, nm, mp = sp.repeat(h, 1), nm.repeat(h, 1, 1), mp.repeat(h, 1)
        for lam in lams:
            lg = lam[torch.arange(lam.shape[0])[:, None], mp]
            reg = reg + cfg.ct_reg * biased_likelihood(lg, nm, sp, None if nm_total is None else nm_total * h) / h
    lab = torch.as_tensor(np.asarray(labels).reshape(-1), dtype=torch.long)
    onehot = torch.nn.functional.one_hot(lab, cfg.I).to(dtype)  # :179 (materialised like the reference)
    w = (lab != 0).to(dtype)
    per = -(lp * onehot).sum(-1)
    ce = (w * per).sum() / ((w.sum() if w_total is None else w_total) + 1e-5)
    return ce + reg, dict(ce=ce, reg=reg, logits=logits, lams=lams, seq_out=so)


class TFAdam:
    """tf.train.AdamOptimizer semantics (Base.py:142-144) over a dict of leaf tensors."""

    def __init__(self, params: Dict[str, torch.Tensor], lr, beta1=0.9, beta2=0.999, eps=1e-8):
        self.p, self.lr, self.b1, self.b2, self.eps = params, lr, beta1, beta2, eps
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.t = 0

    @torch.no_grad()
    def step(self):
        self.t += 1
        lr_t = self.lr * math.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        for k, w in self.p.items():
            g = w.grad
            if g is None:
                continue
            self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
            self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
            w.sub_(lr_t * self.m[k] / (self.v[k].sqrt() + self.eps))
            w.grad = None


def cpu_train_step(cfg, p, opt: TFAdam, mark_table, features, labels, dtype=torch.float32,
                   hidden_drop=0.1, att_drop=0.1):
    """One fwd+bwd+Adam step on the host — the timed body of bench.py's cpu_baseline."""
    loss, _ = train_loss(cfg, p, mark_table, features, labels, dtype, hidden_drop, att_drop)
    loss.backward()
    opt.step()
    return float(loss.detach())

"""Pins of the CTSMA restatement (oracle/ctsma_ref.py) derivable from the reference source: causality, the regressive
loss weights, the closed-form likelihood regulariser, gradients vs finite differences.  CPU only."""
import numpy as np
import torch

from oracle import ctsma_ref as CR
from oracle import easydgl_oracle as O


def _problem(seed=0, B=3, T=7, C=8, h=2, E=3, I=12, nb=2):
    rng = np.random.default_rng(seed)
    p = {k: torch.tensor(v + (0.05 * rng.standard_normal(v.shape) if v.ndim == 1 else 0.0), dtype=torch.float64, requires_grad=True)
         for k, v in CR.init_params(I, T, C, h, E, nb, rng).items()}
    tokens = rng.integers(1, I, size=(B, T + 1))
    tokens[0, :3] = 0                                    # left padding
    ts = np.cumsum(rng.exponential(5.0, size=(B, T + 1)), axis=1).astype(np.float32)
    mt = O.synthetic_mark_table(I, E, multi_hot=True)
    feats = {"seqs_i": tokens[:, :-1], "seqs_t": ts}
    return p, mt, feats, tokens, dict(C=C, h=h, num_blocks=nb, time_scale=2.0)


def test_causality_and_last_position_logits():
    p, mt, feats, tokens, kw = _problem()
    # the joint (T, C) LayerNorm couples positions, so causality is a property of the attention primitive only
    from oracle import torch_ref as R
    B, T = feats["seqs_i"].shape
    C, h = kw["C"], kw["h"]
    rng = np.random.default_rng(1)
    qkvt = torch.tensor(rng.standard_normal((B, T, 4 * C)))
    km = torch.ones(h * B, T, T, dtype=torch.float64)
    pm = {"sequential_temporal_combined/" + k: p["num_blocks_0/attention/modulating_attention/sequential_temporal_combined/" + k]
          for k in ("dense/kernel", "dense/bias", "weight", "scaling")}
    marks = torch.tensor(mt[feats["seqs_i"]], dtype=torch.float64)
    spans = torch.ones(B, T, dtype=torch.float64)
    a, _ = R.bimau(C, h, None, km, spans, marks, pm, "", 0.0, False, causal=True, set_diag=False, qkvt=qkvt, resid=torch.zeros(B, T, C))
    q2 = qkvt.clone()
    q2[:, -1, C:] += 1.0                                   # K, V, T_ of the last key
    b, _ = R.bimau(C, h, None, km, spans, marks, pm, "", 0.0, False, causal=True, set_diag=False, qkvt=q2, resid=torch.zeros(B, T, C))
    assert torch.equal(a[:, :-1], b[:, :-1]) and not torch.equal(a[:, -1], b[:, -1])
    lg = CR.eval_logits(p, mt, feats, **kw)
    assert lg.shape == (B, 12) and torch.all(lg[:, 0] == -1000.0)       # zero row . y + pad bias (Base.py:110)


def test_loss_ignores_padding_labels_and_regulariser_closed_form():
    p, mt, feats, tokens, kw = _problem()
    labels = tokens[:, 1:].copy()
    loss, aux = CR.train_loss(p, mt, feats, labels, ct_reg=0.0, l2_reg=0.0, **kw)
    lab2 = labels.copy()
    lab2[1, 2] = 0                                          # a label 0 carries weight 0 (CTSMA.py:117)
    loss2, _ = CR.train_loss(p, mt, feats, lab2, ct_reg=0.0, l2_reg=0.0, **kw)
    lp = torch.log(torch.softmax(aux["logits"], -1) + 1e-5)
    flat = labels.reshape(-1)
    w = (flat != 0)
    want = -(lp[torch.arange(len(flat)), torch.tensor(flat)][torch.tensor(w)]).sum() / (w.sum() + 1e-5)
    assert abs(float(loss) - float(want)) < 1e-12 and float(loss2) != float(loss)
    # regulariser = ct_reg * biased_likelihood(lam, marks of the next item, raw forward differences), once per block
    loss3, aux3 = CR.train_loss(p, mt, feats, labels, ct_reg=0.3, l2_reg=0.0, **kw)
    raw = feats["seqs_t"].astype(np.float64)
    sp = np.tile(raw[:, 1:] - raw[:, :-1], (kw["h"], 1))
    nm = np.tile(mt[labels].astype(np.float64), (kw["h"], 1, 1))
    reg = sum(0.3 * O.biased_likelihood(l.detach().numpy(), nm, sp) for l in aux3["lams"])
    assert abs(float(aux3["reg"]) - reg) < 1e-10


def test_gradients_match_finite_differences():
    p, mt, feats, tokens, kw = _problem(seed=3, nb=1)
    labels = tokens[:, 1:]
    loss, _ = CR.train_loss(p, mt, feats, labels, ct_reg=0.2, l2_reg=1e-3, **kw)
    loss.backward()
    rng = np.random.default_rng(0)
    for name in ("CSTMA/item_embs/lookup_table", "num_blocks_0/attention/modulating_attention/dense_3/kernel",
                 "num_blocks_0/attention/modulating_attention/sequential_temporal_combined/scaling",
                 "num_blocks_0/feed-forward/Inner/kernel", "outln/LayerNorm/gamma"):
        v = p[name]
        idx = tuple(int(rng.integers(0, s)) for s in v.shape)
        if name.endswith("lookup_table") and idx[0] == 0:
            idx = (1,) + idx[1:]
        eps = 1e-6
        with torch.no_grad():
            old = float(v[idx])
            v[idx] = old + eps
            lp_, _ = CR.train_loss(p, mt, feats, labels, ct_reg=0.2, l2_reg=1e-3, **kw)
            v[idx] = old - eps
            lm_, _ = CR.train_loss(p, mt, feats, labels, ct_reg=0.2, l2_reg=1e-3, **kw)
            v[idx] = old
        fd = (float(lp_) - float(lm_)) / (2 * eps)
        assert abs(fd - float(v.grad[idx])) < 1e-6 * max(1.0, abs(fd)), (name, fd, float(v.grad[idx]))

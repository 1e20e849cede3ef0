"""Pins of the TGAT / TiSASRec restatements (oracle/baselines_ref.py) derivable from the reference source, and of the algebra
the HIP path relies on (the angle-difference form of the time term).  CPU only."""
import numpy as np
import torch

from oracle import baselines_ref as BR


def _tgat_problem(seed=0, B=3, T=9, C=32, h=2, I=20, nb=2, time_scale=50.0):
    rng = np.random.default_rng(seed)
    p = {}
    for k, v in BR.tgat_init_params(I, T, C, nb, rng).items():
        if v.ndim == 1 and "basis_freq" not in k:
            v = v + 0.05 * rng.standard_normal(v.shape)
        p[k] = torch.tensor(v, dtype=torch.float64, requires_grad=True)
    tokens = rng.integers(1, I, size=(B, T + 1))
    tokens[0, :4] = 0
    ts = np.cumsum(rng.exponential(20.0, size=(B, T + 1)), axis=1).astype(np.float32)
    ts[0, :4] = 0.0                                       # left padding carries timestamp 0 (linkpred.py:152-153)
    feats = {"seqs_i": tokens[:, :-1].copy(), "seqs_t": ts}
    return p, feats, tokens, dict(C=C, h=h, nb=nb, time_scale=time_scale)


def test_tgat_time_term_equals_the_angle_difference_form_on_sorted_timestamps():
    """S = Q.(K+pos)^T + sum_d cos(dt w + phi) Q  ==  [Q|Q cos(a w+phi)|Q sin(a w+phi)] . [K+pos|cos(b w)|sin(b w)]^T wherever
    dt = a - b >= 0 — every unmasked (k <= q) pair of a time-sorted sequence (the form csrc/k_tattn.hip computes)."""
    rng = np.random.default_rng(1)
    T, C = 7, 8
    Q, K, pos = (torch.tensor(rng.standard_normal((T, C))) for _ in range(3))
    w, phi = torch.tensor(np.linspace(0, 9, C)), torch.tensor(0.1 * rng.standard_normal(C))
    t = torch.tensor(np.cumsum(rng.exponential(0.3, T + 1)))
    dt = torch.clamp(t[1:, None] - t[None, :-1], min=0.0)                       # TGAT.py:51-54
    S_ref = Q @ K.T + Q @ pos.T + (torch.cos(dt[..., None] * w + phi) * Q[:, None, :]).sum(-1)
    a, b = t[1:] - t[-1], t[:-1] - t[-1]
    qx = torch.cat([Q, Q * torch.cos(a[:, None] * w + phi), Q * torch.sin(a[:, None] * w + phi)], 1)
    kx = torch.cat([K + pos, torch.cos(b[:, None] * w), torch.sin(b[:, None] * w)], 1)
    S = qx @ kx.T
    tril = torch.tril(torch.ones(T, T)).bool()
    assert torch.allclose(S[tril], S_ref[tril], atol=1e-12)
    assert not torch.allclose(S[~tril], S_ref[~tril], atol=1e-3)               # the clamp only differs where the causal mask hides it


def test_tgat_padded_query_rows_are_uniform_over_all_keys_and_logit_column_zero():
    p, feats, tokens, kw = _tgat_problem()
    C, h = kw["C"], kw["h"]
    a = "num_blocks_0/attention/attention/timeinterval/"
    ids = torch.as_tensor(feats["seqs_i"])
    x = torch.zeros(3, 9, C, dtype=torch.float64)
    x[ids != 0] = torch.randn(int((ids != 0).sum()), C, dtype=torch.float64)
    qn = torch.randn(3, 9, C, dtype=torch.float64)
    spans = torch.zeros(3, 9, 9, dtype=torch.float64)
    out = BR.tf_attention(p, a, qn, x, spans, p["TGAT/pcoding_K/embedding/lookup_table"], p["TGAT/tcoding_K/basis_freq"],
                          p["TGAT/tcoding_K/phase"], h)
    V = x @ p[a + "dense_2/kernel"] + p[a + "dense_2/bias"]
    # sample 0, query 1: keys 0..1 are padding and keys 2.. are in the future -> every score is the same constant
    assert torch.allclose(out[0, 1], V[0].mean(0) + qn[0, 1], atol=1e-12)
    lg = BR.tgat_eval_logits(p, feats, **kw)
    assert lg.shape == (3, 20) and torch.all(lg[:, 0] == -1000.0)


def test_tgat_gradients_match_finite_differences():
    p, feats, tokens, kw = _tgat_problem(seed=3, nb=1)
    labels = tokens[:, 1:]
    loss, _ = BR.tgat_train_loss(p, feats, labels, l2_reg=1e-3, **kw)
    loss.backward()
    rng = np.random.default_rng(0)
    for name in ("TGAT/tcoding_K/basis_freq", "TGAT/tcoding_K/phase", "TGAT/pcoding_K/embedding/lookup_table",
                 "num_blocks_0/attention/attention/timeinterval/dense_1/kernel", "num_blocks_0/feedforward/Inner/kernel"):
        v = p[name]
        d = torch.tensor(rng.standard_normal(tuple(v.shape)))
        eps = 1e-6
        with torch.no_grad():
            v += eps * d
            lp, _ = BR.tgat_train_loss(p, feats, labels, l2_reg=1e-3, **kw)
            v -= 2 * eps * d
            lm, _ = BR.tgat_train_loss(p, feats, labels, l2_reg=1e-3, **kw)
            v += eps * d
        fd = (float(lp) - float(lm)) / (2 * eps)
        an = float((v.grad * d).sum())
        assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)), (name, fd, an)


def test_tisasrec_interval_buckets_and_shapes():
    rng = np.random.default_rng(2)
    B, T, C, h, I, nb, timelen = 2, 6, 16, 2, 15, 1, 8
    p = {k: torch.tensor(v, dtype=torch.float64) for k, v in BR.tisasrec_init_params(I, timelen, C, nb, rng).items()}
    tokens = rng.integers(1, I, size=(B, T + 1))
    ts = np.cumsum(rng.exponential(3.0, size=(B, T + 1)), axis=1).astype(np.float32)
    feats = {"seqs_i": tokens[:, :-1], "seqs_t": ts}
    lg = BR.tisasrec_eval_logits(p, feats, C, h, nb, 1.0, timelen)
    assert lg.shape == (B, I) and torch.all(lg[:, 0] == -1000.0)
    # an interval clipped to timelen indexes one past the [timelen, C] table: the GPU lookup of the reference returns zeros
    tab = torch.arange(12.0).reshape(4, 3)
    got = BR._lookup_or_zero(tab, torch.tensor([[0, 3, 4]]))
    assert torch.equal(got[0, 1], tab[3]) and torch.all(got[0, 2] == 0)

"""Known-answer tests that pin the numpy oracle to the reference source.

The reference ships no golden vectors, so these are the closed-form consequences of its code,
each citing the reference lines it follows.
"""
import math

import numpy as np
import pytest

from oracle import easydgl_oracle as O


def small_cfg(**kw):
    base = dict(num_items=10, seqslen=4, num_units=8, num_heads=2, num_blocks=1, masklen=2,
                time_scale=1.0, num_events=3)
    base.update(kw)
    return O.Config(**base)


def test_time_sinusoid_of_zero():
    # coding.py:144-148: sin(0)=0 on even, cos(0)=1 on odd channels
    code = O.time_sinusoid_code(np.zeros((2, 3), np.float32), 8)
    assert code.shape == (2, 3, 8)
    np.testing.assert_array_equal(code[..., 0::2], 0.0)
    np.testing.assert_array_equal(code[..., 1::2], 1.0)


def test_time_sinusoid_interleave_and_scale():
    # coding.py:134,142: channel 2j uses scale 10000^(2j/C)
    t = np.array([[3.0]], np.float32)
    code = O.time_sinusoid_code(t, 4)
    s1 = np.float32(10000.0 ** (2 / 4))
    x1 = np.float64(np.float32(3.0) / s1)
    np.testing.assert_allclose(code[0, 0], [math.sin(3.0), math.cos(3.0), math.sin(x1), math.cos(x1)], rtol=1e-15)


def test_layernorm_of_constant_is_beta():
    # Base.py:52-63: x-mean = 0 -> y = beta
    x = np.full((2, 5, 4), 3.25)
    beta = np.arange(4.0)
    y = O.layernorm(x, np.ones(4) * 2, beta)
    np.testing.assert_allclose(y, np.broadcast_to(beta, x.shape), atol=1e-12)


def test_layernorm_is_joint_over_T_and_C():
    # Base.py:13,51: begin_norm_axis=1 -> one mean/var per sample over (T,C)
    rng = np.random.default_rng(0)
    x = rng.standard_normal((3, 5, 4))
    y = O.layernorm(x, np.ones(4), np.zeros(4))
    np.testing.assert_allclose(y.reshape(3, -1).mean(1), 0, atol=1e-12)
    np.testing.assert_allclose(y.reshape(3, -1).var(1), 1, rtol=1e-9)
    # and it is NOT the per-row (last-axis) layernorm
    assert abs(y[0, 0].mean()) > 1e-3


def test_span_head_copy_and_clip():
    # EasyDGL.py:73-74
    ts = np.array([[0.0, 5.0, 7.0, 300.0, 299.0]], np.float32)
    sp = O.spans_from_times(ts)
    np.testing.assert_array_equal(sp, [[5.0, 5.0, 2.0, 100.0, 0.0]])


def test_fully_masked_row_gives_uniform_softmax():
    # temporal.py:425-429: all keys masked -> every score = -2^32+1 -> uniform
    cfg = small_cfg()
    rng = np.random.default_rng(1)
    p = O.init_params(cfg, rng, perturb=True)
    mt = O.synthetic_mark_table(cfg.num_items, cfg.num_events)
    ids = np.zeros((1, cfg.T), np.int64)  # everything is padding
    ts = np.zeros((1, cfg.T), np.float32)
    x0, spans, marks, km = O.input_encode(cfg, p, mt, ids, ts)
    assert km.sum() == 0
    # P uniform => H = mean_k T_[k]; check through bimau's intermediate by recomputing
    pre = "layer_0/attention/self/TMAU/"
    qkvt = x0 @ p[pre + "dense/kernel"] + p[pre + "dense/bias"]
    S = np.full((cfg.num_heads, cfg.T, cfg.T), O.PAD_SCORE)
    P = O.softmax(S)
    np.testing.assert_allclose(P, 1.0 / cfg.T)
    assert O.PAD_SCORE == -4294967296.0


def test_intensity_ln2_when_weight_and_scaling_zero():
    # temporal.py:299-306: w=0 -> z=0; scaling=0 -> s=1 -> lambda = log(2); all-ones single mark -> Mint = ln2
    cfg = small_cfg(num_events=1)
    H = np.random.default_rng(2).standard_normal((cfg.num_heads * 2, cfg.T, cfg.dh))
    iv = np.ones((2, cfg.T))
    marks = np.ones((2, cfg.T, 1), np.int64)
    W1 = np.random.default_rng(3).standard_normal((cfg.dh + 1, cfg.dh))
    Mint, lam = O.intensity(H, iv, marks, W1, np.zeros(cfg.dh), np.zeros((1, cfg.dh)), np.zeros(1), cfg.num_heads)
    np.testing.assert_allclose(lam, math.log(2.0), rtol=1e-15)
    np.testing.assert_allclose(Mint, math.log(2.0), rtol=1e-15)


def test_bimau_diag_is_one():
    # temporal.py:438-439 with lambda = ln2 off-diagonal => A = P*ln2 off-diag, A = P on diag
    cfg = small_cfg(num_events=1, num_heads=1)
    rng = np.random.default_rng(4)
    B, T, C = 2, cfg.T, cfg.num_units
    x = rng.standard_normal((B, T, C))
    Wq = rng.standard_normal((C, 4 * C)) * 0.1
    km = np.ones((B, T))
    marks = np.ones((B, T, 1), np.int64)
    out, lam = O.bimau(cfg, x, km, np.ones((B, T)), marks, Wq, np.zeros(4 * C),
                       rng.standard_normal((cfg.dh + 1, cfg.dh)), np.zeros(cfg.dh),
                       np.zeros((1, cfg.dh)), np.zeros(1))
    qkvt = x @ Wq
    Q, K, V, T_ = np.split(qkvt, 4, -1)
    P = O.softmax(Q @ K.transpose(0, 2, 1) / math.sqrt(C))
    G = np.full((B, T, T), math.log(2.0))
    G[:, np.arange(T), np.arange(T)] = 1.0
    np.testing.assert_allclose(out, (G * P) @ V + x, rtol=1e-12)


def test_logit_column_zero_is_minus_1000():
    # coding.py:56-57 + Base.py:110: zero row . y + (-1000)
    cfg = small_cfg()
    rng = np.random.default_rng(5)
    p = O.init_params(cfg, rng, perturb=True)
    mt = O.synthetic_mark_table(cfg.num_items, cfg.num_events)
    ids, ts = O.synthetic_sequences(cfg, 3, rng, min_len=2)
    feats, _ = O.mask_last(cfg, ids, ts)
    logits, _ = O.forward(cfg, p, mt, feats, False)
    assert logits.shape == (3, cfg.I)
    np.testing.assert_array_equal(logits[:, 0], -1000.0)


def test_mark_embedding_uses_values_as_indices():
    # EasyDGL.py:87-88: mk = (#active marks) * mark_emb_table[1]
    cfg = small_cfg(num_events=4)
    rng = np.random.default_rng(6)
    p = O.init_params(cfg, rng)
    mt = O.synthetic_mark_table(cfg.num_items, cfg.num_events, multi_hot=True)
    ids = np.array([[0, 1, 2, 3, cfg.mask_id]], np.int64)
    ts = np.arange(cfg.T, dtype=np.float32)[None]
    x0, _, marks, _ = O.input_encode(cfg, p, mt, ids, ts)
    C = cfg.num_units
    n = marks.sum(-1)
    assert n[0, 0] == 0 and n[0, -1] == 0  # pad and MASK -> row 0 -> no marks (EasyDGL.py:76)
    np.testing.assert_allclose(x0[..., 2 * C:], n[..., None] * p["CSTMA/mark_embs/lookup_table"][1], rtol=1e-15)


def test_biased_likelihood_closed_form():
    # temporal.py:321-333: constant lambda=c, one-hot next mark, interval d, E marks:
    # -(N log c - N*E*c*d/2)/N = -(log c - E c d / 2)
    c, d, E, N = 0.7, 3.0, 4, 6
    lam = np.full((N, 1, E), c)
    nm = np.zeros((N, 1, E)); nm[:, 0, 1] = 1
    iv = np.full((N, 1), d)
    got = O.biased_likelihood(lam, nm, iv)
    np.testing.assert_allclose(got, -(math.log(c) - E * c * d / 2), rtol=1e-14)


def test_biased_likelihood_ignores_rows_without_next_mark():
    lam = np.full((2, 1, 3), 0.5)
    nm = np.zeros((2, 1, 3)); nm[0, 0, 2] = 1  # row 1 has no next mark (label = pad)
    iv = np.full((2, 1), 2.0)
    got = O.biased_likelihood(lam, nm, iv)
    np.testing.assert_allclose(got, -(math.log(0.5) - 3 * 0.5 * 2.0 / 2), rtol=1e-14)


def test_topk_tie_goes_to_lower_index():
    x = np.array([[0.1, 0.5, 0.5, 0.2, 0.5]])
    np.testing.assert_array_equal(O.top_k(x, 3), [[1, 2, 4]])


def test_ranking_metrics_hand_example():
    # Base.py:181-198
    topk = np.tile(np.arange(100)[None], (3, 1)) + 1  # predicted ids 1..100
    real = np.array([1, 11, 500])  # rank 0, rank 10, miss
    m = O.ranking_metrics(topk, real)
    np.testing.assert_array_equal(m["H10"], [1, 0, 0])
    np.testing.assert_array_equal(m["H50"], [1, 1, 0])
    np.testing.assert_allclose(m["N10"], [1.0, 0.0, 0.0])
    np.testing.assert_allclose(m["N50"], [1.0, 1 / math.log2(12), 0.0])


def test_seen_label_is_a_miss():
    # Base.py:156-163: the label itself, if present in seqs_i, is masked to -inf -> prob 0 -> miss
    cfg = small_cfg(num_items=120)
    rng = np.random.default_rng(7)
    p = O.init_params(cfg, rng, perturb=True)
    mt = O.synthetic_mark_table(cfg.num_items, cfg.num_events)
    tokens = np.array([[0, 1, 2, 3, 2], [0, 1, 2, 3, 7]], np.int64)  # row 0: label=2 already seen at t=2
    ts = np.tile(np.arange(cfg.T, dtype=np.float32)[None] * 10, (2, 1))
    feats, labels = O.mask_last(cfg, tokens, ts)
    probs = O.eval_scores(cfg, p, mt, feats, mask_seen=True)
    for col in (0, 1, 2, 3, cfg.mask_id):
        assert probs[0, col] == 0.0
    assert probs[1, 7] > 0.0
    metrics, idx = O.evaluate(cfg, p, mt, feats, labels)
    assert 2 not in idx[0]          # 116 items have non-zero probability, so no zero-prob id enters the top-100
    assert idx.shape == (2, 100)


def test_masking_semantics():
    # dataloader.py:166-201
    cfg = small_cfg()
    tokens = np.array([[0, 3, 4, 5, 6]], np.int64)
    ts = np.arange(5, dtype=np.float32)[None]
    f, lab = O.mask_last(cfg, tokens, ts)
    np.testing.assert_array_equal(f["seqs_i"], [[0, 3, 4, 5, cfg.mask_id]])
    np.testing.assert_array_equal(lab, tokens)
    f, lab = O.mask_random(cfg, tokens, ts, np.array([[3, 1]]))
    np.testing.assert_array_equal(f["seqs_i"], [[0, cfg.mask_id, 4, cfg.mask_id, 6]])
    np.testing.assert_array_equal(lab, [[5, 3]])
    mp = O.draw_masked_positions(cfg, 50, np.random.default_rng(0))
    assert mp.min() >= 1 and mp.max() < cfg.T
    assert all(len(set(r)) == cfg.masklen for r in mp)


def test_cross_entropy_floor_and_weights():
    # EasyDGL.py:155,177-185
    cfg = small_cfg(num_items=3)
    logits = np.array([[0.0, 1.0, 2.0, 3.0], [5.0, 0.0, 0.0, 0.0]])
    labels = np.array([2, 0])  # second row: label 0 -> weight 0
    p = O.softmax(logits)
    want = -math.log(p[0, 2] + 1e-5) / (1 + 1e-5)
    np.testing.assert_allclose(O.cross_entropy(cfg, logits, labels), want, rtol=1e-14)


def test_adam_tf_first_step():
    # Base.py:143: after one step theta -= lr * g/(|g| + eps*sqrt(1-b2)) ~ lr*sign(g)
    w, m, v = O.adam_tf(np.array([1.0]), np.array([0.5]), np.zeros(1), np.zeros(1), 1, 0.1)
    lr_t = 0.1 * math.sqrt(1 - 0.999) / (1 - 0.9)
    np.testing.assert_allclose(w, 1.0 - lr_t * 0.05 / (math.sqrt(0.001 * 0.25) + 1e-8), rtol=1e-14)


def test_head_major_layout_and_row_independence():
    # temporal.py:413-416: b' = head*B + b ; each sample only sees itself
    cfg = small_cfg(num_heads=2, num_events=3)
    rng = np.random.default_rng(8)
    p = O.init_params(cfg, rng, perturb=True)
    mt = O.synthetic_mark_table(cfg.num_items, cfg.num_events)
    ids, ts = O.synthetic_sequences(cfg, 4, rng, min_len=2)
    so, lams, _ = O.encoder(cfg, p, mt, ids, ts)
    so1, lams1, _ = O.encoder(cfg, p, mt, ids[1:2], ts[1:2])
    np.testing.assert_allclose(so[1:2], so1, rtol=1e-10, atol=1e-12)
    B = 4
    for hd in range(cfg.num_heads):
        np.testing.assert_allclose(lams[0][hd * B + 1], lams1[0][hd], rtol=1e-10, atol=1e-12)

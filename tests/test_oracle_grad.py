"""Pins oracle/torch_ref.py (the gradient oracle / CPU baseline) to the numpy oracle:
forward agreement in float64, and torch autograd gradients against central finite differences of the
numpy oracle's scalar loss."""
import numpy as np
import pytest
import torch

from oracle import easydgl_oracle as O
from oracle import torch_ref as R


def make_case(seed=0, **kw):
    base = dict(num_items=23, seqslen=6, num_units=8, num_heads=2, num_blocks=2, masklen=3,
                time_scale=86400.0, ct_reg=1e-2, l2_reg=1e-3, num_events=3)
    base.update(kw)
    cfg = O.Config(**base)
    rng = np.random.default_rng(seed)
    params = O.init_params(cfg, rng, perturb=True)
    mt = O.synthetic_mark_table(cfg.num_items, cfg.num_events, multi_hot=True)
    ids, ts = O.synthetic_sequences(cfg, 3, rng, min_len=3)
    mp = O.draw_masked_positions(cfg, 3, rng)
    feats, labels = O.mask_random(cfg, ids, ts, mp)
    return cfg, params, mt, feats, labels


def test_torch_ref_forward_matches_numpy_oracle():
    cfg, params, mt, feats, labels = make_case()
    loss_np, aux_np = O.train_loss(cfg, params, mt, feats, labels)
    p = R.to_torch_params(params)
    loss_t, aux_t = R.train_loss(cfg, p, mt, feats, labels)
    np.testing.assert_allclose(aux_t["logits"].detach().numpy(), aux_np["logits"], rtol=1e-10, atol=1e-10)
    for a, b in zip(aux_t["lams"], aux_np["lams"]):
        np.testing.assert_allclose(a.detach().numpy(), b, rtol=1e-10)
    np.testing.assert_allclose(float(loss_t.detach()), loss_np, rtol=1e-12)
    np.testing.assert_allclose(float(aux_t["reg"].detach()), aux_np["reg"], rtol=1e-12)


def test_torch_ref_eval_forward_matches():
    cfg, params, mt, feats, labels = make_case(seed=3)
    ids = np.where(feats["seqs_i"] == cfg.mask_id, 1, feats["seqs_i"])
    f2, lab2 = O.mask_last(cfg, ids, feats["seqs_t"])
    logits_np, _ = O.forward(cfg, params, mt, f2, False)
    logits_t, _, _ = R.forward(cfg, R.to_torch_params(params, requires_grad=False), mt, f2, False)
    np.testing.assert_allclose(logits_t.numpy(), logits_np, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("seed", [0, 1])
def test_autograd_matches_finite_differences_of_numpy_oracle(seed):
    cfg, params, mt, feats, labels = make_case(seed=seed)
    p = R.to_torch_params(params)
    loss_t, _ = R.train_loss(cfg, p, mt, feats, labels)
    loss_t.backward()
    rng = np.random.default_rng(100 + seed)
    eps = 1e-6
    for name, w in params.items():
        g = p[name].grad
        assert g is not None, name
        g = g.numpy()
        flat = w.reshape(-1)
        for j in rng.choice(flat.size, size=min(4, flat.size), replace=False):
            old = flat[j]
            flat[j] = old + eps
            lp, _ = O.train_loss(cfg, params, mt, feats, labels)
            flat[j] = old - eps
            lm, _ = O.train_loss(cfg, params, mt, feats, labels)
            flat[j] = old
            fd = (lp - lm) / (2 * eps)
            assert abs(fd - g.reshape(-1)[j]) <= 1e-6 * max(1.0, abs(fd)) + 2e-8, (name, j, fd, g.reshape(-1)[j])


def test_table_row0_gets_only_l2_gradient():
    # coding.py:56-57: the used table has a zero CONSTANT in row 0; only the l2 term touches variable row 0
    cfg, params, mt, feats, labels = make_case(seed=5)
    p = R.to_torch_params(params)
    loss_t, _ = R.train_loss(cfg, p, mt, feats, labels)
    loss_t.backward()
    for k in ("CSTMA/item_embs/lookup_table", "CSTMA/mark_embs/lookup_table"):
        np.testing.assert_allclose(p[k].grad[0].numpy(), cfg.l2_reg * params[k][0], rtol=1e-12)


def test_tf_adam_matches_numpy():
    w0 = np.array([0.3, -1.2]); g = np.array([0.05, -0.4])
    pt = {"w": torch.tensor(w0, requires_grad=True)}
    opt = R.TFAdam(pt, lr=0.01)
    m = np.zeros(2); v = np.zeros(2); w = w0.copy()
    for t in range(1, 4):
        pt["w"].grad = torch.tensor(g * t)
        opt.step()
        w, m, v = O.adam_tf(w, g * t, m, v, t, 0.01)
    np.testing.assert_allclose(pt["w"].detach().numpy(), w, rtol=1e-13)

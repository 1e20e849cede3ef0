"""Channel padding of the regressive models (model/base.py): TGAT / TiSASRec / CTSMA at the reference's DEFAULT width
--num_units 50 --num_heads 1 (main.py:35-37) run zero-padded at head dim 64.  Parity of loss / every gradient / eval logits at the
true width against the float64 oracles is part of the models' own test files (their padded CASES); here: the invariants of the
padded storage — padded entries exactly zero after construction, after loading reference-shaped variables and after optimizer
steps with dropout; reference shapes in and out; a wrong shape refused."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import easydgl_oracle as O
from tests._util import to_dev

pytestmark = pytest.mark.gpu


def _flags(model, C, h, nb, T, I, E, mode):
    return SimpleNamespace(model=model, num_items=I, num_units=C, num_heads=h, num_blocks=nb, seqslen=T, timelen=64,
                           time_scale=86400.0, learning_rate=1e-3, l2_reg=1e-4, ct_reg=1e-3, hidden_dropout_rate=0.1,
                           attention_probs_dropout_rate=0.1, mark_table=O.synthetic_mark_table(I, E, multi_hot=True),
                           compute_dtype=mode, num_train_steps=None, num_warmup_steps=None)


def _batch(B, T, I, seed):
    rng = np.random.default_rng(seed)
    tokens = rng.integers(1, I, size=(B, T + 1))
    tokens[0, :T // 3] = 0
    ts = (9.5e8 + np.cumsum(rng.exponential(0.3 * 86400.0, size=(B, T + 1)), axis=1)).astype(np.float32)
    ts[tokens == 0] = 0.0
    return {"seqs_i": tokens[:, :-1].copy(), "seqs_t": ts}, tokens[:, 1:].copy()


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("model,C,h", [("TGAT", 50, 1), ("TiSASREC", 50, 1), ("CTSMA", 50, 1), ("TGAT", 100, 2), ("CTSMA", 24, 2)])
def test_padded_entries_stay_exactly_zero_through_training(model, C, h, mode):
    import easydgl_amd
    T, I, E, B = 20, 200, 6, 16
    m = easydgl_amd.ranking(_flags(model, C, h, 2, T, I, E, mode)).finalize("cuda")
    dh_true = C // h
    dh_pad = next(d for d in (16, 32, 64, 128) if d >= dh_true)
    assert m.pad == (dh_pad, dh_true) and m.num_units == h * dh_pad and m.width_true == C
    assert m.padded_leak() == 0.0
    feats, labels = _batch(B, T, I, 3)
    feats, labels = to_dev(feats), torch.as_tensor(labels).cuda()
    losses = [float(m.train_step(feats, labels)) for _ in range(4)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0]
    assert m.padded_leak() == 0.0, "a padded parameter entry moved"
    # a backward alone does leave entries on padded positions that mask_padded_grads has to clear (or none at all): after it, none
    m.zero_grad_arena()
    m.train_loss(feats, labels).backward()
    m.mask_padded_grads()
    g = {id(p): p.grad.detach().clone() for _, p, _, _ in m._pad_specs()}
    for _, p, maps, _ in m._pad_specs():
        g[id(p)][m._spec_index(p, maps)] = 0
    assert max(float(t.abs().max()) for t in g.values()) == 0.0
    # evaluation runs on the padded storage as well
    m.reset_metrics()
    _, idx = m.eval_topk(feats, mask_seen=True, K=10)
    assert idx.shape == (B, 10)


@pytest.mark.parametrize("model", ["TGAT", "TiSASREC", "CTSMA"])
def test_reference_shapes_in_and_out(model):
    import easydgl_amd
    T, I, E, C = 12, 90, 5, 50
    m = easydgl_amd.ranking(_flags(model, C, 1, 1, T, I, E, "f32")).finalize("cuda")
    vals = m._padded_values(False)
    for name, v in vals.items():
        assert v.shape[-1] not in (64, 128), (name, tuple(v.shape))      # the reference's own widths
    new = {k: (v.cpu().numpy() + 0.01).astype(np.float32) for k, v in vals.items()}
    m.load_tf_variables(new)
    assert m.padded_leak() == 0.0
    back = m._padded_values(False)
    for k in new:
        assert np.array_equal(back[k].cpu().numpy(), new[k]), k
    bad = dict(new)
    k0 = next(k for k in new if k.endswith("item_embs/lookup_table"))
    bad[k0] = np.zeros((new[k0].shape[0], 64), np.float32)
    with pytest.raises(ValueError):
        m.load_tf_variables(bad)

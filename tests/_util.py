"""Shared helpers for the GPU parity tests: build the same problem for the oracle and the HIP model."""
from types import SimpleNamespace

import numpy as np
import torch

from oracle import easydgl_oracle as O


def make_problem(seed=0, batch=4, perturb=True, multi_hot=True, **kw):
    base = dict(num_items=50, seqslen=10, num_units=32, num_heads=2, num_blocks=2, masklen=3,
                time_scale=86400.0, ct_reg=1e-2, l2_reg=1e-3, learning_rate=1e-3, num_events=4)
    base.update(kw)
    cfg = O.Config(**base)
    rng = np.random.default_rng(seed)
    params = O.init_params(cfg, rng, perturb=perturb)
    mt = O.synthetic_mark_table(cfg.num_items, cfg.num_events, multi_hot=multi_hot)
    ids, ts = O.synthetic_sequences(cfg, batch, rng, min_len=3)
    mp = O.draw_masked_positions(cfg, batch, rng)
    feats, labels = O.mask_random(cfg, ids, ts, mp)
    efeats, elabels = O.mask_last(cfg, ids, ts)
    return dict(cfg=cfg, params=params, mark_table=mt, ids=ids, ts=ts, feats=feats, labels=labels,
                efeats=efeats, elabels=elabels)


def flags_from(cfg: O.Config, mark_table, compute_dtype="f32", hidden_drop=0.0, att_drop=0.0):
    return SimpleNamespace(model="EasyDGL", num_items=cfg.num_items, num_units=cfg.num_units, num_heads=cfg.num_heads,
                           num_blocks=cfg.num_blocks, seqslen=cfg.seqslen, masklen=cfg.masklen,
                           time_scale=cfg.time_scale, learning_rate=cfg.learning_rate, l2_reg=cfg.l2_reg,
                           ct_reg=cfg.ct_reg, hidden_dropout_rate=hidden_drop, attention_probs_dropout_rate=att_drop,
                           mark_table=mark_table, compute_dtype=compute_dtype, num_train_steps=None,
                           num_warmup_steps=None)


def build_model(prob, compute_dtype="f32", hidden_drop=0.0, att_drop=0.0):
    import easydgl_amd
    F = flags_from(prob["cfg"], prob["mark_table"], compute_dtype, hidden_drop, att_drop)
    m = easydgl_amd.ranking(F).finalize("cuda")
    m.load_tf_variables(prob["params"])
    return m


def to_dev(feats):
    out = {}
    for k, v in feats.items():
        t = torch.as_tensor(np.asarray(v))
        out[k] = t.cuda().contiguous()
    return out


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def assert_close(a, b, tol, what=""):
    e = rel_err(a, b)
    assert e <= tol, f"{what}: max-abs-err / max-abs-ref = {e:.3e} > {tol:.1e}"

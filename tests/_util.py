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


def relu_flip_err(got, ref, tol):
    """Error measure for the gradients behind a ReLU (FeedForward's Inner kernel / bias) on the bf16 path.  A
    pre-activation within bf16 rounding of 0 flips its ReLU mask against the fp64 reference, which moves ONE whole term of
    the row sum of that hidden unit: the error sits in a few columns (hidden units), at ~1/sqrt(rows) of the column's
    size, and is not a rounding effect the max-norm tolerance can bound (measured with the kernels checked exact on their
    own operands: 5 of 512 columns above 0.1, max 0.42, at 120 rows).  Returns the max error over the columns that are NOT
    flipped, after checking that (a) at most 5 % of the columns exceed `tol` and (b) the rms error stays below tol / 2."""
    got = np.asarray(got, dtype=np.float64).reshape(-1, np.asarray(ref).shape[-1])
    ref = np.asarray(ref, dtype=np.float64).reshape(got.shape)
    mx = np.abs(ref).max() + 1e-30
    err = np.abs(got - ref) / mx
    col = err.max(axis=0)
    flipped = col > tol
    assert flipped.mean() <= 0.05, f"{flipped.sum()} of {len(col)} columns exceed {tol}: not ReLU flips"
    rms = float(np.sqrt((err ** 2).mean()))
    assert rms <= tol / 2, f"rms error {rms:.3e} > {tol / 2:.1e}"
    return float(col[~flipped].max()) if (~flipped).any() else 0.0


# bf16 path, gradients: a max-norm bound alone lets every small entry of a tensor be wrong.  Every gradient
# tensor is held to a relative L2 error AND to max-abs-error / max-abs-reference.  f32 path: 1e-3 on both.
GRAD_TOL = {"f32": (1e-3, 1e-3), "bf16": (2e-2, 5e-2)}
LOSS_TOL = {"f32": 1e-4, "bf16": 5e-3}


def grad_errors(got, want):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    d = got - want
    return float(np.linalg.norm(d) / (np.linalg.norm(want) + 1e-300)), float(np.abs(d).max() / (np.abs(want).max() + 1e-30))


def grad_ok(got, want, mode):
    """(ok, (rel_l2, rel_max)) under GRAD_TOL[mode]; a reference gradient that is exactly zero must be reproduced as (near) zero."""
    l2, mx = grad_errors(got, want)
    t2, tm = GRAD_TOL[mode]
    if not np.any(np.asarray(want)):
        return bool(np.abs(np.asarray(got, dtype=np.float64)).max() < 1e-12), (l2, mx)
    return bool(l2 <= t2 and mx <= tm), (l2, mx)


def dump_errors(tag, errs):
    """EDGL_TEST_DUMP=<dir>: append the measured (rel-L2, rel-max) errors of a parity test to <dir>/tol_<tag>.json — how the
    tolerances of the tests are set (tools/ run on the GPU box)."""
    import json
    import os
    d = os.environ.get("EDGL_TEST_DUMP")
    if not d:
        return
    os.makedirs(d, exist_ok=True)
    path = os.path.join(d, f"tol_{tag}.json")
    old = json.load(open(path)) if os.path.exists(path) else {}
    for k, v in errs.items():
        o = old.get(k, [0.0, 0.0])
        old[k] = [max(o[0], float(v[0])), max(o[1], float(v[1]))]
    json.dump(old, open(path, "w"), indent=1, sort_keys=True)


def regressive_bf16_bounds(model: str, name: str, gtol: float, l2_default: float = 8e-2):
    """(relative-L2 bound, max-norm bound) of one gradient tensor of TGAT / TiSASRec / CTSMA on the bf16 path: per tensor class,
    <= 2 x the errors measured on the GPU (tests/golden/regressive_bf16_bounds.json, written by make_regressive_bounds.py from
    profiles/r05_parity_errors.json).  The ReLU-gated Inner tensors keep relu_flip_err's bound."""
    import json
    import os
    from tests.golden.make_regressive_bounds import tensor_class
    c = tensor_class(name)
    if c == "ffn_inner":
        return l2_default, gtol
    global _REG_BOUNDS
    try:
        tab = _REG_BOUNDS
    except NameError:
        tab = _REG_BOUNDS = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "regressive_bf16_bounds.json")))["bounds"]
    l2, mx = tab[model][c]
    return min(l2, l2_default), min(mx, gtol)

"""Generates tests/golden/models/{ctsma,tgat,tisasrec}_small.npz — golden vectors for the three regressive models.

As for tests/golden/make_fixtures.py: the reference (TensorFlow-1.x) cannot run offline and ships no vectors, so the fixtures
are produced by the float64 restatements (oracle/ctsma_ref.py, oracle/baselines_ref.py), which are pinned to the reference
source by tests/test_ctsma_oracle.py and tests/test_baselines_oracle.py.  A fixture is DATA: seeded tokens / timestamps /
weights and the expected training logits, loss, gradients and evaluation logits.

    python tests/golden/make_model_fixtures.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import baselines_ref as BR  # noqa: E402
from oracle import ctsma_ref as CR  # noqa: E402
from oracle import easydgl_oracle as O  # noqa: E402

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")
DIMS = dict(B=16, T=14, C=32, h=2, I=48, nb=2)


def inputs(seed, time_scale, zero_pad_ts=True):
    rng = np.random.default_rng(seed)
    B, T, I = DIMS["B"], DIMS["T"], DIMS["I"]
    tokens = rng.integers(1, I, size=(B, T + 1))
    tokens[0, :5] = 0
    tokens[1, :1] = 0
    ts = (9.5e8 + np.cumsum(rng.exponential(0.7 * time_scale, size=(B, T + 1)), axis=1)).astype(np.float32)
    if zero_pad_ts:   # left padding carries timestamp 0 (data/linkpred.py:152-153)
        ts[tokens == 0] = 0.0
    return rng, tokens, ts


def perturbed(params, rng, keep=()):
    out = {}
    for k, v in params.items():
        if v.ndim == 1 and not any(s in k for s in keep):
            v = v + 0.05 * rng.standard_normal(v.shape)
        out[k] = v.astype(np.float32).astype(np.float64)
    return out


def pack(name, spec, tokens, ts, params, loss, logits, elogits, p64, extra=None):
    out = dict(tokens=tokens, ts=ts, train_logits=logits.detach().numpy(), loss=np.float64(float(loss.detach())),
               eval_logits=elogits.detach().numpy(), spec=np.array([repr(spec)]))
    for k, v in params.items():
        out["param:" + k] = v
        out["grad:" + k] = p64[k].grad.numpy()
    out.update(extra or {})
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", float(loss.detach()))


def main():
    os.makedirs(HERE, exist_ok=True)
    d = DIMS
    # ---- CTSMA
    # CTSMA's regulariser multiplies RAW forward differences of seqs_t (CTSMA.py:101-112): with zero timestamps on the padding
    # the first real event contributes a ~1e9 s span and the loss is all regulariser; keep running timestamps here so that the
    # fixture exercises every term at comparable magnitude
    rng, tokens, ts = inputs(101, 3600.0, zero_pad_ts=False)
    E = 4
    params = perturbed(CR.init_params(d["I"], d["T"], d["C"], d["h"], E, d["nb"], rng), rng)
    mt = O.synthetic_mark_table(d["I"], E, multi_hot=True)
    spec = dict(model="CTSMA", C=d["C"], h=d["h"], num_blocks=d["nb"], time_scale=3600.0, ct_reg=1e-2, l2_reg=1e-3, E=E, I=d["I"], T=d["T"])
    p64 = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
    feats = {"seqs_i": tokens[:, :-1], "seqs_t": ts}
    kw = dict(C=d["C"], h=d["h"], num_blocks=d["nb"], time_scale=3600.0)
    loss, aux = CR.train_loss(p64, mt, feats, tokens[:, 1:], ct_reg=1e-2, l2_reg=1e-3, **kw)
    loss.backward()
    el = CR.eval_logits({k: v.detach() for k, v in p64.items()}, mt, feats, **kw)
    pack("ctsma_small", spec, tokens, ts, params, loss, aux["logits"], el, p64,
         dict(mark_table=mt, **{f"lam_{i}": l.detach().numpy() for i, l in enumerate(aux["lams"])}))
    # ---- TGAT
    rng, tokens, ts = inputs(102, 86400.0)
    params = perturbed(BR.tgat_init_params(d["I"], d["T"], d["C"], d["nb"], rng), rng, keep=("basis_freq",))
    spec = dict(model="TGAT", C=d["C"], h=d["h"], nb=d["nb"], time_scale=86400.0, l2_reg=1e-3, I=d["I"], T=d["T"])
    p64 = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
    feats = {"seqs_i": tokens[:, :-1], "seqs_t": ts}
    kw = dict(C=d["C"], h=d["h"], nb=d["nb"], time_scale=86400.0)
    loss, aux = BR.tgat_train_loss(p64, feats, tokens[:, 1:], l2_reg=1e-3, **kw)
    loss.backward()
    el = BR.tgat_eval_logits({k: v.detach() for k, v in p64.items()}, feats, **kw)
    pack("tgat_small", spec, tokens, ts, params, loss, aux["logits"], el, p64)
    # ---- TiSASRec
    rng, tokens, ts = inputs(103, 86400.0)
    timelen = 16
    params = perturbed(BR.tisasrec_init_params(d["I"], timelen, d["C"], d["nb"], rng), rng)
    spec = dict(model="TiSASREC", C=d["C"], h=d["h"], nb=d["nb"], time_scale=86400.0, timelen=timelen, l2_reg=1e-3, I=d["I"], T=d["T"])
    p64 = {k: torch.tensor(v, requires_grad=True) for k, v in params.items()}
    feats = {"seqs_i": tokens[:, :-1], "seqs_t": ts}
    kw = dict(C=d["C"], h=d["h"], nb=d["nb"], time_scale=86400.0, timelen=timelen)
    loss, aux = BR.tisasrec_train_loss(p64, feats, tokens[:, 1:], l2_reg=1e-3, **kw)
    loss.backward()
    el = BR.tisasrec_eval_logits({k: v.detach() for k, v in p64.items()}, feats, **kw)
    pack("tisasrec_small", spec, tokens, ts, params, loss, aux["logits"], el, p64)


if __name__ == "__main__":
    main()

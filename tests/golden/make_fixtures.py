"""Generates the committed golden fixtures (tests/golden/*.npz).

The reference (TensorFlow-1.x) cannot run offline and ships no vectors of its own, so these fixtures are
produced by the fp64 oracle (oracle/easydgl_oracle.py + oracle/torch_ref.py for gradients), which is pinned
to the reference source by tests/test_oracle_kat.py and tests/test_oracle_grad.py.  A fixture is DATA:
seeded inputs, weights and the expected logits / loss / intensities / gradients / top-K.

    python tests/golden/make_fixtures.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import easydgl_oracle as O  # noqa: E402
from oracle import torch_ref as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # BASELINE.json configs[0]-like plumbing case: 1K items, L=20, d=32
    "config1_small": dict(num_items=1000, seqslen=20, num_units=32, num_heads=2, num_blocks=1, masklen=4,
                          time_scale=86400.0, ct_reg=1e-2, l2_reg=1e-3, num_events=4, batch=6, multi_hot=True),
    # two blocks, dh=16, odd T
    "two_blocks": dict(num_items=160, seqslen=12, num_units=32, num_heads=2, num_blocks=2, masklen=3,
                       time_scale=3600.0, ct_reg=5e-3, l2_reg=1e-3, num_events=5, batch=4, multi_hot=True),
}


def build(name, spec, seed):
    spec = dict(spec)
    batch, multi_hot = spec.pop("batch"), spec.pop("multi_hot")
    cfg = O.Config(**spec)
    rng = np.random.default_rng(seed)
    params = O.init_params(cfg, rng, perturb=True)
    mt = O.synthetic_mark_table(cfg.num_items, cfg.num_events, multi_hot=multi_hot)
    ids, ts = O.synthetic_sequences(cfg, batch, rng, min_len=3)
    mp = O.draw_masked_positions(cfg, batch, rng)
    feats, labels = O.mask_random(cfg, ids, ts, mp)
    efeats, elabels = O.mask_last(cfg, ids, ts)
    loss, aux = O.train_loss(cfg, params, mt, feats, labels)
    p64 = R.to_torch_params(params)
    tl, _ = R.train_loss(cfg, p64, mt, feats, labels)
    tl.backward()
    assert abs(float(tl.detach()) - loss) < 1e-10 * max(1, abs(loss))
    elogits, _ = O.forward(cfg, params, mt, efeats, False)
    metrics, topk = O.evaluate(cfg, params, mt, efeats, elabels)
    out = dict(ids=ids, ts=ts, masked_positions=mp, labels=labels, mark_table=mt,
               train_logits=aux["logits"], loss=np.float64(loss), ce=np.float64(aux["ce"]), reg=np.float64(aux["reg"]),
               eval_logits=elogits, eval_topk=topk,
               metrics=np.array([metrics[k] for k in ("H10", "H50", "H100", "N10", "N50", "N100")]))
    for i, lam in enumerate(aux["lams"]):
        out[f"lam_{i}"] = lam
    for k, v in params.items():
        out["param:" + k] = v
        out["grad:" + k] = p64[k].grad.numpy()
    out["cfg"] = np.array([repr(spec)])
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print(name, "loss", loss)


if __name__ == "__main__":
    for i, (name, spec) in enumerate(CASES.items()):
        build(name, spec, 1234 + i)

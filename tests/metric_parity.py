#!/usr/bin/env python
"""Ranking-metric parity proxy for north_star's "HR@50 / NDCG@50 reproduce to +-0.001".

The Netflix files are external downloads, so the published metric table cannot be re-measured here.  What CAN be measured is
whether the arithmetic of this build moves the reference's ranking metrics: ONE initialisation and ONE stream of masked
batches are trained, dropout off, by three implementations of the same model —
    ref    the fp64 restatement of the TensorFlow graph with TF-form Adam (oracle/torch_ref.py; TEST INFRASTRUCTURE)
    f32    the HIP kernels in float32 (exact-f32 MFMA path)
    bf16   the HIP kernels with bf16 activations / f32 accumulation (the benchmarked mode)
— then each trained model ranks the full catalogue for a held-out set (last position masked, seen items masked, Base.py:
150-207) and HR@{10,50,100} / NDCG@{10,50,100} are compared.  Synthetic sequences with a learnable structure (a noisy item
-> item transition map over a Zipf popularity prior) keep the metrics away from 0 and 1.

    python tests/metric_parity.py [--out profiles/r02_metric_parity.json]      (under tests/: it drives the oracle)
"""
import argparse
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def make_sequences(cfg, n, rng, follow=0.6):
    """Left-padded records [n, T]: item[t+1] = perm[item[t]] with probability `follow`, else a Zipf draw."""
    from oracle import easydgl_oracle as O
    ids, ts = O.synthetic_sequences(cfg, n, rng, min_len=max(5, cfg.T // 2))
    perm = np.concatenate([[0], 1 + np.random.default_rng(7).permutation(cfg.num_items - 1)])
    for b in range(n):
        nz = np.nonzero(ids[b])[0]
        for t in nz[1:]:
            if rng.random() < follow:
                ids[b, t] = perm[ids[b, t - 1]]
    return ids, ts


def run(num_items=400, seqslen=20, num_units=32, num_heads=2, num_blocks=1, masklen=4, num_events=4, batch=128, steps=24,
        n_eval=2048, lr=2e-3, seed=11, modes=("f32", "bf16")):
    import easydgl_amd
    from oracle import easydgl_oracle as O
    from oracle import torch_ref as R
    cfg = O.Config(num_items=num_items, seqslen=seqslen, num_units=num_units, num_heads=num_heads, num_blocks=num_blocks,
                   masklen=masklen, time_scale=86400.0, ct_reg=1e-3, l2_reg=1e-4, learning_rate=lr, num_events=num_events)
    rng = np.random.default_rng(seed)
    params0 = O.init_params(cfg, rng)
    mt = O.synthetic_mark_table(cfg.num_items, cfg.num_events, multi_hot=True)
    tr_i, tr_t = make_sequences(cfg, batch * steps, rng)
    ev_i, ev_t = make_sequences(cfg, n_eval, rng)
    batches = []
    for s in range(steps):
        sl = slice(s * batch, (s + 1) * batch)
        batches.append(O.mask_random(cfg, tr_i[sl], tr_t[sl], O.draw_masked_positions(cfg, batch, rng)))
    efeats, elabels = O.mask_last(cfg, ev_i, ev_t)

    out = {"config": dict(num_items=num_items, seqslen=seqslen, num_units=num_units, num_heads=num_heads, num_blocks=num_blocks,
                          masklen=masklen, num_events=num_events, batch=batch, steps=steps, n_eval=n_eval, learning_rate=lr,
                          dropout=0.0, l2_reg=cfg.l2_reg, ct_reg=cfg.ct_reg)}
    # ---- reference arithmetic: fp64, TF-form Adam --------------------------------------------------------------------
    p64 = R.to_torch_params(params0)
    opt = R.TFAdam(p64, cfg.learning_rate)
    ref_losses = []
    for feats, labels in batches:
        loss, _ = R.train_loss(cfg, p64, mt, feats, labels)
        loss.backward()
        opt.step()
        ref_losses.append(float(loss))
    trained = {k: v.detach().numpy() for k, v in p64.items()}
    mets = {k: [] for k in ("H10", "H50", "H100", "N10", "N50", "N100")}
    for lo in range(0, n_eval, 256):
        ef = {k: v[lo:lo + 256] for k, v in efeats.items()}
        probs = O.eval_scores(cfg, trained, mt, ef, True)
        per = O.ranking_metrics(O.top_k(probs, 100), elabels[lo:lo + 256, -1])
        for k in mets:
            mets[k].append(per[k])
    out["ref"] = {"loss_first": ref_losses[0], "loss_last": ref_losses[-1], **{k: float(np.concatenate(v).mean()) for k, v in mets.items()}}
    # ---- the HIP kernels ----------------------------------------------------------------------------------------------
    for mode in modes:
        F = SimpleNamespace(model="EasyDGL", num_items=cfg.num_items, num_units=cfg.num_units, num_heads=cfg.num_heads,
                            num_blocks=cfg.num_blocks, seqslen=cfg.seqslen, masklen=cfg.masklen, time_scale=cfg.time_scale,
                            learning_rate=cfg.learning_rate, l2_reg=cfg.l2_reg, ct_reg=cfg.ct_reg, hidden_dropout_rate=0.0,
                            attention_probs_dropout_rate=0.0, mark_table=mt, compute_dtype=mode, num_train_steps=None,
                            num_warmup_steps=None)
        m = easydgl_amd.ranking(F).finalize("cuda")
        m.load_tf_variables(params0)
        losses = []
        for feats, labels in batches:
            f = {k: torch.as_tensor(np.asarray(v)).cuda().contiguous() for k, v in feats.items()}
            losses.append(float(m.train_step(f, torch.as_tensor(labels).cuda())))
        m.reset_metrics()
        for lo in range(0, n_eval, 256):
            ef = {k: torch.as_tensor(np.asarray(v[lo:lo + 256])).cuda().contiguous() for k, v in efeats.items()}
            m.eval_step(ef, torch.as_tensor(elabels[lo:lo + 256]).cuda(), mask_seen=True)
        got = m.metrics()
        out[mode] = {"loss_first": losses[0], "loss_last": losses[-1], **got,
                     "delta_vs_ref": {k: got[k] - out["ref"][k] for k in got}}
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--headline", action="store_true", help="headline widths (num_units 128, 8 heads, seqslen 100, 2000 items)")
    a = ap.parse_args()
    kw = dict(num_items=2000, seqslen=100, num_units=128, num_heads=8, masklen=20, num_events=16, batch=64, steps=24, n_eval=2048) if a.headline else {}
    res = run(**kw)
    txt = json.dumps(res, indent=1)
    print(txt)
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")

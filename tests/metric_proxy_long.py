#!/usr/bin/env python
"""Longer ranking-metric proxy for north_star's "HR@50 / NDCG@50 reproduce to +-0.001": the published recipe's widths and
regularisers (runme.sh:15-23: num_units 512, 8 heads, 1 block, seqslen 30 -> T = 31, masklen 6, hidden / attention dropout 0.1,
learning rate 5e-4, l2 1e-4, ct_reg 1e-7, time_scale 86400) trained for 500 steps WITH DROPOUT ON, by
    ref    the fp64 restatement of the TensorFlow graph with TF-form Adam and torch's dropout (oracle/torch_ref.py; TEST
           INFRASTRUCTURE), three dropout seeds;
    bf16   the HIP kernels (benchmarked mode, counter-based device dropout), three dropout seeds;
    f32    the HIP kernels in float32, three dropout seeds
from ONE initialisation over ONE stream of masked batches; each trained model ranks the full catalogue for a held-out set (last
position masked, seen items masked, Base.py:150-207).  The dropout streams differ by construction (torch's Philox on the CPU, a
hash of (seed, step, stream, element) on the device), so runs are compared as DISTRIBUTIONS: the spread over seeds within an
implementation against the shift between implementations.  What stays scaled down (the Netflix files are external downloads):
3000 items and batch 128 instead of 17771 / 512, synthetic sequences with a learnable item -> item structure.

    python tests/metric_proxy_long.py oracle [--steps 500]      CPU only, ~1 s / step: writes tests/golden/metric_proxy_oracle.json
    python tests/metric_proxy_long.py hip [--out profiles/r03_metric_proxy_long.json]     on the GPU box: reads that fixture
"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FIXTURE = os.path.join(ROOT, "tests", "golden", "metric_proxy_oracle.json")
SHAPE = dict(num_items=3000, seqslen=30, num_units=512, num_heads=8, num_blocks=1, masklen=6, num_events=16)
RECIPE = dict(learning_rate=5e-4, l2_reg=1e-4, ct_reg=1e-7, hidden_drop=0.1, att_drop=0.1, time_scale=86400.0)
BATCH, N_EVAL, DATA_SEED = 128, 2048, 31
KEYS = ("H10", "H50", "H100", "N10", "N50", "N100")


def problem(steps):
    from oracle import easydgl_oracle as O
    from tests.metric_parity import make_sequences
    cfg = O.Config(time_scale=RECIPE["time_scale"], ct_reg=RECIPE["ct_reg"], l2_reg=RECIPE["l2_reg"],
                   learning_rate=RECIPE["learning_rate"], **SHAPE)
    rng = np.random.default_rng(DATA_SEED)
    params0 = O.init_params(cfg, rng)
    mt = O.synthetic_mark_table(cfg.num_items, cfg.num_events, multi_hot=True)
    tr_i, tr_t = make_sequences(cfg, BATCH * steps, rng)
    ev_i, ev_t = make_sequences(cfg, N_EVAL, rng)
    batches = []
    for s in range(steps):
        sl = slice(s * BATCH, (s + 1) * BATCH)
        batches.append(O.mask_random(cfg, tr_i[sl], tr_t[sl], O.draw_masked_positions(cfg, BATCH, rng)))
    efeats, elabels = O.mask_last(cfg, ev_i, ev_t)
    return cfg, params0, mt, batches, efeats, elabels


def run_oracle(prob, drop_seed, log_every=50):
    from oracle import easydgl_oracle as O
    from oracle import torch_ref as R
    cfg, params0, mt, batches, efeats, elabels = prob
    torch.manual_seed(drop_seed)
    p64 = R.to_torch_params(params0)
    opt = R.TFAdam(p64, cfg.learning_rate)
    losses, t0 = [], time.time()
    for s, (feats, labels) in enumerate(batches):
        loss, _ = R.train_loss(cfg, p64, mt, feats, labels, hidden_drop=RECIPE["hidden_drop"], att_drop=RECIPE["att_drop"])
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        if log_every and (s + 1) % log_every == 0:
            print(f"  oracle seed {drop_seed} step {s + 1}: loss {losses[-1]:.4f}  ({(time.time() - t0) / (s + 1):.2f} s/step)", flush=True)
    trained = {k: v.detach().numpy() for k, v in p64.items()}
    mets = {k: [] for k in KEYS}
    for lo in range(0, N_EVAL, 256):
        ef = {k: v[lo:lo + 256] for k, v in efeats.items()}
        per = O.ranking_metrics(O.top_k(O.eval_scores(cfg, trained, mt, ef, True), 100), elabels[lo:lo + 256, -1])
        for k in KEYS:
            mets[k].append(per[k])
    return {"seed": drop_seed, "loss_first": losses[0], "loss_last10": float(np.mean(losses[-10:])),
            **{k: float(np.concatenate(v).mean()) for k, v in mets.items()}}


def run_hip(prob, mode, drop_seed):
    import easydgl_amd
    cfg, params0, mt, batches, efeats, elabels = prob
    F = SimpleNamespace(model="EasyDGL", num_items=cfg.num_items, num_units=cfg.num_units, num_heads=cfg.num_heads,
                        num_blocks=cfg.num_blocks, seqslen=cfg.seqslen, masklen=cfg.masklen, time_scale=cfg.time_scale,
                        learning_rate=cfg.learning_rate, l2_reg=cfg.l2_reg, ct_reg=cfg.ct_reg, hidden_dropout_rate=RECIPE["hidden_drop"],
                        attention_probs_dropout_rate=RECIPE["att_drop"], mark_table=mt, compute_dtype=mode, num_train_steps=None,
                        num_warmup_steps=None, seed=drop_seed)
    m = easydgl_amd.ranking(F).finalize("cuda")
    m.load_tf_variables(params0)                      # ONE initialisation: the seed only moves the dropout stream
    losses = []
    for feats, labels in batches:
        f = {k: torch.as_tensor(np.asarray(v)).cuda().contiguous() for k, v in feats.items()}
        losses.append(m.train_step(f, torch.as_tensor(labels).cuda()))
    losses = [float(x) for x in losses]
    m.reset_metrics()
    for lo in range(0, N_EVAL, 256):
        ef = {k: torch.as_tensor(np.asarray(v[lo:lo + 256])).cuda().contiguous() for k, v in efeats.items()}
        m.eval_step(ef, torch.as_tensor(elabels[lo:lo + 256]).cuda(), mask_seen=True)
    got = m.metrics()
    return {"seed": drop_seed, "loss_first": losses[0], "loss_last10": float(np.mean(losses[-10:])), **{k: float(got[k]) for k in KEYS}}


def summarise(runs):
    return {k: {"mean": float(np.mean([r[k] for r in runs])), "min": float(min(r[k] for r in runs)), "max": float(max(r[k] for r in runs))}
            for k in KEYS + ("loss_last10",)}


def compare(res):
    """Shift between implementations (difference of the seed means) next to the spread over seeds within each."""
    out = {}
    ref = res["ref"]["summary"]
    for mode in ("bf16", "f32"):
        if mode not in res:
            continue
        s = res[mode]["summary"]
        out[mode] = {k: {"shift": s[k]["mean"] - ref[k]["mean"],
                         "spread_ref": ref[k]["max"] - ref[k]["min"], "spread_hip": s[k]["max"] - s[k]["min"]} for k in s}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["oracle", "hip"])
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--seeds", type=int, nargs="+", default=None, help="dropout seeds (default: oracle 101 202 303; hip: the fixture's)")
    ap.add_argument("--merge", default=None, help="oracle: a second result file whose runs are appended to the fixture")
    ap.add_argument("--modes", nargs="+", default=["bf16", "f32"])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    if a.what == "oracle" and a.merge:
        res, more = json.load(open(a.out or FIXTURE)), json.load(open(a.merge))
        assert res["config"] == more["config"]
        res["runs"] += [r for r in more["runs"] if r["seed"] not in {x["seed"] for x in res["runs"]}]
        res["summary"] = summarise(res["runs"])
        with open(a.out or FIXTURE, "w") as f:
            f.write(json.dumps(res, indent=1) + "\n")
        print(json.dumps(res["summary"], indent=1))
        return
    if a.what == "oracle":
        prob = problem(a.steps)
        runs = [run_oracle(prob, s) for s in (a.seeds or [101, 202, 303])]
        res = {"config": dict(**SHAPE, **RECIPE, batch=BATCH, steps=a.steps, n_eval=N_EVAL, data_seed=DATA_SEED),
               "generated_by": "python tests/metric_proxy_long.py oracle", "runs": runs, "summary": summarise(runs)}
        with open(a.out or FIXTURE, "w") as f:
            f.write(json.dumps(res, indent=1) + "\n")
        print(json.dumps(res["summary"], indent=1))
        return
    ref = json.load(open(FIXTURE))
    prob = problem(ref["config"]["steps"])
    res = {"config": ref["config"], "ref": {"runs": ref["runs"], "summary": ref["summary"]}}
    seeds = a.seeds or [r["seed"] for r in ref["runs"]]
    for mode in a.modes:
        runs = [run_hip(prob, mode, s) for s in seeds]
        res[mode] = {"runs": runs, "summary": summarise(runs)}
    res["comparison"] = compare(res)
    txt = json.dumps(res, indent=1)
    print(json.dumps(res["comparison"], indent=1))
    if a.out:
        with open(a.out, "w") as f:
            f.write(txt + "\n")


if __name__ == "__main__":
    main()

"""Train / evaluate driver — the role of the reference's ``src/main.py:78-151`` with ``util.EarlyStopping``
(``src/util.py:14-58``), on the HIP model.

Models: EasyDGL (masked batches, ``MAUPostProcessor``) and CTSMA / TGAT / TiSASREC (regressive batches,
``RegressivePostProcessor``), chosen as ``util.reader`` does (``util.py:99-129``).
Same flags as ``main.py:22-75`` for everything these paths read (``--train/--valid/--test`` file patterns,
``--num_items --num_units --num_heads --num_blocks --seqslen --time_scale --masklen --mark --ct_reg --batch_size
--num_epochs --learning_rate --l2_reg --hidden_dropout_rate --attention_probs_dropout_rate --eval_per_steps
--mask_seen``).  Epoch structure as in the reference: one pass over the training records (masked on the device, one
kernel per batch), then validation and test passes with the last position masked, model selection on validation H@100,
patience 10, stop on a NaN loss, best checkpoint kept.  Differences, stated: records are shuffled with a full
permutation per epoch (the reference: file-order shuffle + a 64-record buffer, ``dataloader.py:222-236``); full batches
run through the static ``TrainEngine`` and the remainder batch through ``model.train_step``.

``python -m easydgl_amd.train --model EasyDGL --train 'data/train*.tfrec' --valid data/validation.tfrec
--test data/test.tfrec --num_items 17770 --mark data/mark.pkl ...`` (``.npz`` files from ``formats.convert`` work too).
"""
from __future__ import annotations

import argparse
import logging
import math
import os
from types import SimpleNamespace
from typing import Dict, Optional

import numpy as np


def args(argv=None):
    p = argparse.ArgumentParser(description="EasyDGL (self-modulating attention) on MI355X — train / evaluate")
    p.add_argument("--train", required=True, help="training data file patterns (.tfrec or .npz)")
    p.add_argument("--valid", required=True)
    p.add_argument("--test", required=True)
    p.add_argument("--model", required=True, help="algorithm name: EasyDGL, CTSMA, TGAT or TiSASREC (util.ranking keys)")
    p.add_argument("--num_items", type=int, required=True)
    # reference default: 50 (main.py:35) — head dim 50 with the default single head: every model runs it zero-padded at head dim 64
    # (model/base.py, model/easydgl.py: exact, the padded channels stay zero); every published recipe passes --num_units=512
    # (runme.sh:15-115)
    p.add_argument("--num_units", type=int, default=50)
    p.add_argument("--num_heads", type=int, default=1)
    p.add_argument("--num_blocks", type=int, default=3)
    p.add_argument("--seqslen", type=int, default=30)
    p.add_argument("--time_scale", type=float, default=1)
    p.add_argument("--masklen", type=int, default=6)
    p.add_argument("--timelen", type=int, default=256)
    p.add_argument("--mark", type=str, help="mark data file (pickled csr matrix or .npy)")
    p.add_argument("--ct_reg", type=float, default=0.0)
    p.add_argument("--batch_size", type=int, default=128)
    p.add_argument("--num_epochs", type=int, default=100)
    p.add_argument("--learning_rate", type=float, default=5e-4)
    p.add_argument("--l2_reg", type=float, default=0.0)
    p.add_argument("--hidden_dropout_rate", type=float, default=0.0)
    p.add_argument("--attention_probs_dropout_rate", type=float, default=0.0)
    p.add_argument("--eval_per_steps", type=int, default=1)
    p.add_argument("--mask_seen", action="store_true")
    # build-specific
    p.add_argument("--dtype", choices=["bf16", "f32"], default="bf16", help="activation dtype of the HIP kernels")
    p.add_argument("--ckpt_dir", default="ckpt")
    p.add_argument("--seed", type=int, default=9876)   # main.py:156-159
    p.add_argument("--patience", type=int, default=10)
    p.add_argument("--graph", action="store_true",
                   help="regressive models: replay the training step of full batches as one HIP graph (Sequential.graphed_train_step)")
    a = p.parse_args(argv)
    return a


class EarlyStopping:
    """``util.EarlyStopping`` (util.py:14-58), decision for decision: the first evaluation sets the reference point; an
    evaluation with ``acc < best_acc`` increments the counter (stop at ``patience``); otherwise the counter resets, the
    checkpoint is saved and every test metric whose validation value did not fall below the FIRST evaluation's value is
    refreshed (``best_valid`` is never updated in the reference — SURVEY Appendix B item 11 — kept as is).  NaN loss stops."""

    def __init__(self, model_name: str, patience: int = 10, saver=None):
        self.model = model_name
        self.patience = patience
        self.counter = 0
        self.res: Optional[Dict[str, float]] = None
        self.best_valid: Optional[Dict[str, float]] = None
        self.best_acc = None
        self.best_loss = None
        self.early_stop = False
        self.saver = saver

    def step(self, loss: float, acc: float, valid: Dict[str, float], test: Dict[str, float]) -> bool:
        if np.isnan(loss):
            self.early_stop = True
        elif self.best_loss is None:
            self.best_acc, self.best_loss = acc, loss
            self.best_valid, self.res = dict(valid), dict(test)
        elif acc < self.best_acc:
            self.counter += 1
            logging.info("EarlyStopping %s counter: %d out of %d", self.model, self.counter, self.patience)
            if self.counter >= self.patience:
                self.early_stop = True
        else:
            self.best_loss = min(loss, self.best_loss)
            self.best_acc = max(acc, self.best_acc)
            for k in self.res:
                if self.best_valid[k] <= valid[k]:
                    self.res[k] = test[k]
            self.counter = 0
            if self.saver is not None:
                self.saver()
        return self.early_stop

    def summary(self) -> Dict[str, float]:
        logging.info("SUMMARY: %s", {k: "{0:.5f}".format(v) for k, v in (self.res or {}).items()})
        return dict(self.res or {})


def save_checkpoint(model, path: str) -> None:
    """Parameters (flat f32 arena), Adam moments and step — everything a resumed run needs."""
    import torch
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    model.settle_state()       # (the static engine keeps the counters one step ahead: the file holds "steps taken")
    torch.save({"arena": model._arena.detach().cpu(), "adam_m": model._adam_m.cpu(), "adam_v": model._adam_v.cpu(),
                "adam_state": model._adam_state.cpu(), "rng_state": model._rng_state.cpu(),
                "offsets": dict(model._offsets)}, path)


def load_checkpoint(model, path: str) -> None:
    import torch
    ck = torch.load(path, map_location="cpu")
    if dict(ck["offsets"]) != dict(model._offsets):
        raise ValueError("checkpoint does not match the model's parameter layout")
    with torch.no_grad():
        model._arena.copy_(ck["arena"])
        model._adam_m.copy_(ck["adam_m"])
        model._adam_v.copy_(ck["adam_v"])
        model._adam_state.copy_(ck["adam_state"])
        model._rng_state.copy_(ck["rng_state"])
    model._state_ahead = False     # the file holds "steps taken" (save_checkpoint settles the engine's look-ahead)
    model.sync_shadow()


def regressive_batch(tok, tim, is_training: bool):
    """RegressivePostProcessor (dataloader.py:88-108; util.reader's choice for CTSMA / TGAT / TiSASREC, util.py:116-129):
    features see tokens[:-1] and every timestamp; labels = tokens[1:] (training) / the whole record (evaluation)."""
    return {"seqs_i": tok[:, :-1].contiguous(), "seqs_t": tim}, (tok[:, 1:].contiguous() if is_training else tok)


def evaluate(model, ids, ts, batch_size: int, mask_seen: bool) -> Dict[str, float]:
    """One pass of ``Sequential.eval`` (Base.py:150-207) over a split: last position masked (EasyDGL) or predicted from the
    prefix (regressive models), streaming means."""
    import torch
    from . import data as D
    model.reset_metrics()
    n = ids.shape[0]
    masked = hasattr(model, "mask")
    for lo in range(0, n, batch_size):
        tok = torch.as_tensor(ids[lo:lo + batch_size]).cuda()
        tim = torch.as_tensor(ts[lo:lo + batch_size]).cuda()
        feats, labels = D.device_mask_last(tok, tim, model.mask) if masked else regressive_batch(tok, tim, False)
        model.eval_step(feats, labels, mask_seen=mask_seen)
    return model.metrics()


def run(FLAGS) -> Dict[str, float]:
    import torch
    from . import data as D
    from . import formats as F
    from .engine import TrainEngine
    from .util import ranking

    torch.manual_seed(FLAGS.seed)
    rng = np.random.default_rng(FLAGS.seed)
    logging.info("1. read data")
    tr_i, tr_t = F.load_sequences(FLAGS.train, FLAGS.seqslen)
    vl_i, vl_t = F.load_sequences(FLAGS.valid, FLAGS.seqslen)
    te_i, te_t = F.load_sequences(FLAGS.test, FLAGS.seqslen)
    logging.info("   train %d, valid %d, test %d sequences of %d positions", len(tr_i), len(vl_i), len(te_i), tr_i.shape[1])

    logging.info("2. create neural model")
    if getattr(FLAGS, "mark", None) and isinstance(FLAGS.mark, str):
        FLAGS.mark_table = F.load_mark_table(FLAGS.mark, FLAGS.num_items)
    FLAGS.compute_dtype = getattr(FLAGS, "dtype", "bf16")
    model = ranking(FLAGS)
    model.finalize(torch.device("cuda", torch.cuda.current_device()))
    bs = FLAGS.batch_size
    masked = FLAGS.model == "EasyDGL"          # MAUPostProcessor; every other model is fed regressively (util.py:99-129)
    if FLAGS.model == "TGAT":                  # the time feature map needs time-sorted sequences (model/tgat.py)
        for name, (ii, tt) in (("train", (tr_i, tr_t)), ("valid", (vl_i, vl_t)), ("test", (te_i, te_t))):
            if np.any((np.diff(tt, axis=1) < 0) & (ii[:, :-1] != 0)):
                raise ValueError(f"{name}: timestamps decrease inside a sequence; TGAT needs time-sorted records")
    engine = TrainEngine(model, bs, use_graph=False) if (masked and len(tr_i) >= bs) else None
    if engine is not None:
        # the loss is read every 10 batches (NaN test) and once per epoch: the engine leaves its loss launches off the step
        # (they ride with the next step) and adds the step losses up on the device — the mode bench.py times
        engine.sync_loss = False
        engine.accumulate_loss = True
    ckpt = os.path.join(FLAGS.ckpt_dir, f"{FLAGS.model}.pt")
    stopper = EarlyStopping(FLAGS.model, patience=FLAGS.patience, saver=lambda: save_checkpoint(model, ckpt))
    mask_state = torch.tensor([FLAGS.seed, 0], dtype=torch.int64, device="cuda")   # (seed, batch counter) of the masker
    gstep = None

    logging.info("3. train and evaluate model")
    for epoch in range(FLAGS.num_epochs):
        order = rng.permutation(len(tr_i))
        # streaming epoch mean of the step losses, as tf.metrics.mean feeds EarlyStopping in the reference (Base.py:133-134,
        # main.py:119-122); accumulated on the device, read back every 10 batches for the NaN test and once per epoch
        loss_sum = torch.zeros((), device="cuda", dtype=torch.float64)
        running_loss, nb = float("nan"), 0
        if engine is not None:
            engine.join_loss()
            engine.loss_sum.zero_()
        for lo in range(0, len(order), bs):
            idx = order[lo:lo + bs]
            tok = torch.as_tensor(tr_i[idx]).cuda()
            tim = torch.as_tensor(tr_t[idx]).cuda()
            if masked:
                feats, labels = D.device_mask_random(tok, tim, model.mask, FLAGS.masklen, mask_state)
                mask_state[1] += 1
            else:
                feats, labels = regressive_batch(tok, tim, True)
            if engine is not None and len(idx) == bs:
                engine.step(feats, labels)
                loss = None      # (added to engine.loss_sum by the engine's own loss launches)
            elif not masked and getattr(FLAGS, "graph", False) and len(idx) == bs:
                if gstep is None:
                    # the warm-up step that precedes the capture IS this batch's optimizer step: no replay for it
                    gstep = model.graphed_train_step(feats, labels, warmup=1)
                    loss = gstep.warmup_loss
                else:
                    loss = gstep(feats, labels)
            else:
                loss = model.train_step(feats, labels)
            nb += 1
            if loss is not None:
                loss_sum += loss.reshape(()).double()
            if nb % 10 == 0 or lo + bs >= len(order):
                if engine is not None:
                    engine.join_loss()       # orders this stream behind the (deferred) loss launches of the last step
                running_loss = (float(loss_sum) + (float(engine.loss_sum) if engine is not None else 0.0)) / nb
                if math.isnan(running_loss):
                    break
        logging.info("%03d: Loss=%.4f", epoch, running_loss)
        if hasattr(model, "check_inputs"):
            model.check_inputs()
        if math.isnan(running_loss):      # util.py:29-30: a NaN loss stops the run, whatever eval_per_steps says
            stopper.step(running_loss, 0.0, {}, {})
            break
        if epoch % FLAGS.eval_per_steps:
            continue
        vl = evaluate(model, vl_i, vl_t, bs, FLAGS.mask_seen)
        logging.info("%03d: %s", epoch, {k: "{0:.5f}".format(v) for k, v in vl.items()})
        te = evaluate(model, te_i, te_t, bs, FLAGS.mask_seen)
        if stopper.step(running_loss, vl["H100"], vl, te):
            break
    return stopper.summary()


def main(argv=None):
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    FLAGS = args(argv)
    return run(SimpleNamespace(**vars(FLAGS)))


if __name__ == "__main__":
    main()

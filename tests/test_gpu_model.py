"""Whole-model parity: the HIP EasyDGL (through the reference's model interface) vs the fp64 oracle on the
same seeded inputs and weights.  Stated tolerances: f32 path rtol 1e-4 on logits / loss, 1e-3 on gradients;
bf16 path <= 2e-2 relative on logits / lambda, 5e-3 on the loss, and every gradient tensor within 2e-2 relative L2 AND 5e-2 of its
largest entry (tests/_util.py GRAD_TOL)."""
import numpy as np
import pytest
import torch

from oracle import easydgl_oracle as O
from oracle import torch_ref as R
from tests._util import GRAD_TOL, LOSS_TOL, assert_close, build_model, grad_ok, make_problem, rel_err, to_dev

pytestmark = pytest.mark.gpu

CASES = [
    dict(),                                                                          # C=32 h=2 (dh=16) nb=2 E=4 T=11
    dict(num_units=64, num_heads=2, num_blocks=1, seqslen=30, masklen=6, num_events=7, num_items=300),   # dh=32, T=31
    dict(num_units=128, num_heads=8, num_blocks=1, seqslen=100, masklen=20, num_events=16, num_items=2000),  # headline shape
    dict(num_units=64, num_heads=4, num_blocks=1, seqslen=200, masklen=40, num_events=7, num_items=500),     # config-3 length (T=201)
    # the published EasyDGL recipe (runme.sh:15-23 + the defaults main.py:38,44): C=512, h=8 (dh=64), 1 block, T=31, M=6
    dict(num_units=512, num_heads=8, num_blocks=1, seqslen=30, masklen=6, num_events=16, num_items=700),
    # more than 16 mark types (EasyDGL.py:46 takes the count from the mark table): groups of 16 (temporal.modulated_attention)
    dict(num_units=64, num_heads=4, num_blocks=2, seqslen=20, masklen=5, num_events=24, num_items=300),      # 16 + 8, dh = 16
    dict(num_units=128, num_heads=2, num_blocks=1, seqslen=18, masklen=4, num_events=40, num_items=200),     # 16 + 16 + 8, dh = 64
]


@pytest.mark.parametrize("mode,ltol", [("f32", 1e-4), ("bf16", 2e-2)])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_forward_loss_and_gradients(mode, ltol, case):
    prob = make_problem(seed=10 + case, batch=4, **CASES[case])
    cfg = prob["cfg"]
    m = build_model(prob, mode)
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    # ---- EasyDGL.__call__(features, is_training=True): logits [B*M, I]
    logits = m(feats, True)
    want_logits, want_lams = O.forward(cfg, prob["params"], prob["mark_table"], prob["feats"], True)
    assert logits.shape == want_logits.shape
    assert_close(logits.detach().cpu().numpy(), want_logits, ltol, "train logits")
    assert float((logits[:, 0] + 1000).abs().max()) == 0.0
    for a, b in zip(m._last_lams, want_lams):
        assert_close(a.detach().cpu().numpy(), b, ltol, "lambda")
    # ---- the loss EasyDGL.train minimises, and its gradients
    m.zero_grad_arena()
    loss = m.train_loss(feats, labels)
    loss.backward()
    p64 = R.to_torch_params(prob["params"])
    ref_loss, _ = R.train_loss(cfg, p64, prob["mark_table"], prob["feats"], prob["labels"])
    ref_loss.backward()
    assert_close(loss.item(), ref_loss.item(), LOSS_TOL[mode], "train loss")
    bad = {}
    for name, p in m.tf_variable_map().items():
        ok, e = grad_ok(p.grad.cpu().numpy(), p64[name].grad.numpy(), mode)
        if not ok:
            bad[name] = e
    assert not bad, f"gradient mismatch (relative L2, max-abs / max-abs-ref) vs {GRAD_TOL[mode]}: {bad}"
    # ---- eval logits (last position)
    elog = m(to_dev(prob["efeats"]), False)
    want_e, _ = O.forward(cfg, prob["params"], prob["mark_table"], prob["efeats"], False)
    assert_close(elog.detach().cpu().numpy(), want_e, ltol, "eval logits")


# head dims the kernels do not tile run zero-padded at the next supported one: the reference's DEFAULT flags (main.py:35-37:
# --num_units 50 --num_heads 1 --num_blocks 3, seqslen 30, masklen 6), and a two-head case whose real channels interleave with pads
PADDED = [dict(num_units=50, num_heads=1, num_blocks=3, seqslen=30, masklen=6, num_events=5, num_items=200),
          dict(num_units=40, num_heads=2, num_blocks=1, seqslen=17, masklen=4, num_events=18, num_items=120)]


@pytest.mark.parametrize("mode,ltol", [("f32", 1e-4), ("bf16", 2e-2)])
@pytest.mark.parametrize("case", range(len(PADDED)))
def test_widths_the_kernels_do_not_tile_run_channel_padded(mode, ltol, case):
    """EasyDGL at the reference's default width against the fp64 oracle AT THAT WIDTH: logits, lambda, loss, every gradient (in the
    reference's shapes: tf_gradients strips the padding), eval logits; then optimizer steps with dropout on leave every padded
    entry of every parameter exactly zero, and three dropout-free steps follow the oracle's Adam trajectory."""
    c = PADDED[case]
    prob = make_problem(seed=70 + case, batch=5, **c)
    cfg = prob["cfg"]
    m = build_model(prob, mode)
    dh = c["num_units"] // c["num_heads"]
    assert m.pad == (32 if dh <= 32 else 64, dh) and m.num_units == c["num_heads"] * m.pad[0] and m.width_true == c["num_units"]
    got_vals = m.tf_values()
    for k, v in prob["params"].items():            # load -> read back: the reference's shapes and values
        assert tuple(got_vals[k].shape) == np.asarray(v).shape
        assert np.array_equal(got_vals[k].cpu().numpy(), np.asarray(v, dtype=np.float32))
    assert m.padded_leak() == 0.0
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    logits = m(feats, True)
    want_logits, want_lams = O.forward(cfg, prob["params"], prob["mark_table"], prob["feats"], True)
    assert logits.shape == want_logits.shape
    assert_close(logits.detach().cpu().numpy(), want_logits, ltol, "train logits")
    for a, b in zip(m._last_lams, want_lams):
        assert_close(a.detach().cpu().numpy(), b, ltol, "lambda")
    m.zero_grad_arena()
    loss = m.train_loss(feats, labels)
    loss.backward()
    p64 = R.to_torch_params(prob["params"])
    ref_loss, _ = R.train_loss(cfg, p64, prob["mark_table"], prob["feats"], prob["labels"])
    ref_loss.backward()
    assert_close(loss.item(), ref_loss.item(), LOSS_TOL[mode], "train loss")
    bad = {}
    for name, g in m.tf_gradients().items():
        ok, e = grad_ok(g.cpu().numpy(), p64[name].grad.numpy(), mode)
        if not ok:
            bad[name] = e
    assert not bad, f"gradient mismatch vs {GRAD_TOL[mode]}: {bad}"
    elog = m(to_dev(prob["efeats"]), False)
    want_e, _ = O.forward(cfg, prob["params"], prob["mark_table"], prob["efeats"], False)
    assert_close(elog.detach().cpu().numpy(), want_e, ltol, "eval logits")
    # ---- padded entries never leave zero (dropout on: hidden and attention)
    md = build_model(prob, mode, hidden_drop=0.1, att_drop=0.1)
    for _ in range(4):
        assert np.isfinite(float(md.train_step(feats, labels)))
    assert md.padded_leak() == 0.0
    if mode == "f32":
        mt = build_model(prob, "f32")
        q64 = R.to_torch_params(prob["params"])
        opt = R.TFAdam(q64, cfg.learning_rate)
        for step in range(3):
            got = float(mt.train_step(feats, labels))
            ref, _ = R.train_loss(cfg, q64, prob["mark_table"], prob["feats"], prob["labels"])
            ref.backward()
            opt.step()
            assert abs(got - float(ref)) <= 2e-4 * abs(float(ref)), (step, got, float(ref))
        for name, v in mt.tf_values().items():
            assert np.abs(v.cpu().numpy() - q64[name].detach().numpy()).max() < 3e-4, name
        assert mt.padded_leak() == 0.0


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_eval_metrics_and_topk(mode):
    prob = make_problem(seed=3, batch=16, num_items=600, seqslen=20, num_units=32, num_heads=2, num_blocks=1)
    cfg = prob["cfg"]
    m = build_model(prob, mode)
    ef, el = to_dev(prob["efeats"]), torch.as_tensor(prob["elabels"]).cuda()
    val, idx = m.eval_topk(ef, mask_seen=True)
    want_metrics, want_idx = O.evaluate(cfg, prob["params"], prob["mark_table"], prob["efeats"], prob["elabels"])
    got = idx.cpu().numpy()
    if mode == "f32":
        # ranking on logits == ranking on softmax probs away from fp ties
        agree = (got == want_idx).mean()
        assert agree > 0.98, agree
    else:
        overlap = np.mean([len(set(got[r, :50]) & set(want_idx[r, :50])) / 50 for r in range(got.shape[0])])
        assert overlap > 0.9, overlap
    seen = prob["efeats"]["seqs_i"]
    for r in range(got.shape[0]):
        assert not (set(got[r]) & set(seen[r]))          # Base.py:156-163
    m.reset_metrics()
    m.eval_step(ef, el)
    mets = m.metrics()
    per = O.ranking_metrics(got, prob["elabels"][:, -1])
    for k in mets:
        assert abs(mets[k] - per[k].mean()) < 1e-5
    if mode == "f32":
        for k in mets:
            assert abs(mets[k] - want_metrics[k]) <= 1.0 / 16 + 1e-6


def test_train_steps_follow_the_oracle_trajectory():
    """Three optimizer steps (dropout off): loss sequence and final weights vs the fp64 torch restatement
    with TF-form Adam (Base.py:142-144)."""
    prob = make_problem(seed=21, batch=6)
    cfg = prob["cfg"]
    m = build_model(prob, "f32")
    p64 = R.to_torch_params(prob["params"])
    opt = R.TFAdam(p64, cfg.learning_rate)
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    for step in range(3):
        got = float(m.train_step(feats, labels))
        ref, _ = R.train_loss(cfg, p64, prob["mark_table"], prob["feats"], prob["labels"])
        ref.backward()
        opt.step()
        assert abs(got - float(ref)) <= 2e-4 * abs(float(ref)), (step, got, float(ref))
    for name, p in m.tf_variable_map().items():
        # Adam's first steps move every weight by ~lr regardless of gradient scale, so compare absolutely
        d = np.abs(p.detach().cpu().numpy() - p64[name].detach().numpy()).max()
        assert d < 2e-4, (name, d)


def test_training_reduces_loss_with_dropout_bf16():
    prob = make_problem(seed=5, batch=32, num_items=400, seqslen=20, num_units=64, num_heads=4, num_blocks=1,
                        masklen=4, num_events=8, learning_rate=2e-3)
    m = build_model(prob, "bf16", hidden_drop=0.1, att_drop=0.1)
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    losses = [float(m.train_step(feats, labels)) for _ in range(30)]
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0] - 0.5, losses


@pytest.mark.parametrize("shards", [2, 8])
def test_sharded_eval_matches_unsharded(shards):
    """K7: row-sharded scoring + local top-K + merge == single-shard top-K (same values, same ids)."""
    prob = make_problem(seed=9, batch=12, num_items=1500, seqslen=20, num_units=32, num_heads=2, num_blocks=1)
    m = build_model(prob, "f32")
    ef = to_dev(prob["efeats"])
    v0, i0 = m.eval_topk(ef, mask_seen=True)
    v1, i1 = m.eval_topk_sharded(ef, mask_seen=True, world=shards)
    assert torch.equal(i0, i1)
    assert torch.allclose(v0, v1, rtol=0, atol=0)


def test_device_masker_matches_reference_semantics():
    """edgl_mask_random / edgl_mask_last against MAUPostProcessor's contract (dataloader.py:159-206): M distinct
    positions in [1, T), token := MASK there, labels = original tokens, everything else untouched; the oracle's
    mask_random reproduces the batch given the drawn positions; draws differ between steps and rows."""
    from easydgl_amd import data as D
    rng = np.random.default_rng(5)
    B, T, M, mask_id = 64, 31, 6, 1000
    tok = torch.tensor(rng.integers(0, 1000, size=(B, T)), dtype=torch.int64).cuda()
    ts = torch.tensor(rng.random((B, T)), dtype=torch.float32).cuda()
    state = torch.tensor([1234, 0], dtype=torch.int64).cuda()
    feats, labels = D.device_mask_random(tok, ts, mask_id, M, state)
    mp = feats["masked_positions"].cpu().numpy()
    assert mp.shape == (B, M) and mp.min() >= 1 and mp.max() <= T - 1
    assert all(len(set(r)) == M for r in mp)
    cfg = O.Config(num_items=mask_id, num_units=8, num_heads=2, num_blocks=1, seqslen=T - 1, masklen=M, num_events=2)
    want_f, want_l = O.mask_random(cfg, tok.cpu().numpy(), ts.cpu().numpy(), mp)
    np.testing.assert_array_equal(feats["seqs_i"].cpu().numpy(), want_f["seqs_i"])
    np.testing.assert_array_equal(labels.cpu().numpy(), want_l)
    # same state -> same draw; next step -> a different one; positions are spread over [1, T)
    f2, _ = D.device_mask_random(tok, ts, mask_id, M, state)
    assert torch.equal(f2["masked_positions"], feats["masked_positions"])
    state2 = torch.tensor([1234, 1], dtype=torch.int64).cuda()
    f3, _ = D.device_mask_random(tok, ts, mask_id, M, state2)
    assert not torch.equal(f3["masked_positions"], feats["masked_positions"])
    big, _ = D.device_mask_random(tok.repeat(64, 1), ts.repeat(64, 1), mask_id, M, state)
    hist = np.bincount(big["masked_positions"].cpu().numpy().ravel(), minlength=T)[1:]
    expect = 64 * B * M / (T - 1)
    assert hist.min() > 0.85 * expect and hist.max() < 1.15 * expect and hist.sum() == 64 * B * M
    # full-length mask: every position in [1, T)
    full, _ = D.device_mask_random(tok, ts, mask_id, T - 1, state)
    assert (np.sort(full["masked_positions"].cpu().numpy(), axis=1) == np.arange(1, T)).all()
    fe, le = D.device_mask_last(tok, ts, mask_id)
    we, wl = O.mask_last(cfg, tok.cpu().numpy(), ts.cpu().numpy())
    np.testing.assert_array_equal(fe["seqs_i"].cpu().numpy(), we["seqs_i"])
    np.testing.assert_array_equal(le.cpu().numpy(), wl)


def test_chunked_eval_scoring_matches_the_single_pass(monkeypatch):
    """ops.score_topk walks the catalogue in bounded logits tiles (no [B, I] tensor): with a tile of 1024 items a 5000-item
    catalogue takes five chunks and two merge levels; values and ids must equal the single-pass result exactly."""
    from easydgl_amd import ops
    prob = make_problem(seed=12, batch=12, num_items=5000, seqslen=20, num_units=32, num_heads=2, num_blocks=1)
    m = build_model(prob, "f32")
    ef = to_dev(prob["efeats"])
    v0, i0 = m.eval_topk(ef, mask_seen=True)
    monkeypatch.setattr(ops, "EVAL_TILE_BYTES", 4 * 12 * 1024)
    v1, i1 = m.eval_topk(ef, mask_seen=True)
    assert torch.equal(i0, i1) and torch.equal(v0, v1)
    monkeypatch.setattr(ops, "EVAL_TILE_BYTES", 4 * 12 * 1024 * 3)     # chunks of 3072: the last one is partial
    v2, i2 = m.eval_topk(ef, mask_seen=True, K=100)
    assert torch.equal(i0, i2) and torch.equal(v0, v2)


@pytest.mark.parametrize("blocks", [1, 2])
def test_inference_through_the_fused_block_tail_matches_the_unfused_modules(blocks, monkeypatch):
    """Sequential.eval's forward (Base.py:150-163) at a shape the fused per-sample block tail takes (bf16, C = 128, T = 101): the head
    rows of `EasyDGL.encoder` through csrc/k_tail.hip — one launch per block, as in the training engine — against the dense /
    LayerNorm modules it replaces (equal up to the last bf16 digit of the dense outputs), and the same top-K lists up to near-ties."""
    prob = make_problem(seed=31, batch=8, num_items=2000, seqslen=100, num_units=128, num_heads=8, num_blocks=blocks, masklen=20,
                        num_events=16)
    m = build_model(prob, "bf16")
    ef = to_dev(prob["efeats"])
    out = {}
    for fused in ("0", "1"):
        monkeypatch.setenv("EDGL_EVAL_FUSED_TAIL", fused)
        with torch.no_grad():
            rows, _ = m.encoder(ef, False, m._gather_pos(ef, False))
            val, idx = m.eval_topk(ef, mask_seen=True)
        out[fused] = (rows.float().cpu().numpy(), idx.cpu().numpy(), val.cpu().numpy())
    assert getattr(m, "_eval_tail_ws", None) is not None          # the fused path did run
    r0, i0, v0 = out["0"]
    r1, i1, v1 = out["1"]
    assert rel_err(r1, r0) < 2.5e-2                                   # bf16 rows: a few ulps of the largest element
    assert float(np.abs(v1 - v0).max()) < 5e-2 * (1.0 + float(np.abs(v0).max()))
    overlap = np.mean([len(set(i0[r, :50]) & set(i1[r, :50])) / 50 for r in range(i0.shape[0])])
    assert overlap > 0.95, overlap


def test_chunked_eval_through_the_gemm_matches_the_scoring_kernel(monkeypatch):
    """bf16, a catalogue of several full chunks: ops.score_topk scores the full chunks behind the padding item through edgl_gemm (f32
    logits = rows . table^T + bias) and the first / ragged chunks through the scoring kernel — the same top-K as the scoring kernel
    everywhere, up to near-ties of logits that differ in their last f32 digits (another summation order)."""
    from easydgl_amd import ops
    prob = make_problem(seed=14, batch=16, num_items=3000, seqslen=20, num_units=64, num_heads=2, num_blocks=1)
    m = build_model(prob, "bf16")
    ef = to_dev(prob["efeats"])
    monkeypatch.setattr(ops, "EVAL_TILE_BYTES", 4 * 16 * 1024)       # chunks of 1024 items: [0, 1024) kernel, GEMM chunks, a ragged tail
    monkeypatch.setattr(ops, "EVAL_GEMM", False)
    v0, i0 = m.eval_topk(ef, mask_seen=True)
    monkeypatch.setattr(ops, "EVAL_GEMM", True)
    v1, i1 = m.eval_topk(ef, mask_seen=True)
    assert float((v0 - v1).abs().max()) <= 1e-5 * (1.0 + float(v0.abs().max()))
    same = (i0 == i1).float().mean()
    assert float(same) > 0.99, float(same)
    seen = prob["efeats"]["seqs_i"]
    for r in range(i1.shape[0]):
        assert not (set(i1[r].tolist()) & set(seen[r].tolist()))

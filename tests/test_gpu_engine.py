"""The static training engine (easydgl_amd/engine.py) must compute the same loss, gradients and weight
trajectory as the autograd path and the fp64 oracle; the HIP-graph replay must match the eager issue order."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from tests._util import GRAD_TOL, LOSS_TOL, build_model, grad_ok, make_problem, rel_err, to_dev

pytestmark = pytest.mark.gpu

CASES = [dict(), dict(num_units=64, num_heads=2, num_blocks=1, seqslen=30, masklen=6, num_events=7, num_items=300),
         dict(num_units=128, num_heads=8, num_blocks=1, seqslen=100, masklen=20, num_events=16, num_items=2000),
         dict(num_units=512, num_heads=8, num_blocks=1, seqslen=30, masklen=6, num_events=16, num_items=700),   # runme.sh:15-23
         # more than 16 mark types in the STATIC engine: mark groups inside the fixed launch sequence (16 + 8 at dh = 16, two
         # blocks; 16 + 16 + 8 at dh = 64) — EasyDGL.py:45-46 takes E from the data set's mark.pkl
         dict(num_units=64, num_heads=4, num_blocks=2, seqslen=20, masklen=5, num_events=24, num_items=300),
         dict(num_units=128, num_heads=2, num_blocks=1, seqslen=18, masklen=4, num_events=40, num_items=200),
         # no block at all: the head transform reads the 3C-wide encoder output (EasyDGL.py:138 sizes its dense kernel by the input)
         dict(num_units=64, num_heads=2, num_blocks=0, seqslen=30, masklen=6, num_events=7, num_items=300),
         # BASELINE.json configs[2]'s width: 256 units in 8 heads (head dim 32) — the wide strip scoring passes (k_score_stripw.hip)
         dict(num_units=256, num_heads=8, num_blocks=1, seqslen=40, masklen=8, num_events=16, num_items=1500)]


@pytest.mark.parametrize("mode", ["f32", "bf16"])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_engine_gradients_match_oracle(mode, case):
    ltol = LOSS_TOL[mode]
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=40 + case, batch=4, **CASES[case])
    cfg = prob["cfg"]
    m = build_model(prob, mode)
    eng = TrainEngine(m, 4, use_graph=False)
    eng.load_batch(to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda())
    m._grad_arena.fill_(float("nan"))          # every gradient must be (over)written by the engine
    eng._issue()
    p64 = R.to_torch_params(prob["params"])
    ref, _ = R.train_loss(cfg, p64, prob["mark_table"], prob["feats"], prob["labels"])
    ref.backward()
    assert abs(float(eng.loss) - float(ref)) <= ltol * abs(float(ref))
    bad = {}
    for name, p in m.tf_variable_map().items():
        want = p64[name].grad.numpy().copy()
        if name in ("CSTMA/item_embs/lookup_table", "CSTMA/mark_embs/lookup_table", "CSTMA/spatial_embs/embedding/lookup_table"):
            want -= cfg.l2_reg * prob["params"][name]      # the engine folds the l2 gradient into the Adam kernel
        ok, e = grad_ok(p.grad.cpu().numpy(), want, mode)
        if not ok:
            bad[name] = e
    assert not bad, (bad, GRAD_TOL[mode])


def test_engine_at_rows_that_reach_the_large_m_kernels():
    """Batch 48 at the headline shape: B*T = 4848 rows >= 4096, the threshold of the 128 x 128 tiled GEMM (QKVT projection and
    its dX) and of realistic row splits in the grouped weight-gradient launch — the small-batch cases above run the strip
    kernels instead.  bf16 against the fp64 oracle."""
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=77, batch=48, num_units=128, num_heads=8, num_blocks=1, seqslen=100, masklen=20, num_events=16,
                        num_items=2000)
    cfg = prob["cfg"]
    m = build_model(prob, "bf16")
    eng = TrainEngine(m, 48, use_graph=False)
    eng.load_batch(to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda())
    m._grad_arena.fill_(float("nan"))
    eng._issue()
    p64 = R.to_torch_params(prob["params"])
    ref, _ = R.train_loss(cfg, p64, prob["mark_table"], prob["feats"], prob["labels"])
    ref.backward()
    assert abs(float(eng.loss) - float(ref)) <= LOSS_TOL["bf16"] * abs(float(ref))
    bad = {}
    for name, p in m.tf_variable_map().items():
        want = p64[name].grad.numpy().copy()
        if name in ("CSTMA/item_embs/lookup_table", "CSTMA/mark_embs/lookup_table", "CSTMA/spatial_embs/embedding/lookup_table"):
            want -= cfg.l2_reg * prob["params"][name]
        ok, e = grad_ok(p.grad.cpu().numpy(), want, "bf16")
        if not ok:
            bad[name] = e
    assert not bad, (bad, GRAD_TOL["bf16"])


@pytest.mark.parametrize("mode", ["f32", "bf16"])
def test_engine_at_the_reference_default_width(mode):
    """main.py:35-37's defaults — num_units 50, 1 head, 3 blocks (head dim 50: run zero-padded at 64) — through the STATIC engine:
    loss and every gradient (in the reference's shapes) against the fp64 oracle at width 50, and four optimizer steps (eager and
    HIP graph, dropout on) that leave every padded entry of every parameter exactly zero."""
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=61, batch=6, num_units=50, num_heads=1, num_blocks=3, seqslen=30, masklen=6, num_events=5, num_items=200)
    cfg = prob["cfg"]
    m = build_model(prob, mode)
    assert m.pad == (64, 50)
    eng = TrainEngine(m, 6, use_graph=False)
    assert eng.fused_tail == (mode == "bf16")      # (bf16: the fused block tail in its channel-padded form, edgl_tail_fwd_ct / _bwd_ct)
    eng.load_batch(to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda())
    m._grad_arena.fill_(float("nan"))
    eng._issue()
    p64 = R.to_torch_params(prob["params"])
    ref, _ = R.train_loss(cfg, p64, prob["mark_table"], prob["feats"], prob["labels"])
    ref.backward()
    assert abs(float(eng.loss) - float(ref)) <= LOSS_TOL[mode] * abs(float(ref))
    bad = {}
    for name, g in m.tf_gradients().items():
        want = p64[name].grad.numpy().copy()
        if name in ("CSTMA/item_embs/lookup_table", "CSTMA/mark_embs/lookup_table", "CSTMA/spatial_embs/embedding/lookup_table"):
            want -= cfg.l2_reg * prob["params"][name]      # the engine folds the l2 gradient into the Adam kernel
        ok, e = grad_ok(g.cpu().numpy(), want, mode)
        if not ok:
            bad[name] = e
    assert not bad, (bad, GRAD_TOL[mode])
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    for use_graph in (False, True):
        md = build_model(prob, mode, hidden_drop=0.1, att_drop=0.1)
        e2 = TrainEngine(md, 6, use_graph=use_graph)
        losses = [float(e2.step(feats, labels)) for _ in range(4)]
        assert all(np.isfinite(losses)), losses
        assert md.padded_leak() == 0.0, use_graph


def test_engine_with_mark_groups_steps_eager_and_graph():
    """E = 24 with dropout on: the eager launch sequence and its HIP-graph capture follow the same trajectory (the group copies,
    the concatenation of lambda and the zero fill of d lambda are part of the captured sequence)."""
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=53, batch=6, num_units=64, num_heads=4, num_blocks=1, seqslen=20, masklen=5, num_events=24, num_items=300)
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    runs = []
    for use_graph in (False, True):
        m = build_model(prob, "f32", hidden_drop=0.1, att_drop=0.1)
        eng = TrainEngine(m, 6, use_graph=use_graph)
        runs.append([float(eng.step(feats, labels)) for _ in range(4)])
    assert all(np.isfinite(runs[0])) and runs[0][-1] < runs[0][0]
    for a, b in zip(*runs):
        assert abs(a - b) <= 1e-5 * abs(a), runs


def test_engine_trajectory_eager_and_graph():
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=50, batch=6)
    cfg = prob["cfg"]
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    p64 = R.to_torch_params(prob["params"])
    opt = R.TFAdam(p64, cfg.learning_rate)
    ref_losses = []
    for _ in range(4):
        ref, _ = R.train_loss(cfg, p64, prob["mark_table"], prob["feats"], prob["labels"])
        ref.backward()
        opt.step()
        ref_losses.append(float(ref))
    for use_graph in (False, True):
        m = build_model(prob, "f32")
        eng = TrainEngine(m, 6, use_graph=use_graph)
        losses = [float(eng.step(feats, labels)) for _ in range(4)]
        for a, b in zip(losses, ref_losses):
            assert abs(a - b) <= 2e-4 * abs(b), (use_graph, losses, ref_losses)
        for name, p in m.tf_variable_map().items():
            d = np.abs(p.detach().cpu().numpy() - p64[name].detach().numpy()).max()
            assert d < 3e-4, (use_graph, name, d)


@pytest.mark.parametrize("num_events", [8, 16])
def test_engine_with_dropout_matches_autograd_path_bf16(num_events):
    """Same (seed, step, op-id) -> same dropout masks in both paths -> same loss and gradients.  16 marks at head dim 16: the engine
    reads the STORED keep bits of the attention dropout (edgl_bimau_dropbits), the autograd path hashes — the same masks."""
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=51, batch=8, num_items=400, seqslen=20, num_units=64, num_heads=4, num_blocks=2, masklen=4,
                        num_events=num_events)
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    m1 = build_model(prob, "bf16", hidden_drop=0.1, att_drop=0.1)
    m2 = build_model(prob, "bf16", hidden_drop=0.1, att_drop=0.1)
    eng = TrainEngine(m2, 8, use_graph=False)
    eng.load_batch(feats, labels)
    eng._issue()
    from easydgl_amd import ops
    ops.rng_advance(m1._rng_state)
    m1.zero_grad_arena()
    loss = m1.train_loss(feats, labels)
    loss.backward()
    assert abs(float(loss) - float(eng.loss)) < 2e-3 * abs(float(loss))
    l2 = m1.l2_reg
    for (n1, p1), (n2, p2) in zip(m1.named_parameters(), m2.named_parameters()):
        g1 = p1.grad.float().cpu().numpy()
        if n1 in m1.l2_param_names():
            g1 = g1 - l2 * p1.detach().cpu().numpy()
        assert rel_err(p2.grad.float().cpu().numpy(), g1) < 3e-2, n1


PADDED_CASE = dict(num_units=50, num_heads=1, num_blocks=2, seqslen=30, masklen=6, num_events=5, num_items=200)   # head dim 50 stored as 64


@pytest.mark.parametrize("case,drop", [(1, 0.0), (2, 0.0), (2, 0.1), ("padded", 0.0), ("padded", 0.1)])
def test_fused_block_tail_matches_the_unfused_kernels(case, drop):
    """csrc/k_tail.hip (dense -> LN -> GELU-dense -> dense -> LN -> head in one launch per block) against the seven
    launches it replaces, on the same weights, batch and dropout stream: every saved tensor, the gathered head rows, the
    loss and the gradients.  bf16 only (C = 64 and the headline C = 128, T = 101); equal up to the last bf16 digit of the
    dense outputs (the two paths feed the MFMA its K slots in a different order)."""
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=67, batch=6, **PADDED_CASE) if case == "padded" else make_problem(seed=60 + case, batch=6, **CASES[case])
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    out = {}
    for fused in (False, True):
        m = build_model(prob, "bf16", hidden_drop=drop, att_drop=drop)
        eng = TrainEngine(m, 6, use_graph=False, fused_tail=fused)
        assert eng.fused_tail == fused
        eng.load_batch(feats, labels)
        eng._issue()
        b = eng.blk[-1]
        out[fused] = dict(loss=float(eng.loss), grads=m._grad_arena.clone(),
                          **{k: b[k].float().clone() for k in ("ao", "a1", "pre_f", "f", "o", "y", "st1", "st2")},
                          pre_t=eng.pre_t.float().clone(), so=eng.so.float().clone(), st3=eng.st3.clone(),
                          hrows=eng.hrows_c[:int(eng.nvalid[0])].float().clone())   # the weighted rows, compacted
    a, b = out[False], out[True]
    for k in ("pre_f", "pre_t"):   # the fused forward saves gelu'(pre-activation) in these buffers (include/easydgl_hip.h),
        x = a[k].double()          # rounded to bf16 once more
        a[k] = (0.5 * (1.0 + torch.erf(x / 2.0 ** 0.5)) + x * torch.exp(-0.5 * x * x) / (2.0 * torch.pi) ** 0.5).float()
        a[k], b[k] = a[k].bfloat16().float(), b[k].bfloat16().float()
    for k in ("ao", "a1", "pre_f", "f", "o", "y", "pre_t", "so", "hrows", "st1", "st2", "st3"):
        err = float((a[k] - b[k]).abs().max() / (a[k].abs().max() + 1e-30))
        assert err < 1.6e-2, (k, err)           # one bf16 ulp (2^-7 relative) of the largest element
        if not k.startswith("st") and k not in ("pre_f", "pre_t"):   # ... and only on a few elements (the [B, 2] statistics differ
            # in their last f32 digits; gelu' of the rounded vs the unrounded pre-activation differs by an ulp on many elements)
            assert float(((a[k] - b[k]).abs() > 1e-6 * a[k].abs().max()).float().mean()) < 0.05, k
    assert abs(a["loss"] - b["loss"]) <= 2e-3 * abs(a["loss"])
    assert rel_err(b["grads"].cpu().numpy(), a["grads"].cpu().numpy()) < 2e-2
    if case == "padded":      # the channel-padded forms: every padded channel of every saved tensor and gradient is exactly zero
        padc = torch.arange(64, device="cuda") >= 50
        for k in ("ao", "a1", "f", "o", "y", "so", "hrows"):
            t = b[k]
            cols = padc if t.shape[-1] == 64 else torch.zeros(t.shape[-1], dtype=torch.bool, device="cuda")
            assert float(t[..., cols].abs().max() if cols.any() else 0.0) == 0.0, k
        for name, (prm, maps, _) in m._pad_specs().items():      # gradients: nothing on a padded entry (the intensity MLP's output
            if name.endswith("sequential_temporal_combined/weight"):   # weights of padded hidden units are the one masked exception)
                continue
            g = prm.grad.detach().clone()
            idx = [torch.arange(g.shape[ax]) if mp is None else mp for ax, mp in enumerate(maps)]
            g[torch.meshgrid(*[i.to(g.device) for i in idx], indexing="ij")] = 0
            assert float(g.abs().max()) == 0.0, name


def test_step_with_the_loss_left_on_the_side_stream_gives_the_same_trajectory():
    """TrainEngine.sync_loss = False: Adam does not wait for the loss kernels (every deferred reduction runs on the main stream), the
    caller orders the loss with join_loss().  Same losses and the same weights after three steps as the joined form, and the
    plain _issue() of the other tests (full join at its end) is unchanged."""
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=52, batch=6, **CASES[2])
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    out = {}
    for sync in (True, False):
        m = build_model(prob, "bf16", hidden_drop=0.1, att_drop=0.1)
        eng = TrainEngine(m, 6, use_graph=False)
        eng.sync_loss = sync
        eng.load_batch(feats, labels)
        losses = []
        for _ in range(3):
            loss = eng.step()
            if not sync:
                assert eng._loss_unjoined
                eng.join_loss()
            assert not eng._loss_unjoined
            losses.append(float(loss))
        torch.cuda.synchronize()
        out[sync] = (losses, m._arena.clone())
    for a, b in zip(out[True][0], out[False][0]):
        assert abs(a - b) <= 1e-5 * abs(a)
    # (not bit for bit: the embedding scatter's f32 atomics commute only up to rounding, in either form)
    assert float((out[True][1] - out[False][1]).abs().max()) <= 1e-5 * float(out[True][1].abs().max())


@pytest.mark.parametrize("parts,defer", [("0", "1"), ("1", "0"), ("1", "1")])
def test_loss_from_the_row_finish_sums_and_launched_with_the_next_step(parts, defer, monkeypatch):
    """The loss from the per-workgroup sums of the scoring rows' finish (edgl_ce_loss_parts) against the kernel that sweeps the rows;
    with sync_loss = False its launches wait for the next step's start (third stream) or for join_loss(): the loss of step n is
    in place once step n + 1 has run, the last one after join_loss(), and the weights follow the joined form."""
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=53, batch=6, **CASES[2])
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()

    def run(sync):
        m = build_model(prob, "bf16", hidden_drop=0.1, att_drop=0.1)
        eng = TrainEngine(m, 6, use_graph=False)
        eng.sync_loss = sync
        eng.load_batch(feats, labels)
        return m, eng

    monkeypatch.setenv("EDGL_CE_PARTS", "0")
    m0, e0 = run(True)
    assert e0.ce_part is None
    want = []
    for _ in range(4):
        want.append(float(e0.step()))
    torch.cuda.synchronize()
    monkeypatch.setenv("EDGL_CE_PARTS", parts)
    monkeypatch.setenv("EDGL_DEFER_LOSS", defer)
    m1, e1 = run(False)
    assert (e1.ce_part is not None) == (parts == "1")
    deferred = parts == "1" and defer == "1"
    got = []
    for n in range(4):
        e1.step()
        assert (e1._deferred_loss is not None) == deferred
        torch.cuda.synchronize()
        if deferred and n > 0:
            got.append(float(e1.loss))      # the previous step's loss: launched at the start of this one
    e1.join_loss()
    assert e1._deferred_loss is None and not e1._loss_unjoined
    got.append(float(e1.loss))
    if not deferred:
        want = want[-1:]
    assert len(got) == len(want)
    for a, b in zip(want, got):
        assert abs(a - b) <= 1e-5 * abs(a), (want, got)
    assert float((m0._arena - m1._arena).abs().max()) <= 1e-5 * float(m0._arena.abs().max())


@pytest.mark.parametrize("path", ["flash", "no_blocks", "two_pass"])
def test_accumulated_loss_covers_every_loss_path(path, monkeypatch):
    """train.py runs the engine with sync_loss = False + accumulate_loss = True and reads engine.loss_sum at its logging points (the
    NaN guard of util.py:29-30 hangs on it).  The running sum must be fed on all three loss paths: the side-stream launches of the
    flash form, the main-stream launch of a model without blocks (num_blocks 0) and the inline loss of the two-pass form
    (EDGL_FLASH_CE=0).  loss_sum == the sum of the step losses the joined form returns."""
    from easydgl_amd.engine import TrainEngine
    if path == "two_pass":
        monkeypatch.setenv("EDGL_FLASH_CE", "0")
    kw = dict(CASES[1])
    if path == "no_blocks":
        kw["num_blocks"] = 0
    prob = make_problem(seed=61, batch=6, **kw)
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    m0 = build_model(prob, "bf16")
    e0 = TrainEngine(m0, 6, use_graph=False)
    e0.load_batch(feats, labels)
    want = sum(float(e0.step()) for _ in range(4))
    m1 = build_model(prob, "bf16")
    e1 = TrainEngine(m1, 6, use_graph=False)
    e1.sync_loss, e1.accumulate_loss = False, True
    e1.load_batch(feats, labels)
    for _ in range(4):
        e1.step()
    e1.join_loss()
    torch.cuda.synchronize()
    got = float(e1.loss_sum)
    assert want > 0.0 and abs(got - want) <= 1e-5 * want, (path, got, want)


def test_l2_term_from_the_optimizer_sums_only_while_this_engine_owns_the_weights():
    """The L2 term of a step comes from the sums of squares the engine's previous optimizer launch left (edgl_adam_apply_l2p) — valid
    only while nobody else rewrote the arena.  Weights reloaded between two steps (load_tf_variables), or another engine of the
    same model stepping in between, take the ownership token away and the term is recomputed from the arena."""
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=62, batch=6, **CASES[1])
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    p2 = {k: (np.asarray(v) * 3.0 if "lookup_table" in k or "embedding" in k.lower() else np.asarray(v)) for k, v in prob["params"].items()}
    assert any(not np.array_equal(np.asarray(prob["params"][k]), p2[k]) for k in p2)
    # reference: a fresh model with the second weight set, first step of its engine (L2 from the arena)
    mref = build_model(prob, "bf16")
    mref.load_tf_variables(p2)
    eref = TrainEngine(mref, 6, use_graph=False)
    eref.load_batch(feats, labels)
    want = float(eref.step())
    # (a) reload between steps
    m = build_model(prob, "bf16")
    e = TrainEngine(m, 6, use_graph=False)
    assert e.l2_parts is not None
    e.load_batch(feats, labels)
    e.step(); e.step()
    assert m._l2_parts_owner == id(e)
    m.load_tf_variables(p2)
    assert m._l2_parts_owner is None
    got = float(e.step())
    assert abs(got - want) <= 1e-5 * abs(want), (got, want)
    # (b) two engines on one model alternate: each step's loss equals the single-engine trajectory
    ma, mb = build_model(prob, "bf16"), build_model(prob, "bf16")
    ea = TrainEngine(ma, 6, use_graph=False); ea.load_batch(feats, labels)
    single = [float(ea.step()) for _ in range(4)]
    e1, e2 = TrainEngine(mb, 6, use_graph=False), TrainEngine(mb, 6, use_graph=False)
    e1.load_batch(feats, labels); e2.load_batch(feats, labels)
    both = [float((e1 if n % 2 == 0 else e2).step()) for n in range(4)]
    for a, b in zip(single, both):
        assert abs(a - b) <= 1e-5 * abs(a), (single, both)


def test_optimizer_launch_that_sums_the_slabs_and_writes_the_next_counters(monkeypatch):
    """edgl_adam_apply_ex (the eager step's optimizer launch): the scoring gradient's row-chunk slabs summed inside the optimizer (no
    slab_reduce launch) and the next step's counters written to a second pair of buffers that the host swaps in (no step_begin
    launch) — against the round-5 launch sequence (EDGL_ADAM_EX=0): the same losses and weights over five steps WITH dropout (the
    dropout step counter and Adam's step / learning rate advance identically), counters that settle to "steps taken", a bare
    _issue() that still leaves complete gradients, two engines alternating on one model, and a checkpoint round trip in between."""
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=63, batch=6, **CASES[2])
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()

    def run(flag, two_engines=False):
        monkeypatch.setenv("EDGL_ADAM_EX", flag)
        m = build_model(prob, "bf16", hidden_drop=0.1, att_drop=0.1)
        engs = [TrainEngine(m, 6, use_graph=False) for _ in range(2 if two_engines else 1)]
        assert all(e.adam_ex == (flag == "1") for e in engs)
        for e in engs:
            e.load_batch(feats, labels)
        losses = [float(engs[n % len(engs)].step()) for n in range(5)]
        torch.cuda.synchronize()
        return m, engs, losses

    m0, _, l0 = run("0")
    m1, e1, l1 = run("1")
    m2, _, l2 = run("1", two_engines=True)
    for a, b, c in zip(l0, l1, l2):
        assert abs(a - b) <= 2e-5 * abs(a) and abs(a - c) <= 2e-5 * abs(a), (l0, l1, l2)
    scale = float(m0._arena.abs().max())
    assert float((m0._arena - m1._arena).abs().max()) <= 2e-5 * scale and float((m0._arena - m2._arena).abs().max()) <= 2e-5 * scale
    # the counters: one step ahead while the engine owns them, "steps taken" once settled — on whichever pair of buffers is current
    for m in (m0, m1, m2):
        assert m._state_ahead and int(m._adam_state[0]) == 6 and int(m._rng_state[1]) == 6
        m.settle_state()
        assert int(m._adam_state[0]) == 5 and int(m._rng_state[1]) == 5
    assert torch.equal(m0._rng_state, m1._rng_state) and torch.equal(m0._adam_state, m1._adam_state)
    # a bare _issue() (the parity tests' entry point) leaves COMPLETE gradients in the arena under either setting
    grads = {}
    for flag in ("0", "1"):
        monkeypatch.setenv("EDGL_ADAM_EX", flag)
        m = build_model(prob, "bf16")
        e = TrainEngine(m, 6, use_graph=False)
        e.load_batch(feats, labels)
        e.step()                        # (a step with the slabs left to the optimizer in front of it)
        e._issue()
        torch.cuda.synchronize()
        grads[flag] = m._grad_arena.clone()
    assert float((grads["0"] - grads["1"]).abs().max()) <= 2e-5 * float(grads["0"].abs().max())

"""Parity at the sizes BASELINE.json quotes, where the kernels change behaviour (item-chunk planner, several row blocks,
vector-width specialisations): scoring at I = 20 001 and I = 200 003, K1 encode at C in {128, 256} against a 1 000 001-row
table, the whole model and the static engine at num_items = 20 000, and the config-3 workload (|items| = 1M, L = 200,
d = 256) end to end.  Same oracle and tolerances as the small cases (tests/test_gpu_ops.py, tests/test_gpu_model.py)."""
import math

import numpy as np
import pytest
import torch

from oracle import easydgl_oracle as O
from oracle import torch_ref as R
from tests._util import LOSS_TOL, assert_close, build_model, grad_ok, make_problem, rel_err, to_dev

pytestmark = pytest.mark.gpu


def _ops():
    from easydgl_amd import ops
    return ops


@pytest.mark.parametrize("name,dt", [("f32", torch.float32), ("bf16", torch.bfloat16)])
@pytest.mark.parametrize("R_,C,I", [(700, 128, 20001), (700, 128, 200003), (1300, 64, 20001)])
def test_score_ce_at_catalogue_sizes(name, dt, R_, C, I):
    """Fused scoring / cross-entropy (EasyDGL.py:149-155,177-185) with several 256-row blocks and many item chunks.  The
    fp64 reference is the plain formula evaluated with torch on the device (the [R, I] logits tensor the kernels avoid)."""
    o = _ops()
    g = torch.Generator(device="cuda").manual_seed(R_ + I)
    rows = (torch.randn((R_, C), generator=g, device="cuda") * 0.5).to(dt).requires_grad_()
    tab = (torch.randn((I, C), generator=g, device="cuda") * 0.3).requires_grad_()
    tab_c = tab.detach().to(dt)
    bias = (torch.randn((I - 1,), generator=g, device="cuda") * 0.2).requires_grad_()
    rng = np.random.default_rng(R_)
    labels = rng.integers(0, I, size=R_)
    labels[rng.random(R_) < 0.15] = 0
    labels[:5] = [1, I - 1, I - 2, 0, 129]           # first / last table rows, a chunk boundary
    lab = torch.tensor(labels, dtype=torch.int64).cuda()
    loss = o.ScoreCEFn.apply(rows, tab, bias, tab_c, lab)
    (loss * 1.3).backward()
    rr = rows.detach().double().requires_grad_()
    tr = tab_c.double().requires_grad_()
    br = bias.detach().double().requires_grad_()
    logits = rr @ R.zero_padded(tr).t() + torch.cat([torch.full((1,), -1000.0, dtype=torch.float64, device="cuda"), br])
    lp = torch.log(torch.softmax(logits, -1) + 1e-5)
    wgt = (lab != 0).double()
    ref = (wgt * -lp[torch.arange(R_, device="cuda"), lab]).sum() / (wgt.sum() + 1e-5)
    (ref * 1.3).backward()
    assert_close(loss.item(), ref.item(), 2e-5 if name == "f32" else 5e-3, "ce loss")
    gt = 1e-4 if name == "f32" else 3e-2
    assert_close(rows.grad.float().cpu().numpy(), rr.grad.cpu().numpy(), gt, "d_rows")
    assert_close(tab.grad.cpu().numpy(), tr.grad.cpu().numpy(), gt, "d_table")
    assert_close(bias.grad.cpu().numpy(), br.grad.cpu().numpy(), gt, "d_bias")
    assert float(tab.grad[0].abs().max()) == 0.0
    lse, lab_logit, _ = o.score_lse(rows.detach(), tab_c, bias.detach(), lab, 0, I)
    assert_close(lse.cpu().numpy(), torch.logsumexp(logits, -1).detach().cpu().numpy(), 1e-5 if name == "f32" else 2e-3, "lse")


@pytest.mark.parametrize("name,dt,C,B", [("f32", torch.float32, 128, 4), ("bf16", torch.bfloat16, 128, 4), ("f32", torch.float32, 256, 4),
                                         ("bf16", torch.bfloat16, 256, 4), ("bf16", torch.bfloat16, 256, 512)])
def test_encode_against_a_million_row_table(name, dt, C, B):
    """K1 (EasyDGL.py:70-95) at config-3 sizes: I = 1 000 001 rows, T = 201, C in {128, 256} — the last case at the FULL batch of 512
    (102 912 tokens, ~ 60 K distinct table rows).  The oracle only ever reads the gathered rows, so it runs on the batch's own rows of
    the table (ids renumbered); the kernel reads the full table."""
    o = _ops()
    num_items, T, E = 1_000_000, 201, 16
    I = num_items + 1
    rng = np.random.default_rng(C)
    cfg_big = O.Config(num_items=num_items, seqslen=T - 1, num_units=C, num_heads=8, time_scale=86400.0, num_events=E)
    ids, ts = O.synthetic_sequences(cfg_big, B, rng, min_len=150)
    ids[0, -3:] = [num_items - 1, 1, num_items]                  # last item row, first item row, the MASK token
    ids[1, 100:110] = rng.integers(900_000, num_items, size=10)   # far rows of the table
    g = torch.Generator(device="cuda").manual_seed(C)
    item = (torch.randn((I, C), generator=g, device="cuda") * 0.05).requires_grad_()
    pos = (torch.randn((T, C), generator=g, device="cuda") * 0.05).requires_grad_()
    mk = (torch.randn((E, C), generator=g, device="cuda") * 0.05).requires_grad_()
    item_c = item.detach().to(dt)
    mt = O.synthetic_mark_table(num_items, E, multi_hot=True)
    tscale = torch.tensor(O.time_sinusoid_scale(C)).cuda()
    x0, spans, marks = o.EncodeFn.apply(item, pos, mk, item_c, torch.tensor(ids).cuda(), torch.tensor(ts).cuda(),
                                        torch.tensor(mt.astype(np.uint8)).cuda(), tscale, cfg_big.mask_id, cfg_big.time_scale,
                                        o.NO_DROP, dt)
    # renumbered problem for the oracle: table rows [0] + the batch's distinct ids (the MASK id keeps the last slot)
    uniq = np.unique(ids[(ids != 0) & (ids != num_items)])
    remap = np.zeros(I, dtype=np.int64)
    remap[uniq] = np.arange(1, len(uniq) + 1)
    small_n = len(uniq) + 1
    remap[num_items] = small_n                                   # MASK token = num_items of the small problem
    cfg = O.Config(num_items=small_n, seqslen=T - 1, num_units=C, num_heads=8, time_scale=86400.0, num_events=E)
    rows_idx = torch.tensor(np.concatenate([[0], uniq, [num_items]]), device="cuda")
    pp = {"CSTMA/item_embs/lookup_table": item_c[rows_idx].double().cpu().numpy(),
          "CSTMA/spatial_embs/embedding/lookup_table": pos.detach().double().cpu().numpy(),
          "CSTMA/mark_embs/lookup_table": mk.detach().double().cpu().numpy()}
    mt_small = np.concatenate([mt[:1], mt[uniq]], axis=0)
    want_x0, want_sp, want_mk, _ = O.input_encode(cfg, pp, mt_small, remap[ids], ts)
    np.testing.assert_array_equal(marks.cpu().numpy(), want_mk)
    np.testing.assert_array_equal(spans.cpu().numpy(), want_sp.astype(np.float32))
    assert_close(x0.detach().float().cpu().numpy(), want_x0, 2e-6 if name == "f32" else 8e-3, "x0")
    G = (torch.randn(tuple(x0.shape), generator=g, device="cuda")).to(dt)
    x0.backward(G)
    Gd = G.double().cpu().numpy()
    d_small = np.zeros((small_n + 1, C))
    np.add.at(d_small, remap[ids].reshape(-1), math.sqrt(C) * Gd[..., :C].reshape(-1, C))
    d_small[0] = 0
    got = item.grad[rows_idx].cpu().numpy()
    assert_close(got, d_small, 1e-5, "d_item (touched rows)")
    tot = float(item.grad.double().abs().sum())
    assert abs(tot - float(np.abs(got.astype(np.float64)).sum())) <= 1e-9 * tot   # no other row of the table was touched
    assert_close(pos.grad.cpu().numpy(), Gd[..., C:2 * C].sum(0), 1e-5, "d_pos")
    d_mk = np.zeros((E, C))
    d_mk[1] = (want_mk.sum(-1)[..., None] * Gd[..., 2 * C:]).sum((0, 1))
    assert_close(mk.grad.cpu().numpy(), d_mk, 1e-5, "d_mark")


HEADLINE_ITEMS = dict(num_units=128, num_heads=8, num_blocks=1, seqslen=100, masklen=20, num_events=16, num_items=20000)


@pytest.mark.parametrize("mode,ltol", [("f32", 1e-4), ("bf16", 2e-2)])
def test_model_and_engine_at_the_headline_catalogue(mode, ltol):
    """EasyDGL at the BASELINE.json shape (T = 101, C = 128, h = 8, M = 20, E = 16, I = 20 001; batch 4 keeps the fp64
    [80, 20001] logits small): logits, loss, every gradient through the autograd path, then the static engine."""
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=77, batch=4, **HEADLINE_ITEMS)
    cfg = prob["cfg"]
    m = build_model(prob, mode)
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    logits = m(feats, True)
    want_logits, _ = O.forward(cfg, prob["params"], prob["mark_table"], prob["feats"], True)
    assert logits.shape == want_logits.shape == (80, 20001)
    assert_close(logits.detach().cpu().numpy(), want_logits, ltol, "train logits")
    m.zero_grad_arena()
    loss = m.train_loss(feats, labels)
    loss.backward()
    p64 = R.to_torch_params(prob["params"])
    ref_loss, _ = R.train_loss(cfg, p64, prob["mark_table"], prob["feats"], prob["labels"])
    ref_loss.backward()
    assert_close(loss.item(), ref_loss.item(), LOSS_TOL[mode], "train loss")
    # every gradient tensor: relative L2 AND max-norm (tests/_util.py GRAD_TOL)
    bad = {}
    for n, p in m.tf_variable_map().items():
        ok, e = grad_ok(p.grad.cpu().numpy(), p64[n].grad.numpy(), mode)
        if not ok:
            bad[n] = e
    assert not bad, f"autograd path: {bad}"
    eng = TrainEngine(m, 4, use_graph=False)
    eng.load_batch(feats, labels)
    m._grad_arena.fill_(float("nan"))
    eng._issue()
    assert abs(float(eng.loss) - float(ref_loss)) <= LOSS_TOL[mode] * abs(float(ref_loss))
    bad = {}
    for name, p in m.tf_variable_map().items():
        want = p64[name].grad.numpy().copy()
        if name in O.EMBEDDING_TABLES:
            want -= cfg.l2_reg * prob["params"][name]      # the engine folds the l2 gradient into the Adam kernel
        ok, e = grad_ok(p.grad.cpu().numpy(), want, mode)
        if not ok:
            bad[name] = e
    assert not bad, f"engine path: {bad}"
    # evaluation at the same catalogue: top-100 vs the oracle's ranking
    ef = to_dev(prob["efeats"])
    _, idx = m.eval_topk(ef, mask_seen=True)
    _, want_idx = O.evaluate(cfg, prob["params"], prob["mark_table"], prob["efeats"], prob["elabels"])
    got = idx.cpu().numpy()
    if mode == "f32":
        assert (got == want_idx).mean() > 0.98
    else:
        assert np.mean([len(set(got[r, :50]) & set(want_idx[r, :50])) / 50 for r in range(got.shape[0])]) > 0.9


def test_config3_workload_end_to_end():
    """BASELINE.json config 3 (|items| = 1M, L = 200 -> T = 201, d = 256, 8 heads, M = 40, E = 16) in bf16, batch 2: loss and
    every gradient of one training step against the fp64 restatement (which materialises the [80, 1 000 001] logits), the
    static engine on the same batch, and the evaluation step (mask_topk over a million logits per row)."""
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=3, batch=2, num_units=256, num_heads=8, num_blocks=1, seqslen=200, masklen=40, num_events=16,
                        num_items=1_000_000, perturb=False)
    cfg = prob["cfg"]
    m = build_model(prob, "bf16")
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda()
    m.zero_grad_arena()
    loss = m.train_loss(feats, labels)
    loss.backward()
    p64 = R.to_torch_params(prob["params"])
    ref_loss, _ = R.train_loss(cfg, p64, prob["mark_table"], prob["feats"], prob["labels"])
    ref_loss.backward()
    assert_close(loss.item(), ref_loss.item(), LOSS_TOL["bf16"], "train loss")
    bad = {}
    for n, p in m.tf_variable_map().items():
        ok, e = grad_ok(p.grad.cpu().numpy(), p64[n].grad.numpy(), "bf16")
        if not ok:
            bad[n] = e
    assert not bad, bad
    eng = TrainEngine(m, 2, use_graph=False)
    l_eng = float(eng.step(feats, labels))
    assert abs(l_eng - float(ref_loss)) <= LOSS_TOL["bf16"] * abs(float(ref_loss))
    ef = to_dev(prob["efeats"])
    val, idx = m.eval_topk(ef, mask_seen=True)
    assert idx.shape == (2, 100) and int(idx.min()) >= 1 and int(idx.max()) <= 1_000_000
    seen = prob["efeats"]["seqs_i"]
    for r in range(2):
        assert not (set(idx[r].cpu().numpy().tolist()) & set(seen[r].tolist()))    # Base.py:156-163
    assert bool((val[:, :-1] >= val[:, 1:]).all())                                  # tf.nn.top_k order
    # the sharded scoring step at this catalogue agrees with the unsharded one
    v8, i8 = m.eval_topk_sharded(ef, mask_seen=True, world=8)
    assert torch.equal(i8, idx) and torch.equal(v8, val)

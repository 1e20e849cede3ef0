"""the HIP CTSMA (easydgl_amd/model/ctsma.py, through the reference's model interface) vs the fp64
restatement oracle/ctsma_ref.py on the same seeded inputs and weights.  Tolerances as for EasyDGL: f32 path 1e-4 on
logits / loss and 1e-3 on gradients; bf16 path 3e-2 and, per gradient tensor, max-norm 1e-1 AND relative L2 8e-2 (BF16_GRAD_L2)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import ctsma_ref as CR
from oracle import easydgl_oracle as O
from tests._util import assert_close, grad_errors, regressive_bf16_bounds, rel_err, relu_flip_err, to_dev

# per-tensor relative L2 bound of the bf16 path beside the max-norm bound `gtol`.  These models gate their feed-forward with a
# ReLU: a pre-activation within bf16 rounding of 0 flips its mask against the fp64 reference, and the flipped unit's whole
# contribution then travels to every gradient upstream of it (measured: up to 0.058 in the first block of the two-block
# cases, 0.02-0.03 elsewhere; the GELU-gated EasyDGL path holds 2e-2: tests/_util.py GRAD_TOL)
BF16_GRAD_L2 = 8e-2

pytestmark = pytest.mark.gpu

CASES = [
    dict(B=3, T=12, C=32, h=2, E=4, I=60, nb=2),
    dict(B=32, T=30, C=64, h=2, E=7, I=300, nb=1),         # dh = 32
    dict(B=4, T=100, C=128, h=8, E=16, I=2000, nb=2),      # headline widths
    dict(B=4, T=30, C=512, h=4, E=16, I=700, nb=2),        # the published recipe runme.sh:107-115 (dh = 128, 2 blocks, seqslen 30)
    dict(B=3, T=14, C=64, h=2, E=24, I=90, nb=1),          # more than 16 mark types: two mark groups, MAU keeps the diagonal
    dict(B=2, T=100, C=256, h=2, E=8, I=400, nb=1),        # CTSMA's head dim 128 at L = 100 (bf16: sweep 2 in two channel slices)
    dict(B=32, T=30, C=50, h=1, E=6, I=300, nb=3),          # the reference's DEFAULT flags (main.py:35-44): head dim 50, zero-padded to 64
    dict(B=24, T=20, C=100, h=2, E=16, I=120, nb=1),       # two padded heads (50 -> 64 each)
    dict(B=3, T=16, C=44, h=2, E=20, I=90, nb=1),          # head dim 22 -> 32 with two mark groups
]


def _problem(seed, B, T, C, h, E, I, nb, time_scale=3600.0):
    rng = np.random.default_rng(seed)
    params = {}
    for k, v in CR.init_params(I, T, C, h, E, nb, rng).items():
        if v.ndim == 1:   # biases / LayerNorm / scaling start at constants in the reference: perturb so they matter
            v = v + 0.05 * rng.standard_normal(v.shape)
        params[k] = v.astype(np.float32).astype(np.float64)
    tokens = rng.integers(1, I, size=(B, T + 1))
    tokens[0, :T // 3] = 0                                   # left padding
    tokens[1, :1] = 0
    ts = np.cumsum(rng.exponential(1800.0, size=(B, T + 1)), axis=1).astype(np.float32)
    mt = O.synthetic_mark_table(I, E, multi_hot=True)
    feats = {"seqs_i": tokens[:, :-1].copy(), "seqs_t": ts}
    return dict(params=params, mt=mt, feats=feats, tokens=tokens,
                kw=dict(C=C, h=h, num_blocks=nb, time_scale=time_scale), dims=dict(B=B, T=T, C=C, h=h, E=E, I=I, nb=nb))


def _model(prob, mode, ct_reg=1e-2, l2_reg=1e-3, hidden_drop=0.0, att_drop=0.0, lr=1e-3):
    import easydgl_amd
    d = prob["dims"]
    F = SimpleNamespace(model="CTSMA", num_items=d["I"], num_units=d["C"], num_heads=d["h"], num_blocks=d["nb"],
                        seqslen=d["T"], time_scale=prob["kw"]["time_scale"], learning_rate=lr, l2_reg=l2_reg, ct_reg=ct_reg,
                        hidden_dropout_rate=hidden_drop, attention_probs_dropout_rate=att_drop, mark_table=prob["mt"],
                        compute_dtype=mode, num_train_steps=None, num_warmup_steps=None)
    m = easydgl_amd.ranking(F).finalize("cuda")
    m.load_tf_variables(prob["params"])
    return m


def _p64(prob):
    return {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in prob["params"].items()}


@pytest.mark.parametrize("mode,ltol,gtol", [("f32", 1e-4, 1e-3), ("bf16", 3e-2, 1e-1)])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_forward_loss_and_gradients(mode, ltol, gtol, case):
    if mode == "f32" and CASES[case]["C"] // CASES[case]["h"] == 128 and CASES[case]["T"] > 64:
        pytest.skip("head dim 128 beyond T = 64 is a bf16 shape (f32 staging of K / T_ / V exceeds the LDS)")
    prob = _problem(40 + case, **CASES[case])
    m = _model(prob, mode)
    feats = to_dev(prob["feats"])
    labels_np = prob["tokens"][:, 1:].copy()
    labels = torch.as_tensor(labels_np).cuda()
    p64 = _p64(prob)
    # ---- CTSMA.__call__(features, True): logits of every position [B*T, I]
    logits = m(feats, True)
    ref_loss, aux = CR.train_loss(p64, prob["mt"], prob["feats"], labels_np, ct_reg=1e-2, l2_reg=1e-3, **prob["kw"])
    assert logits.shape == aux["logits"].shape
    assert_close(logits.detach().float().cpu().numpy(), aux["logits"].detach().numpy(), ltol, "train logits")
    assert float((logits.detach()[:, 0] + 1000).abs().max()) == 0.0
    for a, b in zip(m._last_lams, aux["lams"]):
        assert_close(a.detach().float().cpu().numpy(), b.detach().numpy(), ltol, "lambda")
    # ---- CTSMA.train loss and gradients
    m.zero_grad_arena()
    loss = m.train_loss(feats, labels)
    loss.backward()
    ref_loss.backward()
    assert_close(loss.item(), ref_loss.item(), ltol, "train loss")
    got = m.tf_gradients()
    assert set(got) == set(p64)
    bad, errs = {}, {}
    for name, g in got.items():
        ref = p64[name].grad.numpy()
        # bf16: per tensor class, <= 2 x the measured errors (tests/_util.py regressive_bf16_bounds); f32: gtol on everything
        l2_tol, g_tol = regressive_bf16_bounds("ctsma", name, gtol, BF16_GRAD_L2) if mode == "bf16" else (BF16_GRAD_L2, gtol)
        if name.endswith("modulating_attention/dense_1/bias"):
            # a bias on K shifts every score of a query row by the same amount: its true gradient is identically zero
            # (the oracle holds ~1e-17 there), so measure against the scale of the K kernel's gradient instead
            ref_k = p64[name.replace("bias", "kernel")].grad.numpy()
            assert np.abs(ref).max() < 1e-12 * np.abs(ref_k).max()
            e = float(np.abs(g).max() / np.abs(ref_k).max())
        else:
            e = rel_err(g, ref)
            # bf16: the max-norm bound alone lets every small entry of a tensor be wrong — a relative-L2 bound beside it
            # (the ReLU-gated Inner tensors are held by relu_flip_err's own rms bound instead)
            if mode == "bf16" and "/Inner/" not in name and grad_errors(g, ref)[0] > l2_tol:
                bad[name + " (rel-L2)"] = grad_errors(g, ref)[0]
        # bf16 only: a pre-activation within bf16 rounding of 0 flips its ReLU mask, which moves one whole term of the
        # row sums behind Inner/kernel and Inner/bias (measured: error ~ 1/sqrt(rows), 0.13 at 120 rows, 0.05 at 3840);
        # every other gradient is continuous in the activations.  f32 keeps the plain tolerance.
        if mode == "bf16" and "/Inner/" in name:
            e = relu_flip_err(g, ref, gtol)       # tests/_util.py: flipped hidden units are counted, the rest is held to gtol
        errs[f"{mode}:{name}"] = (grad_errors(g, ref)[0] if np.any(ref) else 0.0, e)
        if e > g_tol:
            bad[name] = (e, g_tol)
    from tests._util import dump_errors
    dump_errors("ctsma", errs)
    assert not bad, f"gradient mismatch (rel to max |ref|): {bad}"
    # ---- evaluation: logits of the last position
    elog = m(feats, False)
    want = CR.eval_logits(_p64(prob), prob["mt"], prob["feats"], **prob["kw"])
    assert_close(elog.detach().float().cpu().numpy(), want.detach().numpy(), ltol, "eval logits")


def test_eval_topk_masks_seen_items_and_ranks_like_the_oracle():
    prob = _problem(7, B=16, T=20, C=32, h=2, E=4, I=600, nb=1)
    m = _model(prob, "f32")
    feats = to_dev(prob["feats"])
    _, idx = m.eval_topk(feats, mask_seen=True)
    got = idx.cpu().numpy()
    lg = CR.eval_logits(_p64(prob), prob["mt"], prob["feats"], **prob["kw"]).detach().numpy().copy()
    seen = prob["feats"]["seqs_i"]
    for r in range(lg.shape[0]):
        lg[r, seen[r]] = -np.inf                                   # Base.py:156-163
        lg[r, 0] = -np.inf
        assert not (set(got[r]) & set(seen[r]))
    want = np.argsort(-lg, axis=1, kind="stable")[:, :100]
    assert (got == want).mean() > 0.98
    m.reset_metrics()
    m.eval_step(feats, torch.as_tensor(prob["tokens"]).cuda())
    per = O.ranking_metrics(got, prob["tokens"][:, -1])
    for k, v in m.metrics().items():
        assert abs(v - per[k].mean()) < 1e-5


def test_three_adam_steps_follow_the_oracle_trajectory():
    from oracle import torch_ref as R
    prob = _problem(11, B=5, T=12, C=32, h=2, E=4, I=60, nb=2)
    m = _model(prob, "f32")
    p64 = _p64(prob)
    opt = R.TFAdam(p64, 1e-3)
    feats = to_dev(prob["feats"])
    labels_np = prob["tokens"][:, 1:].copy()
    labels = torch.as_tensor(labels_np).cuda()
    for step in range(3):
        got = float(m.train_step(feats, labels))
        ref, _ = CR.train_loss(p64, prob["mt"], prob["feats"], labels_np, ct_reg=1e-2, l2_reg=1e-3, **prob["kw"])
        ref.backward()
        opt.step()
        assert abs(got - float(ref.detach())) <= 2e-4 * abs(float(ref.detach())), (step, got)


def test_training_with_dropout_reduces_the_loss_bf16():
    prob = _problem(5, B=32, T=20, C=64, h=4, E=8, I=400, nb=2)
    m = _model(prob, "bf16", hidden_drop=0.1, att_drop=0.1, lr=2e-3)
    feats = to_dev(prob["feats"])
    labels = torch.as_tensor(prob["tokens"][:, 1:].copy()).cuda()
    losses = [float(m.train_step(feats, labels)) for _ in range(30)]
    assert all(np.isfinite(losses))
    assert losses[-1] < losses[0] - 0.3, losses


@pytest.mark.parametrize("name", ["CTSMA", "TGAT", "TiSASREC"])
def test_graphed_train_step_follows_the_eager_trajectory(name):
    """Sequential.graphed_train_step: the autograd step captured into one HIP graph (2 eager warm-up steps, then replays) gives the
    losses of eager train_step calls — dropout counter and Adam step count advance on the device inside the graph."""
    import easydgl_amd
    rng = np.random.default_rng(3)
    B, T, C, I = 6, 14, 32, 50
    tokens = rng.integers(1, I, size=(B, T + 1))
    tokens[0, :4] = 0
    ts = (9.5e8 + np.cumsum(rng.exponential(30000.0, size=(B, T + 1)), axis=1)).astype(np.float32)
    ts[tokens == 0] = 0.0

    def make():
        F = SimpleNamespace(model=name, num_items=I, num_units=C, num_heads=2, num_blocks=2, seqslen=T, timelen=16, time_scale=86400.0,
                            learning_rate=1e-3, l2_reg=1e-3, ct_reg=1e-4, hidden_dropout_rate=0.1, attention_probs_dropout_rate=0.1,
                            mark_table=O.synthetic_mark_table(I, 4, multi_hot=True), compute_dtype="f32", num_train_steps=None,
                            num_warmup_steps=None)
        return easydgl_amd.ranking(F).finalize("cuda")

    feats = to_dev({"seqs_i": tokens[:, :-1].copy(), "seqs_t": ts})
    labels = torch.as_tensor(tokens[:, 1:].copy()).cuda()
    eager, graphed = make(), make()
    want = [float(eager.train_step(feats, labels)) for _ in range(6)]
    step = graphed.graphed_train_step(feats, labels, warmup=2)          # two real steps happen here
    got = [float(step(feats, labels)) for _ in range(4)]
    for a, b in zip(got, want[2:]):
        assert abs(a - b) <= 1e-4 * abs(b), (got, want)
    # a different batch through the same graph
    f2 = {k: v.clone() for k, v in feats.items()}
    f2["seqs_i"][1, 5:8] = 7
    l_g = float(step(f2, labels))
    l_e = float(eager.train_step(f2, labels))
    assert abs(l_g - l_e) <= 1e-4 * abs(l_e)

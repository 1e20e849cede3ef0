"""SURVEY §8 row a-14 (config 5): the generic masked attention kernel (edgl_tattn_*), the time feature map (edgl_timefn_*)
and the TGAT model class, through the C ABI, against torch float64 references / oracle/baselines_ref.py.
Tolerances: f32 path 1e-4 (forward) and 1e-3 (gradients); bf16 path 3e-2 and, per gradient tensor, max-norm 1e-1 AND relative L2 8e-2 (BF16_GRAD_L2)."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import baselines_ref as BR
from oracle import easydgl_oracle as O
from tests._util import assert_close, grad_errors, regressive_bf16_bounds, rel_err, relu_flip_err, to_dev

# per-tensor relative L2 bound of the bf16 path beside the max-norm bound `gtol`.  These models gate their feed-forward with a
# ReLU: a pre-activation within bf16 rounding of 0 flips its mask against the fp64 reference, and the flipped unit's whole
# contribution then travels to every gradient upstream of it (measured: up to 0.058 in the first block of the two-block
# cases, 0.02-0.03 elsewhere; the GELU-gated EasyDGL path holds 2e-2: tests/_util.py GRAD_TOL)
BF16_GRAD_L2 = 8e-2

pytestmark = pytest.mark.gpu
PAD = float(np.float32(-2 ** 32 + 1))


def _attn_ref(qx, kx, v, resid, ids, H, scale, causal=True):
    """float64 restatement of temporal.py:147-181 on already projected operands (head slices along the channels)."""
    B, T, _ = qx.shape
    Dq, Dv = qx.shape[2] // H, v.shape[2] // H
    out = torch.zeros(B, T, H * Dv, dtype=torch.float64)
    tril = torch.tril(torch.ones(T, T, dtype=torch.bool))
    for hd in range(H):
        S = qx[:, :, hd * Dq:(hd + 1) * Dq] @ kx[:, :, hd * Dq:(hd + 1) * Dq].transpose(1, 2) * scale
        S = torch.where((ids == 0)[:, None, :].expand(B, T, T), torch.full_like(S, PAD), S)
        if causal:
            S = torch.where(~tril[None], torch.full_like(S, PAD), S)
        out[:, :, hd * Dv:(hd + 1) * Dv] = torch.softmax(S, -1) @ v[:, :, hd * Dv:(hd + 1) * Dv]
    return out + resid


def _tattn(qx, kx, v, resid, ids, H, scale, rate=0.0, rng=None, saved=True, flags=1):
    from easydgl_amd import _lib as L
    from easydgl_amd import ops
    B, T, _ = qx.shape
    Dq, Dv = qx.shape[2] // H, v.shape[2] // H
    out = torch.empty(B, T, H * Dv, device="cuda", dtype=qx.dtype)
    sv = torch.empty(int(L.lib.edgl_tattn_saved_bytes(B, T, H, Dv)), device="cuda", dtype=torch.uint8) if saved else None
    st = torch.cuda.current_stream().cuda_stream
    L.check(L.lib.edgl_tattn_fwd(qx.data_ptr(), qx.shape[2], kx.data_ptr(), kx.shape[2], v.data_ptr(), v.shape[2], resid.data_ptr(),
                                 resid.shape[2], ids.data_ptr(), B, T, H, Dq, Dv, scale, rate, rng.data_ptr() if rng is not None else None,
                                 7, out.data_ptr(), out.shape[2], sv.data_ptr() if saved else None, flags, ops._code(qx), st), "fwd")
    return out, sv


def _tattn_bwd(qx, kx, v, ids, d_out, sv, H, scale, rate=0.0, rng=None, flags=1):
    from easydgl_amd import _lib as L
    from easydgl_amd import ops
    B, T, _ = qx.shape
    Dq, Dv = qx.shape[2] // H, v.shape[2] // H
    dq, dk, dv = torch.empty_like(qx), torch.empty_like(kx), torch.empty_like(v)
    st = torch.cuda.current_stream().cuda_stream
    L.check(L.lib.edgl_tattn_bwd(qx.data_ptr(), qx.shape[2], kx.data_ptr(), kx.shape[2], v.data_ptr(), v.shape[2], ids.data_ptr(),
                                 d_out.data_ptr(), d_out.shape[2], sv.data_ptr(), B, T, H, Dq, Dv, scale, rate,
                                 rng.data_ptr() if rng is not None else None, 7, dq.data_ptr(), dq.shape[2], dk.data_ptr(),
                                 dk.shape[2], dv.data_ptr(), dv.shape[2], flags, ops._code(qx), st), "bwd")
    return dq, dk, dv


def _operands(seed, B, T, H, Dq, Dv, dtype):
    rng = np.random.default_rng(seed)
    mk = lambda *s: torch.tensor(rng.standard_normal(s), dtype=torch.float32).to(dtype)
    qx, kx, v, resid, d_out = mk(B, T, H * Dq), mk(B, T, H * Dq), mk(B, T, H * Dv), mk(B, T, H * Dv), mk(B, T, H * Dv)
    ids = torch.tensor(rng.integers(1, 50, size=(B, T)))
    ids[0, :max(1, T // 3)] = 0          # left padding: fully masked query rows
    if T > 5:
        ids[-1, T - 3] = 0               # a padded key away from the left edge
    if B >= 3:
        ids[2, :] = 0                    # an all-padding sequence: every row uniform over all T keys
    return qx, kx, v, resid, d_out, ids


@pytest.mark.parametrize("dtype,ftol,gtol", [(torch.float32, 1e-4, 1e-3), (torch.bfloat16, 3e-2, 1e-1)])
@pytest.mark.parametrize("B,T,H,Dq,Dv", [(3, 12, 2, 16, 16), (2, 37, 2, 48, 16), (2, 100, 1, 384, 128), (2, 50, 8, 48, 16),
                                         (2, 33, 2, 96, 32), (1, 201, 2, 64, 64),
                                         (1, 1, 1, 16, 16), (2, 16, 2, 32, 32), (2, 17, 1, 192, 64),   # edges: one key, tile-exact, tile + 1
                                         # wide heads (sliced kernels): TGAT's published head (runme.sh:80-87: Dq = 3*512, Dv = 512),
                                         # two 256-wide heads, a plain (Dq == Dv) wide head
                                         (3, 30, 1, 1536, 512), (2, 45, 2, 768, 256), (2, 20, 1, 256, 256)])
def test_tattn_forward_and_backward_match_float64(dtype, ftol, gtol, B, T, H, Dq, Dv):
    qx, kx, v, resid, d_out, ids = _operands(B * T + Dq, B, T, H, Dq, Dv, dtype)
    scale = 1.0 / np.sqrt(Dv)
    ref_in = [t.double().requires_grad_(True) for t in (qx, kx, v)]
    want = _attn_ref(ref_in[0], ref_in[1], ref_in[2], resid.double(), ids, H, scale)
    want.backward(d_out.double())
    dev = [t.cuda() for t in (qx, kx, v, resid, d_out, ids)]
    out, sv = _tattn(dev[0], dev[1], dev[2], dev[3], dev[5], H, scale)
    assert_close(out.float().cpu().numpy(), want.detach().numpy(), ftol, "attention output")
    dq, dk, dv = _tattn_bwd(dev[0], dev[1], dev[2], dev[5], dev[4], sv, H, scale)
    for name, got, ref in (("d_qx", dq, ref_in[0]), ("d_kx", dk, ref_in[1]), ("d_v", dv, ref_in[2])):
        assert_close(got.float().cpu().numpy(), ref.grad.numpy(), gtol, name)
    # inference call (no saved buffer) gives the same output
    out2, _ = _tattn(dev[0], dev[1], dev[2], dev[3], dev[5], H, scale, saved=False)
    assert torch.equal(out, out2)


def test_tattn_dropout_mask_is_the_same_in_forward_and_both_backward_kernels():
    """With dropout the output is linear in V for a fixed mask (d_v check) and d_qx / d_kx match central differences taken
    with the same counter-based mask."""
    B, T, H, Dq, Dv = 2, 40, 2, 48, 16
    qx, kx, v, resid, d_out, ids = [t.cuda() for t in _operands(5, B, T, H, Dq, Dv, torch.float32)]
    rng = torch.tensor([1234, 5], dtype=torch.int64, device="cuda")
    scale, rate = 0.25, 0.3
    out, sv = _tattn(qx, kx, v, resid, ids, H, scale, rate, rng)
    out_nodrop, _ = _tattn(qx, kx, v, resid, ids, H, scale)
    assert not torch.allclose(out, out_nodrop)
    dq, dk, dv = _tattn_bwd(qx, kx, v, ids, d_out, sv, H, scale, rate, rng)
    g = torch.Generator(device="cuda").manual_seed(0)
    dV = torch.randn(v.shape, device="cuda", generator=g)
    out_v, _ = _tattn(qx, kx, v + dV, resid, ids, H, scale, rate, rng)
    lhs, rhs = float(((out_v - out) * d_out).sum()), float((dv * dV).sum())
    assert abs(lhs - rhs) <= 1e-3 * max(1.0, abs(rhs)), (lhs, rhs)
    for which, grad in (("q", dq), ("k", dk)):
        d = torch.randn(qx.shape, device="cuda", generator=g)
        eps = 1e-2
        args_p = (qx + eps * d, kx) if which == "q" else (qx, kx + eps * d)
        args_m = (qx - eps * d, kx) if which == "q" else (qx, kx - eps * d)
        op, _ = _tattn(args_p[0], args_p[1], v, resid, ids, H, scale, rate, rng)
        om, _ = _tattn(args_m[0], args_m[1], v, resid, ids, H, scale, rate, rng)
        fd = float(((op.double() - om.double()) * d_out.double()).sum()) / (2 * eps)
        an = float((grad.double() * d.double()).sum())
        assert abs(fd - an) <= 2e-2 * max(1.0, abs(an)), (which, fd, an)
    keep = 1.0 - rate    # the mask keeps about (1 - rate) of the probabilities: compare the dropped mass through V = ones
    ones = torch.ones_like(v)
    o1, _ = _tattn(qx, kx, ones, torch.zeros_like(resid), ids, H, scale, rate, rng)
    assert abs(float(o1.mean()) - 1.0) < 0.05 and float(o1.std()) > 0.01 and keep < 1


# ---- model level --------------------------------------------------------------------------------------------------------
CASES = [
    dict(B=3, T=12, C=32, h=2, I=60, nb=2),
    dict(B=8, T=30, C=64, h=2, I=300, nb=1),
    dict(B=4, T=100, C=128, h=1, I=2000, nb=3),       # runme.sh:80-87 heads / blocks at d = 128 (h=1, dh=128, 3 blocks)
    dict(B=4, T=30, C=512, h=1, I=700, nb=3),         # the published recipe runme.sh:80-87 itself: ONE head of 512 channels
    dict(B=3, T=40, C=512, h=2, I=300, nb=1),         # dh = 256: two slices per head
    # the reference's DEFAULT width (main.py:35-37): head dim 50, zero-padded to 64.  Two blocks here: at three blocks of this narrow
    # ReLU-gated width the bf16 path's first-block gradients sit at 0.11-0.15 of the reference's maximum (mask flips of three
    # feed-forwards travelling upstream; the f32 path holds 1e-3 at three blocks: the next case; the driver test trains them)
    dict(B=32, T=30, C=50, h=1, I=300, nb=2),
    dict(B=8, T=30, C=50, h=1, I=300, nb=3, f32_only=True),      # the default flags exactly (3 blocks), f32 path
    dict(B=24, T=20, C=100, h=2, I=120, nb=1),        # two padded heads (50 -> 64 each), C = 100 stored as 128
]


def _problem(seed, B, T, C, h, I, nb, time_scale=3600.0 * 24):
    rng = np.random.default_rng(seed)
    params = {}
    for k, v in BR.tgat_init_params(I, T, C, nb, rng).items():
        if v.ndim == 1 and "basis_freq" not in k:
            v = v + 0.05 * rng.standard_normal(v.shape)
        params[k] = v.astype(np.float32).astype(np.float64)
    tokens = rng.integers(1, I, size=(B, T + 1))
    tokens[0, :T // 3] = 0
    tokens[1, :1] = 0
    ts = (9.5e8 + np.cumsum(rng.exponential(0.2 * time_scale, size=(B, T + 1)), axis=1)).astype(np.float32)
    ts[tokens == 0] = 0.0                                     # left padding carries timestamp 0 (linkpred.py:152-153)
    feats = {"seqs_i": tokens[:, :-1].copy(), "seqs_t": ts}
    return dict(params=params, feats=feats, tokens=tokens, kw=dict(C=C, h=h, nb=nb, time_scale=time_scale),
                dims=dict(B=B, T=T, C=C, h=h, I=I, nb=nb))


def _model(prob, mode, l2_reg=1e-3, hidden_drop=0.0, att_drop=0.0, lr=1e-3):
    import easydgl_amd
    d = prob["dims"]
    F = SimpleNamespace(model="TGAT", num_items=d["I"], num_units=d["C"], num_heads=d["h"], num_blocks=d["nb"], seqslen=d["T"],
                        time_scale=prob["kw"]["time_scale"], learning_rate=lr, l2_reg=l2_reg, hidden_dropout_rate=hidden_drop,
                        attention_probs_dropout_rate=att_drop, compute_dtype=mode, num_train_steps=None, num_warmup_steps=None)
    m = easydgl_amd.ranking(F).finalize("cuda")
    m.load_tf_variables(prob["params"])
    return m


def _p64(prob):
    return {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in prob["params"].items()}


@pytest.mark.parametrize("mode,ltol,gtol", [("f32", 1e-4, 1e-3), ("bf16", 3e-2, 1e-1)])
@pytest.mark.parametrize("case", range(len(CASES)))
def test_tgat_forward_loss_and_gradients(mode, ltol, gtol, case):
    cfg = dict(CASES[case])
    if cfg.pop("f32_only", False) and mode != "f32":
        pytest.skip("f32-only case")
    prob = _problem(60 + case, **cfg)
    m = _model(prob, mode)
    feats = to_dev(prob["feats"])
    labels_np = prob["tokens"][:, 1:].copy()
    labels = torch.as_tensor(labels_np).cuda()
    p64 = _p64(prob)
    logits = m(feats, True)
    ref_loss, aux = BR.tgat_train_loss(p64, prob["feats"], labels_np, l2_reg=1e-3, **prob["kw"])
    assert logits.shape == aux["logits"].shape
    assert_close(logits.detach().float().cpu().numpy(), aux["logits"].detach().numpy(), ltol, "train logits")
    assert float((logits.detach()[:, 0] + 1000).abs().max()) == 0.0
    m.zero_grad_arena()
    loss = m.train_loss(feats, labels)
    loss.backward()
    ref_loss.backward()
    m.check_inputs()
    assert_close(loss.item(), ref_loss.item(), ltol, "train loss")
    got = m.tf_gradients()
    assert set(got) == set(p64)
    bad, errs = {}, {}
    for name, g in got.items():
        ref = p64[name].grad.numpy()
        # bf16: per tensor class, <= 2 x the measured errors (tests/_util.py regressive_bf16_bounds); f32: gtol on everything
        l2_tol, g_tol = regressive_bf16_bounds("tgat", name, gtol, BF16_GRAD_L2) if mode == "bf16" else (BF16_GRAD_L2, gtol)
        if name.endswith("timeinterval/dense_1/bias"):   # a bias on K shifts a whole score row: its true gradient is zero
            ref_k = p64[name.replace("bias", "kernel")].grad.numpy()
            assert np.abs(ref).max() < 1e-12 * max(np.abs(ref_k).max(), 1e-30)
            e = float(np.abs(g).max() / np.abs(ref_k).max())
        else:
            e = rel_err(g, ref)
            # bf16: the max-norm bound alone lets every small entry of a tensor be wrong — a relative-L2 bound beside it
            # (the ReLU-gated Inner tensors are held by relu_flip_err's own rms bound instead)
            if mode == "bf16" and "/Inner/" not in name and grad_errors(g, ref)[0] > l2_tol:
                bad[name + " (rel-L2)"] = grad_errors(g, ref)[0]
        if mode == "bf16" and "/Inner/" in name:   # ReLU mask flips, see tests/_util.py:relu_flip_err
            e = relu_flip_err(g, ref, gtol)
        errs[f"{mode}:{name}"] = (grad_errors(g, ref)[0] if np.any(ref) else 0.0, e)
        if e > g_tol:
            bad[name] = (e, g_tol)
    from tests._util import dump_errors
    dump_errors("tgat", errs)
    assert not bad, f"gradient mismatch (rel to max |ref|): {bad}"
    elog = m(feats, False)
    want = BR.tgat_eval_logits(_p64(prob), prob["feats"], **prob["kw"])
    assert_close(elog.detach().float().cpu().numpy(), want.detach().numpy(), ltol, "eval logits")


def test_tgat_flags_decreasing_timestamps():
    prob = _problem(3, B=3, T=12, C=32, h=2, I=60, nb=1)
    m = _model(prob, "f32")
    feats = to_dev(prob["feats"])
    m(feats, False)
    m.check_inputs()                                            # sorted: fine
    bad = {k: v.clone() for k, v in feats.items()}
    bad["seqs_t"][2, 8] = bad["seqs_t"][2, 6]                   # ts[8] < ts[7]
    m(bad, False)
    with pytest.raises(ValueError):
        m.check_inputs()


def test_tgat_adam_steps_follow_the_oracle_and_training_with_dropout_learns():
    from oracle import torch_ref as R
    prob = _problem(11, B=5, T=12, C=32, h=2, I=60, nb=2)
    m = _model(prob, "f32")
    p64 = _p64(prob)
    opt = R.TFAdam(p64, 1e-3)
    feats = to_dev(prob["feats"])
    labels_np = prob["tokens"][:, 1:].copy()
    labels = torch.as_tensor(labels_np).cuda()
    for step in range(3):
        got = float(m.train_step(feats, labels))
        ref, _ = BR.tgat_train_loss(p64, prob["feats"], labels_np, l2_reg=1e-3, **prob["kw"])
        ref.backward()
        opt.step()
        assert abs(got - float(ref.detach())) <= 2e-4 * abs(float(ref.detach())), (step, got)
    prob = _problem(5, B=32, T=20, C=64, h=4, I=400, nb=2)
    m = _model(prob, "bf16", hidden_drop=0.1, att_drop=0.1, lr=2e-3)
    feats = to_dev(prob["feats"])
    labels = torch.as_tensor(prob["tokens"][:, 1:].copy()).cuda()
    losses = [float(m.train_step(feats, labels)) for _ in range(30)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0] - 0.3, losses
    _, idx = m.eval_topk(feats, mask_seen=True)
    seen = prob["feats"]["seqs_i"]
    got = idx.cpu().numpy()
    assert all(not (set(got[r]) & set(seen[r])) for r in range(got.shape[0]))
    m.reset_metrics()
    m.eval_step(feats, torch.as_tensor(prob["tokens"]).cuda())
    per = O.ranking_metrics(got, prob["tokens"][:, -1])
    for k, v in m.metrics().items():
        assert abs(v - per[k].mean()) < 1e-5

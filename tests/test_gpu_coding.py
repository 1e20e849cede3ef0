"""Operator-level API of SURVEY §8(b): the classes of src/module/coding.py and T.BiMAU called on their own, with the
reference's constructor signatures and methods, against the oracle's functions (coding.py:45-149, temporal.py:401-452)."""
import numpy as np
import pytest
import torch

from oracle import easydgl_oracle as O
from oracle import torch_ref as R
from tests._util import assert_close, rel_err

pytestmark = pytest.mark.gpu


def _gen(seed=0):
    return torch.Generator().manual_seed(seed)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-6), (torch.bfloat16, 8e-3)])
@pytest.mark.parametrize("zero_pad,scale", [(True, True), (False, False), (True, False)])
def test_embedding_call_and_gradient(dt, tol, zero_pad, scale):
    """C.Embedding(vocab, units, l2, zero_pad, scale)(ids) — coding.py:45-64."""
    from easydgl_amd.module import coding as C
    V, U = 37, 64
    emb = C.Embedding(V, U, 0.0, zero_pad=zero_pad, scale=scale, gen=_gen(1)).cuda()
    if dt != torch.float32:
        shadow = emb.lookup_table.detach().to(dt)
        emb.compute = lambda p: shadow
    rng = np.random.default_rng(0)
    ids = rng.integers(0, V, size=(5, 9))
    ids[0, :3] = 0
    out = emb(torch.tensor(ids).cuda())
    tab = emb.compute(emb.lookup_table).detach().double().cpu().numpy()
    want = (O.zero_padded(tab) if zero_pad else tab)[ids] * (U ** 0.5 if scale else 1.0)
    assert out.shape == (5, 9, U) and out.dtype == dt
    assert_close(out.detach().float().cpu().numpy(), want, tol, "embedding")
    G = torch.randn(out.shape, generator=_gen(2)).to(dt).cuda()
    out.backward(G)
    d = np.zeros((V, U))
    np.add.at(d, ids.reshape(-1), G.double().cpu().numpy().reshape(-1, U) * (U ** 0.5 if scale else 1.0))
    if zero_pad:
        d[0] = 0
    assert_close(emb.lookup_table.grad.cpu().numpy(), d, 1e-5, "d_table")
    # an index past the table reads zeros (the reference's GPU lookup; TiSASREC.py:59 relies on it)
    far = emb(torch.tensor([[V, V + 3]]).cuda())
    assert float(far.detach().float().abs().max()) == 0.0


def test_position_coding_code_and_call():
    """C.PositionCoding(vocab, units).code(x) / (x) — coding.py:67-79."""
    from easydgl_amd.module import coding as C
    pc = C.PositionCoding(21, 32, 0.0, gen=_gen(3)).cuda()
    x = torch.randn((4, 13, 32), generator=_gen(4)).cuda()
    code = pc.code(x)
    tab = pc.pembs.lookup_table.detach().cpu().numpy()
    want = np.broadcast_to(tab[:13][None], (4, 13, 32))
    np.testing.assert_array_equal(code.detach().cpu().numpy(), want)
    both = pc(x)
    assert both.shape == (4, 13, 64)
    np.testing.assert_array_equal(both[..., :32].detach().cpu().numpy(), x.cpu().numpy())
    np.testing.assert_array_equal(both[..., 32:].detach().cpu().numpy(), want)
    code.sum().backward()
    g = pc.pembs.lookup_table.grad.cpu().numpy()
    assert np.allclose(g[:13], 4.0) and np.allclose(g[13:], 0.0)


@pytest.mark.parametrize("dt,tol", [(torch.float32, 5e-7), (torch.bfloat16, 5e-3)])
def test_time_sinusoid_code(dt, tol):
    """C.TimeSinusoidCoding(units).code(x[B,T]) — coding.py:125-149, at Netflix-scale arguments and at 0."""
    from easydgl_amd.module import coding as C
    tc = C.TimeSinusoidCoding(128).cuda()
    tc.act_dtype = dt
    cfg = O.Config(num_items=10, seqslen=40, num_units=128, num_heads=8, time_scale=86400.0, num_events=2)
    _, ts = O.synthetic_sequences(cfg, 3, np.random.default_rng(2), min_len=30)
    x32 = O.scaled_times(ts, cfg.time_scale)
    got = tc.code(torch.tensor(x32).cuda())
    want = O.time_sinusoid_code(x32, 128)
    assert got.shape == (3, 41, 128)
    assert np.abs(got.float().cpu().numpy() - want).max() < tol
    z = tc.code(torch.zeros((1, 2)).cuda()).float().cpu().numpy()
    np.testing.assert_array_equal(z[0, 0], np.tile([0.0, 1.0], 64))      # coding.py:144-148 at 0
    with pytest.raises(AssertionError):
        tc.code(torch.zeros((1, 2, 3)).cuda())                            # coding.py:139


# f32: the argument x*f + phi reaches ~110 rad, so its float32 rounding alone moves the cosine by up to ~7e-6
@pytest.mark.parametrize("dt,tol", [(torch.float32, 1e-5), (torch.bfloat16, 8e-3)])
def test_time_function_code_and_gradients(dt, tol):
    """C.TimeFunctionCoding(units).code(x) — coding.py:97-122 — on [B,T] and [B,T,T] inputs."""
    from easydgl_amd.module import coding as C
    tf_ = C.TimeFunctionCoding(64).cuda()
    tf_.act_dtype = dt
    with torch.no_grad():
        tf_.phase.copy_(torch.randn(64, generator=_gen(5)) * 0.3)
    rng = np.random.default_rng(1)
    for shape in [(3, 7), (2, 6, 6)]:
        tf_.basis_freq.grad = tf_.phase.grad = None
        x = rng.uniform(0, 12, size=shape).astype(np.float32)
        got = tf_.code(torch.tensor(x).cuda())
        f64 = tf_.basis_freq.detach().double().cpu().requires_grad_()
        p64 = tf_.phase.detach().double().cpu().requires_grad_()
        xr = torch.tensor(x, dtype=torch.float64).reshape(shape[0], shape[1], -1)
        want = torch.cos(xr.unsqueeze(-1) * f64 + p64)                     # coding.py:113-121
        assert got.shape == want.shape
        assert_close(got.detach().float().cpu().numpy(), want.detach().numpy(), tol, "time function code")
        G = torch.randn(want.shape, generator=_gen(6)).to(dt)
        got.backward(G.cuda())
        (want * G.double()).sum().backward()
        assert_close(tf_.basis_freq.grad.cpu().numpy(), f64.grad.numpy(), 1e-4 if dt == torch.float32 else 2e-2, "d_freq")
        assert_close(tf_.phase.grad.cpu().numpy(), p64.grad.numpy(), 1e-4 if dt == torch.float32 else 2e-2, "d_phase")


def test_time_interval_code():
    """C.TimeIntervalCoding(vocab, units).code(int intervals) — coding.py:82-94."""
    from easydgl_amd.module import coding as C
    ti = C.TimeIntervalCoding(16, 32, 0.0, gen=_gen(7)).cuda()
    iv = torch.tensor(np.random.default_rng(3).integers(0, 17, size=(2, 5, 5))).cuda()   # 16 = one past the table
    got = ti.code(iv)
    tab = np.concatenate([ti.pembs.lookup_table.detach().cpu().numpy(), np.zeros((1, 32), np.float32)])
    np.testing.assert_array_equal(got.detach().cpu().numpy(), tab[iv.cpu().numpy()])


@pytest.mark.parametrize("dt,ftol,gtol", [(torch.float32, 3e-5, 2e-4), (torch.bfloat16, 3e-2, 6e-2)])
def test_bimau_operator_with_the_reference_signature(dt, ftol, gtol):
    """T.BiMAU(num_units, num_heads, num_events, dropout_rate)(queries, keys, masks, intervals, marks, is_training) ->
    (outputs [B,T,C], mark_intensity [hB,T,E]) — temporal.py:401-452; the QKVT projection is created on the first call."""
    from easydgl_amd.module import temporal as T
    B, Tn, C, H, E = 3, 19, 64, 4, 5
    att = T.BiMAU(C, H, E, 0.1).cuda()
    assert att.dense_kernel is None
    rng = np.random.default_rng(8)
    x = torch.tensor(rng.standard_normal((B, Tn, 3 * C)), dtype=dt).cuda().requires_grad_()
    ids = rng.integers(1, 30, size=(B, Tn)); ids[0, :5] = 0
    mt = O.synthetic_mark_table(30, E, multi_hot=True)
    marks = mt[ids]
    spans = rng.uniform(0, 5, size=(B, Tn))
    if dt != torch.float32:
        att(x.detach(), x.detach(), torch.tensor(ids).cuda(), torch.tensor(spans, dtype=torch.float32).cuda(),
            torch.tensor(marks.astype(np.uint8)).cuda(), False)                      # creates the projection
        shadow = {id(p): p.detach().to(dt) for p in att.parameters()}
        att.compute = lambda p: shadow[id(p)]
    out, lam = att(x, x, torch.tensor(ids).cuda(), torch.tensor(spans, dtype=torch.float32).cuda(),
                   torch.tensor(marks.astype(np.uint8)).cuda(), False)
    assert tuple(att.dense_kernel.shape) == (3 * C, 4 * C) and abs(float(att.dense_kernel.detach().std()) - 0.02) < 0.002
    assert out.shape == (B, Tn, C) and lam.shape == (H * B, Tn, E)
    (out.float().sum() + lam.sum()).backward()
    xr = x.detach().double().cpu().requires_grad_()
    pr = {"dense/kernel": att.compute(att.dense_kernel).detach().double().cpu(), "dense/bias": att.dense_bias.detach().double().cpu(),
          "sequential_temporal_combined/dense/kernel": att.st_kernel.detach().double().cpu(),
          "sequential_temporal_combined/dense/bias": att.st_bias.detach().double().cpu(),
          "sequential_temporal_combined/weight": att.weight.detach().double().cpu(),
          "sequential_temporal_combined/scaling": att.scaling.detach().double().cpu()}
    km3 = torch.tensor((ids != 0).astype(np.float64)).unsqueeze(1).repeat(H, Tn, 1)
    out_r, lam_r = R.bimau(C, H, xr, km3, torch.tensor(spans), torch.tensor(marks, dtype=torch.float64), pr, "", 0.0, False)
    (out_r.sum() + lam_r.sum()).backward()
    assert_close(out.detach().float().cpu().numpy(), out_r.detach().numpy(), ftol, "outputs")
    assert_close(lam.detach().cpu().numpy(), lam_r.detach().numpy(), ftol, "mark_intensity")
    assert_close(x.grad.float().cpu().numpy(), xr.grad.numpy(), gtol, "d_queries")


def test_unsupported_head_dim_fails_in_the_constructor():
    from easydgl_amd.module import temporal as T
    with pytest.raises(ValueError, match="head dim"):
        T.BiMAU(50, 1, 4, 0.0)
    with pytest.raises(ValueError, match="num_events"):
        T.MAU(64, 2, 300, 0.0)


def test_bimau_takes_the_reference_key_mask_and_trains_with_dropout_on_its_own():
    """(queries, keys, masks, ...) with the reference's own `masks` — tile(expand_dims(float(ids != 0), 1), [h, T, 1]), EasyDGL.py:94-95
    — gives the outputs of the [B, T] id form; is_training with dropout_rate > 0 and no model around draws from a module-local state."""
    from easydgl_amd import ops
    from easydgl_amd._lib import EdglError
    from easydgl_amd.module import temporal as T
    B, Tn, C, H, E = 3, 19, 64, 4, 5
    att = T.BiMAU(C, H, E, 0.3).cuda()
    rng = np.random.default_rng(9)
    x = torch.tensor(rng.standard_normal((B, Tn, 3 * C)), dtype=torch.float32).cuda()
    ids = rng.integers(1, 30, size=(B, Tn)); ids[1, :7] = 0
    ids_t = torch.tensor(ids).cuda()
    spans = torch.tensor(rng.uniform(0, 5, size=(B, Tn)), dtype=torch.float32).cuda()
    marks = torch.tensor(O.synthetic_mark_table(30, E, multi_hot=True)[ids].astype(np.uint8)).cuda()
    ref_mask = (ids_t != 0).float().unsqueeze(1).repeat(H, Tn, 1)                      # [h*B, T, T]
    o0, l0 = att(x, x, ids_t, spans, marks, False)
    o1, l1 = att(x, x, ref_mask, spans, marks, False)
    assert torch.equal(o0, o1) and torch.equal(l0, l1)
    with pytest.raises(ValueError):
        att(x, x, ref_mask[:, :, :-1], spans, marks, False)
    with pytest.raises(EdglError):                                                     # the raw op refuses what it would mis-read
        ops.BiMAUFn.apply(torch.zeros(B, Tn, 4 * C, device="cuda"), x[:, :, :C], att.st_kernel, att.st_bias, att.weight, att.scaling,
                          ref_mask, spans, marks, H, ops.NO_DROP)
    t1, _ = att(x, x, ids_t, spans, marks, True)
    t2, _ = att(x, x, ids_t, spans, marks, True)
    assert not torch.equal(t1, o0) and not torch.equal(t1, t2)                          # dropout is on, and a fresh mask per call


@pytest.mark.parametrize("unit", ["BiMAU", "MAU"])
def test_unit_called_twice_before_backward_keeps_each_calls_mask(unit):
    """A unit on its own draws from a module-local dropout state that advances per call.  Two calls under ONE loss (shared layer,
    gradient accumulation) must each differentiate under the mask of their OWN forward: the summed gradient equals the sum of the
    gradients of the two calls run one after the other with the same two states."""
    from easydgl_amd.module import temporal as T
    B, Tn, C, H, E = 2, 21, 64, 4, 6
    gen = torch.Generator().manual_seed(11)
    att = (T.BiMAU(C, H, E, 0.3, in_units=C, gen=gen) if unit == "BiMAU" else T.MAU(C, H, E, 0.3, gen=gen)).cuda()
    rng = np.random.default_rng(3)
    xs = [torch.tensor(rng.standard_normal((B, Tn, C)), dtype=torch.float32).cuda() for _ in range(2)]
    ids = rng.integers(1, 30, size=(B, Tn)); ids[0, :4] = 0
    ids_t = torch.tensor(ids).cuda()
    spans = torch.tensor(rng.uniform(0, 5, size=(B, Tn)), dtype=torch.float32).cuda()
    marks = torch.tensor(O.synthetic_mark_table(30, E, multi_hot=True)[ids].astype(np.uint8)).cuda()
    ws = [torch.tensor(rng.standard_normal((B, Tn, C)), dtype=torch.float32).cuda() for _ in range(2)]
    params = [p for p in att.parameters()]

    def run(joint):
        att._own_rng = None                       # both runs start from the same module-local state
        for p in params:
            p.grad = None
        outs = []
        if joint:                                 # two forwards, then one backward
            for x in xs:
                outs.append(att(x, x, ids_t, spans, marks, True)[0])
            (sum((o * w).sum() for o, w in zip(outs, ws))).backward()
        else:                                     # forward / backward, forward / backward (gradients accumulate)
            for x, w in zip(xs, ws):
                o = att(x, x, ids_t, spans, marks, True)[0]
                outs.append(o)
                (o * w).sum().backward()
        return [o.detach().clone() for o in outs], [p.grad.detach().clone() for p in params]

    o_j, g_j = run(True)
    o_s, g_s = run(False)
    for a, b in zip(o_j, o_s):
        assert torch.equal(a, b)                  # same masks in the forwards
    assert not torch.equal(o_j[0], att(xs[0], xs[0], ids_t, spans, marks, False)[0])
    for a, b in zip(g_j, g_s):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) < 1e-5


@pytest.mark.parametrize("unit", ["BiMAU", "MAU"])
def test_mark_groups_share_one_dropout_mask(unit):
    """A unit with 32 mark types whose first 16 mark columns are empty computes what the 16-mark unit built from the second half
    of its intensity weights computes — with attention dropout ON: the two launches of modulated_attention must drop the same
    (query, key) pairs, the first one carries the residual and (BiMAU) the diagonal 1, the second the modulation."""
    from easydgl_amd import ops
    from easydgl_amd.module import temporal as T
    B, Tn, C, H, E = 3, 21, 64, 4, 32
    dh = C // H
    gen = torch.Generator().manual_seed(4)
    mk = (lambda e: T.BiMAU(C, H, e, 0.25, in_units=C, gen=gen)) if unit == "BiMAU" else (lambda e: T.MAU(C, H, e, 0.25, gen=gen))
    big, small = mk(E).cuda(), mk(16).cuda()
    with torch.no_grad():
        for n, p in small.named_parameters():
            q = dict(big.named_parameters())[n]
            if n == "st_kernel": p.copy_(q[:, 16 * dh:])
            elif n == "st_bias": p.copy_(q[16 * dh:])
            elif n in ("weight", "scaling"): p.copy_(q[16:])
            else: p.copy_(q)
        big.scaling.add_(0.1 * torch.randn(E, generator=gen).cuda()); small.scaling.copy_(big.scaling[16:])
    rng = np.random.default_rng(12)
    x = torch.tensor(rng.standard_normal((B, Tn, C)), dtype=torch.float32).cuda()
    ids = rng.integers(1, 40, size=(B, Tn)); ids[2, :6] = 0
    ids_t = torch.tensor(ids).cuda()
    spans = torch.tensor(rng.uniform(0, 5, size=(B, Tn)), dtype=torch.float32).cuda()
    m16 = O.synthetic_mark_table(40, 16, multi_hot=True)[ids].astype(np.uint8)
    marks16 = torch.tensor(m16).cuda()
    marks32 = torch.cat([torch.zeros_like(marks16), marks16], dim=-1).contiguous()
    rng_state = ops.make_rng_state("cuda", seed=77)
    drop = ops.Drop(0.25, rng_state, 3)
    xb, xs = x.clone().requires_grad_(), x.clone().requires_grad_()
    ob, lb = big(xb, xb, ids_t, spans, marks32, True, drop=drop)
    os_, ls = small(xs, xs, ids_t, spans, marks16, True, drop=drop)
    assert lb.shape == (H * B, Tn, E)
    assert float((ob - os_).abs().max()) < 2e-5 * (1 + float(os_.abs().max()))
    assert float((lb[:, :, 16:] - ls).abs().max()) < 1e-5
    w = torch.tensor(rng.standard_normal((B, Tn, C)), dtype=torch.float32).cuda()
    (ob * w).sum().backward(); (os_ * w).sum().backward()
    assert float((xb.grad - xs.grad).abs().max()) < 5e-5 * (1 + float(xs.grad.abs().max()))
    gb, gs = big.st_kernel.grad[:, 16 * dh:], small.st_kernel.grad
    assert float((gb - gs).abs().max()) < 5e-5 * (1 + float(gs.abs().max()))
    assert not torch.equal(ob, big(xb, xb, ids_t, spans, marks32, False)[0])       # dropout was on


def _bimau_problem(B, Tn, C, H, E, rate, seed):
    from easydgl_amd.module import temporal as T
    att = T.BiMAU(C, H, E, rate, in_units=C, gen=_gen(seed)).cuda()
    with torch.no_grad():
        att.scaling.add_(0.2 * torch.randn(E, generator=_gen(seed + 1)).cuda())
        att.dense_kernel.mul_(8.0)                               # scores that spread the softmax
    rng = np.random.default_rng(seed)
    x = torch.tensor(rng.standard_normal((B, Tn, C)), dtype=torch.float32).cuda()
    ids = rng.integers(1, 40, size=(B, Tn)); ids[0, :5] = 0
    spans = torch.tensor(rng.uniform(0, 5, size=(B, Tn)), dtype=torch.float32).cuda()
    marks = torch.tensor(O.synthetic_mark_table(40, E, multi_hot=True)[ids].astype(np.uint8)).cuda()
    return att, x, torch.tensor(ids).cuda(), spans, marks


def test_attention_dropout_is_unbiased():
    """tf.layers.dropout (temporal.py:442) keeps an element with probability 1 - rate and scales it by 1 / (1 - rate): the mean of the
    outputs over many masks approaches the output without dropout like 1 / sqrt(N).  The keep decisions come from 16-bit fields of one
    64-bit hash per four neighbouring (query, key) pairs (drop_hash_quad) and the scale rides on lambda — a rate or scale that is off by
    2 % leaves a bias this test sees."""
    from easydgl_amd import ops
    B, Tn, C, H, E, rate, N = 4, 53, 64, 4, 16, 0.1, 1600
    att, x, ids, spans, marks = _bimau_problem(B, Tn, C, H, E, rate, 21)
    o0, _ = att(x, x, ids, spans, marks, False)
    a0 = o0 - x                                                  # the attention term (the residual is not dropped)
    state = ops.make_rng_state("cuda", seed=5)
    acc = torch.zeros_like(o0)
    first = None
    with torch.no_grad():
        for i in range(N):
            t, _ = att(x, x, ids, spans, marks, True, drop=ops.Drop(rate, state, 3))
            ops.rng_advance(state)
            acc += t
            if first is None:
                first = t.clone()
        e1 = float((first - o0).norm()) / float(a0.norm())
        en = float((acc / N - o0).norm()) / float(a0.norm())
    assert e1 > 0.02                                             # dropout was on
    assert en < 1.5 * e1 / np.sqrt(N), (e1, en)                  # ~ e1 / sqrt(N); a biased mask or scale floors at the bias
    assert en < 0.012, (e1, en)                                  # (a 2 % error of the rate or the scale would leave 0.02 here)


@pytest.mark.parametrize("flags_unit", ["BiMAU", "MAU"])
def test_attention_dropout_backward_uses_the_forward_masks(flags_unit):
    """With a fixed (seed, step, stream) the dropout mask is a pure function of the element index, so the unit is a deterministic
    function of its input and the gradient of the backward sweeps — which re-derive the masks — must match a central finite
    difference of the forward (float32 path).  A sweep that dropped other (query, key) pairs than the forward would be off by O(1)."""
    from easydgl_amd import ops
    from easydgl_amd.module import temporal as T
    B, Tn, C, H, E, rate = 2, 37, 64, 4, 16, 0.25
    att, x, ids, spans, marks = _bimau_problem(B, Tn, C, H, E, rate, 33)
    if flags_unit == "MAU":
        att = T.MAU(C, H, E, rate, gen=_gen(34)).cuda()
        with torch.no_grad():
            att.scaling.add_(0.2 * torch.randn(E, generator=_gen(35)).cuda())
    drop = ops.Drop(rate, ops.make_rng_state("cuda", seed=11), 3)
    rng = np.random.default_rng(3)
    w = torch.tensor(rng.standard_normal((B, Tn, C)), dtype=torch.float32).cuda()
    v = torch.tensor(rng.standard_normal((B, Tn, C)), dtype=torch.float32).cuda()
    v = v / v.norm()

    def f(xx):
        o, lam = att(xx, xx, ids, spans, marks, True, drop=drop)
        return (o * w).sum() + 0.1 * lam.sum()

    xg = x.clone().requires_grad_()
    f(xg).backward()
    analytic = float((xg.grad * v).sum())
    eps = 2e-2
    with torch.no_grad():
        numeric = float(f(x + eps * v) - f(x - eps * v)) / (2 * eps)
    assert abs(analytic - numeric) < 2e-2 * max(abs(numeric), float(xg.grad.norm()) * 0.05), (analytic, numeric)

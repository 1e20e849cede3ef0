"""The BiMAU kernels leave out the key tiles that hold nothing but left padding (KeyMask::kt0, csrc/bimau_common.h;
BiMAU.__call__ temporal.py:404-452, left padding data/linkpred.py:142-157).  P is exactly 0 on a padded key unless the whole
sequence is padding (temporal.py:425-429), so the skipped form must reproduce the kernels that walk every key tile BIT FOR BIT —
with and without dropout (hashed and stored keep bits) — and the fp64 oracle within the bf16 tolerances of test_gpu_ops.py.
Lengths on both sides of every tile boundary, a single real key, an all-padding sequence inside a mixed batch, padding that is
not a prefix (a real key in front of padded tiles: nothing may be skipped past it)."""
import os

import numpy as np
import pytest
import torch

from oracle import easydgl_oracle as O
from oracle import torch_ref as R
from tests._util import assert_close

pytestmark = pytest.mark.gpu


def ops():
    from easydgl_amd import ops as _ops
    return _ops


class _skip_env:
    """EDGL_BIMAU_SKIP is read by the library at every launch (bimau_skip_enabled)."""

    def __init__(self, on):
        self.on = on

    def __enter__(self):
        self.old = os.environ.get("EDGL_BIMAU_SKIP")
        os.environ["EDGL_BIMAU_SKIP"] = "1" if self.on else "0"

    def __exit__(self, *a):
        if self.old is None:
            os.environ.pop("EDGL_BIMAU_SKIP", None)
        else:
            os.environ["EDGL_BIMAU_SKIP"] = self.old


def _lengths(T):
    """Real-token counts n (left padding T - n) around every key-tile boundary, plus 0 (all padding), 1 and T."""
    ns = {0, 1, 2, T - 1, T}
    for first in range(16, T, 16):          # first real key at `first`: n = T - first
        for d in (-1, 0, 1):
            n = T - (first + d)
            if 0 <= n <= T:
                ns.add(n)
    return sorted(ns)


def _case(T, C, H, E, rng, extra_rows=()):
    ns = _lengths(T)
    B = len(ns) + len(extra_rows)
    ids = rng.integers(1, 30, size=(B, T))
    for b, n in enumerate(ns):
        ids[b, :T - n] = 0
    for j, fn in enumerate(extra_rows):
        fn(ids[len(ns) + j])
    return ns, B, ids


def _run(lib, o, code, t, rate, state, sid, bits, order=None):
    B, T, C, H, E = t["shape"]
    dh = C // H
    out = torch.empty((B, T, C), device="cuda", dtype=torch.bfloat16)
    lam = torch.empty((H * B, T, E), device="cuda")
    saved = torch.empty(lib.edgl_bimau_saved_bytes(B, T, C, H, code), device="cuda", dtype=torch.uint8)
    from easydgl_amd import _lib
    _lib.check(lib.edgl_bimau_fwd_ord(t["qkvt"].data_ptr(), t["resid"].data_ptr(), C, t["ids"].data_ptr(), t["spans"].data_ptr(), t["marks"].data_ptr(),
                                      t["pack"].data_ptr(), B, T, C, H, E, rate, state.data_ptr() if state is not None else None, sid,
                                      None if bits is None else bits.data_ptr(), 0.0, out.data_ptr(), lam.data_ptr(), saved.data_ptr(), None,
                                      None if order is None else order.data_ptr(), 0, code, None), "edgl_bimau_fwd_ord")
    dq = torch.full_like(t["qkvt"], float("nan"))
    n1, n2, n3 = (dh + 1) * dh * E, dh * E, E * dh
    g = torch.empty(n1 + n2 + n3 + E, device="cuda")
    ws = torch.empty(lib.edgl_bimau_bwd_workspace(B, T, C, H, E, code), device="cuda", dtype=torch.uint8)
    _lib.check(lib.edgl_bimau_bwd_ord(t["qkvt"].data_ptr(), t["ids"].data_ptr(), t["spans"].data_ptr(), t["marks"].data_ptr(), t["pack"].data_ptr(),
                                      t["d_out"].data_ptr(), t["d_lam"].data_ptr(), None, 0, None, 0.0, None, lam.data_ptr(), saved.data_ptr(),
                                      B, T, C, H, E, rate, state.data_ptr() if state is not None else None, sid,
                                      None if bits is None else bits.data_ptr(), 0.0, dq.data_ptr(), g.data_ptr(), g[n1:].data_ptr(),
                                      g[n1 + n2:].data_ptr(), g[n1 + n2 + n3:].data_ptr(), ws.data_ptr(),
                                      None if order is None else order.data_ptr(), 0, code, None), "edgl_bimau_bwd_ord")
    torch.cuda.synchronize()
    # the Q columns of d_qkvt rows / everything else: all four column blocks are written by the two sweeps
    assert not torch.isnan(dq.float()).any()
    return out, lam, dq, g


def _tensors(B, T, C, H, E, ids, rng):
    from easydgl_amd import _lib
    lib = _lib.lib
    o = ops()
    dh = C // H
    t = {"shape": (B, T, C, H, E)}
    t["qkvt"] = torch.tensor(rng.standard_normal((B, T, 4 * C)) * 0.4, dtype=torch.bfloat16).cuda()
    t["resid"] = torch.tensor(rng.standard_normal((B, T, C)), dtype=torch.bfloat16).cuda()
    t["ids"] = torch.tensor(ids).cuda()
    t["spans"] = torch.tensor(rng.uniform(0, 5, size=(B, T)), dtype=torch.float32).cuda()
    t["marks"] = torch.tensor(O.synthetic_mark_table(30, E, multi_hot=True)[ids].astype(np.uint8)).cuda()
    t["W1"] = torch.tensor(rng.standard_normal((dh + 1, dh * E)) * 0.2, dtype=torch.float32).cuda()
    t["b1"] = torch.tensor(rng.standard_normal(dh * E) * 0.1, dtype=torch.float32).cuda()
    t["w"] = torch.tensor(rng.standard_normal((E, dh)) * 0.3, dtype=torch.float32).cuda()
    t["sc"] = torch.tensor(rng.standard_normal(E) * 0.1, dtype=torch.float32).cuda()
    t["d_out"] = torch.tensor(rng.standard_normal((B, T, C)), dtype=torch.bfloat16).cuda()
    t["d_lam"] = torch.tensor(rng.standard_normal((H * B, T, E)) * 0.01, dtype=torch.float32).cuda()
    code = o._code(t["qkvt"])
    t["pack"] = torch.empty(lib.edgl_bimau_pack_bytes(C, H, E, code), device="cuda", dtype=torch.uint8)
    _lib.check(lib.edgl_bimau_pack(t["W1"].data_ptr(), t["b1"].data_ptr(), t["w"].data_ptr(), t["sc"].data_ptr(), C, H, E, t["pack"].data_ptr(), code, None), "pack")
    return lib, o, code, t


@pytest.mark.parametrize("T", [101, 32, 33, 128, 17])
@pytest.mark.parametrize("mode", ["nodrop", "hash", "bits"])
def test_skipped_key_tiles_reproduce_the_full_walk_bit_for_bit(T, mode):
    rng = np.random.default_rng(100 + T)
    C, H, E = 32, 2, 16

    def interior(row):      # a real key in front, then padding: only tiles BEHIND the first real key may not be skipped
        row[1:T - 3] = 0

    def holes(row):         # left padding followed by padding holes between real keys
        row[:T // 2] = 0
        row[T // 2 + 2:T - 2] = 0

    ns, B, ids = _case(T, C, H, E, rng, extra_rows=(interior, holes))
    lib, o, code, t = _tensors(B, T, C, H, E, ids, rng)
    rate, sid, state, bits = 0.0, 0, None, None
    if mode != "nodrop":
        rate, sid = 0.3, 9
        state = o.make_rng_state("cuda", seed=31)
        o.rng_advance(state)
    if mode == "bits":
        from easydgl_amd import _lib
        bits = torch.zeros(int(lib.edgl_bimau_dropbits_bytes(B, T, H)) // 4, device="cuda", dtype=torch.int32)
        _lib.check(lib.edgl_bimau_dropbits(B, T, H, rate, state.data_ptr(), sid, bits.data_ptr(), None), "edgl_bimau_dropbits")
    with _skip_env(False):
        full = _run(lib, o, code, t, rate, state, sid, bits)
    with _skip_env(True):
        skip = _run(lib, o, code, t, rate, state, sid, bits)
    for a, b_, what in zip(full, skip, ("out", "lambda", "d_qkvt", "weight gradients")):
        assert torch.equal(a, b_), (what, T, mode, float((a.float() - b_.float()).abs().max()))
    # ... and in the launch order of edgl_bimau_job_order (long sequences first): the order decides when a job runs, nothing else
    from easydgl_amd import _lib
    order = torch.full((2 * B,), -1, device="cuda", dtype=torch.int32)
    _lib.check(lib.edgl_bimau_job_order(t["ids"].data_ptr(), B, T, order.data_ptr(), None), "edgl_bimau_job_order")
    with _skip_env(True):
        ordered = _run(lib, o, code, t, rate, state, sid, bits, order=order)
    for a, b_, what in zip(full, ordered, ("out", "lambda", "d_qkvt", "weight gradients")):
        assert torch.equal(a, b_), (what, T, mode, "ordered")
    # gradients of K / V / T_ on the rows of a padded key are exactly zero (and written: the buffer was NaN-filled)
    dq = skip[2].float().view(B, T, 4, C)
    for b, n in enumerate(ns):
        if 0 < n < T:
            assert float(dq[b, :T - n, 1:].abs().max()) == 0.0, (b, n)


@pytest.mark.parametrize("T", [101, 40])
def test_skipped_key_tiles_against_the_fp64_oracle(T):
    """The same length sweep through the autograd op against oracle/torch_ref.py::bimau on the rounded operands (bf16 bounds of
    test_gpu_ops.py::test_bimau_fwd_bwd)."""
    o = ops()
    rng = np.random.default_rng(7 + T)
    C, H, E = 64, 4, 16
    dh = C // H
    ns, B, ids = _case(T, C, H, E, rng)
    cin = 3 * C
    x = rng.standard_normal((B, T, cin))
    mt = O.synthetic_mark_table(30, E, multi_hot=True)
    marks = mt[ids]
    spans = rng.uniform(0, 5, size=(B, T))
    W = dict(Wq=rng.standard_normal((cin, 4 * C)) * 0.15, bq=rng.standard_normal(4 * C) * 0.1, W1=O.glorot_uniform(rng, (dh + 1, dh * E)),
             b1=rng.standard_normal(dh * E) * 0.1, w=O.glorot_uniform(rng, (E, dh)), sc=rng.standard_normal(E) * 0.2)
    dt = torch.bfloat16
    xt = torch.tensor(x, dtype=dt).cuda().requires_grad_()
    Wq = torch.tensor(W["Wq"], dtype=torch.float32).cuda().requires_grad_()
    Wq_c = Wq.detach().to(dt)
    bq = torch.tensor(W["bq"], dtype=torch.float32).cuda().requires_grad_()
    W1, b1, w, sc = (torch.tensor(W[k], dtype=torch.float32).cuda().requires_grad_() for k in ("W1", "b1", "w", "sc"))
    qkvt = o.LinearFn.apply(xt, Wq, bq, Wq_c, False)
    out, lam = o.BiMAUFn.apply(qkvt, xt[:, :, :C], W1, b1, w, sc, torch.tensor(ids).cuda(), torch.tensor(spans, dtype=torch.float32).cuda(),
                               torch.tensor(marks.astype(np.uint8)).cuda(), H, o.NO_DROP)
    G1 = torch.tensor(rng.standard_normal((B, T, C)), dtype=dt).cuda()
    G2 = torch.tensor(rng.standard_normal((H * B, T, E)) * 0.3, dtype=torch.float32).cuda()
    ((out.float() * G1.float()).sum() + (lam * G2).sum()).backward()
    xr = xt.detach().double().cpu().requires_grad_()
    pr = {"dense/kernel": Wq_c.double().cpu().requires_grad_(), "dense/bias": bq.detach().double().cpu().requires_grad_(),
          "sequential_temporal_combined/dense/kernel": W1.detach().double().cpu().requires_grad_(),
          "sequential_temporal_combined/dense/bias": b1.detach().double().cpu().requires_grad_(),
          "sequential_temporal_combined/weight": w.detach().double().cpu().requires_grad_(),
          "sequential_temporal_combined/scaling": sc.detach().double().cpu().requires_grad_()}
    km3 = torch.tensor((ids != 0).astype(np.float64)).unsqueeze(1).repeat(H, T, 1)
    out_r, lam_r = R.bimau(C, H, xr, km3, torch.tensor(spans), torch.tensor(marks, dtype=torch.float64), pr, "", 0.0, False)
    ((out_r * G1.double().cpu()).sum() + (lam_r * G2.double().cpu()).sum()).backward()
    assert_close(lam.detach().cpu().numpy(), lam_r.detach().numpy(), 3e-2, "lambda")
    assert_close(out.float().detach().cpu().numpy(), out_r.detach().numpy(), 3e-2, "out")
    gtol = 6e-2
    assert_close(xt.grad.float().cpu().numpy(), xr.grad.numpy(), gtol, "dx")
    assert_close(Wq.grad.cpu().numpy(), pr["dense/kernel"].grad.numpy(), gtol, "dWqkvt")
    assert_close(W1.grad.cpu().numpy(), pr["sequential_temporal_combined/dense/kernel"].grad.numpy(), gtol, "dW1")
    assert_close(b1.grad.cpu().numpy(), pr["sequential_temporal_combined/dense/bias"].grad.numpy(), gtol, "db1")
    assert_close(w.grad.cpu().numpy(), pr["sequential_temporal_combined/weight"].grad.numpy(), gtol, "dw")
    assert_close(sc.grad.cpu().numpy(), pr["sequential_temporal_combined/scaling"].grad.numpy(), gtol, "dscaling")


@pytest.mark.parametrize("B,T", [(37, 101), (512, 101), (5, 17), (1100, 40), (3, 201)])
def test_job_order_lists_the_samples_by_falling_key_tile_count(B, T):
    """edgl_bimau_job_order: a permutation of 0 .. B-1, stable, by the number of key tiles from the first real key on (a sequence
    without a real key walks all of them: uniform softmax, temporal.py:425-429)."""
    from easydgl_amd import _lib
    lib = _lib.lib
    rng = np.random.default_rng(B + T)
    ids = rng.integers(1, 50, size=(B, T))
    for b in range(B):
        ids[b, :rng.integers(0, T + 1)] = 0
    ids[B // 2, :] = 0
    if B > 3:
        ids[3, 1:] = 0          # one real key in front: nothing is skipped
    nt = (T + 15) // 16
    first = np.where((ids != 0).any(1), (ids != 0).argmax(1), 0)
    nk = nt - first // 16
    want = np.argsort(-nk, kind="stable")
    order = torch.full((2 * B,), -1, device="cuda", dtype=torch.int32)
    _lib.check(lib.edgl_bimau_job_order(torch.tensor(ids).cuda().data_ptr(), B, T, order.data_ptr(), None), "edgl_bimau_job_order")
    torch.cuda.synchronize()
    assert np.array_equal(order[:B].cpu().numpy(), want)


def test_engine_step_in_job_order_equals_the_index_order(monkeypatch):
    """TrainEngine with EDGL_BIMAU_ORDER=1 (edgl_bimau_job_order on the side stream, edgl_bimau_fwd_ord / _bwd_ord): same loss, same
    gradients bit for bit as in index order — with dropout (stored keep bits) and the TPP term inside sweep 1.  The batch keeps
    its left padding free of MASK tokens (masked positions drawn inside the sequences), so the order is not the identity."""
    from easydgl_amd.engine import TrainEngine
    from tests._util import build_model, make_problem, to_dev
    prob = make_problem(seed=77, batch=24, num_items=400, seqslen=40, num_units=64, num_heads=4, num_blocks=2, masklen=3, num_events=16)
    cfg = prob["cfg"]
    rng = np.random.default_rng(5)
    ids, ts = prob["ids"].copy(), prob["ts"]
    T = cfg.seqslen + 1
    for b in range(ids.shape[0]):
        n = int(rng.integers(4, T + 1))
        ids[b, :T - n] = 0
    mp = np.stack([np.sort(rng.choice(np.arange(T - 4, T), size=cfg.masklen, replace=False)) for _ in range(ids.shape[0])])
    feats, labels = O.mask_random(cfg, ids, ts, mp)
    feats, labels = to_dev(feats), torch.as_tensor(labels).cuda()
    grads = []
    for order in ("0", "1"):
        monkeypatch.setenv("EDGL_BIMAU_ORDER", order)
        m = build_model(prob, "bf16", hidden_drop=0.1, att_drop=0.1)
        eng = TrainEngine(m, ids.shape[0], use_graph=False)
        assert (eng.job_order is not None) == (order == "1")
        eng.load_batch(feats, labels)
        eng._issue()
        torch.cuda.synchronize()
        if order == "1":
            o = eng.job_order[:ids.shape[0]].cpu().numpy()
            assert sorted(o.tolist()) == list(range(ids.shape[0])) and not np.array_equal(o, np.arange(ids.shape[0]))
        grads.append((float(eng.loss), {n: p_.grad.clone() for n, p_ in m.named_parameters()}))
    assert grads[0][0] == grads[1][0]
    for n, g0 in grads[0][1].items():
        g1 = grads[1][1][n]
        if "lookup_table" in n:     # the embedding tables take f32 atomics (k_encode.hip): equal up to the order of the sums
            assert float((g0 - g1).abs().max()) <= 1e-5 * float(g0.abs().max()) + 1e-12, n
        else:
            assert torch.equal(g0, g1), n


def test_training_batches_of_the_reference_masker_leave_no_key_tile_to_skip():
    """Why the skip does not move the TRAINING step (DESIGN.md rule 50): MAUPostProcessor.mask_random (dataloader.py:187-191)
    draws the masked positions over all positions 1 .. T-1, padding included; the MASK token is id num_items != 0, i.e. a real key
    for BiMAU's key mask (temporal.py:425-426, `pad keys masked, MASK keys not`).  With 20 of 100 positions masked almost every
    key tile of the benchmark's left-padded batches holds a MASK token; evaluation batches (mask_last) keep their padding."""
    from easydgl_amd import data as D
    num_items, L, M, B = 20000, 100, 20, 512
    T = L + 1
    ids, ts = D.synthetic_batch(num_items, L, B, seed=9876)
    g = torch.Generator().manual_seed(9876)
    mp = D.draw_masked_positions(B, T, M, generator=g)
    feats, _ = D.mask_random(torch.tensor(ids), torch.tensor(ts), num_items, mp)
    nt = (T + 15) // 16

    def skippable(a):
        a = np.asarray(a)
        real = np.zeros((B, nt * 16), bool)
        real[:, :T] = a != 0
        any_real = real.reshape(B, nt, 16).any(-1)
        first = np.where(any_real.any(1), any_real.argmax(1), 0)
        return first.sum() / (B * nt)

    padded = float((ids == 0).mean())
    assert 0.40 < padded < 0.55
    assert skippable(ids) > 0.30                         # evaluation-style batch: a third of the key tiles are padding only
    assert skippable(feats["seqs_i"].numpy()) < 0.03     # training batch of the reference's masker: hardly any

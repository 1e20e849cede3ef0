"""The TPP regulariser inside the attention kernels (csrc/bimau_common.h TppDesc: edgl_tpp_prep,
edgl_bimau_bwd_tpp, edgl_tpp_finish_parts) against the separate launches it replaces (edgl_tpp_fwd_bwd_rows on a d lambda
array) and against the fp64 oracle — MAU.biased_likelihood, temporal.py:317-333, at the masked positions of EasyDGL.py:157-175."""
import ctypes

import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from tests._util import GRAD_TOL, build_model, grad_ok, make_problem, rel_err, to_dev

pytestmark = pytest.mark.gpu

HEADLINE = dict(num_units=128, num_heads=8, num_blocks=1, seqslen=100, masklen=20, num_events=16, num_items=2000)
SMALL = dict(num_units=32, num_heads=2, num_blocks=2, seqslen=21, masklen=5, num_events=16, num_items=300)


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _raw_span(ts_row, t):
    T = len(ts_row)
    if T < 2:
        return 0.0
    t1 = 1 if t == 0 else t
    return float(min(max(np.float32(ts_row[t1]) - np.float32(ts_row[t1 - 1]), 0.0), 100.0))


def _prep_reference(mpos, labels, ts, mtab, T):
    """Slot data as csrc/k_misc.hip tpp_prep_kernel defines it, slot by slot."""
    B, M = mpos.shape
    nm = np.zeros((B, T, 16), np.uint8)
    spr = -np.ones((B, T), np.float32)
    novf = np.zeros(B, np.int32)
    ovf_pos = np.zeros((B, M), np.int32)
    ovf_nm = np.zeros((B, M, 16), np.uint8)
    for b in range(B):
        taken = set()
        for m in range(M):
            pos, row = int(mpos[b, m]), mtab[int(labels[b, m])]
            if not (0 <= pos < T) or row.sum() == 0:
                continue
            if pos not in taken:
                taken.add(pos)
                nm[b, pos] = row
                spr[b, pos] = _raw_span(ts[b], pos)
            else:
                ovf_pos[b, novf[b]] = pos
                ovf_nm[b, novf[b]] = row
                novf[b] += 1
    return nm, spr, novf, ovf_pos, ovf_nm


def _with_repeats(prob, rng):
    """A batch whose masked positions repeat: same label (a multiplicity), another label with marks, a label without marks."""
    feats = {k: np.array(v, copy=True) for k, v in prob["feats"].items()}
    labels = np.array(prob["labels"], copy=True)
    mp = feats["masked_positions"]
    B, M = mp.shape
    empty = [i for i in range(prob["mark_table"].shape[0]) if prob["mark_table"][i].sum() == 0]
    for b in range(B):
        if M >= 3:
            mp[b, 1] = mp[b, 0]                       # same position, its own label: two effective slots with different mark rows
            mp[b, 2] = mp[b, 0]
            labels[b, 2] = labels[b, 0]               # ... and a third one with the first slot's label
        if M >= 5 and empty:
            mp[b, 4] = mp[b, 3]
            labels[b, 4] = empty[0]                   # a slot without marks on a taken position: adds nothing
        if b % 2 and M >= 4:
            mp[b, 3] = int(rng.integers(0, mp.max() + 1))
    feats["masked_positions"] = mp
    return feats, labels


@pytest.mark.parametrize("shape", [SMALL, HEADLINE])
def test_tpp_prep_lists_the_slots_of_every_position(shape):
    from easydgl_amd import _lib
    lib = _lib.lib
    prob = make_problem(seed=5, batch=6, **shape)
    rng = np.random.default_rng(3)
    feats, labels = _with_repeats(prob, rng)
    cfg = prob["cfg"]
    B, M = labels.shape
    T = feats["seqs_i"].shape[1]
    mtab = np.ascontiguousarray(prob["mark_table"], dtype=np.uint8)
    d_mp, d_lab = torch.as_tensor(feats["masked_positions"]).cuda(), torch.as_tensor(labels).cuda()
    d_ts, d_mt = torch.as_tensor(feats["seqs_t"]).float().cuda().contiguous(), torch.as_tensor(mtab).cuda()
    nbytes = int(lib.edgl_tpp_prep_bytes(B, T, M))
    desc = torch.full((nbytes,), 0x5a, dtype=torch.uint8, device="cuda")
    rc = lib.edgl_tpp_prep(_ptr(d_mp), _ptr(d_lab), _ptr(d_ts), _ptr(d_mt), B, T, cfg.num_events, M, _ptr(desc), None)
    assert rc == 0, (lib.edgl_last_error() or b"").decode()
    torch.cuda.synchronize()
    raw = desc.cpu().numpy()
    off_spr = B * T * 16
    off_novf = off_spr + B * T * 4
    off_pos = (off_novf + B * 4 + 15) & ~15
    off_nm = (off_pos + B * M * 4 + 15) & ~15
    assert nbytes == off_nm + B * M * 16 + ((B * 4 + 15) & ~15)
    got_cnt = raw[off_nm + B * M * 16:off_nm + B * M * 16 + B * 4].view(np.int32)      # per-sample marks: the normaliser (temporal.py:330)
    np.testing.assert_array_equal(got_cnt, mtab[labels].astype(np.int64).sum(axis=(1, 2)))
    nm, spr, novf, ovf_pos, ovf_nm = _prep_reference(feats["masked_positions"], labels, np.asarray(feats["seqs_t"], np.float32), mtab, T)
    assert novf.sum() > 0      # the batch does exercise the overflow list
    np.testing.assert_array_equal(raw[:off_spr].reshape(B, T, 16), nm)
    np.testing.assert_array_equal(raw[off_spr:off_novf].view(np.float32).reshape(B, T), spr)
    got_novf = raw[off_novf:off_novf + B * 4].view(np.int32)
    np.testing.assert_array_equal(got_novf, novf)
    got_pos = raw[off_pos:off_pos + B * M * 4].view(np.int32).reshape(B, M)
    got_nm = raw[off_nm:off_nm + B * M * 16].reshape(B, M, 16)
    for b in range(B):
        np.testing.assert_array_equal(got_pos[b, :novf[b]], ovf_pos[b, :novf[b]])
        np.testing.assert_array_equal(got_nm[b, :novf[b]], ovf_nm[b, :novf[b]])


def _engine_step(prob, feats, labels, fused, monkeypatch, drop=0.0, batch=None):
    from easydgl_amd.engine import TrainEngine
    monkeypatch.setenv("EDGL_TPP_FUSED", "1" if fused else "0")
    m = build_model(prob, "bf16", hidden_drop=drop, att_drop=drop)
    B = labels.shape[0]
    eng = TrainEngine(m, B, use_graph=False)
    assert eng.fused_tpp == fused
    eng.load_batch(to_dev(feats), torch.as_tensor(labels).cuda())
    m._grad_arena.fill_(float("nan"))
    eng._issue()
    torch.cuda.synchronize()
    grads = {n: q.grad.float().cpu().numpy().copy() for n, q in m.named_parameters()}    # (the arena's alignment gaps keep the NaN fill)
    return m, eng, dict(loss=float(eng.loss), tpp=float(eng.loss_tpp), grads=grads)


@pytest.mark.parametrize("shape,drop,repeats", [(SMALL, 0.0, False), (SMALL, 0.1, True), (HEADLINE, 0.1, False), (HEADLINE, 0.0, True)])
def test_fused_tpp_matches_the_separate_launches(shape, drop, repeats, monkeypatch):
    """Same weights, batch and dropout streams through both forms of the engine: the regulariser, the loss and every gradient
    (the term reaches the weights through d lambda in sweep 1 only).  f32 arithmetic on both sides; the sums run in another order."""
    prob = make_problem(seed=11, batch=6, **shape)
    feats, labels = (_with_repeats(prob, np.random.default_rng(9)) if repeats else (prob["feats"], prob["labels"]))
    _, _, a = _engine_step(prob, feats, labels, False, monkeypatch, drop)
    _, eng, b = _engine_step(prob, feats, labels, True, monkeypatch, drop)
    assert eng.blk[0]["dlam"] is None
    assert abs(a["tpp"]) > 0
    assert abs(a["tpp"] - b["tpp"]) <= 2e-5 * abs(a["tpp"])
    assert abs(a["loss"] - b["loss"]) <= 2e-5 * abs(a["loss"])
    for n, ga in a["grads"].items():
        gb = b["grads"][n]
        assert np.isfinite(gb).all(), n
        assert rel_err(gb, ga) < 2e-3, n      # (bf16 d_qkvt / dz roundings amplify last-digit differences of d lambda)


def test_fused_tpp_with_repeated_positions_follows_the_oracle(monkeypatch):
    """Repeated masked positions (tf.gather semantics: every slot adds its own term and its own gradient) against the fp64 oracle."""
    prob = make_problem(seed=21, batch=4, **HEADLINE)
    feats, labels = _with_repeats(prob, np.random.default_rng(2))
    m, eng, out = _engine_step(prob, feats, labels, True, monkeypatch)
    cfg = prob["cfg"]
    p64 = R.to_torch_params(prob["params"])
    ref, _ = R.train_loss(cfg, p64, prob["mark_table"], feats, labels)
    ref.backward()
    assert abs(out["loss"] - float(ref)) <= 1e-3 * abs(float(ref))
    bad = {}
    for name, p in m.tf_variable_map().items():
        want = p64[name].grad.numpy().copy()
        if name in ("CSTMA/item_embs/lookup_table", "CSTMA/mark_embs/lookup_table", "CSTMA/spatial_embs/embedding/lookup_table"):
            want -= cfg.l2_reg * prob["params"][name]
        ok, e = grad_ok(p.grad.cpu().numpy(), want, "bf16")
        if not ok:
            bad[name] = e
    assert not bad, (bad, GRAD_TOL["bf16"])


def test_fused_tpp_shapes_it_does_not_take_are_refused():
    from easydgl_amd import _lib
    lib = _lib.lib
    z = torch.zeros(64, dtype=torch.uint8, device="cuda")
    assert lib.edgl_tpp_prep(_ptr(z), _ptr(z), _ptr(z), _ptr(z), 2, 10, 7, 3, _ptr(z), None) != 0     # E != 16
    assert "E = 16" in (lib.edgl_last_error() or b"").decode()
    assert lib.edgl_tpp_prep(_ptr(z), _ptr(z), _ptr(z), _ptr(z), 2, 10, 16, 300, _ptr(z), None) != 0  # M > 256


@pytest.mark.parametrize("shape,dt", [(SMALL, "bf16"), (HEADLINE, "bf16"), (SMALL, "f32")])
def test_batch_preparation_inside_the_encoder_launch_is_the_three_calls(shape, dt, monkeypatch):
    """edgl_encode_fwd_prep — the row compaction map and the regulariser's slot data as the first workgroups of the encoder's launch
    — against edgl_encode_fwd_ct + edgl_compact_scan_labels + edgl_tpp_prep on side streams: every array bit for bit, with repeated
    masked positions and label-0 slots in the batch; the step's loss and gradients follow."""
    from easydgl_amd.engine import TrainEngine
    prob = make_problem(seed=17, batch=6, **shape)
    feats, labels = _with_repeats(prob, np.random.default_rng(4))
    labels = np.array(labels, copy=True)
    labels[1, :2] = 0
    labels[4] = 0          # a sample without weighted rows
    out = {}
    for inside in ("0", "1"):
        monkeypatch.setenv("EDGL_PREP_IN_ENCODER", inside)
        m = build_model(prob, dt, hidden_drop=0.1, att_drop=0.1)
        eng = TrainEngine(m, labels.shape[0], use_graph=False)
        for t in (eng.perm, eng.inv, eng.nvalid, eng.labels_c):
            t.fill_(-7)
        if eng.tpp_desc is not None:
            eng.tpp_desc.fill_(0x5a)
        eng.load_batch(to_dev(feats), torch.as_tensor(labels).cuda())
        eng._issue()
        torch.cuda.synchronize()
        out[inside] = dict(x0=eng.x0.float().clone(), spans=eng.spans.clone(), marks=eng.marks.clone(), perm=eng.perm.clone(),
                           inv=eng.inv.clone(), nvalid=eng.nvalid.clone(), labels_c=eng.labels_c.clone(),
                           desc=None if eng.tpp_desc is None else eng.tpp_desc.clone(), loss=float(eng.loss),
                           grads={n: q.grad.float().cpu().numpy().copy() for n, q in m.named_parameters()})
        assert (eng.tpp_desc is not None) == (dt == "bf16" and shape is HEADLINE or eng.fused_tpp)
    a, b = out["0"], out["1"]
    assert int(a["nvalid"]) == int((labels != 0).sum())
    for k in ("x0", "spans", "marks", "perm", "inv", "nvalid", "labels_c", "desc"):
        if a[k] is None:
            assert b[k] is None
            continue
        assert torch.equal(a[k], b[k]), k
    assert abs(a["loss"] - b["loss"]) <= 1e-6 * abs(a["loss"])
    for n, ga in a["grads"].items():
        assert rel_err(b["grads"][n], ga) < 1e-5, n       # (same kernels on the same inputs; f32 atomics of the scatters reorder)


def test_encode_fwd_prep_refuses_slot_data_it_cannot_build():
    from easydgl_amd import _lib
    lib = _lib.lib
    z = torch.zeros(4096, dtype=torch.uint8, device="cuda")
    args = [_ptr(z)] * 7 + [2, 10, 16, 7, 50, 49, 1.0, 0.0, None, 0] + [_ptr(z)] * 3 + [0, 0, _ptr(z), 3] + [_ptr(z)] * 6
    assert lib.edgl_encode_fwd_prep(*args, 1, None) != 0                    # slot data with E != 16
    assert "E = 16" in (lib.edgl_last_error() or b"").decode()

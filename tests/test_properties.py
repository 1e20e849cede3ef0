"""Size-independent properties of the attention path, checked on the oracle (CPU, hypothesis) and on the
HIP kernels (GPU): BiMAU has no causal mask and no positional term of its own, so permuting the positions of a sequence
(inputs, ids, intervals, marks together) permutes its outputs the same way; samples never see each other; the causal kernels
do not let a future key change an earlier output."""
import numpy as np
import pytest
import torch
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import easydgl_oracle as O


def _case(seed, B, T, C, h, E):
    rng = np.random.default_rng(seed)
    cfg = O.Config(num_items=40, seqslen=T - 1, num_units=C, num_heads=h, num_blocks=1, masklen=1, time_scale=1.0, ct_reg=0.0,
                   l2_reg=0.0, learning_rate=1e-3, num_events=E)
    dh = C // h
    x = rng.standard_normal((B, T, C))
    ids = rng.integers(1, 40, size=(B, T))
    ids[0, : T // 3] = 0
    marks = O.synthetic_mark_table(40, E, multi_hot=True)[ids]
    spans = rng.uniform(0, 5, size=(B, T))
    W = dict(Wqkvt=0.2 * rng.standard_normal((C, 4 * C)), bqkvt=0.1 * rng.standard_normal(4 * C),
             W1=O.glorot_uniform(rng, (dh + 1, dh * E)), b1=0.1 * rng.standard_normal(dh * E), w=O.glorot_uniform(rng, (E, dh)),
             scaling=0.2 * rng.standard_normal(E))
    return cfg, x, ids, marks, spans, W


@settings(max_examples=15, deadline=None)
@given(seed=st.integers(0, 10_000), T=st.integers(2, 12), h=st.sampled_from([1, 2]), E=st.integers(2, 5))
def test_oracle_bimau_is_equivariant_under_position_permutations(seed, T, h, E):
    cfg, x, ids, marks, spans, W = _case(seed, 2, T, 8 * h, h, E)
    out, lam = O.bimau(cfg, x, (ids != 0).astype(np.float64), spans, marks.astype(np.float64), **W)
    perm = np.random.default_rng(seed + 1).permutation(T)
    out_p, lam_p = O.bimau(cfg, x[:, perm], (ids[:, perm] != 0).astype(np.float64), spans[:, perm], marks[:, perm].astype(np.float64), **W)
    np.testing.assert_allclose(out_p, out[:, perm], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(lam_p, lam[:, perm], rtol=1e-9, atol=1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 3e-2)])
@pytest.mark.parametrize("T", [11, 40, 101])
def test_hip_bimau_is_equivariant_and_sample_independent(dt, tol, T):
    from easydgl_amd import ops as o
    from tests._util import assert_close
    B, C, h, E = 3, 32, 2, 4
    cfg, x, ids, marks, spans, W = _case(T, B, T, C, h, E)
    dev = lambda a, d=torch.float32: torch.tensor(np.ascontiguousarray(a), dtype=d).cuda().contiguous()
    Wq, bq, W1, b1, w, sc = dev(W["Wqkvt"]), dev(W["bqkvt"]), dev(W["W1"]), dev(W["b1"]), dev(W["w"]), dev(W["scaling"])

    def run(xx, ii, ss, mm):
        xt = dev(xx, dt)
        qkvt = o.LinearFn.apply(xt, Wq, bq, Wq.to(dt), False)
        out, lam = o.BiMAUFn.apply(qkvt, xt, W1, b1, w, sc, dev(ii, torch.int64), dev(ss), dev(mm, torch.uint8),
                                   h, o.NO_DROP)
        return out.float().cpu().numpy(), lam.cpu().numpy()

    out, lam = run(x, ids, spans, marks)
    perm = np.random.default_rng(1).permutation(T)
    out_p, lam_p = run(x[:, perm], ids[:, perm], spans[:, perm], marks[:, perm])
    assert_close(out_p, out[:, perm], tol, "permuted output")
    assert_close(lam_p, lam[:, perm], tol, "permuted intensity")
    out_1, lam_1 = run(x[1:2], ids[1:2], spans[1:2], marks[1:2])       # a sample alone == the same sample inside a batch
    assert np.array_equal(out_1[0], out[1]) and np.array_equal(lam_1[0], lam[h * 0 + 1]) and np.array_equal(lam_1[1], lam[B + 1])


@pytest.mark.gpu
@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_hip_causal_attention_ignores_future_keys(dt):
    """edgl_tattn_fwd with the causal flag: changing keys / values at positions > q leaves rows <= q bit-identical."""
    from tests.test_gpu_tgat import _operands, _tattn
    B, T, H, Dq, Dv = 2, 45, 2, 48, 16
    qx, kx, v, resid, _, ids = [t.cuda() for t in _operands(3, B, T, H, Dq, Dv, dt)]
    ids[:] = 1                                          # no padding: every row has an unmasked key
    out, _ = _tattn(qx, kx, v, resid, ids, H, 0.25, saved=False)
    kx2, v2 = kx.clone(), v.clone()
    kx2[:, 30:] += 1.0
    v2[:, 30:] -= 2.0
    out2, _ = _tattn(qx, kx2, v2, resid, ids, H, 0.25, saved=False)
    assert torch.equal(out[:, :30], out2[:, :30]) and not torch.equal(out[:, 30:], out2[:, 30:])

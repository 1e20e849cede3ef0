"""The two-workgroups-per-CU form of the fused block tail (csrc/k_tail.hip, namespace t2: C = 128, T <= 101; three unpadded,
swizzled LDS images, <= 128 registers) against the one-workgroup-per-CU kernels it replaces at that shape: the SAME arithmetic in
the same order, so every saved tensor, the gathered head rows, the loss and every gradient must agree BIT FOR BIT
(EasyDGL.py:110-146 and its backward).  The one-per-CU kernels themselves are held to the unfused launches and to the fp64
oracle by tests/test_gpu_engine.py."""
import pytest
import torch

from tests._util import build_model, make_problem, to_dev

pytestmark = pytest.mark.gpu

CASES = {
    "headline": dict(num_units=128, num_heads=8, num_blocks=1, seqslen=100, masklen=20, num_events=16, num_items=2000),
    "two_blocks": dict(num_units=128, num_heads=8, num_blocks=2, seqslen=100, masklen=20, num_events=16, num_items=500),   # a block without the head
    "t31": dict(num_units=128, num_heads=8, num_blocks=1, seqslen=30, masklen=6, num_events=16, num_items=700),           # two row tiles
    "t50": dict(num_units=128, num_heads=4, num_blocks=2, seqslen=49, masklen=9, num_events=7, num_items=300),            # four row tiles
    "t97": dict(num_units=128, num_heads=8, num_blocks=1, seqslen=96, masklen=20, num_events=16, num_items=300),          # one row in the last tile
    "padded": dict(num_units=100, num_heads=2, num_blocks=2, seqslen=40, masklen=8, num_events=5, num_items=200),         # head dim 50 stored as 64
}


def _run(prob, batch, drop, variant):
    from easydgl_amd import _lib
    from easydgl_amd.engine import TrainEngine
    prev = _lib.lib.edgl_tail_variant(variant)
    try:
        m = build_model(prob, "bf16", hidden_drop=drop, att_drop=drop)
        eng = TrainEngine(m, batch, use_graph=False)
        assert eng.fused_tail
        eng.load_batch(to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).cuda())
        eng._issue()
        torch.cuda.synchronize()
        out = dict(loss=eng.loss.clone(), grads=m._grad_arena.clone(), pre_t=eng.pre_t.clone(), so=eng.so.clone(), st3=eng.st3.clone(),
                   hrows=eng.hrows_c[:int(eng.nvalid[0])].clone())
        for i, b in enumerate(eng.blk):
            for k in ("ao", "a1", "pre_f", "f", "o", "y", "st1", "st2"):
                out[f"{k}{i}"] = b[k].clone()
        for k in ("d_pre_t", "d_o", "d_pre_f", "d_ao", "G1", "G2"):
            out[k] = getattr(eng, k).clone()
        return out
    finally:
        _lib.lib.edgl_tail_variant(prev)


@pytest.mark.parametrize("name,drop", [("headline", 0.0), ("headline", 0.1), ("two_blocks", 0.1), ("t31", 0.1), ("t50", 0.0), ("t97", 0.1),
                                       ("padded", 0.1)])
def test_two_per_cu_tail_is_bit_identical_to_the_one_per_cu_kernels(name, drop):
    batch = 9
    prob = make_problem(seed=70 + len(name), batch=batch, **CASES[name])
    a = _run(prob, batch, drop, 0)
    b = _run(prob, batch, drop, 1)
    assert set(a) == set(b)
    # (the embedding scatter's f32 atomics — the item table's gradient — commute only up to rounding: compare everything else
    #  bit for bit and the arena to rounding)
    for k in a:
        if k in ("grads", "loss"):
            continue
        assert torch.equal(a[k], b[k]), (k, float((a[k].float() - b[k].float()).abs().max()))
    assert abs(float(a["loss"]) - float(b["loss"])) <= 1e-6 * abs(float(a["loss"]))
    assert float((a["grads"] - b["grads"]).abs().max()) <= 1e-5 * float(a["grads"].abs().max())


def test_tail_variant_switch_and_shapes_outside_the_two_per_cu_form():
    from easydgl_amd import _lib
    lib = _lib.lib
    prev = lib.edgl_tail_variant(-1)
    assert prev in (0, 1) and lib.edgl_tail_variant(-1) == prev
    assert lib.edgl_tail_variant(0) == prev and lib.edgl_tail_variant(1) == 0 and lib.edgl_tail_variant(prev) == 1
    # C = 64 and T > 101 keep the one-per-CU kernels under either setting: the engine runs and matches itself
    for kw in (dict(num_units=64, num_heads=4, num_blocks=1, seqslen=100, masklen=20, num_events=16, num_items=300),
               dict(num_units=128, num_heads=8, num_blocks=1, seqslen=110, masklen=20, num_events=16, num_items=300)):
        prob = make_problem(seed=9, batch=5, **kw)
        a, b = _run(prob, 5, 0.1, 0), _run(prob, 5, 0.1, 1)
        for k in a:
            if k not in ("grads", "loss"):
                assert torch.equal(a[k], b[k]), k

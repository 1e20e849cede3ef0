"""Fused evaluation scoring (csrc/k_eval_topk.hip): rows . table^T + bias -> seen mask -> top-K without a logits tile in HBM
(Sequential.eval, Base.py:150-181; EasyDGL.py:149-151) against an fp64 reference on the same bf16 operands and against the unfused
kernels (edgl_score_lse_fwd logits tile + edgl_mask_topk).  The two GPU paths accumulate in f32 in different orders, so values agree
to f32 rounding and the index lists may differ only where the reference's neighbours are closer than that rounding; tie blocks
(identical table rows) are exact: lower index first (tf.nn.top_k), through the candidate lists' overflow into the exact kernel."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def ops():
    from easydgl_amd import ops as _ops
    return _ops


def _problem(R, C, I, T, seed, dup=0):
    rng = np.random.default_rng(seed)
    rows = torch.tensor(rng.standard_normal((R, C)) * 0.5, dtype=torch.bfloat16).cuda()
    tab = rng.standard_normal((I, C)) * 0.3
    if dup:                    # blocks of identical rows: exact ties between different items
        tab[1 + dup:1 + 2 * dup] = tab[1:1 + dup]
    table = torch.tensor(tab, dtype=torch.bfloat16).cuda()
    bias = torch.tensor(rng.standard_normal(I - 1) * 0.2, dtype=torch.float32).cuda()
    if dup:
        bias[dup:2 * dup] = bias[:dup]
    seen = rng.integers(0, I, size=(R, T))
    seen[:, 0] = 0
    seen[:, 1] = I - 1         # the MASK column
    return rows, table, bias, torch.tensor(seen).cuda()


def _reference(rows, table, bias, seen, i0, i1, K):
    t = table.double().cpu().numpy().copy()
    t[0] = 0.0                                         # coding.py:56-57: the used table's row 0 is the zero constant
    lg = rows.double().cpu().numpy() @ t.T + np.concatenate([[-1000.0], bias.double().cpu().numpy()])
    lg = lg[:, i0:i1].copy()
    s = seen.cpu().numpy()
    for r in range(lg.shape[0]):
        ids = s[r][(s[r] >= i0) & (s[r] < i1)] - i0
        lg[r, ids] = -np.inf
    return lg


ORDER_GAP = 2e-4      # a reference gap above this cannot be reordered by f32 accumulation order (measured errors: ~1e-6 at |logit| <= 12)


def _check(val, idx, lg, seen, i0, K, tol=2e-4, min_cover=0.9, full_lists=False):
    """Membership, values and ORDER: position p of a row is compared with the reference's position p wherever the reference's
    gaps to BOTH neighbours (p - 1 and p + 1, the K + 1-th element included) exceed ORDER_GAP — f32 rounding cannot move such an
    element — and those positions must be >= min_cover of all positions (0.96-0.99 at the shapes below); full_lists: every row's
    whole list equals the reference's (the K = 10 case, where every gap is resolvable)."""
    val, idx = val.cpu().numpy(), idx.cpu().numpy()
    s = seen.cpu().numpy()
    covered = total = 0
    for r in range(lg.shape[0]):
        order = np.lexsort((np.arange(lg.shape[1]), -lg[r]))[:K + 1]
        kth = lg[r, order[K - 1]]
        assert len(set(idx[r].tolist())) == K and idx[r].min() >= i0, r
        assert not np.isin(idx[r], s[r]).any(), r                                    # no seen id
        got = lg[r, idx[r] - i0]
        assert np.all(np.abs(got - val[r]) <= tol * max(1.0, np.abs(got).max())), r     # the values ARE those items' logits
        assert np.all(got >= kth - tol), r                                           # each one belongs to the top K (up to f32 rounding)
        assert np.all(np.diff(val[r]) <= 0), r                                       # descending
        gaps = np.abs(np.diff(lg[r, order]))                                         # K gaps: gaps[p] between positions p and p + 1
        fixed = gaps > ORDER_GAP
        fixed[1:] &= gaps[:-1] > ORDER_GAP
        assert np.array_equal((idx[r] - i0)[fixed], order[:K][fixed]), (r, np.where(fixed & ((idx[r] - i0) != order[:K]))[0])
        covered += int(fixed.sum()); total += K
        if full_lists:
            assert np.array_equal(idx[r] - i0, order[:K]), r
    assert covered >= min_cover * total, (covered, total)


@pytest.mark.parametrize("R,C,I,T,K,i0,i1", [(512, 128, 20001, 101, 100, 0, 20001), (70, 64, 5000, 20, 10, 0, 5000),
                                              (200, 256, 40000, 201, 100, 0, 40000), (130, 128, 20001, 101, 100, 2504, 12008),
                                              (64, 128, 9000, 31, 128, 0, 9000)])
def test_fused_eval_scoring_matches_the_reference(R, C, I, T, K, i0, i1):
    o = ops()
    rows, table, bias, seen = _problem(R, C, I, T, seed=R + I)
    assert o.EVAL_FUSED
    val, idx = o.score_topk(rows, table, bias, seen, K, i0, i1)
    torch.cuda.synchronize()
    _check(val, idx, _reference(rows, table, bias, seen, i0, i1, K), seen, i0, K, full_lists=(K == 10))
    # the unfused kernels on the same operands: same lists wherever f32 rounding cannot reorder neighbours (checked above against
    # the reference for both), same values to rounding
    o.EVAL_FUSED = False
    try:
        val2, idx2 = o.score_topk(rows, table, bias, seen, K, i0, i1)
    finally:
        o.EVAL_FUSED = True
    _check(val2, idx2, _reference(rows, table, bias, seen, i0, i1, K), seen, i0, K, full_lists=(K == 10))
    same = (idx == idx2).float().mean().item()
    assert same > 0.98, same


def test_fused_eval_at_the_full_size_of_config_3():
    """BASELINE.json configs[2] as the evaluation scoring sees it: 512 rows x 1 000 001 items x 256 channels, 201 seen ids per row, K = 100 —
    membership, values and order (wherever the fp64 gaps to both neighbours exceed what f32 accumulation can reorder) of ALL rows against
    an fp64 reference computed on the GPU (4 GB of logits; torch.topk — no ties in random data)."""
    o = ops()
    R, C, I, T, K = 512, 256, 1_000_001, 201, 100
    g = torch.Generator(device="cuda").manual_seed(77)
    rows = (torch.randn(R, C, device="cuda", generator=g) * 0.5).bfloat16()
    table = (torch.randn(I, C, device="cuda", generator=g) * 0.3).bfloat16()
    bias = torch.randn(I - 1, device="cuda", generator=g) * 0.2
    seen = torch.randint(0, I, (R, T), device="cuda", generator=g)
    seen[:, 0] = 0
    seen[:, 1] = I - 1
    assert o.EVAL_FUSED
    val, idx = o.score_topk(rows, table, bias, seen, K, 0, I)
    torch.cuda.synchronize()
    t = table.double()
    t[0] = 0.0
    lg = rows.double() @ t.T
    del t
    lg += torch.cat([torch.full((1,), -1000.0, dtype=torch.float64, device="cuda"), bias.double()])
    lg.scatter_(1, seen, float("-inf"))
    ref_val, ref_idx = torch.topk(lg, K + 1, dim=1)
    idx = idx.long()
    got = lg.gather(1, idx)
    assert bool(torch.isfinite(got).all())                                                   # no seen id
    assert bool(((got - val.double()).abs() <= 2e-4 * got.abs().max().clamp(min=1.0)).all())  # the values ARE those items' logits
    assert bool((got >= ref_val[:, K - 1:K] - 2e-4).all())                                   # each one belongs to the top K
    assert bool((val[:, 1:] <= val[:, :-1]).all())                                           # descending
    assert bool((torch.sort(idx, dim=1).values.diff(dim=1) != 0).all())                      # K different items
    gaps = ref_val[:, :-1] - ref_val[:, 1:]                                                  # K gaps: gaps[p] between positions p and p + 1
    fixed = gaps > ORDER_GAP
    fixed[:, 1:] &= gaps[:, :-1] > ORDER_GAP
    assert bool((idx[fixed] == ref_idx[:, :K][fixed]).all())
    assert float(fixed.float().mean()) > 0.9, float(fixed.float().mean())


def test_fused_eval_scoring_breaks_ties_by_index_and_survives_overflow():
    """Blocks of identical items: (a) pairs of equal logits inside the top K come out lower index first; (b) a table of ONE repeated
    row makes every logit of a row equal — every element passes the bound, the candidate lists overflow and the exact kernel
    returns the K lowest unseen indices."""
    o = ops()
    from easydgl_amd import _lib
    R, C, I, T, K = 96, 128, 8192, 40, 100
    rows, table, bias, seen = _problem(R, C, I, T, seed=3, dup=2000)
    val, idx = o.score_topk(rows, table, bias, seen, K, 0, I)
    lg = _reference(rows, table, bias, seen, 0, I, K)
    _check(val, idx, lg, seen, 0, K, min_cover=0.0)          # (tie blocks: exact ties are checked below, index order)
    v, ix = val.cpu().numpy(), idx.cpu().numpy()
    for r in range(R):
        eq = np.where(v[r][1:] == v[r][:-1])[0]
        assert np.all(ix[r][eq + 1] > ix[r][eq]), r          # equal values: rising index
    assert sum(len(np.where(v[r][1:] == v[r][:-1])[0]) for r in range(R)) > R      # (the tie blocks are really there)
    # (b)
    table2 = table[5:6].repeat(I, 1).contiguous()
    bias2 = torch.zeros_like(bias)
    val, idx = o.score_topk(rows, table2, bias2, seen, K, 0, I)
    ix, s = idx.cpu().numpy(), seen.cpu().numpy()
    for r in range(R):
        lgr = float((rows[r].double() * table2[1].double()).sum())
        if lgr > -1000.0:
            want = [i for i in range(1, 3 * K) if i not in set(s[r].tolist())][:K]
        else:
            continue
        assert ix[r].tolist() == want, r
    # the workspace's counts say the exact kernel was in charge
    assert int(_lib.lib.edgl_score_topk_fused_supported(R, C, I, T, K, _lib.BF16)) == 1


def test_fused_eval_is_refused_for_shapes_it_does_not_take():
    o = ops()
    from easydgl_amd import _lib
    lib = _lib.lib
    assert lib.edgl_score_topk_fused_supported(512, 128, 3000, 101, 100, _lib.BF16) == 0      # too few items for K + T groups
    assert lib.edgl_score_topk_fused_supported(512, 512, 20001, 101, 100, _lib.BF16) == 0     # width
    assert lib.edgl_score_topk_fused_supported(512, 128, 20001, 101, 100, _lib.F32) == 0
    assert lib.edgl_score_topk_fused_supported(512, 128, 20001, 101, 100, _lib.BF16) == 1
    rows, table, bias, seen = _problem(8, 128, 3000, 11, seed=1)
    out_v = torch.empty((8, 10), device="cuda"); out_i = torch.empty((8, 10), device="cuda", dtype=torch.int32)
    ws = torch.empty(1 << 20, device="cuda", dtype=torch.uint8)
    rc = lib.edgl_score_topk_fused(rows.data_ptr(), table.data_ptr(), bias.data_ptr(), seen.data_ptr(), 11, 8, 128, 3000, 0, 3000, 10,
                                   out_v.data_ptr(), out_i.data_ptr(), ws.data_ptr(), _lib.BF16, None)
    assert rc < 0
    val, idx = o.score_topk(rows, table, bias, seen, 10, 0, 3000)       # the op runs the unfused kernels there
    _check(val, idx, _reference(rows, table, bias, seen, 0, 3000, 10), seen, 0, 10)


def test_fused_eval_scratch_is_bounded_by_row_blocks(monkeypatch):
    """The exact-fallback scratch of the fused plan is [rows, items] f32: ops.score_topk walks the rows in blocks so that it never
    exceeds EVAL_FUSED_SCRATCH_BYTES, with the same lists as the one-call form; the workspace is cached per stream (no allocation
    per call); a seen-id view that is not 16-byte aligned / contiguous is re-laid, not read through its raw pointer."""
    o = ops()
    R, C, I, T, K = 300, 128, 20001, 101, 100
    rows, table, bias, seen = _problem(R, C, I, T, seed=77)
    val, idx = o.score_topk(rows, table, bias, seen, K, 0, I)
    n_ws = {k: v.numel() for k, v in o._FUSED_EVAL_WS.items()}
    monkeypatch.setattr(o, "EVAL_FUSED_SCRATCH_BYTES", 128 * I * 4)       # three row blocks: 128 + 128 + 44
    val2, idx2 = o.score_topk(rows, table, bias, seen, K, 0, I)
    assert torch.equal(idx, idx2) and torch.equal(val, val2)
    assert {k: v.numel() for k, v in o._FUSED_EVAL_WS.items()} == n_ws    # (the smaller plan fits the cached workspace)
    # an odd-offset, strided view of the seen ids
    big = torch.zeros((R, T + 1), dtype=torch.int64, device="cuda")
    big[:, 1:] = seen
    view = big[:, 1:]
    assert not view.is_contiguous()
    val3, idx3 = o.score_topk(rows, table, bias, view, K, 0, I)
    assert torch.equal(idx, idx3) and torch.equal(val, val3)
    flat = torch.zeros(R * T + 1, dtype=torch.int64, device="cuda")
    flat[1:] = seen.reshape(-1)
    off = flat[1:].view(R, T)                                            # contiguous, base 8 bytes off a 16-byte boundary
    assert off.is_contiguous() and off.data_ptr() % 16 == 8
    val4, idx4 = o.score_topk(rows, table, bias, off, K, 0, I)
    assert torch.equal(idx, idx4) and torch.equal(val, val4)
    _check(val, idx, _reference(rows, table, bias, seen, 0, I, K), seen, 0, K)

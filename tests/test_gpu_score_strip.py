"""K5 strip kernels (csrc/k_score_strip.hip: bf16, C = 128; csrc/k_score_stripw.hip: C = 256; one wave per SIMD) behind edgl_score_flash_fwd_coef / _bwd, against an
fp64 restatement of EasyDGL.py:149-155,177-185 (Appendix C of SURVEY.md) computed by torch on the same bf16 operands — at sizes up
to the benchmarked one (R_w ~ 5.4 K weighted rows x 20001 items = 12 item chunks x 21 row blocks), with the label pile-up of the
Zipf recipe, and with logit spreads that force the exact-maximum fallback of the row reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ops():
    from easydgl_amd import ops
    return ops


def _reference(rows, tab, bias, labels):
    """fp64: lse, label logits, coef, d_rows, d_table, d_bias for the weighted rows (label != 0)."""
    R, C = rows.shape
    I = tab.shape[0]
    x = rows.double()
    t = tab.double().clone()
    t[0] = 0.0                                                   # zero-padded table (coding.py:56-57)
    b = torch.cat([torch.full((1,), -1000.0, dtype=torch.float64, device=rows.device), bias.double()])   # Base.py:106-110
    logits = x @ t.T + b
    lse = torch.logsumexp(logits, dim=1)
    ll = logits.gather(1, labels.view(-1, 1)).squeeze(1)
    w = (labels != 0).double()
    n = w.sum()
    py = torch.exp(ll - lse)
    coef = w * (1.0 / (n + 1e-5)) * (py / (py + 1e-5))            # d loss / d logit scale (EasyDGL.py:155,177-185)
    p = torch.exp(logits - lse.view(-1, 1))
    dl = coef.view(-1, 1) * p
    dl[torch.arange(R, device=rows.device), labels] -= coef
    d_rows = dl @ t
    d_tab = dl.T @ x
    d_tab[0] = 0.0
    d_bias = dl[:, 1:].sum(0)
    return lse, ll, coef, d_rows, d_tab, d_bias


def _rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-300))


def _rel_max(a, b):
    a, b = a.double(), b.double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-300))


def _run_flash(rows, tab, bias, labels):
    from easydgl_amd._lib import check, lib
    o = _ops()
    R, C = rows.shape
    I = tab.shape[0]
    rows_c, lab_c, perm, _inv, nvalid = o.compact_rows(rows, labels)
    p, st, code = o._ptr, o._stream(), o._code(rows)
    wsf = torch.empty(lib.edgl_score_flash_workspace(R, C, I, I, code), device="cuda")
    lse = torch.empty(R, device="cuda"); ll = torch.zeros(R, device="cuda"); coef = torch.empty(R, device="cuda")
    check(lib.edgl_score_flash_fwd_coef(p(rows_c), p(tab), p(bias), p(lab_c), R, C, I, p(nvalid), p(lse), p(ll), p(coef), p(wsf), code, st),
          "edgl_score_flash_fwd_coef")
    d_rows = torch.empty_like(rows_c); d_tab = torch.empty((I, C), device="cuda"); d_b = torch.empty(I - 1, device="cuda")
    check(lib.edgl_score_flash_bwd(p(rows_c), p(tab), p(bias), p(lab_c), p(lse), p(coef), None, R, C, I, 0, I, p(nvalid), p(d_rows),
                                   p(d_tab), p(d_b), p(wsf), code, st), "edgl_score_flash_bwd")
    torch.cuda.synchronize()
    n = int(nvalid.item())
    return n, perm[:n].long(), lse[:n], ll[:n], coef[:n], d_rows[:n], d_tab, d_b


def _check(rows, tab, bias, labels, tol_rows=1.2e-2, tol_tab=6e-3):
    n, perm, lse, ll, coef, d_rows, d_tab, d_b = _run_flash(rows, tab, bias, labels)
    assert n == int((labels != 0).sum())
    r_lse, r_ll, r_coef, r_drows, r_dtab, r_db = _reference(rows[perm], tab, bias, labels[perm])
    assert _rel_max(lse, r_lse) < 2e-5, _rel_max(lse, r_lse)
    assert float((ll.double() - r_ll).abs().max()) < 1e-4 * (1.0 + float(r_ll.abs().max()))
    assert _rel_max(coef, r_coef) < 2e-3, _rel_max(coef, r_coef)
    # d_rows is stored in bf16 (half an ulp = 2e-3 relative per element) from P rounded to bf16; d_table / d_bias are f32 sums of
    # bf16-rounded P: per-tensor relative L2 AND max-norm
    e2, em = _rel_l2(d_rows.float(), r_drows), _rel_max(d_rows.float(), r_drows)
    assert e2 < tol_rows and em < 2 * tol_rows, ("d_rows", e2, em)
    e2, em = _rel_l2(d_tab, r_dtab), _rel_max(d_tab, r_dtab)
    assert e2 < tol_tab and em < 2 * tol_tab, ("d_table", e2, em)
    e2, em = _rel_l2(d_b, r_db), _rel_max(d_b, r_db)
    assert e2 < tol_tab and em < 2 * tol_tab, ("d_bias", e2, em)
    assert float(d_tab[0].abs().max()) == 0.0


def _problem(R, I, seed, hot=0.0, zero=0.3, scale_rows=0.6, scale_tab=0.4, C=128):
    g = torch.Generator(device="cuda").manual_seed(seed)
    rows = (torch.randn(R, C, device="cuda", generator=g) * scale_rows * (128 / C) ** 0.5).bfloat16()     # logits of the same spread at every width
    tab = (torch.randn(I, C, device="cuda", generator=g) * scale_tab).bfloat16()
    bias = torch.randn(I - 1, device="cuda", generator=g) * 0.3
    labels = torch.randint(1, I, (R,), device="cuda", generator=g)
    u = torch.rand(R, device="cuda", generator=g)
    labels[u < hot] = I - 2                                       # the clip pile-up of the Zipf recipe
    labels[u > 1.0 - zero] = 0
    return rows, tab, bias, labels


WIDTHS = [128, 256, 512]      # k_score_strip.hip / k_score_stripw.hip (256: stripw_kernel; 512: stripw5_kernel, two channel halves)


@pytest.mark.parametrize("C", WIDTHS)
@pytest.mark.parametrize("R,I,hot", [(70, 130, 0.0), (257, 2701, 0.2), (1000, 20001, 0.35), (640, 4099, 0.0), (31, 33, 0.0), (129, 161, 0.5)])
def test_strip_against_fp64(R, I, hot, C):
    _check(*_problem(R, I, seed=R + I, hot=hot, C=C))


@pytest.mark.parametrize("C", WIDTHS)
def test_strip_at_the_benchmarked_size(C):
    """B = 512, M = 20 -> 10240 masked slots of which ~52 % are weighted: 21 row blocks x 12 item chunks of the 20001-item table."""
    rows, tab, bias, labels = _problem(10240, 20001, seed=5, hot=0.35, zero=0.475, C=C)
    _check(rows, tab, bias, labels)


@pytest.mark.parametrize("C", WIDTHS)
def test_strip_all_rows_weighted(C):
    rows, tab, bias, labels = _problem(10240, 20001, seed=6, hot=0.0, zero=0.0, C=C)
    _check(rows, tab, bias, labels)


@pytest.mark.parametrize("C", [256, 512])
def test_wide_strip_on_a_table_of_many_chunks(C):
    """C = 256 / 512 against a 150 K-item table (config 3's width; the slab cap of the planner bounds the item chunks): every chunk count /
    remainder path of the device plan at a size the fp64 reference still holds (2000 x 150001 logits)."""
    rows, tab, bias, labels = _problem(2000, 150001, seed=8, hot=0.1, zero=0.3, C=C)
    _check(rows, tab, bias, labels)


def _reference_chunked(rows, tab, bias, labels, chunk=16384):
    """_reference over item chunks (two sweeps: log-sum-exp and label logits, then the three gradients) for tables whose [R, I] fp64
    logits do not fit: the same arithmetic, the same outputs."""
    R, C = rows.shape
    I = tab.shape[0]
    dev = rows.device
    x = rows.double()
    b = torch.cat([torch.full((1,), -1000.0, dtype=torch.float64, device=dev), bias.double()])

    def chunk_logits(i0, i1):
        t = tab[i0:i1].double()
        if i0 == 0:
            t[0] = 0.0
        return t, x @ t.T + b[i0:i1]

    m = torch.full((R,), -float("inf"), dtype=torch.float64, device=dev)
    s = torch.zeros(R, dtype=torch.float64, device=dev)
    ll = torch.zeros(R, dtype=torch.float64, device=dev)
    for i0 in range(0, I, chunk):
        i1 = min(I, i0 + chunk)
        _, lg = chunk_logits(i0, i1)
        mc = torch.maximum(m, lg.max(dim=1).values)
        s = s * torch.exp(m - mc) + torch.exp(lg - mc.view(-1, 1)).sum(1)
        m = mc
        inside = (labels >= i0) & (labels < i1)
        r = inside.nonzero().squeeze(1)
        ll[r] = lg[r, labels[r] - i0]
    lse = m + torch.log(s)
    w = (labels != 0).double()
    py = torch.exp(ll - lse)
    coef = w * (1.0 / (w.sum() + 1e-5)) * (py / (py + 1e-5))
    d_rows = torch.zeros(R, C, dtype=torch.float64, device=dev)
    d_tab = torch.empty(I, C, dtype=torch.float64, device=dev)
    d_b = torch.empty(I, dtype=torch.float64, device=dev)
    for i0 in range(0, I, chunk):
        i1 = min(I, i0 + chunk)
        t, lg = chunk_logits(i0, i1)
        dl = torch.exp(lg - lse.view(-1, 1)).mul_(coef.view(-1, 1))
        r = ((labels >= i0) & (labels < i1)).nonzero().squeeze(1)
        dl[r, labels[r] - i0] -= coef[r]
        d_rows += dl @ t
        d_tab[i0:i1] = dl.T @ x
        d_b[i0:i1] = dl.sum(0)
    d_tab[0] = 0.0
    return lse, ll, coef, d_rows, d_tab, d_b[1:]


def test_chunked_reference_is_the_plain_reference():
    rows, tab, bias, labels = _problem(300, 5001, seed=3, hot=0.2, zero=0.3, C=256)
    for a, c in zip(_reference(rows, tab, bias, labels), _reference_chunked(rows, tab, bias, labels, chunk=1000)):
        assert _rel_max(c, a) < 1e-12


def test_wide_strip_at_the_full_size_of_config_3():
    """BASELINE.json configs[2] as the scoring kernels see it: 512 x 40 masked slots (~ 52 % weighted) against the 1 000 001-row table at
    C = 256 — every weighted row's log-sum-exp / label logit / coefficient, d_rows, and ALL 256 M entries of d_table (+ d_bias) against the
    chunked fp64 reference on the same bf16 operands.  (10.7 K x 1 M fp64 logits, twice: ~ 20 s.)"""
    rows, tab, bias, labels = _problem(512 * 40, 1000001, seed=9, hot=0.05, zero=0.475, C=256)
    n, perm, lse, ll, coef, d_rows, d_tab, d_b = _run_flash(rows, tab, bias, labels)
    assert n == int((labels != 0).sum())
    r_lse, r_ll, r_coef, r_drows, r_dtab, r_db = _reference_chunked(rows[perm], tab, bias, labels[perm])
    assert _rel_max(lse, r_lse) < 2e-5, _rel_max(lse, r_lse)
    assert float((ll.double() - r_ll).abs().max()) < 1e-4 * (1.0 + float(r_ll.abs().max()))
    assert _rel_max(coef, r_coef) < 2e-3, _rel_max(coef, r_coef)
    for name, got, ref, tol in (("d_rows", d_rows.float(), r_drows, 1.2e-2), ("d_table", d_tab, r_dtab, 6e-3), ("d_bias", d_b, r_db, 6e-3)):
        e2, em = _rel_l2(got, ref), _rel_max(got, ref)
        assert e2 < tol and em < 2 * tol, (name, e2, em)
    assert float(d_tab[0].abs().max()) == 0.0


@pytest.mark.parametrize("C", WIDTHS)
def test_strip_reference_fallback_on_a_wide_logit_spread(C):
    """A few rows whose largest logit sits ~150 above the logits of the chunk's first unit: exp(logit - reference) overflows f32
    in the first attempt, the workgroup must redo its chunk with the exact row maxima — results as accurate as everywhere else."""
    rows, tab, bias, labels = _problem(600, 9001, seed=11, hot=0.0, zero=0.2, C=C)
    rows = rows.float(); tab = tab.float()
    for r, z in ((3, 4000), (200, 77), (599, 9000), (300, 8000)):
        v = rows[r] / rows[r].norm()
        tab[z] = v * (150.0 / float(rows[r].norm()))              # logit(r, z) ~ +150, the rest stay O(1)
    rows = rows.bfloat16(); tab = tab.bfloat16()
    labels[3] = 4000; labels[200] = 5; labels[599] = 0
    _check(rows, tab, bias, labels, tol_rows=2e-2, tol_tab=1.5e-2)


@pytest.mark.parametrize("C", WIDTHS)
def test_strip_label_scatter_with_every_row_on_one_label(C):
    rows, tab, bias, labels = _problem(900, 3001, seed=13, hot=1.0, zero=0.0, C=C)
    _check(rows, tab, bias, labels)


@pytest.mark.parametrize("dt,C", [(torch.bfloat16, 128), (torch.bfloat16, 64), (torch.float32, 128), (torch.bfloat16, 256), (torch.bfloat16, 512)])
def test_rows_finished_in_the_forward_call_match_the_two_call_sequence(dt, C):
    """edgl_score_flash_fwd_rows_w (lse, label logits, coefficients AND d_rows from one finishing launch) + edgl_score_flash_bwd with
    d_rows = NULL  ==  edgl_score_flash_fwd_coef + edgl_score_flash_bwd: same numbers (chunk sums and label logits are summed in another order:
    2e-6), widths with the fused kernel (64 / 128 / 256) and without (512: the two kernels behind the same entry point)."""
    from easydgl_amd._lib import check, lib
    o = _ops()
    R, I = 777, 3001
    g = torch.Generator(device="cuda").manual_seed(C + 1)
    rows = (torch.randn(R, C, device="cuda", generator=g) * 0.6 * (128 / C) ** 0.5).to(dt)
    tab = (torch.randn(I, C, device="cuda", generator=g) * 0.4).to(dt)
    bias = torch.randn(I - 1, device="cuda", generator=g) * 0.3
    labels = torch.randint(1, I, (R,), device="cuda", generator=g)
    labels[torch.rand(R, device="cuda", generator=g) > 0.6] = 0
    rows_c, lab_c, perm, _inv, nvalid = o.compact_rows(rows, labels)
    p, st, code = o._ptr, o._stream(), o._code(rows)
    ws = torch.empty(lib.edgl_score_flash_workspace(R, C, I, I, code), device="cuda")
    wtot = torch.tensor([1234], device="cuda", dtype=torch.int32)
    gs = torch.tensor([0.7], device="cuda")

    def outs():
        return (torch.empty(R, device="cuda"), torch.zeros(R, device="cuda"), torch.empty(R, device="cuda"), torch.empty_like(rows_c),
                torch.empty((I, C), device="cuda"), torch.empty(I - 1, device="cuda"))
    lse0, ll0, cf0, dr0, dt0, db0 = outs()
    check(lib.edgl_score_flash_fwd_coef_w(p(rows_c), p(tab), p(bias), p(lab_c), R, C, I, p(nvalid), p(wtot), p(lse0), p(ll0), p(cf0), p(ws),
                                          code, st), "fwd_coef_w")
    check(lib.edgl_score_flash_bwd(p(rows_c), p(tab), p(bias), p(lab_c), p(lse0), p(cf0), p(gs), R, C, I, 0, I, p(nvalid), p(dr0), p(dt0),
                                   p(db0), p(ws), code, st), "bwd")
    lse1, ll1, cf1, dr1, dt1, db1 = outs()
    check(lib.edgl_score_flash_fwd_rows_w(p(rows_c), p(tab), p(bias), p(lab_c), R, C, I, p(nvalid), p(wtot), p(gs), p(lse1), p(ll1), p(cf1),
                                          p(dr1), p(ws), code, st), "fwd_rows_w")
    check(lib.edgl_score_flash_bwd(p(rows_c), p(tab), p(bias), p(lab_c), p(lse1), p(cf1), p(gs), R, C, I, 0, I, p(nvalid), None, p(dt1),
                                   p(db1), p(ws), code, st), "bwd (table side only)")
    torch.cuda.synchronize()
    assert float((lse0 - lse1).abs().max()) <= 2e-6 * (1 + float(lse0.abs().max()))      # chunk sums in another order
    assert float((ll0 - ll1).abs().max()) <= 2e-6 * (1 + float(ll0.abs().max()))
    assert _rel_max(cf1, cf0) < 1e-5
    assert _rel_max(dr1.float(), dr0.float()) < (1e-5 if dt == torch.float32 else 1e-2)     # bf16 outputs: an ulp where the coefficient moved
    assert _rel_max(dt1, dt0) < 1e-4 and _rel_max(db1, db0) < 1e-4


@pytest.mark.parametrize("C", WIDTHS)
def test_table_pass_over_item_ranges_adds_up_to_the_full_pass(C):
    """edgl_score_flash_bwd over [0, h) and [h, I) (the item shards of SURVEY §8e: i0 a multiple of 8) with the GLOBAL log-sum-exp and
    coefficients of a full-range forward: each call writes its rows of d_table / d_bias and nothing else, together they are the full pass."""
    from easydgl_amd._lib import check, lib
    o = _ops()
    R, I, h = 700, 5001, 2504
    rows, tab, bias, labels = _problem(R, I, seed=21 + C, hot=0.2, zero=0.3, C=C)
    rows_c, lab_c, perm, _inv, nvalid = o.compact_rows(rows, labels)
    p, st, code = o._ptr, o._stream(), o._code(rows)
    ws = torch.empty(lib.edgl_score_flash_workspace(R, C, I, I, code), device="cuda")
    lse = torch.empty(R, device="cuda"); ll = torch.zeros(R, device="cuda"); coef = torch.empty(R, device="cuda")
    check(lib.edgl_score_flash_fwd_coef(p(rows_c), p(tab), p(bias), p(lab_c), R, C, I, p(nvalid), p(lse), p(ll), p(coef), p(ws), code, st), "fwd_coef")
    full_t = torch.empty((I, C), device="cuda"); full_b = torch.empty(I - 1, device="cuda")
    check(lib.edgl_score_flash_bwd(p(rows_c), p(tab), p(bias), p(lab_c), p(lse), p(coef), None, R, C, I, 0, I, p(nvalid), None, p(full_t), p(full_b),
                                   p(ws), code, st), "bwd full")
    part_t = torch.full((I, C), float("nan"), device="cuda"); part_b = torch.full((I - 1,), float("nan"), device="cuda")
    check(lib.edgl_score_flash_bwd(p(rows_c), p(tab), p(bias), p(lab_c), p(lse), p(coef), None, R, C, I, 0, h, p(nvalid), None, p(part_t), p(part_b),
                                   p(ws), code, st), "bwd [0, h)")
    torch.cuda.synchronize()
    assert bool(torch.isnan(part_t[h:]).all()) and bool(torch.isnan(part_b[h - 1:]).all())      # the other shard's rows: untouched
    assert not bool(torch.isnan(part_t[:h]).any()) and not bool(torch.isnan(part_b[:h - 1]).any())
    check(lib.edgl_score_flash_bwd(p(rows_c), p(tab), p(bias), p(lab_c), p(lse), p(coef), None, R, C, I, h, I, p(nvalid), None, p(part_t), p(part_b),
                                   p(ws), code, st), "bwd [h, I)")
    torch.cuda.synchronize()
    # (the row chunks of a range are summed in another order than those of the full pass)
    assert _rel_max(part_t, full_t) < 2e-5, _rel_max(part_t, full_t)
    assert _rel_max(part_b, full_b) < 2e-5, _rel_max(part_b, full_b)


@pytest.mark.parametrize("C", WIDTHS)
def test_no_weighted_row_at_all(C):
    """Every label 0 (a data-parallel rank whose half of the batch carries no weight): the passes run over zero rows — no NaN, the
    table / bias gradients are exactly zero, the row outputs are never read."""
    from easydgl_amd._lib import check, lib
    o = _ops()
    R, I = 300, 1201
    rows, tab, bias, labels = _problem(R, I, seed=3 + C, C=C)
    labels.zero_()
    rows_c, lab_c, perm, _inv, nvalid = o.compact_rows(rows, labels)
    assert int(nvalid.item()) == 0
    p, st, code = o._ptr, o._stream(), o._code(rows)
    ws = torch.empty(lib.edgl_score_flash_workspace(R, C, I, I, code), device="cuda")
    lse = torch.zeros(R, device="cuda"); ll = torch.zeros(R, device="cuda"); coef = torch.zeros(R, device="cuda")
    check(lib.edgl_score_flash_fwd_coef(p(rows_c), p(tab), p(bias), p(lab_c), R, C, I, p(nvalid), p(lse), p(ll), p(coef), p(ws), code, st), "fwd_coef")
    d_rows = torch.zeros_like(rows_c); d_tab = torch.full((I, C), float("nan"), device="cuda"); d_b = torch.full((I - 1,), float("nan"), device="cuda")
    check(lib.edgl_score_flash_bwd(p(rows_c), p(tab), p(bias), p(lab_c), p(lse), p(coef), None, R, C, I, 0, I, p(nvalid), p(d_rows), p(d_tab), p(d_b),
                                   p(ws), code, st), "bwd")
    torch.cuda.synchronize()
    assert float(d_tab.abs().max()) == 0.0 and float(d_b.abs().max()) == 0.0


@pytest.mark.parametrize("C", WIDTHS)
@pytest.mark.parametrize("i0,i1", [(0, 2504), (2504, 5001), (1000, 1136)])
def test_row_pass_over_an_item_range(C, i0, i1):
    """edgl_score_flash_fwd over [i0, i1) (an item shard: i0 a multiple of 8): the log-sum-exp of the RANGE's logits — the pad item's
    -1000 only where the range holds item 0, nothing of the items outside — against fp64 on the same bf16 operands."""
    from easydgl_amd._lib import check, lib
    o = _ops()
    R, I = 500, 5001
    rows, tab, bias, labels = _problem(R, I, seed=31 + C + i0, hot=0.1, zero=0.25, C=C)
    rows_c, lab_c, perm, _inv, nvalid = o.compact_rows(rows, labels)
    n = int(nvalid.item())
    p, st, code = o._ptr, o._stream(), o._code(rows)
    ws = torch.empty(lib.edgl_score_flash_workspace(R, C, I, I, code), device="cuda")
    lse = torch.empty(R, device="cuda"); ll = torch.zeros(R, device="cuda")
    check(lib.edgl_score_flash_fwd(p(rows_c), p(tab), p(bias), p(lab_c), R, C, I, i0, i1, p(nvalid), p(lse), p(ll), p(ws), code, st), "flash_fwd")
    torch.cuda.synchronize()
    x = rows_c[:n].double()
    t = tab.double().clone(); t[0] = 0.0
    b = torch.cat([torch.full((1,), -1000.0, dtype=torch.float64, device="cuda"), bias.double()])
    logits = (x @ t.T + b)[:, i0:i1]
    want = torch.logsumexp(logits, dim=1)
    assert _rel_max(lse[:n], want) < 2e-5, _rel_max(lse[:n], want)

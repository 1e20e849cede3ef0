"""Random shapes through the strip scoring passes (k_score_strip.hip C = 128, k_score_stripw.hip C = 256 / 512) against the fp64 reference
of tests/test_gpu_score_strip.py — on the GPU box:   python tools/fuzz_strip.py [cases] [seed] [wide]
Draws R, I, the width, the share of unweighted rows, label pile-ups, operand scales (up to logit spreads that force the exact-maximum
fallback), and — for 40 % of the cases — the vocab-parallel protocol over 2 / 3 / 5 / 8 item ranges in one process; prints one line per failure and a summary.  Exit code 1 if anything failed."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_score_strip as T   # noqa: E402


WIDE = len(sys.argv) > 3 and sys.argv[3] == "wide"


def draw(rng):
    C = int(rng.choice([128, 256, 512]))
    kind = rng.integers(0, 6)
    if kind == 0:      # tiny
        R, I = int(rng.integers(1, 70)), int(rng.integers(2, 200))
    elif kind == 1:    # around the unit / block edges
        R = int(rng.choice([31, 32, 33, 127, 128, 129, 255, 256, 257, 511, 513])) + int(rng.integers(0, 2))
        I = int(rng.choice([32, 33, 63, 64, 65, 127, 129, 255, 257, 1023, 1025, 4095, 4097])) + int(rng.integers(0, 2))
    elif kind == 2:    # many items, few rows
        R, I = int(rng.integers(1, 300)), int(rng.integers(20000, 120000))
    elif kind == 3:    # many rows, few items
        R, I = int(rng.integers(3000, 24000)), int(rng.integers(2, 3000))
    else:
        R, I = int(rng.integers(100, 6000)), int(rng.integers(100, 30000))
    zero = float(rng.choice([0.0, 0.3, 0.475, 0.9, 0.99]))
    hot = float(rng.choice([0.0, 0.0, 0.2, 0.6, 1.0]))
    # (logit spread = scale_rows x scale_tab x sqrt(128): beyond ~ 8 the softmax is one-hot, the loss's log(p + 1e-5) saturates and
    #  every gradient underflows in f32 — nothing to compare; the wide-spread regime is drawn as a FEW spiked logits instead)
    sr = float(rng.choice([0.05, 0.6, 0.6, 1.0]))
    stb = float(rng.choice([0.05, 0.4, 0.4, 0.7]))
    if sr * stb > 0.5:   # (1.0 x 0.7: logit spread 7.9 — the edge of that regime: label-term rounding at 1.05 x the bound on one draw in 400)
        stb = 0.4
    if WIDE:           # the saturated regime itself: finite outputs, errors against the natural scales (see check)
        sr, stb = float(rng.choice([1.5, 4.0])), float(rng.choice([1.0, 3.0]))
    spikes = int(rng.choice([0, 0, 0, 1, 4]))
    return R, I, C, zero, hot, sr, stb, spikes


def check(rows, tab, bias, labels, tol_rows, tol_tab):
    """T._check with every error measured against max(|reference|, the tensor's natural scale x FLOOR): when the only weighted rows are
    spiked ones, exp(label logit - lse) underflows, the coefficient and all gradients are 0 in f32 and ~ 1e-60 in the fp64 reference —
    a relative error of 1 that means nothing.  FLOOR = 1e-3, and 0.5 in the saturated regime (`wide`): there the label probabilities
    sit within 1e-2 of 1, the true gradients vanish, and what is left is the rounding of coef x P to bf16 IN FRONT of the label term
    (the strip passes subtract coef x one-hot in a separate f32 scatter): <= 2^-9 x coef x |operand row| per (row, label) — noise
    against the natural scale of an unsaturated gradient, but 100 % of a vanished one (stated tolerances)."""
    FLOOR = 0.5 if WIDE else 1e-3
    n, perm, lse, ll, coef, d_rows, d_tab, d_b = T._run_flash(rows, tab, bias, labels)
    assert n == int((labels != 0).sum())
    r_lse, r_ll, r_coef, r_drows, r_dtab, r_db = T._reference(rows[perm], tab, bias, labels[perm])
    for t in (lse, ll, coef, d_rows, d_tab, d_b):
        assert bool(torch.isfinite(t.float()).all()), "non-finite output"
    assert T._rel_max(lse, r_lse) < 2e-5, ("lse", T._rel_max(lse, r_lse))
    assert float((ll.double() - r_ll).abs().max()) < 1e-4 * (1.0 + float(r_ll.abs().max())), "label logit"
    cs = 1.0 / n                                  # natural scale of a coefficient
    xs, ts = float(rows.float().abs().max()), float(tab.float().abs().max())

    def err(got, ref, scale):
        got, ref = got.double(), ref.double()
        den_m = max(float(ref.abs().max()), FLOOR * scale)
        den_2 = max(float(ref.norm()), FLOOR * scale * ref.numel() ** 0.5)
        return float((got - ref).norm()) / den_2, float((got - ref).abs().max()) / den_m
    _, em = err(coef, r_coef, cs)
    assert em < 2e-3, ("coef", em)
    e2, em = err(d_rows.float(), r_drows, cs * ts)
    assert e2 < tol_rows and em < 2 * tol_rows, ("d_rows", e2, em)
    e2, em = err(d_tab, r_dtab, cs * xs)
    assert e2 < tol_tab and em < 2 * tol_tab, ("d_table", e2, em)
    e2, em = err(d_b, r_db, cs)
    assert e2 < tol_tab and em < 2 * tol_tab, ("d_bias", e2, em)
    assert float(d_tab[0].abs().max()) == 0.0


def check_sharded(rows, tab, bias, labels, nshard, tol_rows, tol_tab):
    """The vocab-parallel protocol (parallel.vocab_parallel_ce) in ONE process: the row pass over each of `nshard` item ranges
    (parallel.shard_bounds), the packed merge of (log-sum-exp, label logit), the coefficients, then the finishing + table pass of every
    range with the GLOBAL log-sum-exp — d_rows summed over the ranges, d_table / d_bias assembled from the ranges — against the
    unsharded fp64 reference."""
    from easydgl_amd import parallel
    from easydgl_amd._lib import check as ck, lib
    o = T._ops()
    R, C = rows.shape
    I = tab.shape[0]
    rows_c, lab_c, perm, _inv, nvalid = o.compact_rows(rows, labels)
    n = int(nvalid.item())
    p, st, code = o._ptr, o._stream(), o._code(rows)
    bounds = [parallel.shard_bounds(I, nshard, r) for r in range(nshard)]
    wss, lses, labs = [], [], []
    for i0, i1 in bounds:
        ws = torch.empty(int(lib.edgl_score_flash_workspace(R, C, I, i1 - i0, code)), device="cuda")
        lse = torch.empty(R, device="cuda"); lab = torch.zeros(R, device="cuda")
        ck(lib.edgl_score_flash_fwd(p(rows_c), p(tab), p(bias), p(lab_c), R, C, I, i0, i1, p(nvalid), p(lse), p(lab), p(ws), code, st), "flash_fwd")
        own = (lab_c >= max(i0, 1)) & (lab_c < i1)
        live = torch.arange(R, device="cuda") < n
        wss.append(ws); lses.append(torch.where(live, lse, torch.zeros_like(lse))); labs.append(torch.where(own, lab, torch.full_like(lab, float("-inf"))))
    g = torch.stack(lses)
    m = g.max(0).values
    lse = m + torch.log(torch.exp(g - m).sum(0))
    lab = torch.stack(labs).max(0).values
    w = (lab_c != 0).float()
    W = w.sum() + 1e-5
    py = torch.exp(lab - lse)
    coef = ((w / W) * py / (py + 1e-5)).contiguous()
    d_rows = torch.zeros(R, C, device="cuda")
    d_tab = torch.full((I, C), float("nan"), device="cuda"); d_b = torch.full((I - 1,), float("nan"), device="cuda")
    for (i0, i1), ws in zip(bounds, wss):
        dr = torch.empty_like(rows_c)
        ck(lib.edgl_score_flash_bwd(p(rows_c), p(tab), p(bias), p(lab_c), p(lse.contiguous()), p(coef), None, R, C, I, i0, i1, p(nvalid), p(dr), p(d_tab),
                                    p(d_b), p(ws), code, st), "flash_bwd")
        d_rows += dr.float()
    torch.cuda.synchronize()
    pm = perm[:n].long()
    r_lse, r_ll, r_coef, r_drows, r_dtab, r_db = T._reference(rows[pm], tab, bias, labels[pm])
    assert bool(torch.isfinite(d_tab).all()) and bool(torch.isfinite(d_b).all()), "a shard left rows of d_table / d_bias unwritten"
    assert T._rel_max(lse[:n], r_lse) < 2e-5, ("sharded lse", T._rel_max(lse[:n], r_lse))
    assert T._rel_max(coef[:n], r_coef) < 2e-3, ("sharded coef", T._rel_max(coef[:n], r_coef))
    for name, got, ref, tol in (("d_rows", d_rows[:n], r_drows, tol_rows), ("d_table", d_tab, r_dtab, tol_tab), ("d_bias", d_b, r_db, tol_tab)):
        e2, em = T._rel_l2(got, ref), T._rel_max(got, ref)
        assert e2 < tol and em < 2 * tol, ("sharded " + name, nshard, e2, em)


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = 0
    for k in range(cases):
        R, I, C, zero, hot, sr, stb, spikes = draw(rng)
        desc = f"case {k}: R={R} I={I} C={C} zero={zero} hot={hot} scale_rows={sr} scale_tab={stb} spikes={spikes}"
        try:
            rows, tab, bias, labels = T._problem(R, I, seed=1000 + k, hot=hot, zero=zero, scale_rows=sr, scale_tab=stb, C=C)
            if int((labels != 0).sum()) == 0:
                labels[0] = min(1, I - 1)
            if spikes and I > 8:   # a few logits ~ +150 (the exact-maximum fallback of the row reference), as in the fallback test
                rf, tf = rows.float(), tab.float()
                for _ in range(spikes):
                    r, z = int(rng.integers(0, R)), int(rng.integers(1, I))
                    nr = float(rf[r].norm())
                    if nr > 0:
                        tf[z] = rf[r] / nr * (150.0 / nr)
                rows, tab = rf.bfloat16(), tf.bfloat16()
            check(rows, tab, bias, labels, tol_rows=2e-2 if spikes else 1.2e-2, tol_tab=1.5e-2 if spikes else 6e-3)
            if not WIDE and not spikes and I >= 64 and rng.random() < 0.4:
                ns = int(rng.choice([2, 3, 5, 8]))
                desc += f" shards={ns}"
                check_sharded(rows, tab, bias, labels, ns, tol_rows=1.5e-2, tol_tab=6e-3)
        except AssertionError as e:
            bad += 1
            print("FAIL", desc, "->", str(e)[:200], flush=True)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("ERROR", desc, "->", type(e).__name__, str(e)[:200], flush=True)
        if (k + 1) % 50 == 0:
            print(f"... {k + 1} cases, {bad} failures", flush=True)
    print(f"fuzz_strip: {cases} cases, {bad} failures (seed {seed})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

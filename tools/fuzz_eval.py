"""Random shapes through ops.score_topk (fused evaluation scoring + seen mask + top-K, csrc/k_eval_topk.hip, and the unfused kernels where the
fused form declines) against an fp64 reference computed on the GPU — on the GPU box:   python tools/fuzz_eval.py [cases] [seed]
Draws rows, width, catalogue size, seen-list length, K, item ranges (i0 a multiple of 8), logit scales and duplicated seen ids; checks
membership, values, descending order, no seen id, K different items, and the order wherever the fp64 gaps to both neighbours exceed
what f32 accumulation can reorder.  Exit code 1 if anything failed."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from easydgl_amd import ops as o   # noqa: E402

ORDER_GAP = 2e-4


def check(rows, table, bias, seen, K, i0, i1, nshard=0):
    if nshard:      # the sharded protocol in one process: a local top-K per item range (parallel.shard_bounds) + the merge kernel
        from easydgl_amd import parallel
        parts = [o.score_topk(rows, table, bias, seen, K, *parallel.shard_bounds(table.shape[0], nshard, r)) for r in range(nshard)]
        val, idx = o.topk_merge(torch.stack([p[0] for p in parts]).contiguous(), torch.stack([p[1] for p in parts]).contiguous())
    else:
        val, idx = o.score_topk(rows, table, bias, seen, K, i0, i1)
    torch.cuda.synchronize()
    if os.environ.get("FUZZ_SELFTEST") == "1" and K >= 2:      # the checker must notice a list with its first two entries swapped
        idx = idx.clone(); idx[:, [0, 1]] = idx[:, [1, 0]]
    I = table.shape[0]
    t = table.double()
    t[0] = 0.0
    lg = rows.double() @ t.T + torch.cat([torch.full((1,), -1000.0, dtype=torch.float64, device="cuda"), bias.double()])
    lg.scatter_(1, seen, float("-inf"))
    lg = lg[:, i0:i1]
    n_ok = int(torch.isfinite(lg).sum(1).min())
    assert n_ok >= K, "degenerate case (fewer unseen items than K)"
    ref_val, ref_idx = torch.topk(lg, min(K + 1, lg.shape[1]), dim=1)
    idx = idx.long() - i0
    assert int(idx.min()) >= 0 and int(idx.max()) < i1 - i0, "index outside the range"
    got = lg.gather(1, idx)
    assert bool(torch.isfinite(got).all()), "seen id in the list"
    assert bool(((got - val.double()).abs() <= 2e-4 * got.abs().max().clamp(min=1.0)).all()), "values are not those items' logits"
    assert bool((got >= ref_val[:, K - 1:K] - 2e-4 * ref_val.abs().max().clamp(min=1.0)).all()), "an item outside the top K"
    assert bool((val[:, 1:] <= val[:, :-1]).all()), "not descending"
    assert bool((torch.sort(idx, dim=1).values.diff(dim=1) != 0).all()), "repeated item"
    if ref_val.shape[1] == K + 1:
        gaps = ref_val[:, :-1] - ref_val[:, 1:]
        fixed = gaps > ORDER_GAP * ref_val.abs().max().clamp(min=1.0)
        fixed[:, 1:] &= fixed[:, :-1].clone()
        assert bool((idx[fixed] == ref_idx[:, :K][fixed]).all()), "order differs where the gaps are resolvable"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = ran = 0
    for k in range(cases):
        C = int(rng.choice([64, 128, 256]))
        R = int(rng.choice([1, 7, 64, 130, 512, 700]))
        kind = int(rng.integers(0, 4))
        I = [int(rng.integers(300, 4000)), int(rng.integers(4096, 30000)), int(rng.integers(30000, 300000)), int(rng.choice([4096, 8192, 262144 + 8, 20001]))][kind]
        T = int(rng.choice([2, 20, 101, 201]))
        K = int(rng.choice([1, 10, 50, 100, 128]))
        if rng.random() < 0.5:
            i0, i1 = 0, I
        else:
            i0 = int(rng.integers(0, max(1, I // 2))) // 8 * 8
            i1 = int(rng.integers(min(I, i0 + K + T + 64), I + 1))
        if i1 - i0 < K + T + 8:
            continue
        sr, stb = float(rng.choice([0.1, 0.5, 1.5])), float(rng.choice([0.05, 0.3, 1.0]))
        desc = f"case {k}: R={R} C={C} I={I} T={T} K={K} range=[{i0},{i1}) scales={sr}/{stb}"
        try:
            g = torch.Generator(device="cuda").manual_seed(5000 + k)
            rows = (torch.randn(R, C, device="cuda", generator=g) * sr).bfloat16()
            table = (torch.randn(I, C, device="cuda", generator=g) * stb).bfloat16()
            bias = torch.randn(I - 1, device="cuda", generator=g) * 0.2
            seen = torch.randint(0, I, (R, T), device="cuda", generator=g)
            seen[:, 0] = 0
            seen[:, 1] = I - 1
            if T > 4 and rng.random() < 0.5:
                seen[:, 3] = seen[:, 2]            # a repeated seen id
            if T > 3 and rng.random() < 0.3:                   # seen ids that ARE the best items of the row (they must not come back)
                with torch.no_grad():
                    lg = rows.float() @ table.float().T
                    top = lg.topk(min(T - 2, 8), dim=1).indices
                    seen[:, 2:2 + top.shape[1]] = top
            ns = int(rng.choice([2, 3, 8])) if (i0 == 0 and i1 == I and I // 8 >= K + T + 64 and rng.random() < 0.5) else 0
            desc += f" shards={ns}" if ns else ""
            check(rows, table, bias, seen, K, i0, i1, ns)
            ran += 1
        except AssertionError as e:
            if "degenerate" in str(e):
                continue
            bad += 1
            print("FAIL", desc, "->", str(e)[:200], flush=True)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("ERROR", desc, "->", type(e).__name__, str(e)[:300], flush=True)
        if (k + 1) % 50 == 0:
            print(f"... {k + 1} cases, {bad} failures", flush=True)
    print(f"fuzz_eval: {cases} draws, {ran} checked, {bad} failures (seed {seed})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

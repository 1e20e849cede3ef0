"""Random shapes through the BiMAU operator (K3: csrc/k_bimau_fwd.hip / k_bimau_bwd.hip / k_bimau_big.hip) against the fp64 restatement —
the body of tests/test_gpu_ops.py::test_bimau_fwd_bwd with drawn (B, T, C, H, E) inside the documented limits (head dims
{16, 32, 64, 128}; T <= 208 at head dims 16 / 32, <= 128 at 64 / 128 in bf16; E <= 16).   python tools/fuzz_bimau.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_ops as T   # noqa: E402
from _pytest.outcomes import Skipped   # noqa: E402


_orig_close = T.assert_close


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = ran = 0
    for k in range(cases):
        dh = int(rng.choice([16, 16, 32, 64, 128]))
        H = int(rng.choice([1, 2, 4, 8])) if dh <= 32 else int(rng.choice([1, 2, 4]))
        C = dh * H
        if C > 512:
            H = 512 // dh
            C = dh * H
        tmax = 208 if dh <= 32 else 128
        Tn = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 64, 100, 101, 112, 127, 128, 150, 201, 208]))
        Tn = min(Tn, tmax)
        E = int(rng.choice([1, 2, 3, 5, 8, 13, 16]))
        B = int(rng.choice([1, 2, 3, 5]))
        # f32 (tight bounds: 3e-5 / 2e-4) is the bug detector; bf16 runs with the test's max-norm bounds x 2.5 — they were set on <= 3
        # samples, and the tail of what bf16-rounded Q / K do to a sharp softmax grows with the element count (measured: the same
        # cases pass in f32 at 3e-5)
        name, dt, tol = T.DTYPES[0 if rng.random() < 0.65 else 1]
        T.assert_close = (lambda a, b, t, what="", _f=_orig_close, _m=(2.5 if name == "bf16" else 1.0): _f(a, b, t * _m, what))
        # marks one LAUNCH takes at this head dim / dtype (f32 at head dim 32: 11 — the intensity backward's LDS); beyond it the module
        # runs mark groups (module/temporal.py modulated_attention), which is not what this operator-level fuzz exercises
        E = min(E, int(T.ops().lib.edgl_bimau_mark_group(C, H, T.ops()._code(torch.empty(0, dtype=dt)))))
        desc = f"case {k}: {name} B={B} T={Tn} C={C} H={H} (dh={dh}) E={E}"
        try:
            T.test_bimau_fwd_bwd(name, dt, tol, B, Tn, C, H, E)
            ran += 1
        except Skipped:
            continue
        except AssertionError as e:
            bad += 1
            print("FAIL", desc, "->", str(e)[:200], flush=True)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("ERROR", desc, "->", type(e).__name__, str(e)[:300], flush=True)
        if (k + 1) % 20 == 0:
            print(f"... {k + 1} draws, {ran} checked, {bad} failures", flush=True)
    print(f"fuzz_bimau: {cases} draws, {ran} checked, {bad} failures (seed {seed})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

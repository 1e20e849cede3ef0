"""Random shapes through the GEMM entry point (ops.gemm: csrc/k_gemm.hip generic kernels, csrc/k_gemm2.hip register strips / weights-streamed /
128 x 128 tiles, every fused epilogue) and the dense layer's backward (LinearFn: dX + the TN weight-gradient product with its split slabs)
against fp64 — bodies of tests/test_gpu_ops.py::test_gemm_layouts / ::test_gemm_bf16_activation_kernels plus a LinearFn check.
   python tools/fuzz_gemm.py [cases] [seed]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import test_gpu_ops as T   # noqa: E402


def linear_case(rng, dt):
    o = T.ops()
    M = int(rng.choice([7, 64, 300, 1000, 4100, 5200, 12000]))
    K = int(rng.choice([32, 64, 128, 256, 384, 512, 1024, 1536]))
    N = int(rng.choice([32, 64, 128, 256, 384, 512, 1024, 2048]))
    g = torch.Generator(device="cuda").manual_seed(int(rng.integers(0, 1 << 30)))
    x = (torch.randn(M, K, device="cuda", generator=g)).to(dt).requires_grad_()
    W = (torch.randn(K, N, device="cuda", generator=g) * 0.05).requires_grad_()
    b = (torch.randn(N, device="cuda", generator=g) * 0.1).requires_grad_()
    Wc = W.detach().to(dt)
    y = o.LinearFn.apply(x, W, b, Wc, False)
    G = torch.randn(M, N, device="cuda", generator=g).to(dt)
    y.backward(G)
    xr, Wr, br = x.detach().double().requires_grad_(), Wc.double().requires_grad_(), b.detach().double().requires_grad_()
    yr = xr @ Wr + br
    yr.backward(G.double())
    tol = 1e-5 if dt == torch.float32 else 1.2e-2
    n = lambda t: t.detach().double().cpu().numpy()
    T.assert_close(n(y), n(yr), tol, f"linear y M={M} K={K} N={N}")
    T.assert_close(n(x.grad), n(xr.grad), tol, f"linear dx M={M} K={K} N={N}")
    T.assert_close(n(W.grad), n(Wr.grad), 2e-5 if dt == torch.float32 else 1.2e-2, f"linear dW M={M} K={K} N={N}")
    T.assert_close(n(b.grad), n(br.grad), 2e-5 if dt == torch.float32 else 1.2e-2, f"linear db M={M} K={K} N={N}")
    return f"M={M} K={K} N={N}"


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 90
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed)
    bad = ran = 0
    for k in range(cases):
        kind = k % 3
        desc = ""
        try:
            if kind == 0:
                name, dt, tol = T.DTYPES[int(rng.integers(0, 2))]
                a_kc, b_kc = int(rng.integers(0, 2)), int(rng.integers(0, 2))
                M, N, K = int(rng.integers(1, 3000)), int(rng.integers(1, 160)) * 8, int(rng.integers(1, 160)) * 8
                if rng.random() < 0.3:      # odd extents: the generic kernels
                    N, K = N + int(rng.integers(-7, 8)), K + int(rng.integers(-7, 8))
                desc = f"layouts {name} a_kc={a_kc} b_kc={b_kc} M={M} N={N} K={K}"
                T.test_gemm_layouts(name, dt, tol, a_kc, b_kc, M, N, K)
            elif kind == 1:
                b_kc = int(rng.integers(0, 2))
                M = int(rng.choice([100, 300, 1030, 2100, 4100, 4352, 5000, 9000]))
                N = int(rng.choice([64, 128, 256, 384, 512, 1024, 1536, 2048]))
                K = int(rng.choice([32, 64, 128, 256, 384, 512, 1024, 1536, 2048]))
                flags = str(rng.choice(["bias", "plain", "gelu", "accum", "dgelu"]))
                desc = f"activation b_kc={b_kc} M={M} N={N} K={K} {flags}"
                T.test_gemm_bf16_activation_kernels(b_kc, M, N, K, flags)
            else:
                dt = torch.float32 if rng.random() < 0.4 else torch.bfloat16
                desc = f"linear {dt} "
                desc += linear_case(rng, dt)
            ran += 1
        except AssertionError as e:
            bad += 1
            print("FAIL", desc, "->", str(e)[:200], flush=True)
        except Exception as e:   # noqa: BLE001
            bad += 1
            print("ERROR", desc, "->", type(e).__name__, str(e)[:300], flush=True)
        if (k + 1) % 30 == 0:
            print(f"... {k + 1} draws, {ran} checked, {bad} failures", flush=True)
    print(f"fuzz_gemm: {cases} draws, {ran} checked, {bad} failures (seed {seed})")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()

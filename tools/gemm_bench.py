"""Times edgl_gemm / edgl_gemm_dw on the hot-path shapes (HIP events), optionally under EDGL_DBG ablations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easydgl_amd import ops, _lib

M = 51712
dt = torch.bfloat16
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for (K, N, name) in [(384, 512, "qkvt fwd"), (128, 128, "proj fwd"), (128, 256, "ffn1 fwd"), (256, 128, "ffn2 fwd")]:
    A = torch.randn(M, K, device="cuda", dtype=dt); W = torch.randn(K, N, device="cuda", dtype=dt) * 0.05
    b = torch.zeros(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=dt)
    t = timeit(lambda: ops.gemm(A, W, M, N, K, K, N, True, False, dt, bias=b, flags=_lib.EPI_BIAS, out=out))
    dz = torch.randn(M, N, device="cuda", dtype=dt); dx = torch.empty(M, K, device="cuda", dtype=dt)
    t2 = timeit(lambda: ops.gemm(dz, W, M, K, N, N, N, True, True, dt, out=dx))
    t3 = timeit(lambda: ops.dense_grads(A, dz, M, K, N))
    gf = 2.0 * M * K * N / 1e9
    print(f"{name:9s} K={K:3d} N={N:3d}  fwd {t:7.1f} us ({gf/t*1e3/1e3:6.1f} TF)  dX {t2:7.1f} us ({gf/t2:6.1f} TF)  dW {t3:7.1f} us ({gf/t3:6.1f} TF)")

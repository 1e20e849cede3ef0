"""Times edgl_gemm at the large projections (k_gemmw.hip against the 128 x 128 tiles: EDGL_GEMMW=0 | 1), forward (B = [K][N]) and dX
(B = [N][K]) with a check against torch on the same bf16 operands.
    python tools/gemmw_bench.py            # recipe (M = 15872) and config-3 (M = 102912) shapes"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from easydgl_amd import ops, _lib  # noqa: E402

dt = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for (M, K, N, name) in [(15872, 1536, 2048, "recipe qkvt"), (15872, 512, 512, "recipe proj"), (15872, 1024, 512, "recipe ffn2"),
                        (102912, 768, 1024, "config3 qkvt"), (102912, 512, 256, "config3 ffn2")]:
    A = torch.randn(M, K, device="cuda", dtype=dt); W = (torch.randn(K, N, device="cuda") * 0.05).to(dt)
    b = torch.randn(N, device="cuda"); out = torch.empty(M, N, device="cuda", dtype=dt)
    t = timeit(lambda: ops.gemm(A, W, M, N, K, K, N, True, False, dt, bias=b, flags=_lib.EPI_BIAS, out=out))
    ref = (A[:512].float() @ W.float() + b)
    e1 = float((out[:512].float() - ref).abs().max() / ref.abs().max())
    dz = torch.randn(M, N, device="cuda", dtype=dt); dx = torch.empty(M, K, device="cuda", dtype=dt)
    t2 = timeit(lambda: ops.gemm(dz, W, M, K, N, N, N, True, True, dt, out=dx))
    ref2 = dz[-300:].float() @ W.float().t()
    e2 = float((dx[-300:].float() - ref2).abs().max() / ref2.abs().max())
    gf = 2.0 * M * K * N / 1e9
    print(f"GEMMW={os.environ.get('EDGL_GEMMW', '1')} {name:13s} M={M} K={K} N={N}  fwd {t:7.1f} us ({gf / t:6.1f} TF/s, err {e1:.1e})   "
          f"dX {t2:7.1f} us ({gf / t2:6.1f} TF/s, err {e2:.1e})", flush=True)

"""Times the flash scoring pair (edgl_score_flash_fwd_coef + edgl_score_flash_bwd) at the headline shape — 10240 masked slots of
which ~52 % are weighted, C = 128, I = 20001, bf16 — with the strip kernels (default) or the round-2 kernels (EDGL_SCORE_STRIP=0):
    python tools/strip_bench.py            # wall times per call (HIP events on the launch stream)
    rocprofv3 --kernel-trace --stats -d gpurun_out/strip -- python tools/strip_bench.py     # per-kernel times"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from easydgl_amd import ops  # noqa: E402
from easydgl_amd._lib import check, lib  # noqa: E402
from easydgl_amd.ops import _ptr as p, _stream  # noqa: E402

R, C, I = (int(os.environ.get(k, d)) for k, d in (("R", "10240"), ("C", "128"), ("I", "20001")))      # R=20480 C=256 I=1000001: config 3
g = torch.Generator(device="cuda").manual_seed(1)
rows = (torch.randn(R, C, device="cuda", generator=g) * 0.6 * (128 / C) ** 0.5).bfloat16()
tab = (torch.randn(I, C, device="cuda", generator=g) * 0.4).bfloat16()
bias = torch.randn(I - 1, device="cuda", generator=g) * 0.3
labels = torch.randint(1, I, (R,), device="cuda", generator=g)
u = torch.rand(R, device="cuda", generator=g)
labels[u < 0.35] = I - 2
labels[u > 1.0 - float(os.environ.get("ZERO", "0.475"))] = 0
rows_c, lab_c, perm, inv, nvalid = ops.compact_rows(rows, labels)
n = int(nvalid.item())
code = ops._code(rows)
wsf = torch.empty(lib.edgl_score_flash_workspace(R, C, I, I, code), device="cuda")
lse = torch.empty(R, device="cuda"); ll = torch.zeros(R, device="cuda"); coef = torch.empty(R, device="cuda")
d_rows = torch.empty_like(rows_c); d_tab = torch.empty((I, C), device="cuda"); d_b = torch.empty(I - 1, device="cuda")


def fwd():
    check(lib.edgl_score_flash_fwd_coef(p(rows_c), p(tab), p(bias), p(lab_c), R, C, I, p(nvalid), p(lse), p(ll), p(coef), p(wsf), code,
                                        _stream()), "fwd")


def bwd():
    check(lib.edgl_score_flash_bwd(p(rows_c), p(tab), p(bias), p(lab_c), p(lse), p(coef), None, R, C, I, 0, I, p(nvalid), p(d_rows),
                                   p(d_tab), p(d_b), p(wsf), code, _stream()), "bwd")


def timeit(fn, n_=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n_):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n_ * 1e3


tf, tb = timeit(fwd), timeit(bwd)
gf = 4.0 * n * C * I / 1e9     # two products per pass
print(f"strip={os.environ.get('EDGL_SCORE_STRIP', '1')} weighted rows {n}: flash fwd {tf:7.1f} us ({gf / tf * 1e3:6.0f} TF/s incl. epilogue kernels)"
      f"   flash bwd {tb:7.1f} us ({gf / tb * 1e3:6.0f} TF/s incl. finish / reduce / scatter)")

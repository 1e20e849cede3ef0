"""Times the scoring kernels at the headline shape (R=10240, C=128, I=20001)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easydgl_amd import ops, _lib
from easydgl_amd._lib import lib, check
from easydgl_amd.ops import _ptr, _stream
R, C, I = 10240, 128, 20001
dt = torch.bfloat16
rows = (torch.randn(R, C, device="cuda") * 0.5).to(dt); tab = (torch.randn(I, C, device="cuda") * 0.3).to(dt)
bias = torch.zeros(I - 1, device="cuda"); lab = torch.randint(1, I, (R,), device="cuda")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
lse, ll, _ = ops.score_lse(rows, tab, bias, lab, 0, I)
coef = torch.rand(R, device="cuda") * 1e-4
d_rows = torch.empty_like(rows); d_tab = torch.empty(I, C, device="cuda"); d_bias = torch.empty(I - 1, device="cuda")
ws = torch.empty(lib.edgl_score_bwd_workspace(R, C, I, I, 1), device="cuda")
t_f = timeit(lambda: ops.score_lse(rows, tab, bias, lab, 0, I))
_lib.profiler.start()
def bwd():
    check(lib.edgl_score_ce_bwd(_ptr(rows), _ptr(tab), _ptr(bias), _ptr(lab), _ptr(lse), _ptr(coef), None, R, C, I, 0, I, None,
                                _ptr(d_rows), _ptr(d_tab), _ptr(d_bias), _ptr(ws), 1, _stream()))
t_b = timeit(bwd)
gf = 2.0 * R * C * I / 1e9
print(f"EDGL_DBG={os.environ.get('EDGL_DBG','0'):>2s}  fwd {t_f:7.1f} us ({gf/t_f*1e6/1e3:6.0f} TF)   bwd(dy+dw) {t_b:7.1f} us ({4*gf/t_b*1e6/1e3:6.0f} TF)")

"""Diagnostic: per-phase wave cycles of strip_gemm_kernel on the headline QKVT projection (library built with
-DEDGL_PHASE_TIMING for k_gemm2.hip).  python tools/phase_probe_gemm.py path/to/lib.so [M K N b_kc]"""
import ctypes
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shutil.copy(sys.argv[1], os.path.join(ROOT, "easydgl_amd", "libeasydgl_hip.so"))
import torch  # noqa: E402
from easydgl_amd import _lib, ops  # noqa: E402

M, K, N = (int(a) for a in sys.argv[2:5]) if len(sys.argv) > 4 else (51712, 384, 512)
b_kc = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
A = torch.randn(M, K, device="cuda").bfloat16()
W = (torch.randn(N, K, device="cuda") if b_kc else torch.randn(K, N, device="cuda")).bfloat16()
bias = torch.zeros(N, device="cuda")
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    ops.gemm(A, W, M, N, K, K, (K if b_kc else N), True, b_kc, torch.bfloat16, flags=_lib.EPI_BIAS, bias=bias)
torch.cuda.synchronize()
raw.edgl_debug_phase_cycles_gemm(buf, 1)
n = 5
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(n):
    ops.gemm(A, W, M, N, K, K, (K if b_kc else N), True, b_kc, torch.bfloat16, flags=_lib.EPI_BIAS, bias=bias)
e1.record()
torch.cuda.synchronize()
raw.edgl_debug_phase_cycles_gemm(buf, 0)
names = ["resident: weight slice -> LDS", "resident: strip loads issued", "resident: MFMA loop (+ strip wait)", "resident: epilogue",
         "stream: prologue (strip + chunk 0)", "stream: MFMA loop issue", "stream: MFMA drain + epilogue", "stream: chunk -> LDS + barrier"]
tot = sum(buf[:8]) or 1
for i, nm in enumerate(names):
    if buf[i]:
        print(f"{nm:36s} {buf[i] / n:16.0f} wave-cycles/launch  {100.0 * buf[i] / tot:5.1f}%")
print(f"launch time {e0.elapsed_time(e1) / n * 1e3:.1f} us (instrumented)")

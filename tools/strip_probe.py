"""Per-phase shader-clock stamps of the strip scoring kernels (a -DSTRIP_TIMING build: tools/build_strip_variant.sh timing -DSTRIP_TIMING):
    EDGL_LIB_PATH=tools/variants/lib_timing.so python tools/strip_probe.py
Prints, per role, mean / max over the workgroups of: x fragments, prologue, main loop (and cycles per MFMA slot), drain, epilogue."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from easydgl_amd import ops  # noqa: E402
from easydgl_amd._lib import _cdll, check, lib  # noqa: E402
from easydgl_amd.ops import _ptr as p, _stream  # noqa: E402

R, C, I = 10240, 128, 20001
g = torch.Generator(device="cuda").manual_seed(1)
rows = (torch.randn(R, C, device="cuda", generator=g) * 0.6).bfloat16()
tab = (torch.randn(I, C, device="cuda", generator=g) * 0.4).bfloat16()
bias = torch.randn(I - 1, device="cuda", generator=g) * 0.3
labels = torch.randint(1, I, (R,), device="cuda", generator=g)
u = torch.rand(R, device="cuda", generator=g)
labels[u < 0.35] = I - 2
labels[u > 1.0 - float(os.environ.get("ZERO", "0.475"))] = 0
rows_c, lab_c, perm, inv, nvalid = ops.compact_rows(rows, labels)
code = ops._code(rows)
wsf = torch.empty(lib.edgl_score_flash_workspace(R, C, I, I, code), device="cuda")
lse = torch.empty(R, device="cuda"); ll = torch.zeros(R, device="cuda"); coef = torch.empty(R, device="cuda")
d_rows = torch.empty_like(rows_c); d_tab = torch.empty((I, C), device="cuda"); d_b = torch.empty(I - 1, device="cuda")
stamps = torch.zeros(1024 * 8, device="cuda", dtype=torch.int64)
_cdll.edgl_debug_strip_stamps.argtypes = [ctypes.c_void_p]
_cdll.edgl_debug_strip_stamps.restype = None


def fwd():
    check(lib.edgl_score_flash_fwd_coef(p(rows_c), p(tab), p(bias), p(lab_c), R, C, I, p(nvalid), p(lse), p(ll), p(coef), p(wsf), code,
                                        _stream()), "fwd")


def bwd():
    check(lib.edgl_score_flash_bwd(p(rows_c), p(tab), p(bias), p(lab_c), p(lse), p(coef), None, R, C, I, 0, I, p(nvalid), p(d_rows),
                                   p(d_tab), p(d_b), p(wsf), code, _stream()), "bwd")


for name, fn in (("rows pass (ROLE_YF)", fwd), ("table pass (ROLE_W)", bwd)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    stamps.zero_()
    _cdll.edgl_debug_strip_stamps(ctypes.c_void_p(stamps.data_ptr()))
    fn()
    torch.cuda.synchronize()
    _cdll.edgl_debug_strip_stamps(None)
    s = stamps.view(-1, 8).cpu()
    s = s[s[:, 5] != 0]
    d = (s[:, 1:6] - s[:, 0:5]).double()
    nt = s[:, 6].double()
    names = ["x fragments", "prologue", "main loop", "drain", "epilogue"]
    print(f"{name}: {len(s)} workgroups, tiles per workgroup {nt.min():.0f}..{nt.max():.0f}, whole kernel {float((s[:, 5] - s[:, 0]).double().mean()):.0f} cycles (max {float((s[:, 5] - s[:, 0]).max())})")
    for i, n_ in enumerate(names):
        print(f"   {n_:12s} mean {float(d[:, i].mean()):9.0f}  max {float(d[:, i].max()):9.0f}")
    print(f"   main loop cycles per MFMA slot: {float((d[:, 2] / (nt * 64)).mean()):.1f}")
    if hasattr(_cdll, "edgl_debug_strip_phases"):
        ph = (ctypes.c_ulonglong * 48)()
        _cdll.edgl_debug_strip_phases(ph)
        n_t = (float(nt.max()) + 1) // 2 * 2      # tiles of the probed workgroup (padded to a pair)
        for w in range(4):
            v = [x / n_t for x in ph[12 * w:12 * w + 12]]
            print("   wave %d, cycles per tile — A: slots 0-3 %.0f | 4-12 (stores) %.0f | 13-15 %.0f | barrier %.0f | O half %.0f   B: slots 0-3 %.0f | 4-12 (loads) %.0f | 13-15 %.0f | - %.0f | O half %.0f | between %.0f"
                  % (w, v[0], v[1], v[2], v[3], v[8], v[4], v[5], v[6], v[7], v[9], v[11]))

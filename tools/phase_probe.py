"""Diagnostic: per-phase wave cycles of the BiMAU backward kernel (needs a library built with -DEDGL_PHASE_TIMING, see
k_bimau_bwd.hip; cycles are per wave with the other resident waves of the SIMD competing).  python tools/phase_probe.py path/to/lib_phase.so"""
import ctypes
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1:
    shutil.copy(sys.argv[1], os.path.join(ROOT, "easydgl_amd", "libeasydgl_hip.so"))
import torch  # noqa: E402
import bench  # noqa: E402
from easydgl_amd import _lib  # noqa: E402
from easydgl_amd.engine import TrainEngine  # noqa: E402

NAMES = ["X: S + softmax", "X: G/dA/dlam/dV sweep", "X: dz, row term", "X: epilogue", "Y: pack -> LDS", "Y: tile operands", "Y: mark loop + dH store", "Y: reduction + partials",
         "Z: softmax", "Z: dP/dS/dQ/dK/dT sweep", "Z: epilogue", "Z: prefetch issue", "Z: first use of operands", "Z: S products", "-", "-"]


def main():
    dev = torch.device("cuda:0")
    cfg = dict(bench.HEADLINE)
    model, feats, labels = bench.make_model_and_batch(cfg, "bf16", dev, seed=1)
    eng = TrainEngine(model, cfg["batch"], use_graph=False)
    eng.load_batch(feats, labels)
    for _ in range(3):
        eng.step()
    torch.cuda.synchronize()
    raw = ctypes.CDLL(_lib.LIB_PATH)
    buf = (ctypes.c_ulonglong * 16)()
    raw.edgl_debug_phase_cycles(buf, 1)
    n = 5
    for _ in range(n):
        eng.step()
    torch.cuda.synchronize()
    raw.edgl_debug_phase_cycles(buf, 0)
    jobs = cfg["batch"] * cfg["num_heads"]
    tot = sum(buf[:14])
    for i, nm in enumerate(NAMES):
        if nm != "-":
            print(f"{nm:28s} {buf[i] / n / jobs:12.0f} cyc/job  {100.0 * buf[i] / tot:5.1f}%")
    print(f"total {tot / n / jobs:.0f} cycles per (b, head) job")
    if buf[15]:
        print(f"sweep 2: {buf[14] / n / jobs:.0f} shader cycles and {buf[15] / n / jobs * 10:.0f} ns per wave -> {buf[14] / buf[15] / 10:.2f} GHz")


main()

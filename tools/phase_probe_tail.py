"""Diagnostic: per-phase wave cycles of the fused tail kernels at the headline shape.  Library variants:
  forward : tools/build_phase_variant.sh k_tail                      -> python tools/phase_probe_tail.py variants/lib_phase_k_tail.so
  backward: EXTRA=-DEDGL_PHASE_BWD tools/build_phase_variant.sh k_tail -> python tools/phase_probe_tail.py variants/lib_phase_k_tail.so bwd"""
import ctypes
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
shutil.copy(sys.argv[1], os.path.join(ROOT, "easydgl_amd", "libeasydgl_hip.so"))
BWD = len(sys.argv) > 2 and sys.argv[2] == "bwd"
import torch  # noqa: E402
import bench  # noqa: E402
from easydgl_amd import _lib  # noqa: E402
from easydgl_amd.engine import TrainEngine  # noqa: E402

model, feats, labels = bench.make_model_and_batch(dict(bench.HEADLINE), "bf16", torch.device("cuda", 0), 9876)
eng = TrainEngine(model, 512, use_graph=False)
eng.load_batch(feats, labels)
raw = ctypes.CDLL(_lib.LIB_PATH)
buf = (ctypes.c_ulonglong * 16)()
for _ in range(3):
    eng.step()
torch.cuda.synchronize()
raw.edgl_debug_phase_cycles_tail(buf, 1)
n = 5
for _ in range(n):
    eng.step()
torch.cuda.synchronize()
raw.edgl_debug_phase_cycles_tail(buf, 0)
if BWD:
    names = ["head inputs + gathered row gradients", "LN3' + gelu' + dX(Wt)", "o, a1 -> z2", "LN2' + d_o", "hidden layer, two halves",
             "ao, x_in -> z1", "LN1' + d_ao + d_res1", "dX(Wo) + d_att"]
else:
    names = ["inputs -> LDS", "G1 + epilogue", "barrier + copy_out(ao)", "LN1", "barrier + copy_out(a1)", "G2 halves + GELU pass",
             "G3 halves", "o + LN2 + y", "head"]
tot = sum(buf[:len(names)]) or 1
waves = 512 * 8 * n      # the head phases run in one of the two blocks only: their per-wave figure is of that block
for i, nm in enumerate(names):
    print(f"{nm:46s} {buf[i] / waves:10.0f} cycles/wave  {100.0 * buf[i] / tot:5.1f}%")
print(f"total {tot / waves:.0f} cycles per wave (both blocks of the step)")

"""On the GPU box: engine step time of the headline workload with overrides, e.g.
   python tools/try_shape.py num_units=256 num_items=20000 [steps=100]
(A/B of library switches at shapes bench.py has no row for: EDGL_SCORE_NW=4 python tools/try_shape.py num_units=256 ...)"""
import sys
sys.path.insert(0, ".")
import torch  # noqa: E402
import bench  # noqa: E402
from easydgl_amd.engine import TrainEngine  # noqa: E402

kw = dict(a.split("=") for a in sys.argv[1:])
steps = int(kw.pop("steps", 100))
c = dict(bench.HEADLINE, **{k: (float(v) if "." in v else int(v)) for k, v in kw.items()})
dev = torch.device("cuda", 0)
model, feats, labels = bench.make_model_and_batch(c, "bf16", dev, 9876)
eng = TrainEngine(model, c["batch"], use_graph=False)
for _ in range(10):
    loss = eng.step(feats, labels)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(steps):
    loss = eng.step(feats, labels)
b.record()
torch.cuda.synchronize()
print(f"{' '.join(sys.argv[1:])}: {a.elapsed_time(b) / steps:.4f} ms/step, loss {float(loss):.5f}", flush=True)

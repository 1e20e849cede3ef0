"""Diagnostic: BASELINE.json configs[2] (1 M items, T = 201, C = 256, batch 512, masklen 40) — autograd and engine step times, the
library profiler's per-call table, one evaluation batch.   python tools/try_config3.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
c = dict(bench.HEADLINE, num_items=1_000_000, seqslen=200, num_units=256, masklen=40)
dev = torch.device("cuda", 0)
t0 = time.time()
model, feats, labels = bench.make_model_and_batch(c, "bf16", dev, 9876)
print("built", time.time() - t0, flush=True)
for i in range(3):
    loss = model.train_step(feats, labels)
    torch.cuda.synchronize()
    print("autograd step", i, float(loss), flush=True)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(5):
    loss = model.train_step(feats, labels)
b.record(); torch.cuda.synchronize()
print("autograd ms/step", a.elapsed_time(b) / 5, float(loss), flush=True)
from easydgl_amd.engine import TrainEngine
eng = TrainEngine(model, c["batch"], use_graph=False)
for i in range(3):
    l = eng.step(feats, labels)
torch.cuda.synchronize()
a.record()
for _ in range(5):
    l = eng.step(feats, labels)
b.record(); torch.cuda.synchronize()
print("engine ms/step", a.elapsed_time(b) / 5, float(l), flush=True)
from easydgl_amd._lib import profiler
profiler.start()
for _ in range(3):
    eng.step(feats, labels)
torch.cuda.synchronize(); profiler.stop()
for k, v in sorted(profiler.summary().items(), key=lambda kv: -kv[1][1])[:14]:
    print(k, v[0] // 3, round(v[1] / 3, 3))
ef, el = feats.copy(), labels
try:
    from easydgl_amd import data as D
    ids = torch.tensor(D.synthetic_batch(c["num_items"], c["seqslen"], 64, seed=1)[0]).cuda()
    ts = torch.tensor(D.synthetic_batch(c["num_items"], c["seqslen"], 64, seed=1)[1]).cuda()
    f2, l2 = D.device_mask_last(ids, ts, model.mask)
    model.reset_metrics(); model.eval_step(f2, l2, mask_seen=True); print("eval", model.metrics())
except Exception as e:
    print("eval failed:", repr(e)[:300])

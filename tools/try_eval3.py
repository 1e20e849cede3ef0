"""Diagnostic: the evaluation step at BASELINE.json configs[2] (1 M items, T = 201, C = 256, 512 sequences): step time and the
per-call table of the library's profiler.   python tools/try_eval3.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from types import SimpleNamespace
import easydgl_amd
from easydgl_amd import data as D
from easydgl_amd._lib import profiler
dev = torch.device("cuda", 0)
num_items, C = 1_000_000, 256
F = SimpleNamespace(model="EasyDGL", num_items=num_items, num_units=C, num_heads=8, num_blocks=1, seqslen=200, masklen=40, time_scale=86400.0,
                    learning_rate=5e-4, l2_reg=1e-4, ct_reg=1e-7, hidden_dropout_rate=0.1, attention_probs_dropout_rate=0.1,
                    mark_table=D.synthetic_mark_table(num_items, 16), compute_dtype="bf16", num_train_steps=None, num_warmup_steps=None, seed=9876)
model = easydgl_amd.ranking(F).finalize(dev)
ids, ts = D.synthetic_batch(num_items, 200, 512, seed=9876)
feats, _ = D.device_mask_last(torch.tensor(ids, device=dev), torch.tensor(ts, device=dev), model.mask)
for _ in range(5): model.eval_topk_sharded(feats, mask_seen=True, K=100)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): model.eval_topk_sharded(feats, mask_seen=True, K=100)
b.record(); torch.cuda.synchronize()
print("eval step ms", a.elapsed_time(b) / 20)
profiler.start()
for _ in range(5): model.eval_topk_sharded(feats, mask_seen=True, K=100)
torch.cuda.synchronize(); profiler.stop()
for k, v in sorted(profiler.summary().items(), key=lambda kv: -kv[1][1])[:14]:
    print(k, v[0] // 5, round(v[1] / 5, 3))

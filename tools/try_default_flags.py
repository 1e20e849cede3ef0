"""On the GPU box: the reference's DEFAULT flags (main.py:35-44: 50 units in one head, 3 blocks, seqslen 30, masklen 6, batch 128)
through the static engine — eager launch sequence against the captured HIP graph, and the launch count of a step.
python tools/try_default_flags.py"""
import sys
sys.path.insert(0, ".")
import torch  # noqa: E402
import bench  # noqa: E402
from easydgl_amd.engine import TrainEngine  # noqa: E402

c = dict(bench.HEADLINE, num_units=50, num_heads=1, num_blocks=3, seqslen=30, masklen=6, batch=128, l2_reg=0.0, ct_reg=0.0,
         hidden_dropout_rate=0.0, attention_probs_dropout_rate=0.0)
dev = torch.device("cuda", 0)
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for use_graph in (False, True, False, True):
    model, feats, labels = bench.make_model_and_batch(c, "bf16", dev, 9876)
    eng = TrainEngine(model, c["batch"], use_graph=use_graph)
    for _ in range(10):
        loss = eng.step(feats, labels)
    torch.cuda.synchronize()
    a.record()
    for _ in range(200):
        loss = eng.step(feats, labels)
    b.record()
    torch.cuda.synchronize()
    print(f"use_graph={use_graph}: {a.elapsed_time(b) / 200:.4f} ms/step, loss {float(loss):.5f}", flush=True)

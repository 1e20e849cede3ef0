"""Times edgl_mask_topk (K6) alone: python tools/topk_bench.py [rows] [items] [K]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from easydgl_amd import ops

R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20001
K = int(sys.argv[3]) if len(sys.argv) > 3 else 100
g = torch.Generator(device="cuda").manual_seed(0)
x0 = torch.randn(R, n, device="cuda", generator=g) * 3.0
seen = torch.randint(0, n, (R, 101), device="cuda", generator=g)
x = x0.clone()
for _ in range(3):
    ops.mask_topk(x, 0, seen, K)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
N = 30
a.record()
for _ in range(N):
    ops.mask_topk(x, 0, seen, K)
b.record(); torch.cuda.synchronize()
us = a.elapsed_time(b) / N * 1e3
print(f"mask_topk R={R} n={n} K={K}: {us:.1f} us  ({R * n * 4 / us / 1e6:.2f} TB/s of one pass over the logits)")

"""On the GPU box: time of the three BiMAU kernels at the headline shape against the sequences' real length (all samples the same
length n: the key-tile skip of csrc/bimau_common.h should make the kernels scale with the key tiles behind the first real key).
    python tools/bimau_len_bench.py [n ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from easydgl_amd import _lib, ops  # noqa: E402
from oracle import easydgl_oracle as O  # noqa: E402

lib = _lib.lib
B, T, C, H, E = 512, 101, 128, 8, 16
dh = C // H
rng = np.random.default_rng(0)
qkvt = torch.tensor(rng.standard_normal((B, T, 4 * C)) * 0.4, dtype=torch.bfloat16).cuda()
resid = torch.tensor(rng.standard_normal((B, T, C)), dtype=torch.bfloat16).cuda()
spans = torch.tensor(rng.uniform(0, 5, size=(B, T)), dtype=torch.float32).cuda()
W1 = torch.tensor(rng.standard_normal((dh + 1, dh * E)) * 0.2, dtype=torch.float32).cuda()
b1 = torch.zeros(dh * E, device="cuda"); w = torch.tensor(rng.standard_normal((E, dh)) * 0.3, dtype=torch.float32).cuda(); sc = torch.zeros(E, device="cuda")
d_out = torch.tensor(rng.standard_normal((B, T, C)), dtype=torch.bfloat16).cuda()
code = ops._code(qkvt)
pack = torch.empty(lib.edgl_bimau_pack_bytes(C, H, E, code), device="cuda", dtype=torch.uint8)
_lib.check(lib.edgl_bimau_pack(W1.data_ptr(), b1.data_ptr(), w.data_ptr(), sc.data_ptr(), C, H, E, pack.data_ptr(), code, None), "pack")
state = ops.make_rng_state("cuda", seed=1); ops.rng_advance(state)
bits = torch.zeros(int(lib.edgl_bimau_dropbits_bytes(B, T, H)) // 4, device="cuda", dtype=torch.int32)
_lib.check(lib.edgl_bimau_dropbits(B, T, H, 0.1, state.data_ptr(), 10, bits.data_ptr(), None), "bits")
out = torch.empty((B, T, C), device="cuda", dtype=torch.bfloat16); lam = torch.empty((H * B, T, E), device="cuda")
saved = torch.empty(lib.edgl_bimau_saved_bytes(B, T, C, H, code), device="cuda", dtype=torch.uint8)
dq = torch.empty_like(qkvt); g = torch.empty((dh + 3) * dh * E + E, device="cuda")
n1, n2, n3 = (dh + 1) * dh * E, dh * E, E * dh
ws = torch.empty(lib.edgl_bimau_bwd_workspace(B, T, C, H, E, code), device="cuda", dtype=torch.uint8)
d_lam = torch.zeros((H * B, T, E), device="cuda")
order = torch.arange(2 * B, device="cuda", dtype=torch.int32)


def run(ids, marks, use_order):
    o = order.data_ptr() if use_order else None
    def fwd():
        _lib.check(lib.edgl_bimau_fwd_ord(qkvt.data_ptr(), resid.data_ptr(), C, ids.data_ptr(), spans.data_ptr(), marks.data_ptr(), pack.data_ptr(), B, T, C, H, E,
                                          0.1, state.data_ptr(), 10, bits.data_ptr(), 0.0, out.data_ptr(), lam.data_ptr(), saved.data_ptr(), None, o, 0, code, None), "fwd")
    def bwd():
        _lib.check(lib.edgl_bimau_bwd_ord(qkvt.data_ptr(), ids.data_ptr(), spans.data_ptr(), marks.data_ptr(), pack.data_ptr(), d_out.data_ptr(), d_lam.data_ptr(),
                                          None, 0, None, 0.0, None, lam.data_ptr(), saved.data_ptr(), B, T, C, H, E, 0.1, state.data_ptr(), 10, bits.data_ptr(), 0.0,
                                          dq.data_ptr(), g.data_ptr(), g[n1:].data_ptr(), g[n1 + n2:].data_ptr(), g[n1 + n2 + n3:].data_ptr(), ws.data_ptr(), o, 0, code, None), "bwd")
    res = []
    for f in (fwd, bwd):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        a, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20):
            f()
        b_.record(); torch.cuda.synchronize()
        res.append(a.elapsed_time(b_) / 20 * 1e3)
    return res


mt = O.synthetic_mark_table(30, E, multi_hot=False)
cases = [int(a) for a in sys.argv[1:]] or [101, 85, 69, 53, 37, 21, 5, 0, -1]
for n in cases:
    ids = rng.integers(1, 30, size=(B, T))
    if n >= 0:
        ids[:, :T - n] = 0
        tag = f"n={n}"
    else:   # the benchmark's own lengths: U{5..T}
        for b in range(B):
            ids[b, :T - rng.integers(5, T + 1)] = 0
        tag = "n~U{5..T}"
    ids_t = torch.tensor(ids).cuda(); marks = torch.tensor(mt[ids].astype(np.uint8)).cuda()
    _lib.check(lib.edgl_bimau_job_order(ids_t.data_ptr(), B, T, order.data_ptr(), None), "order")
    for env in ("0", "1"):
        os.environ["EDGL_BIMAU_SKIP"] = env
        for uo in ((False, True) if (n < 0 and env == "1") else (False,)):
            f, b_ = run(ids_t, marks, uo)
            print(f"{tag:10s} skip={env} ordered={int(uo)}  fwd {f:7.1f} us   bwd (3 kernels + reduce) {b_:7.1f} us")

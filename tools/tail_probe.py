"""Diagnostic: the fused block-tail kernels alone (forward / backward), re-launched on the buffers an engine step left.
    python tools/tail_probe.py [num_units] [heads] [batch ...]
At C = 64 two 4-wave workgroups fit a CU (64 KB LDS, <= 256 registers in the forward): batch 256 = one workgroup per CU, batch 512 =
two per CU at once — the ratio of the two times bounds what a second independent chain per CU can give the C = 128 kernel."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import bench  # noqa: E402
from easydgl_amd import ops  # noqa: E402
from easydgl_amd._lib import check, lib  # noqa: E402
from easydgl_amd.engine import TrainEngine  # noqa: E402
from easydgl_amd.ops import _ptr, _stream  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 128
H = int(sys.argv[2]) if len(sys.argv) > 2 else 8
batches = [int(x) for x in sys.argv[3:]] or [256, 512]
N = 50


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N


for B in batches:
    cfg = dict(bench.HEADLINE, num_units=C, num_heads=H, batch=B)
    model, feats, labels = bench.make_model_and_batch(cfg, "bf16", torch.device("cuda", 0), 9876)
    eng = TrainEngine(model, B, use_graph=False)
    assert eng.fused_tail
    eng.load_batch(feats, labels)
    for _ in range(3):
        eng.step()
    torch.cuda.synchronize()
    m, b, blk = model, eng.blk[0], model.layers[0]
    T, M = eng.T, eng.M
    code, st = eng.code, _stream()
    hd = m.hidden_dropout_rate
    dh1 = ops.Drop(hd, m._rng_state, 11) if hd > 0 else ops.NO_DROP
    x, cin = eng.x0, 3 * C
    pk = eng.tail_pack[0]

    def fwd():
        check(lib.edgl_tail_fwd_ct(_ptr(b["att"]), x.data_ptr(), cin, _ptr(pk), _ptr(blk.att_out.bias), _ptr(blk.inter.bias),
                                   _ptr(blk.out.bias), _ptr(m.transform.bias), _ptr(blk.att_ln.gamma), _ptr(blk.att_ln.beta),
                                   _ptr(blk.out_ln.gamma), _ptr(blk.out_ln.beta), _ptr(m.transform_ln.gamma),
                                   _ptr(m.transform_ln.beta), B, T, C, float(dh1.rate), dh1.ptr(), 11, 12,
                                   _ptr(eng.mpos), M, 1, _ptr(b["ao"]), _ptr(b["a1"]), _ptr(b["st1"]), _ptr(b["pre_f"]),
                                   _ptr(b["f"]), _ptr(b["o"]), _ptr(b["y"]), _ptr(b["st2"]), _ptr(eng.pre_t), _ptr(eng.so),
                                   _ptr(eng.st3), _ptr(eng.hrows_c), _ptr(eng.inv), eng.pad[0], eng.pad[1], code, st), "tail_fwd")

    ws = torch.empty(int(lib.edgl_tail_bwd_workspace(B, C)), device="cuda", dtype=torch.float32)
    tl = m.transform_ln

    def bwd():
        check(lib.edgl_tail_bwd_ct(x.data_ptr(), cin, _ptr(b["ao"]), _ptr(b["a1"]), _ptr(b["pre_f"]), _ptr(b["o"]),
                                   _ptr(eng.pre_t), _ptr(eng.so), _ptr(b["st1"]), _ptr(b["st2"]), _ptr(eng.st3),
                                   _ptr(m.compute(blk.att_out.kernel)), _ptr(m.compute(blk.inter.kernel)),
                                   _ptr(m.compute(blk.out.kernel)), _ptr(m.compute(m.transform.kernel)),
                                   _ptr(blk.att_ln.gamma), _ptr(blk.out_ln.gamma), _ptr(tl.gamma), B, T, C, float(dh1.rate),
                                   dh1.ptr(), 11, 12, 1, _ptr(eng.d_rows), _ptr(eng.mpos), M,
                                   _ptr(eng.inv), None, _ptr(eng.d_pre_t), _ptr(eng.d_o),
                                   _ptr(eng.d_pre_f), _ptr(eng.d_ao), _ptr(eng.G1), _ptr(eng.G2),
                                   _ptr(blk.att_ln.gamma.grad), _ptr(blk.att_ln.beta.grad), _ptr(blk.out_ln.gamma.grad),
                                   _ptr(blk.out_ln.beta.grad), _ptr(tl.gamma.grad), _ptr(tl.beta.grad),
                                   _ptr(ws), eng.pad[0], eng.pad[1], code, st), "tail_bwd")

    tf, tb = timed(fwd), timed(bwd)
    print(f"C={C} H={H} B={B}: tail_fwd {tf:7.1f} us   tail_bwd {tb:7.1f} us   (back-to-back launches, incl. the partial reductions of the backward)")
    del eng, model

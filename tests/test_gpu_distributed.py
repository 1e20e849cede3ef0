"""The multi-GPU path on real devices: one process per GPU over RCCL (torch.distributed backend "nccl").  Skipped on a box
with fewer than two GPUs (the protocol itself is covered on CPU with gloo, tests/test_distributed_cpu.py).
  * evaluation: the item table row-sharded over the ranks, one packed all-gather of the local top-K, merge kernel ==
    the unsharded top-K computed by rank 0 alone;
  * training: two data-parallel engine steps on different batches == one process stepping on the averaged gradients."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    import numpy as np
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from easydgl_amd import parallel
    from easydgl_amd.engine import TrainEngine
    from tests._util import build_model, make_problem, to_dev
    prob = make_problem(seed=9, batch=12, num_items=1500, seqslen=20, num_units=32, num_heads=2, num_blocks=1)
    m = build_model(prob, "f32")
    ef = {k: v.to(dev) for k, v in to_dev(prob["efeats"]).items()}
    v1, i1 = m.eval_topk_sharded(ef, mask_seen=True)          # shards = ranks, RCCL all-gather
    v0, i0 = m.eval_topk(ef, mask_seen=True)
    ok = bool(torch.equal(i0, i1) and torch.equal(v0, v1))
    # data parallel: each rank steps on its half of the batch; the averaged gradients must equal the mean of the two ranks'
    feats, labels = to_dev(prob["feats"]), torch.as_tensor(prob["labels"]).to(dev)
    half = slice(rank * 6, rank * 6 + 6)
    f_half = {k: v[half].to(dev).contiguous() for k, v in feats.items()}
    eng = TrainEngine(m, 6, use_graph=False)
    eng.load_batch(f_half, labels[half].contiguous())
    eng._issue()
    mine = m._grad_arena.clone()
    parallel.allreduce_mean_(m._grad_arena)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    want = sum(gathered) / world
    ok = ok and bool(torch.allclose(m._grad_arena, want, rtol=1e-6, atol=1e-7))
    out[rank] = ok
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL over xGMI)")
def test_sharded_eval_and_data_parallel_step_over_rccl():
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out.get(r, False) for r in range(world)), dict(out)


def _dp_engine_worker(rank, world, port, out, empty_rank=False, headline=False):
    """Two engine ranks on ONE GPU over gloo (the collective goes through the host: a functional check of the protocol, not of
    RCCL): halves of a batch with different numbers of weighted rows."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from easydgl_amd import parallel
    from easydgl_amd.engine import TrainEngine
    from tests._util import build_model, make_problem, to_dev
    if headline:    # the bf16 engine of the headline width: regulariser inside sweep 1, loss from the row finish's sums
        prob = make_problem(seed=31, batch=12, num_items=2000, seqslen=100, num_units=128, num_heads=8, num_blocks=1, masklen=20,
                            num_events=16)
    else:
        prob = make_problem(seed=31, batch=12, num_items=900, seqslen=24, num_units=32, num_heads=2, num_blocks=1, masklen=5,
                            num_events=4)
    dt, tol = ("bf16", 3e-3) if headline else ("f32", 1e-5)
    labels = torch.as_tensor(prob["labels"]).clone()
    labels[:3, :4] = 0                                    # rank 0's half carries far fewer weighted rows
    if empty_rank:
        labels[:6] = 0                                    # ... or none at all: its scoring passes run over zero rows
    feats = to_dev(prob["feats"])
    half = slice(rank * 6, rank * 6 + 6)
    m = build_model(prob, dt)
    eng = TrainEngine(m, 6, use_graph=False)
    if headline:
        assert eng.fused_tpp and eng.ce_part is not None
    eng.load_batch({k: v[half].contiguous() for k, v in feats.items()}, labels[half].cuda().contiguous())
    eng._dp = True
    eng._global_counts()
    m._grad_arena.zero_()                                  # (the flat arena has alignment gaps between the parameters)
    eng._issue()
    loss_global = eng._dp_allreduce()                      # the step's ONE collective: gradients + this rank's loss share
    ok, info = True, None
    if rank == 0:
        m1 = build_model(prob, dt)
        e1 = TrainEngine(m1, 12, use_graph=False)
        e1.load_batch(feats, labels.cuda().contiguous())
        e1._issue()
        torch.cuda.synchronize()
        g, w = m._grad_arena, m1._grad_arena
        err = float((g - w).abs().max() / w.abs().max())
        ok, info = bool(err < tol), (err, int(eng.counts[0]), int(e1.nvalid))
        want = float(e1.loss)
    else:
        want = None
    # the loss the step returns is the loss of the concatenated batch on EVERY rank (the L2 term counted once)
    wl = torch.tensor([want if want is not None else 0.0], dtype=torch.float64)
    dist.broadcast(wl, src=0)
    lerr = abs(float(loss_global) - float(wl)) / abs(float(wl))
    ok = ok and lerr < (1e-4 if headline else 1e-5) and bool(torch.isfinite(m._grad_arena).all())
    out[rank] = (ok, (info, lerr))
    dist.destroy_process_group()


def test_two_engine_ranks_reproduce_the_global_batch_gradient():
    """2 ranks x B/2 == 1 rank x B — global loss normalisers (TrainEngine._global_counts) + SUM all-reduce."""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_engine_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert all(out[r][0] for r in range(world)), dict(out)


def test_two_bf16_headline_ranks_reproduce_the_global_batch_gradient():
    """The same identity through the kernels the headline engine runs under data parallelism: the TPP term inside sweep 1 with the
    all-reduced mark count (edgl_bimau_bwd_tpp `tpp_sums`), the loss from the row finish's sums over the global row count
    (edgl_ce_loss_parts `wtotal`).  bf16: the two sides add the same per-sample terms in another order."""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_engine_worker, args=(world, _free_port(), out, False, True), nprocs=world, join=True)
    assert all(out[r][0] for r in range(world)), dict(out)


def test_a_rank_without_weighted_rows_still_joins_the_step():
    """one rank's half of the batch has NO weighted row (all labels 0): its scoring passes see zero rows, its
    share of the cross-entropy is 0, and the two ranks still reproduce the single-process gradient and loss."""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_engine_worker, args=(world, _free_port(), out, True), nprocs=world, join=True)
    assert all(out[r][0] for r in range(world)), dict(out)


def test_bench_two_ranks_over_gloo_smoke():
    """bench.py --gpus 2 on whatever box this is: EDGL_BENCH_BACKEND=gloo lets both ranks share device 0 (a functional check of the
    multi-process path of the bench contract: launcher re-exec, barriers, max-over-ranks time, one JSON line from rank 0)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EDGL_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-extras",
                        "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 2 and j["steps"] == 5 and j["config"]["global_batch"] == 1024 and j["value"] > 0 and j["scaling"] == "weak"


def test_bench_eight_ranks_over_gloo_smoke():
    """The launch the driver uses for the scaling curve (`bench.py --gpus 8`: eight ranks, one JSON line, whole-job value) on whatever
    box this is — EDGL_BENCH_BACKEND=gloo puts all eight ranks on device 0.  A functional check that SCALE works the first time an
    8-GPU node exists, not a measurement."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EDGL_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", EDGL_BENCH_SPIN_MS="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["n_gpus"] == 8 and j["config"]["global_batch"] == 8 * 512 and j["config"]["parallelism"] == "dp8" and j["scaling"] == "weak"
    assert j["value"] > 0 and abs(j["value"] - 8 * 512 / (j["ms_per_step"] * 1e-3)) < 1e-2 * j["value"]
    # the step's one collective, timed where it stands (exposed: nothing of the step runs beside it)
    ar = j["config"]["allreduce"]
    assert ar["allreduce_bytes_per_rank"] > 11e6 and ar["allreduce_exposed_ms"] > 0 and 0 < ar["allreduce_exposed_share_of_step"] < 1
    # ... and the SAME line carries north_star's other multi-GPU path: the evaluation step with the item table sharded over the
    # eight ranks, one packed all-gather of the local top-100 and the merge
    rows = j["extras"]["eval_sharded"]
    assert rows and all(r_["shards"] == 8 and r_["ms_per_eval_step"] > 0 and r_["allgather_ms"] is not None for r_ in rows)
    assert rows[0]["allgather_bytes_gathered"] == 8 * 512 * 2 * 100 * 4
    # ... and the scoring + cross-entropy step with the table row-sharded for TRAINING (vocab-parallel loss through the
    # strip kernels, one all-gather + one all-reduce)
    vp = j["extras"]["vocab_parallel_ce"]
    assert "error" not in vp, vp
    assert vp["shards"] == 8 and vp["ms_per_call"] > 0 and 3.0 < vp["loss"] < 12.0


def test_bench_autograd_path_is_single_gpu_only():
    """The autograd path has no global-count protocol: averaging per-rank gradients is not the global-batch gradient,
    so `--path autograd` refuses a world size above one instead of timing a wrong step."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, EDGL_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", EDGL_BENCH_SPIN_MS="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--path", "autograd",
                        "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0 and "single-GPU only" in (r.stderr + r.stdout)


def test_bench_refuses_a_world_size_that_differs_from_gpus():
    """Failure path of the bench contract: a launcher that started WORLD_SIZE ranks for another --gpus value is an error, not a run
    whose n_gpus lies."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE=2" in (r.stderr + r.stdout)


def test_dp_counts_in_one_launch():
    """edgl_dp_counts — the two normalisers a data-parallel step all-reduces (weighted rows, EasyDGL.py:183-185; marks of all labels,
    temporal.py:330) — against the label array itself, at 16 marks (16-byte rows) and at another mark count (byte loop)."""
    import ctypes
    import numpy as np
    from easydgl_amd import _lib
    lib = _lib.lib
    rng = np.random.default_rng(3)
    for (B, M, E, I) in ((512, 20, 16, 2000), (7, 5, 7, 50), (1, 1, 16, 3)):
        mtab = (rng.random((I + 1, E)) < 0.3).astype(np.uint8)
        mtab[0] = 0
        labels = rng.integers(0, I + 1, size=(B, M)).astype(np.int64)
        labels[rng.random((B, M)) < 0.4] = 0
        d_lab, d_mt = torch.as_tensor(labels).cuda(), torch.as_tensor(mtab).cuda()
        counts = torch.full((2,), -5, dtype=torch.int32, device="cuda")
        rc = lib.edgl_dp_counts(ctypes.c_void_p(d_lab.data_ptr()), ctypes.c_void_p(d_mt.data_ptr()), B, M, E,
                                ctypes.c_void_p(counts.data_ptr()), None)
        assert rc == 0, (lib.edgl_last_error() or b"").decode()
        assert counts.cpu().tolist() == [int((labels != 0).sum()), int(mtab[labels].sum())]
    assert lib.edgl_dp_counts(ctypes.c_void_p(d_lab.data_ptr()), ctypes.c_void_p(d_mt.data_ptr()), 4096, 32, 16,
                              ctypes.c_void_p(counts.data_ptr()), None) != 0      # more than 65536 labels: refused


def _vp_hip_worker(rank, world, port, out, dtype_name, C=128):
    """ops.vocab_parallel_ce — the HIP scoring kernels over a row shard of the item table per rank, ONE packed all-gather of
    (log-sum-exp, label logit), ONE all-reduce of d_rows (EasyDGL.py:149-155,177-185) — two ranks on one GPU over
    gloo against an fp64 reference on the same (rounded) operands."""
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from easydgl_amd import ops
    dt = torch.bfloat16 if dtype_name == "bf16" else torch.float32
    g = torch.Generator().manual_seed(17)
    R, I = 300, 5003
    rows = (torch.randn((R, C), generator=g) * 0.5 * (128 / C) ** 0.5).to(dt).cuda()
    table = (torch.randn((I, C), generator=g) * 0.3).to(dt).cuda()
    bias = (torch.randn(I - 1, generator=g) * 0.2).cuda()
    labels = torch.randint(1, I, (R,), generator=g)
    labels[::7] = 0
    labels[3] = I - 1; labels[4] = 1
    labels = labels.cuda()
    loss, d_rows, d_table, d_bias, (i0, i1) = ops.vocab_parallel_ce(rows, table, bias, labels)
    torch.cuda.synchronize()
    # fp64 reference on the same operands, unsharded
    r64 = rows.double().cpu().requires_grad_(True); t64 = table.double().cpu().requires_grad_(True); b64 = bias.double().cpu().requires_grad_(True)
    lab = labels.cpu()
    used = torch.cat([torch.zeros_like(t64[:1]), t64[1:]])
    logits = r64 @ used.T + torch.cat([torch.full((1,), -1000.0, dtype=torch.float64), b64])
    p = torch.softmax(logits, dim=-1)
    w = (lab != 0).double()
    ref = (w * -torch.log(p[torch.arange(R), lab] + 1e-5)).sum() / (w.sum() + 1e-5)
    ref.backward()
    tol = 2e-2 if dtype_name == "bf16" else 2e-4          # (bf16: d_rows leaves the kernel in the activation dtype)
    def rel(a, b):
        return float((a.double().cpu() - b).abs().max() / (b.abs().max() + 1e-30))
    errs = dict(loss=abs(float(loss) - float(ref)) / abs(float(ref)), d_rows=rel(d_rows, r64.grad),
                d_table=rel(d_table[i0:i1], t64.grad[i0:i1]), d_bias=rel(d_bias[max(i0, 1) - 1:i1 - 1], b64.grad[max(i0, 1) - 1:i1 - 1]))
    outside = float(d_table[:i0].abs().max() if i0 else 0.0) + float(d_table[i1:].abs().max() if i1 < I else 0.0)
    ok = errs["loss"] < (1e-3 if dtype_name == "bf16" else 1e-5) and all(v < tol for k, v in errs.items() if k != "loss") and outside == 0.0
    out[rank] = (bool(ok), errs, (i0, i1))
    dist.destroy_process_group()


@pytest.mark.parametrize("dtype_name,C", [("f32", 128), ("bf16", 128), ("bf16", 256), ("bf16", 512)])
def test_vocab_parallel_ce_two_ranks_match_the_unsharded_reference(dtype_name, C):
    """(bf16: the flash form over each rank's item range — the strip kernels of all three widths; f32: the generic flash kernels)"""
    import torch.multiprocessing as mp
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_vp_hip_worker, args=(world, _free_port(), out, dtype_name, C), nprocs=world, join=True)
    assert all(out[r][0] for r in range(world)), dict(out)
    assert out[0][2][0] == 0 and out[0][2][1] == out[1][2][0] and out[1][2][1] == 5003

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

"""world_size-2 gloo test of the multi-GPU protocol (parallel.py): item-sharded top-K with a packed
all-gather + merge, and the flat-arena gradient all-reduce.  The HIP ops are replaced by the oracle's
top-k here (the protocol is what is under test); the GPU path passes ops.mask_topk / ops.topk_merge."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import easydgl_oracle as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from easydgl_amd import parallel as P
    rng = np.random.default_rng(7)
    R, n, K = 5, 1003, 100
    logits = rng.standard_normal((R, n)).astype(np.float32)
    logits[1, 10:200] = 3.0                                     # ties across shard boundaries
    seen = rng.integers(0, n, size=(R, 6))

    def local_topk(i0, i1):
        x = logits[:, i0:i1].copy()
        for r in range(R):
            for s in seen[r]:
                if i0 <= s < i1:
                    x[r, s - i0] = -np.inf
        k = min(K, i1 - i0)
        idx = O.top_k(x, k)
        val = np.take_along_axis(x, idx, 1)
        pv = np.full((R, K), -np.inf, np.float32); pi = np.full((R, K), -1, np.int32)
        pv[:, :k] = val; pi[:, :k] = idx + i0
        return torch.tensor(pv), torch.tensor(pi)

    def merge(cv, ci):
        S = cv.shape[0]
        v = cv.permute(1, 0, 2).reshape(R, S * K).numpy(); i = ci.permute(1, 0, 2).reshape(R, S * K).numpy()
        order = np.lexsort((i, -v), axis=1)[:, :K]              # value desc, id asc
        return torch.tensor(np.take_along_axis(v, order, 1)), torch.tensor(np.take_along_axis(i, order, 1))

    val, idx = P.sharded_topk(local_topk, merge, n, K)
    full = logits.copy()
    full[np.arange(R)[:, None], seen] = -np.inf
    want = O.top_k(full, K)
    ok = bool((idx.numpy() == want).all())
    g = torch.full((10,), float(rank + 1))
    P.allreduce_mean_(g)
    ok = ok and bool(torch.allclose(g, torch.full((10,), (1 + world) / 2)))
    out[rank] = ok
    dist.destroy_process_group()


def test_sharded_topk_and_grad_allreduce_world2():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert all(out[r] for r in range(world)), dict(out)


def test_single_process_is_a_noop():
    from easydgl_amd import parallel as P
    v, i = P.sharded_topk(lambda a, b: (torch.zeros(2, 3), torch.arange(6, dtype=torch.int32).view(2, 3)), None, 10, 3)
    assert i.shape == (2, 3)
    g = torch.ones(4)
    P.allreduce_mean_(g)
    assert torch.equal(g, torch.ones(4))

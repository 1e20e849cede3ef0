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


def _dp_worker(rank, world, port, out, empty_rank=False):
    """Data-parallel protocol of TrainEngine.step (engine.py: _global_counts + allreduce_sum_) with the fp64 oracle as the step:
    the ranks hold halves of a batch with DIFFERENT numbers of weighted rows; global normalisers + a SUM all-reduce must give the
    gradient of the loss over the whole batch."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from easydgl_amd import parallel as P
    from oracle import torch_ref as R
    from tests._util import make_problem
    prob = make_problem(seed=21, batch=6, num_items=60, seqslen=12, num_units=16, num_heads=2, num_blocks=1, masklen=4, num_events=3)
    cfg, mt = prob["cfg"], prob["mark_table"]
    labels = np.asarray(prob["labels"]).copy()
    labels[0, :3] = 0                                             # rank 0's half carries fewer weighted rows than rank 1's
    if empty_rank:
        labels[:3] = 0                                            # ... or none: its cross-entropy share and its gradient of it are 0
    half = slice(rank * 3, rank * 3 + 3)
    feats = {k: np.asarray(v)[half] for k, v in prob["feats"].items()}
    # the two integers the engine all-reduces before the step
    counts = torch.tensor([int((labels[half] != 0).sum()), int(np.asarray(mt)[labels[half]].sum())], dtype=torch.int64)
    dist.all_reduce(counts)
    p = R.to_torch_params(prob["params"])
    loss, _ = R.train_loss(cfg, p, mt, feats, labels[half], w_total=float(counts[0]), nm_total=float(counts[1]))
    loss.backward()
    names = sorted(p)
    flat = torch.cat([p[k].grad.reshape(-1) for k in names])
    P.allreduce_sum_(flat)
    # the l2 term is not a batch sum: every rank added it in full
    l2 = torch.cat([(cfg.l2_reg * p[k].detach() if k in R.O.EMBEDDING_TABLES else torch.zeros_like(p[k])).reshape(-1) for k in names])
    flat = flat - (world - 1) * l2
    q = R.to_torch_params(prob["params"])
    ref, _ = R.train_loss(cfg, q, mt, prob["feats"], labels)
    ref.backward()
    want = torch.cat([q[k].grad.reshape(-1) for k in names])
    # the naive protocol (local normalisers, mean of the gradients) is NOT the global gradient when the counts differ
    p2 = R.to_torch_params(prob["params"])
    l_naive, _ = R.train_loss(cfg, p2, mt, feats, labels[half])
    l_naive.backward()
    naive = torch.cat([p2[k].grad.reshape(-1) for k in names])
    P.allreduce_mean_(naive)
    err = float((flat - want).abs().max() / want.abs().max())
    err_naive = float((naive - want).abs().max() / want.abs().max())
    out[rank] = (err < 1e-10 and bool(torch.isfinite(flat).all()), err_naive > 1e-3 or empty_rank, err, err_naive)
    dist.destroy_process_group()


def test_data_parallel_gradient_is_the_global_batch_gradient():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        ok, naive_differs, err, err_naive = out[r]
        assert ok and naive_differs, (r, err, err_naive)


def test_data_parallel_with_a_rank_that_holds_no_weighted_row():
    """a rank whose half of the batch has no weighted row contributes a zero cross-entropy gradient (0 / global
    count, not 0 / 0) and the SUM protocol still gives the gradient of the whole batch."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dp_worker, args=(world, _free_port(), out, True), nprocs=world, join=True)
    for r in range(world):
        ok, _, err, err_naive = out[r]
        assert ok, (r, err, err_naive)


def _vp_problem(R=37, C=16, I=203, seed=5):
    g = torch.Generator().manual_seed(seed)
    rows = torch.randn((R, C), generator=g, dtype=torch.float64) * 0.7
    table = torch.randn((I, C), generator=g, dtype=torch.float64) * 0.5
    table[0] = 0.0                                        # coding.py:56-57: row 0 of the used table is the zero constant
    bias = torch.randn(I - 1, generator=g, dtype=torch.float64) * 0.3
    labels = torch.randint(1, I, (R,), generator=g)
    labels[::5] = 0                                       # weight-0 rows (EasyDGL.py:180)
    labels[1] = I - 1; labels[2] = 1                      # both ends of the catalogue
    return rows, table, bias, labels


def _vp_reference(rows, table, bias, labels):
    """EasyDGL.py:149-155,177-185 unsharded, by autograd in float64."""
    rows = rows.clone().requires_grad_(True); table = table.clone().requires_grad_(True); bias = bias.clone().requires_grad_(True)
    used = torch.cat([torch.zeros_like(table[:1]), table[1:]])
    logits = rows @ used.T + torch.cat([torch.full((1,), -1000.0, dtype=torch.float64), bias])
    p = torch.softmax(logits, dim=-1)
    w = (labels != 0).double()
    loss = (w * -torch.log(p[torch.arange(len(labels)), labels] + 1e-5)).sum() / (w.sum() + 1e-5)
    loss.backward()
    return loss.detach(), rows.grad, table.grad, bias.grad


def _vp_worker(rank, world, port, out):
    """parallel.vocab_parallel_ce with float64 callables: the sharded protocol — ONE packed all-gather of
    (log-sum-exp, label logit), ONE all-reduce of d_rows, the table / bias gradients owned by the shard — against the unsharded
    autograd loss and every gradient."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from easydgl_amd import parallel as P
    rows, table, bias, labels = _vp_problem()
    I = table.shape[0]
    full_bias = torch.cat([torch.full((1,), -1000.0, dtype=torch.float64), bias])
    d_table, d_bias_full = torch.zeros_like(table), torch.zeros(I, dtype=torch.float64)
    calls = {"gather": 0, "reduce": 0}
    ag, ar = dist.all_gather_into_tensor, dist.all_reduce

    def count_ag(*a, **k):
        calls["gather"] += 1
        return ag(*a, **k)

    def count_ar(*a, **k):
        calls["reduce"] += 1
        return ar(*a, **k)
    dist.all_gather_into_tensor, dist.all_reduce = count_ag, count_ar

    def lse_local(i0, i1):
        lg = rows @ table[i0:i1].T + full_bias[i0:i1]
        own = (labels >= i0) & (labels < i1)
        lab = torch.where(own, lg[torch.arange(len(labels)), (labels - i0).clamp(0, i1 - i0 - 1)], torch.full((len(labels),), float("-inf"), dtype=torch.float64))
        return torch.logsumexp(lg, dim=1).float().double(), lab      # (the protocol moves f32: keep the comparison honest below)

    def grad_local(i0, i1, lse, coef):
        lg = rows @ table[i0:i1].T + full_bias[i0:i1]
        dl = torch.exp(lg - lse.double()[:, None])
        own = (labels >= i0) & (labels < i1)
        dl[torch.arange(len(labels))[own], (labels - i0)[own]] -= 1.0
        dl = dl * coef.double()[:, None]
        d_table[i0:i1] = dl.T @ rows
        d_bias_full[i0:i1] = dl.sum(0)
        if i0 == 0:
            d_table[0] = 0.0                              # (the zero row carries no gradient; the pad logit is a constant)
        return dl @ table[i0:i1]
    loss, d_rows, (i0, i1) = P.vocab_parallel_ce(labels, I, lse_local, grad_local)
    dist.all_gather_into_tensor, dist.all_reduce = ag, ar
    want_loss, want_rows, want_table, want_bias = _vp_reference(rows, table, bias, labels)
    # every rank holds the loss and d_rows of the WHOLE catalogue; its table / bias gradient rows are exactly the reference's rows
    tol = 1e-6          # (lse / label logits travel as f32)
    ok = abs(float(loss) - float(want_loss)) <= tol * abs(float(want_loss))
    ok = ok and float((d_rows.double() - want_rows).abs().max()) <= tol * float(want_rows.abs().max())
    ok = ok and float((d_table[i0:i1] - want_table[i0:i1]).abs().max()) <= tol * float(want_table.abs().max())
    lo = max(i0, 1)
    ok = ok and float((d_bias_full[lo:i1] - want_bias[lo - 1:i1 - 1]).abs().max()) <= tol * float(want_bias.abs().max())
    ok = ok and float(d_table[:i0].abs().max() if i0 else 0.0) == 0.0 and float(d_table[i1:].abs().max() if i1 < I else 0.0) == 0.0
    out[rank] = (bool(ok), calls["gather"], calls["reduce"], (i0, i1))
    dist.destroy_process_group()


def test_vocab_parallel_cross_entropy_world2_matches_the_unsharded_loss_and_gradients():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_vp_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    shards = []
    for r in range(world):
        ok, n_gather, n_reduce, shard = out[r]
        assert ok, (r, out[r])
        assert n_gather == 1 and n_reduce == 1, out[r]      # ONE packed all-gather + ONE d_rows all-reduce per call
        shards.append(shard)
    assert shards[0][0] == 0 and shards[0][1] == shards[1][0] and shards[1][1] == 203


def test_vocab_parallel_cross_entropy_single_process_is_the_plain_loss():
    from easydgl_amd import parallel as P
    rows, table, bias, labels = _vp_problem(R=11, I=50)
    full_bias = torch.cat([torch.full((1,), -1000.0, dtype=torch.float64), bias])

    def lse_local(i0, i1):
        lg = rows @ table[i0:i1].T + full_bias[i0:i1]
        return torch.logsumexp(lg, dim=1), lg[torch.arange(len(labels)), labels]

    def grad_local(i0, i1, lse, coef):
        dl = torch.exp(rows @ table[i0:i1].T + full_bias[i0:i1] - lse.double()[:, None])
        dl[torch.arange(len(labels)), labels] -= 1.0
        return (dl * coef.double()[:, None]) @ table[i0:i1]
    loss, d_rows, (i0, i1) = P.vocab_parallel_ce(labels, 50, lse_local, grad_local)
    want_loss, want_rows, _, _ = _vp_reference(rows, table, bias, labels)
    assert (i0, i1) == (0, 50) and abs(float(loss) - float(want_loss)) <= 1e-6 * abs(float(want_loss))
    assert float((d_rows.double() - want_rows).abs().max()) <= 1e-6 * float(want_rows.abs().max())

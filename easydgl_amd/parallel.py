"""Multi-GPU pieces of the hot path (one process per GPU, torch.distributed over RCCL/xGMI).

* Evaluation: the item table is split into contiguous row shards; each rank scores the batch against its
  shard, masks seen items, keeps its local top-K with GLOBAL item ids, then ONE all-gather moves a packed
  [R, 2K] 32-bit buffer per rank (K=100, R=512: 400 KB) and a merge kernel orders the S*K candidates
  by (value desc, id asc) — the tf.nn.top_k tie rule of Base.py:181.
* Training: data parallel over sequences (per-sample LayerNorm and per-sequence attention make samples
  independent).  The loss normalises by sums over the BATCH (weighted rows, next-event marks); the engine all-reduces those
  two integers before the step (TrainEngine._global_counts), every rank differentiates its share of the global-batch loss
  and the flat f32 gradient arena is all-reduced with SUM in one call: the result is the gradient over the concatenated batch.

The scoring / top-K / merge callables are injected so that the protocol can be exercised on CPU with the
gloo backend (tests/test_distributed_cpu.py supplies the oracle there; the product path passes the HIP ops).
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(num_rows: int, world: int, rank: int, align: int = 8) -> Tuple[int, int]:
    """Contiguous row shard [i0, i1) of the item table for `rank`.  Shard starts are multiples of `align`
    (the scoring kernels read 16-byte vectors of the transposed table), sizes differ by at most `align`."""
    blocks = (num_rows + align - 1) // align
    base, rem = divmod(blocks, world)
    b0 = rank * base + min(rank, rem)
    b1 = b0 + base + (1 if rank < rem else 0)
    return min(b0 * align, num_rows), min(b1 * align, num_rows)


def sharded_topk(local_topk: Callable[[int, int], Tuple[torch.Tensor, torch.Tensor]],
                 merge: Callable[[torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]],
                 num_rows: int, K: int, group: Optional[dist.ProcessGroup] = None):
    """local_topk(i0, i1) -> (val [R,K] f32, idx [R,K] i32 global ids, -1 = empty); returns the merged top-K."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    i0, i1 = shard_bounds(num_rows, world, rank)
    val, idx = local_topk(i0, i1)
    if world == 1:
        return val, idx
    R = val.shape[0]
    packed = torch.cat([val.view(torch.int32), idx], dim=1).contiguous()          # [R, 2K] 32-bit words
    flat = torch.empty((world * R, 2 * K), dtype=packed.dtype, device=packed.device)
    dist.all_gather_into_tensor(flat, packed, group=group)     # ONE collective, concatenated along rows
    gathered = flat.view(world, R, 2 * K)
    cand_val = gathered[:, :, :K].contiguous().view(torch.float32)
    cand_idx = gathered[:, :, K:].contiguous()
    return merge(cand_val, cand_idx)


def allreduce_mean_(flat_grad: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> None:
    """Data-parallel gradient averaging over the flat arena (one collective per step)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)
    flat_grad.mul_(1.0 / dist.get_world_size(group))


def allreduce_sum_(flat_grad: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> None:
    """Sum of the ranks' gradient arenas (one collective per step): with the loss normalisers taken over the global batch
    (TrainEngine._global_counts) this is the gradient of the loss over the concatenated batch."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=group)


def vocab_parallel_ce(labels: torch.Tensor, num_rows: int,
                      lse_local: Callable[[int, int], Tuple[torch.Tensor, torch.Tensor]],
                      grad_local: Callable[[int, int, torch.Tensor, torch.Tensor], torch.Tensor],
                      group: Optional[dist.ProcessGroup] = None,
                      ce_local: Optional[Callable[[torch.Tensor, torch.Tensor, torch.Tensor], Tuple[torch.Tensor, torch.Tensor]]] = None):
    """Tied-embedding scoring + cross-entropy (EasyDGL.py:149-155,177-185) with the ITEM TABLE ROW-SHARDED over the ranks
    (SURVEY §8(e) row 3): every rank holds the same R scoring rows and labels and scores them against its shard [i0, i1) only.

      lse_local(i0, i1)              -> (lse_loc [R] f32: log-sum-exp of the row's logits over the shard,
                                         lab_loc [R] f32: the label's logit where i0 <= label < i1, -inf elsewhere)
      ONE packed all-gather of [R, 2] f32 per rank  ->  lse = log-sum-exp over the shards, lab = the owner's label logit
      loss = sum_r w_r (-log(p_y + 1e-5)) / (sum_r w_r + 1e-5),  p_y = exp(lab - lse),  w_r = [label_r != 0]     (the +1e-5 twice: :155, :184)
      coef_r = (w_r / W) p_y / (p_y + 1e-5)
      grad_local(i0, i1, lse, coef)  -> d_rows_loc [R, C]: dl . table[i0:i1] with dl = coef (softmax - onehot) on the shard's columns,
                                         and writes the shard's OWN d_table[i0:i1] / d_bias (no collective: a rank owns its rows' gradient)
      ONE SUM all-reduce of d_rows_loc (f32)  ->  d_rows

    Returns (loss 0-d f32, d_rows [R, C] f32, (i0, i1)).  world size 1: no collective.  The callables are injected so that the protocol
    runs on CPU under gloo with the oracle's arithmetic (tests/test_distributed_cpu.py); ops.vocab_parallel_ce passes the HIP kernels."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    i0, i1 = shard_bounds(num_rows, world, rank)
    lse_loc, lab_loc = lse_local(i0, i1)
    if world > 1:
        R = lse_loc.shape[0]
        packed = torch.stack([lse_loc.to(torch.float32), lab_loc.to(torch.float32)], dim=1).contiguous()
        flat = torch.empty((world * R, 2), dtype=packed.dtype, device=packed.device)
        dist.all_gather_into_tensor(flat, packed, group=group)
        g = flat.view(world, R, 2)
        m = g[:, :, 0].max(dim=0).values
        lse = m + torch.log(torch.exp(g[:, :, 0] - m).sum(dim=0))
        lab = g[:, :, 1].max(dim=0).values                        # exactly one shard owns a label: every other entry is -inf
    else:
        lse, lab = lse_loc.to(torch.float32), lab_loc.to(torch.float32)
    if ce_local is not None:      # (lse, lab, labels) -> (loss, coef): the four lines below as one launch (ops passes edgl_ce_loss_fwd)
        loss, coef = ce_local(lse.contiguous(), lab.contiguous(), labels.reshape(-1))
    else:
        w = (labels.reshape(-1) != 0).to(lse.dtype)
        W = w.sum() + 1e-5
        p_y = torch.exp(lab - lse)
        loss = (w * -torch.log(p_y + 1e-5)).sum() / W
        coef = (w / W) * p_y / (p_y + 1e-5)
    d_rows = grad_local(i0, i1, lse.contiguous(), coef.contiguous()).to(torch.float32).contiguous()
    if world > 1:
        dist.all_reduce(d_rows, op=dist.ReduceOp.SUM, group=group)
    return loss, d_rows, (i0, i1)

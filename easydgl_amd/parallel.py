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

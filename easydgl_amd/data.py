"""Batch construction for the EasyDGL path — the semantics of the reference's MAUPostProcessor
(src/dataloader.py:159-206) without its per-example Python py_func (dataloader.py:34-36,183):
masked positions for a whole batch are drawn at once on the device."""
from __future__ import annotations

from typing import Dict, Tuple

import numpy as np
import torch


def mask_last(tokens: torch.Tensor, timestamps: torch.Tensor, mask_id: int) -> Tuple[Dict[str, torch.Tensor], torch.Tensor]:
    """MAUPostProcessor.mask_last (dataloader.py:166-179): position T-1 := MASK, labels = the full sequence."""
    masked = tokens.clone()
    masked[:, -1] = mask_id
    return {"seqs_i": masked, "seqs_t": timestamps}, tokens


def draw_masked_positions(batch: int, seqslen: int, masklen: int, generator: torch.Generator = None,
                          device="cpu") -> torch.Tensor:
    """`masklen` DISTINCT positions in [1, seqslen) per row (np.random.choice(seqslen-1, masklen,
    replace=False) + 1, dataloader.py:34-36 with ignore_head = 1), vectorised: top-k of i.i.d. uniforms."""
    if masklen > seqslen - 1:
        raise ValueError("masklen must be <= seqslen - 1")
    u = torch.rand((batch, seqslen - 1), generator=generator, device=device)
    return (u.topk(masklen, dim=1).indices + 1).to(torch.int64)


def mask_random(tokens: torch.Tensor, timestamps: torch.Tensor, mask_id: int, masked_positions: torch.Tensor):
    """MAUPostProcessor.mask_random (dataloader.py:181-201)."""
    labels = tokens.gather(1, masked_positions)
    masked = tokens.scatter(1, masked_positions, mask_id)
    return {"seqs_i": masked, "seqs_t": timestamps, "masked_positions": masked_positions}, labels


def device_mask_random(tokens: torch.Tensor, timestamps: torch.Tensor, mask_id: int, masklen: int, rng_state: torch.Tensor,
                       stream_id: int = 0x4d41534b):
    """MAUPostProcessor.mask_random for a whole device batch in ONE kernel (edgl_mask_random): draws `masklen`
    distinct positions in [1, T) per row from the counter-based generator keyed on `rng_state` (uint64[2] = seed,
    step — advance the step between batches), replaces the tokens by MASK and gathers the labels."""
    from . import ops
    from ._lib import check, lib
    tokens = tokens.contiguous()
    B, T = tokens.shape
    masked = torch.empty_like(tokens)
    mpos = torch.empty((B, masklen), device=tokens.device, dtype=torch.int64)
    labels = torch.empty((B, masklen), device=tokens.device, dtype=torch.int64)
    check(lib.edgl_mask_random(ops._ptr(tokens), B, T, masklen, int(mask_id), ops._ptr(rng_state), stream_id,
                               ops._ptr(masked), ops._ptr(mpos), ops._ptr(labels), ops._stream()), "edgl_mask_random")
    return {"seqs_i": masked, "seqs_t": timestamps, "masked_positions": mpos}, labels


def device_mask_last(tokens: torch.Tensor, timestamps: torch.Tensor, mask_id: int):
    """MAUPostProcessor.mask_last on the device (edgl_mask_last)."""
    from . import ops
    from ._lib import check, lib
    tokens = tokens.contiguous()
    B, T = tokens.shape
    masked = torch.empty_like(tokens)
    check(lib.edgl_mask_last(ops._ptr(tokens), B, T, int(mask_id), ops._ptr(masked), ops._stream()), "edgl_mask_last")
    return {"seqs_i": masked, "seqs_t": timestamps}, tokens


def synthetic_batch(num_items: int, seqslen: int, batch: int, seed: int = 9876, min_len: int = 5, ids: str = "zipf"):
    """SURVEY.md §8d synthetic sequences: row length ~ U{min_len..T}, left zero padding, Zipf(1.1) item ids
    clipped to [1, num_items-1], float32 timestamps 9.5e8 + cumsum(Exp(mean 3 days)).  T = seqslen + 1.
    (The clip piles ~35 % of the tokens on id num_items-1: `ids="uniform"` draws U[1, num_items) instead — no hot row, every
    gathered table row distinct with high probability: the variant that makes the embedding gather really read HBM.)"""
    ids_mode = ids
    rng = np.random.default_rng(seed)
    T = seqslen + 1
    ids = np.zeros((batch, T), dtype=np.int64)
    ts = np.zeros((batch, T), dtype=np.float32)
    lens = rng.integers(min(min_len, T), T + 1, size=batch)
    for b in range(batch):
        n = int(lens[b])
        ids[b, T - n:] = (rng.integers(1, num_items, size=n) if ids_mode == "uniform"
                          else np.clip(rng.zipf(1.1, size=n), 1, num_items - 1))
        ts[b, T - n:] = (9.5e8 + np.cumsum(rng.exponential(3 * 86400.0, size=n))).astype(np.float32)
    return ids, ts


def synthetic_mark_table(num_items: int, num_events: int, multi_hot: bool = False) -> np.ndarray:
    """[num_items, E] 0/1 table: mark (i mod E) for item i >= 1 (+ a second mark when multi_hot); row 0 = pad."""
    tab = np.zeros((num_items, num_events), dtype=np.uint8)
    idx = np.arange(1, num_items)
    tab[idx, idx % num_events] = 1
    if multi_hot:
        tab[idx, (idx * 7 + 3) % num_events] = 1
    return tab

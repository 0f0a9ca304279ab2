"""Multi-GPU: batch sharding over the ranks of one node (one process per GPU, torch.distributed).

The hot path is element-wise, so a batch of N independent elements is cut into contiguous blocks,
rank g of G owning ``[g*ceil(N/G), min(N, (g+1)*ceil(N/G)))``.  Key material is tiny
and is simply re-created on every device; no collective is needed while the next operation is
element-wise.  The only exchange is the optional final gather of result shards (RCCL all-gather over
xGMI on GPUs — backend "nccl"; "gloo" on CPU tensors in the tests).
"""
from __future__ import annotations

from typing import List, Tuple

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous block partition; trailing ranks may own empty ranges."""
    per = -(-n // world) if world > 0 else n
    return [(min(n, g * per), min(n, (g + 1) * per)) for g in range(world)]


def my_shard(n: int, rank: int, world: int) -> Tuple[int, int]:
    return shard_bounds(n, world)[rank]


def gather_rows(local: torch.Tensor, n_total: int, group=None) -> torch.Tensor:
    """All-gather row blocks of a [rows_g, W] tensor into the full [n_total, W] tensor on every rank.
    Shards are padded to the uniform block size for the collective and the padding is dropped."""
    world = dist.get_world_size(group)
    per = -(-n_total // world)
    W = local.shape[1:]
    padded = torch.zeros((per,) + tuple(W), dtype=local.dtype, device=local.device)
    padded[: local.shape[0]] = local
    out = torch.empty((world * per,) + tuple(W), dtype=local.dtype, device=local.device)
    if local.is_cuda and dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(out, padded, group=group)          # RCCL all-gather over xGMI
    elif local.is_cuda:                                                # gloo with device tensors (plumbing tests): through the host
        host = torch.empty(out.shape, dtype=out.dtype)
        _gather_cpu(host, padded.cpu(), group)
        out.copy_(host)
    else:
        _gather_cpu(out, padded, group)
    return out[:n_total]


def _gather_cpu(out: torch.Tensor, padded: torch.Tensor, group) -> None:
    parts = [torch.empty_like(padded) for _ in range(dist.get_world_size(group))]
    dist.all_gather(parts, padded, group=group)
    out.copy_(torch.cat(parts, dim=0))

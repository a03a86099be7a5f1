"""Data-parallel plumbing for GATRes training: snapshots shard by graph across ranks (one process per GPU), every
rank holds an identical replica of the flat parameter vector, and the ONLY exchange per step is one all-reduce of
the flat fp32 gradient (RCCL over xGMI on the GPU box: torch.distributed backend "nccl"; "gloo" in CPU tests).

The reference is single-process (train.py:306-309); this is the scheme its DataLoader batches extend to:
a global batch of B graphs is split into `world` contiguous shards of B/world graphs.  Because every graph
contributes the same number of masked nodes (int(n_g * mask_rate), utils/auxil.py:154), the mean of the per-rank
mean-squared errors equals the global-batch loss, and the average of per-rank gradients equals its gradient.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.distributed as dist


def shard_graphs(num_graphs: int, rank: int, world: int) -> range:
    """Contiguous, equal-sized shard of the global batch for `rank`; the ragged tail is dropped so every rank has
    the same number of graphs (keeps plain gradient averaging exact)."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    per = num_graphs // world
    return range(rank * per, (rank + 1) * per)


def allreduce_mean_(flat_grads: torch.Tensor, group=None) -> torch.Tensor:
    """In-place average of the flat gradient over the group: ONE collective per step (263 KB for gatres_small)."""
    world = dist.get_world_size(group)
    if world > 1:
        dist.all_reduce(flat_grads, op=dist.ReduceOp.SUM, group=group)
        flat_grads.div_(world)
    return flat_grads


def broadcast_params_(flat_params: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """Make every replica start from rank `src`'s parameters."""
    if dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
    return flat_params

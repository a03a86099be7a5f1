"""Data-parallel plumbing for GATRes training: snapshots shard by graph across ranks (one process per GPU), every
rank holds an identical replica of the flat parameter vector, and the ONLY exchange per step is the all-reduce of
the flat fp32 gradient (RCCL over xGMI on the GPU box: torch.distributed backend "nccl"; "gloo" in CPU tests).

The reference is single-process (train.py:306-309); this is the scheme its DataLoader batches extend to:
a global batch of B graphs is split into `world` contiguous shards of B/world graphs.  Because every graph
contributes the same number of masked nodes (int(n_g * mask_rate), utils/auxil.py:154), the mean of the per-rank
mean-squared errors equals the global-batch loss, and the average of per-rank gradients equals its gradient.

``GATResTrainer`` drives its multi-rank step through ``run_data_parallel_step`` below; ``tests/test_ddp_gloo.py`` runs
the very same function under two gloo ranks with the oracle as the per-rank compute.
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def shard_graphs(num_graphs: int, rank: int, world: int, drop_ragged: bool = False) -> range:
    """Contiguous, equal-sized shard of the global batch for `rank`.  Every rank must hold the same number of graphs
    (equal masked-node counts keep plain gradient averaging exact), so a batch that does not divide by `world` is an
    ERROR unless the caller asks for the ragged tail to be dropped (``drop_ragged=True``: ``num_graphs % world`` graphs
    are then left out, as a DataLoader's ``drop_last`` would) -- never silently."""
    if world <= 0 or not 0 <= rank < world:
        raise ValueError("bad rank/world")
    if num_graphs % world and not drop_ragged:
        raise ValueError(f"a global batch of {num_graphs} graphs does not split evenly over {world} ranks "
                         f"({num_graphs % world} would be dropped); pad the batch or pass drop_ragged=True")
    per = num_graphs // world
    return range(rank * per, (rank + 1) * per)


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def broadcast_params_(flat_params: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """Make every replica start from rank `src`'s parameters."""
    if world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
    return flat_params


class BucketedAllReduce:
    """SUM all-reduce of contiguous ranges ("buckets") of one flat gradient vector, launched as soon as a range is
    final.  ``launch`` is asynchronous: with the nccl (= RCCL) backend the collective runs on the process group's own
    stream behind an event, so the kernels the caller enqueues next (the rest of the backward pass) overlap with it;
    ``wait_all`` makes the current stream wait for every outstanding bucket.  Both are legal inside a hipGraph capture
    (this is the mechanism DDP uses), so the whole step -- backward pieces, collectives, Adam -- replays as one graph.
    The scale 1/world is NOT applied here: ``gatres_adam_step`` folds it into its first multiply (``grad_scale``)."""

    def __init__(self, flat: torch.Tensor, group=None, force: bool = False):
        if flat.dim() != 1 or not flat.is_contiguous():
            raise ValueError("the gradient must be one flat contiguous vector")
        self.flat, self.group = flat, group
        self.world = world_size(group)
        # force: issue the collectives even in a one-rank group (exercises the RCCL path on a single GPU)
        self.active = self.world > 1 or (force and dist.is_available() and dist.is_initialized())
        self._works: List = []
        self.launched: List[Tuple[int, int]] = []          # (lo, hi) of every bucket of the current step, in launch order
        self.last_buckets: List[Tuple[int, int]] = []
        # time_collective: an event pair on the caller's stream around every SYNCHRONOUS collective (eager launches only: events
        # inside a capture cannot be read) -- ``collective_ms()`` then tells what the all-reduce itself took, so that a scaling
        # loss can be attributed (bench.py prints it at N > 1)
        self.time_collective = False
        self._pairs: List = []

    def launch(self, lo: int, hi: int, alone: bool = False) -> None:
        """``alone``: this bucket is the step's only one (the fused path's flat gradient) -- nothing can overlap it, so it goes
        as a SYNCHRONOUS collective: ordered on the caller's stream by the backend itself, without the event hand-over to the
        process group's stream and back that an asynchronous one costs (measured at world size 1, eager launches, RCCL:
        0.3695 -> 0.3527 ms/step)."""
        if not 0 <= lo < hi <= self.flat.numel():
            raise ValueError(f"bad bucket [{lo}, {hi})")
        self.launched.append((lo, hi))
        if self.active:
            if alone and not self._works:
                timed = self.time_collective and self.flat.is_cuda and not torch.cuda.is_current_stream_capturing()
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=False)
                if timed:
                    e1.record()
                    self._pairs.append((e0, e1))
            else:
                self._works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def collective_ms(self) -> List[float]:
        """Durations (ms) of the collectives timed since the last call (synchronises the device)."""
        if not self._pairs:
            return []
        torch.cuda.synchronize(self.flat.device)
        out = [a.elapsed_time(b) for a, b in self._pairs]
        self._pairs = []
        return out

    def wait_all(self) -> None:
        for w in self._works:
            w.wait()
        self._works.clear()
        covered = sorted(self.launched)
        self.launched = []
        self.last_buckets = covered                         # (what the step just sent: tests look at it)
        pos = 0
        for lo, hi in covered:              # every entry of the gradient exactly once
            if lo != pos:
                raise RuntimeError(f"gradient buckets do not tile the vector: gap or overlap at {pos} (next bucket {lo})")
            pos = hi
        if pos != self.flat.numel():
            raise RuntimeError(f"gradient buckets cover {pos} of {self.flat.numel()} entries")


def run_data_parallel_step(pieces: Iterable[Callable[[], Tuple[int, int]]], reducer: BucketedAllReduce,
                           update: Callable[[], None]) -> None:
    """One multi-rank optimisation step.  Each piece enqueues a part of the backward pass and returns the range of the
    flat gradient that is FINAL once it has run; that range's all-reduce starts immediately and overlaps the next piece
    (for gatres_large: one bucket per group of blocks, in reverse block order; for the single fused launch of
    gatres_small there is one piece -- its gradient only exists after the slab reduction that ends the launch
    sequence).  ``update`` (Adam with grad_scale = 1/world) runs after every bucket has arrived."""
    pieces = list(pieces)
    for piece in pieces:
        lo, hi = piece()
        reducer.launch(lo, hi, alone=len(pieces) == 1)
    reducer.wait_all()
    update()


def block_buckets(num_blocks: int, nc: int, blocks_per_bucket: int) -> List[Tuple[int, int, int, int]]:
    """Pieces of the per-op backward in issue order: (b_hi, b_lo, lo, hi) = blocks b_hi-1 .. b_lo and the range of the
    flat parameter vector they finish (include/gatres.h: gatres_model_backward_per_op_part).  The first piece also does
    lin1 (whose parameters close the vector), the last one lin0 (which open it)."""
    stride, p0 = 9 * nc + 4 * nc * nc, 2 * nc
    P = p0 + num_blocks * stride + nc + 1
    out, b_hi = [], num_blocks
    per = max(1, int(blocks_per_bucket))
    while True:
        b_lo = max(0, b_hi - per)
        lo = 0 if b_lo == 0 else p0 + b_lo * stride
        hi = P if b_hi == num_blocks else p0 + b_hi * stride
        out.append((b_hi, b_lo, lo, hi))
        if b_lo == 0:
            return out
        b_hi = b_lo

"""Device-resident snapshot store: the in-memory data format for a WDN with a static topology.

Stands in for ``WDNDataset`` + ``torch_geometric.loader.DataLoader`` on the hot path (reference
utils/DataLoader.py:120-204, train.py:302-303): the reference deep-copies one PyG ``Data`` per snapshot, collates
``batch_size`` of them on the host every iteration (concatenating x / y and offsetting edge_index) and copies the
result to the device.  With one topology per dataset all of that is a row gather:

  * snapshots live on the device as ONE dense fp32 matrix ``[S, N_g]`` (the layout the reference's zarr store already
    has, DataLoader.py:212-242), z-normalised with the scalar mean / std of the whole array (DataLoader.py:142-147,
    auxil.py:18-39: ``(x - mean) / (std + 1e-8)``);
  * a batch is ``index_select`` of ``B`` rows reshaped to ``[B*N_g, 1]`` -- ``x`` and ``y`` are the same values before
    masking (auxil.py:84-98);
  * the block-diagonal ``edge_index`` of a batch size is built once and cached (same tensor object every iteration, so
    the engine's plan cache takes its identity fast path).
"""
from __future__ import annotations

from typing import Dict, Iterator, Optional, Tuple

import torch


class SnapshotStore:
    def __init__(self, snapshots: torch.Tensor, edge_index: torch.Tensor, device=None, normalize: bool = True,
                 mean: Optional[float] = None, std: Optional[float] = None, eps: float = 1e-8):
        if snapshots.dim() != 2:
            raise ValueError("snapshots must be [S, N_g]")
        if edge_index.dim() != 2 or edge_index.shape[0] != 2 or edge_index.dtype != torch.int64:
            raise ValueError("edge_index must be int64 [2, E] of ONE graph")
        device = torch.device(device if device is not None else snapshots.device)
        data = snapshots.to(device=device, dtype=torch.float32)
        self.mean = float(data.mean()) if mean is None else float(mean)
        self.std = float(data.std(unbiased=False)) if std is None else float(std)     # np.std semantics
        self.norm_type = "znorm" if normalize else "unused"
        self.data = ((data - self.mean) / (self.std + eps)).contiguous() if normalize else data.contiguous()
        self.num_snapshots, self.nodes_per_graph = int(data.shape[0]), int(data.shape[1])
        if int(edge_index.max()) >= self.nodes_per_graph if edge_index.numel() else False:
            raise ValueError("edge_index refers to nodes outside the graph")
        self.edge_index_single = edge_index.to(device)
        self.device = device
        self._batched: Dict[int, torch.Tensor] = {}

    def __len__(self) -> int:
        return self.num_snapshots

    def edge_index(self, batch_size: int) -> torch.Tensor:
        """Block-diagonal edge_index of ``batch_size`` copies (PyG ``Batch`` convention), cached per batch size."""
        ei = self._batched.get(batch_size)
        if ei is None:
            E = self.edge_index_single.shape[1]
            off = (torch.arange(batch_size, device=self.device, dtype=torch.int64) * self.nodes_per_graph)
            ei = (self.edge_index_single.repeat(1, batch_size) + off.repeat_interleave(E).unsqueeze(0)).contiguous()
            self._batched[batch_size] = ei
        return ei

    def batch(self, rows: torch.Tensor) -> torch.Tensor:
        """``[B*N_g, 1]`` node features of the given snapshot rows (x == y before masking)."""
        rows = rows.to(self.device)
        return self.data.index_select(0, rows).reshape(-1, 1)

    def batches(self, batch_size: int, shuffle: bool = True, drop_last: bool = False,
                generator: Optional[torch.Generator] = None) -> Iterator[Tuple[torch.Tensor, torch.Tensor, int]]:
        """Yields ``(x, edge_index, num_graphs)`` like iterating the reference's DataLoader (x is also y)."""
        order = (torch.randperm(self.num_snapshots, generator=generator) if shuffle
                 else torch.arange(self.num_snapshots))
        for s in range(0, self.num_snapshots, batch_size):
            rows = order[s:s + batch_size]
            if drop_last and rows.numel() < batch_size:
                break
            yield self.batch(rows), self.edge_index(int(rows.numel())), int(rows.numel())

    def epoch_order(self, shuffle: bool = True, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """The epoch's snapshot order as a DEVICE int64 vector (one upload per epoch)."""
        order = torch.randperm(self.num_snapshots, generator=generator) if shuffle else torch.arange(self.num_snapshots)
        return order.to(self.device)

    def row_batches(self, batch_size: int, shuffle: bool = True, drop_last: bool = False,
                    generator: Optional[torch.Generator] = None, order: Optional[torch.Tensor] = None,
                    first: int = 0) -> Iterator[Tuple[torch.Tensor, torch.Tensor, int]]:
        """Yields ``(rows, edge_index, num_graphs)``: ``rows`` is a DEVICE int64 slice of the epoch's snapshot order
        (uploaded once per epoch; ``order``: use this one, ``epoch_order()``'s), for consumers that collate on the device
        themselves (``GATResTrainer.step_rows``).  ``first``: start at that batch."""
        if order is None:
            order = self.epoch_order(shuffle, generator)
        for s in range(first * batch_size, self.num_snapshots, batch_size):
            rows = order[s:s + batch_size]
            if drop_last and rows.numel() < batch_size:
                break
            yield rows, self.edge_index(int(rows.numel())), int(rows.numel())

    def descale(self, scaled: torch.Tensor) -> torch.Tensor:
        """auxil.py:42-64 for znorm: ``scaled * std + mean`` (note: no eps here, as in the reference)."""
        return scaled * self.std + self.mean if self.norm_type == "znorm" else scaled

"""Synthetic water-distribution-network (WDN) snapshots with C-Town's shape.

The reference's C-Town inputs are not shipped (inputs/ctown.inp is a Git-LFS pointer and
datasets/ctown.zip is a generated artefact), so benchmarks and tests use a seeded
stand-in with the same sizes: 388 junction nodes, 430 pipes -> 860 directed edges, emitted in the
order ``torch_geometric.utils.from_networkx`` produces for an undirected ``nx.Graph``
(reference: utils/DataLoader.py:28-37, :236 -- grouped by source node, neighbours in insertion order).

Also here: the block-diagonal batch collation the reference gets from
``torch_geometric.loader.DataLoader`` (train.py:302-303) and the reference's host-side mask
sampler (utils/auxil.py:143-182) for callers that want bit-identical host masks.
"""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np
import torch

CTOWN_NODES = 388
CTOWN_PIPES = 430


def make_wdn_topology(num_nodes: int = CTOWN_NODES, num_pipes: int = CTOWN_PIPES, seed: int = 0,
                      max_degree: int = 5, locality: int = 24) -> torch.Tensor:
    """Random near-planar WDN: a spanning tree plus ``num_pipes - (num_nodes-1)`` loop-closing chords.

    Returns ``edge_index`` int64 [2, 2*num_pipes]: both directions of every pipe, grouped by source node
    in ascending order, neighbours in the order the pipes were inserted (``from_networkx`` order).
    """
    assert num_pipes >= num_nodes - 1
    rng = np.random.RandomState(seed)
    adj: List[List[int]] = [[] for _ in range(num_nodes)]
    deg = np.zeros(num_nodes, dtype=np.int64)
    have = set()

    def add(u: int, v: int) -> None:
        adj[u].append(v)
        adj[v].append(u)
        deg[u] += 1
        deg[v] += 1
        have.add((min(u, v), max(u, v)))

    for v in range(1, num_nodes):                      # spanning tree, attach to a nearby earlier node
        lo = max(0, v - locality)
        cand = [u for u in range(lo, v) if deg[u] < max_degree - 1]
        if not cand:
            cand = [u for u in range(0, v) if deg[u] < max_degree]
        add(int(rng.choice(cand)), v)
    chords = num_pipes - (num_nodes - 1)
    tries = 0
    while chords > 0:
        tries += 1
        assert tries < 1000000, "could not place chords under the degree cap"
        u = int(rng.randint(0, num_nodes))
        v = u + int(rng.randint(2, locality + 1)) * (1 if rng.rand() < 0.5 else -1)
        if v < 0 or v >= num_nodes or u == v:
            continue
        if (min(u, v), max(u, v)) in have or deg[u] >= max_degree or deg[v] >= max_degree:
            continue
        add(u, v)
        chords -= 1
    src = [u for u in range(num_nodes) for _ in adj[u]]
    dst = [v for u in range(num_nodes) for v in adj[u]]
    return torch.tensor([src, dst], dtype=torch.int64)


def collate_edge_index(edge_index: torch.Tensor, num_nodes: int, batch_size: int) -> torch.Tensor:
    """Block-diagonal batch of ``batch_size`` copies of one topology (PyG ``Batch`` edge_index:
    per-graph edge lists concatenated, node ids offset by ``k * num_nodes``)."""
    E = edge_index.shape[1]
    off = (torch.arange(batch_size, dtype=torch.int64) * num_nodes).repeat_interleave(E)
    return edge_index.repeat(1, batch_size) + off.unsqueeze(0)


def make_snapshots(num_snapshots: int, num_nodes: int = CTOWN_NODES, seed: int = 1) -> torch.Tensor:
    """z-normalised pressures: fp32 [S, num_nodes] ~ N(0, 1) (reference stores ``[S, N_nodes]`` per split,
    utils/DataLoader.py:212-242, then z-norms with scalar mean/std, :142-147)."""
    g = torch.Generator().manual_seed(seed)
    return torch.randn((num_snapshots, num_nodes), generator=g, dtype=torch.float32)


def collate_snapshots(snapshots: torch.Tensor, rows: Sequence[int]) -> torch.Tensor:
    """x = y = [B*N_g, 1] for the chosen snapshot rows (nx_to_pyg, utils/auxil.py:84-98)."""
    return snapshots[list(rows)].reshape(-1, 1).contiguous()


def mask_nodes(num_nodes: int, masking_rate: float, rng: np.random.RandomState,
               required_idx: Sequence[int] = ()) -> np.ndarray:
    """Reference host sampler, utils/auxil.py:143-163: exactly ``int(num_nodes * rate)`` nodes are masked -- the
    ``required_idx`` ones (the sensor nodes of evaluation.py's second pass) always, the rest chosen without replacement
    among the others."""
    required = sorted(set(int(i) for i in required_idx))
    mask_length = int(num_nodes * masking_rate) - len(required)
    if mask_length <= 0:
        raise ValueError("mask length must be positive")
    if required:
        selected = np.setdiff1d(np.arange(num_nodes), np.asarray(required))
        idx = rng.choice(selected, mask_length, replace=False)
    else:
        idx = rng.choice(num_nodes, mask_length, replace=False)
    mask = np.zeros(num_nodes, dtype=bool)
    mask[idx] = True
    mask[required] = True
    return mask


def generate_batch_mask(num_nodes_per_graph: Sequence[int], mask_rate: float, rng: np.random.RandomState,
                        required_idx: Sequence[int] = ()) -> np.ndarray:
    """utils/auxil.py:166-182 (``required_idx`` are node indices INSIDE a graph, the same for every graph of the batch)."""
    return np.hstack([mask_nodes(int(n), mask_rate, rng, required_idx) for n in num_nodes_per_graph])


def make_batch(batch_size: int, num_nodes: int = CTOWN_NODES, num_pipes: int = CTOWN_PIPES,
               topo_seed: int = 0, data_seed: int = 1, mask_seed: int = 2,
               mask_rate: float = 0.95) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """One training batch: (x, y, edge_index, mask) on CPU; x == y before masking (train.py:162-174)."""
    ei = make_wdn_topology(num_nodes, num_pipes, seed=topo_seed)
    edge_index = collate_edge_index(ei, num_nodes, batch_size)
    snaps = make_snapshots(batch_size, num_nodes, seed=data_seed)
    y = collate_snapshots(snaps, range(batch_size))
    mask = torch.from_numpy(generate_batch_mask([num_nodes] * batch_size, mask_rate,
                                                np.random.RandomState(mask_seed)))
    return y.clone(), y, edge_index, mask

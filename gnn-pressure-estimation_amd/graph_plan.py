"""Graph plan (K0): the per-topology CSR structures the HIP kernels walk.

PyG re-derives the self-loop-augmented edge list inside every ``GATConv.forward`` call
(remove_self_loops -> boolean indexing -> device-to-host sync; 30 times per forward of gatres_small,
reference GraphModels.py:464-465).  A WDN batch has the same topology every iteration, so the plan is built
once (``gatres_graph_build_host``) and cached on the content of ``edge_index``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional, Tuple

import torch

from . import _native


class GraphPlan:
    """Device-resident destination-/source-sorted CSR of one (batched) topology."""

    def __init__(self, edge_index: torch.Tensor, num_nodes: int, device=None, segments: bool = True,
                 merge_upto: int = 64, reorder: Optional[bool] = None):
        """``reorder``: relabel the nodes inside every segment with reverse Cuthill-McKee (``gatres_graph_reorder_host``)
        so that the fused kernels' row windows stay compact whatever order the caller's file lists the junctions in.
        ``None`` (default) = automatic: adopted when it shrinks the 4-part row window by more than 10 %; the environment
        variable GATRES_REORDER=0|1 overrides.  x / y / mask / out keep the CALLER's node order either way."""
        if edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise ValueError(f"edge_index must be [2, E], got {tuple(edge_index.shape)}")
        if edge_index.dtype != torch.int64:
            raise ValueError(f"edge_index must be int64 (PyG convention), got {edge_index.dtype}")
        if num_nodes <= 0:
            raise ValueError("num_nodes must be positive")
        device = torch.device(device if device is not None else edge_index.device)
        lib = _native.load()
        ei_host = edge_index.detach().to("cpu").contiguous()
        E = int(ei_host.shape[1])
        N = int(num_nodes)
        e_gat = C.c_int64(0)
        _native.check(lib.gatres_graph_count_host(ei_host.data_ptr(), E, N, C.byref(e_gat)), "gatres_graph_count_host")
        Eg = int(e_gat.value)
        gflags = C.c_int32(0)          # (properties of the edge SET: a relabelling of the nodes does not change them)
        _native.check(lib.gatres_graph_flags_host(ei_host.data_ptr(), E, N, C.byref(gflags)), "gatres_graph_flags_host")
        self.flags = int(gflags.value)
        i32 = dict(dtype=torch.int32)
        # segment table first: it does not depend on the labelling inside a segment, the relabelling below needs it
        self.num_segments, self.max_segment_nodes = 0, 0
        self.max_segment_edges_gat, self.max_segment_edges_mean = 0, 0
        self.perm_host: Optional[torch.Tensor] = None          # int32 [N]: plan node id -> caller's node id
        seg = None
        if segments:
            seg = torch.empty(N + 1, **i32)
            ns, mx, mg, mm = C.c_int32(0), C.c_int32(0), C.c_int32(0), C.c_int32(0)
            _native.check(lib.gatres_graph_segments_host(ei_host.data_ptr(), E, N, int(merge_upto), seg.data_ptr(),
                                                         C.byref(ns), C.byref(mx), C.byref(mg), C.byref(mm)),
                          "gatres_graph_segments_host")
            self.num_segments, self.max_segment_nodes = int(ns.value), int(mx.value)
            self.max_segment_edges_gat, self.max_segment_edges_mean = int(mg.value), int(mm.value)
            self.segment_ptr_host = seg[: self.num_segments + 1].clone()

            def windows_of(ei_t):                 # 7 x 3 window figures + 7 halo sizes (gatres_graph_t.window / .halo)
                win = torch.zeros(28, **i32)
                _native.check(lib.gatres_graph_windows_host(ei_t.data_ptr(), E, N, self.segment_ptr_host.data_ptr(),
                                                            self.num_segments, win.data_ptr()),
                              "gatres_graph_windows_host")
                return [int(v) for v in win]

            self.windows = windows_of(ei_host)
            env = os.environ.get("GATRES_REORDER")
            want = reorder if reorder is not None else (None if env is None else env not in ("0", ""))
            # (automatic only where the fused kernels apply: segments of up to 4096 nodes, include/gatres.h)
            if want is not False and E > 0 and self.max_segment_nodes > 32 and (want or self.max_segment_nodes <= 4096):
                perm = torch.empty(N, **i32)
                _native.check(lib.gatres_graph_reorder_host(ei_host.data_ptr(), E, N, self.segment_ptr_host.data_ptr(),
                                                            self.num_segments, perm.data_ptr()),
                              "gatres_graph_reorder_host")
                old2new = torch.empty(N, dtype=torch.int64)
                old2new[perm.long()] = torch.arange(N, dtype=torch.int64)
                ei_new = old2new[ei_host].contiguous()
                win_new = windows_of(ei_new)
                if want is True or win_new[6] < 0.9 * self.windows[6]:              # (rows of the 4-part window)
                    self.perm_host, self.windows, ei_host = perm, win_new, ei_new
        else:
            self.windows = [0] * 28
        host = {
            "rowptr": torch.empty(N + 1, **i32), "col": torch.empty(Eg, **i32),
            "t_rowptr": torch.empty(N + 1, **i32), "t_eid": torch.empty(Eg, **i32), "t_dst": torch.empty(Eg, **i32),
            "m_rowptr": torch.empty(N + 1, **i32), "m_col": torch.empty(max(E, 1), **i32),
            "mt_rowptr": torch.empty(N + 1, **i32), "mt_dst": torch.empty(max(E, 1), **i32),
        }
        _native.check(lib.gatres_graph_build_host(ei_host.data_ptr(), E, N, *[t.data_ptr() for t in host.values()]),
                      "gatres_graph_build_host")
        self.num_nodes, self.num_edges_gat, self.num_edges_mean = N, Eg, E
        self.device = device
        self._host = host if segments else None          # kept for the part tables (bind)
        self._part_tables: Dict[int, Tuple[torch.Tensor, int]] = {}
        self._bound: Dict[int, "_native.GatresGraph"] = {}          # workgroups per segment -> struct copy with its part tables
        import threading
        self._bind_lock = threading.Lock()
        self.arrays: Dict[str, torch.Tensor] = {k: v.to(device) for k, v in host.items()}
        seg_dev_ptr = None
        if segments:
            self.arrays["seg_ptr"] = self.segment_ptr_host.to(device)
            seg_dev_ptr = self.arrays["seg_ptr"].data_ptr()
        perm_dev_ptr = None
        if self.perm_host is not None:
            self.arrays["perm"] = self.perm_host.to(device)
            perm_dev_ptr = self.arrays["perm"].data_ptr()
        self.c = _native.GatresGraph(N, Eg, E, self.num_segments, *[self.arrays[k].data_ptr() for k in host.keys()],
                                     seg_dev_ptr, self.max_segment_nodes, self.max_segment_edges_gat,
                                     self.max_segment_edges_mean, self.flags, (C.c_int32 * 21)(*self.windows[:21]), 0,
                                     perm_dev_ptr,
                                     (C.c_int32 * 7)(*self.windows[21:28]), 0, None, 0, 0)

    def bound(self, cmodel_ref) -> "_native.GatresGraph":
        """The plan struct WITH the part tables (``gatres_graph_t.part_tables``) for the split the fused kernels use with this
        model: built once per (plan, workgroups per segment) on the host by ``gatres_graph_part_tables_host`` and kept on
        the device.  Every split has its OWN immutable copy of the struct (binding used to rewrite the one shared
        struct on every forward / backward -- a data race for a cached plan used by two modules or threads); ``self.c`` is
        never modified.  Without tables (GATRES_NO_PART_TABLES=1, unsplit or oversized segments) the window kernel derives
        the same tables in every launch: the two must agree, tests/test_gpu_model.py."""
        if self._host is None or self.num_segments <= 0 or os.environ.get("GATRES_NO_PART_TABLES"):
            return self.c
        lib = _native.load()
        m = int(lib.gatres_fused_cus_per_segment(cmodel_ref, C.byref(self.c)))
        if m < 2 or m > 8 or self.max_segment_nodes > 65535:
            return self.c
        with self._bind_lock:
            got = self._bound.get(m)
            if got is not None:
                return got
            ptrs = [self._host[k].data_ptr() for k in ("rowptr", "col", "t_rowptr", "t_eid", "t_dst", "m_rowptr", "m_col",
                                                       "mt_rowptr", "mt_dst")]
            stride = C.c_int64(0)
            rc = lib.gatres_graph_part_tables_host(*ptrs, self.segment_ptr_host.data_ptr(), self.num_segments, m, None, 0,
                                                   C.byref(stride))
            if rc != 0:                  # (segments beyond the 16-bit tables: the kernels do not take them either)
                self._bound[m] = self.c
                return self.c
            words = torch.zeros(self.num_segments * m * int(stride.value), dtype=torch.int32)
            _native.check(lib.gatres_graph_part_tables_host(*ptrs, self.segment_ptr_host.data_ptr(), self.num_segments, m,
                                                            words.data_ptr(), int(stride.value), C.byref(stride)),
                          "gatres_graph_part_tables_host")
            dev = words.to(self.device)
            self._part_tables[m] = (dev, int(stride.value))
            c = _native.GatresGraph.from_buffer_copy(self.c)
            c.part_tables, c.part_tables_m, c.part_tables_stride = dev.data_ptr(), m, int(stride.value)
            self._bound[m] = c
            return c

    def bind(self, cmodel_ref) -> None:
        """Build (once) the part tables this model's split needs; ``ref(cmodel_ref)`` then hands out the struct that has them."""
        self.bound(cmodel_ref)

    def ref(self, cmodel_ref=None):
        """``gatres_graph_t*`` for a C call: the bare plan, or -- given the model -- the copy that carries the part tables
        of that model's split."""
        return C.byref(self.c if cmodel_ref is None else self.bound(cmodel_ref))

    def window_rows(self, parts: int) -> int:
        """Rows of the largest part window when a segment is carried by ``parts`` (2 .. 8) workgroups; 0 = unknown."""
        return self.windows[3 * (parts - 2)] if 2 <= parts <= 8 else 0

    def halo_edges(self, parts: int) -> int:
        return self.windows[21 + parts - 2] if 2 <= parts <= 8 else 0

    def node_ptr_for(self, nodes_per_graph) -> torch.Tensor:
        """int32 [B+1] node offsets for the device mask sampler."""
        off = [0]
        for n in nodes_per_graph:
            off.append(off[-1] + int(n))
        if off[-1] != self.num_nodes:
            raise ValueError("nodes_per_graph does not add up to num_nodes")
        return torch.tensor(off, dtype=torch.int32, device=self.device)


class PlanCache:
    """edge_index -> GraphPlan.  Key = (device content hash, E, N); the hash is one small kernel plus an 8-byte
    read-back, against PyG's ~30 boolean-index syncs per forward.  A tensor object already seen (same storage,
    same version counter) skips the hash."""

    def __init__(self, max_entries: int = 16, segments: bool = True):
        self.max_entries = max_entries
        self.segments = segments
        self._by_hash: Dict[Tuple[int, int, int, str], GraphPlan] = {}
        self._by_identity: Dict[Tuple[int, int, int, int, str], GraphPlan] = {}

    def clear(self) -> None:
        self._by_hash.clear()
        self._by_identity.clear()

    def get(self, edge_index: torch.Tensor, num_nodes: int) -> GraphPlan:
        _native.require_gpu_tensor(edge_index, "edge_index", torch.int64)
        ident = (edge_index.data_ptr(), edge_index._version, int(edge_index.shape[1]), int(num_nodes),
                 str(edge_index.device))
        plan = self._by_identity.get(ident)
        if plan is not None and getattr(plan, "_src_ref", None) is edge_index:
            return plan
        lib = _native.load()
        h = torch.zeros(1, dtype=torch.int64, device=edge_index.device)
        _native.check(lib.gatres_edge_index_hash(edge_index.data_ptr(), int(edge_index.shape[1]), h.data_ptr(),
                                                 _native.current_stream(edge_index.device)), "gatres_edge_index_hash")
        key = (int(h.item()), int(edge_index.shape[1]), int(num_nodes), str(edge_index.device))
        plan = self._by_hash.get(key)
        if plan is None:
            if len(self._by_hash) >= self.max_entries:
                self.clear()
            plan = GraphPlan(edge_index, num_nodes, segments=self.segments)
            self._by_hash[key] = plan
        plan._src_ref = edge_index        # identity fast path is valid only while this exact tensor is alive
        self._by_identity = {ident: plan}
        return plan

"""Native training step for GATRes: the reference's inner loop body (train.py:159-190) as one stream-ordered
launch sequence -- device mask sampling (auxil.py:143-182), ``x[mask] = 0``, forward, MSE on the masked nodes,
backward, [RCCL gradient all-reduce], Adam(lr, weight_decay) -- captured once into a hipGraph and replayed.

Data-parallel: one process per GPU, snapshots sharded by graph across ranks, every rank holds the same topology
plan and an identical replica of the flat parameter vector; the only exchange is ONE all-reduce of the flat fp32
gradient per step (RCCL over xGMI; `torch.distributed` backend "nccl"), after which each rank applies the same Adam
update with the gradient scaled by 1/world_size.  Equal masked-node counts per graph make the average of per-rank
mean-losses equal the global-batch mean loss, so the result matches single-process training on the global batch.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence

import torch

from . import _native
from .graph_models import GATResMeanConv
from .graph_plan import GraphPlan

PHASE_MASK, PHASE_FORWARD, PHASE_BACKWARD, PHASE_ADAM = 1, 2, 4, 8


class _TrainStepC(C.Structure):
    """gatres_train_step_t"""
    _fields_ = [("model", _native.GatresModel), ("graph", C.POINTER(_native.GatresGraph)),
                ("params", C.c_void_p), ("grads", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p),
                ("step_counter", C.c_void_p), ("x", C.c_void_p), ("y", C.c_void_p), ("mask", C.c_void_p),
                ("node_ptr", C.c_void_p), ("num_graphs", C.c_int32), ("phases", C.c_int32),
                ("mask_rate", C.c_double), ("seed", C.c_uint64),
                ("out", C.c_void_p), ("g_out", C.c_void_p), ("loss", C.c_void_p), ("saved", C.c_void_p),
                ("scratch", C.c_void_p),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("weight_decay", C.c_double), ("grad_scale", C.c_float), ("flags", C.c_int32)]


class GATResTrainer:
    """Owns the static buffers of one (model, topology, batch size) and runs training steps on them.

    ``step(x, y)`` draws a fresh device-side mask; ``step(x, y, mask=...)`` uses the caller's mask (bool/uint8 [N]),
    e.g. one produced by the reference's host sampler, for bit-for-bit comparable runs."""

    def __init__(self, model: GATResMeanConv, edge_index: torch.Tensor, num_nodes: int,
                 nodes_per_graph: Optional[Sequence[int]] = None, lr: float = 5e-4, weight_decay: float = 6e-6,
                 betas=(0.9, 0.999), eps: float = 1e-8, mask_rate: float = 0.95, seed: int = 0,
                 process_group=None, use_graph: bool = True, fused: bool = True,
                 force_collective_path: bool = False, targets_are_inputs: bool = False):
        self.lib = _native.load()
        self.model = model
        params = model.flat_parameters
        if not params.is_cuda:
            raise ValueError("move the model to the ROCm device first; this engine has no CPU path")
        dev = params.device
        self.device = dev
        self.plan: GraphPlan = GraphPlan(edge_index, num_nodes, device=dev, segments=fused)
        self.fused = bool(fused and self.lib.gatres_fused_supported(model._cmodel_ref(), self.plan.ref()))
        N = num_nodes
        self.N = N
        self.P = params.numel()
        f32 = dict(dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.P, **f32)
        self.exp_avg = torch.zeros(self.P, **f32)
        self.exp_avg_sq = torch.zeros(self.P, **f32)
        self.step_counter = torch.zeros(2, dtype=torch.int64, device=dev)
        self.x = torch.zeros(N, **f32)
        # targets_are_inputs: the reference's snapshots have data.y == data.x until the loop masks x (train.py:162-166),
        # and the masking happens inside the kernels here (x itself is never overwritten): one buffer serves both
        self.targets_are_inputs = bool(targets_are_inputs)
        self.y = self.x if self.targets_are_inputs else torch.zeros(N, **f32)
        self.mask = torch.zeros(N, dtype=torch.uint8, device=dev)
        self.out = torch.zeros(N, **f32)
        self.g_out = torch.zeros(N, **f32)
        self.loss = torch.zeros(1, **f32)
        self.saved = torch.empty(model._saved_floats(self.plan), **f32)
        self.scratch = model._scratch_for(self.plan)
        self.node_ptr = self.plan.node_ptr_for(nodes_per_graph) if nodes_per_graph is not None else None
        self.num_graphs = len(nodes_per_graph) if nodes_per_graph is not None else 0
        self.hparams = dict(lr=lr, weight_decay=weight_decay, beta1=betas[0], beta2=betas[1], eps=eps)
        self.mask_rate, self.seed = mask_rate, seed
        self.pg = process_group
        self.world = 1
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
        # force_collective_path: run the multi-GPU sequence (backward | all-reduce | Adam) even at world size 1
        self.split = force_collective_path or self.world > 1
        self.use_graph = use_graph
        self._graphs = {}
        self._params_ptr = params.data_ptr()
        self._wt_version = None

    # ------------------------------------------------------------------------------------------
    def _desc(self, phases: int, device_mask: bool, wt_valid: bool = False) -> _TrainStepC:
        m = self.model
        params = m.flat_parameters
        if params.data_ptr() != self._params_ptr:
            raise RuntimeError("the model's parameter storage moved (e.g. .to()/deepcopy); build a new GATResTrainer")
        h = self.hparams
        return _TrainStepC(
            _native.GatresModel(m.num_blocks, m.nc), C.pointer(self.plan.c), params.data_ptr(), self.grads.data_ptr(),
            self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.step_counter.data_ptr(), self.x.data_ptr(),
            self.y.data_ptr(), self.mask.data_ptr(),
            self.node_ptr.data_ptr() if (device_mask and self.node_ptr is not None) else None,
            self.num_graphs, phases, self.mask_rate, self.seed, self.out.data_ptr(), self.g_out.data_ptr(),
            self.loss.data_ptr(), self.saved.data_ptr(), self.scratch.data_ptr(), h["lr"], h["beta1"], h["beta2"],
            h["eps"], h["weight_decay"], 1.0 / self.world, (0 if self.fused else 1) | (2 if wt_valid else 0))

    def _wt_current(self) -> bool:
        """scratch's transposed conv weights are current: our last fused Adam step wrote them and nothing else has
        modified the parameter storage since (torch bumps the version counter on every in-place op; our kernels do
        not go through torch)."""
        return self._wt_version is not None and self._wt_version == self.model.flat_parameters._version

    def _enqueue(self, phases: int, device_mask: bool, wt_valid: bool = False) -> None:
        ts = self._desc(phases, device_mask, wt_valid)
        _native.check(self.lib.gatres_train_step(C.byref(ts), _native.current_stream(self.device)),
                      "gatres_train_step")

    def _run(self, phases: int, device_mask: bool) -> None:
        full = PHASE_BACKWARD | PHASE_ADAM
        wt_valid = self.fused and (phases & PHASE_BACKWARD) != 0 and self._wt_current()
        try:
            self._run_inner(phases, device_mask, wt_valid)
        finally:
            # a fused backward + Adam step leaves the transposed weights of the NEW parameters in scratch
            if self.fused and (phases & full) == full:
                self._wt_version = self.model.flat_parameters._version

    def _run_inner(self, phases: int, device_mask: bool, wt_valid: bool) -> None:
        if not self.use_graph:
            self._enqueue(phases, device_mask, wt_valid)
            return
        key = (phases, device_mask, wt_valid)
        g = self._graphs.get(key)
        if g is None:
            # warm-up launch outside capture (module load, lazy init), then capture the same sequence once
            state = (self.step_counter.clone(), self.model.flat_parameters.clone(), self.exp_avg.clone(),
                     self.exp_avg_sq.clone())
            self._enqueue(phases, device_mask, wt_valid)
            torch.cuda.synchronize(self.device)
            self.step_counter.copy_(state[0]); self.model.flat_parameters.copy_(state[1])
            self.exp_avg.copy_(state[2]); self.exp_avg_sq.copy_(state[3])
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._enqueue(phases, device_mask, wt_valid)
            self.step_counter.copy_(state[0]); self.model.flat_parameters.copy_(state[1])
            self.exp_avg.copy_(state[2]); self.exp_avg_sq.copy_(state[3])
            if wt_valid:
                # the warm-up step left the transposes of ITS updated weights in scratch; the parameters were rolled
                # back, so make the promise this graph relies on true again
                _native.check(self.lib.gatres_fused_prepare_backward(
                    self.model._cmodel_ref(), self.plan.ref(), self.model.flat_parameters.data_ptr(),
                    self.scratch.data_ptr(), _native.current_stream(self.device)), "gatres_fused_prepare_backward")
            self._graphs[key] = g
        g.replay()

    # ------------------------------------------------------------------------------------------
    def load_batch(self, x: torch.Tensor, y: torch.Tensor, mask: Optional[torch.Tensor] = None) -> None:
        """Stage a batch into the static buffers (async copies on the current stream)."""
        self.x.copy_(x.reshape(-1), non_blocking=True)
        if not self.targets_are_inputs:
            self.y.copy_(y.reshape(-1), non_blocking=True)
        if mask is not None:
            self.mask.copy_(mask.reshape(-1).to(torch.uint8), non_blocking=True)

    def prefetch_batch(self, x: torch.Tensor, y: Optional[torch.Tensor] = None) -> None:
        """Start copying the NEXT batch (typically pinned host memory) into a staging buffer on a side stream, so the
        PCIe transfer overlaps the step that is running; ``commit_batch()`` then moves it into place with a device copy.
        Two staging slots: at most one prefetch may be outstanding per ``commit_batch()``."""
        if not hasattr(self, "_stage"):
            f32 = dict(dtype=torch.float32, device=self.device)
            n = self.x.numel()
            self._stage = [(torch.empty(n, **f32), None if self.targets_are_inputs else torch.empty(n, **f32))
                           for _ in range(2)]
            self._copy_stream = torch.cuda.Stream(device=self.device)
            self._ready = [torch.cuda.Event() for _ in range(2)]
            self._free = [torch.cuda.Event() for _ in range(2)]
            for ev in self._free:
                ev.record(torch.cuda.current_stream(self.device))
            self._slot_in, self._slot_out = 0, 0
        k = self._slot_in
        sx, sy = self._stage[k]
        with torch.cuda.stream(self._copy_stream):
            self._copy_stream.wait_event(self._free[k])
            sx.copy_(x.reshape(-1), non_blocking=True)
            if sy is not None:
                sy.copy_((x if y is None else y).reshape(-1), non_blocking=True)
            self._ready[k].record(self._copy_stream)
        self._slot_in = 1 - k

    def commit_batch(self) -> None:
        """Make the prefetched batch the current one (device-to-device copies on the compute stream)."""
        k = self._slot_out
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self._ready[k])
        sx, sy = self._stage[k]
        self.x.copy_(sx, non_blocking=True)
        if sy is not None:
            self.y.copy_(sy, non_blocking=True)
        self._free[k].record(cur)
        self._slot_out = 1 - k

    def run_step(self, device_mask: bool = True) -> None:
        """One optimisation step on the staged batch.  Nothing is synchronised; read ``self.loss`` afterwards."""
        if device_mask and self.node_ptr is None:
            raise ValueError("device mask sampling needs nodes_per_graph at construction")
        if not self.split:
            self._run(PHASE_MASK | PHASE_FORWARD | PHASE_BACKWARD | PHASE_ADAM, device_mask)
        else:
            self._run(PHASE_MASK | PHASE_FORWARD | PHASE_BACKWARD, device_mask)
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                torch.distributed.all_reduce(self.grads, op=torch.distributed.ReduceOp.SUM, group=self.pg)
            self._run(PHASE_ADAM, device_mask)

    def step(self, x: torch.Tensor, y: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Reference-shaped call: one iteration of train.py:159-190.  Returns the (device) loss tensor."""
        self.load_batch(x, y, mask)
        self.run_step(device_mask=mask is None)
        return self.loss

    def forward_backward(self, x: torch.Tensor, y: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """Forward + loss + gradients only (no optimiser update); gradients land in ``self.grads``."""
        self.load_batch(x, y, mask)
        self._enqueue(PHASE_FORWARD | PHASE_BACKWARD, False)
        return self.loss

    @property
    def optimizer_step(self) -> int:
        return int(self.step_counter[0].item())

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
import os
from collections import OrderedDict
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch

from . import _native, dp
from .graph_models import GATResMeanConv
from .graph_plan import GraphPlan

PHASE_MASK, PHASE_FORWARD, PHASE_BACKWARD, PHASE_ADAM = 1, 2, 4, 8
PART_FIRST, PART_LAST, PART_REDUCE = 1, 2, 4
FLAG_PER_OP, FLAG_WT_VALID, FLAG_GRADS_DEFERRED, FLAG_GRADS_ONLY, FLAG_MASK_NEXT = 1, 2, 4, 8, 16
MAX_CACHED_GRAPHS = 40          # floor of the captured-step cache; bind_batches() sizes it from the number of bound batches


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
                ("weight_decay", C.c_double), ("grad_scale", C.c_float), ("flags", C.c_int32),
                ("hparams", C.c_void_p), ("block_lo", C.c_int32), ("block_hi", C.c_int32), ("mask_next", C.c_void_p)]


class GATResTrainer:
    """Owns the static buffers of one (model, topology, batch size) and runs training steps on them.

    ``step(x, y)`` draws a fresh device-side mask; ``step(x, y, mask=...)`` uses the caller's mask (bool/uint8 [N]),
    e.g. one produced by the reference's host sampler, for bit-for-bit comparable runs."""

    def __init__(self, model: GATResMeanConv, edge_index: torch.Tensor, num_nodes: int,
                 nodes_per_graph: Optional[Sequence[int]] = None, lr: float = 5e-4, weight_decay: float = 6e-6,
                 betas=(0.9, 0.999), eps: float = 1e-8, mask_rate: float = 0.95, seed: int = 0,
                 process_group=None, use_graph: bool = True, fused: bool = True,
                 force_collective_path: bool = False, targets_are_inputs: bool = False,
                 blocks_per_bucket: int = 5, _share_state_with: Optional["GATResTrainer"] = None):
        self.lib = _native.load()
        self.model = model
        params = model.flat_parameters
        if not params.is_cuda:
            raise ValueError("move the model to the ROCm device first; this engine has no CPU path")
        dev = params.device
        self.device = dev
        self.plan: GraphPlan = GraphPlan(edge_index, num_nodes, device=dev, segments=fused)
        self.fused = bool(fused and self.lib.gatres_fused_supported(model._cmodel_ref(), self.plan.ref()))
        # the plan struct this trainer's launches take: with the part tables of this model's split on the fused path
        self._gstruct = self.plan.bound(model._cmodel_ref()) if self.fused else self.plan.c
        N = num_nodes
        self.N = N
        self.P = params.numel()
        f32 = dict(dtype=torch.float32, device=dev)
        self.grads = torch.zeros(self.P, **f32)
        self.hparams = dict(lr=lr, weight_decay=weight_decay, beta1=betas[0], beta2=betas[1], eps=eps)
        if _share_state_with is not None:            # a sibling trainer for another batch size: ONE optimizer state
            self.exp_avg, self.exp_avg_sq = _share_state_with.exp_avg, _share_state_with.exp_avg_sq
            self.step_counter = _share_state_with.step_counter
            self.hp = _share_state_with.hp
            # ... and ONE host copy of the hyper-parameters beside the one device buffer: set_hparams() on either trainer
            # is seen by both (a sibling's set_lr left the parent's host copy stale, and the parent's next
            # set_hparams(weight_decay=...) pushed the old lr back)
            self.hparams = _share_state_with.hparams
        else:
            self.exp_avg = torch.zeros(self.P, **f32)
            self.exp_avg_sq = torch.zeros(self.P, **f32)
            self.step_counter = torch.zeros(2, dtype=torch.int64, device=dev)
            # the hyper-parameters the update kernels read (gatres_train_step_t.hparams): a captured step follows
            # set_lr() without being captured again
            self.hp = torch.tensor(self._hp_list(), dtype=torch.float64, device=dev)
        self.x = torch.zeros(N, **f32)
        # targets_are_inputs: the reference's snapshots have data.y == data.x until the loop masks x (train.py:162-166),
        # and the masking happens inside the kernels here (x itself is never overwritten): one buffer serves both
        self.targets_are_inputs = bool(targets_are_inputs)
        self.y = self.x if self.targets_are_inputs else torch.zeros(N, **f32)
        self.mask = torch.zeros(N, dtype=torch.uint8, device=dev)         # the mask of the step that ran last / is staged
        self._mask_spare = torch.zeros(N, dtype=torch.uint8, device=dev)  # where the update launch samples the NEXT step's mask
        self.out = torch.zeros(N, **f32)
        self.g_out = torch.zeros(N, **f32)
        self.loss = torch.zeros(1, **f32)
        self.saved = torch.empty(model._saved_floats(self.plan), **f32)
        self.scratch = model._scratch_for(self.plan)
        self.node_ptr = self.plan.node_ptr_for(nodes_per_graph) if nodes_per_graph is not None else None
        self.num_graphs = len(nodes_per_graph) if nodes_per_graph is not None else 0
        self.pg = process_group
        self.world, self.rank = 1, 0
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)
        # every rank draws its own masks (its shard holds different snapshots): the rank is mixed into the sampler's seed
        self.mask_rate, self.seed = mask_rate, int(seed) * max(self.world, 1) + self.rank
        # force_collective_path: run the multi-GPU sequence (backward | all-reduce | Adam) even at world size 1
        self.split = force_collective_path or self.world > 1
        if self.world > 1 and _share_state_with is None:
            dp.broadcast_params_(params, src=0, group=process_group)      # identical replicas, whatever each rank initialised
        self.reducer = dp.BucketedAllReduce(self.grads, process_group, force=force_collective_path) if self.split else None
        self.blocks_per_bucket = int(blocks_per_bucket)
        # Gradient buckets of the fused path's data-parallel step.  1 (default): the single-GPU step's own launches (window
        # kernel, parameter gradients, slab sum) + ONE all-reduce of the flat gradient + the Adam launch.  2: the backward chain,
        # then the parameter gradients of the upper and of the lower blocks as range launches, the first bucket on the wire
        # under the second launch (DESIGN.md section 5 has the measured cost of either form on one GPU).
        self.fused_buckets = 1
        self.epoch_graph_steps = 20          # fit_epoch: full batches per captured launch sequence; 1 = step by step
        self.epoch_copy_limit_bytes = 16 << 30   # fit_epoch trains on a shuffled device copy of the store up to this size (else: steps_rows)
        self._max_graphs = MAX_CACHED_GRAPHS
        # Multi-rank steps run as EAGER launch sequences unless GATRES_DP_GRAPH=1: a captured step would hold the RCCL
        # all-reduce, which has only ever been captured with a one-rank group here (no multi-GPU box is reachable), and
        # eager costs nothing on this path (one-rank nccl group, gatres_small bs 32: 0.428 ms/step eager vs 0.432 captured;
        # the step is a handful of native launch sequences, far from host-bound).  One-rank groups (force_collective_path)
        # keep the capture, so the in-graph form stays tested.
        self.use_graph = use_graph and (self.world <= 1 or os.environ.get("GATRES_DP_GRAPH") == "1")
        self._graphs: "OrderedDict[tuple, torch.cuda.CUDAGraph]" = OrderedDict()
        self._params_ptr = params.data_ptr()
        self._wt_sig = None
        self._siblings: Dict[int, "GATResTrainer"] = {}
        off = self.lib.gatres_fused_status_offset(model._cmodel_ref(), self.plan.ref()) if self.fused else -1
        self._status = self.scratch[off:off + 4].view(torch.int32) if off >= 0 else None
        # Single-GPU fused step: the update launch samples the NEXT step's device mask into the spare buffer
        # (GATRES_FLAG_MASK_NEXT); a step that finds it still valid swaps the buffers and leaves the sampler's launch out.
        # self.mask always is the mask of the step that ran last.  _mask_sig: what the spare buffer's mask was sampled for.
        # (only where the parameter gradients are a launch of their own: it leaves the step count / fault snapshot the
        #  sampling tail reads -- batches that leave CUs free use consumer workgroups and keep the sampler's own launch)
        # (the data-parallel step -- one gradient bucket -- does the same from its Adam launch, gatres_adam_step_ex)
        self._mask_next = bool(self.fused and self.node_ptr is not None
                               and self.lib.gatres_fused_finish_folds(model._cmodel_ref(), self.plan.ref())
                               and not os.environ.get("GATRES_NO_MASK_NEXT"))
        self._mask_sig = None
        self._bound: list = []                      # bind_batches(): (x, y) tensors the captured steps read directly
        # the fused path's parameter gradients can be formed range by range (a bucket's all-reduce starts in between)
        self._ranges = bool(self.fused and self.lib.gatres_fused_finish_folds(model._cmodel_ref(), self.plan.ref()))
        self._faults_seen = 0
        self._warn_if_the_chip_is_not_ours()

    def _warn_if_the_chip_is_not_ours(self) -> None:
        """A split launch (several workgroups per snapshot, spin-waiting on each other's granules) needs EVERY workgroup of
        its grid resident at once, one per CU.  The native library sizes the split from the device's CU count; what it cannot
        see is a CU mask or other work on the device.  A partner that never gets a CU costs a DROPPED step (counted in
        ``fault_count``), never a wrong one -- but a trainer that silently skips updates is not what anyone wants: say so at
        construction, and let ``check_no_dropped_steps()`` (called by ``fit_epoch`` and by bench.py) turn drops into an error."""
        if not self.fused:
            return
        cus = int(self.lib.gatres_fused_cus_per_segment(self.model._cmodel_ref(), self.plan.ref()))
        if cus < 2:
            return
        padded = ((self.plan.num_segments + 7) // 8) * 8
        need = padded * cus
        have = int(torch.cuda.get_device_properties(self.device).multi_processor_count)
        if need > have and ((padded // 2 + 7) // 8) * 8 * cus <= have:
            need = ((padded // 2 + 7) // 8) * 8 * cus       # (batches of 49 - 96 snapshots go in two rounds of half the segments)
        masked = [k for k in ("HSA_CU_MASK", "ROC_GLOBAL_CU_MASK") if os.environ.get(k)]
        if need > have or masked:
            import warnings
            warnings.warn(f"GATResTrainer: the per-snapshot kernel launches {need} co-resident workgroups (one per CU, {cus} per "
                          f"snapshot) on a device that reports {have} CUs"
                          + (f" under {', '.join(masked)}" if masked else "") +
                          "; workgroups that cannot be resident together time out and the step is DROPPED (fault_count). "
                          "Set GATRES_FUSED_SPLIT to fewer parts per snapshot or use fused=False.", RuntimeWarning, stacklevel=3)

    def check_no_dropped_steps(self) -> None:
        """Raise if any step was dropped since the last call (a split launch that timed out waiting for a partner workgroup,
        or a data-parallel step whose all-reduced gradient carried a fault mark).  Synchronises."""
        n = self.fault_count + self.dropped_steps
        new, self._faults_seen = n - self._faults_seen, n
        if new > 0:
            raise RuntimeError(f"{new} training step(s) were DROPPED (no update): a workgroup of the per-snapshot kernel was not "
                               f"co-resident with its partners (is the GPU shared, partitioned or CU-masked?) or a replica "
                               f"faulted; fault_count={self.fault_count}, dropped_steps={self.dropped_steps}")

    # ------------------------------------------------------------------------------------------
    def _hp_list(self):
        h = self.hparams
        return [h["lr"], h["beta1"], h["beta2"], h["eps"], h["weight_decay"]]

    def _push_hparams(self) -> None:
        """Host values -> the device buffer (one small stream-ordered copy; never inside a capture)."""
        self.hp.copy_(torch.tensor(self._hp_list(), dtype=torch.float64), non_blocking=False)

    def _desc(self, phases: int, device_mask: bool, wt_valid: bool = False, flags: int = 0, block_lo: int = 0,
              block_hi: int = 0, batch=None, masks=None, loss=None) -> _TrainStepC:
        m = self.model
        bx, by = (self.x, self.y) if batch is None else batch
        mask, mask_next = (self.mask, self._mask_spare) if masks is None else masks
        params = m.flat_parameters
        if params.data_ptr() != self._params_ptr:
            raise RuntimeError("the model's parameter storage moved (e.g. .to()/deepcopy); build a new GATResTrainer")
        h = self.hparams
        return _TrainStepC(
            _native.GatresModel(m.num_blocks, m.nc, m._cmodel.act_dtype, 0), C.pointer(self._gstruct), params.data_ptr(), self.grads.data_ptr(),
            self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.step_counter.data_ptr(), bx.data_ptr(),
            by.data_ptr(), mask.data_ptr(),
            self.node_ptr.data_ptr() if (device_mask and self.node_ptr is not None) else None,
            self.num_graphs, phases, self.mask_rate, self.seed, self.out.data_ptr(), self.g_out.data_ptr(),
            (self.loss if loss is None else loss).data_ptr(), self.saved.data_ptr(), self.scratch.data_ptr(), h["lr"], h["beta1"], h["beta2"],
            h["eps"], h["weight_decay"], 1.0 / self.world,
            (0 if self.fused else FLAG_PER_OP) | (FLAG_WT_VALID if wt_valid else 0) | flags, self.hp.data_ptr(),
            block_lo, block_hi, mask_next.data_ptr())

    def _count_native_update(self) -> None:
        """A native kernel (fused Adam pass, Adam-only phase) just changed the parameters without touching torch's
        version counters.  The count lives on the MODEL: trainers that share it (``fit_epoch``'s sibling for the ragged
        last batch, a trainer per batch size) each keep transposed conv weights in their OWN scratch buffer, and a step
        of one must invalidate the copies of the others (the sibling's step left the main trainer's W^T one
        update behind while its signature still matched)."""
        self.model._native_updates = getattr(self.model, "_native_updates", 0) + 1

    def _param_signature(self) -> Tuple[int, int, int]:
        """Changes whenever anything wrote to the parameters THROUGH TORCH: the flat vector or any of the nn.Parameters
        that are views of it (every Parameter has its own version counter: ``load_state_dict``, ``torch.optim`` steps,
        ``init_`` / ``clip_`` calls bump those, not the flat vector's).  Writes that bypass autograd's bookkeeping
        (``p.data.mul_()``) are invisible to it: call ``invalidate_weights()`` after such an edit."""
        return (self.model.flat_parameters._version, sum(p._version for p in self.model._param_list),
                getattr(self.model, "_native_updates", 0))

    def _wt_current(self) -> bool:
        """scratch's transposed conv weights are current: our last fused Adam step wrote them and nothing has modified
        the parameters since (our kernels do not go through torch, so they leave the signature alone)."""
        return self._wt_sig is not None and self._wt_sig == self._param_signature()

    def invalidate_weights(self) -> None:
        """The parameters were edited behind torch's back: re-derive the transposed weights at the next step."""
        self._wt_sig = None

    def set_lr(self, lr: float) -> None:
        """Learning-rate schedules (the reference runs ReduceLROnPlateau, train.py:349-350,510).  The update kernels read
        the hyper-parameters from a device buffer (``gatres_train_step_t.hparams``), so the captured step is NOT captured
        again: the new value is one 40-byte copy in stream order."""
        self.set_hparams(lr=lr)

    def set_hparams(self, **kw) -> None:
        changed = False
        for k, v in kw.items():
            if k == "mask_rate":
                self.mask_rate = float(v)
            elif k in self.hparams:
                changed = changed or self.hparams[k] != float(v)
                self.hparams[k] = float(v)
            else:
                raise KeyError(k)
        if changed:
            self._push_hparams()                       # (siblings share both the dict and the device buffer)

    def _graph_key(self, what, device_mask: bool, wt_valid: bool) -> tuple:
        # (no hyper-parameter in here: the kernels read them from self.hp; which of the two mask buffers is current is)
        return (what, device_mask, wt_valid, self.mask_rate, self.seed, self.world, self.mask.data_ptr())

    def _take_mask_ahead(self) -> bool:
        """If the previous step's update launch sampled THIS step's mask and nothing has moved since, make that buffer the
        current one (True); else the caller samples / stages a mask into self.mask as before."""
        if self._mask_next and self._mask_sig is not None and self._mask_sig == self._mask_key():
            self.mask, self._mask_spare = self._mask_spare, self.mask
            self._mask_sig = None
            return True
        self._mask_sig = None
        return False

    @property
    def num_captured_graphs(self) -> int:
        return len(self._graphs)

    @property
    def dropped_steps(self) -> int:
        """Data-parallel steps whose update was skipped because the all-reduced gradient carried a fault mark -- the mark is
        "entry 0 is NaN", so a diverging run that puts a NaN there shows up here too (it would otherwise freeze silently:
        ADVICE r3).  Synchronises."""
        return int(self._status[3].item()) if self._status is not None else 0

    @property
    def fault_count(self) -> int:
        """Split launches that gave up waiting for a partner workgroup so far (each cost exactly one dropped step: loss
        NaN, no update).  Synchronises."""
        return int(self._status[1].item()) if self._status is not None else 0

    def _enqueue(self, phases: int, device_mask: bool, wt_valid: bool = False, flags: int = 0, block_lo: int = 0,
                 block_hi: int = 0, batch=None, masks=None, loss=None) -> None:
        ts = self._desc(phases, device_mask, wt_valid, flags, block_lo, block_hi, batch, masks, loss)
        _native.check(self.lib.gatres_train_step(C.byref(ts), _native.current_stream(self.device)),
                      "gatres_train_step")

    def _mask_key(self):
        """What a mask sampled by the update launch is valid for: the optimizer state as this trainer left it (a sibling's or a
        FusedAdam step moves the shared step count: the sampler's key would differ) and the sampler's own settings."""
        return (getattr(self.model, "_native_updates", 0), self.mask_rate, self.seed)

    def _run(self, phases: int, device_mask: bool, batch=None, slot=None) -> None:
        full = PHASE_BACKWARD | PHASE_ADAM
        wt_valid = self.fused and (phases & PHASE_BACKWARD) != 0 and self._wt_current()
        flags = 0
        mask_next = self._mask_next and device_mask and (phases & (PHASE_FORWARD | full)) == (PHASE_FORWARD | full)
        if mask_next:
            flags |= FLAG_MASK_NEXT
            if (phases & PHASE_MASK) and self._take_mask_ahead():
                phases &= ~PHASE_MASK            # the previous step's update launch has sampled this step's mask already
        try:
            self._replay(self._graph_key((phases, flags, slot), device_mask, wt_valid),
                         lambda: self._enqueue(phases, device_mask, wt_valid, flags=flags, batch=batch), wt_valid)
        finally:
            if phases & PHASE_ADAM:
                self._count_native_update()
            # a fused backward + Adam step leaves the transposed weights of the NEW parameters in (this trainer's) scratch
            if self.fused and (phases & full) == full:
                self._wt_sig = self._param_signature()
            self._mask_sig = self._mask_key() if mask_next else None

    def _replay(self, key: tuple, enqueue: Callable[[], None], wt_valid: bool = False) -> None:
        """Run ``enqueue`` -- eagerly, or captured once into a hipGraph (cached by ``key``) and replayed."""
        if not self.use_graph:
            enqueue()
            return
        g = self._graphs.get(key)
        if g is None:
            # warm-up launch outside capture (module load, lazy init, RCCL communicator), then capture the same sequence
            state = (self.step_counter.clone(), self.model.flat_parameters.clone(), self.exp_avg.clone(),
                     self.exp_avg_sq.clone(), self.mask.clone(), self._mask_spare.clone())

            def rollback():
                self.step_counter.copy_(state[0]); self.model.flat_parameters.copy_(state[1])
                self.exp_avg.copy_(state[2]); self.exp_avg_sq.copy_(state[3])
                self.mask.copy_(state[4])        # (a warm-up that ran the sampler's launch re-sampled it: same bits; kept simple)
                self._mask_spare.copy_(state[5])  # (a multi-step sequence samples into BOTH buffers, two steps ahead of the rollback)

            enqueue()
            torch.cuda.synchronize(self.device)
            rollback()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    enqueue()
            except Exception as e:                     # noqa: BLE001
                # A multi-rank step holds RCCL collectives; if this runtime cannot capture them (every rank fails the
                # same way: the sequence is identical), run the launch sequence eagerly from here on instead of dying.
                # Single-rank captures hold only this library's kernels: a failure there is a bug and is raised.
                if self.world <= 1:
                    raise
                import warnings
                warnings.warn(f"hipGraph capture of the data-parallel step failed ({type(e).__name__}: {e}); "
                              f"continuing with eager launches")
                try:                                   # (a failed capture may leave the stream in capture mode: end it)
                    if torch.cuda.is_current_stream_capturing():
                        g.capture_end()
                except Exception:                      # noqa: BLE001
                    pass
                torch.cuda.synchronize(self.device)
                rollback()
                if wt_valid:
                    # the warm-up step left the transposes of ITS updated weights in scratch and the parameters were
                    # rolled back: re-derive them, or this eager step's backward would use W^T of other weights
                    _native.check(self.lib.gatres_fused_prepare_backward(
                        self.model._cmodel_ref(), C.byref(self._gstruct), self.model.flat_parameters.data_ptr(),
                        self.scratch.data_ptr(), _native.current_stream(self.device)), "gatres_fused_prepare_backward")
                self.use_graph = False
                self._graphs.clear()
                enqueue()
                return
            rollback()
            if wt_valid:
                # the warm-up step left the transposes of ITS updated weights in scratch; the parameters were rolled
                # back, so make the promise this graph relies on true again
                _native.check(self.lib.gatres_fused_prepare_backward(
                    self.model._cmodel_ref(), C.byref(self._gstruct), self.model.flat_parameters.data_ptr(),
                    self.scratch.data_ptr(), _native.current_stream(self.device)), "gatres_fused_prepare_backward")
            self._graphs[key] = g
            while len(self._graphs) > self._max_graphs:
                torch.cuda.synchronize(self.device)      # (its last replay may still be in flight on the stream)
                self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        if self.fused:
            # a captured split launch cannot order itself behind one on another stream (include/gatres.h)
            _native.check(self.lib.gatres_fused_serialize(_native.current_stream(self.device)), "gatres_fused_serialize")
        g.replay()

    # ---- the multi-rank step: backward pieces | bucketed all-reduce | Adam, as ONE launch sequence / hipGraph ---------
    def _one_piece(self) -> bool:
        """The fused data-parallel step as ONE gradient bucket (the single-GPU step's own launches in front of the all-reduce)."""
        return self.fused and (not self._ranges or self.model.num_blocks < 2 or self.fused_buckets < 2)

    def _backward_pieces(self, device_mask: bool, wt_valid: bool, premasked: bool = False, batch=None, flags: int = 0):
        """Pieces for ``dp.run_data_parallel_step``.  Fused path (gatres_small): one piece -- mask, forward, loss and the
        whole backward are a single launch whose gradient exists only after the slab reduction that ends it.  Per-op
        path (gatres_large, large graphs): forward, then one piece per ``blocks_per_bucket`` blocks in reverse order,
        each ending with the slab reduction of exactly its parameters."""
        if self.fused:
            m = self.model
            head = (0 if premasked else PHASE_MASK) | PHASE_FORWARD | PHASE_BACKWARD
            if self._one_piece():
                # (flags: GATRES_FLAG_MASK_NEXT makes the parameter-gradient launch leave the step-count snapshot that the
                #  Adam launch's sampling tail reads)
                def whole():
                    self._enqueue(head, device_mask, wt_valid, flags=flags, batch=batch)
                    return 0, self.P
                return [whole]
            # Two pieces: the backward chain + the parameter gradients of the UPPER blocks (whose bucket closes the flat
            # vector: it carries lin1), then the lower blocks (+ lin0).  The first bucket's all-reduce runs on the process
            # group's stream while the second launch forms the rest of the gradient ("a single bucket launched
            # after the last K1b and hidden behind ..."): 130 KB on the wire under ~19 us of kernel.
            k = m.num_blocks // 2
            cut = 2 * m.nc + k * (9 * m.nc + 4 * m.nc * m.nc)

            def upper():
                self._enqueue(head, device_mask, wt_valid, flags=FLAG_GRADS_DEFERRED, batch=batch)
                self._enqueue(PHASE_BACKWARD, device_mask, flags=FLAG_GRADS_ONLY, block_lo=k, block_hi=m.num_blocks)
                return cut, self.P

            def lower():
                self._enqueue(PHASE_BACKWARD, device_mask, flags=FLAG_GRADS_ONLY, block_lo=0, block_hi=k)
                return 0, cut
            return [upper, lower]
        m = self.model
        pieces = []
        bx = self.x if batch is None else batch[0]
        for k, (b_hi, b_lo, lo, hi) in enumerate(dp.block_buckets(m.num_blocks, m.nc, self.blocks_per_bucket)):
            def piece(k=k, b_hi=b_hi, b_lo=b_lo, lo=lo, hi=hi):
                if k == 0:
                    self._enqueue((0 if premasked else PHASE_MASK) | PHASE_FORWARD, device_mask, batch=batch)
                flags = PART_REDUCE | (PART_FIRST if b_hi == m.num_blocks else 0) | (PART_LAST if b_lo == 0 else 0)
                _native.check(self.lib.gatres_model_backward_per_op_part(
                    m._cmodel_ref(), C.byref(self._gstruct), m.flat_parameters.data_ptr(), bx.data_ptr(),
                    self.mask.data_ptr(), self.g_out.data_ptr(), self.saved.data_ptr(), self.scratch.data_ptr(),
                    self.grads.data_ptr(), None, b_hi, b_lo, flags, _native.current_stream(self.device)),
                    "gatres_model_backward_per_op_part")
                return lo, hi
            pieces.append(piece)
        return pieces

    def _run_split(self, device_mask: bool, premasked: bool = False, batch=None, slot=None, ahead: bool = True) -> None:
        """The data-parallel step: backward pieces | bucketed all-reduce | Adam.  With one gradient bucket on the fused path
        this is the single-GPU step's launch sequence with the slab sum and the update as two launches around ONE all-reduce:
        the batch is read in place (``batch``), and the Adam launch samples the next step's mask (``ahead``) exactly as the
        single-GPU update launch does.  ahead=False: the caller samples every step's mask itself (staged batches)."""
        # (fused path: the Adam-only phase refreshes scratch's transposed conv weights like the fused Adam pass does)
        wt_valid = self.fused and self._wt_current()
        ahead = bool(ahead and self._mask_next and device_mask and self._one_piece())
        if ahead and not premasked and self._take_mask_ahead():
            premasked = True                       # the previous step's Adam launch has sampled this step's mask already
        elif not ahead:
            self._mask_sig = None
        flags = FLAG_MASK_NEXT if ahead else 0
        try:
            self._replay(self._graph_key(("split", premasked, flags, slot), device_mask, wt_valid),
                         lambda: dp.run_data_parallel_step(
                             self._backward_pieces(device_mask, wt_valid, premasked, batch, flags), self.reducer,
                             lambda: self._enqueue(PHASE_ADAM, device_mask, flags=flags)), wt_valid)
        finally:
            self._count_native_update()
            if self.fused:
                self._wt_sig = self._param_signature()
            self._mask_sig = self._mask_key() if ahead else None

    # ------------------------------------------------------------------------------------------
    def load_batch(self, x: torch.Tensor, y: torch.Tensor, mask: Optional[torch.Tensor] = None) -> None:
        """Stage a batch into the static buffers (async copies on the current stream)."""
        self.x.copy_(x.reshape(-1), non_blocking=True)
        if not self.targets_are_inputs:
            self.y.copy_(y.reshape(-1), non_blocking=True)
        if mask is not None:
            self.mask.copy_(mask.reshape(-1).to(torch.uint8), non_blocking=True)
            self._mask_sig = None                # (the caller's mask replaced whatever the update launch had sampled)

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

    def step_prefetched(self) -> torch.Tensor:
        """One optimisation step (device-sampled mask) on the prefetched batch IN PLACE: the kernels read the staging slot the
        side stream filled -- they never write x -- so there is no device copy and, with the mask sampled ahead by the previous
        update launch, no sampler launch either: the bound-batch step's three launches, the PCIe transfer of the next batch
        under them.  ``commit_batch()`` + ``run_step()`` is the same step through the static buffers (two launches more)."""
        k = self._slot_out
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self._ready[k])
        sx, sy = self._stage[k]
        batch = (sx, sx if sy is None else sy)
        self._step_on(batch, ("at", batch[0].data_ptr(), batch[1].data_ptr()))
        self._free[k].record(cur)                  # (the slot may be refilled once this step has read it)
        self._slot_out = 1 - k
        return self.loss

    def run_step(self, device_mask: bool = True) -> None:
        """One optimisation step on the staged batch.  Nothing is synchronised; read ``self.loss`` afterwards."""
        if device_mask and self.node_ptr is None:
            raise ValueError("device mask sampling needs nodes_per_graph at construction")
        if not self.split:
            self._run(PHASE_MASK | PHASE_FORWARD | PHASE_BACKWARD | PHASE_ADAM, device_mask)
        else:
            self._run_split(device_mask)

    def _stage_with_mask(self, x: torch.Tensor, y: torch.Tensor) -> bool:
        """Device-resident batch + device mask: the copy into the static buffers rides on the mask
        sampler's launch (``gatres_stage_batch_mask``) -- one small kernel and one kernel boundary less per step than
        ``load_batch`` followed by the sampler.  Returns False when the batch has to go through ``load_batch``."""
        if self.node_ptr is None or os.environ.get("GATRES_NO_STAGE_MASK"):
            return False
        n = self.x.numel()
        for t in (x, y):
            if not (t.is_cuda and t.device == self.x.device and t.dtype == torch.float32 and t.numel() == n and t.is_contiguous()):
                return False
        # (a mask the previous update launch may have sampled ahead is NOT used here: the batch needs a staging launch anyway,
        #  and copy + sampling in one launch is cheaper than a copy launch of its own -- measured 0.4204 vs 0.4237 ms/step;
        #  step_bound() / run_step() are the paths without a staging launch)
        self._mask_sig = None
        ys = None if self.targets_are_inputs else y
        _native.check(self.lib.gatres_stage_batch_mask(
            x.data_ptr(), None if ys is None else ys.data_ptr(), self.x.data_ptr(), None if ys is None else self.y.data_ptr(), n,
            self.node_ptr.data_ptr(), self.num_graphs, self.mask_rate, self.seed, self.step_counter.data_ptr(),
            self.mask.data_ptr(), _native.current_stream(self.device)), "gatres_stage_batch_mask")
        return True

    def step_rows(self, data: torch.Tensor, rows: torch.Tensor) -> bool:
        """One step on the snapshots ``rows`` (device int64 ``[num_graphs]``) of the device-resident matrix ``data``
        ``[S, N_g]`` (a ``SnapshotStore``): the batch is collated INSIDE the mask sampler's launch
        (``gatres_stage_rows_mask``) -- no ``index_select``, no copy.  Returns False (nothing done) when this trainer
        cannot take the path (no per-graph node counts, switched off)."""
        if self.node_ptr is None or os.environ.get("GATRES_NO_STAGE_MASK"):
            return False
        npg = self.N // max(self.num_graphs, 1)
        if not (data.is_cuda and data.device == self.x.device and data.dtype == torch.float32 and data.dim() == 2
                and data.shape[1] == npg and data.is_contiguous() and rows.is_cuda and rows.dtype == torch.int64
                and rows.numel() == self.num_graphs and rows.is_contiguous()):
            return False
        _native.check(self.lib.gatres_stage_rows_mask(
            data.data_ptr(), rows.data_ptr(), npg, self.x.data_ptr(), None if self.targets_are_inputs else self.y.data_ptr(),
            self.node_ptr.data_ptr(), self.num_graphs, self.mask_rate, self.seed, self.step_counter.data_ptr(),
            self.mask.data_ptr(), _native.current_stream(self.device)), "gatres_stage_rows_mask")
        if self.split:
            self._run_split(True, premasked=True, ahead=False)
        else:
            self._run_premasked(ahead=False)
        return True

    def _rows_path_ok(self, data: torch.Tensor) -> bool:
        npg = self.N // max(self.num_graphs, 1)
        return (self.node_ptr is not None and not os.environ.get("GATRES_NO_STAGE_MASK") and data.is_cuda
                and data.device == self.x.device and data.dtype == torch.float32 and data.dim() == 2
                and data.shape[1] == npg and data.is_contiguous())

    def steps_rows(self, data: torch.Tensor, rows: torch.Tensor, total: Optional[torch.Tensor] = None) -> bool:
        """``step_rows`` for k consecutive batches -- ``rows`` is a device int64 ``[k * num_graphs]`` slice of an epoch's
        snapshot order -- as ONE captured launch sequence (k x [collation + mask sampler, window kernel, parameter gradients,
        update]): an epoch then pays the host's per-step work and the gap between two graph launches once per k steps.  The
        rows are copied into a window buffer the captured kernels read (5 KB per 20 steps) and step i's update launch writes
        its loss into entry i of a k-vector; every launch, and so every bit of the result, is the single calls'.  ``total``
        (float64 ``[1]``): += sum of the k losses x num_graphs -- in double, where a sum of fp32 losses is exact, so it does not
        matter that the k terms are added at once.  Steady state only (single GPU, fused path, hipGraph replay, transposed
        weights current); returns False (nothing done) otherwise."""
        bs = self.num_graphs
        k = rows.numel() // max(bs, 1)
        if not (k > 1 and rows.numel() == k * bs and self.use_graph and self.fused and not self.split and self._rows_path_ok(data)
                and rows.is_cuda and rows.dtype == torch.int64 and self._wt_current()):
            return False
        win = getattr(self, "_rows_win", None)
        if win is None or win.numel() < k * bs:
            win = self._rows_win = torch.zeros(k * bs, dtype=torch.int64, device=self.device)
            self._loss_seq = torch.zeros(k, dtype=torch.float32, device=self.device)
        win[:k * bs].copy_(rows)
        lseq = self._loss_seq
        npg = self.N // bs
        full = PHASE_FORWARD | PHASE_BACKWARD | PHASE_ADAM

        def enqueue():
            for i in range(k):
                _native.check(self.lib.gatres_stage_rows_mask(
                    data.data_ptr(), win.data_ptr() + 8 * i * bs, npg, self.x.data_ptr(),
                    None if self.targets_are_inputs else self.y.data_ptr(), self.node_ptr.data_ptr(), bs, self.mask_rate, self.seed,
                    self.step_counter.data_ptr(), self.mask.data_ptr(), _native.current_stream(self.device)), "gatres_stage_rows_mask")
                self._enqueue(full, True, True, flags=0, loss=lseq[i:i + 1])

        try:
            self._replay(("rowseq", k, data.data_ptr(), win.data_ptr(), lseq.data_ptr(), self.mask_rate, self.seed, self.world),
                         enqueue, True)
        finally:
            for _ in range(k):
                self._count_native_update()
            self._wt_sig = self._param_signature()
            self._mask_sig = None
        self.loss.copy_(lseq[k - 1:k])                   # (the loss of the step that ran last, as after single steps)
        if total is not None:
            total.add_(lseq[:k].double().sum(), alpha=float(bs))
        return True

    def step(self, x: torch.Tensor, y: torch.Tensor, mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Reference-shaped call: one iteration of train.py:159-190.  Returns the (device) loss tensor."""
        if mask is None and self._stage_with_mask(x, y):                        # (the mask of this step is in place)
            if self.split:
                self._run_split(True, premasked=True, ahead=False)
            else:
                self._run_premasked(ahead=False)
            return self.loss
        self.load_batch(x, y, mask)
        self.run_step(device_mask=mask is None)
        return self.loss

    def _run_premasked(self, batch=None, slot=None, ahead: bool = True) -> None:
        """The single-GPU step on a mask that is already in self.mask (staged with the batch, or sampled by the previous
        step's update launch).  ahead=False: the caller samples every step's mask itself (the store's row collation)."""
        full = PHASE_FORWARD | PHASE_BACKWARD | PHASE_ADAM
        wt_valid = self.fused and self._wt_current()
        ahead = ahead and self._mask_next
        flags = FLAG_MASK_NEXT if ahead else 0
        try:
            self._replay(self._graph_key((full, flags, slot), True, wt_valid),
                         lambda: self._enqueue(full, True, wt_valid, flags=flags, batch=batch), wt_valid)
        finally:
            self._count_native_update()
            if self.fused:
                self._wt_sig = self._param_signature()
            self._mask_sig = self._mask_key() if ahead else None

    # ---- bound batches: the captured step reads the caller's buffers, nothing is copied -------------------------------------
    def bind_batches(self, xs, ys=None, precapture: bool = True, sequences=()) -> int:
        """Register device-resident batches (flat fp32 ``[N]`` tensors; ``ys`` defaults to ``xs``: y == x before masking,
        train.py:162-166) that ``step_bound(i)`` then trains on IN PLACE: the kernels mask x on the fly and never write it,
        so no staging copy is needed -- one captured step per buffer and mask-buffer orientation.  ``precapture``: capture
        every one of those steps NOW (``precapture_bound()``), so that no ``step_bound`` call ever pays for a capture.  The
        trainer keeps the tensors alive; do not resize them.  Returns the number of bound batches."""
        ys = xs if ys is None else ys
        bound = []
        for x, y in zip(xs, ys):
            for t in (x, y):
                _native.require_gpu_tensor(t, "a bound batch")
                if t.numel() != self.N or t.device != self.x.device:
                    raise ValueError(f"a bound batch must hold {self.N} values on {self.x.device}")
            bound.append((x.reshape(-1), y.reshape(-1)))
        if self._graphs:                              # (captured steps of the previous binding read other buffers)
            torch.cuda.synchronize(self.device)
            self._graphs.clear()
        self._bound = bound
        # two captured steps per bound batch (the two mask buffers take turns) + the first-step variants + whatever else this
        # trainer replays: the cache must hold them all, or FIFO eviction makes every step pay for a capture
        self._max_graphs = max(MAX_CACHED_GRAPHS, 2 * len(bound) + 16 + 2 * len(tuple(sequences)))
        if precapture:
            self.precapture_bound(sequences)
        return len(bound)

    def precapture_bound(self, sequences=()) -> int:
        """Capture the steady-state step of every (bound batch, mask-buffer orientation) pair by RUNNING steps, then put the
        training state back exactly where it was (parameters, Adam moments, step count) -- with the transposed weights current
        and the first step's mask sampled ahead, so that the first real ``step_bound`` call already replays a steady-state
        graph.  Nothing is captured inside a caller's timed region afterwards (bench.py's first timed block held
        four captures).  ``sequences``: tuples of bound-batch indices that the caller will run through ``steps_bound`` -- each is
        captured too, under both orientations of the mask buffers.  Under data parallelism every rank runs the same number of
        steps (the collectives match).  Returns the number of captured graphs."""
        if not (self.use_graph and self._bound and self.node_ptr is not None):
            return len(self._graphs)
        dev = self.device
        torch.cuda.synchronize(dev)
        keep = [t.clone() for t in (self.step_counter, self.model.flat_parameters, self.exp_avg, self.exp_avg_sq, self.loss)]
        try:
            n = len(self._bound)
            # (mask-ahead: the two mask buffers take turns, so a batch has two steady-state graphs; without it, one)
            ahead_ok = self._mask_next and (not self.split or self._one_piece())
            ptrs = (self.mask.data_ptr(), self._mask_spare.data_ptr()) if ahead_ok else (None,)
            need = {(i, p) for i in range(n) for p in ptrs}
            steps = 0
            while need and steps < 4 * n + 8:
                # the step about to run is a steady-state one iff the weights' transposes are current and (mask-ahead) a mask was
                # sampled ahead; it then trains with the (current) spare buffer as its mask
                steady = (not self.fused) or self._wt_current()
                ptr = None
                if ahead_ok:
                    steady = steady and self._mask_sig is not None and self._mask_sig == self._mask_key()
                    ptr = self._mask_spare.data_ptr() if steady else None
                i = next((j for j in range(n) if (j, ptr) in need), 0)
                self.step_bound(i)
                if steady:
                    need.discard((i, ptr))
                steps += 1
            # multi-step sequences (steps_bound): each one under both start orientations of the mask buffers
            for seq in sequences:
                seq = tuple(int(i) for i in seq)
                if len(seq) < 2 or not ahead_ok or self.split:
                    continue
                seen = set()
                for _ in range(4):
                    if len(seen) == 2:
                        break
                    start = self._mask_spare.data_ptr()
                    if start in seen:
                        self.step_bound(seq[0])            # (flip the orientation)
                        continue
                    self.steps_bound(seq)
                    seen.add(start)
            self._max_graphs = max(self._max_graphs, len(self._graphs) + 16)
        finally:
            # whatever happened above (a capture or a launch may raise midway): the caller's model, moments and step count
            # come back, and neither the sampled-ahead mask nor scratch's transposed weights is taken for current
            torch.cuda.synchronize(dev)
            for dst, src in zip((self.step_counter, self.model.flat_parameters, self.exp_avg, self.exp_avg_sq, self.loss), keep):
                dst.copy_(src)
            self._mask_sig = None
            self._wt_sig = None
        if self.fused:
            # scratch holds the transposes of the LAST pre-capture step's weights: re-derive them for the restored parameters
            _native.check(self.lib.gatres_fused_prepare_backward(
                self.model._cmodel_ref(), C.byref(self._gstruct), self.model.flat_parameters.data_ptr(),
                self.scratch.data_ptr(), _native.current_stream(dev)), "gatres_fused_prepare_backward")
            self._count_native_update()                # (siblings / other trainers of this model: their copies are stale too)
            self._wt_sig = self._param_signature()
        if ahead_ok:
            # the first step's mask, sampled where the previous step's update launch would have left it (same key: same bits)
            _native.check(self.lib.gatres_mask_generate(
                self.node_ptr.data_ptr(), self.num_graphs, self.mask_rate, self.seed, self.step_counter.data_ptr(),
                self._mask_spare.data_ptr(), _native.current_stream(dev)), "gatres_mask_generate")
            self._mask_sig = self._mask_key()
        torch.cuda.synchronize(dev)
        self._faults_seen = self.fault_count + self.dropped_steps
        return len(self._graphs)

    def step_bound(self, i: int) -> torch.Tensor:
        """One optimisation step (train.py:159-190) on bound batch ``i`` with a device-sampled mask.  On the single-GPU fused
        path this is three launches -- the per-snapshot kernel, the parameter gradients, the update (which also samples the
        next step's mask) -- replayed from one hipGraph; the first step (and any step after the optimizer state moved under
        the trainer) runs the sampler's launch first."""
        x, y = self._bound[i]
        if self.node_ptr is None:
            raise ValueError("device mask sampling needs nodes_per_graph at construction")
        if self.split:                     # (data-parallel step: the same launches around the all-reduce, batch read in place)
            self._run_split(True, batch=(x, y), slot=i)
            return self.loss
        self._run(PHASE_MASK | PHASE_FORWARD | PHASE_BACKWARD | PHASE_ADAM, True, batch=(x, y), slot=i)
        return self.loss

    def steps_bound(self, indices, bound=None, losses: Optional[torch.Tensor] = None) -> torch.Tensor:
        """``step_bound(i)`` for every i of ``indices``, in order -- as ONE captured launch sequence when the trainer is in its
        steady state (single GPU, fused path, transposed weights current, the first mask sampled ahead): 3 k kernels in one
        hipGraph, so the ~8 us between two graph launches are paid once per k steps instead of once per step.  The two mask
        buffers take turns inside the sequence exactly as they do between ``step_bound`` calls (step j trains on the mask the
        update launch of step j - 1 sampled), so the results are those of the single calls bit for bit.  Anything else (first
        step, data-parallel trainer, eager launches) falls back to the single calls.  ``bound``: another list of resident
        ``(x, y)`` batches than the one ``bind_batches`` registered (``fit_epoch``'s epoch buffer); ``losses``: a float32 vector
        whose entry j receives step j's loss.  Returns the last step's loss tensor."""
        idx = tuple(int(i) for i in indices)
        bnd = self._bound if bound is None else bound
        if losses is not None and losses.numel() < len(idx):
            raise ValueError(f"`losses` holds {losses.numel()} entries for {len(idx)} steps")
        steady = (len(idx) > 1 and self.use_graph and self.fused and not self.split and self._mask_next and self._wt_current()
                  and self._mask_sig is not None and self._mask_sig == self._mask_key() and self.node_ptr is not None)
        if not steady:
            for j, i in enumerate(idx):
                if bound is None:
                    self.step_bound(i)
                else:
                    self._step_on(bnd[i], ("at", bnd[i][0].data_ptr(), bnd[i][1].data_ptr()))
                if losses is not None:
                    losses[j:j + 1].copy_(self.loss)
            return self.loss
        full = PHASE_FORWARD | PHASE_BACKWARD | PHASE_ADAM
        m0, m1 = self._mask_spare, self.mask          # step 0 trains on the mask sampled ahead and samples step 1's into the other

        def enqueue():
            a, b = m0, m1
            for j, i in enumerate(idx):
                self._enqueue(full, True, True, flags=FLAG_MASK_NEXT, batch=bnd[i], masks=(a, b),
                              loss=None if losses is None else losses[j:j + 1])
                a, b = b, a

        try:
            self._replay(("seq", idx, self.mask_rate, self.seed, self.world, m0.data_ptr(), bnd[idx[0]][0].data_ptr(),
                          None if losses is None else losses.data_ptr()), enqueue, True)
        except BaseException:
            # an unknown number of the sequence's steps ran: nothing about the mask buffers or the transposed weights can be
            # relied on -- the next step samples its own mask and re-derives the transposes
            self._count_native_update()
            self._mask_sig = None
            self._wt_sig = None
            raise
        for _ in idx:
            self._count_native_update()
        self._wt_sig = self._param_signature()
        if len(idx) % 2:                               # an odd number of steps leaves the buffers swapped
            self.mask, self._mask_spare = self._mask_spare, self.mask
        self._mask_sig = self._mask_key()              # (the last update launch sampled the next step's mask into the spare buffer)
        if losses is not None:
            self.loss.copy_(losses[len(idx) - 1:len(idx)])
        return self.loss

    def _step_on(self, batch, slot) -> torch.Tensor:
        """``step_bound`` on a resident ``(x, y)`` batch that is not in the bound list (``slot`` names it in the graph cache: use
        the buffers' addresses)."""
        if self.node_ptr is None:
            raise ValueError("device mask sampling needs nodes_per_graph at construction")
        if self.split:
            self._run_split(True, batch=batch, slot=slot)
        else:
            self._run(PHASE_MASK | PHASE_FORWARD | PHASE_BACKWARD | PHASE_ADAM, True, batch=batch, slot=slot)
        return self.loss

    def forward_backward(self, x: torch.Tensor, y: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """Forward + loss + gradients only (no optimiser update); gradients land in ``self.grads``."""
        self.load_batch(x, y, mask)
        self._enqueue(PHASE_FORWARD | PHASE_BACKWARD, False)
        return self.loss

    @property
    def optimizer_step(self) -> int:
        return int(self.step_counter[0].item())

    # ---- epoch-level loop (train.py:112-202) ---------------------------------------------------------------------------
    def _epoch_in_place(self, store, order: torch.Tensor, nfull: int, kseq: int, total: torch.Tensor) -> None:
        bs, npg = self.num_graphs, store.nodes_per_graph
        buf = getattr(self, "_epoch_buf", None)
        if buf is None or tuple(buf.shape) != (nfull * bs, npg):
            if buf is not None:
                self.release_epoch_buffer()
            buf = self._epoch_buf = torch.empty(nfull * bs, npg, dtype=torch.float32, device=self.device)
            views = [buf[j * bs:(j + 1) * bs].reshape(-1) for j in range(nfull)]
            self._epoch_bound = [(v, v) for v in views]            # (x is also y: train.py:162-166)
            self._epoch_losses = None
        if getattr(self, "_epoch_losses", None) is None or self._epoch_losses.numel() < max(kseq, 2):
            # (also when epoch_graph_steps was raised between two epochs on the same store: a step's loss slot must exist --
            #  ADVICE r5; the losses pointer is part of the sequences' graph key, so no stale graph meets the new vector)
            self._epoch_losses = torch.zeros(max(kseq, 2), dtype=torch.float32, device=self.device)
            # sequences of k (one graph per run of the epoch and mask orientation), single steps where a run starts or ends
            self._max_graphs = max(self._max_graphs, MAX_CACHED_GRAPHS + 2 * (nfull // kseq + 1) + 2 * kseq + 8)
        torch.index_select(store.data, 0, order[:nfull * bs], out=buf)
        eb, lseq = self._epoch_bound, self._epoch_losses
        j = 0
        while j < nfull:
            steady = (self._wt_current() and self._mask_sig is not None and self._mask_sig == self._mask_key())
            k = min(kseq, nfull - j) if steady else 1
            if k > 1:
                self.steps_bound(range(j, j + k), bound=eb, losses=lseq)
                total.add_(lseq[:k].double().sum(), alpha=float(bs))
            else:
                self._step_on(eb[j], ("at", eb[j][0].data_ptr(), eb[j][1].data_ptr()))   # (the slot names the buffers: a re-allocated epoch buffer never meets a stale graph)
                total.add_(self.loss.double(), alpha=float(bs))
            j += k

    def _epoch_copy_fits(self, rows: int, npg: int) -> bool:
        """May ``fit_epoch`` keep a device copy of the epoch's shuffled batches (``rows`` x ``npg`` floats: as large as the store
        itself, i.e. the store's footprint doubles)?  Yes if it already exists, or if it is within ``epoch_copy_limit_bytes``
        AND leaves a quarter of the device's free memory untouched; otherwise the epoch goes through the row-gather steps
        (``steps_rows``), which need no copy."""
        buf = getattr(self, "_epoch_buf", None)
        if buf is not None and tuple(buf.shape) == (rows, npg):
            return True
        need = 4 * rows * npg
        if need > int(self.epoch_copy_limit_bytes):
            return False
        held = 0 if buf is None else buf.numel() * 4          # (a buffer of another shape is released first)
        free, _ = torch.cuda.mem_get_info(self.device)
        return need <= 0.75 * (free + held)

    def release_epoch_buffer(self) -> None:
        """Free ``fit_epoch``'s device copy of the epoch (and the graphs captured on it)."""
        if getattr(self, "_epoch_buf", None) is not None:
            torch.cuda.synchronize(self.device)
            self._graphs.clear()
            self._epoch_buf = self._epoch_bound = self._epoch_losses = None

    def _sibling(self, num_graphs: int, nodes_per_graph: int, edge_index: torch.Tensor) -> "GATResTrainer":
        """Trainer for another batch size of the same dataset (the ragged last batch of an epoch): its own plan and
        buffers, the SAME model, Adam moments and step counter."""
        t = self._siblings.get(num_graphs)
        if t is None:
            t = GATResTrainer(self.model, edge_index, num_graphs * nodes_per_graph,
                              nodes_per_graph=[nodes_per_graph] * num_graphs, lr=self.hparams["lr"],
                              weight_decay=self.hparams["weight_decay"],
                              betas=(self.hparams["beta1"], self.hparams["beta2"]), eps=self.hparams["eps"],
                              mask_rate=self.mask_rate, seed=0, process_group=self.pg, use_graph=self.use_graph,
                              fused=self.fused, force_collective_path=self.split and self.world == 1,
                              targets_are_inputs=True, blocks_per_bucket=self.blocks_per_bucket,
                              _share_state_with=self)
            t.seed = self.seed
            self._siblings[num_graphs] = t
        t.mask_rate = self.mask_rate
        return t

    def fit_epoch(self, store, batch_size: Optional[int] = None, shuffle: bool = True, drop_last: bool = False,
                  generator: Optional[torch.Generator] = None, metric_fn_dict: Optional[Dict[str, Callable]] = None,
                  ) -> Tuple[float, Dict[str, float]]:
        """One epoch over a ``SnapshotStore`` -- the loop of ``train_one_epoch`` (train.py:159-198): shuffled batches, a
        fresh per-graph mask every batch, one optimisation step each, loss weighted by the batch's graph count.
        Batches are row gathers of the device-resident store (no host collation, no H2D); a ragged last batch runs on
        a sibling trainer that shares this one's optimizer state.  Returns ``(mean loss, metrics)``; metrics (the
        reference's de-normalised per-batch metrics, train.py:179-198) are only computed when ``metric_fn_dict`` is
        given, since they need a boolean gather per batch.  Under data parallelism every rank passes its own shard of
        the snapshots (``dp.shard_graphs``)."""
        npg = store.nodes_per_graph
        bs = int(batch_size) if batch_size is not None else self.num_graphs
        if bs != self.num_graphs or npg * bs != self.N:
            raise ValueError(f"this trainer was built for batches of {self.num_graphs} graphs x {self.N // max(self.num_graphs, 1)} nodes")
        if self.world > 1:
            drop_last = True                 # equal graph counts on every rank keep plain gradient averaging exact
        # the epoch's loss sum in DOUBLE (train.py:196 accumulates Python floats): a sum of fp32 losses is exact there, so a
        # sequence's k losses added at once and the same losses added one by one give the same bits
        total = torch.zeros(1, dtype=torch.float64, device=self.device)
        sums = {k: 0.0 for k in (metric_fn_dict or {})}
        seen = 0
        # the epoch's snapshot order goes to the device ONCE; a batch is a slice of it, collated inside the mask sampler's
        # launch (step_rows) -- the reference collates on the host and copies every iteration (train.py:302-303).  Runs of
        # `epoch_graph_steps` full batches go through ONE captured sequence each (steps_rows); what is left -- the first step
        # of a run whose transposed weights are not current, the epoch's last few batches, the ragged one -- goes step by step.
        kseq = int(self.epoch_graph_steps)
        if not (self.use_graph and self.fused and not self.split and self._rows_path_ok(store.data)):
            kseq = 1                                     # (eager / data-parallel / per-op trainers: step by step)
        order = store.epoch_order(shuffle=shuffle, generator=generator)
        first = 0                                        # batches [0, first) are done by the in-place path below
        nfull = store.num_snapshots // bs
        if (kseq > 1 and not metric_fn_dict and nfull >= 2 and self._mask_next and self.node_ptr is not None
                and self._epoch_copy_fits(nfull * bs, npg)):
            # The epoch's full batches as ONE device gather into a persistent buffer (the shuffled store: S x N_g floats), then
            # trained IN PLACE, k per captured launch sequence, with the mask sampled ahead by the update launches -- the
            # bound-batch step of bench.py, three launches and nothing else: no collation launch, no per-step host work.
            self._epoch_in_place(store, order, nfull, kseq, total)
            first, seen = nfull, nfull * bs
        pending = []                                     # full batches waiting for a sequence

        def flush(n_keep: int = 0):
            nonlocal seen
            while len(pending) > n_keep:
                rows_one = pending.pop(0)
                if not self.step_rows(store.data, rows_one):
                    x = store.batch(rows_one)
                    self.load_batch(x, x)
                    self.run_step(device_mask=True)
                total.add_(self.loss.double(), alpha=float(bs))
                seen += bs

        for rows, edge_index, ng in store.row_batches(bs, drop_last=drop_last, order=order, first=first):
            if ng == bs and not metric_fn_dict and kseq > 1:
                pending.append(rows)
                if len(pending) == kseq:
                    # (row_batches yields consecutive slices of ONE order tensor: the run is its slice too)
                    run = torch.cat(pending) if pending[0].data_ptr() + 8 * bs * (kseq - 1) != pending[-1].data_ptr() else \
                        torch.as_strided(pending[0], (kseq * bs,), (1,))
                    if self.steps_rows(store.data, run, total):
                        seen += kseq * bs
                        pending.clear()
                    else:
                        flush(n_keep=kseq - 1)           # one step by itself (it makes the weights current), then try again
                continue
            flush()
            tr = self if ng == bs else self._sibling(ng, npg, edge_index)
            if not tr.step_rows(store.data, rows):
                x = store.batch(rows)
                tr.load_batch(x, x)
                tr.run_step(device_mask=True)
            total.add_(tr.loss.double(), alpha=float(ng))
            if metric_fn_dict:
                m = tr.mask.bool()
                p, t = store.descale(tr.out[m]), store.descale(tr.x[m])
                for k, fn in metric_fn_dict.items():
                    sums[k] += float(fn(p, t)) * ng
            seen += ng
        flush()
        if seen == 0:
            raise ValueError("the store yielded no batch")
        mean_loss = float(total.item()) / seen
        # a dropped step (a split launch whose workgroups were not co-resident) trained on nothing: an epoch that contains one
        # is an error, not a slightly smaller epoch
        self.check_no_dropped_steps()
        for t in self._siblings.values():
            t.check_no_dropped_steps()
        return mean_loss, {k: v / seen for k, v in sums.items()}

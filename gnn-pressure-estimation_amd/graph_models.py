"""GATRes on MI355X: the reference's ``nn.Module`` surface over the gfx950 HIP engine.

Mirrors gnn_pressure_estimation/GraphModels.py:454-494 (``GResBlockMeanConv``, ``GATResMeanConv``): same class
names, constructor arguments, ``forward(x, edge_index, batch=None, edge_attr=None)`` signature, ``.name`` attribute
and the ``state_dict`` keys torch_geometric's ``GATConv`` / ``Linear`` produce, so ``train.py`` / ``evaluation.py``
call it unchanged and reference checkpoints load.  Underneath there is no torch_geometric, torch_scatter or
Triton: one autograd node runs the whole network through ``gatres_model_forward`` / ``gatres_model_backward``
(include/gatres.h).  There is no CPU path; tensors must live on the ROCm device.
"""
from __future__ import annotations

import math
import os
from typing import List, Optional

import torch
from torch import Tensor, nn

from . import _native
from .graph_plan import GraphPlan, PlanCache


# ----------------------------------------------------------------------------------------------
# parameter containers with PyG's names / shapes / initialisers
# ----------------------------------------------------------------------------------------------
# Every (re-)registration of a parameter on one of this file's modules bumps this counter (``module.weight = nn.Parameter(..)``,
# ``load_state_dict(assign=True)``): ``GATResMeanConv._flat_is_current`` then compares one integer instead of walking 124
# ``_parameters`` dicts per forward call (the reference's evaluation times every call: utils/timer.py:22-41).
_PARAM_EPOCH = [0]


class _TrackedModule(nn.Module):
    def register_parameter(self, name, param) -> None:
        _PARAM_EPOCH[0] += 1
        super().register_parameter(name, param)

    def __delattr__(self, name) -> None:
        if name in self.__dict__.get("_parameters", ()):
            _PARAM_EPOCH[0] += 1
        super().__delattr__(name)


def _glorot_(t: Tensor) -> None:
    stdv = math.sqrt(6.0 / (t.size(-2) + t.size(-1)))
    with torch.no_grad():
        t.uniform_(-stdv, stdv)


class Linear(_TrackedModule):
    """Parameter holder for ``torch_geometric.nn.dense.linear.Linear`` (GraphModels.py:11,477,484)."""

    def __init__(self, in_channels: int, out_channels: int, bias: bool = True, weight_initializer: Optional[str] = None):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.weight_initializer = weight_initializer
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels))
        self.bias = nn.Parameter(torch.empty(out_channels)) if bias else None
        self.reset_parameters()

    def reset_parameters(self) -> None:
        if self.weight_initializer == "glorot":
            _glorot_(self.weight)
        else:  # PyG default: kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(fan_in))
            bound = 1.0 / math.sqrt(self.in_channels)
            with torch.no_grad():
                self.weight.uniform_(-bound, bound)
        if self.bias is not None:
            bound = 1.0 / math.sqrt(self.in_channels)
            with torch.no_grad():
                self.bias.uniform_(-bound, bound)

    def extra_repr(self) -> str:
        return f"{self.in_channels}, {self.out_channels}, bias={self.bias is not None}"


class GATConv(_TrackedModule):
    """Parameter holder for ``torch_geometric.nn.GATConv(in, out, heads, concat)`` as the reference builds it
    (GraphModels.py:458-459: edge_dim=None, add_self_loops=True, negative_slope=0.2, dropout=0, bias=True).
    ``lin_dst`` aliases ``lin_src`` exactly as PyG 2.3 does, so both keys appear in the state_dict."""

    def __init__(self, in_channels: int, out_channels: int, heads: int = 1, concat: bool = True):
        super().__init__()
        if not concat and heads != 1:
            raise ValueError("the gfx950 engine implements concat=False only for heads=1 (GATRes' conv2)")
        self.in_channels, self.out_channels, self.heads, self.concat = in_channels, out_channels, heads, concat
        self.lin_src = Linear(in_channels, heads * out_channels, bias=False, weight_initializer="glorot")
        self.lin_dst = self.lin_src
        self.att_src = nn.Parameter(torch.empty(1, heads, out_channels))
        self.att_dst = nn.Parameter(torch.empty(1, heads, out_channels))
        self.bias = nn.Parameter(torch.empty(heads * out_channels if concat else out_channels))
        self.reset_parameters()

    def reset_parameters(self) -> None:
        self.lin_src.reset_parameters()
        _glorot_(self.att_src)
        _glorot_(self.att_dst)
        with torch.no_grad():
            self.bias.zero_()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # PyG >= 2.5 stores the shared projection once as `lin.weight`
        k = prefix + "lin.weight"
        if k in state_dict:
            w = state_dict.pop(k)
            state_dict.setdefault(prefix + "lin_src.weight", w)
            state_dict.setdefault(prefix + "lin_dst.weight", w)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def extra_repr(self) -> str:
        return f"{self.in_channels}, {self.out_channels}, heads={self.heads}"


class SimpleConv(nn.Module):
    """Marker for ``torch_geometric.nn.conv.SimpleConv(aggr='mean')`` (no parameters; GraphModels.py:460)."""

    def __init__(self, aggr: str = "mean"):
        super().__init__()
        if aggr != "mean":
            raise ValueError("only aggr='mean' is used by GATRes")
        self.aggr = aggr


class GResBlockMeanConv(nn.Module):
    """GraphModels.py:454-468.  The block's arithmetic runs inside ``GATResMeanConv``'s fused launch sequence."""

    def __init__(self, in_dim: int, out_dim: int, hc: int):
        super().__init__()
        self.conv1 = GATConv(in_dim, hc, 2, concat=True)
        self.conv2 = GATConv(hc * 2, out_dim, 1, concat=False)
        self.mean_conv = SimpleConv(aggr="mean")


# ----------------------------------------------------------------------------------------------
# the autograd node
# ----------------------------------------------------------------------------------------------
class _GATResFunction(torch.autograd.Function):
    """One autograd node for the whole network.

    ``direct=True`` (the default way in, ``GATResMeanConv.direct_param_grads``): the parameters are NOT inputs of the
    node -- one of them rides along as an anchor so that the output carries a ``grad_fn`` -- and ``backward`` writes
    the flat gradient straight into a persistent buffer the parameters' ``.grad`` tensors are views of: no
    ``AccumulateGrad`` node per parameter (124 of them for gatres_small: ~0.5 ms of host time per step), no allocation.
    ``direct=False``: every parameter is an input and receives its gradient through autograd (hooks, ``autograd.grad``,
    ``DistributedDataParallel``)."""

    @staticmethod
    def forward(ctx, module: "GATResMeanConv", plan: GraphPlan, needs_grad: bool, direct: bool, x: Tensor,
                *params: Tensor) -> Tensor:
        out, saved = module._run_forward(plan, x, needs_grad)
        ctx.module, ctx.plan, ctx.saved_acts, ctx.direct = module, plan, saved, direct
        ctx.save_for_backward(x)
        ctx.params_like = params
        return out

    @staticmethod
    def backward(ctx, g_out: Tensor):
        module, plan, saved = ctx.module, ctx.plan, ctx.saved_acts
        (x,) = ctx.saved_tensors
        if saved is None:
            raise RuntimeError("backward through a forward that ran without grad")
        lib = _native.load()
        g_out = g_out.contiguous()
        if g_out.dtype != torch.float32:
            raise ValueError("grad_output must be float32")
        if ctx.direct:
            grads, slot = module._grad_target()
        else:
            grads = torch.empty(module._flat.numel(), dtype=torch.float32, device=x.device)
        g_x = torch.empty_like(x) if ctx.needs_input_grad[4] else None
        st = module._call_state(plan)
        stream = _native.current_stream(x.device)
        _native.check(lib.gatres_model_backward(st["mref"], st["gref"], module._flat.data_ptr(), x.data_ptr(),
                                                None, g_out.data_ptr(), saved.data_ptr(), st["scratch"].data_ptr(),
                                                grads.data_ptr(), _native.ptr(g_x), stream), "gatres_model_backward")
        if ctx.direct:
            module._grad_deliver(slot)
            return (None, None, None, None, g_x, None)
        # one C++ call instead of 182 Python slice+view pairs: views of `grads` shaped like the parameters
        return (None, None, None, None, g_x) + tuple(torch._utils._unflatten_dense_tensors(grads, ctx.params_like))


# ----------------------------------------------------------------------------------------------
# the model
# ----------------------------------------------------------------------------------------------
class GATResMeanConv(_TrackedModule):
    """GraphModels.py:471-494.  ``gatres_small`` = (num_blocks=15, nc=32), ``gatres_large`` = (25, 128)
    (ConfigModels.py:22-42)."""

    def __init__(self, name: str = "GATResMeanConv", num_blocks: int = 5, nc: int = 32, fused: bool = True):
        super().__init__()
        self.fused = fused          # False: always run the per-op kernels (one launch per stage)
        if nc < 4 or nc > 128 or (nc & (nc - 1)):
            raise ValueError(f"nc={nc}: the gfx950 kernels support powers of two in [4, 128]")
        self.num_blocks = num_blocks
        self.nc = nc
        self.lin0 = Linear(1, nc)
        self.blocks = nn.ModuleList()
        self.name = name
        for _ in range(self.num_blocks):
            self.blocks.append(GResBlockMeanConv(nc, nc, nc))
        self.lin1 = Linear(nc, 1)
        self._flat: Optional[Tensor] = None
        # True: loss.backward() writes the parameter gradients in place into persistent ``.grad`` views of one flat buffer
        # (_GATResFunction); set False where gradients must travel through autograd (tensor hooks are detected and
        # switch it off by themselves; torch DistributedDataParallel and torch.autograd.grad(out, parameters) are not)
        self.direct_param_grads = True
        self._plans = PlanCache(segments=fused)
        self._scratch = {}
        self._call_states = {}
        self._cmodel = _native.GatresModel(num_blocks, nc)
        self._flatten_parameters()

    # ---- copy / pickle: engine handles (ctypes structs, device plans, scratch) are rebuilt, not copied ---------
    def __getstate__(self):
        state = self.__dict__.copy()
        for k in ("_plans", "_scratch", "_cmodel", "_flat", "_param_table", "_param_list", "_grad_bufs", "_grad_views",
                  "_grad_cur", "_call_states", "_flat_param"):
            state.pop(k, None)
        state["_act_dtype"] = int(self._cmodel.act_dtype)
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        self._flat = None
        self._plans = PlanCache(segments=getattr(self, 'fused', True))
        self._scratch = {}
        self._call_states = {}
        self._cmodel = _native.GatresModel(self.num_blocks, self.nc, int(getattr(self, "_act_dtype", 0)), 0)
        self._flatten_parameters()

    # ---- flat parameter storage (state_dict order == include/gatres.h layout) ------------------
    def _flatten_parameters(self) -> None:
        params = list(self.parameters())
        total = sum(p.numel() for p in params)
        expect = 2 * self.nc + self.num_blocks * (9 * self.nc + 4 * self.nc * self.nc) + self.nc + 1
        if total != expect:
            raise RuntimeError(f"parameter count {total} != layout {expect}")
        for p in params:
            if p.dtype != torch.float32:
                raise ValueError("the gfx950 engine computes in float32; do not cast the module")
        flat = torch.empty(total, dtype=torch.float32, device=params[0].device)
        off = 0
        with torch.no_grad():
            for p in params:
                n = p.numel()
                flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = flat[off:off + n].view(p.shape)
                off += n
        self._flat = flat
        self._scratch = {}
        self._call_states = {}
        # (owner dict, name, parameter, byte offset) of every parameter in state_dict order: Module.parameters() walks
        # the whole module tree (0.4 ms for 182 parameters), too slow to repeat in every forward
        table = []
        off = 0
        seen = set()
        for mod in self.modules():
            for name, q in mod._parameters.items():
                if q is not None and id(q) not in seen:
                    seen.add(id(q))
                    table.append((mod._parameters, name, q, 4 * off))
                    off += q.numel()
        if len(table) != len(params) or any(t[2] is not q for t, q in zip(table, params)):
            raise RuntimeError("parameter traversal order changed")
        self._param_table = table
        self._param_list = params
        self._param_ptrs = [4 * 0 + flat.data_ptr() + t[3] for t in table]       # (where every parameter must point)
        self._param_epoch = _PARAM_EPOCH[0]
        self._grad_bufs = None          # (two flat gradient buffers and the parameters' views of them: _grad_target)
        self._grad_views = None
        self._grad_cur = None
        object.__setattr__(self, "_flat_param", None)      # (optimizer_parameters(): rebuilt over the new flat vector on demand)

    # ---- gradients delivered in place (direct_param_grads) ------------------------------------------------------
    def _grad_target(self):
        """(flat buffer the backward launch writes, its slot).  Two persistent buffers take turns, so the gradients of the
        previous backward stay intact for whoever still holds them while this one is formed."""
        if self._grad_bufs is None:
            flat = self._flat
            self._grad_bufs = [torch.zeros_like(flat), torch.zeros_like(flat)]
            self._grad_views = []
            for buf in self._grad_bufs:
                views, off = [], 0
                for p in self._param_list:
                    n = p.numel()
                    views.append(buf[off:off + n].view(p.shape))
                    off += n
                self._grad_views.append(views)
            self._grad_cur = None
        fp = self._flat_param
        if fp is not None:
            # flat mode (optimizer_parameters()): ONE gradient buffer, slot 0, which ``flat_parameter.grad`` and the
            # per-parameter ``.grad`` views stay attached to; slot 1 is the temporary of an accumulating backward
            slot = 0 if fp.grad is None else 1
            return self._grad_bufs[slot], slot
        slot = 0 if self._grad_cur is None else 1 - self._grad_cur
        return self._grad_bufs[slot], slot

    def _grad_deliver(self, slot: int) -> None:
        """What AccumulateGrad does, for all parameters at once: every ``.grad`` is None (``zero_grad()``) -> the views of
        the buffer just written become the gradients; every ``.grad`` still is the view this module attached last time
        (``zero_grad(set_to_none=False)``, gradient accumulation) -> one flat ``add_``; anything else -> per parameter."""
        fp = self._flat_param
        if fp is not None:
            if slot == 0:                           # flat_parameter.grad was None: the buffer just written IS the gradient
                fp.grad = self._grad_bufs[0]
                if self._grad_cur != 0:             # (the per-parameter views: attached once, they stay)
                    for p, v in zip(self._param_list, self._grad_views[0]):
                        p.grad = v
                    self._grad_cur = 0
            else:
                g = fp.grad
                if g is not self._grad_bufs[0]:     # (a foreign gradient tensor on the flat parameter)
                    g.add_(self._grad_bufs[1])
                else:
                    self._grad_bufs[0].add_(self._grad_bufs[1])
            return
        params, new = self._param_list, self._grad_views[slot]
        cur = self._grad_views[self._grad_cur] if self._grad_cur is not None else None
        n_none = n_ours = 0
        for i, p in enumerate(params):
            g = p.grad
            if g is None:
                n_none += 1
            elif cur is not None and g is cur[i]:
                n_ours += 1
        if n_none == len(params):
            for p, v in zip(params, new):
                p.grad = v
            self._grad_cur = slot
        elif n_ours == len(params):
            self._grad_bufs[self._grad_cur].add_(self._grad_bufs[slot])
        else:
            with torch.no_grad():
                for p, v in zip(params, new):
                    if p.grad is None:
                        p.grad = v.clone()
                    else:
                        p.grad.add_(v)

    def optimizer_parameters(self) -> List[nn.Parameter]:
        """``[flat_parameter]``: ONE leaf ``nn.Parameter`` over the flat vector every named parameter is a view of -- for
        ``torch.optim.Adam(model.optimizer_parameters(), ...)`` in place of ``model.parameters()`` (train.py:348).  Adam is
        element-wise, so the update is the 124-tensor one bit for bit, while the optimizer's per-tensor host work (0.9 ms per
        step for gatres_small) is paid once.  From this call on the module is in FLAT MODE: ``loss.backward()`` delivers the
        gradient as ``flat_parameter.grad`` (the named parameters' ``.grad`` are views of the same buffer, for inspection);
        what ``zero_grad()`` / accumulation see is the flat parameter's ``.grad`` alone.  ``state_dict()`` is unchanged (the
        flat parameter is not registered); an optimizer ``state_dict`` then holds one tensor, not the reference's 124."""
        if self._flat_param is None:
            if not self._flat_is_current():
                self._flatten_parameters()
            fp = nn.Parameter(self._flat, requires_grad=True)
            if fp.data_ptr() != self._flat.data_ptr():
                raise RuntimeError("flat parameter does not alias the flat vector")
            object.__setattr__(self, "_flat_param", fp)          # (not registered: state_dict / parameters() stay the reference's)
            self._grad_cur = None
        return [self._flat_param]

    @property
    def flat_parameter(self) -> Optional[nn.Parameter]:
        return self._flat_param

    def zero_grad(self, set_to_none: bool = True) -> None:
        super().zero_grad(set_to_none)
        fp = self._flat_param
        if fp is not None:                                   # (flat mode: the flat parameter's gradient is THE gradient)
            if set_to_none:
                fp.grad = None
                self._grad_cur = None                        # (the named parameters' views were just detached: re-attach)
            elif fp.grad is not None:
                fp.grad.zero_()

    def flat_grad(self) -> Optional[Tensor]:
        """The flat fp32 buffer all parameter gradients currently are views of (state_dict order), or None when they are
        not (no backward yet, ``zero_grad()`` since, or gradients that came another way)."""
        fp = self._flat_param
        if fp is not None:
            return fp.grad if fp.grad is self._grad_bufs[0] else None
        if self._grad_cur is None:
            return None
        cur = self._grad_views[self._grad_cur]
        for p, v in zip(self._param_list, cur):
            if p.grad is not v:
                return None
        return self._grad_bufs[self._grad_cur]

    def _direct_ok(self, params) -> bool:
        if not self.direct_param_grads:
            if self._flat_param is not None:
                raise RuntimeError("flat mode (optimizer_parameters()) needs direct_param_grads")
            return False
        for p in params:
            if not p.requires_grad or p._backward_hooks or getattr(p, "_post_accumulate_grad_hooks", None):
                return False
        return True

    def _flat_is_current(self) -> bool:
        flat = self._flat
        table = getattr(self, "_param_table", None)
        if flat is None or table is None:
            return False
        if self._param_epoch != _PARAM_EPOCH[0]:
            # some parameter of some module of this file was (re-)registered since: walk the tables once; if these modules'
            # registrations are untouched, adopt the new epoch
            for owner, name, q, off in table:
                if owner.get(name) is not q:
                    return False
            self._param_epoch = _PARAM_EPOCH[0]
        # every parameter still points into the flat buffer (``p.data = ...`` cannot be intercepted: compare the addresses)
        return [q.data_ptr() for q in self._param_list] == self._param_ptrs

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._flatten_parameters()
        self._plans.clear()
        return out

    @property
    def flat_parameters(self) -> Tensor:
        """The [P] fp32 buffer every parameter is a view of (state_dict order)."""
        if not self._flat_is_current():
            self._flatten_parameters()
        return self._flat

    # ---- storage type of the activations (BASELINE config 3) ------------------------------------
    def set_compute_dtype(self, dtype: str) -> "GATResMeanConv":
        """``"fp32"`` (default: exact fp32 everywhere, the 1e-5 parity path, fused per-snapshot kernels where they apply)
        or ``"bf16"``: activation-sized tensors are stored as bf16 between kernels and the projections run on bf16 MFMA
        with fp32 accumulation; attention logits, softmax, neighbour sums, parameter gradients, the master parameters
        (this module's fp32 nn.Parameters) and Adam stay fp32.  The reference has no reduced-precision mode
        (SURVEY.md F1); predictions then agree with fp32 to about 1e-2 relative.  Per-op kernels, nc >= 32."""
        code = {"fp32": _native.DTYPE_F32, "float32": _native.DTYPE_F32, "bf16": _native.DTYPE_BF16,
                "bfloat16": _native.DTYPE_BF16}.get(str(dtype).replace("torch.", ""))
        if code is None:
            raise ValueError(f"unknown compute dtype {dtype!r}")
        if code == _native.DTYPE_BF16 and self.nc < 32:
            raise ValueError("bf16 projections need nc >= 32 (a 32-deep MFMA reduction)")
        self._cmodel.act_dtype = code
        self._plans.clear()
        self._call_states = {}
        return self

    @property
    def compute_dtype(self) -> str:
        return "bf16" if self._cmodel.act_dtype == _native.DTYPE_BF16 else "fp32"

    # ---- engine plumbing -------------------------------------------------------------------
    def _cmodel_ref(self):
        import ctypes
        return ctypes.byref(self._cmodel)

    def _saved_floats(self, plan: GraphPlan) -> int:
        lib = _native.load()
        n = lib.gatres_saved_floats(self._cmodel_ref(), plan.ref())
        if n < 0:
            _native.check(int(n), "gatres_saved_floats")
        return int(n)

    def _scratch_for(self, plan: GraphPlan) -> Tensor:
        key = (plan.num_nodes, plan.num_edges_gat, plan.num_segments, str(plan.device))
        buf = self._scratch.pop(key, None)
        if buf is not None:
            self._scratch[key] = buf          # most recently used last
        if buf is None:
            lib = _native.load()
            n = lib.gatres_scratch_floats(self._cmodel_ref(), plan.ref())
            if n < 0:
                _native.check(int(n), "gatres_scratch_floats")
            # zeroed once: the split-segment barrier epochs of the fused kernel live in here (include/gatres.h)
            buf = torch.zeros(int(n), dtype=torch.float32, device=plan.device)
            if buf.is_cuda:
                # once per device (cached in the library): does the dispatch put workgroups 8 ids apart on one XCD?  Then the
                # split launches start without their first cross-CU barrier (gatres_probe_xcd_dispatch)
                with torch.cuda.device(buf.device):
                    lib.gatres_probe_xcd_dispatch(_native.current_stream(buf.device))
            self._remember(key, buf)
        return buf

    def _remember(self, key, buf: Tensor, bound: int = 6) -> None:
        """Small LRU of per-plan work buffers: alternating plans (train / validation batch sizes, a ragged last batch)
        keep their buffers -- and the barrier state inside them -- instead of re-allocating on every switch."""
        self._scratch[key] = buf
        while len(self._scratch) > bound:
            self._scratch.pop(next(iter(self._scratch)))

    def _eval_saved_for(self, plan: GraphPlan) -> Tensor:
        key = ("eval_saved", plan.num_nodes, plan.num_edges_gat, plan.num_segments, str(plan.device))
        buf = self._scratch.get(key)
        if buf is None:
            buf = torch.empty(self._saved_floats(plan), dtype=torch.float32, device=plan.device)
            self._remember(key, buf)
        return buf

    def _call_state(self, plan: GraphPlan) -> dict:
        """What a forward / backward call of this model on ``plan`` needs besides the tensors, looked up ONCE per (plan, model
        configuration): the plan struct with this model's part tables (``GraphPlan.bound`` asks the library for the split on
        every call), whether inference takes the window kernel, the scratch buffer.  The reference's evaluation times every
        call with an event pair (utils/timer.py:22-41): what the host spends between the two records is part of the figure."""
        key = (id(plan), int(self._cmodel.act_dtype), bool(self.fused))
        st = self._call_states.get(key)
        if st is None or st["plan"] is not plan:
            lib = _native.load()
            import ctypes
            # (inference calls pass a copy of the model struct flagged GATRES_MODEL_INFERENCE: the window kernel then takes the
            #  instantiation without the saved-activation stores)
            infer = _native.GatresModel(self.num_blocks, self.nc, int(self._cmodel.act_dtype), _native.MODEL_INFERENCE)
            st = {"plan": plan, "gref": plan.ref(self._cmodel_ref()), "mref": self._cmodel_ref(),
                  "infer": infer, "mref_infer": ctypes.byref(infer),
                  "window": bool(lib.gatres_fused_window_kernel(self._cmodel_ref(), plan.ref())),
                  "saved_floats": self._saved_floats(plan)}
            if len(self._call_states) >= 8:
                self._call_states.clear()
            self._call_states[key] = st
        st["scratch"] = self._scratch_for(plan)      # (through the LRU every time: one scratch buffer per plan, whoever asks)
        return st

    def plan_for(self, edge_index: Tensor, num_nodes: int) -> GraphPlan:
        return self._plans.get(edge_index, num_nodes)

    def _run_forward(self, plan: GraphPlan, x: Tensor, keep: bool):
        """The native forward: (out, saved activations or None).  ``keep``: a backward will follow."""
        lib = _native.load()
        st = self._call_state(plan)
        out = torch.empty((plan.num_nodes, 1), dtype=torch.float32, device=x.device)
        saved = None
        if keep:
            saved = torch.empty(st["saved_floats"], dtype=torch.float32, device=x.device)
        elif st["window"]:
            # inference: the window kernel (taken only with a `saved` buffer) is ~17 % faster than the whole-segment-table
            # kernel even though it writes activations nobody reads -- give it a cached throw-away buffer
            saved = self._eval_saved_for(plan)
        _native.check(lib.gatres_model_forward(st["mref"] if keep else st["mref_infer"], st["gref"], self._flat.data_ptr(), x.data_ptr(), None,
                                               out.data_ptr(), _native.ptr(saved), st["scratch"].data_ptr(),
                                               _native.current_stream(x.device)), "gatres_model_forward")
        return out, saved

    # ---- reference surface -----------------------------------------------------------------
    def forward(self, x: Tensor, edge_index: Tensor, batch: Optional[Tensor] = None,
                edge_attr: Optional[Tensor] = None) -> Tensor:
        if edge_attr is not None:
            raise ValueError("GATRes is configured with use_data_edge_attrs=None (ConfigModels.py:38); "
                             "edge_attr must be None")
        if x.dim() != 2 or x.shape[1] != 1:
            raise ValueError(f"x must be [N, 1], got {tuple(x.shape)}")
        _native.require_gpu_tensor(x, "x")
        if not torch.is_grad_enabled() and self._flat is not None and self._flat.device == x.device:
            # Inference (evaluation.py:298,324-326: ``with torch.no_grad()``; every call timed by an event pair, utils/timer.py):
            # no autograd node, and the launch goes out BEFORE the walk over the 124 parameters that checks whether they still
            # are the views of the flat vector the kernels read -- the GPU works while the host checks.  A stale vector (a
            # parameter re-pointed or replaced since the last call: rare) is rebuilt and the forward simply runs again.
            plan = self._plans.get(edge_index, x.shape[0])
            out = self._run_forward(plan, x, False)[0]
            if self._flat_is_current():
                return out
            self._flatten_parameters()
            return self._run_forward(self._plans.get(edge_index, x.shape[0]), x, False)[0]
        if not self._flat_is_current():
            self._flatten_parameters()
        if self._flat.device != x.device:
            raise ValueError(f"model is on {self._flat.device}, x on {x.device}")
        plan = self._plans.get(edge_index, x.shape[0])
        params = self._param_list
        # grad mode is switched off inside Function.forward, so decide here whether activations must be kept
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params))
        if not needs_grad:
            return self._run_forward(plan, x, False)[0]          # (nothing requires a gradient: no autograd node at all)
        if self._direct_ok(params):
            return _GATResFunction.apply(self, plan, True, True, x, params[0])      # (params[0]: the anchor)
        if self._flat_param is not None:
            raise RuntimeError("flat mode (optimizer_parameters()) delivers gradients in place: it cannot be combined with "
                               "tensor hooks on, or frozen, named parameters")
        return _GATResFunction.apply(self, plan, True, False, x, *params)

"""torch.optim.Adam for the drop-in module in ONE kernel launch.

``train.py:348`` builds ``torch.optim.Adam(model.parameters(), lr, weight_decay)``; with the 182 parameter tensors
of gatres_small its ``step()`` costs about 1 ms of host time and a dozen multi-tensor launches.  The module's
parameters are views of one flat fp32 vector and ``loss.backward()`` delivers their gradients as views of one flat
buffer, so the update can be the native ``gatres_adam_step`` (the kernel ``GATResTrainer`` uses): same arithmetic
as ``torch.optim.Adam`` (L2 weight decay added to the gradient, bias correction, ``eps`` outside the square root).

    optimizer = gnn_pressure_estimation_amd.FusedAdam(model, lr=args.lr, weight_decay=args.weight_decay)

is the only line that changes in the reference's loop; ``zero_grad()`` / ``step()`` / ``state_dict()`` behave as usual.
"""
import torch

from . import _native
from .graph_models import GATResMeanConv


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, model: GATResMeanConv, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 0.0):
        if not isinstance(model, GATResMeanConv):
            raise TypeError("FusedAdam drives the flat parameter vector of a GATResMeanConv")
        params = list(model.parameters())
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self.lib = _native.load()
        flat = model.flat_parameters
        self._exp_avg = torch.zeros_like(flat)
        self._exp_avg_sq = torch.zeros_like(flat)
        self._counter = torch.zeros(2, dtype=torch.int64, device=flat.device)     # step, kernel ticket
        self._offsets = []
        off = 0
        for p in params:
            self._offsets.append(off)
            off += p.numel()
        self._total = off

    def _flat_grads(self, params):
        """(pointer, keep-alive) of the gradients as one contiguous fp32 buffer, or None if a gradient is missing.
        loss.backward() of the module hands them out as views of a single allocation in parameter order: then no copy
        is needed, only the base pointer."""
        fg = self.model.flat_grad()          # (gradients delivered in place by the module's backward: one identity check each)
        if fg is not None:
            return fg.data_ptr(), fg
        g0 = params[0].grad
        if g0 is None:
            return None
        base = g0.data_ptr()
        ok = True
        for p, off in zip(params, self._offsets):
            g = p.grad
            if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() != base + 4 * off:
                ok = False
                break
        if ok:
            return base, g0
        if any(p.grad is None for p in params):
            return None
        cat = torch.cat([p.grad.reshape(-1).to(torch.float32) for p in params])
        return cat.data_ptr(), cat

    def zero_grad(self, set_to_none: bool = True) -> None:
        super().zero_grad(set_to_none)
        fp = self.model.flat_parameter          # (a model in flat mode -- optimizer_parameters() was called -- keeps THE gradient there)
        if fp is not None and fp.grad is not None:
            if set_to_none:
                fp.grad = None
                self.model._grad_cur = None
            else:
                fp.grad.zero_()

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        params = group["params"]
        flat = self.model.flat_parameters
        if flat.numel() != self._total or flat.device != self._exp_avg.device:
            raise RuntimeError("the model's parameter storage changed; build a new FusedAdam")
        grads = self._flat_grads(params)
        if grads is None:
            return loss                     # nothing to do (torch.optim.Adam skips parameters without a gradient)
        gptr, _keep = grads
        b1, b2 = group["betas"]
        _native.check(self.lib.gatres_adam_step(flat.data_ptr(), gptr, self._exp_avg.data_ptr(),
                                                self._exp_avg_sq.data_ptr(), self._counter.data_ptr(), self._total,
                                                float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                float(group["weight_decay"]), 1.0,
                                                _native.current_stream(flat.device)), "gatres_adam_step")
        torch.autograd.graph.increment_version(flat)       # the storage changed behind torch's back
        return loss

    # the moments live in two flat vectors rather than per-parameter state entries
    def state_dict(self):
        sd = super().state_dict()
        sd["fused"] = {"exp_avg": self._exp_avg.clone(), "exp_avg_sq": self._exp_avg_sq.clone(),
                       "step": int(self._counter[0].item())}
        return sd

    def load_state_dict(self, state_dict):
        fused = state_dict.get("fused")
        rest = {k: v for k, v in state_dict.items() if k != "fused"}
        super().load_state_dict(rest)
        if fused is not None:
            self._exp_avg.copy_(fused["exp_avg"])
            self._exp_avg_sq.copy_(fused["exp_avg_sq"])
            self._counter.zero_()
            self._counter[0] = int(fused["step"])

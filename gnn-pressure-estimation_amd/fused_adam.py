"""torch.optim.Adam for the drop-in module in ONE kernel launch.

``train.py:348`` builds ``torch.optim.Adam(model.parameters(), lr, weight_decay)``; with the 182 parameter tensors
of gatres_small its ``step()`` costs about 1 ms of host time and a dozen multi-tensor launches.  The module's
parameters are views of one flat fp32 vector and ``loss.backward()`` writes their gradients into one flat buffer
(``model.flat_parameter.grad``: the optimizer holds that ONE parameter and puts the model in flat mode,
``GATResMeanConv.optimizer_parameters``), so the update can be the native ``gatres_adam_step`` (the kernel ``GATResTrainer`` uses): same arithmetic
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
        # The optimizer holds ONE parameter, the model's flat leaf (``model.optimizer_parameters()``: the model is then in flat
        # mode -- loss.backward() delivers the gradient as ``flat_parameter.grad``, the named parameters' ``.grad`` stay
        # attached as views of the same buffer): ``zero_grad()`` and the gradient hand-over touch one tensor, not 124.
        params = model.optimizer_parameters()
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.model = model
        self.lib = _native.load()
        flat = model.flat_parameters
        self._exp_avg = torch.zeros_like(flat)
        self._exp_avg_sq = torch.zeros_like(flat)
        self._counter = torch.zeros(2, dtype=torch.int64, device=flat.device)     # step, kernel ticket
        self._total = flat.numel()
        self._named = list(model.parameters())
        self._offsets = []
        off = 0
        for p in self._named:
            self._offsets.append(off)
            off += p.numel()

    def _flat_grads(self):
        """(pointer, keep-alive) of the gradient as one contiguous fp32 buffer, or None if there is none.  Normally it is
        ``flat_parameter.grad`` -- the buffer the module's backward wrote in place, of which the named parameters' ``.grad``
        are views.  A named gradient that was REPLACED by another tensor since (``p.grad = clipped``) is honoured: its values
        go into a copy of the buffer.  Gradients that arrived on the named parameters only (a model that left flat mode's
        in-place path) are concatenated."""
        model, named = self.model, self._named
        fp = model.flat_parameter
        g = fp.grad if fp is not None else None
        own = model._grad_views[0] if model._grad_views is not None else None
        if g is not None:
            if g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != self._total:
                g = g.to(torch.float32).contiguous().reshape(-1)
            if own is not None:
                replaced = [i for i, (p, v) in enumerate(zip(named, own)) if p.grad is not v and p.grad is not None]
                if replaced:
                    g = g.clone()
                    for i in replaced:
                        off, q = self._offsets[i], named[i]
                        g[off:off + q.numel()].copy_(q.grad.reshape(-1))
            return g.data_ptr(), g
        grads = [p.grad for p in named]
        if own is not None:                      # (views the module attached earlier are not gradients of THIS step)
            grads = [None if gg is v else gg for gg, v in zip(grads, own)]
        if any(gg is None for gg in grads):
            return None                          # nothing to do (torch.optim.Adam skips parameters without a gradient)
        cat = torch.cat([gg.reshape(-1).to(torch.float32) for gg in grads])
        return cat.data_ptr(), cat

    def zero_grad(self, set_to_none: bool = True) -> None:
        super().zero_grad(set_to_none)           # (the flat parameter: THE gradient of a model in flat mode)
        own = self.model._grad_views[0] if self.model._grad_views is not None else None
        for i, p in enumerate(self._named):      # gradients that sit on the named parameters another way (see _flat_grads)
            g = p.grad
            if g is not None and (own is None or g is not own[i]):
                if set_to_none:
                    p.grad = None
                else:
                    g.zero_()
        if set_to_none and own is not None and any(p.grad is None for p in self._named):
            self.model._grad_cur = None          # (a replaced view was dropped: the next backward re-attaches the views)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        group = self.param_groups[0]
        flat = self.model.flat_parameters
        if (flat.numel() != self._total or flat.device != self._exp_avg.device or self.model.flat_parameter is None
                or self.model.flat_parameter is not group["params"][0]):
            raise RuntimeError("the model's parameter storage changed; build a new FusedAdam")
        grads = self._flat_grads()
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

"""CPU oracle for the GATRes hot path  --  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference (DiTEC-project/gnn-pressure-estimation) ships no
tests, golden vectors or weights for this path, and the arithmetic lives in an
un-vendored, un-pinned third-party dependency (torch_geometric >= 2.3, README.md:31)
that is not installed in the build container.  This file restates the published
PyG 2.3.x algorithm of the ops the reference composes; it is pinned only by the
hand-derived known-answer tests in tests/test_oracle_kat.py and by an independent
dense-adjacency restatement (``gat_conv_dense`` below).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
module.  The product (gnn-pressure-estimation_amd/) never does.

What is restated (reference call sites -> PyG op):
  * GATResMeanConv.forward            gnn_pressure_estimation/GraphModels.py:486-494
  * GResBlockMeanConv.forward         gnn_pressure_estimation/GraphModels.py:462-468
  * GATConv(in, out, heads, concat)   constructed GraphModels.py:458-459, called :464-465
        torch_geometric.nn.conv.gat_conv.GATConv.forward / edge_update / message
        torch_geometric.utils.softmax, remove_self_loops, add_self_loops
  * SimpleConv(aggr="mean")           constructed GraphModels.py:460, called :466
  * Linear                            GraphModels.py:477,484 (torch_geometric.nn.dense.linear)
  * the training step around it       gnn_pressure_estimation/train.py:159-190
  * mask_nodes / generate_batch_mask  gnn_pressure_estimation/utils/auxil.py:143-182

Everything is written with plain torch CPU ops (index_select / index_add_ /
scatter_reduce / mm), works in fp32 (the parity target) and fp64 (gradcheck), and is
differentiated by torch autograd exactly as the reference differentiates PyG's ops
(train.py:185).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

NEG_SLOPE = 0.2          # GATConv default negative_slope
SOFTMAX_EPS = 1e-16      # torch_geometric.utils.softmax: out_sum + 1e-16


# ----------------------------------------------------------------------------------------------
# parameters (names follow PyG's state_dict so reference checkpoints map 1:1, train.py:433-436)
# ----------------------------------------------------------------------------------------------
def param_shapes(num_blocks: int, nc: int) -> "OrderedDict[str, Tuple[int, ...]]":
    """state_dict key -> shape, in registration order (GraphModels.py:472-484, :455-460)."""
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    shapes["lin0.weight"] = (nc, 1)
    shapes["lin0.bias"] = (nc,)
    for i in range(num_blocks):
        p = f"blocks.{i}."
        shapes[p + "conv1.att_src"] = (1, 2, nc)
        shapes[p + "conv1.att_dst"] = (1, 2, nc)
        shapes[p + "conv1.bias"] = (2 * nc,)
        shapes[p + "conv1.lin_src.weight"] = (2 * nc, nc)
        shapes[p + "conv2.att_src"] = (1, 1, nc)
        shapes[p + "conv2.att_dst"] = (1, 1, nc)
        shapes[p + "conv2.bias"] = (nc,)
        shapes[p + "conv2.lin_src.weight"] = (nc, 2 * nc)
    shapes["lin1.weight"] = (1, nc)
    shapes["lin1.bias"] = (1,)
    return shapes


def num_params(num_blocks: int, nc: int) -> int:
    return sum(int(np.prod(s)) for s in param_shapes(num_blocks, nc).values())


def init_params(num_blocks: int, nc: int, seed: int = 0, dtype=torch.float32) -> "OrderedDict[str, torch.Tensor]":
    """PyG-style initialisation.

    lin0 / lin1 : torch_geometric Linear default = kaiming_uniform(a=sqrt(5)) -> U(+-1/sqrt(fan_in))
                  for the weight and U(+-1/sqrt(fan_in)) for the bias.
    GATConv.lin : glorot, U(+-sqrt(6/(fan_in+fan_out))), no bias.
    att_src/dst : glorot on a [1,H,C] tensor -> U(+-sqrt(6/(H+C))).
    GATConv.bias: zeros.
    """
    g = torch.Generator().manual_seed(seed)
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()

    def uni(shape, bound):
        return ((torch.rand(shape, generator=g, dtype=torch.float64) * 2 - 1) * bound).to(dtype)

    for name, shape in param_shapes(num_blocks, nc).items():
        if name.startswith("lin0") or name.startswith("lin1"):
            fan_in = 1 if name.startswith("lin0") else nc
            out[name] = uni(shape, 1.0 / math.sqrt(fan_in))
        elif name.endswith("lin_src.weight"):
            out[name] = uni(shape, math.sqrt(6.0 / (shape[0] + shape[1])))
        elif name.endswith("att_src") or name.endswith("att_dst"):
            out[name] = uni(shape, math.sqrt(6.0 / (shape[1] + shape[2])))
        else:  # GATConv bias
            out[name] = torch.zeros(shape, dtype=dtype)
    return out


# ----------------------------------------------------------------------------------------------
# PyG utilities
# ----------------------------------------------------------------------------------------------
def remove_self_loops(edge_index: torch.Tensor) -> torch.Tensor:
    """torch_geometric.utils.remove_self_loops: keep edges with src != dst, order preserved."""
    keep = edge_index[0] != edge_index[1]
    return edge_index[:, keep]


def add_self_loops(edge_index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """torch_geometric.utils.add_self_loops: append (i, i) for i = 0..N-1 AFTER the existing edges."""
    loop = torch.arange(num_nodes, dtype=edge_index.dtype)
    return torch.cat([edge_index, torch.stack([loop, loop])], dim=1)


def segment_softmax(src: torch.Tensor, index: torch.Tensor, num_nodes: int) -> torch.Tensor:
    """torch_geometric.utils.softmax(src, index, num_nodes=N) for src [E, H]."""
    H = src.shape[1]
    idx = index.view(-1, 1).expand(-1, H)
    src_max = torch.full((num_nodes, H), float("-inf"), dtype=src.dtype)
    src_max = src_max.scatter_reduce(0, idx, src.detach(), reduce="amax", include_self=True)
    src_max = torch.where(torch.isinf(src_max), torch.zeros_like(src_max), src_max)  # empty segments -> 0
    out = (src - src_max.index_select(0, index)).exp()
    out_sum = torch.zeros((num_nodes, H), dtype=src.dtype).index_add_(0, index, out) + SOFTMAX_EPS
    return out / out_sum.index_select(0, index)


# ----------------------------------------------------------------------------------------------
# ops
# ----------------------------------------------------------------------------------------------
def gat_conv(x, edge_index, weight, att_src, att_dst, bias, heads: int, concat: bool,
             return_alpha: bool = False):
    """PyG GATConv.forward with add_self_loops=True, edge_dim=None, dropout=0 (GraphModels.py:458-459)."""
    N = x.shape[0]
    H = heads
    C = weight.shape[0] // H
    h = (x @ weight.t()).view(N, H, C)                       # lin_src (shared with lin_dst)
    a_src = (h * att_src).sum(dim=-1)                        # [N, H]
    a_dst = (h * att_dst).sum(dim=-1)
    ei = add_self_loops(remove_self_loops(edge_index), N)    # E' = E_noloop + N
    src, dst = ei[0], ei[1]                                  # flow source_to_target: j = src, i = dst
    s = a_src.index_select(0, src) + a_dst.index_select(0, dst)
    s = torch.nn.functional.leaky_relu(s, NEG_SLOPE)
    alpha = segment_softmax(s, dst, N)                       # [E', H]
    msg = alpha.unsqueeze(-1) * h.index_select(0, src)       # [E', H, C]
    out = torch.zeros((N, H, C), dtype=x.dtype).index_add_(0, dst, msg)
    out = out.reshape(N, H * C) if concat else out.mean(dim=1)
    out = out + bias
    if return_alpha:
        return out, alpha, ei
    return out


def gat_conv_dense(x, edge_index, weight, att_src, att_dst, bias, heads: int, concat: bool):
    """Independent restatement through a dense (multiplicity) adjacency; used to cross-check gat_conv."""
    N = x.shape[0]
    H = heads
    C = weight.shape[0] // H
    h = (x @ weight.t()).view(N, H, C)
    a_src = torch.einsum("nhc,hc->nh", h, att_src[0])
    a_dst = torch.einsum("nhc,hc->nh", h, att_dst[0])
    cnt = torch.zeros((N, N), dtype=x.dtype)                 # cnt[i, j] = multiplicity of edge j -> i
    ei = edge_index[:, edge_index[0] != edge_index[1]]
    cnt.index_put_((ei[1], ei[0]), torch.ones(ei.shape[1], dtype=x.dtype), accumulate=True)
    cnt = cnt + torch.eye(N, dtype=x.dtype)
    s = a_dst.unsqueeze(1) + a_src.unsqueeze(0)              # s[i, j, h]
    s = torch.where(s > 0, s, NEG_SLOPE * s)
    s_masked = torch.where(cnt.unsqueeze(-1) > 0, s, torch.full_like(s, float("-inf")))
    m = s_masked.max(dim=1, keepdim=True).values.detach()
    p = cnt.unsqueeze(-1) * torch.exp(s_masked - m)
    alpha = p / (p.sum(dim=1, keepdim=True) + SOFTMAX_EPS)   # [i, j, h], already includes multiplicity
    out = torch.einsum("ijh,jhc->ihc", alpha, h)
    out = out.reshape(N, H * C) if concat else out.mean(dim=1)
    return out + bias


def simple_conv_mean(x, edge_index):
    """PyG SimpleConv(aggr='mean'): mean of x[src] over the ORIGINAL edges into dst; no self loops added,
    existing ones kept; destinations without in-edges get 0 (count clamped to 1)."""
    N = x.shape[0]
    src, dst = edge_index[0], edge_index[1]
    out = torch.zeros_like(x).index_add_(0, dst, x.index_select(0, src))
    count = torch.zeros(N, dtype=x.dtype).index_add_(0, dst, torch.ones(dst.shape[0], dtype=x.dtype))
    return out / count.clamp(min=1).unsqueeze(-1)


def gres_block(x, edge_index, p: Dict[str, torch.Tensor], prefix: str):
    """GResBlockMeanConv.forward, GraphModels.py:462-468."""
    x0 = x
    x = gat_conv(x, edge_index, p[prefix + "conv1.lin_src.weight"], p[prefix + "conv1.att_src"],
                 p[prefix + "conv1.att_dst"], p[prefix + "conv1.bias"], heads=2, concat=True).relu()
    x = gat_conv(x, edge_index, p[prefix + "conv2.lin_src.weight"], p[prefix + "conv2.att_src"],
                 p[prefix + "conv2.att_dst"], p[prefix + "conv2.bias"], heads=1, concat=False)
    x = simple_conv_mean(x, edge_index) + x0
    return x.relu()


def gatres_forward(p: Dict[str, torch.Tensor], x, edge_index, num_blocks: Optional[int] = None,
                   conv=gat_conv):
    """GATResMeanConv.forward(x, edge_index, batch=None, edge_attr=None), GraphModels.py:486-494."""
    if num_blocks is None:
        num_blocks = sum(1 for k in p if k.endswith("conv1.bias"))
    x = x @ p["lin0.weight"].t() + p["lin0.bias"]
    for i in range(num_blocks):
        if conv is gat_conv:
            x = gres_block(x, edge_index, p, f"blocks.{i}.")
        else:
            pre = f"blocks.{i}."
            x0 = x
            x = conv(x, edge_index, p[pre + "conv1.lin_src.weight"], p[pre + "conv1.att_src"],
                     p[pre + "conv1.att_dst"], p[pre + "conv1.bias"], 2, True).relu()
            x = conv(x, edge_index, p[pre + "conv2.lin_src.weight"], p[pre + "conv2.att_src"],
                     p[pre + "conv2.att_dst"], p[pre + "conv2.bias"], 1, False)
            x = (simple_conv_mean(x, edge_index) + x0).relu()
    return x @ p["lin1.weight"].t() + p["lin1.bias"]


# ----------------------------------------------------------------------------------------------
# bf16 storage mode (BASELINE config 3): the same model with the build's rounding points
# ----------------------------------------------------------------------------------------------
# The reference has no reduced-precision mode; this section restates what the BUILD's bf16 mode does (csrc/model_driver.hip,
# gatres_model_t.act_dtype) so that its whole-model tests can tell rounding from a bug: activation-sized tensors and their
# gradients are STORED as bf16 (round to nearest even), conv weights are read as bf16 copies of the fp32 masters, and
# everything else -- accumulators, attention logits (taken from the fp32 accumulators BEFORE h is rounded), alpha, softmax,
# x / y / out, parameter gradients, Adam -- stays fp32.
class _StoreBF16(torch.autograd.Function):
    """A tensor that lives in HBM as bf16: the forward value is rounded, and so is the gradient that arrives for it
    (the kernels store g_pre / g_y2 / g_h / g_out1 as bf16 after summing every contribution in fp32)."""

    @staticmethod
    def forward(ctx, t, round_value, round_grad):
        ctx.round_grad = round_grad
        return t.to(torch.bfloat16).to(t.dtype) if round_value else t.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(torch.bfloat16).to(g.dtype) if ctx.round_grad else g), None, None


def _store(t):          # value and gradient rounded
    return _StoreBF16.apply(t, True, True)


def _grad_store(t):     # gradient rounded, value kept (the fp32 accumulators the attention logits are formed from)
    return _StoreBF16.apply(t, False, True)


def _value_store(t):    # value rounded, gradient kept (bf16 copies of the fp32 master weights; h for the messages)
    return _StoreBF16.apply(t, True, False)


def gat_conv_bf16(x, edge_index, weight, att_src, att_dst, bias, heads: int, concat: bool):
    """gat_conv with the build's bf16 rounding points: x arrives rounded; h_acc = x @ bf16(W)^T in fp32; the logits use
    h_acc; the messages use bf16(h_acc); the gradient of h_acc (message path + logit path, summed in fp32) is stored as
    bf16 (gatres_t_gat_aggregate_bwd_src's g_h).  The caller rounds the output."""
    N = x.shape[0]
    H = heads
    C = weight.shape[0] // H
    h_acc = _grad_store(x @ _value_store(weight).t()).view(N, H, C)
    a_src = (h_acc * att_src).sum(dim=-1)
    a_dst = (h_acc * att_dst).sum(dim=-1)
    h = _value_store(h_acc)
    ei = add_self_loops(remove_self_loops(edge_index), N)
    src, dst = ei[0], ei[1]
    s = torch.nn.functional.leaky_relu(a_src.index_select(0, src) + a_dst.index_select(0, dst), NEG_SLOPE)
    alpha = segment_softmax(s, dst, N)
    out = torch.zeros((N, H, C), dtype=x.dtype).index_add_(0, dst, alpha.unsqueeze(-1) * h.index_select(0, src))
    out = out.reshape(N, H * C) if concat else out.mean(dim=1)
    return out + bias


def gatres_forward_bf16(p: Dict[str, torch.Tensor], x, edge_index, num_blocks: Optional[int] = None):
    """GATResMeanConv.forward in the build's bf16 storage mode (fp32 arithmetic, bf16 rounding where the kernels store
    bf16: lin0's output, h1, relu(conv1), h2, conv2's output, every block output, and the gradients of those)."""
    if num_blocks is None:
        num_blocks = sum(1 for k in p if k.endswith("conv1.bias"))
    x = _store(x @ p["lin0.weight"].t() + p["lin0.bias"])
    for i in range(num_blocks):
        pre = f"blocks.{i}."
        x0 = x
        o1 = _store(gat_conv_bf16(x, edge_index, p[pre + "conv1.lin_src.weight"], p[pre + "conv1.att_src"],
                                  p[pre + "conv1.att_dst"], p[pre + "conv1.bias"], 2, True)).relu()
        y2 = _store(gat_conv_bf16(o1, edge_index, p[pre + "conv2.lin_src.weight"], p[pre + "conv2.att_src"],
                                  p[pre + "conv2.att_dst"], p[pre + "conv2.bias"], 1, False))
        x = _store(simple_conv_mean(y2, edge_index) + x0).relu()
    return x @ p["lin1.weight"].t() + p["lin1.bias"]


# ----------------------------------------------------------------------------------------------
# caller side: mask, loss, optimiser  (train.py:159-190, auxil.py:143-182)
# ----------------------------------------------------------------------------------------------
def mask_nodes(num_nodes: int, masking_rate: float, rng: np.random.RandomState) -> np.ndarray:
    """auxil.py:143-163 with required_idx=[]: exactly int(n*rate) nodes, without replacement."""
    mask_length = int(num_nodes * masking_rate)
    assert mask_length > 0
    idx = rng.choice(num_nodes, mask_length, replace=False)
    mask = np.zeros(num_nodes, dtype=bool)
    mask[idx] = True
    return mask


def generate_batch_mask(num_nodes_per_graph, mask_rate: float, rng: np.random.RandomState) -> np.ndarray:
    """auxil.py:166-182: hstack of one mask per graph."""
    return np.hstack([mask_nodes(int(n), mask_rate, rng) for n in num_nodes_per_graph])


def masked_mse(out, y, mask):
    """criterion(out[mask], y[mask]) with MSELoss(mean), train.py:177-183 (descale only feeds metrics)."""
    m = torch.as_tensor(mask, dtype=torch.bool)
    return torch.nn.functional.mse_loss(out[m], y[m])


def train_step(p: Dict[str, torch.Tensor], opt_state: Optional[dict], x, y, edge_index, mask,
               lr: float = 5e-4, weight_decay: float = 6e-6, betas=(0.9, 0.999), eps: float = 1e-8):
    """One reference training iteration, train.py:159-190: x[mask]=0 -> forward -> MSE on masked nodes ->
    backward -> Adam (torch.optim.Adam semantics, L2 weight decay folded into the gradient, train.py:348).

    Returns (loss, out, grads, new_params, new_opt_state)."""
    params = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in p.items())
    m = torch.as_tensor(mask, dtype=torch.bool)
    xin = x.clone()
    xin[m] = 0
    out = gatres_forward(params, xin, edge_index)
    loss = masked_mse(out, y, m)
    grads = torch.autograd.grad(loss, list(params.values()))
    grads = OrderedDict(zip(params.keys(), grads))
    if opt_state is None:
        opt_state = {"step": 0, "m": {k: torch.zeros_like(v) for k, v in p.items()},
                     "v": {k: torch.zeros_like(v) for k, v in p.items()}}
    step = opt_state["step"] + 1
    b1, b2 = betas
    new_p, new_m, new_v = OrderedDict(), {}, {}
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    for k, w in p.items():
        g = grads[k] + weight_decay * w
        mk = b1 * opt_state["m"][k] + (1 - b1) * g
        vk = b2 * opt_state["v"][k] + (1 - b2) * g * g
        denom = vk.sqrt() / math.sqrt(bc2) + eps
        new_p[k] = w - (lr / bc1) * (mk / denom)
        new_m[k], new_v[k] = mk, vk
    return loss.detach(), out.detach(), grads, new_p, {"step": step, "m": new_m, "v": new_v}


class OracleTrainer:
    """The reference loop body with the reference's own optimiser object: torch.optim.Adam(params, lr, weight_decay)
    (train.py:348), MSELoss (train.py:364), one ``step`` = train.py:160-188.  Used as the CPU baseline and as the
    multi-step parity checker."""

    def __init__(self, params: Dict[str, torch.Tensor], lr: float = 5e-4, weight_decay: float = 6e-6, bf16: bool = False):
        self.params = OrderedDict((k, v.detach().clone().requires_grad_(True)) for k, v in params.items())
        self.opt = torch.optim.Adam(list(self.params.values()), lr=lr, weight_decay=weight_decay)
        self.forward = gatres_forward_bf16 if bf16 else gatres_forward      # (bf16: the build's storage mode, above)

    def step(self, x, y, edge_index, mask):
        self.opt.zero_grad()
        m = torch.as_tensor(mask, dtype=torch.bool)
        xin = x.clone()
        xin[m] = 0
        out = self.forward(self.params, xin, edge_index)
        loss = torch.nn.functional.mse_loss(out[m], y[m])
        loss.backward()
        self.opt.step()
        return loss.detach(), out.detach()

    def grads(self):
        return OrderedDict((k, v.grad.detach().clone()) for k, v in self.params.items())

    def flat(self, what: str = "params") -> torch.Tensor:
        src = self.params.values() if what == "params" else [v.grad for v in self.params.values()]
        return torch.cat([t.detach().reshape(-1) for t in src])


def flatten(p: Dict[str, torch.Tensor]) -> torch.Tensor:
    """state_dict-ordered flat vector (the layout of include/gatres.h)."""
    return torch.cat([v.detach().reshape(-1) for v in p.values()])

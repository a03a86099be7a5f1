"""Thin test-side wrappers: torch tensors -> C-ABI calls of include/gatres.h (one kernel each)."""
import ctypes as C

import torch

import gnn_pressure_estimation_amd as G

N_ = G._native


def _s(t):
    return N_.current_stream(t.device)


def f32(*shape, device="cuda"):
    return torch.empty(shape, dtype=torch.float32, device=device)


def lin0_fwd(x, w, b, mask=None):
    lib = N_.load()
    n, nc = x.numel(), w.numel()
    out = f32(n, nc)
    N_.check(lib.gatres_lin0_fwd(x.data_ptr(), N_.ptr(mask), w.data_ptr(), b.data_ptr(), out.data_ptr(), n, nc, _s(x)),
             "lin0_fwd")
    return out


def proj_attn_fwd(x, W, att_src, att_dst, H):
    lib = N_.load()
    n, K = x.shape
    C_ = W.shape[0] // H
    h, a_s, a_d = f32(n, H * C_), f32(n, H), f32(n, H)
    N_.check(lib.gatres_proj_attn_fwd(x.data_ptr(), W.data_ptr(), att_src.data_ptr(), att_dst.data_ptr(), h.data_ptr(),
                                      a_s.data_ptr(), a_d.data_ptr(), n, K, H, C_, _s(x)), "proj_attn_fwd")
    return h, a_s, a_d


def gat_aggregate_fwd(plan, h, a_s, a_d, bias, H, relu):
    lib = N_.load()
    n, HC = h.shape
    out, alpha = f32(n, HC), f32(plan.num_edges_gat, H)
    N_.check(lib.gatres_gat_aggregate_fwd(plan.ref(), h.data_ptr(), a_s.data_ptr(), a_d.data_ptr(), bias.data_ptr(),
                                          out.data_ptr(), alpha.data_ptr(), H, HC // H, int(relu), _s(h)), "agg_fwd")
    return out, alpha


def mean_residual_relu_fwd(plan, y, x0):
    lib = N_.load()
    out = torch.empty_like(y)
    N_.check(lib.gatres_mean_residual_relu_fwd(plan.ref(), y.data_ptr(), x0.data_ptr(), out.data_ptr(), y.shape[1],
                                               _s(y)), "mean_fwd")
    return out


def lin1_fwd(x, w, b):
    lib = N_.load()
    n, nc = x.shape
    out = f32(n)
    N_.check(lib.gatres_lin1_fwd(x.data_ptr(), w.data_ptr(), N_.ptr(b), out.data_ptr(), n, nc, _s(x)), "lin1_fwd")
    return out


def mean_bwd(plan, g_pre):
    lib = N_.load()
    g_y = torch.empty_like(g_pre)
    N_.check(lib.gatres_mean_bwd(plan.ref(), g_pre.data_ptr(), g_y.data_ptr(), g_pre.shape[1], _s(g_pre)), "mean_bwd")
    return g_y


def gat_aggregate_bwd(plan, g_out, h, alpha, a_s, a_d, att_src, att_dst, H):
    lib = N_.load()
    n, HC = h.shape
    g_e, g_ad, g_as, g_h = f32(plan.num_edges_gat, H), f32(n, H), f32(n, H), f32(n, HC)
    N_.check(lib.gatres_gat_aggregate_bwd_dst(plan.ref(), g_out.data_ptr(), h.data_ptr(), alpha.data_ptr(),
                                              a_s.data_ptr(), a_d.data_ptr(), g_e.data_ptr(), g_ad.data_ptr(), H,
                                              HC // H, _s(h)), "agg_bwd_dst")
    N_.check(lib.gatres_gat_aggregate_bwd_src(plan.ref(), g_out.data_ptr(), alpha.data_ptr(), g_e.data_ptr(),
                                              g_ad.data_ptr(), att_src.data_ptr(), att_dst.data_ptr(), g_h.data_ptr(),
                                              g_as.data_ptr(), H, HC // H, _s(h)), "agg_bwd_src")
    return g_h, g_as, g_ad, g_e


def proj_bwd_dx(g_h, Wt, resid=None, relu_ref=None):
    lib = N_.load()
    n, HC = g_h.shape
    K = Wt.shape[0]
    g_x = f32(n, K)
    N_.check(lib.gatres_proj_bwd_dx(g_h.data_ptr(), Wt.data_ptr(), N_.ptr(resid), N_.ptr(relu_ref), g_x.data_ptr(), n,
                                    K, HC, _s(g_h)), "proj_bwd_dx")
    return g_x


def proj_bwd_dw(g_h, x, num_slabs):
    lib = N_.load()
    n, HC = g_h.shape
    K = x.shape[1]
    slabs = torch.full((num_slabs, HC * K), float("nan"), dtype=torch.float32, device=g_h.device)
    N_.check(lib.gatres_proj_bwd_dw(g_h.data_ptr(), x.data_ptr(), slabs.data_ptr(), num_slabs, HC * K, n, K, HC,
                                    _s(g_h)), "proj_bwd_dw")
    out = f32(HC * K)
    N_.check(lib.gatres_reduce_slabs(slabs.data_ptr(), num_slabs, HC * K, HC * K, out.data_ptr(), _s(g_h)), "reduce")
    return out.view(HC, K)


def conv_param_grads(h, g_as, g_ad, g_out, H, num_slabs):
    lib = N_.load()
    n, HC = h.shape
    slabs = torch.full((num_slabs, 3 * HC), float("nan"), dtype=torch.float32, device=h.device)
    base = slabs.data_ptr()
    N_.check(lib.gatres_conv_param_grads(h.data_ptr(), g_as.data_ptr(), g_ad.data_ptr(), g_out.data_ptr(), base,
                                         base + 4 * HC, base + 8 * HC, num_slabs, 3 * HC, n, H, HC // H, _s(h)),
             "conv_param_grads")
    out = f32(3 * HC)
    N_.check(lib.gatres_reduce_slabs(slabs.data_ptr(), num_slabs, 3 * HC, 3 * HC, out.data_ptr(), _s(h)), "reduce")
    return out[:HC], out[HC:2 * HC], out[2 * HC:]


def mask_generate(node_ptr, num_graphs, rate, seed, step_counter, n):
    lib = N_.load()
    mask = torch.zeros(n, dtype=torch.uint8, device=node_ptr.device)
    N_.check(lib.gatres_mask_generate(node_ptr.data_ptr(), num_graphs, rate, seed, N_.ptr(step_counter),
                                      mask.data_ptr(), _s(node_ptr)), "mask_generate")
    return mask


def masked_mse(out, y, mask):
    lib = N_.load()
    loss, g = f32(1), f32(out.numel())
    N_.check(lib.gatres_masked_mse(out.data_ptr(), y.data_ptr(), mask.data_ptr(), loss.data_ptr(), g.data_ptr(),
                                   out.numel(), _s(out)), "masked_mse")
    return loss, g


def adam_step(p, g, m, v, step_counter, lr, b1, b2, eps, wd, scale=1.0):
    lib = N_.load()
    N_.check(lib.gatres_adam_step(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), step_counter.data_ptr(),
                                  p.numel(), lr, b1, b2, eps, wd, scale, _s(p)), "adam")

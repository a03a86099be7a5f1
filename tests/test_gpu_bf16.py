"""bf16 storage + bf16 MFMA projections (gatres_model_t.act_dtype = GATRES_DTYPE_BF16; BASELINE.json config 3).

The reference has no reduced-precision mode; the bar SURVEY 8(d) sets for config 3 is ~1e-2 relative on
the predictions against the fp32 oracle plus a loss trajectory that follows fp32 training.  Written tolerances:
  * single kernels against torch on the SAME bf16-rounded inputs: outputs within bf16 rounding (2^-8 relative), fp32
    side outputs (attention logits, parameter-gradient slabs) within 2e-4 of an fp32 matmul of those inputs;
  * whole model vs the fp32 oracle: predictions <= 1e-2 relative in the L2 norm (<= 2.5e-2 at the worst node); the flat
    gradient within 1.5e-1 (L2) -- 15 to 25 blocks of bf16 rounding, on gradients that are themselves cancellation-heavy;
  * 4 training steps: the first loss within 5e-3 of the fp32 trainer's, every later one within 1e-1 (the fp32 run itself
    moves 0.97 -> 2.13 -> 1.03 -> 1.58 over these steps on fresh batches: differences are amplified, not damped)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_model import build, ctown_batch, note, relerr      # noqa: E402

BF16 = 1


def _st(t):
    import gnn_pressure_estimation_amd as G
    return G._native.current_stream(t.device)


@pytest.mark.parametrize("nc,H", [(32, 2), (32, 1), (64, 2), (128, 2), (128, 1)])
def test_bf16_projection_kernels(pkg, lib, nc, H):
    """v_mfma_f32_16x16x32_bf16 projection (forward with the attention-logit epilogue, data gradient with the residual /
    ReLU-mask epilogue) and the fp32-accumulated weight gradient on bf16 operands, against torch."""
    N = 1000 + 7                                         # a ragged last tile
    K = nc if H == 2 else 2 * nc
    C, HC = nc, H * nc
    g = torch.Generator().manual_seed(nc * 10 + H)
    x = torch.randn(N, K, generator=g).cuda().bfloat16()
    W = (torch.randn(HC, K, generator=g) / K ** 0.5).cuda().bfloat16()
    att_s, att_d = torch.randn(HC, generator=g).cuda(), torch.randn(HC, generator=g).cuda()
    h = torch.empty(N, HC, dtype=torch.bfloat16, device="cuda")
    a_s, a_d = torch.empty(N, H, device="cuda"), torch.empty(N, H, device="cuda")
    pkg._native.check(lib.gatres_t_proj_attn_fwd(x.data_ptr(), W.data_ptr(), att_s.data_ptr(), att_d.data_ptr(),
                                                 h.data_ptr(), a_s.data_ptr(), a_d.data_ptr(), N, K, H, C, BF16, _st(x)),
                      "proj_attn_fwd bf16")
    h_ref = x.float() @ W.float().t()                    # fp32 accumulation of the same bf16 operands
    assert relerr(h.float(), h_ref) < 2 ** -8
    assert relerr(a_s, (h_ref.view(N, H, C) * att_s.view(1, H, C)).sum(-1)) < 2e-4       # logits: from the fp32 accumulators
    assert relerr(a_d, (h_ref.view(N, H, C) * att_d.view(1, H, C)).sum(-1)) < 2e-4
    # data gradient: g_x = (g_h @ W + resid) masked by relu_ref > 0, with Wt = W^T [K, HC] row-major
    g_h = torch.randn(N, HC, generator=g).cuda().bfloat16()
    Wt = W.t().contiguous()
    resid = torch.randn(N, K, generator=g).cuda().bfloat16()
    ref_act = torch.randn(N, K, generator=g).cuda().bfloat16()
    g_x = torch.empty(N, K, dtype=torch.bfloat16, device="cuda")
    pkg._native.check(lib.gatres_t_proj_bwd_dx(g_h.data_ptr(), Wt.data_ptr(), resid.data_ptr(), ref_act.data_ptr(),
                                               g_x.data_ptr(), N, K, HC, BF16, _st(x)), "proj_bwd_dx bf16")
    gx_ref = (g_h.float() @ W.float() + resid.float()) * (ref_act.float() > 0)
    assert relerr(g_x.float(), gx_ref) < 2 ** -7
    # weight gradient partials (fp32 MFMA on widened operands), summed over the slabs
    S = 8
    stride = HC * K
    slab = torch.zeros(S * stride, device="cuda")
    pkg._native.check(lib.gatres_t_proj_bwd_dw(g_h.data_ptr(), x.data_ptr(), slab.data_ptr(), S, stride, N, K, HC, BF16,
                                               _st(x)), "proj_bwd_dw bf16")
    assert relerr(slab.view(S, HC, K).sum(0), g_h.float().t() @ x.float()) < 2e-4


def test_bf16_weight_copies(pkg, lib, oracle):
    nb, nc = 3, 32
    model, p = build(pkg, oracle, nb, nc, seed=5)
    per = 2 * nc * nc
    wb = torch.empty(nb * 4 * per, dtype=torch.bfloat16, device="cuda")
    pkg._native.check(lib.gatres_convert_conv_weights_bf16(model.flat_parameters.data_ptr(), wb.data_ptr(), nb, nc,
                                                           _st(wb)), "convert")
    wb = wb.view(nb, 4, per)
    for b in range(nb):
        W1, W2 = p[f"blocks.{b}.conv1.lin_src.weight"].cuda(), p[f"blocks.{b}.conv2.lin_src.weight"].cuda()
        assert torch.equal(wb[b, 0].view(2 * nc, nc), W1.bfloat16()) and torch.equal(wb[b, 1].view(nc, 2 * nc), W2.bfloat16())
        assert torch.equal(wb[b, 2].view(nc, 2 * nc), W1.t().bfloat16()) and torch.equal(wb[b, 3].view(2 * nc, nc), W2.t().bfloat16())


@pytest.mark.parametrize("nb,nc,bs", [(3, 128, 2), (15, 32, 2), (2, 64, 3), (25, 128, 4)])
def test_bf16_model_vs_fp32_oracle(pkg, oracle, nb, nc, bs):
    x, y, ei, mask = ctown_batch(pkg, bs)
    model, p = build(pkg, oracle, nb, nc, seed=3)
    model.set_compute_dtype("bf16")
    assert model.compute_dtype == "bf16"
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xin = x.clone(); xin[mask] = 0
    out_ref = oracle.gatres_forward(leaves, xin, ei, num_blocks=nb)
    loss_ref = torch.nn.functional.mse_loss(out_ref[mask], y[mask])
    loss_ref.backward()
    g_ref = torch.cat([v.grad.reshape(-1) for v in leaves.values()])
    out = model(xin.cuda(), ei.cuda())
    assert out.dtype == torch.float32 and torch.isfinite(out).all()
    e_out = relerr(out, out_ref)                                                  # max-norm: max |err| / max |ref|
    e_l2 = float((out.detach().cpu().double() - out_ref.double()).norm() / out_ref.double().norm())
    m = mask.cuda()
    loss = torch.nn.functional.mse_loss(out[m], y.cuda()[m])
    loss.backward()
    g = torch.cat([q.grad.reshape(-1) for q in model.parameters()])
    e_loss, e_g = relerr(loss, loss_ref), relerr(g, g_ref)
    e_g2 = float((g.cpu().double() - g_ref.double()).norm() / g_ref.double().norm())
    note(f"bf16 {nb}x{nc} bs{bs}: predictions rel-L2 / max-norm, loss, flat gradient rel-L2 / max-norm vs fp32 oracle",
         [e_l2, e_out, e_loss, e_g2, e_g])
    # ~1e-2 relative on the predictions.  Measured on gatres_large (25 x 128): 4e-3 in the L2 norm,
    # 1.4e-2 at the single worst node (bf16 rounding of the residual stream accumulates over 25 blocks)
    assert e_l2 < 1e-2 and e_out < 2.5e-2 and e_loss < 2e-2
    assert torch.isfinite(g).all() and e_g2 < 1.5e-1 and e_g < 1.5e-1      # (measured: 1.0e-1 / 3e-2 on gatres_large)
    # ... and against the oracle that rounds to bf16 exactly where the kernels store bf16 (oracle.gatres_forward_bf16):
    # what is left is fp32 summation order and the stored values it flips by one bf16 ulp.  Shallow models agree to 1e-4
    # (measured: 2 x 64: 1.0e-4 / 1.6e-4, 3 x 128: 8.1e-4 / 2.1e-4 for predictions / flat gradient); through 15 - 25
    # blocks every flip is a fresh 2^-8 perturbation of the residual stream, so the two bf16 trajectories drift apart
    # (25 x 128: 4.5e-3 / 4.2e-3) -- the gradient bound is still 15 - 30x tighter than against the fp32 oracle (1.0e-1),
    # which is what keeps a wrong-but-small term in the bf16 backward from hiding behind rounding
    lb = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    out_b = oracle.gatres_forward_bf16(lb, xin, ei, num_blocks=nb)
    loss_b = torch.nn.functional.mse_loss(out_b[mask], y[mask])
    loss_b.backward()
    g_b = torch.cat([v.grad.reshape(-1) for v in lb.values()])
    f_l2 = float((out.detach().cpu().double() - out_b.detach().double()).norm() / out_b.detach().double().norm())
    f_g2 = float((g.cpu().double() - g_b.double()).norm() / g_b.double().norm())
    f_loss = relerr(loss, loss_b)
    note(f"bf16 {nb}x{nc} bs{bs}: predictions rel-L2, loss, flat gradient rel-L2 vs the bf16-rounding oracle", [f_l2, f_loss, f_g2])
    shallow = nb <= 3
    assert f_l2 < (2e-3 if shallow else 8e-3) and f_loss < 2e-3 and f_g2 < (1e-3 if shallow else 1e-2), (f_l2, f_loss, f_g2)
    # inference path (no saved activations) == training path, and repeatable bit for bit
    with torch.no_grad():
        o2 = model(xin.cuda(), ei.cuda())
    assert torch.equal(o2, out.detach())


def test_bf16_loss_trajectory_follows_fp32(pkg, oracle):
    """four optimisation steps (device masks, Adam on the fp32 master parameters) in bf16 against the
    same steps in fp32 (tolerances: module docstring), parameters within a few learning rates."""
    nb, nc, bs = 6, 128, 4
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
    snaps = pkg.wdn_synth.make_snapshots(4 * bs, 388, seed=9).cuda()
    res = []
    for dtype in ("fp32", "bf16"):
        model, _ = build(pkg, oracle, nb, nc, seed=12)
        model.set_compute_dtype(dtype)
        tr = pkg.GATResTrainer(model, ei, 388 * bs, nodes_per_graph=[388] * bs, seed=3, use_graph=(dtype == "bf16"))
        assert not tr.fused
        losses = []
        for it in range(4):
            yb = snaps[it * bs:(it + 1) * bs].reshape(-1)
            losses.append(float(tr.step(yb, yb)))
        res.append((losses, model.flat_parameters.clone()))
    note("bf16 vs fp32 training, 4 steps: losses fp32 / bf16", [res[0][0], res[1][0]])
    assert abs(res[0][0][0] - res[1][0][0]) <= 5e-3 * abs(res[0][0][0])
    for a, b in zip(res[0][0], res[1][0]):
        assert abs(a - b) <= 1e-1 * abs(a), (res[0][0], res[1][0])
    assert float((res[0][1] - res[1][1]).abs().max()) <= 2 * 4 * 5e-4 * 1.01  # Adam moves a weight by <= ~lr per step, either way


def test_bf16_relabelled_plan_and_bucketed_backward(pkg, oracle):
    """bf16 through the other per-op drivers: a relabelled plan (gather / scatter of the fp32 caller-order vectors) and the
    bucketed backward of the data-parallel step must give exactly the plain bf16 step."""
    nb, nc, bs = 4, 64, 2
    t1 = pkg.wdn_synth.make_wdn_topology(388, 430, seed=0)
    sigma = torch.from_numpy(np.concatenate([np.random.RandomState(3).permutation(388) + 388 * k for k in range(bs)]))
    ei = sigma[pkg.wdn_synth.collate_edge_index(t1, 388, bs)].cuda()
    y = torch.randn(388 * bs, generator=torch.Generator().manual_seed(2)).cuda()
    res = []
    for split in (False, True):
        model, _ = build(pkg, oracle, nb, nc, seed=21)
        model.set_compute_dtype("bf16")
        tr = pkg.GATResTrainer(model, ei, 388 * bs, nodes_per_graph=[388] * bs, seed=5, use_graph=False,
                               force_collective_path=split, blocks_per_bucket=1)
        for _ in range(2):
            tr.step(y, y)
        res.append((tr.loss.clone(), model.flat_parameters.clone()))
    assert torch.isfinite(res[0][0]).all()
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_bf16_weight_gradients_two_dimensional_partials(pkg, oracle, monkeypatch):
    """Wide bf16 models on large batches: the weight-gradient kernel forms 64 row groups x 64 x 64 output blocks and the final
    sum reads 64 slab rows for the weight matrices, all rows for everything else (gatres_model_reduce_grads).  Against the
    one-matrix-per-workgroup form (GATRES_DW_1D=1) only the fp32 summation order differs."""
    nb, nc, bs = 2, 128, 12                         # 4656 rows -> 72 slabs > 64: the region-aware reduction is taken
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(388, 430, seed=0), 388, bs).cuda()
    y = torch.randn(388 * bs, generator=torch.Generator().manual_seed(4)).cuda()
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, np.random.RandomState(1))).cuda()
    grads = []
    for one_d in (False, True):
        if one_d:
            monkeypatch.setenv("GATRES_DW_1D", "1")
        model, _ = build(pkg, oracle, nb, nc, seed=9)
        model.set_compute_dtype("bf16")
        tr = pkg.GATResTrainer(model, ei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=False)
        assert pkg._native.load().gatres_num_slabs(model._cmodel_ref(), 388 * bs) > 64
        tr.forward_backward(y, y, mask)
        grads.append(tr.grads.clone())
    assert torch.isfinite(grads[0]).all() and float(grads[0].abs().max()) > 0
    assert relerr(grads[0], grads[1]) < 1e-5


def test_bf16_gatres_large_full_batch_128(pkg, oracle):
    """BASELINE config 3 at its real size in bf16 (25 x 128, C-Town, bs = 128: the 256-slab two-dimensional weight-gradient
    partials run here, not at a toy batch): block-diagonal independence and bitwise repeatability of the predictions,
    one graph of the batch against the bf16-rounding oracle, and one native training step -- finite loss, exactly
    128 x 368 masked nodes, every parameter moved by at most ~lr, gradient of the step against the same oracle."""
    nb, nc, bs = 25, 128, 128
    model, p = build(pkg, oracle, nb, nc, seed=3)
    model.set_compute_dtype("bf16")
    x, y, ei, mask = ctown_batch(pkg, bs)
    dx, dei = x.cuda(), ei.cuda()
    with torch.no_grad():
        out = model(dx, dei)
        assert out.shape == (388 * bs, 1) and torch.isfinite(out).all()
        one = pkg.wdn_synth.make_wdn_topology()
        first = model(dx[:388], one.cuda())
        assert torch.equal(out[:388], first) and torch.equal(model(dx, dei), out)
        ref = oracle.gatres_forward_bf16(p, x[:388], one)
    e = float((first.cpu().double() - ref.double()).norm() / ref.double().norm())
    note("bf16 gatres_large 25x128 bs128: one graph of the batch vs the bf16-rounding oracle (rel-L2)", e)
    assert e < 8e-3          # (measured 3.9e-3: 25 blocks, see test_bf16_model_vs_fp32_oracle)
    # the step's gradient on a 4-graph slice of the same batch vs the oracle (the oracle needs ~1 min per 4 graphs)
    sub = 4
    xs, ys, ms = x[:388 * sub], y[:388 * sub], mask[:388 * sub]
    eis = pkg.wdn_synth.collate_edge_index(one, 388, sub)
    lb = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xin = xs.clone(); xin[ms] = 0
    ob = oracle.gatres_forward_bf16(lb, xin, eis, num_blocks=nb)
    torch.nn.functional.mse_loss(ob[ms], ys[ms]).backward()
    g_b = torch.cat([v.grad.reshape(-1) for v in lb.values()])
    model.zero_grad()
    o = model(xin.cuda(), eis.cuda())
    torch.nn.functional.mse_loss(o[ms.cuda()], ys.cuda()[ms.cuda()]).backward()
    g = torch.cat([q.grad.reshape(-1) for q in model.parameters()])
    f_g2 = float((g.cpu().double() - g_b.double()).norm() / g_b.double().norm())
    note("bf16 gatres_large 25x128: flat gradient (4 graphs) vs the bf16-rounding oracle (rel-L2)", f_g2)
    assert f_g2 < 1e-2
    before = model.flat_parameters.clone()
    tr = pkg.GATResTrainer(model, dei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=False)
    assert not tr.fused
    loss = tr.step(dx.reshape(-1), dx.reshape(-1))
    assert torch.isfinite(loss).all() and int(tr.mask.sum()) == bs * 368
    d = (model.flat_parameters - before).abs()
    assert torch.isfinite(tr.grads).all() and 0 < float(d.max()) <= 5e-4 * 1.01



# ---------------------------------------------------------------------------------------------- blocked kernels (round 5)
def _plan_for(pkg, bs, nodes=388, pipes=430, seed=0):
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(nodes, pipes, seed=seed), nodes, bs).cuda()
    return pkg.GraphPlan(ei, nodes * bs, device=ei.device, segments=False), ei


@pytest.mark.parametrize("bs,nodes,pipes", [(5, 388, 430), (1, 131, 150)])
def test_blocked_kernels_equal_the_per_op_pairs(pkg, lib, bs, nodes, pipes):
    """k_blocked.hip: one launch = a sparse stage + the projection that consumes it.  Against the two per-op launches it
    replaces, on the same operands, EVERY output bit for bit: the sparse stage's (o1 / x_next / g_h, alpha, g_a_src: the same
    slot arithmetic), the projected rows (the same bf16 operands, the same k order on the matrix cores) and the next
    convolution's attention logits (summed from the fp32 accumulators in the projection kernel's association).  Ragged row
    counts: the last 16-row tile and the last 32-row phase of a workgroup's share are partial."""
    nc = 128
    plan, ei = _plan_for(pkg, bs, nodes, pipes)
    assert plan.flags & 2, "the synthetic WDN topology has low degrees: GATRES_GRAPH_DEG_LE32"
    N, Eg = plan.num_nodes, plan.num_edges_gat
    gp = plan.ref()
    g = torch.Generator().manual_seed(bs * 1000 + nodes)
    rn = lambda *s: torch.randn(*s, generator=g).cuda()
    bf = lambda *s: rn(*s).bfloat16()
    st = lambda: pkg._native.current_stream(ei.device)
    chk = pkg._native.check
    # ---- conv1 aggregation (H = 2, ReLU) -> conv2 projection (K = 256 -> M = 128, H = 1)
    h1, a_s, a_d, bias = bf(N, 256), rn(N, 2), rn(N, 2), rn(256)
    W2 = (rn(128, 256) / 16).bfloat16()
    att_s, att_d = rn(128), rn(128)
    o_a, al_a = torch.empty(N, 256, dtype=torch.bfloat16, device="cuda"), torch.zeros(Eg, 2, device="cuda")
    h2_a, as_a, ad_a = torch.empty(N, 128, dtype=torch.bfloat16, device="cuda"), torch.empty(N, device="cuda"), torch.empty(N, device="cuda")
    chk(lib.gatres_t_gat_aggregate_fwd(gp, h1.data_ptr(), a_s.data_ptr(), a_d.data_ptr(), bias.data_ptr(), o_a.data_ptr(),
                                       al_a.data_ptr(), 2, nc, 1, BF16, st()), "agg")
    chk(lib.gatres_t_proj_attn_fwd(o_a.data_ptr(), W2.data_ptr(), att_s.data_ptr(), att_d.data_ptr(), h2_a.data_ptr(),
                                   as_a.data_ptr(), ad_a.data_ptr(), N, 256, 1, nc, BF16, st()), "proj")
    o_b, al_b = torch.zeros_like(o_a), torch.zeros_like(al_a)
    h2_b, as_b, ad_b = torch.zeros_like(h2_a), torch.zeros_like(as_a), torch.zeros_like(ad_a)
    chk(lib.gatres_bf16_agg_proj_fwd(gp, h1.data_ptr(), a_s.data_ptr(), a_d.data_ptr(), bias.data_ptr(), o_b.data_ptr(),
                                     al_b.data_ptr(), W2.data_ptr(), att_s.data_ptr(), att_d.data_ptr(), h2_b.data_ptr(),
                                     as_b.data_ptr(), ad_b.data_ptr(), nc, st()), "agg_proj")
    torch.cuda.synchronize()
    assert torch.equal(o_a, o_b) and torch.equal(al_a, al_b)
    assert torch.equal(h2_a, h2_b)
    assert torch.equal(as_b, as_a) and torch.equal(ad_b, ad_a)
    # ---- K3 (mean + residual + ReLU) -> the next block's conv1 projection (K = 128 -> M = 256, H = 2)
    y2, x0 = bf(N, 128), bf(N, 128)
    W1 = (rn(256, 128) / 11).bfloat16()
    att_s1, att_d1 = rn(256), rn(256)
    xn_a = torch.empty(N, 128, dtype=torch.bfloat16, device="cuda")
    h1_a, as1_a, ad1_a = torch.empty(N, 256, dtype=torch.bfloat16, device="cuda"), torch.empty(N, 2, device="cuda"), torch.empty(N, 2, device="cuda")
    chk(lib.gatres_t_mean_residual_relu_fwd(gp, y2.data_ptr(), x0.data_ptr(), xn_a.data_ptr(), nc, BF16, st()), "mean")
    chk(lib.gatres_t_proj_attn_fwd(xn_a.data_ptr(), W1.data_ptr(), att_s1.data_ptr(), att_d1.data_ptr(), h1_a.data_ptr(),
                                   as1_a.data_ptr(), ad1_a.data_ptr(), N, 128, 2, nc, BF16, st()), "proj1")
    xn_b, h1_b, as1_b, ad1_b = torch.zeros_like(xn_a), torch.zeros_like(h1_a), torch.zeros_like(as1_a), torch.zeros_like(ad1_a)
    chk(lib.gatres_bf16_mean_proj_fwd(gp, y2.data_ptr(), x0.data_ptr(), xn_b.data_ptr(), W1.data_ptr(), att_s1.data_ptr(),
                                      att_d1.data_ptr(), h1_b.data_ptr(), as1_b.data_ptr(), ad1_b.data_ptr(), nc, st()),
        "mean_proj")
    torch.cuda.synchronize()
    assert torch.equal(xn_a, xn_b) and torch.equal(h1_a, h1_b)
    assert torch.equal(as1_b, as1_a) and torch.equal(ad1_b, ad1_a)
    # ---- source-major backward -> input gradient, both convolutions
    for H, K, M in ((1, 128, 256), (2, 256, 128)):
        g_out, alpha = bf(N, K), torch.rand(Eg, H, generator=g).cuda()
        g_e, g_ad = rn(Eg, H), rn(N, H)
        att_s2, att_d2 = rn(K), rn(K)
        Wt = (rn(M, K) / 13).bfloat16()
        resid = bf(N, M) if H == 2 else None
        ref_act = bf(N, M)
        gh_a, gas_a, gx_a = torch.empty(N, K, dtype=torch.bfloat16, device="cuda"), torch.empty(N, H, device="cuda"), torch.empty(N, M, dtype=torch.bfloat16, device="cuda")
        chk(lib.gatres_t_gat_aggregate_bwd_src(gp, g_out.data_ptr(), alpha.data_ptr(), g_e.data_ptr(), g_ad.data_ptr(),
                                               att_s2.data_ptr(), att_d2.data_ptr(), gh_a.data_ptr(), gas_a.data_ptr(), H, nc,
                                               BF16, st()), "src")
        chk(lib.gatres_t_proj_bwd_dx(gh_a.data_ptr(), Wt.data_ptr(), pkg._native.ptr(resid), ref_act.data_ptr(), gx_a.data_ptr(),
                                     N, M, K, BF16, st()), "dx")
        gh_b, gas_b, gx_b = torch.zeros_like(gh_a), torch.zeros_like(gas_a), torch.zeros_like(gx_a)
        chk(lib.gatres_bf16_src_dx_bwd(gp, g_out.data_ptr(), alpha.data_ptr(), g_e.data_ptr(), g_ad.data_ptr(), att_s2.data_ptr(),
                                       att_d2.data_ptr(), gh_b.data_ptr(), gas_b.data_ptr(), Wt.data_ptr(), pkg._native.ptr(resid),
                                       ref_act.data_ptr(), gx_b.data_ptr(), H, nc, st()), "src_dx")
        torch.cuda.synchronize()
        assert torch.equal(gh_a, gh_b) and torch.equal(gas_a, gas_b), H
        assert torch.equal(gx_a, gx_b), H


def test_blocked_model_equals_the_per_op_model(pkg, oracle, monkeypatch):
    """gatres_large-shaped training steps (bf16, nc = 128) with the blocked launches (GATRES_BLOCKED=1) and without (the
    default): the same predictions, losses, gradients and parameters, bit for bit."""
    nb, nc, bs = 4, 128, 6
    x, y, ei, mask = ctown_batch(pkg, bs)
    res = []
    for off in (True, False):
        if off:
            monkeypatch.delenv("GATRES_BLOCKED", raising=False)
        else:
            monkeypatch.setenv("GATRES_BLOCKED", "1")
        model, p = build(pkg, oracle, nb, nc, seed=11)
        model.set_compute_dtype("bf16")
        tr = pkg.GATResTrainer(model, ei.cuda(), x.shape[0], nodes_per_graph=[388] * bs, use_graph=False, fused=False)
        assert bool(pkg._native.load().gatres_blocked_supported(model._cmodel_ref(), tr.plan.ref())) == (not off)
        losses = [float(tr.step(x.cuda(), y.cuda(), mask.cuda())) for _ in range(2)]
        res.append((tr.out.clone(), losses, tr.grads.clone(), model.flat_parameters.clone()))
    assert res[0][1] == res[1][1], (res[0][1], res[1][1])
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])

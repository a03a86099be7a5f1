"""Per-kernel parity: every HIP kernel of include/gatres.h against the CPU oracle's restatement of the PyG op it
replaces, on seeded inputs.  Tolerances are fp32 round-off: rel 1e-5 of the tensor's max magnitude."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def graph(n, e, seed, self_loops=0, dup=0, hub=False, star=False):
    g = torch.Generator().manual_seed(seed)
    src = torch.randint(0, n, (e,), generator=g)
    dst = torch.randint(0, n, (e,), generator=g)
    keep = src != dst
    ei = torch.stack([src[keep], dst[keep]])
    if hub:      # one destination with far more than 8 in-edges, and one isolated node (n-1)
        s = torch.arange(1, min(n - 1, 40))
        ei = torch.cat([ei[:, (ei[0] != n - 1) & (ei[1] != n - 1)], torch.stack([s, torch.zeros_like(s)])], 1)
    if star:     # in- AND out-degree hubs handled by whole waves (k_aggregate.hip hub paths): node 0 <-> everyone, two
        #          hubs in one wave (nodes 0, 1) and one in the ragged last wave (node n-1)
        o = torch.arange(1, n)
        a, b = torch.arange(2, 62), torch.arange(40, 70)
        ei = torch.cat([ei, torch.stack([o, torch.zeros_like(o)]), torch.stack([torch.zeros_like(o), o]),
                        torch.stack([a, torch.ones_like(a)]), torch.stack([torch.ones_like(a), a]),
                        torch.stack([b, torch.full_like(b, n - 1)]), torch.stack([torch.full_like(b, n - 1), b])], 1)
    if self_loops:
        l = torch.randint(0, n - 1, (self_loops,), generator=g)
        ei = torch.cat([ei, torch.stack([l, l])], 1)
    if dup:
        ei = torch.cat([ei, ei[:, :dup]], 1)
    return ei


@pytest.fixture(scope="module")
def ops():
    from tests import hipops
    return hipops


CASES = [(8, 1), (8, 2), (32, 2), (32, 1), (128, 2), (128, 1), (4, 2), (16, 2), (64, 1)]


@pytest.mark.parametrize("C,H", CASES)
@pytest.mark.parametrize("n,e,kw", [(97, 260, dict(self_loops=3, dup=5, hub=True)), (388 * 2, 860 * 2, {}),
                                    (203, 500, dict(self_loops=2, dup=4, star=True))])
def test_gatconv_forward_and_backward(pkg, oracle, ops, C, H, n, e, kw):
    torch.manual_seed(C * 10 + H)
    K = C if H == 2 else 2 * C
    ei = graph(n, e, 7, **kw)
    x = torch.randn(n, K)
    W = (torch.rand(H * C, K) - 0.5)
    a_s, a_d = torch.rand(1, H, C) - 0.5, torch.rand(1, H, C) - 0.5
    b = torch.rand(H * C) - 0.5
    g_up = torch.randn(n, H * C)
    # oracle
    leaves = [t.clone().requires_grad_(True) for t in (x, W, a_s, a_d, b)]
    out_ref, alpha_ref, _ = oracle.gat_conv(leaves[0], ei, leaves[1], leaves[2], leaves[3], leaves[4], H, True,
                                            return_alpha=True)
    out_ref = out_ref.relu()
    out_ref.backward(g_up)
    # HIP
    dev = "cuda"
    plan = pkg.GraphPlan(ei, n, device=dev, reorder=False)     # single kernels work in PLAN order: keep it the caller's
    xd, Wd, asd, add, bd = [t.to(dev).contiguous() for t in (x, W, a_s.reshape(-1), a_d.reshape(-1), b)]
    h, hs, hd = ops.proj_attn_fwd(xd, Wd, asd, add, H)
    h_ref = (x @ W.t())
    assert relerr(h, h_ref) < 1e-5
    assert relerr(hs, (h_ref.view(n, H, C) * a_s).sum(-1)) < 1e-5
    out, alpha = ops.gat_aggregate_fwd(plan, h, hs, hd, bd, H, relu=True)
    assert relerr(out, out_ref) < 1e-5
    # alpha is stored in destination-sorted order; compare as a multiset per destination via the row sums + values
    order = torch.sort(torch.cat([ei[1][ei[0] != ei[1]], torch.arange(n)]), stable=True).indices
    assert relerr(alpha, alpha_ref.detach()[order]) < 1e-5
    # backward: g wrt pre-ReLU output
    g_pre = (g_up * (out_ref.detach() > 0)).to(dev).contiguous()
    g_h, g_as, g_ad, _ = ops.gat_aggregate_bwd(plan, g_pre, h, alpha, hs, hd, asd, add, H)
    g_x = ops.proj_bwd_dx(g_h, Wd.t().contiguous())
    assert relerr(g_x, leaves[0].grad) < 2e-5
    S = 7
    g_W = ops.proj_bwd_dw(g_h, xd, S)
    assert relerr(g_W, leaves[1].grad) < 2e-5
    g_att_s, g_att_d, g_b = ops.conv_param_grads(h, g_as, g_ad, g_pre, H, S)
    assert relerr(g_att_s, leaves[2].grad.reshape(-1)) < 2e-5
    assert relerr(g_att_d, leaves[3].grad.reshape(-1)) < 2e-5
    assert relerr(g_b, leaves[4].grad) < 2e-5


@pytest.mark.parametrize("C,H", [(32, 2), (128, 2), (128, 1)])
def test_sparse_kernels_lane_feature_instances_agree(pkg, ops, C, H, monkeypatch):
    """Rows of 64 features and more take eight features per lane (half the lanes per row), GATRES_AGG_LANE_FEATURES=4 the
    four-per-lane instances: sums over a row's edges run per feature in edge order either way, so the forward results are
    bit-identical; the backward head dots associate differently (fp32 rounding only)."""
    torch.manual_seed(3)
    n, HC = 500, H * C
    h, g_pre = torch.randn(n, HC).cuda(), torch.randn(n, HC).cuda()
    hs, hd = torch.randn(n, H).cuda(), torch.randn(n, H).cuda()
    b, a_s, a_d = torch.randn(HC).cuda(), torch.randn(HC).cuda(), torch.randn(HC).cuda()
    for hub in (False, True):                  # (hub rows: the wave's edge slots depend on the lanes per row -> rounding only)
        ei = graph(n, 1300, 11, self_loops=2, dup=3, hub=hub)
        plan = pkg.GraphPlan(ei, n, device="cuda", reorder=False)
        res = {}
        for w in ("8", "4"):
            monkeypatch.setenv("GATRES_AGG_LANE_FEATURES", w)
            out, alpha = ops.gat_aggregate_fwd(plan, h, hs, hd, b, H, relu=True)
            g_h, g_as, g_ad, g_e = ops.gat_aggregate_bwd(plan, g_pre, h, alpha, hs, hd, a_s, a_d, H)
            m = ops.mean_residual_relu_fwd(plan, h, g_pre)
            mb = ops.mean_bwd(plan, g_pre)
            torch.cuda.synchronize()
            res[w] = (out, alpha, m, mb, g_h, g_as, g_ad)
        for a, c in list(zip(res["8"], res["4"]))[:4]:
            assert relerr(a, c) < 1e-5 if hub else torch.equal(a, c)
        for a, c in list(zip(res["8"], res["4"]))[4:]:
            assert relerr(a, c) < 1e-5

@pytest.mark.parametrize("star", [False, True])
@pytest.mark.parametrize("C", [4, 32, 128])
def test_mean_residual_relu_and_backward(pkg, oracle, ops, C, star):
    n = 131
    ei = graph(n, 400, 3, self_loops=4, dup=6, hub=True, star=star)
    y = torch.randn(n, C, requires_grad=True)
    x0 = torch.randn(n, C, requires_grad=True)
    ref = (oracle.simple_conv_mean(y, ei) + x0).relu()
    g_up = torch.randn(n, C)
    ref.backward(g_up)
    plan = pkg.GraphPlan(ei, n, device="cuda", reorder=False)
    out = ops.mean_residual_relu_fwd(plan, y.detach().cuda(), x0.detach().cuda())
    assert relerr(out, ref) < 1e-6
    if not star:
        assert float(out[n - 1].cpu().sub(x0.detach()[n - 1].relu()).abs().max()) == 0.0   # isolated node: mean = 0
    g_pre = (g_up * (ref.detach() > 0)).cuda()
    assert relerr(ops.mean_bwd(plan, g_pre), y.grad) < 1e-6
    assert relerr(g_pre, x0.grad) == 0.0


def test_proj_dx_residual_and_relu_mask_epilogue(ops):
    n, K, HC = 50, 32, 64
    g_h, Wt = torch.randn(n, HC).cuda(), torch.randn(K, HC).cuda()
    resid, ref = torch.randn(n, K).cuda(), torch.randn(n, K).cuda()
    out = ops.proj_bwd_dx(g_h, Wt, resid, ref)
    exp = (g_h.double() @ Wt.double().t() + resid.double()) * (ref > 0)
    assert relerr(out, exp) < 1e-5


def test_lin0_lin1(ops):
    n, nc = 777, 32
    x, w0, b0 = torch.randn(n), torch.randn(nc), torch.randn(nc)
    mask = (torch.rand(n) < 0.5).to(torch.uint8)
    out = ops.lin0_fwd(x.cuda(), w0.cuda(), b0.cuda(), mask.cuda())
    xm = torch.where(mask.bool(), torch.zeros_like(x), x)
    assert torch.equal(out.cpu(), xm[:, None] * w0[None] + b0[None])
    z, w1, b1 = torch.randn(n, nc), torch.randn(nc), torch.randn(1)
    assert relerr(ops.lin1_fwd(z.cuda(), w1.cuda(), b1.cuda()), z @ w1 + b1) < 1e-6


@pytest.mark.parametrize("nc,dtype", [(32, torch.float32), (128, torch.bfloat16), (24, torch.float32)])
def test_lin0_lin1_backward_slabs(ops, nc, dtype, monkeypatch):
    """lin0 / lin1 backward (train.py:181 through GraphModels.py:455 and :471): per-slab partial sums of the weight and
    bias gradients and lin1's input gradient with conv-block ReLU mask.  nc = 32 / 128 take the row-wise 256-thread
    kernels, nc = 24 the one-wave-per-slab form; GATRES_LIN_BWD_WAVE=1 must give the same sums."""
    lib = ops.N_.load()
    n, slabs = 3001, 7
    DT = {torch.float32: 0, torch.bfloat16: 1}[dtype]
    stride = 2 * nc + 8
    torch.manual_seed(5)
    g = torch.randn(n, nc).to(dtype).cuda()
    x = torch.randn(n).cuda()
    mask = (torch.rand(n) < 0.5).to(torch.uint8).cuda()
    xm = torch.where(mask.bool(), torch.zeros_like(x), x).double()
    go, w1 = torch.randn(n).cuda(), torch.randn(nc).cuda()
    got = {}
    for form in ("rows", "wave"):
        if form == "wave":
            monkeypatch.setenv("GATRES_LIN_BWD_WAVE", "1")
        sw = torch.zeros(slabs, stride, device="cuda")
        ops.N_.check(lib.gatres_t_lin0_bwd(g.data_ptr(), x.data_ptr(), mask.data_ptr(), sw.data_ptr(), sw.data_ptr() + 4 * nc,
                                           slabs, stride, n, nc, DT, ops._s(g)), "lin0_bwd")
        torch.cuda.synchronize()
        gw, gb = sw[:, :nc].double().sum(0), sw[:, nc:2 * nc].double().sum(0)
        assert relerr(gw, (g.double() * xm[:, None]).sum(0)) < 1e-5
        assert relerr(gb, g.double().sum(0)) < 1e-5
        # lin1: g_x = g_out (x) w under the ReLU mask of x, g_w = sum g_out * x, g_b = sum g_out
        gx = torch.empty(n, nc, dtype=dtype, device="cuda")
        sw1 = torch.zeros(slabs, stride, device="cuda")
        ops.N_.check(lib.gatres_t_lin1_bwd(go.data_ptr(), g.data_ptr(), w1.data_ptr(), gx.data_ptr(), sw1.data_ptr(),
                                           sw1.data_ptr() + 4 * nc, slabs, stride, n, nc, 1, DT, ops._s(g)), "lin1_bwd")
        torch.cuda.synchronize()
        assert relerr(sw1[:, :nc].double().sum(0), (go.double()[:, None] * g.double()).sum(0)) < 1e-5
        assert abs(float(sw1[:, nc].double().sum(0)) - float(go.double().sum())) < 1e-6 * float(go.double().abs().sum())
        exp = (go[:, None] * w1[None]) * (g.float() > 0)
        assert torch.equal(gx.float(), exp.to(dtype).float())
        got[form] = (gw, gb, sw1[:, :nc].double().sum(0))
    for a, b in zip(got["rows"], got["wave"]):
        assert relerr(a, b) < 1e-6


@pytest.mark.parametrize("H,C", [(2, 128), (1, 128), (2, 32)])
def test_conv_param_grads_bf16(ops, H, C):
    """att_src / att_dst / bias gradient partials from bf16 tables (GATConv's parameters, GraphModels.py:464-466 backward):
    the per-slab partial sums add up to an fp64 statement of the sums."""
    lib = ops.N_.load()
    torch.manual_seed(9)
    n, HC, slabs = 2999, H * C, 5
    h = torch.randn(n, HC).to(torch.bfloat16).cuda()
    go = torch.randn(n, HC).to(torch.bfloat16).cuda()
    gs, gd = torch.randn(n, H).cuda(), torch.randn(n, H).cuda()
    stride = 3 * HC
    hd = h.double().view(n, H, C)
    want = ((gs.double()[:, :, None] * hd).sum(0).reshape(-1), (gd.double()[:, :, None] * hd).sum(0).reshape(-1),
            go.double().sum(0))
    for _ in range(1):
        sl = torch.zeros(slabs, stride, device="cuda")
        ops.N_.check(lib.gatres_t_conv_param_grads(h.data_ptr(), gs.data_ptr(), gd.data_ptr(), go.data_ptr(), sl.data_ptr(),
                                                   sl.data_ptr() + 4 * HC, sl.data_ptr() + 8 * HC, slabs, stride, n, H, C, 1,
                                                   ops._s(h)), "conv_param_grads")
        torch.cuda.synchronize()
        for k in range(3):
            assert relerr(sl[:, k * HC:(k + 1) * HC].double().sum(0), want[k]) < 1e-5


def test_device_mask_sampler_exact_count_and_fresh_per_step(ops):
    sizes = [388] * 5 + [17, 1000, 3]
    off = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    node_ptr = torch.from_numpy(off).cuda()
    n = int(off[-1])
    step = torch.zeros(2, dtype=torch.int64, device="cuda")
    m0 = ops.mask_generate(node_ptr, len(sizes), 0.95, 1234, step, n).cpu().numpy()
    for i, s in enumerate(sizes):
        assert int(m0[off[i]:off[i + 1]].sum()) == int(s * 0.95)       # auxil.py:154,161
    assert np.array_equal(m0, ops.mask_generate(node_ptr, len(sizes), 0.95, 1234, step, n).cpu().numpy())
    step[0] = 1
    m1 = ops.mask_generate(node_ptr, len(sizes), 0.95, 1234, step, n).cpu().numpy()
    assert not np.array_equal(m0, m1)
    # uniformity: every node of a 388-node graph is left unmasked ~5% of the time
    cnt = np.zeros(388)
    for t in range(400):
        step[0] = t
        cnt += 1 - ops.mask_generate(node_ptr, len(sizes), 0.95, 99, step, n).cpu().numpy()[:388]
    assert abs(cnt.mean() / 400 - 20 / 388) < 1e-9 and cnt.max() < 60 and cnt.min() > 1


def test_masked_mse_and_adam_match_torch(ops):
    n = 5000
    out, y = torch.randn(n), torch.randn(n)
    mask = torch.rand(n) < 0.95
    o = out.clone().requires_grad_(True)
    ref = torch.nn.functional.mse_loss(o[mask], y[mask])
    ref.backward()
    loss, g = ops.masked_mse(out.cuda(), y.cuda(), mask.to(torch.uint8).cuda())
    assert relerr(loss, ref) < 1e-6 and relerr(g, o.grad) < 1e-6
    _adam_against_torch(ops, 10007)


@pytest.mark.parametrize("P", [512 * 256 + 77, 1700003])
def test_adam_grid_stride_counts_the_step_once(ops, P):
    """More parameters than ADAM_MAX_BLOCKS x 256 (gatres_large has 1.7 M): the launch is grid-strided and the step counter
    still advances exactly once per launch (train.py:187 optimizer.step())."""
    _adam_against_torch(ops, P)


def _adam_against_torch(ops, P):
    p, gr = torch.randn(P), torch.randn(P)
    pt = p.clone().requires_grad_(True)
    opt = torch.optim.Adam([pt], lr=5e-4, weight_decay=6e-6)
    pd, m, v = p.cuda(), torch.zeros(P).cuda(), torch.zeros(P).cuda()
    step = torch.zeros(2, dtype=torch.int64, device="cuda")
    for it in range(3):
        pt.grad = gr * (it + 1)
        opt.step()
        ops.adam_step(pd, (gr * (it + 1)).cuda(), m, v, step, 5e-4, 0.9, 0.999, 1e-8, 6e-6)
        assert relerr(pd, pt) < 1e-6
    assert int(step[0]) == 3 and int(step[1]) == 0


@pytest.mark.parametrize("K,H,C", [(128, 2, 128), (256, 1, 128), (64, 2, 64), (128, 1, 64)])
def test_wide_projection_on_many_rows_matches_the_tile_kernel(ops, K, H, C, monkeypatch):
    """Wide models on tens of thousands of rows take the persistent LDS-staged projection kernel (k_proj.hip:
    proj_lds_kernel); its outputs, attention logits and the dX epilogue must be BIT-identical to the tile-per-wave
    kernel it replaces there, and agree with a float64 torch product."""
    n, M = 20000 + 37, H * C
    g = torch.Generator().manual_seed(K + M)
    x = torch.randn(n, K, generator=g).cuda()
    W = (torch.randn(M, K, generator=g) / K ** 0.5).cuda()
    a_s, a_d = torch.randn(M, generator=g).cuda(), torch.randn(M, generator=g).cuda()
    resid, ref = torch.randn(n, M, generator=g).cuda(), torch.randn(n, M, generator=g).cuda()
    h1, s1, d1 = ops.proj_attn_fwd(x, W, a_s, a_d, H)
    o1 = ops.proj_bwd_dx(x, W, resid, ref)
    monkeypatch.setenv("GATRES_NO_PROJ_LDS", "1")
    h0, s0, d0 = ops.proj_attn_fwd(x, W, a_s, a_d, H)
    o0 = ops.proj_bwd_dx(x, W, resid, ref)
    assert torch.equal(h1, h0) and torch.equal(s1, s0) and torch.equal(d1, d0) and torch.equal(o1, o0)
    h64 = (x.double() @ W.double().t())
    assert float((h1.double() - h64).abs().max() / h64.abs().max()) < 1e-6
    s64 = (h64.view(n, H, C) * a_s.double().view(1, H, C)).sum(-1)
    assert float((s1.double() - s64).abs().max() / s64.abs().max()) < 1e-5


def test_side_stream_fork_join_gives_the_same_gradients(pkg, oracle, monkeypatch):
    """GATRES_SIDE_STREAM (product switch): the per-op backward's parameter-gradient launches on the library's side stream
    (fork / join through events; the default for fp32 at nc >= 128) or on the caller's stream -- the same launches, the same
    bits, also across repeated calls (the join guard must leave no launch of one call running into the next)."""
    from test_gpu_model import build
    x, y, ei, mask = pkg.wdn_synth.make_batch(3, 120, 140)
    res = []
    for v in ("0", "1"):
        monkeypatch.setenv("GATRES_SIDE_STREAM", v)
        model, _ = build(pkg, oracle, 3, 32, seed=5, fused=False)
        gs = []
        for rep in range(3):
            model.zero_grad()
            out = model(x.cuda(), ei.cuda())
            out.backward(torch.ones_like(out) * (rep + 1))
            gs.append(torch.cat([q.grad.reshape(-1) for q in model.parameters()]).clone())
        torch.cuda.synchronize()
        res.append(gs)
    for a, b in zip(*res):
        assert torch.isfinite(a).all() and torch.equal(a, b)

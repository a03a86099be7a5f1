"""Whole-path parity on the GPU: GATResMeanConv forward/backward and the training step against the CPU oracle,
through the nn.Module surface the reference's train.py uses and through the native step driver.

Tolerances (fp32): predictions within 1e-5 relative (north-star), gradients / updated weights within 1e-4
relative (max-abs error over max-abs value of each tensor)."""
import copy
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

REPORT = {}
_G64_CACHE = {}


def relerr(a, b):
    a, b = a.detach().double().cpu().reshape(-1), b.detach().double().cpu().reshape(-1)
    return float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))


def note(name, val):
    """Measured errors of the -m gpu tests, every module's (they all import this function), MERGED into
    gpurun_out/parity_report.json: a run of one module or one test adds to what is there instead of replacing it.  The
    copy judged with a round is committed as profiles/rNN_parity_report.json."""
    REPORT[name] = val
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "parity_report.json")
        merged = {}
        if os.path.exists(path):
            try:
                with open(path) as f:
                    merged = json.load(f)
            except ValueError:
                merged = {}
        merged.update(REPORT)
        with open(path, "w") as f:
            json.dump(merged, f, indent=1, sort_keys=True)
    except OSError:
        pass


def load_params(model, p):
    sd = {}
    for k, v in p.items():
        sd[k] = v
        if k.endswith("lin_src.weight"):
            sd[k.replace("lin_src", "lin_dst")] = v
    model.load_state_dict(sd)


def build(pkg, oracle, nb, nc, seed, fused=True):
    p = oracle.init_params(nb, nc, seed=seed)
    m = pkg.GATResMeanConv(num_blocks=nb, nc=nc, fused=fused)
    load_params(m, p)
    return m.cuda(), p


def ctown_batch(pkg, bs, nodes=388, pipes=430):
    return pkg.wdn_synth.make_batch(bs, nodes, pipes)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "per_op"])
@pytest.mark.parametrize("nb,nc,bs,nodes,pipes", [(2, 8, 3, 40, 47), (15, 32, 2, 388, 430), (3, 128, 2, 60, 70),
                                                   (1, 4, 1, 9, 10), (0, 16, 2, 12, 14), (2, 64, 2, 130, 150),
                                                   (2, 32, 1, 450, 500), (25, 128, 4, 388, 430)])
def test_forward_backward_parity(pkg, oracle, nb, nc, bs, nodes, pipes, fused):
    x, y, ei, mask = ctown_batch(pkg, bs, nodes, pipes)
    model, p = build(pkg, oracle, nb, nc, seed=3, fused=fused)
    # oracle fp32 + fp64
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xin = x.clone(); xin[mask] = 0
    out_ref = oracle.gatres_forward(leaves, xin, ei, num_blocks=nb)
    loss_ref = torch.nn.functional.mse_loss(out_ref[mask], y[mask])
    loss_ref.backward()
    p64 = {k: v.double() for k, v in p.items()}
    out64 = oracle.gatres_forward(p64, xin.double(), ei, num_blocks=nb)
    # HIP through the reference call signature: model(x, edge_index, batch, edge_attr)
    xd = xin.cuda()
    out = model(xd, ei.cuda(), None, None)
    assert out.shape == (x.shape[0], 1) and out.requires_grad
    e_out, e_out64, e_ref64 = relerr(out, out_ref), relerr(out, out64), relerr(out_ref, out64)
    note(f"fwd nb{nb} nc{nc} {'fused' if fused else 'per-op'}: hip-vs-oracle32 / hip-vs-oracle64 / oracle32-vs-oracle64", [e_out, e_out64, e_ref64])
    assert e_out < 1e-5, (e_out, e_out64, e_ref64)
    m = mask.cuda()
    loss = torch.nn.functional.mse_loss(out[m], y.cuda()[m])
    loss.backward()
    assert relerr(loss, loss_ref) < 1e-5
    # gradients: the fp64 oracle arbitrates.  Some tensors (att_dst: sum_e g_e is ~0 by softmax symmetry) are pure
    # cancellation noise in fp32, so each tensor's error is judged against the fp32 oracle's own error vs fp64 and
    # against the scale of the whole gradient.  The network is piecewise linear: when an activation sits within fp32
    # round-off of a ReLU/LeakyReLU kink the gradient is bimodal (the fp64 oracle itself jumps by ~1e-4 relative under
    # 1e-7 parameter perturbations at the 15-block C-Town point), so the check accepts agreement with the fp64
    # gradient on either side of such a kink: the base point or one of 8 perturbed evaluations.
    def grads64(params64):
        l = {k: v.clone().requires_grad_(True) for k, v in params64.items()}
        o = oracle.gatres_forward(l, xin.double(), ei, num_blocks=nb)
        torch.nn.functional.mse_loss(o[mask], y.double()[mask]).backward()
        return {k: v.grad for k, v in l.items()}

    def judge(g64):
        gscale = max(float(v.abs().max()) for v in g64.values())
        worst_ratio, worst_rel, bad, kink = 0.0, 0.0, [], True
        for (k, ref), prm in zip(leaves.items(), model.parameters()):
            e_hip = float((prm.grad.double().cpu() - g64[k]).abs().max())
            e_ref = float((ref.grad.double() - g64[k]).abs().max())
            floor = max(1e-5 * float(g64[k].abs().max()), 2e-6 * gscale)
            if e_hip > max(2 * e_ref, floor):
                bad.append((k, e_hip, e_ref))
                # a kink can only excuse a tensor on which the fp32 ORACLE is off its own fp64 gradient as well
                kink = kink and e_ref > floor
            worst_ratio = max(worst_ratio, e_hip / max(e_ref, 1e-30))
            worst_rel = max(worst_rel, e_hip / gscale)
        return bad, worst_rel, worst_ratio, kink

    for k, prm in zip(leaves, model.parameters()):
        assert prm.grad is not None and prm.grad.shape == leaves[k].shape, k
    key = (nb, nc, bs, nodes, pipes)
    if key not in _G64_CACHE:
        _G64_CACHE[key] = [grads64(p64)]
    bad, worst_rel, worst_ratio, kink = judge(_G64_CACHE[key][0])
    branch = 0
    gen = torch.Generator().manual_seed(1234)
    # the kink fallback fires only where it demonstrably applies: every failing tensor is one on which the fp32 oracle
    # itself disagrees with its fp64 gradient beyond the tolerance floor (committed runs: never needed, branch 0)
    while bad and kink and branch < 8:
        branch += 1
        if len(_G64_CACHE[key]) <= branch:
            pert = {k: v * (1 + 1e-7 * torch.randn(v.shape, generator=gen, dtype=torch.float64)) for k, v in p64.items()}
            _G64_CACHE[key].append(grads64(pert))
        bad, worst_rel, worst_ratio, _ = judge(_G64_CACHE[key][branch])
    assert not bad, (bad[:3], kink, branch)
    flat_hip = torch.cat([q.grad.reshape(-1) for q in model.parameters()])
    flat_ref = torch.cat([v.grad.reshape(-1) for v in leaves.values()])
    note(f"bwd nb{nb} nc{nc} {'fused' if fused else 'per-op'}: flat grad rel err vs oracle32 / worst err over |g|max vs "
         f"fp64 / worst (hip err)/(oracle32 err) / fp64 kink branch used", [relerr(flat_hip, flat_ref), worst_rel,
                                                                            worst_ratio, branch])


def test_module_surface_matches_reference(pkg, oracle):
    m = pkg.GATResMeanConv(name="x", num_blocks=2, nc=8)
    assert m.name == "x" and m.num_blocks == 2
    assert [type(b).__name__ for b in m.blocks] == ["GResBlockMeanConv"] * 2
    keys = set(m.state_dict().keys())
    exp = set()
    for k in oracle.param_shapes(2, 8):
        exp.add(k)
        if "lin_src" in k:
            exp.add(k.replace("lin_src", "lin_dst"))
    assert keys == exp
    for k, s in oracle.param_shapes(2, 8).items():
        assert tuple(m.state_dict()[k].shape) == s
    # PyG >= 2.5 spelling of the shared projection
    sd = {k.replace("lin_src.weight", "lin.weight"): v for k, v in m.state_dict().items() if "lin_dst" not in k}
    m2 = pkg.GATResMeanConv(num_blocks=2, nc=8)
    m2.load_state_dict(sd)
    assert torch.equal(m2.flat_parameters, m.flat_parameters)
    x, _, ei, _ = ctown_batch(pkg, 1, 20, 24)
    m = m.cuda()
    with pytest.raises(ValueError):
        m(x, ei.cuda())                                   # CPU tensor: there is no CPU path
    with pytest.raises(ValueError):
        m(x.cuda(), ei.cuda(), None, torch.zeros(ei.shape[1], 1).cuda())
    with pytest.raises(ValueError):
        pkg.GATResMeanConv(num_blocks=1, nc=24)
    with torch.no_grad():
        o = m(x.cuda(), ei.cuda())
    assert not o.requires_grad
    m3 = copy.deepcopy(m)
    assert relerr(m3(x.cuda(), ei.cuda()), o) == 0.0


def test_drop_in_training_loop_with_torch_adam(pkg, oracle):
    """The reference's loop body verbatim (train.py:160-188) on the HIP module vs the oracle, 4 iterations."""
    nb, nc, bs = 4, 32, 4
    model, p = build(pkg, oracle, nb, nc, seed=11)
    ref = oracle.OracleTrainer(p)
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, weight_decay=6e-6)
    snaps = pkg.wdn_synth.make_snapshots(16, 388, seed=5)
    ei1 = pkg.wdn_synth.make_wdn_topology()
    rng = np.random.RandomState(0)
    for it in range(4):
        y = pkg.wdn_synth.collate_snapshots(snaps, range(it * bs, (it + 1) * bs))
        edge_index = pkg.wdn_synth.collate_edge_index(ei1, 388, bs)          # a fresh tensor per batch, like PyG
        batch_mask = pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, rng)
        l_ref, _ = ref.step(y.clone(), y, edge_index, batch_mask)
        opt.zero_grad()
        dx, dy, dei = y.clone().cuda(), y.cuda(), edge_index.cuda()
        dx[batch_mask] = 0
        out = model(dx, dei, None, None)
        loss = torch.nn.functional.mse_loss(out[batch_mask], dy[batch_mask])
        loss.backward()
        opt.step()
        assert relerr(loss, l_ref) < 1e-5, it
    e = relerr(model.flat_parameters, ref.flat("params"))
    note("drop-in 4 Adam steps: flat param rel err", e)
    assert e < 1e-4


def test_parameter_gradients_delivered_in_place_equal_autograds(pkg, oracle):
    """``direct_param_grads`` (the default): loss.backward() writes the flat gradient into persistent ``.grad`` views and no
    AccumulateGrad node runs.  Held bit for bit to the gradients the same module delivers THROUGH autograd
    (``direct_param_grads = False``) under every state the ``.grad`` fields can be in: None (``zero_grad()``), the module's
    own views (``zero_grad(set_to_none=False)``, accumulation over two backwards), a mixture with foreign tensors; a tensor
    hook switches the path off by itself; the previous backward's gradients survive the next one."""
    nb, nc, bs = 3, 32, 2
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
    snaps = pkg.wdn_synth.make_snapshots(4, 388, seed=21)
    ya = pkg.wdn_synth.collate_snapshots(snaps, range(0, bs)).cuda()
    yb = pkg.wdn_synth.collate_snapshots(snaps, range(bs, 2 * bs)).cuda()
    md, _ = build(pkg, oracle, nb, nc, seed=61)
    ma, _ = build(pkg, oracle, nb, nc, seed=61)
    ma.direct_param_grads = False
    assert md.direct_param_grads

    def bwd(model, y):
        out = model(y, ei)
        (out * out).mean().backward()
        return out

    def flat(model):
        return torch.cat([p.grad.reshape(-1) for p in model.parameters()])

    # 1. every .grad None
    o1, o2 = bwd(md, ya), bwd(ma, ya)
    assert o1.grad_fn is not None and torch.equal(o1, o2)
    assert torch.equal(flat(md), flat(ma))
    fg = md.flat_grad()
    assert fg is not None and torch.equal(fg, flat(ma)) and ma.flat_grad() is None
    assert all(p.grad.shape == p.shape and p.grad.is_contiguous() for p in md.parameters())
    held = [p.grad for p in md.parameters()]
    held_vals = flat(md).clone()
    # 2. zero_grad() (set_to_none=True), another batch: fresh views; the gradients held from step 1 are intact
    md.zero_grad(); ma.zero_grad()
    assert md.flat_grad() is None
    bwd(md, yb); bwd(ma, yb)
    assert torch.equal(flat(md), flat(ma))
    assert torch.equal(torch.cat([g.reshape(-1) for g in held]), held_vals)
    # 3. accumulation on top of the module's own views
    bwd(md, ya); bwd(ma, ya)
    assert torch.equal(flat(md), flat(ma)) and md.flat_grad() is not None
    # 4. zero_grad(set_to_none=False): zeros in place, then accumulate
    md.zero_grad(set_to_none=False); ma.zero_grad(set_to_none=False)
    assert float(flat(md).abs().max()) == 0.0
    bwd(md, yb); bwd(ma, yb)
    assert torch.equal(flat(md), flat(ma))
    # 5. a mixture: one gradient replaced by a foreign tensor, one set to None
    for model in (md, ma):
        model.lin0.weight.grad = model.lin0.weight.grad.clone()
        model.lin1.bias.grad = None
    bwd(md, ya); bwd(ma, ya)
    assert torch.equal(flat(md), flat(ma)) and md.flat_grad() is None
    # 6. torch.optim.Adam and FusedAdam take the views
    md.zero_grad(); ma.zero_grad()
    od = torch.optim.Adam(md.parameters(), lr=5e-4, weight_decay=6e-6)
    oa = torch.optim.Adam(ma.parameters(), lr=5e-4, weight_decay=6e-6)
    for y in (ya, yb, ya):
        for model, opt in ((md, od), (ma, oa)):
            opt.zero_grad()
            bwd(model, y)
            opt.step()
    assert torch.equal(md.flat_parameters, ma.flat_parameters)
    # 7. a tensor hook: the gradient must pass through autograd, and does
    seen = []
    h = md.lin1.weight.register_hook(lambda g: seen.append(float(g.abs().sum())) or g * 2.0)
    md.zero_grad(); ma.zero_grad()
    bwd(md, ya); bwd(ma, ya)
    assert len(seen) == 1 and torch.equal(md.lin1.weight.grad, 2.0 * ma.lin1.weight.grad)
    assert torch.equal(md.lin0.weight.grad, ma.lin0.weight.grad)
    h.remove()
    # 8. torch.autograd.grad needs the autograd path: documented switch
    md.direct_param_grads = False
    gs = torch.autograd.grad((md(ya, ei) ** 2).mean(), list(md.parameters()))
    ma.zero_grad(); bwd(ma, ya)
    assert torch.equal(torch.cat([g.reshape(-1) for g in gs]), flat(ma))


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "per_op"])
@pytest.mark.parametrize("use_graph", [False, True])
def test_native_train_step_matches_oracle(pkg, oracle, use_graph, fused):
    nb, nc, bs = 15, 32, 2
    model, p = build(pkg, oracle, nb, nc, seed=4, fused=fused)
    ref = oracle.OracleTrainer(p)
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs)
    tr = pkg.GATResTrainer(model, ei.cuda(), 388 * bs, nodes_per_graph=[388] * bs, use_graph=use_graph, fused=fused)
    assert tr.fused == fused
    snaps = pkg.wdn_synth.make_snapshots(8, 388, seed=6)
    rng = np.random.RandomState(1)
    for it in range(3):
        y = pkg.wdn_synth.collate_snapshots(snaps, range(it * bs, (it + 1) * bs))
        mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, rng))
        l_ref, o_ref = ref.step(y.clone(), y, ei, mask)
        loss = tr.step(y.cuda(), y.cuda(), mask.cuda())
        assert relerr(loss, l_ref) < 2e-5, it
        assert relerr(tr.out, o_ref) < 2e-5, it
        if it == 0:
            g = relerr(tr.grads, ref.flat("grads"))
            note(f"native step (graph={use_graph}, fused={fused}) flat grad rel err", g)
            assert g < 1e-4
    assert tr.optimizer_step == 3
    # Adam divides by sqrt(v): parameters whose gradient is fp32 cancellation noise move by ~lr per step in a
    # noise-determined direction, in the fp32 oracle just as here.  So: the bulk must agree tightly, nothing may
    # differ by more than the 3 steps could move it, and the loss/prediction trajectory above pins the rest.
    diff = (model.flat_parameters.detach().cpu().double() - ref.flat("params").double()).abs()
    frac_off = float((diff > 1e-5).double().mean())
    note(f"native step (graph={use_graph}, fused={fused}) 3 steps: max |dp| / fraction of params off by > 1e-5",
         [float(diff.max()), frac_off])
    assert float(diff.max()) <= 3 * 5e-4 * 1.01 and frac_off < 0.01
    assert relerr(model.state_dict()["lin0.weight"], ref.flat("params")[:nc]) < 1e-4   # the module sees the update
    # device-side mask path: exact count per graph, loss finite, fresh mask per step
    tr.step(y.cuda(), y.cuda())
    m1 = tr.mask.clone()
    tr.step(y.cuda(), y.cuda())
    assert int(m1.sum()) == 2 * 368 and int(tr.mask.sum()) == 2 * 368 and not torch.equal(m1, tr.mask)
    assert torch.isfinite(tr.loss).all()


@pytest.mark.parametrize("nc", [8, 16])
def test_window_kernel_narrow_models(pkg, oracle, nc):
    """The window kernel at nc = 8 / 16 (different lane groups, W slot sizes, alpha-table fits) against the per-op path
    and the oracle on C-Town-sized graphs."""
    nb, bs = 4, 3
    x, y, ei, mask = ctown_batch(pkg, bs, 388, 430)
    mf, p = build(pkg, oracle, nb, nc, seed=61, fused=True)
    mp, _ = build(pkg, oracle, nb, nc, seed=61, fused=False)
    tf = pkg.GATResTrainer(mf, ei.cuda(), 388 * bs, nodes_per_graph=[388] * bs, use_graph=False, fused=True)
    tp = pkg.GATResTrainer(mp, ei.cuda(), 388 * bs, nodes_per_graph=[388] * bs, use_graph=False, fused=False)
    ref = oracle.OracleTrainer(p)
    for it in range(2):
        l_ref, o_ref = ref.step(y.clone(), y, ei, mask)
        lf = tf.step(y.cuda(), y.cuda(), mask.cuda())
        lp = tp.step(y.cuda(), y.cuda(), mask.cuda())
        if it == 0:
            assert torch.equal(tf.out, tp.out)
        assert relerr(tf.out, o_ref) < 2e-5 and relerr(lf, l_ref) < 2e-5
        assert relerr(tf.grads, tp.grads) < 2e-5 and relerr(tf.grads, ref.flat("grads")) < 1e-4


@pytest.mark.parametrize("split", [3, 4, 7, 8])
def test_window_kernel_on_a_ragged_batch(pkg, oracle, split, monkeypatch):
    """Graphs of different sizes whose node order is local (compact row windows): the window kernel carries them,
    including a 40-node graph that leaves parts without rows.  Training steps must match the per-op path."""
    monkeypatch.setenv("GATRES_FUSED_SPLIT", str(split))
    nb, nc = 3, 32
    sizes = [(388, 430), (40, 45), (200, 230), (388, 430)]
    tops = [pkg.wdn_synth.make_wdn_topology(n, e, seed=7 + i) for i, (n, e) in enumerate(sizes)]
    offs = np.cumsum([0] + [n for n, _ in sizes])
    ei = torch.cat([t + int(o) for t, o in zip(tops, offs[:-1])], dim=1)
    N = int(offs[-1])
    npg = [n for n, _ in sizes]
    y = torch.randn(N, generator=torch.Generator().manual_seed(4))
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask(npg, 0.95, np.random.RandomState(2)))
    mf, _ = build(pkg, oracle, nb, nc, seed=41, fused=True)
    mp, _ = build(pkg, oracle, nb, nc, seed=41, fused=False)
    tf = pkg.GATResTrainer(mf, ei.cuda(), N, nodes_per_graph=npg, use_graph=False, fused=True)
    tp = pkg.GATResTrainer(mp, ei.cuda(), N, nodes_per_graph=npg, use_graph=False, fused=False)
    assert tf.plan.window_rows(split) < 200           # compact windows: the window kernel applies
    for it in range(3):
        lf = tf.step(y.cuda(), y.cuda(), mask.cuda())
        lp = tp.step(y.cuda(), y.cuda(), mask.cuda())
        assert torch.equal(tf.out, tp.out) or relerr(tf.out, tp.out) < 1e-6, it   # bit-identical on the first step
        if it == 0:
            assert torch.equal(tf.out, tp.out)
        assert relerr(lf, lp) < 1e-6
        assert torch.isfinite(tf.grads).all() and relerr(tf.grads, tp.grads) < 2e-5, (it, relerr(tf.grads, tp.grads))


@pytest.mark.parametrize("use_graph", [False, True])
def test_transposed_weights_follow_external_parameter_changes(pkg, oracle, use_graph):
    """The fused Adam pass keeps the backward's transposed conv weights current, and the trainer then skips the
    transposes (GATRES_FLAG_WT_VALID) -- but only while nobody else touches the parameters.  Overwriting them through
    torch between two steps must be noticed (torch's version counter): gradients of the next step match a trainer that
    starts from the overwritten parameters."""
    nb, nc, bs = 3, 32, 2
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
    y = pkg.wdn_synth.collate_snapshots(pkg.wdn_synth.make_snapshots(4, 388, seed=2), range(bs)).cuda()
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, np.random.RandomState(5))).cuda()
    model, _ = build(pkg, oracle, nb, nc, seed=31)
    tr = pkg.GATResTrainer(model, ei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=use_graph)
    for _ in range(3):
        tr.step(y, y, mask)                          # from the second step on the transposes are skipped
    fresh, _ = build(pkg, oracle, nb, nc, seed=77)   # other weights
    with torch.no_grad():
        model.flat_parameters.copy_(fresh.flat_parameters)
    tr.forward_backward(y, y, mask)
    g_after = tr.grads.clone()
    tr.step(y, y, mask)                              # and through the (graph) step path as well
    tr2 = pkg.GATResTrainer(fresh, ei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=False)
    tr2.forward_backward(y, y, mask)
    assert relerr(g_after, tr2.grads) < 1e-6
    assert relerr(tr.grads, tr2.grads) < 1e-6


def test_fused_adam_matches_torch_adam(pkg, oracle):
    """FusedAdam (one native launch on the flat parameter vector, gradients taken in place from the flat buffer the
    module's backward hands out) against torch.optim.Adam on a twin model, reference loop body, 4 steps; also with a
    gradient that is NOT one flat buffer (accumulated from a foreign tensor) and with a state_dict round trip."""
    nb, nc, bs = 3, 32, 2
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
    snaps = pkg.wdn_synth.make_snapshots(8, 388, seed=12)
    ma, _ = build(pkg, oracle, nb, nc, seed=51)
    mb, _ = build(pkg, oracle, nb, nc, seed=51)
    oa = pkg.FusedAdam(ma, lr=5e-4, weight_decay=6e-6)
    ob = torch.optim.Adam(mb.parameters(), lr=5e-4, weight_decay=6e-6)
    rng = np.random.RandomState(3)
    for it in range(4):
        y = pkg.wdn_synth.collate_snapshots(snaps, range(it * bs, (it + 1) * bs)).cuda()
        m = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, rng)).cuda()
        x = y.clone(); x[m] = 0
        for model, opt in ((ma, oa), (mb, ob)):
            opt.zero_grad()
            loss = torch.nn.functional.mse_loss(model(x, ei)[m], y[m])
            loss.backward()
            if it == 2:                      # break the "one flat buffer" layout for one parameter
                model.lin0.weight.grad = model.lin0.weight.grad.clone()
            opt.step()
        if it == 1:                          # optimizer checkpoint round trip
            oa2 = pkg.FusedAdam(ma, lr=5e-4, weight_decay=6e-6)
            oa2.load_state_dict(oa.state_dict())
            oa = oa2
        d = (ma.flat_parameters - mb.flat_parameters).abs()
        # Adam's update is lr * m / (sqrt(v) + eps): one ulp of difference in a noise-sized gradient moves a weight
        # by up to ~lr; the bulk must agree to fp32 rounding
        assert float(d.max()) <= (it + 1) * 5e-4 * 1.01 and float((d > 1e-6).double().mean()) < 0.01, (it, float(d.max()))


def test_targets_are_inputs_alias(pkg, oracle):
    """targets_are_inputs=True lets x and y share one device buffer (the kernels never overwrite x: the masking is
    applied on the fly): same loss and gradients as with separate buffers, for both kernel paths."""
    nb, nc, bs = 2, 32, 2
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
    y = pkg.wdn_synth.collate_snapshots(pkg.wdn_synth.make_snapshots(2, 388, seed=8), range(bs)).cuda()
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, np.random.RandomState(6))).cuda()
    for fused in (True, False):
        res = []
        for alias in (False, True):
            model, _ = build(pkg, oracle, nb, nc, seed=13, fused=fused)
            tr = pkg.GATResTrainer(model, ei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=False, fused=fused,
                                   targets_are_inputs=alias)
            tr.step(y, y, mask)
            tr.step(y, y, mask)
            res.append((tr.loss.clone(), tr.grads.clone(), tr.x.clone()))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        assert torch.equal(res[1][2], y.reshape(-1))  # x is still the unmasked batch


def test_prefetched_host_batches_match_direct_loading(pkg, oracle):
    """prefetch_batch (side-stream H2D into a staging slot) + commit_batch give the same training trajectory as
    load_batch, including the two-slot rotation -- and so does step_prefetched(), which trains on the staging slot in place."""
    nb, nc, bs = 2, 32, 2
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
    snaps = pkg.wdn_synth.make_snapshots(5 * bs, 388, seed=15)
    host = [pkg.wdn_synth.collate_snapshots(snaps, range(i * bs, (i + 1) * bs)).reshape(-1).pin_memory() for i in range(5)]
    losses = []
    for mode in ("direct", "prefetch", "in_place"):
        model, _ = build(pkg, oracle, nb, nc, seed=19)
        tr = pkg.GATResTrainer(model, ei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=True, seed=3,
                               targets_are_inputs=True)
        out = []
        if mode != "direct":
            tr.prefetch_batch(host[0])
        for i in range(5):
            if mode == "in_place":                     # (round 5: the step reads the staging slot itself, mask sampled ahead)
                tr.step_prefetched()
                if i + 1 < 5:
                    tr.prefetch_batch(host[i + 1])
                out.append(float(tr.loss.item()))
                continue
            if mode == "prefetch":
                tr.commit_batch()
                if i + 1 < 5:
                    tr.prefetch_batch(host[i + 1])
            else:
                tr.load_batch(host[i].cuda(), host[i].cuda())
            tr.run_step(device_mask=True)
            out.append(float(tr.loss.item()))
        losses.append(out)
    assert losses[0] == losses[1] == losses[2], losses


def test_reset_sync_recovers_a_corrupted_barrier_state(pkg, oracle):
    """Scribbling over the barrier epochs in scratch makes the parts of a split snapshot miss each other (bounded
    spins -> NaN, no hang); gatres_fused_reset_sync puts the state back and the next step is correct again."""
    nb, nc, bs = 2, 32, 2
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
    y = pkg.wdn_synth.collate_snapshots(pkg.wdn_synth.make_snapshots(2, 388, seed=8), range(bs)).cuda()
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, np.random.RandomState(6))).cuda()
    model, _ = build(pkg, oracle, nb, nc, seed=23)
    tr = pkg.GATResTrainer(model, ei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=False)
    lib = pkg._native.load()
    if lib.gatres_fused_cus_per_segment(model._cmodel_ref(), tr.plan.ref()) < 2:
        pytest.skip("snapshots are not split on this device")
    tr.forward_backward(y, y, mask)
    good = tr.grads.clone()
    assert torch.isfinite(good).all()
    tr.scratch.view(torch.int32)[-(1 << 22):].random_(1, 1000)            # the barrier state sits near the end of scratch
    tr.forward_backward(y, y, mask)
    torch.cuda.synchronize()
    pkg._native.check(lib.gatres_fused_reset_sync(model._cmodel_ref(), tr.plan.ref(), tr.scratch.data_ptr(),
                                                 pkg._native.current_stream(tr.device)), "reset")
    tr.forward_backward(y, y, mask)
    assert torch.equal(tr.grads, good)


def test_edge_cases_and_determinism(pkg, oracle):
    model, p = build(pkg, oracle, 2, 8, seed=9)
    n = 30
    cases = {
        "no_edges": torch.zeros((2, 0), dtype=torch.int64),
        "self_loops_dups_isolated": torch.tensor([[0, 1, 1, 2, 2, 5, 5, 7, 3, 3], [1, 0, 0, 2, 2, 6, 6, 7, 4, 4]]),
        "hub": torch.stack([torch.arange(1, 29), torch.zeros(28, dtype=torch.int64)]),
    }
    x = torch.randn(n, 1)
    for name, ei in cases.items():
        ref = oracle.gatres_forward(p, x, ei)
        out = model(x.cuda(), ei.cuda())
        assert relerr(out, ref) < 1e-5, name
        assert torch.equal(out, model(x.cuda(), ei.clone().cuda())), name      # bitwise reproducible
    # K6 on the device: a batch equals the concatenation of its graphs
    x2, _, ei2, _ = ctown_batch(pkg, 3, 50, 58)
    one = pkg.wdn_synth.make_wdn_topology(50, 58)
    whole = model(x2.cuda(), ei2.cuda())
    parts = torch.cat([model(x2[i * 50:(i + 1) * 50].cuda(), one.cuda()) for i in range(3)])
    assert torch.equal(whole, parts)


def test_full_size_properties_bs32(pkg, oracle):
    """BASELINE config 2 size (gatres_small, C-Town, bs=32): size-independent properties instead of the slow oracle."""
    model, p = build(pkg, oracle, 15, 32, seed=3)
    x, y, ei, mask = ctown_batch(pkg, 32)
    dx, dei = x.cuda(), ei.cuda()
    out = model(dx, dei)
    assert torch.isfinite(out).all()
    one = pkg.wdn_synth.make_wdn_topology()
    first = model(dx[:388], one.cuda())
    assert torch.equal(out[:388], first)                                        # block-diagonal independence
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(0))
    assert relerr(model(dx, ei[:, perm].cuda()), out) < 1e-5                    # edge-order invariance
    # gradient of a linear functional is linear in the upstream gradient
    g1, g2 = torch.randn_like(out), torch.randn_like(out)
    def grad_for(g):
        model.zero_grad()
        model(dx, dei).backward(g)
        return torch.cat([q.grad.reshape(-1) for q in model.parameters()]).clone()
    ga, gb, gab = grad_for(g1), grad_for(g2), grad_for(g1 + 2 * g2)
    assert relerr(gab, ga + 2 * gb) < 1e-4
    # the oracle on one graph of the batch pins the values
    ref = oracle.gatres_forward(p, x[:388], one)
    assert relerr(first, ref) < 1e-5


def test_fused_equals_per_op_bitwise_forward(pkg, oracle):
    """Same arithmetic, statement for statement: the fused per-snapshot kernel and the per-op kernels must give
    bit-identical predictions (LDS-cached and uncached variants), and gradients equal up to the slab partition."""
    for nb, nc, nodes, pipes, bs in [(15, 32, 388, 430, 3), (3, 32, 450, 500, 2), (2, 128, 90, 100, 2), (3, 8, 25, 28, 7),
                                     (15, 32, 388, 430, 40)]:
        x, y, ei, mask = ctown_batch(pkg, bs, nodes, pipes)
        mf, p = build(pkg, oracle, nb, nc, seed=5, fused=True)
        mp, _ = build(pkg, oracle, nb, nc, seed=5, fused=False)
        dx, dei = x.cuda(), ei.cuda()
        of, op = mf(dx, dei), mp(dx, dei)
        assert mf._plans.get(dei, dx.shape[0]).num_segments >= 1 and mp._plans.get(dei, dx.shape[0]).num_segments == 0
        assert torch.equal(of, op), (nb, nc, nodes)
        with torch.no_grad():
            assert torch.equal(mf(dx, dei), of)          # inference path (no saved activations) == training path
        g = torch.randn(of.shape, generator=torch.Generator().manual_seed(nb * 1000 + nc)).cuda()
        of.backward(g)
        op.backward(g)
        gf = torch.cat([q.grad.reshape(-1) for q in mf.parameters()])
        gp = torch.cat([q.grad.reshape(-1) for q in mp.parameters()])
        # parameter gradients: same stage arithmetic, different slab partition (per segment vs per node range)
        assert relerr(gf, gp) < 2e-5, (nb, nc, nodes, relerr(gf, gp))


@pytest.mark.parametrize("nc", [16, 32])
def test_parameter_gradient_items_on_segments_of_any_size(pkg, oracle, nc, monkeypatch):
    """The register-streamed parameter-gradient items (param_grads_reg_kernel: a wave takes 4-row steps round-robin, the
    last step of a segment may hold 1 .. 3 rows, a wave may hold no step at all) on graphs whose node counts are NOT
    multiples of four -- 37, 201, 390 and a 6-node graph that leaves two of an item's four waves without rows -- in the
    stand-alone launch (the default since round 5), against the per-op path and the oracle over three training steps."""
    nb = 3
    sizes = [(390, 433), (37, 41), (201, 230), (6, 5)]
    tops = [pkg.wdn_synth.make_wdn_topology(n, e, seed=11 + i) for i, (n, e) in enumerate(sizes)]
    offs = np.cumsum([0] + [n for n, _ in sizes])
    ei = torch.cat([t + int(o) for t, o in zip(tops, offs[:-1])], dim=1)
    N = int(offs[-1])
    npg = [n for n, _ in sizes]
    y = torch.randn(N, 1, generator=torch.Generator().manual_seed(14))
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask(npg, 0.95, np.random.RandomState(12)))
    mf, p = build(pkg, oracle, nb, nc, seed=43, fused=True)
    mp, _ = build(pkg, oracle, nb, nc, seed=43, fused=False)
    tf = pkg.GATResTrainer(mf, ei.cuda(), N, nodes_per_graph=npg, use_graph=False, fused=True)
    tp = pkg.GATResTrainer(mp, ei.cuda(), N, nodes_per_graph=npg, use_graph=False, fused=False)
    ref = oracle.OracleTrainer(p)
    for it in range(3):
        l_ref, o_ref = ref.step(y.clone(), y, ei, mask)
        lf = tf.step(y.cuda(), y.cuda(), mask.cuda())
        lp = tp.step(y.cuda(), y.cuda(), mask.cuda())
        if it == 0:
            assert torch.equal(tf.out, tp.out)
        assert relerr(tf.out, o_ref) < 2e-5 and relerr(lf, l_ref) < 2e-5 and relerr(lf, lp) < 1e-6
        e_pp, e_or = relerr(tf.grads, tp.grads), relerr(tf.grads, ref.flat("grads"))
        assert torch.isfinite(tf.grads).all() and e_pp < 2e-5 and e_or < 1e-4, (it, e_pp, e_or)
    note(f"param_grad_items_any_size_nc{nc}", {"grads_vs_per_op": e_pp, "grads_vs_oracle": e_or})


@pytest.mark.parametrize("split,mode", [(1, ""), (2, ""), (4, ""), (8, ""), (3, ""), (7, ""), (4, "GATRES_FUSED_NO_HALO"),
                                        (4, "GATRES_FUSED_SAFE_SYNC"), (2, "GATRES_FUSED_WITH_CONSUMERS"),
                                        (4, "GATRES_FUSED_WITH_CONSUMERS")])
def test_fused_split_over_cus_matches_per_op(pkg, oracle, split, mode, monkeypatch):
    """One snapshot carried by 1 / 2 / 4 / 8 workgroups (row windows + flag barriers + halo pulls): predictions stay
    bit-identical to the per-op kernels, gradients agree up to the slab partition.  The batch is ragged (a 388-node
    graph, a 40-node graph whose tiles do not cover every part, a 388-node graph with SHUFFLED node ids: most of its
    neighbours are halo rows) and the launches are repeated so that the persistent barrier epochs are exercised."""
    monkeypatch.setenv("GATRES_FUSED_SPLIT", str(split))
    monkeypatch.setenv("GATRES_REORDER", "0")          # keep the shuffled ids: this test is about large halos
    if mode:                 # the fallbacks: bulk pulls, agent-scope barriers; parameter gradients on consumer workgroups of the launch
        monkeypatch.setenv(mode, "1")
    nb, nc = 4, 32
    t_big = pkg.wdn_synth.make_wdn_topology(388, 430, seed=0)
    t_small = pkg.wdn_synth.make_wdn_topology(40, 45, seed=1)
    perm = torch.from_numpy(np.random.RandomState(3).permutation(388))
    t_shuf = perm[pkg.wdn_synth.make_wdn_topology(388, 430, seed=2)]
    t_shuf = t_shuf[:, torch.argsort(t_shuf[0], stable=True)]
    ei = torch.cat([t_big, t_small + 388, t_shuf + 428], dim=1)
    N = 388 + 40 + 388
    x = torch.randn(N, 1, generator=torch.Generator().manual_seed(9))
    mf, _ = build(pkg, oracle, nb, nc, seed=21, fused=True)          # a fresh model = a fresh (zeroed) scratch buffer
    mp, _ = build(pkg, oracle, nb, nc, seed=21, fused=False)
    dx, dei = x.cuda(), ei.cuda()
    plan = mf._plans.get(dei, N)
    assert plan.num_segments == 3
    lib = pkg._native.load()
    assert lib.gatres_fused_cus_per_segment(mf._cmodel_ref(), plan.ref()) == split
    for rep in range(3):
        of, op = mf(dx, dei), mp(dx, dei)
        assert torch.equal(of, op), (split, rep)
        g = torch.randn(of.shape, generator=torch.Generator().manual_seed(rep)).cuda()
        for m in (mf, mp):
            m.zero_grad()
        of.backward(g)
        op.backward(g)
        gf = torch.cat([q.grad.reshape(-1) for q in mf.parameters()])
        gp = torch.cat([q.grad.reshape(-1) for q in mp.parameters()])
        assert torch.isfinite(gf).all()
        assert relerr(gf, gp) < 2e-5, (split, rep, relerr(gf, gp))
        with torch.no_grad():
            assert torch.equal(mf(dx, dei), of)                      # inference launches (forward-only barriers)


@pytest.mark.parametrize("fused", [True, False], ids=["fused", "per_op"])
@pytest.mark.parametrize("name", ["tiny_nb2_nc8", "ctown_small_bs2"])
def test_golden_vectors(pkg, name, fused):
    """Committed fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle): one reference
    training iteration -> predictions, loss, gradients, weights after one Adam step."""
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", name + ".npz"))
    nb, nc, npg = int(d["num_blocks"]), int(d["nc"]), int(d["nodes_per_graph"])
    model = pkg.GATResMeanConv(num_blocks=nb, nc=nc, fused=fused).cuda()
    with torch.no_grad():
        model.flat_parameters.copy_(torch.from_numpy(d["params"]))
    x = torch.from_numpy(d["x"]).cuda()
    ei = torch.from_numpy(d["edge_index"]).cuda()
    bs = x.shape[0] // npg
    tr = pkg.GATResTrainer(model, ei, x.shape[0], nodes_per_graph=[npg] * bs, use_graph=False, fused=fused)
    loss = tr.step(x, x, torch.from_numpy(d["mask"]).cuda())
    assert relerr(tr.out, torch.from_numpy(d["out"])) < 1e-5
    assert relerr(loss, torch.tensor(float(d["loss"]))) < 1e-5
    assert relerr(tr.grads, torch.from_numpy(d["grads"])) < 1e-5
    # Parameters after the first Adam step, entry by entry (the old bound -- 1e-3 on up to 1 % of the entries --
    # pinned little).  The first update is lr * g / (|g| + eps): its derivative in g is at most 1 / (|g| + eps), so an entry
    # whose gradient differs from the fixture's by dg may differ by lr * dg / (|g| + eps) (x 2 for the curvature near g = 0)
    # plus fp32 round-off of the parameter itself -- and by nothing else.
    # (g here is the gradient Adam sees: the loss gradient + weight_decay * p, train.py:348)
    lr, eps, wd = 5e-4, 1e-8, 6e-6
    g_gold, g_hip = torch.from_numpy(d["grads"]).double(), tr.grads.detach().cpu().double()
    p_gold, p_hip = torch.from_numpy(d["params_after"]).double(), model.flat_parameters.detach().cpu().double()
    g_eff = g_gold + wd * torch.from_numpy(d["params"]).double()
    allow = 2 * lr * (g_hip - g_gold).abs() / (g_eff.abs() + eps) + 4e-7 * (1 + p_gold.abs())
    worst = float(((p_hip - p_gold).abs() - allow).max())
    assert worst <= 0, worst
    assert float((p_hip - p_gold).abs().max()) <= 2 * lr


def test_collective_path_equals_single_call(pkg, oracle):
    """The multi-GPU sequence (mask+fwd+bwd | all-reduce | Adam as separate graph replays) must give exactly the
    single-call step at world size 1."""
    bs = 4
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
    snaps = pkg.wdn_synth.make_snapshots(12, 388, seed=8).cuda()
    res = []
    for split in (False, True):
        model, _ = build(pkg, oracle, 15, 32, seed=2)
        tr = pkg.GATResTrainer(model, ei, 388 * bs, nodes_per_graph=[388] * bs, seed=77, force_collective_path=split)
        assert tr.split == split
        losses = []
        for it in range(3):
            y = snaps[it * bs:(it + 1) * bs].reshape(-1)
            losses.append(float(tr.step(y, y)))
        res.append((losses, model.flat_parameters.clone(), tr.mask.clone()))
    assert res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])


@pytest.mark.gpu
@pytest.mark.parametrize("directed", [False, True], ids=["symmetric", "directed"])
def test_part_tables_equal_in_kernel_derivation(pkg, oracle, directed, monkeypatch):
    """The window kernel with the plan's part tables (gatres_graph_part_tables_host: prologue = one LDS-DMA copy, hand-off
    lists de-duplicated) and without them (GATRES_NO_PART_TABLES=1: tables derived from the CSR arrays in every launch)
    must agree bit for bit: out, loss, gradients, parameters after two steps.  `directed`: half of the reverse edges
    removed -- the plan then lacks GATRES_GRAPH_SYMMETRIC and the hand-offs keep their heartbeat pacing."""
    nb, nc, bs = 4, 32, 3
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(388, 430, seed=0), 388, bs)
    if directed:
        keep = (ei[0] < ei[1]) | (torch.arange(ei.shape[1]) % 2 == 0)
        ei = ei[:, keep]
    N = 388 * bs
    snaps = pkg.wdn_synth.make_snapshots(2 * bs, 388, seed=4)
    res = []
    for no_tables in (False, True):
        if no_tables:
            monkeypatch.setenv("GATRES_NO_PART_TABLES", "1")
        model, _ = build(pkg, oracle, nb, nc, seed=3, fused=True)
        tr = pkg.GATResTrainer(model, ei.cuda(), N, nodes_per_graph=[388] * bs, use_graph=False)
        assert bool(tr._gstruct.part_tables) == (not no_tables) and not tr.plan.c.part_tables      # (the shared struct is never rewritten)
        assert (tr.plan.flags & 1) == (0 if directed else 1)
        lib = pkg._native.load()
        assert lib.gatres_fused_window_kernel(model._cmodel_ref(), tr.plan.ref()) == 1
        rng = np.random.RandomState(6)
        out = []
        for it in range(2):
            y = pkg.wdn_synth.collate_snapshots(snaps, range(it * bs, (it + 1) * bs))
            mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, rng))
            loss = tr.step(y.cuda(), y.cuda(), mask.cuda())
            out.append((tr.out.clone(), loss.clone(), tr.grads.clone(), model.flat_parameters.clone()))
        assert tr.fault_count == 0
        res.append(out)
    for a, b in zip(res[0], res[1]):
        for u, v in zip(a, b):
            assert torch.isfinite(u).all() and torch.equal(u, v)



def test_flat_parameter_mode_equals_per_tensor_adam(pkg, oracle):
    """``torch.optim.Adam(model.optimizer_parameters())``: ONE leaf parameter over the flat vector instead of the 124 named
    tensors (the optimizer line of train.py:348).  Adam is element-wise, so four steps of the reference loop body give the
    SAME BITS as ``torch.optim.Adam(model.parameters())`` on a twin; ``zero_grad`` both ways, accumulation over two backwards,
    ``model.zero_grad()``, and the named parameters' ``.grad`` views are covered; ``state_dict`` keeps the reference's keys."""
    nb, nc, bs = 3, 32, 2
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
    snaps = pkg.wdn_synth.make_snapshots(8, 388, seed=31)
    mf, _ = build(pkg, oracle, nb, nc, seed=71)
    mt, _ = build(pkg, oracle, nb, nc, seed=71)
    keys = list(mf.state_dict().keys())
    of = torch.optim.Adam(mf.optimizer_parameters(), lr=5e-4, weight_decay=6e-6)
    ot = torch.optim.Adam(mt.parameters(), lr=5e-4, weight_decay=6e-6)
    assert len(of.param_groups[0]["params"]) == 1 and list(mf.state_dict().keys()) == keys
    assert len(list(mf.parameters())) == len(list(mt.parameters()))
    rng = np.random.RandomState(5)
    for it in range(4):
        y = pkg.wdn_synth.collate_snapshots(snaps, range(it * bs, (it + 1) * bs)).cuda()
        m = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, rng)).cuda()
        x = y.clone(); x[m] = 0
        for model, opt in ((mf, of), (mt, ot)):
            if it == 2:
                opt.zero_grad(set_to_none=False)
            elif it == 3:
                model.zero_grad()
            else:
                opt.zero_grad()
            loss = torch.nn.functional.mse_loss(model(x, ei)[m], y[m])
            loss.backward()
            if it == 1:                                  # accumulate a second backward on top
                torch.nn.functional.mse_loss(model(x, ei)[m], y[m]).backward()
            opt.step()
        gt = torch.cat([p.grad.reshape(-1) for p in mt.parameters()])
        assert torch.equal(mf.flat_parameter.grad, gt), it
        assert torch.equal(torch.cat([p.grad.reshape(-1) for p in mf.parameters()]), gt), it
        assert torch.equal(mf.flat_parameters, mt.flat_parameters), it
    assert torch.equal(mf.state_dict()["lin1.weight"], mt.state_dict()["lin1.weight"])


def test_inference_notices_parameters_repointed_behind_its_back(pkg, oracle):
    """Under ``torch.no_grad()`` the forward launch goes out BEFORE the module checks that its 124 parameters still are views
    of the flat vector the kernels read (the reference times every inference call, utils/timer.py).  A parameter re-pointed
    (``p.data = ...``) or replaced (``module.weight = nn.Parameter(...)``) since the last call must still be honoured: the flat
    vector is rebuilt and the forward runs again."""
    nb, nc, bs = 2, 32, 2
    x, _, ei, _ = ctown_batch(pkg, bs)
    model, p = build(pkg, oracle, nb, nc, seed=9)
    x, ei = x.cuda(), ei.cuda()
    with torch.no_grad():
        a = model(x, ei).clone()
        model.lin1.bias.data = model.lin1.bias.data + 1.0                    # re-pointed: a new storage
        b = model(x, ei).clone()
        assert float((b - a - 1.0).abs().max()) < 1e-5
        model.lin1.weight = torch.nn.Parameter(torch.zeros_like(model.lin1.weight))      # replaced
        c = model(x, ei)
        assert float((c - float(model.lin1.bias)).abs().max()) < 1e-6
    p2 = dict(p)
    p2["lin1.bias"] = p["lin1.bias"] + 1.0
    p2["lin1.weight"] = torch.zeros_like(p["lin1.weight"])
    assert relerr(c, oracle.gatres_forward(p2, x.cpu(), ei.cpu(), num_blocks=nb)) < 1e-5

"""BASELINE.json configs 3 and 5 on the GPU (fp32 part), the relabelled-plan path, and the round-2 host features:
dropped faulted steps, external parameter edits, learning-rate changes under hipGraph replay, two streams, the epoch loop
over a device SnapshotStore, and the in-graph bucketed all-reduce (RCCL with one rank).

Sizes follow the test strategy of the task: values pinned by the oracle where it finishes in seconds (bs 4 C-Town for
gatres_large, ONE 50 000-node graph), size-independent properties at BASELINE.json's full sizes (bs 128, batch of 2)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_model import build, ctown_batch, note, relerr      # noqa: E402


def _flat_grads(model):
    return torch.cat([q.grad.reshape(-1) for q in model.parameters()])


# ---------------------------------------------------------------------------------------------------- config 3 (fp32)
def test_gatres_large_ctown_bs128_properties(pkg, oracle):
    """gatres_large (25 blocks, nc = 128; ConfigModels.py:22-32) at BASELINE config 3's size (C-Town, bs = 128): the
    properties of test_full_size_properties_bs32, the oracle pinning one graph of the batch."""
    nb, nc, bs = 25, 128, 128
    model, p = build(pkg, oracle, nb, nc, seed=3)
    x, y, ei, mask = ctown_batch(pkg, bs)
    dx, dei = x.cuda(), ei.cuda()
    out = model(dx, dei)
    assert out.shape == (388 * bs, 1) and torch.isfinite(out).all()
    one = pkg.wdn_synth.make_wdn_topology()
    first = model(dx[:388], one.cuda())
    assert torch.equal(out[:388], first)                                        # block-diagonal independence
    last = model(dx[-388:], one.cuda())
    assert torch.equal(out[-388:], last)
    perm = torch.randperm(ei.shape[1], generator=torch.Generator().manual_seed(0))
    assert relerr(model(dx, ei[:, perm].cuda()), out) < 1e-5                    # edge-order invariance
    g1, g2 = torch.randn_like(out), torch.randn_like(out)

    def grad_for(g):
        model.zero_grad()
        model(dx, dei).backward(g)
        return _flat_grads(model).clone()

    ga, gb, gab = grad_for(g1), grad_for(g2), grad_for(g1 + 2 * g2)
    assert torch.isfinite(gab).all() and relerr(gab, ga + 2 * gb) < 1e-4        # backward is linear in the upstream gradient
    assert torch.equal(grad_for(g1), ga)                                        # and bitwise reproducible
    ref = oracle.gatres_forward(p, x[:388], one)
    e = relerr(first, ref)
    note("gatres_large 25x128 bs128: one graph of the batch vs oracle32", e)
    assert e < 1e-5
    # one native training step at full size: finite loss, every parameter moved by at most ~lr
    before = model.flat_parameters.clone()
    tr = pkg.GATResTrainer(model, dei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=False)
    assert not tr.fused                                                         # nc = 128 takes the per-op kernels
    loss = tr.step(dx.reshape(-1), dx.reshape(-1))
    assert torch.isfinite(loss).all() and int(tr.mask.sum()) == bs * 368
    d = (model.flat_parameters - before).abs()
    assert 0 < float(d.max()) <= 5e-4 * 1.01


# ---------------------------------------------------------------------------------------------------- config 5 (fp32)
@pytest.mark.parametrize("nb,nc", [(15, 32), (3, 128)], ids=["gatres_small", "3x128"])
def test_50k_node_graph(pkg, oracle, nb, nc):
    """One 50 000-node / 75 000-pipe WDN (BASELINE config 5's graph): per-op kernels vs the oracle on that single graph
    (forward, loss, gradients of a training step), then a batch of 2: block-diagonal independence and bitwise repeats."""
    n, pipes = 50000, 75000
    one = pkg.wdn_synth.make_wdn_topology(n, pipes, seed=0, max_degree=6)
    assert one.shape == (2, 2 * pipes) and int(torch.bincount(one[1], minlength=n).max()) <= 6
    model, p = build(pkg, oracle, nb, nc, seed=7)
    g = torch.Generator().manual_seed(1)
    y = torch.randn(n, 1, generator=g)
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([n], 0.95, np.random.RandomState(2)))
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xin = y.clone(); xin[mask] = 0
    out_ref = oracle.gatres_forward(leaves, xin, one, num_blocks=nb)
    loss_ref = torch.nn.functional.mse_loss(out_ref[mask], y[mask])
    loss_ref.backward()
    g_ref = torch.cat([v.grad.reshape(-1) for v in leaves.values()])
    tr = pkg.GATResTrainer(model, one.cuda(), n, nodes_per_graph=[n], use_graph=False)
    assert not tr.fused and tr.plan.num_segments == 1                           # > 4096 nodes: per-op path
    tr.forward_backward(y.cuda(), y.cuda(), mask.cuda())
    e_out, e_loss, e_g = relerr(tr.out, out_ref), relerr(tr.loss, loss_ref), relerr(tr.grads, g_ref)
    note(f"50k-node graph {nb}x{nc}: out / loss / flat grad rel err vs oracle32", [e_out, e_loss, e_g])
    assert e_out < 1e-5 and e_loss < 1e-5 and e_g < 1e-4
    # batch of 2 (what one rank of config 5 holds): snapshots do not interact, results repeat bitwise
    ei2 = pkg.wdn_synth.collate_edge_index(one, n, 2).cuda()
    x2 = torch.cat([xin, torch.randn(n, 1, generator=g)]).cuda()
    with torch.no_grad():
        o2 = model(x2, ei2)
        assert torch.equal(o2, model(x2, ei2.clone()))
        assert torch.equal(o2[:n], model(x2[:n], one.cuda()))
    assert relerr(o2[:n], out_ref) < 1e-5
    o = model(x2, ei2)
    gup = torch.randn(o.shape, generator=torch.Generator().manual_seed(3)).cuda()
    o.backward(gup)
    ga = _flat_grads(model).clone()
    model.zero_grad()
    model(x2, ei2).backward(gup)
    assert torch.isfinite(ga).all() and torch.equal(ga, _flat_grads(model))


@pytest.mark.parametrize("bs,parts", [(56, 8), (72, 6), (96, 5)])
def test_batches_of_49_to_96_snapshots_run_in_two_rounds(pkg, oracle, lib, bs, parts, monkeypatch):
    """49 .. 96 C-Town snapshots: the window kernel carries the batch in TWO launches (rounds) of at most 48 segments at the
    parts per snapshot that fit the chip for one round (k_fused_host.hip: fused_split / launch_fused).  Against the resident
    single launch (GATRES_FUSED_NO_ROUNDS=1: whole-segment tables at 4 .. 2 parts) and the per-op kernels: predictions
    bit-identical, loss and gradient to rounding; against the oracle: a training step."""
    nb, nc = 3, 32
    t1 = pkg.wdn_synth.make_wdn_topology(388, 430, seed=0)
    ei = pkg.wdn_synth.collate_edge_index(t1, 388, bs).cuda()
    N = 388 * bs
    snaps = pkg.wdn_synth.make_snapshots(bs, 388, seed=5)
    y = pkg.wdn_synth.collate_snapshots(snaps, range(bs))
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, np.random.RandomState(3)))
    res = {}
    for mode in ("rounds", "resident", "per_op"):
        if mode == "resident":
            monkeypatch.setenv("GATRES_FUSED_NO_ROUNDS", "1")
        model, p = build(pkg, oracle, nb, nc, seed=9, fused=(mode != "per_op"))
        tr = pkg.GATResTrainer(model, ei, N, nodes_per_graph=[388] * bs, use_graph=False, fused=(mode != "per_op"))
        cus = lib.gatres_fused_cus_per_segment(model._cmodel_ref(), tr.plan.ref()) if mode != "per_op" else 0
        if mode == "rounds":
            assert cus == parts and lib.gatres_fused_window_kernel(model._cmodel_ref(), tr.plan.ref()) == 1
        if mode == "resident":
            assert cus * ((bs + 7) // 8 * 8) <= 256
        tr.forward_backward(y.cuda(), y.cuda(), mask.cuda())
        res[mode] = (tr.out.clone(), tr.loss.clone(), tr.grads.clone())
        if mode == "rounds":
            ref = oracle.OracleTrainer(p)
            l_ref, o_ref = ref.step(y.clone(), y, ei.cpu(), mask)
            loss = tr.step(y.cuda(), y.cuda(), mask.cuda())
            assert relerr(tr.out, o_ref) < 1e-5 and relerr(loss, l_ref) < 1e-5
            assert relerr(model.flat_parameters.detach(), ref.flat("params")) < 1e-5
        monkeypatch.delenv("GATRES_FUSED_NO_ROUNDS", raising=False)
    for other in ("resident", "per_op"):
        assert torch.equal(res["rounds"][0], res[other][0]), other
        assert relerr(res["rounds"][1], res[other][1]) < 1e-6 and relerr(res["rounds"][2], res[other][2]) < 1e-5, other


@pytest.mark.parametrize("bs", [56, 72, 96])
def test_module_paths_with_two_round_batches(pkg, bs):
    """The drop-in nn.Module on batches that go in two rounds: forward-only launches (eval: the whole-segment-table kernel, also
    round by round) and autograd's separate forward / backward launches against the per-op kernels -- predictions bit for bit,
    gradients to rounding."""
    t1 = pkg.wdn_synth.make_wdn_topology(388, 430, seed=0)
    ei = pkg.wdn_synth.collate_edge_index(t1, 388, bs).cuda()
    x = torch.randn(388 * bs, 1, generator=torch.Generator().manual_seed(2)).cuda()
    res = {}
    for fused in (True, False):
        torch.manual_seed(1)
        m = pkg.GATResMeanConv(name="gatres_small", num_blocks=3, nc=32, fused=fused).cuda()
        m.eval()
        with torch.no_grad():
            oe = m(x, ei, None, None).clone()
        m.train()
        o = m(x, ei, None, None)
        o.square().mean().backward()
        res[fused] = (oe, o.detach().clone(), _flat_grads(m))
    a, b = res[True], res[False]
    assert torch.equal(a[0], b[0]) and torch.equal(a[0], a[1]) and torch.equal(a[1], b[1])
    assert relerr(a[2], b[2]) < 1e-5


def test_config5_real_size_25x128_on_two_50k_node_graphs(pkg, oracle, lib):
    """BASELINE config 5 as ONE RANK holds it: gatres_large (25 x 128) on a batch of 2 x 50 000-node / 75 000-pipe graphs,
    per-op kernels, fp32.  The saved activations are 7.9 GB: byte offsets pass 2^32 inside the second graph's half of
    every table -- where a 32-bit offset breaks silently.  Checked: (1) block-diagonal independence and bitwise repeats
    of the predictions; (2) the batch's gradient equals the masked-count-weighted mean of the two single-graph
    gradients (each of which runs below the 4-GB line: the combination pins the upper half of every table);
    (3) the oracle, 25 blocks on one graph: predictions, loss and the full gradient."""
    nb, nc, n, pipes = 25, 128, 50000, 75000
    one = pkg.wdn_synth.make_wdn_topology(n, pipes, seed=0, max_degree=6)
    model, p = build(pkg, oracle, nb, nc, seed=7)
    gen = torch.Generator().manual_seed(1)
    ys = [torch.randn(n, 1, generator=gen) for _ in range(2)]
    rng = np.random.RandomState(2)
    masks = [torch.from_numpy(pkg.wdn_synth.generate_batch_mask([n], 0.95, rng)) for _ in range(2)]
    ei2 = pkg.wdn_synth.collate_edge_index(one, n, 2).cuda()
    y2, m2 = torch.cat(ys).cuda(), torch.cat(masks).cuda()
    tr2 = pkg.GATResTrainer(model, ei2, 2 * n, nodes_per_graph=[n, n], use_graph=False)
    assert not tr2.fused
    saved_floats = int(lib.gatres_saved_floats(model._cmodel_ref(), tr2.plan.ref()))
    assert saved_floats * 4 > 2 ** 32 and saved_floats > 1.9e9          # byte offsets beyond 32 bits are what runs here
    tr2.forward_backward(y2, y2, m2)
    out2, g2, l2 = tr2.out.clone(), tr2.grads.clone(), tr2.loss.clone()
    assert torch.isfinite(out2).all() and torch.isfinite(g2).all()
    tr2.forward_backward(y2, y2, m2)
    assert torch.equal(tr2.out, out2) and torch.equal(tr2.grads, g2)                          # bitwise repeatable
    del tr2
    torch.cuda.empty_cache()
    singles = []
    for k in range(2):
        tr1 = pkg.GATResTrainer(model, one.cuda(), n, nodes_per_graph=[n], use_graph=False)
        tr1.forward_backward(ys[k].cuda(), ys[k].cuda(), masks[k].cuda())
        assert torch.equal(tr1.out, out2[k * n:(k + 1) * n]), k                               # snapshots do not interact
        singles.append((tr1.grads.clone(), tr1.loss.clone(), int(masks[k].sum())))
        del tr1
        torch.cuda.empty_cache()
    (ga, la, ka), (gb, lb_, kb) = singles
    comb = (ga * ka + gb * kb) / (ka + kb)
    e_comb = relerr(g2, comb)
    note("config 5 (25x128, 2 x 50k nodes): batch gradient vs weighted mean of the single-graph gradients", e_comb)
    assert e_comb < 2e-5 and relerr(l2, (la * ka + lb_ * kb) / (ka + kb)) < 1e-5
    leaves = {k: v.clone().requires_grad_(True) for k, v in p.items()}
    xin = ys[1].clone(); xin[masks[1]] = 0
    out_ref = oracle.gatres_forward(leaves, xin, one, num_blocks=nb)
    loss_ref = torch.nn.functional.mse_loss(out_ref[masks[1]], ys[1][masks[1]])
    loss_ref.backward()
    g_ref = torch.cat([v.grad.reshape(-1) for v in leaves.values()])
    e_out, e_loss, e_g = relerr(out2[n:], out_ref), relerr(lb_, loss_ref), relerr(gb, g_ref)
    note("config 5 (25x128, 50k nodes): out / loss / flat grad of the batch's SECOND graph vs oracle32", [e_out, e_loss, e_g])
    assert e_out < 1e-5 and e_loss < 1e-5 and e_g < 1e-4


# ---------------------------------------------------------------------------------------------------- relabelled plans
@pytest.mark.parametrize("fused", [True, False], ids=["fused", "per_op"])
def test_shuffled_node_order_is_relabelled_and_exact(pkg, oracle, fused):
    """A C-Town-sized batch whose node ids carry no locality (random relabelling inside every snapshot).  The plan adopts
    the reverse Cuthill-McKee order, the fused launch takes the window kernel at 4 CUs per snapshot again, and x / mask /
    out / gradients keep the CALLER's node order: training steps match the oracle run on the shuffled graph itself, and
    the predictions are bit-identical to the per-op kernels and to the same model on the un-shuffled graph."""
    nb, nc, bs = 15, 32, 3
    t1 = pkg.wdn_synth.make_wdn_topology(388, 430, seed=0)
    rs = np.random.RandomState(11)
    sigma = torch.from_numpy(np.concatenate([rs.permutation(388) + 388 * k for k in range(bs)]))      # old id -> shuffled id
    ei0 = pkg.wdn_synth.collate_edge_index(t1, 388, bs)
    ei = sigma[ei0]
    N = 388 * bs
    model, p = build(pkg, oracle, nb, nc, seed=3, fused=fused)
    tr = pkg.GATResTrainer(model, ei.cuda(), N, nodes_per_graph=[388] * bs, use_graph=False, fused=fused)
    lib = pkg._native.load()
    if fused:
        assert tr.plan.perm_host is not None and tr.plan.window_rows(4) < 388 // 2
        assert lib.gatres_fused_cus_per_segment(model._cmodel_ref(), tr.plan.ref()) == 8      # 8 parts (+ 2 consumer CUs: a batch of 3 leaves CUs free)
        assert lib.gatres_fused_window_kernel(model._cmodel_ref(), tr.plan.ref()) == 1
    snaps = pkg.wdn_synth.make_snapshots(2 * bs, 388, seed=4)
    ref = oracle.OracleTrainer(p)
    rng = np.random.RandomState(6)
    for it in range(2):
        y = pkg.wdn_synth.collate_snapshots(snaps, range(it * bs, (it + 1) * bs))
        mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, rng))
        l_ref, o_ref = ref.step(y.clone(), y, ei, mask)
        loss = tr.step(y.cuda(), y.cuda(), mask.cuda())
        assert relerr(tr.out, o_ref) < 1e-5 and relerr(loss, l_ref) < 1e-5, it
        if it == 0:
            e = relerr(tr.grads, ref.flat("grads"))
            note(f"shuffled C-Town batch ({'fused' if fused else 'per-op'}): flat grad rel err vs oracle32", e)
            assert e < 1e-4
    # module surface: same model, shuffled vs original labelling -> the same numbers at the corresponding nodes, bitwise
    m2, _ = build(pkg, oracle, nb, nc, seed=3, fused=fused)
    x0 = torch.randn(N, 1, generator=torch.Generator().manual_seed(9))
    xs = torch.empty_like(x0); xs[sigma] = x0                                    # x of shuffled node sigma[i] = x0[i]
    o_orig = m2(x0.cuda(), ei0.cuda())
    o_shuf = m2(xs.cuda(), ei.cuda())
    assert torch.equal(o_shuf[sigma.cuda()], o_orig)
    o_shuf.backward(torch.ones_like(o_shuf))
    gs = _flat_grads(m2).clone()
    m2.zero_grad()
    m2(x0.cuda(), ei0.cuda()).backward(torch.ones_like(o_orig))
    assert relerr(gs, _flat_grads(m2)) < 2e-5                                    # (slab partition differs, values do not)


# ---------------------------------------------------------------------------------------------------- host features
def _small_setup(pkg, oracle, nb=3, nc=32, bs=2, seed=31, **kw):
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
    y = pkg.wdn_synth.collate_snapshots(pkg.wdn_synth.make_snapshots(4, 388, seed=2), range(bs)).cuda()
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, np.random.RandomState(5))).cuda()
    model, p = build(pkg, oracle, nb, nc, seed=seed)
    tr = pkg.GATResTrainer(model, ei, 388 * bs, nodes_per_graph=[388] * bs, **kw)
    return model, p, tr, ei, y, mask


@pytest.mark.parametrize("use_graph", [False, True])
def test_transient_fault_drops_one_step_only(pkg, oracle, use_graph):
    """A split launch that gives up waiting for a partner sets the status word.  That step must be DROPPED (loss NaN,
    parameters / moments / step count untouched), the word cleared and counted, and the next step must be exactly the
    step a healthy run would have taken -- no manual reset."""
    model, p, tr, ei, y, mask = _small_setup(pkg, oracle, use_graph=use_graph)
    if tr._status is None or pkg._native.load().gatres_fused_cus_per_segment(model._cmodel_ref(), tr.plan.ref()) < 2:
        pytest.skip("snapshots are not split on this device")
    for _ in range(3):                                       # (both graphs -- transposes skipped / not -- are captured now)
        tr.step(y, y, mask)
    before = (model.flat_parameters.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), tr.optimizer_step)
    tr._status[0] = 1                                        # what a timed-out poller writes
    loss = tr.step(y, y, mask)
    assert torch.isnan(loss).all()
    assert torch.equal(model.flat_parameters, before[0]) and torch.equal(tr.exp_avg, before[1])
    assert torch.equal(tr.exp_avg_sq, before[2]) and tr.optimizer_step == before[3]
    assert int(tr._status[0]) == 0 and tr.fault_count == 1
    with pytest.raises(RuntimeError, match="DROPPED"):       # (round 5: drops are an error for whoever asks -- fit_epoch, bench.py)
        tr.check_no_dropped_steps()
    tr.check_no_dropped_steps()                              # ... reported once
    loss2 = tr.step(y, y, mask)
    twin, _, tr2, *_ = _small_setup(pkg, oracle, use_graph=False)
    for _ in range(3):
        tr2.step(y, y, mask)
    l2 = tr2.step(y, y, mask)
    assert torch.equal(loss2, l2) and torch.equal(model.flat_parameters, twin.flat_parameters)
    # forward-only launches clear the word too (and poison their output once)
    tr._status[0] = 1
    with torch.no_grad():
        o = model(y.reshape(-1, 1), ei)
    assert torch.isnan(o).any() and int(tr._status[0]) == 0 and tr.fault_count == 2
    with torch.no_grad():
        assert torch.isfinite(model(y.reshape(-1, 1), ei)).all()


@pytest.mark.parametrize("use_graph", [False, True])
def test_load_state_dict_between_steps_is_noticed(pkg, oracle, use_graph):
    """every nn.Parameter has its own version counter, so load_state_dict / torch.optim steps never
    bump the flat vector's.  The trainer's "transposed weights still valid" promise must see them anyway."""
    model, p, tr, ei, y, mask = _small_setup(pkg, oracle, use_graph=use_graph)
    for _ in range(3):
        tr.step(y, y, mask)                                   # transposes are skipped from the second step on
    fresh, _ = build(pkg, oracle, 3, 32, seed=77)
    model.load_state_dict(fresh.state_dict())                 # per-parameter copy_: flat._version does not move
    tr.forward_backward(y, y, mask)
    g_fb = tr.grads.clone()
    tr.step(y, y, mask)
    tr2 = pkg.GATResTrainer(fresh, ei, 388 * 2, nodes_per_graph=[388] * 2, use_graph=False)
    tr2.forward_backward(y, y, mask)
    assert relerr(g_fb, tr2.grads) < 1e-6 and relerr(tr.grads, tr2.grads) < 1e-6
    # an optimizer that writes through the parameters (torch.optim.SGD here) between two native steps
    opt = torch.optim.SGD(model.parameters(), lr=0.05)
    for q in model.parameters():
        q.grad = torch.ones_like(q)
    opt.step()
    tr.forward_backward(y, y, mask)
    twin, _ = build(pkg, oracle, 3, 32, seed=1)
    with torch.no_grad():
        twin.flat_parameters.copy_(model.flat_parameters)
    tr3 = pkg.GATResTrainer(twin, ei, 388 * 2, nodes_per_graph=[388] * 2, use_graph=False)
    tr3.forward_backward(y, y, mask)
    assert relerr(tr.grads, tr3.grads) < 1e-6


def test_set_lr_under_graph_replay(pkg, oracle):
    """The update kernels read the hyper-parameters from a device buffer (gatres_train_step_t.hparams): a
    ReduceLROnPlateau-style change (train.py:349-350) takes effect at the next step WITHOUT a new capture -- one family of
    graphs for three learning rates (small batch: the two-launch update with consumer workgroups; the headline shape is in
    tests/test_gpu_headline.py)."""
    model, p, tr, ei, y, mask = _small_setup(pkg, oracle, use_graph=True)
    twin, _, tr2, *_ = _small_setup(pkg, oracle, use_graph=False)
    counts = []
    for lr in (5e-4, 5e-4, 1e-4, 1e-4, 2e-5, 5e-4):
        tr.set_lr(lr); tr2.set_lr(lr)
        tr.step(y, y, mask); tr2.step(y, y, mask)
        assert torch.equal(model.flat_parameters, twin.flat_parameters), lr
        counts.append(tr.num_captured_graphs)
    assert counts[-1] == counts[1] <= 2, counts               # nothing captured after the first two steps


@pytest.mark.parametrize("use_graph", [False, True])
def test_staged_batch_with_device_mask_equals_copy_then_sampler(pkg, oracle, use_graph, monkeypatch):
    """``step(x, y)`` with a device-resident batch stages it inside the mask sampler's launch (gatres_stage_batch_mask); the
    mask stream, the loss and the weights must be those of ``load_batch`` + the sampler inside the step, bit for bit."""
    model, p, tr, ei, y, _ = _small_setup(pkg, oracle, use_graph=use_graph)
    monkeypatch.setenv("GATRES_NO_STAGE_MASK", "1")
    twin, _, tr2, *_ = _small_setup(pkg, oracle, use_graph=use_graph)
    xs = [torch.randn_like(y) for _ in range(3)]
    for x in xs:
        monkeypatch.delenv("GATRES_NO_STAGE_MASK", raising=False)
        l1 = tr.step(x, x).clone()
        monkeypatch.setenv("GATRES_NO_STAGE_MASK", "1")
        l2 = tr2.step(x, x).clone()
        assert torch.equal(tr.mask, tr2.mask) and int(tr.mask.sum()) > 0
        assert torch.equal(tr.x, tr2.x) and torch.equal(l1, l2)
        assert torch.equal(model.flat_parameters, twin.flat_parameters)


def test_training_and_evaluation_on_two_streams(pkg, oracle):
    """Split launches need their whole grid resident, so two of them must never overlap on one device.  Training on one
    stream and evaluation on another (both split) must serialise behind each other: finite, and equal to running them
    one after the other on one stream."""
    nb, nc, bs = 4, 32, 4
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
    snaps = pkg.wdn_synth.make_snapshots(6 * bs, 388, seed=21).cuda()
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, np.random.RandomState(1))).cuda()
    res = []
    for concurrent in (False, True):
        mt, _ = build(pkg, oracle, nb, nc, seed=5)
        me, _ = build(pkg, oracle, nb, nc, seed=6)
        tr = pkg.GATResTrainer(mt, ei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=True)
        s_train, s_eval = (torch.cuda.Stream(), torch.cuda.Stream()) if concurrent else (torch.cuda.current_stream(),) * 2
        torch.cuda.synchronize()
        outs = []
        for it in range(6):
            yb = snaps[it * bs:(it + 1) * bs].reshape(-1)
            with torch.cuda.stream(s_train):
                tr.step(yb, yb, mask)
            with torch.cuda.stream(s_eval), torch.no_grad():
                outs.append(me(yb.reshape(-1, 1), ei))
        torch.cuda.synchronize()
        assert all(torch.isfinite(o).all() for o in outs) and torch.isfinite(tr.loss).all() and tr.fault_count == 0
        res.append((mt.flat_parameters.clone(), torch.stack(outs)))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_snapshot_store_collation_on_device_and_fit_epoch(pkg, oracle):
    """SURVEY 8(f) rank 2 on the GPU: the store's batches equal the reference collation (device = cuda), and
    GATResTrainer.fit_epoch (train.py:159-198: shuffled batches, ragged last batch, loss weighted by graph count) equals
    the same loop written out by hand with the device mask sampler."""
    one = pkg.wdn_synth.make_wdn_topology()
    raw = pkg.wdn_synth.make_snapshots(11, 388, seed=3) * 6 + 35
    st = pkg.SnapshotStore(raw, one, device="cuda")
    mean, std = float(raw.mean()), float(raw.std(unbiased=False))
    rows = torch.tensor([3, 0, 7])
    x = st.batch(rows)
    assert x.is_cuda and x.shape == (3 * 388, 1)
    assert torch.allclose(x.cpu(), ((raw[rows] - mean) / (std + 1e-8)).reshape(-1, 1), atol=1e-5)
    assert torch.equal(st.edge_index(3).cpu(), pkg.wdn_synth.collate_edge_index(one, 388, 3))
    nb, nc, bs = 3, 32, 4
    model, _ = build(pkg, oracle, nb, nc, seed=8)
    tr = pkg.GATResTrainer(model, st.edge_index(bs), 388 * bs, nodes_per_graph=[388] * bs, seed=5, targets_are_inputs=True)
    fns = pkg.evaluation.get_metric_fn_collection("train")
    loss, metrics = tr.fit_epoch(st, bs, shuffle=True, generator=torch.Generator().manual_seed(7), metric_fn_dict=fns)
    assert tr.optimizer_step == 3 and set(metrics) == set(fns) and np.isfinite(loss)      # 4 + 4 + 3 graphs
    # by hand: same order, same per-batch trainers (a sibling for the ragged batch shares the optimizer state)
    twin, _ = build(pkg, oracle, nb, nc, seed=8)
    t4 = pkg.GATResTrainer(twin, st.edge_index(bs), 388 * bs, nodes_per_graph=[388] * bs, seed=5, targets_are_inputs=True)
    t3 = pkg.GATResTrainer(twin, st.edge_index(3), 388 * 3, nodes_per_graph=[388] * 3, seed=5, targets_are_inputs=True,
                           _share_state_with=t4)
    tot, n = 0.0, 0
    for xb, eib, ng in st.batches(bs, shuffle=True, generator=torch.Generator().manual_seed(7)):
        t = t4 if ng == bs else t3
        tot += float(t.step(xb, xb)) * ng
        n += ng
    assert n == 11 and abs(loss - tot / n) < 1e-6 * abs(tot / n)
    assert torch.equal(model.flat_parameters, twin.flat_parameters)


def test_fit_epoch_sequences_equal_single_steps(pkg, oracle):
    """Round 5: fit_epoch trains an epoch's full batches IN PLACE on a shuffled device copy of the store (one gather per epoch),
    `epoch_graph_steps` per captured launch sequence with the mask sampled ahead by the update launches
    (``steps_bound(bound=..., losses=...)``); a store beyond ``epoch_copy_limit_bytes`` goes through ``steps_rows`` instead
    (k x [collation + mask sampler, the step's three launches] per sequence).  Two epochs over 23 snapshots in batches of 2 --
    11 full batches (the first by itself: its transposed weights are not current; sequences of 4; what is left by itself or as
    a shorter sequence) and a ragged batch of 1 on the sibling trainer; the second epoch replays the captured sequences -- must
    equal, bit for bit, the same epochs stepped one batch at a time through the collation launch: mean losses, masks,
    parameters, moments, step count."""
    one = pkg.wdn_synth.make_wdn_topology()
    raw = pkg.wdn_synth.make_snapshots(23, 388, seed=13) * 6 + 35
    st = pkg.SnapshotStore(raw, one, device="cuda")
    st2 = pkg.SnapshotStore(pkg.wdn_synth.make_snapshots(17, 388, seed=14) * 6 + 35, one, device="cuda")
    nb, nc, bs = 3, 32, 2
    res = []
    for kseq, limit, kind in ((4, None, "seq"), (4, 0, "rowseq"), (1, 0, None)):
        model, _ = build(pkg, oracle, nb, nc, seed=18)
        tr = pkg.GATResTrainer(model, st.edge_index(bs), 388 * bs, nodes_per_graph=[388] * bs, seed=9, targets_are_inputs=True)
        tr.epoch_graph_steps = kseq
        if limit is not None:
            tr.epoch_copy_limit_bytes = limit
        losses = [tr.fit_epoch(st, bs, shuffle=True, generator=torch.Generator().manual_seed(31 + ep))[0] for ep in range(2)]
        assert tr.optimizer_step == 24 and tr.dropped_steps == 0
        # a third epoch over a store of another size: the epoch buffer is re-allocated, no graph of the old one may be replayed
        losses.append(tr.fit_epoch(st2, bs, shuffle=True, generator=torch.Generator().manual_seed(40))[0])
        assert tr.optimizer_step == 24 + 9
        taken = {k[0] for k in tr._graphs if isinstance(k[0], str)}
        assert kind is None or kind in taken, (kind, taken)
        assert kind == "seq" or "seq" not in taken
        res.append((losses, model.flat_parameters.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone(), tr.mask.clone()))
    for other in res[1:]:
        assert res[0][0] == other[0], (res[0][0], other[0])
        for a, b in zip(res[0][1:], other[1:]):
            assert torch.equal(a, b)


@pytest.mark.parametrize("use_graph", [False, True])
def test_fit_epoch_two_epochs_with_a_ragged_tail_keeps_transposed_weights_current(pkg, oracle, use_graph):
    """every trainer keeps the transposed conv weights of the fused backward in its OWN scratch buffer
    and skips their re-derivation while "nothing changed the parameters since my last step" -- which a sibling trainer's
    step (the ragged last batch of fit_epoch) does.  Two epochs of 4 + 4 + 3 graphs must equal, bit for bit, the same
    loop on a twin whose trainers are told to re-derive the transposes before EVERY step."""
    one = pkg.wdn_synth.make_wdn_topology()
    raw = pkg.wdn_synth.make_snapshots(11, 388, seed=3) * 6 + 35
    st = pkg.SnapshotStore(raw, one, device="cuda")
    nb, nc, bs = 3, 32, 4
    model, _ = build(pkg, oracle, nb, nc, seed=8)
    tr = pkg.GATResTrainer(model, st.edge_index(bs), 388 * bs, nodes_per_graph=[388] * bs, seed=5, targets_are_inputs=True,
                           use_graph=use_graph)
    losses = [tr.fit_epoch(st, bs, shuffle=True, generator=torch.Generator().manual_seed(7 + ep))[0] for ep in range(2)]
    assert tr.optimizer_step == 6
    twin, _ = build(pkg, oracle, nb, nc, seed=8)
    t4 = pkg.GATResTrainer(twin, st.edge_index(bs), 388 * bs, nodes_per_graph=[388] * bs, seed=5, targets_are_inputs=True,
                           use_graph=False)
    t3 = pkg.GATResTrainer(twin, st.edge_index(3), 388 * 3, nodes_per_graph=[388] * 3, seed=5, targets_are_inputs=True,
                           use_graph=False, _share_state_with=t4)
    for ep in range(2):
        tot, n = 0.0, 0
        for xb, eib, ng in st.batches(bs, shuffle=True, generator=torch.Generator().manual_seed(7 + ep)):
            t = t4 if ng == bs else t3
            t.invalidate_weights()
            tot += float(t.step(xb, xb)) * ng
            n += ng
        assert abs(losses[ep] - tot / n) < 1e-6 * abs(tot / n), ep
    assert torch.equal(model.flat_parameters, twin.flat_parameters)


def test_bucketed_allreduce_in_graph_rccl_one_rank(pkg, oracle):
    """The multi-rank step (backward pieces | bucketed RCCL all-reduce | Adam) captured into ONE hipGraph, exercised on
    a single GPU with a one-rank nccl group: per-op path with one bucket per block (gatres_large's scheme) and the fused
    path (one bucket) must both reproduce the plain step bit for bit."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ["MASTER_PORT"] = str(29600 + os.getpid() % 300)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        bs = 2
        ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
        snaps = pkg.wdn_synth.make_snapshots(4 * bs, 388, seed=8).cuda()
        for nb, nc, fused in ((4, 128, False), (5, 32, True)):
            res = []
            for split in (False, True):
                model, _ = build(pkg, oracle, nb, nc, seed=2, fused=fused)
                tr = pkg.GATResTrainer(model, ei, 388 * bs, nodes_per_graph=[388] * bs, seed=77, fused=fused,
                                       force_collective_path=split, blocks_per_bucket=1, use_graph=True)
                assert tr.split == split and tr.fused == fused
                if split:
                    assert tr.reducer.active
                losses = []
                for it in range(4):
                    yb = snaps[it * bs:(it + 1) * bs].reshape(-1)
                    losses.append(float(tr.step(yb, yb)))
                if split:       # the whole step is one captured graph (fused: one before / one after scratch's W^T became current)
                    assert len(tr._graphs) == (2 if fused else 1)
                res.append((losses, model.flat_parameters.clone()))
            assert res[0][0] == res[1][0] and torch.equal(res[0][1], res[1][1]), (nb, nc, fused)
    finally:
        dist.destroy_process_group()

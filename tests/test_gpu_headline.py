"""The benchmarked path itself, pinned: BASELINE config 2 -- gatres_small (15 x 32), C-Town-sized
snapshots, batch_size 32, fp32 -- through ``GATResTrainer.step`` with hipGraph replay, 8 CUs per snapshot and the folded
parameter-gradient / update launch, against the oracle's training loop (train.py:159-190) under identical host masks.
Also here: the pieces of the step that round 4 changed -- hyper-parameters in a device buffer, the one-launch parameter
gradients + slab sum + Adam against the two-launch form, the fused path's two gradient buckets, counted dropped steps.

The checker is the oracle (parity unpinned: never held to PyG output)."""
import ctypes as C

import numpy as np
import pytest
import torch

from test_gpu_model import build, note, relerr

pytestmark = pytest.mark.gpu
NB, NC, BS, NODES, PIPES = 15, 32, 32, 388, 430


def _trainer(pkg, oracle, seed=3, bs=BS, **kw):
    model, p = build(pkg, oracle, NB, NC, seed=seed)
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(NODES, PIPES, seed=0), NODES, bs)
    tr = pkg.GATResTrainer(model, ei.cuda(), NODES * bs, nodes_per_graph=[NODES] * bs, **kw)
    return model, p, tr, ei


def test_headline_config_training_steps_vs_oracle(pkg, oracle, lib):
    model, p, tr, ei = _trainer(pkg, oracle, use_graph=True)
    # the configuration the bench line reports: window kernel at 8 CUs per snapshot, the parameter gradients as a launch of
    # their own (no consumer workgroups at bs 32) that also carries the slab sum and Adam
    assert tr.fused and tr.use_graph
    assert lib.gatres_fused_cus_per_segment(model._cmodel_ref(), tr.plan.ref()) == 8
    assert lib.gatres_fused_window_kernel(model._cmodel_ref(), tr.plan.ref()) == 1
    assert lib.gatres_fused_finish_folds(model._cmodel_ref(), tr.plan.ref()) == 1
    ref = oracle.OracleTrainer(p)
    snaps = pkg.wdn_synth.make_snapshots(3 * BS, NODES, seed=6)
    rng = np.random.RandomState(1)
    errs = []
    for it in range(3):
        y = pkg.wdn_synth.collate_snapshots(snaps, range(it * BS, (it + 1) * BS))
        mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([NODES] * BS, 0.95, rng))
        if it == 0:
            # fp64 arbiter for the step-0 gradient (as test_forward_backward_parity): a tensor may be off the fp64 gradient by
            # twice what the fp32 oracle itself is off, or by the noise floor of its own / the whole gradient's scale
            xin = y.double().clone()
            xin[mask] = 0

            def grads64(params64):
                l = {k: v.clone().requires_grad_(True) for k, v in params64.items()}
                o = oracle.gatres_forward(l, xin, ei, num_blocks=NB)
                torch.nn.functional.mse_loss(o[mask], y.double()[mask]).backward()
                return {k: v.grad for k, v in l.items()}

            p64 = {k: v.double() for k, v in p.items()}
            g64 = grads64(p64)
        l_ref, o_ref = ref.step(y.clone(), y, ei, mask)
        loss = tr.step(y.cuda(), y.cuda(), mask.cuda())
        e_out, e_loss = relerr(tr.out, o_ref), relerr(loss, l_ref)
        errs.append((e_out, e_loss))
        assert e_out < 1e-5 and e_loss < 1e-5, (it, e_out, e_loss)
        if it == 0:
            g_hip = tr.grads.detach().cpu().double()
            e_flat = relerr(tr.grads, ref.flat("grads"))
            assert e_flat < 1e-4, e_flat
            def judge(gref):
                gscale = max(float(v.abs().max()) for v in gref.values())
                off, bad, worst = 0, [], 0.0
                for k, v in ref.params.items():
                    n = v.numel()
                    e_hip = float((g_hip[off:off + n] - gref[k].reshape(-1)).abs().max())
                    e_ref = float((v.grad.double().reshape(-1) - gref[k].reshape(-1)).abs().max())
                    floor = max(1e-5 * float(gref[k].abs().max()), 2e-6 * gscale)
                    if e_hip > max(2 * e_ref, floor):
                        bad.append((k, e_hip, e_ref, floor))
                    worst = max(worst, e_hip / gscale)
                    off += n
                return bad, worst

            bad, worst = judge(g64)
            branch, bad0 = 0, bad
            # The network is piecewise linear: a pre-activation within fp32 round-off of a ReLU / LeakyReLU kink makes the
            # gradient BIMODAL, and the kernels (fused and per-op alike: tests/micro/grad_err_probe.py) may stand on the other
            # side of it than the fp32 oracle does -- at this very point one conv1 output of block 8 does, and row 12 of its
            # weight gradient differs by 5.6e-5 (4.3e-5 allowed).  Such a tensor is excused ONLY by a demonstrated kink: an
            # fp64 gradient at parameters perturbed by 1e-7 (relative) that (i) jumps on that tensor by more than the
            # tolerance floor against the unperturbed fp64 gradient and (ii) agrees with the kernels' gradient within the
            # usual tolerance on EVERY tensor.
            gen = torch.Generator().manual_seed(1234)
            while bad and branch < 12:
                branch += 1
                pert = {k: v * (1 + 1e-7 * torch.randn(v.shape, generator=gen, dtype=torch.float64)) for k, v in p64.items()}
                gb = grads64(pert)
                bad, worst_b = judge(gb)
                if not bad:
                    for k, _, _, floor in bad0:
                        jump = float((gb[k] - g64[k]).abs().max())
                        assert jump > floor, (k, jump, floor, "the excuse must be a kink the fp64 oracle itself shows")
                    worst = worst_b
            assert not bad, (bad[:3], bad0[:3], branch)
            note("headline config (15x32, bs 32, hipGraph, 8 CUs/snapshot, parameter gradients as their own launch) step 0: flat grad vs oracle32 / "
                 "worst tensor error over |g|max vs fp64 / fp64 kink branch used (0 = none)", [e_flat, worst, branch])
    assert tr.optimizer_step == 3 and tr.fault_count == 0
    assert tr.num_captured_graphs <= 2          # (one before / one after scratch's transposed weights became current)
    diff = (model.flat_parameters.detach().cpu().double() - ref.flat("params").double()).abs()
    note("headline config, 3 steps: worst out / loss error vs oracle, max |dp|, fraction of params off by > 1e-5",
         [max(e[0] for e in errs), max(e[1] for e in errs), float(diff.max()), float((diff > 1e-5).double().mean())])
    assert float(diff.max()) <= 3 * 5e-4 * 1.01 and float((diff > 1e-5).double().mean()) < 0.01


def test_set_lr_follows_a_schedule_without_recapture(pkg, oracle):
    """ReduceLROnPlateau (train.py:349-350,510): three learning rates, ONE family of captured graphs -- the update kernels
    read the hyper-parameters from the trainer's device buffer -- and every step bit-identical to eager launches."""
    model, p, tr, ei = _trainer(pkg, oracle, use_graph=True)
    twin, _, tr2, _ = _trainer(pkg, oracle, use_graph=False)
    y = pkg.wdn_synth.make_snapshots(BS, NODES, seed=9).reshape(-1).cuda()
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([NODES] * BS, 0.95, np.random.RandomState(3))).cuda()
    seen = []
    for lr in (5e-4, 5e-4, 1e-4, 2.5e-5, 2.5e-5, 5e-4):
        tr.set_lr(lr); tr2.set_lr(lr)
        tr.step(y, y, mask); tr2.step(y, y, mask)
        assert torch.equal(model.flat_parameters, twin.flat_parameters), lr
        seen.append(tr.num_captured_graphs)
    assert seen[-1] == seen[1] <= 2, seen                      # nothing was captured for the later learning rates
    # ... and the value really is used: the same step at lr / 10 moves the parameters a tenth as far
    before = model.flat_parameters.clone()
    tr.set_lr(5e-5); tr.step(y, y, mask)
    d_small = (model.flat_parameters - before).abs().max()
    model.flat_parameters.copy_(before); tr.invalidate_weights()
    assert float(d_small) < 5e-5 * 1.01 * 3


def test_folded_update_equals_the_two_launch_form(pkg, oracle, lib):
    """param_grads_finish_kernel (parameter gradients + slab sum + Adam in one launch, the column's last workgroup doing the
    sum) against gatres_fused_param_grads + gatres_fused_finish: gradients, parameters, moments and the transposed weights
    bit for bit, three steps in a row, and again range by range without Adam."""
    from gnn_pressure_estimation_amd import train_step as TS
    y = pkg.wdn_synth.make_snapshots(BS, NODES, seed=11).reshape(-1).cuda()
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([NODES] * BS, 0.95, np.random.RandomState(5))).cuda()
    runs = []
    for folded in (True, False):
        model, p, tr, ei = _trainer(pkg, oracle, use_graph=False)
        st = pkg._native.current_stream(tr.device)
        m, g = model._cmodel_ref(), tr.plan.ref(model._cmodel_ref())
        h = tr.hparams
        hist = []
        for it in range(3):
            tr.load_batch(y, y, mask)
            tr._enqueue(TS.PHASE_FORWARD | TS.PHASE_BACKWARD, False, flags=TS.FLAG_GRADS_DEFERRED)
            tail = (1, model.flat_parameters.data_ptr(), tr.exp_avg.data_ptr(), tr.exp_avg_sq.data_ptr(),
                    tr.step_counter.data_ptr(), h["lr"], h["beta1"], h["beta2"], h["eps"], h["weight_decay"])
            if folded:
                pkg._native.check(lib.gatres_fused_param_grads_finish(m, g, tr.saved.data_ptr(), tr.scratch.data_ptr(),
                                                                      tr.grads.data_ptr(), None, None, *tail, None, 1.0, 0, NB,
                                                                      st), "folded")
            else:
                pkg._native.check(lib.gatres_fused_param_grads(m, g, tr.saved.data_ptr(), tr.scratch.data_ptr(), st), "pg")
                pkg._native.check(lib.gatres_fused_finish(m, g, tr.scratch.data_ptr(), tr.grads.data_ptr(), None, None, *tail,
                                                          1.0, st), "finish")
            torch.cuda.synchronize()
            hist.append((tr.grads.clone(), model.flat_parameters.clone(), tr.exp_avg.clone(), tr.exp_avg_sq.clone()))
        assert tr.optimizer_step == 3
        runs.append(hist)
    for it in range(3):
        for a, b, what in zip(runs[0][it], runs[1][it], ("grads", "params", "exp_avg", "exp_avg_sq")):
            assert torch.equal(a, b), (it, what, float((a - b).abs().max()))
    # range by range (the data-parallel step's two buckets), no Adam: the same gradient
    model, p, tr, ei = _trainer(pkg, oracle, use_graph=False)
    tr.load_batch(y, y, mask)
    tr._enqueue(TS.PHASE_FORWARD | TS.PHASE_BACKWARD, False, flags=TS.FLAG_GRADS_DEFERRED)
    tr.grads.fill_(float("nan"))
    tr._enqueue(TS.PHASE_BACKWARD, False, flags=TS.FLAG_GRADS_ONLY, block_lo=7, block_hi=NB)
    tr._enqueue(TS.PHASE_BACKWARD, False, flags=TS.FLAG_GRADS_ONLY, block_lo=0, block_hi=7)
    torch.cuda.synchronize()
    assert torch.equal(tr.grads, runs[0][0][0])
    assert tr.optimizer_step == 0 and torch.isfinite(tr.loss).all()


def test_fused_data_parallel_step_has_two_buckets_and_equals_the_plain_step(pkg, oracle):
    """SURVEY 8(e) / VERDICT r3 item 3: on the fused path the gradient leaves in TWO buckets -- the upper blocks' (with
    lin1) while the lower blocks' launch runs -- and the sequence [chain | grads hi | all-reduce hi || grads lo | all-reduce
    lo | Adam] gives exactly the single-call step at world size 1 (eager and captured)."""
    snaps = pkg.wdn_synth.make_snapshots(3 * BS, NODES, seed=8).cuda()
    for use_graph in (False, True):
        res = []
        for split in (False, True):
            model, p, tr, ei = _trainer(pkg, oracle, seed=2, use_graph=use_graph, force_collective_path=split)
            tr.seed = 77
            tr.fused_buckets = 2                 # (the default is ONE bucket since round 5: the test below)
            assert tr.split == split
            losses = []
            for it in range(3):
                y = snaps[it * BS:(it + 1) * BS].reshape(-1)
                losses.append(float(tr.step(y, y)))
                if split:
                    cut = 2 * NC + (NB // 2) * (9 * NC + 4 * NC * NC)
                    assert tr.reducer.last_buckets == [(0, cut), (cut, tr.P)], tr.reducer.last_buckets
            res.append((losses, model.flat_parameters.clone(), tr.mask.clone(), tr.grads.clone()))
        assert res[0][0] == res[1][0], use_graph
        assert torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2]) and torch.equal(res[0][3], res[1][3])


def test_dropped_data_parallel_steps_are_counted(pkg, oracle):
    """the data-parallel Adam phase drops a step whose gradient carries the fault mark (entry 0 is NaN) -- a
    diverging run that puts a NaN there is dropped too, and that must be visible: ``dropped_steps`` counts it."""
    from gnn_pressure_estimation_amd import train_step as TS
    model, p, tr, ei = _trainer(pkg, oracle, bs=2, use_graph=False, force_collective_path=True)
    y = pkg.wdn_synth.make_snapshots(2, NODES, seed=2).reshape(-1).cuda()
    tr.step(y, y)
    assert tr.dropped_steps == 0 and tr.optimizer_step == 1
    before = model.flat_parameters.clone()
    tr.grads[0] = float("nan")
    tr._enqueue(TS.PHASE_ADAM, True)
    torch.cuda.synchronize()
    assert tr.dropped_steps == 1 and tr.optimizer_step == 1
    assert torch.equal(model.flat_parameters, before)
    tr.step(y, y)                                               # the next step is a normal one again
    assert tr.dropped_steps == 1 and tr.optimizer_step == 2 and torch.isfinite(tr.loss).all()


def test_mask_sampled_by_the_update_launch_equals_the_samplers_own_launch(pkg, oracle, monkeypatch):
    """Round 4: on the single-GPU fused path the update launch samples the NEXT step's device mask (GATRES_FLAG_MASK_NEXT) and
    bound batches are trained in place -- three launches per step.  Against the sampler's own launch + a staging copy
    (GATRES_NO_MASK_NEXT=1): the same masks, the same losses, the same parameters, bit for bit -- through step(x, y),
    step_bound(i), run_step(), across a change of the mask rate, a host-supplied mask in between and a sibling trainer's
    step (which moves the shared step count under the trainer: its sampled-ahead mask must be dropped)."""
    from gnn_pressure_estimation_amd import train_step as TS
    snaps = pkg.wdn_synth.make_snapshots(4 * BS, NODES, seed=31).cuda()
    batches = [snaps[i * BS:(i + 1) * BS].reshape(-1).contiguous() for i in range(4)]
    host_mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([NODES] * BS, 0.95, np.random.RandomState(9))).cuda()
    runs = []
    for ahead in (False, True):
        if not ahead:
            monkeypatch.setenv("GATRES_NO_MASK_NEXT", "1")
        else:
            monkeypatch.delenv("GATRES_NO_MASK_NEXT", raising=False)
        model, p, tr, ei = _trainer(pkg, oracle, seed=5, use_graph=True)
        assert tr._mask_next == ahead
        tr.bind_batches(batches)
        hist = []

        def rec():
            hist.append((tr.mask.clone(), float(tr.loss), model.flat_parameters.clone()))

        for i in range(3):
            tr.step_bound(i); rec()
        if ahead:       # from the second step on the captured step holds no sampler launch
            assert tr._mask_sig is not None
            assert any(not (k[0][0] & TS.PHASE_MASK) for k in tr._graphs if isinstance(k[0], tuple))
        tr.step(batches[3], batches[3]); rec()                  # staging copy, mask already in place
        tr.set_hparams(mask_rate=0.5)
        tr.step_bound(0); rec()
        assert int(tr.mask.sum()) == BS * int(NODES * 0.5)
        tr.step(batches[1], batches[1], host_mask); rec()       # the caller's mask replaces the sampled-ahead one
        assert torch.equal(tr.mask.bool(), host_mask.bool())
        tr.step_bound(2); rec()
        tr.load_batch(batches[3], batches[3]); tr.run_step(device_mask=True); rec()
        # a sibling (another batch size, the same optimizer state) takes a step in between
        ei2 = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(NODES, PIPES, seed=0), NODES, 2).cuda()
        sib = tr._sibling(2, NODES, ei2)
        sib.step(batches[0][:2 * NODES].contiguous(), batches[0][:2 * NODES].contiguous())
        tr.step_bound(1); rec()
        assert tr.optimizer_step == 10 and tr.fault_count == 0
        runs.append(hist)
    for k, (a, b) in enumerate(zip(*runs)):
        assert torch.equal(a[0], b[0]), f"mask of step {k}"
        assert a[1] == b[1], f"loss of step {k}"
        assert torch.equal(a[2], b[2]), f"parameters after step {k}"


def test_data_parallel_step_is_the_single_gpu_step_around_one_all_reduce(pkg, oracle):
    """Round 5: the multi-rank step of the fused path is the single-GPU step's own launches -- window
    kernel, parameter gradients, slab sum -- then ONE all-reduce of the flat gradient and the Adam launch, which samples the
    next step's mask exactly as the single-GPU update launch does; bound batches are read in place.  At world size 1 the
    sequence gives the plain step bit for bit (masks, losses, parameters), eager and captured, and from the second step on it
    holds no sampler launch."""
    from gnn_pressure_estimation_amd import train_step as TS
    snaps = pkg.wdn_synth.make_snapshots(3 * BS, NODES, seed=18).cuda()
    batches = [snaps[i * BS:(i + 1) * BS].reshape(-1).contiguous() for i in range(3)]
    for use_graph in (False, True):
        res = []
        for split in (False, True):
            model, p, tr, ei = _trainer(pkg, oracle, seed=4, use_graph=use_graph, force_collective_path=split)
            tr.seed = 55
            assert tr.split == split and tr.fused_buckets == 1 and tr._mask_next
            tr.bind_batches(batches)
            hist = []
            for it in range(7):
                loss = float(tr.step_bound(it % 3))
                hist.append((loss, tr.mask.clone(), model.flat_parameters.clone()))
                if split:
                    assert tr.reducer.last_buckets == [(0, tr.P)], tr.reducer.last_buckets
                    assert tr._mask_sig is not None            # the Adam launch sampled the next step's mask
            if split and use_graph:
                keys = [k[0] for k in tr._graphs if isinstance(k[0], tuple) and k[0][0] == "split"]
                # one first-step graph (it still runs the sampler's launch), then premasked steady-state graphs only:
                # two per bound batch (the mask buffers take turns)
                assert sum(1 for k in keys if not k[1]) <= 1 and sum(1 for k in keys if k[1] and k[2] & TS.FLAG_MASK_NEXT) == 6, keys
            assert tr.optimizer_step == 7 and tr.fault_count == 0 and tr.dropped_steps == 0
            res.append(hist)
        for k, (a, b) in enumerate(zip(*res)):
            assert a[0] == b[0], (use_graph, k)
            assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), (use_graph, k)


def test_bound_steps_are_captured_at_bind_time(pkg, oracle):
    """Round 5: ``bind_batches`` captures the steady-state step of every (batch, mask buffer) pair by
    running steps and rolling the training state back, so no ``step_bound`` call ever holds a capture -- whatever the order
    of the batches -- and the run is bit-identical to one whose graphs are captured lazily."""
    snaps = pkg.wdn_synth.make_snapshots(4 * BS, NODES, seed=41).cuda()
    batches = [snaps[i * BS:(i + 1) * BS].reshape(-1).contiguous() for i in range(4)]
    order = [0, 1, 2, 3, 0, 0, 1, 3, 2, 2, 1]                 # (both mask buffers meet every batch)
    runs = []
    for pre in (True, False):
        model, p, tr, ei = _trainer(pkg, oracle, seed=6, use_graph=True)
        before = model.flat_parameters.clone()
        tr.bind_batches(batches, precapture=pre)
        if pre:
            assert tr.num_captured_graphs >= 2 * len(batches)
            assert torch.equal(model.flat_parameters, before) and tr.optimizer_step == 0
            assert float(tr.exp_avg.abs().max()) == 0.0 and float(tr.exp_avg_sq.abs().max()) == 0.0
        n0 = tr.num_captured_graphs
        hist = []
        for i in order:
            hist.append((float(tr.step_bound(i)), tr.mask.clone(), model.flat_parameters.clone()))
        if pre:
            assert tr.num_captured_graphs == n0, (n0, tr.num_captured_graphs)      # nothing was captured by a step
        assert tr.optimizer_step == len(order) and tr.fault_count == 0
        tr.check_no_dropped_steps()
        runs.append(hist)
    for k, (a, b) in enumerate(zip(*runs)):
        assert a[0] == b[0] and torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), k


def test_alternating_launch_kinds_on_one_scratch_buffer(pkg, oracle):
    """the window kernel's prologue publishes the part's geometry record to the other waves through L2
    (tid 0 stores, barrier, scalar loads).  A forward-only launch, a training launch and a backward-only launch write
    DIFFERENT records into the same words (the ReLU-mask tables exist only in training launches), so a stale record shows as
    wrong LDS offsets.  Alternate the three kinds on ONE freshly zeroed scratch buffer, starting with the first launch ever
    into it, and hold every result to kernels that do not share that state."""
    bs = BS
    x, y, ei, mask = pkg.wdn_synth.make_batch(bs, NODES, PIPES)
    dx, dy, dei, dmask = x.cuda(), y.cuda(), ei.cuda(), mask.cuda()
    mf, p = build(pkg, oracle, NB, NC, seed=9, fused=True)
    mp, _ = build(pkg, oracle, NB, NC, seed=9, fused=False)            # per-op kernels: no record, no LDS windows
    with torch.no_grad():
        first = mf(dx.reshape(-1, 1), dei)                             # the FIRST launch into this scratch: forward-only
        assert torch.equal(first, mp(dx.reshape(-1, 1), dei))
    tr = pkg.GATResTrainer(mf, dei, NODES * bs, nodes_per_graph=[NODES] * bs, use_graph=False)
    ref_model, _ = build(pkg, oracle, NB, NC, seed=9, fused=True)      # a trainer that only ever runs training launches
    ref = pkg.GATResTrainer(ref_model, dei, NODES * bs, nodes_per_graph=[NODES] * bs, use_graph=False)
    assert tr.scratch.data_ptr() == mf._scratch_for(tr.plan).data_ptr()
    for it in range(3):
        la, lb = float(tr.step(dx, dy, dmask)), float(ref.step(dx, dy, dmask))          # training launch
        assert la == lb and torch.equal(mf.flat_parameters, ref_model.flat_parameters), it
        load = {k: v.detach().clone() for k, v in mf.state_dict().items()}
        mp.load_state_dict(load)
        with torch.no_grad():                                          # forward-only launch (other record: no mask tables)
            assert torch.equal(mf(dx.reshape(-1, 1), dei), mp(dx.reshape(-1, 1), dei)), it
        xin = dx.clone().reshape(-1, 1); xin[dmask.bool()] = 0
        of = mf(xin, dei)                                              # forward with saved activations, then a
        op = mp(xin, dei)                                              # backward-only launch
        assert torch.equal(of, op), it
        g = torch.randn(of.shape, generator=torch.Generator().manual_seed(it)).cuda()
        mf.zero_grad(); mp.zero_grad()
        of.backward(g); op.backward(g)
        gf = torch.cat([q.grad.reshape(-1) for q in mf.parameters()])
        gp = torch.cat([q.grad.reshape(-1) for q in mp.parameters()])
        assert relerr(gf, gp) < 2e-5, (it, relerr(gf, gp))
    assert tr.fault_count == 0


def test_mask_next_is_refused_where_it_cannot_be_honoured(pkg, oracle, monkeypatch):
    """GATRES_FLAG_MASK_NEXT on a configuration whose parameter gradients run on consumer workgroups (no step-count
    snapshot for the sampling tail) used to be dropped silently; it is an error now, raised before anything is enqueued."""
    from gnn_pressure_estimation_amd import train_step as TS
    monkeypatch.setenv("GATRES_FUSED_WITH_CONSUMERS", "1")              # (opt-in since round 5: the stand-alone launch is the default)
    model, p, tr, ei = _trainer(pkg, oracle, bs=8, use_graph=False)      # bs 8: CUs are left, consumers form the gradients
    if tr._mask_next:
        pytest.skip("this configuration folds the update: nothing to refuse")
    y = pkg.wdn_synth.make_snapshots(8, NODES, seed=2).reshape(-1).cuda()
    tr.load_batch(y, y)
    before = tr.optimizer_step
    with pytest.raises(RuntimeError):
        tr._enqueue(TS.PHASE_MASK | TS.PHASE_FORWARD | TS.PHASE_BACKWARD | TS.PHASE_ADAM, True, flags=TS.FLAG_MASK_NEXT)
    torch.cuda.synchronize()
    assert tr.optimizer_step == before
    assert float(tr.step(y, y)) == float(tr.loss)                        # the trainer itself never asks for it here


def test_multi_step_sequences_equal_single_steps(pkg, oracle):
    """Round 5: ``steps_bound(indices)`` replays k full training steps as ONE hipGraph launch (the ~8 us between two graph
    launches are paid once per k steps).  Masks, losses and parameters after every sequence equal those of the single
    ``step_bound`` calls bit for bit -- even and odd lengths (an odd one leaves the two mask buffers swapped), sequences captured
    at bind time and lazily, a change of the mask rate in between."""
    snaps = pkg.wdn_synth.make_snapshots(4 * BS, NODES, seed=51).cuda()
    batches = [snaps[i * BS:(i + 1) * BS].reshape(-1).contiguous() for i in range(4)]
    plan = [(0, 1, 2, 3), (0, 1, 2), (3, 0), (1, 2, 3, 0, 1), (2,), (3, 0, 1, 2)]
    runs = []
    for seq_mode in (False, True):
        model, p, tr, ei = _trainer(pkg, oracle, seed=7, use_graph=True)
        tr.bind_batches(batches, sequences=[(0, 1, 2, 3), (0, 1, 2)] if seq_mode else ())
        n0 = tr.num_captured_graphs
        hist = []
        for k, seq in enumerate(plan):
            if k == 4:
                tr.set_hparams(mask_rate=0.5)
            if seq_mode:
                tr.steps_bound(seq)
            else:
                for i in seq:
                    tr.step_bound(i)
            hist.append((float(tr.loss), tr.mask.clone(), model.flat_parameters.clone(), tr.optimizer_step))
            if seq_mode and k == 1:
                assert tr.num_captured_graphs == n0          # both sequences so far were captured at bind time
        assert tr.fault_count == 0
        runs.append(hist)
    for k, (a, b) in enumerate(zip(*runs)):
        assert a[3] == b[3] and a[0] == b[0], (k, a[0], b[0])
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2]), k


def test_window_kernel_instantiations_are_bit_identical(pkg, oracle, lib, monkeypatch):
    """Round 6: a launch of the window kernel takes the instantiation that has everything the host knows about it as
    template arguments (k_window.hip: phases, keep-in-LDS, symmetric plan without consumer workgroups, rows of at most six
    entries, part tables, parts of at most 64 rows).  Every one of them -- down to the run-time-phases kernel of round 5 -- runs
    the same stage functions on the same operands: predictions, loss, the flat gradient and the parameters after two Adam
    steps must agree BIT FOR BIT, for the training launch, the forward-only launch (evaluation) and the nn.Module's
    forward + backward launches.  Plans that lack a fact (a directed graph, a hub row) take the slower instantiation by
    themselves: their parity tests are test_gpu_model.py's edge cases."""
    bs = 8
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(NODES, PIPES, seed=0), NODES, bs).cuda()
    y = pkg.wdn_synth.collate_snapshots(pkg.wdn_synth.make_snapshots(bs, NODES, seed=9), range(bs)).cuda()
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([NODES] * bs, 0.95, np.random.RandomState(4))).cuda()
    plan_flags = None

    def run(env):
        nonlocal plan_flags
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        model, _ = build(pkg, oracle, NB, NC, seed=13)
        tr = pkg.GATResTrainer(model, ei, NODES * bs, nodes_per_graph=[NODES] * bs, use_graph=False)
        assert lib.gatres_fused_window_kernel(model._cmodel_ref(), tr.plan.ref()) == 1
        plan_flags = tr.plan.flags
        out = {}
        tr.step(y, y, mask)
        out["pred"], out["loss"], out["grads"] = tr.out.clone(), tr.loss.clone(), tr.grads.clone()
        tr.step(y, y, mask)
        out["params"] = model.flat_parameters.clone()
        with torch.no_grad():
            out["eval"] = model(y.reshape(-1, 1), ei).clone()                      # forward-only launch
        model.zero_grad()
        o = model(y.reshape(-1, 1), ei)                                            # the module's forward + backward launches
        (o * o).mean().backward()
        out["module_out"], out["module_grads"] = o.detach().clone(), model.flat_grad().clone()
        for k in env:
            monkeypatch.delenv(k)
        return out

    ref = run({"GATRES_WINDOW_RUNTIME_PHASES": "1"})
    assert plan_flags & 1 and plan_flags & 4, "the synthetic WDN is symmetric with rows of at most six entries"
    for m in ("65535", "7167", "4095", "2047", "1023", "511"):       # (7167 = 0x1bff: everything but "rows of at most six entries")
        got = run({"GATRES_WINDOW_PH_MASK": m})
        for k, v in ref.items():
            assert torch.equal(v, got[k]), (m, k)
    # the training launch without keep-in-LDS (GATRES_FUSED_NO_KEEP: ReLU sign masks and own-row g_pre come back from HBM in the
    # backward phase, dX through the operand-in-memory form): the same arithmetic, the same bits
    got = run({"GATRES_FUSED_NO_KEEP": "1"})
    for k, v in ref.items():
        assert torch.equal(v, got[k]), ("no-keep", k)

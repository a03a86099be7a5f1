"""The "next" rows of the scope table: snapshot store (data format / collation) and evaluation path (metrics, timing)."""
import numpy as np
import pytest
import torch


def test_snapshot_store_matches_reference_collation(pkg):
    one = pkg.wdn_synth.make_wdn_topology(40, 47)
    raw = torch.randn(10, 40, generator=torch.Generator().manual_seed(0)) * 7 + 30           # "pressures"
    st = pkg.SnapshotStore(raw, one, device="cpu")
    mean, std = float(raw.mean()), float(raw.std(unbiased=False))
    assert abs(st.mean - mean) < 1e-5 and abs(st.std - std) < 1e-5
    rows = torch.tensor([3, 0, 7])
    x = st.batch(rows)
    ref = ((raw[rows] - mean) / (std + 1e-8)).reshape(-1, 1)                                   # auxil.py:18-39
    assert x.shape == (120, 1) and torch.allclose(x, ref, atol=1e-6)
    assert torch.equal(st.edge_index(3), pkg.wdn_synth.collate_edge_index(one, 40, 3))        # PyG Batch convention
    assert st.edge_index(3) is st.edge_index(3)                                               # cached object
    assert torch.allclose(st.descale(x), raw[rows].reshape(-1, 1), atol=1e-4)
    seen = sorted(int(n) for _, _, n in st.batches(4, shuffle=True, generator=torch.Generator().manual_seed(1)))
    assert seen == [2, 4, 4]
    assert [n for _, _, n in st.batches(4, shuffle=False, drop_last=True)] == [4, 4]
    with pytest.raises(ValueError):
        pkg.SnapshotStore(raw, torch.tensor([[0, 41], [1, 2]]), device="cpu")
    # row_batches: the same epoch as batches(), as row indices for consumers that collate on the device themselves
    a = list(st.batches(4, shuffle=True, generator=torch.Generator().manual_seed(9)))
    b = list(st.row_batches(4, shuffle=True, generator=torch.Generator().manual_seed(9)))
    assert len(a) == len(b) == 3
    for (x1, e1, n1), (rows1, e2, n2) in zip(a, b):
        assert n1 == n2 and e1 is e2 and rows1.dtype == torch.int64 and rows1.numel() == n1
        assert torch.equal(st.batch(rows1), x1)


def test_metrics_match_hand_values(pkg):
    E = pkg.evaluation
    t = torch.tensor([10.0, 20.0, 30.0, 0.005, 40.0])
    p = torch.tensor([11.0, 18.0, 33.0, 1.0, 40.0])
    assert abs(float(E.calculate_rmse(p, t)) - np.sqrt(((p - t) ** 2).mean().item())) < 1e-6
    keep = [0, 1, 2, 4]                                                                         # |y_true| > 0.01
    assert abs(float(E.calculate_rel_error(p, t)) - np.mean([1 / 10, 2 / 20, 3 / 30, 0.0])) < 1e-6
    assert abs(float(E.calculate_accuracy(p, t, threshold=0.1)) - np.mean([1, 1, 1, 0, 1])) < 1e-6
    r = np.corrcoef(p.numpy(), t.numpy())[0, 1]
    assert abs(float(E.calculate_correlation_coefficient(p, t)) - r) < 1e-6
    assert abs(float(E.calculate_r2(p, t)) - r * r) < 1e-6
    nse = 1 - ((p - t) ** 2).sum() / (((t - t.mean()) ** 2).sum() + 1e-12)
    assert abs(float(E.calculate_nse(p, t)) - float(nse)) < 1e-6
    keys = list(E.get_metric_fn_collection("val").keys())
    assert keys == ["val_error", "val_0.1", "val_corr", "val_r2", "val_mae", "val_rmse", "val_mynse"]
    assert float(E.descale(torch.tensor(2.0), "znorm", mean=3.0, std=0.5)) == 4.0
    assert float(E.descale(torch.tensor(0.25), "minmax", min=10.0, max=50.0)) == 20.0


@pytest.mark.gpu
def test_evaluation_loop_on_device(pkg, oracle):
    one = pkg.wdn_synth.make_wdn_topology()
    raw = pkg.wdn_synth.make_snapshots(12, 388, seed=3) * 5 + 40
    st = pkg.SnapshotStore(raw, one, device="cuda")
    p = oracle.init_params(15, 32, seed=3)
    model = pkg.GATResMeanConv(num_blocks=15, nc=32)
    sd = {}
    for k, v in p.items():
        sd[k] = v
        if k.endswith("lin_src.weight"):
            sd[k.replace("lin_src", "lin_dst")] = v
    model.load_state_dict(sd)
    model = model.cuda()
    loss, m = pkg.evaluation.test_one_epoch(model, st.batches(4, shuffle=False), 0.95, mean=st.mean, std=st.std,
                                            norm_type="znorm", gpu_warmup_times=2, rng=np.random.RandomState(0))
    assert set(m) == {"test_error", "test_0.1", "test_corr", "test_r2", "test_mae", "test_rmse", "test_mynse",
                      "test_time", "test_throughput"}
    assert np.isfinite(loss) and m["test_time"] > 0 and m["test_throughput"] > 0
    # the same evaluation through the oracle on the CPU (same masks: same RandomState stream)
    rng = np.random.RandomState(0)
    tot, n = 0.0, 0
    fns = pkg.evaluation.get_metric_fn_collection("test")
    acc = {k: 0.0 for k in fns}
    for x, ei, g in pkg.SnapshotStore(raw, one, device="cpu").batches(4, shuffle=False):
        mask = pkg.wdn_synth.generate_batch_mask([388] * g, 0.95, rng)
        x1 = x.clone(); x1[mask] = 0
        out = oracle.gatres_forward(p, x1, ei)
        tot += float(torch.nn.functional.mse_loss(out[mask], x[mask])) * g
        for k, fn in fns.items():
            acc[k] += float(fn(out[mask] * st.std + st.mean, x[mask] * st.std + st.mean)) * g
        n += g
    assert abs(loss - tot / n) < 1e-5 * abs(tot / n)
    for k in fns:
        assert abs(m[k] - acc[k] / n) <= 1e-4 * max(1.0, abs(acc[k] / n)), k


def test_mask_sampler_with_required_indices(pkg):
    """utils/auxil.py:143-163: the required (sensor) nodes are always masked and count towards int(n * rate); without
    them the sampler draws exactly what it drew before (same RandomState stream)."""
    ws = pkg.wdn_synth
    req = [3, 17, 200, 387]
    m = ws.generate_batch_mask([388] * 5, 0.95, np.random.RandomState(4), req)
    assert m.shape == (5 * 388,) and m.dtype == bool
    for g in range(5):
        mg = m[g * 388:(g + 1) * 388]
        assert mg.sum() == int(388 * 0.95) and mg[req].all()
    assert not np.array_equal(m[:388], m[388:776])
    a = ws.generate_batch_mask([40, 388], 0.9, np.random.RandomState(1))
    b = ws.generate_batch_mask([40, 388], 0.9, np.random.RandomState(1), [])
    assert np.array_equal(a, b)
    with pytest.raises(ValueError):
        ws.mask_nodes(10, 0.5, np.random.RandomState(0), [0, 1, 2, 3, 4])          # nothing left to draw


@pytest.mark.gpu
def test_sensor_pass_and_trial_loop(pkg, oracle):
    """evaluation.py:355-403: every trial runs an all-nodes pass and a sensor pass (masks that always contain the sensor
    nodes, metric keys with the `_sensor` postfix); the sensor pass is checked against the oracle under the same masks."""
    one = pkg.wdn_synth.make_wdn_topology()
    raw = pkg.wdn_synth.make_snapshots(6, 388, seed=3) * 5 + 40
    st = pkg.SnapshotStore(raw, one, device="cuda")
    p = oracle.init_params(3, 32, seed=3)
    model = pkg.GATResMeanConv(num_blocks=3, nc=32)
    sd = {}
    for k, v in p.items():
        sd[k] = v
        if k.endswith("lin_src.weight"):
            sd[k.replace("lin_src", "lin_dst")] = v
    model.load_state_dict(sd)
    model = model.cuda()
    req = [5, 9, 120, 300]
    E = pkg.evaluation
    losses, metrics, s_losses, s_metrics = E.test_trials(model, lambda: st.batches(3, shuffle=False), 2, 0.95, required_idx=req,
                                                         mean=st.mean, std=st.std, norm_type="znorm", gpu_warmup_times=1,
                                                         rng=np.random.RandomState(0))
    assert len(losses) == 2 and len(s_losses) == 2 and all(np.isfinite(losses + s_losses))
    assert set(metrics) == {"test_error", "test_0.1", "test_corr", "test_r2", "test_mae", "test_rmse", "test_mynse",
                            "test_time", "test_throughput"}
    assert set(s_metrics) == {k + "_sensor" for k in metrics} and all(len(v) == 2 for v in s_metrics.values())
    # the masks of the run above, replayed: trial 0 = plain pass (2 batches), then the sensor pass (2 batches), ...
    rng = np.random.RandomState(0)
    cpu = pkg.SnapshotStore(raw, one, device="cpu")
    for trial in range(2):
        for sensors in (False, True):
            tot, n = 0.0, 0
            for x, ei, g in cpu.batches(3, shuffle=False):
                mask = pkg.wdn_synth.generate_batch_mask([388] * g, 0.95, rng, req if sensors else [])
                if sensors:
                    assert all(mask[k * 388 + np.array(req)].all() for k in range(g))
                x1 = x.clone(); x1[mask] = 0
                out = oracle.gatres_forward(p, x1, ei)
                tot += float(torch.nn.functional.mse_loss(out[mask], x[mask])) * g
                n += g
            got = (s_losses if sensors else losses)[trial]
            assert abs(got - tot / n) < 1e-5 * abs(tot / n), (trial, sensors)

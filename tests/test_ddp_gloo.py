"""The N>1 path on CPU: world_size-2 gloo processes, snapshots sharded by graph, ONE all-reduce of the flat gradient.
The compute inside each rank is the oracle (no GPU here); what is under test is the data-parallel scheme the
trainer uses on the GPU box: shard assignment, gradient averaging == global-batch gradient, identical replicas."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NB, NC, B, NODES, PIPES = 2, 8, 4, 40, 47


def _global_batch(pkg):
    ei1 = pkg.wdn_synth.make_wdn_topology(NODES, PIPES)
    snaps = pkg.wdn_synth.make_snapshots(B, NODES, seed=11)
    mask = pkg.wdn_synth.generate_batch_mask([NODES] * B, 0.9, np.random.RandomState(5))
    return ei1, snaps, torch.from_numpy(mask)


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gnn_pressure_estimation_amd as G
    from oracle import gatres_oracle as O
    ei1, snaps, mask = _global_batch(G)
    rows = G.dp.shard_graphs(B, rank, world)
    per = len(rows)
    x = G.wdn_synth.collate_snapshots(snaps, rows)
    ei = G.wdn_synth.collate_edge_index(ei1, NODES, per)
    m = mask[rows[0] * NODES:(rows[-1] + 1) * NODES]
    # replicas: rank 1 starts from different weights on purpose; broadcast must fix that
    p = O.init_params(NB, NC, seed=1 + 10 * rank)
    flat = O.flatten(p).clone()
    G.dp.broadcast_params_(flat)
    off = 0
    for k in p:
        n = p[k].numel(); p[k] = flat[off:off + n].view(p[k].shape).clone(); off += n
    tr = O.OracleTrainer(p)
    xin = x.clone(); xin[m] = 0
    out = O.gatres_forward(tr.params, xin, ei)
    loss = torch.nn.functional.mse_loss(out[m], x[m])
    loss.backward()
    g = tr.flat("grads").clone()
    G.dp.allreduce_mean_(g)
    lsum = loss.detach().clone(); dist.all_reduce(lsum)
    torch.save({"grads": g, "loss": lsum / world, "params": flat}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_average_equals_global_batch(pkg, oracle, tmp_path):
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["params"], r1["params"])                       # replicas identical after broadcast
    assert torch.equal(r0["grads"], r1["grads"])                         # every rank holds the same averaged gradient
    # single-process global batch
    ei1, snaps, mask = _global_batch(pkg)
    x = pkg.wdn_synth.collate_snapshots(snaps, range(B))
    ei = pkg.wdn_synth.collate_edge_index(ei1, NODES, B)
    p = oracle.init_params(NB, NC, seed=1)
    tr = oracle.OracleTrainer(p)
    xin = x.clone(); xin[mask] = 0
    out = oracle.gatres_forward(tr.params, xin, ei)
    loss = torch.nn.functional.mse_loss(out[mask], x[mask])
    loss.backward()
    g = tr.flat("grads")
    assert float((r0["grads"] - g).abs().max() / g.abs().max()) < 1e-5
    assert abs(float(r0["loss"]) - float(loss)) < 1e-6 * abs(float(loss))


def test_shard_assignment(pkg):
    assert [list(pkg.dp.shard_graphs(256, r, 8))[:2] for r in (0, 7)] == [[0, 1], [224, 225]]
    assert len(pkg.dp.shard_graphs(35, 3, 4)) == 8                       # ragged tail dropped: equal shards
    with pytest.raises(ValueError):
        pkg.dp.shard_graphs(8, 2, 2)

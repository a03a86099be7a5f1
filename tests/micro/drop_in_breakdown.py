"""Diagnostic (not a test): where does the reference loop body spend its time on the drop-in module?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
torch.set_num_threads(8)
import gnn_pressure_estimation_amd as G

bs, npg = 32, 388
dev = torch.device("cuda:0")
model = G.GATResMeanConv(num_blocks=15, nc=32).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=5e-4, weight_decay=6e-6)
crit = torch.nn.MSELoss()
topo = G.wdn_synth.make_wdn_topology()
snaps = G.wdn_synth.make_snapshots(8 * bs, npg, seed=100)
ei_cpu = G.wdn_synth.collate_edge_index(topo, npg, bs)
rng = np.random.RandomState(0)
T = {}
def tick(name, t0):
    torch.cuda.synchronize()
    T[name] = T.get(name, 0.0) + time.perf_counter() - t0
    return time.perf_counter()
for i in range(60):
    if i == 10:
        T.clear()
    t = time.perf_counter()
    opt.zero_grad(); t = tick("zero_grad", t)
    y = G.wdn_synth.collate_snapshots(snaps, range((i % 8) * bs, (i % 8 + 1) * bs)); t = tick("collate(host)", t)
    x, yd, ei = y.clone().to(dev), y.to(dev), ei_cpu.clone().to(dev); t = tick("H2D", t)
    m = G.wdn_synth.generate_batch_mask([npg] * bs, 0.95, rng); t = tick("mask(host numpy)", t)
    x[m] = 0; t = tick("x[mask]=0", t)
    out = model(x, ei, None, None); t = tick("model forward", t)
    loss = crit(out[m], yd[m]); t = tick("loss", t)
    loss.backward(); t = tick("backward", t)
    opt.step(); t = tick("opt.step", t)
    float(loss); t = tick("loss.item", t)
tot = sum(T.values())
for k, v in T.items():
    print(f"{k:20s} {v / 50 * 1e3:8.3f} ms")
print(f"{'total':20s} {tot / 50 * 1e3:8.3f} ms  (with a sync after every phase)")

"""Diagnostic: where does the bs-32 gradient of one tensor differ from the fp64 oracle?  (fused vs per-op vs oracle32)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnn_pressure_estimation_amd as G
from oracle import gatres_oracle as O
from test_gpu_model import build

NB, NC, BS, NODES, PIPES = 15, 32, 32, 388, 430
p = O.init_params(NB, NC, seed=3)
ei = G.wdn_synth.collate_edge_index(G.wdn_synth.make_wdn_topology(NODES, PIPES, seed=0), NODES, BS)
snaps = G.wdn_synth.make_snapshots(3 * BS, NODES, seed=6)
rng = np.random.RandomState(1)
y = G.wdn_synth.collate_snapshots(snaps, range(0, BS))
mask = torch.from_numpy(G.wdn_synth.generate_batch_mask([NODES] * BS, 0.95, rng))
xin = y.clone(); xin[mask] = 0

def ograds(dt):
    l = {k: v.to(dt).clone().requires_grad_(True) for k, v in p.items()}
    o = O.gatres_forward(l, xin.to(dt), ei, num_blocks=NB)
    torch.nn.functional.mse_loss(o[mask], y.to(dt)[mask]).backward()
    return torch.cat([v.grad.reshape(-1).double() for v in l.values()])
g64, g32 = ograds(torch.float64), ograds(torch.float32)
res = {}
for name, fused in (("fused", True), ("per_op", False)):
    model, _ = build(G, O, NB, NC, seed=3, fused=fused)
    tr = G.GATResTrainer(model, ei.cuda(), NODES * BS, nodes_per_graph=[NODES] * BS, use_graph=False, fused=fused)
    tr.forward_backward(y.cuda(), y.cuda(), mask.cuda())
    torch.cuda.synchronize()
    res[name] = tr.grads.detach().cpu().double()
off = 0
print("tensor: max|g64|, err32, err_fused, err_perop (max abs), argmax_fused")
for k, v in p.items():
    n = v.numel()
    sl = slice(off, off + n)
    e32 = (g32[sl] - g64[sl]).abs(); ef = (res["fused"][sl] - g64[sl]).abs(); ep = (res["per_op"][sl] - g64[sl]).abs()
    if "blocks.8." in k or "blocks.9." in k or "blocks.7." in k or ef.max() > 20 * e32.max():
        print(f"{k:34s} {float(g64[sl].abs().max()):9.3e} {float(e32.max()):9.2e} {float(ef.max()):9.2e} {float(ep.max()):9.2e} {int(ef.argmax())}  "
              f"fused-vs-perop {float((res['fused'][sl]-res['per_op'][sl]).abs().max()):9.2e}")
    if k == "blocks.8.conv1.lin_src.weight":
        top = torch.topk(ef, 6)
        print("   top fused errors:", [(int(i), float(e), float(ep[i]), float(e32[i])) for e, i in zip(top.values, top.indices)])
        print("   median fused err %.2e  median perop %.2e  median o32 %.2e" % (float(ef.median()), float(ep.median()), float(e32.median())))
    off += n

"""Diagnostic: Python-side cost of the drop-in module's forward/backward (cProfile, top entries)."""
import os, sys, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
torch.set_num_threads(8)
import gnn_pressure_estimation_amd as G
bs, npg = 32, 388
dev = torch.device("cuda:0")
model = G.GATResMeanConv(num_blocks=15, nc=32).to(dev)
ei_cpu = G.wdn_synth.collate_edge_index(G.wdn_synth.make_wdn_topology(), npg, bs)
x = torch.randn(bs * npg, 1, device=dev)
def it():
    ei = ei_cpu.clone().to(dev)
    model.zero_grad()
    out = model(x, ei, None, None)
    out.sum().backward()
for _ in range(10):
    it()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(50):
    it()
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)

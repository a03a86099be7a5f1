import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import gnn_pressure_estimation_amd as G
from torch.profiler import profile, ProfilerActivity
bs, npg = 32, 388
dev = torch.device("cuda:0")
model = G.GATResMeanConv(num_blocks=15, nc=32).to(dev)
ei = G.wdn_synth.collate_edge_index(G.wdn_synth.make_wdn_topology(), npg, bs).to(dev)
x = torch.randn(bs * npg, 1, device=dev)
def it():
    model.zero_grad()
    out = model(x, ei, None, None)
    out.sum().backward()
for _ in range(10): it()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): it()
torch.cuda.synchronize()
print("fwd+bwd per iteration: %.3f ms" % ((time.perf_counter() - t0) / 50 * 1e3))
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(10): it()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=60))
print(prof.key_averages().table(sort_by="self_cpu_time_total", row_limit=14, max_name_column_width=60))

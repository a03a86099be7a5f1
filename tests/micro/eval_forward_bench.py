import os, sys, time
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/bench.py') else '.')
import torch, gnn_pressure_estimation_amd as G
bs=32
m=G.GATResMeanConv(num_blocks=15,nc=32).cuda().eval()
ei=G.wdn_synth.collate_edge_index(G.wdn_synth.make_wdn_topology(),388,bs).cuda()
x=torch.randn(388*bs,1).cuda()
with torch.no_grad():
    for _ in range(20): o=m(x,ei)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(300): o=m(x,ei)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
print("eval forward: %.3f ms per batch, %.0f snapshots/s"%(dt/300*1e3, bs*300/dt), "EVAL_SAVED" if os.environ.get("GATRES_EVAL_SAVED") else "")

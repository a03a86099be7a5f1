"""Diagnostic (not a test): per-stage wall time of the fused kernel for segment 0.  python tests/stage_profile.py"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnn_pressure_estimation_amd as G

bs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nb, nc = 15, 32
model = G.GATResMeanConv(num_blocks=nb, nc=nc).cuda()
ei = G.wdn_synth.collate_edge_index(G.wdn_synth.make_wdn_topology(), 388, bs).cuda()
tr = G.GATResTrainer(model, ei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=False)
y = torch.randn(388 * bs, device="cuda")
lib = G._native.load()
cap = 4096
stamps = torch.zeros(cap + 80, dtype=torch.int64, device="cuda")
for _ in range(3):
    tr.step(y, y)
lib.gatres_fused_set_stamps(stamps.data_ptr(), cap)
tr.step(y, y)
torch.cuda.synchronize()
lib.gatres_fused_set_stamps(None, 0)
raw = stamps.cpu().numpy()
clk = raw[cap:cap + 3]
s = raw[:1024]
s = s[s > 0]
cs = raw[1024:cap]
cs = cs[cs > 0]
print('shader clock MHz ~', (clk[1] - clk[0]) / ((clk[2] - s[0]) / 100.0))
d = (s[1:] - s[:-1]) / 100.0   # us
print("stamps", len(s), "total us", (s[-1] - s[0]) / 100.0)
names_f = ["proj1", "agg1", "proj2", "agg2", "mean"]
names_b = ["mean_bwd", "dst2", "src2", "dx2", "dst1", "src1", "dx1"]
i = 0
print("lin0 %.2f" % d[i]); i += 1
acc = {}
for b in range(nb):
    for nme in names_f:
        acc.setdefault("f_" + nme, []).append(d[i]); i += 1
print("lin1 %.2f" % d[i]); i += 1
print("loss %.2f" % d[i]); i += 1
print("lin1_bwd %.2f" % d[i]); i += 1
for b in range(nb):
    for nme in names_b:
        acc.setdefault("b_" + nme, []).append(d[i]); i += 1
tot = 0
for k, v in acc.items():
    print(f"{k:12s} mean {sum(v)/len(v):8.2f} us  x{len(v)}  sum {sum(v):8.1f}")
    tot += sum(v)
print("sum of block stages", tot, "remaining stamps", len(d) - i)

if len(cs) >= 2:
    # consumer workgroup 0: [item available, item done] pairs (100 MHz wall clock)
    t0 = s[0]
    avail, done = cs[0::2], cs[1::2]
    n = min(len(avail), len(done))
    work = (done[:n] - avail[:n]) / 100.0
    print("consumer 0: %d items, work per item mean %.2f us (min %.2f max %.2f), first available at %.1f us, last done at %.1f us"
          % (n, work.mean(), work.min(), work.max(), (avail[0] - t0) / 100.0, (done[n - 1] - t0) / 100.0))
    print("   wait before each item (us):", " ".join("%.1f" % w for w in ((avail[1:n] - done[:n - 1]) / 100.0)))
    print("   work of each item (us):   ", " ".join("%.1f" % w for w in work))

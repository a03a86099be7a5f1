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
cs = raw[1024:2048]
cs = cs[cs > 0]
print('shader clock MHz ~', (clk[1] - clk[0]) / ((clk[2] - s[0]) / 100.0))
d = (s[1:] - s[:-1]) / 100.0   # us
print("stamps", len(s), "total us", (s[-1] - s[0]) / 100.0)
names_f = ["proj1", "agg1", "proj2", "agg2", "mean"]
names_b = ["xch_b1", "k3+dst2", "src2", "dx2", "dst1", "src1", "dx1"]
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

# one backward block of segment 0, every part, wave 0 / last wave (k_fused.hip XSTAMP)
xs = raw[2048:2048 + 16 * 64].reshape(16, 64)
names_x = ["top", "bar", "B1 export", "B1 import", "B1 after", "bar", "k3+dots2+smax2+bar",
           "B2 export", "B2 import", "B2 after", "bar", "agg_src2+bar", "dX2+sync", "dots1+smax1+bar",
           "B3 export", "B3 import", "B3 after", "bar", "agg_src1+bar", "dX1"]
if xs[0, 0] > 0:
    t0 = min(int(xs[r, 0]) for r in range(16) if xs[r, 0] > 0)
    print("\none backward block, times in us since the first part entered it; columns: part p wave 0 | last wave")
    rows = [r for r in range(16) if xs[r, 0] > 0]
    print("%-18s" % "step" + "".join("   p%d w0   wL " % (r // 2) for r in rows if r % 2 == 0))
    for k, nme in enumerate(names_x):
        print("%-18s" % nme + "".join(" %6.2f" % ((int(xs[r, k]) - t0) / 100.0) + ("" if r % 2 else "") for r in rows))

# steps inside the consumer's streamed items (k_fused.hip ISTAMP): 8 stamps per item from slot 3072 + 8
it = raw[3072 + 8:3072 + 8 + 8 * 40].reshape(40, 8)
it = it[it[:, 0] > 0]
if len(it):
    d = (it[:, 1:] - it[:, :1]) / 100.0
    print("\nconsumer item steps, us since the item's start (mean over %d items):" % len(it))
    for k, nme in enumerate(["fold done (compute waves)", "chunk 0 landed", "chunk 1 landed", "last chunk landed", "last chunk computed",
                             "partials in LDS", "slab written"]):
        print("  %-28s %6.2f" % (nme, d[:, k].mean()))

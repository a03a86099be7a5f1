"""Diagnostic (not a test): the blocked launches of k_blocked.hip against the per-op pairs they replace, on config 3's sizes
(gatres_large, C-Town, bs 128: 49 664 rows), each timed alone through the C-ABI with HIP events, cold-ish (rotating buffer sets).
  python tests/micro/blocked_probe.py [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gnn_pressure_estimation_amd as G

bs = 128
if "--lib" in sys.argv:                  # a probe build (tests/micro/build_bk_probes.sh): timing only, results are wrong
    G._native.load_unchecked(sys.argv[sys.argv.index("--lib") + 1])
lib = G._native.load()
chk = G._native.check
ei = G.wdn_synth.collate_edge_index(G.wdn_synth.make_wdn_topology(388, 430, seed=0), 388, bs).cuda()
plan = G.GraphPlan(ei, 388 * bs, device=ei.device, segments=False)
N, Eg, nc, BF16 = plan.num_nodes, plan.num_edges_gat, 128, 1
gp = plan.ref()
st = lambda: G._native.current_stream(ei.device)
R = 6                                    # rotating buffer sets (> L2, < Infinity Cache for the small tables)
rn = lambda *s: torch.randn(*s, device="cuda")
bf = lambda *s: torch.randn(*s, device="cuda").bfloat16()


def timeit(fns, reps=60):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        fns[r % len(fns)]()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


sets = []
for r in range(R):
    d = dict(h1=bf(N, 256), a_s=rn(N, 2), a_d=rn(N, 2), bias=rn(256), W2=(rn(128, 256) / 16).bfloat16(), att_s=rn(128), att_d=rn(128),
             o=bf(N, 256), al=torch.rand(Eg, 2, device="cuda"), h2=bf(N, 128), as2=rn(N), ad2=rn(N),
             y2=bf(N, 128), x0=bf(N, 128), xn=bf(N, 128), W1=(rn(256, 128) / 11).bfloat16(), att_s1=rn(256), att_d1=rn(256), hn=bf(N, 256),
             as1=rn(N, 2), ad1=rn(N, 2),
             go2=bf(N, 128), al2=torch.rand(Eg, 1, device="cuda"), ge2=rn(Eg, 1), gad2=rn(N, 1), gh2=bf(N, 128), gas2=rn(N, 1), Wt2=(rn(256, 128) / 11).bfloat16(), ref2=bf(N, 256), gx2=bf(N, 256),
             go1=bf(N, 256), ge1=rn(Eg, 2), gad1=rn(N, 2), gh1=bf(N, 256), gas1=rn(N, 2), Wt1=(rn(128, 256) / 16).bfloat16(), res1=bf(N, 128), ref1=bf(N, 128), gx1=bf(N, 128))
    sets.append(d)
P = lambda t: t.data_ptr()


def pair_agg(d):
    chk(lib.gatres_t_gat_aggregate_fwd(gp, P(d["h1"]), P(d["a_s"]), P(d["a_d"]), P(d["bias"]), P(d["o"]), P(d["al"]), 2, nc, 1, BF16, st()), "a")
    chk(lib.gatres_t_proj_attn_fwd(P(d["o"]), P(d["W2"]), P(d["att_s"]), P(d["att_d"]), P(d["h2"]), P(d["as2"]), P(d["ad2"]), N, 256, 1, nc, BF16, st()), "p")
def blk_agg(d):
    chk(lib.gatres_bf16_agg_proj_fwd(gp, P(d["h1"]), P(d["a_s"]), P(d["a_d"]), P(d["bias"]), P(d["o"]), P(d["al"]), P(d["W2"]), P(d["att_s"]), P(d["att_d"]), P(d["h2"]), P(d["as2"]), P(d["ad2"]), nc, st()), "b")
def pair_mean(d):
    chk(lib.gatres_t_mean_residual_relu_fwd(gp, P(d["y2"]), P(d["x0"]), P(d["xn"]), nc, BF16, st()), "m")
    chk(lib.gatres_t_proj_attn_fwd(P(d["xn"]), P(d["W1"]), P(d["att_s1"]), P(d["att_d1"]), P(d["hn"]), P(d["as1"]), P(d["ad1"]), N, 128, 2, nc, BF16, st()), "p")
def blk_mean(d):
    chk(lib.gatres_bf16_mean_proj_fwd(gp, P(d["y2"]), P(d["x0"]), P(d["xn"]), P(d["W1"]), P(d["att_s1"]), P(d["att_d1"]), P(d["hn"]), P(d["as1"]), P(d["ad1"]), nc, st()), "b")
def pair_src2(d):
    chk(lib.gatres_t_gat_aggregate_bwd_src(gp, P(d["go2"]), P(d["al2"]), P(d["ge2"]), P(d["gad2"]), P(d["att_s"]), P(d["att_d"]), P(d["gh2"]), P(d["gas2"]), 1, nc, BF16, st()), "s")
    chk(lib.gatres_t_proj_bwd_dx(P(d["gh2"]), P(d["Wt2"]), None, P(d["ref2"]), P(d["gx2"]), N, 256, 128, BF16, st()), "d")
def blk_src2(d):
    chk(lib.gatres_bf16_src_dx_bwd(gp, P(d["go2"]), P(d["al2"]), P(d["ge2"]), P(d["gad2"]), P(d["att_s"]), P(d["att_d"]), P(d["gh2"]), P(d["gas2"]), P(d["Wt2"]), None, P(d["ref2"]), P(d["gx2"]), 1, nc, st()), "b")
def pair_src1(d):
    chk(lib.gatres_t_gat_aggregate_bwd_src(gp, P(d["go1"]), P(d["al"]), P(d["ge1"]), P(d["gad1"]), P(d["att_s1"]), P(d["att_d1"]), P(d["gh1"]), P(d["gas1"]), 2, nc, BF16, st()), "s")
    chk(lib.gatres_t_proj_bwd_dx(P(d["gh1"]), P(d["Wt1"]), P(d["res1"]), P(d["ref1"]), P(d["gx1"]), N, 128, 256, BF16, st()), "d")
def blk_src1(d):
    chk(lib.gatres_bf16_src_dx_bwd(gp, P(d["go1"]), P(d["al"]), P(d["ge1"]), P(d["gad1"]), P(d["att_s1"]), P(d["att_d1"]), P(d["gh1"]), P(d["gas1"]), P(d["Wt1"]), P(d["res1"]), P(d["ref1"]), P(d["gx1"]), 2, nc, st()), "b")


for name, pf, bfn in (("agg1 + proj2", pair_agg, blk_agg), ("mean + proj1", pair_mean, blk_mean), ("src2 + dx2", pair_src2, blk_src2),
                      ("src1 + dx1", pair_src1, blk_src1)):
    tp = timeit([lambda d=d: pf(d) for d in sets])
    tb = timeit([lambda d=d: bfn(d) for d in sets])
    print(f"{name:14s}  per-op pair {tp:7.2f} us   blocked {tb:7.2f} us   ({(tp - tb):+6.2f})")

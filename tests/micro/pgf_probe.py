"""TEMP: time param_grads (old) vs param_grads_finish variants."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gnn_pressure_estimation_amd as G
NB, NC, BS = 15, 32, 32
model = G.GATResMeanConv(num_blocks=NB, nc=NC).cuda()
ei = G.wdn_synth.collate_edge_index(G.wdn_synth.make_wdn_topology(), 388, BS).cuda()
tr = G.GATResTrainer(model, ei, 388 * BS, nodes_per_graph=[388] * BS, use_graph=False)
y = torch.randn(388 * BS, device="cuda")
for _ in range(3): tr.step(y, y)
lib = G._native.load(); st = lambda: G._native.current_stream(tr.device)
m, g = model._cmodel_ref(), tr.plan.ref(model._cmodel_ref())
h = tr.hparams
def t(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
pg = (m, g, tr.saved.data_ptr(), tr.scratch.data_ptr())
def fin(adam):
    return lambda: lib.gatres_fused_param_grads_finish(*pg, tr.grads.data_ptr(), None, None, adam, model.flat_parameters.data_ptr(),
        tr.exp_avg.data_ptr(), tr.exp_avg_sq.data_ptr(), tr.step_counter.data_ptr(), h["lr"], h["beta1"], h["beta2"], h["eps"],
        h["weight_decay"], None, 1.0, 0, NB, st())
old = lambda: lib.gatres_fused_param_grads(*pg, st())
oldfin = lambda: lib.gatres_fused_finish(m, g, tr.scratch.data_ptr(), tr.grads.data_ptr(), None, None, 1, model.flat_parameters.data_ptr(),
        tr.exp_avg.data_ptr(), tr.exp_avg_sq.data_ptr(), tr.step_counter.data_ptr(), h["lr"], h["beta1"], h["beta2"], h["eps"], h["weight_decay"], 1.0, st())
print("old param_grads %.1f us; old finish %.1f us; both %.1f" % (t(old), t(oldfin), t(lambda: (old(), oldfin()))))
print("folded adam=1, hp=None        %.1f us" % t(fin(1)))
def fin_hp(adam):
    return lambda: lib.gatres_fused_param_grads_finish(*pg, tr.grads.data_ptr(), None, None, adam, model.flat_parameters.data_ptr(),
        tr.exp_avg.data_ptr(), tr.exp_avg_sq.data_ptr(), tr.step_counter.data_ptr(), h["lr"], h["beta1"], h["beta2"], h["eps"],
        h["weight_decay"], tr.hp.data_ptr(), 1.0, 0, NB, st())
print("folded adam=1, hp=device buf  %.1f us" % t(fin_hp(1)))
print("folded adam=0                 %.1f us" % t(fin(0)))
for _ in range(300): tr.step(y, y)
print("after 300 more steps: folded adam=1 hp %.1f us, old pair %.1f" % (t(fin_hp(1)), t(lambda: (old(), oldfin()))))

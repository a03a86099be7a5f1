"""Diagnostic (not a test): what the slab sum of the update launch costs as the number of gradient slabs grows -- 32 (one per
snapshot: today) vs 256 (one per (snapshot, part): what forming the weight gradients inside the window kernel's stages would
leave behind).  gatres_reduce_slabs through the C-ABI, HIP events.   python tests/micro/slab_reduce_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import gnn_pressure_estimation_amd as G

lib = G._native.load()
P = int(lib.gatres_param_count(15, 32))
stride = (P + 3) // 4 * 4
out = torch.empty(P, device="cuda")
for S in (32, 64, 128, 256):
    sets = [torch.randn(S * stride, device="cuda") for _ in range(4)]      # rotate: the slabs of a step come from HBM / the Infinity Cache
    st = G._native.current_stream(out.device)
    for s in sets:
        G._native.check(lib.gatres_reduce_slabs(s.data_ptr(), S, stride, P, out.data_ptr(), st), "reduce")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 100
    e0.record()
    for r in range(reps):
        lib.gatres_reduce_slabs(sets[r % 4].data_ptr(), S, stride, P, out.data_ptr(), st)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    print(f"{S:4d} slabs x {P} parameters ({S * stride * 4 / 1e6:6.1f} MB): {us:7.2f} us per launch (eager, incl. ~2 us of host launch cost)")

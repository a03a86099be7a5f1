"""Micro-probe (not a test): the bf16 projection kernels of gatres_large timed alone through the C-ABI.

    python tests/micro/proj_probe.py [--lib path/to/lib.so] [--rows 49664] [--sets 12]

Times gatres_t_proj_attn_fwd (128 -> 2 x 128, 256 -> 1 x 128) and gatres_t_proj_bwd_dx (256 -> 128 with residual + ReLU
reference, 128 -> 256 with ReLU reference) with HIP events over launches that walk `--sets` different buffer sets (so that
neither L2 nor the Infinity Cache holds a launch's operands: the model's launches find theirs in HBM too), and prints
us per launch and the algorithmic GB/s (operands read once + result written once).
"""
import argparse
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--rows", type=int, default=49664)
    ap.add_argument("--sets", type=int, default=12)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--stamps", action="store_true")
    args = ap.parse_args()
    if args.lib:
        os.environ["GATRES_PROBE_LIB"] = os.path.abspath(args.lib)
    import gnn_pressure_estimation_amd as pkg
    lib = pkg._native.load_unchecked(args.lib) if args.lib else pkg._native.load()
    N, S = args.rows, args.sets
    dev = "cuda"
    bf = torch.bfloat16
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    x128 = [torch.randn(N, 128, device=dev).to(bf) for _ in range(S)]
    x256 = [torch.randn(N, 256, device=dev).to(bf) for _ in range(S)]
    o128 = [torch.empty(N, 128, device=dev, dtype=bf) for _ in range(S)]
    o256 = [torch.empty(N, 256, device=dev, dtype=bf) for _ in range(S)]
    W = (torch.randn(256, 128, device=dev) * 0.05).to(bf)          # [M][K] either way round (32 K elements)
    att = torch.randn(2, 256, device=dev)
    a_s = torch.empty(N, 2, device=dev)
    a_d = torch.empty(N, 2, device=dev)
    BF16 = 1

    def fwd(K, H, C, xs, os_):
        def f(k):
            rc = lib.gatres_t_proj_attn_fwd(p(xs[k]), p(W), p(att[0]), p(att[1]), p(os_[k]), p(a_s), p(a_d), N, K, H, C, BF16, st)
            assert rc == 0, rc
        return f

    def dx(K, HC, gs, resid, relu, os_):
        def f(k):
            rc = lib.gatres_t_proj_bwd_dx(p(gs[k]), p(W), p(resid[k]) if resid else None, p(relu[k]) if relu else None, p(os_[k]),
                                          N, K, HC, BF16, st)
            assert rc == 0, rc
        return f

    cases = [
        ("fwd proj1 128 -> 2x128 (EPI_ATT)", fwd(128, 2, 128, x128, o256), N * (128 + 256) * 2 + N * 16),
        ("fwd proj2 256 -> 1x128 (EPI_ATT)", fwd(256, 1, 128, x256, o128), N * (256 + 128) * 2 + N * 8),
        ("dX2 g_h2[128] -> g_o1[256] (ReLU ref)", dx(256, 128, x128, None, x256, o256), N * (128 + 256 + 256) * 2),
        ("dX1 g_h1[256] -> g_x[128] (resid + ReLU ref)", dx(128, 256, x256, x128, o128, [torch.empty_like(o) for o in o128]),
         N * (256 + 128 + 128 + 128) * 2),
    ]
    if args.stamps:
        # a library built with -DPROJ_STAMPS (tests/micro/build_probes.sh): per-wave wall-clock stamps of the last launch
        import numpy as np
        for name, f, nbytes in cases:
            for k in range(S):
                f(k)
            torch.cuda.synchronize()
            buf = (ctypes.c_ulonglong * (512 * 8 * 8))()
            lib_raw = ctypes.CDLL(os.environ["GATRES_PROBE_LIB"])
            assert lib_raw.gatres_probe_pstamps(buf) == 0
            t = np.frombuffer(buf, dtype=np.uint64).reshape(512, 8, 8).astype(np.int64)
            t = t[t[:, 0, 0] > 0]                      # workgroups of the last launch's grid (stamps are zeroed below)
            t0 = t[:, :, 0].min()
            us = lambda v: (v - t0) / 100.0
            print(name)
            labels = ["start", "stage 0 landed", "barrier", "stage 1 barrier", "computed", "-", "end"]
            for k in (0, 1, 2, 3, 4, 6):
                v = us(t[:, :, k])
                print(f"   {labels[k]:16s} min {v.min():6.2f}  mean {v.mean():6.2f}  max {v.max():6.2f} us")
        return
    for name, f, nbytes in cases:
        for k in range(S):
            f(k)
        torch.cuda.synchronize()
        # one hipGraph of S launches, replayed: the host's enqueue rate (ctypes + hipLaunchKernel) is out of the picture
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            st.value = torch.cuda.current_stream().cuda_stream
            for k in range(S):
                f(k)
        st.value = torch.cuda.current_stream().cuda_stream
        g.replay()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.reps):
            g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / (args.reps * S)
        print(f"{name:48s} {us:7.2f} us  {nbytes / us / 1e3:7.1f} GB/s")


if __name__ == "__main__":
    main()

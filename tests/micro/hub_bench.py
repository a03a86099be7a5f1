"""Hub rows in the per-op sparse kernels: times the five k_aggregate.hip kernels on a graph with a heavy-tailed degree
distribution (a few nodes with hundreds to thousands of incident edges) and on a uniform graph of the same size.
--lib <path> loads another build of the library (A/B against the pre-hub-path kernels; _native.load_unchecked).
    python tests/micro/hub_bench.py [--nodes 50000] [--edges 400000] [--hubs 32] [--hub-degree 1500]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import gnn_pressure_estimation_amd as G                      # noqa: E402

if "--lib" in sys.argv:
    G._native.load_unchecked(sys.argv[sys.argv.index("--lib") + 1])
    del sys.argv[sys.argv.index("--lib"):sys.argv.index("--lib") + 2]
from tests import hipops as ops                               # noqa: E402


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nodes", type=int, default=50000)
    ap.add_argument("--edges", type=int, default=400000)
    ap.add_argument("--hubs", type=int, default=32)
    ap.add_argument("--hub-degree", type=int, default=1500)
    ap.add_argument("--nc", type=int, default=32)
    a = ap.parse_args()
    g = torch.Generator().manual_seed(0)
    n, C, H = a.nodes, a.nc, 2
    res = {"lib": G._native.lib_path()}
    for kind in ("uniform", "hubs"):
        src = torch.randint(0, n, (a.edges,), generator=g)
        dst = torch.randint(0, n, (a.edges,), generator=g)
        if kind == "hubs":                                  # the same edge count: hub edges replace random ones
            k = a.hubs * a.hub_degree
            hubs = torch.randperm(n, generator=g)[:a.hubs]
            dst[:k] = hubs.repeat_interleave(a.hub_degree)
            src[k:2 * k] = hubs.repeat_interleave(a.hub_degree)
        keep = src != dst
        ei = torch.stack([src[keep], dst[keep]])
        plan = G.GraphPlan(ei, n, device="cuda", reorder=False)
        h = torch.randn(n, H * C, generator=g).cuda()
        a_s, a_d = torch.randn(n, H, generator=g).cuda(), torch.randn(n, H, generator=g).cuda()
        att_s, att_d, bias = (torch.randn(H * C, generator=g).cuda() for _ in range(3))
        out, alpha = ops.gat_aggregate_fwd(plan, h, a_s, a_d, bias, H, relu=True)
        g_pre = torch.randn(n, H * C, generator=g).cuda()
        y, x0 = torch.randn(n, C, generator=g).cuda(), torch.randn(n, C, generator=g).cuda()
        r = {"max_in_degree": int(torch.bincount(ei[1], minlength=n).max()),
             "gat_aggregate_fwd_us": timeit(lambda: ops.gat_aggregate_fwd(plan, h, a_s, a_d, bias, H, relu=True)),
             "gat_aggregate_bwd_us": timeit(lambda: ops.gat_aggregate_bwd(plan, g_pre, h, alpha, a_s, a_d, att_s, att_d, H)),
             "mean_fwd_us": timeit(lambda: ops.mean_residual_relu_fwd(plan, y, x0)),
             "mean_bwd_us": timeit(lambda: ops.mean_bwd(plan, y))}
        res[kind] = r
    print(json.dumps(res))


if __name__ == "__main__":
    main()

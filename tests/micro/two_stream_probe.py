"""Micro-probe (not a test): does a per-op step chain leave room for a second one beside it?

    python tests/micro/two_stream_probe.py [--model gatres_large] [--batch-size 128] [--dtype bf16] [--parts 2]

Runs the per-op training step of a batch (a) as ONE trainer on the whole batch and (b) as `--parts` independent trainers on
equal shares of it, each replaying its captured step on its OWN stream (two separate models here: the probe only asks how two
launch chains share the chip -- it does not combine gradients).  Prints ms per (whole-batch) step both ways.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="gatres_large")
    ap.add_argument("--batch-size", type=int, default=128)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--parts", type=int, default=2)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--nodes", type=int, default=388)
    ap.add_argument("--pipes", type=int, default=430)
    args = ap.parse_args()
    import gnn_pressure_estimation_amd as G
    dev = torch.device("cuda")
    nb, nc = (25, 128) if args.model == "gatres_large" else (15, 32)
    topo = G.wdn_synth.make_wdn_topology(args.nodes, args.pipes, seed=0)

    def make(bs, seed):
        torch.manual_seed(seed)
        m = G.GATResMeanConv(name=args.model, num_blocks=nb, nc=nc).to(dev)
        if args.dtype == "bf16":
            m.set_compute_dtype("bf16")
        ei = G.wdn_synth.collate_edge_index(topo, args.nodes, bs).to(dev)
        tr = G.GATResTrainer(m, ei, args.nodes * bs, nodes_per_graph=[args.nodes] * bs, seed=seed, use_graph=True, fused=False,
                             targets_are_inputs=True)
        x = G.wdn_synth.make_snapshots(bs, args.nodes, seed=seed).to(dev).reshape(-1).contiguous()
        return tr, x

    def run(trainers, streams):
        for _ in range(3):
            for (tr, x), s in zip(trainers, streams):
                with torch.cuda.stream(s):
                    tr.step(x, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            for (tr, x), s in zip(trainers, streams):
                with torch.cuda.stream(s):
                    tr.step(x, x)
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / args.steps * 1e3

    whole = [make(args.batch_size, 1)]
    ms1 = run(whole, [torch.cuda.Stream()])
    print(f"one chain, {args.batch_size} snapshots: {ms1:.3f} ms/step")
    del whole
    torch.cuda.empty_cache()
    per = args.batch_size // args.parts
    parts = [make(per, 10 + k) for k in range(args.parts)]
    msp = run(parts, [torch.cuda.Stream() for _ in parts])
    print(f"{args.parts} chains of {per} snapshots on {args.parts} streams: {msp:.3f} ms per {per * args.parts} snapshots")
    ms_seq = run(parts, [torch.cuda.Stream()] * args.parts)
    print(f"{args.parts} chains of {per} snapshots on ONE stream: {ms_seq:.3f} ms per {per * args.parts} snapshots")


if __name__ == "__main__":
    main()

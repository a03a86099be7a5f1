"""Regenerates the golden vectors under tests/golden/ from the CPU oracle (oracle/gatres_oracle.py).

    python tests/golden/make_golden.py

The reference ships no vectors of its own for this path and torch_geometric cannot be
installed here, so these pin HIP-vs-oracle parity and guard the oracle against silent edits -- not PyG parity.
Each file holds inputs AND expected outputs of one reference training iteration (train.py:159-190):
  edge_index, x (= y before masking), mask, params (flat, state_dict order) ->
  out (predictions), loss, grads (flat), params_after (one torch.optim.Adam step, lr 5e-4, wd 6e-6).
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import gatres_oracle as O  # noqa: E402
import gnn_pressure_estimation_amd as G  # noqa: E402

CASES = {
    # name: (num_blocks, nc, batch, nodes, pipes, param seed)
    "tiny_nb2_nc8": (2, 8, 3, 40, 47, 7),
    "ctown_small_bs2": (15, 32, 2, 388, 430, 3),
}


def make(name):
    nb, nc, bs, nodes, pipes, seed = CASES[name]
    x, y, ei, mask = G.wdn_synth.make_batch(bs, nodes, pipes)
    p = O.init_params(nb, nc, seed=seed)
    tr = O.OracleTrainer(p)
    loss, out = tr.step(x.clone(), y, ei, mask)
    return dict(num_blocks=nb, nc=nc, nodes_per_graph=nodes, edge_index=ei.numpy(), x=x.numpy(), mask=mask.numpy(),
                params=O.flatten(p).numpy(), out=out.numpy(), loss=np.float32(loss.item()),
                grads=tr.flat("grads").numpy(), params_after=tr.flat("params").numpy())


if __name__ == "__main__":
    torch.set_num_threads(8)
    for name in CASES:
        d = make(name)
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), name + ".npz")
        np.savez_compressed(path, **d)
        print(path, os.path.getsize(path), "bytes; loss", float(d["loss"]))

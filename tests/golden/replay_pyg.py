"""Closes the PyG-parity gap on a machine that HAS torch_geometric (this repo's build container does not: no network).

    python tests/golden/replay_pyg.py --reference /path/to/gnn-pressure-estimation [--tol 1e-5] [--fixtures a.npz b.npz]

For every golden fixture under tests/golden/ (inputs + the CPU oracle's outputs of ONE reference training iteration,
written by make_golden.py) this script
  1. imports the REFERENCE's own model class -- ``GATResMeanConv`` from ``<reference>/gnn_pressure_estimation/GraphModels.py``
     (:471-494, built on torch_geometric's GATConv / SimpleConv / Linear; imported from where it lies, never copied) --
     and constructs it exactly as the registry does (``ConfigModels.py:30-32,40-42``: name, num_blocks, nc);
  2. loads the fixture's flat parameter vector into its ``state_dict`` (key spellings below);
  3. replays the fixture's batch through the reference's loop body (``train.py:159-190``: ``x[mask] = 0`` ->
     ``model(x, edge_index, None, None)`` -> ``MSELoss(out[mask], y[mask])`` -> ``backward`` ->
     ``torch.optim.Adam(lr=5e-4, weight_decay=6e-6).step()``);
  4. compares predictions, loss, the flat gradient and the parameters after the step with the oracle's numbers stored in
     the fixture, prints the relative errors, writes ``<fixture>.pyg.npz`` next to the fixture (PyG's out / loss / grads /
     params_after, with the torch / torch_geometric versions) and exits non-zero if any error exceeds ``--tol``.

A green run turns "parity unpinned"  into "the oracle is pinned against PyG x.y.z"; commit the
``*.pyg.npz`` files then, and tests/test_host_logic.py::test_oracle_matches_pyg_replay (skipped while none exists) holds
the oracle to them from there on.

state_dict spellings: PyG 2.3 / 2.4 GATConv registers ``lin_src`` and ``lin_dst`` (the SAME Linear object when
``in_channels`` is an int, as here: both keys must be given, with the same tensor); PyG >= 2.5 registers a single ``lin``.
The fixture stores ``lin_src.weight``; this script writes whichever keys the installed PyG's module actually has.
``torch_scatter`` is imported by GraphModels.py:9 but used only by GENConvolution (out of scope): a stub module is
installed in sys.modules if it is missing, so that the import of the file succeeds.
"""
import argparse
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def param_shapes(nb, nc):
    """state_dict order of the fixture's flat vector (oracle/gatres_oracle.py: param_shapes)."""
    from oracle import gatres_oracle as O
    return O.param_shapes(nb, nc)


def load_reference_model(reference_root, nb, nc):
    try:
        import torch_geometric  # noqa: F401
    except ImportError:
        raise SystemExit("torch_geometric is not installed here: run this script on a machine that has it (any PyG >= 2.3)")
    if "torch_scatter" not in sys.modules:
        try:
            import torch_scatter  # noqa: F401
        except ImportError:
            stub = types.ModuleType("torch_scatter")
            stub.scatter = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("torch_scatter stub: GENConvolution only"))
            sys.modules["torch_scatter"] = stub
    pkg_dir = os.path.join(reference_root, "gnn_pressure_estimation")
    if not os.path.isfile(os.path.join(pkg_dir, "GraphModels.py")):
        raise SystemExit(f"{pkg_dir}/GraphModels.py not found: --reference must be the root of the reference checkout")
    sys.path.insert(0, reference_root)
    sys.path.insert(0, pkg_dir)                       # (the reference mixes script-relative and package-absolute imports)
    import importlib
    GM = importlib.import_module("gnn_pressure_estimation.GraphModels")
    name = "GATResMeanConv"
    return GM.GATResMeanConv(name=name, num_blocks=nb, nc=nc)


def fill_state_dict(model, flat, nb, nc):
    sd = model.state_dict()
    new, off = {}, 0
    for key, shape in param_shapes(nb, nc).items():
        n = int(np.prod(shape))
        t = torch.from_numpy(flat[off:off + n].copy()).reshape(shape)
        off += n
        if key.endswith("lin_src.weight"):
            base = key[:-len("lin_src.weight")]
            hit = False
            for spelling in ("lin_src.weight", "lin_dst.weight", "lin.weight"):
                if base + spelling in sd:
                    new[base + spelling] = t
                    hit = True
            if not hit:
                raise SystemExit(f"this torch_geometric's GATConv has none of lin_src / lin_dst / lin under {base!r}: {list(sd)[:12]}")
        else:
            if key not in sd:
                raise SystemExit(f"state_dict key {key!r} missing in the reference model (has: {list(sd)[:12]} ...)")
            new[key] = t
    assert off == flat.size
    missing = set(sd) - set(new)
    if missing:
        raise SystemExit(f"reference parameters the fixture does not cover: {sorted(missing)}")
    model.load_state_dict(new)


def flat_from(model, nb, nc, grads=False):
    """Flat vector in the FIXTURE's order from the reference module (lin_src / lin spellings folded back)."""
    named = dict(model.named_parameters())
    out = []
    for key in param_shapes(nb, nc):
        k = key
        if k not in named and key.endswith("lin_src.weight"):
            for spelling in ("lin_dst.weight", "lin.weight"):
                if key[:-len("lin_src.weight")] + spelling in named:
                    k = key[:-len("lin_src.weight")] + spelling
        t = named[k].grad if grads else named[k].data
        out.append(t.detach().reshape(-1))
    return torch.cat(out).numpy()


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64).ravel(), np.asarray(b, dtype=np.float64).ravel()
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def replay(path, reference_root, tol, out_dir=None, versions=None):
    """``out_dir``: where ``<fixture>.pyg.npz`` goes (default: beside the fixture); ``versions``: (torch_geometric version
    string) when the model does not come from an installed torch_geometric -- the CPU suite's self-test of this script runs
    it over a stand-in module (tests/test_host_logic.py)."""
    d = np.load(path)
    nb, nc = int(d["num_blocks"]), int(d["nc"])
    model = load_reference_model(reference_root, nb, nc)
    fill_state_dict(model, d["params"], nb, nc)
    model.train()
    x = torch.from_numpy(d["x"]).clone()
    y = torch.from_numpy(d["x"]).clone()                       # snapshots: y == x before masking (train.py:162-166)
    ei = torch.from_numpy(d["edge_index"])
    mask = torch.from_numpy(d["mask"]).bool()
    opt = torch.optim.Adam(model.parameters(), lr=5e-4, weight_decay=6e-6)      # train.py:348
    opt.zero_grad()
    x[mask] = 0                                                # train.py:174
    out = model(x, ei, None, None)                             # train.py:175 (batch and edge_attr are None for gatres)
    loss = torch.nn.MSELoss()(out[mask], y[mask])              # train.py:177-183, criterion at :364
    loss.backward()
    grads = flat_from(model, nb, nc, grads=True)
    opt.step()
    after = flat_from(model, nb, nc)
    if versions is None:
        import torch_geometric
        versions = str(torch_geometric.__version__)
    errs = {"out": rel(out.detach().numpy(), d["out"]), "loss": rel(float(loss.detach()), float(d["loss"])),
            "grads": rel(grads, d["grads"]), "params_after_abs": float(np.abs(after - d["params_after"]).max())}
    target = os.path.join(out_dir, os.path.basename(path)[:-4] + ".pyg.npz") if out_dir else path[:-4] + ".pyg.npz"
    np.savez_compressed(target, out=out.detach().numpy(), loss=np.float32(float(loss.detach())), grads=grads,
                        params_after=after, torch_version=str(torch.__version__),
                        torch_geometric_version=versions)
    ok = errs["out"] < tol and errs["loss"] < tol and errs["grads"] < 10 * tol and errs["params_after_abs"] < 1e-5
    print(f"{os.path.basename(path)}: PyG {versions} vs oracle: out {errs['out']:.2e}  loss {errs['loss']:.2e}  "
          f"grads {errs['grads']:.2e}  |params_after| {errs['params_after_abs']:.2e}  ->  {'OK' if ok else 'MISMATCH'}")
    return ok


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--reference", default="/root/reference", help="root of the DiTEC-project/gnn-pressure-estimation checkout")
    ap.add_argument("--tol", type=float, default=1e-5, help="relative tolerance on predictions / loss (north star: 1e-5)")
    ap.add_argument("--fixtures", nargs="*", default=None)
    ap.add_argument("--out-dir", default=None, help="write the *.pyg.npz files here instead of beside the fixtures")
    args = ap.parse_args(argv)
    torch.manual_seed(0)
    paths = args.fixtures or sorted(os.path.join(HERE, f) for f in os.listdir(HERE)
                                    if f.endswith(".npz") and not f.endswith(".pyg.npz") and "wdn" not in f)
    ok = all([replay(p, args.reference, args.tol, out_dir=args.out_dir) for p in paths])
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()

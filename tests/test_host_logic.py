"""CPU-side tests: the oracle against the committed golden vectors, the host plan builder, the C-ABI export list
against include/gatres.h, and the nn.Module surface.  No GPU, no kernel launches."""
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def unflatten(oracle, flat, nb, nc):
    out, off = {}, 0
    for k, shp in oracle.param_shapes(nb, nc).items():
        n = int(np.prod(shp))
        out[k] = torch.from_numpy(flat[off:off + n].copy()).reshape(shp)
        off += n
    assert off == flat.size
    return out


@pytest.mark.parametrize("name", ["tiny_nb2_nc8", "ctown_small_bs2"])
def test_oracle_reproduces_golden_vectors(oracle, name):
    d = np.load(os.path.join(GOLDEN, name + ".npz"))
    nb, nc = int(d["num_blocks"]), int(d["nc"])
    p = unflatten(oracle, d["params"], nb, nc)
    tr = oracle.OracleTrainer(p)
    x = torch.from_numpy(d["x"])
    loss, out = tr.step(x.clone(), x, torch.from_numpy(d["edge_index"]), torch.from_numpy(d["mask"]))
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    assert rel(out.numpy(), d["out"]) < 2e-6          # thread-count dependent reduction order only
    assert abs(float(loss) - float(d["loss"])) < 1e-5 * abs(float(d["loss"]))
    assert rel(tr.flat("grads").numpy(), d["grads"]) < 1e-5
    assert np.abs(tr.flat("params").numpy() - d["params_after"]).max() <= 2 * 5e-4      # Adam: |dp| <= ~lr per step


def test_c_abi_exports_every_declared_symbol(pkg, lib):
    """include/gatres.h is the contract: every function it declares must be exported by the built library and bound
    by the ctypes table (and vice versa)."""
    hdr = open(os.path.join(ROOT, "include", "gatres.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(gatres_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in gatres.h but not exported"
    assert declared == set(pkg._native.SIGNATURES), declared ^ set(pkg._native.SIGNATURES)
    assert lib.gatres_param_count(15, 32) == 65857 and lib.gatres_param_count(25, 128) == 1667585


def ref_plan(ei, n):
    """numpy restatement of the plan: PyG edge order, stable sorts."""
    src, dst = ei[0].numpy(), ei[1].numpy()
    keep = src != dst
    gs = np.concatenate([src[keep], np.arange(n)])
    gd = np.concatenate([dst[keep], np.arange(n)])
    order = np.argsort(gd, kind="stable")
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(gd, minlength=n))])
    col = gs[order]
    pos = np.empty_like(order); pos[order] = np.arange(order.size)
    torder = np.argsort(gs, kind="stable")
    t_rowptr = np.concatenate([[0], np.cumsum(np.bincount(gs, minlength=n))])
    morder = np.argsort(dst, kind="stable")
    m_rowptr = np.concatenate([[0], np.cumsum(np.bincount(dst, minlength=n))])
    mtorder = np.argsort(src, kind="stable")
    return dict(rowptr=rowptr, col=col, t_rowptr=t_rowptr, t_eid=pos[torder], t_dst=gd[torder], m_rowptr=m_rowptr,
                m_col=src[morder], mt_rowptr=np.concatenate([[0], np.cumsum(np.bincount(src, minlength=n))]),
                mt_dst=dst[mtorder])


def test_graph_plan_matches_numpy_restatement(pkg):
    g = torch.Generator().manual_seed(0)
    n = 57
    ei = torch.randint(0, n, (2, 300), generator=g)          # self loops and duplicates included
    plan = pkg.GraphPlan(ei, n, device="cpu", reorder=False)
    ref = ref_plan(ei, n)
    assert plan.num_edges_gat == int((ei[0] != ei[1]).sum()) + n and plan.num_edges_mean == 300
    for k, v in ref.items():
        got = plan.arrays[k].numpy()[: v.size]
        assert np.array_equal(got, v), k
    with pytest.raises(RuntimeError):
        pkg.GraphPlan(torch.tensor([[0, 99], [1, 2]]), 5, device="cpu")       # endpoint out of range
    with pytest.raises(ValueError):
        pkg.GraphPlan(torch.zeros((2, 3), dtype=torch.int32), 5, device="cpu")


def test_segments_are_the_snapshots_of_a_batch(pkg):
    one = pkg.wdn_synth.make_wdn_topology()
    plan = pkg.GraphPlan(pkg.wdn_synth.collate_edge_index(one, 388, 5), 388 * 5, device="cpu")
    assert plan.num_segments == 5 and plan.max_segment_nodes == 388
    assert plan.segment_ptr_host.tolist() == [0, 388, 776, 1164, 1552, 1940]
    assert plan.max_segment_edges_mean == 860 and plan.max_segment_edges_gat == 860 + 388
    # an edge across two snapshots fuses them into one segment; isolated nodes are merged up to merge_upto
    ei = torch.cat([pkg.wdn_synth.collate_edge_index(one, 388, 3), torch.tensor([[10], [400]])], 1)
    p2 = pkg.GraphPlan(ei, 388 * 3, device="cpu")
    assert p2.segment_ptr_host.tolist() == [0, 776, 1164]
    p3 = pkg.GraphPlan(torch.zeros((2, 0), dtype=torch.int64), 10, device="cpu", merge_upto=4)
    assert p3.segment_ptr_host.tolist() == [0, 4, 8, 10]
    assert pkg.GraphPlan(one, 388, device="cpu", segments=False).num_segments == 0


def test_row_windows_of_split_segments(pkg):
    """gatres_graph_windows_host against a direct restatement: for 2 / 4 / 8 parts per segment, the largest contiguous
    row range that holds a part's own rows (16-row tiles dealt out evenly) and every row adjacent to them, and the
    in-edge counts of that range (GATConv: self loops removed, one added per node; SimpleConv: as given)."""
    from gnn_pressure_estimation_amd.graph_plan import GraphPlan
    t1 = pkg.wdn_synth.make_wdn_topology(388, 430, seed=0)
    t2 = pkg.wdn_synth.make_wdn_topology(100, 120, seed=3)
    ei = torch.cat([t1, t2 + 388, t1 + 488], dim=1)
    N = 388 + 100 + 388
    plan = GraphPlan(ei, N, torch.device("cpu"))
    assert plan.num_segments == 3
    seg = plan.segment_ptr_host.numpy()
    src, dst = ei[0].numpy(), ei[1].numpy()
    indeg_gat = np.bincount(dst[src != dst], minlength=N) + 1
    indeg_all = np.bincount(dst, minlength=N)
    got = np.array(plan.windows[:21]).reshape(7, 3)
    halo = plan.windows[21:28]                # most edges of one part that cross to another part (in- or out-), per M
    for k, M in enumerate(range(2, 9)):
        best = np.zeros(3, dtype=np.int64)
        hin, hout = np.zeros((3, M), dtype=np.int64), np.zeros((3, M), dtype=np.int64)
        for s in range(3):
            a, b = int(seg[s]), int(seg[s + 1])
            n = b - a
            tiles = (n + 15) // 16
            bounds = [min(n, 16 * (tiles * p // M)) for p in range(M + 1)]
            part = np.searchsorted(np.array(bounds[1:]), np.arange(n), side="right")
            wlo, whi = list(bounds[:-1]), list(bounds[1:])
            for u, v in zip(src, dst):
                if a <= u < b and u != v:
                    pu, pv = part[u - a], part[v - a]
                    if pu != pv:
                        hout[s, pu] += 1; hin[s, pv] += 1
                        wlo[pu] = min(wlo[pu], v - a); whi[pu] = max(whi[pu], v - a + 1)
                        wlo[pv] = min(wlo[pv], u - a); whi[pv] = max(whi[pv], u - a + 1)
            for p in range(M):
                rows = whi[p] - wlo[p]
                best = np.maximum(best, [rows, indeg_gat[a + wlo[p]:a + whi[p]].sum(), indeg_all[a + wlo[p]:a + whi[p]].sum()])
        assert list(got[k]) == list(best), (M, got[k], best)
        assert halo[k] == max(hin.max(), hout.max()), (M, halo[k], hin.max(), hout.max())
    # a locality-preserving order keeps the windows a fraction of the segment; a shuffled one does not
    assert got[2][0] < 388 // 2 and plan.window_rows(4) == got[2][0] and plan.halo_edges(4) == halo[2]
    perm = torch.from_numpy(np.random.RandomState(1).permutation(388))
    shuf = perm[t1]
    assert GraphPlan(shuf, 388, torch.device("cpu"), reorder=False).window_rows(4) > 300


def test_reordered_plan_restores_compact_windows(pkg):
    """gatres_graph_reorder_host (reverse Cuthill-McKee inside every segment): a randomly relabelled C-Town-sized batch
    gets its compact row windows back, nodes never leave their snapshot, the plan's arrays are exactly the plan of the
    relabelled graph (so every row keeps PyG's edge order), and a locality-preserving input is left alone."""
    from gnn_pressure_estimation_amd.graph_plan import GraphPlan
    t1 = pkg.wdn_synth.make_wdn_topology(388, 430, seed=0)
    bs = 3
    rs = np.random.RandomState(7)
    relabel = torch.from_numpy(np.concatenate([rs.permutation(388) + 388 * k for k in range(bs)]))
    ei = relabel[pkg.wdn_synth.collate_edge_index(t1, 388, bs)]
    N = 388 * bs
    plan = GraphPlan(ei, N, torch.device("cpu"))                       # automatic: adopted, the windows shrink > 10 %
    assert plan.perm_host is not None and plan.num_segments == bs
    perm = plan.perm_host.long()
    assert sorted(perm.tolist()) == list(range(N))
    for k in range(bs):                                                # a permutation of every snapshot's own id range
        assert sorted(perm[388 * k:388 * (k + 1)].tolist()) == list(range(388 * k, 388 * (k + 1)))
    assert plan.window_rows(4) < 388 // 2 and plan.window_rows(4) <= 1.15 * GraphPlan(
        pkg.wdn_synth.collate_edge_index(t1, 388, bs), N, torch.device("cpu"), reorder=False).window_rows(4)
    old2new = torch.empty(N, dtype=torch.int64)
    old2new[perm] = torch.arange(N)
    ref = ref_plan(old2new[ei], N)                                     # same edge ORDER, relabelled endpoints
    for k, v in ref.items():
        assert np.array_equal(plan.arrays[k].numpy()[: v.size], v), k
    assert plan.c.perm == plan.arrays["perm"].data_ptr()
    # the generator's own order is already local: automatic mode keeps the identity (no indirection in the kernels)
    keep = GraphPlan(pkg.wdn_synth.collate_edge_index(t1, 388, bs), N, torch.device("cpu"))
    assert keep.perm_host is None and not keep.c.perm
    forced = GraphPlan(pkg.wdn_synth.collate_edge_index(t1, 388, bs), N, torch.device("cpu"), reorder=True)
    assert forced.perm_host is not None


def test_module_surface_on_cpu(pkg, oracle):
    m = pkg.GATResMeanConv(name="GATResMeanConv_small_znorm_15b_32c", num_blocks=15, nc=32)
    assert sum(p.numel() for p in m.parameters()) == 65857
    assert m.flat_parameters.numel() == 65857 and m._flat_is_current()
    keys = [k for k in m.state_dict() if "lin_dst" not in k]
    assert keys == list(oracle.param_shapes(15, 32).keys())
    # parameters are views of the flat vector in state_dict order
    m.lin1.bias.data.fill_(3.5)
    assert float(m.flat_parameters[-1]) == 3.5
    with pytest.raises(ValueError):
        m(torch.zeros(4, 1), torch.zeros((2, 0), dtype=torch.int64))       # CPU tensors: no CPU path
    import argparse
    args, model = pkg.select_model(argparse.Namespace(model="gatres_large"))
    assert (model.num_blocks, model.nc, args.criterion, args.norm_type) == (25, 128, "mse", "znorm")
    assert args.use_data_edge_attrs is None and model.name == "GATRes_Large_znorm_25b_128c"
    with pytest.raises(NotImplementedError):
        pkg.select_model(argparse.Namespace(model="gin"))


def test_synthetic_wdn_has_ctown_shape(pkg):
    ei = pkg.wdn_synth.make_wdn_topology()
    assert ei.shape == (2, 860) and ei.dtype == torch.int64
    assert torch.equal(ei, pkg.wdn_synth.make_wdn_topology())                # seeded
    und = {(min(a, b), max(a, b)) for a, b in ei.t().tolist()}
    assert len(und) == 430 and all(a != b for a, b in und)
    assert torch.equal(ei[0], torch.sort(ei[0], stable=True).values)         # grouped by source (from_networkx order)
    deg = torch.bincount(ei[1], minlength=388)
    assert int(deg.min()) >= 1 and int(deg.max()) <= 5
    x, y, bei, mask = pkg.wdn_synth.make_batch(4)
    assert x.shape == (1552, 1) and torch.equal(x, y) and bei.shape == (2, 3440) and int(bei.max()) == 1551
    assert mask.dtype == torch.bool and all(int(mask[i * 388:(i + 1) * 388].sum()) == 368 for i in range(4))


def test_graph_flags_symmetric(pkg, lib):
    """GATRES_GRAPH_SYMMETRIC: set for an undirected network listed in both directions (pgu.from_networkx), not for a
    directed edge list; self loops and duplicate edges do not matter."""
    import ctypes as C
    t = pkg.wdn_synth.make_wdn_topology(60, 70, seed=2)

    def flags(ei):
        ei = ei.contiguous()
        f = C.c_int32(-1)
        assert lib.gatres_graph_flags_host(ei.data_ptr(), ei.shape[1], 60, C.byref(f)) == 0
        return f.value

    SYM, LE32, LE6 = 1, 2, 4  # GATRES_GRAPH_SYMMETRIC, GATRES_GRAPH_DEG_LE32 (every water network has low degrees), _DEG_LE6
    both = SYM | LE32
    assert flags(t) & both == both
    assert flags(torch.cat([t, t[:, :5], torch.tensor([[3, 7], [3, 7]])], dim=1)) & both == both      # duplicates + self loops
    assert flags(t[:, 1:]) & both == LE32                                                  # one direction missing
    assert flags(t[:, t[0] < t[1]]) & both == LE32
    # GATRES_GRAPH_DEG_LE6: no node with more than 5 edges into or out of it -- every row of every CSR of the plan has at most 6
    # entries (GATConv's self loop included), the slot width of the per-snapshot kernels (k_window.hip: the instantiation without
    # edge-at-a-time paths).  Counted on edge_index as it is (self loops and duplicates count: never optimistic).
    deg = max(int(torch.bincount(t[0], minlength=60).max()), int(torch.bincount(t[1], minlength=60).max()))
    assert deg <= 5 and flags(t) & LE6 == LE6
    star = lambda k: torch.stack([torch.arange(1, k + 1), torch.zeros(k, dtype=torch.long)])   # k edges INTO node 0
    assert flags(star(5)) & LE6 == LE6 and flags(star(6)) & LE6 == 0 and flags(star(6).flip(0)) & LE6 == 0
    assert flags(torch.cat([star(5), torch.tensor([[0], [0]])], dim=1)) & LE6 == 0            # (a self loop counts)
    # GATRES_GRAPH_DEG_LE32: no node with more than 31 edges into or out of it (a row of 32 with GATConv's self loop)
    hub = torch.stack([torch.arange(1, 33), torch.zeros(32, dtype=torch.long)])                # 32 edges INTO node 0
    assert flags(torch.cat([t, hub], dim=1)) & LE32 == 0
    assert flags(torch.cat([t, hub.flip(0)], dim=1)) & LE32 == 0                               # ... and OUT of it
    assert flags(torch.cat([t[:, t[1] != 0][:, :0], hub[:, :31]], dim=1)) & LE32 == LE32      # 31: still a slot / loop row


@pytest.mark.parametrize("m", [2, 4, 8])
def test_part_tables_match_numpy_restatement(pkg, lib, m):
    """gatres_graph_part_tables_host against a direct numpy restatement of what the window kernel's prologues derive:
    row window, edge ranges, padded edge descriptors, de-duplicated hand-off lists (a ragged batch, one shuffled graph)."""
    import ctypes as C
    t_big = pkg.wdn_synth.make_wdn_topology(388, 430, seed=0)
    perm = torch.from_numpy(np.random.RandomState(3).permutation(100))
    t_shuf = perm[pkg.wdn_synth.make_wdn_topology(100, 120, seed=2)]
    ei = torch.cat([t_big, t_shuf + 388], dim=1)
    N = 488
    plan = pkg.GraphPlan(ei, N, device="cpu", reorder=False)
    assert plan.num_segments == 2
    H = {k: v.numpy() for k, v in plan._host.items()}
    seg = plan.segment_ptr_host.numpy()
    ptrs = [plan._host[k].data_ptr() for k in ("rowptr", "col", "t_rowptr", "t_eid", "t_dst", "m_rowptr", "m_col",
                                               "mt_rowptr", "mt_dst")]
    stride = C.c_int64(0)
    assert lib.gatres_graph_part_tables_host(*ptrs, plan.segment_ptr_host.data_ptr(), 2, m, None, 0, C.byref(stride)) == 0
    st = stride.value
    assert st % 4 == 0 and st > 48
    words = torch.zeros(2 * m * st, dtype=torch.int32)
    assert lib.gatres_graph_part_tables_host(*ptrs, plan.segment_ptr_host.data_ptr(), 2, m, words.data_ptr(), st,
                                             C.byref(stride)) == 0
    w = words.numpy()
    for s in range(2):
        n0, n = int(seg[s]), int(seg[s + 1] - seg[s])
        tiles = (n + 15) // 16
        e0 = int(H["rowptr"][n0])
        for p in range(m):
            rec = w[(s * m + p) * st:(s * m + p + 1) * st]
            u16 = rec.view(np.uint16)
            lo, hi = 16 * (tiles * p // m), min(n, 16 * (tiles * (p + 1) // m))
            assert rec[0] == 0x47545031 and rec[1] == m and rec[2] == n0 and rec[3] == n and rec[8] == lo and rec[9] == hi
            rows = np.arange(n0 + lo, n0 + hi)
            nbr = set(range(lo, hi))
            for ptr, idx in (("rowptr", "col"), ("t_rowptr", "t_dst"), ("m_rowptr", "m_col"), ("mt_rowptr", "mt_dst")):
                for r in rows:
                    nbr.update(int(j) - n0 for j in H[idx][H[ptr][r]:H[ptr][r + 1]])
            if not nbr:                                    # (more parts than 16-row tiles: an empty part)
                assert rec[10] == lo and rec[11] == hi
                continue
            assert rec[10] == min(nbr) and rec[11] == max(nbr) + 1                  # the row window
            elo = int(H["rowptr"][n0 + lo]) - e0
            assert rec[12] == elo and rec[13] == int(H["rowptr"][n0 + hi]) - e0 - elo
            # in-edge descriptors of the forward image
            f_img = int(rec[22]) * 2
            ow = hi - lo
            ev = lambda v: (v + 1) & ~1
            oeg, oem = int(rec[13]), int(rec[17])
            off = ev(ow + 1) + ev(oeg) + ev(ow + 1) + ev(oem)
            off = (off + 7) & ~7
            nbin = u16[f_img + off:f_img + off + 8 * ow].reshape(ow, 8)
            for r in range(ow):
                srcs = H["col"][H["rowptr"][n0 + lo + r]:H["rowptr"][n0 + lo + r + 1]] - n0
                assert nbin[r, 0] == H["rowptr"][n0 + lo + r] - e0 - elo and nbin[r, 1] == len(srcs)
                for k in range(6):
                    assert nbin[r, 2 + k] == srcs[min(k, len(srcs) - 1)]
            # forward hand-off lists: unique remote sources of own in-edges; own rows with an out-edge to a remote row
            srcs = H["col"][H["rowptr"][n0 + lo]:H["rowptr"][n0 + hi]] - n0
            want = sorted(set(int(j) for j in srcs if j < lo or j >= hi))
            got = u16[int(rec[24]) * 2:int(rec[24]) * 2 + int(rec[25])]
            assert list(got) == want
            want_e = [r - n0 for r in rows
                      if any((j - n0 < lo or j - n0 >= hi) for j in H["t_dst"][H["t_rowptr"][r]:H["t_rowptr"][r + 1]])]
            got_e = u16[int(rec[26]) * 2:int(rec[26]) * 2 + int(rec[27])]
            assert list(got_e) == want_e
            # symmetric graph: what a part imports in the forward pass is what its partners export, and vice versa
            assert int(rec[31]) == len(want)          # backward import rows = forward import rows on a symmetric graph


@pytest.mark.parametrize("name", ["tiny_nb2_nc8", "ctown_small_bs2"])
def test_oracle_matches_pyg_replay(oracle, name):
    """Holds the oracle to PyG's own numbers once somebody with torch_geometric has run tests/golden/replay_pyg.py and
    committed its ``<fixture>.pyg.npz`` (the reference's GATResMeanConv on the fixture's batch).  Skipped until then:
    parity stays UNPINNED."""
    pyg = os.path.join(GOLDEN, name + ".pyg.npz")
    if not os.path.exists(pyg):
        pytest.skip("no PyG replay committed (tests/golden/replay_pyg.py needs a machine with torch_geometric)")
    d, r = np.load(os.path.join(GOLDEN, name + ".npz")), np.load(pyg)
    rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())
    assert rel(d["out"], r["out"]) < 1e-5 and abs(float(d["loss"]) - float(r["loss"])) < 1e-5 * abs(float(r["loss"]))
    assert rel(d["grads"], r["grads"]) < 1e-4


# ---------------------------------------------------------------------------------------------------------------------
# tests/golden/replay_pyg.py is the one committed way to pin the oracle against PyG; it has never run where PyG exists.
# So that it is known to WORK the day somebody has torch_geometric: its argument parsing, its state_dict mapping for both
# PyG spellings and its whole replay loop run here over stand-in modules that have PyG's parameter names.
# ---------------------------------------------------------------------------------------------------------------------
def _replay_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("replay_pyg", os.path.join(GOLDEN, "replay_pyg.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


class _StandInConv(torch.nn.Module):
    """Parameter names of torch_geometric.nn.GATConv: ``lin_src`` + ``lin_dst`` registered as the SAME Linear (PyG 2.3 / 2.4,
    int in_channels) or a single ``lin`` (PyG >= 2.5)."""

    def __init__(self, cin, heads, c, concat, spelling):
        super().__init__()
        lin = torch.nn.Linear(cin, heads * c, bias=False)
        if spelling == "lin_src":
            self.lin_src = lin
            self.lin_dst = lin
        else:
            self.lin = lin
        self.att_src = torch.nn.Parameter(torch.zeros(1, heads, c))
        self.att_dst = torch.nn.Parameter(torch.zeros(1, heads, c))
        self.bias = torch.nn.Parameter(torch.zeros(heads * c if concat else c))


class _StandInBlock(torch.nn.Module):
    def __init__(self, nc, spelling):
        super().__init__()
        self.conv1 = _StandInConv(nc, 2, nc, True, spelling)
        self.conv2 = _StandInConv(2 * nc, 1, nc, False, spelling)


class _StandInGATRes(torch.nn.Module):
    """The reference module's parameter tree (GraphModels.py:472-484) over the ORACLE's arithmetic: what replay_pyg.py
    would get from the reference, minus torch_geometric."""

    def __init__(self, oracle, nb, nc, spelling):
        super().__init__()
        self.oracle, self.nb = oracle, nb
        self.lin0 = torch.nn.Linear(1, nc)
        self.blocks = torch.nn.ModuleList([_StandInBlock(nc, spelling) for _ in range(nb)])
        self.lin1 = torch.nn.Linear(nc, 1)

    def forward(self, x, edge_index, batch=None, edge_attr=None):
        named = dict(self.named_parameters())
        p = {}
        for key in self.oracle.param_shapes(self.nb, self.lin0.out_features):
            k = key
            if k not in named:
                k = key.replace("lin_src.weight", "lin.weight")
            p[key] = named[k]
        return self.oracle.gatres_forward(p, x, edge_index, num_blocks=self.nb)


@pytest.mark.parametrize("spelling", ["lin_src", "lin"])
def test_replay_pyg_script_runs_over_a_stand_in_module(oracle, spelling, tmp_path, monkeypatch, capsys):
    R = _replay_module()
    with pytest.raises(SystemExit) as e:                                  # argument parsing
        R.main(["--help"])
    assert e.value.code == 0 and "--reference" in capsys.readouterr().out
    fixture = os.path.join(GOLDEN, "tiny_nb2_nc8.npz")
    d = np.load(fixture)
    nb, nc = int(d["num_blocks"]), int(d["nc"])
    # state_dict mapping, both PyG spellings: every key of the module is filled, the flat vector comes back unchanged
    model = _StandInGATRes(oracle, nb, nc, spelling)
    keys = set(model.state_dict())
    assert (f"blocks.0.conv1.{spelling}.weight" in keys) and (("blocks.0.conv1.lin_dst.weight" in keys) == (spelling == "lin_src"))
    R.fill_state_dict(model, d["params"], nb, nc)
    assert np.array_equal(R.flat_from(model, nb, nc), d["params"])
    if spelling == "lin_src":
        assert model.blocks[0].conv1.lin_src.weight is model.blocks[0].conv1.lin_dst.weight
    # the whole replay (train.py:159-190 on the fixture's batch), with the stand-in where the reference import would be
    monkeypatch.setattr(R, "load_reference_model", lambda root, nb_, nc_: _StandInGATRes(oracle, nb_, nc_, spelling))
    assert R.replay(fixture, "/nonexistent", 1e-5, out_dir=str(tmp_path), versions="stand-in") is True
    r = np.load(tmp_path / "tiny_nb2_nc8.pyg.npz")
    assert str(r["torch_geometric_version"]) == "stand-in"
    for k in ("out", "grads", "params_after"):
        assert r[k].shape == d[k].shape
    assert float(np.abs(r["out"] - d["out"]).max()) < 1e-6 and float(np.abs(r["grads"] - d["grads"]).max()) < 1e-5 * float(np.abs(d["grads"]).max()) + 1e-9
    # a reference checkout without torch_geometric: the script must say so, not crash
    monkeypatch.undo()
    R2 = _replay_module()
    try:
        import torch_geometric  # noqa: F401
    except ImportError:
        with pytest.raises(SystemExit, match="torch_geometric is not installed"):
            R2.load_reference_model("/nonexistent", nb, nc)

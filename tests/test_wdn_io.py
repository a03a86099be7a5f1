"""On-disk formats: EPANET .inp topology -> edge_index in the reference's order, the
zarr-v2 ZipStore reader with its Blosc / LZ4 decoder, and the .pth checkpoint dict.  CPU only (the LZ4 decoder is a host
function of the native library).

Pinned by: the networkx pipeline itself for the graph order (networkx is installed here; wntr and torch_geometric are
not: the wntr registry order and from_networkx's edge order are restated from their sources, see wdn_io's docstring),
hand-assembled LZ4 / Blosc byte strings that follow the published formats, and the committed fixture pair
tests/golden/wdn_tiny.{inp,zip} (tests/golden/make_io_fixtures.py)."""
import importlib.util
import os
import struct

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _fixtures():
    spec = importlib.util.spec_from_file_location("make_io_fixtures", os.path.join(GOLDEN, "make_io_fixtures.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _reference_pipeline(inp, keep):
    """DataLoader.py:230-254 + :28-37 with real networkx: wn.to_graph() as wntr builds it (one directed edge per link, nodes
    and links in registry order), nx.Graph(.).to_undirected(), subgraph(keep).copy(), then from_networkx's edge list
    (to_directed(), edges(), node index = position in G.nodes)."""
    import networkx as nx
    g0 = nx.MultiDiGraph()
    for n in inp["JUNCTIONS"] + inp["RESERVOIRS"] + inp["TANKS"]:
        g0.add_node(n)
    for name, a, b in inp["PIPES"] + inp["PUMPS"] + inp["VALVES"]:
        g0.add_edge(a, b, key=name)
    graph = nx.Graph(g0).to_undirected()
    new_graph = graph.subgraph(keep).copy() if keep is not None else graph
    d = new_graph.to_directed()
    mapping = dict(zip(d.nodes(), range(d.number_of_nodes())))
    edges = [(mapping[u], mapping[v]) for u, v in d.edges()]
    return torch.tensor(edges, dtype=torch.int64).t().reshape(2, -1), list(d.nodes())


def test_inp_topology_matches_the_networkx_pipeline(pkg):
    io = pkg.wdn_io
    inp = io.parse_inp(os.path.join(GOLDEN, "wdn_tiny.inp"))
    assert inp["JUNCTIONS"] == ["J1", "J2", "J3", "J4", "J5", "J6"] and inp["RESERVOIRS"] == ["R1"] and inp["TANKS"] == ["T1", "T2"]
    assert inp["PIPES"][2] == ("P3", "J3", "J2") and inp["PUMPS"] == [("PU1", "J5", "J6")] and len(inp["VALVES"]) == 2
    for removal, keep in (("keep_junction", inp["JUNCTIONS"]), ("keep_all", None),
                          ("reservoir", inp["JUNCTIONS"] + inp["TANKS"]), ("tank", inp["JUNCTIONS"] + inp["RESERVOIRS"])):
        ei, names = io.inp_edge_index(inp, removal)
        ref_ei, ref_names = _reference_pipeline(inp, keep)
        assert names == ref_names, removal
        assert torch.equal(ei, ref_ei), (removal, ei, ref_ei)
    ei, names = io.inp_edge_index(inp)                      # the default: junction-only subgraph (train.py:597-603)
    assert ei.shape == (2, 16) and names == inp["JUNCTIONS"]    # 8 junction-junction links (P9 is parallel to P4), both ways
    assert torch.equal(ei[0], torch.sort(ei[0], stable=True).values)
    # a randomly generated network, text round trip included
    rs = np.random.RandomState(3)
    nj = 40
    lines = ["[JUNCTIONS]"] + [f" N{i} 0 0" for i in range(nj)] + ["[RESERVOIRS]", " RES 10", "[TANKS]", " TK 5 1 0 2 3 0", "[PIPES]"]
    pairs = set()
    for k in range(70):
        a, b = rs.randint(0, nj + 2, 2)
        if a != b:
            nm = lambda i: f"N{i}" if i < nj else ("RES" if i == nj else "TK")
            lines.append(f" L{k} {nm(a)} {nm(b)} 10 100 100")
            pairs.add((a, b))
    big = io.parse_inp("\n".join(lines) + "\n[END]\n")
    for removal, keep in (("keep_junction", big["JUNCTIONS"]), ("keep_all", None)):
        ei, names = io.inp_edge_index(big, removal)
        ref_ei, ref_names = _reference_pipeline(big, keep)
        assert names == ref_names and torch.equal(ei, ref_ei), removal


def test_lz4_known_answer_blocks(pkg, lib):
    """Byte strings assembled by hand from the LZ4 block format: literals only; a match that overlaps its own output
    (run-length style); length fields that need 255-extension bytes; malformed input is refused, never over-read."""
    def dec(block, n):
        src = np.frombuffer(bytes(block), dtype=np.uint8)
        dst = np.zeros(n + 8, dtype=np.uint8)
        got = lib.gatres_lz4_decompress_host(src.ctypes.data, len(block), dst.ctypes.data, n)
        return got, dst[:max(got, 0)].tobytes()

    assert dec([0x50] + list(b"hello"), 5) == (5, b"hello")
    # token 0x1F: 1 literal 'a', match length 15 + ext 6 + 4 = 25, offset 1 -> 'a' * 26 ; then a final literal run 'xyz'
    assert dec([0x1F, ord("a"), 0x01, 0x00, 0x06, 0x30] + list(b"xyz"), 29) == (29, b"a" * 26 + b"xyz")
    # 'abc' then match offset 3 length 9 (copies itself forward), then 20 literals through an extension byte
    tail = bytes(range(65, 85))
    assert dec([0x35, 97, 98, 99, 0x03, 0x00, 0xF0, 5] + list(tail), 32) == (32, b"abc" * 4 + tail)
    assert dec([0x1F, ord("a"), 0x00, 0x00, 0x06], 40)[0] < 0          # offset 0
    assert dec([0x1F, ord("a"), 0x05, 0x00, 0x06], 40)[0] < 0          # offset beyond the output so far
    assert dec([0x50] + list(b"hel"), 5)[0] < 0                        # literal run cut short
    assert dec([0x50] + list(b"hello"), 3)[0] < 0                      # destination too small


def test_blosc_frames(pkg):
    io, fx = pkg.wdn_io, _fixtures()
    rs = np.random.RandomState(0)
    smooth = np.round(np.cumsum(rs.randn(5000)).astype("<f4"), 1).tobytes()
    for data, typesize, blocksize, shuffle, codec in ((smooth, 4, 2048, True, "lz4"), (smooth, 4, 4096, False, "lz4"),
                                                      (smooth[:1001], 4, 512, True, "lz4"), (smooth, 8, 8192, True, "zlib"),
                                                      (bytes(rs.randint(0, 256, 3000, dtype=np.uint8)), 1, 1024, False, "lz4")):
        frame = fx.blosc_compress(data, typesize, blocksize, shuffle, codec)
        assert io.blosc_decompress(frame) == data
    assert len(fx.blosc_compress(smooth, 4, 2048, True, "lz4")) < 0.8 * len(smooth)      # (real matches: the shuffle + LZ4 compress)
    # a memcpy frame written by hand: flags 0x2, payload right behind the 16-byte header
    payload = b"0123456789abcdef"
    frame = struct.pack("<BBBBIII", 2, 1, 0x2 | (1 << 5), 4, len(payload), len(payload), 16 + len(payload)) + payload
    assert io.blosc_decompress(frame) == payload
    with pytest.raises(ValueError):
        io.blosc_decompress(struct.pack("<BBBBIII", 2, 1, 4 << 5, 4, 16, 16, 40) + struct.pack("<i", 20) + b"x" * 20)   # zstd


def test_zarr_zip_fixture_and_load_wdn(pkg):
    io, fx = pkg.wdn_io, _fixtures()
    arrays = fx.fixture_arrays()
    root = io.ZarrZip(os.path.join(GOLDEN, "wdn_tiny.zip"))
    assert root.group_keys() == ["demand", "head", "pressure"] and root.attrs["note"] == "hand-made fixture"
    assert root.array_keys("pressure") == ["test", "train", "valid"]
    for split, a in arrays.items():
        assert np.array_equal(root.array(f"pressure/{split}"), a)                   # blosc(lz4, shuffle), 3 / 1 / 1 chunks
        assert np.array_equal(root.array(f"demand/{split}"), (a * 2.0).astype("<f4"))      # zlib
        assert np.array_equal(root.array(f"head/{split}"), (a * -1.0).astype("<f4"))       # uncompressed
    with pytest.raises(KeyError):
        root.array("pressure/nope")
    root.close()
    store, names = io.load_wdn(os.path.join(GOLDEN, "wdn_tiny.inp"), os.path.join(GOLDEN, "wdn_tiny.zip"),
                               feature="pressure", split="train", device="cpu")
    assert names == ["J1", "J2", "J3", "J4", "J5", "J6"] and len(store) == 700 and store.nodes_per_graph == 6
    raw = arrays["train"][:, :6]                                                   # junction columns come first in wn.node_name_list
    assert abs(store.mean - float(raw.mean())) < 1e-4 and abs(store.std - float(raw.std())) < 1e-4
    assert torch.allclose(store.descale(store.batch(torch.tensor([5, 2]))), torch.from_numpy(raw[[5, 2]]).reshape(-1, 1), atol=1e-3)
    assert torch.equal(store.edge_index_single, io.inp_edge_index(io.parse_inp(os.path.join(GOLDEN, "wdn_tiny.inp")))[0])
    with pytest.raises(KeyError):
        io.load_wdn(os.path.join(GOLDEN, "wdn_tiny.inp"), os.path.join(GOLDEN, "wdn_tiny.zip"), feature="flow", device="cpu")


def test_checkpoint_dict_round_trip(pkg, oracle, tmp_path):
    """train.py:433-451 / auxil.py:206-233: the .pth dict with the reference's keys; weights written under PyG's
    state_dict keys (lin_src / lin_dst, or lin.weight of PyG >= 2.5) load into the module."""
    io = pkg.wdn_io
    m = pkg.GATResMeanConv(num_blocks=2, nc=8)
    path = str(tmp_path / "best_gatres_small_x.pth")
    io.save_checkpoint(path, model_state_dict=m.state_dict(), optimizer_state_dict=None, epoch=3, loss=0.25,
                       val_metric_dict={"val_mae": 0.1}, mean=31.5, std=2.0, min=20.0, max=40.0, edge_attrs=None,
                       norm_type="znorm")
    m2, cp = io.load_checkpoint(path, pkg.GATResMeanConv(num_blocks=2, nc=8))
    assert torch.equal(m2.flat_parameters, m.flat_parameters) and cp["epoch"] == 3 and cp["norm_type"] == "znorm"
    assert cp["mean"] == 31.5 and cp["val_metric_dict"] == {"val_mae": 0.1}
    sd25 = {k.replace("lin_src.weight", "lin.weight"): v for k, v in m.state_dict().items() if "lin_dst" not in k}
    io.save_checkpoint(path, model_state_dict=sd25)
    m3, _ = io.load_checkpoint(path, pkg.GATResMeanConv(num_blocks=2, nc=8))
    assert torch.equal(m3.flat_parameters, m.flat_parameters)
    with pytest.raises(ValueError):
        io.save_checkpoint(str(tmp_path / "x.pt"), a=1)


@pytest.mark.gpu
def test_adam_state_moves_between_torch_and_the_native_trainer(pkg, oracle):
    """optimizer_state_dict of a reference checkpoint (torch.optim.Adam) <-> GATResTrainer's flat moments: two native steps,
    export, continue with torch.optim.Adam on the drop-in module == two more native steps."""
    from test_gpu_model import build
    io = pkg.wdn_io
    bs = 2
    ei = pkg.wdn_synth.collate_edge_index(pkg.wdn_synth.make_wdn_topology(), 388, bs).cuda()
    snaps = pkg.wdn_synth.make_snapshots(4 * bs, 388, seed=3).cuda()
    mask = torch.from_numpy(pkg.wdn_synth.generate_batch_mask([388] * bs, 0.95, np.random.RandomState(4))).cuda()
    ma, _ = build(pkg, oracle, 3, 32, seed=9)
    mb, _ = build(pkg, oracle, 3, 32, seed=9)
    ta = pkg.GATResTrainer(ma, ei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=False)
    tb = pkg.GATResTrainer(mb, ei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=False)
    for it in range(2):
        yb = snaps[it * bs:(it + 1) * bs].reshape(-1)
        ta.step(yb, yb, mask); tb.step(yb, yb, mask)
    opt = torch.optim.Adam(mb.parameters(), lr=5e-4, weight_decay=6e-6)
    opt.load_state_dict(io.adam_state_dict(tb))
    for it in range(2, 4):
        yb = snaps[it * bs:(it + 1) * bs].reshape(-1)
        ta.step(yb, yb, mask)
        opt.zero_grad()
        x = yb.reshape(-1, 1).clone(); x[mask] = 0
        out = mb(x, ei)
        torch.nn.functional.mse_loss(out[mask], yb.reshape(-1, 1)[mask]).backward()
        opt.step()
    d = (ma.flat_parameters - mb.flat_parameters).abs()
    assert float(d.max()) <= 2 * 5e-4 * 1.01 and float((d > 1e-6).double().mean()) < 0.01
    # and back: the torch optimizer's state into a fresh native trainer
    tc = pkg.GATResTrainer(mb, ei, 388 * bs, nodes_per_graph=[388] * bs, use_graph=False)
    io.load_adam_state_dict(tc, opt.state_dict())
    assert tc.optimizer_step == 4 and ta.optimizer_step == 4
    assert float((tc.exp_avg - ta.exp_avg).abs().max()) <= 1e-5 * float(ta.exp_avg.abs().max()) + 1e-9


def test_parse_inp_paths_text_and_errors(pkg, tmp_path):
    """a path with brackets in it is a path; text goes through ``text=``; a text without [JUNCTIONS] raises;
    a keep set smaller than half of the network warns that the reference's node order is undefined there."""
    io = pkg.wdn_io
    src = open(os.path.join(GOLDEN, "wdn_tiny.inp")).read()
    d = tmp_path / "run[1]"
    d.mkdir()
    p = d / "net.inp"
    p.write_text(src)
    ref = io.parse_inp(os.path.join(GOLDEN, "wdn_tiny.inp"))
    assert io.parse_inp(str(p)) == ref and io.parse_inp(p) == ref and io.parse_inp(text=src) == ref
    with pytest.raises(FileNotFoundError):
        io.parse_inp(str(d / "missing.inp"))
    with pytest.raises(ValueError):
        io.parse_inp(text="[PIPES]\n P1 A B 1 1 1\n")
    with pytest.raises(ValueError):
        io.parse_inp()
    few = dict(ref)
    few = {k: list(v) for k, v in ref.items()}
    few["JUNCTIONS"], few["TANKS"] = ref["JUNCTIONS"][:2], ref["TANKS"] + ref["JUNCTIONS"][2:]      # 2 of 9 nodes are junctions
    with pytest.warns(UserWarning, match="undefined"):
        ei, names = io.inp_edge_index(few, "keep_junction")
    assert names == few["JUNCTIONS"]

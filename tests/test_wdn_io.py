"""On-disk formats: EPANET .inp topology -> edge_index in the reference's order, the
zarr-v2 ZipStore reader with its Blosc / LZ4 decoder, and the .pth checkpoint dict.  CPU only (the LZ4 decoder is a host
function of the native library).

Pinned by: the networkx pipeline itself for the graph order (networkx is installed here; wntr and torch_geometric are
not: the wntr registry order and from_networkx's edge order are restated from their sources, see wdn_io's docstring),
byte strings ASSEMBLED BY HAND from the published formats -- LZ4 sequences, a split + shuffled multi-block Blosc frame with
raw and compressed streams and a leftover block, a zarr-v2 ZipStore with numcodecs' compressor entry and edge chunks, an
.inp with tabs / comments / pumps / valves: expected values are literals, no encoder of this repo is involved (the last
section of this file) --, and, as a regression of the committed fixture pair tests/golden/wdn_tiny.{inp,zip}, the
round trip through the test-side writer tests/golden/make_io_fixtures.py (which by itself only proves reader and writer
consistent)."""
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


# ------------------------------------------------------------------------------------- bytes this repo did not write
# Everything below is assembled BY HAND from the published formats -- no encoder of tests/golden/make_io_fixtures.py is
# involved, expected values are literals: a reader that agrees with its own writer only proves the pair consistent.
#   LZ4 block format (lz4_Block_format.md): a sequence = token (high nibble: literal length, low nibble: match length - 4;
#   a nibble of 15 is continued by extension bytes, each adding up to 255) | literals | 2-byte little-endian offset |
#   match-length extension; the last sequence ends after its literals.
#   C-Blosc 1.x frame (README_HEADER.rst + blosc.c): 16-byte header {version, versionlz, flags, typesize, nbytes,
#   blocksize, cbytes}; flags: 0x01 byte shuffle, 0x02 memcpyed, 0x10 blocks NOT split, bits 5..7 compressor format
#   (1 = LZ4); then one int32 offset per block; a block of a split frame (typesize <= 16, blocksize / typesize >= 128, not
#   the leftover block) is `typesize` streams -- the bytes 0, 1, ... of every element after the shuffle --, every stream
#   is {int32 compressed size | data}, and a stream whose size equals its uncompressed length is stored raw.
#   zarr v2 (spec v2.0): `.zarray` JSON {chunks, compressor, dtype, fill_value, filters, order, shape, zarr_format}, chunk
#   keys "i.j" under the array's prefix, EDGE chunks stored at the full chunk shape; `.zgroup`, `.zattrs`.
def _lz4_run(byte: int, n: int) -> bytes:
    """n >= 20 copies of one byte as two LZ4 sequences: 1 literal + an overlapping match (offset 1) of n - 6, then the
    5 literals every block must end with."""
    ml = n - 6 - 4                      # match length field: length - 4 (minmatch)
    assert ml >= 15 and ml - 15 < 255
    return bytes([0x1F, byte, 0x01, 0x00, ml - 15, 0x50]) + bytes([byte]) * 5


def _hand_blosc_frame():
    """Two full 512-byte blocks of float32 (split into 4 shuffled byte streams each) + a 64-byte leftover block (one
    stream, stored raw).  Values: 128 x 1.0 | 0x40000000 + k for k = 0 .. 127 (2.0 .. 2.00003) | 16 x 3.0."""
    import struct
    streams0 = [_lz4_run(b, 128) for b in (0x00, 0x00, 0x80, 0x3F)]            # 1.0f = 3F 80 00 00, little endian bytes 0..3
    assert all(len(s) == 11 for s in streams0)
    block0 = b"".join(struct.pack("<i", len(s)) + s for s in streams0)
    raw_lo = bytes(range(128))                                                # byte 0 of every element: 0 .. 127, incompressible
    streams1 = [raw_lo] + [_lz4_run(b, 128) for b in (0x00, 0x00, 0x40)]       # 40 00 00 kk
    block1 = b"".join(struct.pack("<i", len(s)) + s for s in streams1)         # the first stream: size == 128 -> raw
    left = bytes([0x00] * 16 + [0x00] * 16 + [0x40] * 16 + [0x40] * 16)        # 3.0f = 40 40 00 00, shuffled, raw
    block2 = struct.pack("<i", 64) + left
    starts = [16 + 12, 16 + 12 + len(block0), 16 + 12 + len(block0) + len(block1)]
    body = struct.pack("<3i", *starts) + block0 + block1 + block2
    nbytes = 512 + 512 + 64
    hdr = struct.pack("<BBBBIII", 2, 1, 0x01 | (1 << 5), 4, nbytes, 512, 16 + len(body))
    frame = hdr + body
    assert len(frame) == 333 and starts == [28, 88, 265]                       # (the sizes worked out by hand)
    want = np.concatenate([np.full(128, 1.0, "<f4"), (np.arange(128, dtype="<u4") + 0x40000000).view("<f4"),
                           np.full(16, 3.0, "<f4")])
    return frame, want


def test_blosc_frame_assembled_by_hand(pkg):
    frame, want = _hand_blosc_frame()
    got = np.frombuffer(pkg.wdn_io.blosc_decompress(frame), dtype="<f4")
    assert got.shape == (272,) and np.array_equal(got.view("<u4"), want.view("<u4"))
    assert got[0] == 1.0 and got[128] == 2.0 and got[255] > 2.0 and got[-1] == 3.0
    # the same frame marked "blocks not split" must NOT decode (its first block would be read as one stream)
    bad = bytearray(frame); bad[2] |= 0x10
    with pytest.raises(Exception):
        pkg.wdn_io.blosc_decompress(bytes(bad))
    # truncated frame: refused
    with pytest.raises(ValueError):
        pkg.wdn_io.blosc_decompress(frame[:-10])


def test_zarr_zip_assembled_by_hand(pkg, tmp_path):
    """A ZipStore whose members are written byte for byte here: a 3 x 8 float32 array in 2 x 5 chunks (so the right and the
    bottom chunks are EDGE chunks stored at full size), compressor entry exactly as numcodecs.Blosc serialises it; the
    40-byte chunks are below Blosc's 128-byte minimum, so every chunk frame is the `memcpyed` form (flags 0x02, shuffle bit
    recorded, payload NOT shuffled).  A second array uses the 272-element frame above as its single chunk; a third has no
    compressor and a missing chunk (fill value)."""
    import json
    import struct
    import zipfile
    io = pkg.wdn_io
    full = np.arange(24, dtype="<f4").reshape(3, 8) * 0.5 - 2.0
    zarray = ('{\n    "chunks": [\n        2,\n        5\n    ],\n    "compressor": {\n        "blocksize": 0,\n        "clevel": 5,\n'
              '        "cname": "lz4",\n        "id": "blosc",\n        "shuffle": 1\n    },\n    "dtype": "<f4",\n    "fill_value": 0.0,\n'
              '    "filters": null,\n    "order": "C",\n    "shape": [\n        3,\n        8\n    ],\n    "zarr_format": 2\n}')
    assert json.loads(zarray)["compressor"] == {"blocksize": 0, "clevel": 5, "cname": "lz4", "id": "blosc", "shuffle": 1}

    def memcpy_frame(chunk):                       # blosc.c: nbytes < BLOSC_MIN_BUFFERSIZE (128) -> memcpyed, blocksize = nbytes
        raw = chunk.astype("<f4").tobytes()
        return struct.pack("<BBBBIII", 2, 1, 0x02 | 0x01 | (1 << 5), 4, len(raw), len(raw), 16 + len(raw)) + raw

    def chunk_of(i, j):                            # zarr stores edge chunks at the full chunk shape, padded with the fill value
        c = np.zeros((2, 5), "<f4")
        part = full[2 * i:2 * i + 2, 5 * j:5 * j + 5]
        c[:part.shape[0], :part.shape[1]] = part
        return c

    frame272, want272 = _hand_blosc_frame()
    path = tmp_path / "hand.zip"
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:      # zarr.ZipStore: members stored, not deflated
        z.writestr(".zgroup", '{\n    "zarr_format": 2\n}')
        z.writestr(".zattrs", '{\n    "ordered_name_list": [\n        "J1",\n        "J2"\n    ]\n}')
        z.writestr("pressure/.zgroup", '{\n    "zarr_format": 2\n}')
        z.writestr("pressure/train/.zarray", zarray)
        for i in range(2):
            for j in range(2):
                z.writestr(f"pressure/train/{i}.{j}", memcpy_frame(chunk_of(i, j)))
        z.writestr("pressure/valid/.zarray", zarray.replace("3,\n        8", "272").replace("2,\n        5", "272"))
        z.writestr("pressure/valid/0", frame272)
        z.writestr("pressure/test/.zarray", '{"chunks": [2, 4], "compressor": null, "dtype": "<f4", "fill_value": -1.0, "filters": null, '
                                            '"order": "C", "shape": [2, 8], "zarr_format": 2}')
        z.writestr("pressure/test/0.1", np.full((2, 4), 7.0, "<f4").tobytes())      # chunk 0.0 was never written
    root = io.ZarrZip(str(path))
    assert root.group_keys() == ["pressure"] and root.attrs["ordered_name_list"] == ["J1", "J2"]
    assert root.array_keys("pressure") == ["test", "train", "valid"]
    assert np.array_equal(root.array("pressure/train"), full)
    assert np.array_equal(root.array("pressure/valid").view("<u4"), want272.view("<u4"))
    t = root.array("pressure/test")
    assert np.array_equal(t[:, :4], np.full((2, 4), -1.0, "<f4")) and np.array_equal(t[:, 4:], np.full((2, 4), 7.0, "<f4"))
    root.close()


def test_inp_with_tabs_comments_pumps_and_valves(pkg):
    """EPANET 2.2 users manual, appendix C: sections in any order and case, `;` comments (whole-line and trailing), tabs
    as separators, [PUMPS] (id node1 node2 keyword value) and [VALVES] (id node1 node2 diameter type setting loss) lines;
    expected topology written out by hand (DataLoader.py:212-254 order: junctions, reservoirs, tanks; pipes, pumps, valves)."""
    io = pkg.wdn_io
    text = ("[TITLE]\n a ; network\n\n[junctions]\n;ID\tElev\tDemand\tPattern\n A\t10\t1.5\t;first\n B  12 0\n\tC\t9\t2\tP1\n"
            "[RESERVOIRS]\n;ID Head\n R\t100\n[TANKS]\n T 50 3 0 6 12 0\n"
            "[VALVES]\n;ID Node1 Node2 Diameter Type Setting MinorLoss\n V1\tC\tT\t12\tPRV\t40\t0 ; to the tank\n"
            "[PIPES]\n;ID Node1 Node2 Length Diameter Roughness MinorLoss Status\n P1 A B 100 12 100 0 Open\n P2\tB\tC\t50\t8\t100\t0\tOpen\n"
            "[Pumps]\n PU1 R A HEAD 1 ; source\n[END]\n")
    inp = io.parse_inp(text=text)
    assert inp["JUNCTIONS"] == ["A", "B", "C"] and inp["RESERVOIRS"] == ["R"] and inp["TANKS"] == ["T"]
    assert inp["PIPES"] == [("P1", "A", "B"), ("P2", "B", "C")] and inp["PUMPS"] == [("PU1", "R", "A")]
    assert inp["VALVES"] == [("V1", "C", "T")]
    ei, names = io.inp_edge_index(inp, "keep_junction")
    assert names == ["A", "B", "C"] and ei.tolist() == [[0, 1, 1, 2], [1, 0, 2, 1]]        # A-B, B-C, both ways, by source
    ei_all, names_all = io.inp_edge_index(inp, "keep_all")
    assert names_all == ["A", "B", "C", "R", "T"]
    # adjacency after nx.Graph(wn.to_graph()).to_undirected(): A: [B, R]; B: [A, C]; C: [B, T]; R: [A]; T: [C]
    assert ei_all.tolist() == [[0, 0, 1, 1, 2, 2, 3, 4], [1, 3, 0, 2, 1, 4, 0, 2]]
    assert io.inp_node_order(inp) == ["A", "B", "C", "R", "T"]

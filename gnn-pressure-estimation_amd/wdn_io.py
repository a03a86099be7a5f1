"""On-disk formats on either side of the hot path: what the reference's ``WDNDataset.collect``
(gnn_pressure_estimation/utils/DataLoader.py:206-258) reads and what its checkpoints hold, without wntr / zarr /
numcodecs / torch_geometric -- none of them is installable here.

* ``parse_inp`` / ``inp_edge_index``: the topology of an EPANET ``.inp`` file as the ``edge_index`` the reference feeds its
  models.  The reference builds it as ``pgu.from_networkx(nx.Graph(wn.to_graph()).to_undirected().subgraph(keep).copy())``
  (DataLoader.py:28-37, :236-254); the node and edge ORDER that pipeline produces is restated here step by step (it
  decides the order of every neighbour sum, hence the last bits of every fp32 result).  The networkx part is checked
  against networkx itself in tests/test_wdn_io.py; the wntr part (registry order: junctions, reservoirs, tanks; pipes,
  pumps, valves; ``to_graph`` adding one edge per link from start node to end node) is from the author's knowledge of
  wntr >= 0.4 and cannot be verified in this container.
* ``ZarrZip``: reader for the zarr-v2 ``ZipStore`` the scenario generator writes (scenegenv7.py:701-725):
  ``root[feature][split]`` arrays, C order, chunked, compressor ``null`` / ``zlib`` / ``blosc`` (Blosc-1 container with the
  lz4, lz4hc, zlib or memcpy codecs and the byte-shuffle filter; zarr's default is Blosc(lz4, shuffle)).  The container
  and the LZ4 block format are implemented from their published specifications; no real zarr output exists in this
  container to read back, so the committed fixture is written by this module's own (test-side) encoder.
* ``save_checkpoint`` / ``load_checkpoint``: the ``.pth`` dict of train.py:433-451 / utils/auxil.py:206-233, plus the
  conversion between ``GATResTrainer``'s flat Adam moments and ``torch.optim.Adam.state_dict()`` so a run can move
  between the reference loop and the native trainer.
* ``load_wdn``: the three together -> ``(SnapshotStore, node_names)``.
"""
from __future__ import annotations

import io
import json
import struct
import zipfile
import zlib
from collections import OrderedDict
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import os
import warnings

import numpy as np
import torch

from . import _native

# ---------------------------------------------------------------------------------------------------------------- .inp
_NODE_SECTIONS = ("JUNCTIONS", "RESERVOIRS", "TANKS")
_LINK_SECTIONS = ("PIPES", "PUMPS", "VALVES")


def parse_inp(source=None, *, text: Optional[str] = None) -> Dict[str, list]:
    """Topology sections of an EPANET input file.  ``source``: a path (``str`` / ``os.PathLike``); ``text=``: the file's
    text instead.  (A ``str`` that is not an existing path but contains a newline is still taken as text, as before; a path
    with brackets in it -- ``/data/run[1]/ctown.inp`` -- is a path.)  Returns
    ``{"JUNCTIONS": [id, ...], "RESERVOIRS": [...], "TANKS": [...], "PIPES": [(id, node1, node2), ...], "PUMPS": [...],
    "VALVES": [...]}`` in file order.  ``;`` starts a comment, section names are case-insensitive, ids are the first
    whitespace-separated tokens of a line (EPANET 2.2 users manual, appendix C).  Raises if the text has no
    ``[JUNCTIONS]`` section (not an EPANET input file)."""
    if (source is None) == (text is None):
        raise ValueError("parse_inp: give a path OR text=")
    if text is None:
        if isinstance(source, os.PathLike) or os.path.exists(str(source)) or "\n" not in str(source):
            with open(source, "r", encoding="latin-1") as f:       # (a missing file raises FileNotFoundError here)
                text = f.read()
        else:
            text = str(source)
    out: Dict[str, list] = {k: [] for k in _NODE_SECTIONS + _LINK_SECTIONS}
    section = None
    seen_sections = set()
    for raw in str(text).splitlines():
        line = raw.split(";", 1)[0].strip()
        if not line:
            continue
        if line.startswith("["):
            section = line.strip("[]").strip().upper()
            seen_sections.add(section)
            continue
        tok = line.split()
        if section in _NODE_SECTIONS:
            out[section].append(tok[0])
        elif section in _LINK_SECTIONS:
            if len(tok) < 3:
                raise ValueError(f"[{section}] line needs an id and two node ids: {raw!r}")
            out[section].append((tok[0], tok[1], tok[2]))
        if section is not None:
            seen_sections.add(section)
    if "JUNCTIONS" not in seen_sections:
        raise ValueError("no [JUNCTIONS] section: not an EPANET .inp file")
    return out


def _readd(nodes: Sequence[str], adj: "OrderedDict[str, list]") -> "OrderedDict[str, list]":
    """What ``Graph.copy()`` / ``to_undirected()`` / ``subgraph().copy()`` do to the adjacency ORDER: nodes are re-inserted in
    order, then every (u, v) of ``for u in adj: for v in adj[u]`` is re-added, which appends v to u's list AND u to v's
    list if not there yet -- so a node's neighbours that come earlier in the node order end up first."""
    new = OrderedDict((n, []) for n in nodes)
    seen = {n: set() for n in nodes}
    for u in nodes:
        for v in adj[u]:
            if v not in seen[u]:
                seen[u].add(v); new[u].append(v)
            if u not in seen[v]:
                seen[v].add(u); new[v].append(u)
    return new


def inp_edge_index(inp: Dict[str, list], removal: str = "keep_junction") -> Tuple[torch.Tensor, List[str]]:
    """``edge_index`` int64 [2, E] (both directions of every link, grouped by source node) and the kept node names, in the
    order ``get_graph_template(nx.Graph(wn.to_graph()).to_undirected().subgraph(keep_list).copy())`` yields
    (DataLoader.py:28-37, :230-254).  ``removal``: ``keep_junction`` (train.py's default, :597-603), ``keep_all``,
    ``reservoir`` (drop reservoirs) or ``tank`` (drop tanks)."""
    nodes = list(inp["JUNCTIONS"]) + list(inp["RESERVOIRS"]) + list(inp["TANKS"])        # wntr node registry order
    links = list(inp["PIPES"]) + list(inp["PUMPS"]) + list(inp["VALVES"])                 # wntr link registry order
    index = {n: i for i, n in enumerate(nodes)}
    if len(index) != len(nodes):
        raise ValueError("duplicate node id in the .inp file")
    # wn.to_graph(): MultiDiGraph, one edge start -> end per link in registry order => successor lists in link order
    succ = OrderedDict((n, []) for n in nodes)
    for _, a, b in links:
        if a not in index or b not in index:
            raise ValueError(f"link endpoint {a!r} / {b!r} is not a node")
        succ[a].append(b)
    # nx.Graph(multidigraph): add_edge(u, v) for u in node order, v in successor order (parallel links collapse)
    adj = OrderedDict((n, []) for n in nodes)
    seen = {n: set() for n in nodes}
    for u in nodes:
        for v in succ[u]:
            if v not in seen[u]:
                seen[u].add(v); adj[u].append(v)
            if u not in seen[v]:
                seen[v].add(u); adj[v].append(u)
    adj = _readd(nodes, adj)                                                               # .to_undirected()
    subgraph = True
    if removal == "keep_junction":
        keep = set(inp["JUNCTIONS"])
    elif removal == "keep_all":
        keep, subgraph = set(nodes), False
    elif removal == "reservoir":           # (get_keep_list: no reservoirs / tanks = no keep list = the graph itself)
        keep, subgraph = set(nodes) - set(inp["RESERVOIRS"]), bool(inp["RESERVOIRS"])
    elif removal == "tank":
        keep, subgraph = set(nodes) - set(inp["TANKS"]), bool(inp["TANKS"])
    else:
        raise ValueError(f"removal {removal!r}: use keep_junction / keep_all / reservoir / tank")
    kept = [n for n in nodes if n in keep]
    if subgraph and 2 * len(keep) < len(nodes):
        # networkx walks the KEEP SET (hash order of the names) instead of the graph when it is the smaller one
        # (FilterAtlas.__iter__): the reference's node order is then arbitrary from run to run
        warnings.warn("fewer than half of the nodes are kept: the reference's node order is undefined here (networkx "
                      "iterates the keep set); using the .inp registry order", stacklevel=2)
    sub = OrderedDict((n, [v for v in adj[n] if v in keep]) for n in kept)
    if subgraph:
        sub = _readd(kept, sub)                                                            # .subgraph(keep).copy()
    # from_networkx: to_directed() keeps adjacency order; edges() = for u in nodes: for v in succ[u]
    new_id = {n: i for i, n in enumerate(kept)}
    src = [new_id[u] for u in kept for _ in sub[u]]
    dst = [new_id[v] for u in kept for v in sub[u]]
    return torch.tensor([src, dst], dtype=torch.int64).reshape(2, -1), kept


def inp_node_order(inp: Dict[str, list]) -> List[str]:
    """``wn.node_name_list``: the column order of the zarr arrays (DataLoader.py:243-249)."""
    return list(inp["JUNCTIONS"]) + list(inp["RESERVOIRS"]) + list(inp["TANKS"])


# ------------------------------------------------------------------------------------------------------------- blosc
_BLOSC_CODECS = {0: "blosclz", 1: "lz4", 2: "snappy", 3: "zlib", 4: "zstd"}


def _lz4_block(src: bytes, out_len: int) -> bytes:
    lib = _native.load()
    dst = np.empty(out_len, dtype=np.uint8)
    s = np.frombuffer(src, dtype=np.uint8)
    n = lib.gatres_lz4_decompress_host(s.ctypes.data, len(src), dst.ctypes.data, out_len)
    if n != out_len:
        raise ValueError(f"LZ4 block: expected {out_len} bytes, decoder returned {n}")
    return dst.tobytes()


def blosc_decompress(buf: bytes) -> bytes:
    """One Blosc-1 frame (what ``numcodecs.Blosc`` writes per chunk).  Header: version, versionlz, flags, typesize, nbytes,
    blocksize, cbytes (16 bytes, little endian); flags: 0x1 byte shuffle, 0x2 memcpy, 0x4 bit shuffle, 0x10 blocks not
    split, bits 5-7 codec.  Then one int32 start offset per block; a block is ``typesize`` separately compressed streams
    when it was split (each prefixed by its int32 compressed size; a stream stored raw has size == its uncompressed
    length), else one."""
    if len(buf) < 16:
        raise ValueError("Blosc frame shorter than its header")
    _ver, _verlz, flags, typesize = buf[0], buf[1], buf[2], buf[3]
    nbytes, blocksize, cbytes = struct.unpack_from("<III", buf, 4)
    if cbytes > len(buf):
        raise ValueError("Blosc frame truncated")
    if flags & 0x4:
        raise ValueError("Blosc bit-shuffle is not supported (the reference's stores use the default byte shuffle)")
    if flags & 0x2:
        return bytes(buf[16:16 + nbytes])
    codec = _BLOSC_CODECS.get(flags >> 5)
    if codec not in ("lz4", "zlib"):
        raise ValueError(f"Blosc codec {codec!r} is not supported (lz4 / lz4hc / zlib / memcpy are)")
    if nbytes == 0:
        return b""
    nblocks = (nbytes + blocksize - 1) // blocksize
    bstarts = struct.unpack_from(f"<{nblocks}i", buf, 16)
    dont_split = bool(flags & 0x10)
    out = bytearray(nbytes)
    for b in range(nblocks):
        bsize = min(blocksize, nbytes - b * blocksize)
        leftover = bsize != blocksize
        nsplits = typesize if (not dont_split and not leftover and typesize <= 16 and bsize // typesize >= 128) else 1
        neblock = bsize // nsplits
        pos = bstarts[b]
        block = bytearray()
        for _ in range(nsplits):
            (csize,) = struct.unpack_from("<i", buf, pos)
            pos += 4
            chunk = bytes(buf[pos:pos + csize])
            pos += csize
            if csize == neblock:
                block += chunk
            elif codec == "lz4":
                block += _lz4_block(chunk, neblock)
            else:
                block += zlib.decompress(chunk)
        if len(block) != bsize:
            raise ValueError("Blosc block decoded to the wrong length")
        if (flags & 0x1) and typesize > 1:
            n_el = bsize // typesize
            body = np.frombuffer(bytes(block[:n_el * typesize]), dtype=np.uint8).reshape(typesize, n_el).T.reshape(-1)
            block = bytearray(body.tobytes()) + block[n_el * typesize:]
        out[b * blocksize:b * blocksize + bsize] = block
    return bytes(out)


# -------------------------------------------------------------------------------------------------------------- zarr
class ZarrZip:
    """Read-only view of a zarr-v2 group hierarchy inside a zip file (``zarr.ZipStore``)."""

    def __init__(self, path: str):
        self.path = path
        self._zip = zipfile.ZipFile(path, "r")
        self._names = set(self._zip.namelist())
        if ".zgroup" not in self._names and not any(n.endswith(".zarray") for n in self._names):
            raise ValueError(f"{path}: no zarr metadata (.zgroup / .zarray) inside")
        self.attrs = self._json(".zattrs") or {}

    def _json(self, name: str):
        return json.loads(self._zip.read(name).decode("utf-8")) if name in self._names else None

    def close(self) -> None:
        self._zip.close()

    def group_keys(self, prefix: str = "") -> List[str]:
        """Sub-groups directly under ``prefix`` (``root.group_keys()``, DataLoader.py:213)."""
        pre = prefix.strip("/") + "/" if prefix.strip("/") else ""
        keys = set()
        for n in self._names:
            if n.startswith(pre) and n.endswith("/.zgroup"):
                rest = n[len(pre):-len("/.zgroup")]
                if rest and "/" not in rest:
                    keys.add(rest)
        return sorted(keys)

    def array_keys(self, prefix: str = "") -> List[str]:
        pre = prefix.strip("/") + "/" if prefix.strip("/") else ""
        keys = set()
        for n in self._names:
            if n.startswith(pre) and n.endswith("/.zarray"):
                rest = n[len(pre):-len("/.zarray")]
                if rest and "/" not in rest:
                    keys.add(rest)
        return sorted(keys)

    def array(self, key: str) -> np.ndarray:
        """``np.array(root[feature][split])`` for ``key = "feature/split"`` (DataLoader.py:239)."""
        key = key.strip("/")
        meta = self._json(key + "/.zarray")
        if meta is None:
            raise KeyError(f"{self.path}: no array {key!r}")
        if meta.get("zarr_format") != 2:
            raise ValueError("only zarr format 2 is supported")
        if meta.get("order", "C") != "C":
            raise ValueError("only C-order arrays are supported")
        if meta.get("filters"):
            raise ValueError("zarr filters are not supported")
        shape, chunks = tuple(meta["shape"]), tuple(meta["chunks"])
        dtype = np.dtype(meta["dtype"])
        comp = meta.get("compressor")
        sep = meta.get("dimension_separator", ".")
        fill = meta.get("fill_value")
        fillv = {None: 0, "NaN": np.nan, "Infinity": np.inf, "-Infinity": -np.inf}.get(fill, fill) if not isinstance(
            fill, (int, float)) else fill
        out = np.full(shape, fillv, dtype=dtype)
        grid = [max(1, -(-s // c)) for s, c in zip(shape, chunks)]
        for idx in np.ndindex(*grid):
            name = key + "/" + sep.join(str(i) for i in idx) if shape else key + "/0"
            if name not in self._names:
                continue                                  # an unwritten chunk holds the fill value
            raw = self._zip.read(name)
            if comp is None:
                data = raw
            elif comp.get("id") == "blosc":
                data = blosc_decompress(raw)
            elif comp.get("id") == "zlib" or comp.get("id") == "gzip":
                data = zlib.decompress(raw, 15 + 32)
            else:
                raise ValueError(f"zarr compressor {comp.get('id')!r} is not supported")
            chunk = np.frombuffer(data, dtype=dtype).reshape(chunks)
            sel = tuple(slice(i * c, min((i + 1) * c, s)) for i, c, s in zip(idx, chunks, shape))
            out[sel] = chunk[tuple(slice(0, sl.stop - sl.start) for sl in sel)]
        return out


# -------------------------------------------------------------------------------------------------------- checkpoints
def save_checkpoint(path: str, **kwargs) -> str:
    """utils/auxil.py:224-233: ``torch.save(kwargs, path)``.  train.py:433-451 stores ``model_state_dict``,
    ``optimizer_state_dict``, ``epoch``, ``loss``, the metric dicts, ``mean`` / ``std`` / ``min`` / ``max``, the edge statistics
    and ``norm_type``."""
    if not str(path).endswith(".pth"):
        raise ValueError("checkpoint paths end in .pth (utils/auxil.py:216)")
    torch.save(kwargs, path)
    return path


def load_checkpoint(path: str, model: torch.nn.Module, map_location=None):
    """utils/auxil.py:206-221.  The model's loader accepts PyG's ``lin_src`` / ``lin_dst`` pair as well as the single
    ``lin.weight`` of PyG >= 2.5."""
    if not str(path).endswith(".pth"):
        raise ValueError("checkpoint paths end in .pth (utils/auxil.py:216)")
    if model is None:
        raise ValueError("a model to load the weights into is required")
    cp = torch.load(path, map_location=map_location, weights_only=False)
    model.load_state_dict(cp["model_state_dict"])
    return model, cp


def adam_state_dict(trainer) -> dict:
    """``torch.optim.Adam(model.parameters(), ...).state_dict()`` equivalent of a ``GATResTrainer``'s flat moments: one
    state entry per parameter in ``model.parameters()`` order (train.py:348, :436)."""
    params = trainer.model._param_list
    step = int(trainer.step_counter[0].item())
    state, off = {}, 0
    for i, p in enumerate(params):
        n = p.numel()
        state[i] = {"step": torch.tensor(float(step)),
                    "exp_avg": trainer.exp_avg[off:off + n].view(p.shape).clone(),
                    "exp_avg_sq": trainer.exp_avg_sq[off:off + n].view(p.shape).clone()}
        off += n
    h = trainer.hparams
    group = {"lr": h["lr"], "betas": (h["beta1"], h["beta2"]), "eps": h["eps"], "weight_decay": h["weight_decay"],
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
             "fused": None, "params": list(range(len(params)))}
    return {"state": state, "param_groups": [group]}


def load_adam_state_dict(trainer, state_dict: dict) -> None:
    """The reverse: continue a reference run (``optimizer_state_dict`` of a train.py checkpoint) on the native trainer."""
    params = trainer.model._param_list
    st = state_dict["state"]
    if len(st) not in (0, len(params)):
        raise ValueError(f"optimizer state has {len(st)} entries, the model {len(params)} parameters")
    off, step = 0, 0
    with torch.no_grad():
        for i, p in enumerate(params):
            n = p.numel()
            if st:
                e = st[i]
                trainer.exp_avg[off:off + n].copy_(e["exp_avg"].reshape(-1))
                trainer.exp_avg_sq[off:off + n].copy_(e["exp_avg_sq"].reshape(-1))
                step = int(float(e["step"]))
            off += n
        trainer.step_counter[0] = step
    g = state_dict["param_groups"][0]
    trainer.set_hparams(lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"], weight_decay=g["weight_decay"])


# --------------------------------------------------------------------------------------------------------- all of it
def load_wdn(inp_path: str, zip_path: str, feature: str = "pressure", split: str = "train",
             removal: str = "keep_junction", num_records: Optional[int] = None, device=None, **store_kw):
    """``WDNDataset.collect`` + ``compute_stats`` for one file pair (DataLoader.py:206-258, :120-147): the ``feature`` array
    of ``split``, its columns restricted to the kept nodes (``np.take(array, taken_indices, axis=-1)`` over
    ``wn.node_name_list``), on the device as a ``SnapshotStore`` together with the junction-subgraph ``edge_index``."""
    from .snapshot_store import SnapshotStore
    inp = parse_inp(inp_path)
    edge_index, kept = inp_edge_index(inp, removal)
    root = ZarrZip(zip_path)
    try:
        if feature not in root.group_keys():
            raise KeyError(f"feature {feature} is unavailable in zarr file {zip_path}")
        arr = root.array(f"{feature}/{split}")
    finally:
        root.close()
    if num_records is not None:
        arr = arr[:num_records]
    order = inp_node_order(inp)
    keep = set(kept)
    cols = [i for i, n in enumerate(order) if n in keep]
    if arr.shape[-1] < len(order):
        raise ValueError(f"the store has {arr.shape[-1]} columns, the network {len(order)} nodes")
    arr = np.take(arr, cols, axis=-1)
    store = SnapshotStore(torch.from_numpy(np.ascontiguousarray(arr, dtype=np.float32)), edge_index, device=device, **store_kw)
    return store, kept

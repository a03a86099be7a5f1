"""Writes tests/golden/wdn_tiny.inp and tests/golden/wdn_tiny.zip: a hand-made 9-node EPANET topology and a zarr-v2
ZipStore with the layout scenegenv7.py:701-725 produces (root[feature][split], chunked along the snapshot axis), its
chunks compressed with a Blosc-1 / LZ4 ENCODER written here from the published formats (zarr / numcodecs are not
installable in this container; tests/test_wdn_io.py reads the files back with the product's decoder).

    python tests/golden/make_io_fixtures.py
"""
import json
import os
import struct
import zipfile
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

INP = """[TITLE]
tiny test network (hand-made; not from the reference)

[JUNCTIONS]
;ID   Elev  Demand
 J1    10    1.0
 J2    12    0.5     ; a comment
 J3    11    0.7
 J4    9     0.2
 J5    8     0.9
 J6    14    0.1

[RESERVOIRS]
 R1    50

[TANKS]
 T1    40  3 1 6 10 0
 T2    42  3 1 6 10 0

[PIPES]
;ID  Node1 Node2 Length Diam Rough
 P1   R1    J1    100  300  100
 P2   J1    J2    100  200  100
 P3   J3    J2    100  200  100
 P4   J2    J4    100  200  100
 P5   J5    J1    100  200  100
 P6   J4    J5    100  200  100
 P7   J6    J3    100  200  100
 P8   T1    J4    100  200  100
 P9   J2    J4    100  150  100   ; parallel to P4

[PUMPS]
 PU1  J5    J6    HEAD 1

[VALVES]
 V1   J3    T2    150  PRV  30  0
 V2   J6    J1    150  TCV  1   0

[END]
"""


def lz4_compress(src: bytes) -> bytes:
    """Greedy LZ4 block encoder (4-byte hash matches; the last 5 bytes are always literals, as the format requires)."""
    n, out, anchor, i, table = len(src), bytearray(), 0, 0, {}

    def emit(lit: bytes, mlen: int, off: int):
        ll = len(lit)
        ml = mlen - 4 if mlen else 0
        out.append((min(ll, 15) << 4) | (min(ml, 15) if mlen else 0))
        if ll >= 15:
            r = ll - 15
            while r >= 255:
                out.append(255); r -= 255
            out.append(r)
        out.extend(lit)
        if mlen:
            out.extend(struct.pack("<H", off))
            if ml >= 15:
                r = ml - 15
                while r >= 255:
                    out.append(255); r -= 255
                out.append(r)

    while i + 4 <= n - 5:
        key = src[i:i + 4]
        cand = table.get(key)
        table[key] = i
        if cand is not None and i - cand <= 65535:
            m = 4
            while i + m < n - 5 and src[cand + m] == src[i + m]:
                m += 1
            emit(src[anchor:i], m, i - cand)
            i += m
            anchor = i
        else:
            i += 1
    emit(src[anchor:], 0, 0)
    return bytes(out)


def blosc_compress(data: bytes, typesize: int, blocksize: int, shuffle: bool, codec: str = "lz4") -> bytes:
    nbytes = len(data)
    nblocks = max(1, -(-nbytes // blocksize))
    flags = (1 if shuffle and typesize > 1 else 0) | ({"lz4": 1, "zlib": 3}[codec] << 5)
    body, bstarts = bytearray(), []
    header_len = 16 + 4 * nblocks
    for b in range(nblocks):
        blk = data[b * blocksize:(b + 1) * blocksize]
        bsize = len(blk)
        leftover = bsize != blocksize
        if flags & 1:
            n_el = bsize // typesize
            sh = np.frombuffer(blk[:n_el * typesize], dtype=np.uint8).reshape(n_el, typesize).T.reshape(-1).tobytes()
            blk = sh + blk[n_el * typesize:]
        nsplits = typesize if (not leftover and typesize <= 16 and bsize // typesize >= 128) else 1
        ne = bsize // nsplits
        bstarts.append(header_len + len(body))
        for s in range(nsplits):
            piece = blk[s * ne:(s + 1) * ne]
            c = lz4_compress(piece) if codec == "lz4" else zlib.compress(piece, 5)
            if len(c) >= len(piece):
                c = piece                                  # stored raw: its size says so
            body += struct.pack("<i", len(c)) + c
    cbytes = header_len + len(body)
    return struct.pack("<BBBBIII", 2, 1, flags, typesize, nbytes, blocksize, cbytes) + struct.pack(f"<{nblocks}i", *bstarts) + bytes(body)


def fixture_arrays():
    rs = np.random.RandomState(7)
    nodes = 9
    base = 30.0 + 5.0 * rs.rand(nodes)
    out = {}
    for split, s in (("train", 700), ("valid", 130), ("test", 64)):
        t = np.arange(s)[:, None]
        a = base[None, :] + 2.0 * np.sin(t / 11.0 + np.arange(nodes)[None, :]) + 0.01 * rs.randn(s, nodes)
        out[split] = np.round(a, 2).astype("<f4")            # rounded: compressible, like simulator output
    return out


def main():
    with open(os.path.join(HERE, "wdn_tiny.inp"), "w") as f:
        f.write(INP)
    arrays = fixture_arrays()
    path = os.path.join(HERE, "wdn_tiny.zip")
    with zipfile.ZipFile(path, "w", zipfile.ZIP_STORED) as z:
        z.writestr(".zgroup", json.dumps({"zarr_format": 2}))
        z.writestr(".zattrs", json.dumps({"note": "hand-made fixture"}))
        for feature, comp in (("pressure", {"id": "blosc", "cname": "lz4", "clevel": 5, "shuffle": 1, "blocksize": 0}),
                              ("demand", {"id": "zlib", "level": 1}), ("head", None)):
            z.writestr(f"{feature}/.zgroup", json.dumps({"zarr_format": 2}))
            for split, a in arrays.items():
                if feature != "pressure":
                    a = (a * (2.0 if feature == "demand" else -1.0)).astype("<f4")
                chunks = (256, a.shape[1])
                meta = {"zarr_format": 2, "shape": list(a.shape), "chunks": list(chunks), "dtype": "<f4", "order": "C",
                        "fill_value": 0.0, "filters": None, "compressor": comp}
                z.writestr(f"{feature}/{split}/.zarray", json.dumps(meta))
                for ci in range(-(-a.shape[0] // chunks[0])):
                    chunk = np.zeros(chunks, dtype="<f4")
                    part = a[ci * chunks[0]:(ci + 1) * chunks[0]]
                    chunk[:part.shape[0]] = part
                    raw = chunk.tobytes()
                    if comp is None:
                        data = raw
                    elif comp["id"] == "zlib":
                        data = zlib.compress(raw, 1)
                    else:                                   # 2 KiB blocks: several blocks, split into 4 byte planes each
                        data = blosc_compress(raw, 4, 2048, True, "lz4")
                    z.writestr(f"{feature}/{split}/{ci}.0", data)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()

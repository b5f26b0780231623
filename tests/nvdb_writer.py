"""Writer of NanoVDB 32.x FloatGrid files -- TEST INFRASTRUCTURE for host/vspg_nanovdb.{h,cpp}.

The reader's layout is "NanoVDB's published 32.x layout as understood by this build" (the NanoVDB headers are an absent
submodule of the reference and no .nvdb file exists in this environment): this writer shares that understanding, so a
round trip through it shows the reader is self-consistent and refuses what it says it refuses -- NOT that it reads files
written by NanoVDB itself.  Parity of the .nvdb reader stays "unpinned" (INTEGRATION.md 2).

write_nvdb(path, grids) with grids = [dict(name=..., values=ndarray[nx, ny, nz] float32, index_min=(i, j, k),
voxel_size=s, translate=(tx, ty, tz), background=0.0, tiles=[((x0, y0, z0), value), ...])]:
voxels equal to the background are inactive; 8^3 blocks without an active voxel are not stored; `tiles` adds active
lower-node tiles (constant 8^3 blocks) at index coordinates that must not overlap stored leaves.
"""
import struct

import numpy as np

MAGIC = 0x304244566F6E614E  # "NanoVDB0"
VERSION = (32 << 21) | (3 << 10) | 3


def _pad(b, n):
    assert len(b) <= n, (len(b), n)
    return b + b"\0" * (n - len(b))


def _mask(bits, words):
    m = [0] * words
    for n in bits:
        m[n >> 6] |= 1 << (n & 63)
    return struct.pack("<%dQ" % words, *m)


def _grid_blob(g, grid_index, grid_count, version=VERSION, grid_type=1):
    vals = np.asarray(g["values"], dtype=np.float32)
    imin = np.array(g.get("index_min", (0, 0, 0)), dtype=np.int64)
    bg = np.float32(g.get("background", 0.0))
    nx, ny, nz = vals.shape
    # --- leaves: 8^3 blocks (aligned in index space) holding at least one active voxel
    leaves = {}  # (x0, y0, z0) -> float32[512] in leaf order (x << 6 | y << 3 | z)
    lo = (imin // 8) * 8
    hi = ((imin + np.array(vals.shape) - 1) // 8) * 8
    for x0 in range(lo[0], hi[0] + 1, 8):
        for y0 in range(lo[1], hi[1] + 1, 8):
            for z0 in range(lo[2], hi[2] + 1, 8):
                block = np.full((8, 8, 8), bg, dtype=np.float32)
                xs = slice(max(x0, imin[0]) - imin[0], min(x0 + 8, imin[0] + nx) - imin[0])
                ys = slice(max(y0, imin[1]) - imin[1], min(y0 + 8, imin[1] + ny) - imin[1])
                zs = slice(max(z0, imin[2]) - imin[2], min(z0 + 8, imin[2] + nz) - imin[2])
                sub = vals[xs, ys, zs]
                block[xs.start + imin[0] - x0: xs.stop + imin[0] - x0, ys.start + imin[1] - y0: ys.stop + imin[1] - y0,
                      zs.start + imin[2] - z0: zs.stop + imin[2] - z0] = sub
                if np.any(block != bg):
                    leaves[(x0, y0, z0)] = block.reshape(-1)
    tiles = {tuple(int(c) for c in k): np.float32(v) for k, v in g.get("tiles", [])}
    for k in tiles:
        assert all(c % 8 == 0 for c in k) and k not in leaves, k
    lowers = {}  # (x0, y0, z0) aligned to 128 -> {child index: ("leaf", key) | ("tile", value)}
    for k in list(leaves) + list(tiles):
        lk = tuple((c // 128) * 128 for c in k)
        n = ((k[0] >> 3) & 15) << 8 | ((k[1] >> 3) & 15) << 4 | ((k[2] >> 3) & 15)
        lowers.setdefault(lk, {})[n] = ("leaf", k) if k in leaves else ("tile", tiles[k])
    uppers = {}
    for lk in lowers:
        uk = tuple((c // 4096) * 4096 for c in lk)
        n = ((lk[0] >> 7) & 31) << 10 | ((lk[1] >> 7) & 31) << 5 | ((lk[2] >> 7) & 31)
        uppers.setdefault(uk, {})[n] = lk
    # --- sizes and offsets (relative to the blob): GridData, TreeData, root, uppers, lowers, leaves
    GRID, TREE, ROOTH, RTILE, UPH, LOWH, LEAF = 672, 64, 64, 32, 8256, 1088, 2144
    UPB, LOWB = UPH + 32768 * 8, LOWH + 4096 * 8
    root_off = GRID + TREE
    upper_off0 = root_off + ROOTH + RTILE * len(uppers)
    ukeys, lkeys, fkeys = sorted(uppers), sorted(lowers), sorted(leaves)
    upper_off = {k: upper_off0 + UPB * i for i, k in enumerate(ukeys)}
    lower_off0 = upper_off0 + UPB * len(ukeys)
    lower_off = {k: lower_off0 + LOWB * i for i, k in enumerate(lkeys)}
    leaf_off0 = lower_off0 + LOWB * len(lkeys)
    leaf_off = {k: leaf_off0 + LEAF * i for i, k in enumerate(fkeys)}
    total = leaf_off0 + LEAF * len(fkeys)
    total = (total + 31) // 32 * 32
    active = sum(int(np.count_nonzero(v != bg)) for v in leaves.values()) + 512 * len(tiles)
    # --- GridData
    s = float(g.get("voxel_size", 1.0))
    t = [float(c) for c in g.get("translate", (0.0, 0.0, 0.0))]
    mat = g.get("mat", [s, 0, 0, 0, s, 0, 0, 0, s])
    inv = list(np.linalg.inv(np.array(mat, dtype=np.float64).reshape(3, 3)).reshape(-1))
    wmin = [imin[i] * mat[4 * i] + t[i] for i in range(3)]
    wmax = [(imin[i] + vals.shape[i]) * mat[4 * i] + t[i] for i in range(3)]
    name = g["name"].encode() + b"\0"
    gd = struct.pack("<QQIIIIQ", MAGIC, 0, version, 0, grid_index, grid_count, total) + _pad(name, 256)
    gd += struct.pack("<9f9f3ff", *mat, *inv, *t, 1.0) + struct.pack("<9d9d3dd", *mat, *inv, *t, 1.0)
    gd += struct.pack("<6d3d", *wmin, *wmax, s, s, s) + struct.pack("<IIqI", 1, grid_type, 0, 0)
    out = bytearray(_pad(gd, GRID))
    # --- TreeData: node offsets relative to TreeData {leaf, lower, upper, root}
    out += _pad(struct.pack("<4Q3I3IQ", leaf_off0 - GRID, lower_off0 - GRID, upper_off0 - GRID, root_off - GRID, len(fkeys), len(lkeys),
                            len(ukeys), len(tiles), 0, 0, active), TREE)
    # --- root
    bbox = [int(c) for c in imin] + [int(imin[i] + vals.shape[i] - 1) for i in range(3)]
    vmin, vmax = (float(vals.min()), float(vals.max())) if vals.size else (0.0, 0.0)
    out += _pad(struct.pack("<6iI5f", *bbox, len(ukeys), float(bg), vmin, vmax, float(vals.mean()), float(vals.std())), ROOTH)
    for k in ukeys:
        key = (((k[0] & 0xFFFFFFFF) >> 12) << 42) | (((k[1] & 0xFFFFFFFF) >> 12) << 21) | ((k[2] & 0xFFFFFFFF) >> 12)
        out += _pad(struct.pack("<QqIf", key, upper_off[k] - root_off, 0, 0.0), RTILE)
    for k in ukeys:
        kids = uppers[k]
        node = struct.pack("<6iQ", *k, *(c + 4095 for c in k), 0) + _mask([], 512) + _mask(kids.keys(), 512) + struct.pack("<4f", vmin, vmax, 0, 0)
        out += _pad(node, UPH)
        table = np.zeros(32768, dtype=np.int64)
        for n, lk in kids.items():
            table[n] = lower_off[lk] - upper_off[k]
        out += table.tobytes()
    for k in lkeys:
        kids = lowers[k]
        node = struct.pack("<6iQ", *k, *(c + 127 for c in k), 0) + _mask([n for n, (kind, _) in kids.items() if kind == "tile"], 64)
        node += _mask([n for n, (kind, _) in kids.items() if kind == "leaf"], 64) + struct.pack("<4f", vmin, vmax, 0, 0)
        out += _pad(node, LOWH)
        table = np.zeros(4096, dtype=np.int64)
        for n, (kind, v) in kids.items():
            if kind == "leaf":
                table[n] = leaf_off[v] - lower_off[k]
            else:
                table[n] = int(np.frombuffer(np.float32(v).tobytes(), dtype=np.uint32)[0])  # the union's value member (low 4 bytes)
        out += table.tobytes()
    for k in fkeys:
        v = leaves[k]
        head = struct.pack("<3i3BB", *k, 7, 7, 7, 0) + _mask(np.nonzero(v != bg)[0].tolist(), 8) + struct.pack("<4f", float(v.min()), float(v.max()), 0, 0)
        out += _pad(head, 96) + v.astype("<f4").tobytes()
    out = bytes(_pad(bytes(out), total))
    meta = dict(grid_size=total, voxel_count=active, wbbox=wmin + wmax, ibbox=bbox, voxel=(s, s, s), name=name,
                node_count=(len(fkeys), len(lkeys), len(ukeys), 1), tile_count=(len(tiles), 0, 0), grid_type=grid_type)
    return out, meta


def write_nvdb(path, grids, version=VERSION, codec=0, grid_type=1, segments=1):
    """`segments` > 1 writes one file segment per grid (NanoVDB appends a segment per write call); 1 puts all grids in one."""
    groups = [[g] for g in grids] if segments > 1 else [grids]
    with open(path, "wb") as f:
        for group in groups:
            blobs = [_grid_blob(g, i, len(group), version, grid_type) for i, g in enumerate(group)]
            f.write(struct.pack("<QIHH", MAGIC, version, len(group), codec))
            for blob, m in blobs:
                md = struct.pack("<4Q2I", m["grid_size"], len(blob), 0, m["voxel_count"], m["grid_type"], 1)
                md += struct.pack("<6d6i3d", *m["wbbox"], *m["ibbox"], *m["voxel"])
                md += struct.pack("<I4I3IHHI", len(m["name"]), *m["node_count"], *m["tile_count"], codec, 0, version)
                assert len(md) == 176, len(md)
                f.write(md + m["name"])
            for blob, _ in blobs:
                f.write(blob)

"""host/vspg_nanovdb.{h,cpp}: the ".nvdb" reader behind `MakeNamedMedium ... "string type" "nanovdb"` (NanoVDBMedium::Create,
media.cpp:683-734).  PARITY UNPINNED: no NanoVDB header or file exists in this environment, so these tests hold the reader
against tests/nvdb_writer.py, a writer of the SAME understanding of the 32.x layout -- they show self-consistency, the decode
of every node level, and that what the reader does not know it refuses by name.  They do not show that a file written by
NanoVDB itself is read correctly; the checkable route for real clouds remains the reference's nanovdb2pbrt -> "uniformgrid".
Set VSPG_TEST_NVDB=/path/to/file.nvdb to run the reader over a real file (prints what it decoded)."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT
from nvdb_writer import VERSION, write_nvdb

HOST = os.path.join(ROOT, "vspg-pbrt-v4_amd", "host")


@pytest.fixture(scope="module")
def tool():
    subprocess.check_call(["make", "-C", HOST, "vspg_nvdb2grid"])
    return os.path.join(HOST, "vspg_nvdb2grid")


def read_back(tool, path, grid, tmp_path):
    dump = str(tmp_path / "dump.f32")
    r = subprocess.run([tool, str(path), "--grid", grid, "--dump", dump], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    info = json.loads(r.stdout)
    nx, ny, nz = info["dim"]
    return info, np.fromfile(dump, dtype=np.float32).reshape(nz, ny, nx).transpose(2, 1, 0)


def sparse_values(shape, seed, empty=0.4):
    rng = np.random.default_rng(seed)
    v = rng.random(shape, dtype=np.float32) + np.float32(0.01)
    v[rng.random(shape) < empty] = 0
    return v


@pytest.mark.parametrize("shape,index_min", [((20, 13, 9), (-5, 3, -17)), ((8, 8, 8), (0, 0, 0)), ((1, 1, 1), (7, -8, 127)),
                                              ((140, 9, 10), (-130, 120, -3)),          # crosses lower-node (128) borders
                                              ((12, 10, 9), (4090, -4100, 4))])        # crosses upper-node (4096) borders: 4 root tiles
def test_round_trip_every_level(tool, tmp_path, shape, index_min):
    v = sparse_values(shape, 3)
    v[0, 0, 0] = v[-1, -1, -1] = 0.5  # the index bbox is tight
    path = tmp_path / "a.nvdb"
    write_nvdb(path, [dict(name="density", values=v, index_min=index_min, voxel_size=0.25, translate=(1.0, -2.0, 3.5))])
    info, d = read_back(tool, path, "density", tmp_path)
    assert info["index_min"] == list(index_min) and info["dim"] == list(shape)
    assert info["scale"] == [0.25] * 3 and info["translate"] == [1.0, -2.0, 3.5] and info["voxel_size"] == [0.25] * 3
    assert info["world_min"] == [index_min[k] * 0.25 + (1.0, -2.0, 3.5)[k] for k in range(3)]
    assert info["world_max"] == [(index_min[k] + shape[k]) * 0.25 + (1.0, -2.0, 3.5)[k] for k in range(3)]
    assert info["active_voxels"] == int(np.count_nonzero(v))
    assert np.array_equal(d.view(np.uint32), v.view(np.uint32))


def test_background_and_active_tiles(tool, tmp_path):
    """Voxels no node stores read as the root's background; an active lower-node tile reads as its constant, clipped to the bbox."""
    v = np.full((30, 20, 18), 0.125, dtype=np.float32)   # background 0.125 everywhere ...
    v[2:5, 3:9, 1:4] = sparse_values((3, 6, 3), 5, empty=0.0)
    v[29, 19, 17] = 2.0
    path = tmp_path / "b.nvdb"
    write_nvdb(path, [dict(name="density", values=v, index_min=(0, 0, 0), background=0.125, tiles=[((16, 8, 8), 0.75), ((24, 16, 0), 1.5)])])
    info, d = read_back(tool, path, "density", tmp_path)
    want = v.copy()
    want[16:24, 8:16, 8:16] = 0.75
    want[24:30, 16:20, 0:8] = 1.5   # the tile pokes out of the index bbox: clipped
    assert info["background"] == 0.125
    # (29, 19, 17) lies in the leaf at (24, 16, 16), not in the tile at (24, 16, 0)
    assert d[29, 19, 17] == 2.0
    assert np.array_equal(d, want)


@pytest.mark.parametrize("segments", [1, 2])
def test_named_grids_and_segments(tool, tmp_path, segments):
    dens, temp = sparse_values((9, 9, 9), 7), sparse_values((9, 9, 9), 8) * 1000
    path = tmp_path / "c.nvdb"
    write_nvdb(path, [dict(name="density", values=dens), dict(name="temperature", values=temp)], segments=segments)
    assert np.array_equal(read_back(tool, path, "density", tmp_path)[1], dens)
    assert np.array_equal(read_back(tool, path, "temperature", tmp_path)[1], temp)
    r = subprocess.run([tool, str(path), "--grid", "flames"], capture_output=True, text=True)
    assert r.returncode == 1 and "didn't find \"flames\" grid" in r.stderr   # media.cpp:541


@pytest.mark.parametrize("kwargs,needle", [
    (dict(version=(31 << 21) | (3 << 10)), "version 31.3.0"),
    (dict(version=(33 << 21)), "version 33.0.0"),
    (dict(codec=1), "compressed NanoVDB file (codec 1)"),
    (dict(codec=2), "compressed NanoVDB file (codec 2)"),
    (dict(grid_type=2), "not a FloatGrid (grid type 2)"),
])
def test_refuses_what_it_does_not_know(tool, tmp_path, kwargs, needle):
    path = tmp_path / "d.nvdb"
    write_nvdb(path, [dict(name="density", values=sparse_values((9, 9, 9), 9))], **kwargs)
    r = subprocess.run([tool, str(path)], capture_output=True, text=True)
    assert r.returncode == 1 and needle in r.stderr, r.stderr


def test_refuses_damaged_files(tool, tmp_path):
    path = tmp_path / "e.nvdb"
    write_nvdb(path, [dict(name="density", values=sparse_values((9, 9, 9), 9))])
    blob = open(path, "rb").read()
    for cut in (8, 100, 300, len(blob) // 2, len(blob) - 1):
        (tmp_path / "cut.nvdb").write_bytes(blob[:cut])
        r = subprocess.run([tool, str(tmp_path / "cut.nvdb")], capture_output=True, text=True)
        assert r.returncode == 1 and ("truncated" in r.stderr or "didn't find" in r.stderr), (cut, r.stderr)
    (tmp_path / "magic.nvdb").write_bytes(b"NanoVDB1" + blob[8:])
    r = subprocess.run([tool, str(tmp_path / "magic.nvdb")], capture_output=True, text=True)
    assert r.returncode == 1 and "not a NanoVDB file" in r.stderr
    r = subprocess.run([tool, str(tmp_path / "absent.nvdb")], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr
    # a child offset pointing outside the grid must be an error, not a wild read
    bad = bytearray(blob)
    hdr = 16 + 176 + len(b"density\0")
    root_tile = hdr + 672 + 64 + 64
    bad[root_tile + 8: root_tile + 16] = (1 << 40).to_bytes(8, "little")
    (tmp_path / "wild.nvdb").write_bytes(bytes(bad))
    r = subprocess.run([tool, str(tmp_path / "wild.nvdb")], capture_output=True, text=True)
    assert r.returncode == 1 and ("outside the grid" in r.stderr or "truncated" in r.stderr), r.stderr
    # ... including offsets chosen to wrap the position arithmetic (near 2^64 as unsigned, hugely negative) and misaligned ones
    for off in ((1 << 63) - 32, -(1 << 62), -(len(blob) * 4 // 32 * 32), 8256 + 1):
        bad = bytearray(blob)
        bad[root_tile + 8: root_tile + 16] = int(off).to_bytes(8, "little", signed=True)
        (tmp_path / "wrap.nvdb").write_bytes(bytes(bad))
        r = subprocess.run([tool, str(tmp_path / "wrap.nvdb")], capture_output=True, text=True)
        assert r.returncode == 1 and ("outside the grid" in r.stderr or "truncated" in r.stderr), (off, r.returncode, r.stderr)
    # an index bounding box whose extent does not fit 32 bits (max - min + 1 overflows int32)
    bad = bytearray(blob)
    meta = 16
    bad[meta + 88: meta + 92] = (-(1 << 31)).to_bytes(4, "little", signed=True)
    bad[meta + 100: meta + 104] = ((1 << 31) - 1).to_bytes(4, "little", signed=True)
    (tmp_path / "bbox.nvdb").write_bytes(bytes(bad))
    r = subprocess.run([tool, str(tmp_path / "bbox.nvdb")], capture_output=True, text=True)
    assert r.returncode == 1 and "too large" in r.stderr, r.stderr


SCENE = """LookAt 0 0 -0.95   0 0 0   0 1 0
MediumInterface "" "smoke"
Camera "perspective" "float fov" 60
Sampler "independent" "integer pixelsamples" 4
PixelFilter "box"
Film "rgb" "integer xresolution" 64 "integer yresolution" 48 "string filename" "nvdb_box.pfm"
Integrator "guidedvolpathvspg" "integer maxdepth" 5 "bool vspguiding" true "bool surfaceguiding" false
    "bool volumeguiding" false "bool vspsecondaryguiding" false
Option "string rendercoordsys" "world"
WorldBegin
MakeNamedMedium "smoke" "string type" "nanovdb" "string filename" "grids/smoke.nvdb" "rgb sigma_a" [ .05 .08 .1 ] "rgb sigma_s" [ 3 2.6 2.2 ]
    "float g" 0.5 %s
MediumInterface "smoke" "smoke"
Material "diffuse" "rgb reflectance" [ .73 .73 .73 ]
Shape "bilinearmesh" "point3 P" [ -1 -1 -1   -1 -1 1    1 -1 -1    1 -1 1 ]
Shape "bilinearmesh" "point3 P" [ -1 1 -1     1 1 -1   -1 1 1      1 1 1 ]
Shape "bilinearmesh" "point3 P" [ -1 -1 1    -1 1 1     1 -1 1     1 1 1 ]
Shape "bilinearmesh" "point3 P" [ -1 -1 -1    1 -1 -1  -1 1 -1     1 1 -1 ]
Shape "bilinearmesh" "point3 P" [ -1 -1 -1   -1 1 -1   -1 -1 1    -1 1 1 ]
Shape "bilinearmesh" "point3 P" [ 1 -1 -1     1 -1 1    1 1 -1     1 1 1 ]
AttributeBegin
  Material "diffuse" "rgb reflectance" [ 0 0 0 ]
  AreaLightSource "diffuse" "rgb L" [ 17 12 4 ]
  Shape "bilinearmesh" "point3 P" [ -0.25 0.999 -0.25   0.25 0.999 -0.25   -0.25 0.999 0.25   0.25 0.999 0.25 ]
AttributeEnd
"""
VOX, IMIN, ORIGIN, N = (0.066, 0.0625, 0.058), (-3, 2, 0), (-0.6, -0.93, -0.5), (24, 24, 24)


def write_smoke_scene(tmp_path, extra=""):
    from scenes import cloud_density
    dens = cloud_density(24)
    os.makedirs(tmp_path / "grids", exist_ok=True)
    mat = [VOX[0], 0, 0, 0, VOX[1], 0, 0, 0, VOX[2]]
    write_nvdb(tmp_path / "grids" / "smoke.nvdb", [dict(name="density", values=dens.reshape(N[2], N[1], N[0]).transpose(2, 1, 0), index_min=IMIN,
                                                       mat=mat, translate=ORIGIN)])
    (tmp_path / "scene.pbrt").write_text(SCENE % extra)
    return dens


def test_scene_file_with_a_nanovdb_medium_parses(tmp_path):
    """The filename resolves against the scene file's directory (ResolveFilename, media.cpp:686); unknown parameters are errors."""
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "vspg-pbrt-v4_amd", "csrc"), "libvspg_hip.so"])
    subprocess.check_call(["make", "-C", HOST])
    exe = os.path.join(HOST, "vspg_pbrt")
    write_smoke_scene(tmp_path)
    r = subprocess.run([exe, str(tmp_path / "scene.pbrt"), "--parse-only"], capture_output=True, text=True, cwd="/")
    assert r.returncode == 0, r.stderr
    assert "medium type 3" in r.stdout   # VSPG_MEDIUM_NANOVDB
    (tmp_path / "scene.pbrt").write_text(SCENE % '"string gridname" "dens"')
    r = subprocess.run([exe, str(tmp_path / "scene.pbrt"), "--parse-only"], capture_output=True, text=True)
    assert r.returncode == 1 and "didn't find \"dens\" grid" in r.stderr
    (tmp_path / "scene.pbrt").write_text(SCENE % '"float denstiyoffset" 1')
    r = subprocess.run([exe, str(tmp_path / "scene.pbrt"), "--parse-only"], capture_output=True, text=True)
    assert r.returncode == 1 and "denstiyoffset" in r.stderr


@pytest.mark.gpu
def test_scene_file_nanovdb_render_equals_the_api_scene(gpu_pkg, tmp_path):
    """`vspg_pbrt scene.pbrt` with a "nanovdb" medium read from a .nvdb file == the same NanoVDBMedium-semantics medium handed over
    through the C-ABI as a dense array, bit for bit (the device path itself is the one tests/test_gpu_parity.py holds to the oracle)."""
    from scenes import nvdb_scene
    from test_host_adapter import read_pfm
    subprocess.check_call(["make", "-C", HOST])
    dens = write_smoke_scene(tmp_path, '"float densityoffset" 0.02 "float majorantscale" 1.25')
    out = tmp_path / "o.pfm"
    a = subprocess.run([os.path.join(HOST, "vspg_pbrt"), str(tmp_path / "scene.pbrt"), "--outfile", str(out)], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout + a.stderr
    P = gpu_pkg
    W, H, spp = 64, 48, 4
    scene = nvdb_scene(dens, N, (0.05, 0.08, 0.1), (3.0, 2.6, 2.2), g=0.5, index_min=IMIN, voxel=VOX, origin=ORIGIN, density_offset=0.02,
                       majorant_scale=1.25, W=W, H=H)
    r = P.Renderer(scene, P.app_f_params(), W, H)
    for w in range(spp):
        r.render_wave(w, w + 1)
        r.post_process_wave()
    f = r.film()
    r.close()
    assert np.array_equal(read_pfm(str(out)).view(np.uint32), (f[..., :3] / f[..., 3:4]).astype(np.float32).view(np.uint32))


@pytest.mark.skipif(not os.environ.get("VSPG_TEST_NVDB"), reason="set VSPG_TEST_NVDB=/path/to/file.nvdb to read a real file")
def test_a_real_file(tool):
    r = subprocess.run([tool, os.environ["VSPG_TEST_NVDB"], "--grid", os.environ.get("VSPG_TEST_NVDB_GRID", "density")], capture_output=True, text=True)
    print(r.stdout, r.stderr)
    assert r.returncode == 0

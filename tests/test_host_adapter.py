"""C++ host adapter (vspg-pbrt-v4_amd/host): the reference's Integrator / Medium plugin surface
over the C-ABI.  CPU part: parameter names, defaults, unused-parameter and registry errors
(host_selftest, no device calls).  GPU part: the App.-F scene rendered through
Integrator::Create("guidedvolpathvspg") equals the same render driven through the raw C-ABI."""
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

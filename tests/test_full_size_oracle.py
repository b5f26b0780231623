"""One wave at BASELINE.json's full size (1920 x 1080) on the GPU's wave kernels against the CPU oracle, pixel for pixel.

The oracle renders a 1080p wave in 5-9 s on the GPU box's host cores (OpenMP), so the full-size check need not stop at
size-independent properties (tests/test_gpu_parity.py::test_full_size_*): with one sample per pixel the film IS the paths' radiances --
float on both sides -- and every one of the 2 073 600 pixels must hold the oracle's bits, and the counters must be equal.  This is
where the schedulers show: the barrier-free workgroup kernel and its tile cursors, the pipeline's walk kernels and job cursors, the
regrouped no-walk chains of boundary scenes, the guided vertex kernels.  (scripts/full_size_film_check.py runs all eight bench
workloads; here: one of each kernel family.)"""
import numpy as np
import pytest

import oracle_lib

W, H = 1920, 1080


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["fog", "cloud", "cloud-scene", "cloud-scene-guided"])
def test_full_size_wave_bit_identical_to_oracle(gpu_pkg, workload):
    P = gpu_pkg
    guided = workload.endswith("-guided")
    base = workload[:-7] if guided else workload
    scene = P.fog_box_scene(W, H) if base == "fog" else P.cloud_box_scene(W, H, 256) if base == "cloud" else P.cloud_scene(W, H, 256)
    prm = P.default_params() if guided else P.app_f_params()
    g = P.Renderer(scene, prm, W, H, seed=0)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=0)
    if guided:
        import scenes
        field = scenes.light_field(P, n=4)
        g.set_guiding_field(field, field)
        c.set_guiding_field(field, field)
    expected = {"fog": "k_render_wave_wg3", "cloud": "k_wf_dist_walk", "cloud-scene": "k_wf_walk", "cloud-scene-guided": "k_wf_walk"}[workload]
    assert g.kernel_name().startswith(expected), g.kernel_name()
    g.render_wave(0, 1)
    c.render_wave(0, 1, 0)
    fg, fc = g.film(), c.film()
    same = np.all(fg.view(np.uint32) == fc.view(np.uint32), axis=-1)
    assert same.all(), (workload, int(same.sum()), same.size, np.argwhere(~same)[:5])
    assert g.counters() == c.counters()
    g.close()
    c.close()

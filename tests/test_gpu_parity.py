"""GPU parity tests: the HIP path, called through the C-ABI, against the CPU oracle on the same
seeded inputs.  Bars: bit-exact for integer work (RNG, hashes) and for FastExp; for path radiance
the tolerance is stated per test (libm differences: glibc logf/sinf/cosf on the host vs
double-evaluated-and-rounded on device can differ in the last ulp)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib
from conftest import ROOT

pytestmark = pytest.mark.gpu
fh = float.fromhex


@pytest.fixture(scope="module")
def pair(gpu_pkg):
    W, H = 96, 64
    scene = gpu_pkg.fog_box_scene(W, H)
    prm = gpu_pkg.app_f_params()
    g = gpu_pkg.Renderer(scene, prm, W, H)
    c = oracle_lib.OracleRenderer(scene, prm, W, H)
    yield gpu_pkg, g, c
    g.close()
    c.close()


def test_primitives_bit_exact(pair):
    P, g, c = pair
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "primitives.json")))
    rng = np.random.default_rng(0)
    f = np.concatenate([np.array([fh(x) for x, _ in G["hash_float"]], dtype=np.float32),
                        np.array([fh(x) for x, _ in G["fast_exp"]], dtype=np.float32),
                        rng.uniform(-30, 5, 20000).astype(np.float32), rng.random(20000).astype(np.float32)])
    gg = rng.random(f.shape[0]).astype(np.float32)
    h, r, e = g.primitives_batch(f, gg)
    lib = oracle_lib.load()
    import ctypes as C
    for i in range(f.shape[0]):
        assert int(h[i]) == lib.oracle_hash_float(float(f[i]))
    u = (C.c_uint32 * 1)()
    for i in range(0, f.shape[0], 7):
        lib.oracle_rng_seq(lib.oracle_hash_float(float(f[i])), lib.oracle_hash_float(float(gg[i])), 1, 0, 1, u, None)
        assert int(r[i]) == u[0]
    eo = np.array([lib.oracle_fast_exp(float(x)) for x in f], dtype=np.float32)
    assert np.array_equal(e.view(np.uint32), eo.view(np.uint32))
    # golden vectors straight from the reference build
    n = len(G["hash_float"])
    assert [int(x) for x in h[:n]] == [int(v, 16) for _, v in G["hash_float"]]
    m = len(G["fast_exp"])
    ref = np.array([fh(y) for _, y in G["fast_exp"]], dtype=np.float32)
    assert np.array_equal(e[n:n + m].view(np.uint32), ref.view(np.uint32))


def test_device_libm_equals_host_libm(pair, libm_shim):
    # the kernels' logf/sinf/cosf must reproduce the HOST libm bit for bit (see csrc/vspg_libm.h)
    import ctypes as C
    P, g, c = pair
    rng = np.random.default_rng(7)
    two_pi = np.float32(2) * np.float32(np.pi)
    u = np.minimum(rng.integers(0, 2 ** 32, 1_000_000, dtype=np.uint64).astype(np.float32) * np.float32(2.0 ** -32),
                   np.float32(float.fromhex("0x1.fffffep-1")))
    x = np.concatenate([np.float32(1) - u, two_pi * rng.random(1_000_000, dtype=np.float32),
                        rng.uniform(-np.pi / 4, 3 * np.pi / 4, 1_000_000).astype(np.float32),
                        rng.uniform(1e-3, 100.0, 500_000).astype(np.float32)])
    lo, so, co = g.libm_batch(x)
    fp = C.POINTER(C.c_float)
    for name, dev in (("libm_logf", lo), ("libm_sinf", so), ("libm_cosf", co)):
        ref = np.empty_like(x)
        getattr(libm_shim, name)(x.shape[0], x.ctypes.data_as(fp), ref.ctypes.data_as(fp))
        diff = dev.view(np.uint32) != ref.view(np.uint32)
        if name == "libm_logf":
            diff &= x > 0  # the path only takes logs of positive normal floats
        bad = np.nonzero(diff)[0]
        assert bad.size == 0, (name, bad.size, x[bad[:3]], dev[bad[:3]], ref[bad[:3]])


def test_device_double_log_matches_host_libm(pair, libm_shim):
    """-std::log(1.0 - x) in double, rounded to float (media_sampleTMaj.h:379-404): device == host libm."""
    P, g, c = pair
    rng = np.random.default_rng(11)
    u = np.minimum(rng.integers(0, 2 ** 32, 1_000_000, dtype=np.uint64).astype(np.float32) * np.float32(2.0 ** -32),
                   np.float32(float.fromhex("0x1.fffffep-1")))
    x = np.concatenate([u, u * rng.random(u.shape[0], dtype=np.float32),
                        np.ldexp(rng.random(500_000), rng.integers(-24, 0, 500_000)).astype(np.float32),
                        np.array([0.0, 0.0625, 0.5, float.fromhex("0x1.fffffep-1")], dtype=np.float32)])
    dev = g.libm_log1m_batch(x)
    ref = np.empty_like(x)
    fp = C.POINTER(C.c_float)
    libm_shim.libm_neg_log1m(x.shape[0], x.ctypes.data_as(fp), ref.ctypes.data_as(fp))
    bad = np.nonzero(dev.view(np.uint32) != ref.view(np.uint32))[0]
    assert bad.size == 0, (bad.size, x[bad[:3]], dev[bad[:3]], ref[bad[:3]])


def _queries(P, n, seed, vsp=None):
    rng = np.random.default_rng(seed)
    qs = []
    for i in range(n):
        d = rng.normal(size=3)
        d = d / np.linalg.norm(d) * rng.uniform(0.5, 2.0)
        qs.append(P.VspgTmajQuery(P.f3(*rng.uniform(-1, 1, 3)), P.f3(*d), float(rng.uniform(0.0, 4.0)), float(rng.random()),
                                  float(rng.random()), float(rng.random()),
                                  float(rng.random()) if vsp is None else vsp, int(rng.integers(0, 3)), 1))
    return qs


def test_free_flight_known_answers_on_device(pair):
    P, g, c = pair
    # SURVEY.md App. D.3 (reference's own output): t = 0x1.4498p+1, r_u_factor = 0x1.03d048p+0
    for k, ch in enumerate((0, 1, 1, 2)):
        q = P.VspgTmajQuery(P.f3(0, 0, 0), P.f3(0, 0, 1), 3.0, 0.37, 0.25 + k, 0.75, 0.8, ch, 1)
        o = g.sample_tmaj_batch(P.TMAJ_OPTICAL_DEPTH, [q])[0]
        assert o.n_callbacks == 1
        assert o.last_t == fh("0x1.4498p+1")
        assert list(o.r_u_factor) == [fh("0x1.03d048p+0")] * 3
        assert list(o.T_maj) == [1.0, 1.0, 1.0]


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_free_flight_vs_oracle(pair, variant):
    P, g, c = pair
    qs = _queries(P, 20000, 10 + variant) + _queries(P, 2000, 20 + variant, vsp=-1.0)
    go = g.sample_tmaj_batch(variant, qs)
    co = c.sample_tmaj_batch(variant, qs)
    exact = 0
    for a, b in zip(go, co):
        assert a.n_callbacks == b.n_callbacks
        # tolerance: 2 ulp of the double->float rounded log (1.2e-7 relative) on t and the factors
        assert abs(a.last_t - b.last_t) <= 3e-7 * max(1.0, abs(b.last_t))
        for k in range(3):
            assert abs(a.r_u_factor[k] - b.r_u_factor[k]) <= 3e-7 * abs(b.r_u_factor[k])
            assert abs(a.T_maj[k] - b.T_maj[k]) <= 3e-7 * abs(b.T_maj[k]) + 1e-30
        assert abs(a.vrc - b.vrc) <= 3e-7 * abs(b.vrc) and abs(a.majorant_scale - b.majorant_scale) <= 3e-7 * b.majorant_scale
        exact += (a.last_t == b.last_t and list(a.r_u_factor) == list(b.r_u_factor) and list(a.T_maj) == list(b.T_maj))
    # the optical-depth-space variant uses double log on both sides: expect (almost) all bit-equal
    frac = exact / len(qs)
    print("variant", variant, "bit-identical fraction", frac)
    assert frac == 1.0


def test_free_flight_edge_cases(pair):
    P, g, c = pair
    cases = [P.VspgTmajQuery(P.f3(0, 0, 0), P.f3(0, 0, 1), 0.0, 0.5, 0.1, 0.2, 0.5, 0, 0),        # empty segment
             P.VspgTmajQuery(P.f3(0, 0, 0), P.f3(0, 0, 1), 1e-6, 0.999, 0.1, 0.2, 0.999, 1, 0),   # tiny depth, max vsp
             P.VspgTmajQuery(P.f3(0, 0, 0), P.f3(0, 0, 1), 50.0, 0.0, 0.1, 0.2, 0.0, 2, 0),       # deep, min vsp, u=0
             P.VspgTmajQuery(P.f3(0, 0, 0), P.f3(0, 0, 3), 1.0, 0.9999999, 0.1, 0.2, 1.0, 0, 0)]  # u -> 1
    for variant in (0, 1, 2):
        for a, b in zip(g.sample_tmaj_batch(variant, cases), c.sample_tmaj_batch(variant, cases)):
            assert a.n_callbacks == b.n_callbacks
            assert np.allclose(list(a.T_maj), list(b.T_maj), rtol=3e-7, atol=0)
            assert np.allclose(list(a.r_u_factor), list(b.r_u_factor), rtol=3e-7, atol=0)
    assert g.sample_tmaj_batch(1, []) == []


def test_paths_vs_oracle(pair):
    P, g, c = pair
    rng = np.random.default_rng(3)
    n = 40000
    pix = np.stack([rng.integers(0, g.xres, n), rng.integers(0, g.yres, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    same_len = np.mean(sg == sc)
    # per-path tolerance: 1e-4 relative (+1e-6 absolute); a path whose branch flipped because of a
    # last-ulp libm difference may deviate arbitrarily -- such paths must stay below 0.2 %
    ok = np.all(np.abs(Lg - Lc) <= 1e-4 * np.abs(Lc) + 1e-6, axis=1)
    exact = np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1)
    print("paths: same segment count %.5f, within tol %.5f, bit-identical %.5f" % (same_len, ok.mean(), exact.mean()))
    assert same_len == 1.0 and exact.mean() == 1.0  # bit-identical paths
    # the estimator means must agree far inside the Monte-Carlo noise
    assert np.allclose(Lg.mean(0), Lc.mean(0), rtol=2e-3)


PARAM_SWEEP = [
    dict(vspsamplingmethod=1),                                  # "nds"
    dict(vspsamplingmethod=1, vspmisratio=0.2),
    dict(vspcriterion=0),                                       # "contribution"
    dict(usenee=0),
    dict(maxdepth=1), dict(maxdepth=9, minrrdepth=3), dict(maxdepth=0),
    dict(vspguiding=0), dict(vspprimaryguiding=0), dict(vspmisratio=1.0), dict(vspmisratio=0.0),
    dict(_g=0.75), dict(_g=-0.4, _Le=(0.3, 0.2, 0.1)),          # anisotropic phase function, emissive medium
    dict(_sigma=((0.3, 0.1, 0.02), (0.2, 0.9, 1.6))),           # chromatic medium -> generic (non-grey) instantiation
]


@pytest.mark.parametrize("case", range(len(PARAM_SWEEP)))
def test_parameter_sweep_vs_oracle(gpu_pkg, case):
    """Integrator / medium options one at a time: 3 waves (with the VSP-buffer updates after waves 1 and 2)
    on the workgroup kernel and on the per-lane kernel -- same film bit for bit, film == oracle up to the
    float-vs-double accumulation, 6 000 replayed paths bit-identical to the oracle."""
    P = gpu_pkg
    W, H = 64, 40
    kw = dict(PARAM_SWEEP[case])
    scene = P.fog_box_scene(W, H)
    if "_g" in kw:
        scene.medium.g = kw.pop("_g")
    if "_Le" in kw:
        scene.medium.Le[:] = kw.pop("_Le")
    if "_sigma" in kw:
        sa, ss = kw.pop("_sigma")
        scene.medium.sigma_a[:] = sa
        scene.medium.sigma_s[:] = ss
    prm = P.app_f_params()
    for k, v in kw.items():
        setattr(prm, k, v)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=case)
    films = []
    for kernel in ("wg", "lane"):
        os.environ["VSPG_KERNEL"] = kernel
        try:
            g = P.Renderer(scene, prm, W, H, seed=case)
            for w in range(3):
                g.render_wave(w, w + 1)
                g.post_process_wave()
            films.append(g.film())
            if kernel == "wg":
                rng = np.random.default_rng(case)
                pix = np.stack([rng.integers(0, W, 6000), rng.integers(0, H, 6000)], axis=1).astype(np.int32)
                si = rng.integers(0, 64, 6000).astype(np.int32)
                Lg, sg = g.trace_paths(pix, si)
            g.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
    assert np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32))
    for w in range(3):
        c.render_wave(w, w + 1)
        c.post_process_wave()
    fc = c.film()
    assert np.array_equal(films[0][..., 3], fc[..., 3])
    ig, ic = films[0][..., :3] / films[0][..., 3:4], fc[..., :3] / fc[..., 3:4]
    assert np.mean((ig - ic) ** 2 / (ic ** 2 + 1e-4)) <= 1e-10
    Lc, sc = c.trace_paths(pix, si)
    assert np.array_equal(sg, sc)
    assert np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    c.close()


def _tilted_scene(P, W, H):
    """fog box + a tilted (non-axis-aligned) diffuse blocker + a second, two-sided tilted light:
    exercises the generic rectangle code, shadow-ray occlusion and multi-light uniform sampling"""
    import math
    s = P.fog_box_scene(W, H)
    c, sn = math.cos(0.5), math.sin(0.5)
    q = s.quads[s.n_quads]
    q.p00[:] = (-0.5, -0.6, 0.2)
    q.e1[:] = (0.8 * c, 0.8 * sn, 0.0)
    q.e2[:] = (-0.6 * sn * 0.6, 0.6 * c * 0.6, 0.6 * 0.8)
    # make e2 exactly perpendicular to e1 (Gram-Schmidt in float64, then to float32)
    e1 = np.array(q.e1[:], dtype=np.float64)
    e2 = np.array(q.e2[:], dtype=np.float64)
    e2 = e2 - e1 * (e1 @ e2) / (e1 @ e1)
    q.e2[:] = tuple(float(np.float32(v)) for v in e2)
    q.Kd[:] = (0.6, 0.3, 0.2)
    s.n_quads += 1
    q = s.quads[s.n_quads]
    q.p00[:] = (0.4, -0.2, -0.3)
    q.e1[:] = (0.0, 0.3, 0.0)
    q.e2[:] = (0.3 * sn, 0.0, 0.3 * c)
    q.Le[:] = (3.0, 5.0, 9.0)
    q.Kd[:] = (0.2, 0.2, 0.2)
    q.two_sided = 1
    s.n_quads += 1
    s.medium.sigma_a[:] = (0.02, 0.05, 0.11)   # chromatic medium: hero-channel MIS matters
    s.medium.sigma_s[:] = (0.7, 0.45, 0.3)
    s.medium.g = 0.6
    return s


def test_tilted_chromatic_scene_vs_oracle(gpu_pkg):
    P = gpu_pkg
    W, H = 64, 48
    scene = _tilted_scene(P, W, H)
    prm = P.app_f_params()
    prm.lightsampler = P.LIGHTSAMPLER_UNIFORM
    g = P.Renderer(scene, prm, W, H, seed=7)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=7)
    rng = np.random.default_rng(11)
    n = 30000
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 100000, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    exact = np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1)
    ok = np.all(np.abs(Lg - Lc) <= 1e-4 * np.abs(Lc) + 1e-6, axis=1)
    print("tilted scene: same segments %.5f within tol %.5f bit-identical %.5f" % (np.mean(sg == sc), ok.mean(), exact.mean()))
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    for w in range(3):
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    fg, fc = g.film(), c.film()
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    assert np.mean((ig - ic) ** 2 / (ic ** 2 + 1e-4)) <= 1e-4
    g.close()
    c.close()


def test_multi_sample_launch_equals_single_sample_launches(gpu_pkg):
    # render_wave(0, 4) (one launch, 4 samples per pixel) == 4 launches of one sample
    P = gpu_pkg
    W, H = 40, 24
    scene = P.fog_box_scene(W, H)
    a = P.Renderer(scene, P.app_f_params(), W, H)
    b = P.Renderer(scene, P.app_f_params(), W, H)
    a.render_wave(0, 4)
    for w in range(4):
        b.render_wave(w, w + 1)
    assert np.array_equal(a.film().view(np.uint32), b.film().view(np.uint32))
    a.close()
    b.close()


@pytest.mark.parametrize("W,H", [(50, 37), (1, 1), (7, 130), (264, 9)])
def test_ragged_resolutions_film_equals_replayed_paths(gpu_pkg, W, H):
    """Resolutions that are not multiples of the 8x8 work tiles (padding items must neither render nor
    leak slots): every pixel of a one-sample film equals the replay of that pixel's path, both kernels."""
    P = gpu_pkg
    scene = P.fog_box_scene(W, H)
    for kernel in ("wg", "lane"):
        os.environ["VSPG_KERNEL"] = kernel
        try:
            r = P.Renderer(scene, P.app_f_params(), W, H)
            r.render_wave(0, 1)
            film = r.film()
            xy = np.stack(np.meshgrid(np.arange(W), np.arange(H)), -1).reshape(-1, 2).astype(np.int32)
            L, _ = r.trace_paths(xy, np.zeros(len(xy), dtype=np.int32))
            assert np.array_equal(film[..., 3], np.ones((H, W), dtype=np.float32))
            assert np.array_equal(film[..., :3].reshape(-1, 3).view(np.uint32), L.astype(np.float32).view(np.uint32)), kernel
            assert r.counters()["paths"] == W * H
            r.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)


def test_grey_specialisations_are_bit_identical(gpu_pkg):
    """The workgroup kernel has four instantiations for homogeneous media: generic, grey medium (medium spectra
    built from one value), grey scene (surface reflectances and the throughput too) and grey scene with an exactly
    zero null-collision coefficient (shadow rays end at their first tentative collision).  Same film, bit for bit --
    also for a scene where only the medium is grey (one wall coloured), which must pick the second one."""
    P = gpu_pkg
    W, H = 96, 64
    films = []
    for coloured_wall in (False, True):
        scene = P.fog_box_scene(W, H)
        if coloured_wall:
            scene.quads[1].Kd[0], scene.quads[1].Kd[1], scene.quads[1].Kd[2] = 0.63, 0.065, 0.05
        ref = None
        for env in ({"VSPG_NO_GREY": "1"}, {"VSPG_NO_GREY_KD": "1"}, {"VSPG_NO_NULLZERO": "1"}, {}):
            os.environ.update(env)
            try:
                r = P.Renderer(scene, P.app_f_params(), W, H, seed=5)
                for w in range(3):
                    r.render_wave(w, w + 1)
                    r.post_process_wave()
                f = r.film()
                r.close()
            finally:
                for k in env:
                    os.environ.pop(k, None)
            if ref is None:
                ref = f
            assert np.array_equal(ref.view(np.uint32), f.view(np.uint32)), (coloured_wall, env)
        films.append(ref)
    assert not np.array_equal(films[0], films[1])


@pytest.mark.parametrize("W,H", [(96, 64), (50, 37)])
def test_workgroup_schedulers_are_bit_identical(gpu_pkg, W, H):
    """The three schedulers of the workgroup kernel -- k_render_wave_wg (global work head, film flush between the phases),
    k_render_wave_wg2 (tiles from a global head, parked samples, two barriers) and k_render_wave_wg3 (round 5, the default: ring
    queues in LDS, no workgroup barrier, a pool larger than the workgroup) -- over
    the four homogeneous instantiations, with one-sample launches, a multi-sample launch (restarts, film atomics) and the
    image-space buffer updating in between: same films, same VSP buffers, same counters."""
    P = gpu_pkg
    scene = P.fog_box_scene(W, H)
    for env in ({"VSPG_NO_GREY": "1"}, {"VSPG_NO_GREY_KD": "1"}, {"VSPG_NO_NULLZERO": "1"}, {}):
        out = []
        for sched in ("1", "2", "2 mixed", "3"):   # "mixed": 3/8 of the frame from the global tile head, the rest dealt out up front; "3": k_render_wave_wg3 (the default)
            os.environ.update(env)
            os.environ["VSPG_WG_SCHED"] = sched[0]
            if sched.endswith("mixed"):
                os.environ["VSPG_WG2_TAIL"] = "24"
            try:
                r = P.Renderer(scene, P.app_f_params(), W, H, seed=3)
                for w in range(3):
                    r.render_wave(w, w + 1)
                    r.post_process_wave()
                r.render_wave(3, 6)
                for _ in range(3):
                    r.post_process_wave()
                r.render_wave(6, 7)
                out.append((r.film(), r.vsp_buffer()[0], r.counters()))
                r.close()
            finally:
                os.environ.pop("VSPG_WG_SCHED", None)
                os.environ.pop("VSPG_WG2_TAIL", None)
                for k in env:
                    os.environ.pop(k, None)
        for other in out[1:]:
            assert np.array_equal(out[0][0].view(np.uint32), other[0].view(np.uint32)), env
            assert np.array_equal(out[0][1].view(np.uint32), other[1].view(np.uint32)), env
            assert out[0][2] == other[2] and out[0][2]["paths"] == 7 * W * H, (env, out[0][2], other[2])


@pytest.mark.parametrize("guided", [False, True])
def test_parked_samples_reach_the_film_whoever_asks_first(gpu_pkg, guided):
    """A one-sample launch of k_render_wave_wg2 parks its samples; the NEXT such launch resolves them as it starts each pixel, and
    whatever else touches the film or the image-space statistics first -- a film read, the buffer update of PostProcessWave, a
    multi-sample launch, a film clear, the device-pointer getters of the sharded step -- makes them enter before it looks.  The
    same call sequence with VSPG_WG2_DEFER=0 (every launch resolves its own samples at once) gives the same bits at every step."""
    P = gpu_pkg
    W, H = 100, 70
    scene = P.fog_box_scene(W, H)
    prm = P.default_params() if guided else P.app_f_params()
    field = None
    if guided:
        from scenes import light_field
        field = light_field(P, n=4)
    logs = []
    for defer in ("1", "0"):
        os.environ["VSPG_WG2_DEFER"] = defer
        try:
            r = P.Renderer(scene, prm, W, H, seed=9)
            if field is not None:
                r.set_guiding_field(field, field)
            assert r.kernel_name().startswith("k_render_wave_wg2<" if guided else "k_render_wave_wg3<")
            log = []
            r.render_wave(0, 1)
            log.append(r.film())                       # a read right after a launch
            r.post_process_wave()                      # wave counter 1: the buffer updates from the statistics (parked ones included)
            log.append(r.vsp_buffer()[0])
            r.render_wave(1, 2); r.post_process_wave() # update at 2
            r.render_wave(2, 3); r.post_process_wave() # no update: stays parked
            r.render_wave(3, 4); r.post_process_wave() # update at 4, two launches' samples behind it
            log.append(r.vsp_buffer()[0])
            r.render_wave(4, 7)                        # multi-sample launch: film atomics, after the parked samples
            log.append(r.film())
            r.render_wave(7, 8)
            r.film_clear()                             # the parked samples are not resurrected by the next launch
            r.render_wave(8, 9)
            r.render_wave(9, 10)
            log.append(r.film())
            log.append(np.array(sorted(r.counters().items()), dtype=object))
            logs.append(log)
            r.close()
        finally:
            os.environ.pop("VSPG_WG2_DEFER", None)
    for a, b in zip(*logs):
        if a.dtype == object:
            assert (a == b).all()
        else:
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    f = logs[0][4]
    assert np.all(f[..., 3] == 2.0)    # two samples per pixel since the clear


def test_grey_grid_medium_film_equals_replayed_paths(gpu_pkg):
    """A grid medium with grey sigma_a / sigma_s renders through the broadcast-spectrum instantiation of the
    per-lane kernel; the path replay (k_trace_paths) uses the generic one: same pixels bit for bit, and the
    same again with the specialisation switched off."""
    from scenes import cloud_density, grid_scene
    P = gpu_pkg
    W, H = 48, 40
    scene = grid_scene(cloud_density(16), (16, 16, 16), 0.08, 2.6, g=0.5, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    xy = np.stack(np.meshgrid(np.arange(W), np.arange(H)), -1).reshape(-1, 2).astype(np.int32)
    films = []
    for env in ({}, {"VSPG_NO_GREY": "1"}):
        os.environ.update(env)
        try:
            r = P.Renderer(scene, P.app_f_params(), W, H, seed=2)
            r.render_wave(0, 1)
            film = r.film()
            L, _ = r.trace_paths(xy, np.zeros(len(xy), dtype=np.int32))
            r.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
        assert np.array_equal(film[..., :3].reshape(-1, 3).view(np.uint32), L.astype(np.float32).view(np.uint32))
        films.append(film)
    assert np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32))


@pytest.mark.parametrize("W,H", [(1920, 1080), (3840, 2160)])
@pytest.mark.parametrize("kind", ["cloud", "cloud-nvdb"])
def test_full_size_cloud_wave_properties(gpu_pkg, kind, W, H):
    """Configs 3-4 at their full film sizes over the 256^3 heterogeneous stand-in (GridMedium and NanoVDBMedium
    semantics, SampleT_maj_Resampling + reservoir selection): one wave -- every pixel one sample, counters consistent,
    every kernel the library offers for the medium gives the same film bit for bit, the film equals the device's own
    replayed paths AND the oracle's paths on sampled pixels."""
    P = gpu_pkg
    scene = P.cloud_box_scene(W, H, 256) if kind == "cloud" else P.nanovdb_box_scene(W, H, 256)
    prm = P.app_f_params()
    films = {}
    for kernel in ("default", "lane", "wg"):
        if kernel == "default":
            os.environ.pop("VSPG_KERNEL", None)
        else:
            os.environ["VSPG_KERNEL"] = kernel
        try:
            r = P.Renderer(scene, prm, W, H)
            name = r.kernel_name()
            r.render_wave(0, 1)
            film = r.film()
            cnt = r.counters()
            r_trace = r
            assert cnt["paths"] == W * H
            assert W * H <= cnt["segments"] <= 6 * W * H
            assert cnt["density_queries"] > cnt["segments"]          # a heterogeneous walk visits many tentative collisions
            assert cnt["volume_scatters"] + cnt["surface_hits"] <= cnt["segments"]
            assert np.array_equal(film[..., 3], np.ones((H, W), dtype=np.float32))
            assert np.isfinite(film).all()
            if not films:
                rng = np.random.default_rng(9)
                n = 4000
                xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], -1).astype(np.int32)
                L, sg = r.trace_paths(xy, np.zeros(n, dtype=np.int32))
                got = film[xy[:, 1], xy[:, 0], :3]
                assert np.array_equal(got.view(np.uint32), L.astype(np.float32).view(np.uint32))
                c = oracle_lib.OracleRenderer(scene, prm, W, H)
                Lc, sc = c.trace_paths(xy, np.zeros(n, dtype=np.int32))
                c.close()
                assert np.array_equal(sg, sc)
                assert np.array_equal(got.view(np.uint32), Lc.astype(np.float32).view(np.uint32)), "film != oracle paths"
            films[name] = film
            r.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
    ref = next(iter(films.values()))
    print(kind, W, H, "kernels compared:", sorted(films))
    for name, f in films.items():
        assert np.array_equal(ref.view(np.uint32), f.view(np.uint32)), name


@pytest.mark.parametrize("W,H", [(1920, 1080), (3840, 2160)])
def test_full_size_wave_properties(gpu_pkg, W, H):
    """BASELINE sizes (1920x1080 of configs 2-3, 3840x2160 of config 4): size-independent properties of one wave -- every
    pixel got exactly one sample, path / segment counters are consistent, 20 000 random pixels equal their replayed
    paths bit for bit, a second identical renderer reproduces the film exactly (run-to-run determinism), and the
    per-lane and workgroup kernels agree bit for bit."""
    P = gpu_pkg
    scene = P.fog_box_scene(W, H)
    films = {}
    for kernel in ("wg", "lane", "wg"):
        os.environ["VSPG_KERNEL"] = kernel
        try:
            r = P.Renderer(scene, P.app_f_params(), W, H)
            r.render_wave(0, 1)
            film = r.film()
            cnt = r.counters()
            assert cnt["paths"] == W * H
            assert W * H <= cnt["segments"] <= 6 * W * H  # maxdepth 5
            assert np.array_equal(film[..., 3], np.ones((H, W), dtype=np.float32))
            if kernel in films:
                assert np.array_equal(films[kernel].view(np.uint32), film.view(np.uint32)), "not deterministic"
            else:
                rng = np.random.default_rng(5)
                xy = np.stack([rng.integers(0, W, 20000), rng.integers(0, H, 20000)], -1).astype(np.int32)
                L, _ = r.trace_paths(xy, np.zeros(len(xy), dtype=np.int32))
                got = film[xy[:, 1], xy[:, 0], :3]
                assert np.array_equal(got.view(np.uint32), L.astype(np.float32).view(np.uint32)), kernel
                # ... and the ORACLE's paths for the same pixels, at the full film size (camera rays of this resolution)
                c = oracle_lib.OracleRenderer(scene, P.app_f_params(), W, H)
                Lc, _ = c.trace_paths(xy, np.zeros(len(xy), dtype=np.int32))
                c.close()
                assert np.array_equal(got.view(np.uint32), Lc.astype(np.float32).view(np.uint32)), "film != oracle paths (%s)" % kernel
            films[kernel] = film
            r.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
    assert np.array_equal(films["wg"].view(np.uint32), films["lane"].view(np.uint32))


def test_full_size_guided_wave_properties(gpu_pkg):
    """The configuration a `guidedvolpathvspg` user gets by default (guidedvolpathvspgintegrator.cpp:1263-1319: surface RIS +
    volume MIS guiding, primary + secondary VSP) at 1920x1080: the field trains in-loop for four waves; with that field and
    that image-space buffer in place, one wave on the workgroup kernel (the default) and on the per-lane kernel -- every pixel
    one sample, counters consistent, the two films bit-identical, 20 000 random pixels equal to their replayed paths on the
    device AND to the oracle's paths through the same field."""
    P = gpu_pkg
    W, H = 1920, 1080
    scene = P.fog_box_scene(W, H)
    prm = P.default_params()
    prm.guide_num_training_waves = 4
    t = P.Renderer(scene, prm, W, H)
    assert t.kernel_name() == "k_render_wave_wg2<HomogeneousMediumT<2,true>,guided,train>"
    for w in range(4):
        t.render_wave(w, w + 1)
        t.post_process_wave()
    st = t.training_stats()
    assert st["training"] == 0 and st["iteration"] == 4, st
    fields = []
    for vol in (0, 1):
        nodes, regs, nn, nr = t.get_guiding_field(vol)
        assert nn >= 15 and nr >= 8, (vol, nn, nr)   # four updates: every leaf that saw enough samples split once per update
        fields.append(P.Field(list(nodes)[:nn], list(regs)[:nr]))
    vsp, ready = t.vsp_buffer()
    assert ready
    t.close()
    rng = np.random.default_rng(9)
    xy = np.stack([rng.integers(0, W, 20000), rng.integers(0, H, 20000)], -1).astype(np.int32)
    si = np.full(len(xy), 4, dtype=np.int32)
    films = {}
    for kernel in (None, "lane"):
        if kernel:
            os.environ["VSPG_KERNEL"] = kernel
        try:
            r = P.Renderer(scene, prm, W, H)
            r.set_guiding_field(fields[0], fields[1])
            r.load_vsp_buffer(vsp)
            r.render_wave(4, 5)
            film = r.film()
            cnt = r.counters()
            assert cnt["paths"] == W * H and W * H <= cnt["segments"] <= 6 * W * H
            assert np.array_equal(film[..., 3], np.ones((H, W), dtype=np.float32))
            got = film[xy[:, 1], xy[:, 0], :3]
            L, _ = r.trace_paths(xy, si)
            assert np.array_equal(got.view(np.uint32), L.astype(np.float32).view(np.uint32)), r.kernel_name()
            films[r.kernel_name()] = film
            r.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
    assert sorted(films) == ["k_render_wave<HomogeneousMediumT<2,true>,guided>", "k_render_wave_wg2<HomogeneousMediumT<2,true>,guided>"]
    a, b = films.values()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    c = oracle_lib.OracleRenderer(scene, prm, W, H)
    c.set_guiding_field(fields[0], fields[1])
    c.load_vsp_buffer(vsp)
    Lc, _ = c.trace_paths(xy, si)
    c.close()
    assert np.array_equal(a[xy[:, 1], xy[:, 0], :3].view(np.uint32), Lc.astype(np.float32).view(np.uint32)), "film != oracle paths"


@pytest.mark.parametrize("shape", ["box", "scene"])
def test_full_size_guided_cloud_wave_properties(gpu_pkg, shape):
    """Config 5's shape at 1920x1080: a 256^3 density grid with NanoVDBMedium semantics (64^3 majorants) AND a temperature grid
    (accepted under "resampling": the path never evaluates volume emission there), the reference's DEFAULT
    options (surface RIS + volume MIS guiding, primary + secondary VSP), the field trained in the loop for four waves.  `box`: the
    medium fills the fog box (rounds 1-3); `scene`: the reference's scene shape -- camera in vacuum, the medium behind an
    interface-material sphere, ground, sun + sky (round 4).  With that field and that image-space buffer in place: one wave on
    the wavefront pipeline (default) and on the per-lane kernel -- every pixel one sample, counters consistent, the two films
    bit-identical, 20 000 random pixels equal to their replayed paths on the device AND to the oracle's paths."""
    import time
    P = gpu_pkg
    W, H = 1920, 1080
    t0 = time.time()
    scene = P.nanovdb_box_scene(W, H, 256) if shape == "box" else P.cloud_scene(W, H, 256, nvdb=True)
    temp = (300.0 + 1500.0 * np.clip(P.procedural_cloud_density(256, seed=11), 0, 1)).astype(np.float32)
    scene.medium.temperature = temp.ctypes.data_as(C.POINTER(C.c_float))
    scene.medium.nvdb_le_scale, scene.medium.temperature_offset, scene.medium.temperature_scale = 2.0, 0.0, 1.0
    prm = P.default_params()
    prm.guide_num_training_waves = 4
    t = P.Renderer(scene, prm, W, H)
    assert t.kernel_name() == "k_wf_walk<NanoDenseMedium,guided,train>"   # (guided pipelines: both walks in one kernel)
    for w in range(4):
        t.render_wave(w, w + 1)
        t.post_process_wave()
    st = t.training_stats()
    assert st["training"] == 0 and st["iteration"] == 4, st
    fields = []
    for vol in (0, 1):
        nodes, regs, nn, nr = t.get_guiding_field(vol)
        assert nn >= 3 and nr >= 2, (vol, nn, nr)
        fields.append(P.Field(list(nodes)[:nn], list(regs)[:nr]))
    vsp, ready = t.vsp_buffer()
    assert ready
    t.close()
    rng = np.random.default_rng(9)
    xy = np.stack([rng.integers(0, W, 20000), rng.integers(0, H, 20000)], -1).astype(np.int32)
    si = np.full(len(xy), 4, dtype=np.int32)
    films = {}
    for kernel in (None, "lane"):
        if kernel:
            os.environ["VSPG_KERNEL"] = kernel
        try:
            r = P.Renderer(scene, prm, W, H)
            r.set_guiding_field(fields[0], fields[1])
            r.load_vsp_buffer(vsp)
            r.render_wave(4, 5)
            film = r.film()
            cnt = r.counters()
            assert cnt["paths"] == W * H and W * H <= cnt["segments"] <= 30 * W * H
            assert cnt["density_queries"] > 0 and cnt["shadow_density_queries"] > 0
            assert np.array_equal(film[..., 3], np.ones((H, W), dtype=np.float32)) and np.isfinite(film).all()
            got = film[xy[:, 1], xy[:, 0], :3]
            L, _ = r.trace_paths(xy, si)
            assert np.array_equal(got.view(np.uint32), L.astype(np.float32).view(np.uint32)), r.kernel_name()
            films[r.kernel_name()] = (film, cnt)
            r.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
    assert sorted(films) == ["k_render_wave<NanoDenseMedium,guided>", "k_wf_walk<NanoDenseMedium,guided>"], sorted(films)
    (a, ca), (b, cb) = films.values()
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and ca == cb
    c = oracle_lib.OracleRenderer(scene, prm, W, H)
    c.set_guiding_field(fields[0], fields[1])
    c.load_vsp_buffer(vsp)
    Lc, _ = c.trace_paths(xy, si)
    c.close()
    assert np.array_equal(a[xy[:, 1], xy[:, 0], :3].view(np.uint32), Lc.astype(np.float32).view(np.uint32)), "film != oracle paths"
    print("config-5 shape (%s): %.1f s" % (shape, time.time() - t0))
    assert time.time() - t0 < 90


def test_render_waves_vs_oracle(pair):
    P, g, c = pair
    for w in range(6):  # waves 1,2,4 trigger image-space VSP updates
        g.render_wave(w, w + 1)
        g.post_process_wave()
        c.render_wave(w, w + 1)
        c.post_process_wave()
    fg, fc = g.film(), c.film()
    assert np.array_equal(fg[..., 3], fc[..., 3])
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    relmse = np.mean((ig - ic) ** 2 / (ic ** 2 + 1e-4))
    close = np.mean(np.all(np.abs(ig - ic) <= 1e-4 * (1 + np.abs(ic)), axis=-1))
    print("film relMSE %.3e, pixels within tol %.5f" % (relmse, close))
    assert relmse <= 1e-4          # BASELINE.json: relMSE <= 1e-4 vs CPU at equal spp
    assert close == 1.0
    vg, rg = g.vsp_buffer()
    vc, rc = c.vsp_buffer()
    assert rg and rc
    assert np.mean(np.abs(vg - vc) <= 1e-6) == 1.0  # float statistics, same samples in the same order
    cg, cc = g.counters(), c.counters()
    assert cg["paths"] == cc["paths"] == 6 * g.xres * g.yres
    for k in cg:
        assert abs(cg[k] - cc[k]) <= 2e-3 * cc[k] + 5, (k, cg[k], cc[k])


def test_sharded_waves_sum_to_unsharded(gpu_pkg):
    # multi-GPU sharding contract: shard i renders waves w % n == i; the sum of the
    # shard films equals the single-renderer film (VSP buffer frozen at its initial 0.5)
    P = gpu_pkg
    W, H = 40, 24
    scene = P.fog_box_scene(W, H)
    prm = P.app_f_params()
    full = P.Renderer(scene, prm, W, H)
    full.render_wave(0, 4)
    ref = full.film()
    acc = np.zeros_like(ref)
    for i in range(2):
        r = P.Renderer(scene, prm, W, H, shard_index=i, shard_count=2)
        r.render_wave(0, 4)
        acc += r.film()
        r.close()
    assert np.array_equal(acc[..., 3], ref[..., 3])
    assert np.allclose(acc, ref, rtol=1e-6, atol=1e-7)
    full.close()


def test_sharded_steps_with_buffer_updates_vs_oracle_shards(gpu_pkg):
    """The multi-GPU step bench.py runs: two shards of one frame, the image-space VSP statistics summed
    over the shards on the steps where the buffer updates (vspg_isg_update_due / vspg_post_process_step) -- against two
    ORACLE shards stepped the same way (same samples, same float sums: bit-identical buffers and paths), and against one
    renderer that renders two sample indices per step (same estimator, equal up to the summation order of the statistics)."""
    import torch
    P = gpu_pkg
    W, H, steps = 96, 64, 5
    scene = P.fog_box_scene(W, H)
    prm = P.app_f_params()
    g = [P.Renderer(scene, prm, W, H, shard_index=i, shard_count=2) for i in range(2)]
    c = [oracle_lib.OracleRenderer(scene, prm, W, H, shard_index=i, shard_count=2) for i in range(2)]
    one = P.Renderer(scene, prm, W, H)

    def dev_stats(r):
        ptr, n = r.isg_stats_ptr()

        class Dev:
            __cuda_array_interface__ = {"shape": (n,), "typestr": "<f4", "data": (ptr, False), "version": 2}
        return torch.as_tensor(Dev(), device="cuda:0")

    updates = 0
    for step in range(steps):
        for r in g + c:
            r.render_wave(2 * step, 2 * step + 2)
        one.render_wave(2 * step, 2 * step + 2)
        due = g[0].isg_update_due(2)
        assert due == g[1].isg_update_due(2) == c[0].isg_update_due(2) == one.isg_update_due(2)
        updates += due
        tg = tc = None
        if due:
            torch.cuda.synchronize()
            tg = dev_stats(g[0]) + dev_stats(g[1])        # what the RCCL all-reduce hands every rank
            tc = c[0].isg_stats().reshape(-1) + c[1].isg_stats().reshape(-1)
            assert np.array_equal(tg.cpu().numpy().view(np.uint32), tc.view(np.uint32)), "shard statistics differ from the oracle's"
        for r in g:
            r.post_process_step(2, tg.data_ptr() if due else None)
        for r in c:
            r.post_process_step(2, tc if due else None)
        one.post_process_step(2)
        torch.cuda.synchronize()
    assert updates == 3
    v0, ready0 = g[0].vsp_buffer()
    v1, _ = g[1].vsp_buffer()
    vc, readyc = c[0].vsp_buffer()
    assert ready0 and readyc
    assert np.array_equal(v0, v1), "the two shards hold different VSP buffers"
    assert np.mean(np.abs(v0 - vc) <= 1e-6) == 1.0
    fg = g[0].film() + g[1].film()
    fc = c[0].film() + c[1].film()
    assert np.array_equal(fg[..., 3], fc[..., 3]) and np.all(fg[..., 3] == 2 * steps)
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    assert np.mean(np.all(np.abs(ig - ic) <= 1e-4 * (1 + np.abs(ic)), axis=-1)) == 1.0
    # one renderer, two sample indices per step: same estimator (see tests/test_sharding_gloo.py for why not the same bits)
    vo, _ = one.vsp_buffer()
    dv = np.abs(v0 - vo)
    print("sharded vs single renderer: VSP bit-identical %.3f, mean |diff| %.2e, max %.2e" % ((dv == 0).mean(), dv.mean(), dv.max()))
    assert dv.mean() < 2e-3 and dv.max() < 0.05
    fo = one.film()
    io = fo[..., :3] / fo[..., 3:4]
    assert abs(ig.mean() / io.mean() - 1) < 0.02
    for r in g + c + [one]:
        r.close()


class _ThreadDist:
    """An in-process stand-in for torch.distributed: `world` threads, one per rank, meet in all_reduce and every one of them
    leaves with the same sum (rank order, so the same bits).  Enough to drive vspg-pbrt-v4_amd/sharding.py -- the code
    bench.py --gpus N runs -- with two HIP renderers on ONE card (RCCL refuses two ranks on a device)."""

    class ReduceOp:
        SUM = "sum"

    def __init__(self, world, torch):
        import threading
        self.world, self.torch = world, torch
        self.barrier = threading.Barrier(world)
        self.slots = [None] * world
        self.tls = threading.local()
        self.calls = 0

    def all_reduce(self, t, op=None):
        torch = self.torch
        torch.cuda.synchronize()
        self.slots[self.tls.rank] = t
        self.barrier.wait()
        total = self.slots[0].clone()
        for k in range(1, self.world):
            total += self.slots[k]
        torch.cuda.synchronize()
        self.barrier.wait()
        t.copy_(total)
        torch.cuda.synchronize()
        if self.tls.rank == 0:
            self.calls += 1
        self.barrier.wait()


def _run_shard_sync(P, scene, prm, W, H, steps, break_flush=False):
    """Two HIP shards stepped by ShardSync itself (threads as ranks, own stream each); returns per rank the VSP buffer, the
    summed statistics ShardSync handed the update at every due step, and the film after frame_end_allreduce."""
    import importlib.util
    import threading
    import torch
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("vspg_sharding", os.path.join(ROOT, "vspg-pbrt-v4_amd", "sharding.py"))
    sh = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sh)
    world = 2
    fake = _ThreadDist(world, torch)
    dev = torch.device("cuda", 0)
    out, errors = [None] * world, []

    def rank_main(rank):
        try:
            fake.tls.rank = rank
            torch.cuda.set_device(0)
            r = P.Renderer(scene, prm, W, H, shard_index=rank, shard_count=world)
            if break_flush:            # round 3's host: the pointer wrapped once, nobody resolves the parked wave
                r.flush = lambda stream=None: None
            ts = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(ts):
                stream = ts.cuda_stream
                fptr, fn = r.film_ptr()   # taken ONCE, before the first wave, as bench.py does

                class Dev:
                    __cuda_array_interface__ = {"shape": (fn,), "typestr": "<f4", "data": (fptr, False), "version": 2}
                film = torch.as_tensor(Dev(), device=dev)
                sync = sh.ShardSync(fake, r, world, torch, device=dev)
                sums = []
                for step in range(steps):
                    w0, w1 = sh.step_wave_range(step, world)
                    r.render_wave(w0, w1, stream)
                    due = r.isg_update_due(world)
                    sync.post_process_step(stream)
                    if due:
                        ts.synchronize()
                        sums.append(sync._sum.cpu().numpy().copy())
                sh.frame_end_allreduce(fake, film, world, r, stream)
                ts.synchronize()
                summed = film.cpu().numpy().reshape(H, W, 4).copy()
            vsp, ready = r.vsp_buffer()
            out[rank] = (vsp, ready, sums, summed)
            r.close()
        except Exception as e:  # noqa: BLE001
            errors.append(e)
            fake.barrier.abort()
    th = [threading.Thread(target=rank_main, args=(k,)) for k in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    if errors:
        raise errors[0]
    return out


def test_shard_sync_on_the_hip_renderer_vs_oracle_shards(gpu_pkg):
    """sharding.py (ShardSync.post_process_step + frame_end_allreduce: what `bench.py --gpus N` executes) driving TWO HIP
    renderers on one card, against two oracle shards stepped the same way: the statistics every update ran on are
    bit-identical to the oracle's sums, both ranks hold one VSP buffer, and every pixel of the all-reduced film carries
    2 * steps samples -- each rank's LAST wave included (a one-sample wave parks its samples; vspg_flush).  The same
    run with the flush taken out (round 3's host) must fail these checks: the test sees what it is there to see."""
    P = gpu_pkg
    W, H, steps = 96, 64, 6
    scene = P.fog_box_scene(W, H)
    prm = P.app_f_params()
    c = [oracle_lib.OracleRenderer(scene, prm, W, H, shard_index=i, shard_count=2) for i in range(2)]
    csums = []
    for step in range(steps):
        for r in c:
            r.render_wave(2 * step, 2 * step + 2)
        due = c[0].isg_update_due(2)
        tc = None
        if due:
            tc = c[0].isg_stats().reshape(-1) + c[1].isg_stats().reshape(-1)
            csums.append(tc.copy())
        for r in c:
            r.post_process_step(2, tc)
    assert len(csums) == 3                    # the wave counter passes 2, 4, 8
    vc, readyc = c[0].vsp_buffer()
    fc = c[0].film() + c[1].film()
    got = _run_shard_sync(P, scene, prm, W, H, steps)
    for rank in range(2):
        vsp, ready, sums, film = got[rank]
        assert ready and readyc and len(sums) == 3
        for k in range(3):
            assert np.array_equal(sums[k].view(np.uint32), csums[k].view(np.uint32)), "update %d ran on other statistics than the oracle shards'" % k
        assert np.mean(np.abs(vsp - vc) <= 1e-6) == 1.0
        assert np.all(film[..., 3] == 2 * steps), "the all-reduced film lacks samples"
        assert np.array_equal(film[..., 3], fc[..., 3])
        ig, ic = film[..., :3] / film[..., 3:4], fc[..., :3] / fc[..., 3:4]
        assert np.mean(np.all(np.abs(ig - ic) <= 1e-4 * (1 + np.abs(ic)), axis=-1)) == 1.0
    assert np.array_equal(got[0][0], got[1][0]), "the two ranks hold different VSP buffers"
    assert np.array_equal(got[0][3], got[1][3]), "the two ranks hold different films after the all-reduce"
    print("ShardSync on HIP shards: VSP buffer bit-identical to the oracle shards' in %.4f of the pixels" % np.mean(got[0][0] == vc))
    # ... and without the flush: stale statistics from the second update on, a film that lacks each rank's last wave
    bad = _run_shard_sync(P, scene, prm, W, H, steps, break_flush=True)
    stale = [not np.array_equal(bad[0][2][k].view(np.uint32), csums[k].view(np.uint32)) for k in range(3)]
    assert stale[1] and stale[2], "the test would not have caught round 3's stale statistics"
    assert not np.all(bad[0][3][..., 3] == 2 * steps), "the test would not have caught the film that lacks the last wave"
    for r in c:
        r.close()


# ---------------------------------------------------------------------------------------------
# heterogeneous medium (GridMedium: DDA majorants, trilinear density, SampleT_maj_Resampling)
# ---------------------------------------------------------------------------------------------
def test_grid_known_answers_on_device(gpu_pkg):
    from scenes import d3_density, grid_scene
    P = gpu_pkg
    dens = d3_density()
    scene = grid_scene(dens, (8, 8, 8), 0.5, 4.5)
    g = P.Renderer(scene, P.app_f_params(), 16, 16)
    q = P.VspgTmajQuery(P.f3(0.1, 0.2, -0.5), P.f3(0.3, 0.2, 1.0), 2.0, 0.37, 0.25, 0.75, 0.6, 1, 0)
    o = g.sample_tmaj_batch(P.TMAJ_RESAMPLING, [q])[0]
    # SURVEY.md App. D.3 (reference's own output)
    assert o.n_callbacks == 8 and o.sum_sigt_over_maj == fh("0x1.dd496p+1")
    assert o.T_maj[1] == fh("0x1.8f239ep-2") and o.vrc == fh("0x1.356952p-1") and o.majorant_scale == 1.0
    q3 = P.VspgTmajQuery(P.f3(0.1, 0.2, -0.5), P.f3(0.3, 0.2, 1.0), 2.0, 0.37, 0.25, 0.75, -1.0, 1, 3)
    o = g.sample_tmaj_batch(P.TMAJ_PLAIN, [q3])[0]
    assert o.n_callbacks == 3 and o.last_p[2] == fh("0x1.717118p-2") and list(o.T_maj) == [1.0, 1.0, 1.0]
    g.close()


@pytest.fixture(scope="module")
def cloud_pair(gpu_pkg):
    from scenes import cloud_density, grid_scene
    P = gpu_pkg
    W, H = 64, 48
    dens = cloud_density(24)
    scene = grid_scene(dens, (24, 24, 24), (0.05, 0.08, 0.1), (3.0, 2.6, 2.2), g=0.5, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    prm = P.app_f_params()
    g = P.Renderer(scene, prm, W, H, seed=3)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=3)
    yield P, g, c
    g.close()
    c.close()


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_grid_free_flight_vs_oracle(cloud_pair, variant):
    P, g, c = cloud_pair
    rng = np.random.default_rng(40 + variant)
    qs = []
    for i in range(20000):
        d = rng.normal(size=3)
        d = d / np.linalg.norm(d) * rng.uniform(0.5, 2.0)
        stop = int(rng.integers(0, 4))
        qs.append(P.VspgTmajQuery(P.f3(*rng.uniform(-1, 1, 3)), P.f3(*d), float(rng.uniform(0.0, 3.0)), float(rng.random()),
                                  float(rng.random()), float(rng.random()), float(rng.random()) if i % 5 else -1.0,
                                  int(rng.integers(0, 3)), stop))
    go, co = g.sample_tmaj_batch(variant, qs), c.sample_tmaj_batch(variant, qs)
    exact = 0
    for a, b in zip(go, co):
        assert a.n_callbacks == b.n_callbacks
        same = (list(a.T_maj) == list(b.T_maj) and list(a.r_u_factor) == list(b.r_u_factor) and a.last_t == b.last_t
                and a.sum_sigt_over_maj == b.sum_sigt_over_maj and a.vrc == b.vrc and a.majorant_scale == b.majorant_scale)
        exact += same
        assert np.allclose(list(a.T_maj), list(b.T_maj), rtol=1e-5, atol=1e-30)
    print("grid variant", variant, "bit-identical fraction", exact / len(qs))
    assert exact == len(qs)


def test_grid_paths_and_film_vs_oracle(cloud_pair):
    P, g, c = cloud_pair
    rng = np.random.default_rng(9)
    n = 20000
    pix = np.stack([rng.integers(0, g.xres, n), rng.integers(0, g.yres, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    exact = np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1)
    ok = np.all(np.abs(Lg - Lc) <= 1e-4 * np.abs(Lc) + 1e-6, axis=1)
    print("cloud paths: same segments %.5f within tol %.5f bit-identical %.5f" % (np.mean(sg == sc), ok.mean(), exact.mean()))
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    for w in range(5):
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    fg, fc = g.film(), c.film()
    assert np.array_equal(fg[..., 3], fc[..., 3])
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    relmse = np.mean((ig - ic) ** 2 / (ic ** 2 + 1e-4))
    print("cloud film relMSE %.3e" % relmse)
    assert relmse <= 1e-4
    cg, cc = g.counters(), c.counters()
    assert cg["density_queries"] > cg["volume_scatters"]
    for k in cg:
        assert abs(cg[k] - cc[k]) <= 2e-3 * cc[k] + 5, (k, cg[k], cc[k])


def test_loaded_vsp_buffer_vs_oracle(gpu_pkg):
    """loadISGBuffer (:151-159, :251-256): a buffer handed over is used from wave 0 on and never updated -- device == oracle."""
    P = gpu_pkg
    W, H = 96, 64
    scene = P.fog_box_scene(W, H)
    prm = P.app_f_params()
    rng = np.random.default_rng(4)
    vsp = rng.uniform(0.05, 0.95, (H, W)).astype(np.float32)
    vsp[::7, ::5] = -1.0  # pixels without an estimate
    g = P.Renderer(scene, prm, W, H, seed=2)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=2)
    g.load_vsp_buffer(vsp); c.load_vsp_buffer(vsp)
    for w in range(3):
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    vg, rg = g.vsp_buffer()
    assert rg and np.array_equal(vg, vsp)
    fg, fc = g.film(), c.film()
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    relmse = np.mean((ig - ic) ** 2 / (ic ** 2 + 1e-4))
    print("loaded VSP buffer film relMSE %.3e" % relmse)
    assert relmse <= 1e-10
    g.close(); c.close()


# ---------------------------------------------------------------------------------------------
# TrBuffer + NDS+ (cpu/trbuffer.h; guidedvolpathvspgintegrator.cpp:727-728, 929-938, 975-976, 1072-1073)
# ---------------------------------------------------------------------------------------------
def test_device_powf_matches_host_libm(pair, libm_shim):
    """std::pow(float, float) of the NDS+ bias: device == host libm bit for bit, special cases included."""
    P, g, c = pair
    rng = np.random.default_rng(13)
    n = 1_000_000
    sp = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 0.5, -0.5, 2.0, -2.0, 3.0, -3.0, 1e-45, -1e-45,
                   3.4e38, 2.0 ** -126, float.fromhex("0x1.fffffep-1")], dtype=np.float32)
    X, Y = np.meshgrid(sp, sp)
    x = np.concatenate([rng.uniform(0, 1, n), rng.uniform(0, 1.0001, n), np.exp(rng.uniform(-88, 88, n)), np.exp(rng.uniform(-104, -80, n)),
                        np.exp(rng.uniform(-5, 5, n)), -np.exp(rng.uniform(-5, 5, n)), X.ravel()]).astype(np.float32)
    y = np.concatenate([1 / (1 + rng.uniform(0, 1, n)), rng.uniform(0.4, 1.1, n), rng.uniform(-3, 3, n), rng.uniform(-1.5, 1.5, n),
                        rng.uniform(-40, 40, n), rng.integers(-40, 40, n), Y.ravel()]).astype(np.float32)
    dev = g.libm_powf_batch(x, y)
    ref = np.empty_like(x)
    fp = C.POINTER(C.c_float)
    libm_shim.libm_powf(x.shape[0], x.ctypes.data_as(fp), y.ctypes.data_as(fp), ref.ctypes.data_as(fp))
    bad = np.nonzero((dev.view(np.uint32) != ref.view(np.uint32)) & ~(np.isnan(dev) & np.isnan(ref)))[0]
    assert bad.size == 0, (bad.size, x[bad[:3]], y[bad[:3]], dev[bad[:3]], ref[bad[:3]])


@pytest.mark.parametrize("medium", ["grid", "nvdb"])
def test_tr_buffer_and_nds_plus_vs_oracle(gpu_pkg, medium):
    from scenes import cloud_density, grid_scene, nvdb_scene
    P = gpu_pkg
    # a THIN cloud: NDS (and with it NDS+) only acts where the wanted scatter probability exceeds 1 - exp(-tau_maj)
    W, H = 64, 48
    dens = cloud_density(24)
    if medium == "grid":
        scene = grid_scene(dens, (24, 24, 24), (0.02, 0.03, 0.04), (0.9, 0.8, 0.7), g=0.5, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    else:
        scene = nvdb_scene(dens, (24, 24, 24), (0.02, 0.03, 0.04), (0.9, 0.8, 0.7), g=0.5, index_min=(-3, 2, 0), voxel=(0.066, 0.0625, 0.058),
                           origin=(-0.6, -0.93, -0.5), density_offset=0.02, majorant_scale=1.25, W=W, H=H)
    # pass 1: the resampling routine records the primary rays' ratio-tracking transmittance
    prm = P.app_f_params()
    prm.vspsamplingmethod = P.VSP_RESAMPLING
    prm.storeTrBuffer = 1
    g = P.Renderer(scene, prm, W, H, seed=3)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=3)
    g.render_wave(0, 3); c.render_wave(0, 3)          # several samples of a pixel in one launch: in order
    g.post_process_wave(); c.post_process_wave()
    g.render_wave(3, 4); c.render_wave(3, 4)
    pix = np.array([[5, 7], [20, 30]], dtype=np.int32)
    g.trace_paths(pix, np.array([9, 9], dtype=np.int32))   # debug traces do not feed the buffer
    tg, spp = g.tr_buffer()
    tc = c.tr_buffer()
    assert np.all(spp == 4)
    same = np.all(tg.view(np.uint32) == tc.view(np.uint32), axis=2)
    print("%s TrBuffer bit-identical pixels %.5f" % (medium, same.mean()))
    assert same.mean() == 1.0
    assert 0.05 < tc.mean() < 0.98
    g.close(); c.close()
    # pass 2: NDS+ with the stored buffer
    prm = P.app_f_params()
    prm.vspsamplingmethod = P.VSP_NDS
    prm.collisionProbabilityBias = 1
    g = P.Renderer(scene, prm, W, H, seed=3)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=3)
    g.set_tr_buffer(tc); c.set_tr_buffer(tc)
    rng = np.random.default_rng(19)
    n = 20000
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    exact = np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1)
    ok = np.all(np.abs(Lg - Lc) <= 1e-4 * np.abs(Lc) + 1e-6, axis=1)
    print("%s NDS+ paths: same segments %.5f within tol %.5f bit-identical %.5f" % (medium, np.mean(sg == sc), ok.mean(), exact.mean()))
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    for w in range(4):
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    fg, fc = g.film(), c.film()
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    relmse = np.mean((ig - ic) ** 2 / (ic ** 2 + 1e-4))
    print("%s NDS+ film relMSE %.3e" % (medium, relmse))
    assert relmse <= 1e-4
    tg2, _ = g.tr_buffer()
    assert np.array_equal(tg2, tc)   # a loaded buffer is not recorded into
    # the bias really changes the walk: plain NDS on the same samples gives other paths
    prm.collisionProbabilityBias = 0
    g0 = P.Renderer(scene, prm, W, H, seed=3)
    L0, s0 = g0.trace_paths(pix, si)
    assert np.mean(np.any(L0 != Lg, axis=1)) > 0.1
    g0.close(); g.close(); c.close()


CLOUD_SWEEP = [
    dict(),                                                      # baseline
    dict(usenee=0), dict(maxdepth=0), dict(maxdepth=1), dict(maxdepth=9, minrrdepth=3),
    dict(vspguiding=0), dict(vspmisratio=1.0), dict(vspmisratio=0.0), dict(vspcriterion=0),
    dict(_sigma=(0.5, 0.0)),                                     # pure absorber: no scattering event is ever selected
    dict(_sigma=((0.3, 0.1, 0.02), (0.2, 0.9, 1.6))),            # chromatic: the generic (non-grey) walk instantiations
    dict(_dens="zero"),                                          # every brick empty: no density storage at all
    dict(_dens="one_voxel"),                                     # a single non-zero voxel in an otherwise empty grid
    dict(_dens="tiny"),                                          # a 1 x 1 x 1 grid
    dict(_dens="slab"),                                          # 5 x 9 x 3: partial bricks on every axis
    dict(_res=(1, 1)), dict(_res=(9, 7)),                        # less than one tile; ragged tiles
    dict(_kind="nvdb"), dict(_kind="nvdb", usenee=0), dict(_kind="nvdb", _dens="slab"),
]


@pytest.mark.parametrize("case", range(len(CLOUD_SWEEP)))
def test_cloud_parameter_sweep_vs_oracle(gpu_pkg, case):
    """The heterogeneous path's options and degenerate inputs one at a time: the wavefront pipeline and the per-lane kernel
    render the same film bit for bit over 3 waves (with the VSP-buffer updates), the film equals the oracle's up to the
    float-vs-double accumulation, and replayed paths are the oracle's bit for bit."""
    from scenes import cloud_density, grid_scene, nvdb_scene
    P = gpu_pkg
    kw = dict(CLOUD_SWEEP[case])
    W, H = kw.pop("_res", (48, 32))
    dens_kind = kw.pop("_dens", "cloud")
    if dens_kind == "cloud":
        dens, n = cloud_density(24), (24, 24, 24)
    elif dens_kind == "zero":
        dens, n = np.zeros(20 * 20 * 20, dtype=np.float32), (20, 20, 20)
    elif dens_kind == "one_voxel":
        d = np.zeros((20, 20, 20), dtype=np.float32)   # [z][y][x]
        d[11, 9, 10] = 3.0
        dens, n = d.reshape(-1), (20, 20, 20)
    elif dens_kind == "tiny":
        dens, n = np.array([0.8], dtype=np.float32), (1, 1, 1)
    else:
        rng = np.random.default_rng(5)
        dens, n = rng.random(5 * 9 * 3).astype(np.float32), (5, 9, 3)
    sa, ss = kw.pop("_sigma", (0.08, 7.9))
    kind = kw.pop("_kind", "grid")
    if kind == "grid":
        scene = grid_scene(dens, n, sa, ss, g=0.6, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    else:
        vox = tuple(1.6 / n[k] for k in range(3))
        scene = nvdb_scene(dens, n, sa, ss, g=0.6, index_min=(-2, 1, 0), voxel=vox, origin=(-0.75 + 2 * vox[0], -0.8 - vox[1], -0.5),
                           density_offset=0.01, majorant_scale=1.1, W=W, H=H)
    prm = P.app_f_params()
    for k, v in kw.items():
        setattr(prm, k, v)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=case)
    films, names = [], set()
    for kernel in (None, "lane"):
        if kernel:
            os.environ["VSPG_KERNEL"] = kernel
        try:
            g = P.Renderer(scene, prm, W, H, seed=case)
            names.add(g.kernel_name())
            for w in range(3):
                g.render_wave(w, w + 1)
                g.post_process_wave()
            films.append(g.film())
            if kernel is None:
                rng = np.random.default_rng(case)
                pix = np.stack([rng.integers(0, W, 3000), rng.integers(0, H, 3000)], axis=1).astype(np.int32)
                si = rng.integers(0, 64, 3000).astype(np.int32)
                Lg, sg = g.trace_paths(pix, si)
            g.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
    assert len(names) == 2 and any(k.startswith("k_wf_") for k in names), names
    assert np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32))
    for w in range(3):
        c.render_wave(w, w + 1)
        c.post_process_wave()
    fc = c.film()
    assert np.array_equal(films[0][..., 3], fc[..., 3])
    ig, ic = films[0][..., :3] / films[0][..., 3:4], fc[..., :3] / fc[..., 3:4]
    assert np.mean((ig - ic) ** 2 / (ic ** 2 + 1e-4)) <= 1e-10
    Lc, sc = c.trace_paths(pix, si)
    assert np.array_equal(sg, sc)
    assert np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    c.close()


@pytest.mark.parametrize("lescale", ["grid", "default"])
def test_emissive_grid_vs_oracle(gpu_pkg, lescale):
    """GridMedium emission (media.h:326-342): Le = LeScale.Lookup(p) * Le_spec, picked up by the delta-tracking
    callback (:895-906): device == oracle path by path, film and counters."""
    from scenes import cloud_density, grid_scene
    P = gpu_pkg
    W, H = 64, 48
    dens = cloud_density(24)
    scene = grid_scene(dens, (24, 24, 24), (0.3, 0.35, 0.4), (1.6, 1.4, 1.2), g=0.4, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    scene.medium.Le[:] = (2.5, 1.25, 0.4)
    if lescale == "grid":
        rng = np.random.default_rng(31)
        le = np.clip(rng.random((6, 5, 7)).astype(np.float32) * 2 - 0.6, 0, None).astype(np.float32)  # some cells emit nothing
        scene.medium.le_scale = le.ctypes.data_as(C.POINTER(C.c_float))
        scene.medium.le_nz, scene.medium.le_ny, scene.medium.le_nx = le.shape
    prm = P.app_f_params()
    prm.vspsamplingmethod = P.VSP_NDS
    g = P.Renderer(scene, prm, W, H, seed=3)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=3)
    rng = np.random.default_rng(23)
    n = 20000
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    exact = np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1)
    ok = np.all(np.abs(Lg - Lc) <= 1e-4 * np.abs(Lc) + 1e-6, axis=1)
    print("emissive (%s) paths: same segments %.5f within tol %.5f bit-identical %.5f" % (lescale, np.mean(sg == sc), ok.mean(), exact.mean()))
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    for w in range(4):
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    fg, fc = g.film(), c.film()
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    relmse = np.mean((ig - ic) ** 2 / (ic ** 2 + 1e-4))
    print("emissive (%s) film relMSE %.3e" % (lescale, relmse))
    assert relmse <= 1e-4
    # the emission is really there: the same render without Le is darker
    scene.medium.Le[:] = (0, 0, 0)
    g0 = P.Renderer(scene, prm, W, H, seed=3)
    for w in range(4):
        g0.render_wave(w, w + 1); g0.post_process_wave()
    f0 = g0.film()
    assert (fg[..., :3].sum() - f0[..., :3].sum()) / f0[..., :3].sum() > 0.05
    g0.close(); g.close(); c.close()


# ---------------------------------------------------------------------------------------------
# NanoVDBMedium semantics over a dense copy of the grid: 64^3 majorants in HBM, index-space trilinear
# fetch with zero background, densityoffset / majorantscale
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def nvdb_pair(gpu_pkg):
    from scenes import cloud_density, nvdb_scene
    P = gpu_pkg
    W, H = 64, 48
    dens = cloud_density(24)
    scene = nvdb_scene(dens, (24, 24, 24), (0.05, 0.08, 0.1), (3.0, 2.6, 2.2), g=0.5, index_min=(-3, 2, 0), voxel=(0.066, 0.0625, 0.058),
                       origin=(-0.6, -0.93, -0.5), density_offset=0.02, majorant_scale=1.25, W=W, H=H)
    prm = P.app_f_params()
    g = P.Renderer(scene, prm, W, H, seed=3)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=3)
    yield P, g, c
    g.close()
    c.close()


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_nvdb_free_flight_vs_oracle(nvdb_pair, variant):
    P, g, c = nvdb_pair
    rng = np.random.default_rng(70 + variant)
    qs = []
    for i in range(20000):
        d = rng.normal(size=3)
        d = d / np.linalg.norm(d) * rng.uniform(0.5, 2.0)
        qs.append(P.VspgTmajQuery(P.f3(*rng.uniform(-1, 1, 3)), P.f3(*d), float(rng.uniform(0.0, 3.0)), float(rng.random()),
                                  float(rng.random()), float(rng.random()), float(rng.random()) if i % 5 else -1.0,
                                  int(rng.integers(0, 3)), int(rng.integers(0, 4))))
    go, co = g.sample_tmaj_batch(variant, qs), c.sample_tmaj_batch(variant, qs)
    ncb = 0
    for a, b in zip(go, co):
        assert a.n_callbacks == b.n_callbacks
        assert list(a.T_maj) == list(b.T_maj) and list(a.r_u_factor) == list(b.r_u_factor) and a.last_t == b.last_t
        assert a.sum_sigt_over_maj == b.sum_sigt_over_maj and a.vrc == b.vrc and a.majorant_scale == b.majorant_scale
        ncb += a.n_callbacks
    assert ncb > 10000


def test_nvdb_paths_and_film_vs_oracle(nvdb_pair):
    P, g, c = nvdb_pair
    rng = np.random.default_rng(19)
    n = 20000
    pix = np.stack([rng.integers(0, g.xres, n), rng.integers(0, g.yres, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    exact = np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1)
    print("nvdb paths: same segments %.5f bit-identical %.5f" % (np.mean(sg == sc), exact.mean()))
    assert np.mean(sg == sc) == 1.0 and exact.mean() == 1.0
    for w in range(4):
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    fg, fc = g.film(), c.film()
    assert np.array_equal(fg[..., 3], fc[..., 3])
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    relmse = np.mean((ig - ic) ** 2 / (ic ** 2 + 1e-4))
    print("nvdb film relMSE %.3e" % relmse)
    assert relmse <= 1e-4
    cg, cc = g.counters(), c.counters()
    assert cg["density_queries"] > cg["volume_scatters"]
    for k in cg:
        assert abs(cg[k] - cc[k]) <= 2e-3 * cc[k] + 5, (k, cg[k], cc[k])


# ---------------------------------------------------------------------------------------------
# non-identity renderFromMedium (media.h:322, :354; util/transform.h:387-429, transform.cpp:263-303)
# ---------------------------------------------------------------------------------------------
def _placed(P, scene, kind):
    """rotate the cloud about a tilted axis, squash it and move it off-centre (still inside the box)"""
    import math
    ang = math.radians(33.0)
    ax = np.array([0.3, 1.0, -0.2]); ax /= np.linalg.norm(ax)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + math.sin(ang) * K + (1 - math.cos(ang)) * (K @ K)
    S = np.diag([0.8, 0.6, 0.7])
    M = np.eye(4)
    M[:3, :3] = R @ S
    M[:3, 3] = (0.1, -0.15, 0.2) if kind == "grid" else (0.05, 0.1, 0.15)
    return P.set_medium_transform(scene, M.astype(np.float32))


@pytest.mark.parametrize("kind", ["grid", "nvdb"])
def test_transformed_medium_vs_oracle(gpu_pkg, kind):
    """A grid medium PLACED with a rotation * scale * translation: free flight (3 variants), 20 000 paths, the film of every
    kernel the library offers, all against the oracle -- whose ApplyInverse is pinned to the reference's by
    tests/golden/primitives.json "apply_inverse_xform"."""
    from scenes import cloud_density, grid_scene, nvdb_scene
    P = gpu_pkg
    W, H = 64, 48
    dens = cloud_density(24)
    if kind == "grid":
        scene = grid_scene(dens, (24, 24, 24), 0.08, 7.9, g=0.6, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    else:
        scene = nvdb_scene(dens, (24, 24, 24), 0.08, 7.9, g=0.6, index_min=(-3, 2, 0), voxel=(0.066, 0.0625, 0.058), origin=(-0.6, -0.93, -0.5),
                           density_offset=0.01, majorant_scale=1.1, W=W, H=H)
    _placed(P, scene, kind)
    prm = P.app_f_params()
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=4)
    g = P.Renderer(scene, prm, W, H, seed=4)
    # the transform really changes the picture
    plain = grid_scene(dens, (24, 24, 24), 0.08, 7.9, g=0.6, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H) if kind == "grid" else None
    for variant in (0, 1, 2):
        qs = _queries(P, 6000, 50 + variant)
        go, co = g.sample_tmaj_batch(variant, qs), c.sample_tmaj_batch(variant, qs)
        n_cb = 0
        for a, b in zip(go, co):
            assert a.n_callbacks == b.n_callbacks
            assert a.last_t == b.last_t and list(a.T_maj) == list(b.T_maj) and list(a.r_u_factor) == list(b.r_u_factor)
            n_cb += a.n_callbacks
        assert n_cb > 500
    rng = np.random.default_rng(23)
    n = 20000
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    if plain is not None:
        p2 = P.Renderer(plain, prm, W, H, seed=4)
        Lp, _ = p2.trace_paths(pix, si)
        p2.close()
        assert np.mean(np.all(Lp == Lg, axis=1)) < 0.5
    for w in range(3):
        c.render_wave(w, w + 1); c.post_process_wave()
    fc = c.film()
    ic = fc[..., :3] / fc[..., 3:4]
    g.close()
    films = {}
    for kernel in (None, "lane", "wg"):
        if kernel:
            os.environ["VSPG_KERNEL"] = kernel
        try:
            r = P.Renderer(scene, prm, W, H, seed=4)
            for w in range(3):
                r.render_wave(w, w + 1); r.post_process_wave()
            films[r.kernel_name()] = r.film()
            r.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
    print(kind, "kernels:", sorted(films))
    assert len(films) >= 2
    ref = next(iter(films.values()))
    for name, fg in films.items():
        assert np.array_equal(fg.view(np.uint32), ref.view(np.uint32)), name
    ig = ref[..., :3] / ref[..., 3:4]
    assert np.array_equal(ref[..., 3], fc[..., 3])
    assert np.mean(np.all(np.abs(ig - ic) <= 1e-4 * (1 + np.abs(ic)), axis=-1)) == 1.0
    c.close()


# ---------------------------------------------------------------------------------------------
# f1: triangle geometry behind a BVH (shapes.cpp:168-262, shapes.h:883-1010, cpu/aggregates.cpp:529-640)
# ---------------------------------------------------------------------------------------------
def _light_only(P, scene):
    """keep only the emissive rectangle of the fog box (the walls come as triangles)"""
    light = type(scene.quads[6]).from_buffer_copy(scene.quads[6])
    for i in range(P.VSPG_MAX_QUADS):
        scene.quads[i] = type(light)()
    scene.quads[0] = light
    scene.n_quads = 1
    return scene


def test_triangle_box_matches_rectangle_box(gpu_pkg):
    """The box's walls as a 12-triangle soup: bit-identical to the oracle (which tests every triangle, no BVH), and the same
    picture as the rectangle box up to noise (hit points differ in the last bits: b0 p0 + b1 p1 + b2 p2 vs p00 + u e1 + v e2)."""
    from scenes import box_wall_triangles
    P = gpu_pkg
    W, H = 64, 48
    prm = P.app_f_params()
    scene = _light_only(P, P.fog_box_scene(W, H))
    tris, kd = box_wall_triangles()
    P.set_triangles(scene, tris, kd)
    g = P.Renderer(scene, prm, W, H, seed=1)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=1)
    rng = np.random.default_rng(31)
    n = 20000
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    for w in range(8):
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    fg, fc = g.film(), c.film()
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    assert np.mean(np.all(np.abs(ig - ic) <= 1e-4 * (1 + np.abs(ic)), axis=-1)) == 1.0
    rect = P.Renderer(P.fog_box_scene(W, H), prm, W, H, seed=1)
    Lr, sr = rect.trace_paths(pix, si)
    rect.close()
    print("triangle box vs rectangle box: mean radiance %.4f vs %.4f, same segment counts %.4f" % (Lg.mean(), Lr.mean(), np.mean(sg == sr)))
    assert abs(Lg.mean() / Lr.mean() - 1) < 0.03 and np.mean(sg == sr) > 0.9
    g.close(); c.close()


@pytest.mark.parametrize("medium", ["fog", "cloud"])
def test_triangle_terrain_vs_oracle(gpu_pkg, medium):
    """A 20 000-triangle terrain in the box (homogeneous fog: per-lane kernel; heterogeneous cloud: the wavefront pipeline
    and the per-lane kernel): paths and film bit-identical to the oracle's brute-force intersection."""
    from scenes import cloud_density, grid_scene, heightfield_triangles
    P = gpu_pkg
    W, H = 64, 48
    prm = P.app_f_params()
    if medium == "fog":
        scene = P.fog_box_scene(W, H)
    else:
        dens = cloud_density(24)
        scene = grid_scene(dens, (24, 24, 24), 0.08, 7.9, g=0.6, bmin=(-0.8, -0.5, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    tris, kd = heightfield_triangles(100)
    assert tris.shape[0] == 20000
    P.set_triangles(scene, tris, kd)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=6)
    g = P.Renderer(scene, prm, W, H, seed=6)
    rng = np.random.default_rng(37)
    n = 3000
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    cnt = g.counters()
    g.render_wave(0, 1)
    film = g.film()
    names = {g.kernel_name()}
    # every pixel of the film equals its replayed path (and with it the oracle's path, sampled above)
    xy = np.stack(np.meshgrid(np.arange(W), np.arange(H), indexing="xy"), -1).reshape(-1, 2).astype(np.int32)
    L0, _ = g.trace_paths(xy, np.zeros(len(xy), dtype=np.int32))
    assert np.array_equal(film[..., :3].reshape(-1, 3).view(np.uint32), L0.astype(np.float32).view(np.uint32))
    Lc0, _ = c.trace_paths(xy[::7], np.zeros(len(xy[::7]), dtype=np.int32))
    assert np.array_equal(L0[::7].view(np.uint32), Lc0.view(np.uint32))
    g.close()
    if medium in ("cloud", "fog"):   # fog: the workgroup kernel's full-scene instantiation (round 4) against the per-lane kernel
        os.environ["VSPG_KERNEL"] = "lane"
        try:
            g2 = P.Renderer(scene, prm, W, H, seed=6)
            g2.render_wave(0, 1)
            names.add(g2.kernel_name())
            assert np.array_equal(g2.film().view(np.uint32), film.view(np.uint32))
            g2.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
        assert len(names) == 2
    # the terrain is really hit: the picture differs from the empty box
    e = P.Renderer(P.fog_box_scene(W, H) if medium == "fog" else grid_scene(cloud_density(24), (24, 24, 24), 0.08, 7.9, g=0.6, bmin=(-0.8, -0.5, -0.5),
                                                                          bmax=(0.8, 0.7, 0.9), W=W, H=H), prm, W, H, seed=6)
    e.render_wave(0, 1)
    assert np.mean(np.all(e.film() == film, axis=-1)) < 0.95
    e.close(); c.close()


def test_hundred_thousand_triangles(gpu_pkg):
    """SURVEY 8f row 1's size: 100 352 triangles at 256 x 192, fog; the film equals the replayed paths, sampled paths equal the
    oracle's (brute force over all triangles)."""
    from scenes import heightfield_triangles
    P = gpu_pkg
    W, H = 256, 192
    prm = P.app_f_params()
    scene = P.fog_box_scene(W, H)
    tris, kd = heightfield_triangles(224)
    assert tris.shape[0] == 100352
    P.set_triangles(scene, tris, kd)
    g = P.Renderer(scene, prm, W, H, seed=8)
    g.render_wave(0, 1)
    film = g.film()
    assert g.counters()["paths"] == W * H
    rng = np.random.default_rng(41)
    n = 4000
    xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    L, sg = g.trace_paths(xy, np.zeros(n, dtype=np.int32))
    assert np.array_equal(film[xy[:, 1], xy[:, 0], :3].view(np.uint32), L.astype(np.float32).view(np.uint32))
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=8)
    Lc, sc = c.trace_paths(xy[:300], np.zeros(300, dtype=np.int32))
    assert np.array_equal(sg[:300], sc) and np.array_equal(L[:300].view(np.uint32), Lc.view(np.uint32))
    g.close(); c.close()


# ---------------------------------------------------------------------------------------------
# infinite lights: escaped-ray contributions (:353-374), DistantLight NEE (lights.h:320-327, delta-light return :1248-1249)
# ---------------------------------------------------------------------------------------------
def _open_scene(P, W, H, medium):
    """a ground rectangle + a bumpy terrain under an open sky: uniform sky, a sun, NO emissive geometry"""
    from scenes import cloud_density, grid_scene, heightfield_triangles
    if medium == "cloud":
        scene = grid_scene(cloud_density(24), (24, 24, 24), 0.08, 7.9, g=0.6, bmin=(-0.8, -0.3, -0.5), bmax=(0.8, 0.8, 0.9), W=W, H=H)
    else:
        scene = P.fog_box_scene(W, H)
        for k in range(3):
            scene.medium.sigma_a[k] = 0.02
            scene.medium.sigma_s[k] = 0.25
    floor = type(scene.quads[0]).from_buffer_copy(scene.quads[0])
    for i in range(P.VSPG_MAX_QUADS):
        scene.quads[i] = type(floor)()
    scene.quads[0] = floor
    scene.n_quads = 1
    tris, kd = heightfield_triangles(40, y=-0.75, amp=0.2)
    P.set_triangles(scene, tris, kd)
    P.add_infinite_light(scene, P.LIGHT_UNIFORM_INFINITE, (0.35, 0.5, 0.9))
    P.add_infinite_light(scene, P.LIGHT_DISTANT, (9.0, 8.0, 6.5), (0.3, 1.0, -0.4))
    return scene


@pytest.mark.parametrize("sampler", ["power", "bvh"])
@pytest.mark.parametrize("medium", ["fog", "cloud", "fog-guided"])
def test_light_samplers_vs_oracle(gpu_pkg, sampler, medium):
    """PowerLightSampler / BVHLightSampler (lightsamplers.h:63-98, 100-430; "bvh" is the reference's default) on a scene with
    three area lights of very different power, a sky and a sun: lightSampler.Sample in SampleLd, lightSampler.PMF in the MIS
    weight of an emitter hit by a scattered ray.  Paths and films against the oracle, on every kernel that serves the scene."""
    import scenes
    P = gpu_pkg
    W, H = 64, 48
    prm = P.default_params() if medium == "fog-guided" else P.app_f_params()
    prm.lightsampler = P.LIGHTSAMPLER_POWER if sampler == "power" else P.LIGHTSAMPLER_BVH
    scene = _open_scene(P, W, H, "cloud" if medium == "cloud" else "fog")
    # three emitters of very different power and orientation: the (large, dim) floor, a small bright one-sided panel facing down,
    # a two-sided vertical panel -- beside the sky and the sun
    scene.quads[0].Le[:] = (0.25, 0.2, 0.15)

    def panel(k, p00, e1, e2, Le, two_sided):
        q = type(scene.quads[0])()
        q.p00[:], q.e1[:], q.e2[:] = p00, e1, e2
        q.Kd[:] = (0.5, 0.5, 0.5)
        q.Le[:] = Le
        q.two_sided = two_sided
        scene.quads[k] = q

    panel(1, (-0.15, 0.7, 0.1), (0.3, 0, 0), (0, 0, 0.3), (30.0, 24.0, 12.0), 0)     # n = e1 x e2 = -y: shines down
    panel(2, (-0.75, -0.2, 0.4), (0, 0.5, 0), (0, 0, 0.4), (2.0, 4.0, 9.0), 1)
    scene.n_quads = 3
    n_lights = sum(1 for k in range(scene.n_quads) if any(scene.quads[k].Le)) + scene.n_infinite_lights
    assert n_lights == 5
    g = P.Renderer(scene, prm, W, H, seed=9)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=9)
    if medium == "fog-guided":
        field = scenes.light_field(P, n=4, light=(0.3, 5.0, -0.4))
        g.set_guiding_field(field, field)
        c.set_guiding_field(field, field)
    rng = np.random.default_rng(53)
    n = 6000
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    assert Lg.mean() > 0.05 and np.isfinite(Lg).all()
    for w in range(2):
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    fg, fc = g.film(), c.film()
    assert np.array_equal(fg[..., 3], fc[..., 3])
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    assert np.mean(np.all(np.abs(ig - ic) <= 1e-5 * (1 + np.abs(ic)), axis=-1)) == 1.0
    names = {g.kernel_name()}
    g.close(); c.close()
    if medium in ("cloud", "fog"):   # the wavefront pipeline (default) and the per-lane kernel   # fog: the workgroup kernel's full-scene instantiation (round 4) against the per-lane kernel
        os.environ["VSPG_KERNEL"] = "lane"
        try:
            g2 = P.Renderer(scene, prm, W, H, seed=9)
            for w in range(2):
                g2.render_wave(w, w + 1); g2.post_process_wave()
            names.add(g2.kernel_name())
            assert np.array_equal(g2.film().view(np.uint32), fg.view(np.uint32))
            g2.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
        assert len(names) == 2, names


@pytest.mark.parametrize("medium", ["fog", "cloud"])
def test_infinite_lights_vs_oracle(gpu_pkg, medium):
    """Sky (UniformInfiniteLight: reached by escaping rays only -- its SampleLi returns nothing for the incomplete PDF) and
    sun (DistantLight: next-event estimation only past the camera ray, delta-light weighting) over an open scene whose rays
    leave through the medium (SampleDistance with tMax = Infinity): paths and film against the oracle, every kernel."""
    P = gpu_pkg
    W, H = 64, 48
    prm = P.app_f_params()
    # (round 3: the reference's default light sampler, "bvh", stands: sky and sun are its infinite lights, the lamp its one leaf)
    scene = _open_scene(P, W, H, medium)
    g = P.Renderer(scene, prm, W, H, seed=9)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=9)
    rng = np.random.default_rng(43)
    n = 6000
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    assert Lg.mean() > 0.05 and np.isfinite(Lg).all()
    for w in range(3):
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    fg, fc = g.film(), c.film()
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    assert np.array_equal(fg[..., 3], fc[..., 3])
    assert np.mean(np.all(np.abs(ig - ic) <= 1e-4 * (1 + np.abs(ic)), axis=-1)) == 1.0
    names = {g.kernel_name()}
    g.close()
    if medium in ("cloud", "fog"):   # fog: the workgroup kernel's full-scene instantiation (round 4) against the per-lane kernel
        os.environ["VSPG_KERNEL"] = "lane"
        try:
            g2 = P.Renderer(scene, prm, W, H, seed=9)
            for w in range(3):
                g2.render_wave(w, w + 1); g2.post_process_wave()
            names.add(g2.kernel_name())
            assert np.array_equal(g2.film().view(np.uint32), fg.view(np.uint32))
            g2.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
        assert len(names) == 2
    # the sun matters: without it the picture is darker
    dark = _open_scene(P, W, H, medium)
    dark.n_infinite_lights = 1
    d = oracle_lib.OracleRenderer(dark, prm, W, H, seed=9)
    Ld, _ = d.trace_paths(pix[:2000], si[:2000])
    assert Ld.mean() < 0.9 * Lc[:2000].mean()
    c.close(); d.close()


@pytest.mark.parametrize("medium", ["fog", "cloud"])
def test_guiding_with_triangles_and_infinite_lights_vs_oracle(gpu_pkg, medium):
    """The reference's DEFAULT options (directional guiding + secondary VSP) over the open scene -- triangle terrain, uniform sky,
    distant sun, rays that escape through the medium.  Query side (field uploaded): paths and films against the oracle on every
    kernel that serves the configuration.  Training side: the recorded radiance samples, including the segments
    guiding_addInfiniteLightEmission adds for escaped rays (guiding.h:759-784), are the oracle's bit for bit."""
    import scenes
    P = gpu_pkg
    W, H = 64, 48
    prm = P.default_params()
    scene = _open_scene(P, W, H, medium)
    field = scenes.light_field(P, n=4, light=(0.3, 5.0, -0.4))
    g = P.Renderer(scene, prm, W, H, seed=9)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=9)
    g.set_guiding_field(field, field)
    c.set_guiding_field(field, field)
    rng = np.random.default_rng(47)
    n = 6000
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    for w in range(3):
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    fg, fc = g.film(), c.film()
    assert np.array_equal(fg[..., 3], fc[..., 3])
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    assert np.mean(np.all(np.abs(ig - ic) <= 1e-4 * (1 + np.abs(ic)), axis=-1)) == 1.0
    names = {g.kernel_name()}
    g.close(); c.close()
    if medium == "cloud":   # (guided fog over a full scene stays on the per-lane kernel)
        os.environ["VSPG_KERNEL"] = "lane"
        try:
            g2 = P.Renderer(scene, prm, W, H, seed=9)
            g2.set_guiding_field(field, field)
            for w in range(3):
                g2.render_wave(w, w + 1); g2.post_process_wave()
            names.add(g2.kernel_name())
            assert np.array_equal(g2.film().view(np.uint32), fg.view(np.uint32))
            g2.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
        assert len(names) == 2, names
    # training side
    g = P.Renderer(scene, prm, W, H, seed=9)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=9)
    g.render_wave(0, 2)
    c.render_wave(0, 2)
    sg, sc = g.training_stats(), c.training_stats()
    assert sg["training"] == sc["training"] == 1
    assert sg["n_samples"] == sc["n_samples"] > 500 and sg["n_zero"] == sc["n_zero"]  # (an open scene: rays that escape are not sampled, :318)
    a, b = _sorted_samples(g.train_samples()), _sorted_samples(c.train_samples())
    assert a.tobytes() == b.tobytes()
    if medium == "cloud":  # (a homogeneous medium fills space: no ray ever escapes it)
        assert (a["distance"] > 1e5).sum() > 100      # samples whose next vertex is an infinite-light segment
    g.close(); c.close()


# ---------------------------------------------------------------------------------------------
# guiding cache query (own design behind the restated GuidedBSDF / GuidedPhaseFunction logic)
# ---------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def guided_pair(gpu_pkg):
    import scenes
    P = gpu_pkg
    W, H = 64, 48
    scene = P.fog_box_scene(W, H)
    scene.medium.g = 0.4
    prm = P.default_params()          # the reference's defaults: surface RIS, volume MIS, secondary VSP
    field = scenes.light_field(P, n=4)
    g = P.Renderer(scene, prm, W, H, seed=2)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=2)
    # guiding requested and no field uploaded: the renderer trains its own (a18); uploading one stops that
    assert g.training_stats()["training"] == 1
    g.set_guiding_field(field, field)
    assert g.training_stats()["training"] == 0
    c.set_guiding_field(field, field)
    yield P, g, c, field
    g.close()
    c.close()


@pytest.mark.parametrize("is_volume,gg", [(0, 0.0), (1, 0.0), (1, 0.7), (1, -0.4)])
def test_guiding_query_vs_oracle(guided_pair, is_volume, gg):
    P, g, c, field = guided_pair
    rng = np.random.default_rng(17 + is_volume)
    n = 30000
    p = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    a = rng.normal(size=(n, 3)); a = (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)
    wi = rng.normal(size=(n, 3)); wi = (wi / np.linalg.norm(wi, axis=1, keepdims=True)).astype(np.float32)
    u = rng.random((n, 2)).astype(np.float32)
    og = g.guiding_query_batch(is_volume, gg, p, a, wi, u)
    oc = c.guiding_query_batch(is_volume, gg, p, a, wi, u)
    assert np.array_equal(og["ok"], oc["ok"]) and og["ok"].all()
    for k in ("pdf", "incoming_pdf", "vsp", "pdf_s", "ws"):
        same = np.mean(og[k].view(np.uint32) == oc[k].view(np.uint32))
        print(k, "bit-identical fraction %.5f" % same)
        assert np.allclose(og[k], oc[k], rtol=1e-5, atol=1e-7)
        assert same == 1.0


def test_guided_workgroup_kernel_equals_per_lane_kernel(gpu_pkg):
    """The reference-default guided configuration (surface RIS + volume MIS + secondary VSP) with a field in place: the
    workgroup kernel (the default since round 3: lobes in registers, vertices compacted by kind, four waves per SIMD) in its
    grey / zero-null-coefficient and generic instantiations and the per-lane kernel (scratch in LDS) in both of its own
    render the same film bit for bit, at a size with many workgroups."""
    import scenes
    P = gpu_pkg
    W, H = 320, 200
    scene = P.fog_box_scene(W, H)
    scene.medium.g = 0.4
    prm = P.default_params()
    field = scenes.light_field(P, n=4)
    films = {}
    for kernel, nogrey in ((None, ""), ("wg", "1"), ("lane", ""), ("lane", "1")):
        if kernel:
            os.environ["VSPG_KERNEL"] = kernel
        if nogrey:
            os.environ["VSPG_NO_GREY_GUIDED"] = nogrey
        try:
            r = P.Renderer(scene, prm, W, H, seed=2)
            r.set_guiding_field(field, field)
            for w in range(3):
                r.render_wave(w, w + 1); r.post_process_wave()
            films[r.kernel_name()] = r.film()
            cnt = r.counters()
            assert cnt["paths"] == 3 * W * H
            r.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
            os.environ.pop("VSPG_NO_GREY_GUIDED", None)
    assert sorted(films) == ["k_render_wave<HomogeneousMedium,guided>", "k_render_wave<HomogeneousMediumT<2,true>,guided>",
                             "k_render_wave_wg2<HomogeneousMedium,guided>", "k_render_wave_wg2<HomogeneousMediumT<2,true>,guided>"], sorted(films)
    a = next(iter(films.values()))
    for name, f in films.items():
        assert np.array_equal(a.view(np.uint32), f.view(np.uint32)), name


@pytest.mark.parametrize("g,stype,vtype,sg,vg,maxdepth", [(0.0, 1, 0, 1, 1, 5),   # isotropic phase function: no product lobe at volume vertices
                                                          (0.4, 0, 1, 1, 1, 5),   # surfaces MIS, volumes RIS
                                                          (-0.6, 1, 1, 1, 1, 8),  # both RIS, backward scattering, deeper paths
                                                          (0.4, 1, 0, 0, 0, 5),   # secondary-ray VSP only: the field is asked for nothing else
                                                          (0.0, 0, 0, 1, 0, 5)])  # surfaces guided (MIS), volumes not
def test_guided_workgroup_vertex_flavours_equal_per_lane_kernel(gpu_pkg, g, stype, vtype, sg, vg, maxdepth):
    """The workgroup kernel's guided vertex (vspg_guided_wg.h: the mixture built once, every sum that does not depend on its own
    sample taken in that pass, the NEE's shadow ray after it) against the per-lane kernel's straight flow, over the flavours
    the reference's options select: films bit for bit."""
    import scenes
    P = gpu_pkg
    W, H = 192, 128
    scene = P.fog_box_scene(W, H)
    scene.medium.g = g
    prm = P.default_params()
    prm.surfaceguidingtype, prm.volumeguidingtype = stype, vtype
    prm.surfaceguiding, prm.volumeguiding, prm.maxdepth = sg, vg, maxdepth
    field = scenes.light_field(P, n=4)
    films = []
    for kernel in ("wg", "lane"):
        os.environ["VSPG_KERNEL"] = kernel
        os.environ["VSPG_NO_GREY_GUIDED"] = "1"
        try:
            r = P.Renderer(scene, prm, W, H, seed=11)
            r.set_guiding_field(field, field)
            for w in range(3):
                r.render_wave(w, w + 1); r.post_process_wave()
            assert ("_wg" in r.kernel_name()) == (kernel == "wg"), r.kernel_name()
            films.append(r.film())
            assert r.counters()["paths"] == 3 * W * H
            r.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
            os.environ.pop("VSPG_NO_GREY_GUIDED", None)
    assert np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32))


@pytest.mark.parametrize("stype,vtype", [(1, 0), (0, 1)])  # (ris, mis) = reference defaults; (mis, ris)
def test_guided_paths_and_film_vs_oracle(gpu_pkg, stype, vtype):
    import scenes
    P = gpu_pkg
    W, H = 64, 48
    scene = P.fog_box_scene(W, H)
    scene.medium.g = 0.4
    prm = P.default_params()
    prm.surfaceguidingtype, prm.volumeguidingtype = stype, vtype
    field = scenes.light_field(P, n=4)
    g = P.Renderer(scene, prm, W, H, seed=5)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=5)
    g.set_guiding_field(field, field)
    c.set_guiding_field(field, field)
    rng = np.random.default_rng(23)
    n = 30000
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    exact = np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1)
    ok = np.all(np.abs(Lg - Lc) <= 1e-4 * np.abs(Lc) + 1e-6, axis=1)
    print("guided paths (%d,%d): same segments %.5f within tol %.5f bit-identical %.5f" % (stype, vtype, np.mean(sg == sc), ok.mean(), exact.mean()))
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    for w in range(4):
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    fg, fc = g.film(), c.film()
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    relmse = np.mean((ig - ic) ** 2 / (ic ** 2 + 1e-4))
    print("guided film relMSE %.3e" % relmse)
    assert relmse <= 1e-4
    g.close()
    c.close()


# ---------------------------------------------------------------------------------------------
# a18: guiding-cache training (recording hooks restated from guiding.h; PropagateSamples / Field::Update
# are own designs, stated once in oracle/ and once in csrc/vspg_train.h)
# ---------------------------------------------------------------------------------------------
def _sorted_samples(a):
    key = np.lexsort([a["flags"], a["pdf"].view(np.uint32), a["weight"].view(np.uint32)] +
                     [a["dir"][:, k].view(np.uint32) for k in range(3)] + [a["p"][:, k].view(np.uint32) for k in range(3)])
    return a[key]


@pytest.mark.parametrize("medium", ["homogeneous", "grid", "homogeneous-deep", "homogeneous-capped"])
def test_training_samples_bit_identical_to_oracle(gpu_pkg, medium):
    """Wave 0 of a training run (field still empty -> unguided paths): the radiance samples the device
    records and propagates are the oracle's, bit for bit, as a multiset; so is the dropped-sample count.
    ("deep": maxdepth 9 -- more samples per path than a lane of k_propagate stages, so its overflow flush runs.)"""
    import scenes
    P = gpu_pkg
    W, H = 48, 40
    if medium == "grid":
        scene = scenes.grid_scene(scenes.cloud_density(16), (16, 16, 16), (0.05, 0.08, 0.1), (3.0, 2.6, 2.2), g=0.5,
                                  bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    else:
        scene = P.fog_box_scene(W, H)
        scene.medium.g = 0.3
    prm = P.default_params()
    if medium in ("homogeneous-deep", "homogeneous-capped"):
        # long paths; "capped": deeper than the 64 records a path keeps (2 * maxdepth = 80): recording stops there (NextSegment() ==
        # nullptr), the path renders on, and the propagation kernel's staging area overflows into its flush path
        prm.maxdepth = 9 if medium == "homogeneous-deep" else 40
        prm.minrrdepth = prm.maxdepth - 1   # no Russian roulette before the last vertex
        for k in range(3):
            scene.medium.sigma_a[k] = 0.01
            scene.medium.sigma_s[k] = 1.5
    g = P.Renderer(scene, prm, W, H, seed=3)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=3)
    # grid media record on the wavefront pipeline, homogeneous ones on the workgroup kernel (round 3): either way the recorder's
    # state travels in the path record
    assert g.kernel_name() == ("k_wf_walk<GridMedium,guided,train>" if medium == "grid" else "k_render_wave_wg2<HomogeneousMediumT<2,true>,guided,train>")
    # (the sample buffer holds one wave's worth -- pixels x (maxdepth + 1) -- between two updates: the capped case fills it with one)
    n_waves = 1 if medium == "homogeneous-capped" else 2
    g.render_wave(0, n_waves)
    c.render_wave(0, n_waves)
    sg, sc = g.training_stats(), c.training_stats()
    assert sg["training"] == sc["training"] == 1
    assert sg["n_samples"] == sc["n_samples"] > 500 and sg["n_zero"] == sc["n_zero"]  # (an open scene: rays that escape are not sampled, :318)
    a, b = _sorted_samples(g.train_samples()), _sorted_samples(c.train_samples())
    assert a.tobytes() == b.tobytes()
    assert set(np.unique(a["flags"])) <= {0, 1, 2, 3} and (a["flags"] & 1).any() and (~a["flags"] & 1).any()
    g.close()
    # ... and the other kernels that serve the configuration record the same samples: the per-lane training kernel, and for
    # homogeneous media the workgroup kernel's generic instantiation
    others = [({"VSPG_KERNEL": "lane"}, "k_render_wave<GridMedium,guided,train>" if medium == "grid" else "k_render_wave<HomogeneousMediumT<2,true>,guided,train>")]
    if medium != "grid":
        others.append(({"VSPG_NO_GREY_GUIDED": "1"}, "k_render_wave_wg2<HomogeneousMedium,guided,train>"))
    for env, name in others:
        os.environ.update(env)
        try:
            g2 = P.Renderer(scene, prm, W, H, seed=3)
            assert g2.kernel_name() == name
            g2.render_wave(0, n_waves)
            assert _sorted_samples(g2.train_samples()).tobytes() == b.tobytes(), name
            g2.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    c.close()


@pytest.mark.parametrize("case", ["grid", "nvdb", "grid-guided", "grid-emissive", "grid-nds+"])
def test_nds_on_the_pipeline_equals_per_lane_kernel(gpu_pkg, case):
    """vspsamplingmethod "nds" over a heterogeneous medium: the pipeline's shape for it -- k_wf_segment_vertex (segment and vertex
    in the lane) + k_wf_shadow_walk (the NEE's walk, regrouped, its result added by the next launch) -- renders the per-lane
    kernel's film bit for bit; NDS+ (collisionProbabilityBias with a TrBuffer), an emissive grid and the reference's default
    guiding options on top included.  (The per-lane kernel's paths are the oracle's: the NDS tests above.)"""
    import scenes
    P = gpu_pkg
    W, H = 96, 64
    dens = scenes.cloud_density(24)
    if case == "nvdb":
        scene = scenes.nvdb_scene(dens, (24, 24, 24), (0.05, 0.08, 0.1), (3.0, 2.6, 2.2), g=0.5, index_min=(-3, 2, 0), voxel=(0.066, 0.0625, 0.058),
                                  origin=(-0.6, -0.93, -0.5), density_offset=0.02, majorant_scale=1.25, W=W, H=H)
    else:
        scene = scenes.grid_scene(dens, (24, 24, 24), (0.05, 0.08, 0.1), (3.0, 2.6, 2.2), g=0.5, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    if case == "grid-emissive":
        scene.medium.Le[:] = (2.5, 1.25, 0.4)
    guided = case == "grid-guided"
    prm = P.default_params() if guided else P.app_f_params()
    prm.vspsamplingmethod = P.VSP_NDS
    tr = None
    if case == "grid-nds+":
        prm.collisionProbabilityBias = 1
        tr = np.random.default_rng(5).random((H, W, 3)).astype(np.float32) * 0.9 + 0.05
    field = scenes.light_field(P, n=4) if guided else None
    films = {}
    for kernel in (None, "lane"):
        if kernel:
            os.environ["VSPG_KERNEL"] = kernel
        try:
            r = P.Renderer(scene, prm, W, H, seed=14)
            if field is not None:
                r.set_guiding_field(field, field)
            if tr is not None:
                r.set_tr_buffer(tr)
            for w in range(4):
                r.render_wave(w, w + 1); r.post_process_wave()
            films[r.kernel_name()] = (r.film(), r.counters())
            r.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
    print(case, sorted(films))
    assert len(films) == 2 and any(k.startswith("k_wf_segment_vertex<") for k in films) and any(k.startswith("k_render_wave<") for k in films)
    (fa, ca), (fb, cb) = films.values()
    assert ca == cb
    assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32))


@pytest.mark.parametrize("guided", [False, True])
def test_wavefront_pipeline_concurrent_walks_equal_single_stream_order(gpu_pkg, guided):
    """The shadow walk of iteration i runs on a second stream beside the distance walk of iteration i + 1 (own job fields, its
    result added by the next vertex kernel).  The same pass with every kernel on ONE stream (VSPG_WF_SERIAL=1) must give the
    same film and counters bit for bit -- at 1080p, where the two kernels really overlap -- twice in a row (a race between the
    two would not repeat)."""
    P = gpu_pkg
    W, H = 1920, 1080
    scene = P.cloud_box_scene(W, H, 64)
    prm = P.default_params() if guided else P.app_f_params()
    field = None
    if guided:
        import scenes
        field = scenes.light_field(P, n=4)
    results = []
    for serial in ("0", "1", "0"):
        os.environ["VSPG_WF_SERIAL"] = serial
        try:
            r = P.Renderer(scene, prm, W, H, seed=4)
            if field is not None:
                r.set_guiding_field(field, field)
            assert r.kernel_name().startswith("k_wf_")
            for w in range(2):
                r.render_wave(w, w + 1); r.post_process_wave()
            results.append((r.film(), r.counters()))
            r.close()
        finally:
            os.environ.pop("VSPG_WF_SERIAL", None)
    for f, c in results[1:]:
        assert c == results[0][1]
        assert np.array_equal(f.view(np.uint32), results[0][0].view(np.uint32))


@pytest.mark.parametrize("shape", ["box", "box-guided", "scene", "scene-nvdb"])
def test_merged_walk_kernel_equals_the_two_walk_kernels(gpu_pkg, shape):
    """Round 5: the shadow walks of iteration i and the distance walks of iteration i + 1 as ONE persistent kernel over one job
    stream (k_wf_walk: the default for boundary scenes and guided pipelines) against the two kernels side by side on two streams
    (the default for dense unguided clouds): VSPG_WF_MERGED=1 / 0 give the same film and counters bit for bit, at a size where
    wavefronts really hold both kinds of job."""
    P = gpu_pkg
    W, H = 960, 540
    guided = shape.endswith("guided")
    scene = P.cloud_box_scene(W, H, 64) if shape.startswith("box") else P.cloud_scene(W, H, 64, nvdb=shape.endswith("nvdb"))
    prm = P.default_params() if guided else P.app_f_params()
    if not guided:
        prm.vspsamplingmethod = P.VSP_RESAMPLING
    field = None
    if guided:
        import scenes
        field = scenes.light_field(P, n=4)
    results = []
    for merged in ("1", "0", "1"):
        os.environ["VSPG_WF_MERGED"] = merged
        try:
            r = P.Renderer(scene, prm, W, H, seed=6)
            if field is not None:
                r.set_guiding_field(field, field)
            assert r.kernel_name().startswith("k_wf_walk" if merged == "1" else "k_wf_dist_walk")
            for w in range(3):
                r.render_wave(w, w + 1); r.post_process_wave()
            results.append((r.film(), r.counters()))
            r.close()
        finally:
            os.environ.pop("VSPG_WF_MERGED", None)
    assert results[0][1]["shadow_density_queries"] > 0 and results[0][1]["density_queries"] > 0
    for f, c in results[1:]:
        assert c == results[0][1]
        assert np.array_equal(f.view(np.uint32), results[0][0].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["scene", "scene-nvdb", "box-guided"])
def test_job_cursors_do_not_change_the_render(gpu_pkg, shape):
    """Round 5: k_wf_walk's job stream dealt out to eight striped cursors (boundary scenes' default) or handed out by one (everything
    else's): VSPG_WF_SEGS=8 / 1 decide who runs which walk when, nothing else -- same film, same counters, bit for bit; at a size
    where a list spans many stripes and wavefronts move from cursor to cursor."""
    P = gpu_pkg
    W, H = 960, 540
    guided = shape.endswith("guided")
    scene = P.cloud_box_scene(W, H, 64) if shape.startswith("box") else P.cloud_scene(W, H, 64, nvdb=shape.endswith("nvdb"))
    prm = P.default_params() if guided else P.app_f_params()
    field = None
    if guided:
        import scenes
        field = scenes.light_field(P, n=4)
    results = []
    for segs in ("8", "1", "8"):
        os.environ["VSPG_WF_SEGS"] = segs
        try:
            r = P.Renderer(scene, prm, W, H, seed=9)
            if field is not None:
                r.set_guiding_field(field, field)
            assert r.kernel_name().startswith("k_wf_walk")
            for w in range(3):
                r.render_wave(w, w + 1); r.post_process_wave()
            results.append((r.film(), r.counters()))
            r.close()
        finally:
            os.environ.pop("VSPG_WF_SEGS", None)
    assert results[0][1]["density_queries"] > 0
    for f, c in results[1:]:
        assert c == results[0][1]
        assert np.array_equal(f.view(np.uint32), results[0][0].view(np.uint32))


@pytest.mark.parametrize("kind", ["grid", "nvdb"])
def test_guided_wavefront_pipeline_equals_per_lane_kernel(gpu_pkg, kind):
    """The reference-default guided configuration over a heterogeneous medium with a field in place (config 5's query side):
    the wavefront pipeline -- whole guided vertex in k_wf_vertex, a shadow-walked NEE result added by the next k_wf_vertex -- renders the
    per-lane guided kernel's film bit for bit, and both agree with the oracle's."""
    import scenes
    P = gpu_pkg
    W, H = 96, 64
    dens = scenes.cloud_density(24)
    if kind == "grid":
        scene = scenes.grid_scene(dens, (24, 24, 24), (0.05, 0.08, 0.1), (3.0, 2.6, 2.2), g=0.5, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    else:
        scene = scenes.nvdb_scene(dens, (24, 24, 24), (0.05, 0.08, 0.1), (3.0, 2.6, 2.2), g=0.5, index_min=(-3, 2, 0), voxel=(0.066, 0.0625, 0.058),
                                  origin=(-0.6, -0.93, -0.5), density_offset=0.02, majorant_scale=1.25, W=W, H=H)
        _placed(P, scene, "nvdb")
    prm = P.default_params()
    field = scenes.light_field(P, n=4)
    films = {}
    for kernel in (None, "lane"):
        if kernel:
            os.environ["VSPG_KERNEL"] = kernel
        try:
            r = P.Renderer(scene, prm, W, H, seed=12)
            r.set_guiding_field(field, field)
            for w in range(4):
                r.render_wave(w, w + 1); r.post_process_wave()
            films[r.kernel_name()] = (r.film(), r.counters())
            r.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
    print(kind, sorted(films))
    assert len(films) == 2 and any(k.startswith("k_wf_") and "guided" in k for k in films) and any(k.startswith("k_render_wave<") for k in films)
    (fa, ca), (fb, cb) = films.values()
    assert ca == cb
    assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32))
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=12)
    c.set_guiding_field(field, field)
    for w in range(4):
        c.render_wave(w, w + 1); c.post_process_wave()
    fc = c.film()
    c.close()
    assert np.array_equal(fa[..., 3], fc[..., 3])
    ia, ic = fa[..., :3] / fa[..., 3:4], fc[..., :3] / fc[..., 3:4]
    assert np.mean(np.all(np.abs(ia - ic) <= 1e-4 * (1 + np.abs(ic)), axis=-1)) == 1.0


GUIDED_CLOUD_SWEEP = [
    dict(maxdepth=1),                                   # two records per path (pss->Reserve(2 * maxdepth))
    dict(maxdepth=0),                                   # thirty records reserved, none used past the first vertex
    dict(maxdepth=8, minrrdepth=4),
    dict(usenee=0),
    dict(surfaceguiding=0), dict(volumeguiding=0),
    dict(surfaceguidingtype=0, volumeguidingtype=1),    # surface MIS, volume RIS
    dict(vspsecondaryguiding=0), dict(vspguiding=0),
    dict(vspcriterion=0),
    dict(rrguiding=1, maxdepth=8, minrrdepth=1),        # guided Russian roulette on the pipeline (contribution estimate ready from wave 1 on)
    dict(rrguiding=1, surfacerrguiding=0),
]


@pytest.mark.parametrize("case", range(len(GUIDED_CLOUD_SWEEP)))
def test_cloud_guided_sweep_vs_oracle(gpu_pkg, case):
    """Guiding options one at a time over the heterogeneous medium.  Query side (field uploaded): the wavefront pipeline's film
    equals the per-lane guided kernel's bit for bit and replayed paths are the oracle's.  Training side: the samples the
    pipeline records in two passes are the oracle's as a multiset."""
    import scenes
    P = gpu_pkg
    W, H = 48, 32
    kw = dict(GUIDED_CLOUD_SWEEP[case])
    scene = scenes.grid_scene(scenes.cloud_density(24), (24, 24, 24), 0.08, 7.9, g=0.6, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    prm = P.default_params()
    for k, v in kw.items():
        setattr(prm, k, v)
    field = scenes.light_field(P, n=4)
    wants = prm.surfaceguiding or prm.volumeguiding or (prm.vspguiding and prm.vspsecondaryguiding)
    films, names = [], set()
    for kernel in (None, "lane"):
        if kernel:
            os.environ["VSPG_KERNEL"] = kernel
        try:
            g = P.Renderer(scene, prm, W, H, seed=20 + case)
            if wants:
                g.set_guiding_field(field, field)
            names.add(g.kernel_name())
            if kernel is None:  # (before any wave: the replay sees the image-space VSP buffer's state, and the oracle below is fresh)
                rng = np.random.default_rng(case)
                pix = np.stack([rng.integers(0, W, 3000), rng.integers(0, H, 3000)], axis=1).astype(np.int32)
                si = rng.integers(0, 64, 3000).astype(np.int32)
                Lg, sg = g.trace_paths(pix, si)
            for w in range(3):
                g.render_wave(w, w + 1); g.post_process_wave()
            films.append(g.film())
            g.close()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
    assert len(names) == 2, names
    assert np.array_equal(films[0].view(np.uint32), films[1].view(np.uint32))
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=20 + case)
    if wants:
        c.set_guiding_field(field, field)
    Lc, sc = c.trace_paths(pix, si)
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    c.close()
    if not wants:
        return
    g = P.Renderer(scene, prm, W, H, seed=20 + case)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=20 + case)
    assert "train" in g.kernel_name() and g.kernel_name().startswith("k_wf_")
    g.render_wave(0, 2)
    c.render_wave(0, 2)
    tg, tc = g.training_stats(), c.training_stats()
    assert tg["n_samples"] == tc["n_samples"] and tg["n_zero"] == tc["n_zero"] and tg["n_dropped"] == tc["n_dropped"]
    assert _sorted_samples(g.train_samples()).tobytes() == _sorted_samples(c.train_samples()).tobytes()
    g.close(); c.close()


def test_config5_standin_vs_oracle(gpu_pkg):
    """Config 5 ("explosion": NanoVDB medium with a temperature grid, secondary-ray VSPG, cache train + query) in one scene:
    a NanoVDBMedium-semantics medium with density AND temperature grids, placed by a rotation * scale * translation,
    rendered with the reference's DEFAULT guiding options.  (1) query side: with an uploaded field 20 000 paths are the
    oracle's bit for bit; (2) training side: wave 0/1 of an in-loop training run record the oracle's samples bit for bit."""
    import scenes
    P = gpu_pkg
    W, H = 64, 48
    dens = scenes.cloud_density(24)
    scene = scenes.nvdb_scene(dens, (24, 24, 24), (0.05, 0.08, 0.1), (3.0, 2.6, 2.2), g=0.5, index_min=(-3, 2, 0), voxel=(0.066, 0.0625, 0.058),
                              origin=(-0.6, -0.93, -0.5), density_offset=0.02, majorant_scale=1.25, W=W, H=H)
    temp = (dens * 2500 + 300).astype(np.float32)
    scene.medium.temperature = temp.ctypes.data_as(C.POINTER(C.c_float))
    scene.medium.nvdb_le_scale, scene.medium.temperature_offset, scene.medium.temperature_scale = 1.0, 100.0, 1.5
    _placed(P, scene, "nvdb")
    prm = P.default_params()      # surface RIS, volume MIS, primary + secondary VSP guiding, resampling
    assert prm.vspsecondaryguiding and prm.volumeguiding and prm.surfaceguiding
    field = scenes.light_field(P, n=4)
    g = P.Renderer(scene, prm, W, H, seed=8)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=8)
    g.set_guiding_field(field, field)
    c.set_guiding_field(field, field)
    rng = np.random.default_rng(29)
    n = 20000
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    print("config-5 stand-in: segments/path %.2f, bit-identical %.5f" % (sc.mean(), np.mean(np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1))))
    assert np.array_equal(sg, sc) and np.array_equal(Lg.view(np.uint32), Lc.view(np.uint32))
    for w in range(3):
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    fg, fc = g.film(), c.film()
    assert np.array_equal(fg[..., 3], fc[..., 3])
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    assert np.mean(np.all(np.abs(ig - ic) <= 1e-4 * (1 + np.abs(ic)), axis=-1)) == 1.0
    (vg, rg), (vc, rc) = g.vsp_buffer(), c.vsp_buffer()
    assert rg and rc and np.mean(np.abs(vg - vc) <= 1e-6) == 1.0
    g.close(); c.close()
    # training side
    g = P.Renderer(scene, prm, W, H, seed=8)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=8)
    g.render_wave(0, 2)
    c.render_wave(0, 2)
    sg, sc = g.training_stats(), c.training_stats()
    assert sg["training"] == sc["training"] == 1
    assert sg["n_samples"] == sc["n_samples"] > 500 and sg["n_zero"] == sc["n_zero"]  # (an open scene: rays that escape are not sampled, :318)
    a, b = _sorted_samples(g.train_samples()), _sorted_samples(c.train_samples())
    assert a.tobytes() == b.tobytes()
    g.close(); c.close()


def _field_arrays(P, regs, n):
    out = {}
    for name in ("weight", "kappa", "distance", "vsp"):
        out[name] = np.array([[getattr(regs[i], name)[k] for k in range(P.VSPG_FIELD_LOBES)] for i in range(n)])
    out["mu"] = np.array([[[regs[i].mu[a][k] for k in range(P.VSPG_FIELD_LOBES)] for a in range(3)] for i in range(n)])
    out["pivot"] = np.array([[regs[i].pivot[a] for a in range(3)] for i in range(n)])
    out["n_lobes"] = np.array([regs[i].n_lobes for i in range(n)])
    return out


def test_training_update_matches_oracle(gpu_pkg):
    """One Field::Update on identical samples: same tree, same regions; lobe parameters within the
    float-summation-order tolerance (float atomics on the device, doubles in the oracle)."""
    P = gpu_pkg
    W, H = 160, 120
    scene = P.fog_box_scene(W, H)
    prm = P.default_params()
    g = P.Renderer(scene, prm, W, H, seed=1)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=1)
    g.render_wave(0, 1)
    c.render_wave(0, 1)
    g.post_process_wave()
    c.post_process_wave()
    sg, sc = g.training_stats(), c.training_stats()
    assert sg["iteration"] == sc["iteration"] == 1
    assert sg["n_nodes"] == sc["n_nodes"] and sg["n_regions"] == sc["n_regions"]
    assert sg["n_regions"][0] > 1 and sg["n_regions"][1] > 1  # both fields split at this sample count
    for vol in (0, 1):
        ng, rg, nng, nrg = g.get_guiding_field(vol)
        nc, rc, nnc, nrc = c.get_guiding_field(vol)
        for i in range(nng):
            assert ng[i].packed == nc[i].packed
            assert abs(ng[i].split - nc[i].split) <= 1e-5
        fg, fc = _field_arrays(P, rg, nrg), _field_arrays(P, rc, nrc)
        assert np.array_equal(fg["n_lobes"], fc["n_lobes"])
        assert np.allclose(fg["pivot"], fc["pivot"], atol=1e-4)
        assert np.allclose(fg["weight"], fc["weight"], rtol=2e-3, atol=1e-5)
        assert np.allclose(fg["mu"], fc["mu"], atol=2e-3)
        assert np.allclose(fg["kappa"], fc["kappa"], rtol=1e-2)
        assert np.allclose(fg["vsp"], fc["vsp"], atol=2e-3)
        fin = np.isfinite(fc["distance"])
        assert np.array_equal(fin, np.isfinite(fg["distance"]))
        assert np.allclose(fg["distance"][fin], fc["distance"][fin], rtol=1e-2)
    g.close()
    c.close()


@pytest.mark.parametrize("medium", ["homogeneous", "grid"])
def test_sharded_training_fits_one_field_from_all_samples(gpu_pkg, medium):
    """SURVEY 8e for the guiding field: two shards (two renderers, two host threads standing in for two ranks) with the exchange
    hook installed -- Field::Update sums its sufficient statistics over the shards -- fit the SAME field, bit for bit, and that
    field is the one ONE renderer fits that renders both sample indices in a step (the oracle's field_update on the union of
    the samples), up to float summation order."""
    import threading
    import scenes
    P = gpu_pkg
    # device <-> host copies through the HIP runtime the library itself is bound to (dlsym on its handle searches its dependencies;
    # a second runtime in the process -- torch bundles one -- could not open the device)
    lib = P.load()
    lib.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.hipMemcpy.restype = C.c_int
    lib.hipDeviceSynchronize.restype = C.c_int
    W, H = 48, 40
    if medium == "grid":
        scene = scenes.grid_scene(scenes.cloud_density(16), (16, 16, 16), (0.05, 0.08, 0.1), (3.0, 2.6, 2.2), g=0.5,
                                  bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    else:
        scene = P.fog_box_scene(W, H)
        scene.medium.g = 0.3
    prm = P.default_params()
    shards = [P.Renderer(scene, prm, W, H, seed=3, shard_index=k, shard_count=2) for k in range(2)]
    barrier = threading.Barrier(2)
    slots, calls, errors = {}, [0, 0], []

    def make_hook(k):
        def hook(ptr, n, stream):
            calls[k] += 1
            slots[k] = (ptr, n)
            assert lib.hipDeviceSynchronize() == 0
            barrier.wait(timeout=60)
            if k == 0:   # "rank 0" sums the two shards' buffers and hands the sum to both
                assert slots[0][1] == slots[1][1]
                a, b = np.empty(n, np.float32), np.empty(n, np.float32)
                assert lib.hipMemcpy(a.ctypes.data, slots[0][0], 4 * n, 2) == 0 and lib.hipMemcpy(b.ctypes.data, slots[1][0], 4 * n, 2) == 0
                t = a + b
                assert lib.hipMemcpy(slots[0][0], t.ctypes.data, 4 * n, 1) == 0 and lib.hipMemcpy(slots[1][0], t.ctypes.data, 4 * n, 1) == 0
            barrier.wait(timeout=60)
        return hook

    for k in range(2):
        shards[k].set_exchange(make_hook(k))
        shards[k].render_wave(0, 2)          # shard k renders sample index k of the step

    def step(k):
        try:
            shards[k].post_process_step(2)
        except Exception as e:  # noqa
            errors.append(e)
            barrier.abort()
    th = [threading.Thread(target=step, args=(k,)) for k in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errors, errors
    assert calls[0] == calls[1] == 5        # count, weight sum, 3 accumulation points (both fields in each)
    st0, st1 = shards[0].training_stats(), shards[1].training_stats()
    assert st0["iteration"] == st1["iteration"] == 1 and st0["n_nodes"] == st1["n_nodes"] and st0["n_regions"] == st1["n_regions"]
    # one renderer stepping two waves: the oracle
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=3)
    c.render_wave(0, 2)
    c.post_process_step(2, None)
    sc = c.training_stats()
    assert sc["iteration"] == 1 and sc["n_nodes"] == st0["n_nodes"] and sc["n_regions"] == st0["n_regions"]
    assert st0["n_samples"] + st1["n_samples"] == sc["n_samples"]   # the shards' samples are the union renderer's
    for vol in (0, 1):
        n0, r0, nn0, nr0 = shards[0].get_guiding_field(vol)
        n1, r1, nn1, nr1 = shards[1].get_guiding_field(vol)
        assert (nn0, nr0) == (nn1, nr1)
        assert bytes(n0)[:nn0 * C.sizeof(P.VspgKdNode)] == bytes(n1)[:nn1 * C.sizeof(P.VspgKdNode)]          # the two shards' fields:
        assert bytes(r0)[:nr0 * C.sizeof(P.VspgFieldRegion)] == bytes(r1)[:nr1 * C.sizeof(P.VspgFieldRegion)]  # the same bits
        nc, rc, nnc, nrc = c.get_guiding_field(vol)
        assert (nn0, nr0) == (nnc, nrc)
        for i in range(nn0):
            assert n0[i].packed == nc[i].packed and abs(n0[i].split - nc[i].split) <= 1e-5
        fg, fc = _field_arrays(P, r0, nr0), _field_arrays(P, rc, nrc)
        assert np.array_equal(fg["n_lobes"], fc["n_lobes"])
        assert np.allclose(fg["pivot"], fc["pivot"], atol=1e-4)
        assert np.allclose(fg["weight"], fc["weight"], rtol=2e-3, atol=1e-5)
        assert np.allclose(fg["mu"], fc["mu"], atol=2e-3)
        assert np.allclose(fg["kappa"], fc["kappa"], rtol=1e-2)
        assert np.allclose(fg["vsp"], fc["vsp"], atol=2e-3)
    for r in shards:
        r.close()
    c.close()


def test_training_in_loop_unbiased_and_useful(gpu_pkg):
    """Train + query in-loop (cfg 5).  With NEE the trained-guided render has the unguided mean (the field
    only changes sampling densities); without NEE -- where finding the small light is the whole problem --
    the trained field lowers the error against a NEE reference (robust metric: per-pixel relative squared
    error clipped at 4, the no-NEE estimator is heavy-tailed)."""
    P = gpu_pkg
    W, H = 96, 72
    scene = P.fog_box_scene(W, H)

    def render(guided, usenee, waves):
        prm = P.default_params()
        prm.usenee = usenee
        prm.guide_num_training_waves = 24
        if not guided:
            prm.surfaceguiding = prm.volumeguiding = prm.vspsecondaryguiding = 0
        r = P.Renderer(scene, prm, W, H, seed=9)
        for w in range(waves):
            r.render_wave(w, w + 1)
            r.post_process_wave()
        st = r.training_stats()
        f = r.film()
        r.close()
        return f[..., :3] / f[..., 3:4], st

    ref, _ = render(False, 1, 96)
    gn, st = render(True, 1, 96)
    assert st["training"] == 0 and st["iteration"] == 24 and st["n_regions"][1] > 1
    print("NEE means: unguided %.5f trained-guided %.5f" % (ref.mean(), gn.mean()))
    assert abs(gn.mean() - ref.mean()) <= 0.015 * ref.mean()
    a, _ = render(False, 0, 64)
    b, st = render(True, 0, 64)
    assert st["training"] == 0 and st["iteration"] == 24
    ea = np.mean(np.minimum((a - ref) ** 2 / (ref ** 2 + 1e-2), 4.0))
    eb = np.mean(np.minimum((b - ref) ** 2 / (ref ** 2 + 1e-2), 4.0))
    print("no-NEE clipped relMSE vs NEE reference: unguided %.4f, trained-guided %.4f" % (ea, eb))
    assert eb < 0.9 * ea


def test_error_conventions_on_device(gpu_pkg):
    """Reference options outside the build are refused with VSPG_ESCOPE, malformed input with VSPG_EINVAL, and a
    buffer that was never requested cannot be read back -- nothing is silently ignored."""
    P = gpu_pkg
    lib = P.load()
    W, H = 32, 24
    scene = P.fog_box_scene(W, H)
    cfg = P.VspgRenderConfig(W, H, 1, 0, 0, 1, 0)
    h = C.c_void_p()

    def create(sc, prm):
        rc = lib.vspg_renderer_create(C.byref(sc), C.byref(prm), C.byref(cfg), C.byref(h))
        if rc == 0:
            lib.vspg_renderer_destroy(h)
        return rc

    prm = P.app_f_params()
    assert create(scene, prm) == 0
    prm.lightsampler = P.LIGHTSAMPLER_POWER   # (round 3: "power" and "bvh" serve any number of lights)
    s1 = P.fog_box_scene(W, H)
    s1.quads[0].Le[0] = s1.quads[0].Le[1] = s1.quads[0].Le[2] = 1.0
    assert create(s1, prm) == 0
    prm = P.app_f_params()
    prm.maxdepth = -1
    assert create(scene, prm) in (P.VSPG_EINVAL, 0)  # (negative depth is the reference's "no bounce" -- not an error there)
    # NanoVDBMedium emits through its temperature grid only (config 5 "explosion"): the path samples volume emission in the
    # delta-tracking routine, never under "resampling" -- accepted there (and without effect); under "nds": blackbody emission
    # (test_temperature_grid_emission_vs_oracle)
    from scenes import nvdb_scene, cloud_density
    dens = cloud_density(8)
    s2 = nvdb_scene(dens, (8, 8, 8), 0.5, 1.0, W=W, H=H)
    s2.medium.Le[:] = (1, 1, 1)
    assert create(s2, P.app_f_params()) == P.VSPG_EINVAL and b"temperature grid" in lib.vspg_last_error()
    s2.medium.Le[:] = (0, 0, 0)
    temp = (dens * 3000).astype(np.float32)
    s2.medium.temperature = temp.ctypes.data_as(C.POINTER(C.c_float))
    s2.medium.nvdb_le_scale, s2.medium.temperature_offset, s2.medium.temperature_scale = 1.0, 0.0, 1.0
    nds = P.app_f_params()
    nds.vspsamplingmethod = P.VSP_NDS
    assert create(s2, nds) == 0
    from scenes import grid_scene
    s2g = grid_scene(dens, (8, 8, 8), 0.5, 1.0, W=W, H=H)
    s2g.medium.temperature = temp.ctypes.data_as(C.POINTER(C.c_float))
    s2g.medium.temperature_scale = 1.0
    assert create(s2g, nds) == 0
    s2g.medium.Le[:] = (1, 1, 1)     # 'Both "Le" and "temperature" values were provided.' (media.cpp:307-308)
    assert create(s2g, nds) == P.VSPG_EINVAL and b"temperature" in lib.vspg_last_error()
    hot = P.Renderer(s2, P.app_f_params(), W, H)
    s2.medium.temperature = None
    cold = P.Renderer(s2, P.app_f_params(), W, H)
    hot.render_wave(0, 2); cold.render_wave(0, 2)
    assert np.array_equal(hot.film().view(np.uint32), cold.film().view(np.uint32))   # App. C #12: no volume emission under resampling
    hot.close(); cold.close()
    # emissive grid with a malformed Lescale grid
    from scenes import grid_scene
    s3 = grid_scene(dens, (8, 8, 8), 0.5, 1.0, W=W, H=H)
    s3.medium.Le[:] = (1, 1, 1)
    s3.medium.le_scale = dens.ctypes.data_as(C.POINTER(C.c_float))
    s3.medium.le_nx, s3.medium.le_ny, s3.medium.le_nz = 8, 0, 8
    assert create(s3, P.app_f_params()) == P.VSPG_EINVAL
    # buffers that were never requested
    r = P.Renderer(scene, P.app_f_params(), W, H)
    with pytest.raises(Exception):
        r.tr_buffer()
    r.close()


@pytest.mark.parametrize("guiding", [False, True])
def test_guided_russian_roulette_vs_oracle(gpu_pkg, guiding):
    """rrguiding: survival probability from throughput / image-space contribution estimate (own stand-in for OpenPGL's
    GuidedRussianRoulette), minRRDepth 1 -- device == oracle: contribution estimates after the buffer updates,
    replayed paths and film; with and without directional guiding on top."""
    P = gpu_pkg
    W, H = 64, 48
    scene = P.fog_box_scene(W, H)
    prm = P.default_params() if guiding else P.app_f_params()
    prm.rrguiding = 1
    prm.maxdepth = 8
    prm.guide_num_training_waves = 3
    g = P.Renderer(scene, prm, W, H, seed=7)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=7)
    if guiding:  # the same field on both sides (in-loop training sums floats in a different order)
        field = light_field_for(P)
        g.set_guiding_field(field, field); c.set_guiding_field(field, field)
    for w in range(3):   # the image-space buffer (and with it the contribution estimate) becomes ready after wave 0
        g.render_wave(w, w + 1); g.post_process_wave()
        c.render_wave(w, w + 1); c.post_process_wave()
    rng = np.random.default_rng(29)
    n = 20000
    pix = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(8, 4096, n).astype(np.int32)
    Lg, sg = g.trace_paths(pix, si)
    Lc, sc = c.trace_paths(pix, si)
    exact = np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1)
    print("rrguiding (guiding=%s) paths: same segments %.5f bit-identical %.5f" % (guiding, np.mean(sg == sc), exact.mean()))
    assert np.array_equal(sg, sc) and exact.mean() == 1.0
    fg, fc = g.film(), c.film()
    ig, ic = fg[..., :3] / fg[..., 3:4], fc[..., :3] / fc[..., 3:4]
    relmse = np.mean((ig - ic) ** 2 / (ic ** 2 + 1e-4))
    print("rrguiding film relMSE %.3e" % relmse)
    assert relmse <= 1e-8
    # the rule is really in force: the standard rule gives other path lengths on the same samples
    prm.rrguiding = 0
    prm.minrrdepth = 1
    g0 = P.Renderer(scene, prm, W, H, seed=7)
    if guiding:
        g0.set_guiding_field(field, field)
    for w in range(3):
        g0.render_wave(w, w + 1); g0.post_process_wave()
    L0, s0 = g0.trace_paths(pix, si)
    assert np.mean(s0 != sg) > 0.02
    g0.close(); g.close(); c.close()


def light_field_for(P):
    from scenes import light_field
    return light_field(P, n=4)


# ---------------------------------------------------------------------------------------------
# medium boundaries (round 4, row X3): MediumInterface + Material "interface" (:318, :399-404, :1196-1243,
# interaction.h:117-121), Shape "sphere" (shapes.h:107-330) -- the HIP path against the oracle, bit for bit
# ---------------------------------------------------------------------------------------------
def _boundary_scene(P, name, W, H):
    from scenes import add_quad, add_sphere, cloud_density, cloud_scene, empty_scene, interface_box, grid_scene
    if name == "absorbing-sphere-sky":      # camera in vacuum, homogeneous medium inside an interface sphere, uniform sky
        s = empty_scene(W, H, (0, 0, -4), (0, 0, 0), fov=35.0)
        s.medium.type = P.MEDIUM_HOMOGENEOUS
        s.medium.sigma_a[:] = (0.3, 0.5, 0.8)
        s.medium.sigma_s[:] = (1.2, 1.0, 0.7)
        s.medium.g = 0.4
        s.camera_outside_medium = 1
        add_sphere(s, (0.1, 0, 0), 1.0, material=P.MATERIAL_INTERFACE, iface=P.IFACE_INSIDE, scale=(1.2, 1.0, 0.9))
        add_quad(s, (-6, -1.4, -6), (0, 0, 12), (12, 0, 0), kd=(0.5, 0.4, 0.3))
        P.add_infinite_light(s, P.LIGHT_UNIFORM_INFINITE, (0.8, 0.9, 1.0))
        P.add_infinite_light(s, P.LIGHT_DISTANT, (3.0, 2.8, 2.5), (0.3, 0.9, -0.2))
        return s
    if name in ("grid-box-shell", "grid-sphere-shell", "nvdb-sphere-shell"):   # closed diffuse shell with a light, a bounded grid medium
        n = 16
        dens = cloud_density(n)
        s = empty_scene(W, H, (0, 0.1, -2.6), (0, 0, 0), fov=45.0)
        m = s.medium
        m.type = P.MEDIUM_NANOVDB if name.startswith("nvdb") else P.MEDIUM_GRID
        m.sigma_a[:] = (0.4, 0.3, 0.2)
        m.sigma_s[:] = (3.0, 3.2, 3.4)
        m.g = 0.3
        m.nx = m.ny = m.nz = n
        m.bounds_min[:] = (-0.7, -0.7, -0.7)
        m.bounds_max[:] = (0.7, 0.7, 0.7)
        m.density = dens.ctypes.data_as(C.POINTER(C.c_float))
        s._density_keepalive = dens
        if name.startswith("nvdb"):
            for k in range(3):
                m.index_min[k] = 0
                m.voxel_size[k] = 1.4 / n
                m.grid_origin[k] = -0.7
            m.majorant_scale = 1.0
        for p00, e1, e2 in [((-3, -3, -3), (0, 0, 6), (6, 0, 0)), ((-3, 3, -3), (6, 0, 0), (0, 0, 6)), ((-3, -3, 3), (0, 6, 0), (6, 0, 0)),
                            ((-3, -3, -3), (6, 0, 0), (0, 6, 0)), ((-3, -3, -3), (0, 6, 0), (0, 0, 6)), ((3, -3, -3), (0, 0, 6), (0, 6, 0))]:
            add_quad(s, p00, e1, e2, kd=(0.6, 0.55, 0.5))
        add_quad(s, (-0.8, 2.99, -0.8), (1.6, 0, 0), (0, 0, 1.6), le=(6, 6, 5), kd=(0, 0, 0))
        s.camera_outside_medium = 1
        if "box" in name:
            interface_box(s, (-0.7, -0.7, -0.7), (0.7, 0.7, 0.7))
        else:
            add_sphere(s, (0, 0, 0), 1.25, material=P.MATERIAL_INTERFACE, iface=P.IFACE_INSIDE)
        return s
    if name in ("cloud-scene", "cloud-scene-nvdb", "cloud-scene-box"):         # the reference's cloud-scene shape
        return cloud_scene(W, H, cloud_density(24), 24, nvdb=name.endswith("nvdb"), sphere=not name.endswith("box"))
    if name == "fog-diffuse-sphere":        # the fog box with a DIFFUSE sphere in it: BSDF frame, NEE and MIS from a sphere vertex
        s = P.fog_box_scene(W, H)
        add_sphere(s, (0.3, -0.5, 0.2), 0.4, kd=(0.7, 0.5, 0.3), scale=(1.0, 1.3, 0.8))
        return s
    if name == "fog-hollow":                # the fog box with a vacuum bubble: MediumInterface "" "fog" on an interface sphere
        s = P.fog_box_scene(W, H)
        add_sphere(s, (0.0, 0.0, 0.3), 0.5, material=P.MATERIAL_INTERFACE, iface=P.IFACE_OUTSIDE)
        return s
    if name == "tri-interface-box":         # the bounded grid behind an interface box made of TRIANGLES, one side with flipped winding + flag
        s = _boundary_scene(P, "grid-sphere-shell", W, H)
        s.n_spheres = 0
        lo, hi = np.float32([-0.7] * 3), np.float32([0.7] * 3)
        tris, flags = [], []
        for ax in range(3):
            for side in (0, 1):
                u, v = (ax + 1) % 3, (ax + 2) % 3
                c = [np.zeros(3, np.float32) for _ in range(4)]
                for k, (a, b) in enumerate([(0, 0), (1, 0), (1, 1), (0, 1)]):
                    c[k][ax] = hi[ax] if side else lo[ax]
                    c[k][u] = hi[u] if a else lo[u]
                    c[k][v] = hi[v] if b else lo[v]
                quad = [c[0], c[1], c[2], c[3]] if side else [c[0], c[3], c[2], c[1]]   # n = Normalize(Cross(p0 - p2, p1 - p2)): outward
                t0, t1 = [quad[0], quad[1], quad[2]], [quad[0], quad[2], quad[3]]
                fl = P.TRI_INTERFACE | (P.IFACE_INSIDE << P.TRI_IFACE_SHIFT)
                if ax == 2:     # this pair is wound the other way round and says so
                    t0, t1 = t0[::-1], t1[::-1]
                    fl |= P.TRI_FLIP_NORMAL
                tris += [t0, t1]
                flags += [fl, fl]
        P.set_triangles(s, np.array(tris, dtype=np.float32))
        fa = np.array(flags, dtype=np.int32)
        s.tri_flags = fa.ctypes.data_as(C.POINTER(C.c_int32))
        s._tri_keepalive.append(fa)
        return s
    if name == "open-fog-sky":              # :318: a medium that fills an open scene -- rays that escape are not sampled
        s = _boundary_scene(P, "absorbing-sphere-sky", W, H)
        s.n_spheres = 0
        s.camera_outside_medium = 0
        return s
    raise KeyError(name)


BOUNDARY_SCENES = ["absorbing-sphere-sky", "grid-box-shell", "grid-sphere-shell", "nvdb-sphere-shell", "cloud-scene", "cloud-scene-nvdb",
                   "cloud-scene-box", "fog-diffuse-sphere", "fog-hollow", "tri-interface-box", "open-fog-sky"]


@pytest.mark.parametrize("name", BOUNDARY_SCENES)
@pytest.mark.parametrize("options", ["app-f", "defaults", "nds"])
def test_medium_boundaries_vs_oracle(gpu_pkg, name, options):
    """Replayed paths bit-identical to the oracle's, the film of three waves (post-processed: the VSP buffer updates) equal to
    the oracle's film, equal counters -- on every kernel that takes the scene."""
    P = gpu_pkg
    W, H = 64, 48
    scene = _boundary_scene(P, name, W, H)
    if options == "app-f":
        prm = P.app_f_params()
    elif options == "defaults":          # the reference's default options (surface RIS + volume MIS guiding, secondary VSP) over a hand-made
        prm = P.default_params()         # field (scenes.light_field: stands in for a trained one; training has its own tests)
        prm.lightsampler = P.LIGHTSAMPLER_UNIFORM if name == "fog-hollow" else prm.lightsampler
    else:
        prm = P.app_f_params()
        prm.vspsamplingmethod = P.VSP_NDS
    if options == "nds" and name == "fog-diffuse-sphere":
        pytest.skip("one NDS case per medium kind is enough")
    names = set()
    rng = np.random.default_rng(11)
    n = 6000
    xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 64, n).astype(np.int32)
    import scenes
    field = scenes.light_field(P, n=2, bmin=(-3, -3, -3), bmax=(3, 3, 3), light=(0.0, 2.9, 0.0)) if options == "defaults" else None
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=3)
    if field:
        c.set_guiding_field(field, field)
    Lc, sc = c.trace_paths(xy, si)
    def on_kernel(kernel):
        g = P.Renderer(scene, prm, W, H, seed=3)
        if field:
            g.set_guiding_field(field, field)
        kn = g.kernel_name()
        if kn in names:
            g.close()
            return
        names.add(kn)
        Lg, sg = g.trace_paths(xy, si)
        same = np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1)
        assert np.array_equal(sg, sc), (kn, np.flatnonzero(sg != sc)[:5])
        assert same.all(), (kn, np.flatnonzero(~same)[:5], Lg[~same][:3], Lc[~same][:3])
        cc = oracle_lib.OracleRenderer(scene, prm, W, H, seed=3)
        if field:
            cc.set_guiding_field(field, field)
        for w in range(3):
            g.render_wave(w, w + 1)
            g.post_process_wave()
            cc.render_wave(w, w + 1)
            cc.post_process_wave()
        fg, fc = g.film(), cc.film()
        assert np.array_equal(fg[..., 3], fc[..., 3])
        assert np.allclose(fg[..., :3], fc[..., :3], rtol=2e-6, atol=1e-7), kn
        cg, co = g.counters(), cc.counters()
        assert cg == co, (kn, cg, co)
        g.close(); cc.close()
    for kernel in (None, "lane"):     # (the kernel is chosen per launch: the variable stays set while the renderer is used)
        if kernel:
            os.environ["VSPG_KERNEL"] = kernel
        try:
            on_kernel(kernel)
        finally:
            os.environ.pop("VSPG_KERNEL", None)
    # heterogeneous media: the wavefront pipeline AND the per-lane kernel; homogeneous media: the workgroup kernel's full-scene
    # instantiation AND the per-lane kernel (guided renders over a full scene: the per-lane kernel only)
    het = scene.medium.type != P.MEDIUM_HOMOGENEOUS
    assert len(names) == (2 if het or options != "defaults" else 1), names
    assert np.isfinite(Lc).all() and Lc.max() > 0
    if name.startswith("cloud-scene"):
        assert sc.max() >= 5      # paths that enter, scatter, leave and hit the ground: boundary crossings are iterations of the loop
    c.close()
    print(name, options, sorted(names))


# ---------------------------------------------------------------------------------------------
# blackbody emission of temperature grids under "nds" (media.h:333-341, :724-735; util/spectrum.h:83-94, :568-588)
# ---------------------------------------------------------------------------------------------
def test_device_blackbody_vs_reference_goldens_and_host_atanhf(gpu_pkg, libm_shim):
    """The kernels' SampleVisible wavelengths (the HOST's atanhf through csrc/vspg_libm.h) and BlackbodySpectrum(T).Sample: bit for bit
    the golden vectors generated from the reference's headers, the oracle on 30 000 random (u, T), and the running libm's atanhf on
    4e6 wavelength samples."""
    import json
    P = gpu_pkg
    from scenes import cloud_density, grid_scene
    g = P.Renderer(grid_scene(cloud_density(8), (8, 8, 8), 0.5, 1.0, W=8, H=8), P.app_f_params(), 8, 8)
    G = json.load(open(os.path.join(ROOT, "tests", "golden", "primitives.json")))["blackbody"]
    rows = np.array([[float.fromhex(t) for t in row] for row in G], dtype=np.float32)
    out = g.blackbody_batch(rows[:, 0], rows[:, 1])
    assert np.array_equal(out.view(np.uint32), rows[:, 2:8].view(np.uint32))
    rng = np.random.default_rng(17)
    n = 30000
    u = rng.random(n, dtype=np.float32)
    T = (100.0 + rng.random(n) ** 2 * 20000.0).astype(np.float32)
    dev = g.blackbody_batch(u, T)
    lib = oracle_lib.load()
    o6 = (C.c_float * 6)()
    ref = np.empty((n, 6), dtype=np.float32)
    for i in range(n):
        lib.oracle_blackbody(float(u[i]), float(T[i]), o6)
        ref[i] = o6[:]
    assert np.array_equal(dev.view(np.uint32), ref.view(np.uint32))
    n = 4_000_000
    u = rng.random(n, dtype=np.float32)
    lam = g.blackbody_batch(u, np.full(n, 3000.0, dtype=np.float32))[:, :3]
    fp = C.POINTER(C.c_float)
    for k in range(3):
        up = u + np.float32(k) / np.float32(3)
        up = np.where(up > 1, up - np.float32(1), up).astype(np.float32)
        x = (np.float32(0.85691062) - np.float32(1.82750197) * up).astype(np.float32)
        a = np.empty_like(x)
        libm_shim.libm_atanhf(n, x.ctypes.data_as(fp), a.ctypes.data_as(fp))
        want = (np.float32(538) - np.float32(138.888889) * a).astype(np.float32)
        assert np.array_equal(lam[:, k].view(np.uint32), want.view(np.uint32)), k
    g.close()


def _temperature_scene(P, kind, W, H):
    import scenes
    n = 24
    dens = scenes.cloud_density(n)
    if kind == "grid":
        scene = scenes.grid_scene(dens, (n, n, n), (0.3, 0.35, 0.4), (1.6, 1.4, 1.2), g=0.4, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)
    elif kind == "grid-lescale":
        scene = scenes.grid_scene(dens, (n, n, n), 0.4, 1.5, g=0.3, bmin=(-0.8, -0.8, -0.5), bmax=(0.8, 0.7, 0.9), W=W, H=H)   # grey coefficients
        rng = np.random.default_rng(31)
        le = np.clip(rng.random((6, 5, 7)).astype(np.float32) * 2 - 0.6, 0, None).astype(np.float32)  # some cells emit nothing
        scene.medium.le_scale = le.ctypes.data_as(C.POINTER(C.c_float))
        scene.medium.le_nz, scene.medium.le_ny, scene.medium.le_nx = le.shape
        scene._le_keepalive = le
    elif kind == "nvdb":
        scene = scenes.nvdb_scene(dens, (n, n, n), (0.25, 0.3, 0.35), (3.0, 2.6, 2.2), g=0.5, index_min=(-3, 2, 0), voxel=(0.066, 0.0625, 0.058),
                                  origin=(-0.6, -0.93, -0.5), density_offset=0.02, majorant_scale=1.25, W=W, H=H)
    else:                                       # the explosion in the shape of the reference's scenes: camera in vacuum, interface sphere
        scene = P.cloud_scene(W, H, n, nvdb=True)
        dens = None
    m = scene.medium
    nvox = m.nx * m.ny * m.nz
    rng = np.random.default_rng(9)
    base = dens if dens is not None and dens.size == nvox else rng.random(nvox).astype(np.float32)
    temp = (150.0 + 2600.0 * np.clip(base + 0.3 * rng.random(nvox).astype(np.float32), 0, 1.4)).astype(np.float32)   # some of it below the threshold
    m.temperature = temp.ctypes.data_as(C.POINTER(C.c_float))
    # (the cloud scene's albedo is 0.99 and a sun lights it: a strong LeScale, or its emission disappears in the total)
    m.temperature_offset, m.temperature_scale, m.nvdb_le_scale = 120.0, 1.3, (200.0 if kind == "nvdb-cloud-scene" else 0.6)
    scene._temp_keepalive = temp
    return scene


@pytest.mark.parametrize("kind", ["grid", "grid-lescale", "nvdb", "nvdb-cloud-scene"])
@pytest.mark.parametrize("options", ["app-f", "defaults"])
def test_temperature_grid_emission_vs_oracle(gpu_pkg, kind, options):
    """"vspsamplingmethod" "nds" over a medium with a temperature grid: the delta-tracking callback adds
    sigma_a * scale * BlackbodySpectrum(T').Sample(lambda) at every tentative collision (:895-906).  Replayed paths bit-identical to the
    oracle's, three post-processed waves' film and the counters equal to the oracle's -- on the wavefront pipeline and on the per-lane
    kernel; the emission is really there (the same render without the grid is darker); under "resampling" the grid changes nothing."""
    P = gpu_pkg
    W, H = 64, 48
    scene = _temperature_scene(P, kind, W, H)
    prm = P.app_f_params() if options == "app-f" else P.default_params()
    prm.vspsamplingmethod = P.VSP_NDS
    import scenes
    field = scenes.light_field(P, n=2, bmin=(-3, -3, -3), bmax=(3, 3, 3), light=(0.0, 2.9, 0.0)) if options == "defaults" else None
    rng = np.random.default_rng(12)
    n = 8000
    xy = np.stack([rng.integers(0, W, n), rng.integers(0, H, n)], axis=1).astype(np.int32)
    si = rng.integers(0, 256, n).astype(np.int32)
    c = oracle_lib.OracleRenderer(scene, prm, W, H, seed=5)
    if field:
        c.set_guiding_field(field, field)
    Lc, sc = c.trace_paths(xy, si)
    assert np.isfinite(Lc).all() and Lc.max() > 0
    names, films = set(), []
    def on_kernel():
        g = P.Renderer(scene, prm, W, H, seed=5)
        if field:
            g.set_guiding_field(field, field)
        kn = g.kernel_name()
        names.add(kn)
        Lg, sg = g.trace_paths(xy, si)
        same = np.all(Lg.view(np.uint32) == Lc.view(np.uint32), axis=1)
        assert np.array_equal(sg, sc), (kn, np.flatnonzero(sg != sc)[:5])
        assert same.all(), (kn, np.flatnonzero(~same)[:5], Lg[~same][:3], Lc[~same][:3])
        cc = oracle_lib.OracleRenderer(scene, prm, W, H, seed=5)
        if field:
            cc.set_guiding_field(field, field)
        for w in range(3):
            g.render_wave(w, w + 1)
            g.post_process_wave()
            cc.render_wave(w, w + 1)
            cc.post_process_wave()
        fg, fc = g.film(), cc.film()
        assert np.array_equal(fg[..., 3], fc[..., 3])
        assert np.allclose(fg[..., :3], fc[..., :3], rtol=2e-6, atol=1e-7), kn
        assert g.counters() == cc.counters(), kn
        films.append(fg)
        g.close(); cc.close()
    # ("wg": the workgroup kernel's pool record has no room for the wavelength sample -- a medium with a temperature grid is routed
    #  to the per-lane kernel instead, vspg_capi.hip: uses_wg_kernel; round 4 ran k_render_wave_wg with an uninitialised sample)
    for kernel in (None, "lane", "wg"):
        if kernel:
            os.environ["VSPG_KERNEL"] = kernel
        try:
            on_kernel()
        finally:
            os.environ.pop("VSPG_KERNEL", None)
    assert len(names) == 2 and not any("k_render_wave_wg" in k for k in names), names
    for f in films[1:]:
        assert np.array_equal(films[0].view(np.uint32), f.view(np.uint32))   # pipeline == per-lane kernel, bit for bit
    # without the temperature grid: darker
    keep = scene.medium.temperature
    scene.medium.temperature = None
    g0 = P.Renderer(scene, prm, W, H, seed=5)
    if field:
        g0.set_guiding_field(field, field)
    for w in range(3):
        g0.render_wave(w, w + 1)
        g0.post_process_wave()
    f0 = g0.film()
    g0.close()
    gain = (films[0][..., :3].sum() - f0[..., :3].sum()) / max(f0[..., :3].sum(), 1e-6)
    print(kind, options, sorted(names), "emission adds %.1f %%" % (100 * gain))
    assert gain > 0.02
    # "resampling": the grid is accepted and has no effect
    rs = P.app_f_params() if options == "app-f" else P.default_params()
    cold = P.Renderer(scene, rs, W, H, seed=5)
    scene.medium.temperature = keep
    hot = P.Renderer(scene, rs, W, H, seed=5)
    for r in (hot, cold):
        if field:
            r.set_guiding_field(field, field)
        r.render_wave(0, 2)
    assert np.array_equal(hot.film().view(np.uint32), cold.film().view(np.uint32))
    hot.close(); cold.close(); c.close()

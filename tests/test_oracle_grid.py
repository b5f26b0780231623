"""Oracle GridMedium layer (DDA majorant iterator, trilinear density lookup, majorant grid,
SampleT_maj / SampleT_maj_Resampling over it) vs the known answers the survey recorded from the
reference's own code (GridMedium case) and structural properties."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from conftest import load_package

fh = float.fromhex


from scenes import d3_density, grid_scene as _grid_scene


def grid_scene(P, density, n, sigma_a, sigma_s, **kw):
    return _grid_scene(density, n, sigma_a, sigma_s, **kw)


@pytest.fixture(scope="module")
def d3():
    P = load_package()
    dens = d3_density()
    scene = grid_scene(P, dens, (8, 8, 8), 0.5, 4.5)
    r = oracle_lib.OracleRenderer(scene, oracle_lib.app_f_params(), 16, 16)
    r._keep = dens
    yield P, r
    r.close()


def q(P, **kw):
    d = dict(o=(0.1, 0.2, -0.5), d=(0.3, 0.2, 1.0), tMax=2.0, u=0.37, rng_a=0.25, rng_b=0.75, vsp=0.6, channel=1, stop_after=0)
    d.update(kw)
    return P.VspgTmajQuery(P.f3(*d["o"]), P.f3(*d["d"]), d["tMax"], d["u"], d["rng_a"], d["rng_b"], d["vsp"], d["channel"], d["stop_after"])


def test_d3_grid_resampling_known_answer(d3):
    P, r = d3
    # SampleT_maj_Resampling(guide, vsp=0.6) => 8 callbacks, sum(sigma_t/sigma_maj)[1] = 0x1.dd496p+1,
    # T[1] = 0x1.8f239ep-2, vrc = 0x1.356952p-1, majorantScale = 1
    o = r.sample_tmaj_batch(P.TMAJ_RESAMPLING, [q(P)])[0]
    assert o.n_callbacks == 8
    assert o.sum_sigt_over_maj == fh("0x1.dd496p+1")
    assert o.T_maj[1] == fh("0x1.8f239ep-2")
    assert o.vrc == fh("0x1.356952p-1")
    assert o.majorant_scale == 1.0


def test_d3_grid_plain_known_answer(d3):
    P, r = d3
    # plain SampleT_maj stopping at the 3rd callback => last z = 0x1.717118p-2, returns 1
    o = r.sample_tmaj_batch(P.TMAJ_PLAIN, [q(P, vsp=-1.0, stop_after=3)])[0]
    assert o.n_callbacks == 3
    assert o.last_p[2] == fh("0x1.717118p-2")
    assert list(o.T_maj) == [1.0, 1.0, 1.0]


def test_grid_matches_homogeneous_when_density_is_one():
    # a constant-density grid over a box covering the ray: the majorant equals sigma_t everywhere,
    # no null collisions, and delta tracking statistics equal the homogeneous medium's
    P = load_package()
    dens = np.ones(4 * 4 * 4, dtype=np.float32)
    scene = grid_scene(P, dens, (4, 4, 4), 0.05, 0.45, bmin=(-2, -2, -2), bmax=(2, 2, 2))
    r = oracle_lib.OracleRenderer(scene, oracle_lib.app_f_params(), 16, 16)
    rng = np.random.default_rng(0)
    n = 40000
    qs = [q(P, o=(0, 0, 0), d=(0, 0, 1), tMax=1.5, u=float(rng.random()), rng_a=float(rng.random()), vsp=-1.0, channel=0, stop_after=1)
          for _ in range(n)]
    out = r.sample_tmaj_batch(P.TMAJ_PLAIN, qs)
    frac = np.mean([o.n_callbacks for o in out])
    assert abs(frac - (1 - np.exp(-0.75))) < 6e-3
    assert all(abs(o.sum_sigt_over_maj - 1.0) < 1e-6 for o in out if o.n_callbacks)
    r.close()


def test_grid_ray_missing_bounds_and_zero_density():
    P = load_package()
    dens = np.zeros(8, dtype=np.float32)
    scene = grid_scene(P, dens, (2, 2, 2), 1.0, 1.0)
    r = oracle_lib.OracleRenderer(scene, oracle_lib.app_f_params(), 16, 16)
    for variant in (P.TMAJ_PLAIN, P.TMAJ_OPTICAL_DEPTH, P.TMAJ_RESAMPLING):
        # misses the [0,1]^3 bounds entirely
        o = r.sample_tmaj_batch(variant, [q(P, o=(5, 5, 5), d=(0, 0, 1), tMax=3.0)])[0]
        assert o.n_callbacks == 0 and list(o.T_maj) == [1.0, 1.0, 1.0]
        # crosses the bounds but the medium is empty: majorant 0 everywhere
        o = r.sample_tmaj_batch(variant, [q(P, o=(0.5, 0.5, -1), d=(0, 0, 1), tMax=3.0)])[0]
        assert o.n_callbacks == 0 and list(o.T_maj) == [1.0, 1.0, 1.0]
    r.close()


def test_grid_render_is_unbiased_wrt_vsp_guiding():
    # heterogeneous medium: the resampling estimator (VSP-guided) and plain delta tracking
    # (vspguiding off) must converge to the same image mean
    P = load_package()
    rng = np.random.default_rng(5)
    n = 12
    dens = np.clip(rng.random(n ** 3).astype(np.float32) * 1.6 - 0.3, 0, None).astype(np.float32)
    W, H = 48, 36
    means = []
    for guided in (1, 0):
        scene = grid_scene(P, dens, (n, n, n), 0.1, 2.4, g=0.3, bmin=(-0.8, -0.8, -0.6), bmax=(0.8, 0.6, 0.9), W=W, H=H)
        prm = oracle_lib.app_f_params()
        prm.vspguiding = guided
        r = oracle_lib.OracleRenderer(scene, prm, W, H)
        for w in range(96):
            r.render_wave(w, w + 1)
            r.post_process_wave()
        f = r.film_f64()
        means.append((f[..., :3] / f[..., 3:4]).reshape(-1, 3).mean(0))
        c = r.counters()
        assert c["density_queries"] > c["volume_scatters"]  # null collisions happen
        r.close()
    assert np.allclose(means[0], means[1], rtol=0.02), means


# ---------------------------------------------------------------------------------------------
# NanoVDBMedium semantics over a dense copy of the grid (media.h:657-753, media.cpp:549-675)
# ---------------------------------------------------------------------------------------------
def _nvdb(P, dens, n, **kw):
    from scenes import nvdb_scene
    return nvdb_scene(dens, n, kw.pop("sigma_a", 0.5), kw.pop("sigma_s", 4.5), **kw)


def test_nvdb_majorants_bound_the_density_and_offset_scale_apply():
    """64^3 majorant grid (media.cpp:600-671): every tentative collision sees sigma_t(p) <= sigma_maj (the walk
    records sum sigma_t/sigma_maj over its callbacks); "densityoffset" adds to the sampled density and
    "majorantscale" multiplies the majorants."""
    P = load_package()
    from scenes import cloud_density
    n = (20, 20, 20)
    dens = cloud_density(20)
    rng = np.random.default_rng(2)
    for off, scale in ((0.0, 1.0), (0.25, 1.0), (0.0, 1.7)):
        scene = _nvdb(P, dens, n, voxel=(0.08, 0.08, 0.07), origin=(-0.8, -0.8, -0.5), density_offset=off, majorant_scale=scale)
        r = oracle_lib.OracleRenderer(scene, oracle_lib.app_f_params(), 16, 16)
        qs = []
        for i in range(3000):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            qs.append(q(P, o=tuple(rng.uniform(-1, 1, 3)), d=tuple(d), tMax=3.0, u=float(rng.random()), rng_a=float(rng.random()),
                        rng_b=float(rng.random()), vsp=-1.0, channel=int(rng.integers(0, 3)), stop_after=1))
        out = r.sample_tmaj_batch(P.TMAJ_PLAIN, qs)
        ratios = np.array([o.sum_sigt_over_maj for o in out if o.n_callbacks])
        assert len(ratios) > 500
        assert (ratios <= 1.0 + 1e-6).all(), ratios.max()       # conservative majorants
        if off > 0:
            assert ratios.min() > 0                              # the offset makes every collision real-ish
        r.close()


def test_nvdb_constant_grid_matches_homogeneous_statistics():
    P = load_package()
    dens = np.ones(6 * 6 * 6, dtype=np.float32)
    scene = _nvdb(P, dens, (6, 6, 6), sigma_a=0.05, sigma_s=0.45, index_min=(-3, -3, -3), voxel=(1.0, 1.0, 1.0), origin=(0.0, 0.0, 0.0))
    r = oracle_lib.OracleRenderer(scene, oracle_lib.app_f_params(), 16, 16)
    rng = np.random.default_rng(0)
    # rays inside the interior (index -2..1: all 8 trilinear corners are inside the bbox there)
    qs = [q(P, o=(0.2, -0.3, -1.0), d=(0, 0, 1), tMax=1.5, u=float(rng.random()), rng_a=float(rng.random()), vsp=-1.0, channel=0, stop_after=1)
          for _ in range(40000)]
    out = r.sample_tmaj_batch(P.TMAJ_PLAIN, qs)
    frac = np.mean([o.n_callbacks for o in out])
    assert abs(frac - (1 - np.exp(-0.75))) < 6e-3
    assert all(abs(o.sum_sigt_over_maj - 1.0) < 1e-6 for o in out if o.n_callbacks)
    r.close()


def test_nvdb_trilinear_is_index_space_with_zero_background():
    """One voxel of density 1 at index (2,3,1): the sampled density is the tent a+w(b-a) around that index and 0
    beyond one voxel; checked through sigma_t/sigma_maj of a forced first collision (majorant_scale 1)."""
    P = load_package()
    n = (5, 6, 4)
    dens = np.zeros(n[0] * n[1] * n[2], dtype=np.float32)
    dens[(1 * n[1] + 3) * n[0] + 2] = 1.0
    vox, org = (0.5, 0.25, 0.2), (-1.0, -0.5, 0.1)
    scene = _nvdb(P, dens, n, sigma_a=1.0, sigma_s=1.0, voxel=vox, origin=org)
    r = oracle_lib.OracleRenderer(scene, oracle_lib.app_f_params(), 16, 16)
    # a ray along +x through index (x, 3, 1): density at continuous index x is max(0, 1 - |x - 2|)
    y, z = org[1] + 3 * vox[1], org[2] + 1 * vox[2]
    seen = 0
    rng = np.random.default_rng(1)
    for _ in range(400):
        o = r.sample_tmaj_batch(P.TMAJ_PLAIN, [q(P, o=(org[0] + 0.01, y, z), d=(1, 0, 0), tMax=2.4, u=float(rng.random()),
                                                 rng_a=float(rng.random()), vsp=-1.0, channel=0, stop_after=1)])[0]
        if not o.n_callbacks:
            continue
        xi = (o.last_p[0] - org[0]) / vox[0]
        expect = max(0.0, 1.0 - abs(xi - 2.0))
        # the cell's majorant is 1 (the voxel lies in its slop) wherever the tent is non-zero
        assert abs(o.sum_sigt_over_maj - expect) < 2e-5, (xi, o.sum_sigt_over_maj, expect)
        seen += 1
    assert seen > 50
    r.close()


# ---------------------------------------------------------------------------------------------
# TrBuffer + NDS+ (cpu/trbuffer.h, guidedvolpathvspgintegrator.cpp:727-728, 929-938, 975-976, 1072-1073)
# ---------------------------------------------------------------------------------------------
def _tr_scene(P, W, H, seed=11, n=10):
    rng = np.random.default_rng(seed)
    dens = np.clip(rng.random(n ** 3).astype(np.float32) * 1.5 - 0.25, 0, None).astype(np.float32)
    # thin: NDS (hence NDS+) only acts where the wanted scatter probability exceeds 1 - exp(-tau_maj)
    return grid_scene(P, dens, (n, n, n), 0.02, 0.33, g=0.2, bmin=(-0.8, -0.8, -0.6), bmax=(0.8, 0.6, 0.9), W=W, H=H), dens


def test_tr_buffer_is_the_running_mean_of_the_primary_transmittance_estimates():
    P = load_package()
    W, H = 24, 18
    scene, dens = _tr_scene(P, W, H)
    prm = oracle_lib.app_f_params()
    prm.vspsamplingmethod = P.VSP_RESAMPLING
    prm.storeTrBuffer = 1
    per_wave = []
    for w in range(3):  # wave w alone: the estimate of sample w (the VSP buffer is not ready before wave 1's update)
        r = oracle_lib.OracleRenderer(scene, prm, W, H)
        r.render_wave(w, w + 1)
        per_wave.append(r.tr_buffer())
        r.close()
    r = oracle_lib.OracleRenderer(scene, prm, W, H)
    r.render_wave(0, 3)
    tr = r.tr_buffer()
    r.close()
    assert np.all(tr >= 0) and np.all(tr <= 1) and 0.05 < tr.mean() < 0.95
    assert np.allclose(tr, np.mean(per_wave, axis=0), atol=2e-6)
    # without the flag (and without NDS+) no buffer is kept
    prm.storeTrBuffer = 0
    r = oracle_lib.OracleRenderer(scene, prm, W, H)
    with pytest.raises(AssertionError):
        r.tr_buffer()
    r.close()


def test_nds_plus_is_unbiased_and_changes_the_collision_decisions():
    """NDS+ biases the primary ray's real/null-collision probability by the cached transmittance and compensates
    in r_u: the image mean must stay that of plain NDS, while individual paths differ."""
    P = load_package()
    W, H = 40, 30
    scene, dens = _tr_scene(P, W, H, seed=12)
    # pass 1 (reference workflow): resampling + storeTrBuffer
    prm = oracle_lib.app_f_params()
    prm.vspsamplingmethod = P.VSP_RESAMPLING
    prm.storeTrBuffer = 1
    r = oracle_lib.OracleRenderer(scene, prm, W, H)
    r.render_wave(0, 16)
    tr = r.tr_buffer()
    r.close()
    means, films = [], []
    for bias in (0, 1):
        prm = oracle_lib.app_f_params()
        prm.vspsamplingmethod = P.VSP_NDS
        prm.collisionProbabilityBias = bias
        r = oracle_lib.OracleRenderer(scene, prm, W, H)
        if bias:
            r.set_tr_buffer(tr)
        r.render_wave(0, 160)  # (no PostProcessWave: the primary VSP stays 0.5, where NDS acts on most rays of this scene)
        f = r.film_f64()
        films.append(f[..., :3] / f[..., 3:4])
        means.append(films[-1].reshape(-1, 3).mean(0))
        r.close()
    assert np.allclose(means[0], means[1], rtol=0.03), means
    assert np.mean(np.abs(films[0] - films[1]) > 1e-6) > 0.5  # not the same paths
    # requested but no buffer handed over: the renderer records (nothing, under NDS) and behaves like plain NDS
    prm = oracle_lib.app_f_params()
    prm.vspsamplingmethod = P.VSP_NDS
    prm.collisionProbabilityBias = 1
    a = oracle_lib.OracleRenderer(scene, prm, W, H)
    prm.collisionProbabilityBias = 0
    b = oracle_lib.OracleRenderer(scene, prm, W, H)
    a.render_wave(0, 2); b.render_wave(0, 2)
    assert np.array_equal(a.film_f64(), b.film_f64())
    assert np.all(a.tr_buffer() == 0)
    a.close(); b.close()


# ---------------------------------------------------------------------------------------------
# emissive GridMedium (media.h:326-342, media.cpp:316-328): Le = LeScale.Lookup(p) * Le_spec where positive
# ---------------------------------------------------------------------------------------------
def _emissive_scene(P, W, H, le_grid, n=10, seed=21):
    rng = np.random.default_rng(seed)
    dens = np.clip(rng.random(n ** 3).astype(np.float32) * 1.4 - 0.2, 0, None).astype(np.float32)
    scene = grid_scene(P, dens, (n, n, n), 0.4, 1.2, g=0.1, bmin=(-0.8, -0.8, -0.6), bmax=(0.8, 0.6, 0.9), W=W, H=H)
    scene.medium.Le[:] = (3.0, 1.5, 0.5)
    if le_grid is not None:
        g = np.ascontiguousarray(le_grid, dtype=np.float32)
        scene.medium.le_scale = g.ctypes.data_as(C.POINTER(C.c_float))
        scene.medium.le_nz, scene.medium.le_ny, scene.medium.le_nx = g.shape
        scene._le_keepalive = g
    return scene, dens


def _render_mean(P, scene, W, H, spp, **prm_kw):
    prm = oracle_lib.app_f_params()
    for k, v in prm_kw.items():
        setattr(prm, k, v)
    r = oracle_lib.OracleRenderer(scene, prm, W, H)
    r.render_wave(0, spp)
    f = r.film_f64()
    r.close()
    return (f[..., :3] / f[..., 3:4])


def test_emissive_grid_adds_radiance_in_the_delta_tracking_routine_only():
    """Volume emission is sampled by the delta-tracking callback (:895-906); the resampling routine never sees it
    (SURVEY App. C #12).  With all surfaces black and the light off, the image is the medium's own emission."""
    P = load_package()
    W, H = 24, 18
    le = np.zeros((4, 4, 4), dtype=np.float32)
    le[1:3, 1:3, 1:3] = 2.0
    scene, dens = _emissive_scene(P, W, H, le)
    for i in range(scene.n_quads):
        scene.quads[i].Kd[:] = (0, 0, 0)
        scene.quads[i].Le[:] = (0, 0, 0)
    img_nds = _render_mean(P, scene, W, H, 64, vspsamplingmethod=P.VSP_NDS)
    assert img_nds.mean() > 1e-2 and np.all(img_nds >= 0)
    # emission colour: the ratio of the channels follows Le_spec where absorption is grey
    m = img_nds.reshape(-1, 3).mean(0)
    assert abs(m[0] / m[1] - 2.0) < 0.1 and abs(m[1] / m[2] - 3.0) < 0.2
    # twice the LeScale grid -> twice the radiance, sample by sample (same random walk)
    scene2, _ = _emissive_scene(P, W, H, 2 * le)
    for i in range(scene2.n_quads):
        scene2.quads[i].Kd[:] = (0, 0, 0)
        scene2.quads[i].Le[:] = (0, 0, 0)
    img2 = _render_mean(P, scene2, W, H, 64, vspsamplingmethod=P.VSP_NDS)
    assert np.allclose(img2, 2 * img_nds, rtol=1e-5, atol=1e-7)
    # the resampling routine does not sample volume emission
    img_rs = _render_mean(P, scene, W, H, 16, vspsamplingmethod=P.VSP_RESAMPLING)
    assert img_rs.max() == 0
    # no "Lescale": the reference's 1x1x1 grid holding 1 -- a tent over the bounds, positive inside
    scene3, _ = _emissive_scene(P, W, H, None)
    for i in range(scene3.n_quads):
        scene3.quads[i].Kd[:] = (0, 0, 0)
        scene3.quads[i].Le[:] = (0, 0, 0)
    assert _render_mean(P, scene3, W, H, 32, vspsamplingmethod=P.VSP_NDS).mean() > 1e-2
    # NanoVDBMedium emits through a temperature grid only: refused
    from scenes import nvdb_scene
    s4 = nvdb_scene(dens, (10, 10, 10), 0.4, 1.2, W=W, H=H)
    s4.medium.Le[:] = (1, 1, 1)
    with pytest.raises(Exception):
        oracle_lib.OracleRenderer(s4, oracle_lib.app_f_params(), W, H)

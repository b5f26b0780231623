"""Oracle free-flight layer (SampleT_maj family) vs the known answers recorded in SURVEY.md
App. D.3 -- outputs of the reference's own media_sampleTMaj.h code -- and analytic answers."""
import numpy as np
import pytest

import oracle_lib
from conftest import load_package

fh = float.fromhex


@pytest.fixture(scope="module")
def orc():
    scene = oracle_lib.fog_box_scene(16, 16)  # homogeneous sigma_a=.05 sigma_s=.45
    r = oracle_lib.OracleRenderer(scene, oracle_lib.app_f_params(), 16, 16)
    yield r
    r.close()


def q(P, **kw):
    d = dict(o=(0, 0, 0), d=(0, 0, 1), tMax=3.0, u=0.37, rng_a=0.25, rng_b=0.75, vsp=0.8, channel=0, stop_after=0)
    d.update(kw)
    return P.VspgTmajQuery(P.f3(*d["o"]), P.f3(*d["d"]), d["tMax"], d["u"], d["rng_a"], d["rng_b"], d["vsp"],
                           d["channel"], d["stop_after"])


def test_d3_homogeneous_optical_depth_space(orc):
    P = load_package()
    # App. D.3: one callback at t = 0x1.4498p+1, r_u_factor = 0x1.03d048p+0 in all channels,
    # returned T_maj = 1, for hero channels 0,1,1,2 with RNG(Hash(0.25f+k), Hash(0.75f)).
    for k, ch in enumerate((0, 1, 1, 2)):
        out = orc.sample_tmaj_batch(P.TMAJ_OPTICAL_DEPTH, [q(P, rng_a=0.25 + k, channel=ch, stop_after=1)])[0]
        assert out.n_callbacks == 1
        assert out.last_t == fh("0x1.4498p+1")
        assert list(out.r_u_factor) == [fh("0x1.03d048p+0")] * 3
        assert list(out.T_maj) == [1.0, 1.0, 1.0]
    # analytic: u' = 0.74 < vsp -> t = -log(1 - 0.74 (1-e^{-1.5})/0.8) / 0.5
    t = -np.log(1 - 0.74 * (1 - np.exp(-1.5)) / 0.8) / 0.5
    assert abs(out.last_t - t) < 2e-3  # FastExp is a 3e-4 approximation


def test_statistical_known_answer(orc):
    # App. D.3: sigma_t=.5, albedo .9, tMax~U(0.5,4.5), vsp=.7, alpha=.5:
    # scatter fraction 0.6816 (analytic), mean weight E[beta/r_u] 0.93370 (analytic)
    P = load_package()
    rng = np.random.default_rng(1)
    n = 200000
    tmax = rng.uniform(0.5, 4.5, n).astype(np.float32)
    us = rng.random(n).astype(np.float32)
    ra = rng.random(n).astype(np.float32)
    qs = [q(P, tMax=float(tmax[i]), u=float(us[i]), rng_a=float(ra[i]), vsp=0.7, channel=i % 3, stop_after=1)
          for i in range(n)]
    out = orc.sample_tmaj_batch(P.TMAJ_OPTICAL_DEPTH, qs)
    scat = np.array([o.n_callbacks for o in out], dtype=np.float64)
    ruf = np.array([o.r_u_factor[0] for o in out], dtype=np.float64)
    assert set(np.unique(scat)) <= {0.0, 1.0}
    w = np.where(scat > 0, 0.9, 1.0) / ruf
    assert abs(scat.mean() - 0.6816) < 4e-3
    assert abs(w.mean() - 0.93370) < 3e-3
    # pass-through must return T_maj/T_maj[ch] == 1 for a grey medium
    for o in out[:2000]:
        if o.n_callbacks == 0:
            assert abs(o.T_maj[0] / o.T_maj[1] - 1) < 1e-6


def test_plain_delta_tracking_is_exponential(orc):
    # unguided (vsp<0): OpticalDepthSpace falls back to SampleT_maj (media_sampleTMaj.h:287-288);
    # P(collision before tMax) = 1 - exp(-sigma_t tMax)
    P = load_package()
    rng = np.random.default_rng(2)
    n = 100000
    us = rng.random(n).astype(np.float32)
    for variant in (P.TMAJ_PLAIN, P.TMAJ_OPTICAL_DEPTH):
        qs = [q(P, tMax=2.0, u=float(us[i]), vsp=-1.0, stop_after=1) for i in range(n)]
        out = orc.sample_tmaj_batch(variant, qs)
        frac = np.mean([o.n_callbacks for o in out])
        assert abs(frac - (1 - np.exp(-1.0))) < 5e-3
        ts = np.array([o.last_t for o in out if o.n_callbacks])
        assert ts.min() >= 0 and ts.max() < 2.0
        # first collision distance matches -log(1-u)/sigma_t
        o0 = out[0]
        if o0.n_callbacks:
            assert abs(o0.last_t + np.log(1 - us[0]) / 0.5) < 1e-5


def test_unnormalised_direction_scales_tmax(orc):
    # SampleT_maj normalises d and multiplies tMax by |d| (media_sampleTMaj.h:55-56)
    P = load_package()
    a = orc.sample_tmaj_batch(P.TMAJ_OPTICAL_DEPTH, [q(P, d=(0, 0, 2), tMax=1.5, stop_after=1)])[0]
    b = orc.sample_tmaj_batch(P.TMAJ_OPTICAL_DEPTH, [q(P, d=(0, 0, 1), tMax=3.0, stop_after=1)])[0]
    assert a.last_t == b.last_t and list(a.r_u_factor) == list(b.r_u_factor)


def test_edge_cases(orc):
    P = load_package()
    # tMax == 0: t_v == 0 -> returns 1, no callback (media_sampleTMaj.h:316-318)
    o = orc.sample_tmaj_batch(P.TMAJ_OPTICAL_DEPTH, [q(P, tMax=0.0)])[0]
    assert o.n_callbacks == 0 and list(o.T_maj) == [1.0, 1.0, 1.0]
    # vsp clamps to [0.001, 0.999] (guidedvolpathvspgintegrator.cpp:670-671)
    for vsp in (0.0, 1.0):
        o = orc.sample_tmaj_batch(P.TMAJ_OPTICAL_DEPTH, [q(P, vsp=vsp, stop_after=1)])[0]
        assert np.isfinite(o.r_u_factor[0]) and o.r_u_factor[0] > 0
    # resampling on a homogeneous medium: majorant scale kicks in when vsp demands more
    # optical depth than the segment has (media_sampleTMaj.h:171-181)
    o = orc.sample_tmaj_batch(P.TMAJ_RESAMPLING, [q(P, tMax=0.1, vsp=0.9)])[0]
    tau_min = -np.log(1 - np.float32(0.9))
    assert abs(o.majorant_scale - tau_min / 0.05) < 1e-3 * o.majorant_scale
    assert abs(o.vrc - 0.9 / (1 - np.exp(-tau_min))) < 2e-3


def test_loaded_vsp_buffer_is_used_as_is_and_never_updated():
    """loadISGBuffer (guidedvolpathvspgintegrator.cpp:151-159, 251-256): the buffer is ready from the first wave on and
    PostProcessWave leaves it alone; an in-loop buffer of the same scene is estimated at waves 1, 2, 4, ..."""
    import oracle_lib
    from conftest import load_package
    P = load_package()
    W, H = 24, 16
    scene = oracle_lib.fog_box_scene(W, H)
    prm = oracle_lib.app_f_params()
    rng = np.random.default_rng(3)
    vsp = rng.uniform(0.1, 0.9, (H, W)).astype(np.float32)
    vsp[0, 0] = -1.0  # "no estimate" for this pixel: unguided there (:1101-1112)
    a = oracle_lib.OracleRenderer(scene, prm, W, H)
    a.load_vsp_buffer(vsp)
    b = oracle_lib.OracleRenderer(scene, prm, W, H)
    for w in range(4):
        a.render_wave(w, w + 1); a.post_process_wave()
        b.render_wave(w, w + 1); b.post_process_wave()
    va, ra = a.vsp_buffer()
    vb, rb = b.vsp_buffer()
    assert ra and rb and np.array_equal(va, vsp) and not np.array_equal(vb, vsp)
    assert not np.array_equal(a.film_f64(), b.film_f64())  # wave 0 already used the loaded estimates instead of 0.5
    a.close(); b.close()

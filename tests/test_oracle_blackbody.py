"""Blackbody volume emission of temperature grids (GridMedium media.h:333-341, NanoVDBMedium media.h:724-735; BlackbodySpectrum,
util/spectrum.h:568-588) as the delta-tracking callback adds it (guidedvolpathvspgintegrator.cpp:895-906) -- the CPU restatement.

The spectral functions themselves are pinned bit for bit (tests/test_oracle_primitives.py::test_blackbody_emission_against_the_reference).
Here: how the path uses them.  In the RGB build the three channels hold Planck's law at the path's three SAMPLED wavelengths
(SURVEY App. C #13), so a pixel is not a colour but a per-path product -- which is what the known answers below test."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from oracle_lib import load_package
from scenes import grid_scene, nvdb_scene


def _shell_density(n, seed=3):
    """random density, zero in the two outermost voxel layers: wherever sigma_a > 0 the trilinear footprint of a CONSTANT temperature
    (or LeScale) grid lies inside the grid, i.e. the interpolated value is that constant"""
    rng = np.random.default_rng(seed)
    d = (0.3 + rng.random((n, n, n))).astype(np.float32)
    d[:2] = d[-2:] = 0
    d[:, :2] = d[:, -2:] = 0
    d[:, :, :2] = d[:, :, -2:] = 0
    return np.ascontiguousarray(d.reshape(-1))


def _scene(kind, dens, n, W, H, sigma_a=0.9, sigma_s=0.6):
    if kind == "grid":
        s = grid_scene(dens, (n, n, n), sigma_a, sigma_s, g=0.2, bmin=(-0.8, -0.8, -0.6), bmax=(0.8, 0.6, 0.9), W=W, H=H)
    else:
        s = nvdb_scene(dens, (n, n, n), sigma_a, sigma_s, g=0.2, index_min=(-2, 1, 0), voxel=(0.13, 0.12, 0.125), origin=(-0.55, -0.9, -0.6), W=W, H=H)
    for i in range(s.n_quads):   # nothing but the medium's own emission
        s.quads[i].Kd[:] = (0, 0, 0)
        s.quads[i].Le[:] = (0, 0, 0)
    return s


def _set_lescale_grid(s, value, n):
    """an "Lescale" grid of the density grid's size holding one value (GridMedium only; absent, the reference builds the 1x1x1 grid
    {1}, whose lookup is a tent over the bounds: media.cpp:319-320)"""
    g = np.full(n ** 3, value, dtype=np.float32)
    s.medium.le_scale = g.ctypes.data_as(C.POINTER(C.c_float))
    s.medium.le_nx = s.medium.le_ny = s.medium.le_nz = n
    s._le_keepalive = g


def _set_temperature(s, temp, offset=0.0, scale=1.0, le_scale=1.0):
    t = np.ascontiguousarray(temp, dtype=np.float32)
    s.medium.temperature = t.ctypes.data_as(C.POINTER(C.c_float))
    s.medium.temperature_offset, s.medium.temperature_scale, s.medium.nvdb_le_scale = offset, scale, le_scale
    s._temp_keepalive = t


def _nds():
    prm = oracle_lib.app_f_params()
    prm.vspsamplingmethod = load_package().VSP_NDS
    return prm


def _paths(scene, prm, W, H, pix, si, seed=4):
    r = oracle_lib.OracleRenderer(scene, prm, W, H, seed=seed)
    L, seg = r.trace_paths(pix, si)
    r.close()
    return L, seg


def _lu(pix, si, seed=4):
    """the wavelength sample of each path: the first number of its pixel sample (IndependentSampler, cpu/integrators.cpp:272-280)"""
    lib = oracle_lib.load()
    out = (C.c_float * 1)()
    lu = np.empty(len(si), dtype=np.float32)
    for k, ((x, y), s) in enumerate(zip(pix, si)):
        lib.oracle_independent_sampler(int(x), int(y), seed, int(s), 1, out)
        lu[k] = out[0]
    return lu


def _bb(lu, T):
    lib = oracle_lib.load()
    out = (C.c_float * 6)()
    r = np.empty((len(lu), 3), dtype=np.float64)
    for k, u in enumerate(lu):
        lib.oracle_blackbody(float(u), float(T), out)
        r[k] = out[3:6]
    return r


@pytest.mark.parametrize("kind", ["grid", "nvdb"])
def test_uniform_temperature_factorises_per_path(kind):
    """Constant temperature T wherever the medium absorbs: every emission term of a path carries the same
    Le = scale * BlackbodySpectrum(T).Sample(lambda(lu)), so a path's radiance is X (.) BB(T, lambda(lu)) with X independent of T.
    Two temperatures give the same X path by path; for the GridMedium X is also what an Le_spec = (1,1,1) medium over the same
    random walk returns (media.h:341 instead of :333-340).  Grey coefficients: the random walk does not depend on the channel."""
    P = load_package()
    W, H, n = 20, 16, 12
    dens = _shell_density(n)
    rng = np.random.default_rng(1)
    m = 600
    pix = np.stack([rng.integers(0, W, m), rng.integers(0, H, m)], axis=1).astype(np.int32)
    si = rng.integers(0, 512, m).astype(np.int32)
    lu = _lu(pix, si)
    X = []
    for T in (1500.0, 4200.0):
        s = _scene(kind, dens, n, W, H)
        _set_temperature(s, np.full(n ** 3, T), le_scale=1.75)
        if kind == "grid":
            _set_lescale_grid(s, 1.75, n)
        L, seg = _paths(s, _nds(), W, H, pix, si)
        bb = _bb(lu, T) * 1.75
        lit = bb.min(axis=1) > 1e-6
        assert lit.sum() > 0.9 * m and np.all(np.isfinite(L)) and np.all(L >= 0)
        X.append((L[lit] / bb[lit], lit, seg))
    both = X[0][1] & X[1][1]
    a = np.full((m, 3), np.nan); a[X[0][1]] = X[0][0]
    b = np.full((m, 3), np.nan); b[X[1][1]] = X[1][0]
    assert np.array_equal(X[0][2], X[1][2])            # same random walk
    assert (a[both] > 0).mean() > 0.5                  # most paths see emission
    assert np.allclose(a[both], b[both], rtol=2e-5, atol=1e-9)
    if kind == "grid":
        s = _scene(kind, dens, n, W, H)
        s.medium.Le[:] = (1, 1, 1)
        _set_lescale_grid(s, 1.0, n)
        L1, seg1 = _paths(s, _nds(), W, H, pix, si)
        assert np.array_equal(seg1, X[0][2])
        assert np.allclose(a[both], L1[both], rtol=2e-5, atol=1e-9)
        # no "Lescale": the reference's 1x1x1 grid {1}, whose lookup is a tent over the bounds -- dimmer than the constant 1 everywhere
        s = _scene(kind, dens, n, W, H)
        _set_temperature(s, np.full(n ** 3, 1500.0))
        Lt, _ = _paths(s, _nds(), W, H, pix, si)
        tent = Lt[both] / _bb(lu, 1500.0)[both]
        assert Lt.sum() > 0 and np.all(tent <= a[both] * (1 + 1e-5)) and tent.sum() < 0.8 * a[both].sum()


@pytest.mark.parametrize("kind", ["grid", "nvdb"])
def test_temperature_threshold_offset_and_scale(kind):
    """temp = (T(p) - temperatureoffset) * temperaturescale; no emission unless temp > 100 (media.h:337-339 / :731-733)"""
    W, H, n = 12, 10, 10
    dens = _shell_density(n, seed=8)
    rng = np.random.default_rng(2)
    pix = np.stack([rng.integers(0, W, 200), rng.integers(0, H, 200)], axis=1).astype(np.int32)
    si = rng.integers(0, 64, 200).astype(np.int32)

    def run(T, offset, scale):
        s = _scene(kind, dens, n, W, H)
        _set_temperature(s, np.full(n ** 3, T), offset, scale)
        return _paths(s, _nds(), W, H, pix, si)[0]

    assert run(600.0, 400.0, 0.5).max() == 0           # exactly 100: not above
    hot = run(2600.0, 400.0, 0.5)                      # 1100 K
    assert hot.max() > 0
    assert np.array_equal(hot, run(1100.0, 0.0, 1.0))  # the same scaled temperature, the same floats
    assert run(90.0, 0.0, 1.0).max() == 0


@pytest.mark.parametrize("kind", ["grid", "nvdb"])
def test_varying_temperature_renders_and_resampling_never_sees_it(kind):
    P = load_package()
    W, H, n = 16, 12, 12
    rng = np.random.default_rng(5)
    dens = np.clip(rng.random(n ** 3).astype(np.float32) * 1.4 - 0.2, 0, None).astype(np.float32)
    temp = (200.0 + 3000.0 * rng.random(n ** 3)).astype(np.float32)
    s = _scene(kind, dens, n, W, H, sigma_a=(0.5, 0.7, 0.9), sigma_s=(1.2, 1.0, 0.8))
    _set_temperature(s, temp, 50.0, 1.2, 0.8)
    r = oracle_lib.OracleRenderer(s, _nds(), W, H)
    r.render_wave(0, 16)
    f = r.film_f64()
    r.close()
    img = f[..., :3] / f[..., 3:4]
    assert np.all(np.isfinite(img)) and np.all(img >= 0) and img.mean() > 1e-3
    # under "resampling" a heterogeneous medium's emission is never sampled
    r = oracle_lib.OracleRenderer(s, oracle_lib.app_f_params(), W, H)
    r.render_wave(0, 4)
    assert r.film_f64()[..., :3].max() == 0
    r.close()


def test_le_and_temperature_together_are_refused():
    """GridMedium::Create: 'Both "Le" and "temperature" values were provided.' (media.cpp:307-308)"""
    n = 6
    dens = _shell_density(n)
    s = _scene("grid", dens, n, 8, 8)
    _set_temperature(s, np.full(n ** 3, 1500.0))
    s.medium.Le[:] = (1, 0.5, 0.25)
    with pytest.raises(Exception):
        oracle_lib.OracleRenderer(s, _nds(), 8, 8)

"""Guiding-cache query in the oracle.  The OpenPGL side (kd-tree lookup,
parallax-aware vMF mixtures, products, VSP) is this build's own design -- parity unpinned -- so the
checks are mathematical: the mixture PDFs integrate to one, SamplePDF draws from PDF, VSP stays in
[0,1], and rendering with guided BSDF / phase sampling (MIS and RIS) and secondary-ray VSP guiding
converges to the same image as the unguided integrator (the guiding.h wrapper logic is restated
from the reference; any bias there would show here)."""
import numpy as np
import pytest

import oracle_lib
import scenes
from conftest import load_package


def sphere_quadrature(n_theta=400, n_phi=800):
    ct = (np.arange(n_theta) + 0.5) / n_theta * 2 - 1
    ph = (np.arange(n_phi) + 0.5) / n_phi * 2 * np.pi
    CT, PH = np.meshgrid(ct, ph, indexing="ij")
    ST = np.sqrt(1 - CT ** 2)
    dirs = np.stack([ST * np.cos(PH), ST * np.sin(PH), CT], axis=-1).reshape(-1, 3)
    return dirs.astype(np.float32), 4 * np.pi / dirs.shape[0]


@pytest.fixture(scope="module")
def orc():
    P = load_package()
    scene = oracle_lib.fog_box_scene(32, 32)
    prm = oracle_lib.default_params()
    r = oracle_lib.OracleRenderer(scene, prm, 32, 32)
    field = scenes.light_field(P, n=4)
    r.set_guiding_field(field, field)
    r._field = field
    yield P, r
    r.close()


@pytest.mark.parametrize("is_volume,g", [(0, 0.0), (1, 0.0), (1, 0.7), (1, -0.5)])
def test_mixture_pdfs_are_normalised_and_vsp_in_range(orc, is_volume, g):
    P, r = orc
    dirs, dw = sphere_quadrature()
    n = dirs.shape[0]
    for p, a in (((0.1, -0.3, 0.2), (0, 1, 0)), ((-0.8, 0.7, -0.6), (0.6, 0, 0.8)), ((0.5, -0.99, 0.5), (0, 1, 0))):
        out = r.guiding_query_batch(is_volume, g, np.tile(p, (n, 1)), np.tile(a, (n, 1)), dirs, np.zeros((n, 2)))
        assert out["ok"].all()
        assert abs(out["pdf"].astype(np.float64).sum() * dw - 1) < 5e-3
        assert abs(out["incoming_pdf"].astype(np.float64).sum() * dw - 1) < 5e-3
        assert (out["vsp"] >= 0).all() and (out["vsp"] <= 1).all()
        assert (out["pdf"] >= 0).all()


@pytest.mark.parametrize("is_volume,g", [(0, 0.0), (1, 0.6)])
def test_sample_pdf_draws_from_pdf(orc, is_volume, g):
    P, r = orc
    rng = np.random.default_rng(1)
    n = 200000
    p, a = (0.2, -0.4, 0.1), (0.0, 1.0, 0.0) if not is_volume else (0.3, 0.2, 0.9)
    a = np.array(a) / np.linalg.norm(a)
    u = rng.random((n, 2)).astype(np.float32)
    out = r.guiding_query_batch(is_volume, g, np.tile(p, (n, 1)), np.tile(a, (n, 1)), np.tile((0, 0, 1), (n, 1)), u)
    ws, pdf_s = out["ws"].astype(np.float64), out["pdf_s"].astype(np.float64)
    assert np.allclose(np.linalg.norm(ws, axis=1), 1, atol=1e-4)
    # histogram of the samples over a coarse (cos theta, phi) grid == the PDF integrated per cell
    nb_t, nb_p = 8, 12
    dirs, dw = sphere_quadrature(400, 600)
    m = dirs.shape[0]
    qd = r.guiding_query_batch(is_volume, g, np.tile(p, (m, 1)), np.tile(a, (m, 1)), dirs, np.zeros((m, 2)))["pdf"].astype(np.float64)

    def cell(w):
        it = np.clip(((w[:, 2] + 1) / 2 * nb_t).astype(int), 0, nb_t - 1)
        ip = np.clip(((np.arctan2(w[:, 1], w[:, 0]) % (2 * np.pi)) / (2 * np.pi) * nb_p).astype(int), 0, nb_p - 1)
        return it * nb_p + ip

    expected = np.bincount(cell(dirs.astype(np.float64)), weights=qd * dw, minlength=nb_t * nb_p)
    observed = np.bincount(cell(ws), minlength=nb_t * nb_p) / n
    assert abs(expected.sum() - 1) < 5e-3
    sigma = np.sqrt(np.maximum(expected, 1e-9) / n)
    assert np.all(np.abs(observed - expected) < 5 * sigma + 2e-3 * expected + 1e-4), np.max(np.abs(observed - expected) / sigma)
    # PDF evaluated at the sample equals the pdf SamplePDF returned
    out2 = r.guiding_query_batch(is_volume, g, np.tile(p, (n, 1)), np.tile(a, (n, 1)), out["ws"], u)
    assert np.array_equal(out2["pdf"], out["pdf_s"])


def test_untrained_or_outside_region_fails_init(orc):
    P, r = orc
    out = r.guiding_query_batch(0, 0.0, [(0.1, 0.2, 0.3)], [(0, 1, 0)], [(0, 0, 1)], [(0.5, 0.5)])
    assert out["ok"][0] == 1
    scene = oracle_lib.fog_box_scene(16, 16)
    r2 = oracle_lib.OracleRenderer(scene, oracle_lib.default_params(), 16, 16)  # no field uploaded
    out = r2.guiding_query_batch(0, 0.0, [(0.1, 0.2, 0.3)], [(0, 1, 0)], [(0, 0, 1)], [(0.5, 0.5)])
    assert out["ok"][0] == 0 and out["vsp"][0] == -1
    r2.close()


def _render_mean(prm, field, waves, W=40, H=30, seed=0):
    scene = oracle_lib.fog_box_scene(W, H)
    r = oracle_lib.OracleRenderer(scene, prm, W, H, seed=seed)
    if field is not None:
        r.set_guiding_field(field, field)
    for w in range(waves):
        r.render_wave(w, w + 1)
        r.post_process_wave()
    f = r.film_f64()
    img = f[..., :3] / f[..., 3:4]
    r.close()
    return img


def test_guided_rendering_is_unbiased():
    P = load_package()
    field = scenes.light_field(P, n=4)
    base = oracle_lib.app_f_params()
    ref = _render_mean(base, None, 400).reshape(-1, 3).mean(0)
    variants = []
    for sg, vg, stype, vtype, sec in ((1, 1, P.GUIDE_RIS, P.GUIDE_MIS, 1),   # the reference's defaults
                                      (1, 1, P.GUIDE_MIS, P.GUIDE_RIS, 0),
                                      (0, 0, P.GUIDE_RIS, P.GUIDE_MIS, 1)):  # secondary VSP only
        prm = oracle_lib.default_params()
        prm.surfaceguiding, prm.volumeguiding = sg, vg
        prm.surfaceguidingtype, prm.volumeguidingtype = stype, vtype
        prm.vspsecondaryguiding = sec
        variants.append(_render_mean(prm, field, 400).reshape(-1, 3).mean(0))
    for v in variants:
        assert np.allclose(v, ref, rtol=0.015), (v, ref)


def test_guiding_towards_the_light_reduces_variance():
    # lobes aimed at the light: the per-pixel variance of the indirect estimate drops vs unguided
    P = load_package()
    field = scenes.light_field(P, n=4)
    prm_u = oracle_lib.app_f_params()
    prm_u.usenee = 0  # without NEE, finding the small light is all up to directional sampling
    prm_g = oracle_lib.default_params()
    prm_g.usenee = 0
    prm_g.vspsecondaryguiding = 0
    imgs_u = np.stack([_render_mean(prm_u, None, 24, seed=s) for s in range(6)])
    imgs_g = np.stack([_render_mean(prm_g, field, 24, seed=s) for s in range(6)])
    var_u, var_g = imgs_u.var(axis=0).mean(), imgs_g.var(axis=0).mean()
    print("variance unguided %.4g guided %.4g" % (var_u, var_g))
    assert np.allclose(imgs_u.mean(), imgs_g.mean(), rtol=0.06)
    assert var_g < 0.6 * var_u


# ---------------------------------------------------------------------------------------------
# a18: training (recording hooks restated from guiding.h:682-832; PropagateSamples / Field::Update own design)
# ---------------------------------------------------------------------------------------------
def test_training_records_samples_and_fits_the_field():
    P = load_package()
    W, H = 48, 36
    scene = oracle_lib.fog_box_scene(W, H)
    prm = oracle_lib.default_params()
    prm.guide_num_training_waves = 4
    r = oracle_lib.OracleRenderer(scene, prm, W, H, seed=4)
    st = r.training_stats()
    assert st["training"] == 1 and st["iteration"] == 0 and st["n_regions"] == [1, 1]
    r.render_wave(0, 1)
    smp = r.train_samples()
    st = r.training_stats()
    assert st["n_samples"] == len(smp) > 1000
    assert np.allclose(np.linalg.norm(smp["dir"], axis=1), 1, atol=1e-5)
    assert (smp["weight"] > 0).all() and np.isfinite(smp["weight"]).all() and (smp["pdf"] > 0).all() and (smp["distance"] > 0).all()
    assert set(np.unique(smp["flags"])) <= {0, 1, 2, 3}
    vol = (smp["flags"] & 1) != 0
    assert vol.any() and (~vol).any()
    assert (np.abs(smp["p"][vol]) <= 1.0 + 1e-4).all()                       # medium vertices lie inside the fog box
    assert np.isclose(np.abs(smp["p"][~vol]).max(axis=1), 1.0, atol=2e-3).all()  # surface vertices on a wall
    # the distance is the length of the next path segment: the next vertex is again in / on the box
    nxt = smp["p"] + smp["dir"] * smp["distance"][:, None]
    assert (np.abs(nxt) <= 1.0 + 1e-3).all()
    for w in range(4):
        if w:
            r.render_wave(w, w + 1)
        r.post_process_wave()
    st = r.training_stats()
    assert st["training"] == 0 and st["iteration"] == 4 and st["n_samples"] == 0
    for volume in (0, 1):
        nodes, regs, nn, nr = r.get_guiding_field(volume)
        assert nn == 2 * nr - 1  # binary tree: every split adds two nodes and one region
        for i in range(nr):
            R = regs[i]
            assert R.n_lobes == P.VSPG_FIELD_LOBES
            w = np.array(list(R.weight)); k = np.array(list(R.kappa)); v = np.array(list(R.vsp))
            mu = np.array([list(R.mu[a]) for a in range(3)])
            assert abs(w.sum() - 1) < 1e-5 and (w > 0).all()
            assert (k >= 1e-2).all() and (k <= 1e4).all()
            assert np.allclose(np.linalg.norm(mu, axis=0), 1, atol=1e-4)
            assert (v >= 0).all() and (v <= 1).all()
    r.close()


def test_in_loop_training_is_unbiased_and_uploading_a_field_stops_it():
    W, H = 48, 36
    scene = oracle_lib.fog_box_scene(W, H)

    def mean(guided, waves=40):
        prm = oracle_lib.default_params()
        prm.guide_num_training_waves = 12
        if not guided:
            prm.surfaceguiding = prm.volumeguiding = prm.vspsecondaryguiding = 0
        r = oracle_lib.OracleRenderer(scene, prm, W, H, seed=6)
        for w in range(waves):
            r.render_wave(w, w + 1)
            r.post_process_wave()
        f = r.film_f64()
        st = r.training_stats()
        r.close()
        return (f[..., :3] / f[..., 3:4]).mean(), st

    mu, _ = mean(False)
    mg, st = mean(True)
    assert st["iteration"] == 12 and st["training"] == 0
    assert abs(mg - mu) <= 0.02 * mu, (mg, mu)
    P = load_package()
    r = oracle_lib.OracleRenderer(scene, oracle_lib.default_params(), W, H)
    assert r.training_stats()["training"] == 1
    field = scenes.light_field(P, n=2)
    r.set_guiding_field(field, field)
    assert r.training_stats()["training"] == 0
    r.render_wave(0, 1)
    assert r.training_stats()["n_samples"] == 0
    r.close()


def test_guided_russian_roulette_is_unbiased_and_active():
    """rrguiding (guidedvolpathvspgintegrator.cpp:195-197, 274-285, 465-472, 597-600, 817-830): the survival probability comes
    from throughput / pixel contribution estimate once the image-space buffer is ready, minRRDepth becomes 1; the estimator
    stays unbiased (any survival probability in (0,1] is) and the paths really change."""
    import oracle_lib
    from conftest import load_package
    P = load_package()
    W, H = 40, 30
    scene = oracle_lib.fog_box_scene(W, H)
    means, segs = [], []
    for rrg in (0, 1):
        prm = oracle_lib.app_f_params()
        prm.minrrdepth = 1
        prm.maxdepth = 8
        prm.rrguiding = rrg
        r = oracle_lib.OracleRenderer(scene, prm, W, H)
        for w in range(192):
            r.render_wave(w, w + 1)
            r.post_process_wave()
        f = r.film_f64()
        means.append((f[..., :3] / f[..., 3:4]).reshape(-1, 3).mean(0))
        c = r.counters()
        segs.append(c["segments"] / c["paths"])
        r.close()
    assert np.allclose(means[0], means[1], rtol=0.03), means
    assert abs(segs[0] - segs[1]) > 0.02, segs  # another survival rule, other path lengths

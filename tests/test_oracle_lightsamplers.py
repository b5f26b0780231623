"""PowerLightSampler / BVHLightSampler restated in the oracle (lightsamplers.h:63-98, 100-430, lightsamplers.cpp:76-262).
The reference's unit tests for them (lightsamplers_test.cpp) need the full pbrt build, so what pins the restatement here are
the properties those tests check: Sample()'s pmf equals PMF() for the sampled light, the PMFs sum to one, lights are drawn
with the frequencies the PMF states -- and the estimator stays unbiased: uniform, power and bvh renders of a scene with
several area lights, a sky and a sun agree in the mean.  CPU only."""
import numpy as np
import pytest

import oracle_lib
from conftest import load_package


def multi_light_scene(P, W, H, sky=True):
    s = P.fog_box_scene(W, H)
    # two more emitters besides the ceiling lamp: a dim large wall, a bright small floor patch (very different powers)
    for k, Le in ((2, (0.6, 0.5, 0.9)), (0, (9.0, 4.0, 1.0))):
        for c in range(3):
            s.quads[k].Le[c] = Le[c]
    if sky:
        P.add_infinite_light(s, P.LIGHT_UNIFORM_INFINITE, (0.3, 0.4, 0.8))
        P.add_infinite_light(s, P.LIGHT_DISTANT, (4.0, 3.5, 3.0), (0.2, 1.0, -0.3))
    return s


def contexts(n, seed):
    rng = np.random.default_rng(seed)
    p = rng.uniform(-0.95, 0.95, (n, 3)).astype(np.float32)
    ns = rng.normal(size=(n, 3)).astype(np.float32)
    ns /= np.linalg.norm(ns, axis=1, keepdims=True)
    ns[n // 2:] = 0  # medium vertices: no normal
    return p, ns, rng.random(n).astype(np.float32)


@pytest.mark.parametrize("sampler", ["power", "bvh"])
@pytest.mark.parametrize("sky", [False, True])
def test_sampled_pmf_equals_pmf_and_sums_to_one(sampler, sky):
    P = load_package()
    W, H = 16, 12
    scene = multi_light_scene(P, W, H, sky)
    prm = P.app_f_params()
    prm.lightsampler = P.LIGHTSAMPLER_POWER if sampler == "power" else P.LIGHTSAMPLER_BVH
    r = oracle_lib.OracleRenderer(scene, prm, W, H)
    n_lights = 3 + (2 if sky else 0)
    p, ns, u = contexts(20000, 3)
    li, pmf = r.light_sample_batch(p, ns, u)
    ok = li >= 0
    assert ok.mean() > 0.95
    assert li[ok].max() < n_lights and len(np.unique(li[ok])) == n_lights       # every light gets sampled somewhere
    back = r.light_pmf_batch(p[ok], ns[ok], li[ok])
    assert np.array_equal(back.view(np.uint32), pmf[ok].view(np.uint32))           # Sample's pmf IS PMF(ctx, light), bit for bit
    total = sum(r.light_pmf_batch(p, ns, np.full(len(p), k, dtype=np.int32)).astype(np.float64) for k in range(n_lights))
    if sampler == "power":
        assert np.allclose(total, 1, atol=1e-5)
    else:
        # a light whose bounds cannot be seen from the context has importance 0 in both children of its parent only if the
        # whole subtree is invisible; wherever a light WAS sampled the PMFs of all lights add up to one
        assert np.allclose(total[ok], 1, atol=1e-4)
    r.close()


@pytest.mark.parametrize("sampler", ["power", "bvh"])
def test_lights_are_drawn_with_the_stated_frequencies(sampler):
    P = load_package()
    W, H = 16, 12
    scene = multi_light_scene(P, W, H, True)
    prm = P.app_f_params()
    prm.lightsampler = P.LIGHTSAMPLER_POWER if sampler == "power" else P.LIGHTSAMPLER_BVH
    r = oracle_lib.OracleRenderer(scene, prm, W, H)
    rng = np.random.default_rng(11)
    for p0, n0 in (((0.1, -0.4, 0.2), (0, 1, 0)), ((-0.6, 0.5, -0.3), (0, 0, 0)), ((0.7, 0.8, 0.6), (-0.6, -0.8, 0))):
        N = 200000
        p = np.tile(np.asarray(p0, dtype=np.float32), (N, 1))
        ns = np.tile(np.asarray(n0, dtype=np.float32), (N, 1))
        li, _ = r.light_sample_batch(p, ns, rng.random(N).astype(np.float32))
        for k in range(5):
            pk = float(r.light_pmf_batch(p[:1], ns[:1], np.array([k], dtype=np.int32))[0])
            fk = float(np.mean(li == k))
            assert abs(fk - pk) < 4 * np.sqrt(max(pk * (1 - pk), 1e-6) / N) + 1e-4, (sampler, p0, k, fk, pk)
    r.close()


def test_single_light_is_the_round_two_behaviour():
    """One light: every sampler picks it with pmf 1 (what rounds 1-2 rendered, and what their goldens were taken with)."""
    P = load_package()
    W, H = 16, 12
    scene = P.fog_box_scene(W, H)
    p, ns, u = contexts(2000, 5)
    for ls in (P.LIGHTSAMPLER_UNIFORM, P.LIGHTSAMPLER_POWER, P.LIGHTSAMPLER_BVH):
        prm = P.app_f_params()
        prm.lightsampler = ls
        r = oracle_lib.OracleRenderer(scene, prm, W, H)
        li, pmf = r.light_sample_batch(p, ns, u)
        seen = li >= 0
        assert np.all(li[seen] == 0) and np.all(pmf[seen] == 1.0)
        if ls != P.LIGHTSAMPLER_BVH:
            assert seen.all()
        else:  # the bvh sampler returns no light where the lamp cannot be seen (importance 0: behind its emitting side)
            assert seen.mean() > 0.9
        r.close()


def test_uniform_power_and_bvh_renders_agree_in_the_mean():
    P = load_package()
    W, H, spp = 32, 24, 48
    scene = multi_light_scene(P, W, H, True)
    means = {}
    for name, ls in (("uniform", P.LIGHTSAMPLER_UNIFORM), ("power", P.LIGHTSAMPLER_POWER), ("bvh", P.LIGHTSAMPLER_BVH)):
        prm = P.app_f_params()
        prm.lightsampler = ls
        prm.maxdepth = 3
        r = oracle_lib.OracleRenderer(scene, prm, W, H)
        r.render_wave(0, spp, 8)
        f = r.film()
        means[name] = (f[..., :3] / f[..., 3:4]).mean(axis=(0, 1))
        r.close()
    for name in ("power", "bvh"):
        assert np.allclose(means[name], means["uniform"], rtol=0.03), (name, means)

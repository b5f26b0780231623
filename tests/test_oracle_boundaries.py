"""Medium boundaries in the CPU oracle (round 4; row X3 of the verdict): MediumInterface + Material "interface"
(guidedvolpathvspgintegrator.cpp:318, :399-404, :1196-1243; interaction.h:117-121), Shape "sphere" (shapes.h:107-330).
The reference's TUs for these need absent third-party code, so the checks are analytic: known transmittances,
equivalent scene descriptions that must give identical or statistically equal images.  The interval arithmetic under the
sphere test is pinned against the reference's own Interval class (tests/golden/primitives.json: "interval_ops")."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib
from conftest import load_package
from scenes import add_quad, add_sphere, cloud_density, cloud_scene, empty_scene, interface_box


def _img(film):
    return film[..., :3] / np.maximum(film[..., 3:4], 1e-30)


def _render(scene, prm, W, H, spp, threads=8):
    r = oracle_lib.OracleRenderer(scene, prm, W, H)
    for w in range(spp):
        r.render_wave(w, w + 1, threads)
        r.post_process_wave()
    f, c = r.film_f64(), r.counters()
    r.close()
    return f, c


def _sphere_pod(P, center=(0, 0, 0), radius=1.0, scale=(1, 1, 1)):
    s = P.VspgScene()
    add_sphere(s, center, radius, scale=scale)
    return s.spheres[0]


def test_sphere_intersection_against_double_precision():
    """Sphere::BasicIntersect restated: t, hit point and normal against a float64 solution of the same quadratic;
    the returned interval contains the exact point; rays from inside hit the far side; a ray spawned from a hit
    (origin pushed out of the error box) does not hit the surface it left."""
    P = load_package()
    lib = oracle_lib.load()
    rng = np.random.default_rng(3)
    c, R = np.array([0.3, -0.2, 0.5]), 1.25
    sp = _sphere_pod(P, c, R)
    hit, t = C.c_int(), C.c_float()
    plo, phi, n, du = P.f3(), P.f3(), P.f3(), P.f3()
    n_hit = n_inside = 0
    for i in range(4000):
        inside = i % 4 == 0
        o = c + rng.normal(size=3) * (0.4 if inside else 2.5)
        if not inside and np.linalg.norm(o - c) < R * 1.01:
            continue
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        o32, d32 = o.astype(np.float32), d.astype(np.float32)
        lib.oracle_sphere_intersect(C.byref(sp), P.f3(*o32), P.f3(*d32), C.c_float(np.inf), C.byref(hit), C.byref(t), plo, phi, n, du)
        oo, dd = o32.astype(np.float64) - c.astype(np.float32).astype(np.float64), d32.astype(np.float64)
        A, B, Cc = dd @ dd, 2 * (dd @ oo), oo @ oo - R * R
        disc = B * B - 4 * A * Cc
        want = None
        if disc > 1e-9:
            r0, r1 = (-B - np.sqrt(disc)) / (2 * A), (-B + np.sqrt(disc)) / (2 * A)
            want = r0 if r0 > 1e-4 else (r1 if r1 > 1e-4 else None)
        if want is None:
            if disc < -1e-6:
                assert not hit.value
            continue
        assert hit.value, (o, d)
        n_hit += 1
        n_inside += inside
        assert abs(t.value - want) <= 2e-5 * max(1.0, want)
        pe = oo + want * dd + c.astype(np.float32).astype(np.float64)
        lo, hi = np.array(list(plo)), np.array(list(phi))
        assert np.all(lo <= pe + 2e-6) and np.all(pe - 2e-6 <= hi)
        nn = np.array(list(n))
        assert abs(np.linalg.norm(nn) - 1) < 1e-6 and np.allclose(nn, (pe - c) / R, atol=2e-5)   # outward normal
        assert abs(np.dot(np.array(list(du)), nn)) < 1e-5                                          # dpdu is tangent
        # leave the surface along d (to the outside when the ray came from inside and vice versa): no self-hit at t ~ 0
        pm, perr = (lo + hi) / 2, (hi - lo) / 2
        out = P.f3()
        lib.oracle_offset_ray_origin(P.f3(*pm.astype(np.float32)), P.f3(*perr.astype(np.float32)), n, P.f3(*d32), out)
        h2, t2 = C.c_int(), C.c_float()
        lib.oracle_sphere_intersect(C.byref(sp), out, P.f3(*d32), C.c_float(np.inf), C.byref(h2), C.byref(t2), plo, phi, P.f3(), P.f3())
        if h2.value:
            assert t2.value > 1e-3, "self-intersection after SpawnRay"
    assert n_hit > 800 and n_inside > 300


def test_scaled_sphere_is_an_ellipsoid_and_handedness_flips_the_normal():
    P = load_package()
    lib = oracle_lib.load()
    hit, t = C.c_int(), C.c_float()
    plo, phi, n, du = P.f3(), P.f3(), P.f3(), P.f3()
    sp = _sphere_pod(P, (0, 0, 0), 1.0, scale=(2, 1, 0.5))
    lib.oracle_sphere_intersect(C.byref(sp), P.f3(-5, 0, 0), P.f3(1, 0, 0), C.c_float(np.inf), C.byref(hit), C.byref(t), plo, phi, n, du)
    assert hit.value and abs(t.value - 3.0) < 1e-5 and np.allclose(list(n), [-1, 0, 0], atol=1e-6)
    lib.oracle_sphere_intersect(C.byref(sp), P.f3(0, 0, -5), P.f3(0, 0, 1), C.c_float(np.inf), C.byref(hit), C.byref(t), plo, phi, n, du)
    assert hit.value and abs(t.value - 4.5) < 1e-5 and np.allclose(list(n), [0, 0, -1], atol=1e-6)
    # a mirroring transform swaps handedness: reverseOrientation ^ transformSwapsHandedness flips n (shapes.h:276)
    sm = _sphere_pod(P, (0, 0, 0), 1.0, scale=(-1, 1, 1))
    lib.oracle_sphere_intersect(C.byref(sm), P.f3(0, 0, -5), P.f3(0, 0, 1), C.c_float(np.inf), C.byref(hit), C.byref(t), plo, phi, n, du)
    assert hit.value and np.allclose(list(n), [0, 0, 1], atol=1e-6)
    # tMax: BasicIntersect rejects a hit whose interval reaches past tMax
    lib.oracle_sphere_intersect(C.byref(sp), P.f3(-5, 0, 0), P.f3(1, 0, 0), C.c_float(2.9), C.byref(hit), C.byref(t), plo, phi, n, du)
    assert not hit.value


def test_escaping_rays_are_not_sampled():
    """:318 `if (ray.medium && !std::isinf(tMax))`: a medium that fills an EMPTY scene never scatters anything -- every
    camera ray escapes and picks up the sky's radiance exactly (rounds 1-3 sampled such rays)."""
    P = load_package()
    W, H = 16, 12
    s = empty_scene(W, H, (0, 0, -3), (0, 0, 0))
    s.medium.type = P.MEDIUM_HOMOGENEOUS
    s.medium.sigma_a[:] = (0.5,) * 3
    s.medium.sigma_s[:] = (2.0,) * 3
    P.add_infinite_light(s, P.LIGHT_UNIFORM_INFINITE, (0.25, 0.5, 1.0))
    f, c = _render(s, oracle_lib.app_f_params(), W, H, 4)
    assert np.array_equal(_img(f), np.broadcast_to(np.float32([0.25, 0.5, 1.0]).astype(np.float64), (H, W, 3)))
    assert c["volume_scatters"] == 0 and c["segments"] == c["paths"]


def test_non_transition_interfaces_change_nothing():
    """MediumInterface "fog" "fog" on every wall (inside == outside: not a transition, interaction.h:225-228) is the fog box."""
    P = load_package()
    W, H = 24, 16
    a = oracle_lib.fog_box_scene(W, H)
    b = oracle_lib.fog_box_scene(W, H)
    for k in range(b.n_quads):
        b.quads[k].medium_interface = P.IFACE_INSIDE | P.IFACE_OUTSIDE
    prm = oracle_lib.app_f_params()
    fa, _ = _render(a, prm, W, H, 3)
    fb, _ = _render(b, prm, W, H, 3)
    assert np.array_equal(fa, fb)


def test_absorbing_sphere_in_front_of_the_sky_has_the_analytic_transmittance():
    """Camera in vacuum, an interface-material sphere holding a purely absorbing homogeneous medium, a uniform sky:
    a pixel sees L_sky * exp(-sigma_a * chord).  Exercises the camera's medium, SkipIntersection on entry and exit
    (:399-404), GetMedium on both crossings, and :318 on the final escaping segment."""
    P = load_package()
    W, H, spp = 24, 24, 192
    R, sig = 1.0, 0.6
    s = empty_scene(W, H, (0, 0, -4), (0, 0, 0), fov=35.0)
    s.medium.type = P.MEDIUM_HOMOGENEOUS
    s.medium.sigma_a[:] = (sig,) * 3
    s.medium.sigma_s[:] = (0.0,) * 3
    s.camera_outside_medium = 1
    add_sphere(s, (0, 0, 0), R, material=P.MATERIAL_INTERFACE, iface=P.IFACE_INSIDE)
    P.add_infinite_light(s, P.LIGHT_UNIFORM_INFINITE, (1.0, 1.0, 1.0))
    prm = oracle_lib.app_f_params()
    f, c = _render(s, prm, W, H, spp)
    img = _img(f)[..., 0]
    cam = s.camera
    o = np.array(list(cam.origin), dtype=np.float64)
    right, up, fwd = (np.array(list(v), dtype=np.float64) for v in (cam.right, cam.up, cam.fwd))
    want = np.ones((H, W))
    for y in range(H):
        for x in range(W):
            acc = 0.0
            for sy in (0.25, 0.75):
                for sx in (0.25, 0.75):
                    dc = np.array([cam.sx * (x + sx) + cam.ox, cam.sy * (y + sy) + cam.oy, 1.0])
                    d = right * dc[0] + up * dc[1] + fwd * dc[2]
                    d /= np.linalg.norm(d)
                    b, cc = 2 * (d @ o), o @ o - R * R
                    disc = b * b - 4 * cc
                    acc += np.exp(-sig * np.sqrt(disc)) if disc > 0 else 1.0
            want[y, x] = acc / 4
    inside = want < 0.75
    assert inside.sum() > 40
    assert np.all(img[want == 1.0] == 1.0)                       # rays that miss the sphere: the sky, exactly
    assert abs(img[inside].mean() / want[inside].mean() - 1) < 0.02
    assert np.abs(img[inside] - want[inside]).max() < 0.15       # binary estimator, 192 spp


def test_shadow_ray_walks_through_the_boundary():
    """SampleLd's loop (:1196-1243): a distant light straight above, an absorbing interface sphere over a diffuse ground --
    the ground under the sphere's centre receives Kd/pi * E * exp(-sigma_a * 2R); beside the sphere Kd/pi * E."""
    P = load_package()
    W, H, spp = 32, 32, 256
    R, sig, kd = 0.5, 0.7, 0.5
    s = empty_scene(W, H, (0.0, 3.0, -4.0), (0, 0.0, 0.0), fov=20.0)   # looks down at the ground past the sphere, never through it
    s.medium.type = P.MEDIUM_HOMOGENEOUS
    s.medium.sigma_a[:] = (sig,) * 3
    s.medium.sigma_s[:] = (0.0,) * 3
    s.camera_outside_medium = 1
    add_sphere(s, (0, 1.2, 0), R, material=P.MATERIAL_INTERFACE, iface=P.IFACE_INSIDE)
    add_quad(s, (-8, 0, -8), (0, 0, 16), (16, 0, 0), kd=(kd,) * 3)                 # ground y = 0, n = +y
    P.add_infinite_light(s, P.LIGHT_DISTANT, (np.pi,) * 3, (0, 1, 0))
    prm = oracle_lib.app_f_params()
    r = oracle_lib.OracleRenderer(s, prm, W, H)
    # the pixels whose ground point lies at |x| < 0.05, z ~ 0 (under the centre) and at |x| > 0.8 (outside the shadow)
    cam = s.camera
    o = np.array(list(cam.origin), dtype=np.float64)
    right, up, fwd = (np.array(list(v), dtype=np.float64) for v in (cam.right, cam.up, cam.fwd))
    centre, clear = [], []
    for y in range(H):
        for x in range(W):
            dc = np.array([cam.sx * (x + 0.5) + cam.ox, cam.sy * (y + 0.5) + cam.oy, 1.0])
            d = right * dc[0] + up * dc[1] + fwd * dc[2]
            if d[1] >= 0:
                continue
            g = o + d * (-o[1] / d[1])
            rr = np.hypot(g[0], g[2])
            if rr < 0.12:
                centre.append((x, y, rr))
            elif rr > 0.75 and g[2] < 0.3:
                clear.append((x, y))
    assert len(centre) >= 2 and len(clear) >= 8
    for w in range(spp):
        r.render_wave(w, w + 1, 8)
    img = _img(r.film_f64())[..., 0]
    r.close()
    for x, y in clear:
        assert abs(img[y, x] - kd) < 1e-5        # deterministic: delta light, no occluder, no inter-reflection
    got = np.mean([img[y, x] for x, y, _ in centre])
    want = np.mean([kd * np.exp(-sig * 2 * np.sqrt(R * R - rr * rr)) for _, _, rr in centre])
    assert abs(got / want - 1) < 0.05, (got, want)


@pytest.mark.parametrize("shape", ["box", "sphere"])
def test_bounded_medium_equals_the_all_filling_medium(shape):
    """A grid medium is zero outside its bounds, so a scene where it fills everything (rounds 1-3) and the same scene with
    the camera in vacuum and an interface-material shape around the medium's bounds are the same radiometric problem: equal
    images up to Monte-Carlo noise (the paths differ: other sampler dimensions are consumed).  Closed by a diffuse shell with
    a light so that every segment ends on a surface in both descriptions."""
    P = load_package()
    W, H, spp, n = 20, 16, 96, 16
    dens = cloud_density(n)

    def build(bounded):
        s = empty_scene(W, H, (0, 0.1, -2.6), (0, 0, 0), fov=45.0)
        m = s.medium
        m.type = P.MEDIUM_GRID
        m.sigma_a[:] = (0.4,) * 3
        m.sigma_s[:] = (3.0,) * 3
        m.g = 0.3
        m.nx = m.ny = m.nz = n
        m.bounds_min[:] = (-0.7, -0.7, -0.7)
        m.bounds_max[:] = (0.7, 0.7, 0.7)
        m.density = dens.ctypes.data_as(C.POINTER(C.c_float))
        s._density_keepalive = dens
        for p00, e1, e2 in [((-3, -3, -3), (0, 0, 6), (6, 0, 0)), ((-3, 3, -3), (6, 0, 0), (0, 0, 6)),      # floor n=+y, ceiling n=-y
                            ((-3, -3, 3), (0, 6, 0), (6, 0, 0)), ((-3, -3, -3), (6, 0, 0), (0, 6, 0)),       # back n=-z, front n=+z
                            ((-3, -3, -3), (0, 6, 0), (0, 0, 6)), ((3, -3, -3), (0, 0, 6), (0, 6, 0))]:      # left n=+x, right n=-x
            add_quad(s, p00, e1, e2, kd=(0.6, 0.6, 0.6))
        add_quad(s, (-0.8, 2.99, -0.8), (1.6, 0, 0), (0, 0, 1.6), le=(6, 6, 6), kd=(0, 0, 0))                   # light, n = -y
        if bounded:
            s.camera_outside_medium = 1
            if shape == "box":
                interface_box(s, (-0.7, -0.7, -0.7), (0.7, 0.7, 0.7))
            else:
                add_sphere(s, (0, 0, 0), 1.25, material=P.MATERIAL_INTERFACE, iface=P.IFACE_INSIDE)
        return s
    prm = oracle_lib.app_f_params()
    fa, ca = _render(build(False), prm, W, H, spp)
    fb, cb = _render(build(True), prm, W, H, spp)
    ia, ib = _img(fa), _img(fb)
    print("all-filling mean %.5f, bounded mean %.5f; scatters per path %.3f / %.3f" % (
        ia.mean(), ib.mean(), ca["volume_scatters"] / ca["paths"], cb["volume_scatters"] / cb["paths"]))
    assert abs(ia.mean() / ib.mean() - 1) < 0.03
    blk = lambda im: im.reshape(4, H // 4, 4, W // 4, 3).mean(axis=(1, 3, 4))
    assert np.all(np.abs(blk(ia) / blk(ib) - 1) < 0.2)
    assert abs(ca["volume_scatters"] / cb["volume_scatters"] - 1) < 0.05


def test_cloud_scene_shape_renders():
    """The reference's cloud-scene shape (camera in vacuum, MediumInterface "cloud" "" + interface material on a sphere,
    ground, sun + sky) under the reference's default sampling options: finite, and darker under the cloud."""
    P = load_package()
    W, H = 32, 24
    s = cloud_scene(W, H, cloud_density(16), 16)
    prm = oracle_lib.app_f_params()
    f, c = _render(s, prm, W, H, 8)
    img = _img(f)
    assert np.all(np.isfinite(img)) and img.min() >= 0 and img.mean() > 0.05
    assert c["volume_scatters"] > 0 and c["segments"] > 2 * c["paths"]

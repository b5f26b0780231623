"""The reference's own geometry and light-sampler TEST SCENARIOS replayed against the oracle (CPU) and the device (-m gpu).

The reference's tests need the full pbrt build (gtest + its libraries: unbuildable here), so what travels is the scenario: the
statement sequence of each test with its constants, cited line by line, driven by the reference's own RNG (PCG32, pinned
bit for bit: tests/golden/primitives.json) -- and the property the test asserts.  Ray queries go through vspg_ray_batch /
oracle_ray_batch (Integrator::Intersect, Interaction::SpawnRay / SpawnRayTo, Intersect + IntersectP of the spawned ray).

  Triangle, Watertight        shapes_test.cpp:33-154   (disabled there with #if 0 "fails on github's ubuntu-latest servers": run here)
  Triangle, Reintersect       shapes_test.cpp:156-208
  FullSphere, Reintersect     shapes_test.cpp:339-355 over TestReintersectConvex :251-316
  Triangle, BadCases          shapes_test.cpp:435-449
  {BVH,Power,Uniform}LightSampling, PdfMethod   lightsamplers_test.cpp:276-350 (the lights here: emissive rectangles, a sky, a sun)
  Spectrum, Blackbody         util/spectrum_test.cpp:19-48 (Planck's law against spectralcalc.com's values, Wien's displacement law)
  SampleDiscrete, Basics      util/sampling_test.cpp:26-42 (the two-weight cases: the form the path uses, media_sampleTMaj / :635)

Scenario generators mirror the reference's float arithmetic where it decides what is tested (vertices, rays); a random transform
is "a random transform of that family" (Scale * Translate * Rotate, shapes_test.cpp:318-335), not pbrt's bits."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib
from conftest import load_package

f32 = np.float32
ONE_MINUS_EPS = np.nextafter(f32(1), f32(0))


class PbrtRNG:
    """RNG(seqIndex) (util/rng.h): raw PCG32 outputs from the oracle's pinned generator, Uniform<Float>() and Uniform<uint32_t>(b) on top."""

    def __init__(self, seq, n=8192):
        lib = oracle_lib.load()
        u = (C.c_uint32 * n)()
        f = (C.c_float * n)()
        lib.oracle_rng_seq(C.c_uint64(seq), C.c_uint64(0), 0, C.c_int64(0), n, u, f)
        self.u = np.frombuffer(u, dtype=np.uint32).copy()
        self.i = 0

    def u32(self):
        v = int(self.u[self.i])
        self.i += 1
        return v

    def uniform(self):  # rng.h:128-130: min(OneMinusEpsilon, Uniform<uint32_t>() * 0x1p-32f)
        return min(ONE_MINUS_EPS, f32(f32(self.u32()) * f32(2.0 ** -32)))

    def uniform_u32(self, b):  # rng.h:91-99
        threshold = ((1 << 32) - b) % b
        while True:
            r = self.u32()
            if r >= threshold:
                return r % b


def lerp(x, a, b):  # math.h:210: (1 - x) * a + x * b in Float
    return f32(f32(f32(1) - x) * f32(a)) + f32(x * f32(b))


def p_exp(rng, e=8.0):  # shapes_test.cpp:22-25: std::pow(10, Lerp(u, -exp, exp)), computed in double, returned as Float
    return f32(math.pow(10.0, float(lerp(rng.uniform(), -e, e))))


def sample_uniform_sphere(u0, u1):
    out = (C.c_float * 3)()
    oracle_lib.load().oracle_sample_uniform_sphere(C.c_float(u0), C.c_float(u1), out)
    return np.array(out[:], dtype=f32)


def empty_scene(P, W=8, H=8):
    s = P.VspgScene()
    lib = P.load()
    assert lib.vspg_camera_look_at(C.byref(s.camera), P.f3(0, 0, -5), P.f3(0, 0, 0), P.f3(0, 1, 0), 40.0, W, H) == 0
    s.medium.type = P.MEDIUM_NONE
    return s


def renderers(P, scene, gpu):
    """The oracle, and with gpu the HIP renderer beside it (its ray_batch must agree bit for bit)."""
    prm = P.app_f_params()
    o = oracle_lib.OracleRenderer(scene, prm, 8, 8)
    g = P.Renderer(scene, prm, 8, 8) if gpu else None
    return o, g


def ray_batch(o, g, q):
    a = o.ray_batch(q)
    if g is not None:
        b = g.ray_batch(q)
        for name in a.dtype.names:
            assert np.array_equal(a[name].view(np.uint32), b[name].view(np.uint32)), "device != oracle in " + name
    return a


def queries(P, n):
    q = np.zeros(n, dtype=P.RAY_QUERY_DTYPE)
    q["tMax"] = np.inf
    q["tMax2"] = np.inf
    return q


# ---------------------------------------------------------------------------------------------------------------------
def watertight_mesh():
    """shapes_test.cpp:35-100: a 16 x 16 triangulated sphere, vertices pushed out along their normal by 5 * rng.Uniform()."""
    rng = PbrtRNG(12111)
    nTheta, nPhi = 16, 16
    vertices = []
    for t in range(nTheta):
        theta = f32(f32(math.pi) * f32(t) / f32(nTheta - 1))
        cosTheta, sinTheta = f32(math.cos(float(theta))), f32(math.sin(float(theta)))   # std::cos / std::sin of a Float
        for p in range(nPhi):
            phi = f32(f32(2) * f32(math.pi) * f32(p) / f32(nPhi - 1))
            radius = f32(1)
            if t == 0:
                vertices.append(np.array([0, 0, radius], dtype=f32))
            elif t == nTheta - 1:
                vertices.append(np.array([0, 0, -radius], dtype=f32))
            elif p == nPhi - 1:
                vertices.append(vertices[len(vertices) - (nPhi - 1)].copy())    # "close it up exactly at the end"
            else:
                radius = f32(radius + f32(5) * rng.uniform())
                # SphericalDirection(sinTheta, cosTheta, phi) (vecmath.h:1666-1672), clamped sines / cosines
                st, ct = np.clip(sinTheta, -1, 1), np.clip(cosTheta, -1, 1)
                d = np.array([st * f32(math.cos(float(phi))), st * f32(math.sin(float(phi))), ct], dtype=f32)
                vertices.append((radius * d).astype(f32))
    assert len(vertices) == nTheta * nPhi
    off = lambda t, p: t * nPhi + p
    idx = []
    for p in range(nPhi - 1):
        idx += [off(0, 0), off(1, p), off(1, p + 1)]
    for t in range(1, nTheta - 2):
        for p in range(nPhi - 1):
            idx += [off(t, p), off(t + 1, p), off(t + 1, p + 1), off(t, p), off(t + 1, p + 1), off(t, p + 1)]
    for p in range(nPhi - 1):
        idx += [off(nTheta - 1, 0), off(nTheta - 2, p), off(nTheta - 2, p + 1)]
    v = np.array(vertices, dtype=f32)
    return v, v[np.array(idx)].reshape(-1, 9)


def _watertight(P, gpu, n_rays):
    verts, tris = watertight_mesh()
    # (degenerate fan triangles at the seam -- two equal vertices -- are dropped by the upload as by Triangle::Intersect's first test, shapes.cpp:172-173)
    scene = empty_scene(P)
    P.set_triangles(scene, tris)
    o, g = renderers(P, scene, gpu)
    q = queries(P, 2 * n_rays)
    for i in range(n_rays):
        rng = PbrtRNG(i, 16)
        p = (f32(0.5) * sample_uniform_sphere(rng.uniform(), rng.uniform())).astype(f32)     # a point in the sphere of radius 0.5
        d = sample_uniform_sphere(rng.uniform(), rng.uniform())
        q["o"][2 * i], q["d"][2 * i] = p, d
        pv = verts[rng.uniform_u32(len(verts))]                                               # "now tougher: shoot directly at a vertex"
        q["o"][2 * i + 1], q["d"][2 * i + 1] = p, (pv - p).astype(f32)
    res = ray_batch(o, g, q)
    o.close()
    if g is not None:
        g.close()
    return res


def test_triangle_watertight():
    """No ray from inside the perturbed triangulated sphere escapes it: EXPECT_GE(nHits, 1) for a random direction and for a ray
    aimed exactly at a mesh vertex (shapes_test.cpp:103-153; 100 000 origins there, 20 000 here on the CPU)."""
    res = _watertight(load_package(), False, 20000)
    assert res["hit"].all(), "%d rays escaped" % int((res["hit"] == 0).sum())


@pytest.mark.gpu
def test_triangle_watertight_on_device(gpu_pkg):
    res = _watertight(gpu_pkg, True, 100000)      # the reference's count; device == oracle bit for bit (ray_batch)
    assert res["hit"].all()


# ---------------------------------------------------------------------------------------------------------------------
def _triangle_reintersect(P, gpu, n_tris, n_out):
    checked = 0
    for i in range(n_tris):
        rng = PbrtRNG(i, 16 + 5 * n_out)
        v = np.array([[p_exp(rng) for _ in range(3)] for _ in range(3)], dtype=f32)                  # GetRandomTriangle (:158-176)
        c = np.cross((v[1] - v[0]).astype(np.float64), (v[2] - v[0]).astype(np.float64))
        if float(np.dot(c, c)) < 1e-20:
            continue
        u0, u1 = rng.uniform(), rng.uniform()                                                          # tri.Sample(u) (:186-190)
        if u0 < u1:                                                                                    # SampleUniformTriangle (sampling.h:300-312)
            b0 = f32(u0 / f32(2)); b1 = f32(u1 - b0)
        else:
            b1 = f32(u1 / f32(2)); b0 = f32(u0 - b1)
        b2 = f32(f32(f32(1) - b0) - b1)
        ps = (b0 * v[0] + b1 * v[1] + b2 * v[2]).astype(f32)
        org = np.array([p_exp(rng) for _ in range(3)], dtype=f32)                                     # ray origin (:193-195)
        scene = empty_scene(P)
        P.set_triangles(scene, v.reshape(1, 9))
        o, g = renderers(P, scene, gpu)
        q = queries(P, 2 * n_out)
        q["o"][:], q["d"][:] = org, (ps - org).astype(f32)                                            # Ray r(o, ts->intr.p() - o) (:198)
        for j in range(n_out):
            w = sample_uniform_sphere(rng.uniform(), rng.uniform())                                    # SpawnRay(w) (:209-216)
            q["mode"][2 * j], q["w"][2 * j] = 1, w
            p2 = np.array([p_exp(rng) for _ in range(3)], dtype=f32)                                   # SpawnRayTo(p2), tMax 1 (:219-226)
            q["mode"][2 * j + 1], q["w"][2 * j + 1], q["tMax2"][2 * j + 1] = 2, p2, 1.0
        res = ray_batch(o, g, q)
        o.close()
        if g is not None:
            g.close()
        if not res["hit"][0]:
            continue                                        # "rarely miss, due to round-off error. Just do another go-around" (:200-204)
        checked += 1
        assert not res["any2"].any() and not res["hit2"].any(), "triangle %d: a ray leaving the hit point re-intersects the triangle" % i
    return checked


def test_triangle_reintersect():
    """A ray spawned from a triangle hit with SpawnRay / SpawnRayTo never hits that triangle again: EXPECT_FALSE(tri.IntersectP(rOut)),
    EXPECT_FALSE(tri.Intersect(rOut)) (shapes_test.cpp:156-208; 1000 triangles x 1000 rays there)."""
    assert _triangle_reintersect(load_package(), False, 300, 200) > 250


@pytest.mark.gpu
def test_triangle_reintersect_on_device(gpu_pkg):
    assert _triangle_reintersect(gpu_pkg, True, 120, 1000) > 100


# ---------------------------------------------------------------------------------------------------------------------
def random_transform(rng):
    """shapes_test.cpp:318-335: Scale(pExp(4)^3) * Translate(+-pExp) * Rotate(rr() * 20 deg, SampleUniformSphere(u)); float32 products."""
    def mat(rows):
        return np.array(rows, dtype=f32)
    s = [p_exp(rng, 4) for _ in range(3)]
    S = mat([[s[0], 0, 0, 0], [0, s[1], 0, 0], [0, 0, s[2], 0], [0, 0, 0, 1]])

    def rt():
        f = p_exp(rng)
        return f if rng.uniform() > 0.5 else f32(-f)
    t = [rt(), rt(), rt()]
    T = mat([[1, 0, 0, t[0]], [0, 1, 0, t[1]], [0, 0, 1, t[2]], [0, 0, 0, 1]])
    u = (rng.uniform(), rng.uniform())
    rr = f32(-10.0 + 20.0 * float(rng.uniform()))
    theta = math.radians(float(f32(rr * f32(20))))
    a = sample_uniform_sphere(*u).astype(np.float64)
    a /= np.linalg.norm(a)
    st, ct = math.sin(theta), math.cos(theta)
    R = np.eye(4)
    R[:3, :3] = ct * np.eye(3) + (1 - ct) * np.outer(a, a) + st * np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return ((S @ T).astype(f32) @ R.astype(f32)).astype(f32)


def _sphere_reintersect(P, gpu, n_spheres, n_out):
    checked = 0
    lib = P.load()
    for i in range(n_spheres):
        rng = PbrtRNG(i, 64 + 10 * n_out)
        radius = p_exp(rng, 4)
        for transformed in (False, True):
            m = random_transform(rng) if transformed else np.eye(4, dtype=f32)
            scene = empty_scene(P)
            sp = scene.spheres[0]
            sp.render_from_object[:] = [float(x) for x in m.reshape(16)]
            inv = (C.c_float * 16)()
            if lib.vspg_transform_inverse(sp.render_from_object, inv) != 0:
                continue
            sp.object_from_render[:] = list(inv)
            sp.radius = float(radius)
            sp.Kd[:] = (0.5, 0.5, 0.5)
            scene.n_spheres = 1
            # TestReintersectConvex (:251-316): origin pExp^3, destination a random point of the shape's bounding box
            org = np.array([p_exp(rng) for _ in range(3)], dtype=f32)
            tt = np.array([rng.uniform() for _ in range(3)], dtype=f32)
            corners = np.array([[sx * radius, sy * radius, sz * radius, 1] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], dtype=f32)
            wc = (m @ corners.T).T[:, :3]
            lo, hi = wc.min(axis=0), wc.max(axis=0)
            p2 = ((f32(1) - tt) * lo + tt * hi).astype(f32)
            d = (p2 - org).astype(f32)
            if rng.uniform() < 0.5:
                d = (d / f32(np.sqrt(np.sum(d.astype(np.float64) ** 2)))).astype(f32)
            try:
                o, g = renderers(P, scene, gpu)
            except Exception:
                continue                                     # (a transform the C-ABI refuses: not what is tested here)
            q0 = queries(P, 1)
            q0["o"][0], q0["d"][0] = org, d
            first = ray_batch(o, g, q0)
            if first["hit"][0]:
                n = first["n"][0].astype(f32)
                q = queries(P, 2 * n_out)
                q["o"][:], q["d"][:] = org, d
                for j in range(n_out):
                    w = sample_uniform_sphere(rng.uniform(), rng.uniform())
                    if float(np.dot(w, n)) < 0:              # FaceForward(w, isect.n)
                        w = -w
                    q["mode"][2 * j], q["w"][2 * j] = 1, w
                    pp = np.array([p_exp(rng) for _ in range(3)], dtype=f32)
                    wv = (pp - first["p"][0]).astype(f32)    # the target pulled into the normal's hemisphere (:302-307)
                    if float(np.dot(wv, n)) < 0:
                        wv = -wv
                    q["mode"][2 * j + 1], q["w"][2 * j + 1], q["tMax2"][2 * j + 1] = 2, (first["p"][0] + wv).astype(f32), 1.0
                res = ray_batch(o, g, q)
                checked += 1
                bad = int(res["any2"].sum() + res["hit2"].sum())
                assert bad == 0, "sphere %d (transformed %s): %d re-intersections" % (i, transformed, bad)
            o.close()
            if g is not None:
                g.close()
    return checked


def test_full_sphere_reintersect():
    """EXPECT_EQ(0, TestReintersectConvex(&sphere, rng)) for identity and randomly transformed full spheres (shapes_test.cpp:339-355)."""
    assert _sphere_reintersect(load_package(), False, 150, 100) > 60


@pytest.mark.gpu
def test_full_sphere_reintersect_on_device(gpu_pkg):
    assert _sphere_reintersect(gpu_pkg, True, 60, 500) > 20


# ---------------------------------------------------------------------------------------------------------------------
def _bad_case(P, gpu):
    tri = np.array([[-1113.45459, -79.049614, -56.2431908], [-1113.45459, -87.0922699, -56.2431908], [-1113.45459, -79.2090149, -56.2431908]], dtype=f32)
    scene = empty_scene(P)
    P.set_triangles(scene, tri.reshape(1, 9))
    o, g = renderers(P, scene, gpu)
    q = queries(P, 1)
    q["o"][0] = (-1081.47925, 99.9999542, 87.7701111)
    q["d"][0] = (-32.1072998, -183.355865, -144.607635)
    q["tMax"][0] = 0.9999
    res = ray_batch(o, g, q)
    o.close()
    if g is not None:
        g.close()
    return res


def test_triangle_bad_cases():
    """EXPECT_FALSE(tris[0].Intersect(ray).has_value()) (shapes_test.cpp:435-449)."""
    assert not _bad_case(load_package(), False)["hit"][0]


@pytest.mark.gpu
def test_triangle_bad_cases_on_device(gpu_pkg):
    assert not _bad_case(gpu_pkg, True)["hit"][0]


# ---------------------------------------------------------------------------------------------------------------------
def _random_lights_scene(P, rng):
    """randomLights(20, ...) (lightsamplers_test.cpp:231-270) with this build's light types: 14 emissive rectangles of random place,
    size, orientation and power, a sky and a sun."""
    s = P.fog_box_scene(16, 12)
    s.n_quads = 0
    r = lambda: float(rng.uniform())
    for _ in range(14):
        p = [float(lerp(rng.uniform(), -5, 5)) for _ in range(3)]
        e1 = sample_uniform_sphere(rng.uniform(), rng.uniform()).astype(np.float64) * (0.1 + 2 * r())
        e2 = np.cross(e1, sample_uniform_sphere(rng.uniform(), rng.uniform()).astype(np.float64))
        e2 *= (0.1 + 2 * r()) / max(np.linalg.norm(e2), 1e-6)
        P.add_quad(s, p, list(e1), list(e2), kd=(0.5, 0.5, 0.5), le=(10 * r(), 10 * r(), 10 * r()), two_sided=int(r() < 0.5))
    P.add_infinite_light(s, P.LIGHT_UNIFORM_INFINITE, (r(), r(), r()))
    P.add_infinite_light(s, P.LIGHT_DISTANT, (3 * r(), 3 * r(), 3 * r()), (0.3, 1.0, 0.2))
    return s


def _pdf_method(P, sampler):
    rng = PbrtRNG(5251, 4096)
    scene = _random_lights_scene(P, rng)
    prm = P.app_f_params()
    prm.lightsampler = {"uniform": P.LIGHTSAMPLER_UNIFORM, "power": P.LIGHTSAMPLER_POWER, "bvh": P.LIGHTSAMPLER_BVH}[sampler]
    o = oracle_lib.OracleRenderer(scene, prm, 16, 12)
    n = 100                                                                                   # for (int i = 0; i < 100; ++i) (:285 ...)
    p = np.array([[-1 + 3 * float(rng.uniform()) for _ in range(3)] for _ in range(n)], dtype=f32)   # Point3f p{-1 + 3 r(), ...}
    u = np.array([rng.uniform() for _ in range(n)], dtype=f32)
    ns = np.zeros((n, 3), dtype=f32)                                                           # Interaction(Point3fi(p), Normal3f(0, 0, 0), ...)
    li, pmf = o.light_sample_batch(p, ns, u)
    ok = li >= 0
    back = o.light_pmf_batch(p[ok], ns[ok], li[ok])
    o.close()
    if sampler != "bvh":
        assert ok.all()                                                                        # ASSERT_TRUE((bool)sampledLight) (:313, :330, :347)
    assert ok.sum() >= 50
    assert np.allclose(pmf[ok], back, rtol=4e-7, atol=0)                                      # EXPECT_FLOAT_EQ(sampledLight->p, distrib.PMF(intr, light))


@pytest.mark.parametrize("sampler", ["bvh", "uniform", "power"])
def test_light_sampling_pdf_method(sampler):
    """sampledLight->p == distrib.PMF(intr, sampledLight->light) for 100 points p in [-1, 2)^3 (lightsamplers_test.cpp:276-350)."""
    _pdf_method(load_package(), sampler)   # (device == oracle for the samplers: test_gpu_parity.py::test_light_samplers_vs_oracle)


# ---- Spectrum, Blackbody (util/spectrum_test.cpp:19-48) ---------------------------------------------------------------------------
# The oracle's Blackbody() / BlackbodySpectrum are a restatement of util/spectrum.h:83-94, :568-588 whose goldens were transcribed on
# the reference's types; this is the reference's OWN known-answer test for the function: Planck's law at four
# (lambda, T) pairs against an independent calculator, and Wien's displacement law.
BLACKBODY_KNOWN = [(483, 6000, 3.1849e13), (600, 6000, 2.86772e13), (500, 3700, 1.59845e12), (600, 4500, 7.46497e12)]   # :27-32 {lambda, T, expected radiance}
BLACKBODY_WIEN_T = [2700, 3000, 4500, 5600, 6000]                                                                          # :42


def test_spectrum_blackbody():
    lib = oracle_lib.load()
    B = lambda lam, T: float(lib.oracle_blackbody_radiance(C.c_float(lam), C.c_float(T)))
    for lam, T, expected in BLACKBODY_KNOWN:
        assert abs(B(lam, T) - expected) / expected < .001, (lam, T, B(lam, T), expected)   # :36 EXPECT_LT(err(Blackbody(lambda, T), LeExpected), .001)
    for T in BLACKBODY_WIEN_T:
        lmax = f32(f32(2.8977721e-3) / f32(T)) * f32(1e9)                                   # :43 (Float arithmetic)
        lo, hi = f32(.99 * float(lmax)), f32(1.01 * float(lmax))                            # :44 Float(.99 * lambdaMax) -- the product in double
        assert B(lo, T) < B(lmax, T) > B(hi, T), T                                          # :45-46


def _u_for_wavelength(lam):
    """SampleVisibleWavelengths inverted (util/sampling.h:169-171: lambda = 538 - 138.888889 atanh(0.85691062 - 1.82750197 u))."""
    return (0.85691062 - math.tanh((538.0 - lam) / 138.888889)) / 1.82750197


def _planck64(lam_nm, T):
    c, h, kb = 299792458.0, 6.62606957e-34, 1.3806488e-23
    l = lam_nm * 1e-9
    return 2 * h * c * c / (l ** 5 * (math.exp(h * c / (l * kb * T)) - 1))


@pytest.mark.gpu
def test_spectrum_blackbody_on_device(gpu_pkg):
    """The device has no raw Blackbody() entry: vspg_blackbody_batch gives SampleVisible(u)'s wavelengths and BlackbodySpectrum(T).Sample
    at them -- Blackbody(lambda, T) / Blackbody(lambdaMax, T).  u is chosen so that the first wavelength IS the scenario's (to 1e-3 nm;
    the tolerance of the known answers is 1e-3 relative, d ln B / d ln lambda is O(1)), the normalisation is undone with Planck's law
    in double.  Wien's law: the three wavelengths around the peak as three entries; only the peaks the visible sampler reaches
    (360-830 nm: T >= 4500 K)."""
    P = gpu_pkg
    g = P.Renderer(P.fog_box_scene(16, 16), P.app_f_params(), 16, 16)
    u = np.float32([_u_for_wavelength(l) for l, _, _ in BLACKBODY_KNOWN])
    T = np.float32([t for _, t, _ in BLACKBODY_KNOWN])
    out = g.blackbody_batch(u, T)
    for (lam, t, expected), row in zip(BLACKBODY_KNOWN, out):
        assert abs(float(row[0]) - lam) < 2e-3, (lam, row[0])
        lmax = 2.8977721e-3 / t * 1e9
        val = float(row[3]) * _planck64(lmax, t)
        assert abs(val - expected) / expected < .0015, (lam, t, val, expected)   # .001 of the reference's test + the FastExp normalisation's own error
    for t in BLACKBODY_WIEN_T:
        lmax = 2.8977721e-3 / t * 1e9
        if not 365 < lmax < 820:
            continue
        lams = [.99 * lmax, lmax, 1.01 * lmax]
        row = g.blackbody_batch(np.float32([_u_for_wavelength(l) for l in lams]), np.float32([t] * 3))
        assert all(abs(float(row[i, 0]) - lams[i]) < 2e-3 for i in range(3))
        assert row[0, 3] < row[1, 3] > row[2, 3], (t, row[:, 3])
        assert abs(float(row[1, 3]) - 1) < 1e-5   # normalised by the value at Wien's peak (spectrum.h:577-580)
    # and bit for bit the oracle's, at these inputs
    lib = oracle_lib.load()
    o6 = (C.c_float * 6)()
    for i in range(len(u)):
        lib.oracle_blackbody(float(u[i]), float(T[i]), o6)
        assert np.array_equal(np.float32(list(o6)).view(np.uint32), out[i].view(np.uint32))
    g.close()


# ---- SampleDiscrete, Basics (util/sampling_test.cpp:26-42): the two-weight cases -- the delta-tracking callback's event choice ------
def test_sample_discrete_basics():
    lib = oracle_lib.load()
    sd = lambda w0, w1, u: int(lib.oracle_sample_discrete2(C.c_float(w0), C.c_float(w1), C.c_float(u)))
    assert sd(0.5, 0.5, 0.) == 0        # :32
    assert sd(0.5, 0.5, 0.499) == 0     # :35
    assert sd(0.5, 0.5, 0.5) == 1       # :39

"""Oracle (CPU restatement) vs the committed golden vectors generated from the reference's own
headers (tests/golden/primitives.json, made by oracle/make_golden.py) and vs the known-answer
fixtures of the reference's unit tests.  Integer work is bit-exact; float results are bit-exact
too where they depend only on IEEE ops (FastExp, HG, CoordinateSystem, OffsetRayOrigin), and
exact on the generating host for libm-dependent ones (logf / sinf / cosf)."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from conftest import ROOT

G = json.load(open(os.path.join(ROOT, "tests", "golden", "primitives.json")))
fh = float.fromhex


def f3(P, v):
    return P.f3(*v)


def bits(x):
    return np.float32(x).view(np.uint32)


def same(a, b):
    return bits(a) == bits(b)


def test_hash_float(oracle):
    for f, h in G["hash_float"]:
        assert oracle.oracle_hash_float(fh(f)) == int(h, 16)
    # SURVEY.md App. D.3 known answer
    assert oracle.oracle_hash_float(0.25) == 0x3EB9F34EC52A56A0


def test_hash_pixel_seed_point3_mixbits(oracle):
    for x, y, s, h in G["hash_pixel_seed"]:
        assert oracle.oracle_hash_pixel_seed(x, y, s) == int(h, 16)
    for x, y, z, h in G["hash_point3"]:
        assert oracle.oracle_hash_point3(fh(x), fh(y), fh(z)) == int(h, 16)
    for v, m in G["mix_bits"]:
        assert oracle.oracle_mix_bits(int(v, 16)) == int(m, 16)


def test_murmur_reference_unit_test_properties(oracle):
    # src/pbrt/util/hash_test.cpp:13-55: distinct inputs hash differently, buffers hash by content
    buf = bytes(range(64))
    hs = {oracle.oracle_murmur64a(buf[:n], n, 0) for n in range(1, 65)}
    assert len(hs) == 64
    assert oracle.oracle_murmur64a(b"", 0, 0) == 0  # seed ^ (0*m) mixed: h=0 stays 0


def test_rng(oracle):
    for c in G["rng"]:
        n = len(c["u32"])
        u = (C.c_uint32 * n)()
        f = (C.c_float * n)()
        oracle.oracle_rng_seq(int(c["seq"], 16), int(c["seed"], 16), c["has_seed"], c["advance"], n, u, f)
        assert list(u) == c["u32"]
        assert all(same(a, fh(b)) for a, b in zip(f, c["f"]))
    # D.3: RNG(Hash(0.25f),Hash(0.75f)).Uniform<float>() = 0x1.62bf6p-2
    u = (C.c_uint32 * 1)()
    f = (C.c_float * 1)()
    oracle.oracle_rng_seq(oracle.oracle_hash_float(0.25), oracle.oracle_hash_float(0.75), 1, 0, 1, u, f)
    assert f[0] == fh("0x1.62bf6p-2")


def test_rng_reference_unit_test_fixtures(oracle):
    # src/pbrt/util/rng_test.cpp: Advance(n) equals n draws; Advance(-n) rewinds (:16-60)
    n = 24
    a = (C.c_uint32 * n)()
    oracle.oracle_rng_seq(1234, 6502, 1, 0, n, a, None)
    for k in (1, 5, 17):
        b = (C.c_uint32 * (n - k))()
        oracle.oracle_rng_seq(1234, 6502, 1, k, n - k, b, None)
        assert list(b) == list(a)[k:]
    # floats in [0,1)
    f = (C.c_float * 1000)()
    oracle.oracle_rng_seq(9, 9, 1, 0, 1000, None, f)
    arr = np.array(f)
    assert arr.min() >= 0 and arr.max() < 1


def test_independent_sampler(oracle):
    for c in G["independent_sampler"]:
        n = len(c["f"])
        out = (C.c_float * n)()
        oracle.oracle_independent_sampler(c["px"], c["py"], c["seed"], c["sample"], n, out)
        assert all(same(a, fh(b)) for a, b in zip(out, c["f"]))


def test_fast_exp(oracle):
    for x, y in G["fast_exp"]:
        assert same(oracle.oracle_fast_exp(fh(x)), fh(y)), (x, y)
    assert oracle.oracle_fast_exp(-1.5) == fh("0x1.c90078p-3")  # D.3
    # src/pbrt/util/math_test.cpp:365-376: rel. error <= 3e-4 on [-20,20]
    xs = np.linspace(-20, 20, 100, dtype=np.float32)
    for x in xs:
        e = oracle.oracle_fast_exp(float(x))
        assert abs(e - np.exp(np.float64(x))) / np.exp(np.float64(x)) < 3e-4


def test_sample_exponential_discrete(oracle):
    for u, a, t in G["sample_exponential"]:
        assert same(oracle.oracle_sample_exponential(fh(u), fh(a)), fh(t))
    assert oracle.oracle_sample_exponential(0.3, 2.0) == fh("0x1.6d3c34p-3")  # D.3
    for w0, w1, u, k in G["sample_discrete2"]:
        assert oracle.oracle_sample_discrete2(fh(w0), fh(w1), fh(u)) == k
    assert oracle.oracle_sample_discrete2(0.3, 0.7, 0.5) == 1  # D.3


def test_henyey_greenstein(oracle, pkg):
    for c, g, p in G["henyey_greenstein"]:
        assert same(oracle.oracle_henyey_greenstein(fh(c), fh(g)), fh(p))
    for row in G["sample_henyey_greenstein"]:
        wo, g, u0, u1, wi, pdf = [fh(v) for v in row[0:3]], fh(row[3]), fh(row[4]), fh(row[5]), [fh(v) for v in row[6:9]], fh(row[9])
        out = pkg.f3()
        opdf = C.c_float()
        oracle.oracle_sample_henyey_greenstein(f3(pkg, wo), g, u0, u1, out, C.byref(opdf))
        assert all(same(a, b) for a, b in zip(out, wi)) and same(opdf.value, pdf)


def test_hg_reference_unit_test_properties(oracle, pkg):
    # src/pbrt/media_test.cpp:15-98: sampling pdf == p(); normalisation; mean cosine == g
    rng = np.random.default_rng(5)
    for g in (-0.75, -0.3, 0.0, 0.25, 0.8):
        us = rng.random((4000, 2)).astype(np.float32)
        cos_sum = 0.0
        wo = (0.0, 0.0, 1.0)
        for u0, u1 in us:
            out = pkg.f3()
            pdf = C.c_float()
            oracle.oracle_sample_henyey_greenstein(f3(pkg, wo), g, float(u0), float(u1), out, C.byref(pdf))
            c = out[2]  # dot(wo, wi)
            assert abs(pdf.value - oracle.oracle_henyey_greenstein(c, g)) <= 1e-4 * max(1.0, pdf.value)
            cos_sum += c
        # pbrt convention: wo and wi both point away, so E[dot(wo,wi)] = -g
        assert abs(cos_sum / len(us) + g) < 0.03
        # normalisation by quadrature over cos(theta)
        cs = np.linspace(-1, 1, 8001)
        ps = np.array([oracle.oracle_henyey_greenstein(float(c), g) for c in cs])
        integ = np.trapezoid(ps, cs) * 2 * np.pi
        assert abs(integ - 1) < 2e-3


def test_direction_sampling(oracle, pkg):
    for u0, u1, x, y, z in G["sample_uniform_sphere"]:
        out = pkg.f3()
        oracle.oracle_sample_uniform_sphere(fh(u0), fh(u1), out)
        assert all(same(a, fh(b)) for a, b in zip(out, (x, y, z)))
    for u0, u1, x, y, z in G["sample_cosine_hemisphere"]:
        out = pkg.f3()
        oracle.oracle_sample_cosine_hemisphere(fh(u0), fh(u1), out)
        assert all(same(a, fh(b)) for a, b in zip(out, (x, y, z)))
    for row in G["coordinate_system"]:
        v = [fh(t) for t in row]
        a, b = pkg.f3(), pkg.f3()
        oracle.oracle_coordinate_system(f3(pkg, v[0:3]), a, b)
        assert all(same(p, q) for p, q in zip(list(a) + list(b), v[3:9]))


def test_offset_ray_origin(oracle, pkg):
    for row in G["offset_ray_origin"]:
        v = [fh(t) for t in row]
        out = pkg.f3()
        oracle.oracle_offset_ray_origin(f3(pkg, v[0:3]), f3(pkg, v[3:6]), f3(pkg, v[6:9]), f3(pkg, v[9:12]), out)
        assert all(same(a, b) for a, b in zip(out, v[12:15])), row


def test_bounds3_offset_and_intersect(oracle, pkg):
    """Bounds3f::Offset / IntersectP(o, d, tMax, &t0, &t1) of the REFERENCE (vecmath.h:1323-1332, 1547-1571; the
    ray-vs-medium-bounds step of GridMedium::SampleRay): the oracle reproduces hit flag, t0, t1 and the offset bit for bit."""
    bmin, bmax = pkg.f3(-0.8, -0.8, -0.5), pkg.f3(0.8, 0.7, 0.9)
    hits = 0
    for row in G["bounds3"]:
        o, d = [fh(t) for t in row[0:3]], [fh(t) for t in row[3:6]]
        tMax = fh(row[6]) if row[6] != "inf" else float("inf")
        t01 = (C.c_float * 2)()
        off = pkg.f3()
        hit = oracle.oracle_bounds3(bmin, bmax, f3(pkg, o), f3(pkg, d), C.c_float(tMax), C.byref(t01), off)
        assert hit == row[7], row
        if hit:
            hits += 1
            assert same(t01[0], fh(row[8])) and same(t01[1], fh(row[9])), row
        assert all(same(a, fh(b)) for a, b in zip(off, row[10:13])), row
    assert 5 < hits < len(G["bounds3"]) - 5


def test_frame_from_xz(oracle, pkg):
    """Frame::FromXZ / ToLocal / FromLocal of the REFERENCE (vecmath.h:1850-1920): the BSDF's shading frame."""
    for row in G["frame_xz"]:
        v = [fh(t) for t in row]
        y, l, w = pkg.f3(), pkg.f3(), pkg.f3()
        oracle.oracle_frame_xz(f3(pkg, v[0:3]), f3(pkg, v[3:6]), f3(pkg, v[6:9]), y, l, w)
        assert all(same(a, b) for a, b in zip(list(y) + list(l) + list(w), v[9:18])), row


def test_spawn_ray_to(oracle, pkg):
    """SpawnRayTo(Point3fi, Normal3f, time, Point3fi, Normal3f) of the REFERENCE (ray.h:103-108): origin and direction of the
    NEE shadow ray -- the two values that seed its RNG -- bit for bit."""
    for row in G["spawn_ray_to"]:
        v = [fh(t) for t in row]
        o, d = pkg.f3(), pkg.f3()
        oracle.oracle_spawn_ray_to(f3(pkg, v[0:3]), f3(pkg, v[3:6]), f3(pkg, v[6:9]), f3(pkg, v[9:12]), f3(pkg, v[12:15]),
                                   f3(pkg, v[15:18]), o, d)
        assert all(same(a, b) for a, b in zip(list(o) + list(d), v[18:24])), row


def test_apply_inverse_identity(oracle, pkg):
    """Transform::ApplyInverse(Ray, &tMax) of the REFERENCE with an identity matrix (transform.h:416-429): the origin shift
    by the conservative error bound and the shortened tMax -- first step of GridMedium / NanoVDBMedium::SampleRay."""
    moved = 0
    for row in G["apply_inverse_identity"]:
        v = [fh(t) if t not in ("inf", "-inf") else float(t) for t in row]
        out = pkg.f3()
        t = C.c_float()
        oracle.oracle_apply_inverse_identity(f3(pkg, v[0:3]), f3(pkg, v[3:6]), C.c_float(v[6]), out, C.byref(t))
        assert all(same(a, b) for a, b in zip(out, v[7:10])), row
        assert same(t.value, v[10]) or (np.isinf(v[10]) and np.isinf(t.value)), row
        moved += any(not same(a, b) for a, b in zip(out, v[0:3]))
    assert moved > 20  # the shift is real


def test_apply_inverse_general_transforms(oracle, pkg):
    """Transform::ApplyInverse(Ray, &tMax) and ApplyInverse(Vector3f) of the REFERENCE for translations, scales, rotations and
    their products (transform.h:401-429, transform.cpp:263-303) -- renderFromMedium of a placed grid medium (media.h:322, :354).
    Each case carries the reference's own m and mInv.  The library's vspg_transform_inverse (used by callers that only have
    m) must agree with the reference's inverse to rounding."""
    lib = pkg.load()
    moved = 0
    for row in G["apply_inverse_xform"]:
        v = [fh(t) if t not in ("inf", "-inf") else float(t) for t in row]
        m, minv = v[0:16], v[16:32]
        o, d, tMax = v[32:35], v[35:38], v[38]
        want_o, want_d, want_t, want_v = v[39:42], v[42:45], v[45], v[46:49]
        MI = (C.c_float * 16)(*minv)
        out_o, out_d = pkg.f3(), pkg.f3()
        t = C.c_float()
        oracle.oracle_apply_inverse_ray(MI, f3(pkg, o), f3(pkg, d), C.c_float(tMax), out_o, out_d, C.byref(t))
        assert all(same(a, b) for a, b in zip(out_o, want_o)), row
        assert all(same(a, b) for a, b in zip(out_d, want_d)) and all(same(a, b) for a, b in zip(out_d, want_v)), row
        assert same(t.value, want_t) or (np.isinf(want_t) and np.isinf(t.value)), row
        moved += any(not same(a, b) for a, b in zip(out_o, o))
        inv = (C.c_float * 16)()
        assert lib.vspg_transform_inverse((C.c_float * 16)(*m), inv) == 0
        assert np.allclose(list(inv), minv, rtol=2e-6, atol=2e-6), row
    assert moved > 40


def test_channel_idx():
    # spectrum.h:380-384: channelIdx = min(floor(3u), 2); the oracle inlines it in
    # evaluate_pixel_sample -- check the formula against the reference's outputs
    for u, ch in G["channel_idx"]:
        assert min(int(np.floor(np.float32(fh(u)) * np.float32(3))), 2) == ch


def _fhx(t):
    return float("inf") if t == "inf" else (float("-inf") if t == "-inf" else fh(t))


def test_interval_arithmetic(oracle, pkg):
    """Interval +, -, *, /, Sqr, Sqrt(Abs()), Float * Interval of the REFERENCE (util/math.h:818-1010 over the CPU rounding
    helpers of util/float.h:199-297) -- the arithmetic under Sphere::BasicIntersect: the oracle reproduces both bounds bit for bit."""
    out = (C.c_float * 2)()
    n = 0
    for row in G["interval_ops"]:
        v = [_fhx(t) for t in row]
        alo, ahi, blo, bhi, f = v[:5]
        for op in range(7):
            oracle.oracle_interval_op(op, alo, ahi, blo, bhi, f, out)
            want = v[5 + 2 * op: 7 + 2 * op]
            for a, b in zip(out, want):
                assert same(a, b) or (np.isnan(a) and np.isnan(b)), (op, row)
            n += 1
    assert n == 96 * 7


def test_sphere_against_the_reference_types(oracle, pkg):
    """Shape "sphere": Sphere::BasicIntersect (shapes.h:147-229) and InteractionFromIntersection (:237-284) + the transform of
    the interaction to render space -- evaluated by the golden generator on the reference's own Interval / Point3fi / Vector3fi /
    Transform / SurfaceInteraction classes (oracle/ref_primitives.cpp: shapes.h itself does not link here) -- against the oracle's
    C restatement: hit or miss, tHit, the object-space point, the interaction point's BOUNDS, the normal, the shading tangent,
    the origin SkipIntersection spawns the next ray from and the side GetMedium picks: all bit for bit, over identity /
    translated / scaled / rotated / mirrored spheres, rays from inside and outside, finite tMax, unnormalised directions."""
    P = pkg
    hits = 0
    for row in G["sphere"]:
        v = [_fhx(t) if isinstance(t, str) else t for t in row]
        sp = P.VspgSphere()
        sp.render_from_object[:] = v[0:16]
        sp.object_from_render[:] = v[16:32]
        sp.radius, sp.reverse_orientation = v[32], int(v[33])
        o, d, tMax, want_hit = v[34:37], v[37:40], v[40], int(v[41])
        hit, t = C.c_int(), C.c_float()
        plo, phi, n, du = P.f3(), P.f3(), P.f3(), P.f3()
        oracle.oracle_sphere_intersect(C.byref(sp), P.f3(*o), P.f3(*d), C.c_float(tMax), C.byref(hit), C.byref(t), plo, phi, n, du)
        assert hit.value == want_hit, row
        if not want_hit:
            continue
        hits += 1
        assert same(t.value, v[42])
        for got, want in ((plo, v[46:49]), (phi, v[49:52]), (n, v[52:55]), (du, v[55:58])):
            assert all(same(a, b) for a, b in zip(got, want)), (list(got), want)
        h2, pobj, so, side = C.c_int(), P.f3(), P.f3(), C.c_int()
        oracle.oracle_sphere_skip(C.byref(sp), P.f3(*o), P.f3(*d), C.c_float(tMax), C.byref(h2), pobj, so, C.byref(side))
        assert h2.value == 1 and all(same(a, b) for a, b in zip(pobj, v[43:46])) and all(same(a, b) for a, b in zip(so, v[58:61]))
        assert side.value == int(v[61])
    assert hits >= 90


def test_blackbody_emission_against_the_reference(oracle, pkg):
    """Volume emission of a temperature grid in the RGB build: the three wavelengths are the REFERENCE's SampledWavelengths::SampleVisible
    (util/spectrum.h:369-386 over SampleVisibleWavelengths, util/sampling.h:169-171 -- the host's atanhf), the emitted values are the
    statement sequence of Blackbody() / BlackbodySpectrum (spectrum.h:83-94, :568-588) evaluated by the golden generator with the
    reference's own FastExp and Pow<5> (Blackbody() itself does not link there: its CHECK pulls LogFatal).  Bit for bit, from the
    100 K threshold of Le() to 80000 K, and at lu = 0, 1 - ulp, 1/3, 2/3 where `up` wraps."""
    out = (C.c_float * 6)()
    lit = 0
    for row in G["blackbody"]:
        v = [fh(t) for t in row]
        oracle.oracle_blackbody(v[0], v[1], out)
        for a, b in zip(out, v[2:8]):
            assert same(a, b), row
        lit += any(x > 0 for x in v[5:8])
    assert len(G["blackbody"]) == 192 and lit > 120

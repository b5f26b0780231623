// ref_primitives.cpp -- golden-vector generator for the bit-exact primitive layer.
//
// TEST INFRASTRUCTURE.  This driver #includes the reference's OWN headers where they lie
// under /root/reference (nothing is copied, no stand-in headers are written) and links the
// reference's src/pbrt/util/sampling.cpp compiled in place.  It only builds in the container
// that has /root/reference; its binary lives in oracle/_ref/ (git-ignored) and its output is
// committed as tests/golden/primitives.json by oracle/make_golden.py.
//
// Covered (SURVEY.md 8a rows a1-a4 + helpers): Hash / MurmurHash64A / MixBits, RNG (PCG32)
// incl. SetSequence / Advance / Uniform<float>, IndependentSampler, FastExp, SampleExponential,
// SampleDiscrete, HenyeyGreenstein, SampleHenyeyGreenstein, SampleUniformSphere,
// SampleCosineHemisphere, CoordinateSystem, OffsetRayOrigin, SampledWavelengths::ChannelIdx.
// Bounds3f::Offset / IntersectP and Transform::ApplyInverse(Ray, &tMax) (the first two steps of GridMedium::SampleRay;
// links the reference's util/transform.cpp and util/math.cpp compiled in place).
// NOT buildable without the absent third-party headers (nanovdb, openpgl, double-conversion ...): media.h,
// media_sampleTMaj.h, the integrator; SampledGrid's constructors CHECK through util/print.cpp, which needs
// double-conversion -- see DESIGN.md.
#include <pbrt/pbrt.h>
#include <pbrt/interaction.h>
#include <pbrt/ray.h>
#include <pbrt/samplers.h>
#include <pbrt/util/hash.h>
#include <pbrt/util/math.h>
#include <pbrt/util/rng.h>
#include <pbrt/util/sampling.h>
#include <pbrt/util/scattering.h>
#include <pbrt/util/spectrum.h>
#include <pbrt/util/transform.h>
#include <pbrt/util/vecmath.h>

#include <cstdio>
#include <vector>

using namespace pbrt;

static void pf(float v) { printf("\"%a\"", v); }
static void sep(bool &first) {
    if (!first) printf(",");
    first = false;
}

int main() {
    RNG gen(12345, 678);  // input generator
    auto U = [&]() { return gen.Uniform<Float>(); };
    printf("{\n");

    // ---- Hash(float) ----
    {
        printf("\"hash_float\": [");
        bool first = true;
        std::vector<float> in = {0.25f, 0.75f, 0.f, -0.f, 1.f, 0.5f, 1e-20f, 123456.f};
        for (int i = 0; i < 56; ++i) in.push_back(U());
        for (float f : in) {
            sep(first);
            printf("[");
            pf(f);
            printf(",\"%016llx\"]", (unsigned long long)Hash(f));
        }
        printf("],\n");
    }
    // ---- Hash(Point2i, int) and Hash(Point3f) and MixBits ----
    {
        printf("\"hash_pixel_seed\": [");
        bool first = true;
        int px[] = {0, 1, 0, 1919, 511, 37, 1000, 77};
        int py[] = {0, 0, 1, 1079, 511, 911, 3, 77};
        int sd[] = {0, 0, 0, 0, 7, 123456, -1, 2147483647};
        for (int i = 0; i < 8; ++i) {
            sep(first);
            printf("[%d,%d,%d,\"%016llx\"]", px[i], py[i], sd[i],
                   (unsigned long long)Hash(Point2i(px[i], py[i]), sd[i]));
        }
        printf("],\n\"hash_point3\": [");
        first = true;
        for (int i = 0; i < 32; ++i) {
            Point3f p(2 * U() - 1, 2 * U() - 1, 2 * U() - 1);
            sep(first);
            printf("[");
            pf(p.x); printf(","); pf(p.y); printf(","); pf(p.z);
            printf(",\"%016llx\"]", (unsigned long long)Hash(p));
        }
        printf("],\n\"mix_bits\": [");
        first = true;
        uint64_t vals[] = {0ull, 1ull, 0xdeadbeefcafef00dull, 0xffffffffffffffffull, 0x3eb9f34ec52a56a0ull};
        for (uint64_t v : vals) {
            sep(first);
            printf("[\"%016llx\",\"%016llx\"]", (unsigned long long)v, (unsigned long long)MixBits(v));
        }
        printf("],\n");
    }
    // ---- RNG ----
    {
        printf("\"rng\": [");
        bool first = true;
        struct C { uint64_t seq, seed; int has_seed; int64_t adv; };
        C cases[] = {{0, 0, 1, 0}, {Hash(0.25f), Hash(0.75f), 1, 0}, {7, 0, 0, 0}, {42, 99, 1, 65536},
                     {Hash(Point2i(5, 9), 0), 0, 0, 3 * 65536ull}, {0xffffffffffffffffull, 1, 1, 1234567}};
        for (auto &c : cases) {
            RNG a, b;
            if (c.has_seed) { a.SetSequence(c.seq, c.seed); } else { a.SetSequence(c.seq); }
            a.Advance(c.adv);
            b = a;
            sep(first);
            printf("{\"seq\":\"%016llx\",\"seed\":\"%016llx\",\"has_seed\":%d,\"advance\":%lld,\"u32\":[",
                   (unsigned long long)c.seq, (unsigned long long)c.seed, c.has_seed, (long long)c.adv);
            for (int i = 0; i < 16; ++i) printf("%s%u", i ? "," : "", a.Uniform<uint32_t>());
            printf("],\"f\":[");
            for (int i = 0; i < 16; ++i) { if (i) printf(","); pf(b.Uniform<float>()); }
            printf("]}");
        }
        printf("],\n");
    }
    // ---- IndependentSampler ----
    {
        printf("\"independent_sampler\": [");
        bool first = true;
        int cs[][4] = {{0, 0, 0, 0}, {17, 33, 0, 5}, {1919, 1079, 0, 255}, {3, 4, 11, 1000}};
        for (auto &c : cs) {
            IndependentSampler s(256, c[2]);
            s.StartPixelSample(Point2i(c[0], c[1]), c[3], 0);
            sep(first);
            printf("{\"px\":%d,\"py\":%d,\"seed\":%d,\"sample\":%d,\"f\":[", c[0], c[1], c[2], c[3]);
            for (int i = 0; i < 12; ++i) { if (i) printf(","); pf(s.Get1D()); }
            printf("]}");
        }
        printf("],\n");
    }
    // ---- FastExp ----
    {
        printf("\"fast_exp\": [");
        bool first = true;
        std::vector<float> in = {-1.5f, 0.f, -0.f, 1.f, -1.f, -87.f, -88.f, -100.f, 88.f, 89.f, -1e-8f, 1e-8f, -20.f, 20.f};
        for (int i = 0; i < 200; ++i) in.push_back(-12.f * U());
        for (int i = 0; i < 50; ++i) in.push_back(40.f * U() - 20.f);
        for (float f : in) { sep(first); printf("["); pf(f); printf(","); pf(FastExp(f)); printf("]"); }
        printf("],\n");
    }
    // ---- SampleExponential ----
    {
        printf("\"sample_exponential\": [");
        bool first = true;
        std::vector<std::pair<float, float>> in = {{0.3f, 2.f}, {0.f, 1.f}, {OneMinusEpsilon, 0.5f}};
        for (int i = 0; i < 64; ++i) in.push_back({U(), 0.01f + 10.f * U()});
        for (auto &c : in) {
            sep(first); printf("["); pf(c.first); printf(","); pf(c.second); printf(",");
            pf(SampleExponential(c.first, c.second)); printf("]");
        }
        printf("],\n");
    }
    // ---- SampleDiscrete (two weights) ----
    {
        printf("\"sample_discrete2\": [");
        bool first = true;
        struct C { float w0, w1, u; };
        std::vector<C> in = {{0.3f, 0.7f, 0.5f}, {1.f, 0.f, 0.f}, {1.f, 0.f, OneMinusEpsilon}, {0.5f, 0.5f, 0.5f}, {0.f, 1.f, 0.f}};
        for (int i = 0; i < 64; ++i) { float w = U(); in.push_back({w, std::max<Float>(0, 1 - w), U()}); }
        for (auto &c : in) {
            Float w[2] = {c.w0, c.w1};
            sep(first); printf("["); pf(c.w0); printf(","); pf(c.w1); printf(","); pf(c.u);
            printf(",%d]", SampleDiscrete(w, c.u));
        }
        printf("],\n");
    }
    // ---- HenyeyGreenstein / SampleHenyeyGreenstein ----
    {
        printf("\"henyey_greenstein\": [");
        bool first = true;
        float gs[] = {0.f, 0.5f, -0.5f, 0.877f, 0.999f, -0.999f, 0.0005f};
        for (float g : gs)
            for (int i = 0; i < 8; ++i) {
                float c = 2 * U() - 1;
                sep(first); printf("["); pf(c); printf(","); pf(g); printf(","); pf(HenyeyGreenstein(c, g)); printf("]");
            }
        printf("],\n\"sample_henyey_greenstein\": [");
        first = true;
        for (float g : gs)
            for (int i = 0; i < 8; ++i) {
                Vector3f wo = Normalize(Vector3f(2 * U() - 1, 2 * U() - 1, 2 * U() - 1));
                Point2f u(U(), U());
                Float pdf;
                Vector3f wi = SampleHenyeyGreenstein(wo, g, u, &pdf);
                sep(first); printf("[");
                pf(wo.x); printf(","); pf(wo.y); printf(","); pf(wo.z); printf(","); pf(g); printf(",");
                pf(u[0]); printf(","); pf(u[1]); printf(",");
                pf(wi.x); printf(","); pf(wi.y); printf(","); pf(wi.z); printf(","); pf(pdf); printf("]");
            }
        printf("],\n");
    }
    // ---- SampleUniformSphere / SampleCosineHemisphere / CoordinateSystem ----
    {
        printf("\"sample_uniform_sphere\": [");
        bool first = true;
        for (int i = 0; i < 32; ++i) {
            Point2f u(U(), U());
            Vector3f v = SampleUniformSphere(u);
            sep(first); printf("["); pf(u[0]); printf(","); pf(u[1]); printf(","); pf(v.x); printf(","); pf(v.y); printf(","); pf(v.z); printf("]");
        }
        printf("],\n\"sample_cosine_hemisphere\": [");
        first = true;
        std::vector<Point2f> us = {{0.5f, 0.5f}, {0.f, 0.f}, {0.25f, 0.75f}};
        for (int i = 0; i < 32; ++i) us.push_back({U(), U()});
        for (auto u : us) {
            Vector3f v = SampleCosineHemisphere(u);
            sep(first); printf("["); pf(u[0]); printf(","); pf(u[1]); printf(","); pf(v.x); printf(","); pf(v.y); printf(","); pf(v.z); printf("]");
        }
        printf("],\n\"coordinate_system\": [");
        first = true;
        for (int i = 0; i < 24; ++i) {
            Vector3f v = Normalize(Vector3f(2 * U() - 1, 2 * U() - 1, 2 * U() - 1)), a, b;
            CoordinateSystem(v, &a, &b);
            sep(first); printf("[");
            pf(v.x); printf(","); pf(v.y); printf(","); pf(v.z); printf(",");
            pf(a.x); printf(","); pf(a.y); printf(","); pf(a.z); printf(",");
            pf(b.x); printf(","); pf(b.y); printf(","); pf(b.z); printf("]");
        }
        printf("],\n");
    }
    // ---- OffsetRayOrigin ----
    {
        printf("\"offset_ray_origin\": [");
        bool first = true;
        for (int i = 0; i < 32; ++i) {
            Point3f p(2 * U() - 1, 2 * U() - 1, 2 * U() - 1);
            Vector3f e(1e-6f * U(), 1e-6f * U(), 1e-6f * U());
            Normal3f n = (i % 4 == 0) ? Normal3f(0, 0, 0)
                                      : Normal3f(Normalize(Vector3f(2 * U() - 1, 2 * U() - 1, 2 * U() - 1)));
            if (i % 4 == 1) n = Normal3f(0, 1, 0);
            Vector3f w(2 * U() - 1, 2 * U() - 1, 2 * U() - 1);
            Point3f o = OffsetRayOrigin(Point3fi(p, e), n, w);
            sep(first); printf("[");
            pf(p.x); printf(","); pf(p.y); printf(","); pf(p.z); printf(",");
            pf(e.x); printf(","); pf(e.y); printf(","); pf(e.z); printf(",");
            pf(n.x); printf(","); pf(n.y); printf(","); pf(n.z); printf(",");
            pf(w.x); printf(","); pf(w.y); printf(","); pf(w.z); printf(",");
            pf(o.x); printf(","); pf(o.y); printf(","); pf(o.z); printf("]");
        }
        printf("],\n");
    }
    // ---- hero channel of SampledWavelengths::SampleVisible (RGB build) ----
    {
        printf("\"channel_idx\": [");
        bool first = true;
        float us[] = {0.f, 0.1f, 0.333f, 0.3333333f, 0.35f, 0.6f, 0.6666667f, 0.85f, OneMinusEpsilon};
        for (float u : us) {
            SampledWavelengths swl = SampledWavelengths::SampleVisible(u);
            sep(first); printf("["); pf(u); printf(",%d]", swl.ChannelIdx());
        }
        printf("],\n");
    }
    // ---- Bounds3f::Offset / IntersectP(o, d, tMax, &t0, &t1) (util/vecmath.h:1323-1332, 1547-1571) ----
    {
        printf("\"bounds3\": [");
        bool first = true;
        Bounds3f b(Point3f(-0.8f, -0.8f, -0.5f), Point3f(0.8f, 0.7f, 0.9f));
        for (int i = 0; i < 48; ++i) {
            Point3f o(3 * U() - 1.5f, 3 * U() - 1.5f, 3 * U() - 1.5f);
            Vector3f d(2 * U() - 1, 2 * U() - 1, 2 * U() - 1);
            if (i % 8 == 0) d.x = 0;
            if (i % 8 == 1) d = Vector3f(0, -0.f, 1);
            Float tMax = (i % 3 == 0) ? Infinity : 4 * U();
            Float t0 = -1, t1 = -1;
            bool hit = b.IntersectP(o, d, tMax, &t0, &t1);
            Vector3f off = b.Offset(o);
            sep(first); printf("[");
            pf(o.x); printf(","); pf(o.y); printf(","); pf(o.z); printf(",");
            pf(d.x); printf(","); pf(d.y); printf(","); pf(d.z); printf(","); pf(tMax); printf(",");
            printf("%d,", hit ? 1 : 0); pf(hit ? t0 : 0.f); printf(","); pf(hit ? t1 : 0.f); printf(",");
            pf(off.x); printf(","); pf(off.y); printf(","); pf(off.z); printf("]");
        }
        printf("],\n");
    }
    // ---- Transform::ApplyInverse(const Ray &, Float *tMax), identity matrix (util/transform.h:416-429,
    //      util/transform.cpp:263-303): the first step of GridMedium / NanoVDBMedium::SampleRay ----
    {
        printf("\"apply_inverse_identity\": [");
        bool first = true;
        Transform T;
        for (int i = 0; i < 48; ++i) {
            Point3f o(4 * U() - 2, 4 * U() - 2, 4 * U() - 2);
            Vector3f d(2 * U() - 1, 2 * U() - 1, 2 * U() - 1);
            if (i % 8 == 0) d = d * 1e-3f;
            if (i % 8 == 1) o = Point3f(0, 0, 0);
            if (i % 8 == 2) d = Vector3f(0, 0, 0);
            Float tMax = (i % 3 == 0) ? Infinity : 4 * U();
            Float t = tMax;
            Ray q = T.ApplyInverse(Ray(o, d), &t);
            sep(first); printf("[");
            pf(o.x); printf(","); pf(o.y); printf(","); pf(o.z); printf(",");
            pf(d.x); printf(","); pf(d.y); printf(","); pf(d.z); printf(","); pf(tMax); printf(",");
            pf(q.o.x); printf(","); pf(q.o.y); printf(","); pf(q.o.z); printf(","); pf(t); printf("]");
        }
        printf("],\n");
    }
    // ---- Transform::ApplyInverse(const Ray &, Float *tMax) and ApplyInverse(Point3f) for general affine transforms
    //      (util/transform.h:387-429, util/transform.cpp:263-303): renderFromMedium of a placed GridMedium / NanoVDBMedium
    //      (media.h:322, :354).  The reference's own m and mInv are emitted with every case. ----
    {
        printf("\"apply_inverse_xform\": [");
        bool first = true;
        for (int i = 0; i < 48; ++i) {
            Transform T;
            switch (i % 6) {
            case 0: T = Translate(Vector3f(4 * U() - 2, 4 * U() - 2, 4 * U() - 2)); break;
            case 1: T = Scale(0.25f + 3 * U(), 0.25f + 3 * U(), 0.25f + 3 * U()); break;
            case 2: T = Rotate(360 * U(), Normalize(Vector3f(2 * U() - 1, 2 * U() - 1, 2 * U() - 1))); break;
            case 3: T = Translate(Vector3f(2 * U() - 1, 2 * U() - 1, 2 * U() - 1)) * RotateY(360 * U()) * Scale(0.5f + U(), 0.5f + U(), 0.5f + U()); break;
            case 4: T = RotateX(90) * Translate(Vector3f(0.5f, -0.25f, 3 * U())); break;
            default: T = Translate(Vector3f(10 * U(), -7 * U(), 5 * U())) * Rotate(360 * U(), Normalize(Vector3f(U(), U(), U() + 0.1f))) * Scale(2 * U() + 0.1f, 2 * U() + 0.1f, 2 * U() + 0.1f); break;
            }
            Point3f o(4 * U() - 2, 4 * U() - 2, 4 * U() - 2);
            Vector3f d(2 * U() - 1, 2 * U() - 1, 2 * U() - 1);
            if (i % 8 == 7) d = Vector3f(0, 0, 0);
            Float tMax = (i % 5 == 0) ? Infinity : 4 * U();
            Float t = tMax;
            Ray q = T.ApplyInverse(Ray(o, d), &t);
            // (ApplyInverse(Point3f) itself CHECKs through util/print.cpp, which needs the absent double-conversion library:
            //  its arithmetic is the xp/yp/zp expression of the Point3fi form pinned here through the ray's origin)
            Vector3f vq = T.ApplyInverse(d);
            const SquareMatrix<4> &M = T.GetMatrix(), &MI = T.GetInverseMatrix();
            sep(first); printf("[");
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { pf(M[r][c]); printf(","); }
            for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) { pf(MI[r][c]); printf(","); }
            pf(o.x); printf(","); pf(o.y); printf(","); pf(o.z); printf(",");
            pf(d.x); printf(","); pf(d.y); printf(","); pf(d.z); printf(","); pf(tMax); printf(",");
            pf(q.o.x); printf(","); pf(q.o.y); printf(","); pf(q.o.z); printf(","); pf(q.d.x); printf(","); pf(q.d.y); printf(","); pf(q.d.z); printf(","); pf(t); printf(",");
            pf(vq.x); printf(","); pf(vq.y); printf(","); pf(vq.z); printf("]");
        }
        printf("],\n");
    }
    // ---- SpawnRayTo(Point3fi pFrom, Normal3f nFrom, time, Point3fi pTo, Normal3f nTo) (ray.h:103-108): the NEE shadow
    //      ray, whose origin and direction seed the shadow ray's RNG (guidedvolpathvspgintegrator.cpp:1193) ----
    {
        printf("\"spawn_ray_to\": [");
        bool first = true;
        for (int i = 0; i < 48; ++i) {
            Point3f a(2 * U() - 1, 2 * U() - 1, 2 * U() - 1), b(2 * U() - 1, 2 * U() - 1, 2 * U() - 1);
            Vector3f ea(1e-6f * U(), 1e-6f * U(), 1e-6f * U()), eb(1e-6f * U(), 1e-6f * U(), 1e-6f * U());
            Normal3f na = Normal3f(Normalize(Vector3f(2 * U() - 1, 2 * U() - 1, 2 * U() - 1)));
            Normal3f nb = Normal3f(Normalize(Vector3f(2 * U() - 1, 2 * U() - 1, 2 * U() - 1)));
            if (i % 4 == 0) { na = Normal3f(0, 0, 0); ea = Vector3f(0, 0, 0); }  // medium vertex: exact point, no normal
            if (i % 4 == 1) nb = Normal3f(0, -1, 0);
            Ray r = SpawnRayTo(Point3fi(a, ea), na, 0.f, Point3fi(b, eb), nb);
            sep(first); printf("[");
            pf(a.x); printf(","); pf(a.y); printf(","); pf(a.z); printf(","); pf(ea.x); printf(","); pf(ea.y); printf(","); pf(ea.z); printf(",");
            pf(na.x); printf(","); pf(na.y); printf(","); pf(na.z); printf(",");
            pf(b.x); printf(","); pf(b.y); printf(","); pf(b.z); printf(","); pf(eb.x); printf(","); pf(eb.y); printf(","); pf(eb.z); printf(",");
            pf(nb.x); printf(","); pf(nb.y); printf(","); pf(nb.z); printf(",");
            pf(r.o.x); printf(","); pf(r.o.y); printf(","); pf(r.o.z); printf(","); pf(r.d.x); printf(","); pf(r.d.y); printf(","); pf(r.d.z); printf("]");
        }
        printf("],\n");
    }
    // ---- Frame::FromXZ / ToLocal / FromLocal (vecmath.h:1850-1920): the shading frame of BSDF (bsdf.h:20-88) ----
    {
        printf("\"frame_xz\": [");
        bool first = true;
        for (int i = 0; i < 32; ++i) {
            Vector3f z = Normalize(Vector3f(2 * U() - 1, 2 * U() - 1, 2 * U() - 1));
            Vector3f t(2 * U() - 1, 2 * U() - 1, 2 * U() - 1);
            Vector3f x = Normalize(t - Dot(t, z) * z);  // some unit vector orthogonal to z (up to rounding)
            if (i == 0) { x = Vector3f(1, 0, 0); z = Vector3f(0, 0, 1); }
            if (i == 1) { x = Vector3f(0, 0, -1); z = Vector3f(0, 1, 0); }
            Frame f = Frame::FromXZ(x, z);
            Vector3f v(2 * U() - 1, 2 * U() - 1, 2 * U() - 1);
            Vector3f l = f.ToLocal(v), w = f.FromLocal(v);
            sep(first); printf("[");
            pf(x.x); printf(","); pf(x.y); printf(","); pf(x.z); printf(","); pf(z.x); printf(","); pf(z.y); printf(","); pf(z.z); printf(",");
            pf(v.x); printf(","); pf(v.y); printf(","); pf(v.z); printf(",");
            pf(f.y.x); printf(","); pf(f.y.y); printf(","); pf(f.y.z); printf(",");
            pf(l.x); printf(","); pf(l.y); printf(","); pf(l.z); printf(","); pf(w.x); printf(","); pf(w.y); printf(","); pf(w.z); printf("]");
        }
        printf("],\n");
    }
    // ---- round 4: Interval arithmetic (util/math.h:818-1010 over util/float.h:199-297), the arithmetic under Sphere::BasicIntersect ----
    {
        printf("\"interval_ops\": [");
        bool first = true;
        for (int i = 0; i < 96; ++i) {
            auto iv = [&]() {
                Float c = (i % 3 == 0 ? 4.f : 1.f) * (2 * U() - 1), w = (i % 5 == 0) ? 0.f : (i % 2 ? 1e-6f : 0.3f) * U();
                return Interval(c - w, c + w);
            };
            Interval a = iv(), b = iv();
            Float f = (i % 4 == 0) ? -.5f : ((i % 4 == 1) ? 2.f : 3 * (2 * U() - 1));
            Interval r[7] = {a + b, a - b, a * b, a / b, Sqr(a), Sqrt(Abs(a)), f * a};
            sep(first); printf("[");
            pf(a.LowerBound()); printf(","); pf(a.UpperBound()); printf(","); pf(b.LowerBound()); printf(","); pf(b.UpperBound()); printf(","); pf(f);
            for (int k = 0; k < 7; ++k) { printf(","); pf(r[k].LowerBound()); printf(","); pf(r[k].UpperBound()); }
            printf("]");
        }
        printf("],\n");
    }
    // ---- Shape "sphere".  shapes.h itself does not link here (util/mesh.h's STAT counters -> util/stats.cpp -> print.cpp ->
    //      the absent double-conversion), so the two member functions' statement sequences (Sphere::BasicIntersect shapes.h:147-229,
    //      InteractionFromIntersection :237-284; full sphere) are evaluated HERE on the reference's own types -- Interval, Point3fi,
    //      Vector3fi, Transform, SurfaceInteraction and Transform::operator()(SurfaceInteraction) as compiled from
    //      util/transform.cpp: every arithmetic operation is the reference's; what is pinned is the oracle's C restatement of it. ----
    {
        printf("\"sphere\": [");
        bool first = true;
        for (int i = 0; i < 160; ++i) {
            Transform T;
            switch (i % 5) {
            case 0: T = Transform(); break;
            case 1: T = Translate(Vector3f(2 * U() - 1, 2 * U() - 1, 2 * U() - 1)); break;
            case 2: T = Translate(Vector3f(U(), U(), U())) * Scale(0.5f + U(), 0.5f + U(), 0.5f + U()); break;
            case 3: T = Translate(Vector3f(U(), -U(), U())) * Rotate(360 * U(), Normalize(Vector3f(U(), U(), U() + 0.1f))) * Scale(1 + U(), 1, 0.5f + U()); break;
            default: T = Scale(-1, 1, 1) * Translate(Vector3f(0.2f, 0.1f, -0.3f)); break;   // swaps handedness
            }
            const Transform Ti = Inverse(T);
            const Transform *renderFromObject = &T, *objectFromRender = &Ti;
            const Float radius = 0.5f + 1.5f * U();
            const bool reverseOrientation = (i % 7) == 3;
            const bool transformSwapsHandedness = T.SwapsHandedness();
            const Float zMin = -radius, zMax = radius, thetaZMin = std::acos(Clamp(std::min(zMin, zMax) / radius, -1, 1)),
                        thetaZMax = std::acos(Clamp(std::max(zMin, zMax) / radius, -1, 1)), phiMax = Radians(Clamp(360.f, 0, 360));
            Point3f c = T(Point3f(0, 0, 0));
            Point3f o = (i % 4 == 0) ? c + Vector3f(0.3f * (2 * U() - 1), 0.3f * (2 * U() - 1), 0.3f * (2 * U() - 1))   // inside
                                     : c + 4.f * Normalize(Vector3f(2 * U() - 1, 2 * U() - 1, 2 * U() - 1));
            Vector3f d = (i % 4 == 0 || i % 3) ? Normalize(c + Vector3f(radius * (2 * U() - 1), radius * (2 * U() - 1), radius * (2 * U() - 1)) - o)
                                               : Normalize(Vector3f(2 * U() - 1, 2 * U() - 1, 2 * U() - 1));
            if (i % 11 == 5) d = 3.f * d;   // an unnormalised direction (shadow rays)
            const Float tMax = (i % 9 == 8) ? 3.5f : Infinity;
            Ray r(o, d);
            // -- BasicIntersect
            bool hit = false;
            Float tHit = 0, phi = 0;
            Point3f pHit;
            do {
                Point3fi oi = (*objectFromRender)(Point3fi(r.o));
                Vector3fi di = (*objectFromRender)(Vector3fi(r.d));
                Interval t0, t1;
                Interval a = Sqr(di.x) + Sqr(di.y) + Sqr(di.z);
                Interval b = 2 * (di.x * oi.x + di.y * oi.y + di.z * oi.z);
                Interval cc = Sqr(oi.x) + Sqr(oi.y) + Sqr(oi.z) - Sqr(Interval(radius));
                Vector3fi v(oi - b / (2 * a) * di);
                Interval length = Length(v);
                Interval discrim = 4 * a * (Interval(radius) + length) * (Interval(radius) - length);
                if (discrim.LowerBound() < 0) break;
                Interval rootDiscrim = Sqrt(discrim);
                Interval q;
                if ((Float)b < 0) q = -.5f * (b - rootDiscrim);
                else q = -.5f * (b + rootDiscrim);
                t0 = q / a;
                t1 = cc / q;
                if (t0.LowerBound() > t1.LowerBound()) pstd::swap(t0, t1);
                if (t0.UpperBound() > tMax || t1.LowerBound() <= 0) break;
                Interval tShapeHit = t0;
                if (tShapeHit.LowerBound() <= 0) {
                    tShapeHit = t1;
                    if (tShapeHit.UpperBound() > tMax) break;
                }
                pHit = Point3f(oi) + (Float)tShapeHit * Vector3f(di);
                pHit *= radius / Distance(pHit, Point3f(0, 0, 0));
                if (pHit.x == 0 && pHit.y == 0) pHit.x = 1e-5f * radius;
                phi = std::atan2(pHit.y, pHit.x);
                if (phi < 0) phi += 2 * Pi;
                if ((zMin > -radius && pHit.z < zMin) || (zMax < radius && pHit.z > zMax) || phi > phiMax) break;  // (never: full sphere)
                tHit = Float(tShapeHit);
                hit = true;
            } while (false);
            const SquareMatrix<4> &M = T.GetMatrix(), &MI = T.GetInverseMatrix();
            sep(first); printf("[");
            for (int rr = 0; rr < 4; ++rr) for (int c2 = 0; c2 < 4; ++c2) { pf(M[rr][c2]); printf(","); }
            for (int rr = 0; rr < 4; ++rr) for (int c2 = 0; c2 < 4; ++c2) { pf(MI[rr][c2]); printf(","); }
            pf(radius); printf(",%d,", reverseOrientation ? 1 : 0);
            pf(o.x); printf(","); pf(o.y); printf(","); pf(o.z); printf(","); pf(d.x); printf(","); pf(d.y); printf(","); pf(d.z); printf(","); pf(tMax);
            printf(",%d", hit ? 1 : 0);
            if (hit) {
                // -- InteractionFromIntersection
                Float u = phi / phiMax;
                Float cosTheta = pHit.z / radius;
                Float theta = SafeACos(cosTheta);
                Float vv = (theta - thetaZMin) / (thetaZMax - thetaZMin);
                Float zRadius = std::sqrt(Sqr(pHit.x) + Sqr(pHit.y));
                Float cosPhi = pHit.x / zRadius, sinPhi = pHit.y / zRadius;
                Vector3f dpdu(-phiMax * pHit.y, phiMax * pHit.x, 0);
                Float sinTheta = SafeSqrt(1 - Sqr(cosTheta));
                Vector3f dpdv = (thetaZMax - thetaZMin) * Vector3f(pHit.z * cosPhi, pHit.z * sinPhi, -radius * sinTheta);
                Vector3f pError = gamma(5) * Abs((Vector3f)pHit);
                bool flipNormal = reverseOrientation ^ transformSwapsHandedness;
                Vector3f woObject = (*objectFromRender)(-r.d);
                SurfaceInteraction si = (*renderFromObject)(SurfaceInteraction(Point3fi(pHit, pError), Point2f(u, vv), woObject, dpdu, dpdv,
                                                                               Normal3f(0, 0, 0), Normal3f(0, 0, 0), 0.f, flipNormal));
                Vector3f du = Normalize(si.shading.dpdu);
                printf(","); pf(tHit);
                printf(","); pf(pHit.x); printf(","); pf(pHit.y); printf(","); pf(pHit.z);
                printf(","); pf(si.pi.x.LowerBound()); printf(","); pf(si.pi.y.LowerBound()); printf(","); pf(si.pi.z.LowerBound());
                printf(","); pf(si.pi.x.UpperBound()); printf(","); pf(si.pi.y.UpperBound()); printf(","); pf(si.pi.z.UpperBound());
                printf(","); pf(si.n.x); printf(","); pf(si.n.y); printf(","); pf(si.n.z);
                printf(","); pf(du.x); printf(","); pf(du.y); printf(","); pf(du.z);
                // SpawnRay(ray.d) from the hit (SkipIntersection, interaction.cpp:91-97) and GetMedium's test Dot(w, n) > 0 (interaction.h:117-121)
                Point3f so = OffsetRayOrigin(si.pi, si.n, r.d);
                printf(","); pf(so.x); printf(","); pf(so.y); printf(","); pf(so.z);
                printf(",%d", Dot(r.d, si.n) > 0 ? 1 : 0);
            }
            printf("]");
        }
        printf("],\n");
    }
    // ---- volume emission of a temperature grid in the RGB build.  The wavelengths are the reference's own
    //      SampledWavelengths::SampleVisible (spectrum.h:369-386 over SampleVisibleWavelengths, util/sampling.h:169-171, i.e. the host's
    //      atanhf).  Blackbody() itself does not link here (its CHECK(!IsNaN(Le)) pulls LogFatal -> util/log.cpp -> print.cpp ->
    //      double-conversion, absent): as for the sphere, the generator evaluates the STATEMENT SEQUENCE of Blackbody() (spectrum.h:83-94)
    //      and of BlackbodySpectrum's constructor and Sample() (:568-588) with the reference's own FastExp and Pow<5> (util/math.h), as
    //      NanoVDBMedium::Le reaches them (media.h:724-735) ----
    {
        auto blackbody = [](Float lambda, Float T) -> Float {
            if (T <= 0)
                return 0;
            const Float c = 299792458.f;
            const Float h = 6.62606957e-34f;
            const Float kb = 1.3806488e-23f;
            Float l = lambda * 1e-9f;
            Float Le = (2 * h * c * c) / (Pow<5>(l) * (FastExp((h * c) / (l * kb * T)) - 1));
            return Le;
        };
        printf("\"blackbody\": [");
        bool first = true;
        for (int i = 0; i < 192; ++i) {
            Float u = U();
            if (i == 0) u = 0.f;
            if (i == 1) u = 0x1.fffffep-1f;
            if (i == 2) u = 1.f / 3.f;
            if (i == 3) u = 2.f / 3.f;
            Float T = 100.f + U() * U() * 11900.f;  // the Le() threshold is 100 K; explosion-like grids reach a few thousand
            if (i % 16 == 5) T = 100.0001f + U();
            if (i % 16 == 9) T = 20000.f + 60000.f * U();
            SampledWavelengths swl = SampledWavelengths::SampleVisible(u);
            Float lambdaMax = 2.8977721e-3f / T;
            Float normalizationFactor = 1 / blackbody(lambdaMax * 1e9f, T);
            SampledSpectrum Le;
            for (int c = 0; c < NSpectrumSamples; ++c)
                Le[c] = blackbody(swl[c], T) * normalizationFactor;
            sep(first);
            printf("[");
            pf(u); printf(","); pf(T);
            for (int c = 0; c < 3; ++c) { printf(","); pf(swl[c]); }
            for (int c = 0; c < 3; ++c) { printf(","); pf(Le[c]); }
            printf("]");
        }
        printf("]\n");
    }
    printf("}\n");
    return 0;
}

// host_selftest.cpp -- CPU-only checks of the host adapter's plugin-surface behaviour
// (parameter names / defaults / unused-parameter errors / registry names).  No device calls.
#include <cstdio>
#include <cstring>

#include "vspg_host.h"

static int fails = 0;
#define CHECK(c) do { if (!(c)) { std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #c); ++fails; } } while (0)
template <class F> static bool throws(F f) { try { f(); } catch (const vspg::Error &) { return true; } return false; }

int main() {
    using namespace vspg;
    // defaults == GuidedVolPathVSPGIntegrator::Create defaults
    VspgIntegratorParams d = ParseIntegratorParams(ParameterDictionary());
    VspgIntegratorParams ref;
    vspg_integrator_params_default(&ref);
    CHECK(std::memcmp(&d, &ref, sizeof d) == 0);
    // explicit values
    ParameterDictionary p;
    p.Int("maxdepth", 9).Int("minrrdepth", 3).Bool("usenee", false).String("vspcriterion", "Contribution")
        .String("vspsamplingmethod", "NDS").Float("vspmisratio", 0.25f).String("lightsampler", "uniform")
        .String("surfaceguidingtype", "mis").String("volumeguidingtype", "ris");
    VspgIntegratorParams q = ParseIntegratorParams(p);
    CHECK(q.maxdepth == 9 && q.minrrdepth == 3 && q.usenee == 0);
    CHECK(q.vspcriterion == VSPG_VSP_CONTRIBUTION && q.vspsamplingmethod == VSPG_VSP_NDS);
    CHECK(q.vspmisratio == 0.25f && q.lightsampler == VSPG_LIGHTSAMPLER_UNIFORM);
    CHECK(q.surfaceguidingtype == VSPG_GUIDE_MIS && q.volumeguidingtype == VSPG_GUIDE_RIS);
    p.ReportUnused();  // everything was looked up
    // unused parameter is fatal (paramdict.cpp:642-664)
    ParameterDictionary bad;
    bad.Int("maxdepth", 5).Bool("nosuchoption", true);
    (void)ParseIntegratorParams(bad);
    CHECK(throws([&] { bad.ReportUnused(); }));
    CHECK(throws([&] { ParameterDictionary t; t.Float("maxdepth", 1.f); (void)ParseIntegratorParams(t); }));  // wrong type
    CHECK(throws([&] { ParameterDictionary t; t.String("lightsampler", "nope"); (void)ParseIntegratorParams(t); }));
    // media
    VspgMedium m = CreateMedium("homogeneous", ParameterDictionary().RGB("sigma_a", .1f, .2f, .3f).RGB("sigma_s", 1, 2, 3)
                                                   .Float("scale", 2.f).Float("g", .5f));
    CHECK(m.type == VSPG_MEDIUM_HOMOGENEOUS && m.sigma_a[1] == .4f && m.sigma_s[2] == 6.f && m.g == .5f && m.Le[0] == 0);
    VspgMedium md = CreateMedium("homogeneous", ParameterDictionary());
    CHECK(md.sigma_a[0] == 1.f && md.sigma_s[0] == 1.f && md.g == 0.f);  // ConstantSpectrum(1) defaults
    CHECK(throws([] { CreateMedium("nanovdb", ParameterDictionary()); }));
    CHECK(throws([] { CreateMedium("homogeneous", ParameterDictionary().Float("bogus", 1)); }));
    // registry names (integrators.cpp:3739-3764)
    VspgScene scene;
    vspg_scene_fog_box(&scene, 16, 16);
    CHECK(throws([&] { Integrator::Create("path", ParameterDictionary(), scene, 16, 16, 1); }));
    CHECK(throws([&] { Integrator::Create("guidedvolpath", ParameterDictionary(), scene, 16, 16, 1); }));
    // out-of-scope options are refused loudly, before any device work
    CHECK(throws([&] { Integrator::Create("guidedvolpathvspg", ParameterDictionary().Bool("rrguiding", true), scene, 16, 16, 1); }));
    CHECK(throws([&] { Integrator::Create("guidedvolpathvspg", ParameterDictionary().Bool("collisionProbabilityBias", true), scene, 16, 16, 1); }));
    std::printf(fails ? "host_selftest: %d FAILED\n" : "host_selftest: ok\n", fails);
    return fails ? 1 : 0;
}

// host_selftest.cpp -- CPU-only checks of the host adapter's plugin-surface behaviour
// (parameter names / defaults / unused-parameter errors / registry names).  No device calls.
#include <cstdio>
#include <cstring>

#include "vspg_host.h"
#include "vspg_scenefile.h"

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
    CHECK(throws([] { CreateMedium("nanovdb", ParameterDictionary()); }));  // Must supply "filename" (media.cpp:685)
    CHECK(throws([] { std::vector<float> d, t; CreateMedium("nanovdb", ParameterDictionary().String("filename", "/nonexistent/x.nvdb"), &d, &t); }));
    CHECK(throws([] { CreateMedium("homogeneous", ParameterDictionary().Float("bogus", 1)); }));
    // registry names (integrators.cpp:3739-3764)
    VspgScene scene;
    vspg_scene_fog_box(&scene, 16, 16);
    CHECK(throws([&] { Integrator::Create("path", ParameterDictionary(), scene, 16, 16, 1); }));
    CHECK(throws([&] { Integrator::Create("guidedvolpath", ParameterDictionary(), scene, 16, 16, 1); }));
    // guided Russian roulette flags (guidedvolpathvspgintegrator.cpp:1314-1316)
    {
        VspgIntegratorParams rp = ParseIntegratorParams(ParameterDictionary().Bool("rrguiding", true).Bool("volumerrguiding", false));
        CHECK(rp.rrguiding == 1 && rp.surfacerrguiding == 1 && rp.volumerrguiding == 0);
    }
    // the reference's parameter-list text (what cmd/nanovdb2pbrt prints for a grid) -> GridMedium::Create
    {
        const char *txt = "  \"integer nx\" 2 \"integer ny\" [ 3 ] \"integer nz\" 1   # comment\n"
                          "\"point3 p0\" [ -1 -0.5 0 ] \"point3 p1\" [ 1 0.5 2 ]\n"
                          "\"rgb sigma_a\" [ .1 .2 .3 ] \"rgb sigma_s\" [ 1 2 3 ] \"float scale\" 2 \"float g\" [ 0.25 ]\n"
                          "\"float density\" [ 0 0.5 1\n 1.5 2 2.5 ]";
        std::vector<float> storage;
        VspgMedium gm = CreateMedium("uniformgrid", ParameterDictionary::Parse(txt), &storage);
        CHECK(gm.type == VSPG_MEDIUM_GRID && gm.nx == 2 && gm.ny == 3 && gm.nz == 1);
        CHECK(gm.bounds_min[1] == -0.5f && gm.bounds_max[2] == 2.f && gm.g == 0.25f);
        CHECK(gm.sigma_a[2] == 0.6f && gm.sigma_s[0] == 2.f);
        CHECK(storage.size() == 6 && gm.density == storage.data() && storage[5] == 2.5f);
        CHECK(throws([&] { std::vector<float> st; CreateMedium("uniformgrid", ParameterDictionary::Parse("\"integer nx\" 2 \"float density\" [1 2 3]"), &st); }));  // count mismatch
        CHECK(throws([&] { std::vector<float> st; CreateMedium("uniformgrid", ParameterDictionary::Parse("\"integer nx\" 1"), &st); }));  // no density
        {   // "temperature" (media.cpp:276-302, :345-347): as many samples as "density", offset / cutoff / scale, never together with "Le"
            std::vector<float> st, ls, ts;
            VspgMedium tm = CreateMedium("uniformgrid", ParameterDictionary::Parse("\"integer nx\" 2 \"float density\" [1 2] \"float temperature\" [300 1500] "
                                                                                   "\"float temperaturecutoff\" 50 \"float temperaturescale\" 2"), &st, &ls, &ts);
            CHECK(tm.temperature == ts.data() && ts.size() == 2 && ts[1] == 1500.f && tm.temperature_offset == 50.f && tm.temperature_scale == 2.f);
            CHECK(tm.Le[0] == 0 && tm.le_scale == nullptr);
            tm = CreateMedium("uniformgrid", ParameterDictionary::Parse("\"integer nx\" 2 \"float density\" [1 2] \"float temperature\" [300 1500] "
                                                                        "\"float temperatureoffset\" 7 \"float temperaturecutoff\" 50 \"float Lescale\" [0.5 2]"), &st, &ls, &ts);
            CHECK(tm.temperature_offset == 7.f && tm.temperature_scale == 1.f && tm.le_scale == ls.data() && tm.le_nx == 2 && ls[1] == 2.f);
            CHECK(throws([&] { CreateMedium("uniformgrid", ParameterDictionary::Parse("\"float density\" 1 \"float temperature\" [300 400]"), &st, &ls, &ts); }));
            CHECK(throws([&] { CreateMedium("uniformgrid", ParameterDictionary::Parse("\"float density\" 1 \"float temperature\" 300 \"rgb Le\" [1 1 1]"), &st, &ls, &ts); }));
            CHECK(throws([&] { CreateMedium("uniformgrid", ParameterDictionary::Parse("\"float density\" 1 \"float temperature\" 300"), &st, &ls); }));  // nowhere to keep the grid
        }
        CHECK(throws([] { ParameterDictionary::Parse("\"spectrum sigma_a\" \"metal-Au-eta\""); }));   // type outside scope
        CHECK(throws([] { ParameterDictionary::Parse("\"float g\" [ 0.5"); }));                        // missing ]
        ParameterDictionary pd = ParameterDictionary::Parse("\"bool usenee\" false \"string lightsampler\" \"uniform\" \"integer maxdepth\" 7");
        VspgIntegratorParams ip = ParseIntegratorParams(pd);
        CHECK(ip.usenee == 0 && ip.lightsampler == VSPG_LIGHTSAMPLER_UNIFORM && ip.maxdepth == 7);
    }
    // guiding-cache file round trip (Field::Store / Field(file)) and its parameters
    {
        GuidingCache gc;
        for (int f = 0; f < 2; ++f) {
            gc.nodes[f].resize(3);
            gc.nodes[f][0] = VspgKdNode{0.25f, (1u << 2) | (uint32_t)f};
            gc.nodes[f][1] = VspgKdNode{0.f, (0u << 2) | 3u};
            gc.nodes[f][2] = VspgKdNode{0.f, (1u << 2) | 3u};
            gc.regions[f].resize(2);
            std::memset(gc.regions[f].data(), 0, 2 * sizeof(VspgFieldRegion));
            gc.regions[f][1].n_lobes = 2;
            gc.regions[f][1].weight[0] = 0.75f; gc.regions[f][1].weight[1] = 0.25f;
            gc.regions[f][1].kappa[0] = 12.f + f;
        }
        const char *fn = "/tmp/vspg_host_selftest.vspgfld";
        gc.Write(fn);
        GuidingCache rd = GuidingCache::Read(fn);
        for (int f = 0; f < 2; ++f) {
            CHECK(rd.nodes[f].size() == 3 && rd.regions[f].size() == 2);
            CHECK(std::memcmp(rd.nodes[f].data(), gc.nodes[f].data(), 3 * sizeof(VspgKdNode)) == 0);
            CHECK(std::memcmp(rd.regions[f].data(), gc.regions[f].data(), 2 * sizeof(VspgFieldRegion)) == 0);
        }
        std::remove(fn);
        CHECK(throws([&] { GuidingCache::Read("/tmp/vspg_no_such_cache"); }));
        GuidingCacheSettings cs;
        (void)ParseIntegratorParams(ParameterDictionary().Bool("storeGuidingCache", true).String("guidingCacheFileName", "x.fld"), &cs);
        CHECK(cs.store && !cs.load && cs.fileName == "x.fld");
        CHECK(throws([] { (void)ParseIntegratorParams(ParameterDictionary().Bool("loadGuidingCache", true)); }));  // no file name
        CHECK(throws([] { (void)ParseIntegratorParams(ParameterDictionary().Bool("storeISGBuffer", true)); }));    // no file name
    }
    // emissive "uniformgrid" (media.cpp:306-328): "Le" + "Lescale" (LeNorm = 1 in RGB rendering mode)
    {
        std::vector<float> dens(8, 0.5f), lesc = {0, 1, 2, 3, 4, 5, 6, 7}, ds, ls;
        VspgMedium m = CreateMedium("uniformgrid", ParameterDictionary().Int("nx", 2).Int("ny", 2).Int("nz", 2).FloatArray("density", dens)
                                                       .RGB("Le", 1.f, 2.f, 3.f).FloatArray("Lescale", lesc), &ds, &ls);
        CHECK(m.type == VSPG_MEDIUM_GRID && m.Le[1] == 2.f && m.le_scale == ls.data() && ls == lesc && m.le_nx == 2 && m.le_nz == 2);
        m = CreateMedium("uniformgrid", ParameterDictionary().Int("nx", 2).Int("ny", 2).Int("nz", 2).FloatArray("density", dens).RGB("Le", 1.f, 2.f, 3.f), &ds);
        CHECK(m.Le[2] == 3.f && m.le_scale == nullptr);  // the library builds the 1x1x1 default grid
        m = CreateMedium("uniformgrid", ParameterDictionary().Int("nx", 2).Int("ny", 2).Int("nz", 2).FloatArray("density", dens).RGB("Le", 0.f, 0.f, 0.f), &ds);
        CHECK(m.Le[0] == 0.f && m.le_scale == nullptr);  // Le.MaxValue() == 0: not emissive
        CHECK(throws([&] { CreateMedium("uniformgrid", ParameterDictionary().Int("nx", 2).Int("ny", 2).Int("nz", 2).FloatArray("density", dens)
                                                           .RGB("Le", 1.f, 1.f, 1.f).FloatArray("Lescale", {1.f, 2.f}), &ds, &ls); }));
        CHECK(throws([&] { CreateMedium("uniformgrid", ParameterDictionary().Int("nx", 2).Int("ny", 2).Int("nz", 2).FloatArray("density", dens)
                                                           .FloatArray("temperature", dens), &ds); }));
    }
    // TrBuffer persistence (cpu/trbuffer.h Store / Load): PFM raster, bottom scanline first
    {
        TrBuffer tb;
        tb.xres = 5; tb.yres = 3;
        tb.rgb.resize(5 * 3 * 3);
        for (size_t i = 0; i < tb.rgb.size(); ++i) tb.rgb[i] = 0.01f * (float)i;
        const char *fn = "/tmp/vspg_host_selftest_tr.pfm";
        tb.Store(fn);
        TrBuffer rd = TrBuffer::Load(fn);
        CHECK(rd.xres == 5 && rd.yres == 3 && rd.rgb == tb.rgb);
        {  // the file's first raster row is the image's BOTTOM row
            std::FILE *f = std::fopen(fn, "rb");
            char hdr[32];
            CHECK(f && std::fgets(hdr, sizeof hdr, f) && std::string(hdr) == "PF\n");
            CHECK(std::fgets(hdr, sizeof hdr, f) && std::string(hdr) == "5 3\n");
            CHECK(std::fgets(hdr, sizeof hdr, f) && std::string(hdr) == "-1.000000\n");
            float first = -1;
            CHECK(std::fread(&first, 4, 1, f) == 1 && first == tb.rgb[2 * 5 * 3]);
            std::fclose(f);
        }
        std::remove(fn);
        CHECK(throws([&] { tb.Store("/tmp/vspg_host_selftest_tr.exr"); }));
        CHECK(throws([] { TrBuffer::Load("/tmp/vspg_no_such_tr.pfm"); }));
        TrBufferSettings ts;
        VspgIntegratorParams p = ParseIntegratorParams(
            ParameterDictionary().Bool("storeTrBuffer", true).String("trBufferFileName", "t.pfm").Bool("collisionProbabilityBias", true), nullptr, &ts);
        CHECK(ts.store && !ts.load && ts.fileName == "t.pfm" && p.storeTrBuffer == 1 && p.collisionProbabilityBias == 1);
        CHECK(throws([] { (void)ParseIntegratorParams(ParameterDictionary().Bool("loadTrBuffer", true)); }));  // no file name
    }
    // image-space VSP buffer persistence: single-channel PFM
    {
        VspBuffer vb;
        vb.xres = 4; vb.yres = 3; vb.ready = true;
        vb.vsp.resize(12);
        for (size_t i = 0; i < vb.vsp.size(); ++i) vb.vsp[i] = 0.05f * (float)i;
        const char *fn = "/tmp/vspg_host_selftest_isg.pfm";
        vb.Store(fn);
        VspBuffer rd = VspBuffer::Load(fn);
        CHECK(rd.xres == 4 && rd.yres == 3 && rd.ready && rd.vsp == vb.vsp);
        std::remove(fn);
        CHECK(throws([] { VspBuffer::Load("/tmp/vspg_no_such_isg.pfm"); }));
        IsgBufferSettings is;
        (void)ParseIntegratorParams(ParameterDictionary().Bool("loadISGBuffer", true).String("isgBufferFileName", "v.pfm"), nullptr, nullptr, &is);
        CHECK(is.load && !is.store && is.fileName == "v.pfm");
    }
    // scene-file reader: index / orientation / medium checks (round-2 review; no device calls: ParseSceneString only)
    {
        const std::string head = "Camera \"perspective\"\nSampler \"halton\" \"integer pixelsamples\" 2\nFilm \"rgb\" \"integer xresolution\" 8 \"integer yresolution\" 8\n"
                                 "Integrator \"guidedvolpathvspg\"\nWorldBegin\n";
        const std::string quadP = "\"point3 P\" [ 0 0 0  1 0 0  0 1 0  1 1 0 ]";
        CHECK(throws([&] { (void)ParseSceneString(head + "Shape \"bilinearmesh\" " + quadP + " \"integer indices\" [ 0 1 2 500000000 ]\n"); }));
        CHECK(throws([&] { (void)ParseSceneString(head + "Shape \"bilinearmesh\" " + quadP + " \"integer indices\" [ 0 1 2 -7 ]\n"); }));
        CHECK(throws([&] { (void)ParseSceneString(head + "Shape \"bilinearmesh\" \"point3 P\" [ 0 0 0  1 0 0  0 1 0  1 1 0  5 ]\n"); }));
        CHECK(throws([&] { (void)ParseSceneString(head + "Shape \"trianglemesh\" \"point3 P\" [ 0 0 0  1 0 0  0 1 0 ] \"integer indices\" [ 0 1 3 ]\n"); }));
        {  // a mirroring CTM turns a patch over: reverseOrientation ^ transformSwapsHandedness (shapes.h:1163-1164)
            auto a = ParseSceneString(head + "AreaLightSource \"diffuse\" \"rgb L\" [1 1 1]\nShape \"bilinearmesh\" " + quadP + "\n");
            auto b = ParseSceneString(head + "Scale -1 1 1\nAreaLightSource \"diffuse\" \"rgb L\" [1 1 1]\nShape \"bilinearmesh\" " + quadP + "\n");
            auto c2 = ParseSceneString(head + "Scale -1 1 1\nReverseOrientation\nAreaLightSource \"diffuse\" \"rgb L\" [1 1 1]\nShape \"bilinearmesh\" " + quadP + "\n");
            CHECK(a->scene.n_quads == 1 && a->scene.quads[0].reverse_orientation == 0);
            CHECK(b->scene.n_quads == 1 && b->scene.quads[0].reverse_orientation == 1);
            CHECK(c2->scene.n_quads == 1 && c2->scene.quads[0].reverse_orientation == 0);
            bool warned = false;
            for (const auto &w : a->warnings) warned = warned || w.find("halton") != std::string::npos;
            CHECK(warned);  // every sampler runs as "independent", and says so
        }
        {   // medium boundaries (round 4): shapes keep their MediumInterface, the camera its side; two media in one scene are refused by name
            const std::string mm = "MakeNamedMedium \"m\" \"string type\" \"homogeneous\"\n";
            auto a = ParseSceneString(head + mm + "MediumInterface \"m\" \"m\"\nShape \"bilinearmesh\" " + quadP + "\n");   // both sides in m: not a transition
            CHECK(a->scene.medium.type == VSPG_MEDIUM_HOMOGENEOUS && a->scene.camera_outside_medium == 1);
            CHECK(a->scene.quads[0].medium_interface == (VSPG_IFACE_INSIDE | VSPG_IFACE_OUTSIDE) && a->scene.quads[0].material == VSPG_MATERIAL_DIFFUSE);
            auto b = ParseSceneString(head + mm + "MediumInterface \"m\" \"\"\nMaterial \"interface\"\nTranslate 1 2 3\nScale 2 2 2\nShape \"sphere\" \"float radius\" 0.5\n"
                                             "Material \"diffuse\"\nMediumInterface \"\" \"\"\nShape \"trianglemesh\" \"point3 P\" [ 0 0 0  1 0 0  0 1 0 ]\n");
            CHECK(b->scene.n_spheres == 1 && b->scene.spheres[0].material == VSPG_MATERIAL_INTERFACE && b->scene.spheres[0].medium_interface == VSPG_IFACE_INSIDE);
            CHECK(b->scene.spheres[0].radius == 0.5f && b->scene.spheres[0].render_from_object[3] == 1.f && b->scene.spheres[0].render_from_object[0] == 2.f &&
                  b->scene.spheres[0].object_from_render[0] == 0.5f && b->scene.spheres[0].object_from_render[3] == -0.5f);
            CHECK(b->scene.n_triangles == 1 && b->scene.tri_flags && b->scene.tri_flags[0] == 0 && b->scene.camera_outside_medium == 1);
            auto c3 = ParseSceneString(head + mm + "Scale -1 1 1\nMediumInterface \"\" \"m\"\nMaterial \"none\"\nShape \"trianglemesh\" \"point3 P\" [ 0 0 0  1 0 0  0 1 0 ]\n");
            CHECK(c3->scene.tri_flags[0] == (VSPG_TRI_INTERFACE | VSPG_TRI_FLIP_NORMAL | (VSPG_IFACE_OUTSIDE << VSPG_TRI_IFACE_SHIFT)));
            CHECK(throws([&] { (void)ParseSceneString(head + mm + "MakeNamedMedium \"n\" \"string type\" \"homogeneous\"\nMediumInterface \"m\" \"n\"\nShape \"sphere\"\n"); }));
            CHECK(throws([&] { (void)ParseSceneString(head + "Shape \"sphere\" \"float phimax\" 180\n"); }));
        }
    }
    std::printf(fails ? "host_selftest: %d FAILED\n" : "host_selftest: ok\n", fails);
    return fails ? 1 : 0;
}

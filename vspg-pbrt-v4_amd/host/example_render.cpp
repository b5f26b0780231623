// example_render.cpp -- the scene-file lines of SURVEY.md App. F expressed through the host adapter:
//   Integrator "guidedvolpathvspg" "integer maxdepth" 5 "bool vspguiding" true "bool surfaceguiding" false ...
//   MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [.05 .05 .05] "rgb sigma_s" [.45 .45 .45] "float g" 0
// usage: example_render [xres yres spp out.pfm [train|load cachefile | trstore|trload tr.pfm]]
//   train: the reference's default guiding options (cfg 5: the field trains in-loop) + "bool storeGuidingCache" true
//   load:  the same options + "bool loadGuidingCache" true (no training, guidedvolpathvspgintegrator.cpp:117-122)
//   trstore / trload: the NDS+ workflow on a thin 12^3 "uniformgrid" medium -- pass 1 "vspsamplingmethod" resampling +
//          "bool storeTrBuffer" true, pass 2 "vspsamplingmethod" nds + "bool collisionProbabilityBias" true + "bool loadTrBuffer" true
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "vspg_host.h"

int main(int argc, char **argv) {
    int xres = argc > 1 ? std::atoi(argv[1]) : 256, yres = argc > 2 ? std::atoi(argv[2]) : 256;
    int spp = argc > 3 ? std::atoi(argv[3]) : 16;
    const char *out = argc > 4 ? argv[4] : "fogbox.pfm";
    try {
        VspgScene scene;
        if (vspg_scene_fog_box(&scene, xres, yres) != 0) throw vspg::Error(vspg_last_error());
        scene.medium = vspg::CreateMedium("homogeneous", vspg::ParameterDictionary()
                                                             .RGB("sigma_a", .05f, .05f, .05f)
                                                             .RGB("sigma_s", .45f, .45f, .45f)
                                                             .Float("g", 0.f));
        vspg::ParameterDictionary ip;
        const std::string mode = argc > 6 ? argv[5] : "";
        std::vector<float> densityStorage;
        if (mode == "trstore" || mode == "trload") {
            const int n = 12;
            std::vector<float> d((size_t)n * n * n);
            for (int k = 0; k < n; ++k)
                for (int j = 0; j < n; ++j)
                    for (int i = 0; i < n; ++i) d[((size_t)k * n + j) * n + i] = (float)((i * 7 + j * 13 + k * 29) % 17) / 16.f;
            scene.medium = vspg::CreateMedium("uniformgrid", vspg::ParameterDictionary()
                                                                 .Int("nx", n).Int("ny", n).Int("nz", n)
                                                                 .FloatArray("density", d)
                                                                 .Point3("p0", -0.8f, -0.8f, -0.5f).Point3("p1", 0.8f, 0.7f, 0.9f)
                                                                 .RGB("sigma_a", .02f, .03f, .04f).RGB("sigma_s", .5f, .45f, .4f)
                                                                 .Float("g", 0.3f),
                                             &densityStorage);
            ip.Int("maxdepth", 5).Bool("vspguiding", true).Bool("surfaceguiding", false).Bool("volumeguiding", false)
                .Bool("vspsecondaryguiding", false).String("trBufferFileName", argv[6]);
            if (mode == "trstore") ip.String("vspsamplingmethod", "resampling").Bool("storeTrBuffer", true);
            else ip.String("vspsamplingmethod", "nds").Bool("collisionProbabilityBias", true).Bool("loadTrBuffer", true);
        } else if (mode == "train" || mode == "load") {
            ip.Int("maxdepth", 5).Bool("vspguiding", true);  // surface / volume / secondary-VSP guiding default to true
            ip.Bool(mode == "train" ? "storeGuidingCache" : "loadGuidingCache", true).String("guidingCacheFileName", argv[6]);
        } else {
            ip.Int("maxdepth", 5).Bool("vspguiding", true).Bool("surfaceguiding", false).Bool("volumeguiding", false)
                .Bool("vspsecondaryguiding", false);
        }
        auto integrator = vspg::Integrator::Create("guidedvolpathvspg", ip, scene, xres, yres, spp);
        std::printf("%s\n", integrator->ToString().c_str());
        integrator->Render();
        auto *vi = static_cast<vspg::GuidedVolPathVSPGIntegrator *>(integrator.get());
        vspg::Film film = vi->GetFilm();
        film.WritePFM(out);
        VspgTrainStats ts = vi->TrainingStats();
        std::printf("guiding: training %d iterations %d regions %d/%d\n", ts.training, ts.iteration, ts.n_regions[0], ts.n_regions[1]);
        VspgCounters c = vi->Counters();
        float rgb[3];
        film.GetPixelRGB(xres / 2, yres / 2, rgb);
        std::printf("paths %llu segments %llu centre pixel %g %g %g -> %s\n", (unsigned long long)c.paths,
                    (unsigned long long)c.segments, rgb[0], rgb[1], rgb[2], out);
    } catch (const vspg::Error &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

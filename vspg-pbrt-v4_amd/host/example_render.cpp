// example_render.cpp -- the scene-file lines of SURVEY.md App. F expressed through the host adapter:
//   Integrator "guidedvolpathvspg" "integer maxdepth" 5 "bool vspguiding" true "bool surfaceguiding" false ...
//   MakeNamedMedium "fog" "string type" "homogeneous" "rgb sigma_a" [.05 .05 .05] "rgb sigma_s" [.45 .45 .45] "float g" 0
// usage: example_render [xres yres spp out.pfm [train|load cachefile]]
//   train: the reference's default guiding options (cfg 5: the field trains in-loop) + "bool storeGuidingCache" true
//   load:  the same options + "bool loadGuidingCache" true (no training, guidedvolpathvspgintegrator.cpp:117-122)
#include <cstdio>
#include <cstdlib>
#include <string>

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
        if (mode == "train" || mode == "load") {
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

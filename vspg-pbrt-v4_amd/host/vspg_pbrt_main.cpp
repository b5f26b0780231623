// vspg_pbrt -- render a pbrt-v4 scene file (the subset of vspg_scenefile.h) with the MI355X-native GuidedVolPathVSPG path.
//   vspg_pbrt scene.pbrt [--spp N] [--outfile image.pfm] [--seed S] [--device D] [--wave-log waves.jsonl] [--parse-only]
// The counterpart of `pbrt scene.pbrt` for this integrator (cmd/pbrt.cpp -> RenderCPU, cpu/render.cpp:56-57); the image is
// written as PFM (RGBFilm::WriteImage's EXR needs OpenEXR, an absent submodule).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "vspg_scenefile.h"

int main(int argc, char **argv) {
    std::string scene, out;
    int spp = -1, seed = -1, device = 0;
    std::string waveLogPath;
    bool parseOnly = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto next = [&]() -> const char * { if (i + 1 >= argc) { std::fprintf(stderr, "%s needs a value\n", a.c_str()); std::exit(2); } return argv[++i]; };
        if (a == "--spp") spp = std::atoi(next());
        else if (a == "--outfile") out = next();
        else if (a == "--seed") seed = std::atoi(next());
        else if (a == "--device") device = std::atoi(next());
        else if (a == "--wave-log") waveLogPath = next();
        else if (a == "--parse-only") parseOnly = true;
        else if (!a.empty() && a[0] == '-') { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
        else scene = a;
    }
    if (scene.empty()) { std::fprintf(stderr, "usage: vspg_pbrt scene.pbrt [--spp N] [--outfile image.pfm] [--seed S] [--device D] [--wave-log waves.jsonl] [--parse-only]\n"); return 2; }
    try {
        auto sd = vspg::ParseSceneFile(scene);
        if (spp > 0) sd->pixelSamples = spp;
        if (seed >= 0) sd->seed = seed;
        if (!out.empty()) sd->filmFilename = out;
        for (const auto &w : sd->warnings) std::fprintf(stderr, "Warning: %s\n", w.c_str());
        std::printf("scene: %d rectangles, %d triangles, %d infinite lights, medium type %d%s, film %dx%d @ %d spp, integrator \"%s\"",
                    sd->scene.n_quads, sd->scene.n_triangles, sd->scene.n_infinite_lights, sd->scene.medium.type,
                    sd->scene.medium.has_transform ? " (placed)" : "", sd->xres, sd->yres, sd->pixelSamples, sd->integratorName.c_str());
        {   // medium boundaries: spheres, interface-material surfaces, medium transitions, the camera's side
            int n_interface = 0, n_transition = 0;
            auto transition = [](int b) { b &= VSPG_IFACE_INSIDE | VSPG_IFACE_OUTSIDE; return b == VSPG_IFACE_INSIDE || b == VSPG_IFACE_OUTSIDE; };
            for (int i = 0; i < sd->scene.n_quads; ++i) { n_interface += sd->scene.quads[i].material == VSPG_MATERIAL_INTERFACE; n_transition += transition(sd->scene.quads[i].medium_interface); }
            for (int i = 0; i < sd->scene.n_spheres; ++i) { n_interface += sd->scene.spheres[i].material == VSPG_MATERIAL_INTERFACE; n_transition += transition(sd->scene.spheres[i].medium_interface); }
            for (size_t i = 0; i < sd->triFlags.size(); ++i) { n_interface += (sd->triFlags[i] & VSPG_TRI_INTERFACE) != 0; n_transition += transition(sd->triFlags[i] >> VSPG_TRI_IFACE_SHIFT); }
            if (sd->scene.n_spheres || n_interface || n_transition || sd->scene.camera_outside_medium)
                std::printf("; %d spheres, %d interface-material surfaces, %d medium transitions, camera %s the medium", sd->scene.n_spheres, n_interface, n_transition,
                            sd->scene.camera_outside_medium ? "outside" : "in");
        }
        if (sd->scene.medium.temperature) std::printf("; temperature grid (blackbody emission under \"vspsamplingmethod\" \"nds\")");
        std::printf("\n");
        if (parseOnly) return 0;
        auto integrator = vspg::CreateIntegrator(*sd, device);
        std::printf("%s\n", integrator->ToString().c_str());
        auto *vi = static_cast<vspg::GuidedVolPathVSPGIntegrator *>(integrator.get());
        std::FILE *wl = nullptr;
        if (!waveLogPath.empty()) {
            wl = std::fopen(waveLogPath.c_str(), "w");
            if (!wl) throw vspg::Error("cannot open " + waveLogPath);
            vi->SetWaveLog(wl);
        }
        integrator->Render();
        if (wl) { vi->SetWaveLog(nullptr); std::fclose(wl); }
        vspg::Film film = vi->GetFilm();
        film.WritePFM(sd->filmFilename);
        VspgCounters c = vi->Counters();
        std::printf("paths %llu segments %llu -> %s\n", (unsigned long long)c.paths, (unsigned long long)c.segments, sd->filmFilename.c_str());
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

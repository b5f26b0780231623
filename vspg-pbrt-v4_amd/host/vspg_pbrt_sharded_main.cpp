// vspg_pbrt_sharded -- one rank of a multi-GPU render of a pbrt scene file, all C++.
//   RANK=r WORLD_SIZE=N LOCAL_RANK=r [MASTER_PORT=p] vspg_pbrt_sharded scene.pbrt [--spp S] [--outfile image.pfm]
// Start N processes (one per GPU of the node) with the same arguments; rank 0 writes the image.  Every step covers N
// consecutive sample indices, rank r renders index step * N + r; the image-space VSP statistics are summed over the ranks
// where the buffer updates and the float film at frame end -- both through RCCL (include/vspg_rccl.h).  With N = 1 this is
// `vspg_pbrt` (same film, bit for bit).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vspg_rccl.h"
#include "vspg_scenefile.h"

int main(int argc, char **argv) {
    std::string scene, out;
    int spp = -1;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--spp" && i + 1 < argc) spp = std::atoi(argv[++i]);
        else if (a == "--outfile" && i + 1 < argc) out = argv[++i];
        else if (!a.empty() && a[0] == '-') { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
        else scene = a;
    }
    if (scene.empty()) { std::fprintf(stderr, "usage: [RANK= WORLD_SIZE= LOCAL_RANK=] vspg_pbrt_sharded scene.pbrt [--spp S] [--outfile image.pfm]\n"); return 2; }
    int rank = 0, world = 1, local = 0;
    void *comm = nullptr;
    VspgRenderer *r = nullptr;
    int code = 0;
    try {
        auto sd = vspg::ParseSceneFile(scene);
        if (spp > 0) sd->pixelSamples = spp;
        if (!out.empty()) sd->filmFilename = out;
        if (sd->integratorName != "guidedvolpathvspg" && sd->integratorName != "guidedvolpath") throw vspg::Error(sd->integratorName + ": integrator type unknown.");
        VspgIntegratorParams prm = vspg::ParseIntegratorParams(sd->integratorParams);
        sd->integratorParams.ReportUnused();
        if (vspg_rccl_init_from_env(nullptr, &rank, &world, &local, &comm) != 0) throw vspg::Error("RCCL communicator set-up failed");
        VspgRenderConfig cfg;
        std::memset(&cfg, 0, sizeof cfg);
        cfg.xres = sd->xres; cfg.yres = sd->yres; cfg.spp = sd->pixelSamples; cfg.seed = sd->seed;
        cfg.shard_index = rank; cfg.shard_count = world; cfg.device = local;
        if (vspg_renderer_create(&sd->scene, &prm, &cfg, &r) != 0) throw vspg::Error(vspg_last_error());
        int seen = 0;
        if (vspg_rccl_ranks_seen(comm, nullptr, &seen) != 0 || seen != world) throw vspg::Error("RCCL launch check: " + std::to_string(seen) + " of " + std::to_string(world) + " ranks answered");
        if (world > 1 && vspg_rccl_enable_training_exchange(r, comm) != 0) throw vspg::Error("training exchange set-up failed");
        const int steps = (sd->pixelSamples + world - 1) / world;
        for (int s = 0; s < steps; ++s) {
            const int w0 = s * world, w1 = (s + 1) * world < sd->pixelSamples ? (s + 1) * world : sd->pixelSamples;
            if (vspg_render_wave(r, w0, w1, nullptr) != 0) throw vspg::Error(vspg_last_error());   // this rank's index of the step
            // the last step of a frame whose sample count is not a multiple of the rank count covers w1 - w0 < world indices
            if (vspg_rccl_post_process_step_n(r, w1 - w0, world, comm, nullptr) != 0) throw vspg::Error(std::string("post-process step: ") + vspg_last_error());
        }
        if (vspg_rccl_allreduce_film(r, comm, nullptr) != 0) throw vspg::Error("film all-reduce failed");
        VspgCounters c;
        if (vspg_rccl_sum_counters(r, comm, nullptr, &c) != 0) throw vspg::Error("counter reduction failed");
        if (rank == 0) {
            vspg::Film film;
            film.xres = cfg.xres; film.yres = cfg.yres;
            film.rgbw.resize((size_t)cfg.xres * cfg.yres * 4);
            if (vspg_film_read(r, film.rgbw.data(), nullptr) != 0) throw vspg::Error(vspg_last_error());
            film.WritePFM(sd->filmFilename);
            std::printf("ranks %d: paths %llu segments %llu -> %s\n", world, (unsigned long long)c.paths, (unsigned long long)c.segments, sd->filmFilename.c_str());
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "rank %d error: %s\n", rank, e.what());
        code = 1;
    }
    if (r) { vspg_rccl_forget(r); vspg_renderer_destroy(r); }
    if (comm) vspg_rccl_destroy(comm);
    return code;
}

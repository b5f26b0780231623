// vspg_nvdb2grid.cpp -- prints what host/vspg_nanovdb.h reads from a .nvdb file and, with --dump, writes the dense float32
// values (x fastest) -- the same data the reference's converter writes as "uniformgrid" parameters (cmd/nanovdb2pbrt.cpp:97-126).
// The reader's layout is "as understood" (parity unpinned; see vspg_nanovdb.h): this tool is how to look at what it decoded.
//   vspg_nvdb2grid file.nvdb [--grid density] [--dump values.f32]
#include <cstdio>
#include <string>

#include "vspg_host.h"
#include "vspg_nanovdb.h"

int main(int argc, char **argv) {
    std::string file, grid = "density", dump;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--grid" && i + 1 < argc) grid = argv[++i];
        else if (a == "--dump" && i + 1 < argc) dump = argv[++i];
        else if (!a.empty() && a[0] == '-') { std::fprintf(stderr, "unknown option %s\n", a.c_str()); return 2; }
        else file = a;
    }
    if (file.empty()) { std::fprintf(stderr, "usage: vspg_nvdb2grid file.nvdb [--grid density] [--dump values.f32]\n"); return 2; }
    try {
        vspg::NanoVdbFloatGrid g;
        if (!vspg::ReadNanoVdbFloatGrid(file, grid, &g)) { std::fprintf(stderr, "error: %s: didn't find \"%s\" grid.\n", file.c_str(), grid.c_str()); return 1; }
        double lo = 1e300, hi = -1e300, sum = 0;
        for (float v : g.dense) { lo = v < lo ? v : lo; hi = v > hi ? v : hi; sum += v; }
        std::printf("{\"grid\": \"%s\", \"index_min\": [%d, %d, %d], \"dim\": [%d, %d, %d], \"voxel_size\": [%.17g, %.17g, %.17g], "
                    "\"scale\": [%.17g, %.17g, %.17g], \"translate\": [%.17g, %.17g, %.17g], \"world_min\": [%.17g, %.17g, %.17g], "
                    "\"world_max\": [%.17g, %.17g, %.17g], \"active_voxels\": %llu, \"background\": %.9g, \"min\": %.9g, \"max\": %.9g, \"sum\": %.17g}\n",
                    g.name.c_str(), g.indexMin[0], g.indexMin[1], g.indexMin[2], g.dim[0], g.dim[1], g.dim[2], g.voxelSize[0], g.voxelSize[1],
                    g.voxelSize[2], g.mat[0], g.mat[4], g.mat[8], g.vec[0], g.vec[1], g.vec[2], g.worldMin[0], g.worldMin[1], g.worldMin[2],
                    g.worldMax[0], g.worldMax[1], g.worldMax[2], g.activeVoxels, g.background, lo, hi, sum);
        if (!dump.empty()) {
            std::FILE *f = std::fopen(dump.c_str(), "wb");
            if (!f || std::fwrite(g.dense.data(), sizeof(float), g.dense.size(), f) != g.dense.size()) throw vspg::Error("cannot write " + dump);
            std::fclose(f);
        }
    } catch (const std::exception &e) {
        std::fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

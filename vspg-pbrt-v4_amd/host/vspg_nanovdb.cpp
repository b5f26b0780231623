// vspg_nanovdb.cpp -- see vspg_nanovdb.h (layout as understood; parity unpinned).
#include "vspg_nanovdb.h"

#include <cstdint>
#include <cstring>
#include <fstream>

#include "vspg_host.h"

namespace vspg {
namespace {

constexpr uint64_t kMagic = 0x304244566f6e614eull;  // "NanoVDB0"
constexpr size_t kHeaderBytes = 16, kMetaBytes = 176, kGridDataBytes = 672, kTreeDataBytes = 64;
constexpr size_t kRootHeaderBytes = 64, kRootTileBytes = 32;
constexpr size_t kUpperHeaderBytes = 8256, kLowerHeaderBytes = 1088, kLeafHeaderBytes = 96, kLeafBytes = 2144;

struct Reader {
    const std::vector<unsigned char> &buf;
    const std::string &file;
    template <class T>
    T at(size_t off) const {
        if (off > buf.size() || buf.size() - off < sizeof(T)) throw Error(file + ": truncated NanoVDB file (offset " + std::to_string(off) + ")");
        T v;
        std::memcpy(&v, buf.data() + off, sizeof(T));
        return v;
    }
};
std::string version_string(uint32_t v) { return std::to_string(v >> 21) + "." + std::to_string((v >> 10) & 0x7ff) + "." + std::to_string(v & 0x3ff); }
bool mask_on(const Reader &R, size_t mask_off, uint32_t n) { return (R.at<uint64_t>(mask_off + 8 * (n >> 6)) >> (n & 63)) & 1u; }

void decode_grid(const Reader &R, size_t blob, size_t blobSize, NanoVdbFloatGrid *g) {
    if (R.at<uint64_t>(blob) != kMagic) throw Error(R.file + ": grid \"" + g->name + "\": bad GridData magic");
    const uint32_t gridType = R.at<uint32_t>(blob + 636);
    if (gridType != 1) throw Error(R.file + ": grid \"" + g->name + "\" is not a FloatGrid (grid type " + std::to_string(gridType) + "); only float grids are read");
    for (int i = 0; i < 9; ++i) g->mat[i] = R.at<double>(blob + 296 + 88 + 8 * i);
    for (int i = 0; i < 3; ++i) g->vec[i] = R.at<double>(blob + 296 + 88 + 8 * 18 + 8 * i);
    // every node position is file-controlled: a position is `bytes` long and lies inside [blob, blob + blobSize) (no wrap-around)
    const size_t blobEnd = blob + blobSize;  // (the caller checked blob + gridSize <= file size)
    auto inside = [&](size_t off, size_t bytes) { return off >= blob && off <= blobEnd && blobEnd - off >= bytes; };
    // a child's position = its parent's + a signed 64-bit offset from the file
    auto child_at = [&](size_t parent, int64_t rel, const char *what) -> size_t {
        const bool ok = rel >= 0 ? (uint64_t)rel <= (uint64_t)(blobEnd - parent) : (uint64_t)(-(rel + 1)) + 1u <= (uint64_t)(parent - blob);
        if (!ok || ((uint64_t)rel & 31u)) throw Error(R.file + ": grid \"" + g->name + "\": " + what + " node offset outside the grid or misaligned");
        return rel >= 0 ? parent + (size_t)rel : parent - (size_t)(-(rel + 1)) - 1u;
    };
    if (blobSize < kGridDataBytes + kTreeDataBytes) throw Error(R.file + ": grid \"" + g->name + "\": grid shorter than its headers");
    const size_t tree = blob + kGridDataBytes;
    const uint64_t rootRel = R.at<uint64_t>(tree + 24);
    if (rootRel > blobSize) throw Error(R.file + ": grid \"" + g->name + "\": root node outside the grid");
    const size_t root = tree + (size_t)rootRel;
    if (!inside(root, kRootHeaderBytes)) throw Error(R.file + ": grid \"" + g->name + "\": root node outside the grid");
    g->background = R.at<float>(root + 28);
    const uint32_t tableSize = R.at<uint32_t>(root + 24);
    const long long nx = g->dim[0], ny = g->dim[1], nz = g->dim[2];
    g->dense.assign((size_t)(nx * ny * nz), g->background);
    auto put = [&](int x, int y, int z, float v) {
        const long long i = x - g->indexMin[0], j = y - g->indexMin[1], k = z - g->indexMin[2];
        if (i < 0 || j < 0 || k < 0 || i >= nx || j >= ny || k >= nz) return;
        g->dense[(size_t)((k * ny + j) * nx + i)] = v;
    };
    auto fill_box = [&](int x0, int y0, int z0, int size, float v) {  // an active tile: constant over size^3 voxels, clipped to the index bbox
        const long long o[3] = {x0, y0, z0}, n[3] = {nx, ny, nz};
        long long lo[3], hi[3];  // (64-bit: o + size and indexMin + dim may pass INT32_MAX)
        for (int k = 0; k < 3; ++k) {
            lo[k] = o[k] > g->indexMin[k] ? o[k] : (long long)g->indexMin[k];
            hi[k] = o[k] + size < g->indexMin[k] + n[k] ? o[k] + size : g->indexMin[k] + n[k];
        }
        for (long long z = lo[2]; z < hi[2]; ++z)
            for (long long y = lo[1]; y < hi[1]; ++y)
                for (long long x = lo[0]; x < hi[0]; ++x) put((int)x, (int)y, (int)z, v);
    };
    auto leaf = [&](size_t off, int x0, int y0, int z0) {
        if (!inside(off, kLeafBytes)) throw Error(R.file + ": grid \"" + g->name + "\": leaf node outside the grid");
        for (uint32_t n = 0; n < 512; ++n) {  // every value of a leaf is defined (inactive ones hold the background or a tile value)
            const float v = R.at<float>(off + kLeafHeaderBytes + 4 * n);
            put(x0 + (int)(n >> 6), y0 + (int)((n >> 3) & 7), z0 + (int)(n & 7), v);
        }
    };
    auto lower = [&](size_t off, int x0, int y0, int z0) {
        const size_t vmask = off + 32, cmask = off + 32 + 512, table = off + kLowerHeaderBytes;
        if (!inside(off, kLowerHeaderBytes + 4096 * 8)) throw Error(R.file + ": grid \"" + g->name + "\": lower node outside the grid");
        for (uint32_t n = 0; n < 4096; ++n) {
            const int x = x0 + (int)(n >> 8) * 8, y = y0 + (int)((n >> 4) & 15) * 8, z = z0 + (int)(n & 15) * 8;
            if (mask_on(R, cmask, n)) leaf(child_at(off, R.at<int64_t>(table + 8 * n), "leaf"), x, y, z);
            else if (mask_on(R, vmask, n)) fill_box(x, y, z, 8, R.at<float>(table + 8 * n));
        }
    };
    auto upper = [&](size_t off, int x0, int y0, int z0) {
        const size_t vmask = off + 32, cmask = off + 32 + 4096, table = off + kUpperHeaderBytes;
        if (!inside(off, kUpperHeaderBytes + 32768 * 8)) throw Error(R.file + ": grid \"" + g->name + "\": upper node outside the grid");
        for (uint32_t n = 0; n < 32768; ++n) {
            const int x = x0 + (int)(n >> 10) * 128, y = y0 + (int)((n >> 5) & 31) * 128, z = z0 + (int)(n & 31) * 128;
            if (mask_on(R, cmask, n)) lower(child_at(off, R.at<int64_t>(table + 8 * n), "lower"), x, y, z);
            else if (mask_on(R, vmask, n)) fill_box(x, y, z, 128, R.at<float>(table + 8 * n));
        }
    };
    if (!inside(root, kRootHeaderBytes) || (blobEnd - root - kRootHeaderBytes) / kRootTileBytes < tableSize)
        throw Error(R.file + ": grid \"" + g->name + "\": root table outside the grid");
    for (uint32_t t = 0; t < tableSize; ++t) {
        const size_t tile = root + kRootHeaderBytes + kRootTileBytes * t;
        const uint64_t key = R.at<uint64_t>(tile);
        const int64_t child = R.at<int64_t>(tile + 8);
        // the key holds uint32(coordinate) >> 12 -- 20 significant bits, two's complement -- in fields of 21 bits
        auto field = [](uint64_t k, int shift) { int32_t v = (int32_t)((k >> shift) & 0xfffffu); if (v & 0x80000) v |= ~0xfffff; return v * 4096; };
        const int x0 = field(key, 42), y0 = field(key, 21), z0 = field(key, 0);
        if (child != 0) upper(child_at(root, child, "upper"), x0, y0, z0);
        else if (R.at<uint32_t>(tile + 16)) fill_box(x0, y0, z0, 4096, R.at<float>(tile + 20));
    }
}

}  // namespace

bool ReadNanoVdbFloatGrid(const std::string &filename, const std::string &gridName, NanoVdbFloatGrid *out) {
    std::ifstream f(filename, std::ios::binary);
    if (!f) throw Error(filename + ": cannot open");
    std::vector<unsigned char> buf((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    const Reader R{buf, filename};
    size_t pos = 0;
    while (pos + kHeaderBytes <= buf.size()) {  // segments
        if (R.at<uint64_t>(pos) != kMagic) throw Error(filename + ": not a NanoVDB file (magic number)");
        const uint32_t version = R.at<uint32_t>(pos + 8);
        if ((version >> 21) != 32)
            throw Error(filename + ": NanoVDB file version " + version_string(version) + "; this reader knows the 32.x layout only (convert with the reference's "
                        "nanovdb2pbrt, INTEGRATION.md 2)");
        const uint16_t gridCount = R.at<uint16_t>(pos + 12), codec = R.at<uint16_t>(pos + 14);
        size_t meta = pos + kHeaderBytes;
        struct Entry { size_t meta; std::string name; uint64_t gridSize, fileSize; };
        std::vector<Entry> entries;
        for (uint16_t i = 0; i < gridCount; ++i) {
            const uint32_t nameSize = R.at<uint32_t>(meta + 136);
            if (meta > buf.size() || buf.size() - meta < kMetaBytes || buf.size() - meta - kMetaBytes < nameSize) throw Error(filename + ": truncated NanoVDB file (grid names)");
            std::string name((const char *)buf.data() + meta + kMetaBytes, nameSize ? nameSize - 1 : 0);
            entries.push_back(Entry{meta, name, R.at<uint64_t>(meta), R.at<uint64_t>(meta + 8)});
            meta += kMetaBytes + nameSize;
        }
        size_t blob = meta;
        for (const Entry &e : entries) {
            if (e.name == gridName) {
                if (codec != 0) throw Error(filename + ": compressed NanoVDB file (codec " + std::to_string(codec) + "); only uncompressed files are read");
                if (blob > buf.size() || buf.size() - blob < e.gridSize) throw Error(filename + ": truncated NanoVDB file (grid \"" + gridName + "\")");
                out->name = e.name;
                out->activeVoxels = R.at<uint64_t>(e.meta + 24);
                for (int k = 0; k < 3; ++k) {
                    out->worldMin[k] = R.at<double>(e.meta + 40 + 8 * k);
                    out->worldMax[k] = R.at<double>(e.meta + 64 + 8 * k);
                    out->indexMin[k] = R.at<int32_t>(e.meta + 88 + 4 * k);
                    const int64_t d = (int64_t)R.at<int32_t>(e.meta + 100 + 4 * k) - out->indexMin[k] + 1;  // inclusive bbox (nanovdb2pbrt.cpp:103-105); 64-bit: max - min may pass INT32_MAX
                    out->voxelSize[k] = R.at<double>(e.meta + 112 + 8 * k);
                    if (d <= 0) throw Error(filename + ": grid \"" + gridName + "\" has an empty index bounding box");
                    if (d > (1 << 20)) throw Error(filename + ": grid \"" + gridName + "\": index bounding box too large for a dense copy");
                    out->dim[k] = (int)d;
                }
                if ((double)out->dim[0] * out->dim[1] * out->dim[2] > 4e9) throw Error(filename + ": grid \"" + gridName + "\": index bounding box too large for a dense copy");
                decode_grid(R, blob, e.gridSize, out);
                return true;
            }
            if (e.fileSize > buf.size() - (blob < buf.size() ? blob : buf.size())) { blob = buf.size(); break; }  // (a later segment cannot start inside a truncated grid)
            blob += e.fileSize;
        }
        pos = blob;
    }
    return false;
}

}  // namespace vspg

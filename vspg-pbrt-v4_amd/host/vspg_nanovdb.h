// vspg_nanovdb.h -- reader for NanoVDB grid files (".nvdb"), FloatGrid only (the reference reads them with
// nanovdb::io::readGrid, media.cpp:526-547, and samples them through NanoVDBMedium, media.h:657-753).
//
// PARITY UNPINNED, and more than that: the NanoVDB headers are an absent submodule of the reference (openvdb pinned at
// 414bed84..., CMakeLists.txt:67) and no .nvdb file exists in this environment, so the layout below is NanoVDB's published
// 32.x file / memory layout AS UNDERSTOOD BY THIS BUILD, validated only against a writer of the same understanding
// (tests/nvdb_writer.py).  A file of another major version, a compressed file (ZIP / BLOSC codecs) or a non-float grid is
// refused with a message that says which; nothing is guessed.  The supported path that IS checkable end to end remains the
// reference's own converter (cmd/nanovdb2pbrt.cpp -> "uniformgrid" parameters).
//
// Layout read (little endian; NANOVDB_DATA_ALIGNMENT 32):
//   file    = segment*;  segment = Header{u64 magic "NanoVDB0", u32 version (major << 21 | minor << 10 | patch), u16 gridCount,
//             u16 codec} MetaData[gridCount] (176 B each, followed by the grid's name, nameSize bytes incl. NUL) grid blobs
//   MetaData= u64 gridSize, fileSize, nameKey, voxelCount; u32 gridType, gridClass; f64 worldBBox[6]; i32 indexBBox[6];
//             f64 voxelSize[3]; u32 nameSize; u32 nodeCount[4]; u32 tileCount[3]; u16 codec; u16 pad; u32 version
//   blob    = GridData (672 B: u64 magic, checksum; u32 version, flags, gridIndex, gridCount; u64 gridSize; char name[256];
//             Map {f32 mat[9], invMat[9], vec[3], taper; f64 mat[9], invMat[9], vec[3], taper}; f64 worldBBox[6]; f64 voxelSize[3];
//             u32 gridClass, gridType; i64 blindOffset; u32 blindCount; pad)
//             TreeData (64 B: u64 nodeOffset[4] {leaf, lower, upper, root} relative to TreeData; u32 nodeCount[3]; u32 tileCount[3];
//             u64 voxelCount)
//             RootData {i32 bbox[6]; u32 tableSize; f32 background, min, max, avg, stddev; pad to 64} Tile[tableSize]
//               Tile (32 B) = u64 key (x >> 12 << 42 | y >> 12 << 21 | z >> 12); i64 child (byte offset from RootData, 0 = tile); u32 state; f32 value
//             Upper  InternalData<5> {i32 bbox[6]; u64 flags; u64 valueMask[512]; u64 childMask[512]; f32 min, max, avg, stddev; pad to 8256}
//               Tile[32768] (8 B: f32 value | i64 child = byte offset from this node);  4096^3 voxels, child index (x >> 7 & 31) << 10 | (y >> 7 & 31) << 5 | (z >> 7 & 31)
//             Lower  InternalData<4> {... masks of 64 words ...; pad to 1088} Tile[4096];  128^3 voxels, child index (x >> 3 & 15) << 8 | (y >> 3 & 15) << 4 | (z >> 3 & 15)
//             Leaf   {i32 bboxMin[3]; u8 bboxDif[3]; u8 flags; u64 valueMask[8]; f32 min, max, avg, stddev; f32 values[512]} (2144 B); voxel (x & 7) << 6 | (y & 7) << 3 | (z & 7)
#pragma once
#include <string>
#include <vector>

namespace vspg {

struct NanoVdbFloatGrid {
    std::string name;
    int indexMin[3], dim[3];        // index bounding box (inclusive min) and its extent in voxels
    double voxelSize[3];
    double mat[9], vec[3];          // indexToWorld: world = mat * index + vec
    double worldMin[3], worldMax[3];
    unsigned long long activeVoxels;
    float background;
    std::vector<float> dense;       // dim[0] * dim[1] * dim[2] values, x fastest (what cmd/nanovdb2pbrt.cpp:97-126 dumps)
};

// Reads grid `gridName` of `filename`.  Returns false when the file holds no grid of that name (the reference's readGrid returns
// an empty handle then); throws vspg::Error for everything it cannot read.
bool ReadNanoVdbFloatGrid(const std::string &filename, const std::string &gridName, NanoVdbFloatGrid *out);

}  // namespace vspg

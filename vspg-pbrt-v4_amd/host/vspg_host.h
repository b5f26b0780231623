// vspg_host.h -- C++ host adapter: the reference's Integrator / Medium plugin surface for the
// GuidedVolPathVSPG path, implemented on top of the C-ABI (include/vspg.h).
//
// Mirrors (names, argument meaning, defaults, error behaviour):
//   ParameterDictionary::GetOne*/ReportUnused     src/pbrt/paramdict.h, paramdict.cpp:642-664
//   Integrator::Create(name, params, ...)          src/pbrt/cpu/integrators.cpp:3711-3768
//   GuidedVolPathVSPGIntegrator::Create / Render / PostProcessWave / ToString
//                                                  src/pbrt/cpu/guidedvolpathvspgintegrator.cpp:1260-1322, 230-260
//   ImageTileIntegrator::Render wave loop          src/pbrt/cpu/integrators.cpp:75-269 (1-spp waves :239)
//   HomogeneousMedium::Create / Medium::Create     src/pbrt/media.cpp:167-206, 816-841
// The reference aborts the process on fatal errors (ErrorExit); this adapter throws
// vspg::Error with the same trigger conditions so an embedding application can decide.
#pragma once
#include <map>
#include <cstdio>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vspg.h"

namespace vspg {

struct Error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// A small typed parameter dictionary: the subset of pbrt's ParameterDictionary the path reads.
class ParameterDictionary {
  public:
    ParameterDictionary &Int(const std::string &n, int v);
    ParameterDictionary &Float(const std::string &n, float v);
    ParameterDictionary &Bool(const std::string &n, bool v);
    ParameterDictionary &String(const std::string &n, const std::string &v);
    ParameterDictionary &RGB(const std::string &n, float r, float g, float b);
    ParameterDictionary &Point3(const std::string &n, float x, float y, float z);
    ParameterDictionary &FloatArray(const std::string &n, std::vector<float> v);
    ParameterDictionary &IntArray(const std::string &n, std::vector<int> v);
    ParameterDictionary &Point3Array(const std::string &n, std::vector<float> xyz);
    // The reference's scene-file parameter-list syntax ("type name" value | [ values ]), e.g. the block
    // cmd/nanovdb2pbrt.cpp:97-126 prints for a grid:
    //   "integer nx" 64 "integer ny" 64 "integer nz" 32 "point3 p0" [ -1 -1 0 ] "point3 p1" [ 1 1 1 ] "float density" [ ... ]
    // Types: integer (several -> IntArray), float (one value -> Float, several -> FloatArray), bool, string, rgb,
    // point3 (several points -> Point3Array); point2 / vector3 / normal3 arrays are kept as float arrays under their name.
    static ParameterDictionary Parse(const std::string &text);

    int GetOneInt(const std::string &n, int def) const;
    float GetOneFloat(const std::string &n, float def) const;
    bool GetOneBool(const std::string &n, bool def) const;
    std::string GetOneString(const std::string &n, const std::string &def) const;
    // returns false if absent
    bool GetOneRGB(const std::string &n, float rgb[3]) const;
    bool GetOnePoint3(const std::string &n, float p[3]) const;
    std::vector<float> GetFloatArray(const std::string &n) const;  // a single "float" value is a 1-element array
    std::vector<int> GetIntArray(const std::string &n) const;      // a single "integer" value is a 1-element array
    std::vector<float> GetPoint3Array(const std::string &n) const; // x y z x y z ...; a single "point3" is one point
    bool Has(const std::string &n) const { return values.count(n) != 0; }
    // paramdict.cpp:642-664: any parameter that was never looked up is a fatal error
    void ReportUnused() const;

  private:
    struct Value {
        char type;  // i f b s c p a(rray) I(nt array) P(oint array)
        int i = 0;
        std::vector<int> iarr;
        float f[3] = {0, 0, 0};
        std::string s;
        std::vector<float> arr;
        mutable bool lookedUp = false;
    };
    const Value *find(const std::string &n, char type) const;
    std::map<std::string, Value> values;
};

// Medium::Create (media.cpp:816-841): "homogeneous" (HomogeneousMedium::Create :167-206) and "uniformgrid"
// (GridMedium::Create :272-361; the density array is copied into *densityStorage, which must outlive the
// renderer creation -- VspgMedium.density points into it; likewise the "Lescale" and "temperature" grids); "nanovdb"
// (NanoVDBMedium::Create :683-734; its temperature grid goes to temperatureStorage, or to leScaleStorage when that is all the caller
// passed); other names -> Error: outside scope
VspgMedium CreateMedium(const std::string &name, const ParameterDictionary &parameters,
                        std::vector<float> *densityStorage = nullptr, std::vector<float> *leScaleStorage = nullptr,
                        std::vector<float> *temperatureStorage = nullptr);

struct Film {
    int xres = 0, yres = 0;
    std::vector<float> rgbw;  // W*H*4: sum w*r, sum w*g, sum w*b, sum w
    // RGBFilm::GetPixelRGB (film.h:269-287) without the output colour transform
    void GetPixelRGB(int x, int y, float rgb[3]) const;
    void WritePFM(const std::string &filename) const;
};

// Guiding-cache file (openpgl::cpp::Field::Store / Field(device, file), guidedvolpathvspgintegrator.cpp:
// 117-128, 210-213).  Format "VSPGFLD1", little endian:
//   char magic[8] = "VSPGFLD1"; uint32 lobes (= VSPG_FIELD_LOBES); uint32 reserved;
//   for field in {surface, volume}: uint32 n_nodes, n_regions;
//   for field in {surface, volume}: VspgKdNode nodes[n_nodes]; VspgFieldRegion regions[n_regions];
// (structs exactly as declared in include/vspg.h).
struct GuidingCache {
    std::vector<VspgKdNode> nodes[2];
    std::vector<VspgFieldRegion> regions[2];
    void Write(const std::string &filename) const;
    static GuidingCache Read(const std::string &filename);
};
struct GuidingCacheSettings {  // "storeGuidingCache" / "loadGuidingCache" / "guidingCacheFileName"
    bool store = false, load = false;
    std::string fileName;
};

// TrBuffer (src/pbrt/cpu/trbuffer.h:17-104): the per-pixel transmittance estimates of the primary rays, kept between
// runs for NDS+.  The reference stores it through pbrt's Image class (format by file extension, normally OpenEXR, an
// absent submodule); this adapter reads and writes the PFM raster of that class (util/image.cpp:1756-1800 WritePFM:
// "PF", width height, scale -1 = little endian, RGB float32, scanlines bottom to top).  PFM carries no channel
// names ("Transmittance.R/G/B" in the reference's EXR), so the file is this adapter's own persistence, readable by
// any PFM tool.  Other extensions are refused.
struct TrBuffer {
    int xres = 0, yres = 0;
    std::vector<float> rgb;  // row-major, top row first, 3 floats per pixel
    void Store(const std::string &filename) const;
    static TrBuffer Load(const std::string &filename);
};
// The image-space VSP buffer between runs ("storeISGBuffer" / "loadISGBuffer" / "isgBufferFileName",
// guidedvolpathvspgintegrator.cpp:151-159, 214-216).  OpenPGL's own file format is not part of the reference tree; this
// adapter keeps the one plane the path reads -- the per-pixel volume-scatter-probability estimate -- as a single-channel
// PFM ("Pf", same raster conventions as TrBuffer; a value outside [0,1] = no estimate for the pixel).
struct VspBuffer {
    int xres = 0, yres = 0;
    bool ready = false;
    std::vector<float> vsp;  // row-major, top row first
    void Store(const std::string &filename) const;
    static VspBuffer Load(const std::string &filename);
};
struct IsgBufferSettings {
    bool store = false, load = false;
    std::string fileName;
};
struct TrBufferSettings {  // "storeTrBuffer" / "loadTrBuffer" / "trBufferFileName"
    bool store = false, load = false;
    std::string fileName;
};

class Integrator {
  public:
    virtual ~Integrator() = default;
    virtual void Render() = 0;
    virtual std::string ToString() const = 0;
    // Integrator::Create: "guidedvolpathvspg"; "guidedvolpath" is accepted as an alias ONLY when
    // the dictionary carries "vspguiding" (the option BASELINE.json names), see SURVEY.md 0.1
    static std::unique_ptr<Integrator> Create(const std::string &name, const ParameterDictionary &parameters,
                                              const VspgScene &scene, int xres, int yres, int pixelSamples,
                                              int seed = 0, int device = 0);
};

class GuidedVolPathVSPGIntegrator : public Integrator {
  public:
    static std::unique_ptr<GuidedVolPathVSPGIntegrator> Create(const ParameterDictionary &parameters,
                                                               const VspgScene &scene, int xres, int yres,
                                                               int pixelSamples, int seed, int device);
    GuidedVolPathVSPGIntegrator(const VspgIntegratorParams &p, const VspgScene &scene, int xres, int yres,
                                int pixelSamples, int seed, int device, const GuidingCacheSettings &cache = {},
                                const TrBufferSettings &tr = {}, const IsgBufferSettings &isg = {});
    ~GuidedVolPathVSPGIntegrator() override;
    void Render() override;       // wave loop: 1 spp per wave, PostProcessWave after each
    void PostProcessWave();       // guidedvolpathvspgintegrator.cpp:230-260
    std::string ToString() const override;
    Film GetFilm();
    VspgCounters Counters();
    VspgTrainStats TrainingStats();      // guideTraining / guiding_field->GetIteration()
    GuidingCache GetGuidingCache();      // the field as it stands (trained in-loop or loaded)
    TrBuffer GetTrBuffer();              // the transmittance buffer as it stands (recorded or loaded)
    VspBuffer GetVspBuffer();            // the image-space VSP estimate as it stands
    const VspgIntegratorParams &Params() const { return params; }
    // one JSON line per wave while rendering: {wave, ms, paths, segments, density_queries, kernel} (not owned; NULL = off)
    void SetWaveLog(std::FILE *f) { waveLog = f; }

  private:
    std::FILE *waveLog = nullptr;
    VspgIntegratorParams params;
    VspgRenderConfig cfg;
    VspgRenderer *renderer = nullptr;
    int spp;
    GuidingCacheSettings cacheSettings;
    TrBufferSettings trSettings;
    IsgBufferSettings isgSettings;
};

// parameter parsing only (no device): used by Create and by the CPU self test
VspgIntegratorParams ParseIntegratorParams(const ParameterDictionary &parameters, GuidingCacheSettings *cache = nullptr,
                                           TrBufferSettings *tr = nullptr, IsgBufferSettings *isg = nullptr);

}  // namespace vspg

#include "vspg_host.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>

namespace vspg {

// ---------------------------------------------------------------------------------------
ParameterDictionary &ParameterDictionary::Int(const std::string &n, int v) { Value x; x.type = 'i'; x.i = v; values[n] = x; return *this; }
ParameterDictionary &ParameterDictionary::Float(const std::string &n, float v) { Value x; x.type = 'f'; x.f[0] = v; values[n] = x; return *this; }
ParameterDictionary &ParameterDictionary::Bool(const std::string &n, bool v) { Value x; x.type = 'b'; x.i = v; values[n] = x; return *this; }
ParameterDictionary &ParameterDictionary::String(const std::string &n, const std::string &v) { Value x; x.type = 's'; x.s = v; values[n] = x; return *this; }
ParameterDictionary &ParameterDictionary::RGB(const std::string &n, float r, float g, float b) { Value x; x.type = 'c'; x.f[0] = r; x.f[1] = g; x.f[2] = b; values[n] = x; return *this; }

const ParameterDictionary::Value *ParameterDictionary::find(const std::string &n, char type) const {
    auto it = values.find(n);
    if (it == values.end()) return nullptr;
    if (it->second.type != type)
        throw Error("parameter \"" + n + "\" has the wrong type");
    it->second.lookedUp = true;
    return &it->second;
}
int ParameterDictionary::GetOneInt(const std::string &n, int def) const { auto v = find(n, 'i'); return v ? v->i : def; }
float ParameterDictionary::GetOneFloat(const std::string &n, float def) const { auto v = find(n, 'f'); return v ? v->f[0] : def; }
bool ParameterDictionary::GetOneBool(const std::string &n, bool def) const { auto v = find(n, 'b'); return v ? v->i != 0 : def; }
std::string ParameterDictionary::GetOneString(const std::string &n, const std::string &def) const { auto v = find(n, 's'); return v ? v->s : def; }
bool ParameterDictionary::GetOneRGB(const std::string &n, float rgb[3]) const {
    auto v = find(n, 'c');
    if (!v) return false;
    rgb[0] = v->f[0]; rgb[1] = v->f[1]; rgb[2] = v->f[2];
    return true;
}
void ParameterDictionary::ReportUnused() const {
    for (const auto &kv : values)
        if (!kv.second.lookedUp) throw Error("\"" + kv.first + "\": unused parameter.");
}

// ---------------------------------------------------------------------------------------
VspgMedium CreateMedium(const std::string &name, const ParameterDictionary &p) {
    VspgMedium m;
    std::memset(&m, 0, sizeof m);
    if (name != "homogeneous")
        throw Error("medium \"" + name + "\": only \"homogeneous\" is inside this build's scope");
    // HomogeneousMedium::Create (media.cpp:167-206)
    if (!p.GetOneString("preset", "").empty()) throw Error("medium \"preset\" tables are outside this build's scope");
    float sa[3] = {1.f, 1.f, 1.f}, ss[3] = {1.f, 1.f, 1.f}, le[3] = {0, 0, 0};  // defaults: ConstantSpectrum(1)
    p.GetOneRGB("sigma_a", sa);
    p.GetOneRGB("sigma_s", ss);
    bool hasLe = p.GetOneRGB("Le", le);
    float LeScale = p.GetOneFloat("Lescale", 1.f);
    float sigmaScale = p.GetOneFloat("scale", 1.f);
    float g = p.GetOneFloat("g", 0.0f);
    p.ReportUnused();
    m.type = VSPG_MEDIUM_HOMOGENEOUS;
    for (int i = 0; i < 3; ++i) {
        m.sigma_a[i] = sa[i] * sigmaScale;   // RGBUnboundedSpectrum::Scale (media.h:237-239)
        m.sigma_s[i] = ss[i] * sigmaScale;
        m.Le[i] = hasLe ? le[i] * LeScale : 0.f;
    }
    m.g = g;
    return m;
}

// ---------------------------------------------------------------------------------------
VspgIntegratorParams ParseIntegratorParams(const ParameterDictionary &parameters, GuidingCacheSettings *cache) {
    // GuidedVolPathVSPGIntegrator::Create (guidedvolpathvspgintegrator.cpp:1260-1322)
    VspgIntegratorParams p;
    vspg_integrator_params_default(&p);
    p.maxdepth = parameters.GetOneInt("maxdepth", 5);
    p.minrrdepth = parameters.GetOneInt("minrrdepth", 1);
    p.usenee = parameters.GetOneBool("usenee", true);
    p.surfaceguiding = parameters.GetOneBool("surfaceguiding", true);
    p.volumeguiding = parameters.GetOneBool("volumeguiding", true);
    p.surfaceguidingtype = parameters.GetOneString("surfaceguidingtype", "ris") == "mis" ? VSPG_GUIDE_MIS : VSPG_GUIDE_RIS;
    p.volumeguidingtype = parameters.GetOneString("volumeguidingtype", "mis") == "mis" ? VSPG_GUIDE_MIS : VSPG_GUIDE_RIS;
    GuidingCacheSettings cs;
    cs.store = parameters.GetOneBool("storeGuidingCache", false);
    cs.load = parameters.GetOneBool("loadGuidingCache", false);
    cs.fileName = parameters.GetOneString("guidingCacheFileName", "");
    if (cache) *cache = cs;
    p.vspguiding = parameters.GetOneBool("vspguiding", true);
    p.vspprimaryguiding = parameters.GetOneBool("vspprimaryguiding", true);
    p.vspsecondaryguiding = parameters.GetOneBool("vspsecondaryguiding", true);
    p.vspmisratio = parameters.GetOneFloat("vspmisratio", 0.5f);
    std::string crit = parameters.GetOneString("vspcriterion", "variance");
    if (crit == "Contribution" || crit == "contribution") p.vspcriterion = VSPG_VSP_CONTRIBUTION;
    else if (crit == "Variance" || crit == "variance") p.vspcriterion = VSPG_VSP_VARIANCE;
    std::string meth = parameters.GetOneString("vspsamplingmethod", "resampling");
    if (meth == "Resampling" || meth == "resampling") p.vspsamplingmethod = VSPG_VSP_RESAMPLING;
    else if (meth == "NDS" || meth == "nds") p.vspsamplingmethod = VSPG_VSP_NDS;
    p.collisionProbabilityBias = parameters.GetOneBool("collisionProbabilityBias", false);
    bool storeISG = parameters.GetOneBool("storeISGBuffer", false);
    bool loadISG = parameters.GetOneBool("loadISGBuffer", false);
    (void)parameters.GetOneString("isgBufferFileName", "");
    bool storeTr = parameters.GetOneBool("storeTrBuffer", false);
    bool loadTr = parameters.GetOneBool("loadTrBuffer", false);
    (void)parameters.GetOneString("trBufferFileName", "");
    p.rrguiding = parameters.GetOneBool("rrguiding", false);
    (void)parameters.GetOneBool("surfacerrguiding", true);
    (void)parameters.GetOneBool("volumerrguiding", true);
    std::string ls = parameters.GetOneString("lightsampler", "bvh");
    if (ls == "uniform") p.lightsampler = VSPG_LIGHTSAMPLER_UNIFORM;
    else if (ls == "power") p.lightsampler = VSPG_LIGHTSAMPLER_POWER;
    else if (ls == "bvh") p.lightsampler = VSPG_LIGHTSAMPLER_BVH;
    else throw Error("Light sample distribution type \"" + ls + "\" unknown.");
    p.regularize = parameters.GetOneBool("regularize", false);
    if (storeISG || loadISG || storeTr || loadTr)
        throw Error("ISG-buffer / Tr-buffer persistence is outside this build's scope");
    if ((cs.store || cs.load) && cs.fileName.empty()) throw Error("storeGuidingCache / loadGuidingCache need \"guidingCacheFileName\"");
    return p;
}

std::unique_ptr<Integrator> Integrator::Create(const std::string &name, const ParameterDictionary &parameters,
                                               const VspgScene &scene, int xres, int yres, int pixelSamples, int seed,
                                               int device) {
    std::unique_ptr<Integrator> integrator;
    if (name == "guidedvolpathvspg")
        integrator = GuidedVolPathVSPGIntegrator::Create(parameters, scene, xres, yres, pixelSamples, seed, device);
    else if (name == "guidedvolpath") {
        // the reference's "guidedvolpath" has no VSP guiding and rejects "vspguiding" as unused;
        // BASELINE.json nevertheless names `guidedvolpath` + vspguiding: accept that spelling as
        // an alias of the VSPG integrator, and only that spelling
        if (!parameters.GetOneBool("vspguiding", false))
            throw Error("integrator \"guidedvolpath\" without \"vspguiding\" true is outside this build's scope "
                        "(directional guiding only); use \"guidedvolpathvspg\"");
        integrator = GuidedVolPathVSPGIntegrator::Create(parameters, scene, xres, yres, pixelSamples, seed, device);
    } else
        throw Error(name + ": integrator type unknown.");  // integrators.cpp:3760
    parameters.ReportUnused();  // integrators.cpp:3766
    return integrator;
}

std::unique_ptr<GuidedVolPathVSPGIntegrator> GuidedVolPathVSPGIntegrator::Create(const ParameterDictionary &parameters,
                                                                                 const VspgScene &scene, int xres, int yres,
                                                                                 int pixelSamples, int seed, int device) {
    GuidingCacheSettings cache;
    VspgIntegratorParams p = ParseIntegratorParams(parameters, &cache);
    return std::make_unique<GuidedVolPathVSPGIntegrator>(p, scene, xres, yres, pixelSamples, seed, device, cache);
}

GuidedVolPathVSPGIntegrator::GuidedVolPathVSPGIntegrator(const VspgIntegratorParams &p, const VspgScene &scene, int xres,
                                                         int yres, int pixelSamples, int seed, int device,
                                                         const GuidingCacheSettings &cache)
    : params(p), spp(pixelSamples), cacheSettings(cache) {
    std::memset(&cfg, 0, sizeof cfg);
    cfg.xres = xres; cfg.yres = yres; cfg.spp = pixelSamples; cfg.seed = seed;
    cfg.shard_index = 0; cfg.shard_count = 1; cfg.device = device;
    int rc = vspg_renderer_create(&scene, &params, &cfg, &renderer);
    if (rc != 0) throw Error(std::string("GuidedVolPathVSPGIntegrator: ") + vspg_last_error());
    if (cacheSettings.load) {  // :117-122: a loaded cache is used as is, guideTraining = false
        GuidingCache gc = GuidingCache::Read(cacheSettings.fileName);
        VspgField f[2];
        for (int i = 0; i < 2; ++i)
            f[i] = VspgField{(int32_t)gc.nodes[i].size(), (int32_t)gc.regions[i].size(), gc.nodes[i].data(), gc.regions[i].data()};
        if (vspg_renderer_set_guiding_field(renderer, &f[0], &f[1], nullptr) != 0) {
            std::string msg = vspg_last_error();
            vspg_renderer_destroy(renderer);
            throw Error("GuidedVolPathVSPGIntegrator: loading \"" + cacheSettings.fileName + "\": " + msg);
        }
    }
}
GuidedVolPathVSPGIntegrator::~GuidedVolPathVSPGIntegrator() {
    if (cacheSettings.store) {  // :210-213
        try {
            GetGuidingCache().Write(cacheSettings.fileName);
        } catch (const Error &e) {
            std::fprintf(stderr, "GuidedVolPathVSPGIntegrator: storing the guiding cache failed: %s\n", e.what());
        }
    }
    vspg_renderer_destroy(renderer);
}
VspgTrainStats GuidedVolPathVSPGIntegrator::TrainingStats() {
    VspgTrainStats st;
    if (vspg_renderer_training_stats(renderer, &st, nullptr) != 0) throw Error(vspg_last_error());
    return st;
}
GuidingCache GuidedVolPathVSPGIntegrator::GetGuidingCache() {
    GuidingCache gc;
    for (int f = 0; f < 2; ++f) {
        int32_t nn = 0, nr = 0;
        if (vspg_renderer_get_guiding_field(renderer, f, nullptr, nullptr, &nn, &nr, nullptr) != 0) throw Error(vspg_last_error());
        gc.nodes[f].resize((size_t)nn);
        gc.regions[f].resize((size_t)nr);
        if (vspg_renderer_get_guiding_field(renderer, f, gc.nodes[f].data(), gc.regions[f].data(), &nn, &nr, nullptr) != 0)
            throw Error(vspg_last_error());
    }
    return gc;
}

void GuidingCache::Write(const std::string &filename) const {
    std::ofstream f(filename, std::ios::binary);
    if (!f) throw Error("cannot open \"" + filename + "\" for writing");
    const char magic[8] = {'V', 'S', 'P', 'G', 'F', 'L', 'D', '1'};
    const uint32_t head[2] = {VSPG_FIELD_LOBES, 0u};
    f.write(magic, 8);
    f.write(reinterpret_cast<const char *>(head), sizeof head);
    for (int i = 0; i < 2; ++i) {
        const uint32_t n[2] = {(uint32_t)nodes[i].size(), (uint32_t)regions[i].size()};
        f.write(reinterpret_cast<const char *>(n), sizeof n);
    }
    for (int i = 0; i < 2; ++i) {
        f.write(reinterpret_cast<const char *>(nodes[i].data()), (std::streamsize)(nodes[i].size() * sizeof(VspgKdNode)));
        f.write(reinterpret_cast<const char *>(regions[i].data()), (std::streamsize)(regions[i].size() * sizeof(VspgFieldRegion)));
    }
    if (!f) throw Error("short write to \"" + filename + "\"");
}
GuidingCache GuidingCache::Read(const std::string &filename) {
    std::ifstream f(filename, std::ios::binary);
    if (!f) throw Error("cannot open guiding cache \"" + filename + "\"");
    char magic[8];
    uint32_t head[2], n[2][2];
    f.read(magic, 8);
    f.read(reinterpret_cast<char *>(head), sizeof head);
    f.read(reinterpret_cast<char *>(n), sizeof n);
    if (!f || std::memcmp(magic, "VSPGFLD1", 8) != 0) throw Error("\"" + filename + "\" is not a VSPGFLD1 guiding cache");
    if (head[0] != VSPG_FIELD_LOBES) throw Error("guiding cache was written with a different lobe count");
    GuidingCache gc;
    for (int i = 0; i < 2; ++i) {
        if (n[i][0] > (1u << 24) || n[i][1] > (1u << 24)) throw Error("guiding cache header is corrupt");
        gc.nodes[i].resize(n[i][0]);
        gc.regions[i].resize(n[i][1]);
    }
    for (int i = 0; i < 2; ++i) {
        f.read(reinterpret_cast<char *>(gc.nodes[i].data()), (std::streamsize)(gc.nodes[i].size() * sizeof(VspgKdNode)));
        f.read(reinterpret_cast<char *>(gc.regions[i].data()), (std::streamsize)(gc.regions[i].size() * sizeof(VspgFieldRegion)));
    }
    if (!f) throw Error("guiding cache \"" + filename + "\" is truncated");
    return gc;
}

void GuidedVolPathVSPGIntegrator::Render() {
    // ImageTileIntegrator::Render (integrators.cpp:123-239): waves of 1 spp, PostProcessWave each
    int waveStart = 0, waveEnd = 1, nextWaveSize = 1;
    while (waveStart < spp) {
        if (vspg_render_wave(renderer, waveStart, waveEnd, nullptr) != 0) throw Error(vspg_last_error());
        PostProcessWave();
        waveStart = waveEnd;
        waveEnd = std::min(spp, waveEnd + nextWaveSize);
    }
}
void GuidedVolPathVSPGIntegrator::PostProcessWave() {
    if (vspg_post_process_wave(renderer, nullptr) != 0) throw Error(vspg_last_error());
}
std::string GuidedVolPathVSPGIntegrator::ToString() const {
    char buf[256];
    std::snprintf(buf, sizeof buf, "[ GuidedVolPathVSPGIntegrator maxDepth: %d lightSampler: %s regularize: %s ]", params.maxdepth,
                  params.lightsampler == VSPG_LIGHTSAMPLER_BVH ? "bvh" : (params.lightsampler == VSPG_LIGHTSAMPLER_POWER ? "power" : "uniform"),
                  params.regularize ? "true" : "false");
    return buf;
}
Film GuidedVolPathVSPGIntegrator::GetFilm() {
    Film f;
    f.xres = cfg.xres; f.yres = cfg.yres;
    f.rgbw.resize((size_t)cfg.xres * cfg.yres * 4);
    if (vspg_film_read(renderer, f.rgbw.data(), nullptr) != 0) throw Error(vspg_last_error());
    return f;
}
VspgCounters GuidedVolPathVSPGIntegrator::Counters() {
    VspgCounters c;
    if (vspg_get_counters(renderer, &c, nullptr) != 0) throw Error(vspg_last_error());
    return c;
}

void Film::GetPixelRGB(int x, int y, float rgb[3]) const {
    const float *p = &rgbw[((size_t)y * xres + x) * 4];
    float w = p[3];
    for (int c = 0; c < 3; ++c) rgb[c] = w != 0 ? p[c] / w : p[c];
}
void Film::WritePFM(const std::string &filename) const {
    std::ofstream f(filename, std::ios::binary);
    f << "PF\n" << xres << " " << yres << "\n-1.0\n";
    for (int y = yres - 1; y >= 0; --y)
        for (int x = 0; x < xres; ++x) {
            float rgb[3];
            GetPixelRGB(x, y, rgb);
            f.write(reinterpret_cast<const char *>(rgb), sizeof rgb);
        }
}

}  // namespace vspg

// vspg_scenefile.cpp -- see vspg_scenefile.h
#include "vspg_scenefile.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>

namespace vspg {
namespace {

// ---- 4x4 float matrices, composed like pbrt's Transform (util/transform.h; SquareMatrix Mul in util/math.h: FMA accumulation) ----
struct M4 {
    float m[4][4];
};
M4 identity() {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) r.m[i][j] = i == j ? 1.f : 0.f;
    return r;
}
M4 mul(const M4 &a, const M4 &b) {
    M4 r;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s = std::fmaf(a.m[i][k], b.m[k][j], s);
            r.m[i][j] = s;
        }
    return r;
}
M4 translate(float x, float y, float z) {
    M4 r = identity();
    r.m[0][3] = x; r.m[1][3] = y; r.m[2][3] = z;
    return r;
}
M4 scale(float x, float y, float z) {
    M4 r = identity();
    r.m[0][0] = x; r.m[1][1] = y; r.m[2][2] = z;
    return r;
}
M4 rotate(float thetaDeg, float ax, float ay, float az) {  // transform.h:220-247
    const float l = std::sqrt(ax * ax + ay * ay + az * az);
    if (!(l > 0)) throw Error("Rotate: zero-length axis");
    const float x = ax / l, y = ay / l, z = az / l;
    const float rad = (3.14159265358979323846f / 180) * thetaDeg;
    const float s = std::sin(rad), c = std::cos(rad);
    M4 r = identity();
    r.m[0][0] = x * x + (1 - x * x) * c; r.m[0][1] = x * y * (1 - c) - z * s; r.m[0][2] = x * z * (1 - c) + y * s;
    r.m[1][0] = x * y * (1 - c) + z * s; r.m[1][1] = y * y + (1 - y * y) * c; r.m[1][2] = y * z * (1 - c) - x * s;
    r.m[2][0] = x * z * (1 - c) - y * s; r.m[2][1] = y * z * (1 - c) + x * s; r.m[2][2] = z * z + (1 - z * z) * c;
    return r;
}
bool is_identity(const M4 &a) {
    const M4 i = identity();
    return std::memcmp(&a, &i, sizeof a) == 0;
}
void xf_point(const M4 &a, const float p[3], float out[3]) {  // Transform::operator()(Point3) (transform.h:310-319), affine
    for (int i = 0; i < 3; ++i) out[i] = a.m[i][0] * p[0] + a.m[i][1] * p[1] + a.m[i][2] * p[2] + a.m[i][3];
}
void xf_vector(const M4 &a, const float v[3], float out[3]) {
    for (int i = 0; i < 3; ++i) out[i] = a.m[i][0] * v[0] + a.m[i][1] * v[1] + a.m[i][2] * v[2];
}

// ---- tokens ----------------------------------------------------------------------------------------------------
struct Token {
    enum Kind { Word, String, Number, Open, Close } kind;
    std::string text;
    int line;
};
std::vector<Token> tokenize(const std::string &s) {
    std::vector<Token> out;
    int line = 1;
    for (size_t i = 0; i < s.size();) {
        const char c = s[i];
        if (c == '\n') { ++line; ++i; }
        else if (std::isspace((unsigned char)c)) ++i;
        else if (c == '#') { while (i < s.size() && s[i] != '\n') ++i; }
        else if (c == '[') { out.push_back({Token::Open, "[", line}); ++i; }
        else if (c == ']') { out.push_back({Token::Close, "]", line}); ++i; }
        else if (c == '"') {
            const size_t j = s.find('"', i + 1);
            if (j == std::string::npos) throw Error("line " + std::to_string(line) + ": unterminated string");
            out.push_back({Token::String, s.substr(i + 1, j - i - 1), line});
            i = j + 1;
        } else {
            size_t j = i;
            while (j < s.size() && !std::isspace((unsigned char)s[j]) && s[j] != '[' && s[j] != ']' && s[j] != '"' && s[j] != '#') ++j;
            const std::string t = s.substr(i, j - i);
            const bool num = std::isdigit((unsigned char)t[0]) || t[0] == '-' || t[0] == '+' || t[0] == '.';
            out.push_back({num ? Token::Number : Token::Word, t, line});
            i = j;
        }
    }
    return out;
}

struct GraphicsState {
    M4 ctm = identity();
    float Kd[3] = {0.5f, 0.5f, 0.5f};  // DiffuseMaterial default reflectance 0.5
    bool areaLight = false;
    float Le[3] = {0, 0, 0};
    bool twoSided = false;
    bool reverseOrientation = false;
    bool interfaceMaterial = false;   // Material "interface" (also "none", ""): no BSDF, the surface only separates media (scene.cpp:1340)
    std::string insideMedium, outsideMedium;
};
struct MediumNames { std::string inside, outside; };   // a shape's MediumInterface, resolved once the scene's medium is known
struct NamedMedium {
    std::string type;
    ParameterDictionary params;
    M4 ctm;
};

class Parser {
  public:
    explicit Parser(const std::string &text, const std::string &baseDir = "") : tok(tokenize(text)), baseDir(baseDir) {}
    std::unique_ptr<SceneDescription> run() {
        sd = std::make_unique<SceneDescription>();
        std::memset(&sd->scene, 0, sizeof sd->scene);
        while (pos < tok.size()) directive();
        finish();
        return std::move(sd);
    }

  private:
    std::vector<Token> tok;
    std::string baseDir;   // directory of the scene file: Include and relative file names resolve against it
    int includeDepth = 0;
    size_t pos = 0;
    std::unique_ptr<SceneDescription> sd;
    GraphicsState gs;
    std::vector<GraphicsState> stack;
    std::map<std::string, NamedMedium> media;
    std::map<std::string, ParameterDictionary> materials;
    bool haveCamera = false, haveLookAt = false, world = false;
    float eye[3] = {0, 0, 0}, look[3] = {0, 0, -1}, up[3] = {0, 1, 0}, fov = 90.f;
    std::string cameraMedium;
    std::vector<VspgQuad> quads;
    std::vector<VspgSphere> spheres;
    std::vector<MediumNames> quadMedia, sphereMedia, triMedia;   // per rectangle / sphere / triangle

    [[noreturn]] void fail(const std::string &msg) const {
        const int line = pos < tok.size() ? tok[pos].line : (tok.empty() ? 0 : tok.back().line);
        throw Error("scene file line " + std::to_string(line) + ": " + msg);
    }
    float number() {
        if (pos >= tok.size() || tok[pos].kind != Token::Number) fail("expected a number");
        return std::strtof(tok[pos++].text.c_str(), nullptr);
    }
    void numbers(float *out, int n) {
        const bool br = pos < tok.size() && tok[pos].kind == Token::Open;
        if (br) ++pos;
        for (int i = 0; i < n; ++i) out[i] = number();
        if (br) { if (pos >= tok.size() || tok[pos].kind != Token::Close) fail("expected ]"); ++pos; }
    }
    std::string str() {
        if (pos >= tok.size() || tok[pos].kind != Token::String) fail("expected a quoted string");
        return tok[pos++].text;
    }
    // the parameter list that follows: everything up to the next directive word (bare true / false belong to the list),
    // re-assembled for ParameterDictionary::Parse
    ParameterDictionary params_with_bare_bools() {
        std::string text;
        while (pos < tok.size()) {
            if (tok[pos].kind == Token::Word && tok[pos].text == "Include" && pos + 1 < tok.size() && tok[pos + 1].kind == Token::String) {
                // beyond pbrt (whose Include is a directive only): an Include INSIDE a parameter list splices the file's parameters in,
                // so that the block `nanovdb2pbrt` prints can stay in its own file: MakeNamedMedium "c" "string type" "uniformgrid" Include "grid.pbrt"
                ++pos;
                splice_include();
                continue;
            }
            if (!(tok[pos].kind != Token::Word || tok[pos].text == "true" || tok[pos].text == "false")) break;
            const Token &t = tok[pos++];
            if (t.kind == Token::String) text += "\"" + t.text + "\" ";
            else text += t.text + " ";
        }
        return ParameterDictionary::Parse(text);
    }
    // pbrt's Include (parser.cpp): the named file's tokens take the directive's place.  Relative names resolve against the
    // including scene file's directory.
    void splice_include() {
        std::string name = str();
        if (!name.empty() && name[0] != '/' && !baseDir.empty()) name = baseDir + "/" + name;
        if (++includeDepth > 16) fail("Include nested deeper than 16 files");
        std::ifstream f(name);
        if (!f) fail("Include \"" + name + "\": cannot open");
        std::stringstream ss;
        ss << f.rdbuf();
        const std::vector<Token> inc = tokenize(ss.str());
        tok.insert(tok.begin() + (std::ptrdiff_t)pos, inc.begin(), inc.end());
    }
    void concat(const M4 &t) { gs.ctm = mul(gs.ctm, t); }

    void directive() {
        if (tok[pos].kind != Token::Word) fail("expected a directive, found \"" + tok[pos].text + "\"");
        const std::string d = tok[pos++].text;
        if (d == "LookAt") {
            float v[9];
            numbers(v, 9);
            if (world || !is_identity(gs.ctm)) fail("LookAt is supported as the camera's only transform");
            std::memcpy(eye, v, 12); std::memcpy(look, v + 3, 12); std::memcpy(up, v + 6, 12);
            haveLookAt = true;
        } else if (d == "Camera") {
            const std::string type = str();
            ParameterDictionary p = params_with_bare_bools();
            if (type != "perspective") fail("Camera \"" + type + "\": only \"perspective\" is inside this build's scope");
            fov = p.GetOneFloat("fov", 90.f);
            if (p.GetOneFloat("lensradius", 0.f) != 0.f) fail("thin-lens cameras are outside this build's scope");
            p.ReportUnused();
            haveCamera = true;
            cameraMedium = gs.outsideMedium;  // the camera takes the current OUTSIDE medium (scene.cpp:153-155, 664-665)
        } else if (d == "Sampler") {
            const std::string sname = str();  // every sampler is run as "independent": the path's parity is defined on it (samplers.h:442-476)
            if (sname != "independent") sd->warnings.push_back("Sampler \"" + sname + "\" runs as \"independent\"");
            ParameterDictionary p = params_with_bare_bools();
            sd->pixelSamples = p.GetOneInt("pixelsamples", 16);
            sd->seed = p.GetOneInt("seed", 0);
            p.ReportUnused();
        } else if (d == "PixelFilter") {
            const std::string type = str();
            ParameterDictionary p = params_with_bare_bools();
            if (type != "box") fail("PixelFilter \"" + type + "\": the film accumulates with pbrt's box filter of radius 0.5 only");
            if (p.GetOneFloat("radius", 0.5f) != 0.5f || p.GetOneFloat("xradius", 0.5f) != 0.5f || p.GetOneFloat("yradius", 0.5f) != 0.5f)
                fail("box filter radius must be 0.5");
            p.ReportUnused();
        } else if (d == "Film") {
            const std::string type = str();
            ParameterDictionary p = params_with_bare_bools();
            if (type != "rgb") fail("Film \"" + type + "\": only \"rgb\" is inside this build's scope");
            sd->xres = p.GetOneInt("xresolution", 1280);
            sd->yres = p.GetOneInt("yresolution", 720);
            sd->filmFilename = p.GetOneString("filename", "pbrt.pfm");
            p.ReportUnused();
        } else if (d == "Integrator") {
            sd->integratorName = str();
            sd->integratorParams = params_with_bare_bools();
        } else if (d == "Option" || d == "ColorSpace") {
            std::string what = d;
            while (pos < tok.size() && tok[pos].kind != Token::Word) what += " " + tok[pos++].text;
            sd->warnings.push_back("ignored: " + what);
        } else if (d == "WorldBegin") {
            if (!haveCamera) fail("WorldBegin before Camera");
            world = true;
            gs.ctm = identity();
        } else if (d == "AttributeBegin" || d == "TransformBegin") {
            stack.push_back(gs);
        } else if (d == "AttributeEnd" || d == "TransformEnd") {
            if (stack.empty()) fail(d + " without a matching Begin");
            gs = stack.back();
            stack.pop_back();
        } else if (d == "Identity") {
            gs.ctm = identity();
        } else if (d == "Translate") {
            float v[3]; numbers(v, 3); concat(translate(v[0], v[1], v[2]));
        } else if (d == "Scale") {
            float v[3]; numbers(v, 3); concat(scale(v[0], v[1], v[2]));
        } else if (d == "Rotate") {
            float v[4]; numbers(v, 4); concat(rotate(v[0], v[1], v[2], v[3]));
        } else if (d == "Transform" || d == "ConcatTransform") {
            float v[16]; numbers(v, 16);
            M4 t;
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) t.m[i][j] = v[4 * j + i];  // the file lists the matrix column by column (parser.cpp: Transpose)
            if (t.m[3][0] != 0 || t.m[3][1] != 0 || t.m[3][2] != 0 || t.m[3][3] != 1) fail("projective transforms are outside this build's scope");
            if (d == "Transform") gs.ctm = t; else concat(t);
        } else if (d == "ReverseOrientation") {
            gs.reverseOrientation = !gs.reverseOrientation;
        } else if (d == "Material") {
            const std::string type = str();
            ParameterDictionary p = params_with_bare_bools();
            set_material(type, p);
        } else if (d == "MakeNamedMaterial") {
            const std::string name = str();
            materials[name] = params_with_bare_bools();
        } else if (d == "NamedMaterial") {
            const std::string name = str();
            auto it = materials.find(name);
            if (it == materials.end()) fail("NamedMaterial \"" + name + "\" is not defined");
            ParameterDictionary p = it->second;
            set_material(p.GetOneString("type", "diffuse"), p);
        } else if (d == "AreaLightSource") {
            const std::string type = str();
            ParameterDictionary p = params_with_bare_bools();
            if (type != "diffuse") fail("AreaLightSource \"" + type + "\": only \"diffuse\"");
            float L[3] = {1, 1, 1};
            p.GetOneRGB("L", L);
            const float sc = p.GetOneFloat("scale", 1.f);
            gs.twoSided = p.GetOneBool("twosided", false);
            p.ReportUnused();
            for (int k = 0; k < 3; ++k) gs.Le[k] = L[k] * sc;
            gs.areaLight = true;
        } else if (d == "LightSource") {
            const std::string type = str();
            ParameterDictionary p = params_with_bare_bools();
            light_source(type, p);
        } else if (d == "MakeNamedMedium") {
            const std::string name = str();
            NamedMedium m;
            m.params = params_with_bare_bools();
            m.type = m.params.GetOneString("type", "");
            m.ctm = gs.ctm;
            media[name] = m;
        } else if (d == "MediumInterface") {
            gs.insideMedium = str();
            gs.outsideMedium = pos < tok.size() && tok[pos].kind == Token::String ? str() : gs.insideMedium;
        } else if (d == "Shape") {
            const std::string type = str();
            ParameterDictionary p = params_with_bare_bools();
            shape(type, p);
        } else if (d == "Include") {
            splice_include();
        } else {
            fail("directive \"" + d + "\" is outside this build's scope");
        }
    }
    void set_material(const std::string &type, ParameterDictionary &p) {
        if (type == "interface" || type == "none" || type.empty()) {  // a null Material (scene.cpp:1340, materials.cpp:736): medium boundaries
            (void)p.GetOneString("type", "");
            p.ReportUnused();
            gs.interfaceMaterial = true;
            return;
        }
        if (type != "diffuse") fail("Material \"" + type + "\": only \"diffuse\" and \"interface\" are inside this build's scope");
        gs.interfaceMaterial = false;
        float kd[3] = {0.5f, 0.5f, 0.5f};
        if (!p.GetOneRGB("reflectance", kd) && p.Has("reflectance")) { const float f = p.GetOneFloat("reflectance", 0.5f); kd[0] = kd[1] = kd[2] = f; }
        (void)p.GetOneString("type", "");
        p.ReportUnused();
        std::memcpy(gs.Kd, kd, sizeof kd);
    }
    void light_source(const std::string &type, ParameterDictionary &p) {
        if (sd->scene.n_infinite_lights >= VSPG_MAX_INFINITE_LIGHTS) fail("too many infinite lights");
        VspgInfiniteLight &il = sd->scene.infinite_lights[sd->scene.n_infinite_lights];
        float L[3] = {1, 1, 1};
        p.GetOneRGB("L", L);
        const float sc = p.GetOneFloat("scale", 1.f);
        if (type == "infinite") {
            if (!p.GetOneString("filename", "").empty()) fail("image infinite lights are outside this build's scope");
            il.type = VSPG_LIGHT_UNIFORM_INFINITE;
            il.w_light[0] = 0; il.w_light[1] = 1; il.w_light[2] = 0;
        } else if (type == "distant") {
            float from[3] = {0, 0, 0}, to[3] = {0, 0, 1};
            p.GetOnePoint3("from", from);
            p.GetOnePoint3("to", to);
            // DistantLight::Create (lights.cpp): the light shines along (to - from); SampleLi's wi points back towards `from`
            const float w[3] = {from[0] - to[0], from[1] - to[1], from[2] - to[2]};
            float wr[3];
            xf_vector(gs.ctm, w, wr);
            const float l = std::sqrt(wr[0] * wr[0] + wr[1] * wr[1] + wr[2] * wr[2]);
            if (!(l > 0)) fail("distant light: from == to");
            for (int k = 0; k < 3; ++k) il.w_light[k] = wr[k] / l;
            il.type = VSPG_LIGHT_DISTANT;
        } else {
            fail("LightSource \"" + type + "\": only \"infinite\" (uniform) and \"distant\" are inside this build's scope");
        }
        p.ReportUnused();
        for (int k = 0; k < 3; ++k) il.L[k] = L[k] * sc;
        sd->scene.n_infinite_lights++;
    }
    void add_triangle(const float *a, const float *b, const float *c) {
        if (gs.areaLight) fail("emissive triangles are outside this build's scope (emission lives on rectangles: use a planar \"bilinearmesh\" patch)");
        const float *v[3] = {a, b, c};
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k < 3; ++k) sd->triP.push_back(v[i][k]);
        for (int k = 0; k < 3; ++k) sd->triKd.push_back(gs.Kd[k]);
        // the geometric normal Normalize(Cross(p0 - p2, p1 - p2)) flips with reverseOrientation ^ transformSwapsHandedness
        // (shapes.h:934-936) -- the vertices above are already in render space, as the reference's TriangleMesh keeps them
        sd->triFlags.push_back((gs.interfaceMaterial ? VSPG_TRI_INTERFACE : 0) | (gs.reverseOrientation != swaps_handedness(gs.ctm) ? VSPG_TRI_FLIP_NORMAL : 0));
        triMedia.push_back(MediumNames{gs.insideMedium, gs.outsideMedium});
    }
    static bool swaps_handedness(const M4 &m) {  // Transform::SwapsHandedness (util/transform.cpp): det of the upper 3x3 < 0
        const float det = m.m[0][0] * (m.m[1][1] * m.m[2][2] - m.m[1][2] * m.m[2][1]) - m.m[0][1] * (m.m[1][0] * m.m[2][2] - m.m[1][2] * m.m[2][0]) +
                          m.m[0][2] * (m.m[1][0] * m.m[2][1] - m.m[1][1] * m.m[2][0]);
        return det < 0;
    }
    void shape(const std::string &type, ParameterDictionary &p) {
        if (!world) fail("Shape before WorldBegin");
        // MediumInterface (round 4): the shape keeps its inside / outside medium NAMES; finish() turns them into "the scene's medium"
        // or "no medium" once it is known which medium the scene holds (this build holds one: a second one is refused there).
        if (gs.interfaceMaterial && gs.areaLight) fail("an area light on a shape with the \"interface\" material is outside this build's scope");
        if (type == "sphere") {
            const float radius = p.GetOneFloat("radius", 1.f);
            const float zmin = p.GetOneFloat("zmin", -radius), zmax = p.GetOneFloat("zmax", radius), phimax = p.GetOneFloat("phimax", 360.f);
            p.ReportUnused();
            if (!(radius > 0)) fail("sphere: radius must be positive");
            if (zmin > -radius || zmax < radius || phimax < 360.f) fail("partial spheres (zmin / zmax / phimax) are outside this build's scope");
            if (gs.areaLight) fail("emissive spheres are outside this build's scope (emission lives on rectangles)");
            if (spheres.size() >= VSPG_MAX_SPHERES) fail("more than " + std::to_string(VSPG_MAX_SPHERES) + " spheres");
            VspgSphere sp;
            std::memset(&sp, 0, sizeof sp);
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) sp.render_from_object[4 * i + j] = gs.ctm.m[i][j];
            if (vspg_transform_inverse(sp.render_from_object, sp.object_from_render) != 0) fail(std::string("sphere: ") + vspg_last_error());
            sp.radius = radius;
            for (int k = 0; k < 3; ++k) sp.Kd[k] = gs.Kd[k];
            sp.reverse_orientation = gs.reverseOrientation;   // (the handedness of the CTM is the library's to find: Sphere's constructor, shapes.h:122)
            sp.material = gs.interfaceMaterial ? VSPG_MATERIAL_INTERFACE : VSPG_MATERIAL_DIFFUSE;
            spheres.push_back(sp);
            sphereMedia.push_back(MediumNames{gs.insideMedium, gs.outsideMedium});
            return;
        }
        std::vector<float> P = p.GetPoint3Array("P");
        if (p.Has("N") || p.Has("uv") || p.Has("S")) fail("shading normals / tangents / (u,v) on meshes are outside this build's scope");
        std::vector<float> W(P.size());
        for (size_t i = 0; i + 2 < P.size(); i += 3) xf_point(gs.ctm, &P[i], &W[i]);
        if (type == "bilinearmesh") {
            std::vector<int> idx = p.GetIntArray("indices");
            p.ReportUnused();
            if (idx.empty()) idx = {0, 1, 2, 3};
            if (P.size() % 3) fail("bilinearmesh: \"P\" must hold whole points (a multiple of three floats)");
            if (idx.size() != 4 || W.size() < 12) fail("bilinearmesh: one patch of four points is supported");
            const int nv = (int)(W.size() / 3);
            for (int k = 0; k < 4; ++k)
                if (idx[k] < 0 || idx[k] >= nv) fail("bilinearmesh: vertex index out of range");
            const float *p00 = &W[3 * idx[0]], *p10 = &W[3 * idx[1]], *p01 = &W[3 * idx[2]], *p11 = &W[3 * idx[3]];
            float e1[3], e2[3];
            bool parallelogram = true;
            for (int k = 0; k < 3; ++k) {
                e1[k] = p10[k] - p00[k];
                e2[k] = p01[k] - p00[k];
                parallelogram = parallelogram && std::fabs((p10[k] + e2[k]) - p11[k]) <= 1e-6f * (1 + std::fabs(p11[k]));
            }
            if (parallelogram) {
                if (quads.size() >= VSPG_MAX_QUADS) fail("more than " + std::to_string(VSPG_MAX_QUADS) + " rectangles");
                VspgQuad q;
                std::memset(&q, 0, sizeof q);
                for (int k = 0; k < 3; ++k) { q.p00[k] = p00[k]; q.e1[k] = e1[k]; q.e2[k] = e2[k]; q.Kd[k] = gs.Kd[k]; q.Le[k] = gs.areaLight ? gs.Le[k] : 0.f; }
                q.two_sided = gs.twoSided;
                // BilinearPatch flips n when reverseOrientation ^ transformSwapsHandedness (shapes.h:1163-1164, shapes.cpp:1267-1270):
                // a mirroring CTM turns the patch's parametrisation over
                q.reverse_orientation = gs.reverseOrientation != swaps_handedness(gs.ctm);
                q.material = gs.interfaceMaterial ? VSPG_MATERIAL_INTERFACE : VSPG_MATERIAL_DIFFUSE;
                quads.push_back(q);
                quadMedia.push_back(MediumNames{gs.insideMedium, gs.outsideMedium});
            } else {
                add_triangle(p00, p10, p11);
                add_triangle(p00, p11, p01);
            }
        } else if (type == "trianglemesh") {
            // (a triangle's orientation -- reverseOrientation ^ transformSwapsHandedness, shapes.h:934-936 -- travels as a flag: only a
            // medium transition can tell the two sides of a diffuse triangle apart)
            std::vector<int> idx = p.GetIntArray("indices");
            p.ReportUnused();
            if (P.size() % 3) fail("trianglemesh: \"P\" must hold whole points (a multiple of three floats)");
            const int nv = (int)(W.size() / 3);
            if (idx.empty()) { if (nv != 3) fail("trianglemesh without indices must have exactly three points"); idx = {0, 1, 2}; }
            if (idx.size() % 3) fail("trianglemesh: indices come in threes");
            for (size_t i = 0; i < idx.size(); i += 3) {
                for (int k = 0; k < 3; ++k) if (idx[i + k] < 0 || idx[i + k] >= nv) fail("trianglemesh: vertex index out of range");
                add_triangle(&W[3 * idx[i]], &W[3 * idx[i + 1]], &W[3 * idx[i + 2]]);
            }
        } else {
            fail("Shape \"" + type + "\": only \"bilinearmesh\", \"trianglemesh\" and \"sphere\" are inside this build's scope");
        }
    }
    void finish() {
        if (!haveCamera) throw Error("scene file: no Camera");
        if (!stack.empty()) throw Error("scene file: unbalanced AttributeBegin");
        VspgScene &s = sd->scene;
        s.n_quads = (int)quads.size();
        for (int i = 0; i < s.n_quads; ++i) s.quads[i] = quads[i];
        if (!haveLookAt) { eye[0] = eye[1] = eye[2] = 0; look[0] = 0; look[1] = 0; look[2] = 1; up[0] = 0; up[1] = 1; up[2] = 0; }
        if (vspg_camera_look_at(&s.camera, eye, look, up, fov, sd->xres, sd->yres) != 0) throw Error(std::string("Camera: ") + vspg_last_error());
        s.n_spheres = (int)spheres.size();
        for (int i = 0; i < s.n_spheres; ++i) s.spheres[i] = spheres[i];
        // The scene's medium: the library holds ONE (include/vspg.h).  Every name a camera or a shape refers to must be that one
        // (or "", no medium); a second medium is refused by name.
        std::string theMedium = cameraMedium;
        auto see = [&](const std::string &name) {
            if (name.empty()) return;
            if (theMedium.empty()) theMedium = name;
            else if (name != theMedium)
                throw Error("scene file: media \"" + theMedium + "\" and \"" + name + "\" are both in use -- this build renders ONE medium per scene (with or without boundaries)");
        };
        for (const auto *v : {&quadMedia, &sphereMedia, &triMedia})
            for (const MediumNames &m : *v) { see(m.inside); see(m.outside); }
        auto bits = [&](const MediumNames &m) {
            return (!theMedium.empty() && m.inside == theMedium ? VSPG_IFACE_INSIDE : 0) | (!theMedium.empty() && m.outside == theMedium ? VSPG_IFACE_OUTSIDE : 0);
        };
        for (int i = 0; i < s.n_quads; ++i) s.quads[i].medium_interface = bits(quadMedia[i]);
        for (int i = 0; i < s.n_spheres; ++i) s.spheres[i].medium_interface = bits(sphereMedia[i]);
        for (size_t i = 0; i < triMedia.size(); ++i) sd->triFlags[i] |= bits(triMedia[i]) << VSPG_TRI_IFACE_SHIFT;
        s.camera_outside_medium = !theMedium.empty() && cameraMedium != theMedium;   // CameraBase::medium = the outside medium at the Camera directive
        s.medium.type = VSPG_MEDIUM_NONE;
        if (!theMedium.empty()) {
            const std::string cameraMedium = theMedium;   // (the block below creates the scene's medium, whoever named it)
            auto it = media.find(cameraMedium);
            if (it == media.end()) throw Error("medium \"" + cameraMedium + "\" is not defined");
            ParameterDictionary p = it->second.params;
            (void)p.GetOneString("type", "");
            if (it->second.type == "nanovdb") {  // ResolveFilename (media.cpp:686): relative to the scene file's directory
                const std::string fn = p.GetOneString("filename", "");
                if (!fn.empty() && fn[0] != '/' && !baseDir.empty()) p.String("filename", baseDir + "/" + fn);
            }
            s.medium = CreateMedium(it->second.type, p, &sd->density, &sd->leScale, &sd->temperature);
            if (!is_identity(it->second.ctm)) {
                if (s.medium.type == VSPG_MEDIUM_HOMOGENEOUS) { /* a homogeneous medium has no frame */ }
                else {
                    s.medium.has_transform = 1;
                    for (int i = 0; i < 4; ++i)
                        for (int j = 0; j < 4; ++j) s.medium.render_from_medium[4 * i + j] = it->second.ctm.m[i][j];
                    if (vspg_transform_inverse(s.medium.render_from_medium, s.medium.medium_from_render) != 0)
                        throw Error(std::string("MakeNamedMedium \"") + cameraMedium + "\": " + vspg_last_error());
                }
            }
        }
        if (!sd->triP.empty()) {
            s.n_triangles = (int)(sd->triP.size() / 9);
            s.tri_p = sd->triP.data();
            s.tri_kd = sd->triKd.data();
            s.tri_flags = sd->triFlags.data();
        }
    }
};

}  // namespace

std::unique_ptr<SceneDescription> ParseSceneString(const std::string &text) { return Parser(text).run(); }
std::unique_ptr<SceneDescription> ParseSceneFile(const std::string &filename) {
    std::ifstream f(filename);
    if (!f) throw Error(filename + ": cannot open");
    std::stringstream ss;
    ss << f.rdbuf();
    const size_t slash = filename.find_last_of('/');
    return Parser(ss.str(), slash == std::string::npos ? std::string(".") : filename.substr(0, slash)).run();
}
std::unique_ptr<Integrator> CreateIntegrator(const SceneDescription &sd, int device) {
    return Integrator::Create(sd.integratorName, sd.integratorParams, sd.scene, sd.xres, sd.yres, sd.pixelSamples, sd.seed, device);
}

}  // namespace vspg

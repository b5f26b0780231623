// vspg_scenefile.h -- a reader for the subset of the pbrt-v4 scene-file format the GuidedVolPathVSPG path needs
// (SURVEY 8f row 4): the directives of App. F plus what configs 3-5 add -- placed grid media, triangle meshes, sky and sun.
//
// What it stands in for: the parser -> BasicSceneBuilder -> BasicScene::CreateIntegrator chain
// (src/pbrt/parser.cpp, scene.cpp:883-887, cpu/render.cpp:56-57) up to Integrator::Create(name, parameters, ...)
// (cpu/integrators.cpp:3739-3744), Medium::Create (media.cpp:816-841) and Light::Create for "infinite" / "distant"
// (lights.cpp).  Everything is routed through the adapter's ParameterDictionary, so parameter names, defaults and the
// unused-parameter error are the reference's.
//
// Directives: LookAt, Camera "perspective" (fov), Sampler (pixelsamples, seed), PixelFilter "box", Film "rgb"
// (xresolution, yresolution, filename), Integrator, Option (ignored), ColorSpace (ignored), WorldBegin, AttributeBegin/End,
// Identity, Translate, Scale, Rotate, Transform, ConcatTransform, ReverseOrientation, Material "diffuse" (reflectance) / "interface",
// MakeNamedMaterial / NamedMaterial ("diffuse", "interface"), AreaLightSource "diffuse" (L, scale, twosided), LightSource "infinite"
// (L, scale; no image) / "distant" (L, scale, from, to), MakeNamedMedium ("homogeneous", "uniformgrid"), MediumInterface,
// Include (as a directive, and -- beyond pbrt -- inside a parameter list, for the block the reference's nanovdb2pbrt prints),
// Shape "bilinearmesh" (one patch: a parallelogram becomes a rectangle, anything else two triangles) / "trianglemesh"
// (P, indices) / "sphere" (radius; full spheres).  Anything else is an Error naming the directive: nothing is silently dropped.
//
// Conventions of this build: render space == world space (pbrt's --render-coord-sys world); ONE medium per scene,
// with the reference's boundaries (round 4): every shape carries its MediumInterface and its material -- an "interface" material
// makes it a pure medium boundary (guidedvolpathvspgintegrator.cpp:399-404) -- and the camera starts in the "outside" medium of the
// MediumInterface in effect at the Camera directive (scene.cpp:153-155); a second medium in the same scene is refused by name;
// emission on rectangles only; the film is written as PFM (OpenEXR is an absent submodule).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "vspg_host.h"

namespace vspg {

struct SceneDescription {
    VspgScene scene;                     // pointers inside refer to the vectors below: keep the description alive while creating
    std::vector<float> density, leScale, temperature, triP, triKd;
    std::vector<int32_t> triFlags;       // VSPG_TRI_* per triangle (material, MediumInterface, orientation)
    std::string integratorName = "volpath";
    ParameterDictionary integratorParams;
    int xres = 1280, yres = 720;         // Film defaults (film.cpp)
    int pixelSamples = 16;               // Sampler default
    int seed = 0;
    std::string filmFilename = "pbrt.pfm";
    std::vector<std::string> warnings;   // directives that were accepted and ignored (Option, ColorSpace)
};

// Parse scene text (the contents of a .pbrt file).  Throws vspg::Error on anything outside the subset above.
std::unique_ptr<SceneDescription> ParseSceneString(const std::string &text);
std::unique_ptr<SceneDescription> ParseSceneFile(const std::string &filename);
// Integrator::Create for the parsed scene (BasicScene::CreateIntegrator, scene.cpp:883-887)
std::unique_ptr<Integrator> CreateIntegrator(const SceneDescription &sd, int device = 0);

}  // namespace vspg

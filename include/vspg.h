/*
 * vspg.h -- C-ABI of the MI355X-native GuidedVolPathVSPG hot path.
 *
 * The reference (kehanxuuu/vspg-pbrt-v4) has no FFI: the path sits behind C++ plugin
 * classes chosen by string in static Create() functions.  This header is
 * the boundary a maintainer would bind instead; every entry point names the reference
 * interface it replaces (paths relative to the reference root, file:line).
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no C++/torch types.
 *   - every function returns 0 on success, a negative VSPG_E* code on failure;
 *     vspg_last_error() returns a human-readable message for the calling thread.
 *     (The reference aborts the process via ErrorExit/LOG_FATAL,
 *     src/pbrt/cpu/integrators.cpp:3760-3766; a library must not, so fatal conditions
 *     become error codes with the same trigger conditions.)
 *   - the library owns all device memory behind the opaque handle; host buffers are owned
 *     by the caller.  `stream` arguments are hipStream_t passed as void* (NULL = default).
 *   - one renderer per GPU; one host thread drives it (this replaces the
 *     ParallelFor2D thread-pool fan-out of src/pbrt/cpu/integrators.cpp:183-207).
 *   - there is NO CPU fallback: without a usable HIP device every compute entry point
 *     fails with VSPG_ENODEVICE.
 */
#ifndef VSPG_H
#define VSPG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSPG_ABI_VERSION 7

/* ---- error codes ------------------------------------------------------------------- */
#define VSPG_OK 0
#define VSPG_EINVAL (-1)     /* bad argument / unsupported parameter combination          */
#define VSPG_ENODEVICE (-2)  /* no HIP device / kernels not loadable                      */
#define VSPG_EHIP (-3)       /* a HIP runtime call failed (message has the hipError name) */
#define VSPG_ESCOPE (-4)     /* valid in the reference, outside this library's hot path   */

/* ---- scene description (synthetic analytic scene) ---------------- */
#define VSPG_MAX_QUADS 16

/* A rectangle p00 + u*e1 + v*e2, u,v in [0,1], e1 perpendicular to e2.  Stands in for a
 * pbrt "bilinearmesh" rectangle with a "diffuse" material and, when Le != 0, a "diffuse"
 * area light (src/pbrt/shapes.cpp:1128-1143, src/pbrt/bxdfs.h:31-80,
 * src/pbrt/lights.cpp:796-820). */
/* Medium boundaries (round 4).  A surface carries what pbrt's GeometricPrimitive hands the hit (src/pbrt/cpu/primitive.cpp:77): a material -- "diffuse", or NONE for `Material "interface"` (materials.cpp:736, scene.cpp:1340: a null
 * Material; GetBSDF returns no BSDF and Li skips the hit, guidedvolpathvspgintegrator.cpp:399-404) -- and a MediumInterface
 * {inside, outside} (base/medium.h:113-128; `MediumInterface "name_in" "name_out"`).  The library holds ONE medium
 * (VspgScene.medium), so each side is "the medium" or "no medium": medium_interface is a pair of bits.  Only a TRANSITION
 * (inside != outside, MediumInterface::IsMediumTransition) changes anything: Interaction::GetMedium(w) then returns the
 * outside medium for Dot(w, n) > 0, the inside one otherwise (interaction.h:117-121), n being the surface's geometric
 * normal after reverse_orientation; a surface that is not a transition leaves a ray in the medium it arrived in
 * (SurfaceInteraction::SetIntersectionProperties, interaction.h:218-229). */
enum { VSPG_MATERIAL_DIFFUSE = 0, VSPG_MATERIAL_INTERFACE = 1 };
#define VSPG_IFACE_INSIDE 1   /* the scene's medium lies on the inside (behind the normal):  MediumInterface "m" ""  */
#define VSPG_IFACE_OUTSIDE 2  /* ... on the outside (the side the normal points to):         MediumInterface ""  "m" */
/* (0 and VSPG_IFACE_INSIDE | VSPG_IFACE_OUTSIDE: not a transition) */

typedef struct VspgQuad {
    float p00[3];
    float e1[3];
    float e2[3];
    float Kd[3];                  /* diffuse reflectance R (RGB) */
    float Le[3];                  /* emitted radiance (RGB); all zero = not a light */
    int32_t two_sided;            /* DiffuseAreaLight "twosided" */
    int32_t reverse_orientation;  /* flips the geometric normal normalize(e1 x e2) */
    int32_t material;             /* VSPG_MATERIAL_* */
    int32_t medium_interface;     /* VSPG_IFACE_* bits */
} VspgQuad;

/* Shape "sphere" (src/pbrt/shapes.h:107-330): the full sphere of `radius` around the object-space origin (partial spheres
 * -- zmin / zmax / phimax -- are refused), placed by renderFromObject.  Both matrices of pbrt's Transform, row-major, as
 * for VspgMedium (vspg_transform_inverse fills the inverse).  Intersection = Sphere::BasicIntersect with its interval
 * arithmetic (shapes.h:147-229), interaction = InteractionFromIntersection + Transform::operator()(SurfaceInteraction)
 * (shapes.h:237-284, transform.cpp:229-261).  Diffuse or interface; not a light. */
#define VSPG_MAX_SPHERES 8
typedef struct VspgSphere {
    float render_from_object[16];
    float object_from_render[16];
    float radius;
    float Kd[3];
    int32_t reverse_orientation;
    int32_t material;
    int32_t medium_interface;
} VspgSphere;

/* Pinhole camera: raster point (x,y) -> camera-space point (sx*x+ox, sy*y+oy, 1),
 * normalised, then rotated into render space by the orthonormal frame (right, up, fwd).
 * Replaces PerspectiveCamera::GenerateRayDifferential (src/pbrt/cameras.cpp:435-447)
 * for lensRadius == 0.  Fill with vspg_camera_look_at(). */
typedef struct VspgCamera {
    float origin[3];
    float right[3], up[3], fwd[3];
    float sx, ox, sy, oy;
} VspgCamera;

enum { VSPG_MEDIUM_NONE = 0, VSPG_MEDIUM_HOMOGENEOUS = 1, VSPG_MEDIUM_GRID = 2, VSPG_MEDIUM_NANOVDB = 3 };

/* The scene's one medium.  Without medium boundaries (no transition surface, camera_outside_medium == 0) it fills the
 * scene -- ray.medium for every ray, rounds 1-3 -- otherwise a ray is in it or in no medium at all, as the camera and
 * the MediumInterfaces it crossed say.
 * HOMOGENEOUS mirrors HomogeneousMedium (src/pbrt/media.h:221-283; parameters
 * src/pbrt/media.cpp:167-206): sigma_a/sigma_s are RGB, already multiplied by "scale";
 * Le already multiplied by "Lescale".
 * GRID mirrors GridMedium ("uniformgrid", src/pbrt/media.h:284-390,
 * src/pbrt/media.cpp:209-361): density grid nx*ny*nz, x fastest, over `bounds` in medium
 * space == render space (identity renderFromMedium); sigma_a/sigma_s are the RGB spectra
 * already multiplied by "scale"; majorant grid is 16^3 (media.cpp:252). */
typedef struct VspgMedium {
    int32_t type;
    float sigma_a[3];
    float sigma_s[3];
    float g;
    float Le[3];
    /* grid only */
    int32_t nx, ny, nz;
    float bounds_min[3], bounds_max[3];
    const float *density; /* HOST pointer, nx*ny*nz floats; copied at create time */
    /* VSPG_MEDIUM_NANOVDB only -- NanoVDBMedium (src/pbrt/media.h:657-753, media.cpp:549-675) over a DENSE copy
     * of the density grid (what cmd/nanovdb2pbrt.cpp:97-126 dumps): voxel (i,j,k) of the array is index
     * (index_min + (i,j,k)); values outside the index bounding box are the background 0.
     * worldToIndexF(p) = (p - grid_origin) / voxel_size; `bounds` is the grid's world bounding box;
     * majorants live on a 64^3 grid (media.cpp:574); "densityoffset" / "majorantscale" as in the RGB-mode ctor. */
    int32_t index_min[3];
    float voxel_size[3];
    float grid_origin[3];
    float density_offset;
    float majorant_scale;
    /* VSPG_MEDIUM_GRID only -- emission of GridMedium (src/pbrt/media.h:326-342, media.cpp:316-328): Le at a point =
     * LeScale.Lookup(p) * Le when that scale is positive.  `Le` above is Le_spec sampled (RGB); le_scale is the
     * "Lescale" grid ALREADY multiplied by the reference's photometric normalisation 1 / SpectrumToPhotometric(Le)
     * (HOST pointer, le_nx*le_ny*le_nz floats, x fastest, copied at create time).  NULL with a non-zero Le = the
     * reference's default 1x1x1 grid holding 1 (trilinear against the zero background: a tent over the bounds).
     * Temperature grids (blackbody emission) are outside this build's scope. */
    const float *le_scale;
    int32_t le_nx, le_ny, le_nz;
    /* grid media -- renderFromMedium (src/pbrt/media.h:322, :354, :693, :708; the Transform the scene file's CTM gave
     * the medium): has_transform = 0 means identity.  Both matrices of pbrt's Transform are passed, row-major
     * (m and mInv, util/transform.h:186-187): a pbrt host hands over its own inverse, so that
     * Transform::ApplyInverse(Ray, &tMax) / ApplyInverse(Point3f) (transform.h:387-429, transform.cpp:263-303) see the same
     * floats.  vspg_transform_inverse() fills medium_from_render for callers that only have m.  Affine matrices only
     * (last row 0 0 0 1). */
    int32_t has_transform;
    float render_from_medium[16];
    float medium_from_render[16];
    /* Temperature grid of an emissive GridMedium (media.h:333-341; "temperature", media.cpp:276-302) or NanoVDBMedium (media.h:724-735;
     * config 5 "explosion"): as many samples as `density`, same layout -- for NanoVDB the same index bounding box and index-to-world
     * map (the .nvdb reader checks).  Where the scaled temperature T' = (T(p) - temperature_offset) * temperature_scale exceeds 100,
     *   Le(p) = scale * BlackbodySpectrum(T').Sample(lambda),  scale = nvdb_le_scale (NanoVDB) / the le_scale grid's value (GridMedium)
     * at the path's three sampled wavelengths, which the RGB build stores as R, G, B.  The path adds volume
     * emission in the DELTA-TRACKING callback only (guidedvolpathvspgintegrator.cpp:895-906): under "vspsamplingmethod" "nds";
     * a heterogeneous medium under the default "resampling" never evaluates it and the grid has no effect on
     * the result.  A GridMedium with both Le and temperature is VSPG_EINVAL (media.cpp:307-308).  HOST pointer; NULL = none. */
    const float *temperature;
    float nvdb_le_scale, temperature_offset, temperature_scale;
} VspgMedium;

/* Triangle geometry (src/pbrt/shapes.h:828-1030, shapes.cpp:168-262, cpu/aggregates.cpp:529-640):
 * a soup of diffuse, non-emissive, two-sided triangles next to the rectangles; the library builds a BVH over them
 * (own builder) and traverses it on the device.  tri_p: 9 floats per triangle (p0, p1, p2), tri_kd: 3 per triangle.
 * HOST pointers, copied at create time. */
/* Infinite lights (src/pbrt/lights.h:  UniformInfiniteLight :554-601, DistantLight :207-250; escaped-ray MIS
 * guidedvolpathvspgintegrator.cpp:353-374): L = scale * Lemit already multiplied out (RGB).  Distant: w_light = the
 * normalised direction TOWARDS the light in render space. */
enum { VSPG_LIGHT_UNIFORM_INFINITE = 0, VSPG_LIGHT_DISTANT = 1 };
#define VSPG_MAX_INFINITE_LIGHTS 4
typedef struct VspgInfiniteLight {
    int32_t type;
    float L[3];
    float w_light[3];
} VspgInfiniteLight;

/* tri_flags (optional, one int32 per triangle, HOST pointer; NULL = all 0): bit 0 = VSPG_MATERIAL_INTERFACE, bits 1-2 =
 * the VSPG_IFACE_* bits, bit 3 = flip the geometric normal (the mesh's reverseOrientation ^ transformSwapsHandedness,
 * shapes.h:934-936 -- only a medium transition can tell the two sides of a diffuse triangle apart). */
#define VSPG_TRI_INTERFACE 1
#define VSPG_TRI_IFACE_SHIFT 1
#define VSPG_TRI_FLIP_NORMAL 8
typedef struct VspgScene {
    int32_t n_quads;
    VspgQuad quads[VSPG_MAX_QUADS];
    VspgCamera camera;
    VspgMedium medium;
    int32_t n_triangles;
    const float *tri_p;
    const float *tri_kd;
    int32_t n_infinite_lights;
    VspgInfiniteLight infinite_lights[VSPG_MAX_INFINITE_LIGHTS];
    /* round 4: medium boundaries */
    const int32_t *tri_flags;
    int32_t n_spheres;
    VspgSphere spheres[VSPG_MAX_SPHERES];
    /* the camera's medium (CameraBase::medium = the current OUTSIDE medium at the Camera directive, scene.cpp:153-155):
     * 0 = the scene's medium (as ever), 1 = none -- the camera looks at the volume from outside */
    int32_t camera_outside_medium;
} VspgScene;

/* ---- integrator parameters: same names and defaults as
 * GuidedVolPathVSPGIntegrator::Create (src/pbrt/cpu/guidedvolpathvspgintegrator.cpp:
 * 1260-1322).  Use vspg_integrator_params_default() then override. ------------------ */
enum { VSPG_GUIDE_MIS = 0, VSPG_GUIDE_RIS = 1 };              /* guiding.h:52-55 */
enum { VSPG_VSP_CONTRIBUTION = 0, VSPG_VSP_VARIANCE = 1 };    /* "vspcriterion" */
enum { VSPG_VSP_RESAMPLING = 0, VSPG_VSP_NDS = 1 };           /* "vspsamplingmethod" */
enum { VSPG_LIGHTSAMPLER_UNIFORM = 0, VSPG_LIGHTSAMPLER_POWER = 1, VSPG_LIGHTSAMPLER_BVH = 2 };

typedef struct VspgIntegratorParams {
    int32_t maxdepth;                  /* 5 */
    int32_t minrrdepth;                /* 1 */
    int32_t usenee;                    /* true */
    int32_t surfaceguiding;            /* true  (directional guiding of BSDF sampling) */
    int32_t volumeguiding;             /* true  (directional guiding of phase sampling) */
    int32_t surfaceguidingtype;        /* "ris" */
    int32_t volumeguidingtype;         /* "mis" */
    int32_t vspguiding;                /* true */
    int32_t vspprimaryguiding;         /* true */
    int32_t vspsecondaryguiding;       /* true */
    float vspmisratio;                 /* 0.5 */
    int32_t vspcriterion;              /* "variance" */
    int32_t vspsamplingmethod;         /* "resampling" */
    int32_t collisionProbabilityBias;  /* false (NDS+: needs a transmittance buffer, vspg_renderer_set_tr_buffer) */
    int32_t rrguiding;                 /* false: guided Russian roulette (needs the image-space contribution estimate) */
    int32_t lightsampler;              /* "bvh" */
    int32_t regularize;                /* false (no-op for diffuse BxDFs) */
    int32_t guide_num_training_waves;  /* 128, hidden constant integrators.h:502 */
    int32_t storeTrBuffer;             /* false: record the primary rays' transmittance (TrBuffer) for read-back */
    int32_t surfacerrguiding;          /* true  (with rrguiding: guided survival probability at surface vertices, else 1) */
    int32_t volumerrguiding;           /* true  (same for volume vertices) */
} VspgIntegratorParams;

/* Film "rgb" + Sampler "independent" + PixelFilter "box" (SURVEY.md App. F). */
typedef struct VspgRenderConfig {
    int32_t xres, yres;   /* Film xresolution / yresolution */
    int32_t spp;          /* Sampler "pixelsamples" (only used for bookkeeping) */
    int32_t seed;         /* --seed / sampler "seed" (src/pbrt/samplers.h:457-460) */
    /* multi-GPU sharding of sample indices: this renderer handles wave w iff
     * w % shard_count == shard_index.  1-GPU: index 0, count 1. */
    int32_t shard_index, shard_count;
    int32_t device;       /* HIP device ordinal */
} VspgRenderConfig;

typedef struct VspgCounters {
    uint64_t paths;             /* camera samples run to termination */
    uint64_t segments;          /* iterations of the Li() path loop (roofline unit, 8d) */
    uint64_t volume_scatters;   /* real scattering events ("Volume interactions") */
    uint64_t surface_hits;      /* "Surface interactions" */
    uint64_t density_queries;   /* tentative collisions ("Integrator/Density query") */
    uint64_t shadow_rays;
    uint64_t shadow_density_queries; /* tentative collisions of the NEE shadow rays' ratio tracking in a HETEROGENEOUS medium (8 voxels + a
                                        majorant each, like density_queries; the reference's densityQueryCount does not count them) */
} VspgCounters;

/* ---- guiding field (spatial-directional cache) -------------------------------------------
 * Stands in for an openpgl::cpp::Field after Field::Update / Field(file)
 * (guidedvolpathvspgintegrator.cpp:111-128, 234-246).  OpenPGL is not part of the reference tree,
 * so the layout and the mixture math are this build's own design: a kd-tree over
 * positions whose leaves hold a parallax-aware von-Mises-Fisher mixture of the incident radiance
 * plus a per-lobe volume-scatter-probability estimate.  The renderer trains the field in-loop (Field::Update,
 * SURVEY 8a row a18; see "guiding-cache training" below) or takes a trained one (vspg_renderer_set_guiding_field). */
#define VSPG_FIELD_LOBES 8
typedef struct VspgKdNode {
    float split;      /* inner node: split plane position along `axis` */
    uint32_t packed;  /* bits 0-1: axis 0..2, 3 = leaf; bits 2-31: left child index (right = left+1),
                         or the region index for a leaf */
} VspgKdNode;
typedef struct VspgFieldRegion {
    float pivot[3];                    /* reference point of the lobes' parallax distances */
    int32_t n_lobes;                   /* 0 = region not trained: Init() fails there */
    float weight[VSPG_FIELD_LOBES];    /* mixture weights, sum 1 */
    float kappa[VSPG_FIELD_LOBES];     /* vMF concentrations */
    float mu[3][VSPG_FIELD_LOBES];     /* vMF mean directions (unit), SoA */
    float distance[VSPG_FIELD_LOBES];  /* distance of the lobe's source from the pivot; +inf = none */
    float vsp[VSPG_FIELD_LOBES];       /* volume scatter probability along the lobe, in [0,1] */
} VspgFieldRegion;
typedef struct VspgField {
    int32_t n_nodes, n_regions;
    const VspgKdNode *nodes;           /* HOST pointers; node 0 is the root */
    const VspgFieldRegion *regions;
} VspgField;

/* ---- guiding-cache training -------------------------------------------
 * With surfaceguiding / volumeguiding / vspsecondaryguiding set and no field uploaded, the renderer
 * trains the field itself, like the reference (guideTraining, guidedvolpathvspgintegrator.cpp:109,
 * 230-248): every path records its vertices (guiding.h:682-832), turns them into radiance samples when
 * it ends (PathSegmentStorage::PropagateSamples, :627) and vspg_post_process_wave refits the field
 * (Field::Update, :239) while fewer than guide_num_training_waves updates have run. */
typedef struct VspgTrainSample {
    float p[3];       /* vertex position */
    float dir[3];     /* sampled direction at the vertex (direction the radiance arrives from) */
    float weight;     /* incident radiance estimate / pdf */
    float pdf;        /* pdf the direction was sampled with */
    float distance;   /* distance to the next vertex along dir */
    uint32_t flags;   /* VSPG_SAMPLE_* */
} VspgTrainSample;
#define VSPG_SAMPLE_VOLUME 1u       /* vertex lies in the medium -> volume field */
#define VSPG_SAMPLE_NEXT_VOLUME 2u  /* the next event along dir was a volume scatter (VSP statistics) */
typedef struct VspgTrainStats {
    int32_t training;      /* 1 while the field is still being trained */
    int32_t iteration;     /* Field::GetIteration(): number of updates done */
    uint64_t n_samples;    /* radiance samples recorded since the last update */
    uint64_t n_zero;       /* zero-valued samples dropped since the last update */
    int32_t n_nodes[2], n_regions[2];  /* [0] surface field, [1] volume field */
    uint64_t n_dropped;    /* samples beyond the sample buffer's capacity (xres*yres*(maxdepth+1), at most 2^26) since the
                            * last update: a training vspg_render_wave() over SEVERAL sample indices runs them as 1-spp
                            * launches into one buffer that the single following vspg_post_process_wave() consumes --
                            * call render_wave / post_process_wave per sample index (the reference's waves) to lose none */
} VspgTrainStats;

typedef struct VspgRenderer VspgRenderer; /* opaque */

/* ---- helpers (host only, no device needed) ----------------------------------------- */
int vspg_abi_version(void);
const char *vspg_last_error(void);
void vspg_integrator_params_default(VspgIntegratorParams *p);
/* pbrt "LookAt" + Camera "perspective" "float fov" (fov spans the shorter image axis,
 * src/pbrt/cameras.cpp:474-489), world == render space. */
int vspg_camera_look_at(VspgCamera *cam, const float eye[3], const float look[3],
                        const float up[3], float fov_degrees, int xres, int yres);
/* inv = m^-1 for an affine row-major 4x4 (computed in double, rounded once); returns VSPG_EINVAL for a singular matrix. */
int vspg_transform_inverse(const float m[16], float inv[16]);
/* Fills `scene` with the App.-F fog box: box [-1,1]^3, Kd .73 walls, 0.5x0.5 ceiling
 * light Le (17,12,4) at y=.999, homogeneous fog sigma_a .05 sigma_s .45 g 0, camera at
 * (0,0,-.95) looking +z, fov 60. */
int vspg_scene_fog_box(VspgScene *scene, int xres, int yres);

/* ---- renderer life cycle ----------------------------------------------------------- */
/* Replaces Integrator::Create("guidedvolpathvspg", ...) + the integrator constructor
 * (src/pbrt/cpu/integrators.cpp:3739-3744, guidedvolpathvspgintegrator.cpp:61-198):
 * validates parameters, uploads scene and medium, allocates film, image-space VSP buffer
 * and path-state queues on `cfg->device`. */
int vspg_renderer_create(const VspgScene *scene, const VspgIntegratorParams *params,
                         const VspgRenderConfig *cfg, VspgRenderer **out);
/* Replaces ~GuidedVolPathVSPGIntegrator (guidedvolpathvspgintegrator.cpp:200-228). */
int vspg_renderer_destroy(VspgRenderer *r);

/* Uploads trained guiding fields: surface_field feeds SurfaceSamplingDistribution::Init
 * (guiding.h:90), volume_field feeds VolumeSamplingDistribution::Init (guiding.h:388); either may be
 * NULL (= untrained, Init() returns false as OpenPGL does before the first Field::Update).  Until
 * this is called a renderer created with surfaceguiding / volumeguiding / vspsecondaryguiding
 * refuses to render (VSPG_ESCOPE): the reference would train the field itself. */
int vspg_renderer_set_guiding_field(VspgRenderer *r, const VspgField *surface_field,
                                    const VspgField *volume_field, void *stream);

/* Replaces one wave of ImageTileIntegrator::Render -- the ParallelFor2D over all pixels
 * for sample indices [wave_start, wave_end) (src/pbrt/cpu/integrators.cpp:183-207),
 * i.e. EvaluatePixelSample -> Li -> SampleDistance -> film.AddSample for every pixel.
 * Asynchronous on `stream` -- with one exception: a scene with MEDIUM BOUNDARIES over a grid / NanoVDB medium (the wavefront
 * pipeline's boundary flavour) runs path-loop iterations until its path list is empty, and the emptiness test is a host read:
 * the call then synchronises `stream` once per iteration past the ones every pass needs and once at the end of every sample
 * pass (a host that overlaps work on other streams should issue that work first).  A pass whose list does not run dry
 * within the iteration cap returns VSPG_ESCOPE after the samples of the paths that did finish have entered the film. */
int vspg_render_wave(VspgRenderer *r, int wave_start, int wave_end, void *stream);

/* Replaces GuidedVolPathVSPGIntegrator::PostProcessWave
 * (guidedvolpathvspgintegrator.cpp:230-260): waveCounter++, image-space VSP buffer update
 * when waveCounter == 2^bufferWave.  Asynchronous on `stream`. */
int vspg_post_process_wave(VspgRenderer *r, void *stream);

/* The same after a STEP that covered n_waves sample indices -- a sharded render: every rank runs its own
 * sample index of the step, then all ranks post-process with n_waves = the number of ranks.  waveCounter += n_waves;
 * the image-space buffer update runs on the step that takes waveCounter to or past 2^bufferWave (n_waves = 1: exactly
 * the reference's schedule, :251).  isg_stats_sum: DEVICE pointer to W*H*VSPG_ISG_STATS floats holding the SUM over
 * all ranks of the per-rank statistics (vspg_isg_stats_device_ptr, all-reduced by the caller -- only needed when
 * vspg_isg_update_due() says the update falls on this step), or NULL to use the renderer's own.  The renderer's own
 * statistics are never overwritten, so nothing is counted twice.  N ranks stepping this way compute what ONE renderer
 * computes that renders [w, w + N) per step and post-processes with n_waves = N (up to float summation order of the
 * statistics).  vspg_post_process_wave(r, s) == vspg_post_process_step(r, 1, NULL, s). */
int vspg_isg_update_due(VspgRenderer *r, int n_waves);
int vspg_post_process_step(VspgRenderer *r, int n_waves, const float *isg_stats_sum, void *stream);

/* Sharded guiding-field training ("all-reduce guiding samples / statistics per wave").  Field::Update (:239) fits
 * the field from sufficient statistics accumulated over the step's radiance samples.  With an exchange hook installed, the
 * update sums those statistics over the ranks at its accumulation points -- the sample count and weight, then (both fields
 * in one buffer) the position statistics before the split, after it, and the EM step's: five in-place sum all-reduces of at
 * most 2.3 MB per training step -- and continues from the sums, so EVERY rank fits the SAME field from ALL ranks' samples: N ranks stepping
 * this way train what one renderer trains that renders [w, w + N) per step (up to float summation order; the ranks' fields
 * are bit-identical to each other when the all-reduce hands every rank the same bits, as RCCL's and gloo's do).
 * `fn` sums n_floats floats at dev_ptr in place over the ranks, enqueued on `stream`; it returns 0 or an error code that
 * vspg_post_process_step passes on.  Every rank must install a hook (or none): the hook is called the same number of times
 * on every rank -- the decision to update is taken on the summed sample count.  NULL removes it. */
typedef int (*VspgExchangeFn)(float *dev_ptr, size_t n_floats, void *stream, void *user);
int vspg_renderer_set_exchange(VspgRenderer *r, VspgExchangeFn fn, void *user);

/* Name of the kernel instantiation vspg_render_wave launches for this renderer as it stands (bench / profile
 * bookkeeping; static storage). */
const char *vspg_renderer_kernel_name(VspgRenderer *r);

/* The arithmetic the path kernels compute in (ABI 7; csrc/vspg_arith.h).  The reference has no such switch: its float arithmetic is
 * what its compiler makes of src/pbrt/util/sampling.h:222-225, media_sampleTMaj.h:379-404 ... -- IEEE division, glibc's libm.
 *   VSPG_ARITH_EXACT         (default) that arithmetic bit for bit: the mode every parity test and the benchmark's `value` run in;
 *   VSPG_ARITH_FAST_WEIGHTS  quotients that only scale a path's contribution through v_rcp_f32; every path keeps the oracle's
 *                            trajectory, radiance agrees to float rounding (relMSE vs the exact film ~1e-13);
 *   VSPG_ARITH_FAST          every division / square root at 2.5 ulp, the hardware's log / sin / cos: the same estimator, but not the
 *                            oracle's paths (the reference seeds a shadow ray's RNG from the bits of its origin and direction,
 *                            guidedvolpathvspgintegrator.cpp:1193) -- equal in distribution, validated statistically.
 * Returns VSPG_ESCOPE (and names the kernel) for configurations without tolerance-mode instantiations: they cover unguided renders of
 * rectangle scenes over a homogeneous medium and unguided "resampling" renders over a "uniformgrid" medium. */
#define VSPG_ARITH_EXACT 0
#define VSPG_ARITH_FAST_WEIGHTS 1
#define VSPG_ARITH_FAST 2
int vspg_renderer_set_arithmetic(VspgRenderer *r, int mode);
int vspg_renderer_get_arithmetic(VspgRenderer *r);

/* Film access.  The film is W*H float4 {sum w*r, sum w*g, sum w*b, sum w} in HBM
 * (the accumulate contract of RGBFilm::AddSample, src/pbrt/film.h:251-267, in float).
 * vspg_film_device_ptr exposes it for the frame-end RCCL all-reduce.
 * A one-sample vspg_render_wave may leave its samples parked beside the film until the next launch starts (it adds them as each
 * pixel's new path begins, in the same order as ever); every call that reads or writes the film or the image-space statistics
 * adds them first, on the stream it is given (ordered behind the launch that parked them by an event when the streams differ).
 * The two *_device_ptr getters have no stream: they add the parked samples on the stream of the launch that parked them and
 * wait for that stream, so what the pointer shows is complete AT THE TIME OF THE CALL.  The pointer VALUE is fixed for the
 * renderer's lifetime, the CONTENTS are not: a host that keeps the pointer across waves (to all-reduce the film or the
 * statistics plane itself) must call vspg_flush(r, stream) -- asynchronous, no host wait -- after the last vspg_render_wave and
 * before it reads through the pointer on `stream` (or on work ordered behind it); a no-op when nothing is parked. */
int vspg_flush(VspgRenderer *r, void *stream);
int vspg_film_device_ptr(VspgRenderer *r, float **dev_ptr, size_t *n_floats);
int vspg_film_read(VspgRenderer *r, float *host_rgbw /* W*H*4 */, void *stream);
int vspg_film_clear(VspgRenderer *r, void *stream);

/* Image-space VSP buffer (stands in for openpgl ImageSpaceGuidingBuffer,
 * guidedvolpathvspgintegrator.cpp:161-178, 1098-1112). */
int vspg_vsp_buffer_device_ptr(VspgRenderer *r, float **dev_ptr, size_t *n_floats);
int vspg_vsp_buffer_read(VspgRenderer *r, float *host_vsp /* W*H */, int *is_ready,
                         void *stream);
/* ImageSpaceGuidingBuffer(fileName) (guidedvolpathvspgintegrator.cpp:151-159): the buffer is used as handed over
 * (ready from the first wave on) and never updated again (calculateImageSpaceGuidingBuffer = false, :251-256).
 * Values outside [0,1] mean "no estimate for this pixel" (:1101-1112). */
int vspg_vsp_buffer_load(VspgRenderer *r, const float *host_vsp /* W*H */, void *stream);
/* per-pixel sufficient statistics (W*H*VSPG_ISG_STATS floats) for multi-GPU all-reduce */
#define VSPG_ISG_STATS 8
int vspg_isg_stats_device_ptr(VspgRenderer *r, float **dev_ptr, size_t *n_floats);

/* Transmittance buffer (TrBuffer, src/pbrt/cpu/trbuffer.h:17-104): per-pixel running mean of the primary
 * ray's ratio-tracking transmittance estimate, recorded by the resampling routine
 * (guidedvolpathvspgintegrator.cpp:727-728) while params.storeTrBuffer is set (or NDS+ is requested and no
 * buffer was handed over).  get = what TrBuffer::Store writes (RGB per pixel, row-major; host_spp, optional:
 * the per-pixel sample counts, for merging the buffers of sample-sharded ranks).  set = TrBuffer(fileName):
 * the renderer stops recording and NDS+ (vspsamplingmethod "nds" + collisionProbabilityBias) biases the
 * primary ray's real/null-collision probability with it (:929-938). */
int vspg_renderer_get_tr_buffer(VspgRenderer *r, float *host_rgb /* W*H*3 */, int32_t *host_spp /* W*H or NULL */,
                                void *stream);
int vspg_renderer_set_tr_buffer(VspgRenderer *r, const float *host_rgb /* W*H*3 */, void *stream);

int vspg_get_counters(VspgRenderer *r, VspgCounters *out, void *stream);
int vspg_reset_counters(VspgRenderer *r, void *stream);

/* ---- parity / debug entry points ---------------------------------------------------- */
/* Runs EvaluatePixelSample for n explicit (pixel, sampleIndex) pairs with the renderer's
 * current VSP buffer and returns the camera-weighted radiance L (RGB) per path plus the
 * number of path-loop segments -- the analogue of --debugstart x,y,n
 * (src/pbrt/cpu/integrators.cpp:77-95).  Host arrays. */
int vspg_trace_paths(VspgRenderer *r, int n, const int32_t *pixel_xy /* 2n */,
                     const int32_t *sample_index /* n */, float *out_L /* 3n */,
                     int32_t *out_segments /* n or NULL */, void *stream);

/* Batch drivers of the free-flight layer on device, for bit-level checks against the
 * oracle and the reference known answers.  All arrays are HOST
 * arrays of length n (or 3n where noted). */
typedef struct VspgTmajQuery {
    float o[3], d[3];  /* ray (d need not be normalised) */
    float tMax;
    float u;           /* traversal sample */
    float rng_a, rng_b;/* RNG(Hash(rng_a), Hash(rng_b)), integrator.cpp:323-325 */
    float vsp;         /* <0: guideScatterDecision=false */
    int32_t channel;   /* hero channel lambda.ChannelIdx() */
    int32_t stop_after;/* callback returns false at this (1-based) callback; 0 = never */
} VspgTmajQuery;

typedef struct VspgTmajResult {
    float T_maj[3];       /* returned majorant transmittance */
    float r_u_factor[3];  /* OpticalDepthSpace only (else 1) */
    float last_t;         /* distance along the NORMALISED ray of the last callback, -1 if none */
    float last_p[3];      /* its position */
    int32_t n_callbacks;
    float sum_sigt_over_maj; /* sum over callbacks of (sigma_t/sigma_maj)[channel] */
    float vrc;            /* Resampling: volumeRatioZeroCandidateCompensation */
    float majorant_scale; /* Resampling */
} VspgTmajResult;

enum { VSPG_TMAJ_PLAIN = 0, VSPG_TMAJ_OPTICAL_DEPTH = 1, VSPG_TMAJ_RESAMPLING = 2 };

/* Ray queries against the renderer's geometry (ABI 7): the batch driver behind the replays of the reference's own shape tests
 * (src/pbrt/shapes_test.cpp: Triangle Watertight / Reintersect / BadCases, FullSphere Reintersect) -- what Integrator::Intersect
 * (cpu/integrators.cpp:341-349) returns for `ray`, then, from that hit, the ray Interaction::SpawnRay(w) (`mode` 1,
 * interaction.h:99-101) or Interaction::SpawnRayTo(point w) (`mode` 2, interaction.h:104-108, ray.h:93-98) intersected again with
 * tMax2: closest hit (Intersect) and any hit (IntersectP). */
typedef struct {
    float o[3], d[3], tMax;
    int32_t mode;          /* 0: the first intersection only; 1: SpawnRay(w); 2: SpawnRayTo(w) */
    float w[3], tMax2;
} VspgRayQuery;
typedef struct {
    int32_t hit;           /* the first ray hit something */
    int32_t prim;          /* rectangle index | 1000000 + triangle index in the caller's soup | 2000000 + sphere index */
    float t, p[3], n[3];   /* tHit, the interaction point (midpoint of pi), the surface normal */
    float o2[3], d2[3];    /* the spawned ray */
    int32_t hit2, any2;    /* Intersect / IntersectP of the spawned ray */
    float t2;
} VspgRayResult;
int vspg_ray_batch(VspgRenderer *r, int n, const VspgRayQuery *q, VspgRayResult *out, void *stream);
/* Replaces SampleT_maj / SampleT_maj_OpticalDepthSpace / SampleT_maj_Resampling
 * (src/pbrt/media_sampleTMaj.h:49-117, 269-491, 136-248) on the renderer's medium with a
 * recording callback. */
int vspg_sample_tmaj_batch(VspgRenderer *r, int variant, int n, const VspgTmajQuery *q,
                           VspgTmajResult *out, void *stream);

/* Primitive batch (bit-exact layer): for each i computes on device
 *   hash[i]   = Hash(f[i])                      (src/pbrt/util/hash.h:100)
 *   rng_u32[i]= RNG(Hash(f[i]),Hash(g[i])).Uniform<uint32_t>()  (util/rng.h:82-88,119-125)
 *   fastexp[i]= FastExp(f[i])                   (util/math.h:450-474)
 */
int vspg_primitives_batch(VspgRenderer *r, int n, const float *f, const float *g,
                          uint64_t *hash, uint32_t *rng_u32, float *fastexp, void *stream);

/* Batch driver of the guiding-cache query (parity tests): for each i initialises the distribution
 * at p[i] (surface: cosine product with n[i]; volume: HG product with wo = n[i], asymmetry g) and
 * returns PDF(wi[i]), IncomingRadiancePDF(wi[i]), VolumeScatterProbability(wi[i]) and one sample
 * SamplePDF(u[i]) -> (ws[i], pdf_s[i]).  out_ok[i] = Init() result.  HOST arrays. */
int vspg_guiding_query_batch(VspgRenderer *r, int is_volume, float g, int n, const float *p /*3n*/,
                             const float *n_or_wo /*3n*/, const float *wi /*3n*/, const float *u /*2n*/,
                             int32_t *out_ok, float *out_pdf, float *out_incoming_pdf, float *out_vsp,
                             float *out_ws /*3n*/, float *out_pdf_s, void *stream);

/* Device float libm batch: logf(x), sinf(x), cosf(x) as the kernels evaluate them; they must
 * equal the host libm the CPU reference run uses (std::log/std::sin/std::cos of float,
 * src/pbrt/util/sampling.h:222-225, 325-341, src/pbrt/util/vecmath.h:1666-1672). */
/* Training state / radiance samples recorded since the last update (test + diagnostics; order is
 * unspecified) / the field as it stands (Field::Store counterpart: pass NULL arrays to get the sizes). */
int vspg_renderer_training_stats(VspgRenderer *r, VspgTrainStats *out, void *stream);
int vspg_train_samples_read(VspgRenderer *r, VspgTrainSample *out, size_t max_samples, size_t *n_out,
                            void *stream);
int vspg_renderer_get_guiding_field(VspgRenderer *r, int volume_field, VspgKdNode *nodes,
                                    VspgFieldRegion *regions, int32_t *n_nodes, int32_t *n_regions,
                                    void *stream);

int vspg_libm_batch(VspgRenderer *r, int n, const float *x, float *logf_out, float *sinf_out,
                    float *cosf_out, void *stream);
/* out[i] = (float)(-log(1.0 - (double)x[i])) as the kernels evaluate it: the DOUBLE-precision
 * std::log of the optical-depth-space distance sampling (src/pbrt/media_sampleTMaj.h:379-404). */
int vspg_libm_log1m_batch(VspgRenderer *r, int n, const float *x, float *out, void *stream);
/* out[i] = powf(x[i], y[i]) as the kernels evaluate it (the std::pow of the NDS+ bias,
 * src/pbrt/cpu/guidedvolpathvspgintegrator.cpp:937): bit-identical to glibc 2.35's powf. */
int vspg_libm_powf_batch(VspgRenderer *r, int n, const float *x, const float *y, float *out, void *stream);
/* out6[6 i ..] = {lambda_0..2, Le_0..2}: the wavelengths of SampledWavelengths::SampleVisible(u[i]) (util/spectrum.h:369-386; the
 * host's atanhf, bit for bit) and BlackbodySpectrum(T[i]).Sample(lambda) (:568-588) as the kernels evaluate a temperature grid's
 * emission (media.h:333-341, :724-735). */
int vspg_blackbody_batch(VspgRenderer *r, int n, const float *u, const float *T, float *out6, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* VSPG_H */

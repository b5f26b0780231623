/*
 * vspg_oracle.h -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * A plain-C restatement of the reference algorithm for the GuidedVolPathVSPG hot path
 * (kehanxuuu/vspg-pbrt-v4; every function in vspg_oracle.c cites the reference file:line
 * it follows).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / CPU baseline.  The product
 * (vspg-pbrt-v4_amd/) never links, loads or calls anything in oracle/.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - primitives a1-a4 (RNG, Hash, FastExp, SampleExponential, SampleDiscrete, HG, sphere /
 *     hemisphere sampling, OffsetRayOrigin, IndependentSampler): pinned bit-for-bit against
 *     oracle/_ref (the reference's own headers compiled in place) and the reference's
 *     unit-test fixtures.
 *   - SampleT_maj / _OpticalDepthSpace / _Resampling: pinned by the known answers recorded
 *     in SURVEY.md App. D.3 (outputs of the reference's own code) + analytic known answers.
 *   - SampleDistance / SampleLd / Li: restated from text; the TU needs OpenPGL/OpenEXR and
 *     cannot be built here -> pinned only through the layers above and analytic checks.
 *   - image-space VSP buffer + StandardThroughputBasedRussianRoulette: live in OpenPGL,
 *     which is absent from the reference tree -> own design, PARITY UNPINNED.
 *
 * Input PODs are shared with the product's public header so both sides see identical
 * bytes; no code is shared.
 */
#ifndef VSPG_ORACLE_H
#define VSPG_ORACLE_H

#include "../include/vspg.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct OracleRenderer OracleRenderer;

/* primitives, exported for golden-vector tests */
uint64_t oracle_murmur64a(const unsigned char *key, size_t len, uint64_t seed);
uint64_t oracle_mix_bits(uint64_t v);
uint64_t oracle_hash_float(float f);
uint64_t oracle_hash_pixel_seed(int32_t x, int32_t y, int32_t seed);
uint64_t oracle_hash_point3(float x, float y, float z);
void oracle_rng_seq(uint64_t seq, uint64_t seed, int has_seed, int64_t advance, int n,
                    uint32_t *out_u32, float *out_f);
float oracle_fast_exp(float x);
float oracle_sample_exponential(float u, float a);
int oracle_sample_discrete2(float w0, float w1, float u);
float oracle_henyey_greenstein(float cosTheta, float g);
void oracle_sample_henyey_greenstein(const float wo[3], float g, float u0, float u1,
                                     float wi[3], float *pdf);
void oracle_sample_uniform_sphere(float u0, float u1, float out[3]);
void oracle_sample_cosine_hemisphere(float u0, float u1, float out[3]);
void oracle_coordinate_system(const float v[3], float v2[3], float v3[3]);
/* Frame::FromXZ(x, z) (its y axis), Frame::ToLocal(v), Frame::FromLocal(v) */
void oracle_frame_xz(const float x[3], const float z[3], const float v[3], float out_y[3], float out_local[3], float out_world[3]);
/* SpawnRayTo(Point3fi pFrom, Normal3f nFrom, time, Point3fi pTo, Normal3f nTo) (ray.h:103-108): the NEE shadow ray */
void oracle_spawn_ray_to(const float pf[3], const float pferr[3], const float nf[3], const float pt[3], const float pterr[3],
                         const float nt[3], float out_o[3], float out_d[3]);
/* Transform::ApplyInverse(Ray, &tMax) with an identity matrix */
void oracle_apply_inverse_identity(const float o[3], const float d[3], float tMax, float out_o[3], float *out_tMax);
/* Bounds3::IntersectP(o, d, tMax, &t0, &t1) (returns hit) and Bounds3::Offset(o) */
int oracle_bounds3(const float bmin[3], const float bmax[3], const float o[3], const float d[3], float tMax, float t01[2],
                   float offset[3]);
void oracle_offset_ray_origin(const float p[3], const float perr[3], const float n[3],
                              const float w[3], float out[3]);
/* IndependentSampler: first n Get1D() values of (pixel, sampleIndex) */
void oracle_independent_sampler(int32_t px, int32_t py, int32_t seed, int32_t sampleIndex,
                                int n, float *out);

/* renderer */
int oracle_renderer_create(const VspgScene *scene, const VspgIntegratorParams *params,
                           const VspgRenderConfig *cfg, OracleRenderer **out);
void oracle_renderer_destroy(OracleRenderer *r);
/* all pixels, sample indices [wave_start, wave_end); nthreads<=0 -> all cores */
int oracle_render_wave(OracleRenderer *r, int wave_start, int wave_end, int nthreads);
/* a rectangular pixel window only (bounded CPU-baseline sample) */
int oracle_render_window(OracleRenderer *r, int x0, int y0, int x1, int y1, int wave_start,
                         int wave_end, int nthreads);
int oracle_post_process_wave(OracleRenderer *r);
void oracle_apply_inverse_ray(const float minv[16], const float o[3], const float d[3], float tMax, float out_o[3], float out_d[3], float *out_tMax);
void oracle_apply_inverse_point(const float minv[16], const float p[3], float out[3]);
int oracle_isg_update_due(OracleRenderer *r, int n_waves);
int oracle_post_process_step(OracleRenderer *r, int n_waves, const float *stats_sum);
void oracle_film_read(OracleRenderer *r, float *rgbw /* W*H*4, from double accum */);
void oracle_film_read_f64(OracleRenderer *r, double *rgbw);
void oracle_film_clear(OracleRenderer *r);
void oracle_vsp_buffer_read(OracleRenderer *r, float *vsp, int *is_ready);
void oracle_vsp_buffer_write(OracleRenderer *r, const float *vsp, int is_ready);
void oracle_vsp_buffer_load(OracleRenderer *r, const float *vsp); /* ImageSpaceGuidingBuffer(fileName): used as is, never updated */
void oracle_isg_stats_read(OracleRenderer *r, float *stats);
/* TrBuffer (cpu/trbuffer.h): read the running-mean transmittance (xres*yres*3, VSPG_EINVAL if none is kept);
 * write = TrBuffer(fileName): the renderer then uses it (NDS+) and stops recording */
int oracle_tr_buffer_read(OracleRenderer *r, float *rgb);
int oracle_tr_buffer_write(OracleRenderer *r, const float *rgb);
void oracle_get_counters(OracleRenderer *r, VspgCounters *out);
void oracle_reset_counters(OracleRenderer *r);
int oracle_trace_paths(OracleRenderer *r, int n, const int32_t *pixel_xy,
                       const int32_t *sample_index, float *out_L, int32_t *out_segments);
int oracle_sample_tmaj_batch(OracleRenderer *r, int variant, int n, const VspgTmajQuery *q,
                             VspgTmajResult *out);
/* Integrator::Intersect for n rays, then SpawnRay / SpawnRayTo from the hit and Intersect / IntersectP of that ray (VspgRayQuery) */
int oracle_ray_batch(OracleRenderer *r, int n, const VspgRayQuery *q, VspgRayResult *out);

int oracle_renderer_set_guiding_field(OracleRenderer *r, const VspgField *surface_field,
                                      const VspgField *volume_field);
/* lightSampler.Sample(ctx, u) / PMF(ctx, light) of the renderer's light sampler (prm.lightsampler) for n contexts {p, ns} */
int oracle_light_sample_batch(OracleRenderer *r, int n, const float *p, const float *ns, const float *u, int32_t *out_light, float *out_pmf);
int oracle_light_pmf_batch(OracleRenderer *r, int n, const float *p, const float *ns, const int32_t *light, float *out_pmf);
int oracle_guiding_query_batch(OracleRenderer *r, int is_volume, float g, int n, const float *p,
                               const float *n_or_wo, const float *wi, const float *u, int32_t *out_ok,
                               float *out_pdf, float *out_incoming_pdf, float *out_vsp, float *out_ws,
                               float *out_pdf_s);

/* guiding-cache training (a18; the OpenPGL side -- PropagateSamples, Field::Update -- is this build's
 * own design, PARITY UNPINNED; the recording hooks follow src/pbrt/cpu/guiding.h:682-832) */
int oracle_renderer_training_stats(OracleRenderer *r, VspgTrainStats *out);
int oracle_train_samples_read(OracleRenderer *r, VspgTrainSample *out, size_t max_samples, size_t *n_out);
int oracle_renderer_get_guiding_field(OracleRenderer *r, int volume_field, VspgKdNode *nodes,
                                      VspgFieldRegion *regions, int32_t *n_nodes, int32_t *n_regions);

/* scene helpers restated independently of the product's (same formulas, separate code) */
void oracle_integrator_params_default(VspgIntegratorParams *p);
int oracle_camera_look_at(VspgCamera *cam, const float eye[3], const float look[3],
                          const float up[3], float fov_degrees, int xres, int yres);
int oracle_scene_fog_box(VspgScene *scene, int xres, int yres);

#ifdef __cplusplus
}
#endif
#endif

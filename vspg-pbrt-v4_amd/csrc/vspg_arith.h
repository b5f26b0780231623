// vspg_arith.h -- the arithmetic a translation unit's path kernels compute in.
//
// VSPG_ARITH (a compiler flag of the translation unit, csrc/Makefile):
//   0  EXACT (default; vspg_capi.hip, vspg_wf_grid.hip, vspg_wf_nvdb.hip): the reference's arithmetic bit for bit -- IEEE division
//      and square root, glibc's logf / sinf / cosf / double log restated operation by operation (vspg_libm.h).  Every parity test
//      runs this; `value` of the benchmark is this.
//   1  FAST WEIGHTS (vspg_fast.hip -> vspg_fast1.o): quotients that only scale a path's CONTRIBUTION -- throughput, MIS weights,
//      light pdfs, Russian-roulette probabilities, reservoir weights -- are a * v_rcp_f32(b) (wdiv below, ~2 ulp).  Everything a
//      path's TRAJECTORY is made of stays exact: hit distances, free-flight distances, directions, spawn points -- because the
//      reference seeds a shadow ray's random numbers from the BIT PATTERNS of its origin and direction
//      (guidedvolpathvspgintegrator.cpp:1193: RNG(Hash(lightRay.o), Hash(lightRay.d))), a last-ulp change of a vertex position redraws
//      that vertex's whole transmittance estimate.  Paths stay the oracle's paths (up to measure-zero decision flips: a uniform
//      number within 1e-7 of a probability); radiance agrees to float rounding.
//   2  FAST (vspg_fast.hip -> vspg_fast2.o, built with -fno-hip-fp32-correctly-rounded-divide-sqrt): every float division and
//      square root at 2.5 ulp, v_log_f32 / v_sin_f32 / v_cos_f32 where the reference calls libm, float instead of double for the
//      -log(1 - u) of media_sampleTMaj.h.  Integer work (PCG, Murmur, DDA indices) and FastExp's polynomial are untouched.  Same
//      estimator, same sampler dimensions, same decisions up to rounding -- but NOT the oracle's paths: see mode 1 for why.  It is
//      validated statistically (tests/test_fast_arith.py).
//
// The modes coexist in one library: a fast translation unit's symbols live in an inline namespace of their own (every kernel,
// host template and device function), the C-ABI picks per renderer (vspg_renderer_set_arithmetic, include/vspg.h).  This is a
// build configuration of the same CDNA4 code, not a second code path.
#pragma once
#ifndef VSPG_ARITH
#define VSPG_ARITH 0
#endif
#if VSPG_ARITH == 0
#define VSPG_NS_BEGIN namespace vspg {
#define VSPG_NS_END }
#elif VSPG_ARITH == 1
#define VSPG_NS_BEGIN namespace vspg { inline namespace fastw {
#define VSPG_NS_END } }
#else
#define VSPG_NS_BEGIN namespace vspg { inline namespace fast {
#define VSPG_NS_END } }
#endif

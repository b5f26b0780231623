// Test shim: exposes the product's host-exact float functions (vspg_libm.h, the same header the
// HIP kernels use) and the running libm's logf/sinf/cosf/log side by side, for bitwise comparison.
#include <math.h>
#include <stdint.h>

#include "../vspg-pbrt-v4_amd/csrc/vspg_libm.h"

extern "C" {
void model_logf(int n, const float *x, float *y) { for (int i = 0; i < n; ++i) y[i] = vspg_libm::logf_host_exact(x[i]); }
void model_sinf(int n, const float *x, float *y) { for (int i = 0; i < n; ++i) y[i] = vspg_libm::sinf_host_exact(x[i]); }
void model_cosf(int n, const float *x, float *y) { for (int i = 0; i < n; ++i) y[i] = vspg_libm::cosf_host_exact(x[i]); }
void libm_logf(int n, const float *x, float *y) { for (int i = 0; i < n; ++i) y[i] = logf(x[i]); }
void libm_sinf(int n, const float *x, float *y) { for (int i = 0; i < n; ++i) y[i] = sinf(x[i]); }
void libm_cosf(int n, const float *x, float *y) { for (int i = 0; i < n; ++i) y[i] = cosf(x[i]); }
// atanhf / log1pf (the wavelengths of SampledWavelengths::SampleVisible: blackbody emission of temperature grids)
void model_atanhf(int n, const float *x, float *y) { for (int i = 0; i < n; ++i) y[i] = vspg_libm::atanhf_host_exact(x[i]); }
void model_log1pf(int n, const float *x, float *y) { for (int i = 0; i < n; ++i) y[i] = vspg_libm::log1pf_host_exact(x[i]); }
void libm_atanhf(int n, const float *x, float *y) { for (int i = 0; i < n; ++i) y[i] = atanhf(x[i]); }
void libm_log1pf(int n, const float *x, float *y) { for (int i = 0; i < n; ++i) y[i] = log1pf(x[i]); }
// double log (the -std::log(1.0 - x) of the optical-depth-space sampling)
void model_log(int n, const double *x, double *y) { for (int i = 0; i < n; ++i) y[i] = vspg_libm::log_host_exact(x[i]); }
void libm_log(int n, const double *x, double *y) { for (int i = 0; i < n; ++i) y[i] = log(x[i]); }
// powf (the NDS+ collision-probability bias)
void model_powf(int n, const float *x, const float *y, float *o) { for (int i = 0; i < n; ++i) o[i] = vspg_libm::powf_host_exact(x[i], y[i]); }
void libm_powf(int n, const float *x, const float *y, float *o) { for (int i = 0; i < n; ++i) o[i] = powf(x[i], y[i]); }
void libm_neg_log1m(int n, const float *x, float *y) { for (int i = 0; i < n; ++i) y[i] = (float)(-log(1.0 - (double)x[i])); }
}

// vspg_rccl_steps.h -- the entry points of include/vspg_rccl.h above the transport: which collective runs when, on which
// buffers (the sharded PostProcessWave step, the film sum at frame end, the training exchange, the launch check).
// Included once by a translation unit that defines VSPG_TRANSPORT, a struct with
//     static int sum_f32(void *comm, const float *in, float *out, size_t n, hipStream_t s);   // out may be in
//     static int sum_i32(void *comm, int *buf, size_t n, hipStream_t s);
//     static int sum_u64(void *comm, unsigned long long *buf, size_t n, hipStream_t s);
// (device pointers, enqueued on s, 0 = ok): csrc/vspg_rccl.cpp gives it RCCL -- the product -- and tests/rehearse/ a
// shared-memory stand-in that lets two ranks share ONE card (RCCL refuses two ranks on a device), so the stepping logic the
// tests walk is this file, not a copy of it.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdio>
#include <map>
#include <mutex>

#include "../../include/vspg_rccl.h"

namespace {
std::mutex g_mu;
struct Scratch {
    float *p;
    size_t n;  // floats the buffer holds: a renderer re-created at the same address with a larger film gets a new one
};
std::map<VspgRenderer *, Scratch> g_scratch;  // per renderer: the buffer the statistics are summed in
inline void drop_all_scratch() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto &kv : g_scratch) (void)hipFree(kv.second.p);
    g_scratch.clear();
}
#define HCHK(expr)                                                                                      \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) {                                                                         \
            std::fprintf(stderr, "vspg_rccl: %s: %s\n", #expr, hipGetErrorString(e_));                  \
            return VSPG_EHIP;                                                                           \
        }                                                                                               \
    } while (0)
}  // namespace

extern "C" {

int vspg_rccl_forget(VspgRenderer *r) {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_scratch.find(r);
    if (it != g_scratch.end()) {
        (void)hipFree(it->second.p);
        g_scratch.erase(it);
    }
    return 0;
}

int vspg_rccl_post_process_step(VspgRenderer *r, int world, void *comm, void *stream) {
    return vspg_rccl_post_process_step_n(r, world, world, comm, stream);
}

int vspg_rccl_post_process_step_n(VspgRenderer *r, int n_waves, int world, void *comm, void *stream) {
    if (!r || world < 1 || n_waves < 1 || n_waves > world) return VSPG_EINVAL;
    // (every rank evaluates vspg_isg_update_due identically -- same wave counter, same n_waves -- so all enter the collective or none)
    if (world == 1 || !vspg_isg_update_due(r, n_waves)) return vspg_post_process_step(r, n_waves, nullptr, stream);
    float *stats = nullptr;
    size_t n = 0;
    int rc = vspg_flush(r, stream);  // this rank's latest wave enters the statistics on `stream` (no host wait), before the sum reads them
    if (rc) return rc;
    rc = vspg_isg_stats_device_ptr(r, &stats, &n);
    if (rc) return rc;
    float *sum = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_scratch.find(r);
        if (it != g_scratch.end() && it->second.n != n) {  // another renderer lives at this address now
            (void)hipFree(it->second.p);
            g_scratch.erase(it);
            it = g_scratch.end();
        }
        if (it == g_scratch.end()) {
            HCHK(hipMalloc(&sum, n * sizeof(float)));
            g_scratch[r] = Scratch{sum, n};
        } else {
            sum = it->second.p;
        }
    }
    // the renderer's own statistics stay as they are (nothing is counted twice): the sum lands in the scratch buffer
    { const int trc = VSPG_TRANSPORT::sum_f32(comm, stats, sum, n, (hipStream_t)stream); if (trc) return trc; }
    return vspg_post_process_step(r, n_waves, sum, stream);
}

static int rccl_exchange(float *p, size_t n, void *stream, void *user) {
    return VSPG_TRANSPORT::sum_f32(user, p, p, n, (hipStream_t)stream);
}
int vspg_rccl_enable_training_exchange(VspgRenderer *r, void *comm) {
    if (!r) return VSPG_EINVAL;
    return vspg_renderer_set_exchange(r, comm ? rccl_exchange : nullptr, comm);
}

int vspg_rccl_allreduce_film(VspgRenderer *r, void *comm, void *stream) {
    if (!r) return VSPG_EINVAL;
    float *film = nullptr;
    size_t n = 0;
    int rc = vspg_flush(r, stream);  // the frame's last wave enters the film before the sum
    if (rc) return rc;
    rc = vspg_film_device_ptr(r, &film, &n);
    if (rc) return rc;
    return VSPG_TRANSPORT::sum_f32(comm, film, film, n, (hipStream_t)stream);
}

int vspg_rccl_ranks_seen(void *comm, void *stream, int *ranks_seen) {
    if (!comm || !ranks_seen) return VSPG_EINVAL;
    int one = 1, *d = nullptr;
    HCHK(hipMalloc(&d, sizeof(int)));
    HCHK(hipMemcpyAsync(d, &one, sizeof one, hipMemcpyHostToDevice, (hipStream_t)stream));
    const int nr = VSPG_TRANSPORT::sum_i32(comm, d, 1, (hipStream_t)stream);
    hipError_t he = hipMemcpyAsync(&one, d, sizeof one, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d);
    if (nr != 0 || he != hipSuccess) return VSPG_EHIP;
    *ranks_seen = one;
    return 0;
}

int vspg_rccl_sum_counters(VspgRenderer *r, void *comm, void *stream, VspgCounters *out) {
    if (!r || !out) return VSPG_EINVAL;
    VspgCounters c;
    int rc = vspg_get_counters(r, &c, stream);
    if (rc) return rc;
    unsigned long long h[7] = {c.paths, c.segments, c.volume_scatters, c.surface_hits, c.density_queries, c.shadow_rays, c.shadow_density_queries};
    unsigned long long *d = nullptr;
    HCHK(hipMalloc(&d, sizeof h));
    HCHK(hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, (hipStream_t)stream));
    const int nr = VSPG_TRANSPORT::sum_u64(comm, d, 7, (hipStream_t)stream);
    hipError_t he = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d);
    if (nr != 0 || he != hipSuccess) return VSPG_EHIP;
    out->paths = h[0]; out->segments = h[1]; out->volume_scatters = h[2];
    out->surface_hits = h[3]; out->density_queries = h[4]; out->shadow_rays = h[5]; out->shadow_density_queries = h[6];
    return 0;
}

}  // extern "C"

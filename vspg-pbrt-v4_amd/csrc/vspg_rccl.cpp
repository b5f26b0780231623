// vspg_rccl.cpp -- see include/vspg_rccl.h.  Host code only: hipcc -shared -lrccl, links libvspg_hip.so.
#include "../../include/vspg_rccl.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <thread>

namespace {
std::mutex g_mu;
std::map<VspgRenderer *, float *> g_scratch;  // per renderer: the buffer the statistics are summed in
int env_int(const char *n, int def) {
    const char *v = std::getenv(n);
    return v && *v ? std::atoi(v) : def;
}
#define RCHK(expr)                                                                                      \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) {                                                                        \
            std::fprintf(stderr, "vspg_rccl: %s: %s\n", #expr, ncclGetErrorString(r_));                 \
            return VSPG_EHIP;                                                                           \
        }                                                                                               \
    } while (0)
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

int vspg_rccl_init_from_env(const char *id_file, int *rank, int *world, int *local_rank, void **comm) {
    if (!rank || !world || !local_rank || !comm) return VSPG_EINVAL;
    *rank = env_int("RANK", 0);
    *world = env_int("WORLD_SIZE", 1);
    *local_rank = env_int("LOCAL_RANK", *rank);
    *comm = nullptr;
    if (*world < 1 || *rank < 0 || *rank >= *world) return VSPG_EINVAL;
    HCHK(hipSetDevice(*local_rank));
    ncclUniqueId id;
    std::memset(&id, 0, sizeof id);
    if (*world == 1) {
        RCHK(ncclGetUniqueId(&id));
    } else {
        std::string path = id_file ? id_file : "";
        if (path.empty()) {
            const char *e = std::getenv("VSPG_RCCL_ID_FILE");
            path = e && *e ? e : std::string("/tmp/vspg_rccl_id.") + std::to_string(env_int("MASTER_PORT", 29500));
        }
        if (*rank == 0) {
            RCHK(ncclGetUniqueId(&id));
            const std::string tmp = path + ".tmp";
            std::FILE *f = std::fopen(tmp.c_str(), "wb");
            if (!f || std::fwrite(&id, sizeof id, 1, f) != 1) { if (f) std::fclose(f); return VSPG_EINVAL; }
            std::fclose(f);
            if (std::rename(tmp.c_str(), path.c_str()) != 0) return VSPG_EINVAL;  // atomic publish
        } else {
            bool ok = false;
            for (int tries = 0; tries < 600 && !ok; ++tries) {
                std::FILE *f = std::fopen(path.c_str(), "rb");
                if (f) {
                    ok = std::fread(&id, sizeof id, 1, f) == 1;
                    std::fclose(f);
                }
                if (!ok) std::this_thread::sleep_for(std::chrono::milliseconds(100));
            }
            if (!ok) { std::fprintf(stderr, "vspg_rccl: no unique id in %s after 60 s\n", path.c_str()); return VSPG_EINVAL; }
        }
    }
    ncclComm_t c;
    RCHK(ncclCommInitRank(&c, *world, id, *rank));
    *comm = c;
    return 0;
}

int vspg_rccl_destroy(void *comm) {
    {
        std::lock_guard<std::mutex> lk(g_mu);
        for (auto &kv : g_scratch) (void)hipFree(kv.second);
        g_scratch.clear();
    }
    if (comm) RCHK(ncclCommDestroy((ncclComm_t)comm));
    return 0;
}

int vspg_rccl_post_process_step(VspgRenderer *r, int world, void *comm, void *stream) {
    if (!r || world < 1) return VSPG_EINVAL;
    if (world == 1 || !vspg_isg_update_due(r, world)) return vspg_post_process_step(r, world, nullptr, stream);
    float *stats = nullptr;
    size_t n = 0;
    int rc = vspg_isg_stats_device_ptr(r, &stats, &n);
    if (rc) return rc;
    float *sum = nullptr;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        auto it = g_scratch.find(r);
        if (it == g_scratch.end()) {
            HCHK(hipMalloc(&sum, n * sizeof(float)));
            g_scratch[r] = sum;
        } else {
            sum = it->second;
        }
    }
    // the renderer's own statistics stay as they are (nothing is counted twice): the sum lands in the scratch buffer
    RCHK(ncclAllReduce(stats, sum, n, ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
    return vspg_post_process_step(r, world, sum, stream);
}

static int rccl_exchange(float *p, size_t n, void *stream, void *user) {
    const ncclResult_t nr = ncclAllReduce(p, p, n, ncclFloat, ncclSum, (ncclComm_t)user, (hipStream_t)stream);
    if (nr != ncclSuccess) {
        std::fprintf(stderr, "vspg_rccl: training exchange: %s\n", ncclGetErrorString(nr));
        return VSPG_EHIP;
    }
    return 0;
}
int vspg_rccl_enable_training_exchange(VspgRenderer *r, void *comm) {
    if (!r) return VSPG_EINVAL;
    return vspg_renderer_set_exchange(r, comm ? rccl_exchange : nullptr, comm);
}

int vspg_rccl_allreduce_film(VspgRenderer *r, void *comm, void *stream) {
    if (!r) return VSPG_EINVAL;
    float *film = nullptr;
    size_t n = 0;
    int rc = vspg_film_device_ptr(r, &film, &n);
    if (rc) return rc;
    RCHK(ncclAllReduce(film, film, n, ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream));
    return 0;
}

int vspg_rccl_sum_counters(VspgRenderer *r, void *comm, void *stream, VspgCounters *out) {
    if (!r || !out) return VSPG_EINVAL;
    VspgCounters c;
    int rc = vspg_get_counters(r, &c, stream);
    if (rc) return rc;
    unsigned long long h[6] = {c.paths, c.segments, c.volume_scatters, c.surface_hits, c.density_queries, c.shadow_rays};
    unsigned long long *d = nullptr;
    HCHK(hipMalloc(&d, sizeof h));
    HCHK(hipMemcpyAsync(d, h, sizeof h, hipMemcpyHostToDevice, (hipStream_t)stream));
    ncclResult_t nr = ncclAllReduce(d, d, 6, ncclUint64, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    hipError_t he = hipMemcpyAsync(h, d, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (he == hipSuccess) he = hipStreamSynchronize((hipStream_t)stream);
    (void)hipFree(d);
    if (nr != ncclSuccess || he != hipSuccess) return VSPG_EHIP;
    out->paths = h[0]; out->segments = h[1]; out->volume_scatters = h[2];
    out->surface_hits = h[3]; out->density_queries = h[4]; out->shadow_rays = h[5];
    return 0;
}

}  // extern "C"

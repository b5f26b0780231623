// vspg_rccl.cpp -- see include/vspg_rccl.h.  Host code only: hipcc -shared -lrccl, links libvspg_hip.so.
#include "../../include/vspg_rccl.h"
#include "vspg_rendezvous.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

namespace {
using vspg_rendezvous::env_int;
#define RCHK(expr)                                                                                      \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) {                                                                        \
            std::fprintf(stderr, "vspg_rccl: %s: %s\n", #expr, ncclGetErrorString(r_));                 \
            return VSPG_EHIP;                                                                           \
        }                                                                                               \
    } while (0)
struct RcclTransport {
    static int chk(ncclResult_t r, const char *what) {
        if (r == ncclSuccess) return 0;
        std::fprintf(stderr, "vspg_rccl: %s: %s\n", what, ncclGetErrorString(r));
        return VSPG_EHIP;
    }
    static int sum_f32(void *comm, const float *in, float *out, size_t n, hipStream_t s) {
        return chk(ncclAllReduce(in, out, n, ncclFloat, ncclSum, (ncclComm_t)comm, s), "all-reduce (float)");
    }
    static int sum_i32(void *comm, int *buf, size_t n, hipStream_t s) {
        return chk(ncclAllReduce(buf, buf, n, ncclInt32, ncclSum, (ncclComm_t)comm, s), "all-reduce (int32)");
    }
    static int sum_u64(void *comm, unsigned long long *buf, size_t n, hipStream_t s) {
        return chk(ncclAllReduce(buf, buf, n, ncclUint64, ncclSum, (ncclComm_t)comm, s), "all-reduce (uint64)");
    }
};
}  // namespace
#define VSPG_TRANSPORT RcclTransport
#include "vspg_rccl_steps.h"

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
    std::string published;
    if (*world == 1) {
        RCHK(ncclGetUniqueId(&id));
    } else {
        static_assert(sizeof(ncclUniqueId) <= vspg_rendezvous::kBlobBytes, "rendezvous record too small for the id");
        published = vspg_rendezvous::default_path(id_file);
        if (*rank == 0) {
            RCHK(ncclGetUniqueId(&id));
            if (!vspg_rendezvous::publish(published, *world, &id, sizeof id)) return VSPG_EINVAL;
        } else if (!vspg_rendezvous::fetch(published, *world, &id, sizeof id)) {
            std::fprintf(stderr, "vspg_rccl: no unique id of this run in %s after 60 s\n", published.c_str());
            return VSPG_EINVAL;
        }
    }
    ncclComm_t c;
    const ncclResult_t init = ncclCommInitRank(&c, *world, id, *rank);
    // ncclCommInitRank returns once every rank has joined, i.e. has read the record: rank 0 takes it away again, so the next
    // run on this port (and this launcher) starts from an empty name whatever happens to this one later
    if (*rank == 0 && !published.empty()) vspg_rendezvous::retire(published);
    RCHK(init);
    *comm = c;
    return 0;
}

int vspg_rccl_destroy(void *comm) {
    drop_all_scratch();
    if (comm) RCHK(ncclCommDestroy((ncclComm_t)comm));
    return 0;
}

}  // extern "C"

// vspg_rccl_rehearse.cpp -- TEST INFRASTRUCTURE, not part of the product: include/vspg_rccl.h over a host shared-memory
// transport, so that the all-C++ multi-GPU host (host/vspg_pbrt_sharded) can be walked with TWO ranks on ONE card.
// RCCL refuses two ranks on one device ("Duplicate GPU detected"), and this builder's GPU boxes have one; the driver's 8-GPU
// node runs the product library (csrc/libvspg_rccl.so).  What is shared with the product, and therefore under test here:
//   * the rendezvous protocol (csrc/vspg_rendezvous.h: nonce'd record, stale-file removal, retire after the collective join),
//   * every entry point above the transport (csrc/vspg_rccl_steps.h: sharded PostProcessWave step incl. the ragged last step,
//     film sum, training exchange, counters, launch check).
// What is NOT: the RCCL calls themselves (three ncclAllReduce forms in csrc/vspg_rccl.cpp).
// A collective here = stream sync, device -> host, every rank writes its slot of a POSIX shared-memory segment, barrier, every
// rank sums the slots in rank order (so all ranks hold the same bits, like an all-reduce), host -> device.
// Built by tests/test_host_adapter.py into tests/_build/; the binary linked against it is vspg_pbrt_sharded_rehearse.
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "../../vspg-pbrt-v4_amd/csrc/vspg_rendezvous.h"

namespace {
struct ShmHeader {
    std::atomic<uint32_t> arrive, gen;
};
struct RehearseComm {
    int rank, world;
    std::string base;  // shared-memory name stem: /vspg_rehearse.<token>
    ShmHeader *hdr;
    unsigned long long seq;
};
bool barrier(RehearseComm *c) {
    const uint32_t g = c->hdr->gen.load();
    if (c->hdr->arrive.fetch_add(1) + 1 == (uint32_t)c->world) {
        c->hdr->arrive.store(0);
        c->hdr->gen.fetch_add(1);
        return true;
    }
    for (int i = 0; i < 1200000; ++i) {  // 60 s
        if (c->hdr->gen.load() != g) return true;
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    std::fprintf(stderr, "vspg_rccl_rehearse: rank %d: barrier timed out\n", c->rank);
    return false;
}
void *map_segment(const std::string &name, size_t bytes, bool create) {
    int fd = -1;
    for (int tries = 0; tries < 6000 && fd < 0; ++tries) {
        fd = shm_open(name.c_str(), create ? (O_CREAT | O_RDWR | O_EXCL) : O_RDWR, 0600);
        if (fd < 0) {
            if (create) return nullptr;
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    if (fd < 0) return nullptr;
    if (create && ftruncate(fd, (off_t)bytes) != 0) { close(fd); return nullptr; }
    if (!create) {  // the creator sizes it before anyone is told the name is valid, but be safe
        struct stat st;
        for (int tries = 0; tries < 6000; ++tries) {
            if (fstat(fd, &st) == 0 && (size_t)st.st_size >= bytes) break;
            std::this_thread::sleep_for(std::chrono::milliseconds(10));
        }
    }
    void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    return p == MAP_FAILED ? nullptr : p;
}
template <class T>
int host_allreduce(void *comm, const T *in, T *out, size_t n, hipStream_t s);
struct ShmTransport {
    static int sum_f32(void *comm, const float *in, float *out, size_t n, hipStream_t s) { return host_allreduce<float>(comm, in, out, n, s); }
    static int sum_i32(void *comm, int *buf, size_t n, hipStream_t s) { return host_allreduce<int>(comm, buf, buf, n, s); }
    static int sum_u64(void *comm, unsigned long long *buf, size_t n, hipStream_t s) { return host_allreduce<unsigned long long>(comm, buf, buf, n, s); }
};
}  // namespace
#define VSPG_TRANSPORT ShmTransport
#include "../../vspg-pbrt-v4_amd/csrc/vspg_rccl_steps.h"

namespace {
template <class T>
int host_allreduce(void *comm, const T *in, T *out, size_t n, hipStream_t s) {
    RehearseComm *c = (RehearseComm *)comm;
    if (!c) return VSPG_EINVAL;
    std::vector<T> h(n);
    HCHK(hipMemcpyAsync(h.data(), in, n * sizeof(T), hipMemcpyDeviceToHost, s));
    HCHK(hipStreamSynchronize(s));
    if (c->world > 1) {
        const std::string name = c->base + ".d" + std::to_string(c->seq++);
        const size_t bytes = (size_t)c->world * n * sizeof(T);
        T *seg = nullptr;
        if (c->rank == 0) {
            seg = (T *)map_segment(name, bytes, true);
            if (!seg) return VSPG_EINVAL;
        }
        if (!barrier(c)) return VSPG_EHIP;  // the segment exists and has its size
        if (c->rank != 0) {
            seg = (T *)map_segment(name, bytes, false);
            if (!seg) return VSPG_EINVAL;
        }
        std::memcpy(seg + (size_t)c->rank * n, h.data(), n * sizeof(T));
        if (!barrier(c)) return VSPG_EHIP;  // every slot is written
        for (size_t i = 0; i < n; ++i) {
            T acc = seg[i];
            for (int r = 1; r < c->world; ++r) acc += seg[(size_t)r * n + i];
            h[i] = acc;
        }
        if (!barrier(c)) return VSPG_EHIP;  // every rank has read
        munmap(seg, bytes);
        if (c->rank == 0) shm_unlink(name.c_str());
    }
    HCHK(hipMemcpyAsync(out, h.data(), n * sizeof(T), hipMemcpyHostToDevice, s));
    HCHK(hipStreamSynchronize(s));
    return 0;
}
}  // namespace

extern "C" {

int vspg_rccl_init_from_env(const char *id_file, int *rank, int *world, int *local_rank, void **comm) {
    using vspg_rendezvous::env_int;
    if (!rank || !world || !local_rank || !comm) return VSPG_EINVAL;
    *rank = env_int("RANK", 0);
    *world = env_int("WORLD_SIZE", 1);
    *local_rank = env_int("LOCAL_RANK", *rank);
    *comm = nullptr;
    if (*world < 1 || *rank < 0 || *rank >= *world) return VSPG_EINVAL;
    HCHK(hipSetDevice(*local_rank));
    RehearseComm *c = new RehearseComm{*rank, *world, "", nullptr, 0};
    char token[vspg_rendezvous::kBlobBytes];
    std::memset(token, 0, sizeof token);
    std::string published;
    if (*world > 1) {
        published = vspg_rendezvous::default_path(id_file);
        if (*rank == 0) {
            unsigned long long rnd = 0;
            if (FILE *f = std::fopen("/dev/urandom", "rb")) { (void)!std::fread(&rnd, sizeof rnd, 1, f); std::fclose(f); }
            std::snprintf(token, sizeof token, "/vspg_rehearse.%d.%llx", (int)getpid(), rnd);
            c->base = token;
            c->hdr = (ShmHeader *)map_segment(c->base + ".hdr", sizeof(ShmHeader), true);  // zero-filled by ftruncate
            if (!c->hdr || !vspg_rendezvous::publish(published, *world, token, sizeof token)) return VSPG_EINVAL;
        } else {
            if (!vspg_rendezvous::fetch(published, *world, token, sizeof token)) {
                std::fprintf(stderr, "vspg_rccl_rehearse: no record of this run in %s after 60 s\n", published.c_str());
                return VSPG_EINVAL;
            }
            token[sizeof token - 1] = 0;
            c->base = token;
            c->hdr = (ShmHeader *)map_segment(c->base + ".hdr", sizeof(ShmHeader), false);
            if (!c->hdr) return VSPG_EINVAL;
        }
        const bool joined = barrier(c);  // the collective join (ncclCommInitRank in the product)
        if (*rank == 0) vspg_rendezvous::retire(published);
        if (!joined) return VSPG_EHIP;
    }
    *comm = c;
    return 0;
}

int vspg_rccl_destroy(void *comm) {
    drop_all_scratch();
    RehearseComm *c = (RehearseComm *)comm;
    if (c) {
        if (c->hdr) {
            if (c->world > 1) (void)barrier(c);
            munmap(c->hdr, sizeof(ShmHeader));
            if (c->rank == 0) shm_unlink((c->base + ".hdr").c_str());
        }
        delete c;
    }
    return 0;
}

}  // extern "C"

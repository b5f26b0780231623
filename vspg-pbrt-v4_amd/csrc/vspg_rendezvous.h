// vspg_rendezvous.h -- how the ranks of one launch on one node agree on a communicator id without a launcher-side store:
// rank 0 publishes a small record in a file, the others wait for it (include/vspg_rccl.h, vspg_rccl_init_from_env).
// Header-only host code; used by csrc/vspg_rccl.cpp (the blob is an ncclUniqueId) and by the one-card rehearsal transport of
// the tests (tests/rehearse/, the blob names a shared-memory segment), so that the protocol under test is the product's.
//
// What makes a record THIS run's (round-2 review: a stale file of an earlier run on the same port made ranks >= 1 join a dead
// root): the default file name carries the run's nonce, the record carries its hash and the rank count, a reader ignores
// anything else, and rank 0 removes stale files before publishing and its own file after the collective join.
#pragma once
#include <unistd.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>

namespace vspg_rendezvous {

constexpr size_t kBlobBytes = 128;  // sizeof(ncclUniqueId)
struct Record {
    char magic[8];  // "VSPGID2\0"
    int32_t world;
    int32_t pad;
    uint64_t nonce_hash;  // FNV-1a of the run's nonce string
    unsigned char blob[kBlobBytes];
};
inline uint64_t fnv1a(const std::string &s) {
    uint64_t h = 1469598103934665603ull;
    for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
    return h;
}
inline int env_int(const char *n, int def) {
    const char *v = std::getenv(n);
    return v && *v ? std::atoi(v) : def;
}
// The launcher's run id when it hands one out (any launcher: VSPG_RCCL_NONCE; torch.distributed.run: TORCHELASTIC_RUN_ID),
// else the launcher's PID: the ranks of one launch are children of one process (torchrun, mpirun, a shell loop).
inline std::string run_nonce() {
    const char *e = std::getenv("VSPG_RCCL_NONCE");
    if (e && *e) return e;
    e = std::getenv("TORCHELASTIC_RUN_ID");
    if (e && *e && std::strcmp(e, "none") != 0) return e;
    return std::string("ppid") + std::to_string((long)getppid());
}
inline std::string default_path(const char *id_file) {
    if (id_file && *id_file) return id_file;
    const char *e = std::getenv("VSPG_RCCL_ID_FILE");
    if (e && *e) return e;
    return std::string("/tmp/vspg_rccl_id.") + std::to_string(env_int("MASTER_PORT", 29500)) + "." + run_nonce();
}
// rank 0: a stale file goes first, then the record appears atomically (written under a private name, renamed)
inline bool publish(const std::string &path, int world, const void *blob, size_t n) {
    if (n > kBlobBytes) return false;
    Record rec;
    std::memset(&rec, 0, sizeof rec);
    std::memcpy(rec.magic, "VSPGID2", 8);
    rec.world = world;
    rec.nonce_hash = fnv1a(run_nonce());
    std::memcpy(rec.blob, blob, n);
    (void)std::remove(path.c_str());
    const std::string tmp = path + ".tmp." + std::to_string((long)getpid());
    std::FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f || std::fwrite(&rec, sizeof rec, 1, f) != 1) {
        if (f) std::fclose(f);
        return false;
    }
    std::fclose(f);
    return std::rename(tmp.c_str(), path.c_str()) == 0;
}
// ranks >= 1: wait (at most timeout_s) for a record of this run -- right magic, rank count and nonce; anything else is ignored
inline bool fetch(const std::string &path, int world, void *blob, size_t n, int timeout_s = 60) {
    const uint64_t want = fnv1a(run_nonce());
    Record rec;
    for (int tries = 0; tries < timeout_s * 10; ++tries) {
        std::FILE *f = std::fopen(path.c_str(), "rb");
        if (f) {
            const bool ok = std::fread(&rec, sizeof rec, 1, f) == 1 && std::memcmp(rec.magic, "VSPGID2", 8) == 0 && rec.world == world &&
                            rec.nonce_hash == want;
            std::fclose(f);
            if (ok) {
                std::memcpy(blob, rec.blob, n < kBlobBytes ? n : kBlobBytes);
                return true;
            }
        }
        std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
    return false;
}
// rank 0, once the collective join has returned (every rank has read the record by then)
inline void retire(const std::string &path) { (void)std::remove(path.c_str()); }

}  // namespace vspg_rendezvous

/* vspg_rccl.h -- the multi-GPU step of the GuidedVolPathVSPG path in C (north star: "host code stays C++",
 * "RCCL all-reduce over xGMI of the float film/weight tiles at frame end"), exported by csrc/libvspg_rccl.so.
 *
 * One process per GPU.  Every rank renders its own sample indices of the same frame through include/vspg.h
 * (VspgRenderConfig.shard_index / shard_count); these entry points add the two collectives of the path:
 *   - the image-space VSP statistics, summed over the ranks on the steps where the buffer updates
 *     (PostProcessWave, guidedvolpathvspgintegrator.cpp:230-260, with waveCounter advancing by the rank count), and
 *   - the float film {sum w*rgb, sum w} (RGBFilm accumulate contract, film.h:251-267) at frame end.
 * `comm` is an ncclComm_t (RCCL), `stream` a hipStream_t; both collectives are enqueued on `stream`.
 * The reference has no counterpart (it is a single-process CPU renderer): this is what a multi-GPU pbrt host links next to
 * libvspg_hip.so.  Python (bench.py, vspg-pbrt-v4_amd/sharding.py) runs the same two collectives through torch.distributed. */
#ifndef VSPG_RCCL_H
#define VSPG_RCCL_H
#include "vspg.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Communicator for the ranks of one node from the launcher's environment (RANK, WORLD_SIZE, LOCAL_RANK as set by
 * torch.distributed.run / mpirun wrappers): rank 0 creates the ncclUniqueId and publishes it in the file `id_file`
 * (NULL: $VSPG_RCCL_ID_FILE, else /tmp/vspg_rccl_id.<MASTER_PORT or 29500>.<nonce>), the others wait for it (at most 60 s).
 * The record carries the rank count and a hash of the run's nonce -- $VSPG_RCCL_NONCE, else $TORCHELASTIC_RUN_ID, else the
 * launcher's PID (getppid(): the ranks of one launch are children of one process) -- and a reader ignores any record that is
 * not this run's.  Rank 0 removes a stale file before publishing and removes its own once ncclCommInitRank has returned
 * (every rank has read it by then): two runs back to back on one port do not see each other's id.
 * hipSetDevice(LOCAL_RANK) is called.  *comm receives the ncclComm_t.  world == 1 needs no file. */
int vspg_rccl_init_from_env(const char *id_file, int *rank, int *world, int *local_rank, void **comm);
int vspg_rccl_destroy(void *comm);

/* PostProcessWave of a sharded step: vspg_post_process_step(r, world, sum, stream) where `sum` is the all-reduced
 * copy of the ranks' VSP statistics when vspg_isg_update_due(r, world), NULL otherwise. */
int vspg_rccl_post_process_step(VspgRenderer *r, int world, void *comm, void *stream);
/* The same for a step that covers n_waves <= world sample indices (the last step of a frame whose sample count is not a
 * multiple of the rank count: ranks n_waves.. rendered nothing, the wave counter advances by n_waves). */
int vspg_rccl_post_process_step_n(VspgRenderer *r, int n_waves, int world, void *comm, void *stream);
/* Drop what this library keeps per renderer (the device buffer the statistics are summed in).  Call before
 * vspg_renderer_destroy when the communicator outlives the renderer; vspg_rccl_destroy drops everything. */
int vspg_rccl_forget(VspgRenderer *r);
/* all-reduce of a one per rank: *ranks_seen == world iff every rank of the communicator took part (a launch check) */
int vspg_rccl_ranks_seen(void *comm, void *stream, int *ranks_seen);
/* frame end: in-place sum of the film over the ranks */
int vspg_rccl_allreduce_film(VspgRenderer *r, void *comm, void *stream);
/* guiding-field training over all ranks' samples: installs vspg_renderer_set_exchange(r, <ncclAllReduce sum on comm>), so that
 * every rank's Field::Update fits the same field (include/vspg.h).  Call once after vspg_renderer_create, on every rank. */
int vspg_rccl_enable_training_exchange(VspgRenderer *r, void *comm);
/* sum of the path counters over the ranks (host values) */
int vspg_rccl_sum_counters(VspgRenderer *r, void *comm, void *stream, VspgCounters *out);

#ifdef __cplusplus
}
#endif
#endif

// vspg_capi.hip -- HIP kernels (gfx950) + the C-ABI declared in include/vspg.h.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared (see csrc/Makefile).
// No CPU fallback lives here: every compute entry point needs a HIP device.
#include <hip/hip_runtime.h>
#include <dlfcn.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>
#include <algorithm>

#include "vspg_path.h"
#include "vspg_wg_kernel.h"
#include "vspg_guided_wg.h"
#include "vspg_wg3.h"
#include "vspg_trace.h"
#include "vspg_wavefront.h"
#include "vspg_wf_launch.h"
#ifdef VSPG_SINGLE_TU  // diagnostic builds that read device-side globals of the pipeline kernels (VSPG_WF_STATS, VSPG_PROFILE, VSPG_WF_DEBUG)
#include "vspg_wf_grid.hip"
#include "vspg_wf_nvdb.hip"
#endif

using namespace vspg;

// =======================================================================================
// kernels
// =======================================================================================
namespace {

constexpr int kBlock = 256;
static_assert(kBlock == kGuideBlock, "the guiding scratch in LDS is sized for kBlock threads");
#ifndef VSPG_WAVES_PER_SIMD
#define VSPG_WAVES_PER_SIMD 2
#endif
constexpr int kWavesPerSimd = VSPG_WAVES_PER_SIMD;  // register budget of k_render_wave: 512 / kWavesPerSimd VGPRs
constexpr int kBlocksPerCU = kWavesPerSimd;         // persistent 256-thread blocks per CU == resident blocks
constexpr int kChunk = 64;        // dynamic work items a wavefront claims per atomic (one 8x8 pixel tile)
constexpr int kNumCounters = 7;  // paths, segments, volume_scatters, surface_hits, density_queries, shadow_rays, shadow_density_queries

template <class PC>
__device__ __forceinline__ void flush_counters(const PC &pc, uint32_t paths, unsigned long long *g) {
    __shared__ unsigned int s[kNumCounters];
    if (threadIdx.x < kNumCounters) s[threadIdx.x] = 0;
    __syncthreads();
    atomicAdd(&s[0], paths);
    atomicAdd(&s[1], pc.segments);
    atomicAdd(&s[2], pc.volume_scatters);
    atomicAdd(&s[3], pc.surface_hits);
    atomicAdd(&s[4], pc.density_queries);
    atomicAdd(&s[5], pc.shadow_rays);
    atomicAdd(&s[6], pc.shadow_queries);
    __syncthreads();
    if (threadIdx.x < kNumCounters) atomicAdd(&g[threadIdx.x], (unsigned long long)s[threadIdx.x]);
}

// (reset_sibling_head -- the pair of global work heads used by alternate launches -- lives in vspg_wg3.h)
// Persistent wavefront kernel with path regeneration.
//   work item  = one pixel (all its samples [wave_start, wave_end) run in order by the lane that
//                claims it, so film / ISG statistics are read-modify-written by exactly one lane
//                per launch: no atomics, deterministic summation order);
//   claiming   = lanes whose path has ended are compacted with a wave ballot + prefix count and
//                served by ONE atomicAdd per wavefront on the global work head (stochastic path
//                termination -- Russian roulette, absorption at maxdepth -- leaves holes in the
//                wave; regeneration refills them instead of idling until the longest path ends);
//   path state = registers for the whole life of a path (no HBM round trip per segment);
//   items are tile-ordered (8x8 pixels per 64 items) so a fresh wavefront starts on one coherent
//   tile of primary rays.
template <class Medium, bool GUIDED, bool TRAIN = false>
__global__ __launch_bounds__(kBlock, kWavesPerSimd) void k_render_wave(const DScene *__restrict__ Sp, float4 *__restrict__ film,
                                                        float *__restrict__ isg_stats, const float *__restrict__ vsp_buf,
                                                        int vsp_ready, int wave_start, int wave_end,
                                                        int first_sample, int single_sample, PcgJump jump,
                                                        unsigned static_per_wave, unsigned dyn_base,
                                                        unsigned int *__restrict__ work_head,
                                                        unsigned long long *__restrict__ counters, TrainArgs train) {
    // first_sample: first sample index of this shard in [wave_start, wave_end); single_sample: the
    // launch covers exactly one sample per pixel (the reference's 1-spp waves) and `jump` is the
    // PCG skip-ahead for first_sample*65536
    const DScene &S = *Sp;
    const int W = S.xres, H = S.yres;
    const int tilesX = (W + 7) >> 3, tilesY = (H + 7) >> 3;
    const unsigned total_items = (unsigned)(tilesX * tilesY) * 64u;
    const int lane = threadIdx.x & 63;
    reset_sibling_head(work_head);
    // heterogeneous media: the 16^3 majorant grid (16 KB) is staged into LDS once per workgroup with
    // coalesced 16-B loads; every DDA step then reads LDS instead of HBM/L2.  (Guided builds leave it in
    // global memory -- it stays in the vector L1: their 72 KB of guiding scratch plus the grid would allow
    // one block per CU only.)
    const float *maj_ptr = nullptr;
    if constexpr (!GUIDED && (std::is_same<Medium, GridMedium>::value || std::is_same<Medium, GridMediumGrey>::value)) {
        __shared__ float s_maj[kMajRes * kMajRes * kMajRes];
        const float4 *src = reinterpret_cast<const float4 *>(S.majorant);
        float4 *dst = reinterpret_cast<float4 *>(s_maj);
        for (int i = threadIdx.x; i < kMajRes * kMajRes * kMajRes / 4; i += kBlock) dst[i] = src[i];
        __syncthreads();
        maj_ptr = s_maj;
    }
    stage_scene_lds(S);
    __syncthreads();
    const Medium medium = MediumMaker<Medium>::make(S, maj_ptr);
    // guided builds: the per-lane guiding scratch (vspg_guiding.h: 9 floats x 8 lobes) lives in LDS,
    // element e of lane t at s_gmix[e * kBlock + t] (conflict-free)
    float *glds = nullptr;
    if constexpr (GUIDED) glds = guide_lds();
    typename std::conditional<TRAIN, PathCountersT<PathRecorder>, PathCounters>::type pc;
    pc.segments = pc.volume_scatters = pc.surface_hits = pc.density_queries = pc.shadow_rays = pc.shadow_queries = 0;
    if constexpr (TRAIN) {
        pc.rec.base = train.segbuf;  // re-pointed at the lane's work item when a path starts
        pc.rec.stride = (int)train.n_items;
        pc.rec.max_seg = train_rec_capacity(S.prm.maxdepth);
        pc.rec.reset();
    }
    uint32_t paths = 0;

    bool has = false;        // this lane carries a live path
    bool exhausted = false;  // wave-uniform: the global work head ran past the last item
    // wave-local slice [local_next, local_end) of the item space.  Every wavefront starts with a
    // STATIC slice (static_per_wave items, ~3/4 of its fair share, no atomics at all); the remaining
    // items [dyn_base, total) are handed out dynamically in chunks of kChunk through one returning
    // atomic per chunk, which evens out the tail.  (One atomic per refill on a single hot counter
    // saturates at ~10^2 dequeues/us on this chip and capped the kernel at the atomic rate.)
    const unsigned wave_id = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    unsigned local_next = wave_id * static_per_wave, local_end = local_next + static_per_wave;
    int px = 0, py = 0, s = 0, ch = 0;
    Sampler sampler;
    PathState st;
    IsgSample isg;

    // Training launches (one sample per pixel): every path writes its segment records into its OWN column of a
    // buffer sized for the whole wave (1.3 GB at 1080p -- this part has 288 GB) and leaves; PropagateSamples runs
    // afterwards as its own kernel over all paths at full occupancy (k_propagate).  Propagating inside this kernel
    // meant a lock-step loop over the longest recorded path, ~100 dependent loads, for the few lanes that had just
    // finished (8.0 ms per training wave), or parking finished lanes until enough had gathered (4.4 ms).
    unsigned my_item = 0;
    while (true) {
        // ---- regeneration: ballot the empty lanes, prefix-count them, hand out items -----------
        unsigned long long need = __ballot(!has);
        if (need != 0ull && !(exhausted && local_next >= local_end)) {
            if (local_next >= local_end && !exhausted) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(work_head, (unsigned)kChunk);
                base = __builtin_amdgcn_readfirstlane(base) + dyn_base;
                local_next = base;
                local_end = base + (unsigned)kChunk < total_items ? base + (unsigned)kChunk : total_items;
                if (base + (unsigned)kChunk >= total_items) exhausted = true;
                if (base >= total_items) local_next = local_end = 0;
            }
            const unsigned rank = (unsigned)__popcll(need & ((1ull << lane) - 1ull));
            const unsigned avail = local_end - local_next;  // lanes beyond `avail` wait for the next chunk
            if (!has && rank < avail) {
                VSPG_PROF(PS_START);
                const unsigned item = local_next + rank;
                const unsigned tile = item >> 6, l = item & 63u;
                px = (int)(tile % (unsigned)tilesX) * 8 + (int)(l & 7u);
                py = (int)(tile / (unsigned)tilesX) * 8 + (int)(l >> 3);
                s = first_sample;
                if (px < W && py < H && s < wave_end) {
                    if (single_sample)
                        start_path(S, vsp_buf, vsp_ready, px, py, jump, sampler, st, &ch, isg);
                    else
                        start_path(S, vsp_buf, vsp_ready, px, py, s, sampler, st, &ch, isg);
                    has = true;
                    if constexpr (GUIDED) load_contribution_estimate(S, px, py, st);
                    if constexpr (TRAIN) {
                        my_item = item;
                        pc.rec.base = train.segbuf + item;
                        pc.rec.reset();
                    }
                }
            }
            const unsigned cnt = (unsigned)__popcll(need);
            local_next += cnt < avail ? cnt : avail;
        }
        if (__ballot(has) == 0ull) {
            if (exhausted && local_next >= local_end) break;
            continue;
        }
        // ---- one path segment for every live lane ----------------------------------------------
        bool finished = false;
        if (has) {
            const bool alive = li_segment<Medium, GUIDED>(S, medium, vsp_buf, vsp_ready, px, py, st, ch, sampler, isg, pc, glds, kBlock);
            if (!alive) {
                VSPG_PROF(PS_FINISH);
                const Spec L = finish_radiance(st.L);
                const size_t idx = (size_t)py * W + px;
                film_add_sample(film + idx, L);
                isg_add_sample_atomic(isg_stats + idx * VSPG_ISG_STATS, L, isg);
                paths++;
                finished = true;
                if constexpr (TRAIN) train.seg_count[my_item] = pc.rec.n;  // PropagateSamples (:627) follows in k_propagate
            }
        }
        if (finished) {
            s += S.shard_count > 1 ? S.shard_count : 1;
            if (!TRAIN && s < wave_end) {
                start_path(S, vsp_buf, vsp_ready, px, py, s, sampler, st, &ch, isg);  // next sample of the same pixel
                if constexpr (GUIDED) load_contribution_estimate(S, px, py, st);
            } else {
                has = false;
            }
        }
    }
    flush_counters(pc, paths, counters);
}

// PathSegmentStorage::PropagateSamples (:627) for every path of a training wave: one thread per work item
constexpr int kStagePerLane = 3;  // samples staged per lane on average (2.5 at the defaults); a wavefront that collects more flushes early (StageSink)
__global__ __launch_bounds__(kBlock) void k_propagate(TrainArgs train, int max_seg) {
    __shared__ VspgTrainSample s_stage[kBlock * kStagePerLane];  // 31 KB: five workgroups per CU
    __shared__ unsigned int s_wcount[kBlock / 64], s_wzero[kBlock / 64];
    __shared__ unsigned long long s_base;
    const unsigned i = blockIdx.x * kBlock + threadIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    PathRecorder rec;
    rec.base = train.segbuf + (i < train.n_items ? i : 0u);
    rec.stride = (int)train.n_items;
    rec.max_seg = max_seg;
    rec.reset();
    rec.n = i < train.n_items ? train.seg_count[i] : 0;
    StageSink sink{s_stage + wave * 64 * kStagePerLane, 64u * kStagePerLane, 0u, train.samples, train.counters, train.capacity};
    unsigned int zero = propagate_samples(rec, rec.n > 0, sink);
    for (int off = 32; off > 0; off >>= 1) zero += __shfl_xor(zero, off);  // (one atomic per workgroup: the counters are single words)
    if (lane == 0) { s_wcount[wave] = sink.count; s_wzero[wave] = zero; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int total = 0, zeros = 0;
        for (int w = 0; w < kBlock / 64; ++w) { total += s_wcount[w]; zeros += s_wzero[w]; }
        s_base = total ? atomicAdd(&train.counters[0], (unsigned long long)total) : 0ull;
        if (zeros) atomicAdd(&train.counters[1], (unsigned long long)zeros);
    }
    __syncthreads();
    unsigned long long base = s_base;
    for (int w = 0; w < wave; ++w) base += s_wcount[w];
    for (unsigned int j = (unsigned int)lane; j < sink.count; j += 64u)
        if (base + j < train.capacity) train.samples[base + j] = sink.stage[j];
}

// DField::aux / DField::lobes of regions [0, n_regions): the per-lobe constants every mixture evaluation needs, and the
// region records re-laid as arrays of lobes (vspg_guided_wg.h)
__global__ __launch_bounds__(kBlock) void k_field_aux(const DScene *__restrict__ Sp, int f, const VspgFieldRegion *__restrict__ regs,
                                                      float *__restrict__ aux, float4 *__restrict__ lobes) {
    const int n = Sp->field[f].n_regions * GK;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const int r = i / GK, k = i - r * GK;
        const VspgFieldRegion &R = regs[r];
        const float kc = kappa_clamp(R.kappa[k]);
        const float b = R.weight[k] * vmf_norm(kc);  // b_k (vspg_guiding.h)
        aux[(size_t)r * (2 * GK) + k] = b;
        aux[(size_t)r * (2 * GK) + GK + k] = kc;
        float4 *L = lobes + (size_t)r * kRegionLobeQuads;
        if (k == 0) L[0] = make_float4(R.pivot[0], R.pivot[1], R.pivot[2], __builtin_bit_cast(float, (int)R.n_lobes));
        L[1 + 2 * k] = make_float4(R.mu[0][k], R.mu[1][k], R.mu[2][k], R.distance[k]);
        L[2 + 2 * k] = make_float4(R.weight[k], b, kc, R.vsp[k]);
    }
}

// ---- a18: Field::Update stand-in (definitions in vspg_train.h / oracle "Field::Update") --------------
// The surface (0) and the volume (1) field are updated TOGETHER: a sample belongs to exactly one of them (its VOLUME flag),
// the two updates are independent, and every pass over the batch is bound by reading the samples -- so each pass serves both
// fields under the sort key  field * kTrainCapRegions + region  (vspg_train.h: kTrainKeys).
struct TrainFields {
    RegionStats *stats[2];
    VspgFieldRegion *regs[2];
    VspgKdNode *nodes[2];
};
__global__ __launch_bounds__(kBlock) void k_train_decay(const DScene *__restrict__ Sp, TrainFields tf) {
    for (int f = 0; f < 2; ++f) {
        const int n = Sp->field[f].n_regions * kStatFloats;
        for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
            const int r = i / kStatFloats, k = i - r * kStatFloats;
            (&tf.stats[f][r].n)[k] *= kTrainDecay;
        }
    }
}
// ---- counting sort of the batch by key -----------------------------------------------------------------
// Histograms are built per workgroup in LDS over a contiguous chunk of the batch and flushed with one global
// atomic per non-empty bin: device-scope atomics on a handful of hot addresses (few regions in the first
// iterations) cost ~300 ns each when issued per wavefront.
constexpr int kSortBlock = 1024;  // one workgroup per CU: half the flushes of two 512-thread ones at the same occupancy
__device__ __forceinline__ void train_chunk(unsigned long long n, unsigned long long *lo, unsigned long long *hi) {
    unsigned long long per = (n + gridDim.x - 1) / gridDim.x;
    per = (per + kSortBlock - 1) / kSortBlock * kSortBlock;
    *lo = (unsigned long long)blockIdx.x * per < n ? (unsigned long long)blockIdx.x * per : n;
    *hi = *lo + per < n ? *lo + per : n;
}
// s_bins[key] += 1 for every lane with key >= 0; returns the bin's value before the lane's own add.
// A few ballot rounds serve the popular bins with one LDS atomic each, the tail goes lane by lane.
__device__ __forceinline__ unsigned int lds_bin_add(unsigned int *s_bins, int key) {
    const int lane = threadIdx.x & 63;
    unsigned long long todo = __ballot(key >= 0);
    unsigned int rank = 0;
    for (int round = 0; todo != 0ull && round < 4; ++round) {
        const int leader = __ffsll((long long)todo) - 1;
        const int lkey = __shfl(key, leader);
        const unsigned long long same = __ballot(key == lkey) & todo;
        unsigned int base = 0;
        if (lane == leader) base = atomicAdd(&s_bins[lkey], (unsigned int)__popcll(same));
        base = __shfl(base, leader);
        if ((same >> lane) & 1ull) rank = base + (unsigned int)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) rank = atomicAdd(&s_bins[key], 1u);
    return rank;
}
// key of every sample (-1: outside its field's tree) + histogram; the first of an update's two sorts also sums the sample
// weights (sumw != nullptr).  `redo`: the second sort runs only if the split pass changed a tree (*redo != 0); otherwise
// key_of / order / n_sorted of the first sort still stand.
constexpr int kTrainLdsNodes = 512;  // upper kd-tree levels per field staged in LDS for the descent (field_lookup)
__global__ __launch_bounds__(kSortBlock) void k_train_lookup(const DScene *__restrict__ Sp, const VspgTrainSample *__restrict__ samples,
                                                         unsigned long long n, int *__restrict__ key_of, unsigned int *__restrict__ hist,
                                                         float *__restrict__ sumw, const int *__restrict__ redo) {
    __shared__ unsigned int s_hist[kTrainKeys];
    __shared__ VspgKdNode s_nodes[2 * kTrainLdsNodes];
    if (redo && *redo == 0) return;
    const DScene &S = *Sp;
    int nl[2];
    for (int f = 0; f < 2; ++f) {
        nl[f] = S.field[f].n_nodes < kTrainLdsNodes ? S.field[f].n_nodes : kTrainLdsNodes;
        for (int i = threadIdx.x; i < nl[f]; i += kSortBlock) s_nodes[f * kTrainLdsNodes + i] = S.field[f].nodes[i];
    }
    for (int b = threadIdx.x; b < kTrainKeys; b += kSortBlock) s_hist[b] = 0;
    __syncthreads();
    unsigned long long lo, hi;
    train_chunk(n, &lo, &hi);
    float wsum = 0;
    for (unsigned long long i0 = lo; i0 < hi; i0 += kSortBlock) {
        const unsigned long long i = i0 + threadIdx.x;
        int key = -1;
        if (i < hi) {
            const VspgTrainSample sm = samples[i];
            const int f = (sm.flags & VSPG_SAMPLE_VOLUME) ? 1 : 0;
            const int region = field_lookup(S.field[f], ld3(sm.p), s_nodes + f * kTrainLdsNodes, nl[f]);
            if (region >= 0) key = f * kTrainCapRegions + region;
            key_of[i] = key;
            wsum += sm.weight;
        }
        lds_bin_add(s_hist, key);
    }
    if (sumw) {
        for (int off = 32; off > 0; off >>= 1) wsum += __shfl_xor(wsum, off);
        if ((threadIdx.x & 63) == 0 && wsum != 0.f) atomicAdd(sumw, wsum);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kTrainKeys; b += kSortBlock)
        if (s_hist[b]) atomicAdd(&hist[b], s_hist[b]);
}
// exclusive scan of the histogram -> start offsets (one block); cursor = copy for the scatter; total -> *n_sorted
__global__ __launch_bounds__(kBlock) void k_train_scan(const unsigned int *__restrict__ hist, unsigned int *__restrict__ cursor,
                                                       unsigned int *__restrict__ n_sorted, const int *__restrict__ redo) {
    __shared__ unsigned int s_part[kBlock];
    if (redo && *redo == 0) return;
    constexpr int per = (kTrainKeys + kBlock - 1) / kBlock;
    const int lo = threadIdx.x * per < kTrainKeys ? threadIdx.x * per : kTrainKeys, hi = lo + per < kTrainKeys ? lo + per : kTrainKeys;
    unsigned int sum = 0;
    for (int i = lo; i < hi; ++i) sum += hist[i];
    s_part[threadIdx.x] = sum;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned int run = 0;
        for (int t = 0; t < kBlock; ++t) { const unsigned int v = s_part[t]; s_part[t] = run; run += v; }
        *n_sorted = run;
    }
    __syncthreads();
    unsigned int run = s_part[threadIdx.x];
    for (int i = lo; i < hi; ++i) { cursor[i] = run; run += hist[i]; }
}
// a workgroup recounts its chunk, reserves one range per non-empty bin, then places its samples
__global__ __launch_bounds__(kSortBlock) void k_train_scatter(const int *__restrict__ key_of, unsigned long long n, unsigned int *__restrict__ cursor,
                                                          unsigned int *__restrict__ order, const int *__restrict__ redo) {
    __shared__ unsigned int s_cur[kTrainKeys];
    if (redo && *redo == 0) return;
    for (int b = threadIdx.x; b < kTrainKeys; b += kSortBlock) s_cur[b] = 0;
    __syncthreads();
    unsigned long long lo, hi;
    train_chunk(n, &lo, &hi);
    for (unsigned long long i0 = lo; i0 < hi; i0 += kSortBlock) {
        const unsigned long long i = i0 + threadIdx.x;
        lds_bin_add(s_cur, i < hi ? key_of[i] : -1);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < kTrainKeys; b += kSortBlock) {
        const unsigned int c = s_cur[b];
        s_cur[b] = c ? atomicAdd(&cursor[b], c) : 0u;  // from a count to the chunk's first slot of the bin
    }
    __syncthreads();
    for (unsigned long long i0 = lo; i0 < hi; i0 += kSortBlock) {
        const unsigned long long i = i0 + threadIdx.x;
        const int key = i < hi ? key_of[i] : -1;
        const unsigned int slot = lds_bin_add(s_cur, key);
        if (key >= 0) order[slot] = (unsigned int)i;
    }
}
// every wavefront takes a contiguous piece of the sorted order
__device__ __forceinline__ void train_piece(unsigned int n_sorted, unsigned int *lo, unsigned int *hi) {
    const unsigned int waves = gridDim.x * (kBlock / 64), w = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
    unsigned int per = (n_sorted + waves - 1) / waves;
    per = (per + 63u) & ~63u;
    *lo = w * per < n_sorted ? w * per : n_sorted;
    *hi = *lo + per < n_sorted ? *lo + per : n_sorted;
}
// position statistics {n, sum p, [sum p^2]} of the batch per key -> acc
// (`need`: the pass after the split serves k_train_init_regions only -- it runs when a region without lobes exists, flags[1] of k_train_split)
template <bool WITH_P2>
__global__ __launch_bounds__(kBlock) void k_train_pos(const VspgTrainSample *__restrict__ samples, const int *__restrict__ key_of,
                                                      const unsigned int *__restrict__ order, const unsigned int *__restrict__ n_sorted,
                                                      float *__restrict__ acc, const int *__restrict__ need) {
    if (need && *need == 0) return;
    unsigned int lo, hi;
    train_piece(*n_sorted, &lo, &hi);
    constexpr int NV = WITH_P2 ? 7 : 4;
    RunAccumulator<NV> R;
    R.init(acc, 0);
    for (unsigned int j0 = lo; j0 < hi; j0 += 64u) {
        const unsigned int j = j0 + (threadIdx.x & 63);
        const bool valid = j < hi;
        float v[RunAccumulator<NV>::P];
        int key = -1;
        for (int k = 0; k < RunAccumulator<NV>::P; ++k) v[k] = 0.f;
        if (valid) {
            const unsigned int i = order[j];
            key = key_of[i];
            const VspgTrainSample sm = samples[i];
            v[0] = 1.f; v[1] = sm.p[0]; v[2] = sm.p[1]; v[3] = sm.p[2];
            if constexpr (WITH_P2) { v[4] = sm.p[0] * sm.p[0]; v[5] = sm.p[1] * sm.p[1]; v[6] = sm.p[2] * sm.p[2]; }
        }
        R.add(valid, key, v);
    }
    R.flush();
}
// spatial refinement: one workgroup per field, sequential in meaning like the CPU definition so node / region numbering is the same
constexpr int kSplitBlock = 1024;
// flags[0] += regions created (the second sort runs only then), flags[1] |= a region without lobes exists (k_train_init_regions has work)
__global__ __launch_bounds__(kSplitBlock) void k_train_split(DScene *Sp, TrainFields tf, const float *acc_all, int *flags) {
    // The definition is sequential over the nodes (oracle/vspg_oracle.c:field_update_one): a leaf that wants to
    // split takes the next two node slots and the next region slot, until a capacity runs out.  Slots only grow,
    // so "rank among the wanting leaves" (a prefix sum) reproduces the sequential numbering exactly.
    __shared__ unsigned int s_wave[kSplitBlock / 64];
    __shared__ int s_n_nodes, s_n_regions;
    const int f = blockIdx.x;
    DField &F = Sp->field[f];
    RegionStats *stats = tf.stats[f];
    VspgFieldRegion *regs = tf.regs[f];
    VspgKdNode *nodes = tf.nodes[f];
    const float *acc = acc_all + (size_t)f * kTrainCapRegions * kStatFloats;
    const int n_reg0 = F.n_regions, n_nodes0 = F.n_nodes;
    bool bare = false;
    for (int i = threadIdx.x; i < n_reg0; i += kSplitBlock) {
        const float *a = acc + (size_t)i * kStatFloats;
        stats[i].n += a[0];
        for (int k = 0; k < 3; ++k) { stats[i].sum_p[k] += a[1 + k]; stats[i].sum_p2[k] += a[4 + k]; }
        bare = bare || regs[i].n_lobes == 0;  // (the regions a split creates are copies of these)
    }
    if (__ballot(bare) != 0ull && (threadIdx.x & 63) == 0) atomicOr(&flags[1], 1);
    __syncthreads();
    constexpr int kPer = (kTrainCapNodes + kSplitBlock - 1) / kSplitBlock;
    const int first = threadIdx.x * kPer;
    unsigned int want = 0;
    int axis_of[kPer];
    float split_of[kPer];
    for (int j = 0; j < kPer; ++j) {
        const int nd = first + j;
        axis_of[j] = 0; split_of[j] = 0;
        if (nd >= n_nodes0 || (nodes[nd].packed & 3u) != 3u) continue;
        const RegionStats &s0 = stats[nodes[nd].packed >> 2];
        if (!(s0.n > kTrainSplitCount) || s0.depth >= kTrainMaxDepth) continue;
        float mean[3], var[3];
        for (int k = 0; k < 3; ++k) {
            mean[k] = s0.sum_p[k] / s0.n;
            var[k] = s0.sum_p2[k] / s0.n - mean[k] * mean[k];
        }
        const int axis = var[0] >= var[1] ? (var[0] >= var[2] ? 0 : 2) : (var[1] >= var[2] ? 1 : 2);
        if (!(var[axis] > 0)) continue;
        want |= 1u << j;
        axis_of[j] = axis;
        split_of[j] = mean[axis];
    }
    // exclusive prefix sum of popc(want) over the threads: within the wavefront by shuffles, across wavefronts through LDS
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned int mine = (unsigned int)__popc(want);
    unsigned int incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int up = __shfl_up(incl, off);
        if (lane >= off) incl += up;
    }
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    unsigned int before = 0, total = 0;
    for (int w = 0; w < kSplitBlock / 64; ++w) {
        const unsigned int c = s_wave[w];
        if (w < wave) before += c;
        total += c;
    }
    if (threadIdx.x == 0) {
        // how many of the `total` candidates fit: both capacities
        int fit = (int)total;
        if (fit > (kTrainCapNodes - n_nodes0) / 2) fit = (kTrainCapNodes - n_nodes0) / 2;
        if (fit > kTrainCapRegions - n_reg0) fit = kTrainCapRegions - n_reg0;
        if (fit < 0) fit = 0;
        s_n_nodes = n_nodes0 + 2 * fit;
        s_n_regions = n_reg0 + fit;
        if (fit) atomicAdd(&flags[0], fit);
    }
    int rank = (int)(before + incl - mine);
    for (int j = 0; j < kPer; ++j) {
        if (!((want >> j) & 1u)) continue;
        const int left = n_nodes0 + 2 * rank, newreg = n_reg0 + rank;
        ++rank;
        if (left + 2 > kTrainCapNodes || newreg + 1 > kTrainCapRegions) continue;
        const int nd = first + j;
        const int reg = (int)(nodes[nd].packed >> 2);
        RegionStats &s0 = stats[reg];
        float *v = &s0.n;
        for (int k = 0; k < kStatFloats; ++k) v[k] *= 0.5f;
        s0.depth += 1;
        stats[newreg] = s0;
        regs[newreg] = regs[reg];
        nodes[left].split = 0; nodes[left].packed = ((uint32_t)reg << 2) | 3u;
        nodes[left + 1].split = 0; nodes[left + 1].packed = ((uint32_t)newreg << 2) | 3u;
        nodes[nd].split = split_of[j];
        nodes[nd].packed = ((uint32_t)left << 2) | (uint32_t)axis_of[j];
    }
    __syncthreads();
    if (threadIdx.x == 0) { F.n_nodes = s_n_nodes; F.n_regions = s_n_regions; }
}
__global__ __launch_bounds__(kBlock) void k_train_init_regions(const DScene *__restrict__ Sp, TrainFields tf, const float *__restrict__ acc) {
    const int key = blockIdx.x * kBlock + threadIdx.x;
    if (key >= kTrainKeys) return;
    const int f = key >= kTrainCapRegions ? 1 : 0, i = key - f * kTrainCapRegions;
    if (i >= Sp->field[f].n_regions) return;
    const float *a = acc + (size_t)key * kStatFloats;
    VspgFieldRegion &R = tf.regs[f][i];
    if (R.n_lobes == 0 && a[0] > 0) {
        for (int k = 0; k < 3; ++k) R.pivot[k] = a[1 + k] / a[0];
        region_init_lobes(R);
    }
}
__global__ __launch_bounds__(kBlock) void k_train_estep(const DScene *__restrict__ Sp, const VspgTrainSample *__restrict__ samples,
                                                        const int *__restrict__ key_of, const unsigned int *__restrict__ order,
                                                        const unsigned int *__restrict__ n_sorted, const float *__restrict__ sumw,
                                                        float *__restrict__ acc) {
    const DScene &S = *Sp;
    vspg_libm::stage_logf_tab_lds();
    __syncthreads();
    const float wmax = kTrainWeightClamp * (sumw[0] / sumw[1]);  // mean sample weight: sum / count, both over ALL ranks' samples (train_update)
    unsigned int lo, hi;
    train_piece(*n_sorted, &lo, &hi);
    RunAccumulator<8 * GK> R;
    R.init(acc, 7);
    for (unsigned int j0 = lo; j0 < hi; j0 += 64u) {
        const unsigned int j = j0 + (threadIdx.x & 63);
        bool valid = j < hi;
        int key = -1;
        float vals[8 * GK];  // S, R0, R1, R2, D, V, Qv, Qs -- the order of RegionStats
        for (int k = 0; k < 8 * GK; ++k) vals[k] = 0.f;
        if (valid) {
            const unsigned int i = order[j];
            key = key_of[i];
            const VspgTrainSample sm = samples[i];
            const int f = key >= kTrainCapRegions ? 1 : 0;
            const VspgFieldRegion &Rg = S.field[f].regions[key - f * kTrainCapRegions];
            const int nl = Rg.n_lobes < GK ? Rg.n_lobes : GK;
            valid = nl > 0;
            if (valid) {
                const float w = sm.weight < wmax ? sm.weight : wmax;
                const V3 om = train_reaim(Rg, ld3(sm.p), ld3(sm.dir), sm.distance);
                float g[GK], gs = 0;
                for (int k = 0; k < GK; ++k) {
                    g[k] = k < nl ? Rg.weight[k] * vmf_eval(V3{Rg.mu[0][k], Rg.mu[1][k], Rg.mu[2][k]}, kappa_clamp(Rg.kappa[k]), om) : 0.f;
                    gs += g[k];
                }
                valid = gs > 0 && !isinf_(gs);
                if (valid) {
                    const bool nextvol = (sm.flags & VSPG_SAMPLE_NEXT_VOLUME) != 0;
                    const bool hasd = sm.distance > 0 && !isinf_(sm.distance);
                    for (int k = 0; k < GK; ++k) {
                        const float wg = w * (g[k] / gs);
                        vals[0 * GK + k] = wg;
                        vals[1 * GK + k] = wg * om.x; vals[2 * GK + k] = wg * om.y; vals[3 * GK + k] = wg * om.z;
                        vals[4 * GK + k] = hasd ? wg / sm.distance : 0.f;
                        vals[5 * GK + k] = nextvol ? wg : 0.f;
                        vals[6 * GK + k] = nextvol ? wg * w : 0.f;
                        vals[7 * GK + k] = nextvol ? 0.f : wg * w;
                    }
                }
            }
        }
        R.add(valid, key, vals);
    }
    R.flush();
}
__global__ __launch_bounds__(kBlock) void k_train_mstep(const DScene *__restrict__ Sp, TrainFields tf, const float *__restrict__ acc) {
    const int key = blockIdx.x * kBlock + threadIdx.x;
    if (key >= kTrainKeys) return;
    const int f = key >= kTrainCapRegions ? 1 : 0, i = key - f * kTrainCapRegions;
    if (i >= Sp->field[f].n_regions) return;
    VspgFieldRegion &R = tf.regs[f][i];
    if (R.n_lobes <= 0) return;
    const int nl = R.n_lobes < GK ? R.n_lobes : GK;
    RegionStats &s1 = tf.stats[f][i];
    const float *a = acc + (size_t)key * kStatFloats + 7;
    float Stot = 0;
    for (int k = 0; k < nl; ++k) {
        s1.S[k] += a[0 * GK + k];
        s1.R[0][k] += a[1 * GK + k]; s1.R[1][k] += a[2 * GK + k]; s1.R[2][k] += a[3 * GK + k];
        s1.D[k] += a[4 * GK + k]; s1.V[k] += a[5 * GK + k]; s1.Qv[k] += a[6 * GK + k]; s1.Qs[k] += a[7 * GK + k];
        Stot += s1.S[k];
    }
    if (!(Stot > 0)) return;
    const float floorw = 1e-3f / GK;
    float wsum = 0;
    for (int k = 0; k < nl; ++k) {
        float wk = s1.S[k] / Stot;
        wk = wk < floorw ? floorw : wk;
        R.weight[k] = wk;
        wsum += wk;
        const float rl = sqrtf(s1.R[0][k] * s1.R[0][k] + s1.R[1][k] * s1.R[1][k] + s1.R[2][k] * s1.R[2][k]);
        if (s1.S[k] > 0 && rl > 0) {
            R.mu[0][k] = s1.R[0][k] / rl; R.mu[1][k] = s1.R[1][k] / rl; R.mu[2][k] = s1.R[2][k] / rl;
            float rbar = rl / s1.S[k];
            rbar = rbar > 0.9999f ? 0.9999f : rbar;
            R.kappa[k] = kappa_clamp(rbar * (3 - rbar * rbar) / (1 - rbar * rbar));
            R.distance[k] = s1.D[k] > 0 ? s1.S[k] / s1.D[k] : kInf;
            if (Sp->prm.vspcriterion == VSPG_VSP_VARIANCE) {
                const float qv = sqrtf(s1.Qv[k]), qs = sqrtf(s1.Qs[k]);
                R.vsp[k] = qv + qs > 0 ? qv / (qv + qs) : 0.5f;
            } else {
                R.vsp[k] = s1.V[k] / s1.S[k];
            }
        }
    }
    for (int k = 0; k < nl; ++k) R.weight[k] = R.weight[k] / wsum;
}

// Workgroup-level wavefront kernel (see vspg_wg_kernel.h for the design): a persistent workgroup
// advances a pool of NP paths parked in LDS, phase by phase, over compacted lists.  Iteration k
// (parity par = k & 1, nxt = par ^ 1):
//   assign  wavefront 0 maps the free slots of s_free[par] to work items (s_item), claiming chunks of
//           kWgChunk items from the global head;
//   S       "segment" phase over one dense index space
//             [0, nAssign)           fresh slots: camera ray + PRIMARY segment (VSP-guided distance sampling)
//             [.., + A0[par])        slots whose pixel starts its next sample (multi-sample launches): same
//             [.., + A1[par])        continuing paths: one SECONDARY segment (plain delta tracking)
//           survivors go to the vertex list (volume vertices from the front of s_listB, surface vertices
//           from the back), finished paths are splatted and their slot goes to s_free[nxt] or, when the
//           pixel has samples left, to the front of s_listA[nxt];
//   V       vertex phase over s_listB: NEE, Russian roulette, new direction; survivors go to the back of
//           s_listA[nxt].
// Fusing the camera ray with the primary segment keeps the heaviest code (optical-depth-space sampling)
// where the path state is still compile-time constant (beta = r_u = r_l = 1, L = 0), and keeping it out
// of the secondary phase is what lets the whole kernel fit the 128-VGPR budget of 4 waves per SIMD.
// Launch shapes.  Homogeneous media: the segment code has no loops left (see HomogeneousMedium::
// kAlwaysRealCollision) and the kernel fits 128 VGPRs: 512-thread workgroups, 4 waves per SIMD, two
// workgroups per CU, 512 paths each.  Grid media keep the DDA / null-collision loops (about 250 live
// VGPRs): 256-thread workgroups at 2 waves per SIMD, 368 paths beside the 16 KB majorant grid in LDS.
#ifndef VSPG_WG_WAVES
#define VSPG_WG_WAVES 4
#endif
#ifndef VSPG_WG_BLOCK
#define VSPG_WG_BLOCK 512
#endif
#ifndef VSPG_WG_NP
#define VSPG_WG_NP 512
#endif
constexpr int kWgWavesHomog = VSPG_WG_WAVES, kWgBlockHomog = VSPG_WG_BLOCK, kWgPoolHomog = VSPG_WG_NP;
constexpr int kWgWavesGrid = 2, kWgBlockGrid = 256, kWgPoolGrid = 384;
#ifndef VSPG_WGG_NP
#define VSPG_WGG_NP 320
#endif
#ifndef VSPG_WGG_WAVES
#define VSPG_WGG_WAVES 4
#endif
#ifndef VSPG_WGG_BLOCK
#define VSPG_WGG_BLOCK 512
#endif
// guided vertices on the workgroup kernel (round 3, vspg_guided_wg.h): 128 registers, four waves per SIMD like the unguided kernel
constexpr int kWgWavesGuided = VSPG_WGG_WAVES, kWgBlockGuided = VSPG_WGG_BLOCK, kWgPoolGuided = VSPG_WGG_NP;
// Paths per pool: what fits the 80 KB a workgroup may use when two share a CU -- the record (PoolLayout::COUNT dwords), the
// lists' entries per path, and `other` bytes of fixed LDS (scene records, tables, counters, staged kd nodes); a multiple of 32.
#ifndef VSPG_WG_LDS_BUDGET
#define VSPG_WG_LDS_BUDGET 81920
#endif
#ifndef VSPG_WG_NP_CAP
#define VSPG_WG_NP_CAP 512
#endif
template <class LY>
constexpr int wg_pool_paths(int list_bytes_per_path, int other_bytes) {
    return (VSPG_WG_LDS_BUDGET - other_bytes) / (LY::COUNT * 4 + list_bytes_per_path) / 32 * 32;
}
#ifndef VSPG_WG_OTHER
#define VSPG_WG_OTHER 5500
#endif
// The unguided kernels run eight waves per workgroup over 64-entry chunks: 512 paths give every wave one chunk per phase, and a
// pool that is not a multiple of that leaves a second round of partly filled chunks behind (608 paths: 0.849 against 0.804 ms per
// 1080p wave) -- so the records the grey instantiations save are not spent on more paths there.
template <int GREY> constexpr int kWgPoolHomogT = wg_pool_paths<PoolLayout<false, GREY>>(14, VSPG_WG_OTHER) < 512 ? wg_pool_paths<PoolLayout<false, GREY>>(14, VSPG_WG_OTHER) : 512;  // k_render_wave_wg: + s_item
template <int GREY> constexpr int kWg2PoolHomogT = wg_pool_paths<PoolLayout<false, GREY>>(10, VSPG_WG_OTHER) < VSPG_WG_NP_CAP ? wg_pool_paths<PoolLayout<false, GREY>>(10, VSPG_WG_OTHER) : VSPG_WG_NP_CAP;
// full scenes (triangles, spheres, infinite lights, medium boundaries): the generic record + isg.vsp_used; the stage_scene_lds copy is the same
constexpr int kWg2PoolFull = wg_pool_paths<PoolLayout<false, 0, false, true>>(10, VSPG_WG_OTHER) < VSPG_WG_NP_CAP ? wg_pool_paths<PoolLayout<false, 0, false, true>>(10, VSPG_WG_OTHER) : VSPG_WG_NP_CAP;
// guided: every path counts (320 -> 384 -> 416 paths: 2.14 -> 1.86 -> 1.78 ms per trained 1080p wave)
#ifndef VSPG_WGG_NP_G0
#define VSPG_WGG_NP_G0 wg_pool_paths<PoolLayout<true, 0>>(10, VSPG_WG_OTHER + 2 * kKdLdsNodes * 8)
#endif
#ifndef VSPG_WGG_NP_G2
#define VSPG_WGG_NP_G2 wg_pool_paths<PoolLayout<true, 2>>(10, VSPG_WG_OTHER + 2 * kKdLdsNodes * 8)
#endif
template <int GREY> constexpr int kWg2PoolGuidedT = GREY >= 2 ? (VSPG_WGG_NP_G2) : (VSPG_WGG_NP_G0);
template <int GREY> constexpr int kWg2PoolTrainT = wg_pool_paths<PoolLayout<true, GREY, true>>(10, VSPG_WG_OTHER + 2 * kKdLdsNodes * 8);  // + the recorder's state
constexpr int kWgChunk = 256;  // work items (4 pixel tiles) a workgroup claims per global atomic
enum { C_A0 = 0, C_A1 = 2, C_CURA = 4, C_BV = 6, C_BS = 7, C_CURB = 8, C_NFREE = 9, C_NASSIGN = 11, C_RNEXT = 12, C_REND = 13,
       C_EXH = 14, C_RTX = 15, C_RTY = 16, C_COUNT = 17 };

template <class Medium, bool GUIDED, int NP, int kWgBlock, int kWgWavesPerSimd>
__global__ __launch_bounds__(kWgBlock, kWgWavesPerSimd) void k_render_wave_wg(
    const DScene *__restrict__ Sp, float4 *__restrict__ film, float *__restrict__ isg_stats, const float *__restrict__ vsp_buf,
    int vsp_ready, int wave_end, int first_sample, int single_sample, PcgJump jump, unsigned int tiles_magic,
    unsigned int *__restrict__ work_head, unsigned long long *__restrict__ counters) {
    // tiles_magic = ceil(2^32 / tilesX): the one integer division of the kernel (tile index -> tile row, once
    // per claimed chunk) is a multiply-high by it plus a fix-up; pixels travel as packed (x | y << 16)
    const DScene &S = *Sp;
    const int W = S.xres, H = S.yres;
    const int tilesX = (W + 7) >> 3, tilesY = (H + 7) >> 3;
    const unsigned total_items = (unsigned)(tilesX * tilesY) * 64u;
    const int lane = threadIdx.x & 63;
    const int sample_step = S.shard_count > 1 ? S.shard_count : 1;
    reset_sibling_head(work_head);

    using LY = PoolLayout<GUIDED, Medium::kGrey>;
    constexpr int NF = LY::COUNT;
    // (Tried in round 2 and dropped: the pool in the workgroup's own piece of GLOBAL memory instead of LDS -- the segment-streaming
    // layout SURVEY 8d's byte model describes -- 0.82 ms (LDS, 512 paths) -> 2.49 / 3.39 / 2.95 ms with 1024 / 2048 / 4096 paths.)
    __shared__ float s_pool[NF * NP];
    float *const pool_base = s_pool;
    __shared__ unsigned short s_listA[2][NP], s_listB[NP], s_free[2][NP];
    __shared__ unsigned int s_item[NP];
    __shared__ unsigned int s_cnt[C_COUNT];
    const Pool P{pool_base, NP};
    // Heterogeneous media: a tracking walk visits every tentative collision of its ray, and the rays of one list chunk
    // have wildly different expected counts (0 for a ray that misses the cloud, tens through its core): a chunk lasts
    // as long as its longest walk.  Before the segment phase the continuing paths are therefore counting-sorted by the
    // majorant optical depth of their ray (the traversals' own pre-pass quantity), thickest first, so that the lanes of
    // a chunk walk about equally long and the dynamic chunk queue starts with the long chunks.
    constexpr bool kSortWalks = !Medium::kSingleSegment;
    constexpr int kWalkBuckets = 32;
    __shared__ unsigned short s_order[kSortWalks ? NP : 1];
    __shared__ unsigned char s_key[kSortWalks ? NP : 1];
    __shared__ unsigned int s_hist[kSortWalks ? kWalkBuckets : 1];

    const float *maj_ptr = nullptr;
    if constexpr (std::is_same<Medium, GridMedium>::value || std::is_same<Medium, GridMediumGrey>::value) {
        __shared__ float s_maj[kMajRes * kMajRes * kMajRes];
        const float4 *src = reinterpret_cast<const float4 *>(S.majorant);
        float4 *dst = reinterpret_cast<float4 *>(s_maj);
        for (int i = threadIdx.x; i < kMajRes * kMajRes * kMajRes / 4; i += kWgBlock) dst[i] = src[i];
        maj_ptr = s_maj;
    }
    const Medium medium = MediumMaker<Medium>::make(S, maj_ptr);
    // guided builds keep the mixture scratch in registers here (GStoreReg): the LDS belongs to the pool -- and to a copy of
    // the upper levels of the two kd-trees (north star: "LDS-staged kd-tree nodes"), which `glds` then points at
    float *glds = nullptr;
    if constexpr (GUIDED) {
        __shared__ VspgKdNode s_kd[2][kKdLdsNodes];
        for (int f = 0; f < 2; ++f) {
            const int nl = S.field[f].n_nodes < kKdLdsNodes ? S.field[f].n_nodes : kKdLdsNodes;
            for (int i = threadIdx.x; i < nl; i += kWgBlock) s_kd[f][i] = S.field[f].nodes[i];
        }
        glds = reinterpret_cast<float *>(&s_kd[0][0]);
    }
    __shared__ unsigned int s_counters[CNT_COUNT];
#ifdef VSPG_WG_WAVE_COUNTERS  // alternative sink: no per-lane registers, ~2 % slower (measured)
    const WaveCounters pc{s_counters};
#else
    struct LaneCounters : PathCounters { uint32_t paths; VDEV void path() { paths++; } } pc;
    pc.segments = pc.volume_scatters = pc.surface_hits = pc.density_queries = pc.shadow_rays = pc.shadow_queries = pc.paths = 0;
#endif

    stage_scene_lds(S);
    if (threadIdx.x < CNT_COUNT) s_counters[threadIdx.x] = 0;
    if (threadIdx.x < C_COUNT) s_cnt[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < NP; i += kWgBlock) {
        s_free[0][i] = (unsigned short)i;
        P.u(LY::FLAGS, i) = 0;
    }
    __syncthreads();
    if (threadIdx.x == 0) s_cnt[C_NFREE] = NP;
    __syncthreads();

    VSPG_PROF(PS_WG_TOTAL);
    for (int k = 0;; ++k) {
        const int par = k & 1, nxt = par ^ 1;
        // ---- assign: work items for the free slots ---------------------------------------------------
        if (threadIdx.x < 64) {
            const unsigned n_free = s_cnt[C_NFREE + par];
            unsigned rnext = s_cnt[C_RNEXT], rend = s_cnt[C_REND], exh = s_cnt[C_EXH];
            unsigned rtx = s_cnt[C_RTX], rty = s_cnt[C_RTY];  // tile column / row of item rnext
            unsigned filled = 0;
            while (filled < n_free) {
                if (rnext >= rend) {
                    if (exh) break;
                    unsigned base = 0;
                    if (lane == 0) base = atomicAdd(work_head, (unsigned)kWgChunk);
                    base = __builtin_amdgcn_readfirstlane(base);
                    if (base >= total_items) { exh = 1; break; }
                    rnext = base;
                    rend = base + (unsigned)kWgChunk < total_items ? base + (unsigned)kWgChunk : total_items;
                    const unsigned tile = base >> 6;
                    rty = tilesX == 1 ? tile : __umulhi(tile, tiles_magic);
                    rtx = tile - rty * (unsigned)tilesX;
                    while (rtx >= (unsigned)tilesX) { rtx -= (unsigned)tilesX; rty++; }
                }
                const unsigned n = rend - rnext < n_free - filled ? rend - rnext : n_free - filled;
                for (unsigned j = (unsigned)lane; j < n; j += 64u) {
                    const unsigned it = (rnext & 63u) + j;
                    unsigned tx = rtx + (it >> 6), ty = rty;
                    while (tx >= (unsigned)tilesX) { tx -= (unsigned)tilesX; ty++; }
                    s_item[filled + j] = (tx * 8u + (it & 7u)) | ((ty * 8u + ((it >> 3) & 7u)) << 16);
                }
                rtx += ((rnext & 63u) + n) >> 6;
                while (rtx >= (unsigned)tilesX) { rtx -= (unsigned)tilesX; rty++; }
                rnext += n;
                filled += n;
            }
            if (lane == 0) {
                s_cnt[C_RNEXT] = rnext; s_cnt[C_REND] = rend; s_cnt[C_EXH] = exh; s_cnt[C_RTX] = rtx; s_cnt[C_RTY] = rty;
                s_cnt[C_NASSIGN] = filled;  // free slots beyond `filled` stay empty: the work has run out
                s_cnt[C_BV] = 0; s_cnt[C_BS] = 0; s_cnt[C_CURB] = 0;
                s_cnt[C_A0 + nxt] = 0; s_cnt[C_A1 + nxt] = 0; s_cnt[C_CURA + nxt] = 0;
            }
        } else if (single_sample) {
            // ---- film flush, by the waves the assignment leaves idle ----------------------------------
            // Paths that ended in the previous iteration parked their radiance in their (now free) slot.  In a
            // launch of ONE sample per pixel nobody else touches the pixel, so the film and the ISG statistics take a
            // plain 16-byte load / add / store: the same single IEEE addition per channel as the ten no-return float
            // atomics this replaces, which were L2-atomic-unit bound (0.28 ms of a 0.42 ms maxdepth-0 wave at 1080p).
            // Done here in bulk, every lane busy, the load latency is paid once per 64 finished paths.
            // (Tried: reading the parked record here and doing the global read-modify-write after the barrier, so that
            //  its latency overlaps segment work -- the registers live across the barrier cost more than the overlap won.)
            const unsigned n_free = s_cnt[C_NFREE + par];
            for (unsigned base = (threadIdx.x >> 6) * 64u - 64u; base < n_free; base += (unsigned)kWgBlock - 64u) {
                const unsigned i = base + (unsigned)lane;
                if (i < n_free) {
                    const int slot = s_free[par][i];
                    const uint32_t fl = P.u(LY::FLAGS, slot);
                    if (fl & FL_DONE) {
                        const int pxy = P.i(LY::PIXEL, slot);
                        const size_t pidx = (size_t)((unsigned)pxy >> 16) * W + (pxy & 0xffff);
                        const Spec L = P.sp3(LY::L, slot);
                        IsgSample isg;
                        isg.valid = (fl & FL_ISG_VALID) != 0;
                        isg.surface_event = (fl & FL_ISG_SURF) != 0;
                        isg.vsp_used = P.f(LY::VSP, slot);
                        film_add_sample_rmw(film + pidx, L);
                        isg_add_sample_rmw(isg_stats + pidx * VSPG_ISG_STATS, L, isg);
                        P.u(LY::FLAGS, slot) = 0;
                    }
                }
            }
        }
        { VSPG_PROF(PS_WG_BAR_R); __syncthreads(); }
        if (threadIdx.x == 0) s_cnt[C_NFREE + par] = 0;  // (read by the flush above; refilled from the next iteration on)
        const unsigned nFresh = s_cnt[C_NASSIGN], nA0 = s_cnt[C_A0 + par], nA1 = s_cnt[C_A1 + par];
        const unsigned nPrim = nFresh + nA0, nA = nPrim + nA1;
        if (nA == 0 && s_cnt[C_EXH]) break;  // nothing in flight and nothing left to start

        if constexpr (kSortWalks) {  // ---- order the continuing paths by expected walk length -----------------------
            if (threadIdx.x < kWalkBuckets) s_hist[threadIdx.x] = 0;
            __syncthreads();
            for (unsigned j = threadIdx.x; j < nA1; j += (unsigned)kWgBlock) {
                const int slot = s_listA[par][NP - 1 - (int)j];
                const V3 ro = P.v3(LY::RO, slot), rd = P.v3(LY::RD, slot);
                const int ch = (int)((P.u(LY::FLAGS, slot) >> FL_CH_SHIFT) & 3u);
                const Isect si = scene_intersect(S, ro, rd, kInf);
                float tau = 0.f;
                if (si.hit && S.medium_type != VSPG_MEDIUM_NONE) tau = majorant_optical_depth(medium, ro, rd, si.t, ch);
                int b = tau > 0.f ? 1 + (int)(tau * 0.75f) : 0;
                b = b > kWalkBuckets - 1 ? kWalkBuckets - 1 : b;
                s_key[j] = (unsigned char)b;
                atomicAdd(&s_hist[b], 1u);
            }
            __syncthreads();
            if (threadIdx.x == 0) {  // exclusive offsets, thickest bucket first
                unsigned run = 0;
                for (int b = kWalkBuckets - 1; b >= 0; --b) { const unsigned c = s_hist[b]; s_hist[b] = run; run += c; }
            }
            __syncthreads();
            for (unsigned j = threadIdx.x; j < nA1; j += (unsigned)kWgBlock) {
                const unsigned pos = atomicAdd(&s_hist[s_key[j]], 1u);
                s_order[pos] = s_listA[par][NP - 1 - (int)j];
            }
            __syncthreads();
        }
        // ---- S: camera ray + primary segment for new paths, one secondary segment for the others ------
        while (true) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt[C_CURA + par], 64u);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= nA) break;
            VSPG_PROF(PS_WG_A);
            const unsigned i = base + (unsigned)lane;
            bool toV = false, toS = false, restart = false, freed = false;
            int slot = 0;
            if (i < nA) {
                Sampler sampler;
                PathState st;
                IsgSample isg;
                int ch = 0, pxy = 0;
                Vertex vx;
                bool alive = false, valid = true;
                if (i < nPrim) {
                    int px, py, smp;
                    if (i < nFresh) {
                        slot = s_free[par][i];
                        pxy = (int)s_item[i];
                        smp = first_sample;
                    } else {
                        slot = s_listA[par][i - nFresh];
                        pxy = P.i(LY::PIXEL, slot);
                        smp = P.i(LY::SAMPLE, slot);
                    }
                    px = pxy & 0xffff;
                    py = (int)((unsigned)pxy >> 16);
                    valid = px < W && py < H && smp < wave_end;  // tile padding: the slot stays free
                    if (valid) {
                        if (single_sample)
                            start_path(S, vsp_buf, vsp_ready, px, py, jump, sampler, st, &ch, isg);
                        else
                            start_path(S, vsp_buf, vsp_ready, px, py, smp, sampler, st, &ch, isg);
                        P.i(LY::PIXEL, slot) = pxy;
                        P.i(LY::SAMPLE, slot) = smp;
                        alive = li_segment_a<Medium, GUIDED, SEG_PRIMARY>(S, medium, vsp_buf, vsp_ready, px, py, st, ch, sampler,
                                                                          isg, pc, vx);
                        if (alive) {
                            pool_store_full<GUIDED, Medium::kGrey>(P, slot, st, sampler, ch, isg, FL_LIVE | (vx.volume ? (uint32_t)FL_VX_VOLUME : 0u));
                            pool_store_vertex<GUIDED, Medium::kGrey>(P, slot, vx);
                        }
                    } else {
                        freed = true;
                    }
                } else {
                    if constexpr (kSortWalks) slot = s_order[i - nPrim];
                    else slot = s_listA[par][NP - 1 - (int)(i - nPrim)];
                    const uint32_t fl = pool_load<GUIDED, Medium::kGrey>(P, slot, S, st, sampler, &ch, isg);
                    pxy = P.i(LY::PIXEL, slot);
                    const int px = pxy & 0xffff, py = (int)((unsigned)pxy >> 16);
                    alive = li_segment_a<Medium, GUIDED, SEG_SECONDARY>(S, medium, vsp_buf, vsp_ready, px, py, st, ch, sampler,
                                                                        isg, pc, vx);
                    if (alive) pool_store_a<Medium::kGrey, GUIDED>(P, slot, st, sampler, ch, isg, vx, fl & (FL_LIVE | FL_GS_SCATTER | FL_GS_FIELD));
                }
                if (alive) {
                    toV = vx.volume;
                    toS = !vx.volume;
                } else if (valid) {
                    const Spec L = finish_radiance(st.L);
                    const size_t pidx = (size_t)((unsigned)pxy >> 16) * W + (pxy & 0xffff);
                    if (single_sample) {  // parked for the film flush at the top of the next iteration
                        P.sets(LY::L, slot, L);
                        P.i(LY::PIXEL, slot) = pxy;
                        P.f(LY::VSP, slot) = isg.vsp_used;
                        P.u(LY::FLAGS, slot) = (uint32_t)FL_DONE | (isg.valid ? (uint32_t)FL_ISG_VALID : 0u) | (isg.surface_event ? (uint32_t)FL_ISG_SURF : 0u);
                    } else {
                        film_add_sample(film + pidx, L);
                        isg_add_sample_atomic(isg_stats + pidx * VSPG_ISG_STATS, L, isg);
                    }
                    pc.path();
                    const int s2 = P.i(LY::SAMPLE, slot) + sample_step;
                    P.i(LY::SAMPLE, slot) = s2;
                    restart = s2 < wave_end;
                    freed = !restart;
                }
            }
            list_push(toV, slot, s_listB, &s_cnt[C_BV]);
            list_push_back(toS, slot, s_listB + NP - 1, &s_cnt[C_BS]);
            list_push(restart, slot, s_listA[nxt], &s_cnt[C_A0 + nxt]);
            list_push(freed, slot, s_free[nxt], &s_cnt[C_NFREE + nxt]);
        }
        { VSPG_PROF(PS_WG_BAR_A); __syncthreads(); }

        // ---- V: vertex processing (NEE, Russian roulette, new direction) ----------------------------
        const unsigned nBV = s_cnt[C_BV], nB = nBV + s_cnt[C_BS];
        while (true) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt[C_CURB], 64u);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= nB) break;
            VSPG_PROF(PS_WG_B);
            const unsigned i = base + (unsigned)lane;
            bool cont = false, restart = false, freed = false;
            int slot = 0;
            if (i < nB) {
                slot = i < nBV ? s_listB[i] : s_listB[NP - 1 - (int)(i - nBV)];
                Sampler sampler;
                PathState st;
                IsgSample isg;
                int ch;
                const uint32_t fl = pool_load<GUIDED, Medium::kGrey>(P, slot, S, st, sampler, &ch, isg);
                const Vertex vx = pool_load_vertex<GUIDED, Medium::kGrey>(P, slot, fl);
                if (li_segment_b<Medium, GUIDED, GUIDED>(S, medium, st, ch, sampler, pc, vx, glds, kWgBlock)) {
                    pool_store_full<GUIDED, Medium::kGrey>(P, slot, st, sampler, ch, isg, FL_LIVE);
                    cont = true;
                } else {
                    const int pxy = P.i(LY::PIXEL, slot);
                    const size_t pidx = (size_t)((unsigned)pxy >> 16) * W + (pxy & 0xffff);
                    const Spec L = finish_radiance(st.L);
                    if (single_sample) {  // parked for the film flush at the top of the next iteration
                        P.sets(LY::L, slot, L);
                        P.i(LY::PIXEL, slot) = pxy;
                        P.f(LY::VSP, slot) = isg.vsp_used;
                        P.u(LY::FLAGS, slot) = (uint32_t)FL_DONE | (isg.valid ? (uint32_t)FL_ISG_VALID : 0u) | (isg.surface_event ? (uint32_t)FL_ISG_SURF : 0u);
                    } else {
                        film_add_sample(film + pidx, L);
                        isg_add_sample_atomic(isg_stats + pidx * VSPG_ISG_STATS, L, isg);
                    }
                    pc.path();
                    const int s2 = P.i(LY::SAMPLE, slot) + sample_step;
                    P.i(LY::SAMPLE, slot) = s2;
                    restart = s2 < wave_end;
                    freed = !restart;
                }
            }
            list_push_back(cont, slot, s_listA[nxt] + NP - 1, &s_cnt[C_A1 + nxt]);
            list_push(restart, slot, s_listA[nxt], &s_cnt[C_A0 + nxt]);
            list_push(freed, slot, s_free[nxt], &s_cnt[C_NFREE + nxt]);
        }
        { VSPG_PROF(PS_WG_BAR_B); __syncthreads(); }
    }
    static_assert(CNT_COUNT == kNumCounters, "counter layout");
#ifndef VSPG_WG_WAVE_COUNTERS
    atomicAdd(&s_counters[CNT_PATHS], pc.paths); atomicAdd(&s_counters[CNT_SEGMENTS], pc.segments);
    atomicAdd(&s_counters[CNT_VOLUME_SCATTERS], pc.volume_scatters); atomicAdd(&s_counters[CNT_SURFACE_HITS], pc.surface_hits);
    atomicAdd(&s_counters[CNT_DENSITY_QUERIES], pc.density_queries); atomicAdd(&s_counters[CNT_SHADOW_RAYS], pc.shadow_rays);
    atomicAdd(&s_counters[CNT_SHADOW_QUERIES], pc.shadow_queries);
    __syncthreads();
#endif
    if (threadIdx.x < CNT_COUNT) atomicAdd(&counters[threadIdx.x], (unsigned long long)s_counters[threadIdx.x]);
}

// ---------------------------------------------------------------------------------------------------------------------
// k_render_wave_wg2 (round 3): the same pool / phase design with the scheduler's serial pieces taken out.
//   * work assignment: whole pixel tiles for the pool's free slots, claimed from a GLOBAL tile head with one returning atomic per
//     claim (up to NP / 64 tiles at once); a fresh item's pixel is index arithmetic on the claimed tile -- no assignment phase, no
//     s_item array.  (First version: static interleaved shares, workgroup b owning tiles b, b + G, ... -- every workgroup then
//     ended on its own slowest tiles; handing the frame out from the head took the unguided wave from 0.887 to 0.775 ms and the
//     reference-default guided one from 1.59 to 1.45.  VSPG_WG2_TAIL keeps the split adjustable: the share of the frame that
//     comes from the head, in 64ths.)
//   * a finished path of a one-sample-per-pixel launch PARKS {L, ISG code} in a per-pixel buffer with one 16-byte fire-and-forget
//     store; the NEXT launch adds it to the film and the ISG statistics as it starts that pixel's new path (or k_film_resolve
//     does, when something else wants the film first: flush_parked_samples).  The film flush of k_render_wave_wg -- a
//     read-modify-write whose load latency sat between two workgroup barriers of every iteration -- is gone, and with it the
//     third barrier: [segment] barrier [vertex] barrier.  Per pixel and channel it is still the one IEEE addition `film += L` of
//     RGBFilm::AddSample, in the same order: same bits.
//   Measured (profiles/r03_*): per-section wave timers of k_render_wave_wg showed 28 % of the wave cycles in the three
//   barriers and ~20 % in assignment / flush; the SIMDs issued 44 % of the time (scripts/microbench/issue.hip prices).
// Multi-sample launches keep the no-return atomics at the point where a path ends (several samples of a pixel per launch).
enum { D_NFREE = 0, D_A0 = 2, D_A1 = 4, D_CURA = 6, D_BV = 8, D_BS = 10, D_CURB = 12, D_LNEXT = 14, D_COUNT = 15 };
// (isg_code, resolve_sample: vspg_wg3.h)
#ifndef VSPG_WG2_UNIT_SHIFT
#define VSPG_WG2_UNIT_SHIFT 6
#endif
template <class Medium, bool GUIDED, int NP, int kWgBlock, int kWgWavesPerSimd, bool TRAIN = false>
__global__ __launch_bounds__(kWgBlock, kWgWavesPerSimd) void k_render_wave_wg2(
    const DScene *__restrict__ Sp, float4 *__restrict__ film, float *__restrict__ isg_stats, const float *__restrict__ vsp_buf,
    int vsp_ready, int wave_end, int first_sample, int single_sample, PcgJump jump, unsigned int tiles_magic,
    unsigned int static_tiles, unsigned int *__restrict__ work_head, const float4 *__restrict__ prev_samples,
    float4 *__restrict__ wave_samples, unsigned long long *__restrict__ counters, TrainArgs train = TrainArgs{nullptr, nullptr, nullptr, nullptr, 0, 0}) {
    const DScene &S = *Sp;
    const int W = S.xres, H = S.yres;
    const int tilesX = (W + 7) >> 3, tilesY = (H + 7) >> 3;
    const unsigned n_tiles = (unsigned)(tilesX * tilesY);
    const int lane = threadIdx.x & 63;
    const int sample_step = S.shard_count > 1 ? S.shard_count : 1;
    // this workgroup's items: local item j = pixel (j & 63) of tile (j >> 6) * gridDim.x + blockIdx.x, over the first static_tiles
    // tiles; the REST of the frame is handed out tile by tile from a global head once a workgroup has started all of its own --
    // the workgroups finish their static shares at different times (a share's cost follows what its pixels see), and without the
    // shared tail every one of them ended on its own slowest tiles
    const unsigned n_static = static_tiles < n_tiles ? static_tiles : n_tiles;
    constexpr unsigned kUnitShift = VSPG_WG2_UNIT_SHIFT, kUnit = 1u << kUnitShift;
    const unsigned local_tiles = blockIdx.x < n_static ? (n_static - blockIdx.x + gridDim.x - 1) / gridDim.x : 0u;
    const unsigned local_total = local_tiles * 64u;
    reset_sibling_head(work_head);

    // FULL: a scene beyond rectangles and area lights -- triangles (BVH), spheres, infinite lights, non-uniform light samplers, medium
    // boundaries (round 4: these used to fall to the per-lane kernel).  The phases are the same; a segment may end on an interface
    // (LI_SKIP: the path goes round again without a vertex) and a path beyond the camera segment may still be at depth 0.
    constexpr bool FULL = !Medium::kSimpleScene;
    static_assert(!FULL || !GUIDED, "the workgroup kernel's guided vertex (vspg_guided_wg.h) is built for rectangle scenes");
    using LY = PoolLayout<GUIDED, Medium::kGrey, TRAIN, FULL>;
    constexpr int NF = LY::COUNT;
    static_assert(!TRAIN || GUIDED, "segment recording belongs to the guided instantiations");
    __shared__ float s_pool[NF * NP];
    __shared__ unsigned short s_listA[2][NP], s_listB[NP], s_free[2][NP];
    __shared__ unsigned int s_cnt[D_COUNT + 1];
    __shared__ unsigned int s_dyn[2], s_dyn_done[2];  // the tiles claimed for this iteration (first tile, items); the shared tail has run out
    const Pool P{s_pool, NP};
    static_assert(Medium::kSingleSegment, "k_render_wave_wg2 serves homogeneous media (grid media: the wavefront pipeline)");
    const Medium medium = MediumMaker<Medium>::make(S, nullptr);
    float *glds = nullptr;
    if constexpr (GUIDED && kKdLdsNodes > 0) {  // the upper levels of the two kd-trees (north star: "LDS-staged kd-tree nodes")
        __shared__ VspgKdNode s_kd[2][kKdLdsNodes > 0 ? kKdLdsNodes : 1];
        for (int f = 0; f < 2; ++f) {
            const int nl = S.field[f].n_nodes < kKdLdsNodes ? S.field[f].n_nodes : kKdLdsNodes;
            for (int i = threadIdx.x; i < nl; i += kWgBlock) s_kd[f][i] = S.field[f].nodes[i];
        }
        glds = reinterpret_cast<float *>(&s_kd[0][0]);
    }
    __shared__ unsigned int s_counters[CNT_COUNT];
    // counters: per lane in registers (six of them; the unguided instantiations have them to spare), per workgroup in LDS through
    // a ballot and one atomic per count in the guided instantiation, whose vertex phase needs every register it can get
    struct LaneCounters : PathCounters { uint32_t paths; VDEV void path() { paths++; } };
    using Rec = typename std::conditional<TRAIN, PathRecorder, NullRecorder>::type;
    typename std::conditional<GUIDED, WaveCountersT<Rec>, LaneCounters>::type pc = [&] {
        if constexpr (GUIDED) { WaveCountersT<Rec> c; c.c = s_counters; return c; }
        else { LaneCounters c; c.segments = c.volume_scatters = c.surface_hits = c.density_queries = c.shadow_rays = c.shadow_queries = c.paths = 0; return c; }
    }();
    // a18, training launches (one sample per pixel): a path's records go to ITS column of the wave's record buffer -- the
    // column of its work item, which the pixel names (tile-ordered items: vspg_render_wave sizes the buffer by them)
    const auto rec_bind = [&](int pxy) {
        if constexpr (TRAIN) {
            const unsigned px = (unsigned)pxy & 0xffffu, py = (unsigned)pxy >> 16;
            const unsigned item = ((py >> 3) * (unsigned)tilesX + (px >> 3)) * 64u + ((py & 7u) << 3) + (px & 7u);
            pc.rec.base = train.segbuf + item;
            pc.rec.stride = (int)train.n_items;
            pc.rec.max_seg = train_rec_capacity(S.prm.maxdepth);
            return item;
        } else {
            (void)pxy;
            return 0u;
        }
    };

    stage_scene_lds(S);
    if (threadIdx.x < CNT_COUNT) s_counters[threadIdx.x] = 0;
    if (threadIdx.x <= D_COUNT) s_cnt[threadIdx.x] = 0;
    if (threadIdx.x < 2) { s_dyn[threadIdx.x] = 0; s_dyn_done[threadIdx.x] = 0; }
    for (int i = threadIdx.x; i < NP; i += kWgBlock) s_free[0][i] = (unsigned short)i;
    __syncthreads();
    if (threadIdx.x == 0) s_cnt[D_NFREE] = NP;
    __syncthreads();

    // a path ends: its sample leaves the kernel
    auto emit = [&](int pxy, Spec Lraw, const IsgSample &isg) {
        const Spec L = finish_radiance(Lraw);
        const size_t pidx = (size_t)((unsigned)pxy >> 16) * W + (pxy & 0xffff);
        if (single_sample) {
            wave_samples[pidx] = make_float4(L.r, L.g, L.b, isg_code(isg));
        } else {
            film_add_sample(film + pidx, L);
            isg_add_sample_atomic(isg_stats + pidx * VSPG_ISG_STATS, L, isg);
        }
    };

    VSPG_PROF(PS_WG_TOTAL);
    for (int k = 0;; ++k) {
        const int par = k & 1, nxt = par ^ 1;
        const unsigned nFree = s_cnt[D_NFREE + par], nA0 = s_cnt[D_A0 + par], nA1 = s_cnt[D_A1 + par], lnext = s_cnt[D_LNEXT];
        const unsigned left = local_total - lnext;
        // the vertex-list counters of the PREVIOUS iteration (other parity) are free again: nobody reads them before the
        // segment phase of the next iteration pushes into them, two barriers from here
        if (threadIdx.x == 0) { s_cnt[D_BV + nxt] = 0; s_cnt[D_BS + nxt] = 0; s_cnt[D_CURB + nxt] = 0; s_dyn_done[nxt] = s_dyn_done[par]; }
        // own share started: whole tiles from the shared tail for the free slots (one returning atomic per claim).  Every thread
        // takes this branch or none: its condition reads values written at least one barrier ago (the done flag travels by parity).
        unsigned dynBase = 0, nDyn = 0;
        // (the head counts UNITS of 2^kUnitShift work items -- a tile, or half / a quarter of one: smaller units leave fewer slots idle)
        if (left == 0u && nFree >= kUnit && n_static < n_tiles && !s_dyn_done[par]) {
            if (threadIdx.x == 0) {
                const unsigned m = nFree >> kUnitShift, n_dyn = (n_tiles - n_static) << (6 - kUnitShift);
                const unsigned b = atomicAdd(work_head, m);
                const unsigned got = b < n_dyn ? (m < n_dyn - b ? m : n_dyn - b) : 0u;
                s_dyn[0] = (n_static << (6 - kUnitShift)) + b;
                s_dyn[1] = got << kUnitShift;
                if (got < m) s_dyn_done[nxt] = 1u;
            }
            __syncthreads();
            dynBase = s_dyn[0];
            nDyn = s_dyn[1];
        }
        const unsigned nFresh = left > 0u ? (nFree < left ? nFree : left) : nDyn;
        const unsigned nPrim = nFresh + nA0, nA = nPrim + nA1;
        if (nA == 0) break;  // nothing in flight and nothing left to start (free slots exist whenever nothing is in flight)
        const unsigned nSpare = nFree - nFresh;  // carried over to the next iteration's free list (the shared tail will want them)

        // ---- S: camera ray + primary segment for new paths, one secondary segment for the others ------
        while (true) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt[D_CURA + par], 64u);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= nA + nSpare) break;
            VSPG_PROF(PS_WG_A);
            const unsigned i = base + (unsigned)lane;
            bool toV = false, toS = false, restart = false, freed = false, skipped = false;
            int slot = 0;
            if (i >= nA && i < nA + nSpare) {  // a free slot no new path took this iteration: it stays free
                slot = s_free[par][nFresh + (i - nA)];
                freed = true;
            }
            if (i < nA) {
                Sampler sampler;
                PathState st;
                IsgSample isg;
                int ch = 0, pxy = 0;
                Vertex vx;
                bool alive = false, valid = true;
                int seg = LI_END;
                if (i < nPrim) {
                    int px, py, smp;
                    if (i < nFresh) {
                        slot = s_free[par][i];
                        const unsigned item = lnext + i;
                        const unsigned unit = dynBase + (i >> kUnitShift);  // (claimed items only)
                        const unsigned tile = left > 0u ? (item >> 6) * gridDim.x + blockIdx.x : unit >> (6 - kUnitShift);
                        const unsigned l = left > 0u ? item & 63u : ((unit & ((1u << (6 - kUnitShift)) - 1u)) << kUnitShift) + (i & (kUnit - 1u));
                        unsigned ty = tilesX == 1 ? tile : __umulhi(tile, tiles_magic);
                        unsigned tx = tile - ty * (unsigned)tilesX;
                        while (tx >= (unsigned)tilesX) { tx -= (unsigned)tilesX; ty++; }
                        px = (int)(tx * 8u + (l & 7u));
                        py = (int)(ty * 8u + (l >> 3));
                        pxy = px | (py << 16);
                        smp = first_sample;
                        valid = tile < (left > 0u ? n_static : n_tiles);
                        // the PREVIOUS one-sample launch parked this pixel's sample (vspg_render_wave: deferred resolve): it enters
                        // the film now, before this launch's sample of the pixel can (same order of additions as ever)
                        if (prev_samples != nullptr && valid && px < W && py < H) {
                            const size_t pidx = (size_t)py * W + px;
                            resolve_sample(prev_samples[pidx], film + pidx, isg_stats + pidx * VSPG_ISG_STATS);
                        }
                    } else {
                        slot = s_listA[par][i - nFresh];
                        pxy = P.i(LY::PIXEL, slot);
                        smp = P.i(LY::SAMPLE, slot);
                        px = pxy & 0xffff;
                        py = (int)((unsigned)pxy >> 16);
                    }
                    valid = valid && px < W && py < H && smp < wave_end;  // tile padding: the slot stays free
                    if (valid) {
                        if (single_sample)
                            start_path(S, vsp_buf, vsp_ready, px, py, jump, sampler, st, &ch, isg);
                        else
                            start_path(S, vsp_buf, vsp_ready, px, py, smp, sampler, st, &ch, isg);
                        P.i(LY::PIXEL, slot) = pxy;
                        P.i(LY::SAMPLE, slot) = smp;
                        if constexpr (TRAIN) { (void)rec_bind(pxy); pc.rec.reset(); }
                        seg = li_segment_a<Medium, GUIDED, SEG_PRIMARY>(S, medium, vsp_buf, vsp_ready, px, py, st, ch, sampler,
                                                                        isg, pc, vx);
                        alive = seg != LI_END;
                        if (alive) {
                            pool_store_full<GUIDED, Medium::kGrey, FULL>(P, slot, st, sampler, ch, isg, FL_LIVE | (seg == LI_VERTEX && vx.volume ? (uint32_t)FL_VX_VOLUME : 0u));
                            if (seg == LI_VERTEX) pool_store_vertex<GUIDED, Medium::kGrey>(P, slot, vx);
                            if constexpr (TRAIN) pool_store_rec<LY>(P, slot, pc.rec);
                        }
                    } else {
                        freed = true;
                    }
                } else {
                    slot = s_listA[par][NP - 1 - (int)(i - nPrim)];
                    const uint32_t fl = pool_load<GUIDED, Medium::kGrey, FULL>(P, slot, S, st, sampler, &ch, isg);
                    pxy = P.i(LY::PIXEL, slot);
                    const int px = pxy & 0xffff, py = (int)((unsigned)pxy >> 16);
                    if constexpr (TRAIN) { (void)rec_bind(pxy); pool_load_rec<LY>(P, slot, pc.rec); }
                    // (full scenes: a path that crossed a medium boundary on its camera segment is still at depth 0 here)
                    seg = li_segment_a<Medium, GUIDED, FULL ? SEG_ANY : SEG_SECONDARY>(S, medium, vsp_buf, vsp_ready, px, py, st, ch, sampler,
                                                                                      isg, pc, vx);
                    alive = seg != LI_END;
                    if (seg == LI_SKIP) {
                        pool_store_full<GUIDED, Medium::kGrey, FULL>(P, slot, st, sampler, ch, isg, fl & (FL_LIVE | FL_GS_SCATTER | FL_GS_FIELD));
                    } else if (alive) {
                        pool_store_a<Medium::kGrey, GUIDED, FULL>(P, slot, st, sampler, ch, isg, vx, fl & (FL_LIVE | FL_GS_SCATTER | FL_GS_FIELD | FL_INMED));
                        if constexpr (TRAIN) pool_store_rec<LY>(P, slot, pc.rec);
                    }
                }
                if (seg == LI_SKIP) {        // a medium boundary was crossed (:399-404): no vertex, the next segment starts behind it
                    skipped = true;
                } else if (alive) {
                    toV = vx.volume;
                    toS = !vx.volume;
                } else if (valid) {
                    emit(pxy, st.L, isg);
                    if constexpr (TRAIN) train.seg_count[rec_bind(pxy)] = pc.rec.n;  // PropagateSamples (:627) follows in k_propagate
                    pc.path();
                    const int s2 = P.i(LY::SAMPLE, slot) + sample_step;
                    P.i(LY::SAMPLE, slot) = s2;
                    restart = s2 < wave_end;
                    freed = !restart;
                }
            }
            list_push(toV, slot, s_listB, &s_cnt[D_BV + par]);
            list_push_back(toS, slot, s_listB + NP - 1, &s_cnt[D_BS + par]);
            list_push(restart, slot, s_listA[nxt], &s_cnt[D_A0 + nxt]);
            list_push(freed, slot, s_free[nxt], &s_cnt[D_NFREE + nxt]);
            if constexpr (FULL) list_push_back(skipped, slot, s_listA[nxt] + NP - 1, &s_cnt[D_A1 + nxt]);  // joins the paths the vertex phase sends on
        }
        { VSPG_PROF(PS_WG_BAR_A); __syncthreads(); }
        // the segment phase's inputs are consumed (every wave read the counts before it entered the phase)
        if (threadIdx.x == 0) {
            s_cnt[D_LNEXT] = lnext + (left > 0u ? nFresh : 0u);
            s_cnt[D_NFREE + par] = 0; s_cnt[D_A0 + par] = 0; s_cnt[D_A1 + par] = 0; s_cnt[D_CURA + par] = 0;
        }

        // ---- V: vertex processing (NEE, Russian roulette, new direction) ----------------------------
        const unsigned nBV = s_cnt[D_BV + par], nB = nBV + s_cnt[D_BS + par];
        while (true) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&s_cnt[D_CURB + par], 64u);
            base = __builtin_amdgcn_readfirstlane(base);
            if (base >= nB) break;
            VSPG_PROF(PS_WG_B);
            const unsigned i = base + (unsigned)lane;
            bool cont = false, restart = false, freed = false;
            int slot = 0;
            if (i < nB) {
                slot = i < nBV ? s_listB[i] : s_listB[NP - 1 - (int)(i - nBV)];
                Sampler sampler;
                PathState st;
                IsgSample isg;
                int ch;
                bool alive;
                if constexpr (GUIDED) {  // the guided vertex works on the pool record directly (vspg_guided_wg.h)
                    const uint32_t fl = P.u(LY::FLAGS, slot);
                    if constexpr (TRAIN) { (void)rec_bind(P.i(LY::PIXEL, slot)); pool_load_rec<LY>(P, slot, pc.rec); }
                    alive = li_vertex_guided_wg<Medium>(S, medium, P, slot, fl, pc, reinterpret_cast<const VspgKdNode *>(glds), &st.L, &isg);
                    if constexpr (TRAIN) { if (alive) pool_store_rec<LY>(P, slot, pc.rec); }
                    if (!alive) {
                        isg.valid = (fl & FL_ISG_VALID) != 0;
                        isg.surface_event = (fl & FL_ISG_SURF) != 0;
                        isg.vsp_used = P.f(LY::VSP, slot);  // (depth >= 1 at a vertex: the slot holds isg.vsp_used)
                    }
                } else {
                    const uint32_t fl = pool_load<GUIDED, Medium::kGrey, FULL>(P, slot, S, st, sampler, &ch, isg);
                    const Vertex vx = pool_load_vertex<GUIDED, Medium::kGrey, FULL>(P, slot, fl);
                    alive = li_segment_b<Medium, GUIDED, GUIDED>(S, medium, st, ch, sampler, pc, vx, glds, kWgBlock);
                    if (alive) pool_store_full<GUIDED, Medium::kGrey, FULL>(P, slot, st, sampler, ch, isg, FL_LIVE);
                }
                if (alive) {
                    cont = true;
                } else {
                    emit(P.i(LY::PIXEL, slot), st.L, isg);
                    if constexpr (TRAIN) train.seg_count[rec_bind(P.i(LY::PIXEL, slot))] = pc.rec.n;
                    pc.path();
                    const int s2 = P.i(LY::SAMPLE, slot) + sample_step;
                    P.i(LY::SAMPLE, slot) = s2;
                    restart = s2 < wave_end;
                    freed = !restart;
                }
            }
            list_push_back(cont, slot, s_listA[nxt] + NP - 1, &s_cnt[D_A1 + nxt]);
            list_push(restart, slot, s_listA[nxt], &s_cnt[D_A0 + nxt]);
            list_push(freed, slot, s_free[nxt], &s_cnt[D_NFREE + nxt]);
        }
        { VSPG_PROF(PS_WG_BAR_B); __syncthreads(); }
    }
    if constexpr (!GUIDED) {
        atomicAdd(&s_counters[CNT_PATHS], pc.paths); atomicAdd(&s_counters[CNT_SEGMENTS], pc.segments);
        atomicAdd(&s_counters[CNT_VOLUME_SCATTERS], pc.volume_scatters); atomicAdd(&s_counters[CNT_SURFACE_HITS], pc.surface_hits);
        atomicAdd(&s_counters[CNT_DENSITY_QUERIES], pc.density_queries); atomicAdd(&s_counters[CNT_SHADOW_RAYS], pc.shadow_rays);
    atomicAdd(&s_counters[CNT_SHADOW_QUERIES], pc.shadow_queries);
    }
    __syncthreads();
    if (threadIdx.x < CNT_COUNT) atomicAdd(&counters[threadIdx.x], (unsigned long long)s_counters[threadIdx.x]);
}

// film += the launch's sample buffer; ISG statistics likewise (the same read-modify-write forms the in-kernel flush used)
__global__ __launch_bounds__(kBlock) void k_film_resolve(size_t npix, const float4 *__restrict__ wave_samples, float4 *__restrict__ film,
                                                         float *__restrict__ isg_stats) {
    const size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= npix) return;
    resolve_sample(wave_samples[i], film + i, isg_stats + i * VSPG_ISG_STATS);
}

// (k_trace_paths: vspg_trace.h)
// vspg_ray_batch (include/vspg.h): Integrator::Intersect, then Interaction::SpawnRay / SpawnRayTo from the hit and Intersect /
// IntersectP of the spawned ray -- the full-scene intersection code of the path kernels (rectangles, BVH triangles, spheres)
__global__ __launch_bounds__(kBlock) void k_ray_batch(const DScene *__restrict__ Sp, int n, const VspgRayQuery *__restrict__ q, VspgRayResult *__restrict__ out) {
    const DScene &S = *Sp;
    stage_scene_lds(S);
    __syncthreads();
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    VspgRayResult o;
    o.hit = 0; o.prim = 0; o.t = 0.f; o.hit2 = 0; o.any2 = 0; o.t2 = 0.f;
    for (int k = 0; k < 3; ++k) o.p[k] = o.n[k] = o.o2[k] = o.d2[k] = 0.f;
    const VspgRayQuery Q = q[i];
    const Isect si = scene_intersect<true>(S, ld3(Q.o), ld3(Q.d), Q.tMax);
    if (si.hit) {
        const P3i pi = surf_pi<true>(S, si.quad, si.p);
        const V3 pm = pi.mid();
        o.hit = 1;
        o.prim = is_sphere(si.quad) ? 2000000 + sphere_of(si.quad) : (is_tri(si.quad) ? 1000000 + S.tris[tri_of(si.quad)].id : si.quad);
        o.t = si.t;
        o.p[0] = pm.x; o.p[1] = pm.y; o.p[2] = pm.z;
        o.n[0] = si.n.x; o.n[1] = si.n.y; o.n[2] = si.n.z;
        if (Q.mode != 0) {
            const V3 w = ld3(Q.w);
            const V3 d2 = Q.mode == 1 ? w : w - pm;
            const V3 o2 = offset_ray_origin(pi, si.n, d2);
            o.o2[0] = o2.x; o.o2[1] = o2.y; o.o2[2] = o2.z;
            o.d2[0] = d2.x; o.d2[1] = d2.y; o.d2[2] = d2.z;
            const Isect s2 = scene_intersect<true>(S, o2, d2, Q.tMax2);
            o.hit2 = s2.hit ? 1 : 0;
            o.t2 = s2.hit ? s2.t : 0.f;
            o.any2 = scene_intersect_any<true>(S, o2, d2, Q.tMax2) ? 1 : 0;
        }
    }
    out[i] = o;
}

template <class Medium>
__global__ __launch_bounds__(kBlock) void k_tmaj_batch(const DScene *__restrict__ Sp, int variant, int n,
                                                       const VspgTmajQuery *__restrict__ q, VspgTmajResult *__restrict__ out) {
    const DScene &S = *Sp;
    int i = blockIdx.x * kBlock + threadIdx.x;
    stage_scene_lds(S);
    __syncthreads();
    if (i >= n) return;
    const Medium medium = MediumMaker<Medium>::make(S, S.majorant);
    VspgTmajQuery Q = q[i];
    VspgTmajResult R;
    memset(&R, 0, sizeof R);
    R.last_t = -1.f;
    R.majorant_scale = 1.f;
    Rng rng;
    rng.set_sequence(hash_float(Q.rng_a), hash_float(Q.rng_b));
    const int ch = Q.channel;
    V3 ro = ld3(Q.o), rd = ld3(Q.d), rdn = normalize(rd);
    bool guide = Q.vsp >= 0.f;
    float vsp = guide ? fmax_(fmin_(Q.vsp, 0.999f), 0.001f) : Q.vsp;
    int ncb = 0;
    float sum = 0, last_t = -1.f;
    V3 lastp = mk(0, 0, 0);
    auto rec = [&](V3 p, const MediumProps &mp, Spec sigma_maj, Spec, bool) {
        ncb++;
        lastp = p;
        last_t = dot(p - ro, rdn);
        Spec sigma_t = mp.sigma_t;
        sum += ch_of(sigma_t, ch) / ch_of(sigma_maj, ch);
        if (Q.stop_after > 0 && ncb >= Q.stop_after) return false;
        return true;
    };
    Spec T = sp(1.f), rf = sp(1.f);
    if (variant == VSPG_TMAJ_PLAIN) {
        T = sample_T_maj(medium, ro, rd, Q.tMax, Q.u, rng, ch, rec);
    } else if (variant == VSPG_TMAJ_OPTICAL_DEPTH) {
        T = sample_T_maj_ods(medium, ro, rd, Q.tMax, Q.u, rng, ch, guide, vsp, S.prm.vspmisratio,
                             S.prm.vspsamplingmethod == VSPG_VSP_NDS, &rf, rec);
    } else {
        float vrc = vsp, ms = 1.f;
        T = sample_T_maj_resampling(medium, ro, rd, Q.tMax, Q.u, rng, ch, guide, vsp, &vrc, &ms, rec);
        R.vrc = vrc;
        R.majorant_scale = ms;
    }
    R.T_maj[0] = T.r; R.T_maj[1] = T.g; R.T_maj[2] = T.b;
    R.r_u_factor[0] = rf.r; R.r_u_factor[1] = rf.g; R.r_u_factor[2] = rf.b;
    R.last_t = last_t;
    R.last_p[0] = lastp.x; R.last_p[1] = lastp.y; R.last_p[2] = lastp.z;
    R.n_callbacks = ncb;
    R.sum_sigt_over_maj = sum;
    out[i] = R;
}

__global__ __launch_bounds__(kBlock) void k_primitives(int n, const float *__restrict__ f, const float *__restrict__ g,
                                                       uint64_t *__restrict__ hash, uint32_t *__restrict__ rng_u32,
                                                       float *__restrict__ fexp) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    hash[i] = hash_float(f[i]);
    Rng r;
    r.set_sequence(hash_float(f[i]), hash_float(g[i]));
    rng_u32[i] = r.u32();
    fexp[i] = fast_exp(f[i]);
}

__global__ __launch_bounds__(kBlock) void k_guiding_query(const DScene *__restrict__ Sp, int is_volume, float g, int n,
                                                          const float *__restrict__ p, const float *__restrict__ a,
                                                          const float *__restrict__ wi, const float *__restrict__ u,
                                                          int32_t *__restrict__ ok, float *__restrict__ pdf,
                                                          float *__restrict__ inc, float *__restrict__ vsp,
                                                          float *__restrict__ ws, float *__restrict__ pdfs) {
    const DScene &S = *Sp;
    int i = blockIdx.x * kBlock + threadIdx.x;
    vspg_libm::stage_logf_tab_lds();
    __syncthreads();
    if (i >= n) return;
    float *glds = guide_lds();
    V3 pp = ld3(p + 3 * i), aa = ld3(a + 3 * i), w = ld3(wi + 3 * i);
    GDist d = is_volume ? gdist_init_volume(S.field, pp, aa, g, glds, kBlock) : gdist_init_surface(S.field, pp, aa, glds, kBlock);
    ok[i] = d.ok ? 1 : 0;
    pdf[i] = inc[i] = pdfs[i] = 0;
    vsp[i] = -1;
    ws[3 * i] = ws[3 * i + 1] = ws[3 * i + 2] = 0;
    if (!d.ok) return;
    pdf[i] = gdist_pdf(d, w);
    inc[i] = gdist_incoming_pdf(S.field, d, w);
    vsp[i] = gdist_vsp(S.field, d.field, d.region, d, w);
    V3 s;
    pdfs[i] = gdist_sample(d, u[2 * i], u[2 * i + 1], &s);
    ws[3 * i] = s.x; ws[3 * i + 1] = s.y; ws[3 * i + 2] = s.z;
}

__global__ __launch_bounds__(kBlock) void k_libm(int n, const float *__restrict__ x, float *__restrict__ lo,
                                                 float *__restrict__ so, float *__restrict__ co) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    vspg_libm::stage_logf_tab_lds();
    __syncthreads();
    if (i >= n) return;
    lo[i] = logf_(x[i]);
    so[i] = sinf_(x[i]);
    co[i] = cosf_(x[i]);
}
__global__ __launch_bounds__(kBlock) void k_libm_powf(int n, const float *__restrict__ x, const float *__restrict__ y, float *__restrict__ out) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = vspg_libm::powf_host_exact(x[i], y[i]);
}
__global__ __launch_bounds__(kBlock) void k_blackbody(int n, const float *__restrict__ u, const float *__restrict__ T, float *__restrict__ out) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const Spec s = blackbody_sample(T[i], u[i]);
    for (int k = 0; k < 3; ++k) out[6 * i + k] = sample_visible_wavelength(u[i], k);
    out[6 * i + 3] = s.r; out[6 * i + 4] = s.g; out[6 * i + 5] = s.b;
}
__global__ __launch_bounds__(kBlock) void k_libm_log1m(int n, const float *__restrict__ x, float *__restrict__ out) {
    int i = blockIdx.x * kBlock + threadIdx.x;
    vspg_libm::stage_log_tab_lds();
    __syncthreads();
    if (i >= n) return;
    out[i] = neg_log1m_d(x[i]);
}

// ---- octet bricks (DScene::brick_index / octets) built on the device from the uploaded raw samples -------------------
// brick (bx, by, bz) covers the octets of base voxels [8b - 1, 8b + 6]^3, i.e. raw voxels [8b - 1, 8b + 7]^3
__device__ __forceinline__ float raw_at(const float *d, int nx, int ny, int nz, int x, int y, int z) {
    if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) return 0.f;
    return d[((size_t)z * ny + y) * nx + x];
}
__global__ __launch_bounds__(kBlock) void k_brick_flags(const float *__restrict__ d, int nx, int ny, int nz, int bnx, int bny,
                                                        int32_t *__restrict__ flags) {
    const int b = blockIdx.x;
    const int bx = b % bnx, by = (b / bnx) % bny, bz = b / (bnx * bny);
    __shared__ int s_any;
    if (threadIdx.x == 0) s_any = 0;
    __syncthreads();
    bool any = false;
    for (int i = threadIdx.x; i < 9 * 9 * 9; i += kBlock) {
        const int x = 8 * bx - 1 + i % 9, y = 8 * by - 1 + (i / 9) % 9, z = 8 * bz - 1 + i / 81;
        any = any || raw_at(d, nx, ny, nz, x, y, z) != 0.f;
    }
    if (any) s_any = 1;
    __syncthreads();
    if (threadIdx.x == 0) flags[b] = s_any;
}
__global__ __launch_bounds__(512) void k_brick_fill(const float *__restrict__ d, int nx, int ny, int nz, int bnx, int bny,
                                                    const int32_t *__restrict__ active /* brick number of slot */, float4 *__restrict__ octets) {
    const int b = active[blockIdx.x];
    const int bx = b % bnx, by = (b / bnx) % bny, bz = b / (bnx * bny);
    const int l = threadIdx.x, lx = l & 7, ly = (l >> 3) & 7, lz = l >> 6;
    const int ix = 8 * bx + lx - 1, iy = 8 * by + ly - 1, iz = 8 * bz + lz - 1;  // base voxel of this octet
    float4 lo, hi;
    lo.x = raw_at(d, nx, ny, nz, ix, iy, iz);         lo.y = raw_at(d, nx, ny, nz, ix + 1, iy, iz);
    lo.z = raw_at(d, nx, ny, nz, ix, iy + 1, iz);     lo.w = raw_at(d, nx, ny, nz, ix + 1, iy + 1, iz);
    hi.x = raw_at(d, nx, ny, nz, ix, iy, iz + 1);     hi.y = raw_at(d, nx, ny, nz, ix + 1, iy, iz + 1);
    hi.z = raw_at(d, nx, ny, nz, ix, iy + 1, iz + 1); hi.w = raw_at(d, nx, ny, nz, ix + 1, iy + 1, iz + 1);
    float4 *q = octets + ((size_t)blockIdx.x * 512u + (size_t)l) * 2u;
    q[0] = lo;
    q[1] = hi;
}

// ImageSpaceGuidingBuffer::Update stand-in: 5x5 box filter over the sufficient statistics,
// then the contribution / variance criterion (own design, unpinned).
constexpr int kIsgRadius = 2;
// One workgroup = one 16 x 16 pixel tile: the five statistics of the tile and its two-pixel halo are staged in LDS with one
// coalesced pass, and the 25 taps of a pixel read LDS (round 5; as 50 global loads per pixel the kernel took 113 us per 1080p
// update against ~15 us of HBM time).  The sum keeps its order -- rows outer, columns inner, taps outside the image skipped --
// so the buffer keeps its bits.
constexpr int kIsgTile = 16;
static_assert(kIsgTile * kIsgTile == kBlock, "one thread per pixel of the tile");
__global__ __launch_bounds__(kBlock) void k_isg_update(int W, int H, int criterion, const float *__restrict__ stats,
                                                       float *__restrict__ vsp /* null: leave the VSP buffer alone */,
                                                       float *__restrict__ contrib /* null: no contribution estimate */) {
    constexpr int TW = kIsgTile + 2 * kIsgRadius;
    __shared__ float s_t[5][TW * TW];
    const int tx0 = (int)blockIdx.x * kIsgTile - kIsgRadius, ty0 = (int)blockIdx.y * kIsgTile - kIsgRadius;
    for (int j = threadIdx.x; j < TW * TW; j += kBlock) {
        const int xx = tx0 + j % TW, yy = ty0 + j / TW;
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f);
        float s4 = 0.f;
        if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
            const float *st = stats + ((size_t)yy * W + xx) * VSPG_ISG_STATS;
            s0 = *reinterpret_cast<const float4 *>(st);
            s4 = st[4];
        }
        s_t[0][j] = s0.x; s_t[1][j] = s0.y; s_t[2][j] = s0.z; s_t[3][j] = s0.w; s_t[4][j] = s4;
    }
    __syncthreads();
    const int lx = threadIdx.x % kIsgTile, ly = threadIdx.x / kIsgTile;
    const int x = (int)blockIdx.x * kIsgTile + lx, y = (int)blockIdx.y * kIsgTile + ly;
    if (x >= W || y >= H) return;
    const int i = y * W + x;
    float a[5] = {0, 0, 0, 0, 0};
    for (int dy = -kIsgRadius; dy <= kIsgRadius; ++dy)
        for (int dx = -kIsgRadius; dx <= kIsgRadius; ++dx) {
            int xx = x + dx, yy = y + dy;
            if (xx < 0 || xx >= W || yy < 0 || yy >= H) continue;
            const int j = (ly + kIsgRadius + dy) * TW + (lx + kIsgRadius + dx);
            a[0] += s_t[0][j]; a[1] += s_t[1][j]; a[2] += s_t[2][j]; a[3] += s_t[3][j]; a[4] += s_t[4][j];
        }
    float r = -1.f;
    if (a[0] > 0) {
        float v, s;
        if (criterion == VSPG_VSP_VARIANCE) {
            v = __builtin_sqrtf(a[3] / a[0]);
            s = __builtin_sqrtf(a[4] / a[0]);
        } else {
            v = a[1] / a[0];
            s = a[2] / a[0];
        }
        if (v + s > 0) r = v / (v + s);
    }
    if (vsp) vsp[i] = r;
    // contribution estimate for guided RR (own stand-in): filtered mean of the samples' average radiance, 0 = none
    if (contrib) contrib[i] = a[0] > 0 ? (a[1] + a[2]) / a[0] : 0.f;
}

}  // namespace

// =======================================================================================
// host side
// =======================================================================================
static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess)                                                                          \
            return fail(VSPG_EHIP, std::string(#expr) + ": " + hipGetErrorName(e_) + " (" + hipGetErrorString(e_) + ")"); \
    } while (0)

struct VspgRenderer {
    VspgScene scene;
    VspgIntegratorParams prm;
    VspgRenderConfig cfg;
    int arith = VSPG_ARITH_EXACT;   // vspg_renderer_set_arithmetic (vspg_arith.h): which instantiations the path kernels are launched from
    std::string kernel_name_buf, kernel_name_buf2;
    DScene hscene;
    DScene *dscene = nullptr;
    float4 *film = nullptr;
    float *isg_stats = nullptr;
    float *contrib = nullptr;     // guided RR: contribution estimate, W*H
    float *tr_rgb = nullptr;      // TrBuffer: W*H*3 running mean, W*H sample counts
    int32_t *tr_spp = nullptr;
    float *vsp = nullptr;
    unsigned long long *counters = nullptr;
    unsigned int *work_head = nullptr;   // two counters, used by alternate launches (reset_sibling_head)
    unsigned int *work_head8_raw = nullptr, *work_head8 = nullptr;  // k_render_wave_wg3's: two sets of kWg3Heads cursors, 2 KB-aligned
    unsigned int head8_parity = 0;
    unsigned int head_parity = 0;
    VspgKdNode *fnodes[2] = {nullptr, nullptr};        // guiding fields (device copies)
    VspgFieldRegion *fregions[2] = {nullptr, nullptr};
    float *faux[2] = {nullptr, nullptr};               // DField::aux
    float4 *flobes[2] = {nullptr, nullptr};            // DField::lobes
    bool field_set = false;
    bool medium_grey = false;   // homogeneous medium with bitwise-grey sigma_a, sigma_s, Le
    bool surfaces_grey = false; // every rectangle's (clamped) Kd bitwise grey
    bool null_zero = false;     // homogeneous medium whose ClampZero((sigma_t - sigma_a) - sigma_s) is exactly 0 per channel
    // a18: on-device training of the guiding field
    bool training = false;
    int field_iteration = 0;
    float *segbuf = nullptr;                  // segment records of a training wave, one column per work item
    int *seg_count = nullptr;                 // records per work item
    size_t segbuf_items = 0;
    VspgTrainSample *samples = nullptr;
    unsigned long long sample_capacity = 0;
    unsigned long long *train_counters = nullptr;  // [0] samples, [1] zero-valued, [2..3] spare
    RegionStats *rstats[2] = {nullptr, nullptr};
    float *train_acc = nullptr;               // kTrainCapRegions x kStatFloats accumulators of one pass
    float *train_sumw = nullptr;        // [0] sum of sample weights, [1] sample count (as a float, for the cross-rank sum)
    VspgExchangeFn exchange = nullptr;  // sharded training: sums a device buffer over the ranks (vspg_renderer_set_exchange)
    void *exchange_user = nullptr;
    int *train_reg = nullptr;                 // region of every sample of the batch (field being updated)
    unsigned int *train_order = nullptr;      // sample indices sorted by region
    unsigned int *train_hist = nullptr, *train_cursor = nullptr, *train_nsorted = nullptr;
    int *train_nsplit = nullptr;              // [0] regions the split pass of the running update created (both fields), [1] a region without lobes exists
    float *density = nullptr;   // GridMedium density samples (raw; released once the octet bricks are built)
    int32_t *brick_index = nullptr;
    float4 *octets = nullptr;
    size_t n_bricks = 0;
    float *le_scale = nullptr;  // emissive GridMedium: LeScale grid
    float *temperature = nullptr;  // temperature grid (raw samples)
    float *majorant = nullptr;  // 16^3 majorant grid
    DTri *tris = nullptr;          // triangle soup in BVH leaf order + the BVH (depth-first, skip links)
    DBvh4Node *bvh = nullptr;
    // wavefront pipeline (vspg_wavefront.h): path SoA, lists and per-iteration control blocks, allocated at first use
    float *wf_pool = nullptr;
    // k_render_wave_wg2: one {L, ISG code} per pixel of a one-sample launch.  The samples of launch w enter the film at the start of
    // launch w + 1 (inside the kernel, as each pixel's new path begins) or, when anything else wants the film or the statistics
    // first, through k_film_resolve (flush_parked_samples): two buffers, `ws_parked` says the other one holds unresolved samples.
    float4 *wave_samples[2] = {nullptr, nullptr};
    int ws_cur = 0;
    bool ws_parked = false;
    hipStream_t ws_stream = nullptr;  // the stream of the launch that parked them
    hipEvent_t ws_event = nullptr;    // recorded behind that launch: whoever touches the parked samples on another stream waits on it
    unsigned int *wf_lists = nullptr;   // 4 x n_items: active (even / odd iterations), walk, shadow
    hipStream_t wf_stream2 = nullptr;   // the shadow walks' stream (wf_render_pass)
    hipEvent_t wf_ev_vertex = nullptr, wf_ev_shadow = nullptr;
    WfIter *wf_iters = nullptr;
    size_t wf_items = 0;
    int num_cus = 0;
    int vsp_ready = 0;
    bool vsp_loaded = false;  // ImageSpaceGuidingBuffer(fileName): no further updates
    int wave_counter = 0, buffer_wave = 0;
    size_t npix = 0;
};

// ---- host float helpers for scene preprocessing (same formulas as the kernels use) ----
namespace hostmath {
struct H3 { float x, y, z; };
static H3 ld(const float *p) { return H3{p[0], p[1], p[2]}; }
static void stv(float *d, H3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }
static H3 add(H3 a, H3 b) { return H3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static H3 absv(H3 a) { return H3{std::fabs(a.x), std::fabs(a.y), std::fabs(a.z)}; }
static float dop(float a, float b, float c, float d) {
    float cd = c * d;
    float r = std::fmaf(a, b, -cd);
    float e = std::fmaf(-c, d, cd);
    return r + e;
}
static H3 crossv(H3 v, H3 w) { return H3{dop(v.y, w.z, v.z, w.y), dop(v.z, w.x, v.x, w.z), dop(v.x, w.y, v.y, w.x)}; }
static float len2v(H3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
static H3 normv(H3 a) { float l = std::sqrt(len2v(a)); return H3{a.x / l, a.y / l, a.z / l}; }
}  // namespace hostmath

// (accMult, G) of RNG::Advance(delta) (src/pbrt/util/rng.h:137-150), see PcgJump
static PcgJump pcg_jump(unsigned long long delta) {
    unsigned long long curMult = 0x5851f42d4c957f2dULL, curPlus = 1u, accMult = 1u, accPlus = 0u;
    while (delta > 0) {
        if (delta & 1) {
            accMult *= curMult;
            accPlus = accPlus * curMult + curPlus;
        }
        curPlus = (curMult + 1) * curPlus;
        curMult *= curMult;
        delta /= 2;
    }
    return PcgJump{accMult, accPlus};
}

// ---- light samplers: host-side build (lightsamplers.cpp:76-99 PowerLightSampler, :108-262 BVHLightSampler::buildBVH) ----------
// The reference's construction restated: LightBounds of every bounded light (DiffuseAreaLight::Bounds on a rectangle), the
// modified-SAH split over 12 buckets per axis, CompactLightBounds' quantisation.  The CPU checker states the same thing in C
// (oracle/vspg_oracle.c, "light samplers"); both run the same float operations on the same libm, so the trees are equal.
namespace lsb {
struct v3 { float x, y, z; };
static inline v3 V3(float x, float y, float z) { return v3{x, y, z}; }
static inline v3 v_add(v3 a, v3 b) { return v3{a.x + b.x, a.y + b.y, a.z + b.z}; }
static inline v3 v_sub(v3 a, v3 b) { return v3{a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline v3 v_scale(v3 a, float s) { return v3{a.x * s, a.y * s, a.z * s}; }
static inline v3 v_neg(v3 a) { return v3{-a.x, -a.y, -a.z}; }
static inline float v_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float v_len2(v3 a) { return v_dot(a, a); }
static inline float v_len(v3 a) { return sqrtf(v_len2(a)); }
static inline v3 v_normalize(v3 a) { float l = v_len(a); return v3{a.x / l, a.y / l, a.z / l}; }
static inline float dop(float a, float b, float c, float d) { return hostmath::dop(a, b, c, d); }
static inline v3 v_cross(v3 v, v3 w) { return v3{dop(v.y, w.z, v.z, w.y), dop(v.z, w.x, v.x, w.z), dop(v.x, w.y, v.y, w.x)}; }
static inline float sqr(float x) { return x * x; }
static inline float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }
static inline float safe_sqrt(float x) { return sqrtf(x > 0 ? x : 0.f); }
constexpr float PI_F = 3.14159265358979323846f;
constexpr int LBVH_MAX_LIGHTS = kMaxLights;
struct lightbounds_t { v3 bmin, bmax; float phi; v3 w; float cosTheta_o, cosTheta_e; int twoSided; };
static float safe_asin_f(float x) { return asinf(clampf(x, -1, 1)); }
static float safe_acos_f(float x) { return acosf(clampf(x, -1, 1)); }
static float angle_between(v3 a, v3 b) { /* vecmath.h:972-977 */
    if (v_dot(a, b) < 0) return PI_F - 2 * safe_asin_f(v_len(v_add(a, b)) / 2);
    return 2 * safe_asin_f(v_len(v_sub(b, a)) / 2);
}
typedef struct { v3 w; float cosTheta; } dircone_t; /* DirectionCone (vecmath.h:1785-1808); cosTheta == INFINITY: empty */
static dircone_t dircone(v3 w, float c) { dircone_t d; d.w = v_normalize(w); d.cosTheta = c; return d; }
static dircone_t dircone_union(dircone_t a, dircone_t b) { /* vecmath.cpp:56-83 */
    if (std::isinf(a.cosTheta)) return b;
    if (std::isinf(b.cosTheta)) return a;
    float theta_a = safe_acos_f(a.cosTheta), theta_b = safe_acos_f(b.cosTheta);
    float theta_d = angle_between(a.w, b.w);
    if (fminf(theta_d + theta_b, PI_F) <= theta_a) return a;
    if (fminf(theta_d + theta_a, PI_F) <= theta_b) return b;
    float theta_o = (theta_a + theta_d + theta_b) / 2;
    if (theta_o >= PI_F) return dircone(V3(0, 0, 1), -1);
    float theta_r = theta_o - theta_a;
    v3 wr = v_cross(a.w, b.w);
    if (v_len2(wr) == 0) return dircone(V3(0, 0, 1), -1);
    /* Rotate(Degrees(theta_r), wr)(a.w) (transform.h:220-247): sin / cos of Radians(Degrees(theta_r)) */
    float deg = (180 / PI_F) * theta_r, rad = (PI_F / 180) * deg;
    float sinT = sinf(rad), cosT = cosf(rad);
    v3 ax = v_normalize(wr);
    float m[3][3];
    m[0][0] = ax.x * ax.x + (1 - ax.x * ax.x) * cosT; m[0][1] = ax.x * ax.y * (1 - cosT) - ax.z * sinT; m[0][2] = ax.x * ax.z * (1 - cosT) + ax.y * sinT;
    m[1][0] = ax.x * ax.y * (1 - cosT) + ax.z * sinT; m[1][1] = ax.y * ax.y + (1 - ax.y * ax.y) * cosT; m[1][2] = ax.y * ax.z * (1 - cosT) - ax.x * sinT;
    m[2][0] = ax.x * ax.z * (1 - cosT) - ax.y * sinT; m[2][1] = ax.y * ax.z * (1 - cosT) + ax.x * sinT; m[2][2] = ax.z * ax.z + (1 - ax.z * ax.z) * cosT;
    v3 w = V3(m[0][0] * a.w.x + m[0][1] * a.w.y + m[0][2] * a.w.z, m[1][0] * a.w.x + m[1][1] * a.w.y + m[1][2] * a.w.z,
              m[2][0] * a.w.x + m[2][1] * a.w.y + m[2][2] * a.w.z); /* Transform::operator()(Vector3f), transform.h:351-356 */
    return dircone(w, cosf(theta_o));
}
static lightbounds_t lb_empty(void) { lightbounds_t b; std::memset(&b, 0, sizeof b); b.bmin = V3(INFINITY, INFINITY, INFINITY); b.bmax = V3(-INFINITY, -INFINITY, -INFINITY); return b; }
static v3 v_min3(v3 a, v3 b) { return V3(fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)); }
static v3 v_max3(v3 a, v3 b) { return V3(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)); }
static lightbounds_t lb_union(lightbounds_t a, lightbounds_t b) { /* lights.h:137-153 */
    if (a.phi == 0) return b;
    if (b.phi == 0) return a;
    dircone_t ca, cb; ca.w = a.w; ca.cosTheta = a.cosTheta_o; cb.w = b.w; cb.cosTheta = b.cosTheta_o;
    ca.w = v_normalize(ca.w); cb.w = v_normalize(cb.w); /* DirectionCone(w, cosTheta) normalises */
    dircone_t cone = dircone_union(ca, cb);
    lightbounds_t r;
    r.bmin = v_min3(a.bmin, b.bmin); r.bmax = v_max3(a.bmax, b.bmax);
    r.w = v_normalize(cone.w);
    r.phi = a.phi + b.phi;
    r.cosTheta_o = cone.cosTheta;
    r.cosTheta_e = fminf(a.cosTheta_e, b.cosTheta_e);
    r.twoSided = a.twoSided | b.twoSided;
    return r;
}
static v3 lb_centroid(const lightbounds_t *b) { return v_scale(v_add(b->bmin, b->bmax), 0.5f); } /* (pMin + pMax) / 2 */
static float vcomp(v3 v, int d) { return d == 0 ? v.x : (d == 1 ? v.y : v.z); }
/* CompactLightBounds(lb, allb) and back through its accessors (lightsamplers.h:100-142, 212-242; vecmath.h:1733-1782) */
static float quantize_bounds(float c, float mn, float mx) { return mn == mx ? 0.f : 65535.f * clampf((c - mn) / (mx - mn), 0, 1); }
static float lerpf(float t, float a, float b) { return (1 - t) * a + t * b; }
static uint16_t oct_encode(float f) { return (uint16_t)std::round(clampf((f + 1) / 2, 0, 1) * 65535.f); }
static void compact_node(DLightNode *nd, const lightbounds_t *lb, v3 amin, v3 amax) {
    /* OctahedralVector(Normalize(lb.w)) -> Vector3f */
    v3 v = v_normalize(lb->w);
    float l1 = fabsf(v.x) + fabsf(v.y) + fabsf(v.z);
    v = V3(v.x / l1, v.y / l1, v.z / l1);
    uint16_t ox, oy;
    if (v.z >= 0) { ox = oct_encode(v.x); oy = oct_encode(v.y); }
    else { ox = oct_encode((1 - fabsf(v.y)) * copysignf(1.f, v.x)); oy = oct_encode((1 - fabsf(v.x)) * copysignf(1.f, v.y)); }
    v3 d;
    d.x = -1 + 2 * (ox / 65535.f);
    d.y = -1 + 2 * (oy / 65535.f);
    d.z = 1 - (fabsf(d.x) + fabsf(d.y));
    if (d.z < 0) { float xo = d.x; d.x = (1 - fabsf(d.y)) * copysignf(1.f, xo); d.y = (1 - fabsf(xo)) * copysignf(1.f, d.y); }
    { const v3 wn = v_normalize(d); nd->w[0] = wn.x; nd->w[1] = wn.y; nd->w[2] = wn.z; }
    nd->phi = lb->phi;
    unsigned qo = (unsigned)floorf(32767.f * ((lb->cosTheta_o + 1) / 2)), qe = (unsigned)floorf(32767.f * ((lb->cosTheta_e + 1) / 2));
    nd->cosTheta_o = 2 * (qo / 32767.f) - 1;
    nd->cosTheta_e = 2 * (qe / 32767.f) - 1;
    nd->twoSided = lb->twoSided;
    float lo[3], hi[3];
    for (int c = 0; c < 3; ++c) {
        uint16_t q0 = (uint16_t)floorf(quantize_bounds(vcomp(lb->bmin, c), vcomp(amin, c), vcomp(amax, c)));
        uint16_t q1 = (uint16_t)ceilf(quantize_bounds(vcomp(lb->bmax, c), vcomp(amin, c), vcomp(amax, c)));
        lo[c] = lerpf(q0 / 65535.f, vcomp(amin, c), vcomp(amax, c));
        hi[c] = lerpf(q1 / 65535.f, vcomp(amin, c), vcomp(amax, c));
    }
    for (int c = 0; c < 3; ++c) { nd->bmin[c] = lo[c]; nd->bmax[c] = hi[c]; }
}
static float lb_evaluate_cost(const lightbounds_t *b, v3 bdiag, int dim) { /* lightsamplers.h:398-411 */
    float theta_o = acosf(b->cosTheta_o), theta_e = acosf(b->cosTheta_e);
    float theta_w = fminf(theta_o + theta_e, PI_F);
    float sinTheta_o = safe_sqrt(1 - sqr(b->cosTheta_o));
    float M_omega = 2 * PI_F * (1 - b->cosTheta_o) +
                    PI_F / 2 * (2 * theta_w * sinTheta_o - cosf(theta_o - 2 * theta_w) - 2 * theta_o * sinTheta_o + b->cosTheta_o);
    float Kr = fmaxf(bdiag.x, fmaxf(bdiag.y, bdiag.z)) / vcomp(bdiag, dim);
    v3 d = v_sub(b->bmax, b->bmin);
    float area = 2 * (d.x * d.y + d.x * d.z + d.y * d.z); /* Bounds3::SurfaceArea */
    return b->phi * M_omega * Kr * area;
}
typedef struct { int light; lightbounds_t lb; } lbvh_item_t;
static int lbvh_bucket(const lightbounds_t *lb, v3 cmin, v3 cmax, int dim) { /* nBuckets * centroidBounds.Offset(pc)[dim], 12 buckets */
    float pc = vcomp(lb_centroid(lb), dim), mn = vcomp(cmin, dim), mx = vcomp(cmax, dim);
    float o = pc - mn;
    if (mx > mn) o /= mx - mn;
    int b = (int)(12 * o);
    return b == 12 ? 11 : b;
}
static int lbvh_build(DLightSampler *ls, lbvh_item_t *it, int start, int end, uint32_t bitTrail, int depth, v3 amin, v3 amax, lightbounds_t *out) {
    if (end - start == 1) {
        int nodeIndex = ls->n_nodes++;
        compact_node(&ls->nodes[nodeIndex], &it[start].lb, amin, amax);
        ls->nodes[nodeIndex].child_or_light = (uint32_t)it[start].light;
        ls->nodes[nodeIndex].is_leaf = 1;
        ls->bit_trail[it[start].light] = bitTrail;
        *out = it[start].lb;
        return nodeIndex;
    }
    v3 bmin = V3(INFINITY, INFINITY, INFINITY), bmax = V3(-INFINITY, -INFINITY, -INFINITY), cmin = bmin, cmax = bmax;
    for (int i = start; i < end; ++i) {
        bmin = v_min3(bmin, it[i].lb.bmin); bmax = v_max3(bmax, it[i].lb.bmax);
        v3 c = lb_centroid(&it[i].lb);
        cmin = v_min3(cmin, c); cmax = v_max3(cmax, c);
    }
    float minCost = INFINITY;
    int minCostSplitBucket = -1, minCostSplitDim = -1;
    const v3 bdiag = v_sub(bmax, bmin);
    for (int dim = 0; dim < 3; ++dim) {
        if (vcomp(cmax, dim) == vcomp(cmin, dim)) continue;
        lightbounds_t bucket[12];
        for (int b = 0; b < 12; ++b) bucket[b] = lb_empty();
        for (int i = start; i < end; ++i) {
            int b = lbvh_bucket(&it[i].lb, cmin, cmax, dim);
            bucket[b] = lb_union(bucket[b], it[i].lb);
        }
        float cost[11];
        for (int i = 0; i < 11; ++i) {
            lightbounds_t b0 = lb_empty(), b1 = lb_empty();
            for (int j = 0; j <= i; ++j) b0 = lb_union(b0, bucket[j]);
            for (int j = i + 1; j < 12; ++j) b1 = lb_union(b1, bucket[j]);
            cost[i] = lb_evaluate_cost(&b0, bdiag, dim) + lb_evaluate_cost(&b1, bdiag, dim);
        }
        for (int i = 1; i < 11; ++i)
            if (cost[i] > 0 && cost[i] < minCost) { minCost = cost[i]; minCostSplitBucket = i; minCostSplitDim = dim; }
    }
    int mid;
    if (minCostSplitDim == -1) mid = (start + end) / 2;
    else {
        /* std::partition (libstdc++, forward-iterator form is not used for pointers: the bidirectional algorithm) */
        int first = start, last = end;
        while (1) {
            while (1) {
                if (first == last) goto done;
                if (lbvh_bucket(&it[first].lb, cmin, cmax, minCostSplitDim) <= minCostSplitBucket) ++first; else break;
            }
            --last;
            while (1) {
                if (first == last) goto done;
                if (!(lbvh_bucket(&it[last].lb, cmin, cmax, minCostSplitDim) <= minCostSplitBucket)) --last; else break;
            }
            lbvh_item_t t = it[first]; it[first] = it[last]; it[last] = t;
            ++first;
        }
    done:
        mid = first;
        if (mid == start || mid == end) mid = (start + end) / 2;
    }
    int nodeIndex = ls->n_nodes++;
    lightbounds_t l0, l1;
    lbvh_build(ls, it, start, mid, bitTrail, depth + 1, amin, amax, &l0);
    int child1 = lbvh_build(ls, it, mid, end, bitTrail | (1u << depth), depth + 1, amin, amax, &l1);
    lightbounds_t lb = lb_union(l0, l1);
    compact_node(&ls->nodes[nodeIndex], &lb, amin, amax);
    ls->nodes[nodeIndex].child_or_light = (uint32_t)child1;
    ls->nodes[nodeIndex].is_leaf = 0;
    *out = lb;
    return nodeIndex;
}
static lightbounds_t quad_light_bounds(const DQuad &q, int reverse_orientation) {  // lights.cpp:845-864; shapes.cpp:1070-1126
    const v3 p00 = V3(q.p00[0], q.p00[1], q.p00[2]), p10 = V3(q.p10[0], q.p10[1], q.p10[2]), p01 = V3(q.p01[0], q.p01[1], q.p01[2]),
             p11 = V3(q.p11[0], q.p11[1], q.p11[2]);
    lightbounds_t lb;
    lb.bmin = v_min3(v_min3(p00, p01), v_min3(p10, p11));
    lb.bmax = v_max3(v_max3(p00, p01), v_max3(p10, p11));
    v3 n00 = v_normalize(v_cross(v_sub(p10, p00), v_sub(p01, p00)));
    v3 n10 = v_normalize(v_cross(v_sub(p11, p10), v_sub(p00, p10)));
    v3 n01 = v_normalize(v_cross(v_sub(p00, p01), v_sub(p11, p01)));
    v3 n11 = v_normalize(v_cross(v_sub(p01, p11), v_sub(p10, p11)));
    if (reverse_orientation) { n00 = v_neg(n00); n10 = v_neg(n10); n01 = v_neg(n01); n11 = v_neg(n11); }
    v3 n = v_normalize(v_add(v_add(n00, n10), v_add(n01, n11)));
    float cosTheta = fminf(fminf(v_dot(n, n00), v_dot(n, n01)), fminf(v_dot(n, n10), v_dot(n, n11)));
    dircone_t nb = dircone(n, clampf(cosTheta, -1, 1));
    float phi = fmaxf(q.Le[0], fmaxf(q.Le[1], q.Le[2]));  // Lemit->MaxValue(), RGB rendering mode (spectrum.h:772-779)
    phi *= 1.f * q.area * PI_F;                           // scale * area * Pi
    lb.w = v_normalize(nb.w);
    lb.phi = phi;
    lb.cosTheta_o = nb.cosTheta;
    lb.cosTheta_e = cosf(PI_F / 2);
    lb.twoSided = q.two_sided;
    return lb;
}
static void build(const VspgScene &sc, const VspgIntegratorParams &prm, DScene *D) {
    DLightSampler *ls = &D->lsamp;
    std::memset(ls, 0, sizeof *ls);
    const int n_all = D->n_lights + D->n_inf;
    for (int i = 0; i < kMaxLights; ++i) ls->bit_trail[i] = 0xffffffffu;
    for (int i = 0; i < VSPG_MAX_QUADS; ++i) ls->light_of_quad[i] = -1;
    for (int i = 0; i < D->n_lights; ++i) ls->light_of_quad[D->light_quads[i]] = i;
    // every sampler picks a scene's only light with pmf 1: the two-line uniform pick serves (and keeps the workgroup kernels)
    ls->mode = n_all > 1 ? prm.lightsampler : VSPG_LIGHTSAMPLER_UNIFORM;
    lbvh_item_t items[LBVH_MAX_LIGHTS];
    int n_items = 0;
    v3 amin = V3(INFINITY, INFINITY, INFINITY), amax = V3(-INFINITY, -INFINITY, -INFINITY);
    for (int i = 0; i < n_all; ++i) {
        if (i >= D->n_lights) { ls->inf_light[ls->n_inf++] = i; continue; }  // Bounds() == {}
        const int qi = D->light_quads[i];
        lightbounds_t lb = quad_light_bounds(D->quads[qi], sc.quads[qi].reverse_orientation);
        if (lb.phi > 0) {
            items[n_items].light = i; items[n_items].lb = lb; n_items++;
            amin = v_min3(amin, lb.bmin); amax = v_max3(amax, lb.bmax);
        }
    }
    if (n_items > 0) { lightbounds_t root; lbvh_build(ls, items, 0, n_items, 0, 0, amin, amax, &root); }
    // PowerLightSampler: phi = SafeDiv(light.Phi(lambda), lambda.PDF()).Average(), lambda = SampledWavelengths::SampleVisible(0.5f)
    if (n_all > 0) {
        float pdf[3], power[LBVH_MAX_LIGHTS], acc0 = 0.f;
        for (int i = 0; i < 3; ++i) {
            float up = 0.5f + (float)i / 3;
            if (up > 1) up -= 1;
            float lambda = 538 - 138.888889f * atanhf(0.85691062f - 1.82750197f * up);
            pdf[i] = lambda < 360 || lambda > 830 ? 0.f : 0.0039398042f / sqr(coshf(0.0072f * (lambda - 538)));
        }
        for (int i = 0; i < n_all; ++i) {
            float L[3], k;
            if (i < D->n_lights) {
                const DQuad &q = D->quads[D->light_quads[i]];
                for (int c = 0; c < 3; ++c) L[c] = q.Le[c];
                k = PI_F * (q.two_sided ? 2 : 1) * q.area;  // DiffuseAreaLight::Phi (lights.cpp:826-843)
            } else {
                const int j = i - D->n_lights;
                for (int c = 0; c < 3; ++c) L[c] = D->inf_L[j][c];
                k = D->inf_type[j] == VSPG_LIGHT_DISTANT ? PI_F * sqr(D->scene_radius) : 4 * PI_F * PI_F * sqr(D->scene_radius);
            }
            float s = 0.f;
            for (int c = 0; c < 3; ++c) s += pdf[c] != 0 ? (k * L[c]) / pdf[c] : 0.f;
            power[i] = s / 3;
            acc0 += power[i];
        }
        if (acc0 == 0.f) for (int i = 0; i < n_all; ++i) power[i] = 1.f;
        double sum = 0.;  // AliasTable (util/sampling.cpp:563-618)
        for (int i = 0; i < n_all; ++i) sum += power[i];
        const float fsum = (float)sum;
        ls->n_alias = n_all;
        struct { float pHat; int index; } under[LBVH_MAX_LIGHTS], over[LBVH_MAX_LIGHTS];
        int nu = 0, no = 0;
        for (int i = 0; i < n_all; ++i) {
            ls->alias_p[i] = power[i] / fsum;
            float pHat = ls->alias_p[i] * n_all;
            if (pHat < 1) { under[nu].pHat = pHat; under[nu].index = i; nu++; } else { over[no].pHat = pHat; over[no].index = i; no++; }
        }
        while (nu > 0 && no > 0) {
            float upH = under[nu - 1].pHat, ovH = over[no - 1].pHat;
            int ui = under[nu - 1].index, oi = over[no - 1].index;
            nu--; no--;
            ls->alias_q[ui] = upH; ls->alias_i[ui] = oi;
            float pExcess = upH + ovH - 1;
            if (pExcess < 1) { under[nu].pHat = pExcess; under[nu].index = oi; nu++; } else { over[no].pHat = pExcess; over[no].index = oi; no++; }
        }
        while (no > 0) { no--; ls->alias_q[over[no].index] = 1; ls->alias_i[over[no].index] = -1; }
        while (nu > 0) { nu--; ls->alias_q[under[nu].index] = 1; ls->alias_i[under[nu].index] = -1; }
    }
}
}  // namespace lsb

static bool derive_triangle(const float *p9, const float *kd, int id, DTri *T, const int32_t *flags = nullptr);
// SURF_* flags from the C-ABI's material / medium_interface pair: a MediumInterface only counts when it is a TRANSITION
// (inside != outside, base/medium.h:124)
static int32_t surf_flags_of(int material, int medium_interface) {
    const int b = medium_interface & (VSPG_IFACE_INSIDE | VSPG_IFACE_OUTSIDE);
    const int iface = (b == VSPG_IFACE_INSIDE || b == VSPG_IFACE_OUTSIDE) ? b : 0;
    return (material == VSPG_MATERIAL_INTERFACE ? SURF_INTERFACE : 0) | (iface << SURF_IFACE_SHIFT);
}
static void build_dscene(const VspgScene &sc, const VspgIntegratorParams &prm, const VspgRenderConfig &cfg, DScene *D) {
    using namespace hostmath;
    memset(D, 0, sizeof *D);
    D->n_quads = sc.n_quads;
    for (int i = 0; i < sc.n_quads; ++i) {
        const VspgQuad &in = sc.quads[i];
        DQuad &q = D->quads[i];
        H3 p00 = ld(in.p00), e1 = ld(in.e1), e2 = ld(in.e2);
        H3 p10 = add(p00, e1), p01 = add(p00, e2), p11 = add(p10, e2);
        stv(q.p00, p00); stv(q.p10, p10); stv(q.p01, p01); stv(q.p11, p11); stv(q.e1, e1); stv(q.e2, e2);
        H3 c = crossv(e1, e2);
        q.area = std::sqrt(len2v(c));
        H3 n = normv(c);
        if (in.reverse_orientation) n = H3{-n.x, -n.y, -n.z};
        stv(q.n, n);
        stv(q.dpdu_n, normv(e1));
        const float eps = 0x1p-24f;
        float g6 = (6 * eps) / (1 - 6 * eps);  // gamma(6) (util/float.h:195)
        H3 s = add(add(absv(p00), absv(p01)), add(absv(p10), absv(p11)));
        stv(q.perr, H3{s.x * g6, s.y * g6, s.z * g6});
        q.inv_l1 = 1.f / len2v(e1);
        q.inv_l2 = 1.f / len2v(e2);
        bool lobes = false, light = false;
        for (int k = 0; k < 3; ++k) {
            float kd = in.Kd[k];
            kd = kd < 0 ? 0 : (kd > 1 ? 1 : kd);  // DiffuseMaterial clamps reflectance to [0,1]
            q.Kd[k] = kd;
            q.Le[k] = in.Le[k];
            lobes = lobes || kd != 0;
            light = light || in.Le[k] != 0;
        }
        q.two_sided = in.two_sided;
        q.is_light = light;
        q.has_lobes = lobes;
        q.flags = surf_flags_of(in.material, in.medium_interface);
        // axis-aligned fast path: n, e1, e2 each have exactly one non-zero component
        auto single_axis = [](const float *v) {
            int nz = 0, ax = -1;
            for (int k = 0; k < 3; ++k)
                if (v[k] != 0) { nz++; ax = k; }
            return nz == 1 ? ax : -1;
        };
        int an = single_axis(q.n), a1 = single_axis(q.e1), a2 = single_axis(q.e2);
        IsectRec &rec = D->irec[i];
        memset(&rec, 0, sizeof rec);
        if (an >= 0 && a1 >= 0 && a2 >= 0 && an != a1 && an != a2 && a1 != a2 && std::fabs(q.n[an]) == 1.0f) {
            rec.kind = 1;
            rec.axes = an | (a1 << 2) | (a2 << 4);
            rec.f[0] = q.n[an]; rec.f[1] = q.p00[an]; rec.f[2] = q.p00[a1]; rec.f[3] = q.p00[a2];
            rec.f[4] = q.e1[a1]; rec.f[5] = q.e2[a2]; rec.f[6] = q.inv_l1; rec.f[7] = q.inv_l2;
        } else {
            rec.kind = 0;
            for (int k = 0; k < 3; ++k) { rec.f[k] = q.n[k]; rec.f[3 + k] = q.p00[k]; rec.f[6 + k] = q.e1[k]; rec.f[9 + k] = q.e2[k]; }
            rec.f[12] = q.inv_l1; rec.f[13] = q.inv_l2;
        }
        if (light) D->light_quads[D->n_lights++] = i;
    }
    D->n_inf = sc.n_infinite_lights;
    for (int i = 0; i < sc.n_infinite_lights && i < VSPG_MAX_INFINITE_LIGHTS; ++i) {
        D->inf_type[i] = sc.infinite_lights[i].type;
        for (int k = 0; k < 3; ++k) { D->inf_L[i][k] = sc.infinite_lights[i].L[k]; D->inf_w[i][k] = sc.infinite_lights[i].w_light[k]; }
    }
    {   // scene bounds = union of the primitives' bounds; light.Preprocess: BoundingSphere (integrators.h:74-81, vecmath.h:1335-1338)
        float lo[3] = {kInf, kInf, kInf}, hi[3] = {-kInf, -kInf, -kInf};
        auto grow = [&](const float *q) { for (int k = 0; k < 3; ++k) { lo[k] = q[k] < lo[k] ? q[k] : lo[k]; hi[k] = q[k] > hi[k] ? q[k] : hi[k]; } };
        for (int i = 0; i < D->n_quads; ++i) { grow(D->quads[i].p00); grow(D->quads[i].p10); grow(D->quads[i].p01); grow(D->quads[i].p11); }
        for (int i = 0; i < sc.n_triangles && sc.tri_p; ++i) {
            DTri T;
            if (derive_triangle(sc.tri_p + 9 * (size_t)i, nullptr, i, &T)) { grow(T.p0); grow(T.p1); grow(T.p2); }
        }
        for (int i = 0; i < sc.n_spheres; ++i) {  // Sphere::Bounds (shapes.cpp:33-36): the corners of the object-space box, transformed
            const float *m = sc.spheres[i].render_from_object, rr = sc.spheres[i].radius;
            for (int c = 0; c < 8; ++c) {
                const float x = (c & 1) ? rr : -rr, y = (c & 2) ? rr : -rr, z = (c & 4) ? rr : -rr;
                float q[3];
                for (int k = 0; k < 3; ++k) {
                    volatile float a = m[4 * k] * x;  // volatile: one rounding per operation, as the oracle's -ffp-contract=off build
                    volatile float b = m[4 * k + 1] * y;
                    volatile float c2 = m[4 * k + 2] * z;
                    volatile float sum = a + b;
                    sum = sum + c2;
                    sum = sum + m[4 * k + 3];
                    q[k] = sum;
                }
                grow(q);
            }
        }
        D->scene_radius = 0.f;
        if (lo[0] <= hi[0]) {
            volatile float cx = (lo[0] + hi[0]) / 2, cy = (lo[1] + hi[1]) / 2, cz = (lo[2] + hi[2]) / 2;
            volatile float dx = cx - hi[0], dy = cy - hi[1], dz = cz - hi[2];
            volatile float l2 = dx * dx;
            l2 = l2 + dy * dy;
            l2 = l2 + dz * dz;
            D->scene_radius = std::sqrt(l2);
        }
    }
    // Shape "sphere": the constants the Sphere constructor derives (shapes.h:117-128), full sphere
    D->n_spheres = sc.n_spheres;
    for (int i = 0; i < sc.n_spheres; ++i) {
        const VspgSphere &in = sc.spheres[i];
        DSphere &sp = D->spheres[i];
        for (int k = 0; k < 12; ++k) { sp.m[k] = in.render_from_object[k]; sp.mi[k] = in.object_from_render[k]; }
        sp.radius = in.radius;
        const float *m = in.render_from_object;  // Transform::SwapsHandedness: Determinant(SquareMatrix<3>) < 0 (transform.cpp:145-152, math.h:1419-1425)
        const float minor12 = dop(m[5], m[10], m[6], m[9]), minor02 = dop(m[4], m[10], m[6], m[8]), minor01 = dop(m[4], m[9], m[5], m[8]);
        const bool swaps = std::fmaf(m[2], minor01, dop(m[0], minor12, m[1], minor02)) < 0;
        sp.flip = (in.reverse_orientation ? 1 : 0) ^ (swaps ? 1 : 0);
        auto clamp1 = [](float v) { return v < -1.f ? -1.f : (v > 1.f ? 1.f : v); };
        sp.thetaZMin = std::acos(clamp1(-in.radius / in.radius));  // zMin = -radius, zMax = radius (the "sphere" defaults, shapes.cpp:231-237)
        sp.thetaZMax = std::acos(clamp1(in.radius / in.radius));
        {
            volatile float rad = 3.14159265358979323846f / 180;  // Radians(360) (math.h:261-263) == fl(2 pi): no phi clipping
            rad = rad * 360.f;
            sp.phiMax = rad;
        }
        bool lobes = false;
        for (int k = 0; k < 3; ++k) {
            float kd = in.Kd[k];
            kd = kd < 0 ? 0 : (kd > 1 ? 1 : kd);
            sp.Kd[k] = kd;
            lobes = lobes || kd != 0;
        }
        sp.has_lobes = lobes;
        sp.flags = surf_flags_of(in.material, in.medium_interface);
    }
    // medium boundaries: anything that makes "the medium fills the scene" false
    D->camera_in_medium = sc.medium.type != VSPG_MEDIUM_NONE && !sc.camera_outside_medium;
    D->has_boundaries = sc.camera_outside_medium != 0;
    for (int i = 0; i < D->n_quads; ++i) D->has_boundaries |= D->quads[i].flags != 0;
    for (int i = 0; i < D->n_spheres; ++i) D->has_boundaries |= D->spheres[i].flags != 0;
    for (int i = 0; i < sc.n_triangles && sc.tri_flags; ++i)
        D->has_boundaries |= surf_flags_of(sc.tri_flags[i] & VSPG_TRI_INTERFACE, sc.tri_flags[i] >> VSPG_TRI_IFACE_SHIFT) != 0;
    D->cam = sc.camera;
    D->medium_type = sc.medium.type;
    for (int k = 0; k < 3; ++k) {
        D->sigma_a[k] = sc.medium.sigma_a[k];
        D->sigma_s[k] = sc.medium.sigma_s[k];
        {
            volatile float st = D->sigma_s[k] + D->sigma_a[k];  // volatile: one rounding per operation, no contraction
            volatile float sn = st - D->sigma_a[k];
            sn = sn - D->sigma_s[k];
            D->sigma_t[k] = st;
            D->sigma_n_raw[k] = sn;
        }
        D->Le[k] = sc.medium.Le[k];
    }
    D->g = sc.medium.g;
    for (int k = 0; k < 3; ++k) {
        D->index_min[k] = sc.medium.index_min[k];
        D->inv_voxel[k] = sc.medium.type == VSPG_MEDIUM_NANOVDB ? 1.0f / sc.medium.voxel_size[k] : 0.f;
        D->grid_origin[k] = sc.medium.grid_origin[k];
    }
    D->density_offset = sc.medium.density_offset;
    D->has_xform = sc.medium.has_transform ? 1 : 0;
    for (int k = 0; k < 12; ++k) D->minv[k] = sc.medium.has_transform ? sc.medium.medium_from_render[k] : ((k % 5) == 0 ? 1.f : 0.f);
    D->nx = sc.medium.nx;
    D->ny = sc.medium.ny;
    D->nz = sc.medium.nz;
    for (int k = 0; k < 3; ++k) {
        D->bounds_min[k] = sc.medium.bounds_min[k];
        D->bounds_max[k] = sc.medium.bounds_max[k];
    }
    D->prm = prm;
    D->xres = cfg.xres;
    D->yres = cfg.yres;
    D->seed = cfg.seed;
    D->shard_index = cfg.shard_index;
    D->shard_count = cfg.shard_count < 1 ? 1 : cfg.shard_count;
    lsb::build(sc, prm, D);  // LightSampler::Create(prm.lightsampler, lights) (lightsamplers.cpp:49-64)
}

// GridMedium constructor: majorantGrid.Set(x,y,z, densityGrid.MaxValue(VoxelBounds(x,y,z)))
// (src/pbrt/media.cpp:262-269; SampledGrid::MaxValue src/pbrt/util/containers.h:838-854)
static std::vector<float> build_majorant_grid(const VspgMedium &m) {
    const int R = 16;
    std::vector<float> maj((size_t)R * R * R);
    const int n[3] = {m.nx, m.ny, m.nz};
    auto at = [&](int x, int y, int z) -> float {
        if (x < 0 || y < 0 || z < 0 || x >= m.nx || y >= m.ny || z >= m.nz) return 0.f;
        return m.density[((size_t)z * m.ny + y) * m.nx + x];
    };
    for (int z = 0; z < R; ++z)
        for (int y = 0; y < R; ++y)
            for (int x = 0; x < R; ++x) {
                const int c[3] = {x, y, z};
                int lo[3], hi[3];
                for (int k = 0; k < 3; ++k) {
                    float p0 = (float)c[k] / R, p1 = (float)(c[k] + 1) / R;
                    int a = (int)std::floor(p0 * n[k] - .5f), b = (int)std::floor(p1 * n[k] - .5f) + 1;
                    lo[k] = std::max(a, 0);
                    hi[k] = std::min(b, n[k] - 1);
                }
                float mx = at(lo[0], lo[1], lo[2]);
                for (int zz = lo[2]; zz <= hi[2]; ++zz)
                    for (int yy = lo[1]; yy <= hi[1]; ++yy)
                        for (int xx = lo[0]; xx <= hi[0]; ++xx) mx = std::max(mx, at(xx, yy, zz));
                maj[x + R * (y + R * z)] = mx;
            }
    return maj;
}

// ---- f1: BVH over the triangle soup (own builder: binned SAH on centroids, leaves of <= 4 triangles, depth-first
// layout with skip links -- see DBvhNode).  cpu/aggregates.cpp:529-640 is what it stands in for; WHICH triangles a ray
// tests never changes a result (vspg_device.h: bvh_closest), so the builder is free.
namespace bvhbuild {
// the binary tree the builder grows first, depth-first: an inner node's first child is the next node, `skip` is the index behind
// its subtree (so its second child sits at nodes[first child].skip)
struct DBvhNode {
    float bmin[3];
    int32_t skip;
    float bmax[3];
    int32_t leaf;    // >= 0: first triangle * 8 + count (1..7); -1: inner node
};
struct Box { float lo[3], hi[3]; };
static Box empty_box() { return Box{{kInf, kInf, kInf}, {-kInf, -kInf, -kInf}}; }
static void grow(Box &b, const float *p) { for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(b.lo[k], p[k]); b.hi[k] = std::max(b.hi[k], p[k]); } }
static void merge(Box &b, const Box &o) { for (int k = 0; k < 3; ++k) { b.lo[k] = std::min(b.lo[k], o.lo[k]); b.hi[k] = std::max(b.hi[k], o.hi[k]); } }
static float area(const Box &b) {
    const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
    return dx < 0 ? 0.f : 2 * (dx * dy + dy * dz + dz * dx);
}
struct Builder {
    const float *p = nullptr;       // 9 floats per triangle
    std::vector<int> order;         // triangle indices, permuted in place
    std::vector<Box> tbox;
    std::vector<float> cen;         // 3 per triangle
    std::vector<DBvhNode> nodes;
    std::vector<signed char> split_axis;   // per node: the axis its children were separated along (inner nodes)
    // The device tree (DBvh4Node): an inner node takes its two children and, while it holds fewer than four, replaces the inner child
    // of the largest surface area by that child's two children.  Returns the node's index; *depth: levels below and including it.
    bool balanced = false;          // median splits only: the fallback that bounds the depth for the traversal's stack
    std::vector<DBvh4Node> nodes4;
    int collapse(int node, int *depth) {
        const int me = (int)nodes4.size();
        nodes4.push_back(DBvh4Node{});
        int kids[4], nk = 0;
        if (nodes[node].leaf >= 0) {
            kids[nk++] = node;  // (a soup that fits one leaf: a root with one child)
        } else {
            kids[nk++] = node + 1;
            kids[nk++] = nodes[node + 1].skip;
            while (nk < 4) {
                int pick = -1;
                float pa = -1.f;
                for (int k = 0; k < nk; ++k)
                    if (nodes[kids[k]].leaf < 0) {
                        const Box b{{nodes[kids[k]].bmin[0], nodes[kids[k]].bmin[1], nodes[kids[k]].bmin[2]}, {nodes[kids[k]].bmax[0], nodes[kids[k]].bmax[1], nodes[kids[k]].bmax[2]}};
                        const float ar = area(b);
                        if (ar > pa) { pa = ar; pick = k; }
                    }
                if (pick < 0) break;
                const int c = kids[pick];
                kids[pick] = c + 1;
                kids[nk++] = nodes[c + 1].skip;
            }
        }
        int dmax = 0;
        for (int k = 0; k < 4; ++k) {
            DBvh4Node &N = nodes4[me];  // (re-taken every round: the recursion below grows the vector)
            if (k >= nk) {
                N.lox[k] = N.loy[k] = N.loz[k] = kInf; N.hix[k] = N.hiy[k] = N.hiz[k] = -kInf;
                N.child[k] = kBvhAbsent;
                continue;
            }
            const DBvhNode &c = nodes[kids[k]];
            N.lox[k] = c.bmin[0]; N.loy[k] = c.bmin[1]; N.loz[k] = c.bmin[2];
            N.hix[k] = c.bmax[0]; N.hiy[k] = c.bmax[1]; N.hiz[k] = c.bmax[2];
            if (c.leaf >= 0) {
                N.child[k] = -c.leaf - 1;
            } else {
                int dk = 0;
                const int idx = collapse(kids[k], &dk);
                nodes4[me].child[k] = idx;
                dmax = dk > dmax ? dk : dmax;
            }
        }
        for (int k = 0; k < 4; ++k) nodes4[me].pad[k] = 0;
        *depth = dmax + 1;
        return me;
    }
#ifndef VSPG_BVH_LEAF
#define VSPG_BVH_LEAF 2
#endif
    // a range of at most this many triangles is not split further (a leaf holds <= 7).  Measured (scripts/tri_timing.py, fog box + 100 k
    // triangles, ms per 1080p wave): 6: 8.0, 4: 7.4, 2: 6.4, 1: 6.5 -- the watertight triangle test is the expensive part of a visit
    static constexpr int kLeafMax = VSPG_BVH_LEAF;
    int bin_of(int t, int ax, float lo_a, float ext, int nb) const {
        const int k = (int)(nb * ((cen[3 * t + ax] - lo_a) / ext));
        return k < 0 ? 0 : (k >= nb ? nb - 1 : k);
    }
    void build(int lo, int hi) {
        const int me = (int)nodes.size();
        nodes.push_back(DBvhNode{});
        split_axis.push_back(0);
        Box b = empty_box(), cb = empty_box();
        for (int i = lo; i < hi; ++i) { merge(b, tbox[order[i]]); grow(cb, &cen[3 * order[i]]); }
        for (int k = 0; k < 3; ++k) { nodes[me].bmin[k] = b.lo[k]; nodes[me].bmax[k] = b.hi[k]; }
        const int n = hi - lo;
        int axis = 0;
        for (int k = 1; k < 3; ++k) if (cb.hi[k] - cb.lo[k] > cb.hi[axis] - cb.lo[axis]) axis = k;
        int mid = -1;
        if (n > kLeafMax && !balanced) {  // binned SAH over all three axes: the cheapest of the 3 x (NB - 1) candidate planes
            constexpr int NB = 16;
            float best = kInf; int bs = -1, baxis = -1;
            for (int ax = 0; ax < 3; ++ax) {
                const float ext = cb.hi[ax] - cb.lo[ax];
                if (!(ext > 0)) continue;
                Box bb[NB]; int cnt[NB];
                for (int i = 0; i < NB; ++i) { bb[i] = empty_box(); cnt[i] = 0; }
                for (int i = lo; i < hi; ++i) { const int k = bin_of(order[i], ax, cb.lo[ax], ext, NB); cnt[k]++; merge(bb[k], tbox[order[i]]); }
                for (int s = 0; s < NB - 1; ++s) {
                    Box l = empty_box(), rr = empty_box(); int nl = 0, nr = 0;
                    for (int i = 0; i <= s; ++i) { merge(l, bb[i]); nl += cnt[i]; }
                    for (int i = s + 1; i < NB; ++i) { merge(rr, bb[i]); nr += cnt[i]; }
                    if (!nl || !nr) continue;
                    const float c = nl * area(l) + nr * area(rr);
                    if (c < best) { best = c; bs = s; baxis = ax; }
                }
            }
            if (bs >= 0) {
                axis = baxis;
                const float lo_a = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
                auto it = std::partition(order.begin() + lo, order.begin() + hi, [&](int t) { return bin_of(t, axis, lo_a, ext, NB) <= bs; });
                mid = (int)(it - order.begin());
            }
        }
        if ((n > 7 || (balanced && n > kLeafMax)) && (mid <= lo || mid >= hi)) {  // no useful split but too many for a leaf (or the balanced fallback): median by index
            mid = lo + n / 2;
            std::nth_element(order.begin() + lo, order.begin() + mid, order.begin() + hi,
                             [&](int a, int c) { return cen[3 * a + axis] < cen[3 * c + axis]; });
        }
        if (mid > lo && mid < hi) {
            nodes[me].leaf = -1;
            split_axis[me] = (signed char)axis;
            build(lo, mid);
            build(mid, hi);
        } else {
            nodes[me].leaf = lo * 8 + n;  // n <= 7
        }
        nodes[me].skip = (int)nodes.size();
    }
};
}  // namespace bvhbuild

// what Triangle::InteractionFromIntersection derives from the vertices alone (shapes.h:888-938), same float operations
static bool derive_triangle(const float *p9, const float *kd, int id, DTri *T, const int32_t *flags) {
    using namespace hostmath;
    const H3 p0 = ld(p9), p1 = ld(p9 + 3), p2 = ld(p9 + 6);
    auto sub = [](H3 a, H3 b) { return H3{a.x - b.x, a.y - b.y, a.z - b.z}; };
    if (len2v(crossv(sub(p2, p0), sub(p1, p0))) == 0) return false;  // IntersectTriangle: degenerate -> never hit (shapes.cpp:172-173)
    const H3 dp02 = sub(p0, p2), dp12 = sub(p1, p2);
    // default (u,v) = (0,0), (1,0), (1,1): duv02 = (-1,-1), duv12 = (0,-1)
    const float duv02[2] = {0.f - 1.f, 0.f - 1.f}, duv12[2] = {1.f - 1.f, 0.f - 1.f};
    const float determinant = dop(duv02[0], duv12[1], duv02[1], duv12[0]);
    H3 dpdu{0, 0, 0}, dpdv{0, 0, 0};
    const bool degenerateUV = std::fabs(determinant) < 1e-9f;
    if (!degenerateUV) {
        const float invdet = 1 / determinant;
        dpdu = H3{dop(duv12[1], dp02.x, duv02[1], dp12.x) * invdet, dop(duv12[1], dp02.y, duv02[1], dp12.y) * invdet, dop(duv12[1], dp02.z, duv02[1], dp12.z) * invdet};
        dpdv = H3{dop(duv02[0], dp12.x, duv12[0], dp02.x) * invdet, dop(duv02[0], dp12.y, duv12[0], dp02.y) * invdet, dop(duv02[0], dp12.z, duv12[0], dp02.z) * invdet};
    }
    if (degenerateUV || len2v(crossv(dpdu, dpdv)) == 0) return false;  // (the reference falls back to CoordinateSystem(ng); such slivers are dropped here)
    const H3 n = normv(crossv(dp02, dp12));
    memset(T, 0, sizeof *T);
    stv(T->p0, p0); stv(T->p1, p1); stv(T->p2, p2);
    T->nx = n.x; T->ny = n.y; T->nz = n.z;
    stv(T->dpdu_n, normv(dpdu));
    T->id = id;
    for (int k = 0; k < 3; ++k) { float v = kd ? kd[3 * id + k] : 0.5f; T->Kd[k] = v < 0 ? 0 : (v > 1 ? 1 : v); }
    const int fl = flags ? flags[id] : 0;
    T->flags = surf_flags_of(fl & VSPG_TRI_INTERFACE, fl >> VSPG_TRI_IFACE_SHIFT);
    if (fl & VSPG_TRI_FLIP_NORMAL) { T->nx = -T->nx; T->ny = -T->ny; T->nz = -T->nz; }  // reverseOrientation ^ transformSwapsHandedness (shapes.h:934-936)
    return true;
}

static bool wants_guiding(const VspgIntegratorParams &p) {
    return p.surfaceguiding || p.volumeguiding || (p.vspguiding && p.vspsecondaryguiding) || p.rrguiding;  // guided kernels
}
static bool wants_training(const VspgIntegratorParams &p) {  // the field is queried
    return p.surfaceguiding || p.volumeguiding || (p.vspguiding && p.vspsecondaryguiding);
}

// NanoVDBMedium constructor, "Initialize majorantGrid" (media.cpp:600-671) over the dense copy: 64^3 cells; a
// cell's majorant is the largest voxel value in the cell's index-space footprint widened by one voxel (the
// trilinear filter slop), clipped to the index bounding box, then (max + densityOffset) * majorantScale.
static std::vector<float> build_majorant_grid_nvdb(const VspgMedium &m) {
    const int R = kMajResNvdb;
    std::vector<float> maj((size_t)R * R * R);
    const int imin[3] = {m.index_min[0], m.index_min[1], m.index_min[2]};
    const int imax[3] = {m.index_min[0] + m.nx - 1, m.index_min[1] + m.ny - 1, m.index_min[2] + m.nz - 1};
    auto value = [&](int i, int j, int k) -> float {  // accessor.getValue: background outside the tree
        const int x = i - imin[0], y = j - imin[1], z = k - imin[2];
        if (x < 0 || y < 0 || z < 0 || x >= m.nx || y >= m.ny || z >= m.nz) return 0.f;
        return m.density[((size_t)z * m.ny + y) * m.nx + x];
    };
    auto lerp = [](float t, float a, float b) { return (1 - t) * a + t * b; };  // pbrt::Lerp (math.h)
    for (int z = 0; z < R; ++z)
        for (int y = 0; y < R; ++y)
            for (int x = 0; x < R; ++x) {
                const int c[3] = {x, y, z};
                int lo[3], hi[3];
                for (int k = 0; k < 3; ++k) {
                    const float w0 = lerp((float)c[k] / R, m.bounds_min[k], m.bounds_max[k]);
                    const float w1 = lerp((float)(c[k] + 1) / R, m.bounds_min[k], m.bounds_max[k]);
                    const double i0 = ((double)w0 - (double)m.grid_origin[k]) / (double)m.voxel_size[k];  // worldToIndexF(Vec3R)
                    const double i1 = ((double)w1 - (double)m.grid_origin[k]) / (double)m.voxel_size[k];
                    const float delta = 1.f;
                    lo[k] = std::max((int)(i0 - delta), imin[k]);
                    hi[k] = std::min((int)(i1 + delta), imax[k]);
                }
                float mx = 0;
                for (int kk = lo[2]; kk <= hi[2]; ++kk)
                    for (int jj = lo[1]; jj <= hi[1]; ++jj)
                        for (int ii = lo[0]; ii <= hi[0]; ++ii) mx = std::max(mx, value(ii, jj, kk));
                maj[x + R * (y + R * z)] = (mx + m.density_offset) * m.majorant_scale;
            }
    return maj;
}
static int validate(const VspgScene *scene, const VspgIntegratorParams *p, const VspgRenderConfig *cfg) {
    if (!scene || !p || !cfg) return fail(VSPG_EINVAL, "null argument");
    if (cfg->xres <= 0 || cfg->yres <= 0) return fail(VSPG_EINVAL, "film resolution must be positive");
    if (cfg->xres > 32768 || cfg->yres > 32768) return fail(VSPG_EINVAL, "film resolution above 32768 (pixels travel as packed 16-bit pairs)");
    if (scene->n_quads < 0 || scene->n_quads > VSPG_MAX_QUADS) return fail(VSPG_EINVAL, "n_quads out of range");
    if (scene->n_spheres < 0 || scene->n_spheres > VSPG_MAX_SPHERES) return fail(VSPG_EINVAL, "n_spheres out of range");
    for (int i = 0; i < scene->n_spheres; ++i) {
        const VspgSphere &sp = scene->spheres[i];
        if (!(sp.radius > 0)) return fail(VSPG_EINVAL, "sphere radius must be positive");
        const float *a = sp.render_from_object, *b = sp.object_from_render;
        if (a[12] != 0 || a[13] != 0 || a[14] != 0 || a[15] != 1 || b[12] != 0 || b[13] != 0 || b[14] != 0 || b[15] != 1)
            return fail(VSPG_EINVAL, "a sphere's renderFromObject must be affine (last row 0 0 0 1)");
    }
    if (cfg->shard_count > 1 && (cfg->shard_index < 0 || cfg->shard_index >= cfg->shard_count))
        return fail(VSPG_EINVAL, "shard_index out of range");
    if (p->maxdepth < 0) return fail(VSPG_EINVAL, "maxdepth must be >= 0");
    if (p->maxdepth > 254) return fail(VSPG_EINVAL, "maxdepth above 254 (the path depth travels in 8 bits of the packed path flags)");
    if (!(p->vspmisratio >= 0.f && p->vspmisratio <= 1.f)) return fail(VSPG_EINVAL, "vspmisratio must be in [0,1]");
    if (scene->medium.type == VSPG_MEDIUM_GRID || scene->medium.type == VSPG_MEDIUM_NANOVDB) {
        const VspgMedium &m = scene->medium;
        if (m.nx <= 0 || m.ny <= 0 || m.nz <= 0 || !m.density) return fail(VSPG_EINVAL, "grid medium needs nx,ny,nz > 0 and a density array");
        if (m.type == VSPG_MEDIUM_NANOVDB) {
            for (int k = 0; k < 3; ++k)
                if (!(m.voxel_size[k] > 0)) return fail(VSPG_EINVAL, "nanovdb medium needs a positive voxel_size");
            if (!(m.majorant_scale > 0)) return fail(VSPG_EINVAL, "nanovdb medium needs majorant_scale > 0 (reference default 1)");
        }
        if ((long long)m.nx * m.ny * m.nz > (1ll << 31)) return fail(VSPG_EINVAL, "density grid too large");
        for (int k = 0; k < 3; ++k)
            if (!(m.bounds_max[k] > m.bounds_min[k])) return fail(VSPG_EINVAL, "grid medium bounds must have positive extent");
        if (m.has_transform) {
            const float *a = m.render_from_medium, *b = m.medium_from_render;
            if (a[12] != 0 || a[13] != 0 || a[14] != 0 || a[15] != 1 || b[12] != 0 || b[13] != 0 || b[14] != 0 || b[15] != 1)
                return fail(VSPG_EINVAL, "renderFromMedium must be affine (last row 0 0 0 1)");
            for (int k = 0; k < 16; ++k)
                if (!(a[k] == a[k]) || !(b[k] == b[k])) return fail(VSPG_EINVAL, "renderFromMedium holds a NaN");
        }
        if (m.Le[0] != 0 || m.Le[1] != 0 || m.Le[2] != 0) {
            if (m.type == VSPG_MEDIUM_NANOVDB)
                return fail(VSPG_EINVAL, "NanoVDBMedium has no Le spectrum: it emits through its temperature grid (VspgMedium.temperature)");
            if (m.temperature) return fail(VSPG_EINVAL, "Both \"Le\" and \"temperature\" values were provided.");  // media.cpp:307-308
        }
        // temperature grids (media.h:333-341, :724-735): blackbody volume emission, sampled by the delta-tracking routine only
        // (guidedvolpathvspgintegrator.cpp:895-906) -- under "resampling" a heterogeneous medium never evaluates it
        if (m.type == VSPG_MEDIUM_GRID && (m.temperature || m.Le[0] != 0 || m.Le[1] != 0 || m.Le[2] != 0))
            if (m.le_scale && (m.le_nx <= 0 || m.le_ny <= 0 || m.le_nz <= 0 || (long long)m.le_nx * m.le_ny * m.le_nz > (1ll << 31)))
                return fail(VSPG_EINVAL, "emissive grid medium: bad Lescale grid size");
    } else if (scene->medium.type != VSPG_MEDIUM_NONE && scene->medium.type != VSPG_MEDIUM_HOMOGENEOUS)
        return fail(VSPG_EINVAL, "unknown medium type");
    if (scene->n_triangles < 0 || scene->n_triangles > (1 << 27)) return fail(VSPG_EINVAL, "n_triangles out of range");
    if (scene->n_triangles > 0 && !scene->tri_p) return fail(VSPG_EINVAL, "triangles without vertex data");
    if (scene->n_infinite_lights < 0 || scene->n_infinite_lights > VSPG_MAX_INFINITE_LIGHTS) return fail(VSPG_EINVAL, "n_infinite_lights out of range");
    for (int i = 0; i < scene->n_infinite_lights; ++i)
        if (scene->infinite_lights[i].type != VSPG_LIGHT_UNIFORM_INFINITE && scene->infinite_lights[i].type != VSPG_LIGHT_DISTANT)
            return fail(VSPG_EINVAL, "unknown infinite light type");
    int nl = scene->n_infinite_lights;
    for (int i = 0; i < scene->n_quads; ++i)
        if (scene->quads[i].Le[0] != 0 || scene->quads[i].Le[1] != 0 || scene->quads[i].Le[2] != 0) nl++;
    (void)nl;  // (round 3: "power" and "bvh" serve any number of lights -- vspg_lightsampler.h)
    return 0;
}

// One pass of the wavefront pipeline: sample index `sample` of every pixel.  The pipeline's kernels are instantiated in their own
// translation units (vspg_wf_grid.hip / vspg_wf_nvdb.hip: wf_dispatch_*, vspg_wf_launch.h), which `make -j` builds beside this one;
// here the pass is prepared -- buffers, grids, streams -- and handed over as a plain WfLaunch.
static bool wf_merged_walks(const VspgRenderer *r, bool guided) {
    bool merged = r->hscene.has_boundaries != 0 || guided;
    if (const char *e = getenv("VSPG_WF_MERGED")) { if (e[0] == '0') merged = false; else if (e[0] == '1') merged = true; }
    return merged;
}
static int wf_render_pass(VspgRenderer *r, int sample, hipStream_t s, bool nvdb, bool guided, bool train, bool grey) {
    const int tilesX = (r->cfg.xres + 7) / 8, tilesY = (r->cfg.yres + 7) / 8;
    const size_t items = (size_t)tilesX * tilesY * 64;
    // Path-loop iterations of a pass.  Without medium boundaries every iteration ends at a vertex and raises the depth: maxdepth + 1
    // of them.  With boundaries an iteration may instead cross an interface (Li's `continue` at :399-404, depth unchanged): the loop
    // runs until the list is empty -- a convex bounding shape costs at most two crossings per vertex -- under a generous cap.
    const bool bnd = r->hscene.has_boundaries != 0;
    const int base_iters = r->prm.maxdepth + 1;
    const int max_iters = bnd ? 4 * (r->prm.maxdepth + 2) + 16 : base_iters;
    const int n_iters = max_iters + 1;
    if (!r->wf_pool || r->wf_items != items) {
        if (r->wf_pool) (void)hipFree(r->wf_pool);
        if (r->wf_lists) (void)hipFree(r->wf_lists);
        if (r->wf_iters) (void)hipFree(r->wf_iters);
        r->wf_pool = nullptr; r->wf_lists = nullptr; r->wf_iters = nullptr;
        HIPCHK(hipMalloc(&r->wf_pool, (size_t)kWfPoolFloats * items * sizeof(float)));
        HIPCHK(hipMalloc(&r->wf_lists, 4 * items * sizeof(unsigned int)));
        HIPCHK(hipMalloc(&r->wf_iters, (size_t)n_iters * sizeof(WfIter)));
        r->wf_items = items;
    }
    HIPCHK(hipMemsetAsync(r->wf_iters, 0, (size_t)n_iters * sizeof(WfIter), s));
    WfLaunch L;
    WfArgs &a = L.a;
    a.scene = r->dscene;
    a.P = WfPool{r->wf_pool, items};
    a.film = r->film;
    a.isg_stats = r->isg_stats;
    a.vsp_buf = r->vsp;
    a.vsp_ready = r->vsp_ready;
    a.sample = sample;
    a.jump = pcg_jump((unsigned long long)sample * 65536ull);
    a.n_items = (unsigned)items;
    a.tilesX = (unsigned)tilesX;
    a.list_active = r->wf_lists;
    a.list_active2 = r->wf_lists + items;
    a.list_walk = r->wf_lists + 2 * items;
    a.list_shadow = r->wf_lists + 3 * items;
    a.iters = r->wf_iters;
    a.counters = r->counters;
    a.train = TrainArgs{nullptr, nullptr, nullptr, nullptr, 0, 0};
    a.rec_cap = train_rec_capacity(r->prm.maxdepth);
    {   // (VSPG_WF_COMPACT=0: the full layout for a grey medium too -- same results, for A/B runs)
        const char *e = getenv("VSPG_WF_COMPACT");
        a.compact_results = grey && !(e && e[0] == '0') ? 1 : 0;
    }
    if (train) {  // a18: the pass records path segments; PropagateSamples (k_propagate) follows it
        a.train = TrainArgs{r->segbuf, r->seg_count, r->samples, r->train_counters, r->sample_capacity, (unsigned)items};
        HIPCHK(hipMemsetAsync(r->seg_count, 0, items * sizeof(int), s));
    }
    {   // tuning knobs (defaults measured on the 256^3 cloud stand-in)
        const char *e1 = getenv("VSPG_WF_ROUNDS"), *e2 = getenv("VSPG_WF_REFILL");
        a.walk_rounds = e1 ? atoi(e1) : (nvdb ? GridMediumT<true>::kAdvanceRounds : GridMediumT<false>::kAdvanceRounds);
        a.walk_refill = e2 ? atoi(e2) : kWfRefill;
        if (a.walk_rounds < 1) a.walk_rounds = 1;
        if (a.walk_refill < 1) a.walk_refill = 1;
        if (a.walk_refill > 64) a.walk_refill = 64;

    }
    // persistent grids: the dense kernels stride over their list, the walk kernels pull jobs
    const unsigned max_blocks = (unsigned)((items + kWfBlock - 1) / kWfBlock);
    // (the two walk kernels of neighbouring iterations run side by side: VSPG_WF_WALK_BLOCKS / VSPG_WF_SHADOW_BLOCKS = resident
    // workgroups per CU of each, within their launch bounds)
    static const int walk_blocks = [] { const char *e = getenv("VSPG_WF_WALK_BLOCKS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= kWfWalkWavesPerSimd ? v : kWfWalkWavesPerSimd; }();
    static const int shadow_blocks = [] { const char *e = getenv("VSPG_WF_SHADOW_BLOCKS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= kWfShadowWavesPerSimd ? v : kWfShadowWavesPerSimd; }();
    L.dense = (unsigned)r->num_cus * 8u;
    L.walk = (unsigned)r->num_cus * (unsigned)walk_blocks;
    L.swalk = (unsigned)r->num_cus * (unsigned)shadow_blocks;
    const int merged_blocks = [] { const char *e = getenv("VSPG_WF_MERGED_BLOCKS"); const int v = e ? atoi(e) : 0; return v >= 1 && v <= kWfMergedWavesPerSimd ? v : kWfMergedWavesPerSimd; }();  // (read per pass)
    L.mwalk = (unsigned)r->num_cus * (unsigned)merged_blocks;
    if (L.dense > max_blocks) L.dense = max_blocks;
    if (L.walk > max_blocks) L.walk = max_blocks;
    if (L.swalk > max_blocks) L.swalk = max_blocks;
    if (L.mwalk > max_blocks) L.mwalk = max_blocks;
    // The shadow walk of iteration i runs beside the distance walk of iteration i + 1, on the renderer's second stream; the
    // vertex kernel of iteration i + 1 waits for both (it adds the shadow walk's result first thing).  VSPG_WF_SERIAL=1 keeps
    // everything on the caller's stream (same results; for A/B runs and debugging).
    L.serial = [] { const char *e = getenv("VSPG_WF_SERIAL"); return e && *e && *e != '0'; }();  // (read per pass: a test flips it)
    if (!L.serial && !r->wf_stream2) {
        HIPCHK(hipStreamCreateWithFlags(&r->wf_stream2, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&r->wf_ev_vertex, hipEventDisableTiming));
        HIPCHK(hipEventCreateWithFlags(&r->wf_ev_shadow, hipEventDisableTiming));
    }
    // both walks of an iteration as ONE kernel where the job lists are short or a third kernel sits in the chain (boundary scenes,
    // guided pipelines); side by side on two streams for dense unguided clouds (k_wf_walk, vspg_wavefront.h: measured both ways).
    // VSPG_WF_MERGED=0|1 overrides (read per pass: tests and A/Bs flip it).
    L.merged = wf_merged_walks(r, guided);
    // job cursors of the merged walk kernel's stream: boundary scenes' short lists are dealt out to kWfSegs of them (wf_claim_refill;
    // VSPG_WF_SEGS=1|8 overrides, read per pass)
    L.segs = bnd ? kWfSegs : 1;
    if (const char *e = getenv("VSPG_WF_SEGS")) L.segs = atoi(e) > 1 ? kWfSegs : 1;
    L.s = s;
    L.s2 = L.serial ? s : r->wf_stream2;
    L.ev_vertex = r->wf_ev_vertex;
    L.ev_shadow = r->wf_ev_shadow;
    L.bnd = bnd;
    L.nds = r->prm.vspsamplingmethod != VSPG_VSP_RESAMPLING;
    L.emit = r->hscene.temperature != nullptr;
    L.maxdepth = r->prm.maxdepth;
    L.base_iters = base_iters;
    L.max_iters = max_iters;
    // (tolerance modes: vspg_renderer_set_arithmetic accepted the renderer only if vspg_fast.hip holds its instantiation)
    const int rc = r->arith == VSPG_ARITH_FAST_WEIGHTS ? vspg_arith1_wf_grid(&L, grey ? 1 : 0)
                   : r->arith == VSPG_ARITH_FAST     ? vspg_arith2_wf_grid(&L, grey ? 1 : 0)
                   : nvdb                            ? wf_dispatch_nvdb(L, guided, train, grey)
                                                     : wf_dispatch_grid(L, guided, train, grey);
    if (rc == WF_E_NOT_DRAINED)
        return fail(VSPG_ESCOPE, "paths still alive after the pipeline's iteration cap: more medium-boundary crossings per vertex than a pass provides for");
    if (rc != 0) return fail(VSPG_EHIP, std::string("the wavefront pipeline: ") + hipGetErrorName((hipError_t)rc));
    if (train) {
        hipLaunchKernelGGL(k_propagate, dim3((unsigned)((items + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a.train, a.rec_cap);
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// Profiling markers (trace ranges): with VSPG_ROCTX=1 in the environment every vspg_render_wave / vspg_post_process_step
// call is a roctx range (rocprofv3 --marker-trace shows waves and post-processing on the timeline).  libroctx64 is looked up at
// run time, so the library carries no link dependency on it and costs nothing when the variable is unset.
namespace {
struct RoctxApi {
    int (*push)(const char *) = nullptr;
    int (*pop)() = nullptr;
    RoctxApi() {
        const char *e = getenv("VSPG_ROCTX");
        if (!e || !*e || *e == '0') return;
        // rocprofv3 listens to the rocprofiler-sdk's roctx; roctracer's libroctx64 (same entry points) serves older tools
        void *h = nullptr;
        for (const char *name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "libroctx64.so", "libroctx64.so.4"})
            if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return;
        push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
        pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
        if (!push || !pop) push = nullptr, pop = nullptr;
    }
};
const RoctxApi g_roctx;
struct RoctxRange {
    explicit RoctxRange(const char *name) { if (g_roctx.push) g_roctx.push(name); }
    ~RoctxRange() { if (g_roctx.pop) g_roctx.pop(); }
};
}  // namespace

extern "C" {

int vspg_abi_version(void) { return VSPG_ABI_VERSION; }
const char *vspg_last_error(void) { return g_err.c_str(); }

void vspg_integrator_params_default(VspgIntegratorParams *p) {
    // GuidedVolPathVSPGIntegrator::Create defaults (guidedvolpathvspgintegrator.cpp:1263-1319)
    memset(p, 0, sizeof *p);
    p->maxdepth = 5;
    p->minrrdepth = 1;
    p->usenee = 1;
    p->surfaceguiding = 1;
    p->volumeguiding = 1;
    p->surfaceguidingtype = VSPG_GUIDE_RIS;
    p->volumeguidingtype = VSPG_GUIDE_MIS;
    p->vspguiding = 1;
    p->vspprimaryguiding = 1;
    p->vspsecondaryguiding = 1;
    p->vspmisratio = 0.5f;
    p->vspcriterion = VSPG_VSP_VARIANCE;
    p->vspsamplingmethod = VSPG_VSP_RESAMPLING;
    p->lightsampler = VSPG_LIGHTSAMPLER_BVH;
    p->guide_num_training_waves = 128;
    p->surfacerrguiding = 1;
    p->volumerrguiding = 1;
}

int vspg_camera_look_at(VspgCamera *cam, const float eye[3], const float look[3], const float up[3], float fov_degrees,
                        int xres, int yres) {
    if (!cam || xres <= 0 || yres <= 0) return fail(VSPG_EINVAL, "bad camera arguments");
    double e[3], f[3], u[3];
    for (int i = 0; i < 3; ++i) { e[i] = eye[i]; f[i] = (double)look[i] - eye[i]; u[i] = up[i]; }
    double fl = std::sqrt(f[0] * f[0] + f[1] * f[1] + f[2] * f[2]);
    double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
    if (fl == 0 || ul == 0) return fail(VSPG_EINVAL, "degenerate LookAt");
    for (int i = 0; i < 3; ++i) { f[i] /= fl; u[i] /= ul; }
    // pbrt LookAt: right = Normalize(Cross(Normalize(up), dir)); newUp = Cross(dir, right)
    double r[3] = {u[1] * f[2] - u[2] * f[1], u[2] * f[0] - u[0] * f[2], u[0] * f[1] - u[1] * f[0]};
    double rl = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    if (rl == 0) return fail(VSPG_EINVAL, "up vector parallel to view direction");
    for (int i = 0; i < 3; ++i) r[i] /= rl;
    double nu[3] = {f[1] * r[2] - f[2] * r[1], f[2] * r[0] - f[0] * r[2], f[0] * r[1] - f[1] * r[0]};
    for (int i = 0; i < 3; ++i) {
        cam->origin[i] = (float)e[i];
        cam->right[i] = (float)r[i];
        cam->up[i] = (float)nu[i];
        cam->fwd[i] = (float)f[i];
    }
    // screen window spans [-1,1] on the shorter axis (cameras.cpp:474-489); raster y points down
    double aspect = (double)xres / (double)yres;
    double wx = aspect > 1 ? aspect : 1.0, wy = aspect > 1 ? 1.0 : 1.0 / aspect;
    double th = std::tan(fov_degrees * 3.14159265358979323846 / 360.0);
    cam->sx = (float)(2.0 * wx * th / xres);
    cam->ox = (float)(-wx * th);
    cam->sy = (float)(-2.0 * wy * th / yres);
    cam->oy = (float)(wy * th);
    return 0;
}

int vspg_transform_inverse(const float m[16], float inv[16]) {
    if (!m || !inv) return fail(VSPG_EINVAL, "null argument");
    if (m[12] != 0 || m[13] != 0 || m[14] != 0 || m[15] != 1) return fail(VSPG_EINVAL, "affine matrices only (last row 0 0 0 1)");
    const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
    const double A = e * i - f * h, B = -(d * i - f * g), Cc = d * h - e * g;
    const double det = a * A + b * B + c * Cc;
    if (det == 0 || det != det) return fail(VSPG_EINVAL, "singular matrix");
    double r[9] = {A / det, -(b * i - c * h) / det, (b * f - c * e) / det, B / det, (a * i - c * g) / det, -(a * f - c * d) / det,
                   Cc / det, -(a * h - b * g) / det, (a * e - b * d) / det};
    const double t[3] = {m[3], m[7], m[11]};
    for (int row = 0; row < 3; ++row) {
        for (int col = 0; col < 3; ++col) inv[4 * row + col] = (float)r[3 * row + col];
        inv[4 * row + 3] = (float)-(r[3 * row] * t[0] + r[3 * row + 1] * t[1] + r[3 * row + 2] * t[2]);
    }
    inv[12] = inv[13] = inv[14] = 0.f;
    inv[15] = 1.f;
    return 0;
}

static void quad(VspgQuad *q, float px, float py, float pz, float ax, float ay, float az, float bx, float by, float bz,
                 float kd, float lr, float lg, float lb) {
    memset(q, 0, sizeof *q);
    q->p00[0] = px; q->p00[1] = py; q->p00[2] = pz;
    q->e1[0] = ax; q->e1[1] = ay; q->e1[2] = az;
    q->e2[0] = bx; q->e2[1] = by; q->e2[2] = bz;
    q->Kd[0] = q->Kd[1] = q->Kd[2] = kd;
    q->Le[0] = lr; q->Le[1] = lg; q->Le[2] = lb;
}

int vspg_scene_fog_box(VspgScene *s, int xres, int yres) {
    if (!s) return fail(VSPG_EINVAL, "null scene");
    memset(s, 0, sizeof *s);
    const float k = 0.73f;
    s->n_quads = 7;  // normals e1 x e2 face into the box
    quad(&s->quads[0], -1, -1, -1, 0, 0, 2, 2, 0, 0, k, 0, 0, 0);   // floor
    quad(&s->quads[1], -1, 1, -1, 2, 0, 0, 0, 0, 2, k, 0, 0, 0);    // ceiling
    quad(&s->quads[2], -1, -1, 1, 0, 2, 0, 2, 0, 0, k, 0, 0, 0);    // back  z=+1
    quad(&s->quads[3], -1, -1, -1, 2, 0, 0, 0, 2, 0, k, 0, 0, 0);   // front z=-1
    quad(&s->quads[4], -1, -1, -1, 0, 2, 0, 0, 0, 2, k, 0, 0, 0);   // left
    quad(&s->quads[5], 1, -1, -1, 0, 0, 2, 0, 2, 0, k, 0, 0, 0);    // right
    quad(&s->quads[6], -0.25f, 0.999f, -0.25f, 0.5f, 0, 0, 0, 0, 0.5f, 0, 17, 12, 4);  // ceiling light, faces -y
    const float eye[3] = {0, 0, -0.95f}, look[3] = {0, 0, 0}, up[3] = {0, 1, 0};
    int rc = vspg_camera_look_at(&s->camera, eye, look, up, 60.f, xres, yres);
    if (rc) return rc;
    s->medium.type = VSPG_MEDIUM_HOMOGENEOUS;
    for (int i = 0; i < 3; ++i) { s->medium.sigma_a[i] = 0.05f; s->medium.sigma_s[i] = 0.45f; }
    s->medium.g = 0.f;
    return 0;
}

int vspg_renderer_create(const VspgScene *scene, const VspgIntegratorParams *params, const VspgRenderConfig *cfg,
                         VspgRenderer **out) {
    if (!out) return fail(VSPG_EINVAL, "null out pointer");
    *out = nullptr;
    int rc = validate(scene, params, cfg);
    if (rc) return rc;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev == 0)
        return fail(VSPG_ENODEVICE, std::string("no HIP device available (") + hipGetErrorName(e) + "); this library has no CPU fallback");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(VSPG_EINVAL, "device ordinal out of range");
    HIPCHK(hipSetDevice(cfg->device));
    VspgRenderer *r = new VspgRenderer();
    r->scene = *scene;
    r->prm = *params;
    if (r->prm.rrguiding) r->prm.minrrdepth = 1;  // guidedvolpathvspgintegrator.cpp:195-197
    r->cfg = *cfg;
    if (r->cfg.shard_count < 1) { r->cfg.shard_count = 1; r->cfg.shard_index = 0; }
    build_dscene(r->scene, r->prm, r->cfg, &r->hscene);
    {
        auto same = [](const float *v) { return std::memcmp(&v[0], &v[1], 4) == 0 && std::memcmp(&v[1], &v[2], 4) == 0; };
        const VspgMedium &m = r->scene.medium;
        r->medium_grey = m.type != VSPG_MEDIUM_NONE && same(m.sigma_a) && same(m.sigma_s) && same(m.Le) && !m.temperature && !getenv("VSPG_NO_GREY");
        r->surfaces_grey = !getenv("VSPG_NO_GREY_KD");
        r->null_zero = m.type == VSPG_MEDIUM_HOMOGENEOUS && !getenv("VSPG_NO_NULLZERO");
        for (int k = 0; k < 3; ++k) r->null_zero = r->null_zero && !(r->hscene.sigma_n_raw[k] > 0) && r->hscene.sigma_n_raw[k] == r->hscene.sigma_n_raw[k];
        for (int i = 0; i < r->hscene.n_quads; ++i) r->surfaces_grey = r->surfaces_grey && same(r->hscene.quads[i].Kd);
    }
    r->npix = (size_t)cfg->xres * cfg->yres;
#define CK(expr)                                                                                                  \
    do {                                                                                                          \
        hipError_t e2 = (expr);                                                                                   \
        if (e2 != hipSuccess) {                                                                                   \
            vspg_renderer_destroy(r);                                                                             \
            return fail(VSPG_EHIP, std::string(#expr) + ": " + hipGetErrorName(e2));                              \
        }                                                                                                         \
    } while (0)
    if (scene->medium.type == VSPG_MEDIUM_GRID || scene->medium.type == VSPG_MEDIUM_NANOVDB) {
        const size_t n = (size_t)scene->medium.nx * scene->medium.ny * scene->medium.nz;
        std::vector<float> maj = scene->medium.type == VSPG_MEDIUM_GRID ? build_majorant_grid(scene->medium)
                                                                        : build_majorant_grid_nvdb(scene->medium);
        CK(hipMalloc(&r->density, n * sizeof(float)));
        CK(hipMemcpy(r->density, scene->medium.density, n * sizeof(float), hipMemcpyHostToDevice));
        CK(hipMalloc(&r->majorant, maj.size() * sizeof(float)));
        CK(hipMemcpy(r->majorant, maj.data(), maj.size() * sizeof(float), hipMemcpyHostToDevice));
        r->hscene.density = nullptr;
        r->hscene.majorant = r->majorant;
        {   // octet bricks (see DScene): flags on the device, slot numbering on the host, fill on the device
            const int nx = scene->medium.nx, ny = scene->medium.ny, nz = scene->medium.nz;
            const int bnx = (nx + 1 + 7) / 8, bny = (ny + 1 + 7) / 8, bnz = (nz + 1 + 7) / 8;
            const size_t nb = (size_t)bnx * bny * bnz;
            int32_t *dflags = nullptr, *dactive = nullptr;
            CK(hipMalloc(&dflags, nb * sizeof(int32_t)));
            hipLaunchKernelGGL(k_brick_flags, dim3((unsigned)nb), dim3(kBlock), 0, 0, r->density, nx, ny, nz, bnx, bny, dflags);
            std::vector<int32_t> flags(nb), index(nb), active;
            hipError_t ef = hipMemcpy(flags.data(), dflags, nb * sizeof(int32_t), hipMemcpyDeviceToHost);
            (void)hipFree(dflags);
            CK(ef);
            // Dense bricks: when storing EVERY brick fits the budget (16 KB per 8^3 voxels: 0.5 GB for 256^3, 4.3 GB for 512^3 of this
            // part's 288 GB) the index is dropped and a brick's slot is its position in the grid -- a density query is then one
            // memory round trip, not two dependent ones (measured: 11.4 -> 10.9 ms per 1080p cloud wave).  Sparse assets beyond the
            // budget keep the index (empty bricks cost no storage and no fetch).  VSPG_DENSE_BRICKS=0|1 forces either.
            const char *dense_env = getenv("VSPG_DENSE_BRICKS");
            const size_t dense_budget = (size_t)8 << 30;
            const bool dense = dense_env ? dense_env[0] == '1' : nb * 512 * 2 * sizeof(float4) <= dense_budget;
            for (size_t b = 0; b < nb; ++b) {
                const bool keep = dense || flags[b];
                index[b] = keep ? (int32_t)active.size() : -1;
                if (keep) active.push_back((int32_t)b);
            }
            r->n_bricks = active.size();
            if (!dense) {
                CK(hipMalloc(&r->brick_index, nb * sizeof(int32_t)));
                CK(hipMemcpy(r->brick_index, index.data(), nb * sizeof(int32_t), hipMemcpyHostToDevice));
            }
            CK(hipMalloc(&r->octets, (r->n_bricks ? r->n_bricks : 1) * 512 * 2 * sizeof(float4)));
            if (r->n_bricks) {
                CK(hipMalloc(&dactive, r->n_bricks * sizeof(int32_t)));
                hipError_t ea = hipMemcpy(dactive, active.data(), r->n_bricks * sizeof(int32_t), hipMemcpyHostToDevice);
                if (ea == hipSuccess) {
                    hipLaunchKernelGGL(k_brick_fill, dim3((unsigned)r->n_bricks), dim3(512), 0, 0, r->density, nx, ny, nz, bnx, bny, dactive, r->octets);
                    ea = hipDeviceSynchronize();
                }
                (void)hipFree(dactive);
                CK(ea);
            }
            (void)hipFree(r->density);  // the kernels read the bricks only
            r->density = nullptr;
            r->hscene.brick_index = r->brick_index;
            r->hscene.octets = r->octets;
            r->hscene.bnx = bnx; r->hscene.bny = bny; r->hscene.bnz = bnz;
        }
        r->scene.medium.density = nullptr;  // the host array belongs to the caller
        const VspgMedium &m = scene->medium;
        if (m.temperature) {  // as many samples as the density grid (media.cpp:283-288; the .nvdb reader checks the bounding boxes)
            CK(hipMalloc(&r->temperature, n * sizeof(float)));
            CK(hipMemcpy(r->temperature, m.temperature, n * sizeof(float), hipMemcpyHostToDevice));
            r->hscene.temperature = r->temperature;
            r->hscene.temperature_offset = m.temperature_offset;
            r->hscene.temperature_scale = m.temperature_scale;
            r->hscene.nvdb_le_scale = m.nvdb_le_scale;
        }
        // isEmissive = temperatureGrid ? true : Le_spec.MaxValue() > 0 (media.cpp:261)
        if (m.type == VSPG_MEDIUM_GRID && (m.temperature || m.Le[0] != 0 || m.Le[1] != 0 || m.Le[2] != 0)) {
            const float one = 1.f;  // "Lescale" absent: SampledGrid({1}, 1, 1, 1) (media.cpp:319-320)
            const float *src = m.le_scale ? m.le_scale : &one;
            const int lx = m.le_scale ? m.le_nx : 1, ly = m.le_scale ? m.le_ny : 1, lz = m.le_scale ? m.le_nz : 1;
            const size_t ln = (size_t)lx * ly * lz;
            CK(hipMalloc(&r->le_scale, ln * sizeof(float)));
            CK(hipMemcpy(r->le_scale, src, ln * sizeof(float), hipMemcpyHostToDevice));
            r->hscene.le_scale = r->le_scale;
            r->hscene.le_nx = lx; r->hscene.le_ny = ly; r->hscene.le_nz = lz;
        }
        r->scene.medium.le_scale = nullptr;
        r->scene.medium.temperature = nullptr;  // (the host array belongs to the caller)
    }
    if (scene->n_triangles > 0) {
        std::vector<DTri> all;
        all.reserve(scene->n_triangles);
        for (int i = 0; i < scene->n_triangles; ++i) {
            DTri T;
            if (derive_triangle(scene->tri_p + 9 * (size_t)i, scene->tri_kd, i, &T, scene->tri_flags)) all.push_back(T);
        }
        if (!all.empty()) {
            bvhbuild::Builder B;
            B.order.resize(all.size());
            B.tbox.resize(all.size());
            B.cen.resize(3 * all.size());
            for (size_t i = 0; i < all.size(); ++i) {
                B.order[i] = (int)i;
                bvhbuild::Box b = bvhbuild::empty_box();
                bvhbuild::grow(b, all[i].p0); bvhbuild::grow(b, all[i].p1); bvhbuild::grow(b, all[i].p2);
                B.tbox[i] = b;
                for (int k = 0; k < 3; ++k) B.cen[3 * i + k] = 0.5f * (b.lo[k] + b.hi[k]);
            }
            B.nodes.reserve(2 * all.size());
            B.build(0, (int)all.size());
            std::vector<DTri> sorted(all.size());
            for (size_t i = 0; i < all.size(); ++i) sorted[i] = all[B.order[i]];
            CK(hipMalloc(&r->tris, sorted.size() * sizeof(DTri)));
            CK(hipMemcpy(r->tris, sorted.data(), sorted.size() * sizeof(DTri), hipMemcpyHostToDevice));
            int depth4 = 0;
            B.collapse(0, &depth4);
            if (3 * depth4 + 1 > kBvhStack) {  // a degenerate soup made the SAH tree too deep for the traversal's stack: median splits
                bvhbuild::Builder B2;
                B2.balanced = true;
                B2.p = B.p; B2.tbox = B.tbox; B2.cen = B.cen;
                B2.order.resize(all.size());
                for (size_t i = 0; i < all.size(); ++i) B2.order[i] = (int)i;
                B2.nodes.reserve(2 * all.size());
                B2.build(0, (int)all.size());
                B2.collapse(0, &depth4);
                if (3 * depth4 + 1 > kBvhStack) { vspg_renderer_destroy(r); return fail(VSPG_ESCOPE, "triangle soup too large for the BVH traversal's stack"); }
                B.order = B2.order;
                B.nodes4 = B2.nodes4;
                for (size_t i = 0; i < all.size(); ++i) sorted[i] = all[B.order[i]];
                CK(hipMemcpy(r->tris, sorted.data(), sorted.size() * sizeof(DTri), hipMemcpyHostToDevice));
            }
            CK(hipMalloc(&r->bvh, B.nodes4.size() * sizeof(DBvh4Node)));
            CK(hipMemcpy(r->bvh, B.nodes4.data(), B.nodes4.size() * sizeof(DBvh4Node), hipMemcpyHostToDevice));
            r->hscene.n_tris = (int32_t)sorted.size();
            r->hscene.n_bvh_nodes = (int32_t)B.nodes4.size();
            r->hscene.tris = r->tris;
            r->hscene.bvh = r->bvh;
        }
        r->scene.tri_p = nullptr;  // the host arrays belong to the caller
        r->scene.tri_kd = nullptr;
        r->scene.tri_flags = nullptr;
    }
    CK(hipMalloc(&r->dscene, sizeof(DScene)));
    CK(hipMemcpy(r->dscene, &r->hscene, sizeof(DScene), hipMemcpyHostToDevice));
    CK(hipMalloc(&r->film, r->npix * sizeof(float4)));
    CK(hipMemset(r->film, 0, r->npix * sizeof(float4)));
    CK(hipMalloc(&r->isg_stats, r->npix * VSPG_ISG_STATS * sizeof(float)));
    CK(hipMemset(r->isg_stats, 0, r->npix * VSPG_ISG_STATS * sizeof(float)));
    CK(hipMalloc(&r->vsp, r->npix * sizeof(float)));
    CK(hipMemset(r->vsp, 0, r->npix * sizeof(float)));
    CK(hipMalloc(&r->counters, kNumCounters * sizeof(unsigned long long)));
    CK(hipMemset(r->counters, 0, kNumCounters * sizeof(unsigned long long)));
    CK(hipMalloc(&r->work_head, 2 * sizeof(unsigned int)));
    CK(hipMemset(r->work_head, 0, 2 * sizeof(unsigned int)));
    CK(hipMalloc(&r->work_head8_raw, 4 * (size_t)kWg3HeadSetBytes));
    CK(hipMemset(r->work_head8_raw, 0, 4 * (size_t)kWg3HeadSetBytes));
    r->work_head8 = reinterpret_cast<unsigned int *>((reinterpret_cast<uintptr_t>(r->work_head8_raw) + 2 * (uintptr_t)kWg3HeadSetBytes - 1) &
                                                     ~(2 * (uintptr_t)kWg3HeadSetBytes - 1));  // (the sibling set is at address ^ kWg3HeadSetBytes)
    {
        hipDeviceProp_t prop;
        CK(hipGetDeviceProperties(&prop, cfg->device));
        r->num_cus = prop.multiProcessorCount;
    }
    if (r->prm.rrguiding) {
        CK(hipMalloc(&r->contrib, r->npix * sizeof(float)));
        CK(hipMemset(r->contrib, 0, r->npix * sizeof(float)));
        r->hscene.contrib = r->contrib;
        CK(hipMemcpy(r->dscene, &r->hscene, sizeof(DScene), hipMemcpyHostToDevice));
    }
    // calculateTrBuffer (guidedvolpathvspgintegrator.cpp:190-193); trBufferLoad is vspg_renderer_set_tr_buffer
    if (r->prm.storeTrBuffer || (r->prm.vspguiding && r->prm.vspprimaryguiding && r->prm.vspsamplingmethod == VSPG_VSP_NDS &&
                                 r->prm.collisionProbabilityBias)) {
        CK(hipMalloc(&r->tr_rgb, r->npix * 3 * sizeof(float)));
        CK(hipMemset(r->tr_rgb, 0, r->npix * 3 * sizeof(float)));
        CK(hipMalloc(&r->tr_spp, r->npix * sizeof(int32_t)));
        CK(hipMemset(r->tr_spp, 0, r->npix * sizeof(int32_t)));
        r->hscene.tr_rgb = r->tr_rgb;
        r->hscene.tr_spp = r->tr_spp;
        r->hscene.tr_calc = 1;
        CK(hipMemcpy(r->dscene, &r->hscene, sizeof(DScene), hipMemcpyHostToDevice));
    }
    // guideTraining (guidedvolpathvspgintegrator.cpp:109).  The reference also trains when only the guided-RR
    // flags are set (they default to true); this build trains iff the field will be queried.
    if (wants_guiding(r->prm)) {
        r->training = wants_training(r->prm);
        r->field_set = true;
        for (int f = 0; f < 2; ++f) {
            CK(hipMalloc(&r->fnodes[f], sizeof(VspgKdNode) * kTrainCapNodes));
            CK(hipMemset(r->fnodes[f], 0, sizeof(VspgKdNode) * kTrainCapNodes));
            CK(hipMalloc(&r->fregions[f], sizeof(VspgFieldRegion) * kTrainCapRegions));
            CK(hipMemset(r->fregions[f], 0, sizeof(VspgFieldRegion) * kTrainCapRegions));
            CK(hipMalloc(&r->rstats[f], sizeof(RegionStats) * kTrainCapRegions));
            CK(hipMemset(r->rstats[f], 0, sizeof(RegionStats) * kTrainCapRegions));
            const VspgKdNode root = {0.f, 3u};  // one leaf -> region 0, untrained (n_lobes 0)
            CK(hipMemcpy(r->fnodes[f], &root, sizeof root, hipMemcpyHostToDevice));
            CK(hipMalloc(&r->faux[f], sizeof(float) * 2 * GK * kTrainCapRegions));
            CK(hipMemset(r->faux[f], 0, sizeof(float) * 2 * GK * kTrainCapRegions));
            CK(hipMalloc(&r->flobes[f], sizeof(float4) * kRegionLobeQuads * kTrainCapRegions));
            CK(hipMemset(r->flobes[f], 0, sizeof(float4) * kRegionLobeQuads * kTrainCapRegions));
            r->hscene.field[f] = DField{1, 1, r->fnodes[f], r->fregions[f], r->faux[f], r->flobes[f]};
        }
        CK(hipMemcpy(r->dscene, &r->hscene, sizeof(DScene), hipMemcpyHostToDevice));
    }
    // segment records, radiance samples and the update's scratch exist only for a renderer that trains (an
    // rrguiding-only renderer queries nothing and records nothing)
    if (r->training) {
        r->segbuf_items = (size_t)((cfg->xres + 7) / 8) * (size_t)((cfg->yres + 7) / 8) * 64;  // the work items of a 1-spp wave
        CK(hipMalloc(&r->segbuf, r->segbuf_items * (size_t)train_rec_capacity(r->prm.maxdepth) * SG_FLOATS * sizeof(float)));
        CK(hipMalloc(&r->seg_count, r->segbuf_items * sizeof(int)));
        r->sample_capacity = (unsigned long long)r->npix * (unsigned long long)(r->prm.maxdepth + 1);
        if (r->sample_capacity > (1ull << 26)) r->sample_capacity = 1ull << 26;
        CK(hipMalloc(&r->samples, r->sample_capacity * sizeof(VspgTrainSample)));
        CK(hipMalloc(&r->train_counters, 4 * sizeof(unsigned long long)));
        CK(hipMemset(r->train_counters, 0, 4 * sizeof(unsigned long long)));
        CK(hipMalloc(&r->train_acc, (size_t)kTrainKeys * kStatFloats * sizeof(float)));
        CK(hipMalloc(&r->train_sumw, 2 * sizeof(float)));
        CK(hipMalloc(&r->train_reg, r->sample_capacity * sizeof(int)));
        CK(hipMalloc(&r->train_order, r->sample_capacity * sizeof(unsigned int)));
        CK(hipMalloc(&r->train_hist, (size_t)kTrainKeys * sizeof(unsigned int)));
        CK(hipMalloc(&r->train_cursor, (size_t)kTrainKeys * sizeof(unsigned int)));
        CK(hipMalloc(&r->train_nsplit, 2 * sizeof(int)));
        CK(hipMalloc(&r->train_nsorted, sizeof(unsigned int)));
    }
#undef CK
    *out = r;
    return 0;
}

int vspg_renderer_destroy(VspgRenderer *r) {
    if (!r) return 0;
    (void)hipSetDevice(r->cfg.device);
    if (r->dscene) (void)hipFree(r->dscene);
    if (r->film) (void)hipFree(r->film);
    if (r->isg_stats) (void)hipFree(r->isg_stats);
    if (r->contrib) (void)hipFree(r->contrib);
    if (r->tr_rgb) (void)hipFree(r->tr_rgb);
    if (r->tr_spp) (void)hipFree(r->tr_spp);
    if (r->vsp) (void)hipFree(r->vsp);
    if (r->counters) (void)hipFree(r->counters);
    if (r->work_head) (void)hipFree(r->work_head);
    if (r->work_head8_raw) (void)hipFree(r->work_head8_raw);
    for (int f = 0; f < 2; ++f) {
        if (r->fnodes[f]) (void)hipFree(r->fnodes[f]);
        if (r->fregions[f]) (void)hipFree(r->fregions[f]);
        if (r->faux[f]) (void)hipFree(r->faux[f]);
        if (r->flobes[f]) (void)hipFree(r->flobes[f]);
    }
    for (int f = 0; f < 2; ++f)
        if (r->rstats[f]) (void)hipFree(r->rstats[f]);
    if (r->segbuf) (void)hipFree(r->segbuf);
    if (r->seg_count) (void)hipFree(r->seg_count);
    if (r->samples) (void)hipFree(r->samples);
    if (r->train_counters) (void)hipFree(r->train_counters);
    if (r->train_acc) (void)hipFree(r->train_acc);
    if (r->train_sumw) (void)hipFree(r->train_sumw);
    if (r->train_reg) (void)hipFree(r->train_reg);
    if (r->train_order) (void)hipFree(r->train_order);
    if (r->train_hist) (void)hipFree(r->train_hist);
    if (r->train_cursor) (void)hipFree(r->train_cursor);
    if (r->train_nsorted) (void)hipFree(r->train_nsorted);
    if (r->train_nsplit) (void)hipFree(r->train_nsplit);
    if (r->tris) (void)hipFree(r->tris);
    if (r->bvh) (void)hipFree(r->bvh);
    for (int k = 0; k < 2; ++k) if (r->wave_samples[k]) (void)hipFree(r->wave_samples[k]);
    if (r->ws_event) (void)hipEventDestroy(r->ws_event);
    if (r->wf_pool) (void)hipFree(r->wf_pool);
    if (r->wf_lists) (void)hipFree(r->wf_lists);
    if (r->wf_iters) (void)hipFree(r->wf_iters);
    if (r->wf_ev_vertex) (void)hipEventDestroy(r->wf_ev_vertex);
    if (r->wf_ev_shadow) (void)hipEventDestroy(r->wf_ev_shadow);
    if (r->wf_stream2) (void)hipStreamDestroy(r->wf_stream2);
    if (r->density) (void)hipFree(r->density);
    if (r->brick_index) (void)hipFree(r->brick_index);
    if (r->octets) (void)hipFree(r->octets);
    if (r->le_scale) (void)hipFree(r->le_scale);
    if (r->temperature) (void)hipFree(r->temperature);
    if (r->majorant) (void)hipFree(r->majorant);
    delete r;
    return 0;
}

// scheduler: "wg" = workgroup-level wavefront kernel, "lane" = per-lane persistent kernel.  Default: wg for
// homogeneous media (dense, equally long phases); lane for grid media, whose tracking walks have very
// different lengths per path -- a phase lasts as long as its longest walk, while the per-lane kernel
// refills a lane the moment its path ends (measured on the 256^3 cloud stand-in: 30.5 vs 38.6 ms per wave)
// -- and for guided builds.  VSPG_KERNEL=wg|lane overrides (unguided builds only).
// guided renders on the workgroup kernel (VSPG_KERNEL=wg): a trained (or loaded) field being QUERIED over a homogeneous medium
// -- training launches record path segments and guided Russian roulette carries per-pixel state, both on the per-lane kernel.
// Opt-in, not the default: measured on MI355X (1080p fog box, reference-default options) the per-lane kernel
// runs a guided wave in 2.33 ms, this one in 2.58 ms (384-path pool, kd nodes in L2) / 2.93 ms (320-path pool + the upper kd
// levels in LDS): it issues 18 % fewer vector instructions at 68 % instead of 54 % lane utilisation, but the guided vertex code
// needs ~240 registers either way (2 waves per SIMD) and at that occupancy the phase barriers cost more than the compaction saves.
// VSPG_KERNEL=wg|lane|wf picks a path kernel where several serve a configuration (tests, A/B runs); empty == unset
static const char *kernel_env() {
    const char *e = getenv("VSPG_KERNEL");
    return e && *e ? e : nullptr;
}
// Medium boundaries, interface materials and spheres are served by the full-scene code paths (per-lane kernel, pipeline)
static bool has_boundaries_or_spheres(const VspgRenderer *r) { return r->hscene.has_boundaries != 0 || r->hscene.n_spheres > 0; }
static bool uses_wg_guided(const VspgRenderer *r) {
    // round 3: the workgroup kernel's guided vertex (vspg_guided_wg.h, four waves per SIMD) is the DEFAULT for a trained or
    // loaded field over a homogeneous medium in a rectangle scene; VSPG_KERNEL=lane selects the per-lane kernel (tests compare
    // the two).  Guided Russian roulette, triangles / infinite lights and non-uniform light samplers stay per-lane.
    const char *kenv = kernel_env();
    if (kenv && strcmp(kenv, "wg") != 0) return false;
    if (has_boundaries_or_spheres(r)) return false;
    return wants_guiding(r->prm) && !r->prm.rrguiding && r->scene.medium.type == VSPG_MEDIUM_HOMOGENEOUS &&
           r->hscene.n_tris == 0 && r->hscene.n_inf == 0 && r->hscene.lsamp.mode == VSPG_LIGHTSAMPLER_UNIFORM;
}
// guided renders over a homogeneous medium: the grey / zero-null-coefficient / rectangle-scene instantiation of the per-lane
// kernel (the segment half of the loop sheds the same per-channel work as the headline kernel's instantiation)
static bool guided_grey_simple(const VspgRenderer *r) {
    return r->scene.medium.type == VSPG_MEDIUM_HOMOGENEOUS && r->medium_grey && r->surfaces_grey && r->null_zero && r->hscene.n_tris == 0 &&
           r->hscene.n_inf == 0 && r->hscene.lsamp.mode == VSPG_LIGHTSAMPLER_UNIFORM && !has_boundaries_or_spheres(r) && !getenv("VSPG_NO_GREY_GUIDED");
}
// a scene of rectangles and area lights only, one medium filling it: what the workgroup kernel's specialised instantiations are built for
static bool scene_is_simple(const VspgRenderer *r) {
    return r->hscene.n_tris == 0 && r->hscene.n_inf == 0 && r->hscene.lsamp.mode == VSPG_LIGHTSAMPLER_UNIFORM && !has_boundaries_or_spheres(r);
}
// Round 4: everything else over a homogeneous medium, unguided -- triangles (BVH), spheres, infinite lights, power / BVH light samplers,
// medium boundaries -- runs the workgroup kernel's FULL-scene instantiation (k_render_wave_wg2<HomogeneousMedium>) instead of the
// per-lane kernel; VSPG_KERNEL=lane keeps the per-lane kernel (tests compare the two).
static bool uses_wg_full(const VspgRenderer *r) {
    const char *kenv = kernel_env();
    if (kenv && strcmp(kenv, "wg") != 0) return false;
    return r->scene.medium.type == VSPG_MEDIUM_HOMOGENEOUS && !wants_guiding(r->prm) && !scene_is_simple(r);
}
static bool uses_wg_kernel(const VspgRenderer *r) {
    const bool grid = r->scene.medium.type == VSPG_MEDIUM_GRID;
    const bool nvdb = r->scene.medium.type == VSPG_MEDIUM_NANOVDB;
    const bool guided = wants_guiding(r->prm);
    if (uses_wg_guided(r)) return true;
    if (uses_wg_full(r)) return true;
    const char *kenv = kernel_env();
    const bool want_wg = kenv ? strcmp(kenv, "wg") == 0 : !grid;
    // (the TrBuffer's running mean needs a pixel's samples in order: the per-lane kernel owns a pixel per launch)
    // triangle hits carry a per-hit error bound the LDS pool record has no room for, and the kernel's homogeneous instantiations
    // are built for rectangle scenes with area lights only (HomogeneousMediumT::kSimpleScene)
    if (r->hscene.n_tris > 0 || r->hscene.n_inf > 0 || has_boundaries_or_spheres(r)) return false;
    if (r->hscene.lsamp.mode != VSPG_LIGHTSAMPLER_UNIFORM) return false;  // power / BVH picks of a multi-light scene: the full-scene kernels
    // (a temperature grid's blackbody emission needs the path's wavelength sample, which k_render_wave_wg's pool record does not carry)
    if (grid && r->hscene.temperature) return false;
    return !guided && !nvdb && want_wg && !(kenv && strcmp(kenv, "lane") == 0) && !(r->hscene.tr_calc && grid);
}
// "wf" = the multi-kernel wavefront pipeline (vspg_wavefront.h): heterogeneous media whose every segment runs the
// resampling routine (the reference's default vspsamplingmethod), unguided.  Default for those; VSPG_KERNEL=lane|wg
// selects the single-kernel schedulers instead (kept for the guided / NDS configurations and as cross-checks).
static bool uses_wf_pipeline(const VspgRenderer *r) {
    const bool het = r->scene.medium.type == VSPG_MEDIUM_GRID || r->scene.medium.type == VSPG_MEDIUM_NANOVDB;
    const char *kenv = kernel_env();
    if (kenv && strcmp(kenv, "wf") != 0) return false;
    // guided builds too, training passes included (segment recording in the dense kernels), guided Russian roulette (round 3:
    // the vertex kernel reads the pixel's contribution estimate) and, in its own shape, NDS / NDS+ (k_wf_segment_vertex)
    return het;
}
// Which scheduler of the workgroup kernel: k_render_wave_wg2 (tiles from a global head, samples parked and
// resolved by the next launch, two barriers) serves every homogeneous configuration since round 3 -- with the shared tile head it
// beat k_render_wave_wg (film flush between the phases, three barriers) on the unguided workload too (0.776 against 0.808 ms);
// VSPG_WG_SCHED=1 selects k_render_wave_wg for the unguided instantiations (tests compare the two), grid media under
// VSPG_KERNEL=wg stay on it.
static bool uses_wg2(const VspgRenderer *r) {
    if (!uses_wg_kernel(r) || r->scene.medium.type == VSPG_MEDIUM_GRID) return false;
    if (uses_wg_guided(r) || uses_wg_full(r)) return true;
    const char *e = getenv("VSPG_WG_SCHED");
    return !(e && e[0] == '1');
}
// Round 5: the barrier-free scheduler (k_render_wave_wg3, vspg_wg3.h: ring queues in LDS, every wavefront its own scheduler) serves
// whatever k_render_wave_wg2 served; VSPG_WG_SCHED=2 keeps k_render_wave_wg2 (tests compare the three schedulers).
// It lives on the pool's slack over the workgroup's lanes (HISTORY round 5), which the larger records do not leave: the guided /
// training instantiations (544 / 448 paths for 512 lanes: reference-default trained wave 1.45 ms against 1.42) and the full-scene one
// (512) stay on k_render_wave_wg2.
static bool uses_wg3(const VspgRenderer *r) {
    if (!uses_wg2(r) || uses_wg_guided(r) || uses_wg_full(r)) return false;
    const char *e = getenv("VSPG_WG_SCHED");
    return !(e && e[0] == '2');
}
static bool arith_covered(const VspgRenderer *r) {
    if (kernel_env() || getenv("VSPG_WG_SCHED")) return false;  // (the cross-check kernels exist in exact arithmetic only)
    if (wants_guiding(r->prm) || r->hscene.tr_calc) return false;
    if (uses_wf_pipeline(r))
        return r->scene.medium.type == VSPG_MEDIUM_GRID && r->prm.vspsamplingmethod == VSPG_VSP_RESAMPLING && r->hscene.temperature == nullptr;
    return uses_wg3(r) && !uses_wg_guided(r) && !uses_wg_full(r);
}
// The samples a one-sample wg2 launch parked are resolved by the next such launch; anything else that reads or writes the film or
// the image-space statistics calls this first (VSPG_WG2_DEFER=0: every launch resolves its own samples at once).
static bool wg2_defer_enabled() {  // (read per launch: a test flips it)
    const char *e = getenv("VSPG_WG2_DEFER");
    return !(e && e[0] == '0');
}
// (the parking launch ran on ws_stream; work on any other stream that reads its samples is ordered behind it by ws_event)
static int order_after_parking(VspgRenderer *r, hipStream_t s) {
    if (r->ws_parked && r->ws_event && s != r->ws_stream) HIPCHK(hipStreamWaitEvent(s, r->ws_event, 0));
    return 0;
}
static int flush_parked_samples(VspgRenderer *r, hipStream_t s) {
    if (!r->ws_parked) return 0;
    if (const int rc = order_after_parking(r, s)) return rc;
    hipLaunchKernelGGL(k_film_resolve, dim3((unsigned)((r->npix + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, r->npix,
                       r->wave_samples[r->ws_cur ^ 1], r->film, r->isg_stats);
    HIPCHK(hipGetLastError());
    r->ws_parked = false;
    return 0;
}
static const char *kernel_name_exact(VspgRenderer *r);
const char *vspg_renderer_kernel_name(VspgRenderer *r) {
    if (!r) return "";
    if (r->arith == VSPG_ARITH_EXACT) return kernel_name_exact(r);
    r->kernel_name_buf = std::string(r->arith == VSPG_ARITH_FAST_WEIGHTS ? "fastw::" : "fast::") + kernel_name_exact(r);  // (the inline namespace of vspg_fast.hip's symbols)
    return r->kernel_name_buf.c_str();
}
// Which instantiations a renderer's path kernels are launched from (vspg_arith.h).  The tolerance modes exist for the unguided
// rectangle-scene workgroup kernel and the unguided resampling pipeline over GridMedium; anything else is refused by name.
static bool arith_covered(const VspgRenderer *r);
int vspg_renderer_set_arithmetic(VspgRenderer *r, int mode) {
    if (!r) return fail(VSPG_EINVAL, "null renderer");
    if (mode != VSPG_ARITH_EXACT && mode != VSPG_ARITH_FAST_WEIGHTS && mode != VSPG_ARITH_FAST) return fail(VSPG_EINVAL, "unknown arithmetic mode");
    if (mode != VSPG_ARITH_EXACT && !arith_covered(r))
        return fail(VSPG_ESCOPE, "the tolerance-mode instantiations cover unguided renders of rectangle scenes over a homogeneous medium and unguided "
                                 "\"resampling\" renders over a uniformgrid medium (this renderer runs " + std::string(kernel_name_exact(r)) + ")");
    r->arith = mode;
    return 0;
}
int vspg_renderer_get_arithmetic(VspgRenderer *r) { return r ? r->arith : VSPG_EINVAL; }
static const char *kernel_name_exact(VspgRenderer *r) {
    const bool grid = r->scene.medium.type == VSPG_MEDIUM_GRID, nvdb = r->scene.medium.type == VSPG_MEDIUM_NANOVDB;
    const bool guided = wants_guiding(r->prm);
    if (uses_wf_pipeline(r) && r->prm.vspsamplingmethod != VSPG_VSP_RESAMPLING) {
        if (guided && r->training) return nvdb ? "k_wf_segment_vertex<NanoDenseMedium,guided,train>" : "k_wf_segment_vertex<GridMedium,guided,train>";
        if (guided) return nvdb ? "k_wf_segment_vertex<NanoDenseMedium,guided>" : "k_wf_segment_vertex<GridMedium,guided>";
        return nvdb ? (r->medium_grey ? "k_wf_segment_vertex<NanoDenseMediumGrey>" : "k_wf_segment_vertex<NanoDenseMedium>")
                    : (r->medium_grey ? "k_wf_segment_vertex<GridMediumGrey>" : "k_wf_segment_vertex<GridMedium>");
    }
    if (uses_wf_pipeline(r)) {
        // the pipeline is named by its walk kernel: k_wf_dist_walk (beside k_wf_shadow_walk), or k_wf_walk where one kernel runs both
        const char *n = guided && r->training ? (nvdb ? "k_wf_dist_walk<NanoDenseMedium,guided,train>" : "k_wf_dist_walk<GridMedium,guided,train>")
                        : guided ? (nvdb ? "k_wf_dist_walk<NanoDenseMedium,guided>" : "k_wf_dist_walk<GridMedium,guided>")
                        : nvdb ? (r->medium_grey ? "k_wf_dist_walk<NanoDenseMediumGrey>" : "k_wf_dist_walk<NanoDenseMedium>")
                               : (r->medium_grey ? "k_wf_dist_walk<GridMediumGrey>" : "k_wf_dist_walk<GridMedium>");
        if (!wf_merged_walks(r, guided)) return n;
        r->kernel_name_buf2 = std::string("k_wf_walk") + (n + sizeof("k_wf_dist_walk") - 1);
        return r->kernel_name_buf2.c_str();
    }
    const bool w3 = uses_wg3(r);
    if (uses_wg_guided(r)) {
        if (r->training) return guided_grey_simple(r) ? (w3 ? "k_render_wave_wg3<HomogeneousMediumT<2,true>,guided,train>" : "k_render_wave_wg2<HomogeneousMediumT<2,true>,guided,train>")
                                                      : (w3 ? "k_render_wave_wg3<HomogeneousMedium,guided,train>" : "k_render_wave_wg2<HomogeneousMedium,guided,train>");
        return guided_grey_simple(r) ? (w3 ? "k_render_wave_wg3<HomogeneousMediumT<2,true>,guided>" : "k_render_wave_wg2<HomogeneousMediumT<2,true>,guided>")
                                     : (w3 ? "k_render_wave_wg3<HomogeneousMedium,guided>" : "k_render_wave_wg2<HomogeneousMedium,guided>");
    }
    if (uses_wg_full(r)) return w3 ? "k_render_wave_wg3<HomogeneousMedium>" : "k_render_wave_wg2<HomogeneousMedium>";
    if (uses_wg_kernel(r) && uses_wg2(r)) {
        if (r->medium_grey && r->surfaces_grey && r->null_zero) return w3 ? "k_render_wave_wg3<HomogeneousMediumT<2,true>>" : "k_render_wave_wg2<HomogeneousMediumT<2,true>>";
        if (r->medium_grey && r->surfaces_grey) return w3 ? "k_render_wave_wg3<HomogeneousMediumT<2,false>>" : "k_render_wave_wg2<HomogeneousMediumT<2,false>>";
        if (r->medium_grey) return w3 ? "k_render_wave_wg3<HomogeneousMediumT<1,false>>" : "k_render_wave_wg2<HomogeneousMediumT<1,false>>";
        return w3 ? "k_render_wave_wg3<HomogeneousMediumT<0,false>>" : "k_render_wave_wg2<HomogeneousMediumT<0,false>>";
    }
    if (uses_wg_kernel(r)) {
        if (grid) return "k_render_wave_wg<GridMedium>";
        if (r->medium_grey && r->surfaces_grey && r->null_zero) return "k_render_wave_wg<HomogeneousMediumT<2,true>>";
        if (r->medium_grey && r->surfaces_grey) return "k_render_wave_wg<HomogeneousMediumT<2,false>>";
        if (r->medium_grey) return "k_render_wave_wg<HomogeneousMediumT<1,false>>";
        return "k_render_wave_wg<HomogeneousMediumT<0,false>>";
    }
    const bool train = guided && r->training;
    if (nvdb) return guided ? (train ? "k_render_wave<NanoDenseMedium,guided,train>" : "k_render_wave<NanoDenseMedium,guided>") : "k_render_wave<NanoDenseMedium>";
    if (grid) return guided ? (train ? "k_render_wave<GridMedium,guided,train>" : "k_render_wave<GridMedium,guided>")
                            : (r->medium_grey ? "k_render_wave<GridMediumGrey>" : "k_render_wave<GridMedium>");
    if (guided && guided_grey_simple(r)) return train ? "k_render_wave<HomogeneousMediumT<2,true>,guided,train>" : "k_render_wave<HomogeneousMediumT<2,true>,guided>";
    return guided ? (train ? "k_render_wave<HomogeneousMedium,guided,train>" : "k_render_wave<HomogeneousMedium,guided>") : "k_render_wave<HomogeneousMedium>";
}

int vspg_render_wave(VspgRenderer *r, int wave_start, int wave_end, void *stream) {
    const RoctxRange range("vspg_render_wave");
    if (!r) return fail(VSPG_EINVAL, "null renderer");
    if (wave_end < wave_start || wave_start < 0) return fail(VSPG_EINVAL, "bad wave range");
    if (wave_end == wave_start) return 0;
    HIPCHK(hipSetDevice(r->cfg.device));
    const int tilesX = (r->cfg.xres + 7) / 8, tilesY = (r->cfg.yres + 7) / 8;
    const long long items = (long long)tilesX * tilesY * 64;
    // persistent grid: exactly the resident wavefronts (never more blocks than tiles of work)
    long long blocks = (long long)r->num_cus * kBlocksPerCU;
    const long long max_blocks = (items + kBlock - 1) / kBlock;
    if (blocks > max_blocks) blocks = max_blocks;
    const long long n_waves = blocks * (kBlock / 64);
    // static slice per wavefront: 3/4 of the fair share, whole tiles
    // (VSPG_LANE_STATIC: the static share in 64ths, default 48 = 3/4)
    static const int lane_static64 = [] { const char *e = getenv("VSPG_LANE_STATIC"); const int v = e ? atoi(e) : 48; return v < 0 ? 0 : (v > 64 ? 64 : v); }();
    const unsigned static_per_wave = (unsigned)((items * lane_static64 / 64 / n_waves) / 64 * 64);
    const unsigned dyn_base = (unsigned)(static_per_wave * n_waves);
    // first sample index of this shard in the range, and how many it has
    const int sc = r->cfg.shard_count, si = r->cfg.shard_index;
    int first = wave_start;
    if (sc > 1) first += ((si - wave_start % sc) % sc + sc) % sc;
    if (first >= wave_end) return 0;  // nothing for this shard in the range
    const int n_samples = (wave_end - 1 - first) / sc + 1;
    const PcgJump jump = pcg_jump((unsigned long long)first * 65536ull);
    const bool grid = r->scene.medium.type == VSPG_MEDIUM_GRID;
    const bool nvdb = r->scene.medium.type == VSPG_MEDIUM_NANOVDB;
    const bool guided = wants_guiding(r->prm);
    if (guided && !r->field_set) return fail(VSPG_ESCOPE, "guiding enabled but the renderer holds no guiding field");
    // a one-sample launch of k_render_wave_wg2 resolves the samples its predecessor parked; every other launch adds to the film
    // itself, so the parked samples go in first
    const bool defer = !uses_wf_pipeline(r) && uses_wg2(r) && n_samples == 1 && wg2_defer_enabled();
    if (!defer) { const int rc = flush_parked_samples(r, (hipStream_t)stream); if (rc) return rc; }
    if (uses_wf_pipeline(r)) {  // one pass per sample index of this shard, in order
#ifdef VSPG_WF_DEBUG
        auto checksum = [&](const void *dptr, size_t bytes) -> unsigned long long {
            if (!dptr || !bytes) return 0ull;
            std::vector<unsigned char> h(bytes);
            (void)hipDeviceSynchronize();
            (void)hipMemcpy(h.data(), dptr, bytes, hipMemcpyDeviceToHost);
            unsigned long long x = 1469598103934665603ull;
            for (size_t i = 0; i < bytes; ++i) { x ^= h[i]; x *= 1099511628211ull; }
            return x;
        };
        const size_t nb = (size_t)r->hscene.bnx * r->hscene.bny * r->hscene.bnz;
        const void *bufs[6] = {r->tris, r->bvh, r->brick_index, r->octets, r->dscene, r->majorant};
        const size_t sizes[6] = {(size_t)r->hscene.n_tris * sizeof(DTri), (size_t)r->hscene.n_bvh_nodes * sizeof(DBvh4Node), nb * 4, r->n_bricks * 512 * 32,
                                 sizeof(DScene), (size_t)16 * 16 * 16 * 4};
        unsigned long long before[6];
        for (int k = 0; k < 6; ++k) before[k] = checksum(bufs[k], sizes[k]);
#endif
        for (int w = first; w < wave_end; w += sc > 1 ? sc : 1) {
            const hipStream_t hs = (hipStream_t)stream;
            const int rc = wf_render_pass(r, w, hs, nvdb, guided, guided && r->training, r->medium_grey);
            if (rc) return rc;
        }
#ifdef VSPG_WF_DEBUG
        {
            const char *names[6] = {"tris", "bvh", "brick_index", "octets", "dscene", "majorant"};
            for (int k = 0; k < 6; ++k)
                if (checksum(bufs[k], sizes[k]) != before[k]) fprintf(stderr, "VSPG_WF_DEBUG: the pipeline changed %s (%zu bytes at %p)\n", names[k], sizes[k], bufs[k]);
            fprintf(stderr, "VSPG_WF_DEBUG: pool %p..%p lists %p film %p isg %p tris %p bvh %p\n", (void *)r->wf_pool,
                    (void *)((char *)r->wf_pool + (size_t)kWfPoolFloats * r->wf_items * 4), (void *)r->wf_lists, (void *)r->film, (void *)r->isg_stats, (void *)r->tris, (void *)r->bvh);
        }
#endif
        return 0;
    }
    // a18: while the field trains, the guided kernels record path segments and emit radiance samples
    const bool train = guided && r->training;
    TrainArgs targs = {nullptr, nullptr, nullptr, nullptr, 0, 0};
    if (train) {
        if (n_samples > 1) {  // a training launch covers one sample per pixel (the record buffer is sized for that): split
            for (int w = first; w < wave_end; w += sc > 1 ? sc : 1) {
                const int rc = vspg_render_wave(r, w, w + 1, stream);
                if (rc) return rc;
            }
            return 0;
        }
        targs = TrainArgs{r->segbuf, r->seg_count, r->samples, r->train_counters, r->sample_capacity, (unsigned)items};
        HIPCHK(hipMemsetAsync(r->seg_count, 0, (size_t)items * sizeof(int), (hipStream_t)stream));
    }
    // exactly one path-kernel launch follows: it uses the counter the previous launch zeroed and zeroes the other one
    unsigned int *const work_head = r->work_head + (r->head_parity & 1u);
    r->head_parity ^= 1u;
#define VSPG_LAUNCH_RENDER(M, G)                                                                                          \
    do {                                                                                                                  \
        if (G && train)                                                                                                   \
            hipLaunchKernelGGL((k_render_wave<M, G, G>), dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream,    \
                               r->dscene, r->film, r->isg_stats, r->vsp, r->vsp_ready, wave_start, wave_end, first,        \
                               n_samples == 1 ? 1 : 0, jump, static_per_wave, dyn_base, work_head, r->counters, targs); \
        else                                                                                                              \
            hipLaunchKernelGGL((k_render_wave<M, G, false>), dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, \
                               r->dscene, r->film, r->isg_stats, r->vsp, r->vsp_ready, wave_start, wave_end, first,        \
                               n_samples == 1 ? 1 : 0, jump, static_per_wave, dyn_base, work_head, r->counters, targs); \
    } while (0)
    const bool use_wg = uses_wg_kernel(r);
    if (use_wg) {
        const unsigned tiles_magic = tilesX > 1 ? (unsigned)((0x100000000ull + (unsigned)tilesX - 1) / (unsigned)tilesX) : 0u;
        const bool gwg = uses_wg_guided(r);
        const int wwaves = gwg ? kWgWavesGuided : grid ? kWgWavesGrid : kWgWavesHomog, wblock = gwg ? kWgBlockGuided : grid ? kWgBlockGrid : kWgBlockHomog;
        long long wblocks = (long long)r->num_cus * (wwaves * 4 / (wblock / 64));
        const long long wmax = (items + kWgChunk - 1) / kWgChunk;
        if (wblocks > wmax) wblocks = wmax;
        const int single = n_samples == 1 ? 1 : 0;
        // Two schedulers (uses_wg2): k_render_wave_wg2 (tiles from a global head, parked samples, two barriers) by default,
        // k_render_wave_wg (film flush between the phases, three barriers) for grid media and under VSPG_WG_SCHED=1.
        const bool sched2 = uses_wg2(r);
        if (sched2) {
            for (int k = 0; k < 2; ++k)
                if (!r->wave_samples[k]) HIPCHK(hipMalloc(&r->wave_samples[k], r->npix * sizeof(float4)));
            float4 *const ws_out = r->wave_samples[r->ws_cur];
            const float4 *const ws_prev = defer && r->ws_parked ? r->wave_samples[r->ws_cur ^ 1] : nullptr;
            if (ws_prev) { const int rc = order_after_parking(r, (hipStream_t)stream); if (rc) return rc; }
            const long long n_tiles = (long long)tilesX * tilesY;
            if (wblocks > n_tiles) wblocks = n_tiles;
            // the share of the frame handed out from the global head, in 64ths (VSPG_WG2_TAIL; the rest is dealt to the workgroups
            // up front, interleaved).  Measured on the reference-default guided workload / the unguided one (ms per 1080p wave):
            // 0: 1.59 / 0.887, 8: 1.50 / 0.829, 16: 1.46 / 0.800, 32: 1.47 / 0.790, 64 (all of it): 1.454 / 0.775.
            const int tail64 = [] { const char *e = getenv("VSPG_WG2_TAIL"); const int v = e ? atoi(e) : 64; return v < 0 ? 0 : (v > 64 ? 64 : v); }();  // (per launch: a test varies it)
            const unsigned static_tiles = (unsigned)((n_tiles * (64 - tail64) / 64) / wblocks * wblocks);
#define VSPG_LAUNCH_WG2(M, G, NPOOL, BLK, WV)                                                                                         \
    hipLaunchKernelGGL((k_render_wave_wg2<M, G, NPOOL, BLK, WV>), dim3((unsigned)wblocks), dim3(BLK), 0, (hipStream_t)stream, r->dscene, \
                       r->film, r->isg_stats, r->vsp, r->vsp_ready, wave_end, first, single, jump, tiles_magic, static_tiles, work_head, ws_prev, ws_out, r->counters)
            if (uses_wg3(r)) {
                {  // the unguided rectangle-scene instantiations (the headline workload): also built in the tolerance modes
                    static_assert(kWgBlockHomog == VSPG_WG_BLOCK && kWgWavesHomog == VSPG_WG_WAVES, "wg3_launch_unguided's launch shape");
                    // (its tile cursors: a pair of sets of its own, alternating like the counter pair of the other kernels -- which this launch leaves alone)
                    unsigned int *const head8 = r->work_head8 + (r->head8_parity & 1u) * (unsigned)(kWg3HeadSetBytes / 4);
                    r->head8_parity ^= 1u;
                    r->head_parity ^= 1u;  // (undo the toggle above: no launch used that pair)
                    const Wg3Launch L3{r->dscene, r->film, r->isg_stats, r->vsp, r->vsp_ready, wave_end, first, single, jump, tiles_magic, head8, ws_prev, ws_out,
                                       r->counters, (unsigned)wblocks, (hipStream_t)stream, r->medium_grey ? (r->surfaces_grey ? 2 : 1) : 0,
                                       r->medium_grey && r->surfaces_grey && r->null_zero ? 1 : 0};
                    const int lrc = r->arith == VSPG_ARITH_FAST_WEIGHTS ? vspg_arith1_wg3(&L3) : r->arith == VSPG_ARITH_FAST ? vspg_arith2_wg3(&L3) : wg3_launch_unguided(L3);
                    if (lrc != 0) return fail(VSPG_EHIP, std::string("k_render_wave_wg3: ") + hipGetErrorName((hipError_t)lrc));
                }
            } else
            if (gwg && train && guided_grey_simple(r))
                hipLaunchKernelGGL((k_render_wave_wg2<HomogeneousMediumGreySceneNullZero, true, kWg2PoolTrainT<2>, kWgBlockGuided, kWgWavesGuided, true>), dim3((unsigned)wblocks),
                                   dim3(kWgBlockGuided), 0, (hipStream_t)stream, r->dscene, r->film, r->isg_stats, r->vsp, r->vsp_ready, wave_end, first, single, jump,
                                   tiles_magic, static_tiles, work_head, ws_prev, ws_out, r->counters, targs);
            else if (gwg && train)
                hipLaunchKernelGGL((k_render_wave_wg2<HomogeneousMediumSimple, true, kWg2PoolTrainT<0>, kWgBlockGuided, kWgWavesGuided, true>), dim3((unsigned)wblocks),
                                   dim3(kWgBlockGuided), 0, (hipStream_t)stream, r->dscene, r->film, r->isg_stats, r->vsp, r->vsp_ready, wave_end, first, single, jump,
                                   tiles_magic, static_tiles, work_head, ws_prev, ws_out, r->counters, targs);
            else if (gwg && guided_grey_simple(r)) VSPG_LAUNCH_WG2(HomogeneousMediumGreySceneNullZero, true, kWg2PoolGuidedT<2>, kWgBlockGuided, kWgWavesGuided);
            else if (gwg) VSPG_LAUNCH_WG2(HomogeneousMediumSimple, true, kWg2PoolGuidedT<0>, kWgBlockGuided, kWgWavesGuided);
            else if (uses_wg_full(r)) VSPG_LAUNCH_WG2(HomogeneousMedium, false, kWg2PoolFull, kWgBlockHomog, kWgWavesHomog);
            else if (r->medium_grey && r->surfaces_grey && r->null_zero) VSPG_LAUNCH_WG2(HomogeneousMediumGreySceneNullZero, false, kWg2PoolHomogT<2>, kWgBlockHomog, kWgWavesHomog);
            else if (r->medium_grey && r->surfaces_grey) VSPG_LAUNCH_WG2(HomogeneousMediumGreyScene, false, kWg2PoolHomogT<2>, kWgBlockHomog, kWgWavesHomog);
            else if (r->medium_grey) VSPG_LAUNCH_WG2(HomogeneousMediumGrey, false, kWg2PoolHomogT<1>, kWgBlockHomog, kWgWavesHomog);
            else VSPG_LAUNCH_WG2(HomogeneousMediumSimple, false, kWg2PoolHomogT<0>, kWgBlockHomog, kWgWavesHomog);
#undef VSPG_LAUNCH_WG2
            HIPCHK(hipGetLastError());
            if (single) {  // this launch's samples are parked in ws_out (its predecessor's, if any were, have just been resolved)
                r->ws_cur ^= 1;
                r->ws_parked = true;
                r->ws_stream = (hipStream_t)stream;
                if (!r->ws_event) HIPCHK(hipEventCreateWithFlags(&r->ws_event, hipEventDisableTiming));
                HIPCHK(hipEventRecord(r->ws_event, (hipStream_t)stream));
                if (!defer) { const int rc = flush_parked_samples(r, (hipStream_t)stream); if (rc) return rc; }
            }
        } else {
#define VSPG_LAUNCH_WG(M, NPOOL, BLK, WV)                                                                                            \
    hipLaunchKernelGGL((k_render_wave_wg<M, false, NPOOL, BLK, WV>), dim3((unsigned)wblocks), dim3(BLK), 0, (hipStream_t)stream, r->dscene, \
                       r->film, r->isg_stats, r->vsp, r->vsp_ready, wave_end, first, single, jump, tiles_magic, work_head, r->counters)
            if (grid) VSPG_LAUNCH_WG(GridMedium, kWgPoolGrid, kWgBlockGrid, kWgWavesGrid);
            else if (r->medium_grey && r->surfaces_grey && r->null_zero) VSPG_LAUNCH_WG(HomogeneousMediumGreySceneNullZero, kWgPoolHomogT<2>, kWgBlockHomog, kWgWavesHomog);  // ... and the null-collision coefficient is exactly 0
            else if (r->medium_grey && r->surfaces_grey) VSPG_LAUNCH_WG(HomogeneousMediumGreyScene, kWgPoolHomogT<2>, kWgBlockHomog, kWgWavesHomog);  // ... and every Kd bitwise grey: beta is grey by construction too
            else if (r->medium_grey) VSPG_LAUNCH_WG(HomogeneousMediumGrey, kWgPoolHomogT<1>, kWgBlockHomog, kWgWavesHomog);  // sigma_a, sigma_s, Le bitwise grey: the broadcast-spectrum instantiation
            else VSPG_LAUNCH_WG(HomogeneousMediumSimple, kWgPoolHomogT<0>, kWgBlockHomog, kWgWavesHomog);
#undef VSPG_LAUNCH_WG
        }
    } else if (nvdb && guided) VSPG_LAUNCH_RENDER(NanoDenseMedium, true);
    else if (nvdb) VSPG_LAUNCH_RENDER(NanoDenseMedium, false);
    else if (grid && guided) VSPG_LAUNCH_RENDER(GridMedium, true);
    else if (grid && r->medium_grey) VSPG_LAUNCH_RENDER(GridMediumGrey, false);
    else if (grid) VSPG_LAUNCH_RENDER(GridMedium, false);
    else if (guided && guided_grey_simple(r)) VSPG_LAUNCH_RENDER(HomogeneousMediumGreySceneNullZero, true);
    else if (guided) VSPG_LAUNCH_RENDER(HomogeneousMedium, true);
    else VSPG_LAUNCH_RENDER(HomogeneousMedium, false);
#undef VSPG_LAUNCH_RENDER
    HIPCHK(hipGetLastError());
    if (train) {
        hipLaunchKernelGGL(k_propagate, dim3((unsigned)((items + kBlock - 1) / kBlock)), dim3(kBlock), 0, (hipStream_t)stream, targs,
                           train_rec_capacity(r->prm.maxdepth));
        HIPCHK(hipGetLastError());
    }
    return 0;
}

// Field::Update (:239) on device; one host read of the sample count per training wave decides whether the
// update runs (more than 128 valid samples, :238) and sizes the grids.
static int train_update(VspgRenderer *r, hipStream_t s) {
    unsigned long long cnt[2] = {0, 0};
    HIPCHK(hipMemcpyAsync(cnt, r->train_counters, sizeof cnt, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const unsigned long long n = cnt[0] < r->sample_capacity ? cnt[0] : r->sample_capacity;
    // sharded training: `xchg` sums a device buffer over the ranks (no-op without a hook); the decision to update is taken
    // on the SUMMED sample count, so every rank takes it alike and the hook runs the same number of times everywhere
    const auto xchg = [&](float *p, size_t nf) -> int {
        if (!r->exchange) return 0;
        const int rc = r->exchange(p, nf, (void *)s, r->exchange_user);
        return rc ? fail(rc, "the exchange hook of the sharded guiding-field update failed") : 0;
    };
    double n_all = (double)n;
    {
        float nf = (float)n;  // train_sumw[1]: the sample count the E-step divides the weight sum by
        HIPCHK(hipMemcpyAsync(r->train_sumw + 1, &nf, sizeof nf, hipMemcpyHostToDevice, s));
        if (r->exchange) {
            if (int rc = xchg(r->train_sumw + 1, 1)) return rc;
            HIPCHK(hipMemcpyAsync(&nf, r->train_sumw + 1, sizeof nf, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            n_all = (double)nf;
        }
    }
    if (n_all > (double)kTrainMinUpdateSamples) {
        const size_t acc_floats = (size_t)kTrainKeys * kStatFloats, acc_bytes = acc_floats * sizeof(float);
        unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
        if (grid < 1u) grid = 1u;  // (a rank whose own samples ran dry still takes part in the sums)
        if (grid > (unsigned)r->num_cus * 16u) grid = (unsigned)r->num_cus * 16u;
        const unsigned rgrid = (kTrainKeys + kBlock - 1) / kBlock;
        // counting sort: few, long chunks -- every workgroup flushes one atomic per bin its chunk touched
        unsigned sgrid = (unsigned)((n + kSortBlock - 1) / kSortBlock);
        if (sgrid < 1u) sgrid = 1u;
        if (sgrid > (unsigned)r->num_cus) sgrid = (unsigned)r->num_cus;
        // accumulation passes: every wavefront flushes its running sums at least once, so no more wavefronts than fill the chip
        const unsigned agrid = grid < (unsigned)r->num_cus * 4u ? grid : (unsigned)r->num_cus * 4u;
        const TrainFields tf{{r->rstats[0], r->rstats[1]}, {r->fregions[0], r->fregions[1]}, {r->fnodes[0], r->fnodes[1]}};
        const size_t hist_bytes = (size_t)kTrainKeys * sizeof(unsigned int);
        // key_of, order, n_sorted for both fields as the trees stand now; the second sort (redo) only if a tree changed
        auto sort_by_key = [&](float *sumw, const int *redo) -> int {
            HIPCHK(hipMemsetAsync(r->train_hist, 0, hist_bytes, s));
            hipLaunchKernelGGL(k_train_lookup, dim3(sgrid), dim3(kSortBlock), 0, s, r->dscene, r->samples, n, r->train_reg, r->train_hist, sumw, redo);
            hipLaunchKernelGGL(k_train_scan, dim3(1), dim3(kBlock), 0, s, r->train_hist, r->train_cursor, r->train_nsorted, redo);
            hipLaunchKernelGGL(k_train_scatter, dim3(sgrid), dim3(kSortBlock), 0, s, r->train_reg, n, r->train_cursor, r->train_order, redo);
            return 0;
        };
        HIPCHK(hipMemsetAsync(r->train_sumw, 0, sizeof(float), s));
        HIPCHK(hipMemsetAsync(r->train_nsplit, 0, 2 * sizeof(int), s));
        hipLaunchKernelGGL(k_train_decay, dim3(rgrid * 4), dim3(kBlock), 0, s, r->dscene, tf);
        if (int rc = sort_by_key(r->train_sumw, nullptr)) return rc;
        if (int rc = xchg(r->train_sumw, 1)) return rc;
        HIPCHK(hipMemsetAsync(r->train_acc, 0, acc_bytes, s));
        hipLaunchKernelGGL((k_train_pos<true>), dim3(agrid), dim3(kBlock), 0, s, r->samples, r->train_reg, r->train_order, r->train_nsorted,
                           r->train_acc, (const int *)nullptr);
        if (int rc = xchg(r->train_acc, acc_floats)) return rc;
        hipLaunchKernelGGL(k_train_split, dim3(2), dim3(kSplitBlock), 0, s, r->dscene, tf, r->train_acc, r->train_nsplit);
        if (int rc = sort_by_key(nullptr, r->train_nsplit)) return rc;  // the split changed the leaves
        HIPCHK(hipMemsetAsync(r->train_acc, 0, acc_bytes, s));
        hipLaunchKernelGGL((k_train_pos<false>), dim3(agrid), dim3(kBlock), 0, s, r->samples, r->train_reg, r->train_order, r->train_nsorted,
                           r->train_acc, (const int *)(r->train_nsplit + 1));
        if (int rc = xchg(r->train_acc, acc_floats)) return rc;
        hipLaunchKernelGGL(k_train_init_regions, dim3(rgrid), dim3(kBlock), 0, s, r->dscene, tf, r->train_acc);
        HIPCHK(hipMemsetAsync(r->train_acc, 0, acc_bytes, s));
        hipLaunchKernelGGL(k_train_estep, dim3(agrid), dim3(kBlock), 0, s, r->dscene, r->samples, r->train_reg, r->train_order, r->train_nsorted,
                           r->train_sumw, r->train_acc);
        if (int rc = xchg(r->train_acc, acc_floats)) return rc;
        hipLaunchKernelGGL(k_train_mstep, dim3(rgrid), dim3(kBlock), 0, s, r->dscene, tf, r->train_acc);
        for (int f = 0; f < 2; ++f)
            hipLaunchKernelGGL(k_field_aux, dim3((kTrainCapRegions + kBlock - 1) / kBlock * 2), dim3(kBlock), 0, s, r->dscene, f, r->fregions[f],
                               r->faux[f], r->flobes[f]);
        HIPCHK(hipGetLastError());
        r->field_iteration++;
        if (r->field_iteration >= r->prm.guide_num_training_waves) r->training = false;
    }
    return 0;
}

// The image-space buffer update falls on the step that takes the wave counter to (or past) 2^bufferWave; with one
// wave per step this is the reference's `waveCounter == pow(2, bufferWave)` (:251).
static bool isg_update_due(const VspgRenderer *r, int n_waves) {
    return (double)(r->wave_counter + n_waves) >= std::pow(2.0, (double)r->buffer_wave);
}
int vspg_isg_update_due(VspgRenderer *r, int n_waves) {
    if (!r || n_waves < 1) return 0;
    const bool do_vsp = r->prm.vspguiding && r->prm.vspprimaryguiding && !r->vsp_loaded;
    return (do_vsp || r->prm.rrguiding) && isg_update_due(r, n_waves) ? 1 : 0;
}

int vspg_post_process_step(VspgRenderer *r, int n_waves, const float *isg_stats_sum, void *stream) {
    // PostProcessWave (guidedvolpathvspgintegrator.cpp:230-260) after a step that covered n_waves sample indices
    const RoctxRange range("vspg_post_process_step");
    if (!r) return fail(VSPG_EINVAL, "null renderer");
    if (n_waves < 1) return fail(VSPG_EINVAL, "n_waves must be >= 1");
    const bool due = isg_update_due(r, n_waves);
    r->wave_counter += n_waves;
    if (r->train_counters) {
        HIPCHK(hipSetDevice(r->cfg.device));
        if (r->training) {
            int rc = train_update(r, (hipStream_t)stream);
            if (rc) return rc;
        }
        HIPCHK(hipMemsetAsync(r->train_counters, 0, 4 * sizeof(unsigned long long), (hipStream_t)stream));  // Clear() (:248)
    }
    if (due) {
        const bool do_vsp = r->prm.vspguiding && r->prm.vspprimaryguiding && !r->vsp_loaded;  // calculateImageSpaceGuidingBuffer (:251)
        const bool do_contrib = r->prm.rrguiding != 0;  // cfg.EnableContributionEstimate(guideRR) (:164-168)
        if (do_vsp || do_contrib) {
            HIPCHK(hipSetDevice(r->cfg.device));
            { const int rc = flush_parked_samples(r, (hipStream_t)stream); if (rc) return rc; }  // the statistics of every wave so far
            hipLaunchKernelGGL(k_isg_update, dim3((unsigned)((r->cfg.xres + kIsgTile - 1) / kIsgTile), (unsigned)((r->cfg.yres + kIsgTile - 1) / kIsgTile)),
                               dim3(kBlock), 0, (hipStream_t)stream, r->cfg.xres, r->cfg.yres,
                               r->prm.vspcriterion, isg_stats_sum ? isg_stats_sum : r->isg_stats, do_vsp ? r->vsp : nullptr,
                               do_contrib ? r->contrib : nullptr);
            HIPCHK(hipGetLastError());
            if (do_vsp) r->vsp_ready = 1;
            if (do_contrib && !r->hscene.contrib_ready) {
                r->hscene.contrib_ready = 1;
                HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(r->dscene) + offsetof(DScene, contrib_ready), &r->hscene.contrib_ready,
                                      sizeof(int32_t), hipMemcpyHostToDevice, (hipStream_t)stream));
            }
        }
        while (std::pow(2.0, (double)r->buffer_wave) <= (double)r->wave_counter) r->buffer_wave++;
    }
    return 0;
}
int vspg_post_process_wave(VspgRenderer *r, void *stream) { return vspg_post_process_step(r, 1, nullptr, stream); }
int vspg_renderer_set_exchange(VspgRenderer *r, VspgExchangeFn fn, void *user) {
    if (!r) return fail(VSPG_EINVAL, "null renderer");
    r->exchange = fn;
    r->exchange_user = user;
    return 0;
}

int vspg_flush(VspgRenderer *r, void *stream) {
    if (!r) return fail(VSPG_EINVAL, "null renderer");
    if (!r->ws_parked) return 0;
    HIPCHK(hipSetDevice(r->cfg.device));
    return flush_parked_samples(r, (hipStream_t)stream);
}
int vspg_film_device_ptr(VspgRenderer *r, float **dev_ptr, size_t *n_floats) {
    if (!r || !dev_ptr || !n_floats) return fail(VSPG_EINVAL, "null argument");
    if (r->ws_parked) {  // the caller reads the film on a stream of its own: the parked samples go in, and are in, before it gets the pointer
        HIPCHK(hipSetDevice(r->cfg.device));
        const hipStream_t s = r->ws_stream;
        if (const int rc = flush_parked_samples(r, s)) return rc;
        HIPCHK(hipStreamSynchronize(s));
    }
    *dev_ptr = reinterpret_cast<float *>(r->film);
    *n_floats = r->npix * 4;
    return 0;
}
int vspg_film_read(VspgRenderer *r, float *host, void *stream) {
    if (!r || !host) return fail(VSPG_EINVAL, "null argument");
    HIPCHK(hipSetDevice(r->cfg.device));
    if (const int rc = flush_parked_samples(r, (hipStream_t)stream)) return rc;
    HIPCHK(hipMemcpyAsync(host, r->film, r->npix * sizeof(float4), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int vspg_film_clear(VspgRenderer *r, void *stream) {
    if (!r) return fail(VSPG_EINVAL, "null renderer");
    HIPCHK(hipSetDevice(r->cfg.device));
    if (const int rc = flush_parked_samples(r, (hipStream_t)stream)) return rc;  // (their statistics stay; the film is cleared after)
    HIPCHK(hipMemsetAsync(r->film, 0, r->npix * sizeof(float4), (hipStream_t)stream));
    return 0;
}
int vspg_vsp_buffer_device_ptr(VspgRenderer *r, float **dev_ptr, size_t *n_floats) {
    if (!r || !dev_ptr || !n_floats) return fail(VSPG_EINVAL, "null argument");
    *dev_ptr = r->vsp;
    *n_floats = r->npix;
    return 0;
}
int vspg_vsp_buffer_read(VspgRenderer *r, float *host, int *is_ready, void *stream) {
    if (!r || !host) return fail(VSPG_EINVAL, "null argument");
    HIPCHK(hipSetDevice(r->cfg.device));
    HIPCHK(hipMemcpyAsync(host, r->vsp, r->npix * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    if (is_ready) *is_ready = r->vsp_ready;
    return 0;
}
// TrBuffer (cpu/trbuffer.h): read-back = what Store() writes, set = TrBuffer(fileName) / Load()
int vspg_renderer_get_tr_buffer(VspgRenderer *r, float *host_rgb, int32_t *host_spp, void *stream) {
    if (!r || !host_rgb) return fail(VSPG_EINVAL, "null argument");
    if (!r->tr_rgb) return fail(VSPG_EINVAL, "the renderer keeps no transmittance buffer (storeTrBuffer / NDS+ not requested)");
    HIPCHK(hipSetDevice(r->cfg.device));
    HIPCHK(hipMemcpyAsync(host_rgb, r->tr_rgb, r->npix * 3 * sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (host_spp) {
        if (r->tr_spp) HIPCHK(hipMemcpyAsync(host_spp, r->tr_spp, r->npix * sizeof(int32_t), hipMemcpyDeviceToHost, (hipStream_t)stream));
        else memset(host_spp, 0, r->npix * sizeof(int32_t));
    }
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    return 0;
}
int vspg_renderer_set_tr_buffer(VspgRenderer *r, const float *host_rgb, void *stream) {
    if (!r || !host_rgb) return fail(VSPG_EINVAL, "null argument");
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    if (!r->tr_rgb) HIPCHK(hipMalloc(&r->tr_rgb, r->npix * 3 * sizeof(float)));
    HIPCHK(hipMemcpyAsync(r->tr_rgb, host_rgb, r->npix * 3 * sizeof(float), hipMemcpyHostToDevice, s));
    // trBufferLoad = true, calculateTrBuffer = false (:182-184).  Only these members are rewritten: the field
    // counters of the device-side scene belong to the training kernels.
    r->hscene.tr_rgb = r->tr_rgb;
    r->hscene.tr_calc = 0;
    r->hscene.tr_load = 1;
    HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(r->dscene) + offsetof(DScene, tr_rgb), &r->hscene.tr_rgb,
                          sizeof(DScene) - offsetof(DScene, tr_rgb), hipMemcpyHostToDevice, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}
int vspg_vsp_buffer_load(VspgRenderer *r, const float *host, void *stream) {
    if (!r || !host) return fail(VSPG_EINVAL, "null argument");
    HIPCHK(hipSetDevice(r->cfg.device));
    HIPCHK(hipMemcpyAsync(r->vsp, host, r->npix * sizeof(float), hipMemcpyHostToDevice, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    r->vsp_ready = 1;
    r->vsp_loaded = true;
    return 0;
}
int vspg_isg_stats_device_ptr(VspgRenderer *r, float **dev_ptr, size_t *n_floats) {
    if (!r || !dev_ptr || !n_floats) return fail(VSPG_EINVAL, "null argument");
    if (r->ws_parked) {
        HIPCHK(hipSetDevice(r->cfg.device));
        const hipStream_t s = r->ws_stream;
        if (const int rc = flush_parked_samples(r, s)) return rc;
        HIPCHK(hipStreamSynchronize(s));
    }
    *dev_ptr = r->isg_stats;
    *n_floats = r->npix * VSPG_ISG_STATS;
    return 0;
}
int vspg_get_counters(VspgRenderer *r, VspgCounters *out, void *stream) {
    if (!r || !out) return fail(VSPG_EINVAL, "null argument");
    HIPCHK(hipSetDevice(r->cfg.device));
    unsigned long long h[kNumCounters];
    HIPCHK(hipMemcpyAsync(h, r->counters, sizeof h, hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIPCHK(hipStreamSynchronize((hipStream_t)stream));
    out->paths = h[0]; out->segments = h[1]; out->volume_scatters = h[2];
    out->surface_hits = h[3]; out->density_queries = h[4]; out->shadow_rays = h[5];
    out->shadow_density_queries = h[6];
    return 0;
}
int vspg_reset_counters(VspgRenderer *r, void *stream) {
    if (!r) return fail(VSPG_EINVAL, "null renderer");
    HIPCHK(hipSetDevice(r->cfg.device));
    HIPCHK(hipMemsetAsync(r->counters, 0, kNumCounters * sizeof(unsigned long long), (hipStream_t)stream));
    return 0;
}

// scoped device scratch for the batch entry points
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
};

int vspg_trace_paths(VspgRenderer *r, int n, const int32_t *pixel_xy, const int32_t *sample_index, float *out_L,
                     int32_t *out_segments, void *stream) {
    if (!r || n < 0 || (n > 0 && (!pixel_xy || !sample_index || !out_L))) return fail(VSPG_EINVAL, "bad arguments");
    if (n == 0) return 0;
    for (int i = 0; i < n; ++i)
        if (pixel_xy[2 * i] < 0 || pixel_xy[2 * i] >= r->cfg.xres || pixel_xy[2 * i + 1] < 0 || pixel_xy[2 * i + 1] >= r->cfg.yres)
            return fail(VSPG_EINVAL, "pixel outside the film");
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf dp, ds, dl, dg;
    HIPCHK(hipMalloc(&dp.p, (size_t)n * 2 * sizeof(int32_t)));
    HIPCHK(hipMalloc(&ds.p, (size_t)n * sizeof(int32_t)));
    HIPCHK(hipMalloc(&dl.p, (size_t)n * 3 * sizeof(float)));
    HIPCHK(hipMalloc(&dg.p, (size_t)n * sizeof(int32_t)));
    HIPCHK(hipMemcpyAsync(dp.p, pixel_xy, (size_t)n * 2 * sizeof(int32_t), hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(ds.p, sample_index, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, s));
    const bool grid = r->scene.medium.type == VSPG_MEDIUM_GRID;
    const bool guided = wants_guiding(r->prm);
    if (guided && !r->field_set) return fail(VSPG_ESCOPE, "guiding enabled but no guiding field uploaded");
#define VSPG_LAUNCH_TRACE(M, G)                                                                                       \
    hipLaunchKernelGGL((k_trace_paths<M, G>), dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, r->dscene, r->vsp, \
                       r->vsp_ready | VSP_NO_FEED, n, (const int32_t *)dp.p, (const int32_t *)ds.p, (float *)dl.p, (int32_t *)dg.p)
    const bool nvdb = r->scene.medium.type == VSPG_MEDIUM_NANOVDB;
    if (r->arith != VSPG_ARITH_EXACT) {  // the replay in the renderer's arithmetic (set_arithmetic accepted unguided homogeneous / uniformgrid only)
        const TraceLaunch T{r->dscene, r->vsp, r->vsp_ready | VSP_NO_FEED, n, (const int32_t *)dp.p, (const int32_t *)ds.p, (float *)dl.p, (int32_t *)dg.p, s, grid ? 1 : 0};
        const int trc = r->arith == VSPG_ARITH_FAST_WEIGHTS ? vspg_arith1_trace(&T) : vspg_arith2_trace(&T);
        if (trc != 0) return fail(VSPG_EHIP, std::string("k_trace_paths: ") + hipGetErrorName((hipError_t)trc));
    } else
    if (nvdb && guided) VSPG_LAUNCH_TRACE(NanoDenseMedium, true);
    else if (nvdb) VSPG_LAUNCH_TRACE(NanoDenseMedium, false);
    else if (grid && guided) VSPG_LAUNCH_TRACE(GridMedium, true);
    else if (grid) VSPG_LAUNCH_TRACE(GridMedium, false);
    else if (guided) VSPG_LAUNCH_TRACE(HomogeneousMedium, true);
    else VSPG_LAUNCH_TRACE(HomogeneousMedium, false);
#undef VSPG_LAUNCH_TRACE
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_L, dl.p, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, s));
    if (out_segments) HIPCHK(hipMemcpyAsync(out_segments, dg.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int vspg_ray_batch(VspgRenderer *r, int n, const VspgRayQuery *q, VspgRayResult *out, void *stream) {
    if (!r || n < 0 || (n > 0 && (!q || !out))) return fail(VSPG_EINVAL, "bad arguments");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf dq, dout;
    HIPCHK(hipMalloc(&dq.p, (size_t)n * sizeof(VspgRayQuery)));
    HIPCHK(hipMalloc(&dout.p, (size_t)n * sizeof(VspgRayResult)));
    HIPCHK(hipMemcpyAsync(dq.p, q, (size_t)n * sizeof(VspgRayQuery), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_ray_batch, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, r->dscene, n, (const VspgRayQuery *)dq.p, (VspgRayResult *)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, (size_t)n * sizeof(VspgRayResult), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int vspg_sample_tmaj_batch(VspgRenderer *r, int variant, int n, const VspgTmajQuery *q, VspgTmajResult *out, void *stream) {
    if (!r || n < 0 || (n > 0 && (!q || !out))) return fail(VSPG_EINVAL, "bad arguments");
    if (variant < VSPG_TMAJ_PLAIN || variant > VSPG_TMAJ_RESAMPLING) return fail(VSPG_EINVAL, "unknown variant");
    if (n == 0) return 0;
    for (int i = 0; i < n; ++i)
        if (q[i].channel < 0 || q[i].channel > 2) return fail(VSPG_EINVAL, "channel must be 0..2");
    if (r->scene.medium.type == VSPG_MEDIUM_NONE) return fail(VSPG_EINVAL, "renderer has no medium");
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf dq, dr;
    HIPCHK(hipMalloc(&dq.p, (size_t)n * sizeof(VspgTmajQuery)));
    HIPCHK(hipMalloc(&dr.p, (size_t)n * sizeof(VspgTmajResult)));
    HIPCHK(hipMemcpyAsync(dq.p, q, (size_t)n * sizeof(VspgTmajQuery), hipMemcpyHostToDevice, s));
    if (r->scene.medium.type == VSPG_MEDIUM_NANOVDB)
        hipLaunchKernelGGL(k_tmaj_batch<NanoDenseMedium>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, r->dscene, variant, n,
                           (const VspgTmajQuery *)dq.p, (VspgTmajResult *)dr.p);
    else if (r->scene.medium.type == VSPG_MEDIUM_GRID)
        hipLaunchKernelGGL(k_tmaj_batch<GridMedium>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, r->dscene, variant, n,
                           (const VspgTmajQuery *)dq.p, (VspgTmajResult *)dr.p);
    else
        hipLaunchKernelGGL(k_tmaj_batch<HomogeneousMedium>, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, r->dscene, variant, n,
                           (const VspgTmajQuery *)dq.p, (VspgTmajResult *)dr.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dr.p, (size_t)n * sizeof(VspgTmajResult), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int vspg_primitives_batch(VspgRenderer *r, int n, const float *f, const float *g, uint64_t *hash, uint32_t *rng_u32,
                          float *fastexp, void *stream) {
    if (!r || n < 0 || (n > 0 && (!f || !g || !hash || !rng_u32 || !fastexp))) return fail(VSPG_EINVAL, "bad arguments");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf df, dg, dh, du, de;
    HIPCHK(hipMalloc(&df.p, (size_t)n * 4));
    HIPCHK(hipMalloc(&dg.p, (size_t)n * 4));
    HIPCHK(hipMalloc(&dh.p, (size_t)n * 8));
    HIPCHK(hipMalloc(&du.p, (size_t)n * 4));
    HIPCHK(hipMalloc(&de.p, (size_t)n * 4));
    HIPCHK(hipMemcpyAsync(df.p, f, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dg.p, g, (size_t)n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_primitives, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n, (const float *)df.p, (const float *)dg.p,
                       (uint64_t *)dh.p, (uint32_t *)du.p, (float *)de.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(hash, dh.p, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(rng_u32, du.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(fastexp, de.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

static int upload_field(VspgRenderer *r, int f, const VspgField *src, hipStream_t s) {
    if (r->fnodes[f]) { (void)hipFree(r->fnodes[f]); r->fnodes[f] = nullptr; }
    if (r->fregions[f]) { (void)hipFree(r->fregions[f]); r->fregions[f] = nullptr; }
    if (r->faux[f]) { (void)hipFree(r->faux[f]); r->faux[f] = nullptr; }
    if (r->flobes[f]) { (void)hipFree(r->flobes[f]); r->flobes[f] = nullptr; }
    r->hscene.field[f] = DField{0, 0, nullptr, nullptr, nullptr, nullptr};
    if (!src || src->n_nodes <= 0 || src->n_regions <= 0) return 0;
    if (!src->nodes || !src->regions) return fail(VSPG_EINVAL, "guiding field without node / region arrays");
    for (int i = 0; i < src->n_nodes; ++i) {  // structural check: children after their parent, leaves in range
        uint32_t axis = src->nodes[i].packed & 3u, idx = src->nodes[i].packed >> 2;
        if (axis == 3u ? (int)idx >= src->n_regions : ((int)idx + 1 >= src->n_nodes || (int)idx <= i))
            return fail(VSPG_EINVAL, "malformed guiding-field kd-tree");
    }
    for (int i = 0; i < src->n_regions; ++i)
        if (src->regions[i].n_lobes < 0 || src->regions[i].n_lobes > VSPG_FIELD_LOBES)
            return fail(VSPG_EINVAL, "guiding-field region with an invalid lobe count");
    HIPCHK(hipMalloc(&r->fnodes[f], sizeof(VspgKdNode) * src->n_nodes));
    HIPCHK(hipMalloc(&r->fregions[f], sizeof(VspgFieldRegion) * src->n_regions));
    HIPCHK(hipMemcpyAsync(r->fnodes[f], src->nodes, sizeof(VspgKdNode) * src->n_nodes, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(r->fregions[f], src->regions, sizeof(VspgFieldRegion) * src->n_regions, hipMemcpyHostToDevice, s));
    HIPCHK(hipMalloc(&r->faux[f], sizeof(float) * 2 * GK * src->n_regions));
    HIPCHK(hipMalloc(&r->flobes[f], sizeof(float4) * kRegionLobeQuads * src->n_regions));
    r->hscene.field[f] = DField{src->n_nodes, src->n_regions, r->fnodes[f], r->fregions[f], r->faux[f], r->flobes[f]};
    return 0;
}

int vspg_renderer_set_guiding_field(VspgRenderer *r, const VspgField *surface_field, const VspgField *volume_field,
                                    void *stream) {
    if (!r) return fail(VSPG_EINVAL, "null renderer");
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    HIPCHK(hipStreamSynchronize(s));  // no launch may still read the old field
    r->training = false;                // a loaded cache is not trained further (:117-122)
    int rc = upload_field(r, 0, surface_field, s);
    if (rc) return rc;
    rc = upload_field(r, 1, volume_field, s);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(r->dscene, &r->hscene, sizeof(DScene), hipMemcpyHostToDevice, s));
    for (int f = 0; f < 2; ++f)
        if (r->faux[f])
            hipLaunchKernelGGL(k_field_aux, dim3((r->hscene.field[f].n_regions * GK + kBlock - 1) / kBlock), dim3(kBlock), 0, s, r->dscene,
                               f, r->fregions[f], r->faux[f], r->flobes[f]);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    r->field_set = true;
    return 0;
}

int vspg_guiding_query_batch(VspgRenderer *r, int is_volume, float g, int n, const float *p, const float *n_or_wo,
                             const float *wi, const float *u, int32_t *out_ok, float *out_pdf, float *out_incoming_pdf,
                             float *out_vsp, float *out_ws, float *out_pdf_s, void *stream) {
    if (!r || n < 0 || (n > 0 && (!p || !n_or_wo || !wi || !u || !out_ok || !out_pdf || !out_incoming_pdf || !out_vsp || !out_ws || !out_pdf_s)))
        return fail(VSPG_EINVAL, "bad arguments");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf dp, da, dw, du, dok, dpdf, dinc, dvsp, dws, dps;
    const size_t n3 = (size_t)n * 3 * 4, n2 = (size_t)n * 2 * 4, n1 = (size_t)n * 4;
    HIPCHK(hipMalloc(&dp.p, n3)); HIPCHK(hipMalloc(&da.p, n3)); HIPCHK(hipMalloc(&dw.p, n3)); HIPCHK(hipMalloc(&du.p, n2));
    HIPCHK(hipMalloc(&dok.p, n1)); HIPCHK(hipMalloc(&dpdf.p, n1)); HIPCHK(hipMalloc(&dinc.p, n1)); HIPCHK(hipMalloc(&dvsp.p, n1));
    HIPCHK(hipMalloc(&dws.p, n3)); HIPCHK(hipMalloc(&dps.p, n1));
    HIPCHK(hipMemcpyAsync(dp.p, p, n3, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(da.p, n_or_wo, n3, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dw.p, wi, n3, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(du.p, u, n2, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_guiding_query, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, r->dscene, is_volume, g, n,
                       (const float *)dp.p, (const float *)da.p, (const float *)dw.p, (const float *)du.p, (int32_t *)dok.p,
                       (float *)dpdf.p, (float *)dinc.p, (float *)dvsp.p, (float *)dws.p, (float *)dps.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out_ok, dok.p, n1, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_pdf, dpdf.p, n1, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_incoming_pdf, dinc.p, n1, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_vsp, dvsp.p, n1, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_ws, dws.p, n3, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out_pdf_s, dps.p, n1, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int vspg_libm_batch(VspgRenderer *r, int n, const float *x, float *logf_out, float *sinf_out, float *cosf_out,
                    void *stream) {
    if (!r || n < 0 || (n > 0 && (!x || !logf_out || !sinf_out || !cosf_out))) return fail(VSPG_EINVAL, "bad arguments");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf dx, dl, ds, dc;
    HIPCHK(hipMalloc(&dx.p, (size_t)n * 4));
    HIPCHK(hipMalloc(&dl.p, (size_t)n * 4));
    HIPCHK(hipMalloc(&ds.p, (size_t)n * 4));
    HIPCHK(hipMalloc(&dc.p, (size_t)n * 4));
    HIPCHK(hipMemcpyAsync(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_libm, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n, (const float *)dx.p, (float *)dl.p,
                       (float *)ds.p, (float *)dc.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(logf_out, dl.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(sinf_out, ds.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(cosf_out, dc.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int vspg_libm_powf_batch(VspgRenderer *r, int n, const float *x, const float *y, float *out, void *stream) {
    if (!r || !x || !y || !out || n <= 0) return fail(VSPG_EINVAL, "bad arguments");
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf dx, dy, dout;
    HIPCHK(hipMalloc(&dx.p, (size_t)n * 4));
    HIPCHK(hipMalloc(&dy.p, (size_t)n * 4));
    HIPCHK(hipMalloc(&dout.p, (size_t)n * 4));
    HIPCHK(hipMemcpyAsync(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dy.p, y, (size_t)n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_libm_powf, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n, (const float *)dx.p, (const float *)dy.p,
                       (float *)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int vspg_blackbody_batch(VspgRenderer *r, int n, const float *u, const float *T, float *out6, void *stream) {
    if (!r || !u || !T || !out6 || n < 0) return fail(VSPG_EINVAL, "null argument");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf du, dT, dout;
    HIPCHK(hipMalloc(&du.p, (size_t)n * 4));
    HIPCHK(hipMalloc(&dT.p, (size_t)n * 4));
    HIPCHK(hipMalloc(&dout.p, (size_t)n * 24));
    HIPCHK(hipMemcpyAsync(du.p, u, (size_t)n * 4, hipMemcpyHostToDevice, s));
    HIPCHK(hipMemcpyAsync(dT.p, T, (size_t)n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_blackbody, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n, (const float *)du.p, (const float *)dT.p, (float *)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out6, dout.p, (size_t)n * 24, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int vspg_renderer_training_stats(VspgRenderer *r, VspgTrainStats *out, void *stream) {
    if (!r || !out) return fail(VSPG_EINVAL, "null argument");
    memset(out, 0, sizeof *out);
    out->training = r->training ? 1 : 0;
    out->iteration = r->field_iteration;
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    if (r->train_counters) {
        unsigned long long cnt[2] = {0, 0};
        HIPCHK(hipMemcpyAsync(cnt, r->train_counters, sizeof cnt, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        out->n_samples = cnt[0];
        out->n_zero = cnt[1];
        out->n_dropped = cnt[0] > r->sample_capacity ? cnt[0] - r->sample_capacity : 0;
        DScene h;
        HIPCHK(hipMemcpyAsync(&h, r->dscene, sizeof h, hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (int f = 0; f < 2; ++f) { out->n_nodes[f] = h.field[f].n_nodes; out->n_regions[f] = h.field[f].n_regions; }
    } else {
        for (int f = 0; f < 2; ++f) { out->n_nodes[f] = r->hscene.field[f].n_nodes; out->n_regions[f] = r->hscene.field[f].n_regions; }
    }
    return 0;
}
int vspg_train_samples_read(VspgRenderer *r, VspgTrainSample *out, size_t max_samples, size_t *n_out, void *stream) {
    if (!r || !n_out) return fail(VSPG_EINVAL, "null argument");
    *n_out = 0;
    if (!r->train_counters) return 0;
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    unsigned long long cnt = 0;
    HIPCHK(hipMemcpyAsync(&cnt, r->train_counters, sizeof cnt, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (cnt > r->sample_capacity) cnt = r->sample_capacity;
    *n_out = (size_t)cnt;
    const size_t n = cnt < max_samples ? (size_t)cnt : max_samples;
    if (out && n) {
        HIPCHK(hipMemcpyAsync(out, r->samples, n * sizeof(VspgTrainSample), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
    }
    return 0;
}
int vspg_renderer_get_guiding_field(VspgRenderer *r, int volume_field, VspgKdNode *nodes, VspgFieldRegion *regions,
                                    int32_t *n_nodes, int32_t *n_regions, void *stream) {
    if (!r || !n_nodes || !n_regions) return fail(VSPG_EINVAL, "null argument");
    const int f = volume_field ? 1 : 0;
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    DScene h;
    HIPCHK(hipMemcpyAsync(&h, r->dscene, sizeof h, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *n_nodes = h.field[f].n_nodes;
    *n_regions = h.field[f].n_regions;
    if (nodes && h.field[f].nodes && *n_nodes > 0)
        HIPCHK(hipMemcpyAsync(nodes, h.field[f].nodes, sizeof(VspgKdNode) * (size_t)*n_nodes, hipMemcpyDeviceToHost, s));
    if (regions && h.field[f].regions && *n_regions > 0)
        HIPCHK(hipMemcpyAsync(regions, h.field[f].regions, sizeof(VspgFieldRegion) * (size_t)*n_regions, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

int vspg_libm_log1m_batch(VspgRenderer *r, int n, const float *x, float *out, void *stream) {
    if (!r || !x || !out || n < 0) return fail(VSPG_EINVAL, "bad argument");
    if (n == 0) return 0;
    HIPCHK(hipSetDevice(r->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    DevBuf dx, dout;
    HIPCHK(hipMalloc(&dx.p, (size_t)n * 4));
    HIPCHK(hipMalloc(&dout.p, (size_t)n * 4));
    HIPCHK(hipMemcpyAsync(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(k_libm_log1m, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, n, (const float *)dx.p, (float *)dout.p);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return 0;
}

#ifdef VSPG_WF_DEBUG
int vspg_dbg_read(unsigned int *out8) {  // diagnostic build only
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_dbg_err), 8 * sizeof(unsigned int)));
    return 0;
}
#endif
#ifdef VSPG_WF_STATS
int vspg_wf_stats_read(unsigned long long *out16) {  // diagnostic build only: read and clear
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wf_stats), 16 * sizeof(unsigned long long)));
    unsigned long long z[16] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_wf_stats), z, sizeof z));
    return 0;
}
#endif
#ifdef VSPG_PROFILE
// diagnostic build only: dump and clear the per-section counters
int vspg_prof_read(unsigned long long *out /* PS_COUNT*3 */, int *n_sections) {
    unsigned long long h[PS_COUNT][3];
    HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_prof), sizeof h));
    memcpy(out, h, sizeof h);
    memset(h, 0, sizeof h);
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_prof), h, sizeof h));
    *n_sections = PS_COUNT;
    return 0;
}
#endif

}  // extern "C"

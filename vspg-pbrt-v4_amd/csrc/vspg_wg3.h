// vspg_wg3.h -- k_render_wave_wg3: the workgroup kernel of vspg_wg_kernel.h WITHOUT workgroup barriers (round 5).
//
// k_render_wave_wg2 advances its pool in lock step -- [segment phase] barrier [vertex phase] barrier -- and the per-phase table of
// round 4 (profiles/r04_phase_profile_fog.txt) put 18 % of the wavefronts' time at those two barriers and 14 % in the scheduler
// between them: a phase lasts as long as its slowest 64-entry chunk, and the vertex phase has work for six of the eight
// wavefronts.  Here the lists are RING QUEUES in LDS and every wavefront is its own scheduler:
//
//   Q_VV / Q_VS  paths standing at a volume / surface vertex         -> vertex chunk  (NEE, Russian roulette, new direction)
//   Q_A          paths with a ray to follow (bit 15: the slot's pixel starts its next sample: multi-sample launches)
//                                                                     -> segment chunk (intersection, SampleDistance, event)
//   Q_F          free slots; 64 of them + one 8x8 pixel tile from the global tile head -> a chunk of camera rays
//
// A wavefront pops up to 64 entries from one queue (reserve with a compare-and-swap on the queue's head), runs the chunk, pushes
// every slot of it into the queue its path now belongs to (ballot + prefix count, one reserving atomic per queue, entries, one
// committing atomic) and looks for the next chunk -- it never waits for its seven siblings.  A consumer trusts entries only
// up to a commit count it has seen EQUAL to the reserve count (every reservation below it has then been written).
//
// Chunks stay FULL: a queue is served once it holds 64 entries; a partial chunk is taken only when the queue's producers are
// idle (vertex queues: no segment chunk in flight; Q_A: no vertex chunk in flight).  The pool holds MORE paths than the
// workgroup has lanes (the grey instantiations: 640 for 512 lanes; what k_render_wave_wg2 could not use -- its pool had to be
// a whole number of rounds of eight chunks) so that a wavefront that finishes a chunk finds a full one waiting: with seven
// siblings busy on 64 slots each, 640 - 448 = 192 slots sit in three kinds of queue, so one kind holds 64 (the two vertex
// queues count as one kind: a vertex chunk may take the rest of the smaller one and fill up from the other).
// Termination: the tile head has run dry and no slot holds a path (`live`, raised by 64 BEFORE a tile is claimed).
//
// Per path nothing changes: the operations, their order and the sampler dimensions are li_segment_a's and li_segment_b's, a
// finished path parks {L, ISG code} exactly as under k_render_wave_wg2 (deferred resolve, vspg_capi.hip): the three schedulers
// give the same film bit for bit (tests: test_workgroup_schedulers_are_bit_identical).
#pragma once
#include <type_traits>

#include "vspg_guided_wg.h"
#include "vspg_path.h"
#include "vspg_wg_kernel.h"

VSPG_NS_BEGIN

enum { Q_VV = 0, Q_VS = 1, Q_A = 2, Q_F = 3, Q_COUNT = 4 };
enum { QC_RES = 0, QC_COM = 1, QC_HEAD = 2, QC_STRIDE = 4 };
enum { W3_BUSY_S = Q_COUNT * QC_STRIDE, W3_BUSY_V, W3_EXH, W3_LIVE, W3_COUNT };  // (one 16-byte group behind the queues')
enum { W3_NONE = 0, W3_VERTEX = 1, W3_SEGMENT = 2, W3_FRESH = 3, W3_EXIT = 4 };
constexpr unsigned kRestartBit = 0x8000u;

// ISG code of a finished sample, one float: 0 = no ISG record, +q = volume event, -q = surface event, q = the VSP the primary
// segment used (or 0.5): q lies in [0.001, 0.999], so the sign is free and nothing is rounded
VDEV float isg_code(const IsgSample &isg) {
    if (!isg.valid) return 0.f;
    const float q = isg.vsp_used >= 0.f ? isg.vsp_used : 0.5f;
    return isg.surface_event ? -q : q;
}
// one parked sample {L, ISG code} into the film and the image-space statistics (RGBFilm::AddSample + ISG AddSample, the
// read-modify-write forms: one writer per pixel)
__device__ __forceinline__ void resolve_sample(float4 s, float4 *film_px, float *isg_px) {
    const Spec L = Spec{s.x, s.y, s.z};
    film_add_sample_rmw(film_px, L);
    IsgSample isg;
    isg.valid = s.w != 0.f;
    isg.surface_event = s.w < 0.f;
    isg.vsp_used = __builtin_fabsf(s.w);
    isg_add_sample_rmw(isg_px, L, isg);
}
constexpr int kWg3Heads = 8, kWg3HeadSetBytes = kWg3Heads * 128;  // k_render_wave_wg3's tile cursors (see the kernel)
// The global work head is a PAIR of counters used by alternate launches: a launch zeroes the one the NEXT launch will use
// (nobody reads it meanwhile, launches of a renderer are stream-ordered), so no memset sits between two waves.
__device__ __forceinline__ void reset_sibling_head(unsigned int *work_head) {
    if (blockIdx.x == 0 && threadIdx.x == 0) *reinterpret_cast<unsigned int *>(reinterpret_cast<uintptr_t>(work_head) ^ 4u) = 0u;
}

// The scheduler's words in LDS: per queue {reserved, committed, head, -} and {segment chunks in flight, vertex chunks in flight, tile
// head dry, slots holding a path}, each group 16 bytes: a deciding lane reads all five groups with five back-to-back 128-bit
// LDS reads and ONE wait.  A 128-bit read is one LDS instruction: {reserved, committed, head} are a snapshot.
typedef unsigned int w3_u32x4 __attribute__((ext_vector_type(4)));
typedef const volatile __attribute__((address_space(3))) w3_u32x4 *w3_lds_v4;
// Every lane hands its slot to the queue its path now belongs to (dq in [0, Q_COUNT), or -1: none): the four reservations are ONE
// returning LDS atomic (lane q reserves for queue q), then the entries, then the four commits as one atomic.
template <int NP>
VDEV void ring_push_all(int dq, unsigned entry, unsigned short (*ring)[NP], unsigned int *s_w) {
    const int lane = threadIdx.x & 63;
    const unsigned long long m0 = __ballot(dq == 0), m1 = __ballot(dq == 1), m2 = __ballot(dq == 2), m3 = __ballot(dq == 3);
    const unsigned n0 = (unsigned)__popcll(m0), n1 = (unsigned)__popcll(m1), n2 = (unsigned)__popcll(m2), n3 = (unsigned)__popcll(m3);
    const unsigned myn = lane == 0 ? n0 : lane == 1 ? n1 : lane == 2 ? n2 : n3;
    unsigned base = 0;
    if (lane < Q_COUNT && myn > 0u) base = atomicAdd(s_w + lane * QC_STRIDE + QC_RES, myn);
    const unsigned b0 = (unsigned)__builtin_amdgcn_readlane((int)base, 0), b1 = (unsigned)__builtin_amdgcn_readlane((int)base, 1);
    const unsigned b2 = (unsigned)__builtin_amdgcn_readlane((int)base, 2), b3 = (unsigned)__builtin_amdgcn_readlane((int)base, 3);
    if (dq >= 0) {
        const unsigned long long m = dq == 0 ? m0 : dq == 1 ? m1 : dq == 2 ? m2 : m3;
        const unsigned b = dq == 0 ? b0 : dq == 1 ? b1 : dq == 2 ? b2 : b3;
        ring[dq][(b + (unsigned)__popcll(m & ((1ull << lane) - 1ull))) % (unsigned)NP] = (unsigned short)entry;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // the entries (and the pool records behind them) before the commits
    if (lane < Q_COUNT && myn > 0u) atomicAdd(s_w + lane * QC_STRIDE + QC_COM, myn);
}

#ifndef VSPG_WG3_IDLE_SLEEP
#define VSPG_WG3_IDLE_SLEEP 8   // s_sleep argument (x 64 clocks) of a wavefront that found no chunk twice in a row
#endif
#ifndef VSPG_WG3_NP_CAP
#define VSPG_WG3_NP_CAP 768
#endif
#ifndef VSPG_WG3_LDS_BUDGET
#define VSPG_WG3_LDS_BUDGET 81920
#endif
#ifndef VSPG_WG3_OTHER
#define VSPG_WG3_OTHER 5200   // scene records, libm tables, counters, queue words beside the pool and the rings (5104 B measured)
#endif
// paths per pool: what fits the LDS a workgroup may use when two share a CU -- the record, four ring entries per path, `other`
// bytes (scene records, counters, staged kd nodes); a multiple of 32
template <class LY>
constexpr int wg3_pool_paths(int other_bytes) {
    const int n = (VSPG_WG3_LDS_BUDGET - other_bytes) / (LY::COUNT * 4 + Q_COUNT * 2) / 32 * 32;
    return n < VSPG_WG3_NP_CAP ? n : VSPG_WG3_NP_CAP;
}

template <class Medium, bool GUIDED, int NP, int kWgBlock, int kWgWavesPerSimd, bool TRAIN = false>
__global__ __launch_bounds__(kWgBlock, kWgWavesPerSimd) void k_render_wave_wg3(
    const DScene *__restrict__ Sp, float4 *__restrict__ film, float *__restrict__ isg_stats, const float *__restrict__ vsp_buf,
    int vsp_ready, int wave_end, int first_sample, int single_sample, PcgJump jump, unsigned int tiles_magic,
    unsigned int *__restrict__ work_head, const float4 *__restrict__ prev_samples,
    float4 *__restrict__ wave_samples, unsigned long long *__restrict__ counters, TrainArgs train = TrainArgs{nullptr, nullptr, nullptr, nullptr, 0, 0}) {
    const DScene &S = *Sp;
    const int W = S.xres, H = S.yres;
    const int tilesX = (W + 7) >> 3, tilesY = (H + 7) >> 3;
    const unsigned n_tiles = (unsigned)(tilesX * tilesY);
    const int lane = threadIdx.x & 63;
    const int sample_step = S.shard_count > 1 ? S.shard_count : 1;
    // The tile head: kWg3Heads cursors, each on a 128-byte line of its own, cursor k handing out tiles k, k + kWg3Heads, ...  One cursor
    // was a returning atomic per 8x8 tile on ONE address -- 32 400 of them per 1080p launch, at the ~88 per microsecond a hot word
    // serves more than half the kernel's time in the atomic unit's queue.  A wavefront starts at its workgroup's cursor (workgroups go
    // round the XCDs) and moves on when one has run out (it looks before it asks: cursors only grow).  The heads are a PAIR of sets
    // used by alternate launches, 1 KB apart: a launch zeroes the set the next one will use (reset_sibling_head's scheme).
    if (blockIdx.x == 0 && threadIdx.x < (unsigned)kWg3Heads)
        reinterpret_cast<unsigned int *>(reinterpret_cast<uintptr_t>(work_head) ^ (uintptr_t)kWg3HeadSetBytes)[threadIdx.x * 32u] = 0u;
    unsigned wseg = blockIdx.x % (unsigned)kWg3Heads, wtried = 0u;

    constexpr bool FULL = !Medium::kSimpleScene;
    static_assert(!FULL || !GUIDED, "the workgroup kernel's guided vertex (vspg_guided_wg.h) is built for rectangle scenes");
    static_assert(NP < (int)kRestartBit, "a ring entry is the slot plus the restart bit");
    using LY = PoolLayout<GUIDED, Medium::kGrey, TRAIN, FULL>;
    constexpr int NF = LY::COUNT;
    static_assert(!TRAIN || GUIDED, "segment recording belongs to the guided instantiations");
    static_assert(Medium::kSingleSegment, "k_render_wave_wg3 serves homogeneous media (grid media: the wavefront pipeline)");
    __shared__ float s_pool[NF * NP];
    __shared__ unsigned short s_ring[Q_COUNT][NP];
    __shared__ __attribute__((aligned(16))) unsigned int s_w[W3_COUNT];
    const Pool P{s_pool, NP};
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
    if (threadIdx.x < W3_COUNT) s_w[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < NP; i += kWgBlock) s_ring[Q_F][i] = (unsigned short)i;
    __syncthreads();
    if (threadIdx.x == 0) { s_w[Q_F * QC_STRIDE + QC_RES] = NP; s_w[Q_F * QC_STRIDE + QC_COM] = NP; }
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
    VSPG_PROF_ACC(prof_decide, PS_WG_R);      // diagnostic build: the scheduler (decisions that found a chunk)
    VSPG_PROF_ACC(prof_idle, PS_WG_BAR_A);    // ... and the polls that found none, the sleep included
    unsigned idle_polls = 0;
    while (true) {
        VSPG_PROF_ACC_BEGIN(prof_decide);
        VSPG_PROF_ACC_BEGIN(prof_idle);
        // ---- this wavefront's next chunk -----------------------------------------------------------------
        // Every lane reads the same five 16-byte groups (an LDS broadcast) and readfirstlane makes scalars of them: the whole
        // decision is SALU work -- the kernel is bound by vector issue, and a wavefront polls here while its siblings compute.
        // Lane 0 performs the claims.  Order: a FULL chunk of one kind (volume vertices, surface vertices, segments, a fresh
        // tile) before a vertex chunk mixed from both vertex queues, before partial chunks (only while their producers idle).
        unsigned kind = W3_NONE, pos0 = 0, n0 = 0, pos1 = 0, n1 = 0, q0 = Q_VV, tile = 0;
        {
            const auto U = [](unsigned v) { return (unsigned)__builtin_amdgcn_readfirstlane((int)v); };
            const w3_lds_v4 qw = (w3_lds_v4)(s_w);
            const w3_u32x4 wVV = qw[Q_VV], wVS = qw[Q_VS], wA = qw[Q_A], wF = qw[Q_F], wM = qw[Q_COUNT];
            const unsigned hVV = U(wVV.z), hVS = U(wVS.z), hA = U(wA.z), hF = U(wF.z);
            const unsigned busyS = U(wM.x), busyV = U(wM.y), live = U(wM.w);
            const bool exh = U(wM.z) != 0u;
            const auto avail = [&](w3_u32x4 q, unsigned head) {  // committed entries nobody has claimed; 0 while a push is between its reservation and its commit
                const unsigned r = U(q.x), c = U(q.y);
                const int a = (int)(c - head);
                return r == c && a > 0 ? a : 0;
            };
            const int aVV = avail(wVV, hVV), aVS = avail(wVS, hVS), aA = avail(wA, hA), aF = exh ? 0 : avail(wF, hF);
            const auto claim = [&](unsigned q, unsigned head, unsigned n) {  // compare-and-swap on the queue's head: this wavefront owns [head, head + n)
                unsigned old = head + 1u;
                if (lane == 0) old = atomicCAS(s_w + q * QC_STRIDE + QC_HEAD, head, head + n);
                return U(old) == head;
            };
            const auto bump = [&](int w, int d) { if (lane == 0) atomicAdd(s_w + w, (unsigned)d); };
            const auto claim_vertex = [&](int a_first, unsigned h_first, unsigned q_first, int a_second, unsigned h_second, unsigned q_second) {
                // the larger queue first; a chunk it cannot fill takes the rest from the other one
                bump(W3_BUSY_V, 1);
                const unsigned na = (unsigned)(a_first < 64 ? a_first : 64);
                if (na > 0u && claim(q_first, h_first, na)) {
                    kind = W3_VERTEX; q0 = q_first; pos0 = h_first; n0 = na;
                    const unsigned nb = (unsigned)(a_second < 64 - (int)na ? a_second : 64 - (int)na);
                    if (nb > 0u && claim(q_second, h_second, nb)) { pos1 = h_second; n1 = nb; }
                } else {
                    bump(W3_BUSY_V, -1);
                }
            };
            const auto claim_segment = [&]() {
                bump(W3_BUSY_S, 1);
                const unsigned na = (unsigned)(aA < 64 ? aA : 64);
                if (claim(Q_A, hA, na)) { kind = W3_SEGMENT; pos0 = hA; n0 = na; }
                else bump(W3_BUSY_S, -1);
            };
            const int aV = aVV + aVS;
            if (aVV >= 64) claim_vertex(aVV, hVV, Q_VV, 0, hVS, Q_VS);
            else if (aVS >= 64) claim_vertex(aVS, hVS, Q_VS, 0, hVV, Q_VV);
            else if (aA >= 64) claim_segment();
            else if (aF >= 64) {
                bump(W3_BUSY_S, 1);
                bump(W3_LIVE, 64);  // BEFORE the tile is claimed: `live == 0` then says nobody can still start a path
                if (claim(Q_F, hF, 64u)) {
                    kind = W3_FRESH; pos0 = hF; n0 = 64u;
                    tile = n_tiles;
                    while (wtried < (unsigned)kWg3Heads) {
                        unsigned c = 0xffffffffu;
                        if (lane == 0 && __hip_atomic_load(work_head + wseg * 32u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) * (unsigned)kWg3Heads + wseg < n_tiles)
                            c = atomicAdd(work_head + wseg * 32u, 1u);
                        c = U(c);
                        if (c != 0xffffffffu && c * (unsigned)kWg3Heads + wseg < n_tiles) {
                            tile = c * (unsigned)kWg3Heads + wseg;
                            break;
                        }
                        wseg = (wseg + 1u) % (unsigned)kWg3Heads;
                        ++wtried;
                    }
                } else {
                    bump(W3_LIVE, -64);
                    bump(W3_BUSY_S, -1);
                }
            } else if (aV >= 64 || (aV > 0 && busyS == 0u)) {
                if (aVV >= aVS) claim_vertex(aVV, hVV, Q_VV, aVS, hVS, Q_VS);
                else claim_vertex(aVS, hVS, Q_VS, aVV, hVV, Q_VV);
            } else if (aA > 0 && busyV == 0u) {
                claim_segment();
            } else if (exh && live == 0u) {
                kind = W3_EXIT;
            }
        }
        if (kind == W3_EXIT) break;
        if (kind == W3_NONE) {
            if (idle_polls < 2u) __builtin_amdgcn_s_sleep(2);
            else __builtin_amdgcn_s_sleep(VSPG_WG3_IDLE_SLEEP);
            // (safety valve: a wavefront that has found nothing for ~10^7 polls -- seconds -- leaves instead of hanging the device;
            //  the launch's counters and film then show the loss)
            VSPG_PROF_ACC_END(prof_idle);
            if (++idle_polls > (1u << 23)) break;
            continue;
        }
        VSPG_PROF_ACC_END(prof_decide);
        idle_polls = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

        if (kind == W3_VERTEX) {
            // ---- V: vertex processing (NEE, Russian roulette, new direction) ----------------------------
            VSPG_PROF(PS_WG_B);
            bool cont = false, restart = false, freed = false;
            int slot = 0;
            if ((unsigned)lane < n0 + n1) {
                slot = (unsigned)lane < n0 ? s_ring[q0][(pos0 + (unsigned)lane) % (unsigned)NP]
                                           : s_ring[q0 ^ 1u][(pos1 + ((unsigned)lane - n0)) % (unsigned)NP];
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
                    restart = !single_sample && s2 < wave_end;
                    freed = !restart;
                }
            }
            ring_push_all<NP>(cont || restart ? Q_A : (freed ? Q_F : -1), (unsigned)slot | (restart ? kRestartBit : 0u), s_ring, s_w);
            const unsigned n_freed = (unsigned)__popcll(__ballot(freed));
            if (lane == 0) {
                if (n_freed) atomicSub(s_w + W3_LIVE, n_freed);
                atomicSub(s_w + W3_BUSY_V, 1u);
            }
            continue;
        }

        // ---- S: camera ray + primary segment for new paths, one secondary segment for the others ------
        {
            VSPG_PROF(PS_WG_A);
            bool toVV = false, toVS = false, toA = false, restart = false, freed = false;
            int slot = 0;
            const bool fresh = kind == W3_FRESH;
            const bool tile_ok = !fresh || tile < n_tiles;  // (the head has run dry: the 64 slots go back)
            if (fresh && !tile_ok) {
                slot = s_ring[Q_F][(pos0 + (unsigned)lane) % (unsigned)NP];
                freed = true;
            } else if ((unsigned)lane < n0) {
                const unsigned e = s_ring[fresh ? Q_F : Q_A][(pos0 + (unsigned)lane) % (unsigned)NP];
                slot = (int)(e & (kRestartBit - 1u));
                const bool primary = fresh || (e & kRestartBit) != 0u;
                Sampler sampler;
                PathState st;
                IsgSample isg;
                int ch = 0, pxy = 0;
                Vertex vx;
                bool alive = false, valid = true;
                int seg = LI_END;
                if (primary) {
                    int px, py, smp;
                    if (fresh) {
                        unsigned ty = tilesX == 1 ? tile : __umulhi(tile, tiles_magic);
                        unsigned tx = tile - ty * (unsigned)tilesX;
                        while (tx >= (unsigned)tilesX) { tx -= (unsigned)tilesX; ty++; }
                        px = (int)(tx * 8u + ((unsigned)lane & 7u));
                        py = (int)(ty * 8u + ((unsigned)lane >> 3));
                        pxy = px | (py << 16);
                        smp = first_sample;
                        // the PREVIOUS one-sample launch parked this pixel's sample (vspg_render_wave: deferred resolve): it enters
                        // the film now, before this launch's sample of the pixel can (same order of additions as ever)
                        if (prev_samples != nullptr && px < W && py < H) {
                            const size_t pidx = (size_t)py * W + px;
                            resolve_sample(prev_samples[pidx], film + pidx, isg_stats + pidx * VSPG_ISG_STATS);
                        }
                    } else {
                        pxy = P.i(LY::PIXEL, slot);
                        smp = P.i(LY::SAMPLE, slot);
                        px = pxy & 0xffff;
                        py = (int)((unsigned)pxy >> 16);
                    }
                    valid = px < W && py < H && smp < wave_end;  // tile padding: the slot stays free
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
                    toA = true;
                } else if (alive) {
                    toVV = vx.volume;
                    toVS = !vx.volume;
                } else if (valid) {
                    emit(pxy, st.L, isg);
                    if constexpr (TRAIN) train.seg_count[rec_bind(pxy)] = pc.rec.n;  // PropagateSamples (:627) follows in k_propagate
                    pc.path();
                    const int s2 = P.i(LY::SAMPLE, slot) + sample_step;
                    P.i(LY::SAMPLE, slot) = s2;
                    restart = !single_sample && s2 < wave_end;
                    freed = !restart;
                }
            }
            ring_push_all<NP>(toVV ? Q_VV : toVS ? Q_VS : (toA || restart) ? Q_A : (freed ? Q_F : -1), (unsigned)slot | (restart ? kRestartBit : 0u), s_ring, s_w);
            const unsigned n_freed = (unsigned)__popcll(__ballot(freed));
            if (lane == 0) {
                if (fresh && !tile_ok) atomicExch(s_w + W3_EXH, 1u);
                if (n_freed) atomicSub(s_w + W3_LIVE, n_freed);
                atomicSub(s_w + W3_BUSY_S, 1u);
            }
        }
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

// ---- host side: the unguided rectangle-scene instantiations behind one call (what vspg_capi.hip launches for them, and what the
// fast-arithmetic translation units export: vspg_fast.hip, vspg_arith.h) -------------------------------------------------------
#ifndef VSPG_WG_WAVES
#define VSPG_WG_WAVES 4
#endif
#ifndef VSPG_WG_BLOCK
#define VSPG_WG_BLOCK 512
#endif
struct Wg3Launch {
    const DScene *dscene;
    float4 *film;
    float *isg_stats;
    const float *vsp;
    int vsp_ready, wave_end, first_sample, single_sample;
    PcgJump jump;
    unsigned int tiles_magic;
    unsigned int *work_head;
    const float4 *ws_prev;
    float4 *ws_out;
    unsigned long long *counters;
    unsigned int blocks;
    hipStream_t stream;
    int grey;          // 0: chromatic; 1: grey medium; 2: grey medium and grey surfaces (HomogeneousMediumT)
    int null_zero;     // ... whose null-collision coefficient is exactly 0
};
template <int GREY> constexpr int kWg3PoolHomogT = wg3_pool_paths<PoolLayout<false, GREY>>(VSPG_WG3_OTHER);
inline int wg3_launch_unguided(const Wg3Launch &L) {
#define VSPG_WG3_GO(M, NPOOL)                                                                                                                \
    hipLaunchKernelGGL((k_render_wave_wg3<M, false, NPOOL, VSPG_WG_BLOCK, VSPG_WG_WAVES, false>), dim3(L.blocks), dim3(VSPG_WG_BLOCK), 0, L.stream, \
                       L.dscene, L.film, L.isg_stats, L.vsp, L.vsp_ready, L.wave_end, L.first_sample, L.single_sample, L.jump, L.tiles_magic,  \
                       L.work_head, L.ws_prev, L.ws_out, L.counters, TrainArgs{nullptr, nullptr, nullptr, nullptr, 0, 0})
    if (L.grey >= 2 && L.null_zero) VSPG_WG3_GO(HomogeneousMediumGreySceneNullZero, kWg3PoolHomogT<2>);
    else if (L.grey >= 2) VSPG_WG3_GO(HomogeneousMediumGreyScene, kWg3PoolHomogT<2>);
    else if (L.grey == 1) VSPG_WG3_GO(HomogeneousMediumGrey, kWg3PoolHomogT<1>);
    else VSPG_WG3_GO(HomogeneousMediumSimple, kWg3PoolHomogT<0>);
#undef VSPG_WG3_GO
    return (int)hipGetLastError();
}

VSPG_NS_END  // namespace vspg

// vspg_train.h -- guiding-cache training on device.
//
// Recording follows the reference's hooks (src/pbrt/cpu/guiding.h:682-832 and their call sites in
// guidedvolpathvspgintegrator.cpp): a path keeps one segment record per vertex.  What OpenPGL does
// with them -- PathSegmentStorage::PropagateSamples and Field::Update -- is absent from the reference
// tree and is this build's own design (parity unpinned); oracle/vspg_oracle.c states
// the same definitions on the CPU.  Radiance samples are bit-identical to the oracle's; the fitted
// field agrees within float-summation-order tolerance (float atomics here, doubles there).
#pragma once
#include "vspg_guiding.h"

VSPG_NS_BEGIN

// ---- segment records: per lane, SoA over lanes in HBM -----------------------------------------
enum {
    SG_P = 0,        // 3  vertex position
    SG_WI = 3,       // 3  sampled direction
    SG_PDF = 6,      // 1
    SG_MI = 7,       // 1  MIS weight of the emission seen at the vertex
    SG_RR = 8,       // 1  survival probability
    SG_SW = 9,       // 3  scattering weight
    SG_T = 12,       // 3  transmittance weight of the segment LEAVING the vertex
    SG_DIRECT = 15,  // 3
    SG_SCAT = 18,    // 3  NEE estimate at the vertex
    SG_FLAGS = 21,   // 1  bit0 has_wi, bit1 volume, bit2 is_delta
    SG_FLOATS = 22
};

struct NullRecorder {  // every hook compiles to nothing
    static constexpr bool kActive = false;
    VDEV void new_segment(V3, bool) const {}
    VDEV void add_transmittance_weight(Spec) const {}
    VDEV void add_surface_emission(Spec, float) const {}
    VDEV void add_scattered_direct_light(Spec) const {}
    VDEV void add_scatter_data(bool, Spec, V3, float, float, float) const {}
    VDEV void add_infinite_light_emission(V3, Spec, float) const {}
};
// pss->Reserve(maxDepth >= 1 ? maxDepth * 2 : 30) (guidedvolpathvspgintegrator.cpp:135-138): the records a path may keep;
// NextSegment() returns nullptr beyond it (what OpenPGL's storage does then is restated, not pinned: the library is absent).
// The record buffer of a wave holds min(2 * maxdepth, 64) records per path (88 B each, one column per pixel: 11.7 GB at 1080p
// when all 64 are in use); a deeper path keeps rendering and stops recording there -- no configuration is refused for its depth.
constexpr int kTrainMaxSeg = 64;
__host__ __device__ inline int train_rec_capacity(int maxdepth) {
    const int c = maxdepth >= 1 ? maxdepth * 2 : 30;
    return c < kTrainMaxSeg ? c : kTrainMaxSeg;
}
constexpr float kGuidingInfiniteLightDistance = 1e6f;  // integrators.h:608

// Records live in HBM (a column per lane).  A record's groups of fields are written only when the path sets them; the
// flags word says which groups are present and PropagateSamples substitutes the defaults for the others (so a new
// segment costs 4 stores instead of 22, and the propagation loads what exists).  The current record's flags and its
// NEE sum are kept in registers and written through: no read-modify-write of HBM in the path code.
enum : uint32_t {
    SGF_HAS_WI = 1u,      // add_scatter_data ran: WI, PDF, SW, RR present
    SGF_VOLUME = 2u,
    SGF_DELTA = 4u,
    SGF_T = 8u,           // transmittance weight present (default 1,1,1)
    SGF_DIRECT = 16u,     // DIRECT + MI present (defaults 0 and 1)
    SGF_SCAT = 32u,       // SCAT present (default 0)
};
struct PathRecorder {
    static constexpr bool kActive = true;
    float *base;  // this lane's column: element (seg, field) at base[(seg * SG_FLOATS + field) * stride]
    int stride, max_seg;
    int n, cur;  // cur: the reference's pathSegmentData pointer, -1 = nullptr
    uint32_t cur_flags;     // flags of record `cur`
    float scat_r, scat_g, scat_b;  // its accumulated scattered direct light
    VDEV float &at(int seg, int f) const {
        VSPG_DBG_CHECK(seg >= 0 && seg < max_seg && f >= 0 && f < SG_FLOATS, 3);
        return base[(size_t)(seg * SG_FLOATS + f) * (size_t)stride];
    }
    VDEV uint32_t &flags(int seg) const { return reinterpret_cast<uint32_t &>(at(seg, SG_FLAGS)); }
    VDEV void set3(int seg, int f, float x, float y, float z) const { at(seg, f) = x; at(seg, f + 1) = y; at(seg, f + 2) = z; }
    VDEV void reset() { n = 0; cur = -1; cur_flags = 0; scat_r = scat_g = scat_b = 0.f; }
    // guiding_newSurfacePathSegment / guiding_newVolumePathSegment (:682-732)
    VDEV void new_segment(V3 p, bool volume) {
        if (n >= max_seg) { cur = -1; return; }  // NextSegment() == nullptr
        const int s = n;
        set3(s, SG_P, p.x, p.y, p.z);
        cur_flags = volume ? SGF_VOLUME : 0u;
        flags(s) = cur_flags;
        scat_r = scat_g = scat_b = 0.f;
        cur = n++;
    }
    VDEV void add_transmittance_weight(Spec T) {  // :754-764
        if (cur < 0) return;
        T = clamp_zero(T);
        set3(cur, SG_T, T.r, T.g, T.b);
        cur_flags |= SGF_T;
        flags(cur) = cur_flags;
    }
    VDEV void add_surface_emission(Spec Le, float w) {  // :744-752
        if (cur < 0) return;
        Le = clamp_zero(Le);
        set3(cur, SG_DIRECT, Le.r, Le.g, Le.b);
        at(cur, SG_MI) = w;
        cur_flags |= SGF_DIRECT;
        flags(cur) = cur_flags;
    }
    VDEV void add_scattered_direct_light(Spec Ld) {  // :734-742
        if (cur < 0) return;
        Ld = clamp_zero(Ld);
        scat_r = scat_r + Ld.r;
        scat_g = scat_g + Ld.g;
        scat_b = scat_b + Ld.b;
        set3(cur, SG_SCAT, scat_r, scat_g, scat_b);
        cur_flags |= SGF_SCAT;
        flags(cur) = cur_flags;
    }
    // guiding_addInfiniteLightEmission (guiding.h:759-784): a NEW segment at ray.o + guidingInfiniteLightDistance * ray.d that
    // carries the light's emission and its MIS weight (one per infinite light an escaped ray sees)
    VDEV void add_infinite_light_emission(V3 p, Spec Le, float w) {
        new_segment(p, false);
        add_surface_emission(Le, w);
    }
    // guiding_addSurfaceData / guiding_addVolumeData (:791-832)
    VDEV void add_scatter_data(bool volume, Spec weight, V3 wi, float pdf, float roughness, float survivalProb) {
        if (cur < 0) return;
        weight = clamp_zero(weight);
        set3(cur, SG_WI, wi.x, wi.y, wi.z);
        at(cur, SG_PDF) = pdf;
        set3(cur, SG_SW, weight.r, weight.g, weight.b);
        at(cur, SG_RR) = survivalProb;
        // (the reference resets the transmittance weight to 1 here: drop the group) and rewrites the kind bits
        cur_flags = (cur_flags & (SGF_DIRECT | SGF_SCAT)) | SGF_HAS_WI | (volume ? SGF_VOLUME : 0u) | (roughness < 0.001f ? SGF_DELTA : 0u);
        flags(cur) = cur_flags;
    }
};

struct TrainArgs {  // a18: where a training launch records (all null / 0 otherwise)
    float *segbuf;                   // segment records, one column per work item (pixel of the wave): record (seg, field) of
                                     // item i at segbuf[(seg * SG_FLOATS + field) * n_items + i]
    int *seg_count;                  // records written per work item (0 = no path)
    VspgTrainSample *samples;        // radiance samples of this wave
    unsigned long long *counters;    // [0] samples appended, [1] zero-valued samples dropped
    unsigned long long capacity;
    unsigned int n_items;
};

// Sample sink of k_propagate: a wavefront stages its samples in LDS (ballot + prefix count, no atomics at all) and the
// workgroup reserves its range of the global buffer with ONE returning atomic -- the counter is a single address, and
// same-address returning atomics retire at ~10 ns each on this chip: one per wavefront per loop iteration (2*10^5 of
// them for a 1080p wave) made the propagation a 2 ms kernel.
struct StageSink {
    VspgTrainSample *stage;  // this wavefront's LDS staging area
    unsigned int cap;        // its capacity in samples
    unsigned int count;      // staged so far (wave-uniform)
    VspgTrainSample *samples;
    unsigned long long *counters;
    unsigned long long capacity;
    VDEV void flush_wave() {  // overflow path (paths longer than the staging area was sized for): one atomic per wavefront
        if (count == 0) return;
        const int lane = threadIdx.x & 63;
        unsigned long long base = 0;
        if (lane == 0) base = atomicAdd(&counters[0], (unsigned long long)count);
        base = __shfl(base, 0);
        for (unsigned int j = (unsigned int)lane; j < count; j += 64u)
            if (base + j < capacity) samples[base + j] = stage[j];
        count = 0;
    }
    VDEV void append(bool emit, const VspgTrainSample &smp) {
        const unsigned long long m = __ballot(emit);
        if (m == 0ull) return;
        const unsigned int k = (unsigned int)__popcll(m);
        if (count + k > cap) flush_wave();
        const int lane = threadIdx.x & 63;
        if (emit) stage[count + (unsigned int)__popcll(m & ((1ull << lane) - 1ull))] = smp;
        count += k;
    }
};

// PathSegmentStorage::PropagateSamples stand-in (see oracle/vspg_oracle.c:propagate_samples for the
// definition): walk the path's records from the last vertex to the first, in lock step across the wave.
// The kernel is bound by its record loads (1.1 GB per 1080p wave when every field of every record is read), so:
//   * the flags words of the path's records are read first, all at once, and packed six bits apiece (records past the tenth
//     count as "every group present"), so that
//   * a record's groups are loaded only where the path wrote them, with no load depending on another, and
//   * the record below is in flight while this one is processed (two register sets, the loop unrolled by two).
struct SegRegs { float v[SG_FLOATS]; };
VDEV unsigned int propagate_samples(const PathRecorder &rec, bool active, StageSink &sink) {  // returns the lane's zero-valued samples
    const int n = active ? rec.n : 0;
    int nmax = n;
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(nmax, off);
        nmax = o > nmax ? o : nmax;
    }
    constexpr int kPacked = 10;
    unsigned long long packed = 0ull;
    {
        uint32_t f[kPacked];
#pragma unroll
        for (int i = 0; i < kPacked; ++i) f[i] = i < n ? rec.flags(i) : 0u;
#pragma unroll
        for (int i = 0; i < kPacked; ++i) packed |= (unsigned long long)(f[i] & 63u) << (6 * i);
    }
    Spec Lout_next = sp(0.f);
    V3 p_next = mk(0, 0, 0);
    bool have_next = false, next_volume = false;
    unsigned int zero = 0;
    const auto load = [&](SegRegs &R, int i) {
        if (i < 0 || i >= n) return;
        const uint32_t m = i < kPacked ? (uint32_t)(packed >> (6 * i)) & 63u : 63u;
        R.v[SG_FLAGS] = i < kPacked ? __builtin_bit_cast(float, m) : rec.at(i, SG_FLAGS);
        R.v[SG_P] = rec.at(i, SG_P); R.v[SG_P + 1] = rec.at(i, SG_P + 1); R.v[SG_P + 2] = rec.at(i, SG_P + 2);
        if (m & SGF_HAS_WI) {
            R.v[SG_WI] = rec.at(i, SG_WI); R.v[SG_WI + 1] = rec.at(i, SG_WI + 1); R.v[SG_WI + 2] = rec.at(i, SG_WI + 2);
            R.v[SG_PDF] = rec.at(i, SG_PDF);
            R.v[SG_SW] = rec.at(i, SG_SW); R.v[SG_SW + 1] = rec.at(i, SG_SW + 1); R.v[SG_SW + 2] = rec.at(i, SG_SW + 2);
            R.v[SG_RR] = rec.at(i, SG_RR);
        }
        if (m & SGF_T) { R.v[SG_T] = rec.at(i, SG_T); R.v[SG_T + 1] = rec.at(i, SG_T + 1); R.v[SG_T + 2] = rec.at(i, SG_T + 2); }
        if (m & SGF_DIRECT) {
            R.v[SG_DIRECT] = rec.at(i, SG_DIRECT); R.v[SG_DIRECT + 1] = rec.at(i, SG_DIRECT + 1); R.v[SG_DIRECT + 2] = rec.at(i, SG_DIRECT + 2);
            R.v[SG_MI] = rec.at(i, SG_MI);
        }
        if (m & SGF_SCAT) { R.v[SG_SCAT] = rec.at(i, SG_SCAT); R.v[SG_SCAT + 1] = rec.at(i, SG_SCAT + 1); R.v[SG_SCAT + 2] = rec.at(i, SG_SCAT + 2); }
    };
    const auto process = [&](const SegRegs &R, int i) {
        const float *cur = R.v;
        bool emit = false;
        VspgTrainSample smp;
        if (i < n) {
            const uint32_t fl = __builtin_bit_cast(uint32_t, cur[SG_FLAGS]);
            const bool has_wi = (fl & SGF_HAS_WI) != 0, volume = (fl & SGF_VOLUME) != 0, is_delta = (fl & SGF_DELTA) != 0;
            const V3 p = V3{cur[SG_P], cur[SG_P + 1], cur[SG_P + 2]};
            const Spec T = (fl & SGF_T) ? Spec{cur[SG_T], cur[SG_T + 1], cur[SG_T + 2]} : sp(1.f);
            const float pdf = has_wi ? cur[SG_PDF] : 0.f;
            const Spec Lin = have_next ? T * Lout_next : sp(0.f);
            if (has_wi && !is_delta && have_next && pdf > 0) {
                const float w = avg(Lin) / pdf;
                if (w > 0 && !isinf_(w)) {
                    emit = true;
                    smp.p[0] = p.x; smp.p[1] = p.y; smp.p[2] = p.z;
                    smp.dir[0] = cur[SG_WI]; smp.dir[1] = cur[SG_WI + 1]; smp.dir[2] = cur[SG_WI + 2];
                    smp.weight = w;
                    smp.pdf = pdf;
                    smp.distance = len(p_next - p);
                    smp.flags = (volume ? VSPG_SAMPLE_VOLUME : 0u) | (next_volume ? VSPG_SAMPLE_NEXT_VOLUME : 0u);
                } else {
                    zero++;
                }
            }
            const bool has_direct = (fl & SGF_DIRECT) != 0;
            const Spec direct = has_direct ? Spec{cur[SG_DIRECT], cur[SG_DIRECT + 1], cur[SG_DIRECT + 2]} : sp(0.f);
            const float mi = has_direct ? cur[SG_MI] : 1.f;
            const Spec scat = (fl & SGF_SCAT) ? Spec{cur[SG_SCAT], cur[SG_SCAT + 1], cur[SG_SCAT + 2]} : sp(0.f);
            Spec Lout = direct * mi + scat;
            if (has_wi) {
                const Spec sw = Spec{cur[SG_SW], cur[SG_SW + 1], cur[SG_SW + 2]};
                Lout = Lout + (sw * Lin) / cur[SG_RR];
            }
            Lout_next = Lout;
            p_next = p;
            have_next = true;
            next_volume = volume;
        }
        sink.append(emit, smp);
    };
    SegRegs A, B;
#pragma unroll
    for (int f = 0; f < SG_FLOATS; ++f) A.v[f] = B.v[f] = 0.f;
    int i = nmax - 1;
    load(A, i);
    while (i >= 0) {
        load(B, i - 1);
        process(A, i);
        if (--i < 0) break;
        load(A, i - 1);
        process(B, i);
        --i;
    }
    return zero;
}

// ---- Field::Update stand-in (definition: oracle/vspg_oracle.c "Field::Update") ------------------
constexpr float kTrainSplitCount = 4096.0f;
constexpr float kTrainDecay = 0.75f;
constexpr int kTrainMaxDepth = 24;
constexpr float kTrainWeightClamp = 32.0f;
constexpr float kTrainKappaInit = 2.0f;
constexpr unsigned long long kTrainMinUpdateSamples = 128;  // guidedvolpathvspgintegrator.cpp:238
constexpr int kTrainCapNodes = 8192, kTrainCapRegions = 4097;
constexpr int kTrainKeys = 2 * kTrainCapRegions;  // sort key of a sample in an update: field * kTrainCapRegions + region

struct RegionStats {  // decayed sufficient statistics of one region
    float n;
    float sum_p[3], sum_p2[3];
    float S[GK], R[3][GK], D[GK], V[GK], Qv[GK], Qs[GK];
    int32_t depth;
};
constexpr int kStatFloats = 7 + 8 * GK;  // every float member of RegionStats, in order
static_assert(sizeof(RegionStats) == (kStatFloats + 1) * 4, "RegionStats layout");

// Accumulation over samples SORTED by key (counting sort, k_train_lookup / k_train_scan / k_train_scatter):
// a wavefront walks a contiguous piece of the sorted order, so almost every 64-sample group belongs to one
// key.  The group's NV values are summed ACROSS the lanes by a transposing butterfly: at the step with lane distance o
// a lane hands the half of its values its partner is responsible for to that partner and adds the half it receives --
// P - 1 exchanges for P values instead of 6 P -- and ends up holding the group's total of ONE value (index lane >> kShift).
// Each lane keeps the running sum of its value for the current key, and when the key changes (or the piece ends) the
// wavefront flushes all of them with ONE atomic instruction.  Hot keys -- the whole field at iteration 0 -- thus cost one
// atomic per value per wavefront piece instead of one per sample.
template <int NV>
struct RunAccumulator {
    static_assert(NV <= 64, "one lane per statistic");
    static constexpr int P = NV <= 1 ? 1 : NV <= 2 ? 2 : NV <= 4 ? 4 : NV <= 8 ? 8 : NV <= 16 ? 16 : NV <= 32 ? 32 : 64;  // NV padded to a power of two
    static constexpr int kShift = P == 1 ? 6 : P == 2 ? 5 : P == 4 ? 4 : P == 8 ? 3 : P == 16 ? 2 : P == 32 ? 1 : 0;
    float acc;      // running sum of statistic (lane >> kShift) for `key` (the lanes sharing a statistic hold the same sum)
    int key;        // wave-uniform
    float *dst;
    int first;
    VDEV void init(float *d, int f) { acc = 0.f; key = -1; dst = d; first = f; }
    VDEV void flush() {
        const int lane = threadIdx.x & 63, v = lane >> kShift;
        if (key >= 0 && (lane & ((1 << kShift) - 1)) == 0 && v < NV && acc != 0.f) atomicAdd(&dst[(size_t)key * kStatFloats + first + v], acc);
        acc = 0.f;
    }
    // all lanes call; `valid` lanes contribute vals[0 .. NV) to their key `k` (vals[NV .. P) must be zero)
    VDEV void add(bool valid, int k, const float (&vals)[P]) {
        const int lane = threadIdx.x & 63;
        unsigned long long todo = __ballot(valid);
        while (todo != 0ull) {
            const int leader = __ffsll((long long)todo) - 1;
            const int lkey = __shfl(k, leader);
            const unsigned long long same = __ballot(valid && k == lkey) & todo;
            const bool mine = (same >> lane) & 1ull;
            if (lkey != key) {
                flush();
                key = lkey;
            }
            float x[P];
#pragma unroll
            for (int v = 0; v < P; ++v) x[v] = mine ? vals[v] : 0.f;
            int o = 32;
#pragma unroll
            for (int c = P; c > 1; c >>= 1, o >>= 1) {
                const bool upper = (lane & o) != 0;
#pragma unroll
                for (int j = 0; j < c / 2; ++j) {
                    const float a = x[j], b = x[j + c / 2];
                    x[j] = (upper ? b : a) + __shfl_xor(upper ? a : b, o);
                }
            }
            float t = x[0];
            for (; o > 0; o >>= 1) t += __shfl_xor(t, o);
            acc += t;
            todo &= ~same;
        }
    }
};

VDEV V3 train_reaim(const VspgFieldRegion &R, V3 p, V3 w, float dist) {
    if (!(dist > 0) || isinf_(dist)) return w;
    V3 t = (p - ld3(R.pivot)) + w * dist;
    float l2 = len2(t);
    if (!(l2 > 0)) return w;
    return normalize(t);
}
VDEV void region_init_lobes(VspgFieldRegion &R) {
    const float c = 0.57735026918962576451f;
    R.n_lobes = GK;
    for (int k = 0; k < GK; ++k) {
        R.weight[k] = 1.0f / GK;
        R.kappa[k] = kTrainKappaInit;
        R.mu[0][k] = (k & 1) ? -c : c;
        R.mu[1][k] = (k & 2) ? -c : c;
        R.mu[2][k] = (k & 4) ? -c : c;
        R.distance[k] = kInf;
        R.vsp[k] = 0.5f;
    }
}

VSPG_NS_END  // namespace vspg

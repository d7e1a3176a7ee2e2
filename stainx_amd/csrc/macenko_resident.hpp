// Macenko transform, TILE-RESIDENT form: ONE launch, every pixel read from HBM once and written once -- included by macenko.hip.
//
// The four-pass form (macenko.hip) reads a tile four times because every per-tile quantity needs the one before it (moments ->
// plane -> angle percentiles -> stain vectors -> concentration percentiles -> scale) and each percentile needs all pixels.  MI355X
// has 128 MB of vector registers: a tile does not have to leave the chip between those steps if its pixels are small enough.
// They are: image pixels are 8-bit grey levels -- uint8 tiles as they stand, and float tiles made from decoded images
// (`u8 / 255`, what ToDtype(float32, scale=True) and the reference's benchmarks produce) hold one of 256 values per channel.
//
//   * a workgroup of 512 threads (one per CU) owns up to 65536 pixels of a tile, 128 per thread, as 2 x 3 x 16 registers of 8-bit
//     CODES; a float element x is code k iff its bits are those of T(k / 255) -- checked for every element while it is loaded,
//     so the codes are a lossless copy.  The two per-pixel functions everything else is made of (log2(255 x + 1), and the optical
//     density ln240 - ln2 log2(.)) come from a 256-entry table in LDS filled with the very expressions the other forms evaluate
//     per pixel: every pixel gets the bits it gets there;
//   * phase L reads the tile (HBM-bound) and accumulates the ten raw moments exactly as stats_item does (same work items, same
//     fp32 runs, same fp64 order: the covariance has the four-pass form's bits);
//   * the four order statistics are found ON CHIP, exactly, without a sample, a bracket or a speculation: a sweep over the
//     registers bins every key into 1024 value-aligned bins (LDS atomics, 8 bank-striped copies), the bin that holds the wanted rank
//     is picked, a second sweep collects the keys of that bin (<= 4096) and a byte-wise radix select over that list gives the
//     element.  A crowded bin (heavy ties, adversarial data) is split again -- integer bins over the bin's own key range -- until
//     the list fits or one key value is left.  Counts are integers, so the result does not depend on any order of events;
//   * a tile of more than 65536 pixels is shared by G = ceil(items / 4) workgroups (512 x 512: four, on one XCD where the
//     dispatcher deals blocks round-robin).  They exchange their partial sums / histograms / lists through the workspace: write-
//     through (sc1) stores, every wave drained, one arrival add per workgroup, a bounded relaxed poll, sc1 loads -- the hand-off
//     form of MI355X_MICROARCH.md ("Valid forms", first table row).  Every workgroup of a tile then takes the same decisions
//     from the same totals.  All workgroups of the launch are resident at once (grid <= CUs), every wait is bounded;
//   * phase R reconstructs from the codes and writes the output (non-temporal 16-byte stores).
//   * a float tile that is NOT made of 8-bit levels (augmented / resampled data) is noticed in phase L; its workgroups run the
//     same steps with the pixels re-read from memory in every sweep instead of taken from registers -- same results as the
//     four-pass form, about its speed.
//
// HBM traffic of a call = its algorithmic bytes (one read + one write); nothing is speculated, so real tissue costs what
// synthetic tiles cost.
#pragma once

namespace sx {
namespace macenko {

#ifndef SX_RES_HALVES
#define SX_RES_HALVES 2
#endif
constexpr int kResHalves = SX_RES_HALVES;                     // work items a thread takes part in: 1 (1024 threads, 64 pixels and 128 registers each) or 2 (512 threads,
                                                              // 128 pixels and 256 registers each) -- the same 65536 pixels and 192 KB of codes per workgroup either way
constexpr int kResThreads = 1024 / kResHalves;
constexpr int kResQuads = kResHalves * kResThreads / kStreamThreads;      // work items (stats_item's grouping: 256 threads each) a workgroup holds: 4
constexpr int kResPx = 64;                                    // pixels per thread and work item
constexpr int kResBins = 1024;                                // bins per histogram level
constexpr int kResCopies = 8;                                 // bank-striped copies of a histogram in LDS (copy = lane & 7)
constexpr int kResList = 4096;                                // keys of a picked bin that are listed and ranked
constexpr int kResMaxIter = 10;                               // histogram / collect rounds per stage (2 on ordinary tiles)
constexpr int kResMaxGroup = 16;                              // workgroups per tile at most (1024 x 1024)
constexpr int kResHead = 64;                                  // header words of an exchange record (per-wave fills and counts)
constexpr int kResXchgWords = kResHead + 2 * kResBins * kResCopies;      // one workgroup's record of an exchange: header + two histograms / the waves' list segments
constexpr uint32_t kResNone = 0u, kResHist = 1u, kResCollect = 2u;
constexpr uint32_t kResErrSpin = 1u, kResErrCount = 2u, kResErrIter = 4u;

struct ResGeom {
    int64_t n_tiles, pixels;
    int items, chunk;                 // stats_item's work items per tile and their size
    int group;                        // G: workgroups per tile
    int tiles_per_round, rounds;
    int xcd_map;                      // 1: the workgroups of a tile have equal blockIdx % 8 (one XCD under round-robin dealing; speed only)
    int unit;                         // normalize_to_0_1 fused
    uint32_t spin_limit;
};

struct alignas(64) ResSync {
    uint32_t arrive;                  // arrivals at the tile's exchanges, counted up through the call
    uint32_t leave;                   // workgroups that are done with the tile's exchanges (the last one resets both words)
    uint32_t pad[14];
};

struct ResWork {
    GroupState* state;
    ResSync* sync;                    // [n_tiles]
    double* partial;                  // [n_tiles][items][kPartial]
    double* partial_all;              // the same for the all-pixel sums (tiles with fewer than three kept pixels)
    uint32_t* head;                   // [n_tiles][G][16]: per workgroup {not 8-bit levels, log2-level range as six float keys}
    uint32_t* xchg;                   // [n_tiles][G][2][kResXchgWords]
};

static size_t resident_bytes(int64_t n_tiles, int64_t pixels) {
    const size_t n = (size_t)n_tiles, items = (size_t)((pixels + kChunk - 1) / kChunk), group = (items + kResQuads - 1) / kResQuads;
    size_t total = align_up(sizeof(GroupState) * n, 256);
    total += align_up(sizeof(ResSync) * n, 256);
    total += 2 * align_up(sizeof(double) * kPartial * items * n, 256);
    total += align_up(sizeof(uint32_t) * 16 * group * n, 256);
    if (group > 1) total += align_up(sizeof(uint32_t) * 2 * kResXchgWords * group * n, 256);
    return total;
}

static ResWork carve_resident(void* base, int64_t n_tiles, int64_t pixels) {
    ResWork w;
    char* p = static_cast<char*>(base);
    const size_t n = (size_t)n_tiles, items = (size_t)((pixels + kChunk - 1) / kChunk), group = (items + kResQuads - 1) / kResQuads;
    w.state = reinterpret_cast<GroupState*>(p);      // (where every form keeps it: sx_macenko_tile_params reads it)
    p += align_up(sizeof(GroupState) * n, 256);
    w.sync = reinterpret_cast<ResSync*>(p);
    p += align_up(sizeof(ResSync) * n, 256);
    w.partial = reinterpret_cast<double*>(p);
    p += align_up(sizeof(double) * kPartial * items * n, 256);
    w.partial_all = reinterpret_cast<double*>(p);
    p += align_up(sizeof(double) * kPartial * items * n, 256);
    w.head = reinterpret_cast<uint32_t*>(p);
    p += align_up(sizeof(uint32_t) * 16 * group * n, 256);
    w.xchg = reinterpret_cast<uint32_t*>(p);
    return w;
}

// ---- LDS of a workgroup --------------------------------------------------------------------------------------------------
struct alignas(16) ResScratch {
    float l2tab[256 * kResCopies];                     // log2(255 x + 1) of code k, the bits log2_level<T> gives the element: entry k of copy c at [8 k + c].  A lane reads copy
                                                       // (lane & 7): lanes on different copies never share a bank, where ONE 256-entry table (bank = k mod 32) made the 32 lanes
                                                       // of a read collide 2.2-fold on average (SQ_LDS_BANK_CONFLICT: 53 % of the LDS pipe's cycles)
    uint32_t canon[256];                               // float types: the bits of (float)T(k / 255)
    uint32_t hist[2][kResBins * kResCopies];           // histograms of a sweep; afterwards [slot][kResList]: the tile's gathered lists; phase R: store staging
    uint32_t part[2][kResBins];                        // this workgroup's histograms with the copies added up, then the tile's totals
    uint32_t list[2][kResList];                        // keys this workgroup collected
    double red[kResHalves * kResThreads / kWave][kPartial];
    double item_part[kResQuads][kPartial];
    double mom[kMoments];
    alignas(16) uint32_t radix[256];
    float vecs[6], he[6], pinv[6];
    float conc_inv_w[2], conc_w[2], conc_org[2];       // level-0 bins of the concentrations: floor(c / w) - org, w a power of two
    uint32_t list_n[2], below[2], xhead[kResHead];
    uint32_t xfill[2][kResMaxGroup * (kResThreads / kWave)], xoff[2][kResMaxGroup * (kResThreads / kWave)];      // the tile's list segments: fills and where each goes
    alignas(16) uint32_t radix2[2][256];
    uint32_t radix_digit2[2], radix_rank2[2];
    uint32_t klo[2], khi[2], rank[2], n_in[2], done[2], answer[2], action[2], shift[2], n_listed[2];
    uint32_t range_key[6];                             // the tile's log2 levels: min (0..2) and max (3..5) per channel, as float keys
    uint32_t bad, err, radix_digit, radix_rank;
    int use_all;
    unsigned long long n_sel;
};

// The pixels a thread holds: channel c of local pixel j is byte (j & 3) of w[c][j >> 2].  Local pixel j = s V + i is element i of the
// thread's s-th pack of its work item -- the pixels stats_item gives thread (threadIdx & 255) of the item's workgroup.
struct ResPixels {
    uint32_t w[3][kResPx / 4];
};

// ---- exchanges between the workgroups of one tile ----------------------------------------------------------------------------
// Every word another workgroup reads is written with st_agent (sc1, write-through) and read with ld_agent (sc1, past the L1).
__device__ __forceinline__ void res_tile_sync(ResSync* sy, int group, uint32_t& gen, uint32_t spin_limit, ResScratch* sh) {
    if (group == 1) {
        __syncthreads();
        return;
    }
    drain_stores();                      // every storing wave: its write-through stores have left
    __syncthreads();
    ++gen;
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(&sy->arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t want = gen * (uint32_t)group;
        uint32_t spins = 0;
        while (ld_agent(&sy->arrive) < want) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > spin_limit) {      // (all workgroups of the launch are resident: never observed.  The call's output is then wrong and says so.)
                sh->err |= kResErrSpin;
                break;
            }
        }
    }
    __syncthreads();
}

// A workgroup is done with the tile's exchanges; the last one leaves the two words zero for the next call (ready state).
__device__ __forceinline__ void res_tile_leave(ResSync* sy, int group) {
    if (group == 1 || threadIdx.x != 0) return;
    const uint32_t before = __hip_atomic_fetch_add(&sy->leave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (before == (uint32_t)group - 1u) {
        st_agent(&sy->arrive, 0u);
        st_agent(&sy->leave, 0u);
    }
}

// ---- phase L: read the tile, keep its codes, accumulate the kept pixels' moments as stats_item does ----------------------------
template <typename T> __device__ __forceinline__ void res_fill_tables(ResScratch* sh) {
    for (int k = threadIdx.x; k < 256; k += blockDim.x) {
        if constexpr (sizeof(T) == 1) {
            const float l = log2_level<uint8_t>((float)k);
#pragma unroll
            for (int c = 0; c < kResCopies; ++c) sh->l2tab[k * kResCopies + c] = l;
            sh->canon[k] = (uint32_t)k;
        } else {
            const float u = div255_of_level((float)k);                      // float(k) / 255, the IEEE quotient (common.hpp)
            const float v = Elem<T>::load(Elem<T>::store(u));               // ... as the element type holds it (.to(dtype): round to nearest even)
            const float l = log2_level<T>(v);
#pragma unroll
            for (int c = 0; c < kResCopies; ++c) sh->l2tab[k * kResCopies + c] = l;
            sh->canon[k] = __float_as_uint(v);
        }
    }
}

template <typename T, int V>
__device__ __forceinline__ void res_load_phase(const T* __restrict__ img, int64_t pixels, int64_t p_begin, int64_t p_end, ResPixels& px, int& n_px, ResScratch* sh,
                                               double (&acc)[kPartial], uint32_t& bad, float (&l2min)[3], float (&l2max)[3]) {
    constexpr int S = kResPx / V;
    constexpr int kShortRun = 32 / V > 0 ? 32 / V : 1;      // packs per fp32 run -- stats_item's grouping, bit for bit
    constexpr int B = 2;                                    // packs requested together (and as many ahead)
    static_assert(S % B == 0 && S == 2 * kShortRun, "pack geometry: two fp32 runs per work item");
    const int tq = threadIdx.x & (kStreamThreads - 1);
    // (no fp64 accumulators in the loop: a work item is two fp32 runs per thread, and 0.0 + (double)run1 + (double)run2 -- stats_item's
    // additions -- is formed once at the end.  Twenty registers less while the tile's codes, the packs in flight and the sums are live.)
    float m[kPartial], first[kPartial];
#pragma unroll
    for (int k = 0; k < kPartial; ++k) m[k] = first[k] = 0.0f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        l2min[c] = __builtin_huge_valf();
        l2max[c] = -__builtin_huge_valf();
#pragma unroll
        for (int i = 0; i < kResPx / 4; ++i) px.w[c][i] = 0u;
    }
    bad = 0u;
    n_px = 0;
    // The packs of step s0 + B are requested before those of step s0 are worked on (eight waves per CU: a wave has to cover its own
    // memory latency).
    PixelPacks<T, V, false> ahead[B];
    bool ahead_live[B];
    auto request = [&](int s0, PixelPacks<T, V, false> (&pk)[B], bool (&live)[B]) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int64_t p = p_begin + (int64_t)(s0 + b) * (kStreamThreads * V) + (int64_t)tq * V;
            live[b] = p < p_end;
            pk[b].clear();
            if (live[b]) pk[b].load(img, pixels, p);
        }
    };
    request(0, ahead, ahead_live);
#pragma unroll
    for (int s0 = 0; s0 < S; s0 += B) {
        PixelPacks<T, V, false> pk[B];
        bool live[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            pk[b] = ahead[b];
            live[b] = ahead_live[b];
        }
        if (s0 + B < S) request(s0 + B, ahead, ahead_live);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int s = s0 + b;
            if (live[b]) {      // (a pack at a time; inside, straight-line code: the pixels of a group of four overlap)
                n_px += V;
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const int j = s * V + i;
                    float od[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float raw = pk[b].value(c, i);
                        float l2;
                        if constexpr (sizeof(T) == 1) {
                            l2 = sh->l2tab[__float_as_uint(raw) * kResCopies + (lane_id() & (kResCopies - 1))];
                        } else {
                            l2 = log2_level<T>(raw);
                            const uint32_t code = (uint32_t)fminf(fmaxf(fmaf(raw, 255.0f, 0.5f), 0.0f), 255.0f);      // (NaN -> 0)
                            bad |= __float_as_uint(raw) ^ sh->canon[code];
                            px.w[c][j >> 2] |= code << (8 * (j & 3));
                        }
                        od[c] = fmaf(-kLn2, l2, kLnIo);      // = optical_density<T>(raw)
                        l2min[c] = fminf(l2min[c], l2);
                        l2max[c] = fmaxf(l2max[c], l2);
                    }
                    // (stats_item's form: products with a 0 / 1 factor instead of a branch around the sums -- the same bits)
                    const float keep = od_selected(od, false) ? 1.0f : 0.0f;
                    const float k0 = keep * od[0], k1 = keep * od[1], k2 = keep * od[2];
                    m[0] += keep;
                    m[1] += k0;
                    m[2] += k1;
                    m[3] += k2;
                    m[4] = fmaf(k0, od[0], m[4]);
                    m[5] = fmaf(k0, od[1], m[5]);
                    m[6] = fmaf(k0, od[2], m[6]);
                    m[7] = fmaf(k1, od[1], m[7]);
                    m[8] = fmaf(k1, od[2], m[8]);
                    m[9] = fmaf(k2, od[2], m[9]);
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // (four pixels' worth of logarithms / table reads in flight, not sixteen)
                }
                if constexpr (sizeof(T) == 1) {      // the codes are the bytes as they arrived
#pragma unroll
                    for (int c = 0; c < 3; ++c)
#pragma unroll
                        for (int k = 0; k < V / 4; ++k) px.w[c][s * (V / 4) + k] = pk[b].w[c][k];
                }
            }
            if (s + 1 == kShortRun) {
#pragma unroll
                for (int k = 0; k < kPartial; ++k) {
                    first[k] = m[k];
                    m[k] = 0.0f;
                }
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kPartial; ++k) {
        acc[k] = 0.0;
        acc[k] += (double)first[k];
        acc[k] += (double)m[k];
    }
}

// ---- the pixels of a thread, four at a time (one register word per channel): fn(l2[4][3]) with l2[e][c] = log2(255 x_c + 1) ------
// The body of fn is straight-line code for four independent pixels: their twelve table reads go out together and their dependent
// chains interleave.  (One pixel per basic block -- a branch around every pixel's work -- left a wave waiting for one LDS round trip
// and one ~30-deep dependent chain per pixel: 20 us per sweep on an idle chip where the instruction count asks for 8.)
template <class Fn>
__device__ __forceinline__ void res_each_code(const ResPixels& px, int n_px, const float* __restrict__ l2tab, Fn fn) {
    // The table reads of word wi + 1 are issued before word wi is worked on (a two-wave-per-SIMD kernel has nobody else to cover an
    // LDS round trip); words beyond the live prefix hold code 0: their reads are harmless and dropped.
    const float* __restrict__ mine = l2tab + (lane_id() & (kResCopies - 1));      // this lane's copy of the table
    auto fetch = [&](int wi, float (&l2)[4][3]) {
        uint32_t w0 = px.w[0][wi], w1 = px.w[1][wi], w2 = px.w[2][wi];
        // (opaque: otherwise the optimiser sees that the byte extracts and table addresses of a pixel are the same in every sweep,
        // works all 384 of them out once and keeps them alive -- in scratch memory -- instead of these 96 registers)
        asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            l2[e][0] = mine[((w0 >> (8 * e)) & 255u) * kResCopies];
            l2[e][1] = mine[((w1 >> (8 * e)) & 255u) * kResCopies];
            l2[e][2] = mine[((w2 >> (8 * e)) & 255u) * kResCopies];
        }
    };
    float ahead[4][3];
    fetch(0, ahead);
#pragma unroll
    for (int wi = 0; wi < kResPx / 4; ++wi) {
        float l2[4][3];
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int c = 0; c < 3; ++c) l2[e][c] = ahead[e][c];
        if (wi + 1 < kResPx / 4) fetch(wi + 1, ahead);
        __builtin_amdgcn_sched_barrier(0);      // (the next word's reads go out first)
        if (4 * wi < n_px) fn(l2);      // (live pixels are a prefix, whole words of four: packs are 4, 8 or 16 pixels)
        // (without it the scheduler starts the table reads of all sixteen words together: several hundred values live at once, spilled)
        __builtin_amdgcn_sched_barrier(0);
    }
}
// The same pixels re-read from memory (a float tile that is not made of 8-bit levels).
template <typename T, int V, class Fn>
__device__ __forceinline__ void res_each_stream(const T* __restrict__ img, int64_t pixels, int64_t p_begin, int64_t p_end, Fn fn) {
    constexpr int S = kResPx / V, B = V >= 8 ? 2 : 4;
    const int tq = threadIdx.x & (kStreamThreads - 1);
#pragma unroll 1
    for (int s0 = 0; s0 < S; s0 += B) {
        PixelPacks<T, V, false> pk[B];
        bool live[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const int64_t p = p_begin + (int64_t)(s0 + b) * (kStreamThreads * V) + (int64_t)tq * V;
            live[b] = p < p_end;
            pk[b].clear();
            if (live[b]) pk[b].load(img, pixels, p);
        }
#pragma unroll
        for (int b = 0; b < B; ++b)
            if (live[b]) {
#pragma unroll
                for (int i0 = 0; i0 < V; i0 += 4) {
                    float l2[4][3];
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int c = 0; c < 3; ++c) l2[e][c] = log2_level<T>(pk[b].value(c, i0 + e));
                    fn(l2);
                }
            }
    }
}
template <typename T, int V, class Fn>
__device__ __forceinline__ void res_each_word(bool stream, const ResPixels& px0, const ResPixels& px1, int n_px0, int n_px1, const ResScratch* sh, const T* __restrict__ img, int64_t pixels,
                                              int64_t p_begin0, int64_t p_end0, int64_t p_begin1, int64_t p_end1, Fn fn) {
    static_assert(V % 4 == 0 && kResHalves <= 2, "whole words of four pixels; one or two work items per thread");
    // (the two register-resident work items are two NAMED objects everywhere: an index into an array of them that the compiler
    // cannot fold -- one loop it declines to unroll -- would move both to scratch memory)
    if constexpr (sizeof(T) == 1) {
        res_each_code(px0, n_px0, sh->l2tab, fn);
        if constexpr (kResHalves == 2) res_each_code(px1, n_px1, sh->l2tab, fn);
    } else {
        if (!stream) {      // (workgroup-uniform)
            res_each_code(px0, n_px0, sh->l2tab, fn);
            if constexpr (kResHalves == 2) res_each_code(px1, n_px1, sh->l2tab, fn);
        } else {
            res_each_stream<T, V>(img, pixels, p_begin0, p_end0, fn);
            if constexpr (kResHalves == 2) res_each_stream<T, V>(img, pixels, p_begin1, p_end1, fn);
        }
    }
}

// ---- selection --------------------------------------------------------------------------------------------------------------
// Exact element of 0-based rank `rank` among list[0..n) (all inside [klo, khi]): byte-wise radix rounds from the highest byte in
// which klo and khi differ.  Whole workgroup, uniform result; ties need no order.
__device__ __forceinline__ uint32_t res_radix_select(const uint32_t* list, uint32_t n, uint32_t rank, uint32_t klo, uint32_t khi, ResScratch* sh) {
    const uint32_t diff = klo ^ khi;
    if (diff == 0u) return klo;
    const int top = (31 - __clz((int)diff)) / 8 * 8;
    uint32_t mask = top >= 24 ? 0u : ~((1u << (top + 8)) - 1u), prefix = klo & mask;
    __syncthreads();
    if (threadIdx.x == 0) sh->radix_rank = rank;
    for (int shift = top; shift >= 0; shift -= 8) {
        if (threadIdx.x < 256) sh->radix[threadIdx.x] = 0u;
        __syncthreads();
        for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
            const uint32_t k = list[i];
            if (((k ^ prefix) & mask) == 0u) atomicAdd(&sh->radix[(k >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (threadIdx.x < kWave) {
            uint32_t d, rb;
            scan_pick32(sh->radix, sh->radix_rank, d, rb);
            if (lane_id() == 0) {
                sh->radix_digit = d;
                sh->radix_rank = rb;
            }
        }
        __syncthreads();
        prefix |= sh->radix_digit << shift;
        mask |= 0xFFu << shift;
    }
    return prefix;
}

// The same for two lists at once (the two slots of a stage): the rounds -- their barriers -- are shared.
__device__ __forceinline__ void res_radix_select2(const uint32_t* list0, uint32_t n0, uint32_t rank0, uint32_t klo0, uint32_t khi0, const uint32_t* list1, uint32_t n1, uint32_t rank1,
                                                  uint32_t klo1, uint32_t khi1, ResScratch* sh, uint32_t& a0, uint32_t& a1) {
    const uint32_t d0 = klo0 ^ khi0, d1 = klo1 ^ khi1;
    const int top0 = d0 ? (31 - __clz((int)d0)) / 8 * 8 : -8, top1 = d1 ? (31 - __clz((int)d1)) / 8 * 8 : -8;
    uint32_t mask0 = top0 >= 24 ? 0u : (top0 < 0 ? 0xFFFFFFFFu : ~((1u << (top0 + 8)) - 1u)), prefix0 = klo0 & mask0;
    uint32_t mask1 = top1 >= 24 ? 0u : (top1 < 0 ? 0xFFFFFFFFu : ~((1u << (top1 + 8)) - 1u)), prefix1 = klo1 & mask1;
    __syncthreads();
    if (threadIdx.x < 2) sh->radix_rank2[threadIdx.x] = threadIdx.x ? rank1 : rank0;
    for (int shift = max(top0, top1); shift >= 0; shift -= 8) {
        const bool on0 = shift <= top0, on1 = shift <= top1;      // uniform
        for (int i = threadIdx.x; i < 512; i += blockDim.x) (&sh->radix2[0][0])[i] = 0u;
        __syncthreads();
        if (on0)
            for (uint32_t i = threadIdx.x; i < n0; i += blockDim.x) {
                const uint32_t k = list0[i];
                if (((k ^ prefix0) & mask0) == 0u) atomicAdd(&sh->radix2[0][(k >> shift) & 255u], 1u);
            }
        if (on1)
            for (uint32_t i = threadIdx.x; i < n1; i += blockDim.x) {
                const uint32_t k = list1[i];
                if (((k ^ prefix1) & mask1) == 0u) atomicAdd(&sh->radix2[1][(k >> shift) & 255u], 1u);
            }
        __syncthreads();
        if (threadIdx.x < 2 * kWave) {
            const int sl = threadIdx.x / kWave;
            if (sl ? on1 : on0) {
                uint32_t d, rb;
                scan_pick32(sh->radix2[sl], sh->radix_rank2[sl], d, rb);
                if (lane_id() == 0) {
                    sh->radix_digit2[sl] = d;
                    sh->radix_rank2[sl] = rb;
                }
            }
        }
        __syncthreads();
        if (on0) {
            prefix0 |= sh->radix_digit2[0] << shift;
            mask0 |= 0xFFu << shift;
        }
        if (on1) {
            prefix1 |= sh->radix_digit2[1] << shift;
            mask1 |= 0xFFu << shift;
        }
    }
    a0 = prefix0;
    a1 = prefix1;
}

// Everything a stage's workgroups share about the tile, passed around as one bundle.
template <typename T> struct ResTile {
    const T* img;
    int64_t pixels, p_begin0, p_end0, p_begin1, p_end1;
    int tile, r, group;
    bool stream;
    ResSync* sy;
    uint32_t* xchg;      // the tile's exchange records: [group][2][kResXchgWords]
    uint32_t spin_limit;
};

// One wave: bin, rank inside it and count for a kBins-bin histogram (kBins / 64 consecutive bins per lane, then the owner lane's bins
// one per lane).
template <int kBins>
__device__ __forceinline__ bool res_pick(const uint32_t* hist, uint32_t rank, uint32_t& bin, uint32_t& rank_in, uint32_t& count) {
    constexpr int kPer = kBins / kWave;
    static_assert(kPer >= 4 && kPer <= kWave && kPer % 4 == 0, "second level: one bin per lane");
    const int lane = (int)lane_id();
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < kPer; i += 4) {
        const uint4 h = *reinterpret_cast<const uint4*>(hist + kPer * lane + i);
        mine += h.x + h.y + h.z + h.w;
    }
    const uint32_t incl = wave_scan_u32(mine);
    const uint64_t over = __ballot(incl > rank);
    const int owner = over ? (__ffsll((long long)over) - 1) : (kWave - 1);
    const uint32_t r = rank - (uint32_t)__builtin_amdgcn_readlane((int)(incl - mine), owner);
    const uint32_t h = lane < kPer ? hist[kPer * owner + lane] : 0u;
    const uint32_t incl2 = wave_scan_u32(h);
    const uint64_t over2 = __ballot(lane < kPer && incl2 > r);
    const int sub = over2 ? (__ffsll((long long)over2) - 1) : (kPer - 1);
    bin = (uint32_t)(kPer * owner + sub);
    rank_in = r - (uint32_t)__builtin_amdgcn_readlane((int)(incl2 - h), sub);
    count = (uint32_t)__builtin_amdgcn_readlane((int)h, sub);
    return over != 0 && over2 != 0;
}
template <int kBins> __device__ __forceinline__ uint32_t res_hist_total(const uint32_t* hist) {      // one wave, uniform result
    constexpr int kPer = kBins / kWave;
    uint32_t mine = 0;
#pragma unroll
    for (int i = 0; i < kPer; i += 4) {
        const uint4 h = *reinterpret_cast<const uint4*>(hist + kPer * (int)lane_id() + i);
        mine += h.x + h.y + h.z + h.w;
    }
    return wave_total_u32(mine);
}

constexpr int kResFine = 4096;                 // bins of the SAMPLE's histogram (four per level-0 bin)
constexpr int kResGather = kResBins * kResCopies;      // keys of a slot the tile's workgroups may list together (the room of one histogram): 8192
constexpr float kResSigmas = 10.0f;            // half-width of a bracket in standard deviations of an independent sample's rank (the sample is 1/16 of the
                                               // pixels in row pairs: neighbours are correlated -- twelve are six at a quarter of the nominal sample size)

// One stage: the two order statistics of the tile's keys.  kConc == false: both are taken among the angle keys of the selected
// pixels (ranks want[0], want[1]); kConc == true: slot j among concentration j of every pixel.  coef: plane vectors / pseudo-inverse.
//
// Fast path: (1) a SAMPLE of the tile's keys -- one register word of four pixels per work item and thread, 1/16 of the pixels -- goes into a
// histogram of 4096 bins whose edges are exact in the keys; (2) the bins that hold the sample ranks 12 sigma either side of the wanted
// quantile bracket the answer; (3) ONE sweep over all pixels counts the keys below the bracket and lists the keys inside it (~2 %);
// (4) the wanted element is number (rank - below) of the list -- IF that index lies inside the list: the counts are exact integers, so the
// check is a proof, and a bracket that missed (or a list that overflowed) sends the stage to the general path: level-0 histogram of
// every key, picked bin, finer bins or list, until one key is left -- two sweeps more, never a wrong bit.
template <typename T, int V, bool kConc>
__device__ __forceinline__ void res_stage(const ResTile<T>& t, const ResPixels& px0, const ResPixels& px1, int n_px0, int n_px1, ResScratch* sh, uint32_t& gen, const float (&coef)[6], bool use_all,
                                          uint32_t want0, uint32_t want1, uint32_t total0, uint32_t total1, GroupState& st, int stamp0) {
    (void)st;
    (void)stamp0;
    const uint32_t copy = lane_id() & (kResCopies - 1);
    float inv_w[2] = {0.0f, 0.0f}, org[2] = {0.0f, 0.0f};
    if constexpr (kConc) {
        inv_w[0] = sh->conc_inv_w[0];
        inv_w[1] = sh->conc_inv_w[1];
        org[0] = sh->conc_org[0];
        org[1] = sh->conc_org[1];
    }
    // the keys of a pixel and their level-0 bins as floats (angle stage: one key for both slots, selected pixels only)
    auto keys_of = [&](const float (&l2)[3], uint32_t& k0, uint32_t& k1, float& f0, float& f1) -> bool {
        const float od[3] = {fmaf(-kLn2, l2[0], kLnIo), fmaf(-kLn2, l2[1], kLnIo), fmaf(-kLn2, l2[2], kLnIo)};
        if constexpr (!kConc) {
            const float t0 = fmaf(od[2], coef[4], fmaf(od[1], coef[2], od[0] * coef[0]));
            const float t1 = fmaf(od[2], coef[5], fmaf(od[1], coef[3], od[0] * coef[1]));
            const float d = diamond_angle(t1, t0);      // (angle_key's arithmetic)
            k0 = k1 = float_key(d);
            f0 = f1 = d * 512.0f;      // exact: level-0 bin floor(f) holds exactly the keys of [b / 512, (b + 1) / 512), fine bin floor(4 f) those of [b / 2048, (b + 1) / 2048)
            // (od_selected without its short-circuit: no control flow per pixel)
            return (bool)((int)(fminf(od[0], fminf(od[1], od[2])) >= kBeta) | (int)use_all);
        } else {
            float c0, c1;
            concentration(od, coef, c0, c1);
            c0 += 0.0f;      // (-0 -> +0: the bins are told by value, the keys by bits)
            c1 += 0.0f;
            k0 = float_key(c0);
            k1 = float_key(c1);
            f0 = c0 * inv_w[0];      // (exact -- a power of two: floor(f) - org is the level-0 bin, floor(4 f) - 4 org the fine one; the floor comes BEFORE the
            f1 = c1 * inv_w[1];      // subtraction, which is then one of integers: `f - org` may round, and a key next to an edge would change its bin)
            return true;
        }
    };
    auto bin_of_f = [&](float f, int s, int bins, bool fine) {      // f as keys_of returns it; s: the slot (its origin)
        const float o = kConc ? (fine ? 4.0f * org[s] : org[s]) : 0.0f;
        return (uint32_t)fminf(fmaxf(floorf(fine ? 4.0f * f : f) - o, 0.0f), (float)(bins - 1));
    };
    // the smallest key of level-0 bin b (fine: quarter bins) -- exact, see keys_of
    auto edge_key = [&](int s, float b, bool fine) -> uint32_t {
        float v;
        if constexpr (!kConc) v = b * (fine ? 1.0f / 2048.0f : 1.0f / 512.0f);
        else v = (sh->conc_org[s] + (fine ? 0.25f * b : b)) * sh->conc_w[s];
        return float_key(v + 0.0f);
    };
    uint32_t* const hist_flat = &sh->hist[0][0];

    // publish this workgroup's record of an exchange, meet the others; `parity` names the records to read afterwards
    auto publish_header = [&](uint32_t* mine) {
        if (threadIdx.x < 4) st_agent(&mine[threadIdx.x], threadIdx.x < 2 ? sh->list_n[threadIdx.x] : sh->below[threadIdx.x - 2]);
    };
    auto record_of = [&](int q, uint32_t parity) -> const uint32_t* { return &t.xchg[((size_t)q * 2 + parity) * kResXchgWords]; };
    // the tile's lists of a slot, one after the other, into dst (at most kResGather keys); returns how many there are
    auto gather_list = [&](int s, uint32_t parity, uint32_t* dst, uint32_t own_n, bool& overflow) -> uint32_t {
        overflow = false;
        if (t.group == 1) {
            for (uint32_t i = threadIdx.x; i < own_n; i += blockDim.x) dst[i] = sh->list[s][i];
            return own_n;
        }
        uint32_t at = 0;
        for (int q = 0; q < t.group; ++q) {
            const uint32_t* rec = record_of(q, parity);
            const uint32_t n_raw = ld_agent(&rec[s]), n = min(n_raw, (uint32_t)kResList);      // (0xFFFFFFFF: that workgroup's list overflowed)
            if (n_raw > (uint32_t)kResList) overflow = true;
            for (uint32_t i = threadIdx.x; i < n && at + i < (uint32_t)kResGather; i += blockDim.x) dst[at + i] = ld_agent(&rec[kResHead + s * kResList + i]);
            at += n;
        }
        if (at > (uint32_t)kResGather) overflow = true;
        return at;
    };

    // ------------------------------------------------------------------------------------------------ fast path
    bool fast_done = false;
    {
        constexpr int kWaves = kResThreads / kWave;
        constexpr int kSeg = 2 * kResGather / (2 * kWaves);      // keys a wave may list per slot in the sweep (the two histograms' room, split evenly): 1024
        const int wave = threadIdx.x / kWave;
        for (int i = threadIdx.x; i < 2 * kResFine; i += blockDim.x) hist_flat[i] = 0u;
        if (threadIdx.x < 2) {
            sh->list_n[threadIdx.x] = 0u;
            sh->below[threadIdx.x] = 0u;
        }
        __syncthreads();
        // (1) the sample: ONE pixel of every register word -- every fourth column of every row of the tile, a regular grid (a sample of whole
        // rows is as good as its number of rows: neighbours along a row are all but copies of each other)
        {
            auto sample_pixel = [&](const float (&l2)[3]) {
                uint32_t k0, k1;
                float f0, f1;
                const bool valid = keys_of(l2, k0, k1, f0, f1);
                if (valid) {
                    atomicAdd(&hist_flat[bin_of_f(f0, 0, kResFine, true)], 1u);
                    if constexpr (kConc) atomicAdd(&hist_flat[kResFine + bin_of_f(f1, 1, kResFine, true)], 1u);
                }
            };
            bool stream = false;
            if constexpr (sizeof(T) != 1) stream = t.stream;
            if (!stream) {
                const float* __restrict__ mine = sh->l2tab + copy;
                auto half = [&](const ResPixels& px, int n_px) {
#pragma unroll
                    for (int wi = 0; wi < kResPx / 4; ++wi) {
                        if (4 * wi < n_px) {
                            uint32_t w0 = px.w[0][wi], w1 = px.w[1][wi], w2 = px.w[2][wi];
                            asm volatile("" : "+v"(w0), "+v"(w1), "+v"(w2));
                            constexpr int kShift[4] = {0, 8, 16, 24};
                            const int sft = kShift[wi & 3];
                            const float l2[3] = {mine[((w0 >> sft) & 255u) * kResCopies], mine[((w1 >> sft) & 255u) * kResCopies], mine[((w2 >> sft) & 255u) * kResCopies]};
                            sample_pixel(l2);
                        }
                        if ((wi & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                    }
                };
                half(px0, n_px0);
                if constexpr (kResHalves == 2) half(px1, n_px1);
            } else {
                if constexpr (sizeof(T) != 1) {      // the same pixels from memory
                    auto half = [&](int64_t p_begin, int64_t p_end) {
                        constexpr int S = kResPx / V;
#pragma unroll 1
                        for (int s0 = 0; s0 < S; ++s0) {
                            const int64_t p = p_begin + (int64_t)s0 * (kStreamThreads * V) + (int64_t)(threadIdx.x & (kStreamThreads - 1)) * V;
                            if (p < p_end) {
                                PixelPacks<T, V, false> pk;
                                pk.load(t.img, t.pixels, p);
#pragma unroll
                                for (int q = 0; q < V / 4; ++q) {
                                    const int wi = s0 * (V / 4) + q;      // the register word these four pixels would be
                                    float l2[3];
#pragma unroll
                                    for (int e = 0; e < 4; ++e)
                                        if ((wi & 3) == e) {
#pragma unroll
                                            for (int c = 0; c < 3; ++c) l2[c] = log2_level<T>(pk.value(c, 4 * q + e));
                                        }
                                    sample_pixel(l2);
                                }
                            }
                        }
                    };
                    half(t.p_begin0, t.p_end0);
                    if constexpr (kResHalves == 2) half(t.p_begin1, t.p_end1);
                }
            }
        }
        __syncthreads();
        // (2) the tile's sample histogram(s) -- 16-byte write-through stores and sc1 loads --, then the brackets
        constexpr int kHists = kConc ? 2 : 1;
        if (t.group > 1) {
            uint32_t* mine = const_cast<uint32_t*>(record_of(t.r, gen & 1u));
            const auto rs_mine = __builtin_amdgcn_make_buffer_rsrc(mine, 0, (int)(kResXchgWords * 4), 0x00020000);
            for (int i = threadIdx.x; i < kHists * kResFine / 4; i += blockDim.x) {
                sx_u4 v;
                v[0] = hist_flat[4 * i];
                v[1] = hist_flat[4 * i + 1];
                v[2] = hist_flat[4 * i + 2];
                v[3] = hist_flat[4 * i + 3];
                __builtin_amdgcn_raw_buffer_store_b128(v, rs_mine, (int)(kResHead * 4 + 16 * i), 0, 16);
            }
            const uint32_t parity = gen & 1u;
            res_tile_sync(t.sy, t.group, gen, t.spin_limit, sh);
            for (int i = threadIdx.x; i < kHists * kResFine / 4; i += blockDim.x) {
                sx_u4 sum = {0u, 0u, 0u, 0u};
                for (int q = 0; q < t.group; ++q) {
                    const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(record_of(q, parity)), 0, (int)(kResXchgWords * 4), 0x00020000);
                    const sx_u4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(kResHead * 4 + 16 * i), 0, 16);
                    sum += v;
                }
                hist_flat[4 * i] = sum[0];
                hist_flat[4 * i + 1] = sum[1];
                hist_flat[4 * i + 2] = sum[2];
                hist_flat[4 * i + 3] = sum[3];
            }
            __syncthreads();
        }
        if (threadIdx.x < 2 * kWave) {      // wave s: slot s
            const int s = threadIdx.x / kWave;
            const uint32_t* h = hist_flat + (kConc ? s : 0) * kResFine;
            const uint32_t m = res_hist_total<kResFine>(h);
            const uint32_t want = s ? want1 : want0, total = s ? total1 : total0;
            // sample ranks either side of the rank the wanted quantile has in a sample of m: 10 sigma of an independent sample's rank, and
            // 0.3 % of the sample for what a grid sample of a smooth image is not independent in (+ 3)
            const float f = total > 1u ? (float)want * __builtin_amdgcn_rcpf((float)(total - 1u)) : 0.0f;
            const float r = f * (float)((int)m - 1);
            const float half_width = kResSigmas * __builtin_amdgcn_sqrtf(fmaxf((float)m * f * (1.0f - f), 0.0f)) + 0.003f * (float)m + 3.0f;
            const float lo_f = floorf(r - half_width), hi_f = ceilf(r + half_width);
            uint32_t b_lo = 0, b_hi = kResFine - 1, ri, cnt;
            // (hardly any sample -- a tile with little tissue: everything is inside the bracket; the list then holds the few selected pixels)
            const bool tiny = m < 64u;
            bool ok = true;
            if (!tiny && lo_f >= 0.0f) ok = res_pick<kResFine>(h, (uint32_t)lo_f, b_lo, ri, cnt);
            if (!tiny && ok && hi_f <= (float)((int)m - 1)) ok = res_pick<kResFine>(h, (uint32_t)hi_f, b_hi, ri, cnt);
            if (lane_id() == 0) {
                const bool open_lo = tiny || !(lo_f >= 0.0f) || b_lo == 0u, open_hi = tiny || !(hi_f <= (float)((int)m - 1)) || b_hi == (uint32_t)(kResFine - 1);
                sh->klo[s] = open_lo ? 0u : edge_key(s, (float)b_lo, true);
                sh->khi[s] = open_hi ? 0xFFFFFFFFu : edge_key(s, (float)(b_hi + 1u), true) - 1u;
                sh->done[s] = ok ? 0u : 2u;      // 2: no bracket -- the general path
                sh->rank[s] = want;
                sh->n_in[s] = total;
                sh->answer[s] = 0u;
                sh->n_listed[s] = 0u;
            }
        }
        __syncthreads();
        if (t.r == 0) SX_STAMP(st, stamp0);
        if (sh->done[0] == 0u && sh->done[1] == 0u) {      // uniform
            // (3) one sweep: count what lies below each bracket (per lane), list what lies inside -- every wave into its own segment of the
            // histograms' room, its fill in scalar registers: no atomic, no reservation, nothing another wave waits for
            const uint32_t lo0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->klo[0]), hi0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->khi[0]);
            const uint32_t lo1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->klo[1]), hi1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->khi[1]);
            uint32_t* const seg0 = hist_flat + (size_t)wave * kSeg;
            uint32_t* const seg1 = hist_flat + (size_t)(kWaves + wave) * kSeg;
            uint32_t below0 = 0, below1 = 0;      // per lane
            uint32_t fill0 = 0, fill1 = 0;        // per wave
            res_each_word<T, V>(t.stream, px0, px1, n_px0, n_px1, sh, t.img, t.pixels, t.p_begin0, t.p_end0, t.p_begin1, t.p_end1, [&](const float (&l2)[4][3]) {
                uint32_t k0[4], k1[4];
                uint64_t in0[4], in1[4];
                uint64_t any = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float f0, f1;
                    const bool valid = keys_of(l2[e], k0[e], k1[e], f0, f1);
                    asm volatile("" : "+v"(k0[e]), "+v"(k1[e]));      // (pinned: the optimiser otherwise sinks the key computation under `valid`, a branch per pixel)
                    below0 += (uint32_t)((int)valid & (int)(k0[e] < lo0));
                    below1 += (uint32_t)((int)valid & (int)(k1[e] < lo1));
                    in0[e] = __builtin_amdgcn_ballot_w64((bool)((int)valid & (int)(k0[e] >= lo0) & (int)(k0[e] <= hi0)));
                    in1[e] = __builtin_amdgcn_ballot_w64((bool)((int)valid & (int)(k1[e] >= lo1) & (int)(k1[e] <= hi1)));
                    any |= in0[e] | in1[e];
                }
                if (any) {      // wave-uniform
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        if (in0[e]) {
                            const uint32_t at = fill0 + rank_in_mask(in0[e]);
                            if (((in0[e] >> lane_id()) & 1ull) && at < (uint32_t)kSeg) seg0[at] = k0[e];
                            fill0 += (uint32_t)__popcll(in0[e]);
                        }
                        if (in1[e]) {
                            const uint32_t at = fill1 + rank_in_mask(in1[e]);
                            if (((in1[e] >> lane_id()) & 1ull) && at < (uint32_t)kSeg) seg1[at] = k1[e];
                            fill1 += (uint32_t)__popcll(in1[e]);
                        }
                    }
                }
            });
            // (4) every wave publishes its two segments as they lie (16-byte write-through stores) and its counts; the tile's waves'
            // segments are then put one after the other in LDS; the proof; the select
            below0 = wave_total_u32(below0);
            below1 = wave_total_u32(below1);
            const int n_seg = t.group * kWaves;      // list segments of the tile per slot
            uint32_t parity = 0;
            if (t.group > 1) {
                uint32_t* mine = const_cast<uint32_t*>(record_of(t.r, gen & 1u));
                const auto rs_mine = __builtin_amdgcn_make_buffer_rsrc(mine, 0, (int)(kResXchgWords * 4), 0x00020000);
#pragma unroll
                for (int sl = 0; sl < 2; ++sl) {
                    const uint32_t fill = min(sl ? fill1 : fill0, (uint32_t)kSeg);
                    const uint32_t* src = hist_flat + (size_t)(sl * kWaves + wave) * kSeg;
                    for (uint32_t i = 4u * lane_id(); i < fill; i += 4u * kWave) {      // (whole 16-byte pieces: what lies beyond the fill is never read)
                        sx_u4 v;
                        v[0] = src[i];
                        v[1] = src[i + 1];
                        v[2] = src[i + 2];
                        v[3] = src[i + 3];
                        __builtin_amdgcn_raw_buffer_store_b128(v, rs_mine, (int)((kResHead + (sl * kWaves + wave) * kSeg + i) * 4), 0, 16);
                    }
                }
                if (lane_id() < 4) {
                    const uint32_t v = lane_id() == 0 ? fill0 : lane_id() == 1 ? fill1 : lane_id() == 2 ? below0 : below1;
                    st_agent(&mine[lane_id() * kWaves + wave], v);
                }
                parity = gen & 1u;
                res_tile_sync(t.sy, t.group, gen, t.spin_limit, sh);
            } else {
                if (lane_id() < 4) {
                    const uint32_t v = lane_id() == 0 ? fill0 : lane_id() == 1 ? fill1 : lane_id() == 2 ? below0 : below1;
                    sh->xhead[lane_id() * kWaves + wave] = v;
                }
                __syncthreads();
            }
            if (t.r == 0) SX_STAMP(st, stamp0 + 1);
            // the fills and counts of every segment of the tile
            if (threadIdx.x < 2) sh->below[threadIdx.x] = 0u;
            __syncthreads();
            if ((int)threadIdx.x < n_seg) {
                const int q = threadIdx.x / kWaves, w = threadIdx.x % kWaves;
                uint32_t f0, f1, b0, b1;
                if (t.group > 1) {
                    const uint32_t* rec = record_of(q, parity);
                    f0 = ld_agent(&rec[w]);
                    f1 = ld_agent(&rec[kWaves + w]);
                    b0 = ld_agent(&rec[2 * kWaves + w]);
                    b1 = ld_agent(&rec[3 * kWaves + w]);
                } else {
                    f0 = sh->xhead[w];
                    f1 = sh->xhead[kWaves + w];
                    b0 = sh->xhead[2 * kWaves + w];
                    b1 = sh->xhead[3 * kWaves + w];
                }
                sh->xfill[0][threadIdx.x] = f0;
                sh->xfill[1][threadIdx.x] = f1;
                atomicAdd(&sh->below[0], b0);
                atomicAdd(&sh->below[1], b1);
            }
            __syncthreads();
            if (threadIdx.x < 2) {      // where each segment goes (a segment that overflowed: the general path)
                uint32_t at = 0, over = 0;
                for (int i = 0; i < n_seg; ++i) {
                    const uint32_t f = sh->xfill[threadIdx.x][i];
                    sh->xoff[threadIdx.x][i] = at;
                    if (f > (uint32_t)kSeg) over = 1u;
                    at += min(f, (uint32_t)kSeg);
                }
                sh->list_n[threadIdx.x] = (over || at > (uint32_t)kResGather) ? 0xFFFFFFFFu : at;
            }
            __syncthreads();
            const uint32_t n0 = sh->list_n[0], n1 = sh->list_n[1];
            const uint32_t tile_below0 = sh->below[0], tile_below1 = sh->below[1];
            const bool good0 = n0 != 0xFFFFFFFFu && want0 >= tile_below0 && want0 - tile_below0 < n0, good1 = n1 != 0xFFFFFFFFu && want1 >= tile_below1 && want1 - tile_below1 < n1;
            // where the tile's keys are put together: several workgroups -- the histograms' room (their own segments there are dead once published: they
            // read them back from their record like everybody else's), 8192 keys per slot; one workgroup -- its segments are read in place, the lists go
            // side by side into the list area
            const bool fits = n0 != 0xFFFFFFFFu && n1 != 0xFFFFFFFFu && (t.group > 1 || n0 + n1 <= 2u * (uint32_t)kResList);
            if (good0 && good1 && fits) {      // uniform over the tile: every workgroup sees the same counts
                // the tile's keys of the two slots, one segment after the other, into sh->list (as ONE array of 2 x kResList words: slot 0 from the
                // front, slot 1 behind it); every wave takes every kWaves-th segment, all its loads first
                uint32_t* const all = t.group > 1 ? hist_flat : &sh->list[0][0];
                const uint32_t base1 = t.group > 1 ? (uint32_t)kResGather : n0;
                for (int sl = 0; sl < 2; ++sl) {
                    for (int s0 = wave; s0 < n_seg; s0 += kWaves * 4) {
                        sx_u4 v[4][4];      // four segments in flight, up to four pieces of a lane each (kSeg = 1024: 256 pieces of 16 bytes)
                        uint32_t fill[4], off[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int sg = s0 + u * kWaves;
                            fill[u] = sg < n_seg ? min(sh->xfill[sl][sg], (uint32_t)kSeg) : 0u;
                            off[u] = sg < n_seg ? sh->xoff[sl][sg] : 0u;
                            const int q = sg / kWaves, w = sg % kWaves;
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const uint32_t i = 4u * (lane_id() + (uint32_t)c * kWave);
                                v[u][c] = sx_u4{0u, 0u, 0u, 0u};
                                if (i < fill[u]) {
                                    if (t.group > 1) {
                                        const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(record_of(q, parity)), 0, (int)(kResXchgWords * 4), 0x00020000);
                                        v[u][c] = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((kResHead + (sl * kWaves + w) * kSeg + i) * 4), 0, 16);
                                    } else {
                                        const uint32_t* src = hist_flat + (size_t)(sl * kWaves + w) * kSeg + i;
                                        v[u][c] = sx_u4{src[0], src[1], src[2], src[3]};
                                    }
                                }
                            }
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                const uint32_t i = 4u * (lane_id() + (uint32_t)c * kWave);
#pragma unroll
                                for (int e = 0; e < 4; ++e)
                                    if (i + e < fill[u]) all[(sl ? base1 : 0u) + off[u] + i + e] = v[u][c][e];
                            }
                    }
                }
                __syncthreads();
                uint32_t a0, a1;
                res_radix_select2(all, n0, want0 - tile_below0, lo0, hi0, all + base1, n1, want1 - tile_below1, lo1, hi1, sh, a0, a1);
                if (threadIdx.x == 0) {
                    sh->answer[0] = a0;
                    sh->answer[1] = a1;
                    sh->n_listed[0] = n0;
                    sh->n_listed[1] = n1;
                    sh->done[0] = sh->done[1] = 1u;
                }
                fast_done = true;
            }
            __syncthreads();
            if (t.r == 0) SX_STAMP(st, stamp0 + 2);
        }
    }
    if (fast_done) return;

    // ------------------------------------------------------------------------------------------------ general path
    if (threadIdx.x == 0 && t.r == 0) atomicOr(&st.fell_back, kConc ? 12u : 3u);      // (diagnostic: these slots took the general path)
    if (threadIdx.x < 2) {
        const int s = threadIdx.x;
        sh->klo[s] = 0u;
        sh->khi[s] = 0xFFFFFFFFu;
        sh->rank[s] = s ? want1 : want0;
        sh->n_in[s] = s ? total1 : total0;
        sh->done[s] = 0u;
        sh->answer[s] = 0u;
        sh->action[s] = kResHist;
        sh->shift[s] = 0u;
        sh->n_listed[s] = 0u;
    }
    bool level0 = true;
    for (int iter = 0; iter < kResMaxIter; ++iter) {
        for (int i = threadIdx.x; i < 2 * kResBins * kResCopies; i += blockDim.x) hist_flat[i] = 0u;
        if (threadIdx.x < 2) sh->list_n[threadIdx.x] = 0u;
        __syncthreads();
        // (uniform values out of LDS: said so, they live in scalar registers)
        const uint32_t act0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->action[0]), act1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->action[1]);
        const uint32_t klo0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->klo[0]), khi0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->khi[0]);
        const uint32_t klo1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->klo[1]), khi1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->khi[1]);
        const uint32_t sh0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->shift[0]), sh1 = (uint32_t)__builtin_amdgcn_readfirstlane((int)sh->shift[1]);
        // every key that is still in play: into its bin (level 0: value bins; below: integer bins over the picked bin's key range) or onto
        // the list.  Per pixel, with branches: this path is rare.
        res_each_word<T, V>(t.stream, px0, px1, n_px0, n_px1, sh, t.img, t.pixels, t.p_begin0, t.p_end0, t.p_begin1, t.p_end1, [&](const float (&l2)[4][3]) {
#pragma unroll 1
            for (int e = 0; e < 4; ++e) {
                uint32_t k0, k1;
                float f0, f1;
                if (keys_of(l2[e], k0, k1, f0, f1)) {
                    if (level0) {
                        atomicAdd(&sh->hist[0][bin_of_f(f0, 0, kResBins, false) * kResCopies + copy], 1u);
                        if constexpr (kConc) atomicAdd(&sh->hist[1][bin_of_f(f1, 1, kResBins, false) * kResCopies + copy], 1u);
                    } else {
                        if (act0 != kResNone && k0 >= klo0 && k0 <= khi0) {
                            if (act0 == kResHist) {
                                atomicAdd(&sh->hist[0][((k0 - klo0) >> sh0) * kResCopies + copy], 1u);
                            } else {
                                const uint32_t at = atomicAdd(&sh->list_n[0], 1u);
                                if (at < (uint32_t)kResList) sh->list[0][at] = k0;
                            }
                        }
                        if (act1 != kResNone && k1 >= klo1 && k1 <= khi1) {
                            if (act1 == kResHist) {
                                atomicAdd(&sh->hist[1][((k1 - klo1) >> sh1) * kResCopies + copy], 1u);
                            } else {
                                const uint32_t at = atomicAdd(&sh->list_n[1], 1u);
                                if (at < (uint32_t)kResList) sh->list[1][at] = k1;
                            }
                        }
                    }
                }
            }
        });
        __syncthreads();
        // this workgroup's histograms with the copies added up (the angle stage's level 0 has ONE histogram for both slots)
        const bool shared0 = !kConc && level0;
        const bool hist0 = act0 == kResHist, hist1 = act1 == kResHist && !shared0;
        for (int bin = threadIdx.x; bin < kResBins; bin += blockDim.x) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (h == 0 ? hist0 : hist1) {
                    const uint4 a = *reinterpret_cast<const uint4*>(&sh->hist[h][bin * kResCopies]), b = *reinterpret_cast<const uint4*>(&sh->hist[h][bin * kResCopies + 4]);
                    sh->part[h][bin] = a.x + a.y + a.z + a.w + b.x + b.y + b.z + b.w;
                }
        }
        __syncthreads();
        const uint32_t own_n0 = min(sh->list_n[0], (uint32_t)kResList), own_n1 = min(sh->list_n[1], (uint32_t)kResList);
        uint32_t listed0 = own_n0, listed1 = own_n1;
        uint32_t parity = 0;
        if (t.group > 1) {
            uint32_t* mine = const_cast<uint32_t*>(record_of(t.r, gen & 1u));
            if (threadIdx.x < 2) st_agent(&mine[threadIdx.x], sh->list_n[threadIdx.x]);
            for (int bin = threadIdx.x; bin < kResBins; bin += blockDim.x) {
                if (hist0) st_agent(&mine[kResHead + bin], sh->part[0][bin]);
                if (hist1) st_agent(&mine[kResHead + kResList + bin], sh->part[1][bin]);
            }
            if (act0 == kResCollect)
                for (uint32_t i = threadIdx.x; i < own_n0; i += blockDim.x) st_agent(&mine[kResHead + i], sh->list[0][i]);
            if (act1 == kResCollect)
                for (uint32_t i = threadIdx.x; i < own_n1; i += blockDim.x) st_agent(&mine[kResHead + kResList + i], sh->list[1][i]);
            parity = gen & 1u;
            res_tile_sync(t.sy, t.group, gen, t.spin_limit, sh);
            for (int bin = threadIdx.x; bin < kResBins; bin += blockDim.x) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
                    if (h == 0 ? hist0 : hist1) {
                        uint32_t sum = 0;
                        for (int q = 0; q < t.group; ++q) sum += ld_agent(&record_of(q, parity)[kResHead + h * kResList + bin]);
                        sh->part[h][bin] = sum;
                    }
            }
        }
        bool over0 = false, over1 = false;
        if (act0 == kResCollect) listed0 = gather_list(0, parity, &sh->hist[0][0], own_n0, over0);      // (the histograms have been read: their room holds the tile's lists)
        if (act1 == kResCollect) listed1 = gather_list(1, parity, &sh->hist[1][0], own_n1, over1);
        __syncthreads();
        // ---- decisions (the same in every workgroup of the tile)
        if (threadIdx.x < 2 * kWave) {      // wave s: slot s
            const int s = threadIdx.x / kWave;
            const uint32_t act = s ? act1 : act0;
            if (act == kResHist) {
                uint32_t b, r_in, cnt;
                const bool found = res_pick<kResBins>(sh->part[(shared0 || s == 0) ? 0 : 1], sh->rank[s], b, r_in, cnt);
                if (lane_id() == 0) {
                    uint32_t nlo, nhi;
                    if (level0) {
                        nlo = b == 0u ? 0u : edge_key(s, (float)b, false);
                        nhi = b == (uint32_t)(kResBins - 1) ? 0xFFFFFFFFu : edge_key(s, (float)(b + 1u), false) - 1u;
                    } else {
                        const uint32_t klo = s ? klo1 : klo0, khi = s ? khi1 : khi0, shf = s ? sh1 : sh0;
                        const unsigned long long lo64 = (unsigned long long)klo + ((unsigned long long)b << shf), hi64 = lo64 + ((1ull << shf) - 1ull);
                        nlo = (uint32_t)lo64;
                        nhi = (uint32_t)(hi64 < (unsigned long long)khi ? hi64 : (unsigned long long)khi);
                    }
                    if (!found || cnt == 0u || nhi < nlo) sh->err |= kResErrCount;
                    sh->klo[s] = nlo;
                    sh->khi[s] = nhi;
                    sh->rank[s] = r_in;
                    sh->n_in[s] = cnt;
                    if (nlo == nhi || !found || cnt == 0u || nhi < nlo) {
                        sh->done[s] = 1u;
                        sh->answer[s] = nlo;
                        sh->action[s] = kResNone;
                    } else {
                        const uint32_t range = nhi - nlo;
                        const int bits = 32 - __clz((int)range);
                        sh->shift[s] = (uint32_t)(bits > 10 ? bits - 10 : 0);
                        sh->action[s] = cnt <= (uint32_t)kResList ? kResCollect : kResHist;
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint32_t act = s ? act1 : act0;
            if (act == kResCollect) {      // uniform
                const uint32_t n = s ? listed1 : listed0;
                const uint32_t n_use = min(n, (uint32_t)kResGather);
                const uint32_t a = res_radix_select(&sh->hist[s][0], n_use, min(sh->rank[s], n_use ? n_use - 1u : 0u), s ? klo1 : klo0, s ? khi1 : khi0, sh);
                if (threadIdx.x == 0) {
                    if (n != sh->n_in[s] || n == 0u || (s ? over1 : over0)) sh->err |= kResErrCount;
                    sh->done[s] = 1u;
                    sh->answer[s] = a;
                    sh->action[s] = kResNone;
                    sh->n_listed[s] = n;
                }
                __syncthreads();
            }
        }
        __syncthreads();
        level0 = false;
        if (sh->done[0] == 1u && sh->done[1] == 1u) return;
    }
    if (threadIdx.x == 0) sh->err |= kResErrIter;
    __syncthreads();
}

// ---- phase R ------------------------------------------------------------------------------------------------------------------
// One pixel: log2 levels -> the tile's 3x3 (reconstruct_item's folding) -> 2^x -> clamp -> cast (-> / 255), as reconstruct_item.
template <typename T, typename O, bool kUnit>
__device__ __forceinline__ void res_pixel_out(const float (&l)[3], const float (&m)[3][3], const float (&k)[3], O (&res)[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const float x = fmaf(m[c][2], l[2], fmaf(m[c][1], l[1], fmaf(m[c][0], l[0], k[c])));
        float rgb = __builtin_amdgcn_exp2f(x);
        rgb = fminf(fmaxf(rgb, 0.0f), 255.0f);
        if constexpr (kUnit) {
            if constexpr (sizeof(T) == 1) res[c] = Elem<O>::store(div255_of_level((float)Elem<T>::store(rgb)));
            else res[c] = Elem<O>::store(div255_of_level(Elem<T>::load(Elem<T>::store(rgb))));
        } else if constexpr (sizeof(T) == 1 && sizeof(O) == 2) {
            res[c] = Elem<O>::store((float)Elem<T>::store(rgb));
        } else {
            res[c] = Elem<O>::store(rgb);
        }
    }
}

// The V pixels of one pack of every lane of a wave, written to plane c.  Same element width in and out: one 16-byte store per
// lane.  Wider output (uint8 in, float32 / half out): a lane's V outputs are Q x 16 bytes side by side, so a store instruction of
// the wave would touch 64 separate pieces; staged through LDS every instruction writes 1 KB of consecutive bytes.
template <typename O, int V>
__device__ __forceinline__ void res_store_pack(O* __restrict__ plane, int64_t p, const O (&vals)[V], bool wave_full, uint4* __restrict__ stage) {
    constexpr int Q = (int)(sizeof(O) * V) / 16;
    static_assert(Q >= 1 && (int)(sizeof(O) * V) % 16 == 0, "whole 16-byte pieces");
    if constexpr (Q == 1) {
        store_pack_stream<O, V>(plane + p, vals);
    } else {
        if (wave_full) {      // wave-uniform
            const uint32_t lane = lane_id();
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                uint4 v;
                __builtin_memcpy(&v, reinterpret_cast<const char*>(vals) + 16 * q, 16);
                stage[lane * Q + q] = v;
            }
            __builtin_amdgcn_wave_barrier();
            char* base = reinterpret_cast<char*>(plane + (p - (int64_t)lane * V));      // the wave's first byte of this pack
            typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const uint4 v = stage[q * kWave + lane];
                f4 f;
                __builtin_memcpy(&f, &v, 16);
                __builtin_nontemporal_store(f, reinterpret_cast<f4*>(base + ((size_t)q * kWave + lane) * 16));
            }
            __builtin_amdgcn_wave_barrier();
        } else {
            store_pack_stream<O, V>(plane + p, vals);
        }
    }
}

template <typename T, typename O, int V, bool kUnit>
__device__ __forceinline__ void res_reconstruct(const ResTile<T>& t, O* __restrict__ dst, const ResPixels& px, int n_px, int64_t p_begin, int64_t p_end, ResScratch* sh, const float (&m)[3][3],
                                                const float (&k)[3]) {
    constexpr int S = kResPx / V;
    const int tq = threadIdx.x & (kStreamThreads - 1);
    uint4* stage = reinterpret_cast<uint4*>(&sh->hist[0][0]) + (threadIdx.x / kWave) * (kWave * 4);      // 4 KB per wave
    const float* __restrict__ mine = sh->l2tab + (lane_id() & (kResCopies - 1));      // this lane's copy of the table
    bool stream = false;
    if constexpr (sizeof(T) != 1) stream = t.stream;
    if (!stream) {
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int64_t p = p_begin + (int64_t)s * (kStreamThreads * V) + (int64_t)tq * V;
            const bool live = s * V < n_px;
            const bool wave_full = __builtin_amdgcn_ballot_w64(live) == ~0ull;
            if (live) {
                O res[3][V];
                uint32_t cw[3][V / 4];
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int q = 0; q < V / 4; ++q) {
                        cw[c][q] = px.w[c][s * (V / 4) + q];
                        asm volatile("" : "+v"(cw[c][q]));      // (opaque: see res_each_code)
                    }
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    const float l[3] = {mine[((cw[0][i >> 2] >> (8 * (i & 3))) & 255u) * kResCopies], mine[((cw[1][i >> 2] >> (8 * (i & 3))) & 255u) * kResCopies],
                                        mine[((cw[2][i >> 2] >> (8 * (i & 3))) & 255u) * kResCopies]};
                    O one[3];
                    res_pixel_out<T, O, kUnit>(l, m, k, one);
                    res[0][i] = one[0];
                    res[1][i] = one[1];
                    res[2][i] = one[2];
                    if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) res_store_pack<O, V>(dst + (int64_t)c * t.pixels, p, res[c], wave_full, stage);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        if constexpr (sizeof(T) != 1) {
            constexpr int B = V >= 8 ? 2 : 4;
#pragma unroll 1
            for (int s0 = 0; s0 < S; s0 += B) {
                PixelPacks<T, V, false> pk[B];
                bool live[B];
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const int64_t p = p_begin + (int64_t)(s0 + b) * (kStreamThreads * V) + (int64_t)tq * V;
                    live[b] = p < p_end;
                    pk[b].clear();
                    if (live[b]) pk[b].load(t.img, t.pixels, p);
                }
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const int64_t p = p_begin + (int64_t)(s0 + b) * (kStreamThreads * V) + (int64_t)tq * V;
                    const bool wave_full = __builtin_amdgcn_ballot_w64(live[b]) == ~0ull;
                    if (live[b]) {
                        O res[3][V];
#pragma unroll
                        for (int i = 0; i < V; ++i) {
                            const float l[3] = {log2_level<T>(pk[b].value(0, i)), log2_level<T>(pk[b].value(1, i)), log2_level<T>(pk[b].value(2, i))};
                            O one[3];
                            res_pixel_out<T, O, kUnit>(l, m, k, one);
                            res[0][i] = one[0];
                            res[1][i] = one[1];
                            res[2][i] = one[2];
                        }
#pragma unroll
                        for (int c = 0; c < 3; ++c) res_store_pack<O, V>(dst + (int64_t)c * t.pixels, p, res[c], wave_full, stage);
                    }
                }
            }
        }
    }
}

// The tile's moments: every work item's partial sums in index order (exact_plane's order: the same doubles).
__device__ __forceinline__ void res_sum_partials(const double* __restrict__ partial, int items, bool through, double* dst) {
    if (threadIdx.x < kPartial) {
        double running = 0.0;
        for (int b0 = 0; b0 < items; b0 += 16) {
            double part[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                part[u] = 0.0;
                if (b0 + u < items) part[u] = through ? ld_agent(&partial[(size_t)(b0 + u) * kPartial + threadIdx.x]) : partial[(size_t)(b0 + u) * kPartial + threadIdx.x];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (b0 + u < items) running += part[u];
        }
        dst[threadIdx.x] = running;
    }
}

template <typename T, typename O, int V>
__global__ __launch_bounds__(kResThreads) void resident_kernel(const T* __restrict__ images, O* __restrict__ out, ResGeom rg, ResWork rw, const float* __restrict__ stain_matrix,
                                                               const float* __restrict__ target_max_conc) {
    __shared__ ResScratch sh;
    res_fill_tables<T>(&sh);
    if (threadIdx.x == 0) sh.err = 0u;
    const int G = rg.group;
    int tile_local, r;
    if (rg.xcd_map) {
        const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
        tile_local = x + 8 * (y / G);
        r = y % G;
    } else {
        tile_local = blockIdx.x / G;
        r = blockIdx.x % G;
    }
    const int quad = threadIdx.x / kStreamThreads, wave = threadIdx.x / kWave;      // quad q of the workgroup takes part in its work items q and q + 2
    __syncthreads();
    for (int round = 0; round < rg.rounds; ++round) {
        const int64_t tile = (int64_t)round * rg.tiles_per_round + tile_local;
        if (tile >= rg.n_tiles) break;
        ResTile<T> t;
        t.img = images + tile * 3 * rg.pixels;
        t.pixels = rg.pixels;
        {
            const int item0 = r * kResQuads + quad, item1 = item0 + kResQuads / kResHalves;
            t.p_begin0 = item0 < rg.items ? (int64_t)item0 * rg.chunk : 0;
            t.p_end0 = item0 < rg.items ? min(t.p_begin0 + (int64_t)rg.chunk, rg.pixels) : 0;
            t.p_begin1 = item1 < rg.items ? (int64_t)item1 * rg.chunk : 0;
            t.p_end1 = item1 < rg.items ? min(t.p_begin1 + (int64_t)rg.chunk, rg.pixels) : 0;
        }
        t.tile = (int)tile;
        t.r = r;
        t.group = G;
        t.stream = false;
        t.sy = &rw.sync[tile];
        t.xchg = rw.xchg + (size_t)tile * G * 2 * kResXchgWords;
        t.spin_limit = rg.spin_limit;
        GroupState& st = rw.state[tile];
        uint32_t gen = 0;
        if (r == 0) SX_STAMP(st, 0);

        // ---------------------------------------------------------------- phase L
        ResPixels px0, px1;
        int n_px0, n_px1;
        {
            uint32_t bad = 0u;
            float l2min[3] = {__builtin_huge_valf(), __builtin_huge_valf(), __builtin_huge_valf()}, l2max[3] = {-__builtin_huge_valf(), -__builtin_huge_valf(), -__builtin_huge_valf()};
            auto load_half = [&](auto half, ResPixels& px, int& n_px, int64_t p_begin, int64_t p_end) {
                constexpr int h = decltype(half)::value;
                double acc[kPartial];
                uint32_t bad_h;
                float lo[3], hi[3];
                res_load_phase<T, V>(t.img, rg.pixels, p_begin, p_end, px, n_px, &sh, acc, bad_h, lo, hi);
                bad |= bad_h;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    l2min[c] = fminf(l2min[c], lo[c]);
                    l2max[c] = fmaxf(l2max[c], hi[c]);
                }
#pragma unroll
                for (int k = 0; k < kPartial; ++k) {
                    const double s = wave_total_f64(acc[k]);      // fixed order; lane 63 holds the total
                    if (lane_id() == kWave - 1) sh.red[h * (kResThreads / kWave) + wave][k] = s;
                }
            };
            load_half(std::integral_constant<int, 0>{}, px0, n_px0, t.p_begin0, t.p_end0);
            if constexpr (kResHalves == 2) {
                load_half(std::integral_constant<int, 1>{}, px1, n_px1, t.p_begin1, t.p_end1);
            } else {
                n_px1 = 0;
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int i = 0; i < kResPx / 4; ++i) px1.w[c][i] = 0u;
            }
            // the workgroup's range of log2 levels (float keys order like the floats) and whether all its elements are 8-bit levels
            uint32_t kmin[3], kmax[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                kmin[c] = wave_min_u32(float_key(l2min[c]));
                kmax[c] = wave_max_u32(float_key(l2max[c]));
            }
            const uint64_t any_bad = __builtin_amdgcn_ballot_w64(bad != 0u);
            if (threadIdx.x < 6) sh.range_key[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
            if (threadIdx.x == 0) sh.bad = 0u;
            __syncthreads();
            if (lane_id() == 0) {
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    atomicMin(&sh.range_key[c], kmin[c]);
                    atomicMax(&sh.range_key[3 + c], kmax[c]);
                }
                if (any_bad) atomicOr(&sh.bad, 1u);
            }
            if (threadIdx.x < kResQuads * kPartial) {      // work item i of the workgroup: half i / 2, quad i % 2
                const int i = threadIdx.x / kPartial, k = threadIdx.x % kPartial, h = i / (kResQuads / kResHalves), q = i % (kResQuads / kResHalves);
                double s = 0.0;
#pragma unroll
                for (int w = 0; w < kStreamThreads / kWave; ++w) s += sh.red[h * (kResThreads / kWave) + q * (kStreamThreads / kWave) + w][k];
                sh.item_part[i][k] = s;
            }
            __syncthreads();
        }
        if (r == 0) SX_STAMP(st, 1);      // phase L done
        double* tile_partial = rw.partial + (size_t)tile * rg.items * kPartial;
        if (G > 1 && threadIdx.x < kResQuads * kPartial) {
            const int q = threadIdx.x / kPartial, k = threadIdx.x % kPartial;
            if (r * kResQuads + q < rg.items) st_agent(&tile_partial[(size_t)(r * kResQuads + q) * kPartial + k], sh.item_part[q][k]);
        }
        if (G > 1) {
            uint32_t* head = rw.head + ((size_t)tile * G + r) * 16;
            if (threadIdx.x == 0) st_agent(&head[0], sh.bad);
            if (threadIdx.x < 6) st_agent(&head[1 + threadIdx.x], sh.range_key[threadIdx.x]);
        }
        res_tile_sync(t.sy, G, gen, rg.spin_limit, &sh);
        if (G > 1) {
            res_sum_partials(tile_partial, rg.items, true, sh.mom);
        } else if (threadIdx.x < kPartial) {      // (the same additions in the same order, from LDS)
            double running = 0.0;
            for (int q = 0; q < rg.items; ++q) running += sh.item_part[q][threadIdx.x];
            sh.mom[threadIdx.x] = running;
        }
        if (threadIdx.x == kPartial) sh.mom[kPartial] = (double)rg.pixels;
        if (threadIdx.x > kPartial && threadIdx.x < kMoments) sh.mom[threadIdx.x] = 0.0;
        if (G > 1 && threadIdx.x >= kWave && threadIdx.x < kWave + 7) {      // the tile's flags and ranges
            const int f = threadIdx.x - kWave;
            uint32_t v = f == 0 ? 0u : (f <= 3 ? 0xFFFFFFFFu : 0u);
            for (int q = 0; q < G; ++q) {
                const uint32_t x = ld_agent(&rw.head[((size_t)tile * G + q) * 16 + f]);
                v = f == 0 ? (v | x) : (f <= 3 ? min(v, x) : max(v, x));
            }
            if (f == 0) sh.bad = v; else sh.range_key[f - 1] = v;
        }
        __syncthreads();
        if constexpr (sizeof(T) != 1) t.stream = sh.bad != 0u;

        // fewer than three kept pixels (torch_backend.py:409-410): the moments of ALL pixels, same grouping (stats_item_all_pixels)
        if (__builtin_expect(sh.mom[0] < 3.0, 0)) {      // uniform over the tile
            constexpr int kShortRun = 32 / V > 0 ? 32 / V : 1;
            auto all_half = [&](auto half, const ResPixels& px, int n_px, int64_t p_begin, int64_t p_end) {
                constexpr int h = decltype(half)::value;
                double acc[kPartial];
                float m[kPartial];
#pragma unroll
                for (int k = 0; k < kPartial; ++k) {
                    acc[k] = 0.0;
                    m[k] = 0.0f;
                }
                int seen = 0;
                auto add_word = [&](const float (&l2w)[4][3]) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float od[3] = {fmaf(-kLn2, l2w[e][0], kLnIo), fmaf(-kLn2, l2w[e][1], kLnIo), fmaf(-kLn2, l2w[e][2], kLnIo)};
                        m[0] += 1.0f;
                        m[1] += od[0];
                        m[2] += od[1];
                        m[3] += od[2];
                        m[4] = fmaf(od[0], od[0], m[4]);
                        m[5] = fmaf(od[0], od[1], m[5]);
                        m[6] = fmaf(od[0], od[2], m[6]);
                        m[7] = fmaf(od[1], od[1], m[7]);
                        m[8] = fmaf(od[1], od[2], m[8]);
                        m[9] = fmaf(od[2], od[2], m[9]);
                    }
                    seen += 4;
                    if (seen == kShortRun * V) {
#pragma unroll
                        for (int k = 0; k < kPartial; ++k) {
                            acc[k] += (double)m[k];
                            m[k] = 0.0f;
                        }
                        seen = 0;
                    }
                };
                bool stream = false;
                if constexpr (sizeof(T) != 1) stream = t.stream;
                if (!stream) {
                    res_each_code(px, n_px, sh.l2tab, add_word);
                } else {
                    if constexpr (sizeof(T) != 1) res_each_stream<T, V>(t.img, t.pixels, p_begin, p_end, add_word);
                }
#pragma unroll
                for (int k = 0; k < kPartial; ++k) acc[k] += (double)m[k];
#pragma unroll
                for (int k = 0; k < kPartial; ++k) {
                    const double s = wave_total_f64(acc[k]);
                    if (lane_id() == kWave - 1) sh.red[h * (kResThreads / kWave) + wave][k] = s;
                }
            };
            all_half(std::integral_constant<int, 0>{}, px0, n_px0, t.p_begin0, t.p_end0);
            if constexpr (kResHalves == 2) all_half(std::integral_constant<int, 1>{}, px1, n_px1, t.p_begin1, t.p_end1);
            __syncthreads();
            double* tile_all = rw.partial_all + (size_t)tile * rg.items * kPartial;
            if (threadIdx.x < kResQuads * kPartial) {
                const int i = threadIdx.x / kPartial, k = threadIdx.x % kPartial, h = i / (kResQuads / kResHalves), q = i % (kResQuads / kResHalves);
                double s = 0.0;
#pragma unroll
                for (int w = 0; w < kStreamThreads / kWave; ++w) s += sh.red[h * (kResThreads / kWave) + q * (kStreamThreads / kWave) + w][k];
                sh.item_part[i][k] = s;
                if (G > 1 && r * kResQuads + i < rg.items) st_agent(&tile_all[(size_t)(r * kResQuads + i) * kPartial + k], s);
            }
            res_tile_sync(t.sy, G, gen, rg.spin_limit, &sh);
            if (G > 1) {
                res_sum_partials(tile_all, rg.items, true, sh.mom + kPartial);
            } else if (threadIdx.x < kPartial) {
                double running = 0.0;
                for (int q = 0; q < rg.items; ++q) running += sh.item_part[q][threadIdx.x];
                sh.mom[kPartial + threadIdx.x] = running;
            }
            __syncthreads();
        }

        // ---------------------------------------------------------------- plane
        if (threadIdx.x < 2) {
            double cov[9];
            float vecs[6];
            bool use_all;
            unsigned long long n_sel;
            plane_from_moments<true>(sh.mom, true, cov, vecs, use_all, n_sel);
            if (threadIdx.x == 0) {
#pragma unroll
                for (int i = 0; i < 6; ++i) sh.vecs[i] = vecs[i];
                sh.use_all = use_all ? 1 : 0;
                sh.n_sel = n_sel;
                if (r == 0) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) {
                        put(&st.vecs[i], vecs[i]);
                        put(&st.rec[0].coef[i], vecs[i]);
                    }
#pragma unroll
                    for (int k = 0; k < kMoments; ++k) put(&st.mom[k], sh.mom[k]);
#pragma unroll
                    for (int i = 0; i < 9; ++i) put(&st.cov[i], cov[i]);
                    put(&st.use_all, use_all ? 1 : 0);
                    put(&st.rec[0].use_all, use_all ? 1 : 0);
                    put(&st.n_sel, n_sel);
                    put(&st.fell_back, 0u);
                    put(&st.spec, 0u);
                }
            }
        }
        __syncthreads();
        if (r == 0) SX_STAMP(st, 2);      // moments exchanged, plane known
        float v[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = sh.vecs[i];
        const bool use_all = sh.use_all != 0;
        const unsigned long long n_sel = sh.n_sel;

        // ---------------------------------------------------------------- the two angle percentiles (torch_backend.py:421-422)
        const unsigned long long rank1 = nearest_rank_index(1.0, n_sel), rank99 = nearest_rank_index(99.0, n_sel);
        res_stage<T, V, false>(t, px0, px1, n_px0, n_px1, &sh, gen, v, use_all, (uint32_t)rank1, (uint32_t)rank99, (uint32_t)n_sel, (uint32_t)n_sel, st, 3);
        if (threadIdx.x == 0) {
            float he[6], pinv[6];
            stain_vectors_and_pinv(v, sh.answer[0], sh.answer[1], he, pinv);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                sh.he[i] = he[i];
                sh.pinv[i] = pinv[i];
            }
            // level-0 bins of the concentrations: the box of optical densities the tile's levels span bounds row j of pinv . od;
            // bins of width w = 2^e (c / w is exact) aligned at multiples of w, kResBins - 2 of them over that range
            float od_lo[3], od_hi[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                od_lo[c] = fmaf(-kLn2, key_float(sh.range_key[3 + c]), kLnIo);
                od_hi[c] = fmaf(-kLn2, key_float(sh.range_key[c]), kLnIo);
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float lo = 0.0f, hi = 0.0f;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float a = pinv[3 * j + c] * od_lo[c], b = pinv[3 * j + c] * od_hi[c];
                    lo += fminf(a, b);
                    hi += fmaxf(a, b);
                }
                float span = (hi - lo) * 1.001f + 1e-6f;
                if (!(span > 0.0f) || !(span < 1e30f)) span = 1.0f;
                int e;
                (void)frexpf(span / (float)(kResBins - 2), &e);      // span / bins = f 2^e, f in [0.5, 1): w = 2^e >= it
                e = max(min(e, 60), -60);
                const float w = ldexpf(1.0f, e);
                sh.conc_w[j] = w;
                sh.conc_inv_w[j] = ldexpf(1.0f, -e);
                float o = floorf(lo * ldexpf(1.0f, -e)) - 1.0f;
                if (!(fabsf(o) < 8e6f)) o = 0.0f;      // (org + bin must stay an exact float)
                sh.conc_org[j] = o;
            }
            if (r == 0) {
                put(&st.phi_key[0], sh.answer[0]);
                put(&st.phi_key[1], sh.answer[1]);
                put(&st.rank[0], rank1);
                put(&st.rank[1], rank99);
                put(&st.ncand_seen[0], sh.n_listed[0]);
                put(&st.ncand_seen[1], sh.n_listed[1]);
            }
        }
        __syncthreads();
        float pinv[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) pinv[i] = sh.pinv[i];

        // ---------------------------------------------------------------- the two concentration percentiles (torch_backend.py:447-448)
        const unsigned long long k99 = nearest_rank_index(99.0, (unsigned long long)rg.pixels);
        res_stage<T, V, true>(t, px0, px1, n_px0, n_px1, &sh, gen, pinv, true, (uint32_t)k99, (uint32_t)k99, (uint32_t)rg.pixels, (uint32_t)rg.pixels, st, 7);
        res_tile_leave(t.sy, G);
        const float mc0 = key_float(sh.answer[0]), mc1 = key_float(sh.answer[1]);
        const float scale0 = target_max_conc[0] / mc0, scale1 = target_max_conc[1] / mc1;      // torch_backend.py:452
        if (threadIdx.x == 0 && r == 0) {
            put(&st.max_c[0], mc0);
            put(&st.max_c[1], mc1);
            put(&st.rank[2], k99);
            put(&st.rank[3], k99);
            put(&st.ncand_seen[2], sh.n_listed[0]);
            put(&st.ncand_seen[3], sh.n_listed[1]);
            StageRecord* rec = &st.rec[2];
            put(&rec->scale[0], scale0);
            put(&rec->scale[1], scale1);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                put(&rec->coef[i], pinv[i]);
                put(&st.pinv[i], pinv[i]);
                put(&st.he[i], sh.he[i]);
            }
            if (sh.err) {
                put(&st.spec, 0x80000000u | (sh.err << 8));
                atomicAdd(&rw.state[0].spin_timeouts, 1u);
            }
        }

        // ---------------------------------------------------------------- phase R (reconstruct_item's folding of the linear chain)
        float m[3][3], k[3];
        {
            const double s0 = (double)scale0, s1 = (double)scale1;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double row = 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double val = (double)stain_matrix[c * 2] * s0 * (double)pinv[j] + (double)stain_matrix[c * 2 + 1] * s1 * (double)pinv[3 + j];
                    m[c][j] = (float)val;
                    row += (double)m[c][j];
                }
                k[c] = (float)(7.90689059560851852932 * (1.0 - row));      // log2(240)
            }
        }
        __syncthreads();      // (the stage's last reads of the LDS the stores stage through)
        if (r == 0) SX_STAMP(st, 11);
        if (rg.unit) {
            res_reconstruct<T, O, V, true>(t, out + tile * 3 * rg.pixels, px0, n_px0, t.p_begin0, t.p_end0, &sh, m, k);
            if constexpr (kResHalves == 2) res_reconstruct<T, O, V, true>(t, out + tile * 3 * rg.pixels, px1, n_px1, t.p_begin1, t.p_end1, &sh, m, k);
        } else {
            res_reconstruct<T, O, V, false>(t, out + tile * 3 * rg.pixels, px0, n_px0, t.p_begin0, t.p_end0, &sh, m, k);
            if constexpr (kResHalves == 2) res_reconstruct<T, O, V, false>(t, out + tile * 3 * rg.pixels, px1, n_px1, t.p_begin1, t.p_end1, &sh, m, k);
        }
        __syncthreads();
        if (r == 0) SX_STAMP(st, 12);
        if (threadIdx.x == 0) sh.err = 0u;
    }
}

}  // namespace macenko
}  // namespace sx

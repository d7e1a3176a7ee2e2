// Macenko stain normalisation for MI355X (gfx950) -- hand-written HIP, wave64.
//
// What it computes is MacenkoTorch.transform / compute_reference_stain_matrix_torch of
// rendeirolab/stainx (src/stainx/backends/torch_backend.py:399-560).  How it computes it is this
// library's own design:
//
//   * four streaming stages over the pixels (planar NCHW or interleaved NHWC, 16-byte loads per lane, 256-thread
//     workgroups, 16384 pixels per work item, every wave on its own inside the pixel loop) separated by three small
//     per-tile stages (one 1024-thread workgroup per tile, everything they need fetched in one batch of loads):
//        S1 moments -> [plane] -> S2 angle pass -> [stain] -> S3 concentration pass -> [scale] -> S4 reconstruct
//   * the four nearest-rank order statistics per tile (phi@1%, phi@99%, C0@99%, C1@99%;
//     torch_backend.py:363-365) are EXACT but never sort: a 4096-pixel strided sample of the tile (its
//     optical density is written out by S1 on the way) gives a bracket [lo,hi] around each wanted rank; the
//     streaming stage counts the pixels below the bracket (integer adds), gathers the few keys inside it
//     and histograms them bracket-relative; the per-tile stage sums the work items' histograms in a fixed
//     order, picks the bin holding the wanted rank and rank-counts the ~n/256 keys of that bin.  If a
//     bracket misses or overflows, or a bin is crowded (heavy ties, adversarial data), radix selects
//     over the candidates / over the whole tile give the same answer, slower;
//   * the 3x3 covariance is accumulated in fp64 (raw moments cancel catastrophically in fp32) and
//     diagonalised by cyclic Jacobi in fp64; eigenvector signs follow the "positive component sum"
//     convention (the transform is invariant to them on real H&E tiles);
//   * seven stream-ordered launches per transform, no host synchronisation; the pooled fit (one group over
//     all tiles) runs the first six with one group.
#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <type_traits>
#include <cstdlib>

namespace sx {
namespace macenko {

constexpr int kSample = 4096;          // strided sample per tile
constexpr int kMinCap = 8192;          // candidate keys per selection slot: at least this (what the per-tile stages request up front); see cap_for()
constexpr int kChunk = 16384;          // pixels per work item (64 per lane of a 256-thread workgroup)
template <typename T> struct PackOf { static constexpr int n = 16 / (int)sizeof(T); };   // pixels per 16-byte load: f32 4, bf16/f16 8, u8 16, f64 2
constexpr int kSlots = 4;              // 0: phi@1  1: phi@99  2: C0@99  3: C1@99
constexpr int kMoments = 20;           // [cnt,sx,sy,sz,xx,xy,xz,yy,yz,zz] of the pixels kept by the OD filter, then of all pixels
constexpr int kPartial = 10;           // what S1 accumulates per work item: the kept set only (the all-pixel set is a rare fallback)
constexpr int kShortList = 512;        // keys of one histogram bin gathered for rank counting
constexpr int kGroupThreads = 1024;    // per-tile stages: one workgroup per group, as their own launch
constexpr int kKeys = kSample / kGroupThreads;   // sample keys a thread of a per-tile stage holds in registers
constexpr int kPrefetchHist = 8;       // work-item histograms a thread fetches up front (2 threads per bin: 32 work items = one 512x512 tile)
constexpr int kPrefetchCand = 8;       // candidates per slot a thread fetches up front (8192 per slot)
static_assert(kKeys * kGroupThreads == kSample && kPrefetchCand * kGroupThreads <= kMinCap, "per-tile stage geometry");

constexpr float kBeta = 0.15f;         // torch_backend.py:542
constexpr float kIo = 240.0f;          // torch_backend.py:541
constexpr float kLn2 = 0.693147180559945309f;
constexpr float kLnIo = 5.48063892334199f;       // ln 240

// What a streaming stage needs to know about its tile; written once per call by the per-tile stage before
// it (one record per stage, each on its own 128-byte line).
struct alignas(128) StageRecord {
    float coef[6];                // S2: plane vectors (3,2); S3/S4: pseudo-inverse (2,3)
    float scale[2];               // S4: target_max_conc / max_c
    uint32_t lo[2], hi[2];        // brackets of the two slots of the stage
    double bin_origin[2], bin_scale[2];   // bracket-relative bin of a candidate: (value - origin) * scale
    int32_t use_all;              // S2: fewer than 3 pixels pass the OD filter -> every pixel is selected
    int32_t pad;
};

struct alignas(256) GroupState {
    StageRecord rec[3];           // inputs of S2, S3, S4
    double mom[kMoments];
    double cov[9];
    float vecs[6];                // (3,2) row-major, columns [middle, largest] eigenvalue
    float he[6];                  // (3,2) row-major HE_source
    float pinv[6];                // (2,3) row-major pseudo-inverse of HE_source
    uint32_t phi_key[2];          // the two selected angle keys
    float max_c[2];
    unsigned long long n_sel;     // pixels in the selection set (kept by the OD filter, or all)
    unsigned long long rank[kSlots];   // wanted 0-based rank inside the selection set
    uint32_t below[kSlots];       // keys < lo            (atomic, integer => order independent)
    uint32_t ncand[kSlots];       // keys in [lo,hi]      (atomic)
    uint32_t ncand_seen[kSlots];  // copy kept for sx_macenko_tile_params
    int32_t use_all;
    uint32_t fell_back;           // bit s: slot s used the whole-group radix select; bit 4+s: candidate radix select
    uint32_t spec;                // two-pass transform (macenko_twopass.hpp): kSpecSlow | kSpecHazard
    unsigned long long phi_pub[2];      // two-pass transform: {tag, angle key} granules the two stage workgroups of a tile hand each other
    uint32_t over_count[kSlots];  // two-pass transform: candidates of the tile that did not fit their wave's segment (overflow area fill)
    uint32_t slow_slots;          // two-pass transform, tile 0 only: a RUNNING count of selections that left the speculative path (telemetry: only ever added to,
                                  // so that a host that reads it late -- the next call's kernels may already be running -- still sees what happened)
    uint32_t spin_timeouts;       // fused transform, tile 0 only, directly behind slow_slots: a RUNNING count of bounded waits that ran out (the call's
                                  // output is then incomplete; never observed -- the host reads it with the telemetry and raises)
    unsigned long long stamp[16]; // diagnostic: wall_clock64() at stage boundaries of the per-tile stages
};

struct Geometry {
    int64_t n_tiles, pixels;      // P = H*W
    int blocks_per_tile;          // work items per tile and stage
    int vec;                      // 1: 4-pixel packs (16-byte loads), 0: scalar accesses
    int pooled;                   // 1: one group over all tiles (fit), 0: one group per tile
    int sample_stride, sample_count;   // sample j = pixel j*stride of the group, j < count
    uint32_t cap;                 // candidate keys per selection slot
    int interleaved;              // 1: tiles are (H,W,3) -- the three values of a pixel side by side -- instead of three planes
    int spread;                   // 1: pooled group of several tiles; candidates, counters and histograms stay per tile (PoolState)
    int distributed;              // 1: the group also spans other ranks (sx_macenko_pfit_*): totals come from the host, no local fallback
    long long n_all;              // distributed: pixels of the whole group over all ranks
    int fine_chunk, fine_blocks;  // small batches: pixels per work item / work items per tile of the bracket and reconstruct stages (0: kChunk)
    int recon_chunk, recon_blocks; // reconstruct stage: pixels per work item / work items per tile when it splits finer than the other stages (0: as them)
    int fast;                     // precision="fast": sample percentiles instead of the exact ones
    int chunk;                    // pixels per work item: the tile split evenly over its blocks_per_tile work items (<= kChunk)
    int vec_width;                // pixels per 16-byte pack of the element type (host side, for the chunk rounding)
    int no_tie;                   // diagnostic (SX_MACENKO_NO_TIE_SHORTCUT): never resolve a closed bracket from its counts alone
    int out_code;                 // uint8 input only: SX_BF16 / SX_F16 output (SX_MACENKO_OUT_*), 0 = the reference's output type
    int two_pass;                 // transform: the two-pass form (macenko_twopass.hpp) instead of the four passes of this file
    int prior_units;              // two-pass: 16-pixel sectors the prior stage samples per tile
    unsigned prior_step_q16;      // two-pass: sectors per sample cell, 16.16 fixed point
    uint32_t cap2;                // two-pass: candidate records per tile and slot
    uint32_t seg_cap;             // two-pass: ... of which every wave of pass A owns this many (its segment)
    uint32_t over_cap;            // two-pass: ... and the tile's overflow area behind the segments holds this many
    int n_seg;                    // two-pass: segments per tile = waves of pass A per tile
    int spec_fail;                // diagnostic (SX_MACENKO_SPEC_FAIL): treat every speculation as failed -> the slow exact path
    float spec_kw, spec_eff_far, spec_eff_near, spec_rot, spec_sigmas, spec_sigmas_conc, spec_tscale;      // two-pass: the speculation's knobs (macenko_twopass.hpp: kSpecKw ...)
    int fused;                    // two-pass: pass A, the per-tile stages and the reconstruct pass in ONE launch (macenko_fused.hpp)
    int dense;                    // two-pass: candidates in dense record arrays (tiles up to 512 x 512) instead of a segment per wave
    uint32_t fused_cap;           // fused: candidate records per tile and slot
    int fused_items;              // fused: pass-A work items of the call (= reconstruct work items)
    uint32_t code_epoch;          // four passes over float tiles: non-zero = the first pass leaves 8-bit codes, the later passes read them (Coded<F>); the call's number
};

// Every field named: the struct is filled at half a dozen entry points.
static Geometry make_geometry(int64_t n_tiles, int64_t pixels, int pooled) {
    Geometry g{};
    g.n_tiles = n_tiles;
    g.pixels = pixels;
    g.blocks_per_tile = (int)((pixels + kChunk - 1) / kChunk);
    g.pooled = pooled;
    g.sample_stride = 1;
    g.cap = (uint32_t)kMinCap;
    g.chunk = kChunk;
    g.vec_width = 1;
    return g;
}

// Pooled fit over several tiles ("spread" mode): the streaming stages keep candidates, counters and histograms per
// tile exactly as the transform does (no counter shared by 4096 waves); between them a pair of small many-workgroup
// kernels adds the tiles up and moves the few candidates of the wanted histogram bin here, where the one workgroup
// of the group stage finishes the selection.
constexpr int kCompact = 32768;        // candidates of the picked bin, per slot, over all tiles
struct alignas(256) PoolState {
    uint32_t hist[2][2][256];          // [stage][slot][bin]: sum of the work items' histograms over all tiles
    uint32_t below[kSlots], ncand[kSlots];
    uint32_t compact_n[kSlots];
    uint32_t rank_in_bin[kSlots], range[kSlots][2];
    uint32_t ok[kSlots], overflow;     // overflow bit s: a tile's candidate buffer overflowed in slot s
    uint32_t status;                   // distributed fit: non-zero when a bracket did not hold (the caller falls back to the radix rounds)
    uint32_t arrived;                  // distributed fit: workgroups of the gather launch that have moved their candidates (the last one closes the rank's record)
    uint32_t compact[kSlots][kCompact];
};

struct Workspace {
    GroupState* state;
    double* partial;              // [n_tiles*blocks_per_tile][kPartial]
    double* partial_all;          // same shape: moments of ALL pixels, written only by work items without kept pixels
    uint32_t* cand;               // [groups][kSlots][cap]
    uint32_t* block_hist;         // [n_tiles*blocks_per_tile][2][256] bracket-relative histograms of a stage's candidates
    float* sample_od;             // [groups][3][kSample] optical density of the strided sample
    struct PoolState* pool;       // pooled fit over several tiles: group-level sums and the compacted candidates
    struct PriorRecord* prior;    // two-pass transform: one record per tile
    float* cand_od;               // two-pass transform: [n_tiles][kSlots][3][cap2] optical density of the candidates
    uint32_t* seg_count;          // two-pass transform: [n_tiles][kSlots][n_seg] candidates each wave of pass A produced
    uint32_t* key_spill;          // two-pass transform: [n_tiles][kSlots][cap2 - kLdsKeys] keys of a slot beyond the stages' LDS array (big tiles)
    struct FusedSched* fsched;    // fused transform: the launch's ticket counter and error word
    struct FusedTile* ftile;      // fused transform: per tile, the counters the roles of the launch hand each other work through
    uint4* cand_rec;              // fused transform: [n_tiles][kSlots][fused_cap] candidate records (od0, od1, od2, -)
    uint32_t* dense_spill;        // two-pass transform, dense records: [n_tiles][kSlots][fused_cap - kLdsKeys] keys of a slot beyond the stages' LDS array
    uint32_t* code_bad;           // coded four passes: [n_tiles] the number of the last call whose first pass found a non-8-bit element in the tile
    uint8_t* codes;               // coded four passes: [n_tiles][3][pixels] grey-level codes (both overlay the areas behind the base level: the four passes use none of them)
};

__host__ __device__ inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// set_chunk(): which tiles may take one work item more than pixels / kChunk -- those whose two-pass form keeps its candidates in dense
// records (fused_size() below: the per-wave candidate segments of the other tiles are sized per work item and stay as they were)
static bool even_items_size(int64_t pixels) { return pixels >= 16384 && pixels <= 262144 && pixels % 4 == 0; }
// (an upper bound: what the workspace is sized for; the geometry's blocks_per_tile is set_chunk()'s)
#ifdef SX_STAMPS
static int blocks_per_tile_for(int64_t pixels) { return (int)((pixels + kChunk - 1) / kChunk) + (even_items_size(pixels) ? 8 : 0); }
#else
static int blocks_per_tile_for(int64_t pixels) { return (int)((pixels + kChunk - 1) / kChunk) + (even_items_size(pixels) ? 1 : 0); }
#endif

// Candidate capacity per slot of a group of `count` pixels: brackets hold ~2-3 % of the pixels.
static uint32_t cap_for(int64_t count) {
    // 1/8 of a small group's pixels (a 224 x 224 tile: 8192 keys per slot, not the 32768 that made a 256-tile bf16 batch's
    // workspace four times the batch), 32768 from 256 x 256 on, 1/16 of the pixels for big groups
    int64_t cap = kMinCap;
    while (cap < count / 8 && cap < 32768) cap *= 2;
    while (cap < count / 16) cap *= 2;
    return (uint32_t)cap;
}
// A batch of n tiles is used as n groups of P pixels (transform) or as ONE group of n*P pixels (pooled fit); the
// candidate area is sized for the larger of the two layouts.
static size_t cand_words(int64_t n_tiles, int64_t pixels) {
    return std::max((size_t)n_tiles * cap_for(pixels), (size_t)cap_for(n_tiles * pixels)) * kSlots;
}

// Two-pass transform: candidate records per tile and slot (the prior keeps ~3 % of the pixels per slot; an overflow is
// detected and sends the slot to the slow path), and whether a tile size takes that form at all.
constexpr int kLdsKeys = 16384;        // candidate keys of a slot the stages keep in LDS; the rest spill to the classic candidate area
constexpr size_t kPriorRecordBytes = 384;
static_assert(kLdsKeys >= 14336, "a dense candidate array (kFusedCapMax records) fits the stage's LDS key array");
// per tile and slot: one segment per wave of pass A (an eighth of the wave's pixels, at least 128), and an overflow area of half
// the segments' total behind them; cap2 is the stride of the three planes of a slot's arrays
static int two_pass_segments(int64_t pixels) { return (int)((pixels + kChunk - 1) / kChunk) * (kStreamThreads / kWave); }
static uint32_t seg_cap_for(int64_t pixels) {
    const int n_seg = two_pass_segments(pixels);
    return (uint32_t)std::max<int64_t>(std::min<int64_t>(std::max<int64_t>(pixels / 8, 8192), 131072) / n_seg, 128);
}
static uint32_t over_cap_for(int64_t pixels) { return (uint32_t)(((int64_t)two_pass_segments(pixels) * seg_cap_for(pixels) / 2 + 255) / 256 * 256); }
static uint32_t cap2_for(int64_t pixels) { return (uint32_t)((int64_t)two_pass_segments(pixels) * seg_cap_for(pixels)) + over_cap_for(pixels); }
static size_t key_spill_words(int64_t pixels) { return cap2_for(pixels) > (uint32_t)kLdsKeys ? (size_t)cap2_for(pixels) - kLdsKeys : 0; }      // keys of a slot beyond the stages' LDS array
static bool two_pass_size(int64_t pixels) { return pixels >= 256 && pixels <= 64ll * kChunk; }      // (at most 256 waves of pass A per tile)

// Fused transform (macenko_fused.hpp): which tile sizes take it, and its candidate records per tile and slot
constexpr uint32_t kFusedCapMax = 14336;      // (256 threads x 32 keys in registers + 6144 keys in LDS: macenko_fused.hpp)
static bool fused_size(int64_t pixels) { return pixels >= 16384 && pixels <= 262144 && pixels % 4 == 0; }
static uint32_t fused_cap_for(int64_t pixels) { return (uint32_t)std::min<int64_t>(kFusedCapMax, std::max<int64_t>(2048, (pixels / 4 + 255) / 256 * 256)); }
// The four-launch two-pass form's dense records (round 4): an eighth of the tile's pixels per slot, at most 24576 -- the stage keeps the
// first kLdsKeys keys of a slot in LDS and spills the rest (dense_spill).  With the fused launch's 14336 (5.5 % of a 512 x 512 tile) the
// reference's real tiles overflowed: their second concentration slot holds up to 6.5 % of the pixels, an angle slot 3 % on average and
// 10.5 % at most, and an overflow sends the slot to the whole-tile select -- eight of the ten slow slots of tools/diag_real.py's twenty
// tiles.  Over the 150 crops of tools/diag_real_batches.py: 14336 -> (many), 20480 -> 23 tiles with a slow slot (9 of them an overflowing
// angle slot), 32768 -> 15 (none overflowing); 24576 (9.4 % of a 512 x 512 tile) holds all but the one 10.5 % slot and is 34 MB less
// workspace for config 2 than 32768.
constexpr uint32_t kDenseCapMax = 24576;
static uint32_t dense_cap_for(int64_t pixels) { return (uint32_t)std::min<int64_t>(kDenseCapMax, std::max<int64_t>(2048, (pixels / 8 + 255) / 256 * 256)); }
static uint32_t record_cap_for(int64_t pixels) { return std::max(fused_cap_for(pixels), dense_cap_for(pixels)); }      // what the record area is sized for
static size_t dense_spill_words(int64_t pixels) { return dense_cap_for(pixels) > (uint32_t)kLdsKeys ? (size_t)dense_cap_for(pixels) - kLdsKeys : 0; }
constexpr size_t kFusedSchedBytes = 512, kFusedTileBytes = 64;

// The workspace is laid out so that what a form of the transform needs is a PREFIX of the whole:
//   [ base: state, partial sums, classic candidates, histograms, sample, pool ]   every entry point
//   [ prior records | fused: schedule, per-tile counters, candidate records ]      + the fused two-pass transform
//   [ two-pass: candidate optical densities, segment fills, key spill ]            + the four-launch two-pass transform
// sx_macenko_workspace_bytes() is the whole (any call fits); sx_macenko_workspace_bytes_for() the prefix one call needs.
enum WorkspaceLevel { kWsBase = 0, kWsFused = 1, kWsTwoPass = 2 };
// coded four passes (float32 tiles of whole 16-pixel packs, batches of at least 2^20 pixels): the codes and the per-tile flags, behind the base level
static bool coded_size(int64_t n_tiles, int64_t pixels) { return pixels % 16 == 0 && n_tiles * pixels >= (1ll << 20); }
static size_t coded_bytes(int64_t n_tiles, int64_t pixels) { return align_up(sizeof(uint32_t) * (size_t)n_tiles, 256) + align_up((size_t)3 * pixels * n_tiles, 256); }
static size_t workspace_bytes(int64_t n_tiles, int64_t pixels, int level = kWsTwoPass) {
    const size_t b = (size_t)blocks_per_tile_for(pixels), n = (size_t)n_tiles;
    size_t total = align_up(sizeof(GroupState) * n, 256);
    total += 2 * align_up(sizeof(double) * kPartial * b * n, 256);
    total += align_up(sizeof(uint32_t) * cand_words(n_tiles, pixels), 256);
    total += align_up(sizeof(uint32_t) * 512 * b * n, 256);
    total += align_up(sizeof(float) * 3 * kSample * n, 256);
    total += align_up(sizeof(PoolState), 256);
    if (level >= kWsFused && two_pass_size(pixels)) {
        total += align_up(kPriorRecordBytes * n, 256);
        total += kFusedSchedBytes;
        total += align_up(kFusedTileBytes * n, 256);
        if (fused_size(pixels)) total += align_up(sizeof(uint4) * kSlots * (size_t)record_cap_for(pixels) * n, 256);
        if (fused_size(pixels)) total += align_up(sizeof(uint32_t) * kSlots * dense_spill_words(pixels) * n, 256);
    }
    if (level >= kWsTwoPass && two_pass_size(pixels)) {
        total += align_up(sizeof(float) * 3 * kSlots * (size_t)cap2_for(pixels) * n, 256);
        total += align_up(sizeof(uint32_t) * kSlots * 256 * n, 256);
        total += align_up(sizeof(uint32_t) * kSlots * key_spill_words(pixels) * n, 256);
    }
    return total;
}

// codes_at: where the coded passes' area lies (bytes from the base): behind the level of the form the call takes (0: behind the base level)
static Workspace carve(void* base, int64_t n_tiles, int64_t pixels, size_t codes_at = 0) {
    Workspace w;
    char* p = static_cast<char*>(base);
    const size_t b = (size_t)blocks_per_tile_for(pixels), n = (size_t)n_tiles;
    w.state = reinterpret_cast<GroupState*>(p);
    p += align_up(sizeof(GroupState) * n, 256);
    w.partial = reinterpret_cast<double*>(p);
    p += align_up(sizeof(double) * kPartial * b * n, 256);
    w.partial_all = reinterpret_cast<double*>(p);
    p += align_up(sizeof(double) * kPartial * b * n, 256);
    w.cand = reinterpret_cast<uint32_t*>(p);
    p += align_up(sizeof(uint32_t) * cand_words(n_tiles, pixels), 256);
    w.block_hist = reinterpret_cast<uint32_t*>(p);
    p += align_up(sizeof(uint32_t) * 512 * b * n, 256);
    w.sample_od = reinterpret_cast<float*>(p);
    p += align_up(sizeof(float) * 3 * kSample * n, 256);
    w.pool = reinterpret_cast<PoolState*>(p);
    p += align_up(sizeof(PoolState), 256);
    // (beyond the base level the pointers are only used by the forms whose workspace level includes them)
    {
        char* c = codes_at ? static_cast<char*>(base) + codes_at : p;
        w.code_bad = reinterpret_cast<uint32_t*>(c);
        w.codes = reinterpret_cast<uint8_t*>(c + align_up(sizeof(uint32_t) * n, 256));
    }
    w.prior = reinterpret_cast<PriorRecord*>(p);
    p += align_up(kPriorRecordBytes * n, 256);
    w.fsched = reinterpret_cast<FusedSched*>(p);
    p += kFusedSchedBytes;
    w.ftile = reinterpret_cast<FusedTile*>(p);
    p += align_up(kFusedTileBytes * n, 256);
    w.cand_rec = reinterpret_cast<uint4*>(p);
    if (fused_size(pixels)) p += align_up(sizeof(uint4) * kSlots * (size_t)record_cap_for(pixels) * n, 256);
    w.dense_spill = reinterpret_cast<uint32_t*>(p);
    if (fused_size(pixels)) p += align_up(sizeof(uint32_t) * kSlots * dense_spill_words(pixels) * n, 256);
    w.cand_od = reinterpret_cast<float*>(p);
    p += align_up(sizeof(float) * 3 * kSlots * (size_t)cap2_for(pixels) * n, 256);
    w.seg_count = reinterpret_cast<uint32_t*>(p);
    p += align_up(sizeof(uint32_t) * kSlots * 256 * n, 256);
    w.key_spill = reinterpret_cast<uint32_t*>(p);
    return w;
}

// ------------------------------------------------------------------------------------------------
// data handed from one stage to the next: every stage is its own launch, so a kernel boundary publishes it and
// plain accesses suffice.  (Measured alternatives, see DESIGN.md: a persistent work-queue kernel with in-launch
// hand-offs and a 4-way multi-stream split were both slower than seven back-to-back launches.)
// ------------------------------------------------------------------------------------------------
template <typename U> __device__ __forceinline__ void put(U* p, U v) { *p = v; }
template <typename U> __device__ __forceinline__ U get(const U* p) { return *p; }

// Diagnostic time stamps of the per-tile stages (tools/bench_twopass.py, tools/stage_stamps.py) and of the fused launch's units
// (tools/fused_timeline.py): compiled in only with -DSX_STAMPS (tools/build_debug.sh -> libstainx_dbg.so); the product build
// carries none of them (sx_macenko_tile_params then reports zeros in its stamp slots).
#ifdef SX_STAMPS
#define SX_STAMP(st, i) do { if (threadIdx.x == 0) (st).stamp[i] = (unsigned long long)wall_clock64(); } while (0)
#define SX_UNIT_STAMP(ws, unit, i) do { if (threadIdx.x == 0) reinterpret_cast<unsigned long long*>((ws).block_hist)[(size_t)(unit) * 8 + (i)] = (unsigned long long)wall_clock64(); } while (0)
#else
#define SX_STAMP(st, i) do { } while (0)
#define SX_UNIT_STAMP(ws, unit, i) do { } while (0)
#endif

// ------------------------------------------------------------------------------------------------
// per-pixel arithmetic
// ------------------------------------------------------------------------------------------------
// OD = -log((x*255+1)/240) (torch_backend.py:550) evaluated as ln240 - ln2*log2(x*255+1):
// one fma, v_log_f32, one fma.  Differs from the reference's mul/add/div/log chain by ~1e-7 absolute.
// `raw` is the unit value for float inputs and the integer grey level for uint8 (the reference's u/255*255 is u
// up to one rounding).
template <typename T>
__device__ __forceinline__ float log2_level(float raw) {      // log2(255 x + 1): the argument is >= 1 for image data, bare v_log_f32
    const float t = sizeof(T) == 1 ? raw + 1.0f : fmaf(raw, 255.0f, 1.0f);
    return __log2f(t);
}
template <typename T>
__device__ __forceinline__ float optical_density(float raw) {
    return fmaf(-kLn2, log2_level<T>(raw), kLnIo);
}

// uint8 tiles: a channel takes one of 256 values, so its two per-pixel functions -- the optical density and the log2 level --
// come from two 256-entry tables in LDS, filled with the very expressions above (every pixel gets the bits it got before):
// a byte extract and an LDS read instead of convert + add + v_log_f32 (quarter rate) + fma per channel, pixel and pass.  The
// uint8 passes are bound by their vector instructions, not by memory.  Pixel values then travel as integer BITS in the float
// arrays of the loops (load_pixels<..., kBits = true>).
template <typename T> struct LevelTables {
    __device__ __forceinline__ void fill() {}
};
template <> struct LevelTables<uint8_t> {
    float od[256], l2[256];
    __device__ __forceinline__ void fill() {
        for (int t = threadIdx.x; t < 256; t += blockDim.x) {
            od[t] = optical_density<uint8_t>((float)t);
            l2[t] = log2_level<uint8_t>((float)t);
        }
        __syncthreads();
    }
};
// 8-bit CODES of float tiles (round 4).  Image pixels are 8-bit grey levels: a float tile made from a decoded image (u8 / 255: what
// ToDtype(float32, scale=True), the reference's benchmarks and bench.py produce) holds one of 256 values per channel.  The four-pass
// form's first pass (stats_item<..., kEmit>) checks that for every element -- x is code k iff its bits are those of F(k / 255) -- and
// leaves the tile as bytes in the workspace; a tile that passes is read as those bytes by the three later passes (a quarter of the
// bytes, and the table lookups of uint8 tiles instead of a v_log_f32 per channel), a tile that does not is read as it was.  The codes
// are a lossless copy and the tables hold the very expressions the float path evaluates, so every pixel gets the bits it got
// before.  Coded<F> is the element type of such a copy: byte storage (every `sizeof(T) == 1` path applies), F's meaning.
template <typename F> struct Coded {
    uint8_t v;
    __device__ __forceinline__ operator uint32_t() const { return v; }
};
template <typename T> struct Sem { using type = T; };                  // what an element MEANS (output rules), as opposed to how it is stored
template <typename F> struct Sem<Coded<F>> { using type = F; };
template <typename F> __device__ __forceinline__ float code_value(int k) {      // the float a tile of F holds for grey level k: float(k) / 255 (IEEE), rounded to F
    return Elem<F>::load(Elem<F>::store(div255_of_level((float)k)));
}
template <typename F> struct LevelTables<Coded<F>> {
    float od[256], l2[256];
    __device__ __forceinline__ void fill() {
        for (int t = threadIdx.x; t < 256; t += blockDim.x) {
            const float v = code_value<F>(t);
            od[t] = optical_density<F>(v);
            l2[t] = log2_level<F>(v);
        }
        __syncthreads();
    }
};
// The first pass's table: {the bits of code k's float, its optical density}, kCodeCopies bank-striped copies (a lane reads copy
// lane % kCodeCopies: with one copy half of the LDS pipe's cycles were bank conflicts in the tile-resident kernel).
// code_pack(): a pack's 3 x V elements -> their codes (one 32-bit word per plane, stored to the tile's code planes), their optical
// densities, and whether any element of the wave's packs is not a grey level (then: the densities by the expression, the tile marked).
template <typename F, int kCodeCopies> struct CodeTable {      // (4 copies = 8 KB in the moments pass; 2 in pass A of the two-pass form, whose four workgroups per CU leave 4 KB each)
    static constexpr int copies = kCodeCopies;
    uint2 e[256 * kCodeCopies];
    __device__ __forceinline__ void fill() {
        for (int t = threadIdx.x; t < 256; t += blockDim.x) {
            const float v = code_value<F>(t);
            const uint2 entry = make_uint2(__float_as_uint(v), __float_as_uint(optical_density<F>(v)));
#pragma unroll
            for (int c = 0; c < kCodeCopies; ++c) e[t * kCodeCopies + c] = entry;
        }
        __syncthreads();
    }
};
// which streaming instantiations carry the coded variant
// Planar float32 tiles in 16-byte packs.  (bfloat16 / float16 tiles were built and measured too -- code_quad() and store_codes() serve their
// packs of eight -- and do not pay: their passes are bound by instructions and per-item latency, not by bytes.  256 x 224 x 224 bf16:
// moments pass 23.8 -> 34.4 us, bracket passes 23.6 + 21.9 -> 24.3 + 22.3 us, reconstruct 26.1 -> 22.0 us; the call 127 -> 140 us.)
template <typename T, int V, bool kInter> struct Codable { static constexpr bool value = std::is_same<T, float>::value && V == 4 && !kInter; };

template <typename T> __device__ __forceinline__ float od_of(float v, const LevelTables<T>& tb) {
    if constexpr (sizeof(T) == 1) return tb.od[__float_as_uint(v)]; else return optical_density<T>(v);
}
template <typename T> __device__ __forceinline__ float l2_of(float v, const LevelTables<T>& tb) {
    if constexpr (sizeof(T) == 1) return tb.l2[__float_as_uint(v)]; else return log2_level<T>(v);
}

__device__ __forceinline__ bool od_selected(const float od[3], bool use_all) {
    return use_all || (fminf(od[0], fminf(od[1], od[2])) >= kBeta);    // torch_backend.py:404-405
}

// Angle of the plane projection as an order-preserving key.  The reference ranks phi = atan2(t1, t0)
// (torch_backend.py:418); ranking needs only a monotone function of phi, so pixels carry the "diamond angle"
//   r = t1 / (|t0| + |t1|);   t0 >= 0: r in [-1,1];   t0 < 0, t1 >= 0: 2 - r in (1,2];   t0 < 0, t1 < 0: -2 - r in [-2,-1)
// (one v_rcp_f32 instead of a ~40-instruction atan2f).  The two selected keys per tile are turned back into
// phi by angle_from_key() in double precision; the difference from atan2f of the same pixel is ~1e-7 rad.
__device__ __forceinline__ float diamond_angle(float t1, float t0) {
    const float den = fmaxf(fabsf(t0) + fabsf(t1), 1e-30f);      // t0 = t1 = 0 -> r = 0
    const float r = t1 * __builtin_amdgcn_rcpf(den);               // bare v_rcp_f32 (1 ulp): the key only has to be the same everywhere
    return t0 >= 0.0f ? r : (t1 >= 0.0f ? 2.0f - r : -2.0f - r);
}

__device__ __forceinline__ uint32_t angle_key(const float od[3], const float* __restrict__ v) {
    const float t0 = fmaf(od[2], v[4], fmaf(od[1], v[2], od[0] * v[0]));   // That[:,0]  (:417)
    const float t1 = fmaf(od[2], v[5], fmaf(od[1], v[3], od[0] * v[1]));   // That[:,1]
    return float_key(diamond_angle(t1, t0));
}

__device__ inline float angle_from_key(uint32_t key) {
    const double d = (double)key_float(key);
    double x, y;
    if (d > 1.0) {            // second quadrant: d = 2 - r
        y = 2.0 - d;
        x = -(1.0 - y);
    } else if (d < -1.0) {    // third quadrant: d = -2 - r, r in (-1,0]
        y = -2.0 - d;
        x = -(1.0 + y);
    } else {
        y = d;
        x = 1.0 - fabs(d);
    }
    return (float)atan2(y, x);
}

__device__ __forceinline__ void concentration(const float od[3], const float* __restrict__ pinv, float& c0, float& c1) {
    c0 = fmaf(od[2], pinv[2], fmaf(od[1], pinv[1], od[0] * pinv[0]));     // :444
    c1 = fmaf(od[2], pinv[5], fmaf(od[1], pinv[4], od[0] * pinv[3]));
}

// The V pixels starting at pixel p of a tile, as raw channel values u[c][i]; the layout is a compile-time choice (a run-time
// branch here costs the planar path dearly: uint8 146 -> 259 us/call measured).  Planar tiles: one 16-byte pack per plane.
// Interleaved tiles (H,W,3): the 3V values lie side by side -- three packs, de-interleaved in registers (for free: the
// indices are compile-time constants).
// kBits (uint8 only): the grey levels arrive as integer bits in the floats (for the LDS tables: od_of / l2_of), not converted.
template <typename T, int V, bool kBits>
__device__ __forceinline__ void load_values(const T* __restrict__ p, float (&out)[V]) {
    if constexpr (kBits && sizeof(T) == 1) {
        if constexpr (V == 1) {
            out[0] = __uint_as_float((uint32_t)p[0]);
        } else {
            const Pack<T, V> pk = *reinterpret_cast<const Pack<T, V>*>(p);
#pragma unroll
            for (int i = 0; i < V; ++i) out[i] = __uint_as_float((uint32_t)pk.v[i]);
        }
    } else {
        load_raw<T, V>(p, out);
    }
}

template <typename T, int V, bool kInter, bool kBits = false>
__device__ __forceinline__ void load_pixels(const T* __restrict__ img, int64_t pixels, int64_t p, float (&u)[3][V]) {
    if constexpr (kInter) {
        float flat[3][V];
#pragma unroll
        for (int k = 0; k < 3; ++k) load_values<T, V, kBits>(img + 3 * p + k * V, flat[k]);
#pragma unroll
        for (int i = 0; i < V; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) u[c][i] = flat[(3 * i + c) / V][(3 * i + c) % V];
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) load_values<T, V, kBits>(img + c * pixels + p, u[c]);
    }
}

// The V pixels a thread holds, as the three packs arrived (16 bytes each on the vector paths): a pixel's channel is taken out
// of them when its turn comes.  Held as floats -- 3 V registers for the current sweep and 3 V for the one in flight -- uint8
// tiles (V = 16) made pass A a 204-register kernel (two waves per SIMD, half its workgroups waiting for a second round), and in
// every loop that requests the next sweep's pixels ahead the conversion of 16- and 8-bit elements was done at once, i.e. the
// loop WAITED for the load it had just issued (bracket passes of bf16 / f16 / uint8 tiles).
template <typename T, int V, bool kInter>
struct PixelPacks {
    static constexpr int kWords = (int)(sizeof(T) * V + 3) / 4;      // 4 on the vector paths (16-byte packs)
    uint32_t w[3][kWords];      // (32-bit words, not elements: a struct of sixteen bytes is taken apart into sixteen registers the moment it is loaded)
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int i = 0; i < kWords; ++i) w[k][i] = 0u;
    }
    __device__ __forceinline__ void load(const T* __restrict__ img, int64_t pixels, int64_t p) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const T* src = kInter ? img + 3 * p + k * V : img + k * pixels + p;
            if constexpr (sizeof(T) * V == 16) {
                const uint4 q = *reinterpret_cast<const uint4*>(src);
                w[k][0] = q.x; w[k][1] = q.y; w[k][2] = q.z; w[k][3] = q.w;
            } else if constexpr (sizeof(T) * V == 8) {
                const uint2 q = *reinterpret_cast<const uint2*>(src);
                w[k][0] = q.x; w[k][1] = q.y;
            } else {
                static_assert(V == 1 && sizeof(T) <= 4, "single elements of at most four bytes");
                T v = src[0];
                uint32_t bits = 0;
                __builtin_memcpy(&bits, &v, sizeof(T));
                w[k][0] = bits;
            }
        }
    }
    // channel c of pixel i, as load_pixels<T, V, kInter, sizeof(T) == 1> hands it over (uint8: the grey level's integer bits)
    __device__ __forceinline__ float value(int c, int i) const {
        const int idx = kInter ? 3 * i + c : c * V + i, k = idx / V, e = idx % V;
        if constexpr (sizeof(T) == 1) {
            return __uint_as_float((w[k][e / 4] >> (8 * (e % 4))) & 0xFFu);
        } else if constexpr (sizeof(T) == 2) {
            const uint16_t bits = (uint16_t)(w[k][e / 2] >> (16 * (e % 2)));
            T v;
            __builtin_memcpy(&v, &bits, 2);
            return raw_value<T>(v);
        } else if constexpr (sizeof(T) == 4) {
            T v;
            __builtin_memcpy(&v, &w[k][e], 4);
            return raw_value<T>(v);
        } else {
            T v;
            __builtin_memcpy(&v, &w[k][2 * e], 8);
            return raw_value<T>(v);
        }
    }
};

template <int W>      // W 32-bit words of codes per plane: the V = 4 W pixels of a pack
__device__ __forceinline__ void store_codes(const Geometry& g, const Workspace& ws, int64_t tile, int64_t p, const uint32_t (&word)[3][W]) {
    static_assert(W == 1 || W == 2, "four or eight codes per plane");
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint8_t* dst = ws.codes + ((size_t)tile * 3 + c) * g.pixels + p;
        // (plain stores: the later passes find the codes in the L2s / the Infinity Cache -- non-temporal ones made pass A 54 -> 55.6 us and the reconstruct pass 37.4 -> 41.3)
        if constexpr (W == 1) *reinterpret_cast<uint32_t*>(dst) = word[c][0]; else *reinterpret_cast<uint2*>(dst) = make_uint2(word[c][0], word[c][1]);
    }
}
// Four pixels of a pack (i0 ... i0 + 3: one 32-bit word of codes per plane) at a time: twelve optical densities live, not 3 V (a pack of
// eight bf16 / f16 pixels at once spilled the moments pass and pass A to scratch).
template <typename T, int V, bool kInter, class Table>
__device__ __forceinline__ void code_quad(const PixelPacks<T, V, kInter>& u, int i0, const LevelTables<T>& tb, const Table& ct, const Geometry& g, const Workspace& ws, int64_t tile, float (&od_quad)[4][3], uint32_t (&word)[3][V / 4]) {
    static_assert((V == 4 || V == 8) && !kInter, "a pack's codes are one or two 32-bit words of a plane");
    constexpr int kCodeCopies = Table::copies;
    uint32_t differ = 0u;
    const int copy = (int)(threadIdx.x % kCodeCopies);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        uint32_t w = 0u;
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            const float x = u.value(c, i0 + ii);
            // nearest grey level: 255 x + 2^23 has it in the low bits of its mantissa (round to nearest even; an x outside [0, 1] or a
            // NaN leaves some other byte there, and the comparison below fails).  bfloat16 tiles: x = k / 255 rounded to eight bits is off
            // by at most k 2^-9 < 0.5 after the multiplication -- still level k, and 256 different values (1 / 255 exceeds the format's
            // spacing below 1)
            const uint32_t k = __float_as_uint(fmaf(x, 255.0f, 8388608.0f)) & 0xFFu;
            const uint2 entry = ct.e[k * kCodeCopies + copy];
            differ |= entry.x ^ __float_as_uint(x);
            od_quad[ii][c] = __uint_as_float(entry.y);
            w |= k << (8 * ii);
        }
        word[c][i0 / 4] = w;
    }
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(differ != 0u) != 0ull, 0)) {      // some element of the wave's quads is not an 8-bit level: this tile stays float
#pragma unroll
        for (int ii = 0; ii < 4; ++ii)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float x = u.value(c, i0 + ii);
                asm volatile("" : "+v"(x));      // (a real branch: without this the logarithms are computed for every pack and selected away)
                od_quad[ii][c] = od_of<T>(x, tb);
            }
        if (differ != 0u) put(&ws.code_bad[tile], g.code_epoch);
    }
}


template <typename O, int V, bool kInter>
__device__ __forceinline__ void store_pixels(O* __restrict__ dst, int64_t pixels, int64_t p, const O (&res)[3][V]) {
    if constexpr (kInter) {
        O flat[3][V];
#pragma unroll
        for (int i = 0; i < V; ++i)
#pragma unroll
            for (int c = 0; c < 3; ++c) flat[(3 * i + c) / V][(3 * i + c) % V] = res[c][i];
#pragma unroll
        for (int k = 0; k < 3; ++k) store_pack_stream<O, V>(dst + 3 * p + k * V, flat[k]);
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) store_pack_stream<O, V>(dst + c * pixels + p, res[c]);
    }
}

// Interleaved tiles, 16-byte packs: a lane's three packs are 48 consecutive bytes of the output, so a plain store instruction
// of the wave touches 64 different 48-byte pieces.  Staged through 3 KB of LDS per wave, every store instruction writes 1 KB of
// consecutive bytes instead (lane L, instruction s: bytes [1024 s + 16 L, +16) of the wave's 3 KB).  All 64 lanes take part.
template <typename O, int V>
__device__ __forceinline__ void store_pixels_staged(O* __restrict__ dst, int64_t p, const O (&res)[3][V], uint4* __restrict__ stage) {
    static_assert(sizeof(O) * V == 16, "16-byte packs");
    O flat[3][V];
#pragma unroll
    for (int i = 0; i < V; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) flat[(3 * i + c) / V][(3 * i + c) % V] = res[c][i];
    const uint32_t lane = lane_id();
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        uint4 v;
        __builtin_memcpy(&v, flat[k], 16);
        stage[lane * 3 + k] = v;
    }
    __builtin_amdgcn_wave_barrier();
    char* base = reinterpret_cast<char*>(dst + 3 * (p - (int64_t)lane * V));      // the wave's first byte
    typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int s2 = 0; s2 < 3; ++s2) {
        const uint4 v = stage[s2 * kWave + lane];
        f4 f;
        __builtin_memcpy(&f, &v, 16);
        __builtin_nontemporal_store(f, reinterpret_cast<f4*>(base + ((size_t)s2 * kWave + lane) * 16));
    }
    __builtin_amdgcn_wave_barrier();      // the next iteration overwrites the stage
}

template <typename T>
__device__ __forceinline__ void load_od_scalar(const T* __restrict__ images, const Geometry& g, int64_t tile, int64_t p, float od[3]) {
    const T* base = images + tile * 3 * g.pixels + (g.interleaved ? 3 * p : p);
    const int64_t step = g.interleaved ? 1 : g.pixels;
#pragma unroll
    for (int c = 0; c < 3; ++c) od[c] = optical_density<T>(raw_value<T>(base[c * step]));
}

// ------------------------------------------------------------------------------------------------
// small numerics of the per-tile stages
// ------------------------------------------------------------------------------------------------
// One Jacobi rotation in the (p,q) plane of a symmetric 3x3 kept in scalars (r is the third index).  The angle is
// worked out in fp32 (hardware rcp / sqrt / rsq: a chain of software fp64 divisions and roots costs ~1 us per
// rotation on one lane), then (c,s) is renormalised in fp64 so the transform stays orthogonal to 1e-14; the update
// is the full similarity transform, exact for any orthonormal (c,s), so an angle that is only fp32-accurate just
// leaves a 1e-7-times smaller off-diagonal for the next sweep.
#define SX_JACOBI_ROTATE(app, aqq, apq, arp, arq, v0p, v0q, v1p, v1q, v2p, v2q)            \
    if ((apq) != 0.0) {                                                                      \
        const float theta = (float)((aqq) - (app)) * __builtin_amdgcn_rcpf((float)(2.0 * (apq))); \
        const float tf = copysignf(1.0f, theta) * __builtin_amdgcn_rcpf(fabsf(theta) + __builtin_amdgcn_sqrtf(fmaf(theta, theta, 1.0f))); \
        const float cf = __builtin_amdgcn_rsqf(fmaf(tf, tf, 1.0f));                          \
        double c = (double)cf, sn = (double)(tf * cf);                                       \
        const double fix = 1.5 - 0.5 * (c * c + sn * sn);                                    \
        c *= fix;                                                                            \
        sn *= fix;                                                                           \
        const double cc = c * c, ss = sn * sn, cs = c * sn;                                  \
        const double pp = (app), qq = (aqq), pq = (apq);                                     \
        (app) = cc * pp - 2.0 * cs * pq + ss * qq;                                           \
        (aqq) = ss * pp + 2.0 * cs * pq + cc * qq;                                           \
        (apq) = (cc - ss) * pq + cs * (pp - qq);                                             \
        const double rp = (arp), rq = (arq);                                                 \
        (arp) = c * rp - sn * rq;                                                            \
        (arq) = sn * rp + c * rq;                                                            \
        double xp = (v0p), xq = (v0q);                                                       \
        (v0p) = c * xp - sn * xq;                                                            \
        (v0q) = sn * xp + c * xq;                                                            \
        xp = (v1p), xq = (v1q);                                                              \
        (v1p) = c * xp - sn * xq;                                                            \
        (v1q) = sn * xp + c * xq;                                                            \
        xp = (v2p), xq = (v2q);                                                              \
        (v2p) = c * xp - sn * xq;                                                            \
        (v2q) = sn * xp + c * xq;                                                            \
    }

// Eigen-decomposition of a symmetric 3x3 by cyclic Jacobi rotations (fp64, everything in registers).
// Eigenvalues ascending in w[], matching eigenvectors in the columns of q (row-major 3x3).
__device__ void jacobi_eigh3(const double a_in[9], double w[3], double q[9]) {
    double a00 = a_in[0], a01 = a_in[1], a02 = a_in[2], a11 = a_in[4], a12 = a_in[5], a22 = a_in[8];
    double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = a01 * a01 + a02 * a02 + a12 * a12;
        const double diag = a00 * a00 + a11 * a11 + a22 * a22;
        if (off <= 1e-30 * diag || off == 0.0) break;      // |off|/|diag| <= 1e-15
        SX_JACOBI_ROTATE(a00, a11, a01, a02, a12, v00, v01, v10, v11, v20, v21)   // (0,1), r = 2
        SX_JACOBI_ROTATE(a00, a22, a02, a01, a12, v00, v02, v10, v12, v20, v22)   // (0,2), r = 1
        SX_JACOBI_ROTATE(a11, a22, a12, a01, a02, v01, v02, v11, v12, v21, v22)   // (1,2), r = 0
    }
    double d0 = a00, d1 = a11, d2 = a22;
    // stable ascending sort of (eigenvalue, column): bubble network with strict comparisons
#define SX_SWAP_COLS(da, db, xa0, xa1, xa2, xb0, xb1, xb2) \
    if ((da) > (db)) {                                     \
        double t_ = (da); (da) = (db); (db) = t_;          \
        t_ = (xa0); (xa0) = (xb0); (xb0) = t_;             \
        t_ = (xa1); (xa1) = (xb1); (xb1) = t_;             \
        t_ = (xa2); (xa2) = (xb2); (xb2) = t_;             \
    }
    SX_SWAP_COLS(d0, d1, v00, v10, v20, v01, v11, v21)
    SX_SWAP_COLS(d1, d2, v01, v11, v21, v02, v12, v22)
    SX_SWAP_COLS(d0, d1, v00, v10, v20, v01, v11, v21)
#undef SX_SWAP_COLS
    w[0] = d0; w[1] = d1; w[2] = d2;
    q[0] = v00; q[1] = v01; q[2] = v02;
    q[3] = v10; q[4] = v11; q[5] = v12;
    q[6] = v20; q[7] = v21; q[8] = v22;
}
#undef SX_JACOBI_ROTATE

// The two eigenvectors the path needs -- middle and largest eigenvalue -- without the full Jacobi iteration when the
// spectrum is well separated (every stained tile: lambda_1 << lambda_2 << lambda_3).  Repeated squaring of a symmetric
// matrix makes its dominant eigenvector take over every column: A^32 gives the largest one of the covariance, adj(A)^32
// (adj(A) = det(A) A^-1 has the same eigenvectors with the order reversed) the smallest, the middle one is their cross
// product.  ~250 dependent fp64 operations instead of ~1000 (3 us -> 0.8 us on the one lane that runs this).  The result
// is accepted only if both residuals |A v - (v'Av) v| are at rounding level; otherwise -- close eigenvalues, degenerate
// covariance -- the caller runs the Jacobi iteration.  Columns of q as jacobi_eigh3 returns them (ascending eigenvalue).
// kPair: lanes 0 and 1 of a wave call this together with the same matrix; the two power iterations (the same code on different
// data) run side by side, one per lane, and are exchanged; both lanes return the same result.
template <bool kPair = false>
__device__ inline bool plane_basis_by_squaring(const double a[9], double q[9]) {
    const double trace = a[0] + a[4] + a[8];
    if (!(trace > 0.0)) return false;
    auto dominant = [](double m00, double m01, double m02, double m11, double m12, double m22, double (&v)[3]) {
        for (int it = 0; it < 5; ++it) {      // M <- M^2 / max|M|, five times: M^32
            const double s00 = m00 * m00 + m01 * m01 + m02 * m02, s01 = m00 * m01 + m01 * m11 + m02 * m12, s02 = m00 * m02 + m01 * m12 + m02 * m22;
            const double s11 = m01 * m01 + m11 * m11 + m12 * m12, s12 = m01 * m02 + m11 * m12 + m12 * m22, s22 = m02 * m02 + m12 * m12 + m22 * m22;
            const double big = fmax(fmax(s00, s11), s22);      // the diagonal of a square dominates its rows
            if (!(big > 0.0)) return false;
            const double inv = (double)__builtin_amdgcn_rcpf((float)big);      // any scale near 1/big will do
            m00 = s00 * inv; m01 = s01 * inv; m02 = s02 * inv; m11 = s11 * inv; m12 = s12 * inv; m22 = s22 * inv;
        }
        // the column with the largest diagonal entry, normalised
        double x, y, z;
        if (m00 >= m11 && m00 >= m22) { x = m00; y = m01; z = m02; }
        else if (m11 >= m22) { x = m01; y = m11; z = m12; }
        else { x = m02; y = m12; z = m22; }
        const double n2 = x * x + y * y + z * z;
        if (!(n2 > 0.0)) return false;
        double r = (double)__builtin_amdgcn_rsqf((float)n2);
        r = r * (1.5 - 0.5 * n2 * r * r);
        r = r * (1.5 - 0.5 * n2 * r * r);
        v[0] = x * r; v[1] = y * r; v[2] = z * r;
        return true;
    };
    const double s = 1.0 / trace;      // scale the matrix to trace 1
    const double a00 = a[0] * s, a01 = a[1] * s, a02 = a[2] * s, a11 = a[4] * s, a12 = a[5] * s, a22 = a[8] * s;
    double v_max[3], v_min[3];
    // the smallest eigenvector is the dominant one of the adjugate (cofactors) of the scaled matrix
    const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11, c11 = a00 * a22 - a02 * a02, c12 = a01 * a02 - a00 * a12,
                 c22 = a00 * a11 - a01 * a01;
    if constexpr (kPair) {
        const bool second = (lane_id() & 1u) != 0;
        double mine[3], other[3];
        const bool ok_mine = dominant(second ? c00 : a00, second ? c01 : a01, second ? c02 : a02, second ? c11 : a11, second ? c12 : a12, second ? c22 : a22, mine);
        const int partner = (int)(lane_id() ^ 1u);
        const bool ok_other = __shfl(ok_mine ? 1 : 0, partner, kWave) != 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) other[i] = __shfl(ok_mine ? mine[i] : 0.0, partner, kWave);
        if (!(ok_mine && ok_other)) return false;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            v_max[i] = second ? other[i] : mine[i];
            v_min[i] = second ? mine[i] : other[i];
        }
    } else {
        if (!dominant(a00, a01, a02, a11, a12, a22, v_max)) return false;
        if (!dominant(c00, c01, c02, c11, c12, c22, v_min)) return false;
    }
    // the middle eigenvector: orthogonal to both; then v_max once more against (v_min, v_mid) so that the three are
    // orthonormal to rounding
    double mx = v_min[1] * v_max[2] - v_min[2] * v_max[1], my = v_min[2] * v_max[0] - v_min[0] * v_max[2], mz = v_min[0] * v_max[1] - v_min[1] * v_max[0];
    const double mn2 = mx * mx + my * my + mz * mz;
    if (!(mn2 > 0.25)) return false;      // v_min and v_max far from orthogonal: not converged
    double r = (double)__builtin_amdgcn_rsqf((float)mn2);
    r = r * (1.5 - 0.5 * mn2 * r * r);
    r = r * (1.5 - 0.5 * mn2 * r * r);
    mx *= r; my *= r; mz *= r;
    auto residual = [&](double x, double y, double z) {
        const double ax = a00 * x + a01 * y + a02 * z, ay = a01 * x + a11 * y + a12 * z, az = a02 * x + a12 * y + a22 * z;
        const double lam = ax * x + ay * y + az * z;
        const double rx = ax - lam * x, ry = ay - lam * y, rz = az - lam * z;
        return rx * rx + ry * ry + rz * rz;
    };
    // (matrix scaled to trace 1: residual norms below 1e-14 mean eigenvectors good to ~1e-14 / gap)
    if (residual(v_max[0], v_max[1], v_max[2]) > 1e-28 || residual(mx, my, mz) > 1e-28 || residual(v_min[0], v_min[1], v_min[2]) > 1e-28) return false;
    q[0] = v_min[0]; q[3] = v_min[1]; q[6] = v_min[2];
    q[1] = mx; q[4] = my; q[7] = mz;
    q[2] = v_max[0]; q[5] = v_max[1]; q[8] = v_max[2];
    return true;
}

// Raw moments -> unbiased covariance (torch_backend.py:395-397) -> plane vectors, columns [1,2] of eigh
// (torch_backend.py:415), sign convention: positive component sum.  mom[0..9] masked set, mom[10..19] all pixels;
// fewer than 3 masked pixels -> all pixels when allow_fallback (torch_backend.py:409-410).
template <bool kPair = false>
__device__ void plane_from_moments(const double* mom, bool allow_fallback, double cov[9], float vecs[6], bool& use_all, unsigned long long& n_sel) {
    use_all = allow_fallback && mom[0] < 3.0;
    const double* a = use_all ? mom + 10 : mom;
    const double cnt = a[0];
    if (cnt > 1.0) {
        const double inv_n = 1.0 / cnt, inv_d = 1.0 / (cnt - 1.0);      // two fp64 divisions (software, ~40 instructions each) instead of nine
        const double m0 = a[1] * inv_n, m1 = a[2] * inv_n, m2 = a[3] * inv_n;
        cov[0] = (a[4] - a[1] * m0) * inv_d;
        cov[1] = cov[3] = (a[5] - a[1] * m1) * inv_d;
        cov[2] = cov[6] = (a[6] - a[1] * m2) * inv_d;
        cov[4] = (a[7] - a[2] * m1) * inv_d;
        cov[5] = cov[7] = (a[8] - a[2] * m2) * inv_d;
        cov[8] = (a[9] - a[3] * m2) * inv_d;
    } else {
#pragma unroll
        for (int i = 0; i < 9; ++i) cov[i] = 0.0;
    }
    double w[3], q[9];
    if (!plane_basis_by_squaring<kPair>(cov, q)) jacobi_eigh3(cov, w, q);
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const int src = c + 1;
        const double sum = q[0 * 3 + src] + q[1 * 3 + src] + q[2 * 3 + src];
        const double sgn = sum < 0.0 ? -1.0 : 1.0;
#pragma unroll
        for (int r = 0; r < 3; ++r) vecs[r * 2 + c] = (float)(sgn * q[r * 3 + src]);
    }
    n_sel = (unsigned long long)cnt;
}

// (cos phi, sin phi) of the angle a diamond key stands for, without going through phi: the point (x, y) on the
// diamond |x| + |y| = 1 the key encodes, normalised (fp64; the reciprocal root is a v_rsq_f32 seed + one Newton
// step).  Differs from cosf/sinf of the fp32 angle (torch_backend.py:427-430) by ~1e-7.
__device__ __forceinline__ void direction_from_key(uint32_t key, float& c, float& s) {
    const double d = (double)key_float(key);
    double x, y;
    if (d > 1.0) {            // second quadrant: d = 2 - r
        y = 2.0 - d;
        x = -(1.0 - y);
    } else if (d < -1.0) {    // third quadrant: d = -2 - r, r in (-1,0]
        y = -2.0 - d;
        x = -(1.0 + y);
    } else {
        y = d;
        x = 1.0 - fabs(d);
    }
    const double n2 = x * x + y * y;                      // in [0.5, 1]
    double inv = (double)__builtin_amdgcn_rsqf((float)n2);
    inv = inv * (1.5 - 0.5 * n2 * inv * inv);
    inv = inv * (1.5 - 0.5 * n2 * inv * inv);
    c = (float)(x * inv);
    s = (float)(y * inv);
}

// Angle percentiles (as keys) -> extreme stain vectors -> HE_source (H before E) -> its (2,3) pseudo-inverse.
__device__ void stain_vectors_and_pinv(const float* vecs, uint32_t key_lo, uint32_t key_hi, float* he_out, float* pinv_out) {
    float cl, sl, ch, sh;
    direction_from_key(key_lo, cl, sl);
    direction_from_key(key_hi, ch, sh);
    float vmin[3], vmax[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        vmin[r] = fmaf(vecs[r * 2 + 1], sl, vecs[r * 2] * cl);                                   // :436
        vmax[r] = fmaf(vecs[r * 2 + 1], sh, vecs[r * 2] * ch);                                   // :437
    }
    const bool min_first = vmin[0] > vmax[0];                                                    // :439
    float he[6];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        he[r * 2] = min_first ? vmin[r] : vmax[r];
        he[r * 2 + 1] = min_first ? vmax[r] : vmin[r];
    }
    // pseudo-inverse (2,3) of HE (3,2) in fp64 through the eigen-decomposition of HE^T HE, dropping a
    // singular value below 3*eps_f32 of the largest (rank rule of lstsq(rcond=None), torch_backend.py:379)
    const double a = (double)he[0] * he[0] + (double)he[2] * he[2] + (double)he[4] * he[4];
    const double b = (double)he[0] * he[1] + (double)he[2] * he[3] + (double)he[4] * he[5];
    const double d = (double)he[1] * he[1] + (double)he[3] * he[3] + (double)he[5] * he[5];
    // (hardware v_rcp_f64 / v_rsq_f64 seeds + two Newton steps each, ~1e-15 relative: the software fp64 division and square
    // root cost ~40 instructions apiece on the one lane that runs this)
    auto fast_rcp = [](double x) { double r = __builtin_amdgcn_rcp(x); r = r * (2.0 - x * r); return r * (2.0 - x * r); };
    auto fast_rsqrt = [](double x) { double r = __builtin_amdgcn_rsq(x); r = r * (1.5 - 0.5 * x * r * r); return r * (1.5 - 0.5 * x * r * r); };
    const double tr = a + d, df = a - d;
    const double disc2 = df * df + 4.0 * b * b;
    const double disc = disc2 > 0.0 ? disc2 * fast_rsqrt(disc2) : 0.0;
    const double l1 = 0.5 * (tr + disc), l2 = 0.5 * (tr - disc);
    double e1x, e1y;                                   // unit eigenvector of l1
    if (fabs(b) > 0.0) {
        e1x = l1 - d;
        e1y = b;
    } else if (a >= d) {
        e1x = 1.0;
        e1y = 0.0;
    } else {
        e1x = 0.0;
        e1y = 1.0;
    }
    const double inv_nrm = fast_rsqrt(e1x * e1x + e1y * e1y);
    e1x *= inv_nrm;
    e1y *= inv_nrm;
    const double e2x = -e1y, e2y = e1x;
    const double rc = 3.0 * 1.1920928955078125e-07;
    const double i1 = l1 > 0.0 ? fast_rcp(l1) : 0.0;
    const double i2 = (l2 > 0.0 && l2 > rc * rc * l1) ? fast_rcp(l2) : 0.0;      // sigma_2 > rc * sigma_1, squared
    // (HE^T HE)^+ = i1 e1 e1^T + i2 e2 e2^T
    const double g00 = i1 * e1x * e1x + i2 * e2x * e2x, g01 = i1 * e1x * e1y + i2 * e2x * e2y, g11 = i1 * e1y * e1y + i2 * e2y * e2y;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        pinv_out[c] = (float)(g00 * he[c * 2] + g01 * he[c * 2 + 1]);
        pinv_out[3 + c] = (float)(g01 * he[c * 2] + g11 * he[c * 2 + 1]);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) he_out[i] = he[i];
}

// k = round(0.01*q*(n-1)), half to even, evaluated in double like the Python expression at
// torch_backend.py:364 (0-based rank).
__device__ __forceinline__ unsigned long long nearest_rank_index(double q, unsigned long long n) {
    if (n == 0) return 0;
    return (unsigned long long)rint((0.01 * q) * (double)(n - 1));
}

// Sample ranks bracketing the wanted rank: +-6 standard deviations of the rank a sample of m_valid keys
// gives to the k0-th of n_total keys, plus slack.  A miss is detected later and repaired.
__device__ __forceinline__ void bracket_ranks(int m_valid, unsigned long long n_total, unsigned long long k0, long long& lo_r, long long& hi_r) {
    // fp32 with the bare hardware reciprocal / square root: the bracket is an estimate (+-6 sigma + 3 ranks of slack, a
    // miss is detected and repaired), it only has to be the same wherever it is evaluated -- one wave per bracket side
    const float f = n_total > 1 ? (float)k0 * __builtin_amdgcn_rcpf((float)(n_total - 1)) : 0.0f;
    const float r = f * (float)(m_valid - 1);
    const float sd = __builtin_amdgcn_sqrtf(fmaxf((float)m_valid * f * (1.0f - f), 0.0f));
    lo_r = (long long)floorf(r - 6.0f * sd - 3.0f);
    hi_r = (long long)ceilf(r + 6.0f * sd + 3.0f);
}

// One wave turns a 256-bin histogram and a rank into (bin, rank inside that bin).
__device__ __forceinline__ void scan_pick(const uint32_t* hist, unsigned long long rank, uint32_t& digit, unsigned long long& rank_in_bin) {
    const int lane = (int)lane_id();
    const uint32_t h0 = hist[4 * lane], h1 = hist[4 * lane + 1], h2 = hist[4 * lane + 2], h3 = hist[4 * lane + 3];
    const unsigned long long mine = (unsigned long long)h0 + h1 + h2 + h3;
    unsigned long long incl = mine;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const unsigned long long up = __shfl_up(incl, off, kWave);
        if (lane >= off) incl += up;
    }
    const uint64_t over = __ballot(incl > rank);
    const int owner = over ? (__ffsll((long long)over) - 1) : (kWave - 1);
    const unsigned long long before = __shfl(incl - mine, owner, kWave);
    const uint32_t b0 = __shfl(h0, owner, kWave), b1 = __shfl(h1, owner, kWave), b2 = __shfl(h2, owner, kWave);
    unsigned long long r = rank - before;
    uint32_t d = 0;
    if (r >= b0) { r -= b0; d = 1; if (r >= b1) { r -= b1; d = 2; if (r >= b2) { r -= b2; d = 3; } } }
    digit = 4u * (uint32_t)owner + d;
    rank_in_bin = r;
}

// The same for totals below 2^32 (samples, candidates) on the DPP scan; the histogram is 16-byte aligned.
__device__ __forceinline__ void scan_pick32(const uint32_t* hist, uint32_t rank, uint32_t& digit, uint32_t& rank_in_bin) {
    const int lane = (int)lane_id();
    const uint4 h = *reinterpret_cast<const uint4*>(hist + 4 * lane);
    const uint32_t mine = h.x + h.y + h.z + h.w;
    const uint32_t incl = wave_scan_u32(mine);
    const uint64_t over = __ballot(incl > rank);
    const int owner = over ? (__ffsll((long long)over) - 1) : (kWave - 1);
    const uint32_t before = (uint32_t)__builtin_amdgcn_readlane((int)(incl - mine), owner);
    const uint32_t b0 = (uint32_t)__builtin_amdgcn_readlane((int)h.x, owner), b1 = (uint32_t)__builtin_amdgcn_readlane((int)h.y, owner),
                   b2 = (uint32_t)__builtin_amdgcn_readlane((int)h.z, owner);
    uint32_t r = rank - before, d = 0;
    if (r >= b0) { r -= b0; d = 1; if (r >= b1) { r -= b1; d = 2; if (r >= b2) { r -= b2; d = 3; } } }
    digit = 4u * (uint32_t)owner + d;
    rank_in_bin = r;
}

// Exact element of 0-based rank `want` in list[0..n), by `team` consecutive threads (a multiple of 4) starting at
// local index t: four lanes share one element e, each counts a quarter of the list (all lanes of a quad read the
// same LDS words: broadcast), the quad adds up over DPP.  k is the answer iff #(x < k) <= want < #(x <= k); ties need
// no order, every thread that finds it writes the same key.
__device__ __forceinline__ void rank_pick(const uint32_t* list, uint32_t n, uint32_t want, uint32_t t, uint32_t team, uint32_t* result) {
    const uint32_t part = t & 3u, quarter = (n + 3u) / 4u;
    const uint32_t u_begin = part * quarter, u_end = min(u_begin + quarter, n);
    for (uint32_t e = t >> 2; e < n; e += team >> 2) {
        const uint32_t k = list[e];
        uint32_t lt = 0, le = 0;
#pragma unroll 4
        for (uint32_t u = u_begin; u < u_end; ++u) {
            const uint32_t x = list[u];
            lt += x < k ? 1u : 0u;
            le += x <= k ? 1u : 0u;
        }
        lt += dpp_move<0xB1, 0xF>(0u, lt);      // quad_perm [1,0,3,2]
        le += dpp_move<0xB1, 0xF>(0u, le);
        lt += dpp_move<0x4E, 0xF>(0u, lt);      // quad_perm [2,3,0,1]
        le += dpp_move<0x4E, 0xF>(0u, le);
        if (lt <= want && want < le) *result = k;
    }
}

// Bracket-relative bin (0..255) of a key, linear in the float VALUE the key stands for (key-linear bins would
// crowd: float keys spend one binade per exponent) and monotone in the key.  Values beyond the range
// (open brackets) go to the end bins; a degenerate range gets scale 0 (everything in bin 0 -> radix paths).
__device__ __forceinline__ uint32_t bin_of(uint32_t key, double origin, double scale) {
    // fp32: a subtraction of a constant and a multiplication by a non-negative constant are monotone under rounding, which is
    // all a bin function needs (the same function everywhere, non-decreasing in the key); the origin is a float to begin
    // with.  The fp64 form (seven half-rate instructions per key) made the candidate filter the longest phase of the stages.
    const float d = (key_float(key) - (float)origin) * (float)scale;
    return (uint32_t)fminf(fmaxf(d, 0.0f), 255.0f);
}
__device__ __forceinline__ double bin_origin_for(uint32_t lo) { return (double)key_float(lo); }
__device__ __forceinline__ double bin_scale_for(uint32_t lo, uint32_t hi) {
    const double span = (double)key_float(hi) - (double)key_float(lo);
    // bare v_rcp_f64 instead of a software division: any value works as long as everyone uses the SAME one -- inside
    // sample_brackets all threads evaluate this identically, for the candidates it travels in the stage record
    return (span > 0.0 && span < 1e300) ? 256.0 * __builtin_amdgcn_rcp(span) : 0.0;
}

// The keys bin b holds, as an inclusive range [first, last] (bin_of is monotone in the key): the edge value is
// inverted in fp64 and walked to the exact boundary with bin_of itself, so `first <= key <= last` is the same
// predicate as `bin_of(key) == b` at two integer comparisons per key instead of six fp64 operations.
__device__ __forceinline__ uint32_t bin_lower_edge(uint32_t bin, double origin, double scale, double inv_scale) {      // smallest key with bin_of(key) >= bin, bin in 1..255
    uint32_t k = float_key((float)(origin + (double)bin * inv_scale));
    while (k > 0u && bin_of(k - 1u, origin, scale) >= bin) --k;
    while (k < 0xFFFFFFFFu && bin_of(k, origin, scale) < bin) ++k;
    return k;
}
__device__ __forceinline__ void bin_key_range(uint32_t b, double origin, double scale, uint32_t& first, uint32_t& last) {
    if (!(scale > 0.0)) {          // degenerate range: everything is in bin 0
        first = b == 0 ? 0u : 1u;
        last = b == 0 ? 0xFFFFFFFFu : 0u;
        return;
    }
    const double inv_scale = __builtin_amdgcn_rcp(scale);      // seed only: the edges are walked to the exact boundary
    first = b == 0 ? 0u : bin_lower_edge(b, origin, scale, inv_scale);
    last = b >= 255u ? 0xFFFFFFFFu : bin_lower_edge(b + 1u, origin, scale, inv_scale) - 1u;
}

// LDS scratch of a per-tile stage (one workgroup).  The sample (_s) and candidate (_c) selections have their own
// histograms, lists and counters, so one reset at the top of the kernel serves both.
struct alignas(16) TileScratch {
    uint32_t keys[2][kSample];            // keys of the sample (read back only by the crowded-bin fallback)
    uint32_t hist_s[2][256];              // sample: one histogram per key set
    uint32_t hist_c[2][256];              // candidates: one histogram per slot of the pair
    uint32_t radix_hist[256];
    uint32_t list_s[4][kShortList];
    double mom[kMoments];
    double stage[64][kPartial];
    unsigned long long radix_rank, n_sel;
    uint32_t bin_s[4], rank_in_bin_s[4], count_s[4], result_s[4];
    uint32_t rank_s[4], beyond_s[4];      // sample rank of each bracket side (clamped into the sample) and whether it lay beyond the sample
    uint32_t range_c[2][2], rank_in_bin_c[2], count_c[2], result_c[2];
    uint32_t range_lo[2], range_hi[2], radix_digit;
    uint32_t valid;
    float coef[6];
    int flag;
};

__device__ __forceinline__ void reset_scratch(TileScratch* sh) {
    for (int t = threadIdx.x; t < 512; t += blockDim.x) {
        (&sh->hist_s[0][0])[t] = 0;
        (&sh->hist_c[0][0])[t] = 0;
    }
    if (threadIdx.x < 4) {
        sh->count_s[threadIdx.x] = 0;
        sh->result_s[threadIdx.x] = 0;
    }
    if (threadIdx.x < 2) {
        sh->count_c[threadIdx.x] = 0;
        sh->result_c[threadIdx.x] = 0;
        sh->range_lo[threadIdx.x] = 0xFFFFFFFFu;
        sh->range_hi[threadIdx.x] = 0u;
    }
    if (threadIdx.x == 0) sh->valid = 0;
}

// Exact rank-th smallest (0-based) of the valid keys produced by key_at(i), i in [0,count): four 8-bit radix
// rounds, keys recomputed/re-read in every round.  Slow path, whole workgroup.
// `keep` (optional, `count` words of global scratch this workgroup owns for the duration): the first round leaves every key there
// (an invalid entry: 0xFFFFFFFF, which no finite value's key is) and the other three read them back -- one load instead of
// recomputing a key from its pixel (three logarithms and ~60 instructions): the whole-tile select of a slot the two-pass form
// could not speculate on takes ~0.13 ms instead of ~0.4 ms.
// kThrough: the scratch is written THROUGH and read past the caches (agent-scope accesses).  Needed where the words are rewritten
// by another workgroup later in the SAME launch (the fused transform parks the keys in the tile's output, which the tile's
// reconstruct items then write from other XCDs: a plain store would leave a dirty line in this XCD's L2 whose write-back,
// whenever it came, would land on top of the result).
template <bool kThrough = false, class KeyAt, class Scratch>
__device__ uint32_t radix_select_stream(unsigned long long count, unsigned long long rank, KeyAt key_at, Scratch* sh, uint32_t* __restrict__ keep = nullptr) {
    uint32_t prefix = 0, mask = 0;
    __syncthreads();
    if (threadIdx.x == 0) sh->radix_rank = rank;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int t = threadIdx.x; t < 256; t += blockDim.x) sh->radix_hist[t] = 0;
        __syncthreads();
        // (a thread's consecutive keys mostly share their leading bytes -- one tile, one narrow range of angles or concentrations --,
        // and 64 lanes adding to one LDS word take their turns: a thread counts a run of equal digits itself and adds once per run)
        uint32_t last = 0xFFFFFFFFu, run = 0;
        auto count_digit = [&](uint32_t k) {
            const uint32_t d = (k >> shift) & 255u;
            if (d == last) {
                ++run;
            } else {
                if (run) atomicAdd(&sh->radix_hist[last], run);
                last = d;
                run = 1;
            }
        };
        // (four keys per trip: their loads -- of the parked keys, or of the pixels a key is made from -- go out together; one key per
        // trip made every round a chain of `count / blockDim` dependent memory round trips, 0.2 ms per round for a 512 x 512 tile)
        constexpr int kTrip = 4;
        const unsigned long long step = (unsigned long long)blockDim.x * kTrip;
        if (keep != nullptr && shift < 24) {
            for (unsigned long long i0 = threadIdx.x; i0 < count; i0 += step) {      // (written by this very thread in the first round: no fence needed)
                uint32_t k[kTrip];
#pragma unroll
                for (int u = 0; u < kTrip; ++u) {
                    const unsigned long long i = i0 + (unsigned long long)u * blockDim.x;
                    k[u] = i < count ? (kThrough ? __hip_atomic_load(&keep[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : keep[i]) : 0xFFFFFFFFu;
                }
#pragma unroll
                for (int u = 0; u < kTrip; ++u)
                    if (k[u] != 0xFFFFFFFFu && ((k[u] ^ prefix) & mask) == 0) count_digit(k[u]);
            }
        } else {
            for (unsigned long long i0 = threadIdx.x; i0 < count; i0 += step) {
                uint32_t k[kTrip];
                bool valid[kTrip];
#pragma unroll
                for (int u = 0; u < kTrip; ++u) {
                    const unsigned long long i = i0 + (unsigned long long)u * blockDim.x;
                    k[u] = 0;
                    valid[u] = i < count && key_at(i, k[u]);
                }
#pragma unroll
                for (int u = 0; u < kTrip; ++u) {
                    const unsigned long long i = i0 + (unsigned long long)u * blockDim.x;
                    if (keep != nullptr && i < count) {
                        if constexpr (kThrough) __hip_atomic_store(&keep[i], valid[u] ? k[u] : 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else keep[i] = valid[u] ? k[u] : 0xFFFFFFFFu;
                    }
                    if (valid[u] && ((k[u] ^ prefix) & mask) == 0) count_digit(k[u]);
                }
            }
        }
        if (run) atomicAdd(&sh->radix_hist[last], run);
        __syncthreads();
        if (threadIdx.x < kWave) {
            uint32_t d;
            unsigned long long rb;
            scan_pick(sh->radix_hist, sh->radix_rank, d, rb);
            if (lane_id() == 0) {
                sh->radix_digit = d;
                sh->radix_rank = rb;
            }
        }
        __syncthreads();
        prefix |= sh->radix_digit << shift;
        mask |= 0xFFu << shift;
        __syncthreads();
    }
    return prefix;
}

// Two brackets from the sample keys a per-tile stage holds in registers (key[set][i] belongs to sample
// i*kGroupThreads + threadIdx.x; invalid entries are 0xFFFFFFFF; the caller also left them in sh->keys).
// kSets == 1: both brackets (wanted ranks k0[0], k0[1]) are taken in key set 0 (the two angle percentiles);
// kSets == 2: bracket s is taken in key set s (the two concentrations).  Two levels: a 256-bin value-linear
// histogram per key set over its [min,max], one wave per wanted sample rank picks its bin, the keys of that bin
// are listed and the exact element found by rank counting (a crowded bin falls back to radix rounds).  Outputs
// per bracket: the bracket keys and the (origin, scale) of the bracket-relative bins used for the candidates.
// Needs reset_scratch() and a barrier before it; five barriers inside.
template <int kSets, bool kPoint = false>
__device__ void sample_brackets(TileScratch* sh, const uint32_t (&key)[kSets][kKeys], unsigned long long n_total, const unsigned long long (&k0)[2], uint32_t (&lo)[2],
                                uint32_t (&hi)[2], double (&bin_origin)[2], double (&bin_scale)[2]) {
    const uint32_t lane = lane_id();
    {
        uint32_t valid = 0;
#pragma unroll
        for (int set = 0; set < kSets; ++set) {
            uint32_t mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
            for (int i = 0; i < kKeys; ++i) {
                const uint32_t k = key[set][i];
                if (k != 0xFFFFFFFFu) {
                    mn = min(mn, k);
                    mx = max(mx, k);
                    if (set == 0) ++valid;
                }
            }
            mn = wave_min_u32(mn);
            mx = wave_max_u32(mx);
            if (lane == 0 && mn != 0xFFFFFFFFu) {
                atomicMin(&sh->range_lo[set], mn);
                atomicMax(&sh->range_hi[set], mx);
            }
        }
        valid = wave_total_u32(valid);      // both concentration sets hold every sample pixel, so set 0's count serves both
        if (lane == 0 && valid) atomicAdd(&sh->valid, valid);
    }
    __syncthreads();
    const int m_valid = (int)sh->valid;
    if (m_valid == 0) {      // nothing selected in the sample (uniform): open brackets, the count check sends the stage to the slow path
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            lo[s] = 0u;
            hi[s] = 0xFFFFFFFFu;
            bin_origin[s] = 0.0;
            bin_scale[s] = 0.0;
        }
        return;
    }
    double origin[kSets], scale[kSets];
#pragma unroll
    for (int set = 0; set < kSets; ++set) {
        const uint32_t set_lo = sh->range_lo[set], set_hi = sh->range_hi[set];
        origin[set] = bin_origin_for(set_lo);
        scale[set] = bin_scale_for(set_lo, set_hi);
    }
    // Sample rank of query q (0: low side of bracket 0, 1: its high side, 2/3: bracket 1) -- fp64 square root and divisions,
    // so it is worked out only by the wave that scans for it, not by all sixteen.  A bracket rank beyond the sample is
    // clamped to the sample's extreme here (the extreme still sets the bins); that side of the bracket is opened at the end.
    auto rank_of = [&](int q, bool& beyond) -> uint32_t {
        long long lo_r, hi_r;
        if constexpr (kPoint) {      // precision="fast": the sample's own order statistic at the matching rank, no bracket
            const unsigned long long k = (q >> 1) ? k0[1] : k0[0];
            const float f = n_total > 1 ? (float)k * __builtin_amdgcn_rcpf((float)(n_total - 1)) : 0.0f;
            lo_r = hi_r = (long long)rintf(f * (float)(m_valid - 1));
        } else {
            bracket_ranks(m_valid, n_total, (q >> 1) ? k0[1] : k0[0], lo_r, hi_r);
        }
        const long long want = (q & 1) ? hi_r : lo_r;
        beyond = want < 0 || want > (long long)m_valid - 1;
        return (uint32_t)min(max(want, 0ll), (long long)m_valid - 1);
    };
    uint32_t bin[kSets][kKeys];
#pragma unroll
    for (int set = 0; set < kSets; ++set)
#pragma unroll
        for (int i = 0; i < kKeys; ++i) {
            const uint32_t k = key[set][i];
            bin[set][i] = 0xFFFFFFFFu;
            if (k != 0xFFFFFFFFu) {
                bin[set][i] = bin_of(k, origin[set], scale[set]);
                atomicAdd(&sh->hist_s[set][bin[set][i]], 1u);
            }
        }
    __syncthreads();
    const int wave = threadIdx.x / kWave;
    if (wave < 4) {
        uint32_t b, rb;
        bool beyond;
        const uint32_t r = rank_of(wave, beyond);
        scan_pick32(sh->hist_s[kSets == 2 ? (wave >> 1) : 0], r, b, rb);
        if (lane == 0) {
            sh->bin_s[wave] = b;
            sh->rank_in_bin_s[wave] = rb;
            sh->rank_s[wave] = r;
            sh->beyond_s[wave] = beyond ? 1u : 0u;
        }
    }
    __syncthreads();
    {
        uint32_t picked[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) picked[q] = sh->bin_s[q];
#pragma unroll
        for (int set = 0; set < kSets; ++set)
#pragma unroll
            for (int i = 0; i < kKeys; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int q_set = kSets == 2 ? (q >> 1) : 0;
                    if (q_set == set && bin[set][i] == picked[q]) {
                        const uint32_t at = atomicAdd(&sh->count_s[q], 1u);
                        if (at < (uint32_t)kShortList) sh->list_s[q][at] = key[set][i];
                    }
                }
    }
    __syncthreads();
    uint32_t res[4];
    bool crowded = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) crowded = crowded || sh->count_s[q] > (uint32_t)kShortList;
    if (__builtin_expect(!crowded, 1)) {
        // the four short lists are ranked side by side, a quarter of the workgroup each
        const uint32_t per = blockDim.x / 4, q = threadIdx.x / per;
        rank_pick(sh->list_s[q], sh->count_s[q], sh->rank_in_bin_s[q], threadIdx.x - q * per, per, &sh->result_s[q]);
        __syncthreads();
#pragma unroll
        for (int q2 = 0; q2 < 4; ++q2) res[q2] = sh->result_s[q2];
    } else {    // a crowded bin (uniform decision: the counts live in LDS): radix rounds over the sample
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t* keys = sh->keys[kSets == 2 ? (q >> 1) : 0];
            bool unused;
            res[q] = radix_select_stream((unsigned long long)kSample, (unsigned long long)rank_of(q, unused), [keys](unsigned long long i, uint32_t& k) { k = keys[i]; return k != 0xFFFFFFFFu; }, sh);
        }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        lo[s] = res[2 * s];
        hi[s] = res[2 * s + 1];
        bin_origin[s] = bin_origin_for(lo[s]);      // candidate bins span the bracket (keys beyond an opened side land in the end bins)
        bin_scale[s] = bin_scale_for(lo[s], hi[s]);
        if constexpr (!kPoint) {
            // A bracket rank beyond the sample: that side is OPENED, not closed on the sample's extreme.  What lies beyond the
            // extreme of m samples is 1/(m+1) of the selection set -- about one sample stride of keys, whatever m is --, while a
            // bracket closed there misses whenever the wanted quantile is smaller than 1/(m+1): always for tiles with little
            // tissue (a 48x48 patch on a 512x512 tile puts 36 selected pixels in the sample; its 1st percentile lies below all
            // of them), and such a miss costs a whole-tile radix select (0.6 ms for one such tile in a config-2 batch).
            // (A bracket whose two DIFFERENT sample ranks hold one key stays as it is: a tie group, resolved from its counts alone
            // by the stages.  Two equal ranks -- a sample of one -- say nothing about ties.)
            // (the ranks were worked out by the scanning waves; here they are two LDS words per side)
            if (!(res[2 * s] == res[2 * s + 1] && sh->rank_s[2 * s + 1] > sh->rank_s[2 * s])) {
                if (sh->beyond_s[2 * s]) lo[s] = 0u;
                if (sh->beyond_s[2 * s + 1]) hi[s] = 0xFFFFFFFFu;
            }
        }
    }
}

// Pixel walker of one group: tile-local for transform, all tiles for the pooled fit.
struct GroupPixels {
    int64_t first_tile, pixels, count;   // count = pixels in the group
    int pooled;
    __device__ __forceinline__ void locate(int64_t i, int64_t& tile, int64_t& p) const {
        if (!pooled) {       // no 64-bit division on the per-tile path
            tile = first_tile;
            p = i;
        } else {
            tile = i / pixels;
            p = i - tile * pixels;
        }
    }
};

__device__ __forceinline__ GroupPixels group_pixels(const Geometry& g, int group) {
    GroupPixels gp;
    gp.pixels = g.pixels;
    gp.pooled = g.pooled;
    gp.first_tile = g.pooled ? 0 : group;
    gp.count = g.pooled ? g.n_tiles * g.pixels : g.pixels;
    return gp;
}

// Offset of sample j inside its stride-long cell: top `shift` bits of a multiplicative hash of j.
__device__ __forceinline__ uint32_t sample_offset(uint32_t j, int shift) { return shift ? (j * 0x9E3779B1u) >> (32 - shift) : 0u; }
// The same, kept inside a group of `count` pixels: the LAST cell may be partial (300 x 300 = 90000 pixels, stride 32: cell 2812 is
// 16 pixels long) and a hashed offset beyond its end would name a pixel that does not exist -- that sample was never written
// and the stages read whatever the workspace held (with the sampled percentiles of precision="sampled" that reached the output).
__device__ __forceinline__ uint32_t sample_offset_in(uint32_t j, int shift, uint32_t count) {
    const uint32_t off = sample_offset(j, shift), cell_begin = j << shift;
    const uint32_t cell_len = count - cell_begin;      // (callers only ask for cells that begin inside the group)
    return (shift && cell_len < (1u << shift)) ? off % cell_len : off;
}

// ------------------------------------------------------------------------------------------------
// streaming stage S1: raw moments of the OD vectors of one work item (+ the sample's OD on the way)
// ------------------------------------------------------------------------------------------------
template <int TPB> struct StatsScratch {
    double red[TPB / kWave][kPartial];
};

// Moments of ALL pixels of a work item (see the end of stats_item): the same fp32 runs / fp64 sums as the kept set.  Not
// (ordinary tiles never run it).
template <typename T, int V, int TPB, bool kInter>
__device__ __forceinline__ void stats_item_all_pixels(const T* __restrict__ img, int64_t pixels, int64_t p_begin, int64_t p_end, double* __restrict__ dst, StatsScratch<TPB>* sh) {
    constexpr int kShortRun = 32 / V > 0 ? 32 / V : 1;
    double acc[kPartial];
#pragma unroll
    for (int k = 0; k < kPartial; ++k) acc[k] = 0.0;
    for (int64_t run = p_begin; run < p_end; run += (int64_t)TPB * V * kShortRun) {
        float m[kPartial];
#pragma unroll
        for (int k = 0; k < kPartial; ++k) m[k] = 0.0f;
        const int64_t run_end = min(run + (int64_t)TPB * V * kShortRun, p_end);
        for (int64_t p = run + (int64_t)threadIdx.x * V; p < run_end; p += (int64_t)TPB * V) {
            float u[3][V];
            load_pixels<T, V, kInter>(img, pixels, p, u);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float od[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) od[c] = optical_density<T>(u[c][i]);
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
        }
#pragma unroll
        for (int k = 0; k < kPartial; ++k) acc[k] += (double)m[k];
    }
    const int wave = threadIdx.x / kWave;
    __syncthreads();      // everyone is done with the kept set's sums in the scratch
#pragma unroll
    for (int k = 0; k < kPartial; ++k) {
        const double s = wave_total_f64(acc[k]);
        if (lane_id() == kWave - 1) sh->red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < kPartial) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < TPB / kWave; ++w) s += sh->red[w][threadIdx.x];
        put(&dst[threadIdx.x], s);
    }
}

// A work item without kept pixels leaves the all-pixel sums only where the TILE may have fewer than three kept pixels.  Real slides
// are full of work items that are pure background inside tiles that are mostly tissue (32 rows of a 512-wide tile; the reference's
// example images: a third of the work items), and each paid a second sweep for sums nobody reads: 8 us of the 42 us moments pass on
// a batch of real tiles.  A witness settles it: every thread looks at ONE pixel of the tile (TPB pixels spread evenly over it);
// three kept ones among them prove that the tile has three kept pixels -- the same predicate on the same optical density, so there
// is no false proof; a tile whose tissue the witnesses miss just does the sweep as before.  Workgroup-uniform result.
template <typename T, int TPB, bool kInter>
__device__ __forceinline__ bool tile_has_three_kept_witnesses(const T* __restrict__ img, int64_t pixels) {
    const int64_t step = pixels / TPB > 0 ? pixels / TPB : 1;
    const int64_t p = (int64_t)threadIdx.x * step + (int64_t)(((uint32_t)threadIdx.x * 0x9E3779B1u) >> 16) % step;
    int kept = 0;
    if (p < pixels) {
        float v[3][1];
        load_pixels<T, 1, kInter>(img, pixels, p, v);
        const float od[3] = {optical_density<T>(v[0][0]), optical_density<T>(v[1][0]), optical_density<T>(v[2][0])};
        kept = od_selected(od, false) ? 1 : 0;
    }
    return __syncthreads_count(kept) >= 3;
}

// kEmit (Codable instantiations, g.code_epoch != 0): the pass also leaves the tile as 8-bit codes (see Coded<F>) -- a pack's elements
// are looked up in the code table; where all of them are grey levels (the table entry has the element's bits) their optical density
// comes from the table too, else (wave-uniform branch) from the expression as without kEmit and the tile is marked.
template <typename T, int V, int TPB, bool kInter, bool kEmit = false>
__device__ void stats_item(const T* __restrict__ images, const Geometry& g, const Workspace& ws, int64_t tile, int chunk_id, int64_t item, StatsScratch<TPB>* sh, const LevelTables<T>& tb,
                           const CodeTable<T, 4>* __restrict__ ct = nullptr) {
    const int64_t p_begin = (int64_t)chunk_id * g.chunk;
    const int64_t p_end = min(p_begin + (int64_t)g.chunk, g.pixels);
    const T* img = images + tile * 3 * g.pixels;
    const int group = g.pooled ? 0 : (int)tile;
    float* sample_out = ws.sample_od + (size_t)group * 3 * kSample;
    const uint32_t group_offset = g.pooled ? (uint32_t)(tile * g.pixels) : 0u;      // position of this tile inside its group (< 2^32)
    const uint32_t sample_count = (uint32_t)g.sample_count;
    const uint32_t group_count = (uint32_t)(g.pooled ? g.n_tiles * g.pixels : g.pixels);      // pixels of the group the sample is drawn from (< 2^32)
    const uint32_t mask = (uint32_t)g.sample_stride - 1u;          // stride is a power of two
    const int shift = 31 - __clz(g.sample_stride);
    constexpr int kLog2V = V == 16 ? 4 : V == 8 ? 3 : V == 4 ? 2 : V == 2 ? 1 : 0;
    const bool by_pack = shift >= kLog2V;      // a pack lies inside one sample cell (always, except for tiles of < 4096*V pixels)

    // products and the sums over 32 pixels of a lane in fp32, everything beyond in fp64: the fp32 rounding is
    // unbiased and averages out over the tile (~4e-9 on a covariance entry, measured against the fp64 covariance
    // in the tests), the cancellation in sum(xy) - sum(x)*mean(y) happens in fp64
    constexpr int kShortRun = 32 / V > 0 ? 32 / V : 1;      // packs per fp32 run
    double acc[kPartial];
#pragma unroll
    for (int k = 0; k < kPartial; ++k) acc[k] = 0.0;

    // (the next pack is requested before this one is worked on, across the runs: the pass is bound by its arithmetic for one- and
    // two-byte pixels, and a wave that waits for the load it just issued leaves the vector ALU to the three or four others)
    // Two-byte pixels only (64 x 512 x 512 bf16 in this form 145 -> 141 us): float32 / float64 are bound by the read either way, and
    // the uint8 kernel has no registers to spare for a second set of packs (256 x 224 x 224 uint8 107.5 -> 111 us with it).
    constexpr bool kAhead = sizeof(T) == 2 && !kEmit;
    PixelPacks<T, V, kInter> ahead;
    ahead.clear();
    if (kAhead && p_begin + (int64_t)threadIdx.x * V < p_end) ahead.load(img, g.pixels, p_begin + (int64_t)threadIdx.x * V);
    for (int64_t run = p_begin; run < p_end; run += (int64_t)TPB * V * kShortRun) {
        float m[kPartial];
#pragma unroll
        for (int k = 0; k < kPartial; ++k) m[k] = 0.0f;
        const int64_t run_end = min(run + (int64_t)TPB * V * kShortRun, p_end);
        for (int64_t p = run + (int64_t)threadIdx.x * V; p < run_end; p += (int64_t)TPB * V) {
            PixelPacks<T, V, kInter> u;
            if constexpr (kAhead) {
                u = ahead;
                if (p + (int64_t)TPB * V < p_end) ahead.load(img, g.pixels, p + (int64_t)TPB * V);      // (the thread's next pack, in this run or the next)
            } else {
                u.load(img, g.pixels, p);
            }
            // sample j sits in the cell [j*stride, (j+1)*stride) of its group at a hashed offset (a fixed offset would alias
            // with the image width: stride 1024 on a 2048-wide tile samples two columns only); tested once per pack
            const uint32_t gpos = group_offset + (uint32_t)p, j = gpos >> shift, off = sample_offset_in(j, shift, group_count);
            if (by_pack && (((gpos & mask) ^ off) >> kLog2V) == 0u && j < sample_count) {
                float raw[3] = {u.value(0, 0), u.value(1, 0), u.value(2, 0)};
#pragma unroll
                for (int i = 1; i < V; ++i)
                    if ((int)(off & (uint32_t)(V - 1)) == i) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) raw[c] = u.value(c, i);
                    }
#pragma unroll
                for (int c = 0; c < 3; ++c) put(&sample_out[c * kSample + j], od_of<T>(raw[c], tb));
            }
            float od_quad[4][3];
            if constexpr (kEmit) {
                static_assert(!kEmit || V == 4, "one quad per pack");
                uint32_t word[3][1];
                code_quad(u, 0, tb, *ct, g, ws, tile, od_quad, word);
                store_codes<1>(g, ws, tile, p, word);
            }
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float od[3];
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    if constexpr (kEmit) od[c] = od_quad[i % 4][c]; else od[c] = od_of<T>(u.value(c, i), tb);
                }
                if (!by_pack) {      // tiny tiles: stride < V, every pixel looks for itself
                    const uint32_t pos = gpos + (uint32_t)i, jj = pos >> shift;
                    if ((pos & mask) == sample_offset_in(jj, shift, group_count) && jj < sample_count) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) put(&sample_out[c * kSample + jj], od[c]);
                    }
                }
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
            }
        }
#pragma unroll
        for (int k = 0; k < kPartial; ++k) acc[k] += (double)m[k];
    }

    const int wave = threadIdx.x / kWave;
#pragma unroll
    for (int k = 0; k < kPartial; ++k) {
        const double s = wave_total_f64(acc[k]);       // fixed order; lane 63 holds the total
        if (lane_id() == kWave - 1) sh->red[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < kPartial) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < TPB / kWave; ++w) s += sh->red[w][threadIdx.x];
        put(&ws.partial[item * kPartial + threadIdx.x], s);
    }
    // A work item none of whose pixels pass the OD filter also leaves the moments of ALL its pixels: a tile with fewer than 3
    // kept pixels takes every pixel (torch_backend.py:409-410), and then each of its work items is such a one -- the plane
    // stage adds the partial sums up instead of streaming the whole tile through one workgroup (0.1 ms per blank tile).
    // Work items of ordinary tiles never get here; the second sweep reads what the first just brought in.
    double kept = 0.0;
#pragma unroll
    for (int w = 0; w < TPB / kWave; ++w) kept += sh->red[w][0];      // workgroup-uniform
    if (__builtin_expect(kept < 3.0, 0)) {
        if (!tile_has_three_kept_witnesses<T, TPB, kInter>(img, g.pixels)) stats_item_all_pixels<T, V, TPB, kInter>(img, g.pixels, p_begin, p_end, ws.partial_all + item * kPartial, sh);
    }
}

// ------------------------------------------------------------------------------------------------
// streaming stages S2 / S3: count keys below each bracket, gather + histogram the keys inside it
//   kConc == false: slots 0,1 share the angle key of the selected pixels
//   kConc == true : slots 2,3 use the two concentrations of every pixel
// Every wave works on its own: candidates are compacted (ballot + mbcnt, the running count lives in an SGPR)
// into the wave's private LDS queue and moved to the tile's candidate buffer with ONE global atomic per slot when
// the wave is done (or the queue is nearly full); there is no workgroup barrier inside the pixel loop, and the next
// pack of pixels is already in flight while the current one is processed.
// ------------------------------------------------------------------------------------------------
constexpr int kQueue = 512;            // candidate keys a wave queues per slot before it must flush

template <int TPB> struct BracketScratch {
    uint32_t keys[TPB / kWave][2][kQueue];
    uint32_t hist[2][256];        // bracket-relative histogram of everything this work item queued
    uint32_t below[2];
    uint32_t wave_n[2][TPB / kWave], base[2];      // the last flush: one reservation per slot for the whole work item
};

__device__ __forceinline__ void load_record(const StageRecord* rec, StageRecord& out) {
#pragma unroll
    for (int i = 0; i < 6; ++i) out.coef[i] = get(&rec->coef[i]);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        out.scale[i] = get(&rec->scale[i]);
        out.lo[i] = get(&rec->lo[i]);
        out.hi[i] = get(&rec->hi[i]);
        out.bin_origin[i] = get(&rec->bin_origin[i]);
        out.bin_scale[i] = get(&rec->bin_scale[i]);
    }
    out.use_all = get(&rec->use_all);
}

// Moves the n keys a wave queued for one slot to the group's candidate buffer and into the work item's histogram.
__device__ __forceinline__ void flush_queue(const uint32_t* queue, uint32_t n, uint32_t* counter, uint32_t* dst, uint32_t cap, uint32_t* hist, double origin, double scale) {
    if (n == 0) return;      // wave-uniform
    uint32_t base = 0;
    if (lane_id() == 0) base = atomicAdd(counter, n);
    base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
    for (uint32_t i = lane_id(); i < n; i += kWave) {
        const uint32_t key = queue[i];
        if (base + i < cap) put(&dst[base + i], key);
        atomicAdd(&hist[bin_of(key, origin, scale)], 1u);
    }
}

template <typename T, int V, bool kConc, int TPB, bool kInter>
__device__ void bracket_item(const T* __restrict__ images, const Geometry& g, const Workspace& ws, int64_t tile, int chunk_id, int64_t item, BracketScratch<TPB>* sh, const LevelTables<T>& tb) {
    const int group = g.pooled ? 0 : (int)tile;
    GroupState& st = ws.state[group];
    constexpr int s0 = kConc ? 2 : 0;
    const int64_t chunk = g.fine_chunk ? g.fine_chunk : g.chunk;
    const int64_t p_begin = (int64_t)chunk_id * chunk;
    const int64_t p_end = min(p_begin + chunk, g.pixels);
    const T* img = images + tile * 3 * g.pixels;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    constexpr int kCheck = V < 4 ? V : 4;     // pixels between two looks at the queue fill

    // The loop runs on the wave's base position, so all 64 lanes make the same trips (a lane past the end of the chunk
    // carries `live == false`); the first pack is requested before anything else, the stage record right behind it.
    int64_t base = p_begin + (int64_t)wave * kWave * V;
    const int64_t mine = (int64_t)lane_id() * V;
    PixelPacks<T, V, kInter> next;
    next.clear();
    if (base + mine < p_end) next.load(img, g.pixels, base + mine);
    StageRecord rec;
    load_record(&st.rec[kConc ? 1 : 0], rec);
    for (int i = threadIdx.x; i < 512; i += TPB) (&sh->hist[0][0])[i] = 0;
    if (threadIdx.x < 2) sh->below[threadIdx.x] = 0;
    __syncthreads();

    // the OD filter as one comparison: min(OD) >= threshold, with threshold -inf when every pixel is selected
    const float threshold = (kConc || rec.use_all != 0) ? -__builtin_huge_valf() : kBeta;
    const uint32_t lo_a = rec.lo[0], hi_a = rec.hi[0], lo_b = rec.lo[1], hi_b = rec.hi[1];
    GroupState& store = ws.state[g.spread ? (int)tile : group];      // where this tile's candidates and counters live
    uint32_t* cand_a = ws.cand + ((size_t)(g.spread ? (int)tile : group) * kSlots + s0) * g.cap;
    uint32_t* cand_b = cand_a + g.cap;
    uint32_t* queue_a = sh->keys[wave][0];
    uint32_t* queue_b = sh->keys[wave][1];
    // wave-uniform counters (SGPRs): comparisons leave lane masks, counting is a scalar popcount
    uint32_t below_a = 0, below_b = 0;
    uint32_t n_a = 0, n_b = 0;

    for (; base < p_end; base += (int64_t)TPB * V) {
        const uint64_t live_mask = __builtin_amdgcn_ballot_w64(base + mine < p_end);
        const PixelPacks<T, V, kInter> u = next;
        const int64_t p_next = base + (int64_t)TPB * V + mine;
        if (p_next < p_end) next.load(img, g.pixels, p_next);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float od[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) od[c] = od_of<T>(u.value(c, i), tb);
            uint32_t key_a, key_b;
            if constexpr (kConc) {
                float c0, c1;
                concentration(od, rec.coef, c0, c1);
                key_a = float_key(c0);
                key_b = float_key(c1);
            } else {
                key_a = key_b = angle_key(od, rec.coef);
            }
            // five comparisons per pixel; everything else happens on their lane masks with scalar instructions (the
            // ballot of a bare comparison is the comparison's own result register, inverse_ballot makes a mask the
            // branch condition again -- no 0/1 values in vector registers)
            const uint64_t valid = __builtin_amdgcn_ballot_w64(fminf(od[0], fminf(od[1], od[2])) >= threshold) & live_mask;     // torch_backend.py:404-405
            const uint64_t lt_a = __builtin_amdgcn_ballot_w64(key_a < lo_a), le_a = __builtin_amdgcn_ballot_w64(key_a <= hi_a);
            const uint64_t lt_b = __builtin_amdgcn_ballot_w64(key_b < lo_b), le_b = __builtin_amdgcn_ballot_w64(key_b <= hi_b);
            below_a += (uint32_t)__popcll(valid & lt_a);
            below_b += (uint32_t)__popcll(valid & lt_b);
            const uint64_t m_a = valid & ~lt_a & le_a, m_b = valid & ~lt_b & le_b;
            if (__builtin_amdgcn_inverse_ballot_w64(m_a)) queue_a[n_a + rank_in_mask(m_a)] = key_a;
            if (__builtin_amdgcn_inverse_ballot_w64(m_b)) queue_b[n_b + rank_in_mask(m_b)] = key_b;
            n_a += (uint32_t)__popcll(m_a);
            n_b += (uint32_t)__popcll(m_b);
            if ((i + 1) % kCheck == 0) {      // the next kCheck pixels add at most kCheck * 64 keys
                if (__builtin_expect(n_a > (uint32_t)(kQueue - kCheck * kWave), 0)) {
                    flush_queue(queue_a, n_a, &store.ncand[s0], cand_a, g.cap, sh->hist[0], rec.bin_origin[0], rec.bin_scale[0]);
                    n_a = 0;
                }
                if (__builtin_expect(n_b > (uint32_t)(kQueue - kCheck * kWave), 0)) {
                    flush_queue(queue_b, n_b, &store.ncand[s0 + 1], cand_b, g.cap, sh->hist[1], rec.bin_origin[1], rec.bin_scale[1]);
                    n_b = 0;
                }
            }
        }
    }
    {   // the last flush: ONE reservation per slot for all the waves of the work item (a tile's counter is shared by all its work
        // items: 1024 waves of a 2048x2048 tile reserving one by one made the pass twice as long as its pixels need)
        if (lane_id() == 0) {
            sh->wave_n[0][wave] = n_a;
            sh->wave_n[1][wave] = n_b;
        }
        __syncthreads();
        if (threadIdx.x < 2) {
            uint32_t total = 0;
#pragma unroll
            for (int w = 0; w < TPB / kWave; ++w) total += sh->wave_n[threadIdx.x][w];
            sh->base[threadIdx.x] = total ? atomicAdd(&store.ncand[s0 + threadIdx.x], total) : 0u;
        }
        __syncthreads();
        uint32_t base_a = sh->base[0], base_b = sh->base[1];
        for (int w = 0; w < wave; ++w) {
            base_a += sh->wave_n[0][w];
            base_b += sh->wave_n[1][w];
        }
        for (uint32_t i = lane_id(); i < n_a; i += kWave) {
            const uint32_t key = queue_a[i];
            if (base_a + i < g.cap) put(&cand_a[base_a + i], key);
            atomicAdd(&sh->hist[0][bin_of(key, rec.bin_origin[0], rec.bin_scale[0])], 1u);
        }
        for (uint32_t i = lane_id(); i < n_b; i += kWave) {
            const uint32_t key = queue_b[i];
            if (base_b + i < g.cap) put(&cand_b[base_b + i], key);
            atomicAdd(&sh->hist[1][bin_of(key, rec.bin_origin[1], rec.bin_scale[1])], 1u);
        }
    }

    if (lane_id() == 0) {      // the counts are already wave totals
        if (below_a) atomicAdd(&sh->below[0], below_a);
        if (below_b) atomicAdd(&sh->below[1], below_b);
    }
    __syncthreads();
    if (threadIdx.x < 2) {
        const uint32_t sum = sh->below[threadIdx.x];
        if (sum) atomicAdd(&store.below[s0 + threadIdx.x], sum);
    }
    // the work item's histograms are stored -- not added -- so the per-tile stage reads them without a reset
    if (g.fine_chunk) {      // small batches, many small work items: one histogram per tile, added to (integers: any order)
        uint32_t* hist_out = ws.block_hist + (size_t)tile * g.blocks_per_tile * 512;
        for (int i = threadIdx.x; i < 512; i += TPB) {
            const uint32_t v = (&sh->hist[0][0])[i];
            if (v) atomicAdd(&hist_out[i], v);
        }
    } else {
        uint32_t* hist_out = ws.block_hist + (size_t)item * 512;
        for (int i = threadIdx.x; i < 512; i += TPB) put(&hist_out[i], (&sh->hist[0][0])[i]);
    }
}

// ------------------------------------------------------------------------------------------------
// streaming stage S4: concentrations -> rescale -> reconstruct -> clamp -> cast  (torch_backend.py:452-461,560)
// ------------------------------------------------------------------------------------------------
template <typename T, typename O, int V, bool kUnit, int TPB, bool kInter>
__device__ __forceinline__ void reconstruct_item(const T* __restrict__ images, O* __restrict__ out, const Geometry& g, const Workspace& ws, int64_t tile, int chunk_id,
                                 const float* __restrict__ stain_matrix, const LevelTables<T>& tb, uint4* __restrict__ stage = nullptr, const float* __restrict__ given = nullptr) {
    const int64_t chunk = g.recon_chunk ? g.recon_chunk : (g.fine_chunk ? g.fine_chunk : g.chunk);
    const int64_t p_begin = (int64_t)chunk_id * chunk;
    const int64_t p_end = min(p_begin + chunk, g.pixels);
    const T* img = images + tile * 3 * g.pixels;
    O* dst = out + tile * 3 * g.pixels;
    // the tile's pseudo-inverse (6) and scale (2): the stage record of the launch before this one, or (fused transform) what the
    // caller has read from the stage job of the same launch
    float rec8[8];
    if (given) {
#pragma unroll
        for (int i = 0; i < 8; ++i) rec8[i] = given[i];
    } else {
        const StageRecord* rec = &ws.state[tile].rec[2];
#pragma unroll
        for (int i = 0; i < 6; ++i) rec8[i] = get(&rec->coef[i]);
        rec8[6] = get(&rec->scale[0]);
        rec8[7] = get(&rec->scale[1]);
    }

    // The whole chain OD -> C = pinv OD -> C * (tmc/maxC) -> SM C -> 240 exp(-.) (torch_backend.py:444-458) is linear between
    // the logarithm and the exponential, so it is folded into one 3x3 matrix per tile (fp64, then fp32):
    //   rgb_c = 2^x_c,   x_c = sum_j M[c][j] L_j + log2(240) (1 - sum_j M[c][j]),   L_j = log2(255 x_j + 1),
    //   M = SM diag(scale) pinv      (ln2 * log2e = 1 cancels between OD = ln240 - ln2 L and exp(-y) = 2^(-y log2e))
    // 9 fma per pixel instead of 20 multiply-adds; differs from the reference's operation order by ~1e-6 relative.
    float m[3][3], k[3];
    {
        double pinv[6], sm[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            pinv[i] = (double)rec8[i];
            sm[i] = (double)stain_matrix[i];
        }
        const double s0 = (double)rec8[6], s1 = (double)rec8[7];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double row = 0.0;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double v = sm[c * 2] * s0 * pinv[j] + sm[c * 2 + 1] * s1 * pinv[3 + j];
                m[c][j] = (float)v;
                row += (double)m[c][j];
            }
            k[c] = (float)(7.90689059560851852932 * (1.0 - row));      // log2(240)
        }
    }

    for (int64_t p = p_begin + (int64_t)threadIdx.x * V; p < p_end; p += (int64_t)TPB * V) {
        float u[3][V];
        // This pass is the call's LAST reader of the input: planar 16-byte packs are loaded non-temporally (they then leave no lines
        // behind in the L2s for the pass's own stores to push out: reconstruct 65.3 -> 61.5 us and the call 0.1733 -> 0.1657 ms per
        // step in bench.py's rotation over two batches; measured A/B on one box, profiles/r03_reconstruct_nt_loads_ab.txt)
        if constexpr (!kInter && sizeof(T) * V == 16) {
            typedef uint32_t u4v __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const u4v q = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(img + c * g.pixels + p));
                Pack<T, V> pk;
                __builtin_memcpy(&pk, &q, 16);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    if constexpr (sizeof(T) == 1) u[c][i] = __uint_as_float((uint32_t)pk.v[i]); else u[c][i] = raw_value<T>(pk.v[i]);
                }
            }
        } else {
            load_pixels<T, V, kInter, sizeof(T) == 1>(img, g.pixels, p, u);
        }
        O res[3][V];
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float l[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) l[c] = l2_of<T>(u[c][i], tb);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float x = fmaf(m[c][2], l[2], fmaf(m[c][1], l[1], fmaf(m[c][0], l[0], k[c])));
                float rgb = __builtin_amdgcn_exp2f(x);                                // bare v_exp_f32: a result below 2^-126 is 0 either way after the cast
                rgb = fminf(fmaxf(rgb, 0.0f), 255.0f);                                // :459, :128
                using S = typename Sem<T>::type;      // (the output rules follow what the element means: a coded float tile is a float tile)
                if constexpr (kUnit) {
                    // cast to the input dtype first, then /255 in that dtype (_template.py:111-112);
                    // u8 promotes to f32 (and, with SX_MACENKO_OUT_*, that f32 is cast once more: `.to(bfloat16)` fused)
                    if constexpr (sizeof(S) == 1) {
                        res[c][i] = Elem<O>::store(div255_of_level((float)Elem<S>::store(rgb)));
                    } else if constexpr (sizeof(S) == 8) {
                        res[c][i] = (double)rgb / 255.0;
                    } else {
                        res[c][i] = Elem<O>::store(div255_of_level(Elem<S>::load(Elem<S>::store(rgb))));      // (clamped to [0, 255] above)
                    }
                } else if constexpr (sizeof(S) == 1 && sizeof(O) == 2) {
                    res[c][i] = Elem<O>::store((float)Elem<T>::store(rgb));      // the truncated grey level, exactly representable
                } else {
                    res[c][i] = Elem<O>::store(rgb);
                }
            }
        }
        if constexpr (kInter && V > 1 && sizeof(O) * V == 16) {
            if (__builtin_amdgcn_ballot_w64(true) == ~0ull) {      // wave-uniform: every lane has a pack (all but a tile's last sweep)
                store_pixels_staged<O, V>(dst, p, res, stage + (threadIdx.x / kWave) * (3 * kWave));
                continue;
            }
        }
        store_pixels<O, V, kInter>(dst, g.pixels, p, res);
    }
}

// ------------------------------------------------------------------------------------------------
// per-tile stage A ("plane"): moments -> covariance -> plane vectors; angle brackets from the sample
// ------------------------------------------------------------------------------------------------
// Optical density of the sample pixels this thread owns (written by S1).
__device__ __forceinline__ void load_sample(const Workspace& ws, int group, int m, float (&od)[kKeys][3]) {
    const float* sample = ws.sample_od + (size_t)group * 3 * kSample;
#pragma unroll
    for (int i = 0; i < kKeys; ++i) {
        const int j = i * kGroupThreads + (int)threadIdx.x;
#pragma unroll
        for (int c = 0; c < 3; ++c) od[i][c] = j < m ? get(&sample[c * kSample + j]) : 0.0f;
    }
}

// Raw moments of ALL pixels of a tile (torch_backend.py:409-410: fewer than 3 pixels pass the OD filter): every work
// item of such a tile has left them in ws.partial_all (stats_item); added up in index order like the kept set.
__device__ void all_pixel_moments(const Geometry& g, const Workspace& ws, int group, TileScratch* sh) {
    const int64_t first = (int64_t)group * g.blocks_per_tile;
    const int64_t nblk = g.blocks_per_tile;
    const int rows = 64;
    double running = 0.0;
    for (int64_t b0 = 0; b0 < nblk; b0 += rows) {
        const int live = (int)min((int64_t)rows, nblk - b0);
        __syncthreads();
        if ((int)threadIdx.x < live * kPartial) sh->stage[threadIdx.x / kPartial][threadIdx.x % kPartial] = get(&ws.partial_all[(first + b0) * kPartial + threadIdx.x]);
        __syncthreads();
        if (threadIdx.x < kPartial)
            for (int b = 0; b < live; ++b) running += sh->stage[b][threadIdx.x];
    }
    if (threadIdx.x < kPartial) sh->mom[kPartial + threadIdx.x] = running;
    __syncthreads();
}

template <typename T, bool kFast = false>
__device__ void plane_stage(const T* __restrict__ images, const Geometry& g, const Workspace& ws, int group, int allow_fallback, TileScratch* sh, const double* __restrict__ given_moments = nullptr,
                            const float* __restrict__ target_max_conc = nullptr) {
    GroupState& st = ws.state[group];
    SX_STAMP(st, 0);
    reset_scratch(sh);
    // the sample is fetched first: its latency hides behind the moments and the eigen-decomposition
    float sod[kKeys][3];
    load_sample(ws, group, g.sample_count, sod);
    if (given_moments) {      // distributed fit: the moments of the whole group, already summed over the ranks
        if (threadIdx.x < kPartial) sh->mom[threadIdx.x] = get(&given_moments[threadIdx.x]);
        if (threadIdx.x == kPartial) sh->mom[kPartial] = (double)g.n_all;
        if (threadIdx.x > kPartial && threadIdx.x < kMoments) sh->mom[threadIdx.x] = 0.0;
    } else {
        // fixed-order (deterministic) sum of the work items' partial moments: lanes fetch them in parallel, one
        // thread per moment adds them in index order
        const int64_t first = g.pooled ? 0 : (int64_t)group * g.blocks_per_tile;
        const int64_t nblk = g.pooled ? g.n_tiles * g.blocks_per_tile : g.blocks_per_tile;
        const int rows = 64;
        double running = 0.0;
        for (int64_t b0 = 0; b0 < nblk; b0 += rows) {
            const int live = (int)min((int64_t)rows, nblk - b0);
            __syncthreads();
            if ((int)threadIdx.x < live * kPartial) sh->stage[threadIdx.x / kPartial][threadIdx.x % kPartial] = get(&ws.partial[(first + b0) * kPartial + threadIdx.x]);
            __syncthreads();
            if (threadIdx.x < kPartial)
                for (int b = 0; b < live; ++b) running += sh->stage[b][threadIdx.x];
        }
        if (threadIdx.x < kPartial) sh->mom[threadIdx.x] = running;
        const GroupPixels gp = group_pixels(g, group);
        if (threadIdx.x == kPartial) sh->mom[kPartial] = (double)gp.count;       // all-pixel set: the count is known,
        if (threadIdx.x > kPartial && threadIdx.x < kMoments) sh->mom[threadIdx.x] = 0.0;   // the sums only matter in the fallback below
    }
    __syncthreads();
    if (__builtin_expect(allow_fallback && !g.pooled && sh->mom[0] < 3.0, 0)) all_pixel_moments(g, ws, group, sh);     // uniform, rare (blank tiles)
    __syncthreads();
    SX_STAMP(st, 1);
    if (threadIdx.x < 2) {      // two lanes: the eigen step's two power iterations side by side (lane 1 only helps)
        double cov[9];
        bool use_all;
        unsigned long long n_sel;
        float vecs[6];
        plane_from_moments<true>(sh->mom, allow_fallback != 0, cov, vecs, use_all, n_sel);
      if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            sh->coef[i] = vecs[i];
            put(&st.vecs[i], vecs[i]);
        }
        sh->flag = use_all ? 1 : 0;
        sh->n_sel = n_sel;
#pragma unroll
        for (int k = 0; k < kMoments; ++k) put(&st.mom[k], sh->mom[k]);
#pragma unroll
        for (int i = 0; i < 9; ++i) put(&st.cov[i], cov[i]);
        put(&st.use_all, sh->flag);
        put(&st.n_sel, n_sel);
        put(&st.fell_back, 0u);
        put(&st.spec, 0u);
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            put(&st.below[s], 0u);
            put(&st.ncand[s], 0u);
        }
      }
    }
    if (g.fine_chunk) {      // the tile's histogram is added to by many small work items: start from zero
        uint32_t* row = ws.block_hist + (size_t)group * g.blocks_per_tile * 512;
        for (int i = threadIdx.x; i < 512; i += blockDim.x) put(&row[i], 0u);
    }
    if (g.spread) {      // every tile's counters and the group-level sums start from zero
        for (int64_t i = threadIdx.x; i < g.n_tiles * kSlots; i += blockDim.x) {
            put(&ws.state[i / kSlots].below[i % kSlots], 0u);
            put(&ws.state[i / kSlots].ncand[i % kSlots], 0u);
        }
        uint32_t* pool_words = reinterpret_cast<uint32_t*>(ws.pool);
        for (int i = threadIdx.x; i < (int)(offsetof(PoolState, compact) / sizeof(uint32_t)); i += blockDim.x) put(&pool_words[i], 0u);
    }
    __syncthreads();
    SX_STAMP(st, 2);
    const bool use_all = sh->flag != 0;
    const unsigned long long n_sel = sh->n_sel;
    float v[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) v[i] = sh->coef[i];
    // angle keys of the selected sample pixels
    uint32_t key[1][kKeys];
#pragma unroll
    for (int i = 0; i < kKeys; ++i) {
        const int j = i * kGroupThreads + (int)threadIdx.x;
        uint32_t k = 0xFFFFFFFFu;
        if (j < g.sample_count && od_selected(sod[i], use_all)) k = angle_key(sod[i], v);
        key[0][i] = k;
        sh->keys[0][j] = k;
    }
    SX_STAMP(st, 3);
    const unsigned long long k0[2] = {nearest_rank_index(1.0, n_sel), nearest_rank_index(99.0, n_sel)};   // alpha = 1 (torch_backend.py:421-422)
    uint32_t lo[2], hi[2];
    double origin[2], scale[2];
    if constexpr (kFast) {
        // precision="fast" (reference: Macenko(precision="fast"), a relaxed-accuracy path): the percentiles of the
        // 4096-pixel sample stand in for the percentiles of the tile, so the two bracket passes and two of the three
        // per-tile stages disappear -- moments pass, this stage, reconstruct.
        sample_brackets<1, true>(sh, key, n_sel, k0, lo, hi, origin, scale);
        if (threadIdx.x == 0) {
            float he[6], pinv[6];
            stain_vectors_and_pinv(v, lo[0], lo[1], he, pinv);
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                put(&st.he[i], he[i]);
                put(&st.pinv[i], pinv[i]);
                sh->coef[i] = pinv[i];
            }
            put(&st.phi_key[0], lo[0]);
            put(&st.phi_key[1], lo[1]);
        }
        __syncthreads();
        reset_scratch(sh);
        float pinv[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) pinv[i] = sh->coef[i];
        __syncthreads();
        uint32_t ckey[2][kKeys];
#pragma unroll
        for (int i = 0; i < kKeys; ++i) {
            const int j = i * kGroupThreads + (int)threadIdx.x;
            uint32_t ka = 0xFFFFFFFFu, kb = 0xFFFFFFFFu;
            if (j < g.sample_count) {
                float c0, c1;
                concentration(sod[i], pinv, c0, c1);
                ka = float_key(c0);
                kb = float_key(c1);
            }
            ckey[0][i] = ka;
            ckey[1][i] = kb;
            sh->keys[0][j] = ka;
            sh->keys[1][j] = kb;
        }
        const unsigned long long n_all = (unsigned long long)group_pixels(g, group).count;
        const unsigned long long k99 = nearest_rank_index(99.0, n_all);
        const unsigned long long kc[2] = {k99, k99};
        sample_brackets<2, true>(sh, ckey, n_all, kc, lo, hi, origin, scale);
        if (threadIdx.x == 0) {
            const float m0 = key_float(lo[0]), m1 = key_float(lo[1]);
            put(&st.max_c[0], m0);
            put(&st.max_c[1], m1);
            StageRecord* rec = &st.rec[2];
#pragma unroll
            for (int i = 0; i < 6; ++i) put(&rec->coef[i], pinv[i]);
            put(&rec->scale[0], target_max_conc[0] / m0);
            put(&rec->scale[1], target_max_conc[1] / m1);
#pragma unroll
            for (int s = 0; s < kSlots; ++s) put(&st.ncand_seen[s], 0u);
        }
        return;
    }
    sample_brackets<1>(sh, key, n_sel, k0, lo, hi, origin, scale);
    SX_STAMP(st, 4);
    if (threadIdx.x == 0) {
        StageRecord* rec = &st.rec[0];
#pragma unroll
        for (int i = 0; i < 6; ++i) put(&rec->coef[i], v[i]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            put(&st.rank[s], k0[s]);
            put(&rec->lo[s], lo[s]);
            put(&rec->hi[s], hi[s]);
            put(&rec->bin_origin[s], origin[s]);
            put(&rec->bin_scale[s], scale[s]);
        }
        put(&rec->use_all, use_all ? 1 : 0);
    }
}

// ------------------------------------------------------------------------------------------------
// exact order statistics of the two slots of a stage from what the streaming stage left behind
// ------------------------------------------------------------------------------------------------
template <typename T, bool kThrough = false, class Scratch>
__device__ uint32_t select_whole_group(const T* __restrict__ images, const Geometry& g, int group, int slot, unsigned long long rank, const float* coef, bool use_all,
                                       Scratch* sh, uint32_t* __restrict__ keep = nullptr) {
    const GroupPixels gp = group_pixels(g, group);
    return radix_select_stream<kThrough>((unsigned long long)gp.count, rank,
                               [&](unsigned long long i, uint32_t& k) {
                                   int64_t tile, p;
                                   gp.locate((int64_t)i, tile, p);
                                   float od[3];
                                   load_od_scalar<T>(images, g, tile, p, od);
                                   if (!od_selected(od, use_all)) return false;
                                   if (slot < 2) {
                                       k = angle_key(od, coef);
                                   } else {
                                       float c0, c1;
                                       concentration(od, coef, c0, c1);
                                       k = float_key(slot == 2 ? c0 : c1);
                                   }
                                   return true;
                               },
                               sh, keep);
}

// What resolve_pair() needs from memory, fetched at the top of the kernel in one batch (one memory latency for
// the whole stage instead of one per step): the counters of the two slots, this thread's share of the work items'
// histograms and the first 8192 candidates of each slot (loaded before ncand is known: the buffer is always at
// least kMinCap long, entries beyond ncand are ignored later).
struct PairPrefetch {
    uint32_t ncand[2], below[2], lo[2], hi[2];
    unsigned long long rank[2];
    double origin[2], scale[2];
    uint32_t hist[kPrefetchHist];
    uint32_t cand[2][kPrefetchCand];
};

__device__ __forceinline__ void prefetch_pair(PairPrefetch& pf, const Geometry& g, const Workspace& ws, int group, int first_slot) {
    const GroupState& st = ws.state[group];
    const StageRecord* rec = &st.rec[first_slot ? 1 : 0];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int slot = first_slot + j;
        pf.ncand[j] = get(&st.ncand[slot]);
        pf.below[j] = get(&st.below[slot]);
        pf.rank[j] = get(&st.rank[slot]);
        pf.origin[j] = get(&rec->bin_origin[j]);
        pf.scale[j] = get(&rec->bin_scale[j]);
        pf.lo[j] = get(&rec->lo[j]);
        pf.hi[j] = get(&rec->hi[j]);
    }
    const int64_t first = g.pooled ? 0 : (int64_t)group * g.blocks_per_tile;
    const int64_t nblk = g.fine_chunk ? 1 : (g.pooled ? g.n_tiles * g.blocks_per_tile : g.blocks_per_tile);
    const uint32_t* src = ws.block_hist + first * 512 + (threadIdx.x & 511);     // thread owns bin t%256 of slot (t/256)%2 ...
    const int half = (int)threadIdx.x >> 9;                                        // ... for every second work item
#pragma unroll
    for (int u = 0; u < kPrefetchHist; ++u) {
        const int64_t blk = half + 2 * u;
        pf.hist[u] = blk < nblk ? get(&src[blk * 512]) : 0u;
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t* cand = ws.cand + ((size_t)group * kSlots + first_slot + j) * g.cap;
#pragma unroll
        for (int u = 0; u < kPrefetchCand; ++u) pf.cand[j][u] = get(&cand[u * kGroupThreads + threadIdx.x]);
    }
}

// Exact order statistics of the two slots of a stage from what the streaming stage left behind: the work items'
// bracket-relative histograms are summed (integers: any order), one wave per slot picks the bin holding the wanted
// rank, the candidates of that bin (~n/256 keys) are listed and rank-counted.  Anything unusual -- bracket missed
// or overflowed, a crowded bin -- goes to the radix paths (over the candidates, or recomputing every key of the
// group from the pixels).  Needs reset_scratch() before it.
template <typename T>
__device__ void resolve_pair(const T* __restrict__ images, const Geometry& g, const Workspace& ws, int group, int first_slot, const float* coef, bool use_all,
                             const PairPrefetch& pf, uint32_t (&key_out)[2], TileScratch* sh) {
    GroupState& st = ws.state[group];
    bool ok[2], tie[2];
    uint32_t want_in[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        // A bracket that closed on a single key (lo == hi: the sample's order statistics around the wanted rank are one
        // tie group -- blank or flat tiles, few-colour images) needs no candidates at all: every key counted inside it is
        // that key, so the counts alone say whether it is the answer.  Without this such a tile overflows its candidate
        // buffer and one workgroup radix-selects over the whole tile (9 ms for a 1024x1024 tile against 0.1 ms).
        tie[j] = !g.no_tie && pf.lo[j] == pf.hi[j] && pf.rank[j] >= pf.below[j] && pf.rank[j] - pf.below[j] < pf.ncand[j];
        ok[j] = !tie[j] && pf.ncand[j] <= g.cap && pf.rank[j] >= pf.below[j] && pf.rank[j] - pf.below[j] < pf.ncand[j];
        want_in[j] = ok[j] ? (uint32_t)(pf.rank[j] - pf.below[j]) : 0u;
    }
    const int64_t first = g.pooled ? 0 : (int64_t)group * g.blocks_per_tile;
    const int64_t nblk = g.fine_chunk ? 1 : (g.pooled ? g.n_tiles * g.blocks_per_tile : g.blocks_per_tile);
    __syncthreads();      // the scratch reset is visible
    {
        const uint32_t* src = ws.block_hist + first * 512 + (threadIdx.x & 511);
        uint32_t sum = 0;
#pragma unroll
        for (int u = 0; u < kPrefetchHist; ++u) sum += pf.hist[u];
#pragma unroll 8
        for (int64_t blk = ((int)threadIdx.x >> 9) + 2 * kPrefetchHist; blk < nblk; blk += 2) sum += get(&src[blk * 512]);
        if (sum) atomicAdd(&(&sh->hist_c[0][0])[threadIdx.x & 511], sum);
    }
    __syncthreads();
    const int wave = threadIdx.x / kWave;
    // (no runtime indexing of the register arrays: that would send the whole prefetch to scratch memory)
    const bool second = wave == 1;
    if (wave < 2 && (second ? ok[1] : ok[0])) {
        uint32_t b, rb;
        scan_pick32(sh->hist_c[wave], second ? want_in[1] : want_in[0], b, rb);
        if (lane_id() == 0) {
            sh->rank_in_bin_c[wave] = rb;
            sh->range_c[wave][0] = b;
        }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (!ok[j]) continue;
        const uint32_t b = sh->range_c[j][0], n = pf.ncand[j];
#pragma unroll
        for (int u = 0; u < kPrefetchCand; ++u) {
            const uint32_t idx = u * kGroupThreads + threadIdx.x, k = pf.cand[j][u];
            if (idx < n && bin_of(k, pf.origin[j], pf.scale[j]) == b) {
                const uint32_t at = atomicAdd(&sh->count_c[j], 1u);
                if (at < (uint32_t)kSample) sh->keys[j][at] = k;
            }
        }
        // big tiles / pooled groups: the rest of the candidates, eight independent loads in flight
        const uint32_t* cand = ws.cand + ((size_t)group * kSlots + first_slot + j) * g.cap;
        for (uint32_t base = kPrefetchCand * kGroupThreads + threadIdx.x; base < n; base += kGroupThreads * 8) {
            uint32_t k[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t idx = base + u * kGroupThreads;
                k[u] = idx < n ? get(&cand[idx]) : 0u;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const uint32_t idx = base + u * kGroupThreads;
                if (idx < n && bin_of(k[u], pf.origin[j], pf.scale[j]) == b) {
                    const uint32_t at = atomicAdd(&sh->count_c[j], 1u);
                    if (at < (uint32_t)kSample) sh->keys[j][at] = k[u];
                }
            }
        }
    }
    __syncthreads();
    {   // rank counting of the two short lists side by side, half the workgroup each
        const uint32_t per = blockDim.x / 2, j = threadIdx.x / per;
        if ((j ? ok[1] : ok[0]) && sh->count_c[j] <= (uint32_t)kShortList) rank_pick(sh->keys[j], sh->count_c[j], sh->rank_in_bin_c[j], threadIdx.x - j * per, per, &sh->result_c[j]);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int slot = first_slot + j;
        if (__builtin_expect(ok[j] && sh->count_c[j] <= (uint32_t)kShortList, 1)) {
            key_out[j] = sh->result_c[j];
        } else if (tie[j]) {
            key_out[j] = pf.lo[j];
        } else if (ok[j] && sh->count_c[j] <= (uint32_t)kSample) {
            // crowded bin whose keys still fit the LDS list (big tiles: 2 % of a 2048x2048 tile are 84 000 candidates, ~350 per
            // bin on average and more where they are dense): radix rounds over the list, not over all candidates in memory
            const uint32_t* list = sh->keys[j];
            if (threadIdx.x == 0) atomicOr(&st.fell_back, 16u << slot);
            key_out[j] = radix_select_stream((unsigned long long)sh->count_c[j], (unsigned long long)sh->rank_in_bin_c[j], [list](unsigned long long i, uint32_t& k) { k = list[i]; return true; }, sh);
        } else if (ok[j]) {      // crowded bin: radix rounds over the candidates
            const uint32_t* cand = ws.cand + ((size_t)group * kSlots + slot) * g.cap;
            if (threadIdx.x == 0) atomicOr(&st.fell_back, 16u << slot);
            key_out[j] = radix_select_stream((unsigned long long)pf.ncand[j], (unsigned long long)want_in[j], [cand](unsigned long long i, uint32_t& k) { k = get(&cand[i]); return true; }, sh);
        } else {                 // the bracket did not hold: recompute every key of the group
            if (threadIdx.x == 0) atomicOr(&st.fell_back, 1u << slot);
            key_out[j] = select_whole_group<T>(images, g, group, slot, pf.rank[j], coef, use_all, sh);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// pooled fit over several tiles: the two many-workgroup kernels between a streaming stage and the group stage
// ------------------------------------------------------------------------------------------------
// (1) one workgroup of 512 threads per tile adds the tile's work-item histograms and counters into the PoolState
__global__ __launch_bounds__(512) void pool_reduce_kernel(Geometry g, Workspace ws, int stage) {
    const int tile = blockIdx.x;
    const uint32_t* src = ws.block_hist + (size_t)tile * g.blocks_per_tile * 512 + threadIdx.x;
    uint32_t sum = 0;
#pragma unroll 8
    for (int r = 0; r < g.blocks_per_tile; ++r) sum += get(&src[(size_t)r * 512]);
    if (sum) atomicAdd(&(&ws.pool->hist[stage][0][0])[threadIdx.x], sum);
    if (threadIdx.x < 2) {
        const int slot = 2 * stage + threadIdx.x;
        const uint32_t n = get(&ws.state[tile].ncand[slot]), b = get(&ws.state[tile].below[slot]);
        if (b) atomicAdd(&ws.pool->below[slot], b);
        atomicAdd(&ws.pool->ncand[slot], min(n, g.cap));
        if (n > g.cap) atomicOr(&ws.pool->overflow, 1u << slot);
    }
}

// (2) one workgroup per tile: every workgroup scans the pooled histogram for the bin holding the wanted rank (the same
// answer everywhere), then moves its tile's candidates of that bin to the compact list
// Distributed fit (sx_macenko_pfit_gather_packed): `sums` are the counts added up over all ranks (the layout of
// pfit_reduce_export_kernel) and take the place of the pool's local sums -- what a separate import launch did --, and `row_out` is
// the rank's stage record [count, count, stale flag | 2 x share keys]: the candidates go there as well as to the pool's list, and the
// workgroup that arrives last writes the two counts and the flag -- what a separate export launch did.
__global__ __launch_bounds__(kGroupThreads) void pool_gather_kernel(Geometry g, Workspace ws, int stage, const long long* __restrict__ sums = nullptr, const int* __restrict__ stale_flag = nullptr,
                                                                    uint32_t* __restrict__ row_out = nullptr, int share = 0) {
    __shared__ __attribute__((aligned(16))) uint32_t hist[2][256];
    constexpr int kLocal = 2048;
    __shared__ uint32_t range[2][2], live[2], local_n[2], local_base[2];
    __shared__ uint32_t local[2][kLocal];
    const int tile = blockIdx.x, wave = threadIdx.x / kWave;
    if (threadIdx.x < 2) local_n[threadIdx.x] = 0;
    const GroupState& st0 = ws.state[0];
    const GroupState& mine = ws.state[tile];
    PoolState* pool = ws.pool;
    uint32_t pre[2][kPrefetchCand];      // the tile's first 8192 candidates per slot, requested before anything else
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const uint32_t* cand = ws.cand + ((size_t)tile * kSlots + 2 * stage + j) * g.cap;
#pragma unroll
        for (int u = 0; u < kPrefetchCand; ++u) pre[j][u] = get(&cand[u * kGroupThreads + threadIdx.x]);
    }
    if (threadIdx.x < 512) (&hist[0][0])[threadIdx.x] = sums ? (uint32_t)sums[stage * 512 + threadIdx.x] : get(&(&pool->hist[stage][0][0])[threadIdx.x]);
    __syncthreads();
    if (wave < 2) {
        const int slot = 2 * stage + wave;
        const uint32_t ncand = sums ? (uint32_t)sums[1024 + kSlots + slot] : get(&pool->ncand[slot]), below = sums ? (uint32_t)sums[1024 + slot] : get(&pool->below[slot]);
        const uint32_t overflow = sums ? (sums[1024 + 2 * kSlots] ? 0xFu : 0u) : get(&pool->overflow);
        const unsigned long long rank = get(&st0.rank[slot]);
        const bool ok = ((overflow >> slot) & 1u) == 0 && rank >= below && rank - below < ncand;
        if (sums && tile == 0 && lane_id() == 0) {      // (the group stage reads these from the pool)
            put(&pool->ncand[slot], ncand);
            put(&pool->below[slot], below);
            if (wave == 0) put(&pool->overflow, overflow);
        }
        uint32_t b = 0, rb = 0, first = 1, last = 0;
        if (ok) {
            scan_pick32(hist[wave], (uint32_t)(rank - below), b, rb);
            if (lane_id() == 0) bin_key_range(b, get(&st0.rec[stage].bin_origin[wave]), get(&st0.rec[stage].bin_scale[wave]), first, last);
        }
        if (lane_id() == 0) {
            range[wave][0] = first;
            range[wave][1] = last;
            live[wave] = ok ? 1u : 0u;
            if (tile == 0) {
                put(&pool->ok[slot], ok ? 1u : 0u);
                put(&pool->rank_in_bin[slot], rb);
                put(&pool->range[slot][0], first);
                put(&pool->range[slot][1], last);
            }
        }
    }
    __syncthreads();
    // matches are collected in LDS first; one reservation per workgroup and slot in the pooled list (thousands of returning
    // atomics on one address from 64 workgroups cost 60 us)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (!live[j]) continue;
        const int slot = 2 * stage + j;
        const uint32_t k_first = range[j][0], k_last = range[j][1], n = min(get(&mine.ncand[slot]), g.cap);
        const uint32_t* cand = ws.cand + ((size_t)tile * kSlots + slot) * g.cap;
        auto keep = [&](uint32_t k) {
            const uint32_t at = atomicAdd(&local_n[j], 1u);
            if (at < (uint32_t)kLocal) {
                local[j][at] = k;
            } else {      // more than the LDS list holds (degenerate data): straight to the pooled list
                const uint32_t g_at = atomicAdd(&pool->compact_n[slot], 1u);
                if (g_at < (uint32_t)kCompact) put(&pool->compact[slot][g_at], k);
                if (row_out && g_at < (uint32_t)share) row_out[3 + (size_t)j * share + g_at] = k;
            }
        };
#pragma unroll
        for (int u = 0; u < kPrefetchCand; ++u) {
            const uint32_t idx = u * kGroupThreads + threadIdx.x, k = pre[j][u];
            if (idx < n && k >= k_first && k <= k_last) keep(k);
        }
        for (uint32_t idx = kPrefetchCand * kGroupThreads + threadIdx.x; idx < n; idx += kGroupThreads) {
            const uint32_t k = get(&cand[idx]);
            if (k >= k_first && k <= k_last) keep(k);
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 && live[threadIdx.x]) {
        const uint32_t m = min(local_n[threadIdx.x], (uint32_t)kLocal);
        local_base[threadIdx.x] = m ? atomicAdd(&pool->compact_n[2 * stage + threadIdx.x], m) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        if (!live[j]) continue;
        const uint32_t m = min(local_n[j], (uint32_t)kLocal), base = local_base[j];
        for (uint32_t i = threadIdx.x; i < m; i += blockDim.x) {
            if (base + i < (uint32_t)kCompact) put(&pool->compact[2 * stage + j][base + i], local[j][i]);
            if (row_out && base + i < (uint32_t)share) row_out[3 + (size_t)j * share + base + i] = local[j][i];
        }
    }
    if (row_out) {      // the record's head: by the workgroup that arrives last (one counter add per workgroup)
        // (no fence: the keys are read by the NEXT launch, and the counts the last workgroup reads were added by returning atomics that
        // every workgroup has waited for before its arrival add -- an agent-scope release here writes the L2 back: 25 us instead of 11)
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(&pool->arrived, 1u) == (uint32_t)g.n_tiles - 1u) {
            row_out[0] = atomicAdd(&pool->compact_n[2 * stage], 0u);
            row_out[1] = atomicAdd(&pool->compact_n[2 * stage + 1], 0u);
            row_out[2] = stale_flag ? (uint32_t)stale_flag[0] : 0u;
            atomicExch(&pool->arrived, 0u);
        }
    }
}

// (3) in the group stage: the compact list (the candidates of ONE value-linear bin over all tiles: hundreds to a few
// thousand keys, with heavy ties when the tiles come from 8-bit data) goes to LDS and four byte-wise radix rounds over
// it give the wanted element; a list that does not fit LDS is read from memory in every round.
template <typename T>
__device__ void resolve_pair_spread(const T* __restrict__ images, const Geometry& g, const Workspace& ws, int first_slot, const float* coef, bool use_all,
                                    uint32_t (&key_out)[2], TileScratch* sh) {
    GroupState& st = ws.state[0];
    const PoolState* pool = ws.pool;
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int slot = first_slot + j;
        const uint32_t n = get(&pool->compact_n[slot]), want = get(&pool->rank_in_bin[slot]);
        const bool ok = get(&pool->ok[slot]) != 0 && n <= (uint32_t)kCompact && want < n;      // uniform
        if (ok && n <= (uint32_t)kSample) {
            uint32_t* keys = sh->keys[j];      // the sample-key area is free until the brackets are taken
            for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) keys[i] = get(&pool->compact[slot][i]);
            key_out[j] = radix_select_stream((unsigned long long)n, (unsigned long long)want, [keys](unsigned long long i, uint32_t& k) { k = keys[i]; return true; }, sh);
        } else if (ok) {
            const uint32_t* keys = pool->compact[slot];
            if (threadIdx.x == 0) atomicOr(&st.fell_back, 16u << slot);
            key_out[j] = radix_select_stream((unsigned long long)n, (unsigned long long)want, [keys](unsigned long long i, uint32_t& k) { k = get(&keys[i]); return true; }, sh);
        } else if (g.distributed) {      // the other ranks' pixels are out of reach: report, the caller repeats with the radix rounds
            if (threadIdx.x == 0) {
                atomicOr(&st.fell_back, 1u << slot);
                put(&ws.pool->status, 1u);
            }
            key_out[j] = 0u;
        } else {      // bracket missed or a buffer overflowed: recompute every key of the group (exact, slow)
            if (threadIdx.x == 0) atomicOr(&st.fell_back, 1u << slot);
            key_out[j] = select_whole_group<T>(images, g, 0, slot, get(&st.rank[slot]), coef, use_all, sh);
        }
    }
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------
// per-tile stage B ("stain"): angle percentiles -> HE_source -> pseudo-inverse; concentration brackets
// ------------------------------------------------------------------------------------------------
template <typename T, bool kSpread = false>
__device__ void stain_stage(const T* __restrict__ images, const Geometry& g, const Workspace& ws, int group, TileScratch* sh) {
    GroupState& st = ws.state[group];
    const GroupPixels gp = group_pixels(g, group);
    SX_STAMP(st, 6);
    reset_scratch(sh);
    PairPrefetch pf;
    if constexpr (!kSpread) prefetch_pair(pf, g, ws, group, 0);
    float sod[kKeys][3];
    load_sample(ws, group, g.sample_count, sod);
    float vecs[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) vecs[i] = get(&st.rec[0].coef[i]);
    const bool use_all = get(&st.rec[0].use_all) != 0;
    uint32_t phi_key[2];
    if constexpr (kSpread) {
        pf.ncand[0] = get(&ws.pool->ncand[0]);
        pf.ncand[1] = get(&ws.pool->ncand[1]);
        resolve_pair_spread<T>(images, g, ws, 0, vecs, use_all, phi_key, sh);
    } else {
        resolve_pair<T>(images, g, ws, group, 0, vecs, use_all, pf, phi_key, sh);
    }
    SX_STAMP(st, 7);
    if (threadIdx.x == 0) {
        float he[6], pinv[6];
        stain_vectors_and_pinv(vecs, phi_key[0], phi_key[1], he, pinv);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            put(&st.he[i], he[i]);
            put(&st.pinv[i], pinv[i]);
            sh->coef[i] = pinv[i];
        }
        put(&st.phi_key[0], phi_key[0]);       // phi itself is only worked out for sx_macenko_tile_params
        put(&st.phi_key[1], phi_key[1]);
        put(&st.ncand_seen[0], pf.ncand[0]);
        put(&st.ncand_seen[1], pf.ncand[1]);
    }
    __syncthreads();
    SX_STAMP(st, 8);
    // concentration brackets from the same sample (every pixel takes part: torch_backend.py:442-448)
    float pinv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) pinv[i] = sh->coef[i];
    const unsigned long long n_all = g.distributed ? (unsigned long long)g.n_all : (unsigned long long)gp.count;
    const unsigned long long k99 = nearest_rank_index(99.0, n_all);          // torch_backend.py:447-448
    uint32_t key[2][kKeys];
#pragma unroll
    for (int i = 0; i < kKeys; ++i) {
        const int j = i * kGroupThreads + (int)threadIdx.x;
        uint32_t ka = 0xFFFFFFFFu, kb = 0xFFFFFFFFu;
        if (j < g.sample_count) {
            float c0, c1;
            concentration(sod[i], pinv, c0, c1);
            ka = float_key(c0);
            kb = float_key(c1);
        }
        key[0][i] = ka;
        key[1][i] = kb;
        sh->keys[0][j] = ka;
        sh->keys[1][j] = kb;
    }
    const unsigned long long k0[2] = {k99, k99};
    uint32_t lo[2], hi[2];
    double origin[2], scale[2];
    sample_brackets<2>(sh, key, n_all, k0, lo, hi, origin, scale);
    SX_STAMP(st, 10);
    if (threadIdx.x == 0) {
        StageRecord* rec = &st.rec[1];
#pragma unroll
        for (int i = 0; i < 6; ++i) put(&rec->coef[i], pinv[i]);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            put(&st.rank[2 + s], k99);
            put(&rec->lo[s], lo[s]);
            put(&rec->hi[s], hi[s]);
            put(&rec->bin_origin[s], origin[s]);
            put(&rec->bin_scale[s], scale[s]);
        }
        put(&rec->use_all, 1);
    }
    if (g.fine_chunk) {      // the angle histogram has been consumed (prefetched at the top): clear it for the concentration pass
        uint32_t* row = ws.block_hist + (size_t)group * g.blocks_per_tile * 512;
        for (int i = threadIdx.x; i < 512; i += blockDim.x) put(&row[i], 0u);
    }
}

// ------------------------------------------------------------------------------------------------
// per-tile stage C ("scale"): concentration percentiles -> scale factors (transform) / outputs (fit)
// ------------------------------------------------------------------------------------------------
template <typename T, bool kSpread = false>
__device__ void scale_stage(const T* __restrict__ images, const Geometry& g, const Workspace& ws, int group, const float* __restrict__ target_max_conc, float* __restrict__ he_out,
                            float* __restrict__ max_c_out, TileScratch* sh) {
    GroupState& st = ws.state[group];
    SX_STAMP(st, 12);
    reset_scratch(sh);
    PairPrefetch pf;
    if constexpr (!kSpread) prefetch_pair(pf, g, ws, group, 2);
    float pinv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) pinv[i] = get(&st.rec[1].coef[i]);
    uint32_t c_key[2];
    if constexpr (kSpread) {
        pf.ncand[0] = get(&ws.pool->ncand[2]);
        pf.ncand[1] = get(&ws.pool->ncand[3]);
        resolve_pair_spread<T>(images, g, ws, 2, pinv, true, c_key, sh);
    } else {
        resolve_pair<T>(images, g, ws, group, 2, pinv, true, pf, c_key, sh);
    }
    SX_STAMP(st, 13);
    if (threadIdx.x == 0) {
        const float m0 = key_float(c_key[0]), m1 = key_float(c_key[1]);
        put(&st.max_c[0], m0);
        put(&st.max_c[1], m1);
        put(&st.ncand_seen[2], pf.ncand[2 - 2]);
        put(&st.ncand_seen[3], pf.ncand[3 - 2]);
        StageRecord* rec = &st.rec[2];
#pragma unroll
        for (int i = 0; i < 6; ++i) put(&rec->coef[i], pinv[i]);
        if (target_max_conc) {
            put(&rec->scale[0], target_max_conc[0] / m0);      // torch_backend.py:452
            put(&rec->scale[1], target_max_conc[1] / m1);
        }
        if (he_out) {
#pragma unroll
            for (int i = 0; i < 6; ++i) he_out[i] = get(&st.he[i]);
            max_c_out[0] = m0;
            max_c_out[1] = m1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// kernels, one launch per stage
// ------------------------------------------------------------------------------------------------
// (two-byte pixels: five waves per SIMD -- a 224 x 224 tile is five work items (set_chunk), 1280 for the batch that matters, and at the
// 106 registers the compiler would take, four workgroups per CU made that a second round)
template <typename T, int V, bool kInter = false>
__global__ __launch_bounds__(kStreamThreads, sizeof(T) <= 2 ? 5 : 1) void stats_kernel(const T* __restrict__ images, Geometry g, Workspace ws) {
    __shared__ StatsScratch<kStreamThreads> sh;
    __shared__ LevelTables<T> tb;
    tb.fill();
    if constexpr (Codable<T, V, kInter>::value) {
        if (g.code_epoch != 0u) {      // (uniform over the launch)
            __shared__ CodeTable<T, 4> ct;
            ct.fill();
            stats_item<T, V, kStreamThreads, kInter, true>(images, g, ws, blockIdx.x / g.blocks_per_tile, blockIdx.x % g.blocks_per_tile, blockIdx.x, &sh, tb, &ct);
            return;
        }
    }
    stats_item<T, V, kStreamThreads, kInter>(images, g, ws, blockIdx.x / g.blocks_per_tile, blockIdx.x % g.blocks_per_tile, blockIdx.x, &sh, tb);
}

template <typename T, int V, bool kConc, bool kInter = false>
__global__ __launch_bounds__(kStreamThreads) void bracket_kernel(const T* __restrict__ images, Geometry g, Workspace ws) {
    __shared__ BracketScratch<kStreamThreads> sh;
    const int per_tile = g.fine_chunk ? g.fine_blocks : g.blocks_per_tile;
    __shared__ LevelTables<T> tb;
    tb.fill();
    if constexpr (Codable<T, V, kInter>::value) {
        if (g.code_epoch != 0u && get(&ws.code_bad[blockIdx.x / per_tile]) != g.code_epoch) {      // the tile is 8-bit levels: its codes (uniform over the workgroup)
            __shared__ LevelTables<Coded<T>> ctb;
            ctb.fill();
            bracket_item<Coded<T>, 16, kConc, kStreamThreads, false>(reinterpret_cast<const Coded<T>*>(ws.codes), g, ws, blockIdx.x / per_tile, blockIdx.x % per_tile, blockIdx.x, &sh, ctb);
            return;
        }
    }
    bracket_item<T, V, kConc, kStreamThreads, kInter>(images, g, ws, blockIdx.x / per_tile, blockIdx.x % per_tile, blockIdx.x, &sh, tb);
}

template <typename T, typename O, int V, bool kUnit, bool kInter = false>
__global__ __launch_bounds__(kStreamThreads) void reconstruct_kernel(const T* __restrict__ images, O* __restrict__ out, Geometry g, Workspace ws, const float* __restrict__ stain_matrix) {
    const int per_tile = g.recon_chunk ? g.recon_blocks : (g.fine_chunk ? g.fine_blocks : g.blocks_per_tile);
    __shared__ LevelTables<T> tb;
    tb.fill();
    const unsigned item = blockIdx.x;      // (measured: reversing the order, so that the work items pass A touched last come first, changes nothing)
    if constexpr (Codable<T, V, kInter>::value && std::is_same<T, O>::value) {
        if (g.code_epoch != 0u && get(&ws.code_bad[item / per_tile]) != g.code_epoch) {      // the tile is 8-bit levels: its codes in, the same float pixels out
            __shared__ LevelTables<Coded<T>> ctb;
            ctb.fill();
            reconstruct_item<Coded<T>, O, V, kUnit, kStreamThreads, false>(reinterpret_cast<const Coded<T>*>(ws.codes), out, g, ws, item / per_tile, item % per_tile, stain_matrix, ctb);
            return;
        }
    }
    if constexpr (kInter && V > 1) {
        __shared__ uint4 stage[kStreamThreads * 3];      // 3 KB per wave: store_pixels_staged()
        reconstruct_item<T, O, V, kUnit, kStreamThreads, kInter>(images, out, g, ws, item / per_tile, item % per_tile, stain_matrix, tb, stage);
    } else {
        reconstruct_item<T, O, V, kUnit, kStreamThreads, kInter>(images, out, g, ws, item / per_tile, item % per_tile, stain_matrix, tb);
    }
}


template <typename T>
__global__ __launch_bounds__(kGroupThreads) void fast_kernel(const T* __restrict__ images, Geometry g, Workspace ws, const float* __restrict__ target_max_conc) {
    __shared__ TileScratch sh;
    plane_stage<T, true>(images, g, ws, blockIdx.x, 1, &sh, nullptr, target_max_conc);
}

template <typename T>
__global__ __launch_bounds__(kGroupThreads) void plane_kernel(const T* __restrict__ images, Geometry g, Workspace ws, int allow_fallback) {
    __shared__ TileScratch sh;
    plane_stage<T>(images, g, ws, blockIdx.x, allow_fallback, &sh);
}

template <typename T, bool kSpread = false>
__global__ __launch_bounds__(kGroupThreads) void stain_kernel(const T* __restrict__ images, Geometry g, Workspace ws) {
    __shared__ TileScratch sh;
    stain_stage<T, kSpread>(images, g, ws, blockIdx.x, &sh);
}

template <typename T, bool kSpread = false>
__global__ __launch_bounds__(kGroupThreads) void scale_kernel(const T* __restrict__ images, Geometry g, Workspace ws, const float* __restrict__ target_max_conc, float* __restrict__ he_out, float* __restrict__ max_c_out) {
    __shared__ TileScratch sh;
    scale_stage<T, kSpread>(images, g, ws, blockIdx.x, target_max_conc, he_out, max_c_out, &sh);
}

// ------------------------------------------------------------------------------------------------
// distributed pooled fit: the batch is sharded over ranks, every reduction stage is one
// small all-reduce done by the host between these kernels.  Order statistics use a plain 4-round byte
// radix select whose 256-bin histograms are integer sums over ranks => the same (HE, maxC) bits on
// every rank, whatever the sharding.
// ------------------------------------------------------------------------------------------------
struct alignas(256) DFitState {
    double mom[kMoments];
    float vecs[6], he[6], pinv[6];
    float phi[2], max_c[2];
    unsigned long long n_sel, n_all;
    unsigned long long rank[kSlots];
    uint32_t prefix[kSlots], mask[kSlots];
    int round[2];
};

// Sum of the work items' partial moments by one 256-thread workgroup in a fixed order (a function of nblk only): thread
// (part, k) adds rows part, part + 25, ... of moment k, thread k then adds the 25 partial sums in order.
__device__ __forceinline__ double reduce_partials(const double* __restrict__ partial, int64_t nblk) {
    __shared__ double part_sum[25][kPartial];
    const int k = threadIdx.x % kPartial, part = threadIdx.x / kPartial;
    if (part < 25) {
        double s = 0.0;
        for (int64_t b = part; b < nblk; b += 25) s += partial[b * kPartial + k];
        part_sum[part][k] = s;
    }
    __syncthreads();
    double total = 0.0;
    if (threadIdx.x < kPartial)
        for (int p = 0; p < 25; ++p) total += part_sum[p][threadIdx.x];
    return total;      // valid in threads 0..9
}

__global__ __launch_bounds__(256) void dfit_reduce_partials_kernel(const double* __restrict__ partial, int64_t nblk, double n_all, double* __restrict__ moments) {
    const double s = reduce_partials(partial, nblk);
    const int k = threadIdx.x;
    if (k < kPartial) moments[k] = s;
    else if (k == kPartial) moments[k] = n_all;        // all-pixel set: only its count is used by the distributed fit (no fallback there)
    else if (k < kMoments) moments[k] = 0.0;
}

__global__ void dfit_begin_kernel(const double* __restrict__ moments, DFitState* __restrict__ st) {
    if (threadIdx.x != 0) return;
    for (int k = 0; k < kMoments; ++k) st->mom[k] = moments[k];
    double cov[9];
    bool use_all;
    unsigned long long n_sel;
    plane_from_moments(st->mom, false, cov, st->vecs, use_all, n_sel);
    st->n_sel = n_sel;
    st->n_all = (unsigned long long)st->mom[10];
    st->rank[0] = nearest_rank_index(1.0, n_sel);
    st->rank[1] = nearest_rank_index(99.0, n_sel);
    for (int s = 0; s < kSlots; ++s) st->prefix[s] = st->mask[s] = 0;
    st->round[0] = st->round[1] = 0;
}

template <typename T, int V>
__global__ __launch_bounds__(kStreamThreads) void dfit_histogram_kernel(const T* __restrict__ images, Geometry g, const DFitState* __restrict__ st, int stage, unsigned long long* __restrict__ hist) {
    const int64_t tile = blockIdx.x / g.blocks_per_tile;
    const int chunk_id = blockIdx.x % g.blocks_per_tile;
    const int64_t p_begin = (int64_t)chunk_id * g.chunk, p_end = min(p_begin + (int64_t)g.chunk, g.pixels);
    const T* img = images + tile * 3 * g.pixels;
    __shared__ uint32_t local[2][256];
    for (int i = threadIdx.x; i < 512; i += kStreamThreads) (&local[0][0])[i] = 0;
    __syncthreads();
    const int s0 = stage * 2;
    float coef[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) coef[i] = stage ? st->pinv[i] : st->vecs[i];
    const int shift = 24 - 8 * st->round[stage];
    const uint32_t pa = st->prefix[s0], ma = st->mask[s0], pb = st->prefix[s0 + 1], mb = st->mask[s0 + 1];
    for (int64_t p = p_begin + (int64_t)threadIdx.x * V; p < p_end; p += (int64_t)kStreamThreads * V) {
        float u[3][V];
        load_pixels<T, V, false>(img, g.pixels, p, u);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            float od[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) od[c] = optical_density<T>(u[c][i]);
            uint32_t ka, kb;
            if (stage) {
                float c0, c1;
                concentration(od, coef, c0, c1);
                ka = float_key(c0);
                kb = float_key(c1);
            } else {
                if (!od_selected(od, false)) continue;
                ka = kb = angle_key(od, coef);
            }
            if (((ka ^ pa) & ma) == 0) atomicAdd(&local[0][(ka >> shift) & 255u], 1u);
            if (((kb ^ pb) & mb) == 0) atomicAdd(&local[1][(kb >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 512; i += kStreamThreads) {
        const uint32_t v = (&local[0][0])[i];
        if (v) atomicAdd(&hist[i], (unsigned long long)v);
    }
}

__global__ __launch_bounds__(128) void dfit_advance_kernel(DFitState* __restrict__ st, int stage, const unsigned long long* __restrict__ hist) {
    const int s0 = stage * 2;
    // bins can exceed 2^32 in principle: scan in 64 bit with one thread per slot
    if (threadIdx.x < 2) {
        const int j = threadIdx.x;
        unsigned long long r = st->rank[s0 + j], cum = 0;
        int d = 0;
        for (; d < 255; ++d) {
            const unsigned long long h = hist[j * 256 + d];
            if (cum + h > r) break;
            cum += h;
        }
        const int shift = 24 - 8 * st->round[stage];
        st->prefix[s0 + j] |= (uint32_t)d << shift;
        st->mask[s0 + j] |= 0xFFu << shift;
        st->rank[s0 + j] = r - cum;
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    st->round[stage] += 1;
    if (st->round[stage] < 4) return;
    if (stage == 0) {
        const float phi_lo = angle_from_key(st->prefix[0]), phi_hi = angle_from_key(st->prefix[1]);
        st->phi[0] = phi_lo;
        st->phi[1] = phi_hi;
        stain_vectors_and_pinv(st->vecs, st->prefix[0], st->prefix[1], st->he, st->pinv);
        st->rank[2] = st->rank[3] = nearest_rank_index(99.0, st->n_all);
    } else {
        st->max_c[0] = key_float(st->prefix[2]);
        st->max_c[1] = key_float(st->prefix[3]);
    }
}

__global__ void dfit_result_kernel(const DFitState* __restrict__ st, float* __restrict__ he_out, float* __restrict__ max_c_out) {
    if (threadIdx.x < 6) he_out[threadIdx.x] = st->he[threadIdx.x];
    if (threadIdx.x < 2) max_c_out[threadIdx.x] = st->max_c[threadIdx.x];
}

__global__ void export_params_kernel(const GroupState* __restrict__ state, int64_t n_groups, float* __restrict__ out) {
    const int64_t gidx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (gidx >= n_groups) return;
    const GroupState& st = state[gidx];
    float* o = out + gidx * SX_MACENKO_PARAM_FLOATS;
    o[0] = (float)st.n_sel;
    o[1] = (float)st.use_all;
    for (int i = 0; i < 6; ++i) o[2 + i] = st.vecs[i];
    o[8] = angle_from_key(st.phi_key[0]);
    o[9] = angle_from_key(st.phi_key[1]);
    for (int i = 0; i < 6; ++i) o[10 + i] = st.he[i];
    o[16] = st.max_c[0];
    o[17] = st.max_c[1];
    o[18] = (float)(st.fell_back | ((st.spec >> 8) << 8));      // (two-pass form: bits 8.. say why a slot left the speculative path, 4 bits per slot)
    for (int s = 0; s < kSlots; ++s) o[19 + s] = (float)st.ncand_seen[s];
    for (int i = 0; i < 9; ++i) o[23 + i] = (float)st.cov[i];
    for (int i = 0; i < 16; ++i) o[32 + i] = (float)((double)(st.stamp[i] - st.stamp[0]) * 0.01);   // us (100 MHz clock)
}

}  // namespace macenko
}  // namespace sx
#include "macenko_twopass.hpp"
// Diagnostic builds only (-DSX_DIAG: stainx_amd/_lib/libstainx_diag.so, built next to the product by __graft_entry__.build()): two forms of the
// transform that were built, are exact and are SLOWER than the product's forms on every measured input -- kept as measured design studies
// (DESIGN.md sections 4c, 4e), not shipped -- and the flags that force the rare paths for tests.
#ifdef SX_DIAG
#include "macenko_fused.hpp"
#include "macenko_resident.hpp"
#endif
namespace sx {
namespace macenko {
static_assert(sizeof(PriorRecord) == kPriorRecordBytes, "workspace layout");

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
static bool aligned_for(const void* p, size_t bytes) { return (reinterpret_cast<uintptr_t>(p) % bytes) == 0; }

// The tile split evenly over its work items (a 224x224 tile is 3 x 16384 + 1024 pixels otherwise: a quarter of the
// workgroups nearly idle); a function of the tile size and the element type's pack only, so a tile's partial sums are
// grouped the same way whatever batch -- or entry point -- it arrives through.
// One work item MORE than the tile needs at 16384 pixels each, where that evens the items out: every work item of a streaming pass is
// resident at once (1024 of them on 256 CUs for the batches that matter), workgroup b lands on XCD b % 8, and a short last item per
// tile therefore leaves whole XCDs with short items only while the others set the kernel's time.  224 x 224 bf16: 3 x 14336 + 7168
// pixels -> 4 x 10240 + 9216 (57344 -> 51200 pixel-times per CU and sweep); 448 x 448: 13 items with a quarter-sized last one ->
// 14 x 14336.  Taken when it saves more than 3 %, on the tile sizes of even_items_size().
static void set_chunk(Geometry& g, bool may_add_item) {
    const int64_t unit = (int64_t)kStreamThreads * (g.vec ? g.vec_width : 1);
    const int b0 = (int)((g.pixels + kChunk - 1) / kChunk);
    int best_b = b0;
    int64_t best_chunk = 0, best_cost = 0;
    // (two-byte pixels only.  float32 / float64 tiles of these sizes take the two-pass form, whose pass A holds four workgroups per CU:
    // 1280 work items instead of 1024 are a second round there, 256 x 224 x 224 float32 163 -> 177 us.  uint8: a pack is 16 pixels, the
    // rounding unit 4096, and the moments pass needs 112 registers -- four workgroups per CU again; measured below)
    for (int b = b0; b <= b0 + (may_add_item && g.vec_width >= 8 && even_items_size(g.pixels) ? 1 : 0); ++b) {
        const int64_t even = (g.pixels + b - 1) / b;
        const int64_t chunk = std::min<int64_t>(kChunk, (even + unit - 1) / unit * unit);
        if ((int64_t)(b - 1) * chunk >= g.pixels) continue;      // (the last item would be empty)
        const int64_t cost = chunk * b;
        if (best_chunk == 0 || cost * 100 < best_cost * 97) {
            best_b = b;
            best_chunk = chunk;
            best_cost = cost;
        }
    }
#ifdef SX_STAMPS      // diagnostic builds: work items per tile from the environment
    if (const char* e = std::getenv("SX_ITEMS")) {
        const int b = std::atoi(e);
        const int64_t chunk = std::min<int64_t>(kChunk, ((g.pixels + b - 1) / b + unit - 1) / unit * unit);
        if (may_add_item && b >= b0 && b <= blocks_per_tile_for(g.pixels) && (int64_t)(b - 1) * chunk < g.pixels) {
            best_b = b;
            best_chunk = chunk;
        }
    }
#endif
    g.blocks_per_tile = best_b;
    g.chunk = (int)best_chunk;
}

// (may_add_item: set_chunk(); only the transform takes it -- the fused fit and the staged fits stay bit for bit the same)
static void set_sampling(Geometry& g, bool may_add_item = false) {
    const int64_t count = g.pooled ? g.n_tiles * g.pixels : g.pixels;
    int64_t stride = 1;                                   // smallest power of two with ceil(count/stride) <= kSample
    while ((count + stride - 1) / stride > kSample) stride *= 2;
    g.sample_stride = (int)stride;
    g.sample_count = (int)std::min<int64_t>(kSample, (count + stride - 1) / stride);
    g.spread = (g.pooled && g.n_tiles > 1) ? 1 : 0;
    set_chunk(g, may_add_item);
    g.cap = cap_for(g.spread ? g.pixels : count);
}

template <typename T, int V, bool kInter = false>
static int run_estimate(const T* images, const Geometry& g, const Workspace& ws, int n_groups, int allow_fallback, const float* tmc, float* he_out, float* max_c_out, hipStream_t stream) {
    const unsigned grid = (unsigned)(g.n_tiles * g.blocks_per_tile);
    const unsigned grid_b = (unsigned)(g.n_tiles * (g.fine_chunk ? g.fine_blocks : g.blocks_per_tile));      // bracket stages
    hipLaunchKernelGGL((stats_kernel<T, V, kInter>), dim3(grid), dim3(kStreamThreads), 0, stream, images, g, ws);
    hipLaunchKernelGGL((plane_kernel<T>), dim3(n_groups), dim3(kGroupThreads), 0, stream, images, g, ws, allow_fallback);
    hipLaunchKernelGGL((bracket_kernel<T, V, false, kInter>), dim3(grid_b), dim3(kStreamThreads), 0, stream, images, g, ws);
    if (g.spread) {
        hipLaunchKernelGGL(pool_reduce_kernel, dim3((unsigned)g.n_tiles), dim3(512), 0, stream, g, ws, 0);
        hipLaunchKernelGGL(pool_gather_kernel, dim3((unsigned)g.n_tiles), dim3(kGroupThreads), 0, stream, g, ws, 0);
        hipLaunchKernelGGL((stain_kernel<T, true>), dim3(1), dim3(kGroupThreads), 0, stream, images, g, ws);
    } else {
        hipLaunchKernelGGL((stain_kernel<T>), dim3(n_groups), dim3(kGroupThreads), 0, stream, images, g, ws);
    }
    hipLaunchKernelGGL((bracket_kernel<T, V, true, kInter>), dim3(grid_b), dim3(kStreamThreads), 0, stream, images, g, ws);
    if (g.spread) {
        hipLaunchKernelGGL(pool_reduce_kernel, dim3((unsigned)g.n_tiles), dim3(512), 0, stream, g, ws, 1);
        hipLaunchKernelGGL(pool_gather_kernel, dim3((unsigned)g.n_tiles), dim3(kGroupThreads), 0, stream, g, ws, 1);
        hipLaunchKernelGGL((scale_kernel<T, true>), dim3(1), dim3(kGroupThreads), 0, stream, images, g, ws, tmc, he_out, max_c_out);
    } else {
        hipLaunchKernelGGL((scale_kernel<T>), dim3(n_groups), dim3(kGroupThreads), 0, stream, images, g, ws, tmc, he_out, max_c_out);
    }
    return check_launch("macenko estimate");
}

// The estimate of the two-pass transform (macenko_twopass.hpp): prior, ONE pass over the input, two small stages.
template <typename T, int V, bool kInter = false>
static int run_two_pass(const T* images, const Geometry& g, const Workspace& ws, const float* tmc, hipStream_t stream, void* key_scratch = nullptr) {
    const unsigned n = (unsigned)g.n_tiles, grid = (unsigned)(g.n_tiles * g.blocks_per_tile);
#ifndef SX_DIAG
    // The product library builds the two-pass form for what its default serves: whole 16-byte packs (V > 1) and the dense candidate
    // records (tiles up to 512 x 512) -- sx_macenko_form() and transform_typed() send everything else to the four passes.  The
    // per-wave candidate segments, the scalar pass A and the unaligned prior (0.9 MB of code) are in the diagnostic build only.
    if constexpr (V == 1) {
        return fail(SX_ERR_BAD_ARG, "the two-pass form needs 16-byte aligned tiles of whole packs");
    } else {
        if (!g.dense) return fail(SX_ERR_BAD_ARG, "the two-pass form serves tiles of up to 512 x 512 pixels in this build");
        hipLaunchKernelGGL((prior_kernel<T, true, kInter>), dim3(n), dim3(kGroupThreads), 0, stream, images, g, ws);
        hipLaunchKernelGGL((pass_a_kernel<T, V, kInter, true>), dim3(grid), dim3(kStreamThreads), 0, stream, images, g, ws);
        hipLaunchKernelGGL((estimate_stage_kernel<T, true>), dim3(2 * n), dim3(kGroupThreads), 0, stream, images, g, ws, tmc, static_cast<uint32_t*>(key_scratch));
        return check_launch("macenko two-pass estimate");
    }
#else
    const bool quads = (g.pixels % 4 == 0) && aligned_for(images, 4 * sizeof(T));
    if (quads)
        hipLaunchKernelGGL((prior_kernel<T, true, kInter>), dim3(n), dim3(kGroupThreads), 0, stream, images, g, ws);
    else
        hipLaunchKernelGGL((prior_kernel<T, false, kInter>), dim3(n), dim3(kGroupThreads), 0, stream, images, g, ws);
    if (g.dense) {
        hipLaunchKernelGGL((pass_a_kernel<T, V, kInter, true>), dim3(grid), dim3(kStreamThreads), 0, stream, images, g, ws);
        hipLaunchKernelGGL((estimate_stage_kernel<T, true>), dim3(2 * n), dim3(kGroupThreads), 0, stream, images, g, ws, tmc, static_cast<uint32_t*>(key_scratch));
    } else {
        hipLaunchKernelGGL((pass_a_kernel<T, V, kInter>), dim3(grid), dim3(kStreamThreads), 0, stream, images, g, ws);
        hipLaunchKernelGGL((estimate_stage_kernel<T>), dim3(2 * n), dim3(kGroupThreads), 0, stream, images, g, ws, tmc, static_cast<uint32_t*>(key_scratch));
    }
    return check_launch("macenko two-pass estimate");
#endif
}

#ifdef SX_DIAG
// The fused two-pass transform (macenko_fused.hpp): the prior, then pass A + stage jobs + reconstruct items in ONE launch.
template <typename T, typename O, int V>
static int run_fused(const T* images, O* out, const Geometry& g, const Workspace& ws, const float* sm, const float* tmc, bool unit, hipStream_t stream) {
    const unsigned n = (unsigned)g.n_tiles, units = (unsigned)(2 * g.fused_items) + 2u * n;
    hipLaunchKernelGGL((prior_kernel<T, true, false>), dim3(n), dim3(kGroupThreads), 0, stream, images, g, ws);
    if (unit)
        hipLaunchKernelGGL((fused_kernel<T, O, V, true>), dim3(units), dim3(kStreamThreads), 0, stream, images, out, g, ws, sm, tmc);
    else
        hipLaunchKernelGGL((fused_kernel<T, O, V, false>), dim3(units), dim3(kStreamThreads), 0, stream, images, out, g, ws, sm, tmc);
    return check_launch("macenko fused transform");
}

#endif

template <typename T, typename O, int V, bool kInter = false>
static int run_transform(const T* images, O* out, const Geometry& g, const Workspace& ws, const float* sm, const float* tmc, bool unit, hipStream_t stream) {
    const unsigned items = (unsigned)(g.n_tiles * (g.recon_chunk ? g.recon_blocks : (g.fine_chunk ? g.fine_blocks : g.blocks_per_tile)));      // reconstruct work items
    int rc = SX_OK;
    if (g.fast) {
        hipLaunchKernelGGL((stats_kernel<T, V, kInter>), dim3((unsigned)(g.n_tiles * g.blocks_per_tile)), dim3(kStreamThreads), 0, stream, images, g, ws);
        hipLaunchKernelGGL((fast_kernel<T>), dim3((unsigned)g.n_tiles), dim3(kGroupThreads), 0, stream, images, g, ws, tmc);
        rc = check_launch("macenko fast estimate");
    } else if (g.two_pass) {
        // (the output tile doubles as scratch for the keys of a slot that takes the slow exact path: nothing has been written there
        // yet, the reconstruct launch overwrites it -- needs a plane of 4-byte elements per slot pair)
        // (float64 tiles take the four passes: sx_macenko_form never sends them here, and their two-pass kernels -- 0.4 MB of code -- are not built)
        if constexpr (sizeof(T) == 8) rc = fail(SX_ERR_DTYPE, "the two-pass form is not built for float64 tiles");
        else rc = run_two_pass<T, V, kInter>(images, g, ws, tmc, stream, sizeof(O) >= 4 ? static_cast<void*>(out) : nullptr);
    } else {
        rc = run_estimate<T, V, kInter>(images, g, ws, (int)g.n_tiles, 1, tmc, nullptr, nullptr, stream);
    }
    if (rc != SX_OK) return rc;
    // The reconstruct pass is paced by its stores: its pack is 16 bytes of OUTPUT per lane and plane, so that one store
    // instruction of a wave writes 1 KB of consecutive bytes.  With the input's pack (16 pixels per lane for uint8) a wider output
    // type leaves every lane 64-192 bytes of its own and every store instruction 64 different lines: uint8 -> float32 (/255)
    // took 172 us planar and 569 us NHWC for the config-2 batch where the same-width uint8 output takes 24 us.
    constexpr int VR = V == 1 ? 1 : ((int)(16 / sizeof(O)) < V ? (int)(16 / sizeof(O)) : V);
    // Big batches: ONE pack set per thread and work item (no loop: 16384 small workgroups for config 2 instead of 1024 of sixteen
    // sweeps -- the pass has no per-item state, and with 1024 workgroups only half the waves a CU can hold were resident: call
    // 159.2 -> 156.1 us, tools/ab_recon_chunk.py).  Small batches keep their own split (fine_chunk).
    Geometry gr = g;
    unsigned items_r = items;
    if (!g.recon_chunk && !g.fine_chunk && V > 1 && g.n_tiles * g.pixels >= (1ll << 22)) {
        gr.recon_chunk = kStreamThreads * VR;
        gr.recon_blocks = (int)((g.pixels + gr.recon_chunk - 1) / gr.recon_chunk);
        items_r = (unsigned)(g.n_tiles * gr.recon_blocks);
    }
    if (unit)
        hipLaunchKernelGGL((reconstruct_kernel<T, O, VR, true, kInter>), dim3(items_r), dim3(kStreamThreads), 0, stream, images, out, gr, ws, sm);
    else
        hipLaunchKernelGGL((reconstruct_kernel<T, O, VR, false, kInter>), dim3(items_r), dim3(kStreamThreads), 0, stream, images, out, gr, ws, sm);
    return check_launch("macenko reconstruct");
}

#ifdef SX_DIAG
// ---- the tile-resident form (macenko_resident.hpp) ---------------------------------------------------------------------------
// Compute units of the current device (the launch must be resident at once: one 1024-thread workgroup per CU); 256 when no device
// can be asked (workspace sizing on a host without one).
static int device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}
static int elem_bytes(int dtype) { return dtype == SX_U8 ? 1 : (dtype == SX_F16 || dtype == SX_BF16) ? 2 : dtype == SX_F32 ? 4 : 8; }
// Whether a call can take the resident form at all: planar tiles of whole 16-byte packs, one to sixteen workgroups per tile,
// the output of the input's type (uint8 with normalize_to_0_1: float32).
static bool resident_able(int dtype, int64_t n, int64_t pixels, unsigned flags) {
    if (dtype != SX_U8 && dtype != SX_F16 && dtype != SX_BF16 && dtype != SX_F32) return false;
    if (flags & (SX_MACENKO_SAMPLED | SX_MACENKO_CHANNELS_LAST | SX_MACENKO_OUT_BF16 | SX_MACENKO_OUT_F16)) return false;
    const int64_t pack = 16 / elem_bytes(dtype);
    if (pixels % pack != 0 || pixels < 1024) return false;
    const int64_t items = (pixels + kChunk - 1) / kChunk, group = (items + kResQuads - 1) / kResQuads;
    return group <= kResMaxGroup && group <= device_cus() && n >= 1;
}
static ResGeom resident_geometry(int dtype, int64_t n, int64_t pixels, bool unit) {
    Geometry g = make_geometry(n, pixels, 0);
    g.vec = 1;
    g.vec_width = 16 / elem_bytes(dtype);
    set_chunk(g, false);
    ResGeom rg{};
    rg.n_tiles = n;
    rg.pixels = pixels;
    rg.items = g.blocks_per_tile;
    rg.chunk = g.chunk;
    rg.group = (rg.items + kResQuads - 1) / kResQuads;
    const int cus = device_cus();
    int fit = std::max(cus / rg.group, 1);      // tiles whose workgroups are resident together
    if (fit >= 8 && n >= 8) {
        fit = fit / 8 * 8;
        rg.tiles_per_round = (int)std::min<int64_t>((n + 7) / 8 * 8, fit);
        rg.xcd_map = 1;
    } else {
        rg.tiles_per_round = (int)std::min<int64_t>(n, fit);
        rg.xcd_map = 0;
    }
    rg.rounds = (int)((n + rg.tiles_per_round - 1) / rg.tiles_per_round);
    rg.unit = unit ? 1 : 0;
    rg.spin_limit = 1u << 22;      // (seconds)
    return rg;
}
template <typename T, typename O>
static int run_resident(const void* images, void* out, const ResGeom& rg, const ResWork& rw, const float* sm, const float* tmc, hipStream_t stream) {
    constexpr int W = PackOf<T>::n;
    if (rg.group > 1 && hipMemsetAsync(rw.sync, 0, sizeof(ResSync) * (size_t)rg.n_tiles, stream) != hipSuccess) return fail(SX_ERR_LAUNCH, "macenko resident: clearing the arrival counters failed");
    const unsigned grid = (unsigned)(rg.tiles_per_round * rg.group);
    hipLaunchKernelGGL((resident_kernel<T, O, W>), dim3(grid), dim3(kResThreads), 0, stream, static_cast<const T*>(images), static_cast<O*>(out), rg, rw, sm, tmc);
    return check_launch("macenko resident transform");
}
static int resident_transform(const void* images, void* out, int dtype, int64_t n, int64_t pixels, void* ws_ptr, const float* sm, const float* tmc, bool unit, hipStream_t stream) {
    const ResGeom rg = resident_geometry(dtype, n, pixels, unit);
    const ResWork rw = carve_resident(ws_ptr, n, pixels);
    switch (dtype) {
        case SX_U8: return unit ? run_resident<uint8_t, float>(images, out, rg, rw, sm, tmc, stream) : run_resident<uint8_t, uint8_t>(images, out, rg, rw, sm, tmc, stream);
        case SX_F16: return run_resident<__half, __half>(images, out, rg, rw, sm, tmc, stream);
        case SX_BF16: return run_resident<__hip_bfloat16, __hip_bfloat16>(images, out, rg, rw, sm, tmc, stream);
        case SX_F32: return run_resident<float, float>(images, out, rg, rw, sm, tmc, stream);
        default: return fail(SX_ERR_DTYPE, "the resident form takes uint8 / float16 / bfloat16 / float32 tiles");
    }
}

#endif

template <typename T>
static int transform_typed(const void* images, void* out, const Geometry& g0, const Workspace& ws, const float* sm, const float* tmc, bool unit, hipStream_t stream) {
    Geometry g = g0;
    const bool u8_half = sizeof(T) == 1 && g.out_code != 0;      // uint8 in, bf16 / f16 out (an extension: SURVEY.md 8f-2)
    const bool u8_unit = unit && sizeof(T) == 1 && !u8_half;
    const size_t out_elem = u8_half ? 2 : (u8_unit ? sizeof(float) : sizeof(T));
    constexpr int W = PackOf<T>::n;
    const bool vec = (g.pixels % W == 0) && aligned_for(images, 16) && aligned_for(out, out_elem * W);
    g.vec = vec ? 1 : 0;
    g.vec_width = W;
    set_sampling(g, true);
    if (g.fused && !(vec && !g.interleaved && !u8_half && std::is_same<T, float>::value)) {      // (unaligned pointers: the four-pass form serves them; its workspace is a prefix of the fused one)
        g.fused = 0;
        g.two_pass = 0;
    }
#ifndef SX_DIAG
    if (g.two_pass && !vec) g.two_pass = 0;      // (unaligned pointers: the four passes serve them; their workspace is a prefix of the two-pass one)
#endif
    if (!(vec && !g.interleaved && std::is_same<T, float>::value)) g.code_epoch = 0u;      // (the coded passes: planar float32 tiles in 16-byte packs)
    if (g.two_pass) {
        g.dense = (fused_size(g.pixels) && !g.fused) ? 1 : 0;
        g.spec_kw = kSpecKw;
        g.spec_eff_far = kSpecEffFar;
        g.spec_eff_near = kSpecEffNear;
        g.spec_rot = kSpecRot;
        g.spec_sigmas = kSpecSigmas;
        g.spec_sigmas_conc = kSpecSigmasConc;
        g.spec_tscale = kSpecThresholdScale;
#ifdef SX_STAMPS      // diagnostic builds only: the knobs from the environment (tools/sweep_real.py under tools/tune_spec.sh)
        auto knob = [](const char* name, float& v) { if (const char* e = std::getenv(name)) v = (float)std::atof(e); };
        knob("SX_SPEC_KW", g.spec_kw);
        knob("SX_SPEC_EFF_FAR", g.spec_eff_far);
        knob("SX_SPEC_EFF_NEAR", g.spec_eff_near);
        knob("SX_SPEC_ROT", g.spec_rot);
        knob("SX_SPEC_SIGMAS", g.spec_sigmas);
        knob("SX_SPEC_SIGMAS_CONC", g.spec_sigmas_conc);
        knob("SX_SPEC_TSCALE", g.spec_tscale);
#endif
        g.fused_cap = g.fused ? fused_cap_for(g.pixels) : dense_cap_for(g.pixels);      // (records per tile and slot: the fused launch's, or the four launches' dense arrays)
        g.fused_items = (int)(g.n_tiles * g.blocks_per_tile);
        g.cap2 = cap2_for(g.pixels);
        g.n_seg = even_items_size(g.pixels) ? g.blocks_per_tile * (kStreamThreads / kWave) : two_pass_segments(g.pixels);      // (dense records there: no segments; the count of pass-A waves per tile all the same)
        g.seg_cap = seg_cap_for(g.pixels);
        g.over_cap = over_cap_for(g.pixels);
        const int64_t n_sectors = g.pixels / 16;
        g.prior_units = (int)std::min<int64_t>(std::max<int64_t>(n_sectors / 4, std::min<int64_t>(n_sectors, 64)), kPriorUnitsMax);
        g.prior_step_q16 = (unsigned)((n_sectors << 16) / g.prior_units);
    }
    // Small batches: with 16384-pixel work items a single 512x512 tile is 16 workgroups on 256 CUs and a bracket pass takes
    // 15 us of pure latency.  The bracket and reconstruct stages (integer counts / independent pixels: the split cannot
    // change a bit of the result) then use smaller work items -- at least two sweeps of a workgroup, aiming at ~1024 work
    // items; the moments stage keeps its fixed 16384-pixel grouping so that a tile's covariance has the same bits
    // whatever batch it arrives in.
    if (g.n_tiles * (int64_t)g.blocks_per_tile <= 32 && !g.fused) {      // (the fused launch's reconstruct items are its pass-A items) (measured: 1 tile 89 -> 78 us, 2 tiles 90 -> 81 us; from 4 tiles on the fixed cost per work item eats the gain)
        const int64_t floor_px = (int64_t)kStreamThreads * (vec ? W : 1) * 2;
        int64_t chunk = 2048;
        while (chunk < floor_px) chunk *= 2;
        while (chunk * 2 < kChunk && g.n_tiles * ((g.pixels + chunk - 1) / chunk) > 1024) chunk *= 2;
        if (chunk < kChunk) {
            g.fine_chunk = (int)chunk;
            g.fine_blocks = (int)((g.pixels + chunk - 1) / chunk);
        }
    }
#ifdef SX_STAMPS      // diagnostic builds: the reconstruct stage's work-item size from the environment (tools: A/B of its grid)
    if (const char* e = std::getenv("SX_RECON_CHUNK")) {
        const int64_t c = std::atoll(e);
        if (c > 0 && !g.fused && c % ((int64_t)kStreamThreads * (vec ? W : 1)) == 0) {
            g.recon_chunk = (int)c;
            g.recon_blocks = (int)((g.pixels + c - 1) / c);
        }
    }
#endif
    const T* in = static_cast<const T*>(images);
    if constexpr (sizeof(T) == 1) {
        if (u8_half) {
#define SX_RUN_HALF(O)                                                                                                                                          \
    return g.interleaved ? (vec ? run_transform<T, O, W, true>(in, static_cast<O*>(out), g, ws, sm, tmc, unit, stream)                                          \
                                : run_transform<T, O, 1, true>(in, static_cast<O*>(out), g, ws, sm, tmc, unit, stream))                                         \
                         : (vec ? run_transform<T, O, W>(in, static_cast<O*>(out), g, ws, sm, tmc, unit, stream)                                                \
                                : run_transform<T, O, 1>(in, static_cast<O*>(out), g, ws, sm, tmc, unit, stream));
            if (g.out_code == SX_BF16) {
                SX_RUN_HALF(__hip_bfloat16)
            } else {
                SX_RUN_HALF(__half)
            }
#undef SX_RUN_HALF
        }
    }
    if (g.interleaved) {      // (N,H,W,3): its own instantiations, so the planar kernels carry no trace of it
        if constexpr (sizeof(T) == 1) {
            if (u8_unit) {
                return vec ? run_transform<T, float, W, true>(in, static_cast<float*>(out), g, ws, sm, tmc, true, stream)
                           : run_transform<T, float, 1, true>(in, static_cast<float*>(out), g, ws, sm, tmc, true, stream);
            }
        }
        return vec ? run_transform<T, T, W, true>(in, static_cast<T*>(out), g, ws, sm, tmc, unit, stream)
                   : run_transform<T, T, 1, true>(in, static_cast<T*>(out), g, ws, sm, tmc, unit, stream);
    }
    if constexpr (sizeof(T) == 1) {
        if (u8_unit) {
            return vec ? run_transform<T, float, W>(in, static_cast<float*>(out), g, ws, sm, tmc, true, stream)
                       : run_transform<T, float, 1>(in, static_cast<float*>(out), g, ws, sm, tmc, true, stream);
        }
    }
#ifdef SX_DIAG
    if constexpr (std::is_same<T, float>::value) {
        if (g.fused) return run_fused<T, T, W>(in, static_cast<T*>(out), g, ws, sm, tmc, unit, stream);
    }
#endif
    return vec ? run_transform<T, T, W>(in, static_cast<T*>(out), g, ws, sm, tmc, unit, stream)
               : run_transform<T, T, 1>(in, static_cast<T*>(out), g, ws, sm, tmc, unit, stream);
}

template <typename T>
static int fit_typed(const void* images, const Geometry& g0, const Workspace& ws, float* he_out, float* max_c_out, hipStream_t stream) {
    Geometry g = g0;
    constexpr int W = PackOf<T>::n;
    const bool vec = (g.pixels % W == 0) && aligned_for(images, 16);
    g.vec = vec ? 1 : 0;
    g.vec_width = W;
    set_sampling(g);
    if (!(vec && std::is_same<T, float>::value)) g.code_epoch = 0u;      // (the coded passes: planar float32 tiles in 16-byte packs)
    const T* in = static_cast<const T*>(images);
    return vec ? run_estimate<T, W>(in, g, ws, 1, 0, nullptr, he_out, max_c_out, stream)
               : run_estimate<T, 1>(in, g, ws, 1, 0, nullptr, he_out, max_c_out, stream);
}

// ---- distributed pooled fit on the bracket machinery (sx_macenko_pfit_*) ---------------------------------
constexpr int kPoolSums = 1024 + 2 * kSlots + 1;      // hist[2][2][256], below[4], ncand[4], overflow
static_assert(kPoolSums == SX_PFIT_SUMS, "header constant");

// local partial moments -> 10 doubles; the local sample (group 0) -> caller buffer
__global__ __launch_bounds__(256) void pfit_export_stats_kernel(Workspace ws, int64_t nblk, double* __restrict__ moments_out, float* __restrict__ sample_out, long long* __restrict__ tiles_out = nullptr, long long tiles = 0) {
    if (blockIdx.x == 0) {
        const double s = reduce_partials(ws.partial, nblk);
        if (threadIdx.x < kPartial) moments_out[threadIdx.x] = s;
        if (tiles_out && threadIdx.x == 0) tiles_out[0] = tiles;      // (the packed record: the tile count leads it)
    } else {
        for (int i = (blockIdx.x - 1) * blockDim.x + threadIdx.x; i < 3 * kSample; i += (gridDim.x - 1) * blockDim.x) sample_out[i] = ws.sample_od[i];
    }
}

__global__ void pfit_import_sample_kernel(Workspace ws, const float* __restrict__ sample_union) {
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 3 * kSample; i += gridDim.x * blockDim.x) ws.sample_od[i] = sample_union[i];
}

__global__ __launch_bounds__(kGroupThreads) void pfit_plane_kernel(Geometry g, Workspace ws, const double* __restrict__ moments) {
    __shared__ TileScratch sh;
    plane_stage<float>(nullptr, g, ws, 0, 0, &sh, moments);      // no fallback in a fit: the pixels are never touched
}

// A stage's work-item histograms and per-tile counters added up over the rank's tiles, straight into the record the all-reduce takes
// (what pool_reduce_kernel + pfit_export_sums_kernel did in two launches through the pool): workgroup b adds bins 16 b ... 16 b + 15 of
// the stage's 512 over all work items (16 lanes of items per bin), workgroup 0 also the counters.  The other stage's entries are
// zero (nobody reads them); the pooled candidate lists of the stage start empty.
constexpr int kReduceBins = 16, kReduceLanes = 16;
__global__ __launch_bounds__(kReduceBins * kReduceLanes) void pfit_reduce_export_kernel(Geometry g, Workspace ws, int stage, long long* __restrict__ sums_out) {
    __shared__ uint32_t part[kReduceLanes][kReduceBins];
    const int bin = blockIdx.x * kReduceBins + (threadIdx.x % kReduceBins), lane = threadIdx.x / kReduceBins;
    const int64_t items = g.n_tiles * g.blocks_per_tile;
    const uint32_t* src = ws.block_hist + bin;
    uint32_t sum = 0;
#pragma unroll 8
    for (int64_t r = lane; r < items; r += kReduceLanes) sum += get(&src[(size_t)r * 512]);
    part[lane][threadIdx.x % kReduceBins] = sum;
    __syncthreads();
    if (threadIdx.x < kReduceBins) {
        unsigned long long total = 0;
#pragma unroll
        for (int l = 0; l < kReduceLanes; ++l) total += part[l][threadIdx.x];
        sums_out[stage * 512 + bin] = (long long)total;
        sums_out[(1 - stage) * 512 + bin] = 0;
    }
    if (blockIdx.x == 0) {
        __shared__ unsigned long long below_s[2], ncand_s[2];
        __shared__ uint32_t over_s;
        if (threadIdx.x < 2) below_s[threadIdx.x] = ncand_s[threadIdx.x] = 0ull;
        if (threadIdx.x == 0) over_s = get(&ws.pool->overflow);
        __syncthreads();
        for (int64_t i = threadIdx.x; i < 2 * g.n_tiles; i += blockDim.x) {
            const int j = (int)(i & 1), slot = 2 * stage + j;
            const int64_t tile = i >> 1;
            const uint32_t n = get(&ws.state[tile].ncand[slot]), b = get(&ws.state[tile].below[slot]);
            if (b) atomicAdd(&below_s[j], (unsigned long long)b);
            atomicAdd(&ncand_s[j], (unsigned long long)min(n, g.cap));
            if (n > g.cap) atomicOr(&over_s, 1u << slot);
        }
        __syncthreads();
        if (threadIdx.x < kSlots) {
            const int slot = threadIdx.x, j = slot - 2 * stage;
            sums_out[1024 + slot] = (j == 0 || j == 1) ? (long long)below_s[j] : 0;
            sums_out[1024 + kSlots + slot] = (j == 0 || j == 1) ? (long long)ncand_s[j] : 0;
        }
        if (threadIdx.x == 0) {
            sums_out[1024 + 2 * kSlots] = (long long)over_s;
            put(&ws.pool->compact_n[2 * stage], 0u);
            put(&ws.pool->compact_n[2 * stage + 1], 0u);
        }
    }
}

__global__ void pfit_import_sums_kernel(Workspace ws, const long long* __restrict__ sums, int stage) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(ws.pool);
    for (int i = threadIdx.x; i < 512; i += blockDim.x) dst[stage * 512 + i] = (uint32_t)sums[stage * 512 + i];
    if (threadIdx.x < 2) {
        const int slot = 2 * stage + threadIdx.x;
        ws.pool->below[slot] = (uint32_t)sums[1024 + slot];
        ws.pool->ncand[slot] = (uint32_t)sums[1024 + kSlots + slot];
    }
    if (threadIdx.x == 0) {
        ws.pool->overflow = sums[1024 + 2 * kSlots] ? 0xFu : 0u;
        ws.pool->compact_n[2 * stage] = ws.pool->compact_n[2 * stage + 1] = 0u;
    }
}

__global__ void pfit_export_compact_kernel(Workspace ws, int stage, int share, uint32_t* __restrict__ compact_out, int* __restrict__ counts_out) {
    for (int j = 0; j < 2; ++j) {
        const uint32_t n = ws.pool->compact_n[2 * stage + j];
        for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < min(n, (uint32_t)share); i += gridDim.x * blockDim.x) compact_out[(size_t)j * share + i] = ws.pool->compact[2 * stage + j][i];
        if (blockIdx.x == 0 && threadIdx.x == 0) counts_out[j] = (int)n;
    }
}

// the ranks' compact lists, gathered: [world][2][share] keys and [world][2] counts -> the union in the pool
__global__ void pfit_merge_kernel(Workspace ws, int stage, int world, int share, const uint32_t* __restrict__ gathered, const int* __restrict__ counts) {
    __shared__ uint32_t base[64][2];
    if (threadIdx.x < 2) {
        uint32_t run = 0;
        bool bad = false;
        for (int r = 0; r < world; ++r) {
            base[r][threadIdx.x] = run;
            const int n = counts[r * 2 + threadIdx.x];
            if (n > share) bad = true;
            run += (uint32_t)min(n, share);
        }
        ws.pool->compact_n[2 * stage + threadIdx.x] = run;
        if (bad || run > (uint32_t)kCompact) ws.pool->status = 1u;
    }
    __syncthreads();
    for (int r = 0; r < world; ++r)
        for (int j = 0; j < 2; ++j) {
            const int n = min(counts[r * 2 + j], share);
            for (int i = threadIdx.x; i < n; i += blockDim.x)
                if (base[r][j] + i < (uint32_t)kCompact) ws.pool->compact[2 * stage + j][base[r][j] + i] = gathered[((size_t)r * 2 + j) * share + i];
        }
}

__global__ void pfit_status_kernel(Workspace ws, int* __restrict__ status_out) {
    if (threadIdx.x == 0) status_out[0] = (int)(ws.pool->status | (ws.state[0].fell_back & 0xFu));
}

// ---- the same exchanges with the records packed and unpacked HERE (sx_macenko_pfit_*_packed) ----------------------------------
// What travels through a collective is one contiguous record per rank, written and read by these kernels: the host side of the
// unpacked steps did that with a dozen small tensor operations per exchange (cat, slices made contiguous, zero fills, comparisons,
// reductions), ~5 us of launch and gap each -- a third of a pooled fit_transform step.
//   stats record   [int64 tiles | 10 fp64 moments | 3 x 4096 fp32 sample]                      (kPfitStatsRecord bytes)
//   stage record   [int32 count_lo, count_hi, stale flag | 2 x share uint32 candidate keys]    (12 + 8 share bytes)
constexpr size_t kPfitStatsRecord = 8 + 8 * kPartial + 4 * 3 * (size_t)kSample;
static_assert(kPartial == 10 && kPfitStatsRecord == SX_PFIT_STATS_RECORD_BYTES, "header constant");
struct PfitRanks {
    int world;
    int sample_counts[64];      // valid columns of every rank's sample
};

// every rank's stats record -> the moments added up in rank order, the union sample (every world-th column of every rank's sample,
// ranks one after the other, cut at 4096; what stainx_amd/distributed.py assembled with slices), and "some rank's tile count is not
// the one the host took on trust"
// (the prologue of the plane stage's one workgroup: 12 sample values per thread, then the barrier behind which the stage reads them)
__global__ __launch_bounds__(kGroupThreads) void pfit_plane_packed_kernel(Geometry g, Workspace ws, const unsigned char* __restrict__ gathered, PfitRanks ranks, const long long* __restrict__ expected_tiles,
                                                                          int* __restrict__ stale_out, double* __restrict__ moments_out) {
    __shared__ TileScratch sh;
    if (threadIdx.x < kPartial) {
        double s = 0.0;
        for (int r = 0; r < ranks.world; ++r) s += reinterpret_cast<const double*>(gathered + (size_t)r * kPfitStatsRecord + 8)[threadIdx.x];
        moments_out[threadIdx.x] = s;
    }
    if (threadIdx.x == 64 && stale_out) {
        int stale = 0;
        if (expected_tiles)
            for (int r = 0; r < ranks.world; ++r) stale |= reinterpret_cast<const long long*>(gathered + (size_t)r * kPfitStatsRecord)[0] != expected_tiles[r];
        stale_out[0] = stale;
    }
    for (int i = threadIdx.x; i < 3 * kSample; i += blockDim.x) {
        const int ch = i / kSample, pos = i % kSample;
        float v = 0.0f;
        int base = 0;
        for (int r = 0; r < ranks.world; ++r) {
            const int len = (ranks.sample_counts[r] + ranks.world - 1) / ranks.world;      // columns 0, world, 2 world, ... below the rank's count
            if (pos < base + len) {
                v = reinterpret_cast<const float*>(gathered + (size_t)r * kPfitStatsRecord + 8 + 8 * kPartial)[(size_t)ch * kSample + (size_t)(pos - base) * ranks.world];
                break;
            }
            base += len;
        }
        ws.sample_od[i] = v;
    }
    __syncthreads();
    plane_stage<float>(nullptr, g, ws, 0, 0, &sh, moments_out);      // no fallback in a fit: the pixels are never touched
}

// the ranks' stage records, gathered -> the union of the candidates in the pool; any rank's stale flag -> bit 4 of the pool's status
// (the prologue of the group stage's workgroup: ends with a barrier, behind which the workgroup reads what it wrote)
__device__ __forceinline__ void pfit_merge_rows(const Workspace& ws, int stage, int world, int share, const unsigned* __restrict__ rows) {
    __shared__ uint32_t base[64][2];
    const size_t stride = 3 + 2 * (size_t)share;
    if (threadIdx.x < 2) {
        uint32_t run = 0;
        bool bad = false;
        for (int r = 0; r < world; ++r) {
            base[r][threadIdx.x] = run;
            const int n = (int)rows[r * stride + threadIdx.x];
            if (n > share) bad = true;
            run += (uint32_t)min(n, share);
        }
        ws.pool->compact_n[2 * stage + threadIdx.x] = run;
        if (bad || run > (uint32_t)kCompact) ws.pool->status = 1u;
    }
    if (threadIdx.x == 64) {
        unsigned stale = 0;
        for (int r = 0; r < world; ++r) stale |= rows[r * stride + 2];
        if (stale) atomicOr(&ws.pool->status, 16u);      // (bits 0-3: a bracket that did not hold / the slots that fell back)
    }
    __syncthreads();
    for (int r = 0; r < world; ++r)
        for (int j = 0; j < 2; ++j) {
            const int n = min((int)rows[r * stride + j], share);
            for (int i = threadIdx.x; i < n; i += blockDim.x)
                if (base[r][j] + i < (uint32_t)kCompact) ws.pool->compact[2 * stage + j][base[r][j] + i] = rows[r * stride + 3 + (size_t)j * share + i];
        }
    __syncthreads();
}

// The group stage of a distributed fit behind the exchange of the stage records: merge, then the stage itself, in one workgroup
// (stage 1 also leaves the status word: what pfit_status_kernel does for the unpacked steps).
__global__ __launch_bounds__(kGroupThreads) void pfit_stain_rows_kernel(Geometry g, Workspace ws, int world, int share, const unsigned* __restrict__ rows) {
    __shared__ TileScratch sh;
    pfit_merge_rows(ws, 0, world, share, rows);
    stain_stage<float, true>(nullptr, g, ws, 0, &sh);
}
__global__ __launch_bounds__(kGroupThreads) void pfit_scale_rows_kernel(Geometry g, Workspace ws, int world, int share, const unsigned* __restrict__ rows, float* __restrict__ he_out, float* __restrict__ max_c_out,
                                                                        int* __restrict__ status_out) {
    __shared__ TileScratch sh;
    pfit_merge_rows(ws, 1, world, share, rows);
    scale_stage<float, true>(nullptr, g, ws, 0, nullptr, he_out, max_c_out, &sh);
    __syncthreads();
    if (threadIdx.x == 0) status_out[0] = (int)(atomicOr(&ws.pool->status, 0u) | (atomicOr(&ws.state[0].fell_back, 0u) & 0xFu));
}

// ---- distributed pooled fit: staged entry points (host does the all-reduces in between) -----------
template <typename T>
static int dfit_moments_typed(const void* images, const Geometry& g0, const Workspace& ws, double* moments, hipStream_t stream) {
    Geometry g = g0;
    constexpr int W = PackOf<T>::n;
    const bool vec = (g.pixels % W == 0) && aligned_for(images, 16);
    g.vec = vec ? 1 : 0;
    g.vec_width = W;
    set_sampling(g);
    const unsigned grid = (unsigned)(g.n_tiles * g.blocks_per_tile);
    const T* in = static_cast<const T*>(images);
    if (vec)
        hipLaunchKernelGGL((stats_kernel<T, W>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, ws);
    else
        hipLaunchKernelGGL((stats_kernel<T, 1>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, ws);
    hipLaunchKernelGGL(dfit_reduce_partials_kernel, dim3(1), dim3(256), 0, stream, ws.partial, (int64_t)grid, (double)(g.n_tiles * g.pixels), moments);
    return check_launch("macenko dfit moments");
}

template <typename T>
static int dfit_histogram_typed(const void* images, const Geometry& g0, const DFitState* st, int stage, unsigned long long* hist, hipStream_t stream) {
    Geometry g = g0;
    constexpr int W = PackOf<T>::n;
    const bool vec = (g.pixels % W == 0) && aligned_for(images, 16);
    g.vec = vec ? 1 : 0;
    const unsigned grid = (unsigned)(g.n_tiles * g.blocks_per_tile);
    const T* in = static_cast<const T*>(images);
    if (hipMemsetAsync(hist, 0, 512 * sizeof(unsigned long long), stream) != hipSuccess) return fail(SX_ERR_LAUNCH, "hipMemsetAsync failed");
    if (vec)
        hipLaunchKernelGGL((dfit_histogram_kernel<T, W>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, st, stage, hist);
    else
        hipLaunchKernelGGL((dfit_histogram_kernel<T, 1>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, st, stage, hist);
    return check_launch("macenko dfit histogram");
}

template <typename T>
static int pfit_pass_typed(const void* images, const Geometry& g0, const Workspace& ws, int stage, long long* sums_out, hipStream_t stream) {
    Geometry g = g0;
    constexpr int W = PackOf<T>::n;
    const bool vec = (g.pixels % W == 0) && aligned_for(images, 16);
    g.vec = vec ? 1 : 0;
    g.vec_width = W;
    set_chunk(g, false);      // (the staged fit's other steps take the work items per tile from make_geometry())
    const unsigned grid = (unsigned)(g.n_tiles * g.blocks_per_tile);
    const T* in = static_cast<const T*>(images);
    if (stage < 0) {      // stats
        if (vec) hipLaunchKernelGGL((stats_kernel<T, W>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, ws);
        else hipLaunchKernelGGL((stats_kernel<T, 1>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, ws);
        return check_launch("macenko pfit stats");
    }
    if (stage == 0) {
        if (vec) hipLaunchKernelGGL((bracket_kernel<T, W, false>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, ws);
        else hipLaunchKernelGGL((bracket_kernel<T, 1, false>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, ws);
    } else {
        if (vec) hipLaunchKernelGGL((bracket_kernel<T, W, true>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, ws);
        else hipLaunchKernelGGL((bracket_kernel<T, 1, true>), dim3(grid), dim3(kStreamThreads), 0, stream, in, g, ws);
    }
    hipLaunchKernelGGL(pfit_reduce_export_kernel, dim3(512 / kReduceBins), dim3(kReduceBins * kReduceLanes), 0, stream, g, ws, stage, sums_out);
    return check_launch("macenko pfit pass");
}

}  // namespace macenko
}  // namespace sx

using namespace sx;
using namespace sx::macenko;

static int validate_images(const void* images, int64_t n, int64_t h, int64_t w, const void* ws, size_t ws_bytes, size_t need) {
    if (!images) return fail(SX_ERR_BAD_ARG, "images pointer is null");
    if (n <= 0 || h <= 0 || w <= 0) return fail(SX_ERR_BAD_ARG, "images must be (N,3,H,W) with positive sizes, got N=%lld H=%lld W=%lld", (long long)n, (long long)h, (long long)w);
    if (n * h * w >= (1ll << 32)) return fail(SX_ERR_BAD_ARG, "N*H*W must be below 2^32 pixels");
    if (!ws || ws_bytes < need) return fail(SX_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, ws_bytes);
    if (reinterpret_cast<uintptr_t>(ws) % 256 != 0) return fail(SX_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    return SX_OK;
}

// float32 tiles with their 8-bit codes (Coded<F>): the codes lie behind the workspace level of the form the call takes
static int form_level(int form, int64_t pixels) { return form == 0 ? kWsBase : (fused_size(pixels) ? kWsFused : kWsTwoPass); }
static size_t coded_workspace_bytes(int64_t n_tiles, int64_t pixels, int form = 0) {
    return macenko::workspace_bytes(n_tiles, pixels, form_level(form, pixels)) + macenko::coded_bytes(n_tiles, pixels);
}
static bool coded_call(int dtype, int64_t n, int64_t pixels, unsigned flags) {
    return dtype == SX_F32 && macenko::coded_size(n, pixels) && !(flags & (SX_MACENKO_CHANNELS_LAST | SX_MACENKO_SAMPLED));
}
// The number of a coded call: what its first pass writes into a tile's flag word when the tile is not 8-bit levels.  Never zero, never
// the number of another call in this process -- a flag word left by an earlier call (or never written) means nothing to this one.
static uint32_t next_code_epoch() {
    static std::atomic<uint32_t> counter{0x5F3759DFu};
    uint32_t e = counter.fetch_add(1u, std::memory_order_relaxed) + 1u;
    if (e == 0u) e = counter.fetch_add(1u, std::memory_order_relaxed) + 1u;
    return e;
}

extern "C" size_t sx_macenko_workspace_bytes(int64_t n_tiles, int64_t height, int64_t width) {
    if (n_tiles <= 0 || height <= 0 || width <= 0) return 0;
    size_t need = macenko::workspace_bytes(n_tiles, height * width);
    if (macenko::coded_size(n_tiles, height * width)) need += macenko::coded_bytes(n_tiles, height * width);
#ifdef SX_DIAG
    need = std::max(need, macenko::resident_bytes(n_tiles, height * width));
#endif
    return need;
}

// Which form sx_macenko_transform takes for a call: 0 the four passes, 1 the two-pass form as four launches, 2 the two-pass form
// with pass A, the stage jobs and the reconstruct pass fused into one launch.  Where the two-pass form pays (measured,
// tools/bench_twopass.py, tools/survey_forms.py): 4- and 8-byte pixels in batches of at least ~4 M pixels and tiles of ~192 x 192 ... 724 x 724
// (1024 tiles of 128 x 128: 367 us in four passes against 411).  Narrow pixels
// (uint8 / bf16 / f16) make the four passes cheap -- the forms are level there --, big tiles put tens of thousands of candidates on
// one stage workgroup.  The fused launch serves planar float32 tiles of 128 x 128 ... 512 x 512 pixels.  SX_MACENKO_TWO_PASS asks
// for the two-pass form wherever it is able to run (tests, A/B runs); SX_MACENKO_NO_FUSE keeps it to four launches.
extern "C" int sx_macenko_form(int dtype, int64_t n, int64_t h, int64_t w, unsigned flags) {
    if (n <= 0 || h <= 0 || w <= 0 || (flags & SX_MACENKO_SAMPLED)) return 0;
    const int64_t pixels = h * w;
#ifdef SX_DIAG
    // 3: the tile-resident form (macenko_resident.hpp), diagnostic builds, where asked for (measured slower than the multi-launch forms: DESIGN.md 4e)
    if ((flags & SX_MACENKO_RESIDENT) != 0 && !(flags & (SX_MACENKO_CLASSIC | SX_MACENKO_TWO_PASS | SX_MACENKO_FUSE)) && resident_able(dtype, n, pixels, flags)) return 3;
    const unsigned forced_two_pass = flags & SX_MACENKO_TWO_PASS, fuse = flags & SX_MACENKO_FUSE;
#else
    const unsigned forced_two_pass = 0u, fuse = 0u;
#endif
    // (narrow pixels -- uint8 / f16 / bf16 -- since the candidates travel as dense records: tiles of ~360 x 360 ... 512 x 512, where the
    // two-pass form saves two instruction-bound passes -- uint8 64 x 512 x 512: 106 us against 118, bf16 126 against 144; at 320 x 320 and
    // below, and on the per-wave segments of larger tiles, the four passes win: tools/bench_twopass.py, profiles/r03_forms_by_dtype_and_tile.jsonl)
    if (dtype == SX_F64) return 0;      // (a rare element type: the four passes serve it)
    const bool wide = dtype == SX_F32;
    const bool pays = n * pixels >= (1ll << 22) && (wide ? (pixels >= 36864 && pixels <= (1ll << 19)) : (pixels >= 131072 && pixels <= 262144));
    const bool wanted = forced_two_pass != 0 || (pays && !(flags & SX_MACENKO_CLASSIC));
    if (!(wanted && two_pass_size(pixels))) return 0;
#ifndef SX_DIAG
    if (!fused_size(pixels)) return 0;      // (the product build carries the two-pass form with dense candidate records only: tiles up to 512 x 512)
#endif
    const bool fusable = fuse != 0 && dtype == SX_F32 && fused_size(pixels) && !(flags & SX_MACENKO_CHANNELS_LAST);
    return fusable ? 2 : 1;
}
extern "C" int sx_macenko_takes_two_pass(int dtype, int64_t n, int64_t h, int64_t w, unsigned flags) { const int f = sx_macenko_form(dtype, n, h, w, flags); return (f == 1 || f == 2) ? 1 : 0; }

// The part of the workspace ONE call of sx_macenko_transform with these arguments needs (a prefix of sx_macenko_workspace_bytes(),
// which serves any call): without the two-pass areas where the call takes the four-pass form (narrow pixels, small batches), and
// without the four-launch form's candidate arrays where it takes the fused launch.
extern "C" size_t sx_macenko_workspace_bytes_for(int dtype, int64_t n_tiles, int64_t height, int64_t width, unsigned flags) {
    if (n_tiles <= 0 || height <= 0 || width <= 0) return 0;
    const int form = sx_macenko_form(dtype, n_tiles, height, width, flags);
    // (the resident form's buffer also serves the four passes: a call whose pointers turn out not to be 16-byte aligned takes those)
#ifdef SX_DIAG
    if (form == 3) return std::max(macenko::resident_bytes(n_tiles, height * width), macenko::workspace_bytes(n_tiles, height * width, kWsBase));
#endif
    // (the four-launch two-pass form keeps its candidates in the dense record arrays too for tiles up to 512 x 512: only larger
    // tiles need the per-wave segments)
    const bool dense = form != 0 && fused_size(height * width);
    size_t need = macenko::workspace_bytes(n_tiles, height * width, form == 0 ? kWsBase : (dense ? kWsFused : kWsTwoPass));
    // (float32 batches: room for the tiles' 8-bit codes, 3 bytes per pixel -- also where the call itself takes the two-pass form: a
    // caller that sends it to the four passes with SX_MACENKO_CLASSIC on real tissue uses the same buffer)
    if (coded_call(dtype, n_tiles, height * width, flags)) need = std::max(need, coded_workspace_bytes(n_tiles, height * width, form == 1 ? 1 : 0));
    return need;
}

extern "C" int sx_macenko_transform(const void* images, void* out, int dtype, int64_t n, int64_t h, int64_t w, const float* sm, const float* tmc, unsigned flags, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    int rc = validate_images(images, n, h, w, ws_ptr, ws_bytes, sx_macenko_workspace_bytes_for(dtype, n, h, w, flags));
    if (rc != SX_OK) return rc;
    if (!out || !sm || !tmc) return fail(SX_ERR_BAD_ARG, "out / stain_matrix / target_max_conc pointer is null");
    if ((flags & (SX_MACENKO_OUT_BF16 | SX_MACENKO_OUT_F16)) != 0 && (dtype != SX_U8 || (flags & SX_MACENKO_OUT_BF16 && flags & SX_MACENKO_OUT_F16)))
        return fail(SX_ERR_BAD_ARG, "SX_MACENKO_OUT_BF16 / SX_MACENKO_OUT_F16: uint8 input only, one of the two");
    Geometry g = make_geometry(n, h * w, 0);
    g.interleaved = (flags & SX_MACENKO_CHANNELS_LAST) ? 1 : 0;
    g.fast = (flags & SX_MACENKO_SAMPLED) ? 1 : 0;
#ifdef SX_DIAG
    g.no_tie = (flags & SX_MACENKO_NO_TIE_SHORTCUT) ? 1 : 0;
    g.spec_fail = (flags & SX_MACENKO_SPEC_FAIL) ? 1 : 0;
#else
    if (flags & ~(SX_MACENKO_NORMALIZE_0_1 | SX_MACENKO_CHANNELS_LAST | SX_MACENKO_SAMPLED | SX_MACENKO_CLASSIC | SX_MACENKO_OUT_BF16 | SX_MACENKO_OUT_F16))
        return fail(SX_ERR_BAD_ARG, "flags 0x%x: bits outside the public set (the diagnostic flags need the diagnostic build, -DSX_DIAG)", flags);
#endif
    g.out_code = (flags & SX_MACENKO_OUT_BF16) ? SX_BF16 : ((flags & SX_MACENKO_OUT_F16) ? SX_F16 : 0);
    int form = sx_macenko_form(dtype, n, h, w, flags);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    const bool unit = (flags & SX_MACENKO_NORMALIZE_0_1) != 0;
#ifdef SX_DIAG
    if (form == 3) {
        if (aligned_for(images, 16) && aligned_for(out, 16)) return resident_transform(images, out, dtype, n, h * w, ws_ptr, sm, tmc, unit, stream);
        form = 0;
    }
#endif
    g.two_pass = form != 0 ? 1 : 0;
    g.fused = form == 2 ? 1 : 0;
    // the four passes over a batch of float32 tiles: 8-bit codes behind the first pass (a workspace sized by an older rule just runs without)
    // (the two-pass form: its one pass over the input leaves them, its reconstruct pass reads them)
    size_t codes_at = 0;
    if ((form == 0 || form == 1) && coded_call(dtype, n, g.pixels, flags) && ws_bytes >= coded_workspace_bytes(n, g.pixels, form)) {
        g.code_epoch = next_code_epoch();
        codes_at = macenko::workspace_bytes(n, g.pixels, form_level(form, g.pixels));
    }
#ifdef SX_DIAG
    if (flags & SX_MACENKO_NO_CODES) g.code_epoch = 0u;
#endif
    const Workspace ws = carve(ws_ptr, n, g.pixels, codes_at);
    switch (dtype) {
        case SX_U8: return transform_typed<uint8_t>(images, out, g, ws, sm, tmc, unit, stream);
        case SX_F16: return transform_typed<__half>(images, out, g, ws, sm, tmc, unit, stream);
        case SX_BF16: return transform_typed<__hip_bfloat16>(images, out, g, ws, sm, tmc, unit, stream);
        case SX_F32: return transform_typed<float>(images, out, g, ws, sm, tmc, unit, stream);
        case SX_F64: return transform_typed<double>(images, out, g, ws, sm, tmc, unit, stream);
        default: return fail(SX_ERR_DTYPE, "unsupported dtype code %d", dtype);
    }
}

extern "C" int sx_macenko_fit(const void* images, int dtype, int64_t n, int64_t h, int64_t w, float* he_out, float* max_c_out, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    int rc = validate_images(images, n, h, w, ws_ptr, ws_bytes, sx_macenko_workspace_bytes_for(dtype, n, h, w, SX_MACENKO_CLASSIC));
    if (rc != SX_OK) return rc;
    if (!he_out || !max_c_out) return fail(SX_ERR_BAD_ARG, "he_out / max_c_out pointer is null");
    Geometry g = make_geometry(n, h * w, 1);
    // (a float32 batch: the moments pass leaves the tiles' 8-bit codes, the two bracket passes read them -- as in the transform's four passes)
    size_t codes_at = 0;
    if (coded_call(dtype, n, g.pixels, 0u) && ws_bytes >= coded_workspace_bytes(n, g.pixels, 0)) {
        g.code_epoch = next_code_epoch();
        codes_at = macenko::workspace_bytes(n, g.pixels, kWsBase);
    }
    const Workspace ws = carve(ws_ptr, n, g.pixels, codes_at);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    switch (dtype) {
        case SX_U8: return fit_typed<uint8_t>(images, g, ws, he_out, max_c_out, stream);
        case SX_F16: return fit_typed<__half>(images, g, ws, he_out, max_c_out, stream);
        case SX_BF16: return fit_typed<__hip_bfloat16>(images, g, ws, he_out, max_c_out, stream);
        case SX_F32: return fit_typed<float>(images, g, ws, he_out, max_c_out, stream);
        case SX_F64: return fit_typed<double>(images, g, ws, he_out, max_c_out, stream);
        default: return fail(SX_ERR_DTYPE, "unsupported dtype code %d", dtype);
    }
}

extern "C" int sx_macenko_tile_params(const void* ws_ptr, int64_t n_groups, float* params_out, void* stream_ptr) {
    if (!ws_ptr || !params_out || n_groups <= 0) return fail(SX_ERR_BAD_ARG, "bad argument to sx_macenko_tile_params");
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    const unsigned grid = (unsigned)((n_groups + 63) / 64);
    hipLaunchKernelGGL(export_params_kernel, dim3(grid), dim3(64), 0, stream, static_cast<const GroupState*>(ws_ptr), n_groups, params_out);
    return check_launch("macenko export_params");
}

#ifdef SX_STAMPS
// (debug builds only, not part of the ABI) where the fused launch's unit stamps lie: eight uint64 per unit (blockIdx) in the block-histogram area
extern "C" size_t sx_debug_unit_stamp_offset(int64_t n, int64_t h, int64_t w) {
    const Workspace ws = carve(nullptr, n, h * w);
    return (size_t)reinterpret_cast<uintptr_t>(ws.block_hist);
}
#endif

extern "C" size_t sx_macenko_telemetry_offset(void) { return offsetof(GroupState, slow_slots); }

extern "C" size_t sx_macenko_dfit_state_bytes(void) { return sizeof(DFitState); }

extern "C" int sx_macenko_dfit_moments(const void* images, int dtype, int64_t n, int64_t h, int64_t w, double* moments_out, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    int rc = validate_images(images, n, h, w, ws_ptr, ws_bytes, macenko::workspace_bytes(n, h * w, kWsBase));
    if (rc != SX_OK) return rc;
    if (!moments_out) return fail(SX_ERR_BAD_ARG, "moments_out pointer is null");
    Geometry g = make_geometry(n, h * w, 1);
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    switch (dtype) {
        case SX_U8: return dfit_moments_typed<uint8_t>(images, g, ws, moments_out, stream);
        case SX_F16: return dfit_moments_typed<__half>(images, g, ws, moments_out, stream);
        case SX_BF16: return dfit_moments_typed<__hip_bfloat16>(images, g, ws, moments_out, stream);
        case SX_F32: return dfit_moments_typed<float>(images, g, ws, moments_out, stream);
        case SX_F64: return dfit_moments_typed<double>(images, g, ws, moments_out, stream);
        default: return fail(SX_ERR_DTYPE, "unsupported dtype code %d", dtype);
    }
}

extern "C" int sx_macenko_dfit_begin(const double* moments, void* state, void* stream_ptr) {
    if (!moments || !state) return fail(SX_ERR_BAD_ARG, "moments / state pointer is null");
    hipLaunchKernelGGL(dfit_begin_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_ptr), moments, static_cast<DFitState*>(state));
    return check_launch("macenko dfit begin");
}

extern "C" int sx_macenko_dfit_histogram(const void* images, int dtype, int64_t n, int64_t h, int64_t w, const void* state, int stage, unsigned long long* hist_out, void* stream_ptr) {
    if (!images || !state || !hist_out) return fail(SX_ERR_BAD_ARG, "images / state / hist_out pointer is null");
    if (n <= 0 || h <= 0 || w <= 0 || (stage != 0 && stage != 1)) return fail(SX_ERR_BAD_ARG, "bad sizes or stage");
    Geometry g = make_geometry(n, h * w, 1);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    const DFitState* st = static_cast<const DFitState*>(state);
    switch (dtype) {
        case SX_U8: return dfit_histogram_typed<uint8_t>(images, g, st, stage, hist_out, stream);
        case SX_F16: return dfit_histogram_typed<__half>(images, g, st, stage, hist_out, stream);
        case SX_BF16: return dfit_histogram_typed<__hip_bfloat16>(images, g, st, stage, hist_out, stream);
        case SX_F32: return dfit_histogram_typed<float>(images, g, st, stage, hist_out, stream);
        case SX_F64: return dfit_histogram_typed<double>(images, g, st, stage, hist_out, stream);
        default: return fail(SX_ERR_DTYPE, "unsupported dtype code %d", dtype);
    }
}

extern "C" int sx_macenko_dfit_advance(void* state, int stage, const unsigned long long* hist, void* stream_ptr) {
    if (!state || !hist || (stage != 0 && stage != 1)) return fail(SX_ERR_BAD_ARG, "bad argument to sx_macenko_dfit_advance");
    hipLaunchKernelGGL(dfit_advance_kernel, dim3(1), dim3(128), 0, static_cast<hipStream_t>(stream_ptr), static_cast<DFitState*>(state), stage, hist);
    return check_launch("macenko dfit advance");
}

extern "C" int sx_macenko_dfit_result(const void* state, float* he_out, float* max_c_out, void* stream_ptr) {
    if (!state || !he_out || !max_c_out) return fail(SX_ERR_BAD_ARG, "bad argument to sx_macenko_dfit_result");
    hipLaunchKernelGGL(dfit_result_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream_ptr), static_cast<const DFitState*>(state), he_out, max_c_out);
    return check_launch("macenko dfit result");
}

// ------------------------------------------------------------------------------------------------
// distributed pooled fit on the bracket machinery: see include/stainx_hip.h
// ------------------------------------------------------------------------------------------------
static Geometry pfit_geometry(int64_t n, int64_t h, int64_t w, long long n_all, int sample_count) {
    Geometry g = make_geometry(n, h * w, 1);
    set_sampling(g);                 // local sample stride; cap per tile
    set_chunk(g, false);             // (the same work items per tile in every step of the staged fit: pfit_pass_typed() rounds the chunk for its packs)
    g.spread = 1;                    // also for a single local tile: the group spans other ranks
    g.cap = cap_for(g.pixels);
    g.distributed = 1;
    g.n_all = n_all;
    if (sample_count >= 0) g.sample_count = sample_count;
    return g;
}

extern "C" int sx_macenko_pfit_sample_count(int64_t n, int64_t h, int64_t w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    Geometry g = make_geometry(n, h * w, 1);
    set_sampling(g);
    return g.sample_count;
}

extern "C" int sx_macenko_pfit_stats(const void* images, int dtype, int64_t n, int64_t h, int64_t w, double* moments_out, float* sample_out, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    int rc = validate_images(images, n, h, w, ws_ptr, ws_bytes, macenko::workspace_bytes(n, h * w, kWsBase));
    if (rc != SX_OK) return rc;
    if (!moments_out || !sample_out) return fail(SX_ERR_BAD_ARG, "moments_out / sample_out pointer is null");
    const Geometry g = pfit_geometry(n, h, w, 0, -1);
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    switch (dtype) {
        case SX_U8: rc = pfit_pass_typed<uint8_t>(images, g, ws, -1, nullptr, stream); break;
        case SX_F16: rc = pfit_pass_typed<__half>(images, g, ws, -1, nullptr, stream); break;
        case SX_BF16: rc = pfit_pass_typed<__hip_bfloat16>(images, g, ws, -1, nullptr, stream); break;
        case SX_F32: rc = pfit_pass_typed<float>(images, g, ws, -1, nullptr, stream); break;
        case SX_F64: rc = pfit_pass_typed<double>(images, g, ws, -1, nullptr, stream); break;
        default: return fail(SX_ERR_DTYPE, "unsupported dtype code %d", dtype);
    }
    if (rc != SX_OK) return rc;
    hipLaunchKernelGGL(pfit_export_stats_kernel, dim3(13), dim3(256), 0, stream, ws, (int64_t)(g.n_tiles * g.blocks_per_tile), moments_out, sample_out);
    return check_launch("macenko pfit stats export");
}

extern "C" int sx_macenko_pfit_plane(const double* moments, long long n_all, const float* sample_union, int sample_count, int64_t n, int64_t h, int64_t w, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    if (!moments || !sample_union || !ws_ptr) return fail(SX_ERR_BAD_ARG, "moments / sample_union / workspace pointer is null");
    if (n <= 0 || h <= 0 || w <= 0 || n_all <= 0 || sample_count < 0 || sample_count > kSample) return fail(SX_ERR_BAD_ARG, "bad sizes");
    if (ws_bytes < macenko::workspace_bytes(n, h * w, kWsBase)) return fail(SX_ERR_WORKSPACE, "workspace too small");
    const Geometry g = pfit_geometry(n, h, w, n_all, sample_count);
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    hipLaunchKernelGGL(pfit_import_sample_kernel, dim3(12), dim3(256), 0, stream, ws, sample_union);
    hipLaunchKernelGGL(pfit_plane_kernel, dim3(1), dim3(kGroupThreads), 0, stream, g, ws, moments);
    return check_launch("macenko pfit plane");
}

extern "C" int sx_macenko_pfit_stats_packed(const void* images, int dtype, int64_t n, int64_t h, int64_t w, void* record_out, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    if (!record_out || reinterpret_cast<uintptr_t>(record_out) % 8 != 0) return fail(SX_ERR_BAD_ARG, "record_out is null or not 8-byte aligned");
    unsigned char* rec = static_cast<unsigned char*>(record_out);
    int rc = validate_images(images, n, h, w, ws_ptr, ws_bytes, macenko::workspace_bytes(n, h * w, kWsBase));
    if (rc != SX_OK) return rc;
    const Geometry g = pfit_geometry(n, h, w, 0, -1);
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    switch (dtype) {
        case SX_U8: rc = pfit_pass_typed<uint8_t>(images, g, ws, -1, nullptr, stream); break;
        case SX_F16: rc = pfit_pass_typed<__half>(images, g, ws, -1, nullptr, stream); break;
        case SX_BF16: rc = pfit_pass_typed<__hip_bfloat16>(images, g, ws, -1, nullptr, stream); break;
        case SX_F32: rc = pfit_pass_typed<float>(images, g, ws, -1, nullptr, stream); break;
        case SX_F64: rc = pfit_pass_typed<double>(images, g, ws, -1, nullptr, stream); break;
        default: return fail(SX_ERR_DTYPE, "unsupported dtype code %d", dtype);
    }
    if (rc != SX_OK) return rc;
    hipLaunchKernelGGL(pfit_export_stats_kernel, dim3(13), dim3(256), 0, stream, ws, (int64_t)(g.n_tiles * g.blocks_per_tile), reinterpret_cast<double*>(rec + 8),
                       reinterpret_cast<float*>(rec + 8 + 8 * kPartial), reinterpret_cast<long long*>(rec), (long long)n);
    return check_launch("macenko pfit stats export (packed)");
}

extern "C" int sx_macenko_pfit_plane_packed(const void* gathered, int world, const int* sample_counts_host, const long long* expected_tiles, int* stale_out, long long n_all, int sample_count, int64_t n, int64_t h,
                                            int64_t w, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    if (!gathered || !sample_counts_host || !ws_ptr || world < 1 || world > 64) return fail(SX_ERR_BAD_ARG, "null pointer or world size outside 1..64");
    if (n <= 0 || h <= 0 || w <= 0 || n_all <= 0 || sample_count < 0 || sample_count > kSample) return fail(SX_ERR_BAD_ARG, "bad sizes");
    if (ws_bytes < macenko::workspace_bytes(n, h * w, kWsBase)) return fail(SX_ERR_WORKSPACE, "workspace too small");
    PfitRanks ranks{};
    ranks.world = world;
    for (int r = 0; r < world; ++r) {
        if (sample_counts_host[r] < 0 || sample_counts_host[r] > kSample) return fail(SX_ERR_BAD_ARG, "sample count of rank %d outside 0..%d", r, kSample);
        ranks.sample_counts[r] = sample_counts_host[r];
    }
    const Geometry g = pfit_geometry(n, h, w, n_all, sample_count);
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    double* moments = reinterpret_cast<double*>(ws.pool->compact[kSlots - 1] + kCompact - 32);      // (scratch: the tail of the last compact list, free until the second stage's merge)
    hipLaunchKernelGGL(pfit_plane_packed_kernel, dim3(1), dim3(kGroupThreads), 0, stream, g, ws, static_cast<const unsigned char*>(gathered), ranks, expected_tiles, stale_out, moments);
    return check_launch("macenko pfit plane (packed)");
}

extern "C" int sx_macenko_pfit_gather_packed(const long long* sums_global, int stage, long long n_all, int sample_count, int64_t n, int64_t h, int64_t w, int share, const int* stale_flag, unsigned* row_out,
                                             void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    if (share < 1 || share > kCompact) return fail(SX_ERR_BAD_ARG, "share must be in [1, %d]", kCompact);
    if (!sums_global || !row_out || !ws_ptr || (stage != 0 && stage != 1)) return fail(SX_ERR_BAD_ARG, "null pointer or bad stage");
    if (n <= 0 || h <= 0 || w <= 0 || ws_bytes < macenko::workspace_bytes(n, h * w, kWsBase)) return fail(SX_ERR_WORKSPACE, "bad sizes or workspace too small");
    const Geometry g = pfit_geometry(n, h, w, n_all, sample_count);
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    // (one launch: the global sums are read where they lie and the record is written by the workgroups that find its keys --
    // pfit_reduce_export_kernel has emptied the stage's pooled lists)
    hipLaunchKernelGGL(pool_gather_kernel, dim3((unsigned)g.n_tiles), dim3(kGroupThreads), 0, stream, g, ws, stage, sums_global, stale_flag, row_out, share);
    return check_launch("macenko pfit gather (packed)");
}

extern "C" int sx_macenko_pfit_finish_packed(const unsigned* gathered_rows, int world, int share, int stage, long long n_all, int sample_count, int64_t n, int64_t h, int64_t w, float* he_out, float* max_c_out,
                                             int* status_out, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    if (!gathered_rows || !ws_ptr || (stage != 0 && stage != 1) || world < 1 || world > 64) return fail(SX_ERR_BAD_ARG, "null pointer, bad stage or world size");
    if (stage == 1 && (!he_out || !max_c_out || !status_out)) return fail(SX_ERR_BAD_ARG, "he_out / max_c_out / status_out pointer is null");
    if (n <= 0 || h <= 0 || w <= 0 || ws_bytes < macenko::workspace_bytes(n, h * w, kWsBase)) return fail(SX_ERR_WORKSPACE, "bad sizes or workspace too small");
    if (share < 1 || (long long)world * share > kCompact) return fail(SX_ERR_BAD_ARG, "world x share = %d x %d exceeds the compact list (%d)", world, share, kCompact);
    const Geometry g = pfit_geometry(n, h, w, n_all, sample_count);
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    if (stage == 0)
        hipLaunchKernelGGL(pfit_stain_rows_kernel, dim3(1), dim3(kGroupThreads), 0, stream, g, ws, world, share, gathered_rows);
    else
        hipLaunchKernelGGL(pfit_scale_rows_kernel, dim3(1), dim3(kGroupThreads), 0, stream, g, ws, world, share, gathered_rows, he_out, max_c_out, status_out);
    return check_launch("macenko pfit finish (packed)");
}

extern "C" int sx_macenko_pfit_pass(const void* images, int dtype, int64_t n, int64_t h, int64_t w, int stage, long long n_all, int sample_count, long long* sums_out, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    int rc = validate_images(images, n, h, w, ws_ptr, ws_bytes, macenko::workspace_bytes(n, h * w, kWsBase));
    if (rc != SX_OK) return rc;
    if (!sums_out || (stage != 0 && stage != 1)) return fail(SX_ERR_BAD_ARG, "sums_out is null or stage is not 0/1");
    const Geometry g = pfit_geometry(n, h, w, n_all, sample_count);
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    switch (dtype) {
        case SX_U8: return pfit_pass_typed<uint8_t>(images, g, ws, stage, sums_out, stream);
        case SX_F16: return pfit_pass_typed<__half>(images, g, ws, stage, sums_out, stream);
        case SX_BF16: return pfit_pass_typed<__hip_bfloat16>(images, g, ws, stage, sums_out, stream);
        case SX_F32: return pfit_pass_typed<float>(images, g, ws, stage, sums_out, stream);
        case SX_F64: return pfit_pass_typed<double>(images, g, ws, stage, sums_out, stream);
        default: return fail(SX_ERR_DTYPE, "unsupported dtype code %d", dtype);
    }
}

extern "C" int sx_macenko_pfit_gather(const long long* sums_global, int stage, long long n_all, int sample_count, int64_t n, int64_t h, int64_t w, int share, unsigned* compact_out, int* counts_out, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    if (share < 1 || share > kCompact) return fail(SX_ERR_BAD_ARG, "share must be in [1, %d]", kCompact);
    if (!sums_global || !compact_out || !counts_out || !ws_ptr || (stage != 0 && stage != 1)) return fail(SX_ERR_BAD_ARG, "null pointer or bad stage");
    if (n <= 0 || h <= 0 || w <= 0 || ws_bytes < macenko::workspace_bytes(n, h * w, kWsBase)) return fail(SX_ERR_WORKSPACE, "bad sizes or workspace too small");
    const Geometry g = pfit_geometry(n, h, w, n_all, sample_count);
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    hipLaunchKernelGGL(pfit_import_sums_kernel, dim3(1), dim3(256), 0, stream, ws, sums_global, stage);
    hipLaunchKernelGGL(pool_gather_kernel, dim3((unsigned)g.n_tiles), dim3(kGroupThreads), 0, stream, g, ws, stage);
    hipLaunchKernelGGL(pfit_export_compact_kernel, dim3(8), dim3(256), 0, stream, ws, stage, share, compact_out, counts_out);
    return check_launch("macenko pfit gather");
}

extern "C" int sx_macenko_pfit_finish(const unsigned* gathered_compact, const int* gathered_counts, int world, int share, int stage, long long n_all, int sample_count, int64_t n, int64_t h, int64_t w,
                                      float* he_out, float* max_c_out, int* status_out, void* ws_ptr, size_t ws_bytes, void* stream_ptr) {
    if (!gathered_compact || !gathered_counts || !ws_ptr || (stage != 0 && stage != 1) || world < 1 || world > 64) return fail(SX_ERR_BAD_ARG, "null pointer, bad stage or world size");
    if (stage == 1 && (!he_out || !max_c_out || !status_out)) return fail(SX_ERR_BAD_ARG, "he_out / max_c_out / status_out pointer is null");
    if (n <= 0 || h <= 0 || w <= 0 || ws_bytes < macenko::workspace_bytes(n, h * w, kWsBase)) return fail(SX_ERR_WORKSPACE, "bad sizes or workspace too small");
    if (share < 1 || (long long)world * share > kCompact) return fail(SX_ERR_BAD_ARG, "world x share = %d x %d exceeds the compact list (%d)", world, share, kCompact);
    const Geometry g = pfit_geometry(n, h, w, n_all, sample_count);
    const Workspace ws = carve(ws_ptr, n, g.pixels);
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    hipLaunchKernelGGL(pfit_merge_kernel, dim3(1), dim3(1024), 0, stream, ws, stage, world, share, gathered_compact, gathered_counts);
    if (stage == 0) {
        hipLaunchKernelGGL((stain_kernel<float, true>), dim3(1), dim3(kGroupThreads), 0, stream, (const float*)nullptr, g, ws);
    } else {
        hipLaunchKernelGGL((scale_kernel<float, true>), dim3(1), dim3(kGroupThreads), 0, stream, (const float*)nullptr, g, ws, (const float*)nullptr, he_out, max_c_out);
        hipLaunchKernelGGL(pfit_status_kernel, dim3(1), dim3(64), 0, stream, ws, status_out);
    }
    return check_launch("macenko pfit finish");
}

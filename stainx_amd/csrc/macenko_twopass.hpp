// Macenko transform in TWO passes over the pixels ("speculate, then verify") -- included by macenko.hip.
//
// The classic path (macenko.hip) reads every tile four times because each per-tile quantity needs the one before it:
// moments -> plane -> angle percentiles -> stain vectors -> concentration percentiles -> scale.  Here a cheap PRIOR of the
// tile (a presample of 1024 sectors of 16 consecutive pixels: ~6 % of an fp32 512x512 tile) stands in for the quantities
// that are not known yet while ONE pass gathers the exact moments and the few pixels that cannot be ruled out; small
// per-tile stages then work out the exact values, and PROVE from them that nothing outside the gathered pixels could
// have been the wanted order statistic.  A tile whose proof fails takes the slow exact path (whole-tile radix select).
// Every number that reaches the output is computed by the same device functions, in the same order, as on the classic
// path: the two paths agree bit for bit (tests/test_twopass_gpu.py).
//
//   K0 prior_kernel   one workgroup per tile: presample (kept in LDS as fp16 for its second look) -> approximate plane frame
//                     F = [a0 a1 an], the two angle brackets as four boundary DIRECTIONS in the (a0,a1) plane (an open
//                     side: a stand-in just beyond the sample's extreme), and per concentration slot two end directions
//                     with a lower threshold each                                               (~6 % of the input)
//   K1 pass_a_kernel  the moments exactly as stats_kernel accumulates them + per pixel eight half-plane tests in the
//                     prior frame (their linear forms on the matrix core); pixels that pass none are only counted, the
//                     others (~7 %; 12 % counted per slot) are queued in LDS and written out per slot as optical-density
//                     triples                                                                  (one read of the input)
//   K2 estimate_stage two workgroups per tile: exact plane from the moments; workgroup j: exact keys of the candidates of
//                     angle percentile j, exact order statistic, proof; hand-off of the two keys between the partners;
//                     stain vectors, pseudo-inverse; the same for concentration j; scale
//   reconstruct_kernel (unchanged)                                                  (one read + one write of the input)
//
// Why the proofs hold.  Let V be the exact plane (fp32, as the classic path computes it) and F the prior frame; with
// M = F^-1 V the exact projection of a pixel x is t = Rt th + nu w, where th = (a0.x, a1.x), w = an.x, Rt is the 2x2
// in-plane part of M (close to a rotation) and nu its out-of-plane row (the tilt of the prior plane, ~1e-3).
//   * angles: a linear map with positive determinant keeps the cyclic order of directions, so "th is clockwise of the
//     boundary d by more than m" implies "t is clockwise of Rt d" as long as m >= |nu||w| + rounding; pass A uses
//     m = kw |w| + kx |x|_1 and the stage checks |nu| against kw and that the selected key lies strictly between the
//     mapped boundaries.  That every kept pixel's angle lies safely inside (0, pi) is checked once per tile from the frame
//     (kept pixels have positive optical densities), not per pixel.
//   * concentrations: row j of the pseudo-inverse is p = F^-T g; its in-plane part (g0,g1) is a non-negative combination
//     alpha u1 + beta u2 of the slot's two end directions if it lies in their cone (checked), so for a pixel that
//     failed both tests (u1.th < T1 - m, u2.th < T2 - m) the exact concentration is below alpha T1 + beta T2 =: Theta;
//     the stage checks that the selected element is >= Theta.  The rank ORDER of a concentration depends only on the
//     direction of the other stain vector, which is why two thresholds per slot are enough.
#pragma once

namespace sx {
namespace macenko {

constexpr int kPriorSweeps = 4;                       // 4-pixel quads a thread of the prior stage samples
constexpr int kPriorUnitsMax = kGroupThreads / 4 * kPriorSweeps;      // sectors of 16 pixels per tile: 1024
constexpr int kQueue2 = 512;                          // 16-byte records a wave queues in LDS before it must flush
constexpr float kSpecSigmasConc = 5.0f, kSpecThresholdScale = 1.0f;      // the concentration thresholds: their quantile's distance below 99 % in standard deviations, and a factor on them
// (Round 4, tools/diag_real_batches.py over 150 real crops, after the dense records' capacity went from 14336 to 32768: at 5, 15 tiles have a
// slow slot -- 11 of them a concentration answer below its bound --, at 8 only the 4 whose frame check fails in all four slots (a plane
// estimated from a sample is not the tile's plane there: not a matter of margins).  8 costs the synthetic headline 0.148 -> 0.153 ms (0.7 +
// 1.3 % more candidates in the two slots) and changes nothing for a 64-tile batch of real tissue, which has such a tile 8 times in 10 and
// runs the four passes either way: 5 stays.)
constexpr float kSpecSigmas = 5.0f;                   // half-width of a bracket in standard deviations of the sample quantile
// Independent samples per 16-pixel sector (four of its pixels enter the histograms: neighbours).  Sectors far apart -- at least
// 8 sector lengths between sampled ones: tiles from ~360x360 -- are taken for TWO (sweeps over 3840 tiles of 512x512 / 256x256 and
// the random stress lose no slot to it even at three, tools/sweep_cones.py; small tiles, where the sampled sectors are each other's
// neighbours, lose a few at two and keep ONE).  A wrong guess costs time only: the slot's proof fails, the slow exact path runs,
// and the host routes the next calls to the four-pass form.
// (the values travel in Geometry::spec_*: set by the host from the constants below -- diagnostic builds read overrides from the environment)
constexpr float kSpecEffFar = 2.0f, kSpecEffNear = 1.0f;
__device__ __forceinline__ float spec_eff(const Geometry& g) { return g.prior_step_q16 >= (8u << 16) ? g.spec_eff_far : g.spec_eff_near; }
// margin m = kw |w| + kx |od|_1 + 1e-6 (kx: the fp16 rounding of the pixel, 2^-10, with room).  kw buys tolerance to the TILT of the
// prior plane against the exact one (a proof holds up to a tilt of about kw / 1.5): 0.05 was sized on the synthetic tiles, whose
// covariance has lambda_mid / lambda_min ~ 260; the reference's real example tiles have 2 ... 15, the plane of a 6 % presample
// tilts by 0.01 ... 0.03 and a third of the tiles lost their angle slots to it (profiles/r03_real_tiles_speculation_knobs.jsonl:
// 31 slow slots in 64 tiles at 0.05, 14 at 0.1 with the rotation bound below; the synthetic batch pays 0.5 us for it).
constexpr float kSpecKw = 0.10f, kSpecKx = 1.3e-3f;
// largest in-plane rotation |r01| + |r10| between the prior frame and the exact plane a proof accepts.  The proofs map the tested
// boundaries through the exact 2x2 (phi_slot_check: D = Rt d), so nothing in them needs the rotation to be small; 0.04 was a
// sanity bound that real tiles (0.01 ... 0.03 from the presample alone) ran into.  r00, r11 > 0.8 still bound it to ~0.6 rad.
constexpr float kSpecRot = 0.30f;
constexpr uint32_t kSpecSlow = 1u, kSpecHazard = 2u;  // GroupState::spec bits
constexpr int kMaxSegments = 256;                      // waves of pass A per tile (two_pass_size() keeps tiles within 64 work items)

// What pass A tests per pixel is LINEAR in the optical density: nine forms r . od + c (rows of a 16 x 4 matrix).  They run
// on the matrix core -- one v_mfma_f32_32x32x16_f16 per 64 pixels, see pass_a_item -- with the pixel rounded to fp16 on
// the way in; the rows themselves are stored as floats that fp16 represents exactly, so pass A and the proofs of the
// stages talk about the same numbers, and the input rounding (< 2^-10 relative, round-toward-zero) is part of the margin kx.
//   row 0 / 1   cross(d, th) for the lower / upper boundary d of angle slot 0 (zero row: that side is open)
//   row 2 / 3   the same for angle slot 1
//   row 4 / 5   u . th - T at the two end directions of concentration slot 0;   row 6 / 7   slot 1
//   row 8       w = an . od (the out-of-plane coordinate, for the margin)
constexpr int kRowBelowA = 0, kRowAboveA = 1, kRowBelowB = 2, kRowAboveB = 3, kRowConc = 4, kRowW = 8;
struct alignas(128) PriorRecord {
    float a0[3], a1[3], an[3];     // prior frame: th0 = a0.x, th1 = a1.x, w = an.x
    float rows[16][4];             // (r0, r1, r2, c): the form's value is r . od + c
    float kw, kx;                  // margin m = kw |w| + kx |od|_1 + 1e-6
    float cmax[4];                 // largest sample value of u . th per concentration row (the stages size their histogram bins with it)
    int32_t mode;                  // 0: speculate, 1: slow (the stages select over the whole tile)
    int32_t min_first;
    uint32_t open;                 // bit i: angle boundary i is open (nothing is excluded on that side)
};

// ---- fused transform (macenko_fused.hpp): what its roles hand each other inside ONE launch -------------------------------
// Every word below is zeroed by prior_kernel (the launch before), written with agent-scope atomics or write-through (sc1)
// stores and read with agent-scope (sc1) loads: MI355X_MICROARCH.md, "Valid forms".
struct alignas(64) FusedTile {
    uint32_t a_done;             // pass-A work items of the tile that have published their moments and candidates
    uint32_t s_done;             // stage jobs of the tile that have published their part of the stage record (2 = complete)
    uint32_t ncand[kSlots];      // candidate records reserved per slot (beyond fused_cap: overflow, the slot takes the slow path)
    uint32_t pad[10];
};
constexpr int kXcds = 8;
struct alignas(64) FusedQueue {
    uint32_t ticket;             // next unit of this queue (units are handed out in dependency order)
    uint32_t pad[15];
};
struct FusedSched {
    FusedQueue queue[kXcds];     // one queue per XCD, each on a line of its own: a single ticket word hands out ~88 units per
};                               // microsecond, and the launch starts with a thousand workgroups asking at once
static_assert(sizeof(FusedTile) == kFusedTileBytes && sizeof(FusedSched) <= kFusedSchedBytes, "workspace layout");


typedef unsigned int sx_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t ld_agent(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int32_t ld_agent(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_agent(const float* p) { return __uint_as_float(__hip_atomic_load(reinterpret_cast<const uint32_t*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
__device__ __forceinline__ double ld_agent(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_agent(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(float* p, float v) { __hip_atomic_store(reinterpret_cast<uint32_t*>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every storing wave, after its last write-through store and before the barrier in front of the signalling lane (inline asm: the
// compiler drops a builtin wait it can prove redundant -- MI355X_MICROARCH.md, "Compiler hazard")
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

constexpr int kPriorBins = 2048;       // fixed-range histograms of the prior stage
constexpr float kOpenExponent = 30.0f;  // an open bracket side's stand-in direction lies w bracket widths beyond the sample extreme, w in [1/4, 1] such that
                                        // 0.01 n + w m >= this (the wanted quantile lies beyond it with probability ~e^-that: 1e-13)
constexpr float kOpenNear = 16.0f;      // ... used when the closed side has at least this many independent sample pixels on its near side (m)
struct alignas(16) PriorScratch {
    double red[kGroupThreads / kWave][kPartial];
    double mom[kMoments];
    uint32_t hist[4][kPriorBins];
    uint32_t od16[kPriorSweeps][6][kGroupThreads];      // the sample's optical densities in fp16 (a thread's quad of a sweep: 12 halves), for the second look
    uint32_t n_kept, n_all, hazard, r_max;
    float frame[9];
    float bdir[4][2];
    float bkey[4], blevel[4];          // the four boundaries as diamond keys (before an open side is given its stand-in) and their quantile levels
    float cthr[4];
    uint32_t cmax[4];
    uint32_t open;
};

// One wave: the bin of a kPriorBins-bin histogram that holds 0-based rank `rank`: kPriorBins / 64 consecutive bins per lane,
// then the owner lane's bins one per lane.
__device__ __forceinline__ uint32_t prior_pick_bin(const uint32_t* hist, uint32_t rank) {
    constexpr int kPer = kPriorBins / kWave;
    static_assert(kPer <= kWave, "second level: one bin per lane");
    const int lane = (int)lane_id();
    uint32_t mine = 0;
#pragma unroll 8
    for (int i = 0; i < kPer; ++i) mine += hist[kPer * lane + i];
    const uint32_t incl = wave_scan_u32(mine);
    const uint64_t over = __ballot(incl > rank);
    const int owner = over ? (__ffsll((long long)over) - 1) : (kWave - 1);
    const uint32_t r = rank - (uint32_t)__builtin_amdgcn_readlane((int)(incl - mine), owner);
    const uint32_t h = lane < kPer ? hist[kPer * owner + lane] : 0u;
    const uint32_t incl2 = wave_scan_u32(h);
    const uint64_t over2 = __ballot(lane < kPer && incl2 > r);
    const int sub = over2 ? (__ffsll((long long)over2) - 1) : (kPer - 1);
    return (uint32_t)(kPer * owner + sub);
}

__device__ __forceinline__ float wave_total_f32(float x) {      // fixed order; the total in lane 63 only
    auto step = [](float v, auto mover) { return v + __uint_as_float(mover(__float_as_uint(v))); };
    x = step(x, [](uint32_t b) { return dpp_move<0x111, 0xF>(0u, b); });
    x = step(x, [](uint32_t b) { return dpp_move<0x112, 0xF>(0u, b); });
    x = step(x, [](uint32_t b) { return dpp_move<0x114, 0xF>(0u, b); });
    x = step(x, [](uint32_t b) { return dpp_move<0x118, 0xF>(0u, b); });
    x = step(x, [](uint32_t b) { return dpp_move<0x142, 0xA>(0u, b); });
    x = step(x, [](uint32_t b) { return dpp_move<0x143, 0xC>(0u, b); });
    return x;
}

// The presample: unit u is one sector (16 consecutive pixels) of cell u at a hashed offset; four lanes per sector, so a
// thread's share of sweep s is one quad of 4 pixels.
// (fetch and conversion apart: the first look requests the quads of ALL its sweeps before it converts the first -- one round trip to
// memory instead of four one after the other, 6.4 -> ~3 us of the prior launch)
template <typename T, bool kVec, bool kInter>
__device__ __forceinline__ bool prior_fetch(const T* __restrict__ img, const Geometry& g, int s, float (&u)[3][4]) {
    const int tid = threadIdx.x, unit = s * (kGroupThreads / 4) + (tid >> 2);
    const bool have = unit < g.prior_units;
    // cell u = sectors [u step, (u + 1) step) in 16.16 fixed point (no integer divisions here: they cost the two looks ~2 us each)
    const uint64_t step = (uint64_t)g.prior_step_q16;
    const uint32_t start = (uint32_t)(((uint64_t)unit * step) >> 16), width = (uint32_t)(((uint64_t)(unit + 1) * step) >> 16) - start;
    const uint32_t sector = start + (((((uint32_t)unit * 0x9E3779B1u) >> 16) * width) >> 16);
    // (a thread without a unit reads the tile's first quad and drops it -- `have` gates every use: a load inside `if (have)` is a
    // branch the next sweep's loads cannot be moved across)
    const int64_t p = have ? ((int64_t)sector * 4 + (tid & 3)) * 4 : (int64_t)(tid & 3) * 4;
    if constexpr (kVec) {
        load_pixels<T, 4, kInter>(img, g.pixels, p, u);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float v[3][1];
            load_pixels<T, 1, kInter>(img, g.pixels, p + i, v);
#pragma unroll
            for (int c = 0; c < 3; ++c) u[c][i] = v[c][0];
        }
    }
    return have;
}
template <typename T>
__device__ __forceinline__ void prior_densities(const float (&u)[3][4], float (&od)[4][3]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int c = 0; c < 3; ++c) od[i][c] = optical_density<T>(u[c][i]);
}

template <typename T, bool kVec, bool kInter>
__global__ __launch_bounds__(kGroupThreads) void prior_kernel(const T* __restrict__ images, Geometry g, Workspace ws) {
    __shared__ PriorScratch sh;
    const int tile = blockIdx.x, tid = threadIdx.x, wave = tid / kWave;
    const uint32_t lane = lane_id();
    GroupState& st = ws.state[tile];
    PriorRecord* pr = &ws.prior[tile];
    const T* img = images + (int64_t)tile * 3 * g.pixels;
    SX_STAMP(st, 0);
    if (tid < kSlots) {
        put(&st.ncand[tid], 0u);
        put(&st.below[tid], 0u);
        put(&st.ncand_seen[tid], 0u);
        put(&st.over_count[tid], 0u);
    }
    if (tid < 2) put(&st.phi_pub[tid], 0ull);
    if ((g.fused || g.dense) && tid < (int)(sizeof(FusedTile) / 4)) put(reinterpret_cast<uint32_t*>(&ws.ftile[tile]) + tid, 0u);      // (a kernel boundary lies between this and the fused launch)
    if (g.fused && tile == 0 && tid >= 32 && tid < 32 + kXcds) put(&ws.fsched->queue[tid - 32].ticket, 0u);
    if (tid == 0) {
        put(&st.fell_back, 0u);
        put(&st.spec, 0u);
        sh.n_kept = sh.n_all = sh.hazard = sh.open = sh.r_max = 0;
        sh.cmax[0] = sh.cmax[1] = sh.cmax[2] = sh.cmax[3] = 0;
    }
    for (int i = tid; i < 4 * kPriorBins; i += kGroupThreads) (&sh.hist[0][0])[i] = 0;

    // ---- first look at the sample: moments of its kept pixels (fp32 per thread: 16 pixels; fp64 beyond).  The pixels are
    // not kept in registers across the eigen step (its fp64 code needs them all): they are read again afterwards, from L2.
    {
        float m[kPartial];
#pragma unroll
        for (int k = 0; k < kPartial; ++k) m[k] = 0.0f;
        float raw[kPriorSweeps][3][4];
        bool have_quad[kPriorSweeps];
#pragma unroll
        for (int s = 0; s < kPriorSweeps; ++s) have_quad[s] = prior_fetch<T, kVec, kInter>(img, g, s, raw[s]);
#pragma unroll
        for (int s = 0; s < kPriorSweeps; ++s) {
            float od[4][3];
            prior_densities<T>(raw[s], od);
            const bool have = have_quad[s];
            // (kept for the second look in fp16: brackets and thresholds of a SAMPLE need no more, and reading the pixels again
            // -- from L2, with their logarithms -- was 3 us of this kernel)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const int e = 2 * q;      // halves e, e + 1 of the quad's 12: element e = pixel e / 3, channel e % 3
                typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
                const fp16x2 h = __builtin_amdgcn_cvt_pkrtz(od[e / 3][e % 3], od[(e + 1) / 3][(e + 1) % 3]);
                sh.od16[s][q][tid] = __builtin_bit_cast(uint32_t, h);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* o = od[i];
                const float keep = (have && od_selected(o, false)) ? 1.0f : 0.0f;
                const float k0 = keep * o[0], k1 = keep * o[1], k2 = keep * o[2];
                m[0] += keep;
                m[1] += k0;
                m[2] += k1;
                m[3] += k2;
                m[4] = fmaf(k0, o[0], m[4]);
                m[5] = fmaf(k0, o[1], m[5]);
                m[6] = fmaf(k0, o[2], m[6]);
                m[7] = fmaf(k1, o[1], m[7]);
                m[8] = fmaf(k1, o[2], m[8]);
                m[9] = fmaf(k2, o[2], m[9]);
            }
        }
        SX_STAMP(st, 1);
        // (fp32 across the wave: a prior needs no more; the sixteen waves and the covariance are fp64)
#pragma unroll
        for (int k = 0; k < kPartial; ++k) {
            const float s = wave_total_f32(m[k]);
            if (lane == kWave - 1) sh.red[wave][k] = (double)s;
        }
    }
    __syncthreads();
    if (tid < kMoments) {
        double s = 0.0;
        if (tid < kPartial)
            for (int w = 0; w < kGroupThreads / kWave; ++w) s += sh.red[w][tid];
        sh.mom[tid] = s;
    }
    __syncthreads();
    const int m_kept = (int)sh.mom[0];
    SX_STAMP(st, 2);
    if (tid < 2 && m_kept >= 3) {
        double cov[9];
        float vecs[6];
        bool use_all;
        unsigned long long n_sel;
        plane_from_moments<true>(sh.mom, false, cov, vecs, use_all, n_sel);
        if (tid == 0) {
            const double x0 = vecs[0], y0 = vecs[2], z0 = vecs[4], x1 = vecs[1], y1 = vecs[3], z1 = vecs[5];
            double nx = y0 * z1 - z0 * y1, ny = z0 * x1 - x0 * z1, nz = x0 * y1 - y0 * x1;
            const double n2 = nx * nx + ny * ny + nz * nz;
            const double inv = n2 > 0.0 ? 1.0 / sqrt(n2) : 0.0;
            sh.frame[0] = vecs[0]; sh.frame[1] = vecs[2]; sh.frame[2] = vecs[4];
            sh.frame[3] = vecs[1]; sh.frame[4] = vecs[3]; sh.frame[5] = vecs[5];
            sh.frame[6] = (float)(nx * inv); sh.frame[7] = (float)(ny * inv); sh.frame[8] = (float)(nz * inv);
        }
    }
    __syncthreads();
    auto give_up = [&]() {      // uniform: the stages select over the whole tile
        if (tid == 0) {
            put(&pr->mode, 1);
            put(&st.spec, kSpecSlow);
        }
    };
    // Kept pixels have every od_c >= 0.15 > 0, so th1 >= min(a1) |od|_1, |th0| <= max|a0| |od|_1, |w| <= max|an| |od|_1: if the
    // largest-eigenvalue vector is positive enough, every kept pixel's angle lies safely inside (0, pi) and pass A needs no
    // wrap-around test per pixel (real H&E: a1 ~ (0.5, 0.7, 0.5)).
    const float a1_min = fminf(sh.frame[3], fminf(sh.frame[4], sh.frame[5]));
    const float a0_max = fmaxf(fabsf(sh.frame[0]), fmaxf(fabsf(sh.frame[1]), fabsf(sh.frame[2])));
    const float an_max = fmaxf(fabsf(sh.frame[6]), fmaxf(fabsf(sh.frame[7]), fabsf(sh.frame[8])));
    if (m_kept < 3 || g.pixels < 16 || !(a1_min > 0.06f * a0_max + 2.0f * g.spec_kw * an_max + 0.01f)) {
        give_up();
        return;
    }
    SX_STAMP(st, 3);
    float fr[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) fr[i] = sh.frame[i];
    // ---- second look: plane coordinates of the sample; the angle histogram takes ONE pixel per quad (pixel s of sweep s:
    // the four are neighbours with nearly the same key, and LDS atomics on neighbours' bins serialise)
    float t0[kPriorSweeps][4], t1[kPriorSweeps][4];
    uint32_t have_bits = 0, sub_kept = 0;      // bit s: the thread has a quad in sweep s / its histogram pixel passes the OD filter
    {
        uint32_t cnt = 0, all = 0, haz = 0;
        float r_max = 0.0f;
#pragma unroll
        for (int s = 0; s < kPriorSweeps; ++s) {
            float od[4][3];
            const bool have = s * (kGroupThreads / 4) + (tid >> 2) < g.prior_units;      // (as in prior_load)
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                typedef _Float16 half2v __attribute__((ext_vector_type(2)));
                const half2v h = __builtin_bit_cast(half2v, sh.od16[s][q][tid]);      // (written by this thread: no barrier needed)
                const int e = 2 * q;
                od[e / 3][e % 3] = (float)h[0];
                od[(e + 1) / 3][(e + 1) % 3] = (float)h[1];
            }
            if (have) have_bits |= 1u << s;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float* o = od[i];
                t0[s][i] = fmaf(o[2], fr[2], fmaf(o[1], fr[1], o[0] * fr[0]));
                t1[s][i] = fmaf(o[2], fr[5], fmaf(o[1], fr[4], o[0] * fr[3]));
                if (have) {
                    r_max = fmaxf(r_max, fabsf(t0[s][i]) + fabsf(t1[s][i]));
                    if (i == s) ++all;
                    if (od_selected(o, false)) {
                        if (!(t1[s][i] - 0.05f * fabsf(t0[s][i]) > 0.0f)) haz = 1;
                        if (i == s) {
                            ++cnt;
                            sub_kept |= 1u << s;
                        }
                    }
                }
            }
        }
        cnt = wave_total_u32(cnt);
        all = wave_total_u32(all);
        const uint32_t rm = wave_max_u32(__float_as_uint(r_max));      // non-negative floats order like their bits
        const uint64_t hz = __ballot(haz != 0);
        if (lane == 0) {
            if (cnt) atomicAdd(&sh.n_kept, cnt);
            atomicAdd(&sh.n_all, all);
            atomicMax(&sh.r_max, rm);
            if (hz) atomicOr(&sh.hazard, 1u);
        }
        // a kept pixel without a hazard has th1 > 0: its diamond key lies in (0, 2)
#pragma unroll
        for (int s = 0; s < kPriorSweeps; ++s)
            if ((sub_kept >> s) & 1u) {
                const float d = diamond_angle(t1[s][s], t0[s][s]);
                atomicAdd(&sh.hist[0][min((uint32_t)fmaxf(d * (float)(kPriorBins / 2), 0.0f), (uint32_t)(kPriorBins - 1))], 1u);
            }
    }
    __syncthreads();
    SX_STAMP(st, 4);
    if (sh.hazard) {
        give_up();
        return;
    }
    const int mv = (int)sh.n_kept, ms = (int)sh.n_all;
    if (wave < 4) {      // query `wave`: lower / upper boundary of slot 0, lower / upper boundary of slot 1
        const float f = wave < 2 ? 0.01f : 0.99f;
        const float n_eff = fmaxf((float)mv * (spec_eff(g) / 4.0f), 4.0f);      // four histogram pixels per sector
        const float sd = sqrtf(f * (1.0f - f) / n_eff);
        const bool upper = (wave & 1) != 0;
        const float level = upper ? f + g.spec_sigmas * sd : f - g.spec_sigmas * sd;
        const bool open = mv < 16 || (upper ? level >= 1.0f : level <= 0.0f);
        const float pos = fminf(fmaxf(level, 0.0f), 1.0f) * (float)max(mv - 1, 0);
        const uint32_t rank = (uint32_t)min(max((int)(upper ? ceilf(pos) : floorf(pos)), 0), max(mv - 1, 0));
        const uint32_t b = prior_pick_bin(sh.hist[0], rank);
        if (lane == 0) {
            // the bin's outer edge (an open side: the outer edge of the sample's extreme)
            sh.bkey[wave] = upper ? (float)(b + 1) * (2.0f / kPriorBins) : (float)b * (2.0f / kPriorBins);
            sh.blevel[wave] = fminf(fmaxf(level, 0.0f), 1.0f) * n_eff;      // (expected number of independent sample pixels on the low side of the boundary)
            if (open) atomicOr(&sh.open, 1u << wave);
        }
    }
    __syncthreads();
    if (tid < 4) {
        // An open side excludes no pixel from its angle slot, but the concentration tests still need a DIRECTION for that end of
        // the cone the other stain vector lies in.  The range's end (what this used to be) is ~40 degrees from the data, and a
        // test along it passes a different 2.5 % of the pixels than the test at the other end: twice the candidates (5.7 % per
        // slot on config 2 against 4.5 % with one bracket width added, 4.0 % with a quarter).  The stand-in is the sample's extreme
        // moved outwards by w times the bracket's own width.  The wanted quantile lies beyond it only if NO independent sample
        // pixel fell into a stretch that holds the quantile's own 1 % of the tile plus about w times what lies between the extreme
        // and the closed boundary: e^-(0.01 n + w m) for n independent pixels, m of them expected on the closed boundary's near
        // side; w is chosen for an exponent of 30 (kOpenExponent), between 1/4 and 1; with fewer than 16 pixels on the near side
        // the range's end stays.  If it happens the stage's cone check fails and the slot takes the slow exact path: a wrong guess
        // costs time, never a bit.
        const bool upper = (tid & 1) != 0, open = ((sh.open >> tid) & 1u) != 0, partner_open = ((sh.open >> (tid ^ 1)) & 1u) != 0;
        float d = sh.bkey[tid];
        if (open) {
            const float other = sh.bkey[tid ^ 1];
            const float n_eff = fmaxf((float)mv * (spec_eff(g) / 4.0f), 4.0f);
            const float m_near = upper ? n_eff - sh.blevel[tid ^ 1] : sh.blevel[tid ^ 1];      // sample pixels between the partner boundary and this end
            const bool tight = !partner_open && m_near >= kOpenNear;
            const float w = fminf(fmaxf((kOpenExponent - 0.01f * n_eff) / fmaxf(m_near, 1.0f), 0.25f), 1.0f);
            d = !tight ? (upper ? 1.98f : 0.02f) : (upper ? d + w * fmaxf(d - other, 0.0f) + 2.0f / kPriorBins : d - w * fmaxf(other - d, 0.0f) - 2.0f / kPriorBins);
        }
        d = fminf(fmaxf(d, 0.02f), 1.98f);
        float c, s;
        direction_from_key(float_key(d), c, s);
        sh.bdir[tid][0] = c;
        sh.bdir[tid][1] = s;
    }
    __syncthreads();
    SX_STAMP(st, 5);
    // ---- concentration tests: two end directions per slot, a lower threshold each
    float bd[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        bd[q][0] = sh.bdir[q][0];
        bd[q][1] = sh.bdir[q][1];
    }
    // which percentile vector is hematoxylin: the one with the larger first component (torch_backend.py:439), judged at the
    // middle of the brackets (the stage checks the exact vectors against the cone the choice implies)
    const float lo_c = bd[0][0] + bd[1][0], lo_s = bd[0][1] + bd[1][1], hi_c = bd[2][0] + bd[3][0], hi_s = bd[2][1] + bd[3][1];
    const bool min_first = (fr[0] * lo_c + fr[3] * lo_s) * __builtin_amdgcn_rsqf(lo_c * lo_c + lo_s * lo_s) > (fr[0] * hi_c + fr[3] * hi_s) * __builtin_amdgcn_rsqf(hi_c * hi_c + hi_s * hi_s);
    float cu[4][2];      // [2 * slot + end]
    {
        // perpendicular to the LOW vector, positive towards larger angles: (-s, c); to the HIGH vector, towards smaller: (s, -c)
        const int slot_perp_high = min_first ? 0 : 1, slot_perp_low = 1 - slot_perp_high;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            cu[2 * slot_perp_high + e][0] = bd[2 + e][1];
            cu[2 * slot_perp_high + e][1] = -bd[2 + e][0];
            cu[2 * slot_perp_low + e][0] = -bd[e][1];
            cu[2 * slot_perp_low + e][1] = bd[e][0];
        }
    }
    // keys u . th lie in [-r_max, r_max]; the wanted quantile is far up the positive side: bins over [0, r_max], the rest in bin 0
    const float r_max = __uint_as_float(sh.r_max);
    const float c_scale = r_max > 0.0f ? (float)kPriorBins / r_max : 0.0f;
    for (int i = tid; i < kPriorBins; i += kGroupThreads) sh.hist[0][i] = 0;      // (the angle histogram has been read)
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float k_max = 0.0f;
#pragma unroll
        for (int s = 0; s < kPriorSweeps; ++s)
            if ((have_bits >> s) & 1u) {
                const float k = fmaf(cu[q][0], t0[s][s], cu[q][1] * t1[s][s]);
                k_max = fmaxf(k_max, k);
                atomicAdd(&sh.hist[q][min((uint32_t)fmaxf(k * c_scale, 0.0f), (uint32_t)(kPriorBins - 1))], 1u);
            }
        const uint32_t km = wave_max_u32(__float_as_uint(k_max));
        if (lane == 0) atomicMax(&sh.cmax[q], km);
    }
    __syncthreads();
    SX_STAMP(st, 6);
    if (wave < 4) {
        const float n_eff = fmaxf((float)ms * (spec_eff(g) / 4.0f), 4.0f);
        const float level = 0.99f - g.spec_sigmas_conc * sqrtf(0.99f * 0.01f / n_eff);
        const uint32_t rank = (uint32_t)max((int)floorf(fmaxf(level, 0.0f) * (float)max(ms - 1, 0)), 0);
        const uint32_t b = prior_pick_bin(sh.hist[wave], rank);
        if (lane == 0) sh.cthr[wave] = (level > 0.0f && b > 0) ? g.spec_tscale * (float)b * (r_max / (float)kPriorBins) : -__builtin_huge_valf();      // the bin's lower edge
    }
    __syncthreads();
    if (tid < 9) {      // one test row per thread (compile-time indices: a run-time index would send bd / cu to scratch memory)
        const float* a0 = &sh.frame[0];
        const float* a1 = &sh.frame[3];
        const float* an = &sh.frame[6];
        const uint32_t open = sh.open;
        float row[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (tid == q && ((open >> q) & 1u) == 0) {      // cross(d, th) = dx th1 - dy th0; an open side keeps the zero row
#pragma unroll
                for (int c = 0; c < 3; ++c) row[c] = bd[q][0] * a1[c] - bd[q][1] * a0[c];
            }
            if (tid == kRowConc + q) {                       // u . th - T
#pragma unroll
                for (int c = 0; c < 3; ++c) row[c] = cu[q][0] * a0[c] + cu[q][1] * a1[c];
                row[3] = sh.cthr[q] > -1e30f ? -sh.cthr[q] : 60000.0f;      // no threshold: every pixel is a candidate
            }
        }
        if (tid == kRowW) {
#pragma unroll
            for (int c = 0; c < 3; ++c) row[c] = an[c];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) put(&pr->rows[tid][c], __half2float(__float2half_rn(row[c])));
    } else if (tid < 16) {
#pragma unroll
        for (int c = 0; c < 4; ++c) put(&pr->rows[tid][c], 0.0f);
    }
    if (tid == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            put(&pr->a0[i], sh.frame[i]);
            put(&pr->a1[i], sh.frame[3 + i]);
            put(&pr->an[i], sh.frame[6 + i]);
        }
        put(&pr->kw, g.spec_kw);
        put(&pr->kx, kSpecKx);
#pragma unroll
        for (int q = 0; q < 4; ++q) put(&pr->cmax[q], __uint_as_float(sh.cmax[q]));
        put(&pr->mode, 0);
        put(&pr->min_first, min_first ? 1 : 0);
        put(&pr->open, sh.open);
    }
    SX_STAMP(st, 7);
}

// ------------------------------------------------------------------------------------------------
// K1: the one pass over the input that the estimate needs
// ------------------------------------------------------------------------------------------------
template <int TPB> struct PassAScratch {
    StatsScratch<TPB> stats;
    uint32_t queue[TPB / kWave][4][kQueue2];      // (od0, od1, od2, flags) as four planes: a record is four LDS stores of registers where they lie, no packing moves
    uint32_t below[2];
};

// Moves the records a wave queued to its own SEGMENT of the tile's per-slot candidate arrays (record i goes to slot s iff
// bit s of its flag word is set).  Every wave of a tile owns entries [k seg_cap, (k+1) seg_cap) of each slot, k = 4 * work
// item + wave: no reservation, no atomic, nothing another wave waits for -- with a counter shared by the tile, a flush in
// the middle of the pixel loop cost four returning atomics one after the other, ~10 us per wave.  `have[s]` counts what the
// wave has produced so far.  What does not fit the segment (tissue concentrated in a few work items: a tile that is 90 %
// background has all its candidates in two of them) goes to the tile's OVERFLOW area behind the segments, reserved with one
// atomic per flush and slot -- the rare path.
__device__ __forceinline__ void write_records(const uint32_t (*__restrict__ queue)[kQueue2], uint32_t n, uint32_t (&have)[kSlots], float* __restrict__ cand_tile, const Geometry& g, uint32_t seg_base, uint32_t* __restrict__ over_count) {
    const uint32_t cap2 = g.cap2, seg_cap = g.seg_cap, over_base = (uint32_t)g.n_seg * g.seg_cap;
    for (uint32_t i0 = 0; i0 < n; i0 += kWave) {
        const uint32_t i = i0 + lane_id();
        uint4 rec = make_uint4(0u, 0u, 0u, 0u);
        if (i < n) rec = make_uint4(queue[0][i], queue[1][i], queue[2][i], queue[3][i]);
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const bool has = ((rec.w >> s) & 1u) != 0;
            const uint64_t mask = __builtin_amdgcn_ballot_w64(has);
            const uint32_t count = (uint32_t)__popcll(mask), at = have[s] + rank_in_mask(mask);
            uint32_t idx = seg_base + at;
            bool store = has;
            if (__builtin_expect(have[s] + count > seg_cap, 0)) {      // wave-uniform: some of these do not fit the segment any more
                const uint64_t spilled = __builtin_amdgcn_ballot_w64(has && at >= seg_cap);
                uint32_t base = 0;
                if (lane_id() == 0) base = atomicAdd(&over_count[s], (uint32_t)__popcll(spilled));
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                if (has && at >= seg_cap) {
                    const uint32_t o = base + rank_in_mask(spilled);
                    idx = over_base + o;
                    store = o < g.over_cap;
                }
            }
            if (store) {
                float* dst = cand_tile + (size_t)s * 3 * cap2 + idx;
                put(&dst[0], __uint_as_float(rec.x));
                put(&dst[cap2], __uint_as_float(rec.y));
                put(&dst[2 * (size_t)cap2], __uint_as_float(rec.z));
            }
            have[s] += count;
        }
    }
}

// The fused transform's form of the same move: the tile's candidates of a slot are ONE dense array of 16-byte records
// (od0, od1, od2, -), room reserved with one returning atomic per slot and flush -- the four of a flush issued together by
// four lanes, so a flush waits one memory round trip, normally once per work item and wave (a wave queues ~300 of its 4096
// pixels) -- and written with 16-byte write-through stores: the stage job of the same launch reads them from another CU.
// Records beyond the array's capacity are dropped; the count says so and the slot takes the slow exact path.
__device__ __forceinline__ void write_records_fused(const uint32_t (*__restrict__ queue)[kQueue2], uint32_t n, uint4* __restrict__ rec_tile, uint32_t cap, uint32_t* __restrict__ ncand) {
    if (n == 0) return;
    const uint32_t lane = lane_id();
    uint32_t count[kSlots] = {0u, 0u, 0u, 0u};
    for (uint32_t i0 = 0; i0 < n; i0 += kWave) {
        const uint32_t i = i0 + lane, flags = i < n ? queue[3][i] : 0u;
#pragma unroll
        for (int s = 0; s < kSlots; ++s) count[s] += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(((flags >> s) & 1u) != 0));
    }
    uint32_t base = 0;
    {
        const uint32_t mine = lane == 0 ? count[0] : (lane == 1 ? count[1] : (lane == 2 ? count[2] : count[3]));
        if (lane < (uint32_t)kSlots && mine) base = __hip_atomic_fetch_add(&ncand[lane], mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    uint32_t at0[kSlots];
#pragma unroll
    for (int s = 0; s < kSlots; ++s) at0[s] = (uint32_t)__builtin_amdgcn_readlane((int)base, s);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc(rec_tile, 0, (int)(kSlots * cap * 16u), 0x00020000);      // (rec_tile is wave-uniform; stores beyond the tile's records are dropped by the range check)
    for (uint32_t i0 = 0; i0 < n; i0 += kWave) {
        const uint32_t i = i0 + lane;
        sx_u4 rec = {0u, 0u, 0u, 0u};
        uint32_t flags = 0;
        if (i < n) {
            rec[0] = queue[0][i];
            rec[1] = queue[1][i];
            rec[2] = queue[2][i];
            flags = queue[3][i];
        }
#pragma unroll
        for (int s = 0; s < kSlots; ++s) {
            const bool has = ((flags >> s) & 1u) != 0;
            const uint64_t mask = __builtin_amdgcn_ballot_w64(has);
            const uint32_t at = at0[s] + rank_in_mask(mask);
            if (has && at < cap) __builtin_amdgcn_raw_buffer_store_b128(rec, rsrc, (int)(((uint32_t)s * cap + at) * 16u), 0, 16);      // aux 16: sc1 (write-through)
            at0[s] += (uint32_t)__popcll(mask);
        }
    }
}

// kEmit: the pass also leaves the tile as 8-bit codes for the reconstruct pass to read (Coded<F>, code_pack() in macenko.hip): the
// optical densities of a pack of grey levels then come from the code table instead of three v_log_f32 per pixel.
template <typename T, int V, int TPB, bool kInter, bool kFused = false, bool kEmit = false>
__device__ __forceinline__ void pass_a_item(const T* __restrict__ images, const Geometry& g, const Workspace& ws, int64_t tile, int chunk_id, int64_t item, PassAScratch<TPB>* sh, const LevelTables<T>& tb,
                                            const CodeTable<T, 2>* __restrict__ ct = nullptr) {
    const int64_t p_begin = (int64_t)chunk_id * g.chunk;
    const int64_t p_end = min(p_begin + (int64_t)g.chunk, g.pixels);
    const T* img = images + tile * 3 * g.pixels;
    GroupState& st = ws.state[tile];
    const PriorRecord* __restrict__ pr = &ws.prior[tile];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
    const bool speculate = get(&pr->mode) == 0;
    // The A operand of the tests' MFMA (see below): lane l holds A[row r = l & 31][k = 8 (l >> 5) + j], j = 0..7.  Output row
    // R = 8 a + 4 H + c carries test 4 a + c of the pixel in lane half H, so row R reads the pixel data of k-block H only.
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    typedef float float16v __attribute__((ext_vector_type(16)));
    half8 test_rows;
    {
        const uint32_t l = lane_id(), r = l & 31u, h = l >> 5, row_half = (r >> 2) & 1u, test = (r & 3u) + 4u * (r >> 3);
        const float4 coef = *reinterpret_cast<const float4*>(&pr->rows[test][0]);
        const bool mine = h == row_half;
        test_rows[0] = (_Float16)(mine ? coef.x : 0.0f);
        test_rows[1] = (_Float16)(mine ? coef.y : 0.0f);
        test_rows[2] = (_Float16)(mine ? coef.z : 0.0f);
        test_rows[3] = (_Float16)(mine ? coef.w : 0.0f);
        test_rows[4] = test_rows[5] = test_rows[6] = test_rows[7] = (_Float16)0.0f;
    }
    const float kw = get(&pr->kw), kx = get(&pr->kx);
    if (threadIdx.x < 2) sh->below[threadIdx.x] = 0;

    constexpr int kShortRun = 32 / V > 0 ? 32 / V : 1;      // packs per fp32 run -- the grouping of stats_item, bit for bit
    double acc[kPartial];
#pragma unroll
    for (int k = 0; k < kPartial; ++k) acc[k] = 0.0;
    float m[kPartial];
#pragma unroll
    for (int k = 0; k < kPartial; ++k) m[k] = 0.0f;

    uint32_t (*queue)[kQueue2] = sh->queue[wave];
    uint32_t n_q = 0, have[kSlots] = {0u, 0u, 0u, 0u}, below_a = 0, below_b = 0;
    float* cand_tile = ws.cand_od + (size_t)tile * kSlots * 3 * g.cap2;
    const uint32_t seg = (uint32_t)chunk_id * (TPB / kWave) + (uint32_t)wave, seg_base = seg * g.seg_cap;
    auto flush = [&]() {
        if constexpr (kFused)
            write_records_fused(queue, n_q, ws.cand_rec + (size_t)tile * kSlots * g.fused_cap, g.fused_cap, ws.ftile[tile].ncand);
        else
            write_records(queue, n_q, have, cand_tile, g, seg_base, st.over_count);
        n_q = 0;
    };

    const int64_t mine = (int64_t)threadIdx.x * V;
    PixelPacks<T, V, kInter> next;
    next.clear();
    if (p_begin + mine < p_end) next.load(img, g.pixels, p_begin + mine);
    __syncthreads();      // the scratch words above

    int in_run = 0;
    // (the loop exists twice, with and without the speculation: a run-time test per pixel is a branch per pixel)
    auto sweep = [&](auto with_tests) {
        constexpr bool kTests = decltype(with_tests)::value;
        constexpr int G = 1;      // pixels whose MFMAs are in flight together (measured: two together 48.3 us against 46.5 -- 16 more accumulator registers, and the other waves of the SIMD cover one MFMA's latency anyway)
        for (int64_t base_p = p_begin; base_p < p_end; base_p += (int64_t)TPB * V) {
            const bool live = base_p + mine < p_end;
            const PixelPacks<T, V, kInter> u = next;
            const int64_t p_next = base_p + (int64_t)TPB * V + mine;
            if (p_next < p_end) next.load(img, g.pixels, p_next);
            uint64_t live_mask = __builtin_amdgcn_ballot_w64(live);
            asm volatile("" : "+s"(live_mask));      // (opaque: otherwise the comparison behind it is redone for every pixel of the pack -- a 64-bit add and a 64-bit compare each time)
            // (kEmit: the pass moves 201 MB in and 34 MB of candidate records out at ~5 TB/s; the 50 MB of codes cost what 50 MB cost at that
            // rate, 46 -> 55 us -- 49 us with the stores left out, and no cheaper with the stores held back until just before the next
            // pack's request (59 us) or with conflict-free table copies: the reconstruct pass gets 20 us back)
            float od_quad[4][3];
            if constexpr (kEmit) {      // (the whole pack in front of the pixel loop: looked up quad by quad inside it, the pass took 61 us instead of 54)
                static_assert(!kEmit || V == 4, "one quad per pack");
                if (live) {
                    uint32_t word[3][1];
                    code_quad(u, 0, tb, *ct, g, ws, tile, od_quad, word);
                    store_codes<1>(g, ws, tile, base_p + mine, word);
                }
            }
#pragma unroll
            for (int i0 = 0; i0 < V; i0 += G) {
                float od[G][3];
                uint64_t valid[G];
                float16v forms[G];
#pragma unroll
                for (int gi = 0; gi < G; ++gi) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        if constexpr (kEmit) od[gi][c] = od_quad[(i0 + gi) % 4][c]; else od[gi][c] = od_of<T>(u.value(c, i0 + gi), tb);
                    }
                    const bool sel = od_selected(od[gi], false);
                    valid[gi] = __builtin_amdgcn_ballot_w64(sel) & live_mask;      // (the ballot of a bare comparison is the comparison's own mask; of `live && sel` it is a 0 / 1 register compared with 0 again)
                    // (the lanes of `valid` as the branch's mask: `live && sel` made the compiler work `live` out again for every pixel --
                    // a 64-bit add and a 64-bit compare -- because the comparison's result does not survive in vcc)
                    if (__builtin_amdgcn_inverse_ballot_w64(valid[gi])) {      // (stats_item multiplies by a 0 / 1 `keep` instead: the same bits, four instructions more)
                        const float* o = od[gi];
                        m[0] += 1.0f;
                        m[1] += o[0];
                        m[2] += o[1];
                        m[3] += o[2];
                        m[4] = fmaf(o[0], o[0], m[4]);
                        m[5] = fmaf(o[0], o[1], m[5]);
                        m[6] = fmaf(o[0], o[2], m[6]);
                        m[7] = fmaf(o[1], o[1], m[7]);
                        m[8] = fmaf(o[1], o[2], m[8]);
                        m[9] = fmaf(o[2], o[2], m[9]);
                    }
                    if constexpr (kTests) {
                        // Nine linear forms of the pixel on the matrix core: B[k = 8 h + j][col = l & 31] is the lane's own pixel
                        // (od0, od1, od2, 1, 0, 0, 0, 0) in fp16, so column c of the product mixes the pixels of lanes c (k-block 0)
                        // and c + 32 (k-block 1), and the rows sort them apart again (test_rows): forms[i] of EVERY lane is form i
                        // of its own pixel -- C/D layout: col = l & 31, row = (i & 3) + 8 (i >> 2) + 4 (l >> 5).  One instruction (32
                        // cycles of a pipe that is otherwise idle) instead of ~27 multiply-adds per pixel on the vector ALU, which
                        // is what bounds this pass.
                        typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
                        typedef uint32_t uint4v __attribute__((ext_vector_type(4)));
                        const fp16x2 p01 = __builtin_amdgcn_cvt_pkrtz(od[gi][0], od[gi][1]), p2c = __builtin_amdgcn_cvt_pkrtz(od[gi][2], 1.0f);
                        uint4v packed;
                        packed[0] = __builtin_bit_cast(uint32_t, p01);
                        packed[1] = __builtin_bit_cast(uint32_t, p2c);
                        packed[2] = packed[3] = 0u;
                        float16v zero;
#pragma unroll
                        for (int z = 0; z < 16; ++z) zero[z] = 0.0f;
                        forms[gi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(test_rows, __builtin_bit_cast(half8, packed), zero, 0, 0, 0);
                    }
                }
                if constexpr (kTests) {
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        const float* o = od[gi];
                        const float16v& f = forms[gi];
                        const float mrg = fmaf(kw, fabsf(f[kRowW]), fmaf(kx, fabsf(o[0]) + fabsf(o[1]) + fabsf(o[2]), 1e-6f));
                        // angle slots: "below" is cross(d, th) < -m, "above" cross(d, th) > m; an open side has the zero row: never true
                        const uint64_t lt_a = __builtin_amdgcn_ballot_w64(f[kRowBelowA] < -mrg), gt_a = __builtin_amdgcn_ballot_w64(f[kRowAboveA] > mrg);
                        const uint64_t lt_b = __builtin_amdgcn_ballot_w64(f[kRowBelowB] < -mrg), gt_b = __builtin_amdgcn_ballot_w64(f[kRowAboveB] > mrg);
                        below_a += (uint32_t)__popcll(valid[gi] & lt_a);
                        below_b += (uint32_t)__popcll(valid[gi] & lt_b);
                        const uint64_t c_a = valid[gi] & ~lt_a & ~gt_a, c_b = valid[gi] & ~lt_b & ~gt_b;
                        // concentration slots: every pixel takes part; a candidate reaches the threshold at either end direction
                        const uint64_t c_c = live_mask & (__builtin_amdgcn_ballot_w64(f[kRowConc] >= -mrg) | __builtin_amdgcn_ballot_w64(f[kRowConc + 1] >= -mrg));
                        const uint64_t c_d = live_mask & (__builtin_amdgcn_ballot_w64(f[kRowConc + 2] >= -mrg) | __builtin_amdgcn_ballot_w64(f[kRowConc + 3] >= -mrg));
                        const uint64_t any = c_a | c_b | c_c | c_d;
                        if (any) {      // wave-uniform
                            uint32_t flags = __builtin_amdgcn_inverse_ballot_w64(c_a) ? 1u : 0u;
                            flags |= __builtin_amdgcn_inverse_ballot_w64(c_b) ? 2u : 0u;
                            flags |= __builtin_amdgcn_inverse_ballot_w64(c_c) ? 4u : 0u;
                            flags |= __builtin_amdgcn_inverse_ballot_w64(c_d) ? 8u : 0u;
                            if (__builtin_amdgcn_inverse_ballot_w64(any)) {
                                const uint32_t at = n_q + rank_in_mask(any);
                                queue[0][at] = __float_as_uint(o[0]);
                                queue[1][at] = __float_as_uint(o[1]);
                                queue[2][at] = __float_as_uint(o[2]);
                                queue[3][at] = flags;
                            }
                            n_q += (uint32_t)__popcll(any);
                        }
                        if (__builtin_expect(n_q > (uint32_t)(kQueue2 - kWave), 0)) flush();      // the next pixel adds at most 64 records
                    }
                }
                // (narrow pixels: 8 or 16 of these bodies in a row, and the scheduler would start all their table reads and
                // logarithms together -- 48 optical densities live at once, a 200-register kernel at half the occupancy)
                if constexpr (V > 4) __builtin_amdgcn_sched_barrier(0);
            }
            if (++in_run == kShortRun) {
#pragma unroll
                for (int k = 0; k < kPartial; ++k) {
                    acc[k] += (double)m[k];
                    m[k] = 0.0f;
                }
                in_run = 0;
            }
        }
    };
    if (speculate) sweep(std::true_type{}); else sweep(std::false_type{});
    if (in_run) {
#pragma unroll
        for (int k = 0; k < kPartial; ++k) acc[k] += (double)m[k];
    }

    // ---- the work item's partial moments: exactly the reduction of stats_item
    StatsScratch<TPB>* ss = &sh->stats;
#pragma unroll
    for (int k = 0; k < kPartial; ++k) {
        const double s = wave_total_f64(acc[k]);
        if (lane_id() == kWave - 1) ss->red[wave][k] = s;
    }
    if (speculate) {
        flush();
        if (lane_id() == 0) {
            if constexpr (!kFused) {
                uint32_t* counts = ws.seg_count + ((size_t)tile * kSlots) * g.n_seg + seg;
#pragma unroll
                for (int s = 0; s < kSlots; ++s) put(&counts[(size_t)s * g.n_seg], have[s]);
            }
            if (below_a) atomicAdd(&sh->below[0], below_a);
            if (below_b) atomicAdd(&sh->below[1], below_b);
        }
    }
    __syncthreads();
    if (threadIdx.x < kPartial) {
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < TPB / kWave; ++w) s += ss->red[w][threadIdx.x];
        if constexpr (kFused) st_agent(&ws.partial[item * kPartial + threadIdx.x], s); else put(&ws.partial[item * kPartial + threadIdx.x], s);
    }
    if (speculate && threadIdx.x == kPartial) {      // (no value comes back: nothing waits for these)
        if (sh->below[0]) atomicAdd(&st.below[0], sh->below[0]);
        if (sh->below[1]) atomicAdd(&st.below[1], sh->below[1]);
    }
    double kept_total = 0.0;
#pragma unroll
    for (int w = 0; w < TPB / kWave; ++w) kept_total += ss->red[w][0];      // workgroup-uniform
    // (see stats_item: a work item without kept pixels also leaves the moments of ALL its pixels)
    if (__builtin_expect(kept_total < 3.0, 0)) {
        if (!tile_has_three_kept_witnesses<T, TPB, kInter>(img, g.pixels)) {      // (see there: only where the TILE may be blank)
            stats_item_all_pixels<T, V, TPB, kInter>(img, g.pixels, p_begin, p_end, ws.partial_all + item * kPartial, ss);
            if constexpr (kFused) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // (plain stores in there: written back before the flag below; rare path)
        }
    }
    if constexpr (kFused) {
        // publish: every wave has drained its write-through stores and atomics, then ONE lane counts the work item in
        drain_stores();
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_fetch_add(&ws.ftile[tile].a_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// kDense: the candidates go to ONE dense array of 16-byte records per tile and slot, written through (the form the fused launch
// uses, see write_records_fused) instead of a segment per wave: no dirty candidate lines for the kernel boundary to write back
// (the gap between this launch and the stage was 4.7 us with 25 MB of them), one load per candidate in the stage instead of
// three, and a third of the workspace.  Tiles up to 512 x 512; larger ones keep the segments.
// (Occupancy is not what holds this kernel back: built for five waves per SIMD -- 93 registers, a 448-record queue, 29 KB of LDS --
// and run on 20 work items per 512 x 512 tile, all 1280 resident, the call takes 156.9 us against 156.6 us at four waves and 16
// items; tools/ab_items.py on a debug build.)
template <typename T, int V, bool kInter = false, bool kDense = false>
__global__ __launch_bounds__(kStreamThreads, (Codable<T, V, kInter>::value ? 4 : 1)) void pass_a_kernel(const T* __restrict__ images, Geometry g, Workspace ws) {
    __shared__ PassAScratch<kStreamThreads> sh;
    __shared__ LevelTables<T> tb;
    tb.fill();
    if constexpr (Codable<T, V, kInter>::value) {
        if (g.code_epoch != 0u) {      // (uniform over the launch)
            __shared__ CodeTable<T, 2> ct;
            ct.fill();
            pass_a_item<T, V, kStreamThreads, kInter, kDense, true>(images, g, ws, blockIdx.x / g.blocks_per_tile, blockIdx.x % g.blocks_per_tile, blockIdx.x, &sh, tb, &ct);
            return;
        }
    }
    pass_a_item<T, V, kStreamThreads, kInter, kDense>(images, g, ws, blockIdx.x / g.blocks_per_tile, blockIdx.x % g.blocks_per_tile, blockIdx.x, &sh, tb);
}

// ------------------------------------------------------------------------------------------------
// K2 / K3: exact order statistic of one slot from its candidates, and the proof that it is the tile's
// ------------------------------------------------------------------------------------------------
struct alignas(16) SlotScratch {
    TileScratch t;                     // radix_select_stream / rank_pick machinery; t.keys is the list of the picked bin
    uint32_t keys[kLdsKeys];
    float vecs[6], pinv[6], he[6];
    double check[8];
    uint32_t lo, hi, n_list, bin, rank_in_bin, result, range_first, range_last, bin_count;
    int ok, use_all;
    unsigned long long n_sel;
    uint32_t seg_prefix[kMaxSegments + 1], seg_total, seg_overflow, over_n;
    uint32_t dense_n;                  // dense candidate arrays: candidates of the slot (clamped to the array's capacity)
    uint32_t own_key;
    int partner_ok;
    uint32_t k_floor, k_ceil, n_under;      // concentration slots: keys below k_floor are counted, not ranked; keys above k_ceil share the last bin
};

// The slot's candidates lie in one segment per wave of pass A (+ the tile's overflow area): prefix sums of the segment fills
// (one wave, four segments per lane), so that candidate i of the slot is entry i - prefix[k] of segment k; the overflow entries
// follow as candidates prefix[n_seg] ... .  Needs a barrier before the prefix is used.
__device__ __forceinline__ void segment_prefix(SlotScratch* sh, const Geometry& g, const Workspace& ws, int tile, int slot) {
    if (threadIdx.x < kWave) {
        const uint32_t* counts = ws.seg_count + ((size_t)tile * kSlots + slot) * g.n_seg;
        uint32_t c[4], sum = 0, raw = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * (int)threadIdx.x + u;
            const uint32_t v = k < g.n_seg ? get(&counts[k]) : 0u;
            raw += v;
            c[u] = min(v, g.seg_cap);
            sum += c[u];
        }
        const uint32_t incl = wave_scan_u32(sum);
        uint32_t run = incl - sum;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int k = 4 * (int)threadIdx.x + u;
            if (k <= g.n_seg) sh->seg_prefix[k] = run;
            run += c[u];
        }
        const uint32_t total_raw = wave_total_u32(raw);
        if (threadIdx.x == kWave - 1) {
            const uint32_t spilled = get(&ws.state[tile].over_count[slot]);
            sh->seg_prefix[g.n_seg] = incl;      // (n_seg <= 4 * 64: the last lane's inclusive sum is the total)
            sh->seg_total = total_raw;
            sh->over_n = min(spilled, g.over_cap);
            sh->seg_overflow = spilled > g.over_cap ? 1u : 0u;
        }
    }
}

// Every candidate of the slot once: a team of blockDim / n_seg consecutive threads per segment.  The first kPre records of a
// thread are requested at the TOP of the kernel, before the segment fills are known (entries beyond a fill are in-bounds
// garbage that is never used), so the stage pays one memory round trip for counters, moments and candidates together; a
// plain one-record-per-iteration loop paid one per record (7-9 us per stage).  fn(i, od): i is the candidate's index in the
// slot (segments in order), od its optical density.
template <int kPre> struct CandPrefetch {
    float od[kPre][3];
};
template <int kPre>
__device__ __forceinline__ void prefetch_candidates(CandPrefetch<kPre>& pf, const Geometry& g, const float* __restrict__ c0) {
    const uint32_t team = blockDim.x / (uint32_t)g.n_seg, k = threadIdx.x / team, r = threadIdx.x - k * team;
    const float* src = c0 + (size_t)k * g.seg_cap;
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        const uint32_t o = r + u * team;
        const bool in = k < (uint32_t)g.n_seg && o < g.seg_cap;
#pragma unroll
        for (int c = 0; c < 3; ++c) pf.od[u][c] = in ? get(&src[(size_t)c * g.cap2 + o]) : 0.0f;
    }
}
template <int kPre, class Fn>
__device__ __forceinline__ void for_each_candidate(const CandPrefetch<kPre>& pf, const SlotScratch* sh, const Geometry& g, const float* __restrict__ c0, Fn fn) {
    const uint32_t team = blockDim.x / (uint32_t)g.n_seg, k = threadIdx.x / team, r = threadIdx.x - k * team;
    const bool mine = k < (uint32_t)g.n_seg;
    const uint32_t first = mine ? sh->seg_prefix[k] : 0u, cnt = mine ? sh->seg_prefix[k + 1] - first : 0u;
    const float* src = c0 + (size_t)k * g.seg_cap;
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        const uint32_t o = r + u * team;
        if (o < cnt) fn(first + o, pf.od[u]);
    }
    constexpr int kFlight = 4;
    // the overflow area (rare): every thread, plain strided sweep
    for (uint32_t o = threadIdx.x; o < sh->over_n; o += blockDim.x) {
        const float* p = c0 + (size_t)g.n_seg * g.seg_cap + o;
        const float od[3] = {get(&p[0]), get(&p[g.cap2]), get(&p[2 * (size_t)g.cap2])};
        fn(sh->seg_prefix[g.n_seg] + o, od);
    }
    if (k >= (uint32_t)g.n_seg) return;
    for (uint32_t off = r + kPre * team; off < cnt; off += team * kFlight) {
        float od[kFlight][3];
#pragma unroll
        for (int u = 0; u < kFlight; ++u) {
            const uint32_t o = off + u * team;
#pragma unroll
            for (int c = 0; c < 3; ++c) od[u][c] = o < cnt ? get(&src[(size_t)c * g.cap2 + o]) : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kFlight; ++u) {
            const uint32_t o = off + u * team;
            if (o < cnt) fn(first + o, od[u]);
        }
    }
}

// The same two steps over a DENSE candidate array (pass_a_kernel<kDense>): candidate i is record i; thread t takes i = t, t + blockDim, ...
// -- adjacent lanes read adjacent records (one wave-instruction = 1 KB).  The first kPre records of a thread are requested before the
// count is known (the range check of the buffer descriptor answers requests beyond the tile's records with zeros).
template <int kPre>
__device__ __forceinline__ void prefetch_candidates_dense(CandPrefetch<kPre>& pf, __amdgpu_buffer_rsrc_t rsrc) {
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        const sx_u4 rec = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((threadIdx.x + (uint32_t)u * blockDim.x) * 16u), 0, 0);
#pragma unroll
        for (int c = 0; c < 3; ++c) pf.od[u][c] = __uint_as_float(rec[c]);
    }
}
template <int kPre, class Fn>
__device__ __forceinline__ void for_each_candidate_dense(const CandPrefetch<kPre>& pf, uint32_t n, __amdgpu_buffer_rsrc_t rsrc, Fn fn) {
#pragma unroll
    for (int u = 0; u < kPre; ++u) {
        const uint32_t i = threadIdx.x + (uint32_t)u * blockDim.x;
        if (i < n) fn(i, pf.od[u]);
    }
    constexpr int kFlight = 4;
    for (uint32_t i0 = threadIdx.x + (uint32_t)kPre * blockDim.x; i0 < n; i0 += blockDim.x * kFlight) {
        float od[kFlight][3];
#pragma unroll
        for (int u = 0; u < kFlight; ++u) {
            const sx_u4 rec = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)((i0 + (uint32_t)u * blockDim.x) * 16u), 0, 0);
#pragma unroll
            for (int c = 0; c < 3; ++c) od[u][c] = __uint_as_float(rec[c]);
        }
#pragma unroll
        for (int u = 0; u < kFlight; ++u) {
            const uint32_t i = i0 + (uint32_t)u * blockDim.x;
            if (i < n) fn(i, od[u]);
        }
    }
}
// the slot's count (one thread; a barrier before it is used): what segment_prefix() leaves for the segment form
__device__ __forceinline__ void dense_count(SlotScratch* sh, const Geometry& g, const Workspace& ws, int tile, int slot) {
    if (threadIdx.x == kWave - 1) {
        const uint32_t n_raw = get(&ws.ftile[tile].ncand[slot]);
        sh->seg_total = n_raw;
        sh->dense_n = min(n_raw, g.fused_cap);
        sh->over_n = 0;
        sh->seg_overflow = n_raw > g.fused_cap ? 1u : 0u;
    }
}

__device__ __forceinline__ uint32_t slot_key(const SlotScratch* sh, const uint32_t* __restrict__ spill, uint32_t i) { return i < (uint32_t)kLdsKeys ? sh->keys[i] : get(&spill[i - kLdsKeys]); }

// The selection's scratch words; call before the keys are produced (a barrier must lie between this and select_slot_keys).
__device__ __forceinline__ void select_prepare(SlotScratch* sh) {
    if (threadIdx.x == 0) {
        sh->lo = 0xFFFFFFFFu;
        sh->hi = 0u;
        sh->n_list = 0;
        sh->result = 0;
        sh->n_under = 0;
    }
    for (int i = threadIdx.x; i < 512; i += blockDim.x) (&sh->t.hist_c[0][0])[i] = 0;
}
// The range of the keys a thread produced (the phase that makes the keys also finds their range: one sweep less)
__device__ __forceinline__ void publish_range(SlotScratch* sh, uint32_t mn, uint32_t mx) {
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    if (lane_id() == 0 && mn != 0xFFFFFFFFu) {
        atomicMin(&sh->lo, mn);
        atomicMax(&sh->hi, mx);
    }
}

// exact element of 0-based rank `rank` among the n keys of the slot (n >= 1, rank < n); whole workgroup, uniform result.
// select_prepare() and publish_range() of every key come first, then a barrier.  A thread sweeps a CONTIGUOUS share of the
// keys, so the lanes of a wave touch candidates far apart in the tile: neighbours in the arrays are neighbours in the image,
// with nearly the same key -- the same histogram bin, and LDS atomics on one bin take their turns.
// Two levels of 256 value-linear bins: over the keys' range, then -- only when the picked bin holds more keys than the short
// list -- over that bin's own key range.  Tiles from 8-bit data tie heavily (the same colour, the same key): a bin there
// holds thousands of keys of a handful of values, which the second level separates or recognises as one value.
// k_floor / k_ceil: keys below k_floor take no part (the caller has counted them and taken them off `rank`); the bins span
// [max(lo, k_floor), min(hi, k_ceil)], keys beyond share the last bin.
__device__ uint32_t select_slot_keys(SlotScratch* sh, const uint32_t* __restrict__ spill, uint32_t n, uint32_t rank, bool level0_filled, unsigned long long* dbg = nullptr, uint32_t k_floor = 0u, uint32_t k_ceil = 0xFFFFFFFFu) {
    const uint32_t lane = lane_id();
    const int wave = threadIdx.x / kWave;
    // (an ODD share: the lanes of a wave read words `share` apart, and an even stride folds them onto a fraction of the LDS banks)
    const uint32_t share = ((n + blockDim.x - 1) / blockDim.x) | 1u, i_begin = min(threadIdx.x * share, n), i_end = min(i_begin + share, n);
    const uint32_t lo = max(sh->lo, k_floor), hi = max(min(sh->hi, k_ceil), lo);
    double origin = bin_origin_for(lo), scale = bin_scale_for(lo, hi);      // (lo, hi: the keys' range, or the range the caller binned them over)
    uint32_t k_first = k_floor, k_last = 0xFFFFFFFFu, want = rank;      // the keys still in play: k_first <= key <= k_last
    uint32_t picked = 0;
    bool by_bin = false;      // the picked bin's keys are told by bin_of(), not by a key range
    uint32_t* hist = sh->t.hist_c[0];
    for (int level = 0; level < 2; ++level) {
        if (level > 0 || !level0_filled) {
            for (uint32_t i = i_begin; i < i_end; ++i) {
                const uint32_t k = slot_key(sh, spill, i);
                if (k >= k_first && k <= k_last) atomicAdd(&hist[bin_of(k, origin, scale)], 1u);
            }
            __syncthreads();
        }
        if (dbg && threadIdx.x == 0) dbg[level] = (unsigned long long)wall_clock64();
        if (wave == 0) {
            uint32_t b, rb;
            scan_pick32(hist, want, b, rb);
            if (lane == 0) {
                const uint32_t in_bin = hist[b];
                sh->bin = b;
                sh->rank_in_bin = rb;
                sh->bin_count = in_bin;
                // The bin's key range is only needed where the search goes on INSIDE the bin; a short bin is listed by asking every
                // key for its bin again (one subtraction and one multiplication per key instead of ~150 dependent instructions on
                // this one lane -- fp64 edges walked to the exact boundary -- while the workgroup waits).
                if (in_bin > (uint32_t)kShortList || level == 1) {
                    uint32_t first, last;
                    bin_key_range(b, origin, scale, first, last);
                    sh->range_first = max(first, k_first);
                    sh->range_last = min(last, k_last);
                }
            }
        }
        __syncthreads();
        want = sh->rank_in_bin;
        picked = sh->bin;
        const uint32_t in_bin = sh->bin_count;
        if (in_bin <= (uint32_t)kShortList && level == 0) {      // uniform; the rule
            by_bin = true;
            break;
        }
        k_first = sh->range_first;
        k_last = sh->range_last;
        if (level == 1) break;
        // (tighten to the keys actually present in the bin?  not needed: the bin's own range is already 1/256 of the first)
        if (k_first == k_last) return k_first;                      // one key value fills the bin
        hist = sh->t.hist_c[1];      // zeroed by select_prepare
        origin = bin_origin_for(k_first);
        scale = bin_scale_for(k_first, k_last);
    }
    if (!by_bin && k_first == k_last) return k_first;
    // the keys of the picked bin (at most the short list's worth, ties aside), listed and ranked by counting
    uint32_t* list = &sh->t.keys[0][0];      // 2 * kSample entries
    for (uint32_t i = i_begin; i < i_end; ++i) {
        const uint32_t k = slot_key(sh, spill, i);
        if (k >= k_first && k <= k_last && (!by_bin || bin_of(k, origin, scale) == picked)) {
            const uint32_t at = atomicAdd(&sh->n_list, 1u);
            if (at < (uint32_t)(2 * kSample)) list[at] = k;
        }
    }
    __syncthreads();
    if (dbg && threadIdx.x == 0) dbg[2] = (unsigned long long)wall_clock64();
    const uint32_t n_list = sh->n_list;
    if (__builtin_expect(n_list <= (uint32_t)kShortList, 1)) {
        rank_pick(list, n_list, want, threadIdx.x, blockDim.x, &sh->result);
        __syncthreads();
        return sh->result;
    }
    if (n_list <= (uint32_t)(2 * kSample)) return radix_select_stream((unsigned long long)n_list, (unsigned long long)want, [list](unsigned long long i, uint32_t& k) { k = list[i]; return true; }, &sh->t);
    // more keys than the list holds: radix rounds over all the keys
    return radix_select_stream((unsigned long long)n, (unsigned long long)rank, [sh, spill, k_floor](unsigned long long i, uint32_t& k) { k = slot_key(sh, spill, (uint32_t)i); return k >= k_floor; }, &sh->t);
}

// M = F^-1 X for the 3x3 frame F = [a0 a1 an] (columns) and the three-vector X: coordinates of X in the prior frame.
// fp32 with the hardware reciprocal: the results feed checks that carry several per cent of slack (a chain of fp64
// divisions and square roots on one lane was 1-2 us of every stage).
__device__ inline void frame_coordinates(const PriorRecord* pr, const float x[3], float out[3]) {
    const float f00 = pr->a0[0], f10 = pr->a0[1], f20 = pr->a0[2];
    const float f01 = pr->a1[0], f11 = pr->a1[1], f21 = pr->a1[2];
    const float f02 = pr->an[0], f12 = pr->an[1], f22 = pr->an[2];
    const float c00 = f11 * f22 - f12 * f21, c01 = f12 * f20 - f10 * f22, c02 = f10 * f21 - f11 * f20;
    const float det = f00 * c00 + f01 * c01 + f02 * c02;
    const float inv = det != 0.0f ? __builtin_amdgcn_rcpf(det) : 0.0f;
    // inverse = adj / det, adj = cofactor^T
    const float i00 = c00 * inv, i01 = (f02 * f21 - f01 * f22) * inv, i02 = (f01 * f12 - f02 * f11) * inv;
    const float i10 = c01 * inv, i11 = (f00 * f22 - f02 * f20) * inv, i12 = (f02 * f10 - f00 * f12) * inv;
    const float i20 = c02 * inv, i21 = (f01 * f20 - f00 * f21) * inv, i22 = (f00 * f11 - f01 * f10) * inv;
    out[0] = i00 * x[0] + i01 * x[1] + i02 * x[2];
    out[1] = i10 * x[0] + i11 * x[1] + i12 * x[2];
    out[2] = i20 * x[0] + i21 * x[1] + i22 * x[2];
}

// the tile's moments summed in the order plane_stage uses (index order; bit for bit the same doubles), then the plane
template <typename T>
__device__ void exact_plane(const Geometry& g, const Workspace& ws, int tile, SlotScratch* sh) {
    const int64_t first = (int64_t)tile * g.blocks_per_tile;
    if (threadIdx.x < kPartial) {
        double running = 0.0;
        for (int64_t b0 = 0; b0 < g.blocks_per_tile; b0 += 16) {      // sixteen loads in flight, added in index order
            double part[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) part[u] = b0 + u < g.blocks_per_tile ? get(&ws.partial[(first + b0 + u) * kPartial + threadIdx.x]) : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (b0 + u < g.blocks_per_tile) running += part[u];
        }
        sh->t.mom[threadIdx.x] = running;
    }
    if (threadIdx.x == kPartial) sh->t.mom[kPartial] = (double)g.pixels;
    if (threadIdx.x > kPartial && threadIdx.x < kMoments) sh->t.mom[threadIdx.x] = 0.0;
    __syncthreads();
    if (__builtin_expect(sh->t.mom[0] < 3.0, 0)) {      // uniform, rare: every work item of such a tile left its all-pixel sums
        __syncthreads();
        if (threadIdx.x < kPartial) {
            double running = 0.0;
            for (int64_t b = 0; b < g.blocks_per_tile; ++b) running += get(&ws.partial_all[(first + b) * kPartial + threadIdx.x]);
            sh->t.mom[kPartial + threadIdx.x] = running;
        }
        __syncthreads();
    }
    if (threadIdx.x < 2) {
        double cov[9];
        bool use_all;
        unsigned long long n_sel;
        float vecs[6];
        plane_from_moments<true>(sh->t.mom, true, cov, vecs, use_all, n_sel);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) sh->vecs[i] = vecs[i];
            sh->use_all = use_all ? 1 : 0;
            sh->n_sel = n_sel;
            if ((blockIdx.x & 1) == 0) {
                GroupState& st = ws.state[tile];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    put(&st.vecs[i], vecs[i]);
                    put(&st.rec[0].coef[i], vecs[i]);
                }
#pragma unroll
                for (int k = 0; k < kMoments; ++k) put(&st.mom[k], sh->t.mom[k]);
#pragma unroll
                for (int i = 0; i < 9; ++i) put(&st.cov[i], cov[i]);
                put(&st.use_all, use_all ? 1 : 0);
                put(&st.rec[0].use_all, use_all ? 1 : 0);
                put(&st.n_sel, n_sel);
            }
        }
    }
    __syncthreads();
}

// The proof obligations of an angle slot that do not depend on the answer, and the key range its candidates can span
// (one thread).  check[0..1]: keys of the mapped boundaries, check[2..3]: 1 if that side is open.
__device__ inline bool phi_slot_check(const PriorRecord* pr, const float (&v)[6], int j, double (&check)[8], float rot_limit) {
    // the prior frame against the exact plane: V = F M; in-plane part Rt (t = Rt th + nu w), tilt nu
    float m0[3], m1[3];
    const float v0[3] = {v[0], v[2], v[4]}, v1[3] = {v[1], v[3], v[5]};
    frame_coordinates(pr, v0, m0);
    frame_coordinates(pr, v1, m1);
    const float r00 = m0[0], r01 = m0[1], r10 = m1[0], r11 = m1[1], nu0 = m0[2], nu1 = m1[2];
    const float det = r00 * r11 - r01 * r10;
    const float nu = __builtin_amdgcn_sqrtf(nu0 * nu0 + nu1 * nu1);
    const float stretch = __builtin_amdgcn_sqrtf(r00 * r00 + r01 * r01 + r10 * r10 + r11 * r11);      // >= the largest singular value of Rt
    const float kw = get(&pr->kw), kx = get(&pr->kx);
    bool good = det > 0.9f && stretch < 1.6f && r00 > 0.8f && r11 > 0.8f && fabsf(r01) + fabsf(r10) < rot_limit && kx >= 1.2e-3f;
    // every kept pixel's exact angle inside (0, pi): t1 = r10 th0 + r11 th1 + nu1 w with th1 >= min(a1) |od|_1, ... (prior_kernel)
    {
        const float a1_min = fminf(pr->a1[0], fminf(pr->a1[1], pr->a1[2]));
        const float a0_max = fmaxf(fabsf(pr->a0[0]), fmaxf(fabsf(pr->a0[1]), fabsf(pr->a0[2])));
        const float an_max = fmaxf(fabsf(pr->an[0]), fmaxf(fabsf(pr->an[1]), fabsf(pr->an[2])));
        if (!(r11 * a1_min > 1.05f * (fabsf(r10) * a0_max + nu * an_max))) good = false;
    }
    // the two boundaries of this slot as pass A tested them: row . od = g0 th0 + g1 th1 + g2 w = cross(d', th) + g2 w with
    // d' = (g1, -g0).  "row . od < -m" puts the exact t clockwise of D = Rt d' as long as the margin's kw |w| covers both
    // the row's own out-of-plane part g2 and the tilt: det (kw - |g2|) >= |Rt d'| |nu|; the fp16 rounding of the pixel is
    // covered by kx (rows with |entries| <= 1).  The answer has to lie strictly between the keys of the two D.
    const uint32_t open = get(&pr->open);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
        const float* row = pr->rows[2 * j + e];
        float gq[3];
        frame_coordinates(pr, row, gq);
        const float dx = gq[1], dy = -gq[0];
        const float mx = r00 * dx + r01 * dy, my = r10 * dx + r11 * dy;
        const bool is_open = ((open >> (2 * j + e)) & 1u) != 0;
        check[e] = is_open ? (e ? 2.0 : 0.0) : (double)diamond_angle(my, mx);
        check[2 + e] = is_open ? 1.0 : 0.0;
        if (!is_open) {
            const float dn = __builtin_amdgcn_sqrtf(dx * dx + dy * dy);
            if (!(my - 0.04f * fabsf(mx) > 0.0f)) good = false;      // a boundary outside (0, pi)
            if (!(fmaxf(fabsf(row[0]), fmaxf(fabsf(row[1]), fabsf(row[2]))) <= 1.05f)) good = false;
            if (!(nu * stretch * dn * 1.05f <= (kw - fabsf(gq[2])) * det)) good = false;
        }
    }
    if (!(check[1] > check[0])) good = false;
    return good;
}

// The same for a concentration slot: the cone check, the bound Theta every non-candidate stays below, and the range of the bins.
__device__ inline bool conc_slot_check(const PriorRecord* pr, const float (&pinv)[6], int j, double& theta_out, float& hi_out) {
    // row j of the pseudo-inverse in the prior frame: with q = F^T x = (th0, th1, w) and x = F^-T q, p . x = (F^-1 p) . q
    const float p[3] = {pinv[3 * j], pinv[3 * j + 1], pinv[3 * j + 2]};
    float gq[3], g1[3], g2[3];
    frame_coordinates(pr, p, gq);
    // the slot's two test rows the same way: row_e . od = u_e . th + gamma_e w, threshold T_e = -row_e[3]
    const float* row1 = pr->rows[kRowConc + 2 * j];
    const float* row2 = pr->rows[kRowConc + 2 * j + 1];
    frame_coordinates(pr, row1, g1);
    frame_coordinates(pr, row2, g2);
    const float u1x = g1[0], u1y = g1[1], t1 = -row1[3], u2x = g2[0], u2y = g2[1], t2 = -row2[3];
    const float det = u1x * u2y - u1y * u2x;
    const float kw = get(&pr->kw), kx = get(&pr->kx);
    if (!(fabsf(det) > 1e-7f)) return false;
    // (g0, g1) = alpha u1 + beta u2 with alpha, beta >= 0 (inside the cone): a pixel that failed both tests has
    // p . od < alpha T1 + beta T2 - (alpha + beta) m + (alpha |gamma1| + beta |gamma2| + |g2|) |w| + rounding
    const float inv = __builtin_amdgcn_rcpf(det);
    const float alpha = (gq[0] * u2y - gq[1] * u2x) * inv, beta = (u1x * gq[1] - u1y * gq[0]) * inv;
    const float l1 = fabsf(p[0]) + fabsf(p[1]) + fabsf(p[2]);
    const float rmax = fmaxf(fmaxf(fabsf(row1[0]), fmaxf(fabsf(row1[1]), fabsf(row1[2]))), fmaxf(fabsf(row2[0]), fmaxf(fabsf(row2[1]), fabsf(row2[2]))));
    // (alpha, beta carry ~1e-6 of rounding: a cone edge case fails the check rather than passing it by luck)
    const bool good = alpha >= 1e-5f * l1 && beta >= 1e-5f * l1 && rmax <= 1.05f && kx >= 1.2e-3f &&
                      (alpha + beta) * kw >= 1.05f * (alpha * fabsf(g1[2]) + beta * fabsf(g2[2]) + fabsf(gq[2])) && (alpha + beta) * (kx - 1.1e-3f) >= 3e-6f * l1 && t1 < 1e30f && t2 < 1e30f;
    theta_out = (double)alpha * (double)t1 + (double)beta * (double)t2;
    // the largest value the sample saw, with room: keys beyond it share the last bin
    const float span = alpha * fmaxf(get(&pr->cmax[2 * j]) - t1, 0.0f) + beta * fmaxf(get(&pr->cmax[2 * j + 1]) - t2, 0.0f);
    hi_out = (float)theta_out + fmaxf(1.25f * span, 1e-3f * fabsf((float)theta_out) + 1e-6f);
    return good;
}

// K2: both per-tile stages in ONE launch, two workgroups per tile.  Workgroup j works out angle percentile j, hands its key
// to its partner through an 8-byte {key, tag} granule (one agent-scope store, polled with agent-scope loads: the form
// MI355X_MICROARCH.md lists as needing no further ordering), then works out concentration j -- whose candidates it requested
// before it started to wait.  As two launches the stages cost a kernel boundary plus ~3.5 us of launch ramp more, and the
// second could not start its loads early.  The wait is bounded: a partner that does not show up (it would have to be
// unscheduled while this workgroup spins -- the pair has adjacent block indices) is replaced by the slow exact select here.
template <typename T, bool kDense = false>
__global__ __launch_bounds__(kGroupThreads) void estimate_stage_kernel(const T* __restrict__ images, Geometry g, Workspace ws, const float* __restrict__ target_max_conc,
                                                                       uint32_t* __restrict__ key_scratch) {
    __shared__ SlotScratch sh;
    const int tile = blockIdx.x >> 1, j = blockIdx.x & 1, slot = 2 + j;
    // the slow exact path's key scratch: plane j of the tile's (not yet written) output, one word per pixel
    uint32_t* keep = key_scratch ? key_scratch + ((size_t)tile * 3 + j) * (size_t)g.pixels * (sizeof(T) >= 4 ? sizeof(T) / 4 : 1) : nullptr;
    GroupState& st = ws.state[tile];
    const PriorRecord* pr = &ws.prior[tile];
    const uint32_t below = get(&st.below[j]), spec = get(&st.spec);
    const int mode = get(&pr->mode);
    const bool stamps = j == 0;
    if (stamps) SX_STAMP(st, 8);
    const float* c0 = ws.cand_od + ((size_t)tile * kSlots + j) * 3 * g.cap2;
    // (keys of a slot beyond the LDS array: the per-wave segments' total, or the dense array's capacity)
    const size_t spill_words = kDense ? (g.fused_cap > (uint32_t)kLdsKeys ? g.fused_cap - kLdsKeys : 0u) : (g.cap2 > (uint32_t)kLdsKeys ? g.cap2 - kLdsKeys : 0u);
    uint32_t* const spill_area = kDense ? ws.dense_spill : ws.key_spill;
    // (tile and slot are workgroup-uniform; said so, the descriptors stay in scalar registers)
    const uint4* rec_tile = ws.cand_rec + (size_t)__builtin_amdgcn_readfirstlane(tile) * kSlots * g.fused_cap;
    const auto rsrc_phi = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(rec_tile + (size_t)__builtin_amdgcn_readfirstlane(j) * g.fused_cap), 0, (int)(g.fused_cap * 16u), 0x00020000);
    const auto rsrc_conc = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint4*>(rec_tile + (size_t)__builtin_amdgcn_readfirstlane(slot) * g.fused_cap), 0, (int)(g.fused_cap * 16u), 0x00020000);
    float v[6];
    bool use_all;
    unsigned long long rank_other;
    {   // ------------------------------------------------ angle percentile j
        CandPrefetch<8> pf;
        if constexpr (kDense) prefetch_candidates_dense(pf, rsrc_phi); else prefetch_candidates(pf, g, c0);
        select_prepare(&sh);
        if constexpr (kDense) dense_count(&sh, g, ws, tile, j); else segment_prefix(&sh, g, ws, tile, j);
        exact_plane<T>(g, ws, tile, &sh);
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = sh.vecs[i];
        if (stamps) SX_STAMP(st, 9);
        const uint32_t n = kDense ? sh.dense_n : sh.seg_prefix[g.n_seg] + sh.over_n, n_raw = sh.seg_total;
        use_all = sh.use_all != 0;
        const unsigned long long n_sel = sh.n_sel;
        const unsigned long long rank = nearest_rank_index(j ? 99.0 : 1.0, n_sel);      // alpha = 1 (torch_backend.py:421-422)
        rank_other = nearest_rank_index(j ? 1.0 : 99.0, n_sel);
        bool ok = mode == 0 && (spec & (kSpecSlow | kSpecHazard)) == 0 && !use_all && !g.spec_fail && sh.seg_overflow == 0 && rank >= below && rank - below < n;
        uint32_t* spill = spill_area + ((size_t)tile * kSlots + j) * spill_words;
        uint32_t answer = 0, why = 1u;
        if (ok) {      // uniform
            // the exact keys of the candidates and their range; one thread works out the proof obligations meanwhile
            uint32_t mn = 0xFFFFFFFFu, mx = 0u;
            auto phi_key_of = [&](uint32_t i, const float (&od)[3]) {
                const uint32_t k = angle_key(od, v);
                mn = min(mn, k);
                mx = max(mx, k);
                if (i < (uint32_t)kLdsKeys) sh.keys[i] = k; else put(&spill[i - kLdsKeys], k);
            };
            if constexpr (kDense) for_each_candidate_dense(pf, n, rsrc_phi, phi_key_of); else for_each_candidate(pf, &sh, g, c0, phi_key_of);
            publish_range(&sh, mn, mx);
            if (threadIdx.x == kGroupThreads - 1) sh.ok = phi_slot_check(pr, v, j, sh.check, g.spec_rot) ? 1 : 0;
            __syncthreads();
            if (stamps) SX_STAMP(st, 10);
            ok = sh.ok != 0;
            why = 2u;
        }
        if (ok) {
            // (the histogram is NOT filled on the way: consecutive candidates are neighbours in the image with nearly the same
            // key, and a wave's LDS atomics on one bin take their turns -- 5 us against 2 us for the sweep over contiguous shares)
            answer = select_slot_keys(&sh, spill, n, (uint32_t)(rank - below), false);
            const float a = key_float(answer);
            const double slack = 4e-6;
            if (sh.check[2] == 0.0 && !((double)a >= sh.check[0] + slack)) ok = false;
            if (sh.check[3] == 0.0 && !((double)a <= sh.check[1] - slack)) ok = false;
            why = 3u;
        }
        if (!ok) {      // the speculation did not hold for this slot (or was never made): every key of the tile, exact and slow
            __syncthreads();
            if (threadIdx.x == 0) {
                atomicOr(&st.fell_back, 1u << j);
                atomicOr(&st.spec, (sh.seg_overflow ? 4u : why) << (8 + 4 * j));      // diagnostic: why (1 preconditions, 2 frame / boundaries, 3 answer outside, 4 segment overflow)
                atomicAdd(&ws.state[0].slow_slots, 1u);
            }
            reset_scratch(&sh.t);
            answer = select_whole_group<T>(images, g, tile, j, rank, v, use_all, &sh.t, keep);
        }
        if (threadIdx.x == 0) {
            put(&st.phi_key[j], answer);
            put(&st.rank[j], rank);
            put(&st.ncand_seen[j], mode == 0 ? n_raw : 0u);
            __hip_atomic_store(&st.phi_pub[j], (1ull << 32) | (unsigned long long)answer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the hand-off granule
            sh.own_key = answer;
        }
        if (stamps) SX_STAMP(st, 11);
    }
    // ------------------------------------------------ concentration j
    if (stamps) SX_STAMP(st, 12);
    const float* c1 = ws.cand_od + ((size_t)tile * kSlots + slot) * 3 * g.cap2;
    CandPrefetch<12> pf;
    if constexpr (kDense) prefetch_candidates_dense(pf, rsrc_conc); else prefetch_candidates(pf, g, c1);      // in flight while the partner finishes
    __syncthreads();                     // everyone is done with the first selection's scratch
    select_prepare(&sh);
    if constexpr (kDense) dense_count(&sh, g, ws, tile, slot); else segment_prefix(&sh, g, ws, tile, slot);      // (wave 0)
    auto vectors_and_check = [&](uint32_t partner_key) {      // one thread
        float he[6], pinv[6];
        const uint32_t key0 = j == 0 ? sh.own_key : partner_key, key1 = j == 0 ? partner_key : sh.own_key;
        stain_vectors_and_pinv(v, key0, key1, he, pinv);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            sh.pinv[i] = pinv[i];
            sh.he[i] = he[i];
        }
        double theta = 0.0;
        float hi = 0.0f;
        const bool good = mode == 0 && conc_slot_check(pr, pinv, j, theta, hi);
        sh.check[0] = theta;
        sh.ok = good ? 1 : 0;
        // The answer has to reach theta (that is the proof): the four candidates in five that lie below it -- each end test
        // lets through what the other one's threshold would have stopped -- are counted and stay out of the histograms.
        const double bound = theta + 4e-6 * fabs(theta) + 1e-7;
        uint32_t kf = float_key((float)bound);
        if ((double)key_float(kf) > bound && kf > 0u) --kf;      // (the largest float <= bound; consecutive keys are consecutive floats)
        sh.k_floor = good ? kf : 0u;
        sh.k_ceil = good ? max(float_key(hi), kf) : 0xFFFFFFFFu;
    };
    if (threadIdx.x == kWave) {                  // (wave 1, meanwhile) the partner's key
        unsigned long long granule = 0;
        for (int spin = 0; spin < (1 << 16); ++spin) {      // (~0.1 s at the very most)
            granule = __hip_atomic_load(&st.phi_pub[1 - j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((granule >> 32) == 1ull) break;
            __builtin_amdgcn_s_sleep(4);
        }
        sh.partner_ok = (granule >> 32) == 1ull ? 1 : 0;
        if (sh.partner_ok) vectors_and_check((uint32_t)granule);
    }
    __syncthreads();
    if (__builtin_expect(sh.partner_ok == 0, 0)) {      // uniform; the partner never showed up: its percentile, the slow way
        reset_scratch(&sh.t);
        const uint32_t partner_key = select_whole_group<T>(images, g, tile, 1 - j, rank_other, v, use_all, &sh.t, keep);
        __syncthreads();
        select_prepare(&sh);
        if (threadIdx.x == kWave) vectors_and_check(partner_key);
        __syncthreads();
    }
    float pinv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) pinv[i] = sh.pinv[i];
    const unsigned long long n_all = (unsigned long long)g.pixels;
    const unsigned long long k99 = nearest_rank_index(99.0, n_all);          // torch_backend.py:447-448
    const uint32_t n = kDense ? sh.dense_n : sh.seg_prefix[g.n_seg] + sh.over_n, n_raw = sh.seg_total;
    if (stamps) SX_STAMP(st, 13);
    // every pixel that is not a candidate lies below the answer (that is what gets proved): the answer's rank among the candidates
    const unsigned long long outside = n_all - (unsigned long long)n;
    bool ok = mode == 0 && (spec & (kSpecSlow | kSpecHazard)) == 0 && !g.spec_fail && sh.seg_overflow == 0 && k99 >= outside && n > 0;
    uint32_t why = ok && sh.ok == 0 ? 2u : 1u;
    ok = ok && sh.ok != 0;
    uint32_t* spill = spill_area + ((size_t)tile * kSlots + slot) * spill_words;
    uint32_t answer = 0;
    if (ok) {
        const uint32_t k_floor = sh.k_floor, k_ceil = sh.k_ceil;
        uint32_t mn = 0xFFFFFFFFu, mx = 0u, under = 0u;
        auto conc_key_of = [&](uint32_t i, const float (&od)[3]) {
            float ca, cb;
            concentration(od, pinv, ca, cb);
            const uint32_t k = float_key(j ? cb : ca);
            mn = min(mn, k);
            mx = max(mx, k);
            under += k < k_floor ? 1u : 0u;
            if (i < (uint32_t)kLdsKeys) sh.keys[i] = k; else put(&spill[i - kLdsKeys], k);
        };
        if constexpr (kDense) for_each_candidate_dense(pf, n, rsrc_conc, conc_key_of); else for_each_candidate(pf, &sh, g, c1, conc_key_of);
        publish_range(&sh, mn, mx);
        under = wave_total_u32(under);
        if (lane_id() == 0 && under) atomicAdd(&sh.n_under, under);
        __syncthreads();
        if (stamps) SX_STAMP(st, 14);
        const unsigned long long rank_in = k99 - outside;
        const uint32_t n_under = sh.n_under;
        why = 3u;
        if (rank_in < n_under) {      // the answer lies below theta: the proof has failed
            ok = false;
        } else {
            answer = select_slot_keys(&sh, spill, n, (uint32_t)(rank_in - n_under), false, nullptr, k_floor, k_ceil);
            const double a = (double)key_float(answer), theta = sh.check[0];
            if (!(a >= theta + 4e-6 * fabs(theta) + 1e-7)) ok = false;
        }
    }
    if (!ok) {
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicOr(&st.fell_back, 1u << slot);
            atomicOr(&st.spec, (sh.seg_overflow ? 4u : why) << (8 + 4 * slot));
            atomicAdd(&ws.state[0].slow_slots, 1u);
        }
        reset_scratch(&sh.t);
        answer = select_whole_group<T>(images, g, tile, slot, k99, pinv, true, &sh.t, keep);
    }
    if (threadIdx.x == 0) {
        const float mc = key_float(answer);
        put(&st.max_c[j], mc);
        put(&st.rank[slot], k99);
        put(&st.ncand_seen[slot], mode == 0 ? n_raw : 0u);
        StageRecord* rec = &st.rec[2];
        put(&rec->scale[j], target_max_conc[j] / mc);      // torch_backend.py:452
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                put(&rec->coef[i], pinv[i]);
                put(&st.pinv[i], pinv[i]);
                put(&st.he[i], sh.he[i]);
            }
        }
    }
    if (stamps) SX_STAMP(st, 15);
}

}  // namespace macenko
}  // namespace sx

// The two-pass Macenko transform with its three dependent steps in ONE launch -- included by macenko.hip after macenko_twopass.hpp.
//
// The four-launch form (prior | pass A | stage | reconstruct) makes every tile wait for the slowest one three times, and its two
// per-tile launches are pure latency: 64 + 128 workgroups on 256 CUs for 20 + 33 of the call's ~165 us.  Here the prior stays a
// launch of its own (it needs the whole chip idle for 17 us either way) and everything behind it is ONE launch of 256-thread
// workgroups that take their ROLE from a ticket (an atomic counter: HIP promises nothing about dispatch order):
//
//   tickets [0, A)          pass A work items, tile-major (pass_a_item<kFused>): moments + candidate records, published with
//                           write-through stores; the last store of a work item is followed by one count on the tile's a_done
//   tickets [A, A + 2 N)    stage jobs (tile, j): wait for a_done == work items of the tile, then angle percentile j ->
//                           hand-off with the partner job -> stain vectors -> concentration j -> scale; 256 threads, the
//                           candidates' keys in REGISTERS (48 per thread + 4096 in LDS), at the highest wave priority
//   tickets [A + 2 N, ...)  reconstruct work items, tile-major: wait for the tile's two stage jobs (s_done == 2), then the
//                           unchanged reconstruct_item
//
// A unit waits for units with LOWER tickets -- workgroups that are already running or done -- with ONE exception: stage job (tile, 0)
// also waits for the angle key of its partner (tile, 1), which holds the NEXT ticket of the same queue and may not have been handed
// out yet.  That is safe because (a) the two tickets are adjacent, so at most one such pair per queue is ever split across a
// dispatch boundary, (b) the wait is bounded (~2^21 polls) and (c) a partner that does not show up is replaced by the exact slow
// select of its percentile in the waiting job itself: no deadlock, no assumption about co-residency or dispatch order, only time
// lost (a timeout is counted in GroupState::spin_timeouts and the workgroup goes on; ADVICE r3 corrected this paragraph, which used
// to claim the lower-ticket rule without the exception).  The hardware's own dispatcher does the scheduling: the 4 x 256 CUs' worth of resident workgroups
// start as pass A; as they retire (the oldest quarter of the tiles first: s_setprio by ticket quarter on top of the CU's
// oldest-first arbitration) the freed slots take stage jobs and then reconstruct items, which sit waiting on the CU for their
// tile's stage and start the moment it publishes.  So a tile's stage runs while later tiles still stream through pass A, and
// the reconstruct pass of the early tiles runs beside the stages of the late ones.
//
// Results: every number that reaches the output comes from the same device functions, in the same order, as in the other two
// forms; the order statistics are exact elements, so how the candidates are laid out or ranked cannot change a bit
// (tests/test_fused_gpu.py holds the three forms to bitwise equality).
#pragma once

namespace sx {
namespace macenko {

constexpr int kFusedRegKeys = 32;                  // keys of a slot's candidates a thread of a stage job keeps in registers
constexpr int kFusedLdsKeys = 6144;                // ... and the workgroup in LDS behind them
constexpr int kFusedBatch = 8;                     // candidate records a thread requests at once
static_assert(kStreamThreads * kFusedRegKeys + kFusedLdsKeys == (int)kFusedCapMax, "fused candidate capacity");
constexpr uint32_t kFusedSpinMax = 1u << 21;       // polls of a bounded wait (~1 us each with the sleep: seconds)

// One lane: wait until *word >= want.  Relaxed agent-scope polls with a sleep in between (MI355X_MICROARCH.md: polling with
// acquire loads or without a pause costs the streaming workgroups around it bandwidth).
__device__ __forceinline__ bool spin_until_at_least(const uint32_t* word, uint32_t want) {
    for (uint32_t spin = 0; spin < kFusedSpinMax; ++spin) {
        if (ld_agent(word) >= want) return true;
        __builtin_amdgcn_s_sleep(8);
    }
    return false;
}

struct alignas(16) FusedStageScratch {
    uint32_t hist[2][256];                 // the selection's two histogram levels
    uint32_t list[2 * kShortList];         // keys of the picked bin
    uint32_t lds_keys[kFusedLdsKeys];      // keys beyond the registers' share
    uint32_t radix_hist[256];              // radix_select_stream's scratch (slow exact paths)
    unsigned long long radix_rank;
    uint32_t radix_digit;
    double mom[kMoments];
    double check[8];
    float vecs[6], pinv[6], he[6];
    uint32_t lo, hi, n_list, bin, rank_in_bin, result, range_first, range_last, bin_count, n_under, k_floor, k_ceil, own_key;
    uint32_t n_raw, below, ready;
    int ok, use_all, partner_ok;
    unsigned long long n_sel;
};

// candidate i of a slot <-> (thread, register u): i = 256 u + perm(thread).  Neighbours in the array are neighbours in the image
// with nearly the same key -- the same histogram bin -- and a wave's LDS atomics on one bin take their turns: adjacent lanes hold
// candidates 37 apart instead.
__device__ __forceinline__ uint32_t fused_lane_offset() { return ((uint32_t)threadIdx.x * 37u) & (uint32_t)(kStreamThreads - 1); }

struct FusedKeys {
    uint32_t r[kFusedRegKeys];
};

// f(key) for every key this thread holds: its registers' share, then its share of the LDS tail.
template <class F>
__device__ __forceinline__ void fused_each_key(const FusedKeys& keys, const FusedStageScratch* sh, uint32_t n, F f) {
    const uint32_t off = fused_lane_offset();
#pragma unroll
    for (int u = 0; u < kFusedRegKeys; ++u) {
        if ((uint32_t)(kStreamThreads * u) < n) {      // workgroup-uniform
            if ((uint32_t)(kStreamThreads * u) + off < n) f(keys.r[u]);
        }
    }
    for (uint32_t i = (uint32_t)(kStreamThreads * kFusedRegKeys) + threadIdx.x; i < n; i += kStreamThreads) f(sh->lds_keys[i - (uint32_t)(kStreamThreads * kFusedRegKeys)]);
}

// The candidates' keys: records 256 u + perm(thread), fetched eight at a time with write-through-coherent 16-byte loads (`first`:
// the batch the caller requested before it knew how many there are -- entries beyond the fill are in-bounds garbage, never
// used).  key_of(od) -> key; tracks the keys' range.
template <class KeyOf>
__device__ __forceinline__ void fused_make_keys(FusedKeys& keys, FusedStageScratch* sh, __amdgpu_buffer_rsrc_t rsrc, uint32_t n, const sx_u4 (&first)[kFusedBatch], KeyOf key_of, uint32_t& mn, uint32_t& mx) {
    const uint32_t off = fused_lane_offset();
    constexpr int kB = kFusedBatch;
    // (one batch in registers at a time: a batch is one memory round trip, and at ~28 candidates per thread two batches of sixteen
    // are two round trips -- the first requested by the caller before the plane was known -- where batches of four were seven)
#pragma unroll
    for (int b = 0; b < kFusedRegKeys / kB; ++b) {
        if ((uint32_t)(kStreamThreads * kB * b) < n) {      // uniform
            sx_u4 cur[kB];
#pragma unroll
            for (int q = 0; q < kB; ++q) cur[q] = b == 0 ? first[q] : __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((uint32_t)(kStreamThreads * (kB * b + q)) + off) * 16u), 0, 16);
#pragma unroll
            for (int q = 0; q < kB; ++q) {
                const int u = kB * b + q;
                const float od[3] = {__uint_as_float(cur[q][0]), __uint_as_float(cur[q][1]), __uint_as_float(cur[q][2])};
                const bool valid = (uint32_t)(kStreamThreads * u) + off < n;
                const uint32_t k = key_of(od, valid);
                keys.r[u] = k;
                if (valid) {
                    mn = min(mn, k);
                    mx = max(mx, k);
                }
            }
        }
    }
    for (uint32_t i = (uint32_t)(kStreamThreads * kFusedRegKeys) + threadIdx.x; i < n; i += kStreamThreads) {
        const sx_u4 rec = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(i * 16u), 0, 16);
        const float od[3] = {__uint_as_float(rec[0]), __uint_as_float(rec[1]), __uint_as_float(rec[2])};
        const uint32_t k = key_of(od, true);
        sh->lds_keys[i - (uint32_t)(kStreamThreads * kFusedRegKeys)] = k;
        mn = min(mn, k);
        mx = max(mx, k);
    }
}

__device__ __forceinline__ void fused_first_batch(sx_u4 (&first)[kFusedBatch], __amdgpu_buffer_rsrc_t rsrc) {
    const uint32_t off = fused_lane_offset();
#pragma unroll
    for (int q = 0; q < kFusedBatch; ++q) first[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((uint32_t)(kStreamThreads * q) + off) * 16u), 0, 16);
}

__device__ __forceinline__ void fused_select_prepare(FusedStageScratch* sh) {
    if (threadIdx.x == 0) {
        sh->lo = 0xFFFFFFFFu;
        sh->hi = 0u;
        sh->n_list = 0;
        sh->result = 0;
        sh->n_under = 0;
    }
    for (int i = threadIdx.x; i < 512; i += kStreamThreads) (&sh->hist[0][0])[i] = 0;
}
__device__ __forceinline__ void fused_publish_range(FusedStageScratch* sh, uint32_t mn, uint32_t mx) {
    mn = wave_min_u32(mn);
    mx = wave_max_u32(mx);
    if (lane_id() == 0 && mn != 0xFFFFFFFFu) {
        atomicMin(&sh->lo, mn);
        atomicMax(&sh->hi, mx);
    }
}

// Exact byte-wise radix select over keys handed out by `each` (the slowest of the selection's exact paths: more keys of the
// wanted histogram bin than the list holds -- heavy ties).  Keys below k_floor take no part.
template <class Each>
__device__ __forceinline__ uint32_t fused_radix_each(FusedStageScratch* sh, Each each, uint32_t rank, uint32_t k_floor) {
    uint32_t prefix = 0, mask = 0;
    __syncthreads();
    if (threadIdx.x == 0) sh->radix_rank = rank;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int t = threadIdx.x; t < 256; t += kStreamThreads) sh->radix_hist[t] = 0;
        __syncthreads();
        each([&](uint32_t k) {
            if (k >= k_floor && ((k ^ prefix) & mask) == 0) atomicAdd(&sh->radix_hist[(k >> shift) & 255u], 1u);
        });
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

// select_slot_keys (macenko_twopass.hpp) over keys handed out by `each`: exact element of 0-based rank `rank` among the keys
// >= k_floor; whole workgroup, uniform result.  fused_select_prepare(), the keys' range and a barrier come first.
template <class Each>
__device__ __forceinline__ uint32_t fused_select(FusedStageScratch* sh, Each each, uint32_t rank, uint32_t k_floor, uint32_t k_ceil) {
    const uint32_t lane = lane_id();
    const int wave = threadIdx.x / kWave;
    const uint32_t lo = max(sh->lo, k_floor), hi = max(min(sh->hi, k_ceil), lo);
    double origin = bin_origin_for(lo), scale = bin_scale_for(lo, hi);
    uint32_t k_first = k_floor, k_last = 0xFFFFFFFFu, want = rank, picked = 0;
    bool by_bin = false;
    uint32_t* hist = sh->hist[0];
    for (int level = 0; level < 2; ++level) {
        each([&](uint32_t k) {
            if (k >= k_first && k <= k_last) atomicAdd(&hist[bin_of(k, origin, scale)], 1u);
        });
        __syncthreads();
        if (wave == 0) {
            uint32_t b, rb;
            scan_pick32(hist, want, b, rb);
            if (lane == 0) {
                const uint32_t in_bin = hist[b];
                sh->bin = b;
                sh->rank_in_bin = rb;
                sh->bin_count = in_bin;
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
        if (k_first == k_last) return k_first;      // one key value fills the bin
        hist = sh->hist[1];
        origin = bin_origin_for(k_first);
        scale = bin_scale_for(k_first, k_last);
    }
    if (!by_bin && k_first == k_last) return k_first;
    uint32_t* list = sh->list;
    each([&](uint32_t k) {
        if (k >= k_first && k <= k_last && (!by_bin || bin_of(k, origin, scale) == picked)) {
            const uint32_t at = atomicAdd(&sh->n_list, 1u);
            if (at < (uint32_t)(2 * kShortList)) list[at] = k;
        }
    });
    __syncthreads();
    const uint32_t n_list = sh->n_list;
    if (__builtin_expect(n_list <= (uint32_t)kShortList, 1)) {
        rank_pick(list, n_list, want, threadIdx.x, kStreamThreads, &sh->result);
        __syncthreads();
        return sh->result;
    }
    if (n_list <= (uint32_t)(2 * kShortList)) return radix_select_stream((unsigned long long)n_list, (unsigned long long)want, [list](unsigned long long i, uint32_t& k) { k = list[i]; return true; }, sh);
    return fused_radix_each(sh, each, rank, k_floor);
}

// exact_plane (macenko_twopass.hpp) for a stage job of the fused launch: the work items' partial sums were written by other
// workgroups of THIS launch -- agent-scope loads; index order, the same doubles.
__device__ __forceinline__ void fused_exact_plane(const Geometry& g, const Workspace& ws, int tile, int j, FusedStageScratch* sh) {
    const int64_t first = (int64_t)tile * g.blocks_per_tile;
    if (threadIdx.x < kPartial) {
        double running = 0.0;
        for (int64_t b0 = 0; b0 < g.blocks_per_tile; b0 += 16) {
            double part[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) part[u] = b0 + u < g.blocks_per_tile ? ld_agent(&ws.partial[(first + b0 + u) * kPartial + threadIdx.x]) : 0.0;
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (b0 + u < g.blocks_per_tile) running += part[u];
        }
        sh->mom[threadIdx.x] = running;
    }
    if (threadIdx.x == kPartial) sh->mom[kPartial] = (double)g.pixels;
    if (threadIdx.x > kPartial && threadIdx.x < kMoments) sh->mom[threadIdx.x] = 0.0;
    __syncthreads();
    if (__builtin_expect(sh->mom[0] < 3.0, 0)) {      // uniform, rare: every work item of such a tile left its all-pixel sums
        __syncthreads();
        if (threadIdx.x < kPartial) {
            double running = 0.0;
            for (int64_t b = 0; b < g.blocks_per_tile; ++b) running += ld_agent(&ws.partial_all[(first + b) * kPartial + threadIdx.x]);
            sh->mom[kPartial + threadIdx.x] = running;
        }
        __syncthreads();
    }
    if (threadIdx.x < 2) {
        double cov[9];
        bool use_all;
        unsigned long long n_sel;
        float vecs[6];
        plane_from_moments<true>(sh->mom, true, cov, vecs, use_all, n_sel);
        if (threadIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) sh->vecs[i] = vecs[i];
            sh->use_all = use_all ? 1 : 0;
            sh->n_sel = n_sel;
            if (j == 0) {      // (for sx_macenko_tile_params: read by a later launch)
                GroupState& st = ws.state[tile];
#pragma unroll
                for (int i = 0; i < 6; ++i) {
                    put(&st.vecs[i], vecs[i]);
                    put(&st.rec[0].coef[i], vecs[i]);
                }
#pragma unroll
                for (int k = 0; k < kMoments; ++k) put(&st.mom[k], sh->mom[k]);
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

// Stage job (tile, j): estimate_stage_kernel's work for one of a tile's two slot pairs, on 256 threads.
template <typename T>
__device__ __forceinline__ void fused_stage_job(const T* __restrict__ images, const Geometry& g, const Workspace& ws, int tile, int j, const float* __restrict__ target_max_conc, FusedStageScratch* sh, uint32_t unit,
                                                uint32_t* __restrict__ keep) {
    const int slot = 2 + j;
    GroupState& st = ws.state[tile];
    const PriorRecord* pr = &ws.prior[tile];
    FusedTile* ft = &ws.ftile[tile];
    const uint32_t cap = g.fused_cap;
    const auto rsrc_phi = __builtin_amdgcn_make_buffer_rsrc(ws.cand_rec + ((size_t)tile * kSlots + j) * cap, 0, (int)(cap * 16u), 0x00020000);
    const auto rsrc_conc = __builtin_amdgcn_make_buffer_rsrc(ws.cand_rec + ((size_t)tile * kSlots + slot) * cap, 0, (int)(cap * 16u), 0x00020000);

    // ---- the tile's pass-A work items: all published?  (they hold lower tickets: running or done)
    if (threadIdx.x == 0) {
        const bool ok = spin_until_at_least(&ft->a_done, (uint32_t)g.blocks_per_tile);
        if (!ok) atomicAdd(&ws.state[0].spin_timeouts, 1u);
        sh->ready = ok ? 1u : 0u;
    }
    __syncthreads();
    if (sh->ready == 0) return;      // uniform
    SX_UNIT_STAMP(ws, unit, 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // (no instruction: keeps the loads below behind the poll)

    const uint32_t spec = get(&st.spec);      // (prior_kernel's, a launch ago)
    const int mode = get(&pr->mode);
    float v[6];
    bool use_all;
    unsigned long long rank_other;
    {   // ------------------------------------------------ angle percentile j
        sx_u4 first[kFusedBatch];
        fused_first_batch(first, rsrc_phi);
        fused_select_prepare(sh);
        if (threadIdx.x == kWave) {
            sh->n_raw = ld_agent(&ft->ncand[j]);
            sh->below = ld_agent(&st.below[j]);
        }
        fused_exact_plane(g, ws, tile, j, sh);
        SX_UNIT_STAMP(ws, unit, 4);
#pragma unroll
        for (int i = 0; i < 6; ++i) v[i] = sh->vecs[i];
        const uint32_t n_raw = sh->n_raw, n = min(n_raw, cap), below = sh->below;
        use_all = sh->use_all != 0;
        const unsigned long long n_sel = sh->n_sel;
        const unsigned long long rank = nearest_rank_index(j ? 99.0 : 1.0, n_sel);      // alpha = 1 (torch_backend.py:421-422)
        rank_other = nearest_rank_index(j ? 1.0 : 99.0, n_sel);
        const bool overflow = n_raw > cap;
        bool ok = mode == 0 && (spec & (kSpecSlow | kSpecHazard)) == 0 && !use_all && !g.spec_fail && !overflow && rank >= below && rank - below < n;
        uint32_t answer = 0, why = 1u;
        FusedKeys keys;
        if (ok) {      // uniform
            uint32_t mn = 0xFFFFFFFFu, mx = 0u;
            fused_make_keys(keys, sh, rsrc_phi, n, first, [&](const float (&od)[3], bool) { return angle_key(od, v); }, mn, mx);
            fused_publish_range(sh, mn, mx);
            if (threadIdx.x == kStreamThreads - 1) sh->ok = phi_slot_check(pr, v, j, sh->check, g.spec_rot) ? 1 : 0;
            __syncthreads();
            SX_UNIT_STAMP(ws, unit, 5);
            ok = sh->ok != 0;
            why = 2u;
        }
        if (ok) {
            answer = fused_select(sh, [&](auto f) { fused_each_key(keys, sh, n, f); }, (uint32_t)(rank - below), 0u, 0xFFFFFFFFu);
            SX_UNIT_STAMP(ws, unit, 6);
            const float a = key_float(answer);
            const double slack = 4e-6;
            if (sh->check[2] == 0.0 && !((double)a >= sh->check[0] + slack)) ok = false;
            if (sh->check[3] == 0.0 && !((double)a <= sh->check[1] - slack)) ok = false;
            why = 3u;
        }
        if (!ok) {      // the speculation did not hold for this slot (or was never made): every key of the tile, exact and slow
            __syncthreads();
            if (threadIdx.x == 0) {
                atomicOr(&st.fell_back, 1u << j);
                atomicOr(&st.spec, (overflow ? 4u : why) << (8 + 4 * j));
                atomicAdd(&ws.state[0].slow_slots, 1u);
            }
            answer = select_whole_group<T, true>(images, g, tile, j, rank, v, use_all, sh, keep);
        }
        if (threadIdx.x == 0) {
            put(&st.phi_key[j], answer);
            put(&st.rank[j], rank);
            put(&st.ncand_seen[j], mode == 0 ? n_raw : 0u);
            __hip_atomic_store(&st.phi_pub[j], (1ull << 32) | (unsigned long long)answer, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the hand-off granule
            sh->own_key = answer;
        }
    }
    // ------------------------------------------------ concentration j
    sx_u4 first[kFusedBatch];
    fused_first_batch(first, rsrc_conc);      // in flight while the partner finishes
    __syncthreads();                           // everyone is done with the first selection's scratch
    fused_select_prepare(sh);
    if (threadIdx.x == 0) sh->n_raw = ld_agent(&ft->ncand[slot]);
    auto vectors_and_check = [&](uint32_t partner_key) {      // one thread
        float he[6], pinv[6];
        const uint32_t key0 = j == 0 ? sh->own_key : partner_key, key1 = j == 0 ? partner_key : sh->own_key;
        stain_vectors_and_pinv(v, key0, key1, he, pinv);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            sh->pinv[i] = pinv[i];
            sh->he[i] = he[i];
        }
        double theta = 0.0;
        float hi = 0.0f;
        const bool good = mode == 0 && conc_slot_check(pr, pinv, j, theta, hi);
        sh->check[0] = theta;
        sh->ok = good ? 1 : 0;
        const double bound = theta + 4e-6 * fabs(theta) + 1e-7;
        uint32_t kf = float_key((float)bound);
        if ((double)key_float(kf) > bound && kf > 0u) --kf;
        sh->k_floor = good ? kf : 0u;
        sh->k_ceil = good ? max(float_key(hi), kf) : 0xFFFFFFFFu;
    };
    if (threadIdx.x == kWave) {                  // (wave 1, meanwhile) the partner's key
        unsigned long long granule = 0;
        for (uint32_t spin = 0; spin < kFusedSpinMax; ++spin) {
            granule = __hip_atomic_load(&st.phi_pub[1 - j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((granule >> 32) == 1ull) break;
            __builtin_amdgcn_s_sleep(4);
        }
        sh->partner_ok = (granule >> 32) == 1ull ? 1 : 0;
        if (sh->partner_ok) vectors_and_check((uint32_t)granule);
    }
    __syncthreads();
    if (__builtin_expect(sh->partner_ok == 0, 0)) {      // uniform; the partner never showed up: its percentile, the slow way
        const uint32_t partner_key = select_whole_group<T, true>(images, g, tile, 1 - j, rank_other, v, use_all, sh, keep);
        __syncthreads();
        fused_select_prepare(sh);
        if (threadIdx.x == kWave) vectors_and_check(partner_key);
        __syncthreads();
    }
    SX_UNIT_STAMP(ws, unit, 7);
    float pinv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) pinv[i] = sh->pinv[i];
    const unsigned long long n_all = (unsigned long long)g.pixels;
    const unsigned long long k99 = nearest_rank_index(99.0, n_all);          // torch_backend.py:447-448
    const uint32_t n_raw = sh->n_raw, n = min(n_raw, cap);
    const bool overflow = n_raw > cap;
    const unsigned long long outside = n_all - (unsigned long long)n;
    bool ok = mode == 0 && (spec & (kSpecSlow | kSpecHazard)) == 0 && !g.spec_fail && !overflow && k99 >= outside && n > 0;
    uint32_t why = ok && sh->ok == 0 ? 2u : 1u;
    ok = ok && sh->ok != 0;
    uint32_t answer = 0;
    if (ok) {
        const uint32_t k_floor = sh->k_floor, k_ceil = sh->k_ceil;
        uint32_t mn = 0xFFFFFFFFu, mx = 0u, under = 0u;
        FusedKeys keys;
        fused_make_keys(keys, sh, rsrc_conc, n, first, [&](const float (&od)[3], bool valid) {
            float ca, cb;
            concentration(od, pinv, ca, cb);
            const uint32_t k = float_key(j ? cb : ca);
            under += (valid && k < k_floor) ? 1u : 0u;
            return k;
        }, mn, mx);
        fused_publish_range(sh, mn, mx);
        under = wave_total_u32(under);
        if (lane_id() == 0 && under) atomicAdd(&sh->n_under, under);
        __syncthreads();
        const unsigned long long rank_in = k99 - outside;
        const uint32_t n_under = sh->n_under;
        why = 3u;
        if (rank_in < n_under) {      // the answer lies below theta: the proof has failed
            ok = false;
        } else {
            answer = fused_select(sh, [&](auto f) { fused_each_key(keys, sh, n, f); }, (uint32_t)(rank_in - n_under), k_floor, k_ceil);
            const double a = (double)key_float(answer), theta = sh->check[0];
            if (!(a >= theta + 4e-6 * fabs(theta) + 1e-7)) ok = false;
        }
    }
    if (!ok) {
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicOr(&st.fell_back, 1u << slot);
            atomicOr(&st.spec, (overflow ? 4u : why) << (8 + 4 * slot));
            atomicAdd(&ws.state[0].slow_slots, 1u);
        }
        answer = select_whole_group<T, true>(images, g, tile, slot, k99, pinv, true, sh, keep);
    }
    if (threadIdx.x == 0) {
        const float mc = key_float(answer);
        put(&st.max_c[j], mc);
        put(&st.rank[slot], k99);
        put(&st.ncand_seen[slot], mode == 0 ? n_raw : 0u);
        // the stage record: what the reconstruct items of THIS launch wait for -- write-through stores, drained, then the count
        StageRecord* rec = &st.rec[2];
        st_agent(&rec->scale[j], target_max_conc[j] / mc);      // torch_backend.py:452
        if (j == 0) {
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                st_agent(&rec->coef[i], pinv[i]);
                put(&st.pinv[i], pinv[i]);
                put(&st.he[i], sh->he[i]);
            }
        }
        drain_stores();
        __hip_atomic_fetch_add(&ft->s_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

struct FusedRecord {
    float rec8[8];
    uint32_t ready;
};
union FusedScratch {
    PassAScratch<kStreamThreads> pass_a;
    FusedStageScratch stage;
    FusedRecord record;
};

template <typename T, typename O, int V, bool kUnit>
__global__ __launch_bounds__(kStreamThreads, 4) void fused_kernel(const T* __restrict__ images, O* __restrict__ out, Geometry g, Workspace ws, const float* __restrict__ stain_matrix,
                                                                  const float* __restrict__ target_max_conc) {
    __shared__ FusedScratch sh;
    __shared__ uint32_t unit_word[2];
    __shared__ LevelTables<T> tb;
    // ---- which unit: tile t belongs to queue t % 8; a workgroup asks the queue of the XCD it runs on first (workgroups are dealt
    // round-robin over the XCDs, so every queue gets its share of askers; one whose own queue is exhausted -- tile counts that are
    // not a multiple of eight -- serves the next).  Queue q, in ticket order: pass-A items of its tiles, stage jobs, reconstruct items.
    const uint32_t bpt = (uint32_t)g.blocks_per_tile, n_tiles = (uint32_t)g.n_tiles;
    if (threadIdx.x == 0) {
        const uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)) & (uint32_t)(kXcds - 1);      // HW_REG_XCC_ID
        uint32_t q = xcc, ticket = 0xFFFFFFFFu;
        for (int attempt = 0; attempt < kXcds; ++attempt, q = (q + 1) & (uint32_t)(kXcds - 1)) {
            const uint32_t tiles_q = q < n_tiles ? (n_tiles - q + (uint32_t)(kXcds - 1)) / (uint32_t)kXcds : 0u, units_q = tiles_q * (2u * bpt + 2u);
            if (units_q == 0) continue;
            const uint32_t t = __hip_atomic_fetch_add(&ws.fsched->queue[q].ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (t < units_q) {
                ticket = t;
                break;
            }
        }
        unit_word[0] = ticket;
        unit_word[1] = q;
    }
    tb.fill();
    __syncthreads();
    const uint32_t ticket = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit_word[0]), queue = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit_word[1]);      // (uniform, and known to be)
    if (ticket == 0xFFFFFFFFu) return;      // (more workgroups than units: never launched that way)
    const uint32_t tiles_q = (n_tiles - queue + (uint32_t)(kXcds - 1)) / (uint32_t)kXcds, n_items = tiles_q * bpt, n_jobs = 2u * tiles_q;
    SX_UNIT_STAMP(ws, blockIdx.x, 0);
#ifdef SX_STAMPS
    if (threadIdx.x == 0) {      // what the unit was and where it ran: XCC_ID (hwreg 20) and HW_ID (hwreg 4: cu 11:8, sh 12, se 15:13)
        const uint32_t xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | ((4 - 1) << 11)), hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | ((32 - 1) << 11));
        reinterpret_cast<unsigned long long*>(ws.block_hist)[(size_t)blockIdx.x * 8 + 3] = ((unsigned long long)xcc << 60) | ((unsigned long long)queue << 56) | ((unsigned long long)ticket << 32) | hw;
    }
#endif
    if (ticket < n_items) {
        // ---- pass A: the oldest quarter of the queue's tiles first (on top of the CU's oldest-first arbitration)
        const uint32_t quarter = (ticket * 4u) / n_items, tile = (ticket / bpt) * (uint32_t)kXcds + queue;
        if (quarter == 0) __builtin_amdgcn_s_setprio(3);      // (the instruction takes an immediate)
        else if (quarter == 1) __builtin_amdgcn_s_setprio(2);
        else if (quarter == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
        pass_a_item<T, V, kStreamThreads, false, true>(images, g, ws, tile, (int)(ticket % bpt), (int64_t)tile * bpt + ticket % bpt, &sh.pass_a, tb);
        SX_UNIT_STAMP(ws, blockIdx.x, 2);
    } else if (ticket < n_items + n_jobs) {
        const uint32_t k = ticket - n_items, tile = (k >> 1) * (uint32_t)kXcds + queue;
        __builtin_amdgcn_s_setprio(3);      // a chain of short dependent phases: it must not queue behind streaming waves
        // (plane j of the tile's output, not written before the tile's reconstruct items run: key scratch of the slow exact path)
        uint32_t* keep = sizeof(O) >= 4 ? reinterpret_cast<uint32_t*>(out + ((size_t)tile * 3 + (k & 1u)) * (size_t)g.pixels) : nullptr;
        fused_stage_job<T>(images, g, ws, (int)tile, (int)(k & 1u), target_max_conc, &sh.stage, blockIdx.x, keep);
        SX_UNIT_STAMP(ws, blockIdx.x, 2);
    } else {
        const uint32_t k = ticket - n_items - n_jobs, tile = (k / bpt) * (uint32_t)kXcds + queue;
        __builtin_amdgcn_s_setprio(0);
        if (threadIdx.x == 0) {
            const bool ok = spin_until_at_least(&ws.ftile[tile].s_done, 2u);
            if (!ok) atomicAdd(&ws.state[0].spin_timeouts, 1u);
            sh.record.ready = ok ? 1u : 0u;
            const StageRecord* rec = &ws.state[tile].rec[2];
#pragma unroll
            for (int i = 0; i < 6; ++i) sh.record.rec8[i] = ld_agent(&rec->coef[i]);
            sh.record.rec8[6] = ld_agent(&rec->scale[0]);
            sh.record.rec8[7] = ld_agent(&rec->scale[1]);
        }
        __syncthreads();
        if (sh.record.ready == 0) return;      // uniform (a wait that ran out: counted, the host raises)
        SX_UNIT_STAMP(ws, blockIdx.x, 1);
        float rec8[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) rec8[i] = sh.record.rec8[i];
        reconstruct_item<T, O, V, kUnit, kStreamThreads, false>(images, out, g, ws, tile, (int)(k % bpt), stain_matrix, tb, nullptr, rec8);
        SX_UNIT_STAMP(ws, blockIdx.x, 2);
    }
}

}  // namespace macenko
}  // namespace sx

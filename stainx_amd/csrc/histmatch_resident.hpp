// Histogram matching of a batch of planar uint8 tiles in ONE launch (round 4).  The batch histogram needs every pixel before any pixel
// can be written, so the two-kernel form reads the batch twice (3 R + 3 R + 3 W bytes per pixel for the 6 algorithmic).  Here a
// workgroup keeps a part of the batch in its registers between the counting phase and the apply phase -- 1024 threads x 22 sixteen-byte
// packs = 352 KB of a CU's 512 KB register file -- and only what does not fit is read again: of the 402 MB of config 3, ~310 MB are
// read instead of 402.  Every workgroup builds the 3 x 256 look-up table itself from the pooled counts (lut_kernel's expressions and
// order: the same bits).
//
// No co-residency assumption: the batch is cut into CHUNKS (11 sweeps of 1024 packs: 176 KB); workgroup b of G owns chunks b, b + G, ...
// and CLAIMS them (one exchange per chunk on the chunk's flag word, all of a workgroup's claims in flight together) before it touches
// them.  Its first two chunks stay in registers, the others are counted straight from memory.  A workgroup that has finished its own
// looks for chunks nobody has claimed -- those of workgroups that have not started -- and takes them.  Between the phases it waits until
// as many chunks have been COUNTED as the batch has: every claimed chunk belongs to a workgroup that is running and will finish it, so
// the wait ends whether or not all workgroups of the grid are resident (one dispatched late finds its chunks taken and goes straight to
// the wait).  The apply phase hands the chunks that are not in anybody's registers out the same way (a second flag per chunk).  With
// every workgroup running -- the ordinary case -- nobody steals and the split is the static one.  The last workgroup to leave re-arms
// the flags: a completed call leaves the workspace READY.
//
// Counting: as histogram_planar_kernel (32 bank-striped copies of a 256-bin histogram in LDS, integer adds: bit-exact).  A sweep lies in
// one plane, so the channel is uniform per sweep; the LDS histogram goes to the pooled counters when the channel changes and at the
// end.  Pooled counters: kResSets sets per parity (workgroup b adds into set b % kResSets: 256 workgroups adding to one set at the same
// moment queued 17 us on its 768 words), two parities used alternately (Tables::res_parity): a call adds into one and clears the
// other -- the one the call before it used, which nobody reads any more.
#pragma once

namespace sx {
namespace histmatch {

#ifdef SX_STAMPS
#define SX_HM_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) tab->res_stamp[i] = (unsigned long long)wall_clock64(); } while (0)
#else
#define SX_HM_STAMP(i) do { } while (0)
#endif

constexpr int kResThreads = 1024;      // (sixteen waves per CU: the counting phase is paced by the LDS atomic unit and needs them in flight)
constexpr int kResChunkSweeps = 11;
constexpr int kResKeepChunks = 2;
constexpr int kResKeep = kResChunkSweeps * kResKeepChunks;      // packs a thread keeps in registers between the phases
constexpr int kResSweepPacks = kResThreads;
constexpr int kResSweepBytes = kResSweepPacks * 16;

__device__ __forceinline__ void res_count_pack(const uint4& q, uint32_t* __restrict__ mine) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) atomicAdd(&mine[((w[i] >> (8 * b)) & 0xFFu) * kCopies], 1u);
}
__device__ __forceinline__ uint4 res_apply_pack(const uint4& q, const uint8_t* __restrict__ lut) {
    const uint32_t w[4] = {q.x, q.y, q.z, q.w};
    uint32_t r[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r[i] = 0u;
#pragma unroll
        for (int b = 0; b < 4; ++b) r[i] |= (uint32_t)lut[(w[i] >> (8 * b)) & 0xFFu] << (8 * b);
    }
    return make_uint4(r[0], r[1], r[2], r[3]);
}
__device__ __forceinline__ void res_store_stream(uint8_t* __restrict__ p, const uint4& v) {
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const u4v q = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(q, reinterpret_cast<u4v*>(p));
}

constexpr int kResOwnMax = 64;      // chunks a workgroup claims per round (its own: at most kResMaxChunks / grid; then what it finds unclaimed)
struct ResShared {
    uint32_t hist[kBins][kCopies];
    float src_cdf[3][kBins], ref_cdf[3][kBins], src_term[3][kBins], ref_term[3][kBins];
    float ref_denom[3];
    uint8_t lut[3][kBins];
    uint32_t list[kResOwnMax], won[kResOwnMax], n_list;
    int channel;      // whose counts the LDS histogram holds (-1: none)
};

// the LDS copies -> the workgroup's set of pooled counters of `channel`; LDS cleared (uniform call)
__device__ __forceinline__ void res_flush(ResShared* sh, uint32_t* __restrict__ pooled) {
    __syncthreads();
    const int channel = sh->channel;
    if (channel >= 0 && threadIdx.x < kBins) {
        const int t = threadIdx.x;
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < kCopies; ++k) sum += sh->hist[t][(t + k) & (kCopies - 1)];
        if (sum) atomicAdd(&pooled[channel * kBins + t], sum);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < kBins * kCopies; i += kResThreads) (&sh->hist[0][0])[i] = 0;
    __syncthreads();
}
__device__ __forceinline__ void res_enter_channel(ResShared* sh, uint32_t* __restrict__ pooled, int c) {      // uniform
    if (sh->channel != c) {
        if (sh->channel >= 0) res_flush(sh, pooled);
        __syncthreads();
        if (threadIdx.x == 0) sh->channel = c;
        __syncthreads();
    }
}
// Claims for the workgroup: its own chunks (first == true: b, b + G, ... -- all exchanges in flight together), or up to kResOwnMax chunks
// nobody has claimed yet (a scan of the flags).  The chunks won are left in sh->list[0 .. n_list).  Uniform.
__device__ __forceinline__ void res_claim(ResShared* sh, uint32_t* __restrict__ flags, uint32_t total_chunks, bool own) {
    __syncthreads();
    if (threadIdx.x == 0) sh->n_list = 0;
    __syncthreads();
    if (own) {
        const uint32_t c = blockIdx.x + threadIdx.x * gridDim.x;
        if (threadIdx.x < kResOwnMax) sh->won[threadIdx.x] = (c < total_chunks && __hip_atomic_exchange(&flags[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) ? c : 0xFFFFFFFFu;
        __syncthreads();
        if (threadIdx.x == 0) {      // (in order: the workgroup's first chunks are the ones it keeps)
            uint32_t n = 0;
            for (int k = 0; k < kResOwnMax; ++k)
                if (sh->won[k] != 0xFFFFFFFFu) sh->list[n++] = sh->won[k];
            sh->n_list = n;
        }
    } else {
        for (uint32_t c = threadIdx.x; c < total_chunks; c += kResThreads) {
            if (__hip_atomic_load(&flags[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && sh->n_list < (uint32_t)kResOwnMax &&
                __hip_atomic_exchange(&flags[c], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                const uint32_t at = atomicAdd(&sh->n_list, 1u);
                if (at < (uint32_t)kResOwnMax) sh->list[at] = c; else __hip_atomic_store(&flags[c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (no room this round: given back)
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) sh->n_list = min(sh->n_list, (uint32_t)kResOwnMax);
    }
    __syncthreads();
}

// chunk `chunk` into registers keep[BASE ... BASE + 10] (all its loads in flight together), then counted
template <int BASE>
__device__ __forceinline__ void res_keep_chunk(const uint8_t* __restrict__ images, int64_t chunk, int64_t total_sweeps, int sweeps_per_plane, uint4 (&keep)[kResKeep], ResShared* sh,
                                               uint32_t* __restrict__ pooled, uint32_t* __restrict__ mine) {
    const int64_t s0 = chunk * kResChunkSweeps;
    const uint4* src = reinterpret_cast<const uint4*>(images + s0 * kResSweepBytes) + threadIdx.x;
#pragma unroll
    for (int k = 0; k < kResChunkSweeps; ++k) {
        if (s0 + k < total_sweeps) keep[BASE + k] = *src;
        src += kResSweepPacks;
        asm volatile("" : "+v"(src));      // (a running pointer, opaque per step: sweeps are 16 KB apart, beyond a load's immediate offset)
    }
#pragma unroll
    for (int k = 0; k < kResChunkSweeps; ++k) {
        if (s0 + k < total_sweeps) {      // (uniform)
            res_enter_channel(sh, pooled, (int)(((s0 + k) / sweeps_per_plane) % 3));
            res_count_pack(keep[BASE + k], mine);
        }
    }
}
template <int BASE>
__device__ __forceinline__ void res_apply_kept(uint8_t* __restrict__ out, int64_t chunk, int64_t total_sweeps, int sweeps_per_plane, const uint4 (&keep)[kResKeep], const ResShared* sh) {
    const int64_t s0 = chunk * kResChunkSweeps;
    uint8_t* dst = out + s0 * kResSweepBytes + (size_t)threadIdx.x * 16;
#pragma unroll
    for (int k = 0; k < kResChunkSweeps; ++k) {
        if (s0 + k < total_sweeps) res_store_stream(dst, res_apply_pack(keep[BASE + k], sh->lut[((s0 + k) / sweeps_per_plane) % 3]));
        dst += kResSweepBytes;
        asm volatile("" : "+v"(dst));
    }
}

__global__ __launch_bounds__(kResThreads) void resident_kernel(const uint8_t* __restrict__ images, uint8_t* __restrict__ out, int64_t total_sweeps, int sweeps_per_plane, Tables* __restrict__ tab,
                                                               const float* __restrict__ ref_hist, double num_pixels) {
    __shared__ ResShared sh;
    const int tid = threadIdx.x;
    SX_HM_STAMP(0);
    const uint32_t parity = tab->res_parity & 1u;
    const uint32_t total_chunks = (uint32_t)((total_sweeps + kResChunkSweeps - 1) / kResChunkSweeps);
    uint32_t* pooled = &tab->res_counts[parity][blockIdx.x % kResSets][0][0];
    for (int i = tid; i < kBins * kCopies; i += kResThreads) (&sh.hist[0][0])[i] = 0;
    {   // the other parity: what the call before this one counted in (its readers are long gone); every workgroup clears a share
        uint32_t* other = &tab->res_counts[parity ^ 1u][0][0][0];
        for (int i = blockIdx.x * kResThreads + tid; i < kResSets * 3 * kBins; i += gridDim.x * kResThreads) other[i] = 0u;
    }
    if (tid == 0) sh.channel = -1;
    __syncthreads();
    uint32_t* mine = &sh.hist[0][tid & (kCopies - 1)];

    // ---- phase 1: count.  The workgroup's own chunks -- the first two stay in registers --, then whatever nobody has claimed.
    uint4 keep[kResKeep];
    uint32_t kept0 = 0xFFFFFFFFu, kept1 = 0xFFFFFFFFu, counted = 0;
    auto count_listed = [&](uint32_t from) {      // chunks sh.list[from ...) counted straight from memory, as ONE run of sweeps (two packs ahead across chunk ends)
        const int n_sweeps = ((int)sh.n_list - (int)from) * kResChunkSweeps;
        auto sweep_of = [&](int q) { return (int64_t)sh.list[from + q / kResChunkSweeps] * kResChunkSweeps + q % kResChunkSweeps; };
        auto load = [&](int q) {
            const int64_t sw = sweep_of(q);
            return sw < total_sweeps ? *(reinterpret_cast<const uint4*>(images + sw * kResSweepBytes) + tid) : make_uint4(0, 0, 0, 0);
        };
        uint4 next = n_sweeps > 0 ? load(0) : make_uint4(0, 0, 0, 0), next2 = n_sweeps > 1 ? load(1) : next;
        for (int q = 0; q < n_sweeps; ++q) {
            const uint4 cur = next;
            next = next2;
            if (q + 2 < n_sweeps) next2 = load(q + 2);
            const int64_t sw = sweep_of(q);
            if (sw < total_sweeps) {      // (uniform)
                res_enter_channel(&sh, pooled, (int)((sw / sweeps_per_plane) % 3));
                res_count_pack(cur, mine);
            }
        }
        counted += sh.n_list - from;
    };
    res_claim(&sh, tab->res_flag, total_chunks, true);
    {
        uint32_t from = 0;
        if (sh.n_list > 0) {
            kept0 = sh.list[0];
            res_keep_chunk<0>(images, kept0, total_sweeps, sweeps_per_plane, keep, &sh, pooled, mine);
            ++counted;
            from = 1;
        }
        if (sh.n_list > 1) {
            kept1 = sh.list[1];
            res_keep_chunk<kResChunkSweeps>(images, kept1, total_sweeps, sweeps_per_plane, keep, &sh, pooled, mine);
            ++counted;
            from = 2;
        }
        if (tid == 0) {      // (kept chunks are applied from these registers: nobody else's business in the apply phase)
            if (kept0 != 0xFFFFFFFFu) __hip_atomic_store(&tab->res_flag2[kept0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (kept1 != 0xFFFFFFFFu) __hip_atomic_store(&tab->res_flag2[kept1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        count_listed(from);
    }
    for (;;) {      // chunks of workgroups that have not started (none, when the whole grid is running)
        res_claim(&sh, tab->res_flag, total_chunks, false);
        if (sh.n_list == 0) break;
        count_listed(0);
    }
    SX_HM_STAMP(1);
    res_flush(&sh, pooled);
    SX_HM_STAMP(2);
    // every counter add above is a device-scope atomic: acknowledged (vmcnt) before the workgroup reports its chunks counted
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0 && counted) __hip_atomic_fetch_add(&tab->res_done, counted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    // the reference's side of the tables does not depend on the counts: worked out while the other workgroups finish counting
    for (int i = tid; i < 3 * kBins; i += kResThreads) (&sh.ref_term[0][0])[i] = ref_hist[i];      // (raw for now)
    __syncthreads();
    if ((tid & 63) == 0 && tid / 64 < 3) sh.ref_denom[tid / 64] = torch_sum_256([&](int b) { return sh.ref_term[tid / 64][b]; }) + 1e-8f;      // reference: h / (sum(h) + 1e-8)
    __syncthreads();
    for (int i = tid; i < 3 * kBins; i += kResThreads) (&sh.ref_term[0][0])[i] = (&sh.ref_term[0][0])[i] / sh.ref_denom[i / kBins];
    __syncthreads();
    if ((tid & 63) == 0 && tid / 64 < 3) running_sum(sh.ref_term[tid / 64], sh.ref_cdf[tid / 64]);
    if (tid == 0) {      // (every chunk was taken by a workgroup that is running: the count gets there)
        unsigned int polls = 0;
        while (__hip_atomic_load(&tab->res_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < total_chunks) {
            __builtin_amdgcn_s_sleep(2);
            if (++polls > (1u << 24)) {      // (seconds: a device that stopped running a workgroup in the middle of its chunk; reported, never observed)
                atomicOr(&tab->status, 2u);
                break;
            }
        }
    }
    __syncthreads();
    SX_HM_STAMP(3);

    // ---- the look-up tables, in every workgroup (lut_kernel's expressions and order: the same bits)
    unsigned long long my_total = 0;
    for (int i = tid; i < 3 * kBins; i += kResThreads) {
        uint32_t count = 0;
#pragma unroll
        for (int s = 0; s < kResSets; ++s) count += __hip_atomic_load(&tab->res_counts[parity][s][0][0] + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        (&sh.src_term[0][0])[i] = (float)count / (float)(num_pixels + 1e-8);      // source: counts / float(num_pixels + 1e-8)
        if (i < kBins) my_total += count;
        if (blockIdx.x == 0) {
            (&tab->counted[0][0])[i] = count;
            (&tab->counts64[0][0])[i] = count;
        }
    }
    __syncthreads();
    if ((tid & 63) == 0 && tid / 64 < 3) running_sum(sh.src_term[tid / 64], sh.src_cdf[tid / 64]);      // three running sums, one lane of three different waves
    __syncthreads();
    for (int i = tid; i < 3 * kBins; i += kResThreads) {
        const int c = i / kBins, t = i % kBins;
        const float v = lut_value(sh.src_cdf[c][t], sh.ref_cdf[c]);
        sh.lut[c][t] = (uint8_t)v;
        if (blockIdx.x == 0) {
            tab->lut[c][t] = v;
            tab->typed_lut[c][t] = pack_elem<uint8_t>((uint8_t)v);
        }
    }
    if (blockIdx.x == 0) {      // the counters of a call add up to its pixels -- unless the workspace was not ready
        __shared__ unsigned long long totals[kResThreads / kWave];
        const unsigned long long w = (unsigned long long)wave_sum((double)my_total);
        if (lane_id() == 0) totals[tid / kWave] = w;
        __syncthreads();
        if (tid == 0) {
            unsigned long long total = 0;
            for (int i = 0; i < kResThreads / kWave; ++i) total += totals[i];
            if ((double)total != num_pixels) atomicOr(&tab->status, 1u);
        }
    }
    __syncthreads();
    SX_HM_STAMP(4);

    // ---- phase 2: apply.  The kept chunks from the registers, the listed ones read again (handed out by a second ticket counter).
    if (kept0 != 0xFFFFFFFFu) res_apply_kept<0>(out, kept0, total_sweeps, sweeps_per_plane, keep, &sh);
    if (kept1 != 0xFFFFFFFFu) res_apply_kept<kResChunkSweeps>(out, kept1, total_sweeps, sweeps_per_plane, keep, &sh);
    SX_HM_STAMP(5);
    auto apply_listed = [&]() {
        const int n_sweeps = (int)sh.n_list * kResChunkSweeps;
        auto sweep_of = [&](int q) { return (int64_t)sh.list[q / kResChunkSweeps] * kResChunkSweeps + q % kResChunkSweeps; };
        auto load = [&](int q) {
            const int64_t sw = sweep_of(q);
            return sw < total_sweeps ? *(reinterpret_cast<const uint4*>(images + sw * kResSweepBytes) + tid) : make_uint4(0, 0, 0, 0);
        };
        uint4 next = n_sweeps > 0 ? load(0) : make_uint4(0, 0, 0, 0), next2 = n_sweeps > 1 ? load(1) : next;
        for (int q = 0; q < n_sweeps; ++q) {
            const uint4 cur = next;
            next = next2;
            if (q + 2 < n_sweeps) next2 = load(q + 2);
            const int64_t sw = sweep_of(q);
            if (sw < total_sweeps) res_store_stream(out + sw * kResSweepBytes + (size_t)tid * 16, res_apply_pack(cur, sh.lut[(sw / sweeps_per_plane) % 3]));
        }
    };
    res_claim(&sh, tab->res_flag2, total_chunks, true);      // (the kept chunks' second flags are set: not won here)
    apply_listed();
    for (;;) {
        res_claim(&sh, tab->res_flag2, total_chunks, false);
        if (sh.n_list == 0) break;
        apply_listed();
    }
    SX_HM_STAMP(6);
    // ---- the last workgroup to leave re-arms the counters and hands the next call the other parity
    __syncthreads();
    __shared__ bool last_out;
    if (tid == 0) last_out = __hip_atomic_fetch_add(&tab->res_leave, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1u;
    __syncthreads();
    if (last_out) {      // (every workgroup of the grid has been through both phases)
        for (uint32_t c = tid; c < total_chunks; c += kResThreads) {
            __hip_atomic_store(&tab->res_flag[c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&tab->res_flag2[c], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (tid == 0) {
            __hip_atomic_store(&tab->res_done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&tab->res_leave, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&tab->res_parity, parity ^ 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace histmatch
}  // namespace sx

// Histogram matching for MI355X (gfx950): wavefront-private LDS histograms + fused LUT apply.
//
// Numerics follow HistogramMatchingTorch (rendeirolab/stainx src/stainx/backends/torch_backend.py:
// 134-301): pixels become uint8 grey levels (floats: trunc(clamp(x*255,0,255)), :115-120), one
// 256-bin histogram per channel pooled over the WHOLE batch (:229-236), source CDF vs reference CDF
// -> 256-entry float LUT with linear interpolation (:254-281), gather (:285), and the output range /
// dtype rules (:288-298).
//
// Kernels: (1) histogram -- 16-byte loads; planar layout: one plane chunk per workgroup and 32 bank-striped copies of
// the histogram in LDS (no bank conflicts); interleaved layout: one LDS sub-histogram per wave; integer adds only
// (bit-exact, order independent); (2) one small workgroup builds
// the 3 x 256 LUT (sequential double-precision prefix sums rounded to float per entry, exactly what
// torch.cumsum does on CPU floats) and the LUT in the OUTPUT element type; (3) apply -- LDS-resident
// typed LUT, 16-byte loads and stores, writes the final dtype directly (the reference's native path
// takes three more passes: histogram_matching.cu:153-166).
#include "common.hpp"

#include <algorithm>
#include <atomic>
#include <cstddef>
#include <cstdlib>
#include <type_traits>

namespace sx {
namespace histmatch {

constexpr int kBins = 256;
constexpr int kResSets = 16;            // one-launch form: sets of pooled counters per parity
constexpr int kResMaxChunks = 8192;     // ... and the most chunks (176 KB each) a batch may have to take it (1.4 GB)
constexpr int kThreads = 256;
constexpr int kWaves = kThreads / kWave;

struct alignas(256) Tables {
    uint32_t counted[3][kBins];     // pooled integer histogram of the source batch as the last call counted it (for inspection)
    float lut[3][kBins];            // float LUT (torch_backend.py:276-281)
    uint64_t typed_lut[3][kBins];   // LUT already converted to the output element (low bytes)
    unsigned long long counts64[3][kBins];   // the histogram the LUT is built from (local, or all-reduced over ranks)
    // The live counters of the histogram pass.  Every entry point that counts also CONSUMES them (the kernel that reads them writes
    // zeros back), so a workspace that was zero before a call is zero after it: the *_ready entry points rely on that and skip the
    // hipMemsetAsync launch in front of the histogram pass (5 us of a 115 us call); the plain entry points clear them first.
    uint32_t counts[3][kBins];
    uint32_t status;                // bit 0: a *_ready call found counters that do not add up to the pixels it counted (workspace not ready)
#ifdef SX_DIAG
    // the one-launch form (histmatch_resident.hpp, diagnostic build): sets of pooled counters in two parities used alternately -- a call adds
    // into parity `res_parity` and clears the other -- and its chunk flags
    uint32_t res_counts[2][kResSets][3][kBins];
    uint32_t res_parity;
    uint32_t res_done, res_leave;         // chunks counted; workgroups that have left
    uint32_t res_flag[kResMaxChunks];     // chunk claimed for counting
    uint32_t res_flag2[kResMaxChunks];    // chunk claimed for the apply phase (or kept in somebody's registers)
    unsigned long long res_stamp[8];      // wall_clock64() of workgroup 0 at the phase boundaries of the one-launch form
#endif
};

struct Layout {
    int64_t n_tiles, pixels;        // pixels per tile (H*W)
    int channels_last;
    __device__ __forceinline__ int channel_of(int64_t e) const { return channels_last ? (int)(e % 3) : (int)((e / pixels) % 3); }
    __host__ __device__ int64_t elements() const { return n_tiles * 3 * pixels; }
};

template <typename T>
__device__ __forceinline__ uint32_t grey_level(T v) {
    if constexpr (sizeof(T) == 1) {
        return (uint32_t)v;
    } else {
        float x;
        if constexpr (sizeof(T) == 8) x = (float)v; else x = Elem<T>::load(v);
        return (uint32_t)fminf(fmaxf(x * 255.0f, 0.0f), 255.0f);     // torch_backend.py:119 (trunc)
    }
}

template <typename T> struct VecOf { static constexpr int n = 16 / sizeof(T); };   // elements per 16-byte load

template <typename T, bool kVec>
__global__ __launch_bounds__(kThreads) void histogram_kernel(const T* __restrict__ images, Layout lay, uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[kWaves][3][kBins];
    for (int i = threadIdx.x; i < kWaves * 3 * kBins; i += kThreads) (&hist[0][0][0])[i] = 0;
    __syncthreads();
    uint32_t(*mine)[kBins] = hist[threadIdx.x / kWave];
    const int64_t total = lay.elements();
    constexpr int V = kVec ? VecOf<T>::n : 1;
    const int64_t stride = (int64_t)gridDim.x * kThreads * V;
    for (int64_t e = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * V; e < total; e += stride) {
        if constexpr (kVec) {
            const Pack<T, V> pk = *reinterpret_cast<const Pack<T, V>*>(images + e);
            if (lay.channels_last) {
                int c = (int)(e % 3);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    atomicAdd(&mine[c][grey_level<T>(pk.v[i])], 1u);
                    c = c == 2 ? 0 : c + 1;
                }
            } else {
                const int c = lay.channel_of(e);          // pixels % V == 0: a pack never straddles planes
#pragma unroll
                for (int i = 0; i < V; ++i) atomicAdd(&mine[c][grey_level<T>(pk.v[i])], 1u);
            }
        } else {
            atomicAdd(&mine[lay.channel_of(e)][grey_level<T>(images[e])], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * kBins; i += kThreads) {
        uint32_t s = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) s += (&hist[w][0][0])[i];
        if (s) atomicAdd(&counts[i], s);
    }
}

// Channels-last layout, 16-byte packs: the three channels alternate inside a pack, so three histograms are live: 16 copies of
// each (48 KB); word (channel*256 + bin)*16 + k lies in bank 16*(bin & 1) + k, so
// the four lanes that share a copy collide only when their bins have the same parity -- two lanes per bank on average where the
// per-wave histograms of histogram_kernel() put five.
constexpr int kLastCopies = 16;
constexpr int kLastChunk = 65536;       // elements (bytes for uint8) per workgroup; a multiple of 3 * V is not needed: the channel follows e % 3

template <typename T>
__global__ __launch_bounds__(kThreads) void histogram_last_kernel(const T* __restrict__ images, int64_t total, uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[3][kBins][kLastCopies];
    for (int i = threadIdx.x; i < 3 * kBins * kLastCopies; i += kThreads) (&hist[0][0][0])[i] = 0;
    __syncthreads();
    constexpr int V = VecOf<T>::n;
    const int64_t begin = (int64_t)blockIdx.x * kLastChunk, end = min(begin + (int64_t)kLastChunk, total);
    const int copy = threadIdx.x & (kLastCopies - 1);
    for (int64_t e = begin + (int64_t)threadIdx.x * V; e < end; e += (int64_t)kThreads * V) {
        const Pack<T, V> pk = *reinterpret_cast<const Pack<T, V>*>(images + e);
        int c = (int)(e % 3);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            atomicAdd(&hist[c][grey_level<T>(pk.v[i])][copy], 1u);
            c = c == 2 ? 0 : c + 1;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * kBins; i += kThreads) {      // thread t adds up the copies of bin t of each channel, starting at its own bank
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < kLastCopies; ++k) sum += (&hist[0][0][0])[i * kLastCopies + ((threadIdx.x + k) & (kLastCopies - 1))];
        if (sum) atomicAdd(&counts[i], sum);
    }
}

// torch.sum() of 256 contiguous float32 on the CPU (the reference's `counts.sum()` / `ref_hist.float().sum()`,
// torch_backend.py:141,222): not a plain running sum -- ATen's vectorised reduction keeps four accumulators of eight lanes
// over blocks of 32 elements, adds the accumulators in order, then the eight lanes in order.  Reproduced as is (verified
// against torch 2.10 on 300 random histograms, tools/check_torch_sum.py): a sum accumulated any other way differs in the
// last bit for ~20 % of histograms, which moves LUT entries by 1e-7 and flips a grey level at truncation boundaries.
template <class At>
__device__ inline float torch_sum_256(At at) {
    float acc[4][8];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int l = 0; l < 8; ++l) acc[k][l] = 0.0f;
#pragma unroll 1      // (unrolled eight times it needs 140 registers: too many beside the histogram pass that also calls it)
    for (int i = 0; i < kBins / 32; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int l = 0; l < 8; ++l) acc[k][l] = __fadd_rn(acc[k][l], at(i * 32 + k * 8 + l));
    float total = 0.0f;
#pragma unroll
    for (int l = 0; l < 8; ++l) {
        const float lane = __fadd_rn(__fadd_rn(__fadd_rn(acc[0][l], acc[1][l]), acc[2][l]), acc[3][l]);
        total = l == 0 ? lane : __fadd_rn(total, lane);
    }
    return total;
}

// fit: normalised histogram  counts / (sum(counts) + 1e-8)  in float32 (torch_backend.py:139-141)
__global__ void normalise_kernel(Tables* __restrict__ tab, float* __restrict__ hist_out) {
    const int c = blockIdx.x;
    __shared__ float total_s;
    __shared__ float raw[kBins];
    const uint32_t count = tab->counts[c][threadIdx.x];
    tab->counts[c][threadIdx.x] = 0;      // consumed
    tab->counted[c][threadIdx.x] = count;
    raw[threadIdx.x] = (float)count;
    __syncthreads();
    if (threadIdx.x == 0) {
        total_s = torch_sum_256([&](int b) { return raw[b]; }) + 1e-8f;
    }
    __syncthreads();
    hist_out[c * kBins + threadIdx.x] = raw[threadIdx.x] / total_s;
}

template <typename O> __device__ __forceinline__ uint64_t pack_elem(O v) {
    uint64_t bits = 0;
    __builtin_memcpy(&bits, &v, sizeof(O));
    return bits;
}

__global__ void widen_kernel(Tables* __restrict__ tab, unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * kBins) {
        const uint32_t count = (&tab->counts[0][0])[i];
        (&tab->counts[0][0])[i] = 0;      // consumed
        (&tab->counted[0][0])[i] = count;
        out[i] = count;
    }
}

// One LUT entry: where the source's running sum s of grey level t falls among the reference's running sums.  InT decides the range
// rules of the output (:288-298).
__device__ __forceinline__ float lut_value(float s, const float* ref_cdf) {
    // searchsorted(right=False): first index with ref_cdf[idx] >= s; clamp to [1,255] (:260-261)
    int lo = 0, hi = kBins;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (ref_cdf[mid] < s) lo = mid + 1; else hi = mid;
    }
    const int idx = min(max(lo, 1), kBins - 1);
    const float q_lo = ref_cdf[idx - 1], q_hi = ref_cdf[idx];
    const float diff = q_hi - q_lo;
    const float alpha = diff > 1e-10f ? (s - q_lo) / diff : 0.0f;                    // :272-273
    float v = (float)(idx - 1) + alpha * ((float)idx - (float)(idx - 1));             // :276
    if (s <= ref_cdf[0]) v = 0.0f;                                                    // :268, :279
    if (s >= ref_cdf[kBins - 1]) v = 255.0f;                                          // :269, :280
    return fminf(fmaxf(v, 0.0f), 255.0f);                                             // :281
}
template <typename T>
__device__ __forceinline__ void lut_entry(Tables* __restrict__ tab, int c, int t, float s, const float* ref_cdf) {
    const float v = lut_value(s, ref_cdf);
    tab->lut[c][t] = v;
    if constexpr (sizeof(T) == 1) {
        tab->typed_lut[c][t] = pack_elem<uint8_t>((uint8_t)v);                        // stays 0..255, truncated
    } else {
        const float unit = fminf(fmaxf(v / 255.0f, 0.0f), 1.0f);                      // :291, :296
        if constexpr (sizeof(T) == 8) tab->typed_lut[c][t] = pack_elem<double>((double)unit);
        else tab->typed_lut[c][t] = pack_elem<T>(Elem<T>::store(unit));
    }
}

// running sum in double rounded per entry (:236): the order is torch.cumsum's, so one thread per table walks it -- sixteen
// terms are fetched from LDS at a time (one LDS latency per term made the LUT kernel 11 us; the additions alone are ~1 us)
__device__ __forceinline__ void running_sum(const float* term, float* cdf) {
    double run = 0.0;
#pragma unroll 1
    for (int b0 = 0; b0 < kBins; b0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = term[b0 + u];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            run += (double)v[u];
            cdf[b0 + u] = (float)run;
        }
    }
}

// One workgroup of 256 threads per channel.
// (Round 4 tried the tables inside the APPLY launch -- every workgroup builds them from the final counters in its prologue, the workgroup
// that reads the counters last consumes them: same bits, 0.118 -> 0.142 ms per call.  Each workgroup runs six sequential chains of 256
// double-precision additions (torch.cumsum's order), and with eight workgroups per CU the chains of a CU share its four SIMDs: a 25 us
// prologue; with 1024-thread workgroups, two per CU, 0.124 ms -- still behind this 8.4 us launch of three workgroups.)
// (Tried: the LUT inside the histogram launch -- its first workgroup prepares the reference's running sums, the workgroup whose counts
// arrive last does the rest -- to save this launch and its boundary.  Same bits, and slower: 121 against 114 us per call.  Every one of
// the 3072 workgroups then waits for its counter adds to be acknowledged and for a ticket from ONE address before it may leave its
// CU: the histogram launch went from 36 to 57 us, more than this kernel's 8.8 us and the boundary together.)
template <typename T>
__global__ __launch_bounds__(kBins) void lut_kernel(Tables* __restrict__ tab, const unsigned long long* __restrict__ counts, const bool local, const float* __restrict__ ref_hist, double num_pixels) {
    const int c = blockIdx.x, t = threadIdx.x;
    __shared__ float src_cdf[kBins], ref_cdf[kBins];
    __shared__ float src_term[kBins], ref_term[kBins];
    __shared__ float ref_denom_s;
    __shared__ float ref_raw[kBins];
    // the divisions run one per thread; only the two running sums are sequential (that order is torch.cumsum's)
    ref_raw[t] = ref_hist[c * kBins + t];      // (one load per thread: the summing thread reading global memory itself paid eight round trips)
    __syncthreads();
    if (t == 64) {
        // reference: h / (sum(h) + 1e-8) (:222-223)
        ref_denom_s = torch_sum_256([&](int b) { return ref_raw[b]; }) + 1e-8f;
    }
    // source: counts / float(num_pixels + 1e-8) (:235)
    // (the local histogram is read as it was counted; counts pooled over ranks arrive widened to 64 bits)
    unsigned long long count;
    if (local) {
        const uint32_t mine = tab->counts[c][t];
        tab->counts[c][t] = 0;      // consumed: the next call's histogram pass starts from zero
        tab->counted[c][t] = mine;
        count = mine;
        // the counters of a call add up to its pixels -- unless the workspace was not ready (see Tables)
        const double wave_total = wave_sum((double)mine);      // (integers below 2^53: exact)
        __shared__ double parts[kBins / kWave];
        if (lane_id() == 0) parts[t / kWave] = wave_total;
        __syncthreads();
        if (t == 0) {
            double total = 0.0;
            for (int w = 0; w < kBins / kWave; ++w) total += parts[w];
            if (total != num_pixels) atomicOr(&tab->status, 1u);
        }
    } else {
        count = counts[c * kBins + t];
    }
    src_term[t] = (float)count / (float)(num_pixels + 1e-8);
    __syncthreads();
    ref_term[t] = ref_raw[t] / ref_denom_s;
    __syncthreads();
    if (t == 0) running_sum(src_term, src_cdf);
    else if (t == 64) running_sum(ref_term, ref_cdf);
    __syncthreads();
    lut_entry<T>(tab, c, t, src_cdf[t], ref_cdf);
}

// Planar layout, 16-byte packs: a workgroup takes one chunk of ONE channel plane, so a single 256-bin histogram is live
// and LDS has room for 32 copies of it, copy k in bank k: lane l only ever touches bank l % 32, so the 64 lanes of an
// atomic instruction never collide on a bank (random grey levels into a single histogram: ~5 of 64 lanes per bank and
// 13 cycles per instruction measured).  Integer adds: bit-exact.
// What bounds it (tools/histbench.hip, 201 MB of uint8, one MI355X): the LDS atomic unit takes one ds_add_u32 per ~6.7 cycles and
// CU whatever its lanes do (64, 32 or 16 active lanes and fixed conflict-free addresses: 39-40 us) -- 16 instructions per 16-byte
// pack, 38-39 us for the batch against 31 us for reading it.  Reaching that floor is a matter of waves per CU: 256 threads per
// 32 KB of copies left 46-49 us; 16 copies reach 39 us on noise and lose on slide background (four lanes of a wave on one
// address: 50 us); 512 threads SHARING the 32 copies, two packs in flight per thread: 38 us on noise, 36.5 us on background.
constexpr int kCopies = 32;
constexpr int kPlaneChunk = 65536;      // elements of one plane per workgroup
constexpr int kPlaneThreads = 512;
constexpr int kAhead = 2;               // packs loaded before the first of them is counted

template <typename T>
__global__ __launch_bounds__(kPlaneThreads) void histogram_planar_kernel(const T* __restrict__ images, Layout lay, int chunks_per_plane, uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[kBins][kCopies];
    for (int i = threadIdx.x; i < kBins * kCopies; i += kPlaneThreads) (&hist[0][0])[i] = 0;
    __syncthreads();
    constexpr int V = VecOf<T>::n;
    const int64_t plane = blockIdx.x / chunks_per_plane, chunk = blockIdx.x % chunks_per_plane;
    const int channel = (int)(plane % 3);
    const T* src = images + plane * lay.pixels;
    const int64_t begin = chunk * (int64_t)kPlaneChunk, end = min(begin + (int64_t)kPlaneChunk, lay.pixels);
    uint32_t* mine = &hist[0][threadIdx.x & (kCopies - 1)];
    constexpr int64_t kStride = (int64_t)kPlaneThreads * V;
    for (int64_t e = begin + (int64_t)threadIdx.x * V; e < end; e += kStride * kAhead) {
        Pack<T, V> pk[kAhead];
#pragma unroll
        for (int a = 0; a < kAhead; ++a)
            if (e + a * kStride < end) pk[a] = *reinterpret_cast<const Pack<T, V>*>(src + e + a * kStride);
#pragma unroll
        for (int a = 0; a < kAhead; ++a) {
            if (e + a * kStride < end) {
#pragma unroll
                for (int i = 0; i < V; ++i) atomicAdd(&mine[grey_level<T>(pk[a].v[i]) * kCopies], 1u);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < kBins) {   // thread t adds up the copies of bin t, starting at its own bank
        const int t = threadIdx.x;
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < kCopies; ++k) sum += hist[t][(t + k) & (kCopies - 1)];
        if (sum) atomicAdd(&counts[channel * kBins + t], sum);
    }
}

template <typename T, bool kVec>
__global__ __launch_bounds__(kThreads) void apply_kernel(const T* __restrict__ images, T* __restrict__ out, Layout lay, const Tables* __restrict__ tab) {
    __shared__ T lut[3][kBins];
    for (int i = threadIdx.x; i < 3 * kBins; i += kThreads) {
        const uint64_t bits = (&tab->typed_lut[0][0])[i];
        T v;
        __builtin_memcpy(&v, &bits, sizeof(T));
        (&lut[0][0])[i] = v;
    }
    __syncthreads();
    const int64_t total = lay.elements();
    constexpr int V = kVec ? VecOf<T>::n : 1;
    const int64_t stride = (int64_t)gridDim.x * kThreads * V;
    for (int64_t e0 = ((int64_t)blockIdx.x * kThreads + threadIdx.x) * V; e0 < total; e0 += stride) {
        // Back to front: the histogram pass read the batch front to back, so this pass starts on what the Infinity Cache got last (config 3,
        // rotating two batches, A/B on one box: 0.1213 / 0.1201 -> 0.1188 / 0.1179 ms per call).
        const int64_t e = total - V - e0;
        if constexpr (kVec) {
            const Pack<T, V> pk = *reinterpret_cast<const Pack<T, V>*>(images + e);      // (non-temporal loads here: no gain over rotating batches, 3 us lost on one buffer -- tools/ab_rotating_siblings.py)
            Pack<T, V> res;
            int c = lay.channel_of(e);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                res.v[i] = lut[c][grey_level<T>(pk.v[i])];
                if (lay.channels_last) c = c == 2 ? 0 : c + 1;
            }
            store_pack_stream<T, V>(out + e, res.v);      // non-temporal: written once, not read again by this library
        } else {
            out[e] = lut[lay.channel_of(e)][grey_level<T>(images[e])];
        }
    }
}

#ifdef SX_DIAG
}  // namespace histmatch
}  // namespace sx
#include "histmatch_resident.hpp"
namespace sx {
namespace histmatch {
#endif

static size_t workspace_bytes() { return sizeof(Tables); }

#ifdef SX_DIAG
// The one-launch form (histmatch_resident.hpp): planar uint8 batches of at least 32 MB whose planes are whole sweeps.  One workgroup per CU
// (an ordinary launch: the kernel does not depend on its workgroups being resident together).
static int device_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = -1;
        cached[dev] = n;
    }
    return cached[dev] > 0 ? cached[dev] : 0;
}
static bool resident_shape(int64_t n, int64_t pixels) {
    const int64_t sweeps = pixels % kResSweepBytes == 0 ? n * 3 * (pixels / kResSweepBytes) : 0;
    return sweeps >= 2048 && (sweeps + kResChunkSweeps - 1) / kResChunkSweeps <= kResMaxChunks;      // (at least 32 MB: smaller batches are bound by latency either way)
}
static bool run_resident(const uint8_t* images, uint8_t* out, int64_t n, int64_t pixels, Tables* tab, const float* ref_hist, hipStream_t stream) {
    const int cus = device_cus();
    if (cus <= 0) return false;
    const int64_t total_sweeps = n * 3 * (pixels / kResSweepBytes);
    const int sweeps_per_plane = (int)(pixels / kResSweepBytes);
    const int64_t chunks = (total_sweeps + kResChunkSweeps - 1) / kResChunkSweeps;
    const unsigned grid = (unsigned)std::min<int64_t>(cus, chunks);
    hipLaunchKernelGGL(resident_kernel, dim3(grid), dim3(kResThreads), 0, stream, images, out, total_sweeps, sweeps_per_plane, tab, ref_hist, (double)(n * pixels));
    return true;
}
#endif

template <typename T>
static int run(const void* images, void* out, int64_t n, int64_t h, int64_t w, int channels_last, const float* ref_hist, float* hist_out, unsigned long long* counts_out, const unsigned long long* counts_in, double n_total, void* ws, hipStream_t stream, bool ready) {
    Layout lay{n, h * w, channels_last};
    Tables* tab = static_cast<Tables*>(ws);
    const T* in = static_cast<const T*>(images);
    constexpr int V = VecOf<T>::n;
    const int64_t total = lay.elements();
    // vector path: 16-byte aligned base, and a pack never crosses a channel plane
    const bool vec = (reinterpret_cast<uintptr_t>(images) % 16 == 0) && (!out || reinterpret_cast<uintptr_t>(out) % 16 == 0) &&
                     (channels_last ? (total % V == 0) : (lay.pixels % V == 0));
    const int64_t per_block = (int64_t)kThreads * (vec ? V : 1) * 4;
    const unsigned grid = (unsigned)std::min<int64_t>((total + per_block - 1) / per_block, 256 * 8);
    double lut_pixels = n_total;
    if (!counts_in) {
        if (!ready && hipMemsetAsync(tab->counts, 0, sizeof(Tables) - offsetof(Tables, counts), stream) != hipSuccess) return fail(SX_ERR_LAUNCH, "hipMemsetAsync failed");
#ifdef SX_DIAG
        if constexpr (std::is_same<T, uint8_t>::value) {
            static const int mode = [] { const char* e = std::getenv("SX_HM_RESIDENT"); return e ? std::atoi(e) : 1; }();
            if (mode != 0 && vec && !channels_last && out && !hist_out && !counts_out && resident_shape(n, lay.pixels)) {
                if (run_resident(in, static_cast<uint8_t*>(out), n, lay.pixels, tab, ref_hist, stream)) return check_launch("histogram transform (one launch)");
            }
        }
#endif
        if (vec && !channels_last) {
            const int chunks_per_plane = (int)((lay.pixels + kPlaneChunk - 1) / kPlaneChunk);
            hipLaunchKernelGGL((histogram_planar_kernel<T>), dim3((unsigned)(n * 3 * chunks_per_plane)), dim3(kPlaneThreads), 0, stream, in, lay, chunks_per_plane, &tab->counts[0][0]);
        } else if (vec && channels_last) {
            hipLaunchKernelGGL((histogram_last_kernel<T>), dim3((unsigned)((total + kLastChunk - 1) / kLastChunk)), dim3(kThreads), 0, stream, in, total, &tab->counts[0][0]);
        } else if (vec)
            hipLaunchKernelGGL((histogram_kernel<T, true>), dim3(grid), dim3(kThreads), 0, stream, in, lay, &tab->counts[0][0]);
        else
            hipLaunchKernelGGL((histogram_kernel<T, false>), dim3(grid), dim3(kThreads), 0, stream, in, lay, &tab->counts[0][0]);
        if (hist_out) {
            hipLaunchKernelGGL(normalise_kernel, dim3(3), dim3(kBins), 0, stream, tab, hist_out);
            return check_launch("histogram fit");
        }
        if (counts_out) {
            hipLaunchKernelGGL(widen_kernel, dim3(3), dim3(kBins), 0, stream, tab, counts_out);
            return check_launch("histogram counts");
        }
        lut_pixels = (double)(n * h * w);
    }
    hipLaunchKernelGGL((lut_kernel<T>), dim3(3), dim3(kBins), 0, stream, tab, counts_in, counts_in == nullptr, ref_hist, lut_pixels);
    if (vec)
        hipLaunchKernelGGL((apply_kernel<T, true>), dim3(grid), dim3(kThreads), 0, stream, in, static_cast<T*>(out), lay, tab);
    else
        hipLaunchKernelGGL((apply_kernel<T, false>), dim3(grid), dim3(kThreads), 0, stream, in, static_cast<T*>(out), lay, tab);
    return check_launch("histogram transform");
}

static int dispatch(const void* images, void* out, int dtype, int64_t n, int64_t h, int64_t w, int channels_last, const float* ref_hist, float* hist_out, unsigned long long* counts_out, const unsigned long long* counts_in, double n_total, void* ws, size_t ws_bytes, void* stream_ptr, bool ready = false) {
    if (!images) return fail(SX_ERR_BAD_ARG, "images pointer is null");
    if (n <= 0 || h <= 0 || w <= 0) return fail(SX_ERR_BAD_ARG, "images must have positive sizes, got N=%lld H=%lld W=%lld", (long long)n, (long long)h, (long long)w);
    if (!ws || ws_bytes < workspace_bytes()) return fail(SX_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", workspace_bytes(), ws_bytes);
    if (reinterpret_cast<uintptr_t>(ws) % 256 != 0) return fail(SX_ERR_WORKSPACE, "workspace must be 256-byte aligned");
    hipStream_t stream = static_cast<hipStream_t>(stream_ptr);
    switch (dtype) {
        case SX_U8: return run<uint8_t>(images, out, n, h, w, channels_last, ref_hist, hist_out, counts_out, counts_in, n_total, ws, stream, ready);
        case SX_F16: return run<__half>(images, out, n, h, w, channels_last, ref_hist, hist_out, counts_out, counts_in, n_total, ws, stream, ready);
        case SX_BF16: return run<__hip_bfloat16>(images, out, n, h, w, channels_last, ref_hist, hist_out, counts_out, counts_in, n_total, ws, stream, ready);
        case SX_F32: return run<float>(images, out, n, h, w, channels_last, ref_hist, hist_out, counts_out, counts_in, n_total, ws, stream, ready);
        case SX_F64: return run<double>(images, out, n, h, w, channels_last, ref_hist, hist_out, counts_out, counts_in, n_total, ws, stream, ready);
        default: return fail(SX_ERR_DTYPE, "unsupported dtype code %d", dtype);
    }
}

}  // namespace histmatch
}  // namespace sx

using namespace sx;

extern "C" size_t sx_hm_workspace_bytes(int64_t n, int64_t h, int64_t w) {
    if (n <= 0 || h <= 0 || w <= 0) return 0;
    return histmatch::workspace_bytes();
}

extern "C" int sx_hm_fit(const void* images, int dtype, int64_t n, int64_t h, int64_t w, int channels_last, float* hist_out, void* ws, size_t ws_bytes, void* stream) {
    if (!hist_out) return fail(SX_ERR_BAD_ARG, "hist_out pointer is null");
    return histmatch::dispatch(images, nullptr, dtype, n, h, w, channels_last, nullptr, hist_out, nullptr, nullptr, 0.0, ws, ws_bytes, stream);
}

extern "C" int sx_hm_transform(const void* images, void* out, int dtype, int64_t n, int64_t h, int64_t w, int channels_last, const float* ref_hist, void* ws, size_t ws_bytes, void* stream) {
    if (!out || !ref_hist) return fail(SX_ERR_BAD_ARG, "out / ref_hist pointer is null");
    return histmatch::dispatch(images, out, dtype, n, h, w, channels_last, ref_hist, nullptr, nullptr, nullptr, 0.0, ws, ws_bytes, stream);
}

// The same calls on a workspace in the READY state -- zero-filled by sx_hm_workspace_init() or left behind by any completed call of
// this section on it: no clearing launch in front of the histogram pass.  A workspace that was not ready is noticed (the counters do
// not add up to the pixels counted) and reported by sx_hm_workspace_status(); the result of that call is not to be used.
extern "C" int sx_hm_workspace_init(void* ws, size_t ws_bytes, void* stream) {
    if (!ws || ws_bytes < histmatch::workspace_bytes()) return fail(SX_ERR_WORKSPACE, "workspace too small: need %zu bytes, got %zu", histmatch::workspace_bytes(), ws_bytes);
    if (hipMemsetAsync(ws, 0, histmatch::workspace_bytes(), static_cast<hipStream_t>(stream)) != hipSuccess) return fail(SX_ERR_LAUNCH, "hipMemsetAsync failed");
    return SX_OK;
}

extern "C" size_t sx_hm_workspace_status_offset(void) { return offsetof(histmatch::Tables, status); }
#ifdef SX_DIAG
// (tests and tools: the word every call in the one-launch form toggles; its phase stamps)
extern "C" size_t sx_hm_workspace_parity_offset(void) { return offsetof(histmatch::Tables, res_parity); }
extern "C" size_t sx_debug_hm_stamp_offset(void) { return offsetof(histmatch::Tables, res_stamp); }
#endif

extern "C" int sx_hm_fit_ready(const void* images, int dtype, int64_t n, int64_t h, int64_t w, int channels_last, float* hist_out, void* ws, size_t ws_bytes, void* stream) {
    if (!hist_out) return fail(SX_ERR_BAD_ARG, "hist_out pointer is null");
    return histmatch::dispatch(images, nullptr, dtype, n, h, w, channels_last, nullptr, hist_out, nullptr, nullptr, 0.0, ws, ws_bytes, stream, true);
}

extern "C" int sx_hm_transform_ready(const void* images, void* out, int dtype, int64_t n, int64_t h, int64_t w, int channels_last, const float* ref_hist, void* ws, size_t ws_bytes, void* stream) {
    if (!out || !ref_hist) return fail(SX_ERR_BAD_ARG, "out / ref_hist pointer is null");
    return histmatch::dispatch(images, out, dtype, n, h, w, channels_last, ref_hist, nullptr, nullptr, nullptr, 0.0, ws, ws_bytes, stream, true);
}

extern "C" int sx_hm_counts_ready(const void* images, int dtype, int64_t n, int64_t h, int64_t w, int channels_last, unsigned long long* counts_out, void* ws, size_t ws_bytes, void* stream) {
    if (!counts_out) return fail(SX_ERR_BAD_ARG, "counts_out pointer is null");
    return histmatch::dispatch(images, nullptr, dtype, n, h, w, channels_last, nullptr, nullptr, counts_out, nullptr, 0.0, ws, ws_bytes, stream, true);
}

// Source histogram pooled ACROSS RANKS: local integer counts out (3 x 256 u64), all-reduce on the host side, apply with the global counts.
extern "C" int sx_hm_counts(const void* images, int dtype, int64_t n, int64_t h, int64_t w, int channels_last, unsigned long long* counts_out, void* ws, size_t ws_bytes, void* stream) {
    if (!counts_out) return fail(SX_ERR_BAD_ARG, "counts_out pointer is null");
    return histmatch::dispatch(images, nullptr, dtype, n, h, w, channels_last, nullptr, nullptr, counts_out, nullptr, 0.0, ws, ws_bytes, stream);
}

extern "C" int sx_hm_apply(const void* images, void* out, int dtype, int64_t n, int64_t h, int64_t w, int channels_last, const unsigned long long* counts, double n_total_pixels, const float* ref_hist, void* ws, size_t ws_bytes, void* stream) {
    if (!out || !counts || !ref_hist) return fail(SX_ERR_BAD_ARG, "out / counts / ref_hist pointer is null");
    if (!(n_total_pixels >= 1.0)) return fail(SX_ERR_BAD_ARG, "n_total_pixels must be >= 1");
    return histmatch::dispatch(images, out, dtype, n, h, w, channels_last, ref_hist, nullptr, nullptr, counts, n_total_pixels, ws, ws_bytes, stream);
}

// What bounds a 256-bin histogram of bytes on one MI355X: candidate inner loops over the SAME 201 MB uint8 buffer (64 x 3 x 1024 x 1024,
// uniform noise and a smooth image), one plane chunk of 65536 bytes per workgroup as histogram_planar_kernel has it.
//   read     : 16-byte loads and a byte sum, no LDS                                  (the floor of the loop shape)
//   atomic32 : ds_add_u32, 32 copies, copy = lane % 32                               (the library's kernel)
//   atomic64c: ds_add_u32, 64 copies of 16-bit counters packed two to a word, copy = lane (private: no same-address collisions)
//   rmw8     : private byte counters, ds_read_u8 / add / ds_write_b8 one pixel after the other (no atomics)
//   pair     : atomic32 with equal neighbours inside a lane's pack merged into one add
//   half     : atomic32 on every second byte only (compiles to byte loads: not a valid comparison)
//   ahead N  : atomic32 with N packs in flight per thread before the first is counted
// Build: hipcc --offload-arch=gfx950 -O3 tools/histbench.hip -o tools/histbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

constexpr int kThreads = 256, kBins = 256, kChunk = 65536;
struct alignas(16) Pack { uint8_t v[16]; };

__global__ __launch_bounds__(kThreads) void read_kernel(const uint8_t* __restrict__ src, uint32_t* __restrict__ counts) {
    const uint8_t* p = src + (size_t)blockIdx.x * kChunk;
    uint32_t acc = 0;
    for (int e = threadIdx.x * 16; e < kChunk; e += kThreads * 16) {
        const Pack pk = *reinterpret_cast<const Pack*>(p + e);
#pragma unroll
        for (int i = 0; i < 16; ++i) acc += pk.v[i];
    }
    if (acc == 0x12345678u) counts[0] = acc;
}

template <int kStep>
__global__ __launch_bounds__(kThreads) void atomic32_kernel(const uint8_t* __restrict__ src, uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[kBins][32];
    for (int i = threadIdx.x; i < kBins * 32; i += kThreads) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint8_t* p = src + (size_t)blockIdx.x * kChunk;
    uint32_t* mine = &hist[0][threadIdx.x & 31];
    for (int e = threadIdx.x * 16; e < kChunk; e += kThreads * 16) {
        const Pack pk = *reinterpret_cast<const Pack*>(p + e);
#pragma unroll
        for (int i = 0; i < 16; i += kStep) atomicAdd(&mine[pk.v[i] * 32], 1u);
    }
    __syncthreads();
    const int t = threadIdx.x;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) sum += hist[t][(t + k) & 31];
    if (sum) atomicAdd(&counts[(blockIdx.x / 16 % 3) * kBins + t], sum);
}

// atomic32 with kAhead packs loaded before the first of them is counted (the plain loop waits for every 16-byte load in turn:
// 16 dependent round trips to memory per workgroup)
template <int kAhead>
__global__ __launch_bounds__(kThreads) void ahead_kernel(const uint8_t* __restrict__ src, uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[kBins][32];
    for (int i = threadIdx.x; i < kBins * 32; i += kThreads) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint8_t* p = src + (size_t)blockIdx.x * kChunk;
    uint32_t* mine = &hist[0][threadIdx.x & 31];
    for (int e = threadIdx.x * 16; e < kChunk; e += kThreads * 16 * kAhead) {
        Pack pk[kAhead];
#pragma unroll
        for (int a = 0; a < kAhead; ++a) pk[a] = *reinterpret_cast<const Pack*>(p + e + a * kThreads * 16);
#pragma unroll
        for (int a = 0; a < kAhead; ++a) {
#pragma unroll
            for (int i = 0; i < 16; ++i) atomicAdd(&mine[pk[a].v[i] * 32], 1u);
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) sum += hist[t][(t + k) & 31];
    if (sum) atomicAdd(&counts[(blockIdx.x / 16 % 3) * kBins + t], sum);
}

// the atomic unit alone: the same number of ds_add_u32 as atomic32, addresses fixed per lane (bank = lane % 32, bin = position in the
// pack), data still loaded and folded into the added value so the loads stay
template <int kLanes>
__global__ __launch_bounds__(kThreads) void peak_kernel(const uint8_t* __restrict__ src, uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[kBins][32];
    for (int i = threadIdx.x; i < kBins * 32; i += kThreads) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint8_t* p = src + (size_t)blockIdx.x * kChunk;
    uint32_t* mine = &hist[0][threadIdx.x & 31];
    for (int e = threadIdx.x * 16; e < kChunk; e += kThreads * 16) {
        const Pack pk = *reinterpret_cast<const Pack*>(p + e);
        if (kLanes == 64 || (threadIdx.x & 63) < kLanes) {
#pragma unroll
            for (int i = 0; i < 16; ++i) atomicAdd(&mine[(i * 16 + (threadIdx.x >> 6)) * 32], (uint32_t)pk.v[i]);
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) sum += hist[t][(t + k) & 31];
    if (sum == 0x12345678u) atomicAdd(&counts[t], sum);
}

// atomic32 with fewer copies (kCopies lanes-groups share a bank column; less LDS, more workgroups per CU) and kAhead packs in flight
template <int kCopies, int kAhead = 1>
__global__ __launch_bounds__(kThreads) void copies_kernel(const uint8_t* __restrict__ src, uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[kBins][kCopies];
    for (int i = threadIdx.x; i < kBins * kCopies; i += kThreads) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint8_t* p = src + (size_t)blockIdx.x * kChunk;
    uint32_t* mine = &hist[0][threadIdx.x & (kCopies - 1)];
    for (int e = threadIdx.x * 16; e < kChunk; e += kThreads * 16 * kAhead) {
        Pack pk[kAhead];
#pragma unroll
        for (int a = 0; a < kAhead; ++a) pk[a] = *reinterpret_cast<const Pack*>(p + e + a * kThreads * 16);
#pragma unroll
        for (int a = 0; a < kAhead; ++a) {
#pragma unroll
            for (int i = 0; i < 16; ++i) atomicAdd(&mine[pk[a].v[i] * kCopies], 1u);
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < kCopies; ++k) sum += hist[t][(t + k) & (kCopies - 1)];
    if (sum) atomicAdd(&counts[(blockIdx.x / 16 % 3) * kBins + t], sum);
}

// 32 copies shared by MORE waves: kT threads per workgroup (LDS per wave falls, workgroups per CU stay), kAhead packs in flight
template <int kT, int kAhead>
__global__ __launch_bounds__(kT) void wide_kernel(const uint8_t* __restrict__ src, uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[kBins][32];
    for (int i = threadIdx.x; i < kBins * 32; i += kT) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint8_t* p = src + (size_t)blockIdx.x * kChunk;
    uint32_t* mine = &hist[0][threadIdx.x & 31];
    for (int e = threadIdx.x * 16; e < kChunk; e += kT * 16 * kAhead) {
        Pack pk[kAhead];
#pragma unroll
        for (int a = 0; a < kAhead; ++a) pk[a] = *reinterpret_cast<const Pack*>(p + e + a * kT * 16);
#pragma unroll
        for (int a = 0; a < kAhead; ++a) {
#pragma unroll
            for (int i = 0; i < 16; ++i) atomicAdd(&mine[pk[a].v[i] * 32], 1u);
        }
    }
    __syncthreads();
    for (int t = threadIdx.x; t < kBins; t += kT) {
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < 32; ++k) sum += hist[t][(t + k) & 31];
        if (sum) atomicAdd(&counts[(blockIdx.x / 16 % 3) * kBins + t], sum);
    }
}

__global__ __launch_bounds__(kThreads) void atomic64c_kernel(const uint8_t* __restrict__ src, uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[kBins / 2][64];      // word (bin >> 1, copy): two 16-bit counters; 256 threads share 64 copies: <= 4 x 256 = 1024 per counter and chunk
    for (int i = threadIdx.x; i < kBins / 2 * 64; i += kThreads) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint8_t* p = src + (size_t)blockIdx.x * kChunk;
    uint32_t* mine = &hist[0][threadIdx.x & 63];
    for (int e = threadIdx.x * 16; e < kChunk; e += kThreads * 16) {
        const Pack pk = *reinterpret_cast<const Pack*>(p + e);
#pragma unroll
        for (int i = 0; i < 16; ++i) atomicAdd(&mine[(pk.v[i] >> 1) * 64], 1u << (16 * (pk.v[i] & 1)));
    }
    __syncthreads();
    const int t = threadIdx.x;
    uint32_t sum = 0;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) sum += (hist[t >> 1][(t + k) & 63] >> (16 * (t & 1))) & 0xffffu;
    if (sum) atomicAdd(&counts[(blockIdx.x / 16 % 3) * kBins + t], sum);
}

__global__ __launch_bounds__(kThreads) void rmw8_kernel(const uint8_t* __restrict__ src, uint32_t* __restrict__ counts) {
    __shared__ uint8_t hist[kBins / 4][kThreads][4];      // byte counter of (bin, thread) in word (bin >> 2, thread): bank = thread % 32
    for (int i = threadIdx.x; i < kBins / 4 * kThreads; i += kThreads) reinterpret_cast<uint32_t*>(&hist[0][0][0])[i] = 0;
    __syncthreads();
    const uint8_t* p = src + (size_t)blockIdx.x * kChunk;
    // 65536 / 256 = 256 bytes per thread: a counter could wrap at 256 -- the benchmark takes 240 of them (15 packs) to stay exact
    for (int e = threadIdx.x * 16, it = 0; it < 15; e += kThreads * 16, ++it) {
        const Pack pk = *reinterpret_cast<const Pack*>(p + e);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            volatile uint8_t* c = &hist[pk.v[i] >> 2][threadIdx.x][pk.v[i] & 3];
            *c = *c + 1;
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    uint32_t sum = 0;
    for (int k = 0; k < kThreads; ++k) sum += hist[t >> 2][(t + k) & (kThreads - 1)][t & 3];
    if (sum) atomicAdd(&counts[(blockIdx.x / 16 % 3) * kBins + t], sum);
}

__global__ __launch_bounds__(kThreads) void pair_kernel(const uint8_t* __restrict__ src, uint32_t* __restrict__ counts) {
    __shared__ uint32_t hist[kBins][32];
    for (int i = threadIdx.x; i < kBins * 32; i += kThreads) (&hist[0][0])[i] = 0;
    __syncthreads();
    const uint8_t* p = src + (size_t)blockIdx.x * kChunk;
    uint32_t* mine = &hist[0][threadIdx.x & 31];
    for (int e = threadIdx.x * 16; e < kChunk; e += kThreads * 16) {
        const Pack pk = *reinterpret_cast<const Pack*>(p + e);
        uint32_t run = 1;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const bool last = i == 15 || pk.v[i + 1 < 16 ? i + 1 : 15] != pk.v[i];
            if (last) { atomicAdd(&mine[pk.v[i] * 32], run); run = 1; } else ++run;
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    uint32_t sum = 0;
#pragma unroll
    for (int k = 0; k < 32; ++k) sum += hist[t][(t + k) & 31];
    if (sum) atomicAdd(&counts[(blockIdx.x / 16 % 3) * kBins + t], sum);
}

int main() {
    const size_t total = (size_t)64 * 3 * 1024 * 1024;
    uint8_t* src;
    uint32_t* counts;
    CK(hipMalloc(&src, total));
    CK(hipMalloc(&counts, 3 * kBins * 4));
    std::vector<uint8_t> host(total);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)(total / kChunk);
    for (int image = 0; image < 3; ++image) {
        uint32_t s = 12345;
        for (size_t i = 0; i < total; ++i) {
            s = s * 1664525u + 1013904223u;
            if (image == 0) host[i] = (uint8_t)(s >> 24);
            else if (image == 1) host[i] = (uint8_t)(128 + 60 * __builtin_sinf((float)(i % 1024) * 0.01f) + ((s >> 28) & 3));      // smooth rows, two bits of noise
            else host[i] = (uint8_t)(244 + ((s >> 30) & 1));      // slide background: two grey levels
        }
        CK(hipMemcpy(src, host.data(), total, hipMemcpyHostToDevice));
        std::vector<uint32_t> want(3 * kBins, 0), got(3 * kBins);
        for (size_t i = 0; i < total; ++i) want[(i / kChunk / 16 % 3) * kBins + host[i]]++;
        printf("== %s\n", image == 0 ? "uniform noise" : image == 1 ? "smooth rows + 2 bits of noise" : "background: two grey levels");
        struct V { const char* name; int which; bool exact; } vs[] = {{"read", 0, false}, {"atomic32", 1, true}, {"atomic64c", 2, true}, {"rmw8 (15/16 of the bytes)", 3, false}, {"pair", 4, true}, {"half", 5, false}, {"ahead 2", 6, true}, {"ahead 4", 7, true}, {"ahead 8", 8, true}, {"ahead 16", 9, true}, {"peak (fixed addresses)", 10, false}, {"peak, 32 of 64 lanes", 11, false}, {"peak, 16 of 64 lanes", 12, false}, {"64 copies", 13, true}, {"16 copies", 14, true}, {"8 copies", 15, true}, {"16 copies, ahead 2", 16, true}, {"16 copies, ahead 4", 17, true}, {"8 copies, ahead 4", 18, true}, {"4 copies", 19, true}, {"512 threads, ahead 1", 20, true}, {"512 threads, ahead 2", 21, true}, {"1024 threads, ahead 1", 22, true}, {"1024 threads, ahead 2", 23, true}};
        for (const V& v : vs) {
            float best = 1e9f, ms;
            for (int rep = 0; rep < 5; ++rep) {
                CK(hipMemset(counts, 0, 3 * kBins * 4));
                CK(hipEventRecord(e0));
                switch (v.which) {
                    case 0: read_kernel<<<grid, kThreads>>>(src, counts); break;
                    case 1: atomic32_kernel<1><<<grid, kThreads>>>(src, counts); break;
                    case 2: atomic64c_kernel<<<grid, kThreads>>>(src, counts); break;
                    case 3: rmw8_kernel<<<grid, kThreads>>>(src, counts); break;
                    case 4: pair_kernel<<<grid, kThreads>>>(src, counts); break;
                    case 5: atomic32_kernel<2><<<grid, kThreads>>>(src, counts); break;
                    case 6: ahead_kernel<2><<<grid, kThreads>>>(src, counts); break;
                    case 7: ahead_kernel<4><<<grid, kThreads>>>(src, counts); break;
                    case 8: ahead_kernel<8><<<grid, kThreads>>>(src, counts); break;
                    case 9: ahead_kernel<16><<<grid, kThreads>>>(src, counts); break;
                    case 10: peak_kernel<64><<<grid, kThreads>>>(src, counts); break;
                    case 11: peak_kernel<32><<<grid, kThreads>>>(src, counts); break;
                    case 12: peak_kernel<16><<<grid, kThreads>>>(src, counts); break;
                    case 13: copies_kernel<64><<<grid, kThreads>>>(src, counts); break;
                    case 14: copies_kernel<16><<<grid, kThreads>>>(src, counts); break;
                    case 15: copies_kernel<8><<<grid, kThreads>>>(src, counts); break;
                    case 16: copies_kernel<16, 2><<<grid, kThreads>>>(src, counts); break;
                    case 17: copies_kernel<16, 4><<<grid, kThreads>>>(src, counts); break;
                    case 18: copies_kernel<8, 4><<<grid, kThreads>>>(src, counts); break;
                    case 19: copies_kernel<4><<<grid, kThreads>>>(src, counts); break;
                    case 20: wide_kernel<512, 1><<<grid, 512>>>(src, counts); break;
                    case 21: wide_kernel<512, 2><<<grid, 512>>>(src, counts); break;
                    case 22: wide_kernel<1024, 1><<<grid, 1024>>>(src, counts); break;
                    case 23: wide_kernel<1024, 2><<<grid, 1024>>>(src, counts); break;
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < best) best = ms;
            }
            CK(hipMemcpy(got.data(), counts, 3 * kBins * 4, hipMemcpyDeviceToHost));
            bool same = true;
            for (int i = 0; i < 3 * kBins; ++i) same = same && got[i] == want[i];
            printf("%-28s %7.1f us   %s\n", v.name, best * 1e3f, v.exact ? (same ? "exact" : "WRONG") : "-");
        }
    }
    return 0;
}

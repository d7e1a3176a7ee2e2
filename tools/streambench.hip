// How much per-pixel arithmetic a read-only pass over 64x3x512x512 fp32 tiles can carry on one MI355X before it
// stops being memory-bound: same access shape as the Macenko streaming stages (one 256-thread workgroup per 8192
// pixels, three planes, 16 B per lane and plane), a tunable number of v_log_f32 and fma per pixel, with and without
// an explicit one-pack-ahead prefetch, at two occupancies.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/streambench.hip -o tools/streambench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#ifndef SX_TILES
#define SX_TILES 64      // -DSX_TILES=256: 805 MB, past the 256 MB Infinity Cache (what the kernels see on big batches)
#endif
constexpr int kTiles = SX_TILES, kPixels = 512 * 512;

template <int kLogs, int kFma>
__device__ __forceinline__ float work(float r, float g, float b, float acc) {
    float x = r, y = g, z = b;
    if constexpr (kLogs > 0) {
        x = __builtin_amdgcn_logf(fmaf(r, 255.f, 1.f));
        y = __builtin_amdgcn_logf(fmaf(g, 255.f, 1.f));
        z = __builtin_amdgcn_logf(fmaf(b, 255.f, 1.f));
    }
    float s = x + y + z;
#pragma unroll
    for (int k = 0; k < kFma; ++k) s = fmaf(s, 1.0001f, (k & 1) ? x : y);
    return acc + s;
}

template <int kLogs, int kFma, bool kPrefetch, int kChunk, int kMinWaves>
__global__ __launch_bounds__(256, kMinWaves) void pass_kernel(const float* __restrict__ in, float* __restrict__ sink) {
    constexpr int per_tile = kPixels / kChunk;
    const int tile = blockIdx.x / per_tile, chunk = blockIdx.x % per_tile;
    const float* img = in + (size_t)tile * 3 * kPixels;
    const int p_begin = chunk * kChunk, p_end = p_begin + kChunk;
    float acc = 0.f;
    if constexpr (!kPrefetch) {
        for (int p = p_begin + threadIdx.x * 4; p < p_end; p += 1024) {
            const float4 r = *reinterpret_cast<const float4*>(img + p), g = *reinterpret_cast<const float4*>(img + kPixels + p),
                         b = *reinterpret_cast<const float4*>(img + 2 * kPixels + p);
            acc = work<kLogs, kFma>(r.x, g.x, b.x, acc);
            acc = work<kLogs, kFma>(r.y, g.y, b.y, acc);
            acc = work<kLogs, kFma>(r.z, g.z, b.z, acc);
            acc = work<kLogs, kFma>(r.w, g.w, b.w, acc);
        }
    } else {
        int p = p_begin + threadIdx.x * 4;
        float4 nr = *reinterpret_cast<const float4*>(img + p), ng = *reinterpret_cast<const float4*>(img + kPixels + p), nb = *reinterpret_cast<const float4*>(img + 2 * kPixels + p);
        for (; p < p_end; p += 1024) {
            const float4 r = nr, g = ng, b = nb;
            if (p + 1024 < p_end) {
                nr = *reinterpret_cast<const float4*>(img + p + 1024);
                ng = *reinterpret_cast<const float4*>(img + kPixels + p + 1024);
                nb = *reinterpret_cast<const float4*>(img + 2 * kPixels + p + 1024);
            }
            acc = work<kLogs, kFma>(r.x, g.x, b.x, acc);
            acc = work<kLogs, kFma>(r.y, g.y, b.y, acc);
            acc = work<kLogs, kFma>(r.z, g.z, b.z, acc);
            acc = work<kLogs, kFma>(r.w, g.w, b.w, acc);
        }
    }
    if (acc == 1234.5678f) sink[0] = acc;
}

__global__ void fill_kernel(float* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) out[i] = 0.25f + 0.5f * (float)((i * 2654435761u) & 1023) / 1024.f;
}

template <int kLogs, int kFma, bool kPrefetch, int kChunk, int kMinWaves>
int run(const float* in, float* sink, hipEvent_t e0, hipEvent_t e1) {
    const int grid = kTiles * (kPixels / kChunk);
    float best = 1e9f, sum = 0.f, ms;
    for (int rep = 0; rep < 12; ++rep) {
        CK(hipEventRecord(e0));
        pass_kernel<kLogs, kFma, kPrefetch, kChunk, kMinWaves><<<grid, 256>>>(in, sink);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("logs %d fma %3d prefetch %d chunk %5d min-waves %d : avg %6.1f us  min %6.1f us  (%.2f TB/s)\n", kLogs, kFma, (int)kPrefetch, kChunk, kMinWaves, sum / 10 * 1e3, best * 1e3,
           (double)kTiles * 3 * kPixels * 4 / (best * 1e-3) / 1e12);
    return 0;
}

int main() {
    float *in, *sink;
    const size_t n = (size_t)kTiles * 3 * kPixels;
    CK(hipMalloc(&in, n * 4));
    CK(hipMalloc(&sink, 256));
    fill_kernel<<<2048, 256>>>(in, n);
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    run<0, 0, false, 8192, 1>(in, sink, e0, e1);
    run<0, 0, true, 8192, 1>(in, sink, e0, e1);
    run<1, 0, false, 8192, 1>(in, sink, e0, e1);
    run<1, 10, false, 8192, 1>(in, sink, e0, e1);
    run<1, 20, false, 8192, 1>(in, sink, e0, e1);
    run<1, 30, false, 8192, 1>(in, sink, e0, e1);
    run<1, 40, false, 8192, 1>(in, sink, e0, e1);
    run<1, 60, false, 8192, 1>(in, sink, e0, e1);
    run<1, 30, true, 8192, 1>(in, sink, e0, e1);
    run<1, 40, true, 8192, 1>(in, sink, e0, e1);
    run<1, 30, false, 8192, 8>(in, sink, e0, e1);
    run<1, 30, false, 8192, 4>(in, sink, e0, e1);
    run<1, 30, false, 16384, 1>(in, sink, e0, e1);
    run<1, 30, false, 4096, 1>(in, sink, e0, e1);
    run<1, 30, false, 2048, 1>(in, sink, e0, e1);
    run<1, 30, true, 4096, 1>(in, sink, e0, e1);
    run<0, 0, false, 4096, 1>(in, sink, e0, e1);
    run<0, 0, false, 2048, 1>(in, sink, e0, e1);
    run<0, 0, false, 16384, 1>(in, sink, e0, e1);
    return 0;
}

// How much of the memory system's bandwidth does a streaming kernel see as a function of the loads it keeps in flight per thread?
// Read-only stream of 201 MB (fits the 256 MB Infinity Cache: "warm" = back to back, "cold" = after 1 GiB of other traffic) and of
// 805 MB, with U independent 16-byte loads per thread and trip (U = 1, 2, 4, 8), 256-thread workgroups, grid-stride, several grids.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mlpbench.hip -o tools/mlpbench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ in, size_t n4, float* __restrict__ sink) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = i + u * stride < n4 ? in[i + u * stride] : make_float4(0, 0, 0, 0);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 1234.5678f) sink[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n4) v[u] = in[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) if (i + u * stride < n4) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 w = {v[u].x, v[u].y, v[u].z, v[u].w};
            __builtin_nontemporal_store(w, reinterpret_cast<f4*>(&out[i + u * stride]));
        }
    }
}
__global__ __launch_bounds__(256) void fill_kernel(float4* __restrict__ out, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = make_float4(v, v, v, v);
}
template <int U> static void launch_read(int grid, const float4* a, size_t n4, float* sink) { read_kernel<U><<<grid, 256>>>(a, n4, sink); }
template <int U> static void launch_copy(int grid, const float4* a, float4* b, size_t n4) { copy_kernel<U><<<grid, 256>>>(a, b, n4); }

int main() {
    const size_t MB = 1 << 20;
    float *a, *b, *flush, *sink;
    CK(hipMalloc(&a, 805 * MB)); CK(hipMalloc(&b, 805 * MB)); CK(hipMalloc(&flush, 1024 * MB)); CK(hipMalloc(&sink, 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    fill_kernel<<<2048, 256>>>((float4*)a, 805 * MB / 16, 1.f);
    CK(hipDeviceSynchronize());
    printf("%6s %5s %2s | %10s %10s | %10s\n", "MB", "grid", "U", "read warm", "read cold", "copy warm");
    for (size_t smb : {201ul, 805ul}) {
        const size_t n4 = smb * MB / 16;
        for (int grid : {1024, 2048, 8192}) {
            for (int U : {1, 2, 4, 8}) {
                float t_rw = 1e9f, t_rc = 1e9f, t_cw = 1e9f, ms;
                auto rd = [&]() { switch (U) { case 1: launch_read<1>(grid, (const float4*)a, n4, sink); break; case 2: launch_read<2>(grid, (const float4*)a, n4, sink); break; case 4: launch_read<4>(grid, (const float4*)a, n4, sink); break; default: launch_read<8>(grid, (const float4*)a, n4, sink); } };
                auto cp = [&]() { switch (U) { case 1: launch_copy<1>(grid, (const float4*)a, (float4*)b, n4); break; case 2: launch_copy<2>(grid, (const float4*)a, (float4*)b, n4); break; case 4: launch_copy<4>(grid, (const float4*)a, (float4*)b, n4); break; default: launch_copy<8>(grid, (const float4*)a, (float4*)b, n4); } };
                for (int rep = 0; rep < 6; ++rep) { CK(hipEventRecord(e0)); rd(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 0 && ms < t_rw) t_rw = ms; }
                for (int rep = 0; rep < 3; ++rep) { fill_kernel<<<2048, 256>>>((float4*)flush, 1024 * MB / 16, (float)rep); CK(hipEventRecord(e0)); rd(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t_rc) t_rc = ms; }
                for (int rep = 0; rep < 5; ++rep) { CK(hipEventRecord(e0)); cp(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1)); if (rep > 0 && ms < t_cw) t_cw = ms; }
                const double gb = smb * MB / 1e9;
                printf("%6zu %5d %2d | %7.2f TB/s %7.2f TB/s | %7.2f TB/s (r+w)\n", smb, grid, U, gb / t_rw, gb / t_rc, 2 * gb / t_cw);
            }
        }
    }
    return 0;
}

// Device-memory ceilings on one MI355X for the access shapes of the stain kernels:
//   read-only streaming (3 fp32 planes, 16 B/lane), copy (read + write), each at several footprints,
//   back to back (Infinity-Cache warm when the footprint fits) and after a 1 GiB flush.
// Build: hipcc --offload-arch=gfx950 -O3 tools/membench.hip -o tools/membench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ in, size_t n4, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 1234.5678f) sink[0] = acc;
}

__global__ __launch_bounds__(256) void copy_kernel(const float4* __restrict__ in, float4* __restrict__ out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}

__global__ __launch_bounds__(256) void fill_kernel(float4* __restrict__ out, size_t n4, float v) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = make_float4(v, v, v, v);
}

int main() {
    const size_t MB = 1 << 20;
    float *a, *b, *flush, *sink;
    const size_t max_bytes = 805 * MB;
    CK(hipMalloc(&a, max_bytes));
    CK(hipMalloc(&b, max_bytes));
    CK(hipMalloc(&flush, 1024 * MB));
    CK(hipMalloc(&sink, 256));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    fill_kernel<<<2048, 256>>>((float4*)a, max_bytes / 16, 1.f);
    fill_kernel<<<2048, 256>>>((float4*)b, max_bytes / 16, 2.f);
    CK(hipDeviceSynchronize());
    const size_t sizes_mb[] = {24, 96, 192, 402, 805};
    const int grids[] = {1024, 2048, 4096};
    printf("%8s %6s | %12s %12s | %12s %12s\n", "MB", "grid", "read warm", "read cold", "copy warm", "copy cold");
    for (size_t smb : sizes_mb) {
        const size_t bytes = smb * MB, n4 = bytes / 16;
        for (int grid : grids) {
            float t_rw = 1e9, t_rc = 1e9, t_cw = 1e9, t_cc = 1e9, ms;
            for (int rep = 0; rep < 6; ++rep) {   // warm: same buffer back to back
                CK(hipEventRecord(e0));
                read_kernel<<<grid, 256>>>((const float4*)a, n4, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < t_rw) t_rw = ms;
            }
            for (int rep = 0; rep < 3; ++rep) {   // cold: 1 GiB of other traffic in between
                fill_kernel<<<2048, 256>>>((float4*)flush, 1024 * MB / 16, (float)rep);
                CK(hipEventRecord(e0));
                read_kernel<<<grid, 256>>>((const float4*)a, n4, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < t_rc) t_rc = ms;
            }
            for (int rep = 0; rep < 6; ++rep) {
                CK(hipEventRecord(e0));
                copy_kernel<<<grid, 256>>>((const float4*)a, (float4*)b, n4 / 2);   // bytes/2 read + bytes/2 written
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep > 0 && ms < t_cw) t_cw = ms;
            }
            for (int rep = 0; rep < 3; ++rep) {
                fill_kernel<<<2048, 256>>>((float4*)flush, 1024 * MB / 16, (float)rep);
                CK(hipEventRecord(e0));
                copy_kernel<<<grid, 256>>>((const float4*)a, (float4*)b, n4 / 2);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < t_cc) t_cc = ms;
            }
            const double gb = bytes / 1e9;
            printf("%8zu %6d | %7.1f us %6.0f GB/s %7.1f us %6.0f GB/s | %7.1f us %6.0f GB/s %7.1f us %6.0f GB/s\n", smb, grid, t_rw * 1e3, gb / (t_rw * 1e-3),
                   t_rc * 1e3, gb / (t_rc * 1e-3), t_cw * 1e3, gb / (t_cw * 1e-3), t_cc * 1e3, gb / (t_cc * 1e-3));
        }
    }
    return 0;
}

// Can the per-tile stages overlap the next streaming stage without mixed launches?  ("measured and dropped")
//   mode 0: one stream, [read 201 MB | 12 us spin on 64 workgroups] x 3 + copy -- the shape of a transform
//   mode 1: the same on a side stream with an event fork/join around it: the cost of a fork/join alone (+30 us measured)
//   mode 2/3: two half-size sequences on two side streams, the second one delayed: overlap gains vs fork/join cost
// (hipExtAnyOrderLaunch was tested with the same kernels: the flag is ignored on gfx9xx.)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_overlap.hip -o tools/launch_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin_kernel(unsigned long long* out, unsigned long long ticks) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) out[blockIdx.x] = t0;
}
__global__ __launch_bounds__(256) void read_kernel(const float4* in, size_t n4, float* sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) { const float4 v = in[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 1234.5f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void copy_kernel(const float4* in, float4* out, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) out[i] = in[i];
}
int main() {
    float *a, *b, *sink; unsigned long long* o;
    const size_t bytes = 201u << 20;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&sink, 256)); CK(hipMalloc(&o, 8 * 1024));
    CK(hipMemset(a, 0, bytes));
    hipStream_t s0, s1, s2; CK(hipStreamCreate(&s0)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t e0, e1, fork, j1, j2; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j1, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&j2, hipEventDisableTiming));
    const float4* a4 = (const float4*)a; float4* b4 = (float4*)b; const size_t n4 = bytes / 16, h4 = n4 / 2;
    const unsigned long long stage = 1200;   // 12 us
    auto seq = [&](hipStream_t s, const float4* in, float4* out, size_t n, int wgs, int grid) {
        for (int k = 0; k < 3; ++k) { read_kernel<<<grid, 256, 0, s>>>(in, n, sink); spin_kernel<<<wgs, 1024, 0, s>>>(o, stage); }
        copy_kernel<<<grid, 256, 0, s>>>(in, out, n);
    };
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f, sum = 0, ms;
        for (int rep = 0; rep < 12; ++rep) {
            CK(hipEventRecord(e0, s0));
            if (mode == 0) {
                seq(s0, a4, b4, n4, 64, 1024);
            } else if (mode == 1) {     // one side stream only: cost of a fork/join
                CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0));
                seq(s1, a4, b4, n4, 64, 1024);
                CK(hipEventRecord(j1, s1)); CK(hipStreamWaitEvent(s0, j1, 0));
            } else {                    // two halves on two side streams, second one delayed
                CK(hipEventRecord(fork, s0)); CK(hipStreamWaitEvent(s1, fork, 0)); CK(hipStreamWaitEvent(s2, fork, 0));
                if (mode == 3) spin_kernel<<<1, 64, 0, s2>>>(o + 512, 1500);
                seq(s1, a4, b4, h4, 32, 512);
                seq(s2, a4 + h4, b4 + h4, h4, 32, 512);
                CK(hipEventRecord(j1, s1)); CK(hipEventRecord(j2, s2)); CK(hipStreamWaitEvent(s0, j1, 0)); CK(hipStreamWaitEvent(s0, j2, 0));
            }
            CK(hipEventRecord(e1, s0));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep >= 2) { sum += ms; if (ms < best) best = ms; }
        }
        const char* names[] = {"single stream", "fork/join to one side stream", "two halves, two streams", "two halves, second delayed 15 us"};
        printf("%-36s avg %.1f us  min %.1f us\n", names[mode], sum / 10 * 1e3, best * 1e3);
    }
    return 0;
}

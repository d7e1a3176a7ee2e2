// Exhaustive checks on the device that x / 255.0f (IEEE division: what torch's `/ 255` computes on the float32 value) equals the
// three-instruction form  q = x * c,  r = fma(-q, 255, x)  (the exact residual),  q + r * c   bit for bit:
//   (1) the 256 integer grey levels (uint8 pixels, Elem<uint8_t>::load),
//   (2) every bfloat16 and every float16 bit pattern (the fused /255 of half-precision tiles),
//   (3) every float32 in [0, 256) (the fused /255 of float32 tiles: the clamped result lies in [0, 255]).
// hipcc --offload-arch=gfx950 -O3 tools/check_div255.hip -o tools/check_div255 && tools/check_div255
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <hip/hip_bf16.h>
#include <cstdio>
#include <cstdint>
__device__ __forceinline__ bool same(float x) {
    const float want = x / 255.0f;
    const float c = 1.0f / 255.0f;
    const float q = x * c;
    const float got = fmaf(fmaf(-q, 255.0f, x), c, q);
    return __float_as_uint(got) == __float_as_uint(want) || (want != want && got != got);      // NaN payloads aside
}
__global__ void levels(unsigned long long* bad) { if (!same((float)threadIdx.x)) atomicAdd(bad, 1ull); }
__global__ void halves(unsigned long long* bad) {
    const unsigned short bits = (unsigned short)(blockIdx.x * blockDim.x + threadIdx.x);
    if (!same(__bfloat162float(__ushort_as_bfloat16(bits)))) atomicAdd(bad + 1, 1ull);
    if (!same(__half2float(__ushort_as_half(bits)))) atomicAdd(bad + 2, 1ull);
}
__global__ void floats(unsigned long long* bad, uint32_t last) {      // every pattern 0 .. last (positive floats up to 256)
    for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b <= last; b += (uint64_t)gridDim.x * blockDim.x)
        if (!same(__uint_as_float((uint32_t)b))) atomicAdd(bad + 3, 1ull);
}
int main() {
    unsigned long long* bad;
    hipMalloc(&bad, 4 * sizeof(unsigned long long));
    hipMemset(bad, 0, 4 * sizeof(unsigned long long));
    hipLaunchKernelGGL(levels, dim3(1), dim3(256), 0, 0, bad);
    hipLaunchKernelGGL(halves, dim3(256), dim3(256), 0, 0, bad);
    const uint32_t last = 0x43800000u;      // 256.0f
    hipLaunchKernelGGL(floats, dim3(4096), dim3(256), 0, 0, bad, last);
    unsigned long long h[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost);
    printf("mismatches: grey levels %llu of 256, bfloat16 %llu of 65536, float16 %llu of 65536, float32 in [0,256] %llu of %u\n", h[0], h[1], h[2], h[3], last + 1);
    return (h[0] | h[1] | h[2] | h[3]) != 0;
}

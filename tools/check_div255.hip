// Exhaustive check (256 values) on the device: u / 255.0f (IEEE division, what torch's `.float() / 255` computes) against the
// three-instruction form used by Elem<uint8_t>::load: q = u * c, r = fma(-q, 255, u) (the exact residual), q + r * c.
// hipcc --offload-arch=gfx950 tools/check_div255.hip -o tools/check_div255 && tools/check_div255
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void check(int* bad) {
    const float u = (float)threadIdx.x;
    const float want = u / 255.0f;
    const float c = 1.0f / 255.0f;      // 0x3b808081
    const float q = u * c;
    const float r = fmaf(-q, 255.0f, u);
    const float got = fmaf(r, c, q);
    if (__float_as_uint(got) != __float_as_uint(want)) atomicAdd(bad, 1);
}
int main() {
    int* bad;
    hipMalloc(&bad, sizeof(int));
    hipMemset(bad, 0, sizeof(int));
    hipLaunchKernelGGL(check, dim3(1), dim3(256), 0, 0, bad);
    int h = -1;
    hipMemcpy(&h, bad, sizeof(int), hipMemcpyDeviceToHost);
    printf("mismatches: %d of 256\n", h);
    return h != 0;
}

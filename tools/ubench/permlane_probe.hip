// Probe (gfx950): what v_permlane16_swap_b32 leaves in its two operands (rows = groups of 16 lanes).
// hipcc --offload-arch=gfx950 -O3 -o permlane_probe permlane_probe.hip && ./permlane_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned *o)
{
    unsigned a = threadIdx.x, b = threadIdx.x + 100;
    auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    o[threadIdx.x] = r[0];
    o[64 + threadIdx.x] = r[1];
}
int main()
{
    unsigned *d, h[128];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int v = 0; v < 2; ++v) {
        printf("result %d (a = lane, b = 100 + lane):", v);
        for (int l = 0; l < 64; l += 16) printf("  lanes %2d-%2d: %3u..%3u", l, l + 15, h[v * 64 + l], h[v * 64 + l + 15]);
        printf("\n");
    }
    return 0;
}

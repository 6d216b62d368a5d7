// store_probe.hip -- how fast can a CU write accumulator tiles?  (round 3: the GEMM epilogue question)
// Each workgroup of 256 threads writes `iters` tiles of 128 rows x 256 fp32 columns (128 KiB) in the GEMM epilogue's shape:
// per wave 128 store instructions, each 2 x 128 contiguous bytes (one accumulator register of a 32x32 tile: lanes 0-31 one
// row, lanes 32-63 the row four below).  Modes: 0 plain dword stores, 1 non-temporal, 2 dwordx4 (4 rows x 64 B per
// 16-lane group -- partial lines), 3 plain stores into a small per-block window (L2-resident).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/store_probe tools/store_probe.hip ;  run: tools/store_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void store_kernel(float *out, size_t tile_stride, int iters, int mode, int window)
{
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float v = (float)threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        size_t tile = (size_t)blockIdx.x * iters + it;
        if (mode == 3) tile = (size_t)(blockIdx.x % window);
        float *base = out + tile * tile_stride + wid * 64;          // wave's 64 columns of the 256
        const int loff = (4 * (lane >> 5)) * 256 + (lane & 31);
#pragma unroll 16
        for (int i = 0; i < 64; ++i) {
            const int r = (i >> 4) * 32 + (i & 3) + 8 * ((i >> 2) & 3);
            float *rowp = base + (size_t)r * 256 + loff;
            if (mode == 1) {
                __builtin_nontemporal_store(v, rowp);
                __builtin_nontemporal_store(v + 1.0f, rowp + 32);
            } else {
                rowp[0] = v;
                rowp[32] = v + 1.0f;
            }
            v += 0.5f;
        }
    }
}

int main(int argc, char **argv)
{
    const int iters = 64;
    const size_t tile_floats = 128 * 256;
    const int maxblocks = 2048;
    float *buf;
    if (hipMalloc(&buf, sizeof(float) * tile_floats * (size_t)maxblocks * iters) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int mode : {0, 1, 3}) {
        for (int nb : {8, 32, 64, 128, 256, 512, 1024, 2048}) {
            hipLaunchKernelGGL(store_kernel, dim3(nb), dim3(256), 0, 0, buf, tile_floats, iters, mode, 64);
            hipDeviceSynchronize();
            hipEventRecord(a, 0);
            hipLaunchKernelGGL(store_kernel, dim3(nb), dim3(256), 0, 0, buf, tile_floats, iters, mode, 64);
            hipEventRecord(b, 0);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            const double bytes = (double)nb * iters * tile_floats * 4;
            printf("mode %d blocks %5d : %8.3f ms  %8.1f GB/s total  %7.1f GB/s per block  (%.1f us per 128 KiB tile)\n", mode, nb, ms,
                   bytes / ms / 1e6, bytes / ms / 1e6 / nb, 1e3 * ms / iters);
        }
    }
    return 0;
}

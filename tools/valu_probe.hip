// valu_probe.hip -- diagnostic: cycles per wave64 VALU instruction (independent and dependent v_fma_f32 chains, DPP max,
// LDS round trip + barrier) at 1, 2, 4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o valu_probe valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP>
__global__ void fma_kernel(float *out, int iters, unsigned long long *cyc)
{
    float a[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) a[i] = threadIdx.x * 1e-3f + i;
    const float b = 1.0000001f, c = 1e-7f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) a[i] = __builtin_fmaf(a[i], b, c);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// one "CRF-like" step: barrier, LDS read, few dependent ops, LDS write
__global__ void step_kernel(float *out, int iters, unsigned long long *cyc)
{
    __shared__ float s[2][1024];
    const int tid = threadIdx.x;
    s[0][tid] = tid;
    __syncthreads();
    float v = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        const float x = s[it & 1][(tid * 7 + 3) % blockDim.x];
        v = __builtin_fmaf(x, 0.999f, v * 1e-3f);
        s[(it + 1) & 1][tid] = v;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + tid] = v;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <typename F>
double run(F launch, int blocks, unsigned long long *d_cyc, std::vector<unsigned long long> &h)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(a);
    launch();
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    h.resize(blocks);
    hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    return ms;
}

int main()
{
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, sizeof(float) * 256 * 16 * 1024);
    hipMalloc(&cyc, sizeof(unsigned long long) * 4096);
    std::vector<unsigned long long> h;
    const int iters = 20000;
    for (int wps : {1, 2, 4}) {          // waves per SIMD: one block per CU of wps*4 waves
        const int threads = wps * 4 * 64, blocks = 256;
        double ms = run([&] { hipLaunchKernelGGL(fma_kernel<1>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); }, blocks, cyc, h);
        printf("dependent fma chain, %d waves/SIMD: %.2f cycles per instr per wave (wall %.2f ms)\n", wps, (double)h[0] / (iters * 16.0), ms);
        ms = run([&] { hipLaunchKernelGGL(fma_kernel<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); }, blocks, cyc, h);
        printf("8 independent chains,  %d waves/SIMD: %.2f cycles per instr per wave -> %.2f cycles per instr per SIMD (wall %.2f ms)\n", wps,
               (double)h[0] / (iters * 16.0 * 8), (double)h[0] / (iters * 16.0 * 8) / wps, ms);
    }
    for (int threads : {64, 128, 256, 512}) {
        for (int bpc : {1, 2}) {
            const int blocks = 256 * bpc;
            double ms = run([&] { hipLaunchKernelGGL(step_kernel, dim3(blocks), dim3(threads), 0, 0, out, iters, cyc); }, blocks, cyc, h);
            printf("barrier+LDS step, %d threads, %d blocks/CU: %.1f cycles per step (wall %.2f ms)\n", threads, bpc, (double)h[0] / iters, ms);
        }
    }
    return 0;
}

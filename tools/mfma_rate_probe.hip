// Diagnostic: issue rate of the MFMA forms the recurrence uses (cycles per instruction, one wave per SIMD, four independent
// accumulators): build with hipcc --offload-arch=gfx950 -O3 tools/mfma_rate_probe.hip -o tools/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v16i __attribute__((ext_vector_type(16)));

template <int KIND>
__global__ __launch_bounds__(256) void probe(unsigned long long *out, int iters)
{
    f16v c[4] = {};
    v16i ci[4] = {};
    half8 a, b;
    v4i ai, bi;
    v8i aq, bq;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f - i * 0.01f); }
    for (int i = 0; i < 4; ++i) { ai[i] = threadIdx.x * 77 + i; bi[i] = threadIdx.x * 13 + 5 * i; }
    for (int i = 0; i < 8; ++i) { aq[i] = threadIdx.x * 77 + i; bq[i] = threadIdx.x * 13 + 5 * i; }
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (KIND == 0) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[j], 0, 0, 0);
            if (KIND == 1) ci[j] = __builtin_amdgcn_mfma_i32_32x32x32_i8(ai, bi, ci[j], 0, 0, 0);
            if (KIND == 2) c[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq, bq, c[j], 0, 0, 0, 127, 0, 127);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0;
    for (int j = 0; j < 4; ++j) for (int r = 0; r < 16; ++r) s += c[j][r] + (float)ci[j][r];
    if (s == 12345.678f) out[1] = 1;
    if (blockIdx.x == 0 && threadIdx.x == 0) out[0] = t1 - t0;
}

int main()
{
    unsigned long long *d, h[2];
    hipMalloc(&d, 16);
    const int iters = 4096;
    const char *names[3] = {"v_mfma_f32_32x32x16_f16", "v_mfma_i32_32x32x32_i8", "v_mfma_scale_f32_32x32x64_f8f6f4 (fp8)"};
    for (int k = 0; k < 3; ++k) {
        for (int rep = 0; rep < 2; ++rep) {
            if (k == 0) hipLaunchKernelGGL(probe<0>, dim3(256), dim3(256), 0, 0, d, iters);
            if (k == 1) hipLaunchKernelGGL(probe<1>, dim3(256), dim3(256), 0, 0, d, iters);
            if (k == 2) hipLaunchKernelGGL(probe<2>, dim3(256), dim3(256), 0, 0, d, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
        printf("%-45s %.1f cycles per instruction (one wave per SIMD, 4 independent accumulators)\n", names[k], (double)h[0] / (4.0 * iters));
    }
    return 0;
}

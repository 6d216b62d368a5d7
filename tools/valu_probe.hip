// valu_probe.hip -- diagnostic: SIMD issue cost of wave64 VALU instructions on gfx950, measured with inline asm so that
// hipcc can neither pack nor reorder them: plain v_fma_f32, packed v_pk_fma_f32, v_max_f32_dpp, and a dependent chain,
// at 1, 2, 4 waves per SIMD; plus a barrier + LDS round trip.
// Build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-result -o valu_probe valu_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define R8(x) x x x x x x x x
template <int MODE>
__global__ void k(float *out, int iters)
{
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    float b = 1.0000001f, c = 1e-7f;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0)        // 8 independent scalar FMAs x 8
            asm volatile(R8("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                            "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        else if (MODE == 1)   // dependent scalar FMA chain x 64
            asm volatile(R8(R8("v_fma_f32 %0, %0, %1, %2\n")) : "+v"(a0) : "v"(b), "v"(c));
        else if (MODE == 2) { // 4 independent packed FMAs x 16 (same number of FLOPs as MODE 0)
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, bb = {b, b}, cc = {c, c};
            asm volatile(R8("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                            "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n")
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb), "v"(cc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else                // 8 independent v_max_f32_dpp x 8
            asm volatile(R8("v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                            "v_max_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                            "v_max_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                            "v_max_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_max_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE>
double run(float *out, int threads, int iters)
{
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipDeviceSynchronize();
    hipEventRecord(a);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, iters);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float *out;
    hipMalloc(&out, sizeof(float) * 256 * 1024);
    const int iters = 40000;
    const double ghz = 2.4;
    const char *names[4] = {"v_fma_f32 (8 independent)", "v_fma_f32 (dependent chain)", "v_pk_fma_f32 (4 independent)", "v_max_f32_dpp (8 independent)"};
    for (int wps : {1, 2, 4}) {
        const int threads = wps * 4 * 64;
        double ms[4] = {run<0>(out, threads, iters), run<1>(out, threads, iters), run<2>(out, threads, iters), run<3>(out, threads, iters)};
        for (int m = 0; m < 4; ++m) {
            const double per_wave = 64.0 * iters;                 // instructions per wave
            const double ns = ms[m] * 1e6 / (per_wave * wps);     // per instruction per SIMD
            printf("%-32s %d waves/SIMD: %.2f ns = %.2f cycles @%.1f GHz per instruction per SIMD (%.2f per wave)\n", names[m], wps, ns,
                   ns * ghz, ghz, ns * ghz * wps);
        }
    }
    return 0;
}

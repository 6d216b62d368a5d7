// Microbenchmark (gfx950): cycles per v_mfma_f32_32x32x16_f16 on one wave per SIMD for dependent accumulation chains with
// different instructions in the gaps.  hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef int v8i __attribute__((ext_vector_type(8)));

template <int CHAINS, int GAP, bool MIXQ>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float seed)
{
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed * i;
    __syncthreads();
    floatx16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(seed + threadIdx.x * 0.001f); b[i] = (_Float16)(seed * 2 + i); }
    v8i qa, qb;
    for (int i = 0; i < 8; ++i) { qa[i] = threadIdx.x * 77 + i; qb[i] = threadIdx.x * 31 + i; }
    float v0 = seed, v1 = seed * 3;
    unsigned addr = (threadIdx.x & 63) * 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) {
                if (MIXQ && (u & 1) && c == 0)
                    acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(qa, qb, acc[c], 0, 0, 0, 127, 0, 127);
                else
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[c], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (GAP == 1) {          // two plain VALU
                    asm volatile("v_add_f32 %0, %0, %1\n\tv_mul_f32 %1, %1, %0" : "+v"(v0), "+v"(v1));
                } else if (GAP == 2) {   // six VALU
                    asm volatile("v_add_f32 %0, %0, %1\n\tv_mul_f32 %1, %1, %0\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %1, %1, %0\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %1, %1, %0" : "+v"(v0), "+v"(v1));
                } else if (GAP == 3) {   // an LDS read and its wait
                    float4 t;
                    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(addr));
                    v0 += t.x;
                } else if (GAP == 4) {   // two transcendental + two plain VALU
                    asm volatile("v_exp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_add_f32 %0, %0, %1\n\tv_mul_f32 %1, %1, %0" : "+v"(v0), "+v"(v1));
                } else if (GAP == 5) {   // s_nop only
                    asm volatile("s_nop 3");
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = v0 + v1;
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int CHAINS, int GAP, bool MIXQ>
void run(const char *name, float *out, unsigned long long *cyc)
{
    const int iters = 2000;
    k<CHAINS, GAP, MIXQ><<<256, 256>>>(out, cyc, iters, 0.001f);
    hipDeviceSynchronize();
    k<CHAINS, GAP, MIXQ><<<256, 256>>>(out, cyc, iters, 0.001f);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-52s chains %d  %7.1f cycles per MFMA (s_memtime ticks; 100 MHz -> x clock/100MHz)\n", name, CHAINS, (double)c / (iters * 8.0 * CHAINS));
}

int main()
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 8);
    run<1, 0, false>("dependent chain, nothing in the gaps", out, cyc);
    run<1, 5, false>("dependent chain, s_nop 3", out, cyc);
    run<1, 1, false>("dependent chain, 2 VALU", out, cyc);
    run<1, 2, false>("dependent chain, 6 VALU", out, cyc);
    run<1, 4, false>("dependent chain, 2 trans + 2 VALU", out, cyc);
    run<1, 3, false>("dependent chain, ds_read_b128 + wait", out, cyc);
    run<2, 0, false>("two chains, nothing", out, cyc);
    run<2, 1, false>("two chains, 2 VALU", out, cyc);
    run<2, 2, false>("two chains, 6 VALU", out, cyc);
    run<2, 4, false>("two chains, 2 trans + 2 VALU", out, cyc);
    run<3, 2, false>("three chains, 6 VALU", out, cyc);
    run<4, 2, false>("four chains, 6 VALU", out, cyc);
    run<4, 4, false>("four chains, 2 trans + 2 VALU", out, cyc);
    run<1, 0, true>("one chain, every other MFMA the FP8 32x32x64", out, cyc);
    run<1, 1, true>("one chain, FP8 mixed, 2 VALU", out, cyc);
    run<2, 0, true>("two chains, chain 0 mixed with FP8", out, cyc);
    run<2, 2, true>("two chains, chain 0 mixed with FP8, 6 VALU", out, cyc);
    return 0;
}

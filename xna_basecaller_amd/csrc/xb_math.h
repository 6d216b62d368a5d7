// The decode contract's scalar math (shared by specification with oracle/xna_oracle.c, written independently): IEEE
// binary32, fma only where __builtin_fmaf is written (files that include this are built with -ffp-contract=off), exp / log
// are these fixed polynomials.  Device-only, internal linkage.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {

__device__ __forceinline__ float bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2bits(float f) { return __builtin_bit_cast(uint32_t, f); }

constexpr float XB_EXP_MAGIC = 12582912.0f;     // 1.5 * 2^23: fma(x, log2e, magic) leaves round(x log2e) in the low mantissa bits
__device__ __forceinline__ float xb_exp_scale(float t)   // 2^n from t = magic + n
{
    return bits2f((f2bits(t) << 23) + 0x3f800000u);
}
__device__ __forceinline__ float xb_expf(float x)
{
    x = __builtin_amdgcn_fmed3f(x, -87.0f, 88.0f);              // clamp (one instruction), no flush to zero
    const float t = __builtin_fmaf(x, 1.44269504088896341f, XB_EXP_MAGIC);
    const float n = t - XB_EXP_MAGIC;
    float r = __builtin_fmaf(n, -0.693359375f, x);
    r = __builtin_fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    const float r2 = r * r;
    const float y = __builtin_fmaf(p, r2, r) + 1.0f;
    return y * xb_exp_scale(t);
}

// range reduction of the log: x = m * 2^e with m in (sqrt(1/2), sqrt(2)]; returns f = m - 1 and e as a float
__device__ __forceinline__ void xb_log_reduce(float x, float &f, float &fe)
{
    const uint32_t ix = f2bits(x);
    int e = (int)(ix >> 23) - 127;
    float m = bits2f((ix & 0x007fffffu) | 0x3f800000u);
    const bool big = m > 1.41421356237309505f;
    m = big ? m * 0.5f : m;
    e = big ? e + 1 : e;
    f = m - 1.0f;
    fe = (float)e;
}
__device__ __forceinline__ float xb_logf(float x)
{
    float f, fe;
    xb_log_reduce(x, f, fe);
    const float z = f * f;
    const float z2 = z * z;
    const float z4 = z2 * z2;
    const float q01 = __builtin_fmaf(-2.4999993993e-1f, f, 3.3333331174e-1f);
    const float q23 = __builtin_fmaf(-1.6668057665e-1f, f, 2.0000714765e-1f);
    const float q45 = __builtin_fmaf(-1.2420140846e-1f, f, 1.4249322787e-1f);
    const float q67 = __builtin_fmaf(-1.1514610310e-1f, f, 1.1676998740e-1f);
    const float q03 = __builtin_fmaf(q23, z, q01);
    const float q47 = __builtin_fmaf(q67, z, q45);
    const float q07 = __builtin_fmaf(q47, z2, q03);
    const float p = __builtin_fmaf(7.0376836292e-2f, z4, q07);
    float y = (f * z) * p;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = f + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}
}  // namespace

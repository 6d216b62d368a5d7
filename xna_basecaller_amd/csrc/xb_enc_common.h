// xb_enc_common.h -- typedefs and scalar helpers shared by xb_encoder.hip (conv front end, GEMMs) and xb_lstm.hip (the recurrence).
// Internal linkage (anonymous namespace): each translation unit gets its own copy.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <type_traits>

#include "xb_internal.h"

namespace {

using xb::half_t;
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float fast_sigmoid(float x) { return fast_rcp(1.0f + __expf(-x)); }
__device__ __forceinline__ float fast_tanh(float x)
{
    // 1 - 2/(e^{2x}+1); exact limits at +-inf, abs error ~1e-7
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f * fast_rcp(e + 1.0f);
}
__device__ __forceinline__ float silu(float x) { return x * fast_sigmoid(x); }

__device__ __forceinline__ void split_f16(float v, half_t &hi, half_t &lo)
{
    hi = (half_t)v;
    lo = (half_t)(v - (float)hi);
}

// ---- q8 image helpers (xb_internal.h "q8 image"): OCP e4m3 bytes of hi * 2^e and of (v - hi) * 2^(e+11)
typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ float clamp448(float v) { return __builtin_fminf(__builtin_fmaxf(v, -448.0f), 448.0f); }
// the conversion returns NaN (0x7f) above 448, hence the clamp wherever the magnitude is not bounded by construction
template <bool HIGH_WORD>
__device__ __forceinline__ unsigned fp8_pair(float a, float b, unsigned old)
{
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, (int)old, HIGH_WORD);
}
__device__ __forceinline__ void q8_bytes(float v, int e, half_t &hi, unsigned char &h8, unsigned char &l8)
{
    hi = (half_t)v;
    const float lo = v - (float)hi;
    const unsigned pk = fp8_pair<false>(clamp448(__builtin_ldexpf((float)hi, e)), clamp448(__builtin_ldexpf(lo, e + 11)), 0u);
    h8 = (unsigned char)(pk & 0xff);
    l8 = (unsigned char)((pk >> 8) & 0xff);
}
// byte offset of element (row, col) inside a q8 image with `ld` columns: the h8 byte (its l8 byte is 32 further)
__device__ __forceinline__ size_t q8_offset(size_t row, int ld, int col)
{
    return (row * ld + (size_t)(col & ~31)) * 2 + (col & 31);
}

}  // namespace

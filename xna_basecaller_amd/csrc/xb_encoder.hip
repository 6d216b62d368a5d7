// xb_encoder.hip -- Conv1d front-end, split-fp16 MFMA GEMM and LSTM recurrence for gfx950.
//
// Replaces Model.forward = bonito.nn Serial of (ub-bonito/bonito/crf/model.py:147-160):
//   Convolution x3 (nn.py:57-68), Permute (nn.py:156-164), LSTM x5 with alternating direction
//   (nn.py:176-193,216-220), LinearCRFEncoder (nn.py:112-133).
//
// Arithmetic: the dense contractions (conv3 as an im2col GEMM, the LSTM input and recurrent
// projections, the CRF linear layer) run on v_mfma_f32_32x32x16_f16 with every fp32 operand
// split into hi + lo fp16 halves and three products (hi*hi + hi*lo + lo*hi) accumulated in
// fp32 -- ~2^-21 relative operand error, i.e. fp32-grade scores (|err| ~1e-5) at 3/16 of the
// cost of the exact-f32 MFMA.  nsplit = 1 keeps only hi*hi (the reference's model.half()).
// Gates, cell state and activations are fp32 VALU.
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

// ======================================================================================
// conv1 + conv2 + im2col of conv3's input
// ======================================================================================
constexpr int CF_TT = 32;   // output time steps per workgroup

__global__ __launch_bounds__(256) void conv_front_kernel(xb::ConvFrontParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x;
    const int n = blockIdx.y;
    const int t0 = blockIdx.x * CF_TT;
    const int W = p.winlen, ST = p.stride, pad = W / 2, L = p.L;
    const int nq = (CF_TT - 1) * ST + W;      // conv2 positions needed
    const int q0 = t0 * ST - pad;
    float *sig = reinterpret_cast<float *>(smem_raw);   // [nq + 8]
    float *a1 = sig + nq + 8;                            // [4][nq + 4]
    float *a2 = a1 + 4 * (nq + 4);                       // [16][nq]
    float *w2s = a2 + 16 * nq;                           // [320] + b2 [16] + w1 [20] + b1 [4]
    float *b2s = w2s + 320, *w1s = b2s + 16, *b1s = w1s + 20;

    for (int i = tid; i < 320; i += 256) w2s[i] = p.w2[i];
    if (tid < 16) b2s[tid] = p.b2[tid];
    if (tid < 20) w1s[tid] = p.w1[tid];
    if (tid < 4) b1s[tid] = p.b1[tid];
    const float *x = (p.signal2 && n >= p.split) ? p.signal2 + (size_t)(n - p.split) * L : p.signal + (size_t)n * L;
    for (int i = tid; i < nq + 8; i += 256) {
        const int pos = q0 - 4 + i;
        sig[i] = (pos >= 0 && pos < L) ? x[pos] : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < 4 * (nq + 4); i += 256) {
        const int c = i / (nq + 4), r = i % (nq + 4);
        const int pos = q0 - 2 + r;
        float v = 0.0f;
        if (pos >= 0 && pos < L) {
            float acc = b1s[c];
#pragma unroll
            for (int k = 0; k < 5; ++k) acc += w1s[c * 5 + k] * sig[r + k];
            v = silu(acc);
        }
        a1[c * (nq + 4) + r] = v;
    }
    __syncthreads();
    for (int i = tid; i < 16 * nq; i += 256) {
        const int c = i / nq, r = i % nq;
        const int pos = q0 + r;
        float v = 0.0f;
        if (pos >= 0 && pos < L) {
            float acc = b2s[c];
#pragma unroll
            for (int ci = 0; ci < 4; ++ci)
#pragma unroll
                for (int k = 0; k < 5; ++k) acc += w2s[(c * 4 + ci) * 5 + k] * a1[ci * (nq + 4) + r + k];
            v = silu(acc);
        }
        a2[c * nq + r] = v;
    }
    __syncthreads();
    // im2col rows of this tile, four consecutive columns per thread: 8 bytes of hi and either 8 bytes of lo or, for the
    // q8 image, one dword of h8 and one of l8 (four columns never straddle a 32-column block)
    const int kp = p.kp, kv = 16 * W, kq = kp / 4;
    for (int i = tid; i < CF_TT * kq; i += 256) {
        const int tt = i / kq, col = (i % kq) * 4;
        const int t = t0 + tt;
        if (t >= p.T) break;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cj = col + j;
            v[j] = cj < kv ? a2[(cj / W) * nq + tt * ST + cj % W] : 0.0f;
        }
        const size_t orow = (size_t)t * p.N + n;
        half_t hi[4];
        unsigned short hb[4];
        if (p.q8) {
            float lo[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hi[j] = (half_t)v[j];
                lo[j] = clamp448((v[j] - (float)hi[j]) * 2048.0f);      // exponent 0: l8 = lo * 2^11
                hb[j] = __builtin_bit_cast(unsigned short, hi[j]);
            }
            unsigned h8 = fp8_pair<false>(clamp448((float)hi[0]), clamp448((float)hi[1]), 0u);
            h8 = fp8_pair<true>(clamp448((float)hi[2]), clamp448((float)hi[3]), h8);
            unsigned l8 = fp8_pair<false>(lo[0], lo[1], 0u);
            l8 = fp8_pair<true>(lo[2], lo[3], l8);
            unsigned char *q = reinterpret_cast<unsigned char *>(p.a_lo) + q8_offset(orow, kp, col);
            *reinterpret_cast<unsigned *>(q) = h8;
            *reinterpret_cast<unsigned *>(q + 32) = l8;
        } else {
            unsigned short lb[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                half_t lo;
                split_f16(v[j], hi[j], lo);
                hb[j] = __builtin_bit_cast(unsigned short, hi[j]);
                lb[j] = __builtin_bit_cast(unsigned short, lo);
            }
            *reinterpret_cast<uint2 *>(p.a_lo + orow * kp + col) =
                make_uint2(lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16));
        }
        *reinterpret_cast<uint2 *>(p.a_hi + orow * kp + col) =
            make_uint2(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16));
    }
}

// ======================================================================================
// split-fp16 MFMA GEMM:  D[m][n] = sum_k A[m][k] B[n][k]
// 256x256 block tile, BK = 32, 8 waves (2 in M x 4 in N), each wave 128x64 = 4x2 MFMA 32x32x16
// tiles, three products per tile pair (lo*hi, hi*lo, hi*hi).  Per 16-deep k-step a wave reads
// (4 + 2) fragments x (hi, lo) = 12 KiB from LDS for 24 MFMAs, which keeps the LDS read path at
// about half of its peak with all 8 waves running (a 128x128 tile saturated it).
// LDS image per operand part: [256 rows][4 cells of 16 B], cell index XOR-swizzled with
// (row >> 2) & 3 so that every 16-lane ds_read_b128 group hits 16 distinct bank slots.
// Register-staged double buffering: global loads of tile k+1 are in flight during the MFMAs of
// tile k and are written to the other LDS stage afterwards; one barrier per tile.
// ======================================================================================
constexpr int GBM = 256, GBN = 256, GBK = 32, GTHREADS = 512;


// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (block b -> XCD b & 7), each with its own
// L2.  The 32 blocks an XCD runs side by side are made one super-tile of SM x SN output tiles, so an A row panel is
// fetched by SN neighbours and a B panel by SM neighbours out of the SAME L2 instead of by all eight.
// N tiles per XCD super-tile (both GEMM kernels).  Two (three where the tile count is odd and a multiple of three): the weight
// panels an XCD works on at a time then stay in its 4 MiB L2 beside the streaming A panels, and every super-tile is full -- four
// (round 2) left the CRF linear layer's (6 tiles of 256) and conv3's (3 tiles) last super-tile half empty.  Measured with
// gemm4p_kernel, ms per step at batch 512, overlapped: 120.1 (two or three) vs 122.8 (four) vs 123.3 / 125.5 (six / twelve);
// linear layer alone 5.7 vs 6.9 ms (profiles/r03_gemm_supertile_width.txt).
__host__ __device__ inline int gemm_super_n(int NT) { return NT % 2 == 0 ? 2 : (NT % 3 == 0 ? 3 : 1); }

__device__ __forceinline__ bool gemm_tile_origin(const xb::GemmParams &p, int &m0, int &n0)
{
    const int MT = (p.M + GBM - 1) / GBM, NT = (p.Nn + GBN - 1) / GBN;
    const int SN = gemm_super_n(NT), SM = 32 / SN;
    const int ngroups = (NT + SN - 1) / SN, msup = (MT + SM - 1) / SM;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int q = (j >> 5) * 8 + xcd, w = j & 31;          // super-tile id, slot in it
    if (q >= msup * ngroups || w >= SM * SN) return false;
    const int mt = (q / ngroups) * SM + w / SN, nt = (q % ngroups) * SN + w % SN;
    if (mt >= MT || nt >= NT) return false;
    m0 = mt * GBM;
    n0 = nt * GBN;
    return true;
}

// epilogue shared by the GEMM kernels: a wave holds 4 x 2 accumulator tiles of 32x32 whose origin is (mw, nw): rows
// mw .. mw + 127, columns nw .. nw + 63 (nw a multiple of 64); a lane holds column (lane & 31) and 16 rows of each tile.
// The bias is loaded ONCE, ahead of all stores: a load inside the store loop makes every store wait (vmcnt counts stores
// too) for the one before it.
template <int EPI>
__device__ __forceinline__ void gemm_epilogue(const xb::GemmParams &p, const floatx16 (&acc)[4][2], int mw, int nw, int lane)
{
    if (mw >= p.M || nw >= p.Nn) return;         // a wave wholly outside the matrix (edge tiles)
    float bj[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = nw + j * 32 + (lane & 31);
        bj[j] = (p.bias && n < p.Nn) ? p.bias[n] : 0.0f;
    }
    const bool interior = mw + 128 <= p.M && nw + 64 <= p.Nn;
    // EPI_SILU_SPLIT: the second part of the output follows the arithmetic of the GEMM that CONSUMES it (out_fmt), which in a
    // mixed-precision encoder need not be this GEMM's own
    const bool out_q8 = p.out_fmt ? p.out_fmt == 2 : p.nsplit == 2;
    if (EPI == xb::EPI_BIAS_F32 && interior && (p.gin_n == 0 || p.gin_n >= 43)) {
        // interior wave tile: no bounds checks; wave-uniform row bases + one 32-bit lane offset.  Member-major gin
        // (xb_internal.h): row m = t * n + chunk lives at ((t * MB + member) * n + chunk) * 128; the wave's 64 columns lie
        // in one member block.  Its 128 rows start at (t0, c0) and cross into the next time step where c0 + r reaches n
        // (at most three times for n >= 43): each crossing adds (MB - 1) * n rows of 128 floats.  When the rows stay
        // inside one time step (always so for n a multiple of 128) the offsets are compile-time multiples of the stride.
        const int ld = p.gin_n ? 128 : p.ldc;
        float *tile;
        int c0 = 0, n = 1 << 30;
        long long wrap = 0;
        if (p.gin_n) {
            n = p.gin_n;
            const int t0 = mw / n;
            c0 = mw - t0 * n;
            const int MB = p.Nn >> 7;
            tile = p.out_f32 + (((size_t)t0 * MB + (size_t)(nw >> 7)) * n + c0) * 128 + (nw & 127);
            wrap = (long long)(MB - 1) * n * 128;
        } else {
            tile = p.out_f32 + (size_t)mw * p.ldc + nw;
        }
        const int loff = (4 * (lane >> 5)) * ld + (lane & 31);
        if (c0 + 128 <= n) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *rowp = tile + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * ld;
#pragma unroll
                    // non-temporal: 12.6 GB per layer that nobody reads again before the L2 / Infinity Cache have turned over
                    for (int j = 0; j < 2; ++j) __builtin_nontemporal_store(acc[i][j][r] + bj[j], rowp + loff + j * 32);
                }
        } else {
            const int cl = c0 + 4 * (lane >> 5);          // chunk index of this lane's row 0 (before wrapping)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = i * 32 + (r & 3) + 8 * (r >> 2);
                    const int c = cl + rr;
                    const int k = (c >= n) + (c >= 2 * n) + (c >= 3 * n);
                    float *rowp = tile + (size_t)rr * ld + (long long)k * wrap;
#pragma unroll
                    for (int j = 0; j < 2; ++j) __builtin_nontemporal_store(acc[i][j][r] + bj[j], rowp + loff + j * 32);
                }
        }
        return;
    }
    if (EPI == xb::EPI_TANH_SCALE && interior && !p.expand) {
        // interior wave tile of the CRF linear layer without the blank column (the fused path's layout): rows ldc apart, no
        // bounds checks, no per-element column arithmetic
        float *tile = p.out_f32 + (size_t)mw * p.ldc + nw;
        const int loff = (4 * (lane >> 5)) * p.ldc + (lane & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *rowp = tile + (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * p.ldc;
#pragma unroll
                for (int j = 0; j < 2; ++j) rowp[loff + j * 32] = p.scale * fast_tanh(acc[i][j][r] + bj[j]);
            }
        return;
    }
    if (EPI == xb::EPI_SILU_SPLIT && interior) {
        // interior tile of the conv3 GEMM: hi as fp16, second part as fp16 residual or q8 bytes; no bounds checks
        const size_t tile = (size_t)mw * p.ldc + nw;       // element offset of the wave's tile
        const int loff = (4 * (lane >> 5)) * p.ldc + (lane & 31);
        unsigned char *q8base = reinterpret_cast<unsigned char *>(p.out_lo) + tile * 2;    // nw is a multiple of 32
        const int qoff = (4 * (lane >> 5)) * p.ldc * 2 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t ro = (size_t)(i * 32 + (r & 3) + 8 * (r >> 2)) * p.ldc;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const float v = silu(acc[i][j][r] + bj[j]);
                    if (out_q8) {
                        half_t hi;
                        unsigned char h8, l8;
                        q8_bytes(v, p.out_exp, hi, h8, l8);
                        p.out_hi[tile + ro + loff + j * 32] = hi;
                        unsigned char *q = q8base + ro * 2 + qoff + j * 64;
                        q[0] = h8;
                        q[32] = l8;
                    } else {
                        half_t hi, lo;
                        split_f16(v, hi, lo);
                        p.out_hi[tile + ro + loff + j * 32] = hi;
                        p.out_lo[tile + ro + loff + j * 32] = lo;
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = nw + j * 32 + (lane & 31);
            if (n >= p.Nn) continue;
            const float bias = bj[j];
            int ocol = n;
            if (EPI == xb::EPI_TANH_SCALE && p.expand) ocol = (n / p.nb) * (p.nb + 1) + 1 + n % p.nb;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m >= p.M) continue;
                const float v = acc[i][j][r] + bias;
                if (EPI == xb::EPI_BIAS_F32) {
                    p.out_f32[p.gin_n ? xb::gin_offset((size_t)m, n, p.gin_n, p.Nn) : (size_t)m * p.ldc + n] = v;
                } else if (EPI == xb::EPI_SILU_SPLIT) {
                    if (out_q8) {
                        half_t hi;
                        unsigned char h8, l8;
                        q8_bytes(silu(v), p.out_exp, hi, h8, l8);
                        unsigned char *q = reinterpret_cast<unsigned char *>(p.out_lo) + q8_offset((size_t)m, p.ldc, n);
                        p.out_hi[(size_t)m * p.ldc + n] = hi;
                        q[0] = h8;
                        q[32] = l8;
                    } else {
                        half_t hi, lo;
                        split_f16(silu(v), hi, lo);
                        p.out_hi[(size_t)m * p.ldc + n] = hi;
                        p.out_lo[(size_t)m * p.ldc + n] = lo;
                    }
                } else {
                    p.out_f32[(size_t)m * p.ldc + ocol] = p.scale * fast_tanh(v);
                    if (p.expand && n % p.nb == 0) p.out_f32[(size_t)m * p.ldc + ocol - 1] = p.blank;
                }
            }
        }
}

// ======================================================================================
// split-fp16 MFMA GEMM:  D[m][n] = sum_k A[m][k] B[n][k]   (gemm8r_kernel)
// 256x256 block tile, BK = 32, 8 waves (2 in M x 4 in N), each wave 128x64 = 4x2 MFMA 32x32 tiles.
// Products per tile pair and k-tile: NSPLIT 3: six fp16 MFMAs (lo*hi, hi*lo, hi*hi per 16-deep k-step);
// NSPLIT 2: two fp16 MFMAs + ONE block-scaled FP8 MFMA (K = 64) for both correction products (q8 images);
// NSPLIT 1: two fp16 MFMAs.
//  * LDS image per half-tile part: [128 rows][4 cells of 16 B], cell index XOR-swizzled with (row >> 2) & 3 so that every
//    16-lane ds_read_b128 group hits 16 distinct bank slots (SQ_LDS_BANK_CONFLICT = 0).
//  * Operand tiles are staged in four 16-KiB half-tiles per k-tile (SA0, SA1 = block rows 0..127 / 128..255 of A; SB0, SB1
//    likewise for B) through 32 staging registers per lane: loaded from global memory one tile ahead, written to the
//    other LDS buffer a tile later (three half-tiles always in flight; the tile body is branch-free so that the compiler's
//    own vmcnt bookkeeping stays exact).  LDS-DMA staging was measured at the same speed (its issue cost inside a
//    phase that also carries the fragment reads is 100+ cycles per piece) and dropped.
//  * A k-tile is four phases, one 64x32 output quadrant of the wave each:  q0 (A0,B0)  q1 (A0,B1)  q2 (A1,B1)  q3 (A1,B0).
//    Phase = {fragment reads + staging moves} barrier {MFMA cluster} barrier.  Waves 4-7 (wr = 1) run one barrier behind
//    waves 0-3, so on every SIMD one wave is in its MFMA cluster while the other reads LDS / moves staging data
//    (cdna_hip_programming.md "256^2 8-phase template", adapted to 4-byte elements).
//  * Workgroup -> tile order is XCD-aware (gemm_tile_origin), the epilogue stores interior tiles branch-free.
// ======================================================================================
#define G8_BARRIER()                                   \
    do {                                               \
        __builtin_amdgcn_sched_barrier(0);             \
        __builtin_amdgcn_s_barrier();                  \
        __builtin_amdgcn_sched_barrier(0);             \
    } while (0)

template <int EPI, int NSPLIT>
__global__ __launch_bounds__(GTHREADS) void gemm8r_kernel(xb::GemmParams p)
{
    constexpr int NP = NSPLIT == 1 ? 1 : 2;     // operand parts: hi (+ lo, or + the q8 image when NSPLIT == 2)
    constexpr int PARTB = 128 * 64;            // one part (hi or lo) of a half-tile: 128 rows x 32 halves
    constexpr int HTB = NP * PARTB;            // half-tile bytes
    // [4 half-tiles][2 buffers][HTB]: the buffer index is the INNER dimension so that every fragment address of a wave is
    // its one base register plus an immediate below 64 KiB (ds_read offsets are 16 bits)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wid >> 2, wc = wid & 3;
    int m0, n0;
    if (!gemm_tile_origin(p, m0, n0)) return;
    const int nk = p.K / GBK;
    // (Starting the first round of workgroups in four phases spread over a tile time, so that one CU's epilogue burst would
    // overlap its neighbours' main loops, was measured: 49.5 vs 49.4 ms per five GEMMs -- the CUs do not run in lock-step.
    // Exchanging the MFMA operand roles so that a lane holds four consecutive gin columns of one row -- 32 dwordx4 stores
    // per lane instead of 128 single-dword ones -- was measured too: 54.0 ms (plain stores) / 77.9 ms (non-temporal): each
    // 128-byte line is then written in four 32-byte pieces; whole lines per half-wave, as below, are what the memory side wants.)

    // ---- staging source: wave-uniform tile bases (SGPRs) + 32-bit lane byte offsets.  Lane i of wave w moves slot i of rows
    //      16w..16w+15 of a half-tile part: row 16w + (i >> 2), LDS cell (i & 3) <- source cell (i & 3) ^ ((row >> 2) & 3).
    //      Rows past the matrix end are clamped to the last row (their products land in rows / columns never stored).
    unsigned offs[4];
    {
        const int drow = 16 * wid + (lane >> 2), dch = ((lane & 3) ^ ((lane >> 4) & 3)) * 8;
        const int Mrem = p.M - 1 - m0, Nrem = p.Nn - 1 - n0;      // last valid row, relative to the tile
        int r;
        r = drow;        offs[0] = (unsigned)(((r > Mrem ? Mrem : r) * p.lda + dch) * 2);
        r = 128 + drow;  offs[1] = (unsigned)(((r > Mrem ? Mrem : r) * p.lda + dch) * 2);
        r = drow;        offs[2] = (unsigned)(((r > Nrem ? Nrem : r) * p.ldb + dch) * 2);
        r = 128 + drow;  offs[3] = (unsigned)(((r > Nrem ? Nrem : r) * p.ldb + dch) * 2);
    }
    const unsigned char *const tA_hi = reinterpret_cast<const unsigned char *>(p.a_hi + (size_t)m0 * p.lda);
    const unsigned char *const tA_lo = reinterpret_cast<const unsigned char *>(p.a_lo + (size_t)m0 * p.lda);
    const unsigned char *const tB_hi = reinterpret_cast<const unsigned char *>(p.b_hi + (size_t)n0 * p.ldb);
    const unsigned char *const tB_lo = reinterpret_cast<const unsigned char *>(p.b_lo + (size_t)n0 * p.ldb);
    unsigned char *const wdst = smem_raw + wid * 1024;
    // register staging: st[h][part] = this lane's 16 bytes of the wave's piece of half-tile h.  Loaded for tile t+2 at
    // phase (t, h), written to LDS (tile t+1's bytes, loaded one tile earlier) just before that.
    u32x4 st[4][NP];
    unsigned char *const ldst = wdst + lane * 16;
#define G8_LOAD(h, t)                                                                           \
    do {                                                                                        \
        const size_t kb_ = (size_t)(t) * (GBK * 2);                                             \
        st[(h)][0] = *reinterpret_cast<const u32x4 *>(((h) < 2 ? tA_hi : tB_hi) + kb_ + offs[(h)]); \
        if (NP == 2) st[(h)][1] = *reinterpret_cast<const u32x4 *>(((h) < 2 ? tA_lo : tB_lo) + kb_ + offs[(h)]); \
    } while (0)
#define G8_WRITE(h, t)                                                                          \
    do {                                                                                        \
        unsigned char *dst_ = ldst + (((h) * 2 + ((t) & 1)) * HTB);                             \
        *reinterpret_cast<u32x4 *>(dst_) = st[(h)][0];                                          \
        if (NP == 2) *reinterpret_cast<u32x4 *>(dst_ + PARTB) = st[(h)][1];                     \
    } while (0)

    // ---- fragment read offsets (bytes): row (lane & 31) of a 32-row tile, cell (2 ks + (lane >> 5)) ^ ((row >> 2) & 3)
    unsigned la[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
        la[ks] = (unsigned)((lane & 31) * 64 + (((2 * ks + (lane >> 5)) ^ ((lane >> 2) & 3)) * 16));
    const unsigned char *const fragA = smem_raw + wr * 2 * HTB;                              // SA_wr
    const unsigned char *const fragB = smem_raw + (2 + (wc >> 1)) * 2 * HTB + (wc & 1) * 4096;   // 64 rows of SB_(wc/2)

    // q8 image (NSPLIT == 2): a lane reads two cells of its row's 64-byte block -- A: lanes 0-31 the h8 half (cells 0, 1),
    // lanes 32-63 the l8 half (cells 2, 3); B the other way round, so that the block-scaled MFMA (whose k index is a
    // function of (lane half, byte) alone) pairs Ah8 with Bl8 and Al8 with Bh8.
    unsigned lqa[2], lqb[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int h = lane >> 5, sw = (lane >> 2) & 3;
        lqa[j] = (unsigned)((lane & 31) * 64 + (((2 * h + j) ^ sw) * 16));
        lqb[j] = (unsigned)((lane & 31) * 64 + (((2 * (1 - h) + j) ^ sw) * 16));
    }
    const int sca = 127 - p.a_exp - 11, scb = 127 - p.b_exp;      // E8M0 scale bytes: 2^-(a_exp + b_exp + 11) in all

    // A: [tile of the current M half][ks]; B: [ks] of ONE n tile (B0 is read again for the last quadrant rather than kept:
    // the LDS has the headroom, the register file has not)
    half8 ah[2][2], al[2][2], bh[2], bl[2];
    v8i aq[2], bq;                                    // NSPLIT == 2: q8 fragments (one 32-column block = the whole k-tile)
#define G8_READ_A(d, mh)                                                                                  \
    _Pragma("unroll") for (int i2 = 0; i2 < 2; ++i2) {                                                    \
        const unsigned char *t_ = fragA + (d) * HTB + ((mh) * 2 + i2) * 2048;                         \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                \
            ah[i2][ks] = *reinterpret_cast<const half8 *>(t_ + la[ks]);                                   \
            if (NSPLIT == 3) al[i2][ks] = *reinterpret_cast<const half8 *>(t_ + PARTB + la[ks]);          \
        }                                                                                                 \
        if (NSPLIT == 2) {                                                                                \
            const v4i x_ = *reinterpret_cast<const v4i *>(t_ + PARTB + lqa[0]);                           \
            const v4i y_ = *reinterpret_cast<const v4i *>(t_ + PARTB + lqa[1]);                           \
            aq[i2] = __builtin_shufflevector(x_, y_, 0, 1, 2, 3, 4, 5, 6, 7);                             \
        }                                                                                                 \
    }
#define G8_READ_B(d, n)                                                                                   \
    do {                                                                                                  \
        const unsigned char *t_ = fragB + (d) * HTB + (n) * 2048;                                     \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                \
            bh[ks] = *reinterpret_cast<const half8 *>(t_ + la[ks]);                                  \
            if (NSPLIT == 3) bl[ks] = *reinterpret_cast<const half8 *>(t_ + PARTB + la[ks]);         \
        }                                                                                                 \
        if (NSPLIT == 2) {                                                                                \
            const v4i x_ = *reinterpret_cast<const v4i *>(t_ + PARTB + lqb[0]);                           \
            const v4i y_ = *reinterpret_cast<const v4i *>(t_ + PARTB + lqb[1]);                           \
            bq = __builtin_shufflevector(x_, y_, 0, 1, 2, 3, 4, 5, 6, 7);                            \
        }                                                                                                 \
    } while (0)
#define G8_F16(i2, ks, n, A_, B_)                                                                         \
    acc[(mh_) * 2 + (i2)][(n)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[(i2)][(ks)], B_[(ks)], acc[(mh_) * 2 + (i2)][(n)], 0, 0, 0)
#define G8_MFMA(mh, n)                                                                                    \
    do {                                                                                                  \
        constexpr int mh_ = (mh);                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                    \
        if (NSPLIT == 2) {                                                                                \
            G8_F16(0, 0, n, ah, bh);                                                                      \
            G8_F16(1, 0, n, ah, bh);                                                                      \
            _Pragma("unroll") for (int i2 = 0; i2 < 2; ++i2) acc[mh_ * 2 + i2][(n)] =                     \
                __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq[i2], bq, acc[mh_ * 2 + i2][(n)], 0, 0, 0, sca, 0, scb); \
            G8_F16(0, 1, n, ah, bh);                                                                      \
            G8_F16(1, 1, n, ah, bh);                                                                      \
        } else {                                                                                          \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                            \
                if (NSPLIT == 3) {                                                                        \
                    G8_F16(0, ks, n, al, bh);                                                             \
                    G8_F16(1, ks, n, al, bh);                                                             \
                    G8_F16(0, ks, n, ah, bl);                                                             \
                    G8_F16(1, ks, n, ah, bl);                                                             \
                }                                                                                         \
                G8_F16(0, ks, n, ah, bh);                                                                 \
                G8_F16(1, ks, n, ah, bh);                                                                 \
            }                                                                                             \
        }                                                                                                 \
        __builtin_amdgcn_s_setprio(0);                                                                    \
    } while (0)

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // ---- prologue: tile 0 in LDS, tile 1 in flight into the staging registers
#pragma unroll
    for (int h = 0; h < 4; ++h) G8_LOAD(h, 0);
#pragma unroll
    for (int h = 0; h < 4; ++h) G8_WRITE(h, 0);
#pragma unroll
    for (int h = 0; h < 4; ++h) G8_LOAD(h, nk > 1 ? 1 : 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G8_BARRIER();
    if (wr == 1) G8_BARRIER();          // the second wave group runs one barrier behind the first

    // phase (t, q): fragment reads of quadrant q; write half-tile q of tile t+1 into the other buffer (its last reads were
    // in tile t-1); reload the staging registers with half-tile q of tile t+2; retire the LDS operations; barrier.
#define G8_MOVE(q, t)                                                                       \
    do {                                                                                    \
        G8_WRITE(q, (t) + 1);                                                               \
        __builtin_amdgcn_sched_barrier(0);      /* reload the SAME registers after the write */ \
        G8_LOAD(q, (t) + 2 < nk ? (t) + 2 : nk - 1);                                        \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                  \
    } while (0)
#define G8_TILE(d, t)                                                                       \
    do {                                                                                    \
        G8_READ_B(d, 0);                                                                    \
        G8_READ_A(d, 0);                                                                    \
        G8_MOVE(0, t);                                                                      \
        G8_BARRIER();                                                                       \
        G8_MFMA(0, 0);                                                                      \
        G8_BARRIER();                                                                       \
        G8_READ_B(d, 1);                                                                    \
        G8_MOVE(1, t);                                                                      \
        G8_BARRIER();                                                                       \
        G8_MFMA(0, 1);                                                                      \
        G8_BARRIER();                                                                       \
        G8_READ_A(d, 1);                                                                    \
        G8_MOVE(2, t);                                                                      \
        G8_BARRIER();                                                                       \
        G8_MFMA(1, 1);                                                                      \
        G8_BARRIER();                                                                       \
        G8_READ_B(d, 0);                                                                    \
        G8_MOVE(3, t);                                                                      \
        G8_BARRIER();                                                                       \
        G8_MFMA(1, 0);                                                                      \
        G8_BARRIER();                                                                       \
    } while (0)

    // The tile body is branch-free so that the compiler's vmcnt bookkeeping stays exact (three half-tiles in flight):
    // past the end it re-loads the last tile (L2 hits) and writes it into the buffer nobody reads any more.
    int t = 0;
    for (; t + 1 < nk; t += 2) {
        G8_TILE(0, t);
        G8_TILE(1, t + 1);
    }
    if (t < nk) G8_TILE(0, t);
    if (wr == 0) G8_BARRIER();          // matches the extra barrier of the second group
#undef G8_MOVE
#undef G8_LOAD
#undef G8_WRITE
#undef G8_TILE
#undef G8_MFMA
#undef G8_F16
#undef G8_READ_A
#undef G8_READ_B

    // the epilogue's per-lane indices must not be computed (and kept in registers) ahead of the main loop
    int lane_e = lane, mw_e = m0 + wr * 128, nw_e = n0 + wc * 64;
    asm volatile("" : "+v"(lane_e), "+s"(mw_e), "+s"(nw_e));       // (the tile origin too: its divisions belong behind the loop)
    gemm_epilogue<EPI>(p, acc, mw_e, nw_e, lane_e);
}

// ======================================================================================
// gemm4p_kernel: the product's GEMM (round 3).  The same contraction as gemm8r_kernel, bit for bit (per accumulator the
// products arrive in the same order), with TWO workgroups per CU:
//   * 128 (M) x 256 (N) block tile, BK = 32, 4 waves side by side in N, each wave 128 x 64 = 4 x 2 MFMA 32x32 tiles.  Two
//     independent 4-wave workgroups share a CU (2 waves per SIMD, 256 registers each, 48 KiB of LDS each): a tile is only
//     24 k-tiles deep here (K = 768), so prologue and epilogue are a third of a workgroup's life, and one workgroup's
//     runs under the other's MFMA clusters; the hardware interleaves the two main loops.
//   * B (the weights) never touches LDS.  With the waves side by side in N no two waves share a B row, so each wave
//     loads its own fragments straight into registers from a FRAGMENT-MAJOR weight image built once on the host
//     (xb_api.hip: fragment_major): [k-tile][32-row block][piece][lane][16 B], i.e. every wave-instruction reads 1 KiB of
//     consecutive bytes.  LDS carries only the A tile (shared by the four waves): 16 ds_read_b128 per wave and k-tile
//     instead of 28, no ds_write at all.
//   * TWO k-tiles of both operands are in flight (24 KiB per wave):
//       A by LDS-DMA (global_load_lds_dwordx4, no staging registers) into a ring of three stages, two tiles ahead;
//       B in two register sets (even / odd k-tile, the loop is unrolled by two), every piece reloaded for tile t + 2 right
//       behind its last MFMA of tile t.  These loads are inline asm so that hipcc neither counts them nor drains the
//       LDS-DMAs for them (cdna_hip_programming.md 5, trap (b)); every wait is a counted s_waitcnt written here:
//         per wave and k-tile NA LDS-DMAs, then the B pieces in groups g0, g1(, g2); issue order A(t) B(t) A(t+1) B(t+1) ..
//         top of tile t (A(t) landed):      vmcnt(2 NB + NA)        -- everything younger than A(t) may be in flight
//         before the MFMAs of B group k:    vmcnt(2 NB + 2 NA - gk) -- the gk oldest are that group
//       (loads return in order; past the last tile the same tiles are loaded again, so the counts never change).
//   * the A fragments are read one half sub-phase ahead (32 registers live), which is what makes room for the second B
//     set: 128 accumulators + 64 B + 32 A.
//   * the member-major gin epilogue takes any batch: a wave's 128 rows may cross into the next time step (gemm_epilogue).
// What bounds it (round 3 measurements): with 4-byte operands (fp16 + q8 image) a 256 x 256 output patch
// per CU needs 64-96 KiB from L2 per k-tile against 2048 MFMA-pipe cycles -- 32-47 B/clk, which IS what a CU's vector
// memory path delivers (~70 GB/s per CU from L2, MI355X_MICROARCH.md); ablated builds without the loop's loads and the
// epilogue's stores ran at the MFMA rate (5.0 ms per LSTM-input GEMM), each of the two costs ~2.2 ms and the two add up.
// De-phasing the workgroups' starts, doubling the loads in flight (this kernel vs its one-tile-deep predecessor: 46.7 vs
// 47.9 ms per five GEMMs) and the workgroup shape (one 256 x 256 workgroup per CU, gemm8r: 50 ms) move it by a few percent:
// the bytes per element are the lever that is left.
// ======================================================================================
#ifndef XB_GEMM_LATE_A           // 1: (three- and one-product kernels) a k-tile's LDS-DMA requests behind its first MFMA group (A/B builds)
#define XB_GEMM_LATE_A 1
#endif
#ifndef XB_GEMM_DMA_ASM          // 1: gemm4p_kernel's A-tile LDS-DMA requests as inline asm (see dma_a); 0: the builtin (A/B builds)
#define XB_GEMM_DMA_ASM 1
#endif
constexpr int G4_BM = 128, G4_BN = 256, G4_THREADS = 256;

__device__ __forceinline__ bool gemm4_tile_origin(const xb::GemmParams &p, int &m0, int &n0)
{
    const int MT = (p.M + G4_BM - 1) / G4_BM, NT = (p.Nn + G4_BN - 1) / G4_BN;
    const int snw = p.sn > 0 ? p.sn : gemm_super_n(NT);
    const int SN = NT < snw ? NT : snw, SM = 64 / SN;        // 64 workgroups per XCD side by side (two per CU)
    const int ngroups = (NT + SN - 1) / SN, msup = (MT + SM - 1) / SM;
    const int b = blockIdx.x, xcd = b & 7, j = b >> 3;
    const int q = (j >> 6) * 8 + xcd, w = j & 63;            // super-tile id, slot in it
    if (q >= msup * ngroups || w >= SM * SN) return false;
    // neighbouring slots share an A panel (giving it to slots SM apart instead measured 121.8 vs 118.9 ms per step)
    const int mt = (q / ngroups) * SM + w / SN, nt = (q % ngroups) * SN + w % SN;
    if (mt >= MT || nt >= NT) return false;
    m0 = mt * G4_BM;
    n0 = nt * G4_BN;
    return true;
}

template <int EPI, int NSPLIT>
__global__ __launch_bounds__(G4_THREADS, 2) void gemm4p_kernel(xb::GemmParams p)
{
    constexpr int NPA = NSPLIT == 1 ? 1 : 2;   // A parts in LDS: hi (+ lo, or the q8 image)
    constexpr int NPC = NSPLIT == 1 ? 2 : 4;   // B pieces (16 B per lane) per 32-row block and k-tile
    constexpr int PARTB = 128 * 64;            // one A part of a stage: 128 rows x 32 halves
    constexpr int STB = NPA * PARTB;           // stage bytes
    constexpr int NA = 2 * NPA, NB = 2 * NPC;  // LDS-DMAs / B loads per wave and k-tile
    constexpr int INFL = 2 * NB + 2 * NA;      // loads in flight once a tile's DMAs are issued
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];      // [3 stages][STB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    int m0, n0;
    if (!gemm4_tile_origin(p, m0, n0)) return;
    const int nk = p.K / GBK;

    // ---- A: lane i of wave w fills, per part, LDS cells of rows 16 w + (i >> 2) and 64 + that; cell (i & 3) of a row holds
    //      source cell (i & 3) ^ ((row >> 2) & 3).  Rows past the matrix end are clamped to the last row.
    unsigned offs[2];
    {
        const int r0 = tid >> 2, sc = ((tid & 3) ^ ((tid >> 4) & 3)) * 16;
        const int Mrem = p.M - 1 - m0;
        offs[0] = (unsigned)((r0 > Mrem ? Mrem : r0) * p.lda * 2 + sc);
        offs[1] = (unsigned)((r0 + 64 > Mrem ? Mrem : r0 + 64) * p.lda * 2 + sc);
    }
    const unsigned char *const tA_hi = reinterpret_cast<const unsigned char *>(p.a_hi + (size_t)m0 * p.lda);
    const unsigned char *const tA_lo = reinterpret_cast<const unsigned char *>(p.a_lo + (size_t)m0 * p.lda);
    // (XB_GEMM_DMA_ASM: the requests as inline asm -- wave-uniform base in an SGPR pair, 32-bit lane offset -- so that hipcc does not
    //  book them as LDS events: with the builtin every wait for an A fragment in the k loop is lgkmcnt(0), i.e. all eight fragment
    //  reads of a k-step are waited for before its first MFMA; hidden from the compiler it counts, and the MFMAs on the first two
    //  row tiles start while the other two tiles' fragments are still in flight)
    auto dma_a = [&](int t, int stage) {
        const size_t kb = (size_t)t * (GBK * 2);
        unsigned char *dst = smem_raw + stage * STB + wid * 1024;
#pragma unroll
        for (int part = 0; part < NPA; ++part)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#if XB_GEMM_DMA_ASM
                const unsigned m0v = __builtin_amdgcn_readfirstlane(
                    (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(dst + part * PARTB + h * 4096));
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             ::"v"(offs[h]), "s"((part ? tA_lo : tA_hi) + kb), "s"(m0v) : "memory", "m0");
#else
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)((part ? tA_lo : tA_hi) + kb + offs[h]),
                    (__attribute__((address_space(3))) void *)(dst + part * PARTB + h * 4096), 16, 0, 0);
#endif
            }
    };

    // ---- B: this wave's two 32-row blocks; lane byte offsets for block 0 / 1 (the pieces are immediates)
    const unsigned boff = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)(n0 + wid * 64) >> 5) * NPC * 1024u));
    const unsigned char *const tB = p.b4 + boff;       // wave-uniform by construction: an SGPR pair for the asm loads
    const size_t bks = p.b4_kstride;
    const unsigned voff0 = (unsigned)lane * 16, voff1 = voff0 + NPC * 1024;
    u32x4 bE[2][NPC], bO[2][NPC];               // even / odd k-tile
    // (default cache policy: with the nt bit -- the weights bypassing the L1 -- the five input GEMMs take 65 instead of 47 ms)
#define G4P_LDB(dst, base, j, pc)                                                               \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"((j) ? voff1 : voff0), "s"(base), "i"((pc) * 1024))
#define G4P_LDB_GROUP(bS, base, pc)                                                             \
    do {                                                                                        \
        G4P_LDB(bS[0][(pc)], base, 0, pc);                                                      \
        G4P_LDB(bS[1][(pc)], base, 1, pc);                                                      \
    } while (0)
    // counted waits that also make the named registers "written here" for the compiler
#define G4P_WAIT2(n, a, b) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "i"(n))
#define G4P_WAIT4(n, a, b, c, d) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "i"(n))

    // ---- A fragment read offsets (bytes) inside a part: row (lane & 31) of a 32-row tile, cell c ^ ((row >> 2) & 3)
    const int hs = lane >> 5, sw = (lane >> 2) & 3;
    unsigned la[2], lqa[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        la[j] = (unsigned)((lane & 31) * 64 + (((2 * j + hs) ^ sw) * 16));       // fp16 k-step j
        lqa[j] = (unsigned)((lane & 31) * 64 + (((2 * hs + j) ^ sw) * 16));      // q8 image, A role: lanes 0-31 h8, 32-63 l8
    }
    const int sca = 127 - p.a_exp - 11, scb = 127 - p.b_exp;      // E8M0 scale bytes: 2^-(a_exp + b_exp + 11) in all

    floatx16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

#define G4P_SB() __builtin_amdgcn_sched_barrier(0)
    constexpr bool G4P_LATE_A = XB_GEMM_LATE_A != 0 && NSPLIT != 2;
    // fragments of two 32-row tiles (ih = 0: rows 0..63, ih = 1: rows 64..127) of one part and k-step / of the q8 image
#define G4P_RD_H(d, sa, part, ih, ks)                                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                            \
        d[i_] = *reinterpret_cast<const half8 *>((sa) + (part) * PARTB + ((ih) * 2 + i_) * 2048 + la[(ks)])
#define G4P_RD_Q(d, sa, ih)                                                                     \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_) {                                          \
        const v4i x_ = *reinterpret_cast<const v4i *>((sa) + PARTB + ((ih) * 2 + i_) * 2048 + lqa[0]); \
        const v4i y_ = *reinterpret_cast<const v4i *>((sa) + PARTB + ((ih) * 2 + i_) * 2048 + lqa[1]); \
        d[i_] = __builtin_shufflevector(x_, y_, 0, 1, 2, 3, 4, 5, 6, 7);                        \
    }
#define G4P_F16(A_, ih, bS, pc)                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                        \
            acc[(ih) * 2 + i_][j_] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[i_], __builtin_bit_cast(half8, bS[j_][(pc)]), \
                                                                          acc[(ih) * 2 + i_][j_], 0, 0, 0)
#define G4P_F8(A_, ih, bS)                                                                      \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_) {                                      \
            const v8i bq_ = __builtin_shufflevector(__builtin_bit_cast(v4i, bS[j_][2]), __builtin_bit_cast(v4i, bS[j_][3]), \
                                                    0, 1, 2, 3, 4, 5, 6, 7);                    \
            acc[(ih) * 2 + i_][j_] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A_[i_], bq_, acc[(ih) * 2 + i_][j_], \
                                                                                   0, 0, 0, sca, 0, scb); \
        }
#define G4P_MFMA_BEGIN() do { G4P_SB(); __builtin_amdgcn_s_setprio(1); } while (0)
#define G4P_MFMA_END() do { __builtin_amdgcn_s_setprio(0); G4P_SB(); } while (0)

    // one k-tile: tile t from LDS stage `cur` and register set bS; DMA of tile t + 2 into stage `nxt2`; reload of bS with
    // tile t + 2 (base address b2)
#define G4P_TILE(bS, t)                                                                         \
    do {                                                                                        \
        const unsigned char *const sa = smem_raw + cur * STB;                                   \
        const int t2_ = (t) + 2 < nk ? (t) + 2 : nk - 1;                                        \
        const unsigned char *const b2 = tB + (size_t)t2_ * bks;                                 \
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * NB + NA) : "memory");                      \
        G4P_SB();                                                                               \
        __builtin_amdgcn_s_barrier();                                                           \
        G4P_SB();                                                                               \
        if constexpr (!G4P_LATE_A) dma_a(t2_, nxt2);                                            \
        if constexpr (NSPLIT == 2) {                                                            \
            /* fragments of the NEXT group are requested ahead of the last four MFMAs of the current one: hipcc forgets its */ \
            /* lgkmcnt bookkeeping at every asm statement and waits lgkmcnt(0) behind it, which is free once they landed */ \
            half8 h0[2], h1[2], g0[2], g1[2];                                                   \
            v8i q0[2], q1[2];                                                                   \
            G4P_RD_H(h0, sa, 0, 0, 0);                                                          \
            G4P_RD_H(h1, sa, 0, 1, 0);                                                          \
            G4P_WAIT2(INFL - 2, bS[0][0], bS[1][0]);                                            \
            G4P_MFMA_BEGIN();                                                                   \
            G4P_F16(h0, 0, bS, 0);                                                              \
            G4P_MFMA_END();                                                                     \
            G4P_RD_Q(q0, sa, 0);                                                                \
            G4P_RD_Q(q1, sa, 1);                                                                \
            G4P_MFMA_BEGIN();                                                                   \
            G4P_F16(h1, 1, bS, 0);                                                              \
            G4P_MFMA_END();                                                                     \
            G4P_LDB_GROUP(bS, b2, 0);                                                           \
            G4P_WAIT4(INFL - 4, bS[0][2], bS[0][3], bS[1][2], bS[1][3]);                        \
            G4P_MFMA_BEGIN();                                                                   \
            G4P_F8(q0, 0, bS);                                                                  \
            G4P_MFMA_END();                                                                     \
            G4P_RD_H(g0, sa, 0, 0, 1);                                                          \
            G4P_RD_H(g1, sa, 0, 1, 1);                                                          \
            G4P_MFMA_BEGIN();                                                                   \
            G4P_F8(q1, 1, bS);                                                                  \
            G4P_MFMA_END();                                                                     \
            G4P_LDB_GROUP(bS, b2, 2);                                                           \
            G4P_LDB_GROUP(bS, b2, 3);                                                           \
            G4P_WAIT2(INFL - 2, bS[0][1], bS[1][1]);                                            \
            G4P_MFMA_BEGIN();                                                                   \
            G4P_F16(g0, 0, bS, 1);                                                              \
            G4P_F16(g1, 1, bS, 1);                                                              \
            G4P_MFMA_END();                                                                     \
            G4P_LDB_GROUP(bS, b2, 1);                                                           \
        } else {                                                                                \
            /* NSPLIT 3: per k-step lo*hi, hi*lo, hi*hi (pieces: 0, 1 = hi of k-step 0, 1; 2, 3 = lo); NSPLIT 1: hi*hi.         */ \
            /* The A fragments of the two row-tile pairs (ih = 0: rows 0..63, ih = 1: rows 64..127) are software-pipelined by  */ \
            /* half k-steps (round 4): the reads for (ks + 1, ih) go out right behind the MFMAs of (ks, ih) and have the other  */ \
            /* pair's twelve MFMAs to land; only the tile's first reads -- behind the barrier -- are waited for.  The order of */ \
            /* the vector-memory instructions (hence every counted vmcnt) is unchanged.                                         */ \
            half8 ah0[2], al0[2], ah1[2], al1[2];                                               \
            G4P_RD_H(ah0, sa, 0, 0, 0);                                                         \
            if (NSPLIT == 3) G4P_RD_H(al0, sa, 1, 0, 0);                                        \
            G4P_RD_H(ah1, sa, 0, 1, 0);                                                         \
            if (NSPLIT == 3) G4P_RD_H(al1, sa, 1, 1, 0);                                        \
            _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                  \
                /* (G4P_LATE_A: the tile's LDS-DMA requests go out behind its first twelve MFMAs instead of in front of them, */ \
                /*  so the first wait has NA fewer younger operations)                                                        */ \
                constexpr int LA_ = G4P_LATE_A ? NA : 0;                                        \
                if (NSPLIT == 3) {                                                              \
                    if (ks == 0) G4P_WAIT4(INFL - 4 - LA_, bS[0][0], bS[1][0], bS[0][2], bS[1][2]); \
                    else G4P_WAIT4(INFL - 4, bS[0][1], bS[1][1], bS[0][3], bS[1][3]);           \
                } else {                                                                        \
                    if (ks == 0) G4P_WAIT2(INFL - 2 - LA_, bS[0][0], bS[1][0]);                 \
                    else G4P_WAIT2(INFL - 2, bS[0][1], bS[1][1]);                               \
                }                                                                               \
                G4P_MFMA_BEGIN();                                                               \
                if (NSPLIT == 3) {                                                              \
                    G4P_F16(al0, 0, bS, ks);                                                    \
                    G4P_F16(ah0, 0, bS, 2 + ks);                                                \
                }                                                                               \
                G4P_F16(ah0, 0, bS, ks);                                                        \
                G4P_MFMA_END();                                                                 \
                if (ks == 0) {                                                                  \
                    if constexpr (G4P_LATE_A) dma_a(t2_, nxt2);                                 \
                    G4P_RD_H(ah0, sa, 0, 0, 1);                                                 \
                    if (NSPLIT == 3) G4P_RD_H(al0, sa, 1, 0, 1);                                \
                }                                                                               \
                G4P_MFMA_BEGIN();                                                               \
                if (NSPLIT == 3) {                                                              \
                    G4P_F16(al1, 1, bS, ks);                                                    \
                    G4P_F16(ah1, 1, bS, 2 + ks);                                                \
                }                                                                               \
                G4P_F16(ah1, 1, bS, ks);                                                        \
                G4P_MFMA_END();                                                                 \
                if (ks == 0) {                                                                  \
                    G4P_RD_H(ah1, sa, 0, 1, 1);                                                 \
                    if (NSPLIT == 3) G4P_RD_H(al1, sa, 1, 1, 1);                                \
                    G4P_LDB_GROUP(bS, b2, 0);                                                   \
                    if (NSPLIT == 3) G4P_LDB_GROUP(bS, b2, 2);                                  \
                } else {                                                                        \
                    G4P_LDB_GROUP(bS, b2, 1);                                                   \
                    if (NSPLIT == 3) G4P_LDB_GROUP(bS, b2, 3);                                  \
                }                                                                               \
            }                                                                                   \
        }                                                                                       \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                      \
        { const int c_ = cur; cur = nxt1; nxt1 = nxt2; nxt2 = c_; }                             \
    } while (0)

    // ---- a wave whose 64 columns lie wholly in the padding behind the matrix (the CRF linear layer: 1296 = 5 x 256 + 16 columns,
    //      i.e. three of the last N tile's four waves) has no products to add and nothing to store: it only keeps up its share of
    //      the A tile's LDS-DMAs and the tile barriers -- the MFMA pipe and the weight loads it would have taken go to the CU's
    //      other workgroup (round 4).
    if (__builtin_amdgcn_readfirstlane(n0 + wid * 64 >= p.Nn ? 1 : 0)) {
        dma_a(0, 0);
        dma_a(nk > 1 ? 1 : 0, 1);
        int c0 = 0, c1 = 1, c2 = 2;
#pragma unroll 1
        for (int t = 0; t < nk; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NA) : "memory");      // A(t) landed; A(t + 1) may be in flight
            __builtin_amdgcn_s_barrier();
            dma_a(t + 2 < nk ? t + 2 : nk - 1, c2);
            const int c_ = c0; c0 = c1; c1 = c2; c2 = c_;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // ---- prologue: A(0) B(0) A(1) B(1) in the loop's issue order
    int cur = 0, nxt1 = 1, nxt2 = 2;
    {
        const int t1 = nk > 1 ? 1 : 0;
        const unsigned char *const b0 = tB, *const b1 = tB + (size_t)t1 * bks;
        dma_a(0, 0);
        if constexpr (NSPLIT == 2) {
            G4P_LDB_GROUP(bE, b0, 0); G4P_LDB_GROUP(bE, b0, 2); G4P_LDB_GROUP(bE, b0, 3); G4P_LDB_GROUP(bE, b0, 1);
        } else {
            G4P_LDB_GROUP(bE, b0, 0); if (NSPLIT == 3) G4P_LDB_GROUP(bE, b0, 2);
            G4P_LDB_GROUP(bE, b0, 1); if (NSPLIT == 3) G4P_LDB_GROUP(bE, b0, 3);
        }
        dma_a(t1, 1);
        if constexpr (NSPLIT == 2) {
            G4P_LDB_GROUP(bO, b1, 0); G4P_LDB_GROUP(bO, b1, 2); G4P_LDB_GROUP(bO, b1, 3); G4P_LDB_GROUP(bO, b1, 1);
        } else {
            G4P_LDB_GROUP(bO, b1, 0); if (NSPLIT == 3) G4P_LDB_GROUP(bO, b1, 2);
            G4P_LDB_GROUP(bO, b1, 1); if (NSPLIT == 3) G4P_LDB_GROUP(bO, b1, 3);
        }
    }
    int t = 0;
#pragma unroll 1
    for (; t + 1 < nk; t += 2) {
        G4P_TILE(bE, t);
        G4P_TILE(bO, t + 1);
    }
    if (t < nk) G4P_TILE(bE, t);
    // drain the look-ahead loads (they target registers and LDS stages that are dead, but must have landed before reuse)
    if constexpr (NSPLIT == 1)
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bE[0][0]), "+v"(bE[1][0]), "+v"(bE[0][1]), "+v"(bE[1][1]),
                       "+v"(bO[0][0]), "+v"(bO[1][0]), "+v"(bO[0][1]), "+v"(bO[1][1]) :: "memory");
    else
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(bE[0][0]), "+v"(bE[1][0]), "+v"(bE[0][1]), "+v"(bE[1][1]),
                       "+v"(bE[0][2]), "+v"(bE[1][2]), "+v"(bE[0][3]), "+v"(bE[1][3]),
                       "+v"(bO[0][0]), "+v"(bO[1][0]), "+v"(bO[0][1]), "+v"(bO[1][1]),
                       "+v"(bO[0][2]), "+v"(bO[1][2]), "+v"(bO[0][3]), "+v"(bO[1][3]) :: "memory");
    G4P_SB();
#undef G4P_TILE
#undef G4P_LDB
#undef G4P_LDB_GROUP
#undef G4P_WAIT2
#undef G4P_WAIT4
#undef G4P_RD_H
#undef G4P_RD_Q
#undef G4P_F16
#undef G4P_F8
#undef G4P_MFMA_BEGIN
#undef G4P_MFMA_END
#undef G4P_SB

    int lane_e = lane, mw_e = m0, nw_e = n0 + wid * 64;
    asm volatile("" : "+v"(lane_e), "+s"(mw_e), "+s"(nw_e));
    gemm_epilogue<EPI>(p, acc, mw_e, nw_e, lane_e);
}

// ======================================================================================
// LSTM recurrence.
// A *group* = LG_BN chunks; its F/32 member workgroups each own 32 hidden units (128
// gate-interleaved rows of W_hh: wave w holds rows [32w, 32w+32) = units 8w..8w+7 as MFMA
// A-operand fragments in registers for the whole launch).  Per step a member needs the whole
// h_{t-1} of its group's chunks.  The members exchange h through a small ping-pong buffer
// xh[parity][group][part][64 chunks][F] (a few MB, L2/Infinity-Cache resident) and, beside it,
// write the layer output y (T,N,F) that the next layer consumes after the launch.
// The exchange rows are pulled into LDS by LDS-DMA in K pieces of KP columns, double buffered
// against the MFMAs, B fragments software-pipelined one k-step ahead.
// MFMA tile orientation: rows = gate rows (one lane owns i,f,g,o of a unit in four consecutive
// accumulator registers), columns = chunks.
// persistent = 1: all steps in one launch.  Hand-off protocol per step (cdna_hip_programming.md
// Guideline 16, form R1): exchange stores are 16-byte write-through (sc1) stores; every storing
// wave drains them (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane adds to the group's
// monotonic agent-scope counter.  Consumer: ONE lane polls that counter with relaxed sc1 loads
// (bounded spin), workgroup barrier, then EVERY load of the exchanged bytes is an sc1 load
// (LDS-DMA with the sc1 bit), so no L1 line can be stale and no fence is needed.
// ======================================================================================
#ifndef XB_LSTM_DMA_ASM          // 1: the exchange pieces' LDS-DMA requests as inline asm (see dma16_sc1); 0: the builtin (A/B builds)
#define XB_LSTM_DMA_ASM 1
#endif
constexpr int LG_BN = 64;        // chunks per group (2 MFMA column tiles)
constexpr int LG_UNITS = 32;     // hidden units per member workgroup
constexpr int LG_SYNC = 64;      // words between the counter slots of consecutive 64-chunk groups (lstm_quad_kernel's 32-chunk groups: 32)
constexpr unsigned long long LG_SPIN_CYCLES = 4000000000ull;   // ~2 s at 2 GHz
constexpr int CPOL_SC1 = 16;     // gfx940+ cache-policy immediate: sc0 = 1, nt = 2, sc1 = 16
constexpr int ST_LD = 68;        // dword stride of one unit-pair row of the h staging (64 chunks + 4: 2-way reads)

// (Inline asm, not __builtin_amdgcn_global_load_lds: hipcc books an LDS-DMA as an LDS event of the lgkm counter, and with two
// kinds of events pending it can no longer count -- every wait for a B fragment in the MFMA loop became lgkmcnt(0), i.e. the
// fragment reads issued one k-step AHEAD were waited for at once and their latency (~120 cycles per k-step pair, ~3 k cycles per
// group-step) sat on the critical path.  Hidden from the compiler, the requests leave its bookkeeping alone and it emits the
// counted waits the software pipeline needs; completion is the explicit s_waitcnt vmcnt(0) + barrier that closes every piece.)
__device__ __forceinline__ void dma16_sc1(const void *g, void *lds_wave_base)
{
#if XB_LSTM_DMA_ASM
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" ::"v"(g), "s"(m0v) : "memory", "m0");
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, CPOL_SC1);
#endif
}
// the same with a wave-uniform base (an SGPR pair) and a 32-bit per-lane byte offset: no 64-bit address arithmetic per request
__device__ __forceinline__ void dma16_sc1_off(const void *ubase, int byte_off, void *lds_wave_base)
{
#if XB_LSTM_DMA_ASM
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(byte_off), "s"(ubase), "s"(m0v) : "memory", "m0");
#else
    dma16_sc1(reinterpret_cast<const unsigned char *>(ubase) + byte_off, lds_wave_base);
#endif
}

// Round 5: the LEAN request.  The kernel is bound by its one wave's instruction COUNT (every instruction, scalar ones included,
// takes an issue slot of >= 4 cycles), and a request used to cost seven: 64-bit base add (2), generic -> LDS pointer cast (2: a
// null check), s_mov m0, s_nop, the load.  Here it costs three: the wave's LDS base (an SGPR) plus a literal straight into m0, the
// lane's byte offset plus a literal (the VALU add doubles as the wait state m0 needs before an LDS-DMA), and the load from ONE
// wave-uniform base (SGPR pair, per group-step); the immediate offset stays 0 (it would move the LDS address as well).  The
// literals are "n" operands: the callers' loop variables are constants after unrolling.
__device__ __forceinline__ void dma16_lean_sc1(unsigned vlane, int vconst, const void *sbase, unsigned lds_wave, int lconst, int ioff)
{
    unsigned t;
    asm volatile("s_add_i32 m0, %[lb], %[lc]\n\tv_add_u32 %[t], %[vc], %[vo]\n\tglobal_load_lds_dwordx4 %[t], %[sb] offset:%[io] sc1"
                 : [t] "=&v"(t) : [lb] "s"(lds_wave), [lc] "n"(lconst), [vc] "n"(vconst), [vo] "v"(vlane), [sb] "s"(sbase), [io] "n"(ioff)
                 : "memory", "m0");
}
__device__ __forceinline__ void dma16_lean_nt(unsigned vlane, int vconst, const void *sbase, unsigned lds_wave, int lconst)
{
    unsigned t;
    asm volatile("s_add_i32 m0, %[lb], %[lc]\n\tv_add_u32 %[t], %[vc], %[vo]\n\tglobal_load_lds_dwordx4 %[t], %[sb] nt"
                 : [t] "=&v"(t) : [lb] "s"(lds_wave), [lc] "n"(lconst), [vc] "n"(vconst), [vo] "v"(vlane), [sb] "s"(sbase)
                 : "memory", "m0");
}

// 16-byte plain store: the line stays in the XCD's L2 (same asm form as the write-through one below)
__device__ __forceinline__ void store16_l2(void *g, uint4 v)
{
    const u32x4 d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(g), "v"(d) : "memory");
}

// 16-byte write-through store (the `s_nop 1` keeps the data registers intact until the store has read them)
__device__ __forceinline__ void store16_sc1(void *g, uint4 v)
{
    const u32x4 d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(g), "v"(d) : "memory");
}

#ifdef XB_LSTM_STAMPS
// diagnostic build only: per-phase cycle sums of workgroup 0 (never compiled into the product).  The sums are kept
// in LDS (no vector-memory traffic, so the stamps neither drain vmcnt nor absorb store latencies) and copied out
// once at the end of the kernel.
__device__ unsigned long long g_lstm_stamps[10];   // 0..7 cycle sums, 8 = early first-piece requests, 9 = group-steps
#define XB_STAMP(i)                                                                         \
    do {                                                                                    \
        const unsigned long long now_ = __builtin_readcyclecounter();                       \
        if (tid == 0) sStamp[(i)] += now_ - stamp_prev;                                      \
        stamp_prev = now_;                                                                  \
    } while (0)
#else
#define XB_STAMP(i) do { } while (0)
#endif

// DUAL = true: one workgroup serves TWO groups (g and g + gh) alternately with the same W_hh registers.  A group-step's
// hand-off (stores reaching L2, the other members' arrivals, the poll) then completes while the workgroup runs the other
// group's step, the first h piece of the coming group-step is requested before the gate math of the current one, and
// the gin tile is requested a whole group-step ahead: a launch holds twice the chunks at the same residency.
#ifndef XB_LSTM_DMA_SPREAD       // 1: every LDS-DMA request of the MFMA loop directly behind one MFMA; 0 (default): in pairs behind the k-step's
                                 // MFMAs -- measured on one box (profiles/r04_lstm_loop_ab.txt): no gain on top of XB_LSTM_DMA_ASM, a loss with one group per workgroup
#define XB_LSTM_DMA_SPREAD 0
#endif
#ifndef XB_LSTM_DEFER_ARRIVE     // 1 (default): two groups per workgroup -- the arrival of a group-step is issued behind the first
                                 // piece-closing drain + barrier of the OTHER group's step instead of behind a drain of its own (A/B builds: 0)
#define XB_LSTM_DEFER_ARRIVE 1
#endif
#ifndef XB_LSTM_PIPE_PIECES      // 1: the k-step pipeline runs across the piece boundary (see PIPE in lstm_kernel); with two piece buffers it measured
                                 // 4 % SLOWER (profiles/r05_lstm_lean_ab.txt: the piece is then closed a k-step earlier and waits longer for its successors DMA)
#define XB_LSTM_PIPE_PIECES 0
#endif
#ifdef XB_NO_SIGNAL
#define XB_SIG(x) false
#else
#define XB_SIG(x) (x)
#endif
// YALT = true (NSPLIT 2 or 3 only): the layer output y carries the OTHER second part than the exchange image -- the fp16
// residual when the recurrence itself runs on q8 images (NSPLIT 2), the q8 image when it runs on residuals (NSPLIT 3) -- because
// the GEMM that consumes y runs in the other arithmetic (mixed-precision encoders: xb_api.hip stage_nsplit).  The extra
// staging rows live in piece buffer 1, which is idle between the last piece's closing barrier and the next group-step's
// piece-1 requests.
template <int KS, int NSPLIT, bool DUAL, bool YALT = false>
__global__ __launch_bounds__(256) void lstm_kernel(xb::LstmParams p)
{
    static_assert(!YALT || NSPLIT == 2 || NSPLIT == 3, "an alternative y image exists for the q8 and the residual arithmetic only");
    constexpr int F = KS * 16;
    constexpr int KP = F < 128 ? F : 128;       // columns per piece
    constexpr int NP = F / KP;                  // pieces per step
    constexpr int KSP = KP / 16;                // MFMA k-steps per piece
    // NSPLIT == 4: the int8-limb recurrence.  h (|h| < 1) and each W_hh row (per-row scale) are 16-bit fixed point, split
    // into two balanced signed 8-bit digits q = 256 d1 + d0; the four digit products run on v_mfma_i32_32x32x32_i8 (exact
    // int32 sums, three accumulator sets by weight 2^16 / 2^8 / 1) and are combined in fp32 once per step.  The exchange
    // image is one byte per element and part (part 0 = d1, part 1 = d0): half the DMA and fragment bytes of fp16 + q8.
    constexpr bool I8 = NSPLIT == 4 || NSPLIT == 5;
    constexpr bool LOLO = NSPLIT == 4;          // NSPLIT == 5: without the d0 x d0 product (2^-16 of the leading one per term)
    constexpr int ES = I8 ? 1 : 2;              // bytes per element of one exchange part
    constexpr int CPR = KP * ES / 16;           // 16-byte cells per row per piece
    constexpr int SWZ = (CPR & -CPR) - 1;       // XOR mask that stays inside the row
    constexpr int NPARTS = NSPLIT == 1 ? 1 : 2; // hi (, lo or, NSPLIT == 2, the q8 image; NSPLIT == 4: the two digits)
    constexpr int PIECE_BYTES = LG_BN * KP * ES; // one part of one piece
    constexpr int STP = I8 ? 3 : NPARTS;        // staging arrays: hi pairs, lo / q8 (, NSPLIT == 4: the digit bytes)
    static_assert(!I8 || KP == 128 || KP == 64, "int8-limb pieces are 64 or 128 columns (at least four cells per row)");
    constexpr int NG = DUAL ? 2 : 1;            // groups per workgroup
    static_assert(F % KP == 0, "feature size must be a multiple of the piece width");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char *sPiece = smem_raw;                                           // [2][NPARTS][PIECE_BYTES]
    // h staging for the 16-byte row stores: packed unit pairs, [NPARTS][16 pairs][ST_LD dwords] (chunk minor)
    unsigned *sT = reinterpret_cast<unsigned *>(smem_raw + 2 * NPARTS * PIECE_BYTES);
    unsigned *sTy = reinterpret_cast<unsigned *>(smem_raw + NPARTS * PIECE_BYTES);   // YALT: [16][ST_LD] in piece buffer 1
    static_assert(!YALT || NPARTS * PIECE_BYTES >= 16 * ST_LD * 4, "piece buffer 1 holds the alternative y staging");
    float *sC0 = reinterpret_cast<float *>(sT + STP * 16 * ST_LD);                // [NG][32 units][64 chunks] cell state
    // input-projection tile of the step: [NG][64 chunks][32 cells of 16 B = the four gates of one unit], cell XOR (chunk & 31)
    unsigned char *sG0 = reinterpret_cast<unsigned char *>(sC0 + NG * LG_UNITS * LG_BN);
    constexpr unsigned OFF_G = 2 * NPARTS * PIECE_BYTES + STP * 16 * ST_LD * 4 + NG * LG_UNITS * LG_BN * 4;   // of sG0 in the block
    constexpr unsigned G_TILE = LG_BN * LG_UNITS * 16;                                                    // one group's gin tile
    int *sFlag = reinterpret_cast<int *>(sG0 + NG * LG_BN * LG_UNITS * 16);
    // DUAL: the first W_hh fragment lives in LDS (16 B per thread behind the flags and stamps) and is read back at the top of
    // every group-step: with all 512 registers taken hipcc otherwise parks half of it in scratch, and the reload -- a
    // scratch load with vmcnt(0) behind it -- would wait for the other group's gin tile and y stores still in flight
    unsigned char *sW0 = reinterpret_cast<unsigned char *>(sFlag) + 16 + 80;
    float *sScale = reinterpret_cast<float *>(sW0 + 256 * 16);                  // NSPLIT == 4: row scales of the 128 gate rows
    // DUAL, even piece count: the coming group-step's first piece is requested in the DMA-free issue slots of the last piece
    constexpr bool EIL = DUAL && NP >= 2 && NP % 2 == 0;
#ifdef XB_LSTM_STAMPS
    constexpr bool PARK = !I8;          // the stamp bookkeeping costs the single-group kernel the same registers
#else
    constexpr bool PARK = DUAL && !I8;
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    // LDS byte addresses as 32-bit integers (a generic pointer cast to the LDS address space costs a null check per use)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)smem_raw;
    const unsigned lds_w = lds0 + (unsigned)wid * 1024u;       // this wave's 1 KiB slot of a request group (wave-uniform: an SGPR)
    const int members = F / LG_UNITS;
    const int ngroups = (p.nslab + LG_BN - 1) / LG_BN;
    const int gh = DUAL ? (ngroups + 1) / 2 : ngroups;          // workgroup slots: slot g serves group g (and g + gh)
    const int g8 = (gh + 7) & ~7;
    // default: blocks b, b+8, b+16.. (one XCD under round-robin dispatch) form a group -- speed only.
    // spread = 1 deals a group's members over consecutive blocks, i.e. over all XCDs (placement test).
    const int grp = p.spread ? (int)blockIdx.x / members : (int)blockIdx.x % g8;
    const int mb = p.spread ? (int)blockIdx.x % members : (int)blockIdx.x / g8;
    if (grp >= gh) return;
    const bool second = DUAL && grp + gh < ngroups;             // this slot has a second group
    const int N = p.N, T = p.T;
    const int nlast = p.n0 + p.nslab - 1;
    const int hsel = lane >> 5;
    const int ubase = mb * LG_UNITS + wid * 8;   // first unit of this wave

    // ---- W_hh fragments: row = gate-interleaved (unit*4+gate), lane l: row (l&31), k-chunk (l>>5)
    // NSPLIT == 2: per 32 columns one q8 fragment instead of two lo fragments -- lanes 0-31 hold the block's Wh8 half,
    // lanes 32-63 its Wl8 half (A operand of the block-scaled MFMA; the h fragments below take the opposite halves)
    half8 wh[I8 ? 1 : KS], wl[I8 ? 1 : KS];
    v8i wq[KS / 2 > 0 ? KS / 2 : 1];
    // NSPLIT == 4: per 32 columns the lane's 16 bytes of each digit of its row (k = 32 b + 16 hsel + byte)
    v4i wd1[I8 ? KS / 2 : 1], wd0[I8 ? KS / 2 : 1];
    if constexpr (I8) {
        const size_t row = (size_t)ubase * 4 + (lane & 31);
#pragma unroll
        for (int b = 0; b < KS / 2; ++b) {
            wd1[b] = *reinterpret_cast<const v4i *>(p.wq1 + row * F + b * 32 + hsel * 16);
            wd0[b] = *reinterpret_cast<const v4i *>(p.wq0 + row * F + b * 32 + hsel * 16);
        }
        if (tid < 128) sScale[tid] = p.wscale[(size_t)mb * 128 + tid];
    } else {
        const size_t row = (size_t)ubase * 4 + (lane & 31);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            wh[k] = *reinterpret_cast<const half8 *>(p.w_hi + row * F + k * 16 + hsel * 8);
            if (NSPLIT == 3) wl[k] = *reinterpret_cast<const half8 *>(p.w_lo + row * F + k * 16 + hsel * 8);
        }
        if (NSPLIT == 2) {
            const unsigned char *wq8 = reinterpret_cast<const unsigned char *>(p.w_lo);
#pragma unroll
            for (int b = 0; b < KS / 2; ++b)
                wq[b] = *reinterpret_cast<const v8i *>(wq8 + (row * F + b * 32) * 2 + hsel * 32);
        }
    }
    const int sca = 127 - p.w_exp, scb = 127 - 8 - 11;     // E8M0 scale bytes: W image exponent; h image exponent 8 (+11)
    if (PARK) *reinterpret_cast<half8 *>(sW0 + tid * 16) = wh[0];

    // ---- the group being served (wave-uniform; re-pointed at every group-step when DUAL)
    int cbase = 0;                 // first chunk of the group
    unsigned *cnt = nullptr;       // its arrival counter
    half_t *xg = nullptr;          // its exchange buffer [parity][part][64 rows][F]
    float *sC = sC0;
    unsigned char *sG = sG0;
    unsigned lds_g = lds_w + OFF_G;   // LDS address of this wave's first 1 KiB of the group's gin tile
    constexpr size_t XPAR = (size_t)2 * LG_BN * F, XPART = (size_t)LG_BN * F;
    auto serve = [&](int gi) {
        const int g = grp + gi * gh;
        cbase = p.n0 + g * LG_BN;
        cnt = p.sync + (size_t)(p.grp0 + g) * LG_SYNC;
        xg = p.xh + (size_t)(p.grp0 + g) * (2 * 2 * LG_BN * F);
        sC = sC0 + gi * (LG_UNITS * LG_BN);
        sG = sG0 + gi * G_TILE;
        lds_g = lds_w + OFF_G + (unsigned)gi * G_TILE;
    };

    // ---- cell state lives in LDS as [unit][chunk] (the register file is full of W_hh): lane owns
    //      (chunk = 32*nt + (l&31), unit = 8*wid + 2*rg + hsel), only ever touched by that lane
#pragma unroll 1
    for (int gi = 0; gi < NG; ++gi) {
        if (gi == 1 && !second) break;
        serve(gi);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = cbase + nt * 32 + (lane & 31);
            const int ch = n <= nlast ? n : nlast;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                sC[(wid * 8 + 2 * rg + hsel) * LG_BN + nt * 32 + (lane & 31)] =
                    p.c_state[(size_t)ch * F + ubase + 2 * rg + hsel];
        }
    }

    // ---- same-XCD exchange (round 3).  A write-through (sc1) store DROPS the line from the XCD's L2, so every member's h
    //      fetch of the next step starts with a fabric round trip; a plain store keeps it there, and a group's members
    //      normally share an XCD (blocks b, b + 8, .. under round-robin dispatch).  "Normally" is not a contract, so the
    //      members PROVE it per launch: each ORs the bit of the XCD it actually runs on (HW_REG_XCC_ID) into a word of its
    //      group's sync slot; when a member's second wait for the group has completed, every member has posted its bit (the OR
    //      is older than the member's first arrival), and a mask with exactly one bit set switches this workgroup's later
    //      exchange stores to plain ones (the loads stay sc1 = L2-served).  Any other mask -- members on several XCDs, the
    //      placement test -- keeps the write-through form, which is correct anywhere.  One-group-per-workgroup kernel only.
    // (mask word and shift are recomputed from the kernel arguments where they are used: nothing extra stays live in the loop)
    // (two groups per workgroup: both groups have the same member workgroups, each posts into both groups' words and decides
    //  per group at that group's first wait; sFlag[2 + gi])
    if (p.persistent && p.xcd_local && tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;        // HW_REG_XCC_ID[3:0]
        for (int gi = 0; gi < (second ? 2 : 1); ++gi)
            __hip_atomic_fetch_or(p.sync + (size_t)(p.grp0 + grp + gi * gh) * LG_SYNC + 1 + ((p.slab >> 2) & 3), 1u << (8 * (p.slab & 3) + xcc),
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) { sFlag[2] = 0; sFlag[3] = 0; }
#ifdef XB_LSTM_STAMPS
    unsigned long long *sStamp = reinterpret_cast<unsigned long long *>(sFlag + 4);
    if (tid == 0) for (int i = 0; i < 10; ++i) sStamp[i] = 0;
    unsigned long long stamp_prev = __builtin_readcyclecounter();
#endif
    // The input projection of a step (gin: 64 chunks x 128 gate rows x 4 B = 32 KiB per workgroup) is pulled into LDS by
    // LDS-DMA one step AHEAD, right after the exchange stores of the previous step: it is in flight during drain / arrive /
    // poll instead of starting at the top of the step (where the polling wave's wait absorbed its whole HBM latency),
    // it needs no registers, and the store drain becomes a counted wait (all but these eight youngest operations).
    // Instruction q = 4 d + wid (d = 0..7) fills chunk rows 2q, 2q + 1: lane i -> row 8 d + 2 wid + (i >> 5), cell i & 31,
    // source cell (i & 31) ^ (row & 7): the lane part of the address is the same for all eight instructions.  Rows
    // past the slab's last chunk read whatever follows (other chunks' rows or the 64 slack rows behind the buffer):
    // their results are never stored.
    auto issue_gin = [&](int tn) {
        // lane part recomputed per call (a few VALU) rather than kept in a register across the MFMA loop, where it would be
        // spilled and its reload (a scratch load + wait) would drain whatever is in flight
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const unsigned gin_lane = (unsigned)(((2 * wid + (lo >> 5)) * 128 + (((lo & 31) ^ ((2 * wid + (lo >> 5)) & 7)) * 4)) * 4);
        // ONE wave-uniform base (an SGPR pair) per tile; request d adds 8 chunk rows = 4 KiB as a literal to the lane's byte
        // offset and 4 KiB to the LDS address (lean request, see dma16_lean_sc1); member-major gin (xb_internal.h): this
        // workgroup's 64 chunk rows of 128 gate columns are contiguous
        const unsigned char *base = reinterpret_cast<const unsigned char *>(p.gin + (((size_t)tn * members + mb) * N + cbase) * 128);
#pragma unroll
        for (int d = 0; d < 8; ++d) dma16_lean_nt(gin_lane, d * 4096, base, lds_g, d * 4096);
    };
    // (Requesting the tile still earlier -- in the free issue slots of the second-to-last piece, that piece ending on
    // vmcnt(8) -- was measured too: the poll no longer waits for it, but the first-piece landing and the MFMA phase grow by
    // as much: 76.1 vs 74.5 ms per five layers.)
    // accumulators start from the input projection (+ biases): lane (chunk row r, unit u) reads cell u ^ (r & 7)
    // (two lanes of a 16-lane read group share a bank slot: a 2-way conflict on eight reads per step)
    auto acc_from_gin = [&](floatx16 (&acc)[2]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int r = nt * 32 + (lane & 31);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int u = wid * 8 + 2 * rg + hsel;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(sG + r * (LG_UNITS * 16) + ((u ^ (r & 7)) * 16));
                acc[nt][4 * rg + 0] = v[0]; acc[nt][4 * rg + 1] = v[1];
                acc[nt][4 * rg + 2] = v[2]; acc[nt][4 * rg + 3] = v[3];
            }
        }
    };

    // h_{t-1} of a group's chunks comes piece by piece through LDS (every load sc1).
    // A piece is NPARTS * CPR wave-instructions of 1 KiB; wave w issues NDMA = NPARTS * CPR / 4 of
    // them (q = w, w+4, ..), exactly one per MFMA k-step when NSPLIT == 3, so the next piece's
    // DMA is issued in the shadow of this piece's MFMAs instead of in front of them.
    // Instruction q covers cells [64q, 64q+64) = rows RPI*q .. of CPR cells.  For power-of-two CPR the
    // per-lane part of the source offset is the same for every q of a wave (q = wid mod 4 and
    // RPI*4 = 0 mod CPR), so it is ONE register; everything else is wave-uniform scalar arithmetic.
    constexpr int NDMA = NPARTS * CPR / 4;
    constexpr bool POW2 = (CPR & (CPR - 1)) == 0;
    constexpr int RPI = 64 / (POW2 ? CPR : 1);
    int lane_off_step = 0;      // computed once per (group-)step: as a value kept across the gate math it would be spilled
    // POW2 (every shipped size): lane_off_step is the lane's BYTE offset inside the exchange image, the wave's row block
    // included -- row RPI * wid + lane / CPR, swizzled cell -- and a request is the lean three-instruction form: request (part,
    // j) of piece pc adds the literal part * (part stride) + 4 j RPI rows, the piece's column offset is the load's immediate
    auto issue_dma = [&](const half_t *xprev, int pc, int d) {
        const int lo = lane;
        const int lane_off = lane_off_step;
        const int part = NPARTS == 2 ? (d & 1) : 0;
        const int j = NPARTS == 2 ? (d >> 1) : d;
        const int lconst = (pc & 1) * NPARTS * PIECE_BYTES + part * PIECE_BYTES + 4 * j * 1024;
        if constexpr (POW2) {
            constexpr int ROWB = I8 ? F : F * 2;                         // bytes per row of one part of the image
            // (the piece's column offset goes into the literal as well: the load's IMMEDIATE offset is added to the LDS address
            //  too -- measured in round 5: with offset:pc * 256 every piece but the first landed 256 pc bytes off)
            const int vconst = part * (int)(XPART * 2) + 4 * j * RPI * ROWB + pc * KP * ES;
            dma16_lean_sc1((unsigned)lane_off, vconst, xprev, lds_w, lconst, 0);
        } else {
            const int q = wid + 4 * j;
            const int cell = 64 * q + lo;
            const int row = cell / CPR, pos = cell % CPR;
            dma16_sc1(xprev + part * XPART + (size_t)row * F + pc * KP + (pos ^ (row & SWZ)) * 8, sPiece + lconst + wid * 1024);
        }
    };

#pragma unroll 1
    for (int gi = 0; gi < NG; ++gi) {
        if (gi == 1 && !second) break;
        serve(gi);
        issue_gin(p.reverse ? T - 1 - p.s_begin : p.s_begin);
    }
    bool early = false;     // DUAL: the first piece of the coming group-step was requested during the previous one
    // DUAL with both groups present: the exchange stores of a group-step are not drained at its end; the next full drain +
    // barrier -- the one that closes the first piece of the other group's step, ~2.7 k cycles later -- covers them, and the
    // arrival goes out behind that (the group's hand-off still has most of the other group's step to complete: its members
    // are looked at ~3.8 k cycles after that point).  One group per slot: its next step waits for this very arrival -- not deferred.
    constexpr bool DEFER = DUAL && XB_LSTM_DEFER_ARRIVE != 0;
    unsigned *arrive_due = nullptr;
    int sig_i = 0, sig_next = XB_SIG(p.sig_flag) ? (int)((long long)T / p.sig_nts) : -1;     // slab being worked on, its end step
    for (int s = p.s_begin; s < p.s_end; ++s) {
        const int t = p.reverse ? T - 1 - s : s;
#pragma unroll 1
        for (int gi = 0; gi < NG; ++gi) {
            if (gi == 1 && !second) break;
            if (DUAL) serve(gi);
            XB_STAMP(0);   // loop overhead / y stores of the previous group-step
            floatx16 acc[2];
            v16i a11[I8 ? 2 : 1], amid[I8 ? 2 : 1], a00[I8 ? 2 : 1];      // NSPLIT == 4: digit-product sums by weight
            // DUAL: the group-step this workgroup serves next, whether it has a recurrent term (s > 0) and whether its
            // group has to be polled first (not in the first step of a launch: the previous launch has retired)
            const int ngi = (DUAL && gi == 0 && second) ? 1 : 0;
            const int ns = (DUAL && gi == 0 && second) ? s : s + 1;
            const bool nxt_h = DUAL && ns < p.s_end && ns > 0 && s > 0;
            const bool nxt_poll = p.persistent && ns > p.s_begin;
            unsigned *ncnt = p.sync + (size_t)(p.grp0 + grp + ngi * gh) * LG_SYNC;
            const unsigned ntarget = (unsigned)members * (p.sync_base + (unsigned)(ns - p.s_begin));
            unsigned seen = 0;
            const half_t *xnext = p.xh + (size_t)(p.grp0 + grp + ngi * gh) * (2 * 2 * LG_BN * F) + (size_t)((ns - 1) & 1) * XPAR;
            const bool was_early = early;
            early = false;
            int go = 0;         // EIL: request the coming group-step's first piece during the last piece

            if (s > 0) {
                const half_t *xprev = xg + (size_t)((s - 1) & 1) * XPAR;      // (NSPLIT == 4: same byte offset, XPAR * 2)
                {
                    int lo = lane;
                    if (PARK) asm volatile("" : "+v"(lo));
                    const int lrow = RPI * wid + lo / CPR;
                    lane_off_step = POW2 ? lrow * (I8 ? F : F * 2) + (((lo % CPR) ^ (lrow & SWZ)) * 16) : 0;
                }
                if (!DUAL || !was_early) {
                    if (p.persistent && s > p.s_begin) {
                        // wait until every member of the group has published h_{t-1}
                        if (tid == 0) {
                            const unsigned target = (unsigned)members * (p.sync_base + (unsigned)(s - p.s_begin));
                            const unsigned long long t0 = __builtin_readcyclecounter();
                            int ok = 1;
                            // the counter is polled back to back (one L2 round trip per poll); the error word and the
                            // timeout are looked at every 64th poll only
                            for (unsigned spins = 1; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins) {
                                if ((spins & 63u) == 0 &&
                                    (__hip_atomic_load(p.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                                     __builtin_readcyclecounter() - t0 > LG_SPIN_CYCLES)) {
                                    __hip_atomic_store(p.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    ok = 0;
                                    break;
                                }
                            }
                            *sFlag = ok;
                        }
                        __syncthreads();
                        if (*sFlag == 0) {
                            // timed out (the error word is set, the host fails the batch): release the stream that waits for
                            // this launch's slabs -- a wait on the flag has no timeout of its own
                            if (tid == 0 && XB_SIG(p.sig_flag))
                                __hip_atomic_fetch_max(p.sig_flag, p.sig_base + (unsigned)p.sig_nts, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                            return;
                        }
                    }
                    XB_STAMP(1);   // gin loads issued + wait for the group
#pragma unroll
                    for (int d = 0; d < NDMA; ++d) issue_dma(xprev, 0, d);
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first piece and the gin tile (this wave's shares)
                    __syncthreads();
                }
                // (early: the drain wait and barrier that ended the previous group-step covered the first piece and this
                //  group's gin tile, both older than the exchange stores drained there)
                XB_STAMP(2);   // first piece landed
                // the group's second hand-off is complete (blocking poll or, with two groups per workgroup, the look-ahead one):
                // every member has arrived at least once, so every member's XCD bit is in the mask
                if (p.xcd_local && s == p.s_begin + 2 && tid == 0) {
                    const unsigned m = (__hip_atomic_load(cnt + 1 + ((p.slab >> 2) & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >>
                                        (8 * (p.slab & 3))) & 0xffu;
                    sFlag[2 + gi] = (m != 0 && (m & (m - 1)) == 0) ? 1 : 0;      // all members on ONE XCD
                }
                half8 w0 = wh[0];
                if (PARK) {
                    int to = tid;
                    asm volatile("" : "+v"(to));
                    w0 = *reinterpret_cast<const half8 *>(sW0 + to * 16);
                }
                acc_from_gin(acc);
                // lane byte offsets of the B-fragment cells inside a piece part (one register per k-step / q8 cell; the
                // piece buffer, the part and the column tile are immediates)
                unsigned fa[KSP], qa[KSP / 2 > 0 ? KSP / 2 : 1][2];
                {
                    // one cell address each for the fp16 and the q8 fragments; every other k-step's is an XOR away: the
                    // k-step moves bits 1.. of the cell index, the swizzle key XORs into the same bits, (a | b) ^ c splits
                    // (round 5: 2 + 14 VALU per group-step instead of three per address)
                    int lo = lane;
                    asm volatile("" : "+v"(lo));
                    const unsigned r = (unsigned)lo & 31u, hs = (unsigned)lo >> 5;
                    static_assert(!POW2 || I8 || 2 * KSP <= CPR, "the k-step bits stay inside the row (int8 limbs: only k-steps below KSP / 2 are used)");
                    const unsigned fa0 = (r * CPR + (hs ^ (r & SWZ))) * 16;
                    const unsigned qa0 = (r * CPR + ((2 * (1 - hs)) ^ (r & SWZ))) * 16;
#pragma unroll
                    for (int ks = 0; ks < KSP; ++ks)
                        fa[ks] = POW2 ? fa0 ^ (unsigned)(ks << 5) : (r * CPR + ((2 * ks + hs) ^ (r & SWZ))) * 16;
#pragma unroll
                    for (int b = 0; b < KSP / 2; ++b)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            qa[b][j] = POW2 ? qa0 ^ (unsigned)((4 * b + j) << 4) : (r * CPR + ((4 * b + 2 * (1 - hs) + j) ^ (r & SWZ))) * 16;
                }
                // B fragments double-buffered by k-step: the 4 reads of k-step ks+1 are issued before the 6 MFMAs
                // of ks (sched_barrier keeps hipcc from sinking the reads back to their first use).  Round 5: the k-step
                // pipeline runs ACROSS the piece boundary (PIPE): a piece that has a successor is closed (DMA drain +
                // barrier) in front of its LAST k-step's MFMAs, whose fragments are in registers by then, and the
                // successor's first fragments are requested behind that barrier -- their LDS latency, which used to sit
                // exposed at the top of every piece with the matrix pipe drained, runs under those MFMAs.
                half8 fh[2][2], fl[2][2];
                v8i fq[2];
                constexpr bool PIPE = !I8 && XB_LSTM_PIPE_PIECES != 0 && KSP % 2 == 0;
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {
                    const unsigned char *buf = sPiece + (pc & 1) * NPARTS * PIECE_BYTES;
                    const unsigned char *bufn = sPiece + ((pc + 1) & 1) * NPARTS * PIECE_BYTES;      // the successor's buffer
                    auto load_frags_at = [&](const unsigned char *bf, int ks, half8 (&h)[2], half8 (&l)[2]) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const unsigned char *a = bf + nt * (32 * CPR * 16) + fa[ks];
                            h[nt] = *reinterpret_cast<const half8 *>(a);
                            if (NSPLIT == 3) l[nt] = *reinterpret_cast<const half8 *>(a + PIECE_BYTES);
                        }
                    };
                    auto load_frags = [&](int ks, half8 (&h)[2], half8 (&l)[2]) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            // row = 32 nt + (lane & 31); row & SWZ does not depend on nt, so nt is an immediate offset
                            const unsigned char *a = buf + nt * (32 * CPR * 16) + fa[ks];
                            h[nt] = *reinterpret_cast<const half8 *>(a);
                            if (NSPLIT == 3) l[nt] = *reinterpret_cast<const half8 *>(a + PIECE_BYTES);
                        }
                    };
                    // q8 fragment of 32-column block `blk` of the piece: the half OPPOSITE to the W fragment's
                    // (lanes 0-31: the l8 cells 2, 3 of the block; lanes 32-63: the h8 cells 0, 1)
                    auto load_q8 = [&](int blk) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const unsigned char *rb = buf + PIECE_BYTES + nt * (32 * CPR * 16);
                            const v4i x = *reinterpret_cast<const v4i *>(rb + qa[blk][0]);
                            const v4i y = *reinterpret_cast<const v4i *>(rb + qa[blk][1]);
                            fq[nt] = __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                    };
                    // NSPLIT == 4: the lane's 16 bytes of each digit for 32-column block b (cell 2 b + hsel of the row)
                    v4i bd1[2][2], bd0[2][2];
                    auto load_dig = [&](int b, v4i (&d1)[2], v4i (&d0)[2]) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const unsigned char *a = buf + nt * (32 * CPR * 16) + fa[b];
                            d1[nt] = *reinterpret_cast<const v4i *>(a);
                            d0[nt] = *reinterpret_cast<const v4i *>(a + PIECE_BYTES);
                        }
                    };
                    if constexpr (I8) load_dig(0, bd1[0], bd0[0]);
                    else if (!PIPE || pc == 0) load_frags(0, fh[0], fl[0]);      // (PIPE, pc > 0: requested behind the previous piece's closing barrier)
                    // DUAL: has the group of the coming group-step arrived?  One look at its counter (its members had a whole
                    // group-step for it) at the start of the piece whose closing barrier publishes the answer: the last
                    // piece, or (EIL) the one before it.
                    constexpr int PCHK = EIL ? NP - 2 : NP - 1;
                    if (DUAL && pc == PCHK && nxt_h && nxt_poll && tid == 0)
                        seen = __hip_atomic_load(ncnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (EIL && pc == NP - 1) go = __builtin_amdgcn_readfirstlane(nxt_h ? sFlag[1] : 0);
                    if constexpr (I8) {
                        constexpr int KBP = KP / 32;            // 32-column blocks per piece = DMA requests per wave and piece
                        static_assert(NDMA == KBP, "one piece request per block");
                        const v16i zero16 = {};
#pragma unroll
                        for (int b = 0; b < KBP; ++b) {
                            const int kb = pc * KBP + b;
                            if (b + 1 < KBP) load_dig(b + 1, bd1[(b + 1) & 1], bd0[(b + 1) & 1]);
                            __builtin_amdgcn_sched_barrier(0);
                            // column tiles interleaved: the two products into amid[nt] are two issues apart (a dependent MFMA
                            // issued back to back waits for the whole latency of its predecessor)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                a11[nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wd1[kb], bd1[b & 1][nt], kb == 0 ? zero16 : a11[nt], 0, 0, 0);
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                amid[nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wd1[kb], bd0[b & 1][nt], kb == 0 ? zero16 : amid[nt], 0, 0, 0);
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                amid[nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wd0[kb], bd1[b & 1][nt], amid[nt], 0, 0, 0);
                            if constexpr (LOLO) {
#pragma unroll
                                for (int nt = 0; nt < 2; ++nt)
                                    a00[nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wd0[kb], bd0[b & 1][nt], kb == 0 ? zero16 : a00[nt], 0, 0, 0);
                            }
                            // the piece requests go out in the first half of the piece so that the last has landed at its barrier
                            constexpr int DPB = KBP >= 2 ? 2 : 1;
                            if (b * DPB < NDMA) {
#pragma unroll
                                for (int j = 0; j < DPB; ++j) {
                                    if (pc + 1 < NP) issue_dma(xprev, pc + 1, b * DPB + j);
                                    else if (EIL && go) issue_dma(xnext, 0, b * DPB + j);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
#pragma unroll
                    for (int ks = 0; ks < KSP; ++ks) {
                        const int kg = pc * KSP + ks;
                        const bool closing = PIPE && ks + 1 == KSP && pc + 1 < NP;     // this piece is closed in front of these MFMAs
                        if (ks + 1 < KSP) load_frags(ks + 1, fh[(ks + 1) & 1], fl[(ks + 1) & 1]);
                        if (NSPLIT == 2 && (ks & 1) == 0) load_q8(ks >> 1);        // used by the odd k-step that follows
                        __builtin_amdgcn_sched_barrier(0);
                        // ONE counted wait per k-step (round 5): everything but the reads just issued has landed, i.e. every
                        // fragment this k-step's MFMAs take (they were requested a k-step ago).  hipcc otherwise puts a counted
                        // lgkmcnt in front of EVERY MFMA -- 130 s_waitcnt per group-step on a wave whose every instruction costs an
                        // issue slot; with this wait in its scoreboard it emits none.  (the builtin needs a literal: spelled out)
                        {
                            constexpr int RD = NSPLIT == 3 ? 4 : 2;                    // ds_reads of one load_frags
                            const bool more = ks + 1 < KSP, q8 = NSPLIT == 2 && (ks & 1) == 0;
#define XB_LGKM(n) __builtin_amdgcn_s_waitcnt(0xC07F | ((n) << 8))
                            if (more && q8) XB_LGKM(RD + 4);
                            else if (more) XB_LGKM(RD);
                            else if (q8) XB_LGKM(4);
                            else XB_LGKM(0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (closing) {
                            // every fragment of this piece is in registers (lgkmcnt(0) above: nothing was requested in this
                            // k-step).  Close the piece as its end used to: look-ahead flag, DMA drain, barrier, deferred arrival.
                            XB_STAMP(3);
                            if (DUAL && pc == PCHK && tid == 0) {
                                sFlag[1] = (nxt_h && (!nxt_poll || seen >= ntarget)) ? 1 : 0;
                                XB_LGKM(0);
                            }
                            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next piece landed (this wave's share)
                            __builtin_amdgcn_s_barrier();
                            __builtin_amdgcn_sched_barrier(0);
                            if (DEFER && pc == 0 && arrive_due) {      // the other group's exchange stores are at L2 in every wave
                                if (tid == 0) __hip_atomic_fetch_add(arrive_due, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                arrive_due = nullptr;
                            }
                            XB_STAMP(7);
                            load_frags_at(bufn, 0, fh[0], fl[0]);      // fh[0] is free: this k-step is odd (KSP is even)
                            __builtin_amdgcn_sched_barrier(0);
                        }
#undef XB_LGKM
                        // The next piece's requests, two per k-step so that the last one is issued by mid-piece and has landed at
                        // the barrier.  (XB_LSTM_DMA_SPREAD: every request directly behind ONE MFMA -- behind the FP8 ones, which
                        // keep the pipe busy for 64 cycles, where the k-step has them -- instead of in pairs behind the k-step's
                        // MFMAs: measured, not adopted.)
                        auto dma_slot = [&](int j) {
                            if (2 * ks + j >= NDMA) return;
                            if (pc + 1 < NP) issue_dma(xprev, pc + 1, 2 * ks + j);
                            else if (EIL && go) issue_dma(xnext, 0, 2 * ks + j);
                            __builtin_amdgcn_sched_barrier(0);
                        };
                        constexpr bool XB_DMA_SPREAD = XB_LSTM_DMA_SPREAD != 0;
                        const bool q_step = NSPLIT == 2 && (ks & 1) == 1;
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            if (NSPLIT == 3) {
                                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[kg], fh[ks & 1][nt], acc[nt], 0, 0, 0);
                                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kg], fl[ks & 1][nt], acc[nt], 0, 0, 0);
                            }
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kg == 0 ? w0 : wh[kg], fh[ks & 1][nt], acc[nt], 0, 0, 0);
                            if (XB_DMA_SPREAD) __builtin_amdgcn_sched_barrier(0);      // (pins the MFMA order: hipcc otherwise pairs
                            if (XB_DMA_SPREAD && !q_step) dma_slot(nt);                //  each column tile's dependent fp16 / FP8 MFMAs)
                        }
                        if (q_step) {
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) {
                                acc[nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wq[kg >> 1], fq[nt], acc[nt], 0, 0, 0, sca, 0, scb);
                                if (XB_DMA_SPREAD) __builtin_amdgcn_sched_barrier(0);
                                if (XB_DMA_SPREAD) dma_slot(nt);
                            }
                        }
                        if (!XB_DMA_SPREAD) {
                            dma_slot(0);
                            dma_slot(1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    }
                    if (!PIPE || pc + 1 == NP) {
                    XB_STAMP(3);   // piece compute (ds_read + MFMA + next piece's DMA issue)
                    if (DUAL && pc == PCHK && tid == 0) sFlag[1] = (nxt_h && (!nxt_poll || seen >= ntarget)) ? 1 : 0;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // next piece landed (this wave's share)
                    __syncthreads();
                    if (DEFER && pc == 0 && arrive_due) {      // the other group's exchange stores are at L2 in every wave
                        if (tid == 0) __hip_atomic_fetch_add(arrive_due, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        arrive_due = nullptr;
                    }
                    XB_STAMP(7);   // piece DMA wait + barrier
                    }
                }
                if constexpr (I8) {
                    // pre-activation = gin + row scale * (2^16 S11 + 2^8 (S10 + S01) + S00): every sum is exact, the fp32
                    // combination rounds once per term (|S11| < 2^24)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const f32x4 sc = *reinterpret_cast<const f32x4 *>(sScale + (wid * 8 + 2 * rg + hsel) * 4);
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int r = 4 * rg + g;
                                const float t = __builtin_fmaf(65536.0f, (float)a11[nt][r],
                                                               LOLO ? __builtin_fmaf(256.0f, (float)amid[nt][r], (float)a00[nt][r])
                                                                    : 256.0f * (float)amid[nt][r]);
                                acc[nt][r] = __builtin_fmaf(sc[g], t, acc[nt][r]);
                            }
                        }
                }
            }

            if (s == 0) {      // no recurrent term in the very first step: the accumulators are the input projection
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (DEFER && arrive_due) {
                    if (tid == 0) __hip_atomic_fetch_add(arrive_due, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    arrive_due = nullptr;
                }
                acc_from_gin(acc);
            }

            // DUAL: request the first piece of the coming group-step now -- it lands behind the gate math, and the drain
            // wait below (everything but the eight youngest operations) covers it
            if (EIL) {
                early = go != 0;    // requested inside the last piece; its closing wait and barrier have landed it
            } else if (DUAL && nxt_h && sFlag[1] != 0) {
#pragma unroll
                for (int d = 0; d < NDMA; ++d) issue_dma(xnext, 0, d);
                early = true;
            }
#ifdef XB_LSTM_STAMPS
            if (tid == 0) { sStamp[8] += early ? 1 : 0; sStamp[9] += 1; }
#endif

            // the lane's eight cell states, requested together (round 5): hipcc otherwise reads each right before its use and
            // waits for it there -- eight exposed LDS round trips per group-step.  The fragment registers are dead by now.
            float cprev[2][4];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) cprev[nt][rg] = sC[(wid * 8 + 2 * rg + hsel) * LG_BN + nt * 32 + (lane & 31)];
            __builtin_amdgcn_sched_barrier(0);
            // gates -> cell -> hidden.  A lane owns units 2*rg + hsel of chunk (lane & 31); v_permlane32_swap pairs them
            // with the other half-wave's units so that each lane packs two ADJACENT units into one dword, written to
            // the [pair][chunk] staging (consecutive lanes -> consecutive dwords: conflict-free).
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                unsigned phi[4], plo[4];
                float hq[4], lq[4];
                unsigned dg1[4], dg0[4];            // NSPLIT == 4: digit bytes of the four units
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const float ig = fast_sigmoid(acc[nt][4 * rg + 0]);
                    const float fg = fast_sigmoid(acc[nt][4 * rg + 1]);
                    const float gg = fast_tanh(acc[nt][4 * rg + 2]);
                    const float og = fast_sigmoid(acc[nt][4 * rg + 3]);
                    float *cp = sC + (wid * 8 + 2 * rg + hsel) * LG_BN + nt * 32 + (lane & 31);
                    const float cn = __builtin_fmaf(ig, gg, fg * cprev[nt][rg]);     // spelled out: lstm_quad_kernel must round the same way
                    *cp = cn;
                    const float hv = og * fast_tanh(cn);
                    half_t hi, lo;
                    split_f16(hv, hi, lo);
                    phi[rg] = (unsigned)__builtin_bit_cast(unsigned short, hi);
                    plo[rg] = (unsigned)__builtin_bit_cast(unsigned short, lo);
                    hq[rg] = (float)hi * 256.0f;                   // |h| < 1: below the e4m3 maximum by construction
                    lq[rg] = (hv - (float)hi) * 524288.0f;         // 2^19; |residual| <= 2^-11 |hi|
                    if constexpr (I8) {
                        // 16-bit fixed point of h (|q| <= 32512) as two balanced signed digits q = 256 d1 + d0
                        const int q = (int)__builtin_rintf(hv * 32512.0f);
                        const int d0 = ((q + 128) & 255) - 128;
                        dg1[rg] = (unsigned)((q - d0) >> 8) & 255u;
                        dg0[rg] = (unsigned)d0 & 255u;
                    }
                }
                // lanes < 32 hold even units v[rg] = unit 2rg, lanes >= 32 the odd ones v[rg] = unit 2rg+1.
                // v_permlane32_swap(vdst, src) exchanges vdst's upper half-wave with src's lower half-wave, so
                //   swap(v[0], v[2]) -> {r[0], r[1]} = low lanes {unit 0, unit 1}, high lanes {unit 4, unit 5}
                //   swap(v[1], v[3]) -> low lanes {unit 2, unit 3}, high lanes {unit 6, unit 7}
#pragma unroll
                for (int part = 0; part < ((NSPLIT == 3 || (YALT && NSPLIT == 2)) ? 2 : 1); ++part) {
                    unsigned *v = part == 0 ? phi : plo;
                    auto r0 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
                    const unsigned e0 = r0[0], o0 = r0[1], e1 = r1[0], o1 = r1[1];
                    const int pr = wid * 4 + hsel * 2;                  // first unit pair of this lane
                    // (YALT, NSPLIT 2: the residual pairs are not part of the exchange image: they go to the y staging)
                    unsigned *dst = ((YALT && NSPLIT == 2 && part == 1) ? sTy : sT + part * 16 * ST_LD) + nt * 32 + (lane & 31);
                    dst[(pr + 0) * ST_LD] = e0 | (o0 << 16);
                    dst[(pr + 1) * ST_LD] = e1 | (o1 << 16);
                }
                if constexpr (I8) {
                    // digit image of the 32 units, laid out like the q8 image below: rows 0..7 the d1 bytes of unit quads
                    // 0..7, rows 8..15 their d0 bytes (third staging array)
                    unsigned X = dg1[0] | (dg1[1] << 8) | (dg0[0] << 16) | (dg0[1] << 24);
                    unsigned Y = dg1[2] | (dg1[3] << 8) | (dg0[2] << 16) | (dg0[3] << 24);
                    auto r = __builtin_amdgcn_permlane32_swap(X, Y, false, false);
                    const unsigned r0 = r[0], r1 = r[1];
                    unsigned *dst = sT + 2 * 16 * ST_LD + nt * 32 + (lane & 31);
                    dst[(wid * 2 + hsel) * ST_LD] = __builtin_amdgcn_perm(r1, r0, 0x05010400u);
                    dst[(8 + wid * 2 + hsel) * ST_LD] = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
                }
                if (NSPLIT == 2 || I8 || (YALT && NSPLIT == 3)) {
                    // q8 image of the 32 units: 16 dword rows in the place of the lo staging -- rows 0..7 the h8 bytes of unit
                    // quads 0..7, rows 8..15 their l8 bytes, so the 16-byte cell reads below need no change.
                    // X = {h8(u_a), h8(u_b), l8(u_a), l8(u_b)} of this lane's units (rg 0, 1), Y of (rg 2, 3); after the swap
                    // low lanes hold units (0,2) / (1,3), high lanes (4,6) / (5,7): one byte permute per image interleaves them.
                    unsigned X = fp8_pair<false>(hq[0], hq[1], 0u), Y = fp8_pair<false>(hq[2], hq[3], 0u);
                    X = fp8_pair<true>(lq[0], lq[1], X);
                    Y = fp8_pair<true>(lq[2], lq[3], Y);
                    auto r = __builtin_amdgcn_permlane32_swap(X, Y, false, false);
                    const unsigned r0 = r[0], r1 = r[1];
                    unsigned *dst = ((YALT && NSPLIT == 3) ? sTy : sT + 16 * ST_LD) + nt * 32 + (lane & 31);
                    dst[(wid * 2 + hsel) * ST_LD] = __builtin_amdgcn_perm(r1, r0, 0x05010400u);
                    dst[(8 + wid * 2 + hsel) * ST_LD] = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
                }
            }
            if (DUAL) {
                // raw barrier: __syncthreads() would drain the first piece just requested
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                __syncthreads();
            }
            // 64 rows x 64 B per part = 256 cells of 16 B: one per thread per part (cell = 4 unit pairs of one chunk)
            const int orow = tid >> 2, occ = tid & 3;
            uint4 vhi, vlo = make_uint4(0, 0, 0, 0);
            {
                const unsigned *src = sT + (occ * 4) * ST_LD + orow;
                vhi = make_uint4(src[0], src[ST_LD], src[2 * ST_LD], src[3 * ST_LD]);
                if (NSPLIT != 1) {
                    const unsigned *sl = src + 16 * ST_LD;
                    vlo = make_uint4(sl[0], sl[ST_LD], sl[2 * ST_LD], sl[3 * ST_LD]);
                }
            }
            uint4 vylo = vlo;            // second part of the layer output: the exchange image's, or (YALT) the other form
            if (YALT) {
                const unsigned *sl = sTy + (occ * 4) * ST_LD + orow;
                vylo = make_uint4(sl[0], sl[ST_LD], sl[2 * ST_LD], sl[3 * ST_LD]);
            }
            if (s + 1 < T) {
                // publish h_t for the group -- also on the last step of a launch: the next launch (next step, or next time
                // slab) starts from the exchange buffer (rows beyond the slab are scratch rows of the exchange buffer)
                if constexpr (I8) {
                    // one 16-byte cell per thread: occ 0, 1 = the d1 bytes of units 0..15 / 16..31 (part 0), occ 2, 3 = d0 (part 1)
                    const unsigned *sd = sT + 2 * 16 * ST_LD + (occ * 4) * ST_LD + orow;
                    const uint4 vd = make_uint4(sd[0], sd[ST_LD], sd[2 * ST_LD], sd[3 * ST_LD]);
                    unsigned char *xb = reinterpret_cast<unsigned char *>(xg) + (size_t)(s & 1) * (XPAR * 2) +
                                        (size_t)(occ >> 1) * (XPART * 2) + (size_t)orow * F + mb * LG_UNITS + (occ & 1) * 16;
                    store16_sc1(xb, vd);
                } else {
                half_t *xcur = xg + (size_t)(s & 1) * XPAR + (size_t)orow * F + mb * LG_UNITS + occ * 8;
                if (__builtin_amdgcn_readfirstlane(sFlag[2 + gi]) != 0) {      // the group sits on one XCD (proven above)
                    store16_l2(xcur, vhi);
                    if (NSPLIT != 1) store16_l2(xcur + XPART, vlo);
                } else {
                    store16_sc1(xcur, vhi);
                    if (NSPLIT != 1) store16_sc1(xcur + XPART, vlo);
                }
                }
            }
            XB_STAMP(4);   // pointwise + exchange stores issued
            if (p.persistent && s + 1 < p.s_end) {
                // next step's gin tile (eight LDS-DMAs per wave), then every storing wave drains its exchange stores: all but
                // the eight youngest operations (raw barrier: __syncthreads() would drain the DMAs as well; the LDS reads of
                // the staging are retired here)
                __builtin_amdgcn_sched_barrier(0);
                issue_gin(p.reverse ? T - 2 - s : s + 1);
                if (DEFER && second) {
                    // no drain here (see arrive_due); the barrier stays: the staging (and, YALT, piece buffer 1) is free for the
                    // other group's step once every wave has read it
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    arrive_due = cnt;
                    XB_STAMP(5);
                    XB_STAMP(6);
                } else {
                asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                XB_STAMP(5);   // stores drained
                if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                XB_STAMP(6);   // arrive
                }
            } else {
                __syncthreads();   // sT is rewritten next step (this also lands an early first piece)
            }
            // layer output for the next layer: plain stores, nobody in this launch reads them
            // (address recomputed from the thread index here: a value kept across the loop gets spilled, and its reload -- a
            // scratch load with a vmcnt(0) behind it -- would wait for the gin DMAs just issued)
            {
                int to = tid;
                asm volatile("" : "+v"(to));
                const int n = cbase + (to >> 2);
                if (n <= nlast) {
                    const size_t o = ((size_t)t * N + n) * F + mb * LG_UNITS + (to & 3) * 8;
                    if (XB_SIG(p.sig_flag)) {           // read by another stream's kernel while this launch is still running: write-through
                        store16_sc1(p.y_hi + o, vhi);
                        store16_sc1(p.y_lo + o, vylo);
                    } else {
                        *reinterpret_cast<uint4 *>(p.y_hi + o) = vhi;
                        *reinterpret_cast<uint4 *>(p.y_lo + o) = vylo;
                    }
                }
            }
        }
        // time slab complete: every wave's output stores are at the coherence point, then one arrival per workgroup; the last
        // one to arrive publishes the slab to the stream that waits on the flag
        if (XB_SIG(p.sig_flag) && s + 1 == sig_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned before = __hip_atomic_fetch_add(p.sig_done + sig_i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before + 1 == (unsigned)(gh * members))          // workgroups beyond the group slots left at the top
                    __hip_atomic_fetch_max(p.sig_flag, p.sig_base + (unsigned)sig_i + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            ++sig_i;
            sig_next = (int)((long long)T * (sig_i + 1) / p.sig_nts);
        }
    }

#ifdef XB_LSTM_STAMPS
    if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 10; ++i) g_lstm_stamps[i] += sStamp[i];
#endif
#pragma unroll 1
    for (int gi = 0; gi < NG; ++gi) {
        if (gi == 1 && !second) break;
        serve(gi);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = cbase + nt * 32 + (lane & 31);
            if (n <= nlast)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    p.c_state[(size_t)n * F + ubase + 2 * rg + hsel] =
                        sC[(wid * 8 + 2 * rg + hsel) * LG_BN + nt * 32 + (lane & 31)];
        }
    }
}

#ifdef XB_WITH_QUAD          // the software-pipelined experiment: diagnostic library only (make diag), never libxnacall.so
#include "../../tools/diag/xb_lstm_quad.h"
#endif

// dynamic LDS of lstm_kernel<KS, nsplit, dual>
template <int KS>
static size_t lstm_lds_bytes(int nsplit, bool dual)
{
    constexpr int F = KS * 16;
    constexpr int KP = F < 128 ? F : 128;
    const int nparts = nsplit == 1 ? 1 : 2;
    const int ng = dual ? 2 : 1;
    const int es = nsplit >= 4 ? 1 : 2, stp = nsplit >= 4 ? 3 : nparts;     // lstm_kernel: ES, STP
    return (size_t)2 * nparts * LG_BN * KP * es + (size_t)stp * 16 * ST_LD * 4 +
           (size_t)ng * (sizeof(float) * LG_UNITS * LG_BN + (size_t)LG_BN * LG_UNITS * 16) + 16 + 80 + 256 * 16 + 128 * 4;
}

template <int KS, int NSPLIT, bool DUAL, bool YALT = false>
hipError_t launch_lstm_v(const xb::LstmParams &p, dim3 grid, size_t lds, hipStream_t stream)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_kernel<KS, NSPLIT, DUAL, YALT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((lstm_kernel<KS, NSPLIT, DUAL, YALT>), grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int KS>
hipError_t launch_lstm_ks(const xb::LstmParams &p, hipStream_t stream)
{
    constexpr int F = KS * 16;
#ifdef XB_WITH_QUAD
    if constexpr (KS == Q_KS) {
        if (p.quad) return launch_lstm_quad(p, stream);
    }
#endif
    const int ngroups = (p.nslab + LG_BN - 1) / LG_BN;
    const bool dual = p.dual != 0;
    const int gh = dual ? (ngroups + 1) / 2 : ngroups;      // workgroup slots (lstm_kernel)
    const int g8 = (gh + 7) & ~7;
    const int members = F / LG_UNITS;
    const size_t lds = lstm_lds_bytes<KS>(p.nsplit, dual);
    const dim3 grid(g8 * members);
    if constexpr (KS % 8 == 0 || KS == 4) {                    // int8-limb pieces: 64 or 128 columns
        if (p.nsplit == 4) return dual ? launch_lstm_v<KS, 4, true>(p, grid, lds, stream) : launch_lstm_v<KS, 4, false>(p, grid, lds, stream);
        if (p.nsplit == 5) return dual ? launch_lstm_v<KS, 5, true>(p, grid, lds, stream) : launch_lstm_v<KS, 5, false>(p, grid, lds, stream);
    } else if (p.nsplit >= 4) {
        return hipErrorInvalidValue;
    }
    // y_alt: the layer output carries the other second part than the exchange image (lstm_kernel YALT)
    if (p.y_alt) {
        if (p.nsplit == 3) return dual ? launch_lstm_v<KS, 3, true, true>(p, grid, lds, stream) : launch_lstm_v<KS, 3, false, true>(p, grid, lds, stream);
        if (p.nsplit == 2) return dual ? launch_lstm_v<KS, 2, true, true>(p, grid, lds, stream) : launch_lstm_v<KS, 2, false, true>(p, grid, lds, stream);
        return hipErrorInvalidValue;
    }
    if (dual) {
        if (p.nsplit == 3) return launch_lstm_v<KS, 3, true>(p, grid, lds, stream);
        if (p.nsplit == 2) return launch_lstm_v<KS, 2, true>(p, grid, lds, stream);
        return launch_lstm_v<KS, 1, true>(p, grid, lds, stream);
    }
    if (p.nsplit == 3) return launch_lstm_v<KS, 3, false>(p, grid, lds, stream);
    if (p.nsplit == 2) return launch_lstm_v<KS, 2, false>(p, grid, lds, stream);
    return launch_lstm_v<KS, 1, false>(p, grid, lds, stream);
}

template <int EPI, int NSPLIT>
hipError_t launch_gemm_ns(const xb::GemmParams &p, hipStream_t stream)
{
    const int MT = (p.M + GBM - 1) / GBM, NT = (p.Nn + GBN - 1) / GBN;
    const int SN = gemm_super_n(NT), SM = 32 / SN;
    const int supers = ((NT + SN - 1) / SN) * ((MT + SM - 1) / SM);     // see gemm_tile_origin
    dim3 grid(8 * 32 * ((supers + 7) / 8)), block(GTHREADS);
    const size_t lds = (size_t)2 * 4 * (NSPLIT == 1 ? 1 : 2) * 128 * 64;    // [4 half-tiles][2 buffers][parts][128 rows x 64 B]
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm8r_kernel<EPI, NSPLIT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((gemm8r_kernel<EPI, NSPLIT>), grid, block, lds, stream, p);
    return hipGetLastError();
}

template <int EPI, int NSPLIT>
hipError_t launch_gemm4p_ns(const xb::GemmParams &p, hipStream_t stream)
{
    const int MT = (p.M + G4_BM - 1) / G4_BM, NT = (p.Nn + G4_BN - 1) / G4_BN;
    const int snw = p.sn > 0 ? p.sn : gemm_super_n(NT);
    const int SN = NT < snw ? NT : snw, SM = 64 / SN;
    const int supers = ((NT + SN - 1) / SN) * ((MT + SM - 1) / SM);     // see gemm4_tile_origin
    dim3 grid(8 * 64 * ((supers + 7) / 8)), block(G4_THREADS);
    size_t lds = (size_t)3 * (NSPLIT == 1 ? 1 : 2) * 128 * 64;             // [3 stages][parts][128 rows x 64 B]
    if (p.one_per_cu) {
        // a launch that shares the chip with the recurrence: more than half of a CU's LDS keeps it to ONE workgroup per CU
        // (half the memory traffic per CU on the recurrence's L2 / fabric), see xb_api.hip run_lstm_layer
        lds = 96 * 1024;
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&gemm4p_kernel<EPI, NSPLIT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((gemm4p_kernel<EPI, NSPLIT>), grid, block, lds, stream, p);
    return hipGetLastError();
}

template <int EPI>
hipError_t launch_gemm_epi(const xb::GemmParams &p, hipStream_t stream)
{
    if (p.b4) {
        switch (p.nsplit) {
        case 1: return launch_gemm4p_ns<EPI, 1>(p, stream);
        case 2: return launch_gemm4p_ns<EPI, 2>(p, stream);
        default: return launch_gemm4p_ns<EPI, 3>(p, stream);
        }
    }
    switch (p.nsplit) {
    case 1: return launch_gemm_ns<EPI, 1>(p, stream);
    case 2: return launch_gemm_ns<EPI, 2>(p, stream);
    default: return launch_gemm_ns<EPI, 3>(p, stream);
    }
}

}  // namespace

namespace xb {

hipError_t launch_conv_front(const ConvFrontParams &p, hipStream_t stream)
{
    if (p.N < 1 || p.T < 1 || p.kp < 16 * p.winlen || p.kp % 32 != 0) return hipErrorInvalidValue;
    const int nq = (CF_TT - 1) * p.stride + p.winlen;
    const size_t lds = sizeof(float) * ((size_t)(nq + 8) + 4 * (nq + 4) + 16 * (size_t)nq + 360);
    if (lds > 60000) return hipErrorInvalidValue;
    dim3 grid((p.T + CF_TT - 1) / CF_TT, p.N), block(256);
    hipLaunchKernelGGL(conv_front_kernel, grid, block, lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_gemm(const GemmParams &p, int epilogue, hipStream_t stream)
{
    if (p.M < 1 || p.Nn < 1 || p.K < GBK || p.K % GBK != 0 || p.lda % 8 != 0 || p.ldb % 8 != 0 ||
        p.lda < p.K || p.ldb < p.K)
        return hipErrorInvalidValue;
    if (p.nsplit < 1 || p.nsplit > 3) return hipErrorInvalidValue;
    switch (epilogue) {
    case EPI_BIAS_F32: return launch_gemm_epi<EPI_BIAS_F32>(p, stream);
    case EPI_SILU_SPLIT: return launch_gemm_epi<EPI_SILU_SPLIT>(p, stream);
    case EPI_TANH_SCALE: return launch_gemm_epi<EPI_TANH_SCALE>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

#ifdef XB_LSTM_STAMPS
void lstm_read_stamps(unsigned long long out[10], bool reset)
{
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lstm_stamps), sizeof(unsigned long long) * 10);
    if (reset) {
        unsigned long long z[10] = {};
        hipMemcpyToSymbol(HIP_SYMBOL(g_lstm_stamps), z, sizeof z);
    }
}
#endif

bool lstm_supported_features(int F)
{
    switch (F) {
    case 32: case 64: case 96: case 128: case 256: case 384: case 512: case 768: return true;
    default: return false;
    }
}
template <int KS, int NSPLIT, bool DUAL>
static int lstm_occupancy_v(size_t lds)
{
    int nb = 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_kernel<KS, NSPLIT, DUAL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm_kernel<KS, NSPLIT, DUAL>, 256, lds);
    return e == hipSuccess ? nb : 0;
}

template <int KS>
static int lstm_occupancy_ks(int nsplit, bool dual)
{
    const size_t lds = lstm_lds_bytes<KS>(nsplit, dual);
    if constexpr (KS % 8 == 0 || KS == 4) {
        if (nsplit == 4) return dual ? lstm_occupancy_v<KS, 4, true>(lds) : lstm_occupancy_v<KS, 4, false>(lds);
        if (nsplit == 5) return dual ? lstm_occupancy_v<KS, 5, true>(lds) : lstm_occupancy_v<KS, 5, false>(lds);
    } else if (nsplit >= 4) {
        return 0;
    }
    if (dual) {
        if (nsplit == 3) return lstm_occupancy_v<KS, 3, true>(lds);
        if (nsplit == 2) return lstm_occupancy_v<KS, 2, true>(lds);
        return lstm_occupancy_v<KS, 1, true>(lds);
    }
    if (nsplit == 3) return lstm_occupancy_v<KS, 3, false>(lds);
    if (nsplit == 2) return lstm_occupancy_v<KS, 2, false>(lds);
    return lstm_occupancy_v<KS, 1, false>(lds);
}

int lstm_resident_per_cu(int F, int nsplit, int dual)
{
    switch (F / 16) {
    case 2: return lstm_occupancy_ks<2>(nsplit, dual != 0);
    case 4: return lstm_occupancy_ks<4>(nsplit, dual != 0);
    case 6: return lstm_occupancy_ks<6>(nsplit, dual != 0);
    case 8: return lstm_occupancy_ks<8>(nsplit, dual != 0);
    case 16: return lstm_occupancy_ks<16>(nsplit, dual != 0);
    case 24: return lstm_occupancy_ks<24>(nsplit, dual != 0);
    case 32: return lstm_occupancy_ks<32>(nsplit, dual != 0);
    case 48: return lstm_occupancy_ks<48>(nsplit, dual != 0);
    default: return 0;
    }
}

#ifdef XB_WITH_QUAD
int lstm_quad_resident_per_cu() { return lstm_quad_occupancy(); }
#else
int lstm_quad_resident_per_cu() { return 0; }
#endif
int lstm_members(int F) { return F / LG_UNITS; }
int lstm_group_chunks() { return LG_BN; }

hipError_t launch_lstm(const LstmParams &p, hipStream_t stream)
{
    if (!lstm_supported_features(p.F) || p.nslab < 1 || p.s_begin < 0 || p.s_end > p.T || p.s_begin >= p.s_end)
        return hipErrorInvalidValue;
    if (p.n0 < 0 || p.n0 + p.nslab > p.N) return hipErrorInvalidValue;
    if (p.nsplit < 1 || p.nsplit > 5) return hipErrorInvalidValue;
    if (p.nsplit >= 4 && (!p.wq1 || !p.wq0 || !p.wscale)) return hipErrorInvalidValue;
#ifdef XB_WITH_QUAD
    if (p.quad && (p.F != Q_F || p.nsplit != 2 || !p.persistent || !p.dual)) return hipErrorInvalidValue;
#else
    if (p.quad) return hipErrorInvalidValue;
#endif
    if (p.sig_flag && (!p.persistent || p.s_begin != 0 || p.s_end != p.T || !p.sig_done || p.sig_nts < 1 || p.sig_nts > p.T))
        return hipErrorInvalidValue;
    switch (p.F / 16) {
    case 2: return launch_lstm_ks<2>(p, stream);
    case 4: return launch_lstm_ks<4>(p, stream);
    case 6: return launch_lstm_ks<6>(p, stream);
    case 8: return launch_lstm_ks<8>(p, stream);
    case 16: return launch_lstm_ks<16>(p, stream);
    case 24: return launch_lstm_ks<24>(p, stream);
    case 32: return launch_lstm_ks<32>(p, stream);
    case 48: return launch_lstm_ks<48>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace xb

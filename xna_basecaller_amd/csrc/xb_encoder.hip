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
#include "xb_enc_common.h"

namespace {

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
// mw .. mw + 127, columns nw .. nw + 63 (nw a multiple of 64); a lane holds column (lane & 31) and 16 rows of each tile:
// register r of a lane in half h = lane >> 5 is row (r & 3) + 8 (r >> 2) + 4 h of the tile (the 32x32 MFMA's own layout) or,
// L16 (the 16x16x32 arithmetic, see acc16_lines), row (r & 7) + 16 (r >> 3) + 8 h.
// The bias is loaded ONCE, ahead of all stores: a load inside the store loop makes every store wait (vmcnt counts stores
// too) for the one before it.
template <bool L16>
__device__ __forceinline__ constexpr int acc_row(int r) { return L16 ? (r & 7) + 16 * (r >> 3) : (r & 3) + 8 * (r >> 2); }

// The 16x16x32 kernels' accumulators -- 8 x 4 tiles of 16x16, a lane holding column (lane & 15) and rows 4 (lane >> 4) + r --
// rearranged so that, as with the 32x32 tiles, every half-wave holds 32 CONSECUTIVE columns of one row (whole 128-byte lines per
// store instruction: what the memory side wants, see gemm8r_kernel): v_permlane16_swap_b32 on register r of two tiles side by
// side leaves [a.row0 b.row0 a.row2 b.row2] and [a.row1 b.row1 a.row3 b.row3] (rows = groups of 16 lanes;
// profiles/r05_mfma_shape_ubench.txt), i.e. columns 32 jp + (lane & 31) of tile rows r + 8 (lane >> 5) and r + 4 + 8 (lane >> 5).
// (inline asm: hipcc 7.2 miscompiles __builtin_amdgcn_permlane16_swap on elements of vector-typed values -- it swaps element 0
//  only and reuses the result for the other three; the accumulators were written by MFMAs that retired long before -- the loop's
//  closing s_waitcnt vmcnt(0) lies in between -- so no wait states are owed here)
__device__ __forceinline__ void acc16_lines(const f32x4 (&a)[8][4], floatx16 (&o)[4][2])
{
    asm volatile("s_nop 7\n\ts_nop 7");
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jp = 0; jp < 2; ++jp)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float u = a[2 * i + ib][2 * jp][r], v = a[2 * i + ib][2 * jp + 1][r];
                    asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(u), "+v"(v));
                    o[i][jp][ib * 8 + r] = u;
                    o[i][jp][ib * 8 + 4 + r] = v;
                }
}

template <int EPI, bool L16 = false>
__device__ __forceinline__ void gemm_epilogue(const xb::GemmParams &p, const floatx16 (&acc)[4][2], int mw, int nw, int lane)
{
    constexpr int HS = L16 ? 8 : 4;              // rows between the two lane halves
    if (mw >= p.M || nw >= p.Nn) return;         // a wave wholly outside the matrix (edge tiles)
#if defined(XB_GEMM_ABL_EPI)                     // timing-only builds (WRONG results): no epilogue at all
    if (p.M > 0) return;
#endif
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
        const int loff = (HS * (lane >> 5)) * ld + (lane & 31);
        if (c0 + 128 <= n) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float *rowp = tile + (size_t)(i * 32 + acc_row<L16>(r)) * ld;
#pragma unroll
                    // non-temporal: 12.6 GB per layer that nobody reads again before the L2 / Infinity Cache have turned over
                    for (int j = 0; j < 2; ++j) __builtin_nontemporal_store(acc[i][j][r] + bj[j], rowp + loff + j * 32);
                }
        } else {
            const int cl = c0 + HS * (lane >> 5);          // chunk index of this lane's row 0 (before wrapping)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = i * 32 + acc_row<L16>(r);
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
        const int loff = (HS * (lane >> 5)) * p.ldc + (lane & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float *rowp = tile + (size_t)(i * 32 + acc_row<L16>(r)) * p.ldc;
#pragma unroll
                for (int j = 0; j < 2; ++j) rowp[loff + j * 32] = p.scale * fast_tanh(acc[i][j][r] + bj[j]);
            }
        return;
    }
    if (EPI == xb::EPI_SILU_SPLIT && interior) {
        // interior tile of the conv3 GEMM: hi as fp16, second part as fp16 residual or q8 bytes; no bounds checks
        const size_t tile = (size_t)mw * p.ldc + nw;       // element offset of the wave's tile
        const int loff = (HS * (lane >> 5)) * p.ldc + (lane & 31);
        unsigned char *q8base = reinterpret_cast<unsigned char *>(p.out_lo) + tile * 2;    // nw is a multiple of 32
        const int qoff = (HS * (lane >> 5)) * p.ldc * 2 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const size_t ro = (size_t)(i * 32 + acc_row<L16>(r)) * p.ldc;
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
                const int m = mw + i * 32 + acc_row<L16>(r) + HS * (lane >> 5);
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
    // S16 (the three-product arithmetic on v_mfma_f32_16x16x32_f16, see gemm4p_kernel): row (lane & 15) of a 16-row tile, the
    // lane's eight k values are cell (lane >> 4) of the row
    constexpr bool S16 = NSPLIT == 3 && XB_GEMM_S16 != 0;
    const unsigned la16 = (unsigned)((lane & 15) * 64 + (((lane >> 4) ^ ((lane >> 2) & 3)) * 16));
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
    /* (S16: ah / al [i4 >> 1][i4 & 1] = 16-row tile i4 of the 64-row half, bh / bl [j2] = 16-column tile j2 of the 32 columns) */
#define G8_READ_A(d, mh)                                                                                  \
    _Pragma("unroll") for (int i2 = 0; i2 < 2; ++i2) {                                                    \
        const unsigned char *t_ = fragA + (d) * HTB + ((mh) * 2 + i2) * 2048;                         \
        _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) {                                                \
            ah[i2][ks] = *reinterpret_cast<const half8 *>(t_ + (S16 ? ks * 1024 + la16 : la[ks]));        \
            if (NSPLIT == 3) al[i2][ks] = *reinterpret_cast<const half8 *>(t_ + PARTB + (S16 ? ks * 1024 + la16 : la[ks])); \
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
            bh[ks] = *reinterpret_cast<const half8 *>(t_ + (S16 ? ks * 1024 + la16 : la[ks]));            \
            if (NSPLIT == 3) bl[ks] = *reinterpret_cast<const half8 *>(t_ + PARTB + (S16 ? ks * 1024 + la16 : la[ks])); \
        }                                                                                                 \
        if (NSPLIT == 2) {                                                                                \
            const v4i x_ = *reinterpret_cast<const v4i *>(t_ + PARTB + lqb[0]);                           \
            const v4i y_ = *reinterpret_cast<const v4i *>(t_ + PARTB + lqb[1]);                           \
            bq = __builtin_shufflevector(x_, y_, 0, 1, 2, 3, 4, 5, 6, 7);                            \
        }                                                                                                 \
    } while (0)
#define G8_F16(i2, ks, n, A_, B_)                                                                         \
    acc[(mh_) * 2 + (i2)][(n)] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A_[(i2)][(ks)], B_[(ks)], acc[(mh_) * 2 + (i2)][(n)], 0, 0, 0)
#define G8_S16(A_, B_, n)                                                                                 \
    _Pragma("unroll") for (int i4 = 0; i4 < 4; ++i4)                                                      \
        _Pragma("unroll") for (int j2 = 0; j2 < 2; ++j2)                                                  \
            acc16[mh_ * 4 + i4][(n) * 2 + j2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_[i4 >> 1][i4 & 1], B_[j2], acc16[mh_ * 4 + i4][(n) * 2 + j2], 0, 0, 0)
#define G8_MFMA(mh, n)                                                                                    \
    do {                                                                                                  \
        constexpr int mh_ = (mh);                                                                         \
        __builtin_amdgcn_s_setprio(1);                                                                    \
        if constexpr (S16) {                                                                              \
            G8_S16(al, bh, n);                                                                            \
            G8_S16(ah, bl, n);                                                                            \
            G8_S16(ah, bh, n);                                                                            \
        } else if (NSPLIT == 2) {                                                                                \
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
    f32x4 acc16[8][4];                  // S16: 8 x 4 tiles of 16x16
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.0f;

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
#undef G8_S16
#undef G8_READ_A
#undef G8_READ_B

    // the epilogue's per-lane indices must not be computed (and kept in registers) ahead of the main loop
    int lane_e = lane, mw_e = m0 + wr * 128, nw_e = n0 + wc * 64;
    asm volatile("" : "+v"(lane_e), "+s"(mw_e), "+s"(nw_e));       // (the tile origin too: its divisions belong behind the loop)
    if constexpr (S16) acc16_lines(acc16, acc);
    gemm_epilogue<EPI, S16>(p, acc, mw_e, nw_e, lane_e);
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
#ifndef XB_GEMM_ABL              // timing-only builds (WRONG results unless 0): bit 0 = every weight reload reads k-tile 0 again (L1-resident), bit 1 = every
#define XB_GEMM_ABL 0            // A-tile request reads k-tile 0 again: what the loop's L2 -> CU traffic costs, instruction stream unchanged
#endif
#ifndef XB_GEMM_S16_V            // A/B builds of the 16x16x32 loop: bit 0 = lo weight pieces reloaded behind their last product (the prologue's order
#define XB_GEMM_S16_V 1          // follows), bit 1 = the second A fragment set requested behind the first product group of phase 0
#endif
#ifndef XB_GEMM_TAIL             // 1 (round 5): the requests the branch-free k loop issues past the last k-tile (two tiles' worth per workgroup, so that
#define XB_GEMM_TAIL 1           // every counted wait keeps its value) fetch one cache line each instead of the last tile again; 0: rounds 3-4 (A/B builds)
#endif
#ifndef XB_GEMM_XTILE            // 1 (round 5, the 16x16x32 loop): the k-tile's barrier sits between its third and fourth MFMA phase instead of at its top, and
#define XB_GEMM_XTILE 1          // the first fragments of tile t + 1 are requested behind it -- their LDS latency runs under the fourth phase's 24 MFMAs; 0: rounds 3-4
#endif
#ifndef XB_GEMM_DMA_ASM          // 1: gemm4p_kernel's A-tile LDS-DMA requests as inline asm (see dma_a); 0: the builtin (A/B builds)
#define XB_GEMM_DMA_ASM 1
#endif
constexpr int G4_BM = 128, G4_BN = 256, G4_THREADS = 256;
#ifdef XB_GEMM_STAMPS
// diagnostic build only: cycle sums over wave 0 of every gemm4p_kernel<*, 3> workgroup, by phase -- 0 prologue (kernel start to the first
// k-tile), 1 top of a k-tile (the tile's A requests landed + barrier), 2 weight pieces landed (+ the first fragment reads issued),
// 3 the tile's MFMA phases, 4 loop end to kernel end (drain + epilogue); [6] = workgroups, [7] = k-tiles
__device__ unsigned long long g_gemm_stamps[8];
#define G4P_STAMP(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); st_acc[(i)] += n_ - st_prev; st_prev = n_; } while (0)
#else
#define G4P_STAMP(i) do { } while (0)
#endif

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
#ifdef XB_GEMM_STAMPS
    unsigned long long st_prev = __builtin_readcyclecounter(), st_acc[5] = {0, 0, 0, 0, 0};
#endif

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
    // `live` false (round 5, XB_GEMM_TAIL): a request past the last k-tile -- issued all the same so that every counted wait keeps
    // its value, but with every lane on the tile's first 16 bytes: one cache line through the vector-memory path instead of sixteen
    auto dma_a = [&](int t, int stage, bool live = true) {
        const size_t kb = (size_t)t * (GBK * 2);
        unsigned char *dst = smem_raw + stage * STB + wid * 1024;
        const unsigned ol[2] = {live ? offs[0] : 0u, live ? offs[1] : 0u};
#pragma unroll
        for (int part = 0; part < NPA; ++part)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#if XB_GEMM_DMA_ASM
                const unsigned m0v = __builtin_amdgcn_readfirstlane(
                    (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)(dst + part * PARTB + h * 4096));
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             ::"v"(ol[h]), "s"((part ? tA_lo : tA_hi) + kb), "s"(m0v) : "memory", "m0");
#else
                __builtin_amdgcn_global_load_lds(
                    (const __attribute__((address_space(1))) void *)((part ? tA_lo : tA_hi) + kb + ol[h]),
                    (__attribute__((address_space(3))) void *)(dst + part * PARTB + h * 4096), 16, 0, 0);
#endif
            }
    };

    // ---- B: this wave's two 32-row blocks; lane byte offsets for block 0 / 1 (the pieces are immediates)
    const unsigned boff = (unsigned)__builtin_amdgcn_readfirstlane((int)(((unsigned)(n0 + wid * 64) >> 5) * NPC * 1024u));
    const unsigned char *const tB = p.b4 + boff;       // wave-uniform by construction: an SGPR pair for the asm loads
    const size_t bks = p.b4_kstride;
    const unsigned voff0 = (unsigned)lane * 16;       // the wave's second 32-row block is NPC KiB further on: in the (scalar) base
    u32x4 bE[2][NPC], bO[2][NPC];               // even / odd k-tile
    // (default cache policy: with the nt bit -- the weights bypassing the L1 -- the five input GEMMs take 65 instead of 47 ms)
#define G4P_LDB(dst, vo, base, j, pc)                                                           \
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(vo), "s"((base) + (j) * (NPC * 1024)), "i"((pc) * 1024))
#define G4P_LDB_GROUP_V(bS, vo, base, pc)                                                       \
    do {                                                                                        \
        G4P_LDB(bS[0][(pc)], vo, base, 0, pc);                                                  \
        G4P_LDB(bS[1][(pc)], vo, base, 1, pc);                                                  \
    } while (0)
#define G4P_LDB_GROUP(bS, base, pc) G4P_LDB_GROUP_V(bS, voff0, base, pc)
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
    // S16 (round 5): the three-product arithmetic on v_mfma_f32_16x16x32_f16 -- the same FLOPs per cycle as 32x32x16, but the chip,
    // which lowers its clock under this kernel's load (1.6 GHz), holds a 13 % higher one on this shape (tools/ubench/mfma_shape.hip,
    // profiles/r05_mfma_shape_ubench.txt; MI355X_MICROARCH.md, DVFS give-back item 7).  A wave's 128 x 64 outputs are 8 x 4 tiles of
    // 16x16; a k-tile is ONE k-step of 32; A fragment of a 16-row tile: row (lane & 15), cell (lane >> 4) of the row (the LDS image
    // and its swizzle are unchanged); the weight image's four pieces of a 32-row block are (hi, lo) x (rows 0-15, 16-31) with lane
    // l = row (l & 15), k 8 (l >> 4) .. + 8 (xb_api.hip: fragment_major).
    constexpr bool S16 = NSPLIT == 3 && XB_GEMM_S16 != 0;
    const unsigned la16 = (unsigned)((lane & 15) * 64 + (((lane >> 4) ^ sw) * 16));

    constexpr bool XT = S16 && XB_GEMM_XTILE != 0 && (XB_GEMM_S16_V & 2) == 0;
    half8 xh[2], xl[2];                         // S16: the A fragments of phase 0 / 2 (XT: requested one tile ahead)
    floatx16 acc[4][2];
    f32x4 acc16[8][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.0f;

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
    /* S16: fragments of the row tiles 2 q, 2 q + 1 of one part; the 24 MFMAs of a phase (rows 32 q .. 32 q + 31, all 64 columns): */
    /* lo*hi, hi*lo, hi*hi per accumulator, eight accumulators between two uses of one                                          */
#define G4P_RD16(d, sa, part, q)                                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                            \
        d[i_] = *reinterpret_cast<const half8 *>((sa) + (part) * PARTB + ((q) * 2 + i_) * 1024 + la16)
#define G4P_M16(A_, q, bS, pc0)                                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 2; ++i_)                                            \
        _Pragma("unroll") for (int j_ = 0; j_ < 2; ++j_)                                        \
            _Pragma("unroll") for (int c_ = 0; c_ < 2; ++c_)                                    \
                acc16[(q) * 2 + i_][2 * j_ + c_] = __builtin_amdgcn_mfma_f32_16x16x32_f16(      \
                    A_[i_], __builtin_bit_cast(half8, bS[j_][(pc0) + c_]), acc16[(q) * 2 + i_][2 * j_ + c_], 0, 0, 0)
#define G4P_S16(AH_, AL_, q, bS)                                                                \
    do {                                                                                        \
        G4P_M16(AL_, q, bS, 0);                                                                 \
        G4P_M16(AH_, q, bS, 2);                                                                 \
        G4P_M16(AH_, q, bS, 0);                                                                 \
    } while (0)
#define G4P_WAIT8(n, bS)                                                                        \
    asm volatile("s_waitcnt vmcnt(%8)" : "+v"(bS[0][0]), "+v"(bS[1][0]), "+v"(bS[0][1]), "+v"(bS[1][1]), \
                 "+v"(bS[0][2]), "+v"(bS[1][2]), "+v"(bS[0][3]), "+v"(bS[1][3]) : "i"(n))
#define G4P_MFMA_BEGIN() do { G4P_SB(); __builtin_amdgcn_s_setprio(1); } while (0)
#define G4P_MFMA_END() do { __builtin_amdgcn_s_setprio(0); G4P_SB(); } while (0)

    // one k-tile: tile t from LDS stage `cur` and register set bS; DMA of tile t + 2 into stage `nxt2`; reload of bS with
    // tile t + 2 (base address b2)
#define G4P_TILE(bS, t)                                                                         \
    do {                                                                                        \
        const unsigned char *const sa = smem_raw + cur * STB;                                   \
        const int t2r_ = (t) + 2 < nk ? (t) + 2 : nk - 1;                                       \
        const int t2_ = (XB_GEMM_ABL & 2) ? 0 : t2r_;       /* timing-only builds, see XB_GEMM_ABL */ \
        const unsigned char *const b2 = tB + (size_t)((XB_GEMM_ABL & 1) ? 0 : t2r_) * bks;      \
        /* the last two tiles have no tile t + 2: their requests keep the counts and fetch one line each (see dma_a) */ \
        const bool live_ = XB_GEMM_TAIL == 0 || (t) + 2 < nk;                                   \
        const unsigned vb_ = live_ ? voff0 : 0u;                                                \
        if ((t) == 0) G4P_STAMP(0);                                                             \
        if constexpr (!XT) {                                                                    \
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * NB + NA) : "memory");                  \
            G4P_SB();                                                                           \
            __builtin_amdgcn_s_barrier();                                                       \
            G4P_SB();                                                                           \
            G4P_STAMP(1);                                                                       \
        }                                                                                       \
        if constexpr (!G4P_LATE_A) dma_a(t2_, nxt2, live_);                                            \
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
            G4P_LDB_GROUP_V(bS, vb_, b2, 0);                                                           \
            G4P_WAIT4(INFL - 4, bS[0][2], bS[0][3], bS[1][2], bS[1][3]);                        \
            G4P_MFMA_BEGIN();                                                                   \
            G4P_F8(q0, 0, bS);                                                                  \
            G4P_MFMA_END();                                                                     \
            G4P_RD_H(g0, sa, 0, 0, 1);                                                          \
            G4P_RD_H(g1, sa, 0, 1, 1);                                                          \
            G4P_MFMA_BEGIN();                                                                   \
            G4P_F8(q1, 1, bS);                                                                  \
            G4P_MFMA_END();                                                                     \
            G4P_LDB_GROUP_V(bS, vb_, b2, 2);                                                           \
            G4P_LDB_GROUP_V(bS, vb_, b2, 3);                                                           \
            G4P_WAIT2(INFL - 2, bS[0][1], bS[1][1]);                                            \
            G4P_MFMA_BEGIN();                                                                   \
            G4P_F16(g0, 0, bS, 1);                                                              \
            G4P_F16(g1, 1, bS, 1);                                                              \
            G4P_MFMA_END();                                                                     \
            G4P_LDB_GROUP_V(bS, vb_, b2, 1);                                                           \
        } else if constexpr (S16) {                                                             \
            /* four phases of 24 MFMAs; the A fragments of phases q and q + 1 in two register sets (x, y), those of phase q + 2  */ \
            /* requested right behind the MFMAs of phase q; all eight B pieces are live until the last phase and are reloaded     */ \
            /* for tile t + 2 behind it -- a whole k-tile (two, with the CU's other workgroup) ahead of their first use.  Issue   */ \
            /* order per tile, hence the counted waits: A(t + 2) behind phase 0, B(t + 2) at the end.                             */ \
            /* XT (round 5): the fragments of phase 0 (xh, xl: loop-carried) were requested behind the barrier inside the      */ \
            /* previous tile (the prologue, for tile 0); that barrier -- every wave has A(t + 1) landed, vmcnt(NB + NA): B(t + 1) */ \
            /* and A(t + 2) are younger, and has finished reading tile t -- sits between phases 2 and 3.  The order of the     */ \
            /* vector-memory instructions, hence every other counted wait, is unchanged.                                         */ \
            half8 yh[2], yl[2];                                                                 \
            constexpr int LA_ = G4P_LATE_A ? NA : 0;                                            \
            if constexpr (!XT) {                                                                \
                G4P_RD16(xl, sa, 1, 0);                                                         \
                G4P_RD16(xh, sa, 0, 0);                                                         \
            }                                                                                   \
            if constexpr (!(XB_GEMM_S16_V & 2)) {                                               \
                G4P_RD16(yl, sa, 1, 1);                                                         \
                G4P_RD16(yh, sa, 0, 1);                                                         \
            }                                                                                   \
            G4P_WAIT8(INFL - 8 - LA_, bS);                                                      \
            G4P_STAMP(2);                                                                       \
            if constexpr (XB_GEMM_S16_V & 2) {                                                  \
                /* the second set's reads behind the first product group of phase 0 */          \
                G4P_MFMA_BEGIN();                                                               \
                G4P_M16(xl, 0, bS, 0);                                                          \
                G4P_MFMA_END();                                                                 \
                G4P_RD16(yl, sa, 1, 1);                                                         \
                G4P_RD16(yh, sa, 0, 1);                                                         \
                G4P_MFMA_BEGIN();                                                               \
                G4P_M16(xh, 0, bS, 2);                                                          \
                G4P_M16(xh, 0, bS, 0);                                                          \
                G4P_MFMA_END();                                                                 \
            } else {                                                                            \
                G4P_MFMA_BEGIN();                                                               \
                G4P_S16(xh, xl, 0, bS);                                                         \
                G4P_MFMA_END();                                                                 \
            }                                                                                   \
            if constexpr (G4P_LATE_A) dma_a(t2_, nxt2, live_);                                         \
            G4P_RD16(xl, sa, 1, 2);                                                             \
            G4P_RD16(xh, sa, 0, 2);                                                             \
            G4P_MFMA_BEGIN();                                                                   \
            G4P_S16(yh, yl, 1, bS);                                                             \
            G4P_MFMA_END();                                                                     \
            G4P_RD16(yl, sa, 1, 3);                                                             \
            G4P_RD16(yh, sa, 0, 3);                                                             \
            G4P_MFMA_BEGIN();                                                                   \
            G4P_S16(xh, xl, 2, bS);                                                             \
            G4P_MFMA_END();                                                                     \
            if constexpr (XT) {                                                                 \
                G4P_STAMP(3);                                                                   \
                asm volatile("s_waitcnt vmcnt(%0)\n\ts_waitcnt lgkmcnt(0)" ::"i"(NB + NA) : "memory"); \
                G4P_SB();                                                                       \
                __builtin_amdgcn_s_barrier();                                                   \
                G4P_SB();                                                                       \
                G4P_STAMP(1);                                                                   \
                const unsigned char *const sn = smem_raw + nxt1 * STB;                          \
                G4P_RD16(xl, sn, 1, 0);                                                         \
                G4P_RD16(xh, sn, 0, 0);                                                         \
            }                                                                                   \
            if constexpr (XB_GEMM_S16_V & 1) {                                                  \
                /* the lo pieces are reloaded behind their last product, the hi pieces at the end */ \
                G4P_MFMA_BEGIN();                                                               \
                G4P_M16(yl, 3, bS, 0);                                                          \
                G4P_M16(yh, 3, bS, 2);                                                          \
                G4P_MFMA_END();                                                                 \
                G4P_LDB_GROUP_V(bS, vb_, b2, 2);                                                       \
                G4P_LDB_GROUP_V(bS, vb_, b2, 3);                                                       \
                G4P_MFMA_BEGIN();                                                               \
                G4P_M16(yh, 3, bS, 0);                                                          \
                G4P_MFMA_END();                                                                 \
                G4P_LDB_GROUP_V(bS, vb_, b2, 0);                                                       \
                G4P_LDB_GROUP_V(bS, vb_, b2, 1);                                                       \
            } else {                                                                            \
                G4P_MFMA_BEGIN();                                                               \
                G4P_S16(yh, yl, 3, bS);                                                         \
                G4P_MFMA_END();                                                                 \
                G4P_LDB_GROUP_V(bS, vb_, b2, 0);                                                       \
                G4P_LDB_GROUP_V(bS, vb_, b2, 1);                                                       \
                G4P_LDB_GROUP_V(bS, vb_, b2, 2);                                                       \
                G4P_LDB_GROUP_V(bS, vb_, b2, 3);                                                       \
            }                                                                                   \
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
                    if constexpr (G4P_LATE_A) dma_a(t2_, nxt2, live_);                                 \
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
                    G4P_LDB_GROUP_V(bS, vb_, b2, 0);                                                   \
                    if (NSPLIT == 3) G4P_LDB_GROUP_V(bS, vb_, b2, 2);                                  \
                } else {                                                                        \
                    G4P_LDB_GROUP_V(bS, vb_, b2, 1);                                                   \
                    if (NSPLIT == 3) G4P_LDB_GROUP_V(bS, vb_, b2, 3);                                  \
                }                                                                               \
            }                                                                                   \
        }                                                                                       \
        if constexpr (!XT) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                   \
        G4P_STAMP(3);                                                                           \
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
        // (XT: barrier k certifies A(k) landed and tile k - 1 read; there are nk + 1 of them, the prologue's and one inside every tile)
        for (int t = 0; t < nk + (XT ? 1 : 0); ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NA) : "memory");      // A(t) landed; A(t + 1) may be in flight
            __builtin_amdgcn_s_barrier();
            dma_a(t + 2 < nk ? t + 2 : nk - 1, c2, XB_GEMM_TAIL == 0 || t + 2 < nk);
            const int c_ = c0; c0 = c1; c1 = c2; c2 = c_;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }

    // (Round 5, measured on real data with cycle stamps -- tools/gemm_stamps.py, profiles/r05_gemm_stamps.txt: a workgroup spends 26 % of its life
    //  outside its MFMA phases (prologue 4, k-tile waits 13, drain + epilogue 9) and its MFMA phases take 1.92x their own pipe time: the
    //  pipe is saturated while both workgroups of a CU are in them.  Starting the workgroup in the CU's second wave slot (HW_REG_HW_ID) a
    //  quarter or half a workgroup life late changes neither the stamps nor the time (54.1 vs 53.2 ms per five input GEMMs): the two are
    //  not in lock-step, their waits coincide because the memory side makes them wait at the same moments.)
    // ---- prologue: A(0) B(0) A(1) B(1) in the loop's issue order
    int cur = 0, nxt1 = 1, nxt2 = 2;
    {
        const int t1 = nk > 1 ? 1 : 0;
        const unsigned char *const b0 = tB, *const b1 = tB + (size_t)t1 * bks;
        dma_a(0, 0);
        if constexpr (NSPLIT == 2) {
            G4P_LDB_GROUP(bE, b0, 0); G4P_LDB_GROUP(bE, b0, 2); G4P_LDB_GROUP(bE, b0, 3); G4P_LDB_GROUP(bE, b0, 1);
        } else if constexpr (S16 && (XB_GEMM_S16_V & 1)) {
            G4P_LDB_GROUP(bE, b0, 2); G4P_LDB_GROUP(bE, b0, 3); G4P_LDB_GROUP(bE, b0, 0); G4P_LDB_GROUP(bE, b0, 1);
        } else {
            G4P_LDB_GROUP(bE, b0, 0); if (NSPLIT == 3) G4P_LDB_GROUP(bE, b0, 2);
            G4P_LDB_GROUP(bE, b0, 1); if (NSPLIT == 3) G4P_LDB_GROUP(bE, b0, 3);
        }
        dma_a(t1, 1);
        if constexpr (NSPLIT == 2) {
            G4P_LDB_GROUP(bO, b1, 0); G4P_LDB_GROUP(bO, b1, 2); G4P_LDB_GROUP(bO, b1, 3); G4P_LDB_GROUP(bO, b1, 1);
        } else if constexpr (S16 && (XB_GEMM_S16_V & 1)) {
            G4P_LDB_GROUP(bO, b1, 2); G4P_LDB_GROUP(bO, b1, 3); G4P_LDB_GROUP(bO, b1, 0); G4P_LDB_GROUP(bO, b1, 1);
        } else {
            G4P_LDB_GROUP(bO, b1, 0); if (NSPLIT == 3) G4P_LDB_GROUP(bO, b1, 2);
            G4P_LDB_GROUP(bO, b1, 1); if (NSPLIT == 3) G4P_LDB_GROUP(bO, b1, 3);
        }
    }
    if constexpr (XT) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * NB + NA) : "memory");      // A(0) landed
        G4P_SB();
        __builtin_amdgcn_s_barrier();
        G4P_SB();
        G4P_RD16(xl, smem_raw, 1, 0);
        G4P_RD16(xh, smem_raw, 0, 0);
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
#undef G4P_LDB_GROUP_V
#undef G4P_WAIT2
#undef G4P_WAIT4
#undef G4P_RD_H
#undef G4P_RD16
#undef G4P_M16
#undef G4P_S16
#undef G4P_WAIT8
#undef G4P_RD_Q
#undef G4P_F16
#undef G4P_F8
#undef G4P_MFMA_BEGIN
#undef G4P_MFMA_END
#undef G4P_SB

    int lane_e = lane, mw_e = m0, nw_e = n0 + wid * 64;
    asm volatile("" : "+v"(lane_e), "+s"(mw_e), "+s"(nw_e));
    if constexpr (S16) acc16_lines(acc16, acc);
    gemm_epilogue<EPI, S16>(p, acc, mw_e, nw_e, lane_e);
#ifdef XB_GEMM_STAMPS
    if (NSPLIT == 3) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the stores have left the wave
        G4P_STAMP(4);
        if (tid == 0) {
            for (int i = 0; i < 5; ++i) atomicAdd(&g_gemm_stamps[i], st_acc[i]);
            atomicAdd(&g_gemm_stamps[5], (unsigned long long)(__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 1u));   // workgroups in an odd wave slot
            atomicAdd(&g_gemm_stamps[6], 1ull);
            atomicAdd(&g_gemm_stamps[7], (unsigned long long)nk);
        }
    }
#endif
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

#ifdef XB_GEMM_STAMPS
void gemm_read_stamps(unsigned long long out[8], bool reset)
{
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * 8);
    if (reset) {
        unsigned long long z[8] = {};
        hipMemcpyToSymbol(HIP_SYMBOL(g_gemm_stamps), z, sizeof z);
    }
}
#endif

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

}  // namespace xb

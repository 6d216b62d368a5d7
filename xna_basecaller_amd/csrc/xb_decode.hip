// xb_decode.hip -- CRF posterior + max-plus decode for gfx950 (MI355X).
//
// Replaces SeqdistModel.decode_batch (ub-bonito/bonito/crf/model.py:215-218) =
//   seqdist posteriors (Log semiring fwd/bwd scans, crf/model.py:41-46)
//   -> +1e-8 -> log -> seqdist posteriors with the Max semiring -> argmax % (n_base+1)
//   (crf/model.py:92-95), then path_to_str (crf/model.py:97-100) and the left-pack of
//   compute_scores (crf/basecall.py:60-67).
//
// One workgroup per chunk; a CRF state is served by LPS = 1 or 2 adjacent lanes that split its E = nb+1 edges in
// two halves.  The state vectors (alpha, beta, max-plus alpha/beta) live in LDS and are exchanged once per time step
// behind a single LDS-only barrier.  Three sweeps over the scores:
//   1. Log forward           (write alpha)
//   2. Log backward fused with Max backward (read alpha; write bmax) -- also writes the log-posteriors
//      Q = log(P + 1e-8) of every edge, (T, N, S*E) fp32, scattered by destination edge into an LDS row and stored as
//      coalesced rows one step late
//   3. Max forward over Q + per-step arg-max of the max-marginals (read Q, bmax), then pack.
//
// What bounds a step is the LENGTH OF ITS DEPENDENCY CHAIN, not bytes or instruction count (a dependent VALU
// instruction issues every ~8 cycles, independent ones every ~2.4 per SIMD: tools/valu_probe.hip), so the kernel is
// built to keep the chain short:
//   * sweeps 1 and 3 are "destination owned": a lane's score / Q values are E (or E/2) CONTIGUOUS floats of the row,
//     loaded straight into a register ring RDEPTH steps ahead (no LDS staging); the state vector is kept in LDS in a
//     transposed order, position (i % hi) * nb + i / hi, so that the nb source states of a destination are contiguous;
//     a state's own previous value (the stay edge) never leaves its registers;
//   * sweep 2 is "source owned"; its strided score reads go through an LDS-staged row, and the state -> lane map is
//     permuted (lane group = nb sources that share their destinations) so that those reads are at most 2-way conflicted;
//   * reductions use v_max_f32_dpp / v_add_f32_dpp directly (hipcc emits v_mov_b32_dpp + a separate op + s_nops for the
//     update_dpp builtin); the per-step arg-max is a wave max + ballot (lane order == flat edge order, so the lowest
//     matching lane IS the lowest flat index), per-wave partials go to a 128-step LDS ring and are finalised 64 steps
//     at a time, one step per lane;
//   * the log polynomial is evaluated in Estrin form (depth 4 instead of 8).
//
// Floating-point contract (shared by specification with oracle/xna_oracle.c, written independently): IEEE binary32,
// no contraction (this file is built with -ffp-contract=off), fma only where __builtin_fmaf is written, exp/log are the
// fixed polynomials below, logsumexp = max / exps summed in edge order / log, ties resolve to the lowest
// flat edge index.  The recursions stay in the log domain on purpose (see the oracle's header).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <type_traits>

#include "xb_internal.h"
#include "xb_math.h"

namespace {


// ---- the contract's exp / log, scalar and two-wide.  The two-wide forms run the SAME IEEE operations on both
// elements with v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 (a plain wave64 VALU instruction and a packed one both occupy
// the SIMD for 4 cycles, and the kernel is bound by VALU issue); only the integer steps stay per element.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float x) { return (f32x2){x, x}; }
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

__device__ __forceinline__ f32x2 xb_expf2(f32x2 x)
{
    x.x = __builtin_amdgcn_fmed3f(x.x, -87.0f, 88.0f);
    x.y = __builtin_amdgcn_fmed3f(x.y, -87.0f, 88.0f);
    const f32x2 t = fma2(x, splat2(1.44269504088896341f), splat2(XB_EXP_MAGIC));
    const f32x2 n = t - splat2(XB_EXP_MAGIC);
    f32x2 r = fma2(n, splat2(-0.693359375f), x);
    r = fma2(n, splat2(2.12194440e-4f), r);
    f32x2 p = splat2(1.9875691500e-4f);
    p = fma2(p, r, splat2(1.3981999507e-3f));
    p = fma2(p, r, splat2(8.3334519073e-3f));
    p = fma2(p, r, splat2(4.1665795894e-2f));
    p = fma2(p, r, splat2(1.6666665459e-1f));
    p = fma2(p, r, splat2(5.0000001201e-1f));
    const f32x2 r2 = r * r;
    const f32x2 y = fma2(p, r2, r) + splat2(1.0f);
    const f32x2 sc = {xb_exp_scale(t.x), xb_exp_scale(t.y)};
    return y * sc;
}

__device__ __forceinline__ f32x2 xb_logf2(f32x2 x)
{
    f32x2 f, fe;
    {
        float f0, f1, e0, e1;
        xb_log_reduce(x.x, f0, e0);
        xb_log_reduce(x.y, f1, e1);
        f = (f32x2){f0, f1};
        fe = (f32x2){e0, e1};
    }
    const f32x2 z = f * f;
    const f32x2 z2 = z * z;
    const f32x2 z4 = z2 * z2;
    const f32x2 q01 = fma2(splat2(-2.4999993993e-1f), f, splat2(3.3333331174e-1f));
    const f32x2 q23 = fma2(splat2(-1.6668057665e-1f), f, splat2(2.0000714765e-1f));
    const f32x2 q45 = fma2(splat2(-1.2420140846e-1f), f, splat2(1.4249322787e-1f));
    const f32x2 q67 = fma2(splat2(-1.1514610310e-1f), f, splat2(1.1676998740e-1f));
    const f32x2 q03 = fma2(q23, z, q01);
    const f32x2 q47 = fma2(q67, z, q45);
    const f32x2 q07 = fma2(q47, z2, q03);
    const f32x2 p = fma2(splat2(7.0376836292e-2f), z4, q07);
    f32x2 y = (f * z) * p;
    y = fma2(fe, splat2(-2.12194440e-4f), y);
    y = fma2(splat2(-0.5f), z, y);
    f32x2 r = f + y;
    r = fma2(fe, splat2(0.693359375f), r);
    return r;
}

// log of N values, two at a time (the odd last one alone)
template <int N>
__device__ __forceinline__ void xb_log_n(const float (&x)[N], float (&y)[N])
{
#pragma unroll
    for (int r = 0; r + 1 < N; r += 2) {
        const f32x2 v = xb_logf2((f32x2){x[r], x[r + 1]});
        y[r] = v.x;
        y[r + 1] = v.y;
    }
    if (N & 1) y[N - 1] = xb_logf(x[N - 1]);
}

// exp of N values, two at a time (the odd last one alone)
template <int N>
__device__ __forceinline__ void xb_exp_n(const float (&x)[N], float (&y)[N])
{
#pragma unroll
    for (int r = 0; r + 1 < N; r += 2) {
        const f32x2 v = xb_expf2((f32x2){x[r], x[r + 1]});
        y[r] = v.x;
        y[r + 1] = v.y;
    }
    if (N & 1) y[N - 1] = xb_expf(x[N - 1]);
}

__device__ __forceinline__ float maxf(float a, float b) { return b > a ? b : a; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory, i.e. emits
// s_waitcnt vmcnt(0), which would drain the prefetch ring (a full HBM latency) at every time step.
// Inside the sweeps the only cross-thread data is in LDS; the global stashes are re-read by the thread that
// wrote them (and the sweeps are separated by real __syncthreads()).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- cross-lane arithmetic on DPP, one instruction per step ---------------------------------------------------
// (2 wait states between a VALU write of a register and a DPP read of it: the s_nop 1 in front of every step)
// max over the 64 lanes, valid in lane 63; lanes without a DPP source keep their own value
__device__ __forceinline__ float wave_max63(float v)
{
    asm volatile(
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return v;
}
// the same for four independent values at once: the four chains interleave, which also provides the two wait states
// between a register's write and its next DPP read (no s_nop between the levels)
__device__ __forceinline__ void wave_max63_x4(float &a, float &b, float &c, float &d)
{
#define XB_L4(ctrl) \
    "v_max_f32_dpp %0, %0, %0 " ctrl "\n\tv_max_f32_dpp %1, %1, %1 " ctrl "\n\t" \
    "v_max_f32_dpp %2, %2, %2 " ctrl "\n\tv_max_f32_dpp %3, %3, %3 " ctrl "\n\t"
    asm volatile("s_nop 1\n\t"
                 XB_L4("row_shr:1 row_mask:0xf bank_mask:0xf")
                 XB_L4("row_shr:2 row_mask:0xf bank_mask:0xf")
                 XB_L4("row_shr:4 row_mask:0xf bank_mask:0xf")
                 XB_L4("row_shr:8 row_mask:0xf bank_mask:0xf")
                 XB_L4("row_bcast:15 row_mask:0xa bank_mask:0xf")
                 XB_L4("row_bcast:31 row_mask:0xc bank_mask:0xf")
                 "s_nop 1"
                 : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
#undef XB_L4
}
// max / sum with the partner lane of a pair (lanes 2s, 2s+1): both lanes receive the result
__device__ __forceinline__ float pair_max(float v)
{
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_nop 0" : "+v"(v));
    return v;
}
// Sum, in edge order 0..E-1, of the E values a lane pair holds (lane 0: edges [0, H) in ex[0..], lane 1: edges [H, E)):
// every term is fetched from the lane that holds it by a fused broadcast-add, so both lanes run the identical chain
// e_0, + e_1, .. + e_{E-1}.  ONE asm statement: all ex[] are inputs, i.e. written before it starts (the leading s_nop
// covers the VALU-write -> DPP-read wait states of the last one); inside, only `s` is written and it is never DPP-read.
#define XB_BC0 "quad_perm:[0,0,2,2] row_mask:0xf bank_mask:0xf\n\t"
#define XB_BC1 "quad_perm:[1,1,3,3] row_mask:0xf bank_mask:0xf\n\t"
template <int E>
__device__ __forceinline__ float pair_sum_ordered(const float (&ex)[(E + 1) / 2])
{
    float s;
    static_assert(E >= 5 && E <= 7, "pair_sum_ordered is written out for 4, 5 and 6 bases");
    if constexpr (E == 5)
        asm volatile("s_nop 1\n\t"
                     "v_mov_b32_dpp %0, %1 " XB_BC0 "v_add_f32_dpp %0, %2, %0 " XB_BC0 "v_add_f32_dpp %0, %3, %0 " XB_BC0
                     "v_add_f32_dpp %0, %1, %0 " XB_BC1 "v_add_f32_dpp %0, %2, %0 " XB_BC1 "s_nop 0"
                     : "=&v"(s) : "v"(ex[0]), "v"(ex[1]), "v"(ex[2]));
    else if constexpr (E == 6)
        asm volatile("s_nop 1\n\t"
                     "v_mov_b32_dpp %0, %1 " XB_BC0 "v_add_f32_dpp %0, %2, %0 " XB_BC0 "v_add_f32_dpp %0, %3, %0 " XB_BC0
                     "v_add_f32_dpp %0, %1, %0 " XB_BC1 "v_add_f32_dpp %0, %2, %0 " XB_BC1 "v_add_f32_dpp %0, %3, %0 " XB_BC1 "s_nop 0"
                     : "=&v"(s) : "v"(ex[0]), "v"(ex[1]), "v"(ex[2]));
    else
        asm volatile("s_nop 1\n\t"
                     "v_mov_b32_dpp %0, %1 " XB_BC0 "v_add_f32_dpp %0, %2, %0 " XB_BC0 "v_add_f32_dpp %0, %3, %0 " XB_BC0
                     "v_add_f32_dpp %0, %4, %0 " XB_BC0
                     "v_add_f32_dpp %0, %1, %0 " XB_BC1 "v_add_f32_dpp %0, %2, %0 " XB_BC1 "v_add_f32_dpp %0, %3, %0 " XB_BC1 "s_nop 0"
                     : "=&v"(s) : "v"(ex[0]), "v"(ex[1]), "v"(ex[2]), "v"(ex[3]));
    return s;
}
#undef XB_BC0
#undef XB_BC1

// CNT consecutive floats from a 4-byte aligned address in the widest pieces (global_load_dwordx4 / x2 / dword)
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int CNT>
__device__ __forceinline__ void load_seg(const float *p, float (&r)[CNT])
{
    int o = 0;
    if constexpr (CNT >= 4) {
        const f32x4u v = *reinterpret_cast<const f32x4u *>(p);
        r[0] = v[0]; r[1] = v[1]; r[2] = v[2]; r[3] = v[3];
        o = 4;
    }
    if constexpr ((CNT & 3) >= 2) {
        const f32x2u v = *reinterpret_cast<const f32x2u *>(p + (CNT & ~3));
        r[(CNT & ~3)] = v[0]; r[(CNT & ~3) + 1] = v[1];
        o = (CNT & ~3) + 2;
    }
    if constexpr (CNT & 1) r[CNT - 1] = p[CNT - 1];
    (void)o;
}

// Ring depths and occupancy (round 4, tools/r04_decode_variants.sh, profiles/r04_decode_variants.txt): with rings of 4 / 8 steps
// (rounds 2-3: 8 / 16) the nb = 6 kernel fits 168 registers, i.e. three workgroups per CU instead of two (LDS allows three);
// the pass over a co-scheduled pair's 1024 chunks gets 1.9 % (nb 6) / 3.5 % (nb 5) faster, 512 chunks are unchanged, and the
// 1024-thread variants spill less.  The hint is a MINIMUM of three waves per SIMD (a 1024-thread workgroup needs four).
#ifndef XB_DEC_RDEPTH          // (tuning builds override these)
#define XB_DEC_RDEPTH 4
#endif
#ifndef XB_DEC_RDEPTH13
#define XB_DEC_RDEPTH13 8
#endif
#ifndef XB_DEC_WAVES_ATTR
#define XB_DEC_WAVES_ATTR __attribute__((amdgpu_waves_per_eu(3)))
#endif
constexpr int RDEPTH = XB_DEC_RDEPTH;       // register-ring depth (time steps in flight), sweep 2 (a whole staged row per step)
constexpr int RDEPTH13 = XB_DEC_RDEPTH13;   // sweeps 1 and 3: a lane's ring entry is only E (or E/2) floats
constexpr int LRING = 128;    // arg-max partial ring (steps); finalised 64 at a time

// coalesced row staging for sweep 2: NR 16-byte groups per thread, loaded RDEPTH steps ahead, written to LDS per step
template <int NR, int BS>
struct RowRegs {
    f32x4 r[NR];
    __device__ __forceinline__ void load(const float *row, int lim, int tid)
    {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int g = (tid + BS * i) * 4;
            r[i] = *reinterpret_cast<const f32x4 *>(row + (g < lim ? g : 0));
        }
    }
    __device__ __forceinline__ void store(float *lds, int tid) const
    {
#pragma unroll
        for (int i = 0; i < NR; ++i) *reinterpret_cast<f32x4 *>(lds + (tid + BS * i) * 4) = r[i];
    }
};
// the same with 4-byte loads for rows that are not 16-byte aligned (scores handed in with an odd row length)
template <int NR, int BS>
struct RowRegs1 {
    float r[NR];
    __device__ __forceinline__ void load(const float *row, int lim, int tid)
    {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int g = tid + BS * i;
            r[i] = row[g < lim ? g : 0];
        }
    }
    __device__ __forceinline__ void store(float *lds, int tid) const
    {
#pragma unroll
        for (int i = 0; i < NR; ++i) lds[tid + BS * i] = r[i];
    }
};

// Edge bookkeeping of a lane.  A state's E edges are split in two halves [0, H) and [H, E), H = ceil(E/2).
// LPS == 1: the lane owns all E edges (local index r = edge).  LPS == 2: lane `ph` of the pair owns half `ph`
// (local index r = edge - ph * H; the second half has E - H <= H edges, its last local index may be invalid).
template <int E, int LPS>
struct Edges {
    static constexpr int H = (E + 1) / 2;
    static constexpr int EPER = LPS == 1 ? E : H;
};

// NB bases, BS threads (multiple of 64, >= LPS * S), HB: the scores carry the blank column, LPS lanes per state,
// VW: 4 = 16-byte staging loads in sweep 2 (row stride and base 16-byte aligned), 1 = 4-byte loads
// SCAN = true: the xb_crf_scans variant (optional beta / posterior outputs, early return after sweep 1 or 2); the decode
// proper is compiled without those paths (measured: 1-2 % of the decode time when they are run-time branches).
template <int NB, int BS, bool HB, int LPS, int VW, bool SCAN>
__global__ __launch_bounds__(BS) XB_DEC_WAVES_ATTR void crf_decode_kernel(xb::DecodeParams p)
{
    constexpr int E = NB + 1;
    constexpr int H = Edges<E, LPS>::H;
    constexpr int EPER = Edges<E, LPS>::EPER;
    constexpr int NW = BS / 64;
    // Score columns a lane loads per row in sweep 1.  With the blank column: its EPER edge columns (the second lane of a
    // pair loads the LAST EPER columns of the state so that it never reads past the row).  Without it: all NB columns
    // (LPS 1), or E - H columns (LPS 2: first lane columns 0.., second lane columns H-1..).
    constexpr int CPL = HB ? EPER : (LPS == 1 ? NB : E - H);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int S = p.S, hi = p.hi, T = p.T, N = p.N, cin = p.cin, ldq = p.ldq;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = blockIdx.x;
    const int st = tid / LPS;
    const int ph = LPS == 1 ? 0 : (tid & 1);      // half served by this lane
    const bool act = st < S;
    const int stc = act ? st : S - 1;              // clamped state (always a valid column)
    const int kcnt = LPS == 1 ? E : (ph == 0 ? H : E - H);   // valid local edge indices: r < kcnt

    // staging pieces per thread: BS * NRS * VW >= S * E >= cin.  BS >= LPS * S, so NRS * VW * LPS >= E is enough (round 5: with two
    // lanes per state half the pieces -- nb = 5: one 16-byte piece per thread instead of two, the staged rows 4 KiB instead of
    // 8 KiB, 25 KB of LDS per workgroup instead of 41 KB: FOUR workgroups per CU, as the registers allow, instead of three)
    constexpr int NRS = VW == 4 ? (E + 4 * LPS - 1) / (4 * LPS) : (E + LPS - 1) / LPS;
    constexpr int cpad = BS * NRS * VW;                // staged row (floats)
    const int lim_m = VW == 4 ? (cin + 3) & ~3 : cin;  // vector loads may touch the row's padding columns
    const int Tpad = (T + RDEPTH13 - 1) / RDEPTH13 * RDEPTH13;   // multiple of both ring depths

    float *sM = reinterpret_cast<float *>(smem_raw);             // [2][cpad]  staged score row (sweep 2)
    float *sQ = sM + 2 * cpad;                                    // [2][cpad]  Q rows being assembled (sweep 2)
    float *sA = sQ + 2 * cpad;                                    // [2][S]  alpha (transposed order) / beta
    float *sX = sA + 2 * S;                                       // [2][S]  max-plus alpha (transposed order) / beta
    float *sG = sX + 2 * S;                                       // [S]     scratch (logZ)
    float *sRv = sG + S;                                          // [LRING][NW] arg-max partials: value
    int *sRi = reinterpret_cast<int *>(sRv + LRING * NW);         // [LRING][NW] arg-max partials: flat edge index
    float *sBc = reinterpret_cast<float *>(sRi + LRING * NW);     // [4] broadcast scratch
    int8_t *sLab = reinterpret_cast<int8_t *>(sBc + 4);           // [T]

    const float *sc = p.scores + (size_t)n * p.ld;
    const size_t tstride = (size_t)N * p.ld;
    float *alpha = p.alpha + (size_t)n * S;
    float *bmax = p.bmax + (size_t)n * S;
    const size_t sstride = (size_t)N * S;
    float *qrow = p.qbuf + (size_t)n * ldq;
    const size_t qstride = (size_t)N * ldq;
    const float blank = p.blank;

    // logsumexp tail shared by sweeps 1 and 2: x[r] (invalid entries = -inf), mx = max over the state's edges.
    // The exps are summed in edge order 0..E-1 (the contract's order; LPS == 2: pair_sum_ordered).
    auto lse_tail = [&](const float (&x)[EPER], float mx) -> float {
        float d[EPER], ex[EPER];
#pragma unroll
        for (int r = 0; r < EPER; ++r) d[r] = x[r] - mx;
        xb_exp_n<EPER>(d, ex);
        float s;
        if constexpr (LPS == 1) {
            s = ex[0];
#pragma unroll
            for (int r = 1; r < EPER; ++r) s += ex[r];
        } else {
            s = pair_sum_ordered<E>(ex);
        }
        return mx + xb_logf(s);
    };

    // slot of state i in the transposed state vector of the destination-owned sweeps: the nb sources
    // (k-1) * hi + j / nb, k = 1..nb, of a destination j are the contiguous slots (j / nb) * nb + (k - 1)
    const int dj = stc, djq = dj / NB;
    const int tpos = (dj % hi) * NB + dj / hi;
    // LDS slot read for local edge r: r0slot for r == 0 (the stay edge = the state's own slot for the first half),
    // sbase + r for r >= 1
    const int sbase = djq * NB + (ph == 0 ? -1 : H - 1);
    const int r0slot = ph == 0 ? tpos : djq * NB + H - 1;

    // ------------------------------------------------------------------ sweep 1: Log forward
    {
        const int j = dj;
        // first score column this lane loads, and the register index of local edge r: mv[r + moff]
        const int col0 = HB ? (ph == 0 ? j * E : j * E + E - EPER) : (ph == 0 ? j * NB : j * NB + H - 1);
        if (tid < S) { sA[tid] = 0.0f; alpha[tid] = 0.0f; }        // alpha_0 = 0 in any order
        float ring[RDEPTH13][CPL];
#pragma unroll
        for (int d = 0; d < RDEPTH13; ++d) load_seg<CPL>(sc + (size_t)(d < T ? d : T - 1) * tstride + col0, ring[d]);
        float aown = 0.0f;
        for (int t0 = 0; t0 < Tpad; t0 += RDEPTH13) {
#pragma unroll
          for (int d = 0; d < RDEPTH13; ++d) {
            const int t = t0 + d;                                // >= T in the padding iterations
            float mv[CPL];
#pragma unroll
            for (int r = 0; r < CPL; ++r) mv[r] = ring[d][r];
            load_seg<CPL>(sc + (size_t)(t + RDEPTH13 < T ? t + RDEPTH13 : T - 1) * tstride + col0, ring[d]);
            lds_barrier();
            if (t < T) {                                         // block-uniform
                const float *a0 = sA + (t & 1) * S;
                float x[EPER];
                float mx = -__builtin_inff();
#pragma unroll
                for (int r = 0; r < EPER; ++r) {
                    // score of local edge r.  Register index: with the blank column first half r, second half
                    // r + (EPER - (E - H)); without it first half r - 1 (r >= 1; r == 0 is the constant blank), second
                    // half r.  Indices are compile-time per half, the half is selected per lane.
                    constexpr int SH = EPER - (E - H);           // 0 or 1
                    float m0v, m1v;                              // value if this lane serves half 0 / half 1
                    if (HB) {
                        m0v = mv[r < CPL ? r : CPL - 1];
                        m1v = mv[r + SH < CPL ? r + SH : CPL - 1];
                    } else {
                        m0v = r == 0 ? blank : mv[r - 1 < CPL ? r - 1 : CPL - 1];
                        m1v = mv[r < CPL ? r : CPL - 1];
                    }
                    const float mk = (LPS == 1 || ph == 0) ? m0v : m1v;
                    float ak;
                    if (r == 0) ak = LPS == 1 ? aown : a0[r0slot];
                    else ak = a0[sbase + r];
                    x[r] = r < kcnt ? mk + ak : -__builtin_inff();
                    mx = maxf(mx, x[r]);
                }
                if (LPS == 2) mx = pair_max(mx);
                const float v = lse_tail(x, mx);
                aown = v;
                if (ph == 0 && act) {
                    sA[((t + 1) & 1) * S + tpos] = v;
                    alpha[(size_t)(t + 1) * sstride + j] = v;
                }
            }
          }
        }
        __syncthreads();
    }

    // logZ = logsumexp_j alpha_T[j], summed in order j = 0..S-1 (alpha_T sits in LDS in transposed order)
    {
        const float *aT = sA + (T & 1) * S;
        const int tp = tid < S ? (tid % hi) * NB + tid / hi : 0;
        const float mine = aT[tp];
        const float wm = wave_max63(tid < S ? mine : -__builtin_inff());
        if (lane == 63) sRv[wave] = wm;                          // the partial ring is free until sweep 3
        __syncthreads();
        float mx = sRv[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) mx = maxf(mx, sRv[w]);
        if (tid < S) sG[tid] = xb_expf(mine - mx);
        __syncthreads();
        if (tid == 0) {
            float s = sG[0];
            for (int j = 1; j < S; ++j) s += sG[j];
            const float lz = mx + xb_logf(s);
            sBc[1] = lz;
            if (p.logz) p.logz[n] = lz;
        }
        __syncthreads();
    }
    const float logZ = sBc[1];
    if (SCAN && p.stop_after == 1) return;      // forward scores / partition function only (uniform)
#ifdef XB_LSTM_STAMPS
    if (p.debug_stop == 1) return;   // diagnostic build only: time sweep 1 alone
#endif

    // -------------------------------------------- sweep 2: Log backward + Max backward (fused)
    {
        // source state of this lane (pair), permuted: consecutive groups of NB sources share their NB destinations
        // (i = (st % NB) * hi + st / NB), so the strided reads of the staged row are contiguous inside a group
        const int i = (stc % NB) * hi + stc / NB;
        const int kk = stc % NB + 1;                               // = i / hi + 1: column of the new-base edges
        const int jb = (stc / NB) * NB;                            // = (i % hi) * NB: first destination
        float *beta_out = SCAN && p.beta_out ? p.beta_out + (size_t)n * S : nullptr;     // uniform
        const bool post_mode = SCAN && p.post_mode != 0;
        if (tid < S) {
            sA[(T & 1) * S + tid] = 0.0f;
            sX[(T & 1) * S + tid] = 0.0f;
            bmax[(size_t)T * sstride + tid] = 0.0f;
            if (beta_out) beta_out[(size_t)T * sstride + tid] = 0.0f;
        }
        // per-lane constants of local edge r: destination state and staged-row index of its score
        // (e = 0: stay, j = i, column 0; e >= 1: j = jb + e - 1, column kk)
        int dstj[EPER], midx[EPER], qidx[EPER];
        const bool stay0 = LPS == 1 || ph == 0;                    // local edge 0 is the stay edge
#pragma unroll
        for (int r = 0; r < EPER; ++r) {
            const int e = (LPS == 1 ? 0 : ph * H) + r;
            const bool val = r < kcnt;
            const int j = e == 0 ? i : (val ? jb + e - 1 : jb);
            dstj[r] = j;
            midx[r] = HB ? j * E + (e == 0 ? 0 : kk) : (e == 0 ? 0 : j * NB + kk - 1);
            qidx[r] = j * E + ((r == 0 && stay0) ? 0 : kk);       // slot of the edge's Q value in the row being assembled
        }
        int wi = i;                                                // slot of this lane's state in the beta vectors
#ifdef XB_LSTM_STAMPS
        // diagnostic build only (XB_DECODE_LINEAR_LDS=1, results are WRONG): every per-edge LDS access of this sweep at a
        // lane-linear address, i.e. free of bank conflicts -- what the conflicts cost at most (profiles/r04_decode_lds_conflicts.txt)
        if (p.debug_lds) {
#pragma unroll
            for (int r = 0; r < EPER; ++r) {
                dstj[r] = tid % S;
                midx[r] = (tid + BS * r) % cpad;
                qidx[r] = (tid + BS * r) % cpad;
            }
            wi = tid % S;
        }
#endif
        // the Q row of step t is complete once every thread has passed the barrier of step t-1:
        // it is stored (coalesced 16-byte groups) during iteration t-1
        auto store_qrow = [&](int t) {
            const float *src = sQ + (t & 1) * cpad;
            float *dst = qrow + (size_t)t * qstride;
            for (int g = tid * 4; g < ldq; g += BS * 4)
                *reinterpret_cast<f32x4 *>(dst + g) = *reinterpret_cast<const f32x4 *>(src + g);
        };
        typename std::conditional<VW == 4, RowRegs<NRS, BS>, RowRegs1<NRS, BS>>::type ring[RDEPTH];
        float aring[RDEPTH];
#pragma unroll
        for (int d = 0; d < RDEPTH; ++d) {
            const int t = T - 1 - d >= 0 ? T - 1 - d : 0;
            ring[d].load(sc + (size_t)t * tstride, lim_m, tid);
            aring[d] = alpha[(size_t)t * sstride + i];
        }
        for (int s0 = 0; s0 < Tpad; s0 += RDEPTH) {
#pragma unroll
          for (int d = 0; d < RDEPTH; ++d) {
            const int t = T - 1 - (s0 + d);                      // < 0 in the padding iterations
            float *m = sM + (t & 1) * cpad;
            ring[d].store(m, tid);
            const float a0 = aring[d];
            {
                const int tn = t - RDEPTH >= 0 ? t - RDEPTH : 0;
                ring[d].load(sc + (size_t)tn * tstride, lim_m, tid);
                aring[d] = alpha[(size_t)tn * sstride + i];
            }
            lds_barrier();
            if (t >= 0 && t + 1 < T) store_qrow(t + 1);
            if (t >= 0) {                                        // block-uniform
                float *qs = sQ + (t & 1) * cpad;
                const float *b1 = sA + ((t + 1) & 1) * S;
                const float *m1 = sX + ((t + 1) & 1) * S;
                float u[EPER], y[EPER], mjv[EPER];
                float mx = -__builtin_inff(), mm = -__builtin_inff();
#pragma unroll
                for (int r = 0; r < EPER; ++r) {
                    const bool val = r < kcnt;
                    float mv = m[midx[r]];
                    if (!HB && r == 0) mv = stay0 ? blank : mv;
                    const float bj = b1[dstj[r]];
                    mjv[r] = m1[dstj[r]];
                    u[r] = ((a0 + mv) + bj) - logZ;              // log-posterior of the edge
                    y[r] = val ? mv + bj : -__builtin_inff();    // beta recursion term
                    mx = maxf(mx, y[r]);
                }
                if (LPS == 2) mx = pair_max(mx);
                // one two-wide exp per edge: (posterior, logsumexp term)
                float P[EPER], ex[EPER];
#pragma unroll
                for (int r = 0; r < EPER; ++r) {
                    const f32x2 v = xb_expf2((f32x2){u[r], y[r] - mx});
                    P[r] = v.x;
                    ex[r] = v.y;
                }
                float sm;
                if constexpr (LPS == 1) {
                    sm = ex[0];
#pragma unroll
                    for (int r = 1; r < EPER; ++r) sm += ex[r];
                } else {
                    sm = pair_sum_ordered<E>(ex);
                }
                if (post_mode) {                                 // uniform, xb_crf_scans: the row carries P itself (stored here,
#pragma unroll                                                   // so that P is dead before the logs in the decode proper)
                    for (int r = 0; r < EPER; ++r)
                        if (r < kcnt && act) qs[qidx[r]] = P[r];
                }
                // the EPER logs of Q = log(P + 1e-8) and the log of the logsumexp, two at a time
                float la[EPER + 1], lo[EPER + 1];
#pragma unroll
                for (int r = 0; r < EPER; ++r) la[r] = P[r] + 1e-8f;
                la[EPER] = sm;
                xb_log_n<EPER + 1>(la, lo);
                const float bv = mx + lo[EPER];
#pragma unroll
                for (int r = 0; r < EPER; ++r) {
                    if (r < kcnt) {
                        if (act && !post_mode) qs[qidx[r]] = lo[r];
                        mm = maxf(mm, lo[r] + mjv[r]);
                    }
                }
                if (LPS == 2) mm = pair_max(mm);
                if (ph == 0 && act) {
                    sA[(t & 1) * S + wi] = bv;
                    sX[(t & 1) * S + wi] = mm;
                    bmax[(size_t)t * sstride + i] = mm;
                    if (beta_out) beta_out[(size_t)t * sstride + i] = bv;
                }
            }
          }
        }
        lds_barrier();
        store_qrow(0);
        __syncthreads();
    }

    if (SCAN && p.stop_after == 2) return;      // scans only (uniform)
#ifdef XB_LSTM_STAMPS
    if (p.debug_stop == 2) return;   // diagnostic build only: sweeps 1+2
#endif
    // --------------------------- sweep 3: Max forward over Q + per-step arg-max of the max-marginals
    {
        const int j = dj;
        constexpr int SH = EPER - (E - H);                        // register shift of the second half (0 or 1)
        const int col0 = ph == 0 ? j * E : j * E + E - EPER;      // this lane's first Q column (never past the row)
        const int k0 = LPS == 1 ? 0 : ph * H;
        if (tid < S) sX[tid] = 0.0f;
        float qring[RDEPTH13][EPER];
        float mring[RDEPTH13];
#pragma unroll
        for (int d = 0; d < RDEPTH13; ++d) {
            const int t = d < T ? d : T - 1;
            load_seg<EPER>(qrow + (size_t)t * qstride + col0, qring[d]);
            mring[d] = bmax[(size_t)(t + 1) * sstride + j];
        }
        // finalise the labels of steps [tb, tb + 64): lane l of the calling wave takes step tb + l
        auto finalise = [&](int tb) {
            const int t = tb + lane;
            if (t < T) {
                const float *rv = sRv + (t & (LRING - 1)) * NW;
                const int *ri = sRi + (t & (LRING - 1)) * NW;
                float bv = rv[0];
                int bi = ri[0];
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    const float v = rv[w];
                    const int c = ri[w];
                    if (v > bv || (v == bv && c < bi)) { bv = v; bi = c; }
                }
                sLab[t] = (int8_t)(bi % E);
            }
        };
        float aown = 0.0f;
        static_assert(RDEPTH13 % 4 == 0 && RDEPTH13 % RDEPTH == 0, "the arg-max reductions are batched four steps at a time");
        for (int t0 = 0; t0 < Tpad; t0 += RDEPTH13) {
          float bestv[RDEPTH13];
          int bestc[RDEPTH13];
#pragma unroll
          for (int d = 0; d < RDEPTH13; ++d) {
            const int t = t0 + d;                                // >= T in the padding iterations
            float qv[EPER];
#pragma unroll
            for (int r = 0; r < EPER; ++r) qv[r] = qring[d][r];
            const float m1j = mring[d];
            {
                const int tn = t + RDEPTH13 < T ? t + RDEPTH13 : T - 1;
                load_seg<EPER>(qrow + (size_t)tn * qstride + col0, qring[d]);
                mring[d] = bmax[(size_t)(tn + 1) * sstride + j];
            }
            lds_barrier();
            // every 64 steps one wave (taking turns) finalises the 64 steps before this one: their partials were all
            // written before this barrier, and the ring slots being rewritten meanwhile are at least 56 steps away
            if ((t & 63) == 0 && t > 0 && t <= T && wave == ((t >> 6) % NW)) finalise(t - 64);
            bestv[d] = -__builtin_inff();
            bestc[d] = 0x7fffffff;
            if (t < T) {                                         // block-uniform
                const float *am = sX + (t & 1) * S;
                float mm = -__builtin_inff();
                float best = -__builtin_inff();
                int bc = 0x7fffffff;
#pragma unroll
                for (int r = 0; r < EPER; ++r) {
                    const float Q = (LPS == 1 || ph == 0) ? qv[r] : qv[r + SH < EPER ? r + SH : EPER - 1];
                    float av;
                    if (r == 0) av = LPS == 1 ? aown : am[r0slot];
                    else av = am[sbase + r];
                    if (r < kcnt) {
                        mm = maxf(mm, Q + av);
                        const float scv = (av + Q) + m1j;
                        if (scv > best) { best = scv; bc = j * E + k0 + r; }   // increasing flat index
                    }
                }
                if (LPS == 2) mm = pair_max(mm);
                aown = mm;
                if (ph == 0 && act) sX[((t + 1) & 1) * S + tpos] = mm;
                bestv[d] = act ? best : -__builtin_inff();
                bestc[d] = bc;
            }
            // Arg-max over the workgroup's edges, four steps at a time (the four reduction chains interleave; done per
            // step, the serial chain would be the longest part of the step).  Lanes are in flat-index order (state major,
            // the two halves of a state on adjacent lanes) and each lane holds its lowest-index maximiser, so the lowest
            // lane that attains the wave maximum holds the wave's lowest flat index.
            if ((d & 3) == 3) {
                float w0 = bestv[d - 3], w1 = bestv[d - 2], w2 = bestv[d - 1], w3 = bestv[d];
                wave_max63_x4(w0, w1, w2, w3);
                const float wv[4] = {w0, w1, w2, w3};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int tq = t - 3 + q;
                    const float wmx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, wv[q]), 63));
                    const unsigned long long hit = __ballot(bestv[d - 3 + q] == wmx);
                    const int wl = __builtin_amdgcn_readfirstlane(hit ? (int)__builtin_ctzll(hit) : 0);
                    const int wc = __builtin_amdgcn_readlane(bestc[d - 3 + q], wl);
                    if (lane == 0 && tq < T) {
                        sRv[(tq & (LRING - 1)) * NW + wave] = wmx;
                        sRi[(tq & (LRING - 1)) * NW + wave] = wc;
                    }
                }
            }
          }
        }
        __syncthreads();
        // the loop finalised steps [0, done): its iterations t = 64, 128, .. <= min(T, Tpad - 1)
        {
            const int last = T < Tpad - 1 ? T : Tpad - 1;
            const int done = (last / 64) * 64;
            if (wave == 0) for (int tb = done; tb < T; tb += 64) finalise(tb);
        }
        __syncthreads();
    }

    // ------------------------------------------------ labels out + path_to_str + left-pack
    if (p.labels)
        for (int t = tid; t < T; t += BS) p.labels[(size_t)n * T + t] = sLab[t];
    if (p.seq || p.seq_len) {
        int *sCnt = reinterpret_cast<int *>(sM);      // BS+1 ints; the row staging is free now
        const int per = (T + BS - 1) / BS;
        const int lo = tid * per, hiT = (lo + per < T) ? lo + per : T;
        int cnt = 0;
        for (int t = lo; t < hiT; ++t) cnt += sLab[t] != 0;
        sCnt[tid + 1] = cnt;
        __syncthreads();
        if (tid == 0) {
            sCnt[0] = 0;
            for (int w = 1; w <= BS; ++w) sCnt[w] += sCnt[w - 1];
        }
        __syncthreads();
        const int total = sCnt[BS];
        if (p.seq) {
            int8_t *out = p.seq + (size_t)n * T;
            int pos = sCnt[tid];
            for (int t = lo; t < hiT; ++t) {
                const int l = sLab[t];
                if (l != 0) out[pos++] = (int8_t)p.alphabet[l];
            }
            for (int t = total + tid; t < T; t += BS) out[t] = 0;
        }
        if (p.seq_len && tid == 0) p.seq_len[n] = total;
    }
}

template <int NB, int BS, int LPS, bool SCAN>
hipError_t launch_nb_bs(const xb::DecodeParams &p, int vw, hipStream_t stream)
{
    // must mirror the kernel's LDS carve
    constexpr int E = NB + 1;
    auto lds_bytes = [&](int w) {
        const int nrs = w == 4 ? (E + 4 * LPS - 1) / (4 * LPS) : (E + LPS - 1) / LPS;      // the kernel's NRS
        const size_t cpad = (size_t)BS * nrs * w;
        const size_t b = sizeof(float) * (4 * cpad + 5 * (size_t)p.S + (size_t)LRING * (BS / 64)) +
                         sizeof(int) * (size_t)LRING * (BS / 64) + sizeof(float) * 4 + (size_t)p.T;
        return (b + 15) & ~(size_t)15;
    };
    // the four-wide row loads pad a state's edges to a multiple of four: where that no longer fits (4^5 states), one by one
    if (vw == 4 && lds_bytes(4) > 160 * 1024) vw = 1;
    const size_t lds = lds_bytes(vw);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid(p.N), block(BS);
#define XB_LAUNCH(HB, VW)                                                                                          \
    do {                                                                                                           \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&crf_decode_kernel<NB, BS, HB, LPS, VW, SCAN>),         \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        hipLaunchKernelGGL((crf_decode_kernel<NB, BS, HB, LPS, VW, SCAN>), grid, block, lds, stream, p);                 \
    } while (0)
    if (p.has_blank) {
        if (vw == 4) XB_LAUNCH(true, 4); else XB_LAUNCH(true, 1);
    } else {
        if (vw == 4) XB_LAUNCH(false, 4); else XB_LAUNCH(false, 1);
    }
#undef XB_LAUNCH
    return hipGetLastError();
}

// Block = smallest multiple of 64 threads (from a short list, pruned per alphabet) that holds LPS lanes for each state.
template <int NB, int LPS, bool SCAN>
hipError_t launch_nb_lps(const xb::DecodeParams &p, int vw, hipStream_t stream)
{
    const int need = LPS * p.S;
    if (need <= 64) return launch_nb_bs<NB, 64, LPS, SCAN>(p, vw, stream);
    if (need <= 128) return launch_nb_bs<NB, 128, LPS, SCAN>(p, vw, stream);
    if (need <= 256) return launch_nb_bs<NB, 256, LPS, SCAN>(p, vw, stream);
    if constexpr (NB == 6) {
        if (need <= 448) return launch_nb_bs<NB, 448, LPS, SCAN>(p, vw, stream);     // 2 x 216 states
    } else {
        if (need <= 640) return launch_nb_bs<NB, 640, LPS, SCAN>(p, vw, stream);     // 5^4 states / 2 x 4^4
        if constexpr (NB == 4 && LPS == 1) {
            if (need <= 1024) return launch_nb_bs<NB, 1024, LPS, SCAN>(p, vw, stream);   // 4^5 states
        }
    }
    return hipErrorInvalidValue;
}
template <int NB>
hipError_t launch_nb(const xb::DecodeParams &p, int vw, hipStream_t stream)
{
    // the scan variant exists with one lane per state only (the results do not depend on the lane split)
    if (p.stop_after || p.beta_out || p.post_mode) return launch_nb_lps<NB, 1, true>(p, vw, stream);
    if (xb::decode_lanes_per_state(p.S, p.N) == 2) return launch_nb_lps<NB, 2, false>(p, vw, stream);
    return launch_nb_lps<NB, 1, false>(p, vw, stream);
}


// ======================================================================================================================
// CTC-CRF loss scans: seqdist.ctc_simple's logZ over the stay / move lattice of a target sequence
// (CTC_CRF.ctc_loss / ctc_viterbi_alignments, ub-bonito/bonito/crf/model.py:102-135; xb_internal.h CtcParams).
// One workgroup per chunk, CTC_BS threads, position l of the target handled by thread l % CTC_BS (CTC_PMAX positions per
// thread at most); the position vector lives in LDS (double buffered, one barrier per time step); the two gathered
// scores a position needs per step come from the score row through a CTC_RD-deep register ring so that their latency is
// off the step's dependency chain.  Arithmetic: the decode's contract (xb_expf / xb_logf, sum2 = max, exp, exp, add, log in
// the order stay term, move term), written independently of oracle/xna_oracle.c: xo_ctc_logz and bit-equal to it.
// ======================================================================================================================
constexpr int CTC_BS = 256, CTC_PMAX = 8, CTC_RD = 4;
constexpr float CTC_ZERO = -1e38f;               // seqdist's Log.zero / Max.zero

template <bool MAXS>
__device__ __forceinline__ float ctc_sum2(float x0, float x1)
{
    const float m = x0 > x1 ? x0 : x1;
    if (MAXS) return m;
    return m + xb_logf(xb_expf(x0 - m) + xb_expf(x1 - m));
}

template <bool MAXS>
__global__ __launch_bounds__(CTC_BS) void ctc_scan_kernel(xb::CtcParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char ctc_smem[];
    float *vec = reinterpret_cast<float *>(ctc_smem);            // [2][n + 2]: slot l + 1 = position l, slots 0 and n + 1 = `zero`
    const int tid = threadIdx.x, b = blockIdx.x, n = p.n, T = p.T;
    const int len = p.tlen[b] + 1 - p.sl;
    if (len < 1 || len > n) {
        if (tid == 0) atomicOr(p.error, 2u);
        return;
    }
    const int32_t *si = p.stay_idx + (size_t)b * n, *mi = p.move_idx + (size_t)b * (n - 1);
    const float *srow = p.scores + (size_t)b * p.C;
    const size_t tstride = (size_t)p.N * p.C;
    float *stash = p.alpha ? p.alpha + (size_t)b * (T + 1) * n : nullptr;
    const int vs = n + 2;
    // this thread's positions and their gather columns: stay (own), move into the position (l - 1 -> l), move out of it (l -> l + 1)
    int cs[CTC_PMAX], cin[CTC_PMAX], cout[CTC_PMAX];
#pragma unroll
    for (int k = 0; k < CTC_PMAX; ++k) {
        const int l = tid + k * CTC_BS;
        cs[k] = l < n ? si[l] : 0;
        cin[k] = (l > 0 && l < n) ? mi[l - 1] : 0;
        cout[k] = l + 1 < n ? mi[l] : 0;
    }
    for (int i = tid; i < 2 * vs; i += CTC_BS) vec[i] = CTC_ZERO;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < CTC_PMAX; ++k) {
        const int l = tid + k * CTC_BS;
        if (l < n) {
            const float a0 = l == 0 ? 0.0f : CTC_ZERO;
            vec[l + 1] = a0;
            if (stash) stash[l] = a0;
        }
    }
    __syncthreads();

    // ---- forward: alpha_{t+1}[l] = sum2(alpha_t[l] + stay[t][l], alpha_t[l-1] + move[t][l-1])
    float rs[CTC_RD][CTC_PMAX], rm[CTC_RD][CTC_PMAX];
    auto fetch = [&](int t, float (&s_)[CTC_PMAX], float (&m_)[CTC_PMAX], const int (&cm)[CTC_PMAX]) {
        const float *row = srow + (size_t)(t < 0 ? 0 : (t >= T ? T - 1 : t)) * tstride;
#pragma unroll
        for (int k = 0; k < CTC_PMAX; ++k)
            if (tid + k * CTC_BS < n) {
                s_[k] = row[cs[k]];
                m_[k] = row[cm[k]];
            }
    };
#pragma unroll
    for (int d = 0; d < CTC_RD; ++d) fetch(d, rs[d], rm[d], cin);
    int cur = 0;
    for (int t0 = 0; t0 < T; t0 += CTC_RD) {
#pragma unroll
        for (int d = 0; d < CTC_RD; ++d) {
            const int t = t0 + d;
            if (t < T) {                                   // uniform
                const float *vc = vec + cur * vs;
                float *vn = vec + (cur ^ 1) * vs;
#pragma unroll
                for (int k = 0; k < CTC_PMAX; ++k) {
                    const int l = tid + k * CTC_BS;
                    if (l < n) {
                        const float x0 = vc[l + 1] + rs[d][k];
                        const float x1 = l > 0 ? vc[l] + rm[d][k] : CTC_ZERO;
                        const float v = ctc_sum2<MAXS>(x0, x1);
                        vn[l + 1] = v;
                        if (stash) stash[(size_t)(t + 1) * n + l] = v;
                    }
                }
                fetch(t + CTC_RD, rs[d], rm[d], cin);
                __syncthreads();
                cur ^= 1;
            }
        }
    }
    const float lz = vec[cur * vs + len];                  // alpha_T[len - 1]
    if (tid == 0 && p.logz) p.logz[b] = lz;
    if (!stash) return;
    __syncthreads();

    if (!MAXS) {
        // ---- backward with the restricted posteriors: beta_t[l] = sum2(stay[t][l] + beta_{t+1}[l], move[t][l] + beta_{t+1}[l+1])
        for (int i = tid; i < 2 * vs; i += CTC_BS) vec[i] = CTC_ZERO;
        __syncthreads();
        if (tid == 0) vec[len] = 0.0f;                     // beta_T: `one` at position len - 1
        __syncthreads();
        cur = 0;
#pragma unroll
        for (int d = 0; d < CTC_RD; ++d) fetch(T - 1 - d, rs[d], rm[d], cout);
        for (int t0 = T - 1; t0 >= 0; t0 -= CTC_RD) {
#pragma unroll
            for (int d = 0; d < CTC_RD; ++d) {
                const int t = t0 - d;
                if (t >= 0) {
                    const float *vc = vec + cur * vs;
                    float *vn = vec + (cur ^ 1) * vs;
#pragma unroll
                    for (int k = 0; k < CTC_PMAX; ++k) {
                        const int l = tid + k * CTC_BS;
                        if (l < n) {
                            const float a = stash[(size_t)t * n + l];
                            const float st = rs[d][k], mv = rm[d][k];
                            const float bn = vc[l + 1];
                            const float bx = l + 1 < n ? vc[l + 2] : CTC_ZERO;
                            if (p.gstay) p.gstay[((size_t)t * p.N + b) * n + l] = xb_expf(((a + st) + bn) - lz);
                            if (p.gmove && l + 1 < n) p.gmove[((size_t)t * p.N + b) * (n - 1) + l] = xb_expf(((a + mv) + bx) - lz);
                            const float x0 = st + bn;
                            const float x1 = l + 1 < n ? mv + bx : CTC_ZERO;
                            vn[l + 1] = ctc_sum2<false>(x0, x1);
                        }
                    }
                    fetch(t - CTC_RD, rs[d], rm[d], cout);
                    __syncthreads();
                    cur ^= 1;
                }
            }
        }
    } else if (p.gstay) {
        // ---- back-trace of the max path (one lane; T dependent steps): the alignment rows were zeroed by the host
        __threadfence_block();
        if (tid == 0) {
            int l = len - 1;
            for (int t = T - 1; t >= 0; --t) {
                const float *row = srow + (size_t)t * tstride;
                const float *a = stash + (size_t)t * n;
                const float x0 = a[l] + row[si[l]];
                const float x1 = l > 0 ? a[l - 1] + row[mi[l - 1]] : CTC_ZERO;
                if (!(x0 >= x1)) l -= 1;                    // the move edge was strictly better (ties prefer the stay edge)
                p.gstay[((size_t)t * p.N + b) * n + l] = 1.0f;
            }
        }
    }
}

}  // namespace

namespace xb {

int decode_lanes_per_state(int S, int N)
{
    if (const char *e = getenv("XB_DECODE_LPS")) {
        const int v = atoi(e);
        if ((v == 1 || v == 2) && v * S <= 1024) return v;
    }
    // Two lanes per state halve the per-lane chain of exps at the price of a workgroup twice as wide (the pair's per-state
    // work is done by both lanes).  That pays while the batch leaves a CU only ~2 workgroups; with more chunks per CU the
    // narrower workgroup wins.  Measured on MI355X, T = 2000 (ms, 2 lanes vs 1 lane): S = 125: N = 512 3.78 vs 4.77,
    // N = 1024 7.16 vs 6.22, N = 2048 12.99 vs 12.20; S = 216: N = 512 8.72 vs 6.28, N = 1024 17.41 vs 12.22.
    return (2 * S <= 256 && N <= 768) ? 2 : 1;
}

// Host-side launch.  Shapes are validated here so the kernel's indexing assumptions hold:
//   S = NB^state_len <= 1024, cin = S*(NB+1) or S*NB, ld >= cin (the pack scratch of BS+1 ints always
//   fits the staging area).
hipError_t launch_crf_decode(const DecodeParams &p, hipStream_t stream)
{
    if (p.S < 1 || p.S > 1024 || p.T < 1 || p.N < 1) return hipErrorInvalidValue;
    if (p.S % p.nb != 0 || p.hi * p.nb != p.S) return hipErrorInvalidValue;
    const int E = p.nb + 1;
    if (p.cin != (p.has_blank ? p.S * E : p.S * p.nb) || p.ld < p.cin) return hipErrorInvalidValue;
    if (!p.qbuf || p.ldq % 4 != 0 || p.ldq < p.S * E || reinterpret_cast<uintptr_t>(p.qbuf) % 16 != 0) return hipErrorInvalidValue;
    int vw = 1;
    const uintptr_t a = reinterpret_cast<uintptr_t>(p.scores);
    // vector loads may run into the row's padding columns (ld >= cin rounded up), never past the row
    if (p.ld % 4 == 0 && p.ld >= ((p.cin + 3) & ~3) && a % 16 == 0) vw = 4;
    switch (p.nb) {
    case 4: return launch_nb<4>(p, vw, stream);
    case 5: return launch_nb<5>(p, vw, stream);
    case 6: return launch_nb<6>(p, vw, stream);
    default: return hipErrorInvalidValue;
    }
}


hipError_t launch_ctc_scan(const CtcParams &p, hipStream_t stream)
{
    if (p.T < 1 || p.N < 1 || p.n < 1 || p.n > CTC_BS * CTC_PMAX || p.C < 1 || !p.scores || !p.stay_idx || !p.move_idx || !p.tlen || !p.error)
        return hipErrorInvalidValue;
    if ((p.gstay || p.gmove) && !p.alpha) return hipErrorInvalidValue;
    const size_t lds = sizeof(float) * 2 * (size_t)(p.n + 2);
    if (p.semiring) hipLaunchKernelGGL(ctc_scan_kernel<true>, dim3(p.N), dim3(CTC_BS), lds, stream, p);
    else hipLaunchKernelGGL(ctc_scan_kernel<false>, dim3(p.N), dim3(CTC_BS), lds, stream, p);
    return hipGetLastError();
}
int ctc_max_positions() { return CTC_BS * CTC_PMAX; }

}  // namespace xb

// xb_decode.hip -- CRF posterior + max-plus decode for gfx950 (MI355X).
//
// Replaces SeqdistModel.decode_batch (ub-bonito/bonito/crf/model.py:215-218) =
//   seqdist posteriors (Log semiring fwd/bwd scans, crf/model.py:41-46)
//   -> +1e-8 -> log -> seqdist posteriors with the Max semiring -> argmax % (n_base+1)
//   (crf/model.py:92-95), then path_to_str (crf/model.py:97-100) and the left-pack of
//   compute_scores (crf/basecall.py:60-67).
//
// One workgroup per chunk, one thread per CRF state; the state vectors (alpha, beta, max-plus
// alpha/beta) live in LDS and are exchanged once per time step behind a single barrier.
// Scores stream from HBM through a register ring (D steps ahead) in 4/8/16-byte coalesced
// loads, are staged in LDS and consumed by state.  Three sweeps over the scores:
//   1. Log forward           (write alpha)
//   2. Log backward fused with Max backward (read alpha; write beta, bmax) -- also writes the
//      log-posteriors Q = log(P + 1e-8) of every edge, (T, N, S*E) fp32, staged by destination
//      edge in LDS and stored as coalesced rows one step late
//   3. Max forward over Q + per-step arg-max of the max-marginals (read Q, bmax), then pack.
// (Storing Q trades C*4 extra bytes per step for not recomputing 2*E transcendentals per state
//  in sweep 3, which then is as light as sweep 1; the values are the same floats either way.)
//
// Floating-point contract (shared by specification with oracle/xna_oracle.c, written
// independently): IEEE binary32, no contraction (this file is built with -ffp-contract=off),
// fma only where __builtin_fmaf is written, exp/log are the fixed polynomials below, logsumexp
// is max / ordered sum of exp / log, ties resolve to the lowest flat edge index.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "xb_internal.h"

namespace {

__device__ __forceinline__ float bits2f(uint32_t u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ uint32_t f2bits(float f) { return __builtin_bit_cast(uint32_t, f); }

__device__ __forceinline__ float xb_expf(float x)
{
    const bool tiny = x < -87.0f;
    x = x > 88.0f ? 88.0f : x;
    x = tiny ? 0.0f : x;
    const float n = __builtin_rintf(x * 1.44269504088896341f);
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
    const int ni = (int)n;
    const float s = bits2f((uint32_t)(ni + 127) << 23);
    const float v = y * s;
    return tiny ? 0.0f : v;
}

__device__ __forceinline__ float xb_logf(float x)
{
    const uint32_t ix = f2bits(x);
    int e = (int)(ix >> 23) - 127;
    float m = bits2f((ix & 0x007fffffu) | 0x3f800000u);
    const bool big = m > 1.41421356237309505f;
    m = big ? m * 0.5f : m;
    e = big ? e + 1 : e;
    const float f = m - 1.0f;
    const float z = f * f;
    float p = 7.0376836292e-2f;
    p = __builtin_fmaf(p, f, -1.1514610310e-1f);
    p = __builtin_fmaf(p, f, 1.1676998740e-1f);
    p = __builtin_fmaf(p, f, -1.2420140846e-1f);
    p = __builtin_fmaf(p, f, 1.4249322787e-1f);
    p = __builtin_fmaf(p, f, -1.6668057665e-1f);
    p = __builtin_fmaf(p, f, 2.0000714765e-1f);
    p = __builtin_fmaf(p, f, -2.4999993993e-1f);
    p = __builtin_fmaf(p, f, 3.3333331174e-1f);
    float y = (f * z) * p;
    const float fe = (float)e;
    y = __builtin_fmaf(fe, -2.12194440e-4f, y);
    y = __builtin_fmaf(-0.5f, z, y);
    float r = f + y;
    r = __builtin_fmaf(fe, 0.693359375f, r);
    return r;
}

__device__ __forceinline__ float maxf(float a, float b) { return b > a ? b : a; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also fences global memory, i.e. emits
// s_waitcnt vmcnt(0), which would drain the score prefetch ring (a full HBM latency) at every time step.
// Inside the sweeps the only cross-thread data is in LDS; the global stashes are re-read by the thread that
// wrote them (and the sweeps are separated by real __syncthreads()).
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int VW> struct VecT;
template <> struct VecT<1> { using type = float; };
// native clang vectors (HIP's float2/float4 wrapper structs defeat scalar replacement in arrays)
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <> struct VecT<2> { using type = f32x2; };
template <> struct VecT<4> { using type = f32x4; };

// ---- prefetch: register ring with PLAIN (compiler-visible) loads -------------------------------------------
// Score / Q rows of the next RDEPTH steps are held in registers and staged through LDS once per step.  The loads
// are ordinary loads, fully tracked by hipcc: nothing can be copied or reused while in flight.  (An inline-asm
// load ring with hand-counted s_waitcnt was faster by ~8 % but NOT safe: hipcc may re-allocate or copy an asm
// load's destination before the data arrives, and an issued-but-unconsumed load corrupts whatever reuses its
// register; an LDS-DMA ring is safe but ~25 % slower, each DMA instruction stalls its wave for 60-185 cycles.)
// The memory part of every sweep is branch-free -- loops run over T rounded up to RDEPTH with clamped
// addresses, LDS staging is unconditional (padded buffers), only the arithmetic is guarded -- which is what
// lets the compiler's own s_waitcnt insertion count the in-order loads (vmcnt(9..12)) instead of draining them.
template <int VW, int NR, int BS>
struct RowRegs {
    using V = typename VecT<VW>::type;
    V r[NR];
    __device__ __forceinline__ void load(const float *row, int lim, int tid)
    {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int g = (tid + BS * i) * VW;
            r[i] = *reinterpret_cast<const V *>(row + (g < lim ? g : 0));
        }
    }
    __device__ __forceinline__ void store(float *lds, int tid) const
    {
#pragma unroll
        for (int i = 0; i < NR; ++i) *reinterpret_cast<V *>(lds + (tid + BS * i) * VW) = r[i];
    }
};
constexpr int RDEPTH = 4;  // register-ring depth (steps in flight)


// wave64 arg-max of (value, flat index), ties to the lowest index, on DPP row shifts / row broadcasts
// (no LDS round trips).  The result is valid in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ void argmax_dpp_step(float &v, int &c)
{
    // lanes without a source keep `old` = the identity (-inf, INT_MAX)
    const float ov = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
        (int)0xff800000, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
    const int oc = __builtin_amdgcn_update_dpp(0x7fffffff, c, CTRL, ROW_MASK, 0xf, false);
    if (ov > v || (ov == v && oc < c)) { v = ov; c = oc; }
}
// ---- lane clusters: a state is served by LPS = 1, 2 or 4 adjacent lanes (inside one DPP quad) --------------
template <int CTRL> __device__ __forceinline__ float quad_perm(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
// max over the LPS lanes of a cluster (max is exact: any order)
template <int LPS> __device__ __forceinline__ float cluster_max(float v)
{
    if (LPS >= 2) { const float o = quad_perm<0xB1>(v); v = o > v ? o : v; }    // quad_perm [1,0,3,2]
    if (LPS >= 4) { const float o = quad_perm<0x4E>(v); v = o > v ? o : v; }    // quad_perm [2,3,0,1]
    return v;
}
// value held by lane P of this lane's cluster
template <int LPS, int P> __device__ __forceinline__ float cluster_get(float v)
{
    if (LPS == 1) return v;
    if (LPS == 2) return quad_perm<(P) | (P << 2) | ((2 + P) << 4) | ((2 + P) << 6)>(v);   // pairs {0,1} {2,3}
    return quad_perm<(P & 3) | ((P & 3) << 2) | ((P & 3) << 4) | ((P & 3) << 6)>(v);
}
// Ordered sum over the E terms of a cluster: lane p holds terms e = p*EPER + r in x[r]; every lane of the cluster
// accumulates all terms in edge order e = 0..E-1 (the contract's summation order), fetching them by DPP.
template <int LPS, int E, int EPER, int PP = 0, int R = 0>
__device__ __forceinline__ void ordered_sum(const float (&x)[EPER], float &s)
{
    if constexpr (PP < LPS) {
        if constexpr (PP * EPER + R < E) {
            const float v = cluster_get<LPS, PP>(x[R]);
            s = (PP == 0 && R == 0) ? v : s + v;
        }
        if constexpr (R + 1 < EPER) ordered_sum<LPS, E, EPER, PP, R + 1>(x, s);
        else ordered_sum<LPS, E, EPER, PP + 1, 0>(x, s);
    }
}
__device__ __forceinline__ void wave_argmax(float &v, int &c)
{
    argmax_dpp_step<0x111, 0xf>(v, c);   // row_shr:1
    argmax_dpp_step<0x112, 0xf>(v, c);   // row_shr:2
    argmax_dpp_step<0x114, 0xf>(v, c);   // row_shr:4
    argmax_dpp_step<0x118, 0xf>(v, c);   // row_shr:8   -> lane 15 of every row holds the row result
    argmax_dpp_step<0x142, 0xa>(v, c);   // row_bcast:15 into rows 1 and 3
    argmax_dpp_step<0x143, 0xc>(v, c);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave result
}

// M[j][k] from the staged row: with the blank column present it is lds[j*E+k]; otherwise column 0
// is the constant blank and column k>=1 is lds[j*NB + k-1].
template <int NB, bool HB>
__device__ __forceinline__ float score_at(const float *lds, int j, int k, float blank)
{
    constexpr int E = NB + 1;
    if (HB) return lds[j * E + k];
    return k == 0 ? blank : lds[j * NB + k - 1];
}

// LPS = lanes per state (1, 2 or 4): the E edges of a state are split in blocks of EPER over LPS adjacent lanes;
// the block has BS >= LPS*S threads.
template <int NB, int BS, int VW, bool HB, int LPS>
__global__ __launch_bounds__(BS) void crf_decode_kernel(xb::DecodeParams p)
{
    constexpr int E = NB + 1;
    constexpr int NW = BS / 64;
    constexpr int EPER = (E + LPS - 1) / LPS;    // edges per lane
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int S = p.S, hi = p.hi, T = p.T, N = p.N, cin = p.cin, ldq = p.ldq;
    const int tid = threadIdx.x, lane = tid & 63;
    const int n = blockIdx.x;
    const int st = tid / LPS, ph = tid % LPS;    // state and position inside its lane cluster
    const bool act = st < S;
    const int stc = act ? st : S - 1;            // clamped state (always a valid column)

    constexpr int NR = (E + LPS * VW - 1) / (LPS * VW);   // BS*NR*VW >= LPS*S*NR*VW >= S*E >= cin
    constexpr int NRQ = (E + 4 * LPS - 1) / (4 * LPS);   // 16-byte groups per thread of a Q row
    constexpr int cpad = BS * 8;                          // staging row >= BS*NR*VW and >= BS*NRQ*4 (E <= 8)
    const int lim_m = VW == 4 ? (cin + 3) & ~3 : cin;     // vector loads may touch the row's padding columns
    const int Tpad = (T + RDEPTH - 1) / RDEPTH * RDEPTH;

    float *sM = reinterpret_cast<float *>(smem_raw);             // [2][cpad]  staged score / Q row
    float *sQ = sM + 2 * cpad;                                    // [2][cpad]  Q rows being assembled (sweep 2)
    float *sA = sQ + 2 * cpad;                                    // [2][S]  alpha / beta
    float *sX = sA + 2 * S;                                       // [2][S]  max-plus alpha / beta
    float *sG = sX + 2 * S;                                       // [S]     scratch (logZ)
    float *sRv = sG + S;                                          // [2][NW] arg-max partials
    int *sRi = reinterpret_cast<int *>(sRv + 2 * NW);             // [2][NW]
    float *sBc = reinterpret_cast<float *>(sRi + 2 * NW);         // [4] broadcast scratch
    int8_t *sLab = reinterpret_cast<int8_t *>(sBc + 4);           // [T]

    const float *sc = p.scores + (size_t)n * p.ld;
    const size_t tstride = (size_t)N * p.ld;
    float *alpha = p.alpha + (size_t)n * S;
    float *beta = p.beta + (size_t)n * S;
    float *bmax = p.bmax + (size_t)n * S;
    const size_t sstride = (size_t)N * S;
    float *qrow = p.qbuf + (size_t)n * ldq;
    const size_t qstride = (size_t)N * ldq;
    const float blank = p.blank;

    RowRegs<VW, NR, BS> ring[RDEPTH];

    // ------------------------------------------------------------------ sweep 1: Log forward
    if (tid < S) { sA[tid] = 0.0f; alpha[tid] = 0.0f; }
#pragma unroll
    for (int d = 0; d < RDEPTH; ++d) ring[d].load(sc + (size_t)(d < T ? d : T - 1) * tstride, lim_m, tid);
    for (int t0 = 0; t0 < Tpad; t0 += RDEPTH) {
#pragma unroll
      for (int d = 0; d < RDEPTH; ++d) {
        const int t = t0 + d;                                // >= T in the padding iterations
        float *m = sM + (t & 1) * cpad;
        ring[d].store(m, tid);
        ring[d].load(sc + (size_t)(t + RDEPTH < T ? t + RDEPTH : T - 1) * tstride, lim_m, tid);
        lds_barrier();
        if (act && t < T) {
            const float *a0 = sA + (t & 1) * S;
            const int j = stc;
            const int jq = j / NB;
            float x[EPER];
            float mx = -__builtin_inff();
#pragma unroll
            for (int r = 0; r < EPER; ++r) {
                const int k = ph * EPER + r;                 // in-edge of state j (0 = stay)
                const bool val = k < E;
                const int kc = val ? k : 0;
                const int src = kc == 0 ? j : (kc - 1) * hi + jq;
                x[r] = val ? score_at<NB, HB>(m, j, kc, blank) + a0[src] : -__builtin_inff();
                mx = maxf(mx, x[r]);
            }
            mx = cluster_max<LPS>(mx);
            float ex[EPER];
#pragma unroll
            for (int r = 0; r < EPER; ++r) ex[r] = xb_expf(x[r] - mx);
            float s;
            ordered_sum<LPS, E, EPER>(ex, s);
            const float v = mx + xb_logf(s);
            if (ph == 0) {
                sA[((t + 1) & 1) * S + j] = v;
                alpha[(size_t)(t + 1) * sstride + j] = v;
            }
        }
      }
    }
    __syncthreads();

    // logZ = logsumexp_j alpha_T[j], summed in order j = 0..S-1
    {
        const float *aT = sA + (T & 1) * S;
        if (tid == 0) {
            float mx = aT[0];
            for (int j = 1; j < S; ++j) mx = maxf(mx, aT[j]);
            sBc[0] = mx;
        }
        __syncthreads();
        const float mx = sBc[0];
        if (tid < S) sG[tid] = xb_expf(aT[tid] - mx);
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
#ifdef XB_LSTM_STAMPS
    if (p.debug_stop == 1) return;   // diagnostic build only: time sweep 1 alone
#endif

    // -------------------------------------------- sweep 2: Log backward + Max backward (fused)
    {
        if (tid < S) {
            sA[(T & 1) * S + tid] = 0.0f;
            sX[(T & 1) * S + tid] = 0.0f;
            beta[(size_t)T * sstride + tid] = 0.0f;
            bmax[(size_t)T * sstride + tid] = 0.0f;
        }
        // cluster = source state i; lane ph owns the out-edges e = ph*EPER .. (0 = stay, e >= 1 = new base e-1)
        const int i = stc;
        const int kk = i / hi + 1;
        const int jb = (i % hi) * NB;
        // the Q row of step t is complete once every thread has passed the barrier of step t-1:
        // it is stored (coalesced 16-byte groups) during iteration t-1
        auto store_qrow = [&](int t) {
            const float *src = sQ + (t & 1) * cpad;
            float *dst = qrow + (size_t)t * qstride;
            for (int g = tid * 4; g < ldq; g += BS * 4)
                *reinterpret_cast<f32x4 *>(dst + g) = *reinterpret_cast<const f32x4 *>(src + g);
        };
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
            const int t = T - 1 - (s0 + d);                  // < 0 in the padding iterations
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
            if (act && t >= 0) {
                float *qs = sQ + (t & 1) * cpad;
                const float *b1 = sA + ((t + 1) & 1) * S;
                const float *m1 = sX + ((t + 1) & 1) * S;
                float y[EPER];
                float mx = -__builtin_inff(), mm = -__builtin_inff();
#pragma unroll
                for (int r = 0; r < EPER; ++r) {
                    const int e = ph * EPER + r;
                    const bool val = e < E;
                    const int ee = val ? e : 0;
                    const int j = ee == 0 ? i : jb + ee - 1;
                    const int k = ee == 0 ? 0 : kk;
                    const float mv = score_at<NB, HB>(m, j, k, blank);
                    const float bj = b1[j];
                    const float xx = ((a0 + mv) + bj) - logZ;
                    const float q = xb_logf(xb_expf(xx) + 1e-8f);
                    y[r] = val ? mv + bj : -__builtin_inff();
                    if (val) {
                        qs[j * E + k] = q;
                        mx = maxf(mx, y[r]);
                        mm = maxf(mm, q + m1[j]);
                    }
                }
                mx = cluster_max<LPS>(mx);
                mm = cluster_max<LPS>(mm);
                float ex[EPER];
#pragma unroll
                for (int r = 0; r < EPER; ++r) ex[r] = xb_expf(y[r] - mx);
                float sm;
                ordered_sum<LPS, E, EPER>(ex, sm);
                const float bv = mx + xb_logf(sm);
                if (ph == 0) {
                    sA[(t & 1) * S + i] = bv;
                    sX[(t & 1) * S + i] = mm;
                    beta[(size_t)t * sstride + i] = bv;
                    bmax[(size_t)t * sstride + i] = mm;
                }
            }
          }
        }
        lds_barrier();
        store_qrow(0);
        __syncthreads();
    }

#ifdef XB_LSTM_STAMPS
    if (p.debug_stop == 2) return;   // diagnostic build only: sweeps 1+2
#endif
    // --------------------------- sweep 3: Max forward over Q + per-step arg-max of the max-marginals
    {
        if (tid < S) sX[tid] = 0.0f;
        RowRegs<4, NRQ, BS> qring[RDEPTH];
        float mring[RDEPTH];
#pragma unroll
        for (int d = 0; d < RDEPTH; ++d) {
            const int t = d < T ? d : T - 1;
            qring[d].load(qrow + (size_t)t * qstride, ldq, tid);
            mring[d] = bmax[(size_t)(t + 1) * sstride + stc];
        }
        const int j = stc;
        const int jq = j / NB;
        const int wave = tid >> 6;
        for (int t0 = 0; t0 < Tpad; t0 += RDEPTH) {
#pragma unroll
          for (int d = 0; d < RDEPTH; ++d) {
            const int t = t0 + d;                            // >= T in the padding iterations
            float *m = sM + (t & 1) * cpad;
            qring[d].store(m, tid);
            const float m1j = mring[d];
            {
                const int tn = t + RDEPTH < T ? t + RDEPTH : T - 1;
                qring[d].load(qrow + (size_t)tn * qstride, ldq, tid);
                mring[d] = bmax[(size_t)(tn + 1) * sstride + stc];
            }
            lds_barrier();
            // finalise the previous step's arg-max (partials were written before this barrier)
            if (tid == 0 && t > 0 && t <= T) {
                const float *rv = sRv + ((t - 1) & 1) * NW;
                const int *ri = sRi + ((t - 1) & 1) * NW;
                float bv = rv[0];
                int bi = ri[0];
                for (int w = 1; w < NW; ++w)
                    if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
                sLab[t - 1] = (int8_t)(bi % E);
            }
            if (t < T) {                                     // wave-uniform
            float best = -__builtin_inff();
            int bestc = 0x7fffffff;
            if (act) {
                const float *am = sX + (t & 1) * S;
                float mm = -__builtin_inff();
#pragma unroll
                for (int r = 0; r < EPER; ++r) {
                    const int k = ph * EPER + r;             // in-edge of state j, increasing flat index
                    if (k < E) {
                        const int src = k == 0 ? j : (k - 1) * hi + jq;
                        const float Q = m[j * E + k];
                        const float av = am[src];
                        mm = maxf(mm, Q + av);
                        const float scv = (av + Q) + m1j;
                        if (scv > best) { best = scv; bestc = j * E + k; }
                    }
                }
                mm = cluster_max<LPS>(mm);
                if (ph == 0) sX[((t + 1) & 1) * S + j] = mm;
            }
            wave_argmax(best, bestc);
            if (lane == 63) {
                sRv[(t & 1) * NW + wave] = best;
                sRi[(t & 1) * NW + wave] = bestc;
            }
            }
          }
        }
        __syncthreads();
        if (tid == 0 && Tpad == T) {                             // otherwise a padding iteration already did it
            const float *rv = sRv + ((T - 1) & 1) * NW;
            const int *ri = sRi + ((T - 1) & 1) * NW;
            float bv = rv[0];
            int bi = ri[0];
            for (int w = 1; w < NW; ++w)
                if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
            sLab[T - 1] = (int8_t)(bi % E);
        }
        __syncthreads();
    }

    // ------------------------------------------------ labels out + path_to_str + left-pack
    if (p.labels)
        for (int t = tid; t < T; t += BS) p.labels[(size_t)n * T + t] = sLab[t];
    if (p.seq || p.seq_len) {
        int *sCnt = reinterpret_cast<int *>(sM);      // BS+1 ints; the row ring is free now (size checked on host)
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

template <int NB, int BS, int LPS>
hipError_t launch_nb_bs(const xb::DecodeParams &p, int vw, hipStream_t stream)
{
    // must mirror the kernel's LDS carve
    const int cpad = BS * 8;
    size_t lds = sizeof(float) * (4 * (size_t)cpad + 5 * (size_t)p.S + 2 * (BS / 64)) + sizeof(int) * 2 * (BS / 64) +
                 sizeof(float) * 4 + (size_t)p.T;
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid(p.N), block(BS);
#define XB_LAUNCH(VW, HB)                                                                                          \
    do {                                                                                                           \
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&crf_decode_kernel<NB, BS, VW, HB, LPS>),         \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        hipLaunchKernelGGL((crf_decode_kernel<NB, BS, VW, HB, LPS>), grid, block, lds, stream, p);                 \
    } while (0)
    if (p.has_blank) {
        if (vw == 4) XB_LAUNCH(4, true); else XB_LAUNCH(1, true);
    } else {
        if (vw == 4) XB_LAUNCH(4, false); else XB_LAUNCH(1, false);
    }
#undef XB_LAUNCH
    return hipGetLastError();
}

// Block = smallest of 64/128/256/512/1024 threads that holds LPS lanes for each of the S states.
template <int NB, int LPS>
hipError_t launch_nb_lps(const xb::DecodeParams &p, int vw, hipStream_t stream)
{
    const int need = LPS * p.S;
    if (need <= 64) return launch_nb_bs<NB, 64, LPS>(p, vw, stream);
    if (need <= 128) return launch_nb_bs<NB, 128, LPS>(p, vw, stream);
    if (need <= 256) return launch_nb_bs<NB, 256, LPS>(p, vw, stream);
    if (need <= 512) return launch_nb_bs<NB, 512, LPS>(p, vw, stream);
    if (LPS == 1 && need <= 1024) return launch_nb_bs<NB, 1024, 1>(p, vw, stream);
    return hipErrorInvalidValue;
}
template <int NB>
hipError_t launch_nb(const xb::DecodeParams &p, int vw, hipStream_t stream)
{
    switch (xb::decode_lanes_per_state(p.S, p.N)) {
    case 4: return launch_nb_lps<NB, 4>(p, vw, stream);
    case 2: return launch_nb_lps<NB, 2>(p, vw, stream);
    default: return launch_nb_lps<NB, 1>(p, vw, stream);
    }
}

}  // namespace

namespace xb {

int decode_lanes_per_state(int S, int N)
{
    if (const char *e = getenv("XB_DECODE_LPS")) {
        const int v = atoi(e);
        if ((v == 1 || v == 2 || v == 4) && v * S <= 512) return v;
    }
    // measured on MI355X (T = 2000).  The kernel is VALU-issue bound, lane clusters duplicate the per-state work: two lanes
    // per state win while the block stays <= 256 threads and the batch leaves the CUs few workgroups each (N = 512, S = 64:
    // 4.7 vs 4.9 ms, S = 125: 5.4 vs 6.7 ms; S = 216 in a 512-thread block loses, 9.7 vs 8.5 ms); with many chunks per CU one
    // lane per state is ahead for the small state space (N = 2048, S = 64: 6.9 vs 8.1 ms) but not for S = 125 (16.9 vs 16.0)
    if (2 * S > 256) return 1;
    if (S <= 64 && N >= 1024) return 1;
    return 2;
}

// Host-side launch.  Shapes are validated here so the kernel's indexing assumptions hold:
//   S = NB^state_len <= 1024, cin = S*(NB+1) or S*NB, ld >= cin (the pack scratch of BS+1 ints always
//   fits the 2*BS*NR*VW-float staging area).
hipError_t launch_crf_decode(const DecodeParams &p, hipStream_t stream)
{
    if (p.S < 1 || p.S > 1024 || p.T < 1 || p.N < 1) return hipErrorInvalidValue;
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

}  // namespace xb

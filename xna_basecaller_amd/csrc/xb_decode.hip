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

// ---- explicit prefetch control ------------------------------------------------------------------
// hipcc's s_waitcnt insertion loses the in-order vmcnt arithmetic across the loop's control flow and waits
// vmcnt(0/1) before every use of the ring, i.e. drains the whole prefetch each step.  Every global LOAD inside
// the sweeps is therefore an inline-asm load the compiler does not track, and each ring slot is consumed behind
// a hand-counted s_waitcnt vmcnt(N): N = (DEPTH-1) * (loads issued per step).  Stores issued in between also
// count on vmcnt, in order, so ignoring them only makes the wait longer, never too short.
__device__ __forceinline__ void ld_asm(float &d, const float *p) { asm volatile("global_load_dword %0, %1, off" : "=v"(d) : "v"(p)); }
__device__ __forceinline__ void ld_asm(f32x2 &d, const float *p) { asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(d) : "v"(p)); }
__device__ __forceinline__ void ld_asm(f32x4 &d, const float *p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(d) : "v"(p)); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// makes `x` opaque at this point: no use of x may be scheduled above a preceding wait_vm
template <typename Tp> __device__ __forceinline__ void pin(Tp &x) { asm volatile("" : "+v"(x)); }

// Score row of one (t, chunk): `cin` floats at the row pointer; all BS threads move groups of VW floats
// (NR groups per thread).  Loads are unconditional: out-of-range groups re-read group 0 and are parked in the
// padding of the LDS buffer (BS*NR*VW floats), so the memory part of the loop is branch-free.
template <int VW, int NR, int BS>
struct ScoreRing {
    using V = typename VecT<VW>::type;
    V r[NR];
    __device__ __forceinline__ void load(const float *row, int cin, int tid)
    {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int g = (tid + BS * i) * VW;
            ld_asm(r[i], row + (g < cin ? g : 0));
        }
    }
    __device__ __forceinline__ void store(float *lds, int tid)
    {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            pin(r[i]);
            *reinterpret_cast<V *>(lds + (tid + BS * i) * VW) = r[i];
        }
    }
};

constexpr int DEPTH = 4;   // time steps of scores in flight per workgroup

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
// value held by the other lane of an adjacent lane pair (quad_perm [1,0,3,2])
__device__ __forceinline__ float dpp_swap(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, false));
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

// LPS = lanes per state in sweep 2 (1 or 2); the block has BS >= LPS*S threads, sweeps 1 and 3 use the first S.
template <int NB, int BS, int VW, bool HB, int LPS>
__global__ __launch_bounds__(BS) void crf_decode_kernel(xb::DecodeParams p)
{
    constexpr int E = NB + 1;
    constexpr int NR = (E + LPS * VW - 1) / (LPS * VW);   // BS*NR*VW >= LPS*S*NR*VW >= S*E >= cin
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int S = p.S, hi = p.hi, T = p.T, N = p.N, cin = p.cin;
    const int tid = threadIdx.x;
    const int n = blockIdx.x;
    const bool act = tid < S;
    constexpr int cpad = BS * 8;                 // staging row (>= BS*NR*VW and >= BS*NRQ*4) incl. parking space
    constexpr int NRQ = (E + 4 * LPS - 1) / (4 * LPS);   // 16-byte groups per thread of a Q row (S*E floats)

    float *sM = reinterpret_cast<float *>(smem_raw);             // [2][cpad]
    float *sQ = sM + 2 * cpad;                                    // [2][cpad] log-posterior rows (sweep 2)
    float *sA = sQ + 2 * cpad;                                    // [2][S]  alpha / beta
    float *sX = sA + 2 * S;                                       // [2][S]  max-plus alpha / beta
    float *sG = sX + 2 * S;                                       // [2][S]  gathered alpha row (sweep 3)
    float *sRv = sG + 2 * S;                                      // [2][BS/64] arg-max partials
    int *sRi = reinterpret_cast<int *>(sRv + 2 * (BS / 64));      // [2][BS/64]
    float *sBc = reinterpret_cast<float *>(sRi + 2 * (BS / 64));  // [4] broadcast scratch
    int8_t *sLab = reinterpret_cast<int8_t *>(sBc + 4);           // [T]

    const float *sc = p.scores + (size_t)n * p.ld;
    const size_t tstride = (size_t)N * p.ld;
    float *alpha = p.alpha + (size_t)n * S;
    float *beta = p.beta + (size_t)n * S;
    float *bmax = p.bmax + (size_t)n * S;
    const size_t sstride = (size_t)N * S;
    const float blank = p.blank;

    ScoreRing<VW, NR, BS> ring[DEPTH];
    const int stid = act ? tid : S - 1;          // stash column read by this thread (clamped, always valid)

    // ------------------------------------------------------------------ sweep 1: Log forward
    if (act) { sA[tid] = 0.0f; alpha[tid] = 0.0f; }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) ring[d].load(sc + (size_t)(d < T ? d : T - 1) * tstride, cin, tid);

    for (int t0 = 0; t0 < T; t0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int t = t0 + d;
            if (t < T) {                                   // wave-uniform
                float *m = sM + (t & 1) * cpad;
                wait_vm<(DEPTH - 1) * NR>();
                ring[d].store(m, tid);
                {
                    const int tn = t + DEPTH < T ? t + DEPTH : T - 1;
                    ring[d].load(sc + (size_t)tn * tstride, cin, tid);
                }
                lds_barrier();
                if (act) {
                    const float *a0 = sA + (t & 1) * S;
                    const int j = tid;
                    float x[E];
                    x[0] = score_at<NB, HB>(m, j, 0, blank) + a0[j];
                    float mx = x[0];
                    const int jq = j / NB;
#pragma unroll
                    for (int k = 1; k < E; ++k) {
                        x[k] = score_at<NB, HB>(m, j, k, blank) + a0[(k - 1) * hi + jq];
                        mx = maxf(mx, x[k]);
                    }
                    float s = xb_expf(x[0] - mx);
#pragma unroll
                    for (int k = 1; k < E; ++k) s += xb_expf(x[k] - mx);
                    const float v = mx + xb_logf(s);
                    sA[((t + 1) & 1) * S + j] = v;
                    alpha[(size_t)(t + 1) * sstride + j] = v;
                }
            }
        }
    }
    wait_vm<0>();
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
        if (act) sG[tid] = xb_expf(aT[tid] - mx);
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
    float *qrow = p.qbuf + (size_t)n * p.ldq;
    const size_t qstride = (size_t)N * p.ldq;
    {
        float aring[DEPTH];
        // LPS == 1: thread = source state.  LPS == 2: an adjacent lane pair shares a state, lane `ph` owns the
        // out-edges e = ph*E0 .. (stay first, then new base b = e-1); exps/logs run in both lanes, the ORDERED
        // logsumexp sum is finished in lane 0 with the partner's terms fetched by DPP in edge order.
        constexpr int E0 = LPS == 2 ? (E + 1) / 2 : E;
        const int i = LPS == 2 ? tid >> 1 : tid;
        const int ph = LPS == 2 ? tid & 1 : 0;
        const bool act2 = i < S;
        const int ic = act2 ? i : S - 1;
        const int kk = ic / hi + 1;
        const int jb = (ic % hi) * NB;
        if (act) {
            sA[(T & 1) * S + tid] = 0.0f;
            sX[(T & 1) * S + tid] = 0.0f;
            beta[(size_t)T * sstride + tid] = 0.0f;
            bmax[(size_t)T * sstride + tid] = 0.0f;
        }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int t = T - 1 - d >= 0 ? T - 1 - d : 0;
            ring[d].load(sc + (size_t)t * tstride, cin, tid);
            ld_asm(aring[d], alpha + (size_t)t * sstride + ic);
        }
        // the Q row of step t is complete once every thread has passed the barrier of step t-1:
        // it is stored (coalesced 16-byte groups) during iteration t-1
        auto store_qrow = [&](int t) {
            const float *src = sQ + (t & 1) * cpad;
            float *dst = qrow + (size_t)t * qstride;
#pragma unroll
            for (int r = 0; r < NRQ; ++r) {
                const int g = (tid + BS * r) * 4;
                if (g < p.ldq) *reinterpret_cast<f32x4 *>(dst + g) = *reinterpret_cast<const f32x4 *>(src + g);
            }
        };
        for (int s0 = 0; s0 < T; s0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int t = T - 1 - (s0 + d);
                if (t >= 0) {
                    float *m = sM + (t & 1) * cpad;
                    float *qs = sQ + (t & 1) * cpad;
                    wait_vm<(DEPTH - 1) * (NR + 1)>();
                    ring[d].store(m, tid);
                    pin(aring[d]);
                    const float a0 = aring[d];
                    {
                        const int tn = t - DEPTH >= 0 ? t - DEPTH : 0;
                        ring[d].load(sc + (size_t)tn * tstride, cin, tid);
                        ld_asm(aring[d], alpha + (size_t)tn * sstride + ic);
                    }
                    lds_barrier();
                    if (t + 1 < T) store_qrow(t + 1);
                    if (act2) {
                        const float *b1 = sA + ((t + 1) & 1) * S;
                        const float *m1 = sX + ((t + 1) & 1) * S;
                        float y[E0], q[E0], mb[E0];
                        bool val[E0];
#pragma unroll
                        for (int r = 0; r < E0; ++r) {
                            const int e = ph * E0 + r;               // edge: 0 = stay, e >= 1 = new base e-1
                            val[r] = e < E;
                            const int ee = val[r] ? e : 0;
                            const int j = ee == 0 ? ic : jb + ee - 1;
                            const int k = ee == 0 ? 0 : kk;
                            const float mv = score_at<NB, HB>(m, j, k, blank);
                            const float bj = b1[j];
                            y[r] = mv + bj;
                            const float xx = ((a0 + mv) + bj) - logZ;
                            q[r] = xb_logf(xb_expf(xx) + 1e-8f);
                            mb[r] = m1[j];
                            if (val[r]) qs[j * E + k] = q[r];
                        }
                        float mx = y[0];                             // r = 0 is always a valid edge
                        float mm = q[0] + mb[0];
#pragma unroll
                        for (int r = 1; r < E0; ++r)
                            if (val[r]) { mx = maxf(mx, y[r]); mm = maxf(mm, q[r] + mb[r]); }
                        if (LPS == 2) {
                            mx = maxf(mx, dpp_swap(mx));             // max is exact: any order
                            mm = maxf(mm, dpp_swap(mm));
                        }
                        float ex[E0];
#pragma unroll
                        for (int r = 0; r < E0; ++r) ex[r] = xb_expf(y[r] - mx);
                        float s = ex[0];
#pragma unroll
                        for (int r = 1; r < E0; ++r) s += ex[r];     // lane 0: edges 0..E0-1 in order
                        if (LPS == 2) {
#pragma unroll
                            for (int r = 0; r < E - E0; ++r) s += dpp_swap(ex[r]);   // then the partner's, in order
                        }
                        const float bv = mx + xb_logf(s);
                        if (ph == 0) {
                            sA[(t & 1) * S + ic] = bv;
                            sX[(t & 1) * S + ic] = mm;
                            beta[(size_t)t * sstride + ic] = bv;
                            bmax[(size_t)t * sstride + ic] = mm;
                        }
                    }
                }
            }
        }
        lds_barrier();
        store_qrow(0);
        wait_vm<0>();
        __syncthreads();
    }

#ifdef XB_LSTM_STAMPS
    if (p.debug_stop == 2) return;   // diagnostic build only: sweeps 1+2
#endif
    // --------------------------- sweep 3: Max forward over Q + per-step arg-max of the max-marginals
    {
        ScoreRing<4, NRQ, BS> qring[DEPTH];
        float mring[DEPTH];
        const int ldq = p.ldq;
        if (act) sX[tid] = 0.0f;
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {
            const int t = d < T ? d : T - 1;
            qring[d].load(qrow + (size_t)t * qstride, ldq, tid);
            ld_asm(mring[d], bmax + (size_t)(t + 1) * sstride + stid);
        }
        const int j = tid;
        const int jq = j / NB;
        const int lane = tid & 63, wave = tid >> 6;
        for (int t0 = 0; t0 < T; t0 += DEPTH) {
#pragma unroll
            for (int d = 0; d < DEPTH; ++d) {
                const int t = t0 + d;
                if (t < T) {
                    float *m = sM + (t & 1) * cpad;
                    wait_vm<(DEPTH - 1) * (NRQ + 1)>();
                    qring[d].store(m, tid);
                    pin(mring[d]);
                    const float m1j = mring[d];
                    {
                        const int tn = t + DEPTH < T ? t + DEPTH : T - 1;
                        qring[d].load(qrow + (size_t)tn * qstride, ldq, tid);
                        ld_asm(mring[d], bmax + (size_t)(tn + 1) * sstride + stid);
                    }
                    lds_barrier();
                    // finalise the previous step's arg-max (partials were written before this barrier)
                    if (tid == 0 && t > 0) {
                        const float *rv = sRv + ((t - 1) & 1) * (BS / 64);
                        const int *ri = sRi + ((t - 1) & 1) * (BS / 64);
                        float bv = rv[0];
                        int bi = ri[0];
                        for (int w = 1; w < BS / 64; ++w)
                            if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
                        sLab[t - 1] = (int8_t)(bi % E);
                    }
                    float best = -__builtin_inff();
                    int bestc = 0x7fffffff;
                    if (act) {
                        const float *am = sX + (t & 1) * S;
                        float mm;
                        {
                            const float Q = m[j * E];
                            const float av = am[j];
                            mm = Q + av;
                            best = (av + Q) + m1j;
                            bestc = j * E;
                        }
#pragma unroll
                        for (int k = 1; k < E; ++k) {
                            const int src = (k - 1) * hi + jq;
                            const float Q = m[j * E + k];
                            const float av = am[src];
                            mm = maxf(mm, Q + av);
                            const float scv = (av + Q) + m1j;
                            if (scv > best) { best = scv; bestc = j * E + k; }
                        }
                        sX[((t + 1) & 1) * S + j] = mm;
                    }
                    wave_argmax(best, bestc);
                    if (lane == 63) {
                        sRv[(t & 1) * (BS / 64) + wave] = best;
                        sRi[(t & 1) * (BS / 64) + wave] = bestc;
                    }
                }
            }
        }
        wait_vm<0>();
        __syncthreads();
        if (tid == 0) {
            const float *rv = sRv + ((T - 1) & 1) * (BS / 64);
            const int *ri = sRi + ((T - 1) & 1) * (BS / 64);
            float bv = rv[0];
            int bi = ri[0];
            for (int w = 1; w < BS / 64; ++w)
                if (rv[w] > bv || (rv[w] == bv && ri[w] < bi)) { bv = rv[w]; bi = ri[w]; }
            sLab[T - 1] = (int8_t)(bi % E);
        }
        __syncthreads();
    }

    // ------------------------------------------------ labels out + path_to_str + left-pack
    if (p.labels)
        for (int t = tid; t < T; t += BS) p.labels[(size_t)n * T + t] = sLab[t];
    if (p.seq || p.seq_len) {
        int *sCnt = reinterpret_cast<int *>(sM);      // BS+1 ints, sM (2*BS*NR*VW floats) is free now
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
    const int cpad = BS * 8;
    size_t lds = sizeof(float) * (4 * (size_t)cpad + 6 * (size_t)p.S + 2 * (BS / 64)) + sizeof(int) * 2 * (BS / 64) +
                 sizeof(float) * 4 + (size_t)p.T;
    lds = (lds + 15) & ~(size_t)15;
    dim3 grid(p.N), block(BS);
#define XB_LAUNCH(VW, HB) hipLaunchKernelGGL((crf_decode_kernel<NB, BS, VW, HB, LPS>), grid, block, lds, stream, p)
    if (p.has_blank) {
        if (vw == 4) XB_LAUNCH(4, true); else if (vw == 2) XB_LAUNCH(2, true); else XB_LAUNCH(1, true);
    } else {
        if (vw == 4) XB_LAUNCH(4, false); else if (vw == 2) XB_LAUNCH(2, false); else XB_LAUNCH(1, false);
    }
#undef XB_LAUNCH
    return hipGetLastError();
}

// two lanes per state in sweep 2 while the doubled block stays within 256 threads (measured on MI355X at
// N = 512: nb = 5 gains 17 %, a 512-thread block for nb = 6 loses 10 % to barrier / arg-max overheads)
template <int NB>
hipError_t launch_nb(const xb::DecodeParams &p, int vw, hipStream_t stream)
{
    if (2 * p.S <= 64) return launch_nb_bs<NB, 64, 2>(p, vw, stream);
    if (2 * p.S <= 128) return launch_nb_bs<NB, 128, 2>(p, vw, stream);
    if (2 * p.S <= 256) return launch_nb_bs<NB, 256, 2>(p, vw, stream);
    if (p.S <= 256) return launch_nb_bs<NB, 256, 1>(p, vw, stream);
    return launch_nb_bs<NB, 1024, 1>(p, vw, stream);
}

}  // namespace

namespace xb {

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
    else if (p.ld % 2 == 0 && p.ld >= ((p.cin + 1) & ~1) && a % 8 == 0) vw = 2;
    switch (p.nb) {
    case 4: return launch_nb<4>(p, vw, stream);
    case 5: return launch_nb<5>(p, vw, stream);
    case 6: return launch_nb<6>(p, vw, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace xb

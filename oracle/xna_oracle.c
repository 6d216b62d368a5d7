/*
 * xna_oracle.c -- CPU restatement of the ub-bonito CRF basecalling hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product (xna_basecaller_amd/) never
 * links, imports or calls anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned" for the CRF arithmetic.  The reference delegates the
 * forward/backward recursions to the third-party wheel ont-seqdist-cuda*==0.0.4
 * (requirements-cuda113.txt:18), which is absent from /root/reference, CUDA-only and has
 * no tests/golden vectors in the reference tree.  The recursions below restate seqdist's
 * published algorithm (sparse.logZ / SequenceDist.posteriors with the Log and Max
 * semirings) anchored on the reference's own call sites:
 *   ub-bonito/bonito/crf/model.py:31-36   state/edge index table          -> xo_crf_idx
 *   ub-bonito/bonito/crf/model.py:41-46   logZ(Ms, idx, alpha_0=0, beta_T=0, S)
 *   ub-bonito/bonito/crf/model.py:63-76   (new_state, dropped_base) edge layout
 *   ub-bonito/bonito/crf/model.py:92-95   viterbi = posteriors(., Max).argmax(2) % len(alphabet)
 *   ub-bonito/bonito/crf/model.py:97-100  path_to_str                      -> xo_pack
 *   ub-bonito/bonito/crf/model.py:215-218 decode_batch: posteriors(+1e-8).log() -> viterbi
 * The encoder restatement (Conv1d/SiLU, LSTM, Linear/tanh/scale/blank pad) follows
 *   ub-bonito/bonito/nn.py:57-68,112-133,176-193,216-220 and crf/model.py:138-160
 * and IS pinned: tests/golden/ holds outputs of the reference's own nn.py modules run in
 * this container (tests/golden/make_golden.py).
 *
 * Floating-point contract of the decode (shared, by specification, with the HIP kernels; the two
 * implementations are written independently):
 *   - all arithmetic in IEEE binary32, round-to-nearest-even, subnormals kept, no contraction
 *     (build with -ffp-contract=off), fused multiply-add only where fmaf() is written;
 *   - exp/log are the fixed polynomial routines xo_expf/xo_logf below (cephes-style; exp clamps its
 *     argument to [-87, 88], log evaluates its polynomial in Estrin form), NOT libm, so that a CPU
 *     and a GPU produce the same bits;
 *   - logsumexp over a list x_0..x_{n-1} (xo_lse):  m = max_k x_k;  s = exp(x_0-m); s += exp(x_k-m)
 *     for k=1..n-1 in list order;  result = m + log(s)  (seqdist's CUDA logsumexp sums in this order);
 *     the final logZ = logsumexp_j(alpha_T[j]) sums in order j = 0..S-1;
 *   - in-edges of state j are listed k = 0..nb (k=0 is the stay/blank edge);
 *     out-edges of state i are listed stay first, then new base b = 0..nb-1;
 *   - posterior of edge (t,j,k):  P = exp(((alpha_t[src] + M) + beta_{t+1}[j]) - logZ),
 *     Q = log(P + 1e-8f);
 *   - max-marginal of edge c=j*E+k at time t: (amax_t[src] + Q) + bmax_{t+1}[j];
 *     label[t] = (lowest flat index attaining the maximum) % E.
 * The recursions stay in the LOG domain on purpose: alpha ~ 1e3..1e4 is then rounded to an ulp of
 * 1e-4..1e-3 at every step, which absorbs the last-bit differences between any two accurate exp/log
 * implementations -- differently rounded log-domain builds (this contract, libm, libm with seqdist's
 * softmax-normalised posteriors) differ in a few labels per million time steps of the census
 * (tools/decode_census.py), a float64-accurate evaluation in ~400 per million.  xo_decode_scaled() and xo_decode_logdomain() are those comparison
 * models, NOT part of the contract.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define XO_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------ */
/* Specified math                                                                        */
/* ------------------------------------------------------------------------------------ */

static inline float xo_bits2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static inline uint32_t xo_f2bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

/* exp(x) with the argument clamped to [-87, 88] (no flush to zero: exp(-87) = 1.6e-38 stands for anything smaller).
 * n = round-to-nearest-even of x*log2(e) comes from ONE fma against the magic constant 1.5*2^23 (the integer then sits in
 * the low mantissa bits of t, which also gives 2^n by a shift): no rint / float->int conversion, every step has a packed
 * (two values per instruction) form on the GPU. */
static inline float xo_expf_i(float x)
{
    x = x < -87.0f ? -87.0f : x;
    x = x > 88.0f ? 88.0f : x;
    const float t = fmaf(x, 1.44269504088896341f, 12582912.0f);
    const float n = t - 12582912.0f;
    float r = fmaf(n, -0.693359375f, x);
    r = fmaf(n, 2.12194440e-4f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float r2 = r * r;
    float y = fmaf(p, r2, r) + 1.0f;
    float s = xo_bits2f((xo_f2bits(t) << 23) + 0x3f800000u);      /* 2^n, n in [-126, 127] */
    return y * s;
}

/* log(x) for normal positive x.  The degree-8 polynomial is evaluated in Estrin form (dependency depth 4 instead of 8:
 * the log sits on the critical path of every time step of the recursions). */
static inline float xo_logf_i(float x)
{
    uint32_t ix = xo_f2bits(x);
    int32_t e = (int32_t)(ix >> 23) - 127;
    float m = xo_bits2f((ix & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421356237309505f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float z = f * f;
    float z2 = z * z;
    float z4 = z2 * z2;
    /* p(f) = c0 + c1 f + ... + c8 f^8 */
    float q01 = fmaf(-2.4999993993e-1f, f, 3.3333331174e-1f);
    float q23 = fmaf(-1.6668057665e-1f, f, 2.0000714765e-1f);
    float q45 = fmaf(-1.2420140846e-1f, f, 1.4249322787e-1f);
    float q67 = fmaf(-1.1514610310e-1f, f, 1.1676998740e-1f);
    float q03 = fmaf(q23, z, q01);
    float q47 = fmaf(q67, z, q45);
    float q07 = fmaf(q47, z2, q03);
    float p = fmaf(7.0376836292e-2f, z4, q07);
    float y = (f * z) * p;
    float fe = (float)e;
    y = fmaf(fe, -2.12194440e-4f, y);
    y = fmaf(-0.5f, z, y);
    float r = f + y;
    r = fmaf(fe, 0.693359375f, r);
    return r;
}

XO_API float xo_expf(float x) { return xo_expf_i(x); }
XO_API float xo_logf(float x) { return xo_logf_i(x); }

XO_API void xo_expf_array(const float *x, float *y, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) y[i] = xo_expf_i(x[i]);
}
XO_API void xo_logf_array(const float *x, float *y, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) y[i] = xo_logf_i(x[i]);
}

/* ------------------------------------------------------------------------------------ */
/* CRF state table (crf/model.py:31-36)                                                  */
/* ------------------------------------------------------------------------------------ */

static int64_t ipow(int64_t b, int e) { int64_t r = 1; while (e-- > 0) r *= b; return r; }

/* idx[j*E + 0] = j ; idx[j*E + k] = (k-1) * nb^(sl-1) + j / nb  for k >= 1 */
XO_API void xo_crf_idx(int nb, int sl, int32_t *idx)
{
    const int64_t S = ipow(nb, sl), hi = ipow(nb, sl - 1);
    const int E = nb + 1;
    for (int64_t j = 0; j < S; ++j) {
        idx[j * E] = (int32_t)j;
        for (int k = 1; k < E; ++k) idx[j * E + k] = (int32_t)((k - 1) * hi + j / nb);
    }
}

/* ------------------------------------------------------------------------------------ */
/* decode_batch (crf/model.py:215-218) -- THE CONTRACT                                   */
/* ------------------------------------------------------------------------------------ */

/* logsumexp of x[0..n): m = max; s = exp(x_0 - m), s += exp(x_k - m) for k = 1..n-1 in list order; m + log(s).
 * (Sequential order on purpose: it is the order of seqdist's CUDA logsumexp, and the census shows that the summation
 * order is what moves alpha's last bit most often.) */
static inline float xo_lse(const float *x, int n)
{
    float m = x[0];
    for (int k = 1; k < n; ++k) m = x[k] > m ? x[k] : m;
    float s = xo_expf_i(x[0] - m);
    for (int k = 1; k < n; ++k) s += xo_expf_i(x[k] - m);
    return m + xo_logf_i(s);
}

/*
 * scores : (T, N, Cin) fp32.  has_blank != 0: Cin = S*E with the stay score in column 0 of
 *          every state row (LinearCRFEncoder expand_blanks layout, nn.py:123-130).
 *          has_blank == 0: Cin = S*nb and the stay score is the constant `blank`.
 * labels : (N, T) int8 out -- argmax % E per time step (0 = no base emitted).
 * Optional outputs (may be NULL): alpha (T+1,N,S), beta (T+1,N,S), logz (N),
 *          post (T,N,S*E) posteriors P, qlog (T,N,S*E) = log(P + 1e-8), amax (T+1,N,S), bmax (T+1,N,S).
 */
XO_API int xo_decode(const float *scores, int T, int N, int nb, int sl, int has_blank, float blank,
                     int8_t *labels, float *alpha_out, float *beta_out, float *logz_out,
                     float *post_out, float *qlog_out, float *amax_out, float *bmax_out)
{
    const int S = (int)ipow(nb, sl), E = nb + 1, C = S * E;
    const int Cin = has_blank ? C : S * nb;
    const int hi = (int)ipow(nb, sl - 1);
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)C);
    if (!idx) return -1;
    xo_crf_idx(nb, sl, idx);
    int err = 0;

#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; ++n) {
        const size_t TS = (size_t)(T + 1) * S;
        float *al = (float *)malloc(sizeof(float) * TS);
        float *be = (float *)malloc(sizeof(float) * TS);
        float *bm = (float *)malloc(sizeof(float) * TS);
        float *M = (float *)malloc(sizeof(float) * (size_t)C);
        float *Q = (float *)malloc(sizeof(float) * (size_t)T * C);
        float *am = (float *)malloc(sizeof(float) * 2 * (size_t)S);
        if (!al || !be || !bm || !M || !Q || !am) {
            err = -1;
            free(al); free(be); free(bm); free(M); free(Q); free(am);
            continue;
        }
#define LOADM(t)                                                                                    \
    do {                                                                                            \
        const float *row = scores + ((size_t)(t) * N + n) * Cin;                                    \
        if (has_blank) memcpy(M, row, sizeof(float) * (size_t)C);                                   \
        else for (int j_ = 0; j_ < S; ++j_) {                                                       \
            M[j_ * E] = blank;                                                                      \
            for (int k_ = 1; k_ < E; ++k_) M[j_ * E + k_] = row[j_ * nb + k_ - 1];                  \
        }                                                                                           \
    } while (0)

        /* ---- Log semiring forward: alpha ---- */
        for (int j = 0; j < S; ++j) al[j] = 0.0f;
        for (int t = 0; t < T; ++t) {
            LOADM(t);
            const float *a0 = al + (size_t)t * S;
            float *a1 = al + (size_t)(t + 1) * S;
            for (int j = 0; j < S; ++j) {
                float x[16] = {0};
                for (int k = 0; k < E; ++k) x[k] = M[j * E + k] + a0[idx[j * E + k]];
                a1[j] = xo_lse(x, E);
            }
        }
        float logZ;
        {   /* logsumexp over the final states, summed sequentially in state order */
            const float *aT = al + (size_t)T * S;
            float m = -INFINITY;
            for (int j = 0; j < S; ++j) m = aT[j] > m ? aT[j] : m;
            float s = xo_expf_i(aT[0] - m);
            for (int j = 1; j < S; ++j) s += xo_expf_i(aT[j] - m);
            logZ = m + xo_logf_i(s);
        }
        if (logz_out) logz_out[n] = logZ;

        /* ---- Log semiring backward (beta) fused with Max semiring backward (bmax); Q = log(P + 1e-8) is kept ---- */
        for (int i = 0; i < S; ++i) { be[(size_t)T * S + i] = 0.0f; bm[(size_t)T * S + i] = 0.0f; }
        for (int t = T - 1; t >= 0; --t) {
            LOADM(t);
            const float *a0 = al + (size_t)t * S;
            const float *b1 = be + (size_t)(t + 1) * S;
            const float *m1 = bm + (size_t)(t + 1) * S;
            float *b0 = be + (size_t)t * S;
            float *m0 = bm + (size_t)t * S;
            float *qrow = Q + (size_t)t * C;
            for (int i = 0; i < S; ++i) {
                /* out-edges of i: stay (j=i,k=0), then new base b: j=(i%hi)*nb+b, k=i/hi+1 */
                float y[16] = {0};
                const int kk = i / hi + 1;
                float mm = -INFINITY;
                for (int e = 0; e < E; ++e) {
                    const int j = e == 0 ? i : (i % hi) * nb + e - 1;
                    const size_t c = (size_t)j * E + (e == 0 ? 0 : kk);
                    const float mv = M[c];
                    y[e] = mv + b1[j];
                    const float xx = ((a0[i] + mv) + b1[j]) - logZ;
                    const float P = xo_expf_i(xx);
                    const float q = xo_logf_i(P + 1e-8f);
                    qrow[c] = q;
                    if (post_out) post_out[((size_t)t * N + n) * C + c] = P;
                    const float v = q + m1[j];
                    mm = v > mm ? v : mm;
                }
                b0[i] = xo_lse(y, E);
                m0[i] = mm;
            }
        }

        /* ---- Max semiring forward over Q with per-step arg-max of the max-marginals ---- */
        float *am0 = am, *am1 = am + S;
        for (int j = 0; j < S; ++j) am0[j] = 0.0f;
        if (amax_out) for (int j = 0; j < S; ++j) amax_out[(size_t)n * S + j] = 0.0f;
        for (int t = 0; t < T; ++t) {
            const float *qrow = Q + (size_t)t * C;
            const float *m1 = bm + (size_t)(t + 1) * S;
            float best = -INFINITY;
            int bestc = 0;
            for (int j = 0; j < S; ++j) {
                float mm = -INFINITY;
                for (int k = 0; k < E; ++k) {
                    const int src = idx[j * E + k];
                    const float q = qrow[j * E + k];
                    const float v = q + am0[src];
                    mm = v > mm ? v : mm;
                    const float sc = (am0[src] + q) + m1[j];
                    if (sc > best) { best = sc; bestc = j * E + k; }
                }
                am1[j] = mm;
            }
            labels[(size_t)n * T + t] = (int8_t)(bestc % E);
            if (amax_out) memcpy(amax_out + ((size_t)(t + 1) * N + n) * S, am1, sizeof(float) * (size_t)S);
            float *tmp = am0; am0 = am1; am1 = tmp;
        }
#undef LOADM
        if (alpha_out) for (int t = 0; t <= T; ++t)
            memcpy(alpha_out + ((size_t)t * N + n) * S, al + (size_t)t * S, sizeof(float) * (size_t)S);
        if (beta_out) for (int t = 0; t <= T; ++t)
            memcpy(beta_out + ((size_t)t * N + n) * S, be + (size_t)t * S, sizeof(float) * (size_t)S);
        if (qlog_out) for (int t = 0; t < T; ++t)
            memcpy(qlog_out + ((size_t)t * N + n) * C, Q + (size_t)t * C, sizeof(float) * (size_t)C);
        if (bmax_out) for (int t = 0; t <= T; ++t)
            memcpy(bmax_out + ((size_t)t * N + n) * S, bm + (size_t)t * S, sizeof(float) * (size_t)S);
        free(al); free(be); free(bm); free(M); free(Q); free(am);
    }
    free(idx);
    return err;
}

/* ------------------------------------------------------------------------------------ */
/* Comparison models for the tie-margin census (tests only; NOT the contract)            */
/* ------------------------------------------------------------------------------------ */

/*
 * (1) xo_decode_scaled: the same sums evaluated on SCALED PROBABILITIES (the classical scaled forward-backward form):
 *     with w = exp(score),  a_{t+1}[j] = r_t * sum_k w_k a_t[src_k],  r_t an exact power of two that renormalises the
 *     vector, the integer Ka[t] its running exponent.  In fp32 this is accurate to ~1e-6 against float64, whereas the
 *     log-domain recursions carry alpha ~ 1e3..1e4 with an ulp of 1e-4..1e-3 -- it stands in for "the exact answer"
 *     at full-size T where a float64 autograd evaluation is too slow.
 */

/* exact power of two 2^(bits-127) from a biased exponent field in [1, 254] */
static inline float xo_pow2_field(int field) { return xo_bits2f((uint32_t)field << 23); }

/* Per-step normaliser of the scaled recursions: the exact power of two r = 2^-(floor(log2 m)) that brings the
 * largest entry m of the state vector into [1, 2); *shift receives log2(r).  The exponent field of r is clamped to
 * [1, 254] (only reached by vectors outside the supported score range). */
static inline float xo_norm_scale(float m, int *shift)
{
    const int eb = (int)((xo_f2bits(m) >> 23) & 0xffu);
    int rf = 254 - eb;
    if (rf < 1) rf = 1;
    *shift = rf - 127;
    return xo_pow2_field(rf);
}

/*
 * scores : (T, N, Cin) fp32.  has_blank != 0: Cin = S*E with the stay score in column 0 of
 *          every state row (LinearCRFEncoder expand_blanks layout, nn.py:123-130).
 *          has_blank == 0: Cin = S*nb and the stay score is the constant `blank`.
 * labels : (N, T) int8 out -- argmax % E per time step (0 = no base emitted).
 * Optional outputs (may be NULL): avec / bvec (T+1,N,S) the scaled forward / backward vectors, aexp / bexp (T+1,N)
 *          their binary exponents (alpha_t[j] = log(avec) + aexp*ln2), logz (N), post (T,N,S*E) posteriors P,
 *          qlog (T,N,S*E) = log(P + 1e-8), amax (T+1,N,S), bmax (T+1,N,S).
 */
XO_API int xo_decode_scaled(const float *scores, int T, int N, int nb, int sl, int has_blank, float blank,
                     int8_t *labels, float *avec_out, float *bvec_out, int32_t *aexp_out, int32_t *bexp_out,
                     float *logz_out, float *post_out, float *qlog_out, float *amax_out, float *bmax_out)
{
    const int S = (int)ipow(nb, sl), E = nb + 1, C = S * E;
    const int Cin = has_blank ? C : S * nb;
    const int hi = (int)ipow(nb, sl - 1);
    const int H = (E + 1) / 2;                    /* edges 0..H-1 form the first half-chain, H..E-1 the second */
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)C);
    if (!idx) return -1;
    xo_crf_idx(nb, sl, idx);
    const float wblank = xo_expf_i(blank);
    int err = 0;

#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; ++n) {
        const size_t TS = (size_t)(T + 1) * S;
        float *av = (float *)malloc(sizeof(float) * TS);
        float *bv = (float *)malloc(sizeof(float) * 2 * (size_t)S);
        float *bm = (float *)malloc(sizeof(float) * TS);
        float *W = (float *)malloc(sizeof(float) * (size_t)C);
        float *Q = (float *)malloc(sizeof(float) * (size_t)T * C);
        float *am = (float *)malloc(sizeof(float) * 2 * (size_t)S);
        int32_t *Ka = (int32_t *)malloc(sizeof(int32_t) * (size_t)(T + 1));
        if (!av || !bv || !bm || !W || !Q || !am || !Ka) {
            err = -1;
            free(av); free(bv); free(bm); free(W); free(Q); free(am); free(Ka);
            continue;
        }
        /* edge weights of one time step: w = exp(score), the stay weight in column 0 */
#define LOADW(t)                                                                                    \
    do {                                                                                            \
        const float *row = scores + ((size_t)(t) * N + n) * Cin;                                    \
        if (has_blank) for (int c_ = 0; c_ < C; ++c_) W[c_] = xo_expf_i(row[c_]);                   \
        else for (int j_ = 0; j_ < S; ++j_) {                                                       \
            W[j_ * E] = wblank;                                                                     \
            for (int k_ = 1; k_ < E; ++k_) W[j_ * E + k_] = xo_expf_i(row[j_ * nb + k_ - 1]);       \
        }                                                                                           \
    } while (0)

        /* ---- Log semiring forward, evaluated on scaled probabilities: a_t[j] * 2^Ka[t] = exp(alpha_t[j]) ---- */
        for (int j = 0; j < S; ++j) av[j] = 1.0f;
        Ka[0] = 0;
        for (int t = 0; t < T; ++t) {
            LOADW(t);
            const float *a0 = av + (size_t)t * S;
            float *a1 = av + (size_t)(t + 1) * S;
            float m = a0[0];
            for (int j = 1; j < S; ++j) m = a0[j] > m ? a0[j] : m;
            int shift;
            const float r = xo_norm_scale(m, &shift);
            Ka[t + 1] = Ka[t] - shift;
            for (int j = 0; j < S; ++j) {
                const float *w = W + (size_t)j * E;
                const int32_t *src = idx + (size_t)j * E;
                float p0 = w[0] * a0[src[0]];
                for (int k = 1; k < H; ++k) p0 = fmaf(w[k], a0[src[k]], p0);
                float p1 = w[H] * a0[src[H]];
                for (int k = H + 1; k < E; ++k) p1 = fmaf(w[k], a0[src[k]], p1);
                a1[j] = (p0 + p1) * r;
            }
        }
        float zs;
        {
            const float *aT = av + (size_t)T * S;
            zs = aT[0];
            for (int j = 1; j < S; ++j) zs += aT[j];
        }
        const float rZ = 1.0f / zs;
        if (logz_out) logz_out[n] = (float)Ka[T] * 0.693147180559945f + xo_logf_i(zs);
        if (aexp_out) for (int t = 0; t <= T; ++t) aexp_out[(size_t)t * N + n] = Ka[t];

        /* ---- backward on scaled probabilities (b) fused with the Max semiring backward over Q (bmax) ---- */
        float *b1 = bv, *b0 = bv + S;
        for (int i = 0; i < S; ++i) { b1[i] = 1.0f; bm[(size_t)T * S + i] = 0.0f; }
        int Kb = 0;
        if (bvec_out) for (int i = 0; i < S; ++i) bvec_out[((size_t)T * N + n) * S + i] = 1.0f;
        if (bexp_out) bexp_out[(size_t)T * N + n] = 0;
        for (int t = T - 1; t >= 0; --t) {
            LOADW(t);
            const float *a0 = av + (size_t)t * S;
            const float *m1 = bm + (size_t)(t + 1) * S;
            float *m0 = bm + (size_t)t * S;
            float *qrow = Q + (size_t)t * C;
            float m = b1[0];
            for (int i = 1; i < S; ++i) m = b1[i] > m ? b1[i] : m;
            int shift;
            const float r = xo_norm_scale(m, &shift);
            /* P = a_t[i] w b_{t+1}[j] 2^(Ka[t] + Kb[t+1] - Ka[T]) / zs */
            int kt = Ka[t] + Kb - Ka[T];
            kt = kt < -126 ? -126 : (kt > 127 ? 127 : kt);
            const float g = rZ * xo_pow2_field(kt + 127);
            for (int i = 0; i < S; ++i) {
                /* out-edges of i: stay (j=i,k=0), then new base b: j=(i%hi)*nb+b, k=i/hi+1 */
                float we[16], be[16];
                int dst[16], col[16];
                const int kk = i / hi + 1;
                dst[0] = i; col[0] = 0;
                for (int b = 0; b < nb; ++b) { dst[b + 1] = (i % hi) * nb + b; col[b + 1] = kk; }
                for (int e = 0; e < E; ++e) { we[e] = W[dst[e] * E + col[e]]; be[e] = b1[dst[e]]; }
                float q0 = we[0] * be[0];
                for (int e = 1; e < H; ++e) q0 = fmaf(we[e], be[e], q0);
                float q1 = we[H] * be[H];
                for (int e = H + 1; e < E; ++e) q1 = fmaf(we[e], be[e], q1);
                b0[i] = (q0 + q1) * r;
                float mm = -INFINITY;
                for (int e = 0; e < E; ++e) {
                    const float P = ((a0[i] * we[e]) * be[e]) * g;
                    const float q = xo_logf_i(P + 1e-8f);
                    const size_t c = (size_t)dst[e] * E + col[e];
                    qrow[c] = q;
                    if (post_out) post_out[((size_t)t * N + n) * C + c] = P;
                    const float v = q + m1[dst[e]];
                    mm = v > mm ? v : mm;
                }
                m0[i] = mm;
            }
            Kb -= shift;
            if (bvec_out) memcpy(bvec_out + ((size_t)t * N + n) * S, b0, sizeof(float) * (size_t)S);
            if (bexp_out) bexp_out[(size_t)t * N + n] = Kb;
            float *tmp = b0; b0 = b1; b1 = tmp;
        }

        /* ---- Max semiring forward over Q with per-step arg-max of the max-marginals ---- */
        float *am0 = am, *am1 = am + S;
        for (int j = 0; j < S; ++j) am0[j] = 0.0f;
        if (amax_out) for (int j = 0; j < S; ++j) amax_out[(size_t)n * S + j] = 0.0f;
        for (int t = 0; t < T; ++t) {
            const float *qrow = Q + (size_t)t * C;
            const float *m1 = bm + (size_t)(t + 1) * S;
            float best = -INFINITY;
            int bestc = 0;
            for (int j = 0; j < S; ++j) {
                float mm = -INFINITY;
                for (int k = 0; k < E; ++k) {
                    const int src = idx[j * E + k];
                    const float q = qrow[j * E + k];
                    const float v = q + am0[src];
                    mm = v > mm ? v : mm;
                    const float sc = (am0[src] + q) + m1[j];
                    if (sc > best) { best = sc; bestc = j * E + k; }
                }
                am1[j] = mm;
            }
            labels[(size_t)n * T + t] = (int8_t)(bestc % E);
            if (amax_out) memcpy(amax_out + ((size_t)(t + 1) * N + n) * S, am1, sizeof(float) * (size_t)S);
            float *tmp = am0; am0 = am1; am1 = tmp;
        }
#undef LOADW
        if (avec_out) for (int t = 0; t <= T; ++t)
            memcpy(avec_out + ((size_t)t * N + n) * S, av + (size_t)t * S, sizeof(float) * (size_t)S);
        if (qlog_out) for (int t = 0; t < T; ++t)
            memcpy(qlog_out + ((size_t)t * N + n) * C, Q + (size_t)t * C, sizeof(float) * (size_t)C);
        if (bmax_out) for (int t = 0; t <= T; ++t)
            memcpy(bmax_out + ((size_t)t * N + n) * S, bm + (size_t)t * S, sizeof(float) * (size_t)S);
        free(av); free(bv); free(bm); free(W); free(Q); free(am); free(Ka);
    }
    free(idx);
    return err;
}

/*
 * (2) xo_decode_logdomain: the decode evaluated the way a straightforward log-domain fp32 implementation (such as seqdist's
 * CUDA kernels) would: alpha/beta by logsumexp (max, ordered sum of exp, log), posteriors
 * P = exp(alpha + M + beta - logZ), Q = log(P + 1e-8), then the Max-semiring passes.
 * math: 0 = the polynomial xo_expf/xo_logf, 1 = libm expf/logf, 2 = libm with the posteriors normalised per time
 * step by a softmax over all edges, P = exp(x - max x) / sum exp(x - max x) with x = (M + alpha[src]) + beta[dst]
 * (the form seqdist's Log.dsum takes) instead of exp(x - logZ).
 * gap (N,T) optional: max-marginal of the winning edge minus the best max-marginal among edges with a
 * DIFFERENT label (the margin a differently rounded implementation would have to overcome to flip the call).
 */
static inline float cm_exp(float x, int math) { return math ? expf(x) : xo_expf_i(x); }
static inline float cm_log(float x, int math) { return math ? logf(x) : xo_logf_i(x); }
/* max and sum of exp(x - max) over all edges of one time step, x = (M + alpha[src]) + beta[dst] */
static void cm_softmax_norm(const float *M, const float *a0, const float *b1, const int32_t *idx, int S, int E,
                            float *mx, float *sum)
{
    float m = -INFINITY;
    for (int j = 0; j < S; ++j)
        for (int k = 0; k < E; ++k) {
            const float x = (M[j * E + k] + a0[idx[j * E + k]]) + b1[j];
            m = x > m ? x : m;
        }
    float s = 0.0f;
    for (int j = 0; j < S; ++j)
        for (int k = 0; k < E; ++k) s += expf(((M[j * E + k] + a0[idx[j * E + k]]) + b1[j]) - m);
    *mx = m;
    *sum = s;
}

XO_API int xo_decode_logdomain(const float *scores, int T, int N, int nb, int sl, int has_blank, float blank,
                               int math, int8_t *labels, float *gap_out, float *logz_out, float *post_out)
{
    const int S = (int)ipow(nb, sl), E = nb + 1, C = S * E;
    const int Cin = has_blank ? C : S * nb;
    const int hi = (int)ipow(nb, sl - 1);
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)C);
    if (!idx) return -1;
    xo_crf_idx(nb, sl, idx);
    int err = 0;

#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; ++n) {
        const size_t TS = (size_t)(T + 1) * S;
        float *al = (float *)malloc(sizeof(float) * TS);
        float *be = (float *)malloc(sizeof(float) * TS);
        float *bm = (float *)malloc(sizeof(float) * TS);
        float *M = (float *)malloc(sizeof(float) * (size_t)C);
        float *am = (float *)malloc(sizeof(float) * 2 * (size_t)S);
        if (!al || !be || !bm || !M || !am) {
            err = -1;
            free(al); free(be); free(bm); free(M); free(am);
            continue;
        }
#define LOADM(t)                                                                                    \
    do {                                                                                            \
        const float *row = scores + ((size_t)(t) * N + n) * Cin;                                    \
        if (has_blank) memcpy(M, row, sizeof(float) * (size_t)C);                                   \
        else for (int j_ = 0; j_ < S; ++j_) {                                                       \
            M[j_ * E] = blank;                                                                      \
            for (int k_ = 1; k_ < E; ++k_) M[j_ * E + k_] = row[j_ * nb + k_ - 1];                  \
        }                                                                                           \
    } while (0)
        for (int j = 0; j < S; ++j) al[j] = 0.0f;
        for (int t = 0; t < T; ++t) {
            LOADM(t);
            const float *a0 = al + (size_t)t * S;
            float *a1 = al + (size_t)(t + 1) * S;
            for (int j = 0; j < S; ++j) {
                float x[16];
                float m = -INFINITY;
                for (int k = 0; k < E; ++k) {
                    x[k] = M[j * E + k] + a0[idx[j * E + k]];
                    m = x[k] > m ? x[k] : m;
                }
                float s = cm_exp(x[0] - m, math);
                for (int k = 1; k < E; ++k) s += cm_exp(x[k] - m, math);
                a1[j] = m + cm_log(s, math);
            }
        }
        float logZ;
        {
            const float *aT = al + (size_t)T * S;
            float m = -INFINITY;
            for (int j = 0; j < S; ++j) m = aT[j] > m ? aT[j] : m;
            float s = cm_exp(aT[0] - m, math);
            for (int j = 1; j < S; ++j) s += cm_exp(aT[j] - m, math);
            logZ = m + cm_log(s, math);
        }
        if (logz_out) logz_out[n] = logZ;
        for (int i = 0; i < S; ++i) { be[(size_t)T * S + i] = 0.0f; bm[(size_t)T * S + i] = 0.0f; }
        for (int t = T - 1; t >= 0; --t) {
            LOADM(t);
            const float *a0 = al + (size_t)t * S;
            const float *b1 = be + (size_t)(t + 1) * S;
            const float *m1 = bm + (size_t)(t + 1) * S;
            float *b0 = be + (size_t)t * S;
            float *m0 = bm + (size_t)t * S;
            float smx = 0.0f, ssum = 1.0f;
            if (math == 2) cm_softmax_norm(M, a0, b1, idx, S, E, &smx, &ssum);
            for (int i = 0; i < S; ++i) {
                float y[16], q[16];
                int dst[16];
                const int kk = i / hi + 1;
                dst[0] = i;
                for (int b = 0; b < nb; ++b) dst[b + 1] = (i % hi) * nb + b;
                float m = -INFINITY;
                for (int e = 0; e < E; ++e) {
                    const int j = dst[e];
                    const float mv = M[j * E + (e == 0 ? 0 : kk)];
                    y[e] = mv + b1[j];
                    m = y[e] > m ? y[e] : m;
                    const float P = math == 2 ? cm_exp(((mv + a0[i]) + b1[j]) - smx, 1) / ssum
                                              : cm_exp(((a0[i] + mv) + b1[j]) - logZ, math);
                    q[e] = cm_log(P + 1e-8f, math);
                    if (post_out) post_out[((size_t)t * N + n) * C + (size_t)j * E + (e == 0 ? 0 : kk)] = P;
                }
                float s = cm_exp(y[0] - m, math);
                for (int e = 1; e < E; ++e) s += cm_exp(y[e] - m, math);
                b0[i] = m + cm_log(s, math);
                float mm = q[0] + m1[dst[0]];
                for (int e = 1; e < E; ++e) {
                    const float v = q[e] + m1[dst[e]];
                    mm = v > mm ? v : mm;
                }
                m0[i] = mm;
            }
        }
        float *am0 = am, *am1 = am + S;
        for (int j = 0; j < S; ++j) am0[j] = 0.0f;
        for (int t = 0; t < T; ++t) {
            LOADM(t);
            const float *a0 = al + (size_t)t * S;
            const float *b1 = be + (size_t)(t + 1) * S;
            const float *m1 = bm + (size_t)(t + 1) * S;
            float bestk[16];
            for (int k = 0; k < E; ++k) bestk[k] = -INFINITY;
            float smx = 0.0f, ssum = 1.0f;
            if (math == 2) cm_softmax_norm(M, a0, b1, idx, S, E, &smx, &ssum);
            float best = -INFINITY;
            int bestc = 0;
            for (int j = 0; j < S; ++j) {
                float mm = -INFINITY;
                for (int k = 0; k < E; ++k) {
                    const int src = idx[j * E + k];
                    const float Pv = math == 2 ? cm_exp(((M[j * E + k] + a0[src]) + b1[j]) - smx, 1) / ssum
                                               : cm_exp(((a0[src] + M[j * E + k]) + b1[j]) - logZ, math);
                    const float Qv = cm_log(Pv + 1e-8f, math);
                    const float v = Qv + am0[src];
                    mm = v > mm ? v : mm;
                    const float sc = (am0[src] + Qv) + m1[j];
                    if (sc > best) { best = sc; bestc = j * E + k; }
                    if (sc > bestk[k]) bestk[k] = sc;
                }
                am1[j] = mm;
            }
            labels[(size_t)n * T + t] = (int8_t)(bestc % E);
            if (gap_out) {
                float second = -INFINITY;
                for (int k = 0; k < E; ++k) if (k != bestc % E && bestk[k] > second) second = bestk[k];
                gap_out[(size_t)n * T + t] = best - second;
            }
            float *tmp = am0; am0 = am1; am1 = tmp;
        }
#undef LOADM
        free(al); free(be); free(bm); free(M); free(am);
    }
    free(idx);
    return err;
}

/*
 * path_to_str (crf/model.py:97-100) + the left-pack of compute_scores (crf/basecall.py:60-67):
 * seq[n, 0:len] = alphabet[label] for label != 0, zero padded to T; qstring 'O' (79) likewise.
 * alphabet: E bytes, e.g. "NACGTXY".
 */
XO_API void xo_pack(const int8_t *labels, int N, int T, const char *alphabet,
                    int8_t *seq, int8_t *qstring, int32_t *seq_len)
{
    for (int n = 0; n < N; ++n) {
        int len = 0;
        for (int t = 0; t < T; ++t) {
            const int l = labels[(size_t)n * T + t];
            if (l != 0) {
                seq[(size_t)n * T + len] = (int8_t)alphabet[l];
                if (qstring) qstring[(size_t)n * T + len] = 79;
                ++len;
            }
        }
        for (int t = len; t < T; ++t) {
            seq[(size_t)n * T + t] = 0;
            if (qstring) qstring[(size_t)n * T + t] = 0;
        }
        if (seq_len) seq_len[n] = len;
    }
}

/* ------------------------------------------------------------------------------------ */
/* Encoder (nn.py) -- fp32, tolerance-compared (summation order not part of a contract)  */
/* ------------------------------------------------------------------------------------ */

static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

/*
 * Conv1d + bias + SiLU (nn.py:57-68; torch.nn.Conv1d cross-correlation).
 * x: (N, Cin, L)   w: (Cout, Cin, K)   b: (Cout)   y: (N, Cout, Lout),
 * Lout = (L + 2*pad - K)/stride + 1.
 */
XO_API void xo_conv1d_silu(const float *x, int N, int Cin, int L, const float *w, const float *b,
                           int Cout, int K, int stride, int pad, float *y)
{
    const int Lout = (L + 2 * pad - K) / stride + 1;
#pragma omp parallel for collapse(2) schedule(static)
    for (int n = 0; n < N; ++n)
        for (int co = 0; co < Cout; ++co) {
            float *yo = y + ((size_t)n * Cout + co) * Lout;
            for (int t = 0; t < Lout; ++t) {
                float acc = b ? b[co] : 0.0f;
                const int base = t * stride - pad;
                for (int ci = 0; ci < Cin; ++ci) {
                    const float *xi = x + ((size_t)n * Cin + ci) * L;
                    const float *wi = w + ((size_t)co * Cin + ci) * K;
                    for (int k = 0; k < K; ++k) {
                        const int p = base + k;
                        if (p >= 0 && p < L) acc += wi[k] * xi[p];
                    }
                }
                yo[t] = acc * sigmoidf_(acc);
            }
        }
}

/* C(M,Nn) = A(M,K) * B(Nn,K)^T + bias(Nn) ; row-major, B given as [out][in] (torch layout).
 * Work is split over (row block, column block) so that the small-M recurrent products use all cores. */
static inline float dot8(const float *ap, const float *bp, int K)
{
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    int k = 0;
    for (; k + 8 <= K; k += 8) {
        s0 += ap[k] * bp[k];         s1 += ap[k + 1] * bp[k + 1];
        s2 += ap[k + 2] * bp[k + 2]; s3 += ap[k + 3] * bp[k + 3];
        s4 += ap[k + 4] * bp[k + 4]; s5 += ap[k + 5] * bp[k + 5];
        s6 += ap[k + 6] * bp[k + 6]; s7 += ap[k + 7] * bp[k + 7];
    }
    float s = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
    for (; k < K; ++k) s += ap[k] * bp[k];
    return s;
}

static void gemm_nt_bias(const float *A, const float *B, const float *bias, float *C,
                         int64_t M, int Nn, int K)
{
    const int MB = 16, NB = 32;
    const int64_t mblocks = (M + MB - 1) / MB;
    const int nblocks = (Nn + NB - 1) / NB;
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t mb = 0; mb < mblocks; ++mb)
        for (int nbk = 0; nbk < nblocks; ++nbk) {
            const int64_t m0 = mb * MB, m1 = (m0 + MB < M) ? m0 + MB : M;
            const int n0 = nbk * NB, n1 = (n0 + NB < Nn) ? n0 + NB : Nn;
            for (int n = n0; n < n1; ++n) {
                const float *bp = B + (size_t)n * K;
                const float bv = bias ? bias[n] : 0.0f;
                for (int64_t m = m0; m < m1; ++m)
                    C[(size_t)m * Nn + n] = dot8(A + (size_t)m * K, bp, K) + bv;
            }
        }
}

/*
 * One torch.nn.LSTM(size, insize) layer, seq-first, zero initial state, optional reverse
 * (nn.py:176-193: flip -> run -> flip).  Gate order i,f,g,o (PyTorch).
 * x: (T, N, I)  w_ih: (4H, I)  w_hh: (4H, H)  b_ih,b_hh: (4H)  y: (T, N, H)
 */
XO_API int xo_lstm(const float *x, int T, int N, int I, int H, const float *w_ih,
                   const float *w_hh, const float *b_ih, const float *b_hh, int reverse, float *y)
{
    const size_t G = (size_t)4 * H;
    float *gin = (float *)malloc(sizeof(float) * (size_t)T * N * G);
    float *g = (float *)malloc(sizeof(float) * (size_t)N * G);
    float *h = (float *)calloc((size_t)N * H, sizeof(float));
    float *c = (float *)calloc((size_t)N * H, sizeof(float));
    float *bsum = (float *)malloc(sizeof(float) * G);
    if (!gin || !g || !h || !c || !bsum) { free(gin); free(g); free(h); free(c); free(bsum); return -1; }
    for (size_t i = 0; i < G; ++i) bsum[i] = (b_ih ? b_ih[i] : 0.0f) + (b_hh ? b_hh[i] : 0.0f);
    gemm_nt_bias(x, w_ih, bsum, gin, (int64_t)T * N, (int)G, I);
    for (int s = 0; s < T; ++s) {
        const int t = reverse ? T - 1 - s : s;
        gemm_nt_bias(h, w_hh, NULL, g, N, (int)G, H);
        const float *gi = gin + (size_t)t * N * G;
#pragma omp parallel for schedule(static)
        for (int n = 0; n < N; ++n)
            for (int u = 0; u < H; ++u) {
                const size_t o = (size_t)n * G;
                const float ig = sigmoidf_(g[o + u] + gi[o + u]);
                const float fg = sigmoidf_(g[o + H + u] + gi[o + H + u]);
                const float gg = tanhf(g[o + 2 * H + u] + gi[o + 2 * H + u]);
                const float og = sigmoidf_(g[o + 3 * H + u] + gi[o + 3 * H + u]);
                const float cn = fg * c[(size_t)n * H + u] + ig * gg;
                c[(size_t)n * H + u] = cn;
                const float hn = og * tanhf(cn);
                h[(size_t)n * H + u] = hn;
                y[((size_t)t * N + n) * H + u] = hn;
            }
    }
    free(gin); free(g); free(h); free(c); free(bsum);
    return 0;
}

/*
 * LinearCRFEncoder.forward (nn.py:112-133): scale * tanh(W x + b), then (expand != 0) a column
 * of `blank` is inserted in front of every group of nb outputs.
 * x: (M, I)  w: (O, I)  b: (O)   y: (M, O) or (M, O/nb*(nb+1))
 */
XO_API int xo_linear_crf(const float *x, int64_t M, int I, const float *w, const float *b, int O,
                         float scale, int nb, int expand, float blank, float *y)
{
    float *tmp = (float *)malloc(sizeof(float) * (size_t)M * O);
    if (!tmp) return -1;
    gemm_nt_bias(x, w, b, tmp, M, O, I);
    const int E = nb + 1, S = O / nb;
#pragma omp parallel for schedule(static)
    for (int64_t m = 0; m < M; ++m) {
        const float *ti = tmp + (size_t)m * O;
        if (expand) {
            float *yo = y + (size_t)m * S * E;
            for (int s = 0; s < S; ++s) {
                yo[s * E] = blank;
                for (int k = 0; k < nb; ++k) yo[s * E + 1 + k] = scale * tanhf(ti[s * nb + k]);
            }
        } else {
            float *yo = y + (size_t)m * O;
            for (int o = 0; o < O; ++o) yo[o] = scale * tanhf(ti[o]);
        }
    }
    free(tmp);
    return 0;
}

/*
 * Whole encoder, rnn_encoder() of crf/model.py:142-160 (inference form, no dropout):
 * conv(1->4,k5,p2) conv(4->16,k5,p2) conv(16->F,k=winlen,s=stride,p=winlen/2) each +SiLU,
 * permute to (T,N,F), 5 LSTMs reverse,fwd,reverse,fwd,reverse, LinearCRFEncoder.
 * weights: array of 28 pointers in state-dict order
 *   conv{0,1,2}.{weight,bias}, lstm{0..4}.{weight_ih,weight_hh,bias_ih,bias_hh}, linear.{weight,bias}
 * signal: (N, L)   scores: (T, N, S*E) if expand else (T, N, S*nb)
 */
XO_API int xo_encode(const float *signal, int N, int L, const float *const *weights, int features,
                     int winlen, int stride, int nb, int sl, float scale, float blank, int expand,
                     float *scores, float *lstm_out /* optional (T,N,F): input of the linear */)
{
    const int F = features;
    const int T = (L + 2 * (winlen / 2) - winlen) / stride + 1;
    float *c1 = (float *)malloc(sizeof(float) * (size_t)N * 4 * L);
    float *c2 = (float *)malloc(sizeof(float) * (size_t)N * 16 * L);
    float *c3 = (float *)malloc(sizeof(float) * (size_t)N * F * T);
    float *xa = (float *)malloc(sizeof(float) * (size_t)T * N * F);
    float *xb = (float *)malloc(sizeof(float) * (size_t)T * N * F);
    int rc = -1;
    if (!c1 || !c2 || !c3 || !xa || !xb) goto done;
    xo_conv1d_silu(signal, N, 1, L, weights[0], weights[1], 4, 5, 1, 2, c1);
    xo_conv1d_silu(c1, N, 4, L, weights[2], weights[3], 16, 5, 1, 2, c2);
    xo_conv1d_silu(c2, N, 16, L, weights[4], weights[5], F, winlen, stride, winlen / 2, c3);
    /* Permute([2,0,1]): (N,F,T) -> (T,N,F) */
#pragma omp parallel for collapse(2) schedule(static)
    for (int t = 0; t < T; ++t)
        for (int n = 0; n < N; ++n)
            for (int f = 0; f < F; ++f)
                xa[((size_t)t * N + n) * F + f] = c3[((size_t)n * F + f) * T + t];
    {
        float *src = xa, *dst = xb;
        for (int l = 0; l < 5; ++l) {
            const float *const *w = weights + 6 + 4 * l;
            if (xo_lstm(src, T, N, F, F, w[0], w[1], w[2], w[3], (l % 2) == 0, dst)) goto done;
            float *tmp = src; src = dst; dst = tmp;
        }
        if (lstm_out) memcpy(lstm_out, src, sizeof(float) * (size_t)T * N * F);
        const int O = (int)ipow(nb, sl + 1);
        if (xo_linear_crf(src, (int64_t)T * N, F, weights[26], weights[27], O, scale, nb, expand,
                          blank, scores)) goto done;
    }
    rc = 0;
done:
    free(c1); free(c2); free(c3); free(xa); free(xb);
    return rc;
}

/* ------------------------------------------------------------------------------------ */
/* CTC-CRF loss scans (crf/model.py:102-135)                                             */
/* ------------------------------------------------------------------------------------ */
/*
 * prepare_ctc_scores (crf/model.py:102-116): targets (N, Lt) are CTC labels (0 = blank / padding), shifted to zero-based
 * bases and clamped; with n = Lt - (sl - 1) target positions
 *   stay_idx[b][l] = (sum_i targets[b][l + i] * nb^(sl - i - 1)) * E          l = 0 .. n-1
 *   move_idx[b][l] = stay_idx[b][l + 1] + targets[b][l] + 1                   l = 0 .. n-2
 * index the C = S * E score columns of a time step.
 */
XO_API void xo_ctc_indices(const int32_t *targets, int N, int Lt, int nb, int sl, int32_t *stay_idx, int32_t *move_idx)
{
    const int n = Lt - (sl - 1), E = nb + 1;
    for (int b = 0; b < N; ++b) {
        for (int l = 0; l < n; ++l) {
            int64_t st = 0;
            for (int i = 0; i < sl; ++i) {
                int v = targets[(size_t)b * Lt + l + i] - 1;
                if (v < 0) v = 0;
                st += (int64_t)v * ipow(nb, sl - i - 1);
            }
            stay_idx[(size_t)b * n + l] = (int32_t)(st * E);
        }
        for (int l = 0; l + 1 < n; ++l) {
            int v = targets[(size_t)b * Lt + l] - 1;
            if (v < 0) v = 0;
            move_idx[(size_t)b * (n - 1) + l] = stay_idx[(size_t)b * n + l + 1] + v + 1;
        }
    }
}

/*
 * seqdist.ctc_simple.logZ (logZ_cupy at crf/model.py:122; viterbi_alignments at :135) restated -- "parity unpinned" like the
 * decode: seqdist 0.0.4 is not in the tree.  With stay[t][l] = scores[t][b][stay_idx[b][l]], move[t][l] = scores[t][b][move_idx[b][l]]:
 *   alpha_0[l] = (l == 0 ? one : zero);   alpha_{t+1}[l] = sum2(alpha_t[l] + stay[t][l], alpha_t[l-1] + move[t][l-1])
 *   logZ[b] = alpha_T[len_b - 1]          (beta_T is `one` at position len_b - 1 and `zero` elsewhere)
 *   beta_t[l] = sum2(stay[t][l] + beta_{t+1}[l], move[t][l] + beta_{t+1}[l+1])
 * where len_b = target_lengths[b] + 1 - sl positions are in use, zero = -1e38 (seqdist's Log.zero), one = 0, and
 *   semiring 0 (Log): sum2(x0, x1) = m + log(exp(x0 - m) + exp(x1 - m)), m = max(x0, x1), with the contract's exp / log;
 *                     a missing neighbour (l - 1 < 0, l + 1 >= n) enters as `zero`, which changes nothing in binary32;
 *   semiring 1 (Max): sum2 = max.
 * Optional outputs (Log): the restricted posteriors = d logZ / d stay, d logZ / d move,
 *   gstay[t][b][l] = exp(((alpha_t[l] + stay[t][l]) + beta_{t+1}[l]) - logZ),  gmove[t][b][l] = exp(((alpha_t[l] + move[t][l]) + beta_{t+1}[l+1]) - logZ);
 * (Max): the Viterbi alignment, one-hot over positions: align[t][b][l] = 1 where the best path sits at position l before
 *   step t (stay.grad + shifted move.grad of seqdist's viterbi_alignments); ties prefer the stay edge.
 * scores (T, N, C) fp32; stay_idx (N, n), move_idx (N, n-1) from xo_ctc_indices; gstay / align (T, N, n), gmove (T, N, n-1).
 */
XO_API int xo_ctc_logz(const float *scores, int T, int N, int C, const int32_t *stay_idx, const int32_t *move_idx, int n,
                       const int32_t *target_lengths, int sl, int semiring, float *logz, float *gstay, float *gmove)
{
    const float ZERO = -1e38f;
    int rc = 0;
#pragma omp parallel for schedule(dynamic)
    for (int b = 0; b < N; ++b) {
        const int len = target_lengths[b] + 1 - sl;
        if (len < 1 || len > n) { rc = -1; continue; }
        float *alpha = (float *)malloc(sizeof(float) * (size_t)(T + 1) * n);
        float *beta = (float *)malloc(sizeof(float) * (size_t)2 * n);
        if (!alpha || !beta) { rc = -2; free(alpha); free(beta); continue; }
        const int32_t *si = stay_idx + (size_t)b * n, *mi = move_idx + (size_t)b * (n - 1);
        for (int l = 0; l < n; ++l) alpha[l] = l == 0 ? 0.0f : ZERO;
        for (int t = 0; t < T; ++t) {
            const float *row = scores + ((size_t)t * N + b) * C;
            const float *a = alpha + (size_t)t * n;
            float *an = alpha + (size_t)(t + 1) * n;
            for (int l = 0; l < n; ++l) {
                const float x0 = a[l] + row[si[l]];
                const float x1 = l > 0 ? a[l - 1] + row[mi[l - 1]] : ZERO;
                const float m = x0 > x1 ? x0 : x1;
                an[l] = semiring ? m : m + xo_logf_i(xo_expf_i(x0 - m) + xo_expf_i(x1 - m));
            }
        }
        const float lz = alpha[(size_t)T * n + len - 1];
        if (logz) logz[b] = lz;
        if (semiring == 0 && (gstay || gmove)) {
            float *bn = beta, *bc = beta + n;
            for (int l = 0; l < n; ++l) bn[l] = l == len - 1 ? 0.0f : ZERO;
            for (int t = T - 1; t >= 0; --t) {
                const float *row = scores + ((size_t)t * N + b) * C;
                const float *a = alpha + (size_t)t * n;
                for (int l = 0; l < n; ++l) {
                    const float st = row[si[l]];
                    const float mv = l + 1 < n ? row[mi[l]] : 0.0f;
                    const float bnext = l + 1 < n ? bn[l + 1] : ZERO;
                    if (gstay) gstay[((size_t)t * N + b) * n + l] = xo_expf_i(((a[l] + st) + bn[l]) - lz);
                    if (gmove && l + 1 < n) gmove[((size_t)t * N + b) * (n - 1) + l] = xo_expf_i(((a[l] + mv) + bnext) - lz);
                    const float x0 = st + bn[l];
                    const float x1 = l + 1 < n ? mv + bnext : ZERO;
                    const float m = x0 > x1 ? x0 : x1;
                    bc[l] = m + xo_logf_i(xo_expf_i(x0 - m) + xo_expf_i(x1 - m));
                }
                float *tmp = bn; bn = bc; bc = tmp;
            }
        }
        if (semiring == 1 && gstay) {
            /* back-trace of the max path from position len - 1 at time T */
            int l = len - 1;
            for (int t = T - 1; t >= 0; --t) {
                const float *row = scores + ((size_t)t * N + b) * C;
                const float *a = alpha + (size_t)t * n;
                float *out = gstay + ((size_t)t * N + b) * n;
                for (int k = 0; k < n; ++k) out[k] = 0.0f;
                const float x0 = a[l] + row[si[l]];
                const float x1 = l > 0 ? a[l - 1] + row[mi[l - 1]] : ZERO;
                if (!(x0 >= x1)) l -= 1;          /* the move edge l-1 -> l was strictly better */
                out[l] = 1.0f;
            }
        }
        free(alpha);
        free(beta);
    }
    return rc;
}

/* ------------------------------------------------------------------------------------ */
/* CRF beam search with qualities and moves (the non-Viterbi branch of compute_scores,   */
/* crf/basecall.py:33-46: koi.decode.beam_search(scores, beam_width=32, beam_cut=100,     */
/* scale, offset, blank_score=2.0) -> sequence, qstring, moves)                           */
/* ------------------------------------------------------------------------------------ */
/*
 * koi 0.0.5 is not in /root/reference (a pip dependency, CUDA only) and the reference holds no vectors for it: PARITY
 * UNPINNED.  What is restated here is the algorithm ONT publishes for this decoder (the open CPU decoder of its production
 * basecaller, decode/beam_search.cpp, which mirrors koi's kernels), generalised from 4 bases / power-of-two k-mers to the
 * n_base-ary state table of crf/model.py:31-36 and put on this file's arithmetic contract (xo_expf_i / xo_logf_i, no
 * contraction), so that the HIP kernel can be checked bit for bit:
 *
 *   back guide   beta (T+1, N, S): the Log-semiring backward scores of the contract above (xo_decode), stay score = the
 *                blank column (or the constant `blank`).
 *   beam element (hash, state, prev element, stay flag) + score (log-sum over merged paths, WITHOUT the back guide).
 *   start        every state whose beta_0 is among the beam_width best ((beam_width+1)-th largest value as threshold, states
 *                in ascending order, at most beam_width), hash = crc32c(seed, state), score 0.
 *   block t      candidates in this order: for every element p (beam order) and base b the step into
 *                j = (state % nb^(sl-1)) * nb + b with score (s_p + M[t, j, 1 + state / nb^(sl-1)]) + beta_{t+1}[j] and hash
 *                crc32c(hash_p, j); then for every element p the stay, (s_p + M[t, state, 0]) + beta_{t+1}[state], same hash.
 *                A stay whose hash equals that of a step into the same latest base (the same base sequence reached by
 *                a move) is merged with it: the better of the two keeps lse2(a, b), the other gets -FLT_MAX.
 *                Kept: the first beam_width candidates, in candidate order, whose score reaches max - log(beam_cut); if more
 *                than beam_width reach it the threshold is bisected between it and the maximum (at most 10 guesses, a
 *                count in [0.8 beam_width, beam_width] stops the search, after 10 guesses the upper limit is taken).
 *                The back guide is subtracted again from the kept scores.  After the last block the best element
 *                (first maximum) is element 0.
 *   trace back   from element 0 of the last block: state and stay flag per block; moves[0] = 1.
 *   qualities    per block the posterior mass of the path k-mer and of its nb left- and nb right-shifted neighbours at t+1,
 *                P_t(s) = exp((alpha_t[s] + beta_t[s]) - logZ), clamped to [0, 1], raised to 0.4; a base's probability is the
 *                mean over the blocks it spans (sum of p over sum of (p + (nb-1) * (1-p)/(nb-1))); q = -10 log10(1 - prob)
 *                * qscale + qoffset, clamped to [1, 50], character (int)(33.5 + q).
 *   outputs      sequence (N, T) int8: alphabet[1 + base] at the blocks that emit (moves = 1), else 0; qstring (N, T) int8 the
 *                quality character at the same blocks; moves (N, T) uint8; score (N) the path score of element 0 [optional].
 */
#define XO_BEAM_MAX 32
#define XO_CRC_SEED 0x12345678u
#define XO_HASH_BITS 4096

static inline uint32_t xo_crc32c_u32(uint32_t crc, uint32_t v)
{
    crc ^= v;
    for (int i = 0; i < 32; ++i) crc = (crc >> 1) ^ (0x82F63B78u & (0u - (crc & 1u)));
    return crc;
}

static inline float xo_lse2(float x, float y)
{
    const float d = fabsf(x - y);
    const float m = x > y ? x : y;
    return d < 17.0f ? m + xo_logf_i(1.0f + xo_expf_i(-d)) : m;
}

static int xo_cmp_desc(const void *a, const void *b)
{
    const float x = *(const float *)a, y = *(const float *)b;
    return x > y ? -1 : (x < y ? 1 : 0);
}

XO_API int xo_beam_search(const float *scores, int T, int N, int nb, int sl, int has_blank, float blank,
                          const float *alpha, const float *beta, const float *logz,
                          int beam_width, float log_beam_cut, float qscale, float qoffset, const char *alphabet,
                          int8_t *sequence, int8_t *qstring, uint8_t *moves, float *score_out)
{
    const int S = (int)ipow(nb, sl), E = nb + 1;
    const int Cin = has_blank ? S * E : S * nb;
    const int hi = (int)ipow(nb, sl - 1);
    const int W = beam_width;
    if (W < 1 || W > XO_BEAM_MAX || T < 1) return -3;
    int err = 0;

#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; ++n) {
        uint32_t fh[XO_BEAM_MAX], ch[XO_BEAM_MAX * 8];
        int fs[XO_BEAM_MAX], cs[XO_BEAM_MAX * 8], cp[XO_BEAM_MAX * 8];
        uint8_t cstay[XO_BEAM_MAX * 8], present[XO_HASH_BITS];
        float fsc[XO_BEAM_MAX], csc[XO_BEAM_MAX * 8];
        /* history: state, prev, stay per (block, element) */
        int32_t *hst = (int32_t *)malloc(sizeof(int32_t) * (size_t)(T + 1) * W);
        uint8_t *hprev = (uint8_t *)malloc((size_t)(T + 1) * W), *hstay = (uint8_t *)malloc((size_t)(T + 1) * W);
        int32_t *path = (int32_t *)malloc(sizeof(int32_t) * (size_t)T);
        float *prob = (float *)malloc(sizeof(float) * (size_t)T);
        float *sorted = (float *)malloc(sizeof(float) * (size_t)S);
        if (!hst || !hprev || !hstay || !path || !prob || !sorted) {
            err = -1;
            free(hst); free(hprev); free(hstay); free(path); free(prob); free(sorted);
            continue;
        }
#define BETA(t, s) beta[((size_t)(t) * N + n) * S + (s)]
#define ALPHA(t, s) alpha[((size_t)(t) * N + n) * S + (s)]
        /* ---- start ---- */
        float thr = -3.402823466e+38f;
        if (W < S) {
            for (int s_ = 0; s_ < S; ++s_) sorted[s_] = BETA(0, s_);
            qsort(sorted, (size_t)S, sizeof(float), xo_cmp_desc);
            thr = sorted[W];
        }
        int Wc = 0;
        for (int s_ = 0; s_ < S && Wc < W; ++s_)
            if (BETA(0, s_) >= thr) {
                fh[Wc] = xo_crc32c_u32(XO_CRC_SEED, (uint32_t)s_);
                fs[Wc] = s_;
                fsc[Wc] = 0.0f;
                hst[Wc] = s_; hprev[Wc] = 0; hstay[Wc] = 0;
                ++Wc;
            }
        /* ---- blocks ---- */
        for (int t = 0; t < T; ++t) {
            const float *row = scores + ((size_t)t * N + n) * Cin;
            float max_score = -3.402823466e+38f;
            memset(present, 0, sizeof(present));
            int nc = 0;
            for (int p = 0; p < Wc; ++p)
                for (int b = 0; b < nb; ++b) {
                    const int j = (fs[p] % hi) * nb + b, k = fs[p] / hi;
                    const float m = has_blank ? row[(size_t)j * E + 1 + k] : row[(size_t)j * nb + k];
                    const float v = (fsc[p] + m) + BETA(t + 1, j);
                    const uint32_t h = xo_crc32c_u32(fh[p], (uint32_t)j);
                    present[h % XO_HASH_BITS] = 1;
                    ch[nc] = h; cs[nc] = j; cp[nc] = p; cstay[nc] = 0; csc[nc] = v;
                    max_score = v > max_score ? v : max_score;
                    ++nc;
                }
            for (int p = 0; p < Wc; ++p) {
                const int st = fs[p];
                const float m = has_blank ? row[(size_t)st * E] : blank;
                const float v = (fsc[p] + m) + BETA(t + 1, st);
                const int si = nc;
                ch[si] = fh[p]; cs[si] = st; cp[si] = p; cstay[si] = 1; csc[si] = v;
                max_score = v > max_score ? v : max_score;
                if (present[fh[p] % XO_HASH_BITS]) {
                    const int latest = st % nb;
                    for (int q = 0; q < Wc; ++q) {
                        const int ti = q * nb + latest;
                        if (ch[si] == ch[ti]) {
                            const float f = xo_lse2(csc[si], csc[ti]);
                            if (csc[si] > csc[ti]) { csc[si] = f; csc[ti] = -3.402823466e+38f; }
                            else { csc[ti] = f; csc[si] = -3.402823466e+38f; }
                            max_score = f > max_score ? f : max_score;
                        }
                    }
                }
                ++nc;
            }
            float cutoff = max_score - log_beam_cut;
            int count = 0;
#define XO_COUNT() do { count = 0; for (int c_ = 0; c_ < nc; ++c_) count += csc[c_] >= cutoff; } while (0)
            XO_COUNT();
            if (count > W) {
                const int minw = (W * 8) / 10;
                float lo = cutoff, hi_s = max_score;
                int guesses = 1;
                while ((count > W || count < minw) && guesses < 10) {
                    if (count > W) { lo = cutoff; cutoff = (cutoff + hi_s) / 2.0f; }
                    else { hi_s = cutoff; cutoff = (cutoff + lo) / 2.0f; }
                    XO_COUNT();
                    ++guesses;
                }
                if (guesses == 10) { cutoff = hi_s; XO_COUNT(); }
                count = count < W ? count : W;
            }
#undef XO_COUNT
            int w = 0;
            for (int c = 0; c < nc && w < W; ++c)
                if (csc[c] >= cutoff) {
                    fh[w] = ch[c]; fs[w] = cs[c]; fsc[w] = csc[c];
                    hst[(size_t)(t + 1) * W + w] = cs[c]; hprev[(size_t)(t + 1) * W + w] = (uint8_t)cp[c];
                    hstay[(size_t)(t + 1) * W + w] = cstay[c];
                    ++w;
                }
            Wc = w;
            if (t == T - 1) {       /* the best element becomes element 0 (first maximum) */
                int best = 0;
                for (int i = 1; i < Wc; ++i) if (fsc[i] > fsc[best]) best = i;
                if (best != 0) {
                    const size_t o = (size_t)T * W;
                    uint32_t th = fh[0]; fh[0] = fh[best]; fh[best] = th;
                    int ts = fs[0]; fs[0] = fs[best]; fs[best] = ts;
                    float tf = fsc[0]; fsc[0] = fsc[best]; fsc[best] = tf;
                    int32_t a = hst[o]; hst[o] = hst[o + best]; hst[o + best] = a;
                    uint8_t u = hprev[o]; hprev[o] = hprev[o + best]; hprev[o + best] = u;
                    u = hstay[o]; hstay[o] = hstay[o + best]; hstay[o + best] = u;
                }
            }
            for (int i = 0; i < Wc; ++i) fsc[i] -= BETA(t + 1, fs[i]);
        }
        if (score_out) score_out[n] = fsc[0];
        /* ---- trace back ---- */
        uint8_t *mv = moves + (size_t)n * T;
        int el = 0;
        for (int t = T; t >= 1; --t) {
            const size_t a = (size_t)t * W + el;
            path[t - 1] = hst[a];
            mv[t - 1] = hstay[a] ? 0 : 1;
            el = hprev[a];
        }
        mv[0] = 1;
        /* ---- per-block probability of the path k-mer ---- */
        const float lz = logz[n];
        for (int t = 0; t < T; ++t) {
            const int st = path[t];
#define XO_POST(s) xo_expf_i((ALPHA(t + 1, (s)) + BETA(t + 1, (s))) - lz)
            float p = XO_POST(st);
            const int l0 = st / nb, r0 = (st % hi) * nb;
            for (int b = 0; b < nb; ++b) {
                p += XO_POST(l0 + hi * b);
                p += XO_POST(r0 + b);
            }
#undef XO_POST
            p = p > 1.0f ? 1.0f : p;
            p = p < 0.0f ? 0.0f : p;
            prob[t] = p > 0.0f ? xo_expf_i(0.4f * xo_logf_i(p)) : 0.0f;
        }
        /* ---- sequence and quality string at the emitting blocks ---- */
        int8_t *sq = sequence + (size_t)n * T, *qs = qstring + (size_t)n * T;
        for (int t = 0; t < T; ++t) { sq[t] = 0; qs[t] = 0; }
        for (int t = 0; t < T; ++t) {
            if (!mv[t]) continue;
            float bp = 0.0f, tot = 0.0f;
            for (int u = t; u < T && (u == t || !mv[u]); ++u) {
                const float p = prob[u], wrong = (1.0f - p) / (float)(nb - 1);
                bp += p;
                float one = p;
                for (int j = 1; j < nb; ++j) one += wrong;
                tot += one;
            }
            const float e = 1.0f - bp / tot;
            float q = e > 0.0f ? xo_logf_i(e) * -4.3429448190325175f : 3.402823466e+38f;
            q = q * qscale;
            q = q + qoffset;
            q = q < 1.0f ? 1.0f : q;
            q = q > 50.0f ? 50.0f : q;
            sq[t] = (int8_t)alphabet[1 + path[t] % nb];
            qs[t] = (int8_t)(int)(33.5f + q);
        }
#undef BETA
#undef ALPHA
        free(hst); free(hprev); free(hstay); free(path); free(prob); free(sorted);
    }
    return err;
}

XO_API void xo_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

XO_API int xo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

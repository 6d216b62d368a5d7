// CRF beam search with qualities and moves: the non-Viterbi branch of compute_scores (crf/basecall.py:33-46,
// koi.decode.beam_search(scores, beam_width=32, beam_cut=100.0, scale, offset, blank_score=2.0) -> sequence, qstring, moves).
//
// koi 0.0.5 is absent from the reference tree; the algorithm is the one ONT publishes for this decoder (back guide, CRC-32C
// sequence hashes, stay / step merging, beam cut by bisection, k-mer posterior qualities), generalised to the n_base-ary state
// table of crf/model.py:31-36.  Its specification is the header of the same section in oracle/xna_oracle.c; this kernel is
// written independently against that specification and is checked bit for bit (tests/test_gpu_beam.py).  PARITY UNPINNED.
//
// MI355X mapping: a beam of 32 elements and its 32 * (n_base + 1) candidates are less than one wave's worth of work and the
// blocks of a chunk are strictly sequential, so a chunk is ONE wave (a 64-thread workgroup): no workgroup barriers on the
// critical path, wave ballots for the counts and the order-preserving compaction, the beam front and the candidate list in
// LDS.  512 chunks = 512 waves spread over all 256 CUs; the history (state, previous element, stay flag per block and element)
// goes to HBM and is read back tile by tile for the trace-back.  Scores and back guide are gathered per candidate (L2).
// Floating-point contract as in xb_decode.hip (built with -ffp-contract=off).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "xb_internal.h"
#include "xb_math.h"

namespace {

constexpr int BW = xb::BEAM_MAX_WIDTH;      // 32
constexpr int BCAND = 256;                  // 4 candidates per lane >= 32 * (n_base + 1), n_base <= 7
constexpr uint32_t CRC_SEED = 0x12345678u;
constexpr float NEG_MAX = -3.402823466e+38f;

__device__ __forceinline__ unsigned long long ballot(bool v) { return __builtin_amdgcn_ballot_w64(v); }
__device__ __forceinline__ int lanes_below(unsigned long long m, int lane)
{
    return __builtin_popcountll(m & ((1ull << lane) - 1ull));
}
__device__ __forceinline__ float wave_maxf(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        const float w = __shfl_xor(v, o, 64);
        v = w > v ? w : v;
    }
    return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ void wave_sync() { __syncthreads(); }     // one wave per workgroup: orders the LDS traffic

__device__ __forceinline__ float lse2(float x, float y)
{
    const float d = __builtin_fabsf(x - y);
    const float m = x > y ? x : y;
    return d < 17.0f ? m + xb_logf(1.0f + xb_expf(-d)) : m;
}

__global__ __launch_bounds__(64) void beam_kernel(xb::BeamParams p)
{
    __shared__ uint32_t crc_tab[256];
    __shared__ uint32_t f_hash[BW], f_info[BW];
    __shared__ int f_state[BW];
    __shared__ float f_score[BW];
    __shared__ uint32_t c_hash[BCAND], c_info[BCAND];
    __shared__ float c_score[BCAND];
    __shared__ int claim[BCAND];
    __shared__ uint32_t tile[64 * BW];
    __shared__ uint32_t keys[xb::BEAM_MAX_STATES];
    __shared__ int t_state[64];
    __shared__ uint8_t t_move[64];

    const int lane = threadIdx.x, n = blockIdx.x;
    const int T = p.T, N = p.N, S = p.S, nb = p.nb, hi = p.hi, W = p.W, E = nb + 1;
    const size_t sstride = (size_t)N * S;
    const float *beta = p.beta + (size_t)n * S, *alpha = p.alpha + (size_t)n * S;
    uint32_t *hist = p.hist + (size_t)n * (T + 1) * BW;

    for (int i = lane; i < 256; i += 64) {
        uint32_t c = (uint32_t)i;
#pragma unroll
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0x82F63B78u & (0u - (c & 1u)));
        crc_tab[i] = c;
    }
    wave_sync();
    auto crc32c = [&](uint32_t crc, uint32_t v) {
        crc ^= v;
#pragma unroll
        for (int k = 0; k < 4; ++k) crc = crc_tab[crc & 0xffu] ^ (crc >> 8);
        return crc;
    };

    // ---- start: the states whose back guide is among the W best
    float thr = NEG_MAX;
    if (W < S) {
        for (int s = lane; s < S; s += 64) {
            const uint32_t u = f2bits(beta[s]);
            keys[s] = (u & 0x80000000u) ? ~u : (u | 0x80000000u);      // unsigned order == float order
        }
        wave_sync();
        uint32_t v = 0;     // the largest key with at least W + 1 keys >= it = the (W + 1)-th largest key
        for (int bit = 31; bit >= 0; --bit) {
            const uint32_t tryv = v | (1u << bit);
            int cnt = 0;
            for (int s = lane; s < S; s += 64) cnt += keys[s] >= tryv ? 1 : 0;
            cnt = wave_sum(cnt);
            if (cnt >= W + 1) v = tryv;
        }
        thr = bits2f((v & 0x80000000u) ? (v & 0x7fffffffu) : ~v);
    }
    int Wc = 0;
    for (int s0 = 0; s0 < S && Wc < W; s0 += 64) {
        const int s = s0 + lane;
        const bool take = s < S && beta[s] >= thr;
        const unsigned long long m = ballot(take);
        const int pos = Wc + lanes_below(m, lane);
        if (take && pos < W) {
            f_hash[pos] = crc32c(CRC_SEED, (uint32_t)s);
            f_state[pos] = s;
            f_score[pos] = 0.0f;
            hist[pos] = (uint32_t)s;
        }
        Wc += __builtin_popcountll(m);
    }
    Wc = Wc < W ? Wc : W;
    wave_sync();

    // ---- blocks
    float sc[4];
    for (int t = 0; t < T; ++t) {
        const float *row = p.scores + ((size_t)t * N + n) * p.ld;
        const float *b1 = beta + (size_t)(t + 1) * sstride;
        const int nstep = Wc * nb, nc = nstep + Wc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q * 64 + lane;
            sc[q] = NEG_MAX;
            if (c < nstep) {
                const int pi = c / nb, b = c - pi * nb;
                const int st = f_state[pi];
                const int k = st / hi, j = (st - k * hi) * nb + b;
                const float m = p.has_blank ? row[(size_t)j * E + 1 + k] : row[(size_t)j * nb + k];
                sc[q] = (f_score[pi] + m) + b1[j];
                c_hash[c] = crc32c(f_hash[pi], (uint32_t)j);
                c_info[c] = (uint32_t)j | ((uint32_t)pi << 16);
                c_score[c] = sc[q];
                claim[c] = -1;
            } else if (c < nc) {
                const int pi = c - nstep;
                const int st = f_state[pi];
                const float m = p.has_blank ? row[(size_t)st * E] : p.blank;
                sc[q] = (f_score[pi] + m) + b1[st];
                c_hash[c] = f_hash[pi];
                c_info[c] = (uint32_t)st | ((uint32_t)pi << 16) | (1u << 24);
                c_score[c] = sc[q];
            }
        }
        wave_sync();
        // ---- a stay and a step that spell the same sequence are one path: merge them
        {
            uint32_t match = 0;
            int latest = 0;
            if (lane < Wc) {
                latest = f_state[lane] % nb;
                const uint32_t h = f_hash[lane];
                for (int j = 0; j < Wc; ++j) match |= (c_hash[j * nb + latest] == h ? 1u : 0u) << j;
            }
            const unsigned long long any = ballot(match != 0);
            if (any) {
                const int si = nstep + lane;
                const unsigned long long multi = ballot(__builtin_popcount(match) > 1);
                const int tgt = match ? (__builtin_ctz(match) * nb + latest) : 0;
                if (match) claim[tgt] = lane;
                wave_sync();
                const unsigned long long clash = ballot(match != 0 && claim[tgt] != lane);
                if (!multi && !clash) {            // every merge touches its own two candidates: all at once
                    if (match) {
                        const float a = c_score[si], b = c_score[tgt];
                        const float f = lse2(a, b);
                        c_score[si] = a > b ? f : NEG_MAX;
                        c_score[tgt] = a > b ? NEG_MAX : f;
                    }
                } else {                            // hash collisions: the specification's order, one merge at a time
                    unsigned long long rest = any;
                    while (rest) {
                        const int i = __builtin_ctzll(rest);
                        rest &= rest - 1;
                        if (lane == i) {
                            uint32_t mm = match;
                            while (mm) {
                                const int j = __builtin_ctz(mm);
                                mm &= mm - 1;
                                const int ti = j * nb + latest;
                                const float a = c_score[si], b = c_score[ti];
                                const float f = lse2(a, b);
                                c_score[si] = a > b ? f : NEG_MAX;
                                c_score[ti] = a > b ? NEG_MAX : f;
                            }
                        }
                        wave_sync();
                    }
                }
                wave_sync();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = q * 64 + lane;
                    if (c < nc) sc[q] = c_score[c];
                }
            }
        }
        // ---- the cut
        float mx = sc[0];
        mx = sc[1] > mx ? sc[1] : mx;
        mx = sc[2] > mx ? sc[2] : mx;
        mx = sc[3] > mx ? sc[3] : mx;
        const float max_score = wave_maxf(mx);
        float cutoff = max_score - p.log_cut;
        auto count_kept = [&](float cut) {
            int cnt = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) cnt += __builtin_popcountll(ballot(q * 64 + lane < nc && sc[q] >= cut));
            return cnt;
        };
        int count = count_kept(cutoff);
        if (count > W) {
            const int minw = (W * 8) / 10;
            float lo = cutoff, hi_s = max_score;
            int guesses = 1;
            while ((count > W || count < minw) && guesses < 10) {
                if (count > W) { lo = cutoff; cutoff = (cutoff + hi_s) / 2.0f; }
                else { hi_s = cutoff; cutoff = (cutoff + lo) / 2.0f; }
                count = count_kept(cutoff);
                ++guesses;
            }
            if (guesses == 10) cutoff = hi_s;
        }
        // ---- the first W candidates that reach the cut, in candidate order
        int base = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = q * 64 + lane;
            const bool keep = c < nc && sc[q] >= cutoff;
            const unsigned long long m = ballot(keep);
            const int pos = base + lanes_below(m, lane);
            if (keep && pos < W) {
                const uint32_t info = c_info[c];
                f_hash[pos] = c_hash[c];
                f_info[pos] = info;
                f_state[pos] = (int)(info & 0xffffu);
                f_score[pos] = sc[q];
            }
            base += __builtin_popcountll(m);
        }
        Wc = base < W ? base : W;
        wave_sync();
        if (t == T - 1) {       // the best element (first maximum) becomes element 0
            const float v = lane < Wc ? f_score[lane] : NEG_MAX;
            const float best_v = wave_maxf(v);
            const int best = __builtin_ctzll(ballot(lane < Wc && v == best_v));
            if (best != 0 && lane == 0) {
                const uint32_t h = f_hash[0], inf = f_info[0];
                const int s0 = f_state[0];
                const float v0 = f_score[0];
                f_hash[0] = f_hash[best]; f_info[0] = f_info[best]; f_state[0] = f_state[best]; f_score[0] = f_score[best];
                f_hash[best] = h; f_info[best] = inf; f_state[best] = s0; f_score[best] = v0;
            }
            wave_sync();
        }
        if (lane < Wc) {
            f_score[lane] = f_score[lane] - b1[f_state[lane]];
            hist[(size_t)(t + 1) * BW + lane] = f_info[lane];
        }
        wave_sync();
    }
    if (p.score && lane == 0) p.score[n] = f_score[0];
    __threadfence();
    wave_sync();

    // ---- trace back, 64 blocks of history per LDS tile
    int32_t *path = p.path + (size_t)n * T;
    uint8_t *mv = p.moves + (size_t)n * T;
    int el = 0;
    for (int thi = T; thi >= 1; thi -= 64) {
        const int tlo = thi - 63 > 1 ? thi - 63 : 1;          // blocks tlo..thi of the history (entry t describes block t - 1)
        const int rows = thi - tlo + 1;
        for (int i = lane; i < rows * BW; i += 64) tile[i] = __builtin_nontemporal_load(hist + (size_t)tlo * BW + i);
        wave_sync();
        if (lane == 0) {
            for (int t = thi; t >= tlo; --t) {
                const uint32_t info = tile[(t - tlo) * BW + el];
                t_state[t - tlo] = (int)(info & 0xffffu);
                t_move[t - tlo] = (info >> 24) & 1u ? 0 : 1;
                el = (int)((info >> 16) & 0xffu);
            }
        }
        el = __shfl(el, 0, 64);
        wave_sync();
        if (lane < rows) {
            const int t = tlo + lane;                            // history entry t -> block t - 1
            path[t - 1] = t_state[lane];
            mv[t - 1] = t == 1 ? 1 : t_move[lane];               // always a step in the first block
        }
        wave_sync();
    }

    // ---- per-block probability of the path k-mer and of its shifted neighbours (posteriors at t + 1)
    const float lz = p.logz[n];
    float *prob = p.prob + (size_t)n * T;
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        if (t < T) {
            const int st = __builtin_nontemporal_load(path + t);
            const float *a1 = alpha + (size_t)(t + 1) * sstride, *b1 = beta + (size_t)(t + 1) * sstride;
            auto post = [&](int s) { return xb_expf((a1[s] + b1[s]) - lz); };
            float pr = post(st);
            const int l0 = st / nb, r0 = (st % hi) * nb;
            for (int b = 0; b < nb; ++b) {
                pr = pr + post(l0 + hi * b);
                pr = pr + post(r0 + b);
            }
            pr = pr > 1.0f ? 1.0f : pr;
            pr = pr < 0.0f ? 0.0f : pr;
            prob[t] = pr > 0.0f ? xb_expf(0.4f * xb_logf(pr)) : 0.0f;
        }
    }
    __threadfence();
    wave_sync();

    // ---- bases and qualities at the emitting blocks
    int8_t *sq = p.seq + (size_t)n * T, *qs = p.qstr + (size_t)n * T;
    const float nwrong = (float)(nb - 1);
    for (int t0 = 0; t0 < T; t0 += 64) {
        const int t = t0 + lane;
        if (t >= T) continue;
        int8_t base_c = 0, qual_c = 0;
        if (__builtin_nontemporal_load(mv + t)) {
            float bp = 0.0f, tot = 0.0f;
            for (int u = t; u < T && (u == t || !__builtin_nontemporal_load(mv + u)); ++u) {
                const float pr = __builtin_nontemporal_load(prob + u), wrong = (1.0f - pr) / nwrong;
                bp = bp + pr;
                float one = pr;
                for (int j = 1; j < nb; ++j) one = one + wrong;
                tot = tot + one;
            }
            const float e = 1.0f - bp / tot;
            float q = e > 0.0f ? xb_logf(e) * -4.3429448190325175f : 3.402823466e+38f;
            q = q * p.qscale;
            q = q + p.qoffset;
            q = q < 1.0f ? 1.0f : q;
            q = q > 50.0f ? 50.0f : q;
            base_c = (int8_t)p.base_chars[__builtin_nontemporal_load(path + t) % nb];
            qual_c = (int8_t)(int)(33.5f + q);
        }
        sq[t] = base_c;
        qs[t] = qual_c;
    }
}

}  // namespace

namespace xb {

hipError_t launch_beam_search(const BeamParams &p, hipStream_t stream)
{
    if (p.W < 1 || p.W > BEAM_MAX_WIDTH || p.S < 1 || p.S > BEAM_MAX_STATES || p.S > 65535 || p.nb < 2 || p.nb > 7 || p.T < 1 ||
        p.N < 1)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(beam_kernel, dim3(p.N), dim3(64), 0, stream, p);
    return hipGetLastError();
}

}  // namespace xb

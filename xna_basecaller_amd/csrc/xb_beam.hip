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
// goes to HBM and is read back tile by tile for the trace-back.  A block's score row and back-guide row are fetched one block
// ahead by LDS-DMA (global_load_lds, no staging registers) into a double buffer, so the per-candidate gathers are LDS reads and
// the rows' HBM latency is off the block's dependency chain (PRE = false: direct gathers, for rows that do not fit in LDS).
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
// max over the 64 lanes on DPP (one instruction per level; 2 wait states between a VALU write and a DPP read of a register),
// complete in lane 63 and broadcast from there
__device__ __forceinline__ float wave_maxf(float v)
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
    return bits2f((uint32_t)__builtin_amdgcn_readlane((int)f2bits(v), 63));
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
// One wave per workgroup: the LDS unit serves a wave's instructions in order, so a write is visible to every later read of the
// same wave without any wait -- what must not happen is the COMPILER moving an access across the hand-over.  (A __syncthreads()
// here would also wait for vmcnt(0), i.e. for the rows that were requested for the NEXT block a moment ago.)
__device__ __forceinline__ void wave_sync() { asm volatile("" ::: "memory"); }
// the same plus completion of this wave's global stores / loads (history and path written, then read back by other lanes)
__device__ __forceinline__ void wave_sync_global()
{
    __threadfence();
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}

__device__ __forceinline__ float lse2(float x, float y)
{
    const float d = __builtin_fabsf(x - y);
    const float m = x > y ? x : y;
    return d < 17.0f ? m + xb_logf(1.0f + xb_expf(-d)) : m;
}

__device__ __forceinline__ void dma_to_lds(const float *g, float *lds_wave_base, bool vec16)
{
    if (vec16)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
    else
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)lds_wave_base, 4, 0, 0);
}

template <bool PRE>
__global__ __launch_bounds__(64) void beam_kernel(xb::BeamParams p)
{
    extern __shared__ __attribute__((aligned(16))) float pre_smem[];      // PRE: [2][row image | back-guide image]
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
    __shared__ uint32_t st_tab[xb::BEAM_MAX_STATES];      // per state: leading digit | latest base << 4 | shifted rest * nb << 8

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
    for (int st = lane; st < S; st += 64) {
        const int k = st / hi;
        st_tab[st] = (uint32_t)k | ((uint32_t)(st % nb) << 4) | ((uint32_t)((st - k * hi) * nb) << 8);
    }
    int cand_pi[4], cand_b[4];          // step candidate c = q * 64 + lane: element c / nb, base c % nb (the same in every block)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = q * 64 + lane;
        cand_pi[q] = c / nb;
        cand_b[q] = c - cand_pi[q] * nb;
    }
    wave_sync();
    auto crc32c = [&](uint32_t crc, uint32_t v) {
        crc ^= v;
#pragma unroll
        for (int k = 0; k < 4; ++k) crc = crc_tab[crc & 0xffu] ^ (crc >> 8);
        return crc;
    };

    // row t and back guide t + 1 -> buffer b (exact images: buffer[j] = row[j], buffer[beta_off + s] = beta[s])
    auto prefetch = [&](int t, int b) {
        const float *row = p.scores + ((size_t)t * N + n) * p.ld;
        const float *bt = beta + (size_t)(t + 1) * sstride;
        float *dst = pre_smem + (size_t)b * p.pre_stride;
        const int rstep = p.row_vec16 ? 256 : 64, rper = p.row_vec16 ? 4 : 1;
        for (int i = 0; i < p.row_floats; i += rstep) {
            const int j = i + lane * rper;
            if (j < p.row_floats) dma_to_lds(row + j, dst + i, p.row_vec16 != 0);
        }
        dst += p.beta_off;
        const int bstep = p.beta_vec16 ? 256 : 64, bper = p.beta_vec16 ? 4 : 1;
        for (int i = 0; i < S; i += bstep) {
            const int j = i + lane * bper;
            if (j < S) dma_to_lds(bt + j, dst + i, p.beta_vec16 != 0);
        }
    };
    if (PRE) prefetch(0, 0);

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
        const float *row, *b1;
        if (PRE) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // block t's images have landed (requested a block ago)
            wave_sync();
            row = pre_smem + (size_t)(t & 1) * p.pre_stride;
            b1 = row + p.beta_off;
            if (t + 1 < T) prefetch(t + 1, (t + 1) & 1);
        } else {
            row = p.scores + ((size_t)t * N + n) * p.ld;
            b1 = beta + (size_t)(t + 1) * sstride;
        }
        const int nstep = Wc * nb, nc = nstep + Wc;
        // every lane builds its (up to) four candidates level by level, so that the four chains of dependent LDS reads
        // (element -> state table -> score / back guide, and the four CRC table steps) are in flight together
        {
            bool isstep[4], valid[4];
            int pi[4], st[4], j[4];
            uint32_t tab[4], fh[4], crc[4];
            float fs[4], m[4], bj[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = q * 64 + lane;
                isstep[q] = c < nstep;
                valid[q] = c < nc;
                pi[q] = isstep[q] ? cand_pi[q] : (valid[q] ? c - nstep : 0);
                st[q] = f_state[pi[q]];
                fs[q] = f_score[pi[q]];
                fh[q] = f_hash[pi[q]];
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) tab[q] = st_tab[st[q]];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = (int)(tab[q] & 15u);
                j[q] = isstep[q] ? (int)(tab[q] >> 8) + cand_b[q] : st[q];
                const int col = p.has_blank ? (isstep[q] ? j[q] * E + 1 + k : st[q] * E) : (isstep[q] ? j[q] * nb + k : 0);
                m[q] = row[col];
                bj[q] = b1[j[q]];
                crc[q] = fh[q] ^ (uint32_t)j[q];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) crc[q] = crc_tab[crc[q] & 0xffu] ^ (crc[q] >> 8);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = q * 64 + lane;
                const float mv = (!p.has_blank && !isstep[q]) ? p.blank : m[q];
                const float v = (fs[q] + mv) + bj[q];
                sc[q] = valid[q] ? v : NEG_MAX;
                if (valid[q]) {
                    c_hash[c] = isstep[q] ? crc[q] : fh[q];
                    c_info[c] = (uint32_t)j[q] | ((uint32_t)pi[q] << 16) | (isstep[q] ? 0u : 1u << 24);
                    c_score[c] = v;
                    if (isstep[q]) claim[c] = -1;
                }
            }
        }
        wave_sync();
        // ---- a stay and a step that spell the same sequence are one path: merge them
        {
            uint32_t match = 0;
            int latest = 0;
            if (lane < Wc) {
                latest = (int)((st_tab[f_state[lane]] >> 4) & 15u);
                const uint32_t h = f_hash[lane];
#pragma unroll
                for (int j = 0; j < BW; ++j)               // all 32 reads in flight at once; elements beyond the front masked out
                    match |= ((c_hash[(j < Wc ? j : 0) * nb + latest] == h && j < Wc) ? 1u : 0u) << j;
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
    wave_sync_global();

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
    wave_sync_global();

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
    wave_sync_global();

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
    // the double buffer of (score row, back-guide row) images; rows that do not fit are gathered from memory directly
    BeamParams q = p;
    const int cin = p.has_blank ? p.S * (p.nb + 1) : p.S * p.nb;
    q.row_vec16 = (p.ld % 4 == 0 && p.ld >= ((cin + 3) & ~3) && reinterpret_cast<uintptr_t>(p.scores) % 16 == 0) ? 1 : 0;
    q.beta_vec16 = (p.S % 4 == 0 && reinterpret_cast<uintptr_t>(p.beta) % 16 == 0) ? 1 : 0;
    q.row_floats = q.row_vec16 ? (cin + 3) & ~3 : cin;
    q.beta_off = (q.row_floats + 3) & ~3;
    q.pre_stride = (q.beta_off + p.S + 3) & ~3;
    const size_t lds = sizeof(float) * 2 * (size_t)q.pre_stride;
    if (lds <= 96 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&beam_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(beam_kernel<true>, dim3(p.N), dim3(64), lds, stream, q);
    } else {
        hipLaunchKernelGGL(beam_kernel<false>, dim3(p.N), dim3(64), 0, stream, q);
    }
    return hipGetLastError();
}

}  // namespace xb

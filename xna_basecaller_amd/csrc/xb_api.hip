// xb_api.hip -- C ABI of libxnacall.so (see include/xna_basecaller.h for the contract and the
// reference call sites each entry point replaces).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/xna_basecaller.h"
#include "xb_internal.h"

using xb::half_t;

namespace {

thread_local std::string g_create_error;

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
};

struct StageEvent {
    int stage;
    hipEvent_t a, b;
};

}  // namespace

struct xb_ctx {
    xb_config cfg{};
    int device = 0;
    int cu_count = 256;
    hipStream_t stream = nullptr;    // main stream (highest priority): everything except the overlapped GEMM slabs
    hipStream_t stream2 = nullptr;   // low-priority stream: the next layer's input GEMM, slab by slab, beside the recurrence
    hipStream_t stream3 = nullptr;   // low-priority stream: CRF decode of batch k beside the encoder of batch k+1
    hipEvent_t dec_done[2] = {};     // decode that last read scores buffer p has finished
    bool dec_pending[2] = {};
    unsigned batch_idx = 0;
    hipStream_t result_stream = nullptr;   // stream that produces the outputs of the most recent *_dev call
    // host pipeline (xb_submit_chunks / xb_collect_chunks): two slots of pinned staging + device buffers
    struct Slot {
        float *h_signal = nullptr, *d_signal = nullptr;
        int8_t *h_seq = nullptr, *d_seq = nullptr;
        int32_t *h_len = nullptr, *d_len = nullptr;
        unsigned *h_err = nullptr;             // snapshot of the device error word taken on the result stream behind this batch
        hipEvent_t h2d = nullptr, done = nullptr;
        int n = 0;
        bool busy = false;
    } slots[XB_PIPELINE_SLOTS];
    bool pipeline_failed = false;          // a collected batch reported a lost rendezvous: every batch in flight fails with it
    hipStream_t stream_copy = nullptr;     // H2D of the next batch beside the compute of the current one
    std::vector<hipEvent_t> deps;    // timing-less events for the cross-stream dependencies (reused every call)
    size_t dep_next = 0;
    int overlap = 1, time_slabs = 16;   // XB_OVERLAP / XB_TIME_SLABS (upper bound; a slab is at least 125 steps)
    int slab_steps = 0;                 // XB_SLAB_STEPS: minimum steps per time slab (default 125)
    // one recurrence launch per layer that reports its time slabs to the GEMM stream (XB_LSTM_SIGNAL, default on where
    // hipStreamWaitValue32 is supported): flag word, the value the last slab of the previous layer published, slab counters
    int lstm_signal = 2;                // 0 off, 1 whenever one launch holds the batch, 2 (default) only above 512 chunks (two groups per workgroup)
    unsigned *sig_flag = nullptr, *sig_done = nullptr;
    unsigned sig_seq = 0;
    mutable std::string err;
    int T = 0, S = 0, hi = 0, O = 0, kp = 0, ld_nb = 0;
    bool weights_ready = false;
    std::map<std::string, std::vector<float>> host_w;
    std::vector<DevBuf> bufs;
    std::vector<DevBuf> wsbufs;                // the batch-sized workspaces (alloc_workspaces)
    bool alloc_ws = false;

    // weights on device
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr, *b3 = nullptr;
    half_t *w3_hi = nullptr, *w3_lo = nullptr;
    half_t *wih_hi[5] = {}, *wih_lo[5] = {}, *whh_hi[5] = {}, *whh_lo[5] = {};
    float *lbias[5] = {};
    half_t *wl_hi = nullptr, *wl_lo = nullptr;
    float *bl = nullptr;
    int w3_exp = 0, wih_exp[5] = {}, whh_exp[5] = {}, wl_exp = 0;   // q8 exponents (XB_PREC_F16F8)
    // fragment-major images of the GEMM B operands (gemm4p_kernel, xb_internal.h) and their k-tile strides; the input
    // projections also as hi-only images for XB_PREC_F16F8_IN1
    unsigned char *w3_f4 = nullptr, *wih_f4[5] = {}, *wih_f4h[5] = {}, *wl_f4 = nullptr;
    size_t w3_ks = 0, wih_ks = 0, wih_ksh = 0, wl_ks = 0;
    int gemm_sn = 0;                           // XB_GEMM_SN: N tiles per XCD super-tile of gemm4p_kernel (0 = gemm_super_n's rule; experiments)
    int gemm_shadow_kernel = 0;                // XB_GEMM_SHADOW: 0 auto (by batch size), 4 gemm4p_kernel, 8 gemm8r_kernel for the slabs beside the recurrence
    int gemm_shadow_wgs = 2;                   // XB_GEMM_SHADOW_WGS=1: GEMM slabs beside the recurrence run one workgroup per CU
    int gemm4 = 1;                             // XB_GEMM4=0: gemm8r_kernel (one workgroup per CU) instead of gemm4p_kernel (A/B comparisons)
    std::vector<void *> wbufs;                 // weight allocations of the current xb_weights_ready (freed by the next one)
    int8_t *whh_q1[5] = {}, *whh_q0[5] = {};   // int8-limb recurrence (lstm_i8): balanced digits of W_hh, gate-interleaved rows
    float *whh_sc[5] = {};                     // ... and the factor that turns the integer sum into the recurrent term
    int lstm_i8 = 0;                           // XB_LSTM_I8 (with precision f16f8 / f16f8i): recurrence on int8 digits; 1 = all
                                               // four digit products, 2 = without d0 x d0

    // activations / workspaces
    float *d_signal = nullptr;
    half_t *im_hi = nullptr, *im_lo = nullptr;
    half_t *x_hi[2] = {}, *x_lo[2] = {};
    float *gin = nullptr, *gin2 = nullptr, *c_state = nullptr, *scores = nullptr, *scores2 = nullptr;
    half_t *xh = nullptr;        // LSTM exchange buffer: 64 groups x 2 parity x 2 parts x 64 chunks x F
    float *alpha = nullptr, *beta = nullptr, *bmax = nullptr, *qbuf = nullptr, *logz = nullptr;
    // beam search workspaces and staging (lazily allocated: most contexts never use them)
    uint32_t *beam_hist = nullptr;
    int32_t *beam_path = nullptr;
    float *beam_prob = nullptr, *beam_score = nullptr;
    int8_t *beam_seq = nullptr, *beam_q = nullptr;
    uint8_t *beam_moves = nullptr;
    int8_t *labels = nullptr, *seq = nullptr;
    int32_t *seq_len = nullptr;
    unsigned *sync = nullptr;    // [64 groups * 32] counters + error word at the end
    unsigned *error = nullptr;
    int lstm_mode = 0;
    // workgroups of the persistent kernel admitted per CU, by recurrence arithmetic (nsplit 1..5) and one / two groups per
    // workgroup (occupancy query, lazily; -1 = not asked yet)
    int lstm_resident[6][2] = {{-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}, {-1, -1}};
    // Arithmetic of every contraction stage (GemmParams::nsplit: 1 fp16 product, 2 + FP8 corrections, 3 three fp16 products):
    // conv3, the five input projections, the five recurrences, the CRF linear layer.  One value everywhere for the plain
    // precisions; XB_PREC_MIXED (and the diagnostic XB_X3_STAGES mask) mix 2 and 3.  An activation tensor's second part
    // (q8 image or fp16 residual) follows the stage that CONSUMES it.
    int ns_conv = 3, ns_in[5] = {3, 3, 3, 3, 3}, ns_rec[5] = {3, 3, 3, 3, 3}, ns_lin = 3;
    int in1_layers = 31;         // XB_IN1_LAYERS (diagnostic): layers whose input projection XB_PREC_F16F8_IN1 reduces
    int decode_async = 0;        // XB_DECODE_ASYNC=1: the decode of a batch runs on the third stream beside the next batch's conv + first
                                 // GEMM (rounds 2-3) instead of on the main stream with the chip to itself (round 4 default: the same step
                                 // time at every batch size -- the step is bound by the kernels' summed CU-time -- and the decode at 0.49-0.52
                                 // of the HBM roofline instead of 0.35-0.42: profiles/r04_decode_placement.txt)
    int lstm_local = 1;          // XB_LSTM_LOCAL=0: always exchange h with write-through stores (A/B)
    int lstm_wide = 1;           // XB_LSTM_WIDE: 1 (default) batches just above a launch's XCD-local capacity get up to cu_count / members group
                                 // slots with the groups dealt over all XCDs instead of a second round (run_lstm_layer); 0: never
    int lstm_dual = 1;           // XB_LSTM_DUAL: 0 never, 1 when a launch would otherwise need a second chunk slab, 2 always
    int lstm_quad = 0;           // XB_LSTM_QUAD=1: the two-groups-per-workgroup launches run the software-pipelined kernel of xb_lstm_quad.h
                                 // where it applies (F = 768, q8 exchange image).  Bit-identical and SLOWER (41.9 vs 31.0 ms per layer of
                                 // 1024 chunks): an experiment kept for its measurements, never the default
    int lstm_quad_res = -1;      // its occupancy (queried once)

    // Two asynchronous basecalls in flight are co-scheduled once the caller has opted in with xb_reserve_pairing (contexts of at
    // most 512 chunks; XB_FUSE=0 refuses): the first xb_basecall_chunks_dev of a pair is held back until the second arrives, then
    // both batches go through the encoder and the decode as ONE batch (the recurrence then runs two chunk groups per workgroup,
    // DESIGN.md 4.1 / 4.5).  Every other entry point, xb_synchronize and xb_result_stream first launch a held-back call on its own.
    // Without the opt-in every asynchronous call is enqueued before it returns.
    struct Call {
        const float *signal = nullptr;
        int n = 0;
        char alphabet[16] = {};
        int8_t *seq = nullptr;
        int32_t *len = nullptr;
        int slot = -1;                          // host pipeline slot whose D2H copies and done event follow the launch
        void (*after)(void *) = nullptr;        // xb_comm: the gather of this call's results, enqueued right behind it
        void *after_arg = nullptr;
    };
    int fuse_ok = 1;                            // pairing is possible in this context (schedule, batch size, XB_FUSE)
    int fuse = 0;                               // ... and the caller asked for it (xb_reserve_pairing)
    int cap = 0;                                // chunks the workspaces hold (2 * max_batch when fusing is possible)
    Call held;
    bool holding = false, flushing = false;
    int deferred_rc = 0;                        // failure of a held-back call that was launched where no status could be returned
    int8_t *fseq = nullptr;                     // (cap, T) / (cap) results of a fused pair before they are split
    int32_t *flen = nullptr;

    bool profiling = false;
    std::vector<StageEvent> events;
    float stage_ms[XB_STAGE_COUNT] = {};
    int64_t stage_launches[XB_STAGE_COUNT] = {};
};

extern "C" int flush_held(xb_ctx *ctx);      // launches a held-back asynchronous basecall (defined with it below)

namespace {

int fail(const xb_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf; else g_create_error = buf;
    return code;
}

#define XB_HIP(ctx, call)                                                                     \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(ctx, e_ == hipErrorOutOfMemory ? XB_ERR_NOMEM : XB_ERR_HIP,           \
                        "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int64_t ipow(int64_t b, int e) { int64_t r = 1; while (e-- > 0) r *= b; return r; }

template <typename Tp>
int dev_alloc(xb_ctx *ctx, Tp **out, size_t count)
{
    void *p = nullptr;
    const size_t bytes = (count * sizeof(Tp) + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
        return fail(ctx, XB_ERR_NOMEM, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    (ctx->alloc_ws ? ctx->wsbufs : ctx->bufs).push_back({p, bytes});
    *out = reinterpret_cast<Tp *>(p);
    return XB_OK;
}

// The workspaces whose size follows the number of chunks in a pass.  A context starts with room for max_batch chunks; the
// first time two calls are co-scheduled (or on xb_reserve_pairing) they are replaced by twice that.
int alloc_workspaces(xb_ctx *ctx, int cap)
{
    for (auto &b : ctx->wsbufs) (void)hipFree(b.p);
    ctx->wsbufs.clear();
    ctx->cap = 0;
    const xb_config *cfg = &ctx->cfg;
    const int S = ctx->S;
    const size_t N = (size_t)cap, T = ctx->T, F = cfg->features, L = cfg->chunk_len;
    const size_t Cb = (size_t)S * (cfg->n_base + 1);
    const size_t Cmax = Cb > (size_t)ctx->ld_nb ? Cb : (size_t)ctx->ld_nb;
    ctx->alloc_ws = true;
    int rc = XB_OK;
    rc = rc ? rc : dev_alloc(ctx, &ctx->d_signal, N * L);
    rc = rc ? rc : dev_alloc(ctx, &ctx->im_hi, T * N * ctx->kp);
    rc = rc ? rc : dev_alloc(ctx, &ctx->im_lo, T * N * ctx->kp);
    for (int i = 0; i < 2 && !rc; ++i) {
        rc = rc ? rc : dev_alloc(ctx, &ctx->x_hi[i], T * N * F);
        rc = rc ? rc : dev_alloc(ctx, &ctx->x_lo[i], T * N * F);
    }
    // + 64 rows: the recurrence's LDS-DMA of a ragged last group reads (and ignores) up to 63 rows past the last chunk
    rc = rc ? rc : dev_alloc(ctx, &ctx->gin, (T * N + 64) * 4 * F);
    if (ctx->overlap) rc = rc ? rc : dev_alloc(ctx, &ctx->gin2, (T * N + 64) * 4 * F);
    rc = rc ? rc : dev_alloc(ctx, &ctx->c_state, N * F);
    rc = rc ? rc : dev_alloc(ctx, &ctx->scores, T * N * Cmax);
    if (ctx->overlap && ctx->decode_async) rc = rc ? rc : dev_alloc(ctx, &ctx->scores2, T * N * (size_t)ctx->ld_nb);
    rc = rc ? rc : dev_alloc(ctx, &ctx->alpha, (T + 1) * N * S);
    rc = rc ? rc : dev_alloc(ctx, &ctx->beta, (T + 1) * N * S);
    rc = rc ? rc : dev_alloc(ctx, &ctx->bmax, (T + 1) * N * S);
    rc = rc ? rc : dev_alloc(ctx, &ctx->logz, N);
    rc = rc ? rc : dev_alloc(ctx, &ctx->qbuf, T * N * ((Cb + 3) & ~(size_t)3));
    rc = rc ? rc : dev_alloc(ctx, &ctx->labels, N * T);
    rc = rc ? rc : dev_alloc(ctx, &ctx->seq, N * T);
    rc = rc ? rc : dev_alloc(ctx, &ctx->seq_len, N);
    if (ctx->fuse_ok) {      // results of a pair before they are split (two short calls can pair inside max_batch chunks)
        rc = rc ? rc : dev_alloc(ctx, &ctx->fseq, N * T);
        rc = rc ? rc : dev_alloc(ctx, &ctx->flen, N);
    }
    ctx->alloc_ws = false;
    if (!rc) ctx->cap = cap;
    return rc;
}

struct StageScope {
    xb_ctx *c;
    int stage;
    hipEvent_t a = nullptr, b = nullptr;
    hipStream_t st_;
    StageScope(xb_ctx *ctx, int st, int launches, hipStream_t stream = nullptr)
        : c(ctx), stage(st), st_(stream ? stream : ctx->stream)
    {
        c->stage_launches[st] += launches;
        if (c->profiling && hipEventCreate(&a) == hipSuccess && hipEventCreate(&b) == hipSuccess)
            (void)hipEventRecord(a, st_);
    }
    ~StageScope()
    {
        if (a && b) {
            (void)hipEventRecord(b, st_);
            c->events.push_back({stage, a, b});
        }
    }
};

int collect_events(xb_ctx *ctx)
{
    for (auto &ev : ctx->events) {
        float ms = 0.f;
        if (hipEventSynchronize(ev.b) == hipSuccess && hipEventElapsedTime(&ms, ev.a, ev.b) == hipSuccess)
            ctx->stage_ms[ev.stage] += ms;
        (void)hipEventDestroy(ev.a);
        (void)hipEventDestroy(ev.b);
    }
    ctx->events.clear();
    return XB_OK;
}

// expected element count of each state-dict tensor
int64_t expected_size(const xb_ctx *c, const std::string &name)
{
    const int64_t F = c->cfg.features, W = c->cfg.winlen;
    if (name == "encoder.0.conv.weight") return 4 * 1 * 5;
    if (name == "encoder.0.conv.bias") return 4;
    if (name == "encoder.1.conv.weight") return 16 * 4 * 5;
    if (name == "encoder.1.conv.bias") return 16;
    if (name == "encoder.2.conv.weight") return F * 16 * W;
    if (name == "encoder.2.conv.bias") return F;
    for (int l = 4; l <= 8; ++l) {
        const std::string pre = "encoder." + std::to_string(l) + ".rnn.";
        if (name == pre + "weight_ih_l0" || name == pre + "weight_hh_l0") return 4 * F * F;
        if (name == pre + "bias_ih_l0" || name == pre + "bias_hh_l0") return 4 * F;
    }
    if (name == "encoder.9.linear.weight") return (int64_t)c->O * F;
    if (name == "encoder.9.linear.bias") return c->O;
    return -1;
}

// OCP e4m3 (fn) encoding of x: round to nearest even, saturating at +-448 (the MFMA's operand format on gfx950)
uint8_t f32_to_e4m3(float x)
{
    const uint8_t sign = std::signbit(x) ? 0x80 : 0x00;
    const float a = std::fabs(x);
    if (!(a == a)) return sign | 0x7f;
    if (a >= 448.0f) return sign | 0x7e;
    if (a < 0.015625f) {                                   // subnormal: steps of 2^-9
        const int q = (int)std::nearbyint(std::ldexp(a, 9));
        return sign | (uint8_t)q;                          // q == 8 is the smallest normal, encoded 0x08 as well
    }
    int ex;
    (void)std::frexp(a, &ex);                              // a = m * 2^ex, m in [0.5, 1)
    int e = ex - 1;
    int q = (int)std::nearbyint(std::ldexp(a, 3 - e));     // 8 .. 16
    if (q == 16) { q = 8; ++e; }
    const int code = ((e + 7) << 3) | (q - 8);
    return sign | (uint8_t)(code > 0x7e ? 0x7e : code);
}

// rows of `cols` floats -> split fp16 (hi, lo) with leading dimension ld.  q8_exp != nullptr: `lo` receives the q8
// image instead (xb_internal.h): per row and 32 columns [32 x e4m3(hi * 2^e) | 32 x e4m3(lo * 2^(e+11))], e chosen so
// that the largest |value| lands near 224, and *q8_exp = e.
void split_rows(const float *src, int rows, int cols, int ld, std::vector<half_t> &hi, std::vector<half_t> &lo,
                int *q8_exp = nullptr)
{
    hi.assign((size_t)rows * ld, (half_t)0.0f);
    lo.assign((size_t)rows * ld, (half_t)0.0f);
    int e = 0;
    if (q8_exp) {
        float amax = 0.0f;
        for (size_t i = 0; i < (size_t)rows * cols; ++i) amax = std::fmax(amax, std::fabs(src[i]));
        if (amax > 0.0f && std::isfinite(amax)) e = (int)std::floor(std::log2(448.0f / amax)) - 1;
        e = e < -16 ? -16 : (e > 32 ? 32 : e);
        *q8_exp = e;
    }
    uint8_t *q = reinterpret_cast<uint8_t *>(lo.data());
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            const float v = src[(size_t)r * cols + c];
            const half_t h = (half_t)v;
            const float l = v - (float)h;
            hi[(size_t)r * ld + c] = h;
            if (!q8_exp) {
                lo[(size_t)r * ld + c] = (half_t)l;
            } else {
                uint8_t *blk = q + ((size_t)r * ld + (c & ~31)) * 2;
                blk[c & 31] = f32_to_e4m3(std::ldexp((float)h, e));
                blk[32 + (c & 31)] = f32_to_e4m3(std::ldexp(l, e + 11));
            }
        }
}

// Fragment-major image of a GEMM B operand for gemm4p_kernel (layout: xb_internal.h, GemmParams::b4).  `hi` / `lo` are
// split_rows outputs with leading dimension ld (lo = the fp16 residual for nsplit 3, the q8 image for nsplit 2, unused for
// nsplit 1); rows are padded with zeros to a multiple of 256 so that no tile needs a bounds check.
void fragment_major(const std::vector<half_t> &hi, const std::vector<half_t> &lo, int rows, int ld, int K, int nsplit,
                    std::vector<unsigned char> &out, size_t *kstride)
{
    const int rows4 = (rows + 255) & ~255, nt32 = rows4 / 32, npc = xb::gemm4_pieces(nsplit), nk = K / 32;
    *kstride = (size_t)nt32 * npc * 1024;
    out.assign((size_t)nk * *kstride, 0);
    const unsigned char *hib = reinterpret_cast<const unsigned char *>(hi.data());
    const unsigned char *lob = reinterpret_cast<const unsigned char *>(lo.data());
    if (nsplit == 3 && XB_GEMM_S16 != 0) {
        // the 16x16x32 arithmetic: piece 2 * part + c = rows 16 c .. 16 c + 15 of the block, lane l = row (l & 15), k 8 (l >> 4) .. + 8
        for (int kt = 0; kt < nk; ++kt)
            for (int nt = 0; nt < nt32; ++nt)
                for (int c = 0; c < 2; ++c)
                    for (int l = 0; l < 64; ++l) {
                        const int r = nt * 32 + c * 16 + (l & 15);
                        if (r >= rows) continue;
                        unsigned char *blk = out.data() + (size_t)kt * *kstride + (size_t)nt * npc * 1024 + (size_t)l * 16;
                        const size_t e0 = (size_t)r * ld + (size_t)kt * 32 + (size_t)(l >> 4) * 8;
                        memcpy(blk + c * 1024, hib + e0 * 2, 16);
                        memcpy(blk + (2 + c) * 1024, lob + e0 * 2, 16);
                    }
        return;
    }
    for (int kt = 0; kt < nk; ++kt)
        for (int nt = 0; nt < nt32; ++nt)
            for (int l = 0; l < 64; ++l) {
                const int r = nt * 32 + (l & 31), h = l >> 5;
                if (r >= rows) continue;
                unsigned char *blk = out.data() + (size_t)kt * *kstride + (size_t)nt * npc * 1024 + (size_t)l * 16;
                const size_t e0 = (size_t)r * ld + (size_t)kt * 32;          // element offset of the row's k-tile
                for (int ks = 0; ks < 2; ++ks) {
                    memcpy(blk + ks * 1024, hib + (e0 + ks * 16 + h * 8) * 2, 16);
                    if (nsplit == 3) memcpy(blk + (2 + ks) * 1024, lob + (e0 + ks * 16 + h * 8) * 2, 16);
                }
                if (nsplit == 2) {
                    // q8 block of the 32 columns: [h8 x 32 | l8 x 32]; the B role reads l8 in lanes 0-31, h8 in lanes 32-63
                    const unsigned char *q = lob + e0 * 2 + (h == 0 ? 32 : 0);
                    memcpy(blk + 2 * 1024, q, 16);
                    memcpy(blk + 3 * 1024, q + 16, 16);
                }
            }
}

// weight upload: the allocation belongs to the current weight set (ctx->wbufs), which the next xb_weights_ready releases
template <typename Tp>
int upload(xb_ctx *ctx, Tp **dst, const std::vector<Tp> &src)
{
    void *p = nullptr;
    const size_t bytes = (src.size() * sizeof(Tp) + 255) & ~(size_t)255;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
        return fail(ctx, XB_ERR_NOMEM, "hipMalloc of %zu bytes failed: %s", bytes, hipGetErrorString(e));
    ctx->wbufs.push_back(p);
    *dst = reinterpret_cast<Tp *>(p);
    XB_HIP(ctx, hipMemcpy(*dst, src.data(), src.size() * sizeof(Tp), hipMemcpyHostToDevice));
    return XB_OK;
}

int check_ready(xb_ctx *ctx, int n)
{
    if (!ctx) return XB_ERR_INVALID;
    if (!ctx->weights_ready) return fail(ctx, XB_ERR_STATE, "weights not loaded: call xb_load_weights for all 28 tensors, then xb_weights_ready");
    if (n < 1 || n > ctx->cfg.max_batch) return fail(ctx, XB_ERR_INVALID, "batch %d outside [1, max_batch=%d]", n, ctx->cfg.max_batch);
    if (ctx->cap < ctx->cfg.max_batch) return fail(ctx, XB_ERR_NOMEM, "the context lost its workspaces (a reallocation failed)");
    return XB_OK;
}

int check_device_error(xb_ctx *ctx)
{
    unsigned e = 0;
    XB_HIP(ctx, hipMemcpyAsync(&e, ctx->error, sizeof e, hipMemcpyDeviceToHost, ctx->stream));
    XB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (e != 0) {
        (void)hipMemsetAsync(ctx->error, 0, sizeof(unsigned), ctx->stream);
        if (e == 2u) return fail(ctx, XB_ERR_DEVICE, "CTC scan: a target length was outside the lattice");
        return fail(ctx, XB_ERR_DEVICE, "LSTM inter-workgroup sync timed out (persistent kernel was not fully resident?)");
    }
    return XB_OK;
}

// ---- encoder orchestration --------------------------------------------------------------
int precision_nsplit(int pr)
{
    return pr == XB_PREC_F16 ? 1 : ((pr == XB_PREC_F16F8 || pr == XB_PREC_F16F8_IN1 || pr == XB_PREC_MIXED) ? 2 : 3);
}

// Stage mask of the three-product arithmetic inside an f16f8 context: bits 0-4 = input projection of LSTM layer l, bits 5-9 =
// recurrence of layer l, bit 10 = CRF linear layer, bit 11 = conv3.  XB_PREC_MIXED: every feed-forward projection.  Measured on the
// peaky model (profiles/r04_x3_attribution.txt): the error variance of plain f16f8 splits as input projections 65 %,
// linear 18 %, recurrences 13 %, conv3 5 %; per ms of step time the recurrences buy the least, and they are the critical path.
constexpr int X3_MIXED_STAGES = 0x1f | (1 << 10) | (1 << 11);
void set_stage_arithmetic(xb_ctx *ctx, int x3_mask)
{
    const int base = precision_nsplit(ctx->cfg.precision);
    const bool mix = base == 2;
    for (int l = 0; l < 5; ++l) {
        ctx->ns_in[l] = mix && ((x3_mask >> l) & 1) ? 3 : base;
        ctx->ns_rec[l] = mix && ((x3_mask >> (5 + l)) & 1) ? 3 : base;
    }
    ctx->ns_lin = mix && ((x3_mask >> 10) & 1) ? 3 : base;
    ctx->ns_conv = mix && ((x3_mask >> 11) & 1) ? 3 : base;
}
// second part an activation tensor needs for a consumer of arithmetic ns: 2 = q8 image, 1 = fp16 residual, 0 = none read
int second_part(int ns) { return ns == 2 ? 2 : (ns == 3 ? 1 : 0); }

// the GEMM that consumes a layer's output rows of time steps [ta, tb): the input projection of LSTM layer `layer`
// (layer < 5, into `gin_out`) or the CRF linear layer (layer == 5, into the scores)
struct NextGemm {
    int layer;
    const half_t *x_hi, *x_lo;
    float *out;
    int ldc, expand;
};

int launch_row_gemm(xb_ctx *ctx, const NextGemm &ng, int n, int ta, int tb, hipStream_t st, bool shadow = false)
{
    const xb_config &c = ctx->cfg;
    const int F = c.features;
    xb::GemmParams g{};
    const size_t r0 = (size_t)ta * n;
    g.a_hi = ng.x_hi + r0 * F; g.a_lo = ng.x_lo + r0 * F;
    g.M = (tb - ta) * n; g.K = F; g.lda = F; g.ldb = F; g.nsplit = ng.layer < 5 ? ctx->ns_in[ng.layer] : ctx->ns_lin;
    g.ldc = ng.ldc; g.out_f32 = ng.out + r0 * ng.ldc;
    g.one_per_cu = shadow && ctx->gemm_shadow_wgs == 1;
    g.sn = ctx->gemm_sn;
    // which kernel: gemm4p_kernel, except for the slabs that run beside the two-groups-per-workgroup recurrence of a batch
    // above 1024 chunks, where the one-workgroup-per-CU gemm8r_kernel disturbs the recurrence less (same box, ms per step at
    // batch 2048: 477 vs 488 (gemm4p, one workgroup per CU) vs 499; batch 1024: 240 vs 246 vs 235; batch 512: 126.7 vs 129.7 vs
    // 123.0 -- profiles/r03_gemm_kernel_by_batch.txt).  XB_GEMM_SHADOW=4 / 8 forces one of them.
    const bool use4 = ctx->gemm4 && !(shadow && (ctx->gemm_shadow_kernel == 8 || (ctx->gemm_shadow_kernel == 0 && n > 1024)));
    if (ng.layer < 5) {
        StageScope sc(ctx, XB_STAGE_LSTM_IN, 1, st);
        g.b_hi = ctx->wih_hi[ng.layer]; g.b_lo = ctx->wih_lo[ng.layer]; g.Nn = 4 * F; g.bias = ctx->lbias[ng.layer];
        // main product only (the q8 images stay unused) -- in1_layers: bit l = input projection of layer l (diagnostic
        // XB_IN1_LAYERS, default all five)
        if (use4) { g.b4 = ctx->wih_f4[ng.layer]; g.b4_kstride = ctx->wih_ks; }
        if (ctx->cfg.precision == XB_PREC_F16F8_IN1 && ((ctx->in1_layers >> ng.layer) & 1)) {
            g.nsplit = 1;
            if (use4) { g.b4 = ctx->wih_f4h[ng.layer]; g.b4_kstride = ctx->wih_ksh; }
        }
        g.a_exp = ng.layer == 0 ? 0 : 8; g.b_exp = ctx->wih_exp[ng.layer];     // conv3 output / LSTM output
        g.gin_n = n;                                                           // member-major gin (xb_internal.h)
        XB_HIP(ctx, xb::launch_gemm(g, xb::EPI_BIAS_F32, st));
    } else {
        StageScope sc(ctx, XB_STAGE_LINEAR, 1, st);
        g.b_hi = ctx->wl_hi; g.b_lo = ctx->wl_lo; g.Nn = ctx->O; g.bias = ctx->bl;
        g.scale = c.scale; g.nb = c.n_base; g.expand = ng.expand; g.blank = c.blank_score;
        g.a_exp = 8; g.b_exp = ctx->wl_exp;
        if (use4) { g.b4 = ctx->wl_f4; g.b4_kstride = ctx->wl_ks; }
        XB_HIP(ctx, xb::launch_gemm(g, xb::EPI_TANH_SCALE, st));
    }
    return XB_OK;
}

int next_dep(xb_ctx *ctx, hipEvent_t *ev)
{
    if (ctx->dep_next == ctx->deps.size()) {
        hipEvent_t e;
        XB_HIP(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->deps.push_back(e);
    }
    *ev = ctx->deps[ctx->dep_next++];
    return XB_OK;
}

int sync_all(xb_ctx *ctx);

// Recurrence of one layer from `gin` into (xout_hi, xout_lo).  With `next` set, the GEMM that consumes this layer's output
// is issued as well: either afterwards on the main stream, or -- overlapped mode -- slab by slab on the second stream while
// the recurrence (192 of the 256 CUs, latency bound) is still running; the main stream then waits for the last slab.
// group slots of the arrival counters: 64 groups of 64 chunks, or (lstm_quad_kernel) 128 groups of 32
constexpr int SYNC_SLOTS = 128;

int run_lstm_layer(xb_ctx *ctx, int layer, int n, const float *gin, half_t *xout_hi, half_t *xout_lo, const NextGemm *next)
{
    const int F = ctx->cfg.features, T = ctx->T;
    const int members = xb::lstm_members(F), bn = xb::lstm_group_chunks();
    int mode = ctx->lstm_mode;
    // groups per persistent launch: every workgroup must be resident at once, and workgroups are dealt to the 8 XCDs
    // strictly round-robin (block b -> XCD b % 8), i.e. groups g, g + 8, .. share ONE XCD's CUs: F = 768 (24 members per
    // group, 32 CUs per XCD) allows one group per XCD = 8 groups = 512 chunks per launch
    // ... and the occupancy calculator has to admit at least one such workgroup per CU (queried once per context); a
    // context that cannot keep the persistent kernel resident falls back to one launch per time step
    // what the GEMM that reads this layer's output needs as the second part; the int8-limb recurrence writes hi + q8 only
    const int y_need = second_part(layer < 4 ? ctx->ns_in[layer + 1] : ctx->ns_lin);
    const bool i8 = ctx->lstm_i8 && ctx->whh_q1[layer] && ctx->ns_rec[layer] == 2 && y_need != 1;
    const int rec_nsplit = i8 ? (ctx->lstm_i8 == 2 ? 5 : 4) : ctx->ns_rec[layer];
    int (&res)[2] = ctx->lstm_resident[rec_nsplit];
    if (res[0] < 0) res[0] = xb::lstm_resident_per_cu(F, rec_nsplit, 0);
    if (res[1] < 0) res[1] = xb::lstm_resident_per_cu(F, rec_nsplit, 1);
    const bool dual_ok = ctx->lstm_dual != 0 && res[1] >= 1;
    const int gmax = res[0] >= 1 ? 8 * ((ctx->cu_count / 8) / members) : 0;
    if (mode == 0) mode = gmax >= 1 ? 2 : 1;
    if (mode == 2 && gmax < 1) return fail(ctx, XB_ERR_INVALID, "persistent LSTM needs %d co-resident workgroups, device has %d CUs", members, ctx->cu_count);
    XB_HIP(ctx, hipMemsetAsync(ctx->c_state, 0, sizeof(float) * (size_t)n * F, ctx->stream));
    xb::LstmParams p{};
    p.gin = gin; p.w_hi = ctx->whh_hi[layer]; p.w_lo = ctx->whh_lo[layer];
    p.y_hi = xout_hi; p.y_lo = xout_lo; p.c_state = ctx->c_state; p.xh = ctx->xh;
    p.T = T; p.N = n; p.F = F; p.reverse = (layer % 2) == 0;
    p.sync = ctx->sync; p.error = ctx->error; p.nsplit = ctx->ns_rec[layer]; p.w_exp = ctx->whh_exp[layer];
    p.y_alt = (p.nsplit == 2 || p.nsplit == 3) && y_need != 0 && y_need != second_part(p.nsplit);
    if (i8) {
        p.nsplit = ctx->lstm_i8 == 2 ? 5 : 4; p.wq1 = ctx->whh_q1[layer]; p.wq0 = ctx->whh_q0[layer]; p.wscale = ctx->whh_sc[layer];
    }
    if (const char *e = getenv("XB_LSTM_SPREAD")) p.spread = atoi(e) != 0;
    // the software-pipelined kernel serves the launches that put two groups of 64 on a workgroup (as four groups of 32)
    bool quad_ok = ctx->lstm_quad && F == 768 && p.nsplit == 2 && !i8;
    if (quad_ok && ctx->lstm_quad_res < 0) ctx->lstm_quad_res = xb::lstm_quad_resident_per_cu();
    quad_ok = quad_ok && ctx->lstm_quad_res >= 1;
    bool overlapped = false;
    if (mode == 2) {
        // a workgroup can serve two groups alternately (lstm_kernel DUAL): a launch then holds 2 * gmax groups, and a
        // group's hand-off latency is covered by the other group's step.  Used when the batch does not fit gmax groups.
        // WIDE (round 5, the batch cliffs): the XCD-local placement holds gmax = 8 group slots (one group's 24 member workgroups per
        // XCD), so 513 chunks -- nine groups -- used to take the two-groups-per-workgroup kernel over five slots, i.e. the time of
        // 1024 chunks, and 1025 chunks a second launch.  The device has cu_count / members = 10 slots' worth of CUs: with a
        // group's members dealt over ALL XCDs (LstmParams.spread: 3 per XCD and group, 30 of an XCD's 32 CUs at ten slots; the
        // exchange then goes through write-through stores, a few percent slower per step) a launch holds up to 640 chunks with
        // one group per workgroup and 1280 with two.  Used exactly where it saves a round: 513..640 and 1025..1280 chunks.
        const int gslab0 = gmax > 64 ? 64 : gmax;
        const int gwide = ctx->lstm_wide && ctx->cu_count / members > gslab0 ? (ctx->cu_count / members > 64 ? 64 : ctx->cu_count / members) : gslab0;
        const bool wide = gwide > gslab0 && ((n > gslab0 * bn && n <= gwide * bn) ||
                                             (dual_ok && ctx->lstm_dual == 1 && n > 2 * gslab0 * bn && n <= 2 * gwide * bn));
        const int gslab = wide ? gwide : gslab0;
        if (wide) p.spread = 1;
        const bool dual_batch = dual_ok && (ctx->lstm_dual == 2 ? n > bn : n > gslab * bn);
        const int slab = (dual_batch ? (2 * gslab > 64 ? 64 : 2 * gslab) : gslab) * bn;
        // the exchange buffer and the counters have 64 group slots: with the whole batch inside them every group keeps its
        // own slot across launches, so chunk slabs and time slabs combine freely; a larger batch falls back to one launch
        // per chunk slab over all steps with launch-local slots
        const bool global_groups = n <= 64 * bn;
        const int min_steps = ctx->slab_steps > 0 ? ctx->slab_steps : 125;
        int nts = T / min_steps < ctx->time_slabs ? T / min_steps : ctx->time_slabs;
        overlapped = next && ctx->overlap && ctx->stream2 && global_groups && nts >= 2;
        if (!overlapped) nts = 1;
        int launches = 0;
        for (int n0 = 0; n0 < n; n0 += slab) launches += nts;
        if (global_groups) XB_HIP(ctx, hipMemsetAsync(ctx->sync, 0, sizeof(unsigned) * SYNC_SLOTS * 32, ctx->stream));
        // One launch over all steps that reports its time slabs: the GEMM stream waits on the flag word instead of on an event
        // behind a slab launch, so the recurrence is not relaunched 16 times per layer (each relaunch costs ~30 us: its
        // workgroups find their CUs taken by GEMM workgroups that slipped in at the boundary).
        // Measured (profiles/r03_lstm_slab_signal.txt): the recurrence itself gets 13 % faster (98 -> 85 ms per step at batch 512,
        // 187 -> 152 ms at 1024), but at batch 512 the GEMM then gets that much less of the chip and the step stays where it was
        // (120.5 vs 121.2 ms); with two groups per workgroup (batch 1024) the step gains 2 %.  Default: only there.
        const bool signal_mode = ctx->lstm_signal == 1 || (ctx->lstm_signal == 2 && dual_batch);
        if (overlapped && ctx->overlap == 1 && signal_mode && ctx->sig_flag && n <= slab && nts <= 64) {
            if (ctx->sig_seq > (1u << 30)) {           // keep the 32-bit flag monotonic: start over from an idle device
                int rc = sync_all(ctx);
                if (rc) return rc;
                XB_HIP(ctx, hipMemset(ctx->sig_flag, 0, 4));
                ctx->sig_seq = 0;
            }
            XB_HIP(ctx, hipMemsetAsync(ctx->sig_done, 0, sizeof(unsigned) * 64, ctx->stream));
            {
                StageScope sc(ctx, XB_STAGE_LSTM_REC, 1);
                p.n0 = 0; p.nslab = n; p.s_begin = 0; p.s_end = T; p.persistent = 1;
                p.dual = dual_batch && (ctx->lstm_dual == 2 ? n > bn : n > gslab * bn);
                p.quad = p.dual && quad_ok;
                p.grp0 = 0; p.slab = 0; p.xcd_local = ctx->lstm_local; p.sync_base = 0;
                p.sig_flag = ctx->sig_flag; p.sig_done = ctx->sig_done; p.sig_base = ctx->sig_seq; p.sig_nts = nts;
                XB_HIP(ctx, xb::launch_lstm(p, ctx->stream));
            }
            for (int i = 0; i < nts; ++i) {
                const int s0 = (int)((long long)T * i / nts), s1 = (int)((long long)T * (i + 1) / nts);
                XB_HIP(ctx, hipStreamWaitValue32(ctx->stream2, ctx->sig_flag, ctx->sig_seq + (unsigned)i + 1u, hipStreamWaitValueGte,
                                                 0xffffffffu));
                const int ta = p.reverse ? T - s1 : s0, tb = p.reverse ? T - s0 : s1;
                if (int rc = launch_row_gemm(ctx, *next, n, ta, tb, ctx->stream2, true)) return rc;
            }
            ctx->sig_seq += (unsigned)nts;
            hipEvent_t ev;
            int rc = next_dep(ctx, &ev);
            if (rc) return rc;
            XB_HIP(ctx, hipEventRecord(ev, ctx->stream2));
            XB_HIP(ctx, hipStreamWaitEvent(ctx->stream, ev, 0));
            return XB_OK;
        }
        unsigned arrivals = 0;       // per member and group so far in this layer (a launch of k steps arrives k - 1 times)
        for (int i = 0; i < nts; ++i) {
            const int s0 = (int)((long long)T * i / nts), s1 = (int)((long long)T * (i + 1) / nts);
            {
                StageScope sc(ctx, XB_STAGE_LSTM_REC, i == 0 ? launches : 0);
                for (int n0 = 0; n0 < n; n0 += slab) {
                    p.n0 = n0; p.nslab = (n - n0) < slab ? (n - n0) : slab;
                    p.s_begin = s0; p.s_end = s1; p.persistent = 1;
                    // a tail slab that fits the single-group launch gets one workgroup per group (twice the CUs at work)
                    p.dual = dual_batch && (ctx->lstm_dual == 2 ? p.nslab > bn : p.nslab > gslab * bn);
                    p.quad = p.dual && quad_ok;
                    // counters are zeroed once per layer (above): consecutive launches follow each other without a memset in
                    // between, so the next launch's workgroups are dispatched the moment the previous one retires
                    p.grp0 = global_groups ? n0 / bn : 0;
                    p.slab = i; p.xcd_local = ctx->lstm_local && i < 16;   // 16 mask bytes per group slot
                    p.sync_base = global_groups ? arrivals : 0;
                    if (!global_groups) XB_HIP(ctx, hipMemsetAsync(ctx->sync, 0, sizeof(unsigned) * SYNC_SLOTS * 32, ctx->stream));
                    XB_HIP(ctx, xb::launch_lstm(p, ctx->stream));
                }
            }
            arrivals += (unsigned)(s1 - s0 - 1);
            if (overlapped && ctx->overlap == 1) {
                hipEvent_t ev;
                int rc = next_dep(ctx, &ev);
                if (rc) return rc;
                XB_HIP(ctx, hipEventRecord(ev, ctx->stream));
                XB_HIP(ctx, hipStreamWaitEvent(ctx->stream2, ev, 0));
                const int ta = p.reverse ? T - s1 : s0, tb = p.reverse ? T - s0 : s1;
                if ((rc = launch_row_gemm(ctx, *next, n, ta, tb, ctx->stream2, true))) return rc;
            }
        }
        if (overlapped && ctx->overlap == 1) {
            hipEvent_t ev;
            int rc = next_dep(ctx, &ev);
            if (rc) return rc;
            XB_HIP(ctx, hipEventRecord(ev, ctx->stream2));
            XB_HIP(ctx, hipStreamWaitEvent(ctx->stream, ev, 0));
        } else if (overlapped) {
            overlapped = false;     // XB_OVERLAP=2: the GEMM follows on the main stream
        }
    } else {
        if (n > 64 * bn) return fail(ctx, XB_ERR_INVALID, "one-launch-per-step LSTM mode handles at most %d chunks per batch", 64 * bn);
        StageScope sc(ctx, XB_STAGE_LSTM_REC, T);
        p.n0 = 0; p.nslab = n; p.persistent = 0;
        p.dual = dual_ok && ctx->lstm_dual == 2 && n > bn;      // tests only: the per-step variant of the dual kernel
        for (int s = 0; s < T; ++s) {
            p.s_begin = s; p.s_end = s + 1;
            XB_HIP(ctx, xb::launch_lstm(p, ctx->stream));
        }
    }
    if (next && !overlapped) return launch_row_gemm(ctx, *next, n, 0, T, ctx->stream);
    return XB_OK;
}

// scores_out: (T, n, ldc) with ldc given; expand selects the blank-column layout
int run_encoder(xb_ctx *ctx, const float *d_signal, int n, int expand, float *scores_out, int ldc,
                const float *d_signal2 = nullptr, int split = 0)
{
    const xb_config &c = ctx->cfg;
    const int F = c.features, T = ctx->T;
    const int nsplit = ctx->ns_conv;
    ctx->dep_next = 0;
    {
        StageScope sc(ctx, XB_STAGE_CONV, 2);
        xb::ConvFrontParams cf{};
        cf.signal = d_signal; cf.signal2 = d_signal2; cf.split = d_signal2 ? split : n; cf.N = n; cf.L = c.chunk_len; cf.T = T; cf.winlen = c.winlen; cf.stride = c.stride;
        cf.kp = ctx->kp; cf.w1 = ctx->w1; cf.b1 = ctx->b1; cf.w2 = ctx->w2; cf.b2 = ctx->b2;
        cf.a_hi = ctx->im_hi; cf.a_lo = ctx->im_lo; cf.q8 = nsplit == 2;
        XB_HIP(ctx, xb::launch_conv_front(cf, ctx->stream));
        xb::GemmParams g{};
        g.a_hi = ctx->im_hi; g.a_lo = ctx->im_lo; g.b_hi = ctx->w3_hi; g.b_lo = ctx->w3_lo;
        g.M = T * n; g.Nn = F; g.K = ctx->kp; g.lda = ctx->kp; g.ldb = ctx->kp;
        g.bias = ctx->b3; g.out_hi = ctx->x_hi[0]; g.out_lo = ctx->x_lo[0]; g.ldc = F; g.nsplit = nsplit;
        g.a_exp = 0; g.b_exp = ctx->w3_exp; g.out_exp = 0;
        g.out_fmt = second_part(ctx->ns_in[0]);            // (0: hi only is read -- the GEMM's own form)
        if (ctx->gemm4) { g.b4 = ctx->w3_f4; g.b4_kstride = ctx->w3_ks; }
        XB_HIP(ctx, xb::launch_gemm(g, xb::EPI_SILU_SPLIT, ctx->stream));
    }
    // layer l reads gin[l & 1] while the next layer's input projection is written into the other buffer
    float *gin[2] = {ctx->gin, ctx->gin2 ? ctx->gin2 : ctx->gin};
    int cur = 0;
    {
        const NextGemm first{0, ctx->x_hi[0], ctx->x_lo[0], gin[0], 4 * F, 0};
        int rc = launch_row_gemm(ctx, first, n, 0, T, ctx->stream);
        if (rc) return rc;
    }
    for (int l = 0; l < 5; ++l) {
        NextGemm next{l + 1, ctx->x_hi[cur ^ 1], ctx->x_lo[cur ^ 1], l < 4 ? gin[(l + 1) & 1] : scores_out,
                      l < 4 ? 4 * F : ldc, expand};
        int rc = run_lstm_layer(ctx, l, n, gin[l & 1], ctx->x_hi[cur ^ 1], ctx->x_lo[cur ^ 1], &next);
        if (rc) return rc;
        cur ^= 1;
    }
    return XB_OK;
}

// optional outputs of the Log scans (xb_crf_logz / xb_crf_scans)
struct ScanOut {
    float *alpha = nullptr, *beta = nullptr, *logz = nullptr, *post = nullptr;   // device; post has row stride ldq
};

int run_decode(xb_ctx *ctx, const float *d_scores, int T, int n, int has_blank, int ld, const char *alphabet,
               int8_t *d_labels, int8_t *d_seq, int32_t *d_len, hipStream_t st = nullptr, const ScanOut *scan = nullptr)
{
    if (!st) st = ctx->stream;
    const xb_config &c = ctx->cfg;
    if (T < 1 || T > ctx->T) return fail(ctx, XB_ERR_INVALID, "T=%d outside [1, %d]", T, ctx->T);
    xb::DecodeParams p{};
    p.scores = d_scores; p.T = T; p.N = n; p.S = ctx->S; p.hi = ctx->hi; p.nb = c.n_base;
    p.cin = has_blank ? ctx->S * (c.n_base + 1) : ctx->S * c.n_base;
    p.ld = ld; p.has_blank = has_blank; p.blank = c.blank_score;
    p.alpha = ctx->alpha; p.beta = ctx->beta; p.bmax = ctx->bmax;
    p.qbuf = ctx->qbuf; p.ldq = (ctx->S * (c.n_base + 1) + 3) & ~3;
    if (scan) {
        if (scan->alpha) p.alpha = scan->alpha;
        p.logz = scan->logz;
        p.beta_out = scan->beta;
        if (scan->post) { p.qbuf = scan->post; p.post_mode = 1; }
        p.stop_after = (scan->beta || scan->post) ? 2 : 1;
    }
    p.labels = d_labels; p.seq = d_seq; p.seq_len = d_len;
    memset(p.alphabet, 0, sizeof p.alphabet);
    if (alphabet) {
        if ((int)strlen(alphabet) < c.n_base + 1) return fail(ctx, XB_ERR_INVALID, "alphabet needs %d symbols", c.n_base + 1);
        memcpy(p.alphabet, alphabet, (size_t)c.n_base + 1);
    } else if (d_seq) {
        return fail(ctx, XB_ERR_INVALID, "alphabet is required when seq is requested");
    }
#ifdef XB_LSTM_STAMPS
    if (const char *e = getenv("XB_DECODE_STOP")) p.debug_stop = atoi(e);
    if (const char *e = getenv("XB_DECODE_LINEAR_LDS")) p.debug_lds = atoi(e);
#endif
    StageScope sc(ctx, XB_STAGE_DECODE, 1, st);
    hipError_t e = xb::launch_crf_decode(p, st);
    if (e != hipSuccess) return fail(ctx, e == hipErrorInvalidValue ? XB_ERR_INVALID : XB_ERR_HIP, "crf decode launch failed: %s", hipGetErrorString(e));
    return XB_OK;
}

// every entry point except the asynchronous basecall first orders the main stream behind decodes still in flight on the
// third stream (they share the decode workspaces and the score buffers)
int join_async_decode(xb_ctx *ctx)
{
    if (int rc = flush_held(ctx)) return rc;
    for (int p = 0; p < 2; ++p)
        if (ctx->dec_pending[p]) {
            XB_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->dec_done[p], 0));
            ctx->dec_pending[p] = false;
        }
    return XB_OK;
}

int sync_all(xb_ctx *ctx)
{
    XB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->stream2) XB_HIP(ctx, hipStreamSynchronize(ctx->stream2));
    if (ctx->stream3) XB_HIP(ctx, hipStreamSynchronize(ctx->stream3));
    ctx->dec_pending[0] = ctx->dec_pending[1] = false;
    return XB_OK;
}

}  // namespace

// ===========================================================================================
extern "C" {

XB_API const char *xb_version(void) { return "xnacall 0.1.0 (gfx950)"; }

XB_API int xb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

XB_API const char *xb_last_error(const xb_ctx *ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

XB_API int xb_ctx_create(xb_ctx **out, int device, const xb_config *cfg)
{
    if (!out || !cfg) return fail(nullptr, XB_ERR_INVALID, "null argument");
    *out = nullptr;
    if (cfg->n_base < 4 || cfg->n_base > 6) return fail(nullptr, XB_ERR_INVALID, "n_base %d not in {4,5,6}", cfg->n_base);
    if (cfg->state_len < 2 || cfg->state_len > 5) return fail(nullptr, XB_ERR_INVALID, "state_len %d not in [2,5]", cfg->state_len);
    const int64_t S = ipow(cfg->n_base, cfg->state_len);
    if (S > 1024) return fail(nullptr, XB_ERR_INVALID, "n_base^state_len = %lld exceeds 1024 states", (long long)S);
    if (!xb::lstm_supported_features(cfg->features))
        return fail(nullptr, XB_ERR_INVALID, "features %d unsupported (32,64,96,128,256,384,512,768)", cfg->features);
    if (cfg->winlen < 1 || cfg->winlen > 31 || cfg->winlen % 2 == 0 || cfg->stride < 1 || cfg->stride > 8)
        return fail(nullptr, XB_ERR_INVALID, "winlen %d / stride %d unsupported", cfg->winlen, cfg->stride);
    if (cfg->chunk_len < cfg->stride || cfg->max_batch < 1) return fail(nullptr, XB_ERR_INVALID, "bad chunk_len/max_batch");
    if (cfg->precision < XB_PREC_F16X3 || cfg->precision > XB_PREC_MIXED) return fail(nullptr, XB_ERR_INVALID, "bad precision");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(nullptr, XB_ERR_NO_GPU, "no HIP device available");
    if (device < 0 || device >= ndev) return fail(nullptr, XB_ERR_INVALID, "device %d out of range (%d devices)", device, ndev);

    xb_ctx *ctx = new (std::nothrow) xb_ctx();
    if (!ctx) return fail(nullptr, XB_ERR_NOMEM, "out of host memory");
    ctx->cfg = *cfg;
    ctx->device = device;
    const int pad = cfg->winlen / 2;
    ctx->T = (cfg->chunk_len + 2 * pad - cfg->winlen) / cfg->stride + 1;
    ctx->S = (int)S;
    ctx->hi = (int)ipow(cfg->n_base, cfg->state_len - 1);
    ctx->O = (int)(S * cfg->n_base);
    ctx->kp = (16 * cfg->winlen + 31) & ~31;
    ctx->ld_nb = (ctx->O + 3) & ~3;
    ctx->lstm_mode = cfg->lstm_mode;
    {
        int x3 = cfg->precision == XB_PREC_MIXED ? X3_MIXED_STAGES : 0;
        if (const char *e = getenv("XB_X3_STAGES")) x3 = (int)strtol(e, nullptr, 0) & 0xfff;      // diagnostic (tools/x3_stages.py)
        set_stage_arithmetic(ctx, x3);
    }
    if (const char *e = getenv("XB_LSTM_MODE")) ctx->lstm_mode = atoi(e);
    if (const char *e = getenv("XB_LSTM_DUAL")) ctx->lstm_dual = atoi(e);
    if (const char *e = getenv("XB_LSTM_WIDE")) ctx->lstm_wide = atoi(e) != 0;
    if (const char *e = getenv("XB_LSTM_QUAD")) ctx->lstm_quad = atoi(e) != 0;
    if (const char *e = getenv("XB_LSTM_LOCAL")) ctx->lstm_local = atoi(e) != 0;
    if (const char *e = getenv("XB_DECODE_ASYNC")) ctx->decode_async = atoi(e) != 0;
    if (const char *e = getenv("XB_IN1_LAYERS")) ctx->in1_layers = atoi(e) & 31;
    if (const char *e = getenv("XB_LSTM_I8")) ctx->lstm_i8 = atoi(e) == 2 ? 2 : (atoi(e) != 0);
    if (const char *e = getenv("XB_GEMM4")) ctx->gemm4 = atoi(e) != 0;
    if (const char *e = getenv("XB_GEMM_SN")) ctx->gemm_sn = atoi(e) > 0 && atoi(e) <= 64 ? atoi(e) : 0;
    if (const char *e = getenv("XB_GEMM_SHADOW_WGS")) ctx->gemm_shadow_wgs = atoi(e) == 1 ? 1 : 2;
    if (const char *e = getenv("XB_GEMM_SHADOW")) ctx->gemm_shadow_kernel = atoi(e) == 4 ? 4 : (atoi(e) == 8 ? 8 : 0);

#define XB_CREATE_HIP(call)                                                                   \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            int rc_ = fail(nullptr, XB_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
            xb_ctx_destroy(ctx);                                                              \
            return rc_;                                                                       \
        }                                                                                     \
    } while (0)
    XB_CREATE_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    XB_CREATE_HIP(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        int rc = fail(nullptr, XB_ERR_NO_GPU, "device %d is %s, this library is built for gfx950 only", device, prop.gcnArchName);
        xb_ctx_destroy(ctx);
        return rc;
    }
    ctx->cu_count = prop.multiProcessorCount;
    {
        int least = 0, greatest = 0;
        XB_CREATE_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        XB_CREATE_HIP(hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, greatest));
        XB_CREATE_HIP(hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, least));
        // (a high-priority decode stream was measured: no difference, 6.64 vs 6.69 ms per decode, 125.0 vs 124.7 ms per step)
        XB_CREATE_HIP(hipStreamCreateWithPriority(&ctx->stream3, hipStreamNonBlocking, least));
        for (int p = 0; p < 2; ++p) XB_CREATE_HIP(hipEventCreateWithFlags(&ctx->dec_done[p], hipEventDisableTiming));
        if (const char *e = getenv("XB_OVERLAP")) ctx->overlap = atoi(e);   // 0 serial, 1 overlapped, 2 time slabs but serial GEMM (A/B)
        if (const char *e = getenv("XB_TIME_SLABS")) ctx->time_slabs = atoi(e) > 0 ? atoi(e) : 1;
        if (const char *e = getenv("XB_SLAB_STEPS")) ctx->slab_steps = atoi(e) >= 8 ? atoi(e) : 0;
        if (const char *e = getenv("XB_LSTM_SIGNAL")) ctx->lstm_signal = atoi(e) < 0 || atoi(e) > 2 ? 2 : atoi(e);
        if (const char *e = getenv("XB_FUSE")) ctx->fuse_ok = atoi(e) != 0;
        // rocprofv3 counter collection (--pmc) runs one kernel at a time; hipStreamWaitValue32 is a spinning kernel
        // (__amd_rocclr_streamOpsWait) there, which would wait for a flag the serialised recurrence can never raise: slab launches
        {
            const char *cc = getenv("ROCPROF_COUNTER_COLLECTION"), *cn = getenv("ROCPROF_COUNTERS");
            if ((cc && atoi(cc) != 0) || (cn && *cn)) ctx->lstm_signal = 0;
        }
    }

    // co-scheduling two calls: only where the pair fits one launch of two groups per workgroup
    // (512 = the XCD-local capacity of a launch at features 768; with the wide placement -- run_lstm_layer -- a pair of up to
    //  2 x 640 chunks still is ONE launch of two groups per workgroup, so batch sizes 513..640 pair as well)
    int pair_cap = 512;
    {
        const int members = xb::lstm_members(cfg->features), slots = members > 0 ? ctx->cu_count / members : 0;
        if (ctx->lstm_wide && members > 0 && 8 * ((ctx->cu_count / 8) / members) * xb::lstm_group_chunks() == 512 && slots > 8)
            pair_cap = (slots > 64 ? 64 : slots) * xb::lstm_group_chunks();
    }
    if (!(ctx->fuse_ok && ctx->overlap == 1 && ctx->lstm_dual != 0 && cfg->max_batch <= pair_cap)) ctx->fuse_ok = 0;
    const size_t F = cfg->features;
    int rc = alloc_workspaces(ctx, cfg->max_batch);
    rc = rc ? rc : dev_alloc(ctx, &ctx->xh, (size_t)64 * 2 * 2 * 64 * F);
    rc = rc ? rc : dev_alloc(ctx, &ctx->sync, (size_t)SYNC_SLOTS * 32 + 32 + 64);      // group slots, error word, slab arrival counters
    if (rc) {
        g_create_error = ctx->err;
        xb_ctx_destroy(ctx);
        return rc;
    }
    ctx->error = ctx->sync + SYNC_SLOTS * 32;
    XB_CREATE_HIP(hipMemset(ctx->sync, 0, sizeof(unsigned) * (SYNC_SLOTS * 32 + 32 + 64)));
    ctx->sig_done = ctx->sync + SYNC_SLOTS * 32 + 32;
    if (ctx->lstm_signal) {
        int can = 0;
        if (hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, device) != hipSuccess || !can) {
            ctx->lstm_signal = 0;
        } else {
            void *fp = nullptr;
            if (hipExtMallocWithFlags(&fp, 8, hipMallocSignalMemory) != hipSuccess) {
                (void)hipGetLastError();
                if (hipMalloc(&fp, 8) != hipSuccess) { (void)hipGetLastError(); fp = nullptr; }
            }
            if (fp) {
                ctx->bufs.push_back({fp, 8});
                ctx->sig_flag = static_cast<unsigned *>(fp);
                XB_CREATE_HIP(hipMemset(fp, 0, 8));
            } else {
                ctx->lstm_signal = 0;
            }
        }
    }
#undef XB_CREATE_HIP
    *out = ctx;
    return XB_OK;
}

XB_API void xb_ctx_destroy(xb_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    // a call nobody waited for: launch it all the same -- its deferred gather (xb_gather_called) is a collective the other ranks
    // enter too, and the streams are drained below before anything is freed
    if (ctx->holding && ctx->weights_ready && ctx->cap > 0) (void)flush_held(ctx);
    ctx->holding = false;
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    if (ctx->stream3) (void)hipStreamSynchronize(ctx->stream3);
    for (auto &e : ctx->dec_done) if (e) (void)hipEventDestroy(e);
    for (auto &sl : ctx->slots) {
        if (sl.h_signal) (void)hipHostFree(sl.h_signal);
        if (sl.h_seq) (void)hipHostFree(sl.h_seq);
        if (sl.h_len) (void)hipHostFree(sl.h_len);
        if (sl.h_err) (void)hipHostFree(sl.h_err);
        if (sl.h2d) (void)hipEventDestroy(sl.h2d);
        if (sl.done) (void)hipEventDestroy(sl.done);
    }
    if (ctx->stream_copy) { (void)hipStreamSynchronize(ctx->stream_copy); (void)hipStreamDestroy(ctx->stream_copy); }
    for (auto &e : ctx->deps) (void)hipEventDestroy(e);
    for (auto &ev : ctx->events) { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    for (auto &b : ctx->bufs) (void)hipFree(b.p);
    for (auto &b : ctx->wsbufs) (void)hipFree(b.p);
    for (void *w : ctx->wbufs) (void)hipFree(w);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
    delete ctx;
}

XB_API int xb_load_weights(xb_ctx *ctx, const char *name, const float *host, int64_t n)
{
    if (!ctx || !name || !host) return fail(ctx, XB_ERR_INVALID, "null argument");
    const int64_t want = expected_size(ctx, name);
    if (want < 0) return fail(ctx, XB_ERR_INVALID, "unknown state-dict key '%s'", name);
    if (want != n) return fail(ctx, XB_ERR_INVALID, "'%s': got %lld elements, config implies %lld", name, (long long)n, (long long)want);
    // a held-back basecall belongs to the weights it was called with (loading clears weights_ready, the launch needs it)
    if (ctx->holding) {
        XB_HIP(ctx, hipSetDevice(ctx->device));
        if (int rc = flush_held(ctx)) return rc;
    }
    ctx->host_w[name].assign(host, host + n);
    ctx->weights_ready = false;
    return XB_OK;
}

XB_API int xb_weights_ready(xb_ctx *ctx)
{
    if (!ctx) return XB_ERR_INVALID;
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcf = flush_held(ctx)) return rcf;          // with the weights it was called with
    const int F = ctx->cfg.features, W = ctx->cfg.winlen;
    auto need = [&](const std::string &k) -> const std::vector<float> * {
        auto it = ctx->host_w.find(k);
        return it == ctx->host_w.end() ? nullptr : &it->second;
    };
    std::vector<std::string> keys = {"encoder.0.conv.weight", "encoder.0.conv.bias", "encoder.1.conv.weight",
                                     "encoder.1.conv.bias", "encoder.2.conv.weight", "encoder.2.conv.bias",
                                     "encoder.9.linear.weight", "encoder.9.linear.bias"};
    for (int l = 4; l <= 8; ++l)
        for (const char *s : {"weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"})
            keys.push_back("encoder." + std::to_string(l) + ".rnn." + s);
    for (auto &k : keys)
        if (!need(k)) return fail(ctx, XB_ERR_STATE, "missing tensor '%s'", k.c_str());
    // a second load_state_dict on a live context: nothing may still be reading the previous weight set
    if (!ctx->wbufs.empty()) {
        if (int rcs = sync_all(ctx)) return rcs;
        for (void *w : ctx->wbufs) (void)hipFree(w);
        ctx->wbufs.clear();
    }

    int rc;
    if ((rc = upload(ctx, &ctx->w1, *need("encoder.0.conv.weight")))) return rc;
    if ((rc = upload(ctx, &ctx->b1, *need("encoder.0.conv.bias")))) return rc;
    if ((rc = upload(ctx, &ctx->w2, *need("encoder.1.conv.weight")))) return rc;
    if ((rc = upload(ctx, &ctx->b2, *need("encoder.1.conv.bias")))) return rc;
    if ((rc = upload(ctx, &ctx->b3, *need("encoder.2.conv.bias")))) return rc;
    std::vector<half_t> hi, lo;
    // every weight tensor in the form its stage's arithmetic reads (q8 image for nsplit 2, fp16 residual for 3)
    split_rows(need("encoder.2.conv.weight")->data(), F, 16 * W, ctx->kp, hi, lo, ctx->ns_conv == 2 ? &ctx->w3_exp : nullptr);
    if ((rc = upload(ctx, &ctx->w3_hi, hi))) return rc;
    if ((rc = upload(ctx, &ctx->w3_lo, lo))) return rc;
    std::vector<unsigned char> f4;
    fragment_major(hi, lo, F, ctx->kp, ctx->kp, ctx->ns_conv, f4, &ctx->w3_ks);
    if ((rc = upload(ctx, &ctx->w3_f4, f4))) return rc;
    for (int l = 0; l < 5; ++l) {
        const std::string pre = "encoder." + std::to_string(4 + l) + ".rnn.";
        const float *wih = need(pre + "weight_ih_l0")->data(), *whh = need(pre + "weight_hh_l0")->data();
        const float *bih = need(pre + "bias_ih_l0")->data(), *bhh = need(pre + "bias_hh_l0")->data();
        // gate-interleaved row order: row' = unit*4 + gate  <-  row = gate*F + unit  (gates i,f,g,o)
        std::vector<float> wi((size_t)4 * F * F), wh((size_t)4 * F * F), bb((size_t)4 * F);
        for (int u = 0; u < F; ++u)
            for (int q = 0; q < 4; ++q) {
                memcpy(&wi[((size_t)u * 4 + q) * F], &wih[((size_t)q * F + u) * F], sizeof(float) * F);
                memcpy(&wh[((size_t)u * 4 + q) * F], &whh[((size_t)q * F + u) * F], sizeof(float) * F);
                bb[(size_t)u * 4 + q] = bih[(size_t)q * F + u] + bhh[(size_t)q * F + u];
            }
        split_rows(wi.data(), 4 * F, F, F, hi, lo, ctx->ns_in[l] == 2 ? &ctx->wih_exp[l] : nullptr);
        if ((rc = upload(ctx, &ctx->wih_hi[l], hi))) return rc;
        if ((rc = upload(ctx, &ctx->wih_lo[l], lo))) return rc;
        fragment_major(hi, lo, 4 * F, F, F, ctx->ns_in[l], f4, &ctx->wih_ks);       // (the k-tile stride is the same for nsplit 2 and 3)
        if ((rc = upload(ctx, &ctx->wih_f4[l], f4))) return rc;
        if (ctx->cfg.precision == XB_PREC_F16F8_IN1) {
            fragment_major(hi, lo, 4 * F, F, F, 1, f4, &ctx->wih_ksh);
            if ((rc = upload(ctx, &ctx->wih_f4h[l], f4))) return rc;
        }
        split_rows(wh.data(), 4 * F, F, F, hi, lo, ctx->ns_rec[l] == 2 ? &ctx->whh_exp[l] : nullptr);
        if ((rc = upload(ctx, &ctx->whh_hi[l], hi))) return rc;
        if ((rc = upload(ctx, &ctx->whh_lo[l], lo))) return rc;
        ctx->whh_q1[l] = nullptr;
        if (ctx->lstm_i8 && ctx->ns_rec[l] == 2 && (F == 64 || F % 128 == 0)) {
            // int8-limb image: per row q = round(W / s * 32512), s = max |W| of the row; q = 256 d1 + d0 with both digits in
            // [-128, 127]; h is published as round(h * 32512) the same way, so W h = s / 32512^2 * sum q_w q_h
            std::vector<int8_t> d1((size_t)4 * F * F), d0((size_t)4 * F * F);
            std::vector<float> sc((size_t)4 * F);
            for (int r = 0; r < 4 * F; ++r) {
                float mx = 0.0f;
                for (int k = 0; k < F; ++k) mx = std::max(mx, std::fabs(wh[(size_t)r * F + k]));
                const float sr = mx > 0.0f ? mx : 1.0f;
                sc[r] = sr / (32512.0f * 32512.0f);
                for (int k = 0; k < F; ++k) {
                    const int q = (int)std::lrintf(wh[(size_t)r * F + k] / sr * 32512.0f);
                    const int lo8 = ((q + 128) & 255) - 128;
                    d0[(size_t)r * F + k] = (int8_t)lo8;
                    d1[(size_t)r * F + k] = (int8_t)((q - lo8) >> 8);
                }
            }
            if ((rc = upload(ctx, &ctx->whh_q1[l], d1))) return rc;
            if ((rc = upload(ctx, &ctx->whh_q0[l], d0))) return rc;
            if ((rc = upload(ctx, &ctx->whh_sc[l], sc))) return rc;
        }
        if ((rc = upload(ctx, &ctx->lbias[l], bb))) return rc;
    }
    split_rows(need("encoder.9.linear.weight")->data(), ctx->O, F, F, hi, lo, ctx->ns_lin == 2 ? &ctx->wl_exp : nullptr);
    if ((rc = upload(ctx, &ctx->wl_hi, hi))) return rc;
    if ((rc = upload(ctx, &ctx->wl_lo, lo))) return rc;
    fragment_major(hi, lo, ctx->O, F, F, ctx->ns_lin, f4, &ctx->wl_ks);
    if ((rc = upload(ctx, &ctx->wl_f4, f4))) return rc;
    if ((rc = upload(ctx, &ctx->bl, *need("encoder.9.linear.bias")))) return rc;
    ctx->host_w.clear();
    ctx->weights_ready = true;
    return XB_OK;
}

XB_API int xb_synchronize(xb_ctx *ctx)
{
    if (!ctx) return XB_ERR_INVALID;
    XB_HIP(ctx, hipSetDevice(ctx->device));
    int rc = flush_held(ctx);
    if (!rc && ctx->deferred_rc) rc = ctx->deferred_rc;
    ctx->deferred_rc = 0;
    if (rc) return rc;
    rc = sync_all(ctx);
    if (rc) return rc;
    return check_device_error(ctx);
}

XB_API void *xb_result_stream(xb_ctx *ctx)
{
    if (!ctx) return nullptr;
    (void)hipSetDevice(ctx->device);
    (void)flush_held(ctx);        // a failure is recorded (pipeline_failed, xb_last_error) and reported by the next call
    return ctx->result_stream ? ctx->result_stream : ctx->stream;
}

// xb_comm: run fn(arg) right behind the held-back call that will write d_seq, instead of now (returns 1), or report that no
// such call is held (0: the caller proceeds at once).  Not part of the public header.
extern "C" __attribute__((visibility("default"))) int xb_internal_defer_after(xb_ctx *ctx, const void *d_seq, void (*fn)(void *), void *arg)
{
    if (!ctx || !ctx->holding || ctx->held.seq != d_seq || ctx->held.after) return 0;
    ctx->held.after = fn;
    ctx->held.after_arg = arg;
    return 1;
}

XB_API int xb_stream_wait_event(xb_ctx *ctx, void *hip_event)
{
    if (!ctx || !hip_event) return XB_ERR_INVALID;
    XB_HIP(ctx, hipSetDevice(ctx->device));
    hipEvent_t ev = static_cast<hipEvent_t>(hip_event);
    XB_HIP(ctx, hipStreamWaitEvent(ctx->stream, ev, 0));
    if (ctx->stream3) XB_HIP(ctx, hipStreamWaitEvent(ctx->stream3, ev, 0));
    return XB_OK;
}

XB_API int xb_encode_dev(xb_ctx *ctx, const float *d_signal, int n, int expand_blanks, float *d_scores)
{
    int rc = check_ready(ctx, n);
    if (rc) return rc;
    if (!d_signal || !d_scores) return fail(ctx, XB_ERR_INVALID, "null device pointer");
    XB_HIP(ctx, hipSetDevice(ctx->device));
    ctx->result_stream = ctx->stream;
    if ((rc = join_async_decode(ctx))) return rc;
    const int ldc = expand_blanks ? ctx->S * (ctx->cfg.n_base + 1) : ctx->O;
    return run_encoder(ctx, d_signal, n, expand_blanks ? 1 : 0, d_scores, ldc);
}

XB_API int xb_encode(xb_ctx *ctx, const float *signal, int n, int expand_blanks, float *scores)
{
    int rc = check_ready(ctx, n);
    if (rc) return rc;
    if (!signal || !scores) return fail(ctx, XB_ERR_INVALID, "null host pointer");
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = join_async_decode(ctx))) return rc;
    XB_HIP(ctx, hipMemcpyAsync(ctx->d_signal, signal, sizeof(float) * (size_t)n * ctx->cfg.chunk_len,
                               hipMemcpyHostToDevice, ctx->stream));
    const int ldc = expand_blanks ? ctx->S * (ctx->cfg.n_base + 1) : ctx->O;
    rc = run_encoder(ctx, ctx->d_signal, n, expand_blanks ? 1 : 0, ctx->scores, ldc);
    if (rc) return rc;
    XB_HIP(ctx, hipMemcpyAsync(scores, ctx->scores, sizeof(float) * (size_t)ctx->T * n * ldc, hipMemcpyDeviceToHost,
                               ctx->stream));
    return xb_synchronize(ctx);
}

XB_API int xb_decode_dev(xb_ctx *ctx, const float *d_scores, int T, int n, int has_blank, const char *alphabet,
                         int8_t *d_labels, int8_t *d_seq, int32_t *d_seq_len)
{
    if (!ctx) return XB_ERR_INVALID;
    if (n < 1 || n > ctx->cfg.max_batch) return fail(ctx, XB_ERR_INVALID, "batch %d outside [1, max_batch=%d]", n, ctx->cfg.max_batch);
    if (!d_scores) return fail(ctx, XB_ERR_INVALID, "null device pointer");
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = join_async_decode(ctx)) return rc;
    ctx->result_stream = ctx->stream;
    const int ld = has_blank ? ctx->S * (ctx->cfg.n_base + 1) : ctx->O;
    return run_decode(ctx, d_scores, T, n, has_blank ? 1 : 0, ld, alphabet, d_labels, d_seq, d_seq_len);
}

XB_API int xb_decode(xb_ctx *ctx, const float *scores, int T, int n, int has_blank, const char *alphabet,
                     int8_t *labels, int8_t *seq, int32_t *seq_len)
{
    if (!ctx) return XB_ERR_INVALID;
    if (n < 1 || n > ctx->cfg.max_batch) return fail(ctx, XB_ERR_INVALID, "batch %d outside [1, max_batch=%d]", n, ctx->cfg.max_batch);
    if (!scores) return fail(ctx, XB_ERR_INVALID, "null host pointer");
    if (T < 1 || T > ctx->T) return fail(ctx, XB_ERR_INVALID, "T=%d outside [1, %d]", T, ctx->T);
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcj = join_async_decode(ctx)) return rcj;
    const int ld = has_blank ? ctx->S * (ctx->cfg.n_base + 1) : ctx->O;
    XB_HIP(ctx, hipMemcpyAsync(ctx->scores, scores, sizeof(float) * (size_t)T * n * ld, hipMemcpyHostToDevice, ctx->stream));
    int rc = run_decode(ctx, ctx->scores, T, n, has_blank ? 1 : 0, ld, alphabet, ctx->labels, seq ? ctx->seq : nullptr,
                        ctx->seq_len);
    if (rc) return rc;
    if (labels) XB_HIP(ctx, hipMemcpyAsync(labels, ctx->labels, (size_t)n * T, hipMemcpyDeviceToHost, ctx->stream));
    if (seq) XB_HIP(ctx, hipMemcpyAsync(seq, ctx->seq, (size_t)n * T, hipMemcpyDeviceToHost, ctx->stream));
    if (seq_len) XB_HIP(ctx, hipMemcpyAsync(seq_len, ctx->seq_len, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    return xb_synchronize(ctx);
}

XB_API int xb_crf_scans_dev(xb_ctx *ctx, const float *d_scores, int T, int n, int has_blank, float *d_alpha, float *d_beta,
                            float *d_logz, float *d_post)
{
    if (!ctx) return XB_ERR_INVALID;
    if (n < 1 || n > ctx->cfg.max_batch) return fail(ctx, XB_ERR_INVALID, "batch %d outside [1, max_batch=%d]", n, ctx->cfg.max_batch);
    if (!d_scores) return fail(ctx, XB_ERR_INVALID, "null device pointer");
    if (!d_alpha && !d_beta && !d_logz && !d_post) return fail(ctx, XB_ERR_INVALID, "no output requested");
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = join_async_decode(ctx)) return rc;
    const int ld = has_blank ? ctx->S * (ctx->cfg.n_base + 1) : ctx->O;
    ScanOut so;
    so.alpha = d_alpha; so.beta = d_beta; so.logz = d_logz; so.post = d_post;
    ctx->result_stream = ctx->stream;       // the scans run on the main stream (not on the async decode stream)
    return run_decode(ctx, d_scores, T, n, has_blank ? 1 : 0, ld, nullptr, nullptr, nullptr, nullptr, nullptr, &so);
}

XB_API int xb_crf_scans(xb_ctx *ctx, const float *scores, int T, int n, int has_blank, float *alpha, float *beta, float *logz,
                        float *post)
{
    if (!ctx) return XB_ERR_INVALID;
    if (n < 1 || n > ctx->cfg.max_batch) return fail(ctx, XB_ERR_INVALID, "batch %d outside [1, max_batch=%d]", n, ctx->cfg.max_batch);
    if (!scores) return fail(ctx, XB_ERR_INVALID, "null host pointer");
    if (!alpha && !beta && !logz && !post) return fail(ctx, XB_ERR_INVALID, "no output requested");
    if (T < 1 || T > ctx->T) return fail(ctx, XB_ERR_INVALID, "T=%d outside [1, %d]", T, ctx->T);
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcj = join_async_decode(ctx)) return rcj;
    const int S = ctx->S, E = ctx->cfg.n_base + 1;
    const int ld = has_blank ? S * E : ctx->O;
    XB_HIP(ctx, hipMemcpyAsync(ctx->scores, scores, sizeof(float) * (size_t)T * n * ld, hipMemcpyHostToDevice, ctx->stream));
    // device staging in the decode's own workspaces: alpha -> its stash, beta -> the (otherwise unused) beta stash, P -> the
    // Q buffer (row stride ldq), logZ -> its own (max_batch) buffer
    const int ldq = (S * E + 3) & ~3;
    ScanOut so;
    so.alpha = ctx->alpha; so.beta = beta ? ctx->beta : nullptr; so.post = post ? ctx->qbuf : nullptr;
    so.logz = logz ? ctx->logz : nullptr;
    int rc = run_decode(ctx, ctx->scores, T, n, has_blank ? 1 : 0, ld, nullptr, nullptr, nullptr, nullptr, nullptr, &so);
    if (rc) return rc;
    const size_t sv = sizeof(float) * (size_t)(T + 1) * n * S;
    if (alpha) XB_HIP(ctx, hipMemcpyAsync(alpha, ctx->alpha, sv, hipMemcpyDeviceToHost, ctx->stream));
    if (beta) XB_HIP(ctx, hipMemcpyAsync(beta, ctx->beta, sv, hipMemcpyDeviceToHost, ctx->stream));
    if (logz) XB_HIP(ctx, hipMemcpyAsync(logz, so.logz, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    if (post)
        XB_HIP(ctx, hipMemcpy2DAsync(post, sizeof(float) * (size_t)S * E, ctx->qbuf, sizeof(float) * (size_t)ldq,
                                     sizeof(float) * (size_t)S * E, (size_t)T * n, hipMemcpyDeviceToHost, ctx->stream));
    return xb_synchronize(ctx);
}

XB_API int xb_crf_logz_dev(xb_ctx *ctx, const float *d_scores, int T, int n, int has_blank, float *d_logz)
{
    return xb_crf_scans_dev(ctx, d_scores, T, n, has_blank, nullptr, nullptr, d_logz, nullptr);
}

XB_API int xb_crf_logz(xb_ctx *ctx, const float *scores, int T, int n, int has_blank, float *logz)
{
    return xb_crf_scans(ctx, scores, T, n, has_blank, nullptr, nullptr, logz, nullptr);
}

// prepare_ctc_scores' gather columns (crf/model.py:102-116) for n targets of Lt labels: np = Lt - (sl - 1) positions
static void ctc_indices(const int32_t *targets, int n, int Lt, int nb, int sl, std::vector<int32_t> &stay, std::vector<int32_t> &move)
{
    const int np = Lt - (sl - 1), E = nb + 1;
    stay.assign((size_t)n * np, 0);
    move.assign((size_t)n * (np > 1 ? np - 1 : 1), 0);
    for (int b = 0; b < n; ++b) {
        for (int l = 0; l < np; ++l) {
            int64_t st = 0;
            for (int i = 0; i < sl; ++i) {
                const int v = std::max(targets[(size_t)b * Lt + l + i] - 1, 0);        // torch.clamp(targets - 1, 0)
                st += (int64_t)v * ipow(nb, sl - i - 1);
            }
            stay[(size_t)b * np + l] = (int32_t)(st * E);
        }
        for (int l = 0; l + 1 < np; ++l)
            move[(size_t)b * (np - 1) + l] = stay[(size_t)b * np + l + 1] + std::max(targets[(size_t)b * Lt + l] - 1, 0) + 1;
    }
}

static int run_ctc(xb_ctx *ctx, const float *scores, int T, int n, const int32_t *targets, int Lt, const int32_t *tlen,
                   int semiring, float *logz, float *gstay, float *gmove)
{
    if (!ctx) return XB_ERR_INVALID;
    const int sl = ctx->cfg.state_len, nb = ctx->cfg.n_base, S = ctx->S, C = S * (nb + 1);
    if (n < 1 || n > ctx->cfg.max_batch) return fail(ctx, XB_ERR_INVALID, "batch %d outside [1, max_batch=%d]", n, ctx->cfg.max_batch);
    if (T < 1 || T > ctx->T) return fail(ctx, XB_ERR_INVALID, "T=%d outside [1, %d]", T, ctx->T);
    if (!scores || !targets || !tlen) return fail(ctx, XB_ERR_INVALID, "null host pointer");
    if (!logz && !gstay && !gmove) return fail(ctx, XB_ERR_INVALID, "no output requested");
    const int np = Lt - (sl - 1);
    if (np < 1 || np > xb::ctc_max_positions())
        return fail(ctx, XB_ERR_INVALID, "target width %d gives %d positions, supported: 1..%d", Lt, np, xb::ctc_max_positions());
    for (int b = 0; b < n; ++b) {
        if (tlen[b] < sl || tlen[b] > Lt)
            return fail(ctx, XB_ERR_INVALID, "target_lengths[%d] = %d outside [state_len = %d, %d]", b, tlen[b], sl, Lt);
        for (int l = 0; l < Lt; ++l)
            if (targets[(size_t)b * Lt + l] < 0 || targets[(size_t)b * Lt + l] > nb)
                return fail(ctx, XB_ERR_INVALID, "targets[%d][%d] = %d outside [0, n_base = %d]", b, l, targets[(size_t)b * Lt + l], nb);
    }
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcj = join_async_decode(ctx)) return rcj;
    std::vector<int32_t> stay, move;
    ctc_indices(targets, n, Lt, nb, sl, stay, move);
    const size_t nm = np > 1 ? np - 1 : 1;
    // per-call device scratch (a training-side operator: sizes follow the targets, not the context)
    int32_t *d_stay = nullptr, *d_move = nullptr, *d_len = nullptr;
    float *d_alpha = nullptr, *d_gs = nullptr, *d_gm = nullptr;
    int rc = XB_OK;
    auto cleanup = [&]() {
        (void)hipFree(d_stay); (void)hipFree(d_move); (void)hipFree(d_len); (void)hipFree(d_alpha); (void)hipFree(d_gs); (void)hipFree(d_gm);
    };
#define XB_CTC_HIP(call)                                                                       \
    do {                                                                                      \
        hipError_t e_ = (call);                                                               \
        if (e_ != hipSuccess) {                                                               \
            rc = fail(ctx, e_ == hipErrorOutOfMemory ? XB_ERR_NOMEM : XB_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
            cleanup();                                                                        \
            return rc;                                                                        \
        }                                                                                     \
    } while (0)
    const bool grads = gstay || gmove;
    XB_CTC_HIP(hipMalloc(reinterpret_cast<void **>(&d_stay), sizeof(int32_t) * stay.size()));
    XB_CTC_HIP(hipMalloc(reinterpret_cast<void **>(&d_move), sizeof(int32_t) * move.size()));
    XB_CTC_HIP(hipMalloc(reinterpret_cast<void **>(&d_len), sizeof(int32_t) * (size_t)n));
    if (grads) {
        XB_CTC_HIP(hipMalloc(reinterpret_cast<void **>(&d_alpha), sizeof(float) * (size_t)n * (T + 1) * np));
        XB_CTC_HIP(hipMalloc(reinterpret_cast<void **>(&d_gs), sizeof(float) * (size_t)T * n * np));
        if (gmove) XB_CTC_HIP(hipMalloc(reinterpret_cast<void **>(&d_gm), sizeof(float) * (size_t)T * n * nm));
    }
    hipStream_t st = ctx->stream;
    ctx->result_stream = st;
    XB_CTC_HIP(hipMemcpyAsync(d_stay, stay.data(), sizeof(int32_t) * stay.size(), hipMemcpyHostToDevice, st));
    XB_CTC_HIP(hipMemcpyAsync(d_move, move.data(), sizeof(int32_t) * move.size(), hipMemcpyHostToDevice, st));
    XB_CTC_HIP(hipMemcpyAsync(d_len, tlen, sizeof(int32_t) * (size_t)n, hipMemcpyHostToDevice, st));
    XB_CTC_HIP(hipMemcpyAsync(ctx->scores, scores, sizeof(float) * (size_t)T * n * C, hipMemcpyHostToDevice, st));
    if (d_gs) XB_CTC_HIP(hipMemsetAsync(d_gs, 0, sizeof(float) * (size_t)T * n * np, st));
    if (d_gm) XB_CTC_HIP(hipMemsetAsync(d_gm, 0, sizeof(float) * (size_t)T * n * nm, st));
    xb::CtcParams p{};
    p.scores = ctx->scores; p.T = T; p.N = n; p.C = C; p.stay_idx = d_stay; p.move_idx = d_move; p.n = np; p.tlen = d_len;
    p.sl = sl; p.semiring = semiring; p.alpha = d_alpha; p.logz = ctx->logz; p.gstay = d_gs; p.gmove = d_gm; p.error = ctx->error;
    XB_CTC_HIP(xb::launch_ctc_scan(p, st));
    if (logz) XB_CTC_HIP(hipMemcpyAsync(logz, ctx->logz, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, st));
    if (gstay) XB_CTC_HIP(hipMemcpyAsync(gstay, d_gs, sizeof(float) * (size_t)T * n * np, hipMemcpyDeviceToHost, st));
    if (gmove && np > 1) XB_CTC_HIP(hipMemcpyAsync(gmove, d_gm, sizeof(float) * (size_t)T * n * nm, hipMemcpyDeviceToHost, st));
    XB_CTC_HIP(hipStreamSynchronize(st));
#undef XB_CTC_HIP
    cleanup();
    return check_device_error(ctx);
}

XB_API int xb_ctc_logz(xb_ctx *ctx, const float *scores, int T, int n, const int32_t *targets, int Lt,
                       const int32_t *target_lengths, float *logz, float *gstay, float *gmove)
{
    return run_ctc(ctx, scores, T, n, targets, Lt, target_lengths, 0, logz, gstay, gmove);
}

XB_API int xb_ctc_alignments(xb_ctx *ctx, const float *scores, int T, int n, const int32_t *targets, int Lt,
                             const int32_t *target_lengths, float *alignments, float *max_score)
{
    if (!alignments) return fail(ctx, XB_ERR_INVALID, "null host pointer");
    return run_ctc(ctx, scores, T, n, targets, Lt, target_lengths, 1, max_score, alignments, nullptr);
}

// beam search over device-resident scores: the Log scans (alpha, beta, logZ) into the decode workspaces, then one wave per chunk
static int run_beam(xb_ctx *ctx, const float *d_scores, int T, int n, int has_blank, int ld, const char *alphabet, int beam_width,
                    float beam_cut, float qscale, float qoffset, int8_t *d_seq, int8_t *d_q, uint8_t *d_moves, float *d_score)
{
    const xb_config &c = ctx->cfg;
    if (beam_width < 1 || beam_width > xb::BEAM_MAX_WIDTH)
        return fail(ctx, XB_ERR_INVALID, "beam_width %d outside [1, %d]", beam_width, xb::BEAM_MAX_WIDTH);
    if (ctx->S > xb::BEAM_MAX_STATES) return fail(ctx, XB_ERR_INVALID, "beam search supports at most %d states", xb::BEAM_MAX_STATES);
    if (!alphabet || (int)strlen(alphabet) < c.n_base + 1) return fail(ctx, XB_ERR_INVALID, "alphabet needs %d symbols", c.n_base + 1);
    if (!ctx->beam_hist) {
        const size_t N = (size_t)c.max_batch, Tm = (size_t)ctx->T;
        int rc = dev_alloc(ctx, &ctx->beam_hist, N * (Tm + 1) * xb::BEAM_MAX_WIDTH);
        rc = rc ? rc : dev_alloc(ctx, &ctx->beam_path, N * Tm);
        rc = rc ? rc : dev_alloc(ctx, &ctx->beam_prob, N * Tm);
        rc = rc ? rc : dev_alloc(ctx, &ctx->beam_score, N);
        rc = rc ? rc : dev_alloc(ctx, &ctx->beam_seq, N * Tm);
        rc = rc ? rc : dev_alloc(ctx, &ctx->beam_q, N * Tm);
        rc = rc ? rc : dev_alloc(ctx, &ctx->beam_moves, N * Tm);
        if (rc) { ctx->beam_hist = nullptr; return rc; }
    }
    ScanOut so;
    so.alpha = ctx->alpha; so.beta = ctx->beta; so.logz = ctx->logz;
    int rc = run_decode(ctx, d_scores, T, n, has_blank, ld, nullptr, nullptr, nullptr, nullptr, nullptr, &so);
    if (rc) return rc;
    xb::BeamParams p{};
    p.scores = d_scores; p.ld = ld; p.has_blank = has_blank; p.blank = c.blank_score;
    p.alpha = ctx->alpha; p.beta = ctx->beta; p.logz = ctx->logz;
    p.T = T; p.N = n; p.S = ctx->S; p.nb = c.n_base; p.hi = ctx->hi;
    p.W = beam_width;
    p.log_cut = beam_cut > 0.0f ? (float)log((double)beam_cut) : 3.402823466e+38f;
    p.qscale = qscale; p.qoffset = qoffset;
    memset(p.base_chars, 0, sizeof p.base_chars);
    memcpy(p.base_chars, alphabet + 1, (size_t)c.n_base);
    p.hist = ctx->beam_hist; p.path = ctx->beam_path; p.prob = ctx->beam_prob;
    p.seq = d_seq ? d_seq : ctx->beam_seq; p.qstr = d_q ? d_q : ctx->beam_q; p.moves = d_moves ? d_moves : ctx->beam_moves;
    p.score = d_seq ? d_score : ctx->beam_score;
    StageScope sc(ctx, XB_STAGE_DECODE, 1, ctx->stream);
    hipError_t e = xb::launch_beam_search(p, ctx->stream);
    if (e != hipSuccess) return fail(ctx, e == hipErrorInvalidValue ? XB_ERR_INVALID : XB_ERR_HIP, "beam search launch failed: %s", hipGetErrorString(e));
    return XB_OK;
}

XB_API int xb_beam_search_dev(xb_ctx *ctx, const float *d_scores, int T, int n, int has_blank, const char *alphabet,
                              int beam_width, float beam_cut, float qscale, float qoffset, int8_t *d_sequence, int8_t *d_qstring,
                              uint8_t *d_moves, float *d_score)
{
    if (!ctx) return XB_ERR_INVALID;
    if (n < 1 || n > ctx->cfg.max_batch) return fail(ctx, XB_ERR_INVALID, "batch %d outside [1, max_batch=%d]", n, ctx->cfg.max_batch);
    if (!d_scores || !d_sequence || !d_qstring || !d_moves) return fail(ctx, XB_ERR_INVALID, "null device pointer");
    if (T < 1 || T > ctx->T) return fail(ctx, XB_ERR_INVALID, "T=%d outside [1, %d]", T, ctx->T);
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = join_async_decode(ctx)) return rc;
    ctx->result_stream = ctx->stream;
    const int ld = has_blank ? ctx->S * (ctx->cfg.n_base + 1) : ctx->O;
    return run_beam(ctx, d_scores, T, n, has_blank ? 1 : 0, ld, alphabet, beam_width, beam_cut, qscale, qoffset, d_sequence,
                    d_qstring, d_moves, d_score);
}

// the three (n, T) byte planes and the optional path scores back to the host
static int beam_results_to_host(xb_ctx *ctx, int T, int n, int8_t *sequence, int8_t *qstring, uint8_t *moves, float *score)
{
    const size_t nt = (size_t)n * T;
    XB_HIP(ctx, hipMemcpyAsync(sequence, ctx->beam_seq, nt, hipMemcpyDeviceToHost, ctx->stream));
    XB_HIP(ctx, hipMemcpyAsync(qstring, ctx->beam_q, nt, hipMemcpyDeviceToHost, ctx->stream));
    XB_HIP(ctx, hipMemcpyAsync(moves, ctx->beam_moves, nt, hipMemcpyDeviceToHost, ctx->stream));
    if (score) XB_HIP(ctx, hipMemcpyAsync(score, ctx->beam_score, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    return xb_synchronize(ctx);
}

XB_API int xb_beam_search(xb_ctx *ctx, const float *scores, int T, int n, int has_blank, const char *alphabet, int beam_width,
                          float beam_cut, float qscale, float qoffset, int8_t *sequence, int8_t *qstring, uint8_t *moves,
                          float *score)
{
    if (!ctx) return XB_ERR_INVALID;
    if (n < 1 || n > ctx->cfg.max_batch) return fail(ctx, XB_ERR_INVALID, "batch %d outside [1, max_batch=%d]", n, ctx->cfg.max_batch);
    if (!scores || !sequence || !qstring || !moves) return fail(ctx, XB_ERR_INVALID, "null host pointer");
    if (T < 1 || T > ctx->T) return fail(ctx, XB_ERR_INVALID, "T=%d outside [1, %d]", T, ctx->T);
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (int rcj = join_async_decode(ctx)) return rcj;
    ctx->result_stream = ctx->stream;
    const int ld = has_blank ? ctx->S * (ctx->cfg.n_base + 1) : ctx->O;
    XB_HIP(ctx, hipMemcpyAsync(ctx->scores, scores, sizeof(float) * (size_t)T * n * ld, hipMemcpyHostToDevice, ctx->stream));
    // the first call allocates the staging planes inside run_beam; pass them after the allocation
    int rc = run_beam(ctx, ctx->scores, T, n, has_blank ? 1 : 0, ld, alphabet, beam_width, beam_cut, qscale, qoffset, nullptr,
                      nullptr, nullptr, nullptr);
    if (rc) return rc;
    return beam_results_to_host(ctx, T, n, sequence, qstring, moves, score);
}

// signal chunks -> encoder (scores without the blank column, as the reference's beam branch sees them) -> beam search
XB_API int xb_basecall_chunks_beam(xb_ctx *ctx, const float *signal, int n, const char *alphabet, int beam_width, float beam_cut,
                                   float qscale, float qoffset, int8_t *sequence, int8_t *qstring, uint8_t *moves, float *score)
{
    int rc = check_ready(ctx, n);
    if (rc) return rc;
    if (!signal || !sequence || !qstring || !moves || !alphabet) return fail(ctx, XB_ERR_INVALID, "null argument");
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = join_async_decode(ctx))) return rc;
    ctx->result_stream = ctx->stream;
    XB_HIP(ctx, hipMemcpyAsync(ctx->d_signal, signal, sizeof(float) * (size_t)n * ctx->cfg.chunk_len, hipMemcpyHostToDevice,
                               ctx->stream));
    rc = run_encoder(ctx, ctx->d_signal, n, 0, ctx->scores, ctx->ld_nb);
    if (rc) return rc;
    rc = run_beam(ctx, ctx->scores, ctx->T, n, 0, ctx->ld_nb, alphabet, beam_width, beam_cut, qscale, qoffset, nullptr, nullptr,
                  nullptr, nullptr);
    if (rc) return rc;
    return beam_results_to_host(ctx, ctx->T, n, sequence, qstring, moves, score);
}

// room for two co-scheduled calls (once per context; everything in flight is waited for, no held call exists here)
static int reserve_pairing(xb_ctx *ctx)
{
    if (!ctx->fuse_ok) return XB_ERR_STATE;
    if (ctx->cap >= 2 * ctx->cfg.max_batch) return XB_OK;
    int rc = sync_all(ctx);
    if (rc) return rc;
    rc = alloc_workspaces(ctx, 2 * ctx->cfg.max_batch);
    if (rc) {                                           // out of memory: back to one call per pass for good
        ctx->fuse = ctx->fuse_ok = 0;
        const int rc2 = alloc_workspaces(ctx, ctx->cfg.max_batch);
        // (should even that fail the context has no workspaces left: cap == 0, and check_ready refuses every call)
        return rc2 ? rc2 : rc;
    }
    return XB_OK;
}

XB_API int xb_reserve_pairing(xb_ctx *ctx)
{
    if (!ctx) return XB_ERR_INVALID;
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = flush_held(ctx)) return rc;
    if (!ctx->fuse_ok) return XB_OK;                    // this context does not pair calls: every call runs on its own
    const int rc = reserve_pairing(ctx);
    if (rc == XB_OK) ctx->fuse = 1;                     // from now on an asynchronous basecall may be held back for its partner
    return rc;                                          // (XB_ERR_NOMEM: no room for a pair; the context carries on unpaired)
}

XB_API int xb_pairing_active(const xb_ctx *ctx) { return ctx && ctx->fuse ? 1 : 0; }

// what follows a call's decode on its result stream: the host pipeline's D2H copies and done event, a deferred gather
static int call_post_actions(xb_ctx *ctx, const xb_ctx::Call &c, hipStream_t rs)
{
    if (c.slot >= 0) {
        xb_ctx::Slot &sl = ctx->slots[c.slot];
        XB_HIP(ctx, hipMemcpyAsync(sl.h_seq, sl.d_seq, (size_t)c.n * ctx->T, hipMemcpyDeviceToHost, rs));
        XB_HIP(ctx, hipMemcpyAsync(sl.h_len, sl.d_len, sizeof(int32_t) * (size_t)c.n, hipMemcpyDeviceToHost, rs));
        // the error word as THIS batch left it (stream order: behind its recurrences and its decode), not as whatever batch
        // happens to be running when the slot is collected finds it
        XB_HIP(ctx, hipMemcpyAsync(sl.h_err, ctx->error, sizeof(unsigned), hipMemcpyDeviceToHost, rs));
        XB_HIP(ctx, hipEventRecord(sl.done, rs));
    }
    if (c.after) c.after(c.after_arg);
    return XB_OK;
}

// The asynchronous basecall of one call, or of two calls as one batch (chunks [0, a.n) = a, [a.n, a.n + b->n) = b).
static int launch_calls(xb_ctx *ctx, const xb_ctx::Call &a, const xb_ctx::Call *b)
{
    const int n = a.n + (b ? b->n : 0);
    int8_t *d_seq = b ? ctx->fseq : a.seq;
    int32_t *d_len = b ? ctx->flen : a.len;
    const float *sig2 = b ? b->signal : nullptr;
    int rc;
    hipStream_t rs;
    if (!ctx->overlap || !ctx->stream3 || !ctx->scores2 || !ctx->decode_async) {
        rs = ctx->result_stream = ctx->stream;
        rc = run_encoder(ctx, a.signal, n, 0, ctx->scores, ctx->ld_nb, sig2, a.n);
        if (rc) return rc;
        rc = run_decode(ctx, ctx->scores, ctx->T, n, 0, ctx->ld_nb, a.alphabet, nullptr, d_seq, d_len);
        if (rc) return rc;
    } else {
        // asynchronous decode: the encoder of this batch writes score buffer p while the decode of the previous batch may
        // still be reading buffer p ^ 1 on the third stream; the decode that used buffer p two calls ago must be done first
        const int pb = (int)(ctx->batch_idx++ & 1u);
        float *sc = pb ? ctx->scores2 : ctx->scores;
        rs = ctx->result_stream = ctx->stream3;
        if (ctx->dec_pending[pb]) XB_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->dec_done[pb], 0));
        rc = run_encoder(ctx, a.signal, n, 0, sc, ctx->ld_nb, sig2, a.n);
        if (rc) return rc;
        hipEvent_t enc;
        if ((rc = next_dep(ctx, &enc))) return rc;
        XB_HIP(ctx, hipEventRecord(enc, ctx->stream));
        XB_HIP(ctx, hipStreamWaitEvent(ctx->stream3, enc, 0));
        rc = run_decode(ctx, sc, ctx->T, n, 0, ctx->ld_nb, a.alphabet, nullptr, d_seq, d_len, ctx->stream3);
        if (rc) return rc;
    }
    if (b) {        // the pair's rows back to where each caller wants them
        const size_t T = (size_t)ctx->T;
        XB_HIP(ctx, hipMemcpyAsync(a.seq, ctx->fseq, (size_t)a.n * T, hipMemcpyDeviceToDevice, rs));
        XB_HIP(ctx, hipMemcpyAsync(b->seq, ctx->fseq + (size_t)a.n * T, (size_t)b->n * T, hipMemcpyDeviceToDevice, rs));
        if (a.len) XB_HIP(ctx, hipMemcpyAsync(a.len, ctx->flen, sizeof(int32_t) * (size_t)a.n, hipMemcpyDeviceToDevice, rs));
        if (b->len) XB_HIP(ctx, hipMemcpyAsync(b->len, ctx->flen + a.n, sizeof(int32_t) * (size_t)b->n, hipMemcpyDeviceToDevice, rs));
    }
    if (rs == ctx->stream3) {
        const int pb = (int)((ctx->batch_idx - 1) & 1u);
        XB_HIP(ctx, hipEventRecord(ctx->dec_done[pb], ctx->stream3));
        ctx->dec_pending[pb] = true;
    }
    if ((rc = call_post_actions(ctx, a, rs))) return rc;
    if (b && (rc = call_post_actions(ctx, *b, rs))) return rc;
    return XB_OK;
}

// launch a held-back call on its own (every entry point that is not the asynchronous basecall comes through here first)
int flush_held(xb_ctx *ctx)
{
    if (!ctx->holding || ctx->flushing) return XB_OK;
    ctx->flushing = true;
    const xb_ctx::Call h = ctx->held;
    ctx->holding = false;
    const int rc = launch_calls(ctx, h, nullptr);
    ctx->flushing = false;
    if (rc) {                                   // the call itself had already returned XB_OK
        ctx->pipeline_failed = true;
        ctx->deferred_rc = rc;                  // ... so the next call that can return a status reports it
    }
    return rc;
}

static int enqueue_call(xb_ctx *ctx, const xb_ctx::Call &c)
{
    if (ctx->deferred_rc) {
        const int rc = ctx->deferred_rc;
        ctx->deferred_rc = 0;
        return rc;               // xb_last_error still holds the message of the launch that failed
    }
    if (ctx->holding) {
        const xb_ctx::Call h = ctx->held;
        ctx->holding = false;
        bool pair = ctx->fuse && h.n + c.n <= 2 * ctx->cfg.max_batch && strcmp(h.alphabet, c.alphabet) == 0 && h.seq != c.seq;
        if (pair && h.n + c.n > ctx->cap && reserve_pairing(ctx) != XB_OK) pair = false;     // no room for both: one by one
        if (pair) {
            const int rc = launch_calls(ctx, h, &c);
            if (rc) ctx->pipeline_failed = true;
            return rc;
        }
        const int rc = launch_calls(ctx, h, nullptr);
        if (rc) { ctx->pipeline_failed = true; return rc; }
    }
    if (ctx->fuse) {
        ctx->held = c;
        ctx->holding = true;
        return XB_OK;
    }
    return launch_calls(ctx, c, nullptr);
}

XB_API int xb_basecall_chunks_dev(xb_ctx *ctx, const float *d_signal, int n, const char *alphabet, int8_t *d_seq,
                                  int32_t *d_seq_len)
{
    int rc = check_ready(ctx, n);
    if (rc) return rc;
    if (!d_signal || !d_seq || !alphabet) return fail(ctx, XB_ERR_INVALID, "null argument");
    if ((int)strlen(alphabet) < ctx->cfg.n_base + 1 || strlen(alphabet) >= sizeof(xb_ctx::Call{}.alphabet))
        return fail(ctx, XB_ERR_INVALID, "alphabet needs %d symbols", ctx->cfg.n_base + 1);
    XB_HIP(ctx, hipSetDevice(ctx->device));
    xb_ctx::Call c;
    c.signal = d_signal; c.n = n; c.seq = d_seq; c.len = d_seq_len;
    strcpy(c.alphabet, alphabet);
    return enqueue_call(ctx, c);
}

XB_API int xb_basecall_chunks(xb_ctx *ctx, const float *signal, int n, const char *alphabet, int8_t *seq,
                              int32_t *seq_len)
{
    int rc = check_ready(ctx, n);
    if (rc) return rc;
    if (!signal || !seq || !alphabet) return fail(ctx, XB_ERR_INVALID, "null argument");
    XB_HIP(ctx, hipSetDevice(ctx->device));
    XB_HIP(ctx, hipMemcpyAsync(ctx->d_signal, signal, sizeof(float) * (size_t)n * ctx->cfg.chunk_len,
                               hipMemcpyHostToDevice, ctx->stream));
    rc = xb_basecall_chunks_dev(ctx, ctx->d_signal, n, alphabet, ctx->seq, ctx->seq_len);
    if (rc) return rc;
    if ((rc = join_async_decode(ctx))) return rc;
    XB_HIP(ctx, hipMemcpyAsync(seq, ctx->seq, (size_t)n * ctx->T, hipMemcpyDeviceToHost, ctx->stream));
    if (seq_len) XB_HIP(ctx, hipMemcpyAsync(seq_len, ctx->seq_len, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
    return xb_synchronize(ctx);
}

// lazily created: most contexts (tests, bench) never use the host pipeline
static int ensure_slot(xb_ctx *ctx, int slot)
{
    xb_ctx::Slot &sl = ctx->slots[slot];
    if (sl.h_signal) return XB_OK;
    const size_t N = ctx->cfg.max_batch, L = ctx->cfg.chunk_len, T = ctx->T;
    if (!ctx->stream_copy) XB_HIP(ctx, hipStreamCreateWithFlags(&ctx->stream_copy, hipStreamNonBlocking));
    XB_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&sl.h_signal), sizeof(float) * N * L, hipHostMallocDefault));
    XB_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&sl.h_seq), N * T, hipHostMallocDefault));
    XB_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&sl.h_len), sizeof(int32_t) * N, hipHostMallocDefault));
    XB_HIP(ctx, hipHostMalloc(reinterpret_cast<void **>(&sl.h_err), sizeof(unsigned), hipHostMallocDefault));
    int rc = dev_alloc(ctx, &sl.d_signal, N * L);
    rc = rc ? rc : dev_alloc(ctx, &sl.d_seq, N * T);
    rc = rc ? rc : dev_alloc(ctx, &sl.d_len, N);
    if (rc) return rc;
    XB_HIP(ctx, hipEventCreateWithFlags(&sl.h2d, hipEventDisableTiming));
    XB_HIP(ctx, hipEventCreateWithFlags(&sl.done, hipEventDisableTiming));
    return XB_OK;
}

XB_API int xb_submit_chunks(xb_ctx *ctx, int slot, const float *signal, int n, const char *alphabet)
{
    int rc = check_ready(ctx, n);
    if (rc) return rc;
    if (slot < 0 || slot >= XB_PIPELINE_SLOTS || !signal || !alphabet) return fail(ctx, XB_ERR_INVALID, "bad slot / null argument");
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if ((rc = ensure_slot(ctx, slot))) return rc;
    xb_ctx::Slot &sl = ctx->slots[slot];
    if (sl.busy) return fail(ctx, XB_ERR_STATE, "slot %d was submitted and not collected", slot);
    const size_t bytes = sizeof(float) * (size_t)n * ctx->cfg.chunk_len;
    memcpy(sl.h_signal, signal, bytes);                       // the caller's buffer is free again on return
    XB_HIP(ctx, hipMemcpyAsync(sl.d_signal, sl.h_signal, bytes, hipMemcpyHostToDevice, ctx->stream_copy));
    XB_HIP(ctx, hipEventRecord(sl.h2d, ctx->stream_copy));
    XB_HIP(ctx, hipStreamWaitEvent(ctx->stream, sl.h2d, 0));
    if ((int)strlen(alphabet) < ctx->cfg.n_base + 1 || strlen(alphabet) >= sizeof(xb_ctx::Call{}.alphabet))
        return fail(ctx, XB_ERR_INVALID, "alphabet needs %d symbols", ctx->cfg.n_base + 1);
    // the D2H copies of the results, the error-word snapshot and the slot's done event follow the launch of this call
    // (call_post_actions) -- which may be held back until the next submit so that the two batches share one pass
    xb_ctx::Call c;
    c.signal = sl.d_signal; c.n = n; c.seq = sl.d_seq; c.len = sl.d_len; c.slot = slot;
    strcpy(c.alphabet, alphabet);
    rc = enqueue_call(ctx, c);
    if (rc) return rc;
    sl.n = n;
    sl.busy = true;
    return XB_OK;
}

XB_API int xb_collect_chunks(xb_ctx *ctx, int slot, int8_t *seq, int32_t *seq_len)
{
    if (!ctx) return XB_ERR_INVALID;
    if (slot < 0 || slot >= XB_PIPELINE_SLOTS || !seq) return fail(ctx, XB_ERR_INVALID, "bad slot / null argument");
    xb_ctx::Slot &sl = ctx->slots[slot];
    if (!sl.busy) return fail(ctx, XB_ERR_STATE, "slot %d has nothing in flight", slot);
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (ctx->holding && ctx->held.slot == slot) (void)flush_held(ctx);      // a failure shows as pipeline_failed below
    XB_HIP(ctx, hipEventSynchronize(sl.done));
    sl.busy = false;
    memcpy(seq, sl.h_seq, (size_t)sl.n * ctx->T);
    if (seq_len) memcpy(seq_len, sl.h_len, sizeof(int32_t) * (size_t)sl.n);
    // the persistent recurrence reports a lost rendezvous through the error word (snapshot taken behind this batch):
    // results would be garbage.  The word is not cleared while another batch is in flight -- that batch fails too (it ran
    // on a device in an unknown state) -- and is reset once the pipeline has drained.
    if (*sl.h_err != 0) ctx->pipeline_failed = true;
    if (ctx->pipeline_failed) {
        bool any_busy = false;
        for (auto &s2 : ctx->slots) any_busy = any_busy || s2.busy;
        if (!any_busy) {
            (void)sync_all(ctx);
            (void)hipMemset(ctx->error, 0, sizeof(unsigned));
            ctx->pipeline_failed = false;
        }
        return fail(ctx, XB_ERR_DEVICE, "LSTM inter-workgroup sync timed out (persistent kernel was not fully resident?)");
    }
    return XB_OK;
}

XB_API int xb_set_profiling(xb_ctx *ctx, int on)
{
    if (!ctx) return XB_ERR_INVALID;
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = flush_held(ctx)) return rc;            // a held-back call is timed (or not) as it was when it was made
    ctx->profiling = on != 0;
    return XB_OK;
}

XB_API int xb_get_stage_times(xb_ctx *ctx, float ms[XB_STAGE_COUNT], int64_t launches[XB_STAGE_COUNT])
{
    if (!ctx) return XB_ERR_INVALID;
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = flush_held(ctx)) return rc;
    if (int rc = sync_all(ctx)) return rc;
    collect_events(ctx);
    for (int i = 0; i < XB_STAGE_COUNT; ++i) {
        if (ms) ms[i] = ctx->stage_ms[i];
        if (launches) launches[i] = ctx->stage_launches[i];
    }
    return XB_OK;
}

XB_API int xb_reset_stage_times(xb_ctx *ctx)
{
    if (!ctx) return XB_ERR_INVALID;
    if (int rc = flush_held(ctx)) return rc;
    if (int rc = sync_all(ctx)) return rc;
    collect_events(ctx);
    for (int i = 0; i < XB_STAGE_COUNT; ++i) { ctx->stage_ms[i] = 0.f; ctx->stage_launches[i] = 0; }
    return XB_OK;
}

XB_API int xb_geometry(const xb_ctx *ctx, int *T, int *S, int *C_blank, int *C_noblank)
{
    if (!ctx) return XB_ERR_INVALID;
    if (T) *T = ctx->T;
    if (S) *S = ctx->S;
    if (C_blank) *C_blank = ctx->S * (ctx->cfg.n_base + 1);
    if (C_noblank) *C_noblank = ctx->O;
    return XB_OK;
}

// Diagnostic (tests of the mixed-precision encoder): the activation tensors the last xb_encode / xb_encode_dev of `n` chunks left
// in the ping-pong buffers -- which = 0: output of LSTM layer 3, 1: output of LSTM layer 4 -- as (T, n, features) fp16 bit
// patterns `hi` and the raw 2-byte-per-element second part (fp16 residual or q8 image, whichever the consuming stage reads).
XB_API int xb_debug_layer_output(xb_ctx *ctx, int which, int n, uint16_t *hi, uint16_t *second)
{
    if (!ctx || !hi || !second || which < 0 || which > 1) return XB_ERR_INVALID;
    if (n < 1 || n > ctx->cap) return fail(ctx, XB_ERR_INVALID, "batch %d outside [1, %d]", n, ctx->cap);
    XB_HIP(ctx, hipSetDevice(ctx->device));
    if (int rc = join_async_decode(ctx)) return rc;
    if (int rc = sync_all(ctx)) return rc;
    const size_t bytes = sizeof(uint16_t) * (size_t)ctx->T * n * ctx->cfg.features;
    XB_HIP(ctx, hipMemcpy(hi, ctx->x_hi[which], bytes, hipMemcpyDeviceToHost));
    XB_HIP(ctx, hipMemcpy(second, ctx->x_lo[which], bytes, hipMemcpyDeviceToHost));
    return XB_OK;
}

#ifdef XB_LSTM_STAMPS
// diagnostic build only (csrc/Makefile target `diag`): per-phase cycle sums of the LSTM kernel's workgroup 0
XB_API int xb_debug_lstm_stamps(xb_ctx *ctx, unsigned long long out[10], int reset)
{
    if (!ctx) return XB_ERR_INVALID;
    XB_HIP(ctx, hipStreamSynchronize(ctx->stream));
    xb::lstm_read_stamps(out, reset != 0);
    return XB_OK;
}
#endif
#ifdef XB_GEMM_STAMPS
// diagnostic build only: per-phase cycle sums of gemm4p_kernel<*, 3> (xb_encoder.hip, g_gemm_stamps)
XB_API int xb_debug_gemm_stamps(xb_ctx *ctx, unsigned long long out[8], int reset)
{
    if (!ctx) return XB_ERR_INVALID;
    if (int rc = sync_all(ctx)) return rc;
    xb::gemm_read_stamps(out, reset != 0);
    return XB_OK;
}
#endif

}  // extern "C"

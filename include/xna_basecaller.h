/*
 * xna_basecaller.h -- C ABI of libxnacall.so, the MI355X (gfx950) implementation of the
 * ub-bonito CRF basecalling hot path.
 *
 * The reference (CSB5/XNA_Basecaller) is 100 % Python; its device arithmetic lives in
 * un-vendored CUDA wheels (torch/cuDNN, ont-seqdist-cuda 0.0.4, koi 0.0.5).  There is therefore
 * no native reference interface to mirror: every entry point below replaces a *Python-level*
 * operator of the reference, cited as file:line under /root/reference/ub-bonito/bonito/.
 * The binding a maintainer adds on the reference side is a ctypes stub.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no exceptions across the ABI.
 *   - every function returns XB_OK (0) or a negative xb_status; xb_last_error(ctx) gives text.
 *   - one xb_ctx per GPU, and the GPU to itself: the persistent LSTM kernel keeps up to 192 workgroups (one per CU, all
 *     of the CU's registers) resident and exchanging data for a whole layer; a second tenant on the same device (another
 *     process, a CU mask) can keep part of them from becoming resident, in which case the waiting workgroups give up
 *     after a bounded spin and the next xb_synchronize / xb_collect_chunks reports XB_ERR_DEVICE.
 *   - the ctx owns its HIP streams (a main stream plus two low-priority side streams: the next
 *     layer's input GEMM runs beside the current layer's recurrence; the CRF decode of a batch runs on the MAIN stream
 *     behind its encoder -- round 4's schedule; XB_DECODE_ASYNC=1 restores the side-stream decode beside the next batch's
 *     encoder, the only case in which xb_result_stream is not the main stream) and every device buffer it allocates.  A ctx is
 *     used by one thread at a time (the reference calls compute_scores from ONE pipeline
 *     thread, crf/basecall.py:109-111); distinct ctxs are independent.
 *   - "host" entry points take host buffers owned by the caller and block until the result is
 *     in them.  "_dev" entry points take device pointers valid on the ctx's device (e.g.
 *     torch tensor data_ptr()), enqueue on the ctx's streams and return without waiting.  The ONLY completion
 *     point is xb_synchronize (it joins all of the ctx's streams): call it before reading results or reusing /
 *     freeing the buffers passed in.  Consecutive xb_basecall_chunks_dev calls queue up on the device back to back (no host
 *     synchronisation between them; what runs concurrently is the co-scheduled pair below and, inside a batch, the next layer's
 *     input GEMM beside the current recurrence): give each in-flight batch its own d_seq / d_seq_len -- and its own d_signal
 *     that stays untouched until xb_synchronize (or until work ordered behind xb_result_stream has run).
 *   - a caller that keeps two batches in flight can have them CO-SCHEDULED: after xb_reserve_pairing (an explicit opt-in;
 *     contexts of at most 640 chunks at features 768 -- a pair must fit ONE recurrence launch of two chunk groups per
 *     workgroup: 2 x 8 groups of 64 with a group's members on one XCD, 2 x 10 with the members dealt over all XCDs, which the
 *     library does for 513..640 chunks, XB_LSTM_WIDE=0 restores the 512 limit; XB_FUSE=0 refuses) an xb_basecall_chunks_dev / xb_submit_chunks that finds nothing
 *     held back is itself held back (NOTHING is enqueued yet -- a device-wide synchronise or an event recorded on a stream
 *     fetched earlier does not cover it; its launch status is reported by the call that launches it) until the next such call
 *     arrives; the two batches then go through the encoder and the decode as one -- the recurrence serves two chunk groups
 *     per workgroup and hides one group's hand-off behind the other's arithmetic -- and each call's results land in its own
 *     buffers.  A held-back call is launched on its own, with the weights and the profiling state it was called under, by
 *     every other entry point (xb_load_weights, xb_weights_ready and xb_set_profiling included), by xb_synchronize,
 *     xb_result_stream, xb_collect_chunks of its slot and xb_ctx_destroy; results are the same bytes either way.  Without the
 *     opt-in every asynchronous call is enqueued before it returns.
 *   - layouts are the reference's: signal (N, L) fp32 [= (N,1,L)], scores (T, N, C) fp32
 *     time-major, labels / seq (N, T) int8.
 */
#ifndef XNA_BASECALLER_H
#define XNA_BASECALLER_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define XB_API __attribute__((visibility("default")))
#else
#define XB_API
#endif

typedef struct xb_ctx xb_ctx;

typedef enum xb_status {
    XB_OK = 0,
    XB_ERR_INVALID = -1,      /* bad argument / unsupported configuration */
    XB_ERR_HIP = -2,          /* a HIP runtime call failed */
    XB_ERR_NOMEM = -3,        /* host or device allocation failed */
    XB_ERR_STATE = -4,        /* call out of order (e.g. weights not loaded) */
    XB_ERR_DEVICE = -5,       /* a kernel reported an internal failure (e.g. sync timeout) */
    XB_ERR_NO_GPU = -6        /* no usable gfx950 device */
} xb_status;

/* Arithmetic of the dense projections (Conv1d k19, LSTM, Linear). */
typedef enum xb_precision {
    XB_PREC_F16X3 = 0,        /* split-fp16 MFMA, 3 products, fp32 accumulate: |score err| ~3e-6 */
    XB_PREC_F16 = 1,          /* single fp16 MFMA, fp32 accumulate (the reference's model.half()): ~1e-3 */
    XB_PREC_F16F8 = 2,        /* fp16 main product + both correction products on the block-scaled FP8 MFMA: ~4e-5 */
    XB_PREC_F16F8_IN1 = 3,    /* as F16F8, but the LSTM input projections (45 % of the FLOPs, no feedback through time) keep
                                 only the fp16 main product: |score err| ~6e-4 max / 1e-4 rms, 1.19x the throughput */
    XB_PREC_MIXED = 4         /* the feed-forward projections (conv3, the five LSTM input projections, the CRF linear layer) in
                                 the three-product F16X3 arithmetic, the five recurrent projections (W_hh register-resident,
                                 the critical path) in F16F8.  Default of Model, the CLI and bench.py: |score err| 3.4e-4 max on
                                 trained-like (peaky) weights, where plain F16F8 sits on the 1e-3 tolerance (1.08e-3), at 0.87x
                                 its throughput (profiles/r04_x3_attribution.txt) */
} xb_precision;

/*
 * Model + geometry.  Mirrors config.toml [encoder]/[global_norm]/[labels]
 * (models/xna_r9.4.1_e8_sup@v3.3/config.toml:1-29) and rnn_encoder() (crf/model.py:142-160).
 */
typedef struct xb_config {
    int32_t n_base;           /* len(labels) - 1 : 4, 5 or 6                                   */
    int32_t state_len;        /* [global_norm] state_len (3)                                   */
    int32_t features;         /* [encoder] features (768); multiple of 32                      */
    int32_t winlen;           /* [encoder] winlen (19)                                         */
    int32_t stride;           /* [encoder] stride (5)                                          */
    float scale;              /* [encoder] scale (5.0)                                         */
    float blank_score;        /* [encoder] blank_score (2.0)                                   */
    int32_t chunk_len;        /* samples per chunk L (basecaller chunksize); T = L / stride    */
    int32_t max_batch;        /* largest N passed to any call                                  */
    int32_t precision;        /* xb_precision                                                  */
    int32_t lstm_mode;        /* 0 = auto, 1 = one launch per time step, 2 = persistent kernel */
} xb_config;

/* ---- lifetime ------------------------------------------------------------------------- */

/* Create a context on HIP device `device`.  Replaces Model(config).to(device), util.py:295,365. */
XB_API int xb_ctx_create(xb_ctx **out, int device, const xb_config *cfg);
XB_API void xb_ctx_destroy(xb_ctx *ctx);
/* Text of the last error on `ctx` (or of the last failed xb_ctx_create when ctx is NULL). */
XB_API const char *xb_last_error(const xb_ctx *ctx);
XB_API int xb_device_count(void);

/*
 * Load one tensor of the reference state dict (model.load_state_dict, util.py:354) in PyTorch
 * layout, fp32, from host memory.  `name` is the inference-encoder key:
 *   encoder.{0,1,2}.conv.{weight,bias}          Conv1d (out, in, k)
 *   encoder.{4..8}.rnn.{weight_ih_l0,weight_hh_l0,bias_ih_l0,bias_hh_l0}   (4H, H) / (4H), gates i,f,g,o
 *   encoder.9.linear.{weight,bias}              (n_base^(state_len+1), H)
 * `n` is the element count and must match the shape implied by the config.
 */
XB_API int xb_load_weights(xb_ctx *ctx, const char *name, const float *host, int64_t n);
/* Re-layout the loaded tensors for the kernels (model.eval()/.to(device)); requires all 28 tensors. */
XB_API int xb_weights_ready(xb_ctx *ctx);

/* ---- operators -------------------------------------------------------------------------- */

/*
 * Model.forward: scores = model(batch) (crf/basecall.py:53, crf/model.py:212-213, nn.py).
 * signal (n, L) fp32.  scores (T, n, C): C = S*(n_base+1) when expand_blanks != 0 (the
 * LinearCRFEncoder layout with the blank column, nn.py:123-130), else S*n_base.
 */
XB_API int xb_encode(xb_ctx *ctx, const float *signal, int n, int expand_blanks, float *scores);
XB_API int xb_encode_dev(xb_ctx *ctx, const float *d_signal, int n, int expand_blanks, float *d_scores);

/*
 * SeqdistModel.decode_batch + path_to_str + the left-pack of compute_scores
 * (crf/model.py:215-218, 92-100; crf/basecall.py:57-76).
 * scores (T, n, C) fp32, C = S*(n_base+1) if has_blank else S*n_base (blank = cfg.blank_score).
 * labels (n, T) int8 [optional]: per-step arg-max edge % (n_base+1).
 * seq    (n, T) int8 [optional]: ASCII of alphabet[label] for label != 0, left-packed, zero padded.
 * seq_len (n) int32 [optional].   alphabet: n_base+1 bytes, e.g. "NACGTXY".
 */
XB_API int xb_decode(xb_ctx *ctx, const float *scores, int T, int n, int has_blank,
                     const char *alphabet, int8_t *labels, int8_t *seq, int32_t *seq_len);
XB_API int xb_decode_dev(xb_ctx *ctx, const float *d_scores, int T, int n, int has_blank,
                         const char *alphabet, int8_t *d_labels, int8_t *d_seq, int32_t *d_seq_len);

/*
 * The Log-semiring scans of the CRF on their own (seqdist `sparse` operators behind crf/model.py:41-61): any of
 *   alpha (T+1, n, S)  CTC_CRF.forward_scores  (crf/model.py:50-54): alpha_0 = 0, alpha_{t+1}[j] = LSE_k(M[t,j,k] + alpha_t[idx[j,k]])
 *   beta  (T+1, n, S)  CTC_CRF.backward_scores (crf/model.py:56-60): beta_T = 0, beta_t[i] = LSE over edges (j,k) leaving i of
 *                      (M[t,j,k] + beta_{t+1}[j])
 *   logz  (n)          CTC_CRF.logZ            (crf/model.py:41-46): LSE_j alpha_T[j]   (`normalise` = scores - logz / T)
 *   post  (T, n, S*(n_base+1))  the edge posteriors exp(alpha_t[src] + M + beta_{t+1}[dst] - logZ) = d logZ / d scores
 *                      (seqdist `posteriors`, Log semiring; the blank column is part of the layout also when the scores
 *                      come without it); the _dev variant writes rows of stride (S*(n_base+1) + 3) & ~3 floats.
 * NULL outputs are skipped (alpha/logz alone stop after the forward sweep).  Same arithmetic contract as the decode:
 * bit-equal to the oracle.  xb_crf_logz = xb_crf_scans with only logz.
 */
XB_API int xb_crf_scans(xb_ctx *ctx, const float *scores, int T, int n, int has_blank,
                        float *alpha, float *beta, float *logz, float *post);
XB_API int xb_crf_scans_dev(xb_ctx *ctx, const float *d_scores, int T, int n, int has_blank,
                            float *d_alpha, float *d_beta, float *d_logz, float *d_post);
XB_API int xb_crf_logz(xb_ctx *ctx, const float *scores, int T, int n, int has_blank, float *logz);
XB_API int xb_crf_logz_dev(xb_ctx *ctx, const float *d_scores, int T, int n, int has_blank, float *d_logz);

/*
 * Beam search with qualities and moves: the non-Viterbi branch of compute_scores (crf/basecall.py:33-46:
 * `sequence, qstring, moves = koi.decode.beam_search(scores, beam_width=32, beam_cut=100.0, scale=1.0, offset=0.0,
 * blank_score=2.0)`, selected when the model's last layer has expand_blanks = False).  koi 0.0.5 is not part of the
 * reference tree; the algorithm restated is the one ONT publishes for this decoder (back-guided beam with CRC-32C sequence
 * hashes, stay / step merging, beam cut by bisection, k-mer posterior qualities), generalised to n_base in 2..7 -- its
 * specification is the "CRF beam search" section of oracle/xna_oracle.c, against which the kernel is bit-exact.  PARITY UNPINNED.
 *   scores    (T, n, C) fp32; has_blank = 0: C = S * n_base and the stay score is the context's blank_score (what the
 *             reference passes); has_blank = 1: the stay score is column 0 of every state row.
 *   beam_width 1..32; beam_cut > 0 (candidates below max - log(beam_cut) are dropped; <= 0: no cut); qscale / qoffset =
 *             koi's scale / offset on the phred value.  States: at most 1024 (the limit of the scans).
 *   sequence, qstring (n, T) int8: the base character alphabet[1 + base] / the quality character (33 + q, q in 1..50) at the
 *             blocks that emit a base, 0 elsewhere (koi.decode.to_str drops the zeros); moves (n, T) uint8 1 = a base is
 *             emitted in this block (moves[0] is always 1); score (n) [optional] the log-sum path score of the result.
 * xb_basecall_chunks_beam = encoder (no blank column) + beam search without the scores leaving the device.
 */
XB_API int xb_beam_search(xb_ctx *ctx, const float *scores, int T, int n, int has_blank, const char *alphabet, int beam_width,
                          float beam_cut, float qscale, float qoffset, int8_t *sequence, int8_t *qstring, uint8_t *moves,
                          float *score);
XB_API int xb_beam_search_dev(xb_ctx *ctx, const float *d_scores, int T, int n, int has_blank, const char *alphabet,
                              int beam_width, float beam_cut, float qscale, float qoffset, int8_t *d_sequence,
                              int8_t *d_qstring, uint8_t *d_moves, float *d_score);
XB_API int xb_basecall_chunks_beam(xb_ctx *ctx, const float *signal, int n, const char *alphabet, int beam_width,
                                   float beam_cut, float qscale, float qoffset, int8_t *sequence, int8_t *qstring,
                                   uint8_t *moves, float *score);

/*
 * The CTC-CRF loss scans (CTC_CRF.ctc_loss and ctc_viterbi_alignments, crf/model.py:102-135: prepare_ctc_scores +
 * seqdist.ctc_simple.logZ_cupy / viterbi_alignments), for `bonito evaluate` / fine-tuning on the same device:
 *   scores  (T, n, S*(n_base+1)) fp32 with the blank column -- ctc_loss passes the NORMALISED scores
 *           (scores - xb_crf_logz / T, crf/model.py:48-49,120-121); targets (n, Lt) int32 CTC labels (1..n_base, 0 = padding);
 *           target_lengths (n) in bases, state_len <= length <= Lt.  With np = Lt - state_len + 1 target positions the lattice
 *           has a stay edge per position (score column stay_idx) and a move edge between consecutive positions (move_idx).
 *   xb_ctc_logz:  logz (n) = log-sum over all monotone paths that end at position target_length - state_len at time T
 *           (loss = -logz / target_length);  gstay (T, n, np) / gmove (T, n, np-1) [optional] = d logz / d stay, d logz / d move,
 *           the restricted posteriors -- scatter-added over (stay_idx, move_idx) they are d logz / d scores.
 *   xb_ctc_alignments:  the Max-semiring path: alignments (T, n, np) one-hot over positions (where the best path sits before
 *           step t: seqdist's stay.grad + shifted move.grad; ties prefer the stay edge), max_score (n) [optional].
 * Same arithmetic contract as the decode (bit-equal to the oracle; seqdist itself is absent: "parity unpinned").
 */
XB_API int xb_ctc_logz(xb_ctx *ctx, const float *scores, int T, int n, const int32_t *targets, int Lt,
                       const int32_t *target_lengths, float *logz, float *gstay, float *gmove);
XB_API int xb_ctc_alignments(xb_ctx *ctx, const float *scores, int T, int n, const int32_t *targets, int Lt,
                             const int32_t *target_lengths, float *alignments, float *max_score);

/* compute_scores (crf/basecall.py:27-82), viterbi branch: encode + decode without materialising
 * the blank column or copying scores off the device. */
XB_API int xb_basecall_chunks(xb_ctx *ctx, const float *signal, int n, const char *alphabet,
                              int8_t *seq, int32_t *seq_len);
XB_API int xb_basecall_chunks_dev(xb_ctx *ctx, const float *d_signal, int n, const char *alphabet,
                                  int8_t *d_seq, int32_t *d_seq_len);
/* Opt in to the co-scheduling of two calls in flight (see the header comment) and make room for it: the workspaces for
 * max_batch chunks are replaced by twice that -- which waits for everything in flight and takes a second or two, so callers
 * do it once, up front (bench.py: outside its timed region; Model: when the host pipeline starts).  XB_OK also when the
 * context cannot pair calls (XB_FUSE=0, max_batch > 512, serial schedule): xb_pairing_active tells (1: asynchronous calls
 * may be held back for a partner from now on; 0: every call is enqueued before it returns).  XB_ERR_NOMEM: no room for a
 * pair -- the context carries on unpaired.  No counterpart in the reference (single batch in flight, crf/basecall.py:109-111). */
XB_API int xb_reserve_pairing(xb_ctx *ctx);
XB_API int xb_pairing_active(const xb_ctx *ctx);

/*
 * The same operator for a host pipeline that keeps the device busy (crf/basecall.py:96-119: the reference overlaps its
 * stages with threads and bounded queues; here the overlap is two batches in flight on the device).
 * xb_submit_chunks copies `signal` (n, L) into pinned staging of `slot` (0 .. XB_PIPELINE_SLOTS - 1), enqueues H2D on a
 * copy stream, the fused encode + decode, and the D2H of the results into pinned staging, and returns without waiting.
 * xb_collect_chunks waits for that slot's batch only and copies seq (n, T) / seq_len (n) out.  Typical use: submit
 * batch k+1 into the next slot, then collect batch k -- results arrive one batch late, in order; with co-scheduled pairs
 * (xb_reserve_pairing) rotate all four slots -- submit batch k+3, then collect batch k -- so that pair (k+2, k+3) is on the
 * device before the host waits for pair (k, k+1).
 * A slot must be collected before it is submitted again (XB_ERR_STATE otherwise).
 */
#define XB_PIPELINE_SLOTS 4
XB_API int xb_submit_chunks(xb_ctx *ctx, int slot, const float *signal, int n, const char *alphabet);
XB_API int xb_collect_chunks(xb_ctx *ctx, int slot, int8_t *seq, int32_t *seq_len);

XB_API int xb_synchronize(xb_ctx *ctx);

/*
 * The HIP stream (a hipStream_t, returned as void *) on which the outputs of the most recent *_dev call are produced.
 * A caller that consumes d_seq / d_seq_len on a stream of its own without a host-side xb_synchronize records an event
 * on this stream right after the call and waits for it there; before the same output buffers are handed to a later
 * call it makes this stream wait for its own "consumed" event.  This is how the gather of called sequences
 * (RCCL all_gather on a side stream) overlaps the next batch; the reference, single-device Python, has no
 * counterpart.
 */
XB_API void *xb_result_stream(xb_ctx *ctx);

/* ---- multi-GPU: the gather of called sequences --------------------------------------------------
 *
 * Reads shard over the GPUs of a node, one process and one xb_ctx per GPU, with no data-path collective; the only exchange
 * is this gather of the packed sequences (RCCL over xGMI; librccl is opened at run time).  The reference is single-device
 * Python and has no counterpart.
 *   xb_comm_unique_id   rank 0 draws the 128-byte id (ncclGetUniqueId) and hands it to the other ranks out of band (the
 *                       launcher's rendezvous: xna_basecaller_amd/dist.py exchanges it over a socket on MASTER_ADDR);
 *   xb_comm_create      every rank joins (ncclCommInitRank) on its device; collective: returns when all `world` ranks have;
 *   xb_gather_called    all-gather of one batch: d_seq (n, T) int8 / d_seq_len (n) int32 of this rank into
 *                       d_all_seq (world, n, T) / d_all_len (world, n) on every rank, enqueued on the communicator's own stream
 *                       behind xb_result_stream(ctx) (ctx may be NULL when the caller has synchronised) -- returns at once;
 *                       every rank passes the same n and T;
 *   xb_comm_fence       makes the streams of ctx wait (on the device) for the gather issued `lag` calls ago (0: the latest, 1: the
 *                       one before it) and everything older -- call it before output buffers a gather still reads are handed
 *                       to a later xb_basecall_chunks_dev (two buffer sets in rotation: lag 1 right before enqueueing a batch);
 *   xb_comm_synchronize host-side completion of the gathers asked for so far (one still waiting for a held-back basecall is
 *                       launched first).
 */
#define XB_COMM_ID_BYTES 128
typedef struct xb_comm xb_comm;
XB_API int xb_comm_unique_id(char id[XB_COMM_ID_BYTES]);
XB_API int xb_comm_create(xb_comm **out, int device, int rank, int world, const char id[XB_COMM_ID_BYTES]);
/* xb_comm_destroy: DESTROY ORDER -- the communicator first or the contexts first, both are safe, under one rule: every
 * context it gathered for and every buffer handed to xb_gather_called must still be alive when xb_comm_destroy is called if a
 * gather is still waiting for a held-back basecall of that context (see the header comment on co-scheduling).
 * xb_comm_destroy and xb_comm_synchronize launch such a basecall themselves (as xb_result_stream(ctx) would) and wait for
 * its gather, so no deferred gather outlives its communicator; after xb_comm_destroy the context holds no reference to it.
 * Destroying the context first launches the held call too (xb_ctx_destroy), with the communicator still alive.  What is NOT
 * allowed: freeing d_seq / d_seq_len / d_all_* of a batch before xb_synchronize(ctx) (or xb_comm_synchronize for the
 * gathered rows) has returned -- a held call writes them when it is launched. */
XB_API void xb_comm_destroy(xb_comm *comm);
XB_API int xb_comm_rank(const xb_comm *comm);
XB_API int xb_comm_world(const xb_comm *comm);
XB_API const char *xb_comm_last_error(const xb_comm *comm);
XB_API int xb_gather_called(xb_comm *comm, xb_ctx *ctx, const int8_t *d_seq, const int32_t *d_seq_len, int n, int T,
                            int8_t *d_all_seq, int32_t *d_all_len);
XB_API int xb_comm_fence(xb_comm *comm, xb_ctx *ctx, int lag);
XB_API int xb_comm_synchronize(xb_comm *comm);
/* Makes every stream of ctx wait for `hip_event` (a hipEvent_t passed as void *) -- the edge xb_comm_fence uses. */
XB_API int xb_stream_wait_event(xb_ctx *ctx, void *hip_event);

/* ---- host-side scoring used by `bonito evaluate` -----------------------------------------------------------------
 * util.accuracy(ref, seq, balanced, min_coverage) (util.py:402-424): Smith-Waterman local alignment of the called sequence
 * against its reference (gap open 8 / extend 4, match +5 / mismatch -4) and '=' / ('=' + 'I' + 'X' + 'D') * 100 from its trace
 * (balanced: ('=' - 'I') / ('=' + 'X' + 'D')); 0 when less than min_coverage of the reference is aligned.  A restatement of
 * the parasail call the reference makes (parasail is in no image): see csrc/xb_align.hip for the two stated choices.  Pure host
 * code, no context needed.  counts (optional) receives the numbers of '=', 'X', 'I', 'D' columns.
 */
XB_API int xb_align_accuracy(const char *ref, int ref_len, const char *seq, int seq_len, double min_coverage, int balanced,
                             double *accuracy, int32_t counts[4]);

/* ---- introspection / measurement ---------------------------------------------------------- */

enum { XB_STAGE_CONV = 0, XB_STAGE_LSTM_IN = 1, XB_STAGE_LSTM_REC = 2, XB_STAGE_LINEAR = 3,
       XB_STAGE_DECODE = 4, XB_STAGE_COUNT = 5 };

/* Turn per-stage HIP-event timing on/off (events are recorded on the ctx stream). */
XB_API int xb_set_profiling(xb_ctx *ctx, int on);
/* Accumulated per-stage device time (ms) and kernel launch counts since the last reset;
 * synchronises the stream.  Either pointer may be NULL. */
XB_API int xb_get_stage_times(xb_ctx *ctx, float ms[XB_STAGE_COUNT], int64_t launches[XB_STAGE_COUNT]);
XB_API int xb_reset_stage_times(xb_ctx *ctx);
/* Output time steps per chunk, states, score columns of the loaded config. */
XB_API int xb_geometry(const xb_ctx *ctx, int *T, int *S, int *C_blank, int *C_noblank);
/* Diagnostic, no counterpart in the reference (tests of the mixed-precision encoder, XB_PREC_MIXED): the activations the
 * last xb_encode / xb_encode_dev of n chunks left on the device -- which = 0: output of LSTM layer 3 (nn.py:216-220, module
 * encoder.7), 1: of LSTM layer 4 (encoder.8) -- as (T, n, features) fp16 bit patterns `hi` plus the raw second part (2 bytes per
 * element: the fp16 residual, or the q8 image of DESIGN.md 2, whichever the consuming stage's arithmetic reads). */
XB_API int xb_debug_layer_output(xb_ctx *ctx, int which, int n, uint16_t *hi, uint16_t *second);
XB_API const char *xb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* XNA_BASECALLER_H */

// xb_internal.h -- declarations shared by the translation units of libxnacall.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace xb {

typedef _Float16 half_t;

// ---------------------------------------------------------------- CRF decode (xb_decode.hip)
struct DecodeParams {
    const float *scores;   // (T, N, ld) fp32, first `cin` columns of each row are used
    int T, N, S, hi, nb;   // hi = nb^(state_len-1)
    int cin, ld, has_blank;
    float blank;
    float *alpha, *beta, *bmax;   // (T+1, N, S) fp32 stashes
    float *qbuf;                  // (T, N, ldq) fp32 log-posteriors written by sweep 2, read by sweep 3
    int ldq;                      // >= S*(nb+1), multiple of 4
    float *logz;                  // (N) or nullptr
    int stop_after;               // 0: full decode; 1: return after the Log forward sweep (alpha, logZ); 2: after the Log
                                  // backward sweep (xb_crf_scans)
    float *beta_out;              // (T+1, N, S) or nullptr: the Log backward scores, written by sweep 2
    int post_mode;                // 1: sweep 2 writes the posteriors P instead of Q = log(P + 1e-8) into qbuf
    int8_t *labels;               // (N, T) or nullptr
    int8_t *seq;                  // (N, T) or nullptr
    int32_t *seq_len;             // (N) or nullptr
    char alphabet[8];
    int debug_stop;               // diagnostic builds only (XB_LSTM_STAMPS): return after sweep 1 / 2
    int debug_lds;                // diagnostic builds only: sweep 2 with lane-linear (conflict-free) LDS addresses, WRONG results
};
hipError_t launch_crf_decode(const DecodeParams &p, hipStream_t stream);
int decode_lanes_per_state(int S, int N);   // 1, 2 or 4 lanes serve one CRF state (env XB_DECODE_LPS overrides for tests)

// ---------------------------------------------------------------- CTC-CRF loss scans (xb_decode.hip)
// seqdist.ctc_simple logZ over the stay / move lattice of the targets (crf/model.py:102-135): position l of n = Lt - sl + 1,
// stay[t][l] = scores[t][b][stay_idx[b][l]], move[t][l] = scores[t][b][move_idx[b][l]] (l -> l + 1).
struct CtcParams {
    const float *scores;         // (T, N, C) fp32 with the blank column
    int T, N, C;
    const int32_t *stay_idx;     // (N, n)
    const int32_t *move_idx;     // (N, n - 1)
    int n;
    const int32_t *tlen;         // (N) target lengths (in bases); positions in use = tlen + 1 - sl
    int sl;
    int semiring;                // 0 Log, 1 Max
    float *alpha;                // (N, T + 1, n) workspace, required for gstay / gmove
    float *logz;                 // (N) or nullptr
    float *gstay;                // (T, N, n) or nullptr: Log: d logZ / d stay; Max: the one-hot alignment (zeroed by the caller)
    float *gmove;                // (T, N, n - 1) or nullptr (Log only)
    unsigned *error;             // bit 1 set when a target length is out of range
};
hipError_t launch_ctc_scan(const CtcParams &p, hipStream_t stream);
int ctc_max_positions();

// ---------------------------------------------------------------- beam search (xb_beam.hip)
// koi.decode.beam_search as compute_scores calls it (crf/basecall.py:43-46): back-guided beam over the CRF with sequence
// hashes, stay / step merging, qualities from k-mer posteriors, moves.  One wave per chunk.
constexpr int BEAM_MAX_WIDTH = 32;
constexpr int BEAM_MAX_STATES = 1024;      // = the CRF scans' limit (launch_crf_decode)
struct BeamParams {
    const float *scores;         // (T, N, ld) fp32
    int ld, has_blank;
    float blank;                 // the stay score when the scores come without the blank column
    const float *alpha, *beta;   // (T + 1, N, S) Log-semiring scans of the same scores (xb_crf_scans)
    const float *logz;           // (N)
    int T, N, S, nb, hi;         // hi = nb^(state_len - 1)
    int W;                       // beam width, 1..BEAM_MAX_WIDTH
    float log_cut;               // log(beam_cut), FLT_MAX = no cut
    float qscale, qoffset;
    char base_chars[8];          // alphabet[1 + base]
    uint32_t *hist;              // (N, T + 1, BEAM_MAX_WIDTH) workspace: state | prev << 16 | stay << 24
    int32_t *path;               // (N, T) workspace: the state of the traced path per block
    float *prob;                 // (N, T) workspace: per-block probability of the path k-mer
    int8_t *seq, *qstr;          // (N, T): ASCII at the emitting blocks, 0 elsewhere
    uint8_t *moves;              // (N, T)
    float *score;                // (N) or nullptr
    // filled in by launch_beam_search: the LDS images of a block's score row and back-guide row
    int row_floats, beta_off, pre_stride, row_vec16, beta_vec16;
};
hipError_t launch_beam_search(const BeamParams &p, hipStream_t stream);

// ---------------------------------------------------------------- encoder (xb_encoder.hip)

// conv1(1->4,k5,p2)+SiLU, conv2(4->16,k5,p2)+SiLU, then the im2col rows of conv3
// (row (t*N+n), col c*winlen+k = a2[c][t*stride - winlen/2 + k]) as split fp16, K padded to kp.
struct ConvFrontParams {
    const float *signal;   // (N, L); with signal2 set: chunks [0, split) come from signal, chunks [split, N) from signal2
    const float *signal2;  // (N - split, L) or nullptr: the second of two batches that share one pass through the encoder
    int split;
    int N, L, T, winlen, stride, kp;
    const float *w1, *b1;  // (4,1,5), (4)
    const float *w2, *b2;  // (16,4,5), (16)
    half_t *a_hi, *a_lo;   // (T*N, kp); a_lo holds the q8 image instead when q8 != 0
    int q8;                // 1: second part = fp8 image (see "q8 image" below), activation exponent 0
};
hipError_t launch_conv_front(const ConvFrontParams &p, hipStream_t stream);

enum GemmEpilogue { EPI_BIAS_F32 = 0, EPI_SILU_SPLIT = 1, EPI_TANH_SCALE = 2 };

// q8 image (precision XB_PREC_F16F8): the second part of every operand is not the fp16 residual but, byte for byte in
// its place, per row and per 32 columns a 64-byte block [h8: 32 x e4m3 of hi * 2^e | l8: 32 x e4m3 of lo * 2^(e+11)]
// (e = the tensor's exponent: 8 for LSTM outputs |h| < 1, 0 for the conv tensors, chosen per weight tensor from max|W|).
// One block-scaled MFMA (K = 64) then computes BOTH correction products of 32 columns: lanes 0-31 feed Ah8 x Bl8,
// lanes 32-63 feed Al8 x Bh8, both with the scale 2^-(ea + eb + 11).

// D[m][n] = sum_k A[m][k] * B[n][k]  (+ epilogue); A, B split fp16 (hi, lo / q8), K multiple of 32.
struct GemmParams {
    const half_t *a_hi, *a_lo;   // (M, lda)
    const half_t *b_hi, *b_lo;   // (Nn, ldb)
    int M, Nn, K, lda, ldb;
    const float *bias;           // (Nn) or nullptr
    float *out_f32;              // EPI_BIAS_F32 / EPI_TANH_SCALE : (M, ldc)
    half_t *out_hi, *out_lo;     // EPI_SILU_SPLIT : (M, ldc)
    int ldc;
    float scale;                 // EPI_TANH_SCALE
    int nb, expand;              // EPI_TANH_SCALE : insert blank column in front of every nb outputs
    float blank;
    int nsplit;                  // 3 = hi*hi + hi*lo + lo*hi ; 1 = hi*hi only ; 2 = hi*hi + fp8 corrections (q8 images)
    int a_exp, b_exp;            // nsplit == 2: exponents of the A and B q8 images
    int out_exp;                 // EPI_SILU_SPLIT, q8 output: exponent of the q8 image written to out_lo
    int out_fmt;                 // EPI_SILU_SPLIT: second part of the output -- 0: this GEMM's own form (q8 image when nsplit == 2,
                                 // else the fp16 residual), 1: fp16 residual, 2: q8 image (what the CONSUMING GEMM's arithmetic reads)
    int gin_n;                   // EPI_BIAS_F32: > 0 selects the member-major gin layout (below) with gin_n chunks per time step
    // gemm4p_kernel (two workgroups per CU, B straight from a fragment-major weight image, see xb_encoder.hip): used when b4
    // is set; otherwise gemm8r_kernel (one 256 x 256 workgroup per CU, both operands through LDS; kept as the A/B reference)
    const unsigned char *b4;     // [K / 32][rows4 / 32][pieces][64 lanes][16 B], rows4 = Nn rounded up to 256 (zero rows)
    size_t b4_kstride;           // bytes between consecutive k-tiles of b4 = rows4 / 32 * pieces * 1024
    int one_per_cu;              // gemm4p: 1 = at most one workgroup per CU (launches beside the recurrence)
    int sn;                      // gemm4p: N tiles per XCD super-tile (0 = the rule gemm_super_n; XB_GEMM_SN, experiments)
};
// XB_GEMM_S16 (compile time, both GEMM kernels and the host's weight images): 1 (default, round 5) = the three-product arithmetic
// (nsplit 3) runs on v_mfma_f32_16x16x32_f16 -- per accumulator and k-tile of 32: lo*hi, hi*lo, hi*hi --, 0 = on 32x32x16 (per
// k-step of 16) as in rounds 1-4.  The two sum the same products in different orders (low-bit differences); the chip holds a
// 13 % higher clock on the 16x16x32 shape (profiles/r05_mfma_shape_ubench.txt).
#ifndef XB_GEMM_S16
#define XB_GEMM_S16 1
#endif
// pieces per 32-row block and k-tile of the fragment-major image for a given nsplit (lane l = 32 h + r holds row r):
//   piece 0, 1: the 8 fp16 `hi` values of columns 32 kt + 16 ks + 8 h .. + 8, ks = 0, 1
//   nsplit 3: piece 2, 3: the same of `lo`;   nsplit 2: pieces 2, 3 = bytes 0..15 / 16..31 of the q8 half the B role reads
//   (h = 0: the l8 codes of the 32 columns, h = 1: the h8 codes)
//   nsplit 3 with XB_GEMM_S16: lane l = 16 g + r; piece 2 part + c (part 0 = hi, 1 = lo; c = 0, 1) holds row 16 c + r's eight
//   values of columns 32 kt + 8 g .. + 8
inline int gemm4_pieces(int nsplit) { return nsplit == 1 ? 2 : 4; }
// gin layout (input projection of an LSTM layer, written by the GEMM, read by lstm_kernel): row m = t * n + chunk, column
// c = unit * 4 + gate.  Stored member-major, [t][c / 128][chunk][c % 128]: the 64 chunks x 128 gate columns a recurrence
// workgroup (32 units) needs per step are ONE contiguous 32 KiB block instead of 64 segments 4F floats apart.
__host__ __device__ inline size_t gin_offset(size_t m, int c, int n, int cols)
{
    const size_t t = m / (size_t)n, chunk = m % (size_t)n;
    return ((t * (size_t)(cols / 128) + (size_t)(c >> 7)) * (size_t)n + chunk) * 128 + (size_t)(c & 127);
}
hipError_t launch_gemm(const GemmParams &p, int epilogue, hipStream_t stream);

// One LSTM layer's recurrence over a slab of chunks.  Gate pre-activations of the input
// projection (+ both biases) are in `gin` with gate-interleaved columns (col = unit*4 + gate,
// gates i,f,g,o); w_hh rows are in the same order.  y receives h_t as split fp16.
struct LstmParams {
    const float *gin;            // (T, N, 4F)
    const half_t *w_hi, *w_lo;   // (4F, F) gate-interleaved rows
    half_t *y_hi, *y_lo;         // (T, N, F)
    float *c_state;              // (N, F) fp32 cell state (in/out across launches)
    half_t *xh;                  // exchange buffer [groups][2 parity][2 parts][64][F] (h of the previous step)
    int T, N, F;
    int n0, nslab;               // chunks [n0, n0+nslab) are processed by this launch
    int grp0;                    // index of this launch's first group in the exchange buffer and the counters (n0 / 64 when the
                                 // whole batch fits the 64 group slots, so that h and the counters survive between the launches
                                 // of different chunk slabs and time slabs; else 0)
    int reverse;                 // time runs T-1..0
    int s_begin, s_end;          // recurrence steps [s_begin, s_end) of this launch (s = 0 is the first step)
    int persistent;              // 1: all steps in one launch with inter-workgroup sync
    unsigned *sync;              // per-group monotonic arrival counters, 64 words per 64-chunk group slot (zeroed once per layer and chunk slab)
    unsigned sync_base;          // arrivals per member already counted by earlier launches of this layer (time slabs)
    unsigned *error;             // set non-zero when a sync wait timed out
    int nsplit;                  // as GemmParams::nsplit (2: w_lo, y_lo and the exchange "lo" part are q8 images, h exponent 8);
                                 // 4: int8-limb recurrence (wq1, wq0, wscale below; y stays hi + q8 image for the next GEMM);
                                 // 5: the same without the d0 x d0 product
    const int8_t *wq1, *wq0;     // nsplit == 4: (4F, F) balanced signed digits of round(W_hh / row scale * 32512), gate-interleaved rows
    const float *wscale;         // nsplit == 4: (4F) row scale / 32512^2: the factor that turns the integer digit sums into W_hh h
    int w_exp;                   // nsplit == 2: exponent of the W_hh q8 image
    int y_alt;                   // 1 (nsplit 2 or 3): y_lo receives the OTHER second part than the exchange image -- the fp16
                                 // residual when nsplit == 2, the q8 image (exponent 8) when nsplit == 3 -- for a next GEMM that
                                 // runs in the other arithmetic
    int spread;                  // 1: spread each group's members over all XCDs (placement-independence test)
    int dual;                    // 1: a workgroup serves two groups alternately (a launch then holds twice the groups)
    int quad;                    // 1 (with dual, F = 768, nsplit 2, persistent): the software-pipelined kernel of xb_lstm_quad.h -- four
                                 // groups of 32 chunks per workgroup, the gate math of one group-step between the MFMAs of the next; the
                                 // exchange buffer and the counters are then indexed in 32-chunk groups (slot 2 grp0 + g, 128 slots)
    int slab;                    // index of this launch among the layer's time slabs (selects the byte of the XCD mask below)
    int xcd_local;               // 1: members prove per launch that their group sits on one XCD (words 1..4 of the group's sync
                                 // slot, one byte per time slab, zeroed with the counters) and then exchange h with plain stores
    // One launch over all steps that reports its time slabs (sig_flag != nullptr): the layer output is stored write-through,
    // and when the last workgroup has finished slab i (steps [T i / sig_nts, T (i + 1) / sig_nts)) it stores sig_base + i + 1
    // to *sig_flag -- the word the GEMM stream waits on (hipStreamWaitValue32) before it consumes that slab.
    unsigned *sig_flag;          // device word, monotonic across layers and batches
    unsigned *sig_done;          // sig_nts arrival counters, zeroed with the group counters
    unsigned sig_base;
    int sig_nts;
};
hipError_t launch_lstm(const LstmParams &p, hipStream_t stream);
// members (workgroups per group) and chunks per group of the LSTM kernel for feature size F
// workgroups of the persistent kernel the occupancy calculator admits per CU for feature size F (0: the kernel cannot be
// resident at all, e.g. LDS or registers taken by another tenant's limits); the persistent mode needs >= 1
int lstm_resident_per_cu(int F, int nsplit, int dual);
int lstm_quad_resident_per_cu();   // the same for the software-pipelined kernel (F = 768, nsplit 2)
int lstm_members(int F);
int lstm_group_chunks();
bool lstm_supported_features(int F);
#ifdef XB_LSTM_STAMPS
void lstm_read_stamps(unsigned long long out[10], bool reset);   // diagnostic build only
void gemm_read_stamps(unsigned long long out[8], bool reset);    // diagnostic build only (XB_GEMM_STAMPS)
#endif

}  // namespace xb

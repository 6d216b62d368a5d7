// xb_align.hip -- host-side accuracy of a called sequence against its reference (no device code).
//
// `bonito evaluate` scores every call with util.accuracy (ub-bonito/bonito/util.py:402-424):
//     parasail.sw_trace_striped_32(seq, ref, 8, 4, parasail.dnafull) -> CIGAR -> '=' / ('=' + 'I' + 'X' + 'D') * 100,
//     0 when the alignment (its columns, insertions included: util.py:410) is shorter than min_coverage of the reference.
// parasail is a third-party CPU library that no image here has; this is a restatement of what that call computes -- a
// Smith-Waterman local alignment with affine gaps (a gap of length k costs 8 + 4 (k - 1)), match + 5 / mismatch - 4 (the
// A, C, G, T block of NUC.4.4 = parasail.dnafull) -- with two stated choices where the restatement cannot be pinned:
//   * the extended letters X and Y score as ordinary letters (+ 5 / - 4).  parasail's dnafull would read 'Y' as the IUPAC
//     pyrimidine code and has no 'X'; the reference's own XNA accuracies come from analyze_paf.py, not from this function;
//   * among equal-score predecessors the trace-back prefers the diagonal, then a deletion (gap in the query), then an
//     insertion; the end cell is the first maximum in row-major order.
// It is post-processing on the host, like the reference's: nothing of the basecalling path runs here.
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../../include/xna_basecaller.h"

extern "C" XB_API int xb_align_accuracy(const char *ref, int ref_len, const char *seq, int seq_len, double min_coverage,
                                        int balanced, double *accuracy, int32_t counts[4])
{
    if (!ref || !seq || !accuracy || ref_len < 0 || seq_len < 0) return XB_ERR_INVALID;
    int32_t c4[4] = {0, 0, 0, 0};                       // '=', 'X', 'I' (in the query only), 'D' (in the reference only)
    *accuracy = 0.0;
    if (counts) memcpy(counts, c4, sizeof c4);
    if (ref_len == 0 || seq_len == 0) return XB_OK;
    const int OPEN = 8, EXT = 4, MATCH = 5, MIS = -4;
    const int n = seq_len, m = ref_len;                 // rows: query (seq), columns: reference
    const size_t W = (size_t)m + 1;
    // three full (n + 1) x (m + 1) int32 matrices (the trace-back needs them): bounded -- `bonito evaluate` aligns chunk-level
    // calls of a few thousand bases; 2^26 cells = 768 MiB is far beyond that and far below what would exhaust the host
    if ((size_t)(n + 1) * W > ((size_t)1 << 26)) return XB_ERR_NOMEM;
    const int NEG = -(1 << 28);
    // H: best score ending at (i, j); E: ... with a gap in the query (deletion, consumes ref); F: ... gap in the ref (insertion)
    std::vector<int32_t> H((size_t)(n + 1) * W, 0), E((size_t)(n + 1) * W, NEG), F((size_t)(n + 1) * W, NEG);
    int best = 0, bi = 0, bj = 0;
    for (int i = 1; i <= n; ++i) {
        for (int j = 1; j <= m; ++j) {
            const size_t k = (size_t)i * W + j;
            E[k] = std::max(E[k - 1] - EXT, H[k - 1] - OPEN);
            F[k] = std::max(F[k - W] - EXT, H[k - W] - OPEN);
            const int d = H[k - W - 1] + (seq[i - 1] == ref[j - 1] ? MATCH : MIS);
            int h = std::max(0, d);
            h = std::max(h, std::max(E[k], F[k]));
            H[k] = h;
            if (h > best) { best = h; bi = i; bj = j; }
        }
    }
    if (best == 0) return XB_OK;
    // trace back from (bi, bj) until a cell of score 0
    int i = bi, j = bj, state = 0;                       // 0: in H, 1: in E, 2: in F
    while (i > 0 && j > 0) {
        const size_t k = (size_t)i * W + j;
        if (state == 0) {
            if (H[k] == 0) break;
            const int d = H[k - W - 1] + (seq[i - 1] == ref[j - 1] ? MATCH : MIS);
            if (H[k] == d) { c4[seq[i - 1] == ref[j - 1] ? 0 : 1] += 1; --i; --j; }
            else if (H[k] == E[k]) state = 1;
            else state = 2;
        } else if (state == 1) {
            c4[3] += 1;
            state = (E[k] == H[k - 1] - OPEN) ? 0 : 1;
            --j;
        } else {
            c4[2] += 1;
            state = (F[k] == H[k - W] - OPEN) ? 0 : 2;
            --i;
        }
    }
    if (counts) memcpy(counts, c4, sizeof c4);
    // util.py:410: r_coverage = len(alignment.traceback.ref) / len(ref) -- the traceback string carries a '-' for every column
    // the reference does not take part in, so its length is the number of alignment COLUMNS ('=', X, D and I), not of
    // reference bases
    const int columns = c4[0] + c4[1] + c4[2] + c4[3];
    if ((double)columns / (double)m < min_coverage) return XB_OK;
    const double den = balanced ? (double)(c4[0] + c4[1] + c4[3]) : (double)(c4[0] + c4[1] + c4[2] + c4[3]);
    const double num = balanced ? (double)(c4[0] - c4[2]) : (double)c4[0];
    *accuracy = den > 0 ? 100.0 * num / den : 0.0;
    return XB_OK;
}

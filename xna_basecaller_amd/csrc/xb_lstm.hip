// xb_lstm.hip -- the LSTM recurrence for gfx950 (split from xb_encoder.hip in round 5: the two build in parallel, and the
// recurrence can take compiler flags of its own -- see LSTM_FLAGS in the Makefile for what was measured).
//
// Replaces the LSTM x5 of Model.forward (ub-bonito/bonito/nn.py:176-193,216-220; crf/model.py:147-160).
#include "xb_enc_common.h"

namespace {
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ======================================================================================
// LSTM recurrence.
// A *group* = LG_BN chunks; its F/32 member workgroups each own 32 hidden units (128
// gate-interleaved rows of W_hh: wave w holds rows [32w, 32w+32) = units 8w..8w+7 as MFMA
// A-operand fragments in registers for the whole launch).  Per step a member needs the whole
// h_{t-1} of its group's chunks.  The members exchange h through a small ping-pong buffer
// xh[parity][group][part][64 chunks][F] (a few MB, L2/Infinity-Cache resident) and, beside it,
// write the layer output y (T,N,F) that the next layer consumes after the launch.
// The exchange rows are pulled into LDS by LDS-DMA in K pieces of KP columns, double buffered
// against the MFMAs, B fragments software-pipelined one k-step ahead.
// MFMA tile orientation: rows = gate rows (one lane owns i,f,g,o of a unit in four consecutive
// accumulator registers), columns = chunks.
// persistent = 1: all steps in one launch.  Hand-off protocol per step (cdna_hip_programming.md
// Guideline 16, form R1): exchange stores are 16-byte write-through (sc1) stores; every storing
// wave drains them (s_waitcnt vmcnt(0)); workgroup barrier; ONE lane adds to the group's
// monotonic agent-scope counter.  Consumer: ONE lane polls that counter with relaxed sc1 loads
// (bounded spin), workgroup barrier, then EVERY load of the exchanged bytes is an sc1 load
// (LDS-DMA with the sc1 bit), so no L1 line can be stale and no fence is needed.
// ======================================================================================
#ifndef XB_LSTM_DMA_ASM          // 1: the exchange pieces' LDS-DMA requests as inline asm (see dma16_sc1); 0: the builtin (A/B builds)
#define XB_LSTM_DMA_ASM 1
#endif
constexpr int LG_BN = 64;        // chunks per group (2 MFMA column tiles)
constexpr int LG_UNITS = 32;     // hidden units per member workgroup
constexpr int LG_SYNC = 64;      // words between the counter slots of consecutive 64-chunk groups (lstm_quad_kernel's 32-chunk groups: 32)
constexpr unsigned long long LG_SPIN_CYCLES = 4000000000ull;   // ~2 s at 2 GHz
constexpr int CPOL_SC1 = 16;     // gfx940+ cache-policy immediate: sc0 = 1, nt = 2, sc1 = 16
constexpr int ST_LD = 68;        // dword stride of one unit-pair row of the h staging (64 chunks + 4: 2-way reads)

// (Inline asm, not __builtin_amdgcn_global_load_lds: hipcc books an LDS-DMA as an LDS event of the lgkm counter, and with two
// kinds of events pending it can no longer count -- every wait for a B fragment in the MFMA loop became lgkmcnt(0), i.e. the
// fragment reads issued one k-step AHEAD were waited for at once and their latency (~120 cycles per k-step pair, ~3 k cycles per
// group-step) sat on the critical path.  Hidden from the compiler, the requests leave its bookkeeping alone and it emits the
// counted waits the software pipeline needs; completion is the explicit s_waitcnt vmcnt(0) + barrier that closes every piece.)
__device__ __forceinline__ void dma16_sc1(const void *g, void *lds_wave_base)
{
#if XB_LSTM_DMA_ASM
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off sc1" ::"v"(g), "s"(m0v) : "memory", "m0");
#else
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, CPOL_SC1);
#endif
}
// the same with a wave-uniform base (an SGPR pair) and a 32-bit per-lane byte offset: no 64-bit address arithmetic per request
__device__ __forceinline__ void dma16_sc1_off(const void *ubase, int byte_off, void *lds_wave_base)
{
#if XB_LSTM_DMA_ASM
    const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(byte_off), "s"(ubase), "s"(m0v) : "memory", "m0");
#else
    dma16_sc1(reinterpret_cast<const unsigned char *>(ubase) + byte_off, lds_wave_base);
#endif
}

// Round 5: the LEAN request.  The kernel is bound by its one wave's instruction COUNT (every instruction, scalar ones included,
// takes an issue slot of >= 4 cycles), and a request used to cost seven: 64-bit base add (2), generic -> LDS pointer cast (2: a
// null check), s_mov m0, s_nop, the load.  Here it costs three: the wave's LDS base (an SGPR) plus a literal straight into m0, the
// lane's byte offset plus a literal (the VALU add doubles as the wait state m0 needs before an LDS-DMA), and the load from ONE
// wave-uniform base (SGPR pair, per group-step); the immediate offset stays 0 (it would move the LDS address as well).  The
// literals are "n" operands: the callers' loop variables are constants after unrolling.
__device__ __forceinline__ void dma16_lean_sc1(unsigned vlane, int vconst, const void *sbase, unsigned lds_wave, int lconst, int ioff)
{
    unsigned t;
    asm volatile("s_add_i32 m0, %[lb], %[lc]\n\tv_add_u32 %[t], %[vc], %[vo]\n\tglobal_load_lds_dwordx4 %[t], %[sb] offset:%[io] sc1"
                 : [t] "=&v"(t) : [lb] "s"(lds_wave), [lc] "n"(lconst), [vc] "n"(vconst), [vo] "v"(vlane), [sb] "s"(sbase), [io] "n"(ioff)
                 : "memory", "m0");
}
__device__ __forceinline__ void dma16_lean_nt(unsigned vlane, int vconst, const void *sbase, unsigned lds_wave, int lconst)
{
    unsigned t;
    asm volatile("s_add_i32 m0, %[lb], %[lc]\n\tv_add_u32 %[t], %[vc], %[vo]\n\tglobal_load_lds_dwordx4 %[t], %[sb] nt"
                 : [t] "=&v"(t) : [lb] "s"(lds_wave), [lc] "n"(lconst), [vc] "n"(vconst), [vo] "v"(vlane), [sb] "s"(sbase)
                 : "memory", "m0");
}

// 16-byte plain store: the line stays in the XCD's L2 (same asm form as the write-through one below)
__device__ __forceinline__ void store16_l2(void *g, uint4 v)
{
    const u32x4 d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(g), "v"(d) : "memory");
}

// 16-byte write-through store (the `s_nop 1` keeps the data registers intact until the store has read them)
__device__ __forceinline__ void store16_sc1(void *g, uint4 v)
{
    const u32x4 d = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(g), "v"(d) : "memory");
}

#ifdef XB_LSTM_STAMPS
// diagnostic build only: per-phase cycle sums of workgroup 0 (never compiled into the product).  The sums are kept
// in LDS (no vector-memory traffic, so the stamps neither drain vmcnt nor absorb store latencies) and copied out
// once at the end of the kernel.
__device__ unsigned long long g_lstm_stamps[10];   // 0..7 cycle sums, 8 = early first-piece requests, 9 = group-steps
#define XB_STAMP(i)                                                                         \
    do {                                                                                    \
        const unsigned long long now_ = __builtin_readcyclecounter();                       \
        if (tid == 0) sStamp[(i)] += now_ - stamp_prev;                                      \
        stamp_prev = now_;                                                                  \
    } while (0)
#else
#define XB_STAMP(i) do { } while (0)
#endif

// DUAL = true: one workgroup serves TWO groups (g and g + gh) alternately with the same W_hh registers.  A group-step's
// hand-off (stores reaching L2, the other members' arrivals, the poll) then completes while the workgroup runs the other
// group's step, the first h piece of the coming group-step is requested before the gate math of the current one, and
// the gin tile is requested a whole group-step ahead: a launch holds twice the chunks at the same residency.
#ifndef XB_LSTM_DMA_SPREAD       // 1: every LDS-DMA request of the MFMA loop directly behind one MFMA; 0 (default): in pairs behind the k-step's
                                 // MFMAs -- measured on one box (profiles/r04_lstm_loop_ab.txt): no gain on top of XB_LSTM_DMA_ASM, a loss with one group per workgroup
#define XB_LSTM_DMA_SPREAD 0
#endif
#ifndef XB_LSTM_DEFER_ARRIVE     // 1 (default): two groups per workgroup -- the arrival of a group-step is issued behind the first
                                 // piece-closing drain + barrier of the OTHER group's step instead of behind a drain of its own (A/B builds: 0)
#define XB_LSTM_DEFER_ARRIVE 1
#endif
#ifndef XB_LSTM_LEAN             // the three-instruction LDS-DMA requests (dma16_lean_sc1): 1 everywhere, 2 only with two groups per workgroup,
                                 // 0 nowhere (the seven-instruction form of rounds 1-4).  Measured (profiles/r05_lstm_lean_ab.txt): with two groups
                                 // per workgroup -5 % per launch; with ONE group the launch gets 4-6 % SLOWER -- its critical path is the hand-off
                                 // chain, and the gin / first-piece requests bunched into a third of the issue time fill the CU's request queue
#define XB_LSTM_LEAN 2
#endif
#ifndef XB_LSTM_PKGATE           // 1: the gate math on PAIRS of cells with the packed fp32 VALU (v_pk_mul / v_pk_add / v_pk_fma_f32: two lanes' worth of
                                 // work per issue slot) -- the same IEEE operations in the same order per cell, so the results do not change;
                                 // 0: one cell at a time (rounds 1-4; A/B builds).  Not combined with XB_LSTM_GIN_SPREAD.
#define XB_LSTM_PKGATE 1
#endif
#ifndef XB_LSTM_RING3            // 1: two groups per workgroup, piece count a multiple of three -- the pieces go through a ring of THREE buffers and are
                                 // requested up to two pieces ahead (see R3 in lstm_kernel); 0 (default): two buffers, one piece ahead.  Measured:
                                 // bit-identical and NEUTRAL (28.43-28.55 vs 28.51-28.56 ms per paired launch, profiles/r05_lstm_lean_ab.txt 8):
                                 // what a piece's closing waits for is not its successor's landing
#define XB_LSTM_RING3 0
#endif
#ifndef XB_LSTM_GIN_SPREAD       // 1: two groups per workgroup (no counted drain behind the exchange stores) -- the eight gin requests of the
                                 // coming step go out ONE PER CELL inside the gate math instead of back to back behind the exchange stores, where each
                                 // found the CU's request queue still full of its predecessors and held the wave ~50 cycles.  Measured: no gain
                                 // (29.16-29.35 vs 29.06-29.13 ms per paired launch, profiles/r05_lstm_lean_ab.txt) -> 0 (default)
#define XB_LSTM_GIN_SPREAD 0
#endif
#ifndef XB_LSTM_ONE_WAIT         // 1 (default, round 5): ONE counted LDS wait per k-step in front of its MFMAs; 0: hipcc's own wait in front of every MFMA (A/B builds)
#define XB_LSTM_ONE_WAIT 1
#endif
#ifndef XB_LSTM_CPREFETCH        // 1 (default, round 5): the lane's eight cell states are requested together in front of the gate math; 0: each at its use (A/B builds)
#define XB_LSTM_CPREFETCH 1
#endif
#ifndef XB_LSTM_GIN_DIRECT       // 1 (default, round 5): the input projection goes from global memory STRAIGHT INTO THE ACCUMULATORS (plain
                                 // loads one group-step ahead, an L2 prefetch two ahead); 0: through a 32 KiB LDS tile per group by LDS-DMA (rounds 1-4)
#define XB_LSTM_GIN_DIRECT 0
#endif
#ifndef XB_GIN_NT
#define XB_GIN_NT 1
#endif
#ifndef XB_GIN_PREFETCH
#define XB_GIN_PREFETCH 1
#endif
#ifndef XB_LSTM_PIPE_PIECES      // 1: the k-step pipeline runs across the piece boundary (see PIPE in lstm_kernel); with two piece buffers it measured
                                 // 4 % SLOWER (profiles/r05_lstm_lean_ab.txt: the piece is then closed a k-step earlier and waits longer for its successors DMA)
#define XB_LSTM_PIPE_PIECES 0
#endif
#ifdef XB_NO_SIGNAL
#define XB_SIG(x) false
#else
#define XB_SIG(x) (x)
#endif
// YALT = true (NSPLIT 2 or 3 only): the layer output y carries the OTHER second part than the exchange image -- the fp16
// residual when the recurrence itself runs on q8 images (NSPLIT 2), the q8 image when it runs on residuals (NSPLIT 3) -- because
// the GEMM that consumes y runs in the other arithmetic (mixed-precision encoders: xb_api.hip stage_nsplit).  The extra
// staging rows live in piece buffer 1, which is idle between the last piece's closing barrier and the next group-step's
// piece-1 requests.
template <int KS, int NSPLIT, bool DUAL, bool YALT = false>
__global__ __launch_bounds__(256) void lstm_kernel(xb::LstmParams p)
{
    static_assert(!YALT || NSPLIT == 2 || NSPLIT == 3, "an alternative y image exists for the q8 and the residual arithmetic only");
    constexpr int F = KS * 16;
    constexpr int KP = F < 128 ? F : 128;       // columns per piece
    constexpr int NP = F / KP;                  // pieces per step
    constexpr int KSP = KP / 16;                // MFMA k-steps per piece
    // NSPLIT == 4: the int8-limb recurrence.  h (|h| < 1) and each W_hh row (per-row scale) are 16-bit fixed point, split
    // into two balanced signed 8-bit digits q = 256 d1 + d0; the four digit products run on v_mfma_i32_32x32x32_i8 (exact
    // int32 sums, three accumulator sets by weight 2^16 / 2^8 / 1) and are combined in fp32 once per step.  The exchange
    // image is one byte per element and part (part 0 = d1, part 1 = d0): half the DMA and fragment bytes of fp16 + q8.
    constexpr bool I8 = NSPLIT == 4 || NSPLIT == 5;
    constexpr bool LOLO = NSPLIT == 4;          // NSPLIT == 5: without the d0 x d0 product (2^-16 of the leading one per term)
    constexpr int ES = I8 ? 1 : 2;              // bytes per element of one exchange part
    constexpr int CPR = KP * ES / 16;           // 16-byte cells per row per piece
    constexpr int SWZ = (CPR & -CPR) - 1;       // XOR mask that stays inside the row
    constexpr int NPARTS = NSPLIT == 1 ? 1 : 2; // hi (, lo or, NSPLIT == 2, the q8 image; NSPLIT == 4: the two digits)
    constexpr int PIECE_BYTES = LG_BN * KP * ES; // one part of one piece
    constexpr int STP = I8 ? 3 : NPARTS;        // staging arrays: hi pairs, lo / q8 (, NSPLIT == 4: the digit bytes)
    static_assert(!I8 || KP == 128 || KP == 64, "int8-limb pieces are 64 or 128 columns (at least four cells per row)");
    constexpr int NG = DUAL ? 2 : 1;            // groups per workgroup
    static_assert(F % KP == 0, "feature size must be a multiple of the piece width");

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char *sPiece = smem_raw;                                           // [2][NPARTS][PIECE_BYTES]
    // h staging for the 16-byte row stores: packed unit pairs, [NPARTS][16 pairs][ST_LD dwords] (chunk minor)
    unsigned *sT = reinterpret_cast<unsigned *>(smem_raw + 2 * NPARTS * PIECE_BYTES);
    static_assert(!YALT || NPARTS * PIECE_BYTES >= 16 * ST_LD * 4, "piece buffer 1 holds the alternative y staging");
    float *sC0 = reinterpret_cast<float *>(sT + STP * 16 * ST_LD);                // [NG][32 units][64 chunks] cell state
    // input-projection tile of the step: [NG][64 chunks][32 cells of 16 B = the four gates of one unit], cell XOR (chunk & 31)
    unsigned char *sG0 = reinterpret_cast<unsigned char *>(sC0 + NG * LG_UNITS * LG_BN);
    constexpr unsigned OFF_G = 2 * NPARTS * PIECE_BYTES + STP * 16 * ST_LD * 4 + NG * LG_UNITS * LG_BN * 4;   // of sG0 in the block
    constexpr bool GDIR = XB_LSTM_GIN_DIRECT != 0;
    // GDIR: no tile -- what is left at sG0 is 1 KiB that the L2 prefetch requests write into (never read)
    constexpr unsigned G_TILE = GDIR ? 0 : LG_BN * LG_UNITS * 16;                                         // one group's gin tile
    int *sFlag = reinterpret_cast<int *>(sG0 + (GDIR ? 1024 : NG * LG_BN * LG_UNITS * 16));
    // ---- R3 (round 5): a ring of THREE piece buffers without a byte of extra LDS.  With two groups per workgroup each group has a gin
    // tile buffer of 32 KiB -- and a group's tile is dead from the moment its accumulators have been read at the top of the
    // group-step until the next tile is requested at its end.  In between the buffer is the third piece buffer: piece pc sits in
    // slot pc % 3 (piece buffers 0, 1, the current group's tile buffer) and is requested up to TWO pieces ahead:
    //     during piece 0: nothing (the tile buffer is still being read by slower waves until this piece's closing barrier)
    //     during piece 1: piece 2 (first half of the k-steps) and piece 3 (second half);   during piece pc >= 2: piece pc + 2,
    //     i.e. during the last two pieces the first two pieces of the coming group-step (when the look-ahead poll, one piece
    //     earlier than before, has seen the other group's members arrive).
    // A piece then has two piece times to land instead of one (piece 2: one), and what closes a piece is a COUNTED wait that leaves
    // the requests issued behind its successor's in flight -- the previous group-step's gin tile included, which the first
    // closing used to wait for.  The alternative y staging (YALT) moves from piece buffer 1 (no longer idle during the gate math)
    // into the group's tile buffer, and the tile request moves behind the barrier that ends the staging reads.
    constexpr bool R3 = XB_LSTM_RING3 != 0 && DUAL && !I8 && !GDIR && (XB_LSTM_LEAN == 2 || XB_LSTM_LEAN == 1) &&
                        (CPR & (CPR - 1)) == 0 && NP >= 3 && NP % 3 == 0 && G_TILE >= (unsigned)(NPARTS * PIECE_BYTES);
    unsigned *sTy = reinterpret_cast<unsigned *>(smem_raw + NPARTS * PIECE_BYTES);   // YALT: [16][ST_LD] in piece buffer 1 (R3: re-pointed per group, see serve)
    // DUAL: the first W_hh fragment lives in LDS (16 B per thread behind the flags and stamps) and is read back at the top of
    // every group-step: with all 512 registers taken hipcc otherwise parks half of it in scratch, and the reload -- a
    // scratch load with vmcnt(0) behind it -- would wait for the other group's gin tile and y stores still in flight
    unsigned char *sW0 = reinterpret_cast<unsigned char *>(sFlag) + 16 + 80;
    float *sScale = reinterpret_cast<float *>(sW0 + 256 * 16);                  // NSPLIT == 4: row scales of the 128 gate rows
    // DUAL, even piece count: the coming group-step's first piece is requested in the DMA-free issue slots of the last piece
    constexpr bool EIL = DUAL && NP >= 2 && NP % 2 == 0;
#ifdef XB_LSTM_STAMPS
    constexpr bool PARK = !I8;          // the stamp bookkeeping costs the single-group kernel the same registers
#else
    constexpr bool PARK = DUAL && !I8;
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform
    // LDS byte addresses as 32-bit integers (a generic pointer cast to the LDS address space costs a null check per use)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)smem_raw;
    const unsigned lds_w = lds0 + (unsigned)wid * 1024u;       // this wave's 1 KiB slot of a request group (wave-uniform: an SGPR)
    constexpr bool LEAN = XB_LSTM_LEAN == 2 ? DUAL : XB_LSTM_LEAN != 0;
    const int members = F / LG_UNITS;
    const int ngroups = (p.nslab + LG_BN - 1) / LG_BN;
    const int gh = DUAL ? (ngroups + 1) / 2 : ngroups;          // workgroup slots: slot g serves group g (and g + gh)
    const int g8 = (gh + 7) & ~7;
    // default: blocks b, b+8, b+16.. (one XCD under round-robin dispatch) form a group -- speed only.
    // spread = 1 deals a group's members over consecutive blocks, i.e. over all XCDs (placement test).
    const int grp = p.spread ? (int)blockIdx.x / members : (int)blockIdx.x % g8;
    const int mb = p.spread ? (int)blockIdx.x % members : (int)blockIdx.x / g8;
    if (grp >= gh) return;
    const bool second = DUAL && grp + gh < ngroups;             // this slot has a second group
    const int N = p.N, T = p.T;
    const int nlast = p.n0 + p.nslab - 1;
    const int hsel = lane >> 5;
    const int ubase = mb * LG_UNITS + wid * 8;   // first unit of this wave

    // ---- W_hh fragments: row = gate-interleaved (unit*4+gate), lane l: row (l&31), k-chunk (l>>5)
    // NSPLIT == 2: per 32 columns one q8 fragment instead of two lo fragments -- lanes 0-31 hold the block's Wh8 half,
    // lanes 32-63 its Wl8 half (A operand of the block-scaled MFMA; the h fragments below take the opposite halves)
    half8 wh[I8 ? 1 : KS], wl[I8 ? 1 : KS];
    v8i wq[KS / 2 > 0 ? KS / 2 : 1];
    // NSPLIT == 4: per 32 columns the lane's 16 bytes of each digit of its row (k = 32 b + 16 hsel + byte)
    v4i wd1[I8 ? KS / 2 : 1], wd0[I8 ? KS / 2 : 1];
    if constexpr (I8) {
        const size_t row = (size_t)ubase * 4 + (lane & 31);
#pragma unroll
        for (int b = 0; b < KS / 2; ++b) {
            wd1[b] = *reinterpret_cast<const v4i *>(p.wq1 + row * F + b * 32 + hsel * 16);
            wd0[b] = *reinterpret_cast<const v4i *>(p.wq0 + row * F + b * 32 + hsel * 16);
        }
        if (tid < 128) sScale[tid] = p.wscale[(size_t)mb * 128 + tid];
    } else {
        const size_t row = (size_t)ubase * 4 + (lane & 31);
#pragma unroll
        for (int k = 0; k < KS; ++k) {
            wh[k] = *reinterpret_cast<const half8 *>(p.w_hi + row * F + k * 16 + hsel * 8);
            if (NSPLIT == 3) wl[k] = *reinterpret_cast<const half8 *>(p.w_lo + row * F + k * 16 + hsel * 8);
        }
        if (NSPLIT == 2) {
            const unsigned char *wq8 = reinterpret_cast<const unsigned char *>(p.w_lo);
#pragma unroll
            for (int b = 0; b < KS / 2; ++b)
                wq[b] = *reinterpret_cast<const v8i *>(wq8 + (row * F + b * 32) * 2 + hsel * 32);
        }
    }
    const int sca = 127 - p.w_exp, scb = 127 - 8 - 11;     // E8M0 scale bytes: W image exponent; h image exponent 8 (+11)
    if (PARK) *reinterpret_cast<half8 *>(sW0 + tid * 16) = wh[0];

    // ---- the group being served (wave-uniform; re-pointed at every group-step when DUAL)
    int cbase = 0;                 // first chunk of the group
    unsigned *cnt = nullptr;       // its arrival counter
    half_t *xg = nullptr;          // its exchange buffer [parity][part][64 rows][F]
    float *sC = sC0;
    unsigned char *sG = sG0;
    unsigned lds_g = lds_w + OFF_G;   // LDS address of this wave's first 1 KiB of the group's gin tile
    constexpr size_t XPAR = (size_t)2 * LG_BN * F, XPART = (size_t)LG_BN * F;
    auto serve = [&](int gi) {
        const int g = grp + gi * gh;
        cbase = p.n0 + g * LG_BN;
        cnt = p.sync + (size_t)(p.grp0 + g) * LG_SYNC;
        xg = p.xh + (size_t)(p.grp0 + g) * (2 * 2 * LG_BN * F);
        sC = sC0 + gi * (LG_UNITS * LG_BN);
        sG = sG0 + gi * G_TILE;
        lds_g = lds_w + OFF_G + (unsigned)gi * G_TILE;
        if constexpr (R3) sTy = reinterpret_cast<unsigned *>(sG);
    };

    // ---- cell state lives in LDS as [unit][chunk] (the register file is full of W_hh): lane owns
    //      (chunk = 32*nt + (l&31), unit = 8*wid + 2*rg + hsel), only ever touched by that lane
#pragma unroll 1
    for (int gi = 0; gi < NG; ++gi) {
        if (gi == 1 && !second) break;
        serve(gi);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = cbase + nt * 32 + (lane & 31);
            const int ch = n <= nlast ? n : nlast;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                sC[(wid * 8 + 2 * rg + hsel) * LG_BN + nt * 32 + (lane & 31)] =
                    p.c_state[(size_t)ch * F + ubase + 2 * rg + hsel];
        }
    }

    // ---- same-XCD exchange (round 3).  A write-through (sc1) store DROPS the line from the XCD's L2, so every member's h
    //      fetch of the next step starts with a fabric round trip; a plain store keeps it there, and a group's members
    //      normally share an XCD (blocks b, b + 8, .. under round-robin dispatch).  "Normally" is not a contract, so the
    //      members PROVE it per launch: each ORs the bit of the XCD it actually runs on (HW_REG_XCC_ID) into a word of its
    //      group's sync slot; when a member's second wait for the group has completed, every member has posted its bit (the OR
    //      is older than the member's first arrival), and a mask with exactly one bit set switches this workgroup's later
    //      exchange stores to plain ones (the loads stay sc1 = L2-served).  Any other mask -- members on several XCDs, the
    //      placement test -- keeps the write-through form, which is correct anywhere.  One-group-per-workgroup kernel only.
    // (mask word and shift are recomputed from the kernel arguments where they are used: nothing extra stays live in the loop)
    // (two groups per workgroup: both groups have the same member workgroups, each posts into both groups' words and decides
    //  per group at that group's first wait; sFlag[2 + gi])
    if (p.persistent && p.xcd_local && tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;        // HW_REG_XCC_ID[3:0]
        for (int gi = 0; gi < (second ? 2 : 1); ++gi)
            __hip_atomic_fetch_or(p.sync + (size_t)(p.grp0 + grp + gi * gh) * LG_SYNC + 1 + ((p.slab >> 2) & 3), 1u << (8 * (p.slab & 3) + xcc),
                                  __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid == 0) { sFlag[2] = 0; sFlag[3] = 0; }
#ifdef XB_LSTM_STAMPS
    unsigned long long *sStamp = reinterpret_cast<unsigned long long *>(sFlag + 4);
    if (tid == 0) for (int i = 0; i < 10; ++i) sStamp[i] = 0;
    unsigned long long stamp_prev = __builtin_readcyclecounter();
#endif
    // The input projection of a step (gin: 64 chunks x 128 gate rows x 4 B = 32 KiB per workgroup) is pulled into LDS by
    // LDS-DMA one step AHEAD, right after the exchange stores of the previous step: it is in flight during drain / arrive /
    // poll instead of starting at the top of the step (where the polling wave's wait absorbed its whole HBM latency),
    // it needs no registers, and the store drain becomes a counted wait (all but these eight youngest operations).
    // Instruction q = 4 d + wid (d = 0..7) fills chunk rows 2q, 2q + 1: lane i -> row 8 d + 2 wid + (i >> 5), cell i & 31,
    // source cell (i & 31) ^ (row & 7): the lane part of the address is the same for all eight instructions.  Rows
    // past the slab's last chunk read whatever follows (other chunks' rows or the 64 slack rows behind the buffer):
    // their results are never stored.
    auto issue_gin = [&](int tn) {
        // lane part recomputed per call (a few VALU) rather than kept in a register across the MFMA loop, where it would be
        // spilled and its reload (a scratch load + wait) would drain whatever is in flight
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const unsigned gin_lane = (unsigned)(((2 * wid + (lo >> 5)) * 128 + (((lo & 31) ^ ((2 * wid + (lo >> 5)) & 7)) * 4)) * 4);
        // ONE wave-uniform base (an SGPR pair) per tile; request d adds 8 chunk rows = 4 KiB as a literal to the lane's byte
        // offset and 4 KiB to the LDS address (lean request, see dma16_lean_sc1); member-major gin (xb_internal.h): this
        // workgroup's 64 chunk rows of 128 gate columns are contiguous
        const unsigned char *base = reinterpret_cast<const unsigned char *>(p.gin + (((size_t)tn * members + mb) * N + cbase) * 128);
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            if constexpr (LEAN) dma16_lean_nt(gin_lane, d * 4096, base, lds_g, d * 4096);
            else __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(base + d * 4096 + gin_lane),
                                                  (__attribute__((address_space(3))) void *)(sG + (4 * d + wid) * 1024), 16, 0, 2 /* nt: read once */);
        }
    };
    // (Requesting the tile still earlier -- in the free issue slots of the second-to-last piece, that piece ending on
    // vmcnt(8) -- was measured too: the poll no longer waits for it, but the first-piece landing and the MFMA phase grow by
    // as much: 76.1 vs 74.5 ms per five layers.)
    // accumulators start from the input projection (+ biases): lane (chunk row r, unit u) reads cell u ^ (r & 7)
    // (two lanes of a 16-lane read group share a bank slot: a 2-way conflict on eight reads per step)
    auto acc_from_gin = [&](floatx16 (&acc)[2]) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int r = nt * 32 + (lane & 31);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int u = wid * 8 + 2 * rg + hsel;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(sG + r * (LG_UNITS * 16) + ((u ^ (r & 7)) * 16));
                acc[nt][4 * rg + 0] = v[0]; acc[nt][4 * rg + 1] = v[1];
                acc[nt][4 * rg + 2] = v[2]; acc[nt][4 * rg + 3] = v[3];
            }
        }
    };

    // ---- GDIR (round 5): the tile never touches LDS.  The accumulators START from the input projection, so the lane's 32 values
    // (chunk row 32 nt + (lane & 31), the four gates of units 8 wid + 2 rg + hsel: 16 bytes each) are loaded straight into the
    // accumulator registers -- eight dwordx4 loads per lane, issued at
    // the END of the previous group-step, when the gate math has taken the last value out of them (one accumulator set, one
    // tile in flight).  What that buys: the eight LDS-DMA requests per wave and group-step (14 % of the bytes through the CU's
    // 64 B/clk vector-memory -> LDS path), the eight conflicting ds_read_b128 that fetched the tile back, and 64 KiB of LDS.
    // The tile is streamed from HBM (read once), and a group-step is too short to cover that latency, so the tile of the
    // group-step AFTER the coming one is pulled into L2 first: one LDS-DMA of one dword per lane and wave, each lane touching
    // one of the tile's 256 lines (the four bytes land in a scratch area nobody reads -- a load with a register destination
    // would leave that register unusable until it has landed).  Rows past the slab's last chunk read the slack rows behind
    // the buffer, as before.
    auto gin_tile = [&](int g_i, int s_i) {         // wave-uniform address of the tile of (group g_i of this slot, step s_i)
        const int tn = p.reverse ? T - 1 - s_i : s_i;
        return reinterpret_cast<const unsigned char *>(p.gin + (((size_t)tn * members + mb) * N + p.n0 + (grp + g_i * gh) * LG_BN) * 128);
    };
    auto load_gin_acc = [&](floatx16 (&acc)[2], int g_i, int s_i) {
        const unsigned char *base = gin_tile(g_i, s_i);
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const unsigned voff = (unsigned)((lo & 31) * 512 + wid * 128 + (lo >> 5) * 16);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
#if XB_GIN_NT
                const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(base + voff + nt * 16384 + rg * 32));
#else
                const f32x4 v = *reinterpret_cast<const f32x4 *>(base + voff + nt * 16384 + rg * 32);
#endif
                acc[nt][4 * rg + 0] = v[0]; acc[nt][4 * rg + 1] = v[1];
                acc[nt][4 * rg + 2] = v[2]; acc[nt][4 * rg + 3] = v[3];
            }
    };
    auto prefetch_gin = [&](int g_i, int s_i) {
        if (!XB_GIN_PREFETCH) return;
        const unsigned char *base = gin_tile(g_i, s_i);
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const unsigned voff = (unsigned)((wid * 64 + lo) * 128);
        unsigned t;
        asm volatile("s_add_i32 m0, %[lb], %[lc]\n\tv_add_u32 %[t], 0, %[vo]\n\tglobal_load_lds_dword %[t], %[sb]"
                     : [t] "=&v"(t) : [lb] "s"(lds0 + (unsigned)wid * 256u), [lc] "n"((int)OFF_G), [vo] "v"(voff), [sb] "s"(base) : "memory", "m0");
    };
    // the group-step after (g_i, s_i) in this slot's order
    auto next_gs = [&](int g_i, int s_i, int &g_o, int &s_o) {
        const bool to_second = DUAL && second && g_i == 0;
        g_o = to_second ? 1 : 0;
        s_o = to_second ? s_i : s_i + 1;
    };

    // h_{t-1} of a group's chunks comes piece by piece through LDS (every load sc1).
    // A piece is NPARTS * CPR wave-instructions of 1 KiB; wave w issues NDMA = NPARTS * CPR / 4 of
    // them (q = w, w+4, ..), exactly one per MFMA k-step when NSPLIT == 3, so the next piece's
    // DMA is issued in the shadow of this piece's MFMAs instead of in front of them.
    // Instruction q covers cells [64q, 64q+64) = rows RPI*q .. of CPR cells.  For power-of-two CPR the
    // per-lane part of the source offset is the same for every q of a wave (q = wid mod 4 and
    // RPI*4 = 0 mod CPR), so it is ONE register; everything else is wave-uniform scalar arithmetic.
    constexpr int NDMA = NPARTS * CPR / 4;
    constexpr bool POW2 = (CPR & (CPR - 1)) == 0;
    constexpr int RPI = 64 / (POW2 ? CPR : 1);
    int lane_off_step = 0;      // computed once per (group-)step: as a value kept across the gate math it would be spilled
    // POW2 (every shipped size): lane_off_step is the lane's BYTE offset inside the exchange image, the wave's row block
    // included -- row RPI * wid + lane / CPR, swizzled cell -- and a request is the lean three-instruction form: request (part,
    // j) of piece pc adds the literal part * (part stride) + 4 j RPI rows, the piece's column offset is the load's immediate
    auto issue_dma = [&](const half_t *xprev, int pc, int d) {
        const int lo = lane;
        const int lane_off = lane_off_step;
        const int part = NPARTS == 2 ? (d & 1) : 0;
        const int j = NPARTS == 2 ? (d >> 1) : d;
        const bool slot2 = R3 && pc % 3 == 2;                     // the current group's tile buffer (LDS base lds_g, a runtime value)
        const int lconst = (R3 ? (pc % 3 == 1 ? NPARTS * PIECE_BYTES : 0) : (pc & 1) * NPARTS * PIECE_BYTES) + part * PIECE_BYTES + 4 * j * 1024;
        if constexpr (POW2 && !LEAN) {
            // rounds 1-4: wave-uniform 64-bit base per request + the lane's byte offset (lane_off_step is the same value in both forms)
            constexpr int ROWB = I8 ? F : F * 2;
            const unsigned char *base = reinterpret_cast<const unsigned char *>(xprev) + (size_t)part * (XPART * 2) +
                                        (size_t)(4 * j * RPI) * ROWB + pc * KP * ES;
            dma16_sc1_off(base, lane_off, sPiece + lconst + wid * 1024);
        } else if constexpr (POW2) {
            constexpr int ROWB = I8 ? F : F * 2;                         // bytes per row of one part of the image
            // (the piece's column offset goes into the literal as well: the load's IMMEDIATE offset is added to the LDS address
            //  too -- measured in round 5: with offset:pc * 256 every piece but the first landed 256 pc bytes off)
            const int vconst = part * (int)(XPART * 2) + 4 * j * RPI * ROWB + pc * KP * ES;
            dma16_lean_sc1((unsigned)lane_off, vconst, xprev, slot2 ? lds_g : lds_w, lconst, 0);
        } else {
            const int q = wid + 4 * j;
            const int cell = 64 * q + lo;
            const int row = cell / CPR, pos = cell % CPR;
            dma16_sc1(xprev + part * XPART + (size_t)row * F + pc * KP + (pos ^ (row & SWZ)) * 8, sPiece + lconst + wid * 1024);
        }
    };

    floatx16 acc[2];            // (GDIR: carried from group-step to group-step -- the coming one's input projection is loaded into it)
    if constexpr (GDIR) {
        load_gin_acc(acc, 0, p.s_begin);
        int g1, s1;
        next_gs(0, p.s_begin, g1, s1);
        if (s1 < p.s_end) prefetch_gin(g1, s1);
    } else {
#pragma unroll 1
    for (int gi = 0; gi < NG; ++gi) {
        if (gi == 1 && !second) break;
        serve(gi);
        issue_gin(p.reverse ? T - 1 - p.s_begin : p.s_begin);
    }
    }
    bool early = false;     // DUAL: the first piece (R3: the first two pieces) of the coming group-step was requested during the previous one
    bool gin_prev = false;  // R3: the previous group-step requested a gin tile (eight requests the first closing leaves in flight)
    // DUAL with both groups present: the exchange stores of a group-step are not drained at its end; the next full drain +
    // barrier -- the one that closes the first piece of the other group's step, ~2.7 k cycles later -- covers them, and the
    // arrival goes out behind that (the group's hand-off still has most of the other group's step to complete: its members
    // are looked at ~3.8 k cycles after that point).  One group per slot: its next step waits for this very arrival -- not deferred.
    constexpr bool DEFER = DUAL && XB_LSTM_DEFER_ARRIVE != 0;
    unsigned *arrive_due = nullptr;
    int sig_i = 0, sig_next = XB_SIG(p.sig_flag) ? (int)((long long)T / p.sig_nts) : -1;     // slab being worked on, its end step
    for (int s = p.s_begin; s < p.s_end; ++s) {
        const int t = p.reverse ? T - 1 - s : s;
#pragma unroll 1
        for (int gi = 0; gi < NG; ++gi) {
            if (gi == 1 && !second) break;
            if (DUAL) serve(gi);
            XB_STAMP(0);   // loop overhead / y stores of the previous group-step
            v16i a11[I8 ? 2 : 1], amid[I8 ? 2 : 1], a00[I8 ? 2 : 1];      // NSPLIT == 4: digit-product sums by weight
            // DUAL: the group-step this workgroup serves next, whether it has a recurrent term (s > 0) and whether its
            // group has to be polled first (not in the first step of a launch: the previous launch has retired)
            const int ngi = (DUAL && gi == 0 && second) ? 1 : 0;
            const int ns = (DUAL && gi == 0 && second) ? s : s + 1;
            const bool nxt_h = DUAL && ns < p.s_end && ns > 0 && s > 0;
            const bool nxt_poll = p.persistent && ns > p.s_begin;
            unsigned *ncnt = p.sync + (size_t)(p.grp0 + grp + ngi * gh) * LG_SYNC;
            const unsigned ntarget = (unsigned)members * (p.sync_base + (unsigned)(ns - p.s_begin));
            unsigned seen = 0;
            const half_t *xnext = p.xh + (size_t)(p.grp0 + grp + ngi * gh) * (2 * 2 * LG_BN * F) + (size_t)((ns - 1) & 1) * XPAR;
            const bool was_early = early;
            early = false;
            const bool gin_was = gin_prev;
            gin_prev = false;
            int go = 0;         // EIL: request the coming group-step's first piece during the last piece

            if (s > 0) {
                const half_t *xprev = xg + (size_t)((s - 1) & 1) * XPAR;      // (NSPLIT == 4: same byte offset, XPAR * 2)
                {
                    int lo = lane;
                    if (PARK) asm volatile("" : "+v"(lo));
                    const int lrow = RPI * wid + lo / CPR;
                    lane_off_step = POW2 ? lrow * (I8 ? F : F * 2) + (((lo % CPR) ^ (lrow & SWZ)) * 16) : 0;
                }
                if (!DUAL || !was_early) {
                    if (p.persistent && s > p.s_begin) {
                        // wait until every member of the group has published h_{t-1}
                        if (tid == 0) {
                            const unsigned target = (unsigned)members * (p.sync_base + (unsigned)(s - p.s_begin));
                            const unsigned long long t0 = __builtin_readcyclecounter();
                            int ok = 1;
                            // the counter is polled back to back (one L2 round trip per poll); the error word and the
                            // timeout are looked at every 64th poll only
                            for (unsigned spins = 1; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins) {
                                if ((spins & 63u) == 0 &&
                                    (__hip_atomic_load(p.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                                     __builtin_readcyclecounter() - t0 > LG_SPIN_CYCLES)) {
                                    __hip_atomic_store(p.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                    ok = 0;
                                    break;
                                }
                            }
                            *sFlag = ok;
                        }
                        __syncthreads();
                        if (*sFlag == 0) {
                            // timed out (the error word is set, the host fails the batch): release the stream that waits for
                            // this launch's slabs -- a wait on the flag has no timeout of its own
                            if (tid == 0 && XB_SIG(p.sig_flag))
                                __hip_atomic_fetch_max(p.sig_flag, p.sig_base + (unsigned)p.sig_nts, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                            return;
                        }
                    }
                    XB_STAMP(1);   // gin loads issued + wait for the group
#pragma unroll
                    for (int d = 0; d < NDMA; ++d) issue_dma(xprev, 0, d);
                    if constexpr (R3) {
#pragma unroll
                        for (int d = 0; d < NDMA; ++d) issue_dma(xprev, 1, d);
                        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NDMA) : "memory");   // all but the second piece's requests
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // first piece and the gin tile (this wave's shares)
                    __syncthreads();
                    }
                }
                // (early: the drain wait and barrier that ended the previous group-step covered the first piece and this
                //  group's gin tile, both older than the exchange stores drained there)
                XB_STAMP(2);   // first piece landed
                // the group's second hand-off is complete (blocking poll or, with two groups per workgroup, the look-ahead one):
                // every member has arrived at least once, so every member's XCD bit is in the mask
                if (p.xcd_local && s == p.s_begin + 2 && tid == 0) {
                    const unsigned m = (__hip_atomic_load(cnt + 1 + ((p.slab >> 2) & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >>
                                        (8 * (p.slab & 3))) & 0xffu;
                    sFlag[2 + gi] = (m != 0 && (m & (m - 1)) == 0) ? 1 : 0;      // all members on ONE XCD
                }
                half8 w0 = wh[0];
                if (PARK) {
                    int to = tid;
                    asm volatile("" : "+v"(to));
                    w0 = *reinterpret_cast<const half8 *>(sW0 + to * 16);
                }
                if constexpr (!GDIR) acc_from_gin(acc);
                // lane byte offsets of the B-fragment cells inside a piece part (one register per k-step / q8 cell; the
                // piece buffer, the part and the column tile are immediates)
                unsigned fa[KSP], qa[KSP / 2 > 0 ? KSP / 2 : 1][2];
                {
                    // one cell address each for the fp16 and the q8 fragments; every other k-step's is an XOR away: the
                    // k-step moves bits 1.. of the cell index, the swizzle key XORs into the same bits, (a | b) ^ c splits
                    // (round 5: 2 + 14 VALU per group-step instead of three per address)
                    int lo = lane;
                    asm volatile("" : "+v"(lo));
                    const unsigned r = (unsigned)lo & 31u, hs = (unsigned)lo >> 5;
                    static_assert(!POW2 || I8 || 2 * KSP <= CPR, "the k-step bits stay inside the row (int8 limbs: only k-steps below KSP / 2 are used)");
                    const unsigned fa0 = (r * CPR + (hs ^ (r & SWZ))) * 16;
                    const unsigned qa0 = (r * CPR + ((2 * (1 - hs)) ^ (r & SWZ))) * 16;
#pragma unroll
                    for (int ks = 0; ks < KSP; ++ks)
                        fa[ks] = POW2 ? fa0 ^ (unsigned)(ks << 5) : (r * CPR + ((2 * ks + hs) ^ (r & SWZ))) * 16;
#pragma unroll
                    for (int b = 0; b < KSP / 2; ++b)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
                            qa[b][j] = POW2 ? qa0 ^ (unsigned)((4 * b + j) << 4) : (r * CPR + ((4 * b + 2 * (1 - hs) + j) ^ (r & SWZ))) * 16;
                }
                // B fragments double-buffered by k-step: the 4 reads of k-step ks+1 are issued before the 6 MFMAs
                // of ks (sched_barrier keeps hipcc from sinking the reads back to their first use).  Round 5: the k-step
                // pipeline runs ACROSS the piece boundary (PIPE): a piece that has a successor is closed (DMA drain +
                // barrier) in front of its LAST k-step's MFMAs, whose fragments are in registers by then, and the
                // successor's first fragments are requested behind that barrier -- their LDS latency, which used to sit
                // exposed at the top of every piece with the matrix pipe drained, runs under those MFMAs.
                half8 fh[2][2], fl[2][2];
                v8i fq[2];
                constexpr bool PIPE = !I8 && XB_LSTM_PIPE_PIECES != 0 && KSP % 2 == 0;
#pragma unroll
                for (int pc = 0; pc < NP; ++pc) {
                    const unsigned char *buf = R3 ? (pc % 3 == 2 ? sG : sPiece + (pc % 3) * NPARTS * PIECE_BYTES) : sPiece + (pc & 1) * NPARTS * PIECE_BYTES;
                    const unsigned char *bufn = R3 ? ((pc + 1) % 3 == 2 ? sG : sPiece + ((pc + 1) % 3) * NPARTS * PIECE_BYTES)
                                                   : sPiece + ((pc + 1) & 1) * NPARTS * PIECE_BYTES;      // the successor's buffer
                    // what closes this piece.  R3: everything has landed but the requests issued behind the successor's -- this piece's
                    // (piece 1: its second half), and at the first closing the previous group-step's gin tile; else: everything
                    auto close_wait = [&]() {
#define XB_VM(n) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(n) : "memory")
                        if constexpr (R3) {
                            if (pc == 0) { if (was_early && gin_was) XB_VM(8); else XB_VM(0); }
                            else if (pc == 1) { if (NP > 3 || go) XB_VM(NDMA); else XB_VM(0); }
                            else { if (pc + 2 < NP || go) XB_VM(NDMA); else XB_VM(0); }
                        } else {
                            XB_VM(0);
                        }
#undef XB_VM
                    };
                    auto load_frags_at = [&](const unsigned char *bf, int ks, half8 (&h)[2], half8 (&l)[2]) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const unsigned char *a = bf + nt * (32 * CPR * 16) + fa[ks];
                            h[nt] = *reinterpret_cast<const half8 *>(a);
                            if (NSPLIT == 3) l[nt] = *reinterpret_cast<const half8 *>(a + PIECE_BYTES);
                        }
                    };
                    auto load_frags = [&](int ks, half8 (&h)[2], half8 (&l)[2]) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            // row = 32 nt + (lane & 31); row & SWZ does not depend on nt, so nt is an immediate offset
                            const unsigned char *a = buf + nt * (32 * CPR * 16) + fa[ks];
                            h[nt] = *reinterpret_cast<const half8 *>(a);
                            if (NSPLIT == 3) l[nt] = *reinterpret_cast<const half8 *>(a + PIECE_BYTES);
                        }
                    };
                    // q8 fragment of 32-column block `blk` of the piece: the half OPPOSITE to the W fragment's
                    // (lanes 0-31: the l8 cells 2, 3 of the block; lanes 32-63: the h8 cells 0, 1)
                    auto load_q8 = [&](int blk) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const unsigned char *rb = buf + PIECE_BYTES + nt * (32 * CPR * 16);
                            const v4i x = *reinterpret_cast<const v4i *>(rb + qa[blk][0]);
                            const v4i y = *reinterpret_cast<const v4i *>(rb + qa[blk][1]);
                            fq[nt] = __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
                        }
                    };
                    // NSPLIT == 4: the lane's 16 bytes of each digit for 32-column block b (cell 2 b + hsel of the row)
                    v4i bd1[2][2], bd0[2][2];
                    auto load_dig = [&](int b, v4i (&d1)[2], v4i (&d0)[2]) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            const unsigned char *a = buf + nt * (32 * CPR * 16) + fa[b];
                            d1[nt] = *reinterpret_cast<const v4i *>(a);
                            d0[nt] = *reinterpret_cast<const v4i *>(a + PIECE_BYTES);
                        }
                    };
                    if constexpr (I8) load_dig(0, bd1[0], bd0[0]);
                    else if (!PIPE || pc == 0) load_frags(0, fh[0], fl[0]);      // (PIPE, pc > 0: requested behind the previous piece's closing barrier)
                    // DUAL: has the group of the coming group-step arrived?  One look at its counter (its members had a whole
                    // group-step for it) at the start of the piece whose closing barrier publishes the answer: the last
                    // piece, or (EIL) the one before it.
                    constexpr int PCHK = R3 ? NP - 3 : (EIL ? NP - 2 : NP - 1);
                    if (DUAL && pc == PCHK && nxt_h && nxt_poll && tid == 0) {
                        // (R3: as inline asm -- hipcc would guard the use of a load it can see with vmcnt(0), and this wave would drain
                        //  the requests the counted wait of the piece's closing is there to leave in flight)
                        if constexpr (R3) asm volatile("global_load_dword %0, %1, off sc1" : "=v"(seen) : "v"(ncnt) : "memory");
                        else seen = __hip_atomic_load(ncnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    if (R3 ? pc == NP - 2 : (EIL && pc == NP - 1)) go = __builtin_amdgcn_readfirstlane(nxt_h ? sFlag[1] : 0);
                    if constexpr (I8) {
                        constexpr int KBP = KP / 32;            // 32-column blocks per piece = DMA requests per wave and piece
                        static_assert(NDMA == KBP, "one piece request per block");
                        const v16i zero16 = {};
#pragma unroll
                        for (int b = 0; b < KBP; ++b) {
                            const int kb = pc * KBP + b;
                            if (b + 1 < KBP) load_dig(b + 1, bd1[(b + 1) & 1], bd0[(b + 1) & 1]);
                            __builtin_amdgcn_sched_barrier(0);
                            // column tiles interleaved: the two products into amid[nt] are two issues apart (a dependent MFMA
                            // issued back to back waits for the whole latency of its predecessor)
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                a11[nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wd1[kb], bd1[b & 1][nt], kb == 0 ? zero16 : a11[nt], 0, 0, 0);
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                amid[nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wd1[kb], bd0[b & 1][nt], kb == 0 ? zero16 : amid[nt], 0, 0, 0);
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt)
                                amid[nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wd0[kb], bd1[b & 1][nt], amid[nt], 0, 0, 0);
                            if constexpr (LOLO) {
#pragma unroll
                                for (int nt = 0; nt < 2; ++nt)
                                    a00[nt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(wd0[kb], bd0[b & 1][nt], kb == 0 ? zero16 : a00[nt], 0, 0, 0);
                            }
                            // the piece requests go out in the first half of the piece so that the last has landed at its barrier
                            constexpr int DPB = KBP >= 2 ? 2 : 1;
                            if (b * DPB < NDMA) {
#pragma unroll
                                for (int j = 0; j < DPB; ++j) {
                                    if (pc + 1 < NP) issue_dma(xprev, pc + 1, b * DPB + j);
                                    else if (EIL && go) issue_dma(xnext, 0, b * DPB + j);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    } else {
#pragma unroll
                    for (int ks = 0; ks < KSP; ++ks) {
                        const int kg = pc * KSP + ks;
                        const bool closing = PIPE && ks + 1 == KSP && pc + 1 < NP;     // this piece is closed in front of these MFMAs
                        if (ks + 1 < KSP) load_frags(ks + 1, fh[(ks + 1) & 1], fl[(ks + 1) & 1]);
                        if (NSPLIT == 2 && (ks & 1) == 0) load_q8(ks >> 1);        // used by the odd k-step that follows
                        __builtin_amdgcn_sched_barrier(0);
                        // ONE counted wait per k-step (round 5): everything but the reads just issued has landed, i.e. every
                        // fragment this k-step's MFMAs take (they were requested a k-step ago).  hipcc otherwise puts a counted
                        // lgkmcnt in front of EVERY MFMA -- 130 s_waitcnt per group-step on a wave whose every instruction costs an
                        // issue slot; with this wait in its scoreboard it emits none.  (the builtin needs a literal: spelled out)
                        if (XB_LSTM_ONE_WAIT != 0 || closing) {
                            constexpr int RD = NSPLIT == 3 ? 4 : 2;                    // ds_reads of one load_frags
                            const bool more = ks + 1 < KSP, q8 = NSPLIT == 2 && (ks & 1) == 0;
#define XB_LGKM(n) __builtin_amdgcn_s_waitcnt(0xC07F | ((n) << 8))
                            if (more && q8) XB_LGKM(RD + 4);
                            else if (more) XB_LGKM(RD);
                            else if (q8) XB_LGKM(4);
                            else XB_LGKM(0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        if (closing) {
                            // every fragment of this piece is in registers (lgkmcnt(0) above: nothing was requested in this
                            // k-step).  Close the piece as its end used to: look-ahead flag, DMA drain, barrier, deferred arrival.
                            XB_STAMP(3);
                            if (!R3 && DUAL && pc == PCHK && tid == 0) {
                                sFlag[1] = (nxt_h && (!nxt_poll || seen >= ntarget)) ? 1 : 0;
                                XB_LGKM(0);
                            }
                            close_wait();                                      // next piece landed (this wave's share)
                            if (R3 && pc == PCHK) {                            // (the asm poll load is older than this piece's requests: landed)
                                asm volatile("" : "+v"(seen));
                                if (tid == 0) sFlag[1] = (nxt_h && (!nxt_poll || seen >= ntarget)) ? 1 : 0;
                                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                            }
                            __builtin_amdgcn_s_barrier();
                            __builtin_amdgcn_sched_barrier(0);
                            if (DEFER && pc == 0 && arrive_due) {      // the other group's exchange stores are at L2 in every wave
                                if (tid == 0) __hip_atomic_fetch_add(arrive_due, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                                arrive_due = nullptr;
                            }
                            XB_STAMP(7);
                            load_frags_at(bufn, 0, fh[0], fl[0]);      // fh[0] is free: this k-step is odd (KSP is even)
                            __builtin_amdgcn_sched_barrier(0);
                        }
#undef XB_LGKM
                        // The next piece's requests, two per k-step so that the last one is issued by mid-piece and has landed at
                        // the barrier.  (XB_LSTM_DMA_SPREAD: every request directly behind ONE MFMA -- behind the FP8 ones, which
                        // keep the pipe busy for 64 cycles, where the k-step has them -- instead of in pairs behind the k-step's
                        // MFMAs: measured, not adopted.)
                        auto dma_slot = [&](int j) {
                            if constexpr (R3) {
                                // piece 0: nothing; piece 1: pieces 2 and 3 (one half of the k-steps each); piece pc >= 2: piece pc + 2;
                                // targets beyond this group-step's last piece are the coming group-step's first two (if it is ready)
                                const int i = 2 * ks + j;
                                if (pc == 0 || i >= (pc == 1 ? 2 : 1) * NDMA) return;
                                const int tgt = pc == 1 ? 2 + i / NDMA : pc + 2;
                                if (tgt < NP) issue_dma(xprev, tgt, i % NDMA);
                                else if (go) issue_dma(xnext, tgt - NP, i % NDMA);
                            } else {
                            if (2 * ks + j >= NDMA) return;
                            if (pc + 1 < NP) issue_dma(xprev, pc + 1, 2 * ks + j);
                            else if (EIL && go) issue_dma(xnext, 0, 2 * ks + j);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        };
                        constexpr bool XB_DMA_SPREAD = XB_LSTM_DMA_SPREAD != 0;
                        const bool q_step = NSPLIT == 2 && (ks & 1) == 1;
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt) {
                            if (NSPLIT == 3) {
                                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl[kg], fh[ks & 1][nt], acc[nt], 0, 0, 0);
                                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kg], fl[ks & 1][nt], acc[nt], 0, 0, 0);
                            }
                            acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kg == 0 ? w0 : wh[kg], fh[ks & 1][nt], acc[nt], 0, 0, 0);
                            if (XB_DMA_SPREAD) __builtin_amdgcn_sched_barrier(0);      // (pins the MFMA order: hipcc otherwise pairs
                            if (XB_DMA_SPREAD && !q_step) dma_slot(nt);                //  each column tile's dependent fp16 / FP8 MFMAs)
                        }
                        if (q_step) {
#pragma unroll
                            for (int nt = 0; nt < 2; ++nt) {
                                acc[nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wq[kg >> 1], fq[nt], acc[nt], 0, 0, 0, sca, 0, scb);
                                if (XB_DMA_SPREAD) __builtin_amdgcn_sched_barrier(0);
                                if (XB_DMA_SPREAD) dma_slot(nt);
                            }
                        }
                        if (!XB_DMA_SPREAD) {
                            dma_slot(0);
                            dma_slot(1);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    }
                    if (!PIPE || pc + 1 == NP) {
                    XB_STAMP(3);   // piece compute (ds_read + MFMA + next piece's DMA issue)
                    if (!R3 && DUAL && pc == PCHK && tid == 0) sFlag[1] = (nxt_h && (!nxt_poll || seen >= ntarget)) ? 1 : 0;
                    close_wait();                                      // next piece landed (this wave's share)
                    if constexpr (R3) {                                // (raw barrier: __syncthreads() may drain the requests left in flight)
                        if (pc == PCHK) {                              // the asm poll load is older than this piece's requests: it has landed
                            asm volatile("" : "+v"(seen));
                            if (tid == 0) sFlag[1] = (nxt_h && (!nxt_poll || seen >= ntarget)) ? 1 : 0;
                        }
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                        __builtin_amdgcn_sched_barrier(0);
                    } else {
                    __syncthreads();
                    }
                    if (DEFER && pc == 0 && arrive_due) {      // the other group's exchange stores are at L2 in every wave
                        if (tid == 0) __hip_atomic_fetch_add(arrive_due, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        arrive_due = nullptr;
                    }
                    XB_STAMP(7);   // piece DMA wait + barrier
                    }
                }
                if constexpr (I8) {
                    // pre-activation = gin + row scale * (2^16 S11 + 2^8 (S10 + S01) + S00): every sum is exact, the fp32
                    // combination rounds once per term (|S11| < 2^24)
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) {
                            const f32x4 sc = *reinterpret_cast<const f32x4 *>(sScale + (wid * 8 + 2 * rg + hsel) * 4);
#pragma unroll
                            for (int g = 0; g < 4; ++g) {
                                const int r = 4 * rg + g;
                                const float t = __builtin_fmaf(65536.0f, (float)a11[nt][r],
                                                               LOLO ? __builtin_fmaf(256.0f, (float)amid[nt][r], (float)a00[nt][r])
                                                                    : 256.0f * (float)amid[nt][r]);
                                acc[nt][r] = __builtin_fmaf(sc[g], t, acc[nt][r]);
                            }
                        }
                }
            }

            if (s == 0) {      // no recurrent term in the very first step: the accumulators are the input projection
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (DEFER && arrive_due) {
                    if (tid == 0) __hip_atomic_fetch_add(arrive_due, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    arrive_due = nullptr;
                }
                if constexpr (!GDIR) acc_from_gin(acc);
            }

            // DUAL: request the first piece of the coming group-step now -- it lands behind the gate math, and the drain
            // wait below (everything but the eight youngest operations) covers it
            if (EIL || R3) {
                early = go != 0;    // requested inside the last piece(s); the closing wait and barrier have landed it
            } else if (DUAL && nxt_h && sFlag[1] != 0) {
#pragma unroll
                for (int d = 0; d < NDMA; ++d) issue_dma(xnext, 0, d);
                early = true;
            }
#ifdef XB_LSTM_STAMPS
            if (tid == 0) { sStamp[8] += early ? 1 : 0; sStamp[9] += 1; }
#endif

            // the lane's eight cell states, requested together (round 5): hipcc otherwise reads each right before its use and
            // waits for it there -- eight exposed LDS round trips per group-step.  The fragment registers are dead by now.
            float cprev[2][4];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) cprev[nt][rg] = sC[(wid * 8 + 2 * rg + hsel) * LG_BN + nt * 32 + (lane & 31)];
            if (XB_LSTM_CPREFETCH != 0) __builtin_amdgcn_sched_barrier(0);
            // (XB_LSTM_GIN_SPREAD) the coming step's gin tile of this group: its buffer has been free since acc_from_gin at the top
            // of this group-step; one request behind each cell's gate math
            const bool gin_spread = XB_LSTM_GIN_SPREAD != 0 && !R3 && LEAN && !GDIR && DEFER && second && p.persistent && s + 1 < p.s_end;
            unsigned gs_lane = 0;
            const unsigned char *gs_base = nullptr;
            if (gin_spread) {
                int lo = lane;
                asm volatile("" : "+v"(lo));
                gs_lane = (unsigned)(((2 * wid + (lo >> 5)) * 128 + (((lo & 31) ^ ((2 * wid + (lo >> 5)) & 7)) * 4)) * 4);
                const int tn = p.reverse ? T - 2 - s : s + 1;
                gs_base = reinterpret_cast<const unsigned char *>(p.gin + (((size_t)tn * members + mb) * N + cbase) * 128);
            }
            // gates -> cell -> hidden.  A lane owns units 2*rg + hsel of chunk (lane & 31); v_permlane32_swap pairs them
            // with the other half-wave's units so that each lane packs two ADJACENT units into one dword, written to
            // the [pair][chunk] staging (consecutive lanes -> consecutive dwords: conflict-free).
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                unsigned phi[4], plo[4];
                float hq[4], lq[4];
                unsigned dg1[4], dg0[4];            // NSPLIT == 4: digit bytes of the four units
                constexpr bool PKG = XB_LSTM_PKGATE != 0 && XB_LSTM_GIN_SPREAD == 0;
                float og4[4], th4[4];
                if constexpr (PKG) {
                    // sigmoid(x) = rcp(1 + exp2(-log2e x)), tanh(x) = 1 - 2 rcp(exp2(2 log2e x) + 1): fast_sigmoid / fast_tanh spelled out on pairs
                    // (x * (2 log2e) is (x + x) * log2e bit for bit: a power-of-two factor commutes with the rounding)
                    constexpr float L2E = 1.44269504088896340736f;
                    const f32x2 K_IF = {-L2E, -L2E}, K_GO = {2.0f * L2E, -L2E}, K_T = {2.0f * L2E, 2.0f * L2E}, ONE = {1.0f, 1.0f}, M2 = {-2.0f, -2.0f};
                    auto exp2_2 = [](f32x2 v) { return (f32x2){__builtin_amdgcn_exp2f(v.x), __builtin_amdgcn_exp2f(v.y)}; };
                    auto rcp_2 = [](f32x2 v) { return (f32x2){__builtin_amdgcn_rcpf(v.x), __builtin_amdgcn_rcpf(v.y)}; };
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        const int r0 = 2 * rp, r1 = r0 + 1;
                        const f32x2 if0 = rcp_2(exp2_2((f32x2){acc[nt][4 * r0 + 0], acc[nt][4 * r0 + 1]} * K_IF) + ONE);      // i, f of cell r0
                        const f32x2 go0 = rcp_2(exp2_2((f32x2){acc[nt][4 * r0 + 2], acc[nt][4 * r0 + 3]} * K_GO) + ONE);      // rcp of g's tanh, o
                        const f32x2 if1 = rcp_2(exp2_2((f32x2){acc[nt][4 * r1 + 0], acc[nt][4 * r1 + 1]} * K_IF) + ONE);
                        const f32x2 go1 = rcp_2(exp2_2((f32x2){acc[nt][4 * r1 + 2], acc[nt][4 * r1 + 3]} * K_GO) + ONE);
                        const f32x2 gg = __builtin_elementwise_fma((f32x2){go0.x, go1.x}, M2, ONE);
                        const f32x2 ig = {if0.x, if1.x}, fg = {if0.y, if1.y}, og = {go0.y, go1.y};
                        const f32x2 cn = __builtin_elementwise_fma(ig, gg, fg * (f32x2){cprev[nt][r0], cprev[nt][r1]});
                        sC[(wid * 8 + 2 * r0 + hsel) * LG_BN + nt * 32 + (lane & 31)] = cn.x;
                        sC[(wid * 8 + 2 * r1 + hsel) * LG_BN + nt * 32 + (lane & 31)] = cn.y;
                        // (h = o * tanh(c) itself stays a scalar product below: hipcc contracts it into the f16 split -- fma_mix forms that
                        //  take the residual from the UNROUNDED product -- and a packed multiply in front of that would change the low bits)
                        const f32x2 th = __builtin_elementwise_fma(rcp_2(exp2_2(cn * K_T) + ONE), M2, ONE);
                        og4[r0] = og.x; og4[r1] = og.y;
                        th4[r0] = th.x; th4[r1] = th.y;
                    }
                }
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    float hv;
                    if constexpr (PKG) {
                        hv = og4[rg] * th4[rg];
                    } else {
                    const float ig = fast_sigmoid(acc[nt][4 * rg + 0]);
                    const float fg = fast_sigmoid(acc[nt][4 * rg + 1]);
                    const float gg = fast_tanh(acc[nt][4 * rg + 2]);
                    const float og = fast_sigmoid(acc[nt][4 * rg + 3]);
                    float *cp = sC + (wid * 8 + 2 * rg + hsel) * LG_BN + nt * 32 + (lane & 31);
                    const float cn = __builtin_fmaf(ig, gg, fg * cprev[nt][rg]);     // spelled out: lstm_quad_kernel must round the same way
                    *cp = cn;
                    if constexpr (LEAN && !GDIR && DEFER) {
                        if (gin_spread) {
                            __builtin_amdgcn_sched_barrier(0);
                            dma16_lean_nt(gs_lane, (nt * 4 + rg) * 4096, gs_base, lds_g, (nt * 4 + rg) * 4096);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    hv = og * fast_tanh(cn);
                    }
                    half_t hi, lo;
                    split_f16(hv, hi, lo);
                    phi[rg] = (unsigned)__builtin_bit_cast(unsigned short, hi);
                    plo[rg] = (unsigned)__builtin_bit_cast(unsigned short, lo);
                    hq[rg] = (float)hi * 256.0f;                   // |h| < 1: below the e4m3 maximum by construction
                    lq[rg] = (hv - (float)hi) * 524288.0f;         // 2^19; |residual| <= 2^-11 |hi|
                    if constexpr (I8) {
                        // 16-bit fixed point of h (|q| <= 32512) as two balanced signed digits q = 256 d1 + d0
                        const int q = (int)__builtin_rintf(hv * 32512.0f);
                        const int d0 = ((q + 128) & 255) - 128;
                        dg1[rg] = (unsigned)((q - d0) >> 8) & 255u;
                        dg0[rg] = (unsigned)d0 & 255u;
                    }
                }
                // lanes < 32 hold even units v[rg] = unit 2rg, lanes >= 32 the odd ones v[rg] = unit 2rg+1.
                // v_permlane32_swap(vdst, src) exchanges vdst's upper half-wave with src's lower half-wave, so
                //   swap(v[0], v[2]) -> {r[0], r[1]} = low lanes {unit 0, unit 1}, high lanes {unit 4, unit 5}
                //   swap(v[1], v[3]) -> low lanes {unit 2, unit 3}, high lanes {unit 6, unit 7}
#pragma unroll
                for (int part = 0; part < ((NSPLIT == 3 || (YALT && NSPLIT == 2)) ? 2 : 1); ++part) {
                    unsigned *v = part == 0 ? phi : plo;
                    auto r0 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);
                    auto r1 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
                    const unsigned e0 = r0[0], o0 = r0[1], e1 = r1[0], o1 = r1[1];
                    const int pr = wid * 4 + hsel * 2;                  // first unit pair of this lane
                    // (YALT, NSPLIT 2: the residual pairs are not part of the exchange image: they go to the y staging)
                    unsigned *dst = ((YALT && NSPLIT == 2 && part == 1) ? sTy : sT + part * 16 * ST_LD) + nt * 32 + (lane & 31);
                    dst[(pr + 0) * ST_LD] = e0 | (o0 << 16);
                    dst[(pr + 1) * ST_LD] = e1 | (o1 << 16);
                }
                if constexpr (I8) {
                    // digit image of the 32 units, laid out like the q8 image below: rows 0..7 the d1 bytes of unit quads
                    // 0..7, rows 8..15 their d0 bytes (third staging array)
                    unsigned X = dg1[0] | (dg1[1] << 8) | (dg0[0] << 16) | (dg0[1] << 24);
                    unsigned Y = dg1[2] | (dg1[3] << 8) | (dg0[2] << 16) | (dg0[3] << 24);
                    auto r = __builtin_amdgcn_permlane32_swap(X, Y, false, false);
                    const unsigned r0 = r[0], r1 = r[1];
                    unsigned *dst = sT + 2 * 16 * ST_LD + nt * 32 + (lane & 31);
                    dst[(wid * 2 + hsel) * ST_LD] = __builtin_amdgcn_perm(r1, r0, 0x05010400u);
                    dst[(8 + wid * 2 + hsel) * ST_LD] = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
                }
                if (NSPLIT == 2 || I8 || (YALT && NSPLIT == 3)) {
                    // q8 image of the 32 units: 16 dword rows in the place of the lo staging -- rows 0..7 the h8 bytes of unit
                    // quads 0..7, rows 8..15 their l8 bytes, so the 16-byte cell reads below need no change.
                    // X = {h8(u_a), h8(u_b), l8(u_a), l8(u_b)} of this lane's units (rg 0, 1), Y of (rg 2, 3); after the swap
                    // low lanes hold units (0,2) / (1,3), high lanes (4,6) / (5,7): one byte permute per image interleaves them.
                    unsigned X = fp8_pair<false>(hq[0], hq[1], 0u), Y = fp8_pair<false>(hq[2], hq[3], 0u);
                    X = fp8_pair<true>(lq[0], lq[1], X);
                    Y = fp8_pair<true>(lq[2], lq[3], Y);
                    auto r = __builtin_amdgcn_permlane32_swap(X, Y, false, false);
                    const unsigned r0 = r[0], r1 = r[1];
                    unsigned *dst = ((YALT && NSPLIT == 3) ? sTy : sT + 16 * ST_LD) + nt * 32 + (lane & 31);
                    dst[(wid * 2 + hsel) * ST_LD] = __builtin_amdgcn_perm(r1, r0, 0x05010400u);
                    dst[(8 + wid * 2 + hsel) * ST_LD] = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
                }
            }
            if (DUAL) {
                // raw barrier: __syncthreads() would drain the first piece just requested
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            } else {
                __syncthreads();
            }
            // 64 rows x 64 B per part = 256 cells of 16 B: one per thread per part (cell = 4 unit pairs of one chunk)
            const int orow = tid >> 2, occ = tid & 3;
            uint4 vhi, vlo = make_uint4(0, 0, 0, 0);
            {
                const unsigned *src = sT + (occ * 4) * ST_LD + orow;
                vhi = make_uint4(src[0], src[ST_LD], src[2 * ST_LD], src[3 * ST_LD]);
                if (NSPLIT != 1) {
                    const unsigned *sl = src + 16 * ST_LD;
                    vlo = make_uint4(sl[0], sl[ST_LD], sl[2 * ST_LD], sl[3 * ST_LD]);
                }
            }
            uint4 vylo = vlo;            // second part of the layer output: the exchange image's, or (YALT) the other form
            if (YALT) {
                const unsigned *sl = sTy + (occ * 4) * ST_LD + orow;
                vylo = make_uint4(sl[0], sl[ST_LD], sl[2 * ST_LD], sl[3 * ST_LD]);
            }
            if (s + 1 < T) {
                // publish h_t for the group -- also on the last step of a launch: the next launch (next step, or next time
                // slab) starts from the exchange buffer (rows beyond the slab are scratch rows of the exchange buffer)
                if constexpr (I8) {
                    // one 16-byte cell per thread: occ 0, 1 = the d1 bytes of units 0..15 / 16..31 (part 0), occ 2, 3 = d0 (part 1)
                    const unsigned *sd = sT + 2 * 16 * ST_LD + (occ * 4) * ST_LD + orow;
                    const uint4 vd = make_uint4(sd[0], sd[ST_LD], sd[2 * ST_LD], sd[3 * ST_LD]);
                    unsigned char *xb = reinterpret_cast<unsigned char *>(xg) + (size_t)(s & 1) * (XPAR * 2) +
                                        (size_t)(occ >> 1) * (XPART * 2) + (size_t)orow * F + mb * LG_UNITS + (occ & 1) * 16;
                    store16_sc1(xb, vd);
                } else {
                half_t *xcur = xg + (size_t)(s & 1) * XPAR + (size_t)orow * F + mb * LG_UNITS + occ * 8;
                if (__builtin_amdgcn_readfirstlane(sFlag[2 + gi]) != 0) {      // the group sits on one XCD (proven above)
                    store16_l2(xcur, vhi);
                    if (NSPLIT != 1) store16_l2(xcur + XPART, vlo);
                } else {
                    store16_sc1(xcur, vhi);
                    if (NSPLIT != 1) store16_sc1(xcur + XPART, vlo);
                }
                }
            }
            // layer output for the next layer: plain stores, nobody in this launch reads them
            // (address recomputed from the thread index here: a value kept across the loop gets spilled, and its reload -- a
            // scratch load with a vmcnt(0) behind it -- would wait for the gin DMAs just issued)
            bool y_done = false;
            auto store_y = [&]() {
                int to = tid;
                asm volatile("" : "+v"(to));
                const int n = cbase + (to >> 2);
                if (n <= nlast) {
                    const size_t o = ((size_t)t * N + n) * F + mb * LG_UNITS + (to & 3) * 8;
                    if (XB_SIG(p.sig_flag)) {           // read by another stream's kernel while this launch is still running: write-through
                        store16_sc1(p.y_hi + o, vhi);
                        store16_sc1(p.y_lo + o, vylo);
                    } else {
                        *reinterpret_cast<uint4 *>(p.y_hi + o) = vhi;
                        *reinterpret_cast<uint4 *>(p.y_lo + o) = vylo;
                    }
                }
            };
            XB_STAMP(4);   // pointwise + exchange stores issued
            if (p.persistent && s + 1 < p.s_end) {
                // next step's gin tile (eight LDS-DMAs per wave), then every storing wave drains its exchange stores: all but
                // the eight youngest operations (raw barrier: __syncthreads() would drain the DMAs as well; the LDS reads of
                // the staging are retired here)
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (GDIR) {
                    // one L2 prefetch + eight loads, ALWAYS (the non-deferring drain below counts them): the coming group-step's
                    // tile into the accumulators, the tile after it -- clamped to the launch's last step -- into L2
                    int g1, s1, g2, s2;
                    next_gs(gi, s, g1, s1);
                    next_gs(g1, s1, g2, s2);
                    prefetch_gin(g2, s2 < p.s_end ? s2 : p.s_end - 1);
                    load_gin_acc(acc, g1, s1);              // s1 <= s + 1 < s_end
                    __builtin_amdgcn_sched_barrier(0);
                } else {
                if (!R3 && !gin_spread) issue_gin(p.reverse ? T - 2 - s : s + 1);
                }
                if (DEFER && second) {
                    // no drain here (see arrive_due); the barrier stays: the staging (and, YALT, piece buffer 1) is free for the
                    // other group's step once every wave has read it
                    if constexpr (R3) { store_y(); y_done = true; }    // (before the tile requests: the first closing of the coming
                                                                      //  group-step leaves exactly those eight in flight)
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    arrive_due = cnt;
                    if constexpr (R3) {     // the tile buffer held the alternative y staging: requests only behind the barrier
                        issue_gin(p.reverse ? T - 2 - s : s + 1);
                        gin_prev = true;
                    }
                    XB_STAMP(5);
                    XB_STAMP(6);
                } else if constexpr (R3) {
                    // one group in this slot: its next step waits for this very arrival -- drain everything (nothing is younger than
                    // the exchange stores), arrive, THEN the tile requests (behind the barrier, as above)
                    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    XB_STAMP(5);
                    if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    XB_STAMP(6);
                    issue_gin(p.reverse ? T - 2 - s : s + 1);
                    gin_prev = true;
                } else {
                if constexpr (GDIR && XB_GIN_PREFETCH != 0) asm volatile("s_waitcnt vmcnt(9) lgkmcnt(0)" ::: "memory");      // all but the prefetch and the eight loads
                else if constexpr (GDIR) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                XB_STAMP(5);   // stores drained
                if (tid == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                XB_STAMP(6);   // arrive
                }
            } else {
                if constexpr (GDIR) {       // the launch's last step: the second group's step may still follow
                    int g1, s1;
                    next_gs(gi, s, g1, s1);
                    if (s1 < p.s_end) load_gin_acc(acc, g1, s1);
                }
                __syncthreads();   // sT is rewritten next step (this also lands an early first piece)
            }
            if (!y_done) store_y();
        }
        // time slab complete: every wave's output stores are at the coherence point, then one arrival per workgroup; the last
        // one to arrive publishes the slab to the stream that waits on the flag
        if (XB_SIG(p.sig_flag) && s + 1 == sig_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                const unsigned before = __hip_atomic_fetch_add(p.sig_done + sig_i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (before + 1 == (unsigned)(gh * members))          // workgroups beyond the group slots left at the top
                    __hip_atomic_fetch_max(p.sig_flag, p.sig_base + (unsigned)sig_i + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            ++sig_i;
            sig_next = (int)((long long)T * (sig_i + 1) / p.sig_nts);
        }
    }

#ifdef XB_LSTM_STAMPS
    if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 10; ++i) g_lstm_stamps[i] += sStamp[i];
#endif
#pragma unroll 1
    for (int gi = 0; gi < NG; ++gi) {
        if (gi == 1 && !second) break;
        serve(gi);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int n = cbase + nt * 32 + (lane & 31);
            if (n <= nlast)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg)
                    p.c_state[(size_t)n * F + ubase + 2 * rg + hsel] =
                        sC[(wid * 8 + 2 * rg + hsel) * LG_BN + nt * 32 + (lane & 31)];
        }
    }
}

#ifdef XB_WITH_QUAD          // the software-pipelined experiment: diagnostic library only (make diag), never libxnacall.so
#include "../../tools/diag/xb_lstm_quad.h"
#endif

// dynamic LDS of lstm_kernel<KS, nsplit, dual>
template <int KS>
static size_t lstm_lds_bytes(int nsplit, bool dual)
{
    constexpr int F = KS * 16;
    constexpr int KP = F < 128 ? F : 128;
    const int nparts = nsplit == 1 ? 1 : 2;
    const int ng = dual ? 2 : 1;
    const int es = nsplit >= 4 ? 1 : 2, stp = nsplit >= 4 ? 3 : nparts;     // lstm_kernel: ES, STP
    const size_t gin_bytes = XB_LSTM_GIN_DIRECT != 0 ? 1024 : (size_t)ng * LG_BN * LG_UNITS * 16;      // prefetch scratch, or the tiles
    return (size_t)2 * nparts * LG_BN * KP * es + (size_t)stp * 16 * ST_LD * 4 +
           (size_t)ng * sizeof(float) * LG_UNITS * LG_BN + gin_bytes + 16 + 80 + 256 * 16 + 128 * 4;
}

template <int KS, int NSPLIT, bool DUAL, bool YALT = false>
hipError_t launch_lstm_v(const xb::LstmParams &p, dim3 grid, size_t lds, hipStream_t stream)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_kernel<KS, NSPLIT, DUAL, YALT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL((lstm_kernel<KS, NSPLIT, DUAL, YALT>), grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int KS>
hipError_t launch_lstm_ks(const xb::LstmParams &p, hipStream_t stream)
{
    constexpr int F = KS * 16;
#ifdef XB_WITH_QUAD
    if constexpr (KS == Q_KS) {
        if (p.quad) return launch_lstm_quad(p, stream);
    }
#endif
    const int ngroups = (p.nslab + LG_BN - 1) / LG_BN;
    const bool dual = p.dual != 0;
    const int gh = dual ? (ngroups + 1) / 2 : ngroups;      // workgroup slots (lstm_kernel)
    const int g8 = (gh + 7) & ~7;
    const int members = F / LG_UNITS;
    const size_t lds = lstm_lds_bytes<KS>(p.nsplit, dual);
    const dim3 grid(g8 * members);
    if constexpr (KS % 8 == 0 || KS == 4) {                    // int8-limb pieces: 64 or 128 columns
        if (p.nsplit == 4) return dual ? launch_lstm_v<KS, 4, true>(p, grid, lds, stream) : launch_lstm_v<KS, 4, false>(p, grid, lds, stream);
        if (p.nsplit == 5) return dual ? launch_lstm_v<KS, 5, true>(p, grid, lds, stream) : launch_lstm_v<KS, 5, false>(p, grid, lds, stream);
    } else if (p.nsplit >= 4) {
        return hipErrorInvalidValue;
    }
    // y_alt: the layer output carries the other second part than the exchange image (lstm_kernel YALT)
    if (p.y_alt) {
        if (p.nsplit == 3) return dual ? launch_lstm_v<KS, 3, true, true>(p, grid, lds, stream) : launch_lstm_v<KS, 3, false, true>(p, grid, lds, stream);
        if (p.nsplit == 2) return dual ? launch_lstm_v<KS, 2, true, true>(p, grid, lds, stream) : launch_lstm_v<KS, 2, false, true>(p, grid, lds, stream);
        return hipErrorInvalidValue;
    }
    if (dual) {
        if (p.nsplit == 3) return launch_lstm_v<KS, 3, true>(p, grid, lds, stream);
        if (p.nsplit == 2) return launch_lstm_v<KS, 2, true>(p, grid, lds, stream);
        return launch_lstm_v<KS, 1, true>(p, grid, lds, stream);
    }
    if (p.nsplit == 3) return launch_lstm_v<KS, 3, false>(p, grid, lds, stream);
    if (p.nsplit == 2) return launch_lstm_v<KS, 2, false>(p, grid, lds, stream);
    return launch_lstm_v<KS, 1, false>(p, grid, lds, stream);
}

}  // namespace

namespace xb {

#ifdef XB_LSTM_STAMPS
void lstm_read_stamps(unsigned long long out[10], bool reset)
{
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_lstm_stamps), sizeof(unsigned long long) * 10);
    if (reset) {
        unsigned long long z[10] = {};
        hipMemcpyToSymbol(HIP_SYMBOL(g_lstm_stamps), z, sizeof z);
    }
}
#endif

bool lstm_supported_features(int F)
{
    switch (F) {
    case 32: case 64: case 96: case 128: case 256: case 384: case 512: case 768: return true;
    default: return false;
    }
}
template <int KS, int NSPLIT, bool DUAL>
static int lstm_occupancy_v(size_t lds)
{
    int nb = 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_kernel<KS, NSPLIT, DUAL>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm_kernel<KS, NSPLIT, DUAL>, 256, lds);
    return e == hipSuccess ? nb : 0;
}

template <int KS>
static int lstm_occupancy_ks(int nsplit, bool dual)
{
    const size_t lds = lstm_lds_bytes<KS>(nsplit, dual);
    if constexpr (KS % 8 == 0 || KS == 4) {
        if (nsplit == 4) return dual ? lstm_occupancy_v<KS, 4, true>(lds) : lstm_occupancy_v<KS, 4, false>(lds);
        if (nsplit == 5) return dual ? lstm_occupancy_v<KS, 5, true>(lds) : lstm_occupancy_v<KS, 5, false>(lds);
    } else if (nsplit >= 4) {
        return 0;
    }
    if (dual) {
        if (nsplit == 3) return lstm_occupancy_v<KS, 3, true>(lds);
        if (nsplit == 2) return lstm_occupancy_v<KS, 2, true>(lds);
        return lstm_occupancy_v<KS, 1, true>(lds);
    }
    if (nsplit == 3) return lstm_occupancy_v<KS, 3, false>(lds);
    if (nsplit == 2) return lstm_occupancy_v<KS, 2, false>(lds);
    return lstm_occupancy_v<KS, 1, false>(lds);
}

int lstm_resident_per_cu(int F, int nsplit, int dual)
{
    switch (F / 16) {
    case 2: return lstm_occupancy_ks<2>(nsplit, dual != 0);
    case 4: return lstm_occupancy_ks<4>(nsplit, dual != 0);
    case 6: return lstm_occupancy_ks<6>(nsplit, dual != 0);
    case 8: return lstm_occupancy_ks<8>(nsplit, dual != 0);
    case 16: return lstm_occupancy_ks<16>(nsplit, dual != 0);
    case 24: return lstm_occupancy_ks<24>(nsplit, dual != 0);
    case 32: return lstm_occupancy_ks<32>(nsplit, dual != 0);
    case 48: return lstm_occupancy_ks<48>(nsplit, dual != 0);
    default: return 0;
    }
}

#ifdef XB_WITH_QUAD
int lstm_quad_resident_per_cu() { return lstm_quad_occupancy(); }
#else
int lstm_quad_resident_per_cu() { return 0; }
#endif
int lstm_members(int F) { return F / LG_UNITS; }
int lstm_group_chunks() { return LG_BN; }

hipError_t launch_lstm(const LstmParams &p, hipStream_t stream)
{
    if (!lstm_supported_features(p.F) || p.nslab < 1 || p.s_begin < 0 || p.s_end > p.T || p.s_begin >= p.s_end)
        return hipErrorInvalidValue;
    if (p.n0 < 0 || p.n0 + p.nslab > p.N) return hipErrorInvalidValue;
    if (p.nsplit < 1 || p.nsplit > 5) return hipErrorInvalidValue;
    if (p.nsplit >= 4 && (!p.wq1 || !p.wq0 || !p.wscale)) return hipErrorInvalidValue;
#ifdef XB_WITH_QUAD
    if (p.quad && (p.F != Q_F || p.nsplit != 2 || !p.persistent || !p.dual)) return hipErrorInvalidValue;
#else
    if (p.quad) return hipErrorInvalidValue;
#endif
    if (p.sig_flag && (!p.persistent || p.s_begin != 0 || p.s_end != p.T || !p.sig_done || p.sig_nts < 1 || p.sig_nts > p.T))
        return hipErrorInvalidValue;
    switch (p.F / 16) {
    case 2: return launch_lstm_ks<2>(p, stream);
    case 4: return launch_lstm_ks<4>(p, stream);
    case 6: return launch_lstm_ks<6>(p, stream);
    case 8: return launch_lstm_ks<8>(p, stream);
    case 16: return launch_lstm_ks<16>(p, stream);
    case 24: return launch_lstm_ks<24>(p, stream);
    case 32: return launch_lstm_ks<32>(p, stream);
    case 48: return launch_lstm_ks<48>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace xb
// xb_lstm_quad.h -- the software-pipelined recurrence of the shipped model size (F = 768, q8 exchange image = nsplit 2).
// Included by xb_encoder.hip inside its anonymous namespace, behind lstm_kernel (whose helpers and hand-off protocol it shares).
//
// lstm_kernel<48, 2, DUAL> runs a group-step as MFMA phase -> gate math -> hand-off on ONE wave per SIMD, so the matrix pipe
// idles during the gate math (20 % of a group-step) and the VALU during the MFMA phase.  Here a workgroup serves FOUR groups
// of 32 chunks round-robin (the same 128 chunks per workgroup as two groups of 64) and the gate math of the group-step that
// just left the matrix pipe is issued, slice by slice, BETWEEN the MFMAs of the next group's step:
//
//   slot k:   MFMA phase of group-step k   ||   gates -> cell -> h of group-step k - 1, its exchange stores
//   top of slot k + 1:  arrive for k - 1, its layer-output stores, the request for its next input-projection tile
//
// One accumulator chain per group-step (a single chain of v_mfma_f32_32x32x16 issues back to back, MI355X_MICROARCH.md) in
// exactly the order lstm_kernel adds the same products, the same gate arithmetic statement by statement: the two kernels'
// outputs are bit-identical (tests/test_gpu_lstm_quad.py), so the pairing of calls and the batch size change nothing.
//
// Geometry: a group is 32 chunks x 24 member workgroups (32 hidden units each, 4 waves x 8 units).  h_{t-1} of a group comes in
// three pieces of 256 columns (32 rows x 512 B per part, 2 parts, double buffered: 64 KiB); the buffer a piece uses alternates
// ACROSS slots as well (three pieces per slot), so the next slot's first piece is requested during this slot's last one.
// A group's hand-off (stores reaching L2, the members' arrivals) has the two slots of the other groups to complete.
constexpr int Q_F = 768, Q_KS = 48, Q_BN = 32, Q_KP = 256, Q_NP = 3, Q_NG = 4;
constexpr int Q_PART = Q_BN * Q_KP * 2;         // bytes of one part of one piece (32 rows x 512 B)
constexpr int Q_ST = 36;                        // dword stride of a staging row (32 chunks + 4)
constexpr int Q_TILE = Q_BN * LG_UNITS * 16;    // bytes of a group's input-projection tile (32 chunks x 128 gate columns x 4 B)
constexpr size_t Q_LDS = (size_t)4 * Q_PART + 3 * 16 * Q_ST * 4 + (size_t)Q_NG * (LG_UNITS * Q_BN * 4 + Q_TILE) + 64 + 96;

// LDS-DMA requests from a wave-uniform base + 32-bit lane byte offset to the LDS byte address lds_addr (inline asm: see dma16_sc1);
// nt = read once (input-projection tiles), sc1 = served by L2 (exchange pieces)
__device__ __forceinline__ void qdma16_nt(const void *ubase, unsigned byte_off, unsigned lds_addr)
{
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" ::"v"(byte_off), "s"(ubase), "s"(m0v) : "memory", "m0");
}
__device__ __forceinline__ void qdma16_sc1(const void *ubase, unsigned byte_off, unsigned lds_addr)
{
    const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 sc1" ::"v"(byte_off), "s"(ubase), "s"(m0v) : "memory", "m0");
}

#ifndef XB_Q_ABL       // timing experiments (WRONG results unless 0): bit 0 = no LDS-DMA requests in the MFMA loop, bit 1 = no fragment reads in it, bit 2 = no gate-math slices, bit 3 = no FP8 products
#define XB_Q_ABL 0
#endif
// s_waitcnt vmcnt(0) as the BUILTIN (SIMM16 = vmcnt 0, expcnt 7, lgkmcnt 15), not as inline asm: hipcc's wait-count pass must see
// it.  The W_hh fragments are loaded once in front of the loops and nothing the compiler can see ever waits for them, so
// it guarded their first uses inside the loop with s_waitcnt vmcnt(15) .. vmcnt(0) -- which, with the LDS-DMA requests it
// cannot see in flight, waited for each request the moment it was issued (the MFMA phase ran at 2.3x its length).
__device__ __forceinline__ void q_drain()
{
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0x0F70);
    asm volatile("" ::: "memory");
}

template <bool YALT>
__global__ __launch_bounds__(256) void lstm_quad_kernel(xb::LstmParams p)
{
    constexpr int F = Q_F, KS = Q_KS;
    constexpr size_t XPART = (size_t)Q_BN * F, XPAR = 2 * XPART, XGRP = 2 * XPAR;      // half_t units: part, parity, group
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char *const sPiece = smem_raw;                                            // [2 buffers][2 parts][Q_PART]
    unsigned *const sT = reinterpret_cast<unsigned *>(smem_raw + 4 * Q_PART);          // [2 parts][16 rows][Q_ST]: hi pairs, q8 image
    unsigned *const sTy = sT + 2 * 16 * Q_ST;                                          // [16 rows][Q_ST]: residual pairs (YALT)
    float *const sC0 = reinterpret_cast<float *>(sTy + 16 * Q_ST);                     // [4 groups][32 units][32 chunks] cell state
    unsigned char *const sG0 = reinterpret_cast<unsigned char *>(sC0 + Q_NG * LG_UNITS * Q_BN);   // [4 groups][Q_TILE]
    int *const sFlag = reinterpret_cast<int *>(sG0 + Q_NG * Q_TILE);                   // [0] poll result, [1] look-ahead, [2..5] one-XCD proof
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) void *)smem_raw;   // LDS byte address of the block
    constexpr unsigned OFF_G = 4 * Q_PART + 3 * 16 * Q_ST * 4 + Q_NG * LG_UNITS * Q_BN * 4;           // of sG0 inside it

    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hsel = lane >> 5;
    constexpr int members = F / LG_UNITS;
    const int ngroups = (p.nslab + Q_BN - 1) / Q_BN;
    const int gh = (ngroups + Q_NG - 1) / Q_NG;                 // workgroup slots: slot g serves groups g, g + gh, g + 2 gh, g + 3 gh
    const int g8 = (gh + 7) & ~7;
    const int grp = p.spread ? (int)blockIdx.x / members : (int)blockIdx.x % g8;
    const int mb = p.spread ? (int)blockIdx.x % members : (int)blockIdx.x / g8;
    if (grp >= gh) return;
    int nq = (ngroups - grp + gh - 1) / gh;                     // groups of this slot that exist
    if (nq > Q_NG) nq = Q_NG;
    const int N = p.N, T = p.T;
    const int nlast = p.n0 + p.nslab - 1;
    const int ubase = mb * LG_UNITS + wid * 8;
    const int gbase = 2 * p.grp0 + grp;                         // exchange / counter slot of this workgroup's first group

    // ---- W_hh fragments (as lstm_kernel, NSPLIT == 2): row = gate-interleaved (unit * 4 + gate), lane l: row (l & 31), k half (l >> 5)
    half8 wh[KS];
    v8i wq[KS / 2];
    {
        const size_t row = (size_t)ubase * 4 + (lane & 31);
        const unsigned char *wq8 = reinterpret_cast<const unsigned char *>(p.w_lo);
#pragma unroll
        for (int k = 0; k < KS; ++k) wh[k] = *reinterpret_cast<const half8 *>(p.w_hi + row * F + k * 16 + hsel * 8);
#pragma unroll
        for (int b = 0; b < KS / 2; ++b) wq[b] = *reinterpret_cast<const v8i *>(wq8 + (row * F + b * 32) * 2 + hsel * 32);
    }
    const int sca = 127 - p.w_exp, scb = 127 - 8 - 11;
    q_drain();                  // the fragments have arrived: no wait for them inside the loops (see q_drain)

    // ---- the four groups' descriptors are recomputed from gi where they are needed (wave-uniform scalar arithmetic)
    auto g_cbase = [&](int gi) { return p.n0 + (grp + gi * gh) * Q_BN; };
    auto g_cnt = [&](int gi) { return p.sync + (size_t)(gbase + gi * gh) * 32; };
    auto g_xg = [&](int gi) { return p.xh + (size_t)(gbase + gi * gh) * XGRP; };
    auto g_sC = [&](int gi) { return sC0 + gi * (LG_UNITS * Q_BN); };
    auto g_sG = [&](int gi) { return sG0 + gi * Q_TILE; };
    auto t_of = [&](int s) { return p.reverse ? T - 1 - s : s; };

    // ---- cell state in LDS: lane owns (chunk = l & 31, unit = 8 wid + 2 rg + hsel) of each group
#pragma unroll 1
    for (int gi = 0; gi < nq; ++gi) {
        const int n = g_cbase(gi) + (lane & 31);
        const int ch = n <= nlast ? n : nlast;
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
            g_sC(gi)[(wid * 8 + 2 * rg + hsel) * Q_BN + (lane & 31)] = p.c_state[(size_t)ch * F + ubase + 2 * rg + hsel];
    }
    // ---- same-XCD proof (lstm_kernel): post this workgroup's XCD into every group's mask word
    if (p.persistent && p.xcd_local && tid == 0) {
        const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7u;
        for (int gi = 0; gi < nq; ++gi)
            __hip_atomic_fetch_or(g_cnt(gi) + 1 + ((p.slab >> 2) & 3), 1u << (8 * (p.slab & 3) + xcc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (tid < 16) sFlag[tid] = 0;
#ifdef XB_LSTM_STAMPS
    unsigned long long *sStamp = reinterpret_cast<unsigned long long *>(sFlag + 16);
    if (tid == 0) for (int i = 0; i < 10; ++i) sStamp[i] = 0;
    unsigned long long stamp_prev = __builtin_readcyclecounter();
#endif

    // input-projection tile of (group gi, time step tn): four 1 KiB requests per wave (rows 8 d + 2 wid + (lane >> 5))
    auto issue_gin = [&](int gi, int tn) {
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const unsigned gin_lane = (unsigned)(((2 * wid + (lo >> 5)) * 128 + (((lo & 31) ^ ((2 * wid + (lo >> 5)) & 7)) * 4)) * 4);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned char *base = reinterpret_cast<const unsigned char *>(p.gin + (((size_t)tn * members + mb) * N + g_cbase(gi) + 8 * d) * 128);
            qdma16_nt(base, gin_lane, lds0 + OFF_G + (unsigned)(gi * Q_TILE + (4 * d + wid) * 1024));
        }
    };
    // accumulators start from the input projection (+ biases): lane (chunk row r, unit u) reads cell u ^ (r & 7)
    auto acc_from_gin = [&](int gi, floatx16 &a) {
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int r = lo & 31, hs = lo >> 5;
        const unsigned char *sG = g_sG(gi);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int u = wid * 8 + 2 * rg + hs;
            const f32x4 v = *reinterpret_cast<const f32x4 *>(sG + r * (LG_UNITS * 16) + ((u ^ (r & 7)) * 16));
            a[4 * rg + 0] = v[0]; a[4 * rg + 1] = v[1]; a[4 * rg + 2] = v[2]; a[4 * rg + 3] = v[3];
        }
    };
    // piece pc of the exchange image at xsrc into piece buffer buf: request d of this wave (part d & 1, rows 2 q, 2 q + 1 with
    // q = wid + 4 (d >> 1)); lane i -> row 2 q + (i >> 5), 16-byte cell (i & 31) ^ row
    int lane_off = 0;
    auto set_lane_off = [&]() {
        int lo = lane;
        asm volatile("" : "+v"(lo));
        const int lrow = lo >> 5, pos = lo & 31;
        lane_off = lrow * (F * 2) + ((pos ^ (2 * wid + lrow)) * 16);
    };
    auto issue_dma = [&](const half_t *xsrc, int pc, int d, int buf) {
        const int part = d & 1, j = d >> 1, q = wid + 4 * j;
        const unsigned dst = lds0 + (unsigned)(buf * (2 * Q_PART) + part * Q_PART + q * 1024);
        const half_t *base = xsrc + part * XPART + (size_t)(2 * q) * F + pc * Q_KP;
        qdma16_sc1(base, (unsigned)(lane_off ^ (128 * j)), dst);
    };

    // ---- pipeline state
    floatx16 acc, pacc;                 // the group-step in the matrix pipe / the one whose gate math is pending
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[i] = 0.0f; pacc[i] = 0.0f; }
    bool pend = false;                  // pacc holds the pre-activations of (pgi, ps)
    int pgi = 0, ps = 0;
    bool svc = false;                   // the exchange rows of (vgi, vs) are at L2: arrival, layer output and next tile are due
    int vgi = 0, vs = 0;
    bool early = false;                 // the first piece of the coming slot was requested during the last piece
    int bsel = 0;                       // piece buffer of the coming slot's first piece
    bool sig_due = false;
    int sig_i = 0, sig_next = XB_SIG(p.sig_flag) ? (int)((long long)T / p.sig_nts) : -1;

    // a time slab is complete when its last group-step's layer-output stores are at the coherence point: called behind a
    // full drain + barrier that follows those stores
    auto sig_point = [&]() {
        if (!sig_due) return;
        sig_due = false;
        if (tid == 0) {
            const unsigned before = __hip_atomic_fetch_add(p.sig_done + sig_i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before + 1 == (unsigned)(gh * members))
                __hip_atomic_fetch_max(p.sig_flag, p.sig_base + (unsigned)sig_i + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        ++sig_i;
        sig_next = (int)((long long)T * (sig_i + 1) / p.sig_nts);
    };

    // ---- gate math of (pgi, ps) in slices.  State between the slices of a cell / between the cells and the packing:
    float ig = 0, fg = 0, gg = 0, og = 0, cn = 0, th = 0, hv = 0;
    half_t hi_h = (half_t)0.0f;
    unsigned phi[4], plo[4];
    float hq[4], lq[4];
    // cell rg, first half: gates i, f, g and the new cell state (three slices), second half: o, h and its split (three slices)
    // (the empty asm statements pin a slice's results to the slice: hipcc otherwise sinks the arithmetic to its first use,
    //  i.e. out of the MFMA shadows into the packing)
    auto pinf = [](float &x) { asm volatile("" : "+v"(x)); };
    auto pinu = [](unsigned &x) { asm volatile("" : "+v"(x)); };
    auto cell_slice = [&](int rg, int half, int sl) {
        float *cp = g_sC(pgi) + (wid * 8 + 2 * rg + hsel) * Q_BN + (lane & 31);
        if (half == 0) {
            if (sl == 0) { ig = fast_sigmoid(pacc[4 * rg + 0]); fg = fast_sigmoid(pacc[4 * rg + 1]); pinf(ig); pinf(fg); }
            if (sl == 1) { gg = fast_tanh(pacc[4 * rg + 2]); cn = *cp; pinf(gg); }
            if (sl == 2) { cn = __builtin_fmaf(ig, gg, fg * cn); *cp = cn;      // (the contraction lstm_kernel's statement compiles to)
                          og = fast_sigmoid(pacc[4 * rg + 3]); pinf(og); }
        } else {
            if (sl == 0) { th = fast_tanh(cn); pinf(th); }
            if (sl == 1) {
                hv = og * th;
                half_t lo_h;
                split_f16(hv, hi_h, lo_h);
                phi[rg] = (unsigned)__builtin_bit_cast(unsigned short, hi_h);
                plo[rg] = (unsigned)__builtin_bit_cast(unsigned short, lo_h);
                pinu(phi[rg]); pinu(plo[rg]);
            }
            if (sl == 2) {
                hq[rg] = (float)hi_h * 256.0f;                  // |h| < 1: below the e4m3 maximum by construction
                lq[rg] = (hv - (float)hi_h) * 524288.0f;        // 2^19; |residual| <= 2^-11 |hi|
                pinf(hq[rg]); pinf(lq[rg]);
            }
        }
    };
    // packing into the 16-byte row staging (lstm_kernel's layout with 32 chunks per row): slice 0 the hi pairs, 1 the residual
    // pairs (YALT), 2 the q8 image
    auto pack_slice = [&](int sl) {
        if (sl == 0 || (sl == 1 && YALT)) {
            unsigned *v = sl == 0 ? phi : plo;
            auto r0 = __builtin_amdgcn_permlane32_swap(v[0], v[2], false, false);
            auto r1 = __builtin_amdgcn_permlane32_swap(v[1], v[3], false, false);
            const unsigned e0 = r0[0], o0 = r0[1], e1 = r1[0], o1 = r1[1];
            const int pr = wid * 4 + hsel * 2;
            unsigned *dst = (sl == 0 ? sT : sTy) + (lane & 31);
            dst[(pr + 0) * Q_ST] = e0 | (o0 << 16);
            dst[(pr + 1) * Q_ST] = e1 | (o1 << 16);
        }
        if (sl == 2) {
            unsigned X = fp8_pair<false>(hq[0], hq[1], 0u), Y = fp8_pair<false>(hq[2], hq[3], 0u);
            X = fp8_pair<true>(lq[0], lq[1], X);
            Y = fp8_pair<true>(lq[2], lq[3], Y);
            auto r = __builtin_amdgcn_permlane32_swap(X, Y, false, false);
            const unsigned r0 = r[0], r1 = r[1];
            unsigned *dst = sT + 16 * Q_ST + (lane & 31);
            dst[(wid * 2 + hsel) * Q_ST] = __builtin_amdgcn_perm(r1, r0, 0x05010400u);
            dst[(8 + wid * 2 + hsel) * Q_ST] = __builtin_amdgcn_perm(r1, r0, 0x07030602u);
        }
    };
    // the staged rows as 16-byte cells: 32 rows x 4 cells per part, one cell per thread (threads 0..127 part 0, 128..255 part 1)
    auto staged_cell = [&](const unsigned *base) {
        int to = tid;
        asm volatile("" : "+v"(to));
        const unsigned *src = base + ((to & 3) * 4) * Q_ST + ((to >> 2) & 31);
        return make_uint4(src[0], src[Q_ST], src[2 * Q_ST], src[3 * Q_ST]);
    };
    // publish h of (pgi, ps) for the group (also on the last step of a launch: the next launch starts from the exchange buffer)
    auto exchange_store = [&]() {
        if (ps + 1 >= T) return;
        int to = tid;
        asm volatile("" : "+v"(to));
        const int part = to >> 7, orow = (to >> 2) & 31, occ = to & 3;
        const uint4 v = staged_cell(sT + part * 16 * Q_ST);
        half_t *xcur = g_xg(pgi) + (size_t)(ps & 1) * XPAR + (size_t)part * XPART + (size_t)orow * F + mb * LG_UNITS + occ * 8;
        if (__builtin_amdgcn_readfirstlane(sFlag[2 + pgi]) != 0) store16_l2(xcur, v);      // the group sits on one XCD (proven)
        else store16_sc1(xcur, v);
    };
    // behind the drain + barrier that follow the exchange stores of (vgi, vs): one arrival per workgroup, the group's next
    // input-projection tile, the layer output rows
    auto service2 = [&]() {
        if (p.persistent && vs + 1 < p.s_end) {
            if (tid == 0) __hip_atomic_fetch_add(g_cnt(vgi), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            issue_gin(vgi, t_of(vs + 1));
        }
        int to = tid;
        asm volatile("" : "+v"(to));
        const int part = to >> 7, orow = (to >> 2) & 31, occ = to & 3;
        const int n = g_cbase(vgi) + orow;
        const uint4 v = staged_cell((YALT && part == 1) ? sTy : sT + part * 16 * Q_ST);
        if (n <= nlast) {
            half_t *y = (part == 0 ? p.y_hi : p.y_lo) + ((size_t)t_of(vs) * N + n) * F + mb * LG_UNITS + occ * 8;
            if (XB_SIG(p.sig_flag)) store16_sc1(y, v);          // read by another stream's kernel while this launch is running
            else *reinterpret_cast<uint4 *>(y) = v;
        }
        if (XB_SIG(p.sig_flag) && vgi == nq - 1 && vs + 1 == sig_next) sig_due = true;
        svc = false;
    };
    // the pending gate math on its own (first step, a slot whose group is the pending one, the end of the launch)
    auto p_alone = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // the staging may still be read by service2 of the previous group-step
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int sl = 0; sl < 3; ++sl) cell_slice(rg, h, sl);
#pragma unroll
        for (int sl = 0; sl < 3; ++sl) pack_slice(sl);
        __syncthreads();
        exchange_store();
        q_drain();
        __syncthreads();
        sig_point();
        vgi = pgi; vs = ps;
        service2();
        pend = false;
    };

    // blocking wait until every member has published h_{s-1} of group gi (lstm_kernel's poll); false = timed out
    auto wait_group = [&](int gi, int s) -> bool {
        if (tid == 0) {
            const unsigned target = (unsigned)members * (p.sync_base + (unsigned)(s - p.s_begin));
            unsigned *cnt = g_cnt(gi);
            const unsigned long long t0 = __builtin_readcyclecounter();
            int ok = 1;
            for (unsigned spins = 1; __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++spins) {
                if ((spins & 63u) == 0 &&
                    (__hip_atomic_load(p.error, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 ||
                     __builtin_readcyclecounter() - t0 > LG_SPIN_CYCLES)) {
                    __hip_atomic_store(p.error, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    ok = 0;
                    break;
                }
            }
            sFlag[0] = ok;
        }
        __syncthreads();
        if (sFlag[0] == 0) {
            if (tid == 0 && XB_SIG(p.sig_flag))
                __hip_atomic_fetch_max(p.sig_flag, p.sig_base + (unsigned)p.sig_nts, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        return true;
    };

    // ---- one slot: the MFMA phase of (gi, s), s > 0, with the gate math of (pgi, ps) between its MFMAs when PEND
    auto m_slot = [&](auto pend_c, int gi, int s) __attribute__((always_inline)) -> bool {
        constexpr bool PEND = decltype(pend_c)::value;
        const half_t *xprev = g_xg(gi) + (size_t)((s - 1) & 1) * XPAR;
        set_lane_off();
        // the slot this workgroup runs next: does it have a recurrent term in this launch, does its group have to be looked at
        const int ngi = gi + 1 < nq ? gi + 1 : 0;
        const int ns = gi + 1 < nq ? s : s + 1;
        const bool nxt_h = ns < p.s_end;
        const bool nxt_poll = p.persistent && ns > p.s_begin;
        unsigned *ncnt = g_cnt(ngi);
        const unsigned ntarget = (unsigned)members * (p.sync_base + (unsigned)(ns - p.s_begin));
        const half_t *xnext = g_xg(ngi) + (size_t)((ns - 1) & 1) * XPAR;
        unsigned seen = 0;
        int go = 0;
        const bool was_early = early;
        early = false;
        XB_STAMP(0);   // top of the slot: arrival, layer output, next tile of the serviced group-step
        if (!was_early) {
            if (p.persistent && s > p.s_begin && !wait_group(gi, s)) return false;
            XB_STAMP(1);
#pragma unroll
            for (int d = 0; d < 8; ++d) issue_dma(xprev, 0, d, bsel);
            q_drain();   // first piece and the group's input-projection tile
            __syncthreads();
        }
        XB_STAMP(2);   // first piece landed
        if (p.xcd_local && s == p.s_begin + 2 && tid == 0) {
            const unsigned m = (__hip_atomic_load(g_cnt(gi) + 1 + ((p.slab >> 2) & 3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >>
                                (8 * (p.slab & 3))) & 0xffu;
            sFlag[2 + gi] = (m != 0 && (m & (m - 1)) == 0) ? 1 : 0;
        }
        acc_from_gin(gi, acc);
#pragma unroll
        for (int pc = 0; pc < Q_NP; ++pc) {
            const int buf = (pc + bsel) & 1;
            // lane byte addresses of the B-fragment cells of k-step 0 / block 0 inside this piece's buffer; k-step ks is
            // ^ (32 ks), block b is ^ (64 b) (the XOR stays inside the row's 512 bytes)
            unsigned A0, Q0, Q1;
            {
                int lo = lane;
                asm volatile("" : "+v"(lo));
                const unsigned r = (unsigned)lo & 31u, hs = (unsigned)lo >> 5;
                const unsigned rowb = (unsigned)(sPiece - smem_raw) + (unsigned)buf * (2 * Q_PART) + r * 512u;
                A0 = rowb + ((hs ^ r) * 16u);
                Q0 = rowb + Q_PART + (((2u * (1u - hs) + 0u) ^ r) * 16u);
                Q1 = rowb + Q_PART + (((2u * (1u - hs) + 1u) ^ r) * 16u);
            }
            auto ld_h = [&](int ks) { return *reinterpret_cast<const half8 *>(smem_raw + (A0 ^ (unsigned)(32 * ks))); };
            auto ld_q = [&](int b) {
                const v4i x = *reinterpret_cast<const v4i *>(smem_raw + (Q0 ^ (unsigned)(64 * b)));
                const v4i y = *reinterpret_cast<const v4i *>(smem_raw + (Q1 ^ (unsigned)(64 * b)));
                return __builtin_shufflevector(x, y, 0, 1, 2, 3, 4, 5, 6, 7);
            };
            half8 fh0 = ld_h(0), fh1 = ld_h(1);
            v8i fq = ld_q(0);
            // one look at the coming slot's group counter at the start of the second piece (its members had the other groups'
            // slots for the hand-off); the closing barrier of that piece publishes the answer
            if (pc == 1 && nxt_h && nxt_poll && tid == 0) seen = __hip_atomic_load(ncnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (pc == 2) go = __builtin_amdgcn_readfirstlane(nxt_h ? sFlag[1] : 0);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kg = pc * 16 + 2 * j;
                auto slice = [&](int sl) {
                    if (!PEND || (XB_Q_ABL & 4)) return;
                    if (pc == 0) cell_slice(j >> 1, j & 1, sl);
                    if (pc == 1 && j == 0) pack_slice(sl);
                    if (pc == 2 && j == 0 && sl == 0) exchange_store();
                };
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kg], fh0, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                slice(0);
                if (j < 7 && !(XB_Q_ABL & 2)) fh0 = ld_h(2 * j + 2);
                __builtin_amdgcn_sched_barrier(0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh[kg + 1], fh1, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                slice(1);
                if (j < 7 && !(XB_Q_ABL & 2)) fh1 = ld_h(2 * j + 3);
                __builtin_amdgcn_sched_barrier(0);
                if (!(XB_Q_ABL & 8)) acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wq[kg >> 1], fq, acc, 0, 0, 0, sca, 0, scb);
                __builtin_amdgcn_sched_barrier(0);
                slice(2);
                if (j < 7 && !(XB_Q_ABL & 2)) fq = ld_q(j + 1);
                // the next piece's request d = j of this wave (the last piece: the coming slot's first piece, if its group is there)
                if (!(XB_Q_ABL & 1)) {
                    if (pc + 1 < Q_NP) issue_dma(xprev, pc + 1, j, buf ^ 1);
                    else if (go) issue_dma(xnext, 0, j, buf ^ 1);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            XB_STAMP(3);   // piece compute
            if (pc == 1 && tid == 0) sFlag[1] = (nxt_h && (!nxt_poll || seen >= ntarget)) ? 1 : 0;
            q_drain();   // the next piece (this wave's share), the exchange stores
            __syncthreads();
            if (pc == 0) sig_point();
            XB_STAMP(7);   // piece DMA wait + barrier
        }
        early = go != 0;
        bsel ^= 1;              // three pieces per slot: the coming slot starts in the other buffer
        if (PEND) { svc = true; vgi = pgi; vs = ps; }
#ifdef XB_LSTM_STAMPS
        if (tid == 0) { sStamp[8] += early ? 1 : 0; sStamp[9] += 1; }
#endif
        return true;
    };

#pragma unroll 1
    for (int gi = 0; gi < nq; ++gi) issue_gin(gi, t_of(p.s_begin));

#pragma unroll 1
    for (int s = p.s_begin; s < p.s_end; ++s) {
#pragma unroll 1
        for (int gi = 0; gi < nq; ++gi) {
            if (svc) service2();
            if (pend && (s == 0 || pgi == gi)) p_alone();       // nothing to run it beside
            if (s == 0) {
                q_drain();
                __syncthreads();
                sig_point();
                acc_from_gin(gi, acc);
            } else {
                const bool ok = pend ? m_slot(std::true_type{}, gi, s) : m_slot(std::false_type{}, gi, s);
                if (!ok) return;
            }
            pacc = acc;
            pend = true; pgi = gi; ps = s;
        }
    }
    if (svc) service2();
    if (pend) p_alone();
    q_drain();
    __syncthreads();
    sig_point();

#ifdef XB_LSTM_STAMPS
    if (blockIdx.x == 0 && tid == 0) for (int i = 0; i < 10; ++i) g_lstm_stamps[i] += sStamp[i];
#endif
#pragma unroll 1
    for (int gi = 0; gi < nq; ++gi) {
        const int n = g_cbase(gi) + (lane & 31);
        if (n <= nlast)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg)
                p.c_state[(size_t)n * F + ubase + 2 * rg + hsel] = g_sC(gi)[(wid * 8 + 2 * rg + hsel) * Q_BN + (lane & 31)];
    }
}

hipError_t launch_lstm_quad(const xb::LstmParams &p, hipStream_t stream)
{
    const int ngroups = (p.nslab + Q_BN - 1) / Q_BN;
    const int gh = (ngroups + Q_NG - 1) / Q_NG, g8 = (gh + 7) & ~7;
    const dim3 grid(g8 * (Q_F / LG_UNITS));
    if (p.y_alt) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_quad_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q_LDS);
        hipLaunchKernelGGL((lstm_quad_kernel<true>), grid, dim3(256), Q_LDS, stream, p);
    } else {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_quad_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q_LDS);
        hipLaunchKernelGGL((lstm_quad_kernel<false>), grid, dim3(256), Q_LDS, stream, p);
    }
    return hipGetLastError();
}

int lstm_quad_occupancy()
{
    int nb = 0;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&lstm_quad_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)Q_LDS);
    const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, lstm_quad_kernel<true>, 256, Q_LDS);
    return e == hipSuccess ? nb : 0;
}

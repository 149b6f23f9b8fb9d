// NOTE (round 4): the 16-bit halves of a split operand are IEEE fp16 now (mp_lstm_dev.h pair_of: 24-bit operands, weights split
// as 16 w, MFMA v_mfma_f32_16x16x32_f16), not bf16 as in rounds 1-3 when this file was written; "bf16" in the comments below
// describes the same data path with the other half format.  The hidden-state tag sits in bit 30 of the word (hpair_of).
// K2x -- the persistent fused nn.LSTM layer of mp_lstm_persist.hip (models/rnn.py:27) with SPLIT-bf16 MFMA
// operands: same decomposition, same fp32 state / gates / accumulation, but every fp32
// product a*w inside the two matrix products of a step is evaluated as
//        a_hi*w_hi + a_hi*w_lo + a_lo*w_hi         (a = a_hi + a_lo, w = w_hi + w_lo, each part a bf16 number)
// on v_mfma_f32_16x16x32_bf16 (fp32 accumulate).  One such MFMA covers 8x the K of v_mfma_f32_16x16x4_f32 in
// about the same issue time, so 3 of them replace 8 fp32 MFMAs: the matrix pipe is no longer what bounds a
// step.  The dropped a_lo*w_lo term is <= 2^-18 relative per product, hi+lo itself carries 16 significand
// bits; measured end to end (tools/accuracy.py, tests/test_gpu_parity.py) the outputs stay <= 5e-7 from the same
// arithmetic in float64 -- 200x inside the 1e-4 parity bound, the level of fp32's own rounding noise.
// A single bf16 term (1e-4 .. 5e-4) does NOT pass; that is why the split exists.
//
// Data formats (all 4 bytes per value, so buffers, LDS images and register counts keep their sizes):
//   * activations that feed an LSTM layer or a linear layer (linear1's output X1, both layers' outputs) are "pairs":
//     (bf16 hi << 16) | bf16 lo  (mp_lstm_dev.h pair_of); the exchanged hidden state is a pair with a 7-bit lo
//     mantissa and an epoch tag in bit 0 (below);
//   * W_hh / W_ih are packed as separate hi and lo bf16x8 B-fragments (mp_pack_*_x3 below);
//   * a lane's A fragment (8 consecutive k of one sequence row) is built from 8 pair words with 8 v_perm_b32.
// k mapping: wave kq owns K quarter kq; chunk c = 32 k of it; lane (row r16, k-block q) holds
//   k = kq*KQ + c*32 + q*8 + e,  e = 0..7.
// Hidden-state exchange.  With the matrix pipe out of the way a step is bound by how fast h_t crosses the 8 (16)
// workgroups of a cluster, and the tagged 8-byte granules of the fp32 kernel become the bottleneck: every CU
// issued 128 wave-wide loads per step (64 KB through a 64 B/clk L1 path; every value fetched twice, half of the
// bytes tags) -- measured 1500-2500 cycles of load issue on the critical path of a 6400-cycle step.  Here the
// exchanged word is its own flag at no extra bytes:
//   * the lo part of an exchanged pair is rounded to 7 mantissa bits (hpair_of: hi + lo then carries 15 significand
//     bits, measured <= 1e-6 end to end) and bit 0 holds an epoch tag: ((step / 2) + 1) & 1.  Blocks are parity-
//     double-buffered, so a word of one buffer is rewritten every second step and its tag alternates; a producer can
//     be at most one step ahead of a consumer (it needs this workgroup's h to go further), writing the OTHER buffer:
//     "tag == tag_of_step(step - 1)" is the complete test.  The area is zeroed by the kernel that runs before the
//     layer (rearm_exchange), the first write of every word carries tag 1.
//   * a producer wave stores its (row, unit) words into its slice's [16 rows][U units] block and goes on -- no
//     acknowledgement wait, no flag, no barrier on the publishing side;
//   * a consumer wave fetches the 2 KB (1 KB) block of the producer slice(s) it is responsible for with two (four)
//     16-byte loads per lane that bypass the L1, optimistically (peers run in the same phase: issued at the top of a
//     step, one chunk of MFMAs after its own stores, the block is complete 99 % of the time; requested 200 cycles
//     earlier it is stale 95 % of the time), checks the tags after the MFMAs that hide the flight, refetches while any
//     word is stale (bounded), writes the block -- tags cleared -- into an LDS tile [16 rows][256 units], and after one
//     barrier every wave reads its MFMA A fragments from LDS: each value crosses the L1 path once per CU (16 KB/step).
//   One round trip per step (store -> L2 -> load) instead of the two of a data + flag protocol (store, acknowledge,
//   flag store, flag poll, block fetch), and nothing in the loop waits for a store: the previous version's
//   `s_waitcnt vmcnt(0)` before the flag also waited for the x_{t+1} prefetch (HBM/MALL misses) on every step.
//   Same two transports as the fp32 kernel, chosen per producer from the real XCC ids: L (stores that stay in this
//   XCD's L2) / R (write-through stores), both read with L1-bypassing loads.  Waits are bounded.
// H = 256 only (a K quarter must hold a 32-wide chunk); the H = 64 foot-contact block keeps the fp32 kernel.
#include "mp_lstm_dev.h"

namespace {

template <int NSLICE, int KIN>
struct CfgX {
    static constexpr int H = 256;
    static constexpr int U = H / NSLICE;              // hidden units per workgroup (32 | 16)
    static constexpr int TW = U / 16;                 // unit blocks = tile groups (2 | 1)
    static constexpr int NWV = 4 * TW;                // waves: (K quarter kq, unit block tw)
    static constexpr int KQ = KIN / 4;                // x: K range of one wave (64 | 128)
    static constexpr int NXC = KQ / 32;               // x: chunks per wave (2 | 4)
    static constexpr int NHC = 2;                     // h: chunks per wave (K quarter 64)
    static constexpr int CH_U4 = 8 * 64;              // uint4 per wave per chunk: [tile*2 + part][lane]
    static constexpr int RED_F4 = NWV * 4 * 64;       // one reduction buffer: [finishing wave][source kq][lane]
    // x chunks [0, XRC) of a wave's W_ih live in registers and are multiplied first (nothing to wait for at the top
    // of a step); chunks [XRC, NXC) stream from LDS and are fetched while the register chunks run.
    // K_in = 256: 1 + 1 (212 + 32 VGPRs), K_in = 512: 2 + 2.
    static constexpr int XLC = KIN > H ? 2 : 1;       // x chunks served from LDS
    static constexpr int XRC = NXC - XLC;             // x chunks served from registers
    // the LDS that K_in = 256 no longer needs for weights holds a second reduction buffer: steps alternate
    // buffers, which removes the write-after-read barrier of the K reduction
    static constexpr int RED_BUFS = (KIN > H || NSLICE == 16) ? 1 : 2;     // (4-wave packing: 80 KB per workgroup)
    static constexpr bool BIG = KIN > H;
    static constexpr int WG_PER_CU = NSLICE == 16 ? 2 : 1;
    // hidden-state exchange (see the kernel header): a wave fetches the blocks of PPW producer slices
    static constexpr int PPW = NSLICE / NWV;          // producers per consumer wave (1 | 4)
    static constexpr int LPB = 64 / PPW;              // lanes that share one producer block (64 | 16)
    static constexpr int WPL = 16 * U / LPB;          // 32-bit words per lane (8 | 16)
    static constexpr int PARTS = U / WPL;             // lanes per row of a block (4 | 1)
    // x sharing (8-wave packing, K_in = 256): the two waves (kq, tw = 0 | 1) that own the two unit blocks of a K quarter
    // need the same x words.  Each of them loads ONE of the quarter's two chunks (two steps ahead), passes it to its
    // partner through a 16 KB LDS tile and reads the other chunk from there: a CU pulls 16 KB of x per step through its
    // L1 instead of 32 KB (the loads cost 8 % of such a layer; K_in = 512 has no LDS left for this -- mp_lstm_x3w.hip).
    // The 16 KB come from an unpadded, XOR-swizzled h tile (the budget closes with 0 bytes to spare).
#ifndef X3_XSHARE
#define X3_XSHARE 1
#endif
    static constexpr bool XSHARE = X3_XSHARE && TW == 2 && KIN == H;
    static constexpr bool SWZ = XSHARE;
    static constexpr int HPITCH = SWZ ? H : H + 4; // LDS row pitch of the staged h tile (conflict-free b128 reads: padding | swizzle)
    static constexpr int HT_BYTES = 16 * HPITCH * 4;
    static constexpr int XT_BYTES = XSHARE ? 4 * NXC * 64 * 32 : 0;   // [kq][chunk][lane][8 words]
    // K_in = 512 has no LDS left: the h tile shares the (single) reduction buffer -- the barriers of the step
    // separate the two uses (h tile: written .. barrier .. read | barrier | partial sums: written .. barrier .. read)
    static constexpr bool HT_ALIAS = KIN > H;
    static constexpr int LDS_BYTES = RED_BUFS * RED_F4 * 16 + NWV * XLC * CH_U4 * 16 + (HT_ALIAS ? 0 : HT_BYTES) + XT_BYTES;
    static_assert(LDS_BYTES <= (NSLICE == 16 ? 80 : 160) * 1024, "LDS budget (two workgroups per CU for the 4-wave packing)");
};

constexpr int x3_threads(int nslice) { return 64 * 4 * (256 / nslice / 16); }
constexpr int x3_wg_per_cu(int nslice) { return nslice == 16 ? 2 : 1; }

template <int NSLICE, int KIN, bool PROF>
MP_KERNEL __launch_bounds__(x3_threads(NSLICE), x3_wg_per_cu(NSLICE)) void mp_lstm_x3(LstmPersistArgs a) {
    // test hook (mp_debug_drop_workgroup): a workgroup that never shows up.  Only in the PROF instantiation, which the launcher
    // picks when the hook is armed -- the product kernels carry no test code (round 4)
    if (PROF && a.debug_drop && (int)blockIdx.x == a.debug_drop - 1) return;
    using C = CfgX<NSLICE, KIN>;
    constexpr int H = 256, U = C::U, NWV = C::NWV, KQ = C::KQ, NXC = C::NXC, NHC = C::NHC, XLC = C::XLC, XRC = C::XRC;
    constexpr int CH_U4 = C::CH_U4, PPW = C::PPW, LPB = C::LPB, WPL = C::WPL, PARTS = C::PARTS, HPITCH = C::HPITCH;
    constexpr int NTHREADS = 64 * NWV;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);                        // [finishing wave][source kq][lane]
    u32x4* wxl = reinterpret_cast<u32x4*>(smem) + C::RED_BUFS * C::RED_F4;   // [wave][LDS chunk][tile*2+part][lane]
    unsigned* hT = C::HT_ALIAS ? reinterpret_cast<unsigned*>(smem)          // staged h_{t-1} tile [16 rows][HPITCH] pair words
                               : reinterpret_cast<unsigned*>(wxl + (size_t)NWV * XLC * CH_U4);

    // block -> (cluster = (direction, slab), slice): see mp_lstm_persist.hip (slices of a cluster share an XCD)
    const int ncl = a.ndir * a.nslab;
    const int cl = ((int)(blockIdx.x >> 3) / NSLICE) * 8 + (int)(blockIdx.x & 7);
    const int slice = (int)(blockIdx.x >> 3) % NSLICE;
    if (cl >= ncl) return;
    const int dir = cl / a.nslab, slab = cl % a.nslab;
    const LstmDir d = a.d[dir];
    // the next layer's launch uses another exchange area: its cluster `cl` is re-armed here (each slice its share), which
    // saves a separate kernel -- and a kernel boundary on the critical path -- between the two layers
    if (a.hx_next != nullptr) rearm_exchange(a.hx_next + (size_t)cl * ((size_t)4 * 16 * H + 16), 1, slice, NSLICE, threadIdx.x, NTHREADS);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kq = wave & 3, tw = wave >> 2;
    const int q = lane >> 4, r16 = lane & 15;
    const int B = a.B, T = a.T;
    const int brow0 = (a.slab0 + slab) * 16;

    // ---- W_ih slice: chunks [0, XRC) -> registers, [XRC, NXC) -> LDS; W_hh slice -> registers
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(d.wihpack) + (size_t)slice * NWV * NXC * CH_U4;
        for (int w = 0; w < NWV; ++w)
            for (int i = threadIdx.x; i < XLC * CH_U4; i += NTHREADS)
                wxl[(size_t)w * XLC * CH_U4 + i] = src[((size_t)w * NXC + XRC) * CH_U4 + i];
    }
    u32x4 wxr[XRC][8];
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(d.wihpack) + (size_t)(slice * NWV + wave) * NXC * CH_U4 + lane;
#pragma unroll
        for (int c = 0; c < XRC; ++c)
#pragma unroll
            for (int tp = 0; tp < 8; ++tp) wxr[c][tp] = src[(size_t)(c * 8 + tp) * 64];
    }
    u32x4 whh[NHC][8];
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(d.wpack) + (size_t)(slice * NWV + wave) * NHC * CH_U4 + lane;
#pragma unroll
        for (int c = 0; c < NHC; ++c)
#pragma unroll
            for (int tp = 0; tp < 8; ++tp) whh[c][tp] = src[(size_t)(c * 8 + tp) * 64];
    }

    // ---- the (sequence, unit) pair this lane finishes: accumulator reg kq of tile column r16 of unit block tw
    const int jown = slice * U + tw * 16 + r16;
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(d.bias + 4 * jown);
    const int bown = brow0 + q * 4 + kq;
    const bool inb = bown < B;
    const int blen = inb ? a.lengths[bown] : 0;
    float cst = (inb && !a.zero_state) ? d.cbuf[(size_t)bown * H + jown] : 0.f;
    float hst = (inb && !a.zero_state) ? d.hbuf[(size_t)bown * H + jown] : 0.f;

    // ---- A-operand row of this lane (row r16 of the slab)
    const int arow = brow0 + r16;
    const bool arow_in = arow < B;
    const int alen = arow_in ? a.lengths[arow] : 0;
    const unsigned* xbase = reinterpret_cast<const unsigned*>(d.xin) + (size_t)(arow_in ? arow : 0) * KIN + kq * KQ + q * 8;
    const size_t xtstride = (size_t)B * KIN;

    // A operand of the recurrent part for step 0 from the initial fp32 state: pairs of h0[row][kq*64 + c*32 + q*8 + e]
    u32x4 hw[NHC][2];
    {
        const float* p = d.hin + (size_t)(arow_in ? arow : 0) * H + kq * 64 + q * 8;
#pragma unroll
        for (int c = 0; c < NHC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                hw[c][e >> 2][e & 3] = (arow_in && !a.zero_state) ? pair_of(p[c * 32 + e]) : 0u;
    }

    // exchange area of this cluster (32-bit words): hx[cluster] = { dataL[2 parities][16*H], dataR[2][16*H],
    //   xcc table (64-bit granules at word 2*4*16*H) }
    constexpr size_t SLABW = (size_t)4 * 16 * H + 16;                       // in 64-bit words (host allocation unit)
    unsigned* hxw = reinterpret_cast<unsigned*>(a.hx + (size_t)cl * SLABW);
    unsigned* dataL = hxw;
    unsigned* dataR = hxw + 2 * 16 * H;
    u64* xtab = a.hx + (size_t)cl * SLABW + (size_t)4 * 16 * H;
    unsigned spin_budget = a.max_spin;
    const unsigned my_xcc = xcc_id();
    unsigned long long same = ~0ull;
    bool all_local = true;
    {
        if (threadIdx.x == 0) granule_store(xtab + slice, XCC_TAG, __uint_as_float(my_xcc));
        unsigned peer = my_xcc;
        if (lane < NSLICE) {
            unsigned spins = 0; u64 wt0 = 0;
            while (true) {
                const u64 g = granule_load(xtab + lane);
                if ((unsigned)(g >> 32) == XCC_TAG) { peer = (unsigned)g; break; }
                if (wait_over(spins, spin_budget, wt0, a.max_ticks)) { mp_set_error(a.err, 1000000); peer = ~0u; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        same = __ballot(peer == my_xcc);
        all_local = (same & ((1ull << NSLICE) - 1)) == ((1ull << NSLICE) - 1);
        if (__ballot(peer == ~0u)) { spin_budget = 0; poison_cells(cst); }
        if (a.force_remote) { all_local = false; same = 0; }      // test hook: exercise the any-placement transport
    }
    // consumer role of this lane: producer slice PPW*wave + lane/LPB; inside its [16 rows][U units] block this lane
    // fetches WPL consecutive words of row li/PARTS and puts them at the same (row, unit) of the LDS tile
    const int cprod = PPW * wave + lane / LPB;
    const int cli = lane % LPB;
    const int crow = cli / PARTS, cpart = cli % PARTS;
    const bool cloc = (same >> cprod) & 1;
    const unsigned* csrc = (cloc ? dataL : dataR) + (size_t)cprod * 16 * U + crow * U + cpart * WPL;
    // the blocks are fetched with buffer loads (16 bytes per lane, sc1 = past the L1) that the compiler can see: with
    // inline-asm loads its s_waitcnt arithmetic does not know about them and every wait for an OLDER load (the x words of
    // the chunk multiplied under the fetch) silently waits for the fetch as well -- no overlap at all
    const __amdgpu_buffer_rsrc_t hxrsrc = __builtin_amdgcn_make_buffer_rsrc(hxw, 0, (int)(SLABW * 8), 0x27000);
    const int csrc_byte = (int)((csrc - hxw) * 4);
    // (16-byte unit u of row r lives at unit u ^ (r & 7) when the tile is unpadded: XSHARE)
    auto hT_off = [&](int row, int unit) { return row * HPITCH + ((C::SWZ ? unit ^ (row & 7) : unit) << 2); };
    int cdst_off[WPL / 4];
#pragma unroll
    for (int i = 0; i < WPL / 4; ++i) cdst_off[i] = hT_off(crow, (cprod * U + cpart * WPL) / 4 + i);
    // producer role: this lane's (row q*4+kq, unit jown) word of the slice's block
    unsigned* pdstL = dataL + (size_t)slice * 16 * U + (q * 4 + kq) * U + (jown - slice * U);
    unsigned* pdstR = dataR + (size_t)slice * 16 * U + (q * 4 + kq) * U + (jown - slice * U);
    // reader role: A fragments of the recurrent product from the LDS tile
    int hrd_off[NHC][2];
#pragma unroll
    for (int c = 0; c < NHC; ++c)
#pragma unroll
        for (int j = 0; j < 2; ++j) hrd_off[c][j] = hT_off(r16, kq * 16 + c * 8 + q * 2 + j);

    // ---- x_0: pair words of this lane's row, chunk c: k = kq*KQ + c*32 + q*8 + e
    u32x4 xw[NXC][2];
    constexpr bool SPLIT_X = C::BIG;                      // x words of the LDS-resident chunks prefetched at a different point
    constexpr int XC_PRE = SPLIT_X ? XRC : NXC;
    auto load_x = [&](int step, int c0, int c1) {
        const bool on = step < alen;
        const int t = on ? (d.reverse ? alen - 1 - step : step) : 0;
        const unsigned* p = xbase + (size_t)t * xtstride;
#pragma unroll
        for (int c = 0; c < NXC; ++c)
            if (c >= c0 && c < c1) {
                // (unconditional loads from a clamped, always valid row; rows past their length are zeroed where the words are
                //  USED (xsel): a branch around the loads makes the compiler merge wait counts at the join, and a select here
                //  is a use that the scheduler sinks to the next scheduling barrier -- behind the stores of the exchange,
                //  where waiting for these loads means waiting for the stores' acknowledgements too)
                xw[c][0] = *reinterpret_cast<const u32x4*>(p + c * 32);
                xw[c][1] = *reinterpret_cast<const u32x4*>(p + c * 32 + 4);
            }
    };
    auto xsel = [&](u32x4 v, int step) { return step < alen ? v : u32x4{0u, 0u, 0u, 0u}; };
    load_x(0, 0, NXC);
    // K_in = 256: the x words of chunk 0 (multiplied at the END of the previous step, right behind the stores of the
    // exchange) are prefetched TWO steps ahead through a second register pair, so that those MFMAs never wait for a load
    u32x4 xq[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    auto load_xq = [&](int step) {
        const bool on = step < alen;
        const int t = on ? (d.reverse ? alen - 1 - step : step) : 0;
        const unsigned* p = xbase + (size_t)t * xtstride;
        xq[0] = *reinterpret_cast<const u32x4*>(p);
        xq[1] = *reinterpret_cast<const u32x4*>(p + 4);
    };
    // x sharing: xw[0] / xw[1] hold this wave's OWN chunk (tw) and the partner's chunk of the current x; xqa / xqb are the
    // own chunk two steps ahead (two sets that alternate: the step loop is unrolled by two, no copies on the back edge)
    unsigned* xT = hT + 16 * HPITCH;                          // [kq][chunk][lane][8 words]
    const unsigned* xown = xbase + tw * 32;
    const bool tw0 = tw == 0;
    u32x4 xqa[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}}, xqb[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    auto load_own = [&](u32x4 (&dst)[2], int step) {
        const bool on = step < alen;
        const int t = on ? (d.reverse ? alen - 1 - step : step) : 0;
        const unsigned* p = xown + (size_t)t * xtstride;
        dst[0] = *reinterpret_cast<const u32x4*>(p);
        dst[1] = *reinterpret_cast<const u32x4*>(p + 4);
    };
    // chunk c of the current x from (own, partner's): a wave-uniform select
    auto xpick = [&](int c, int k) { return (c == 0) == tw0 ? xw[0][k] : xw[1][k]; };
    if constexpr (C::XSHARE) {
        // (x_0 was loaded in full by every wave: xw[c] = chunk c; re-label as (own, partner's))
        const u32x4 o0 = tw0 ? xw[0][0] : xw[1][0], o1 = tw0 ? xw[0][1] : xw[1][1];
        const u32x4 p0 = tw0 ? xw[1][0] : xw[0][0], p1 = tw0 ? xw[1][1] : xw[0][1];
        xw[0][0] = o0; xw[0][1] = o1; xw[1][0] = p0; xw[1][1] = p1;
        load_own(xqa, 1);
    }
    __syncthreads();                                          // W_ih LDS image complete

    long long pt[6] = {0, 0, 0, 0, 0, 0};
    const bool prof = PROF && a.prof != nullptr && threadIdx.x == 0;
#define PROF_T(i) do { if (PROF && prof) pt[i] -= (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define PROF_E(i) do { if (PROF && prof) pt[i] += (long long)__builtin_amdgcn_s_memtime(); } while (0)

    const u32x4* wxw = wxl + (size_t)wave * XLC * CH_U4 + lane;

    f32x4 acc[4];
    // one chunk of a matrix product: 4 MFMAs per gate tile (hi*hi, hi*lo, lo*hi, lo*lo -- round 6: rounds 1-5 dropped the last
    // one, 2^-22 relative per product = 4 x fp32's rounding unit, and sat at 1.8-3.3 x fp32's noise), tiles interleaved so that
    // consecutive MFMAs never depend on each other
    auto chunk_mma = [&](u32x4 x0, u32x4 x1, const u32x4 (&w)[8]) {
        u32x4 ahi, alo;
        split_pairs(x0, x1, ahi, alo);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(ahi, w[2 * t], acc[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(ahi, w[2 * t + 1], acc[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(alo, w[2 * t], acc[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(alo, w[2 * t + 1], acc[t]);      // lo*lo (round 6): every product exact
    };
    // LDS-resident chunks: 8 fragments per chunk = (hi, lo) of the 4 gate tiles.  K_in = 256 fetches a whole chunk
    // at the top of a step (32 registers, under the register chunks).  K_in = 512 cannot hold that through the
    // register chunks: only the hi fragments are fetched there, the lo fragments follow when the x words of the
    // register chunks are dead, and each half is refilled right after the MFMAs that read it were issued.
    u32x4 wl[8];
    auto lds_fetch = [&](int c, int part) {                    // part: 0 = hi fragments, 1 = lo fragments, 2 = both
#pragma unroll
        for (int tp = 0; tp < 8; ++tp)
            if (part == 2 || (tp & 1) == part) wl[tp] = wxw[(size_t)(c * 8 + tp) * 64];
    };
    auto lds_chunk_mma = [&](int c, u32x4 x0, u32x4 x1) {      // multiplies chunk c (already fetched), fetches the next
        if constexpr (C::BIG) {
            u32x4 ahi, alo;
            split_pairs(x0, x1, ahi, alo);
            if (c == 0) lds_fetch(0, 1);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(ahi, wl[2 * t], acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(alo, wl[2 * t], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < XLC) lds_fetch(c + 1, 0);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(ahi, wl[2 * t + 1], acc[t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = mfma_bf16(alo, wl[2 * t + 1], acc[t]);     // lo*lo (round 6)
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < XLC) lds_fetch(c + 1, 1);
        } else {
            chunk_mma(x0, x1, wl);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < XLC) lds_fetch(c + 1, 2);
        }
    };
    // The first register chunk of step t+1 is multiplied at the END of step t, between issuing the stores of h_t
    // and waiting for their acknowledgement (the loop is rotated): the matrix pipe works through the store latency
    // instead of after it.  Step 0's is done here.
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef X3_SKIP_PROJ
    if constexpr (C::XSHARE) chunk_mma(xsel(xpick(0, 0), 0), xsel(xpick(0, 1), 0), wxr[0]);
    else chunk_mma(xsel(xw[0][0], 0), xsel(xw[0][1], 0), wxr[0]);
#endif
    if constexpr (!C::BIG && !C::XSHARE) load_xq(1);

    // the blocks of the h written at step `pstep` (parity pstep & 1)
    u32x4 blk[WPL / 4];
    auto fetch_blocks = [&](int pstep) {
        const int poff_b = (pstep & 1) * 16 * H * 4;
#pragma unroll
        for (int i = 0; i < WPL / 4; ++i)
            blk[i] = __builtin_amdgcn_raw_buffer_load_b128(hxrsrc, csrc_byte + poff_b + 16 * i, 0, 16 /* sc1 */);
    };

    // one time step.  (xqc, xqn): x sharing only -- xqc holds the own chunk of x_{step+1} (requested during step-1), xqn
    // receives the own chunk of x_{step+2}
    auto body = [&](int step, u32x4 (&xqc)[2], u32x4 (&xqn)[2]) {
        PROF_T(0);
#ifdef X3_TRACE
        long long* tr = (PROF && a.prof && lane == 0 && step >= 64 && step < 96) ? a.prof + 4096 + ((size_t)(blockIdx.x * 8 + wave) * 32 + (step - 64)) * 8 : nullptr;
#define TR(i) do { if (tr) tr[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TR(i) do { } while (0)
#endif
        TR(0);
        // (Wait counts across the loop back-edge are merged conservatively: the first use of a prefetched x word in a step
        //  costs an `s_waitcnt vmcnt(0)`.  Consume that wait HERE, while only old loads are outstanding, so that it does not
        //  land behind the block fetch and the next prefetches and serialise them with the MFMAs.)
        if constexpr (XRC > 1) asm volatile("" :: "v"(xw[1][0]), "v"(xw[1][1]));
        // ---- x_t W_ih^T (independent of h: this is what fills the exchange latencies); chunk 0 was multiplied at
        // the end of the previous step, the first LDS-resident chunk is fetched here
        lds_fetch(0, C::BIG ? 0 : 2);
        TR(6);
        constexpr int XL_EARLY = 0;                            // LDS chunks multiplied before the blocks are requested (measured: 0 is best)
#ifndef X3_SKIP_PROJ
#pragma unroll
        for (int c = 1; c < XRC; ++c) {
            chunk_mma(xsel(xw[c][0], step), xsel(xw[c][1], step), wxr[c]);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
#ifndef X3_SKIP_PROJ
#pragma unroll
        for (int c = 0; c < XL_EARLY; ++c) lds_chunk_mma(c, xsel(xw[XRC + c][0], step), xsel(xw[XRC + c][1], step));
#endif
        TR(7);
        PROF_E(0); PROF_T(1);

        // ---- h_{step-1}: request the producers' blocks (16-byte loads that bypass the L1) -- optimistically: every word
        // carries its own epoch tag, the peers are in the same phase and published a chunk of MFMAs ago, so the words are
        // normally there; they are checked below, after the MFMAs that hide the flight.
        // The fetch is issued on EVERY step, also on step 0 where its result is not used: inside `if (step > 0)` the
        // compiler has to merge the two paths' wait counts and then waits for these loads as soon as an older load (the x
        // words of the chunks below) is needed -- which serialises the fetch and the MFMAs that are meant to hide it.
        fetch_blocks(step - 1);
        TR(1);
        __builtin_amdgcn_sched_barrier(0);                     // (the MFMAs below stay below the fetch)
        PROF_E(1); PROF_T(0);
        // ---- ... and multiply what is left of x_t W_ih^T while they are in flight
#ifndef X3_SKIP_PROJ
#pragma unroll
        for (int c = XL_EARLY; c < XLC; ++c) {
            if constexpr (C::XSHARE) lds_chunk_mma(c, xsel(xpick(1, 0), step), xsel(xpick(1, 1), step));
            else lds_chunk_mma(c, xsel(xw[XRC + c][0], step), xsel(xw[XRC + c][1], step));
        }
#endif
        // K_in = 512: the x words of the LDS-resident chunks are dead now -- fetch the next step's right away (issued
        // after the block fetch, so the staging wait below does not include them; a full step of latency to hide in)
        if (SPLIT_X) load_x(step + 1, XC_PRE, NXC);
        PROF_E(0); PROF_T(1);
        // ---- check the tags (refetch until every word of the block is of epoch step-1: bounded), stage the blocks in LDS
        // and read this lane's A fragments of the recurrent product from there
        if (step > 0) {
            const unsigned want = tag_of_step(step - 1);
            auto stale = [&]() {
                unsigned m = 0;
#pragma unroll
                for (int i = 0; i < WPL / 4; ++i) m |= (blk[i][0] ^ want) | (blk[i][1] ^ want) | (blk[i][2] ^ want) | (blk[i][3] ^ want);
                return (m & kHTagBit) != 0;
            };
            bool late = stale();
            if (!__all(!late)) {
                if (PROF && prof) pt[5] += 1;                   // steps whose optimistic fetch came too early
                unsigned spins = 0; u64 wt0 = 0;
                do {
                    if (wait_over(spins, spin_budget, wt0, a.max_ticks)) {                // bounded: flag the error and never wait again
                        if (lane == 0) mp_set_error(a.err, 1 + step);
                        spin_budget = 0; poison_cells(cst);
                        break;
                    }
                    if (late) {
                        fetch_blocks(step - 1);
                        late = stale();
                    }
                } while (!__all(!late));
            }
            // (shared with the reduction buffer: every wave must be done with the previous step's partial sums)
            if (C::HT_ALIAS) __syncthreads();
#pragma unroll
            for (int i = 0; i < WPL / 4; ++i)
                *reinterpret_cast<u32x4*>(hT + cdst_off[i]) = blk[i] & u32x4{~kHTagBit, ~kHTagBit, ~kHTagBit, ~kHTagBit};
            __syncthreads();
#pragma unroll
            for (int c = 0; c < NHC; ++c) {
                hw[c][0] = *reinterpret_cast<const u32x4*>(hT + hrd_off[c][0]);
                hw[c][1] = *reinterpret_cast<const u32x4*>(hT + hrd_off[c][1]);
            }
        }
        TR(2);
#ifndef X3_SKIP_XLOAD
        if constexpr (C::BIG) {
            load_x(step + 1, 0, XC_PRE); // next step's x (the scheduler sinks these loads among the MFMAs below: measured better
                                         // than pinning them here, on the critical chain in front of the recurrent product)
        } else if constexpr (C::XSHARE) {
            // own chunk of x_{t+1} (requested a whole step ago): keep it, pass it to the partner wave; request x_{t+2}'s
            xw[0][0] = xqc[0]; xw[0][1] = xqc[1];
            unsigned* xo = xT + ((kq * 2 + tw) * 64 + lane) * 8;
            *reinterpret_cast<u32x4*>(xo) = xqc[0];
            *reinterpret_cast<u32x4*>(xo + 4) = xqc[1];
            load_own(xqn, step + 2);
        } else {
            xw[0][0] = xq[0]; xw[0][1] = xq[1];               // x_{t+1} chunk 0, requested a whole step ago
            load_xq(step + 2);
            load_x(step + 1, 1, XC_PRE);
        }
#endif
        // (issued after the staging wait above, so that this wait does not drain these loads as well)
        PROF_E(1); PROF_T(2);

        // ---- recurrent part: h_{t-1} W_hh^T on top of the input projection
#ifndef X3_SKIP_HMMA
#pragma unroll
        for (int c = 0; c < NHC; ++c) chunk_mma(hw[c][0], hw[c][1], whh[c]);
#else
        acc[0][0] += __uint_as_float(hw[0][0][0] ^ hw[1][1][3]) * 1e-30f;
#endif
        TR(3);
        PROF_E(2); PROF_T(3);

        // ---- K reduction through LDS: finishing wave (dk, tw) takes accumulator reg dk of unit block tw.  With two
        // buffers a step never overwrites what a slower wave may still be reading (it is two barriers behind).
        f32x4* redb = red + (C::RED_BUFS == 2 ? (step & 1) * C::RED_F4 : 0);
        // (LDS-only barriers: __syncthreads() would also drain vmcnt, i.e. wait here for the x_{t+1} prefetch)
        if (C::RED_BUFS == 1) barrier_lds_only();              // previous step's reads of `red` are done
#pragma unroll
        for (int dk = 0; dk < 4; ++dk)
            redb[((tw * 4 + dk) * 4 + kq) * 64 + lane] = f32x4{acc[0][dk], acc[1][dk], acc[2][dk], acc[3][dk]};
        barrier_lds_only();
        if constexpr (C::XSHARE) {                            // the partner's chunk of x_{t+1} (written before this barrier)
            const unsigned* xp = xT + ((kq * 2 + (1 - tw)) * 64 + lane) * 8;
            xw[1][0] = *reinterpret_cast<const u32x4*>(xp);
            xw[1][1] = *reinterpret_cast<const u32x4*>(xp + 4);
        }
        f32x4 gate = redb[(wave * 4 + 0) * 64 + lane];
#pragma unroll
        for (int sw = 1; sw < 4; ++sw) gate += redb[(wave * 4 + sw) * 64 + lane];
        gate = gate * kPairWInv + bias4;                       // (weights were split as 16 w: mp_lstm_dev.h pair_of)
        TR(4);
        PROF_E(3); PROF_T(4);

        // ---- cell update (fp32, register-local), publish h_step as a pair, write the layer output
        const bool act = step < blen;
        const int tt = act ? (d.reverse ? blen - 1 - step : step) : step;
        float oval = 0.f;
        if (act) {
            const float ig = sigmoidf_(gate[0]);
            const float fg = sigmoidf_(gate[1]);
            const float gg = tanhf_(gate[2]);
            const float og = sigmoidf_(gate[3]);
            cst = __builtin_fmaf(fg, cst, ig * gg);    // (explicit: which product gets fused must not depend on the compiler's mood --
                                                       //  mp_lstm_x3w does the same and the two kernels are tested to agree bitwise)
            hst = og * tanhf_(cst);
            oval = hst;
        }
        // the x words of the chunks that are multiplied before the next fetch must have arrived BEFORE the stores below are
        // issued: a wait for them afterwards (one in-order counter) would wait for the stores' acknowledgements as well
#pragma unroll
        for (int c = 0; c < XRC; ++c) asm volatile("" :: "v"(xw[c][0]), "v"(xw[c][1]));
        if constexpr (C::XSHARE) asm volatile("" :: "v"(xw[1][0]), "v"(xw[1][1]));
        const unsigned hp = hpair_of(hst);
        const unsigned hpt = hp | tag_of_step(step);
        const size_t doff = (size_t)(step & 1) * 16 * H;
        store_word_xcd(pdstL + doff, hpt);
        if (!all_local) store_word_dev(pdstR + doff, hpt);
#ifdef X3_SKIP_OUT
        if (inb && step == T - 1) {
#else
        if (inb) {
#endif
            unsigned* op = reinterpret_cast<unsigned*>(d.out + ((size_t)tt * B + bown) * d.outStride + jown);
            store_word_plain(op, a.out_pairs ? (act ? pair_of(hst) : 0u) : __float_as_uint(oval));
        }
        // next step's first register chunk while the stores travel (x_{t+1} was prefetched after the staging above); nobody
        // waits for an acknowledgement: the tagged words are the publication
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#ifndef X3_SKIP_PROJ
        if constexpr (C::XSHARE) chunk_mma(xsel(xpick(0, 0), step + 1), xsel(xpick(0, 1), step + 1), wxr[0]);
        else chunk_mma(xsel(xw[0][0], step + 1), xsel(xw[0][1], step + 1), wxr[0]);
#endif
        __builtin_amdgcn_sched_barrier(0);
        TR(5);
        PROF_E(4);
    };
    if constexpr (C::XSHARE) {
        for (int step = 0; step < T; step += 2) {
            body(step, xqa, xqb);
            if (step + 1 < T) body(step + 1, xqb, xqa);
        }
    } else {
        for (int step = 0; step < T; ++step) body(step, xqa, xqa);
    }
    if (PROF && prof) {
        long long* o = a.prof + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 5; ++i) o[i] = pt[i];
        o[5] = T;
        o[6] = pt[5];
        o[7] = (all_local ? 256 : 0) | my_xcc;
    }
#undef PROF_T
#undef PROF_E

    // ---- final state (h_n, c_n of models/rnn.py:33) back to hbuf / cbuf, fp32
    if (inb) {
        d.hbuf[(size_t)bown * H + jown] = hst;
        d.cbuf[(size_t)bown * H + jown] = cst;
    }
}

// W (4H x K fp32, PyTorch gate order) -> hi / lo bf16x8 B-fragments of the split-bf16 kernel:
//   dst[((((slice*NWV + w)*NC + c)*4 + t)*2 + part)*64 + lane][d]  (32-bit word d = elements e = 2d, 2d+1)
//     = bf16 part of W[t*H + slice*U + tw*16 + (lane&15)][kq*KQ + c*32 + (lane>>4)*8 + e],  w = (kq = w&3, tw = w>>2)
template <int NSLICE>
MP_KERNEL void mp_pack_w_x3(const float* __restrict__ w, unsigned* __restrict__ dst, int K) {
    constexpr int H = 256, U = H / NSLICE, TW = U / 16, NWV = 4 * TW;
    const int KQ = K / 4, NC = KQ / 32;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // one 32-bit word (2 bf16)
    if (idx >= (size_t)4 * H * K) return;
    const int dd = idx & 3;
    const int lane = (idx >> 2) & 63;
    size_t rest = idx >> 8;
    const int part = rest & 1; rest >>= 1;
    const int t = rest & 3; rest >>= 2;
    const int c = rest % NC; rest /= NC;
    const int wv = rest % NWV; rest /= NWV;
    const int slice = (int)rest;
    const int kq = wv & 3, tw = wv >> 2;
    const int row = t * H + slice * U + tw * 16 + (lane & 15);
    const int col = kq * KQ + c * 32 + (lane >> 4) * 8 + 2 * dd;
    const unsigned p0 = wpair_of(w[(size_t)row * K + col]);
    const unsigned p1 = wpair_of(w[(size_t)row * K + col + 1]);
    dst[idx] = part == 0 ? ((p1 & 0xffff0000u) | (p0 >> 16)) : ((p1 << 16) | (p0 & 0xffffu));
}

template <int NSLICE, int KIN>
void launch_x3(const LstmPersistArgs& a, hipStream_t s) {
    using C = CfgX<NSLICE, KIN>;
    const size_t lds = (size_t)C::LDS_BYTES;
    const dim3 grid(((a.nslab * a.ndir + 7) / 8) * 8 * NSLICE);
    if (a.prof || a.debug_drop) hipLaunchKernelGGL((mp_lstm_x3<NSLICE, KIN, true>), grid, dim3(64 * C::NWV), lds, s, a);
    else hipLaunchKernelGGL((mp_lstm_x3<NSLICE, KIN, false>), grid, dim3(64 * C::NWV), lds, s, a);
}

template <int NSLICE, int KIN>
hipError_t x3_attrs() {
    const int lds = (int)CfgX<NSLICE, KIN>::LDS_BYTES;
    hipError_t e = hipFuncSetAttribute((const void*)mp_lstm_x3<NSLICE, KIN, true>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)mp_lstm_x3<NSLICE, KIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace

// H = 256 only, 8 slices; K = 256 | 512
void mp_launch_pack_w_x3(const float* w, float* dst, int K, int nslice, hipStream_t s) {
    const size_t n = (size_t)4 * 256 * K;
    const int grid = (int)((n + 255) / 256);
    (void)nslice;
    hipLaunchKernelGGL((mp_pack_w_x3<8>), dim3(grid), dim3(256), 0, s, w, reinterpret_cast<unsigned*>(dst), K);
}

void mp_launch_lstm_x3(const LstmPersistArgs& a, int KIN, int nslice, hipStream_t s) {
    // (8 slices only: the 16-slice / two-workgroups-per-CU packing was measured slower on every layer -- profiles/NOTES_r01-r03.md 4.2 -- and is
    //  no longer instantiated; `nslice` stays in the signature for the packing helper's sake)
    (void)nslice;
    if (KIN == 256) launch_x3<8, 256>(a, s);
    else launch_x3<8, 512>(a, s);
}

// per-device dynamic-LDS limits of the kernels above (called by mp_create after hipSetDevice, outside of any capture)
hipError_t mp_lstm_x3_device_attrs() {
    hipError_t e = x3_attrs<8, 256>();
    return e != hipSuccess ? e : x3_attrs<8, 512>();
}

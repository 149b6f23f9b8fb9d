// NOTE (round 4): the 16-bit halves of a split operand are IEEE fp16 now (mp_lstm_dev.h pair_of: 24-bit operands, weights split
// as 16 w, MFMA v_mfma_f32_16x16x32_f16), not bf16 as in rounds 1-3 when this file was written; "bf16" in the comments below
// describes the same data path with the other half format.  The hidden-state tag sits in bit 30 of the word (hpair_of).
// K2w -- the split-bf16 persistent LSTM layer of mp_lstm_x3.hip with FOUR 512-register waves per workgroup instead
// of eight 256-register ones.  Same arithmetic, same packed weights (mp_pack_w_x3<8>), same exchange protocol and
// exchange area, same decomposition of a layer into (direction, slab of 16 sequences, slice of 32 hidden units)
// workgroups -- what changes is who does what inside a workgroup:
//   mp_lstm_x3 :  wave (kq, tw) = K quarter kq x unit block tw  (4 gate tiles, 2 waves per SIMD)
//   here       :  wave kq       = K quarter kq x BOTH unit blocks (8 gate tiles, 1 wave per SIMD)
// Why: in mp_lstm_x3 the two waves that share a K quarter load the same x_t words (the MFMA A operand does not depend
// on the unit block), so a CU pulls every x row through its 64 B/clk L1 path twice -- 32 KB (K_in = 256) or 64 KB
// (K_in = 512) per step next to the 16 KB of the hidden-state exchange, and the L1 returns data in order per CU.
// Measured with the duplicate loads replaced by a broadcast load (wrong results): the whole forward 1.89 -> 1.71 ms.
// With one wave per K quarter nothing is loaded twice (x: 16 / 32 KB per step, the h tile is read from LDS once per
// K quarter), a wave may use all 512 registers of its SIMD lane, and K_in = 256 keeps all of W_ih in registers (no
// LDS streaming at all); K_in = 512 keeps two of its four x chunks in LDS exactly as mp_lstm_x3 does.
// The matrix pipe sees the same 96 (144) MFMAs per SIMD and step, now from one wave.
// Recurrent A operand: chunk c of K quarter kq IS row r16 of producer slice 2*kq + c (32 units = one chunk), so a wave
// fetches its A fragments straight from the producers' blocks -- 32 contiguous bytes per lane and chunk, the same 16 KB
// per CU and step as whole blocks -- with no LDS tile and no workgroup barrier between the fetch and the recurrent MFMAs
// (372 -> 354 us on the K_in = 512 layer).  In mp_lstm_x3 the two unit-block waves of a K quarter would both have to
// fetch them (32 KB per CU and step): measured 391 -> 453 us there, which is why that kernel stages blocks in LDS.
// Exchange protocol, data formats, k mapping: see mp_lstm_x3.hip.
#include "mp_lstm_dev.h"
#include <type_traits>

namespace {

// MFMAs as inline asm, so that the B operand (weight fragments) can be named as an AccVGPR: the compiler's own
// v_mfma selection only takes VGPR sources and copies AGPR-resident weights back with four v_accvgpr_read per MFMA --
// issued by the same (only) wave of the SIMD, that halves the MFMA rate.  VOP3P-MAI encodes AccVGPR sources directly.
// The accumulator is an AccVGPR too.  The hazard recogniser does not see inside asm: mfma_drain() supplies the wait
// states the ISA asks for between the last MFMA and a VALU / LDS read of its result.
template <bool B_IN_AGPR>
static __device__ __forceinline__ void mfma_x(f32x4& c, u32x4 a, u32x4 b) {
    if constexpr (B_IN_AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "a"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// first MFMA of an accumulation: C = 0 as an inline constant (no VALU write of the AccVGPRs in front of an MFMA that
// the hazard recogniser cannot see)
template <bool B_IN_AGPR>
static __device__ __forceinline__ void mfma_x0(f32x4& c, u32x4 a, u32x4 b) {
    if constexpr (B_IN_AGPR) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "a"(b));
    else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
// gfx950 wants two wait states between a VALU write of a VGPR and an MFMA that reads it as SrcA / SrcB (the compiler
// puts `s_nop` there for its own MFMAs; measured without: the MFMA multiplies the previous chunk's fragment)
static __device__ __forceinline__ void valu_to_mfma(u32x4& hi, u32x4& lo) { asm volatile("s_nop 1" : "+v"(hi), "+v"(lo)); }
// (24 wait states; an 8-pass MFMA needs 11 before a VALU / LDS read of its result.  The accumulators are operands so that
//  no read of them can be scheduled in front of the wait.)
static __device__ __forceinline__ void mfma_drain(f32x4 (&acc)[2][4]) {
    asm volatile("s_nop 15\n\ts_nop 7"
                 : "+a"(acc[0][0]), "+a"(acc[0][1]), "+a"(acc[0][2]), "+a"(acc[0][3]),
                   "+a"(acc[1][0]), "+a"(acc[1][1]), "+a"(acc[1][2]), "+a"(acc[1][3]));
}

template <int KIN>
struct CfgW {
    static constexpr int H = 256, NSLICE = 8, U = 32, UB = 2, NWV = 4;
    static constexpr int KQ = KIN / 4;                // x: K range of one wave (64 | 128)
    static constexpr int NXC = KQ / 32;               // x: chunks per wave (2 | 4)
    static constexpr int NHC = 2;                     // h: chunks per wave (K quarter 64)
    static constexpr int FR = 8;                      // uint4 fragments per (chunk, unit block): (hi, lo) x 4 gate tiles
    static constexpr int CH_U4 = FR * 64;             // uint4 per (wave of mp_lstm_x3, chunk) in the packed weights
    static constexpr bool BIG = KIN > H;
    static constexpr int XLC = BIG ? 2 : 0;           // x chunks whose W_ih fragments stream from LDS
    static constexpr int XRC = NXC - XLC;             // x chunks whose W_ih fragments live in registers (2 | 2)
    static constexpr int RED_F4 = NWV * 4 * UB * 64;  // one reduction buffer: [finishing wave][source kq][unit block][lane]
    static constexpr int RED_BUFS = BIG ? 1 : 2;
    static constexpr int LDS_BYTES = RED_BUFS * RED_F4 * 16 + NWV * XLC * UB * CH_U4 * 16;
    static_assert(LDS_BYTES <= 160 * 1024, "LDS budget");
};

template <int KIN, bool PROF>
MP_KERNEL __launch_bounds__(256, 1) void mp_lstm_x3w(LstmPersistArgs a) {
    // test hook (mp_debug_drop_workgroup): a workgroup that never shows up.  Only in the PROF instantiation, which the launcher
    // picks when the hook is armed -- the product kernels carry no test code (round 4)
    if (PROF && a.debug_drop && (int)blockIdx.x == a.debug_drop - 1) return;
    using C = CfgW<KIN>;
    constexpr int H = 256, NSLICE = 8, U = 32, UB = 2, NWV = 4, KQ = C::KQ, NXC = C::NXC, NHC = C::NHC, XLC = C::XLC, XRC = C::XRC;
    constexpr int FR = C::FR, CH_U4 = C::CH_U4;
    constexpr int NTHREADS = 64 * NWV;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);
    u32x4* wxl = reinterpret_cast<u32x4*>(smem) + C::RED_BUFS * C::RED_F4;   // [wave][LDS chunk][unit block][fragment][lane]

    // block -> (cluster = (direction, slab), slice): as mp_lstm_x3 (slices of a cluster share an XCD)
    const int ncl = a.ndir * a.nslab;
    const int cl = ((int)(blockIdx.x >> 3) / NSLICE) * 8 + (int)(blockIdx.x & 7);
    const int slice = (int)(blockIdx.x >> 3) % NSLICE;
    if (cl >= ncl) return;
    const int dir = cl / a.nslab, slab = cl % a.nslab;
    const LstmDir d = a.d[dir];
    if (a.hx_next != nullptr) rearm_exchange(a.hx_next + (size_t)cl * ((size_t)4 * 16 * H + 16), 1, slice, NSLICE, threadIdx.x, NTHREADS);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kq = wave;
    const int q = lane >> 4, r16 = lane & 15;
    const int B = a.B, T = a.T;
    const int brow0 = (a.slab0 + slab) * 16;

    // ---- weights.  Packed as for mp_lstm_x3<8, KIN>: [slice][wave8 = tw*4 + kq][chunk][fragment][lane]
    auto wsrc = [&](const float* pack, int nchunk, int ub, int c) {
        return reinterpret_cast<const u32x4*>(pack) + ((size_t)(slice * 8 + ub * 4 + kq) * nchunk + c) * CH_U4;
    };
    if constexpr (XLC > 0) {
        for (int w = 0; w < NWV; ++w)
            for (int c = 0; c < XLC; ++c)
                for (int ub = 0; ub < UB; ++ub) {
                    const u32x4* src = reinterpret_cast<const u32x4*>(d.wihpack) + ((size_t)(slice * 8 + ub * 4 + w) * NXC + XRC + c) * CH_U4;
                    u32x4* dst = wxl + ((size_t)(w * XLC + c) * UB + ub) * CH_U4;
                    for (int i = threadIdx.x; i < CH_U4; i += NTHREADS) dst[i] = src[i];
                }
    }
    u32x4 wxr[XRC][UB][FR];
#pragma unroll
    for (int c = 0; c < XRC; ++c)
#pragma unroll
        for (int ub = 0; ub < UB; ++ub)
#pragma unroll
            for (int f = 0; f < FR; ++f) wxr[c][ub][f] = wsrc(d.wihpack, NXC, ub, c)[f * 64 + lane];
    u32x4 whh[NHC][UB][FR];
#pragma unroll
    for (int c = 0; c < NHC; ++c)
#pragma unroll
        for (int ub = 0; ub < UB; ++ub)
#pragma unroll
            for (int f = 0; f < FR; ++f) whh[c][ub][f] = wsrc(d.wpack, NHC, ub, c)[f * 64 + lane];

    // ---- the two (sequence, unit) pairs this lane finishes: row q*4 + wave, units ub*16 + r16
    const int bown = brow0 + q * 4 + wave;
    const bool inb = bown < B;
    const int blen = inb ? a.lengths[bown] : 0;
    int jown[UB];
    f32x4 bias4[UB];
    float cst[UB], hst[UB];
#pragma unroll
    for (int ub = 0; ub < UB; ++ub) {
        jown[ub] = slice * U + ub * 16 + r16;
        bias4[ub] = *reinterpret_cast<const f32x4*>(d.bias + 4 * jown[ub]);
        cst[ub] = (inb && !a.zero_state) ? d.cbuf[(size_t)bown * H + jown[ub]] : 0.f;
        hst[ub] = (inb && !a.zero_state) ? d.hbuf[(size_t)bown * H + jown[ub]] : 0.f;
    }

    // ---- A-operand row of this lane (row r16 of the slab)
    const int arow = brow0 + r16;
    const bool arow_in = arow < B;
    const int alen = arow_in ? a.lengths[arow] : 0;
    const unsigned* xbase = reinterpret_cast<const unsigned*>(d.xin) + (size_t)(arow_in ? arow : 0) * KIN + kq * KQ + q * 8;
    const size_t xtstride = (size_t)B * KIN;

    u32x4 hw[NHC][2];
    {
        const float* p = d.hin + (size_t)(arow_in ? arow : 0) * H + kq * 64 + q * 8;
#pragma unroll
        for (int c = 0; c < NHC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                hw[c][e >> 2][e & 3] = (arow_in && !a.zero_state) ? pair_of(p[c * 32 + e]) : 0u;
    }

    // ---- exchange area of this cluster (layout of mp_lstm_x3)
    constexpr size_t SLABW = (size_t)4 * 16 * H + 16;
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
        if (a.force_remote) { all_local = false; same = 0; }
    }
    // consumer role: chunk c of this wave's K quarter = row r16 of producer slice 2*kq + c, units q*8 .. q*8+7
    const __amdgpu_buffer_rsrc_t hxrsrc = __builtin_amdgcn_make_buffer_rsrc(hxw, 0, (int)(SLABW * 8), 0x27000);
    int dsrc_byte[NHC];
#pragma unroll
    for (int c = 0; c < NHC; ++c) {
        const int p = 2 * kq + c;
        const unsigned* src = (((same >> p) & 1) ? dataL : dataR) + (size_t)p * 16 * U + r16 * U + q * 8;
        dsrc_byte[c] = (int)((src - hxw) * 4);
    }
    // producer role: this lane's two words (row q*4 + wave, units ub*16 + r16) of the slice's block
    unsigned* pdstL = dataL + (size_t)slice * 16 * U + (q * 4 + wave) * U + r16;
    unsigned* pdstR = dataR + (size_t)slice * 16 * U + (q * 4 + wave) * U + r16;

    // ---- x: pair words of this lane's row, chunk c: k = kq*KQ + c*32 + q*8 + e.  Loads are unconditional (clamped row),
    // rows past their length are zeroed where the words are used (see mp_lstm_x3.hip, wait-count hygiene).
    // All of x is requested TWO steps ahead into two register sets that alternate (the step loop is unrolled by two, so
    // no register copies -- and no wait for a fresh load -- sit on the loop's back edge): x_{t+2} is requested right after
    // the tag check of step t, into the registers of x_t, which is dead by then; nothing in the loop ever waits for an x
    // load (K_in = 512: 1.806 -> 1.77 ms for the whole forward against loading x_{t+1} during step t)
    typedef u32x4 XBuf[NXC][2];
    XBuf xa, xb;
    auto load_x = [&](XBuf& xw, int step, int c0, int c1) {
        const bool on = step < alen;
        const int t = on ? (d.reverse ? alen - 1 - step : step) : 0;
        const unsigned* p = xbase + (size_t)t * xtstride;
#pragma unroll
        for (int c = 0; c < NXC; ++c)
            if (c >= c0 && c < c1) {
                xw[c][0] = *reinterpret_cast<const u32x4*>(p + c * 32);
                xw[c][1] = *reinterpret_cast<const u32x4*>(p + c * 32 + 4);
            }
    };
    auto xsel = [&](u32x4 v, int step) { return step < alen ? v : u32x4{0u, 0u, 0u, 0u}; };
    load_x(xa, 0, 0, NXC);
    load_x(xb, 1, 0, NXC);
    __syncthreads();                                          // W_ih LDS image complete

    long long pt[6] = {0, 0, 0, 0, 0, 0};
    const bool prof = PROF && a.prof != nullptr && threadIdx.x == 0;
#define PROF_T(i) do { if (PROF && prof) pt[i] -= (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define PROF_E(i) do { if (PROF && prof) pt[i] += (long long)__builtin_amdgcn_s_memtime(); } while (0)

    f32x4 acc[UB][4];
    // one chunk of a matrix product for both unit blocks: per block 4 MFMAs per gate tile (hi*hi, hi*lo, lo*hi, lo*lo)
    auto chunk_mma = [&](auto in_agpr, u32x4 x0, u32x4 x1, const u32x4 (&w)[UB][FR], bool first = false) {
        constexpr bool AG = decltype(in_agpr)::value;
        u32x4 ahi, alo;
        split_pairs(x0, x1, ahi, alo);
        valu_to_mfma(ahi, alo);
#pragma unroll
        for (int ub = 0; ub < UB; ++ub)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (first) mfma_x0<AG>(acc[ub][t], ahi, w[ub][2 * t]);
                else mfma_x<AG>(acc[ub][t], ahi, w[ub][2 * t]);
            }
#pragma unroll
        for (int ub = 0; ub < UB; ++ub)
#pragma unroll
            for (int t = 0; t < 4; ++t) mfma_x<AG>(acc[ub][t], ahi, w[ub][2 * t + 1]);
#pragma unroll
        for (int ub = 0; ub < UB; ++ub)
#pragma unroll
            for (int t = 0; t < 4; ++t) mfma_x<AG>(acc[ub][t], alo, w[ub][2 * t]);
#pragma unroll
        for (int ub = 0; ub < UB; ++ub)
#pragma unroll
            for (int t = 0; t < 4; ++t) mfma_x<AG>(acc[ub][t], alo, w[ub][2 * t + 1]);      // lo*lo (round 6)
    };
    constexpr std::integral_constant<bool, true> IN_A{};
    constexpr std::integral_constant<bool, false> IN_V{};
    // LDS-resident chunk (K_in = 512): the fragments of one unit block at a time through a 32-register buffer
    const u32x4* wxw = wxl + (size_t)wave * XLC * UB * CH_U4 + lane;
    auto lds_chunk_mma = [&](int c, u32x4 x0, u32x4 x1) {
        u32x4 ahi, alo;
        split_pairs(x0, x1, ahi, alo);
        valu_to_mfma(ahi, alo);
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {
            u32x4 wl[FR];
#pragma unroll
            for (int f = 0; f < FR; ++f) wl[f] = wxw[((size_t)(c * UB + ub) * FR + f) * 64];
            // (term order of mp_lstm_x3's LDS-streamed chunks -- hi*hi, lo*hi, hi*lo -- so that the two kernels stay bitwise identical)
#pragma unroll
            for (int t = 0; t < 4; ++t) mfma_x<false>(acc[ub][t], ahi, wl[2 * t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) mfma_x<false>(acc[ub][t], alo, wl[2 * t]);
#pragma unroll
            for (int t = 0; t < 4; ++t) mfma_x<false>(acc[ub][t], ahi, wl[2 * t + 1]);
#pragma unroll
            for (int t = 0; t < 4; ++t) mfma_x<false>(acc[ub][t], alo, wl[2 * t + 1]);      // lo*lo (round 6)
        }
    };
    // the first register chunk of step t+1 is multiplied at the END of step t, right behind the stores of h_t
    chunk_mma(IN_V, xsel(xa[0][0], 0), xsel(xa[0][1], 0), wxr[0], true);

    // the blocks of the h written at step `pstep` (parity pstep & 1)
    u32x4 blk[2 * NHC];
    auto fetch_blocks = [&](int pstep) {
        const int poff_b = (pstep & 1) * 16 * H * 4;
#pragma unroll
        for (int i = 0; i < 2 * NHC; ++i)
            blk[i] = __builtin_amdgcn_raw_buffer_load_b128(hxrsrc, dsrc_byte[i >> 1] + 16 * (i & 1) + poff_b, 0, 16 /* sc1 */);
    };

    // one time step; xc = x_step (chunk 0 already multiplied), xn = x_{step+1}  (K_in = 512: the same set)
    auto body = [&](int step, XBuf& xc, XBuf& xn) {
        PROF_T(0);
        // ---- h_{step-1}: request the producers' blocks optimistically (every word carries its epoch tag); K_in = 512
        // multiplies its second register chunk first, which puts the request where mp_lstm_x3 has it
        if constexpr (C::BIG) {
            chunk_mma(IN_A, xsel(xc[1][0], step), xsel(xc[1][1], step), wxr[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        PROF_E(0); PROF_T(1);
        fetch_blocks(step - 1);
        __builtin_amdgcn_sched_barrier(0);
        PROF_E(1); PROF_T(0);
        // ---- what is left of x_t W_ih^T runs under the flight of the blocks
        if constexpr (C::BIG) {
#pragma unroll
            for (int c = 0; c < XLC; ++c) lds_chunk_mma(c, xsel(xc[XRC + c][0], step), xsel(xc[XRC + c][1], step));
        } else {
            chunk_mma(IN_A, xsel(xc[1][0], step), xsel(xc[1][1], step), wxr[1]);
        }
        PROF_E(0); PROF_T(1);
        // ---- check the tags (refetch while any word is stale: bounded); no staging: the words are the A fragments
        if (step > 0) {
            const unsigned want = tag_of_step(step - 1);
            auto stale = [&]() {
                unsigned m = 0;
#pragma unroll
                for (int i = 0; i < 2 * NHC; ++i) m |= (blk[i][0] ^ want) | (blk[i][1] ^ want) | (blk[i][2] ^ want) | (blk[i][3] ^ want);
                return (m & kHTagBit) != 0;
            };
            bool late = stale();
            if (!__all(!late)) {
                if (PROF && prof) pt[5] += 1;
                unsigned spins = 0; u64 wt0 = 0;
                do {
                    if (wait_over(spins, spin_budget, wt0, a.max_ticks)) {
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
            // the fetched words ARE the A fragments (tags cleared)
#pragma unroll
            for (int c = 0; c < NHC; ++c) {
                hw[c][0] = blk[2 * c] & u32x4{~kHTagBit, ~kHTagBit, ~kHTagBit, ~kHTagBit};
                hw[c][1] = blk[2 * c + 1] & u32x4{~kHTagBit, ~kHTagBit, ~kHTagBit, ~kHTagBit};
            }
        }
        // ---- next x (after the wait for the blocks, so that this wait does not drain these loads as well)
        load_x(xc, step + 2, 0, NXC);                       // (x_step is dead: its registers take x_{step+2})
        PROF_E(1); PROF_T(2);

        // ---- recurrent part: h_{t-1} W_hh^T on top of the input projection
#pragma unroll
        for (int c = 0; c < NHC; ++c) chunk_mma(IN_A, hw[c][0], hw[c][1], whh[c]);
        mfma_drain(acc);
        PROF_E(2); PROF_T(3);

        // ---- K reduction through LDS: finishing wave dk takes accumulator reg dk of both unit blocks
        f32x4* redb = red + (C::RED_BUFS == 2 ? (step & 1) * C::RED_F4 : 0);
        if (C::RED_BUFS == 1) barrier_lds_only();
#pragma unroll
        for (int dk = 0; dk < 4; ++dk)
#pragma unroll
            for (int ub = 0; ub < UB; ++ub)
                redb[((dk * 4 + kq) * UB + ub) * 64 + lane] = f32x4{acc[ub][0][dk], acc[ub][1][dk], acc[ub][2][dk], acc[ub][3][dk]};
        barrier_lds_only();
        f32x4 gate[UB];
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {
            gate[ub] = redb[((wave * 4 + 0) * UB + ub) * 64 + lane];
#pragma unroll
            for (int sw = 1; sw < 4; ++sw) gate[ub] += redb[((wave * 4 + sw) * UB + ub) * 64 + lane];
            gate[ub] = gate[ub] * kPairWInv + bias4[ub];       // (weights were split as 16 w: mp_lstm_dev.h pair_of)
        }
        PROF_E(3); PROF_T(4);

        // ---- cell update (fp32, register-local), publish h_step, write the layer output
        const bool act = step < blen;
        const int tt = act ? (d.reverse ? blen - 1 - step : step) : step;
        // (branch-free and written "across" the two cells, so that the only wave of the SIMD has two independent
        //  dependency chains to interleave: v_exp / v_rcp are quarter-rate with long latencies)
        unsigned hpt[UB], ow[UB];
        {
            float ig[UB], fg[UB], gg[UB], og[UB], cn[UB], hn[UB];
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) {
                ig[ub] = sigmoidf_(gate[ub][0]);
                fg[ub] = sigmoidf_(gate[ub][1]);
                gg[ub] = tanhf_(gate[ub][2]);
                og[ub] = sigmoidf_(gate[ub][3]);
            }
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) cn[ub] = __builtin_fmaf(fg[ub], cst[ub], ig[ub] * gg[ub]);
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) hn[ub] = og[ub] * tanhf_(cn[ub]);
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) {
                cst[ub] = act ? cn[ub] : cst[ub];
                hst[ub] = act ? hn[ub] : hst[ub];
                hpt[ub] = hpair_of(hst[ub]) | tag_of_step(step);
                ow[ub] = a.out_pairs ? (act ? pair_of(hst[ub]) : 0u) : (act ? __float_as_uint(hst[ub]) : 0u);
            }
        }
        const size_t doff = (size_t)(step & 1) * 16 * H;
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) store_word_xcd(pdstL + doff + ub * 16, hpt[ub]);
        if (!all_local) {
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) store_word_dev(pdstR + doff + ub * 16, hpt[ub]);
        }
        if (inb) {
            unsigned* op = reinterpret_cast<unsigned*>(d.out + ((size_t)tt * B + bown) * d.outStride + jown[0]);
#pragma unroll
            for (int ub = 0; ub < UB; ++ub) store_word_plain(op + ub * 16, ow[ub]);
        }
        // next step's first register chunk while the stores travel
        __builtin_amdgcn_sched_barrier(0);
        chunk_mma(IN_V, xsel(xn[0][0], step + 1), xsel(xn[0][1], step + 1), wxr[0], true);
        __builtin_amdgcn_sched_barrier(0);
        PROF_E(4);
    };
    for (int step = 0; step < T; step += 2) {
        body(step, xa, xb);
        if (step + 1 < T) body(step + 1, xb, xa);
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
#pragma unroll
        for (int ub = 0; ub < UB; ++ub) {
            d.hbuf[(size_t)bown * H + jown[ub]] = hst[ub];
            d.cbuf[(size_t)bown * H + jown[ub]] = cst[ub];
        }
    }
}

template <int KIN>
void launch_x3w(const LstmPersistArgs& a, hipStream_t s) {
    using C = CfgW<KIN>;
    const size_t lds = (size_t)C::LDS_BYTES;
    const dim3 grid(((a.nslab * a.ndir + 7) / 8) * 8 * 8);
    if (a.prof || a.debug_drop) hipLaunchKernelGGL((mp_lstm_x3w<KIN, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((mp_lstm_x3w<KIN, false>), grid, dim3(256), lds, s, a);
}

template <int KIN>
hipError_t x3w_attrs() {
    const int lds = (int)CfgW<KIN>::LDS_BYTES;
    hipError_t e = hipFuncSetAttribute((const void*)mp_lstm_x3w<KIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)mp_lstm_x3w<KIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace

// H = 256, 8 slices (weights packed by mp_launch_pack_w_x3(.., nslice = 8)); KIN = 256 | 512
void mp_launch_lstm_x3w(const LstmPersistArgs& a, int KIN, hipStream_t s) {
    if (KIN == 256) launch_x3w<256>(a, s);
    else launch_x3w<512>(a, s);
}

// per-device dynamic-LDS limits (called by mp_create after hipSetDevice, outside of any capture)
hipError_t mp_lstm_x3w_device_attrs() {
    hipError_t e = x3w_attrs<256>();
    return e != hipSuccess ? e : x3w_attrs<512>();
}

// K2 -- persistent, fused nn.LSTM layer (models/rnn.py:27): ONE launch computes, for all T time steps of up to
// two directions,   gates_t = x_t W_ih^T + (b_ih + b_hh) + h_{t-1} W_hh^T ;  i,f,g,o = s,s,tanh,s ;
//                   c_t = f c_{t-1} + i g ;  h_t = o tanh(c_t)            (PyTorch gate order i,f,g,o)
// with packed-sequence semantics (rnn.py:25-31): sequence b is active for steps s < len_b, forward
// visits t = s, reverse t = len_b-1-s, inactive rows keep (h,c) and the padded output position is written 0.
//
// Why fused and persistent (gfx950): the recurrence is a chain of T dependent steps, each a tiny GEMM, so
// a step is latency-bound (hidden state must cross workgroups) while the input projection is a big
// MFMA-bound GEMM with no dependence on h.  Doing both in one persistent kernel lets the x_t W_ih^T MFMAs
// of step t run exactly where the kernel would otherwise sit waiting for h_{t-1} from its peers, and
// removes the [B*T, 4H] gate pre-activation round trip through HBM (2 KB/frame/direction written + read).
//
// Mapping: a workgroup owns (direction, slab of 16 sequences = the M of v_mfma_f32_16x16x4_f32, slice of
// U hidden units); its 4 waves split every K range four ways.  Everything a step needs stays on chip:
//   * W_hh slice: registers (H=256, U=32: 128 VGPRs per lane), loaded once in B-fragment order;
//   * W_ih slice: LDS (up to 128 KB as 16-byte B-fragment groups) and, when K_in = 2H, the rest in registers;
//   * c_t and the lane's own h_t: registers for all T steps;
//   * x_{t+1}: prefetched from HBM into registers while step t finishes (16-byte loads, 64-byte segments).
// h_t crosses workgroups (the NSLICE slices of a slab need each other's units every step) in one of two forms:
//   * 8-byte {epoch, value} granules -- the data is its own flag, no fence / barrier / flag word (cdna_hip_programming.md
//     Guideline 16, form R2): the H = 64 kernels (described in this header);
//   * plain 4-byte words in 16-byte pieces with a one-bit tag in bit 30 of every word -- free, |h| <= 1 -- ("TAGX" in the kernel
//     body): the H = 256 kernels, where 16 granule loads + 16 tag compares per lane and step cost more than the hand-off
//     latency they hid; one L2 round trip per step (rounds 2-4 signalled these words with a flag per producer wave: two).
// Round 5: the K_in = 256 kernel on 8 slices also runs the unidirectional velocity block as a two-layer wavefront (WF) and
// carries the H = 64 foot-contact layers as riders (FK) -- see the comment at the kernel.
// Two transports, chosen per PRODUCER from where it really runs (its XCC id,
// published once at kernel start), so the result never depends on placement, only the speed does:
//   R: write-through (sc1) stores + sc1 loads -- coherent for any placement (fabric round trip, ~1 us);
//   L: ordinary stores + L1-bypassing (sc1) loads -- producer and consumer share an XCD (one L2), ~4x faster.
// Granules are laid out [unit/4][row][unit%4] so a consumer wave's 64 lanes (16 rows x 4 consecutive units)
// read 512 contiguous bytes per instruction and feed the MFMA A operand straight from the load registers.
// They are requested half-way through the x-projection MFMAs and validated before use, so their latency is
// normally hidden; a stale granule falls back to a bounded poll (cheap gate + sweep).  The 4 K-partials
// meet in LDS and every wave finishes a quarter of the (sequence, unit) pairs.
// All workgroups of a launch must be co-resident (grid <= 256, one per CU); every spin is bounded and a
// timeout raises a device-side error word instead of hanging the GPU.
#include "mp_lstm_dev.h"

namespace {

// granule index of (row, hidden unit j) inside one [16][H] slab-parity block
__device__ __forceinline__ int granule_index(int row, int j) { return (((j >> 2) * 16 + row) << 2) + (j & 3); }

// Wave layout: four waves, one per SIMD.  Wave kq takes K quarter kq of both GEMM parts and computes all 4 * NUB gate tiles.
// (Rounds 1-4 also had an eight-wave layout, two waves per SIMD with <= 256 registers each; the four-wave kernels with
//  AccVGPR-resident weights replaced it in round 3 and it was removed in round 5 -- git has it.)
template <int H, int NSLICE, int KIN>
struct Cfg {
    static constexpr int U = H / NSLICE;              // hidden units per workgroup (16 | 32)
    static constexpr int NUB = U / 16;                // 16-unit blocks per workgroup (1 | 2)
    static constexpr int NWV = 4;                     // waves per workgroup
    static constexpr int NTW = 4 * NUB;               // MFMA tiles per wave
    static constexpr int NTG = NTW / 4;               // 16-byte groups of 4 tiles per wave
    static constexpr int KW = H / 4;                  // h: K range of one wave (64 | 16)
    static constexpr int NKS = KW / 4;                // h: k-steps per wave (16 | 4)
    static constexpr int KQ = KIN / 4;                // x: K range of one wave
    static constexpr int NXS = KQ / 4;                // x: k-steps per wave
    static constexpr int NXJ = KQ / 16;               // x: 16-byte loads per lane per step
    // PER_UB (one unit block per workgroup): finishing wave kq takes accumulator reg kq of every gate tile; otherwise
    // (two unit blocks) a wave finishes NOWN regs of one block
    static constexpr bool PER_UB = NUB == 1;
    static constexpr int NOWN = PER_UB ? 1 : 2;
    static constexpr int NPW = NSLICE >= 4 ? NSLICE / 4 : 1;   // producer slices inside one wave's K quarter
    static constexpr int RED_F4 = NWV * 4 * NOWN * 64;   // float4 slots: [finishing wave][source kq][o][lane]
    // LDS budget 160 KB: reduction scratch + as many x k-steps of W_ih as fit; the rest lives in registers
    static constexpr int STEP_BYTES = NWV * NTG * 64 * 16;                     // all waves, one k-step
    static constexpr int LDS_BUDGET = NSLICE == 16 ? 80 * 1024 : 160 * 1024;   // 2 workgroups per CU
    static constexpr int LDS_STEPS_MAX = (LDS_BUDGET - RED_F4 * 16) / STEP_BYTES;
    // (four-wave 8-slice configuration, K_in = 256: ALL of W_ih fits the AccVGPRs beside W_hh -- no weight comes from LDS)
    static constexpr bool ALLREG = H == 256 && NSLICE == 8 && KIN == 256;
    static constexpr int XL = ALLREG ? 0 : (NXS <= LDS_STEPS_MAX ? NXS : (LDS_STEPS_MAX / 4) * 4);  // x k-steps served from LDS
    static constexpr int XR = NXS - XL;                                        // x k-steps served from registers
    static constexpr bool BIG = KIN > H;              // part of W_ih in registers: late gather, split x prefetch
    static constexpr int WG_PER_CU = NSLICE == 16 ? 2 : 1;                     // co-resident workgroups wanted
    static_assert(NUB == 1 || NUB == 2, "16 or 32 hidden units per workgroup");
};

// WREG configuration (H = 256, 8 slices, FOUR waves with up to 512 registers each instead of eight with 256): the
// register-resident weight fragments live in AccVGPRs and every MFMA is inline asm that names them as its B operand
// (VOP3P-MAI encodes an AccVGPR source directly; left to itself the compiler parks the 256 weight registers in AccVGPRs
// and copies each one back with a v_accvgpr_read -- a VALU instruction per MFMA).  The hazard recogniser does not look
// inside asm, so the two hazards that exist here are handled by hand: accumulators start from an inline-constant C = 0
// MFMA (no VALU-zeroed register is read as SrcC), and mfma_drain() supplies the wait states between the last MFMA
// and the first VALU / LDS read of its result (8-pass MFMA: 11 by the ISA tables, 18 given).
template <bool ZERO, bool WACC>
__device__ __forceinline__ void mfma_asm(f32x4& c, float a, float w) {
    if (ZERO) {
        if (WACC) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(w));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(w));
    } else {
        if (WACC) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(w));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(w));
    }
}
__device__ __forceinline__ void mfma_drain() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// FK != 0 (the K_in = 256 kernels on 16 or on 8 slices): a bidirectional H = 64, K_in = FK layer -- the foot-contact block of the
// same slab -- rides along: a slice also computes 8 units of one direction of it (two extra MFMA tiles: gates i|f and g|o of 8
// units), K split over the four waves like everything else, its hidden state exchanged through the same area with the same
// kind of tagged words.  16 slices (one direction of the carrier): slice j carries units 8*(j & 7) .. +7 of rider direction
// j >> 3.  8 slices (two clusters per slab): slice j of cluster direction d carries units 8*j .. +7 of rider direction d.
// Why: run as a kernel of its own beside this one, the H = 64 layer and this layer slow each other on every shared SIMD
// (profiles/r03_class_times.txt: 335 -> 398 us and 180 -> 450 us per layer); inside these waves it costs its instructions and
// nothing else.
//
// WF (round 5; <256, 8, 256> only): the two "directions" of the launch are the two LAYERS of a unidirectional 2-layer LSTM
// (the velocity block, models/velocity.py:29) -- a wavefront: d[0] = layer 0, d[1] = layer 1 with d[1].xin = d[0].out, both
// forward in time.  Layer 1 at time t needs layer 0's output at time t and nothing else of layer 0, so the two layers of a
// slab run side by side, layer 1 one to two steps behind, at the speed of ONE 8-slice launch instead of two 16-slice launches
// that each carry a full step's fixed cost on half the MFMAs (rounds 3-4: 2 x 397 us at 0.62 of peak).
//   * cluster index: cl = slab * 2 + layer, so that the four clusters of an XCD are two slabs x both layers and the link
//     stays inside one L2 (placement only decides speed: the transport is chosen from the real XCC ids, as for h);
//   * link layer 0 -> layer 1: layer 1 reads layer 0's OUTPUT buffer ([T][B][256], written once per step and never
//     overwritten within a launch), not the two-slot exchange words -- nothing holds layer 0 back, it may run any number of
//     steps ahead.  Every layer-0 wave raises ONE progress word (epoch_base + number of finished steps) after the
//     s_waitcnt vmcnt(0) that acknowledges its output stores -- at the second k-step of the next step's projection, where the
//     stores are long done -- and once more after its last step; a layer-1 wave requests the 8 words of the two slices its K
//     quarter of x comes from at the top of a step and compares (>=) before it issues the prefetch of x_{t+1}: off the
//     critical path once layer 1 has fallen far enough behind that the words are there at first look (a wait makes it fall
//     behind further: the lag settles by itself).  x is read with sc1 loads (never through a CU's L1: a line is requested
//     only after its producers said it is complete); when the two clusters do not share an XCD the output stores and the
//     progress words are written through (sc1), like the R transport of h.
template <int H, int NSLICE, int KIN, bool PROF, int FK = 0, bool WF = false>
MP_KERNEL __launch_bounds__(256, (FK || WF ? 1 : Cfg<H, NSLICE, KIN>::WG_PER_CU)) void mp_lstm_fused(LstmPersistArgs a) {
    // test hook (mp_debug_drop_workgroup): a workgroup that never shows up.  Only in the PROF instantiation, which the launcher
    // picks when the hook is armed -- the product kernels carry no test code (round 4)
    if (PROF && a.debug_drop && (int)blockIdx.x == a.debug_drop - 1) return;
    const long long tl_entry = PROF ? (long long)__builtin_amdgcn_s_memrealtime() : 0;   // (PROF: launch timeline, 100 MHz)
    using C = Cfg<H, NSLICE, KIN>;
    constexpr bool WREG = H == 256 && NSLICE == 8;
    constexpr int U = C::U, NUB = C::NUB, NWV = C::NWV, NTW = C::NTW, NTG = C::NTG, KW = C::KW, NKS = C::NKS, KQ = C::KQ;
    // (FK: one workgroup per CU and 512 registers per lane to spend: the W_ih slice lives in registers, not in LDS)
    constexpr int NXS = C::NXS, NXJ = C::NXJ, NOWN = C::NOWN, XL = FK ? 0 : C::XL, XR = NXS - XL, NPW = C::NPW;
    constexpr bool PER_UB = C::PER_UB;
    constexpr bool TAGX = H == 256;                                   // plain words with a tag in bit 30 (below); H = 64: granules
    constexpr int NP = NKS / 4 > 0 ? NKS / 4 : 1;                     // TAGX: 16-byte pieces of h per lane (4 k-steps each)
    constexpr int PPP = NP / NPW > 0 ? NP / NPW : 1;                  //       pieces per producer slice
    // k-steps of the projection in front of the request for h_{t-1}: half of them (one k-step is 256 cycles in the 8-slice
    // kernels and 128 in the 16-slice ones)
    // (round 5 A/B on one box, 2/8, 3/8, 5/8 of the projection in front of the request instead of 4/8: 3.635 / 3.631 / 3.629 vs
    //  3.625 ms per step, outputs bit-identical -- the ~530 cycles of the validation phase are its instructions, not a wait)
    constexpr int XSPLIT = NXS / 2;
    constexpr int NTHREADS = 64 * NWV;
    static_assert(!WF || (H == 256 && NSLICE == 8 && KIN == 256), "the two-layer wavefront lives in the all-register 8-slice kernel");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);                       // [finishing wave][source kq][o][lane]
    f32x4* wxl = reinterpret_cast<f32x4*>(smem) + C::RED_F4;           // [wave][x-step < XL][tile group][lane]

    // 1-D grid of 8 * max(xcd_cnt) * NSLICE blocks over (cluster = (direction, slab), slice).  The NSLICE workgroups
    // of a cluster get block ids that are congruent mod 8, i.e. (dispatch: block b -> XCD b % 8) they share an XCD and
    // its L2 -- speed only, never correctness (the transport is chosen from the real XCC ids); which clusters an XCD
    // runs is the host's table (LstmPersistArgs::xcd_cnt / xcd_base).  Blocks beyond their XCD's count exit at once.
    const int ncl = a.ndir * a.nslab;
    const int xcd = a.xcd_physical ? (int)(xcc_id() & 7) : (int)(blockIdx.x & 7);
    const int kth = (int)(blockIdx.x >> 3) / NSLICE;                                     // the XCD's kth cluster
    const int slice = (int)(blockIdx.x >> 3) % NSLICE;
    if (kth >= mp_xcd_count(a, xcd)) return;
    const int cl = mp_xcd_first(a, xcd) + kth;
    if (cl >= ncl) return;
    const int dir = WF ? (cl & 1) : cl / a.nslab, slab = WF ? (cl >> 1) : cl % a.nslab;
    const LstmDir d = a.d[dir];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kq = wave;
    const int q = lane >> 4, r16 = lane & 15;
    const int B = a.B, T = a.T;
    const int brow0 = (a.slab0 + slab) * 16;

    // ---- W_ih slice: k-steps [0, XL) -> LDS, [XL, NXS) -> registers; W_hh slice -> registers
    f32x4 wxr[XR > 0 ? XR : 1][NTG];
    if (XR > 0) {
        const f32x4* src = reinterpret_cast<const f32x4*>(d.wihpack) + ((size_t)(slice * NWV + wave) * NXS + XL) * NTG * 64 + lane;
#pragma unroll
        for (int s = 0; s < XR; ++s)
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) wxr[s][tg] = src[(size_t)(s * NTG + tg) * 64];
    }
    float wv[NKS][NTW];
    {
        const float* wp = d.wpack + ((size_t)(slice * NWV + wave) * NKS * NTW) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int t = 0; t < NTW; ++t) wv[ks][t] = wp[(size_t)(ks * NTW + t) * 64];
    }
    // the LDS image of W_ih, AFTER the register-resident weights have been requested (round 4): in batches of 16 pieces per
    // thread, all requested before the first is written.  The plain copy loop compiled into load -> wait -> write per piece for
    // most of its trips -- 16 dependent memory round trips -- and the register loads were only issued behind it: first step of a
    // K_in = 512 launch 10.3 -> 7.7 us after kernel entry (tools/debug/launch_timeline.py; profiles/r03_launch_timeline.txt).
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(d.wihpack) + (size_t)slice * NWV * NXS * NTG * 64;
        constexpr int PER_W = XL * NTG * 64;                     // pieces per wave image
        if constexpr (PER_W > 0 && PER_W % NTHREADS == 0 && (NWV * (PER_W / NTHREADS)) % 16 == 0) {
            constexpr int TRIPS = PER_W / NTHREADS, TOTAL = NWV * TRIPS;
#pragma unroll 1
            for (int b0 = 0; b0 < TOTAL; b0 += 16) {
                f32x4 tmp[16];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int w = (b0 + k) / TRIPS, i = threadIdx.x + ((b0 + k) % TRIPS) * NTHREADS;
                    tmp[k] = src[(size_t)w * NXS * NTG * 64 + i];
                }
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int w = (b0 + k) / TRIPS, i = threadIdx.x + ((b0 + k) % TRIPS) * NTHREADS;
                    wxl[(size_t)w * PER_W + i] = tmp[k];
                }
            }
        } else {
            for (int w = 0; w < NWV; ++w)
                for (int i = threadIdx.x; i < PER_W; i += NTHREADS)
                    wxl[(size_t)w * PER_W + i] = src[(size_t)w * NXS * NTG * 64 + i];
        }
    }

    // ---- the (sequence, unit) pairs this lane finishes: accumulator regs of tile column r16
    const int ubo = PER_UB ? 0 : (kq & 1);                                 // unit block this wave finishes
    const int reg0 = PER_UB ? kq : 2 * (kq >> 1);                          // first accumulator reg it finishes
    const int jown = slice * U + ubo * 16 + r16;                           // hidden unit
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(d.bias + 4 * jown);
    float cst[NOWN], hst[NOWN];
    float fcst = 0.f, fhst = 0.f;                                        // the rider's cell (FK)
    int blen[NOWN], bidx[NOWN];
    float* outb[NOWN];                                                   // &out[t = 0][sequence][jown]
    const unsigned out_row_bytes = (unsigned)a.B * (unsigned)d.outStride * 4u;   // one time step of the layer output
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        const int b = brow0 + q * 4 + reg0 + o;
        bidx[o] = b;
        outb[o] = d.out + (size_t)(b < B ? b : 0) * d.outStride + jown;
        const bool inb = b < B;
        blen[o] = inb ? a.lengths[b] : 0;
        cst[o] = (inb && !a.zero_state) ? d.cbuf[(size_t)b * H + jown] : 0.f;
        hst[o] = (inb && !a.zero_state) ? d.hbuf[(size_t)b * H + jown] : 0.f;
    }

    // ---- A-operand row of this lane (row r16 of the slab): its sequence, length, x pointer pieces
    const int arow = brow0 + r16;
    const bool arow_in = arow < B;
    const int alen = arow_in ? a.lengths[arow] : 0;
    const float* xbase = d.xin + (size_t)(arow_in ? arow : 0) * KIN + kq * KQ + q * 4;
    const size_t xtstride = (size_t)B * KIN;

    // A operand of the recurrent part for step 0 from the initial state: h0[row r16][k = kq*KW + 4*ks + q]
    float av[NKS];
    {
        const float* p = d.hin + (size_t)(arow_in ? arow : 0) * H + kq * KW + q;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) av[ks] = (arow_in && !a.zero_state) ? p[4 * ks] : 0.f;
    }

    // granules of this slab: hx[dir][slab] = { L[2 parities][16*H], R[2 parities][16*H], xcc[8] }
    constexpr size_t SLABW = (size_t)4 * 16 * H + 16;
    u64* hxL = a.hx + (size_t)cl * SLABW;
    u64* hxR = hxL + (size_t)2 * 16 * H;
    u64* xtab = hxL + (size_t)4 * 16 * H;
    // TAGX (the H = 256 kernels): the same area holds plain 4-byte words h[parity][16 * H] for both transports: 4 x 16 bytes
    // per lane and step instead of 16 x 8 (measured on the granule form, profiles/r02_flagx.md: the 16 requests cost 700
    // cycles of issue inside the projection and their 16 compare / s_and pairs 600 cycles, while the words themselves were
    // ALWAYS there at the first look -- the hand-off latency was never the problem, its instruction count was).
    // Word (row, unit j) sits where consumer lane (q = j & 3, row) finds k-steps 4i .. 4i+3 of its K quarter in one
    // 16-byte piece:  (((j >> 6) * 4 + ((j & 63) >> 4)) * 64 + (j & 3) * 16 + row) * 4 + ((j >> 2) & 3).
    // Rounds 2-4 signalled these words with one flag per (parity, producer slice, producer wave) -- a chain of two dependent
    // L2 round trips per step, flag then values, behind the producer's store acknowledgement: ~3 400 cycles that the 16-slice
    // kernels' 2 300 cycles of projection MFMAs could not cover (profiles/r04_handoff.md; the flagged form was removed in round
    // 5 -- git has it).  Since round 4 every word carries a tag in bit 30 -- the top exponent bit, 0 for every |h| < 2, and
    // h = o * tanh(c) never leaves [-1, 1] -- that alternates with every write to its parity slot: the consumer requests the
    // 16-byte pieces ONCE, half-way through the projection, and at the end ORs all 16 (+4) words -- after an XOR that clears
    // the tag on the steps that expect a set one: bit 30 of the result is set exactly when a word was stale (then: bounded
    // re-request loop).  ~20 VALU instructions per step (they cost ~8 cycles each in a lone wave -- a first version with
    // twice as many gave the round trip's cycles straight back) instead of a round trip; the initial state goes out as the
    // words of "step -1", so step 0 is no special case; results are bit-identical to the flagged form, except that a
    // one-step launch on a carried state no longer races (there a fast workgroup could write its final state over the
    // initial state another one had yet to read as its step-0 operand -- each cell's initial h is now read by its owner only).
    // A NaN h loses its NaN-ness on the way (bit 30 is the tag): the NaN still reaches every output of its sequence through
    // the producing unit's own column of the layer output -- linear2 / the next layer sum over all columns -- but peers no
    // longer turn NaN through the recurrence (a poisoned wave's cells are NaN for all 16 rows of the slab: same effect).
    // Which tag a launch starts with: LstmPersistArgs::tag_flip.
    unsigned* hdL = reinterpret_cast<unsigned*>(hxL);                 // [2 parities][16 * H]
    constexpr unsigned HD_R = 2 * 16 * H * 4;                         // byte offset of the R copy of the values
    unsigned spin_budget = a.max_spin;
    // TAGX: tag of the h words written at `step` = -1 (the initial state, published before the loop: step 0 is a step like
    // any other), 0, 1, ...: the ((step + 1) / 2)-th write of this launch to parity slot step & 1
    auto tag_of = [&](int step) -> unsigned { return ((((unsigned)(step + 1) >> 1) ^ (a.tag_flip >> (step & 1))) & 1u) << 30; };
    // ... and the word itself: h with bit 30 replaced by the tag.  (An inactive row keeps publishing the h it was given -- a
    // caller's initial state may be anything: whatever its bit 30 was, the tag is what consumers find there.)
    auto hword_of = [&](float h, int step) -> unsigned {
        if (!TAGX) return __float_as_uint(h);
        return (__float_as_uint(h) & ~kHTagBit) | tag_of(step);
    };
    const long long tl_w = PROF ? (long long)__builtin_amdgcn_s_memrealtime() : 0;      // weight loads issued (not yet waited for)
    const unsigned my_xcc = xcc_id();
    bool src_local[NPW];                   // is the producer slice of each part of this wave's K quarter on my XCD?
#pragma unroll
    for (int i = 0; i < NPW; ++i) src_local[i] = true;
    bool all_local = true;
    bool link_local = true;                // WF: all eight slices of the partner cluster (the other layer of this slab) on my XCD
    unsigned long long same_xcd = ~0ull;   // bit s: producer slice s runs on my XCD
    if (NSLICE > 1) {
        // (epoch_base != 0: the exchange area is NOT zeroed between launches -- every launch uses tags nobody has written
        //  there before: base + step, and base itself for the XCC table; see LstmPersistArgs::epoch_base)
        const unsigned xtag = a.epoch_base ? a.epoch_base : XCC_TAG;
        if (threadIdx.x == 0) granule_store(xtab + slice, xtag, __uint_as_float(my_xcc));
        unsigned peer = my_xcc;
        // (WF: lanes NSLICE .. 2*NSLICE-1 read the table of the partner cluster, cl ^ 1)
        const u64* xlook = (WF && lane >= NSLICE) ? a.hx + (size_t)(cl ^ 1) * SLABW + (size_t)4 * 16 * H + (lane - NSLICE) : xtab + lane;
        if (lane < (WF ? 2 * NSLICE : NSLICE)) {
            unsigned spins = 0; u64 wt0 = 0;
            while (true) {
                const u64 g = granule_load(xlook);
                if ((unsigned)(g >> 32) == xtag) { peer = (unsigned)g; break; }
                if (wait_over(spins, spin_budget, wt0, a.max_ticks)) { mp_set_error(a.err, 1000000); peer = ~0u; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        const unsigned long long same = __ballot(peer == my_xcc);
        same_xcd = same;
        all_local = (same & ((1ull << NSLICE) - 1)) == ((1ull << NSLICE) - 1);
        if (WF) link_local = ((same >> NSLICE) & ((1ull << NSLICE) - 1)) == ((1ull << NSLICE) - 1);
#pragma unroll
        for (int i = 0; i < NPW; ++i) src_local[i] = (same >> (NPW * kq + i)) & 1;   // k-steps of part i come from slice NPW*kq+i
        if (__ballot(peer == ~0u)) { spin_budget = 0; poison_cells(cst); fcst = __builtin_nanf(""); }
        if (PROF && a.force_remote) {                       // test hook (PROF instantiation only): the any-placement transport
            all_local = false;
            link_local = false;
            same_xcd = 0;
#pragma unroll
            for (int i = 0; i < NPW; ++i) src_local[i] = false;
        }
    }
    // ---- WF: the link.  Progress words of layer 0 live in ITS cluster's area (cl & ~1), in the 1024 words between the h words
    // and the rider's: word [slice * 4 + wave].  A layer-1 wave watches the words of slices 2*kq and 2*kq + 1 -- all lanes load
    // (8 distinct addresses), so nothing about the check is per lane.
    unsigned* plink = reinterpret_cast<unsigned*>(a.hx + (size_t)(cl & ~1) * SLABW) + (size_t)4 * 16 * H;
    const unsigned* pwatch = plink + 8 * kq + (lane & 7);
    const unsigned pbase = a.epoch_base;
    unsigned pw = 0, link_waits = 0;
    const bool wf_l0 = WF && dir == 0, wf_l1 = WF && dir == 1;
    // wait until layer 0 has finished `need` steps (bounded like every wait; a wave that gives up poisons its cells)
    auto link_wait = [&](int need, int step) {
        unsigned spins = 0; u64 wt0 = 0;
        // (a word of THIS launch is pbase + n with need <= n <= T; anything else -- a zeroed word, an earlier launch's smaller
        //  value -- is far outside that window whatever pbase is: one unsigned compare, no wrap-around case.  A signed
        //  "pw - (pbase + need) >= 0" took the zeros of a freshly zeroed area for progress when pbase was above 2^31.)
        while (!__all(pw - pbase - (unsigned)need <= (unsigned)(T - need))) {
            if (PROF && spins == 0) ++link_waits;
            if (wait_over(spins, spin_budget, wt0, a.max_ticks)) {
                if (lane == 0) mp_set_error(a.err, 1 + step);
                spin_budget = 0; poison_cells(cst); fcst = __builtin_nanf("");
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            pw = __hip_atomic_load(pwatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto link_publish = [&](int done) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // this wave's output stores of the steps before are acknowledged
        if (lane == 0) {
            unsigned* f = plink + slice * 4 + wave;
            if (link_local) __hip_atomic_store(f, pbase + (unsigned)done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            else __hip_atomic_store(f, pbase + (unsigned)done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };

    // ---- TAGX: consumer offsets (piece i comes from producer slice NPW*kq + i/PPP), store slots
    __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hdL, 0, 4 * 16 * H * 4, 0x00020000);
    unsigned hvoff[NP];
    unsigned hslot[NOWN];
    f32x4 hr[NP];                                                      // recurrent A operand: k-steps 4i .. 4i+3 in hr[i]
    if (TAGX) {
#pragma unroll
        for (int i = 0; i < NP; ++i) hvoff[i] = (src_local[(i / PPP) % NPW] ? 0u : HD_R) + (unsigned)(((kq * NP + i) * 64 + lane) * 16);
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            const int row = q * 4 + reg0 + o;
            hslot[o] = (unsigned)(((((jown / KW) * NP + ((jown % KW) >> 4)) * 64 + (jown & 3) * 16 + row) << 2) + ((jown >> 2) & 3));
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) hr[i] = f32x4{av[(4 * i) % NKS], av[(4 * i + 1) % NKS], av[(4 * i + 2) % NKS], av[(4 * i + 3) % NKS]};
    }

    // ---- FK: the H = 64 rider.  Wave kq takes K quarter kq of both of its products: x k = kq*FK/4 + (i/4)*16 + q*4 + i%4
    // (x-step i < FNX), h k = kq*16 + 4*ks + q (ks < 4).  Exchange words of the rider: [transport][parity][direction][1024]
    // behind the link words of this cluster's area; word (row, unit u) at ((u/16*4 + u%4)*16 + row)*4 + (u/4)%4, so that consumer
    // lane (kq, q, row) finds its four k-steps in one 16-byte piece.  Producers of a piece: slices 8*dir + 2*kq (+1).
    static_assert(FK == 0 || (H == 256 && (NSLICE == 16 || NSLICE == 8) && KIN == 256), "the rider lives in the K_in = 256 kernels");
    constexpr int FNX = FK / 16, FNS = FNX + 4, FNJ = FK / 64;
    constexpr unsigned F_WORD0 = 4 * 16 * H + 1024;                  // first rider word of the cluster's area
    constexpr unsigned F_TR = 2 * 2 * 1024;                           // words per transport
    // rider direction, unit group, and which of the area's two rider blocks the words live in (16 slices: one cluster carries
    // both rider directions; 8 slices: the cluster's direction IS the rider's, block 0)
    const int fdir = NSLICE == 16 ? slice >> 3 : dir, fug = slice & 7, fblk = NSLICE == 16 ? fdir : 0;
    // TAGX tags of the rider's words: a bookkeeping of their own (LstmPersistArgs::tag_flip_f) -- a launch without a rider
    // leaves them alone
    auto ftag_of = [&](int step) -> unsigned { return ((((unsigned)(step + 1) >> 1) ^ (a.tag_flip_f >> (step & 1))) & 1u) << 30; };
    auto fword_of = [&](float h, int step) -> unsigned { return (__float_as_uint(h) & ~kHTagBit) | ftag_of(step); };
    float fw[FK ? FNS : 1][2];
    f32x4 fbias4 = f32x4{0.f, 0.f, 0.f, 0.f}, fxa[FNJ > 0 ? FNJ : 1], fhr = f32x4{0.f, 0.f, 0.f, 0.f};
    int flen = 0, frow = 0, funit = 0;
    unsigned fvoff = 0, fslot = 0;
    const float *fxp_cur = nullptr, *fxp_nxt = nullptr;
    size_t fxt = 0;
    __amdgpu_buffer_rsrc_t frs = __builtin_amdgcn_make_buffer_rsrc(hdL + (FK ? F_WORD0 : 0), 0, 2 * F_TR * 4, 0x00020000);
    if (FK) {
        const float* p = a.f_w[fdir] + ((size_t)(fug * 4 + kq) * FNS * 2) * 64 + lane;
#pragma unroll
        for (int st = 0; st < FNS; ++st) { fw[st][0] = p[(size_t)(st * 2) * 64]; fw[st][1] = p[(size_t)(st * 2 + 1) * 64]; }
        // the cell this lane finishes (lanes 0..31 of wave kq): row 4*kq + lane/8 of the slab, unit 8*fug + lane%8
        frow = brow0 + 4 * kq + ((lane >> 3) & 3);
        funit = 8 * fug + (lane & 7);
        flen = (lane < 32 && frow < B) ? a.lengths[frow] : 0;
        fbias4 = *reinterpret_cast<const f32x4*>(a.f_bias[fdir] + 4 * funit);
        fslot = (unsigned)((((funit >> 4) * 4 + (funit & 3)) * 16 + 4 * kq + ((lane >> 3) & 3)) * 4 + ((funit >> 2) & 3));
        const int fs0 = fblk * 8 + 2 * kq;                             // producer slices of this wave's piece
        const bool floc = ((same_xcd >> fs0) & 1) && ((same_xcd >> (fs0 + 1)) & 1);
        fvoff = (floc ? 0u : F_TR * 4u) + (unsigned)fblk * 4096u + (unsigned)(((kq * 4 + q) * 16 + r16) * 16);
        fxt = (size_t)B * FK;
        fxp_cur = a.f_xin + (size_t)(arow_in ? arow : 0) * FK + kq * (FK / 4) + q * 4 + (size_t)(fdir && alen > 0 ? alen - 1 : 0) * fxt;
        fxp_nxt = fxp_cur;
#pragma unroll
        for (int j = 0; j < FNJ; ++j) fxa[j] = *reinterpret_cast<const f32x4*>(fxp_cur + j * 16);
    }
    // rider partials [kq][tile][lane] behind the reduction scratch (ALLREG: two copies of both, used in turn -- RED_DB below)
    f32x4* fred0 = reinterpret_cast<f32x4*>(smem) + C::RED_F4 * (C::ALLREG ? 2 : 1) + (size_t)NWV * XL * NTG * 64;
    if (TAGX && NSLICE > 1 && !all_local) {
        // The R copies (any-placement transport) are written only by launches whose cluster does NOT share an XCD, the host's
        // tag bookkeeping (tag_flip) advances with every launch: what an earlier launch left in an R copy may carry exactly the
        // tag this launch expects (round 5: found by toggling the transport test hook between calls -- a consumer that looked
        // early took a stale word; placement itself is stable from launch to launch, so the product never got here).  So a
        // cluster on the R transport first makes its R words INVALID for this launch -- every owner writes the complement of
        // the tag its first write to that slot will carry -- and meets once more (second table, in the gap behind the h words)
        // before anybody publishes or reads: whoever has seen all peers' second entries knows every R word is either this
        // launch's or recognisably not.  all_local is the same for all workgroups of a cluster (one workgroup elsewhere makes
        // it false for everybody), so they all come here or none does.  Nothing of this runs on the L transport.
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            __hip_atomic_store(hdL + HD_R / 4 + hslot[o], tag_of(0) ^ kHTagBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(hdL + (size_t)16 * H + HD_R / 4 + hslot[o], tag_of(-1) ^ kHTagBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (FK && lane < 32) {
            unsigned* fwd = hdL + F_WORD0 + F_TR + (unsigned)fblk * 1024u + fslot;
            __hip_atomic_store(fwd, ftag_of(0) ^ kHTagBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(fwd + 2048u, ftag_of(-1) ^ kHTagBit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                       // all four waves' words are out
        u64* xtab2 = hxL + (size_t)2 * 16 * H + 64;            // words 16512 .. of the area (16 entries)
        const unsigned xtag = a.epoch_base ? a.epoch_base : XCC_TAG;
        if (threadIdx.x == 0) granule_store(xtab2 + slice, xtag, 1.0f);
        if (lane < NSLICE) {
            unsigned spins = 0; u64 wt0 = 0;
            while ((unsigned)(granule_load(xtab2 + lane) >> 32) != xtag) {
                if (wait_over(spins, spin_budget, wt0, a.max_ticks)) { mp_set_error(a.err, 1000000); spin_budget = 0; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        if (__ballot(spin_budget == 0)) { spin_budget = 0; poison_cells(cst); fcst = __builtin_nanf(""); }
    }
    if (TAGX) {
        // the initial state goes out as the words of "step -1" (parity slot 1): step 0 requests and validates its recurrent
        // operand like every other step -- no `step > 0` around the request, the check or the operand (each was a branch whose
        // merge cost register copies on every step)
        unsigned* hw = hdL + (size_t)16 * H;
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            // (an initial h with bit 30 set cannot travel as a tagged word.  Finite |h| >= 2 / inf -- which no LSTM produces: say so
            //  (state code 2000000, error word [1]) and poison the cell; with recovery on the call is run again by the per-step
            //  kernels, which take any state.  NaN -- what the reference carries forward in velocity.rnn_state after ONE NaN sample
            //  (velocity.py:45-48 keeps the state, golden G16) -- is no error: it is treated like a NaN that turns up in the middle
            //  of a sequence: the cell is poisoned (its h, and with it its column of the layer output, is NaN for every step, so
            //  every output of that sequence is) and peers get a finite word with a clean tag that only reaches that sequence's
            //  own gates.  Round 4 raised the code for NaN too, which put every later call of a handle with one glitched stream
            //  on the fail-then-recover path, per-step kernels and a warning per call.)
            if (blen[o] > 0 && (__float_as_uint(hst[o]) & kHTagBit)) {
                if (hst[o] == hst[o]) mp_set_error(a.err + 1, 2000000);
                cst[o] = __builtin_nanf("");
            }
            const unsigned w0 = hword_of(hst[o], -1);
            __hip_atomic_store(hw + hslot[o], w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!all_local) __hip_atomic_store(hw + HD_R / 4 + hslot[o], w0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (FK && lane < 32) {
            unsigned* fwd = hdL + F_WORD0 + 2048u + (unsigned)fblk * 1024u + fslot;
            __hip_atomic_store(fwd, fword_of(0.f, -1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!all_local) __hip_atomic_store(fwd + F_TR, fword_of(0.f, -1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }

    // ---- x_0: this lane's A values of the input projection, k = kq*KQ + j*16 + q*4 + i  (x-step s = 4j+i)
    f32x4 xa[NXJ];
    // With part of W_ih in registers (XR > 0) there is no room for a whole prefetched x row beside the
    // granules: the second half of x_t is then fetched at the top of step t (it is first used ~2000 cycles later).
    constexpr bool SPLIT_X = C::BIG && H == 256;            // (H = 64: registers to spare -- the whole row one step ahead)
    constexpr int XJ_PRE = SPLIT_X ? NXJ / 2 : NXJ;        // 16-byte pieces prefetched one step ahead
    // LEAN (one wave per SIMD -- every configuration since round 5): VALU instructions do not hide under
    // MFMAs on this hardware (profiles/r02_persist_phases.md) and a lone wave has nobody to cover them, so the step is put on
    // a VALU diet -- (1) x_t of this lane's row is read UNCONDITIONALLY from a clamped time index (a row past its length
    // multiplies whatever finite values it finds there: row r of the A operand only reaches row r of the gates, and an
    // inactive row's gates are discarded by the cell update; the zero fill cost 32 v_mov + exec juggling per step) through
    // a pointer that is stepped, not re-multiplied: forward t = min(step, T-1), reverse t = max(len-1-step, 0); (2) the
    // granules are requested on every step and feed the recurrent MFMAs straight from their low words (a load under a
    // branch made `gr` a phi of defined / undefined values: 60-80 v_mov per step).  Measured, same box: velocity layers
    // 577 -> 504-513 us, foot-contact layers 426-440 -> 408-414 us.
    constexpr bool LEAN = true;
    const float* xp_cur = xbase + (size_t)(d.reverse ? (alen > 0 ? alen - 1 : 0) : 0) * xtstride;   // time index of `step`
    const float* xp_nxt = xp_cur;                                                                  // ... of `step + 1`
    // WF: x through a buffer resource over the whole [T][B][K_in] array with sc1 loads (the host takes the wavefront only when
    // the array is smaller than 4 GB); byte offsets instead of pointers, both layers forward in time
    __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.xin), 0,
                                                                   WF ? (int)((size_t)T * xtstride * 4) : 0, 0x00020000);
    unsigned xo_cur = (unsigned)(((size_t)(arow_in ? arow : 0) * KIN + kq * KQ + q * 4) * 4), xo_nxt = xo_cur;
    auto load_x = [&](int step, int j0, int j1) {
        if (WF) {
            const unsigned o = step & 0x40000000 ? xo_nxt : xo_cur;
#pragma unroll
            for (int j = 0; j < NXJ; ++j)
                if (j >= j0 && j < j1) xa[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, o, j * 64, 16 /* sc1 */));
        } else if (LEAN) {
            const float* p = step & 0x40000000 ? xp_nxt : xp_cur;       // (bit 30 = "the prefetch of the next step")
#pragma unroll
            for (int j = 0; j < NXJ; ++j)
                if (j >= j0 && j < j1) xa[j] = *reinterpret_cast<const f32x4*>(p + j * 16);
        } else {
            step &= 0x3fffffff;
            const bool on = step < alen;
            const int t = on ? (d.reverse ? alen - 1 - step : step) : 0;
            const float* p = xbase + (size_t)t * xtstride;
#pragma unroll
            for (int j = 0; j < NXJ; ++j)
                if (j >= j0 && j < j1) xa[j] = on ? *reinterpret_cast<const f32x4*>(p + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };
    if (wf_l1) {                                              // x_0 of layer 1 = layer 0's first output
        pw = __hip_atomic_load(pwatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        link_wait(1, 0);
    }
    load_x(0, 0, XJ_PRE);
    __syncthreads();                                          // W_ih LDS image complete
    // (x_0 has arrived before the loop is entered, x_{t+1} before the stores of step t are issued -- see the cell update --
    //  so the compiler places no vector-memory wait at the top of the loop, where the previous step's stores are pending)
#pragma unroll
    for (int j = 0; j < XJ_PRE; ++j) asm volatile("" : "+v"(xa[j]));

    long long pt[6] = {0, 0, 0, 0, 0, 0};
    const long long tl_loop = PROF ? (long long)__builtin_amdgcn_s_memrealtime() : 0;   // first step starts
    const bool prof = PROF && a.prof != nullptr && threadIdx.x == 0;
    const long long t_start = PROF ? (long long)__builtin_amdgcn_s_memtime() : 0;
#define PROF_T(i) do { if (PROF && prof) pt[i] -= (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define PROF_E(i) do { if (PROF && prof) pt[i] += (long long)__builtin_amdgcn_s_memtime(); } while (0)

    const f32x4* wxw = wxl + (size_t)wave * XL * NTG * 64 + lane;

    for (int step = 0; step < T; ++step) {
        PROF_T(0);
        if (LEAN) xp_cur = xp_nxt;
        if (WF) xo_cur = xo_nxt;
        if (wf_l1) pw = __hip_atomic_load(pwatch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // looked at before the prefetch of x_{t+1}
        if (FK) fxp_cur = fxp_nxt;
        if (SPLIT_X && !TAGX) load_x(step, XJ_PRE, NXJ);
        f32x4 acc[NTW];
        if (!WREG) {                                           // (WREG: the first MFMA of every tile has C = 0)
#pragma unroll
            for (int t = 0; t < NTW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // ---- first half of x_t W_ih^T (independent of h: this is what fills the wait for the peers)
        // (LDS-resident weights are fetched exactly one k-step ahead; the scheduling barriers keep the compiler
        //  from hoisting dozens of ds_reads -- and their 4 destination registers each -- to the top of the loop)
        f32x4 wl[NTG], wn[NTG];
#pragma unroll
        for (int tg = 0; tg < NTG; ++tg) wl[tg] = XL > 0 ? wxw[(size_t)tg * 64] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < XSPLIT; ++s) {
            const float a_s = xa[s >> 2][s & 3];
            if (s + 1 < XL) {
#pragma unroll
                for (int tg = 0; tg < NTG; ++tg) wn[tg] = wxw[(size_t)((s + 1) * NTG + tg) * 64];
            }
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                const f32x4 w4 = s < XL ? wl[tg] : wxr[s >= XL ? s - XL : 0][tg];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (WREG) {
                        if (s == 0) { if (s < XL) mfma_asm<true, false>(acc[tg * 4 + i], a_s, w4[i]); else mfma_asm<true, true>(acc[tg * 4 + i], a_s, w4[i]); }
                        else { if (s < XL) mfma_asm<false, false>(acc[tg * 4 + i], a_s, w4[i]); else mfma_asm<false, true>(acc[tg * 4 + i], a_s, w4[i]); }
                    } else {
                        acc[tg * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_s, w4[i], acc[tg * 4 + i], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) wl[tg] = wn[tg];
            if (TAGX) {
                constexpr int PUB_S = 1;
                if (s == PUB_S) {
                    if (SPLIT_X) load_x(step, XJ_PRE, NXJ);          // (second half of a K_in = 512 row: first used 14 k-steps on)
                }
                // WF, layer 0: steps 0 .. step-1 are in the output buffer -- said as late as the projection allows (the request
                // for h behind k-step XSPLIT - 1 must not be in flight when vmcnt(0) is waited for): the stores of the last step
                // have had 6 k-steps = 1 500 cycles to be acknowledged, and layer 1 is a step or two behind anyway
                if (WF && s == XSPLIT - 2) { if (wf_l0 && step > 0) link_publish(step); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- FK: the rider's input projection (independent of h: it lengthens the window that hides the hand-off)
        // (WREG: the rider's MFMAs are inline asm with VGPR operands like the carrier's.  As builtins the compiler put `facc`
        //  into AccVGPRs -- all 256 of which hold the carrier's weights -- and moved the displaced weights through ONE AccVGPR with
        //  a v_accvgpr_write right in front of each asm MFMA that reads it: a VALU-write -> MFMA-read hazard the hazard recogniser
        //  cannot see inside asm.  First GPU run of round 5: pose and velocity off by 5e-3, the rider itself right.  The two
        //  accumulators alternate, so dependent MFMAs are 64 cycles apart; mfma_drain() below covers the reads of the results.)
        f32x4 facc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        auto rider_xproj = [&]() {
#pragma unroll
            for (int i = 0; i < FNX; ++i) {
                const float a_s = fxa[(i >> 2) % (FNJ > 0 ? FNJ : 1)][i & 3];
                if (WREG) {
                    if (i == 0) { mfma_asm<true, false>(facc[0], a_s, fw[i % FNS][0]); mfma_asm<true, false>(facc[1], a_s, fw[i % FNS][1]); }
                    else { mfma_asm<false, false>(facc[0], a_s, fw[i % FNS][0]); mfma_asm<false, false>(facc[1], a_s, fw[i % FNS][1]); }
                } else {
                    facc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_s, fw[i % FNS][0], facc[0], 0, 0, 0);
                    facc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_s, fw[i % FNS][1], facc[1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        };
        // (the rider's projection sits behind the request below -- it is part of what hides the round trip)

        // ---- request h_{step-1}: granule (row r16, unit kq*KW + 4*ks + q), 512 contiguous bytes per instruction
        u64 gr[NKS];
        const unsigned epoch = a.epoch_base + (unsigned)step;  // written by the producers at the end of step-1
        const size_t goff = (size_t)((step + 1) & 1) * 16 * H + (size_t)kq * NKS * 64 + r16 * 4 + q;
        const u64* srcp[NPW];
#pragma unroll
        for (int i = 0; i < NPW; ++i) srcp[i] = (src_local[i] ? hxL : hxR) + goff;
        constexpr int KSP = NKS / NPW;                          // k-steps per producer slice
        // (when part of W_ih lives in registers there is no room to hold 16 granules in flight beside it:
        //  request them after the projection instead; the second wave on the SIMD covers the L2 latency)
        constexpr bool EARLY_GATHER = !C::BIG || WREG;         // (WREG: 512 registers per wave -- room for the granules)
        // (LEAN: requested on EVERY step, step 0 included -- there the words are simply not looked at)
        if (TAGX) {
            // every piece of x_t has been requested long ago (the previous step's prefetch; the second half of a K_in = 512
            // row 14 k-steps ago): have the compiler wait for them HERE -- its own placement, piece by piece inside the rest
            // of the projection with counts that assume nothing younger is in flight, would also wait for the h words
            // requested just below (measured in the K_in = 256 kernel: a full L2 round trip per step)
#pragma unroll
            for (int j = 0; j < NXJ; ++j) asm volatile("" : "+v"(xa[j]));
            {
                const int par_off = ((step + 1) & 1) * (16 * H * 4);
#pragma unroll
                for (int i = 0; i < NP; ++i)
                    hr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, hvoff[i], par_off, 16 /* sc1 */));
                if (FK) fhr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frs, fvoff, ((step + 1) & 1) * 8192, 16 /* sc1 */));
            }
        } else if (EARLY_GATHER && (LEAN || step > 0)) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) gr[ks] = granule_load(srcp[ks / KSP] + (size_t)ks * 64);
        }
        if (FK) rider_xproj();
        // ---- second half of the input projection
#pragma unroll
        for (int s = XSPLIT; s < NXS; ++s) {
            const float a_s = xa[s >> 2][s & 3];
            if (s + 1 < XL) {
#pragma unroll
                for (int tg = 0; tg < NTG; ++tg) wn[tg] = wxw[(size_t)((s + 1) * NTG + tg) * 64];
            }
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) {
                const f32x4 w4 = s < XL ? wl[tg] : wxr[s >= XL ? s - XL : 0][tg];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (WREG) {
                        if (s == 0) { if (s < XL) mfma_asm<true, false>(acc[tg * 4 + i], a_s, w4[i]); else mfma_asm<true, true>(acc[tg * 4 + i], a_s, w4[i]); }
                        else { if (s < XL) mfma_asm<false, false>(acc[tg * 4 + i], a_s, w4[i]); else mfma_asm<false, true>(acc[tg * 4 + i], a_s, w4[i]); }
                    } else {
                        acc[tg * 4 + i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_s, w4[i], acc[tg * 4 + i], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tg = 0; tg < NTG; ++tg) wl[tg] = wn[tg];
        }
        PROF_E(0); PROF_T(1);

        // ---- validate the granules; the slow path (cheap gate, then sweep) only runs when some were stale
        // (K_in = 512: the registers are full -- the words are requested only now and only when they are needed, and the
        //  recurrent A operand is extracted into 16 registers so that the 32 of `gr` die before the MFMAs)
        constexpr bool DIRECT_GR = EARLY_GATHER && LEAN && !TAGX;    // recurrent MFMAs read the low words of `gr` directly
        if (TAGX) {
            // the values were requested half a projection ago: make the compiler wait for them HERE, before the prefetch
            // of x_{t+1} below is issued (a wait placed after it would also drain those HBM loads)
#pragma unroll
            for (int i = 0; i < NP; ++i) asm volatile("" : "+v"(hr[i]));
            if (FK) asm volatile("" : "+v"(fhr));
            {
                // clear the expected tag, OR what is left: bit 30 set = some word is not (yet) the one of step - 1.  ONE copy of
                // this code, in a loop whose body normally runs once (the re-request at its bottom writes the same registers:
                // no phi copies -- the first version, with a separate slow path, cost 30-40 v_mov per step)
                const unsigned etag = tag_of(step - 1), fetag = FK ? ftag_of(step - 1) : 0u;
                const int par_off = ((step + 1) & 1) * (16 * H * 4);
                unsigned spins = 0; u64 wt0 = 0;
                while (true) {
                    // (the XOR only when there is a tag to clear -- a scalar branch around 20 VALU instructions, which cost
                    //  ~8 cycles each in a lone wave; in place, so nothing to merge behind it)
                    if (etag) {
#pragma unroll
                        for (int i = 0; i < NP; ++i)
#pragma unroll
                            for (int c = 0; c < 4; ++c) hr[i][c] = __uint_as_float(__float_as_uint(hr[i][c]) ^ kHTagBit);
                    }
                    if (FK && fetag) {                     // (the rider's words: tags of their own, LstmPersistArgs::tag_flip_f)
#pragma unroll
                        for (int c = 0; c < 4; ++c) fhr[c] = __uint_as_float(__float_as_uint(fhr[c]) ^ kHTagBit);
                    }
                    unsigned left = 0;
#pragma unroll
                    for (int i = 0; i < NP; ++i)
#pragma unroll
                        for (int c = 0; c < 4; ++c) left |= __float_as_uint(hr[i][c]);
                    if (FK) {
#pragma unroll
                        for (int c = 0; c < 4; ++c) left |= __float_as_uint(fhr[c]);
                    }
                    if (__all((left & kHTagBit) == 0u)) break;
                    if (PROF && prof && spins == 0) pt[5] += 1;                          // slow-path entries
                    if (wait_over(spins, spin_budget, wt0, a.max_ticks)) {               // bounded: flag the error and never wait again
                        if (lane == 0) mp_set_error(a.err, 1 + step);
                        spin_budget = 0; poison_cells(cst); fcst = __builtin_nanf("");
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int i = 0; i < NP; ++i)
                        hr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, hvoff[i], par_off, 16 /* sc1 */));
                    if (FK) fhr = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(frs, fvoff, ((step + 1) & 1) * 8192, 16 /* sc1 */));
                }
            }
        } else if (step > 0) {
            if (!EARLY_GATHER) {
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) gr[ks] = granule_load(srcp[ks / KSP] + (size_t)ks * 64);
            }
            bool ok = true;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) ok = ok && ((unsigned)(gr[ks] >> 32) == epoch);
            unsigned spins = 0; u64 wt0 = 0;
            bool timed_out = false;
            if (PROF && prof && !__all(ok)) pt[5] += 1;      // slow-path entries
            while (!__all(ok) && !timed_out) {
                // gate: NPW lanes per wave watch ONE granule of each producer (polling with everything floods the
                // fabric with sc1 loads and slows every hand-off on the chip: MI355X_MICROARCH "polling-cost")
                while (true) {
                    bool ready = true;
#pragma unroll
                    for (int i = 0; i < NPW; ++i)
                        if (lane == i) ready = (unsigned)(granule_load(srcp[i] + (size_t)(i * KSP) * 64) >> 32) == epoch;
                    if (__all(ready)) break;
                    if (wait_over(spins, spin_budget, wt0, a.max_ticks)) { timed_out = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                ok = true;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    gr[ks] = granule_load(srcp[ks / KSP] + (size_t)ks * 64);
                    ok = ok && ((unsigned)(gr[ks] >> 32) == epoch);
                }
                if (wait_over(spins, spin_budget, wt0, a.max_ticks)) timed_out = true;
            }
            if (timed_out) {                                   // bounded: flag the error and never wait again
                if (lane == 0) mp_set_error(a.err, 1 + step);
                spin_budget = 0; poison_cells(cst); fcst = __builtin_nanf("");
            }
            if (!DIRECT_GR) {
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) av[ks] = __uint_as_float((unsigned)gr[ks]);
            }
        } else if (DIRECT_GR) {
            // step 0: the recurrent A operand is the initial state; it takes the place of the (unused) words just loaded
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) gr[ks] = (u64)__float_as_uint(av[ks]);
        }
        // WF, layer 1: is layer 0's output of step t + 1 complete?  Looked at HERE, right behind the wait for the h words -- every
        // older load, the progress words among them, has arrived, so the compiler's vmcnt(0) in front of the compare costs
        // nothing.  (First version: behind the rider's x loads of step t + 1 -- the same vmcnt(0) then sat out their whole
        // latency, 760 cycles per step in layer 1, profiles/r05_wavefront.md.)
        if (wf_l1) link_wait(step + 2 < T ? step + 2 : T, step);
        if (LEAN) {   // time index of step + 1, clamped
            const bool adv = d.reverse ? (alen - 2 - step >= 0) : (step + 1 < T);
            const long dlt = d.reverse ? -(long)xtstride : (long)xtstride;
            xp_nxt = adv ? xp_cur + dlt : xp_cur;
            if (WF) xo_nxt = step + 1 < T ? xo_cur + (unsigned)(xtstride * 4) : xo_cur;
        }
        if (FK) {      // the rider's x of step + 1 (forward: t = step + 1, reverse: t = len - 2 - step, both clamped)
            // (one signed stride, as for the layer's own x above: the direction is uniform but the compiler branched on it)
            const bool adv = fdir ? (alen - 2 - step >= 0) : (step + 1 < T);
            const long fdlt = fdir ? -(long)fxt : (long)fxt;
            fxp_nxt = adv ? fxp_cur + fdlt : fxp_cur;
#pragma unroll
            for (int j = 0; j < FNJ; ++j) fxa[j] = *reinterpret_cast<const f32x4*>(fxp_nxt + j * 16);
        }
        load_x((step + 1) | 0x40000000, 0, XJ_PRE);   // next step's x: issued only now so that the granule wait above does not
                                                      // also drain these HBM loads; they land under the MFMAs / cell update below
        PROF_E(1); PROF_T(2);

        // ---- recurrent part: h_{t-1} W_hh^T on top of the input projection
        // RED3 (the four-wave 8-slice K_in = 256 kernel, round 4): the K reduction's LDS writes go out UNDER the last recurrent MFMAs.
        // The last four k-steps run pair of tiles by pair of tiles (gate p of both unit blocks: two independent accumulators
        // alternate, 64 cycles between dependent MFMAs), and while pair p + 1 is in the matrix pipe the finished tiles of pair p
        // are stored as they are -- registers (0, 1) and (2, 3) as two 8-byte pieces, one ds_write2st64_b64 per tile, layout
        // [tile][source kq][half][lane]; LDS instructions issue beside a wave's MFMAs (VALU ones do not, so the old form's 32
        // v_mov that gathered {i, f, g, o} pieces could not have moved there).  The barrier that protects the previous step's
        // reads of `red` moves in front of the recurrent part.  A finishing wave then reads its half of the four gate tiles of its
        // unit block from the four sources (8 ds_read2st64_b64).  Same bytes, same summation order: bit-identical.  Measured
        // (profiles/r04_handoff.md): reduction phase 725 -> 390 cycles, recurrent phase 4 252 -> 4 440 (the writes are not free),
        // launch 551 -> 546 us; in the K_in = 512 kernel (LDS full: one copy of the scratch, the barrier stays) the same change
        // moved the phases but not the launch time (783.7 vs 783.5 us), so it keeps the old form.
        constexpr bool RED3 = WREG && C::ALLREG && !PER_UB && NUB == 2 && NOWN == 2 && NKS >= 8 && TAGX;
        // (K_in = 256: all weights in registers, LDS to spare -- two copies of the scratch, used in turn, and the barrier that
        //  protects the previous step's reads is not needed at all: whoever writes copy s & 1 has passed the barrier of step
        //  s - 1, which every wave reaches only after its reads of step s - 2)
        constexpr bool RED_DB = RED3 && C::ALLREG;
        float2* red2 = reinterpret_cast<float2*>(smem) + (RED_DB ? (size_t)(step & 1) * C::RED_F4 * 2 : 0);
        f32x4* fred = fred0 + (RED_DB ? (size_t)(step & 1) * (4 * 2 * 64) : 0);
        if (RED3 && !RED_DB) barrier_lds_only();               // previous step's reads of `red` are done
        if (WREG) asm volatile("s_nop 3" ::: "memory");        // (step 0 writes the A operand with VALU moves just above)
#pragma unroll
        for (int ks = 0; ks < (RED3 ? NKS - 4 : NKS); ++ks)
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const float a_h = TAGX ? hr[(ks >> 2) % NP][ks & 3] : DIRECT_GR ? __uint_as_float((unsigned)gr[ks]) : av[ks];
                if (WREG) mfma_asm<false, true>(acc[t], a_h, wv[ks][t]);
                else acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_h, wv[ks][t], acc[t], 0, 0, 0);
            }
        if (RED3) {
#pragma unroll
            for (int p = 0; p < NTW / 2; ++p) {
#pragma unroll
                for (int ks = NKS - 4; ks < NKS; ++ks) {
#pragma unroll
                    for (int t = 2 * p; t < 2 * p + 2; ++t) mfma_asm<false, true>(acc[t % NTW], hr[(ks >> 2) % NP][ks & 3], wv[ks][t % NTW]);
                    if (p > 0 && ks == NKS - 3) {
                        // (the tiles of pair p - 1: their last MFMAs were issued >= 4 MFMAs = 128 cycles ago)
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int t = 2 * p - 2; t < 2 * p; ++t)
#pragma unroll
                            for (int hf = 0; hf < 2; ++hf)
                                red2[(((t % NTW) * 4 + kq) * 2 + hf) * 64 + lane] = float2{acc[t % NTW][2 * hf], acc[t % NTW][2 * hf + 1]};
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
        if (FK) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                if (WREG) {
                    mfma_asm<false, false>(facc[0], fhr[ks], fw[(FNX + ks) % FNS][0]);
                    mfma_asm<false, false>(facc[1], fhr[ks], fw[(FNX + ks) % FNS][1]);
                } else {
                    facc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(fhr[ks], fw[(FNX + ks) % FNS][0], facc[0], 0, 0, 0);
                    facc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(fhr[ks], fw[(FNX + ks) % FNS][1], facc[1], 0, 0, 0);
                }
            }
        }
        if (WREG) mfma_drain();
        PROF_E(2); PROF_T(3);

        // ---- K reduction through LDS: the 4 K-quarter waves of a tile group hand each finishing wave the 4 gate
        // values of the accumulator regs it finishes (own share included: register indices stay compile-time)
        if (!RED3) __syncthreads();                            // previous step's reads of `red` are done
        if (FK) {      // rider partials, transposed for the finishing lanes: [source kq][row 4q+reg][column][tile]
            float2* ft = reinterpret_cast<float2*>(fred) + (size_t)(kq * 16 + 4 * q) * 16 + r16;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) ft[rg * 16] = float2{facc[0][rg], facc[1][rg]};
        }
        if (RED3) {                                            // (the last pair of tiles; the others went out above)
#pragma unroll
            for (int t = NTW - 2; t < NTW; ++t)
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
                    red2[((t * 4 + kq) * 2 + hf) * 64 + lane] = float2{acc[t][2 * hf], acc[t][2 * hf + 1]};
        } else if (!PER_UB) {
#pragma unroll
            for (int dw = 0; dw < 4; ++dw) {
                const int dub = dw & 1;
                const int dreg0 = 2 * (dw >> 1);
#pragma unroll
                for (int o = 0; o < NOWN; ++o)
                    red[((dw * 4 + kq) * NOWN + o) * 64 + lane] =
                        f32x4{acc[(0 * NUB + dub) % NTW][dreg0 + o], acc[(1 * NUB + dub) % NTW][dreg0 + o],
                              acc[(2 * NUB + dub) % NTW][dreg0 + o], acc[(3 * NUB + dub) % NTW][dreg0 + o]};
            }
        } else {
#pragma unroll
            for (int dk = 0; dk < 4; ++dk)                     // finishing wave dk takes accumulator reg dk
                red[((dk * 4 + kq)) * 64 + lane] =
                    f32x4{acc[0][dk], acc[1 % NTW][dk], acc[2 % NTW][dk], acc[3 % NTW][dk]};
        }
        __syncthreads();
        f32x4 gate[NOWN];
        if (RED3) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int t = g * NUB + (wave & 1);
                float2 v = red2[((t * 4 + 0) * 2 + (wave >> 1)) * 64 + lane];
#pragma unroll
                for (int sw = 1; sw < 4; ++sw) {
                    const float2 u = red2[((t * 4 + sw) * 2 + (wave >> 1)) * 64 + lane];
                    v.x += u.x; v.y += u.y;
                }
                gate[0][g] = v.x + bias4[g];
                gate[NOWN - 1][g] = v.y + bias4[g];
            }
        } else
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            f32x4 v = red[((wave * 4 + 0) * NOWN + o) * 64 + lane];
#pragma unroll
            for (int sw = 1; sw < 4; ++sw) v += red[((wave * 4 + sw) * NOWN + o) * 64 + lane];
            gate[o] = v + bias4;
        }
        // rider gates of cell (row 4*kq + lane/8, unit u = lane%8): (i, g) = column u, (f, o) = column 8 + u of tiles (0, 1)
        float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f;
        if (FK) {
            const float2* ft = reinterpret_cast<const float2*>(fred) + (size_t)(4 * kq + ((lane >> 3) & 3)) * 16 + (lane & 7);
#pragma unroll
            for (int ksrc = 0; ksrc < 4; ++ksrc) {
                const float2 ig2 = ft[ksrc * 256], fo2 = ft[ksrc * 256 + 8];
                gi += ig2.x; gg += ig2.y; gf += fo2.x; go += fo2.y;
            }
            gi += fbias4[0]; gf += fbias4[1]; gg += fbias4[2]; go += fbias4[3];
        }
        PROF_E(3); PROF_T(4);

        // ---- cell update (register-local), publish h_step, write the layer output
        // (x_{t+1}, requested before the recurrent MFMAs, is waited for HERE, while only loads are in flight: vmcnt counts
        //  loads and stores together and stores are acknowledged out of order, so once the stores below are pending the only
        //  safe wait for a load is vmcnt(0) -- the top of the next step would sit out the acknowledgement of every store)
#pragma unroll
        for (int j = 0; j < XJ_PRE; ++j) asm volatile("" : "+v"(xa[j]));
        const size_t doff = (size_t)(step & 1) * 16 * H;
        if (WREG) {
            // two cells per lane: both are computed first, branch-free (an inactive row's gates are whatever its clamped input
            // row gives -- finite or not, they are never kept), and all stores follow -- one basic block, so the scheduler
            // interleaves the two dependent exp / rcp chains instead of running them one after the other
            float oval[NOWN], cnew[NOWN], hnew[NOWN];
            int tt[NOWN];
#pragma unroll
            for (int o = 0; o < NOWN; ++o) {
                const float ig = sigmoidf_(gate[o][0]);
                const float fg = sigmoidf_(gate[o][1]);
                const float gv = tanhf_(gate[o][2]);
                const float og = sigmoidf_(gate[o][3]);
                cnew[o] = fg * cst[o] + ig * gv;
                hnew[o] = og * tanhf_(cnew[o]);
            }
            // (FK: the rider's cell -- row 4*kq + lane/8, unit lane%8 on lanes 0..31 -- in the same block: a third chain to interleave)
            float fcnew = 0.f, fhnew = 0.f;
            if (FK) {
                const float fig = sigmoidf_(gi), ffg = sigmoidf_(gf), fgv = tanhf_(gg), fog = sigmoidf_(go);
                fcnew = ffg * fcst + fig * fgv;
                fhnew = fog * tanhf_(fcnew);
            }
            // Every chain's results are USED here, whether their row is active or not: without this the compiler sinks each chain
            // into an exec-masked region of its own (`act ? cnew : cst` -- why compute what an inactive lane drops) and the two or
            // three dependent exp / rcp chains run one after the other instead of interleaved (the ISA of rounds 3-4 had exactly
            // that: two s_and_saveexec blocks back to back, although the source was written branch-free)
            if (NOWN == 2) asm volatile("" : "+v"(cnew[0]), "+v"(hnew[0]), "+v"(cnew[NOWN - 1]), "+v"(hnew[NOWN - 1]));
            else asm volatile("" : "+v"(cnew[0]), "+v"(hnew[0]));
            if (FK) asm volatile("" : "+v"(fcnew), "+v"(fhnew));
#pragma unroll
            for (int o = 0; o < NOWN; ++o) {
                const bool act = step < blen[o];
                tt[o] = act ? (d.reverse ? blen[o] - 1 - step : step) : step;
                cst[o] = act ? cnew[o] : cst[o];
                hst[o] = act ? hnew[o] : hst[o];
                oval[o] = act ? hnew[o] : 0.f;
            }
            const bool fact = FK && step < flen;                 // (flen = 0 on lanes 32..63 and for rows past the batch)
            const int ftt = fact ? (fdir ? flen - 1 - step : step) : step;
            if (FK) {
                fcst = fact ? fcnew : fcst;
                fhst = fact ? fhnew : fhst;
            }
            unsigned* hw = hdL + (size_t)(step & 1) * 16 * H;
            unsigned hword[NOWN];
#pragma unroll
            for (int o = 0; o < NOWN; ++o) hword[o] = hword_of(hst[o], step);
#pragma unroll
            for (int o = 0; o < NOWN; ++o)
                __hip_atomic_store(hw + hslot[o], hword[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!all_local) {
#pragma unroll
                for (int o = 0; o < NOWN; ++o)
                    __hip_atomic_store(hw + HD_R / 4 + hslot[o], hword[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (FK && lane < 32) {
                const unsigned fhword = fword_of(fhst, step);
                unsigned* fwd = hdL + F_WORD0 + (unsigned)(step & 1) * 2048u + (unsigned)fblk * 1024u + fslot;
                __hip_atomic_store(fwd, fhword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!all_local) __hip_atomic_store(fwd + F_TR, fhword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (frow < B) a.f_out[((size_t)ftt * B + frow) * 128 + fdir * 64 + funit] = fact ? fhnew : 0.f;
            }
            if (WF && !link_local) {       // the other layer's cluster is on another XCD: layer 0's output is written through
#pragma unroll
                for (int o = 0; o < NOWN; ++o)
                    if (bidx[o] < B)
                        __hip_atomic_store(reinterpret_cast<unsigned*>(reinterpret_cast<char*>(outb[o]) + (size_t)(unsigned)tt[o] * out_row_bytes),
                                           __float_as_uint(oval[o]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
#pragma unroll
            for (int o = 0; o < NOWN; ++o)
                if (bidx[o] < B) *reinterpret_cast<float*>(reinterpret_cast<char*>(outb[o]) + (size_t)(unsigned)tt[o] * out_row_bytes) = oval[o];
            }
        } else if (FK) {
            // this layer's cell and the rider's cell (row 4*kq + lane/8, unit lane%8 on lanes 0..31; the other lanes compute the
            // same numbers and keep nothing) in ONE branch-free block, all stores behind it: the scheduler interleaves the two
            // dependent exp / rcp chains.  Rider gates: i = tile 0 column u, f = tile 0 column 8 + u, g / o = the same columns of
            // tile 1; rows 4*kq .. 4*kq+3 are accumulator regs 0 .. 3 of the lanes with q == kq.
            const bool act = step < blen[0];
            const int tt = act ? (d.reverse ? blen[0] - 1 - step : step) : step;
            const bool fact = step < flen;                       // (flen = 0 on lanes 32..63 and for rows past the batch)
            const int ftt = fact ? (fdir ? flen - 1 - step : step) : step;
            const float ig = sigmoidf_(gate[0][0]), fig = sigmoidf_(gi);
            const float fg = sigmoidf_(gate[0][1]), ffg = sigmoidf_(gf);
            const float gv = tanhf_(gate[0][2]), fgv = tanhf_(gg);
            const float og = sigmoidf_(gate[0][3]), fog = sigmoidf_(go);
            const float cnew = fg * cst[0] + ig * gv, fcnew = ffg * fcst + fig * fgv;
            const float hnew = og * tanhf_(cnew), fhnew = fog * tanhf_(fcnew);
            cst[0] = act ? cnew : cst[0];
            hst[0] = act ? hnew : hst[0];
            fcst = fact ? fcnew : fcst;
            fhst = fact ? fhnew : fhst;
            unsigned* hw = hdL + (size_t)(step & 1) * 16 * H + hslot[0];
            const unsigned hword = hword_of(hst[0], step), fhword = fword_of(fhst, step);
            __hip_atomic_store(hw, hword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!all_local) __hip_atomic_store(hw + HD_R / 4, hword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane < 32) {
                unsigned* fwd = hdL + F_WORD0 + (unsigned)(step & 1) * 2048u + (unsigned)fblk * 1024u + fslot;
                __hip_atomic_store(fwd, fhword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!all_local) __hip_atomic_store(fwd + F_TR, fhword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (frow < B) a.f_out[((size_t)ftt * B + frow) * 128 + fdir * 64 + funit] = fact ? fhnew : 0.f;
            }
            if (bidx[0] < B) *reinterpret_cast<float*>(reinterpret_cast<char*>(outb[0]) + (size_t)(unsigned)tt * out_row_bytes) = act ? hnew : 0.f;
        } else
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            const bool act = step < blen[o];
            const int tt = act ? (d.reverse ? blen[o] - 1 - step : step) : step;
            float oval = 0.f;
            if (act) {
                const float ig = sigmoidf_(gate[o][0]);
                const float fg = sigmoidf_(gate[o][1]);
                const float gg = tanhf_(gate[o][2]);
                const float og = sigmoidf_(gate[o][3]);
                cst[o] = fg * cst[o] + ig * gg;
                hst[o] = og * tanhf_(cst[o]);
                oval = hst[o];
            }
            if (TAGX) {
                unsigned* hw = hdL + (size_t)(step & 1) * 16 * H + hslot[o];
                const unsigned hword = hword_of(hst[o], step);
                __hip_atomic_store(hw, hword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (!all_local) __hip_atomic_store(hw + HD_R / 4, hword, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
            const int gi = granule_index(q * 4 + reg0 + o, jown);
            granule_store_l2(hxL + doff + gi, a.epoch_base + (unsigned)(step + 1), hst[o]);
            if (!all_local) granule_store(hxR + doff + gi, a.epoch_base + (unsigned)(step + 1), hst[o]);
            }
            if (bidx[o] < B) *reinterpret_cast<float*>(reinterpret_cast<char*>(outb[o]) + (size_t)(unsigned)tt * out_row_bytes) = oval;
        }
        PROF_E(4);
    }
    if (wf_l0) link_publish(T);                               // the last step's outputs are in the buffer
    if (PROF && prof) {
        long long* o = a.prof + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 5; ++i) o[i] = pt[i];
        o[5] = T;
        o[6] = pt[5];
        o[7] = (all_local ? 256 : 0) | (link_local ? 512 : 0) | my_xcc | ((long long)link_waits << 16);   // placement: all slices on my XCD?, partner cluster too?, XCC id; WF: steps that waited for layer 0
        long long* tl = a.prof + 4096 + (size_t)blockIdx.x * 4;                    // launch timeline (tools/debug/launch_timeline.py)
        tl[0] = tl_entry; tl[1] = tl_w; tl[2] = tl_loop; tl[3] = (long long)__builtin_amdgcn_s_memrealtime();
    }

    // ---- final state (h_n, c_n of models/rnn.py:33) back to hbuf / cbuf
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        if (bidx[o] < B) {
            d.hbuf[(size_t)bidx[o] * H + jown] = hst[o];
            d.cbuf[(size_t)bidx[o] * H + jown] = cst[o];
        }
    }
}


// tile lt of wave kq:  g = lt / NUB, ub = lt % NUB
// W_hh: dst[(((slice*4 + kq)*NKS + ks)*NTW + lt)*64 + lane]
//         = W_hh[g*H + slice*U + ub*16 + (lane&15)][kq*KW + 4*ks + (lane>>4)]
template <int H, int NSLICE>
MP_KERNEL void mp_pack_whh_persist(const float* __restrict__ whh, float* __restrict__ dst) {
    constexpr int U = H / NSLICE, NUB = U / 16, NWV = 4, NTW = 4 * NUB, KW = H / 4, NKS = KW / 4;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)4 * H * H) return;
    const int lane = idx & 63;
    size_t rest = idx >> 6;
    const int lt = rest % NTW; rest /= NTW;
    const int ks = rest % NKS; rest /= NKS;
    const int kq = rest % NWV; rest /= NWV;
    const int slice = (int)rest;
    const int g = lt / NUB, ub = lt % NUB;
    const int row = g * H + slice * U + ub * 16 + (lane & 15);
    const int col = kq * KW + 4 * ks + (lane >> 4);
    dst[idx] = whh[(size_t)row * H + col];
}

// W_ih: dst[((((slice*4 + kq)*NXS + s)*NTG + tg)*64 + lane)*4 + i]   (tile lt = tg*4 + i)
//         = W_ih[g*H + slice*U + ub*16 + (lane&15)][kq*KQ + (s/4)*16 + (lane>>4)*4 + (s%4)]
template <int H, int NSLICE>
MP_KERNEL void mp_pack_wih_persist(const float* __restrict__ wih, float* __restrict__ dst, int KIN) {
    constexpr int U = H / NSLICE, NUB = U / 16, NWV = 4, NTW = 4 * NUB, NTG = NTW / 4;
    const int KQ = KIN / 4, NXS = KQ / 4;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)4 * H * KIN) return;
    const int i = idx & 3;
    const int lane = (idx >> 2) & 63;
    size_t rest = idx >> 8;
    const int tg = rest % NTG; rest /= NTG;
    const int s = rest % NXS; rest /= NXS;
    const int kq = rest % NWV; rest /= NWV;
    const int slice = (int)rest;
    const int lt = tg * 4 + i;
    const int g = lt / NUB, ub = lt % NUB;
    const int row = g * H + slice * U + ub * 16 + (lane & 15);
    const int col = kq * KQ + (s >> 2) * 16 + (lane >> 4) * 4 + (s & 3);
    dst[idx] = wih[(size_t)row * KIN + col];
}

constexpr int kExclusiveLds = 84 * 1024;                             // > half of a CU's 160 KB

template <int H, int NSLICE, int KIN>
constexpr size_t fused_lds() {
    using C = Cfg<H, NSLICE, KIN>;
    return (size_t)C::RED_F4 * 16 * (C::ALLREG ? 2 : 1) + (size_t)C::NWV * C::XL * C::NTG * 64 * 16;   // (ALLREG: two copies, RED_DB)
}
constexpr size_t kVfLds = fused_lds<256, 16, 256>() + 4 * 2 * 64 * 16;         // + the rider's partial sums [kq][tile][lane]

template <int FK>
void launch_vf(const LstmPersistArgs& a, hipStream_t s) {
    size_t lds = kVfLds;
    if ((size_t)a.min_lds > lds) lds = (size_t)a.min_lds;
    LstmPersistArgs b = a;
    int most = 0, total = 0;
    for (int x = 0; x < 8; ++x) { most = b.xcd_cnt[x] > most ? b.xcd_cnt[x] : most; total += b.xcd_cnt[x]; }
    if (total != a.nslab * a.ndir) {
        mp_fill_xcd_table(b, nullptr);
        most = (a.nslab * a.ndir + 7) / 8;
    }
    const dim3 grid(8 * most * 16);
    // (PROF instantiation: phase counters and the test hooks -- a dropped workgroup, the forced any-placement transport)
    if (a.prof || a.debug_drop || a.force_remote) hipLaunchKernelGGL((mp_lstm_fused<256, 16, 256, true, FK>), grid, dim3(256), lds, s, b);
    else hipLaunchKernelGGL((mp_lstm_fused<256, 16, 256, false, FK>), grid, dim3(256), lds, s, b);
}
template <int FK>
hipError_t vf_attrs() {
    hipError_t e = hipFuncSetAttribute((const void*)mp_lstm_fused<256, 16, 256, true, FK>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVfLds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)mp_lstm_fused<256, 16, 256, false, FK>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)kVfLds);
}

// Rider weights of one direction: dst[((((ug*4 + kq)*FNS + st)*2 + tile)*64 + lane] (ug = group of 8 units, FNS = FK/16 + 4)
//   = W[gate*64 + 8*ug + n%8][k],  gate = 2*tile + n/8,  n = lane % 16,  q = lane / 16
//   st <  FK/16: W = W_ih, k = kq*FK/4 + (st/4)*16 + q*4 + st%4;   st >= FK/16: W = W_hh, k = kq*16 + 4*(st - FK/16) + q
MP_KERNEL void mp_pack_foot_vf(const float* __restrict__ wih, const float* __restrict__ whh, float* __restrict__ dst, int FK) {
    const int FNX = FK / 16, FNS = FNX + 4;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)8 * 4 * FNS * 2 * 64) return;
    const int lane = idx & 63;
    size_t rest = idx >> 6;
    const int tile = rest & 1; rest >>= 1;
    const int st = rest % FNS; rest /= FNS;
    const int kq = rest & 3; rest >>= 2;
    const int ug = (int)rest;
    const int n = lane & 15, q = lane >> 4;
    const int row = (2 * tile + (n >> 3)) * 64 + 8 * ug + (n & 7);
    if (st < FNX) dst[idx] = wih[(size_t)row * FK + kq * (FK / 4) + (st >> 2) * 16 + q * 4 + (st & 3)];
    else dst[idx] = whh[(size_t)row * 64 + kq * 16 + 4 * (st - FNX) + q];
}

constexpr size_t kRiderLds8 = 2 * 4 * 2 * 64 * 16;                   // 8-slice kernels: two copies of the rider's partial sums

template <int H, int NSLICE, int KIN, int FK = 0, bool WF = false>
void launch_fused(const LstmPersistArgs& a, hipStream_t s) {
    using C = Cfg<H, NSLICE, KIN>;
    size_t lds = fused_lds<H, NSLICE, KIN>() + (FK ? kRiderLds8 : 0);
    if ((size_t)a.min_lds > lds) lds = (size_t)a.min_lds;
    LstmPersistArgs b = a;
    int most = 0, total = 0;
    for (int x = 0; x < 8; ++x) { most = b.xcd_cnt[x] > most ? b.xcd_cnt[x] : most; total += b.xcd_cnt[x]; }
    if (total != a.nslab * a.ndir) {                      // no table (or not one for this launch): round robin
        mp_fill_xcd_table(b, nullptr);
        most = (a.nslab * a.ndir + 7) / 8;
    }
    const dim3 grid(8 * most * NSLICE);
    if (a.prof || a.debug_drop || a.force_remote) hipLaunchKernelGGL((mp_lstm_fused<H, NSLICE, KIN, true, FK, WF>), grid, dim3(64 * C::NWV), lds, s, b);
    else hipLaunchKernelGGL((mp_lstm_fused<H, NSLICE, KIN, false, FK, WF>), grid, dim3(64 * C::NWV), lds, s, b);
}

// the dynamic-LDS limit is a per-device function attribute: set for the CURRENT device, outside of any capture
template <int H, int NSLICE, int KIN, int FK = 0, bool WF = false>
hipError_t fused_attrs() {
    int lds = (int)(fused_lds<H, NSLICE, KIN>() + (FK ? kRiderLds8 : 0));
    if (lds < kExclusiveLds) lds = kExclusiveLds;                  // (LstmPersistArgs::min_lds)
    hipError_t e = hipFuncSetAttribute((const void*)mp_lstm_fused<H, NSLICE, KIN, true, FK, WF>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)mp_lstm_fused<H, NSLICE, KIN, false, FK, WF>,
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace

// nslice: slices per slab -- H = 256: 16 or 8 (one four-wave workgroup per slice either way); H = 64: 4
void mp_launch_pack_whh_persist(const float* whh, float* dst, int H, int nslice, hipStream_t s) {
    const size_t n = (size_t)4 * H * H;
    const int grid = (int)((n + 255) / 256);
    if (H == 256 && nslice == 16) hipLaunchKernelGGL((mp_pack_whh_persist<256, 16>), dim3(grid), dim3(256), 0, s, whh, dst);
    else if (H == 256) hipLaunchKernelGGL((mp_pack_whh_persist<256, 8>), dim3(grid), dim3(256), 0, s, whh, dst);
    else hipLaunchKernelGGL((mp_pack_whh_persist<64, 4>), dim3(grid), dim3(256), 0, s, whh, dst);
}

void mp_launch_pack_wih_persist(const float* wih, float* dst, int H, int KIN, int nslice, hipStream_t s) {
    const size_t n = (size_t)4 * H * KIN;
    const int grid = (int)((n + 255) / 256);
    if (H == 256 && nslice == 16) hipLaunchKernelGGL((mp_pack_wih_persist<256, 16>), dim3(grid), dim3(256), 0, s, wih, dst, KIN);
    else if (H == 256) hipLaunchKernelGGL((mp_pack_wih_persist<256, 8>), dim3(grid), dim3(256), 0, s, wih, dst, KIN);
    else hipLaunchKernelGGL((mp_pack_wih_persist<64, 4>), dim3(grid), dim3(256), 0, s, wih, dst, KIN);
}

int mp_persist_max_wg(int H, int nslice) { return H == 256 && nslice == 16 ? 512 : 256; }

hipError_t mp_lstm_persist_device_attrs() {
    hipError_t e = hipSuccess;
    if (!e) e = fused_attrs<256, 16, 256>();
    if (!e) e = fused_attrs<256, 16, 512>();
    if (!e) e = fused_attrs<256, 8, 256>();
    if (!e) e = fused_attrs<256, 8, 512>();
    if (!e) e = fused_attrs<256, 8, 256, 64>();                   // pose layer 0 with foot-contact layer 0 riding
    if (!e) e = fused_attrs<256, 8, 256, 128, true>();            // velocity wavefront with foot-contact layer 1 riding
    if (!e) e = fused_attrs<256, 8, 256, 0, true>();              // velocity wavefront alone
    if (!e) e = fused_attrs<64, 4, 64>();
    if (!e) e = fused_attrs<64, 4, 128>();
    if (!e) e = vf_attrs<64>();
    if (!e) e = vf_attrs<128>();
    return e;
}

void mp_launch_lstm_vf(const LstmPersistArgs& a, int fk, hipStream_t s) {
    if (fk == 64) launch_vf<64>(a, s);
    else launch_vf<128>(a, s);
}
size_t mp_foot_vf_floats(int fk) { return (size_t)8 * 4 * (fk / 16 + 4) * 2 * 64; }
void mp_launch_pack_foot_vf(const float* wih, const float* whh, float* dst, int fk, hipStream_t s) {
    const size_t n = mp_foot_vf_floats(fk);
    hipLaunchKernelGGL(mp_pack_foot_vf, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wih, whh, dst, fk);
}

// the K_in = 256 kernel on 8 slices with its options: fk = K_in of a riding H = 64 layer (0: none; 64 = its layer 0 in a
// bidirectional launch, 128 = its layer 1 in a wavefront launch), wf = the two "directions" are the two layers of a
// unidirectional block (d[0] = layer 0, d[1] = layer 1, d[1].xin = d[0].out)
bool mp_launch_lstm_persist8(const LstmPersistArgs& a, int fk, bool wf, hipStream_t s) {
    if (fk == 0 && !wf) launch_fused<256, 8, 256>(a, s);
    else if (fk == 64 && !wf) launch_fused<256, 8, 256, 64>(a, s);
    else if (fk == 128 && wf) launch_fused<256, 8, 256, 128, true>(a, s);
    else if (fk == 0 && wf) launch_fused<256, 8, 256, 0, true>(a, s);
    else return false;
    return true;
}

void mp_launch_lstm_persist(const LstmPersistArgs& a, int H, int KIN, int nslice, hipStream_t s) {
    if (H == 256 && nslice == 16) {
        if (KIN == 256) launch_fused<256, 16, 256>(a, s);
        else launch_fused<256, 16, 512>(a, s);
    } else if (H == 256) {                                         // 8 slices: four 512-register waves, AccVGPR-resident weights
        if (KIN == 256) launch_fused<256, 8, 256>(a, s);
        else launch_fused<256, 8, 512>(a, s);
    } else if (KIN == 64) launch_fused<64, 4, 64>(a, s);         // H = 64 (foot contact): 4 slices of 16 units, small
    else launch_fused<64, 4, 128>(a, s);                          // footprint so that it fits beside the velocity layers
}


// cnt[x] clusters on XCD x (nullptr: the ndir * nslab clusters spread round robin), bases = running sum
void mp_fill_xcd_table(LstmPersistArgs& a, const unsigned char* cnt) {
    const int ncl = a.ndir * a.nslab;
    int base = 0;
    for (int x = 0; x < 8; ++x) {
        const int c = cnt ? cnt[x] : (ncl + 7 - x) / 8;
        a.xcd_cnt[x] = (unsigned char)c;
        a.xcd_base[x] = (unsigned short)base;
        base += c;
    }
}

// ---- does this device deal workgroups round robin over 8 XCDs?  (64 workgroups report their XCC ids)
namespace {
MP_KERNEL void mp_xcc_probe(int* out) { if (threadIdx.x == 0) out[blockIdx.x] = (int)xcc_id(); }
}
void mp_launch_xcc_probe(int* out64, hipStream_t s) { hipLaunchKernelGGL(mp_xcc_probe, dim3(64), dim3(64), 0, s, out64); }

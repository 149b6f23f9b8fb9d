// K2x -- the persistent fused nn.LSTM layer of mp_lstm_persist.hip (models/rnn.py:27) with SPLIT-bf16 MFMA
// operands: same mapping, same hand-off protocol, same fp32 state / gates / accumulation, but every fp32
// product a*w inside the two matrix products of a step is evaluated as
//        a_hi*w_hi + a_hi*w_lo + a_lo*w_hi         (a = a_hi + a_lo, w = w_hi + w_lo, each part a bf16 number)
// on v_mfma_f32_16x16x32_bf16 (fp32 accumulate).  One such MFMA covers 8x the K of v_mfma_f32_16x16x4_f32 in
// about the same issue time, so 3 of them replace 8 fp32 MFMAs: the matrix pipe is no longer what bounds a
// step.  The dropped a_lo*w_lo term is <= 2^-18 relative per product, hi+lo itself carries 16 significand
// bits; measured end to end (oracle experiment and tests/test_gpu_parity.py) the outputs stay 4e-7 from the
// fp32 reference -- 250x inside the 1e-4 parity bound, the same distance as the exact-fp32 kernel's.
// A single bf16 term (1e-4 .. 5e-4) does NOT pass; that is why the split exists.
//
// Data formats (all 4 bytes per value, so buffers, LDS images and register counts keep their sizes):
//   * activations that feed an LSTM layer (linear1's output X1, layer 0's output) and the h granules are
//     "pairs": (bf16 hi << 16) | bf16 lo  (mp_lstm_dev.h pair_of);
//   * W_hh / W_ih are packed as separate hi and lo bf16x8 B-fragments (mp_pack_*_x3 below);
//   * a lane's A fragment (8 consecutive k of one sequence row) is built from 8 pair words with 8 v_perm_b32.
// k mapping: wave kq owns K quarter kq; chunk c = 32 k of it; lane (row r16, k-block q) holds
//   k = kq*KQ + c*32 + q*8 + e,  e = 0..7.
// Granules are laid out [k/32][k%8][row][(k/8)%4]: the 64 lanes of a consumer wave still read 512 contiguous
// bytes per instruction and a producer workgroup's granules still form one contiguous block.
// H = 256 only (a K quarter must hold a 32-wide chunk); the H = 64 foot-contact block keeps the fp32 kernel.
#include "mp_lstm_dev.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// 8 pair words (k = e) -> hi fragment (8 bf16, element e in the low/high half of dword e/2) and lo fragment
__device__ __forceinline__ void split_pairs(u32x4 w0, u32x4 w1, u32x4& hi, u32x4& lo) {
    hi[0] = __builtin_amdgcn_perm(w0[1], w0[0], 0x07060302u);
    hi[1] = __builtin_amdgcn_perm(w0[3], w0[2], 0x07060302u);
    hi[2] = __builtin_amdgcn_perm(w1[1], w1[0], 0x07060302u);
    hi[3] = __builtin_amdgcn_perm(w1[3], w1[2], 0x07060302u);
    lo[0] = __builtin_amdgcn_perm(w0[1], w0[0], 0x05040100u);
    lo[1] = __builtin_amdgcn_perm(w0[3], w0[2], 0x05040100u);
    lo[2] = __builtin_amdgcn_perm(w1[1], w1[0], 0x05040100u);
    lo[3] = __builtin_amdgcn_perm(w1[3], w1[2], 0x05040100u);
}
// granule index of (row, hidden unit j) inside one [16][256] slab-parity block: [j/32][j%8][row][(j/8)%4]
__device__ __forceinline__ int granule_index_x3(int row, int j) {
    return ((((j >> 5) * 8 + (j & 7)) * 16 + row) << 2) + ((j >> 3) & 3);
}

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
    static constexpr int RED_BUFS = KIN > H ? 1 : 2;
    static constexpr bool BIG = KIN > H;
    static constexpr int WG_PER_CU = NSLICE == 16 ? 2 : 1;
    static constexpr int NPW = NSLICE / 4;            // producer slices inside one wave's K quarter (2 | 4)
    static constexpr int LDS_BYTES = RED_BUFS * RED_F4 * 16 + NWV * XLC * CH_U4 * 16;
    static_assert(LDS_BYTES <= (NSLICE == 16 ? 80 : 160) * 1024, "LDS budget (two workgroups per CU for the 4-wave packing)");
};

constexpr int x3_threads(int nslice) { return 64 * 4 * (256 / nslice / 16); }
constexpr int x3_wg_per_cu(int nslice) { return nslice == 16 ? 2 : 1; }

template <int NSLICE, int KIN, bool PROF>
MP_KERNEL __launch_bounds__(x3_threads(NSLICE), x3_wg_per_cu(NSLICE)) void mp_lstm_x3(LstmPersistArgs a) {
    using C = CfgX<NSLICE, KIN>;
    constexpr int H = 256, U = C::U, NWV = C::NWV, KQ = C::KQ, NXC = C::NXC, NHC = C::NHC, XLC = C::XLC, XRC = C::XRC;
    constexpr int NPW = C::NPW, CH_U4 = C::CH_U4;
    constexpr int NTHREADS = 64 * NWV;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);                        // [finishing wave][source kq][lane]
    u32x4* wxl = reinterpret_cast<u32x4*>(smem) + C::RED_BUFS * C::RED_F4;   // [wave][LDS chunk][tile*2+part][lane]

    // block -> (cluster = (direction, slab), slice): see mp_lstm_persist.hip (slices of a cluster share an XCD)
    const int ncl = a.ndir * a.nslab;
    const int cl = ((int)(blockIdx.x >> 3) / NSLICE) * 8 + (int)(blockIdx.x & 7);
    const int slice = (int)(blockIdx.x >> 3) % NSLICE;
    if (cl >= ncl) return;
    const int dir = cl / a.nslab, slab = cl % a.nslab;
    const LstmDir d = a.d[dir];
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
        const float* p = d.hbuf + (size_t)(arow_in ? arow : 0) * H + kq * 64 + q * 8;
#pragma unroll
        for (int c = 0; c < NHC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e)
                hw[c][e >> 2][e & 3] = (arow_in && !a.zero_state) ? pair_of(p[c * 32 + e]) : 0u;
    }

    // granules of this slab: hx[cluster] = { L[2 parities][16*H], R[2 parities][16*H], xcc[16] }
    constexpr size_t SLABW = (size_t)4 * 16 * H + 16;
    u64* hxL = a.hx + (size_t)cl * SLABW;
    u64* hxR = hxL + (size_t)2 * 16 * H;
    u64* xtab = hxL + (size_t)4 * 16 * H;
    unsigned spin_budget = a.max_spin;
    const unsigned my_xcc = xcc_id();
    unsigned long long same = ~0ull;
    bool all_local = true;
    {
        if (threadIdx.x == 0) granule_store(xtab + slice, XCC_TAG, __uint_as_float(my_xcc));
        unsigned peer = my_xcc;
        if (lane < NSLICE) {
            unsigned spins = 0;
            while (true) {
                const u64 g = granule_load(xtab + lane);
                if ((unsigned)(g >> 32) == XCC_TAG) { peer = (unsigned)g; break; }
                if (++spins > spin_budget) { atomicExch(a.err, 1000000); peer = ~0u; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        same = __ballot(peer == my_xcc);
        all_local = (same & ((1ull << NSLICE) - 1)) == ((1ull << NSLICE) - 1);
        if (__ballot(peer == ~0u)) spin_budget = 0;
        if (a.force_remote) { all_local = false; same = 0; }      // test hook: exercise the any-placement transport
    }
    // producer slice of unit k = kq*64 + c*32 + q*8 + e:  NSLICE = 8: 2*kq + c;  NSLICE = 16: 4*kq + 2*c + (q >> 1)
    const u64* srcb[NHC];                       // this lane's source block per chunk (parity / chunk offsets added later)
#pragma unroll
    for (int c = 0; c < NHC; ++c) {
        const int prod = NSLICE == 8 ? 2 * kq + c : 4 * kq + 2 * c + (q >> 1);
        srcb[c] = (((same >> prod) & 1) ? hxL : hxR) + (size_t)((kq * 2 + c) * 8) * 64 + r16 * 4 + q;
    }
    // gate lanes: lane i < NPW watches one granule of producer slice NPW*kq + i
    const u64* gatep = hxL;
    if (lane < NPW) {
        const int prod = NPW * kq + lane;
        const int gc = NSLICE == 8 ? lane : lane >> 1, gq = NSLICE == 8 ? 0 : 2 * (lane & 1);
        gatep = (((same >> prod) & 1) ? hxL : hxR) + (size_t)((kq * 2 + gc) * 8) * 64 + gq;
    }

    // ---- x_0: pair words of this lane's row, chunk c: k = kq*KQ + c*32 + q*8 + e
    u32x4 xw[NXC][2];
    constexpr bool SPLIT_X = C::BIG;                      // second half of x_t fetched at the top of step t
    constexpr int XC_PRE = SPLIT_X ? XRC : NXC;
    auto load_x = [&](int step, int c0, int c1) {
        const bool on = step < alen;
        const int t = on ? (d.reverse ? alen - 1 - step : step) : 0;
        const unsigned* p = xbase + (size_t)t * xtstride;
#pragma unroll
        for (int c = 0; c < NXC; ++c)
            if (c >= c0 && c < c1) {
                xw[c][0] = on ? *reinterpret_cast<const u32x4*>(p + c * 32) : u32x4{0u, 0u, 0u, 0u};
                xw[c][1] = on ? *reinterpret_cast<const u32x4*>(p + c * 32 + 4) : u32x4{0u, 0u, 0u, 0u};
            }
    };
    load_x(0, 0, XC_PRE);
    __syncthreads();                                          // W_ih LDS image complete

    long long pt[6] = {0, 0, 0, 0, 0, 0};
    const bool prof = PROF && a.prof != nullptr && threadIdx.x == 0;
#define PROF_T(i) do { if (PROF && prof) pt[i] -= (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define PROF_E(i) do { if (PROF && prof) pt[i] += (long long)__builtin_amdgcn_s_memtime(); } while (0)

    const u32x4* wxw = wxl + (size_t)wave * XLC * CH_U4 + lane;

    f32x4 acc[4];
    // one chunk of a matrix product: 3 MFMAs per gate tile (hi*hi, hi*lo, lo*hi), tiles interleaved so that
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
    };
    u32x4 wl[8];                                               // the LDS-resident chunk being multiplied next
    auto lds_chunk = [&](int c) {                              // c: LDS chunk index in [0, XLC)
#pragma unroll
        for (int tp = 0; tp < 8; ++tp) wl[tp] = wxw[(size_t)(c * 8 + tp) * 64];
    };

    for (int step = 0; step < T; ++step) {
        PROF_T(0);
#ifdef X3_TRACE
        long long* tr = (PROF && a.prof && lane == 0 && step >= 64 && step < 96) ? a.prof + 4096 + ((size_t)(blockIdx.x * 8 + wave) * 32 + (step - 64)) * 8 : nullptr;
#define TR(i) do { if (tr) tr[i] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#else
#define TR(i) do { } while (0)
#endif
        TR(0);
        if (SPLIT_X) load_x(step, XC_PRE, NXC);
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // ---- x_t W_ih^T, register-resident chunks (independent of h: this is what fills the wait for the peers);
        // the first LDS-resident chunk is fetched underneath
        lds_chunk(0);
        TR(6);
#ifndef X3_SKIP_PROJ
#pragma unroll
        for (int c = 0; c < XRC; ++c) {
            chunk_mma(xw[c][0], xw[c][1], wxr[c]);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif

        TR(7);
        // ---- request h_{step-1}: 16 granules per lane, 512 contiguous bytes per instruction
        u64 gr[NHC][8];
        const unsigned epoch = (unsigned)step;                 // written by the producers at the end of step-1
        const size_t poff = (size_t)((step + 1) & 1) * 16 * H;
        constexpr bool EARLY_GATHER = !C::BIG;
        if (EARLY_GATHER && step > 0) {
#pragma unroll
            for (int c = 0; c < NHC; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) gr[c][e] = granule_load(srcb[c] + poff + (size_t)e * 64);
        }
        TR(1);
        // ---- LDS-resident chunks
#ifndef X3_SKIP_PROJ
#pragma unroll
        for (int c = 0; c < XLC; ++c) {
            chunk_mma(xw[XRC + c][0], xw[XRC + c][1], wl);
            __builtin_amdgcn_sched_barrier(0);
            if (c + 1 < XLC) lds_chunk(c + 1);
        }
#endif
        PROF_E(0); PROF_T(1);

        // ---- validate the granules; the slow path (cheap gate, then sweep) only runs when some were stale
        if (step > 0) {
            if (!EARLY_GATHER) {
#pragma unroll
                for (int c = 0; c < NHC; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) gr[c][e] = granule_load(srcb[c] + poff + (size_t)e * 64);
            }
            bool ok = true;
#pragma unroll
            for (int c = 0; c < NHC; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) ok = ok && ((unsigned)(gr[c][e] >> 32) == epoch);
            unsigned spins = 0;
            bool timed_out = false;
            if (PROF && prof && !__all(ok)) pt[5] += 1;      // slow-path entries
            while (!__all(ok) && !timed_out) {
                while (true) {
                    bool ready = true;
                    if (lane < NPW) ready = (unsigned)(granule_load(gatep + poff) >> 32) == epoch;
                    if (__all(ready)) break;
                    if (++spins > spin_budget) { timed_out = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                ok = true;
#pragma unroll
                for (int c = 0; c < NHC; ++c)
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        gr[c][e] = granule_load(srcb[c] + poff + (size_t)e * 64);
                        ok = ok && ((unsigned)(gr[c][e] >> 32) == epoch);
                    }
                if (++spins > spin_budget) timed_out = true;
            }
            if (timed_out) {                                   // bounded: flag the error and never wait again
                if (lane == 0) atomicExch(a.err, 1 + step);
                spin_budget = 0;
            }
#pragma unroll
            for (int c = 0; c < NHC; ++c)
#pragma unroll
                for (int e = 0; e < 8; ++e) hw[c][e >> 2][e & 3] = (unsigned)gr[c][e];
        }
        TR(2);
#ifndef X3_SKIP_XLOAD
        load_x(step + 1, 0, XC_PRE);     // next step's x
#endif
        //, issued after the granule wait (see mp_lstm_persist.hip)
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
        if (C::RED_BUFS == 1) __syncthreads();                 // previous step's reads of `red` are done
#pragma unroll
        for (int dk = 0; dk < 4; ++dk)
            redb[((tw * 4 + dk) * 4 + kq) * 64 + lane] = f32x4{acc[0][dk], acc[1][dk], acc[2][dk], acc[3][dk]};
        __syncthreads();
        f32x4 gate = redb[(wave * 4 + 0) * 64 + lane];
#pragma unroll
        for (int sw = 1; sw < 4; ++sw) gate += redb[(wave * 4 + sw) * 64 + lane];
        gate += bias4;
        TR(4);
        PROF_E(3); PROF_T(4);

        // ---- cell update (fp32, register-local), publish h_step as a pair, write the layer output
        const size_t doff = (size_t)(step & 1) * 16 * H;
        const bool act = step < blen;
        const int tt = act ? (d.reverse ? blen - 1 - step : step) : step;
        float oval = 0.f;
        if (act) {
            const float ig = sigmoidf_(gate[0]);
            const float fg = sigmoidf_(gate[1]);
            const float gg = tanhf_(gate[2]);
            const float og = sigmoidf_(gate[3]);
            cst = fg * cst + ig * gg;
            hst = og * tanhf_(cst);
            oval = hst;
        }
        const unsigned hp = pair_of(hst);
        const int gi = granule_index_x3(q * 4 + kq, jown);
        granule_store_l2_bits(hxL + doff + gi, (unsigned)(step + 1), hp);
        if (!all_local) granule_store_bits(hxR + doff + gi, (unsigned)(step + 1), hp);
#ifdef X3_SKIP_OUT
        if (inb && step == T - 1) {
#else
        if (inb) {
#endif
            float* op = d.out + ((size_t)tt * B + bown) * d.outStride + jown;
            if (a.out_pairs) *reinterpret_cast<unsigned*>(op) = act ? hp : 0u;
            else *op = oval;
        }
        TR(5);
        PROF_E(4);
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
    const unsigned p0 = pair_of(w[(size_t)row * K + col]);
    const unsigned p1 = pair_of(w[(size_t)row * K + col + 1]);
    dst[idx] = part == 0 ? ((p1 & 0xffff0000u) | (p0 >> 16)) : ((p1 << 16) | (p0 & 0xffffu));
}

template <int NSLICE, int KIN>
void launch_x3(const LstmPersistArgs& a, hipStream_t s) {
    using C = CfgX<NSLICE, KIN>;
    const size_t lds = (size_t)C::LDS_BYTES;
    const dim3 grid(((a.nslab * a.ndir + 7) / 8) * 8 * NSLICE);
    if (a.prof) {
        static bool once = (hipFuncSetAttribute((const void*)mp_lstm_x3<NSLICE, KIN, true>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
        (void)once;
        hipLaunchKernelGGL((mp_lstm_x3<NSLICE, KIN, true>), grid, dim3(64 * C::NWV), lds, s, a);
    } else {
        static bool once = (hipFuncSetAttribute((const void*)mp_lstm_x3<NSLICE, KIN, false>,
                                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds), true);
        (void)once;
        hipLaunchKernelGGL((mp_lstm_x3<NSLICE, KIN, false>), grid, dim3(64 * C::NWV), lds, s, a);
    }
}

}  // namespace

// H = 256 only; nslice 8 (8-wave workgroups) or 16 (4-wave workgroups, two per CU); K = 256 | 512
void mp_launch_pack_w_x3(const float* w, float* dst, int K, int nslice, hipStream_t s) {
    const size_t n = (size_t)4 * 256 * K;
    const int grid = (int)((n + 255) / 256);
    if (nslice == 16) hipLaunchKernelGGL((mp_pack_w_x3<16>), dim3(grid), dim3(256), 0, s, w, reinterpret_cast<unsigned*>(dst), K);
    else hipLaunchKernelGGL((mp_pack_w_x3<8>), dim3(grid), dim3(256), 0, s, w, reinterpret_cast<unsigned*>(dst), K);
}

void mp_launch_lstm_x3(const LstmPersistArgs& a, int KIN, int nslice, hipStream_t s) {
    if (nslice == 16) {
        if (KIN == 256) launch_x3<16, 256>(a, s);
        else launch_x3<16, 512>(a, s);
    } else {
        if (KIN == 256) launch_x3<8, 256>(a, s);
        else launch_x3<8, 512>(a, s);
    }
}

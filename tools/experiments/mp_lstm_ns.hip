// EXPERIMENT, not part of the library (not compiled by __graft_entry__.build): measured, parity-green, no faster than the
// kernel it was meant to replace -- profiles/r02_nsplit.md.  To try it again: copy into mobileposer_amd/csrc/, declare
// mp_launch_lstm_ns / mp_launch_pack_w_ns / mp_lstm_ns_device_attrs in mp_common.h, pack W_hh / W_ih of the bidirectional
// H = 256 blocks with mp_launch_pack_w_ns and launch it where mp_api.hip launches mp_launch_lstm_persist_w.
//
// K2n -- the persistent fp32 nn.LSTM layer (models/rnn.py:27) of mp_lstm_persist.hip for H = 256 / 8 slices per slab, with
// the gate columns -- not the K range -- split over the four waves of a workgroup.
//
// mp_lstm_fused<256,8,KIN,1> gives every wave a quarter of K and all 8 gate tiles, so the four partial sums of every gate
// meet in LDS once per step: 64 KB per CU and step through a 128 B/clk port, two barriers, 32 register moves -- 740 of
// 14 870 cycles, and the cell update behind it waits for those LDS reads (another ~200).  Skipping the reduction (wrong
// results) takes the 256 x 125 forward from 3.83 to 3.62 ms (profiles/r02_nsplit.md).  Here wave w owns 8 of the
// workgroup's 32 hidden units -- 2 MFMA tiles whose 16 columns are 4 gates x 4 units -- over the WHOLE K range:
//   * no reduction: an accumulator is a finished gate pre-activation; the four gates of a (sequence, unit) cell sit in
//     four lanes of one wave and meet through a 2 KB wave-private LDS transpose (no barrier);
//   * all weights in registers (KIN = 512: 256 AccVGPRs + 128 VGPRs per lane; KIN = 256: 256 AccVGPRs), none in LDS;
//   * the A operand -- x_t and h_{t-1} of the slab's 16 sequences, which every wave now needs whole -- is staged in LDS
//     (x_{t+1} fetched one step ahead with fully coalesced 1 KB loads: 16 rows x (KIN + 4) floats, double buffered;
//     h_{t-1}: every wave fetches the 64 units of its two producer slices after the flag check and the four quarters are
//     joined by one barrier) and read as 16-byte fragments under the MFMAs (LDS reads are free beside an MFMA stream).
// Same flagged hand-off as mp_lstm_fused ("FLAGX": plain words + one flag per producer wave, raised under the next
// step's first MFMAs), same transports, same exchange area (values here simply row-major [row][unit]), same bounded
// waits / error word, same packed-sequence semantics.  The k-steps of one MFMA group take k = 16 i + 4 q + j (q = lane / 16)
// so that a lane's four consecutive k-steps are one 16-byte LDS read; sums run over k in that order in ONE accumulator
// (x first, then h) -- not bitwise the K-split kernels' four partial sums, equal to fp32 rounding
// (checked against mp_lstm_fused at 5e-6 on five shapes, both transports, when it was wired in).
#include "mp_lstm_dev.h"

namespace {

template <bool ZERO, bool WACC>
__device__ __forceinline__ void mfma_ns(f32x4& c, float a, float w) {
    if (ZERO) {
        if (WACC) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "a"(w));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, 0" : "=&v"(c) : "v"(a), "v"(w));
    } else {
        if (WACC) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "a"(w));
        else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(w));
    }
}
__device__ __forceinline__ void mfma_drain_ns() {           // 8-pass MFMA result -> first VALU / LDS read of it
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_nop 15\n\ts_nop 1" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

template <int KIN>
struct NsCfg {
    static constexpr int H = 256, NSLICE = 8, U = 32;
    static constexpr int NXS = KIN / 4, NXG = KIN / 16;       // x: k-steps, groups of 4 k-steps (one 16-byte A fragment)
    static constexpr int NHS = H / 4, NHG = H / 16;           // h: likewise
    static constexpr int KP = KIN + 4, HP = H + 4;            // LDS row pitches in floats (pitch mod 32 = 4: conflict-free)
    static constexpr int NXA = KIN == 512 ? NXS / 2 : NXS;    // x k-steps whose weights live in AccVGPRs (the rest: VGPRs)
    static constexpr int XLD = KIN / 64;                      // 16-byte x pieces per lane and step (4 rows x KIN floats per wave)
    static constexpr size_t LDS_BYTES = (size_t)(2 * 16 * KP + 16 * HP) * 4 + (size_t)4 * 2 * 64 * 16;
};

template <int KIN, bool PROF>
MP_KERNEL __launch_bounds__(256, 1) void mp_lstm_ns(LstmPersistArgs a) {
    using C = NsCfg<KIN>;
    constexpr int H = C::H, NSLICE = C::NSLICE, U = C::U, NXS = C::NXS, NXG = C::NXG, NHS = C::NHS, NHG = C::NHG;
    constexpr int KP = C::KP, HP = C::HP, NXA = C::NXA, XLD = C::XLD;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* xb = smem;                                                   // [2][16][KP]
    float* hb = smem + 2 * 16 * KP;                                     // [16][HP]
    f32x4* tb = reinterpret_cast<f32x4*>(hb + 16 * HP);                 // [wave][tile][lane]: the gate transpose

    // ---- which cluster (direction, slab) and slice: as mp_lstm_fused (host table by XCD, or round robin)
    const int ncl = a.ndir * a.nslab;
    const int xcd = a.xcd_physical ? (int)(xcc_id() & 7) : (int)(blockIdx.x & 7);
    const int kth = (int)(blockIdx.x >> 3) / NSLICE;
    const int slice = (int)(blockIdx.x >> 3) % NSLICE;
    if (kth >= (int)a.xcd_cnt[xcd]) return;
    const int cl = (int)a.xcd_base[xcd] + kth;
    if (cl >= ncl) return;
    const int dir = cl / a.nslab, slab = cl % a.nslab;
    const LstmDir d = a.d[dir];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int r16 = lane & 15, q = lane >> 4;                           // A operand: row r16, k = 16 i + 4 q + j
    const int g = (lane >> 2) & 3, u = lane & 3;                        // B / D column lane & 15 = gate g, unit u of the tile
    const int B = a.B, T = a.T;
    const int brow0 = (a.slab0 + slab) * 16;

    // ---- weights: [slice][wave][k-step][tile][lane], all in registers
    float wx[NXS][2], wh[NHS][2];
    {
        const float* px = d.wihpack + ((size_t)(slice * 4 + wave) * NXS * 2) * 64 + lane;
#pragma unroll
        for (int s = 0; s < NXS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) wx[s][t] = px[(size_t)(s * 2 + t) * 64];
        const float* ph = d.wpack + ((size_t)(slice * 4 + wave) * NHS * 2) * 64 + lane;
#pragma unroll
        for (int s = 0; s < NHS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) wh[s][t] = ph[(size_t)(s * 2 + t) * 64];
    }

    // ---- the two cells of this lane: sequence row 4 q + g of the slab, units slice*32 + wave*8 + t*4 + u
    const int crow = 4 * q + g;
    const int cb = brow0 + crow;
    const bool cin = cb < B;
    const int clen = cin ? a.lengths[cb] : 0;
    int junit[2];
    f32x4 bias4[2];
    float cst[2], hst[2];
    float* outb[2];
    const unsigned out_row_bytes = (unsigned)B * (unsigned)d.outStride * 4u;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        junit[t] = slice * U + wave * 8 + t * 4 + u;
        bias4[t] = *reinterpret_cast<const f32x4*>(d.bias + 4 * junit[t]);
        cst[t] = (cin && !a.zero_state) ? d.cbuf[(size_t)cb * H + junit[t]] : 0.f;
        hst[t] = (cin && !a.zero_state) ? d.hbuf[(size_t)cb * H + junit[t]] : 0.f;
        outb[t] = d.out + (size_t)(cin ? cb : 0) * d.outStride + junit[t];
    }

    // ---- exchange area of this cluster (layout of mp_lstm_fused; values row-major [row][unit] here), XCC table, transports
    constexpr size_t SLABW = (size_t)4 * 16 * H + 16;
    u64* hxL = a.hx + (size_t)cl * SLABW;
    u64* xtab = hxL + (size_t)4 * 16 * H;
    unsigned* hdL = reinterpret_cast<unsigned*>(hxL);                   // values: L [2 parities][16 * H], then R
    unsigned* hfL = hdL + (size_t)4 * 16 * H;                           // flags: L [2][NSLICE * 4], then R
    constexpr unsigned HD_R = 2 * 16 * H * 4;                           // byte offset of the R values
    constexpr unsigned HF_R = 2 * NSLICE * 4;                           // word offset of the R flags
    unsigned spin_budget = a.max_spin;
    const unsigned my_xcc = xcc_id();
    bool src_local[2] = {true, true};                                   // producer slices 2 wave, 2 wave + 1 on my XCD?
    bool all_local = true;
    {
        const unsigned xtag = a.epoch_base ? a.epoch_base : XCC_TAG;
        if (threadIdx.x == 0) granule_store(xtab + slice, xtag, __uint_as_float(my_xcc));
        unsigned peer = my_xcc;
        if (lane < NSLICE) {
            unsigned spins = 0;
            while (true) {
                const u64 gw = granule_load(xtab + lane);
                if ((unsigned)(gw >> 32) == xtag) { peer = (unsigned)gw; break; }
                if (++spins > spin_budget) { mp_set_error(a.err, 1000000); peer = ~0u; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        const unsigned long long same = __ballot(peer == my_xcc);
        all_local = (same & 0xffull) == 0xffull;
        src_local[0] = (same >> (2 * wave)) & 1;
        src_local[1] = (same >> (2 * wave + 1)) & 1;
        if (__ballot(peer == ~0u)) spin_budget = 0;
        if (a.force_remote) { all_local = false; src_local[0] = src_local[1] = false; }
    }
    __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hdL, 0, 4 * 16 * H * 4, 0x00020000);
    // h fetch of this wave: units 64 wave .. 64 wave + 63 of all 16 rows; instruction pg: lane (row = lane / 4, p = lane % 4)
    // takes 16 bytes at [row][64 wave + 16 pg + 4 p] -- pieces 0, 1 from producer slice 2 wave, pieces 2, 3 from 2 wave + 1
    const int hrow = lane >> 2, hp = lane & 3;
    unsigned hvoff[4];
#pragma unroll
    for (int pg = 0; pg < 4; ++pg)
        hvoff[pg] = (src_local[pg >> 1] ? 0u : HD_R) + (unsigned)((hrow * H + 64 * wave + 16 * pg + 4 * hp) * 4);
    const unsigned* hflag = hfL + (src_local[(lane >> 2) & 1] ? 0 : HF_R) + 8 * wave + (lane & 7);
    float* hdst = hb + hrow * HP + 64 * wave + 4 * hp;                  // + 16 pg

    // ---- initial h into LDS (this wave's quarter), x_0 into xb[0]
    {
        const int hbq = brow0 + hrow;
#pragma unroll
        for (int pg = 0; pg < 4; ++pg) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (hbq < B && !a.zero_state) v = *reinterpret_cast<const f32x4*>(d.hbuf + (size_t)hbq * H + 64 * wave + 16 * pg + 4 * hp);
            *reinterpret_cast<f32x4*>(hdst + 16 * pg) = v;
        }
    }
    // x rows of this wave: 4 wave .. 4 wave + 3; lane (xr = lane / 16, piece lane % 16 + 16 j) -- 256 contiguous bytes per row
    const int xr = 4 * wave + (lane >> 4);
    const int xbq = brow0 + xr;
    const bool xin = xbq < B;
    const int alen = xin ? a.lengths[xbq] : 0;
    const size_t xtstride = (size_t)B * KIN;
    const float* xp_cur = d.xin + (size_t)(xin ? xbq : 0) * KIN + (lane & 15) * 4 +
                          (size_t)(d.reverse ? (alen > 0 ? alen - 1 : 0) : 0) * xtstride;      // time index of `step`, clamped
    const float* xp_nxt = xp_cur;
    float* xdst = xb + xr * KP + (lane & 15) * 4;                       // + parity * 16 * KP + 64 j
    f32x4 xs[XLD];
#pragma unroll
    for (int j = 0; j < XLD; ++j) xs[j] = *reinterpret_cast<const f32x4*>(xp_cur + 64 * j);
#pragma unroll
    for (int j = 0; j < XLD; ++j) *reinterpret_cast<f32x4*>(xdst + 64 * j) = xs[j];

    long long pt[6] = {0, 0, 0, 0, 0, 0};
    const bool prof = PROF && a.prof != nullptr && threadIdx.x == 0;
#define NS_T(i) do { if (PROF && prof) pt[i] -= (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define NS_E(i) do { if (PROF && prof) pt[i] += (long long)__builtin_amdgcn_s_memtime(); } while (0)

    const float* xa = xb + r16 * KP + 4 * q;                            // A fragments: + parity * 16 * KP + 16 i
    const float* ha = hb + r16 * HP + 4 * q;                            //              + 16 i
    unsigned hflags = 0;
    constexpr int PUB_G = 1, REQ_G = NXG / 4, CHK_G = NXG / 2;

    for (int step = 0; step < T; ++step) {
        NS_T(0);
        barrier_lds_only();                                             // x_step (written during the previous step) is in xb
        xp_cur = xp_nxt;
        const unsigned epoch = a.epoch_base + (unsigned)step;           // tag of h_{step-1}
        const float* xf = xa + (step & 1) * 16 * KP;
        f32x4 acc[2];
        f32x4 h4[4];
        // ---- input projection: all of K_in in one accumulator per tile
        f32x4 a4 = *reinterpret_cast<const f32x4*>(xf), an = a4;
#pragma unroll
        for (int i = 0; i < NXG; ++i) {
            if (i + 1 < NXG) an = *reinterpret_cast<const f32x4*>(xf + 16 * (i + 1));
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int s = 4 * i + j;
                    if (s == 0) mfma_ns<true, true>(acc[t], a4[j], wx[s][t]);
                    else if (s < NXA) mfma_ns<false, true>(acc[t], a4[j], wx[s][t]);
                    else mfma_ns<false, false>(acc[t], a4[j], wx[s][t]);
                }
            __builtin_amdgcn_sched_barrier(0);
            a4 = an;
            if (i == PUB_G && step > 0) {
                // the values stored at the end of step-1 have had two groups of MFMAs to be acknowledged: raise the flag
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    unsigned* f = hfL + ((step + 1) & 1) * (NSLICE * 4) + slice * 4 + wave;
                    __hip_atomic_store(f, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (!all_local) __hip_atomic_store(f + HF_R, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (i == REQ_G) hflags = __hip_atomic_load(hflag + ((step + 1) & 1) * (NSLICE * 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (i == CHK_G - 1 && step > 0) {
                bool ok = hflags == epoch;
                unsigned spins = 0;
                if (PROF && prof && !__all(ok)) pt[5] += 1;
                while (!__all(ok)) {
                    if (++spins > spin_budget) {                        // bounded: flag the error and never wait again
                        if (lane == 0) mp_set_error(a.err, 1 + step);
                        spin_budget = 0;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    ok = __hip_atomic_load(hflag + ((step + 1) & 1) * (NSLICE * 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
                }
                const int par_off = ((step + 1) & 1) * (16 * H * 4);
#pragma unroll
                for (int pg = 0; pg < 4; ++pg)
                    h4[pg] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, hvoff[pg], par_off, 16 /* sc1 */));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        NS_E(0); NS_T(1);
        // ---- h_{step-1}: this wave's quarter into LDS, all four joined by the barrier; then the prefetch of x_{step+1}
        if (step > 0) {
#pragma unroll
            for (int pg = 0; pg < 4; ++pg) *reinterpret_cast<f32x4*>(hdst + 16 * pg) = h4[pg];
        }
        barrier_lds_only();
        {
            const bool adv = d.reverse ? (alen - 2 - step >= 0) : (step + 1 < T);
            const long dlt = d.reverse ? -(long)xtstride : (long)xtstride;
            xp_nxt = adv ? xp_cur + dlt : xp_cur;
        }
        NS_E(1); NS_T(2);
        // ---- recurrent part on top of the projection
        a4 = *reinterpret_cast<const f32x4*>(ha);
#pragma unroll
        for (int i = 0; i < NHG; ++i) {
            if (i + 1 < NHG) an = *reinterpret_cast<const f32x4*>(ha + 16 * (i + 1));
            // (the prefetch of x_{step+1}: one request per MFMA group -- back to back they stall the wave's issue)
            if (i < XLD) xs[i] = *reinterpret_cast<const f32x4*>(xp_nxt + 64 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) mfma_ns<false, true>(acc[t], a4[j], wh[4 * i + j][t]);
            __builtin_amdgcn_sched_barrier(0);
            a4 = an;
        }
        mfma_drain_ns();
        NS_E(2); NS_T(3);
        // ---- the four gates of a cell sit in four lanes (columns g*4 + u): wave-private transpose through LDS
        f32x4* tw = tb + (size_t)wave * 2 * 64;
        tw[lane] = acc[0];
        tw[64 + lane] = acc[1];
        f32x4 gate[2];
        {
            const float* tf = reinterpret_cast<const float*>(tw) + (q * 16 + u) * 4 + g;     // lane (q, g', u), element g
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int gg = 0; gg < 4; ++gg) gate[t][gg] = tf[t * 256 + gg * 16] + bias4[t][gg];
        }
        NS_E(3); NS_T(4);
        // ---- cell update (both cells first, then the stores), x_{step+1} into LDS, publish, layer output
        const bool act = step < clen;
        const int tt = act ? (d.reverse ? clen - 1 - step : step) : step;
        float oval[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const float ig = sigmoidf_(gate[t][0]);
            const float fg = sigmoidf_(gate[t][1]);
            const float gt = tanhf_(gate[t][2]);
            const float og = sigmoidf_(gate[t][3]);
            const float cnew = fg * cst[t] + ig * gt;
            const float hnew = og * tanhf_(cnew);
            cst[t] = act ? cnew : cst[t];
            hst[t] = act ? hnew : hst[t];
            oval[t] = act ? hnew : 0.f;
        }
        // (x_{step+1} is waited for and written while only loads are in flight -- see mp_lstm_fused)
        {
            float* xd = xdst + ((step + 1) & 1) * 16 * KP;
#pragma unroll
            for (int j = 0; j < XLD; ++j) *reinterpret_cast<f32x4*>(xd + 64 * j) = xs[j];
        }
        unsigned* hw = hdL + (size_t)(step & 1) * 16 * H + crow * H;
#pragma unroll
        for (int t = 0; t < 2; ++t)
            __hip_atomic_store(hw + junit[t], __float_as_uint(hst[t]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!all_local) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
                __hip_atomic_store(hw + HD_R / 4 + junit[t], __float_as_uint(hst[t]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (cin) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
                *reinterpret_cast<float*>(reinterpret_cast<char*>(outb[t]) + (size_t)(unsigned)tt * out_row_bytes) = oval[t];
        }
        NS_E(4);
    }
    if (PROF && prof) {
        long long* o = a.prof + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 5; ++i) o[i] = pt[i];
        o[5] = T;
        o[6] = pt[5];
        o[7] = (all_local ? 256 : 0) | my_xcc;
    }
    if (cin) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            d.hbuf[(size_t)cb * H + junit[t]] = hst[t];
            d.cbuf[(size_t)cb * H + junit[t]] = cst[t];
        }
    }
}

// W [4H][K] (rows gate * H + unit) -> [slice][wave][k-step][tile][lane]: column lane & 15 = gate (lane & 15) / 4 of unit
// slice*32 + wave*8 + tile*4 + lane % 4; k = 16 (s / 4) + 4 (lane / 16) + s % 4
MP_KERNEL void mp_pack_w_ns(const float* __restrict__ w, float* __restrict__ dst, int K) {
    constexpr int H = 256;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)4 * H * K) return;
    const int NS = K / 4;
    size_t rest = idx;
    const int lane = (int)(rest % 64); rest /= 64;
    const int t = (int)(rest % 2); rest /= 2;
    const int s = (int)(rest % NS); rest /= NS;
    const int wave = (int)(rest % 4); rest /= 4;
    const int slice = (int)rest;
    const int c16 = lane & 15, qq = lane >> 4;
    const int row = (c16 >> 2) * H + slice * 32 + wave * 8 + t * 4 + (c16 & 3);
    const int k = 16 * (s >> 2) + 4 * qq + (s & 3);
    dst[idx] = w[(size_t)row * K + k];
}

template <int KIN>
void launch_ns(const LstmPersistArgs& a, hipStream_t s) {
    LstmPersistArgs b = a;
    int most = 0, total = 0;
    for (int x = 0; x < 8; ++x) { most = b.xcd_cnt[x] > most ? b.xcd_cnt[x] : most; total += b.xcd_cnt[x]; }
    if (total != a.nslab * a.ndir) {
        mp_fill_xcd_table(b, nullptr);
        most = (a.nslab * a.ndir + 7) / 8;
    }
    size_t lds = NsCfg<KIN>::LDS_BYTES;
    if ((size_t)a.min_lds > lds) lds = (size_t)a.min_lds;
    const dim3 grid(8 * most * 8);
    if (a.prof) hipLaunchKernelGGL((mp_lstm_ns<KIN, true>), grid, dim3(256), lds, s, b);
    else hipLaunchKernelGGL((mp_lstm_ns<KIN, false>), grid, dim3(256), lds, s, b);
}

template <int KIN>
hipError_t ns_attrs() {
    const int lds = (int)NsCfg<KIN>::LDS_BYTES;
    hipError_t e = hipFuncSetAttribute((const void*)mp_lstm_ns<KIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)mp_lstm_ns<KIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace

void mp_launch_lstm_ns(const LstmPersistArgs& a, int KIN, hipStream_t s) {
    if (KIN == 256) launch_ns<256>(a, s);
    else launch_ns<512>(a, s);
}
void mp_launch_pack_w_ns(const float* w, float* dst, int K, hipStream_t s) {
    const size_t n = (size_t)4 * 256 * K;
    hipLaunchKernelGGL(mp_pack_w_ns, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, dst, K);
}
hipError_t mp_lstm_ns_device_attrs() {
    hipError_t e = ns_attrs<256>();
    if (!e) e = ns_attrs<512>();
    return e;
}

// K2s -- the persistent fp32 nn.LSTM layer (models/rnn.py:27) for batches of at most a few slabs: H = 256 on 32 slices of 8
// hidden units per slab.
//
// A recurrence step of a small batch is a latency chain -- `evaluate.py`'s own call is ONE sequence of 3000 frames -- and a
// slab of 16 sequences cannot use more than the CUs its slices run on.  mp_lstm_fused offers 8 or 16 slices per slab (tiles of
// one gate x 16 units: a workgroup needs at least 16 units); here a tile's 16 columns are 4 gates x 4 units, so a workgroup
// owns 8 units (2 tiles), a slab spreads over 32 workgroups -- one whole XCD, the most that can still share an L2 -- and
// the matrix work per workgroup and step is half that of the 16-slice kernels (64 / 96 instead of 128 / 192 MFMAs per wave).
//   * 4 waves = 4 K quarters of both tiles; weights in VGPRs (96 per lane at K_in = 512);
//   * A operand straight from global memory: x_t prefetched one step ahead (16-byte pieces of the lane's row, k = 16 i + 4 q + j
//     so that four consecutive k-steps are one piece), h_{t-1} through the flagged hand-off of mp_lstm_fused ("FLAGX": plain
//     words, here row-major [row][unit], + one flag per producer wave) -- a consumer wave watches the 32 flags of the 8 slices
//     its K quarter comes from;
//   * the 4 K partials meet in LDS (8 KB), and because the four gates of a cell sit in four LANES of a tile, the same LDS pass
//     transposes them: finishing wave w takes rows 4 q + w, lane (q, t, u) the cell of unit 8 slice + 4 t + u.
// Same packed-sequence semantics, transports, bounded waits / error word and XCD table as mp_lstm_fused.  Sums run over k in
// another order than in the 8 / 16-slice kernels: equal to fp32 rounding, not bitwise
// (tests/test_gpu_parity.py::test_32_slice_fp32_kernel_matches_16_slice).
#include "mp_lstm_dev.h"

namespace {

template <int KIN>
struct U8Cfg {
    static constexpr int H = 256, NSLICE = 32, U = 8;
    static constexpr int KQ = KIN / 4;                        // x: K range of one wave
    static constexpr int NXS = KQ / 4, NXG = KQ / 16;         //    k-steps, groups of 4 k-steps (one 16-byte piece)
    static constexpr int NHS = 16, NHG = 4;                   // h: 64 units per wave
    static constexpr size_t LDS_BYTES = (size_t)4 * 2 * 64 * 16;
    // exchange area (words from its start): values L / R [2 parities][16 * H] each, flags L / R [2][NSLICE * 4] each, XCC table
    static constexpr unsigned HD_R = 2 * 16 * H * 4;          // byte offset of the R values
    static constexpr unsigned HF0 = 4 * 16 * H;               // word offset of the flags
    static constexpr unsigned HF_R = 2 * NSLICE * 4;          // word offset of the R flags behind the L flags
    static constexpr unsigned XT0 = (HF0 + 2 * HF_R + 512) / 2;   // u64 offset of the XCC table (32 granules)
};

template <int KIN, bool PROF>
MP_KERNEL __launch_bounds__(256, 1) void mp_lstm_u8(LstmPersistArgs a) {
    // test hook (mp_debug_drop_workgroup): a workgroup that never shows up.  Only in the PROF instantiation, which the launcher
    // picks when the hook is armed -- the product kernels carry no test code (round 4)
    if (PROF && a.debug_drop && (int)blockIdx.x == a.debug_drop - 1) return;
    using C = U8Cfg<KIN>;
    constexpr int H = C::H, NSLICE = C::NSLICE, U = C::U, KQ = C::KQ, NXS = C::NXS, NXG = C::NXG, NHS = C::NHS, NHG = C::NHG;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);                        // [source kq][tile][lane]

    // ---- cluster (direction, slab) and slice: host table by XCD, or round robin (see mp_lstm_fused)
    const int ncl = a.ndir * a.nslab;
    const int xcd = a.xcd_physical ? (int)(xcc_id() & 7) : (int)(blockIdx.x & 7);
    const int kth = (int)(blockIdx.x >> 3) / NSLICE;
    const int slice = (int)(blockIdx.x >> 3) % NSLICE;
    if (kth >= mp_xcd_count(a, xcd)) return;
    const int cl = mp_xcd_first(a, xcd) + kth;
    if (cl >= ncl) return;
    const int dir = cl / a.nslab, slab = cl % a.nslab;
    const LstmDir d = a.d[dir];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;        // wave = K quarter = finishing group
    const int r16 = lane & 15, q = lane >> 4;                           // A operand: row r16, k = 16 i + 4 q + j
    const int g = (lane >> 2) & 3, u = lane & 3;                        // D column lane & 15 = gate g, unit u of the tile
    const int B = a.B, T = a.T;
    const int brow0 = (a.slab0 + slab) * 16;

    // ---- weights: [slice][wave][k-step][tile][lane]
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

    // ---- the cell of this lane (lanes with g < 2): sequence row 4 q + wave, unit slice*8 + 4 t + u, t = g
    const bool cell = g < 2;
    const int ct = g & 1;
    const int crow = 4 * q + wave;
    const int cb = brow0 + crow;
    const bool cin = cell && cb < B;
    const int clen = cin ? a.lengths[cb] : 0;
    const int junit = slice * U + ct * 4 + u;
    const f32x4 bias4 = *reinterpret_cast<const f32x4*>(d.bias + 4 * junit);
    float cst = (cin && !a.zero_state) ? d.cbuf[(size_t)cb * H + junit] : 0.f;
    float hst = (cin && !a.zero_state) ? d.hbuf[(size_t)cb * H + junit] : 0.f;
    float* outb = d.out + (size_t)(cin ? cb : 0) * d.outStride + junit;
    const unsigned out_row_bytes = (unsigned)B * (unsigned)d.outStride * 4u;

    // ---- exchange area, XCC table, transports
    constexpr size_t SLABW = (size_t)4 * 16 * H + 16;
    u64* hx0 = a.hx + (size_t)cl * SLABW;
    unsigned* hdL = reinterpret_cast<unsigned*>(hx0);
    unsigned* hfL = hdL + C::HF0;
    u64* xtab = hx0 + C::XT0;
    unsigned spin_budget = a.max_spin;
    const unsigned my_xcc = xcc_id();
    unsigned long long same = ~0ull;
    bool all_local = true;
    {
        const unsigned xtag = a.epoch_base ? a.epoch_base : XCC_TAG;
        if (threadIdx.x == 0) granule_store(xtab + slice, xtag, __uint_as_float(my_xcc));
        unsigned peer = my_xcc;
        if (lane < NSLICE) {
            unsigned spins = 0; u64 wt0 = 0;
            while (true) {
                const u64 gw = granule_load(xtab + lane);
                if ((unsigned)(gw >> 32) == xtag) { peer = (unsigned)gw; break; }
                if (wait_over(spins, spin_budget, wt0, a.max_ticks)) { mp_set_error(a.err, 1000000); peer = ~0u; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        same = __ballot(peer == my_xcc) | ~0xffffffffull;              // bit s: producer slice s is on my XCD
        all_local = (same & 0xffffffffull) == 0xffffffffull;
        if (__ballot(peer == ~0u)) { spin_budget = 0; poison_cells(cst); }
        if (a.force_remote) { all_local = false; same = 0; }
    }
    __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(hdL, 0, 4 * 16 * H * 4, 0x00020000);
    // h pieces of this lane: [row r16][64 wave + 16 i + 4 q .. + 3], i = 0..3: producer slice 8 wave + 2 i + q / 2
    unsigned hvoff[NHG];
#pragma unroll
    for (int i = 0; i < NHG; ++i) {
        const int ps = 8 * wave + 2 * i + (q >> 1);
        hvoff[i] = (((same >> ps) & 1) ? 0u : C::HD_R) + (unsigned)((r16 * H + 64 * wave + 16 * i + 4 * q) * 4);
    }
    // flags this wave watches: the 4 finishing waves of slices 8 wave .. 8 wave + 7 (lane & 31 -> slice 8 wave + (lane & 31) / 4)
    const unsigned* hflag = hfL + (((same >> (8 * wave + ((lane & 31) >> 2))) & 1) ? 0 : C::HF_R) + 32 * wave + (lane & 31);

    // ---- A operand of the recurrent part for step 0: the initial state
    f32x4 hr[NHG];
    {
        const int ab = brow0 + r16;
#pragma unroll
        for (int i = 0; i < NHG; ++i) {
            hr[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ab < B && !a.zero_state) hr[i] = *reinterpret_cast<const f32x4*>(d.hin + (size_t)ab * H + 64 * wave + 16 * i + 4 * q);
        }
    }
    // ---- x: this lane's row r16, K quarter `wave`: pieces [KQ wave + 16 i + 4 q .. + 3]
    const int arow = brow0 + r16;
    const bool arow_in = arow < B;
    const int alen = arow_in ? a.lengths[arow] : 0;
    const size_t xtstride = (size_t)B * KIN;
    const float* xp_cur = d.xin + (size_t)(arow_in ? arow : 0) * KIN + wave * KQ + 4 * q +
                          (size_t)(d.reverse ? (alen > 0 ? alen - 1 : 0) : 0) * xtstride;      // time index of `step`, clamped
    const float* xp_nxt = xp_cur;
    f32x4 xa[NXG];
#pragma unroll
    for (int i = 0; i < NXG; ++i) xa[i] = *reinterpret_cast<const f32x4*>(xp_cur + 16 * i);
#pragma unroll
    for (int i = 0; i < NXG; ++i) asm volatile("" : "+v"(xa[i]));       // (x_0 has arrived before the loop: see mp_lstm_fused)

    long long pt[6] = {0, 0, 0, 0, 0, 0};
    const bool prof = PROF && a.prof != nullptr && threadIdx.x == 0;
#define U8_T(i) do { if (PROF && prof) pt[i] -= (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define U8_E(i) do { if (PROF && prof) pt[i] += (long long)__builtin_amdgcn_s_memtime(); } while (0)
    unsigned hflags = 0;
    constexpr int CHK_S = NXS / 2, REQ_S = NXS / 4 > 0 ? NXS / 4 : 1;

    for (int step = 0; step < T; ++step) {
        U8_T(0);
        xp_cur = xp_nxt;
        const unsigned epoch = a.epoch_base + (unsigned)step;           // tag of h_{step-1}
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        // ---- input projection (this wave's K quarter); half way: flag check, then the h words are requested
#pragma unroll
        for (int s = 0; s < NXS; ++s) {
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[s >> 2][s & 3], wx[s][t], acc[t], 0, 0, 0);
            if (s == REQ_S) hflags = __hip_atomic_load(hflag + ((step + 1) & 1) * (NSLICE * 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (s == CHK_S - 1) {
#pragma unroll
                for (int i = 0; i < NXG; ++i) asm volatile("" : "+v"(xa[i]));   // (all of x_t waited for before h is requested)
                if (step > 0) {
                    bool ok = hflags == epoch;
                    unsigned spins = 0; u64 wt0 = 0;
                    if (PROF && prof && !__all(ok)) pt[5] += 1;
                    while (!__all(ok)) {
                        if (wait_over(spins, spin_budget, wt0, a.max_ticks)) {                    // bounded: flag the error and never wait again
                            if (lane == 0) mp_set_error(a.err, 1 + step);
                            spin_budget = 0; poison_cells(cst);
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        ok = __hip_atomic_load(hflag + ((step + 1) & 1) * (NSLICE * 4), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch;
                    }
                    const int par_off = ((step + 1) & 1) * (16 * H * 4);
#pragma unroll
                    for (int i = 0; i < NHG; ++i)
                        hr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(hrs, hvoff[i], par_off, 16 /* sc1 */));
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        U8_E(0); U8_T(1);
#pragma unroll
        for (int i = 0; i < NHG; ++i) asm volatile("" : "+v"(hr[i]));   // (h waited for before the prefetch of x_{step+1} is issued)
        {
            const bool adv = d.reverse ? (alen - 2 - step >= 0) : (step + 1 < T);
            const long dlt = d.reverse ? -(long)xtstride : (long)xtstride;
            xp_nxt = adv ? xp_cur + dlt : xp_cur;
#pragma unroll
            for (int i = 0; i < NXG; ++i) xa[i] = *reinterpret_cast<const f32x4*>(xp_nxt + 16 * i);
        }
        U8_E(1); U8_T(2);
        // ---- recurrent part
#pragma unroll
        for (int s = 0; s < NHS; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(hr[s >> 2][s & 3], wh[s][t], acc[t], 0, 0, 0);
        U8_E(2); U8_T(3);
        // ---- K reduction + gate transpose through LDS
        barrier_lds_only();                                             // previous step's reads of `red` are done
        red[(wave * 2 + 0) * 64 + lane] = acc[0];
        red[(wave * 2 + 1) * 64 + lane] = acc[1];
        barrier_lds_only();                                             // (LDS only: the prefetch of x_{step+1} stays in flight)
        f32x4 gate;
        {
            const float* rf = reinterpret_cast<const float*>(red) + ((ct * 64 + q * 16 + u) << 2) + wave;   // + kq * 512 + gate * 16
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
                float v = rf[gg * 16];
#pragma unroll
                for (int kq = 1; kq < 4; ++kq) v += rf[kq * 512 + gg * 16];
                gate[gg] = v + bias4[gg];
            }
        }
        U8_E(3); U8_T(4);
        // ---- cell update, publish, layer output (x_{step+1} waited for while only loads are in flight)
        const bool act = step < clen;
        const int tt = act ? (d.reverse ? clen - 1 - step : step) : step;
        const float ig = sigmoidf_(gate[0]);
        const float fg = sigmoidf_(gate[1]);
        const float gt = tanhf_(gate[2]);
        const float og = sigmoidf_(gate[3]);
        const float cnew = fg * cst + ig * gt;
        const float hnew = og * tanhf_(cnew);
        cst = act ? cnew : cst;
        hst = act ? hnew : hst;
        const float oval = act ? hnew : 0.f;
#pragma unroll
        for (int i = 0; i < NXG; ++i) asm volatile("" : "+v"(xa[i]));
        if (cell) {
            unsigned* hw = hdL + (size_t)(step & 1) * 16 * H + crow * H + junit;
            __hip_atomic_store(hw, __float_as_uint(hst), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!all_local) __hip_atomic_store(hw + C::HD_R / 4, __float_as_uint(hst), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // The hand-off IS the critical path of this kernel (its projection is too short to hide anything), so the flag follows
        // its values at once: wait for their acknowledgement, raise the flag, and only then write the layer output.
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            unsigned* f = hfL + (step & 1) * (NSLICE * 4) + slice * 4 + wave;
            __hip_atomic_store(f, a.epoch_base + (unsigned)step + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (!all_local) __hip_atomic_store(f + C::HF_R, a.epoch_base + (unsigned)step + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (cin) *reinterpret_cast<float*>(reinterpret_cast<char*>(outb) + (size_t)(unsigned)tt * out_row_bytes) = oval;
        U8_E(4);
    }
    if (PROF && prof) {
        long long* o = a.prof + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 5; ++i) o[i] = pt[i];
        o[5] = T;
        o[6] = pt[5];
        o[7] = (all_local ? 256 : 0) | my_xcc;
    }
    if (cin) {
        d.hbuf[(size_t)cb * H + junit] = hst;
        d.cbuf[(size_t)cb * H + junit] = cst;
    }
}

// W [4H][K] (rows gate * H + unit) -> [slice][wave][k-step][tile][lane]: column lane & 15 = gate (lane & 15) / 4 of unit
// slice*8 + tile*4 + lane % 4; k = wave * K/4 + 16 (s / 4) + 4 (lane / 16) + s % 4
MP_KERNEL void mp_pack_w_u8(const float* __restrict__ w, float* __restrict__ dst, int K) {
    constexpr int H = 256;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)4 * H * K) return;
    const int NS = K / 16;                                              // k-steps per wave
    size_t rest = idx;
    const int lane = (int)(rest % 64); rest /= 64;
    const int t = (int)(rest % 2); rest /= 2;
    const int s = (int)(rest % NS); rest /= NS;
    const int wave = (int)(rest % 4); rest /= 4;
    const int slice = (int)rest;
    const int c16 = lane & 15, qq = lane >> 4;
    const int row = (c16 >> 2) * H + slice * 8 + t * 4 + (c16 & 3);
    const int k = wave * (K / 4) + 16 * (s >> 2) + 4 * qq + (s & 3);
    dst[idx] = w[(size_t)row * K + k];
}

template <int KIN>
void launch_u8(const LstmPersistArgs& a, hipStream_t s) {
    LstmPersistArgs b = a;
    int most = 0, total = 0;
    for (int x = 0; x < 8; ++x) { most = b.xcd_cnt[x] > most ? b.xcd_cnt[x] : most; total += b.xcd_cnt[x]; }
    if (total != a.nslab * a.ndir) {
        mp_fill_xcd_table(b, nullptr);
        most = (a.nslab * a.ndir + 7) / 8;
    }
    size_t lds = U8Cfg<KIN>::LDS_BYTES;
    if ((size_t)a.min_lds > lds) lds = (size_t)a.min_lds;
    const dim3 grid(8 * most * 32);
    if (a.prof || a.debug_drop) hipLaunchKernelGGL((mp_lstm_u8<KIN, true>), grid, dim3(256), lds, s, b);
    else hipLaunchKernelGGL((mp_lstm_u8<KIN, false>), grid, dim3(256), lds, s, b);
}

template <int KIN>
hipError_t u8_attrs() {
    const int lds = 96 * 1024;                                          // (room for LstmPersistArgs::min_lds)
    hipError_t e = hipFuncSetAttribute((const void*)mp_lstm_u8<KIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)mp_lstm_u8<KIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace

void mp_launch_lstm_u8(const LstmPersistArgs& a, int KIN, hipStream_t s) {
    if (KIN == 256) launch_u8<256>(a, s);
    else launch_u8<512>(a, s);
}
void mp_launch_pack_w_u8(const float* w, float* dst, int K, hipStream_t s) {
    const size_t n = (size_t)4 * 256 * K;
    hipLaunchKernelGGL(mp_pack_w_u8, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, dst, K);
}
hipError_t mp_lstm_u8_device_attrs() {
    hipError_t e = u8_attrs<256>();
    if (!e) e = u8_attrs<512>();
    return e;
}

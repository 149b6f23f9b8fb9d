// K2v -- the persistent fp32 nn.LSTM layer (models/rnn.py:27) for ONE sequence (H = 256): `evaluate.py`'s own calls -- one
// sequence of thousands of frames offline (evaluate.py:57-60), one 45-frame window per forward_online call (evaluate.py:62-64,
// net.py:144-171) -- and the serial chain of `mp_stream_replay`.
//
// At B = 1 a recurrence step is a matrix-VECTOR product and a hand-off; the MFMA kernels spend 64 MFMAs (2 048 cycles per wave,
// 15 of every tile's 16 rows empty) on it, which is most of mp_lstm_u8's 1.3 us step.  Here the product runs on the vector ALU:
//   * a (direction, sequence) cluster is 32 workgroups of 4 waves on one XCD, as in mp_lstm_u8; every WAVE owns 2 hidden units
//     (8 gate rows) and needs nobody else in its workgroup -- no LDS shared between waves, no barrier anywhere in the kernel;
//   * lane (s, r) = (lane / 8, lane % 8): gate row r = 4 * (unit of the wave) + gate, K segment s -- K_in / 8 weights of W_ih and
//     32 of W_hh in registers (64 / 96 VGPRs), the 8 partial sums of a row meet through DPP row_ror:8 and the gfx950
//     v_permlane16_swap / v_permlane32_swap (three additions, no LDS);
//   * x_t and h_{t-1} are the same for every row: each wave loads the vector once (one or two 16-byte loads per lane), drops
//     it into its own strip of LDS and every lane reads back its segment as broadcast 16-byte reads (segment strips padded by
//     16 bytes so that the 8 segments of a wave hit different banks);
//   * h_t travels as granules {epoch, value} (8-byte atomic stores, mp_lstm_dev.h) -- one L2 round trip per step, no tag bit,
//     no flags: a wave polls ALL 256 granules of the parity slot with two 16-byte loads per lane;
//   * the four gates of a unit sit in the four lanes of a quad: every lane applies ITS gate's non-linearity (one exp2 + one rcp
//     for all four, by per-lane constants -- the expressions of sigmoidf_ / tanhf_, so the values are theirs bit for bit), the quad
//     exchanges them with DPP quad_perm, and every lane of the quad keeps the unit's cell state.
// Per step: about 130 cycles of FMAs behind the arrival of h_{t-1}, 60 of reduction, 200 of cell, against 2 048 + 1 000 in
// mp_lstm_u8; the rest of the step is the hand-off through L2 (330-380 ns one way, tools/micro/pingpong.hip).
// WF (the unidirectional block): both layers in one launch, cluster 2 seq + layer, layer 1 reading layer 0's output row t behind
// per-wave progress words (the mechanism of mp_lstm_fused<...,WF>).  The host puts the two clusters of a sequence on ONE XCD --
// two workgroups per CU: these waves wait most of the time and use a quarter of the register file -- so the link lives in that
// XCD's L2: plain output stores, workgroup-scope progress words, nothing written through (with the layers on two XCDs layer 0's
// step doubled: its polls had to sit out the write-through acknowledgements of its own output stores).  Layer 1 reads the
// progress words only when what it last saw does not cover the row it needs, and then waits for a lead of kLead steps.  The two
// roles run two instantiations of the step loop (ROLE): with one loop and run-time role flags the compiler's wait-count pass
// put a vmcnt(0) between layer 0's progress store and its next x request.
// Same packed-sequence semantics, transports (L: this XCD's L2; R: written through, any placement), bounded waits / error
// word and XCD table as the other persistent kernels.  Another order of summation than theirs: equal to fp32 rounding, not
// bitwise (tests/test_gpu_round5.py::test_single_sequence_kernel_*).
#include "mp_lstm_dev.h"

namespace {

// mp_set_error for the step loop: the same system-scope store, as inline asm (a store the compiler sees anywhere in the loop --
// even on this cold path -- makes its wait-count pass drain every pending load at the next wait: store_granule_xcd, mp_lstm_dev.h)
static __device__ __forceinline__ void set_error_hidden(int* err, int code) {
    asm volatile("global_store_dword %0, %1, off sc0 sc1" :: "v"(err), "v"(code) : "memory");
}

template <int KIN>
struct V1Cfg {
    static constexpr int H = 256, NSLICE = 32;
    static constexpr int KS = KIN / 8;                          // x values per K segment
    static constexpr int XSTR = KS + 4, HSTR = 32 + 4;          // padded segment strips (floats)
    static constexpr int WAVE_FLOATS = 8 * XSTR + 8 * HSTR;     // LDS per wave
    static constexpr size_t LDS_BYTES = (size_t)4 * WAVE_FLOATS * 4;
    static constexpr int NX = KIN / 256;                        // 16-byte pieces of x_t per lane
    // exchange area of a cluster (u64 from its start): granules L [2 parities][H], R [2][H]; XCC table and WF progress words
    // where mp_lstm_u8 keeps them
    static constexpr unsigned G_R = 2 * H;
    static constexpr unsigned XT0 = 8704;                       // = U8Cfg::XT0
    static constexpr unsigned PL0 = 2 * XT0 + 128;              // word offset of the progress words (NSLICE * 4)
};

static __device__ __forceinline__ float dpp_quad(float v, int lane_of_quad) {
    const int x = __float_as_int(v);
    int r;
    switch (lane_of_quad) {
        case 0: r = __builtin_amdgcn_update_dpp(x, x, 0x00, 0xf, 0xf, false); break;
        case 1: r = __builtin_amdgcn_update_dpp(x, x, 0x55, 0xf, 0xf, false); break;
        case 2: r = __builtin_amdgcn_update_dpp(x, x, 0xAA, 0xf, 0xf, false); break;
        default: r = __builtin_amdgcn_update_dpp(x, x, 0xFF, 0xf, 0xf, false); break;
    }
    return __int_as_float(r);
}
// sum over the 8 lanes {lane % 8 + 8 s}: every lane ends up with the total (additions commute: the same bits in all eight)
static __device__ __forceinline__ float sum_over_segments(float v) {
    const int x = __float_as_int(v);
    v += __int_as_float(__builtin_amdgcn_update_dpp(x, x, 0x128 /* row_ror:8 */, 0xf, 0xf, false));
    {
        const unsigned u = __float_as_uint(v);
        const auto p = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    }
    {
        const unsigned u = __float_as_uint(v);
        const auto p = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        v = __uint_as_float(p[0]) + __uint_as_float(p[1]);
    }
    return v;
}

constexpr int kLead = 16;      // WF: a layer-1 wave that has to wait for layer 0 waits for this many steps at once

// ROLE: 0 = a layer of its own, 1 = layer 0 of a wavefront launch, 2 = layer 1 of one
template <int KIN, bool PROF, int ROLE>
static __device__ __forceinline__ void v1_cluster(const LstmPersistArgs& a, float* smem, int cl, int dir, int seq, int slice) {
    using C = V1Cfg<KIN>;
    constexpr int H = C::H, NSLICE = C::NSLICE, KS = C::KS, XSTR = C::XSTR, HSTR = C::HSTR, NX = C::NX;
    constexpr bool wf_l0 = ROLE == 1, wf_l1 = ROLE == 2;
    const LstmDir d = a.d[dir];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane >> 3, g = lane & 3;                              // K segment; gate of the quad's unit
    const int junit = slice * 8 + wave * 2 + ((lane >> 2) & 1);
    const int B = a.B, T = a.T;
    const int b = a.slab0 + seq;                                        // (a "slab" of this kernel is one sequence)
    const bool in = b < B;
    const int len = in ? a.lengths[b] : 0;

    // ---- weights of gate row g * H + junit, segment s: packed [slice][wave][16-byte piece][lane] (mp_pack_w_v1), so that a
    // wave's load is one contiguous KB (straight from the row-major matrices every lane walked a 128-byte line of its own, 64
    // lines per load and 4 waves per CU: the prologue was 8 / 17 us of a 45-step launch)
    float wx[KS], wh[32];
    {
        const f32x4* px = reinterpret_cast<const f32x4*>(d.wihpack) + (size_t)(slice * 4 + wave) * (KS / 4) * 64 + lane;
#pragma unroll
        for (int i = 0; i < KS / 4; ++i) {
            const f32x4 v = px[(size_t)i * 64];
            wx[4 * i] = v[0]; wx[4 * i + 1] = v[1]; wx[4 * i + 2] = v[2]; wx[4 * i + 3] = v[3];
        }
        const f32x4* ph = reinterpret_cast<const f32x4*>(d.wpack) + (size_t)(slice * 4 + wave) * 8 * 64 + lane;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 v = ph[(size_t)i * 64];
            wh[4 * i] = v[0]; wh[4 * i + 1] = v[1]; wh[4 * i + 2] = v[2]; wh[4 * i + 3] = v[3];
        }
    }
    const float bias = d.bias[4 * junit + g];
    const bool isg = g == 2;                                            // the cell-input gate: tanh; the others: sigmoid
    const float cx = isg ? 2.8853900817779268f : -1.4426950408889634f;
    float cst = (in && !a.zero_state) ? d.cbuf[(size_t)b * H + junit] : 0.f;
    float hst = (in && !a.zero_state) ? d.hbuf[(size_t)b * H + junit] : 0.f;
    const bool owner = in && s == 0 && g == 0;                          // the lane that publishes / stores its unit
    float* outb = d.out + (size_t)(in ? b : 0) * d.outStride + junit;
    const unsigned out_row_bytes = (unsigned)B * (unsigned)d.outStride * 4u;

    // ---- exchange area, XCC table, transport
    constexpr size_t SLABW = (size_t)4 * 16 * H + 16;
    u64* hx0 = a.hx + (size_t)cl * SLABW;
    u64* xtab = hx0 + C::XT0;
    unsigned spin_budget = a.max_spin;
    const unsigned my_xcc = xcc_id();
    bool all_local = true, link_local = true;
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
        all_local = (__ballot(peer == my_xcc) & 0xffffffffull) == 0xffffffffull;
        if (__ballot(peer == ~0u)) { spin_budget = 0; poison_cells(cst); }
        if (ROLE != 0) {            // the other layer's cluster: on my XCD too?
            const u64* ptab = a.hx + (size_t)(cl ^ 1) * SLABW + C::XT0;
            unsigned other = my_xcc;
            if (lane < NSLICE && spin_budget) {
                unsigned spins = 0; u64 wt0 = 0;
                while (true) {
                    const u64 gw = granule_load(ptab + lane);
                    if ((unsigned)(gw >> 32) == xtag) { other = (unsigned)gw; break; }
                    if (wait_over(spins, spin_budget, wt0, a.max_ticks)) { mp_set_error(a.err, 1000000); other = ~0u; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            link_local = all_local && (__ballot(other == my_xcc) & 0xffffffffull) == 0xffffffffull;
            if (__ballot(other == ~0u)) { spin_budget = 0; poison_cells(cst); }
        }
        if (a.force_remote) { all_local = false; link_local = false; }
    }
    __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(hx0, 0, 4 * H * 8, 0x00020000);
    const unsigned gvoff = (all_local ? 0u : C::G_R * 8u) + 16u * (unsigned)lane;      // granules 2 lane, 2 lane + 1 (and + 128)
    u64* gown = hx0 + junit;

    // ---- WF: progress words of layer 0 (in the area of cluster cl & ~1): word [slice * 4 + wave] = epoch_base + visible steps
    unsigned* plink = reinterpret_cast<unsigned*>(a.hx + (size_t)(cl & ~1) * SLABW) + C::PL0;
    const unsigned pbase = a.epoch_base;
    unsigned link_waits = 0;
    int seen = 0;                                                       // layer 0 has made at least this many steps visible
    // make sure layer 0's outputs of the first `need` steps are visible (a word of this launch is pbase + n, n <= T: one
    // unsigned window compare per word, see mp_lstm_fused); bounded like every wait
    auto link_need = [&](int need, int step) {
        if (seen >= need) return;
        const int target = need + kLead < T ? need + kLead : T;
        unsigned spins = 0; u64 wt0 = 0;
        while (true) {
            const unsigned p0 = __hip_atomic_load(plink + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned p1 = __hip_atomic_load(plink + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            auto reached = [&](int n) {
                const unsigned span = (unsigned)(T - n);
                return __all(p0 - pbase - (unsigned)n <= span && p1 - pbase - (unsigned)n <= span) != 0;
            };
            if (reached(target)) { seen = target; return; }
            // first look: layer 0 is ahead, by less than the lead -- go on; otherwise this wave has caught up with layer 0: let it
            // get kLead steps ahead, then run that many steps without looking at the words again
            if (spins == 0 && reached(need)) { seen = need; return; }
            if (PROF && spins == 0) ++link_waits;
            if (wait_over(spins, spin_budget, wt0, a.max_ticks)) {
                if (lane == 0) set_error_hidden(a.err, 1 + step);
                spin_budget = 0; poison_cells(cst);
                seen = T;
                return;
            }
            __builtin_amdgcn_s_sleep(4);
        }
    };
    auto link_publish = [&](int done) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's output stores of the steps before are acknowledged
        if (lane == 0) {
            if (link_local) store_word_xcd(plink + slice * 4 + wave, pbase + (unsigned)done);
            else store_word_dev(plink + slice * 4 + wave, pbase + (unsigned)done);
        }
    };

    // ---- this wave's LDS strips
    float* xs = smem + wave * C::WAVE_FLOATS;
    float* hs = xs + 8 * XSTR;
    const float* xseg = xs + s * XSTR;
    const float* hseg = hs + s * HSTR;
    {   // the initial state as h_{-1}: values 4 lane .. 4 lane + 3
        f32x4 h0 = {0.f, 0.f, 0.f, 0.f};
        if (in && !a.zero_state) h0 = *reinterpret_cast<const f32x4*>(d.hin + (size_t)b * H + 4 * lane);
        *reinterpret_cast<f32x4*>(hs + (lane >> 3) * HSTR + 4 * (lane & 7)) = h0;
    }
    // ---- x: time row of step `st` (reverse direction: from the sequence's last frame down; padding steps re-read a valid row)
    const size_t xtstride = (size_t)B * KIN;
    auto time_of = [&](int st) {
        int t = d.reverse ? len - 1 - st : st;
        t = t < 0 ? 0 : t;
        return t < T ? t : T - 1;
    };
    f32x4 xr[NX];
    auto load_x = [&](int t) {
        const float* row = d.xin + (size_t)t * xtstride + (size_t)(in ? b : 0) * KIN;
        if (wf_l1) {      // layer 0's output, written by another XCD: sc1, one buffer resource per time row
            __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(row), 0, KIN * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < NX; ++i)
                xr[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, 1024u * i + 16u * lane, 0, 16 /* sc1 */));
        } else {
#pragma unroll
            for (int i = 0; i < NX; ++i) xr[i] = *reinterpret_cast<const f32x4*>(row + 256 * i + 4 * lane);
        }
    };
    auto stage_x = [&]() {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int k = 256 * i + 4 * lane;
            *reinterpret_cast<f32x4*>(xs + (k / KS) * XSTR + (k % KS)) = xr[i];
        }
    };
    if (wf_l1) link_need(1, 0);
    load_x(time_of(0));

    long long pt[5] = {0, 0, 0, 0, 0};
    long long polls = 0;
    const bool prof = PROF && a.prof != nullptr && threadIdx.x == 0;
#define V1_T(i) do { if (PROF && prof) pt[i] -= (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define V1_E(i) do { if (PROF && prof) pt[i] += (long long)__builtin_amdgcn_s_memtime(); } while (0)

    for (int step = 0; step < T; ++step) {
        V1_T(0);
        // ---- input projection: x_t (requested a step ago) through this wave's LDS strip
        stage_x();
        asm volatile("" ::: "memory");
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < KS / 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xseg + 4 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(wx[4 * i + j], v[j], acc[j]);
        }
        V1_E(0); V1_T(1);
        // ---- h_{t-1}: all 256 granules of the parity slot, epoch = epoch_base + step.  (The first poll BEHIND the input
        // projection: issued in front of it -- it then returns while x_t is consumed -- the replay chain took 166.7 instead of
        // 158.2 ms and one 3000-frame sequence 12.25 instead of 12.11 ms, r05_v1_ab.txt: half of the early polls fail, and the
        // retry costs a full round trip.)
        if (step > 0) {
            const unsigned want = a.epoch_base + (unsigned)step;
            const unsigned soff = (unsigned)((step + 1) & 1) * (H * 8);
            unsigned spins = 0; u64 wt0 = 0;
            u32x4 ga, gb;
            while (true) {
                ga = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, gvoff, soff, 16 /* sc1 */));
                gb = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, gvoff + 1024u, soff, 16 /* sc1 */));
                const bool ok = ga[1] == want && ga[3] == want && gb[1] == want && gb[3] == want;
                if (PROF && prof) ++polls;
                if (__all(ok)) break;
                if (wait_over(spins, spin_budget, wt0, a.max_ticks)) {      // bounded: flag the error and never wait again
                    if (lane == 0) set_error_hidden(a.err, 1 + step);
                    spin_budget = 0; poison_cells(cst);
                    break;
                }
                // (no s_sleep in front of the retry: with s_sleep 1 the replay chain took 162 instead of 152 ms, with s_sleep 3
                //  173 -- the retry is a full round trip already)
            }
            // values 2 lane, 2 lane + 1 and 128 + 2 lane, + 1
            float* w0 = hs + (lane >> 4) * HSTR + 2 * (lane & 15);
            w0[0] = __uint_as_float(ga[0]); w0[1] = __uint_as_float(ga[2]);
            float* w1 = w0 + 4 * HSTR;
            w1[0] = __uint_as_float(gb[0]); w1[1] = __uint_as_float(gb[2]);
            // (everything this wave had in flight has been acknowledged: the outputs of the steps before are visible)
            if (wf_l0) link_publish(step);
        }
        asm volatile("" ::: "memory");
        V1_E(1); V1_T(2);
        // ---- x_{t+1} is requested now: nothing of it is in flight while the next poll runs (loads return in order)
        if (step + 1 < T) {
            const int tn = time_of(step + 1);
            if (wf_l1) link_need(tn + 1, step);
            load_x(tn);
        }
        // ---- recurrent part
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(hseg + 4 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(wh[4 * i + j], v[j], acc[j]);
        }
        V1_E(2); V1_T(3);
        const float gate = sum_over_segments((acc[0] + acc[1]) + (acc[2] + acc[3])) + bias;
        V1_E(3); V1_T(4);
        // ---- cell: every lane its gate's non-linearity (sigmoidf_ / tanhf_ of mp_lstm_dev.h by per-lane constants), quad exchange
        const float e = __builtin_amdgcn_exp2f(cx * gate);
        const float r = __builtin_amdgcn_rcpf(1.0f + e);
        const float tv = 1.0f - 2.0f * r;
        const float av = isg ? tv : r;
        const float ig = dpp_quad(av, 0), fg = dpp_quad(av, 1), gt = dpp_quad(av, 2), og = dpp_quad(av, 3);
        const bool act = step < len;
        const int tt = act ? (d.reverse ? len - 1 - step : step) : step;
        const float cnew = fg * cst + ig * gt;
        const float hnew = og * tanhf_(cnew);
        cst = act ? cnew : cst;
        hst = act ? hnew : hst;
        const float oval = act ? hnew : 0.f;
        // (all stores of the loop as inline asm: store_granule_xcd in mp_lstm_dev.h)
        if (s == 0 && g == 0) {
            u64* gp = gown + (size_t)(step & 1) * H;
            store_granule_xcd(gp, a.epoch_base + (unsigned)step + 1u, hst);
            if (!all_local) store_granule_dev(gp + C::G_R, a.epoch_base + (unsigned)step + 1u, hst);
        }
        if (owner) {
            unsigned* op = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(outb) + (size_t)(unsigned)tt * out_row_bytes);
            if (wf_l0 && !link_local) store_word_dev(op, __float_as_uint(oval));
            else store_word_plain(op, __float_as_uint(oval));
        }
        V1_E(4);
    }
    if (wf_l0) link_publish(T);                                         // the last steps' outputs are in the buffer
    if (PROF && prof) {
        long long* o = a.prof + (size_t)blockIdx.x * 8;
        for (int i = 0; i < 5; ++i) o[i] = pt[i];
        o[5] = T;
        o[6] = polls;
        o[7] = (all_local ? 256 : 0) | (link_local ? 512 : 0) | my_xcc | ((long long)link_waits << 16);
    }
    if (owner) {
        d.hbuf[(size_t)b * H + junit] = hst;
        d.cbuf[(size_t)b * H + junit] = cst;
    }
}

template <int KIN, bool PROF, bool WF>
MP_KERNEL __launch_bounds__(256, (WF ? 2 : 1)) void mp_lstm_v1(LstmPersistArgs a) {
    if (PROF && a.debug_drop && (int)blockIdx.x == a.debug_drop - 1) return;      // test hook (mp_debug_drop_workgroup)
    constexpr int NSLICE = V1Cfg<KIN>::NSLICE;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    // ---- cluster (direction, sequence) and slice: host table by XCD, or round robin (see mp_lstm_fused)
    const int ncl = a.ndir * a.nslab;
    const int xcd = a.xcd_physical ? (int)(xcc_id() & 7) : (int)(blockIdx.x & 7);
    const int kth = (int)(blockIdx.x >> 3) / NSLICE;
    const int slice = (int)(blockIdx.x >> 3) % NSLICE;
    if (kth >= mp_xcd_count(a, xcd)) return;
    const int cl = mp_xcd_first(a, xcd) + kth;
    if (cl >= ncl) return;
    if (WF) {
        if (cl & 1) v1_cluster<KIN, PROF, 2>(a, smem, cl, 1, cl >> 1, slice);
        else v1_cluster<KIN, PROF, 1>(a, smem, cl, 0, cl >> 1, slice);
    } else {
        v1_cluster<KIN, PROF, 0>(a, smem, cl, cl / a.nslab, cl % a.nslab, slice);
    }
}

// ---- H = 64 (the foot-contact block), one sequence: a whole (direction, sequence) layer in ONE workgroup of 16 waves -- 256 gate
// rows x 4 K segments, (K_in + 64) / 4 = 32 / 48 weights per lane -- so h_t never leaves the CU: owners drop it into LDS (two
// parity slots), one LDS-only barrier per step, everybody reads back its segment.  No exchange area, no polling, nothing to
// time out.  A step is ~600 cycles, far less than a trip to memory: rows of x are requested in bursts of four, four to eight
// steps before their use (two register sets of four 16-byte pieces, the step loop unrolled eight times: the wait the compiler
// puts at the loop header -- it drains everything there -- then meets requests that are four steps old), and the first
// K_in / 4 lanes of wave 0 put row t + 1 into LDS during step t.  The same quad layout, reductions (v_permlane16_swap / v_permlane32_swap)
// and cell as mp_lstm_v1.  Placement: LstmPersistArgs' XCD table with one workgroup per cluster (beside the other blocks'
// clusters the host gives every cluster an XCD with room).

template <int KIN>
MP_KERNEL __launch_bounds__(1024, 1) void mp_lstm_v1s(LstmPersistArgs a) {
    constexpr int H = 64, KS = KIN / 4, HSTR = 16 + 4, XSTR = KS + 4;
    __shared__ __attribute__((aligned(16))) float hs[2][4 * HSTR];
    __shared__ __attribute__((aligned(16))) float xs[2][4 * XSTR];
    const int ncl = a.ndir * a.nslab;
    const int xcd = a.xcd_physical ? (int)(xcc_id() & 7) : (int)(blockIdx.x & 7);
    const int kth = (int)(blockIdx.x >> 3);
    if (kth >= mp_xcd_count(a, xcd)) return;
    const int cl = mp_xcd_first(a, xcd) + kth;
    if (cl >= ncl) return;
    const int dir = cl / a.nslab, seq = cl % a.nslab;
    const LstmDir d = a.d[dir];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = lane >> 4, g = lane & 3;
    const int junit = wave * 4 + ((lane >> 2) & 3);
    const int B = a.B, T = a.T;
    const int b = a.slab0 + seq;
    const bool in = b < B;
    const int len = in ? a.lengths[b] : 0;

    float wx[KS], wh[16];
    {
        const float* px = d.wihpack + (size_t)(g * H + junit) * KIN + s * KS;
#pragma unroll
        for (int i = 0; i < KS / 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(px + 4 * i);
            wx[4 * i] = v[0]; wx[4 * i + 1] = v[1]; wx[4 * i + 2] = v[2]; wx[4 * i + 3] = v[3];
        }
        const float* ph = d.wpack + (size_t)(g * H + junit) * H + s * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(ph + 4 * i);
            wh[4 * i] = v[0]; wh[4 * i + 1] = v[1]; wh[4 * i + 2] = v[2]; wh[4 * i + 3] = v[3];
        }
    }
    float bias = d.bias[4 * junit + g];
    const bool isg = g == 2;
    const float cx = isg ? 2.8853900817779268f : -1.4426950408889634f;
    float cst = (in && !a.zero_state) ? d.cbuf[(size_t)b * H + junit] : 0.f;
    float hst = (in && !a.zero_state) ? d.hbuf[(size_t)b * H + junit] : 0.f;
    const bool pub = s == 0 && g == 0;                                  // the lane that holds its unit for everybody else
    const unsigned out_row_bytes = (unsigned)B * (unsigned)d.outStride * 4u;
    float* hown = &hs[0][(junit >> 4) * HSTR + (junit & 15)];
    if (pub) hown[4 * HSTR] = hst;                                      // the initial state as h_{-1}: slot 1

    // ---- x rows: loader lanes (values 4 tid .. 4 tid + 3 of a row)
    const size_t xtstride = (size_t)B * KIN;
    auto time_of = [&](int st) {
        int t = d.reverse ? len - 1 - st : st;
        t = t < 0 ? 0 : t;
        return t < T ? t : T - 1;
    };
    const bool loader = threadIdx.x < KIN / 4;
    const float* xlane = d.xin + (size_t)(in ? b : 0) * KIN + 4 * (threadIdx.x & (KIN / 4 - 1));
    float* xput = &xs[0][((4 * (threadIdx.x & (KIN / 4 - 1))) / KS) * XSTR + (4 * (threadIdx.x & (KIN / 4 - 1))) % KS];
    f32x4 qa[4], qb[4];                                                 // rows step0 + 1 .. + 4 | step0 + 5 .. + 8 of the trip that starts at step0
    // (every thread requests the rows -- 16 waves x 256 / 512 bytes, nothing -- and only the loader lanes use them: with the
    //  requests under `if (loader)` the wait-count pass merged the two paths into a vmcnt(0) in front of every use)
    if (loader) *reinterpret_cast<f32x4*>(xput) = *reinterpret_cast<const f32x4*>(xlane + (size_t)time_of(0) * xtstride);
#pragma unroll
    for (int k = 0; k < 4; ++k) qa[k] = *reinterpret_cast<const f32x4*>(xlane + (size_t)time_of(1 + k) * xtstride);
    __syncthreads();
    // the layer output of step st: wave 1 reads h_st back from its LDS slot (one 256-byte row store instead of 64 lanes of 16
    // waves storing a word each).  Wave 0's loads are the ones that matter and it issues no stores; the store is inline asm so
    // that the compiler's wait-count pass does not see a store among the pending loads (with both kinds pending it drains
    // everything -- vmcnt(0) -- at the next use of a loaded row).
    auto put_row = [&](int st) {
        const bool act1 = st < len;
        const int tt = act1 ? (d.reverse ? len - 1 - st : st) : st;
        const float hv = hs[st & 1][(lane >> 4) * HSTR + (lane & 15)];
        if (in) store_word_plain(reinterpret_cast<unsigned*>(reinterpret_cast<char*>(d.out + (size_t)b * d.outStride + lane) + (size_t)(unsigned)tt * out_row_bytes),
                                 __float_as_uint(act1 ? hv : 0.f));
    };
    auto one_step = [&](int step, const f32x4& row_next) {
        if (loader) *reinterpret_cast<f32x4*>(xput + ((step + 1) & 1) * 4 * XSTR) = row_next;     // x_{step+1} into the other slot
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const float* xseg = &xs[step & 1][s * XSTR];
#pragma unroll
        for (int i = 0; i < KS / 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(xseg + 4 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(wx[4 * i + j], v[j], acc[j]);
        }
        const float* hseg = &hs[(step + 1) & 1][s * HSTR];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(hseg + 4 * i);
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = __builtin_fmaf(wh[4 * i + j], v[j], acc[j]);
        }
        float sum = (acc[0] + acc[1]) + (acc[2] + acc[3]);
        {
            const unsigned u = __float_as_uint(sum);
            const auto p = __builtin_amdgcn_permlane16_swap(u, u, false, false);
            sum = __uint_as_float(p[0]) + __uint_as_float(p[1]);
        }
        {
            const unsigned u = __float_as_uint(sum);
            const auto p = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            sum = __uint_as_float(p[0]) + __uint_as_float(p[1]);
        }
        const float gate = sum + bias;
        const float e = __builtin_amdgcn_exp2f(cx * gate);
        const float r = __builtin_amdgcn_rcpf(1.0f + e);
        const float tv = 1.0f - 2.0f * r;
        const float av = isg ? tv : r;
        const float ig = dpp_quad(av, 0), fg = dpp_quad(av, 1), gt = dpp_quad(av, 2), og = dpp_quad(av, 3);
        const bool act = step < len;
        const float cnew = fg * cst + ig * gt;
        const float hnew = og * tanhf_(cnew);
        cst = act ? cnew : cst;
        hst = act ? hnew : hst;
        if (pub) hown[(step & 1) * 4 * HSTR] = hst;
        if (wave == 1 && step > 0) put_row(step - 1);
        barrier_lds_only();                                             // h_t and x_{t+1} are in their slots; everybody is done with the other ones
    };

    for (int step0 = 0; step0 < T; step0 += 8) {
#pragma unroll
        for (int k = 0; k < 4; ++k) qb[k] = *reinterpret_cast<const f32x4*>(xlane + (size_t)time_of(step0 + 5 + k) * xtstride);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (step0 + k < T) one_step(step0 + k, qa[k]);
        if (step0 + 4 >= T) break;
#pragma unroll
        for (int k = 0; k < 4; ++k) qa[k] = *reinterpret_cast<const f32x4*>(xlane + (size_t)time_of(step0 + 9 + k) * xtstride);
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (step0 + 4 + k < T) one_step(step0 + 4 + k, qb[k]);
    }
    if (wave == 1) put_row(T - 1);
    if (pub && in) {
        d.hbuf[(size_t)b * H + junit] = hst;
        d.cbuf[(size_t)b * H + junit] = cst;
    }
}

// W [4H][K] (H = 256, rows gate * H + unit) -> [slice][wave][piece][lane][4]: lane (s, u2, g) = (lane / 8, lane / 4 % 2, lane % 4)
// holds row g * H + slice * 8 + wave * 2 + u2, k = s * K / 8 + 4 piece + e
MP_KERNEL void mp_pack_w_v1(const float* __restrict__ w, float* __restrict__ dst, int K) {
    constexpr int H = 256;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)4 * H * K) return;
    const int NP = K / 32;                                               // pieces per lane
    size_t rest = idx;
    const int e = (int)(rest % 4); rest /= 4;
    const int lane = (int)(rest % 64); rest /= 64;
    const int p = (int)(rest % NP); rest /= NP;
    const int wave = (int)(rest % 4); rest /= 4;
    const int slice = (int)rest;
    const int row = (lane & 3) * H + slice * 8 + wave * 2 + ((lane >> 2) & 1);
    const int k = (lane >> 3) * (K / 8) + 4 * p + e;
    dst[idx] = w[(size_t)row * K + k];
}

template <int KIN, bool WF>
void launch_v1(const LstmPersistArgs& a, hipStream_t s) {
    LstmPersistArgs b = a;
    int most = 0, total = 0;
    for (int x = 0; x < 8; ++x) { most = b.xcd_cnt[x] > most ? b.xcd_cnt[x] : most; total += b.xcd_cnt[x]; }
    if (total != a.nslab * a.ndir) {
        mp_fill_xcd_table(b, nullptr);
        most = (a.nslab * a.ndir + 7) / 8;
    }
    size_t lds = V1Cfg<KIN>::LDS_BYTES;
    if ((size_t)a.min_lds > lds) lds = (size_t)a.min_lds;
    const dim3 grid(8 * most * 32);
    if (a.prof || a.debug_drop) hipLaunchKernelGGL((mp_lstm_v1<KIN, true, WF>), grid, dim3(256), lds, s, b);
    else hipLaunchKernelGGL((mp_lstm_v1<KIN, false, WF>), grid, dim3(256), lds, s, b);
}

template <int KIN, bool WF>
hipError_t v1_attrs() {
    const int lds = 96 * 1024;                                          // (room for LstmPersistArgs::min_lds)
    hipError_t e = hipFuncSetAttribute((const void*)mp_lstm_v1<KIN, true, WF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)mp_lstm_v1<KIN, false, WF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace

// a.nslab = number of sequences (one cluster per direction and sequence); d[].wpack / wihpack = W_hh / W_ih from
// mp_launch_pack_w_v1.  wavefront: a.ndir = 2, d[0] = layer 0, d[1] = layer 1 reading d[0].out (K_in = 256).
void mp_launch_lstm_v1(const LstmPersistArgs& a, int KIN, bool wavefront, hipStream_t s) {
    if (wavefront) launch_v1<256, true>(a, s);
    else if (KIN == 256) launch_v1<256, false>(a, s);
    else launch_v1<512, false>(a, s);
}
void mp_launch_pack_w_v1(const float* w, float* dst, int K, hipStream_t s) {
    const size_t n = (size_t)4 * 256 * K;
    hipLaunchKernelGGL(mp_pack_w_v1, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, dst, K);
}
// H = 64, one workgroup per (direction, sequence); the ROW-MAJOR W_hh [256][64], W_ih [256][K_in] of torch.nn.LSTM
void mp_launch_lstm_v1s(const LstmPersistArgs& a, int KIN, hipStream_t s) {
    LstmPersistArgs b = a;
    int most = 0, total = 0;
    for (int x = 0; x < 8; ++x) { most = b.xcd_cnt[x] > most ? b.xcd_cnt[x] : most; total += b.xcd_cnt[x]; }
    if (total != a.nslab * a.ndir) {
        mp_fill_xcd_table(b, nullptr);
        most = (a.nslab * a.ndir + 7) / 8;
    }
    const dim3 grid(8 * most);
    if (KIN == 64) hipLaunchKernelGGL((mp_lstm_v1s<64>), grid, dim3(1024), 0, s, b);
    else hipLaunchKernelGGL((mp_lstm_v1s<128>), grid, dim3(1024), 0, s, b);
}
hipError_t mp_lstm_v1_device_attrs() {
    hipError_t e = v1_attrs<256, false>();
    if (!e) e = v1_attrs<512, false>();
    if (!e) e = v1_attrs<256, true>();
    return e;
}

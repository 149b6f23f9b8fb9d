// K2p -- the fused persistent nn.LSTM layer of mp_lstm_persist.hip (same arithmetic, same summation order, same
// granule exchange: see the header of that file) with TWO slabs of 16 sequences per workgroup.
//
// Why: in mp_lstm_fused a workgroup's 8 waves all belong to one slab, so the two waves of every SIMD are in the same
// phase of a step: when the slab waits for its peers' hidden state (2300 cycles per step, rocprof/phase counters in
// profiles/r02_persist_phases.md), reduces (800) or updates cells (600), the matrix pipe of that CU idles -- 67 % MFMA
// busy on the K_in = 512 layers, 69 % on K_in = 256.  Here a workgroup = (direction, PAIR of slabs, slice of 16 hidden
// units); wave (kq, s) does for slab s what wave kq of a 4-wave mp_lstm_fused<256,16,KIN,1> workgroup does, and the two
// waves of a SIMD -- (kq, 0) and (kq, 1) -- belong to DIFFERENT slabs: two independent recurrences whose MFMA phases
// interleave on the pipe.  Slab 1 starts half a step late (s_sleep) so that the two stay out of phase.
//   * 16 slices x 16 units: the W_hh slice of a wave is 64 VGPRs as before; the W_ih slice (K_in x 64 columns) is ONE
//     LDS image shared by both slabs (64 KB for K_in = 256; K_in = 512: 96 KB in LDS + 8 k-steps in registers);
//   * the 4 K-quarter waves of a slab meet in LDS for the K reduction behind a 4-wave barrier built from an LDS
//     counter (gfx950 has no named barriers; s_barrier would couple the two slabs again); the reduction scratch is
//     double-buffered by step parity, so one such barrier per step suffices;
//   * grid = ndir x ceil(nslab / 2) x 16 workgroups = 256 for a bidirectional layer at B = 256: one per CU, a pair's 16
//     slices on one XCD (same blockIdx -> XCD mapping as mp_lstm_fused).
// Weights come from the 16-slice packing of mp_lstm_persist.hip (mp_launch_pack_{whh,wih}_persist(.., nslice = 16)).
#include "mp_lstm_dev.h"

namespace {

__device__ __forceinline__ int granule_index2(int row, int j) { return (((j >> 2) * 16 + row) << 2) + (j & 3); }

template <int KIN>
struct CfgP {
    static constexpr int H = 256, NSLICE = 16, U = 16;
    static constexpr int KW = H / 4, NKS = KW / 4;       // h: K range / k-steps of one wave (64 / 16)
    static constexpr int KQ = KIN / 4, NXS = KQ / 4;     // x: K range / k-steps of one wave
    static constexpr int NXJ = KQ / 16;                  // x: 16-byte loads per lane per step
    static constexpr int NPW = 4;                        // producer slices inside one wave's K quarter
    static constexpr int RED_F4 = 2 * 2 * 4 * 4 * 64;    // [parity][slab][finishing wave][source kq][lane] float4
    static constexpr int STEP_BYTES = 4 * 64 * 16;       // W_ih: the 4 K-quarter waves, one k-step
    static constexpr int LDS_BUDGET = 160 * 1024 - 512;  // minus the barrier counters and pipe tokens
    static constexpr int LDS_STEPS_MAX = (LDS_BUDGET - RED_F4 * 16) / STEP_BYTES;
    static constexpr int XL = NXS <= LDS_STEPS_MAX ? NXS : (LDS_STEPS_MAX / 4) * 4;
    static constexpr int XR = NXS - XL;
    static constexpr bool BIG = KIN > H;
    static constexpr size_t LDS_BYTES = (size_t)RED_F4 * 16 + (size_t)4 * XL * 64 * 16 + 512;
};

// 4-wave barrier of one slab group on an LDS counter.  Only LDS traffic is waited for (no vmcnt drain: the x prefetch
// stays in flight).  The LDS unit serves a CU's requests in order, so a wave's partial sums are in LDS before its
// increment, and a wave that has seen the target reads them afterwards.
// Bounded like every other wait of these kernels (a peer wave cannot really stay away, but a hang must be impossible).
__device__ __forceinline__ bool slab_barrier(unsigned* cnt, unsigned target, int lane) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned spins = 0;
    bool ok = true;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) {
        if (++spins > (1u << 22)) { ok = false; break; }
        __builtin_amdgcn_s_sleep(1);                 // keep the LDS port free for the other slab's weight reads
    }
    asm volatile("" ::: "memory");
    return ok;
}

template <int KIN, bool PROF>
MP_KERNEL __launch_bounds__(512, 1) void mp_lstm_pair(LstmPersistArgs a, int mode) {
    using C = CfgP<KIN>;
    constexpr int H = C::H, NSLICE = C::NSLICE, U = C::U, KW = C::KW, NKS = C::NKS, KQ = C::KQ, NXS = C::NXS, NXJ = C::NXJ;
    constexpr int XL = C::XL, XR = C::XR, NPW = C::NPW;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* red = reinterpret_cast<f32x4*>(smem);                       // [parity][slab][finishing wave][source kq][lane]
    f32x4* wxl = reinterpret_cast<f32x4*>(smem) + C::RED_F4;           // [kq][x-step < XL][lane]
    unsigned* cnts = reinterpret_cast<unsigned*>(wxl + (size_t)4 * XL * 64);   // [slab] at 128-byte distance

    const int npair = (a.nslab + 1) / 2;
    const int npc = a.ndir * npair;
    const int pc = ((int)(blockIdx.x >> 3) / NSLICE) * 8 + (int)(blockIdx.x & 7);
    const int slice = (int)(blockIdx.x >> 3) % NSLICE;
    if (pc >= npc) return;
    const int dir = pc / npair, pair = pc % npair;
    const LstmDir d = a.d[dir];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int kq = wave & 3, sl = wave >> 2;                           // K quarter, slab of the pair
    const int slab = 2 * pair + sl;
    const int q = lane >> 4, r16 = lane & 15;
    const int B = a.B, T = a.T;

    // ---- W_ih slice: k-steps [0, XL) -> LDS (shared by both slabs), [XL, NXS) -> registers; W_hh slice -> registers
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(d.wihpack) + (size_t)slice * 4 * NXS * 64;
        for (int w = 0; w < 4; ++w)
            for (int i = threadIdx.x; i < XL * 64; i += 512)
                wxl[(size_t)w * XL * 64 + i] = src[(size_t)w * NXS * 64 + i];
        if (threadIdx.x < 128) cnts[threadIdx.x] = 0u;
    }
    __syncthreads();                                                   // the only workgroup-wide barrier
    if (slab >= a.nslab) return;                                       // odd slab count: the second half idles
    const int cl = dir * a.nslab + slab;                               // exchange area of this (direction, slab)
    const int brow0 = (a.slab0 + slab) * 16;
    unsigned* cnt = cnts + sl * 32;
    // Matrix-pipe token of this SIMD (waves (kq, 0) and (kq, 1) share one, tools/micro/hwid.hip): the two slabs take
    // turns burst by burst in a FIXED order -- A.x_t, B.x_t, A.h_t, B.h_t, A.x_{t+1} ... -- instead of colliding at
    // random: a collision stretches a burst up to 2x on ONE of the 64 SIMDs of a cluster, and the lock-step of the
    // cluster turns the worst SIMD into everybody's step time (measured: 13 700 cycles per step without the token,
    // profiles/r02_persist_phases.md).  With the fixed order every wait of a slab (hidden-state exchange, reduction,
    // cell update) lies under the other slab's burst.
    unsigned* tok = cnts + 64 + kq * 8;
    const bool duo = (mode & 1) && (2 * pair + 1 < a.nslab);
    auto acquire = [&](unsigned ticket) {
        if (!duo) return;
        unsigned spins = 0;
        while (__hip_atomic_load(tok, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) != ticket) {
            if (++spins > (1u << 22)) { if (lane == 0) mp_set_error(a.err, 3000000); break; }
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
    };
    auto release = [&](unsigned ticket) {
        __builtin_amdgcn_sched_barrier(0);            // after the burst's last MFMA has been issued
        if (duo && lane == 0) __hip_atomic_store(tok, ticket + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        __builtin_amdgcn_sched_barrier(0);
    };

    f32x4 wxr[XR > 0 ? XR : 1];
    if (XR > 0) {
        const f32x4* src = reinterpret_cast<const f32x4*>(d.wihpack) + ((size_t)(slice * 4 + kq) * NXS + XL) * 64 + lane;
#pragma unroll
        for (int s = 0; s < XR; ++s) wxr[s] = src[(size_t)s * 64];
    }
    float wv[NKS][4];
    {
        const float* wp = d.wpack + ((size_t)(slice * 4 + kq) * NKS * 4) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int t = 0; t < 4; ++t) wv[ks][t] = wp[(size_t)(ks * 4 + t) * 64];
    }

    // ---- the (sequence, unit) this lane finishes: accumulator reg kq of tile column r16
    const int jown = slice * U + r16;
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
    const float* xbase = d.xin + (size_t)(arow_in ? arow : 0) * KIN + kq * KQ + q * 4;
    const size_t xtstride = (size_t)B * KIN;
    float av[NKS];
    {
        const float* p = d.hbuf + (size_t)(arow_in ? arow : 0) * H + kq * KW + q;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) av[ks] = (arow_in && !a.zero_state) ? p[4 * ks] : 0.f;
    }

    // granules of this slab: hx[cl] = { L[2 parities][16*H], R[2 parities][16*H], xcc[16] }
    constexpr size_t SLABW = (size_t)4 * 16 * H + 16;
    u64* hxL = a.hx + (size_t)cl * SLABW;
    u64* hxR = hxL + (size_t)2 * 16 * H;
    u64* xtab = hxL + (size_t)4 * 16 * H;
    unsigned spin_budget = a.max_spin;
    const unsigned my_xcc = xcc_id();
    bool src_local[NPW];
    bool all_local = true;
    {
        if (kq == 0 && lane == 0) granule_store(xtab + slice, XCC_TAG, __uint_as_float(my_xcc));
        unsigned peer = my_xcc;
        if (lane < NSLICE) {
            unsigned spins = 0;
            while (true) {
                const u64 g = granule_load(xtab + lane);
                if ((unsigned)(g >> 32) == XCC_TAG) { peer = (unsigned)g; break; }
                if (++spins > spin_budget) { mp_set_error(a.err, 1000000); peer = ~0u; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        const unsigned long long same = __ballot(peer == my_xcc);
        all_local = (same & 0xffffull) == 0xffffull;
#pragma unroll
        for (int i = 0; i < NPW; ++i) src_local[i] = (same >> (NPW * kq + i)) & 1;
        if (__ballot(peer == ~0u)) spin_budget = 0;
        if (a.force_remote) {
            all_local = false;
#pragma unroll
            for (int i = 0; i < NPW; ++i) src_local[i] = false;
        }
    }

    f32x4 xa[NXJ];
    constexpr bool SPLIT_X = C::BIG;
    constexpr int XJ_PRE = SPLIT_X ? NXJ / 2 : NXJ;
    auto load_x = [&](int step, int j0, int j1) {
        const bool on = step < alen;
        const int t = on ? (d.reverse ? alen - 1 - step : step) : 0;
        const float* p = xbase + (size_t)t * xtstride;
#pragma unroll
        for (int j = 0; j < NXJ; ++j)
            if (j >= j0 && j < j1) xa[j] = on ? *reinterpret_cast<const f32x4*>(p + j * 16) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    load_x(0, 0, XJ_PRE);


    long long pt[6] = {0, 0, 0, 0, 0, 0};
    const bool prof = PROF && a.prof != nullptr && lane == 0 && kq == 0;
    // (debug) absolute time stamps of steps 64..71 of workgroup 0, both slabs: a.prof[4096 + ((sl*8 + step-64)*8 + i)]
#define PROF_T(i) do { if (PROF && prof) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); pt[i] -= t_; \
        if (blockIdx.x == 0 && step >= 64 && step < 72) a.prof[4096 + ((sl * 8 + step - 64) * 8 + i)] = t_; } } while (0)
#define PROF_E(i) do { if (PROF && prof) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); pt[i] += t_; \
        if (blockIdx.x == 0 && step >= 64 && step < 72 && i == 4) a.prof[4096 + ((sl * 8 + step - 64) * 8 + 5)] = t_; } } while (0)

    const f32x4* wxw = wxl + (size_t)kq * XL * 64 + lane;

    for (int step = 0; step < T; ++step) {
        PROF_T(0);
        if (SPLIT_X) load_x(step, XJ_PRE, NXJ);
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        // LDS-resident W_ih fragments are fetched TWO k-steps ahead (one k-step = 4 MFMAs = 128 cycles does not cover a
        // ds_read_b128 under load: 47 instead of 32 cycles per MFMA when the wave has the pipe to itself)
        f32x4 wl = wxw[0], wn = XL > 1 ? wxw[64] : wl, wnn = wn;
        acquire(4u * (unsigned)step + (unsigned)sl);
        // ---- first half of x_t W_ih^T
#pragma unroll
        for (int s = 0; s < NXS / 2; ++s) {
            const float a_s = xa[s >> 2][s & 3];
            if (s + 2 < XL) wnn = wxw[(size_t)(s + 2) * 64];
            const f32x4 w4 = s < XL ? wl : wxr[s >= XL ? s - XL : 0];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_s, w4[i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            wl = wn; wn = wnn;
        }
        // ---- request h_{step-1}
        u64 gr[NKS];
        const unsigned epoch = (unsigned)step;
        const size_t goff = (size_t)((step + 1) & 1) * 16 * H + (size_t)kq * NKS * 64 + r16 * 4 + q;
        const u64* srcp[NPW];
#pragma unroll
        for (int i = 0; i < NPW; ++i) srcp[i] = (src_local[i] ? hxL : hxR) + goff;
        constexpr int KSP = NKS / NPW;
        // (K_in = 512: no room to hold 16 granules in flight beside the register-resident W_ih k-steps and the second half
        //  of x: they are requested after the projection; the other slab's wave covers the latency)
        constexpr bool EARLY_GATHER = !C::BIG;
        if (EARLY_GATHER && step > 0) {
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) gr[ks] = granule_load(srcp[ks / KSP] + (size_t)ks * 64);
        }
        // ---- second half of the input projection
#pragma unroll
        for (int s = NXS / 2; s < NXS; ++s) {
            const float a_s = xa[s >> 2][s & 3];
            if (s + 2 < XL) wnn = wxw[(size_t)(s + 2) * 64];
            const f32x4 w4 = s < XL ? wl : wxr[s >= XL ? s - XL : 0];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_s, w4[i], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            wl = wn; wn = wnn;
        }
        release(4u * (unsigned)step + (unsigned)sl);
        PROF_E(0); PROF_T(1);

        // ---- validate the granules; slow path (cheap gate, then sweep) only when some were stale
        if (step > 0) {
            if (!EARLY_GATHER) {
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) gr[ks] = granule_load(srcp[ks / KSP] + (size_t)ks * 64);
            }
            bool ok = true;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) ok = ok && ((unsigned)(gr[ks] >> 32) == epoch);
            unsigned spins = 0;
            bool timed_out = false;
            if (PROF && prof && !__all(ok)) pt[5] += 1;
            while (!__all(ok) && !timed_out) {
                while (true) {
                    bool ready = true;
#pragma unroll
                    for (int i = 0; i < NPW; ++i)
                        if (lane == i) ready = (unsigned)(granule_load(srcp[i] + (size_t)(i * KSP) * 64) >> 32) == epoch;
                    if (__all(ready)) break;
                    if (++spins > spin_budget) { timed_out = true; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                ok = true;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    gr[ks] = granule_load(srcp[ks / KSP] + (size_t)ks * 64);
                    ok = ok && ((unsigned)(gr[ks] >> 32) == epoch);
                }
                if (++spins > spin_budget) timed_out = true;
            }
            if (timed_out) {
                if (lane == 0) mp_set_error(a.err, 1 + step);
                spin_budget = 0;
            }
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) av[ks] = __uint_as_float((unsigned)gr[ks]);
        }
        load_x(step + 1, 0, XJ_PRE);
        PROF_E(1); PROF_T(2);

        // ---- recurrent part
        acquire(4u * (unsigned)step + 2u + (unsigned)sl);
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], wv[ks][t], acc[t], 0, 0, 0);
        release(4u * (unsigned)step + 2u + (unsigned)sl);
        PROF_E(2); PROF_T(3);

        // ---- K reduction of this slab's 4 waves through LDS (scratch double-buffered by step parity)
        f32x4* rbuf = red + (size_t)((step & 1) * 2 + sl) * 16 * 64;
#pragma unroll
        for (int dk = 0; dk < 4; ++dk)
            rbuf[(dk * 4 + kq) * 64 + lane] = f32x4{acc[0][dk], acc[1][dk], acc[2][dk], acc[3][dk]};
        if (!slab_barrier(cnt, 4u * (unsigned)(step + 1), lane) && lane == 0) mp_set_error(a.err, 2000000 + step);
        f32x4 gate = rbuf[(kq * 4 + 0) * 64 + lane];
#pragma unroll
        for (int sw = 1; sw < 4; ++sw) gate += rbuf[(kq * 4 + sw) * 64 + lane];
        gate += bias4;
        PROF_E(3); PROF_T(4);

        // ---- cell update, publish h_step, layer output
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
        const int gi = granule_index2(q * 4 + kq, jown);
        granule_store_l2(hxL + doff + gi, (unsigned)(step + 1), hst);
        if (!all_local) granule_store(hxR + doff + gi, (unsigned)(step + 1), hst);
        if (inb) d.out[((size_t)tt * B + bown) * d.outStride + jown] = oval;
        PROF_E(4);
    }
    if (PROF && prof) {
        long long* o = a.prof + ((size_t)blockIdx.x * 2 + sl) * 8;
        for (int i = 0; i < 5; ++i) o[i] = pt[i];
        o[5] = T;
        o[6] = pt[5];
        o[7] = (all_local ? 256 : 0) | my_xcc;
    }
    if (inb) {
        d.hbuf[(size_t)bown * H + jown] = hst;
        d.cbuf[(size_t)bown * H + jown] = cst;
    }
}

template <int KIN>
void launch_pair(const LstmPersistArgs& a, int mode, hipStream_t s) {
    const int npc = a.ndir * ((a.nslab + 1) / 2);
    const dim3 grid(((npc + 7) / 8) * 8 * 16);
    if (a.prof) hipLaunchKernelGGL((mp_lstm_pair<KIN, true>), grid, dim3(512), CfgP<KIN>::LDS_BYTES, s, a, mode);
    else hipLaunchKernelGGL((mp_lstm_pair<KIN, false>), grid, dim3(512), CfgP<KIN>::LDS_BYTES, s, a, mode);
}

template <int KIN>
hipError_t pair_attrs() {
    const int lds = (int)CfgP<KIN>::LDS_BYTES;
    hipError_t e = hipFuncSetAttribute((const void*)mp_lstm_pair<KIN, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute((const void*)mp_lstm_pair<KIN, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace

// H = 256, 16-slice weight packing; mode bit 0: the two slabs of a workgroup take turns on the matrix pipe (token)
void mp_launch_lstm_pair(const LstmPersistArgs& a, int KIN, int mode, hipStream_t s) {
    if (KIN == 256) launch_pair<256>(a, mode, s);
    else launch_pair<512>(a, mode, s);
}

hipError_t mp_lstm_pair_device_attrs() {
    hipError_t e = pair_attrs<256>();
    return e != hipSuccess ? e : pair_attrs<512>();
}

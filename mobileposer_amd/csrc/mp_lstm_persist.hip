// K2p -- persistent nn.LSTM recurrence (models/rnn.py:27): ONE launch walks all T time steps of up to two
// (layer, direction) instances.  Same arithmetic, gate order and packed-sequence semantics as the per-step
// kernel in mp_lstm.hip (which stays as the reference implementation / fallback); what changes is where
// the data lives between steps:
//
//   * W_hh never leaves the register file.  A workgroup owns (direction, slab of 16 sequences, slice of
//     U hidden units); its 4 waves split K = H four ways, so a lane holds 4U/16 * H/16 weight values
//     (128 VGPRs for H = 256, U = 32; 64 for H = 64, U = 64), loaded once in B-fragment order.
//   * c_t, and h_t of the lane's own (sequence, unit), stay in registers for all T steps.
//   * h_t crosses workgroups (the 8 slices of a slab need each other's units every step) as 8-byte
//     {epoch, value} granules: one relaxed agent-scope (sc1) store per value, and the consumer lane that
//     needs the value as an MFMA A operand re-reads its 16 granules until every tag equals the step's
//     epoch -- the data is its own flag, so there is no fence, no barrier and no separate flag word
//     (cdna_hip_programming.md Guideline 16, form R2; placement independent).  Two parities suffice: a
//     producer can only be one step ahead of its slowest peer.  Gathered values feed
//     v_mfma_f32_16x16x4_f32 straight from the load registers (no LDS staging).
//   * The 4 K-partials meet in LDS; each wave finishes a quarter of the (sequence, unit) pairs, so the
//     cell update is spread over all lanes.
//
// Per step the critical path is: granule visibility (~1 us) -> 128 MFMAs per wave (1.7 us) -> LDS
// reduction + cell update (~0.3 us) instead of a kernel boundary + a full 128 KB weight re-read.
// All workgroups of a launch must be co-resident (grid <= 256, one per CU); every spin is bounded and a
// timeout raises a device-side error word instead of hanging the GPU.
#include "mp_common.h"

namespace {

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;

// v_exp_f32 / v_rcp_f32 are 1-ulp instructions: sigma and tanh come out within ~2e-7 absolute of libm
__device__ __forceinline__ float sigmoidf_(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x));
}
__device__ __forceinline__ float tanhf_(float x) {
    const float e = __builtin_amdgcn_exp2f(2.8853900817779268f * x);
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
// granule index of (row, hidden unit j) inside one [16][H] slab-parity block: [j/4][row][j%4], so that the
// 64 lanes of a consumer wave (16 rows x 4 consecutive units) read 512 contiguous bytes per instruction
__device__ __forceinline__ int granule_index(int row, int j) { return (((j >> 2) * 16 + row) << 2) + (j & 3); }

__device__ __forceinline__ u64 granule_load(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(u64* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// same 8-byte granule as an ordinary store: it goes through the (write-through) L1 into THIS XCD's L2 and stays
// there, where a consumer on the same XCD finds it with an L1-bypassing (sc1) load at L2-hit latency
__device__ __forceinline__ void granule_store_l2(u64* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned xcc_id() {
    return __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xF;   // s_getreg_b32 hwreg(HW_REG_XCC_ID)
}
constexpr unsigned XCC_TAG = 0x7fffffffu;

// H: hidden size; NSLICE: workgroups sharing one slab (8 for H = 256, 1 for H = 64)
template <int H, int NSLICE>
__global__ __launch_bounds__(256, 1) void mp_lstm_persist(LstmPersistArgs a) {
    constexpr int U = H / NSLICE;              // hidden units per workgroup (32 | 64)
    constexpr int NUB = U / 16;                // 16-unit blocks per workgroup (2 | 4)
    constexpr int NT = 4 * NUB;                // MFMA tiles per wave: gates x unit blocks (8 | 16)
    constexpr int KW = H / 4;                  // K range of one wave (64 | 16)
    constexpr int NKS = KW / 4;                // k-steps per wave (16 | 4)
    constexpr int NOWN = NUB == 2 ? 2 : 4;     // accumulator regs a lane finishes
    // reduction scratch: red[parity][dst wave][src wave][o][lane] of float4 (i,f,g,o partial sums)
    __shared__ __attribute__((aligned(16))) float red[2 * 4 * 4 * 4 * NOWN * 64];

    const LstmDir d = a.d[blockIdx.y];
    int slab, slice;
    if (NSLICE == 8 && (a.nslab & 7) == 0) {
        // keep the 8 slices of a slab on one XCD (block b runs on XCD b % 8): faster hand-off, never needed
        // for correctness
        const int xcd = blockIdx.x & 7, i = blockIdx.x >> 3;
        slab = (i >> 3) * 8 + xcd;
        slice = i & 7;
    } else {
        slab = blockIdx.x / NSLICE;
        slice = blockIdx.x % NSLICE;
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = lane >> 4, r16 = lane & 15;
    const int B = a.B, T = a.T;
    const int brow0 = (a.slab0 + slab) * 16;

    // ---- W_hh slice -> registers (once).  wv[ks][t]: tile t = g*NUB + ub
    float wv[NKS][NT];
    {
        const float* wp = d.wpack + ((size_t)(slice * 4 + wave) * NKS * NT) * 64 + lane;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int t = 0; t < NT; ++t) wv[ks][t] = wp[(size_t)(ks * NT + t) * 64];
    }

    // ---- the (sequence, unit) pairs this lane finishes: accumulator regs of tile column r16
    const int ubo = NUB == 2 ? (wave & 1) : wave;          // unit block this wave finishes
    const int reg0 = NUB == 2 ? 2 * (wave >> 1) : 0;       // first accumulator reg it finishes
    const int jown = slice * U + ubo * 16 + r16;           // hidden unit
    float cst[NOWN], hst[NOWN];
    int blen[NOWN], bidx[NOWN];
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        const int b = brow0 + q * 4 + reg0 + o;
        bidx[o] = b;
        const bool inb = b < B;
        blen[o] = inb ? a.lengths[b] : 0;
        cst[o] = inb ? d.cbuf[(size_t)b * H + jown] : 0.f;
        hst[o] = inb ? d.hbuf[(size_t)b * H + jown] : 0.f;
    }

    // ---- A operand for step 0 from the initial state: h0[row r16][k = wave*KW + 4*ks + q]
    float av[NKS];
    {
        const int b = brow0 + r16;
        const float* p = d.hbuf + (size_t)(b < B ? b : 0) * H + wave * KW + q;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) av[ks] = b < B ? p[4 * ks] : 0.f;
    }

    // granules of this slab: hx[dir][slab] = { L[2 parities][16*H], R[2 parities][16*H], xcc[8] }.
    // Two transports, chosen per PRODUCER from where it really runs (its XCC id, published once):
    //   R: write-through (sc1) stores + sc1 loads -- coherent for any placement, a fabric round trip (~1 us);
    //   L: ordinary stores + sc1 loads -- only when producer and consumer share an XCD (one L2), ~4x faster.
    // A producer always fills L and fills R unless every slice of its slab is on its own XCD, so the result
    // never depends on placement -- only the speed does (cdna_hip_programming.md Guideline 16).
    constexpr size_t SLABW = (size_t)4 * 16 * H + 8;
    u64* hxL = a.hx + (size_t)(blockIdx.y * a.nslab + slab) * SLABW;
    u64* hxR = hxL + (size_t)2 * 16 * H;
    u64* xtab = hxL + (size_t)4 * 16 * H;
    unsigned spin_budget = a.max_spin;
    const unsigned my_xcc = xcc_id();
    bool src_local[2] = {true, true};      // is the producer of this wave's first / second k-half on my XCD?
    bool all_local = true;
    if (NSLICE > 1) {
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
        const unsigned long long same = __ballot(peer == my_xcc);
        all_local = (same & ((1ull << NSLICE) - 1)) == ((1ull << NSLICE) - 1);
        src_local[0] = (same >> (2 * wave)) & 1;           // k-steps 0..NKS/2-1 come from slice 2*wave
        src_local[1] = (same >> (2 * wave + 1)) & 1;       // the rest from slice 2*wave+1
        if (peer == ~0u) spin_budget = 0;
    }

    long long pt[6] = {0, 0, 0, 0, 0, 0};
    const bool prof = a.prof != nullptr && threadIdx.x == 0;
#define PROF_T(i) do { if (prof) pt[i] -= (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define PROF_E(i) do { if (prof) pt[i] += (long long)__builtin_amdgcn_s_memtime(); } while (0)
    for (int step = 0; step < T; ++step) {
        PROF_T(0);
        // ---- gate pre-activations from the input projection (issued early; consumed after the MFMAs)
        f32x4 xp[NOWN];
        int tt[NOWN];
        bool act[NOWN];
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            act[o] = step < blen[o];
            tt[o] = act[o] ? (d.reverse ? blen[o] - 1 - step : step) : step;
            xp[o] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (act[o])
                xp[o] = *reinterpret_cast<const f32x4*>(d.xproj + ((size_t)tt[o] * B + bidx[o]) * d.xprojStride + 4 * jown);
        }

        // ---- gather h_{step-1} and multiply.  The lane's A value for k-step ks is granule
        //      (row r16, unit wave*KW + 4*ks + q): one coalesced 512-byte read per instruction.
        f32x4 acc[NT];
        if (step > 0) {
            const unsigned epoch = (unsigned)step;            // written by the producers at the end of step-1
            const size_t goff = (size_t)((step - 1) & 1) * 16 * H + (size_t)wave * NKS * 64 + r16 * 4 + q;
            const u64* src0 = (src_local[0] ? hxL : hxR) + goff;
            const u64* src1 = NSLICE > 1 ? (src_local[1] ? hxL : hxR) + goff : src0;
            unsigned spins = 0;
            bool timed_out = false;
            // (1) cheap gate: 4 lanes per wave watch ONE granule each (all lanes x all granules would flood the
            //     fabric with sc1 loads and slow every hand-off on the chip: MI355X_MICROARCH "polling-cost")
            while (true) {
                bool ready = true;
                if (r16 == 0) ready = (unsigned)(granule_load(src0) >> 32) == epoch;
                if (r16 == 1) ready = (unsigned)(granule_load(src1 + (size_t)(NKS / 2) * 64) >> 32) == epoch;
                if (__all(ready)) break;
                if (++spins > spin_budget) { timed_out = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            PROF_E(0); PROF_T(1);
            // (2) speculative sweep: issue all loads, run the MFMAs as the values land, validate the tags last;
            //     a stale granule (rare once the gate has opened) just repeats the sweep
            while (true) {
                u64 gr[NKS];
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) gr[ks] = granule_load((ks < NKS / 2 ? src0 : src1) + (size_t)ks * 64);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
                bool ok = true;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const float a_s = __uint_as_float((unsigned)gr[ks]);
                    ok = ok && ((unsigned)(gr[ks] >> 32) == epoch);
#pragma unroll
                    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_s, wv[ks][t], acc[t], 0, 0, 0);
                }
                if (__all(ok) || timed_out) break;
                if (++spins > spin_budget) { timed_out = true; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            if (timed_out) {                                  // bounded: flag the error and never wait again
                if (lane == 0) atomicExch(a.err, 1 + step);
                spin_budget = 0;
            }
        } else {
            PROF_E(0); PROF_T(1);
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], wv[ks][t], acc[t], 0, 0, 0);
        }
        PROF_E(1); PROF_T(2);
        // ---- K reduction through LDS (double-buffered by step parity: one barrier per step).  Every wave
        // drops, for each finishing wave dw (itself included, so that all register indices stay compile-time
        // constants), the 4 gate values of the accumulator regs dw finishes as one 16-byte store.
        f32x4* redp = reinterpret_cast<f32x4*>(red) + (size_t)(step & 1) * (4 * 4 * NOWN * 64);
#pragma unroll
        for (int dw = 0; dw < 4; ++dw) {
            const int dub = NUB == 2 ? (dw & 1) : dw;
            const int dreg0 = NUB == 2 ? 2 * (dw >> 1) : 0;
#pragma unroll
            for (int o = 0; o < NOWN; ++o)
                redp[((dw * 4 + wave) * NOWN + o) * 64 + lane] =
                    f32x4{acc[0 * NUB + dub][dreg0 + o], acc[1 * NUB + dub][dreg0 + o], acc[2 * NUB + dub][dreg0 + o],
                          acc[3 * NUB + dub][dreg0 + o]};
        }
        __syncthreads();
        f32x4 gate[NOWN];
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            f32x4 v = redp[((wave * 4 + 0) * NOWN + o) * 64 + lane];
#pragma unroll
            for (int sw = 1; sw < 4; ++sw) v += redp[((wave * 4 + sw) * NOWN + o) * 64 + lane];
            gate[o] = v;
        }

        PROF_E(3); PROF_T(4);
        // ---- cell update (register-local), publish h_step, write the layer output
        const size_t doff = (size_t)(step & 1) * 16 * H;
#pragma unroll
        for (int o = 0; o < NOWN; ++o) {
            float oval = 0.f;
            if (act[o]) {
                const float ig = sigmoidf_(gate[o][0] + xp[o][0]);
                const float fg = sigmoidf_(gate[o][1] + xp[o][1]);
                const float gg = tanhf_(gate[o][2] + xp[o][2]);
                const float og = sigmoidf_(gate[o][3] + xp[o][3]);
                cst[o] = fg * cst[o] + ig * gg;
                hst[o] = og * tanhf_(cst[o]);
                oval = hst[o];
            }
            const int gi = granule_index(q * 4 + reg0 + o, jown);
            granule_store_l2(hxL + doff + gi, (unsigned)(step + 1), hst[o]);
            if (!all_local) granule_store(hxR + doff + gi, (unsigned)(step + 1), hst[o]);
            if (bidx[o] < B) d.out[((size_t)tt[o] * B + bidx[o]) * d.outStride + jown] = oval;
        }
        PROF_E(4);
    }
    if (prof) {
        long long* o = a.prof + (size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 6;
        for (int i = 0; i < 5; ++i) o[i] = pt[i];
        o[5] = T;
    }

    // ---- final state (h_n, c_n of models/rnn.py:33) back to hbuf parity 0 / cbuf
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        if (bidx[o] < B) {
            d.hbuf[(size_t)bidx[o] * H + jown] = hst[o];
            d.cbuf[(size_t)bidx[o] * H + jown] = cst[o];
        }
    }
}

// dst[(((slice*4 + wave)*NKS + ks)*NT + t)*64 + lane] = W_hh[g*H + slice*U + ub*16 + (lane&15)][wave*KW + 4*ks + (lane>>4)]
// with t = g*NUB + ub
template <int H, int NSLICE>
__global__ void mp_pack_whh_persist(const float* __restrict__ whh, float* __restrict__ dst) {
    constexpr int U = H / NSLICE, NUB = U / 16, NT = 4 * NUB, KW = H / 4, NKS = KW / 4;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)4 * H * H) return;
    const int lane = idx & 63;
    size_t rest = idx >> 6;
    const int t = rest % NT; rest /= NT;
    const int ks = rest % NKS; rest /= NKS;
    const int wave = rest % 4; rest /= 4;
    const int slice = (int)rest;
    const int g = t / NUB, ub = t % NUB;
    const int row = g * H + slice * U + ub * 16 + (lane & 15);
    const int col = wave * KW + 4 * ks + (lane >> 4);
    dst[idx] = whh[(size_t)row * H + col];
}

}  // namespace

void mp_launch_pack_whh_persist(const float* whh, float* dst, int H, hipStream_t s) {
    const size_t n = (size_t)4 * H * H;
    const int grid = (int)((n + 255) / 256);
    if (H == 256) hipLaunchKernelGGL((mp_pack_whh_persist<256, 8>), dim3(grid), dim3(256), 0, s, whh, dst);
    else hipLaunchKernelGGL((mp_pack_whh_persist<64, 1>), dim3(grid), dim3(256), 0, s, whh, dst);
}

int mp_persist_nslice(int H) { return H == 256 ? 8 : 1; }

void mp_launch_lstm_persist(const LstmPersistArgs& a, int H, hipStream_t s) {
    if (H == 256) hipLaunchKernelGGL((mp_lstm_persist<256, 8>), dim3(a.nslab * 8, a.ndir), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((mp_lstm_persist<64, 1>), dim3(a.nslab, a.ndir), dim3(256), 0, s, a);
}

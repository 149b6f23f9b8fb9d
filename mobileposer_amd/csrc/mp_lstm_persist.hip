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

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}

__device__ __forceinline__ u64 granule_load(const u64* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void granule_store(u64* p, unsigned epoch, float v) {
    __hip_atomic_store(p, ((u64)epoch << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// H: hidden size; NSLICE: workgroups sharing one slab (8 for H = 256, 1 for H = 64)
template <int H, int NSLICE>
__global__ __launch_bounds__(256, 1) void mp_lstm_persist(LstmPersistArgs a) {
    constexpr int U = H / NSLICE;              // hidden units per workgroup (32 | 64)
    constexpr int NUB = U / 16;                // 16-unit blocks per workgroup (2 | 4)
    constexpr int NT = 4 * NUB;                // MFMA tiles per wave: gates x unit blocks (8 | 16)
    constexpr int KW = H / 4;                  // K range of one wave (64 | 16)
    constexpr int NKS = KW / 4;                // k-steps per wave (16 | 4)
    constexpr int NOWN = NUB == 2 ? 2 : 4;     // accumulator regs a lane finishes
    // reduction scratch: red[dst wave][src wave][g*NOWN + o][lane]
    __shared__ __attribute__((aligned(16))) float red[4 * 4 * 4 * NOWN * 64];

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

    // ---- A operand for step 0 from the initial state: h0[row r16][k = wave*KW + q*NKS + ks]
    float av[NKS];
    {
        const int b = brow0 + r16;
        const float* p = d.hbuf + (size_t)(b < B ? b : 0) * H + wave * KW + q * NKS;
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks) av[ks] = b < B ? p[ks] : 0.f;
    }

    // granules of this slab: hx[dir][slab][parity][16 rows][H units]
    u64* hx = a.hx + ((size_t)(blockIdx.y * a.nslab + slab) * 2) * 16 * H;
    unsigned spin_budget = a.max_spin;

    for (int step = 0; step < T; ++step) {
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

        // ---- gather h_{step-1}: the lane's 16 A values are granules [row r16][wave*KW + q*NKS + ks]
        if (step > 0) {
            const unsigned epoch = (unsigned)step;            // written by the producers at the end of step-1
            const u64* src = hx + ((size_t)((step - 1) & 1) * 16 + r16) * H + wave * KW + q * NKS;
            bool ok = false;
            unsigned spins = 0;
            while (true) {
                ok = true;
#pragma unroll
                for (int ks = 0; ks < NKS; ++ks) {
                    const u64 g = granule_load(src + ks);
                    av[ks] = __uint_as_float((unsigned)g);
                    ok = ok && ((unsigned)(g >> 32) == epoch);
                }
                if (__all(ok)) break;
                if (++spins > spin_budget) {                  // bounded: flag the error and stop waiting for good
                    if (lane == 0) atomicExch(a.err, 1 + step);
                    spin_budget = 0;
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
            }
        }

        // ---- partial gates over this wave's K quarter: NKS x NT MFMAs
        f32x4 acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks], wv[ks][t], acc[t], 0, 0, 0);

        // ---- K reduction through LDS: every wave drops, for each finishing wave dw (itself included, so that
        // all register indices stay compile-time constants), the accumulator regs dw finishes
        __syncthreads();                                      // previous step's reads of `red` are done
#pragma unroll
        for (int dw = 0; dw < 4; ++dw) {
            const int dub = NUB == 2 ? (dw & 1) : dw;
            const int dreg0 = NUB == 2 ? 2 * (dw >> 1) : 0;
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int o = 0; o < NOWN; ++o)
                    red[(((dw * 4 + wave) * 4 + g) * NOWN + o) * 64 + lane] = acc[g * NUB + dub][dreg0 + o];
        }
        __syncthreads();
        float gate[NOWN][4];
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int o = 0; o < NOWN; ++o) {
                float v = 0.f;
#pragma unroll
                for (int s = 0; s < 4; ++s) v += red[(((wave * 4 + s) * 4 + g) * NOWN + o) * 64 + lane];
                gate[o][g] = v;
            }

        // ---- cell update (register-local), publish h_step, write the layer output
        u64* dst = hx + ((size_t)(step & 1) * 16) * H;
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
            granule_store(dst + (size_t)(q * 4 + reg0 + o) * H + jown, (unsigned)(step + 1), hst[o]);
            if (bidx[o] < B) d.out[((size_t)tt[o] * B + bidx[o]) * d.outStride + jown] = oval;
        }
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

// dst[(((slice*4 + wave)*NKS + ks)*NT + t)*64 + lane] = W_hh[g*H + slice*U + ub*16 + (lane&15)][wave*KW + (lane>>4)*NKS + ks]
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
    const int col = wave * KW + (lane >> 4) * NKS + ks;
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

// K2 -- one time step of the nn.LSTM recurrence (models/rnn.py:27) for up to two (layer, direction)
// instances at once, plus the weight re-layout kernels run once at model load.
//
// Per step and direction:  gates[B,4H] = xproj[t] (W_ih x + b_ih + b_hh, from K1) + h_prev[B,H] W_hh^T;
// i,f,g,o = sigma,sigma,tanh,sigma;  c' = f c + i g;  h' = o tanh(c')      (PyTorch gate order i,f,g,o).
// Packed-sequence semantics (models/rnn.py:25-31): sequence b is active for steps
// s < len_b; forward visits t = s, reverse visits t = len_b-1-s; inactive rows keep (h,c) and the
// padded output position t = s is written as 0 so that linear2 yields its bias there.
//
// gfx950 mapping: the batch is cut into slabs of 16 sequences (the M of v_mfma_f32_16x16x4_f32), the
// hidden units into slices of 16*UBW units; one 4-wave workgroup owns (direction, slab, slice) and
// computes its 16 x (4 gates x 16*UBW units) gate tile with exact-fp32 MFMA.  For H = 256 a workgroup
// owns 32 units (8 slices; blockIdx.x % 8 = slice = XCD, so each XCD's L2 keeps one 128 KB W_hh slice
// per direction) and splits K = 256 over two wave pairs (partial sums meet in LDS); for H = 64 it owns
// all 64 units and no K split is needed.  W_hh is pre-packed in B-fragment order with the 4 gates of a
// unit adjacent, so a lane streams it as fully coalesced 16-byte loads and holds i,f,g,o of one
// (sequence, unit) in the same accumulator position: the cell update is register-local.
// A lane's k values are a contiguous run (lane group q = lane>>4 takes k = q*KW/4 + ks), so h_prev rows
// are read as 16-byte loads too.  Steps are separate launches replayed from a hipGraph: on this chip a
// dependent kernel boundary (~1.5 us) is cheaper than a software grid barrier (~4 us, MI355X_MICROARCH
// price list), and the sequences are independent, so there is no other cross-workgroup traffic.
#include "mp_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float tanhf_(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1): exact to ~1e-7 abs, saturates correctly for large |x|
    const float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}

// H: hidden size; KSPLIT: wave groups splitting K (2 for H=256, 1 for H=64); UBW = 4/KSPLIT unit blocks
template <int H, int KSPLIT>
MP_KERNEL __launch_bounds__(256) void mp_lstm_step(LstmStepArgs a) {
    constexpr int UBW = 4 / KSPLIT;            // waves along units
    constexpr int UNITS = 16 * UBW;            // hidden units per workgroup
    constexpr int NSLICE = H / UNITS;
    constexpr int KW = H / KSPLIT;             // K range of one wave
    constexpr int NKS = KW / 4;                // MFMA k-steps per wave
    constexpr int AQ = NKS / 4;                // float4 loads of h_prev per lane

    __shared__ __attribute__((aligned(16))) float red[KSPLIT == 2 ? 4 * 64 * 8 : 4];

    const LstmDir d = a.d[blockIdx.y];
    const int slice = blockIdx.x % NSLICE;
    const int slab = blockIdx.x / NSLICE;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ub = wave % UBW, kh = wave / UBW;
    const int q = lane >> 4, r16 = lane & 15;
    const int B = a.B, step = a.step;

    const float* hprev = d.hbuf + (size_t)(step & 1) * B * H;
    float* hnext = d.hbuf + (size_t)((step + 1) & 1) * B * H;

    // ---- A fragments: h_prev[slab*16 + r16][kh*KW + q*NKS + ks], ks = 0..NKS-1
    f32x4 av[AQ];
    {
        const int b = slab * 16 + r16;
        const float* p = hprev + (size_t)(b < B ? b : 0) * H + kh * KW + q * NKS;
#pragma unroll
        for (int i = 0; i < AQ; ++i) {
            av[i] = *reinterpret_cast<const f32x4*>(p + 4 * i);
            if (b >= B) av[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }

    // ---- rows this wave finishes: with a K split, wave group kh owns accumulator regs {2kh, 2kh+1}
    constexpr int NOWN = 4 / KSPLIT;
    const int j = slice * UNITS + ub * 16 + r16;            // hidden unit of this lane
    f32x4 xp[NOWN];
    float cold[NOWN], hold[NOWN];
    int tt[NOWN];
    bool act[NOWN], inb[NOWN];
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        const int reg = (KSPLIT == 2 ? 2 * kh : 0) + o;
        const int b = slab * 16 + q * 4 + reg;
        inb[o] = b < B;
        const int len = inb[o] ? a.lengths[b] : 0;
        act[o] = step < len;
        tt[o] = act[o] ? (d.reverse ? len - 1 - step : step) : step;
        xp[o] = f32x4{0.f, 0.f, 0.f, 0.f};
        cold[o] = 0.f;
        hold[o] = 0.f;
        if (act[o]) {
            xp[o] = *reinterpret_cast<const f32x4*>(d.xproj + ((size_t)tt[o] * B + b) * d.xprojStride + 4 * j);
            cold[o] = d.cbuf[(size_t)b * H + j];
        } else if (inb[o]) {
            hold[o] = hprev[(size_t)b * H + j];
        }
    }

    // ---- gates += h_prev W_hh^T : NKS k-steps x 4 gate tiles of v_mfma_f32_16x16x4_f32
    f32x4 acc[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[g] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f32x4* wp = reinterpret_cast<const f32x4*>(d.wpack) +
                      ((size_t)((slice * UBW + ub) * KSPLIT + kh) * NKS) * 64 + lane;
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks) {
        const f32x4 w = wp[(size_t)ks * 64];
        const float av_s = av[ks >> 2][ks & 3];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_s, w[g], acc[g], 0, 0, 0);
    }

    // ---- combine the two K halves: each wave hands over the accumulator regs its partner owns
    float gate[NOWN][4];
    if constexpr (KSPLIT == 2) {
        // red[(ub*2 + kh_owner)][g*2 + o][lane]
        const int other = 1 - kh;
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int o = 0; o < 2; ++o)
                red[((ub * 2 + other) * 8 + g * 2 + o) * 64 + lane] = acc[g][2 * other + o];
        __syncthreads();
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int o = 0; o < 2; ++o)
                gate[o][g] = acc[g][2 * kh + o] + red[((ub * 2 + kh) * 8 + g * 2 + o) * 64 + lane];
    } else {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int o = 0; o < 4; ++o) gate[o][g] = acc[g][o];
    }

    // ---- cell update, register-local
#pragma unroll
    for (int o = 0; o < NOWN; ++o) {
        if (!inb[o]) continue;
        const int reg = (KSPLIT == 2 ? 2 * kh : 0) + o;
        const int b = slab * 16 + q * 4 + reg;
        float hval, oval;
        if (act[o]) {
            const float ig = sigmoidf_(gate[o][0] + xp[o][0]);
            const float fg = sigmoidf_(gate[o][1] + xp[o][1]);
            const float gg = tanhf_(gate[o][2] + xp[o][2]);
            const float og = sigmoidf_(gate[o][3] + xp[o][3]);
            const float cn = fg * cold[o] + ig * gg;
            hval = og * tanhf_(cn);
            oval = hval;
            d.cbuf[(size_t)b * H + j] = cn;
        } else {
            hval = hold[o];
            oval = 0.f;
        }
        hnext[(size_t)b * H + j] = hval;
        if (tt[o] < a.T) d.out[((size_t)tt[o] * B + b) * d.outStride + j] = oval;
    }
}

// ---- weight re-layout (once per model load) ------------------------------------------------------
// dst[((((slice*UBW + ub)*KSPLIT + kh)*NKS + ks)*64 + lane)*4 + g]
//   = W_hh[g*H + slice*UNITS + ub*16 + (lane&15)][kh*KW + (lane>>4)*NKS + ks]
template <int H, int KSPLIT>
MP_KERNEL void mp_pack_whh(const float* __restrict__ whh, float* __restrict__ dst) {
    constexpr int UBW = 4 / KSPLIT, UNITS = 16 * UBW, KW = H / KSPLIT, NKS = KW / 4;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)4 * H * H) return;
    const int g = idx & 3;
    const int lane = (idx >> 2) & 63;
    size_t rest = idx >> 8;
    const int ks = rest % NKS; rest /= NKS;
    const int kh = rest % KSPLIT; rest /= KSPLIT;
    const int ub = rest % UBW; rest /= UBW;
    const int slice = (int)rest;
    const int row = g * H + slice * UNITS + ub * 16 + (lane & 15);
    const int col = kh * KW + (lane >> 4) * NKS + ks;
    dst[idx] = whh[(size_t)row * H + col];
}

// W_ih [4H][K] (rows g*H + j) -> rows dirOff + 4*j + g of dstW [.][Kpad]; bias = b_ih + b_hh
MP_KERNEL void mp_pack_wih(const float* __restrict__ wih, const float* __restrict__ bih,
                            const float* __restrict__ bhh, float* __restrict__ dstW, float* __restrict__ dstBias,
                            int H, int K, int Kpad, int dirOff) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)4 * H * K) return;
    const int k = idx % K;
    const int srow = idx / K;
    const int g = srow / H, j = srow % H;
    const int drow = dirOff + 4 * j + g;
    dstW[(size_t)drow * Kpad + k] = wih[idx];
    if (k == 0) dstBias[drow] = bih[srow] + bhh[srow];
}

MP_KERNEL void mp_pack_linear(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ dstW,
                               float* __restrict__ dstBias, int N, int K, int Kpad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)N * K) return;
    const int k = idx % K, n = idx / K;
    dstW[(size_t)n * Kpad + k] = w[idx];
    if (k == 0) dstBias[n] = b[n];
}

}  // namespace

size_t mp_whh_pack_floats(int H) { return (size_t)4 * H * H; }

void mp_launch_pack_whh(const float* whh, float* dst, int H, hipStream_t s) {
    const size_t n = (size_t)4 * H * H;
    const int grid = (int)((n + 255) / 256);
    if (H == 256) hipLaunchKernelGGL((mp_pack_whh<256, 2>), dim3(grid), dim3(256), 0, s, whh, dst);
    else hipLaunchKernelGGL((mp_pack_whh<64, 1>), dim3(grid), dim3(256), 0, s, whh, dst);
}

void mp_launch_pack_wih(const float* wih, const float* bih, const float* bhh, float* dstW, float* dstBias, int H,
                        int K, int Kpad, int dirOff, hipStream_t s) {
    const size_t n = (size_t)4 * H * K;
    hipLaunchKernelGGL(mp_pack_wih, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wih, bih, bhh, dstW, dstBias,
                       H, K, Kpad, dirOff);
}

void mp_launch_pack_linear(const float* w, const float* b, float* dstW, float* dstBias, int N, int K, int Kpad,
                           hipStream_t s) {
    const size_t n = (size_t)N * K;
    hipLaunchKernelGGL(mp_pack_linear, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, w, b, dstW, dstBias, N, K,
                       Kpad);
}

void mp_launch_lstm_step(const LstmStepArgs& a, int H, hipStream_t s) {
    const int nslab = (a.B + 15) / 16;
    if (H == 256) hipLaunchKernelGGL((mp_lstm_step<256, 2>), dim3(nslab * 8, a.ndir), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((mp_lstm_step<64, 1>), dim3(nslab, a.ndir), dim3(256), 0, s, a);
}

// NOTE (round 4): the 16-bit halves of a split operand are IEEE fp16 now (mp_lstm_dev.h pair_of: 24-bit operands, weights split
// as 16 w, MFMA v_mfma_f32_16x16x32_f16), not bf16 as in rounds 1-3 when this file was written; "bf16" in the comments below
// describes the same data path with the other half format.  The hidden-state tag sits in bit 30 of the word (hpair_of).
// K1x/K3x -- the linear layers (models/rnn.py:22,32; the fused torch.cat of net.py:106,113) with SPLIT-bf16 MFMA
// operands, for the default LSTM mode: same arithmetic contract as mp_lstm_x3.hip (every fp32 product a*w as
// a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on v_mfma_f32_16x16x32_bf16, fp32 accumulate, fp32 bias / ReLU / output).
// Why: mp_gemm_f32 leaves the fp32 matrix pipe 26-33 % busy on these shapes (M = B*T = 32000, N = 72..256,
// K = 60..512: one 128-row tile per CU, 64-cycle MFMAs between two barriers per k-tile) and the four linear layers
// on the critical path of a forward cost 170 us of 2.2 ms; with 3 x 19-cycle MFMAs per 32 k the same tile loop is
// bound by its loads.
// A operand: two K segments (RowMap) of fp32 values, converted to pair words while they are staged to LDS, or of pair
// words already (the layer-1 output of mp_lstm_x3, aPairs).  W: pair words [Npad][Kpad] (mp_launch_pairs).
// 64 x BN block tile (BM is a template parameter), 32-wide k-chunks, 4 waves (16 rows x BN each), LDS pitch 36 words,
// register prefetch of the next chunk, XCD-aware tile order as mp_gemm_f32.
#include "mp_lstm_dev.h"

namespace {

constexpr int BK = 32, LDK = 36;

template <int BN, int BM>
MP_KERNEL __launch_bounds__(256) void mp_gemm_x3(GemmArgs g, int nTilesM, int nTilesN) {
    constexpr int NCT = BN / 16, NRT = BM / 64;                 // 16-wide column tiles, 16-row tiles per wave
    constexpr int A_ROWS_PER_THREAD = BM / 32, W_ROWS_PER_THREAD = BN / 32;
    // (the output tile is staged in the same LDS after the k loop: BM x (BN + 4) words)
    constexpr int OPITCH = BN + 4;
    constexpr int SMEM_MAIN = (BM + BN) * LDK, SMEM_OUT = BM * OPITCH;
    __shared__ __attribute__((aligned(16))) unsigned smem[(SMEM_MAIN > SMEM_OUT ? SMEM_MAIN : SMEM_OUT) + 2 * BM];
    unsigned* As = smem;
    unsigned* Ws = smem + BM * LDK;
    long* rowOffC = reinterpret_cast<long*>(smem + (SMEM_MAIN > SMEM_OUT ? SMEM_MAIN : SMEM_OUT));

    const int bid = blockIdx.x;
    if (g.zero_ncl > 0) rearm_exchange(g.zero_hx, g.zero_ncl, bid, gridDim.x, threadIdx.x, 256);
    if (g.zero_ncl2 > 0) rearm_exchange(g.zero_hx2, g.zero_ncl2, bid, gridDim.x, threadIdx.x, 256);
    const int xcd = bid & 7, idx = bid >> 3;
    const int mt = (idx / nTilesN) * 8 + xcd;
    const int nt = idx % nTilesN;
    if (mt >= nTilesM) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lr = tid >> 3;            // row within a 32-row group
    const int kc = (tid & 7) * 4;       // k column of this thread's 4 words

    long offA0[A_ROWS_PER_THREAD], offA1[A_ROWS_PER_THREAD];
    bool rowOk[A_ROWS_PER_THREAD];
#pragma unroll
    for (int j = 0; j < A_ROWS_PER_THREAD; ++j) {
        const int m = m0 + lr + 32 * j;
        rowOk[j] = m < g.M;
        const int mm = rowOk[j] ? m : 0;
        const int b = mm % g.B, t = mm / g.B;
        offA0[j] = (long)b * g.a0.strideB + (long)t * g.a0.strideT;
        offA1[j] = (long)b * g.a1.strideB + (long)t * g.a1.strideT;
    }
    if (tid < BM) {
        const int m = m0 + tid;
        const int mm = m < g.M ? m : 0;
        rowOffC[tid] = (long)(mm % g.B) * g.cStrideB + (long)(mm / g.B) * g.cStrideT;
    }

    u32x4 ra[A_ROWS_PER_THREAD], rw[W_ROWS_PER_THREAD];
    bool ra_live = false;                                     // k range of the prefetched A words inside [0, K)
    const unsigned* Wp = reinterpret_cast<const unsigned*>(g.W);
    auto load_tile = [&](int k0) {
        const int k = k0 + kc;
        // unconditional loads from a clamped (always valid) address, then a select: loads under branches are serialised by
        // the compiler's wait-count merging at every join (load, vmcnt(0), load, vmcnt(0) ... in the ISA), which is fatal
        // for a loop that is bound by load latency.  fp32 zero and the zero pair are the same bits.
        const bool in0 = k < g.a0.width, in1 = !in0 && k < g.K;
#pragma unroll
        for (int j = 0; j < A_ROWS_PER_THREAD; ++j) {
            const float* src = in1 ? g.a1.base + offA1[j] + (k - g.a0.width) : g.a0.base + offA0[j] + (in0 ? k : 0);
            ra[j] = *reinterpret_cast<const u32x4*>(src);     // (masked when it is stored to LDS: a select here would be a use)
        }
        ra_live = in0 || in1;
#pragma unroll
        for (int j = 0; j < W_ROWS_PER_THREAD; ++j)
            rw[j] = *reinterpret_cast<const u32x4*>(Wp + (long)(n0 + lr + 32 * j) * g.Kpad + k);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < A_ROWS_PER_THREAD; ++j) {
            u32x4 v = (rowOk[j] && ra_live) ? ra[j] : u32x4{0u, 0u, 0u, 0u};
            if (!g.aPairs) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = pair_of(__uint_as_float(v[e]));
            }
            *reinterpret_cast<u32x4*>(As + (lr + 32 * j) * LDK + kc) = v;
        }
#pragma unroll
        for (int j = 0; j < W_ROWS_PER_THREAD; ++j)
            *reinterpret_cast<u32x4*>(Ws + (lr + 32 * j) * LDK + kc) = rw[j];
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int r16 = lane & 15, q = lane >> 4;
    f32x4 acc[NRT][NCT];
#pragma unroll
    for (int a = 0; a < NRT; ++a)
#pragma unroll
        for (int b = 0; b < NCT; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nk = g.Kpad / BK;
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        u32x4 ahi[NRT], alo[NRT];
#pragma unroll
        for (int a = 0; a < NRT; ++a) {
            const unsigned* p = As + (wave * (BM / 4) + a * 16 + r16) * LDK + q * 8;
            split_pairs(*reinterpret_cast<const u32x4*>(p), *reinterpret_cast<const u32x4*>(p + 4), ahi[a], alo[a]);
        }
#pragma unroll
        for (int b = 0; b < NCT; ++b) {
            const unsigned* p = Ws + (b * 16 + r16) * LDK + q * 8;
            u32x4 whi, wlo;
            split_pairs(*reinterpret_cast<const u32x4*>(p), *reinterpret_cast<const u32x4*>(p + 4), whi, wlo);
#pragma unroll
            for (int a = 0; a < NRT; ++a) acc[a][b] = mfma_bf16(ahi[a], whi, acc[a][b]);
#pragma unroll
            for (int a = 0; a < NRT; ++a) acc[a][b] = mfma_bf16(ahi[a], wlo, acc[a][b]);
#pragma unroll
            for (int a = 0; a < NRT; ++a) acc[a][b] = mfma_bf16(alo[a], whi, acc[a][b]);
#pragma unroll
            for (int a = 0; a < NRT; ++a) acc[a][b] = mfma_bf16(alo[a], wlo, acc[a][b]);      // lo*lo (round 6)
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }

    // epilogue: D[row = 4*q + reg][col = r16] of tile (a, b); bias (+ReLU), fp32 or pair-word output.  The tile goes through
    // LDS so that a wave writes whole row segments (16 bytes per lane, 512 contiguous bytes per row of a 128-wide tile)
    // instead of 64-byte pieces of four rows per store instruction: these GEMMs are bound by their output traffic.
    unsigned* Os = smem;                                  // (the k loop ended with a barrier: A / W tiles are dead)
#pragma unroll
    for (int b = 0; b < NCT; ++b) {
        const int n = n0 + b * 16 + r16;
        const float bias = n < g.N ? g.bias[n] : 0.f;
#pragma unroll
        for (int a = 0; a < NRT; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ml = wave * (BM / 4) + a * 16 + 4 * q + r;
                float v = acc[a][b][r] * kPairWInv + bias;      // (W was split as 16 W: mp_lstm_dev.h pair_of)
                if (g.relu) v = relu_(v);
                Os[ml * OPITCH + b * 16 + r16] = g.pairOut ? pair_of(v) : __float_as_uint(v);
            }
    }
    __syncthreads();
    {
        const bool second = g.nsplit > 0 && n0 >= g.nsplit;          // (uniform per block: BN divides nsplit)
        float* Cb = second ? g.C2 : g.C;
        const int nb = second ? n0 - g.nsplit : n0;                  // first output column of the tile in its matrix
        const int nvalid = g.N - n0 < BN ? g.N - n0 : BN;            // columns of the tile that exist
        constexpr int V4 = BN / 4;
        for (int idx = tid; idx < BM * V4; idx += 256) {
            const int ml = idx / V4, c4 = (idx % V4) * 4;
            if (m0 + ml >= g.M || c4 >= nvalid) continue;
            float* dst = Cb + rowOffC[ml] + nb + c4;
            const u32x4 v = *reinterpret_cast<const u32x4*>(Os + ml * OPITCH + c4);
            if (c4 + 4 <= nvalid && (reinterpret_cast<size_t>(dst) & 15) == 0) {
                *reinterpret_cast<u32x4*>(dst) = v;
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (c4 + e < nvalid) dst[e] = __uint_as_float(v[e]);
            }
        }
    }
}

MP_KERNEL void mp_pairs(const float* __restrict__ src, unsigned* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = wpair_of(src[i]);                      // weights only (mp_handle.hip pack_weights)
}

template <int BN, int BM>
void launch(const GemmArgs& g, hipStream_t s) {
    const int nTilesM = (g.M + BM - 1) / BM;
    const int nTilesN = (g.N + BN - 1) / BN;
    const int grid = ((nTilesM + 7) / 8) * 8 * nTilesN;
    hipLaunchKernelGGL((mp_gemm_x3<BN, BM>), dim3(grid), dim3(256), 0, s, g, nTilesM, nTilesN);
}

}  // namespace

// g.W: pair words of the padded weight matrix (mp_launch_pairs); bn as mp_gemm_pick_bn
void mp_launch_gemm_x3(const GemmArgs& g, int bn, hipStream_t s) {
    // 64-row tiles: 500 blocks for M = 32000, two or more per CU -- the tile loop is latency-bound once the MFMAs are cheap
    // (measured 29 vs 36 us per launch against 128-row tiles)
    if (bn == 128) launch<128, 64>(g, s);
    else if (bn == 96) launch<96, 64>(g, s);
    else launch<32, 64>(g, s);
}

void mp_launch_pairs(const float* src, float* dst, size_t n, hipStream_t s) {
    hipLaunchKernelGGL(mp_pairs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, reinterpret_cast<unsigned*>(dst), n);
}

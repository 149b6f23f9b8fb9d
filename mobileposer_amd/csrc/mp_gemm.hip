// K1/K3 -- fp32 MFMA GEMM for the batched (non-recurrent) 60 % of the MobilePoser FLOPs:
//   linear1+ReLU (models/rnn.py:22), the W_ih input projections of nn.LSTM for all B*T frames at once
//   (models/rnn.py:27), linear2 (models/rnn.py:32), and the fused torch.cat((pred_joints, imu))
//   of models/net.py:106,113 as two K-segments of the A operand.
//
// gfx950 design: v_mfma_f32_32x32x2_f32 (exact fp32, 157 TFLOP/s chip peak) -- bf16 fails the 1e-4
// parity bound.  128 x BN block tile, BK = 32, 4 waves; A and W tiles staged through LDS
// with 16-byte stores/loads (row pitch 36 floats: ds_read_b128 of 16 consecutive rows is conflict-free).
// The two k values one MFMA consumes are taken 16 apart (lanes 0-31: k = s, lanes 32-63: k = 16+s), so a
// lane reads its 16 k-values of a tile row as 4 contiguous ds_read_b128 -- any k pairing is valid as long as
// A and W use the same one.  Register prefetch of the next k-tile overlaps HBM/L2 latency with the MFMAs.
// blockIdx -> tile mapping keeps all n-tiles of one m-tile on one XCD (same L2) so the A panel is fetched
// from HBM once.
#include "mp_lstm_dev.h"

namespace {

constexpr int BK = 32;
constexpr int LDK = 36;   // LDS row pitch in floats (144 B: 16-B aligned, conflict-free b128 reads)

template <int WAVES_M, int WAVES_N, int TM, int TN>
MP_KERNEL __launch_bounds__(256) void mp_gemm_f32(GemmArgs g, int nTilesM, int nTilesN) {
    constexpr int BM = WAVES_M * TM * 32;
    constexpr int BN = WAVES_N * TN * 32;
    static_assert(BM == 128, "BM is 128");
    constexpr int A_ROWS_PER_THREAD = BM / 32;   // 4
    constexpr int W_ROWS_PER_THREAD = BN / 32;   // 4, 3 or 1

    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDK + 2 * BM];
    float* As = smem;
    float* Ws = smem + BM * LDK;
    long* rowOffC = reinterpret_cast<long*>(smem + (BM + BN) * LDK);

    // XCD-aware tile order: block id -> (xcd, idx); every XCD walks its own m-tiles, n fastest.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int mt = (idx / nTilesN) * 8 + xcd;
    const int nt = idx % nTilesN;
    if (mt >= nTilesM) return;
    const int m0 = mt * BM, n0 = nt * BN;

    const int tid = threadIdx.x;
    const int lr = tid >> 3;            // 0..31 row within a 32-row group
    const int kc = (tid & 7) * 4;       // k column of this thread's float4

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

    f32x4 ra[A_ROWS_PER_THREAD], rw[W_ROWS_PER_THREAD];
    auto load_tile = [&](int k0) {
        const int k = k0 + kc;
#pragma unroll
        for (int j = 0; j < A_ROWS_PER_THREAD; ++j) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (rowOk[j]) {
                if (k < g.a0.width) v = *reinterpret_cast<const f32x4*>(g.a0.base + offA0[j] + k);
                else if (k < g.K)   v = *reinterpret_cast<const f32x4*>(g.a1.base + offA1[j] + (k - g.a0.width));
            }
            ra[j] = v;
        }
#pragma unroll
        for (int j = 0; j < W_ROWS_PER_THREAD; ++j)
            rw[j] = *reinterpret_cast<const f32x4*>(g.W + (long)(n0 + lr + 32 * j) * g.Kpad + k);
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < A_ROWS_PER_THREAD; ++j)
            *reinterpret_cast<f32x4*>(As + (lr + 32 * j) * LDK + kc) = ra[j];
#pragma unroll
        for (int j = 0; j < W_ROWS_PER_THREAD; ++j)
            *reinterpret_cast<f32x4*>(Ws + (lr + 32 * j) * LDK + kc) = rw[j];
    };

    const int wave = tid >> 6, lane = tid & 63;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int li = lane & 31, lh = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int nk = g.Kpad / BK;
    load_tile(0);
    store_tile();
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
        f32x4 fa[TM][4], fb[TN][4];
#pragma unroll
        for (int a = 0; a < TM; ++a)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                fa[a][q] = *reinterpret_cast<const f32x4*>(As + ((wm * TM + a) * 32 + li) * LDK + lh * 16 + q * 4);
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                fb[b][q] = *reinterpret_cast<const f32x4*>(Ws + ((wn * TN + b) * 32 + li) * LDK + lh * 16 + q * 4);
#pragma unroll
        for (int s = 0; s < 16; ++s)
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][s >> 2][s & 3], fb[b][s >> 2][s & 3],
                                                                     acc[a][b], 0, 0, 0);
        __syncthreads();
        if (kt + 1 < nk) {
            store_tile();
            __syncthreads();
        }
    }

    // epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]; bias (+ReLU), row-mapped store
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + (wn * TN + b) * 32 + li;
        if (n >= g.N) continue;
        const float bias = g.bias[n];
#pragma unroll
        for (int a = 0; a < TM; ++a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (m0 + ml < g.M) {
                    float v = acc[a][b][r] + bias;
                    if (g.relu) v = fmaxf(v, 0.f);
                    g.C[rowOffC[ml] + n] = g.pairOut ? __uint_as_float(pair_of(v)) : v;
                }
            }
        }
    }
}

template <int WAVES_M, int WAVES_N, int TM, int TN>
void launch(const GemmArgs& g, hipStream_t s) {
    constexpr int BN = WAVES_N * TN * 32;
    const int nTilesM = (g.M + 127) / 128;
    const int nTilesN = (g.N + BN - 1) / BN;
    const int grid = ((nTilesM + 7) / 8) * 8 * nTilesN;
    hipLaunchKernelGGL((mp_gemm_f32<WAVES_M, WAVES_N, TM, TN>), dim3(grid), dim3(256), 0, s, g, nTilesM, nTilesN);
}

}  // namespace

int mp_gemm_pick_bn(int N) { return N > 96 ? 128 : (N > 32 ? 96 : 32); }

void mp_launch_gemm(const GemmArgs& g, int bn, hipStream_t s) {
    if (bn == 128) launch<2, 2, 2, 2>(g, s);
    else if (bn == 96) launch<4, 1, 1, 3>(g, s);
    else launch<4, 1, 1, 1>(g, s);
}

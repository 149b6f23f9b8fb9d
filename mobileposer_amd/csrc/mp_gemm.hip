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
#include <cstdlib>
#include <cstring>

#ifndef MP_GEMM_EXP
#define MP_GEMM_EXP 0      // timing experiments of tools/micro/gemm_bench.hip (1: no output stores, 2: no operand loads after the first k-tile,
                           // 3: no barriers / LDS refills between k-tiles, 4: no MFMAs)
#endif

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
#if MP_GEMM_EXP != 2
        if (kt + 1 < nk) load_tile((kt + 1) * BK);
#endif
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
#if MP_GEMM_EXP == 4
                    acc[a][b][s & 15] += fa[a][s >> 2][s & 3] * fb[b][s >> 2][s & 3];
#else
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][s >> 2][s & 3], fb[b][s >> 2][s & 3],
                                                                     acc[a][b], 0, 0, 0);
#endif
#if MP_GEMM_EXP != 3
        __syncthreads();
        if (kt + 1 < nk) {
            store_tile();
            __syncthreads();
        }
#endif
    }

    // epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]; bias (+ReLU), row-mapped store
    // (two stacked linear layers in one launch: output columns [nsplit, N) go to C2 as its columns [0, N - nsplit))
    float* const Cb = (g.nsplit > 0 && n0 >= g.nsplit) ? g.C2 : g.C;       // uniform per block: BN divides nsplit
    const int ncol0 = (g.nsplit > 0 && n0 >= g.nsplit) ? g.nsplit : 0;
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
                    if (g.relu) v = relu_(v);
#if MP_GEMM_EXP == 1
                    if (v == 123456.f)
#endif
                    Cb[rowOffC[ml] + (n - ncol0)] = g.pairOut ? __uint_as_float(pair_of(v)) : v;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// mp_gemm_f32_rows<TN, NK> -- the same GEMM for the linear2 shapes of this path (M = B*T rows in the tens of thousands,
// K = 128 .. 512, N = 2 .. 96): no LDS, no barrier, every WAVE on its own.  A wave owns 32 rows x TN*32 columns and
// streams K through registers: per 32-wide k-tile and lane 4 x 16 bytes of its A row and TN x 4 x 16 bytes of W rows
// (W -- at most 0.2 MB -- is read by every wave and lives in L1 / L2), consumed in four slots of 4*TN MFMAs; the registers
// of a slot are re-requested as soon as its MFMAs are issued (A two k-tiles ahead, W one).  With the LDS-staged kernel
// above these launches are latency chains -- one workgroup per CU at most, load -> LDS -> barrier -> MFMA per k-tile.
// Measured inside a 256 x 125 forward (tools/debug/timeline.py): linear2 of joints 63 -> 46 us, of velocity 49 -> 39 us,
// of foot contact 13 -> 10 us; 1024 x 125: 15.6 -> 15.25 ms.  (Still 2x the 20 us of matrix work: a lane reads 16 bytes
// per row and instruction, so every 128-byte line goes through the L1 tag lookup four times.)  The wide linear1 outputs
// (N = 256) are store-bound and did no better this way (86 vs 54 us for pose's linear1): they stay on the staged kernel.
// Same k pairing as above (lanes 0-31: k = s, lanes 32-63: k = 16 + s), so the summation order -- and every result
// bit -- is that of mp_gemm_f32.
template <int TN, int NK>
MP_KERNEL __launch_bounds__(256, 2) void mp_gemm_f32_rows(GemmArgs g, int nTilesM, int nTilesN) {
    __shared__ long rowOffC[4][32];
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int mt = (idx / nTilesN) * 8 + xcd;               // all n-tiles of an m-tile on one XCD (A panel from HBM once)
    const int nt = idx % nTilesN;
    if (mt >= nTilesM) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = mt * 128 + wave * 32, n0 = nt * (TN * 32);
    if (m0 >= g.M) return;

    // this lane's A row (clamped: rows past M are computed and not stored), its two segments
    const int m = m0 + li < g.M ? m0 + li : g.M - 1;
    const int rb = m % g.B, rt = m / g.B;
    const float* pa0 = g.a0.base + (long)rb * g.a0.strideB + (long)rt * g.a0.strideT;
    const float* pa1 = g.a1.base ? g.a1.base + (long)rb * g.a1.strideB + (long)rt * g.a1.strideT - g.a0.width : pa0;
    if (lane < 32) rowOffC[wave][lane] = (long)rb * g.cStrideB + (long)rt * g.cStrideT;
    const int w0 = g.a0.width, klast = g.K - 4;
    // (k >= K only in the last k-tile: W is zero there, so A may be anything finite -- the row's last four values again)
    auto a_piece = [&](int k) {
        const int kk = k < klast ? k : klast;
        return *reinterpret_cast<const f32x4*>((kk < w0 ? pa0 : pa1) + kk);
    };
    const float* pw = g.W + (long)(n0 + li) * g.Kpad + lh * 16;

    f32x16 acc[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    // The loads are inline asm and the waits are placed by hand: left to the compiler, the loop gets ONE s_waitcnt
    // vmcnt(0) at its top, i.e. the full latency of the group requested last, once per k-tile.  Before the MFMAs of a
    // group everything except the three younger groups must have arrived: vmcnt(3 * (1 + TN)).
    // A comes from HBM (it is streamed once), W from L2: the A pieces are requested TWO k-tiles ahead, W one; two A buffers
    // take turns (even / odd k-tiles) and every buffer is re-requested right after the MFMAs that read it.  The k loop is
    // unrolled completely (NK is a template parameter): in straight-line code the compiler's s_waitcnt counts are exact
    // (vmcnt(3 * (1 + TN)) in front of every slot); around a real loop it merges the states at the loop header into ONE
    // vmcnt(0) at the top, which puts the latency of the group requested last on every k-tile.
    f32x4 fa[2][4], fw[TN][4];
    auto request_a = [&](f32x4& dst, int q, int k0) {            // A piece q of the k-tile at k0 (past K: the row's last piece)
        const int k = k0 + lh * 16 + q * 4;
        const int kk = k < klast ? k : klast;
        dst = *reinterpret_cast<const f32x4*>((kk < w0 ? pa0 : pa1) + kk);
    };
    auto request_w = [&](int q, int k0) {
#pragma unroll
        for (int b = 0; b < TN; ++b) fw[b][q] = *reinterpret_cast<const f32x4*>(pw + (long)b * 32 * g.Kpad + k0 + q * 4);
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) request_a(fa[0][q], q, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { request_w(q, 0); request_a(fa[1][q], q, BK); }
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int b = 0; b < TN; ++b)
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kt & 1][q][s4], fw[b][q][s4], acc[b], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 2 < NK) request_a(fa[kt & 1][q], q, (kt + 2) * BK);
            if (kt + 1 < NK) request_w(q, (kt + 1) * BK);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // epilogue: D[row = (r&3) + 8*(r>>2) + 4*(lane>>5)][col = lane&31]; bias (+ReLU), row-mapped store
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // rowOffC of this wave (written by its own lanes)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + b * 32 + li;
        if (n >= g.N) continue;
        const float bias = g.bias[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m0 + ml < g.M) {
                float v = acc[b][r] + bias;
                if (g.relu) v = relu_(v);
                g.C[rowOffC[wave][ml] + n] = g.pairOut ? __uint_as_float(pair_of(v)) : v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// mp_gemm_f32_frag<TN, NK> (round 4) -- mp_gemm_f32_rows with W read in MFMA B-fragment order (GemmArgs::Wf).
// What bounded the rows kernel: a lane's 16-byte piece of "its" W row sits 4 * Kpad bytes from its neighbour's, so one wave-wide
// W load touches 64 different 128-byte lines, and with four waves per CU re-reading the same W the L1 does 4 096 tag look-ups
// per k-tile and CU against 3 072 cycles of MFMAs.  In fragment order the 64 pieces of a load are one contiguous KB (8 lines):
// ~600 look-ups per k-tile and CU, A included.  Same k pairing, same MFMA order: bit-identical to both older kernels.
// Also serves the wide linear1 shapes (N = 256 / 512: TN = 4, n-tiles of an m-tile on one XCD so that A comes from L2).
template <int TN, int NK>
__device__ __forceinline__ void gemm_frag_body(const GemmArgs& g, int bid, int nTilesM, int nTilesN, long (*rowOffC)[32]) {
    const int xcd = bid & 7, idx = bid >> 3;
    const int mt = (idx / nTilesN) * 8 + xcd;               // all n-tiles of an m-tile on one XCD (A panel from HBM once)
    const int nt = idx % nTilesN;
    if (mt >= nTilesM) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = mt * 128 + wave * 32, n0 = nt * (TN * 32);
    if (m0 >= g.M) return;

    // which of the stacked outputs this block's columns belong to (uniform per block: TN*32 divides the split points)
    float* Cb = g.C;
    int ncol0 = 0;
    long sB = g.cStrideB, sT = g.cStrideT;
    if (g.nsplit3 > 0 && n0 >= g.nsplit3) { Cb = g.C3; ncol0 = g.nsplit3; sB = g.c3StrideB; sT = g.c3StrideT; }
    else if (g.nsplit > 0 && n0 >= g.nsplit) { Cb = g.C2; ncol0 = g.nsplit; }
    const int ncols_end = (g.nsplit3 > 0 && n0 < g.nsplit3) ? g.nsplit3 : g.N;   // (columns of this output only)

    const int m = m0 + li < g.M ? m0 + li : g.M - 1;
    const int rb = m % g.B, rt = m / g.B;
    const float* pa0 = g.a0.base + (long)rb * g.a0.strideB + (long)rt * g.a0.strideT;
    const float* pa1 = g.a1.base ? g.a1.base + (long)rb * g.a1.strideB + (long)rt * g.a1.strideT - g.a0.width : pa0;
    if (lane < 32) rowOffC[wave][lane] = (long)rb * sB + (long)rt * sT;
    const int w0 = g.a0.width, klast = g.K - 4;
    // piece (kt, q, b) of this lane: Wf + (((kt*4 + q) * NB + nt*TN + b) * 64 + lane) * 4
    const float* pw = g.Wf + ((long)(nt * TN) * 64 + lane) * 4;
    const long wq = (long)g.NB * 256;                        // floats between consecutive (kt, q) slabs

    f32x16 acc[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    // (the biases are requested FIRST: loaded in the epilogue their latency -- an L2 round trip -- is paid at the end of every tile)
    float biasv[TN];
#pragma unroll
    for (int b = 0; b < TN; ++b) { const int n = n0 + b * 32 + li; biasv[b] = g.bias[n < ncols_end ? n : ncols_end - 1]; }

    f32x4 fa[2][4], fw[TN][4];
    auto request_a = [&](f32x4& dst, int q, int k0) {            // A piece q of the k-tile at k0 (past K: the row's last piece)
        const int k = k0 + lh * 16 + q * 4;
        const int kk = k < klast ? k : klast;
        dst = *reinterpret_cast<const f32x4*>((kk < w0 ? pa0 : pa1) + kk);
    };
    auto request_w = [&](int q, int kt) {
#pragma unroll
        for (int b = 0; b < TN; ++b) fw[b][q] = *reinterpret_cast<const f32x4*>(pw + (long)(kt * 4 + q) * wq + b * 256);
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) request_a(fa[0][q], q, 0);
#pragma unroll
    for (int q = 0; q < 4; ++q) { request_w(q, 0); if (NK > 1) request_a(fa[1][q], q, BK); }
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int b = 0; b < TN; ++b)
#if MP_GEMM_EXP == 4
                    acc[b][s4] += fa[kt & 1][q][s4] * fw[b][q][s4];
#else
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kt & 1][q][s4], fw[b][q][s4], acc[b], 0, 0, 0);
#endif
            __builtin_amdgcn_sched_barrier(0);
#if MP_GEMM_EXP != 2 && MP_GEMM_EXP != 5
            if (kt + 2 < NK) request_a(fa[kt & 1][q], q, (kt + 2) * BK);
#endif
#if MP_GEMM_EXP != 2 && MP_GEMM_EXP != 6
            if (kt + 1 < NK) request_w(q, kt + 1);
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // rowOffC of this wave (written by its own lanes)
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int n = n0 + b * 32 + li;
        if (n >= ncols_end) continue;
        const float bias = biasv[b];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (m0 + ml < g.M) {
                float v = acc[b][r] + bias;
                if (g.relu) v = relu_(v);
#if MP_GEMM_EXP == 1
                if (v == 123456.f)
#endif
                Cb[rowOffC[wave][ml] + (n - ncol0)] = g.pairOut ? __uint_as_float(pair_of(v)) : v;
            }
        }
    }
}

template <int TN, int NK>
MP_KERNEL __launch_bounds__(256, 2) void mp_gemm_f32_frag(GemmArgs g, int nTilesM, int nTilesN) {
    __shared__ long rowOffC[4][32];
    gemm_frag_body<TN, NK>(g, (int)blockIdx.x, nTilesM, nTilesN, rowOffC);
}

// ---------------------------------------------------------------------------------------------------------
// mp_gemm_l2l1<NK2, NT1> (round 4) -- the seam between the joints block and the three blocks that read its output
// (models/net.py:103-117): joints.linear2 (y = out1 W2^T + b2, N2 <= 96 columns) and the stacked linear1 of pose | velocity |
// foot contact over cat(y, imu) (net.py:106,113) in ONE launch.  A wave owns 32 rows: it computes their y tile exactly as
// mp_gemm_f32_frag<3, NK2> does, stores it (the caller wants pred_joints) and keeps it in LDS -- the rows of y are the first
// K segment of the second GEMM's A operand, so they never come back from HBM and there is no kernel boundary, no launch
// ramp and no tail between the two; the second GEMM then runs all NT1 64-column groups of the stacked output for those rows
// from A fragments held in registers (80 VGPRs for K = 132) with W streamed in fragment order, one long MFMA stream per wave.
// Same k pairing and MFMA order per output element as the separate launches: bit-identical results.
constexpr int L2L1_YP = 76;     // LDS row pitch of the y tile in floats (16-byte aligned; 32 rows x b128 reads conflict-free)

// FULL: M is a multiple of 32 -- no row of any wave's tile lies past M, and the per-group epilogues of phase 2 carry no branch:
// behind a branch the compiler merges the wait counts of both paths into vmcnt(0), i.e. the first MFMAs of the next group
// would wait for the 32 stores of this one to be acknowledged (measured: 128 -> ... us per launch).
// Round 6 measured the three changes the SQ counters of round 5 had named -- buffer stores with one lane offset per tile (no 64-bit
// address arithmetic per element: 2.35 M VALU instructions beside 2.21 M MFMAs), the stores of a column group issued inside the next
// group behind its W requests, both -- on one box, three rounds interleaved: 3.617-3.631 ms per step and 0.262-0.269 ms of linear
// layers per forward for every variant, outputs bit-identical (profiles/r06_l2l1_ab.txt).  Neither the VALU count nor the store
// acknowledgements are what holds this kernel at 0.65 of the MFMA time of its padded tiles; the variants were removed.
template <int NK2, int NT1, bool FULL>
MP_KERNEL __launch_bounds__(256, 1) void mp_gemm_l2l1(GemmArgs g2, GemmArgs g1) {
    constexpr int TN2 = 3, NK1 = 5, TN1 = 2;
    __shared__ long rowOff2[4][32], rowOffA[4][32], rowOffF[4][32];
    __shared__ __attribute__((aligned(16))) float ytile[4][32 * L2L1_YP];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = ((int)blockIdx.x * 4 + wave) * 32;
    if (m0 >= g2.M) return;
    const int m = m0 + li < g2.M ? m0 + li : g2.M - 1;
    const int rb = m % g2.B, rt = m / g2.B;
    if (lane < 32) {
        rowOff2[wave][lane] = (long)rb * g2.cStrideB + (long)rt * g2.cStrideT;
        rowOffA[wave][lane] = (long)rb * g1.cStrideB + (long)rt * g1.cStrideT;
        rowOffF[wave][lane] = (long)rb * g1.c3StrideB + (long)rt * g1.c3StrideT;
    }
    float* Y = ytile[wave];

    // ---- phase 1: y = A2 W2^T + b2 (one K segment: the layer-1 output, time-major)
    {
        const float* pa = g2.a0.base + (long)rb * g2.a0.strideB + (long)rt * g2.a0.strideT;
        const float* pw = g2.Wf + (long)lane * 4;
        const long wq = (long)g2.NB * 256;
        f32x16 acc[TN2];
#pragma unroll
        for (int b = 0; b < TN2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
        float bias2[TN2];
#pragma unroll
        for (int b = 0; b < TN2; ++b) { const int n = b * 32 + li; bias2[b] = g2.bias[n < g2.N ? n : g2.N - 1]; }
        f32x4 fa[2][4], fw[TN2][4];
        auto request_a = [&](f32x4& dst, int q, int k0) { dst = *reinterpret_cast<const f32x4*>(pa + k0 + lh * 16 + q * 4); };
        auto request_w = [&](int q, int kt) {
#pragma unroll
            for (int b = 0; b < TN2; ++b) fw[b][q] = *reinterpret_cast<const f32x4*>(pw + (long)(kt * 4 + q) * wq + b * 256);
        };
#pragma unroll
        for (int q = 0; q < 4; ++q) request_a(fa[0][q], q, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) { request_w(q, 0); request_a(fa[1][q], q, BK); }
#pragma unroll
        for (int kt = 0; kt < NK2; ++kt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int b = 0; b < TN2; ++b)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kt & 1][q][s4], fw[b][q][s4], acc[b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (kt + 2 < NK2) request_a(fa[kt & 1][q], q, (kt + 2) * BK);
                if (kt + 1 < NK2) request_w(q, kt + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the row offsets of this wave (written by its own lanes)
#pragma unroll
        for (int b = 0; b < TN2; ++b) {
            const int n = b * 32 + li;
            if (n >= g2.N) continue;
            const float bias = bias2[b];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const float v = acc[b][r] + bias;
                Y[ml * L2L1_YP + n] = v;                          // (rows past M: computed from the clamped row, never stored)
                if (m0 + ml < g2.M) g2.C[rowOff2[wave][ml] + n] = v;
            }
        }
    }
    // the y tile is written and read by this wave only: a wave-level fence orders its LDS writes before its reads
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

    // ---- phase 2: X1 = relu(cat(y, imu) W1^T + b1) for all NT1 64-column groups; A fragments of the 32 rows in registers
    const int w0 = g1.a0.width, klast = g1.K - 4;               // w0 = columns of y (72); k >= K: the row's last piece (W is 0 there)
    const float* pa1 = g1.a1.base + (long)rb * g1.a1.strideB + (long)rt * g1.a1.strideT - w0;
    f32x4 fa[NK1][4];
#pragma unroll
    for (int kt = 0; kt < NK1; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = kt * BK + lh * 16 + q * 4;
            const int kk = k < klast ? k : klast;
            fa[kt][q] = kk < w0 ? *reinterpret_cast<const f32x4*>(Y + li * L2L1_YP + kk) : *reinterpret_cast<const f32x4*>(pa1 + kk);
        }
    // (all biases up front: a bias load inside an epilogue is a load whose first use waits with vmcnt(0) -- for the output
    //  stores in front of it too, one store round trip per group)
    float bias1[NT1][TN1];
#pragma unroll
    for (int ng = 0; ng < NT1; ++ng)
#pragma unroll
        for (int b = 0; b < TN1; ++b) bias1[ng][b] = g1.bias[(ng * TN1 + b) * 32 + li];
    const float* pw = g1.Wf + (long)lane * 4;
    const long wq = (long)g1.NB * 256;
    f32x4 fw[TN1][4];
    auto request_w = [&](int q, int flat) {                      // flat = group * NK1 + kt
        const int ng = flat / NK1, kt = flat - ng * NK1;
#pragma unroll
        for (int b = 0; b < TN1; ++b) fw[b][q] = *reinterpret_cast<const f32x4*>(pw + (long)(kt * 4 + q) * wq + (ng * TN1 + b) * 256);
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) request_w(q, 0);
    // (unrolled completely: with a real loop the compiler merges the wait counts at its header into one vmcnt(0), i.e. every
    //  group would wait for the previous group's 32 output stores to be acknowledged before its first MFMA)
    auto put_group = [&](int ng, const f32x16 (&val)[TN1]) {
        const int n0 = ng * TN1 * 32;
        const bool toF = g1.nsplit3 > 0 && n0 >= g1.nsplit3, toB = !toF && g1.nsplit > 0 && n0 >= g1.nsplit;
        const int ncol0 = toF ? g1.nsplit3 : toB ? g1.nsplit : 0;
        float* Cb = toF ? g1.C3 : toB ? g1.C2 : g1.C;
        const long* roff = toF ? rowOffF[wave] : rowOffA[wave];
#pragma unroll
        for (int b = 0; b < TN1; ++b) {
            const int n = n0 + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (FULL || m0 + ml < g1.M) Cb[roff[ml] + (n - ncol0)] = val[b][r];
            }
        }
    };
#pragma unroll
    for (int ng = 0; ng < NT1; ++ng) {
        f32x16 acc[TN1];
#pragma unroll
        for (int b = 0; b < TN1; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NK1; ++kt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int b = 0; b < TN1; ++b)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kt][q][s4], fw[b][q][s4], acc[b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int nxt = ng * NK1 + kt + 1;
                if (nxt < NT1 * NK1) request_w(q, nxt);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int b = 0; b < TN1; ++b) {
            const float bias = bias1[ng][b];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = relu_(acc[b][r] + bias);
        }
        put_group(ng, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// mp_gemm_f32_wide<NK, NT, FULL> (round 4) -- phase 2 of mp_gemm_l2l1 on its own: a wide linear1 (N = NT * 64 columns, K <= 160)
// whose A rows come from global memory.  A wave owns 32 rows and ALL columns: its A fragments (16 * NK registers) are loaded once,
// W streams through in fragment order, one uninterrupted MFMA stream of NT * NK * 32 instructions per wave, branch-free
// epilogues (FULL: M % 32 == 0).  For full batches (>= one wave per SIMD); smaller ones keep mp_gemm_f32_frag's many small tiles.
template <int NK, int NT, bool FULL>
MP_KERNEL __launch_bounds__(256, 1) void mp_gemm_f32_wide(GemmArgs g) {
    constexpr int TN1 = 2;
    __shared__ long rowOffA[4][32], rowOffF[4][32];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int li = lane & 31, lh = lane >> 5;
    const int m0 = ((int)blockIdx.x * 4 + wave) * 32;
    if (m0 >= g.M) return;
    const int m = m0 + li < g.M ? m0 + li : g.M - 1;
    const int rb = m % g.B, rt = m / g.B;
    if (lane < 32) {
        rowOffA[wave][lane] = (long)rb * g.cStrideB + (long)rt * g.cStrideT;
        rowOffF[wave][lane] = (long)rb * g.c3StrideB + (long)rt * g.c3StrideT;
    }
    const int w0 = g.a0.width, klast = g.K - 4;
    const float* pa0 = g.a0.base + (long)rb * g.a0.strideB + (long)rt * g.a0.strideT;
    const float* pa1 = g.a1.base ? g.a1.base + (long)rb * g.a1.strideB + (long)rt * g.a1.strideT - w0 : pa0;
    float bias1[NT][TN1];
#pragma unroll
    for (int ng = 0; ng < NT; ++ng)
#pragma unroll
        for (int b = 0; b < TN1; ++b) bias1[ng][b] = g.bias[(ng * TN1 + b) * 32 + li];
    f32x4 fa[NK][4];
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int k = kt * BK + lh * 16 + q * 4;
            const int kk = k < klast ? k : klast;
            fa[kt][q] = *reinterpret_cast<const f32x4*>((kk < w0 ? pa0 : pa1) + kk);
        }
    const float* pw = g.Wf + (long)lane * 4;
    const long wq = (long)g.NB * 256;
    f32x4 fw[TN1][4];
    auto request_w = [&](int q, int flat) {                      // flat = group * NK + kt
        const int ng = flat / NK, kt = flat - ng * NK;
#pragma unroll
        for (int b = 0; b < TN1; ++b) fw[b][q] = *reinterpret_cast<const f32x4*>(pw + (long)(kt * 4 + q) * wq + (ng * TN1 + b) * 256);
    };
#pragma unroll
    for (int q = 0; q < 4; ++q) request_w(q, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the row offsets of this wave (written by its own lanes)
#pragma unroll
    for (int ng = 0; ng < NT; ++ng) {
        f32x16 acc[TN1];
#pragma unroll
        for (int b = 0; b < TN1; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                    for (int b = 0; b < TN1; ++b)
                        acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kt][q][s4], fw[b][q][s4], acc[b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                const int nxt = ng * NK + kt + 1;
                if (nxt < NT * NK) request_w(q, nxt);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const int n0 = ng * TN1 * 32;
        float* Cb = g.C;
        int ncol0 = 0;
        const long* roff = rowOffA[wave];
        if (g.nsplit3 > 0 && n0 >= g.nsplit3) { Cb = g.C3; ncol0 = g.nsplit3; roff = rowOffF[wave]; }
        else if (g.nsplit > 0 && n0 >= g.nsplit) { Cb = g.C2; ncol0 = g.nsplit; }
#pragma unroll
        for (int b = 0; b < TN1; ++b) {
            const int n = n0 + b * 32 + li;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ml = (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (FULL || m0 + ml < g.M) {
                    float v = acc[b][r] + bias1[ng][b];
                    if (g.relu) v = relu_(v);
                    Cb[roff[ml] + (n - ncol0)] = v;
                }
            }
        }
    }
}

template <int NK, int NT>
void launch_wide(const GemmArgs& g, hipStream_t s) {
    const int blocks = (g.M + 127) / 128;
    if (g.M % 32 == 0) hipLaunchKernelGGL((mp_gemm_f32_wide<NK, NT, true>), dim3(blocks), dim3(256), 0, s, g);
    else hipLaunchKernelGGL((mp_gemm_f32_wide<NK, NT, false>), dim3(blocks), dim3(256), 0, s, g);
}

// two independent GEMMs in one launch: workgroups [0, grid1) run g1, the others g2 (mp_launch_gemm_pair)
template <int TN1, int NK1, int TN2, int NK2>
MP_KERNEL __launch_bounds__(256, 2) void mp_gemm_f32_frag2(GemmArgs g1, int nTilesM1, int nTilesN1, int grid1,
                                                             GemmArgs g2, int nTilesM2, int nTilesN2) {
    __shared__ long rowOffC[4][32];
    if ((int)blockIdx.x < grid1) gemm_frag_body<TN1, NK1>(g1, (int)blockIdx.x, nTilesM1, nTilesN1, rowOffC);
    else gemm_frag_body<TN2, NK2>(g2, (int)blockIdx.x - grid1, nTilesM2, nTilesN2, rowOffC);
}

MP_KERNEL __launch_bounds__(256) void mp_pack_wfrag(const float* __restrict__ W, float* __restrict__ Wf, int Npad, int Kpad) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;       // destination float
    const long total = (long)Npad * Kpad;
    if (gid >= total) return;
    const int NB = Npad / 32;
    const int j = (int)(gid & 3), lane = (int)((gid >> 2) & 63);
    const long piece = gid >> 8;                                  // (kt*4 + q) * NB + b
    const int b = (int)(piece % NB);
    const int kq = (int)(piece / NB);
    const int kt = kq >> 2, q = kq & 3;
    const int li = lane & 31, lh = lane >> 5;
    Wf[gid] = W[(long)(b * 32 + li) * Kpad + kt * 32 + lh * 16 + q * 4 + j];
}

template <int TN, int NK>
void launch_frag(const GemmArgs& g, hipStream_t s) {
    const int nTilesM = (g.M + 127) / 128;
    const int nTilesN = (g.N + TN * 32 - 1) / (TN * 32);
    const int grid = ((nTilesM + 7) / 8) * 8 * nTilesN;
    hipLaunchKernelGGL((mp_gemm_f32_frag<TN, NK>), dim3(grid), dim3(256), 0, s, g, nTilesM, nTilesN);
}

template <int TN>
bool launch_frag_k(const GemmArgs& g, hipStream_t s) {
    switch (g.Kpad / BK) {
        case 2: launch_frag<TN, 2>(g, s); return true;
        case 4: launch_frag<TN, 4>(g, s); return true;
        case 5: launch_frag<TN, 5>(g, s); return true;
        case 8: launch_frag<TN, 8>(g, s); return true;
        case 16: launch_frag<TN, 16>(g, s); return true;
        default: return false;
    }
}

template <int TN, int NK>
void launch_rows(const GemmArgs& g, hipStream_t s) {
    const int nTilesM = (g.M + 127) / 128;
    const int nTilesN = (g.N + TN * 32 - 1) / (TN * 32);
    const int grid = ((nTilesM + 7) / 8) * 8 * nTilesN;
    hipLaunchKernelGGL((mp_gemm_f32_rows<TN, NK>), dim3(grid), dim3(256), 0, s, g, nTilesM, nTilesN);
}

// the linear2 shapes of the four modules (and foot contact's linear1): K = 512 / 256 / 128 / 132, N = 72 / 96 / 2 / 64
template <int TN>
bool launch_rows_k(const GemmArgs& g, hipStream_t s) {
    switch (g.Kpad / BK) {
        case 4: launch_rows<TN, 4>(g, s); return true;
        case 5: launch_rows<TN, 5>(g, s); return true;
        case 8: launch_rows<TN, 8>(g, s); return true;
        case 16: launch_rows<TN, 16>(g, s); return true;
        default: return false;
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

void mp_launch_pack_wfrag(const float* W, float* Wf, int Npad, int Kpad, hipStream_t s) {
    const long n = (long)Npad * Kpad;
    hipLaunchKernelGGL(mp_pack_wfrag, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, W, Wf, Npad, Kpad);
}

// fragment-ordered W and rows of whole 16-byte pieces: the round-4 kernels; anything else (a caller's odd row width) falls through
// to the row-streaming / LDS-staged kernels of rounds 1-3 at the end of mp_launch_gemm
static bool frag_usable(const GemmArgs& g) { return g.Wf && g.NB > 0 && (g.K & 3) == 0 && (g.a0.width & 3) == 0; }

bool mp_gemm_l2l1_applicable(const GemmArgs& g2, const GemmArgs& g1) {
    // joints.linear2 (K = 512, N = 72) -> stacked linear1 over cat(y, imu) (K = 72 + 60, N = 576 = nine 64-column groups)
    if (!frag_usable(g2) || !frag_usable(g1) || g2.a1.base || !g1.a1.base) return false;
    if (!(g2.NB == 3 && g2.Kpad == 512 && g2.K == 512 && g2.N > 64 && g2.N <= L2L1_YP && (g2.N & 3) == 0 && !g2.relu && !g2.pairOut)) return false;
    return g1.Kpad == 160 && g1.a0.width == g2.N && g1.NB == 18 && g1.N == 576 && g1.relu && !g1.pairOut && g1.M == g2.M && g1.B == g2.B &&
           g1.nsplit > 0 && g1.nsplit % 64 == 0 && g1.nsplit3 > g1.nsplit && g1.nsplit3 % 64 == 0 && g1.C && g1.C2 && g1.C3;
}

bool mp_launch_gemm_l2l1(const GemmArgs& g2, const GemmArgs& g1, hipStream_t s) {
    if (!mp_gemm_l2l1_applicable(g2, g1)) return false;
    const int blocks = (g2.M + 127) / 128;
    if (g2.M % 32 == 0) hipLaunchKernelGGL((mp_gemm_l2l1<16, 9, true>), dim3(blocks), dim3(256), 0, s, g2, g1);
    else hipLaunchKernelGGL((mp_gemm_l2l1<16, 9, false>), dim3(blocks), dim3(256), 0, s, g2, g1);
    return true;
}

bool mp_launch_gemm_pair(const GemmArgs& g1, const GemmArgs& g2, hipStream_t s) {
    // velocity.linear2 (K = 256, N = 72: three 32-column tiles, 8 k-tiles) + foot_contact.linear2 (K = 128, N = 2: one tile, 4 k-tiles)
    if (!frag_usable(g1) || !frag_usable(g2) || g1.nsplit || g2.nsplit || g1.nsplit3 || g2.nsplit3) return false;
    if (!(g1.NB == 3 && g1.Kpad == 256 && g1.N > 64 && g1.N <= 96 && g2.NB == 1 && g2.Kpad == 128 && g2.N <= 32)) return false;
    const int tm1 = (g1.M + 127) / 128, tm2 = (g2.M + 127) / 128;
    const int grid1 = ((tm1 + 7) / 8) * 8, grid2 = ((tm2 + 7) / 8) * 8;
    hipLaunchKernelGGL((mp_gemm_f32_frag2<3, 8, 1, 4>), dim3(grid1 + grid2), dim3(256), 0, s, g1, tm1, 1, grid1, g2, tm2, 1);
    return true;
}

void mp_launch_gemm(const GemmArgs& g, int bn, hipStream_t s) {
    static const int frag_tn = getenv("MP_GEMM_FRAG_TN") ? atoi(getenv("MP_GEMM_FRAG_TN")) : 0;          // micro-benchmark only (tools/micro)
    if (frag_usable(g)) {
        // full batches (at least three quarters of a wave per SIMD), wide outputs without padding columns: one wave = 32 rows x ALL columns
        if (!frag_tn && !g.pairOut && g.M >= 24576 && g.N == g.NB * 32 && g.N % 64 == 0 && (g.nsplit % 64) == 0 && (g.nsplit3 % 64) == 0 &&
            (g.nsplit3 == 0 || g.C3) && (g.nsplit == 0 || g.C2)) {
            const int nk = g.Kpad / BK, nt = g.N / 64;
            if (nk == 2 && nt == 4) { launch_wide<2, 4>(g, s); return; }          // joints.linear1
            if (nk == 5 && nt == 4) { launch_wide<5, 4>(g, s); return; }          // a single H = 256 block's linear1
            if (nk == 5 && nt == 8) { launch_wide<5, 8>(g, s); return; }          // pose | velocity
            if (nk == 5 && nt == 9) { launch_wide<5, 9>(g, s); return; }          // pose | velocity | foot contact
        }
        const int npad32 = g.NB;                                      // 32-column tiles of the padded W
        // columns per wave: all of a narrow output (linear2: N <= 96); 64 of a wide one (linear1: four waves per SIMD keep the
        // MFMA pipe busier than two waves with 128 columns each -- 78.6 vs 85.0 us for the stacked pose + velocity linear1)
        int tn = g.N <= 32 ? 1 : g.N <= 64 ? 2 : g.N <= 96 ? 3 : 2;
        // a handful of rows (one 128-row tile: a one-stream tick has 45): one 32-column tile per wave -- a wave's MFMA stream is
        // all there is (joints / pose linear2 at 45 rows: 768 MFMAs = 20 us for the one wave that owned 96 columns), so the
        // columns go to as many workgroups as there are tiles.  Same k order per output element: bit-identical.
        if (g.M <= 128) tn = 1;
        if (frag_tn) tn = frag_tn;
        if (npad32 % tn == 0 && (g.nsplit == 0 || g.nsplit % (tn * 32) == 0) && (g.nsplit3 == 0 || g.nsplit3 % (tn * 32) == 0)) {
            const bool done = tn == 1 ? launch_frag_k<1>(g, s) : tn == 2 ? launch_frag_k<2>(g, s) : tn == 3 ? launch_frag_k<3>(g, s)
                                                                                                             : launch_frag_k<4>(g, s);
            if (done) return;
        }
    }
    // (row-streaming kernel for the linear2 shapes -- few columns, K >= 128)
    if (g.N <= 96 && g.Kpad >= 128) {
        // (W is padded to a multiple of bn rows: bn = 32 -> 1 tile, 96 -> up to 3)
        const bool done = g.N > 64 ? launch_rows_k<3>(g, s) : g.N > 32 ? launch_rows_k<2>(g, s) : launch_rows_k<1>(g, s);
        if (done) return;
    }
    if (bn == 128) launch<2, 2, 2, 2>(g, s);
    else if (bn == 96) launch<4, 1, 1, 3>(g, s);
    else launch<4, 1, 1, 1>(g, s);
}

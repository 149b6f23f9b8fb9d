// Evaluator metrics on the device: FullMotionEvaluator.__call__ (articulate/evaluator.py:292-343) after its two
// forward_kinematics calls -- the step right behind the hot path in evaluate.py:20-29.
//
// Ten rows of [mean, std(dim=0).mean()] over value matrices x[n][c] (n = frame, c = joint / vertex / 1):
//   0 je   |j_p + off - j_t|            [N][24]     off = j_t[align] - j_p[align]              (evaluator.py:321,323)
//   1 ve   |v_p + off - v_t|            [N][V]                                                  (:322)
//   2 lae  angle(R_local_p, R_local_t)  [N][24]  degrees                                        (:324)
//   3 gae  angle(R_global_p, R_global_t)[N][24]  degrees                                        (:325)
//   4 jkp  |(j_p[n+3] - 3 j_p[n+2] + 3 j_p[n+1] - j_p[n]) f^3|   [N-3][24]                      (:326)
//   5 jkt  the same on j_t                                                                       (:327)
//   6 te   |(j_p[n+f][0] - j_p[n][0]) - (j_t[n+f][0] - j_t[n][0])| * 100   [N-f][1]             (:328)
//   7-9    rows 0, 2, 3 restricted to the joints of joint_mask                                   (:329-331)
// angle(Ra, Rb) = |rotation vector of Ra^T Rb| (angular.py:86-99,154-164: one cv2.Rodrigues per matrix upstream) =
// 2 asin(|Ra^T Rb - I|_F / (2 sqrt 2)), evaluated in fp64.
//
// HBM-bound (the vertex matrices dominate: 2 x N x V x 12 bytes read once).  Two kernels: mp_eval_partial -- one thread per
// column c, the frames split into R row chunks (grid.y), every thread runs down its chunk accumulating sum and sum of
// squares in fp64 (consecutive threads read consecutive columns of a frame: coalesced) -- and mp_eval_finish, one workgroup
// that turns the partials into per-column mean / unbiased variance and reduces over the columns.
#include "mp_common.h"

namespace {

constexpr int NJ = 24;
constexpr int ROW_CHUNKS = 32;

struct EvalArgs {
    const float *pose_p, *pose_t, *rg_p, *rg_t, *j_p, *j_t, *v_p, *v_t;   // v_*: nullptr = no mesh
    long N;
    int V, fps, align;
    unsigned mask;
    double* part;      // [ROW_CHUNKS][C_total][2]
    float* table;      // [10][2]
};

// flat column space: [je 24][lae 24][gae 24][jkp 24][jkt 24][te 1][ve V]
__host__ __device__ inline int col_total(int V) { return 5 * NJ + 1 + V; }

__device__ __forceinline__ double angle_deg(const float* __restrict__ a, const float* __restrict__ b) {
    // D = a^T b;  |D - I|_F^2 = sum (D_ij - delta_ij)^2
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double d = (double)a[0 * 3 + i] * b[0 * 3 + j] + (double)a[1 * 3 + i] * b[1 * 3 + j] +
                             (double)a[2 * 3 + i] * b[2 * 3 + j] - (i == j ? 1.0 : 0.0);
            s += d * d;
        }
    double x = sqrt(s) * 0.35355339059327379;        // / (2 sqrt 2)
    x = x > 1.0 ? 1.0 : x;
    return 2.0 * asin(x) * 57.295779513082323;
}

__device__ __forceinline__ float norm3(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }

MP_KERNEL __launch_bounds__(256) void mp_eval_partial(EvalArgs a) {
    const int C = col_total(a.V);
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const int chunk = blockIdx.y;
    int metric, col;
    if (c < 5 * NJ) { metric = c / NJ; col = c % NJ; }          // 0 je, 1 lae, 2 gae, 3 jkp, 4 jkt
    else if (c == 5 * NJ) { metric = 5; col = 0; }              // te
    else { metric = 6; col = c - 5 * NJ - 1; }                  // ve
    const long rows = metric == 3 || metric == 4 ? a.N - 3 : (metric == 5 ? a.N - a.fps : a.N);
    double s = 0.0, ss = 0.0;
    if (rows > 0) {
        const long per = (rows + ROW_CHUNKS - 1) / ROW_CHUNKS;
        const long n0 = chunk * per, n1 = n0 + per < rows ? n0 + per : rows;
        const float f3 = (float)a.fps * (float)a.fps * (float)a.fps;
        for (long n = n0; n < n1; ++n) {
            float x;
            if (metric == 0 || metric == 6) {
                const float* jp = a.j_p + (n * NJ + a.align) * 3;
                const float* jt = a.j_t + (n * NJ + a.align) * 3;
                const float ox = jt[0] - jp[0], oy = jt[1] - jp[1], oz = jt[2] - jp[2];
                const float* p = metric == 0 ? a.j_p + (n * NJ + col) * 3 : a.v_p + (n * a.V + col) * 3;
                const float* t = metric == 0 ? a.j_t + (n * NJ + col) * 3 : a.v_t + (n * a.V + col) * 3;
                x = norm3(p[0] + ox - t[0], p[1] + oy - t[1], p[2] + oz - t[2]);
            } else if (metric == 1) {
                x = (float)angle_deg(a.pose_p + (n * NJ + col) * 9, a.pose_t + (n * NJ + col) * 9);
            } else if (metric == 2) {
                x = (float)angle_deg(a.rg_p + (n * NJ + col) * 9, a.rg_t + (n * NJ + col) * 9);
            } else if (metric == 3 || metric == 4) {
                const float* j = (metric == 3 ? a.j_p : a.j_t) + (n * NJ + col) * 3;
                float d[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    d[k] = (((j[3 * NJ * 3 + k] - 3.f * j[2 * NJ * 3 + k]) + 3.f * j[1 * NJ * 3 + k]) - j[k]) * f3;
                x = norm3(d[0], d[1], d[2]);
            } else {
                const long m = n + a.fps;
                float d[3];
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    d[k] = (a.j_p[m * NJ * 3 + k] - a.j_p[n * NJ * 3 + k]) - (a.j_t[m * NJ * 3 + k] - a.j_t[n * NJ * 3 + k]);
                x = norm3(d[0], d[1], d[2]) * 100.f;
            }
            s += (double)x;
            ss += (double)x * (double)x;
        }
    }
    double* o = a.part + ((size_t)chunk * C + c) * 2;
    o[0] = s; o[1] = ss;
}

__device__ double block_sum(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int st = blockDim.x / 2; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) sh[threadIdx.x] += sh[threadIdx.x + st];
        __syncthreads();
    }
    const double r = sh[0];
    __syncthreads();
    return r;
}

// table row = { sum over all entries / (rows * cols),  mean over columns of the unbiased std over rows }
MP_KERNEL __launch_bounds__(256) void mp_eval_finish(EvalArgs a) {
    __shared__ double sh[256];
    const int C = col_total(a.V);
    const float nanv = __builtin_nanf("");
    // (metric id in the flat space, first column, columns, rows, use joint mask?, table row)
    const int first[10] = {0, 5 * NJ + 1, NJ, 2 * NJ, 3 * NJ, 4 * NJ, 5 * NJ, 0, NJ, 2 * NJ};
    const int ncol[10] = {NJ, a.V, NJ, NJ, NJ, NJ, 1, NJ, NJ, NJ};
    const long nrow[10] = {a.N, a.N, a.N, a.N, a.N - 3, a.N - 3, a.N - a.fps, a.N, a.N, a.N};
    for (int row = 0; row < 10; ++row) {
        const bool masked = row >= 7;
        double s = 0.0, sd = 0.0, cnt = 0.0;
        const long n = nrow[row];
        const bool have = n > 0 && !(row == 1 && a.v_p == nullptr) && !(masked && a.mask == 0u);
        if (have) {
            for (int k = threadIdx.x; k < ncol[row]; k += blockDim.x) {
                if (masked && !((a.mask >> k) & 1u)) continue;
                double cs = 0.0, css = 0.0;
                for (int r = 0; r < ROW_CHUNKS; ++r) {
                    const double* p = a.part + ((size_t)r * C + first[row] + k) * 2;
                    cs += p[0]; css += p[1];
                }
                s += cs;
                if (n > 1) {
                    double var = (css - cs * cs / (double)n) / (double)(n - 1);
                    sd += sqrt(var > 0.0 ? var : 0.0);
                }
                cnt += 1.0;
            }
        }
        s = block_sum(s, sh); sd = block_sum(sd, sh); cnt = block_sum(cnt, sh);
        if (threadIdx.x == 0) {
            float mean, stdm;
            if (masked && a.mask == 0u) { mean = 0.f; stdm = nanv; }        // torch.zeros(1): mean 0, std of one element NaN
            else if (!have || cnt == 0.0) { mean = nanv; stdm = nanv; }     // empty matrix (sequence shorter than the window; no mesh)
            else { mean = (float)(s / (cnt * (double)n)); stdm = n > 1 ? (float)(sd / cnt) : nanv; }
            a.table[row * 2 + 0] = mean;
            a.table[row * 2 + 1] = stdm;
        }
    }
}

// pose with the joints of `ignored` replaced by the identity (evaluate.py:25-26)
MP_KERNEL __launch_bounds__(256) void mp_mask_pose(const float* __restrict__ in, float* __restrict__ out, long N, unsigned ignored) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;        // (frame, joint, element)
    if (gid >= N * NJ * 9) return;
    const int e = (int)(gid % 9), j = (int)((gid / 9) % NJ);
    out[gid] = ((ignored >> j) & 1u) ? ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f) : in[gid];
}

}  // namespace

size_t mp_eval_partial_doubles(int V) { return (size_t)ROW_CHUNKS * col_total(V) * 2; }

void mp_launch_mask_pose(const float* in, float* out, long N, unsigned ignored, hipStream_t s) {
    if (N <= 0) return;
    const long n = N * NJ * 9;
    hipLaunchKernelGGL(mp_mask_pose, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, in, out, N, ignored);
}

void mp_launch_eval_metrics(const float* pose_p, const float* pose_t, const float* rg_p, const float* rg_t, const float* j_p,
                            const float* j_t, const float* v_p, const float* v_t, long N, int V, int fps, int align,
                            unsigned mask, double* part, float* table, hipStream_t s) {
    EvalArgs a{pose_p, pose_t, rg_p, rg_t, j_p, j_t, v_p, v_t, N, v_p ? V : 0, fps, align, mask, part, table};
    const int C = col_total(a.V);
    hipLaunchKernelGGL(mp_eval_partial, dim3((C + 255) / 256, ROW_CHUNKS), dim3(256), 0, s, a);
    hipLaunchKernelGGL(mp_eval_finish, dim3(1), dim3(256), 0, s, a);
}

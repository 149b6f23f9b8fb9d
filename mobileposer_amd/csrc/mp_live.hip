// Live front-end, per tick and for S streams at once: raw sensor samples -> the 60-d network input frame
//   (mobileposer/live_demo.py:213-236: unit quaternion -> rotation, sensor frame -> SMPL frame with the T-pose calibration
//    of :161-174, channel re-order [1,4,3,0,2] (:219-220, combiner.py:14-17), acceleration / acc_scale, combo mask).
// One thread per (stream, output slot): 5 x 12 outputs per stream, everything a thread needs is < 200 bytes -- the kernel
// exists so that S = 512 streams are ONE launch in front of mp_stream_step instead of 512 x a dozen host tensor ops.
#include "mp_common.h"

namespace {

__constant__ int c_order[5] = {1, 4, 3, 0, 2};

__device__ __forceinline__ void mat3_mul(const float* A, const float* B, float* C) {
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3 + 0] * B[0 * 3 + c] + A[r * 3 + 1] * B[1 * 3 + c] + A[r * 3 + 2] * B[2 * 3 + c];
}

// quat [S,5,4] wxyz (any norm), acc [S,5,3] m/s^2; calibration per stream: smpl2imu [S,3,3], device2bone [S,5,3,3],
// acc_off [S,5,3]; keep: bit k set = output slot k belongs to the device combo; frames [S,60] = [5 x 3 acc | 5 x 3x3 ori]
MP_KERNEL __launch_bounds__(256) void mp_live_frames(const float* __restrict__ quat, const float* __restrict__ acc,
                                                       const float* __restrict__ smpl2imu, const float* __restrict__ device2bone,
                                                       const float* __restrict__ acc_off, unsigned keep, float acc_scale, int S,
                                                       float* __restrict__ frames) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= S * 5) return;
    const int s = gid / 5, k = gid - s * 5;
    float* fa = frames + (size_t)s * 60 + k * 3;
    float* fo = frames + (size_t)s * 60 + 15 + k * 9;
    if (!((keep >> k) & 1u)) {
#pragma unroll
        for (int i = 0; i < 3; ++i) fa[i] = 0.f;
#pragma unroll
        for (int i = 0; i < 9; ++i) fo[i] = 0.f;
        return;
    }
    const int src = c_order[k];
    const float* qp = quat + ((size_t)s * 5 + src) * 4;
    const float n = sqrtf(qp[0] * qp[0] + qp[1] * qp[1] + qp[2] * qp[2] + qp[3] * qp[3]);
    const float w = qp[0] / n, x = qp[1] / n, y = qp[2] / n, z = qp[3] / n;
    const float R[9] = {1.f - 2.f * (y * y + z * z), 2.f * (x * y - w * z), 2.f * (x * z + w * y),
                        2.f * (x * y + w * z), 1.f - 2.f * (x * x + z * z), 2.f * (y * z - w * x),
                        2.f * (x * z - w * y), 2.f * (y * z + w * x), 1.f - 2.f * (x * x + y * y)};
    const float* M = smpl2imu + (size_t)s * 9;
    float MR[9], G[9];
    mat3_mul(M, R, MR);
    mat3_mul(MR, device2bone + ((size_t)s * 5 + src) * 9, G);
#pragma unroll
    for (int i = 0; i < 9; ++i) fo[i] = G[i];
    const float* ap = acc + ((size_t)s * 5 + src) * 3;
    const float* op = acc_off + ((size_t)s * 5 + src) * 3;
#pragma unroll
    for (int r = 0; r < 3; ++r) fa[r] = ((M[r * 3 + 0] * ap[0] + M[r * 3 + 1] * ap[1] + M[r * 3 + 2] * ap[2]) - op[r]) / acc_scale;
}

}  // namespace

void mp_launch_live_frames(const float* quat, const float* acc, const float* smpl2imu, const float* device2bone,
                           const float* acc_off, unsigned keep, float acc_scale, int S, float* frames, hipStream_t s) {
    hipLaunchKernelGGL(mp_live_frames, dim3((S * 5 + 255) / 256), dim3(256), 0, s, quat, acc, smpl2imu, device2bone, acc_off,
                       keep, acc_scale, S, frames);
}

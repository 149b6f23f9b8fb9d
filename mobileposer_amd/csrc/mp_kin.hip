// K4 -- fused 6D -> SO(3) (Gram-Schmidt), 16 -> 24 joint scatter and global -> local rotations:
//   MobilePoserNet._reduced_global_to_full (models/net.py:93-99)
//     = r6d_to_rotation_matrix (articulate/math/angular.py:167-182, normalize_tensor general.py:27-39)
//     + reduced_pose_to_full   (utils/model_utils.py:18-25)
//     + inverse_kinematics_R   (articulate/math/spatial.py:197-221, _inverse_tree :115-123)
//     + identity on joint_set.ignored and root = global root (net.py:97-98).
// K5 -- SMPL forward kinematics without mesh: ParametricModel.forward_kinematics
//   (articulate/model.py:208-232; spatial.py:60-75 transformation_matrix, :104-112 _forward_tree).
//
// Both are HBM-bound elementwise-class kernels (K4: 384 B in, 864 B out per frame; K5: 864(+12) B in,
// 1152 B out).  K4 runs one thread per (frame, joint): the parent's global rotation is recomputed from
// its 6 numbers instead of being exchanged, so there is no dependency chain at all (IK on rotations only
// needs the parent's INPUT) and a wave's 64 x 36-byte outputs are one contiguous 2304-byte run.
// K5 has a real chain (tree depth 8): lanes are joints, two frames per wave (32-lane halves), the tree is
// walked level by level and a lane fetches its parent's global transform with wavefront shuffles
// (ds_bpermute) -- the kinematic reduction never touches LDS or HBM.
#include "mp_common.h"
#include <cstdlib>
#include <cstring>

// MP_VARIANT kin_scalar=1: the scalar-access kernels even for aligned buffers (A/B runs and the cross-check test)
static bool mp_kin_scalar_forced() {
    static const bool f = getenv("MP_VARIANT") && strstr(getenv("MP_VARIANT"), "kin_scalar=1");
    return f;
}

namespace {

// joint -> slot in the 16-joint reduced set (config.py:134), -1 = not predicted (identity)
__constant__ int c_slot[24] = {0, 1, 2, 3, 4, 5, 6, -1, -1, 7, -1, -1, 8, 9, 10, 11, 12, 13, 14, 15, -1, -1, -1, -1};
// joint_set.ignored (config.py:135) as a bit mask
constexpr unsigned IGNORED_MASK = (1u << 0) | (1u << 7) | (1u << 8) | (1u << 10) | (1u << 11) | (1u << 20) |
                                  (1u << 21) | (1u << 22) | (1u << 23);

__device__ __forceinline__ float nan0(float x) { return x != x ? 0.f : x; }

// R (row-major 3x3) from 6 numbers = first two COLUMNS of R (angular.py:180); NaN -> 0 (angular.py:181)
// Every product and sum is rounded separately (no FMA contraction) and summed left to right, as the
// reference's elementwise torch ops do: for (near-)colinear inputs the second column is normalised rounding
// noise, and only the same operation sequence reproduces the reference's NaN/0 pattern there.
__device__ __forceinline__ void gram_schmidt(const float* __restrict__ p, float R[9]) {
#pragma clang fp contract(off)
    const float ax = p[0], ay = p[1], az = p[2], bx = p[3], by = p[4], bz = p[5];
    const float na = sqrtf((ax * ax + ay * ay) + az * az);
    const float c0x = ax / na, c0y = ay / na, c0z = az / na;
    const float d = (c0x * bx + c0y * by) + c0z * bz;
    const float ux = bx - d * c0x, uy = by - d * c0y, uz = bz - d * c0z;
    const float nu = sqrtf((ux * ux + uy * uy) + uz * uz);
    const float c1x = ux / nu, c1y = uy / nu, c1z = uz / nu;
    const float c2x = c0y * c1z - c0z * c1y, c2y = c0z * c1x - c0x * c1z, c2z = c0x * c1y - c0y * c1x;
    R[0] = nan0(c0x); R[1] = nan0(c1x); R[2] = nan0(c2x);
    R[3] = nan0(c0y); R[4] = nan0(c1y); R[5] = nan0(c2y);
    R[6] = nan0(c0z); R[7] = nan0(c1z); R[8] = nan0(c2z);
}

__device__ __forceinline__ void global_rot(const float* __restrict__ row96, int joint, float R[9]) {
    const int s = joint >= 0 ? c_slot[joint] : -1;
    if (s >= 0) {
        gram_schmidt(row96 + 6 * s, R);
    } else {
        R[0] = 1.f; R[1] = 0.f; R[2] = 0.f; R[3] = 0.f; R[4] = 1.f; R[5] = 0.f; R[6] = 0.f; R[7] = 0.f; R[8] = 1.f;
    }
}

// frame n reads its 96 numbers at r6d + n*rowStride + rowOffset (lets the streaming step convert only
// window index 40 of every stream); writes pose[n][24][9]
MP_KERNEL __launch_bounds__(256) void mp_r6d_ik(const float* __restrict__ r6d, long N, long rowStride,
                                                  long rowOffset, float* __restrict__ pose,
                                                  const int* __restrict__ parent) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N * 24) return;
    const long n = gid / 24;
    const int i = (int)(gid - n * 24);
    const float* row = r6d + n * rowStride + rowOffset;
    float G[9], out[9];
    global_rot(row, i, G);
    if (i == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) out[k] = G[k];
    } else if ((IGNORED_MASK >> i) & 1u) {
        out[0] = 1.f; out[1] = 0.f; out[2] = 0.f; out[3] = 0.f; out[4] = 1.f; out[5] = 0.f; out[6] = 0.f; out[7] = 0.f; out[8] = 1.f;
    } else {
        float P[9];
        global_rot(row, parent[i], P);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                out[r * 3 + c] = P[0 * 3 + r] * G[0 * 3 + c] + P[1 * 3 + r] * G[1 * 3 + c] + P[2 * 3 + r] * G[2 * 3 + c];
    }
    float* o = pose + gid * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = out[k];
}

// r6d_to_rotation_matrix (articulate/math/angular.py:167-182) on n six-vectors: the ground-truth side of evaluate.py:60
MP_KERNEL __launch_bounds__(256) void mp_r6d_to_rot(const float* __restrict__ r6d, long n, float* __restrict__ out) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    float R[9];
    gram_schmidt(r6d + gid * 6, R);
    float* o = out + gid * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = R[k];
}

// The tree walk and the outputs of mp_fk, shared with the fused mp_r6d_ik_fk: lane = (frame parity) * 32 + joint, G = p's frame's
// local rotation of joint i (also the start value of its global one), p = bv = its bone vector, par = its parent
__device__ __forceinline__ void fk_walk_and_store(bool live, int lane, int i, long n, float G[9], float p[3], int par,
                                                  const float* __restrict__ tran, float* __restrict__ rglobal,
                                                  float* __restrict__ joint) {
    // Tree walk by pointer jumping (round 4): every lane holds the transform (G | p) of its joint relative to its `anc`-th
    // ancestor's frame and doubles that distance per round -- 1, 2, 4, 8 levels: four rounds of 13 wavefront shuffles cover
    // the SMPL tree (depth 8; setup_smpl rejects deeper trees) where the level-by-level walk needed eight rounds of 12.  The
    // shuffles (ds_bpermute) are what this kernel is bound by: 21.6 -> 13 us for 32 000 frames.  Products are associated as
    // (T_a T_b)(T_c T_d) instead of ((T_a T_b) T_c) T_d: results differ from the sequential walk in the last bits only.
    int anc = live && i > 0 ? par : -1;               // nearest ancestor not yet folded in (-1: relative to the world)
#pragma unroll 1
    for (int round = 0; round < 4; ++round) {
        const int srcLane = (lane & 32) + (anc >= 0 ? anc : 0);
        float Pg[9], pp[3];
#pragma unroll
        for (int k = 0; k < 9; ++k) Pg[k] = __shfl(G[k], srcLane, 64);
#pragma unroll
        for (int k = 0; k < 3; ++k) pp[k] = __shfl(p[k], srcLane, 64);
        const int up = __shfl(anc, srcLane, 64);
        if (anc >= 0) {
            float Gn[9], pn[3];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    Gn[r * 3 + c] = Pg[r * 3 + 0] * G[0 * 3 + c] + Pg[r * 3 + 1] * G[1 * 3 + c] + Pg[r * 3 + 2] * G[2 * 3 + c];
                pn[r] = Pg[r * 3 + 0] * p[0] + Pg[r * 3 + 1] * p[1] + Pg[r * 3 + 2] * p[2] + pp[r];
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) G[k] = Gn[k];
            p[0] = pn[0]; p[1] = pn[1]; p[2] = pn[2];
            anc = up;
        }
    }
    if (live) {
        float* og = rglobal + (n * 24 + i) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) og[k] = G[k];
        float tx = 0.f, ty = 0.f, tz = 0.f;
        if (tran) { tx = tran[n * 3 + 0]; ty = tran[n * 3 + 1]; tz = tran[n * 3 + 2]; }
        float* oj = joint + (n * 24 + i) * 3;
        oj[0] = p[0] + tx; oj[1] = p[1] + ty; oj[2] = p[2] + tz;
    }
}

// lanes = joints, 2 frames per wave; bone[24][3] = j_i - j_parent(i) (bone[0] = j_0 = 0), depth[24]
// (boneStride: 0 = one body for all frames; 72 = frame n uses bone + n*72, forward_kinematics with per-frame shapes)
MP_KERNEL __launch_bounds__(256) void mp_fk(const float* __restrict__ pose, const float* __restrict__ tran, long N,
                                              const float* __restrict__ bone, long boneStride,
                                              const int* __restrict__ parent,
                                              const int* __restrict__ depth, float* __restrict__ rglobal,
                                              float* __restrict__ joint) {
    const int lane = threadIdx.x & 63;
    const int i = lane & 31;                       // joint
    const long n = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    const bool live = (i < 24) && (n < N);
    float G[9], p[3], L[9], bv[3];
    int par = 0, dep = -1;
    if (live) {
        const float* src = pose + (n * 24 + i) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) { L[k] = src[k]; G[k] = L[k]; }
        const float* bn = bone + n * boneStride;
        bv[0] = bn[i * 3 + 0]; bv[1] = bn[i * 3 + 1]; bv[2] = bn[i * 3 + 2];
        p[0] = bv[0]; p[1] = bv[1]; p[2] = bv[2];
        par = i > 0 ? parent[i] : 0;
        dep = depth[i];
        (void)dep;
    } else {
#pragma unroll
        for (int k = 0; k < 9; ++k) { L[k] = 0.f; G[k] = 0.f; }
        bv[0] = bv[1] = bv[2] = 0.f; p[0] = p[1] = p[2] = 0.f;
    }
    fk_walk_and_store(live, lane, i, n, G, p, par, tran, rglobal, joint);
}

// ---- coalesced forms (round 4).  The two kernels above move 36-byte records with nine scalar loads / stores per lane:
// every 128-byte line passes the L1 tags nine times and a wave's store instruction touches 18 lines -- measured 1.7-2.3 TB/s.
// Here a workgroup stages 8 frames through LDS: global traffic is 16-byte pieces of contiguous runs (6912-byte pose blocks,
// 384-byte r6d rows), lanes pick their record out of LDS (word stride 9: conflict-free) and results leave the same way.
// Same arithmetic in the same order: bit-identical outputs.  Used when the buffers are 16-byte aligned.
constexpr int kFkFrames = 8;          // frames per workgroup (4 waves x 2)

// 8 frames x 24 joints = 192 threads; frame n reads its 96 numbers at r6d + n*rowStride + rowOffset
MP_KERNEL __launch_bounds__(192) void mp_r6d_ik_lds(const float* __restrict__ r6d, long N, long rowStride, long rowOffset,
                                                      float* __restrict__ pose, const int* __restrict__ parent) {
    __shared__ __attribute__((aligned(16))) float sR[kFkFrames * 96];
    __shared__ float sG[kFkFrames * 16 * 9];           // global rotations of the 16 predicted joints, computed ONCE per frame
    __shared__ __attribute__((aligned(16))) float sO[kFkFrames * 216];
    const long n0 = (long)blockIdx.x * kFkFrames;
    const int nf = (int)(N - n0 < kFkFrames ? N - n0 : kFkFrames);
    const int tid = threadIdx.x;
    const int f = tid / 24, i = tid - f * 24;          // frame within the block; 16-byte piece of its row, then joint
    if (f < nf)
        reinterpret_cast<f32x4*>(sR)[tid] = *reinterpret_cast<const f32x4*>(r6d + (n0 + f) * rowStride + rowOffset + i * 4);
    __syncthreads();
    if (f < nf && i < 16) gram_schmidt(sR + f * 96 + 6 * i, sG + (f * 16 + i) * 9);
    __syncthreads();
    if (f < nf) {
        auto rot = [&](int joint, float R[9]) {
            const int sl = c_slot[joint];
            if (sl >= 0) {
                const float* g = sG + (f * 16 + sl) * 9;
#pragma unroll
                for (int k = 0; k < 9; ++k) R[k] = g[k];
            } else {
                R[0] = 1.f; R[1] = 0.f; R[2] = 0.f; R[3] = 0.f; R[4] = 1.f; R[5] = 0.f; R[6] = 0.f; R[7] = 0.f; R[8] = 1.f;
            }
        };
        float G[9], out[9];
        rot(i, G);
        if (i == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) out[k] = G[k];
        } else if ((IGNORED_MASK >> i) & 1u) {
            out[0] = 1.f; out[1] = 0.f; out[2] = 0.f; out[3] = 0.f; out[4] = 1.f; out[5] = 0.f; out[6] = 0.f; out[7] = 0.f; out[8] = 1.f;
        } else {
            float P[9];
            rot(parent[i], P);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    out[r * 3 + c] = P[0 * 3 + r] * G[0 * 3 + c] + P[1 * 3 + r] * G[1 * 3 + c] + P[2 * 3 + r] * G[2 * 3 + c];
        }
        float* o = sO + tid * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = out[k];
    }
    __syncthreads();
    const f32x4* src = reinterpret_cast<const f32x4*>(sO);
    f32x4* dst = reinterpret_cast<f32x4*>(pose + n0 * 216);
    for (int e = tid; e < nf * 54; e += 192) dst[e] = src[e];
}

// ParametricModel.inverse_kinematics_R (articulate/model.py:146-164 -> spatial.py:197-221, _inverse_tree :115-123) on its own:
// R_local[i] = R_global[parent[i]]^T R_global[i], R_local[0] = R_global[0] -- no chain (only INPUTS of the parent are read), so a
// thread per (frame, joint).  HBM-bound: 864 B in + 864 B out per frame; 8 frames per workgroup staged through LDS in 16-byte
// pieces of one contiguous 6912-byte run, records picked out of LDS at word stride 9 (conflict-free), as mp_r6d_ik_lds does.
MP_KERNEL __launch_bounds__(192) void mp_global_to_local_lds(const float* __restrict__ rglobal, long N, float* __restrict__ rlocal,
                                                               const int* __restrict__ parent) {
    __shared__ __attribute__((aligned(16))) float sG[kFkFrames * 216];
    __shared__ __attribute__((aligned(16))) float sO[kFkFrames * 216];
    const long n0 = (long)blockIdx.x * kFkFrames;
    const int nf = (int)(N - n0 < kFkFrames ? N - n0 : kFkFrames);
    const int tid = threadIdx.x;
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(rglobal + n0 * 216);
        f32x4* dst = reinterpret_cast<f32x4*>(sG);
        for (int e = tid; e < nf * 54; e += 192) dst[e] = src[e];
    }
    __syncthreads();
    const int f = tid / 24, i = tid - f * 24;
    if (f < nf) {
        const float* G = sG + tid * 9;
        float* o = sO + tid * 9;
        if (i == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) o[k] = G[k];
        } else {
            const float* P = sG + (f * 24 + parent[i]) * 9;
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    o[r * 3 + c] = P[0 * 3 + r] * G[0 * 3 + c] + P[1 * 3 + r] * G[1 * 3 + c] + P[2 * 3 + r] * G[2 * 3 + c];
        }
    }
    __syncthreads();
    const f32x4* src = reinterpret_cast<const f32x4*>(sO);
    f32x4* dst = reinterpret_cast<f32x4*>(rlocal + n0 * 216);
    for (int e = tid; e < nf * 54; e += 192) dst[e] = src[e];
}

// ... and for buffers that are not 16-byte aligned: scalar accesses, a thread per (frame, joint)
MP_KERNEL __launch_bounds__(256) void mp_global_to_local(const float* __restrict__ rglobal, long N, float* __restrict__ rlocal,
                                                           const int* __restrict__ parent) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= N * 24) return;
    const long n = gid / 24;
    const int i = (int)(gid - n * 24);
    const float* G = rglobal + gid * 9;
    float out[9];
    if (i == 0) {
#pragma unroll
        for (int k = 0; k < 9; ++k) out[k] = G[k];
    } else {
        const float* P = rglobal + (n * 24 + parent[i]) * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                out[r * 3 + c] = P[0 * 3 + r] * G[0 * 3 + c] + P[1 * 3 + r] * G[1 * 3 + c] + P[2 * 3 + r] * G[2 * 3 + c];
    }
    float* o = rlocal + gid * 9;
#pragma unroll
    for (int k = 0; k < 9; ++k) o[k] = out[k];
}

// mp_r6d_ik_lds and mp_fk in ONE kernel (round 5: the tail of a forward that also wants the FK outputs -- bench.py's call): the local
// rotations go out to `pose` through LDS as above AND stay in registers as the start values of the tree walk; thread (frame f =
// tid / 32, joint i = tid % 32 < 24) is lane (f & 1) * 32 + i of wave f / 2 -- mp_fk's mapping.  Same arithmetic in the same order
// as the two kernels: bit-identical outputs; one launch and one read of the pose less (13 + 16 -> 20 us at 32 000 frames).
MP_KERNEL __launch_bounds__(256) void mp_r6d_ik_fk(const float* __restrict__ r6d, long N, long rowStride, long rowOffset,
                                                     float* __restrict__ pose, const float* __restrict__ bone,
                                                     const int* __restrict__ parent, float* __restrict__ rglobal,
                                                     float* __restrict__ joint) {
    __shared__ __attribute__((aligned(16))) float sR[kFkFrames * 96];
    __shared__ float sG[kFkFrames * 16 * 9];
    __shared__ __attribute__((aligned(16))) float sO[kFkFrames * 216];
    const long n0 = (long)blockIdx.x * kFkFrames;
    const int nf = (int)(N - n0 < kFkFrames ? N - n0 : kFkFrames);
    const int tid = threadIdx.x;
    if (tid < kFkFrames * 24 && tid / 24 < nf)
        reinterpret_cast<f32x4*>(sR)[tid] = *reinterpret_cast<const f32x4*>(r6d + (n0 + tid / 24) * rowStride + rowOffset + (tid % 24) * 4);
    __syncthreads();
    if (tid < kFkFrames * 16 && (tid >> 4) < nf) gram_schmidt(sR + (tid >> 4) * 96 + 6 * (tid & 15), sG + tid * 9);
    __syncthreads();
    const int f = tid >> 5, i = tid & 31, lane = tid & 63;
    const bool live = i < 24 && f < nf;
    float G[9], p[3];
    int par = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) G[k] = 0.f;
    p[0] = p[1] = p[2] = 0.f;
    if (live) {
        auto rot = [&](int jn, float R[9]) {
            const int sl = c_slot[jn];
            if (sl >= 0) {
                const float* g = sG + (f * 16 + sl) * 9;
#pragma unroll
                for (int k = 0; k < 9; ++k) R[k] = g[k];
            } else {
                R[0] = 1.f; R[1] = 0.f; R[2] = 0.f; R[3] = 0.f; R[4] = 1.f; R[5] = 0.f; R[6] = 0.f; R[7] = 0.f; R[8] = 1.f;
            }
        };
        float Gg[9];
        rot(i, Gg);
        par = i > 0 ? parent[i] : 0;
        if (i == 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) G[k] = Gg[k];
        } else if ((IGNORED_MASK >> i) & 1u) {
            G[0] = 1.f; G[1] = 0.f; G[2] = 0.f; G[3] = 0.f; G[4] = 1.f; G[5] = 0.f; G[6] = 0.f; G[7] = 0.f; G[8] = 1.f;
        } else {
            float P[9];
            rot(par, P);
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 3; ++c)
                    G[r * 3 + c] = P[0 * 3 + r] * Gg[0 * 3 + c] + P[1 * 3 + r] * Gg[1 * 3 + c] + P[2 * 3 + r] * Gg[2 * 3 + c];
        }
        float* o = sO + (f * 24 + i) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) o[k] = G[k];
        p[0] = bone[i * 3 + 0]; p[1] = bone[i * 3 + 1]; p[2] = bone[i * 3 + 2];
    }
    __syncthreads();
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(sO);
        f32x4* dst = reinterpret_cast<f32x4*>(pose + n0 * 216);
        for (int e = tid; e < nf * 54; e += 256) dst[e] = src[e];
    }
    fk_walk_and_store(live, lane, i, n0 + f, G, p, par, nullptr, rglobal, joint);
}

// Linear blend skinning of the SMPL mesh on top of mp_fk's outputs (articulate/model.py:234-240, no pose
// blendshape):  T_j = [R_j | p_j - R_j j_j],  vert_v = sum_j w_vj T_j [v_rest; 1]  (+ tran).
// One workgroup per (frame, 256-vertex chunk); the frame's 24 transforms sit in LDS.  HBM-bound on the output:
// 12 B per vertex written, weights / template stay in L2.
MP_KERNEL __launch_bounds__(256) void mp_lbs(const float* __restrict__ rglobal, const float* __restrict__ joint,
                                               const float* __restrict__ tran, const float* __restrict__ jrest,
                                               long jrestStride, const float* __restrict__ vrest, long vrestStride,
                                               const float* __restrict__ weights,
                                               int V, float* __restrict__ vert) {
    __shared__ float T[24 * 12];
    const long n = blockIdx.y;
    jrest += n * jrestStride;
    vrest += n * vrestStride;
    float tx = 0.f, ty = 0.f, tz = 0.f;
    if (tran) { tx = tran[n * 3 + 0]; ty = tran[n * 3 + 1]; tz = tran[n * 3 + 2]; }
    if (threadIdx.x < 24) {
        const int j = threadIdx.x;
        const float* R = rglobal + (n * 24 + j) * 9;
        const float* p = joint + (n * 24 + j) * 3;          // already translated by `tran`
        const float jx = jrest[j * 3 + 0], jy = jrest[j * 3 + 1], jz = jrest[j * 3 + 2];
        float* t = T + j * 12;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            t[r * 4 + 0] = R[r * 3 + 0]; t[r * 4 + 1] = R[r * 3 + 1]; t[r * 4 + 2] = R[r * 3 + 2];
            t[r * 4 + 3] = (p[r] - (r == 0 ? tx : (r == 1 ? ty : tz))) - (R[r * 3 + 0] * jx + R[r * 3 + 1] * jy + R[r * 3 + 2] * jz);
        }
    }
    __syncthreads();
    const int v = blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const float vx = vrest[v * 3 + 0], vy = vrest[v * 3 + 1], vz = vrest[v * 3 + 2];
    const float* w = weights + (size_t)v * 24;
    float m[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) m[k] = 0.f;
#pragma unroll
    for (int j = 0; j < 24; ++j) {
        const float wj = w[j];
#pragma unroll
        for (int k = 0; k < 12; ++k) m[k] += wj * T[j * 12 + k];
    }
    float* o = vert + ((size_t)n * V + v) * 3;
    o[0] = m[0] * vx + m[1] * vy + m[2] * vz + m[3] + tx;
    o[1] = m[4] * vx + m[5] * vy + m[6] * vz + m[7] + ty;
    o[2] = m[8] * vx + m[9] * vy + m[10] * vz + m[11] + tz;
}

// ---- zero-pose body of a given shape: ParametricModel.get_zero_pose_joint_and_vertex(shape) (articulate/model.py:84-89)
//   v = shapedirs . shape + v_template;  j = J_regressor v;  j, v = j - j[0], v - j[0];  bone_i = j_i - j_parent(i)
// three small kernels per call (ns = 1 or N bodies); the results feed mp_fk / mp_lbs through their per-frame strides.
MP_KERNEL __launch_bounds__(256) void mp_shape_verts(const float* __restrict__ shape, int ns,
                                                       const float* __restrict__ shapedirs,
                                                       const float* __restrict__ vtemplate, int V,
                                                       float* __restrict__ vraw) {
    const long gid = (long)blockIdx.x * blockDim.x + threadIdx.x;        // (body, vertex, component)
    if (gid >= (long)ns * V * 3) return;
    const long sidx = gid / ((long)V * 3);
    const long vc = gid - sidx * (long)V * 3;
    const float* sd = shapedirs + vc * 10;
    const float* sh = shape + sidx * 10;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < 10; ++k) acc += sh[k] * sd[k];
    vraw[gid] = acc + vtemplate[vc];
}

MP_KERNEL __launch_bounds__(256) void mp_shape_joints(const float* __restrict__ vraw, const float* __restrict__ jreg,
                                                        int V, float* __restrict__ jraw) {
    __shared__ float red[3][256];
    const int j = blockIdx.x;                          // joint
    const long sidx = blockIdx.y;                      // body
    const float* v = vraw + sidx * (long)V * 3;
    const float* w = jreg + (long)j * V;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int k = threadIdx.x; k < V; k += 256) {
        const float wk = w[k];
        a0 += wk * v[k * 3 + 0]; a1 += wk * v[k * 3 + 1]; a2 += wk * v[k * 3 + 2];
    }
    red[0][threadIdx.x] = a0; red[1][threadIdx.x] = a1; red[2][threadIdx.x] = a2;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
            for (int c = 0; c < 3; ++c) red[c][threadIdx.x] += red[c][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x < 3) jraw[(sidx * 24 + j) * 3 + threadIdx.x] = red[threadIdx.x][0];
}

MP_KERNEL __launch_bounds__(256) void mp_shape_align(float* __restrict__ v, const float* __restrict__ jraw,
                                                       const int* __restrict__ parent, int V,
                                                       float* __restrict__ jrest, float* __restrict__ bone) {
    const long sidx = blockIdx.y;
    const float* jr = jraw + sidx * 72;
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < V * 3) v[sidx * (long)V * 3 + k] -= jr[k % 3];
    if (blockIdx.x == 0 && threadIdx.x < 72) {
        const int i = threadIdx.x / 3, c = threadIdx.x % 3;
        const float ji = jr[i * 3 + c] - jr[c];
        jrest[sidx * 72 + threadIdx.x] = ji;
        bone[sidx * 72 + threadIdx.x] = i == 0 ? ji : ji - (jr[parent[i] * 3 + c] - jr[c]);
    }
}

// Pose blend shapes (articulate/model.py:236-238, use_pose_blendshape=True):
//   v_posed[n][v][c] = v_rest[(n)][v][c] + sum_k (pose[n][1 + k/9] - I)[k%9] * posedirs[v][c][k],  k < 207.
// A [F frames x 207] x [207 x 3V] product, bandwidth-bound on the 3V x 207 posedirs (17 MB at V = 6890): posedirs are
// stored TRANSPOSED [207][3V] at upload so that a wave's loads of one k are contiguous; a workgroup keeps the 207
// coefficients of 8 frames in LDS and every posedirs element it loads serves those 8 frames.
constexpr int kBlendFrames = 8;
MP_KERNEL __launch_bounds__(256) void mp_pose_blend(const float* __restrict__ pose, long N, const float* __restrict__ vrest,
                                                      long vrestStride, const float* __restrict__ posedirsT, int V3,
                                                      float* __restrict__ vposed) {
    __shared__ float r[kBlendFrames][208];
    const long n0 = (long)blockIdx.y * kBlendFrames;
    for (int e = threadIdx.x; e < kBlendFrames * 207; e += 256) {
        const int f = e / 207, k = e - f * 207;
        const long n = n0 + f;
        float x = 0.f;
        if (n < N) x = pose[n * 216 + 9 + k] - ((k % 9) % 4 == 0 ? 1.f : 0.f);      // minus the identity's diagonal
        r[f][k] = x;
    }
    __syncthreads();
    const int vc = blockIdx.x * 256 + threadIdx.x;
    if (vc >= V3) return;
    float acc[kBlendFrames];
#pragma unroll
    for (int f = 0; f < kBlendFrames; ++f) acc[f] = 0.f;
    for (int k = 0; k < 207; ++k) {
        const float p = posedirsT[(size_t)k * V3 + vc];
#pragma unroll
        for (int f = 0; f < kBlendFrames; ++f) acc[f] += r[f][k] * p;
    }
#pragma unroll
    for (int f = 0; f < kBlendFrames; ++f) {
        const long n = n0 + f;
        if (n < N) vposed[(size_t)n * V3 + vc] = vrest[n * vrestStride + vc] + acc[f];
    }
}

MP_KERNEL __launch_bounds__(256) void mp_transpose_posedirs(const float* __restrict__ src, int V3, float* __restrict__ dst) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;               // (k, vc) of the destination
    if (gid >= (long)V3 * 207) return;
    const long k = gid / V3, vc = gid - k * V3;
    dst[gid] = src[vc * 207 + k];
}

}  // namespace

void mp_launch_pose_blend(const float* pose, long N, const float* vrest, long vrestStride, const float* posedirsT, int V,
                          float* vposed, hipStream_t s) {
    if (N <= 0 || V <= 0) return;
    const int V3 = V * 3;
    for (long n0 = 0; n0 < N; n0 += 32768L * kBlendFrames) {                // grid.y limit
        const long cnt = N - n0 < 32768L * kBlendFrames ? N - n0 : 32768L * kBlendFrames;
        hipLaunchKernelGGL(mp_pose_blend, dim3((V3 + 255) / 256, (unsigned)((cnt + kBlendFrames - 1) / kBlendFrames)), dim3(256), 0,
                           s, pose + n0 * 216, cnt, vrest + n0 * vrestStride, vrestStride, posedirsT, V3,
                           vposed + (size_t)n0 * V3);
    }
}

void mp_launch_transpose_posedirs(const float* src, int V, float* dst, hipStream_t s) {
    const long n = (long)V * 3 * 207;
    hipLaunchKernelGGL(mp_transpose_posedirs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, src, V * 3, dst);
}

void mp_launch_shape_body(const float* shape, int ns, const float* shapedirs, const float* vtemplate_raw,
                          const float* jreg, const int* parent_dev, int V, float* vrest, float* jraw, float* jrest,
                          float* bone, hipStream_t s) {
    if (ns <= 0 || V <= 0) return;
    const long n = (long)ns * V * 3;
    hipLaunchKernelGGL(mp_shape_verts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, shape, ns, shapedirs,
                       vtemplate_raw, V, vrest);
    hipLaunchKernelGGL(mp_shape_joints, dim3(24, (unsigned)ns), dim3(256), 0, s, vrest, jreg, V, jraw);
    hipLaunchKernelGGL(mp_shape_align, dim3((V * 3 + 255) / 256, (unsigned)ns), dim3(256), 0, s, vrest, jraw, parent_dev,
                       V, jrest, bone);
}

void mp_launch_lbs(const float* rglobal, const float* joint, const float* tran, long N, const float* jrest_dev,
                   long jrestStride, const float* vrest_dev, long vrestStride, const float* weights_dev, int V,
                   float* vert, hipStream_t s) {
    if (N <= 0 || V <= 0) return;
    for (long n0 = 0; n0 < N; n0 += 32768) {                              // grid.y limit
        const long cnt = N - n0 < 32768 ? N - n0 : 32768;
        hipLaunchKernelGGL(mp_lbs, dim3((V + 255) / 256, (unsigned)cnt), dim3(256), 0, s, rglobal + n0 * 216, joint + n0 * 72,
                           tran ? tran + n0 * 3 : nullptr, jrest_dev + n0 * jrestStride, jrestStride,
                           vrest_dev + n0 * vrestStride, vrestStride, weights_dev, V, vert + (size_t)n0 * V * 3);
    }
}

void mp_launch_r6d_ik_strided(const float* r6d, long N, long rowStride, long rowOffset, float* pose,
                              const int* parent_dev, hipStream_t s) {
    if (N <= 0) return;
    const bool aligned = ((reinterpret_cast<uintptr_t>(r6d) | reinterpret_cast<uintptr_t>(pose)) & 15) == 0 &&
                         ((rowStride | rowOffset) & 3) == 0;
    if (aligned && !mp_kin_scalar_forced()) {
        hipLaunchKernelGGL(mp_r6d_ik_lds, dim3((unsigned)((N + kFkFrames - 1) / kFkFrames)), dim3(192), 0, s, r6d, N, rowStride,
                           rowOffset, pose, parent_dev);
        return;
    }
    const long threads = N * 24;
    hipLaunchKernelGGL(mp_r6d_ik, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, r6d, N, rowStride,
                       rowOffset, pose, parent_dev);
}

// r6d -> local pose AND its forward kinematics (shared body, no translation) in one launch; false = buffers not 16-byte aligned
// (or MP_VARIANT kin_scalar=1): the caller runs mp_launch_r6d_ik_strided + mp_launch_fk instead
bool mp_launch_r6d_ik_fk(const float* r6d, long N, long rowStride, long rowOffset, float* pose, const float* bone_dev,
                         const int* parent_dev, float* rglobal, float* joint, hipStream_t s) {
    if (N <= 0) return true;
    const bool aligned = ((reinterpret_cast<uintptr_t>(r6d) | reinterpret_cast<uintptr_t>(pose)) & 15) == 0 &&
                         ((rowStride | rowOffset) & 3) == 0;
    if (!aligned || mp_kin_scalar_forced()) return false;
    hipLaunchKernelGGL(mp_r6d_ik_fk, dim3((unsigned)((N + kFkFrames - 1) / kFkFrames)), dim3(256), 0, s, r6d, N, rowStride, rowOffset,
                       pose, bone_dev, parent_dev, rglobal, joint);
    return true;
}

void mp_launch_global_to_local(const float* rglobal, long N, float* rlocal, const int* parent_dev, hipStream_t s) {
    if (N <= 0) return;
    const bool aligned = ((reinterpret_cast<uintptr_t>(rglobal) | reinterpret_cast<uintptr_t>(rlocal)) & 15) == 0;
    if (aligned && !mp_kin_scalar_forced()) {
        hipLaunchKernelGGL(mp_global_to_local_lds, dim3((unsigned)((N + kFkFrames - 1) / kFkFrames)), dim3(192), 0, s, rglobal, N,
                           rlocal, parent_dev);
        return;
    }
    const long threads = N * 24;
    hipLaunchKernelGGL(mp_global_to_local, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, rglobal, N, rlocal, parent_dev);
}

void mp_launch_r6d_to_rot(const float* r6d, long n, float* out, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(mp_r6d_to_rot, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, r6d, n, out);
}

void mp_launch_r6d_ik(const float* r6d, long N, float* pose, const int* parent_dev, hipStream_t s) {
    mp_launch_r6d_ik_strided(r6d, N, 96, 0, pose, parent_dev, s);
}

void mp_launch_fk(const float* pose, const float* tran, long N, const float* bone_dev, const int* parent_dev,
                  const int* depth_dev, float* rglobal, float* joint, hipStream_t s, long boneStride) {
    if (N <= 0) return;
    const long blocks = (N + 7) / 8;   // 4 waves x 2 frames per block
    hipLaunchKernelGGL(mp_fk, dim3((unsigned)blocks), dim3(256), 0, s, pose, tran, N, bone_dev, boneStride, parent_dev,
                       depth_dev, rglobal, joint);
}

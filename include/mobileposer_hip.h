/*
 * mobileposer_hip.h -- C ABI of libmobileposer_hip.so, the MI355X (gfx950) implementation of the
 * MobilePoser per-frame inference path.
 *
 * The reference (SPICExLAB/MobilePoser) is 100 % Python and has no FFI of its own; the boundary it
 * offers is the method surface of MobilePoserNet.  Every entry point below therefore names the
 * reference METHOD it replaces (paths relative to /root/reference/mobileposer).  The Python facade
 * (mobileposer_amd/net.py) keeps those method signatures and calls these functions through ctypes.
 *
 * Conventions
 *   - plain pointers and sizes only; every `*_dev` pointer is device (HBM) memory owned by the
 *     caller, fp32 row-major; `*_host` pointers are host memory.
 *   - every function returns MP_OK (0) or a negative mp_status; mp_last_error() gives the text.
 *     No C++ exception crosses the ABI.
 *   - a handle is bound to one device, is NOT thread-safe, and owns its weights, workspaces,
 *     streams and the carried state (velocity LSTM state, streaming windows).  Every entry point
 *     selects that device for the duration of the call and puts the calling thread's current
 *     device back before it returns (mp_create* and mp_destroy included).
 *   - `stream` is the caller's HIP stream (void* = hipStream_t; NULL = the legacy default
 *     stream).  Work is ordered after everything already enqueued on it and the caller's later
 *     work on that stream is ordered after the call's results (event fork/join onto the
 *     library's own streams).  Kinematics / solver / evaluator entry points are asynchronous with
 *     respect to the host; the network entry points (mp_forward, mp_forward_offline,
 *     mp_rnn_forward, mp_stream_step) wait for their own completion unless recovery is switched
 *     off (mp_set_recovery, "error behaviour" below).
 */
#ifndef MOBILEPOSER_HIP_H
#define MOBILEPOSER_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mp_handle mp_handle;

typedef enum mp_status {
    MP_OK = 0,
    MP_ERR_INVALID = -1,     /* bad argument (NULL pointer, non-positive size, unknown module ...)   */
    MP_ERR_HIP = -2,         /* a HIP runtime call failed; text in mp_last_error()                   */
    MP_ERR_STATE_SHAPE = -3, /* carried velocity state has another batch size (reference quirk Q2:   */
                             /* nn.LSTM raises when h0's batch differs, models/velocity.py:45-48)    */
    MP_ERR_NO_STREAMS = -4,  /* mp_stream_* called before mp_stream_create                           */
    MP_ERR_LENGTHS = -5,     /* lengths[] not in 1..T or max(lengths) != T (the reference's           */
                             /* torch.cat at models/net.py:106 fails in that case)                    */
    MP_ERR_DEVICE = -6       /* a persistent LSTM kernel gave up a time-bounded wait (its grid was    */
                             /* starved of CUs) and recovery is off or failed: the affected outputs   */
                             /* are NaN.  With recovery on (the default) the call repairs itself and  */
                             /* returns MP_OK; see mp_set_recovery / mp_finish.                        */
} mp_status;

/* module ids for mp_rnn_forward: the four RNN blocks built at models/net.py:40-43 */
enum { MP_MOD_JOINTS = 0, MP_MOD_POSE = 1, MP_MOD_FOOT_CONTACT = 2, MP_MOD_VELOCITY = 3 };

/* Number of fp32 values in the flat weight blob (6 674 994): the 72 state-dict tensors in the
 * reference's registration order (pose.*, joints.*, foot_contact.*, velocity.*), each row-major.
 * Replaces: the `.pth` state_dict written by combine_weights.py:53-56. */
size_t mp_weight_count(void);

/* Build id of this binary: the md5 over the sources it was compiled from (the .hip and .h files of mobileposer_amd/csrc and the headers of include/, in
 * name order -- mobileposer_amd/_lib.py source_md5()), baked in at compile time (-DMP_SRC_MD5).  The Python binding, bench.py
 * and the tests compare it with the md5 of the sources beside them and rebuild (or refuse) on a mismatch, so a stale library
 * can be neither timed nor tested silently.  "unknown" for a build outside of __graft_entry__.build().
 * Replaces: nothing in the reference (pure Python: source and executable are the same file). */
const char* mp_build_id(void);

/* i-th manifest entry (0 <= i < 72): key name, number of dims (1|2), shape, offset in the blob.
 * Replaces: MobilePoserNet.state_dict() key/shape enumeration (utils/model_utils.py:11). */
int mp_manifest_entry(int i, char* name, size_t name_cap, int* ndim, int64_t shape[2], size_t* offset);

/* Create a model instance on `device` from a host weight blob and the SMPL constants the path uses.
 * parent[24]: kinematic tree (parent[0] = -1); J[72]: raw SMPL joint positions (root-aligned inside).
 * Replaces: MobilePoserNet.__init__ (models/net.py:28-74) + load_state_dict in load_model
 * (utils/model_utils.py:6-15) + ParametricModel.__init__ (articulate/model.py:20-39). */
int mp_create(mp_handle** out, int device, const float* weights_host, size_t n_floats,
              const int32_t parent[24], const float J[72]);

/* Same, from a blob that already lives in device memory (e.g. after the RCCL broadcast of the
 * weights from rank 0).  The blob is only read during the call. */
int mp_create_from_device(mp_handle** out, int device, const float* weights_dev, size_t n_floats,
                          const int32_t parent[24], const float J[72]);

/* A body-only instance: the SMPL constants without network weights.  Serves the kinematics entry points (mp_fk,
 * mp_set_mesh, mp_fk_mesh, mp_reduced_global_to_full, mp_translate_offline); the network entry points return
 * MP_ERR_INVALID on it.  Replaces: a stand-alone ParametricModel(paths.smpl_file) as data.py:24 and
 * articulate/evaluator.py:293 build it (articulate/model.py:20-39). */
int mp_create_body(mp_handle** out, int device, const int32_t parent[24], const float J[72]);

void mp_destroy(mp_handle* h);
const char* mp_last_error(const mp_handle* h);   /* h may be NULL: error of a failed mp_create */

/* What the handle found on its device at creation: the device index it is bound to, the number of compute units, and
 * whether the probe saw workgroups dealt round robin over 8 XCDs (bit 0; the side-by-side schedules' placement tables rest
 * on it; speed only) and whether the handle still uses those tables (bit 1: cleared for good by a starvation error).  Any pointer may be NULL.  bench.py logs it per rank.  Replaces: nothing in the reference (`device` global,
 * config.py:9). */
int mp_device_info(const mp_handle* h, int* device, int* n_cu, int* xcd_round_robin);

/* Read constants back: floor_y (models/net.py:49), feet_pos[2*3] (models/net.py:48). */
int mp_get_constants(const mp_handle* h, float* floor_y, float feet_pos[6]);

/* MobilePoserNet.forward (models/net.py:101-119).
 *   imu_dev      [B,T,60]      in
 *   lengths_host [B]           in   (1..T each, max == T; reference passes a Python list)
 *   pose_dev     [B*T,24,3,3]  out  local rotations after _reduced_global_to_full (net.py:93-99)
 *   joints_dev   [B,T,72]      out
 *   vel_dev      [B,T,72]      out  (velocity LSTM state is carried across calls: velocity.py:45-48)
 *   contact_dev  [B,T,2]       out  raw logits
 *   r6d_dev      [B,T,96]      out, optional (NULL to skip): the Poser output before net.py:110 */
int mp_forward(mp_handle* h, const float* imu_dev, const int32_t* lengths_host, int B, int T,
               float* pose_dev, float* joints_dev, float* vel_dev, float* contact_dev,
               float* r6d_dev, void* stream);

/* MobilePoserNet.forward_offline (models/net.py:121-171, PHYSICS off) batched over sequences, as ONE call:
 * mp_forward + the translation solver (tran_dev [B,T,3]) and, when rglobal_dev / joint_dev are given
 * (both or neither), the SMPL forward kinematics of the predicted pose that the evaluator applies next
 * (articulate/evaluator.py:319; R_global [B*T,24,3,3], joint [B*T,24,3], no translation added). */
int mp_forward_offline(mp_handle* h, const float* imu_dev, const int32_t* lengths_host, int B, int T,
                       float* pose_dev, float* joints_dev, float* vel_dev, float* contact_dev, float* tran_dev,
                       float* rglobal_dev, float* joint_dev, void* stream);

/* RNN.forward of one module (models/rnn.py:20-33): Linear+ReLU -> 2-layer LSTM (packed-sequence
 * semantics) -> Linear.  x_dev [B,T,n_in] -> y_dev [B,T,n_out].
 * state_in_dev / state_out_dev: optional (h then c, each [layers*dirs, B, H] contiguous, nn.LSTM
 * order l0, l0_reverse, l1, l1_reverse) -- the `h` argument / third return value of rnn.py:20,33.
 * Any state is accepted, as by nn.LSTM.  A finite initial |h| >= 2 (or inf) -- nothing an LSTM produces -- is outside what the
 * fused layer kernels exchange between workgroups (device code 2000000, a STATE code: the handle's placement tables stay on):
 * with recovery on (the default) the call is run by the per-step kernels and returns the exact result, with recovery off it
 * reports MP_ERR_DEVICE like a starved call.  A NaN initial h -- what the reference's velocity.rnn_state holds for good after
 * one NaN sample (velocity.py:45-48) -- is no error and costs nothing (round 5): that cell is NaN for the whole call, so every
 * output of its sequence is (as in the reference); only the RETURNED state differs from nn.LSTM's when a row of the initial h
 * is NaN in some units and finite in others: nn.LSTM returns NaN for the whole row, this library for those units (every
 * output of that sequence is NaN either way, and stays so in later calls). */
int mp_rnn_forward(mp_handle* h, int module, const float* x_dev, const int32_t* lengths_host,
                   int B, int T, float* y_dev, const float* state_in_dev, float* state_out_dev,
                   void* stream);

/* MobilePoserNet._reduced_global_to_full (models/net.py:93-99): r6d [N,96] -> pose [N,24,3,3]. */
int mp_reduced_global_to_full(mp_handle* h, const float* r6d_dev, int64_t N, float* pose_dev, void* stream);

/* ParametricModel.inverse_kinematics_R (articulate/model.py:146-164 -> articulate/math/spatial.py:197-221): global joint
 * rotations [N,24,3,3] -> local ones, R_local[i] = R_global[parent[i]]^T R_global[i], R_local[0] = R_global[0] -- what
 * MobilePoserNet.global_to_local_pose is bound to (models/net.py:38).  Distinct buffers; works on a body-only handle. */
int mp_inverse_kinematics_r(mp_handle* h, const float* rglobal_dev, int64_t N, float* rlocal_dev, void* stream);

/* art.math.r6d_to_rotation_matrix (articulate/math/angular.py:167-182) on n six-vectors [n,6] -> [n,3,3]: the first two
 * COLUMNS of R, Gram-Schmidt, NaN -> 0.  What evaluate.py:60 applies to the ground-truth pose of a dataset item. */
int mp_r6d_to_rotation_matrix(mp_handle* h, const float* r6d_dev, int64_t n, float* rot_dev, void* stream);

/* The translation solver of forward_offline (models/net.py:130-154), batched over sequences.
 * joints [B,T,72], vel [B,T,72], contact [B,T,2] (as returned by mp_forward) -> tran [B,T,3].
 * Frames t >= lengths[b] get the last valid translation. */
int mp_translate_offline(mp_handle* h, const float* joints_dev, const float* vel_dev,
                         const float* contact_dev, const int32_t* lengths_host, int B, int T,
                         float* tran_dev, void* stream);

/* ParametricModel.forward_kinematics, calc_mesh=False, shape=None (articulate/model.py:208-232).
 * pose [N,24,3,3] local, tran [N,3] optional (NULL) -> R_global [N,24,3,3], joint [N,24,3]. */
int mp_fk(mp_handle* h, const float* pose_dev, const float* tran_dev, int64_t N,
          float* rglobal_dev, float* joint_dev, void* stream);

/* Mesh part of ParametricModel (articulate/model.py:28-35): v_template [V,3] (raw; root-aligned inside) and
 * skinning weights [V,24], both host pointers.  Needed only by mp_fk_mesh. */
int mp_set_mesh(mp_handle* h, const float* v_template_host, const float* weights_host, int n_vertex);

/* ParametricModel.forward_kinematics with calc_mesh=True, shape=None, no pose blendshape
 * unless mp_set_pose_blendshape was called (articulate/model.py:208-240): as mp_fk, plus vert [N,V,3] by linear blend skinning.  This is what
 * FullMotionEvaluator.__call__ runs on prediction and ground truth (articulate/evaluator.py:319-320). */
int mp_fk_mesh(mp_handle* h, const float* pose_dev, const float* tran_dev, int64_t N,
               float* rglobal_dev, float* joint_dev, float* vert_dev, void* stream);

/* Shape space of ParametricModel (articulate/model.py:29,31): shapedirs [V,3,10] and the DENSE J_regressor [24,V]
 * (the pickle holds it scipy-sparse; model.py:29 densifies it), host pointers, V as given to mp_set_mesh. */
int mp_set_shape_space(mp_handle* h, const float* shapedirs_host, const float* j_regressor_host);

/* ParametricModel.forward_kinematics with shape != None (articulate/model.py:208-240 through
 * get_zero_pose_joint_and_vertex(shape), :84-89): v = shapedirs.shape + v_template, j = J_regressor v, both aligned to
 * j[0]; then as mp_fk / mp_fk_mesh on that body.  shape_dev [n_shape,10] with n_shape == 1 (one body for all frames) or
 * n_shape == N (a body per frame); vert_dev [N,V,3] optional (NULL = calc_mesh False). */
int mp_fk_shape(mp_handle* h, const float* pose_dev, const float* shape_dev, int n_shape, const float* tran_dev,
                int64_t N, float* rglobal_dev, float* joint_dev, float* vert_dev, void* stream);

/* ParametricModel.get_zero_pose_joint_and_vertex(shape) (articulate/model.py:77-92, the shape != None branch :84-89):
 * shape_dev [n_shape,10] -> joint_dev [n_shape,24,3] and vert_dev [n_shape,V,3], both aligned to the body's own root joint.
 * (shape == None is two host constants: J - J[0] and v_template - J[0].) */
int mp_zero_pose_body(mp_handle* h, const float* shape_dev, int n_shape, float* joint_dev, float* vert_dev, void* stream);

/* ParametricModel(..., use_pose_blendshape=True) (articulate/model.py:30,236-238): posedirs [V,3,207] (host pointer, V as
 * given to mp_set_mesh).  From then on the mesh of mp_fk_mesh / mp_fk_shape is skinned from
 * v + posedirs . (pose[:, 1:] - I) instead of v.  NULL switches it off again (the default; no reference caller enables it:
 * net.py:37, evaluator.py:293, data.py:24).  mp_eval_metrics follows the reference's evaluator and never applies it. */
int mp_set_pose_blendshape(mp_handle* h, const float* posedirs_host);

/* FullMotionEvaluator.__call__ (articulate/evaluator.py:292-343, mean shape) as PoseEvaluator.eval calls it
 * (evaluate.py:20-29): the joints of ignored_mask (bit j = joint j; evaluate.py:25-26) of both poses are set to the
 * identity, forward kinematics (with linear blend skinning when use_mesh != 0) run on prediction and ground truth, and the
 * 10 x [mean, std] error table is written to table_dev [10][2]: joint position, vertex position (NaN without mesh), local
 * angle, global angle (degrees), predicted / true jerk, 1-second root translation error (x 100), and rows 0, 2, 3 restricted
 * to the joints of joint_mask (0 = no mask: {0, NaN} like torch.zeros(1)).  A matrix without rows (N <= 3, N <= fps) gives
 * NaN, like the reference.  pose_*_dev [N,24,3,3] local rotations, tran_*_dev [N,3] or NULL. */
int mp_eval_metrics(mp_handle* h, const float* pose_p_dev, const float* pose_t_dev, const float* tran_p_dev,
                    const float* tran_t_dev, int64_t N, int fps, int align_joint, unsigned joint_mask,
                    unsigned ignored_mask, int use_mesh, float* table_dev, void* stream);

/* MobilePoserNet.reset (models/net.py:84-88) clears nothing this library owns for the batch path;
 * clear_velocity != 0 additionally drops the carried velocity LSTM state (what setting
 * `model.velocity.rnn_state = None` does in the reference). */
int mp_reset_state(mp_handle* h, int clear_velocity);

/* Carried velocity state, h then c, each [2,B,256] (velocity.py:30,47).  get: returns B (0 = none). */
int mp_get_velocity_state(mp_handle* h, float* state_dev, int* batch);
int mp_set_velocity_state(mp_handle* h, const float* state_dev, int batch);

/* ---- streaming: MobilePoserNet.forward_online (models/net.py:173-219) for S concurrent streams --
 * Each stream owns a 45-frame window (40 past + 5 future, config.py:52-54), last foot positions,
 * current_root_y, last_root_pos and its rows of the velocity LSTM state.  One step consumes one
 * new frame per stream (eager launches on the library's streams; a captured hipGraph with
 * mp_set_graph_mode(h, 1)). */
int mp_stream_create(mp_handle* h, int S);
/* frames [S,60] -> pose [S,24,9], joints [S,45,72] (optional NULL), root_pos [S,3], contact [S,2] */
int mp_stream_step(mp_handle* h, const float* frames_dev, float* pose_dev, float* joints_dev,
                   float* root_pos_dev, float* contact_dev, void* stream);
/* N consecutive MobilePoserNet.forward_online calls of ONE stream (mp_stream_create(h, 1)) in one call -- what evaluate.py:62-64
 * does with `[model.forward_online(f) for f in torch.cat((x, x[-1].repeat(5, 1)))]`, T + 5 calls per sequence.
 *   frames_dev   [N,60]     in   the frames in call order
 *   pose_dev     [N,24,9]   out  pose of window index 40 of every call (net.py:181)
 *   joints_dev   [N,45,72]  out  pred_joints of every call's window, or NULL
 *   root_pos_dev [N,3]      out  last_root_pos after every call (net.py:208)
 *   contact_dev  [N,2]      out  foot-contact logits of window index 40 (net.py:187)
 * Same semantics and carried state as N calls of mp_stream_step (the window before the first call, the velocity LSTM state --
 * every call runs its 45 steps ON the state the previous one left, SURVEY Q6 --, last foot positions, root height / position;
 * all of them are left as the N-th call leaves them), but the joints / pose / foot-contact blocks of all N windows run as ONE
 * N x 45 batch and the velocity block as one N*45-step sequence.  Results agree with the tick-by-tick path to fp32 noise (other
 * kernels serve the other batch shape), not bit for bit.  Exact-fp32 operands only. */
int mp_stream_replay(mp_handle* h, const float* frames_dev, int N, float* pose_dev, float* joints_dev, float* root_pos_dev,
                     float* contact_dev, void* stream);

/* reset(): all streams (mask_host == NULL) or those with mask_host[s] != 0.  Like the reference's
 * reset() it leaves the velocity LSTM state alone unless clear_velocity != 0. */
int mp_stream_reset(mp_handle* h, const uint8_t* mask_host, int clear_velocity);
/* The per-stream variables of the reference object (models/net.py:59-64, updated at :205-208), read back for stream s
 * (synchronises the library stream; every pointer optional): window [45,60] = `self.imu` (device buffer), last_foot
 * [2*3] = last_lfoot_pos, last_rfoot_pos, root_y = current_root_y, root_pos [3] = last_root_pos, fresh != 0 while the
 * stream has not seen a frame since reset() (`self.imu is None`). */
int mp_stream_get_state(mp_handle* h, int s, float* window_dev, float last_foot_host[6], double* root_y_host,
                        float root_pos_host[3], int* fresh_host);

/* The counterpart: assignments to those variables (`model.last_root_pos = ...`, `model.imu = None` -> fresh != 0, as
 * viewer / live code resets them).  Every pointer optional; synchronises the library stream. */
int mp_stream_set_state(mp_handle* h, int s, const float* window_dev, const float last_foot_host[6],
                        const double* root_y_host, const float root_pos_host[3], const int* fresh_host);

/* ---- live front-end: raw sensor samples -> network input frames, S streams per call --------------------------------
 * The per-frame arithmetic between the sensor packets and forward_online in the reference's live demo
 * (mobileposer/live_demo.py:213-236; calibration quantities from :161-174), batched over streams:
 *   quat [S,5,4] wxyz (unnormalised), acc [S,5,3] m/s^2, smpl2imu [S,3,3], device2bone [S,5,3,3], acc_offsets [S,5,3]
 *   -> frames [S,60] = [5 x 3 accelerations / acc_scale | 5 x 3x3 orientations], slots in network order (sensors
 *   [1,4,3,0,2]); slots whose bit in keep_mask is clear are zero (device-location combo, config.py:60-73).
 * The output is what mp_stream_step takes as frames_dev. */
int mp_live_form_frames(mp_handle* h, const float* quat_dev, const float* acc_dev, const float* smpl2imu_dev,
                        const float* device2bone_dev, const float* acc_offsets_dev, unsigned keep_mask, int S,
                        float* frames_dev, void* stream);

/* ---- measurement hooks (bench.py) ---------------------------------------------------------------
 * With timing on, every call runs eagerly and brackets each kernel launch of the classes below
 * with HIP events on the library stream that launches it.  mp_timing_read returns, for the last call, the
 * number of launches, the summed event-measured milliseconds and the algorithmic GFLOP of a class:
 *   0 = MFMA GEMM (linear1 / linear2),   1 = fused LSTM layer H=256 bidirectional K_in=256,
 *   4 = fused LSTM layer H=256 bidirectional K_in=512,   5 = fused LSTM layer H=256 unidirectional,
 *   6 = fused LSTM layer H=64,   7 = per-step LSTM kernels (fallback mode),   2 = r6d/IK,   3 = whole call. */
int mp_timing_enable(mp_handle* h, int on);
int mp_timing_read(mp_handle* h, int cls, int* launches, float* ms, double* gflop);
/* 0 (default): eager launches on the library's three streams; 1 (env MP_GRAPH=1): capture every (entry point, shape,
 * buffer set) once into a hipGraph and replay it; 2: the same as a SINGLE-BRANCH graph (every launch captured on one
 * stream, no parallel branches -- nothing for the graph executor's stream assignment to get wrong).  The multi-branch
 * graph executor of the HIP runtime in this image can crash in hipGraphLaunch depending on the hardware-queue placement of
 * the streams a process has created (profiles/r02_hipgraph_segv.md; GPU_MAX_HW_QUEUES=8 avoids it), so since round 5 a
 * request for mode 1 gives mode 2 unless the environment says MP_GRAPH_MULTIBRANCH=1 (mode 2 is bitwise equal and as
 * fast: 2.786 vs 2.782 ms per 512-stream tick); outputs are bitwise identical in all three modes. */
int mp_set_graph_mode(mp_handle* h, int on);
/* LSTM implementation of the H = 256 layers (the H = 64 foot-contact block always uses the fp32 kernels):
 *   1 (default; env MP_LSTM_MODE=fp32): fused persistent layer kernels, one launch per layer, exact-fp32 MFMA operands
 *       (v_mfma_f32_16x16x4_f32, mp_lstm_persist.hip) -- the reference's arithmetic;
 *   3 (opt-in; env MP_LSTM_MODE=x3): the same layers (and the linear layers of the H = 256 blocks) with every fp32 product
 *       as hi*hi + hi*lo + lo*hi of two 16-bit halves per operand on v_mfma_f32_16x16x32_f16, fp32 accumulate, fp32 state
 *       (mp_lstm_x3.hip), 2.1x faster.  Since round 4 the halves are IEEE fp16 (hi + lo = 24 significand bits; weights are
 *       split as 16 w so that their low half stays a normal fp16 number): measured on init-scale AND on trained-regime weights
 *       (saturated gates, recurrent gain > 1) the outputs are as close to float64 as mode 1's and PyTorch's CPU path's are
 *       (profiles/r04_accuracy.json; 1.8-3.3 x the fp32 oracle's own noise at the worst element of a 256 x 125 batch, mode 1:
 *       0.9-1.5 x).  NOT bit-compatible with mode 1: a 300-shape fuzz on trained-regime weights found outputs of the two modes
 *       up to 1.5e-3 apart (profiles/r04_validation5.txt) although each is within the tests' bounds of the oracle at the tested
 *       sizes; operands above 65504 (weights: 4094) in magnitude turn into inf / NaN.  Narrower than the reference's
 *       arithmetic: never the credited configuration.
 *       (Rounds 1-3 used bf16 halves -- 17 bits: fine on init-scale weights, 2e-3 .. 2e-2 off on trained-regime ones.)
 *   2: accepted and treated as 1 (rounds 1-4: mode 1 with the velocity block as one two-layer wavefront launch of a separate
 *       kernel; since round 5 mode 1's full-batch schedule runs the velocity block as a wavefront of the 8-slice kernel anyway);
 *   0 (env MP_LSTM_MODE=step): input-projection GEMM + one launch per time step. */
int mp_set_lstm_mode(mp_handle* h, int mode);
/* ---- error behaviour of the persistent LSTM kernels ---------------------------------------------------------------
 * A fused layer launch is a grid of workgroups that wait for each other's hidden state every time step; all of them
 * must be resident at once.  The library plans its launches for a GPU it has to itself; when something else (another
 * process, another handle on another host thread) occupies CUs, a wait can starve.  Every wait is bounded in TIME
 * (0.25 s of the device's constant clock).  A wave that gives up (1) stores 1+step (1000000 = start-up handshake) into
 * the handle's error word and (2) turns its cell state into NaN, so that its slab's rows of every output of the call
 * become NaN -- a starved call never returns plausible numbers (the reference's PyTorch path never returns wrong
 * numbers silently either).
 *
 * mp_set_recovery(h, 1) -- the DEFAULT: mp_forward, mp_forward_offline, mp_rnn_forward and mp_stream_step wait for
 * their own completion before they return (the reference's calls are synchronous too) and, if the error word is set,
 * restore the carried state the call started from and run the call again with per-step kernels (LSTM mode 0: no
 * cross-workgroup waits, parity-tested like the fused kernels).  The call then returns MP_OK, mp_last_error() holds a
 * warning, mp_recovery_count() counts such calls, and the handle stops using physical-XCD placement tables.
 * mp_set_recovery(h, 0): calls return as soon as their work is enqueued (throughput loops); an error is reported as
 * MP_ERR_DEVICE by the next API entry, by mp_finish() or read by mp_device_error(); the outputs of the failed call are NaN
 * AND the state it carried forward is lost: when the error is reported the handle drops the carried velocity LSTM state
 * (as mp_reset_state(h, 1)) and puts every stream back to its state after mp_stream_create (fresh window, root height /
 * position 0, rest-pose foot positions) -- later calls are finite again, but the caller's sequences start over. */
int mp_set_recovery(mp_handle* h, int on);
/* Number of calls that were repaired by the mode-0 re-run since mp_create. */
int mp_recovery_count(const mp_handle* h);
/* Waits for everything the handle has enqueued; MP_OK, or MP_ERR_DEVICE if a kernel gave up a wait since the last
 * report (cleared).  What a caller with recovery off runs before it trusts the outputs of its last call. */
int mp_finish(mp_handle* h);
/* Synchronises the library's stream and returns (and clears) the error word of the persistent kernels:
 * 0 = ok, 1+step = a bounded wait for another workgroup's hidden state timed out (1000000: start-up handshake). */
int mp_device_error(mp_handle* h, int* code);
#ifdef __cplusplus
}
#endif
#endif /* MOBILEPOSER_HIP_H */

// The forward / kinematics / evaluator / state entry points of the C ABI (include/mobileposer_hip.h): mp_forward,
// mp_forward_offline, mp_rnn_forward, mp_reduced_global_to_full, mp_translate_offline, mp_fk*, mp_eval_metrics, the velocity
// state, the operand and graph modes.  Each one: enter (a pending device error is reported first, the caller's stream is
// joined) -> plan for the shape -> the schedule of mp_schedule.hip, eager or as a captured graph (run_maybe_graph) ->
// finish_or_recover -> leave.  The rest of the host side: mp_host.h has the map.
//
// Launches are eager by default, or, opt-in, captured once per (shape, buffer set) into a hipGraph and replayed.
// Why eager is the default: the multi-branch graph executor of the HIP runtime this image ships (libamdhip64 of
// ROCm 7.0, hip::Graph::UpdateStreams) picks max_streams-1 of the max_streams internal streams a hipGraphExec_t owns,
// skipping those that share a hardware queue with the launch stream, WITHOUT a bounds check: when two of them map to
// the launch stream's queue (GPU_MAX_HW_QUEUES = 4 by default, so this depends on every stream the process has ever
// created) it reads past the end of the vector and the process dies with SIGSEGV inside hipGraphLaunch
// (profiles/r02_hipgraph_segv.md: backtrace, disassembly, and the GPU_MAX_HW_QUEUES experiment).  Eager launches on
// the library's streams measure the same step time; graph mode 1 means mode 2 (single-branch) unless the environment asks
// for the real thing (mp_handle.hip multibranch_graphs_allowed).
#include "mp_host.h"

// ================================================================================================ C ABI
extern "C" {

int mp_forward(mp_handle* h, const float* imu_dev, const int32_t* lengths_host, int B, int T, float* pose_dev,
               float* joints_dev, float* vel_dev, float* contact_dev, float* r6d_dev, void* stream) {
    if (!h) return MP_ERR_INVALID;
    if (int rc = need_weights(h, "mp_forward")) return rc;
    if (!imu_dev || !lengths_host || !pose_dev || !joints_dev || !vel_dev || !contact_dev || B < 1 || T < 1)
        return fail(h, MP_ERR_INVALID, "mp_forward: NULL buffer or non-positive shape");
    if (h->vstate.B != 0 && h->vstate.B != B)
        return fail(h, MP_ERR_STATE_SHAPE, "carried velocity state has batch %d, call has batch %d (reset it first; "
                    "the reference raises here too, velocity.py:45-48)", h->vstate.B, B);
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    Plan* p = nullptr;
    if (int rc = get_plan(h, B, T, &p)) return rc;
    if (int rc = upload_lengths(h, p, lengths_host)) return rc;
    if (int rc = ensure_vstate(h, h->vstate, B)) return rc;
    const bool has_state = h->vstate.B == B;
    float* r6d = r6d_dev ? r6d_dev : p->r6d;
    h->segs.clear(); h->ev_used = 0;
    if (int rc = snapshot_vstate(h, B, has_state)) return rc;
    GraphKey key;
    memset(&key, 0, sizeof(key));
    key.kind = 0; key.B = B; key.T = T; key.flags = (has_state ? 1 : 0) | (h->persist ? 2 : 0) | (h->x3 ? 8 : 0);
    key.p[0] = imu_dev; key.p[1] = pose_dev; key.p[2] = joints_dev; key.p[3] = vel_dev; key.p[4] = contact_dev;
    key.p[5] = r6d; key.p[6] = h->vstate.h;
    auto body = [&]() {
        return forward_body(h, p, imu_dev, pose_dev, (long)B * T, 96, 0, joints_dev, vel_dev, contact_dev, r6d,
                            h->vstate, has_state);
    };
    int rc;
    {
        SegScope whole(h, h->s_main, 3, 1);
        rc = run_maybe_graph(h, key, body);
    }
    if (rc) return rc;
    if (int rc2 = finish_or_recover(h, p, "mp_forward", [&]() { return restore_vstate(h, B, has_state); }, body)) return rc2;
    h->vstate.B = B;
    return leave(h, stream);
}

int mp_forward_offline(mp_handle* h, const float* imu_dev, const int32_t* lengths_host, int B, int T, float* pose_dev,
                       float* joints_dev, float* vel_dev, float* contact_dev, float* tran_dev, float* rglobal_dev,
                       float* joint_dev, void* stream) {
    if (!h) return MP_ERR_INVALID;
    if (int rc = need_weights(h, "mp_forward_offline")) return rc;
    if (!imu_dev || !lengths_host || !pose_dev || !joints_dev || !vel_dev || !contact_dev || !tran_dev || B < 1 || T < 1 ||
        ((rglobal_dev == nullptr) != (joint_dev == nullptr)))
        return fail(h, MP_ERR_INVALID, "mp_forward_offline: NULL buffer or non-positive shape");
    if (h->vstate.B != 0 && h->vstate.B != B)
        return fail(h, MP_ERR_STATE_SHAPE, "carried velocity state has batch %d, call has batch %d", h->vstate.B, B);
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    Plan* p = nullptr;
    if (int rc = get_plan(h, B, T, &p)) return rc;
    if (int rc = upload_lengths(h, p, lengths_host)) return rc;
    if (int rc = ensure_vstate(h, h->vstate, B)) return rc;
    const bool has_state = h->vstate.B == B;
    h->segs.clear(); h->ev_used = 0;
    if (int rc = snapshot_vstate(h, B, has_state)) return rc;
    GraphKey key;
    memset(&key, 0, sizeof(key));
    key.kind = 2; key.B = B; key.T = T; key.flags = (has_state ? 1 : 0) | (h->persist ? 2 : 0) | (h->x3 ? 8 : 0);
    key.p[0] = imu_dev; key.p[1] = pose_dev; key.p[2] = joints_dev; key.p[3] = vel_dev; key.p[4] = contact_dev;
    key.p[5] = tran_dev; key.p[6] = h->vstate.h; key.p[7] = rglobal_dev;
    auto body = [&]() {
        bool tail = false;                 // pose's linear2 / IK / FK still running on the side stream (the solver does not read them)
        if (int r = forward_body(h, p, imu_dev, pose_dev, (long)B * T, 96, 0, joints_dev, vel_dev, contact_dev, p->r6d,
                                 h->vstate, has_state, rglobal_dev, joint_dev, &tail)) return r;
        mp_launch_translate_offline(joints_dev, vel_dev, contact_dev, p->lengths_dev, B, T, h->floor_y, tran_dev,
                                    h->s_main);                                       // net.py:130-154
        if (tail) HIPCHK(h, hipStreamWaitEvent(h->s_main, h->ev_x[3], 0));
        HIPCHK(h, hipGetLastError());
        return (int)MP_OK;
    };
    int rc;
    {
        SegScope whole(h, h->s_main, 3, 1);
        rc = run_maybe_graph(h, key, body);
    }
    if (rc) return rc;
    if (int rc2 = finish_or_recover(h, p, "mp_forward_offline", [&]() { return restore_vstate(h, B, has_state); }, body)) return rc2;
    h->vstate.B = B;
    return leave(h, stream);
}

int mp_rnn_forward(mp_handle* h, int module, const float* x_dev, const int32_t* lengths_host, int B, int T,
                   float* y_dev, const float* state_in_dev, float* state_out_dev, void* stream) {
    if (!h) return MP_ERR_INVALID;
    if (int rc = need_weights(h, "mp_rnn_forward")) return rc;
    if (module < 0 || module > 3 || !x_dev || !y_dev || !lengths_host || B < 1 || T < 1)
        return fail(h, MP_ERR_INVALID, "mp_rnn_forward: bad argument");
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    Plan* p = nullptr;
    if (int rc = get_plan(h, B, T, &p)) return rc;
    if (int rc = upload_lengths(h, p, lengths_host)) return rc;
    const ModuleW& m = h->mod[module];
    const size_t half = (size_t)2 * m.dirs * B * m.H;
    const RowMap none{nullptr, 0, 0, 0};
    h->segs.clear(); h->ev_used = 0;
    RnnJob job{h, p, module, user_map(x_dev, T, m.n_in), none, y_dev, (long)T * m.n_out, m.n_out,
               state_in_dev ? STATE_FROM : STATE_ZERO, state_in_dev, state_in_dev ? state_in_dev + half : nullptr,
               state_out_dev, state_out_dev ? state_out_dev + half : nullptr};
    // state_in_dev == state_out_dev: the fused kernels update the state in place, so a starved run would leave NaN where the
    // repair has to start from -- keep a copy for the restore (recovery on only)
    const bool aliased = state_in_dev && state_in_dev == state_out_dev;
    const size_t state_bytes = 2 * half * sizeof(float);
    if (aliased && h->recovery) {
        if (state_bytes > h->rnn_snap_bytes) {
            HIPCHK(h, hipStreamSynchronize(h->s_main));
            if (h->rnn_snap) (void)hipFree(h->rnn_snap);
            h->rnn_snap = nullptr; h->rnn_snap_bytes = 0;
            if (int rc = dev_alloc(h, (void**)&h->rnn_snap, state_bytes)) return rc;
            h->rnn_snap_bytes = state_bytes;
        }
        HIPCHK(h, hipMemcpyAsync(h->rnn_snap, state_in_dev, state_bytes, hipMemcpyDeviceToDevice, h->s_main));
    }
    int rc = run_rnn(job, h->s_main);
    if (rc) return rc;
    if (int rc2 = finish_or_recover(h, p, "mp_rnn_forward", [&]() {
            if (aliased) HIPCHK(h, hipMemcpyAsync(state_out_dev, h->rnn_snap, state_bytes, hipMemcpyDeviceToDevice, h->s_main));
            return (int)MP_OK;
        }, [&]() { return run_rnn(job, h->s_main); })) return rc2;
    return leave(h, stream);
}

int mp_reduced_global_to_full(mp_handle* h, const float* r6d_dev, int64_t N, float* pose_dev, void* stream) {
    if (!h || !r6d_dev || !pose_dev || N < 0) return h ? fail(h, MP_ERR_INVALID, "mp_reduced_global_to_full: bad argument") : MP_ERR_INVALID;
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    mp_launch_r6d_ik(r6d_dev, (long)N, pose_dev, h->parent_dev, h->s_main);
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_inverse_kinematics_r(mp_handle* h, const float* rglobal_dev, int64_t N, float* rlocal_dev, void* stream) {
    if (!h || !rglobal_dev || !rlocal_dev || N < 0) return h ? fail(h, MP_ERR_INVALID, "mp_inverse_kinematics_r: bad argument") : MP_ERR_INVALID;
    if (rglobal_dev == rlocal_dev) return fail(h, MP_ERR_INVALID, "mp_inverse_kinematics_r: in-place call (a joint reads its parent's INPUT)");
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    mp_launch_global_to_local(rglobal_dev, (long)N, rlocal_dev, h->parent_dev, h->s_main);
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_r6d_to_rotation_matrix(mp_handle* h, const float* r6d_dev, int64_t n, float* rot_dev, void* stream) {
    if (!h || !r6d_dev || !rot_dev || n < 0) return h ? fail(h, MP_ERR_INVALID, "mp_r6d_to_rotation_matrix: bad argument") : MP_ERR_INVALID;
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    mp_launch_r6d_to_rot(r6d_dev, (long)n, rot_dev, h->s_main);
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_translate_offline(mp_handle* h, const float* joints_dev, const float* vel_dev, const float* contact_dev,
                         const int32_t* lengths_host, int B, int T, float* tran_dev, void* stream) {
    if (!h || !joints_dev || !vel_dev || !contact_dev || !lengths_host || !tran_dev || B < 1 || T < 1)
        return h ? fail(h, MP_ERR_INVALID, "mp_translate_offline: bad argument") : MP_ERR_INVALID;
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    Plan* p = nullptr;
    if (int rc = get_plan(h, B, T, &p)) return rc;
    if (int rc = upload_lengths(h, p, lengths_host)) return rc;
    mp_launch_translate_offline(joints_dev, vel_dev, contact_dev, p->lengths_dev, B, T, h->floor_y, tran_dev, h->s_main);
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_fk(mp_handle* h, const float* pose_dev, const float* tran_dev, int64_t N, float* rglobal_dev, float* joint_dev,
          void* stream) {
    if (!h || !pose_dev || !rglobal_dev || !joint_dev || N < 0) return h ? fail(h, MP_ERR_INVALID, "mp_fk: bad argument") : MP_ERR_INVALID;
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    mp_launch_fk(pose_dev, tran_dev, (long)N, h->bone_dev, h->parent_dev, h->depth_dev, rglobal_dev, joint_dev, h->s_main);
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_set_mesh(mp_handle* h, const float* v_template_host, const float* weights_host, int n_vertex) {
    if (!h || !v_template_host || !weights_host || n_vertex < 1) return h ? fail(h, MP_ERR_INVALID, "mp_set_mesh: bad argument") : MP_ERR_INVALID;
    ON_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    for (float** q : {&h->vrest_dev, &h->skinw_dev, &h->vtpl_dev, &h->shapedirs_dev, &h->jreg_dev, &h->posedirsT_dev}) {
        if (*q) (void)hipFree(*q);
        *q = nullptr;
    }
    h->n_vertex = 0;
    std::vector<float> v((size_t)n_vertex * 3);
    for (int i = 0; i < n_vertex; ++i)
        for (int c = 0; c < 3; ++c) v[(size_t)i * 3 + c] = v_template_host[(size_t)i * 3 + c] - h->J0[c];   // model.py:87
    if (int rc = dev_alloc(h, (void**)&h->vrest_dev, v.size() * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->skinw_dev, (size_t)n_vertex * 24 * sizeof(float))) return rc;
    HIPCHK(h, hipMemcpy(h->vrest_dev, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->skinw_dev, weights_host, (size_t)n_vertex * 24 * sizeof(float), hipMemcpyHostToDevice));
    if (int rc = dev_alloc(h, (void**)&h->vtpl_dev, v.size() * sizeof(float))) return rc;
    HIPCHK(h, hipMemcpy(h->vtpl_dev, v_template_host, v.size() * sizeof(float), hipMemcpyHostToDevice));
    h->n_vertex = n_vertex;
    return MP_OK;
}

int mp_set_shape_space(mp_handle* h, const float* shapedirs_host, const float* j_regressor_host) {
    if (!h || !shapedirs_host || !j_regressor_host) return h ? fail(h, MP_ERR_INVALID, "mp_set_shape_space: bad argument") : MP_ERR_INVALID;
    if (!h->n_vertex) return fail(h, MP_ERR_INVALID, "mp_set_shape_space before mp_set_mesh");
    ON_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    for (float** q : {&h->shapedirs_dev, &h->jreg_dev}) { if (*q) (void)hipFree(*q); *q = nullptr; }
    const size_t V = (size_t)h->n_vertex;
    if (int rc = dev_alloc(h, (void**)&h->shapedirs_dev, V * 30 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->jreg_dev, V * 24 * sizeof(float))) return rc;
    HIPCHK(h, hipMemcpy(h->shapedirs_dev, shapedirs_host, V * 30 * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->jreg_dev, j_regressor_host, V * 24 * sizeof(float), hipMemcpyHostToDevice));
    return MP_OK;
}

// forward_kinematics in all its forms (articulate/model.py:208-240): optional shape (n_shape bodies: 1 or N), optional
// mesh, optional pose blend shapes (the handle's, mp_set_pose_blendshape).  Workspace layout (floats):
//   vrest [ns][V][3] | jraw, jrest, bone [ns][72] each | vposed [N][V][3] (pose blend shapes only)
namespace {
int ensure_shape_ws(mp_handle* h, size_t need) {
    if (need <= h->shape_ws_floats) return MP_OK;
    HIPCHK(h, hipDeviceSynchronize());
    if (h->shape_ws) (void)hipFree(h->shape_ws);
    h->shape_ws = nullptr; h->shape_ws_floats = 0;
    if (int rc = dev_alloc(h, (void**)&h->shape_ws, need * sizeof(float))) return rc;
    h->shape_ws_floats = need;
    return MP_OK;
}

int fk_general(mp_handle* h, const char* what, const float* pose_dev, const float* shape_dev, int n_shape,
               const float* tran_dev, int64_t N, float* rglobal_dev, float* joint_dev, float* vert_dev, void* stream) {
    if (N == 0) return MP_OK;
    if (vert_dev && !h->n_vertex) return fail(h, MP_ERR_INVALID, "%s before mp_set_mesh", what);
    if (shape_dev && !h->shapedirs_dev) return fail(h, MP_ERR_INVALID, "%s before mp_set_shape_space", what);
    ON_DEVICE(h);
    const size_t V = (size_t)h->n_vertex, ns = shape_dev ? (size_t)n_shape : 0;
    const bool blend = vert_dev && h->posedirsT_dev;
    const size_t f_body = ns * (V * 3 + 3 * 72);
    if (int rc = ensure_shape_ws(h, f_body + (blend ? (size_t)N * V * 3 : 0))) return rc;
    if (int rc = enter(h, stream)) return rc;
    const float *bone = h->bone_dev, *jrest = h->jrest_dev, *vrest = h->vrest_dev;
    long bstride = 0, vstride = 0;
    if (shape_dev) {
        float* vr = h->shape_ws;
        float* jraw = vr + ns * V * 3;
        float* jr = jraw + ns * 72;
        float* bn = jr + ns * 72;
        mp_launch_shape_body(shape_dev, n_shape, h->shapedirs_dev, h->vtpl_dev, h->jreg_dev, h->parent_dev, h->n_vertex, vr,
                             jraw, jr, bn, h->s_main);                                        // model.py:84-89
        bone = bn; jrest = jr; vrest = vr;
        if (n_shape != 1) { bstride = 72; vstride = (long)V * 3; }
    }
    mp_launch_fk(pose_dev, tran_dev, (long)N, bone, h->parent_dev, h->depth_dev, rglobal_dev, joint_dev, h->s_main, bstride);
    if (vert_dev) {
        if (blend) {                                                                          // model.py:236-238
            float* vposed = h->shape_ws + f_body;
            mp_launch_pose_blend(pose_dev, (long)N, vrest, vstride, h->posedirsT_dev, h->n_vertex, vposed, h->s_main);
            vrest = vposed; vstride = (long)V * 3;
        }
        mp_launch_lbs(rglobal_dev, joint_dev, tran_dev, (long)N, jrest, bstride, vrest, vstride, h->skinw_dev, h->n_vertex,
                      vert_dev, h->s_main);
    }
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}
}  // namespace

int mp_fk_shape(mp_handle* h, const float* pose_dev, const float* shape_dev, int n_shape, const float* tran_dev, int64_t N,
                float* rglobal_dev, float* joint_dev, float* vert_dev, void* stream) {
    if (!h || !pose_dev || !shape_dev || !rglobal_dev || !joint_dev || N < 0 || !(n_shape == 1 || n_shape == N))
        return h ? fail(h, MP_ERR_INVALID, "mp_fk_shape: bad argument (n_shape must be 1 or N)") : MP_ERR_INVALID;
    return fk_general(h, "mp_fk_shape", pose_dev, shape_dev, n_shape, tran_dev, N, rglobal_dev, joint_dev, vert_dev, stream);
}

int mp_fk_mesh(mp_handle* h, const float* pose_dev, const float* tran_dev, int64_t N, float* rglobal_dev, float* joint_dev,
               float* vert_dev, void* stream) {
    if (!h || !pose_dev || !rglobal_dev || !joint_dev || !vert_dev || N < 0) return h ? fail(h, MP_ERR_INVALID, "mp_fk_mesh: bad argument") : MP_ERR_INVALID;
    return fk_general(h, "mp_fk_mesh", pose_dev, nullptr, 0, tran_dev, N, rglobal_dev, joint_dev, vert_dev, stream);
}

int mp_set_pose_blendshape(mp_handle* h, const float* posedirs_host) {
    if (!h) return MP_ERR_INVALID;
    ON_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    if (h->posedirsT_dev) { (void)hipFree(h->posedirsT_dev); h->posedirsT_dev = nullptr; }
    if (!posedirs_host) return MP_OK;                                          // use_pose_blendshape = False
    if (!h->n_vertex) return fail(h, MP_ERR_INVALID, "mp_set_pose_blendshape before mp_set_mesh");
    const size_t n = (size_t)h->n_vertex * 3 * 207;
    float* staging = nullptr;
    if (int rc = dev_alloc(h, (void**)&staging, n * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->posedirsT_dev, n * sizeof(float))) { (void)hipFree(staging); return rc; }
    hipError_t e = hipMemcpy(staging, posedirs_host, n * sizeof(float), hipMemcpyHostToDevice);
    if (e == hipSuccess) { mp_launch_transpose_posedirs(staging, h->n_vertex, h->posedirsT_dev, h->s_main); e = hipStreamSynchronize(h->s_main); }
    (void)hipFree(staging);
    if (e != hipSuccess) return fail(h, MP_ERR_HIP, "mp_set_pose_blendshape: %s", hipGetErrorString(e));
    return MP_OK;
}

int mp_zero_pose_body(mp_handle* h, const float* shape_dev, int n_shape, float* joint_dev, float* vert_dev, void* stream) {
    if (!h || !shape_dev || n_shape < 1 || !joint_dev || !vert_dev)
        return h ? fail(h, MP_ERR_INVALID, "mp_zero_pose_body: bad argument") : MP_ERR_INVALID;
    if (!h->shapedirs_dev) return fail(h, MP_ERR_INVALID, "mp_zero_pose_body before mp_set_shape_space");
    ON_DEVICE(h);
    const size_t ns = (size_t)n_shape;
    if (int rc = ensure_shape_ws(h, ns * 2 * 72)) return rc;
    if (int rc = enter(h, stream)) return rc;
    // vertices and joints go straight into the caller's buffers; the raw joints and the bone vectors are scratch
    mp_launch_shape_body(shape_dev, n_shape, h->shapedirs_dev, h->vtpl_dev, h->jreg_dev, h->parent_dev, h->n_vertex, vert_dev,
                         h->shape_ws, joint_dev, h->shape_ws + ns * 72, h->s_main);           // model.py:84-89
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_eval_metrics(mp_handle* h, const float* pose_p_dev, const float* pose_t_dev, const float* tran_p_dev,
                    const float* tran_t_dev, int64_t N, int fps, int align_joint, unsigned joint_mask, unsigned ignored_mask,
                    int use_mesh, float* table_dev, void* stream) {
    if (!h || !pose_p_dev || !pose_t_dev || !table_dev || N < 1 || fps < 1 || align_joint < 0 || align_joint > 23)
        return h ? fail(h, MP_ERR_INVALID, "mp_eval_metrics: bad argument") : MP_ERR_INVALID;
    if (use_mesh && !h->n_vertex) return fail(h, MP_ERR_INVALID, "mp_eval_metrics: use_mesh without mp_set_mesh");
    ON_DEVICE(h);
    const size_t V = use_mesh ? (size_t)h->n_vertex : 0, n = (size_t)N;
    const size_t f_pose = n * 216, f_j = n * 72, f_v = n * V * 3;
    const size_t floats = 2 * (2 * f_pose + f_j + f_v);
    const size_t bytes = floats * sizeof(float) + mp_eval_partial_doubles((int)V) * sizeof(double) + 64;
    if (bytes > h->eval_ws_bytes) {
        HIPCHK(h, hipDeviceSynchronize());
        if (h->eval_ws) (void)hipFree(h->eval_ws);
        h->eval_ws = nullptr; h->eval_ws_bytes = 0;
        if (int rc = dev_alloc(h, (void**)&h->eval_ws, bytes)) return rc;
        h->eval_ws_bytes = bytes;
    }
    if (int rc = enter(h, stream)) return rc;
    float* w = h->eval_ws;
    float* mp_ = w;            float* mt_ = mp_ + f_pose;        // masked local poses
    float* rp = mt_ + f_pose;  float* rt = rp + f_pose;          // global rotations
    float* jp = rt + f_pose;   float* jt = jp + f_j;             // joints
    float* vp = jt + f_j;      float* vt = vp + f_v;             // vertices
    double* part = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(vt + f_v) + 63) & ~(uintptr_t)63);
    hipStream_t s = h->s_main;
    mp_launch_mask_pose(pose_p_dev, mp_, (long)N, ignored_mask, s);                                  // evaluate.py:25-26
    mp_launch_mask_pose(pose_t_dev, mt_, (long)N, ignored_mask, s);
    const float* pose[2] = {mp_, mt_};
    const float* tran[2] = {tran_p_dev, tran_t_dev};
    float* rg[2] = {rp, rt}; float* jj[2] = {jp, jt}; float* vv[2] = {vp, vt};
    for (int k = 0; k < 2; ++k) {                                                                    // evaluator.py:319-320
        mp_launch_fk(pose[k], tran[k], (long)N, h->bone_dev, h->parent_dev, h->depth_dev, rg[k], jj[k], s);
        if (V) mp_launch_lbs(rg[k], jj[k], tran[k], (long)N, h->jrest_dev, 0, h->vrest_dev, 0, h->skinw_dev, h->n_vertex, vv[k], s);
    }
    mp_launch_eval_metrics(mp_, mt_, rp, rt, jp, jt, V ? vp : nullptr, V ? vt : nullptr, (long)N, (int)V, fps, align_joint,
                           joint_mask, part, table_dev, s);                                          // evaluator.py:321-343
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_reset_state(mp_handle* h, int clear_velocity) {
    if (!h) return MP_ERR_INVALID;
    if (clear_velocity) h->vstate.B = 0;
    return MP_OK;
}

int mp_get_velocity_state(mp_handle* h, float* state_dev, int* batch) {
    if (!h || !batch) return MP_ERR_INVALID;
    ON_DEVICE(h);
    *batch = h->vstate.B;
    if (h->vstate.B && state_dev) {
        const size_t n = (size_t)2 * h->vstate.B * 256;
        HIPCHK(h, hipStreamSynchronize(h->s_main));
        HIPCHK(h, hipMemcpy(state_dev, h->vstate.h, n * sizeof(float), hipMemcpyDeviceToDevice));
        HIPCHK(h, hipMemcpy(state_dev + n, h->vstate.c, n * sizeof(float), hipMemcpyDeviceToDevice));
    }
    return MP_OK;
}

int mp_set_velocity_state(mp_handle* h, const float* state_dev, int batch) {
    if (!h || batch < 0 || (batch > 0 && !state_dev)) return MP_ERR_INVALID;
    ON_DEVICE(h);
    if (batch == 0) { h->vstate.B = 0; return MP_OK; }
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    if (int rc = ensure_vstate(h, h->vstate, batch)) return rc;
    const size_t n = (size_t)2 * batch * 256;
    HIPCHK(h, hipMemcpy(h->vstate.h, state_dev, n * sizeof(float), hipMemcpyDeviceToDevice));
    HIPCHK(h, hipMemcpy(h->vstate.c, state_dev + n, n * sizeof(float), hipMemcpyDeviceToDevice));
    h->vstate.B = batch;
    return MP_OK;
}

int mp_set_lstm_mode(mp_handle* h, int mode) {
    if (!h || mode < 0 || mode > 3) return MP_ERR_INVALID;
    ON_DEVICE(h);
    HIPCHK(h, hipDeviceSynchronize());
    h->persist = mode != 0;              // (mode 2, rounds 1-4's separate two-layer wavefront kernel: since round 5 the velocity
    h->x3 = mode == 3;                   //  block of mode 1's full-batch schedule IS a two-layer wavefront -- 2 means 1)
    return MP_OK;
}

int mp_set_graph_mode(mp_handle* h, int on) {
    if (!h || on < 0 || on > 2) return MP_ERR_INVALID;
    ON_DEVICE(h);
    h->use_graph = on != 0;
    h->graph_serial = on == 2 || (on == 1 && !multibranch_graphs_allowed());
    return MP_OK;
}

}  // extern "C"

// forward_online as a service (models/net.py:173-219, live_demo.py:207-264): S concurrent streams per handle, one tick = one
// new frame per stream (mp_stream_step), N calls of one stream as one call (mp_stream_replay, evaluate.py:62-64), per-stream state.
#include "mp_host.h"

// ================================================================================================ C ABI
extern "C" {

// ------------------------------------------------------------------------------------------ streaming
int mp_stream_create(mp_handle* h, int S) {
    if (!h || S < 1) return h ? fail(h, MP_ERR_INVALID, "mp_stream_create: S must be positive") : MP_ERR_INVALID;
    if (int rc = need_weights(h, "mp_stream_create")) return rc;
    ON_DEVICE(h);
    StreamCtx& c = h->sc;
    if (c.S) return fail(h, MP_ERR_INVALID, "streams already created (S = %d)", c.S);
    const int W = 45;
    if (int rc = dev_alloc(h, (void**)&c.window, (size_t)S * W * 60 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&c.fresh, S)) return rc;
    if (int rc = dev_alloc(h, (void**)&c.mask_dev, S)) return rc;
    if (int rc = dev_alloc(h, (void**)&c.st.last_foot, (size_t)S * 6 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&c.st.root_y, (size_t)S * sizeof(double))) return rc;
    if (int rc = dev_alloc(h, (void**)&c.st.root_pos, (size_t)S * 3 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->st_snap.last_foot, (size_t)S * 6 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->st_snap.root_y, (size_t)S * sizeof(double))) return rc;
    if (int rc = dev_alloc(h, (void**)&h->st_snap.root_pos, (size_t)S * 3 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&c.joints, (size_t)S * W * 72 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&c.vel, (size_t)S * W * 72 * sizeof(float))) return rc;
    if (int rc = dev_alloc(h, (void**)&c.contact, (size_t)S * W * 2 * sizeof(float))) return rc;
    std::vector<float> lf((size_t)S * 6);
    for (int s = 0; s < S; ++s) memcpy(&lf[(size_t)s * 6], h->feet_pos, sizeof(h->feet_pos));   // net.py:59
    HIPCHK(h, hipMemcpy(c.st.last_foot, lf.data(), lf.size() * sizeof(float), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemset(c.fresh, 1, S));
    HIPCHK(h, hipMemset(c.st.root_y, 0, (size_t)S * sizeof(double)));
    HIPCHK(h, hipMemset(c.st.root_pos, 0, (size_t)S * 3 * sizeof(float)));
    c.S = S;
    Plan* p = nullptr;
    if (int rc = get_plan(h, S, W, &p)) return rc;
    p->streaming = true;
    std::vector<int32_t> len(S, W);
    if (int rc = upload_lengths(h, p, len.data())) return rc;
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    return MP_OK;
}

int mp_stream_step(mp_handle* h, const float* frames_dev, float* pose_dev, float* joints_dev, float* root_pos_dev,
                   float* contact_dev, void* stream) {
    if (!h) return MP_ERR_INVALID;
    StreamCtx& c = h->sc;
    if (!c.S) return fail(h, MP_ERR_NO_STREAMS, "mp_stream_step before mp_stream_create");
    if (!frames_dev || !pose_dev || !root_pos_dev || !contact_dev) return fail(h, MP_ERR_INVALID, "mp_stream_step: NULL buffer");
    const int S = c.S, W = 45, PAST = 40;
    // one velocity.rnn_state per model, shared by the batch and the online path (velocity.py:30)
    if (h->vstate.B != 0 && h->vstate.B != S)
        return fail(h, MP_ERR_STATE_SHAPE, "carried velocity state has batch %d, streaming has %d streams", h->vstate.B, S);
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    if (int rc = ensure_vstate(h, h->vstate, S)) return rc;
    Plan* p = nullptr;
    if (int rc = get_plan(h, S, W, &p)) return rc;
    {   // the (S,45) plan may have been used by mp_forward with other lengths in between
        std::vector<int32_t> len(S, W);
        if (int rc = upload_lengths(h, p, len.data())) return rc;
    }
    float* joints = joints_dev ? joints_dev : c.joints;
    const bool has_state = h->vstate.B == S;
    h->segs.clear(); h->ev_used = 0;
    {
        CopyJobs js;
        if (h->recovery) {      // the solver state of the tick (net.py:59-64): last foot positions, root height, root position
            js.add(h->st_snap.last_foot, c.st.last_foot, (size_t)S * 6 * sizeof(float));
            js.add(h->st_snap.root_y, c.st.root_y, (size_t)S * sizeof(double));
            js.add(h->st_snap.root_pos, c.st.root_pos, (size_t)S * 3 * sizeof(float));
        }
        if (int rc = snapshot_vstate(h, S, has_state, &js)) return rc;      // (+ the velocity state: one launch)
    }
    GraphKey key;
    memset(&key, 0, sizeof(key));
    key.kind = 1; key.B = S; key.T = W; key.flags = (has_state ? 1 : 0) | (h->persist ? 2 : 0) | (h->x3 ? 8 : 0);
    key.p[0] = frames_dev; key.p[1] = pose_dev; key.p[2] = joints; key.p[3] = root_pos_dev; key.p[4] = contact_dev;
    key.p[6] = h->vstate.h;
    auto net_and_solver = [&]() {
        // forward on the 45-frame window (net.py:178); pose only for index 40 (net.py:181)
        bool tail = false;
        if (int r = forward_body(h, p, c.window, pose_dev, S, (long)W * 96, (long)PAST * 96, joints, c.vel, c.contact,
                                 p->r6d, h->vstate, has_state, nullptr, nullptr, &tail)) return r;
        mp_launch_translate_online(joints, c.vel, c.contact, S, W, PAST, h->floor_y, c.st, root_pos_dev, contact_dev,
                                   h->s_main);                                                // net.py:186-208
        if (tail) HIPCHK(h, hipStreamWaitEvent(h->s_main, h->ev_x[3], 0));
        HIPCHK(h, hipGetLastError());
        return (int)MP_OK;
    };
    int rc;
    {
        SegScope whole(h, h->s_main, 3, 1);
        rc = run_maybe_graph(h, key, [&]() {
            mp_launch_window_push(c.window, frames_dev, c.fresh, S, W, h->s_main);               // net.py:175
            return net_and_solver();
        });
    }
    if (rc) return rc;
    // (a repaired tick does not push the frame again: the window already holds it -- only network and solver are redone)
    if (int rc2 = finish_or_recover(h, p, "mp_stream_step", [&]() {
            if (int r = restore_vstate(h, S, has_state)) return r;
            HIPCHK(h, hipMemcpyAsync(c.st.last_foot, h->st_snap.last_foot, (size_t)S * 6 * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(c.st.root_y, h->st_snap.root_y, (size_t)S * sizeof(double), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(c.st.root_pos, h->st_snap.root_pos, (size_t)S * 3 * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            return (int)MP_OK;
        }, net_and_solver)) return rc2;
    h->vstate.B = S;
    return leave(h, stream);
}

// N consecutive forward_online calls of a single stream as ONE call (round 5; evaluate.py:62-64 runs
// `[model.forward_online(f) for f in ...]`, T + 5 of them per sequence, each a full launch chain on a 1 x 45 batch).
// Three of the four blocks are stateless per window (net.py:103-114): joints, pose and foot contact of all N windows run as ONE
// N x 45 batch -- the windows are never materialised, window k is rows k+1 .. k+45 of the frame history, which a RowMap with
// strideB = strideT = 60 addresses in place.  The velocity block is not: every call runs its 45 steps ON the state the previous
// call left (velocity.py:45-48), i.e. the N calls together are one 2-layer LSTM over a single sequence of N * 45
// steps whose input is the stacked linear1 of the N windows -- computed in the batch, written in sequence order, then two layer
// launches at B = 1, T = N * 45.  Only index 40 of every window is needed behind the layers (net.py:181-187): pose's and
// velocity's linear2 / IK run on N rows.  The solver chain over the N frames is one serial kernel.
int mp_stream_replay(mp_handle* h, const float* frames_dev, int N, float* pose_dev, float* joints_dev, float* root_pos_dev,
                     float* contact_dev, void* stream) {
    if (!h) return MP_ERR_INVALID;
    StreamCtx& c = h->sc;
    if (!c.S) return fail(h, MP_ERR_NO_STREAMS, "mp_stream_replay before mp_stream_create");
    if (c.S != 1) return fail(h, MP_ERR_INVALID, "mp_stream_replay drives a single stream (S = %d)", c.S);
    if (!frames_dev || !pose_dev || !root_pos_dev || !contact_dev || N < 1) return fail(h, MP_ERR_INVALID, "mp_stream_replay: NULL buffer or N < 1");
    const int W = 45, PAST = 40;
    if ((long)N * W > 0x3fffffffL / 256) return fail(h, MP_ERR_INVALID, "mp_stream_replay: %d frames in one call is beyond the supported size; split it", N);
    if (h->vstate.B != 0 && h->vstate.B != 1)
        return fail(h, MP_ERR_STATE_SHAPE, "carried velocity state has batch %d, the replayed stream has 1", h->vstate.B);
    if (use_x3(h, h->mod[MP_MOD_VELOCITY]))      // (before anything is enqueued; the Python facade feeds the frames tick by tick in mode 3)
        return fail(h, MP_ERR_INVALID, "mp_stream_replay runs on exact-fp32 operands (LSTM mode 1 or 0)");
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    if (int rc = ensure_vstate(h, h->vstate, 1)) return rc;
    const bool has_state = h->vstate.B == 1;
    // workspaces: the batch plan (N windows x 45), the chain plan (1 sequence x N*45), history / index-40 rows
    Plan *pb = nullptr, *pc = nullptr;
    if (int rc = get_plan(h, N, W, &pb)) return rc;
    if (int rc = get_plan(h, 1, N * W, &pc, pb)) return rc;      // (pb is in use: neither the victim of this acquisition nor its result)
    {
        std::vector<int32_t> len(N, W);
        if (int rc = upload_lengths(h, pb, len.data())) return rc;
        const int32_t one = N * W;
        if (int rc = upload_lengths(h, pc, &one)) return rc;
    }
    const size_t need = ((size_t)(W + N) * 60 + (size_t)N * 72 + (size_t)N * W * 72 + (size_t)N * W * 2) * sizeof(float);
    if (need > c.replay_bytes) {
        HIPCHK(h, hipStreamSynchronize(h->s_main));
        if (c.replay_ws) (void)hipFree(c.replay_ws);
        c.replay_ws = nullptr; c.replay_bytes = 0;
        if (int rc = dev_alloc(h, (void**)&c.replay_ws, need)) return rc;
        c.replay_bytes = need;
    }
    float* hist = c.replay_ws;
    float* vel40 = hist + (size_t)(W + N) * 60;
    float* joints_own = vel40 + (size_t)N * 72;
    float* contact_b = joints_own + (size_t)N * W * 72;
    float* joints = joints_dev ? joints_dev : joints_own;
    h->segs.clear(); h->ev_used = 0;
    {
        CopyJobs js;
        if (h->recovery) {
            js.add(h->st_snap.last_foot, c.st.last_foot, 6 * sizeof(float));
            js.add(h->st_snap.root_y, c.st.root_y, sizeof(double));
            js.add(h->st_snap.root_pos, c.st.root_pos, 3 * sizeof(float));
        }
        if (int rc = snapshot_vstate(h, 1, has_state, &js)) return rc;
    }
    mp_launch_replay_history(c.window, c.fresh, frames_dev, N, W, hist, h->s_main);
    const RowMap none{nullptr, 0, 0, 0};
    auto body = [&]() -> int {
        hipStream_t sm = h->s_main;
        const RowMap xi{hist + 60, 60, 60, 60};                            // window k, frame i = history row k + 1 + i
        const RowMap xj = user_map(joints, W, 72);
        RnnJob J{h, pb, MP_MOD_JOINTS, xi, none, joints, (long)W * 72, 72, STATE_ZERO, nullptr, nullptr, nullptr, nullptr};
        RnnJob P{h, pb, MP_MOD_POSE, xj, xi, pb->r6d, (long)W * 96, 96, STATE_ZERO, nullptr, nullptr, nullptr, nullptr};
        RnnJob F{h, pb, MP_MOD_FOOT_CONTACT, xj, xi, contact_b, (long)W * 2, 2, STATE_ZERO, nullptr, nullptr, nullptr, nullptr};
        if (int r = run_rnn(J, sm)) return r;                              // net.py:103
        if (int r = run_rnn(P, sm)) return r;                              // net.py:106-107
        mp_launch_r6d_ik_strided(pb->r6d, N, (long)W * 96, (long)PAST * 96, pose_dev, h->parent_dev, sm);   // net.py:110,181
        if (int r = run_rnn(F, sm)) return r;                              // net.py:113-114
        // velocity (net.py:117): linear1 of every window in the batch, rows written in (window, frame) order = the chain's time order
        const ModuleW& mv = h->mod[MP_MOD_VELOCITY];
        ModuleWS& wc = pc->ws[MP_MOD_VELOCITY];
        float* X1 = x1_buffer(h, mv, wc);
        run_gemm(h, sm, xj, xi, mv.lin1, X1, (long)W * mv.H, mv.H, N * W, N, 1);
        if (!h->persist && !wc.xproj) return fail(h, MP_ERR_INVALID, "internal: per-step workspace missing");
        if (!h->persist) run_gemm(h, sm, internal_map(X1, 1, mv.H), none, mv.ih[0], wc.xproj, 4 * mv.H, (long)4 * mv.H, N * W, 1, 0);
        RnnJob V{h, pc, MP_MOD_VELOCITY, none, none, nullptr, 0, 0, has_state ? STATE_FROM : STATE_ZERO, h->vstate.h, h->vstate.c, h->vstate.h, h->vstate.c};
        if (!h->persist) {       // per-step kernels keep their state in the plan's buffers: stage it in and out
            for (int l = 0; l < 2; ++l) {
                const size_t n = (size_t)mv.H * sizeof(float);
                if (has_state) {
                    HIPCHK(h, hipMemcpyAsync(wc.hbuf[l][0], h->vstate.h + (size_t)l * mv.H, n, hipMemcpyDeviceToDevice, sm));
                    HIPCHK(h, hipMemcpyAsync(wc.cbuf[l][0], h->vstate.c + (size_t)l * mv.H, n, hipMemcpyDeviceToDevice, sm));
                } else {
                    HIPCHK(h, hipMemsetAsync(wc.hbuf[l][0], 0, n, sm));
                    HIPCHK(h, hipMemsetAsync(wc.cbuf[l][0], 0, n, sm));
                }
            }
        }
        if (int r = rnn_rec(V, 0, sm)) return r;
        if (int r = rnn_g1(V, sm)) return r;
        if (int r = rnn_rec(V, 1, sm)) return r;
        if (!h->persist) {
            const size_t fin = (size_t)((N * W) & 1) * mv.H;
            for (int l = 0; l < 2; ++l) {
                const size_t n = (size_t)mv.H * sizeof(float);
                HIPCHK(h, hipMemcpyAsync(h->vstate.h + (size_t)l * mv.H, wc.hbuf[l][0] + fin, n, hipMemcpyDeviceToDevice, sm));
                HIPCHK(h, hipMemcpyAsync(h->vstate.c + (size_t)l * mv.H, wc.cbuf[l][0], n, hipMemcpyDeviceToDevice, sm));
            }
        }
        // linear2 on row 40 of every window only (net.py:196 reads nothing else)
        run_gemm(h, sm, RowMap{wc.out1 + (size_t)PAST * mv.H, (long)W * mv.H, 0, mv.H}, none, mv.lin2, vel40, 72, 0, N, N, 0);
        mp_launch_translate_replay(joints, vel40, contact_b, N, W, PAST, h->floor_y, c.st, root_pos_dev, contact_dev, sm);   // net.py:186-208
        HIPCHK(h, hipGetLastError());
        return (int)MP_OK;
    };
    int rc;
    {
        SegScope whole(h, h->s_main, 3, 1);
        rc = body();
    }
    if (rc) return rc;
    if (int rc2 = finish_or_recover(h, pb, "mp_stream_replay", [&]() {
            if (int r = ensure_step_ws(h, pc)) return r;
            if (int r = restore_vstate(h, 1, has_state)) return r;
            HIPCHK(h, hipMemcpyAsync(c.st.last_foot, h->st_snap.last_foot, 6 * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(c.st.root_y, h->st_snap.root_y, sizeof(double), hipMemcpyDeviceToDevice, h->s_main));
            HIPCHK(h, hipMemcpyAsync(c.st.root_pos, h->st_snap.root_pos, 3 * sizeof(float), hipMemcpyDeviceToDevice, h->s_main));
            return (int)MP_OK;
        }, body)) return rc2;
    mp_launch_replay_window(hist, N, W, c.window, c.fresh, h->s_main);   // net.py:175: the stream's window after the last call
    HIPCHK(h, hipGetLastError());
    h->vstate.B = 1;
    return leave(h, stream);
}

int mp_stream_reset(mp_handle* h, const uint8_t* mask_host, int clear_velocity) {
    if (!h) return MP_ERR_INVALID;
    StreamCtx& c = h->sc;
    if (!c.S) return fail(h, MP_ERR_NO_STREAMS, "mp_stream_reset before mp_stream_create");
    ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    if (mask_host) HIPCHK(h, hipMemcpy(c.mask_dev, mask_host, c.S, hipMemcpyHostToDevice));
    const bool vel = clear_velocity && h->vstate.B == c.S;
    mp_launch_stream_reset(mask_host ? c.mask_dev : nullptr, c.fresh, c.st.root_y, c.st.root_pos, vel ? h->vstate.h : nullptr,
                           vel ? h->vstate.c : nullptr, c.S, h->s_main);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    return MP_OK;
}

int mp_live_form_frames(mp_handle* h, const float* quat_dev, const float* acc_dev, const float* smpl2imu_dev,
                        const float* device2bone_dev, const float* acc_offsets_dev, unsigned keep_mask, int S,
                        float* frames_dev, void* stream) {
    if (!h || !quat_dev || !acc_dev || !smpl2imu_dev || !device2bone_dev || !acc_offsets_dev || !frames_dev || S < 1)
        return h ? fail(h, MP_ERR_INVALID, "mp_live_form_frames: bad argument") : MP_ERR_INVALID;
    ON_DEVICE(h);
    if (int rc = enter(h, stream)) return rc;
    mp_launch_live_frames(quat_dev, acc_dev, smpl2imu_dev, device2bone_dev, acc_offsets_dev, keep_mask, 30.0f /* config.py:74 */,
                          S, frames_dev, h->s_main);
    HIPCHK(h, hipGetLastError());
    return leave(h, stream);
}

int mp_stream_get_state(mp_handle* h, int s, float* window_dev, float last_foot_host[6], double* root_y_host,
                        float root_pos_host[3], int* fresh_host) {
    if (!h) return MP_ERR_INVALID;
    StreamCtx& c = h->sc;
    if (!c.S) return fail(h, MP_ERR_NO_STREAMS, "mp_stream_get_state before mp_stream_create");
    if (s < 0 || s >= c.S) return fail(h, MP_ERR_INVALID, "mp_stream_get_state: stream %d outside 0..%d", s, c.S - 1);
    ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    if (window_dev)
        HIPCHK(h, hipMemcpy(window_dev, c.window + (size_t)s * 45 * 60, (size_t)45 * 60 * sizeof(float), hipMemcpyDeviceToDevice));
    if (last_foot_host) HIPCHK(h, hipMemcpy(last_foot_host, c.st.last_foot + (size_t)s * 6, 6 * sizeof(float), hipMemcpyDeviceToHost));
    if (root_y_host) HIPCHK(h, hipMemcpy(root_y_host, c.st.root_y + s, sizeof(double), hipMemcpyDeviceToHost));
    if (root_pos_host) HIPCHK(h, hipMemcpy(root_pos_host, c.st.root_pos + (size_t)s * 3, 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (fresh_host) {
        uint8_t f = 0;
        HIPCHK(h, hipMemcpy(&f, c.fresh + s, 1, hipMemcpyDeviceToHost));
        *fresh_host = f;
    }
    return MP_OK;
}

int mp_stream_set_state(mp_handle* h, int s, const float* window_dev, const float last_foot_host[6], const double* root_y_host,
                        const float root_pos_host[3], const int* fresh_host) {
    if (!h) return MP_ERR_INVALID;
    StreamCtx& c = h->sc;
    if (!c.S) return fail(h, MP_ERR_NO_STREAMS, "mp_stream_set_state before mp_stream_create");
    if (s < 0 || s >= c.S) return fail(h, MP_ERR_INVALID, "mp_stream_set_state: stream %d outside 0..%d", s, c.S - 1);
    ON_DEVICE(h);
    HIPCHK(h, hipStreamSynchronize(h->s_main));
    if (window_dev)
        HIPCHK(h, hipMemcpy(c.window + (size_t)s * 45 * 60, window_dev, (size_t)45 * 60 * sizeof(float), hipMemcpyDeviceToDevice));
    if (last_foot_host) HIPCHK(h, hipMemcpy(c.st.last_foot + (size_t)s * 6, last_foot_host, 6 * sizeof(float), hipMemcpyHostToDevice));
    if (root_y_host) HIPCHK(h, hipMemcpy(c.st.root_y + s, root_y_host, sizeof(double), hipMemcpyHostToDevice));
    if (root_pos_host) HIPCHK(h, hipMemcpy(c.st.root_pos + (size_t)s * 3, root_pos_host, 3 * sizeof(float), hipMemcpyHostToDevice));
    if (fresh_host) {
        const uint8_t f = *fresh_host ? 1 : 0;
        HIPCHK(h, hipMemcpy(c.fresh + s, &f, 1, hipMemcpyHostToDevice));
    }
    return MP_OK;
}


}  // extern "C"
